source scripts/r3_run.sh r3n
step fp8t 900 python -m pytest tests/test_fp8_gpu.py -q -m gpu
step b5 600 python bench.py --config 5 --no-cpu-baseline
step b8bf 600 python bench.py --batch 8 --no-cpu-baseline --no-other-configs
tail -4 gpurun_out/r3n_fp8t.log
python - <<'PY'
import json
for f in ("r3n_b5.log","r3n_b8bf.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
