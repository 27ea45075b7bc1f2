source scripts/r3_run.sh r3p
step fp8t 900 python -m pytest tests/test_fp8_gpu.py -q -m gpu
step b5 600 python bench.py --config 5 --no-cpu-baseline
step b5nm 600 env UIG_MX_DGRAD_MIRROR=0 python bench.py --config 5 --no-cpu-baseline
tail -4 gpurun_out/r3p_fp8t.log
python - <<'PY'
import json
for f in ("r3p_b5.log","r3p_b5nm.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
