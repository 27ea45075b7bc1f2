source scripts/r3_run.sh r3u
export PK_VARIANTS="pk,pk 4iss,pk pkrt"
step pk 600 python scripts/bench_strip_pk.py
step fp8t 600 python -m pytest tests/test_fp8_gpu.py -q -m gpu -x
step b5 600 python bench.py --config 5 --no-cpu-baseline
step b8bf 600 python bench.py --batch 8 --no-cpu-baseline --no-other-configs
tail -14 gpurun_out/r3u_pk.log; tail -3 gpurun_out/r3u_pk.err; tail -2 gpurun_out/r3u_fp8t.log
python - <<'PY'
import json
for f in ("r3u_b5.log","r3u_b8bf.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
