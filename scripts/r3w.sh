source scripts/r3_run.sh r3w
step twg 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad or strip128"
step b1 600 python bench.py --no-cpu-baseline --no-other-configs
step b0 600 env UIG_WGRAD_REDUCE4=0 python bench.py --no-cpu-baseline --no-other-configs
tail -5 gpurun_out/r3w_twg.log
python - <<'PY'
import json
for f in ("r3w_b1.log","r3w_b0.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"], "in_step", j["roofline"].get("in_step_frac"), "g_fwd", j["g_fwd"]["ms"])
    except Exception as e: print(f, "ERR", e)
PY
