source scripts/r3_run.sh r4b
step tstrip 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "strip or mirror or norm_conv"
step b1 600 python bench.py --no-cpu-baseline
step b8 600 env UIG_X=1 python bench.py --no-cpu-baseline --no-other-configs
tail -3 gpurun_out/r4b_tstrip.log
python - <<'PY'
import json
for f in ("b1","b8"):
    try:
        j=json.loads(open(f"gpurun_out/r4b_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"], "in_step", j["roofline"].get("in_step_frac"), "g_fwd", j["g_fwd"]["ms"], j["g_fwd"]["mfma_frac"]); print(j.get("other_configs"))
    except Exception as e: print(f, "ERR", e)
PY
