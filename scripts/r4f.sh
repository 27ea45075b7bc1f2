source scripts/r3_run.sh r4f
step n1 400 python bench.py --no-cpu-baseline --no-other-configs
step o1 400 env UIG_DEBUG_HOOKS=strip_pk=23:0 python bench.py --no-cpu-baseline --no-other-configs
step n2 400 python bench.py --no-cpu-baseline --no-other-configs
step o2 400 env UIG_DEBUG_HOOKS=strip_pk=23:0 python bench.py --no-cpu-baseline --no-other-configs
python - <<'PY'
import json
for f in ("n1","o1","n2","o2"):
    try:
        j=json.loads(open(f"gpurun_out/r4f_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"], "in_step", j["roofline"].get("in_step_frac"), "g_fwd", j["g_fwd"]["ms"], j["g_fwd"]["mfma_frac"])
    except Exception as e: print(f, "ERR", e)
PY
