source scripts/r3_run.sh r4e
step v20a 400 env UIG_DEBUG_HOOKS=strip_pk=20:0 python bench.py --no-cpu-baseline --no-other-configs
step v23a 400 env UIG_DEBUG_HOOKS=strip_pk=23:0 python bench.py --no-cpu-baseline --no-other-configs
step v20b 400 env UIG_DEBUG_HOOKS=strip_pk=20:0 python bench.py --no-cpu-baseline --no-other-configs
step v23b 400 env UIG_DEBUG_HOOKS=strip_pk=23:0 python bench.py --no-cpu-baseline --no-other-configs
python - <<'PY'
import json
for f in ("v20a","v23a","v20b","v23b"):
    try:
        j=json.loads(open(f"gpurun_out/r4e_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"], "in_step", j["roofline"].get("in_step_frac"), "g_fwd", j["g_fwd"]["ms"], j["g_fwd"]["mfma_frac"])
    except Exception as e: print(f, "ERR", e)
PY
