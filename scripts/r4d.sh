source scripts/r3_run.sh r4d
step n1 400 python bench.py --config 5 --no-cpu-baseline
step o1 400 env UIG_DEBUG_HOOKS=mx_issuers=8 python bench.py --config 5 --no-cpu-baseline
step n2 400 python bench.py --config 5 --no-cpu-baseline
step o2 400 env UIG_DEBUG_HOOKS=mx_issuers=8 python bench.py --config 5 --no-cpu-baseline
step bf 400 env UIG_DEBUG_HOOKS=strip_pk=23:0 python bench.py --batch 8 --no-cpu-baseline --no-other-configs
step bfn 400 python bench.py --batch 8 --no-cpu-baseline --no-other-configs
python - <<'PY'
import json
for f in ("n1","o1","n2","o2","bf","bfn"):
    try:
        j=json.loads(open(f"gpurun_out/r4d_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "frac", j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
