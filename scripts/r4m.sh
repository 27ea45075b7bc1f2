source scripts/r3_run.sh r4m
step mb 300 python scripts/bench_wgrad_s2.py 4
step n1 400 python bench.py --no-cpu-baseline --no-other-configs
step o1 400 env UIG_DEBUG_HOOKS=wgrad_rows_s2=0 python bench.py --no-cpu-baseline --no-other-configs
step n2 400 python bench.py --no-cpu-baseline --no-other-configs
step o2 400 env UIG_DEBUG_HOOKS=wgrad_rows_s2=0 python bench.py --no-cpu-baseline --no-other-configs
cat gpurun_out/r4m_mb.log
python - <<'PY'
import json
for f in ("n1","o1","n2","o2"):
    try:
        j=json.loads(open(f"gpurun_out/r4m_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"])
    except Exception as e: print(f, "ERR", e)
PY
