source scripts/r3_run.sh r3y
step b0 600 python bench.py --no-cpu-baseline --no-other-configs
step c11n 600 env UIG_DP_STAGES_G=1 UIG_DP_STAGES_D=1 UIG_OVERLAP_UPDATE=0 python bench.py --no-cpu-baseline --no-other-configs --force-comm
step c42n 600 env UIG_OVERLAP_UPDATE=0 python bench.py --no-cpu-baseline --no-other-configs --force-comm
step c12n 600 env UIG_DP_STAGES_G=1 UIG_OVERLAP_UPDATE=0 python bench.py --no-cpu-baseline --no-other-configs --force-comm
step c11 600 env UIG_DP_STAGES_G=1 UIG_DP_STAGES_D=1 python bench.py --no-cpu-baseline --no-other-configs --force-comm
python - <<'PY'
import json
for f in ("b0","c11n","c42n","c12n","c11"):
    try:
        j=json.loads(open(f"gpurun_out/r3y_{f}.log").read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"])
    except Exception as e: print(f, "ERR", e)
PY
