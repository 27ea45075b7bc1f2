source scripts/r3_run.sh r3x
step twg 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad"
step b1 600 python bench.py --no-cpu-baseline --no-other-configs
step b0 600 env UIG_WGRAD_REDUCE4=0 python bench.py --no-cpu-baseline --no-other-configs
step c42 600 python bench.py --no-cpu-baseline --no-other-configs --force-comm
step c12 600 env UIG_DP_STAGES_G=1 python bench.py --no-cpu-baseline --no-other-configs --force-comm
step c11 600 env UIG_DP_STAGES_G=1 UIG_DP_STAGES_D=1 python bench.py --no-cpu-baseline --no-other-configs --force-comm
tail -3 gpurun_out/r3x_twg.log
python - <<'PY'
import json
for f in ("r3x_b1.log","r3x_b0.log","r3x_c42.log","r3x_c12.log","r3x_c11.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "roofline us", j["roofline"]["avg_us"], "comm", j.get("comm"))
    except Exception as e: print(f, "ERR", e)
PY
