source scripts/r3_run.sh r3j
step wide 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad"
step m512 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "512 or combined_pass"
step b512 600 python bench.py --config 4 --no-cpu-baseline
UIG_DEBUG_HOOKS=wgrad_rows=2 step b512old 600 python bench.py --config 4 --no-cpu-baseline
tail -5 gpurun_out/r3j_wide.log; tail -3 gpurun_out/r3j_m512.log
python - <<'PY'
import json
for f in ("r3j_b512.log","r3j_b512old.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"])
    except Exception as e: print(f, "ERR", e)
PY
