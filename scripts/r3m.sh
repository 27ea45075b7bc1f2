source scripts/r3_run.sh r3m
step s512 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "512_row or persistent or conv_fwd_bwd"
step m512 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "512"
step b512 600 python bench.py --config 4 --no-cpu-baseline
UIG_DEBUG_HOOKS=strip_wide=0 step b512old 600 python bench.py --config 4 --no-cpu-baseline
step b512b 600 python bench.py --config 4 --no-cpu-baseline
tail -5 gpurun_out/r3m_s512.log; tail -3 gpurun_out/r3m_m512.log
python - <<'PY'
import json
for f in ("r3m_b512.log","r3m_b512old.log","r3m_b512b.log"):
    try:
        j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, "ms/step", j["ms_per_step"], "img/s", j["value"], "g_fwd", j["g_fwd"]["ms"])
    except Exception as e: print(f, "ERR", e)
PY
