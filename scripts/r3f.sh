source scripts/r3_run.sh r3f
export UIG_NORM_CONV=1
step normconv 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "norm_conv"
step bench1 600 python bench.py --no-cpu-baseline --no-other-configs
UIG_NORM_CONV=0 step bench0 600 python bench.py --no-cpu-baseline --no-other-configs
step bench1b 600 python bench.py --no-cpu-baseline --no-other-configs
tail -4 gpurun_out/r3f_normconv.log
python - <<'PY'
import json
for f in ("r3f_bench1.log","r3f_bench0.log","r3f_bench1b.log"):
    j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, "ms/step", j["ms_per_step"], "g_fwd ms", j["g_fwd"]["ms"], "roofline us", j["roofline"]["avg_us"], "strip family ms", j["strip_family"]["ms"])
PY
