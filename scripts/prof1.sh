set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "golden_fp32" > gpurun_out/model3.log 2>&1 || true
grep -E "frac\(|passed|failed|Error" gpurun_out/model3.log | head -20
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_eager -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > gpurun_out/prof_r1_eager.log 2>&1
ls gpurun_out/prof_r1_eager/*/ | head
