source scripts/r3_run.sh r4r
step t 600 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -q -m gpu -k "multi_packer or quantize_multi or fused_mx"
step tm 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "golden or config1 or graph_equals or resume"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r_tr; rm -rf $O; mkdir -p $O
step trace 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
python scripts/summarize_trace.py $O/kernel_trace.csv gpurun_out/r4r_kernels.csv > gpurun_out/r4r_serial.txt 2>&1
rm -rf $O
tail -3 gpurun_out/r4r_t.log; tail -3 gpurun_out/r4r_tm.log; head -3 gpurun_out/r4r_serial.txt; grep -i "pack\|adam" gpurun_out/r4r_serial.txt
