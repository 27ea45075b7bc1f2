source scripts/r3_run.sh r4u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u_tr; rm -rf $O; mkdir -p $O
step trace 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python scripts/trace_infer.py
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
python scripts/trace_infer.py summarize $O/kernel_trace.csv > gpurun_out/r4u_infer_trace.txt 2>&1
rm -rf $O
cat gpurun_out/r4u_infer_trace.txt
