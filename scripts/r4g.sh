source scripts/r3_run.sh r4g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in new old; do
  O=gpurun_out/r4g_$v; rm -rf $O; mkdir -p $O
  if [ $v = old ]; then export UIG_DEBUG_HOOKS=strip_pk=23:0; else unset UIG_DEBUG_HOOKS; fi
  step trace_$v 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs
  cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
  python scripts/summarize_trace.py $O/kernel_trace.csv gpurun_out/r4g_${v}_kernels.csv > gpurun_out/r4g_${v}_serial.txt 2>&1
  rm -rf $O
done
paste <(head -34 gpurun_out/r4g_new_serial.txt | cut -c1-95) <(head -34 gpurun_out/r4g_old_serial.txt | cut -c1-60)
