source scripts/r3_run.sh r3o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o_strace; rm -rf $O; mkdir -p $O
step trace 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --config 5 --steps 3 --warmup 2 --no-cpu-baseline
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
python scripts/summarize_trace.py $O/kernel_trace.csv gpurun_out/r3o_fp8_step_serial_kernels.csv gpurun_out/r3o_fp8_step_small_layers.csv > gpurun_out/r3o_fp8_step_serial.txt 2>&1
rm -f $O/kernel_trace.csv
step fp8prof 900 bash scripts/r3_prof_fp8.sh r3o
head -40 gpurun_out/r3o_fp8_step_serial.txt
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3o_fp8_kernel_pmc.json"))
for k in ("avg_duration_us","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_INSTS_VALU","SQ_INSTS_MFMA","SQ_INSTS_SALU","SQ_INSTS_LDS","hbm_bytes_per_launch"): print(k, j.get(k))
PY
