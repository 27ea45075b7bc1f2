cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/r3_prof.sh r4p > gpurun_out/r4p_prof.log 2>&1
echo "prof rc=$?"
bash scripts/r3_prof_fp8.sh r4p > gpurun_out/r4p_prof_fp8.log 2>&1
echo "prof fp8 rc=$?"
tail -30 gpurun_out/r4p_prof.log
tail -40 gpurun_out/r4p_prof_fp8.log
cat gpurun_out/r4p_exit.log
