# rocprofv3 kernel trace of the bench (eager mode so every kernel is attributed); summary -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UIG_BENCH_SOFT_EXIT=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > gpurun_out/$TAG.log 2>&1
tail -1 gpurun_out/$TAG.log
cp gpurun_out/$TAG/*/*_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/$TAG
