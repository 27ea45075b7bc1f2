#!/bin/bash
# round-4 baseline: GPU suite with durations, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r4a_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r4a_rc.txt
timeout -k 10 300 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4a_rc.txt
tail -40 gpurun_out/r4a_tests.log
