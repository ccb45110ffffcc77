#!/bin/bash
# One GPU-box pass: parity tests, then a profiled short bench.  Usage: tools/gpu_check.sh <tag> [bench args]
TAG=${1:-x}; shift
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/gpu_tests_$TAG.log 2>&1
echo "tests exit=$?" | tee -a $R/gpurun_out/gpu_tests_$TAG.log
tail -4 $R/gpurun_out/gpu_tests_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
echo "bench exit=$?"
cat $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | cut -c1-140
rm -f $R/gpurun_out/prof_$TAG/*/*kernel_trace.csv
grep -o "\"value\": [0-9.]*\|\"frac\": [0-9.]*\|avg_launch_us\": [0-9.]*\|ms_per_step\": [0-9.]*" $R/gpurun_out/prof_$TAG.log
