#!/bin/bash
# Round profile: the default bench command under rocprofv3 kernel-trace/stats, then the PMC passes.
# Usage (on the GPU box): bash tools/profile_round.sh r01
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
cp $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
grep "^{\"metric\"" $R/gpurun_out/prof_$TAG.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
cut -c1-150 $R/gpurun_out/${TAG}_kernel_stats.csv | head -12
python3 $R/tools/search_duration_by_phase.py $R/gpurun_out/prof_$TAG/*/*kernel_trace.csv 562 | tee $R/gpurun_out/${TAG}_search_duration_by_phase.txt
bash $R/tools/pmc_passes.sh > $R/gpurun_out/${TAG}_pmc_totals.txt 2>&1 || exit 1
tail -25 $R/gpurun_out/${TAG}_pmc_totals.txt
