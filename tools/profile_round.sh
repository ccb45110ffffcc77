#!/bin/bash
# Round profile.  (1) the default bench command (all configs; the RCCL split workers are left out:
# a profiled process must not start other programs) under rocprofv3 kernel-trace/stats;
# (2) the headline alone (`--configs ""`), whose launch order lets the search launches be split into
# the bench's phases (graph replays / exclusive-GPU event pass / overlapped event pass);
# (3) the PMC passes.  Usage (on the GPU box): bash tools/profile_round.sh r02
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-split > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
cp $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_all_configs.csv
grep "^{\"metric\"" $R/gpurun_out/prof_$TAG.log > $R/gpurun_out/${TAG}_bench_under_rocprofv3.json
rm -f $R/gpurun_out/prof_$TAG/*/*kernel_trace.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profh_$TAG -- python3 $R/bench.py --configs "" --no-host-io --no-split --no-cpu-baseline > $R/gpurun_out/profh_$TAG.log 2>&1 || exit 1
cp $R/gpurun_out/profh_$TAG/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_headline.csv
grep "^{\"metric\"" $R/gpurun_out/profh_$TAG.log > $R/gpurun_out/${TAG}_bench_headline_under_rocprofv3.json
cut -c1-150 $R/gpurun_out/${TAG}_kernel_stats_headline.csv | head -12
PER_PASS=$(python3 -c "import json; print(json.load(open('$R/gpurun_out/${TAG}_bench_headline_under_rocprofv3.json'))['roofline']['launches'])")
python3 $R/tools/search_duration_by_phase.py $R/gpurun_out/profh_$TAG/*/*kernel_trace.csv $PER_PASS | tee $R/gpurun_out/${TAG}_search_duration_by_phase.txt
rm -f $R/gpurun_out/profh_$TAG/*/*kernel_trace.csv
bash $R/tools/pmc_passes.sh $R/gpurun_out/${TAG}_kernel_stats_headline.csv > $R/gpurun_out/${TAG}_pmc_totals.txt 2>&1 || exit 1
tail -25 $R/gpurun_out/${TAG}_pmc_totals.txt
