#!/bin/bash
# Baseline of the single-image / small-batch configs: GPU tests, in-kernel stamps at 2048^2, per-kernel
# averages of 2048^2 p9 / p15, 4096^2 and the 100-image set.  Usage (GPU box): bash tools/baseline_r03.sh <tag> [lib dir]
TAG=${1:-base}
L=${2:-lib}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
O=$R/gpurun_out/$TAG
mkdir -p $O
for CASE in "2048 9" "2048 15" "4096 9"; do
  set -- $CASE
  D=/tmp/k_$1_$2.$$
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/solo_profile.py 1 4000 $1 $2 2>&1 | grep "us per iteration" || exit 1
  python3 $R/tools/kstats.py $D/*/*kernel_stats.csv | tee $O/kstats_$1_p$2.txt
done
D=/tmp/k_ds.$$
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/dataset_run.py 2>&1 | grep "lock-step" || exit 1
python3 $R/tools/kstats.py $D/*/*kernel_stats.csv | tee $O/kstats_dataset.txt
# the same runs without the profiler: the time per iteration as the host sees it
for CASE in "2048 9" "2048 15" "4096 9"; do
  set -- $CASE
  timeout -k 10 120 python3 $R/tools/solo_profile.py 1 20000 $1 $2 | tee -a $O/unprofiled.txt || exit 1
done
timeout -k 10 120 python3 $R/tools/dataset_run.py 5 | tee -a $O/unprofiled.txt
