#!/bin/bash
# Search-kernel time per launch (each alone on the GPU) against the batch size: tools/solo_sweep.sh lib B1 B2 ...
R=$GRAFT_REPO_ROOT
L=$1; shift
cd /tmp && export TMPDIR=/tmp
export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
for B in "$@"; do
  D=/tmp/sw_$B.$$
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/solo_profile.py $B 100 > /dev/null 2>&1 || exit 1
  echo "B=$B $(python3 $R/tools/kstats.py $D/*/*kernel_stats.csv | grep 'k_search\|k_priority\|k_update\|k_select' | awk '{print $1,$2,$3,$(NF-1)}' | tr '\n' ' ')"
done
