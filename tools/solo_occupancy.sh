#!/bin/bash
# Search launch time against blocks per CU: one block of 8 row tiles per image (256 x 256), batches of
# 256 k images = k blocks per CU, every launch alone on the GPU, the batch kernel (no selection prologue).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CRIM_PIPE_MIN_BLOCKS=1 CRIM_PIPE_T=8 CRIM_SELECT_IN_SEARCH=0
for B in "$@"; do
  D=/tmp/so_$B.$$
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/solo_profile.py $B 60 > /dev/null 2>&1 || exit 1
  echo "B=$B $(python3 $R/tools/kstats.py $D/*/*kernel_stats.csv | grep 'k_search' | awk '{print $1,$2,$3,$(NF-1)}' | tr '\n' ' ')"
done
