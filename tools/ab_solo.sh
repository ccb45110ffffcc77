#!/bin/bash
# Per-kernel averages with every launch alone on the GPU (one lane, one stream), for each lib dir given
# (relative to image-processing_amd/).  Usage: tools/ab_solo.sh "<solo_profile.py args>" lib lib_vX ...
R=$GRAFT_REPO_ROOT
ARGS=${1:-"4096 200"}; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
  echo "=== $L"
  D=/tmp/abs_$L.$$
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/solo_profile.py $ARGS 2>&1 | grep -v "^W2\|^E2\|rocprof" | tail -2 || exit 1
  python3 $R/tools/kstats.py $D/*/*kernel_stats.csv
done
