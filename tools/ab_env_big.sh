#!/bin/bash
# configs[3] / configs[4] of the bench (single large images) for "<lib dir>[:VAR=value]" arguments
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  L=${A%%:*}; E=${A#*:}; [ "$E" = "$A" ] && E="X_UNUSED=1"
  echo "=== $A"
  for S in 2048 4096; do
    env CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so $E timeout -k 10 100 python3 $R/tools/solo_profile.py 1 6000 $S 9 || exit 1
  done
done
