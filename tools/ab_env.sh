#!/bin/bash
# Headline bench (images/s only) for "<lib dir>[:VAR=value]" arguments, e.g. lib lib:CRIM_SELECT_IN_SEARCH=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  L=${A%%:*}; E=${A#*:}; [ "$E" = "$A" ] && E="X_UNUSED=1"
  echo "=== $A"
  env CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so $E timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --configs "" --no-cpu-baseline --no-host-io --no-roofline > /tmp/abe.json 2> /tmp/abe.err || { tail -3 /tmp/abe.err; exit 1; }
  python3 $R/tools/show_bench.py /tmp/abe.json | head -1
done
