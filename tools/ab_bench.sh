#!/bin/bash
# Timing-only A/B of library builds (results of experimental variants may be wrong on purpose).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
  echo "=== $L"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$L -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/ab_$L.log 2>&1 || exit 1
  grep -E "k_priority|k_search|k_update|k_select" $R/gpurun_out/ab_$L/*/*kernel_stats.csv | sed "s/, /;/" | cut -d, -f1,2,4
  grep -o "\"value\": [0-9.]*" $R/gpurun_out/ab_$L.log
done
