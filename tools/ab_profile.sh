#!/bin/bash
# Per-kernel averages of the headline workload under rocprofv3 for each lib dir given (results of a
# build that changes what the kernels read may be wrong on purpose: timing experiments).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
  echo "=== $L"
  D=/tmp/abp_$L.$$
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 1 --warmup 1 --configs "" --no-cpu-baseline --no-roofline --no-host-io > /dev/null 2>&1 || exit 1
  python3 $R/tools/kstats.py $D/*/*kernel_stats.csv
done
