#!/bin/bash
# A/B of library builds on one box, headline workload only: for each lib dir given (relative to
# image-processing_amd/), one plain bench run (images/s) and one under rocprofv3 (per-kernel averages).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export CRIM_HIP_LIB=$R/image-processing_amd/$L/libcriminisi_hip.so
  echo "=== $L"
  timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --configs "" --no-cpu-baseline --no-host-io > $R/gpurun_out/abh_$L.json 2> $R/gpurun_out/abh_$L.err || exit 1
  python3 $R/tools/show_bench.py $R/gpurun_out/abh_$L.json | head -2
  D=/tmp/abh_$L.$$
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 1 --warmup 1 --configs "" --no-cpu-baseline --no-roofline --no-host-io > /dev/null 2>&1 || exit 1
  python3 $R/tools/kstats.py $D/*/*kernel_stats.csv
done
