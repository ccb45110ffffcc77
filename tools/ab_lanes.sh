#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CRIM_LANES=3 timeout -k 10 300 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
for L in 1 2 4 8; do
  echo "=== lanes $L"
  CRIM_LANES=$L timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|ms_per_step\": [0-9.]*"
done
