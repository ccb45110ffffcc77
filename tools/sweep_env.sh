#!/bin/bash
# Unprofiled time per iteration of the single-image configs for each "VAR=value[,VAR=value]" argument.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  echo "=== $A"
  for CASE in "2048 9" "2048 15" "4096 9"; do
    set -- $CASE
    env ${A//,/ } timeout -k 10 120 python3 $R/tools/solo_profile.py 1 12000 $1 $2 || exit 1
  done
  env ${A//,/ } timeout -k 10 120 python3 $R/tools/dataset_run.py 5 || exit 1
done
