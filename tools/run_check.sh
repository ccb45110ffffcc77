#!/bin/bash
# GPU tests, then the per-kernel profile of the single-image configs.  Usage (GPU box): bash tools/run_check.sh <tag> [lib dir]
TAG=${1:-chk}
cd $GRAFT_REPO_ROOT
[ -n "$2" ] && export CRIM_HIP_LIB=$GRAFT_REPO_ROOT/image-processing_amd/$2/libcriminisi_hip.so
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$TAG.log 2>&1
RC=$?
echo "tests exit=$RC"; tail -15 gpurun_out/gpu_tests_$TAG.log
[ $RC -ne 0 ] && exit 1
bash tools/baseline_r03.sh $TAG ${2:-lib}
