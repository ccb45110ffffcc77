cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_base.log 2>&1; echo "tests exit=$?"; tail -3 gpurun_out/gpu_tests_base.log
bash tools/baseline_r03.sh base lib && timeout -k 10 120 python3 tools/stamps.py 1 2048 > gpurun_out/base/stamps_2048.txt 2>&1; tail -8 gpurun_out/base/stamps_2048.txt
