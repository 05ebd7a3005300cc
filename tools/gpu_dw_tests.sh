cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_device_walk.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/dw_tests.log; tail -12 gpurun_out/dw_tests.log
