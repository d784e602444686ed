cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 600 python3 -m pytest tests/test_tracking_gpu.py tests/test_small_path_gpu.py tests/test_dist_gloo.py -x -q -m gpu > gpurun_out/r03a/track_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03a/track_tests.log
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "split_products or stage_checks" > gpurun_out/r03a/parity_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03a/parity_tests.log
