cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03a/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03a/gputests.log
timeout -k 10 300 python3 tools/c5_bench.py --steps 3 > gpurun_out/r03a/c5.log 2>&1; echo "c5 rc=$?"; tail -2 gpurun_out/r03a/c5.log
