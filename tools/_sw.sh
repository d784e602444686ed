set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/staged_window.py 2>&1 | grep 'K=2'
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
