set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/staged_window.py 2>&1 | grep 'captured\|fwd + loss'
bash tools/_sw.sh 2>&1 | grep 'per replay\|k_gru_fwd<\|k_gru_bwd_data'
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
