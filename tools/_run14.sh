cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/_t16_check.py 2>&1 | grep -v Warn | tail -3 &&
TILE_ROWS=16 timeout -k 10 200 python3 tools/recompute_ab.py kernels 2>&1 | grep "^{"
