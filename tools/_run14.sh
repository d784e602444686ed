cd $GRAFT_REPO_ROOT
for pad in 0 64 1088 16448 0 1088; do
PLANE_PAD=$pad timeout -k 10 200 python3 tools/recompute_ab.py kernels 2>&1 | grep "^{"
done
