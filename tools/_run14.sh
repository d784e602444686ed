cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 200 python3 tools/recompute_ab.py kernels 2>&1 | grep "^{"
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_dmaold.so timeout -k 10 200 python3 tools/recompute_ab.py kernels 2>&1 | grep "^{"
done
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "stage_checks or golden" 2>&1 | tail -2
