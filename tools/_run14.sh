cd $GRAFT_REPO_ROOT
for v in gstore gstore2 gstore gstore2; do
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_$v.so timeout -k 10 300 python3 tools/c5_bench.py --steps 3 | tail -1 | cut -c1-140
done
