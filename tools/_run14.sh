cd $GRAFT_REPO_ROOT
for v in fttl fttl_ns fttl_nd; do
echo "=== $v"
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_$v.so timeout -k 10 200 python3 tools/fwd_timeline.py 2>&1 | grep -v Warn | head -11
done
