cd $GRAFT_REPO_ROOT
for sh in "7 93" "3 300"; do
  set -- $sh
  timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames $1 --dets $2 2>&1 | grep "^E" || exit 1
  TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_gstore.so timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames $1 --dets $2 2>&1 | grep "^E" || exit 1
done
timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames 6 --dets 50 --hidden 128 2>&1 | grep "^E" && TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_gstore.so timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames 6 --dets 50 --hidden 128 2>&1 | grep "^E" || exit 1
timeout -k 10 300 python3 tools/c5_bench.py --steps 3 | tail -1 &&
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_dwlast.so timeout -k 10 300 python3 tools/c5_bench.py --steps 3 | tail -1
