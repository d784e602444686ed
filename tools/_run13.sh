cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
for sh in "7 93" "3 300" "5 41"; do
  set -- $sh
  timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames $1 --dets $2 || exit 1
  TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_gstore.so timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames $1 --dets $2 || exit 1
done
timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames 6 --dets 50 --hidden 128 && TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_gstore.so timeout -k 10 120 python3 tools/wide_bwd_ab.py --frames 6 --dets 50 --hidden 128 || exit 1
timeout -k 10 300 python3 tools/c5_bench.py --steps 3 | tail -1 &&
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_gstore.so timeout -k 10 300 python3 tools/c5_bench.py --steps 3 | tail -1
