cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
for v in ${VARIANTS:-"" abl1 abl2 abl3}; do
  if [ -n "$v" ]; then export TMPNN_LIB_PATH=$GRAFT_REPO_ROOT/trackmpnn_amd/lib/libtmpnn_$v.so; else unset TMPNN_LIB_PATH; fi
  echo "== variant '$v'"
  timeout -k 10 200 python3 tools/wide_fwd_bench.py --only tiled 2>&1 | grep -E "^tiled|Error|error" 
done
