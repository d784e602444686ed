cd $GRAFT_REPO_ROOT
for cfg in "main 1" "hplate 1" "main 0"; do
  set -- $cfg
  if [ "$1" != "main" ]; then export TMPNN_LIB_PATH=$GRAFT_REPO_ROOT/trackmpnn_amd/lib/libtmpnn_$1.so; else unset TMPNN_LIB_PATH; fi
  echo "== lib=$1 FWD_TILED=$2"
  TMPNN_FWD_TILED=$2 timeout -k 10 300 python3 tools/stage_bench.py 2>&1 | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['stages'].items() if 'fwd' in k or 'one' in k})"
done
