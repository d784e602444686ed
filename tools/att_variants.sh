#!/bin/bash
# usage: tools/att_variants.sh tag1 tag2 ...  -- tools/att_bench.py with libtmpnn_<tag>.so for each tag ("base" = the default library)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/att_variants
for t in "$@"; do
  if [ "$t" = base ]; then unset TMPNN_LIB_PATH; else export TMPNN_LIB_PATH=$R/trackmpnn_amd/lib/libtmpnn_$t.so; fi
  echo -n "$t: "
  timeout -k 10 200 python3 $R/tools/att_bench.py --windows ${WINDOWS:-16384} --train ${ATT_ARGS} 2> $R/gpurun_out/att_variants/$t.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join(f'{k}={v[\"ms\"]:.3f}' for k,v in d.items() if isinstance(v,dict)))"
done
