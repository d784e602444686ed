#!/bin/bash
# k_wide_gru_fwd_pp at C5, complete and with parts removed at compile time (tools/build_variant.sh w3_<X> wide -DW3_<X>; wrong results)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pp; mkdir -p $O
for v in ${VARIANTS:-full w3_NOEPI w3_NOMMA w3_NOSTORE w3_NODMA w3_NOSPLIT}; do
  if [ "$v" != full ]; then export TMPNN_LIB_PATH=$R/trackmpnn_amd/lib/libtmpnn_$v.so; else unset TMPNN_LIB_PATH; fi
  timeout -k 10 200 python3 $R/tools/wide_fwd_bench.py --reps 10 --only tiled > $O/abl_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/abl_$v.log; exit 1; }
  echo "$v: $(grep '^tiled' $O/abl_$v.log | cut -c1-40)"
done
