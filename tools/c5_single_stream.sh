#!/bin/bash
# C5 on the GPU box (gpurun): kernel stats of tools/c5_bench.py with the auxiliary stream on (default) and off
# (TMPNN_WIDE_OVERLAP=0: every kernel's duration un-shared) -> gpurun_out/r04_c5/{overlap,single}_kernel_stats.csv + step logs
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in overlap single; do
  if [ $mode = single ]; then export TMPNN_WIDE_OVERLAP=0; else unset TMPNN_WIDE_OVERLAP; fi
  python3 $R/tools/c5_bench.py --steps 3 > $O/${mode}_plain.log 2>&1 || { tail -5 $O/${mode}_plain.log; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -o r -- python3 $R/tools/c5_bench.py --steps 2 > $O/${mode}_stats.log 2>&1 || { tail -5 $O/${mode}_stats.log; exit 1; }
  cp $(ls $O/stats_$mode/*kernel_stats.csv $O/stats_$mode/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${mode}_kernel_stats.csv
  rm -rf $O/stats_$mode
  tail -1 $O/${mode}_plain.log
done
