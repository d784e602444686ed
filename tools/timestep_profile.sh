#!/bin/bash
# Dispatches per timestep of the two batch-1 loops (gpurun): rocprofv3 --kernel-trace --stats over tools/timestep_trace.py
# -> gpurun_out/r04_tt/{greedy,train}_stats.csv + logs   (tools/collect_r04.py writes profiles/r04_timestep_dispatches.md)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_tt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in greedy train; do
  python3 $R/tools/timestep_trace.py --mode $m --reps 50 > $O/${m}_plain.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -o r -- python3 $R/tools/timestep_trace.py --mode $m --reps 20 > $O/$m.log 2>&1
  cp $(ls $O/$m/*kernel_stats.csv $O/$m/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${m}_stats.csv
  rm -rf $O/$m
  tail -1 $O/${m}_plain.log
done
