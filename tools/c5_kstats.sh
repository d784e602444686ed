#!/bin/bash
# per-kernel stats of the C5 step (rocprofv3 --kernel-trace --stats of tools/c5_bench.py --steps 2) -> gpurun_out/$1/kstats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_c5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o r -- python3 $R/tools/c5_bench.py --steps 2 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
f=$(ls $O/st/*kernel_stats.csv $O/st/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $O/kernel_stats.csv; rm -rf $O/st
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e6:8.3f} ms  tot {float(r['TotalDurationNs'])/1e6:8.1f} ms  {float(r['Percentage']):5.1f}%")
PY
