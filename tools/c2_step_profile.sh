#!/bin/bash
# Per-kernel GPU time of the C2 bench steps ALONE (no stage profile, latency or loop blocks): rocprofv3 kernel stats of
# `bench.py --steps 10 --warmup 2 --no-*` -> gpurun_out/r03_c2_steps/kernel_stats.csv + a per-step table on stdout
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_c2_steps
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage-profile --no-latency --no-loops > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp $(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<'PY'
import csv, json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03_c2_steps'
rows = list(csv.DictReader(open(O + '/kernel_stats.csv')))
line = json.loads([l for l in open(O + '/bench.log') if l.startswith('{')][-1])
steps = 12
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
print(f"bench (profiled): {line['ms_per_step']:.2f} ms/step; GPU kernel time {tot / steps:.2f} ms/step over {steps} steps")
for r in rows[:24]:
    ms = float(r['TotalDurationNs']) / 1e6
    print(f"{ms / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls/step  {r['Name'].split('(')[0][-60:]}")
PY
