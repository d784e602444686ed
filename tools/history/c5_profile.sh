#!/bin/bash
# C5 on the GPU box (gpurun): rocprofv3 kernel stats of tools/c5_bench.py, FETCH_SIZE / WRITE_SIZE passes, an un-profiled
# step time and the CPU oracle on a bounded sample of the same window kind -> gpurun_out/r03_c5/*
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/c5_bench.py --steps 3 > $O/plain.log 2>&1 || { tail -5 $O/plain.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 $R/tools/c5_bench.py --steps 2 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o r --output-format csv -- python3 $R/tools/c5_bench.py --steps 1 > $O/pmc_$C.log 2>&1 || echo "pmc pass failed: $C"
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03_c5'
res = collections.defaultdict(dict)
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{O}/pmc_{C}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == C:
                acc[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res[k][C] = dict(n=len(v), mean=sum(v) / len(v), max=max(v))
json.dump(res, open(O + '/pmc.json', 'w'), indent=1)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/stats
cd $R
# CPU oracle on the host cores: the same static window kind at a size that finishes in ~20 s (T = 6 frames x 300 dets: E = 450 000)
timeout -k 10 300 python3 tools/c5_cpu.py > $O/cpu.json 2> $O/cpu.err || tail -3 $O/cpu.err
tail -2 $O/plain.log; cat $O/cpu.json
