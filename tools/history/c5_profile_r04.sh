#!/bin/bash
# C5 on the GPU box (gpurun), round 4: un-profiled step + rocprofv3 kernel stats on one stream (default: every kernel's duration
# un-shared) and with the det-side branches on the auxiliary stream (TMPNN_WIDE_OVERLAP=1), then FETCH_SIZE / WRITE_SIZE passes in the single-stream form
# (a counter pass serialises the kernels anyway) -> gpurun_out/r04_c5/*   (tools/collect_r04.py writes profiles/r04_c5_dense_stress.md)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in overlap single; do
  if [ $mode = single ]; then unset TMPNN_WIDE_OVERLAP; else export TMPNN_WIDE_OVERLAP=1; fi
  python3 $R/tools/c5_bench.py --steps 4 > $O/${mode}_plain.log 2>&1 || { tail -5 $O/${mode}_plain.log; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -o r -- python3 $R/tools/c5_bench.py --steps 2 > $O/${mode}_stats.log 2>&1 || { tail -5 $O/${mode}_stats.log; exit 1; }
  cp $(ls $O/stats_$mode/*kernel_stats.csv $O/stats_$mode/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${mode}_kernel_stats.csv
  rm -rf $O/stats_$mode
  tail -1 $O/${mode}_plain.log
done
unset TMPNN_WIDE_OVERLAP
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o r --output-format csv -- python3 $R/tools/c5_bench.py --steps 1 > $O/pmc_$C.log 2>&1 || echo "pmc pass failed: $C"
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04_c5'
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
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
