#!/bin/bash
# C3 sweep on the GPU box (gpurun): HIP-event times for B in {1, 64, 1024, 16384}, then FETCH_SIZE / WRITE_SIZE passes per B.
#   -> gpurun_out/r03_c3/{times.jsonl, pmc.json}; tools/c3_sweep_report.py turns them into profiles/r03_c3_sweep.md
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_c3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/c3_sweep.py > $O/times.jsonl 2> $O/times.err || { tail -5 $O/times.err; exit 1; }
for B in 1 64 1024 16384; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $O/pmc_${C}_$B -o r --output-format csv -- python3 $R/tools/c3_sweep.py --one $B > $O/pmc_${C}_$B.log 2>&1 || { echo "pmc pass failed: $C $B"; tail -3 $O/pmc_${C}_$B.log; }
  done
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03_c3'
res = collections.defaultdict(dict)
for B in (1, 64, 1024, 16384):
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        acc = collections.defaultdict(list)
        for f in glob.glob(f'{O}/pmc_{C}_{B}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == C and ('k_gather_pipe' in r['Kernel_Name'] or 'k_segsum_pipe' in r['Kernel_Name']):
                    acc['gather' if 'k_gather_pipe' in r['Kernel_Name'] else 'segsum'].append(float(r['Counter_Value']))
        for k, v in acc.items():
            res[str(B)][f'{k}_{C}'] = dict(n=len(v), mean=sum(v) / len(v), max=max(v))
json.dump(res, open(O + '/pmc.json', 'w'), indent=1)
print(json.dumps(res)[:1500])
PY
# the directories of raw counter CSVs are large: keep only the summary and the logs' tails
rm -rf $O/pmc_FETCH_SIZE_* $O/pmc_WRITE_SIZE_*
