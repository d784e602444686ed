#!/bin/bash
# FETCH_SIZE calibration (tools/ubench/fetch_calib.hip) on the GPU box -> gpurun_out/fetch_calib.txt
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/fetch_calib.hip -o $O/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -o r -- $O/fetch_calib > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/fetch_calib'
acc = collections.defaultdict(list)
for f in glob.glob(O + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
B = 2 << 30
with open(O + '/../fetch_calib.txt', 'w') as out:
    for k, v in acc.items():
        kb = sum(v) / len(v)
        line = f'{k:28s} FETCH_SIZE {kb * 1024 / 1e9:7.3f} GB for {B / 1e9:.3f} GB read -> factor {B / (kb * 1024):.3f} ({len(v)} launches)'
        print(line); out.write(line + '\n')
PY
rm -rf $O/pmc
