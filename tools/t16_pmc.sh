#!/bin/bash
# SQ counters of the H = 64 tiled forward kernels (32-row and 16-row forms) on the C2 stage graph
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/t16pmc
rm -rf $O; mkdir -p $O
for rows in 32 16; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  tag=r${rows}_$(echo $set | tr ' ' '_' | cut -c1-30)
  TILE_ROWS=$rows rocprofv3 --kernel-trace --pmc $set -d $O/$tag -o r --output-format csv -- python3 $R/tools/recompute_ab.py kernels > $O/$tag.log 2>&1 || echo "pass failed: $set"
done
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/t16pmc/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'fwd_split' in k:
            agg[k[:50]][r['Counter_Name']].append(float(r['Counter_Value']))
names=sorted(agg)
cs=sorted({c for k in agg for c in agg[k]})
print('%-28s'%'counter (mean over launches)', *['%22s'%n[7:29] for n in names])
for c in cs:
    print('%-28s'%c, *['%22.4g'%(sum(agg[n][c])/max(len(agg[n][c]),1)) for n in names])
PY
rm -rf $O/r*_SQ*/
