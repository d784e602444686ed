#!/bin/bash
# SQ counters of the wide forward kernels on the C5 graph (gpurun): tools/wide_pmc.sh <kernel-substring> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-k_wide_gru_fwd}; shift
OUT=$R/gpurun_out/widepmc
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o r --output-format csv -- python3 $R/tools/wide_fwd_bench.py --reps 2 "$@" > $OUT/$tag.log 2>&1 || echo "pass failed: $set"
done
K=$K python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']; K=os.environ['K']
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/widepmc/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if K in k:
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   ',c, 'n=%d'%len(v), 'mean=%.5g'%(sum(v)/len(v)))
PY
