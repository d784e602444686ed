#!/bin/bash
# usage: tools/one_pmc.sh  -- PMC passes over tools/bwd_variants.py for the one-pass backward kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/onepmc/$tag -o r --output-format csv -- python3 $R/tools/bwd_variants.py > $R/gpurun_out/onepmc_$tag.log 2>&1 || echo "pass failed: $set"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/onepmc/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'bwd_one' in k or 'bwd_two' in k or 'bwd_data_split' in k or 'bwd_weights_split' in k:
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   ',c, 'n=%d'%len(v), 'mean=%.4g'%(sum(v)/len(v)))
PY
