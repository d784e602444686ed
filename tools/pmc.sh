#!/bin/bash
# usage: KERN="k_att" tools/pmc.sh <tag> <python script + args...>  -- SQ counter passes (own runs, kernel-trace only) over one command;
# prints per-kernel means of every counter for kernels whose name contains $KERN
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=${SETS:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY|SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM|SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"}
IFS='|' read -ra arr <<< "$SETS"
i=0
for set in "${arr[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r --output-format csv -- python3 "$@" > $O/p$i.log 2>&1 || echo "pass failed: $set"
done
KERN="${KERN:-k_att}" python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+os.environ.get('PMC_TAG','')
PY
python3 - "$O" "${KERN:-k_att}" <<'PY'
import csv, glob, sys, collections
O, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if kern in k:
            agg[k.split('(')[0][-50:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    print('   ' + '  '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(d.items())))
PY
rm -rf $O/p*/
