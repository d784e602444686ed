cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bwdta
rm -rf $O; mkdir -p $O
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_BUSY_max" "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $O/$tag -o r --output-format csv -- python3 $R/tools/stage_bench.py --windows 16384 > $O/$tag.log 2>&1 || echo "pass failed: $set"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/bwdta/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_gru_bwd_two<1' in k or 'k_segsum_pipe' in k or 'k_gru_bwd_data_split<64, 3' in k or 'k_gru_bwd_weights_split' in k or 'k_gather_pipe' in k:
            agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print('   %-36s n=%d mean=%.4g max=%.4g'%(c,len(v),sum(v)/len(v),max(v)))
PY
rm -rf $O/*/
