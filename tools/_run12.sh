cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
bash tools/_run11.sh 2>&1 | grep -E "ALL OK|BAD|input_tf=|worst" | tail -14
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03a/tfprof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-latency --no-loops --no-cpu-baseline --no-stage-profile > $GRAFT_REPO_ROOT/gpurun_out/r03a/tfprof.log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/r03a/tfprof/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-70s calls %5s total %8.3f ms avg %8.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03a/tfprof
