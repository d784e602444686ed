#!/bin/bash
# Round-5 measurements on the GPU box (through gpurun, one part per call); then here: python tools/collect_r05.py -> profiles/r05_*
#   bench  rocprofv3 kernel stats of bench.py; FETCH / WRITE passes over the stage kernels (stage_bench.py); the plain bench line;
#          the kernel table of the bench steps alone
#   c5     C5: plain step in the round-5 forms and in the round-3/4 ring forms, kernel stats, FETCH / WRITE passes
#   loops  the batch-1 loops (device / host Hungarian) + a kernel trace of them (the slowest dispatches of every tracker kernel)
set -o pipefail
R=$GRAFT_REPO_ROOT
T=r05
part=${1:-bench}
cd /tmp && export TMPDIR=/tmp
case $part in
bench)
  rm -rf $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write $R/gpurun_out/${T}_steps
  echo "[measure] stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 10 --warmup 2 > $R/gpurun_out/${T}_bench.log 2> $R/gpurun_out/${T}_bench.err || { tail -5 $R/gpurun_out/${T}_bench.err; exit 1; }
  echo "[measure] fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_fetch.log 2>&1 || exit 1
  echo "[measure] write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_write.log 2>&1 || exit 1
  echo "[measure] steps alone"; mkdir -p $R/gpurun_out/${T}_steps
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_steps/stats -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage-profile --no-latency --no-loops > $R/gpurun_out/${T}_steps/bench.log 2> $R/gpurun_out/${T}_steps/bench.err || { tail -5 $R/gpurun_out/${T}_steps/bench.err; exit 1; }
  cp $(ls $R/gpurun_out/${T}_steps/stats/*kernel_stats.csv $R/gpurun_out/${T}_steps/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${T}_steps/kernel_stats.csv
  rm -rf $R/gpurun_out/${T}_steps/stats
  cd $R
  echo "[measure] plain bench"; python3 bench.py > gpurun_out/${T}_bench_plain.json 2> gpurun_out/${T}_bench_plain.err || { tail -5 gpurun_out/${T}_bench_plain.err; exit 1; }
  find gpurun_out/${T}_stats gpurun_out/${T}_fetch gpurun_out/${T}_write -name '*kernel_trace.csv' -delete
  tail -c 300 gpurun_out/${T}_bench_plain.json
  ;;
c5)
  O=$R/gpurun_out/${T}_c5; rm -rf $O; mkdir -p $O
  python3 $R/tools/c5_bench.py --steps 4 > $O/pp_plain.log 2>&1 || { tail -5 $O/pp_plain.log; exit 1; }
  TMPNN_WIDE_FWD_RING=1 TMPNN_WIDE_GEMM_RING=1 python3 $R/tools/c5_bench.py --steps 4 > $O/ring_plain.log 2>&1 || { tail -5 $O/ring_plain.log; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o r -- python3 $R/tools/c5_bench.py --steps 2 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
  cp $(ls $O/st/*kernel_stats.csv $O/st/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv; rm -rf $O/st
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o r --output-format csv -- python3 $R/tools/c5_bench.py --steps 1 > $O/pmc_$C.log 2>&1 || echo "pmc pass failed: $C"
  done
  python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_c5'
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
  tail -1 $O/pp_plain.log | cut -c1-200; tail -1 $O/ring_plain.log | cut -c1-200
  ;;
loops)
  O=$R/gpurun_out/${T}_loops; rm -rf $O; mkdir -p $O
  cd $R
  python3 tools/loop_bench.py > $O/device.log 2>&1 || { tail -5 $O/device.log; exit 1; }
  TMPNN_HUNGARIAN_HOST=1 python3 tools/loop_bench.py > $O/host.log 2>&1 || { tail -5 $O/host.log; exit 1; }
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/tr -o r -- python3 $R/tools/loop_bench.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
  python3 - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_loops'
f = glob.glob(f'{O}/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    by[r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp']), i))
with open(O + '/slowest.txt', 'w') as out:
    for k, v in sorted(by.items(), key=lambda kv: -max(x[0] for x in kv[1])):
        if not k.startswith('k_track') and 'train_losses' not in k:
            continue
        v.sort(reverse=True)
        ds = sorted(x[0] for x in v)
        out.write(f'{k}: {len(v)} dispatches, median {ds[len(ds)//2]/1e3:.1f} us, mean {sum(ds)/len(ds)/1e3:.1f} us, max {ds[-1]/1e3:.1f} us\n')
        for d, i in v[:3]:
            r = rows[i]
            prev = rows[i - 1] if i else None
            gap = (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3 if prev else 0
            out.write(f"    {d/1e3:9.1f} us  dispatch #{i} at +{(int(r['Start_Timestamp']) - t0)/1e6:.1f} ms, grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')} LDS {r.get('LDS_Block_Size', '?')}; previous: {prev['Kernel_Name'].split('(')[0][-40:] if prev else '-'} (gap {gap:.1f} us)\n")
print(open(O + '/slowest.txt').read())
PY
  rm -rf $O/tr
  # the greedy / Hungarian loops of the C2 sequence alone: GPU kernel time per timestep against the wall time per timestep
  for M in greedy hungarian; do
    python3 $R/tools/greedy_trace.py C2 $M > $O/wall_$M.log 2>&1 || { tail -5 $O/wall_$M.log; exit 1; }
    rocprofv3 --kernel-trace --output-format csv -d $O/tr_$M -o r -- python3 $R/tools/greedy_trace.py C2 $M > $O/trace_$M.log 2>&1 || { tail -5 $O/trace_$M.log; exit 1; }
  done
  python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_loops'
out = {}
for M in ('greedy', 'hungarian'):
    f = glob.glob(f'{O}/tr_{M}/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    wall = json.loads([l for l in open(f'{O}/wall_{M}.log') if l.startswith('{')][-1])
    prof = json.loads([l for l in open(f'{O}/trace_{M}.log') if l.startswith('{')][-1])
    nts = prof['sequences'] * prof['frames']
    by = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')[:48]
        by[k][0] += 1; by[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out[M] = dict(wall_ms_per_timestep=wall['ms_per_timestep'], profiled_ms_per_timestep=prof['ms_per_timestep'],
                  kernel_us_per_timestep=round(sum(v[1] for v in by.values()) / nts / 1e3, 2),
                  kernels={k: dict(per_timestep=round(v[0] / nts, 2), us_per_timestep=round(v[1] / nts / 1e3, 2), avg_us=round(v[1] / v[0] / 1e3, 2))
                           for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]})
json.dump(out, open(O + '/c2_timestep.json', 'w'), indent=1)
print(json.dumps({m: {k: v for k, v in d.items() if k != 'kernels'} for m, d in out.items()}))
PY
  rm -rf $O/tr_greedy $O/tr_hungarian
  ;;
*) echo "unknown part $part"; exit 2;;
esac
echo "[measure] $part done"
