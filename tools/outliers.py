"""Dispatches of the tracker kernels above a duration threshold in a rocprofv3 --kernel-trace csv, with their neighbours in time
(which call of a bench run is a 30-ms k_track_retire?).  python tools/outliers.py DIR [threshold_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1e6
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')[:44]
cols = [c for c in ('Queue_Id', 'Stream_Id', 'Thread_Id', 'Dispatch_Id', 'Grid_Size_X', 'Workgroup_Size_X', 'LDS_Block_Size', 'Scratch_Size') if c in rows[0]]
n = 0
for i, r in enumerate(rows):
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if 'k_track' not in r['Kernel_Name'] or d < thr:
        continue
    n += 1
    print(f"--- {name(r)}: {d / 1e3:.1f} us at +{(int(r['Start_Timestamp']) - t0) / 1e6:.1f} ms  " + ' '.join(f'{c}={r[c]}' for c in cols))
    for j in range(max(0, i - 4), min(len(rows), i + 4)):
        q = rows[j]
        print(f"   {'>>' if j == i else '  '} +{(int(q['Start_Timestamp']) - t0) / 1e6:10.3f} ms .. +{(int(q['End_Timestamp']) - t0) / 1e6:10.3f} ms  "
              f"{(int(q['End_Timestamp']) - int(q['Start_Timestamp'])) / 1e3:9.1f} us  {name(q)}  " + ' '.join(f'{c}={q[c]}' for c in cols[:2]))
print(f'{n} tracker dispatches above {thr / 1e3:.0f} us of {len(rows)} dispatches')
