"""Dispatch timeline of the steady-state greedy timestep from a rocprofv3 --kernel-trace csv of tools/greedy_trace.py:
per kernel of the repeating sequence, mean duration and mean gap to the previous dispatch (us)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')[:40]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
for i in range(1, len(rows)):
    k = name(rows[i])
    dur[k] += (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    gap[k] += (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
    cnt[k] += 1
tot = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print(f'{len(rows)} dispatches over {tot / 1e3:.1f} ms')
for k, n in cnt.most_common(12):
    print(f'{k:42s} n={n:6d}  dur {dur[k] / n:7.2f} us   gap before {gap[k] / n:7.2f} us')
