"""Copy the rocprofv3 summaries of the last GPU run (gpurun_out/r01_*) into profiles/ (tracked)."""
import csv, glob, json, collections, shutil, os, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
src = newest(f'gpurun_out/{tag}_stats/runc/*_kernel_stats.csv')
shutil.copy(src, f'profiles/{tag}_bench_kernel_stats.csv')
rows = list(csv.DictReader(open(src)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(f'profiles/{tag}_bench_kernel_stats.md', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats -- python3 bench.py   ({tag}, 1x MI355X, default flags: 16384 windows, '
            '10 steps + 2 warm-up, stage profile, CPU baseline)\n\n')
    f.write('| kernel | calls | total ms | avg us | % of GPU time |\n|---|---|---|---|---|\n')
    for r in rows[:32]:
        f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
    f.write(f'\ntotal GPU kernel time {tot/1e6:.1f} ms\n')
line = [l for l in open(f'gpurun_out/{tag}_bench.log', errors='ignore') if l.startswith('{"metric"')][0]
open(f'profiles/{tag}_bench_line.json', 'w').write(line)
def pmc(path, name):
    rows = list(csv.DictReader(open(newest(path))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == name:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch = pmc(f'gpurun_out/{tag}_fetch/runc/*_counter_collection.csv', 'FETCH_SIZE')
write = pmc(f'gpurun_out/{tag}_write/runc/*_counter_collection.csv', 'WRITE_SIZE')
stages = json.loads([l for l in open(f'gpurun_out/{tag}_write.log', errors='ignore') if l.startswith('{"E"')][0])
out = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes, --kernel-trace only) -- python3 tools/stage_bench.py --windows 16384',
       'graph': {k: stages[k] for k in ('E', 'Dn', 'N')},
       'note': 'per-launch averages; KB as reported by rocprofv3 (x1024 = bytes), RAW. gfx950 tallies the 128-B requests of wide '
               'coalesced reads (16 B per lane) at 64 B (MI355X_MICROARCH.md, HBM section): bench.py prices reads as 2 x FETCH_SIZE '
               'and writes as WRITE_SIZE.',
       'kernels': {k[:90]: dict(FETCH_SIZE_KB=fetch[k], WRITE_SIZE_KB=write.get(k)) for k in fetch if 'tmpnn' in k}}
json.dump(out, open(f'profiles/{tag}_pmc_traffic_stage_kernels.json', 'w'), indent=1)
d = json.loads(line)
print(d['value'], d['ms_per_step'], d['roofline'], d['roofline_aggregation']['frac'], d['cpu_baseline']['value'])
