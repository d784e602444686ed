set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for n in 10 60; do
rm -rf /tmp/prof_cw$n
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cw$n -o r -- python3 $R/tools/_cw.py $n > /tmp/cw$n.log 2>&1
grep replays /tmp/cw$n.log
done
python3 - <<'PY'
import csv, glob, collections
def load(n):
    f = glob.glob('/tmp/prof_cw%d/**/*kernel_stats.csv' % n, recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load(10), load(60)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0.0)); c1, t1 = b[k]
    if c1 > c0: rows.append(((t1 - t0) / 50 / 1e3, (c1 - c0) / 50, k))
rows.sort(reverse=True)
print('per replay: kernels %.1f, GPU us %.1f' % (sum(r[1] for r in rows), sum(r[0] for r in rows)))
for us, n, k in rows[:40]: print('%8.1f us %6.1f x  %8.1f us each  %s' % (us, n, us / n, k.split('(')[0][-90:]))
PY
