"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory: python tools/kstats.py DIR [N]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fs = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
if not fs:
    sys.exit('no kernel_stats.csv under ' + d)
for r in list(csv.DictReader(open(fs[0])))[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} {r.get('Percentage','')}")
