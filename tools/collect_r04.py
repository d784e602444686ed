"""Round-4 summaries under profiles/ from the outputs of tools/measure_r04.sh (gpurun_out/r04_*):
  r04_bench_kernel_stats.{csv,md}  r04_bench_line.json  r04_bench_line_unprofiled.json  r04_pmc_traffic_stage_kernels.json
  r04_loops.md  r04_variants.md  (the shared parts of tools/collect_r03.py, run with the r04 tag)  and  r04_attention.md
(per-kernel table, byte model and PMC traffic of the attention stage; one KITTI window eager / captured)."""
import csv, glob, json, os, collections, runpy, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
os.environ['COLLECT_TAG'] = 'r04'
runpy.run_path(os.path.join(root, 'tools', 'collect_r03.py'), run_name='__main__')
tag = 'r04'
HBM = 8000.0


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def pmc(path, name):
    f = newest(path)
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats = newest(f'gpurun_out/{tag}_astats/**/*_kernel_stats.csv')
log = f'gpurun_out/{tag}_astats.log'
if stats and os.path.exists(log):
    line = json.loads([l for l in open(log) if l.startswith('{')][-1])
    rows = list(csv.DictReader(open(stats)))
    fetch = pmc(f'gpurun_out/{tag}_afetch/**/*_counter_collection.csv', 'FETCH_SIZE')
    write = pmc(f'gpurun_out/{tag}_awrite/**/*_counter_collection.csv', 'WRITE_SIZE')
    E, Dn, K, H = line['E'], line['Dn'], line['K'], line['H']
    with open(f'profiles/{tag}_attention.md', 'w') as f:
        f.write(f'''# The attention stage (SURVEY 8(a) row G) on the last call's graph of the C2 bench batch ({tag}, 1x MI355X)

`python tools/att_bench.py --windows 16384 --train` (K = {K} heads, H = {H}, train-mode dropout mask; E = {E:,} edge rows,
Dn = {Dn:,} dets): `tmpnn_att_fwd` / `tmpnn_att_bwd` through the C ABI, HIP events over {5} launches; the same command under
`rocprofv3 --kernel-trace --stats` (10 launches) for the kernel split and in two further passes (`--pmc FETCH_SIZE`, `--pmc
WRITE_SIZE`, kernel-trace only) for the HBM traffic.  Algorithmic bytes: `bench.att_bytes` (every array once; DESIGN section 12).

| stage | ms | algorithmic GB | GB/s | of 8 TB/s |
|---|---|---|---|---|
''')
        for k in ('att_fwd', 'att_bwd', 'segsum', 'segsum_adjoint'):
            v = line[k]
            f.write(f"| {k} | {v['ms']} | {v.get('GB', '')} | {v.get('GBs', '')} | {v.get('frac', '')} |\n")
        f.write('\n(`segsum` + `segsum_adjoint` are the two row movers the stage replaces when a model has heads.)\n\n')
        f.write('| kernel | launches | avg ms | PMC read GB (2 x FETCH_SIZE) | PMC write GB |\n|---|---|---|---|---|\n')
        for r in rows:
            n = r['Name']
            if 'tmpnn' not in n:
                continue
            short = n.split('(')[0].replace('void ', '').replace('tmpnn::', '')
            rd = next((2 * v * 1024 / 1e9 for k, v in fetch.items() if k.split('(')[0] == n.split('(')[0]), None)
            wr = next((v * 1024 / 1e9 for k, v in write.items() if k.split('(')[0] == n.split('(')[0]), None)
            f.write(f"| `{short[:60]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {'%.2f' % rd if rd is not None else ''} | "
                    f"{'%.2f' % wr if wr is not None else ''} |\n")
        f.write('''
Reading the table (DESIGN section 12 has the experiments behind it).  The det-owned forward pass (`k_att_fwd`) fetches every edge
row twice -- once per endpoint, as the plain segment sum does -- and that, not its arithmetic, is what it costs over the segment
sum it replaces.  The backward reads `h[e]` ONCE: everything of the adjoint that involves the edge row is edge-owned
(`k_att_bwd_edge`); the det-owned pass of the backward (`k_att_bwd_dha`) reads only the projected det table.  The three
projections (`k_rows_gemm_split` x 1 + K, `k_rows_outer`) run on the matrix pipe.
''')
    v = f'gpurun_out/{tag}_var/staged_window.log'
    if os.path.exists(v):
        with open(f'profiles/{tag}_attention.md', 'a') as f:
            f.write('\n## One KITTI-sized window (7 calls, fwd + loss + bwd) for the models outside the plain fused path (`tools/staged_window.py`)\n\n```\n')
            f.write(open(v).read())
            f.write('```\n')
    print('attention:', line['att_fwd'], line['att_bwd'])
else:
    print('attention part skipped (missing gpurun_out/r04_astats*)')
