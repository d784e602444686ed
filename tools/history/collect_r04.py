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


# ------------------------------------------------------------------------------------------------ C5 (tools/c5_profile_r04.sh)
def _jl(path, prefix='{"workload"'):
    return json.loads([l for l in open(path) if l.startswith(prefix)][-1])


c5dir = f'gpurun_out/{tag}_c5'
if all(os.path.exists(f'{c5dir}/{m}_{x}') for m in ('overlap', 'single') for x in ('plain.log', 'kernel_stats.csv')):
    import shutil
    shutil.copy(f'{c5dir}/single_kernel_stats.csv', f'profiles/{tag}_c5_kernel_stats_single_stream.csv')
    shutil.copy(f'{c5dir}/overlap_kernel_stats.csv', f'profiles/{tag}_c5_kernel_stats.csv')
    c5o, c5s = _jl(f'{c5dir}/overlap_plain.log'), _jl(f'{c5dir}/single_plain.log')
    pm = json.load(open(f'{c5dir}/pmc.json')) if os.path.exists(f'{c5dir}/pmc.json') else {}
    E, N, H = c5o['E'], c5o['N'], 256
    Dn = c5o.get('Dn', 15000)
    sp = c5o.get('seg_plan') or dict(T=0, I=0)
    P = 8 * sp['T'] + 16 * sp['I']
    alg = {
        'k_wide_gru_fwd_ring': dict(bytes=E * (4 * H + 4 * H + 16 * H + 12) + Dn * 12 * H, flops=2.0 * 6 * E * H * 3 * H,
                                    what='h in, h_out + 4 gate planes out, tile descriptors; P rows of the tile from LDS'),
        'k_wide_gemm_ring': dict(bytes=E * (12 * H + 4 * H + 4 * H + 4), flops=2.0 * 6 * E * 3 * H * H,
                                 what='d_gh (3H of the 4H image) in, d_h read + written'),
        'k_wide_dw': dict(bytes=E * (16 * H + 4 * H) / 2 + 0, flops=2.0 * 6 * E * 3 * H * H / 2,
                          what='two launches per iteration, each half of the rows: [dr dz dn dn.r] image + h in'),
        'k_wide_gates_bwd4': dict(bytes=E * (4 * H + 16 * H + 4 * H + 16 * H + 4), flops=0,
                                  what='d_hout, 4 gate planes, h in; the 4H gate-gradient image out'),
        'k_segsum_tiles': dict(bytes=E * 4 * H + P * 4 * H + 4 * 128 * sp['T'], flops=0,
                               what='E rows of H in ONCE, one partial row per (tile, src) and (item, dst) out, the tile row lists'),
        'k_segsum_pipe': dict(bytes=P * 4 * H + Dn * 4 * H + 4 * (P + Dn + 1), flops=0,
                              what='second pass: the partial rows in, det rows out'),
        'k_heads_fwd': dict(bytes=N * (4 * H + 8), flops=0, what='h_out in, logits + scores out'),
        'k_heads_bwd': dict(bytes=N * (4 * H + 16), flops=0, what='h_out in, dy out (the d_h term is folded into the cell backward)'),
    }
    MFMA_PEAK = 2500.0

    def table(f, path, with_pmc):
        rows = list(csv.DictReader(open(path)))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        f.write('| kernel | launches / step | avg ms | % of GPU time | algorithmic GB | GB/s | of 8 TB/s | bf16-MFMA TFLOP/s | of 2.5 PF |'
                + (' PMC traffic GB | traffic / algorithmic |' if with_pmc else '') + '\n')
        f.write('|---|---|---|---|---|---|---|---|---|' + ('---|---|' if with_pmc else '') + '\n')
        out = {}
        for r in rows[:12]:
            name = r['Name']
            key = next((k for k in alg if k in name), None)
            calls = int(r['Calls'])
            avg_ms = float(r['AverageNs']) / 1e6
            pct = 100 * float(r['TotalDurationNs']) / tot
            short = name.split('(')[0].replace('void ', '').replace('tmpnn::', '')[:44]
            if key is None:
                f.write(f"| `{short}` | {calls / 3:.1f} | {avg_ms:.3f} | {pct:.1f} | | | | | |" + (' | |' if with_pmc else '') + '\n')
                continue
            a = alg[key]
            gbs = a['bytes'] / 1e9 / (avg_ms / 1e3)
            tf = a['flops'] / 1e12 / (avg_ms / 1e3) if a['flops'] else None
            line = (f"| `{short}` | {calls / 3:.1f} | {avg_ms:.3f} | {pct:.1f} | {a['bytes'] / 1e9:.2f} | {gbs:.0f} | {gbs / HBM:.2f} | "
                    f"{'%.0f' % tf if tf else ''} | {'%.2f' % (tf / MFMA_PEAK) if tf else ''} |")
            if with_pmc:
                pk = next((v for k, v in pm.items() if key in k), None)
                tr = (2 * pk['FETCH_SIZE']['mean'] + pk['WRITE_SIZE']['mean']) * 1024 / 1e9 if pk and 'FETCH_SIZE' in pk and 'WRITE_SIZE' in pk else None
                line += f" {'%.2f' % tr if tr else ''} | {'%.2f' % (tr / (a['bytes'] / 1e9)) if tr else ''} |"
            f.write(line + '\n')
            out[key] = avg_ms
        return out

    with open(f'profiles/{tag}_c5_dense_stress.md', 'w') as f:
        f.write(f"""# C5 (BASELINE.json configs[4]) -- dense stress, {tag}, 1x MI355X

`bash tools/c5_profile_r04.sh` (tools/c5_bench.py): static 50-frame window, 300 dets/frame, H = 256, K = 0, diff, 4 MP iterations
(first call h_in=None with all 4.425 M rows new, then 3 empty-x calls), one backward of sum(logits): N = {N:,} rows, E = {E:,} edges.

| | ms / step (4 fwd + bwd) | graph-edges/s | effective TFLOP/s (36 H^2 per edge-iteration) | peak HBM |
|---|---|---|---|---|
| round 1 (f32-input MFMA, weights streamed from L2) | 622 | 28.4 M | 67.0 | 97.6 GB |
| round 2 (LDS-tiled bf16x6 GEMMs, det-side W_ih products) | 257 | 68.6 M | 161.9 | 114.7 GB |
| round 3 (edge tiles + ring kernels: LDS-DMA half steps, operands one step ahead) | 192-195 | 91 M | 211 | 114.7 GB |
| round 4 (segment sum that reads every edge row once; one stream, the default since) | **{c5s['ms_per_step']:.1f}** | **{c5s['edges_per_s'] / 1e6:.1f} M** | **{c5s['tflops']:.1f}** | {c5s['mem_GB']:.1f} GB |
| round 4, det-side branches on the auxiliary stream (`TMPNN_WIDE_OVERLAP=1`, round 3's default) | {c5o['ms_per_step']:.1f} | {c5o['edges_per_s'] / 1e6:.1f} M | {c5o['tflops']:.1f} | {c5o['mem_GB']:.1f} GB |

The plan of the single-read segment sum (`struct tmpnn_seg_plan`): T = {sp['T']:,} tiles of 8 src x 16 dst, I = {sp['I']:,} work items,
{P:,} partial rows of 1 KB ({P * 1024 / 1e9:.2f} GB next to {E * 1024 / 1e9:.2f} GB of edge rows).

## Per kernel, one stream (default): every duration un-shared

rocprofv3 --kernel-trace --stats of `tools/c5_bench.py --steps 2` (1 warm-up + 2 steps); FETCH_SIZE / WRITE_SIZE in two further
passes of the same single-stream form.  `achieved` = algorithmic bytes (every array once, fp32) / average duration, against 8 TB/s;
for the three matrix kernels also the MFMA-pipe fraction: 6 bf16 products per fp32 product (bf16x6, fp32-accurate) x 2 flops /
duration against the dense bf16 peak of 2.5 PFLOP/s.  PMC traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024; the gfx950 correction
for 16-byte-per-lane streams).

""")
        single = table(f, f'{c5dir}/single_kernel_stats.csv', True)
        f.write("""
## Per kernel with the det-side branches on the auxiliary stream (`TMPNN_WIDE_OVERLAP=1`)

Here the segment sums, the Dn-row `k_wide_gemm_store` and the node cell `k_gru_fwd<2, 0>` execute CONCURRENTLY with
`k_wide_gru_fwd_ring`, `k_wide_gemm_ring256` and `k_wide_dw2`: a duration includes the time the kernel shares the GPU, so the
fractions of the overlapped kernels understate what each reaches alone (the table above).

""")
        table(f, f'{c5dir}/overlap_kernel_stats.csv', False)
        f.write('\n(byte models: ' + '; '.join(f'`{k}`: {v["what"]}' for k, v in alg.items()) + ')\n')
        if 'k_segsum_tiles' in single and 'k_segsum_pipe' in single:
            f.write(f"""
## The segment sum, before and after

Round 3 (`k_segsum_pipe<false, 8, 16>` over the graph's CSR, every edge row fetched once per endpoint): 1.53 ms per launch alone,
PMC traffic 1.99 x the rows.  Round 4: `k_segsum_tiles` {single['k_segsum_tiles']:.3f} ms + `k_segsum_pipe<DUAL>`
{single['k_segsum_pipe']:.3f} ms = **{single['k_segsum_tiles'] + single['k_segsum_pipe']:.3f} ms** per launch (16 launches per step).
With the row movers this short the second stream no longer pays for itself: four alternating un-profiled runs each on one box: one
stream 186.6 / 186.4 / 185.3 / 188.7 ms per step, two streams 189.0 / 189.2 / 188.1 / 187.6; the pair of this profile (table
above) has the two forms the other way round by 1.3 ms -- a wash within the box-to-box and run-to-run spread, so the default is the
simpler one-stream form again.
""")
    print('c5 r04:', c5o['ms_per_step'], c5s['ms_per_step'])


# ------------------------------------------------------------------------------------------------ dispatches per timestep
ttdir = f'gpurun_out/{tag}_tt'
if all(os.path.exists(f'{ttdir}/{m}_{x}') for m in ('greedy', 'train') for x in ('stats.csv', 'plain.log')):
    with open(f'profiles/{tag}_timestep_dispatches.md', 'w') as f:
        f.write(f"""# Dispatches per model call of the two batch-1 loops ({tag}, 1x MI355X)

`bash tools/timestep_profile.sh`: `tools/timestep_trace.py` (the C2 loop shapes of `bench.py`'s `loop_batch1`: greedy inference over a
40-frame sequence, 39 model calls; one train chunk, 6 model calls, WITHOUT the optimizer step and without zeroing the gradients
between repetitions -- the 18 gradient accumulations per chunk at the end of the train list are an artefact of that) un-profiled for
the time, then under `rocprofv3 --kernel-trace --stats` (3 warm-up + 20 repetitions) for the dispatch counts.  Round 3 (`BENCH_r03`,
review): ~45 launches and two host reads per greedy timestep.

""")
        for m, ncall in (('greedy', 39), ('train', 6)):
            line = json.loads([l for l in open(f'{ttdir}/{m}_plain.log') if l.startswith('{')][-1])
            rows = list(csv.DictReader(open(f'{ttdir}/{m}_stats.csv')))
            div = 23 * ncall
            tot = sum(int(r['Calls']) for r in rows)
            f.write(f"## {m}: {line['ms_per_call']} ms per {'sequence' if m == 'greedy' else 'chunk'} = {line['ms_per_model_call']} ms per model call; "
                    f"{tot / div:.1f} dispatches per model call\n\n| kernel | per model call | avg us |\n|---|---|---|\n")
            for r in sorted(rows, key=lambda r: -int(r['Calls'])):
                c = int(r['Calls']) / div
                if c < 0.02:
                    continue
                short = r['Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')[:70]
                f.write(f"| `{short}` | {c:.2f} | {float(r['AverageNs']) / 1e3:.1f} |\n")
            f.write('\n')
    print('timestep dispatches written')
