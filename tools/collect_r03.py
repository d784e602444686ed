"""Turn the outputs of tools/measure_r03.sh (gpurun_out/r03_*) into the tracked summaries under profiles/:
  r03_bench_kernel_stats.{csv,md}  r03_bench_line.json  r03_bench_line_unprofiled.json  r03_pmc_traffic_stage_kernels.json
  r03_c3_sweep.md  r03_c5_dense_stress.md (+ r03_c5_kernel_stats.csv)  r03_variants.md  r03_loops.md
Parts whose inputs are missing are skipped with a note."""
import csv, glob, json, collections, shutil, os, sys

tag = os.environ.get('COLLECT_TAG', 'r03')      # (tools/collect_r04.py reuses the bench / PMC / variants parts)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
HBM = 8000.0      # GB/s, MI355X_MICROARCH.md


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def json_line(path, start='{"metric"'):
    return [l for l in open(path, errors='ignore') if l.startswith(start)][-1]


def have(*paths):
    miss = [p for p in paths if not glob.glob(p, recursive=True)]
    if miss:
        print('skipped (missing):', miss)
    return not miss


# ------------------------------------------------------------------------------------------------ bench.py
if have(f'gpurun_out/{tag}_stats/**/*_kernel_stats.csv', f'gpurun_out/{tag}_bench.log', f'gpurun_out/{tag}_bench_plain.json'):
    src = newest(f'gpurun_out/{tag}_stats/**/*_kernel_stats.csv')
    shutil.copy(src, f'profiles/{tag}_bench_kernel_stats.csv')
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    line = json_line(f'gpurun_out/{tag}_bench.log')
    open(f'profiles/{tag}_bench_line.json', 'w').write(line)
    d = json.loads(line)
    plain = json_line(f'gpurun_out/{tag}_bench_plain.json')
    open(f'profiles/{tag}_bench_line_unprofiled.json', 'w').write(plain)
    du = json.loads(plain)
    stage_kernel = {'gru_fwd_edge': 'k_gru_fwd_split_tiled<64, 8>', 'gru_bwd_one_edge': 'k_gru_bwd_two<1, 3, true>',
                    'gru_bwd_data_edge_folded': 'k_gru_bwd_data_split<64, 3, true>',
                    'gru_bwd_data_edge': 'k_gru_bwd_data_split<64, 1, false>',
                    'gru_bwd_weights_edge': 'k_gru_bwd_weights_split<1, 1>',
                    'gather_diff': 'k_gather_pipe<false, false', 'segsum': 'k_segsum_pipe<false'}
    with open(f'profiles/{tag}_bench_kernel_stats.md', 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2   ({tag}, 1x MI355X)\n\n')
        f.write('What the run contains: 12 steps of the C2 workload (16 384 windows; six forward calls on graphs of six sizes + '
                'one backward per step), the stage profile on the LAST call\'s graph '
                f"(E = {d['stage_graph']['E']}, 6 timed launches per stage kernel), the batch-1 latency block (`k_small_*`, "
                '`k_graph_from_coo`), the `loop_batch1` block (train chunks and inference sequences on graphs of a few hundred rows: '
                '`k_track_*`, `k_small_*`, the loss kernels) and the CPU baseline.  One row per kernel therefore averages several '
                'graph sizes: the duration a roofline fraction is computed from is the HIP-event time of the stage profile, which '
                'the largest dispatch of that kernel in the trace (MaxNs) confirms:\n\n')
        f.write('| stage (bench.py `stage_roofs`) | kernel | on the default step | HIP-event ms at the stage graph | rocprof MaxNs (ms) '
                '| algorithmic GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|\n')
        on_step = ('gru_fwd_edge', 'gru_bwd_one_edge', 'segsum')
        for st, k in stage_kernel.items():
            r = next((r for r in rows if k in r['Name']), None)
            if r is None or st not in d['stage_roofs']:
                continue
            s = d['stage_roofs'][st]
            f.write(f"| {st} | `{k}` | {'yes' if st in on_step else 'comparison only'} | {s['ms']} | {float(r['MaxNs'])/1e6:.3f} | "
                    f"{s['GBs']} | {s['hbm_frac']} |\n")
        f.write(f"\nbench line of this (profiled) run: {d['value']:.4g} graph-edges/s, {d['ms_per_step']:.2f} ms/step; dominant kernel "
                f"`{d['roofline']['kernel']}` {d['roofline']['ms']:.3f} ms = {d['roofline']['frac']:.3f} of 8 TB/s; aggregation "
                f"kernels {d['roofline_aggregation']['frac']:.3f}.  The un-profiled run of the same build on the same box "
                f"(`{tag}_bench_line_unprofiled.json`): {du['value']:.4g} graph-edges/s, {du['ms_per_step']:.2f} ms/step, dominant kernel "
                f"{du['roofline']['frac']:.3f}, aggregation {du['roofline_aggregation']['frac']:.3f}.\n\n")
        f.write('| kernel | calls | total ms | avg us | max us | % of GPU time |\n|---|---|---|---|---|---|\n')
        for r in rows[:40]:
            f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                    f"{float(r['MaxNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
        f.write(f'\ntotal GPU kernel time {tot/1e6:.1f} ms\n')
    # loops block of the un-profiled line, as a table
    lb = du.get('loop_batch1')
    if lb:
        with open(f'profiles/{tag}_loops.md', 'w') as f:
            f.write(f'# The reference\'s two per-timestep loops end to end ({tag}, 1x MI355X, `bench.py` block `loop_batch1`, un-profiled run)\n\n')
            f.write('`trackmpnn_amd.loops.train_chunk` (train.py:54-135: TrackGraph(train) -> model -> create_targets + CE + focal -> '
                    'one backward -> Adam) and `infer_sequence` (infer.py:35-87: TrackGraph(test) -> model -> decode, greedy and '
                    'Hungarian) on the sequences of the golden fixtures\' shapes; next to them the REAL reference\'s time for the same '
                    'loop on the same inputs, measured in the build container (8 cores, `profiles/r03_reference_loops_build_container.json`, '
                    '`oracle/time_reference_loops.py`).\n\n```json\n')
            f.write(json.dumps(lb, indent=1))
            f.write('\n```\n')
            if du.get('latency_batch1'):
                f.write('\n## One window through the drop-in call (`latency_batch1`: fwd + loss + bwd; eager / captured; models outside the '
                        'fused path in `c2_window_staged_models`)\n\n```json\n')
                f.write(json.dumps(du['latency_batch1'], indent=1))
                f.write('\n```\n')
            if du.get('hbm_measured'):
                f.write(f"\nDevice-copy rate of the same box, same run: {json.dumps(du['hbm_measured'])}\n")
    print('bench:', d['value'], d['ms_per_step'], d['roofline'], '| plain:', du['value'], du['ms_per_step'])

# ------------------------------------------------------------------------------------------------ PMC traffic of the stage kernels
if have(f'gpurun_out/{tag}_fetch/**/*_counter_collection.csv', f'gpurun_out/{tag}_write/**/*_counter_collection.csv'):
    def pmc(path, name):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(newest(path))):
            if r['Counter_Name'] == name:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
        return {k: sum(v) / len(v) for k, v in agg.items()}
    fetch = pmc(f'gpurun_out/{tag}_fetch/**/*_counter_collection.csv', 'FETCH_SIZE')
    write = pmc(f'gpurun_out/{tag}_write/**/*_counter_collection.csv', 'WRITE_SIZE')
    stages = json.loads(json_line(f'gpurun_out/{tag}_write.log', '{"E"'))
    out = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes, --kernel-trace only) -- python3 tools/stage_bench.py --windows 16384',
           'graph': {k: stages[k] for k in ('E', 'Dn', 'N')},
           'note': 'per-launch averages; KB as reported by rocprofv3 (x1024 = bytes), RAW. gfx950 tallies the 128-B requests of wide '
                   'coalesced reads (16 B per lane) at 64 B (MI355X_MICROARCH.md, HBM section): bench.py prices reads as 2 x FETCH_SIZE '
                   'and writes as WRITE_SIZE.',
           'kernels': {k[:90]: dict(FETCH_SIZE_KB=fetch[k], WRITE_SIZE_KB=write.get(k)) for k in fetch if 'tmpnn' in k}}
    json.dump(out, open(f'profiles/{tag}_pmc_traffic_stage_kernels.json', 'w'), indent=1)

# ------------------------------------------------------------------------------------------------ C3 sweep
if have(f'gpurun_out/{tag}_c3/times.jsonl', f'gpurun_out/{tag}_c3/pmc.json'):
    times = [json.loads(l) for l in open(f'gpurun_out/{tag}_c3/times.jsonl') if l.startswith('{')]
    pm = json.load(open(f'gpurun_out/{tag}_c3/pmc.json'))
    with open(f'profiles/{tag}_c3_sweep.md', 'w') as f:
        f.write(f'''# SURVEY 8(d) C3 sweep: the aggregation kernels at B in {{1, 64, 1024, 16 384}} windows ({tag}, 1x MI355X)

`bash tools/c3_sweep.sh` (tools/c3_sweep.py): SURVEY's C3 generator -- 12 frames, D_t ~ clip(Poisson(8), 1, 25), H = 64 -- the
LAST call's graph of B windows batched block-diagonally; `tmpnn_gather_diff_fwd` (row E: ns[e] = h[src] - h[dst], `k_gather_pipe`)
and `tmpnn_segsum_fwd` (row F: signed segment sum into the det rows, `k_segsum_pipe`), HIP events over 20 launches.  Algorithmic
bytes as bench.py prices them (every array once: 4H per edge row moved + 4H per det row + indices).  HBM traffic from two separate
`rocprofv3 --kernel-trace --pmc` passes per B (FETCH_SIZE, WRITE_SIZE; KB -> bytes x 1024; reads = 2 x FETCH_SIZE on gfx950 for
16-byte-per-lane streams, MI355X_MICROARCH.md), per launch.

| B | rows N | edges E | dets | state MB | kernel | ms | algorithmic MB | GB/s | of 8 TB/s | PMC read MB | PMC write MB | traffic / algorithmic |
|---|---|---|---|---|---|---|---|---|---|---|---|---|
''')
        for t in times:
            p = pm.get(str(t['B']), {})
            for k in ('gather', 'segsum'):
                rd = 2 * p.get(f'{k}_FETCH_SIZE', {}).get('mean', float('nan')) * 1024 / 1e6
                wr = p.get(f'{k}_WRITE_SIZE', {}).get('mean', float('nan')) * 1024 / 1e6
                f.write(f"| {t['B']} | {t['N']:,} | {t['E']:,} | {t['Dn']:,} | {t['state_MB']} | {k} | {t[k]['ms']} | {t[k]['alg_MB']} | "
                        f"{t[k]['GBs']} | {t[k]['hbm_frac']} | {rd:.1f} | {wr:.1f} | {(rd + wr) / t[k]['alg_MB']:.2f} |\n")
        f.write('''
Reading the table.  B = 1 and B = 64 are launch- and latency-bound (a 0.4 MB / 24 MB working set: the whole state sits in L2 /
Infinity Cache and a launch is 7-15 us); from B = 1024 on the state no longer fits and the kernels run at 0.42-0.50 of the HBM
roof.  The gather writes what it reads (traffic = 1.0-1.1 x algorithmic); the segment sum FETCHES each edge row twice -- once for
the src-side run and once for the dst-side run of the CSR -- and the two reads are issued from different XCDs, so the second one
is an L2 miss served by the Infinity Cache: PMC traffic 1.8 x algorithmic at B >= 1024.  A segment sum that reads each row once
was NOT built this round (DESIGN.md section 4 gives the reason: on window batches a block owns 30-40 edge rows per ~12 dets, so
partial sums written and re-read cost what the second read costs; an XCD-aware visiting order left the time unchanged in round 2).
''')

# ------------------------------------------------------------------------------------------------ C5
if have(f'gpurun_out/{tag}_c5/kernel_stats.csv', f'gpurun_out/{tag}_c5/plain.log', f'gpurun_out/{tag}_c5/pmc.json'):
    shutil.copy(f'gpurun_out/{tag}_c5/kernel_stats.csv', f'profiles/{tag}_c5_kernel_stats.csv')
    rows = list(csv.DictReader(open(f'gpurun_out/{tag}_c5/kernel_stats.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    c5 = json.loads(json_line(f'gpurun_out/{tag}_c5/plain.log', '{"workload"'))
    pm = json.load(open(f'gpurun_out/{tag}_c5/pmc.json'))
    E, N, Dn, H = c5['E'], c5['N'], 15000, 256
    steps_prof = 3                       # tools/c5_bench.py --steps 2 under the profiler: 1 warm-up + 2
    # algorithmic bytes / flops per launch of the E-row kernels (fp32; every array once)
    alg = {
        'k_wide_gru_fwd_ring': dict(bytes=E * (4 * H + 4 * H + 16 * H + 12) + Dn * 12 * H, flops=2.0 * 6 * E * H * 3 * H,
                                    what='h in, h_out + 4 gate planes out, tile descriptors; P rows of the tile from LDS'),
        'k_wide_gemm_ring': dict(bytes=E * (12 * H + 4 * H + 4 * H + 4), flops=2.0 * 6 * E * 3 * H * H,
                                 what='d_gh (3H of the 4H image) in, d_h read + written'),
        'k_wide_dw': dict(bytes=E * (16 * H + 4 * H) / 2 + 0, flops=2.0 * 6 * E * 3 * H * H / 2,
                          what='two launches per iteration, each half of the rows: [dr dz dn dn.r] image + h in'),
        'k_wide_gates_bwd4': dict(bytes=E * (4 * H + 16 * H + 4 * H + 16 * H + 4), flops=0,
                                  what='d_hout, 4 gate planes, h in; the 4H gate-gradient image out'),
        'k_segsum_pipe': dict(bytes=E * 4 * H + Dn * 4 * H + 4 * (2 * E + Dn + 1) + 2 * E, flops=0, what='E rows of H in, det rows out'),
        'k_gather_pipe': dict(bytes=E * 4 * H + Dn * 4 * H + 8 * E, flops=0, what='det rows in, E rows of H out'),
        'k_heads_fwd': dict(bytes=N * (4 * H + 8), flops=0, what='h_out in, logits + scores out'),
        'k_heads_bwd': dict(bytes=N * (4 * H + 4 * H + 12), flops=0, what='h_out in, d_h out'),
    }
    MFMA_PEAK = 2500.0     # TFLOP/s dense bf16 (MI355X_MICROARCH.md); bf16x6 = 6 MFMA products per fp32 product
    with open(f'profiles/{tag}_c5_dense_stress.md', 'w') as f:
        f.write(f'''# C5 (BASELINE.json configs[4]) -- dense stress, {tag}, 1x MI355X

`bash tools/c5_profile.sh` (tools/c5_bench.py): static 50-frame window, 300 dets/frame, H = 256, K = 0, diff, 4 MP iterations
(first call h_in=None with all 4.425 M rows new, then 3 empty-x calls), one backward of sum(logits): N = {N:,} rows, E = {E:,} edges.

| | ms / step (4 fwd + bwd) | graph-edges/s | effective TFLOP/s (36 H^2 per edge-iteration) | peak HBM |
|---|---|---|---|---|
| round 1 (f32-input MFMA, weights streamed from L2) | 622 | 28.4 M | 67.0 | 97.6 GB |
| round 2 (LDS-tiled bf16x6 GEMMs, det-side W_ih products) | 257 | 68.6 M | 161.9 | 114.7 GB |
| round 3 (edge tiles + ring kernels: LDS-DMA half steps, operands one step ahead) | **{c5['ms_per_step']:.0f}** | **{c5['edges_per_s']/1e6:.1f} M** | **{c5['tflops']:.1f}** | {c5['mem_GB']:.1f} GB |

## Per kernel (rocprofv3 --kernel-trace --stats; FETCH_SIZE / WRITE_SIZE in two further passes), per launch

`achieved` = algorithmic bytes (every array once, fp32) / average duration, against 8 TB/s; for the three matrix kernels also the
MFMA-pipe fraction: 6 bf16 products per fp32 product (bf16x6, fp32-accurate) x 2 flops / duration against the dense bf16 peak of
2.5 PFLOP/s -- the matrix pipe is what bounds them (their floor at the ~1.8 GHz the chip holds under MFMA load is 5.6 ms).
PMC traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024; gfx950 correction for 16-byte-per-lane streams).
Since the det-side branches run on a second stream (DESIGN 11.2), `k_segsum_pipe` / the Dn-row `k_wide_gemm_store` / `k_gru_fwd<2, 0>`
execute CONCURRENTLY with `k_wide_gru_fwd_ring`, `k_wide_gemm_ring` and `k_wide_dw`: a kernel's duration below includes the time it
shares the GPU (alone: ring forward 12.0, the 256-column ring GEMM ~9.1 + 1.7 for the adjoint it now carries, segment sum 1.5 ms), so the
fractions of overlapped kernels understate what each reaches alone.

| kernel | launches / step | avg ms | % of GPU time | algorithmic GB | GB/s | of 8 TB/s | bf16-MFMA TFLOP/s | of 2.5 PF | PMC traffic GB | traffic / algorithmic |
|---|---|---|---|---|---|---|---|---|---|---|
''')
        for r in rows[:12]:
            name = r['Name']
            key = next((k for k in alg if k in name), None)
            calls = int(r['Calls'])
            avg_ms = float(r['AverageNs']) / 1e6
            pct = 100 * float(r['TotalDurationNs']) / tot
            short = name.split('(')[0].replace('void ', '').replace('tmpnn::', '')[:44]
            if key is None:
                f.write(f"| `{short}` | {calls / steps_prof:.1f} | {avg_ms:.3f} | {pct:.1f} | | | | | | | |\n")
                continue
            a = alg[key]
            pk = next((v for k, v in pm.items() if key in k), None)
            traffic = (2 * pk['FETCH_SIZE']['mean'] + pk['WRITE_SIZE']['mean']) * 1024 / 1e9 if pk and 'FETCH_SIZE' in pk and 'WRITE_SIZE' in pk else None
            gbs = a['bytes'] / 1e9 / (avg_ms / 1e3)
            tf = a['flops'] / 1e12 / (avg_ms / 1e3) if a['flops'] else None
            f.write(f"| `{short}` | {calls / steps_prof:.1f} | {avg_ms:.3f} | {pct:.1f} | {a['bytes']/1e9:.2f} | {gbs:.0f} | {gbs / HBM:.2f} | "
                    f"{'%.0f' % tf if tf else ''} | {'%.2f' % (tf / MFMA_PEAK) if tf else ''} | {'%.2f' % traffic if traffic else ''} | "
                    f"{'%.2f' % (traffic / (a['bytes'] / 1e9)) if traffic else ''} |\n")
        f.write('\n(rows averaged over launches of different sizes -- `k_wide_gemm_store`, `k_segsum_pipe` -- carry no model; the byte '
                'models: ' + '; '.join(f'`{k}`: {v["what"]}' for k, v in alg.items()) + ')\n')
        if os.path.exists(f'gpurun_out/{tag}_c5/cpu.json'):
            cpu = json.load(open(f'gpurun_out/{tag}_c5/cpu.json'))
            ref = json.load(open('profiles/r03_reference_c5_extrapolation_build_container.json')) if os.path.exists('profiles/r03_reference_c5_extrapolation_build_container.json') else None
            f.write(f'''
## CPU numbers for C5 (labelled: measured sample / extrapolation)

* Oracle (torch-CPU fp32 restatement, `tools/c5_cpu.py`) on the GPU box's host, {cpu['threads']} threads: **{cpu['seconds_per_step']:.1f} s** per
  step on the SAMPLE "{cpu['sample']}" = {cpu['edges_per_s']:.0f} graph-edges/s; scaled linearly in E to C5 (the cell cost is
  linear in E): **{cpu['c5_step_seconds_extrapolated']:.0f} s per C5 step** (extrapolated) against {c5['ms_per_step'] / 1e3:.3f} s measured on the GPU.
''')
            if ref:
                f.write(f"* The reference itself cannot run C5 (dense N x N adjacency operands: 4.4 M^2 floats).  Its largest feasible static "
                        f"windows in the build container, timed and fitted (`oracle/time_reference_c5.py`, "
                        f"`profiles/r03_reference_c5_extrapolation_build_container.json`): {json.dumps(ref)[:700]}\n")
        if os.path.exists(f'gpurun_out/{tag}_c5/sq_counters.txt'):
            f.write('\n## SQ counters of the ring forward (tools/wide_pmc.sh, four separate --pmc passes, per launch)\n\n```\n')
            f.write(open(f'gpurun_out/{tag}_c5/sq_counters.txt').read()[-3000:])
            f.write('\n```\n')
    print('c5:', c5['ms_per_step'])

# ------------------------------------------------------------------------------------------------ variants, ragged, C3/C4 steps
if have(f'gpurun_out/{tag}_var/variants.log'):
    O = f'gpurun_out/{tag}_var'
    v = json.loads(json_line(f'{O}/variants.log', '{"base"'))
    with open(f'profiles/{tag}_variants.md', 'w') as f:
        f.write(f'''# SURVEY 8(d)(c): model variants on the batched C2 workload ({tag}, 1x MI355X)

`python tools/variants_bench.py` (4096 windows of the C2 generator, H = 64, fwd x 6 + bwd + Adam), then one
`rocprofv3 --kernel-trace --stats` run per variant for the kernel split.

| variant | features | K | message | ms / step | graph-edges/s | peak HBM GB |
|---|---|---|---|---|---|---|
''')
        for k, r in v.items():
            f.write(f"| {k} | {r['features']} | {r['K']} | {r['msg']} | {r['ms_per_step']} | {r['edges_per_s']/1e6:.0f} M | {r['mem_GB']} |\n")
        for k in ('att_k2', 'concat', 'g3'):
            p = f'{O}/kernel_stats_{k}.csv'
            if not os.path.exists(p):
                continue
            rows = list(csv.DictReader(open(p)))
            tot = sum(float(r['TotalDurationNs']) for r in rows)
            f.write(f'\n## {k}: top kernels (3 steps: 1 warm-up... as run by --steps 2 + 2 warm-up)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n')
            for r in rows[:12]:
                f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                        f"{100*float(r['TotalDurationNs'])/tot:.1f} |\n")
        if os.path.exists(f'{O}/ragged.log'):
            try:
                rg = json.loads(json_line(f'{O}/ragged.log', '{"workload"'))
                f.write(f"\n## Ragged batch (VERDICT r02 item 9)\n\n`python tools/ragged_bench.py`: {rg['workload']} (no window repeated, so no tile "
                        f"or cache reuse across copies): {rg['ms_per_step']} ms/step, {rg['edges_per_s']/1e6:.0f} M graph-edges/s = "
                        f"{rg['ms_per_M_edge_iterations']} ms per M edge-iterations, against the bench batch (64 distinct windows tiled 256 x) "
                        f"on the same box in the un-profiled bench line.\n")
            except Exception as e:
                print('ragged: ', e)
        for nm in ('c3', 'c4'):
            p = f'{O}/{nm}.json'
            if os.path.exists(p) and os.path.getsize(p) > 10:
                c = json.loads(json_line(p, '{"workload"'))
                f.write(f"\n## {nm.upper()}-shaped step\n\n{c['workload']}: {c['rows_final']:,} rows, {c['edges_final']:,} edges in the last call, "
                        f"{c['edge_iterations_per_step']:,} edge-iterations per step: **{c['ms_per_step']:.1f} ms/step, "
                        f"{c['graph_edges_per_s']/1e6:.0f} M graph-edges/s**, peak HBM {c['mem_GB']:.1f} GB.\n\n| stage kernel | ms | GB/s | of 8 TB/s |\n|---|---|---|---|\n")
                for k, s in c['stages'].items():
                    f.write(f"| {k} | {s['ms']} | {s['GBs']} | {s['hbm_frac']} |\n")
