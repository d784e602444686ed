"""Turn the outputs of tools/measure_round.sh (gpurun_out/<tag>_*) into the tracked summaries under profiles/:
<tag>_bench_kernel_stats.{csv,md}, <tag>_bench_line.json, <tag>_bench_line_unprofiled.json,
<tag>_pmc_traffic_stage_kernels.json, <tag>_c3_kitti_all_shape.md, <tag>_c5_dense_stress.md."""
import csv, glob, json, collections, shutil, os, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def json_line(path, start='{"metric"'):
    return [l for l in open(path, errors='ignore') if l.startswith(start)][0]


tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)

# ---- bench.py under rocprofv3 --kernel-trace --stats
src = newest(f'gpurun_out/{tag}_stats/*/*_kernel_stats.csv')
shutil.copy(src, f'profiles/{tag}_bench_kernel_stats.csv')
rows = list(csv.DictReader(open(src)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
line = json_line(f'gpurun_out/{tag}_bench.log')
open(f'profiles/{tag}_bench_line.json', 'w').write(line)
d = json.loads(line)
plain = json_line(f'gpurun_out/{tag}_bench_plain.json')
open(f'profiles/{tag}_bench_line_unprofiled.json', 'w').write(plain)
du = json.loads(plain)
stage_kernel = {'gru_fwd_edge': 'k_gru_fwd_split<64, 8>', 'gru_bwd_one_edge': 'k_gru_bwd_two<1, 3, true>',
                'gru_bwd_data_edge_folded': 'k_gru_bwd_data_split<64, 3, true>',
                'gru_bwd_data_edge': 'k_gru_bwd_data_split<64, 1, false>',
                'gru_bwd_weights_edge': 'k_gru_bwd_weights_split<1, 1>',
                'gather_diff': 'k_gather_pipe<false, false>', 'segsum': 'k_segsum_pipe<false>'}
with open(f'profiles/{tag}_bench_kernel_stats.md', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2   ({tag}, 1x MI355X)\n\n')
    f.write('What the run contains: 12 steps of the C2 workload (16 384 windows; six forward calls on graphs of six sizes + '
            'one backward per step), the stage profile on the LAST call\'s graph '
            f"(E = {d['stage_graph']['E']}, 6 timed launches per stage kernel), the batch-1 latency block (`k_small_*`, "
            '`k_graph_from_coo`: thousands of launches on graphs of a few hundred rows) and the CPU baseline.  One row per '
            'kernel therefore averages several graph sizes: the duration a roofline fraction is computed from is the '
            'HIP-event time of the stage profile, which the largest dispatch of that kernel in the trace (MaxNs) confirms:\n\n')
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
    for r in rows[:36]:
        f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['MaxNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
    f.write(f'\ntotal GPU kernel time {tot/1e6:.1f} ms\n')


# ---- FETCH_SIZE / WRITE_SIZE passes over the stage kernels
def pmc(path, name):
    rows = list(csv.DictReader(open(newest(path))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == name:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch = pmc(f'gpurun_out/{tag}_fetch/*/*_counter_collection.csv', 'FETCH_SIZE')
write = pmc(f'gpurun_out/{tag}_write/*/*_counter_collection.csv', 'WRITE_SIZE')
stages = json.loads(json_line(f'gpurun_out/{tag}_write.log', '{"E"'))
out = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes, --kernel-trace only) -- python3 tools/stage_bench.py --windows 16384',
       'graph': {k: stages[k] for k in ('E', 'Dn', 'N')},
       'note': 'per-launch averages; KB as reported by rocprofv3 (x1024 = bytes), RAW. gfx950 tallies the 128-B requests of wide '
               'coalesced reads (16 B per lane) at 64 B (MI355X_MICROARCH.md, HBM section): bench.py prices reads as 2 x FETCH_SIZE '
               'and writes as WRITE_SIZE.',
       'kernels': {k[:90]: dict(FETCH_SIZE_KB=fetch[k], WRITE_SIZE_KB=write.get(k)) for k in fetch if 'tmpnn' in k}}
json.dump(out, open(f'profiles/{tag}_pmc_traffic_stage_kernels.json', 'w'), indent=1)

# ---- C3-shaped workload
c3 = json.loads(open(f'gpurun_out/{tag}_c3.json').read())
r1 = {'gru_fwd_edge': 0.40, 'gru_bwd_data_edge': 0.54, 'gru_bwd_data_edge_folded': 0.51, 'gru_bwd_weights_edge': 0.31,
      'gru_bwd_weights_edge_f32mfma': 0.40, 'gather_diff': 0.39, 'segsum': 0.34}
with open(f'profiles/{tag}_c3_kitti_all_shape.md', 'w') as f:
    f.write(f'''# C3 (BASELINE.json configs[2]) -- KITTI All / CenterTrack-shaped windows, cur-win-size 10, {tag}, 1x MI355X

`python tools/c3_profile.py` -- 2048 windows x 12 frames, D_t ~ clip(Poisson(10), 1, 30), F = 8, H = 64, K = 0, diff;
rolling forward (11 calls, state carried) + one backward + Adam, batched block-diagonally: {c3['rows_final']:,} rows,
{c3['edges_final']:,} edges in the last call, {c3['edge_iterations_per_step']:,} edge-iterations per step.

| | ms / step | graph-edges/s |
|---|---|---|
| round 1 | 54.5 | 428 M |
| round 2 | {c3['ms_per_step']:.1f} | {c3['graph_edges_per_s']/1e6:.0f} M |

Stage kernels on the last call's graph (HIP events, algorithmic bytes as in DESIGN.md section 4, roof = 8 TB/s).  The default
step runs `gru_fwd_edge`, `gru_bwd_one_edge` (one pass for the data and weight gradients) and `segsum` (forward and adjoint);
the two stand-alone backward kernels and `gather_diff` are measured for comparison:

| kernel | ms | GB/s | of 8 TB/s | round 1 |
|---|---|---|---|---|
''')
    for k, v in c3['stages'].items():
        f.write(f"| {k} | {v['ms']} | {v['GBs']} | {v['hbm_frac']} | {r1.get(k, '')} |\n")
    f.write('''
Dets here have ~34 incident edges (C2: ~16): the segment sum takes three passes per det; round 2's kernel pipelines the
index loads of every pass (and of the next det) behind the row loads of the current one, keeps eight rows in flight per
lane group on such high-degree graphs (two passes) and walks 16-entry chunks of the visiting order per block.  History of
this shape on the same binary family: 0.34 (round 1) -> 0.39-0.43 (pipelined) -> 0.47-0.49 (rows in flight, chunking, finer
grids); the boxes differ by up to 10 % on these latency-sensitive row movers.
''')

# ---- C5
c5 = json.loads(json_line(f'gpurun_out/{tag}_c5.json', '{"workload"'))
with open(f'profiles/{tag}_c5_dense_stress.md', 'w') as f:
    f.write(f'''# C5 (BASELINE.json configs[4]) -- dense stress, {tag}, 1x MI355X

`python tools/c5_bench.py --steps 3` -- static 50-frame window, 300 dets/frame, H = 256, K = 0, diff, 4 MP iterations
(first call h_in=None with all 4.425 M rows new, then 3 empty-x calls), one backward of sum(logits).

| | ms / step (4 fwd + bwd) | graph-edges/s | effective TFLOP/s (36 H^2 per edge-iteration) | peak HBM |
|---|---|---|---|---|
| round 1 (f32-input MFMA, weights streamed from L2) | 622 | 28.4 M | 67.0 | 97.6 GB |
| round 2 (edge cell: LDS-tiled bf16x6 GEMMs, csrc/wide.hip) | **{c5['ms_per_step']:.0f}** | **{c5['edges_per_s']/1e6:.1f} M** | **{c5['tflops']:.1f}** | {c5['mem_GB']:.1f} GB |

The step through round 2: 622 (round 1) -> 409 (forward + backward-data on LDS-tiled bf16x6 GEMMs) -> 349-366 (weight gradient
from the materialised gate gradients, `k_wide_dw`) -> **265** (both W_ih products of the backward on the DET side,
`tmpnn_wide_gru_bwd_diff`: by linearity of the diff message x = h[src] - h[dst], sum_e d_gi[e]^T x[e] = sum_d S[d]^T h[d] and the
message adjoint is S W_ih with S[d] = signed segment sum of d_gi over the det's incident edges -- two of the four
(4.41 M x 768 x 256) products run over 15 000 rows instead) -> ~255 with the per-segment BatchNorm reductions on row groups
(a one-window batch is ONE segment of 15 000 dets: 7.4 + 4.1 ms per step on one thread per column before).

Per iteration (`rocprofv3 --kernel-trace --stats -- python3 tools/c5_bench.py`, gpurun_out/r02_c5prof): edge forward 19.1 ms
(round 1: 54), gate gradients 8.5 (one [N][4H] image = [dr | dz | dn | dn r]; 9.7 as two [E][3H] images), d_gh W_hh 12.7,
weight gradient dW_hh 11.9 (`k_wide_dw`), signed segment sum of d_gi 3 x 1.5 = 4.6, the two det-side products < 0.5, the
other aggregations 3.3.  The forward is held by its memory-system traffic (the P gathers of the diff projection, 6 KB per
row from L2 / Infinity Cache, plus four gate planes written), the W_hh products by reading the materialised gate gradients
(3 KB per row per product).  Tried without effect: one- and two-deep register prefetch of the next K-step, XCD-aware tile
order.  What helped: stores and gathers as 16-byte accesses through an LDS-staged epilogue (gemm 17 -> 12.7 ms).  The
reference cannot run this configuration at all (dense N x N adjacency: 4.4 M^2 floats).
''')

# ---- C4 (optional: written when tools/measure_round.sh produced it)
if os.path.exists(f'gpurun_out/{tag}_c4.json'):
    c4 = json.loads(json_line(f'gpurun_out/{tag}_c4.json', '{"workload"'))
    with open(f'profiles/{tag}_c4_bdd_shape.md', 'w') as f:
        f.write(f"""# C4 (BASELINE.json configs[3]) -- BDD100K All / libra-shaped windows, per-GPU step, {tag}, 1x MI355X

`python tools/c4_profile.py` -- 4096 windows x 7 frames, D_t ~ clip(Poisson(12), 1, 40), 8 categories => F = 13, H = 64, K = 0,
diff; rolling forward (6 calls, state carried) + one backward + Adam, batched block-diagonally: {c4['rows_final']:,} rows,
{c4['edges_final']:,} edges in the last call, {c4['edge_iterations_per_step']:,} edge-iterations per step.  BASELINE's C4 is this step on
each of 8 ranks plus ONE flat-bucket all-reduce of 55 234 floats per step (bench.py --gpus 8; no 8-GPU node was available to the
builder: DESIGN.md section 6).

| | ms / step | graph-edges/s | peak HBM |
|---|---|---|---|
| round 2 | {c4['ms_per_step']:.1f} | {c4['graph_edges_per_s']/1e6:.0f} M | {c4['mem_GB']:.1f} GB |

Stage kernels on the last call's graph (HIP events, algorithmic bytes as in DESIGN.md section 4, roof = 8 TB/s); dets here have
~29 incident edges (C2: ~16):

| kernel | ms | GB/s | of 8 TB/s |
|---|---|---|---|
""")
        for k, v in c4['stages'].items():
            f.write(f"| {k} | {v['ms']} | {v['GBs']} | {v['hbm_frac']} |\n")
print(d['value'], d['ms_per_step'], d['roofline'], d['roofline_aggregation']['frac'], d['cpu_baseline']['value'])
