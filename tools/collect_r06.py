"""Round-6 summaries under profiles/ from the outputs of tools/measure_r06.sh (gpurun_out/r06_*):
  r06_bench_kernel_stats.{csv,md}  r06_bench_line.json  r06_bench_line_unprofiled.json  r06_pmc_traffic_stage_kernels.json (with the
  date, HEAD and kernel-source digest of the pass: bench.py refuses the file once the kernel sources change)  -- the shared
  parts of tools/collect_r03.py, run with the r06 tag --  r06_c2_step_kernels.md  r06_c5_dense_stress.md  r06_loops.md"""
import csv, datetime, glob, json, os, runpy, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
sys.path.insert(0, root)
tag = 'r06'
os.environ['COLLECT_TAG'] = tag
HBM, MFMA_PEAK = 8000.0, 2500.0
if glob.glob(f'gpurun_out/{tag}_stats/**/*_kernel_stats.csv', recursive=True):
    runpy.run_path(os.path.join(root, 'tools', 'collect_r03.py'), run_name='__main__')
    p = f'profiles/{tag}_pmc_traffic_stage_kernels.json'
    if os.path.exists(p):
        import bench
        d = json.load(open(p))
        d['collected'] = datetime.date.today().isoformat()
        d['head'] = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True).stdout.strip()
        d['kernel_source_digest'] = bench.kernel_source_digest()
        d['digest_of'] = 'sha256[:16] of trackmpnn_amd/csrc/{gru_common.h, gru_fwd.hip, gru_bwd.hip, agg.hip, common.h} (bench.kernel_source_digest)'
        json.dump(d, open(p, 'w'), indent=1)
        print('pmc traffic json: provenance added', d['head'], d['kernel_source_digest'])

# ---------------------------------------------------------------------------------------------- the bench steps alone
sd = f'gpurun_out/{tag}_steps'
if os.path.exists(f'{sd}/kernel_stats.csv'):
    rows = list(csv.DictReader(open(f'{sd}/kernel_stats.csv')))
    line = json.loads([l for l in open(f'{sd}/bench.log') if l.startswith('{')][-1])
    steps = 10 + 2 + 2                      # timed + warm-up + the two set-up steps of bench.py
    tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
    with open(f'profiles/{tag}_c2_step_kernels.md', 'w') as f:
        f.write(f'''# Per-kernel GPU time of the C2 bench steps alone ({tag}, 1x MI355X)

`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage-profile --no-latency --no-loops`
(tools/measure_r06.sh bench): {steps} steps in the trace (10 timed + 2 warm-up + 2 set-up), {line['ms_per_step']:.2f} ms per step under the
profiler; GPU kernel time {tot / steps:.2f} ms per step.

| kernel | ms / step | calls / step | avg us | max us | % of GPU time |
|---|---|---|---|---|---|
''')
        for r in rows[:26]:
            ms = float(r['TotalDurationNs']) / 1e6
            f.write(f"| `{r['Name'].split('(')[0].replace('void ', '').replace('tmpnn::', '')[-70:]}` | {ms / steps:.3f} | {int(r['Calls']) / steps:.1f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {100 * ms / tot:.1f} |\n")
    print('c2 steps:', line['ms_per_step'], tot / steps)


# ---------------------------------------------------------------------------------------------- C5
def _jl(path, prefix='{"workload"'):
    return json.loads([l for l in open(path) if l.startswith(prefix)][-1])


c5 = f'gpurun_out/{tag}_c5'
if os.path.exists(f'{c5}/kernel_stats.csv') and os.path.exists(f'{c5}/pp_plain.log'):
    import shutil
    shutil.copy(f'{c5}/kernel_stats.csv', f'profiles/{tag}_c5_kernel_stats.csv')
    pp = _jl(f'{c5}/pp_plain.log')
    pm = json.load(open(f'{c5}/pmc.json')) if os.path.exists(f'{c5}/pmc.json') else {}
    E, N, H, Dn = pp['E'], pp['N'], 256, pp.get('Dn', 15000)
    sp = pp.get('seg_plan') or dict(T=0, I=0)
    P = 8 * sp['T'] + 16 * sp['I']
    alg = {
        'k_wide_gru_fwd_pp': dict(bytes=E * (4 * H + 4 * H + 16 * H + 12) + Dn * 12 * H, flops=2.0 * 6 * E * H * 3 * H),
        'k_wide_gemm_pp256': dict(bytes=E * (12 * H + 4 * H + 4 * H + 4), flops=2.0 * 6 * E * 3 * H * H),
        'k_wide_dw': dict(bytes=E * (16 * H + 4 * H) / 2, flops=2.0 * 6 * E * 3 * H * H / 2),
        'k_wide_gates_bwd4': dict(bytes=E * (4 * H + 16 * H + 4 * H + 16 * H + 4), flops=0),
        'k_segsum_tiles': dict(bytes=E * 4 * H + P * 4 * H + 4 * 128 * sp['T'], flops=0),
        'k_segsum_pipe': dict(bytes=P * 4 * H + Dn * 4 * H + 4 * (P + Dn + 1), flops=0),
        'k_heads_fwd': dict(bytes=N * (4 * H + 8), flops=0),
        'k_heads_bwd': dict(bytes=N * (4 * H + 16), flops=0),
    }
    rows = list(csv.DictReader(open(f'{c5}/kernel_stats.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    with open(f'profiles/{tag}_c5_dense_stress.md', 'w') as f:
        f.write(f"""# C5 (BASELINE.json configs[4]) -- dense stress, {tag}, 1x MI355X

`bash tools/measure_r06.sh c5` (tools/c5_bench.py): static 50-frame window, 300 dets/frame, H = 256, K = 0, diff, 4 MP iterations
(first call h_in=None with all 4.425 M rows new, then 3 empty-x calls), one backward of sum(logits): N = {N:,} rows, E = {E:,} edges.

| | ms / step (4 fwd + bwd) | graph-edges/s | effective TFLOP/s (36 H^2 per edge-iteration) |
|---|---|---|---|
| round 1 (f32-input MFMA, weights streamed from L2) | 622 | 28.4 M | 67.0 |
| round 2 (LDS-tiled bf16x6 GEMMs, det-side W_ih products) | 257 | 68.6 M | 161.9 |
| round 3 (edge tiles + ring kernels) | 192-195 | 91 M | 211 |
| round 4 (single-read segment sum) | 183-189 | 95.5 M | 225 |
| round 5: `k_wide_gru_fwd_pp` + `k_wide_gemm_pp256<2>`, the block's halves in opposite phases | 180.3 | 97.9 M | 230.9 |
| round 6 (same kernels; this box, loss = sum of logits as in the rounds above) | **{pp['ms_per_step']:.1f}** | **{pp['edges_per_s'] / 1e6:.1f} M** | **{pp['tflops']:.1f}** |

## Per kernel (one stream; every duration un-shared)

rocprofv3 --kernel-trace --stats of `tools/c5_bench.py --steps 2` (1 warm-up + 2 steps); FETCH_SIZE / WRITE_SIZE in two further passes.
`achieved` = algorithmic bytes (every array once, fp32) / average duration, against 8 TB/s; for the matrix kernels also the MFMA-pipe
fraction (6 bf16 products per fp32 product x 2 flops / duration against the dense bf16 peak of 2.5 PFLOP/s).  PMC traffic =
2 x FETCH_SIZE + WRITE_SIZE (KB x 1024; the gfx950 correction for 16-byte-per-lane streams).

| kernel | launches / step | avg ms | % of GPU time | algorithmic GB | GB/s | of 8 TB/s | bf16-MFMA TFLOP/s | of 2.5 PF | PMC traffic GB | traffic / algorithmic |
|---|---|---|---|---|---|---|---|---|---|---|
""")
        for r in rows[:12]:
            name = r['Name']
            key = next((k for k in alg if k in name), None)
            calls, avg_ms = int(r['Calls']), float(r['AverageNs']) / 1e6
            pct = 100 * float(r['TotalDurationNs']) / tot
            short = name.split('(')[0].replace('void ', '').replace('tmpnn::', '')[:44]
            if key is None:
                f.write(f"| `{short}` | {calls / 3:.1f} | {avg_ms:.3f} | {pct:.1f} | | | | | | | |\n")
                continue
            a = alg[key]
            gbs = a['bytes'] / 1e9 / (avg_ms / 1e3)
            tf = a['flops'] / 1e12 / (avg_ms / 1e3) if a['flops'] else None
            pk = next((v for k, v in pm.items() if key in k), None)
            tr = (2 * pk['FETCH_SIZE']['mean'] + pk['WRITE_SIZE']['mean']) * 1024 / 1e9 if pk and 'FETCH_SIZE' in pk and 'WRITE_SIZE' in pk else None
            f.write(f"| `{short}` | {calls / 3:.1f} | {avg_ms:.3f} | {pct:.1f} | {a['bytes'] / 1e9:.2f} | {gbs:.0f} | {gbs / HBM:.2f} | "
                    f"{'%.0f' % tf if tf else ''} | {'%.2f' % (tf / MFMA_PEAK) if tf else ''} | {'%.2f' % tr if tr else ''} | "
                    f"{'%.2f' % (tr / (a['bytes'] / 1e9)) if tr else ''} |\n")
        cal = 'gpurun_out/fetch_calib.txt'
        f.write("""
## Is the 2 x FETCH_SIZE correction right for these kernels' access widths?  (round 6: `tools/ubench/fetch_calib.hip`)

Round 5 suspected that the forward's 1.51 x came from the correction overstating its 4-byte-per-lane previous-state reads (the guide
calibrates 16 B per lane only).  Measured on a 2 GiB buffer read exactly once per kernel (8 x the Infinity Cache):

```
""" + (open(cal).read() if os.path.exists(cal) else '(no calibration run in gpurun_out/)\n') + """```

FETCH_SIZE is half the bytes for EVERY width these kernels use -- 16 B and 4 B per lane, contiguous or split in two 128-byte runs,
into registers or by LDS-DMA: the factor 2 applies unchanged and the forward's traffic ratio is real, not a counter artefact.  What it
is made of (per 128-row item: A rows requested once per 128-column block = twice at H = 256, the previous-state rows again in the
epilogue, the staged P rows): DESIGN.md section 4 (wide cells).
""")
    print('c5:', pp['ms_per_step'])

# ---------------------------------------------------------------------------------------------- loops
ld = f'gpurun_out/{tag}_loops'
if os.path.exists(f'{ld}/device.log') and os.path.exists(f'{ld}/host.log'):
    dev = json.loads([l for l in open(f'{ld}/device.log') if l.startswith('{')][-1])
    host = json.loads([l for l in open(f'{ld}/host.log') if l.startswith('{')][-1])
    with open(f'profiles/{tag}_loops.md', 'w') as f:
        f.write(f'''# The reference's loops at batch 1 ({tag}, 1x MI355X; `tools/loop_bench.py` = bench.py's `loop_batch1` block)

Inference loop (`infer.py:35-87`), ms per timestep; Hungarian association (README.md:67,122 recommends `--hungarian`) on the device
since round 5 (`csrc/trackops.hip d_track_hungarian`: scipy's linear_sum_assignment restated inside the select / retire launches,
bit-equal to the host matching incl. ties) against the host matching (`TMPNN_HUNGARIAN_HOST=1`: scipy, two more device <-> host copies
per sweep), same box, same sequences:

| sequence | greedy | Hungarian on the device | Hungarian, model without TP classifier (the README's commands) | Hungarian on the host (round 4's form) |
|---|---|---|---|---|
''')
        for c in ('C2', 'C3', 'C4'):
            f.write(f"| {c} | {dev['infer'][c + '/greedy']['ms_per_timestep']:.3f} | **{dev['infer'][c + '/hungarian']['ms_per_timestep']:.3f}** | "
                    f"{dev['infer'].get(c + '/hungarian_no_tp_classifier', {}).get('ms_per_timestep', float('nan')):.3f} | "
                    f"{host['infer'][c + '/hungarian']['ms_per_timestep']:.3f} |\n")
        f.write('''
Round 5 on the same sequences: greedy 0.097 / 0.110 / 0.121, Hungarian on the device 0.150 / 0.308 / 0.242.  What changed (DESIGN.md
section 7, item 4): the retire launch mirrors its counters into pinned host memory and the host polls a flag there (no copy-back);
block append + index form + the model call's input transform in one launch (`k_track_extend_tf`), enqueued before the previous decode's
counters are read and moving that decode's kept state rows as well; greedy association a DPP row per det; the finalisation walk by
pointer doubling; the counters published before that walk; the fused iteration's det tiles summing their incident rows as
straight-line loads; steady-state timesteps back to back inside the native driver; the second Hungarian sweep of a launch reuses the
first's outcome per unchanged problem; models without TP classifier (README.md:52-67) ride the native timestep.
''')
        f.write('\nTrain chunk (`train.py:54-135`), ms per chunk: ' + ', '.join(f"{c} {dev['train'][c]['ms_per_chunk']:.2f}" for c in ('C2', 'C3', 'C4')) + '\n')
        if os.path.exists(f'{ld}/slowest.txt'):
            f.write('''
## The slowest dispatches of the tracker kernels (`rocprofv3 --kernel-trace` of the same command)

Round 4's review found maxima of 29.2 ms (`k_track_retire`) and 20.9 ms (`k_track_append`) in the kernel table of the full bench
run and asked which call they were.  In a trace of the loops alone (every dispatch of the train chunks and of the greedy / Hungarian
inference loops of the three sequence shapes: ~0.5 M dispatches) there is no such dispatch: per kernel, dispatch count, median / mean
/ maximum, and its three slowest dispatches with their position in the trace and the kernel in front of them:

```
''')
            f.write(open(f'{ld}/slowest.txt').read())
            f.write('''```

The maxima of `k_track_retire` / `k_track_select` are the Hungarian sweeps of the C3 sequences (12-frame windows: ten timesteps'
problems solved one after another inside the launch); nothing waits or spins.

The outliers of the full bench run (`profiles/r06_bench_kernel_stats.md`: MaxNs of `k_track_retire` 20-31 ms): `tools/outliers.py`
over a `rocprofv3 --kernel-trace` of `bench.py` finds ONE or TWO such dispatches among ~770 000 (20.6 ms in one trace, 31.2 ms +
1.1 ms in another), each in the middle of a Hungarian inference loop between neighbours of ordinary length, on the same queue:

```
      + 23928.723 ms .. + 23928.947 ms      223.7 us  k_track_retire
      + 23928.947 ms .. + 23928.951 ms        4.2 us  k_track_gather
      + 23928.964 ms .. + 23928.979 ms       14.4 us  k_track_extend_tf<64>
      + 23928.979 ms .. + 23929.001 ms       22.1 us  k_small_iter_fwd<64, 64>
   >> + 23929.001 ms .. + 23960.169 ms    31168.4 us  k_track_retire
      + 23960.169 ms .. + 23960.176 ms        7.3 us  k_track_gather
      + 23960.176 ms .. + 23960.181 ms        4.1 us  __amd_rocclr_copyBuffer      (the host's poll gave up after 20 ms and copied)
      + 23960.231 ms .. + 23960.251 ms       19.6 us  k_track_extend_tf<64>
```

The kernel has no wait in it (the device never waits for the host; its loops are bounded by the problem sizes), the same inputs
take 0.1-0.3 ms in the launches around it, and a trace of the loops alone has never shown one (round 4's 29.2 ms / 20.9 ms maxima
were found in traces of the full bench run too).  `bench.py` now reports the slowest single sequence of every timed loop
(`loop_batch1.infer.*.slowest_sequence_ms`, host time of one 40-frame sequence) so that the un-profiled run can be asked the same
question: in four runs, two show every slowest sequence within 25 % of its mean (e.g. C2 / C3 / C4 Hungarian 4.79 / 9.40 / 7.71 ms
against 4.66 / 9.33 / 7.64 over 322 / 161 / 197 sequences), and two show one or two sequences 5-10 ms above it (C3 greedy 12.4 ms
and C4 greedy 7.4 ms against means of 2.2, in ~700 sequences each; freezing the interpreter's garbage collector before the loops
does not remove them).  So a stall of 5-30 ms hits the loop about once per 1-2 s of run time, with or without the profiler, not
reproducibly, and lands on whatever is running -- under the profiler the longest kernel of the loop; it comes from outside the
kernels (the host thread or the queue losing its slot on a shared box is what fits; host timing alone cannot tell which).
''')
        ts = f'{ld}/c2_timestep.json'
        if os.path.exists(ts):
            d = json.load(open(ts))
            f.write('''
## Where a C2 timestep goes (`tools/greedy_trace.py`: the inference loop of the C2 sequence alone, 50 sequences back to back)

Wall time per timestep (un-profiled run) against the GPU kernel time per timestep (`rocprofv3 --kernel-trace` of the same command):
three launches per steady-state timestep (`k_track_extend_tf` -- enqueued behind the previous decode before its counters are read, it
also moves that decode's kept state rows --, `k_small_iter_fwd`, `k_track_retire`) whose lengths are the latency of their dependent
memory round trips (every kernel starts on a cold L2); the rest is dispatch gaps and the host's part (polling the counters' flag,
sizing and issuing the iteration and the decode).  The Hungarian loop is bound by the retire launch's two
assignment sweeps (the solver itself, one wave, ~6 us per problem, a timestep's problems one after another: the rows of problem t
are the dets still unassociated after the problems before it).

''')
            for m in ('greedy', 'hungarian'):
                x = d[m]
                f.write(f"**{m}**: wall {x['wall_ms_per_timestep'] * 1e3:.0f} us per timestep, GPU kernels {x['kernel_us_per_timestep']:.1f} us per timestep "
                        f"({x['profiled_ms_per_timestep'] * 1e3:.0f} us per timestep under the profiler)\n\n| kernel | launches / timestep | us / timestep | avg us |\n|---|---|---|---|\n")
                for k, v in x['kernels'].items():
                    f.write(f"| `{k}` | {v['per_timestep']} | {v['us_per_timestep']} | {v['avg_us']} |\n")
                f.write('\n')
    print('loops: written')
