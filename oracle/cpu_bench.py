#!/usr/bin/env python3
"""CPU legs of bench.py's `cpu_baseline` (TEST / MEASUREMENT INFRASTRUCTURE ONLY, like the rest of oracle/).

The oracle (torch-CPU fp32 restatement of the reference's hot path) timed on the C2 workload of bench.py in three
forms, so that the strongest CPU form is the stated baseline:
  single   one window at a time on ONE core (batch 1, as the reference runs: utils/graph.py:117)
  procs    P independent batch-1 worker processes, one core each (this file run as a script is the worker)
  batched  B windows batched block-diagonally in one process with all threads (index-based oracle, per-window BN)

    python oracle/cpu_bench.py --worker --seconds 5 --seed 1      # prints "EDGES <n> SECONDS <t>"
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _params(F, H, seed=0):
    from oracle import trackmpnn_oracle as orc
    cfg = orc.OracleConfig('2d', F - 5, H, 0, 'diff')
    p = orc.random_params(cfg, seed=seed, scale=0.05)
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
            v.requires_grad_(True)
    return cfg, p


def build_cases(frames, mean_dets, max_dets, F, seed, nwin=64, batch=1):
    """`nwin` windows of the bench generator, grouped `batch` at a time into block-diagonal batches."""
    from oracle import trackmpnn_oracle as orc
    from trackmpnn_amd.graph import WindowBuilder, batch_windows, synth_window
    cases = []
    wins = [WindowBuilder(synth_window(seed * 1000 + s, frames, mean_dets, max_dets)).calls() for s in range(nwin)]
    for b0 in range(0, nwin, batch):
        group = wins[b0:b0 + batch]
        plans, refs = batch_windows(group)
        gen = torch.Generator().manual_seed(b0)
        graphs, xs, segs = [], [], []
        for plan, ref in zip(plans, refs):
            g = plan.graph
            graphs.append(orc.OracleGraph(g.N, g.is_edge.numpy().astype(bool), g.src.numpy().astype(np.int64),
                                          g.dst.numpy().astype(np.int64), g.edge_row.numpy().astype(np.int64),
                                          g.det_row.numpy().astype(np.int64)))
            x = torch.zeros(plan.n_new, F)
            x[plan.new_det_local] = torch.randn(len(ref), F, generator=gen)
            xs.append(x)
            segs.append(plan.seg_of_new)
        cases.append((graphs, xs, segs))
    return cases


def run_case(cfg, p, case):
    from oracle import trackmpnn_oracle as orc
    import torch.nn.functional as Fnn
    graphs, xs, segs = case
    h, loss = None, 0.0
    for g, x, sg in zip(graphs, xs, segs):
        s, l, h, _ = orc.forward(p, cfg, x, h, g, training=True, seg_ids=sg)
        loss = loss + Fnn.binary_cross_entropy_with_logits(l, torch.zeros_like(l), reduction='sum')
    for v in p.values():
        v.grad = None
    loss.backward()
    return sum(g.E for g in graphs)


def timed(cfg, p, cases, seconds):
    run_case(cfg, p, cases[0])
    t0 = time.perf_counter()
    edges = n = 0
    while True:
        for case in cases:
            edges += run_case(cfg, p, case)
            n += 1
            if time.perf_counter() - t0 > seconds:
                return edges, time.perf_counter() - t0, n


def measure(frames, mean_dets, max_dets, F, H, seed, budget_s=20.0, procs=None):
    """The three CPU forms; returns a dict with the best one on top."""
    ncpu = os.cpu_count() or 1
    threads = min(16, ncpu)                      # the GPU box's CPU share for one GPU
    # worker PROCESSES: importing torch opens the GPU device node, and a GPU box admits at most 6 processes with the
    # card open (bench.py itself is one of them) -- so 4 workers; the batched form below uses all threads instead
    procs = procs or min(4, ncpu)
    cfg, p = _params(F, H)
    out = {}
    # (1) one core, batch 1
    torch.set_num_threads(1)
    cases1 = build_cases(frames, mean_dets, max_dets, F, seed, nwin=64, batch=1)
    e, t, n = timed(cfg, p, cases1, budget_s / 5)
    out['single'] = dict(value=e / t, cores=1, windows=n, edge_iterations=e, seconds=round(t, 2))
    # (2) P batch-1 processes, one core each (no GPU in the children)
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    secs = budget_s / 4
    cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--seconds', str(secs), '--frames', str(frames),
           '--mean-dets', str(mean_dets), '--max-dets', str(max_dets), '--features', str(F), '--hidden', str(H)]
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd + ['--seed', str(seed + i)], env=env, stdout=subprocess.PIPE, text=True) for i in range(procs)]
    tot_e, max_t, ok = 0, 0.0, 0
    for pr in ps:
        so, _ = pr.communicate(timeout=600)
        for ln in so.splitlines():
            if ln.startswith('EDGES'):
                _, e_, _, t_ = ln.split()
                tot_e += int(e_)
                max_t = max(max_t, float(t_))
                ok += 1
    if ok:
        out['procs'] = dict(value=tot_e / max_t, cores=ok, edge_iterations=tot_e, seconds=round(max_t, 2),
                            wall_incl_startup=round(time.perf_counter() - t0, 1))
    # (3) block-diagonal batches, all threads of the share
    torch.set_num_threads(threads)
    casesB = build_cases(frames, mean_dets, max_dets, F, seed, nwin=256, batch=256)
    e, t, n = timed(cfg, p, casesB, budget_s / 4)
    out['batched'] = dict(value=e / t, cores=threads, windows_per_batch=256, edge_iterations=e, seconds=round(t, 2))
    torch.set_num_threads(1)
    best = max(out, key=lambda k: out[k]['value'])
    return best, out


def _worker(a):
    torch.set_num_threads(1)
    cfg, p = _params(a.features, a.hidden)
    cases = build_cases(a.frames, a.mean_dets, a.max_dets, a.features, a.seed, nwin=32, batch=1)
    e, t, _ = timed(cfg, p, cases, a.seconds)
    print(f'EDGES {e} SECONDS {t:.4f}', flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--worker', action='store_true')
    ap.add_argument('--seconds', type=float, default=5.0)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--frames', type=int, default=7)
    ap.add_argument('--mean-dets', type=float, default=6.0)
    ap.add_argument('--max-dets', type=int, default=20)
    ap.add_argument('--features', type=int, default=8)
    ap.add_argument('--hidden', type=int, default=64)
    a = ap.parse_args()
    if a.worker:
        _worker(a)
    else:
        best, out = measure(a.frames, a.mean_dets, a.max_dets, a.features, a.hidden, a.seed, budget_s=a.seconds * 4)
        print(best, out)
