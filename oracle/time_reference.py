#!/usr/bin/env python3
"""Time the REAL reference (build container only: needs /root/reference) against the oracle restatement on the
C2 workload of bench.py, one window at a time, model forward + backward only (graphs prebuilt).

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference.py

Prints graph-edges/s for both so the `cpu_baseline` of bench.py (the oracle, kind "port", timed on the GPU
box's host) can be translated into reference-equivalent numbers (SURVEY 8(d) "CPU reference timing").
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')

from oracle import trackmpnn_oracle as orc                    # noqa: E402
from trackmpnn_amd.graph import synth_window                   # noqa: E402  (host-only generator, no GPU code)


def ref_window(model, X, y):
    from utils.graph import initialize_graph, update_graph
    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    calls = [(feats, node_adj, edge_adj)]
    for t in range(t_st, t_end):
        y_pred, feats, node_adj, edge_adj, labels = update_graph(
            node_adj, labels, torch.zeros(node_adj.shape[0], 1), y_pred, X, y, t, mode='train', cuda=False)
        calls.append((feats, node_adj, edge_adj))
    return calls


def main():
    from models.track_mpnn import TrackMPNN
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').train()
    cfg = orc.OracleConfig('2d', 3, 64, 0, 'diff')
    p = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
            v.requires_grad_(True)
    shape = sys.argv[1] if len(sys.argv) > 1 else 'C2'
    frames, mean, mx, ncat, nwin = {'C2': (7, 6.0, 20, 3, 16), 'C3': (12, 8.0, 25, 3, 6), 'C4': (7, 12.0, 40, 8, 6)}[shape]
    if ncat != 3:
        torch.manual_seed(5)
        model = TrackMPNN('2d', ncat, 64, 0, 'diff').train()
        cfg = orc.OracleConfig('2d', ncat, 64, 0, 'diff')
        p = {k: v.clone() for k, v in model.state_dict().items()}
        for k, v in p.items():
            if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
                v.requires_grad_(True)
    wins = []
    for s in range(nwin):
        yy = synth_window(1000 + s, frames, mean, mx)
        y = torch.from_numpy(yy)[None]
        X = torch.randn(1, yy.shape[0], ncat + 5, generator=torch.Generator().manual_seed(s))
        calls = ref_window(model, X, y)
        graphs = [orc.graph_from_adjacency(c[1], c[2]) for c in calls]
        wins.append((calls, graphs))
    edges = sum(g.E for _, gs in wins for g in gs)

    def run_ref():
        for calls, _ in wins:
            h = None
            loss = 0.0
            for x, na, ea in calls:
                s, l, h, _ = model(x, h, na, ea)
                loss = loss + l.sum()
            model.zero_grad()
            loss.backward()

    def run_orc():
        for calls, graphs in wins:
            h = None
            loss = 0.0
            for (x, _, _), g in zip(calls, graphs):
                s, l, h, _ = orc.forward(p, cfg, x, h, g, training=True)
                loss = loss + l.sum()
            for v in p.values():
                v.grad = None
            loss.backward()

    print(f'{shape} windows: {nwin}, edge-iterations per pass: {edges}; host: {os.cpu_count()} cores, torch {torch.__version__}')
    for nt in (1, 8):
        torch.set_num_threads(nt)
        for name, fn in (('reference', run_ref), ('oracle', run_orc)):
            fn()
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 8.0:
                fn()
                reps += 1
            dt = (time.perf_counter() - t0) / reps
            print(f'{name:9s} threads={nt}: {edges / dt:10.0f} graph-edges/s  ({dt * 1e3 / nwin:.2f} ms / window)')


if __name__ == '__main__':
    main()
