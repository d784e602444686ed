#!/usr/bin/env python3
"""Time the REAL reference's two per-timestep loops (build container only: needs /root/reference) on the synthetic
sequences bench.py's `loop_batch1` block uses, so that the GPU numbers have the reference's own time next to them.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_loops.py  > /tmp/ref_loops.json

  train chunk  = train.py:54-135 without logging: initialize_graph, then per timestep update_graph(mode='train') ->
                 model -> create_targets + CELoss + FocalLoss; one backward + Adam step per chunk
  inference    = infer.py:35-87: update_graph(mode='test', greedy | Hungarian) -> model -> decode_tracks

The drivers themselves cannot be imported (argparse at import); the loops below call the reference's own
functions in the drivers' order.  CPU only (the container has no GPU): threads 1 and 8."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')

from trackmpnn_amd.graph import synth_window                   # noqa: E402  (host-only generator, no GPU code)

SHAPES = {'C2': dict(frames=7, mean=6.0, mx=20, ncat=3, win=5), 'C3': dict(frames=12, mean=8.0, mx=25, ncat=3, win=10),
          'C4': dict(frames=7, mean=12.0, mx=40, ncat=8, win=5)}
INFER_FRAMES = 40


def sequence(seed, frames, mean, mx, ncat):
    yy = synth_window(seed, frames, mean, mx)
    y = torch.from_numpy(yy)[None]
    X = torch.randn(1, yy.shape[0], ncat + 5, generator=torch.Generator().manual_seed(seed + 1000))
    return X, y


def perturb(model, seed):
    """scores on both sides of 0.5 (as the inference fixtures of gen_golden.py); the same code runs on the GPU side"""
    gp = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, prm in model.named_parameters():
            prm.add_(0.1 * torch.randn(prm.shape, generator=gp))
            if k.startswith('output_transform') and k.endswith('bias'):
                prm.copy_(0.5 * torch.randn(prm.shape, generator=gp))


def train_chunk(model, opt, X, y, L):
    from utils.graph import initialize_graph, update_graph
    create_targets, ce, fn, fe = L
    opt.zero_grad()
    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    scores, logits, states, _ = model(feats, None, node_adj, edge_adj)

    def terms(scores, logits, labels, node_adj, y_pred):
        idx_edge = torch.nonzero((y_pred[:, 0] == -1))[:, 0]
        idx_node = torch.nonzero((y_pred[:, 0] != -1))[:, 0]
        targets = create_targets(labels, node_adj, idx_node)
        return (ce(logits, targets, node_adj, idx_node),
                fn(scores[idx_node, 0], targets[idx_node]) + fe(scores[idx_edge, 0], targets[idx_edge]))

    loss_c, loss_f = terms(scores, logits, labels, node_adj, y_pred)
    edges = int((y_pred[:, 0] == -1).sum())
    for t_cur in range(t_st, t_end):
        s2 = torch.cat((1 - scores, scores), dim=1)
        y_pred, feats, node_adj, edge_adj, labels = update_graph(node_adj, labels, s2, y_pred, X, y, t_cur,
                                                                 use_hungraian=False, mode='train', cuda=False)
        scores, logits, states, _ = model(feats, states, node_adj, edge_adj)
        lc, lf = terms(scores, logits, labels, node_adj, y_pred)
        loss_c, loss_f = loss_c + lc, loss_f + lf
        edges += int((y_pred[:, 0] == -1).sum())
    (loss_c + loss_f).backward()
    opt.step()
    return edges


def infer_sequence(model, X, y, cur_win, ret_win, hungarian):
    from utils.graph import decode_tracks, initialize_graph, update_graph
    y_out = y.squeeze(0).numpy().astype('int64').copy()
    y_out[:, 1] = -1
    edges = 0
    with torch.no_grad():
        y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, t_st=0, mode='test', cuda=False)
        scores, logits, states, _ = model(feats, None, node_adj, edge_adj)
        edges += int((y_pred[:, 0] == -1).sum())
        scores = torch.cat((1 - scores, scores), dim=1)
        t_skip = t_st
        for t_cur in range(t_st, t_end):
            if t_cur < t_skip:
                continue
            if feats.size()[0] == 0 and states.size()[0] == 0:
                y_pred, feats, node_adj, edge_adj, labels, t_skip, _ = initialize_graph(X, y, t_st=t_cur, mode='test', cuda=False)
                if y_pred is None:
                    break
                states = None
            else:
                y_pred, feats, node_adj, edge_adj, labels = update_graph(
                    node_adj, labels, scores, y_pred, X, y, t_cur, use_hungraian=hungarian, mode='test', cuda=False)
            scores, logits, states, _ = model(feats, states, node_adj, edge_adj)
            edges += int((y_pred[:, 0] == -1).sum())
            scores = torch.cat((1 - scores, scores), dim=1)
            t_upto = t_end if t_cur == t_end - 1 else t_cur - cur_win + 2
            y_pred, y_out, states, node_adj, labels, scores = decode_tracks(
                states, node_adj, labels, scores, y_pred, y_out, t_upto, ret_win, use_hungraian=hungarian, cuda=False)
    return y_out, edges


def timed(fn, budget=6.0):
    fn()
    ts = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget or len(ts) < 3:
        a = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - a)
    return float(np.median(ts)), r


def main():
    from models.loss import CELoss, FocalLoss, create_targets
    from models.track_mpnn import TrackMPNN
    out = dict(host=f'{os.cpu_count()} vCPU (build container)', torch=torch.__version__, train={}, infer={})
    for tag, s in SHAPES.items():
        torch.manual_seed(5)
        model = TrackMPNN('2d', s['ncat'], 64, 0, 'diff')
        perturb(model, 4242)
        L = (create_targets, CELoss(), FocalLoss(gamma=0), FocalLoss(gamma=0))
        X, y = sequence(1001, s['frames'], s['mean'], s['mx'], s['ncat'])
        Xi, yi = sequence(2001, INFER_FRAMES, s['mean'], s['mx'], s['ncat'])
        for nt in (1, 8):
            torch.set_num_threads(nt)
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)
            dt, edges = timed(lambda: train_chunk(model, opt, X, y, L))
            out['train'][f'{tag}/threads{nt}'] = dict(ms_per_chunk=dt * 1e3, edge_iterations=edges, edges_per_s=edges / dt)
            model.eval()
            for hung in (False, True):
                dt, (y_out, edges) = timed(lambda: infer_sequence(model, Xi, yi, s['win'], 0, hung))
                out['infer'][f"{tag}/{'hungarian' if hung else 'greedy'}/threads{nt}"] = dict(
                    ms_per_sequence=dt * 1e3, frames=INFER_FRAMES, ms_per_timestep=dt * 1e3 / INFER_FRAMES, edge_iterations=edges,
                    tracks=int(y_out[:, 1].max()) + 1)
            print(json.dumps({k: v for k, v in out.items() if k in ('train', 'infer')})[:400], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
