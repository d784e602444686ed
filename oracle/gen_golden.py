#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference; CPU).  The reference never travels:
what is committed is data only -- inputs (X, y, per-call x and adjacency exactly as the
reference's own utils/graph.py produced them, parameters) and outputs (scores, logits, h_out,
attention at the incident positions, parameter / input gradients, BatchNorm buffers).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--reference-path /root/reference]

Each fixture is one rolling sequence that reproduces the reference call pattern
(train.py:65-68,92-107,132-135): initialize_graph -> forward(h_in=None, dense adjacency),
then per timestep update_graph(mode='train') -> forward(sparse adjacency), then one extra
forward with an empty x, then ONE backward of a seeded linear functional of all outputs.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.trackmpnn_oracle import graph_from_adjacency  # noqa: E402


def synth_sequence(seed, T, dmean, ncat, feats, fp_rate=0.15, miss=0.2):
    """Seeded synthetic detections: X [1, ND, F], y [1, ND, 2] = [ts, track_id] (-1 = false positive)."""
    rng = np.random.RandomState(seed)
    ntracks = max(2, int(dmean) + 1)
    rows = []
    for t in range(T):
        ids = [i for i in range(ntracks) if rng.rand() > miss]
        if len(ids) == 0:
            ids = [int(rng.randint(ntracks))]
        nfp = int(rng.rand() < fp_rate * 2) + int(rng.rand() < fp_rate)
        ids = ids + [-1] * nfp
        rng.shuffle(ids)
        rows += [(t, i) for i in ids]
    y = torch.tensor(rows, dtype=torch.int64)[None]
    F = 0
    if '2d' in feats:
        F += ncat + 5
    if 'temp' in feats:
        F += 2
    if 'vis' in feats:
        F += 128
    g = torch.Generator().manual_seed(seed + 1000)
    X = torch.randn(1, y.shape[1], F, generator=g)
    return X, y


def adj_arrays(adj):
    if adj.is_sparse:
        return adj._indices().numpy().copy(), adj._values().numpy().copy(), 0
    nz = torch.nonzero(adj)
    return nz.t().numpy().copy(), adj[nz[:, 0], nz[:, 1]].numpy().copy(), 1


def run_fixture(name, out_dir, features, ncat, H, K, msg, mode, seed, T=4, dmean=3, static_iters=0, pscale=0.3,
                sequence=None, extra_iter=True, h_every=1):
    from models.track_mpnn import TrackMPNN
    from utils.graph import initialize_graph, update_graph

    torch.manual_seed(5)
    model = TrackMPNN(features, ncat, H, K, msg)
    gp = torch.Generator().manual_seed(seed + 77)
    with torch.no_grad():
        for _, prm in model.named_parameters():
            prm.add_(pscale * torch.randn(prm.shape, generator=gp))
        for k, b in model.named_buffers():
            if k.endswith('running_mean'):
                b.copy_(0.2 * torch.randn(b.shape, generator=gp))
            elif k.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=gp))
    model.train() if mode == 'train' else model.eval()
    out = {}
    for k, v in model.state_dict().items():
        out['param/' + k] = v.detach().numpy().copy()

    X, y = sequence if sequence is not None else synth_sequence(seed, T, dmean, ncat, features)
    X.requires_grad_(True)
    out['X'] = X.detach().numpy().copy()
    out['y'] = y.numpy().copy()

    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    calls = [(feats, node_adj, edge_adj)]
    # graphs depend only on GT in train mode, so they can be built before any forward
    scores_dummy = torch.zeros(node_adj.shape[0], 1)
    for t in range(t_st, t_end):
        y_pred, feats, node_adj, edge_adj, labels = update_graph(
            node_adj, labels, scores_dummy, y_pred, X, y, t, mode='train', cuda=False)
        scores_dummy = torch.zeros(node_adj.shape[0], 1)
        calls.append((feats, node_adj, edge_adj))
    if static_iters > 0:
        # static mode: only the final graph; x = every row, then empty-x iterations
        allx = torch.cat([c[0] for c in calls], 0)
        calls = [(allx, node_adj, edge_adj)] + [(allx[:0], node_adj, edge_adj)] * (static_iters - 1)
    elif extra_iter:
        calls.append((feats[:0], node_adj, edge_adj))   # extra MP iteration with empty x

    gw = torch.Generator().manual_seed(seed + 99)
    loss = 0.0
    h = None
    G = len(model.feature_idx)
    for c, (x, na, ea) in enumerate(calls):
        scores, logits, h, att = model(x, h, na, ea)
        N = logits.shape[0]
        graph = graph_from_adjacency(na, ea)
        wl = torch.randn(N, 1, generator=gw)
        ws = torch.randn(N, 1, generator=gw)
        loss = loss + (wl * logits).sum() + (ws * scores).sum()
        pre = f'c{c}/'
        out[pre + 'x'] = x.detach().numpy().copy()
        for nm, a in (('node_adj', na), ('edge_adj', ea)):
            idx, val, dense = adj_arrays(a.detach())
            out[pre + nm + '_idx'] = idx.astype(np.int64)
            out[pre + nm + '_val'] = val.astype(np.float32)
            out[pre + nm + '_dense'] = np.int64(dense)
        out[pre + 'N'] = np.int64(N)
        out[pre + 'wl'] = wl.numpy().copy()
        out[pre + 'ws'] = ws.numpy().copy()
        if (static_iters == 0 and c % h_every == 0) or c == len(calls) - 1:
            out[pre + 'h_out'] = h.detach().numpy().copy()
        out[pre + 'scores'] = scores.detach().numpy().copy()
        out[pre + 'logits'] = logits.detach().numpy().copy()
        if K > 0:
            er = torch.from_numpy(graph.edge_row)
            s_, d_ = torch.from_numpy(graph.src), torch.from_numpy(graph.dst)
            for g in range(G):
                keeps = []
                for k in range(K):
                    a = att[g][k].detach()
                    vals = torch.stack([a[s_, er], a[d_, er]], 1)
                    out[pre + f'att_g{g}_k{k}'] = vals.numpy().copy()
                    keeps.append((vals != 0).to(torch.uint8))
                    if mode != 'train':
                        # edge rows are the uniform 1/N of an all-masked softmax (layers.py:35-36)
                        assert torch.allclose(a[er], torch.full_like(a[er], 1.0 / N))
                        # nothing outside the incident positions on det rows
                        dense = torch.zeros_like(a)
                        dense[s_, er] = vals[:, 0]
                        dense[d_, er] = vals[:, 1]
                        dr = torch.from_numpy(graph.det_row)
                        has_inc = torch.zeros(N, dtype=torch.bool)
                        has_inc[s_] = True
                        has_inc[d_] = True
                        rows = dr[has_inc[dr]]
                        assert torch.allclose(a[rows], dense[rows], atol=1e-7)
                if mode == 'train':
                    out[pre + f'keep_g{g}'] = torch.stack(keeps, 0).numpy().copy()
    V = torch.randn(h.shape, generator=gw)
    loss = loss + (V * h).sum()
    out['V'] = V.numpy().copy()
    loss.backward()
    for k, prm in model.named_parameters():
        out['grad/' + k] = prm.grad.detach().numpy().copy()
    out['grad/X'] = X.grad.detach().numpy().copy()
    out['loss'] = np.float64(loss.item())
    for k, b in model.named_buffers():
        out['final/' + k] = b.detach().numpy().copy()
    meta = dict(name=name, features=features, ncategories=ncat, nhidden=H, nattheads=K, msg_type=msg,
                mode=mode, ncalls=len(calls), seed=seed, T=T, static_iters=static_iters, pscale=pscale,
                torch=torch.__version__, reference='arangesh/TrackMPNN @ /root/reference')
    out['meta'] = np.array(json.dumps(meta))
    path = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: calls={len(calls)} N_final={h.shape[0]} loss={loss.item():.6f} '
          f'-> {os.path.getsize(path) / 1024:.0f} KiB')


def run_loss_fixture(name, out_dir, seed, T=5, dmean=4):
    """Targets and losses of the reference (models/loss.py) on reference-built train-mode graphs."""
    from models.loss import CELoss, FocalLoss, create_targets
    from utils.graph import initialize_graph, update_graph
    X, y = synth_sequence(seed, T, dmean, 3, '2d', fp_rate=0.25)
    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    graphs = [(node_adj, labels, y_pred)]
    for t in range(t_st, t_end):
        y_pred, feats, node_adj, edge_adj, labels = update_graph(
            node_adj, labels, torch.zeros(node_adj.shape[0], 1), y_pred, X, y, t, mode='train', cuda=False)
        graphs.append((node_adj, labels, y_pred))
    gen = torch.Generator().manual_seed(seed + 5)
    out = {}
    for c, (na, lab, yp) in enumerate(graphs):
        N = na.shape[0]
        logits = (2.0 * torch.randn(N, 1, generator=gen)).requires_grad_(True)
        scores = torch.sigmoid(logits)
        idx_edge = torch.nonzero((yp[:, 0] == -1))[:, 0]
        idx_node = torch.nonzero((yp[:, 0] != -1))[:, 0]
        targets = create_targets(lab, na, idx_node)
        loss_c = CELoss()(logits, targets, na, idx_node)
        loss_f = FocalLoss(gamma=0)(scores[idx_node, 0], targets[idx_node]) + FocalLoss(gamma=0)(scores[idx_edge, 0], targets[idx_edge])
        loss_g = FocalLoss(gamma=2, alpha=0.25, size_average=False)(scores[idx_edge, 0], targets[idx_edge])
        (loss_c + loss_f + 0.5 * loss_g).backward()
        pre = f'c{c}/'
        idx, val, dense = adj_arrays(na.detach())
        out[pre + 'node_adj_idx'] = idx.astype(np.int64)
        out[pre + 'node_adj_val'] = val.astype(np.float32)
        out[pre + 'node_adj_dense'] = np.int64(dense)
        out[pre + 'N'] = np.int64(N)
        out[pre + 'labels'] = lab.numpy().copy()
        out[pre + 'logits'] = logits.detach().numpy().copy()
        out[pre + 'targets'] = targets.numpy().copy()
        out[pre + 'loss_c'] = np.float64(loss_c.item())
        out[pre + 'loss_f'] = np.float64(loss_f.item())
        out[pre + 'loss_g'] = np.float64(loss_g.item())
        out[pre + 'grad_logits'] = logits.grad.numpy().copy()
    out['meta'] = np.array(json.dumps(dict(name=name, ncalls=len(graphs), seed=seed, kind='loss',
                                           reference='arangesh/TrackMPNN models/loss.py')))
    path = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: calls={len(graphs)} -> {os.path.getsize(path) / 1024:.0f} KiB')



def run_train_chunk_fixture(name, out_dir, seed, H=64, sequence=None, T=6, dmean=4):
    """One training chunk of train.py:54-135 on the real reference: initialize_graph / update_graph(mode='train'), the
    model in train mode with the hidden state carried, create_targets + CELoss + FocalLoss (node and edge, the default
    --tp-classifier) after every forward, ONE backward of loss_c + loss_f.  Stored: the sequence, the parameters, the
    loss terms of every call, the total and every parameter gradient (and the BatchNorm buffers after the chunk)."""
    from models.loss import CELoss, FocalLoss, create_targets
    from models.track_mpnn import TrackMPNN
    from utils.graph import initialize_graph, update_graph

    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, H, 0, 'diff')
    gp = torch.Generator().manual_seed(seed + 77)
    with torch.no_grad():
        for k, prm in model.named_parameters():
            prm.add_(0.1 * torch.randn(prm.shape, generator=gp))
            if k.startswith('output_transform') and k.endswith('bias'):
                prm.copy_(0.5 * torch.randn(prm.shape, generator=gp))    # scores on both sides of 0.5
    model.train()
    out = {}
    for k, v in model.state_dict().items():
        out['param/' + k] = v.detach().numpy().copy()
    X, y = sequence if sequence is not None else synth_sequence(seed, T, dmean, 3, '2d', fp_rate=0.25)
    out['X'] = X.numpy().copy()
    out['y'] = y.numpy().copy()
    ce_loss, focal_node, focal_edge = CELoss(), FocalLoss(gamma=0), FocalLoss(gamma=0)

    def terms(scores, logits, labels, node_adj, y_pred):
        idx_edge = torch.nonzero((y_pred[:, 0] == -1))[:, 0]
        idx_node = torch.nonzero((y_pred[:, 0] != -1))[:, 0]
        targets = create_targets(labels, node_adj, idx_node)
        lc = ce_loss(logits, targets, node_adj, idx_node)
        lf = focal_node(scores[idx_node, 0], targets[idx_node]) + focal_edge(scores[idx_edge, 0], targets[idx_edge])
        return lc, lf

    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    scores, logits, states, _ = model(feats, None, node_adj, edge_adj)
    loss_c, loss_f = terms(scores, logits, labels, node_adj, y_pred)
    per_call = [(float(loss_c), float(loss_f), int(logits.shape[0]))]
    for t_cur in range(t_st, t_end):
        scores2 = torch.cat((1 - scores, scores), dim=1)
        y_pred, feats, node_adj, edge_adj, labels = update_graph(node_adj, labels, scores2, y_pred, X, y, t_cur,
                                                                 use_hungraian=False, mode='train', cuda=False)
        scores, logits, states, _ = model(feats, states, node_adj, edge_adj)
        lc, lf = terms(scores, logits, labels, node_adj, y_pred)
        loss_c, loss_f = loss_c + lc, loss_f + lf
        per_call.append((float(lc), float(lf), int(logits.shape[0])))
    loss = loss_c + loss_f
    loss.backward()
    out['loss_c'] = np.float64(loss_c.item())
    out['loss_f'] = np.float64(loss_f.item())
    out['per_call'] = np.asarray(per_call, dtype=np.float64)
    for k, prm in model.named_parameters():
        out['grad/' + k] = prm.grad.numpy().copy()
    for k, b in model.named_buffers():
        out['buf_after/' + k] = b.detach().numpy().copy()
    meta = dict(name=name, kind='train_chunk', features='2d', ncategories=3, nhidden=H, nattheads=0, msg_type='diff',
                ncalls=len(per_call), seed=seed, torch=torch.__version__, reference='arangesh/TrackMPNN train.py:54-135 chunk')
    out['meta'] = np.array(json.dumps(meta))
    path = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: calls={len(per_call)} N_final={per_call[-1][2]} loss_c={loss_c.item():.4f} loss_f={loss_f.item():.4f} '
          f'-> {os.path.getsize(path) / 1024:.0f} KiB')


def window_sequence(seed, frames, mean_dets, max_dets, ncat):
    """One KITTI/BDD-shaped chunk of the bench generator: y from trackmpnn_amd.graph.synth_window
    (a pure numpy function: track survival 0.9, 10 % false positives, 20 % missed detections), X ~ N(0, 1)."""
    from trackmpnn_amd.graph import synth_window
    yy = synth_window(seed, frames, mean_dets, max_dets)
    y = torch.from_numpy(yy)[None]
    X = torch.randn(1, yy.shape[0], ncat + 5, generator=torch.Generator().manual_seed(seed + 1000))
    return X, y


def node_adj_diag_zero(na):
    """bool [N]: rows of node_adj whose diagonal entry is 0, i.e. the edge rows."""
    d = na.to_dense() if na.is_sparse else na
    return torch.diagonal(d) == 0


def run_infer_fixture(name, out_dir, seed, T, dmean, cur_win, ret_win, hungarian, H=32, K=0, msg='diff', light=False,
                      gap=None, shuffle=False, no_tp=False):
    """The inference loop of infer.py:48-87 on the real reference: eval-mode model, update_graph(mode='test'),
    decode_tracks(cuda=False) with its row deletion between calls.  Every forward call is stored with ITS inputs
    (x, the row-deleted carried state, the adjacency pair as produced) and outputs; every decode step with the rows
    it kept (read off a marker array passed as `labels`), y_pred before / after and the finalised tracks y_out.
    light=True (dense scenes, thousands of rows): only what drives and checks the graph maintenance is kept -- the
    scores of every call, y_pred, kept rows, y_out -- no states, adjacency or parameters.
    no_tp=True: a model trained with --no-tp-classifier (the reference README's commands, README.md:52-67): every detection counts
    as a true positive -- its row of `scores` is overwritten with (0, 1) after every model call (infer.py:53-56, 77-80)."""
    from models.track_mpnn import TrackMPNN
    from utils.graph import decode_tracks, initialize_graph, update_graph

    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, H, K, msg)
    gp = torch.Generator().manual_seed(seed + 77)
    with torch.no_grad():
        for k, prm in model.named_parameters():
            prm.add_(0.3 * torch.randn(prm.shape, generator=gp))
            if k.startswith('output_transform') and k.endswith('bias'):
                prm.copy_(0.5 * torch.randn(prm.shape, generator=gp))    # scores on both sides of 0.5
        for k, b in model.named_buffers():
            if k.endswith('running_mean'):
                b.copy_(0.2 * torch.randn(b.shape, generator=gp))
            elif k.endswith('running_var'):
                b.copy_(0.5 + torch.rand(b.shape, generator=gp))
    model.eval()
    out = {}
    if not light:
        for k, v in model.state_dict().items():
            out['param/' + k] = v.detach().numpy().copy()
    X, y = synth_sequence(seed, T, dmean, 3, '2d', fp_rate=0.2)
    if gap is not None:
        # no detections at timesteps gap[0] .. gap[1] - 1: the rolling window runs empty and infer.py:62-68 starts the
        # graph again from the next two non-empty timesteps
        keep_det = (y[0, :, 0] < gap[0]) | (y[0, :, 0] >= gap[1])
        X, y = X[:, keep_det], y[:, keep_det]
    if shuffle:
        # the detections listed in NO particular order (the reference indexes them by np.where(y[:, 0] == t) and finalises
        # tracks in det-id order, utils/graph.py:456-490: the track numbering depends on that order)
        perm = torch.randperm(y.shape[1], generator=torch.Generator().manual_seed(seed + 99))
        X, y = X[:, perm], y[:, perm]
    out['X'] = X.numpy().copy()
    out['y'] = y.numpy().copy()
    y_out = y.squeeze(0).numpy().astype('int64').copy()
    y_out[:, 1] = -1
    n_reinit = 0

    def record_call(c, x, h_in, na, ea, scores, logits, h, y_pred):
        pre = f'c{c}/'
        out[pre + 'x'] = x.detach().numpy().copy()
        if light:
            out[pre + 'N'] = np.int64(logits.shape[0])
            out[pre + 'E'] = np.int64(int((node_adj_diag_zero(na)).sum()))
            out[pre + 'scores'] = scores.detach().numpy().copy()
            out[pre + 'y_pred'] = y_pred.numpy().astype(np.int32)
            return
        out[pre + 'has_h_in'] = np.int64(h_in is not None)
        if h_in is not None:
            out[pre + 'h_in'] = h_in.detach().numpy().copy()
        for nm, a in (('node_adj', na), ('edge_adj', ea)):
            idx, val, dense = adj_arrays(a.detach())
            out[pre + nm + '_idx'] = idx.astype(np.int64)
            out[pre + nm + '_val'] = val.astype(np.float32)
            out[pre + nm + '_dense'] = np.int64(dense)
        out[pre + 'N'] = np.int64(logits.shape[0])
        out[pre + 'scores'] = scores.detach().numpy().copy()
        out[pre + 'logits'] = logits.detach().numpy().copy()
        out[pre + 'h_out'] = h.detach().numpy().copy()
        out[pre + 'y_pred'] = y_pred.numpy().copy()

    with torch.no_grad():
        y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, t_st=0, mode='test', cuda=False)
        scores, logits, states, _ = model(feats, None, node_adj, edge_adj)
        c = 0
        record_call(c, feats, None, node_adj, edge_adj, scores, logits, states, y_pred)
        scores = torch.cat((1 - scores, scores), dim=1)
        if no_tp:
            idx_node = torch.nonzero((y_pred[:, 0] != -1))[:, 0]
            scores[idx_node, 0] = 0
            scores[idx_node, 1] = 1
        nsteps = 0
        t_skip = t_st
        for t_cur in range(t_st, t_end):
            if t_cur < t_skip:
                continue
            if feats.size()[0] == 0 and states.size()[0] == 0:
                y_pred, feats, node_adj, edge_adj, labels, t_skip, _ = initialize_graph(X, y, t_st=t_cur, mode='test', cuda=False)
                if y_pred is None:
                    break
                states = None
                n_reinit += 1
            else:
                y_pred, feats, node_adj, edge_adj, labels = update_graph(
                    node_adj, labels, scores, y_pred, X, y, t_cur, use_hungraian=hungarian, mode='test', cuda=False)
            h_in = states
            scores, logits, states, _ = model(feats, states, node_adj, edge_adj)
            c += 1
            record_call(c, feats, h_in, node_adj, edge_adj, scores, logits, states, y_pred)
            scores = torch.cat((1 - scores, scores), dim=1)
            if no_tp:
                idx_node = torch.nonzero((y_pred[:, 0] != -1))[:, 0]
                scores[idx_node, 0] = 0
                scores[idx_node, 1] = 1
            t_upto = t_end if t_cur == t_end - 1 else t_cur - cur_win + 2
            N = states.shape[0]
            marker = torch.arange(N, dtype=torch.int64)
            y_pred_b = y_pred.clone()
            y_pred, y_out, states, node_adj, kept, scores = decode_tracks(
                states, node_adj, marker, scores, y_pred, y_out, t_upto, ret_win, use_hungraian=hungarian, cuda=False)
            labels = labels[kept]
            pre = f'd{c}/'
            out[pre + 'keep'] = kept.numpy().astype(np.int32) if light else kept.numpy().copy()
            out[pre + 't_upto'] = np.int64(t_upto)
            if light:
                out[pre + 'n_before'] = np.int64(y_pred_b.shape[0])
                out[pre + 'y_pred_after'] = y_pred.numpy().astype(np.int32)
                out[pre + 'y_out'] = y_out.astype(np.int32)
            else:
                out[pre + 'y_pred_before'] = y_pred_b.numpy().copy()
                out[pre + 'y_pred_after'] = y_pred.numpy().copy()
                out[pre + 'y_out'] = y_out.copy()
                out[pre + 'h_kept'] = states.numpy().copy()
            nsteps += 1
    ncalls = c + 1
    n_del = sum(int((out[f'd{i}/n_before'] if light else out[f'd{i}/y_pred_before'].shape[0]) - out[f'd{i}/keep'].shape[0])
                for i in range(1, ncalls))
    frac_pos = float(np.mean(np.concatenate([out[f'c{i}/scores'].ravel() for i in range(ncalls)]) >= 0.5))
    meta = dict(name=name, kind='infer_light' if light else 'infer', features='2d', ncategories=3, nhidden=H, nattheads=K, msg_type=msg, mode='eval',
                ncalls=ncalls, seed=seed, T=T, cur_win_size=cur_win, ret_win_size=ret_win, hungarian=bool(hungarian),
                rows_deleted=n_del, reinitialisations=n_reinit, torch=torch.__version__,
                reference='arangesh/TrackMPNN infer.py:48-87 loop')
    if no_tp:
        meta['tp_classifier'] = False
    if shuffle:
        meta['detections'] = 'listed in shuffled order'
    out['meta'] = np.array(json.dumps(meta))
    path = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: calls={ncalls} rows deleted={n_del} tracks={int(y_out[:, 1].max()) + 1} '
          f'score>=0.5: {frac_pos:.2f} -> {os.path.getsize(path) / 1024:.0f} KiB')


def run_init_fixture(out_dir):
    """Un-perturbed initial weights of the reference under torch.manual_seed(5) (train.py:42-45,318): one small
    state_dict in full and SHA-256 digests of every tensor for the larger configurations (bit-equality test)."""
    import hashlib
    from models.track_mpnn import TrackMPNN
    out = {}
    digests = {}
    for tag, (feats, ncat, H, K, msg) in (('2d_h32_k0_diff', ('2d', 3, 32, 0, 'diff')),
                                           ('2d_h64_k0_diff', ('2d', 3, 64, 0, 'diff')),
                                           ('bdd_h64_k0_diff', ('2d', 8, 64, 0, 'diff')),
                                           ('2d-temp-vis_h64_k2_concat', ('2d+temp+vis', 3, 64, 2, 'concat')),
                                           ('2d_h256_k0_diff', ('2d', 3, 256, 0, 'diff'))):
        torch.manual_seed(5)
        sd = TrackMPNN(feats, ncat, H, K, msg).state_dict()
        digests[tag] = dict(args=[feats, ncat, H, K, msg],
                            sha256={k: hashlib.sha256(v.numpy().tobytes()).hexdigest() for k, v in sd.items()})
        if tag == '2d_h32_k0_diff':
            for k, v in sd.items():
                out['full/' + k] = v.numpy().copy()
    out['meta'] = np.array(json.dumps(dict(name='init_seed5', kind='init', seed=5, digests=digests,
                                           torch=torch.__version__)))
    path = os.path.join(out_dir, 'init_seed5.npz')
    np.savez_compressed(path, **out)
    print(f'init_seed5: {len(digests)} configurations -> {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference-path', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    ap.add_argument('--only-new', action='store_true', help='skip the round-1 fixtures (debugging aid)')
    ap.add_argument('--only', default='', help='generate only the fixture of this name (round-3 additions)')
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.reference_path)
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(1)
    if args.only.startswith('chunk_'):
        run_train_chunk_fixture('chunk_c2_kitti_car_w5', args.out, 800, H=64, sequence=window_sequence(1001, 7, 6.0, 20, 3))
        run_train_chunk_fixture('chunk_small_h32', args.out, 801, H=32)
        return
    if args.only == 'infer_greedy_w3_r0_reinit':
        run_infer_fixture('infer_greedy_w3_r0_reinit', args.out, 503, T=14, dmean=4, cur_win=3, ret_win=0, hungarian=False,
                          gap=(5, 10))
        return
    if args.only == 'unsorted':
        run_unsorted_fixtures(args.out)
        return
    if args.only == 'wide':
        run_wide_fixtures(args.out)
        return
    if args.only == 'notp':
        run_notp_fixtures(args.out)
        return
    seed = 0
    for feats, ncat in (('2d', 3), ('2d+temp+vis', 3)):
        for msg in ('diff', 'concat'):
            for K in (0, 2):
                for mode in ('train', 'eval'):
                    # H=64 (the headline width) where it is cheap to store, H=32 elsewhere
                    H = 64 if (feats == '2d' and K == 0 and mode == 'train') else 32
                    if feats != '2d' and (msg, K, mode) not in (('diff', 0, 'train'), ('diff', 2, 'eval'),
                                                                ('concat', 2, 'train'), ('concat', 0, 'eval')):
                        seed += 1
                        continue    # three-group fixtures are bulky; keep one per factor level
                    fname = f"roll_{feats.replace('+', '-')}_{msg}_k{K}_{mode}"
                    run_fixture(fname, args.out, feats, ncat, H, K, msg, mode, seed)
                    seed += 1
    # BDD-shaped input width (ncat=8 -> F=13), one rolling train fixture
    run_fixture('roll_bdd_diff_k0_train', args.out, '2d', 8, 64, 0, 'diff', 'train', 100, T=5, dmean=4)
    for i in range(3):
        run_loss_fixture(f'loss_{i}', args.out, 300 + i)
    # C1 of BASELINE.json: static 5-frame window, 20 dets/frame, 64-d, 2 MP iterations
    run_c1(args.out)
    # round 2: one rolling train-mode window each at the real size of BASELINE.json configs[1..3] (bench generator),
    # weights perturbed by 0.1 (see run_c1), h_out kept for every third call
    run_fixture('roll_c2_kitti_car_w5', args.out, '2d', 3, 64, 0, 'diff', 'train', 400, pscale=0.1,
                sequence=window_sequence(1001, 7, 6.0, 20, 3), h_every=3)
    run_fixture('roll_c3_kitti_all_w10', args.out, '2d', 3, 64, 0, 'diff', 'train', 401, pscale=0.1,
                sequence=window_sequence(1002, 12, 8.0, 25, 3), h_every=4)
    run_fixture('roll_c4_bdd_w5', args.out, '2d', 8, 64, 0, 'diff', 'train', 402, pscale=0.1,
                sequence=window_sequence(1003, 7, 12.0, 40, 8), h_every=3)
    # the inference loop (update_graph(mode='test') + decode_tracks row deletion), greedy and Hungarian matching
    run_infer_fixture('infer_greedy_w3_r0', args.out, 500, T=9, dmean=4, cur_win=3, ret_win=0, hungarian=False)
    run_infer_fixture('infer_greedy_w4_r2', args.out, 501, T=10, dmean=5, cur_win=4, ret_win=2, hungarian=False, H=64)
    run_infer_fixture('infer_hungarian_w3_r1', args.out, 502, T=9, dmean=4, cur_win=3, ret_win=1, hungarian=True)
    run_init_fixture(args.out)
    # hidden widths between the instantiated kernel widths (the reference takes any int, utils/training_options.py:22):
    # the GPU path runs them zero-padded to 64 / 32
    run_fixture('roll_2d_diff_k0_train_h48', args.out, '2d', 3, 48, 0, 'diff', 'train', 600)
    run_fixture('roll_2d-temp_concat_k2_eval_h20', args.out, '2d+temp', 3, 20, 2, 'concat', 'eval', 601)
    # a dense scene (~70 dets per frame): inference graphs beyond 4096 rows, graph maintenance only (light)
    run_infer_fixture('dense_infer_greedy_w3_r1', args.out, 700, T=6, dmean=70, cur_win=3, ret_win=1, hungarian=False, light=True)
    # round 3: whole training chunks (train.py:54-135 with the reference's own targets and losses)
    run_train_chunk_fixture('chunk_c2_kitti_car_w5', args.out, 800, H=64, sequence=window_sequence(1001, 7, 6.0, 20, 3))
    run_train_chunk_fixture('chunk_small_h32', args.out, 801, H=32)
    # round 3: a sequence with five empty timesteps in the middle -- the window runs empty and the loop takes the
    # re-initialisation branch (infer.py:62-68)
    run_infer_fixture('infer_greedy_w3_r0_reinit', args.out, 503, T=14, dmean=4, cur_win=3, ret_win=0, hungarian=False,
                      gap=(5, 10))
    run_unsorted_fixtures(args.out)
    run_wide_fixtures(args.out)
    run_notp_fixtures(args.out)


def run_wide_fixtures(out_dir):
    """round 6: hidden widths served by the wide-cell kernels (csrc/wide.hip: H >= 128), pinned to the real reference like the
    H <= 64 paths (so far they were compared with the oracle only).  Perturbation 0.15: the same per-layer gain as 0.3 at H = 32."""
    run_fixture('roll_2d_diff_k0_train_h128', out_dir, '2d', 3, 128, 0, 'diff', 'train', 900, T=5, dmean=4, pscale=0.15)
    run_fixture('roll_2d_concat_k0_eval_h128', out_dir, '2d', 3, 128, 0, 'concat', 'eval', 901, T=4, dmean=3, pscale=0.15)


def run_unsorted_fixtures(out_dir):
    """round 4: the detections of the sequence listed in shuffled order (the reference finalises tracks in det-id order,
    utils/graph.py:456-490, so the track numbering depends on it)."""
    run_infer_fixture('infer_greedy_w3_r0_unsorted', out_dir, 504, T=10, dmean=5, cur_win=3, ret_win=0, hungarian=False,
                      shuffle=True)
    run_infer_fixture('infer_hungarian_w4_r1_unsorted', out_dir, 505, T=9, dmean=4, cur_win=4, ret_win=1, hungarian=True,
                      shuffle=True)


def run_notp_fixtures(out_dir):
    """round 6: the inference loop as the reference README runs it (README.md:52-67, 113-122): a model trained with
    --no-tp-classifier, --hungarian association (and the greedy rule), cur_win_size 5, ret_win_size 0, H = 64."""
    run_infer_fixture('infer_hungarian_w5_r0_notp', out_dir, 506, T=9, dmean=4, cur_win=5, ret_win=0, hungarian=True, H=64,
                      no_tp=True)
    run_infer_fixture('infer_greedy_w5_r0_notp', out_dir, 507, T=8, dmean=3, cur_win=5, ret_win=0, hungarian=False, H=64,
                      no_tp=True)


def run_c1(out_dir):
    """C1: every frame a random permutation of ids 0..D-1 (all TPs), X ~ N(0,1) seed 0."""
    T, D = 5, 20

    def c1_seq(seed, T_, dmean, ncat, feats, **kw):
        g = torch.Generator().manual_seed(0)
        X = torch.randn(1, T * D, ncat + 5, generator=g)
        y = torch.zeros(1, T * D, 2, dtype=torch.int64)
        for t in range(T):
            y[0, t * D:(t + 1) * D, 0] = t
            y[0, t * D:(t + 1) * D, 1] = torch.randperm(D, generator=g)
        return X, y

    global synth_sequence
    keep = synth_sequence
    synth_sequence = c1_seq
    try:
        # weights perturbed by 0.1 (not 0.3): at N=1700 with 40 incident edges per det the 0.3 problem is
        # ill-conditioned -- an fp64 evaluation differs from the reference's own fp32 result by 1.4e-3 in
        # the logits, so no implementation could be pinned to 1e-4 on it
        run_fixture('c1_static_diff_k0_train', out_dir, '2d', 3, 64, 0, 'diff', 'train', 200, T=T, static_iters=2,
                    pscale=0.1)
    finally:
        synth_sequence = keep


if __name__ == '__main__':
    main()
