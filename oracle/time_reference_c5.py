#!/usr/bin/env python3
"""C5 (static 50 x 300, H = 256, 4 iterations) cannot run on the reference: its forward builds dense N x N tensors
(track_mpnn.py:55-56, layers.py:85-88) and N = 4.425 M.  This script (build container only) times the REAL reference on the
same kind of window -- static T x D graphs, H = 256, fwd + bwd of sum(logits), the graph presented in ONE call plus
iters - 1 empty-x calls -- at the largest sizes that fit here, fits time ~ a N^p, and prints the extrapolation to C5 next to
the oracle (O(E H^2)) timed on the same graphs.  Both numbers are labelled extrapolations wherever they are quoted."""
import json, os, sys, time
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.dont_write_bytecode = True; sys.path.insert(0, '/root/reference')
from oracle import trackmpnn_oracle as orc            # noqa: E402


def static_inputs(T, D, F=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(1, T * D, F, generator=g)
    y = torch.zeros(1, T * D, 2, dtype=torch.int64)
    for t in range(T):
        y[0, t * D:(t + 1) * D, 0] = t
        y[0, t * D:(t + 1) * D, 1] = torch.randperm(D, generator=g)
    return X, y


def ref_graph(X, y):
    from utils.graph import initialize_graph, update_graph
    y_pred, feats, node_adj, edge_adj, labels, t_st, t_end = initialize_graph(X, y, 0, 'train', cuda=False)
    xs = [feats]
    for t in range(t_st, t_end):
        y_pred, feats, node_adj, edge_adj, labels = update_graph(node_adj, labels, torch.zeros(node_adj.shape[0], 1), y_pred,
                                                                 X, y, t, mode='train', cuda=False)
        xs.append(feats)
    return torch.cat(xs, 0), node_adj, edge_adj


def main():
    from models.track_mpnn import TrackMPNN
    H, iters = 256, 4
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, H, 0, 'diff').train()
    cfg = orc.OracleConfig('2d', 3, H, 0, 'diff')
    p = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
            v.requires_grad_(True)
    rows = []
    torch.set_num_threads(8)
    for T, D in ((4, 20), (4, 40), (4, 60), (3, 90)):
        X, y = static_inputs(T, D)
        x_all, na, ea = ref_graph(X, y)
        N = na.shape[0]
        E = (T - 1) * D * D
        og = orc.graph_from_adjacency(na, ea)

        def run_ref():
            h, loss = None, 0.0
            for it in range(iters):
                s, l, h, _ = model(x_all if it == 0 else x_all[:0], h, na, ea)
                loss = loss + l.sum()
            model.zero_grad()
            loss.backward()

        def run_orc():
            h, loss = None, 0.0
            for it in range(iters):
                s, l, h, _ = orc.forward(p, cfg, x_all if it == 0 else x_all[:0], h, og, training=True)
                loss = loss + l.sum()
            for v in p.values():
                v.grad = None
            loss.backward()

        rec = dict(T=T, D=D, N=N, E=E)
        for name, fn in (('reference', run_ref), ('oracle', run_orc)):
            fn()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 6.0 or n < 2:
                fn(); n += 1
            rec[name + '_s'] = (time.perf_counter() - t0) / n
        print(json.dumps(rec), file=sys.stderr, flush=True)
        rows.append(rec)
    N = np.array([r['N'] for r in rows], float); E = np.array([r['E'] for r in rows], float)
    pr = np.polyfit(np.log(N), np.log([r['reference_s'] for r in rows]), 1)
    po = np.polyfit(np.log(E), np.log([r['oracle_s'] for r in rows]), 1)
    N5, E5 = 4425000.0, 4410000.0
    out = dict(host=f'{os.cpu_count()} vCPU build container, 8 threads', H=H, iters=iters, points=rows,
               reference_fit=dict(exponent_in_N=float(pr[0]), seconds_at_C5=float(np.exp(pr[1]) * N5 ** pr[0]),
                                  note='extrapolated; the dense N x N temporaries (78 TB each at C5) make the run impossible'),
               oracle_fit=dict(exponent_in_E=float(po[0]), seconds_at_C5=float(np.exp(po[1]) * E5 ** po[0]),
                               edges_per_s_at_C5=float(E5 * iters / (np.exp(po[1]) * E5 ** po[0])), note='extrapolated from E <= 24 300'))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
