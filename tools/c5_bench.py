"""C5 of BASELINE.json (dense stress): static 50-frame window, 300 dets/frame, H=256, 4 MP iterations, fwd+bwd."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trackmpnn_amd import TrackMPNN, graph_from_edges
from trackmpnn_amd.graph import CallPlan, plan_single

from trackmpnn_amd.graph import dense_static_graph

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=50); ap.add_argument('--dets', type=int, default=300)
    ap.add_argument('--hidden', type=int, default=256); ap.add_argument('--iters', type=int, default=4)
    ap.add_argument('--steps', type=int, default=2)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    t0 = time.time(); g = dense_static_graph(a.frames, a.dets, 'cpu').to(dev); print('graph', g.N, g.E, g.Dn, f'{time.time()-t0:.1f}s', flush=True)
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, a.hidden, 0, 'diff').to(dev).train()
    x = torch.zeros(g.N, 8, device=dev); x[g.det_row.long()] = torch.randn(g.Dn, 8, device=dev)
    plan0 = plan_single(g, g.N); planr = plan_single(g, 0)
    def step():
        h = None; loss = 0.0
        for it in range(a.iters):
            s, l, h, _ = model.forward_graph(x if it == 0 else x[:0], h, plan0 if it == 0 else planr)
            loss = loss + l.sum()
        for p in model.parameters(): p.grad = None
        loss.backward()
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    sp = g.__dict__.get('_seg_plan')
    print(json.dumps(dict(workload=f'C5 static {a.frames}x{a.dets}, H={a.hidden}, {a.iters} iters', N=g.N, E=g.E, Dn=g.Dn,
                          seg_plan=None if sp is None else dict(T=sp.T, I=sp.I), ms_per_step=dt * 1e3,
                          edges_per_s=g.E * a.iters / dt, tflops=36.0 * a.hidden ** 2 * g.E * a.iters / dt / 1e12,
                          mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30)))
