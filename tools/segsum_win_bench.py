"""Window-owned segment sum (k_segsum_win, struct tmpnn_win_plan) against the CSR kernel (k_segsum_pipe) on the last call's graph
of a C2-shaped batch (B windows of 12 frames, H = 64): bitwise equality at full size, then HIP-event times of both, accumulate
off / on.  --one: a single launch of each form per variant (for rocprofv3 --pmc passes)."""
import argparse, copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import _lib
from trackmpnn_amd.graph import win_plan

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=4096)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--shape', default='c3', help='c3: 12 frames, Poisson(8) / 25 (tools/c3_sweep.py) ; c2: the bench step, 7 frames, Poisson(6) / 20')
ap.add_argument('--one', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda:0')
H = 64
st = torch.cuda.current_stream().cuda_stream
plans, xs, _ = (bench.build_batch(a.B, 12, 8.0, 25, 8, seed=3, device=dev) if a.shape == 'c3'
                else bench.build_batch(a.B, 7, 6.0, 20, 8, seed=1, device=dev))
g = plans[-1].graph
wp = win_plan(g)
assert wp is not None
g0 = copy.copy(g)
g0.__dict__.pop('_win_plan', None)
g0._c = None
h = torch.randn(g.N, H, device=dev)
rec = dict(B=a.B, N=g.N, E=g.E, Dn=g.Dn, W=wp.W, nbig=wp.nbig)
alg = 4 * H * g.E + 4 * H * g.Dn + 4 * (2 * g.E + g.Dn + 1)
for acc in (0, 1):
    outs = []
    for graph in (g0, g):
        o = torch.ones(g.Dn, H, device=dev)
        _lib.call('tmpnn_segsum_fwd', graph.cref(), h.data_ptr(), H, o.data_ptr(), H, H, acc, 1, st)
        outs.append(o)
    torch.cuda.synchronize()
    rec[f'bitwise_acc{acc}'] = bool(torch.equal(outs[0], outs[1]))
    if a.one:
        continue
    for name, graph in (('pipe', g0), ('win', g)):
        o = torch.ones(g.Dn, H, device=dev)
        f = lambda: _lib.call('tmpnn_segsum_fwd', graph.cref(), h.data_ptr(), H, o.data_ptr(), H, H, acc, 1, st)
        ms = bench.time_stage(f, a.iters)
        rec[f'{name}_acc{acc}'] = dict(ms=round(ms, 4), GBs=round(alg / ms / 1e6, 1))
print(json.dumps(rec), flush=True)
