"""Stage timing of the attention aggregation on the LAST call's graph of the batched C2 workload:
tmpnn_att_fwd (1 GEMM + k_att_score + k_att_fwd) and tmpnn_att_bwd (k_att_bwd_det, k_att_bwd_edge, k_att_bwd_dha + 2 GEMMs)
with HIP events, against the byte model of bench.att_bytes, next to the plain segment sum it replaces.

    python tools/att_bench.py --windows 16384 --K 2 [--train]        (rocprofv3 --kernel-trace --stats for the split)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, default=16384)
ap.add_argument('--K', type=int, default=2)
ap.add_argument('--H', type=int, default=64)
ap.add_argument('--train', action='store_true')
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--frames', type=int, default=7)
ap.add_argument('--mean', type=float, default=6.0)
ap.add_argument('--max', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
plans, xs, _ = bench.build_batch(a.windows, a.frames, a.mean, a.max, 8, seed=1, device=dev)
g = plans[-1].graph
del xs
t, nbytes = bench.att_stage_profile(g, a.H, a.K, a.train, iters=a.iters)
out = dict(windows=a.windows, K=a.K, H=a.H, train=a.train, E=g.E, Dn=g.Dn, N=g.N)
for k, ms in t.items():
    out[k] = dict(ms=round(ms, 4))
    if k in nbytes:
        gbs = nbytes[k] / ms / 1e6
        out[k].update(GB=round(nbytes[k] / 1e9, 3), GBs=round(gbs, 1), frac=round(gbs / bench.HBM_PEAK_GBS, 3))
print(json.dumps(out))
