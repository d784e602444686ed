"""Time only the per-stage kernels of bench.py's stage_profile on a C2 batch (fast kernel iteration / PMC runs)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=16384)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
plans, xs, edge_iters = bench.build_batch(args.windows, 7, 6.0, 20, 8, seed=1, device=dev)
t, flops, nbytes = bench.stage_profile(model, plans[-1], 64)
g = plans[-1].graph
out = {k: dict(ms=round(v, 4), TF=round(flops[k] / v / 1e9, 1) if k in flops else None,
               GBs=round(nbytes[k] / v / 1e6, 0) if k in nbytes else None) for k, v in t.items()}
print(json.dumps(dict(E=g.E, Dn=g.Dn, N=g.N, stages=out)))
