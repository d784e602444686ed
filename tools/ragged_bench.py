"""The C2 bench step on 16 384 DISTINCT windows (every window its own seed: no tiling of 64 windows) against bench.py's tiled
batch of the same size: same generator, same step; confirms that the headline number does not profit from identical tiles."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN, WindowBuilder, batch_windows, synth_window
from trackmpnn_amd.dist import GradBucket
ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, default=16384); ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda:0')
B, F = a.windows, 8
t0 = time.time()
wins = [WindowBuilder(synth_window(7000 + s, 7, 6.0, 20)).calls() for s in range(B)]
plans, refs = batch_windows(wins, device='cpu')
gen = torch.Generator().manual_seed(1)
xs = []
for plan, ref in zip(plans, refs):
    x = torch.zeros(plan.n_new, F); x[plan.new_det_local] = torch.randn(len(ref), F, generator=gen); xs.append(x.to(dev))
plans = [p.to(dev) for p in plans]
edge_iters = sum(p.graph.E for p in plans)
print(f'built {B} distinct windows in {time.time() - t0:.0f} s: {edge_iters} edge-iterations, rows {plans[-1].graph.N}', flush=True)
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4); bucket = GradBucket(model)
g2 = torch.Generator().manual_seed(0)
targets = [(torch.rand(p.graph.N, 1, generator=g2) < 0.3).float().to(dev) for p in plans]
for _ in range(2): bench.step(model, plans, xs, targets, opt, bucket, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): bench.step(model, plans, xs, targets, opt, bucket, 1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps(dict(workload=f'C2, {B} DISTINCT windows', edge_iterations=edge_iters, ms_per_step=round(dt * 1e3, 3),
                      edges_per_s=round(edge_iters / dt), ms_per_M_edge_iterations=round(dt * 1e3 / (edge_iters / 1e6), 4))))
