"""C3 of BASELINE.json (KITTI All / CenterTrack-shaped dets, cur-win-size 10 => 12-frame chunks): rolling fwd+bwd step
time and every stage kernel's achieved HBM GB/s against the 8 TB/s roof.  Synthetic windows of that shape
(D_t ~ clip(Poisson(8), 1, 25), 3 categories => F = 8, H = 64, K = 0, diff), batched block-diagonally."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.dist import GradBucket

ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, default=2048)
ap.add_argument('--frames', type=int, default=12)
ap.add_argument('--mean-dets', type=float, default=8.0)
ap.add_argument('--max-dets', type=int, default=25)
ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(5)
H, F = 64, 8
model = TrackMPNN('2d', F - 5, H, 0, 'diff').to(dev).train()
plans, xs, edge_iters = bench.build_batch(a.windows, a.frames, a.mean_dets, a.max_dets, F, seed=2, device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
targets = [(torch.rand(p.graph.N, 1, device=dev, generator=gen) < 0.3).float() for p in plans]
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
bucket = GradBucket(model)
for _ in range(2):
    bench.step(model, plans, xs, targets, opt, bucket, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    bench.step(model, plans, xs, targets, opt, bucket, 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
t, flops, nbytes = bench.stage_profile(model, plans[-1], H)
g = plans[-1].graph
out = dict(workload=f'C3-shaped: {a.windows} windows x {a.frames} frames, D_t~clip(Poisson({a.mean_dets}),1,{a.max_dets}), H=64, diff',
           rows_final=g.N, edges_final=g.E, dets_final=g.Dn, edge_iterations_per_step=edge_iters,
           ms_per_step=dt * 1e3, graph_edges_per_s=edge_iters / dt, mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30,
           stages={k: dict(ms=round(v, 4), GBs=round(nbytes[k] / (v * 1e-3) / 1e9, 1),
                           hbm_frac=round(nbytes[k] / (v * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 3)) for k, v in t.items()})
print(json.dumps(out))
