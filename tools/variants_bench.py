"""model variants on the batched C2 workload -- attention heads (K = 2), concat messages, three feature groups
(the reference's default --feats 2d+temp+vis, F = 8 + 2 + 128) -- whole-step time and graph-edges/s, 4096 windows.  Run
under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/variants_profile.sh)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.dist import GradBucket
ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, default=4096); ap.add_argument('--steps', type=int, default=4)
ap.add_argument('--only', default='')
a = ap.parse_args()
dev = torch.device('cuda:0')
out = {}
for tag, feats, K, msg, F in (('base', '2d', 0, 'diff', 8), ('att_k2', '2d', 2, 'diff', 8), ('concat', '2d', 0, 'concat', 8),
                              ('g3', '2d+temp+vis', 0, 'diff', 138)):
    if a.only and a.only != tag:
        continue
    torch.manual_seed(5)
    model = TrackMPNN(feats, 3, 64, K, msg).to(dev).train()
    plans, xs, edge_iters = bench.build_batch(a.windows, 7, 6.0, 20, F, seed=1, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    targets = [(torch.rand(p.graph.N, 1, device=dev, generator=gen) < 0.3).float() for p in plans]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    bucket = GradBucket(model)
    for _ in range(2):
        bench.step(model, plans, xs, targets, opt, bucket, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.step(model, plans, xs, targets, opt, bucket, 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    g = plans[-1].graph
    out[tag] = dict(features=feats, K=K, msg=msg, windows=a.windows, E_last=g.E, Dn_last=g.Dn, edge_iterations=edge_iters,
                    ms_per_step=round(dt * 1e3, 3), edges_per_s=round(edge_iters / dt), mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    print(tag, json.dumps(out[tag]), flush=True)
    del model, plans, xs, targets, opt, bucket
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
print(json.dumps(out))
