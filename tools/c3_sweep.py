"""KITTI All / CenterTrack-shaped rolling windows (12 frames = W 10 + 2, D_t ~ clip(Poisson(8), 1, 25),
F = 8, H = 64, K = 0, diff), B windows batched block-diagonally with B swept over {1, 64, 1024, 16384}: the aggregation
kernels (node -> edge gather, edge -> node segment sum) on the LAST call's graph, HIP events on the launch stream,
algorithmic bytes of SURVEY 8(d) (B_agg) against 8 TB/s.  With --one B only that batch runs (the rocprofv3 PMC passes of
tools/c3_sweep.sh wrap this form and read FETCH_SIZE / WRITE_SIZE per dispatch)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--one', type=int, default=0)
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
H = 64
st = torch.cuda.current_stream().cuda_stream
out = []
for B in ([a.one] if a.one else [1, 64, 1024, 16384]):
    plans, xs, edge_iters = bench.build_batch(B, 12, 8.0, 25, 8, seed=3, device=dev)
    g = plans[-1].graph
    N, E, Dn = g.N, g.E, g.Dn
    h = torch.randn(N, H, device=dev)
    msg = torch.empty(N, H, device=dev)
    es = torch.empty(Dn, H, device=dev)
    gather = lambda: _lib.call('tmpnn_gather_diff_fwd', g.cref(), h.data_ptr(), H, msg.data_ptr(), H, H, 0, st)
    segsum = lambda: _lib.call('tmpnn_segsum_fwd', g.cref(), h.data_ptr(), H, es.data_ptr(), H, H, 0, 1, st)
    b_gather = 4 * H * E + 4 * H * Dn + 8 * E                       # write ns, each det row once, src/dst
    b_segsum = 4 * H * E + 4 * H * Dn + 4 * (2 * E + Dn + 1)        # each edge row once, write, CSR (4 B per incidence here)
    iters = a.iters if not a.one else 3
    tg, ts = bench.time_stage(gather, iters), bench.time_stage(segsum, iters)
    rec = dict(B=B, N=N, E=E, Dn=Dn, state_MB=round(N * H * 4 / 2 ** 20, 2), edges_per_det=round(2 * E / max(Dn, 1), 1),
               gather=dict(ms=round(tg, 5), alg_MB=round(b_gather / 1e6, 3), GBs=round(b_gather / tg / 1e6, 1),
                           hbm_frac=round(b_gather / tg / 1e6 / bench.HBM_PEAK_GBS, 4)),
               segsum=dict(ms=round(ts, 5), alg_MB=round(b_segsum / 1e6, 3), GBs=round(b_segsum / ts / 1e6, 1),
                           hbm_frac=round(b_segsum / ts / 1e6 / bench.HBM_PEAK_GBS, 4)))
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del plans, xs, h, msg, es
    torch.cuda.empty_cache()
