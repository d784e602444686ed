"""CPU oracle on a bounded sample of the C5 window kind (static T x 300 dets, H = 256, 4 iterations, fwd + bwd of
sum(logits)) on the GPU box's host cores: T = 6 (E = 450 000: one tenth of C5's edges; the oracle is O(E H^2))."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import trackmpnn_oracle as orc
from trackmpnn_amd.graph import dense_static_graph
T, D, H, iters = 6, 300, 256, 4
g = dense_static_graph(T, D, 'cpu')
og = orc.OracleGraph(g.N, g.is_edge.numpy().astype(bool), g.src.numpy().astype(np.int64), g.dst.numpy().astype(np.int64),
                     g.edge_row.numpy().astype(np.int64), g.det_row.numpy().astype(np.int64))
from trackmpnn_amd import TrackMPNN
torch.manual_seed(5)
m = TrackMPNN('2d', 3, H, 0, 'diff')
cfg = orc.OracleConfig('2d', 3, H, 0, 'diff')
p = {k: v.clone() for k, v in m.state_dict().items()}
for k, v in p.items():
    if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
        v.requires_grad_(True)
x = torch.zeros(g.N, 8); x[g.det_row.long()] = torch.randn(g.Dn, 8)
nt = min(16, os.cpu_count() or 1)
torch.set_num_threads(nt)
def run():
    h, loss = None, 0.0
    for it in range(iters):
        s, l, h, _ = orc.forward(p, cfg, x if it == 0 else x[:0], h, og, training=True)
        loss = loss + l.sum()
    for v in p.values(): v.grad = None
    loss.backward()
run()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 15.0 or n < 2:
    run(); n += 1
dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(sample=f'static {T} x {D}, H={H}, {iters} iterations: E={g.E} (C5 has 4 410 000), torch-CPU fp32 oracle fwd+bwd',
                      threads=nt, host_cores=os.cpu_count(), seconds_per_step=dt, edges_per_s=g.E * iters / dt,
                      c5_step_seconds_extrapolated=dt * 4410000 / g.E)))
