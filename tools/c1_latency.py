"""C1 of BASELINE.json (one static 5-frame window, 20 dets/frame, H = 64, 2 MP iterations) through the DROP-IN call
`model(x, h_in, node_adj, edge_adj)` with the reference's adjacency tensors (golden fixture c1_static): the latency of
one forward+backward on a single small graph, adjacency -> index conversion included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import Golden
from tests.test_parity_gpu import build_model, DEV

gold = Golden('c1_static_diff_k0_train')
model = build_model(gold.meta, gold.params())
calls = []
for c in range(gold.ncalls):
    calls.append((gold.t(f'c{c}/x').to(DEV), gold.adjacency(c, 'node_adj', DEV), gold.adjacency(c, 'edge_adj', DEV)))
N = calls[-1][1].shape[0]
from trackmpnn_amd import graph_from_adjacency
E = sum(graph_from_adjacency(na, ea).E for _, na, ea in calls)

def step():
    h, loss = None, 0.0
    for x, na, ea in calls:
        s, l, h, _ = model(x, h, na, ea)
        loss = loss + l.sum() + s.sum()
    model.zero_grad(set_to_none=False)
    loss.backward()

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f'C1 ({gold.ncalls} calls, N={N} rows, {E} edge-iterations): {ms:.2f} ms per fwd+bwd step through model(x, h, node_adj, edge_adj) '
      f'= {E / ms * 1e3:.3g} graph-edges/s (launch-latency bound: one window)')

# where the time goes: adjacency -> index conversion alone, and the same step on prebuilt plans
from trackmpnn_amd import plan_single
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    graphs = [graph_from_adjacency(na, ea) for _, na, ea in calls]
torch.cuda.synchronize(); conv = (time.perf_counter() - t0) / n * 1e3
plans = [plan_single(g, x.shape[0]) for g, (x, _, _) in zip(graphs, calls)]
def step2():
    h, loss = None, 0.0
    for (x, _, _), p in zip(calls, plans):
        s, l, h, _ = model.forward_graph(x, h, p)
        loss = loss + l.sum() + s.sum()
    model.zero_grad(set_to_none=False)
    loss.backward()
for _ in range(5): step2()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step2()
torch.cuda.synchronize(); pre = (time.perf_counter() - t0) / n * 1e3
print(f'   adjacency -> index form: {conv:.2f} ms per step; forward_graph on prebuilt plans: {pre:.2f} ms per step')
