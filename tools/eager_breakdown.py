"""Host-side cost of the eager batch-1 path, piece by piece (one C2 window, fixture roll_c2_kitti_car_w5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import Golden
from tests.test_parity_gpu import build_model, DEV
from trackmpnn_amd.dist import GradBucket
from trackmpnn_amd.graph import device_graph_from_adjacency

gold = Golden('roll_c2_kitti_car_w5')
model = build_model(gold.meta, gold.params())
bucket = GradBucket(model)
calls = []
for c in range(gold.ncalls):
    na, ea = gold.adjacency(c, 'node_adj', DEV), gold.adjacency(c, 'edge_adj', DEV)
    if not na.is_sparse:
        na, ea = na.to_sparse(), ea.to_sparse()
    calls.append((gold.t(f'c{c}/x').to(DEV), na, ea))
n = 300


def timed(fn, sync_inside=False):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    enq = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    return enq, tot


def conv():
    return [device_graph_from_adjacency(na, ea, DEV) for _, na, ea in calls]


graphs = conv()
for g in graphs:
    g.check()


def fwd_only():
    h, outs = None, []
    with torch.no_grad():
        for (x, _, _), g in zip(calls, graphs):
            s, l, h, _ = model.forward_dgraph(x, h, g)
    return h


def fwd_grad():
    h, outs = None, []
    for (x, _, _), g in zip(calls, graphs):
        s, l, h, _ = model.forward_dgraph(x, h, g)
        outs.append(l)
    return outs


def fwd_loss_bwd():
    outs = fwd_grad()
    loss = torch.cat(outs).sum()
    bucket.zero()
    loss.backward()


def full():
    h, outs = None, []
    for x, na, ea in calls:
        s, l, h, _ = model(x, h, na, ea)
        outs.append(l)
    loss = torch.cat(outs).sum()
    bucket.zero()
    loss.backward()


for name, fn in (('7 conversions', conv), ('7 forward_dgraph, no grad', fwd_only), ('7 forward_dgraph, grad mode', fwd_grad),
                 ('forward_dgraph + loss + backward', fwd_loss_bwd), ('model(x, h, node_adj, edge_adj) + loss + backward', full)):
    enq, tot = timed(fn)
    print(f'{name:55s} host enqueue {enq:.3f} ms   wall {tot:.3f} ms', flush=True)

# the same window WITHOUT GradBucket (a reference loop that only swaps the import: optimizer.zero_grad() + loss.backward(),
# gradients returned to autograd -- the Python autograd node)
model2 = build_model(gold.meta, gold.params())
opt2 = torch.optim.Adam(model2.parameters(), lr=1e-5)


def plain():
    h, outs = None, []
    for x, na, ea in calls:
        s, l, h, _ = model2(x, h, na, ea)
        outs.append(l)
    loss = torch.cat(outs).sum()
    opt2.zero_grad()
    loss.backward()


enq, tot = timed(plain)
print(f"{'plain loop (no GradBucket): fwd + loss + backward':55s} host enqueue {enq:.3f} ms   wall {tot:.3f} ms", flush=True)
