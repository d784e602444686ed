"""Batch-1 latency through the DROP-IN call `model(x, h_in, node_adj, edge_adj)` with the reference's own adjacency
tensors: C1 of BASELINE.json (static 5-frame window, 20 dets/frame, H = 64, 2 MP iterations; fixture c1_static) and one
C2 window (fixture roll_c2_kitti_car_w5: 6 rolling calls + 1 extra iteration), forward + backward + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import Golden
from tests.test_parity_gpu import build_model, DEV
from trackmpnn_amd.dist import GradBucket


def measure(name, n=200):
    gold = Golden(name)
    model = build_model(gold.meta, gold.params())
    bucket = GradBucket(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    calls = []
    for c in range(gold.ncalls):
        na, ea = gold.adjacency(c, 'node_adj', DEV), gold.adjacency(c, 'edge_adj', DEV)
        if not na.is_sparse:                       # initialize_graph(cuda=True) hands over sparse tensors
            na, ea = na.to_sparse(), ea.to_sparse()
        calls.append((gold.t(f'c{c}/x').to(DEV), na, ea))
    from trackmpnn_amd import graph_from_adjacency
    E = sum(graph_from_adjacency(na, ea).E for _, na, ea in calls)

    def step(with_opt=True):
        h, outs = None, []
        for x, na, ea in calls:
            s, l, h, _ = model(x, h, na, ea)
            outs.append(l)
        loss = torch.cat(outs).sum()
        bucket.zero()
        loss.backward()
        if with_opt:
            opt.step()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        step(False)
    enq = (time.perf_counter() - t0) / n * 1e3          # host time to ENQUEUE a step (no sync inside the loop)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / n * 1e3
    # the same with the graphs cached (conversion excluded): re-use each call's adjacency objects
    # the same step captured once into a hipGraph and replayed (CapturedWindow): forward calls + loss + backward + Adam
    from trackmpnn_amd import CapturedWindow
    opt2 = torch.optim.Adam(model.parameters(), lr=1e-5, capturable=True)
    win = CapturedWindow(model, calls, lambda outs, h: torch.cat([l for _, l in outs]).sum(), optimizer=opt2, bucket=bucket)
    for _ in range(5):
        win.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        win.replay()
    torch.cuda.synchronize()
    ms3 = (time.perf_counter() - t0) / n * 1e3
    print(f'{name}: captured window replay (fwd + loss + bwd + Adam in one hipGraph launch): {ms3:.3f} ms per step '
          f'= {E / ms3 * 1e3:.3g} graph-edges/s')
    print(f'{name}: {gold.ncalls} calls, N={calls[-1][1].shape[0]} rows, {E} edge-iterations: {ms:.3f} ms per fwd+bwd+Adam step, '
          f'{ms2:.3f} ms without the optimizer (host enqueue {enq:.3f} ms) = {E / ms2 * 1e3:.3g} graph-edges/s through model(x, h, node_adj, edge_adj)')
    return ms, ms2, ms3


if __name__ == '__main__':
    if '--profile' in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        measure('roll_c2_kitti_car_w5', n=100)
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(35)
    else:
        measure('c1_static_diff_k0_train')
        measure('roll_c2_kitti_car_w5')
