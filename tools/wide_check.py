"""GPU check + timing of the wide-cell kernels (H = 128 / 256): wide path vs round 1's f32-MFMA kernels on the same
inputs (forward outputs, gate planes, data gradients), then the C5-shaped timing of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trackmpnn_amd.functional as F
from trackmpnn_amd import TrackMPNN, concat_static_graphs, dense_static_graph, plan_single

dev = torch.device('cuda:0')
for H, T, D in ((128, 6, 12), (256, 5, 20), (256, 12, 60)):
    g = dense_static_graph(T, D).to(dev)
    res = []
    for wide in (False, True):
        F.WIDE = wide
        torch.manual_seed(5)
        model = TrackMPNN('2d', 3, H, 0, 'diff').to(dev).train()
        with torch.no_grad():
            gen = torch.Generator().manual_seed(1)
            for p in model.parameters():
                p.add_((1.0 / H ** 0.5) * torch.randn(p.shape, generator=gen).to(dev))
        x = torch.zeros(g.N, 8, device=dev)
        x[g.det_row.long()] = torch.randn(g.Dn, 8, generator=torch.Generator().manual_seed(2)).to(dev)
        h, loss = None, 0.0
        for it in range(3):
            s, l, h, _ = model.forward_graph(x if it == 0 else x[:0], h, plan_single(g, g.N if it == 0 else 0))
            loss = loss + (l * l).sum() + s.sum()
        loss.backward()
        res.append((s.detach(), h.detach(), [p.grad.clone() for p in model.parameters()]))
    (s0, h0, g0), (s1, h1, g1) = res
    gmax = max(float(a.abs().max()) for a in g0)
    print(f'H={H} N={g.N}: scores {float((s0 - s1).abs().max()):.2e}  h {float((h0 - h1).abs().max()):.2e} (|h| {float(h0.abs().max()):.2f})  '
          f'grads {max(float((a - b).abs().max()) for a, b in zip(g0, g1)) / gmax:.2e} of max', flush=True)
