"""Wide cells (H = 128 / 256): the det-side form of the W_ih backward products (tmpnn_wide_gru_bwd_diff, default) against
the per-edge products (TMPNN_WIDE_DET=0) on the same inputs -- all parameter gradients and d_x / d_h."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trackmpnn_amd.functional as F
from trackmpnn_amd import TrackMPNN, dense_static_graph, plan_single

dev = torch.device('cuda:0')
for H, T, D in ((128, 6, 12), (256, 5, 20), (256, 12, 60)):
    g = dense_static_graph(T, D).to(dev)
    res = []
    for det in (False, True):
        F.WIDE_DET = det
        torch.manual_seed(5)
        model = TrackMPNN('2d', 3, H, 0, 'diff').to(dev).train()
        with torch.no_grad():
            gen = torch.Generator().manual_seed(1)
            for p in model.parameters():
                p.add_((1.0 / H ** 0.5) * torch.randn(p.shape, generator=gen).to(dev))
        x = torch.zeros(g.N, 8, device=dev)
        x[g.det_row.long()] = torch.randn(g.Dn, 8, generator=torch.Generator().manual_seed(2)).to(dev)
        x.requires_grad_(True)
        h, loss = None, 0.0
        for it in range(3):
            s, l, h, _ = model.forward_graph(x if it == 0 else x[:0], h, plan_single(g, g.N if it == 0 else 0))
            loss = loss + (l * l).sum() + s.sum()
        loss.backward()
        res.append([p.grad.clone() for p in model.parameters()] + [x.grad.clone()])
    g0, g1 = res
    names = [n for n, _ in model.named_parameters()] + ['d_x']
    gmax = max(float(a.abs().max()) for a in g0[:-1])          # parameter gradients against the largest of them
    worst = max(((float((a - b).abs().max()) / (gmax if n != 'd_x' else float(a.abs().max())), n)
                 for a, b, n in zip(g0, g1, names)))
    print(f'H={H} N={g.N} E={g.E}: worst gradient difference {worst[0]:.2e} of max|grad| ({worst[1]})', flush=True)
    assert worst[0] < 2e-4, worst
