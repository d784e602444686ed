"""Gradient digest of a dense static window (H = 256 wide path) for A/B runs of library variants: run once per
TMPNN_LIB_PATH, compare the printed digests (bit-identical variants print identical lines)."""
import argparse, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.graph import dense_static_graph, plan_single

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=7); ap.add_argument('--dets', type=int, default=93)
    ap.add_argument('--hidden', type=int, default=256); ap.add_argument('--iters', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = dense_static_graph(a.frames, a.dets, 'cpu').to(dev)
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, a.hidden, 0, 'diff').to(dev).train()
    x = torch.zeros(g.N, 8, device=dev); x[g.det_row.long()] = torch.randn(g.Dn, 8, device=dev)
    x.requires_grad_(True)
    plan0 = plan_single(g, g.N); planr = plan_single(g, 0)
    h = None; loss = 0.0
    for it in range(a.iters):
        s, l, h, _ = model.forward_graph(x if it == 0 else x[:0], h, plan0 if it == 0 else planr)
        loss = loss + (l * torch.linspace(-1, 1, l.numel(), device=dev).view_as(l)).sum()
    loss.backward()
    torch.cuda.synchronize()
    m = hashlib.sha256()
    for n, p in list(model.named_parameters()) + [('x', x)]:
        m.update(p.grad.detach().cpu().numpy().tobytes())
    print('E', g.E, 'loss', float(loss), 'grad digest', m.hexdigest()[:24], 'absmax', max(float(p.grad.abs().max()) for p in model.parameters()))
