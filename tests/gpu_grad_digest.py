"""Child process of tests/test_parity_gpu.py::test_gradients_bitwise_equal_across_processes: one rolling fwd+bwd step
over > 2^20 edge rows in a FRESH process; prints the SHA-256 of all parameter gradients and of the last scores."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from trackmpnn_amd import TrackMPNN
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
    with torch.no_grad():
        gen = torch.Generator().manual_seed(1)
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen).to(dev))
    plans, xs, _ = bench.build_batch(4096, 7, 6.0, 20, 8, seed=3, device=dev)
    assert plans[-1].graph.E >= (1 << 20)
    h, loss = None, 0.0
    for plan, x in zip(plans, xs):
        s, l, h, _ = model.forward_graph(x, h, plan)
        loss = loss + torch.nn.functional.softplus(l).sum()
    loss.backward()
    torch.cuda.synchronize()
    hs = hashlib.sha256()
    for p in model.parameters():
        hs.update(p.grad.cpu().numpy().tobytes())
    print('GRAD_DIGEST', hs.hexdigest(), hashlib.sha256(s.detach().cpu().numpy().tobytes()).hexdigest())


if __name__ == '__main__':
    main()
