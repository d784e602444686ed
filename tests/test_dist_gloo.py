"""world_size-2 gloo test of the gradient bucket all-reduce (the only collective of the path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from trackmpnn_amd import TrackMPNN
    from trackmpnn_amd.dist import GradBucket, allreduce_grads, shard_windows
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 32, 2, 'concat')
    bucket = GradBucket(model)
    assert bucket.flat.numel() == sum(p.numel() for p in model.parameters())
    # fake per-rank gradients accumulated by autograd-style in-place adds
    gen = torch.Generator().manual_seed(100 + rank)
    local = []
    for p in model.parameters():
        g = torch.randn(p.shape, generator=gen)
        p.grad.add_(g)
        local.append(g)
    assert bucket.check_alias()
    allreduce_grads(model, bucket, world)
    # expected: mean over ranks
    exp = []
    for r in range(world):
        gen = torch.Generator().manual_seed(100 + r)
        exp.append([torch.randn(p.shape, generator=gen) for p in model.parameters()])
    ok = all(torch.allclose(p.grad, sum(e[i] for e in exp) / world, atol=1e-6)
             for i, p in enumerate(model.parameters()))
    # optimizer keeps the aliasing with set_to_none=False and breaks it with True
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.step()
    opt.zero_grad(set_to_none=False)
    ok = ok and bucket.check_alias() and float(bucket.flat.abs().sum()) == 0.0
    opt.zero_grad(set_to_none=True)
    ok = ok and not bucket.check_alias()
    ok = ok and shard_windows(5, rank, world) == list(range(rank, 5, world))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bucket_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
