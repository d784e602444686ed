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


def _gpu_worker(rank, world, port, q):
    """One DP rank on cuda:0 (both ranks share the card: the multi-GPU code path with one GPU): real windows, real
    forward / backward through the HIP kernels, GradBucket all-reduce, Adam."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from trackmpnn_amd import TrackMPNN
    from trackmpnn_amd.dist import GradBucket, allreduce_grads
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)

    def make():
        torch.manual_seed(5)
        m = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
        return m, GradBucket(m), torch.optim.Adam(m.parameters(), lr=1e-3)

    def fwd_bwd(model, bucket, plans, xs):
        h, loss = None, 0.0
        for plan, x in zip(plans, xs):
            s, l, h, _ = model.forward_graph(x, h, plan)
            loss = loss + torch.nn.functional.softplus(l).sum()
        loss.backward()

    shards = [bench.build_batch(32, 7, 6.0, 20, 8, seed=r + 1, device=dev)[:2] for r in range(world)]
    model, bucket, opt = make()
    g_first = None
    for it in range(2):
        bucket.zero()
        fwd_bwd(model, bucket, *shards[rank])
        allreduce_grads(model, bucket, world)
        if it == 0:
            g_first = bucket.flat.clone()
        opt.step()
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    # every replica must hold the same parameters ...
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    same = all(torch.equal(both[0], b) for b in both)
    # ... and the averaged gradient must be what ONE process computes from both shards.  (Gradients, not parameters:
    # Adam turns the rounding noise of exactly-zero gradients -- the pre-BatchNorm bias -- into +-lr updates.)
    ok_ref = True
    if rank == 0:
        ref, rb, _ = make()
        rb.zero()
        for r in range(world):
            fwd_bwd(ref, rb, *shards[r])              # gradients of both shards accumulate in the bucket
        rb.flat.mul_(1.0 / world)
        ok_ref = bool((rb.flat - g_first).abs().max() <= 1e-5 * rb.flat.abs().max())
    q.put((rank, same and ok_ref and bool(torch.isfinite(mine).all())))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_real_steps_keep_replicas_identical():
    """The N > 1 path with real work: two ranks (sharing cuda:0, gloo) run two training steps on different windows --
    forward / backward through the HIP kernels, ONE flat-bucket all-reduce, Adam -- and must end with bit-identical
    parameters that match a single-process run over both shards.  (No scaling curve is measured here or anywhere in
    this round: 8-GPU runs are the driver's.)"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_shard_windows_balances_edge_work():
    """variable graph sizes are dealt by sum of E (longest-processing-time greedy); every rank computes the
    same partition, it covers every window once, and the heaviest rank stays within one window of the mean."""
    import numpy as np
    from trackmpnn_amd.dist import shard_windows
    rng = np.random.RandomState(0)
    for world in (2, 4, 8):
        counts = np.concatenate([rng.poisson(800, 40), rng.poisson(9000, 9), [60000]]).tolist()   # KITTI-, BDD-sized, one huge
        shards = [shard_windows(len(counts), r, world, counts) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(len(counts)))
        loads = [sum(counts[i] for i in s) for s in shards]
        mean = sum(counts) / world
        assert max(loads) <= max(mean + max(c for c in counts if c < 60000), 60000)
        naive = [sum(counts[i] for i in range(r, len(counts), world)) for r in range(world)]
        assert max(loads) <= max(naive)
        assert all(s == sorted(s) for s in shards)
    assert shard_windows(5, 1, 2) == [1, 3]
    assert shard_windows(3, 0, 4, [5, 5, 5]) == [0] and shard_windows(3, 3, 4, [5, 5, 5]) == []
    with pytest.raises(ValueError):
        shard_windows(3, 0, 2, [1, 2])


def _nccl_worker(port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from trackmpnn_amd import TrackMPNN
    from trackmpnn_amd.dist import GradBucket, allreduce_grads
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').to('cuda:0')
    bucket = GradBucket(model)
    bucket.flat.copy_(torch.arange(bucket.flat.numel(), device='cuda:0', dtype=torch.float32) % 97)
    before = bucket.flat.clone()
    allreduce_grads(model, bucket, 1)
    torch.cuda.synchronize()
    ok = torch.equal(bucket.flat, before) and bucket.check_alias() and dist.get_backend() == 'nccl'
    q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_bucket_allreduce_through_rccl_world1():
    """The collective of the N > 1 path on the backend it will use: `nccl` (= RCCL on ROCm) initialised at world size 1
    on cuda:0, the flat gradient bucket handed to all_reduce -- RCCL loads, accepts the bucket and returns it unchanged.
    (A gpurun box has one GPU; the multi-rank exchange itself is covered over gloo above.)"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    ok = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0 and ok


@pytest.mark.gpu
def test_bench_two_rank_branch_runs_end_to_end():
    """bench.py's own world > 1 branch (process-group init, barrier around the build, MAX / SUM reductions of time and
    edges, the `allreduce` block) launched the way the driver launches it -- `python -m torch.distributed.run
    --nproc-per-node 2 bench.py --gpus 2` -- as a FRESH child process (the launcher starts before any GPU call of its
    own), two ranks sharing cuda:0 over gloo.  One JSON line, n_gpus 2, value = both ranks' edges over the max time."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
           '--single-device', '--windows', '256', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-latency',
           '--no-loops', '--extras-c4-windows', '64', '--extras-c5-frames', '6', '--extras-c5-dets', '40']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'sequence-dp2'
    assert out['allreduce']['us'] > 0 and out['allreduce']['bytes'] == 4 * 54914 and out['allreduce']['backend'] == 'gloo'
    # whole-job value: the SUM of both ranks' edge iterations over the MAX time (the ranks hold different seeds, so
    # their counts differ a little: the total lies between 1.8 and 2.2 times rank 0's)
    per_rank = out['config']['edge_iterations_per_gpu_step']
    total = out['value'] * out['ms_per_step'] * 1e-3
    assert 1.8 * per_rank < total < 2.2 * per_rank
    assert out['roofline'] is not None and out['cpu_baseline'] is None
    # BASELINE.json configs[3] / configs[4] behind the C2 region (toy sizes here): C4 = BDD-shaped windows, F = 13, the 221 KB
    # bucket; C5 = one dense static window per rank, H = 256, 4 iterations, the 3.4 MB bucket
    ex = out['scaling_extras']
    assert 'error' not in ex['c4'] and 'error' not in ex['c5'], ex
    c4, c5 = ex['c4'], ex['c5']
    assert c4['n_gpus'] == 2 and c4['allreduce']['bytes'] == 4 * 55234 and c4['allreduce']['us'] > 0
    assert c4['config']['windows_per_gpu'] == 64 and 'F=13' in c4['config']['workload']
    tot4 = c4['value'] * c4['ms_per_step'] * 1e-3
    assert 1.6 * c4['config']['edge_iterations_per_gpu_step'] < tot4 < 2.4 * c4['config']['edge_iterations_per_gpu_step']
    assert c5['n_gpus'] == 2 and c5['allreduce']['bytes'] == 4 * 858626 and c5['allreduce']['us'] > 0
    per5 = c5['config']['edge_iterations_per_gpu_step']
    assert per5 % 4 == 0 and per5 // 4 >= 5 * 40 * 40
    tot5 = c5['value'] * c5['ms_per_step'] * 1e-3
    assert abs(tot5 - 2 * per5) <= 1e-6 * tot5
