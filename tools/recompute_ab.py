"""A/B of the activation policy "save h only, recompute the gates"  at the C2 bench shape.

  part 1  the H = 64 tiled edge forward on the largest graph of the C2 step, with and without its four gate planes
          (what the forward would save), and the one-pass backward next to it (what the recompute is added to)
  part 2  whole C2 steps with TMPNN_RECOMPUTE_GATES=0 / 1 in child processes (the switch is read at import), with a
          digest of the gradients (must be identical)
"""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def child():
    import bench
    from trackmpnn_amd import TrackMPNN
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
    plans, xs, edge_iters = bench.build_batch(16384, 7, 6.0, 20, 8, seed=1, device=dev)

    def step():
        h = None; loss = 0.0
        for c, (plan, x) in enumerate(zip(plans, xs)):
            nxt = plans[c + 1].n_new if c + 1 < len(plans) else 0
            s, l, h, _ = model.forward_graph(x, h, plan, reserve_rows=nxt)
            loss = loss + l.sum() * 1e-3 + s.sum() * 1e-3
        for p in model.parameters(): p.grad = None
        loss.backward()
    for _ in range(2): step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    K = 8
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    m = hashlib.sha256()
    for p in model.parameters(): m.update(p.grad.detach().cpu().numpy().tobytes())
    print(json.dumps(dict(recompute=os.environ.get('TMPNN_RECOMPUTE_GATES', '0'), ms_per_step=round(dt * 1e3, 3),
                          edges_per_s=edge_iters / dt, grad_digest=m.hexdigest()[:20],
                          peak_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))), flush=True)


def kernels():
    import bench
    from trackmpnn_amd import TrackMPNN, _lib
    from trackmpnn_amd.graph import edge_tiles
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
    plans, xs, edge_iters = bench.build_batch(16384, 7, 6.0, 20, 8, seed=1, device=dev)
    g = plans[-1].graph; H = 64; N, E = g.N, g.E
    st = torch.cuda.current_stream().cuda_stream
    P = dict(model.named_parameters()); f = 'factor_grus.0.'
    pad = int(os.environ.get('PLANE_PAD', '0'))                  # floats between two gate planes (channel-alignment experiment)
    h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); gates = torch.empty(4 * (N * H + pad), device=dev)
    wih_t = P[f + 'edge_gru.weight_ih'].detach().t().contiguous(); whh_t = P[f + 'edge_gru.weight_hh'].detach().t().contiguous()
    bih, bhh = P[f + 'edge_gru.bias_ih'].detach(), P[f + 'edge_gru.bias_hh'].detach()
    proj = torch.empty(g.Dn, 3 * H, device=dev)
    _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), g.Dn, h.data_ptr(), H, H, wih_t.data_ptr(), 3 * H, proj.data_ptr(), 3 * H, st)
    tl = edge_tiles(g, int(os.environ.get('TILE_ROWS', '32')))
    whead = torch.randn(H, device=dev); part = torch.empty(8, N, device=dev)
    def mk(save, head):
        def fn():
            _lib.call('tmpnn_gru_fwd_tiles', tl.cref(), E, proj.data_ptr(), 3 * H, h.data_ptr(), H, H, whh_t.data_ptr(),
                      bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H, gates.data_ptr() if save else None, N * H + pad,
                      whead.data_ptr() if head else None, part.data_ptr() if head else None, N, st)
        return fn
    res = dict(E=E, N=N, plane_pad_floats=pad)
    for name, a in (('fwd_tiles gates+head', (1, 1)), ('fwd_tiles no gates, head', (0, 1)), ('fwd_tiles gates, no head', (1, 0))):
        res[name + ' ms'] = round(bench.time_stage(mk(*a)), 4)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    elif len(sys.argv) > 1 and sys.argv[1] == 'kernels':
        kernels()
    else:
        me = os.path.abspath(__file__)
        subprocess.run([sys.executable, me, 'kernels'], check=True)
        for v in ('0', '1', '0', '1'):
            subprocess.run([sys.executable, me, 'child'], check=True, env=dict(os.environ, TMPNN_KEEP_VARIANTS='1', TMPNN_RECOMPUTE_GATES=v))
