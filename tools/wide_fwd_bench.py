"""Wide edge-cell forward on the C5 graph (static 50 x 300, H = 256): per-row gathers (tmpnn_wide_gru_fwd) against edge
tiles (tmpnn_wide_gru_fwd_tiled), HIP events on the launch stream; also the bit-equality of the two at full size."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trackmpnn_amd import _lib
from trackmpnn_amd.graph import build_edge_tiles, dense_static_graph

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=50); ap.add_argument('--dets', type=int, default=300)
    ap.add_argument('--hidden', type=int, default=256); ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    H = a.hidden
    g = dense_static_graph(a.frames, a.dets, 'cpu').to(dev)
    t0 = time.time(); tiles = build_edge_tiles(g, 128, stats=True); torch.cuda.synchronize()
    cnt = (tiles.t_dptr[1:] - tiles.t_dptr[:-1])
    print(f'graph N={g.N} E={g.E} Dn={g.Dn}; {tiles.T} tiles built in {time.time() - t0:.2f}s, dets per tile mean '
          f'{float(cnt.float().mean()):.1f} max {tiles.max_dets}, listed (> 40): {int((cnt > 40).sum())}', flush=True)
    torch.manual_seed(0)
    h = torch.randn(g.N, H, device=dev)
    sc = 1.0 / H ** 0.5
    wih, whh = sc * torch.randn(3 * H, H, device=dev), sc * torch.randn(3 * H, H, device=dev)
    bih, bhh = 0.3 * torch.randn(3 * H, device=dev), 0.3 * torch.randn(3 * H, device=dev)
    lib = _lib.load()
    prep = torch.empty(int(lib.tmpnn_wide_prep_bytes(H, H)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('tmpnn_wide_prepare', wih.data_ptr(), whh.data_ptr(), H, H, prep.data_ptr(), st)
    P = torch.empty(g.Dn, 3 * H, device=dev)
    res = {}
    outs = {}
    for name in ('rows', 'tiled'):
        if a.only and a.only != name:
            continue
        out = torch.zeros(g.N, H, device=dev)
        gates = torch.zeros(4, g.N, H, device=dev)

        def run():
            if name == 'tiled':
                _lib.call('tmpnn_wide_gru_fwd_tiled', prep.data_ptr(), g.det_row.data_ptr(), g.Dn, tiles.cref(), g.E,
                          h.data_ptr(), H, H, bih.data_ptr(), bhh.data_ptr(), P.data_ptr(), out.data_ptr(), H,
                          gates.data_ptr(), g.N * H, st)
            else:
                _lib.call('tmpnn_wide_gru_fwd', prep.data_ptr(), g.det_row.data_ptr(), g.Dn, g.edge_row.data_ptr(), g.E,
                          g.src_pos.data_ptr(), g.dst_pos.data_ptr(), h.data_ptr(), H, H, bih.data_ptr(), bhh.data_ptr(),
                          P.data_ptr(), out.data_ptr(), H, gates.data_ptr(), g.N * H, st)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        alg = (24 * H + 12) * g.E + 28 * H * g.Dn
        res[name] = dict(ms=ms, alg_GB=alg / 1e9, TBps=alg / ms / 1e9, frac_hbm=alg / ms / 1e9 / 8.0,
                         tflops_f32eq=6.0 * H * H * g.E / ms / 1e9)
        print(name, json.dumps(res[name]), flush=True)
        outs[name] = (out[g.edge_row.long()[::97]].clone(), gates[:, g.edge_row.long()[::89]].clone(),
                      float(out.double().sum()), float(gates.double().sum()))
        del out, gates
    if len(outs) == 2:
        eq = (torch.equal(outs['rows'][0], outs['tiled'][0]) and torch.equal(outs['rows'][1], outs['tiled'][1])
              and outs['rows'][2] == outs['tiled'][2] and outs['rows'][3] == outs['tiled'][3])
        print('bit-equal (sampled rows + fp64 checksums of h_out and the gate planes):', eq)
        res['bit_equal'] = eq
    print(json.dumps(res))
