"""Per-stage checks of the C ABI against torch-CPU / oracle formulas (used by test_parity_gpu.py and
runnable as a script on the GPU box: prints one line per stage instead of stopping at the first error)."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle import trackmpnn_oracle as orc
from trackmpnn_amd import WindowBuilder, _lib, batch_windows, synth_window

DEV = 'cuda:0'


def make_graph(B=6, frames=5, mean=4, seed=0):
    wins = [WindowBuilder(synth_window(seed + s, frames, mean, 10)).calls() for s in range(B)]
    plans, _ = batch_windows(wins, static=True)
    return plans[-1].graph


def st():
    return torch.cuda.current_stream().cuda_stream


def idx(t):
    return t.long().cpu()


def check_gather(H, concat, acc, g, gd):
    IN = 2 * H if concat else H
    h = torch.randn(g.N, H + 32)      # ld > H on purpose
    out0 = torch.randn(g.N, IN + 4)
    exp = out0.clone()
    hs, hd = h[idx(g.src), :H], h[idx(g.dst), :H]
    val = torch.cat([hs, hd], 1) if concat else hs - hd
    er = idx(g.edge_row)
    exp[er, :IN] = (exp[er, :IN] + val) if acc else val
    hd_, od = h.to(DEV), out0.to(DEV)
    _lib.call('tmpnn_gather_concat_fwd' if concat else 'tmpnn_gather_diff_fwd', gd.cref(), hd_.data_ptr(), h.shape[1],
              od.data_ptr(), out0.shape[1], H, int(acc), st())
    return (od.cpu() - exp).abs().max().item()


def check_segsum(H, mode, acc, compact, g, gd):
    # mode: 'fwd' (row F), 'gdiff_bwd', 'gconcat_bwd'
    W = 2 * H if mode == 'gconcat_bwd' else H
    x = torch.randn(g.N, W + 4)
    rows_out = g.Dn if compact else g.N
    out0 = torch.randn(rows_out, H + 8)
    exp = out0.clone()
    er, s, d, dr = idx(g.edge_row), idx(g.src), idx(g.dst), idx(g.det_row)
    full = torch.zeros(g.N, H)
    if mode == 'gconcat_bwd':
        full.index_add_(0, s, x[er, :H])
        full.index_add_(0, d, x[er, H:2 * H])
    else:
        full.index_add_(0, s, x[er, :H])
        full.index_add_(0, d, -x[er, :H])
    tgt = torch.arange(g.Dn) if compact else dr
    exp[tgt, :H] = (exp[tgt, :H] + full[dr]) if acc else full[dr]
    xd, od = x.to(DEV), out0.to(DEV)
    if mode == 'fwd':
        _lib.call('tmpnn_segsum_fwd', gd.cref(), xd.data_ptr(), x.shape[1], od.data_ptr(), out0.shape[1], H, int(acc),
                  int(compact), st())
    else:
        name = 'tmpnn_gather_concat_bwd' if mode == 'gconcat_bwd' else 'tmpnn_gather_diff_bwd'
        _lib.call(name, gd.cref(), xd.data_ptr(), x.shape[1], od.data_ptr(), out0.shape[1], H, int(acc), st())
    return (od.cpu() - exp).abs().max().item()


def check_segsum_live(H, compact, g, gd, frac=0.6):
    """tmpnn_segsum_fwd_live: rows >= row_limit are promised to be zero and must not change a bit of the sum -- the limited
    launch on a state whose tail rows ARE zero equals the full read; and the tail is really left unread: poisoned with NaN
    behind the limit, the result is still that of the zero tail."""
    lim = int(g.N * frac)
    x = torch.randn(g.N, H + 4)
    x[lim:] = 0
    xd = x.to(DEV)
    rows_out = g.Dn if compact else g.N
    outs = []
    for name, src, extra in (('tmpnn_segsum_fwd', xd, (0, int(compact))), ('tmpnn_segsum_fwd_live', xd, (int(compact), lim)),
                             ('tmpnn_segsum_fwd_live', None, (int(compact), lim)), ('tmpnn_segsum_fwd_live', xd, (int(compact), g.N))):
        if src is None:
            src = xd.clone()
            src[lim:] = float('nan')
        od = torch.full((rows_out, H + 8), 7.0, device=DEV)
        _lib.call(name, gd.cref(), src.data_ptr(), x.shape[1], od.data_ptr(), od.shape[1], H, *extra, st())
        outs.append(od.cpu())
    return float(not (torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])))


def gru_ref(x, h, wih, whh, bih, bhh):
    gi = F.linear(x, wih, bih)
    gh = F.linear(h, whh, bhh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h, (r, z, n, h_n + 0)


def check_gru(H, xmode, g, gd, rows_kind='edge'):
    """fwd + bwd_data + bwd_weights of one cell; returns dict of max errors."""
    torch.manual_seed(H + xmode)
    G = 2                                  # exercise the group stride
    ld = G * H
    hfull = torch.randn(g.N, ld)
    h = hfull[:, H:2 * H]
    if rows_kind == 'edge':
        rows = idx(g.edge_row)
        src, dst = idx(g.src), idx(g.dst)
    else:
        rows = idx(g.det_row)
        src = dst = None
    R = rows.numel()
    IN = 2 * H if xmode == 2 else H
    msg = torch.randn(R, IN)               # compact messages (xmode 0)
    if xmode == 1:
        x = h[src] - h[dst]
    elif xmode == 2:
        x = torch.cat([h[src], h[dst]], 1)
    else:
        x = msg
    x = x.clone().requires_grad_(True)
    hrow = h[rows].clone().requires_grad_(True)
    wih = (0.3 * torch.randn(3 * H, IN)).requires_grad_(True)
    whh = (0.3 * torch.randn(3 * H, H)).requires_grad_(True)
    bih = (0.3 * torch.randn(3 * H)).requires_grad_(True)
    bhh = (0.3 * torch.randn(3 * H)).requires_grad_(True)
    out, (r, z, n, hn) = gru_ref(x, hrow, wih, whh, bih, bhh)
    dout = torch.randn(R, H)
    out.backward(dout)

    d = lambda t: t.detach().to(DEV).contiguous()
    hD = d(hfull)
    outD = torch.zeros(g.N, ld, device=DEV)
    gates = torch.zeros(4, g.N, H, device=DEV)
    rowsD = rows.to(torch.int32).to(DEV)
    wih_t, whh_t = d(wih.t()), d(whh.t())
    msgD = d(msg)
    bihD, bhhD = d(bih), d(bhh)
    _lib.call('tmpnn_gru_fwd', rowsD.data_ptr(), R, xmode, gd.src.data_ptr() if xmode else None,
              gd.dst.data_ptr() if xmode else None, msgD.data_ptr() if xmode == 0 else None, IN, 1, IN,
              hD.data_ptr() + 4 * H, ld, H, wih_t.data_ptr(), whh_t.data_ptr(), bihD.data_ptr(), bhhD.data_ptr(),
              outD.data_ptr() + 4 * H, ld, gates.data_ptr(), g.N * H, None, None, 0, st())
    res = {}
    res['fwd'] = (outD.cpu()[rows, H:2 * H] - out.detach()).abs().max().item()
    gc = gates.cpu()
    res['gates'] = max((gc[i][rows] - t.detach()).abs().max().item() for i, t in enumerate((r, z, n, hn)))
    other = torch.ones(g.N, dtype=torch.bool)
    other[rows] = False
    res['untouched'] = outD.cpu()[other].abs().max().item() if other.any() else 0.0
    res['untouched'] = max(res['untouched'], outD.cpu()[:, :H].abs().max().item())
    nparts = _lib.load().tmpnn_gru_fwd_head_parts(H, IN, xmode)
    if nparts > 0:
        wv_h = torch.randn(H)
        wv_hD = d(wv_h)
        parts = torch.zeros(nparts, g.N, device=DEV)
        outH = torch.zeros(g.N, ld, device=DEV)
        _lib.call('tmpnn_gru_fwd', rowsD.data_ptr(), R, xmode, gd.src.data_ptr() if xmode else None,
                  gd.dst.data_ptr() if xmode else None, msgD.data_ptr() if xmode == 0 else None, IN, 1, IN,
                  hD.data_ptr() + 4 * H, ld, H, wih_t.data_ptr(), whh_t.data_ptr(), bihD.data_ptr(), bhhD.data_ptr(),
                  outH.data_ptr() + 4 * H, ld, None, 0, wv_hD.data_ptr(), parts.data_ptr(), g.N, st())
        res['fused_head'] = (parts.cpu().sum(0)[rows] - out.detach() @ wv_h).abs().max().item()
    if xmode == 1 and H <= 64:
        # pre-projected variant (xmode 3): P = h[dets] @ W_ih^T, then gi = P[src] - P[dst]
        dr = idx(g.det_row)
        proj = torch.zeros(g.Dn, 3 * H + 4, device=DEV)
        _lib.call('tmpnn_rows_linear', gd.det_row.data_ptr(), g.Dn, hD.data_ptr() + 4 * H, ld, H, wih_t.data_ptr(), 3 * H,
                  proj.data_ptr(), 3 * H + 4, st())
        res['proj'] = (proj.cpu()[:, :3 * H] - h[dr] @ wih.detach().t()).abs().max().item()
        out3 = torch.zeros(g.N, ld, device=DEV)
        gates3 = torch.zeros(4, g.N, H, device=DEV)
        _lib.call('tmpnn_gru_fwd', rowsD.data_ptr(), R, 3, gd.src_pos.data_ptr(), gd.dst_pos.data_ptr(), proj.data_ptr(),
                  3 * H + 4, 0, IN, hD.data_ptr() + 4 * H, ld, H, None, whh_t.data_ptr(), bihD.data_ptr(), bhhD.data_ptr(),
                  out3.data_ptr() + 4 * H, ld, gates3.data_ptr(), g.N * H, None, None, 0, st())
        res['fwd_proj'] = (out3.cpu()[rows, H:2 * H] - out.detach()).abs().max().item()
        g3 = gates3.cpu()
        res['gates_proj'] = max((g3[i][rows] - t.detach()).abs().max().item() for i, t in enumerate((r, z, n, hn)))

    # backward, data
    doutfull = torch.zeros(g.N, ld)
    doutfull[rows, H:2 * H] = dout
    doutD = d(doutfull)
    dmsg = torch.zeros(g.N, IN + 4, device=DEV)
    dh = torch.zeros(g.N, ld, device=DEV)
    wihD, whhD = d(wih), d(whh)
    _lib.call('tmpnn_gru_bwd_data', rowsD.data_ptr(), R, IN, hD.data_ptr() + 4 * H, ld, H, wihD.data_ptr(),
              whhD.data_ptr(), gates.data_ptr(), g.N * H, doutD.data_ptr() + 4 * H, ld, None, None,
              dmsg.data_ptr(), IN + 4, dh.data_ptr() + 4 * H, ld, None, None, None, 0, st())
    res['dx'] = (dmsg.cpu()[rows, :IN] - x.grad).abs().max().item()
    res['dh'] = (dh.cpu()[rows, H:2 * H] - hrow.grad).abs().max().item()
    # same gradient split as d_hout' + dy * w_head (the folded output-head term), plus the fused
    # aggregation adjoint on edge rows
    dyv = torch.randn(g.N)
    wv = torch.randn(H)
    split = doutfull.clone()
    split[:, H:2 * H] -= dyv[:, None] * wv[None, :]
    splitD, dyD, wD = d(split), d(dyv), d(wv)
    addm = torch.randn(g.N, H + 8)
    addD = d(addm)
    dmsg2 = torch.zeros(g.N, IN + 4, device=DEV)
    dh2 = torch.zeros(g.N, ld, device=DEV)
    fuse = rows_kind == 'edge'
    _lib.call('tmpnn_gru_bwd_data', rowsD.data_ptr(), R, IN, hD.data_ptr() + 4 * H, ld, H, wihD.data_ptr(),
              whhD.data_ptr(), gates.data_ptr(), g.N * H, splitD.data_ptr() + 4 * H, ld, dyD.data_ptr(), wD.data_ptr(),
              dmsg2.data_ptr(), IN + 4, dh2.data_ptr() + 4 * H, ld,
              gd.src.data_ptr() if fuse else None, gd.dst.data_ptr() if fuse else None,
              addD.data_ptr() if fuse else None, H + 8, st())
    exp_dh = hrow.grad.clone()
    if fuse:
        exp_dh = exp_dh + addm[idx(g.src), :H] - addm[idx(g.dst), :H]
    res['dx_split'] = (dmsg2.cpu()[rows, :IN] - x.grad).abs().max().item()
    res['dh_split_fused'] = (dh2.cpu()[rows, H:2 * H] - exp_dh).abs().max().item()
    # head term only (d_hout = NULL)
    only = torch.zeros(g.N, ld)
    only[:, H:2 * H] = dyv[:, None] * wv[None, :]
    onlyD = d(only)
    dm_a, dh_a = torch.zeros(g.N, IN + 4, device=DEV), torch.zeros(g.N, ld, device=DEV)
    dm_b, dh_b = torch.zeros(g.N, IN + 4, device=DEV), torch.zeros(g.N, ld, device=DEV)
    _lib.call('tmpnn_gru_bwd_data', rowsD.data_ptr(), R, IN, hD.data_ptr() + 4 * H, ld, H, wihD.data_ptr(),
              whhD.data_ptr(), gates.data_ptr(), g.N * H, onlyD.data_ptr() + 4 * H, ld, None, None,
              dm_a.data_ptr(), IN + 4, dh_a.data_ptr() + 4 * H, ld, None, None, None, 0, st())
    _lib.call('tmpnn_gru_bwd_data', rowsD.data_ptr(), R, IN, hD.data_ptr() + 4 * H, ld, H, wihD.data_ptr(),
              whhD.data_ptr(), gates.data_ptr(), g.N * H, None, 0, dyD.data_ptr(), wD.data_ptr(),
              dm_b.data_ptr(), IN + 4, dh_b.data_ptr() + 4 * H, ld, None, None, None, 0, st())
    res['dy_only'] = max((dm_a - dm_b).abs().max().item(), (dh_a - dh_b).abs().max().item())
    # backward, weights (accumulate into non-zero buffers)
    base = 0.5
    dW_ih = torch.full((3 * H, IN), base, device=DEV)
    dW_hh = torch.full((3 * H, H), base, device=DEV)
    db_ih = torch.full((3 * H,), base, device=DEV)
    db_hh = torch.full((3 * H,), base, device=DEV)
    wsb = _lib.load().tmpnn_gru_bwd_weights_ws(R, IN, H)
    ws = torch.empty(wsb // 4 + 1, device=DEV)
    _lib.call('tmpnn_gru_bwd_weights', rowsD.data_ptr(), R, xmode, gd.src.data_ptr() if xmode else None,
              gd.dst.data_ptr() if xmode else None, msgD.data_ptr() if xmode == 0 else None, IN, 1, IN,
              hD.data_ptr() + 4 * H, ld, H, gates.data_ptr(), g.N * H, splitD.data_ptr() + 4 * H, ld,
              dyD.data_ptr(), wD.data_ptr(),
              dW_ih.data_ptr(), dW_hh.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(), ws.data_ptr(), wsb, st())
    if _lib.load().tmpnn_gru_bwd_fused_available(H, IN, xmode):
        # fused data + weights backward, head term folded, row-F adjoint fused on edge rows (with compact messages,
        # xmode 0, the fused adjoint exists in TMPNN_KEEP_VARIANTS builds only: the default library is checked without)
        fuse_k = fuse and (xmode != 0 or os.environ.get('TMPNN_TEST_VARIANTS', '0') == '1')
        f_dm = torch.zeros(g.N, IN + 4, device=DEV)
        f_dh = torch.zeros(g.N, ld, device=DEV)
        fW_ih = torch.full((3 * H, IN), base, device=DEV); fW_hh = torch.full((3 * H, H), base, device=DEV)
        fb_ih = torch.full((3 * H,), base, device=DEV); fb_hh = torch.full((3 * H,), base, device=DEV)
        fwsb = _lib.load().tmpnn_gru_bwd_fused_ws(R, IN, H)
        fws = torch.empty(fwsb // 4 + 1, device=DEV)
        _lib.call('tmpnn_gru_bwd_fused', rowsD.data_ptr(), R, xmode, gd.src.data_ptr() if xmode else None,
                  gd.dst.data_ptr() if xmode else None, msgD.data_ptr() if xmode == 0 else None, IN, 1, IN,
                  hD.data_ptr() + 4 * H, ld, H, wihD.data_ptr(), whhD.data_ptr(), gates.data_ptr(), g.N * H,
                  splitD.data_ptr() + 4 * H, ld, dyD.data_ptr(), wD.data_ptr(), f_dm.data_ptr(), IN + 4,
                  f_dh.data_ptr() + 4 * H, ld, gd.src.data_ptr() if fuse_k else None, gd.dst.data_ptr() if fuse_k else None,
                  addD.data_ptr() if fuse_k else None, H + 8, fW_ih.data_ptr(), fW_hh.data_ptr(), fb_ih.data_ptr(),
                  fb_hh.data_ptr(), fws.data_ptr(), fwsb, st())
        scf = max(1.0, wih.grad.abs().max().item())
        res['fused_dx'] = (f_dm.cpu()[rows, :IN] - x.grad).abs().max().item()
        res['fused_dh'] = (f_dh.cpu()[rows, H:2 * H] - (exp_dh if fuse_k else hrow.grad)).abs().max().item()
        res['fused_dW_ih'] = (fW_ih.cpu() - base - wih.grad).abs().max().item() / scf
        res['fused_dW_hh'] = (fW_hh.cpu() - base - whh.grad).abs().max().item() / scf
        res['fused_db_ih'] = (fb_ih.cpu() - base - bih.grad).abs().max().item() / scf
        res['fused_db_hh'] = (fb_hh.cpu() - base - bhh.grad).abs().max().item() / scf
    sc = max(1.0, wih.grad.abs().max().item())
    res['dW_ih'] = (dW_ih.cpu() - base - wih.grad).abs().max().item() / sc
    res['dW_hh'] = (dW_hh.cpu() - base - whh.grad).abs().max().item() / sc
    res['db_ih'] = (db_ih.cpu() - base - bih.grad).abs().max().item() / sc
    res['db_hh'] = (db_hh.cpu() - base - bhh.grad).abs().max().item() / sc
    return res


def check_heads(C, g, gd):
    torch.manual_seed(C)
    N = g.N
    h = torch.randn(N, C + 4)
    hh = h[:, :C].clone().requires_grad_(True)
    wn = torch.randn(1, C, requires_grad=True)
    we = torch.randn(1, C, requires_grad=True)
    bn = torch.randn(1, requires_grad=True)
    be = torch.randn(1, requires_grad=True)
    ie = g.is_edge.cpu().bool()
    y = torch.where(ie[:, None], F.linear(hh, we, be), F.linear(hh, wn, bn))
    s = torch.sigmoid(y)
    dl, ds = torch.randn(N, 1), torch.randn(N, 1)
    (y * dl + s * ds).sum().backward()
    d = lambda t: t.detach().to(DEV).contiguous()
    hD = d(h)
    logits = torch.empty(N, 1, device=DEV)
    scores = torch.empty(N, 1, device=DEV)
    wnD, weD, bnD, beD = d(wn), d(we), d(bn), d(be)     # keep alive: freed blocks are reused at once
    _lib.call('tmpnn_heads_fwd', hD.data_ptr(), C + 4, C, N, gd.is_edge.data_ptr(), wnD.data_ptr(), bnD.data_ptr(),
              weD.data_ptr(), beD.data_ptr(), logits.data_ptr(), scores.data_ptr(), st())
    res = {'logits': (logits.cpu() - y.detach()).abs().max().item(),
           'scores': (scores.cpu() - s.detach()).abs().max().item()}
    pre = torch.randn(N, C)
    dh = d(pre)
    g_wn, g_we = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    g_bn, g_be = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    wsb = _lib.load().tmpnn_heads_bwd_ws(N, C)
    ws = torch.empty(wsb // 4 + 1, device=DEV)
    dlD, dsD = d(dl), d(ds)
    dyo = torch.zeros(N, device=DEV)
    _lib.call('tmpnn_heads_bwd', hD.data_ptr(), C + 4, C, N, gd.is_edge.data_ptr(), wnD.data_ptr(), weD.data_ptr(),
              scores.data_ptr(), dlD.data_ptr(), dsD.data_ptr(), dyo.data_ptr(), dh.data_ptr(), C, 1,
              g_wn.data_ptr(), g_bn.data_ptr(), g_we.data_ptr(), g_be.data_ptr(), ws.data_ptr(), wsb, st())
    sc = max(1.0, wn.grad.abs().max().item(), we.grad.abs().max().item())
    res['dh'] = (dh.cpu() - pre - hh.grad).abs().max().item()
    res['dy'] = (dyo.cpu()[:, None] - (dl + ds * s.detach() * (1 - s.detach()))).abs().max().item()
    res['dw'] = max((g_wn.cpu() - wn.grad[0]).abs().max().item(), (g_we.cpu() - we.grad[0]).abs().max().item()) / sc
    res['db'] = max((g_bn.cpu() - bn.grad).abs().max().item(), (g_be.cpu() - be.grad).abs().max().item()) / sc
    return res


def check_input_bn(H, F_, training, S=4, fused=False, maxn=40, dx=True):
    """fused=False: the staged launches of tmpnn_input_bn_*; fused=True: tmpnn_input_tf_* (one launch per direction), with
    enough segments that several workgroups and several chunks per workgroup take part."""
    torch.manual_seed(H + F_)
    # S segments, each with nd_s det rows and some zero rows
    if S == 4:
        nds = [3, 1, 5, 2]
        cnts = [n + z for n, z in zip(nds, [6, 4, 0, 9])]
    else:
        rs = np.random.RandomState(S)
        nds = [int(v) for v in rs.randint(1, maxn, S)]
        cnts = [n + int(z) for n, z in zip(nds, rs.randint(0, 30, S))]
        cnts = [max(c, 2) for c in cnts]
    nd = sum(nds)
    Ft = F_ + 3                                   # group columns embedded in a wider feature row
    xdet = torch.randn(nd, Ft)
    cfg = orc.OracleConfig('2d', F_ - 5, H, 0, 'diff')
    p = orc.random_params(cfg, seed=H)
    p = {k: v for k, v in p.items() if k.startswith('input_transforms.0.')}
    pg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
          for k, v in p.items()}
    # oracle on the full row set: per segment [zeros ..., dets ...]
    xs, seg, is_det = [], [], []
    o = 0
    for s in range(S):
        nz = cnts[s] - nds[s]
        xs.append(torch.zeros(nz, F_))
        xs.append(xdet[o:o + nds[s], 1:1 + F_])
        seg += [s] * cnts[s]
        is_det += [False] * nz + [True] * nds[s]
        o += nds[s]
    xfull = torch.cat(xs).requires_grad_(True)
    seg = torch.tensor(seg)
    is_det = torch.tensor(is_det)
    out = orc._input_transform(pg, 0, xfull, seg, S, training, True)
    dsel = torch.randn(nd, H)
    (out[is_det] * dsel).sum().backward()

    d = lambda t: t.detach().to(DEV).contiguous()
    P = {k: d(v) for k, v in p.items()}
    t = 'input_transforms.0.'
    seg_ptr = torch.tensor(np.concatenate([[0], np.cumsum(nds)]), dtype=torch.int32, device=DEV)
    seg_cnt = torch.tensor(cnts, dtype=torch.int32, device=DEV)
    xD = d(xdet)
    y_save = torch.empty(nd, H, device=DEV)
    SS = S if training else 1
    mean, rstd = torch.empty(SS, H, device=DEV), torch.empty(SS, H, device=DEV)
    ws_a = torch.empty(nd, H, device=DEV)
    Nrows, ld = nd + 5, 2 * H
    out_row = torch.tensor(np.random.RandomState(0).permutation(Nrows)[:nd], dtype=torch.int32, device=DEV)
    h_new = torch.zeros(Nrows, ld, device=DEV)
    seg_of_det = torch.repeat_interleave(torch.arange(S, dtype=torch.int32), torch.tensor(nds)).to(DEV) if (H != 32 or fused) else None
    # the one-launch form also takes x through a row list (the det rows of the caller's [n, F] x): a shuffled, larger copy
    xr_idx = torch.tensor(np.random.RandomState(1).permutation(nd + 7)[:nd], dtype=torch.int64, device=DEV)
    xS = torch.full((nd + 7, xD.shape[1]), float('nan'), device=DEV)
    xS[xr_idx] = xD
    if fused:
        _lib.call('tmpnn_input_tf_fwd', xS.data_ptr() + 4, xr_idx.data_ptr(), Ft, F_, nd, seg_ptr.data_ptr(), seg_cnt.data_ptr(), _lib.ptr(seg_of_det), S,
                  max(nds), H, int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(), P[t + '1.weight'].data_ptr(),
                  P[t + '1.bias'].data_ptr(), P[t + '1.running_mean'].data_ptr(), P[t + '1.running_var'].data_ptr(),
                  P[t + '3.weight'].data_ptr(), P[t + '3.bias'].data_ptr(), y_save.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), out_row.data_ptr(), h_new.data_ptr() + 4 * H, ld, st())
    else:
        _lib.call('tmpnn_input_bn_fwd', xD.data_ptr() + 4, Ft, F_, nd, seg_ptr.data_ptr(), seg_cnt.data_ptr(), _lib.ptr(seg_of_det), S, H,
                  int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(), P[t + '1.weight'].data_ptr(),
                  P[t + '1.bias'].data_ptr(), P[t + '1.running_mean'].data_ptr(), P[t + '1.running_var'].data_ptr(),
                  P[t + '3.weight'].data_ptr(), P[t + '3.bias'].data_ptr(), y_save.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), ws_a.data_ptr(), out_row.data_ptr(), h_new.data_ptr() + 4 * H, ld, st())
    res = {}
    got = h_new.cpu()[out_row.long().cpu(), H:2 * H]
    res['fwd'] = (got - out[is_det].detach()).abs().max().item()
    res['running_mean'] = (P[t + '1.running_mean'].cpu() - pg[t + '1.running_mean']).abs().max().item()
    res['running_var'] = (P[t + '1.running_var'].cpu() - pg[t + '1.running_var']).abs().max().item()
    # backward
    d_h = torch.zeros(Nrows, ld)
    d_h[out_row.long().cpu(), H:2 * H] = dsel
    d_hD = d(d_h)
    grads = {k: torch.zeros_like(P[t + k]) for k in ('0.weight', '0.bias', '1.weight', '1.bias', '3.weight', '3.bias')}
    d_xdet = torch.zeros(nd, Ft, device=DEV)
    d_xzero = torch.zeros(S, F_, device=DEV)
    if fused:
        wsb = int(_lib.load().tmpnn_input_tf_bwd_ws(nd, S, H, F_, int(training)))
        ws = torch.empty(wsb // 4 + 1, device=DEV)
        _lib.call('tmpnn_input_tf_bwd', xS.data_ptr() + 4, xr_idx.data_ptr(), Ft, F_, nd, seg_ptr.data_ptr(), seg_cnt.data_ptr(), _lib.ptr(seg_of_det), S,
                  max(nds), H, int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(), P[t + '1.weight'].data_ptr(),
                  P[t + '1.bias'].data_ptr(), P[t + '3.weight'].data_ptr(), y_save.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), out_row.data_ptr(), d_hD.data_ptr() + 4 * H, ld, (d_xdet.data_ptr() + 4) if dx else None, Ft,
                  d_xzero.data_ptr() if dx else None, grads['0.weight'].data_ptr(), grads['0.bias'].data_ptr(),
                  grads['1.weight'].data_ptr(), grads['1.bias'].data_ptr(), grads['3.weight'].data_ptr(),
                  grads['3.bias'].data_ptr(), ws.data_ptr(), wsb, st())
    else:
        wsn = _lib.load().tmpnn_input_bn_bwd_ws(nd, S, H, F_)
        ws = torch.empty(wsn + 1, device=DEV)
        _lib.call('tmpnn_input_bn_bwd', xD.data_ptr() + 4, Ft, F_, nd, seg_ptr.data_ptr(), seg_cnt.data_ptr(), _lib.ptr(seg_of_det), S, H,
                  int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(), P[t + '1.weight'].data_ptr(),
                  P[t + '1.bias'].data_ptr(), P[t + '3.weight'].data_ptr(), y_save.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), out_row.data_ptr(), d_hD.data_ptr() + 4 * H, ld, d_xdet.data_ptr() + 4, Ft,
                  d_xzero.data_ptr(), grads['0.weight'].data_ptr(), grads['0.bias'].data_ptr(),
                  grads['1.weight'].data_ptr(), grads['1.bias'].data_ptr(), grads['3.weight'].data_ptr(),
                  grads['3.bias'].data_ptr(), ws.data_ptr(), wsn + 1, st())
    gs = max(1.0, max(pg[t + k].grad.abs().max().item() for k in grads))
    for k in grads:
        res['d' + k] = (grads[k].cpu() - pg[t + k].grad).abs().max().item() / gs
    if not dx:               # (no gradient with respect to x asked for: the wave-owned form of the one-launch transform)
        return res
    res['d_xdet'] = (d_xdet.cpu()[:, 1:1 + F_] - xfull.grad[is_det]).abs().max().item() / gs
    gz = xfull.grad[~is_det]
    segz = seg[~is_det]
    res['d_xzero'] = (d_xzero.cpu()[segz] - gz).abs().max().item() / gs if gz.numel() else 0.0
    return res


def check_attention(H, K, training, g, gd):
    torch.manual_seed(H * 10 + K)
    G = 2
    ld = G * H
    hfull = torch.randn(g.N, ld)
    h = hfull[:, H:2 * H].clone().requires_grad_(True)
    og = orc.OracleGraph(g.N, g.is_edge.cpu().numpy().astype(bool), idx(g.src).numpy(), idx(g.dst).numpy(),
                         idx(g.edge_row).numpy(), idx(g.det_row).numpy())
    p = {}
    for k in range(K):
        p[f'gat.{k}.W_att'] = (0.4 * torch.randn(H, H)).requires_grad_(True)
        p[f'gat.{k}.a'] = (0.4 * torch.randn(H, 1)).requires_grad_(True)
    keep = (torch.rand(K, g.E, 2) > 0.5).to(torch.uint8) if training else None
    es = 0
    alphas = []
    for k in range(K):
        e_k, a_k = orc._attention(p, '', k, h, og, None if keep is None else keep[k])
        es = es + e_k
        alphas.append(a_k)
    es = es / K
    dr = idx(g.det_row)
    d_es = torch.randn(g.Dn, H)
    (es[dr] * d_es).sum().backward()

    d = lambda t: t.detach().to(DEV).contiguous()
    e_of_p, ep = gd.inc_edge_endpoint()
    keepD = None
    if keep is not None:                   # [K, 2E] in CSR order -> one byte per position, bit k = head k keeps it
        kk = d(keep)[:, e_of_p, ep]
        keepD = sum(kk[k] << k for k in range(K)).to(torch.uint8).contiguous()
    W = d(torch.cat([p[f'gat.{k}.W_att'] for k in range(K)], 1))          # [H][K*H]: the heads side by side
    a = d(torch.stack([p[f'gat.{k}.a'].reshape(-1) for k in range(K)]))
    hD = d(hfull)
    ws_ha = torch.empty(g.Dn, K * H, device=DEV)
    score = torch.full((2 * g.E, K), float('nan'), device=DEV)
    erec, inc_other = gd.att_index()
    stats = torch.empty(g.Dn, K, 2, device=DEV)
    esk = torch.empty(K, g.Dn, H, device=DEV)
    alpha = torch.empty(K, 2 * g.E, device=DEV)
    out = torch.empty(g.Dn, H, device=DEV)
    _lib.call('tmpnn_att_fwd', gd.cref(), erec.data_ptr(), hD.data_ptr() + 4 * H, ld, H, K,
              W.data_ptr(), a.data_ptr(), _lib.ptr(keepD), 0.5, ws_ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
              esk.data_ptr(), alpha.data_ptr(), out.data_ptr(), H, st())
    res = {'es': (out.cpu() - es[dr].detach()).abs().max().item()}
    # the one-launch index arrays (tmpnn_att_index) against the torch index ops on the host copy of the graph
    erec_h, other_h = g.att_index()
    res['att_index bits'] = float(not (torch.equal(erec.cpu()[:g.E, :7], erec_h[:g.E, :7]) and torch.equal(inc_other.cpu()[:2 * g.E], other_h[:2 * g.E])))
    res['es = sum of the per-head aggregates'] = (esk.sum(0) - out).abs().max().item()
    al = torch.zeros(K, g.E, 2)
    al[:, e_of_p.cpu(), ep.cpu()] = alpha.cpu()
    res['alpha'] = max((al[k] - alphas[k].detach()).abs().max().item() for k in range(K))
    # backward
    dmsg = torch.zeros(g.N, H + 4)
    dmsg[dr, :H] = d_es
    dmsgD = d(dmsg)
    pre = torch.randn(g.N, ld)
    d_h = d(pre)
    dW, da = torch.zeros(K, H, H, device=DEV), torch.zeros_like(a)
    wsn = _lib.load().tmpnn_att_bwd_ws(g.E, g.Dn, H, K)
    ws = torch.full((wsn + 4,), float('nan'), device=DEV)     # (every entry that is read is written first)
    args = (gd.cref(), erec.data_ptr(), inc_other.data_ptr(), hD.data_ptr() + 4 * H, ld, H, K,
            W.data_ptr(), a.data_ptr(), _lib.ptr(keepD), 0.5, ws_ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
            esk.data_ptr(), dmsgD.data_ptr(), H + 4, ws.data_ptr(), wsn)
    _lib.call('tmpnn_att_bwd', *args, d_h.data_ptr() + 4 * H, ld, dW.data_ptr(), da.data_ptr(), st())
    gs = max(1.0, h.grad.abs().max().item())
    res['d_h'] = ((d_h.cpu() - pre)[:, H:2 * H] - h.grad).abs().max().item() / gs
    res['d_h_other_group'] = (d_h.cpu() - pre)[:, :H].abs().max().item()
    res['dW'] = max((dW.cpu()[k] - p[f'gat.{k}.W_att'].grad).abs().max().item() for k in range(K)) / max(
        1.0, max(p[f'gat.{k}.W_att'].grad.abs().max().item() for k in range(K)))
    res['da'] = max((da.cpu()[k] - p[f'gat.{k}.a'].grad[:, 0]).abs().max().item() for k in range(K)) / max(
        1.0, max(p[f'gat.{k}.a'].grad.abs().max().item() for k in range(K)))
    # the same backward with one gradient pointer per head (accumulating into buffers that already hold something, as
    # p.grad does): bit-identical sums, and a second run of the stacked entry reproduces the first bit for bit
    import ctypes
    base_W = [torch.randn(H, H, device=DEV) for _ in range(K)]
    base_a = [torch.randn(H, 1, device=DEV) for _ in range(K)]
    gW, ga = [t.clone() for t in base_W], [t.clone() for t in base_a]
    d_h2 = d(pre)
    ws.fill_(float('nan'))
    pW = (ctypes.c_void_p * K)(*[t.data_ptr() for t in gW])
    pa = (ctypes.c_void_p * K)(*[t.data_ptr() for t in ga])
    _lib.call('tmpnn_att_bwd_heads', *args, d_h2.data_ptr() + 4 * H, ld, ctypes.cast(pW, ctypes.c_void_p),
              ctypes.cast(pa, ctypes.c_void_p), st())
    res['heads entry: d_h bits'] = float(not torch.equal(d_h2, d_h))
    res['heads entry: dW'] = max(((gW[k] - base_W[k]) - dW[k]).abs().max().item() for k in range(K)) / max(1.0, dW.abs().max().item())
    res['heads entry: da'] = max(((ga[k] - base_a[k])[:, 0] - da[k]).abs().max().item() for k in range(K)) / max(1.0, da.abs().max().item())
    d_h3 = d(pre)
    dW3, da3 = torch.zeros_like(dW), torch.zeros_like(da)
    _lib.call('tmpnn_att_bwd', *args, d_h3.data_ptr() + 4 * H, ld, dW3.data_ptr(), da3.data_ptr(), st())
    res['second run: bits'] = float(not (torch.equal(d_h3, d_h) and torch.equal(dW3, dW) and torch.equal(da3, da)))
    return res


def check_wide(H, g, gd):
    """Wide edge cell (csrc/wide.hip), diff message: forward through the projected det rows, then BOTH backward forms --
    per-edge products (tmpnn_wide_gru_bwd_data + _weights + the message adjoint's segment sum) and the det-side form
    (tmpnn_wide_gru_bwd_diff) -- against autograd on the CPU."""
    torch.manual_seed(1000 + H)
    ld = 2 * H
    hfull = torch.randn(g.N, ld)
    hleaf = hfull[:, H:2 * H].clone().requires_grad_(True)
    rows, dets = idx(g.edge_row), idx(g.det_row)
    src, dst = idx(g.src), idx(g.dst)
    R, Dn = rows.numel(), dets.numel()
    sc = 1.0 / H ** 0.5
    wih = (sc * torch.randn(3 * H, H)).requires_grad_(True)
    whh = (sc * torch.randn(3 * H, H)).requires_grad_(True)
    bih = (0.3 * torch.randn(3 * H)).requires_grad_(True)
    bhh = (0.3 * torch.randn(3 * H)).requires_grad_(True)
    out, (r, z, n, hn) = gru_ref(hleaf[src] - hleaf[dst], hleaf[rows], wih, whh, bih, bhh)
    dout = torch.randn(R, H)
    out.backward(dout)
    d = lambda t: t.detach().to(DEV).contiguous()
    hD = d(hfull)
    lib = _lib.load()
    prep = torch.empty(int(lib.tmpnn_wide_prep_bytes(H, H)), dtype=torch.uint8, device=DEV)
    wihD, whhD, bihD, bhhD = d(wih), d(whh), d(bih), d(bhh)
    _lib.call('tmpnn_wide_prepare', wihD.data_ptr(), whhD.data_ptr(), H, H, prep.data_ptr(), st())
    P = torch.empty(Dn, 3 * H, device=DEV)
    outD = torch.zeros(g.N, ld, device=DEV)
    gates = torch.zeros(4, g.N, H, device=DEV)
    _lib.call('tmpnn_wide_gru_fwd', prep.data_ptr(), gd.det_row.data_ptr(), Dn, gd.edge_row.data_ptr(), R,
              gd.src_pos.data_ptr(), gd.dst_pos.data_ptr(), hD.data_ptr() + 4 * H, ld, H, bihD.data_ptr(), bhhD.data_ptr(),
              P.data_ptr(), outD.data_ptr() + 4 * H, ld, gates.data_ptr(), g.N * H, st())
    res = {'fwd': (outD.cpu()[rows, H:2 * H] - out.detach()).abs().max().item()}
    gc = gates.cpu()
    res['gates'] = max((gc[i][rows] - t.detach()).abs().max().item() for i, t in enumerate((r, z, n, hn)))
    doutD = torch.zeros(g.N, ld, device=DEV)
    doutD[gd.edge_row.long(), H:2 * H] = dout.to(DEV)
    pre = torch.randn(g.N, H)                  # what the det rows of d_h hold before the call (the node cell's gradient)
    ref_dh = hleaf.grad.clone()
    ref_dh[dets] += pre[dets]
    gsc = lambda t: max(1.0, t.abs().max().item())

    def fresh():
        dh = torch.full((g.N, ld), 7.0, device=DEV)
        dh[gd.det_row.long(), H:2 * H] = pre[dets].to(DEV)
        return dh, [torch.zeros(3 * H, H, device=DEV), torch.zeros(3 * H, H, device=DEV),
                    torch.zeros(3 * H, device=DEV), torch.zeros(3 * H, device=DEV)]

    def compare(tag, dh, gr):
        res[tag + ' d_h'] = (dh.cpu()[:, H:2 * H] - ref_dh).abs().max().item() / gsc(ref_dh)
        res[tag + ' untouched'] = (dh.cpu()[:, :H] - 7.0).abs().max().item()
        for nm, a, b in zip(('dW_ih', 'dW_hh', 'db_ih', 'db_hh'), gr, (wih, whh, bih, bhh)):
            res[f'{tag} {nm}'] = (a.cpu() - b.grad).abs().max().item() / gsc(b.grad)

    # det-side form
    dh, gr = fresh()
    wsb = int(lib.tmpnn_wide_gru_bwd_diff_ws(g.N, R, Dn, H))
    ws = torch.empty(wsb // 4 + 1, device=DEV)
    _lib.call('tmpnn_wide_gru_bwd_diff', prep.data_ptr(), gd.cref(), hD.data_ptr() + 4 * H, ld, H, gates.data_ptr(), g.N * H,
              doutD.data_ptr() + 4 * H, ld, None, None, dh.data_ptr() + 4 * H, ld, gr[0].data_ptr(), gr[1].data_ptr(),
              gr[2].data_ptr(), gr[3].data_ptr(), ws.data_ptr(), wsb, st())
    compare('det-side', dh, gr)
    # the same call with its det-side branch on a second stream, and with the adjoint of row F (add_msg[src] - add_msg[dst]
    # into the edge rows of d_h) taken in the epilogue of the E-row product: bit-identical to the plain call (+ the gather)
    addm = torch.randn(g.N, H + 8, device=DEV)
    _lib.call('tmpnn_gather_diff_fwd', gd.cref(), addm.data_ptr(), H + 8, dh.data_ptr() + 4 * H, ld, H, 1, st())
    aux = torch.cuda.Stream(DEV)
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(2)]       # the caller's fork / join events
    for e in evs:
        e.record()
    evf, evj = evs[0].cuda_event, evs[1].cuda_event
    has_variants = hasattr(lib, 'tmpnn_wide_gru_bwd_data')      # (a -DTMPNN_KEEP_VARIANTS build: the superseded forms too)
    forms = (('fused adjoint', False, True), ('fused adjoint + aux', True, True))
    if has_variants:
        forms = (('aux stream', True, False),) + forms
    for tag, use_aux, fused in forms:
        dh2, gr2 = fresh()
        args = (prep.data_ptr(), gd.cref(), hD.data_ptr() + 4 * H, ld, H, gates.data_ptr(), g.N * H,
                doutD.data_ptr() + 4 * H, ld, None, None, dh2.data_ptr() + 4 * H, ld, gr2[0].data_ptr(), gr2[1].data_ptr(),
                gr2[2].data_ptr(), gr2[3].data_ptr(), ws.data_ptr(), wsb)
        if fused:
            _lib.call('tmpnn_wide_gru_bwd_diff_fused', *args, addm.data_ptr(), H + 8, st(), aux.cuda_stream if use_aux else None,
                      evf if use_aux else None, evj if use_aux else None)
        else:
            _lib.call('tmpnn_wide_gru_bwd_diff_aux', *args, st(), aux.cuda_stream, evf, evj)
            _lib.call('tmpnn_gather_diff_fwd', gd.cref(), addm.data_ptr(), H + 8, dh2.data_ptr() + 4 * H, ld, H, 1, st())
        torch.cuda.synchronize()
        same = torch.equal(dh2, dh) and all(torch.equal(a, b) for a, b in zip(gr2, gr))
        res[f'det-side, {tag}: bits'] = 0.0 if same else 1.0
    # per-edge form (comparison builds only: the shipped library takes both W_ih products on the det side)
    if not has_variants:
        return res
    dh, gr = fresh()
    wsb = int(lib.tmpnn_wide_gru_bwd_data_ws(R, H))
    ws = torch.empty(wsb // 4 + 1, device=DEV)
    dmsg = torch.zeros(g.N, H, device=DEV)
    _lib.call('tmpnn_wide_gru_bwd_data', prep.data_ptr(), gd.edge_row.data_ptr(), R, hD.data_ptr() + 4 * H, ld, H,
              gates.data_ptr(), g.N * H, doutD.data_ptr() + 4 * H, ld, None, None, dmsg.data_ptr(), H,
              dh.data_ptr() + 4 * H, ld, ws.data_ptr(), wsb, st())
    _lib.call('tmpnn_gather_diff_bwd', gd.cref(), dmsg.data_ptr(), H, dh.data_ptr() + 4 * H, ld, H, 1, st())
    ws2b = int(lib.tmpnn_wide_gru_bwd_weights_ws(R, H))
    ws2 = torch.empty(ws2b // 4 + 1, device=DEV)
    _lib.call('tmpnn_wide_gru_bwd_weights', ws.data_ptr(), gd.edge_row.data_ptr(), R, gd.src.data_ptr(), gd.dst.data_ptr(),
              hD.data_ptr() + 4 * H, ld, H, gr[0].data_ptr(), gr[1].data_ptr(), gr[2].data_ptr(), gr[3].data_ptr(),
              ws2.data_ptr(), ws2b, st())
    compare('per-edge', dh, gr)
    return res


def check_wide_tiled(H, g):
    """tmpnn_wide_gru_fwd_tiled (edge tiles, projected det rows staged in LDS / read through the tile's det list) must
    reproduce tmpnn_wide_gru_fwd BIT FOR BIT: same products in the same order, only the gathers differ."""
    from trackmpnn_amd.graph import build_edge_tiles
    torch.manual_seed(2000 + H)
    gd = g.to(DEV)
    ld = 2 * H
    hD = torch.randn(g.N, ld, device=DEV)
    R, Dn = g.E, g.Dn
    sc = 1.0 / H ** 0.5
    wih, whh = sc * torch.randn(3 * H, H, device=DEV), sc * torch.randn(3 * H, H, device=DEV)
    bih, bhh = 0.3 * torch.randn(3 * H, device=DEV), 0.3 * torch.randn(3 * H, device=DEV)
    lib = _lib.load()
    prep = torch.empty(int(lib.tmpnn_wide_prep_bytes(H, H)), dtype=torch.uint8, device=DEV)
    _lib.call('tmpnn_wide_prepare', wih.data_ptr(), whh.data_ptr(), H, H, prep.data_ptr(), st())
    outs = []
    tiles = build_edge_tiles(gd, 128, stats=True)
    for tiled in (False, True):
        P = torch.empty(Dn, 3 * H, device=DEV)
        out = torch.full((g.N, ld), 3.0, device=DEV)
        gates = torch.full((4, g.N, H), 5.0, device=DEV)
        if tiled:
            _lib.call('tmpnn_wide_gru_fwd_tiled', prep.data_ptr(), gd.det_row.data_ptr(), Dn, tiles.cref(), R,
                      hD.data_ptr() + 4 * H, ld, H, bih.data_ptr(), bhh.data_ptr(), P.data_ptr(),
                      out.data_ptr() + 4 * H, ld, gates.data_ptr(), g.N * H, st())
        else:
            _lib.call('tmpnn_wide_gru_fwd', prep.data_ptr(), gd.det_row.data_ptr(), Dn, gd.edge_row.data_ptr(), R,
                      gd.src_pos.data_ptr(), gd.dst_pos.data_ptr(), hD.data_ptr() + 4 * H, ld, H, bih.data_ptr(),
                      bhh.data_ptr(), P.data_ptr(), out.data_ptr() + 4 * H, ld, gates.data_ptr(), g.N * H, st())
        torch.cuda.synchronize()
        outs.append((out.cpu(), gates.cpu()))
    cnt = (tiles.t_dptr[1:] - tiles.t_dptr[:-1])
    return {'h_out bits': float(not torch.equal(outs[0][0], outs[1][0])),
            'gates bits': float(not torch.equal(outs[0][1], outs[1][1])),
            'staged tiles': int((cnt <= 40).sum()), 'listed tiles': int((cnt > 40).sum())}


def check_fwd_tiles(H, g, order=None, rows=32):
    """tmpnn_gru_fwd_tiles (32-row edge tiles, projected det rows staged in LDS an item ahead; big tiles through their det
    list) must reproduce tmpnn_gru_fwd's xmode 3 BIT FOR BIT: h_out, the four gate planes and the fused head partials."""
    from trackmpnn_amd.graph import build_edge_tiles, edge_tiles
    torch.manual_seed(3000 + H)
    gd = g.to(DEV)
    ld = H + 32
    hD = torch.randn(g.N, ld, device=DEV)
    E, Dn = g.E, g.Dn
    sc = 1.0 / H ** 0.5
    wih_t, whh_t = sc * torch.randn(H, 3 * H, device=DEV), sc * torch.randn(H, 3 * H, device=DEV)
    bih, bhh = 0.3 * torch.randn(3 * H, device=DEV), 0.3 * torch.randn(3 * H, device=DEV)
    w_head = torch.randn(H, device=DEV)
    proj = torch.empty(Dn, 3 * H, device=DEV)
    _lib.call('tmpnn_rows_linear', gd.det_row.data_ptr(), Dn, hD.data_ptr(), ld, H, wih_t.data_ptr(), 3 * H,
              proj.data_ptr(), 3 * H, st())
    tiles = edge_tiles(gd, rows) if order is None else build_edge_tiles(gd, rows, 4, 8 if rows == 32 else 4, order=order)
    cw = H // 32
    outs = []
    for tiled in (False, True):
        out = torch.full((g.N, ld), 3.0, device=DEV)
        gates = torch.full((4, g.N, H), 5.0, device=DEV)
        parts = torch.full((cw, g.N), 7.0, device=DEV)
        if tiled:
            _lib.call('tmpnn_gru_fwd_tiles', tiles.cref(), E, proj.data_ptr(), 3 * H, hD.data_ptr(), ld, H, whh_t.data_ptr(),
                      bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), ld, gates.data_ptr(), g.N * H, w_head.data_ptr(),
                      parts.data_ptr(), g.N, st())
        else:
            _lib.call('tmpnn_gru_fwd', gd.edge_row.data_ptr(), E, 3, gd.src_pos.data_ptr(), gd.dst_pos.data_ptr(),
                      proj.data_ptr(), 3 * H, 0, H, hD.data_ptr(), ld, H, None, whh_t.data_ptr(), bih.data_ptr(), bhh.data_ptr(),
                      out.data_ptr(), ld, gates.data_ptr(), g.N * H, w_head.data_ptr(), parts.data_ptr(), g.N, st())
        torch.cuda.synchronize()
        outs.append((out.cpu(), gates.cpu(), parts.cpu()))
    cnt = (tiles.t_dptr[1:] - tiles.t_dptr[:-1])
    if rows == 16:
        # the 16-row form sums 32 k per MFMA instead of 16: same split products, another order -> equal to rounding.  Both
        # must sit within the bf16x6 bound of the fp64 result; rows outside the edge rows stay untouched bit for bit.
        er = g.edge_row.long()
        h64 = hD.double().cpu()[:, :H]
        gh = h64[er] @ whh_t.double().cpu()
        gi = proj.double().cpu()[g.src_pos.long()] - proj.double().cpu()[g.dst_pos.long()]
        b_i, b_h = bih.double().cpu(), bhh.double().cpu()
        r = torch.sigmoid(gi[:, :H] + b_i[:H] + gh[:, :H] + b_h[:H])
        z = torch.sigmoid(gi[:, H:2 * H] + b_i[H:2 * H] + gh[:, H:2 * H] + b_h[H:2 * H])
        hn = gh[:, 2 * H:] + b_h[2 * H:]
        nn_ = torch.tanh(gi[:, 2 * H:] + b_i[2 * H:] + r * hn)
        ho = (1 - z) * nn_ + z * h64[er]
        res = {}
        for name, (o, gt, pt) in (('ref32', outs[0]), ('t16', outs[1])):
            res[f'{name} h_out vs fp64'] = (o[er, :H].double() - ho).abs().max().item()
            res[f'{name} gates vs fp64'] = max((gt[i][er].double() - x).abs().max().item() for i, x in enumerate((r, z, nn_, hn)))
            res[f'{name} head vs fp64'] = (pt.double().sum(0)[er] - ho @ w_head.double().cpu()).abs().max().item()
        mask = torch.ones(g.N, dtype=torch.bool); mask[er] = False
        res['untouched rows bits'] = float(not (torch.equal(outs[0][0][mask], outs[1][0][mask]) and
                                                torch.equal(outs[0][1][:, mask], outs[1][1][:, mask]) and
                                                torch.equal(outs[0][0][:, H:], outs[1][0][:, H:])))
        res['t16 vs ref32 h_out'] = (outs[0][0] - outs[1][0]).abs().max().item()
        return res
    return {'h_out bits': float(not torch.equal(outs[0][0], outs[1][0])),
            'gates bits': float(not torch.equal(outs[0][1], outs[1][1])),
            'head bits': float(not torch.equal(outs[0][2], outs[1][2])),
            'staged tiles': int((cnt <= 24).sum()), 'listed tiles': int((cnt > 24).sum())}


def run_attention(rec, g):
    from trackmpnn_amd.graph import dense_static_graph
    # the small batch; a ragged batch; a dense window whose dets have 70 / 140 incidences (runs longer than one 64-position
    # chunk, which the pipeline does not prefetch); every head count the dispatch instantiates
    g_rag = make_graph(B=40, frames=7, mean=7, seed=3)
    g_den = dense_static_graph(3, 70)
    for H, K, tag, ga in ((64, 2, 'small batch', g), (32, 1, 'small batch', g), (128, 3, 'small batch', g), (64, 2, 'ragged batch', g_rag),
                          (64, 2, 'dense 3x70', g_den), (32, 4, 'dense 3x70', g_den), (256, 1, 'ragged batch', g_rag),
                          (64, 5, 'small batch', g), (32, 6, 'ragged batch', g_rag), (64, 7, 'small batch', g),
                          (64, 8, 'ragged batch', g_rag)):
        gad = ga.to(DEV)
        for training in (False, True):
            for k, v in check_attention(H, K, training, ga, gad).items():
                rec(f'attention H={H} K={K} {tag} train={training} {k}', v, 0.0 if k.endswith('bits') else 2e-4)


def run_all(report=print):
    """Yield (name, worst error, tolerance) for every stage/width combination."""
    g = make_graph()
    gd = g.to(DEV)
    results = []

    def rec(name, err, tol):
        results.append((name, err, tol))
        report(f'{"OK  " if err <= tol else "FAIL"} {name:48s} err={err:.3e} tol={tol:.0e}')

    for H in (32, 64, 128, 256):
        for concat in (False, True):
            for acc in (False, True):
                rec(f'gather H={H} concat={concat} acc={acc}', check_gather(H, concat, acc, g, gd), 1e-6)
        for mode in ('fwd', 'gdiff_bwd', 'gconcat_bwd'):
            for acc in (False, True):
                rec(f'segsum {mode} H={H} acc={acc}', check_segsum(H, mode, acc, False, g, gd), 2e-5)
        rec(f'segsum fwd compact H={H}', check_segsum(H, 'fwd', False, True, g, gd), 2e-5)
        for compact in (False, True):
            rec(f'segsum fwd live rows H={H} compact={compact} bit-equal', check_segsum_live(H, compact, g, gd), 0.0)
        for xmode, kind in ((1, 'edge'), (2, 'edge'), (0, 'det'), (0, 'edge')):
            r = check_gru(H, xmode, g, gd, kind)
            for k, v in r.items():
                rec(f'gru H={H} xmode={xmode} rows={kind} {k}', v, 0.0 if k == 'untouched' else 2e-4)
    for H in (128, 256):
        for k, v in check_wide(H, g, gd).items():
            rec(f'wide cell H={H} {k}', v, 0.0 if (k.endswith('untouched') or k.endswith('bits')) else 2e-4)
    from trackmpnn_amd.graph import dense_static_graph
    # E = 16 384 rows: eight weight-gradient slabs -- the slab -> XCD placement of k_wide_dw2 applies from a slab count
    # that is a multiple of eight -- and 128 full row tiles of the ring products (N = 256: the 256-column form)
    g8 = dense_static_graph(5, 64)
    g8d = g8.to(DEV)
    for H in (128, 256):
        for k, v in check_wide(H, g8, g8d).items():
            rec(f'wide cell H={H}, 5x64 window (8 slabs) {k}', v, 0.0 if (k.endswith('untouched') or k.endswith('bits')) else 2e-4)
    # odd sizes: fewer rows than one 16-row chunk of the weight gradient / one 128-row tile, a row count that ends a slab
    # mid-chunk, det counts that are no multiple of anything
    for T, D in ((2, 3), (3, 17), (2, 47)):
        go = dense_static_graph(T, D)
        god = go.to(DEV)
        for k, v in check_wide(128, go, god).items():
            rec(f'wide cell H=128, {T}x{D} window (E={go.E}) {k}', v, 0.0 if (k.endswith('untouched') or k.endswith('bits')) else 2e-4)
    for H in (128, 256, 384, 512):
        # a dense 3-block window (tiles staged in LDS), a larger ragged batch (both kinds: det lists above and below the staged
        # size, a partial last tile) and the small batch (one tile); H = 384 / 512: the periodic request stream of the pp
        # forward beyond the two widths C5 and the H = 128 models use
        for tag, gt in (('dense 4x40', dense_static_graph(4, 40)), ('ragged batch', make_graph(B=40, frames=7, mean=7, seed=3)),
                        ('small batch', g)):
            r = check_wide_tiled(H, gt)
            rec(f'wide tiled H={H} {tag} h_out bit-equal', r['h_out bits'], 0.0)
            rec(f'wide tiled H={H} {tag} gates bit-equal', r['gates bits'], 0.0)
            report(f'     tiles staged in LDS / read through their det list: {r["staged tiles"]} / {r["listed tiles"]}')
    import os
    for H in ((32, 64) if os.environ.get('TMPNN_SPLIT', '1')[:1] != '0' else ()):      # (the tiled forward is a bf16x6 kernel)
        for tag, gt, order in (('small batch', g, None), ('ragged batch', make_graph(B=40, frames=7, mean=7, seed=3), None),
                               ('dense 4x40, blocks', dense_static_graph(4, 40), 'blocks'),
                               ('dense 3x70, rows', dense_static_graph(3, 70), 'rows')):
            r = check_fwd_tiles(H, gt, order)
            for k in ('h_out bits', 'gates bits', 'head bits'):
                rec(f'fwd tiles H={H} {tag} {k[:-5]} bit-equal', r[k], 0.0)
            report(f'     tiles staged in LDS / read through their det list: {r["staged tiles"]} / {r["listed tiles"]}')
            # the 16-row form (TMPNN_FWD_TILE_ROWS=16): equal to rounding, both within the bf16x6 bound of the fp64 result
            r = check_fwd_tiles(H, gt, order, rows=16)
            rec(f'fwd tiles16 H={H} {tag} untouched rows', r['untouched rows bits'], 0.0)
            for k, v in r.items():
                if 'vs fp64' in k:
                    rec(f'fwd tiles16 H={H} {tag} {k}', v, 6e-6 if 'head' in k else 3e-6)
    for C in (32, 64, 96, 192, 256, 768):
        for k, v in check_heads(C, g, gd).items():
            rec(f'heads C={C} {k}', v, 2e-4)
    for H, F_ in ((64, 8), (32, 13), (128, 2), (64, 128)):
        for training in (True, False):
            for k, v in check_input_bn(H, F_, training).items():
                rec(f'input_bn H={H} F={F_} train={training} {k}', v, 2e-4)
    for H, F_, S in ((64, 8, 4), (64, 8, 150), (32, 13, 70), (64, 128, 40), (32, 2, 4)):
        for training in (True, False):
            for k, v in check_input_bn(H, F_, training, S=S, fused=True).items():
                rec(f'input_tf (one launch) H={H} F={F_} S={S} train={training} {k}', v, 2e-4)
    # the wave-owned form (training, H = 64, F <= 16, segments of <= 32 det rows, no gradient with respect to x): many
    # segments, tiles of one to a dozen segments, several groups per workgroup
    for F_, S, maxn in ((8, 150, 33), (13, 700, 12), (8, 40, 2), (16, 9000, 8)):
        for k, v in check_input_bn(64, F_, True, S=S, fused=True, maxn=maxn, dx=False).items():
            rec(f'input_tf (wave-owned) H=64 F={F_} S={S} rows<{maxn} {k}', v, 2e-4)
    run_attention(rec, g)
    torch.cuda.synchronize()
    return results


if __name__ == '__main__':
    if 'attention' in sys.argv[1:]:
        res = []

        def _rec(name, err, tol):
            res.append((name, err, tol))
            print(f'{"OK  " if err <= tol else "FAIL"} {name:48s} err={err:.3e} tol={tol:.0e}')
        run_attention(_rec, make_graph())
        torch.cuda.synchronize()
    else:
        res = run_all()
    bad = [r for r in res if not (r[1] <= r[2])]
    print(f'{len(res) - len(bad)}/{len(res)} stage checks passed')
    sys.exit(1 if bad else 0)
