#!/usr/bin/env python3
"""graph-edges/sec (fwd+bwd) of the TrackMPNN message-passing hot path on MI355X (BASELINE.json metric).

One step = one pass of the hot path over one batch of synthetic input: B KITTI-Car/RRC-shaped
rolling tracking windows (BASELINE.json configs[1] = SURVEY 8(d) C2: 7 frames, D_t ~ clip(Poisson(6),1,20),
F = 8 '2d' features, H = 64, no attention, diff messages), batched block-diagonally.  Per window the
reference call pattern is reproduced exactly (train.py:65-68,92-107,132-135): one forward per frame on the
growing graph with the hidden state carried (BPTT), one backward of a BCE loss over all logits; then (N > 1)
ONE flat-bucket RCCL all-reduce of the gradients and an Adam step.  Graphs are prebuilt and resident in HBM.
A "graph-edge" = one edge node processed by one forward call.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured stream copy
MFMA_F32_PEAK_TF = 157.3     # dense fp32-input MFMA peak (same guide)


def build_batch(B, frames, mean_dets, max_dets, F, seed, device):
    """B windows = 64 distinct seeded windows tiled; returns (plans on device, per-call x on device, stats)."""
    from trackmpnn_amd import WindowBuilder, batch_windows, synth_window
    distinct = min(B, 64)
    wins = [WindowBuilder(synth_window(seed * 1000 + s, frames, mean_dets, max_dets)).calls() for s in range(distinct)]
    reps = (B + distinct - 1) // distinct
    wins = (wins * reps)[:B]
    plans, refs = batch_windows(wins, device='cpu')
    gen = torch.Generator().manual_seed(seed)
    xs = []
    for plan, ref in zip(plans, refs):
        x = torch.zeros(plan.n_new, F)
        x[plan.new_det_local] = torch.randn(len(ref), F, generator=gen)     # standardised features (kitti_mot.py:545-566)
        xs.append(x.to(device))
    plans = [p.to(device) for p in plans]
    edge_iters = sum(p.graph.E for p in plans)
    return plans, xs, edge_iters


def make_adam(model):
    """optim.Adam(model.parameters(), lr, weight_decay) as train.py:329 constructs it; the single-launch (fused) form of torch's
    Adam where this torch build offers it for the device (same update rule; the default form is ~20 small launches per step)."""
    kw = dict(lr=1e-4, weight_decay=5e-4)
    try:
        return torch.optim.Adam(model.parameters(), fused=True, **kw)
    except (RuntimeError, TypeError, ValueError):
        return torch.optim.Adam(model.parameters(), **kw)


def step(model, plans, xs, targets, opt, bucket, world):
    """fwd over every call of the window batch, one backward, (all-reduce), Adam.  Loss: BCE with logits over ALL logits of
    every call against fixed {0,1} targets, summed -- trackmpnn_amd.loss.bce_with_logits_sum, one launch per
    direction (equal to torch.nn.functional.binary_cross_entropy_with_logits(reduction='sum'), tests/test_loss.py)."""
    from trackmpnn_amd.functional import weight_cache
    from trackmpnn_amd.loss import bce_with_logits_sum
    h = None
    loss = 0.0
    with weight_cache():        # the weights do not change between the forward calls of one step: their transposes are built once
        for c, (plan, x, t) in enumerate(zip(plans, xs, targets)):
            nxt = plans[c + 1].n_new if c + 1 < len(plans) else 0
            scores, logits, h, _ = model.forward_graph(x, h, plan, reserve_rows=nxt)
            loss = loss + bce_with_logits_sum(logits, t)
    opt.zero_grad(set_to_none=False)
    loss.backward()
    if world > 1:
        import torch.distributed as dist
        from trackmpnn_amd.dist import allreduce_grads
        allreduce_grads(model, bucket, world)
    opt.step()
    return loss


def time_stage(fn, iters=5):
    """Average duration (ms) of one enqueue of `fn` on the current stream, measured with HIP events."""
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def stage_profile(model, plan, H):
    """Per-kernel timing on the LAST call's graph (the largest): the fused edge GRU kernels (MFMA-bound) and
    the stand-alone aggregation kernels (HBM-bound, SURVEY 8(d) byte model)."""
    from trackmpnn_amd import _lib
    g = plan.graph
    dev = g.device
    N, E, Dn = g.N, g.E, g.Dn
    st = torch.cuda.current_stream().cuda_stream
    f = 'factor_grus.0.'
    P = dict(model.named_parameters())
    h = torch.randn(N, H, device=dev)
    out = torch.empty(N, H, device=dev)
    gates = torch.empty(4, N, H, device=dev)
    wih_t = P[f + 'edge_gru.weight_ih'].detach().t().contiguous()
    whh_t = P[f + 'edge_gru.weight_hh'].detach().t().contiguous()
    wih, whh = P[f + 'edge_gru.weight_ih'].detach(), P[f + 'edge_gru.weight_hh'].detach()
    bih, bhh = P[f + 'edge_gru.bias_ih'].detach(), P[f + 'edge_gru.bias_hh'].detach()
    dout = torch.randn(N, H, device=dev)
    dmsg = torch.empty(N, H, device=dev)
    dh = torch.empty(N, H, device=dev)
    gW = [torch.zeros_like(wih), torch.zeros_like(whh), torch.zeros_like(bih), torch.zeros_like(bhh)]
    wsb = _lib.load().tmpnn_gru_bwd_weights_ws(E, H, H)
    ws = torch.empty(wsb // 4 + 1, device=dev)
    es = torch.empty(Dn, H, device=dev)

    proj = torch.empty(Dn, 3 * H, device=dev)

    from trackmpnn_amd import functional as _fn
    from trackmpnn_amd.graph import edge_tiles
    tiles32 = edge_tiles(g, _fn.FWD_TILE_ROWS) if _fn.FWD_TILED else None

    def gru_fwd():      # as the training step runs it: det rows projected once, edge cell takes P[src] - P[dst] per edge tile
        _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), Dn, h.data_ptr(), H, H, wih_t.data_ptr(), 3 * H,
                  proj.data_ptr(), 3 * H, st)
        if tiles32 is not None:
            _lib.call('tmpnn_gru_fwd_tiles', tiles32.cref(), E, proj.data_ptr(), 3 * H, h.data_ptr(), H, H, whh_t.data_ptr(),
                      bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H, gates.data_ptr(), N * H, None, None, 0, st)
        else:
            _lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, 3, g.src_pos.data_ptr(), g.dst_pos.data_ptr(),
                      proj.data_ptr(), 3 * H, 0, H, h.data_ptr(), H, H, None, whh_t.data_ptr(), bih.data_ptr(),
                      bhh.data_ptr(), out.data_ptr(), H, gates.data_ptr(), N * H, None, None, 0, st)

    def gru_bwd_data():
        _lib.call('tmpnn_gru_bwd_data', g.edge_row.data_ptr(), E, H, h.data_ptr(), H, H, wih.data_ptr(), whh.data_ptr(),
                  gates.data_ptr(), N * H, dout.data_ptr(), H, None, None, dmsg.data_ptr(), H, dh.data_ptr(), H,
                  None, None, None, 0, st)

    dyv = torch.randn(N, device=dev)
    w_head = torch.randn(H, device=dev)

    def gru_bwd_data_folded():      # as the training step runs it: head term folded in, row-F adjoint fused
        _lib.call('tmpnn_gru_bwd_data', g.edge_row.data_ptr(), E, H, h.data_ptr(), H, H, wih.data_ptr(), whh.data_ptr(),
                  gates.data_ptr(), N * H, dout.data_ptr(), H, dyv.data_ptr(), w_head.data_ptr(), dmsg.data_ptr(), H,
                  dh.data_ptr(), H, g.src.data_ptr(), g.dst.data_ptr(), dmsg.data_ptr(), H, st)

    def gru_bwd_w():
        _lib.call('tmpnn_gru_bwd_weights', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0,
                  H, h.data_ptr(), H, H, gates.data_ptr(), N * H, dout.data_ptr(), H, None, None, gW[0].data_ptr(),
                  gW[1].data_ptr(), gW[2].data_ptr(), gW[3].data_ptr(), ws.data_ptr(), wsb, st)

    fwsb = _lib.load().tmpnn_gru_bwd_fused_ws(E, H, H) if _lib.load().tmpnn_gru_bwd_fused_available(H, H, 1) else 0
    fws = torch.empty(fwsb // 4 + 1, device=dev)

    def gru_bwd_one():              # as the training step runs it: ONE pass for data + weights, head folded, row F fused
        _lib.call('tmpnn_gru_bwd_fused', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0, H,
                  h.data_ptr(), H, H, wih.data_ptr(), whh.data_ptr(), gates.data_ptr(), N * H, dout.data_ptr(), H,
                  dyv.data_ptr(), w_head.data_ptr(), dmsg.data_ptr(), H, dh.data_ptr(), H, g.src.data_ptr(),
                  g.dst.data_ptr(), dmsg.data_ptr(), H, gW[0].data_ptr(), gW[1].data_ptr(), gW[2].data_ptr(),
                  gW[3].data_ptr(), fws.data_ptr(), fwsb, st)

    nwih_t = P[f + 'node_gru.weight_ih'].detach().t().contiguous()
    nwhh_t = P[f + 'node_gru.weight_hh'].detach().t().contiguous()
    nbih, nbhh = P[f + 'node_gru.bias_ih'].detach(), P[f + 'node_gru.bias_hh'].detach()

    def gru_fwd_node():             # as the training step runs it: x = the compact aggregate es[d], both products per det row
        _lib.call('tmpnn_gru_fwd', g.det_row.data_ptr(), Dn, 0, None, None, es.data_ptr(), H, 1, H, h.data_ptr(), H, H,
                  nwih_t.data_ptr(), nwhh_t.data_ptr(), nbih.data_ptr(), nbhh.data_ptr(), out.data_ptr(), H, gates.data_ptr(),
                  N * H, None, None, 0, st)

    def gather():
        _lib.call('tmpnn_gather_diff_fwd', g.cref(), h.data_ptr(), H, out.data_ptr(), H, H, 0, st)

    def segsum():
        _lib.call('tmpnn_segsum_fwd', g.cref(), h.data_ptr(), H, es.data_ptr(), H, H, 0, 1, st)

    gru_fwd()          # gates must hold sane values before the backward kernels read them
    t = {name: time_stage(fn) for name, fn in (('gru_fwd_edge', gru_fwd), ('gru_bwd_data_edge', gru_bwd_data),
                                               ('gru_bwd_data_edge_folded', gru_bwd_data_folded),
                                               ('gru_bwd_weights_edge', gru_bwd_w),
                                               ('gather_diff', gather),
                                               ('segsum', segsum), ('gru_fwd_node', gru_fwd_node)) + ((('gru_bwd_one_edge', gru_bwd_one),) if fwsb else ())}
    flops = {'gru_fwd_node': 12.0 * H * H * Dn, 'gru_fwd_edge': 12.0 * H * H * E, 'gru_bwd_data_edge': 12.0 * H * H * E,
             'gru_bwd_data_edge_folded': 12.0 * H * H * E,
             'gru_bwd_weights_edge': 12.0 * H * H * E, 'gru_bwd_one_edge': 24.0 * H * H * E}
    # SURVEY 8(d) algorithmic bytes per launch (every array counted once; det-row gathers count the det table once)
    b_gather = 4.0 * H * E + 4.0 * H * Dn + 8.0 * E
    b_segsum = 4.0 * H * E + 4.0 * H * Dn + 4.0 * (2 * E + Dn + 1) + 2.0 * E
    nbytes = {'gather_diff': b_gather, 'segsum': b_segsum,
              # per det row: es 4H + h 4H in, h_out 4H + gates 16H out, the row id
              'gru_fwd_node': (28.0 * H + 4.0) * Dn,
              # det rows -> P (4H in, 12H out per det), then per edge: h 4H in, h_out 4H + gates 16H out, 3 ids; P read once
              'gru_fwd_edge': (24.0 * H + 12.0) * E + 28.0 * H * Dn,
              # per edge: dh 4H + gates 16H + h 4H in, d_msg 4H + d_h 4H out, row id
              'gru_bwd_data_edge': (32.0 * H + 4.0) * E,
              # ... + dy, two ids and the fused row-F adjoint (its det table once)
              'gru_bwd_data_edge_folded': (32.0 * H + 16.0) * E + 4.0 * H * Dn,
              # per edge: dh 4H + gates 16H + h 4H in, 3 ids; h[src], h[dst] from the det table (once)
              'gru_bwd_weights_edge': (24.0 * H + 12.0) * E + 4.0 * H * Dn}
    # one pass: dh 4H + gates 16H + h 4H in, d_msg 4H + d_h 4H out, dy and three ids; h and d_es det tables once
    nbytes['gru_bwd_one_edge'] = (32.0 * H + 16.0) * E + 8.0 * H * Dn
    return t, flops, nbytes


def wide_stage_profile(g, H):
    """C5 (H >= 128 cells, csrc/wide.hip): the dominant kernel of the step -- the wide edge cell's forward over 128-row edge
    tiles, `tmpnn_wide_gru_fwd_tiled` (k_wide_prep / k_wide_gemm_store for the projected det rows + k_wide_gru_fwd_pp) -- timed
    with HIP events on the launch stream and priced against both roofs: algorithmic bytes (24 H + 12) E + 28 H Dn over 8 TB/s,
    and the executed bf16 products (6 per fp32 product of the 3H x H recurrent GEMM) over the 2.5 PFLOP/s dense bf16 peak."""
    from trackmpnn_amd import _lib
    from trackmpnn_amd.graph import edge_tiles
    dev = g.device
    N, E, Dn = g.N, g.E, g.Dn
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    h = torch.randn(N, H, device=dev, generator=gen)
    sc = 1.0 / H ** 0.5
    wih, whh = sc * torch.randn(3 * H, H, device=dev, generator=gen), sc * torch.randn(3 * H, H, device=dev, generator=gen)
    bih, bhh = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
    prep = torch.empty(int(lib.tmpnn_wide_prep_bytes(H, H)) // 4 + 4, device=dev)
    _lib.call('tmpnn_wide_prepare', wih.data_ptr(), whh.data_ptr(), H, H, prep.data_ptr(), st)
    P = torch.empty(Dn, 3 * H, device=dev)
    out = torch.empty(N, H, device=dev)
    gates = torch.empty(4, N, H, device=dev)
    tiles = edge_tiles(g, 128)

    def fwd():
        _lib.call('tmpnn_wide_gru_fwd_tiled', prep.data_ptr(), g.det_row.data_ptr(), Dn, tiles.cref(), E, h.data_ptr(), H, H,
                  bih.data_ptr(), bhh.data_ptr(), P.data_ptr(), out.data_ptr(), H, gates.data_ptr(), N * H, st)
    ms = time_stage(fwd, iters=3)
    nb = (24.0 * H + 12.0) * E + 28.0 * H * Dn
    gbs = nb / (ms * 1e-3) / 1e9
    bf16_tf = 6.0 * 2.0 * 3.0 * H * H * (E + Dn) / (ms * 1e-3) / 1e12
    hbm_frac, pipe_frac = gbs / HBM_PEAK_GBS, bf16_tf / 2500.0
    if hbm_frac >= pipe_frac:
        roof = dict(bound='hbm', kernel='wide_gru_fwd_edge (k_wide_gru_fwd_pp)', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=hbm_frac, traffic=None, ms=ms, algorithmic_bytes=nb, matrix_pipe_frac=pipe_frac)
    else:
        roof = dict(bound='mfma', kernel='wide_gru_fwd_edge (k_wide_gru_fwd_pp)', achieved=bf16_tf, peak=2500.0, unit='TFLOP/s',
                    frac=pipe_frac, traffic=None, ms=ms, hbm_frac=hbm_frac, dtype='bf16 pieces of fp32 operands (bf16x6)')
    return roof, {'wide_gru_fwd_edge': dict(ms=round(ms, 4), GBs=round(gbs, 1), hbm_frac=round(hbm_frac, 3),
                                            bf16_tflops=round(bf16_tf, 1))}


def att_bytes(E, Dn, H, K, train):
    """Algorithmic bytes of the attention stage per call (every array counted once; the det tables once).
    SURVEY 8(d) budgets, per head, a 4 E score, 8 E of alpha per incidence and ANOTHER 4 H E read of h[e]; this implementation
    reads h[e] once for all heads in the forward and once in the backward.
    forward  = row GEMM (h dets in, ha out) + score (ha table, 16-byte edge record, score out at both positions) + det pass
               (h edge rows, score, incidences, rowptr + order, keep byte, alpha out, es + per-head es + statistics out);
    backward = det records (d_es, es_k, statistics in, 16 K out) + edge pass (32-byte edge record, h[e], d_h[e] read +
               written, d_es / ha / record tables, score at one position, two keep bytes, dpre out at both positions) +
               det pass (other endpoints, dpre, ha table, rowptr + order, d_ha out) + three row products (d_ha twice,
               h dets, d_h dets read + written)."""
    kb = 2.0 if train else 0.0
    fwd = E * (4.0 * H + 24.0 * K + 24.0 + kb) + Dn * (8.0 * H + 12.0 * K * H + 8.0 * K + 8.0)
    bwd = E * (12.0 * H + 20.0 * K + 40.0 + kb) + Dn * (20.0 * H + 24.0 * K * H + 40.0 * K + 8.0)
    return fwd, bwd


def att_stage_profile(g, H, K, train, iters=5):
    """HIP-event timing of tmpnn_att_fwd / tmpnn_att_bwd (and the plain segment sum + its adjoint, which they replace)
    on graph `g`: ({stage: ms}, {stage: algorithmic bytes})."""
    from trackmpnn_amd import _lib
    dev = g.device
    N, E, Dn = g.N, g.E, g.Dn
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev).manual_seed(3)
    h = torch.randn(N, H, device=dev, generator=gen)
    W = 0.3 * torch.randn(H, K * H, device=dev, generator=gen)
    av = 0.3 * torch.randn(K, H, device=dev, generator=gen)
    ha = torch.empty(Dn, K * H, device=dev)
    score = torch.empty(2 * E, K, device=dev)
    stats = torch.empty(Dn, K, 2, device=dev)
    esk = torch.empty(K, Dn, H, device=dev)
    alpha = torch.empty(K, 2 * E, device=dev)
    es = torch.empty(Dn, H, device=dev)
    keep = torch.empty(2 * E, dtype=torch.uint8, device=dev).random_(0, 1 << K) if train else None
    d_out = torch.randn(N, H, device=dev, generator=gen)
    d_h = torch.zeros(N, H, device=dev)
    dW = torch.zeros(K, H, H, device=dev)
    da = torch.zeros(K, H, device=dev)
    wsn = _lib.load().tmpnn_att_bwd_ws(E, Dn, H, K)
    ws = torch.empty(wsn + 4, device=dev)
    erec, inc_other = g.att_index()

    def att_fwd():
        _lib.call('tmpnn_att_fwd', g.cref(), erec.data_ptr(), h.data_ptr(), H, H, K, W.data_ptr(),
                  av.data_ptr(), _lib.ptr(keep), 0.5, ha.data_ptr(), score.data_ptr(), stats.data_ptr(), esk.data_ptr(),
                  alpha.data_ptr(), es.data_ptr(), H, st)

    def att_bwd():
        _lib.call('tmpnn_att_bwd', g.cref(), erec.data_ptr(), inc_other.data_ptr(), h.data_ptr(), H,
                  H, K, W.data_ptr(), av.data_ptr(), _lib.ptr(keep), 0.5, ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
                  esk.data_ptr(), d_out.data_ptr(), H, ws.data_ptr(), wsn, d_h.data_ptr(), H, dW.data_ptr(), da.data_ptr(), st)

    def segsum():
        _lib.call('tmpnn_segsum_fwd', g.cref(), h.data_ptr(), H, es.data_ptr(), H, H, 0, 1, st)

    def gather():
        _lib.call('tmpnn_gather_diff_fwd', g.cref(), d_out.data_ptr(), H, d_h.data_ptr(), H, H, 1, st)

    t = {name: time_stage(fn, iters) for name, fn in (('att_fwd', att_fwd), ('att_bwd', att_bwd), ('segsum', segsum),
                                                       ('segsum_adjoint', gather))}
    bf, bb = att_bytes(E, Dn, H, K, train)
    nbytes = {'att_fwd': bf, 'att_bwd': bb,
              'segsum': 4.0 * H * E + 4.0 * H * Dn + 4.0 * (2 * E + Dn + 1) + 2.0 * E,
              'segsum_adjoint': 8.0 * H * E + 4.0 * H * Dn + 12.0 * E}
    return t, nbytes


def split_enabled():
    """The library default: GRU GEMMs on the bf16 matrix pipe as fp32-accurate 3 x 3 split products (bf16x6,
    csrc/gru_common.h); TMPNN_SPLIT=0 keeps them on the f32-input MFMA."""
    return os.environ.get('TMPNN_SPLIT', '1')[:1] != '0'


def _pmc_kernel(stage):
    if split_enabled():
        from trackmpnn_amd import functional as _fn
        return {'gru_fwd_edge': 'k_gru_fwd_split_tiled<64, 8>' if _fn.FWD_TILED else 'k_gru_fwd_split<64, 8>', 'gru_bwd_data_edge': 'k_gru_bwd_data_split<64, 1, false>',
                'gru_bwd_data_edge_folded': 'k_gru_bwd_data_split<64, 3, true>',
                'gru_bwd_weights_edge': 'k_gru_bwd_weights_split<1, 1>',
                'segsum': 'k_segsum_pipe<false, 4, 32, false', 'gather_diff': 'k_gather_pipe<false, false',
                'gru_fwd_node': 'k_gru_fwd_split_node<64',
                'gru_bwd_one_edge': ('k_gru_bwd_one<1, 3, true>' if os.environ.get('TMPNN_BWD_TWO', '1')[:1] == '0'
                                     else 'k_gru_bwd_two<1, 3, true')}.get(stage, '?')       # (prefix: the template list grew in round 4)
    return {'gru_fwd_edge': 'k_gru_fwd_lds<64, 64, 3,', 'gru_bwd_data_edge': 'k_gru_bwd_data_lds<64, 64, 1, false>',
            'gru_bwd_data_edge_folded': 'k_gru_bwd_data_lds<64, 64, 3, true>',
            'gru_bwd_weights_edge': 'k_gru_bwd_weights_lds<64, 1, 1>'}.get(stage, '?')


def kernel_source_digest(files=('gru_common.h', 'gru_fwd.hip', 'gru_bwd.hip', 'agg.hip', 'common.h')):
    """sha256 (first 16 hex digits) of the kernel sources a committed PMC pass belongs to: tools/collect_r05.py records it
    next to the counters, `pmc_traffic` compares it with the tree that is running."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, 'trackmpnn_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(stage, E):
    """(bytes, source) -- HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in two separate runs of tools/stage_bench.py, profiles/rNN_pmc_traffic_stage_kernels.json), corrected as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream
    (16 B per lane -- every row stream of these kernels) at 64 bytes, so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact for
    16-byte-per-lane stores.  The counters are NOT collected inside the bench run (a --pmc pass serialises and slows every
    kernel); `source` says where the number comes from (file, the date and HEAD of the pass, the digest of the kernel
    sources it was taken on).  bytes is None when no profile was taken on a graph of exactly this size, or when the
    profile records a source digest and the kernel sources have changed since (`source.stale`)."""
    for tag in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        name_ = f'{tag}_pmc_traffic_stage_kernels.json'
        try:
            prof = json.load(open(os.path.join(ROOT, 'profiles', name_)))
        except OSError:
            continue
        if prof.get('graph', {}).get('E') != E:
            continue
        src = dict(file='profiles/' + name_, collected=prof.get('collected'), head=prof.get('head'),
                   kernel_source_digest=prof.get('kernel_source_digest'), method='committed rocprofv3 --pmc passes, not live')
        if prof.get('kernel_source_digest') and prof['kernel_source_digest'] != kernel_source_digest():
            src['stale'] = 'kernel sources changed since the pass (digest now %s)' % kernel_source_digest()
            return None, src
        for name, v in prof['kernels'].items():
            if _pmc_kernel(stage) in name and v.get('WRITE_SIZE_KB') is not None:
                return (2.0 * v['FETCH_SIZE_KB'] + v['WRITE_SIZE_KB']) * 1024.0, src
    return None, None


def cpu_baseline(frames, mean_dets, max_dets, F, H, seed, budget_s=20.0):
    """The oracle (CPU restatement, kind 'port') on the host cores, same C2 generator and call pattern, bounded
    sample, in three forms (oracle/cpu_bench.py): one core at batch 1 (how the reference runs, utils/graph.py:117),
    4 batch-1 worker processes (a GPU box admits 6 processes on the card, and importing torch opens it), and
    block-diagonal batches of 256 windows with all 16 threads of the box's CPU share.  The BEST form is the
    stated baseline; the others ride along in `forms`."""
    from oracle import cpu_bench
    best, forms = cpu_bench.measure(frames, mean_dets, max_dets, F, H, seed, budget_s=budget_s)
    b = forms[best]
    desc = {'single': 'one window at a time on one core (batch 1 as the reference)',
            'procs': f"{b['cores']} independent batch-1 worker processes, one core each",
            'batched': f"block-diagonal batches of 256 windows, {b['cores']} threads"}[best]
    return dict(value=b['value'], unit='graph-edges/s', cores=b['cores'], kind='port',
                sample=f"C2 generator, torch-CPU fp32 oracle fwd+bwd, {desc}: {b['edge_iterations']} edge-iterations "
                       f"in {b['seconds']} s; best of single / procs / batched on a {os.cpu_count()}-core host",
                forms={k: dict(value=round(v['value'], 1), cores=v['cores']) for k, v in forms.items()})


def latency_batch1():
    """The reference's REAL call pattern (batch size 1, train.py:92-107): one window through the drop-in call
    `model(x, h_in, node_adj, edge_adj)` with the reference's own adjacency tensors (fixtures generated by the real
    reference): C1 of BASELINE.json (static 5 x 20 window, 2 MP iterations) and one C2 window (7 frames: 6 rolling calls
    + 1 extra iteration).  A step = every forward call + loss + backward (no optimizer), (a) issued eagerly, adjacency ->
    index conversion included, (b) replayed from a hipGraph captured once for the window (CapturedWindow; conversion
    done at capture time), next to the CPU oracle on the same window on this host (1 thread: more threads are slower at
    this size)."""
    from oracle import trackmpnn_oracle as orc
    from tests.golden_util import Golden
    from trackmpnn_amd import CapturedWindow, TrackMPNN
    from trackmpnn_amd.dist import GradBucket
    dev = torch.device('cuda', torch.cuda.current_device())
    out = {}
    for tag, name in (('c1', 'c1_static_diff_k0_train'), ('c2_window', 'roll_c2_kitti_car_w5')):
        gold = Golden(name)
        m = gold.meta
        model = TrackMPNN(m['features'], m['ncategories'], m['nhidden'], m['nattheads'], m['msg_type'])
        model.load_state_dict(gold.params(), strict=True)
        model = model.to(dev).train()
        bucket = GradBucket(model)
        calls, ograph, ox = [], [], []
        for c in range(gold.ncalls):
            na, ea = gold.adjacency(c, 'node_adj'), gold.adjacency(c, 'edge_adj')
            ograph.append(orc.graph_from_adjacency(na, ea))
            ox.append(gold.t(f'c{c}/x'))
            na, ea = na.to(dev), ea.to(dev)
            if not na.is_sparse:                   # initialize_graph(cuda=True) hands over sparse tensors
                na, ea = na.to_sparse(), ea.to_sparse()
            calls.append((ox[-1].to(dev), na, ea))
        E = sum(g.E for g in ograph)
        loss_fn = lambda outs, h: torch.cat([l for _, l in outs]).sum()      # noqa: E731

        def step():
            h, outs = None, []
            for x, na, ea in calls:
                s, l, h, _ = model(x, h, na, ea)
                outs.append((s, l))
            bucket.zero()
            loss_fn(outs, h).backward()

        def timed(fn, n):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        eager = timed(step, 200)
        # the same loop as a reference user gets it by swapping the import ONLY: no GradBucket, gradients returned to
        # autograd (the native node's gradient sink), .grad released between steps as optimizer.zero_grad() does
        plain_model = TrackMPNN(m['features'], m['ncategories'], m['nhidden'], m['nattheads'], m['msg_type'])
        plain_model.load_state_dict(gold.params(), strict=True)
        plain_model = plain_model.to(dev).train()

        def plain_step():
            h, outs = None, []
            for x, na, ea in calls:
                s, l, h, _ = plain_model(x, h, na, ea)
                outs.append((s, l))
            for p_ in plain_model.parameters():
                p_.grad = None
            loss_fn(outs, h).backward()

        eager_plain = timed(plain_step, 200)
        win = CapturedWindow(model, calls, loss_fn, optimizer=None, bucket=bucket)
        captured = timed(win.replay, 200)
        # CPU oracle, same window, same loss
        cfg = orc.OracleConfig(m['features'], m['ncategories'], m['nhidden'], m['nattheads'], m['msg_type'])
        p = gold.params()
        for k, v in p.items():
            if v.dtype.is_floating_point and not k.endswith(orc.BUFFER_SUFFIXES):
                v.requires_grad_(True)

        def cpu_step():
            h, loss = None, 0.0
            for g, x in zip(ograph, ox):
                s, l, h, _ = orc.forward(p, cfg, x, h, g, training=True)
                loss = loss + l.sum()
            for v in p.values():
                v.grad = None
            loss.backward()

        torch.set_num_threads(1)
        cpu_step()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            cpu_step()
            reps += 1
        cpu_ms = (time.perf_counter() - t0) / reps * 1e3
        out[tag] = dict(fixture=name, calls=gold.ncalls, rows=int(calls[-1][1].shape[0]), edge_iterations=E,
                        eager_ms=round(eager, 4), eager_plain_ms=round(eager_plain, 4), captured_ms=round(captured, 4),
                        cpu_oracle_ms=round(cpu_ms, 3), speedup_vs_cpu_oracle_eager_plain=round(cpu_ms / eager_plain, 2),
                        eager_edges_per_s=E / eager * 1e3, captured_edges_per_s=E / captured * 1e3,
                        speedup_vs_cpu_oracle_eager=round(cpu_ms / eager, 2),
                        speedup_vs_cpu_oracle_captured=round(cpu_ms / captured, 2))
    # models OUTSIDE the fused batch-1 path on the same C2 window: staged kernels per call, eager and replayed from one hipGraph
    try:
        var = {}
        for vtag, H_, K_ in (('k2_heads_h64', 64, 2), ('k0_h128', 128, 0), ('k0_h48_padded', 48, 0)):
            torch.manual_seed(5)
            model = TrackMPNN('2d', 3, H_, K_, 'diff').to(dev).train()
            bucket = GradBucket(model)

            def vstep():
                h, outs = None, []
                for x, na, ea in calls:
                    s, l, h, _ = model(x, h, na, ea)
                    outs.append((s, l))
                bucket.zero()
                loss_fn(outs, h).backward()
            for _ in range(3):
                vstep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                vstep()
            torch.cuda.synchronize()
            eager_v = (time.perf_counter() - t0) / 10 * 1e3
            win = CapturedWindow(model, calls, loss_fn, optimizer=None, bucket=bucket)
            for _ in range(3):
                win.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                win.replay()
            torch.cuda.synchronize()
            var[vtag] = dict(eager_ms=round(eager_v, 3), captured_ms=round((time.perf_counter() - t0) / 20 * 1e3, 3))
        out['c2_window_staged_models'] = var
    except Exception as e:                                    # noqa: BLE001  (a reporting extra must not sink the bench line)
        out['c2_window_staged_models'] = dict(error=f'{type(e).__name__}: {e}'[:200])
    out['note'] = ('fwd + loss + bwd of ONE window, fp32; eager = model(x, h, node_adj, edge_adj) per call incl. the '
                   'adjacency conversion, gradients accumulated in place (GradBucket: one extra line in the training loop); '
                   'eager_plain = the import swap alone (gradients returned to autograd); captured = the eager step '
                   'replayed from one hipGraph; cpu oracle: 1 thread; c2_window_staged_models = the same window for models the '
                   'fused iteration does not cover (attention heads, nhidden 128, a zero-padded width): staged kernels, eager / captured')
    return out


LOOP_SHAPES = {'C2': dict(frames=7, mean=6.0, mx=20, ncat=3, win=5), 'C3': dict(frames=12, mean=8.0, mx=25, ncat=3, win=10),
               'C4': dict(frames=7, mean=12.0, mx=40, ncat=8, win=5)}
LOOP_INFER_FRAMES = 40


def loop_batch1(budget_s=1.5):
    """The reference's two REAL per-timestep loops, end to end at batch 1 (SURVEY 8(f) rows 1-3 composed,
    trackmpnn_amd/loops.py), on the synthetic sequences oracle/time_reference_loops.py times the real reference on
    (BASELINE.md has those numbers; the reference cannot run on the GPU box):
      train chunk (train.py:54-135): initialize_graph, then per timestep update_graph(mode='train') -> model ->
        create_targets + CELoss + FocalLoss; one backward + Adam step -- C2 / C3 / C4-shaped chunks;
      inference (infer.py:35-87): update_graph(mode='test') -> model -> decode_tracks over a 40-frame sequence, greedy
        and Hungarian association.
    Per case: the un-instrumented wall time and, from a second pass with a device synchronisation around every stage,
    where it goes (graph maintenance / model forward / targets + losses / backward + optimizer / decode)."""
    from trackmpnn_amd import TrackMPNN
    from trackmpnn_amd.graph import synth_window
    from trackmpnn_amd.loops import infer_sequence, train_chunk
    dev = torch.device('cuda', torch.cuda.current_device())

    def sequence(seed, frames, mean, mx, ncat):
        yy = synth_window(seed, frames, mean, mx)
        y = torch.from_numpy(yy)[None]
        X = torch.randn(1, yy.shape[0], ncat + 5, generator=torch.Generator().manual_seed(seed + 1000))
        return X, y

    def perturb(model, seed):        # scores on both sides of 0.5; the same code as oracle/time_reference_loops.py
        gp = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for k, prm in model.named_parameters():
                prm.add_(0.1 * torch.randn(prm.shape, generator=gp))
                if k.startswith('output_transform') and k.endswith('bias'):
                    prm.copy_(0.5 * torch.randn(prm.shape, generator=gp))

    # objects alive here (the C2 batch's plans, the models) go to the permanent generation: a full collection of the interpreter's
    # garbage collector in the middle of a timed sequence otherwise shows up as a multi-millisecond `slowest_sequence_ms`
    import gc
    gc.collect()
    gc.freeze()
    worst = {}

    def timed(fn, tag=None):
        for _ in range(3):
            r = fn()
        torch.cuda.synchronize()
        n, t0, mx = 0, time.perf_counter(), 0.0
        while time.perf_counter() - t0 < budget_s or n < 3:
            t1 = time.perf_counter()
            r = fn()
            mx = max(mx, time.perf_counter() - t1)       # (host time of one call: the loops read the device every timestep)
            n += 1
        torch.cuda.synchronize()
        if tag is not None:
            worst[tag] = (round(mx * 1e3, 3), n)
        return (time.perf_counter() - t0) / n * 1e3, r

    out = dict(train={}, infer={})
    for tag, s in LOOP_SHAPES.items():
        torch.manual_seed(5)
        model = TrackMPNN('2d', s['ncat'], 64, 0, 'diff')
        perturb(model, 4242)
        model = model.to(dev)
        X, y = sequence(1001, s['frames'], s['mean'], s['mx'], s['ncat'])
        Xi, yi = sequence(2001, LOOP_INFER_FRAMES, s['mean'], s['mx'], s['ncat'])
        model.train()
        opt = make_adam(model)

        def chunk(stages=None):
            opt.zero_grad(set_to_none=True)
            r = train_chunk(model, X, y, dev, stages=stages)
            if stages is not None:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            opt.step()
            if stages is not None:
                torch.cuda.synchronize()
                stages['optimizer'] = stages.get('optimizer', 0.0) + time.perf_counter() - t1
            return r

        ms, (loss, ncalls, edges) = timed(chunk)
        st = {}
        chunk(st)
        out['train'][tag] = dict(ms_per_chunk=round(ms, 3), calls=ncalls, edge_iterations=edges, edges_per_s=round(edges / ms * 1e3),
                                 stages_ms={k: round(v * 1e3, 3) for k, v in st.items()})
        model.eval()
        for hung in (False, True):
            key = f"{tag}/{'hungarian' if hung else 'greedy'}"
            ms, (y_out, ncalls, edges) = timed(lambda: infer_sequence(model, Xi, yi, s['win'], 0, hung, dev), key)
            st = {}
            infer_sequence(model, Xi, yi, s['win'], 0, hung, dev, stages=st)
            out['infer'][key] = dict(
                ms_per_sequence=round(ms, 3), slowest_sequence_ms=worst[key][0], sequences_timed=worst[key][1],
                frames=LOOP_INFER_FRAMES, ms_per_timestep=round(ms / LOOP_INFER_FRAMES, 4),
                calls=ncalls, edge_iterations=edges, tracks=int(y_out[:, 1].max()) + 1,
                stages_ms={k: round(v * 1e3, 3) for k, v in st.items()})
        # the loop as the reference README runs it (README.md:52-67, 113-122): --hungarian with a model trained by
        # --no-tp-classifier -- every detection counts as a true positive (infer.py:77-80), so active sets and graphs are larger
        ms, (y_out, ncalls, edges) = timed(lambda: infer_sequence(model, Xi, yi, s['win'], 0, True, dev, False))
        out['infer'][f'{tag}/hungarian_no_tp_classifier'] = dict(
            ms_per_sequence=round(ms, 3), frames=LOOP_INFER_FRAMES, ms_per_timestep=round(ms / LOOP_INFER_FRAMES, 4), calls=ncalls,
            edge_iterations=edges, tracks=int(y_out[:, 1].max()) + 1)
    out['note'] = ('batch 1, fp32, one process; ms_* = un-instrumented wall time incl. every host read; stages_ms = a second, '
                   'instrumented pass (device sync around each stage: its sum exceeds the wall time).  The real reference on '
                   'the same sequences: BASELINE.md section 5 (build-container CPU; it cannot travel to this box)')
    return out


# BASELINE.json configs as bench workloads.  c2 is the headline (configs[1]) and the default for every N;
# c4 (configs[3]) and c5 (configs[4]) are the two configurations that are multi-GPU by definition: `--workload c4|c5` times
# them as the main line, and with N > 1 the default c2 run appends both as `scaling_extras` behind the timed c2 region.
WORKLOADS = {
    'c2': dict(kind='rolling', frames=7, mean_dets=6.0, max_dets=20, ncat=3, H=64, windows=16384,
               label='C2 KITTI Car/RRC-shaped rolling windows: 7 frames, D_t~clip(Poisson(6),1,20), F=8 (2d), H=64, K=0, diff; '
                     '1 fwd per frame + 1 bwd per window'),
    'c4': dict(kind='rolling', frames=7, mean_dets=12.0, max_dets=40, ncat=8, H=64, windows=4096,
               label='C4 BDD100K All/libra-shaped rolling windows: 7 frames, D_t~clip(Poisson(12),1,40), F=13 (2d, 8 categories), '
                     'H=64, K=0, diff; 1 fwd per frame + 1 bwd per window'),
    'c5': dict(kind='static', frames=50, dets=300, ncat=3, H=256, iters=4, windows=1,
               label='C5 dense stress: one static window of 50 frames x 300 dets per GPU, F=8, H=256, K=0, diff, 4 MP iterations '
                     '+ 1 bwd'),
}


def static_step(model, plan0, planr, x, targets, iters, opt, bucket, world):
    """The static window mode of SURVEY 8(d): first call with every row new, then iters - 1 empty-x calls on the same graph,
    one backward of the BCE loss over every call's logits, (all-reduce), Adam."""
    from trackmpnn_amd.loss import bce_with_logits_sum
    from trackmpnn_amd.functional import weight_cache
    h = None
    loss = 0.0
    with weight_cache():        # (one set of weight images for the iterations of a step)
        for it in range(iters):
            scores, logits, h, _ = model.forward_graph(x if it == 0 else x[:0], h, plan0 if it == 0 else planr)
            loss = loss + bce_with_logits_sum(logits, targets)
    opt.zero_grad(set_to_none=False)
    loss.backward()
    if world > 1:
        from trackmpnn_amd.dist import allreduce_grads
        allreduce_grads(model, bucket, world)
    opt.step()
    return loss


def make_workload(name, rank, dev, windows=None, frames=None, dets=None):
    """Everything a workload's step needs, resident on `dev`: returns dict(step=callable, edge_iters, model, bucket, plans,
    H, F, config).  Rank r draws its own windows (seed r + 1) -- sequence-level data parallelism, no data-path collective."""
    from trackmpnn_amd import TrackMPNN
    from trackmpnn_amd.dist import GradBucket
    w = dict(WORKLOADS[name])
    H, F = w['H'], w['ncat'] + 5
    torch.manual_seed(5)
    model = TrackMPNN('2d', w['ncat'], H, 0, 'diff').to(dev).train()
    opt = make_adam(model)                                                        # train.py:329
    bucket = GradBucket(model)      # flat gradient storage for every N: p.grad aliases it, one all-reduce when N > 1
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if w['kind'] == 'rolling':
        B = int(windows or w['windows'])
        fr = int(frames or w['frames'])
        plans, xs, edge_iters = build_batch(B, fr, w['mean_dets'], w['max_dets'], F, seed=rank + 1, device=dev)
        gen = torch.Generator().manual_seed(rank)
        targets = [(torch.rand(p.graph.N, 1, generator=gen) < 0.3).float().to(dev) for p in plans]
        cfg = dict(workload=f"{w['label']}; {B} windows/GPU batched block-diagonally (64 distinct seeds tiled)",
                   windows_per_gpu=B, edge_iterations_per_gpu_step=edge_iters, rows_final=plans[-1].graph.N)
        return dict(step=lambda: step(model, plans, xs, targets, opt, bucket, world), edge_iters=edge_iters, model=model,
                    bucket=bucket, plans=plans, H=H, F=F, config=cfg, frames=fr, shape=w)
    from trackmpnn_amd.graph import dense_static_graph, plan_single
    fr, D = int(frames or w['frames']), int(dets or w['dets'])
    g = dense_static_graph(fr, D, 'cpu').to(dev)
    gen = torch.Generator().manual_seed(rank + 1)
    x = torch.zeros(g.N, F, device=dev)
    x[g.det_row.long()] = torch.randn(g.Dn, F, generator=gen).to(dev)
    targets = (torch.rand(g.N, 1, generator=torch.Generator().manual_seed(rank)) < 0.3).float().to(dev)
    plan0, planr = plan_single(g, g.N), plan_single(g, 0)
    iters = w['iters']
    cfg = dict(workload=f"{w['label']} ({fr} x {D}: Dn={g.Dn}, E={g.E})", windows_per_gpu=1,
               edge_iterations_per_gpu_step=g.E * iters, rows_final=g.N)
    return dict(step=lambda: static_step(model, plan0, planr, x, targets, iters, opt, bucket, world), edge_iters=g.E * iters,
                model=model, bucket=bucket, plans=[plan0], H=H, F=F, config=cfg, frames=fr, shape=w)


def timed_steps(fn, steps, warmup, world, dev, setup=0):
    """`setup` + `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    import torch.distributed as dist
    for _ in range(setup):
        fn()
    torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    return dt


def total_over_ranks(v, world, dev):
    if world == 1:
        return float(v)
    import torch.distributed as dist
    tot = torch.tensor([float(v)], device=dev, dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return tot.item()


def allreduce_block(bucket, world, dev, backend):
    """the step's one collective on its own (flat fp32 gradient bucket, SUM), outside the timed region:
    median of 20 all-reduces bracketed by device syncs, max over ranks."""
    import torch.distributed as dist
    ts = []
    for _ in range(22):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t1)
    ar = torch.tensor([sorted(ts[2:])[10]], device=dev, dtype=torch.float64)
    dist.all_reduce(ar, op=dist.ReduceOp.MAX)
    return dict(us=round(ar.item() * 1e6, 1), bytes=int(bucket.flat.numel()) * 4, backend=backend,
                note='one flat-bucket all_reduce(SUM) per optimizer step')


def scaling_extras(args, rank, world, dev):
    """BASELINE.json configs[3] (C4: BDD-shaped sequences data-parallel + gradient all-reduce) and configs[4] (C5: one dense
    window per GPU, 1 -> 8 weak scaling) behind the timed C2 region of an N > 1 run: a short step block each (same bracketing:
    barrier + sync, MAX over ranks, SUM of edges) and the workload's own all-reduce (221 KB / 3.4 MB bucket)."""
    out = {}
    for name, kw, steps in (('c4', dict(windows=args.extras_c4_windows), 3),
                            ('c5', dict(frames=args.extras_c5_frames, dets=args.extras_c5_dets), 2)):
        try:
            wl, err = None, None
            try:
                wl = make_workload(name, rank, dev, **kw)          # (no collective inside: a rank may fail here alone, e.g. out of memory)
            except Exception as e:                                 # noqa: BLE001
                err = e
            # every rank learns whether ALL ranks built the workload before the first collective of its steps is issued
            import torch.distributed as dist
            ok = torch.tensor([0.0 if wl is None else 1.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0:
                raise err if err is not None else RuntimeError('another rank could not build the workload')
            dt = timed_steps(wl['step'], steps, 1, world, dev, setup=1)
            edges = total_over_ranks(wl['edge_iters'], world, dev)
            out[name] = dict(value=edges * steps / dt, unit='graph-edges/s', ms_per_step=dt / steps * 1e3, steps=steps, warmup=1,
                             n_gpus=world, scaling='weak', config=dict(wl['config'], parallelism=f'sequence-dp{world}'),
                             allreduce=allreduce_block(wl['bucket'], world, dev, args.backend),
                             mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
        except Exception as e:                                # noqa: BLE001  (a reporting extra must not sink the bench line)
            out[name] = dict(error=f'{type(e).__name__}: {e}'[:300])
        wl = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c2',
                    help='BASELINE.json config timed as the main line: c2 = configs[1] (headline, default), c4 = configs[3], '
                         'c5 = configs[4]')
    ap.add_argument('--windows', type=int, default=None, help='tracking windows per GPU per step (rolling workloads; default '
                    '16384 for c2, 4096 for c4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stage-profile', action='store_true')
    ap.add_argument('--no-latency', action='store_true', help='skip the batch-1 latency block')
    ap.add_argument('--no-loops', action='store_true', help='skip the loop_batch1 block (train chunk / inference loop)')
    ap.add_argument('--no-scaling-extras', action='store_true', help='N > 1: skip the C4 / C5 blocks behind the C2 region')
    ap.add_argument('--extras-c4-windows', type=int, default=4096, help='windows per GPU of the C4 extras block')
    ap.add_argument('--extras-c5-frames', type=int, default=50, help='frames of the C5 window (extras block and --workload c5)')
    ap.add_argument('--extras-c5-dets', type=int, default=300, help='dets per frame of the C5 window')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse '
                    'the N > 1 code path with several ranks sharing one GPU)')
    ap.add_argument('--single-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with python -m torch.distributed.run --nproc-per-node N for --gpus N > 1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import __graft_entry__
    if rank == 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):       # (stdout carries the ONE JSON line of the contract, nothing else)
            __graft_entry__.build()
    if world > 1:
        dist.barrier()

    wl = make_workload(args.workload, rank, dev, windows=args.windows,
                       frames=args.extras_c5_frames if args.workload == 'c5' else None,
                       dets=args.extras_c5_dets if args.workload == 'c5' else None)
    model, bucket, plans, H, F, edge_iters = wl['model'], wl['bucket'], wl['plans'], wl['H'], wl['F'], wl['edge_iters']
    frames, mean_dets, max_dets = wl['frames'], wl['shape'].get('mean_dets'), wl['shape'].get('max_dets')
    # SETUP_STEPS untimed steps BEFORE the W warm-up steps of the contract (reported as `setup_steps`): one pass over the batch
    # builds the per-graph caches (edge tiles, index records) and brings a fresh box's clocks and allocator up (a first process
    # on a cold box was seen at 36 ms per step with W = 2 where every later run gave 29)
    SETUP_STEPS = 2
    dt = timed_steps(wl['step'], args.steps, args.warmup, world, dev, setup=SETUP_STEPS)
    cfg_main = wl['config']
    total_edges = total_over_ranks(edge_iters, world, dev)
    value = total_edges * args.steps / dt

    roofline = None
    extra = {}
    if world > 1:
        extra['allreduce'] = allreduce_block(bucket, world, dev, args.backend)
    if rank == 0 and not args.no_stage_profile and wl['shape']['kind'] == 'static':
        roofline, extra['stage_roofs'] = wide_stage_profile(plans[-1].graph, H)
    elif rank == 0 and not args.no_stage_profile:
        t, flops, nbytes = stage_profile(model, plans[-1], H)
        # the dominant kernel AMONG THOSE THE STEP RUNS: with the one-pass backward (default) the two stand-alone
        # backward kernels are measured for comparison only
        from trackmpnn_amd import functional as _F
        one_pass = bool(_F.FUSED_BWD) and 'gru_bwd_one_edge' in t
        on_step = ('gru_fwd_edge', 'gru_bwd_one_edge') if one_pass else \
            ('gru_fwd_edge', 'gru_bwd_data_edge_folded', 'gru_bwd_weights_edge')
        dom = max(on_step, key=lambda k: t[k])
        # the box's own device-copy rate (quote the measured peak next to the 8 TB/s spec): a 2 GiB -> 2 GiB
        # torch copy (read + write counted) and a fill (write only), HIP events
        try:
            src_ = torch.empty(1 << 29, dtype=torch.float32, device=dev)
            dst_ = torch.empty_like(src_)
            t_copy = time_stage(lambda: dst_.copy_(src_), iters=5)
            t_fill = time_stage(lambda: dst_.fill_(1.0), iters=5)
            extra['hbm_measured'] = dict(copy_GBs=round(2 * src_.numel() * 4 / (t_copy * 1e-3) / 1e9, 0),
                                         fill_GBs=round(src_.numel() * 4 / (t_fill * 1e-3) / 1e9, 0),
                                         note='2 GiB device copy (read + write) / fill, same box, this run')
            del src_, dst_
        except RuntimeError:
            pass
        # the dominant kernel against BOTH roofs; the one it sits closer to is reported as its bound.  Matrix-pipe
        # time: f32-equivalent flops at the f32-input MFMA rate, or 6 bf16 MFMAs (1/16 of the f32 cost each) per
        # f32 MFMA of work on the split path.
        tf = flops[dom] / (t[dom] * 1e-3) / 1e12
        gbs = nbytes[dom] / (t[dom] * 1e-3) / 1e9
        pipe_frac = tf / MFMA_F32_PEAK_TF * (6.0 / 16.0 if split_enabled() else 1.0)
        hbm_frac = gbs / HBM_PEAK_GBS
        if hbm_frac >= pipe_frac:
            traffic, traffic_source = pmc_traffic(dom, plans[-1].graph.E)
            roofline = dict(bound='hbm', kernel=dom, achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=hbm_frac,
                            traffic=traffic, traffic_source=traffic_source, ms=t[dom], algorithmic_bytes=nbytes[dom],
                            matrix_pipe_frac=pipe_frac, f32_equiv_tflops=tf)
            if 'hbm_measured' in extra:        # the same achieved rate against what a plain device copy reaches on this box
                roofline['frac_of_measured_copy'] = round(gbs / extra['hbm_measured']['copy_GBs'], 3)
        else:
            traffic, traffic_source = pmc_traffic(dom, plans[-1].graph.E)
            roofline = dict(bound='mfma', kernel=dom, achieved=tf, peak=MFMA_F32_PEAK_TF, unit='TFLOP/s', frac=pipe_frac,
                            traffic=traffic, traffic_source=traffic_source, ms=t[dom], hbm_frac=hbm_frac)
        extra['stage_roofs'] = {k: dict(ms=round(t[k], 4), GBs=round(nbytes[k] / (t[k] * 1e-3) / 1e9, 1),
                                        hbm_frac=round(nbytes[k] / (t[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                        f32_equiv_tflops=round(flops[k] / (t[k] * 1e-3) / 1e12, 1) if k in flops else None)
                                for k in t}
        agg_b = nbytes['gather_diff'] + nbytes['segsum']
        agg_t = (t['gather_diff'] + t['segsum']) * 1e-3
        tr_s, _ = pmc_traffic('segsum', plans[-1].graph.E)
        tr_g, _ = pmc_traffic('gather_diff', plans[-1].graph.E)
        extra['roofline_aggregation'] = dict(
            bound='hbm', kernel='gather_diff+segsum', achieved=agg_b / agg_t / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
            frac=agg_b / agg_t / 1e9 / HBM_PEAK_GBS,
            traffic=(tr_s + tr_g) if (tr_s is not None and tr_g is not None) else None,
            traffic_segsum=tr_s, traffic_ratio_segsum=(round(tr_s / nbytes['segsum'], 3) if tr_s is not None else None),
            traffic_gather=tr_g, traffic_ratio_gather=(round(tr_g / nbytes['gather_diff'], 3) if tr_g is not None else None),
            gather_GBs=nbytes['gather_diff'] / (t['gather_diff'] * 1e-3) / 1e9,
            segsum_GBs=nbytes['segsum'] / (t['segsum'] * 1e-3) / 1e9)
        extra['stage_ms'] = {k: round(v, 4) for k, v in t.items()}
        # SURVEY 8(a) row G on the same graph (the bench model has no heads: the stage is timed through the C ABI with K = 2,
        # train-mode dropout mask, as tools/att_bench.py): tmpnn_att_fwd / tmpnn_att_bwd against their byte model
        try:
            ta, nba = att_stage_profile(plans[-1].graph, H, 2, True)
            for k in ('att_fwd', 'att_bwd'):
                extra['stage_roofs'][k] = dict(ms=round(ta[k], 4), GBs=round(nba[k] / (ta[k] * 1e-3) / 1e9, 1),
                                               hbm_frac=round(nba[k] / (ta[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                               f32_equiv_tflops=None, heads=2)
        except RuntimeError as e:                                   # noqa: BLE001  (a reporting extra must not sink the line)
            extra['stage_roofs']['att_error'] = str(e)[:200]

        # what ONE edge row costs per forward call + its share of the backward in the staged step (algorithmic bytes of
        # the kernels as the step runs them: edge forward, folded backward-data, backward-weights, row F and its
        # adjoint) against the aggregation-only model of SURVEY 8(d) (16H + 36): saving the gates dominates
        gE = float(plans[-1].graph.E)
        extra['step_bytes_per_edge_iteration'] = dict(
            staged_step=round((nbytes['gru_fwd_edge'] + (nbytes['gru_bwd_one_edge'] if one_pass else
                                                         nbytes['gru_bwd_data_edge_folded'] + nbytes['gru_bwd_weights_edge'])
                               + 2 * nbytes['segsum']) / gE, 1),
            survey_aggregation_model=16 * H + 36)
        extra['backward'] = 'one-pass (tmpnn_gru_bwd_fused)' if one_pass else 'data + weights kernels'
        if one_pass:
            two = t['gru_bwd_data_edge_folded'] + t['gru_bwd_weights_edge']
            extra['roofline_note'] = (
                f"the one-pass backward ({t['gru_bwd_one_edge']:.2f} ms) replaces the data + weights kernels ({two:.2f} ms "
                f"together at {nbytes['gru_bwd_data_edge_folded'] / (t['gru_bwd_data_edge_folded'] * 1e6) / HBM_PEAK_GBS:.2f} / "
                f"{nbytes['gru_bwd_weights_edge'] / (t['gru_bwd_weights_edge'] * 1e6) / HBM_PEAK_GBS:.2f} of the HBM peak): it moves "
                f"{nbytes['gru_bwd_one_edge'] / 1e9:.1f} GB instead of {(nbytes['gru_bwd_data_edge_folded'] + nbytes['gru_bwd_weights_edge']) / 1e9:.1f} "
                'and is bound by instruction issue and the matrix pipe (k_gru_bwd_two: two 256-register waves per SIMD; the '
                'MFMAs alone hold the pipe for ~45 % of the kernel), not by HBM -- DESIGN.md section 4')
        extra['stage_graph'] = dict(N=plans[-1].graph.N, E=plans[-1].graph.E, Dn=plans[-1].graph.Dn)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and wl['shape']['kind'] == 'rolling':
        cpu = cpu_baseline(frames, mean_dets, max_dets, F, H, seed=1)

    lat = None
    if rank == 0 and world == 1 and not args.no_latency and args.workload == 'c2':
        lat = latency_batch1()

    if world > 1 and args.workload == 'c2' and not args.no_scaling_extras:
        # BASELINE.json configs[3] / configs[4] behind the timed C2 region (every rank takes part: barriers + all-reduces)
        wl = model = bucket = None
        plans = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        extra['scaling_extras'] = scaling_extras(args, rank, world, dev)

    if rank == 0:
        out = dict(metric='graph_edges_per_sec_fwd_bwd', value=value, unit='graph-edges/s', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, setup_steps=SETUP_STEPS, ms_per_step=dt / args.steps * 1e3,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   arithmetic=('fp32 in/out; GRU GEMMs as bf16x6 split products on the bf16 matrix pipe, fp32 accumulate, '
                               'error <= the f32 MFMA chain (tools/pilot_split.py)') if split_enabled() else 'fp32 MFMA',
                   config=dict(cfg_main, parallelism=f'sequence-dp{world}'),
                   roofline=roofline, cpu_baseline=cpu)
        out.update(extra)
        out['latency_batch1'] = lat
        out['loop_batch1'] = loop_batch1() if (world == 1 and not args.no_loops and args.workload == 'c2') else None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
