"""One TrackMPNN message-passing call (reference/models/track_mpnn.py:54-75) on the HIP kernels.

`mp_forward` / `mp_backward` sequence the C-ABI stages of include/tmpnn.h on the current HIP
stream; `MPIteration` wraps them in a torch.autograd.Function so that the carried hidden state
`h_in` (BPTT over a chunk, reference/train.py:104-107,132-135), the new-row features `x` and every
parameter receive gradients exactly as in the reference.  torch is used for memory, streams and
index plumbing only -- every flop of the path runs in libtmpnn.so, and there is no fallback.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .graph import CallPlan, dense_seg_plan, edge_tiles, win_plan

ATT_DROPOUT_P = 0.5
# OPT-IN fast path: accumulate parameter gradients straight into existing p.grad buffers (the kernels add into their
# outputs) instead of returning per-call tensors for autograd to add: saves a zero-fill and ~20 small add kernels per
# call.  It bypasses autograd's bookkeeping for the parameters (autograd receives None), so it is only valid for a
# plain `loss.backward()` with no parameter hooks, no DistributedDataParallel wrapper and no torch.autograd.grad();
# `GradBucket(model)` (trackmpnn_amd.dist) turns it on for its module (`module.inplace_param_grads = True`), the
# environment variable TMPNN_INPLACE_GRADS=1 for every module.  Default: off -- real gradients are returned.
INPLACE_GRADS = os.environ.get('TMPNN_INPLACE_GRADS', '0') == '1'
FUSED_BWD = os.environ.get('TMPNN_FUSED_BWD', '1') == '1'     # one-pass cell backward (see mp_backward); its A/B is a test
# wide cells: the det-side branch of the backward on a second stream next to the E-row matrix kernels (tmpnn_wide_gru_bwd_diff_aux)
# (det-side branches of the wide cells on a second stream: worth 1.5-6 ms of a 195-ms C5 step until round 4's single-read segment sum
#  took most of what it hid -- since then the two forms are within the run-to-run spread (4 alternating pairs: 186.8 vs 188.5 ms, a
#  later pair 184.7 vs 183.4); off by default: the simpler form)
WIDE_OVERLAP = os.environ.get('TMPNN_WIDE_OVERLAP', '0') == '1' and os.environ.get('TMPNN_KEEP_VARIANTS', '0') == '1'   # (tmpnn_wide_gru_bwd_diff_aux: comparison builds)
ATT_KMAX = 8                 # heads per tmpnn_att_fwd / _bwd call (include/tmpnn.h)
_SPLIT = os.environ.get('TMPNN_SPLIT', '1')[:1] != '0'        # TMPNN_SPLIT=0: every GEMM on the f32-input MFMA (tested)

# Switches whose A/B has been run and lost (DESIGN sections 4, 11): they are constants of a normal process and only read from the
# environment when TMPNN_KEEP_VARIANTS=1 (the Python side of the -DTMPNN_KEEP_VARIANTS build flag; tools/ set it for comparisons).
_VARIANTS = os.environ.get('TMPNN_KEEP_VARIANTS', '0') == '1'


def _variant(name: str, default: str) -> str:
    return os.environ.get(name, default) if _VARIANTS else default


WIDE_DW = _variant('TMPNN_WIDE_DW', '1') == '1'              # wide cells: dW from the materialised gate gradients
# wide cells: both W_ih products of the backward on the DET side (linearity of the diff message, tmpnn_wide_gru_bwd_diff);
# TMPNN_WIDE_DET=0 keeps the per-edge products (tmpnn_wide_gru_bwd_data + _weights + the message adjoint's segment sum)
WIDE_DET = _variant('TMPNN_WIDE_DET', '1') == '1'
# H = 128 / 256 edge cells as LDS-tiled bf16x6 GEMMs (csrc/wide.hip); TMPNN_WIDE=0 keeps round 1's f32-MFMA kernels
WIDE = _variant('TMPNN_WIDE', '1') != '0' and _SPLIT
# wide cells: forward over edge tiles (projected det rows staged in LDS); TMPNN_WIDE_TILED=0 keeps the per-row gathers
WIDE_TILED = _variant('TMPNN_WIDE_TILED', '1') != '0'
# H <= 64 edge cells: forward over 32-row edge tiles (projected det rows staged in LDS an item ahead); TMPNN_FWD_TILED=0
# keeps the per-row gathers of tmpnn_gru_fwd (xmode 3)
FWD_TILED = _variant('TMPNN_FWD_TILED', '1') != '0' and _SPLIT
# rows per edge tile of the H <= 64 forward: 32 (k_gru_fwd_split_tiled, the default) or 16 (k_gru_fwd_split_t16, sixteen waves
# per CU, stores straight from the accumulators: measured 7 % slower with the gate planes, 5 % faster without)
FWD_TILE_ROWS = 16 if _variant('TMPNN_FWD_TILE_ROWS', '32') == '16' else 32
# input transform in one launch per direction where every window adds few det rows (csrc/intf.hip); TMPNN_INPUT_TF=0 keeps
# the staged launches of tmpnn_input_bn_*
INPUT_TF = _variant('TMPNN_INPUT_TF', '1') != '0'
# Experiment ("save h only"): the H <= 64 edge cell's forward does not write its four gate planes; the
# backward runs the forward kernel again into the gate planes (and a scratch state) right before the one-pass backward
# reads them.  Same gradients bit for bit; measured SLOWER (numbers in DESIGN), so off by default.
RECOMPUTE_GATES = _variant('TMPNN_RECOMPUTE_GATES', '0') == '1'
CONCAT_PROJ = _variant('TMPNN_CONCAT_PROJ', '1') != '0'
# the window-owned segment sum (csrc/agg.hip k_segsum_win) on graphs with window labels: bit-equal to the CSR kernel, 0.9 ms
# against its 0.43 ms per 6 M edges on MI355X -- kept opt-in
WIN_SEGSUM = _variant('TMPNN_SEGSUM_WIN', '0') == '1'
WIDE_FUSED_ADJOINT = _variant('TMPNN_WIDE_FUSED_ADJOINT', '1') != '0'
_aux_streams: Dict[torch.device, 'torch.cuda.Stream'] = {}


def _aux_stream(dev) -> Optional[int]:
    """Raw handle of this device's auxiliary stream, or None while the current stream is being captured (a forked stream
    inside a capture does not survive hipStreamEndCapture on this stack) or when the current stream is the auxiliary one."""
    if not WIDE_OVERLAP or torch.cuda.is_current_stream_capturing():
        return None
    key = torch.device(dev)
    s = _aux_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(key)
        _aux_streams[key] = s
    if s.cuda_stream == torch.cuda.current_stream(key).cuda_stream:
        return None
    return s.cuda_stream


def _seg_plan_guard(g, dev, on_aux: bool) -> None:
    """The dense segment sum's partial-row buffer belongs to the graph's cached plan (struct tmpnn_seg_plan.ws): one segment sum
    at a time per plan.  Uses on ONE stream are ordered by the stream; when the stream changes (main <-> auxiliary, or a
    caller that moves the graph to another stream) the new stream first waits for everything the previous user's stream
    has been given so far."""
    plan = g.__dict__.get('_seg_plan')
    if plan is None or torch.cuda.is_current_stream_capturing():
        return
    cur = _aux_streams[torch.device(dev)] if on_aux else torch.cuda.current_stream(dev)
    last = plan.__dict__.get('_last_stream')
    if last is not None and last.cuda_stream != cur.cuda_stream:
        ev = torch.cuda.Event()
        ev.record(last)
        cur.wait_event(ev)
    plan._last_stream = cur


_aux_events: Dict[torch.device, tuple] = {}


def _aux_event_pair(dev):
    """The device's fork / join events for the two-stream forms (created once, under the device, and recorded once so that
    their native handles exist): the C entry points only record and wait on what the caller hands them."""
    key = torch.device(dev)
    ev = _aux_events.get(key)
    if ev is None:
        with torch.cuda.device(key):
            ev = (torch.cuda.Event(enable_timing=False), torch.cuda.Event(enable_timing=False))
            for e in ev:
                e.record(torch.cuda.current_stream(key))
        _aux_events[key] = ev
    return ev


_wide_ws: Dict[torch.device, torch.Tensor] = {}


def _wide_workspace(nbytes: int, dev, slot: int = 0) -> torch.Tensor:
    """The materialised gate gradients of a wide cell's backward (24 H bytes per row: 27 GB at C5) live in ONE
    grow-only buffer per device, reused by every call (all users are ordered on the stream): handing tens of GB back
    and forth through the caching allocator cost up to 200 ms per C5 step in allocator stalls."""
    key = (torch.device(dev), slot)           # slot 0: gate gradients; slot 1: the weight-gradient slabs that read them
    ws = _wide_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes + 16:
        _wide_ws.pop(key, None)
        ws = torch.empty((nbytes // 4 + 4,), dtype=torch.float32, device=dev)
        _wide_ws[key] = ws
    return ws


def _input_tf(plan: CallPlan, H: int, F: int) -> bool:
    """The one-launch input transform serves plans whose windows each add at most 128 det rows (H in {32, 64})."""
    return (INPUT_TF and plan.max_seg_nd >= 0 and plan.seg_of_det is not None
            and bool(_lib.load().tmpnn_input_tf_supported(H, F, plan.max_seg_nd)))


_weight_images = None      # inside weight_cache(): {key: (source tensors kept alive, derived tensor)}


@contextlib.contextmanager
def weight_cache():
    """Within the context a weight's derived images (the transposes, the wide cells' MFMA operand images) are built once
    per source tensor instead of once per forward call -- for callers that KNOW the weights do not change inside: the
    forward calls of one training step (CapturedWindow: 4-6 launches less per call of a window).  Nothing outlives the
    context, so nothing can go stale; outside it every forward call rebuilds them."""
    global _weight_images
    prev = _weight_images
    _weight_images = {}
    try:
        yield
    finally:
        _weight_images = prev


def _cached(key, sources, build):
    if _weight_images is None:
        return build()
    hit = _weight_images.get(key)
    if hit is None:                                  # (the sources stay referenced: their addresses cannot be reused)
        hit = _weight_images[key] = (sources, build())
    return hit[1]


def _wide_prep(w_ih: torch.Tensor, w_hh: torch.Tensor, H: int) -> torch.Tensor:
    """MFMA operand images of one wide cell's weights (tmpnn_wide_prepare: four small launches).  Rebuilt on every
    forward call -- microseconds next to a wide cell's work, and never stale (weight_cache() narrows that to once per
    context); the backward reuses the call's images."""
    def build():
        nb = int(_lib.load().tmpnn_wide_prep_bytes(H, H))
        prep = torch.empty((nb // 4 + 4,), dtype=torch.float32, device=w_ih.device)
        _lib.call('tmpnn_wide_prepare', w_ih.data_ptr(), w_hh.data_ptr(), H, H, prep.data_ptr(), _stream())
        return prep
    return _cached(('wide', w_ih.data_ptr(), w_hh.data_ptr(), H), (w_ih, w_hh), build)


@dataclass(frozen=True)
class ModelSpec:
    groups: Tuple[Tuple[str, int], ...]   # (name, F_g)
    H: int
    K: int
    msg_type: str

    @property
    def G(self) -> int:
        return len(self.groups)

    @property
    def IN_e(self) -> int:
        return 2 * self.H if self.msg_type == 'concat' else self.H

    @property
    def F_total(self) -> int:
        return sum(f for _, f in self.groups)

    def param_names(self) -> List[str]:
        """state_dict names of every nn.Parameter, in the order MPIteration takes them."""
        names = []
        for g in range(self.G):
            t = f'input_transforms.{g}.'
            names += [t + '0.weight', t + '0.bias', t + '1.weight', t + '1.bias', t + '3.weight', t + '3.bias']
        for g in range(self.G):
            f = f'factor_grus.{g}.'
            names += [f + 'edge_gru.weight_ih', f + 'edge_gru.weight_hh', f + 'edge_gru.bias_ih', f + 'edge_gru.bias_hh']
            for k in range(self.K):
                names += [f + f'gat.{k}.W_att', f + f'gat.{k}.a']
            names += [f + 'node_gru.weight_ih', f + 'node_gru.weight_hh', f + 'node_gru.bias_ih', f + 'node_gru.bias_hh']
        names += ['output_transform_node.weight', 'output_transform_node.bias',
                  'output_transform_edge.weight', 'output_transform_edge.bias']
        return names


def _stream() -> int:
    return _lib.raw_stream()


def _keep_bits(keep, K: int, n: int, dev) -> torch.Tensor:
    """The attention dropout mask as the kernels take it: uint8 [n], bit k set = head k keeps the position.  `keep` None:
    drawn here (p = 0.5: every bit of a uniform byte is an independent fair coin -- one launch for all heads); else a
    [K, n] mask (the reference's drawn mask in the parity tests) packed."""
    n = max(n, 1)
    if keep is None:
        if ATT_DROPOUT_P == 0.5:
            return torch.empty((n,), dtype=torch.uint8, device=dev).random_(0, 1 << K)
        keep = torch.empty((K, n), dtype=torch.uint8, device=dev).bernoulli_(1.0 - ATT_DROPOUT_P)
    keep = (keep != 0).to(torch.uint8)
    bits = keep[0].clone()
    for k in range(1, K):
        bits |= keep[k] << k
    return bits.contiguous()


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f'{what} is on {t.device}: trackmpnn_amd runs on the MI355X HIP kernels only '
                           '(no CPU or torch fallback exists)')


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _transpose(w: torch.Tensor) -> torch.Tensor:
    def build():
        out = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float32, device=w.device)
        _lib.call('tmpnn_transpose', w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr(), _stream())
        return out
    return _cached(('t', w.data_ptr(), w.shape[0], w.shape[1]), (w,), build)


def mp_forward(spec: ModelSpec, plan: CallPlan, x: torch.Tensor, h_in: Optional[torch.Tensor],
               P: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], training: bool, save: bool,
               keep: Optional[Sequence[torch.Tensor]] = None, reserve_rows: int = 0, h_spare: int = 0):
    """Returns (scores [N,1], logits [N,1], h_out [N,G*H], alphas, saved).

    reserve_rows > 0 allocates h_out with that many spare rows behind it; the NEXT call, when handed
    that h_out as h_in with n <= spare new rows, appends its new rows in place instead of copying
    the carried state (the caller promises to continue from a given h_out at most once).
    """
    g = plan.graph
    H, G, K = spec.H, spec.G, spec.K
    GH = G * H
    N, E, Dn = g.N, g.E, g.Dn
    n = plan.n_new
    N_old = N - n
    dev = x.device
    st = _stream()
    if h_in is None:
        if N_old != 0:
            raise ValueError(f'h_in is None but the graph has {N_old} rows that are not new')
    else:
        if h_in.shape[0] != N_old or h_in.shape[1] != GH:
            raise ValueError(f'h_in must be [{N_old}, {GH}] (N - n, G*H), got {tuple(h_in.shape)}')
    if x.shape[0] != n or (n > 0 and x.shape[1] != spec.F_total):
        raise ValueError(f'x must be [{n}, {spec.F_total}], got {tuple(x.shape)}')

    opts = dict(dtype=torch.float32, device=dev)
    h_cat = None
    if h_in is not None and N_old > 0 and n > 0 and h_in.is_contiguous():
        st_ = h_in.untyped_storage()
        if h_spare >= n and st_.nbytes() >= 4 * (h_in.storage_offset() + N * GH):
            h_cat = torch.empty(0, **opts).set_(st_, h_in.storage_offset(), (N, GH), (GH, 1))   # append in place
    if h_cat is None:
        if h_in is not None and n == 0:
            h_cat = h_in                               # pure extra iteration: nothing to append
        else:
            h_cat = torch.empty((N, GH), **opts)
            if N_old > 0:
                h_cat[:N_old].copy_(h_in)
    saved = dict(n=n)
    if n > 0:
        h_cat[N_old:].zero_()                       # new edge rows start at 0 (track_mpnn.py:61)
        nd = int(plan.new_det_row.numel())
        S = plan.S
        if training and plan.min_seg_cnt <= 1:
            # torch.nn.functional.batch_norm refuses a single row in training mode; so does the reference
            raise ValueError('Expected more than 1 value per channel when training, got input size '
                             f'[1, {H}]')
        # the det rows of x: gathered here for the staged transform; the one-launch transform reads x through the row list
        # (tmpnn_input_tf_*'s x_rows) -- no gather launch, no compact copy
        tf_all = nd > 0 and all(_input_tf(plan, H, F) for _, F in spec.groups)
        if tf_all:
            xsrc, xrows = _f32c(x.detach()), plan.new_det_local
            if xrows.dtype != torch.int64 or not xrows.is_contiguous():
                xrows = xrows.long().contiguous()
            xdet = None
        else:
            xdet = _f32c(x.detach().index_select(0, plan.new_det_local)) if nd > 0 else torch.empty((0, spec.F_total), **opts)
            xsrc, xrows = xdet, None
        ws_a = torch.empty((max(nd, 1), H), **opts)
        y_saves, means, rstds = [], [], []
        f0 = 0
        for gi, (_, F) in enumerate(spec.groups):
            t = f'input_transforms.{gi}.'
            y_save = torch.empty((max(nd, 1), H), **opts)
            SS = S if training else 1
            mean = torch.empty((SS, H), **opts)
            rstd = torch.empty((SS, H), **opts)
            if _input_tf(plan, H, F):
                _lib.call('tmpnn_input_tf_fwd', xsrc.data_ptr() + 4 * f0, _lib.ptr(xrows), spec.F_total, F, nd,
                          plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(), _lib.ptr(plan.seg_of_det), S, plan.max_seg_nd, H,
                          int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                          P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(),
                          buffers[t + '1.running_mean'].data_ptr(), buffers[t + '1.running_var'].data_ptr(),
                          P[t + '3.weight'].data_ptr(), P[t + '3.bias'].data_ptr(),
                          y_save.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          plan.new_det_row.data_ptr(), h_cat.data_ptr() + 4 * gi * H, GH, st)
            else:
                _lib.call('tmpnn_input_bn_fwd', xdet.data_ptr() + 4 * f0, spec.F_total, F, nd,
                          plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(), _lib.ptr(plan.seg_of_det), S, H, int(training),
                          P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                          P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(),
                          buffers[t + '1.running_mean'].data_ptr(), buffers[t + '1.running_var'].data_ptr(),
                          P[t + '3.weight'].data_ptr(), P[t + '3.bias'].data_ptr(),
                          y_save.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws_a.data_ptr(),
                          plan.new_det_row.data_ptr(), h_cat.data_ptr() + 4 * gi * H, GH, st)
            if training:
                buffers[t + '1.num_batches_tracked'] += S
            y_saves.append(y_save)
            means.append(mean)
            rstds.append(rstd)
            f0 += F
        if save:
            saved.update(xdet=xsrc, xrows=xrows, y_save=y_saves, mean=means, rstd=rstds)

    spare = max(int(reserve_rows), 0)
    if spare > 0:      # plain (non-view) tensor over a larger storage: the next call may extend it in place
        buf = torch.empty(((N + spare) * GH,), **opts)
        h_out = torch.empty(0, **opts).set_(buf.untyped_storage(), 0, (N, GH), (GH, 1))
    else:
        h_out = torch.empty((N, GH), **opts)
    gates = torch.empty((G, 4, N, H), **opts) if save else None
    es_all = torch.empty((G, max(Dn, 1), H), **opts)
    alphas: List[Optional[List[torch.Tensor]]] = []
    att_saved = []
    xmode = 2 if spec.msg_type == 'concat' else 1
    plane = N * H
    lib = _lib.load()
    use_proj = spec.msg_type == 'diff' and H <= 64 and g.src_pos is not None and Dn > 0
    # concat: [h_src | h_dst] W_ih^T = P1[src] + P2[dst] -- the same tiled kernel on a stacked table [P1; -P2] and tile lists
    # whose dst entries are offset by Dn (TMPNN_CONCAT_PROJ=0 keeps the per-edge GEMM over IN = 2H)
    use_proj_cat = (CONCAT_PROJ and spec.msg_type == 'concat' and H <= 64 and FWD_TILED and g.src_pos is not None
                    and Dn > 0 and E > 0)
    use_wide = (WIDE and spec.msg_type == 'diff' and H >= 128 and bool(lib.tmpnn_wide_supported(H, H))
                and g.src_pos is not None and Dn > 0 and E > 0)
    if use_wide and H % 256 == 0 and K == 0:
        # dense scenes: the plan of the single-read segment sum rides on the graph's C struct (tmpnn_segsum_fwd here and inside
        # the wide backward take it on 256-column blocks); None for ragged graphs
        dense_seg_plan(g)
    if WIN_SEGSUM and H == 64 and E > 0:
        # batches of small windows (batch_windows): the plan of the window-owned segment sum rides on the graph's C struct
        # (None for graphs without window labels).  Opt-in: measured slower than the CSR kernel
        win_plan(g)
    wide_preps = []
    # the call's new EDGE rows enter the state as zeros (h_cat[N_old:] zero-filled above, only det rows written since): the
    # segment sum does not read them.  (Measured and dropped in round 6: the tiled edge forward skipping the 72 MFMAs of tiles
    # made of such rows -- bit-equal, 8.10 -> 8.09 ms per step: the matrix pipe is not what an item waits for.)
    zero_from = N_old if n > 0 else N
    # output head fused into the cells' epilogues where the LDS-resident kernel runs (else tmpnn_heads_fwd)
    cw = min(lib.tmpnn_gru_fwd_head_parts(H, H if use_proj_cat else spec.IN_e, 3 if (use_proj or use_proj_cat) else xmode),
             lib.tmpnn_gru_fwd_head_parts(H, H, 0))
    parts = torch.empty((G * cw, N), **opts) if cw > 0 else None
    w_node, w_edge = P['output_transform_node.weight'], P['output_transform_edge.weight']
    for gi in range(G):
        f = f'factor_grus.{gi}.'
        hg = h_cat.data_ptr() + 4 * gi * H
        og = h_out.data_ptr() + 4 * gi * H
        gp = gates[gi].data_ptr() if save else None
        part_g = (parts.data_ptr() + 4 * gi * cw * N) if cw > 0 else None
        we_g = (w_edge.data_ptr() + 4 * gi * H) if cw > 0 else None
        wn_g = (w_node.data_ptr() + 4 * gi * H) if cw > 0 else None
        # edge update: GRU(h[src]-h[dst] | concat, h[e])      (layers.py:90-97)
        # (temporaries stay referenced until their consumer is enqueued: the caching allocator may
        #  hand a freed block to the very next allocation)
        e_wih_t, e_whh_t = _transpose(P[f + 'edge_gru.weight_ih']), _transpose(P[f + 'edge_gru.weight_hh'])
        n_wih_t, n_whh_t = _transpose(P[f + 'node_gru.weight_ih']), _transpose(P[f + 'node_gru.weight_hh'])
        # wide cells without attention: the det-side chain (row F, then the node cell: row movers + a Dn-row cell) goes to the
        # auxiliary stream next to the edge cell's persistent matrix kernel -- both only read h_cat and write disjoint rows of
        # h_out / the gate planes; nothing is allocated under the auxiliary stream
        st_det, aux_obj = st, None
        if use_wide and K == 0:
            aux = _aux_stream(dev)
            if aux is not None:
                aux_obj = _aux_streams[torch.device(dev)]
                fork, _ = _aux_event_pair(dev)
                fork.record(torch.cuda.current_stream(dev))
                aux_obj.wait_event(fork)
                st_det = aux
        if use_wide:
            # H = 128 / 256: LDS-tiled bf16x6 GEMMs, the diff message through the projected det rows (csrc/wide.hip)
            prep = _wide_prep(P[f + 'edge_gru.weight_ih'], P[f + 'edge_gru.weight_hh'], H)
            wide_preps.append(prep)
            proj = torch.empty((Dn, 3 * H), **opts)
            if WIDE_TILED:
                _lib.call('tmpnn_wide_gru_fwd_tiled', prep.data_ptr(), g.det_row.data_ptr(), Dn, edge_tiles(g, 128).cref(), E,
                          hg, GH, H, P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                          proj.data_ptr(), og, GH, gp, plane, st)
            else:
                _lib.call('tmpnn_wide_gru_fwd', prep.data_ptr(), g.det_row.data_ptr(), Dn, g.edge_row.data_ptr(), E,
                          g.src_pos.data_ptr(), g.dst_pos.data_ptr(), hg, GH, H,
                          P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                          proj.data_ptr(), og, GH, gp, plane, st)
        elif use_proj_cat:
            proj = torch.empty((2 * Dn, 3 * H), **opts)
            _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), Dn, hg, GH, H, e_wih_t.data_ptr(), 3 * H,
                      proj.data_ptr(), 3 * H, st)
            # (the kernel forms P[src] - P[dst]: the dst half of the table is projected with -W2^T -- negating H x 3H weights
            #  instead of Dn x 3H projected rows)
            w2n = _cached(('negt', e_wih_t.data_ptr(), H), (e_wih_t,), lambda: e_wih_t[H:].neg().contiguous())
            _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), Dn, hg, GH, H, w2n.data_ptr(), 3 * H,
                      proj.data_ptr() + 4 * Dn * 3 * H, 3 * H, st)
            _lib.call('tmpnn_gru_fwd_tiles', edge_tiles(g, FWD_TILE_ROWS, dst_offset=Dn).cref(), E, proj.data_ptr(), 3 * H, hg, GH,
                      H, e_whh_t.data_ptr(), P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                      og, GH, gp, plane, we_g, part_g, N, st)
        elif use_proj:
            # (h[src]-h[dst]) W_ih^T = P[src] - P[dst] with P = h[dets] W_ih^T: the x-half of the edge cell's
            # forward GEMM runs over the Dn det rows instead of the E edge rows
            proj = torch.empty((Dn, 3 * H), **opts)
            _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), Dn, hg, GH, H, e_wih_t.data_ptr(), 3 * H,
                      proj.data_ptr(), 3 * H, st)
            if FWD_TILED and E > 0:
                recompute = RECOMPUTE_GATES and save
                _lib.call('tmpnn_gru_fwd_tiles', edge_tiles(g, FWD_TILE_ROWS).cref(), E, proj.data_ptr(), 3 * H, hg, GH, H,
                          e_whh_t.data_ptr(), P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                          og, GH, None if recompute else gp, plane, we_g, part_g, N, st)
                if recompute:
                    saved.setdefault('proj', {})[gi] = (proj, e_whh_t)
            else:
                _lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, 3, g.src_pos.data_ptr(), g.dst_pos.data_ptr(),
                          proj.data_ptr(), 3 * H, 0, H, hg, GH, H, None, e_whh_t.data_ptr(),
                          P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                          og, GH, gp, plane, we_g, part_g, N, st)
        else:
            _lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, xmode, g.src.data_ptr(), g.dst.data_ptr(),
                      None, 0, 0, spec.IN_e, hg, GH, H,
                      e_wih_t.data_ptr(), e_whh_t.data_ptr(),
                      P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                      og, GH, gp, plane, we_g, part_g, N, st)
        # edge -> node aggregation                              (layers.py:99-112)
        es = es_all[gi]
        if K == 0:
            _seg_plan_guard(g, dev, aux_obj is not None)
            # (the call's new edge rows are 0 and are not read: rows >= N_old -- h_cat[N_old:] was just zero-filled and only
            #  its det rows written; without new rows every row is read)
            _lib.call('tmpnn_segsum_fwd_live', g.cref(), hg, GH, es.data_ptr(), H, H, 1, zero_from, st_det)
            alphas.append(None)
        else:
            # the kernels take up to ATT_KMAX heads per call (all of them from one read of h[e]); more heads run in groups whose
            # means are combined with their share K_g / K (reference: any number of heads, utils/training_options.py:23)
            erec, inc_other = g.att_index() if E > 0 else (None, None)
            groups, al = [], []
            for k0 in range(0, K, ATT_KMAX):
                Kg = min(ATT_KMAX, K - k0)
                Ws = [P[f + f'gat.{k}.W_att'] for k in range(k0, k0 + Kg)]
                As = [P[f + f'gat.{k}.a'] for k in range(k0, k0 + Kg)]
                # (the heads' weights side by side for the kernels: one copy per call, or per weight_cache() context)
                W = _cached(('attW', tuple(t.data_ptr() for t in Ws)), tuple(Ws), lambda: torch.cat(Ws, 1).contiguous())
                a = _cached(('atta', tuple(t.data_ptr() for t in As)), tuple(As),
                            lambda: torch.stack([t.reshape(-1) for t in As]).contiguous())
                ws_ha = torch.empty((max(Dn, 1), Kg * H), **opts)
                score = torch.empty((max(2 * E, 1), Kg), **opts)  # (k_att_score writes both CSR positions of every edge)
                stats = torch.empty((max(Dn, 1), Kg, 2), **opts)
                esk = torch.empty((Kg, max(Dn, 1), H), **opts)
                alpha = torch.empty((Kg, max(2 * E, 1)), **opts)
                kp = None
                if training:
                    kp = _keep_bits(None if keep is None else keep[gi][k0:k0 + Kg], Kg, 2 * E, dev)
                out_g = es if Kg == K else torch.empty_like(es)
                _lib.call('tmpnn_att_fwd', g.cref(), _lib.ptr(erec), hg, GH, H, Kg, W.data_ptr(),
                          a.data_ptr(), _lib.ptr(kp), ATT_DROPOUT_P, ws_ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
                          esk.data_ptr(), alpha.data_ptr(), out_g.data_ptr(), H, st)
                if Kg != K:
                    if k0 == 0:
                        torch.mul(out_g, Kg / K, out=es)
                    else:
                        es.add_(out_g, alpha=Kg / K)
                al += [alpha[k, :2 * E] for k in range(Kg)]
                groups.append((k0, Kg, W, a, kp, ws_ha, score, stats, esk))
            alphas.append(al)
            att_saved.append(groups)
        # node update: GRU(es, h[d])                            (layers.py:114)
        _lib.call('tmpnn_gru_fwd', g.det_row.data_ptr(), Dn, 0, None, None,
                  es.data_ptr(), H, 1, H, hg, GH, H,
                  n_wih_t.data_ptr(), n_whh_t.data_ptr(),
                  P[f + 'node_gru.bias_ih'].data_ptr(), P[f + 'node_gru.bias_hh'].data_ptr(),
                  og, GH, gp, plane, wn_g, part_g, N, st_det)
        if aux_obj is not None:
            _, join = _aux_event_pair(dev)
            join.record(aux_obj)
            torch.cuda.current_stream(dev).wait_event(join)
    logits = torch.empty((N, 1), **opts)
    scores = torch.empty((N, 1), **opts)
    if cw > 0:
        _lib.call('tmpnn_heads_finish', parts.data_ptr(), N, G * cw, N, g.is_edge.data_ptr(),
                  P['output_transform_node.bias'].data_ptr(), P['output_transform_edge.bias'].data_ptr(),
                  logits.data_ptr(), scores.data_ptr(), st)
    else:
        _lib.call('tmpnn_heads_fwd', h_out.data_ptr(), GH, GH, N, g.is_edge.data_ptr(),
                  P['output_transform_node.weight'].data_ptr(), P['output_transform_node.bias'].data_ptr(),
                  P['output_transform_edge.weight'].data_ptr(), P['output_transform_edge.bias'].data_ptr(),
                  logits.data_ptr(), scores.data_ptr(), st)
    if save:
        saved.update(h_cat=h_cat, gates=gates, es=es_all, att=att_saved, h_out=h_out, scores=scores,
                     wide=wide_preps if use_wide else None)
    return scores, logits, h_out, alphas, saved


def mp_backward(spec: ModelSpec, plan: CallPlan, saved: dict, P: Dict[str, torch.Tensor], training: bool,
                d_scores: Optional[torch.Tensor], d_logits: Optional[torch.Tensor], d_hout: Optional[torch.Tensor],
                need_x: bool, need_h: bool, grad_out=None):
    """Returns (d_x | None, d_h_in | None, {param name: grad})."""
    g = plan.graph
    H, G, K = spec.H, spec.G, spec.K
    GH = G * H
    N, E, Dn = g.N, g.E, g.Dn
    n = saved['n']
    N_old = N - n
    h_cat, gates, es_all, h_out = saved['h_cat'], saved['gates'], saved['es'], saved['h_out']
    dev = h_cat.device
    st = _stream()
    opts = dict(dtype=torch.float32, device=dev)
    # one zero-filled buffer for every parameter gradient of this call (the kernels accumulate with +=)
    names = spec.param_names()
    sizes = [P[nm].numel() for nm in names]
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + ((sz + 63) // 64) * 64)          # 256-byte aligned slices
    if grad_out is not None:
        grads = dict(grad_out)                 # caller-owned accumulators (p.grad): every kernel below adds into them
    else:
        flat = torch.zeros((offs[-1],), **opts)
        grads = {nm: flat[o:o + sz].view(P[nm].shape) for nm, o, sz in zip(names, offs, sizes)}

    # heads (track_mpnn.py:72-75): dy = d_logits + d_scores * s(1-s); its contribution dy * w_type to the
    # gradient of h_out is folded into the GRU backward kernels (never materialised)
    lib = _lib.load()
    dh_up = _f32c(d_hout) if d_hout is not None else None
    dy = None
    if d_scores is not None or d_logits is not None:
        dy = torch.empty((N,), **opts)
        dl = _f32c(d_logits) if d_logits is not None else None
        ds = _f32c(d_scores) if d_scores is not None else None
        # (the kernel takes up to 1024 columns of h_out per call: wider states -- three feature groups above nhidden 256 --
        #  go in column slices; dy is the same in every slice, the two bias gradients are taken from the first)
        for c0 in range(0, GH, 1024):
            cw_ = min(1024, GH - c0)
            ws_b = lib.tmpnn_heads_bwd_ws(N, cw_)
            ws = torch.empty((max(ws_b // 4, 1),), **opts)
            db_n = grads['output_transform_node.bias'] if c0 == 0 else torch.zeros((1,), **opts)
            db_e = grads['output_transform_edge.bias'] if c0 == 0 else torch.zeros((1,), **opts)
            _lib.call('tmpnn_heads_bwd', h_out.data_ptr() + 4 * c0, GH, cw_, N, g.is_edge.data_ptr(),
                      P['output_transform_node.weight'].data_ptr() + 4 * c0, P['output_transform_edge.weight'].data_ptr() + 4 * c0,
                      saved['scores'].data_ptr(), _lib.ptr(dl), _lib.ptr(ds), dy.data_ptr(), None, 0, 0,
                      grads['output_transform_node.weight'].data_ptr() + 4 * c0, db_n.data_ptr(),
                      grads['output_transform_edge.weight'].data_ptr() + 4 * c0, db_e.data_ptr(),
                      ws.data_ptr(), ws_b, st)
    elif dh_up is None:
        dh_up = torch.zeros((N, GH), **opts)

    d_hcat = torch.empty((N, GH), **opts)
    IN_e = spec.IN_e
    dmsg = torch.empty((N, IN_e), **opts)
    xmode = 2 if spec.msg_type == 'concat' else 1
    plane = N * H
    ws_e = lib.tmpnn_gru_bwd_weights_ws(E, IN_e, H)
    ws_n = lib.tmpnn_gru_bwd_weights_ws(Dn, H, H)
    # The one-pass backward (tmpnn_gru_bwd_fused: H = 64, K-independent) reads dh, the gates and h ONCE for the data
    # and the weight gradient: 2.39 vs 2.85 ms per 3 M edge rows for the two stand-alone kernels, 35.3 vs 37.9 ms per
    # C2 step (round 2).  TMPNN_FUSED_BWD=0 keeps the two kernels.
    use_fused_bwd = (FUSED_BWD and lib.tmpnn_gru_bwd_fused_available(H, H, 0)
                     and lib.tmpnn_gru_bwd_fused_available(H, IN_e, xmode))
    ws_f = max(lib.tmpnn_gru_bwd_fused_ws(E, IN_e, H), lib.tmpnn_gru_bwd_fused_ws(Dn, H, H)) if use_fused_bwd else 0
    ws_w = torch.empty((max(ws_e, ws_n, ws_f) // 4 + 1,), **opts)
    w_node, w_edge = P['output_transform_node.weight'], P['output_transform_edge.weight']
    for gi in range(G):
        f = f'factor_grus.{gi}.'
        hg = h_cat.data_ptr() + 4 * gi * H
        dog = (dh_up.data_ptr() + 4 * gi * H) if dh_up is not None else None
        dhg = d_hcat.data_ptr() + 4 * gi * H
        gp = gates[gi].data_ptr()
        es = es_all[gi]
        dyp = _lib.ptr(dy)
        wn = (w_node.data_ptr() + 4 * gi * H) if dy is not None else None
        we = (w_edge.data_ptr() + 4 * gi * H) if dy is not None else None
        fuse = K == 0
        if use_fused_bwd:
            # one pass over the gates per cell: data and weight gradients together
            _lib.call('tmpnn_gru_bwd_fused', g.det_row.data_ptr(), Dn, 0, None, None, es.data_ptr(), H, 1, H,
                      hg, GH, H, P[f + 'node_gru.weight_ih'].data_ptr(), P[f + 'node_gru.weight_hh'].data_ptr(),
                      gp, plane, dog, GH, dyp, wn, dmsg.data_ptr(), IN_e, dhg, GH, None, None, None, 0,
                      grads[f + 'node_gru.weight_ih'].data_ptr(), grads[f + 'node_gru.weight_hh'].data_ptr(),
                      grads[f + 'node_gru.bias_ih'].data_ptr(), grads[f + 'node_gru.bias_hh'].data_ptr(),
                      ws_w.data_ptr(), ws_w.numel() * 4, st)
            if saved.get('proj') and gi in saved['proj']:
                # (RECOMPUTE_GATES) the edge rows of the gate planes, formed again from the saved state
                proj_s, whh_t_s = saved['proj'][gi]
                h_scr = _wide_workspace(4 * N * GH, dev, slot=2)
                _lib.call('tmpnn_gru_fwd_tiles', edge_tiles(g, FWD_TILE_ROWS).cref(), E, proj_s.data_ptr(), 3 * H, hg, GH, H,
                          whh_t_s.data_ptr(), P[f + 'edge_gru.bias_ih'].data_ptr(), P[f + 'edge_gru.bias_hh'].data_ptr(),
                          h_scr.data_ptr() + 4 * gi * H, GH, gp, plane, None, None, 0, st)
            _lib.call('tmpnn_gru_bwd_fused', g.edge_row.data_ptr(), E, xmode, g.src.data_ptr(), g.dst.data_ptr(),
                      None, 0, 0, IN_e, hg, GH, H,
                      P[f + 'edge_gru.weight_ih'].data_ptr(), P[f + 'edge_gru.weight_hh'].data_ptr(),
                      gp, plane, dog, GH, dyp, we, dmsg.data_ptr(), IN_e, dhg, GH,
                      g.src.data_ptr() if fuse else None, g.dst.data_ptr() if fuse else None,
                      dmsg.data_ptr() if fuse else None, IN_e,
                      grads[f + 'edge_gru.weight_ih'].data_ptr(), grads[f + 'edge_gru.weight_hh'].data_ptr(),
                      grads[f + 'edge_gru.bias_ih'].data_ptr(), grads[f + 'edge_gru.bias_hh'].data_ptr(),
                      ws_w.data_ptr(), ws_w.numel() * 4, st)
        else:
            # node GRU backward: d_es -> dmsg[det rows, 0:H], d_hcat[det rows]
            _lib.call('tmpnn_gru_bwd_data', g.det_row.data_ptr(), Dn, H, hg, GH, H,
                      P[f + 'node_gru.weight_ih'].data_ptr(), P[f + 'node_gru.weight_hh'].data_ptr(),
                      gp, plane, dog, GH, dyp, wn, dmsg.data_ptr(), IN_e, dhg, GH, None, None, None, 0, st)
            _lib.call('tmpnn_gru_bwd_weights', g.det_row.data_ptr(), Dn, 0, None, None, es.data_ptr(), H, 1, H,
                      hg, GH, H, gp, plane, dog, GH, dyp, wn,
                      grads[f + 'node_gru.weight_ih'].data_ptr(), grads[f + 'node_gru.weight_hh'].data_ptr(),
                      grads[f + 'node_gru.bias_ih'].data_ptr(), grads[f + 'node_gru.bias_hh'].data_ptr(),
                      ws_w.data_ptr(), ws_w.numel() * 4, st)
            # edge GRU backward: d_ns -> dmsg[edge rows, 0:IN_e], d_hcat[edge rows]; without attention the
            # adjoint of the edge -> node sum (d_es[src] - d_es[dst], read from dmsg's det rows) rides along
            wide_det = bool(saved.get('wide')) and WIDE_DET
            if wide_det:
                # whole edge-cell backward in one call; the message adjoint lands on d_hcat's det rows directly
                wsb = int(lib.tmpnn_wide_gru_bwd_diff_ws(N, E, Dn, H))
                ws_wide = _wide_workspace(wsb, dev)
                aux = _aux_stream(dev)
                args = (saved['wide'][gi].data_ptr(), g.cref(), hg, GH, H, gp, plane, dog, GH, dyp, we, dhg, GH,
                        grads[f + 'edge_gru.weight_ih'].data_ptr(), grads[f + 'edge_gru.weight_hh'].data_ptr(),
                        grads[f + 'edge_gru.bias_ih'].data_ptr(), grads[f + 'edge_gru.bias_hh'].data_ptr(),
                        ws_wide.data_ptr(), wsb, st)
                # (every buffer the auxiliary stream touches was allocated on, and is next used on, the current stream,
                #  which the call leaves waiting for the auxiliary work: no record_stream needed)
                evf = evj = None
                if aux is not None:
                    ef, ej = _aux_event_pair(dev)
                    evf, evj = ef.cuda_event, ej.cuda_event
                    if not evf or not evj:                     # (no native handle: one stream)
                        aux = evf = evj = None
                _seg_plan_guard(g, dev, aux is not None)       # (the three d_gi segment sums run on `aux` when it is given)
                if fuse and WIDE_FUSED_ADJOINT:
                    # the adjoint of the edge -> node sum rides in the epilogue of the E-row product (no separate pass over d_h)
                    _lib.call('tmpnn_wide_gru_bwd_diff_fused', *args[:-1], dmsg.data_ptr(), IN_e, st, aux, evf, evj)
                else:
                    if aux is not None:
                        _lib.call('tmpnn_wide_gru_bwd_diff_aux', *args, aux, evf, evj)
                    else:
                        _lib.call('tmpnn_wide_gru_bwd_diff', *args)
                    if fuse:
                        _lib.call('tmpnn_gather_diff_fwd', g.cref(), dmsg.data_ptr(), IN_e, dhg, GH, H, 1, st)
            elif saved.get('wide'):
                wsb = int(lib.tmpnn_wide_gru_bwd_data_ws(E, H))
                ws_wide = _wide_workspace(wsb, dev)
                _lib.call('tmpnn_wide_gru_bwd_data', saved['wide'][gi].data_ptr(), g.edge_row.data_ptr(), E, hg, GH, H,
                          gp, plane, dog, GH, dyp, we, dmsg.data_ptr(), IN_e, dhg, GH, ws_wide.data_ptr(), wsb, st)
                if fuse:
                    _lib.call('tmpnn_gather_diff_fwd', g.cref(), dmsg.data_ptr(), IN_e, dhg, GH, H, 1, st)
            else:
              _lib.call('tmpnn_gru_bwd_data', g.edge_row.data_ptr(), E, IN_e, hg, GH, H,
                      P[f + 'edge_gru.weight_ih'].data_ptr(), P[f + 'edge_gru.weight_hh'].data_ptr(),
                      gp, plane, dog, GH, dyp, we, dmsg.data_ptr(), IN_e, dhg, GH,
                      g.src.data_ptr() if fuse else None, g.dst.data_ptr() if fuse else None,
                      dmsg.data_ptr() if fuse else None, IN_e, st)
            if wide_det:
                pass
            elif saved.get('wide') and WIDE_DW:
                # ... and the weight gradient from the gate gradients that call left in its workspace
                ws2b = int(lib.tmpnn_wide_gru_bwd_weights_ws(E, H))
                ws2 = _wide_workspace(ws2b, dev, slot=1)
                _lib.call('tmpnn_wide_gru_bwd_weights', ws_wide.data_ptr(), g.edge_row.data_ptr(), E, g.src.data_ptr(),
                          g.dst.data_ptr(), hg, GH, H,
                          grads[f + 'edge_gru.weight_ih'].data_ptr(), grads[f + 'edge_gru.weight_hh'].data_ptr(),
                          grads[f + 'edge_gru.bias_ih'].data_ptr(), grads[f + 'edge_gru.bias_hh'].data_ptr(),
                          ws2.data_ptr(), ws2b, st)
            else:
                _lib.call('tmpnn_gru_bwd_weights', g.edge_row.data_ptr(), E, xmode, g.src.data_ptr(), g.dst.data_ptr(),
                          None, 0, 0, IN_e, hg, GH, H, gp, plane, dog, GH, dyp, we,
                          grads[f + 'edge_gru.weight_ih'].data_ptr(), grads[f + 'edge_gru.weight_hh'].data_ptr(),
                          grads[f + 'edge_gru.bias_ih'].data_ptr(), grads[f + 'edge_gru.bias_hh'].data_ptr(),
                          ws_w.data_ptr(), ws_w.numel() * 4, st)
        if K > 0:
            erec, inc_other = g.att_index() if E > 0 else (None, None)
            for k0, Kg, W, a, kp, ws_ha, score, stats, esk in saved['att'][gi]:
                ws_n_att = lib.tmpnn_att_bwd_ws(E, Dn, H, Kg)
                ws_att = torch.empty((max(ws_n_att, 1),), **opts)
                if Kg == K:
                    d_out, ld_dout = dmsg.data_ptr(), IN_e
                else:           # this group's share of the head mean: d es / d es_group = K_g / K
                    d_es_g = torch.zeros((N, H), **opts)
                    d_es_g[g.det_row.long()] = dmsg[g.det_row.long(), :H] * (Kg / K)
                    d_out, ld_dout = d_es_g.data_ptr(), H
                gW = [grads.get(f + f'gat.{k}.W_att') for k in range(k0, k0 + Kg)] if grad_out is not None else []
                ga = [grads.get(f + f'gat.{k}.a') for k in range(k0, k0 + Kg)] if grad_out is not None else []
                args = (g.cref(), _lib.ptr(erec), _lib.ptr(inc_other), hg, GH, H, Kg, W.data_ptr(),
                        a.data_ptr(), _lib.ptr(kp), ATT_DROPOUT_P, ws_ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
                        esk.data_ptr(), d_out, ld_dout, ws_att.data_ptr(), ws_att.numel(), dhg, GH)
                if grad_out is not None and all(t is not None and t.is_contiguous() for t in gW + ga):
                    # in-place mode: the heads' gradient buffers are accumulated directly (no stacked temporary, no adds)
                    pW = (ctypes.c_void_p * Kg)(*[t.data_ptr() for t in gW])
                    pa = (ctypes.c_void_p * Kg)(*[t.data_ptr() for t in ga])
                    _lib.call('tmpnn_att_bwd_heads', *args, ctypes.cast(pW, ctypes.c_void_p), ctypes.cast(pa, ctypes.c_void_p), st)
                else:
                    dW = torch.zeros((Kg, H, H), **opts)
                    da = torch.zeros_like(a)
                    _lib.call('tmpnn_att_bwd', *args, dW.data_ptr(), da.data_ptr(), st)
                    for k in range(Kg):
                        if grad_out is not None:
                            grads[f + f'gat.{k0 + k}.W_att'].add_(dW[k])
                            grads[f + f'gat.{k0 + k}.a'].add_(da[k].reshape(-1, 1))
                        else:
                            grads[f + f'gat.{k0 + k}.W_att'] = dW[k]
                            grads[f + f'gat.{k0 + k}.a'] = da[k].reshape(-1, 1)
        # adjoint of the node -> edge message: into d_hcat[det rows] (the det-side wide backward has already put it there)
        name = 'tmpnn_gather_concat_bwd' if spec.msg_type == 'concat' else 'tmpnn_gather_diff_bwd'
        if not (not use_fused_bwd and saved.get('wide') and WIDE_DET):
            _lib.call(name, g.cref(), dmsg.data_ptr(), IN_e, dhg, GH, H, 1, st)

    d_x = None
    if n > 0:
        nd = int(plan.new_det_row.numel())
        S = plan.S
        xdet = saved['xdet']
        Ft = spec.F_total
        d_xdet = torch.empty((max(nd, 1), Ft), **opts) if need_x else None
        d_xzero = torch.empty((max(S, 1), Ft), **opts) if need_x else None
        f0 = 0
        for gi, (_, F) in enumerate(spec.groups):
            t = f'input_transforms.{gi}.'
            # d_xzero is [S][F] per group: write into a per-group buffer, then place it
            dz_g = torch.empty((max(S, 1), F), **opts) if need_x else None
            if _input_tf(plan, H, F):
                wsb = int(lib.tmpnn_input_tf_bwd_ws(nd, S, H, F, int(training)))
                ws = torch.empty((wsb // 4 + 1,), **opts)
                _lib.call('tmpnn_input_tf_bwd', xdet.data_ptr() + 4 * f0, _lib.ptr(saved.get('xrows')), Ft, F, nd,
                          plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(), _lib.ptr(plan.seg_of_det), S, plan.max_seg_nd, H,
                          int(training), P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                          P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(), P[t + '3.weight'].data_ptr(),
                          saved['y_save'][gi].data_ptr(), saved['mean'][gi].data_ptr(), saved['rstd'][gi].data_ptr(),
                          plan.new_det_row.data_ptr(), d_hcat.data_ptr() + 4 * gi * H, GH,
                          (d_xdet.data_ptr() + 4 * f0) if need_x else None, Ft, _lib.ptr(dz_g),
                          grads[t + '0.weight'].data_ptr(), grads[t + '0.bias'].data_ptr(),
                          grads[t + '1.weight'].data_ptr(), grads[t + '1.bias'].data_ptr(),
                          grads[t + '3.weight'].data_ptr(), grads[t + '3.bias'].data_ptr(),
                          ws.data_ptr(), wsb, st)
                if need_x:
                    d_xzero[:, f0:f0 + F] = dz_g
                f0 += F
                continue
            wsn = lib.tmpnn_input_bn_bwd_ws(nd, S, H, F)
            ws = torch.empty((max(wsn, 1),), **opts)
            _lib.call('tmpnn_input_bn_bwd', xdet.data_ptr() + 4 * f0, Ft, F, nd,
                      plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(), _lib.ptr(plan.seg_of_det), S, H, int(training),
                      P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                      P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(), P[t + '3.weight'].data_ptr(),
                      saved['y_save'][gi].data_ptr(), saved['mean'][gi].data_ptr(), saved['rstd'][gi].data_ptr(),
                      plan.new_det_row.data_ptr(), d_hcat.data_ptr() + 4 * gi * H, GH,
                      (d_xdet.data_ptr() + 4 * f0) if need_x else None, Ft, _lib.ptr(dz_g),
                      grads[t + '0.weight'].data_ptr(), grads[t + '0.bias'].data_ptr(),
                      grads[t + '1.weight'].data_ptr(), grads[t + '1.bias'].data_ptr(),
                      grads[t + '3.weight'].data_ptr(), grads[t + '3.bias'].data_ptr(),
                      ws.data_ptr(), ws.numel(), st)
            if need_x:
                d_xzero[:, f0:f0 + F] = dz_g
            f0 += F
        if need_x:
            # all-zero (edge) rows get the gradient that flows through their segment's batch statistics
            d_x = d_xzero[:S].index_select(0, plan.seg_of_new) if S > 0 else torch.zeros((n, Ft), **opts)
            if nd > 0:
                d_x.index_copy_(0, plan.new_det_local, d_xdet[:nd])
    elif need_x:
        d_x = torch.zeros((0, spec.F_total), **opts)
    d_h_in = d_hcat[:N_old] if (need_h and N_old > 0) else None
    return d_x, d_h_in, grads


class MPIteration(torch.autograd.Function):
    """autograd node of one forward call.  Inputs: (ctx_obj, x, h_in_or_None, *params)."""

    @staticmethod
    def forward(ctx, call, x, h_in, *params):
        spec: ModelSpec = call['spec']
        names = spec.param_names()
        P = {nm: _f32c(p.detach()) for nm, p in zip(names, params)}
        for nm, p in P.items():
            _require_device(p, nm)
        need_grad = call['need_grad']
        scores, logits, h_out, alphas, saved = mp_forward(
            spec, call['plan'], x.detach(), None if h_in is None else _f32c(h_in.detach()), P, call['buffers'],
            call['training'], need_grad, call.get('keep'), call.get('reserve', 0), call.get('h_spare', 0))
        call['alphas'] = alphas
        ctx.call = call
        ctx.saved = saved
        ctx.P = P
        ctx.has_h = h_in is not None
        ctx.set_materialize_grads(False)
        return scores, logits, h_out

    @staticmethod
    def backward(ctx, d_scores, d_logits, d_hout):
        call = ctx.call
        spec: ModelSpec = call['spec']
        need = ctx.needs_input_grad
        names = spec.param_names()
        objs = call.get('param_objs')
        grad_out = None
        if (INPLACE_GRADS or call.get('inplace')) and objs is not None and all(need[3:]):
            gs = [p.grad for p in objs]
            if all(g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == ctx.P[nm].device
                   and g.shape == ctx.P[nm].shape for g, nm in zip(gs, names)):
                grad_out = dict(zip(names, gs))
        d_x, d_h_in, grads = mp_backward(spec, call['plan'], ctx.saved, ctx.P, call['training'],
                                         d_scores, d_logits, d_hout, need_x=need[1],
                                         need_h=ctx.has_h and need[2], grad_out=grad_out)
        ctx.saved = None
        if grad_out is not None:
            return (None, d_x, d_h_in) + (None,) * len(names)      # already added into p.grad
        return (None, d_x, d_h_in) + tuple(grads[nm] for nm in names)
