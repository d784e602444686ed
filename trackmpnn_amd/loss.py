"""Training targets and losses of the reference (models/loss.py) on the HIP kernels -- SURVEY 8(f) row 1.

Same call surface as the reference (`train.py:73-81,112-120`):

    targets = create_targets(labels, node_adj, idx_node)
    loss_c  = CELoss()(logits, targets, node_adj, idx_node)
    loss_f  = FocalLoss(gamma=0)(scores[idx_node, 0], targets[idx_node]) + ...

`node_adj` may also be a `FrameGraph`, `CallPlan` or `DeviceGraph` (e.g. `TrackGraph.graph`), which skips the adjacency conversion; `idx_node`
is accepted for signature compatibility and ignored (the graph knows its det rows).
"""
from __future__ import annotations

from typing import Optional, Union

import weakref

import ctypes

import torch
import torch.nn as nn

from . import _lib
from .graph import CallPlan, DeviceGraph, FrameGraph, graph_from_adjacency

_cache = {}


def _as_graph(adj: Union[torch.Tensor, FrameGraph, CallPlan, DeviceGraph]) -> FrameGraph:
    if isinstance(adj, CallPlan):
        return adj.graph
    if isinstance(adj, FrameGraph):
        return adj
    if isinstance(adj, DeviceGraph):
        return adj.frame_graph()
    # one-entry cache for the three calls of a timestep (create_targets, CELoss, ...) on the same adjacency object;
    # an in-place edit bumps the version counter and misses
    # (a weak reference: the cache does not keep a dense N x N adjacency alive, and a recycled id() cannot alias)
    key = (getattr(adj, '_version', 0), adj.device)
    hit = _cache.get('k')
    if hit is not None and hit[0]() is adj and hit[1] == key:
        return hit[2]
    g = graph_from_adjacency(adj if adj.is_cuda else adj.cuda(), None)
    _cache['k'] = (weakref.ref(adj), key, g)
    return g


def _stream() -> int:
    return _lib.raw_stream()


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f'{what} is on {t.device}: trackmpnn_amd runs on the MI355X HIP kernels only '
                           '(no CPU or torch fallback exists)')


def create_targets(labels: torch.Tensor, node_adj, idx_node=None, as_bytes: bool = False) -> torch.Tensor:
    """reference models/loss.py:8-44.  labels int64 [N] -> targets int64 [N] (as_bytes: uint8 0 / 1, what the loss kernels
    read -- CELoss / FocalLoss take them without another conversion)."""
    _need_cuda(labels, 'labels')
    g = _as_graph(node_adj)
    lab = labels if (labels.dtype == torch.uint8 and labels.is_contiguous()) else (labels != 0).to(torch.uint8).contiguous()
    out = torch.empty_like(lab)
    _lib.call('tmpnn_targets', g.cref(), lab.data_ptr(), out.data_ptr(), _stream())
    return out if as_bytes else out.to(labels.dtype)


def _as_u8(targets: torch.Tensor) -> torch.Tensor:
    """0 / 1 bytes of a target vector (create_targets(..., as_bytes=True) hands them over as they are)."""
    t = targets.reshape(-1)
    if t.dtype == torch.uint8 and t.is_contiguous():
        return t
    return (t != 0).to(torch.uint8).contiguous()


class _CE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets_u8, graph):
        g: FrameGraph = graph
        lg = logits.detach().reshape(-1).float().contiguous()
        stats = torch.empty((max(g.Dn, 1), 2, 4), dtype=torch.float32, device=lg.device)
        loss = torch.empty((1,), dtype=torch.float32, device=lg.device)
        wsn = _lib.load().tmpnn_ce_loss_ws(g.Dn)
        ws = torch.empty((wsn,), dtype=torch.float32, device=lg.device)
        _lib.call('tmpnn_ce_loss_fwd', g.cref(), lg.data_ptr(), targets_u8.data_ptr(), stats.data_ptr(), loss.data_ptr(),
                  ws.data_ptr(), wsn, _stream())
        ctx.g, ctx.lg, ctx.stats, ctx.shape = g, lg, stats, logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        g: FrameGraph = ctx.g
        d = torch.zeros_like(ctx.lg)
        dl = d_loss.reshape(1).float().contiguous()
        _lib.call('tmpnn_ce_loss_bwd', g.cref(), _lib.ptr(g.src_pos), _lib.ptr(g.dst_pos), ctx.lg.data_ptr(),
                  ctx.stats.data_ptr(), dl.data_ptr(), d.data_ptr(), _stream())
        return d.reshape(ctx.shape), None, None


class CELoss(nn.Module):
    """reference models/loss.py:77-115: per det, softmax cross-entropy over its past / future incident edge
    logits against the (last / first) positive target of the set, divided by the set size; summed."""

    def forward(self, outputs, targets, node_adj, idx_node=None):
        _need_cuda(outputs, 'outputs')
        g = _as_graph(node_adj)
        return _CE.apply(outputs, _as_u8(targets), g)


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets_u8, gamma, use_alpha, a0, a1, mean, rows=None):
        # rows (int32, device): the loss runs over outputs[rows] / targets[rows] -- the kernels take the row list, so the
        # reference's `scores[idx, 0]`, `targets[idx]` selections (train.py:76-81) cost no gather and no scatter-add
        s = outputs.detach().reshape(-1).float().contiguous()
        R = s.numel() if rows is None else int(rows.numel())
        ctx.shape = outputs.shape
        ctx.R = R
        if R == 0:
            # empty selection: the reference's loss.mean() is nan and loss.sum() is 0 (models/loss.py:71-74); no kernel
            return s.new_full((), float('nan') if mean else 0.0)
        loss = torch.empty((1,), dtype=torch.float32, device=s.device)
        wsn = _lib.load().tmpnn_focal_loss_ws(R)
        ws = torch.empty((wsn,), dtype=torch.float32, device=s.device)
        if rows is None:
            rows = torch.arange(R, dtype=torch.int32, device=s.device)
        _lib.call('tmpnn_focal_loss_fwd', rows.data_ptr(), R, s.data_ptr(), targets_u8.data_ptr(), float(gamma),
                  int(use_alpha), float(a0), float(a1), loss.data_ptr(), ws.data_ptr(), wsn, _stream())
        ctx.s, ctx.t, ctx.rows, ctx.args = s, targets_u8, rows, (float(gamma), int(use_alpha), float(a0), float(a1))
        ctx.scale = (1.0 / R) if mean else 1.0
        return (loss * ctx.scale).reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        if ctx.R == 0:
            return d_loss.new_zeros(ctx.shape), None, None, None, None, None, None, None
        d = torch.zeros_like(ctx.s)
        dl = d_loss.reshape(1).float().contiguous()
        gamma, ua, a0, a1 = ctx.args
        _lib.call('tmpnn_focal_loss_bwd', ctx.rows.data_ptr(), ctx.R, ctx.s.data_ptr(), ctx.t.data_ptr(), gamma,
                  ua, a0, a1, dl.data_ptr(), float(ctx.scale), d.data_ptr(), _stream())
        return d.reshape(ctx.shape), None, None, None, None, None, None, None


class FocalLoss(nn.Module):
    """reference models/loss.py:47-74 (gamma = 0 is the BCE with eps 1e-10 the reference trains with)."""

    def __init__(self, gamma=0, alpha=None, size_average=True):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha
        if isinstance(alpha, (float, int)):
            self.alpha = [1 - alpha, alpha]
        self.size_average = size_average
        self.eps = 1e-10

    def forward(self, outputs, targets, rows=None):
        """rows=None: the reference's call, `loss(outputs[idx], targets[idx])` on selections made by the caller.
        rows (int32 device tensor): `outputs` / `targets` are the FULL per-row vectors and the loss runs over those rows
        (equal to the former, without the gathers and their scatter-add backward)."""
        _need_cuda(outputs, 'outputs')
        t8 = _as_u8(targets)
        ua = self.alpha is not None
        a0, a1 = (float(self.alpha[0]), float(self.alpha[1])) if ua else (1.0, 1.0)
        if rows is not None:
            if rows.dtype != torch.int32 or not rows.is_contiguous():
                rows = rows.to(torch.int32).contiguous()
            if t8.numel() != outputs.numel():
                raise ValueError('FocalLoss(rows=...): outputs and targets must be the full per-row vectors')
        return _Focal.apply(outputs, t8, self.gamma, ua, a0, a1, self.size_average, rows)


class _BCELogitsSum(torch.autograd.Function):
    """sum_i BCE-with-logits(l_i, t_i) in one launch (+ a one-wave finish); its backward in one (csrc/loss.hip k_bce_*)."""

    @staticmethod
    def forward(ctx, logits, targets):
        l = logits.detach().reshape(-1)
        l = l if (l.dtype == torch.float32 and l.is_contiguous()) else l.float().contiguous()
        t = targets.detach().reshape(-1)
        t = t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        if t.numel() != l.numel():
            raise ValueError(f'bce_with_logits_sum: {l.numel()} logits, {t.numel()} targets')
        n = l.numel()
        loss = torch.empty((1,), dtype=torch.float32, device=l.device)
        wsn = int(_lib.load().tmpnn_bce_logits_ws(n))
        ws = torch.empty((wsn,), dtype=torch.float32, device=l.device)
        _lib.call('tmpnn_bce_logits_sum_fwd', l.data_ptr(), t.data_ptr(), n, loss.data_ptr(), ws.data_ptr(), wsn, _stream())
        ctx.l, ctx.t, ctx.shape = l, t, logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        d = torch.empty_like(ctx.l)
        dl = d_loss.reshape(1)
        dl = dl if (dl.dtype == torch.float32 and dl.is_contiguous()) else dl.float().contiguous()
        _lib.call('tmpnn_bce_logits_sum_bwd', ctx.l.data_ptr(), ctx.t.data_ptr(), ctx.l.numel(), dl.data_ptr(), d.data_ptr(),
                  _stream())
        return d.reshape(ctx.shape), None


def bce_with_logits_sum(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """`torch.nn.functional.binary_cross_entropy_with_logits(logits, targets, reduction='sum')` -- the loss SURVEY 8(d)'s metric
    is quoted with -- as one HIP launch per direction (torch: six element-wise launches and a reduction forward, five
    backward, over a call's millions of logits).  Deterministic summation order; gradient only with respect to the logits."""
    _need_cuda(logits, 'logits')
    return _BCELogitsSum.apply(logits, targets)


class _TrainLosses(torch.autograd.Function):
    """train.py:70-81 / :109-120 for one forward call as ONE autograd node: create_targets, CELoss over the logits and the
    two FocalLoss terms (gamma = 0, no alpha: what train.py constructs) over the scores of the edge rows and -- with the TP
    classifier -- of the det rows, through the same C entry points as the separate modules, with one buffer for every
    workspace and one pass of torch bookkeeping instead of four.  Returns (loss_c, loss_f); values and gradients equal the
    separate calls bit for bit."""

    @staticmethod
    def forward(ctx, logits, scores, labels_u8, graph, tp_classifier):
        g: FrameGraph = graph
        lib = _lib.load()
        dev = logits.device
        st = _stream()
        N, E, Dn = g.N, g.E, g.Dn
        lg = logits.detach().reshape(-1).float().contiguous()
        sc = scores.detach().reshape(-1).float().contiguous()
        targets = torch.empty_like(labels_u8)
        ctx.fused = bool(lib.tmpnn_train_losses_supported(E, Dn))
        if ctx.fused:
            # batch-1 windows: the whole section in ONE launch (csrc/loss.hip k_train_losses_fwd; same values bit for bit)
            nd8 = max(Dn, 1) * 8
            n_ws = int(lib.tmpnn_train_losses_ws(E, Dn))
            buf = torch.empty((nd8 + 4 + n_ws,), dtype=torch.float32, device=dev)
            stats, out = buf[:nd8], buf[nd8:nd8 + 4]
            _lib.call('tmpnn_train_losses_fwd', g.cref(), lg.data_ptr(), sc.data_ptr(), labels_u8.data_ptr(),
                      1 if tp_classifier else 0, targets.data_ptr(), stats.data_ptr(), out.data_ptr(),
                      buf.data_ptr() + 4 * (nd8 + 4), n_ws, st)
            ctx.g, ctx.lg, ctx.sc, ctx.targets, ctx.stats = g, lg, sc, targets, stats
            ctx.tp, ctx.shapes = bool(tp_classifier), (logits.shape, scores.shape)
            return out[0], out[3]
        _lib.call('tmpnn_targets', g.cref(), labels_u8.data_ptr(), targets.data_ptr(), st)
        n_ce, n_fe, n_fd = int(lib.tmpnn_ce_loss_ws(Dn)), int(lib.tmpnn_focal_loss_ws(E)), int(lib.tmpnn_focal_loss_ws(Dn))
        buf = torch.empty((max(Dn, 1) * 8 + 4 + n_ce + n_fe + n_fd,), dtype=torch.float32, device=dev)
        stats = buf[:max(Dn, 1) * 8]
        out = buf[max(Dn, 1) * 8:max(Dn, 1) * 8 + 4]            # loss_c, focal(edge rows) sum, focal(det rows) sum
        o = max(Dn, 1) * 8 + 4
        _lib.call('tmpnn_ce_loss_fwd', g.cref(), lg.data_ptr(), targets.data_ptr(), stats.data_ptr(), out.data_ptr(),
                  buf.data_ptr() + 4 * o, n_ce, st)
        mean_e = mean_d = None
        if E > 0:
            _lib.call('tmpnn_focal_loss_fwd', g.edge_row.data_ptr(), E, sc.data_ptr(), targets.data_ptr(), 0.0, 0, 1.0, 1.0,
                      out.data_ptr() + 4, buf.data_ptr() + 4 * (o + n_ce), n_fe, st)
            mean_e = out[1] * (1.0 / E)
        else:
            mean_e = sc.new_full((), float('nan'))             # loss.mean() of an empty selection (models/loss.py:71-74)
        loss_f = mean_e
        if tp_classifier:
            if Dn > 0:
                _lib.call('tmpnn_focal_loss_fwd', g.det_row.data_ptr(), Dn, sc.data_ptr(), targets.data_ptr(), 0.0, 0, 1.0, 1.0,
                          out.data_ptr() + 8, buf.data_ptr() + 4 * (o + n_ce + n_fe), n_fd, st)
                mean_d = out[2] * (1.0 / Dn)
            else:
                mean_d = sc.new_full((), float('nan'))
            loss_f = mean_d + mean_e                           # train.py:81: focal_node(...) + focal_edge(...)
        ctx.g, ctx.lg, ctx.sc, ctx.targets, ctx.stats = g, lg, sc, targets, stats
        ctx.tp, ctx.shapes = bool(tp_classifier), (logits.shape, scores.shape)
        return out[0].clone().reshape(()), loss_f.reshape(())

    @staticmethod
    def backward(ctx, d_c, d_f):
        g: FrameGraph = ctx.g
        st = _stream()
        d_logits = d_scores = None
        if ctx.fused:
            # every row of both gradients written by one launch (no zero fills, no read-modify-write passes)
            if d_c is not None:
                d_logits = torch.empty_like(ctx.lg)
                d_c = d_c.reshape(1).float().contiguous()
            if d_f is not None:
                d_scores = torch.empty_like(ctx.sc)
                d_f = d_f.reshape(1).float().contiguous()
            sp = getattr(g, 'src_pos_ptr', None) or _lib.ptr(g.src_pos)
            dp = getattr(g, 'dst_pos_ptr', None) or _lib.ptr(g.dst_pos)
            _lib.call('tmpnn_train_losses_bwd', g.cref(), sp, dp, ctx.lg.data_ptr(),
                      ctx.sc.data_ptr(), ctx.targets.data_ptr(), ctx.stats.data_ptr(), _lib.ptr(d_c), _lib.ptr(d_f),
                      1 if ctx.tp else 0, _lib.ptr(d_logits), _lib.ptr(d_scores), st)
            return (None if d_logits is None else d_logits.reshape(ctx.shapes[0]),
                    None if d_scores is None else d_scores.reshape(ctx.shapes[1]), None, None, None)
        if d_c is not None:
            d_logits = torch.zeros_like(ctx.lg)
            dl = d_c.reshape(1).float().contiguous()
            _lib.call('tmpnn_ce_loss_bwd', g.cref(), _lib.ptr(g.src_pos), _lib.ptr(g.dst_pos), ctx.lg.data_ptr(),
                      ctx.stats.data_ptr(), dl.data_ptr(), d_logits.data_ptr(), st)
            d_logits = d_logits.reshape(ctx.shapes[0])
        if d_f is not None:
            d_scores = torch.zeros_like(ctx.sc)
            df = d_f.reshape(1).float().contiguous()
            if g.E > 0:
                _lib.call('tmpnn_focal_loss_bwd', g.edge_row.data_ptr(), g.E, ctx.sc.data_ptr(), ctx.targets.data_ptr(), 0.0,
                          0, 1.0, 1.0, df.data_ptr(), 1.0 / g.E, d_scores.data_ptr(), st)
            if ctx.tp and g.Dn > 0:
                _lib.call('tmpnn_focal_loss_bwd', g.det_row.data_ptr(), g.Dn, ctx.sc.data_ptr(), ctx.targets.data_ptr(), 0.0,
                          0, 1.0, 1.0, df.data_ptr(), 1.0 / g.Dn, d_scores.data_ptr(), st)
            d_scores = d_scores.reshape(ctx.shapes[1])
        return d_logits, d_scores, None, None, None


class _DGView:
    """What the one-launch losses need of a DeviceGraph whose sizes the host knows: the sizes and a struct tmpnn_graph over its
    arena -- no tensor views (DeviceGraph.frame_graph() builds a dozen of them: ~20 us per call of a batch-1 loop)."""

    def __init__(self, dg: DeviceGraph):
        E, Dn, _ = dg.meta()
        c = dg.c
        self.N, self.E, self.Dn = dg.N, E, Dn
        self._c = _lib.CGraph(dg.N, E, Dn, c.src, c.dst, c.edge_row, c.det_row, c.rowptr, c.inc, None, None)
        self.src_pos_ptr, self.dst_pos_ptr = c.src_pos, c.dst_pos
        self._dg = dg                                    # (keeps the arena alive)

    def cref(self):
        return ctypes.byref(self._c)


def train_losses(scores: torch.Tensor, logits: torch.Tensor, labels: torch.Tensor, node_adj, tp_classifier: bool = True):
    """(loss_c, loss_f) of one forward call as train.py:70-81 computes them (CELoss on the logits; FocalLoss(gamma=0) on the
    scores of the edge rows, plus that of the det rows with the TP classifier) from the row labels."""
    _need_cuda(scores, 'scores')
    if (isinstance(node_adj, DeviceGraph) and node_adj._meta is not None and node_adj._meta[2] == 0
            and _lib.load().tmpnn_train_losses_supported(node_adj._meta[0], node_adj._meta[1])):
        g = _DGView(node_adj)
    else:
        g = _as_graph(node_adj)
    lab = labels.reshape(-1)
    lab = lab if (lab.dtype == torch.uint8 and lab.is_contiguous()) else (lab != 0).to(torch.uint8).contiguous()
    # ([N, 1] scores: a reshape is a view both ways; scores[:, 0] would cost a zero fill + a copy in the backward)
    sc = (scores.reshape(-1) if scores.shape[1] == 1 else scores[:, 0]) if scores.dim() == 2 else scores
    if isinstance(g, _DGView):
        fast = _fast_losses()
        if fast is not None:
            # the same two launches from a native autograd node (csrc_host/fast_iter.cpp TrainLossesFn)
            a = _fast_addrs()
            info = [a[0], a[1], a[2], ctypes.addressof(g._c), int(bool(tp_classifier)), g.src_pos_ptr or 0, g.dst_pos_ptr or 0,
                    _stream(), int(_lib.load().tmpnn_train_losses_ws(g.E, g.Dn))]
            loss_c, loss_f = fast.train_losses(logits, sc, lab, info, [g._dg.arena])
            return loss_c, loss_f
    return _TrainLosses.apply(logits, sc, lab, g, bool(tp_classifier))


_fast_state = {}


def _fast_losses():
    if 'mod' not in _fast_state:
        from .small import fast_module
        m = fast_module()
        _fast_state['mod'] = m if (m is not None and hasattr(m, 'train_losses')) else None
    return _fast_state['mod']


def _fast_addrs():
    if 'addrs' not in _fast_state:
        lib = _lib.load()
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value
        _fast_state['addrs'] = (addr(lib.tmpnn_train_losses_fwd), addr(lib.tmpnn_train_losses_bwd), addr(lib.tmpnn_last_error))
    return _fast_state['addrs']
