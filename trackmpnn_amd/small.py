"""Batch-1 path: one TrackMPNN.forward call on ONE small graph = one ctypes call (tmpnn_mp_iter_fwd / _bwd).

The reference calls the model once per timestep on a single tracking window (train.py:92-107, infer.py:60-87).  At
that size the cost of a call is launches and host bookkeeping, not flops, so this path keeps both minimal:
the graph is a `DeviceGraph` (sizes stay on the device, no host round trip), the iteration is two launches forward and
two backward (csrc/small.hip), parameters travel as one cached pointer struct, and everything saved for the backward
lives in one buffer.  Eligible calls: K = 0 attention heads, H in {32, 64}, N <= 65535 rows, one BatchNorm segment;
anything else takes the staged path of functional.py.  There is no fallback to torch ops or to the oracle.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .graph import DG_BIG_ROWS, DeviceGraph

_PTR_FIELDS = ('w1', 'b1', 'gamma', 'beta', 'w2', 'b2', 'run_mean', 'run_var', 'e_wih', 'e_whh', 'e_bih', 'e_bhh',
               'n_wih', 'n_whh', 'n_bih', 'n_bhh')


def _stream() -> int:
    return _lib.raw_stream()


_fast_mod = False       # False: not tried yet ; None: unavailable


def fast_module():
    """The C++ autograd node (trackmpnn_amd/lib/_tmpnn_fast.so, built by __graft_entry__.build_host from
    csrc_host/fast_iter.cpp): the same node as `_SmallIter` without the interpreter, for the in-place-gradient mode.
    TMPNN_FAST=0 (or a missing / unloadable extension) keeps the Python node -- both drive the same HIP kernels."""
    global _fast_mod
    if _fast_mod is False:
        _fast_mod = None
        import os
        if os.environ.get('TMPNN_FAST', '1') != '0':
            path = os.path.join(os.path.dirname(_lib.LIB_PATH), '_tmpnn_fast.so')
            if os.path.exists(path):
                try:
                    import importlib.util
                    spec = importlib.util.spec_from_file_location('_tmpnn_fast', path)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    _fast_mod = mod
                except Exception:      # noqa: BLE001  (ABI mismatch with the installed torch: stay on the Python node)
                    _fast_mod = None
    return _fast_mod


_DEBUG = os.environ.get('TMPNN_DEBUG', '0') == '1'


class SmallPath:
    """Per-model cache of what a fused call needs: the parameter pointer struct, the MFMA operand images of the GRU
    weights (rebuilt when a weight's version counter changes, i.e. once per optimizer step) and the layout of the
    flat per-call gradient buffer."""

    def __init__(self, model, padded: bool = False):
        self.model = model
        spec = model.spec
        self.spec = spec
        # padded: the state of a zero-padded width (nhidden not instantiated, padded to 32 / 64): the parameters it is
        # handed are the padded COPIES (TrackMPNN._padded_params) and its BatchNorm buffers the padded ones (`pad_buffers`)
        self.padded = padded
        self.pad_buffers = None
        self.names: List[str] = spec.param_names()
        self.eligible = spec.H in (32, 64) and spec.K <= 8      # (more than 8 heads: the staged kernels run them in groups)
        # attention heads (K > 0): the same two launches per direction with the attention stage (tmpnn_att_fwd / _bwd) between
        # them (tmpnn_mp_iter_*_parts) -- through the Python node below; the C++ node serves K = 0
        self.att = spec.K > 0
        self.f_fwd_parts = self.f_bwd_parts = None
        self._ptr_key = None
        self._ver_key = None
        self.cparams = None
        self.prep = None
        self._grad_struct_cache = {}
        self._tmpl = None
        self.f_fwd = self.f_bwd = None
        self._fast_addrs = None
        # layout of the flat gradient buffer (256-byte aligned slices), in param_names() order
        named = dict(model.named_parameters())
        self.set_grad_layout([named[nm] for nm in self.names])

    def set_grad_layout(self, plist) -> None:
        """Slices of the per-call flat gradient buffer for parameters of these shapes (the padded copies of a padded width)."""
        sizes, offs = [], [0]
        for p in plist:
            sizes.append(p.numel())
            offs.append(offs[-1] + ((sizes[-1] + 63) // 64) * 64)
        self.grad_sizes, self.grad_offs, self.grad_total = sizes, offs[:-1], offs[-1]

    # -- structs ---------------------------------------------------------------------------------------------
    def _fill(self, st: _lib.CMpParams, ptr_of) -> None:
        spec = self.spec
        st.G, st.H, st.IN_e, st.F_total = spec.G, spec.H, spec.IN_e, spec.F_total
        for g, (_, F) in enumerate(spec.groups):
            st.F[g] = F
            t, f = f'input_transforms.{g}.', f'factor_grus.{g}.'
            st.w1[g], st.b1[g] = ptr_of(t + '0.weight'), ptr_of(t + '0.bias')
            st.gamma[g], st.beta[g] = ptr_of(t + '1.weight'), ptr_of(t + '1.bias')
            st.w2[g], st.b2[g] = ptr_of(t + '3.weight'), ptr_of(t + '3.bias')
            st.e_wih[g], st.e_whh[g] = ptr_of(f + 'edge_gru.weight_ih'), ptr_of(f + 'edge_gru.weight_hh')
            st.e_bih[g], st.e_bhh[g] = ptr_of(f + 'edge_gru.bias_ih'), ptr_of(f + 'edge_gru.bias_hh')
            st.n_wih[g], st.n_whh[g] = ptr_of(f + 'node_gru.weight_ih'), ptr_of(f + 'node_gru.weight_hh')
            st.n_bih[g], st.n_bhh[g] = ptr_of(f + 'node_gru.bias_ih'), ptr_of(f + 'node_gru.bias_hh')
        st.w_node, st.b_node = ptr_of('output_transform_node.weight'), ptr_of('output_transform_node.bias')
        st.w_edge, st.b_edge = ptr_of('output_transform_edge.weight'), ptr_of('output_transform_edge.bias')

    def params(self, plist) -> _lib.CMpParams:
        """Pointer struct of the parameters (+ BatchNorm buffers) and fresh operand images; cached by data pointers
        and version counters."""
        if self.f_fwd is None:
            self.f_fwd, self.f_bwd = _lib.fn('tmpnn_mp_iter_fwd'), _lib.fn('tmpnn_mp_iter_bwd')
            self.f_fwd_parts, self.f_bwd_parts = _lib.fn('tmpnn_mp_iter_fwd_parts'), _lib.fn('tmpnn_mp_iter_bwd_parts')
        k = self._ptr_key
        if k is not None and (plist[0].data_ptr() != k[0] or plist[-1].data_ptr() != k[-1]):
            self.invalidate()                # storage replaced behind our back (p.data = ...): cheap sentinel check
        if self._ptr_key is None:
            key = tuple(p.data_ptr() for p in plist)
            for nm, p in zip(self.names, plist):
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError(f'{nm}: the HIP path needs contiguous fp32 parameters on the GPU (no CPU or '
                                       'torch fallback exists)')
            ptrs = dict(zip(self.names, key))
            st = _lib.CMpParams()
            self._fill(st, ptrs.__getitem__)
            bufs = self.pad_buffers if self.padded else dict(self.model.named_buffers())
            for g in range(self.spec.G):
                st.run_mean[g] = bufs[f'input_transforms.{g}.1.running_mean'].data_ptr()
                st.run_var[g] = bufs[f'input_transforms.{g}.1.running_var'].data_ptr()
                st.num_batches_tracked[g] = bufs[f'input_transforms.{g}.1.num_batches_tracked'].data_ptr()
            self.cparams, self._ptr_key, self._ver_key = st, key, None
            self._w_idx = [i for i, nm in enumerate(self.names) if nm.endswith(('gru.weight_ih', 'gru.weight_hh'))]
        vkey = tuple(plist[i]._version for i in self._w_idx)
        capturing = torch.cuda.is_current_stream_capturing()
        # (edits through `.data` do not move the version counters: TrackMPNN.refresh_weights(); TMPNN_DEBUG=1 rebuilds always)
        if vkey != self._ver_key or capturing or _DEBUG:
            lib = _lib.load()
            spec = self.spec
            if self.prep is None or self.prep.device != plist[0].device:
                self.prep = torch.empty((int(lib.tmpnn_mp_iter_prep_floats(spec.G, spec.H, spec.IN_e)),),
                                        dtype=torch.float32, device=plist[0].device)
            _lib.call('tmpnn_mp_iter_prepare', C.byref(self.cparams), self.prep.data_ptr(), _stream())
            self._ver_key = None if capturing else vkey
        return self.cparams

    def fast_info(self, plist, graph, n_grad_struct, training: bool, need_grad: bool, append: bool, spare: int,
                  sink_total: int = 0):
        """The 18-integer call descriptor of the C++ node (function addresses, struct addresses, sizes, flags;
        sink_total > 0: n_grad_struct is the offset template of the gradient sink, see grad_template)."""
        cp = self.params(plist)
        if self._fast_addrs is None:
            lib = _lib.load()
            addr = lambda f: C.cast(f, C.c_void_p).value
            self._fast_addrs = (addr(lib.tmpnn_mp_iter_fwd), addr(lib.tmpnn_mp_iter_bwd), addr(lib.tmpnn_last_error),
                                addr(lib.tmpnn_dgraph_bind))
        a = self._fast_addrs
        spec = self.spec
        return [a[0], a[1], a[2], C.addressof(cp), C.addressof(n_grad_struct), self.prep.data_ptr(), a[3], graph.N,
                spec.G, spec.H, spec.IN_e, spec.F_total, _stream(), spare, int(training), int(need_grad), int(append),
                int(sink_total)]

    def grad_template(self, plist):
        """(offset template, total floats, offsets, shapes) of the gradient SINK: a tmpnn_mp_params whose pointer fields
        hold 1 + the byte offset of that parameter's gradient inside one flat buffer (param_names() order, every slice
        16-byte aligned); the C++ node rebases it onto the buffer it allocates in its backward."""
        if self._tmpl is None:
            offs, shapes, o = [], [], 0
            for p in plist:
                offs.append(o)
                shapes.append(tuple(p.shape))
                o += (p.numel() + 3) & ~3
            st = _lib.CMpParams()
            enc = dict(zip(self.names, (1 + 4 * v for v in offs)))
            self._fill(st, enc.__getitem__)
            self._tmpl = (st, o, offs, shapes)
        return self._tmpl

    def invalidate(self) -> None:
        """Parameter / buffer storage may have moved (Module._apply, load_state_dict(assign=True)): re-read pointers."""
        self._ptr_key = None
        self._ver_key = None
        self._grad_struct_cache.clear()

    def grad_struct(self, tensors) -> _lib.CMpParams:
        """Pointer struct over a list of gradient tensors (param_names() order); cached by their data pointers.  The
        tensors ride along on the struct (`_keep`): the native node holds them for as long as it may write through it."""
        key = tuple(t.data_ptr() for t in tensors)
        st = self._grad_struct_cache.get(key)
        if st is None:
            if len(self._grad_struct_cache) > 8:
                self._grad_struct_cache.clear()
            st = _lib.CMpParams()
            self._fill(st, dict(zip(self.names, key)).__getitem__)
            st._keep = list(tensors)
            self._grad_struct_cache[key] = st
        return st

    def keep(self, gst=None) -> list:
        """What a native node must keep alive besides its own buffers: the operand images and, in the in-place gradient
        mode, the gradient tensors of `gst`."""
        return [self.prep] + gst._keep if gst is not None and hasattr(gst, '_keep') else [self.prep]


class _ParamSink(torch.autograd.Function):
    """The model's gradient sink: a tensor nobody reads (uninitialised, no kernel) that the native call nodes take as an
    input.  Each node's backward returns ALL parameter gradients of its call as one flat buffer; autograd sums those per
    call, and this node hands the slices to the parameters ONCE per backward pass -- hooks, torch.autograd.grad and DDP
    see ordinary gradients, at 6 + 22 small launches per window instead of 154 (7 calls x 22 parameters)."""

    @staticmethod
    def forward(ctx, total, offs, shapes, *params):
        ctx.meta = (offs, shapes)
        return params[0].new_empty((total,))

    @staticmethod
    def backward(ctx, g):
        offs, shapes = ctx.meta
        out = []
        for o, shp in zip(offs, shapes):
            n = 1
            for d in shp:
                n *= d
            out.append(g[o:o + n].view(shp))
        return (None, None, None) + tuple(out)


class _SmallIter(torch.autograd.Function):
    """autograd node of one fused forward call.  Inputs: (call dict, x, h_in | None, *params)."""

    @staticmethod
    def forward(ctx, call, x, h_in, *params):
        # params: every parameter in param_names() order, or -- when gradients are accumulated in place -- ONE
        # dummy tensor that only tells autograd the outputs need a backward
        sp: SmallPath = call['small']
        spec = sp.spec
        dg: DeviceGraph = call['graph']
        lib = _lib.load()
        H, G = spec.H, spec.G
        GH = G * H
        N, n = dg.N, int(x.shape[0])
        N_old = N - n
        dev = x.device
        if h_in is None:
            if N_old != 0:
                raise ValueError(f'h_in is None but the graph has {N_old} rows that are not new')
        elif h_in.shape[0] != N_old or h_in.shape[1] != GH:
            raise ValueError(f'h_in must be [{N_old}, {GH}] (N - n, G*H), got {tuple(h_in.shape)}')
        if n > 0 and x.shape[1] != spec.F_total:
            raise ValueError(f'x must be [{n}, {spec.F_total}], got {tuple(x.shape)}')
        training = call['training']
        if training and n == 1:
            raise ValueError(f'Expected more than 1 value per channel when training, got input size [1, {H}]')
        cp = sp.params(call['param_objs'])
        xd = x.detach()
        if xd.dtype != torch.float32 or not xd.is_contiguous():
            xd = xd.float().contiguous()
        opts = dict(dtype=torch.float32, device=dev)
        # ---- the state the iteration reads: carried rows + room for the new ones (appended in place when the carried
        # tensor was produced by this path with spare rows and has not been continued from before)
        h_cat = None
        if call['append']:
            hd = h_in.detach()
            if hd.is_contiguous() and hd.dtype == torch.float32:
                h_cat = torch.empty(0, **opts).set_(hd.untyped_storage(), hd.storage_offset(), (N, GH), (GH, 1))
        if h_cat is None:
            if h_in is not None and n == 0:
                hd = h_in.detach()
                h_cat = hd if (hd.is_contiguous() and hd.dtype == torch.float32) else hd.float().contiguous()
            else:
                h_cat = torch.empty((N, GH), **opts)
                if N_old > 0:
                    h_cat[:N_old].copy_(h_in.detach())
        need_grad = call['need_grad']
        spare_out = call['spare']
        buf = torch.empty(((N + spare_out) * GH,), **opts)
        h_out = torch.empty(0, **opts).set_(buf.untyped_storage(), 0, (N, GH), (GH, 1))
        logits = torch.empty((N, 1), **opts)
        scores = torch.empty((N, 1), **opts)
        save = None
        nsave = 0
        if need_grad or n > 0 or sp.att:
            nsave = save_floats(N, n, G, H)
            save = torch.empty((nsave,), **opts)
        fargs = (C.byref(cp), sp.prep.data_ptr(), dg.cref(), n, xd.data_ptr() if n > 0 else None,
                 int(xd.shape[1]) if n > 0 else 0, h_cat.data_ptr(), int(training), h_out.data_ptr(), logits.data_ptr(),
                 scores.data_ptr(), _lib.ptr(save), nsave)
        att = None
        if not sp.att:
            rc = sp.f_fwd(*fargs, _stream())
        else:
            # attention heads: input transform | tmpnn_att_fwd per feature group (its aggregate lands in the save buffer's
            # es area, where the det tiles of the iteration launch read it) | iteration
            rc = sp.f_fwd_parts(*fargs, 2, _stream())
            if not rc:
                att = _att_forward(sp, call, dg, h_cat, save, N, n, training)
                rc = sp.f_fwd_parts(*fargs, 1 | 4, _stream())
        if rc:
            raise RuntimeError(f'tmpnn_mp_iter_fwd failed (code {rc}): {_lib.last_error()}')
        ctx.call = call
        ctx.att = att if need_grad else None
        ctx.saved = (xd, h_cat, h_out, scores, save, cp) if need_grad else None
        ctx.has_h = h_in is not None
        ctx.n = n
        ctx.set_materialize_grads(False)
        return scores, logits, h_out

    @staticmethod
    def backward(ctx, d_scores, d_logits, d_hout):
        call = ctx.call
        sp: SmallPath = call['small']
        spec = sp.spec
        dg: DeviceGraph = call['graph']
        call['check_pending']()                      # deferred graph validation (one host read for the whole chunk)
        lib = _lib.load()
        xd, h_cat, h_out, scores, save, cp = ctx.saved
        ctx.saved = None
        H, G = spec.H, spec.G
        GH = G * H
        N, n = dg.N, ctx.n
        dev = h_cat.device
        opts = dict(dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad
        names = sp.names
        objs = call['param_objs']

        def f32c(t):
            return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()

        def strided(t):
            """(tensor, element stride) of an [N, 1] / [N] gradient; expanded (stride 0) and strided views are
            passed as they are -- the gradient of a sum or of a slice never has to be materialised"""
            if t is None:
                return None, 0
            if t.dtype != torch.float32:
                t = t.float()
            st = t.stride(0) if t.numel() > 1 else 1
            if st < 0:
                t, st = t.contiguous(), 1
            return t, st

        ds, st_ds = strided(d_scores)
        dl, st_dl = strided(d_logits)
        dh = f32c(d_hout) if d_hout is not None else None
        inplace = call['anchored']
        if inplace:
            gts = [p.grad for p in objs]
            if any(g is None for g in gts):
                raise RuntimeError('in-place parameter gradients: a p.grad buffer disappeared between forward and '
                                   'backward (use zero_grad(set_to_none=False))')
            flat = None
        else:
            flat = torch.zeros((sp.grad_total,), **opts)
            gts = [flat[o:o + sz] for o, sz in zip(sp.grad_offs, sp.grad_sizes)]
        gst = sp.grad_struct(gts)
        d_h = torch.empty((N, GH), **opts)
        need_x = need[1] and n > 0
        d_x = torch.empty((n, spec.F_total), **opts) if need_x else None
        wsb = bwd_ws_bytes(N, n, G, H, spec.IN_e)
        ws = torch.empty((wsb // 4 + 4,), **opts)
        bargs = (C.byref(cp), sp.prep.data_ptr(), dg.cref(), n, xd.data_ptr() if n > 0 else None,
                 int(xd.shape[1]) if n > 0 else 0, h_cat.data_ptr(), h_out.data_ptr(), scores.data_ptr(), save.data_ptr(),
                 int(call['training']), _lib.ptr(ds), st_ds, _lib.ptr(dl), st_dl, _lib.ptr(dh), d_h.data_ptr(), _lib.ptr(d_x),
                 C.byref(gst), ws.data_ptr(), wsb)
        if not sp.att:
            rc = sp.f_bwd(*bargs, _stream())
        else:
            # tiles (d_h, d_x of every row at the head of ws) | tmpnn_att_bwd per feature group | finish without the row-F adjoint
            rc = sp.f_bwd_parts(*bargs, 2, _stream())
            if not rc:
                _att_backward(sp, ctx.att, dg, h_cat, ws, d_h, dict(zip(names, gts)))
                rc = sp.f_bwd_parts(*bargs, 1 | 4, _stream())
            ctx.att = None
        if rc:
            raise RuntimeError(f'tmpnn_mp_iter_bwd failed (code {rc}): {_lib.last_error()}')
        if need[1] and d_x is None:
            d_x = torch.zeros((n, spec.F_total), **opts)
        d_h_in = d_h[:N - n] if (ctx.has_h and need[2] and N - n > 0) else None
        if inplace:
            return (None, d_x, d_h_in, None)
        named = dict(zip(names, objs))
        return (None, d_x, d_h_in) + tuple(g.view(named[nm].shape) for nm, g in zip(names, gts))


def _att_forward(sp: 'SmallPath', call: dict, dg: DeviceGraph, h_cat: torch.Tensor, save: torch.Tensor, N: int, n: int,
                 training: bool):
    """tmpnn_att_fwd per feature group on the state the input transform launch completed; the aggregate goes straight into
    the save buffer's es area ([G][N][H] by det index).  Returns what the backward needs + the attention values."""
    from . import functional as F_
    spec = sp.spec
    H, G, K = spec.H, spec.G, spec.K
    GH = G * H
    fg = dg.frame_graph()                      # (host-side E / Dn: the attention launches are sized from them)
    E, Dn = fg.E, fg.Dn
    dev = h_cat.device
    opts = dict(dtype=torch.float32, device=dev)
    named = dict(zip(sp.names, call['param_objs']))
    erec, inc_other = fg.att_index() if E > 0 else (None, None)
    es_off = G * 4 * N * H                     # tmpnn_mp_iter_save_es_offset restated (tests/test_abi.py)
    keep = call.get('keep')
    st = _stream()
    out, alphas = [], []
    for gi in range(G):
        f = f'factor_grus.{gi}.'
        Ws = [named[f + f'gat.{k}.W_att'].detach() for k in range(K)]
        As = [named[f + f'gat.{k}.a'].detach() for k in range(K)]
        W = F_._cached(('attW', tuple(t.data_ptr() for t in Ws)), tuple(Ws), lambda: torch.cat(Ws, 1).contiguous())
        a = F_._cached(('atta', tuple(t.data_ptr() for t in As)), tuple(As),
                       lambda: torch.stack([t.reshape(-1) for t in As]).contiguous())
        ha = torch.empty((max(Dn, 1), K * H), **opts)
        score = torch.empty((max(2 * E, 1), K), **opts)
        stats = torch.empty((max(Dn, 1), K, 2), **opts)
        esk = torch.empty((K, max(Dn, 1), H), **opts)
        alpha = torch.empty((K, max(2 * E, 1)), **opts)
        kp = F_._keep_bits(None if keep is None else keep[gi], K, 2 * E, dev) if training else None
        _lib.call('tmpnn_att_fwd', fg.cref(), _lib.ptr(erec), h_cat.data_ptr() + 4 * gi * H, GH, H, K, W.data_ptr(), a.data_ptr(),
                  _lib.ptr(kp), F_.ATT_DROPOUT_P, ha.data_ptr(), score.data_ptr(), stats.data_ptr(), esk.data_ptr(),
                  alpha.data_ptr(), save.data_ptr() + 4 * (es_off + gi * N * H), H, st)
        out.append((W, a, kp, ha, score, stats, esk))
        alphas.append([alpha[k, :2 * E] for k in range(K)])
    call['alphas'] = alphas
    return (fg, erec, inc_other, out)


def _att_backward(sp: 'SmallPath', att, dg: DeviceGraph, h_cat: torch.Tensor, ws: torch.Tensor, d_h: torch.Tensor, grads: dict):
    """tmpnn_att_bwd per feature group between the two launches of the iteration's backward: d_es is read from the det rows
    of the tile launch's d_x (the head of ws), d_h and the heads' gradient buffers are accumulated in place."""
    from . import functional as F_
    spec = sp.spec
    H, G, K, IN_e = spec.H, spec.G, spec.K, spec.IN_e
    GH = G * H
    fg, erec, inc_other, per_group = att
    E, Dn = fg.E, fg.Dn
    lib = _lib.load()
    st = _stream()
    wsn = int(lib.tmpnn_att_bwd_ws(E, Dn, H, K))
    ws_att = torch.empty((max(wsn, 1),), dtype=torch.float32, device=h_cat.device)
    for gi, (W, a, kp, ha, score, stats, esk) in enumerate(per_group):
        f = f'factor_grus.{gi}.'
        gW = [grads[f + f'gat.{k}.W_att'] for k in range(K)]
        ga = [grads[f + f'gat.{k}.a'] for k in range(K)]
        pW = (C.c_void_p * K)(*[t.data_ptr() for t in gW])
        pa = (C.c_void_p * K)(*[t.data_ptr() for t in ga])
        _lib.call('tmpnn_att_bwd_heads', fg.cref(), _lib.ptr(erec), _lib.ptr(inc_other), h_cat.data_ptr() + 4 * gi * H, GH, H, K,
                  W.data_ptr(), a.data_ptr(), _lib.ptr(kp), F_.ATT_DROPOUT_P, ha.data_ptr(), score.data_ptr(), stats.data_ptr(),
                  esk.data_ptr(), ws.data_ptr() + 4 * gi * IN_e, G * IN_e, ws_att.data_ptr(), ws_att.numel(),
                  d_h.data_ptr() + 4 * gi * H, GH, C.cast(pW, C.c_void_p), C.cast(pa, C.c_void_p), st)


def save_floats(N: int, n: int, G: int, H: int) -> int:
    """tmpnn_mp_iter_save_floats restated (one ctypes call less per forward; tests/test_abi.py checks the two agree)."""
    return G * 4 * N * H + G * N * H + G * max(n, 1) * H + 2 * G * H + G * (n + 1) + 4


def bwd_ws_bytes(N: int, n: int, G: int, H: int, IN_e: int) -> int:
    """tmpnn_mp_iter_bwd_ws restated."""
    nb = min(max((N + 15) // 16 + 2, 2), 96)
    slab = 3 * H * (IN_e + H) + 7 * H + 4
    return 4 * (N * G * IN_e + nb * G * slab + G * 2 * n * H + 16)


def small_eligible(model, N: int) -> bool:
    sp = model._small
    return sp.eligible and 0 < N <= DG_BIG_ROWS
