"""The rolling tracking graph kept ON THE DEVICE between model calls.

Mirror of the reference's graph bookkeeping -- `initialize_graph`, `update_graph`, `decode_tracks`
(reference/utils/graph.py:96-186, 189-334, 392-539) -- with the same arguments' meaning and the same results, but
the graph and the hidden state never visit the host: the graph lives in HBM in row form (csrc/trackops.hip), is
edited by small kernels (association rule, active set, block append, row deletion as a stream compaction) and its index
form (`DeviceGraph`) is re-derived on the device after every edit; the walk that finalises tracks into `y_out` runs on
the device too and `y_out` stays there until the sequence is done.  What stays on the host is what the reference itself
solves with scipy on a few dozen detections: the Hungarian matching (it reads a handful of int32 per row, never the
state).  Host reads per timestep in greedy mode: ONE -- decode's (kept rows + the next timestep's active-set size, `next_t`).

    tg, feats, t_st, t_end = TrackGraph.initialize(X, y, t_st=0, mode='test', device='cuda:0')
    scores, logits, h, _ = model.forward_dgraph(feats, None, tg.graph)
    for t in range(t_st, t_end):
        feats = tg.update(sc, X, y, t, mode='test')                             # update_graph
        scores, logits, h, _ = model.forward_dgraph(feats, h, tg.graph)
        h, sc = tg.decode(h, scores[:, 0], None, t - cur_win + 2, ret_win, next_t=t + 1)   # decode_tracks
    y_out[:, 1] = tg.tracks()
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .graph import DG_MAX_ROWS, DeviceGraph

TRACK_MAX_ROWS = 32768       # TMPNN_TRACK_MAX_ROWS
# update(mode='train'): the active set's size from the labels on the host instead of a device -> host read per timestep
# (TMPNN_TRACK_HOST_COUNTS=0 keeps the read); TMPNN_TRACK_VERIFY=1 reads the device's count as well and compares (tests)
_TRAIN_HOST_COUNTS = os.environ.get('TMPNN_TRACK_HOST_COUNTS', '1') != '0'
_TRACK_VERIFY = os.environ.get('TMPNN_TRACK_VERIFY', '0') == '1'
# decode / the native timestep: the launch mirrors its counters into pinned host memory and the host polls the mirror's flag
# instead of copying `small` back (include/tmpnn.h tmpnn_track_retire `notify`); TMPNN_TRACK_NOTIFY=0 keeps the copy
_TRACK_NOTIFY = os.environ.get('TMPNN_TRACK_NOTIFY', '1') != '0'
_NOTIFY_POOL: list = []      # pinned int32 [8] mirrors not in use (list.pop / append are atomic under the GIL)
# the native timestep: block append + the model call's input transform in one launch (tmpnn_track_extend_tf);
# TMPNN_TRACK_EXTEND_TF=0 keeps tmpnn_track_extend + the two launches of tmpnn_mp_iter_fwd
_TRACK_EXTEND_TF = os.environ.get('TMPNN_TRACK_EXTEND_TF', '1') != '0'
# the native driver enqueues a timestep's first launch behind the previous decode BEFORE reading that decode's counters (the launch
# reads N and A on the device; buffers sized by upper bounds); TMPNN_TRACK_EARLY_FRONT=0: after the read, as every other launch
_TRACK_EARLY_FRONT = os.environ.get('TMPNN_TRACK_EARLY_FRONT', '1') != '0'


def _stream() -> int:
    return _lib.raw_stream()


class TrackGraph:
    """Row form (+ index form) of one sequence's rolling graph on the device."""

    def __init__(self, device, cap: int = TRACK_MAX_ROWS):
        self.device = torch.device(device)
        self.cap = cap
        self.N = 0
        i32 = dict(dtype=torch.int32, device=self.device)
        u8 = dict(dtype=torch.uint8, device=self.device)
        # two sets of row arrays: deletion compacts from one into the other
        self._rows = [dict(ts=torch.empty(cap, **i32), det_id=torch.empty(cap, **i32), assoc=torch.empty(cap, **i32),
                           is_edge=torch.empty(cap, **u8), src=torch.empty(cap, **i32), dst=torch.empty(cap, **i32),
                           labels=torch.empty(cap, **u8)) for _ in range(2)]
        self._crows = [_lib.CTrackRows(*(d[k].data_ptr() for k in ('ts', 'det_id', 'assoc', 'is_edge', 'src', 'dst', 'labels')))
                       for d in self._rows]
        self._cur = 0
        self._prefetch = None                            # (timestep, the score tensor decode() returned, active-set size)
        self._fast_addrs = None
        self._fast_tpl = None
        self._hung_max = None
        self._hung_ws = None
        self.E = 0
        self.Dn = 0
        self._active = torch.empty(cap, **i32)
        self._keep = torch.empty(cap, **i32)
        self._small = torch.zeros(4, **i32)              # [0] count, [1] status, [2] kept dets, [3] next active-set size
        # the mirror of `small` a retire launch writes for the host: [0..3] the counters, [4] the flag (pinned, device-mapped)
        # (taken from a process-wide free list and handed back when this graph goes away: the loops make one TrackGraph per
        #  sequence, and pinning fresh host memory means a page-table update on the device under whatever kernel is running)
        self._notify = None
        if _TRACK_NOTIFY and self.device.type == 'cuda':
            try:
                self._notify = _NOTIFY_POOL.pop() if _NOTIFY_POOL else torch.zeros(8, dtype=torch.int32).pin_memory()
            except (RuntimeError, IndexError):      # (no pinned memory to be had: the counters are copied back instead)
                self._notify = None
        self._notify_np = None if self._notify is None else self._notify.numpy()
        self.track: Optional[torch.Tensor] = None        # int32 [ND] track id of every detection (training labels)
        self._graph: Optional[DeviceGraph] = None        # index form of the rows; None: stale (re-derived on first use)
        # per sequence, set by initialize(): the finalised tracks y_out[:, 1] (device, -1 = none yet), the detections of
        # every timestep (device ids sorted by time + host offsets) and the features, uploaded ONCE
        self.y_track: Optional[torch.Tensor] = None
        self._pos_of_det: Optional[torch.Tensor] = None
        self._ids_sorted: Optional[torch.Tensor] = None
        self._t_range: dict = {}
        self._Xd: Optional[torch.Tensor] = None
        self._fin_ws: Optional[torch.Tensor] = None

    def __del__(self):
        # every launch that writes the mirror has been waited for by the call that issued it: the buffer is free to reuse
        try:
            n = getattr(self, '_notify', None)
            if n is not None and len(_NOTIFY_POOL) < 64:
                _NOTIFY_POOL.append(n)
        except Exception:       # (interpreter shutdown: the module's globals may be gone)
            pass

    @property
    def rows(self):
        return self._rows[self._cur]

    def y_pred(self) -> torch.Tensor:
        """[N, 3] int64 as the reference keeps it (ts, det id, associated det id)."""
        r = self.rows
        return torch.stack([r['ts'][:self.N], r['det_id'][:self.N], r['assoc'][:self.N]], 1).long()

    def labels(self) -> torch.Tensor:
        return self.rows['labels'][:self.N].long()

    def labels_u8(self) -> torch.Tensor:
        """The row labels as the 0 / 1 bytes the loss kernels read (no int64 round trip)."""
        lab = self.rows['labels'][:self.N]
        return lab if lab.dtype == torch.uint8 else (lab != 0).to(torch.uint8)

    @property
    def graph(self) -> DeviceGraph:
        """Index form (CSR etc.) of the current rows.  After a decode() it is derived on first use only: in the greedy loop
        the next update() appends a block and derives the grown graph, the intermediate one is never looked at."""
        if self._graph is None:
            self._rebuild()
        return self._graph

    def _new_graph(self, N: int):
        """An unbuilt DeviceGraph for N rows (+ the conversion's global scratch beyond the LDS-resident size)."""
        g = DeviceGraph(N, self.device)
        ws = None
        if N > DG_MAX_ROWS:                       # dense scene: the conversion's work arrays in a global scratch
            ws = torch.empty((8 * N + 1,), dtype=torch.int32, device=self.device)
            g._keep = (ws,)
        return g, ws

    def _rebuild(self) -> None:
        r = self.rows
        g, ws = self._new_graph(self.N)
        _lib.call('tmpnn_graph_from_rows_ws', self.N, r['is_edge'].data_ptr(), r['src'].data_ptr(), r['dst'].data_ptr(),
                  g.cref(), _lib.ptr(ws), 0 if ws is None else ws.numel(), _stream())
        # the host knows E and Dn of its own graph (initial block, appended blocks, the delete kernel's counts): nothing
        # downstream (staged kernels, losses) has to read them back; graphs built by these kernels are valid by construction
        g._meta = (self.E, self.Dn, 0)
        self._graph = g

    # ---------------------------------------------------------------------------------------------------------------
    @classmethod
    def initialize(cls, X: torch.Tensor, y: torch.Tensor, t_st: int = 0, mode: str = 'test', device='cuda:0'):
        """reference initialize_graph (utils/graph.py:96-186): the first two non-empty timesteps at or after t_st.
        Returns (graph, feats [N, F] on the device, next timestep, end timestep) or None where the reference returns
        Nones.  (Built on the host: there is no device state yet and the block is a few dozen rows.)"""
        assert X.shape[0] == y.shape[0] == 1 and X.shape[1] == y.shape[1], 'Only batch size 1 supported!'
        yy = y[0].detach().cpu().numpy().astype(np.int64)
        times = np.unique(yy[:, 0])
        later = times[times >= t_st]
        if later.size < 2 or ((yy[:, 1] == -1).all() and mode == 'train'):
            return None
        t0, t1, tN = int(later[0]), int(later[1]), int(times[-1])
        ids0, ids1 = np.nonzero(yy[:, 0] == t0)[0], np.nonzero(yy[:, 0] == t1)[0]
        n0, n1 = ids0.size, ids1.size
        N = n0 + n0 * n1 + n1
        if N > TRACK_MAX_ROWS:
            raise ValueError(f'TrackGraph: {N} rows exceed the device-resident limit of {TRACK_MAX_ROWS}')
        ts = np.full(N, -1, np.int32)
        did = np.full(N, -1, np.int32)
        ts[:n0], ts[n0 + n0 * n1:] = t0, t1
        did[:n0], did[n0 + n0 * n1:] = ids0, ids1
        is_edge = (ts == -1).astype(np.uint8)
        src = np.full(N, -1, np.int32)
        dst = np.full(N, -1, np.int32)
        src[n0:n0 + n0 * n1] = np.repeat(np.arange(n0), n1)
        dst[n0:n0 + n0 * n1] = n0 + n0 * n1 + np.tile(np.arange(n1), n0)
        lab = np.zeros(N, np.uint8)
        trk = yy[:, 1]
        lab[:n0] = trk[ids0] >= 0
        lab[n0 + n0 * n1:] = trk[ids1] >= 0
        same = (trk[ids0][:, None] == trk[ids1][None, :]) & (trk[ids1][None, :] != -1)
        if (same.sum(0) > 1).any():
            raise AssertionError('More than one detection from same timestep assinged to same track!')
        lab[n0:n0 + n0 * n1] = same.reshape(-1)
        tg = cls(device)
        tg.N, tg.E, tg.Dn = N, n0 * n1, n0 + n1
        # ONE upload for the whole sequence: the block's rows, the detections sorted by time, their tracks and the features
        ND, F = int(yy.shape[0]), int(X.shape[2])
        order = np.argsort(yy[:, 0], kind='stable').astype(np.int32)
        parts = [ts, did, is_edge.astype(np.int32), src, dst, lab.astype(np.int32), order, trk.astype(np.int32)]
        x_host = not X.is_cuda
        if x_host:
            parts.append(np.ascontiguousarray(X[0].detach().float().numpy()).view(np.int32).reshape(-1))
        pk = torch.from_numpy(np.concatenate(parts)).to(tg.device)
        tg._ids_sorted = pk[6 * N:6 * N + ND]
        tg.track = pk[6 * N + ND:6 * N + 2 * ND]
        if x_host:
            tg._Xf = pk[6 * N + 2 * ND:].view(torch.float32).view(ND, F)
            tg._Xd = tg._Xf if X.dtype == torch.float32 else X[0].to(tg.device)
        else:
            tg._Xd = X[0].to(tg.device)
            tg._Xf = tg._Xd if (tg._Xd.dtype == torch.float32 and tg._Xd.is_contiguous()) else tg._Xd.float().contiguous()
        tg._sequence(yy, order)
        tg._order_host, tg._trk_host = order, trk          # (train mode: the active-set SIZE follows from the labels alone, see update)
        tg._tr = dict(last={}, dup=False, t_prev=t0, n_prev=0)
        tg._train_state_add(t0, trk[ids0].tolist())
        tg._train_state_add(t1, trk[ids1].tolist())
        tg._X_src, tg._y_src = X, y
        tg._src_versions = (X._version, y._version)        # (an in-place edit after initialize() must not pass the identity check)
        tg._y_host = y.detach().cpu() if y.is_cuda else y.detach().clone()
        feats = torch.empty((N, F), dtype=torch.float32, device=tg.device)
        g, ws = tg._new_graph(N)
        _lib.call('tmpnn_track_load', N, ND, pk.data_ptr(), C.byref(tg._crows[tg._cur]), tg._Xf.data_ptr(), F, F, feats.data_ptr(),
                  F, tg.y_track.data_ptr(), g.cref(), _lib.ptr(ws), 0 if ws is None else ws.numel(), _stream())
        g._meta = (tg.E, tg.Dn, 0)
        tg._graph = g
        tg._pk = pk                                        # (the views above keep it alive as well)
        return tg, (feats if tg._Xd.dtype == torch.float32 else feats.to(tg._Xd.dtype)), t1 + 1, tN + 1

    def _sequence(self, yy: np.ndarray, order: np.ndarray) -> None:
        """Per-sequence host state: where each timestep's detections sit in the time-sorted id list; device scratch."""
        ND = int(yy.shape[0])
        ts_sorted = yy[order, 0]
        self._t_range = {}
        if ND:
            cut = np.flatnonzero(np.diff(ts_sorted)) + 1
            lo = np.concatenate([[0], cut])
            hi = np.concatenate([cut, [ND]])
            self._t_range = {int(ts_sorted[a]): (int(a), int(b)) for a, b in zip(lo, hi)}
        self.y_track = torch.empty((max(ND, 1),), dtype=torch.int32, device=self.device)     # (-1 everywhere: tmpnn_track_load)
        self._pos_of_det = torch.empty((max(ND, 1),), dtype=torch.int32, device=self.device)

    def tracks(self) -> np.ndarray:
        """y_out[:, 1] of the sequence so far (one device -> host copy; call it when the sequence is done)."""
        return self.y_track.cpu().numpy().astype(np.int64)

    # ---------------------------------------------------------------------------------------------------------------
    def _hungarian(self, score_pos: torch.Tensor) -> None:
        """Frame-by-frame optimal assignment (reference hungarian(), utils/graph.py:33-93) on the host: a few dozen
        detections per frame, scipy's linear_sum_assignment; cost of an association = P(edge is negative) = 1 - score.
        ONE device -> host copy (row form + edge list + scores, packed) and one copy back; the cost blocks are filled
        with array indexing, no per-edge Python."""
        from scipy.optimize import linear_sum_assignment
        g = self.graph.frame_graph()
        N, E = self.N, self.E
        r = self.rows
        packed = torch.cat([r['ts'][:N], r['det_id'][:N], g.src[:E], g.dst[:E], g.edge_row[:E],
                            score_pos[:N].detach().float().contiguous().view(torch.int32)]).cpu().numpy()
        ts, did = packed[:N], packed[N:2 * N]
        src, dst, erow = packed[2 * N:2 * N + E], packed[2 * N + E:2 * N + 2 * E], packed[2 * N + 2 * E:2 * N + 3 * E]
        sc = packed[2 * N + 3 * E:].view(np.float32)
        assoc = np.full(N, -1, np.int32)
        if E:
            e_t = ts[dst]                                          # timestep an edge leads into
            e_cost = (1.0 - sc[erow]).astype(np.float32)
            order = np.argsort(e_t, kind='stable')
            bounds = np.flatnonzero(np.diff(e_t[order])) + 1
            for grp in np.split(order, bounds):                    # ascending timestep, as the reference sweeps
                s_g, d_g = src[grp], dst[grp]
                free = assoc[s_g] == -1                            # associated earlier in this sweep: taken
                if not free.any():
                    continue
                s_g, d_g, c_g = s_g[free], d_g[free], e_cost[grp][free]
                prev, pi = np.unique(s_g, return_inverse=True)
                cur = np.flatnonzero(ts == e_t[grp[0]])            # EVERY det of the timestep is a column, as in the reference
                cost = np.full((prev.size, cur.size), 100.0, np.float32)
                cost[pi, np.searchsorted(cur, d_g)] = c_g
                rows_i, cols_j = linear_sum_assignment(cost)
                ok = cost[rows_i, cols_j] <= 0.5
                assoc[prev[rows_i[ok]]] = did[cur[cols_j[ok]]]
        r['assoc'][:N].copy_(torch.from_numpy(assoc))

    def _hungarian_on_device(self) -> bool:
        """The optimal assignment runs inside the select / retire launches (csrc/trackops.hip d_track_hungarian: scipy's
        linear_sum_assignment restated, ties included) where the graph fits their one-launch form and no timestep's
        problem can exceed the device solver (rows and columns are dets of the graph); else `_hungarian` on the host."""
        if self._hung_max is None:
            self._hung_max = int(_lib.load().tmpnn_track_hungarian_max_dets())
        return 0 < self.N <= DG_MAX_ROWS and self.Dn <= self._hung_max and os.environ.get('TMPNN_HUNGARIAN_HOST', '0') != '1'

    # ---- update_graph(mode='train') without a host read: the active set's SIZE follows from the labels alone -----------------
    # (utils/graph.py:228-245, 271-274)  active = the dets of the previous non-empty timestep + every true-positive det that is
    # still unassociated.  A TP det's first later same-track det always finds it active (nothing associated it before), so the
    # edge between them exists and is its positive one: a TP det is unassociated iff the graph holds no later det of its track,
    # and it owns two positive edges ("More than one GT edge from same node!", raised by the reference at the NEXT update) iff
    # that first later timestep holds two dets of its track.  Training never deletes rows.  The state below is a few Python ints
    # per track, advanced once per appended timestep; the device derives the same set (tmpnn_track_select: the append needs its
    # order), TMPNN_TRACK_VERIFY=1 reads the device's count as well and compares.
    def _train_state_add(self, t: int, tracks) -> None:
        st = self._tr
        seen = {}
        for k in tracks:
            if k >= 0:
                seen[k] = seen.get(k, 0) + 1
        for k, m in seen.items():
            if m > 1 and k in st['last']:
                st['dup'] = True                           # (an earlier det of track k now has two positive edges)
            st['last'][k] = (t, m)
        st['t_prev'], st['n_prev'] = t, len(tracks)

    def _train_active_count(self):
        st = self._tr
        tp = st['t_prev']
        return st['n_prev'] + sum(m for (ts, m) in st['last'].values() if ts < tp), st['dup']

    def _hung_scratch(self) -> torch.Tensor:
        if self._hung_ws is None:
            self._hung_ws = torch.empty((self._hung_max * self._hung_max,), dtype=torch.float32, device=self.device)
        return self._hung_ws

    def update(self, score_pos: Optional[torch.Tensor], X: torch.Tensor, y: torch.Tensor, t: int, mode: str = 'test',
               use_hungarian: bool = False) -> torch.Tensor:
        """reference update_graph (utils/graph.py:189-334): re-derive the associations, pick the active dets, append
        the A x D_t edge rows and the D_t det rows of timestep t.  score_pos: P(positive) per row [N] (unused in
        training).  Returns the features of the new rows [A*D_t + D_t, F] (zeros on edge rows), on the device.
        One host read: the size of the active set."""
        # features and per-timestep detection ids were uploaded ONCE at initialize(): a different X / y here would be
        # silently ignored, so refuse it (the reference's loops pass the same sequence tensors every step)
        # Contract: X / y must hold what initialize() was given.  The same tensor objects pass when their
        # version counters have not moved; other tensors pass when they alias the same storage un-edited, or -- a loop that
        # re-slices y or moves X per step -- when shape, dtype and contents agree (both compared in full: y on the host, it is a
        # few hundred rows; X on the device against the cached copy).
        if X is self._X_src and y is self._y_src:
            if (X._version, y._version) != self._src_versions:
                raise ValueError('TrackGraph.update: X / y were modified in place after initialize(); their contents are '
                                 'cached on the device once per sequence -- start a new TrackGraph for new data')
        else:
            for given, kept, nm in ((X, self._X_src, 'X'), (y, self._y_src, 'y')):
                if kept is None or given is kept:
                    continue
                if given.shape == kept.shape and given.data_ptr() == kept.data_ptr() and given._version == kept._version:
                    continue
                same = given.shape == kept.shape and given.dtype == kept.dtype
                if same and nm == 'y':
                    same = bool(torch.equal(given.detach().cpu(), self._y_host))
                elif same and given.numel():
                    # compared IN FULL against the device copy (one device compare + one host read; the verdict is cached by
                    # taking `given` over as the identity to compare with, so a loop that keeps passing it pays this once)
                    b = self._Xd
                    same = bool(torch.equal(given.detach()[0].to(b.device, b.dtype), b))
                if not same:
                    raise ValueError(f'TrackGraph.update: {nm} differs from what initialize() was given (its contents are '
                                     'cached on the device once per sequence); start a new TrackGraph for new data')
            self._X_src, self._y_src, self._src_versions = X, y, (X._version, y._version)
        train = mode == 'train'
        pf, self._prefetch = self._prefetch, None
        N = self.N
        st = _stream()
        lo, hi = self._t_range.get(int(t), (0, 0))
        D = hi - lo
        Xd = self._Xd
        if (pf is not None and not train and bool(use_hungarian) == pf[4] and pf[0] == int(t) and score_pos is pf[1]
                and score_pos._version == pf[3]):
            # decode() already derived this timestep's active set on the compacted rows (greedy associations carry over; the
            # optimal assignment was re-derived over the rows that stayed, inside the same launch)
            A, status = pf[2], 0
        else:
            sp = None
            if not train:
                sp = score_pos.detach().reshape(-1).float().contiguous()
            hung = not train and use_hungarian
            hung_dev = hung and self._hungarian_on_device()
            if hung and not hung_dev:
                self._hungarian(sp)
            ws = self._hung_scratch() if hung_dev else None
            _lib.call('tmpnn_track_select_ws', self.graph.cref(), C.byref(self._crows[self._cur]), _lib.ptr(sp),
                      0 if train else 1, int(t), 2 if hung_dev else (0 if hung else 1), self._active.data_ptr(),
                      self._small.data_ptr(), _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, st)
            if D == 0 and not train and not hung_dev:
                return torch.zeros((0, Xd.shape[1]), dtype=Xd.dtype, device=self.device)
            if train and _TRAIN_HOST_COUNTS:
                # training: the active set's size and the label rule's assertion follow from the labels (no host read)
                A, dup = self._train_active_count()
                status = 1 if dup else 0
                if _TRACK_VERIFY:
                    A_dev, st_dev = self._small[:2].tolist()
                    assert (A_dev, st_dev & 1) == (A, status), f'train-mode active set: host {A, status}, device {A_dev, st_dev}'
            else:
                A, status = self._small[:2].tolist()       # the ONE host read of an update: the size of the active set
            if hung_dev and (status & 2):                  # a timestep's problem did not fit the device solver: match on the host
                self._hungarian(sp)
                _lib.call('tmpnn_track_select', self.graph.cref(), C.byref(self._crows[self._cur]), _lib.ptr(sp), 1, int(t), 0,
                          self._active.data_ptr(), self._small.data_ptr(), st)
                A, status = self._small[:2].tolist()
        if train and (status & 1):                         # (the label rule's assertion holds on empty timesteps too)
            raise AssertionError('More than one GT edge from same node!')
        if D == 0:
            return torch.zeros((0, Xd.shape[1]), dtype=Xd.dtype, device=self.device)
        n_new = A * D + D
        if N + n_new > self.cap:
            raise ValueError(f'TrackGraph: {N + n_new} rows exceed the device-resident limit of {self.cap}')
        F = int(Xd.shape[1])
        feats = torch.empty((n_new, F), dtype=torch.float32, device=self.device)
        g, ws = self._new_graph(N + n_new)
        _lib.call('tmpnn_track_extend', N, A, D, self._active.data_ptr(), self._ids_sorted.data_ptr() + 4 * lo, int(t),
                  _lib.ptr(self.track), C.byref(self._crows[self._cur]), self._Xf.data_ptr(), F, F, feats.data_ptr(), F, g.cref(),
                  _lib.ptr(ws), 0 if ws is None else ws.numel(), st)
        self.N, self.E, self.Dn = N + n_new, self.E + A * D, self.Dn + D
        if train:
            self._train_state_add(int(t), self._trk_host[self._order_host[lo:hi]].tolist())
        g._meta = (self.E, self.Dn, 0)
        self._graph = g
        return feats if Xd.dtype == torch.float32 else feats.to(Xd.dtype)

    @staticmethod
    def _check_assoc_status(status: int, hung_dev: bool) -> None:
        """Status word of a retire launch (include/tmpnn.h): bit 2 = a timestep's assignment problem did not fit the device
        solver and was left unassociated.  The pre-checks (`_hungarian_on_device`, the Dn + D test of the native step) keep it
        from firing; if it ever does, the tracks this launch finalised are incomplete -- fail loudly instead of carrying on."""
        if hung_dev and (status & 2):
            raise RuntimeError('TrackGraph: the device Hungarian solver reported an assignment problem it could not take '
                               '(status bit 2) inside a decode launch: the finalised tracks of this step are incomplete. '
                               'Set TMPNN_HUNGARIAN_HOST=1 to match on the host')

    # ---------------------------------------------------------------------------------------------------------------
    def decode(self, h: torch.Tensor, score_pos: torch.Tensor, y_out: Optional[np.ndarray], t_upto: int, ret_win_size: int,
               use_hungarian: bool = False, next_t: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """reference decode_tracks (utils/graph.py:392-539): re-derive the associations from the scores, finalise
        tracks up to t_upto and delete the decoded part of the graph -- all on the device, one call (tmpnn_track_retire:
        associate, finalize, delete, gather); ONE host read: the number of kept rows.  `next_t` (inference): the
        timestep the loop will hand to update() next -- its active set is then derived here, on the compacted rows, and
        read with the same host read, so that update(next_t) given the scores this call returns reads nothing.  Greedy
        associations carry over to those rows; under `use_hungarian` (matching on the device) the launch runs the sweep
        update_graph would run on the compacted graph and leaves ITS associations in the rows (what `y_pred()` shows between
        this call and the next update is then the next update's column 2, not the decode's).  The
        finalised tracks live in `self.y_track` (fetch them with tracks() when the sequence is done); pass a host array
        `y_out` [ND, 2] only where it must be current after every call (tests): it costs a device -> host copy.
        Returns the compacted (h', score_pos')."""
        N = self.N
        self._prefetch = None
        if N == 0:              # an emptied window (cur_win_size = 1 and a timestep without detections): nothing to decode
            sc0 = score_pos.detach().reshape(-1)[:0].float()
            if next_t is not None:
                self._prefetch = (int(next_t), sc0, 0, sc0._version, bool(use_hungarian))
            return h.detach()[:0].float(), sc0
        sp = score_pos.detach().reshape(-1).float().contiguous()
        hung_dev = use_hungarian and self._hungarian_on_device()
        if use_hungarian and not hung_dev:
            self._hungarian(sp)
        ND = int(self.y_track.numel())
        wsb = int(_lib.load().tmpnn_track_finalize_ws(N))
        if wsb and (self._fin_ws is None or self._fin_ws.numel() * 4 < wsb):
            self._fin_ws = torch.empty((wsb // 4 + 1,), dtype=torch.int32, device=self.device)
        hd = h.detach()
        hd = hd if (hd.dtype == torch.float32 and hd.is_contiguous()) else hd.float().contiguous()
        W = int(hd.shape[1])
        h_new = torch.empty((N, W), dtype=torch.float32, device=self.device)
        s_new = torch.empty((N, 1), dtype=torch.float32, device=self.device)
        nt = -1 if (next_t is None or (use_hungarian and not hung_dev)) else int(next_t)
        # ---- device, one call: associations, finalisation walk, deletion as a stream compaction of rows, state and scores,
        # and (next_t) the active set of the next timestep on the compacted rows
        if hung_dev:                                       # (N <= 4096: the finalisation needs no scratch; the slot carries the cost scratch)
            hws = self._hung_scratch()
            fin_ptr, fin_bytes = hws.data_ptr(), hws.numel() * 4
        else:
            fin_ptr, fin_bytes = (_lib.ptr(self._fin_ws) if wsb else None), wsb
        _lib.call('tmpnn_track_retire', self.graph.cref(), C.byref(self._crows[self._cur]), sp.data_ptr(),
                  2 if hung_dev else (0 if use_hungarian else 1), int(t_upto), int(ret_win_size), self.y_track.data_ptr(), ND,
                  self._pos_of_det.data_ptr(), fin_ptr, fin_bytes, self._keep.data_ptr(),
                  self._small.data_ptr(), C.byref(self._crows[1 - self._cur]), hd.data_ptr(), W, W, h_new.data_ptr(), W,
                  s_new.data_ptr(), nt, self._active.data_ptr(), self._notify_arm(), _stream())
        if y_out is not None:
            y_out[:, 1] = self.y_track[:y_out.shape[0]].cpu().numpy()
        # the ONE host read of a decode: kept rows (how many are dets; next A)
        n_keep, status, n_det, a_next = self._notify_wait() or self._small.tolist()
        self._check_assoc_status(status, hung_dev)
        self._cur = 1 - self._cur
        self.N, self.E, self.Dn = n_keep, n_keep - n_det, n_det
        self._graph = None                                 # (derived on first use: see `graph`)
        sc = s_new[:n_keep, 0]
        if nt >= 0:
            self._prefetch = (nt, sc, a_next, sc._version, bool(use_hungarian))
        return h_new[:n_keep], sc

    # ---------------------------------------------------------------------------------------------------------------
    def greedy_step_fast(self, fast, model_info, h: torch.Tensor, cap_rows: int, t: int, t_upto: int, ret_win_size: int,
                         next_t: Optional[int], use_hungarian: bool = False, tp_classifier: bool = True):
        """One steady-state inference timestep through the native driver: greedy_run_fast with a single step.  Returns
        (h', score', capacity of h' in rows) or None when the preconditions do not hold."""
        r = self.greedy_run_fast(fast, model_info, h, cap_rows, [(int(t), int(t_upto), -1 if next_t is None else int(next_t))],
                                 ret_win_size, use_hungarian, tp_classifier)
        return None if r is None else r[:3]

    def greedy_run_fast(self, fast, model_info, h: torch.Tensor, cap_rows: int, steps, ret_win_size: int,
                        use_hungarian: bool = False, tp_classifier: bool = True):
        """Steady-state inference timesteps (update -> eval model call -> decode; greedy or device-Hungarian association) through
        the native driver (csrc_host/fast_iter.cpp greedy_run / greedy_step): per timestep the same three tracker calls and the
        same model call as update() / TrackMPNN.forward_dgraph / decode(), issued without the interpreter between them or between
        the timesteps, and the timestep's one host read.  steps: [(t, t_upto, next_t or -1), ...] in loop order; as many of them
        as the native step takes are run (it stops in front of a timestep without detections, a grown graph beyond the
        one-launch kernels' 4096 rows, a problem the device solver may not take).
        Preconditions for the first step (else None and the caller takes the Python path): the previous decode / step prefetched
        its active set (`_prefetch`), D_t > 0, the grown graph fits.
        Returns (h', score', capacity of h' in rows, steps done, sum of E over their model calls) or None."""
        pf = self._prefetch
        t = int(steps[0][0])
        lo, hi = self._t_range.get(t, (0, 0))
        D = hi - lo
        if pf is None or pf[0] != t or D == 0 or pf[4] != bool(use_hungarian):
            return None
        A = pf[2]
        N = self.N
        n_new = A * D + D
        if N == 0 or N + n_new > DG_MAX_ROWS or self._Xd.dtype != torch.float32:
            return None
        hung_max = 0
        if use_hungarian:           # the device solver must take every timestep's problem of the grown graph (rows, columns: dets)
            if self._hung_max is None:
                self._hung_max = int(_lib.load().tmpnn_track_hungarian_max_dets())
            if self.Dn + D > self._hung_max or os.environ.get('TMPNN_HUNGARIAN_HOST', '0') == '1':
                return None
            hung_max = self._hung_max
        # what the native call would refuse with an exception is checked HERE, while nothing has been touched: the caller
        # then takes the Python path (update / forward_dgraph / decode) with the prefetched active set still in place
        GH = int(model_info[8]) * int(model_info[9])
        if not (h.dim() == 2 and h.dtype == torch.float32 and h.is_contiguous() and h.shape[0] == N and h.shape[1] == GH
                and int(self._Xf.shape[1]) == int(model_info[11])):
            return None
        self._prefetch = None
        tpl = self._fast_tpl
        if tpl is None or tpl[30] != bool(use_hungarian):
            # everything of the call descriptor that does not change within a sequence, built once: a dozen data_ptr() calls per
            # timestep sat between the host read of one timestep and the first launch of the next
            lib = _lib.load()
            addr = lambda f: C.cast(f, C.c_void_p).value
            tpl = [addr(lib.tmpnn_track_extend), addr(lib.tmpnn_track_retire), addr(lib.tmpnn_dgraph_ints), 0, 0, 0, 0, 0, 0, 0,
                   self._active.data_ptr(), self._ids_sorted.data_ptr(), self.track.data_ptr(), C.addressof(self._crows[0]),
                   C.addressof(self._crows[1]), self._Xf.data_ptr(), int(self._Xf.shape[1]), self.y_track.data_ptr(),
                   int(self.y_track.numel()), self._pos_of_det.data_ptr(), self._keep.data_ptr(), self._small.data_ptr(), 0, 0]
            if use_hungarian:
                hws = self._hung_scratch()
                tpl += [2, hws.data_ptr(), hws.numel() * 4]
            else:
                tpl += [1, 0, 0]
            tpl += [0 if self._notify is None else self._notify.data_ptr(),
                    addr(lib.tmpnn_track_extend_tf) if _TRACK_EXTEND_TF else 0, addr(lib.tmpnn_mp_iter_fwd_parts), bool(use_hungarian)]
            self._fast_tpl = tpl
        ti = tpl[:30]
        ti[8] = int(ret_win_size)
        if self._cur:
            ti[13], ti[14] = ti[14], ti[13]
        ti[23] = _stream()
        ti.append(0 if tp_classifier else 8)     # (no TP classifier: every detection's score is 1, infer.py:77-80)
        ids0 = tpl[11]
        flat = []
        tr = self._t_range
        for (ts_, tu_, nt_) in steps:
            l_, h_ = tr.get(int(ts_), (0, 0))
            flat += (int(ts_), int(tu_), int(nt_), ids0 + 4 * l_, h_ - l_)
        try:
            out, st = fast.greedy_run(ti, model_info, h, int(cap_rows), flat, [N, A, self.E, self.Dn],
                                      [DG_MAX_ROWS, hung_max, 1 if _TRACK_EARLY_FRONT else 0])
        except RuntimeError:
            # a C entry point refused its arguments before launching anything that changes the graph: rows [0, N) and the
            # counters are as they were (the appended block sits beyond N and the grown index form in its own arena)
            self._prefetch = pf
            raise
        done, edges, n_keep, e_keep, dn_keep, a_next, flips, cap = st
        if done == 0:               # (cannot happen after the checks above; nothing was touched)
            self._prefetch = pf
            return None
        h_keep, sc, counts = out
        self.last_E = edges                                # (sum of the edges of the graphs the model calls ran on)
        if flips & 1:
            self._cur = 1 - self._cur
        self.N, self.E, self.Dn = n_keep, e_keep, dn_keep
        self._graph = None
        self._check_assoc_status(int(counts[1]), bool(use_hungarian))
        nt = int(steps[done - 1][2])
        if nt >= 0:
            self._prefetch = (nt, sc, a_next, sc._version, bool(use_hungarian))
        return h_keep, sc, cap, done, edges

    def _notify_arm(self):
        """Clear the mirror's flag; its address for the launch (None: no mirror, the caller copies `small` back)."""
        if self._notify is None:
            return None
        self._notify_np[4] = 0
        return self._notify.data_ptr()

    def _notify_wait(self, limit_s: float = 0.02):
        """The counters once the launch has published them (polling the flag: no copy, no stream synchronisation); None when
        there is no mirror or the flag stays down for `limit_s` -- the caller then reads `small` (which synchronises)."""
        m = self._notify_np
        if m is None:
            return None
        spins = 0
        t0 = None
        while m[4] == 0:
            spins += 1
            if (spins & 0x3ff) == 0:
                now = time.perf_counter()
                if t0 is None:
                    t0 = now
                elif now - t0 > limit_s:
                    return None
        return [int(m[0]), int(m[1]), int(m[2]), int(m[3])]

    def kept_rows(self) -> torch.Tensor:
        """Rows of the previous graph that the last decode() kept (ascending)."""
        return self._keep[:self.N].long()
