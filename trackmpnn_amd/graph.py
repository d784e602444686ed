"""Device graph format of the hot path and the host code that produces it.

The reference hands `TrackMPNN.forward` a pair of N x N adjacency tensors
(reference/utils/graph.py:151-163, 294-308).  The HIP kernels want index arrays instead
(`FrameGraph`, = `struct tmpnn_graph` of include/tmpnn.h):

    src[e], dst[e]   det rows holding +1 / -1 in node_adj[edge e, :]
    edge_row, det_row
    rowptr/inc       det -> incident edge rows (CSR), sign of edge_adj[d, e] in bit 31

`graph_from_adjacency` converts (and validates) whatever the reference passes -- dense, coalesced
with explicit zeros, or uncoalesced COO.  `graph_from_edges` builds the same thing straight from
edge lists; `WindowBuilder` re-creates the reference's train-mode rolling construction
(utils/graph.py:96-186, 189-334) in index form, and `batch_windows` lays many windows out
block-diagonally in call-major row order so one launch serves thousands of tracking windows.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


@dataclass
class FrameGraph:
    N: int
    E: int
    Dn: int
    src: torch.Tensor        # int32 [E]
    dst: torch.Tensor        # int32 [E]
    edge_row: torch.Tensor   # int32 [E]
    det_row: torch.Tensor    # int32 [Dn]
    rowptr: torch.Tensor     # int32 [Dn+1]
    inc: torch.Tensor        # int32 [2E]   edge row | sign bit
    is_edge: torch.Tensor    # uint8 [N]
    pos: torch.Tensor        # int32 [N]    row -> index within its type (det index | edge index)
    src_pos: Optional[torch.Tensor] = None   # int32 [E]  det INDEX of src[e] (= pos[src])
    dst_pos: Optional[torch.Tensor] = None   # int32 [E]  det INDEX of dst[e]
    det_order: Optional[torch.Tensor] = None  # int32 [Dn] visiting order of the det -> edge reductions (tmpnn.h)
    _c: Optional[_lib.CGraph] = field(default=None, repr=False, compare=False)

    @property
    def device(self):
        return self.src.device

    def to(self, device) -> 'FrameGraph':
        if torch.device(device) == self.device:
            return self
        mv = lambda t: t.to(device)
        g = FrameGraph(self.N, self.E, self.Dn, mv(self.src), mv(self.dst), mv(self.edge_row), mv(self.det_row),
                       mv(self.rowptr), mv(self.inc), mv(self.is_edge), mv(self.pos),
                       None if self.src_pos is None else mv(self.src_pos),
                       None if self.dst_pos is None else mv(self.dst_pos),
                       None if self.det_order is None else mv(self.det_order))
        if '_det_group' in self.__dict__:                         # (the window labels: win_plan() is rebuilt over there)
            g.__dict__['_det_group'] = mv(self.__dict__['_det_group'])
        return g

    def cstruct(self) -> _lib.CGraph:
        if self._c is None:
            plan, wplan = self.__dict__.get('_seg_plan'), self.__dict__.get('_win_plan')
            self._c = _lib.CGraph(self.N, self.E, self.Dn, self.src.data_ptr(), self.dst.data_ptr(),
                                  self.edge_row.data_ptr(), self.det_row.data_ptr(), self.rowptr.data_ptr(),
                                  self.inc.data_ptr(), _lib.ptr(self.det_order),
                                  None if plan is None else C.addressof(plan.c),
                                  None if wplan is None else C.addressof(wplan.c))
        return self._c

    def cref(self):
        return C.byref(self.cstruct())

    def att_index(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(erec int32 [E][8], inc_other int32 [2E]) for the attention kernels (tmpnn_att_fwd / _bwd).  erec[e] = (src det
        index, dst det index, CSR position of e in the src det's run, ... in the dst det's run | src row, dst row, edge row,
        0): everything an edge-owned pass needs as one 32-byte record; inc_other[p] = det INDEX of the OTHER endpoint of CSR
        position p (a src det's positions hold dst_pos[e] and vice versa) with bit 31 set on dst-side positions.
        Index plumbing (torch ops on the device, no host round trip), built on first use and cached on the graph."""
        t = self.__dict__.get('_att_index')
        if t is None:
            if self.src_pos is None or self.dst_pos is None:
                raise ValueError('att_index: the graph carries no src_pos / dst_pos')
            erec = torch.empty((max(self.E, 1), 8), dtype=torch.int32, device=self.device)
            other = torch.empty((max(2 * self.E, 1),), dtype=torch.int32, device=self.device)
            if self.device.type == 'cuda':
                _lib.call('tmpnn_att_index', self.cref(), self.pos.data_ptr(), self.src_pos.data_ptr(), self.dst_pos.data_ptr(),
                          erec.data_ptr(), other.data_ptr(), _lib.raw_stream(self.device))
            else:                               # (host graphs, CPU tests: the same arrays with torch index ops)
                e = self.pos.long()[(self.inc & 0x7FFFFFFF).long()]
                neg = self.inc < 0
                other = torch.where(neg, self.src_pos[e] | torch.tensor(-2 ** 31, dtype=torch.int32), self.dst_pos[e]).to(torch.int32)
                erec.zero_()
                if self.E > 0:
                    erec[:, 0], erec[:, 1] = self.src_pos, self.dst_pos
                    erec[e, 2 + neg.long()] = torch.arange(2 * self.E, dtype=torch.int32)
                    erec[:, 4], erec[:, 5], erec[:, 6] = self.src, self.dst, self.edge_row
            t = (erec, other.contiguous())
            self.__dict__['_att_index'] = t
        return t

    def inc_edge_endpoint(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """For every CSR position: (edge index e, endpoint 0 = src side / 1 = dst side)."""
        row = (self.inc & 0x7FFFFFFF).long()
        endpoint = (self.inc < 0).long()
        epos = torch.full((self.N,), -1, dtype=torch.long, device=self.device)
        epos[self.edge_row.long()] = torch.arange(self.E, device=self.device)
        return epos[row], endpoint


@dataclass
class EdgeTiles:
    """struct tmpnn_edge_tiles: the edge rows cut into tiles that touch few distinct dets (include/tmpnn.h)."""
    T: int
    rows_per_tile: int
    t_row: torch.Tensor      # int32 [T * rows_per_tile]  graph row of each slot, -1 = padding
    t_loc: torch.Tensor      # int32 [T * rows_per_tile]  src position | dst position << 16 in the tile's det list
    t_dptr: torch.Tensor     # int32 [T + 1]
    t_dets: torch.Tensor     # int32 [t_dptr[T]]          det indices, ascending within a tile
    max_dets: int = 0        # longest det list (diagnostics; filled by build_edge_tiles(stats=True))
    _c: Optional[_lib.CEdgeTiles] = field(default=None, repr=False, compare=False)

    def cref(self):
        if self._c is None:
            self._c = _lib.CEdgeTiles(self.T, self.rows_per_tile, self.t_row.data_ptr(), self.t_loc.data_ptr(),
                                      self.t_dptr.data_ptr(), self.t_dets.data_ptr())
        return C.byref(self._c)


def build_edge_tiles(graph: FrameGraph, rows_per_tile: int = 128, src_group: int = 8, dst_group: int = 16,
                     stats: bool = False, order: str = 'blocks', dst_offset: int = 0) -> EdgeTiles:
    """Cut the graph's edge rows into tiles of `rows_per_tile` rows that touch few distinct dets.

    A frame block of the rolling graph is a dense [A srcs x D_t dsts] set of rows in src-major order
    (reference/utils/graph.py:141-156, 285-301).  The edges are ordered by (dst // dst_group, src // src_group, src, dst)
    and cut every `rows_per_tile`: where blocks are dense a tile is (src_group x dst_group) edges over
    src_group + dst_group dets; ragged graphs (after decode_tracks' row deletion) just get longer det lists.
    `order='rows'` keeps the graph's own row order instead (tiles of consecutive edge rows): the better choice when a
    src's run of edges is much shorter than a tile, as in batches of small windows (KITTI: ~6 dets per frame, 32
    consecutive rows touch ~16 dets, any regrouping touches more).  Index plumbing only (torch ops on the graph's
    device, no host round trip unless `stats`).

    `dst_offset`: added to every dst entry of the det lists -- the concat message reads its src half and its dst half from two
    different projected tables, stacked as rows [0, Dn) and [Dn, 2 Dn) of one (dst_offset = Dn)."""
    if graph.src_pos is None or graph.dst_pos is None:
        raise ValueError('build_edge_tiles: the graph carries no src_pos / dst_pos')
    dev = graph.device
    E, R = graph.E, int(rows_per_tile)
    T = (E + R - 1) // R
    i32 = lambda t: t.to(torch.int32).contiguous()
    if T == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return EdgeTiles(0, R, z, z.clone(), torch.zeros(1, dtype=torch.int32, device=dev), z.clone())
    s, d = graph.src_pos.long(), graph.dst_pos.long()
    ns = graph.Dn // src_group + 1
    if order == 'rows':
        so, do, ro = s, d, graph.edge_row.long()
    else:
        key = (((d // dst_group) * ns + s // src_group) * src_group + s % src_group) * dst_group + d % dst_group
        perm = torch.argsort(key)
        so, do, ro = s[perm], d[perm], graph.edge_row.long()[perm]
    pad = T * R - E
    if pad:
        so = torch.cat([so, so[-1:].expand(pad)])            # padding slots repeat the last edge's dets (no new det)
        do = torch.cat([do, do[-1:].expand(pad)])
        ro = torch.cat([ro, torch.full((pad,), -1, dtype=torch.long, device=dev)])
    both = torch.cat([so.view(T, R), do.view(T, R) + int(dst_offset)], 1)      # [T, 2R]
    vals, perm = both.sort(1)
    first = torch.ones_like(vals, dtype=torch.bool)
    first[:, 1:] = vals[:, 1:] != vals[:, :-1]
    rank = first.long().cumsum(1) - 1                        # position in the tile's det list, per sorted element
    local = torch.empty_like(rank)
    local.scatter_(1, perm, rank)
    cnt = rank[:, -1] + 1
    dptr = torch.zeros(T + 1, dtype=torch.long, device=dev)
    dptr[1:] = cnt.cumsum(0)
    loc = local[:, :R] | (local[:, R:] << 16)
    tiles = EdgeTiles(T, R, i32(ro), i32(loc.reshape(-1)), i32(dptr), i32(vals[first]))
    if stats:
        tiles.max_dets = int(cnt.max())
    return tiles


def edge_tiles(graph: FrameGraph, rows_per_tile: int = 128, dst_offset: int = 0) -> EdgeTiles:
    """The graph's cached tile list (built on first use)."""
    cache = graph.__dict__.setdefault('_tiles', {})
    key = rows_per_tile if not dst_offset else (rows_per_tile, int(dst_offset))
    t = cache.get(key)
    if t is None:
        # dense frame blocks (a src's run of edges spans a good part of a tile) are cut into src x dst sub-blocks; graphs of
        # small windows keep their row order.  The mean run length costs one host round trip, once per graph.
        runs = 1
        if graph.E > 1:
            if graph.src_pos.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('edge_tiles: the tile list of this graph is not built yet and building it reads one '
                                   'count back to the host, which a stream capture cannot do -- run one eager call on the '
                                   'graph (CapturedWindow does) before capturing')
            runs = 1 + int((graph.src_pos[1:] != graph.src_pos[:-1]).sum())
        blocks = graph.E / runs >= rows_per_tile / 2
        od = 'blocks' if blocks else 'rows'
        if rows_per_tile == 128:
            t = build_edge_tiles(graph, 128, 8, 16, order=od, dst_offset=dst_offset)
        elif rows_per_tile == 16:
            t = build_edge_tiles(graph, 16, 4, 4, order=od, dst_offset=dst_offset)
        else:
            t = build_edge_tiles(graph, rows_per_tile, 4, 8, order=od, dst_offset=dst_offset)
        cache[key] = t
    return t


class SegPlan:
    """struct tmpnn_seg_plan (include/tmpnn.h) with the tensors it points at."""

    def __init__(self, T, I, nsplit, t_row, items, rowptr2, inc2, ws):
        self.T, self.I, self.nsplit = int(T), int(I), int(nsplit)
        self.t_row, self.items, self.rowptr2, self.inc2, self.ws = t_row, items, rowptr2, inc2, ws
        self.c = _lib.CSegPlan(self.T, self.I, self.nsplit, t_row.data_ptr(), items.data_ptr(), rowptr2.data_ptr(),
                               inc2.data_ptr(), ws.data_ptr(), ws.numel())


SEG_ITEM_TILES = 13          # tiles per work item of the dense segment sum (a 300-src frame block: 37 tiles -> 13 + 12 + 12)


def build_seg_plan(graph: FrameGraph, item_tiles: int = SEG_ITEM_TILES, min_fill: float = 0.5) -> Optional[SegPlan]:
    """The single-read segment sum's plan of a DENSE graph (struct tmpnn_seg_plan; csrc/agg.hip k_segsum_tiles).

    A frame block of the rolling graph is a dense [A srcs x D dsts] set of edge rows (reference/utils/graph.py:141-156,
    285-301).  Over det INDICES, the cell (dst // 16, src // 8) of an edge is its TILE and (src % 8, dst % 16) its slot in
    the tile's [8][16] row list (-1: no such edge -- block borders, deleted rows); the workgroup that streams a tile sums it.
    Consecutive tiles of one dst group form work items of at most `item_tiles` tiles whose 16 dst sums stay in registers.  The
    second pass's CSR lists, per det, its partial rows: one per tile it is a src of, one per item it is a dst of.  Returns
    None where the tiles would be less than `min_fill` full (ragged graphs: k_segsum_pipe stays the better kernel) or an edge
    is listed twice.  Index plumbing only, on the graph's device; two host reads (the tile / item counts size the arrays)."""
    if graph.src_pos is None or graph.dst_pos is None or graph.E == 0:
        return None
    dev = graph.device
    E, N, Dn = graph.E, graph.N, graph.Dn
    s, d = graph.src_pos.long(), graph.dst_pos.long()
    ns = Dn // 8 + 1
    cell = (d // 16) * ns + s // 8
    slot = (s % 8) * 16 + d % 16
    ucell, tile_of = torch.unique(cell, return_inverse=True)     # sorted: tiles in (dst group, src group) order
    T = int(ucell.numel())                                       # (host read)
    if E < min_fill * 128 * T:
        return None
    t_row = torch.full((T * 128,), -1, dtype=torch.int32, device=dev)
    flat = tile_of * 128 + slot
    t_row[flat] = graph.edge_row.to(torch.int32)
    dgrp = ucell // ns
    # work items: runs of consecutive tiles of one dst group, cut every `item_tiles`
    new_run = torch.ones(T, dtype=torch.bool, device=dev)
    new_run[1:] = dgrp[1:] != dgrp[:-1]
    run_start = torch.nonzero(new_run).flatten()
    run_id = torch.cumsum(new_run.long(), 0) - 1
    in_run = torch.arange(T, device=dev) - run_start[run_id]
    item_start = torch.nonzero(in_run % int(item_tiles) == 0).flatten()
    # (second host read, with the duplicate check: two edges in one slot would lose one of them)
    I, filled = item_start.numel(), int((t_row >= 0).sum())
    if filled != E:
        return None
    item_cnt = torch.diff(item_start, append=torch.tensor([T], device=dev))
    items = torch.stack([item_start, item_cnt], 1).to(torch.int32).contiguous()
    item_of_tile = torch.cumsum((in_run % int(item_tiles) == 0).long(), 0) - 1
    # second pass: per det its partial rows -- (tile, src i) -> N + 8 t + i ; (item, dst j) -> N + 8 T + 16 k + j
    src_slot = torch.unique(tile_of * 8 + s % 8)                  # the (tile, i) pairs that exist
    src_det = (ucell[src_slot // 8] % ns) * 8 + src_slot % 8
    dst_slot = torch.unique(item_of_tile[tile_of] * 16 + d % 16)  # the (item, j) pairs that exist
    dst_det = dgrp[item_start[dst_slot // 16]] * 16 + dst_slot % 16
    P = 8 * T + 16 * I
    if N + P >= 2 ** 31:
        return None
    det_all = torch.cat([src_det, dst_det])
    ent_all = torch.cat([N + src_slot, N + 8 * T + dst_slot])
    neg_all = torch.cat([torch.zeros_like(src_det, dtype=torch.bool), torch.ones_like(dst_det, dtype=torch.bool)])
    order = torch.argsort(det_all * (N + P + 1) + ent_all)
    inc2 = torch.where(neg_all[order], ent_all[order] - 2 ** 31, ent_all[order]).to(torch.int32).contiguous()
    counts = torch.bincount(det_all, minlength=Dn)
    rowptr2 = torch.zeros(Dn + 1, dtype=torch.long, device=dev)
    rowptr2[1:] = torch.cumsum(counts, 0)
    ws = torch.empty((P * 256,), dtype=torch.float32, device=dev)
    return SegPlan(T, I, N, t_row, items, rowptr2.to(torch.int32).contiguous(), inc2, ws)


def dense_seg_plan(graph: FrameGraph) -> Optional[SegPlan]:
    """The graph's cached single-read segment-sum plan (built on first use; None for ragged graphs), attached to the graph's C
    struct so that tmpnn_segsum_fwd and the wide cells' backward take the dense form on 256-column blocks."""
    d = graph.__dict__
    if '_seg_plan' not in d:
        if graph.E > 0 and graph.src_pos is not None and graph.src_pos.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('dense_seg_plan: the plan of this graph is not built yet and building it reads two counts back '
                               'to the host, which a stream capture cannot do -- run one eager call on the graph before capturing')
        plan = None
        # (dense enough to be worth it: a src's run of edges spans a good part of a tile -- the test edge_tiles() makes)
        if graph.E >= 128 * 64 and graph.E >= 32 * graph.Dn:
            plan = build_seg_plan(graph)
        d['_seg_plan'] = plan
        graph._c = None                                           # (the C struct carries the plan's address)
    return d['_seg_plan']


def set_det_groups(graph: FrameGraph, det_group) -> FrameGraph:
    """Visit the dets group by group in the det -> edge reductions (struct tmpnn_graph, det_order).

    det_group [Dn]: any integer label per det, in det order; dets that share edges should share a label (the window
    id in a block-diagonal batch).  Only the memory access order changes, never a result."""
    dg = torch.as_tensor(det_group, dtype=torch.long, device=graph.device)
    if dg.numel() != graph.Dn:
        raise ValueError('set_det_groups: one label per det expected')
    graph.det_order = torch.argsort(dg, stable=True).to(torch.int32).contiguous()
    graph.__dict__['_det_group'] = dg
    graph.__dict__.pop('_win_plan', None)
    graph._c = None
    return graph


# csrc/agg.hip k_segsum_win: edge rows of a chunk, chunks / dets of a window it serves, lane groups of a workgroup, steps of a chunk
WIN_CH, WIN_MAXCH, WIN_CAP_DETS, WIN_NHG, WIN_LCAP = 160, 24, 160, 128, 15


class WinPlan:
    """struct tmpnn_win_plan (include/tmpnn.h) with the tensors it points at."""

    def __init__(self, W, nbig, wrec, erow, recs, det, drow, big_order):
        self.W, self.nbig = int(W), int(nbig)
        self.t = (wrec, erow, recs, det, drow, big_order)
        assert not wrec.is_cuda or (wrec.data_ptr() % 32 == 0 and erow.data_ptr() % 16 == 0 and recs.data_ptr() % 256 == 0)
        self.c = _lib.CWinPlan(self.W, self.nbig, wrec.data_ptr(), erow.data_ptr(), recs.data_ptr(), det.data_ptr(),
                               drow.data_ptr(), _lib.ptr(big_order))


def build_win_plan(graph: FrameGraph, det_group: torch.Tensor) -> Optional[WinPlan]:
    """The window-owned segment sum's plan (struct tmpnn_win_plan; csrc/agg.hip k_segsum_win) of a block-diagonal batch of small
    windows -- the graphs batch_windows() builds out of the reference's rolling windows (reference/utils/graph.py:141-156: a
    window's edges join dets of that window only).

    A window's edge rows are listed in ascending order and cut into chunks of WIN_CH.  An incidence i of a det's CSR run belongs to
    the STREAM (det, i % 4) -- the lane group of k_segsum_pipe that adds it -- and streams are dealt round the 128 lane groups of a
    workgroup (stream % 128).  Per chunk, every lane group gets the list of its incidences whose edge row lies in the chunk, stream
    by stream and in run order; the lists of a chunk are padded to the longest (`steps`) and stored step-major as 16-bit RECORDS
    (place of the row in the chunk | sign << 8 | stream / 128 << 9 | nothing-to-do << 15).  Returns None when an edge crosses
    windows, a run does not ascend in edge row, or too few edges sit in windows the kernel serves.  Index plumbing on the graph's
    device; two host reads."""
    if graph.src_pos is None or graph.dst_pos is None or graph.E == 0 or graph.Dn == 0:
        return None
    dev, E, Dn = graph.device, graph.E, graph.Dn
    CH, NHG = WIN_CH, WIN_NHG
    i32 = lambda t: t.to(torch.int32).contiguous()
    ar = lambda n: torch.arange(n, device=dev)
    uw, wdet = torch.unique(det_group, return_inverse=True)
    s, d = graph.src_pos.long(), graph.dst_pos.long()
    wedge = wdet[s]
    rowptr, inc = graph.rowptr.long(), graph.inc.long()
    # (a run of the CSR ascends in edge row -- graph_from_edges keeps it so: a stream then meets the chunks in order)
    rows_p = inc & 0x7fffffff
    starts = torch.zeros(2 * E, dtype=torch.bool, device=dev)
    starts[rowptr[:-1][rowptr[:-1] < 2 * E]] = True
    unsorted_run = ((rows_p[1:] <= rows_p[:-1]) & ~starts[1:]).any()
    W, bad = int(uw.numel()), bool((wedge != wdet[d]).any() | unsorted_run)   # (host read)
    if bad:
        return None
    order = torch.argsort(wdet, stable=True)                               # visiting position -> det index
    vis = torch.empty_like(order)
    vis[order] = ar(Dn)
    nd = torch.bincount(wdet, minlength=W)
    dptr = torch.zeros(W + 1, dtype=torch.long, device=dev)
    dptr[1:] = torch.cumsum(nd, 0)
    ne = torch.bincount(wedge, minlength=W)
    eptr = torch.zeros(W + 1, dtype=torch.long, device=dev)               # a window's list starts at a multiple of 4 (16 bytes)
    eptr[1:] = torch.cumsum((ne + 3) // 4 * 4, 0)
    eorder = torch.argsort(wedge, stable=True)                             # edge indices window by window, rows ascending
    cum = torch.zeros(W + 1, dtype=torch.long, device=dev)
    cum[1:] = torch.cumsum(ne, 0)
    local = torch.empty(E, dtype=torch.long, device=dev)                   # edge index -> its place in its window's list
    local[eorder] = ar(E) - cum[wedge[eorder]]
    erow = torch.zeros(int(((ne + 3) // 4 * 4).sum()) + 4, dtype=torch.long, device=dev)
    erow[eptr[wedge] + local] = graph.edge_row.long()
    nch = (ne + CH - 1) // CH
    cbase = torch.zeros(W + 1, dtype=torch.long, device=dev)
    cbase[1:] = torch.cumsum(nch, 0)
    # per incidence (CSR position p of det dd): stream, lane group, chunk
    deg = rowptr[1:] - rowptr[:-1]
    dd = torch.repeat_interleave(ar(Dn), deg)
    i = ar(2 * E) - rowptr[dd]
    wp_ = wdet[dd]
    sid = (vis[dd] - dptr[wp_]) * 4 + (i & 3)
    hg, j, seq = sid % NHG, sid // NHG, i >> 2
    eloc = local[graph.pos.long()[rows_p]]
    gc = cbase[wp_] + eloc // CH
    ok = (nd[wp_] <= WIN_CAP_DETS) & (nch[wp_] <= WIN_MAXCH)                 # (the others: CSR kernel)
    key = (((gc * NHG + hg) * 8 + j.clamp(max=7)) << 12) + seq.clamp(max=4095)
    key = torch.where(ok, key, torch.full_like(key, 2 ** 62))
    perm = torch.argsort(key, stable=True)
    cell = (gc * NHG + hg)[perm]
    ncell = int(cbase[-1]) * NHG                                           # (host read, with the two below)
    cnt = torch.bincount(cell[ok[perm]], minlength=ncell)
    cstart = torch.zeros(ncell + 1, dtype=torch.long, device=dev)
    cstart[1:] = torch.cumsum(cnt, 0)
    step = ar(2 * E) - cstart[cell.clamp(max=ncell - 1)]                   # (only the ok incidences, sorted first, are used)
    steps = cnt.view(-1, NHG).max(1).values if ncell else torch.zeros(0, dtype=torch.long, device=dev)   # per chunk
    # windows whose longest list is beyond what 4 bits / the LDS hold go to the CSR kernel as well
    wofc = torch.repeat_interleave(ar(W), nch)
    too_long = torch.zeros(W, dtype=torch.bool, device=dev)
    too_long[wofc[steps > WIN_LCAP]] = True
    big_w = (nd > WIN_CAP_DETS) | (nch > WIN_MAXCH) | too_long
    steps = torch.where(big_w[wofc], torch.zeros_like(steps), steps)
    big_order = order[big_w[wdet[order]]]
    nbig, e_big = int(big_order.numel()), int(ne[big_w].sum())             # (host read)
    if 2 * e_big > E:
        return None
    soff = torch.zeros(steps.numel() + 1, dtype=torch.long, device=dev)    # first step of a chunk in the record array
    soff[1:] = torch.cumsum(steps, 0)
    recs = torch.full(((int(soff[-1]) + 4) * NHG,), -2 ** 15, dtype=torch.int16, device=dev)      # 0x8000: nothing to do
    use = ok[perm] & ~big_w[wp_[perm]]
    pp = perm[use]
    val = (eloc[pp] % CH) | torch.where(inc[pp] < 0, 0x100, 0) | (j[pp] << 9)
    recs[(soff[gc[pp]] + step[use]) * NHG + hg[pp]] = val.to(torch.int16)
    wrec = torch.zeros(W, 8, dtype=torch.long, device=dev)
    wrec[:, 0], wrec[:, 1], wrec[:, 2], wrec[:, 3] = eptr[:-1], ne, dptr[:-1], nd
    wrec[:, 4] = soff[cbase[:-1]]
    cidx = ar(steps.numel()) - cbase[wofc]                                 # chunk index within its window
    fits = cidx < WIN_MAXCH
    wrec.view(-1).index_add_(0, (wofc * 8 + 5 + cidx // 8)[fits], (steps << (4 * (cidx % 8)))[fits])
    wrec[big_w, 3] = WIN_CAP_DETS + 1 + 0 * wrec[big_w, 3]                 # (what the kernel reads as "not mine")
    return WinPlan(W, nbig, i32(wrec), i32(erow), recs, i32(order), i32(graph.det_row.long()[order]),
                   i32(big_order) if nbig else None)


def win_plan(graph: FrameGraph) -> Optional[WinPlan]:
    """The graph's cached window plan (None without det groups or where build_win_plan declines), attached to the graph's C
    struct: the segment sums at H = 32 / 64 then read every edge row once.  Not built inside a stream capture (host reads):
    there the call returns None without caching and the CSR kernel serves the graph."""
    d = graph.__dict__
    if '_win_plan' not in d:
        dg = d.get('_det_group')
        if dg is None:
            return None
        if dg.is_cuda and torch.cuda.is_current_stream_capturing():
            return None
        d['_win_plan'] = build_win_plan(graph, dg)
        graph._c = None                                           # (the C struct carries the plan's address)
    return d['_win_plan']


def graph_from_edges(N: int, is_edge: torch.Tensor, src: torch.Tensor, dst: torch.Tensor,
                     device=None) -> FrameGraph:
    """Build the FrameGraph from the type mask and the per-edge (src, dst) det rows.

    `src`/`dst` are given per edge in ascending edge-row order.  Pure index plumbing (torch ops,
    any device); the CSR keeps, for each det, its incidences in ascending edge-row order.
    """
    is_edge = torch.as_tensor(is_edge).to(torch.bool)
    dev = device if device is not None else is_edge.device
    is_edge = is_edge.to(dev)
    src = torch.as_tensor(src).to(dev).long()
    dst = torch.as_tensor(dst).to(dev).long()
    edge_row = torch.nonzero(is_edge).flatten()
    det_row = torch.nonzero(~is_edge).flatten()
    E, Dn = int(edge_row.numel()), int(det_row.numel())
    if src.numel() != E or dst.numel() != E:
        raise ValueError(f'need one (src, dst) per edge row: E={E}, got {src.numel()}/{dst.numel()}')
    det_pos = torch.full((N,), -1, dtype=torch.long, device=dev)
    det_pos[det_row] = torch.arange(Dn, device=dev)
    pos = det_pos.clone()
    pos[edge_row] = torch.arange(E, device=dev)
    if E > 0:
        # both invariants in ONE host round trip (this runs once per forward call of the drop-in API)
        bad = torch.stack([is_edge[src].any() | is_edge[dst].any(),
                           ~((src < edge_row) & (edge_row < dst)).all()]).tolist()
        if bad[0]:
            raise ValueError('edge endpoint is not a det row')
        if bad[1]:
            raise ValueError('expected src row < edge row < dst row (utils/graph.py:153-156,298-301)')
    # incidences: (det index, edge row, sign) ; sorted by det, then by edge row
    d_all = torch.cat([det_pos[src], det_pos[dst]])
    r_all = torch.cat([edge_row, edge_row])
    neg = torch.cat([torch.zeros(E, dtype=torch.bool, device=dev), torch.ones(E, dtype=torch.bool, device=dev)])
    key = d_all * (N + 1) + r_all
    order = torch.argsort(key)
    inc = r_all[order].to(torch.int32)
    inc = torch.where(neg[order], inc | torch.tensor(-2 ** 31, dtype=torch.int32, device=dev), inc)
    counts = torch.bincount(d_all, minlength=Dn) if E > 0 else torch.zeros(Dn, dtype=torch.long, device=dev)
    rowptr = torch.zeros(Dn + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    i32 = lambda t: t.to(torch.int32).contiguous()
    return FrameGraph(N=N, E=E, Dn=Dn, src=i32(src), dst=i32(dst), edge_row=i32(edge_row), det_row=i32(det_row),
                      rowptr=i32(rowptr), inc=inc.contiguous(), is_edge=is_edge.to(torch.uint8).contiguous(),
                      pos=i32(pos), src_pos=i32(det_pos[src]), dst_pos=i32(det_pos[dst]))


def _coo(adj: torch.Tensor):
    if adj.is_sparse:
        adj = adj.coalesce()
        idx, val = adj.indices(), adj.values()
        keep = val != 0
        return idx[0][keep], idx[1][keep], val[keep]
    nz = torch.nonzero(adj)
    return nz[:, 0], nz[:, 1], adj[nz[:, 0], nz[:, 1]]


def graph_from_adjacency(node_adj: torch.Tensor, edge_adj: Optional[torch.Tensor] = None,
                         validate: bool = True) -> FrameGraph:
    """Convert the reference's adjacency pair into a FrameGraph on the same device.

    Accepts dense tensors (first CPU-path call, utils/graph.py:180-184 only sparsifies under
    cuda=True), coalesced COO with explicit zeros (node_adj) and uncoalesced COO (edge_adj).
    Raises ValueError when the factor-graph invariants (one +1 and one -1 per edge row,
    no off-diagonals on det rows, edge_adj = node_adj^T off the diagonal) do not hold.
    """
    N = int(node_adj.shape[0])
    dev = node_adj.device
    r, c, v = _coo(node_adj.detach())
    diag = r == c
    is_det = torch.zeros(N, dtype=torch.bool, device=dev)
    is_det[r[diag]] = True
    is_edge = ~is_det
    ro, co, vo = r[~diag], c[~diag], v[~diag]
    pos, neg = vo > 0, vo < 0
    src = torch.full((N,), -1, dtype=torch.long, device=dev)
    dst = torch.full((N,), -1, dtype=torch.long, device=dev)
    src[ro[pos]] = co[pos]
    dst[ro[neg]] = co[neg]
    edge_row = torch.nonzero(is_edge).flatten()
    if validate:
        E = int(edge_row.numel())
        # every condition as a 0-dim tensor, fetched with one host round trip
        conds = [pos.sum() == E, neg.sum() == E, (vo.abs() == 1).all(), ~is_det[ro].any(),
                 (src[edge_row] >= 0).all(), (dst[edge_row] >= 0).all()]
        if edge_adj is not None:
            # edge_adj must be node_adj^T off the diagonal: entry (det d, edge e, v) <=> d is the +1 (v > 0) or the
            # -1 (v < 0) det of e, and there are exactly 2E of them -- elementwise, no sorting
            r2, c2, v2 = _coo(edge_adj.detach())
            d2 = r2 == c2
            ie = torch.zeros(N, dtype=torch.bool, device=dev)
            ie[r2[d2]] = True
            ro2, co2, vo2 = r2[~d2], c2[~d2], v2[~d2]
            match = torch.where(vo2 > 0, src[co2] == ro2, dst[co2] == ro2) & (vo2.abs() == 1)
            conds += [(ie == is_edge).all(), match.all(), torch.as_tensor(ro2.numel() == 2 * E, device=dev)]
        ok = torch.stack([c.reshape(()) for c in conds]).tolist()
        if not all(ok[:6]):
            raise ValueError('node_adj is not a TrackMPNN factor graph: every edge row needs exactly one +1 and '
                             'one -1 off-diagonal entry and det rows none')
        if edge_adj is not None:
            if not ok[6]:
                raise ValueError('diag(edge_adj) does not complement diag(node_adj)')
            if not (ok[7] and ok[8]):
                raise ValueError('edge_adj is not node_adj^T off the diagonal')
    return graph_from_edges(N, is_edge, src[edge_row], dst[edge_row], device=dev)


# ----------------------------------------------------------------------------------------------
# train-mode rolling window construction in index form (reference/utils/graph.py:96-186,189-334)
# ----------------------------------------------------------------------------------------------
@dataclass
class WindowCall:
    """What one forward call of one window appends: rows [n_edges new edge rows][n_dets new det rows]."""
    n_new: int
    new_is_edge: np.ndarray     # bool [n_new]
    new_src: np.ndarray         # int64 [n_new_edges] window-local det row
    new_dst: np.ndarray         # int64 [n_new_edges]
    det_ids: np.ndarray         # int64 [n_new_dets] index into the window's detection list (rows of X)


class WindowBuilder:
    """Index-form replay of initialize_graph / update_graph(mode='train') for one chunk.

    y [ND, 2] = [timestep, track id (-1 = false positive)].  Row layout per call, as in the
    reference: `[edges active x new dets, src-major][new dets]` appended after the old rows
    (utils/graph.py:141-156, 285-301); first call: `[dets t0][edges t0 x t1][dets t1]`.
    Active set at time t = dets of the previous non-empty timestep + earlier true-positive dets
    whose track has no later detection yet (utils/graph.py:229-245, 271-274).
    """

    def __init__(self, y: np.ndarray):
        self.y = np.asarray(y, dtype=np.int64)

    def calls(self) -> List[WindowCall]:
        y = self.y
        times = np.unique(y[:, 0])
        if times.size < 2:
            return []
        out: List[WindowCall] = []
        t0, t1 = int(times[0]), int(times[1])
        ids0 = np.nonzero(y[:, 0] == t0)[0]
        ids1 = np.nonzero(y[:, 0] == t1)[0]
        n0, n1 = ids0.size, ids1.size
        is_edge = np.concatenate([np.zeros(n0, bool), np.ones(n0 * n1, bool), np.zeros(n1, bool)])
        src = np.repeat(np.arange(n0), n1)
        dst = n0 + n0 * n1 + np.tile(np.arange(n1), n0)
        out.append(WindowCall(n0 + n0 * n1 + n1, is_edge, src, dst, np.concatenate([ids0, ids1])))
        # bookkeeping per det row: (row, timestep, det id, track id, associated?)
        det_rows = np.concatenate([np.arange(n0), n0 + n0 * n1 + np.arange(n1)])
        det_ts = np.concatenate([np.full(n0, t0), np.full(n1, t1)])
        det_ids = np.concatenate([ids0, ids1])
        N = n0 + n0 * n1 + n1
        # which det pairs have an edge (for the association rule): set of (src det id, dst det id)
        has_edge = set((int(a), int(b)) for a in ids0 for b in ids1)
        t_prev = t1
        for t in times[2:]:
            t = int(t)
            trk = y[det_ids, 1]
            # y_pred[:, 2] update (utils/graph.py:229-245): a TP det is associated iff one of its
            # FUTURE edges leads to a det of the same track; FPs self-associate (stay inactive)
            assoc = np.zeros(det_ids.size, bool)
            for i in range(det_ids.size):
                if trk[i] < 0:
                    assoc[i] = True
                    continue
                later = np.nonzero((trk == trk[i]) & (det_ts > det_ts[i]))[0]
                assoc[i] = any((int(det_ids[i]), int(det_ids[j])) in has_edge for j in later)
            active = np.nonzero((~assoc) | (det_ts == t_prev))[0]      # utils/graph.py:273-274
            ids_t = np.nonzero(y[:, 0] == t)[0]
            nt, na = ids_t.size, active.size
            n_new = na * nt + nt
            is_edge = np.concatenate([np.ones(na * nt, bool), np.zeros(nt, bool)])
            src = np.repeat(det_rows[active], nt)
            dst = N + na * nt + np.tile(np.arange(nt), na)
            out.append(WindowCall(n_new, is_edge, src, dst, ids_t))
            for a in active:
                for b in ids_t:
                    has_edge.add((int(det_ids[a]), int(b)))
            det_rows = np.concatenate([det_rows, N + na * nt + np.arange(nt)])
            det_ts = np.concatenate([det_ts, np.full(nt, t)])
            det_ids = np.concatenate([det_ids, ids_t])
            N += n_new
            t_prev = t
        return out


def synth_window(seed: int, frames: int, mean_dets: float, max_dets: int, survival: float = 0.9,
                 fp_rate: float = 0.1, dropout: float = 0.2) -> np.ndarray:
    """KITTI/BDD-shaped synthetic chunk: y [ND, 2] = [timestep, track id].

    Tracks are born to keep ~Poisson(mean_dets) alive, survive a frame with prob `survival`, are
    missed by the detector with prob `dropout` (reference/dataset/kitti_mot.py:102,530-532) and
    ~`fp_rate` of the detections are false positives (track id -1).
    """
    rng = np.random.RandomState(seed)
    alive: List[int] = []
    next_id = 0
    rows = []
    for t in range(frames):
        alive = [i for i in alive if rng.rand() < survival]
        target = int(np.clip(rng.poisson(mean_dets), 1, max_dets))
        while len(alive) < target:
            alive.append(next_id)
            next_id += 1
        seen = [i for i in alive if rng.rand() >= dropout]
        if not seen:
            seen = [alive[0]]
        nfp = int(rng.binomial(len(seen), fp_rate))
        ids = (seen + [-1] * nfp)[:max_dets]
        rng.shuffle(ids)
        rows += [(t, i) for i in ids]
    return np.asarray(rows, dtype=np.int64)


@dataclass
class CallPlan:
    """One forward call over a (possibly batched) graph: everything the device needs, resident."""
    graph: FrameGraph
    n_new: int
    new_det_local: torch.Tensor   # int64 [nd]  index into x (the call's new rows) of the new det rows
    new_det_row: torch.Tensor     # int32 [nd]  global row of each new det
    seg_ptr: torch.Tensor         # int32 [S+1] new det rows of window s
    seg_cnt: torch.Tensor         # int32 [S]   ALL new rows of window s
    seg_of_new: torch.Tensor      # int64 [n_new] window of each new row
    min_seg_cnt: int
    seg_of_det: Optional[torch.Tensor] = None   # int32 [nd] window of each new det row
    max_seg_nd: int = -1                        # most new det rows of one window (-1: unknown -> staged input transform)

    @property
    def S(self) -> int:
        return int(self.seg_cnt.numel())

    def to(self, device) -> 'CallPlan':
        return CallPlan(self.graph.to(device), self.n_new, self.new_det_local.to(device),
                        self.new_det_row.to(device), self.seg_ptr.to(device), self.seg_cnt.to(device),
                        self.seg_of_new.to(device), self.min_seg_cnt,
                        None if self.seg_of_det is None else self.seg_of_det.to(device), self.max_seg_nd)


def plan_single(graph: FrameGraph, n_new: int) -> CallPlan:
    """CallPlan of an unbatched reference-style call: the last n_new rows are new, one segment."""
    dev = graph.device
    N = graph.N
    new_is_det = graph.is_edge[N - n_new:] == 0 if n_new > 0 else torch.zeros(0, dtype=torch.bool, device=dev)
    loc = torch.nonzero(new_is_det).flatten()
    nd = int(loc.numel())
    return CallPlan(graph=graph, n_new=n_new, new_det_local=loc, new_det_row=(loc + (N - n_new)).to(torch.int32),
                    seg_ptr=torch.tensor([0, nd], dtype=torch.int32, device=dev),
                    seg_cnt=torch.tensor([n_new], dtype=torch.int32, device=dev),
                    seg_of_new=torch.zeros(n_new, dtype=torch.long, device=dev), min_seg_cnt=n_new,
                    seg_of_det=torch.zeros(nd, dtype=torch.int32, device=dev), max_seg_nd=nd)



def dense_static_graph(T: int, D: int, device='cpu') -> FrameGraph:
    """The final graph of a window in which every frame holds D true-positive dets seen in every frame: the active
    set is always the previous frame (utils/graph.py:271-274), so consecutive frames are fully connected and the row
    order is [dets t0][D*D edges, src-major][dets t1]...  (BASELINE.json C1: T=5, D=20; C5: T=50, D=300)."""
    N = T * D + (T - 1) * D * D
    is_edge = np.zeros(N, bool)
    src = np.empty((T - 1) * D * D, np.int64)
    dst = np.empty_like(src)
    row = e = 0
    prev = None
    for t in range(T):
        if t > 0:
            is_edge[row:row + D * D] = True
            src[e:e + D * D] = np.repeat(prev, D)
            dst[e:e + D * D] = row + D * D + np.tile(np.arange(D), D)
            row += D * D
            e += D * D
        prev = row + np.arange(D)
        row += D
    return graph_from_edges(N, torch.from_numpy(is_edge), torch.from_numpy(src), torch.from_numpy(dst), device=device)


def concat_static_graphs(graphs: Sequence[FrameGraph], device='cpu') -> Tuple[FrameGraph, 'CallPlan']:
    """Block-diagonal union of whole graphs presented in ONE call (all rows new, one BatchNorm segment per graph)."""
    is_edge, src, dst, seg_cnt, seg_nd = [], [], [], [], []
    off = 0
    for g in graphs:
        is_edge.append(g.is_edge.cpu().numpy().astype(bool))
        src.append(g.src.cpu().numpy().astype(np.int64) + off)
        dst.append(g.dst.cpu().numpy().astype(np.int64) + off)
        seg_cnt.append(g.N)
        seg_nd.append(g.Dn)
        off += g.N
    ie = np.concatenate(is_edge)
    graph = graph_from_edges(off, torch.from_numpy(ie), torch.from_numpy(np.concatenate(src)),
                             torch.from_numpy(np.concatenate(dst)), device=device)
    seg_ids = np.repeat(np.arange(len(graphs)), seg_cnt)
    loc = np.nonzero(~ie)[0]
    plan = CallPlan(graph=graph, n_new=off, new_det_local=torch.from_numpy(loc).to(device),
                    new_det_row=torch.from_numpy(loc.astype(np.int32)).to(device),
                    seg_ptr=torch.from_numpy(np.concatenate([[0], np.cumsum(seg_nd)]).astype(np.int32)).to(device),
                    seg_cnt=torch.from_numpy(np.asarray(seg_cnt, np.int32)).to(device),
                    seg_of_new=torch.from_numpy(seg_ids).to(device), min_seg_cnt=int(min(seg_cnt)),
                    seg_of_det=torch.from_numpy(seg_ids[loc].astype(np.int32)).to(device), max_seg_nd=int(max(seg_nd)))
    return graph, plan


def batch_windows(windows: Sequence[Sequence[WindowCall]], static: bool = False,
                  device='cpu') -> Tuple[List[CallPlan], List[np.ndarray]]:
    """Block-diagonal batch of many windows, rows in CALL-MAJOR order.

    Call c of the batch appends, for every window b in turn, the rows window b would append at its
    call c -- so the append-only contract of `forward(x, h_in, ...)` (N' = N + n) holds for the
    batch and no row of an earlier call ever moves.  Returns one CallPlan per call plus, per call,
    the (window, det id) of every new det row (to fetch features).  `static=True` collapses each
    window to its final graph presented in ONE call (SURVEY 8(d) static mode).
    """
    B = len(windows)
    ncalls = max(len(w) for w in windows)
    base = [dict() for _ in range(B)]           # window-local row -> global row, per window as arrays
    local2global = [np.zeros(0, np.int64) for _ in range(B)]
    plans: List[CallPlan] = []
    det_refs: List[np.ndarray] = []
    N = 0
    g_is_edge: List[np.ndarray] = []
    g_src: List[np.ndarray] = []
    g_dst: List[np.ndarray] = []
    g_det_window: List[np.ndarray] = []         # window of every det row, in det order
    call_range = [range(ncalls)] if static else [[c] for c in range(ncalls)]
    for group in call_range:
        seg_cnt, seg_nd, refs, new_is_edge_all, seg_ids = [], [], [], [], []
        n_before = N
        for b, w in enumerate(windows):
            cnt = nd = 0
            for c in group:
                if c >= len(w):
                    continue
                wc = w[c]
                l2g = np.concatenate([local2global[b], N + np.arange(wc.n_new)])
                local2global[b] = l2g
                g_is_edge.append(wc.new_is_edge)
                g_src.append(l2g[wc.new_src])
                g_dst.append(l2g[wc.new_dst])
                new_is_edge_all.append(wc.new_is_edge)
                refs.append(np.stack([np.full(wc.det_ids.size, b), wc.det_ids], 1))
                g_det_window.append(np.full(int((~wc.new_is_edge).sum()), b, np.int64))
                N += wc.n_new
                cnt += wc.n_new
                nd += int((~wc.new_is_edge).sum())
            if cnt > 0:
                seg_cnt.append(cnt)
                seg_nd.append(nd)
                seg_ids.append(np.full(cnt, len(seg_cnt) - 1))
        is_edge = np.concatenate(g_is_edge) if g_is_edge else np.zeros(0, bool)
        src = np.concatenate(g_src) if g_src else np.zeros(0, np.int64)
        dst = np.concatenate(g_dst) if g_dst else np.zeros(0, np.int64)
        # edges must be listed in ascending edge-row order: they are, rows are appended in order
        graph = graph_from_edges(N, torch.from_numpy(is_edge), torch.from_numpy(src), torch.from_numpy(dst),
                                 device=device)
        if B > 1:
            # visit each window's dets together: both endpoint reads of an edge row then come from one CU within a
            # window's ~100 KB working set instead of from two CUs a whole call block apart
            set_det_groups(graph, np.concatenate(g_det_window) if g_det_window else np.zeros(0, np.int64))
        n_new = N - n_before
        nie = np.concatenate(new_is_edge_all) if new_is_edge_all else np.zeros(0, bool)
        loc = np.nonzero(~nie)[0]
        seg_ptr = np.concatenate([[0], np.cumsum(seg_nd)]).astype(np.int32)
        plans.append(CallPlan(
            graph=graph, n_new=n_new,
            new_det_local=torch.from_numpy(loc).to(device),
            new_det_row=torch.from_numpy((loc + n_before).astype(np.int32)).to(device),
            seg_ptr=torch.from_numpy(seg_ptr).to(device),
            seg_cnt=torch.from_numpy(np.asarray(seg_cnt, dtype=np.int32)).to(device),
            seg_of_new=torch.from_numpy(np.concatenate(seg_ids) if seg_ids else np.zeros(0, np.int64)).to(device),
            min_seg_cnt=int(min(seg_cnt)) if seg_cnt else 0,
            seg_of_det=torch.from_numpy((np.concatenate(seg_ids)[loc] if seg_ids else np.zeros(0)).astype(np.int32)).to(device),
            max_seg_nd=int(max(seg_nd)) if seg_nd else 0))
        det_refs.append(np.concatenate(refs) if refs else np.zeros((0, 2), np.int64))
    return plans, det_refs


# ----------------------------------------------------------------------------------------------
# batch-1 path: graphs whose sizes stay on the device (struct tmpnn_dgraph, include/tmpnn.h)
# ----------------------------------------------------------------------------------------------
DG_MAX_ROWS = 4096          # TMPNN_DG_MAX_ROWS: conversion work arrays in LDS, tracker-side operations
DG_BIG_ROWS = 65535         # TMPNN_DG_BIG_ROWS: fused iteration, conversion with its work arrays in a global scratch
_DG_STATUS = ((1, 'an off-diagonal adjacency entry is not +-1 (or an index is out of range)'),
              (2, 'node_adj is not a TrackMPNN factor graph: every edge row needs exactly one +1 and one -1 '
                  'off-diagonal entry and det rows none'),
              (4, 'edge endpoint is not a det row'),
              (8, 'expected src row < edge row < dst row (utils/graph.py:153-156,298-301)'),
              (16, 'diag(edge_adj) does not complement diag(node_adj)'),
              (32, 'edge_adj is not node_adj^T off the diagonal'),
              (64, 'x has non-zero features on new EDGE rows: the reference would feed them to the BatchNorm statistics '
                   '(utils/graph.py:148,291 always passes zeros); this implementation reads det rows only'))


class DeviceGraph:
    """Index-form graph of ONE call whose sizes (E, Dn) and validation status live in device memory.

    Produced by `tmpnn_graph_from_coo` in one launch straight from the reference's adjacency tensors; consumed by
    the fused iteration (`tmpnn_mp_iter_fwd/_bwd`) without the host ever reading E or Dn, so a forward call does not
    synchronise.  The factor-graph validation therefore reports LATE: `status()` / `check()` read it back (one host
    round trip) whenever the caller chooses -- `TrackMPNN` does so before the first backward of a chunk, every 64
    calls, or immediately with TMPNN_STRICT_GRAPH=1.  An invalid graph is presented to the kernels as empty."""

    def __init__(self, N: int, device, cap: Optional[int] = None):
        self.N = int(N)
        self.cap = cap = int(cap if cap is not None else N)
        # the layout of tmpnn_dgraph_bind, restated (tests/test_abi.py checks it against the library)
        c = (cap + 1 + 3) & ~3
        nbytes = (((cap + 3) // 4) + 3) & ~3
        self.arena = torch.empty((8 + nbytes + 10 * c,), dtype=torch.int32, device=device)
        self._layout = (nbytes, c)
        self._c = None
        self._meta = None
        self._frame = None

    @classmethod
    def from_arena(cls, N: int, arena: torch.Tensor, meta=None) -> 'DeviceGraph':
        """A DeviceGraph over an arena some native caller allocated and filled (tmpnn_dgraph_ints(N) int32)."""
        self = cls.__new__(cls)
        self.N = self.cap = int(N)
        c = (self.cap + 1 + 3) & ~3
        nbytes = (((self.cap + 3) // 4) + 3) & ~3
        assert arena.numel() >= 8 + nbytes + 10 * c
        self.arena, self._layout, self._c, self._meta, self._frame = arena, (nbytes, c), None, meta, None
        return self

    @property
    def c(self) -> '_lib.CDGraph':
        """struct tmpnn_dgraph over the arena (built on first use: the fused C++ node re-binds the arena itself)."""
        if self._c is None:
            nbytes, c = self._layout
            b = self.arena.data_ptr()
            o = b + 32 + 4 * nbytes
            self._c = _lib.CDGraph(self.N, self.cap, b, b + 32, o, o + 4 * c, o + 8 * c, o + 12 * c, o + 16 * c, o + 20 * c,
                                   o + 24 * c, o + 28 * c, o + 32 * c)
        return self._c

    @property
    def device(self):
        return self.arena.device

    def cref(self):
        return C.byref(self.c)

    def _view(self, ptr, n, dtype=torch.int32):
        off = (int(ptr) - self.arena.data_ptr()) // 4
        t = self.arena[off:off + (n if dtype == torch.int32 else (n + 3) // 4)]
        return t if dtype == torch.int32 else t.view(torch.uint8)[:n]

    def meta(self):
        """(E, Dn, status) -- synchronises with the stream that built the graph (cached afterwards)."""
        if self._meta is None:
            m = self.arena[:8].tolist()
            self._meta = (m[4], m[5], m[2])
        return self._meta

    def status(self) -> int:
        return self.meta()[2]

    def check(self) -> 'DeviceGraph':
        st = self.status()
        if st:
            raise ValueError('; '.join(msg for bit, msg in _DG_STATUS if st & bit))
        return self

    @property
    def E(self) -> int:
        return self.meta()[0]

    @property
    def Dn(self) -> int:
        return self.meta()[1]

    def frame_graph(self) -> FrameGraph:
        """The classic FrameGraph (host-known sizes) as views into the arena; validates first."""
        if self._frame is None:
            self.check()
            E, Dn, _ = self.meta()
            c = self.c
            self._frame = FrameGraph(N=self.N, E=E, Dn=Dn, src=self._view(c.src, E), dst=self._view(c.dst, E),
                                     edge_row=self._view(c.edge_row, E), det_row=self._view(c.det_row, Dn),
                                     rowptr=self._view(c.rowptr, Dn + 1), inc=self._view(c.inc, 2 * E),
                                     is_edge=self._view(c.is_edge, self.N, torch.uint8), pos=self._view(c.pos, self.N),
                                     src_pos=self._view(c.src_pos, E), dst_pos=self._view(c.dst_pos, E))
        return self._frame


def _coo_parts(adj: torch.Tensor, device):
    """(indices int64 [2, nnz] contiguous, values fp32 [nnz]) of a dense or sparse-COO adjacency, on `device`;
    sparse tensors are taken as stored (uncoalesced is fine: the converter sums duplicate diagonal entries)."""
    if adj.is_sparse:
        idx, val = adj._indices(), adj._values()
        if idx.device == device and val.dtype == torch.float32 and idx.is_contiguous() and val.is_contiguous():
            return idx, val                      # the common case (utils/graph.py hands over sparse CUDA tensors)
    else:
        adj = adj.detach()
        idx = torch.nonzero(adj).t()
        val = adj[idx[0], idx[1]]
    idx = idx.detach().to(device=device, dtype=torch.int64)
    val = val.detach().to(device=device, dtype=torch.float32)
    return (idx if idx.is_contiguous() else idx.contiguous()), (val if val.is_contiguous() else val.contiguous())


_f_from_coo = None


def _from_coo():
    global _f_from_coo
    if _f_from_coo is None:
        _f_from_coo = _lib.fn('tmpnn_graph_from_coo_arena')
    return _f_from_coo


def device_graph_from_adjacency(node_adj: torch.Tensor, edge_adj: Optional[torch.Tensor], device) -> DeviceGraph:
    """One-launch conversion of the reference's adjacency pair (N <= DG_BIG_ROWS) on `device`; no host round trip
    for sparse inputs (a dense input costs the `nonzero` that sparsifies it)."""
    N = int(node_adj.shape[0])
    if N > DG_BIG_ROWS:
        raise ValueError(f'device_graph_from_adjacency: N={N} > {DG_BIG_ROWS}; use graph_from_adjacency')
    device = torch.device(device)
    g = DeviceGraph(N, device)
    nidx, nval = _coo_parts(node_adj, device)
    if edge_adj is not None:
        eidx, eval_ = _coo_parts(edge_adj, device)
        ep, ev, en = eidx.data_ptr(), eval_.data_ptr(), int(eval_.numel())
    else:
        eidx = eval_ = None
        ep, ev, en = None, None, 0
    if N <= DG_MAX_ROWS:
        ws = None
        rc = _from_coo()(N, nidx.data_ptr(), nval.data_ptr(), int(nval.numel()), ep, ev, en, g.arena.data_ptr(), g.cap,
                         _lib.raw_stream(device))
    else:                                     # dense scene: the work arrays (8 N + 1 ints) in a global scratch
        ws = torch.empty((8 * N + 1,), dtype=torch.int32, device=device)
        rc = _lib.fn('tmpnn_graph_from_coo_arena_ws')(N, nidx.data_ptr(), nval.data_ptr(), int(nval.numel()), ep, ev, en,
                                                      g.arena.data_ptr(), g.cap, ws.data_ptr(), ws.numel(),
                                                      _lib.raw_stream(device))
    if rc:
        raise RuntimeError(f'tmpnn_graph_from_coo_arena failed (code {rc}): {_lib.last_error()}')
    g._keep = (nidx, nval, eidx, eval_, ws)   # until the launch has consumed them (freed with the graph)
    return g
