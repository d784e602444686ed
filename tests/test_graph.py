"""Host-side graph logic: adjacency ingestion, the index-form window builder, block-diagonal batching."""
import numpy as np
import pytest
import torch

from oracle import trackmpnn_oracle as orc
from tests.conftest import golden_names
from tests.golden_util import Golden
from trackmpnn_amd import (TrackMPNN, WindowBuilder, batch_windows, graph_from_adjacency, graph_from_edges,
                           plan_single, synth_window)


def as_oracle_graph(g):
    return orc.OracleGraph(g.N, g.is_edge.cpu().numpy().astype(bool), g.src.cpu().numpy().astype(np.int64),
                           g.dst.cpu().numpy().astype(np.int64), g.edge_row.cpu().numpy().astype(np.int64),
                           g.det_row.cpu().numpy().astype(np.int64))


@pytest.mark.parametrize('name', golden_names())
def test_adjacency_ingestion_matches_oracle(name):
    gold = Golden(name)
    for c in range(gold.ncalls):
        na, ea = gold.adjacency(c, 'node_adj'), gold.adjacency(c, 'edge_adj')
        g = graph_from_adjacency(na, ea)
        o = orc.graph_from_adjacency(na, ea)
        assert (g.N, g.E, g.Dn) == (o.N, o.E, o.Dn)
        assert np.array_equal(g.src.numpy(), o.src) and np.array_equal(g.dst.numpy(), o.dst)
        assert np.array_equal(g.edge_row.numpy(), o.edge_row) and np.array_equal(g.det_row.numpy(), o.det_row)
        # CSR: per det, incident edge rows ascending, sign bit = det is the later endpoint
        rowptr, inc = g.rowptr.numpy(), g.inc.numpy()
        assert rowptr[0] == 0 and rowptr[-1] == 2 * g.E
        for d, drow in enumerate(o.det_row):
            rows = inc[rowptr[d]:rowptr[d + 1]] & 0x7FFFFFFF
            neg = inc[rowptr[d]:rowptr[d + 1]] < 0
            exp_pos = o.edge_row[o.src == drow]
            exp_neg = o.edge_row[o.dst == drow]
            assert np.array_equal(np.sort(rows[~neg]), exp_pos) and np.array_equal(np.sort(rows[neg]), exp_neg)
            assert np.all(np.diff(rows) > 0)
        assert np.array_equal(g.pos.numpy()[o.det_row], np.arange(o.Dn))
        assert np.array_equal(g.pos.numpy()[o.edge_row], np.arange(o.E))


def test_adjacency_invariants_rejected():
    a = torch.zeros(3, 3)
    a[0, 0] = a[2, 2] = 1
    a[1, 0] = 1                         # edge row with only a +1
    with pytest.raises(ValueError):
        graph_from_adjacency(a)
    b = torch.zeros(3, 3)
    b[0, 0] = b[2, 2] = 1
    b[1, 0], b[1, 2] = 1, -1
    e = b.t().clone()
    e[0, 0] = e[2, 2] = 0
    e[1, 1] = 1
    graph_from_adjacency(b, e)          # valid
    e[0, 1] = -1                        # sign flipped
    with pytest.raises(ValueError):
        graph_from_adjacency(b, e)


def test_empty_and_edgeless_graphs():
    g = graph_from_edges(0, torch.zeros(0, dtype=torch.bool), torch.zeros(0), torch.zeros(0))
    assert (g.N, g.E, g.Dn) == (0, 0, 0)
    g = graph_from_adjacency(torch.eye(4))
    assert (g.N, g.E, g.Dn) == (4, 0, 4) and g.rowptr.tolist() == [0] * 5


@pytest.mark.parametrize('name', [n for n in golden_names() if n.startswith('roll_')])
def test_window_builder_matches_reference_graphs(name):
    """The index-form replay of initialize_graph/update_graph(train) builds the reference's graphs."""
    gold = Golden(name)
    y = gold.d['y'][0]
    calls = WindowBuilder(y).calls()
    assert len(calls) == gold.ncalls - 1          # the fixture's last call is an empty-x iteration
    plans, refs = batch_windows([calls])
    for c, plan in enumerate(plans):
        g = graph_from_adjacency(gold.adjacency(c, 'node_adj'), gold.adjacency(c, 'edge_adj'))
        for f in ('src', 'dst', 'edge_row', 'det_row', 'rowptr', 'inc'):
            assert torch.equal(getattr(plan.graph, f), getattr(g, f)), (c, f)
        x = gold.t(f'c{c}/x')
        assert plan.n_new == x.shape[0]
        X = torch.from_numpy(gold.d['X'][0])
        assert torch.equal(x[plan.new_det_local], X[refs[c][:, 1]])
        ps = plan_single(g, x.shape[0])
        assert torch.equal(ps.new_det_local, plan.new_det_local) and torch.equal(ps.new_det_row, plan.new_det_row)


def _run_oracle(cfg, p, plans, xs, training):
    h = None
    outs = []
    for plan, x in zip(plans, xs):
        s, l, h, _ = orc.forward(p, cfg, x, h, as_oracle_graph(plan.graph), training=training,
                                 seg_ids=plan.seg_of_new)
        outs.append((s, l, h))
    return outs


@pytest.mark.parametrize('static', [False, True])
def test_block_diagonal_batch_equals_per_window(static):
    """Batching B windows call-major with per-window BatchNorm segments changes no per-window result."""
    cfg = orc.OracleConfig('2d', 3, 32, 0, 'diff')
    B = 5
    ys = [synth_window(seed=s, frames=5, mean_dets=3, max_dets=6) for s in range(B)]
    wins = [WindowBuilder(y).calls() for y in ys]
    Xs = [torch.randn(y.shape[0], 8, generator=torch.Generator().manual_seed(100 + i)) for i, y in enumerate(ys)]
    plans, refs = batch_windows(wins, static=static)
    xs = []
    for plan, ref in zip(plans, refs):
        x = torch.zeros(plan.n_new, 8)
        x[plan.new_det_local] = torch.stack([Xs[b][i] for b, i in ref]) if len(ref) else x[:0]
        xs.append(x)
    p = orc.random_params(cfg, seed=3)
    batched = _run_oracle(cfg, {k: v.clone() for k, v in p.items()}, plans, xs, True)
    # per window, rows of window b inside the batch: recover through a marker pass
    for b in range(B):
        plans_b, refs_b = batch_windows([wins[b]], static=static)
        xs_b = []
        for plan, ref in zip(plans_b, refs_b):
            x = torch.zeros(plan.n_new, 8)
            x[plan.new_det_local] = Xs[b][ref[:, 1]]
            xs_b.append(x)
        single = _run_oracle(cfg, {k: v.clone() for k, v in p.items()}, plans_b, xs_b, True)
        # rows of window b in the batched layout, call by call
        rows = []
        for c, plan in enumerate(plans):
            seg = plan.seg_of_new
            # segments are numbered by order of appearance among windows that have rows in this call
            present = [w for w in range(B) if (static or c < len(wins[w]))]
            if b in present:
                s_id = present.index(b)
                new_rows = (plan.graph.N - plan.n_new) + torch.nonzero(seg == s_id).flatten()
                rows.append(new_rows)
            if static or c < len(plans_b):
                allrows = torch.cat(rows)
                cb = 0 if static else c
                assert torch.allclose(batched[c][1][allrows], single[cb][1], atol=2e-5, rtol=1e-5)
                assert torch.allclose(batched[c][2][allrows], single[cb][2], atol=2e-5, rtol=1e-5)


def test_batch_sets_window_major_det_order():
    """batch_windows lists each window's dets together (struct tmpnn_graph, det_order): a permutation of the dets
    whose window labels are non-decreasing; single graphs carry no order."""
    from trackmpnn_amd import WindowBuilder, batch_windows, synth_window
    from trackmpnn_amd.graph import set_det_groups
    wins = [WindowBuilder(synth_window(s, 5, 4, 8)).calls() for s in range(3)]
    plans, det_refs = batch_windows(wins)
    for plan in plans:
        g = plan.graph
        assert g.det_order is not None and g.det_order.dtype == torch.int32
        assert torch.equal(torch.sort(g.det_order.long()).values, torch.arange(g.Dn))
    g = plans[-1].graph
    win_of_det = torch.from_numpy(np.concatenate([r[:, 0] for r in det_refs]))
    lab = win_of_det[g.det_order.long()]
    assert bool((lab[1:] >= lab[:-1]).all())
    single, _ = batch_windows(wins[:1])
    assert single[-1].graph.det_order is None
    with pytest.raises(ValueError):
        set_det_groups(g, np.zeros(g.Dn + 1))


def test_state_dict_keys_match_oracle_shapes():
    for feats, K, msg in (('2d', 0, 'diff'), ('2d+temp+vis', 2, 'concat')):
        m = TrackMPNN(feats, 3, 32, K, msg)
        sd = m.state_dict()
        shapes = orc.param_shapes(orc.OracleConfig(feats, 3, 32, K, msg))
        assert list(sd.keys()) == list(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == shapes[k], k
        assert sorted(m.spec.param_names()) == sorted(k for k, _ in m.named_parameters())
    # widths: 1 .. 1024; above 256 without attention heads only (padded to the next multiple of 128)
    wide = TrackMPNN('2d', 3, 257, 0, 'diff')
    assert wide.hpad == 384 and wide._padded and tuple(wide.state_dict()['factor_grus.0.edge_gru.weight_hh'].shape) == (771, 257)
    assert TrackMPNN('2d', 3, 512, 0, 'diff').hpad == 512
    assert TrackMPNN('2d', 3, 384, 0, 'concat').hpad == 384
    for bad in (dict(nhidden=1025, nattheads=0, msg_type='diff'), dict(nhidden=384, nattheads=2, msg_type='diff')):
        with pytest.raises(ValueError):
            TrackMPNN('2d', 3, bad['nhidden'], bad['nattheads'], bad['msg_type'])
    with pytest.raises(AssertionError):
        TrackMPNN('2d', 3, 32, 0, 'sum')


def test_initial_weights_bit_equal_to_reference():
    """Same torch.nn containers created in the same order => the same RNG stream under torch.manual_seed(5)
    (train.py:42-45,318): the initial state_dict equals the reference's bit for bit (fixture init_seed5, generated
    by oracle/gen_golden.py from the real reference)."""
    import hashlib
    import json
    import os
    from tests.conftest import GOLDEN_DIR
    d = np.load(os.path.join(GOLDEN_DIR, 'init_seed5.npz'))
    meta = json.loads(str(d['meta']))
    assert len(meta['digests']) >= 5
    for tag, rec in meta['digests'].items():
        torch.manual_seed(meta['seed'])
        sd = TrackMPNN(*rec['args']).state_dict()
        assert list(sd.keys()) == list(rec['sha256'].keys()), tag
        for k, v in sd.items():
            assert hashlib.sha256(v.numpy().tobytes()).hexdigest() == rec['sha256'][k], (tag, k)
    torch.manual_seed(meta['seed'])
    sd = TrackMPNN('2d', 3, 32, 0, 'diff').state_dict()
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), d['full/' + k]), k


def test_any_hidden_width_is_accepted_and_keeps_reference_shapes():
    """nhidden is any int in the reference (utils/training_options.py:22): widths between the instantiated kernel
    widths keep the reference's parameter shapes (state_dict compatibility) and run zero-padded."""
    from trackmpnn_amd import TrackMPNN
    m = TrackMPNN('2d+temp', 3, 48, 1, 'concat')
    assert (m.nhidden, m.hpad, m.spec.H) == (48, 64, 64)
    sd = m.state_dict()
    assert sd['factor_grus.0.edge_gru.weight_ih'].shape == (3 * 48, 2 * 48)
    assert sd['factor_grus.1.gat.0.W_att'].shape == (48, 48) and sd['factor_grus.1.gat.0.a'].shape == (48, 1)
    assert sd['output_transform_edge.weight'].shape == (1, 2 * 48)
    assert sd['input_transforms.0.1.running_var'].shape == (48,)
    named = dict(m.named_parameters())
    padded = m._pad_params([named[k] for k in m.spec.param_names()])
    shapes = {k: tuple(t.shape) for k, t in zip(m.spec.param_names(), padded)}
    assert shapes['factor_grus.0.edge_gru.weight_ih'] == (192, 128) and shapes['factor_grus.0.node_gru.weight_hh'] == (192, 64)
    assert shapes['output_transform_node.weight'] == (1, 128) and shapes['input_transforms.1.3.weight'] == (64, 64)
    w, wp = named['factor_grus.0.edge_gru.weight_ih'], padded[m.spec.param_names().index('factor_grus.0.edge_gru.weight_ih')]
    # gate g, unit i, input block b, column j  ->  row g*64 + i, column b*64 + j ; everything else exactly zero
    assert torch.equal(wp.reshape(3, 64, 2, 64)[:, :48, :, :48], w.reshape(3, 48, 2, 48))
    assert float(wp.abs().sum()) == float(w.abs().sum())
    h = torch.randn(5, 2 * 48)
    assert torch.equal(m._unpad_state(m._pad_state(h)), h)
    assert TrackMPNN('2d', 3, 64, 0, 'diff')._padded is False
    big = TrackMPNN('2d', 3, 300, 0, 'diff')                     # above 256: padded to the next multiple of 128
    assert (big.nhidden, big.hpad, big._padded) == (300, 384, True)
    import pytest
    with pytest.raises(ValueError):
        TrackMPNN('2d', 3, 1100, 0, 'diff')


def _check_tiles(g, tiles, R, dst_offset=0):
    """Every edge row exactly once; each slot's (src, dst) recoverable through the tile's det list; lists ascending."""
    E = g.E
    T = tiles.T
    assert T == (E + R - 1) // R and tiles.rows_per_tile == R
    rows = tiles.t_row.long().view(T, R)
    valid = rows >= 0
    assert int(valid.sum()) == E and bool(valid.view(-1)[:E].all())        # padding only at the very end
    assert torch.equal(torch.sort(rows[valid]).values, g.edge_row.long())
    dptr = tiles.t_dptr.long()
    assert dptr[0] == 0 and dptr[-1] == tiles.t_dets.numel() and bool((dptr[1:] > dptr[:-1]).all())
    epos = torch.full((g.N,), -1, dtype=torch.long)
    epos[g.edge_row.long()] = torch.arange(E)
    loc = tiles.t_loc.long().view(T, R)
    ls, ld = loc & 0xFFFF, loc >> 16
    for t in range(T):
        dets = tiles.t_dets.long()[dptr[t]:dptr[t + 1]]
        assert bool((dets[1:] > dets[:-1]).all())
        assert int(ls[t].max()) < dets.numel() and int(ld[t].max()) < dets.numel()
        v = valid[t]
        e = epos[rows[t][v]]
        assert torch.equal(dets[ls[t][v]], g.src_pos.long()[e]) and torch.equal(dets[ld[t][v]], g.dst_pos.long()[e] + dst_offset)
        assert set(dets.tolist()) == set(g.src_pos.long()[e].tolist()) | set((g.dst_pos.long()[e] + dst_offset).tolist())
    return int((dptr[1:] - dptr[:-1]).max())


def test_edge_tiles_cover_every_edge_and_index_their_dets():
    """struct tmpnn_edge_tiles (include/tmpnn.h): built by build_edge_tiles for the wide cells (128 rows) and the
    H <= 64 cells (32 rows); a dense frame block gives (8 srcs x 16 dsts) tiles over at most 8 + 16 dets away from the
    seams, a ragged batch of small windows still covers every edge once."""
    from trackmpnn_amd import WindowBuilder, batch_windows, synth_window
    from trackmpnn_amd.graph import build_edge_tiles, dense_static_graph, edge_tiles
    g = dense_static_graph(4, 40)                       # 3 blocks of 40 x 40 edges
    tiles = build_edge_tiles(g, 128, stats=True)
    worst = _check_tiles(g, tiles, 128)
    assert worst == tiles.max_dets <= 48
    cnt = (tiles.t_dptr[1:] - tiles.t_dptr[:-1]).long()
    assert int((cnt <= 40).sum()) >= int(0.8 * tiles.T)   # what the kernel stages in LDS (WT_DMAX = 40)
    # consecutive rows would touch up to 128 + 1 dets per tile on a 300-wide block; here a block of 40 still needs 40 + 4
    wins = [WindowBuilder(synth_window(s, 6, 5, 10)).calls() for s in range(5)]
    plans, _ = batch_windows(wins)
    gr = plans[-1].graph
    _check_tiles(gr, build_edge_tiles(gr, 128), 128)
    _check_tiles(gr, build_edge_tiles(gr, 32, 4, 8), 32)
    assert edge_tiles(gr, 128) is edge_tiles(gr, 128)   # cached on the graph
    # the concat message's lists: dst entries offset by Dn (rows [Dn, 2 Dn) of the stacked projected table), cached apart
    _check_tiles(gr, build_edge_tiles(gr, 32, 4, 8, dst_offset=gr.Dn), 32, dst_offset=gr.Dn)
    _check_tiles(g, build_edge_tiles(g, 128, dst_offset=g.Dn), 128, dst_offset=g.Dn)
    assert edge_tiles(gr, 32, dst_offset=gr.Dn) is edge_tiles(gr, 32, dst_offset=gr.Dn) is not edge_tiles(gr, 32)
    empty = dense_static_graph(1, 5)
    assert build_edge_tiles(empty, 128).T == 0


def test_module_copy_and_pickle_on_the_host():
    """copy.deepcopy / pickle of the module (what EMA copies, best-model snapshots and torch.save(model) do): the batch-1
    path's caches (ctypes pointer structs, function pointers, a non-leaf sink tensor) are dropped and rebuilt, parameters
    and buffers are copied, the copy is independent.  (The GPU variant -- after the caches have been populated -- is
    tests/test_small_path_gpu.py::test_model_can_be_copied_and_pickled_after_use.)"""
    import copy
    import pickle
    from trackmpnn_amd import TrackMPNN
    m = TrackMPNN('2d+temp', 3, 64, 1, 'diff')
    m._anch_calls = 5
    m._pending_graphs.append('x')
    for twin in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert twin._small is not m._small and twin._small.model is twin
        assert twin._anch_calls == 0 and twin._pending_graphs == [] and twin._sink is None
        for (k, a), (_, b) in zip(m.state_dict().items(), twin.state_dict().items()):
            assert torch.equal(a, b) and (a.data_ptr() != b.data_ptr() or a.numel() == 0), k
        with torch.no_grad():
            twin.factor_grus[0].edge_gru.weight_hh.mul_(0.0)
        assert float(m.factor_grus[0].edge_gru.weight_hh.abs().sum()) > 0
    m.refresh_weights()                                # drops every cached pointer; nothing to rebuild on the host
    assert m._plist is None and m._sink is None


@pytest.mark.parametrize('T,D', [(4, 45), (3, 64), (5, 37)])
def test_dense_seg_plan_covers_every_incidence_once(T, D):
    """struct tmpnn_seg_plan (trackmpnn_amd.graph.build_seg_plan, consumed by csrc/agg.hip k_segsum_tiles +
    k_segsum_pipe<DUAL>): emulate the two passes with torch ops -- per-tile src sums and per-item dst sums into partial rows,
    then per det the signed sum over its second-pass CSR -- and compare with the plain signed segment sum
    (models/layers.py:103) on integer-valued rows (exact in fp32 in any order)."""
    from trackmpnn_amd.graph import build_seg_plan, dense_static_graph
    g = dense_static_graph(T, D)
    plan = build_seg_plan(g, item_tiles=3, min_fill=0.0)
    assert plan is not None and plan.T > 0 and plan.nsplit == g.N
    H = 8
    x = torch.randint(-8, 9, (g.N, H), generator=torch.Generator().manual_seed(T * 100 + D)).float()
    # reference: es[d] = sum over incident edges of (+x[e] if d is the src, -x[e] if the dst)
    ref = torch.zeros(g.Dn, H)
    ref.index_add_(0, g.src_pos.long(), x[g.edge_row.long()])
    ref.index_add_(0, g.dst_pos.long(), -x[g.edge_row.long()])
    # pass 1: every edge row sits in exactly one slot of one tile
    t_row = plan.t_row.long().view(plan.T, 128)
    live = t_row >= 0
    assert int(live.sum()) == g.E and torch.equal(torch.sort(t_row[live])[0], torch.sort(g.edge_row.long())[0])
    tiles = (x[t_row.clamp(min=0)] * live[..., None]).view(plan.T, 8, 16, H)
    part = torch.zeros(8 * plan.T + 16 * plan.I, H)
    part[:8 * plan.T] = tiles.sum(2).reshape(-1, H)
    for k, (t0, nt) in enumerate(plan.items.tolist()):
        part[8 * plan.T + 16 * k:8 * plan.T + 16 * k + 16] = tiles[t0:t0 + nt].sum((0, 1))
    assert int(plan.items[:, 1].sum()) == plan.T and int(plan.items[:, 1].max()) <= 3
    # pass 2
    inc2 = plan.inc2.long()
    neg = inc2 < 0
    idx = torch.where(neg, inc2 + 2 ** 31, inc2)
    assert bool((idx >= plan.nsplit).all())
    rows = part[idx - plan.nsplit]
    rows = torch.where(neg[:, None], -rows, rows)
    det = torch.repeat_interleave(torch.arange(g.Dn), torch.diff(plan.rowptr2.long()))
    out = torch.zeros(g.Dn, H)
    out.index_add_(0, det, rows)
    assert torch.equal(out, ref)
    # a graph with an edge listed twice has no plan (a slot holds one row)
    g2 = dense_static_graph(2, 20)
    g2.src_pos[1], g2.dst_pos[1] = g2.src_pos[0], g2.dst_pos[0]
    assert build_seg_plan(g2, min_fill=0.0) is None


@pytest.mark.gpu
def test_dense_segsum_reads_rows_once_and_matches_the_csr_kernel():
    """tmpnn_segsum_fwd with a tmpnn_seg_plan on the graph (dense form: k_segsum_tiles + k_segsum_pipe<DUAL>) against the same
    entry point without one (k_segsum_pipe over the graph's CSR): exact on integer-valued rows, 1e-5 relative on random rows
    (another summation order), bitwise repeatable; compact and scattered output, accumulate on and off, a column block of a
    wider row (the wide backward's d_gi planes)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd.graph import dense_static_graph
    dev = torch.device('cuda:0')
    _dense_segsum_case(dense_static_graph(5, 150, device=dev), dense_static_graph(5, 150, device=dev))
    # the same scene after decode_tracks-like damage: a random 15 % of the edge rows gone (holes inside the tiles)
    from trackmpnn_amd.graph import graph_from_edges
    full = dense_static_graph(4, 120)
    gen = torch.Generator().manual_seed(17)
    drop = torch.rand(full.E, generator=gen) < 0.15
    keep_row = torch.ones(full.N, dtype=torch.bool)
    keep_row[full.edge_row.long()[drop]] = False
    new_index = torch.cumsum(keep_row.long(), 0) - 1
    is_edge = full.is_edge.bool()[keep_row]
    src = new_index[full.src.long()[~drop]]
    dst = new_index[full.dst.long()[~drop]]
    ragged = [graph_from_edges(int(keep_row.sum()), is_edge, src, dst, device=dev) for _ in range(2)]
    _dense_segsum_case(*ragged)


def _dense_segsum_case(g, g0):
    from trackmpnn_amd import _lib
    from trackmpnn_amd.graph import dense_seg_plan
    dev = g.device
    plan = dense_seg_plan(g)
    assert plan is not None and plan.T > 0 and 128 * plan.T > g.E        # (border tiles / damaged tiles have empty slots)
    H, LD = 256, 1024
    st = _lib.raw_stream()
    gen = torch.Generator().manual_seed(3)
    for kind in ('int', 'rand'):
        x = (torch.randint(-8, 9, (g.N, LD), generator=gen).float() if kind == 'int'
             else torch.randn(g.N, LD, generator=gen)).to(dev)
        for c0, compact, acc in ((0, 1, 0), (256, 1, 0), (512, 0, 1), (0, 0, 0)):
            rows = g.Dn if compact else g.N
            base = torch.randn(rows, H, generator=gen).to(dev) if acc else torch.zeros(rows, H, device=dev)
            outs = []
            for graph in (g, g0, g):
                o = base.clone()
                _lib.call('tmpnn_segsum_fwd', graph.cref(), x.data_ptr() + 4 * c0, LD, o.data_ptr(), H, H, acc, compact, st)
                outs.append(o)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[2]), (kind, c0, compact, acc)
            if kind == 'int' and not acc:
                assert torch.equal(outs[0], outs[1]), (kind, c0, compact, acc)
            else:
                scale = outs[1].abs().max().item()
                assert (outs[0] - outs[1]).abs().max().item() <= 1e-5 * scale, (kind, c0, compact, acc)


def _window_batch_graph(B, device):
    """The last call's graph of B block-diagonal rolling windows (12 frames, ~8 dets a frame): ~1.4 k edges a window."""
    from trackmpnn_amd import WindowBuilder, batch_windows, synth_window
    distinct = min(B, 24)
    wins = [WindowBuilder(synth_window(7000 + s, 12, 8.0, 25)).calls() for s in range(distinct)]
    wins = (wins * ((B + distinct - 1) // distinct))[:B]
    plans, _ = batch_windows(wins, device='cpu')
    return plans[-1].graph.to(device)


def _emulate_window_kernel(g, wp, x):
    """k_segsum_win (csrc/agg.hip) restated over the plan's records: per window and chunk, step by step, every lane group adds
    the row its record names to the stream the record names; a det's four streams are combined as (s0 + s1) + (s2 + s3)."""
    from trackmpnn_amd.graph import WIN_CH, WIN_NHG, WIN_CAP_DETS
    wrec, erow, recs, det, _, _ = wp.t
    recs = recs.long() & 0xffff
    out = torch.full((g.Dn, x.shape[1]), float('nan'), dtype=x.dtype)
    used = 0
    for w in range(wp.W):
        e0, ne, q0, nd, st = (int(v) for v in wrec[w, :5])
        lm = [int(v) & 0xffffffff for v in wrec[w, 5:8]]
        if nd > WIN_CAP_DETS:
            continue
        acc = torch.zeros(WIN_CAP_DETS * 4, x.shape[1], dtype=x.dtype)
        for c in range((ne + WIN_CH - 1) // WIN_CH):
            L = (lm[c // 8] >> (4 * (c % 8))) & 15
            for s_ in range(L):
                for hg in range(WIN_NHG):
                    r = int(recs[(st + s_) * WIN_NHG + hg])
                    if r & 0x8000:
                        continue
                    used += 1
                    row = x[int(erow[e0 + c * WIN_CH + (r & 255)])]
                    acc[((r >> 9) & 7) * WIN_NHG + hg] += -row if r & 0x100 else row
            st += L
        for q in range(nd):
            a = acc[4 * q:4 * q + 4]
            out[int(det[q0 + q])] = (a[0] + a[1]) + (a[2] + a[3])
    return out, used


def test_window_plan_adds_every_incidence_once_in_the_csr_kernels_order():
    """build_win_plan on the CPU: the kernel's algorithm restated over the plan gives k_segsum_pipe's sums BIT FOR BIT (fp32, its
    lane-group order: positions i, i + 4, ... of a run per group, then (g0 + g1) + (g2 + g3)) and uses every incidence once;
    windows beyond the kernel's capacity are listed for the CSR kernel; a crossing edge declines the plan."""
    from trackmpnn_amd.graph import build_win_plan, WIN_CAP_DETS
    g = _window_batch_graph(12, 'cpu')
    dg = g.__dict__['_det_group']
    x = torch.randn(g.N, 3, generator=torch.Generator().manual_seed(2))
    rp, inc = g.rowptr.long(), g.inc.long()
    ref = torch.zeros(g.Dn, 3)
    for d in range(g.Dn):
        a = torch.zeros(4, 3)
        for i, p in enumerate(range(int(rp[d]), int(rp[d + 1]))):
            v = int(inc[p])
            a[i & 3] += -x[v & 0x7fffffff] if v < 0 else x[v & 0x7fffffff]
        ref[d] = (a[0] + a[1]) + (a[2] + a[3])
    wp = build_win_plan(g, dg)
    assert wp is not None and wp.W == 12 and wp.nbig == 0
    out, used = _emulate_window_kernel(g, wp, x)
    assert used == 2 * g.E and torch.equal(out, ref)
    # four windows under one label: more dets than the kernel keeps streams for -> that window goes to the CSR kernel
    merged = torch.where(dg < 4, torch.zeros_like(dg), dg)
    wp2 = build_win_plan(g, merged)
    assert wp2 is not None and wp2.W == 9 and wp2.nbig > WIN_CAP_DETS
    assert sorted(wp2.t[5].tolist()) == torch.nonzero(merged == 0).flatten().tolist()
    out2, _ = _emulate_window_kernel(g, wp2, x)
    served = ~torch.isnan(out2[:, 0])
    assert torch.equal(served, merged != 0) and torch.equal(out2[served], ref[served])
    # labels that cut through a window: an edge would cross -> no plan
    assert build_win_plan(g, torch.arange(g.Dn) % 2) is None


@pytest.mark.gpu
def test_window_segsum_reads_rows_once_and_equals_the_csr_kernel_bitwise():
    """tmpnn_segsum_fwd with a tmpnn_win_plan on the graph (k_segsum_win: chunks of a window's edge rows staged in LDS once, the
    partial sums of a det's four streams in LDS) against the same entry point without one (k_segsum_pipe): BITWISE equal -- a
    stream is k_segsum_pipe's lane group and is added to in the same order.  Compact and scattered output, accumulate on and off, a column
    block of a wider row, a window beyond the kernel's capacity (served by the CSR kernel through the plan's list)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import copy
    from trackmpnn_amd import _lib
    from trackmpnn_amd.graph import build_win_plan, win_plan
    if not hasattr(_lib.load(), 'tmpnn_wide_gru_bwd_data'):
        pytest.skip('k_segsum_win compiles only with -DTMPNN_KEEP_VARIANTS (tools/build_variants_all.sh; load it with '
                    'TMPNN_LIB_PATH): the shipped library ignores tmpnn_graph.win_plan')
    dev = torch.device('cuda:0')
    g = _window_batch_graph(200, dev)
    dg = g.__dict__['_det_group']
    g0 = copy.copy(g)                                   # the same graph without a plan
    g0.__dict__.pop('_win_plan', None)
    g0.__dict__.pop('_det_group', None)
    g0._c = None
    assert win_plan(g) is not None and win_plan(g).nbig == 0 and g.cstruct().win_plan and not g0.cstruct().win_plan
    g2 = copy.copy(g)                                   # windows 0..3 as one: beyond the capacity
    g2._c = None
    g2.__dict__['_win_plan'] = build_win_plan(g, torch.where(dg < 4, torch.zeros_like(dg), dg))
    assert g2.__dict__['_win_plan'].nbig > 0
    H, LD = 64, 192
    st = _lib.raw_stream()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(g.N, LD, generator=gen).to(dev)
    for c0, compact, acc in ((0, 1, 0), (64, 1, 1), (128, 0, 1), (0, 0, 0)):
        rows = g.Dn if compact else g.N
        base = torch.randn(rows, H, generator=gen).to(dev)
        outs = []
        for graph in (g0, g, g2, g):
            o = base.clone()
            _lib.call('tmpnn_segsum_fwd', graph.cref(), x.data_ptr() + 4 * c0, LD, o.data_ptr(), H, H, acc, compact, st)
            outs.append(o)
        torch.cuda.synchronize()
        if not compact:                                 # (edge rows of a scattered output are not written)
            keep = g.det_row.long()
            outs = [o[keep] for o in outs]
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (c0, compact, acc)


def test_train_mode_active_set_rule_on_the_host_equals_the_replayed_reference_rule():
    """TrackGraph.update(mode='train') sizes its append from the labels (tracking.TrackGraph._train_active_count: the dets of the
    previous non-empty timestep + every true-positive det whose track has no later detection yet) instead of reading the
    device's count back.  WindowBuilder replays the reference's rule literally (utils/graph.py:228-245, 271-274: associations
    through existing edges); on chunks of every shape -- false positives, missed detections, pauses, empty timesteps -- the
    two must give the same active-set size at every timestep."""
    from trackmpnn_amd import WindowBuilder, synth_window
    from trackmpnn_amd.tracking import TrackGraph

    class Host:                                         # the two host methods on a bare state (no device)
        _train_state_add = TrackGraph._train_state_add
        _train_active_count = TrackGraph._train_active_count

    checked = 0
    for seed in range(80):
        yy = synth_window(900 + seed, 5 + seed % 8, 3.0 + seed % 7, 30, survival=0.6 + 0.05 * (seed % 8), fp_rate=0.05 * (seed % 6),
                          dropout=0.1 * (seed % 5))
        if seed % 4 == 0:
            yy = yy[yy[:, 0] != 2]                      # an empty timestep
        times = np.unique(yy[:, 0])
        calls = WindowBuilder(yy).calls()
        if len(calls) < 2:
            continue
        h = Host()
        h._tr = dict(last={}, dup=False, t_prev=int(times[0]), n_prev=0)
        for tt in times[:2]:
            h._train_state_add(int(tt), yy[yy[:, 0] == tt, 1].tolist())
        for c, tt in zip(calls[1:], times[2:]):
            nt = int((yy[:, 0] == tt).sum())
            A, dup = h._train_active_count()
            assert not dup and A * nt + nt == c.n_new, (seed, int(tt), A, c.n_new)
            h._train_state_add(int(tt), yy[yy[:, 0] == tt, 1].tolist())
            checked += 1
    assert checked > 300
