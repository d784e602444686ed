"""Targets and losses: oracle vs the reference-generated fixtures on CPU, HIP vs fixtures on GPU."""
import numpy as np
import pytest
import torch

from oracle import trackmpnn_oracle as orc
from tests.conftest import loss_golden_names
from tests.golden_util import Golden


def _call(gold, c):
    na = gold.adjacency(c, 'node_adj')
    return (na, gold.t(f'c{c}/labels'), gold.t(f'c{c}/logits'), gold.t(f'c{c}/targets'), gold.t(f'c{c}/grad_logits'),
            float(gold.d[f'c{c}/loss_c']), float(gold.d[f'c{c}/loss_f']), float(gold.d[f'c{c}/loss_g']))


@pytest.mark.parametrize('name', loss_golden_names())
def test_oracle_losses_match_reference(name):
    gold = Golden(name)
    for c in range(gold.ncalls):
        na, labels, logits, targets, grad, lc, lf, lg = _call(gold, c)
        g = orc.graph_from_adjacency(na)
        t = orc.create_targets(labels, g)
        assert torch.equal(t, targets), c
        lo = logits.clone().requires_grad_(True)
        sc = torch.sigmoid(lo)
        dr, er = torch.from_numpy(g.det_row), torch.from_numpy(g.edge_row)
        loss_c = orc.ce_loss(lo, t, g)
        loss_f = orc.focal_loss(sc[dr, 0], t[dr]) + orc.focal_loss(sc[er, 0], t[er])
        loss_g = orc.focal_loss(sc[er, 0], t[er], gamma=2, alpha=[0.75, 0.25], size_average=False)
        assert abs(loss_c.item() - lc) <= 1e-5 * max(1.0, abs(lc))
        assert abs(loss_f.item() - lf) <= 1e-5 * max(1.0, abs(lf))
        assert abs(loss_g.item() - lg) <= 1e-5 * max(1.0, abs(lg))
        (loss_c + loss_f + 0.5 * loss_g).backward()
        assert torch.allclose(lo.grad, grad, atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('name', loss_golden_names())
def test_hip_losses_match_reference(name):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd.loss import CELoss, FocalLoss, create_targets
    from trackmpnn_amd import graph_from_adjacency
    dev = 'cuda:0'
    gold = Golden(name)
    for c in range(gold.ncalls):
        na, labels, logits, targets, grad, lc, lf, lg = _call(gold, c)
        na_d = na.to(dev)
        t = create_targets(labels.to(dev), na_d, None)                      # reference signature: adjacency tensor
        assert t.dtype == labels.dtype and torch.equal(t.cpu(), targets), c
        g = graph_from_adjacency(na_d)
        lo = logits.to(dev).requires_grad_(True)
        sc = torch.sigmoid(lo)
        dr, er = g.det_row.long(), g.edge_row.long()
        loss_c = CELoss()(lo, t, g, None)                                   # ... or a prebuilt FrameGraph
        loss_f = FocalLoss(gamma=0)(sc[dr, 0], t[dr]) + FocalLoss(gamma=0)(sc[er, 0], t[er])
        loss_g = FocalLoss(gamma=2, alpha=0.25, size_average=False)(sc[er, 0], t[er])
        assert abs(loss_c.item() - lc) <= 2e-5 * max(1.0, abs(lc))
        assert abs(loss_f.item() - lf) <= 2e-5 * max(1.0, abs(lf))
        assert abs(loss_g.item() - lg) <= 2e-5 * max(1.0, abs(lg))
        (loss_c + loss_f + 0.5 * loss_g).backward()
        assert torch.allclose(lo.grad.cpu(), grad, atol=2e-5, rtol=2e-4)


@pytest.mark.gpu
def test_hip_losses_at_scale_vs_torch():
    """A 2048-window batch: CE loss against an independent torch formulation (scatter logsumexp per set)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd import WindowBuilder, batch_windows, synth_window
    from trackmpnn_amd.loss import CELoss, create_targets
    dev = 'cuda:0'
    wins = [WindowBuilder(synth_window(s, 7, 6, 20)).calls() for s in range(64)]
    plans, _ = batch_windows(wins * 32, static=True)
    g = plans[-1].graph.to(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    labels = (torch.rand(g.N, device=dev, generator=gen) < 0.15).long()
    logits = torch.randn(g.N, 1, device=dev, generator=gen, requires_grad=True)
    t = create_targets(labels, g)
    loss = CELoss()(logits, t, g)
    loss.backward()
    # independent formulation: set id = 2*det + (0 past | 1 future) per CSR entry
    counts = (g.rowptr[1:] - g.rowptr[:-1]).long()
    det_of = torch.repeat_interleave(torch.arange(g.Dn, device=dev), counts)
    row = (g.inc & 0x7FFFFFFF).long()
    sid = 2 * det_of + (g.inc >= 0).long()
    lo = logits.detach()[row, 0].double().requires_grad_(True)
    S = 2 * g.Dn
    mx = torch.full((S,), -1e30, device=dev, dtype=torch.float64).scatter_reduce(0, sid, lo.detach(), 'amax')
    z = torch.zeros(S, device=dev, dtype=torch.float64).index_add(0, sid, torch.exp(lo - mx[sid]))
    n = torch.zeros(S, device=dev, dtype=torch.float64).index_add(0, sid, torch.ones_like(lo))
    tpos = t[row] != 0
    # target entry per set: last positive (past, sid even) / first positive (future, sid odd)
    pos_idx = torch.arange(row.numel(), device=dev)
    last = torch.full((S,), -1, device=dev, dtype=torch.long).scatter_reduce(0, sid[tpos], pos_idx[tpos], 'amax')
    first = torch.full((S,), 2 ** 62, device=dev, dtype=torch.long).scatter_reduce(0, sid[tpos], pos_idx[tpos], 'amin')
    is_past = (torch.arange(S, device=dev) % 2) == 0
    tgt = torch.where(is_past, last, torch.where(first == 2 ** 62, torch.full_like(first, -1), first))
    has = tgt >= 0
    ref = ((torch.log(z[has]) + mx[has] - lo[tgt[has]]) / n[has]).sum()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    ref.backward()
    gref = torch.zeros(g.N, device=dev, dtype=torch.float64).index_add(0, row, lo.grad)
    assert torch.allclose(logits.grad[:, 0].double(), gref, atol=1e-6, rtol=1e-4)


@pytest.mark.gpu
def test_focal_loss_empty_selection_matches_reference_semantics():
    """models/loss.py:71-74 on an empty index set: mean() of nothing is nan, sum() of nothing is 0; zero gradient."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd.loss import FocalLoss
    s = torch.zeros(0, device='cuda:0', requires_grad=True)
    t = torch.zeros(0, device='cuda:0')
    assert torch.isnan(FocalLoss(gamma=0)(s, t))
    out = FocalLoss(gamma=2, alpha=0.25, size_average=False)(s, t)
    assert out.item() == 0.0
    out.backward()
    assert s.grad.shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize('gamma,alpha,mean', [(0, None, True), (2, 0.25, False)])
def test_focal_loss_over_a_row_list_equals_the_gathered_form(gamma, alpha, mean):
    """FocalLoss(scores_full, targets_full, rows=idx) == FocalLoss(scores_full[idx], targets_full[idx]) -- train.py:76-81's
    selections without the gathers: same value bit for bit, same gradient on the full score vector (zero outside idx);
    byte targets (create_targets(as_bytes=True)) are taken as they are; an empty list behaves like an empty selection."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd.loss import FocalLoss
    gen = torch.Generator().manual_seed(7)
    N = 5000
    s0 = torch.rand(N, 1, generator=gen).to('cuda:0')
    t = (torch.rand(N, generator=gen) < 0.3).to('cuda:0')
    idx = torch.randperm(N, generator=gen)[:1777].sort().values.to('cuda:0')
    loss = FocalLoss(gamma=gamma, alpha=alpha, size_average=mean)
    a = s0.clone().requires_grad_(True)
    la = loss(a[idx, 0], t[idx].long())
    la.backward()
    b = s0.clone().requires_grad_(True)
    lb = loss(b[:, 0], t.to(torch.uint8), rows=idx.to(torch.int32))
    lb.backward()
    assert la.item() == lb.item()
    assert torch.equal(a.grad, b.grad)
    outside = torch.ones(N, dtype=torch.bool, device='cuda:0')
    outside[idx] = False
    assert float(b.grad[outside].abs().max()) == 0.0
    empty = loss(b[:, 0], t.to(torch.uint8), rows=idx[:0].to(torch.int32))
    assert torch.isnan(empty) if mean else empty.item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('tp', [True, False])
def test_train_losses_equals_the_separate_modules(tp):
    """trackmpnn_amd.loss.train_losses (one autograd node for create_targets + CELoss + the focal terms, as
    trackmpnn_amd.loops uses it) against the reference-shaped sequence of calls (train.py:70-81): same two losses and the
    same gradients on scores and logits, bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from trackmpnn_amd.graph import synth_window
    from trackmpnn_amd.loss import CELoss, FocalLoss, create_targets, train_losses
    from trackmpnn_amd.tracking import TrackGraph
    dev = 'cuda:0'
    yy = synth_window(777, 6, 6.0, 15)
    y = torch.from_numpy(yy)[None]
    X = torch.randn(1, yy.shape[0], 8, generator=torch.Generator().manual_seed(3))
    tg, feats, t_st, t_end = TrackGraph.initialize(X, y, 0, 'train', dev)
    for t_cur in range(t_st, min(t_st + 2, t_end)):
        tg.update(None, X, y, t_cur, mode='train')
    g = tg.graph.frame_graph()
    gen = torch.Generator().manual_seed(11)
    s0 = torch.rand(g.N, 1, generator=gen).to(dev)
    l0 = torch.randn(g.N, 1, generator=gen).to(dev)
    labels = tg.labels()
    # the separate modules, as train.py calls them
    sa, la = s0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
    targets = create_targets(labels, g)
    loss_c = CELoss()(la, targets, g)
    idx_e, idx_n = g.edge_row.long(), g.det_row.long()
    loss_f = FocalLoss(gamma=0, alpha=None)(sa[idx_e, 0], targets[idx_e])
    if tp:
        loss_f = FocalLoss(gamma=0, alpha=None)(sa[idx_n, 0], targets[idx_n]) + loss_f
    (loss_c * 0.7 + loss_f * 1.3).backward()
    # the composed node
    sb, lb = s0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
    c2, f2 = train_losses(sb, lb, tg.labels_u8(), g, tp)
    (c2 * 0.7 + f2 * 1.3).backward()
    assert c2.item() == loss_c.item() and f2.item() == loss_f.item()
    assert torch.equal(la.grad, lb.grad)
    assert torch.equal(sa.grad, sb.grad)
    # handed the DeviceGraph itself (as trackmpnn_amd.loops does): the native autograd node over the same two launches
    sc, lc = s0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
    c3, f3 = train_losses(sc, lc, tg.labels_u8(), tg.graph, tp)
    (c3 * 0.7 + f3 * 1.3).backward()
    assert c3.item() == loss_c.item() and f3.item() == loss_f.item()
    assert torch.equal(la.grad, lc.grad) and torch.equal(sa.grad, sc.grad)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [0, 1, 5, 4095, 4096, 4097, 1_000_003])
def test_bce_with_logits_sum_equals_torch(n):
    """trackmpnn_amd.loss.bce_with_logits_sum (the loss bench.py's step uses; SURVEY 8(d): BCE on all logits vs fixed {0,1}
    targets) against torch.nn.functional.binary_cross_entropy_with_logits(reduction='sum') evaluated in fp64: value within
    1e-6 relative, gradient within 1e-6 absolute incl. an upstream factor; logits span +-40 (saturated sigmoids); bitwise
    repeatable."""
    import torch.nn.functional as Fnn
    from trackmpnn_amd.loss import bce_with_logits_sum
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    gen = torch.Generator().manual_seed(n)
    l = (torch.randn(n, 1, generator=gen) * 8.0).clamp(-40, 40)
    t = (torch.rand(n, 1, generator=gen) < 0.3).float()
    ld = l.to('cuda:0').requires_grad_(True)
    loss = bce_with_logits_sum(ld, t.to('cuda:0'))
    (loss * 1.7).backward()
    l64 = l.double().requires_grad_(True)
    ref = Fnn.binary_cross_entropy_with_logits(l64, t.double(), reduction='sum')
    (ref * 1.7).backward()
    assert loss.shape == () and abs(loss.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    if n:
        assert ld.grad.shape == ld.shape
        assert (ld.grad.cpu().double() - l64.grad).abs().max().item() <= 1e-6
    loss2 = bce_with_logits_sum(ld.detach(), t.to('cuda:0'))
    assert torch.equal(loss2, loss.detach())
