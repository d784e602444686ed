"""CPU oracle for the TrackMPNN message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``trackmpnn_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU baseline -- never as the product path.

It is a *restatement* (index based, O(E*H^2)) of what the reference computes with dense
N x N algebra, written with stock torch-CPU fp32 ops so autograd supplies the adjoint:

  reference/models/track_mpnn.py:54-75   TrackMPNN.forward          -> ``forward``
  reference/models/track_mpnn.py:45-52   input transform (Lin-BN-ReLU-Lin) -> ``_input_transform``
  reference/models/layers.py:84-116      FactorGraphGRU.forward     -> ``_factor_gru``
  reference/models/layers.py:26-43       GraphAttentionLayer.forward-> ``_attention``
  torch.nn.GRUCell (gate order r,z,n)                               -> ``_gru_cell``
  reference/utils/graph.py:151-163,294-308  adjacency invariants    -> ``graph_from_adjacency``

Parity pin: ``tests/golden/*.npz`` were produced by ``oracle/gen_golden.py`` which imports
the real reference in the build container and dumps its inputs/outputs/gradients;
``tests/test_oracle_golden.py`` checks this file against every one of them.

Extensions over the reference (needed to batch many tracking windows block-diagonally
without changing any per-window result):
  * ``seg_ids``: BatchNorm statistics are taken per *segment* (= per window) of the new rows;
    a single segment reproduces the reference exactly.
  * ``dropout_keep``: the attention dropout mask is an explicit input ([K, E, 2] per group),
    because the reference's dense N x N bernoulli stream cannot be reproduced by a sparse
    implementation.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.2
ATT_DROPOUT_P = 0.5


# ----------------------------------------------------------------------------------------
# graph
# ----------------------------------------------------------------------------------------
@dataclass
class OracleGraph:
    """Index form of the bipartite det/edge factor graph (reference/utils/graph.py:151-163)."""
    N: int
    is_edge: np.ndarray   # bool [N]
    src: np.ndarray       # int64 [E]  det row with +1 in node_adj[e, :]
    dst: np.ndarray       # int64 [E]  det row with -1 in node_adj[e, :]
    edge_row: np.ndarray  # int64 [E]  row index of edge e (ascending)
    det_row: np.ndarray   # int64 [Dn]

    @property
    def E(self) -> int:
        return int(self.edge_row.shape[0])

    @property
    def Dn(self) -> int:
        return int(self.det_row.shape[0])


def _coo(adj: torch.Tensor) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(rows, cols, vals) with duplicates summed and explicit zeros dropped."""
    adj = adj.detach().cpu()
    if adj.is_sparse:
        adj = adj.coalesce()
        idx = adj.indices().numpy()
        val = adj.values().numpy()
        keep = val != 0
        return idx[0][keep], idx[1][keep], val[keep]
    nz = torch.nonzero(adj)
    return nz[:, 0].numpy(), nz[:, 1].numpy(), adj[nz[:, 0], nz[:, 1]].numpy()


def graph_from_adjacency(node_adj: torch.Tensor, edge_adj: Optional[torch.Tensor] = None) -> OracleGraph:
    """Derive (src, dst, type mask) from the reference adjacency pair and check its invariants.

    node_adj: diag = 1 on det rows; an edge row e holds +1 at its earlier det and -1 at its
    later det (reference/utils/graph.py:153-156,298-301).  edge_adj = node_adj^T off the
    diagonal + 1 on edge-row diagonals (reference/utils/graph.py:158-163,303-308).
    """
    N = int(node_adj.shape[0])
    r, c, v = _coo(node_adj)
    diag = r == c
    is_det = np.zeros(N, dtype=bool)
    is_det[r[diag]] = v[diag] != 0
    is_edge = ~is_det
    ro, co, vo = r[~diag], c[~diag], v[~diag]
    pos = vo > 0
    neg = vo < 0
    edge_row = np.nonzero(is_edge)[0].astype(np.int64)
    src = np.full(N, -1, dtype=np.int64)
    dst = np.full(N, -1, dtype=np.int64)
    if np.unique(ro[pos]).size != pos.sum() or np.unique(ro[neg]).size != neg.sum():
        raise ValueError("edge row with more than one +1 / -1 entry")
    src[ro[pos]] = co[pos]
    dst[ro[neg]] = co[neg]
    if is_det[ro].any():
        raise ValueError("det row with off-diagonal entries in node_adj")
    if (src[edge_row] < 0).any() or (dst[edge_row] < 0).any():
        raise ValueError("edge row without exactly one +1 and one -1")
    if not (np.abs(vo) == 1).all():
        raise ValueError("node_adj off-diagonals must be +-1")
    if edge_adj is not None:
        r2, c2, v2 = _coo(edge_adj)
        d2 = r2 == c2
        ie = np.zeros(N, dtype=bool)
        ie[r2[d2]] = v2[d2] != 0
        if not (ie == is_edge).all():
            raise ValueError("diag(edge_adj) does not complement diag(node_adj)")
        a = np.lexsort((r2[~d2], c2[~d2]))
        b = np.lexsort((co, ro))
        if not (np.array_equal(c2[~d2][a], ro[b]) and np.array_equal(r2[~d2][a], co[b])
                and np.array_equal(v2[~d2][a], vo[b])):
            raise ValueError("edge_adj is not node_adj^T off the diagonal")
    return OracleGraph(N=N, is_edge=is_edge, src=src[edge_row], dst=dst[edge_row],
                       edge_row=edge_row, det_row=np.nonzero(is_det)[0].astype(np.int64))


# ----------------------------------------------------------------------------------------
# model description
# ----------------------------------------------------------------------------------------
def feature_groups(features: str, ncategories: int) -> List[Tuple[str, int]]:
    """reference/models/track_mpnn.py:17-33 (order 2d, temp, vis)."""
    out = []
    if '2d' in features:
        out.append(('2d', ncategories + 5))
    if 'temp' in features:
        out.append(('temp', 2))
    if 'vis' in features:
        out.append(('vis', 128))
    return out


@dataclass
class OracleConfig:
    features: str
    ncategories: int
    nhidden: int
    nattheads: int
    msg_type: str

    @property
    def groups(self):
        return feature_groups(self.features, self.ncategories)


def _gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1.0 - z) * n + z * h


def _input_transform(p: Dict[str, torch.Tensor], g: int, xg: torch.Tensor, seg: torch.Tensor,
                     nseg: int, training: bool, update_running: bool) -> torch.Tensor:
    """Lin1 -> BatchNorm1d (stats per segment over ALL new rows, zero rows included) -> ReLU -> Lin2."""
    pre = f'input_transforms.{g}.'
    y = F.linear(xg, p[pre + '0.weight'], p[pre + '0.bias'])
    if training:
        cnt = torch.bincount(seg, minlength=nseg).to(y.dtype)
        if (cnt == 1).any():
            # torch.nn.functional.batch_norm raises for a single row in training mode
            raise ValueError("Expected more than 1 value per channel when training")
        H = y.shape[1]
        s1 = torch.zeros(nseg, H, dtype=y.dtype).index_add(0, seg, y)
        mean = s1 / cnt[:, None]
        d = y - mean[seg]
        var = torch.zeros(nseg, H, dtype=y.dtype).index_add(0, seg, d * d) / cnt[:, None]
        yhat = d / torch.sqrt(var[seg] + BN_EPS)
        if update_running:
            with torch.no_grad():
                rm, rv = p[pre + '1.running_mean'], p[pre + '1.running_var']
                for s in range(nseg):   # windows are seen one after another in the reference
                    unb = var[s] * (cnt[s] / (cnt[s] - 1.0))
                    rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean[s])
                    rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb)
                p[pre + '1.num_batches_tracked'] += nseg
    else:
        yhat = (y - p[pre + '1.running_mean']) / torch.sqrt(p[pre + '1.running_var'] + BN_EPS)
    a = torch.relu(yhat * p[pre + '1.weight'] + p[pre + '1.bias'])
    return F.linear(a, p[pre + '3.weight'], p[pre + '3.bias'])


def _attention(p, pre, k, h, graph: OracleGraph, keep: Optional[torch.Tensor]):
    """One GAT head (reference/models/layers.py:26-43) on the incidence list.

    Returns (edge_support [N,H], alpha [E,2]) with alpha[:,0] the weight the src det gives the
    edge and alpha[:,1] the weight the dst det gives it (after dropout, as the reference returns).
    """
    N, E = graph.N, graph.E
    src = torch.from_numpy(graph.src)
    dst = torch.from_numpy(graph.dst)
    er = torch.from_numpy(graph.edge_row)
    W = p[f'{pre}gat.{k}.W_att']
    a = p[f'{pre}gat.{k}.a']
    ha = h @ W
    s = F.leaky_relu(torch.abs(ha[src] - ha[dst]) @ a, LEAKY_SLOPE).squeeze(1)      # [E]
    det = torch.cat([src, dst])                                                    # incidence -> det
    sc = torch.cat([s, s])
    mx = torch.full((N,), -float('inf'), dtype=h.dtype).scatter_reduce(0, det, sc.detach(), 'amax')
    ex = torch.exp(sc - mx[det])
    den = torch.zeros(N, dtype=h.dtype).index_add(0, det, ex)
    alpha = ex / den[det]
    if keep is not None:
        alpha = alpha * torch.cat([keep[:, 0], keep[:, 1]]).to(h.dtype) / (1.0 - ATT_DROPOUT_P)
    sign = torch.cat([torch.ones(E, dtype=h.dtype), -torch.ones(E, dtype=h.dtype)])
    vals = h[torch.cat([er, er])] * (alpha * sign)[:, None]
    es = torch.zeros_like(h).index_add(0, det, vals)
    return es, torch.stack([alpha[:E], alpha[E:]], dim=1)


def _factor_gru(p, cfg: OracleConfig, g: int, h: torch.Tensor, graph: OracleGraph,
                keep: Optional[torch.Tensor]):
    pre = f'factor_grus.{g}.'
    src = torch.from_numpy(graph.src)
    dst = torch.from_numpy(graph.dst)
    er = torch.from_numpy(graph.edge_row)
    dr = torch.from_numpy(graph.det_row)
    # node -> edge message (layers.py:90-95)
    if cfg.msg_type == 'concat':
        ns = torch.cat([h[src], h[dst]], dim=1)
    else:
        ns = h[src] - h[dst]
    edge_out = _gru_cell(ns, h[er], p[pre + 'edge_gru.weight_ih'], p[pre + 'edge_gru.weight_hh'],
                         p[pre + 'edge_gru.bias_ih'], p[pre + 'edge_gru.bias_hh'])
    # edge -> node aggregation (layers.py:99-112)
    att = None
    if cfg.nattheads <= 0:
        es = torch.zeros_like(h).index_add(0, src, h[er]).index_add(0, dst, -h[er])
    else:
        att = []
        es = 0
        for k in range(cfg.nattheads):
            e_k, a_k = _attention(p, pre, k, h, graph, None if keep is None else keep[k])
            es = es + e_k
            att.append(a_k)
        es = es / cfg.nattheads
    node_out = _gru_cell(es[dr], h[dr], p[pre + 'node_gru.weight_ih'], p[pre + 'node_gru.weight_hh'],
                         p[pre + 'node_gru.bias_ih'], p[pre + 'node_gru.bias_hh'])
    # type-masked merge (layers.py:116)
    out = torch.zeros_like(h).index_copy(0, er, edge_out).index_copy(0, dr, node_out)
    return out, att


def forward(p: Dict[str, torch.Tensor], cfg: OracleConfig, x: torch.Tensor, h_in: Optional[torch.Tensor],
            graph: OracleGraph, training: bool = True, seg_ids: Optional[torch.Tensor] = None,
            dropout_keep: Optional[Sequence[torch.Tensor]] = None, update_running: bool = True):
    """One TrackMPNN.forward call (reference/models/track_mpnn.py:54-75).

    x [n, sum F_g] features of the NEW rows (edge rows all-zero), h_in None | [N-n, G*H].
    seg_ids int64 [n] window id of each new row (None = one window).
    dropout_keep: per group a [K, E, 2] {0,1} tensor (only used when training and K > 0).
    Returns scores [N,1], logits [N,1], h_out [N, G*H], attention (tuple over groups of
    None | list over heads of alpha [E,2]).
    """
    H = cfg.nhidden
    groups = cfg.groups
    N = graph.N
    n = int(x.shape[0])
    is_det = torch.from_numpy(~graph.is_edge)
    if n > 0:
        if seg_ids is None:
            seg_ids = torch.zeros(n, dtype=torch.int64)
        nseg = int(seg_ids.max().item()) + 1
        new_det = is_det[N - n:].to(x.dtype)[:, None]
        hs = []
        f0 = 0
        for g, (_, Fg) in enumerate(groups):
            xs = _input_transform(p, g, x[:, f0:f0 + Fg], seg_ids, nseg, training, update_running)
            f0 += Fg
            upd = xs * new_det                                    # track_mpnn.py:61
            if h_in is None:
                hs.append(upd)
            else:
                hs.append(torch.cat([h_in[:, g * H:(g + 1) * H], upd], dim=0))
    else:
        hs = [h_in[:, g * H:(g + 1) * H] for g in range(len(groups))]
    outs, atts = [], []
    for g in range(len(groups)):
        keep = None
        if training and cfg.nattheads > 0 and dropout_keep is not None:
            keep = dropout_keep[g]
        o, a = _factor_gru(p, cfg, g, hs[g], graph, keep)
        outs.append(o)
        atts.append(a)
    h_out = torch.cat(outs, dim=1)
    yn = F.linear(h_out, p['output_transform_node.weight'], p['output_transform_node.bias'])
    ye = F.linear(h_out, p['output_transform_edge.weight'], p['output_transform_edge.bias'])
    y = torch.where(is_det[:, None], yn, ye)
    return torch.sigmoid(y), y, h_out, tuple(atts)


def attention_to_reference_dense(alpha: torch.Tensor, graph: OracleGraph) -> torch.Tensor:
    """Scatter alpha [E,2] to the reference's dense [N,N] layout (det rows only; the reference
    fills edge rows / isolated det rows with the uniform 1/N of an all-masked softmax)."""
    out = torch.zeros(graph.N, graph.N, dtype=alpha.dtype)
    er = torch.from_numpy(graph.edge_row)
    out[torch.from_numpy(graph.src), er] = alpha[:, 0]
    out[torch.from_numpy(graph.dst), er] = alpha[:, 1]
    return out


# ----------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of the reference module (probed)."""
    H, K = cfg.nhidden, cfg.nattheads
    sh: Dict[str, Tuple[int, ...]] = {}
    for g, (_, Fg) in enumerate(cfg.groups):
        t = f'input_transforms.{g}.'
        sh[t + '0.weight'] = (H, Fg)
        sh[t + '0.bias'] = (H,)
        sh[t + '1.weight'] = (H,)
        sh[t + '1.bias'] = (H,)
        sh[t + '1.running_mean'] = (H,)
        sh[t + '1.running_var'] = (H,)
        sh[t + '1.num_batches_tracked'] = ()
        sh[t + '3.weight'] = (H, H)
        sh[t + '3.bias'] = (H,)
    for g in range(len(cfg.groups)):
        f = f'factor_grus.{g}.'
        xin = 2 * H if cfg.msg_type == 'concat' else H
        sh[f + 'edge_gru.weight_ih'] = (3 * H, xin)
        sh[f + 'edge_gru.weight_hh'] = (3 * H, H)
        sh[f + 'edge_gru.bias_ih'] = (3 * H,)
        sh[f + 'edge_gru.bias_hh'] = (3 * H,)
        for k in range(K):
            sh[f + f'gat.{k}.W_att'] = (H, H)
            sh[f + f'gat.{k}.a'] = (H, 1)
        sh[f + 'node_gru.weight_ih'] = (3 * H, H)
        sh[f + 'node_gru.weight_hh'] = (3 * H, H)
        sh[f + 'node_gru.bias_ih'] = (3 * H,)
        sh[f + 'node_gru.bias_hh'] = (3 * H,)
    G = len(cfg.groups)
    sh['output_transform_node.weight'] = (1, G * H)
    sh['output_transform_node.bias'] = (1,)
    sh['output_transform_edge.weight'] = (1, G * H)
    sh['output_transform_edge.bias'] = (1,)
    return sh


BUFFER_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked')


def random_params(cfg: OracleConfig, seed: int = 0, scale: float = 0.3) -> Dict[str, torch.Tensor]:
    """Parameters well away from the N(0, 0.01) init so parity errors are visible."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, s in param_shapes(cfg).items():
        if k.endswith('num_batches_tracked'):
            p[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith('running_var'):
            p[k] = 0.5 + torch.rand(s, generator=g)
        elif k.endswith('1.weight'):
            p[k] = 1.0 + scale * torch.randn(s, generator=g)
        else:
            p[k] = scale * torch.randn(s, generator=g)
    return p


# ----------------------------------------------------------------------------------------
# training targets and losses (reference models/loss.py) -- SURVEY 8(f) row 1
# ----------------------------------------------------------------------------------------
def _det_sets(graph: OracleGraph):
    """per det row: (past edge rows ascending, future edge rows ascending)"""
    past = {int(d): [] for d in graph.det_row}
    fut = {int(d): [] for d in graph.det_row}
    for e, s, d in zip(graph.edge_row, graph.src, graph.dst):
        fut[int(s)].append(int(e))
        past[int(d)].append(int(e))
    return past, fut


def create_targets(labels: torch.Tensor, graph: OracleGraph) -> torch.Tensor:
    """models/loss.py:8-44: dets keep their label; per det the LAST positive past edge and the FIRST positive
    future edge get target 1."""
    lab = labels.reshape(-1)
    targets = torch.zeros_like(lab)
    dr = torch.from_numpy(graph.det_row)
    targets[dr] = lab[dr]
    past, fut = _det_sets(graph)
    for d in graph.det_row:
        p = [e for e in past[int(d)] if lab[e] != 0]
        f = [e for e in fut[int(d)] if lab[e] != 0]
        if p:
            targets[p[-1]] = 1
        if f:
            targets[f[0]] = 1
    return targets


def ce_loss(logits: torch.Tensor, targets: torch.Tensor, graph: OracleGraph) -> torch.Tensor:
    """models/loss.py:77-115."""
    out = logits.reshape(-1)
    tg = targets.reshape(-1)
    past, fut = _det_sets(graph)
    loss = torch.zeros((), dtype=out.dtype)
    for d in graph.det_row:
        for rows, pick_last in ((past[int(d)], True), (fut[int(d)], False)):
            if not rows:
                continue
            pos = [i for i, e in enumerate(rows) if tg[e] != 0]
            if not pos:
                continue
            t = pos[-1] if pick_last else pos[0]
            sel = out[torch.tensor(rows)]
            loss = loss + (torch.logsumexp(sel, 0) - sel[t]) / len(rows)
    return loss


def focal_loss(outputs: torch.Tensor, targets: torch.Tensor, gamma=0.0, alpha=None, size_average=True) -> torch.Tensor:
    """models/loss.py:47-74 (eps 1e-10 inside the log)."""
    s = outputs.reshape(-1)
    t = targets.reshape(-1) != 0
    pt_in = torch.where(t, s, 1 - s)
    logpt = torch.log(pt_in + 1e-10)
    pt = torch.exp(logpt)
    if alpha is not None:
        a = torch.where(t, torch.tensor(float(alpha[1])), torch.tensor(float(alpha[0])))
        logpt = logpt * a
    loss = -1 * (1 - pt) ** gamma * logpt
    return loss.mean() if size_average else loss.sum()
