"""Drop-in `TrackMPNN` (reference/models/track_mpnn.py:8-75) running on the MI355X HIP kernels.

Same constructor, same `state_dict` keys, same
`forward(x, h_in, node_adj, edge_adj) -> (scores, logits, h_out, attention)` signature, so the
reference's train.py / infer.py call pattern works unchanged (train.py:320-325, 68, 107;
infer.py:106-110, 51, 75).  Added on top: `forward_graph(x, h_in, plan)` takes a prebuilt
`CallPlan` (graph already in index form and resident on the device; many windows may be batched
block-diagonally), which is what the benchmark and any loop that owns its graphs should call.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .functional import MPIteration, ModelSpec
from .graph import CallPlan, FrameGraph, graph_from_adjacency, plan_single
from .layers import FactorGraphGRU


class SparseAttention:
    """Attention weights of one head in CSR (per det, per incident edge) form.

    The reference returns a dense [N, N] matrix per head (models/layers.py:35-37) whose only
    informative entries are (det row, incident edge row); `to_reference_dense()` rebuilds that
    matrix, including the uniform 1/N rows an all-masked softmax produces for edge rows and
    isolated dets (eval mode; in train mode the reference additionally applies a dense dropout to
    those uninformative rows, which is not reproduced).
    """

    def __init__(self, graph: FrameGraph, alpha: torch.Tensor):
        self.graph = graph
        self.alpha = alpha          # [2E] CSR order

    def to_reference_dense(self) -> torch.Tensor:
        g = self.graph
        N = g.N
        out = torch.full((N, N), 1.0 / max(N, 1), dtype=self.alpha.dtype, device=self.alpha.device)
        counts = (g.rowptr[1:] - g.rowptr[:-1]).long()
        det_of_pos = torch.repeat_interleave(g.det_row.long(), counts)
        has = g.det_row.long()[counts > 0]
        out[has] = 0.0
        out[det_of_pos, (g.inc & 0x7FFFFFFF).long()] = self.alpha
        return out

    def per_edge(self) -> torch.Tensor:
        """[E, 2]: weight the src det / the dst det gives each edge (oracle layout)."""
        g = self.graph
        e, ep = g.inc_edge_endpoint()
        out = torch.zeros((g.E, 2), dtype=self.alpha.dtype, device=self.alpha.device)
        out[e, ep] = self.alpha
        return out


class TrackMPNN(nn.Module):
    def __init__(self, features, ncategories, nhidden, nattheads, msg_type):
        super().__init__()
        if nhidden not in (32, 64, 128, 256):
            raise ValueError(f'nhidden={nhidden}: the gfx950 kernels support 32, 64, 128 or 256')
        self.input_transforms = nn.ModuleList([])
        self.factor_grus = nn.ModuleList([])
        self.feature_idx = []
        self.nhidden = nhidden
        groups = []
        nfeatures = 0
        for key, width in (('2d', ncategories + 5), ('temp', 2), ('vis', 128)):     # track_mpnn.py:17-33
            if key in features:
                self.input_transforms.append(self.get_input_transform(width, nhidden))
                self.factor_grus.append(FactorGraphGRU(nhidden, nattheads, msg_type, True))
                self.feature_idx.append(list(range(nfeatures, nfeatures + width)))
                groups.append((key, width))
                nfeatures += width
        if not groups:
            raise ValueError("features must contain at least one of '2d', 'temp', 'vis'")
        self.output_transform_node = nn.Linear(len(groups) * nhidden, 1, bias=True)
        self.output_transform_node.weight.data.normal_(mean=0.0, std=0.01)
        self.output_transform_node.bias.data.uniform_(+4.595, +4.595)
        self.output_transform_edge = nn.Linear(len(groups) * nhidden, 1, bias=True)
        self.output_transform_edge.weight.data.normal_(mean=0.0, std=0.01)
        self.output_transform_edge.bias.data.uniform_(-4.595, -4.595)
        self.output_activation = nn.Sigmoid()
        self.spec = ModelSpec(tuple(groups), nhidden, max(int(nattheads), 0), msg_type)
        self._graph_cache = None
        # set by trackmpnn_amd.dist.GradBucket: parameter gradients are added straight into p.grad (functional.py)
        self.inplace_param_grads = False

    def get_input_transform(self, n_in, n_out):
        lin1 = nn.Linear(n_in, n_out, bias=True)
        lin1.weight.data.normal_(mean=0.0, std=0.01)
        lin1.bias.data.uniform_(0, 0)
        lin2 = nn.Linear(n_out, n_out, bias=True)
        lin2.weight.data.normal_(mean=0.0, std=0.01)
        lin2.bias.data.uniform_(0, 0)
        return nn.Sequential(lin1, nn.BatchNorm1d(n_out), nn.ReLU(), lin2)

    # ------------------------------------------------------------------------------------------
    def _params_and_buffers(self):
        named = dict(self.named_parameters())
        params = [named[nm] for nm in self.spec.param_names()]
        buffers = dict(self.named_buffers())
        return params, buffers

    def forward_graph(self, x: torch.Tensor, h_in: Optional[torch.Tensor], plan: CallPlan,
                      dropout_keep: Optional[Sequence[torch.Tensor]] = None, reserve_rows: int = 0):
        """One message-passing call on a prebuilt CallPlan (see trackmpnn_amd.graph).

        dropout_keep: optional per-group uint8 [K, 2E] keep masks (CSR order) replacing the
        internally drawn attention dropout (train mode, nattheads > 0).
        reserve_rows: rows the NEXT call will append; h_out is then allocated with that much spare room
        and the next call extends it in place instead of copying the carried state (only valid when
        each h_out is continued from at most once, as in the reference's train / infer loops).
        """
        if not x.is_cuda:
            raise RuntimeError(f'x is on {x.device}: trackmpnn_amd runs on the MI355X HIP kernels only '
                               '(no CPU or torch fallback exists)')
        params, buffers = self._params_and_buffers()
        need_grad = torch.is_grad_enabled() and (
            x.requires_grad or (h_in is not None and h_in.requires_grad) or any(p.requires_grad for p in params))
        call = dict(spec=self.spec, plan=plan, buffers=buffers, training=self.training, need_grad=need_grad,
                    keep=dropout_keep, reserve=reserve_rows,
                    h_spare=getattr(h_in, '_tmpnn_spare_rows', 0) if h_in is not None else 0,
                    param_objs=params, inplace=self.inplace_param_grads)
        scores, logits, h_out = MPIteration.apply(call, x, h_in, *params)
        h_out._tmpnn_spare_rows = max(int(reserve_rows), 0)
        attention = tuple(None if a is None else [SparseAttention(plan.graph, ak) for ak in a]
                          for a in call['alphas'])
        return scores, logits, h_out, attention

    def forward(self, x, h_in, node_adj, edge_adj):
        """reference/models/track_mpnn.py:54-75.  node_adj / edge_adj: dense or sparse-COO [N, N]."""
        if not x.is_cuda:
            raise RuntimeError(f'x is on {x.device}: trackmpnn_amd runs on the MI355X HIP kernels only '
                               '(no CPU or torch fallback exists)')
        key = (id(node_adj), id(edge_adj), int(node_adj.shape[0]))
        if self._graph_cache is not None and self._graph_cache[0] == key:
            graph = self._graph_cache[1]
        else:
            graph = graph_from_adjacency(node_adj.to(x.device), edge_adj.to(x.device))
            self._graph_cache = (key, graph, node_adj, edge_adj)   # keep the tensors alive: id() stays unique
        plan = plan_single(graph, int(x.shape[0]))
        return self.forward_graph(x, h_in, plan)
