"""Parameter containers mirroring reference/models/layers.py.

The reference's `FactorGraphGRU` / `GraphAttentionLayer` are rebuilt here ONLY as holders of the
same `torch.nn` parameter objects, created in the same order and initialised the same way
(layers.py:11-24, 54-82), so that (a) `state_dict()` keys and shapes are identical and a
reference checkpoint loads with `strict=True`, and (b) under the same `torch.manual_seed` the
RNG stream is consumed identically, giving bit-identical initial weights.  Their compute lives in
the HIP kernels (functional.py); calling `.forward` on these containers is an error.
"""
import torch
import torch.nn as nn


class GraphAttentionLayer(nn.Module):
    """W_att [H, H], a [H, 1], xavier-uniform gain 1.414 (reference/models/layers.py:18-21)."""

    def __init__(self, in_features, out_features, alpha=0.2, concat=False):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.alpha = alpha
        self.concat = concat
        self.W_att = nn.Parameter(torch.zeros(size=(in_features, out_features)))
        nn.init.xavier_uniform_(self.W_att.data, gain=1.414)
        self.a = nn.Parameter(torch.zeros(size=(out_features, 1)))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)

    def forward(self, *args, **kwargs):
        raise RuntimeError('parameter container only: attention runs inside libtmpnn (tmpnn_att_fwd)')

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class FactorGraphGRU(nn.Module):
    """edge_gru / gat / node_gru containers (reference/models/layers.py:50-82)."""

    def __init__(self, nhidden, nattheads=0, msg_type='diff', bias=True):
        super().__init__()
        self.nhidden = nhidden
        self.msg_type = msg_type
        self.nattheads = nattheads
        self.bias = bias
        if msg_type == 'concat':
            self.edge_gru = nn.GRUCell(2 * nhidden, nhidden, bias=bias)
        elif msg_type == 'diff':
            self.edge_gru = nn.GRUCell(nhidden, nhidden, bias=bias)
        else:
            raise AssertionError('Incorrect message type for model!')
        if nattheads <= 0:
            self.gat = None
        else:
            self.gat = nn.ModuleList([GraphAttentionLayer(nhidden, nhidden) for _ in range(nattheads)])
        self.node_gru = nn.GRUCell(nhidden, nhidden, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        for cell in (self.edge_gru, self.node_gru):
            cell.weight_ih.data.normal_(mean=0.0, std=0.01)
            cell.weight_hh.data.normal_(mean=0.0, std=0.01)
            if self.bias:
                cell.bias_ih.data.uniform_(0, 0)
                cell.bias_hh.data.uniform_(0, 0)

    def forward(self, *args, **kwargs):
        raise RuntimeError('parameter container only: the factor-graph update runs inside libtmpnn')

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.nhidden) + ' -> ' + str(self.nhidden) + ')'
