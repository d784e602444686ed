"""trackmpnn_amd: MI355X-native (gfx950) implementation of the TrackMPNN message-passing hot path.

    from trackmpnn_amd import TrackMPNN          # drop-in for reference models/track_mpnn.py
"""
from .graph import (CallPlan, DeviceGraph, FrameGraph, device_graph_from_adjacency, WindowBuilder, batch_windows, concat_static_graphs, dense_static_graph,
                    graph_from_adjacency, graph_from_edges, plan_single, synth_window)
from .capture import CapturedWindow
from .loss import CELoss, FocalLoss, create_targets
from .track_mpnn import SparseAttention, TrackMPNN
from .tracking import TrackGraph

__all__ = ['TrackMPNN', 'CapturedWindow', 'TrackGraph', 'SparseAttention', 'create_targets', 'CELoss', 'FocalLoss', 'FrameGraph', 'CallPlan', 'graph_from_adjacency', 'graph_from_edges',
           'plan_single', 'DeviceGraph', 'device_graph_from_adjacency', 'WindowBuilder', 'batch_windows', 'synth_window', 'dense_static_graph', 'concat_static_graphs']
