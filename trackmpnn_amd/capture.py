"""Whole-window hipGraph capture: every forward call of ONE tracking window, the loss, the backward
and -- optionally -- the optimizer step recorded once and replayed as a single graph launch.

The batch-1 path never synchronises and only enqueues work (include/tmpnn.h), so a window's whole training step is
capturable as it stands: the graphs are converted BEFORE the capture (`DeviceGraph`s, device-side sizes), the features
live in static buffers that `replay()` refreshes, and the GRU operand images are rebuilt inside the captured region
so that a replay always sees the current weights.  A training loop that revisits the same chunk (every epoch does:
reference train.py:54) replays its graph instead of re-issuing ~40 launches and their host bookkeeping.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .graph import DeviceGraph, device_graph_from_adjacency


class CapturedWindow:
    """calls: [(x, node_adj, edge_adj)] of one window in call order (what train.py:65-68,92-107 feeds the model).
    loss_fn(outputs, h_last) -> scalar, with outputs = [(scores, logits)] per call.  `optimizer` (optional) must be
    capturable (e.g. torch.optim.Adam(..., capturable=True)); parameter gradients must already exist (GradBucket or
    zero_grad(set_to_none=False)), because their addresses are baked into the graph."""

    def __init__(self, model, calls: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                 loss_fn: Callable, optimizer=None, bucket=None, warmup: int = 3):
        self.model, self.loss_fn, self.optimizer, self.bucket = model, loss_fn, optimizer, bucket
        dev = calls[0][0].device
        if dev.type != 'cuda':
            raise RuntimeError('CapturedWindow needs CUDA/HIP tensors (there is no CPU path)')
        from .small import small_eligible
        nmax = max(int(na.shape[0]) for _, na, _ in calls)
        if getattr(model, '_padded', False) or not small_eligible(model, nmax):
            raise RuntimeError('CapturedWindow records the fused batch-1 path only: no attention heads, nhidden 32 or 64 '
                               f'(this model: nhidden={model.nhidden}, heads={model.spec.K}), at most 65535 rows per call '
                               f'(this window: {nmax}); other models run eagerly')
        self.static_x: List[torch.Tensor] = [x.detach().clone() for x, _, _ in calls]
        self.graphs: List[DeviceGraph] = [device_graph_from_adjacency(na, ea, dev) for _, na, ea in calls]
        for g in self.graphs:
            g.check()                                   # validate NOW: nothing can be read back during a capture
        if any(p.requires_grad and p.grad is None for p in model.parameters()):
            raise RuntimeError('CapturedWindow: every trainable parameter needs a .grad buffer before the capture '
                               '(GradBucket(model) or zero_grad(set_to_none=False) after a first backward)')
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._step()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.outputs, self.h_last = self._step()

    def _step(self):
        h, outs = None, []
        for x, g in zip(self.static_x, self.graphs):
            s, l, h, _ = self.model.forward_dgraph(x, h, g)
            outs.append((s, l))
        loss = self.loss_fn(outs, h)
        if self.bucket is not None:
            self.bucket.zero()
        else:
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
        loss.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss.detach(), [(s.detach(), l.detach()) for s, l in outs], h.detach()

    def replay(self, xs: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """One training step of the window: refresh the features (same shapes), launch the graph.  Returns the
        static loss tensor (valid until the next replay); `outputs` / `h_last` hold the per-call scores / logits."""
        if xs is not None:
            for dst, src in zip(self.static_x, xs):
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
