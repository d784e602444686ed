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

from .functional import weight_cache
from .graph import DeviceGraph, device_graph_from_adjacency


class CapturedWindow:
    """calls: [(x, node_adj, edge_adj)] of one window in call order (what train.py:65-68,92-107 feeds the model).
    Models on the fused batch-1 path (nhidden <= 64: 32 / 64 natively, other widths zero-padded when they have no attention
    heads) are recorded through it; every other model (nhidden 128 ..., padded widths with heads) through the staged kernels
    on plans built before the capture.
    loss_fn(outputs, h_last) -> scalar, with outputs = [(scores, logits)] per call.  `optimizer` (optional) must be
    capturable (e.g. torch.optim.Adam(..., capturable=True)); parameter gradients must already exist (GradBucket or
    zero_grad(set_to_none=False)), because their addresses are baked into the graph."""

    def __init__(self, model, calls: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                 loss_fn: Callable, optimizer=None, bucket=None, warmup: int = 3):
        self.model, self.loss_fn, self.optimizer, self.bucket = model, loss_fn, optimizer, bucket
        dev = calls[0][0].device
        if dev.type != 'cuda':
            raise RuntimeError('CapturedWindow needs CUDA/HIP tensors (there is no CPU path)')
        from .graph import DG_BIG_ROWS, plan_single
        from .small import small_eligible
        nmax = max(int(na.shape[0]) for _, na, _ in calls)
        if nmax > DG_BIG_ROWS:
            raise RuntimeError(f'CapturedWindow: at most {DG_BIG_ROWS} rows per call (this window: {nmax})')
        # models outside the fused batch-1 path (attention heads, nhidden 128 / 256, padded widths) are recorded through the
        # STAGED kernels on prebuilt plans: launch sizes are host values fixed at capture time, nothing reads back
        padded = bool(getattr(model, '_padded', False))
        self.staged = not small_eligible(model, nmax) or (padded and model._small.att)
        self.static_x: List[torch.Tensor] = [x.detach().clone() for x, _, _ in calls]
        self.graphs: List[DeviceGraph] = [device_graph_from_adjacency(na, ea, dev) for _, na, ea in calls]
        for g in self.graphs:
            g.check()                                   # validate NOW: nothing can be read back during a capture
        self.inplace = not bool(getattr(model, '_padded', False))
        self.plans = None
        if self.staged:
            self.plans = [plan_single(g.frame_graph(), int(x.shape[0])) for g, (x, _, _) in zip(self.graphs, calls)]
        if any(p.requires_grad and p.grad is None for p in model.parameters()):
            raise RuntimeError('CapturedWindow: every trainable parameter needs a .grad buffer before the capture '
                               '(GradBucket(model) or zero_grad(set_to_none=False) after a first backward)')
        # Python's cyclic collector must not run inside the capture: it can release device memory (and whole graph pools) of
        # objects that died earlier -- a previous CapturedWindow's tensors held by a module cycle, say -- in the middle of the
        # recording, which ends in a crash at capture end.  Collect now, keep it off until the graph is instantiated.
        import gc
        if hasattr(model, '_drop_caches'):
            model._drop_caches()            # (a gradient sink made by an earlier eager step lives on ANOTHER stream: see below)
        gc.collect()
        torch.cuda.synchronize(dev)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._step()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # recorded on the warm-up's stream: autograd runs a node's backward on the stream of its forward, and objects that
            # outlive a step (the model's gradient sink, created in the warm-up) would otherwise pull a second stream into
            # the capture -- hipStreamEndCapture does not survive that on this stack
            with torch.cuda.graph(self.graph, stream=side):
                self.loss, self.outputs, self.h_last = self._step()
        finally:
            if gc_was_on:
                gc.enable()

    def _step(self):
        # A captured step owns zero -> backward -> optimizer, so it always runs in the in-place gradient mode (the kernels add
        # each call's parameter gradients straight into the existing .grad buffers; GradBucket or not): no parameter is an
        # autograd input, so no AccumulateGrad node -- which belongs to the stream it was created on and is shared with any
        # autograd graph the caller still holds -- can pull another stream into the capture.  Padded widths need autograd
        # through their zero-padding ops and take the branch below.
        flag = self.model.inplace_param_grads
        self.model.inplace_param_grads = self.inplace
        try:
            return self._step_inner()
        finally:
            self.model.inplace_param_grads = flag

    def _step_inner(self):
        h, outs = None, []
        if getattr(self.model, '_pad_cache', None) is not None:
            self.model._pad_cache = None                # (zero-padded copies of an earlier step must not enter this one)
        with weight_cache():                            # (the weights are constant over the forward calls of one step)
            for c, (x, g) in enumerate(zip(self.static_x, self.graphs)):
                if self.staged:
                    s, l, h, _ = self.model.forward_graph(x, h, self.plans[c])
                else:
                    s, l, h, _ = self.model.forward_dgraph(x, h, g)
                outs.append((s, l))
        loss = self.loss_fn(outs, h)
        if self.bucket is not None:
            self.bucket.zero()
        else:
            grads = [p.grad for p in self.model.parameters() if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)             # (a few fused fill launches instead of one memset per parameter)
        if self.inplace:
            loss.backward()                             # (in-place mode: the kernels add into p.grad, no parameter is an autograd input)
        else:
            # padded widths: gradients as values, added into the existing .grad buffers (parameter hooks do not run inside a
            # captured step).  The parameters ARE autograd inputs here: create the window while no autograd graph of an
            # earlier eager step of this model is alive (drop its outputs first), see _step.
            params = [p for p in self.model.parameters() if p.requires_grad]
            grads = torch.autograd.grad(loss, params, allow_unused=True)
            pairs = [(p.grad, g) for p, g in zip(params, grads) if g is not None]
            if pairs:
                torch._foreach_add_([a for a, _ in pairs], [g for _, g in pairs])
        if self.optimizer is not None:
            self.optimizer.step()
        if getattr(self.model, '_pad_cache', None) is not None:
            self.model._pad_cache = None
        return loss.detach(), [(s.detach(), l.detach()) for s, l in outs], h.detach()

    def replay(self, xs: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """One training step of the window: refresh the features (same shapes), launch the graph.  Returns the
        static loss tensor (valid until the next replay); `outputs` / `h_last` hold the per-call scores / logits."""
        if xs is not None:
            for dst, src in zip(self.static_x, xs):
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
