"""The reference's two per-timestep loops, composed from the device-resident pieces.

    train_chunk     reference/train.py:54-135   one tracking chunk: initialize_graph -> model -> targets + CE + focal
                                                losses, then per timestep update_graph(mode='train') -> model -> losses
                                                (hidden state carried, BPTT), ONE backward for the chunk
    infer_sequence  reference/infer.py:35-87    one sequence: per timestep update_graph(mode='test', greedy or Hungarian)
                                                -> model -> decode_tracks (track finalisation + rolling-window deletion)

Same order of operations, same arguments' meaning and the same results as the reference's drivers (which cannot be imported:
they parse the command line at import) -- but the graph, the hidden state, the losses' inputs and the tracks stay
in HBM: `TrackGraph` (tracking.py), the batch-1 model path (`TrackMPNN.forward_dgraph`) and the HIP losses (loss.py).  The
drivers' logging (F1 per forward, prints) is not part of the loop here or in the timings it is compared with.

`stages`, when given, is a dict that accumulates wall time per stage with a device synchronisation around each stage
(an instrumented pass: its total is larger than an un-instrumented one).
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch

from .loss import CELoss, FocalLoss, train_losses
from .tracking import TrackGraph


class _Stages:
    def __init__(self, acc: Optional[Dict[str, float]]):
        self.acc = acc
        self.t = 0.0

    def start(self):
        if self.acc is not None:
            torch.cuda.synchronize()
            self.t = time.perf_counter()

    def stop(self, name: str):
        if self.acc is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.acc[name] = self.acc.get(name, 0.0) + (now - self.t)
            self.t = now


def _loss_terms(tg: TrackGraph, scores, logits, ce, focal_node, focal_edge, tp_classifier: bool):
    """train.py:70-81 / :109-120 for one forward call."""
    # one autograd node for targets + CE + the focal terms (trackmpnn_amd.loss.train_losses: the same C entry points as the
    # CELoss / FocalLoss modules `ce`, `focal_node`, `focal_edge` -- train.py's gamma = 0, alpha = None -- would call one by one)
    # (the DeviceGraph itself: the one-launch losses take its arrays through the C struct, no tensor views)
    return train_losses(scores, logits, tg.labels_u8(), tg.graph, tp_classifier)


def train_chunk(model, X: torch.Tensor, y: torch.Tensor, device='cuda:0', tp_classifier: bool = True,
                stages: Optional[Dict[str, float]] = None):
    """One chunk of train.py:54-135 up to and including loss.backward().  X [1, ND, F], y [1, ND, 2] (host or device).
    Returns (loss, number of forward calls, sum of E over them) or None where the reference skips the chunk."""
    st = _Stages(stages)
    ce, focal_node, focal_edge = CELoss(), FocalLoss(gamma=0, alpha=None), FocalLoss(gamma=0, alpha=None)
    st.start()
    init = TrackGraph.initialize(X, y, 0, 'train', device)
    if init is None:
        return None
    tg, feats, t_st, t_end = init
    st.stop('graph')
    scores, logits, h, _ = model.forward_dgraph(feats, None, tg.graph)
    st.stop('model_fwd')
    loss_c, loss_f = _loss_terms(tg, scores, logits, ce, focal_node, focal_edge, tp_classifier)
    st.stop('targets_losses')
    ncalls, edge_iters = 1, tg.E
    t_skip = t_st
    for t_cur in range(t_st, t_end):
        if t_cur < t_skip:
            continue
        if feats.shape[0] == 0 and h.shape[0] == 0:            # train.py:95-100: nothing carried over -> start again
            init = TrackGraph.initialize(X, y, t_cur, 'train', device)
            if init is None:
                break
            tg, feats, t_skip, _ = init
            h = None
        else:
            feats = tg.update(None, X, y, t_cur, mode='train')
        st.stop('graph')
        scores, logits, h, _ = model.forward_dgraph(feats, h, tg.graph)
        st.stop('model_fwd')
        lc, lf = _loss_terms(tg, scores, logits, ce, focal_node, focal_edge, tp_classifier)
        loss_c, loss_f = loss_c + lc, loss_f + lf
        st.stop('targets_losses')
        ncalls += 1
        edge_iters += tg.E
    loss = loss_c + loss_f
    loss.backward()
    st.stop('backward')
    return loss, ncalls, edge_iters


def infer_sequence(model, X: torch.Tensor, y: torch.Tensor, cur_win_size: int = 5, ret_win_size: int = 0,
                   use_hungarian: bool = False, device='cuda:0', tp_classifier: bool = True,
                   stages: Optional[Dict[str, float]] = None):
    """One sequence of infer.py:35-87 (model in eval mode).  Returns (y_out [ND, 2] int64 as the reference keeps it, number
    of forward calls, sum of E over them)."""
    st = _Stages(stages)
    yy = y[0].detach().cpu().numpy().astype('int64')
    y_out = yy.copy()
    y_out[:, 1] = -1
    fast, finfo, h_cap = _fast_greedy(model, use_hungarian, tp_classifier, stages)
    with torch.no_grad():
        st.start()
        init = TrackGraph.initialize(X, y, 0, 'test', device)
        if init is None:
            return y_out, 0, 0
        tg, feats, t_st, t_end = init
        st.stop('graph')
        scores, logits, h, _ = model.forward_dgraph(feats, None, tg.graph)
        sc = _pos_score(tg, scores, tp_classifier)
        st.stop('model_fwd')
        ncalls, edge_iters = 1, tg.E
        step_info = None
        done = []                                              # TrackGraphs abandoned by a re-initialisation
        t_skip = t_st
        n_added = int(feats.shape[0])                          # rows the last update added (feats.shape[0] of infer.py:62)
        for t_cur in range(t_st, t_end):
            if t_cur < t_skip:
                continue
            if n_added == 0 and h.shape[0] == 0:               # infer.py:62-68: the graph emptied -> initialise again
                init = TrackGraph.initialize(X, y, t_cur, 'test', device)
                if init is None:
                    break
                done.append(tg)
                prev = tg
                tg, feats, t_skip, _ = init
                tg.y_track.copy_(prev.y_track)                 # (the tracks finalised so far belong to the sequence)
                h = None
                n_added = int(feats.shape[0])
            else:
                t_upto = t_end if t_cur == t_end - 1 else t_cur - cur_win_size + 2
                if fast is not None and h is not None:
                    # steady state: the whole timestep (append, model call, decode, the one host read) in the native driver.
                    # Its model descriptor is built once per sequence (eval mode under no_grad: the parameters cannot change
                    # inside this call; the per-step field, the row count, is overwritten by the driver) -- building it
                    # sits between the host read of one timestep and the first launch of the next, i.e. on the critical path
                    if step_info is None:
                        step_info = finfo()
                    # every remaining timestep is offered: the driver runs them back to back and stops in front of the first
                    # one it does not take (no detections, a graph beyond the one-launch kernels, ...)
                    steps = [(t, t_end if t == t_end - 1 else t - cur_win_size + 2, t + 1 if t + 1 < t_end else -1)
                             for t in range(t_cur, t_end)]
                    r = tg.greedy_run_fast(fast, step_info, h, h_cap, steps, ret_win_size, use_hungarian, tp_classifier)
                    if r is not None:
                        h, sc, h_cap, n_done, edges = r
                        n_added = 1                            # (a native step only runs with D_t > 0 new detections)
                        ncalls += n_done
                        edge_iters += edges
                        t_skip = t_cur + n_done                # (the loop variable moves past the timesteps that ran)
                        continue
                feats = tg.update(sc, X, y, t_cur, mode='test', use_hungarian=use_hungarian)
                n_added = int(feats.shape[0])
            st.stop('graph')
            scores, logits, h, _ = model.forward_dgraph(feats, h, tg.graph)
            h_cap = 0
            sc = _pos_score(tg, scores, tp_classifier)
            st.stop('model_fwd')
            ncalls += 1
            edge_iters += tg.E
            t_upto = t_end if t_cur == t_end - 1 else t_cur - cur_win_size + 2
            h, sc = tg.decode(h, sc, None, t_upto, ret_win_size, use_hungarian=use_hungarian,
                              next_t=t_cur + 1 if t_cur + 1 < t_end else None)
            st.stop('decode')
        y_out[:, 1] = tg.tracks()[:y_out.shape[0]]
        st.stop('decode')
    return y_out, ncalls, edge_iters


def _fast_greedy(model, use_hungarian: bool, tp_classifier: bool, stages):
    """(native module, call-descriptor factory, 0) where a steady-state greedy timestep can run in csrc_host/fast_iter.cpp's
    greedy_step: models on the fused batch-1 path without attention heads, eval mode, greedy or (where the device solver takes
    the sequence's problems: the driver checks per timestep) Hungarian association, with or without the TP
    classifier (without: the iteration writes 1 as every detection's score, infer.py:77-80), no per-stage instrumentation;
    (None, None, 0) otherwise."""
    from .small import fast_module, small_eligible
    if stages is not None or model.training or getattr(model, '_padded', False):
        return None, None, 0
    sp = getattr(model, '_small', None)
    if sp is None or not sp.eligible or sp.att or not small_eligible(model, 1):
        return None, None, 0
    fast = fast_module()
    if fast is None or not hasattr(fast, 'greedy_run'):
        return None, None, 0
    if model._plist is None:
        named = dict(model.named_parameters())
        model._plist = [named[nm] for nm in model.spec.param_names()]
        model._bufs = dict(model.named_buffers())

    class _G:                     # fast_info() only reads .N of the graph it is given (overwritten per step by the driver)
        N = 0

    def finfo():
        params = model._plist
        return sp.fast_info(params, _G, sp.params(params), False, False, False, 0)

    return fast, finfo, 0


def _pos_score(tg: TrackGraph, scores: torch.Tensor, tp_classifier: bool) -> torch.Tensor:
    """P(positive) per row; without the TP classifier every detection counts as a true positive (infer.py:53-56)."""
    sc = scores[:, 0]
    if not tp_classifier:
        sc = sc.clone()
        sc[tg.graph.frame_graph().det_row.long()] = 1.0
    return sc
