"""Print (not assert) the error of the HIP path against every golden fixture -- a diagnostic for the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import golden_names
from tests.golden_util import Golden
from tests.test_parity_gpu import build_model, DEV
from trackmpnn_amd import graph_from_adjacency, plan_single

for name in golden_names():
    gold = Golden(name)
    meta = gold.meta
    model = build_model(meta, gold.params())
    K, G = meta['nattheads'], len(model.feature_idx)
    train = meta['mode'] == 'train'
    h = None
    loss = 0.0
    es, el, eh, ml = [], [], [], []
    for c in range(gold.ncalls):
        na, ea = gold.adjacency(c, 'node_adj', DEV), gold.adjacency(c, 'edge_adj', DEV)
        x = gold.t(f'c{c}/x').to(DEV).requires_grad_(True)
        graph = graph_from_adjacency(na, ea)
        keep = None
        if train and K > 0:
            e, ep = graph.inc_edge_endpoint()
            keep = [gold.t(f'c{c}/keep_g{g}').to(DEV)[:, e, ep].contiguous() for g in range(G)]
        scores, logits, h, att = model.forward_graph(x, h, plan_single(graph, x.shape[0]), dropout_keep=keep)
        es.append((scores.detach().cpu() - gold.t(f'c{c}/scores')).abs().max().item())
        rl = gold.t(f'c{c}/logits')
        d = (logits.detach().cpu() - rl).abs()
        el.append(d.max().item()); ml.append((d / (1 + rl.abs())).max().item())
        if gold.has(f'c{c}/h_out'):
            eh.append((h.detach().cpu() - gold.t(f'c{c}/h_out')).abs().max().item())
        loss = loss + (gold.t(f'c{c}/wl').to(DEV) * logits).sum() + (gold.t(f'c{c}/ws').to(DEV) * scores).sum()
    loss = loss + (gold.t('V').to(DEV) * h).sum()
    loss.backward()
    grads = gold.grads()
    gscale = max(1.0, max(v.abs().max().item() for k, v in grads.items() if k != 'X'))
    worst = max(((prm.grad.cpu() - grads[k]).abs().max().item() / gscale, k) for k, prm in model.named_parameters())
    print(f'{name:40s} score {max(es):.1e} logit {max(el):.1e} (rel {max(ml):.1e}) h {max(eh):.1e} '
          f'loss {abs(loss.item()-float(gold.d["loss"])):.1e} grad/gmax {worst[0]:.1e} ({worst[1]})', flush=True)
