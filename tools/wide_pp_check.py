"""Round 5: tmpnn_wide_gru_fwd_tiled (k_wide_gru_fwd_pp, or the ring form with TMPNN_WIDE_FWD_RING=1) against tmpnn_wide_gru_fwd
bit for bit on small graphs, at H = 128 / 256 / 384 (tests/gpu_stage_checks.py::check_wide_tiled)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import gpu_stage_checks as gs
from trackmpnn_amd.graph import dense_static_graph

bad = 0
for H in (128, 256, 384):
    for tag, g in (('dense 4x40', dense_static_graph(4, 40)), ('ragged batch', gs.make_graph(B=40, frames=7, mean=7, seed=3)),
                   ('small batch', gs.make_graph()), ('dense 6x90', dense_static_graph(6, 90))):
        r = gs.check_wide_tiled(H, g)
        print(H, tag, r, flush=True)
        bad += int(r['h_out bits'] != 0.0 or r['gates bits'] != 0.0)
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
