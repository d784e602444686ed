cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 300 python3 -c "
import sys; sys.path.insert(0,'.')
import torch
from tests import gpu_stage_checks as c
from trackmpnn_amd.graph import dense_static_graph
g = c.make_graph()
for H in (128, 256):
    for tag, gt in (('dense 4x40', dense_static_graph(4, 40)), ('ragged batch', c.make_graph(B=40, frames=7, mean=7, seed=3)), ('small batch', g)):
        print(H, tag, c.check_wide_tiled(H, gt), flush=True)
" > gpurun_out/r03a/ring_check.log 2>&1 ; echo "check rc=$?"; tail -8 gpurun_out/r03a/ring_check.log
timeout -k 10 300 python3 tools/wide_fwd_bench.py > gpurun_out/r03a/wide_fwd_ring.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/r03a/wide_fwd_ring.log
