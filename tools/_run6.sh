cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 300 python3 -c "
import sys; sys.path.insert(0,'.')
import torch
from tests import gpu_stage_checks as c
from trackmpnn_amd.graph import dense_static_graph
g = c.make_graph()
for H in (32, 64):
    for tag, gt, order in (('small batch', g, None), ('ragged batch', c.make_graph(B=40, frames=7, mean=7, seed=3), None), ('dense 4x40 blocks', dense_static_graph(4, 40), 'blocks'), ('dense 3x70 rows', dense_static_graph(3, 70), 'rows')):
        print(H, tag, c.check_fwd_tiles(H, gt, order), flush=True)
" > gpurun_out/r03a/fwdtiles_check.log 2>&1 ; echo "check rc=$?"; tail -9 gpurun_out/r03a/fwdtiles_check.log
for v in 0 1; do TMPNN_FWD_TILED=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r03a/bench_tiled$v.json 2> gpurun_out/r03a/bench_tiled$v.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r03a/bench_tiled$v.json') if l.startswith('{\"metric')][0]); print('tiled=$v', d['value'], d['ms_per_step'], d['roofline'], {k:v for k,v in d['stage_roofs'].items() if 'fwd' in k})"; done
