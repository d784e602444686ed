import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import Golden
from tests.test_parity_gpu import DEV
from trackmpnn_amd import TrackMPNN, CapturedWindow
from trackmpnn_amd.dist import GradBucket
N = int(sys.argv[1])
gold = Golden('roll_c2_kitti_car_w5')
calls = []
for c in range(gold.ncalls):
    na, ea = gold.adjacency(c, 'node_adj', DEV), gold.adjacency(c, 'edge_adj', DEV)
    if not na.is_sparse: na, ea = na.to_sparse(), ea.to_sparse()
    calls.append((gold.t(f'c{c}/x').to(DEV), na, ea))
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 128, 0, 'diff').to(DEV).train()
bucket = GradBucket(model)
loss_fn = lambda outs, h: torch.cat([l for _, l in outs]).sum()
win = CapturedWindow(model, calls, loss_fn, optimizer=None, bucket=bucket)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): win.replay()
torch.cuda.synchronize()
print('replays', N, 'ms each', (time.perf_counter() - t0) / max(N, 1) * 1e3)
