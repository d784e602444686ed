"""The greedy inference loop of bench.py's C2 sequence alone, N sequences back to back (for rocprofv3 --kernel-trace --stats:
GPU kernel time per timestep against the loop's wall time per timestep -- is the loop bound by the device or by the host?)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.graph import synth_window
from trackmpnn_amd.loops import infer_sequence

tag = sys.argv[1] if len(sys.argv) > 1 else 'C2'
hung = len(sys.argv) > 2 and sys.argv[2] == 'hungarian'
dev = torch.device('cuda:0')
s = bench.LOOP_SHAPES[tag]
torch.manual_seed(5)
model = TrackMPNN('2d', s['ncat'], 64, 0, 'diff').to(dev).eval()
yy = synth_window(2001, bench.LOOP_INFER_FRAMES, s['mean'], s['mx'])
y = torch.from_numpy(yy)[None]
X = torch.randn(1, yy.shape[0], s['ncat'] + 5, generator=torch.Generator().manual_seed(3001))
for _ in range(3):
    infer_sequence(model, X, y, s['win'], 0, hung, dev)
torch.cuda.synchronize()
n, t0 = 50, time.perf_counter()
for _ in range(n):
    infer_sequence(model, X, y, s['win'], 0, hung, dev)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(json.dumps(dict(seq=tag, hungarian=hung, sequences=n + 3, frames=bench.LOOP_INFER_FRAMES, ms_per_timestep=round(ms / bench.LOOP_INFER_FRAMES, 4))))
