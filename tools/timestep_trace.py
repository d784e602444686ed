"""Kernel launches of the two per-timestep loops (loops.py) at batch 1: run under `rocprofv3 --kernel-trace --stats` to get the
dispatch count per kernel; divided by the number of timesteps that is the launch list of one timestep.

    python tools/timestep_trace.py [--mode greedy|hungarian|train] [--reps 20]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.graph import synth_window
from trackmpnn_amd.loops import infer_sequence, train_chunk

ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='greedy')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--shape', default='C2')
a = ap.parse_args()
dev = torch.device('cuda:0')
s = bench.LOOP_SHAPES[a.shape]
torch.manual_seed(5)
model = TrackMPNN('2d', s['ncat'], 64, 0, 'diff')
gp = torch.Generator().manual_seed(4242)
with torch.no_grad():
    for k, prm in model.named_parameters():
        prm.add_(0.1 * torch.randn(prm.shape, generator=gp))
        if k.startswith('output_transform') and k.endswith('bias'):
            prm.copy_(0.5 * torch.randn(prm.shape, generator=gp))
model = model.to(dev)


def sequence(seed, frames):
    yy = synth_window(seed, frames, s['mean'], s['mx'])
    X = torch.randn(1, yy.shape[0], s['ncat'] + 5, generator=torch.Generator().manual_seed(seed + 1000))
    return X, torch.from_numpy(yy)[None]


if a.mode == 'train':
    X, y = sequence(1001, s['frames'])
    model.train()
    fn = lambda: train_chunk(model, X, y, dev)
    steps = None
else:
    X, y = sequence(2001, bench.LOOP_INFER_FRAMES)
    model.eval()
    fn = lambda: infer_sequence(model, X, y, s['win'], 0, a.mode == 'hungarian', dev)
for _ in range(3):
    r = fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    r = fn()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.reps * 1e3
print(json.dumps(dict(mode=a.mode, shape=a.shape, ms_per_call=round(ms, 3), model_calls=r[1], runs=a.reps + 3,
                      ms_per_model_call=round(ms / max(r[1], 1), 4))))
