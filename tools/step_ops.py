"""Which torch (ATen) ops does one bench step issue besides the library's kernels?  torch.profiler over two steps of
bench.step at --windows W; prints every ATen op with a device kernel, grouped by (name, input shapes): count per step and
device time -- the list the 'torch glue' line of profiles/rNN_c2_step_kernels.md is made of."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--windows', type=int, default=16384)
ap.add_argument('--workload', default='c2')
a = ap.parse_args()
dev = torch.device('cuda:0')
import __graft_entry__
__graft_entry__.build()
wl = bench.make_workload(a.workload, 0, dev, windows=a.windows)
for _ in range(3):
    wl['step']()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(STEPS):
        wl['step']()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    if ev.name.startswith('aten::') and dt > 0 and not [c for c in ev.cpu_children if c.name.startswith('aten::')]:
        k = (ev.name, str(ev.input_shapes)[:90])
        acc[k][0] += 1
        acc[k][1] += dt
tot = 0.0
for (name, shp), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f'{us / STEPS / 1e3:8.3f} ms/step {n / STEPS:6.1f} calls/step  {name:32s} {shp}')
print(f'total ATen device time {tot / STEPS / 1e3:.3f} ms/step')
