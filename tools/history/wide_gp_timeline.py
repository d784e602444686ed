"""s_memtime stamps of k_wide_gemm_pp256 (variant build -DW3_TIMELINE via TMPNN_LIB_PATH): steps 8..12 of block 0's second tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.graph import dense_static_graph, plan_single
dev = torch.device('cuda:0')
g = dense_static_graph(12, 300, 'cpu').to(dev)
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 256, 0, 'diff').to(dev).train()
x = torch.zeros(g.N, 8, device=dev); x[g.det_row.long()] = torch.randn(g.Dn, 8, device=dev)
s, l, h, _ = model.forward_graph(x, None, plan_single(g, g.N))
l.sum().backward()
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 256)()
raw = ctypes.CDLL(os.environ['TMPNN_LIB_PATH'])
assert raw.tmpnn_debug_pp_timeline(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(8, 32)
names = {0: ['mma', 'bar', 'load', 'wait+bar'], 1: ['load', 'wait+bar', 'mma', 'bar']}
print('epilogue ticks:', [int(t[w, 31] - t[w, 30]) for w in range(8)])
for w in (0, 2, 4, 6):
    hx = w >> 2
    for s_ in range(4):
        st_ = t[w, 6 * s_:6 * s_ + 5]
        print(f'wave {w} step {8 + s_}: ' + '  '.join(f'{n} {int(v)}' for n, v in zip(names[hx], np.diff(st_))) + f'   next step in {int(t[w, 6 * s_ + 6] - st_[4])}  step total {int(t[w, 6 * s_ + 6] - st_[0])}')
