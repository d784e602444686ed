"""s_memtime stamps of k_wide_gru_fwd_pp (variant build -DW3_TIMELINE, TMPNN_LIB_PATH): per wave, steps 4..8 of block 0's second
item: X: step start | MFMAs issued | barrier | requests issued | reads + split | wait + barrier ; Y: start | requests | reads +
split | wait + barrier | MFMAs issued | barrier.  Prints the phase durations in ticks (s_memtime: 100 MHz on gfx950)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trackmpnn_amd import _lib
from trackmpnn_amd.graph import build_edge_tiles, dense_static_graph
dev = torch.device('cuda:0'); H = 256
g = dense_static_graph(12, 300, 'cpu').to(dev)
tiles = build_edge_tiles(g, 128)
torch.manual_seed(0)
h = torch.randn(g.N, H, device=dev); sc = 1.0 / H ** 0.5
wih, whh = sc * torch.randn(3 * H, H, device=dev), sc * torch.randn(3 * H, H, device=dev)
bih, bhh = 0.3 * torch.randn(3 * H, device=dev), 0.3 * torch.randn(3 * H, device=dev)
lib = _lib.load()
prep = torch.empty(int(lib.tmpnn_wide_prep_bytes(H, H)), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call('tmpnn_wide_prepare', wih.data_ptr(), whh.data_ptr(), H, H, prep.data_ptr(), st)
P = torch.empty(g.Dn, 3 * H, device=dev); out = torch.zeros(g.N, H, device=dev); gates = torch.zeros(4, g.N, H, device=dev)
for _ in range(3):
    _lib.call('tmpnn_wide_gru_fwd_tiled', prep.data_ptr(), g.det_row.data_ptr(), g.Dn, tiles.cref(), g.E, h.data_ptr(), H, H,
              bih.data_ptr(), bhh.data_ptr(), P.data_ptr(), out.data_ptr(), H, gates.data_ptr(), g.N * H, st)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 256)()
raw = ctypes.CDLL(os.environ['TMPNN_LIB_PATH'])
assert raw.tmpnn_debug_pp_timeline(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(8, 32)
t0 = t[:, :31].min()
names = {0: ['mma', 'bar', 'reads+req issued', 'lgkm wait', 'split'], 1: ['reads+req issued', 'split', 'wait+bar', 'mma', 'bar']}
print('epilogue ticks per wave:', [int(t[w, 30] - t[w, 31]) for w in range(8)], ' epilogue start offsets:', [int(t[w, 31] - t[:, 31].min()) for w in range(8)])
for w in range(8):
    hx = w >> 2
    print(f'wave {w} ({"XY"[hx]}): first stamp +{t[w, 0] - t0}')
    for s_ in range(5):
        st_ = t[w, 6 * s_:6 * s_ + 6]
        nxt = t[w, 6 * s_ + 6] if s_ < 4 else t[w, 30]
        d = np.diff(st_)
        print(f'   step {4 + s_}: ' + '  '.join(f'{n} {int(x)}' for n, x in zip(names[hx], d)) + f'   | to next stamp {int(nxt - st_[5])}   step total {int(nxt - st_[0])}')
