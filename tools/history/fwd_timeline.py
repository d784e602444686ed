"""Per-phase cycle sums of k_gru_fwd_split_tiled (a -DFT_TIMELINE build, tools/build_variant.sh): where a wave's item goes.
usage: TMPNN_LIB_PATH=.../libtmpnn_fttl.so python3 tools/fwd_timeline.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from trackmpnn_amd import TrackMPNN, _lib
from trackmpnn_amd.graph import edge_tiles

dev = torch.device('cuda:0')
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
plans, xs, edge_iters = bench.build_batch(16384, 7, 6.0, 20, 8, seed=1, device=dev)
g = plans[-1].graph; H = 64; N, E = g.N, g.E
st = torch.cuda.current_stream().cuda_stream
P = dict(model.named_parameters()); f = 'factor_grus.0.'
h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); gates = torch.empty(4, N, H, device=dev)
wih_t = P[f + 'edge_gru.weight_ih'].detach().t().contiguous(); whh_t = P[f + 'edge_gru.weight_hh'].detach().t().contiguous()
bih, bhh = P[f + 'edge_gru.bias_ih'].detach(), P[f + 'edge_gru.bias_hh'].detach()
proj = torch.empty(g.Dn, 3 * H, device=dev)
_lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), g.Dn, h.data_ptr(), H, H, wih_t.data_ptr(), 3 * H, proj.data_ptr(), 3 * H, st)
tl = edge_tiles(g, 32)
whead = torch.randn(H, device=dev); part = torch.empty(8, N, device=dev)
lib = _lib.load()
fn_tl = lib.tmpnn_debug_ft_timeline
fn_tl.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
names = ['claim + operand wait + split + next requests', 'matrix phase', 'wait for staged P', 'P reads + h_prev + gate math',
         'head partial', 'h_out store', 'gate planes store', 'next P DMA requests']
for label, save, head in (('gates+head', 1, 1), ('no gates, head', 0, 1), ('gates, no head', 1, 0)):
    def fn():
        _lib.call('tmpnn_gru_fwd_tiles', tl.cref(), E, proj.data_ptr(), 3 * H, h.data_ptr(), H, H, whh_t.data_ptr(),
                  bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H, gates.data_ptr() if save else None, N * H,
                  whead.data_ptr() if head else None, part.data_ptr() if head else None, N, st)
    fn(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn_tl(buf, 1)
    ms = bench.time_stage(fn, iters=4)          # 1 + 4 launches
    torch.cuda.synchronize()
    fn_tl(buf, 1)
    items = buf[8]
    tot = sum(buf[i] for i in range(8))
    print(f'--- {label}: {ms:.3f} ms per launch; {items} items over 5 launches; {tot / items:.0f} memtime ticks per item')
    for i in range(8):
        print(f'   {names[i]:48s} {buf[i] / items:9.0f} ticks  {100.0 * buf[i] / tot:5.1f} %')
