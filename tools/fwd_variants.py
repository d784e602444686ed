import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from trackmpnn_amd import TrackMPNN, _lib
dev = torch.device('cuda:0')
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
plans, xs, edge_iters = bench.build_batch(8192, 7, 6.0, 20, 8, seed=1, device=dev)
g = plans[-1].graph; H = 64; N, E = g.N, g.E
st = torch.cuda.current_stream().cuda_stream
P = dict(model.named_parameters()); f = 'factor_grus.0.'
h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); gates = torch.empty(4, N, H, device=dev)
wih, whh = P[f+'edge_gru.weight_ih'].detach(), P[f+'edge_gru.weight_hh'].detach()
wih_t, whh_t = wih.t().contiguous(), whh.t().contiguous()
bih, bhh = P[f+'edge_gru.bias_ih'].detach(), P[f+'edge_gru.bias_hh'].detach()
msg = torch.randn(E, H, device=dev)
def mk(save, xmode):
    def fn():
        _lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, xmode, g.src.data_ptr(), g.dst.data_ptr(), msg.data_ptr() if xmode == 0 else None, H, 1, H,
                  h.data_ptr(), H, H, wih_t.data_ptr(), whh_t.data_ptr(), bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H,
                  gates.data_ptr() if save else None, N*H, None, None, 0, st)
    return fn
for name, a in (('diff+gates', (1, 1)), ('diff nogates', (0, 1)), ('buf+gates', (1, 0)), ('buf nogates', (0, 0))):
    t = bench.time_stage(mk(*a)); print(name, round(t, 3), 'ms', round(12*H*H*E/t/1e9, 1), 'TF', flush=True)
proj = torch.empty(g.Dn, 3*H, device=dev)
def projk():
    _lib.call('tmpnn_rows_linear', g.det_row.data_ptr(), g.Dn, h.data_ptr(), H, H, wih_t.data_ptr(), 3*H, proj.data_ptr(), 3*H, st)
def mk3(save):
    def fn():
        _lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, 3, g.src_pos.data_ptr(), g.dst_pos.data_ptr(), proj.data_ptr(), 3*H, 0, H,
                  h.data_ptr(), H, H, None, whh_t.data_ptr(), bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H,
                  gates.data_ptr() if save else None, N*H, None, None, 0, st)
    return fn
print('rows_linear (P)', round(bench.time_stage(projk), 3), 'ms', flush=True)
for name, a in (('proj+gates', 1), ('proj nogates', 0)):
    t = bench.time_stage(mk3(a)); print(name, round(t, 3), 'ms', flush=True)
# cache-resident experiment: all rows folded into 4096 distinct rows -> gate stores stay in L2
rows_small = (g.edge_row % 4096).to(torch.int32).contiguous()
def mk2(save):
    def fn():
        _lib.call('tmpnn_gru_fwd', rows_small.data_ptr(), E, 0, None, None, msg.data_ptr(), H, 1, H,
                  h.data_ptr(), H, H, wih_t.data_ptr(), whh_t.data_ptr(), bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H,
                  gates.data_ptr() if save else None, N*H, None, None, 0, st)
    return fn
for name, a in (('L2-resident buf+gates', 1), ('L2-resident buf nogates', 0)):
    t = bench.time_stage(mk2(a)); print(name, round(t, 3), 'ms', round(12*H*H*E/t/1e9, 1), 'TF', flush=True)
