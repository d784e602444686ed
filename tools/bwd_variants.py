import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
dout = torch.randn(N, H, device=dev); dmsg = torch.empty(N, H, device=dev); dh = torch.empty(N, H, device=dev)
dyv = torch.randn(N, device=dev); w_head = torch.randn(H, device=dev)
_lib.call('tmpnn_gru_fwd', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0, H, h.data_ptr(), H, H, wih_t.data_ptr(), whh_t.data_ptr(), bih.data_ptr(), bhh.data_ptr(), out.data_ptr(), H, gates.data_ptr(), N*H, None, None, 0, st)
def mk(dho, dy, fuse):
    def fn():
        _lib.call('tmpnn_gru_bwd_data', g.edge_row.data_ptr(), E, H, h.data_ptr(), H, H, wih.data_ptr(), whh.data_ptr(),
                  gates.data_ptr(), N*H, dout.data_ptr() if dho else None, H, dyv.data_ptr() if dy else None, w_head.data_ptr() if dy else None,
                  dmsg.data_ptr(), H, dh.data_ptr(), H, g.src.data_ptr() if fuse else None, g.dst.data_ptr() if fuse else None, dmsg.data_ptr() if fuse else None, H, st)
    return fn
for name, args in (('up1', (1,0,0)), ('up2', (0,1,0)), ('up3', (1,1,0)), ('up1_fuse', (1,0,1)), ('up2_fuse', (0,1,1)), ('up3_fuse', (1,1,1))):
    print(name, round(bench.time_stage(mk(*args)), 3), 'ms', flush=True)

gW = [torch.zeros_like(wih), torch.zeros_like(whh), torch.zeros_like(bih), torch.zeros_like(bhh)]
wsb = _lib.load().tmpnn_gru_bwd_weights_ws(E, H, H); ws = torch.empty(wsb // 4 + 1, device=dev)
def wfn():
    _lib.call('tmpnn_gru_bwd_weights', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0, H, h.data_ptr(), H, H,
              gates.data_ptr(), N*H, dout.data_ptr(), H, dyv.data_ptr(), w_head.data_ptr(), gW[0].data_ptr(), gW[1].data_ptr(), gW[2].data_ptr(), gW[3].data_ptr(), ws.data_ptr(), wsb, st)
print('weights up3', round(bench.time_stage(wfn), 3), 'ms', flush=True)
fwsb = _lib.load().tmpnn_gru_bwd_fused_ws(E, H, H); fws = torch.empty(fwsb // 4 + 1, device=dev)
def ffn():
    _lib.call('tmpnn_gru_bwd_fused', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0, H, h.data_ptr(), H, H,
              wih.data_ptr(), whh.data_ptr(), gates.data_ptr(), N*H, dout.data_ptr(), H, dyv.data_ptr(), w_head.data_ptr(),
              dmsg.data_ptr(), H, dh.data_ptr(), H, g.src.data_ptr(), g.dst.data_ptr(), dmsg.data_ptr(), H,
              gW[0].data_ptr(), gW[1].data_ptr(), gW[2].data_ptr(), gW[3].data_ptr(), fws.data_ptr(), fwsb, st)
print('FUSED data+weights up3_fuse', round(bench.time_stage(ffn), 3), 'ms', flush=True)
for nm, dho, dy in (('up1', 1, 0), ('up2', 0, 1), ('up3', 1, 1)):
    def ffn2(dho=dho, dy=dy):
        _lib.call('tmpnn_gru_bwd_fused', g.edge_row.data_ptr(), E, 1, g.src.data_ptr(), g.dst.data_ptr(), None, 0, 0, H, h.data_ptr(), H, H,
                  wih.data_ptr(), whh.data_ptr(), gates.data_ptr(), N*H, dout.data_ptr() if dho else None, H,
                  dyv.data_ptr() if dy else None, w_head.data_ptr() if dy else None,
                  dmsg.data_ptr(), H, dh.data_ptr(), H, g.src.data_ptr(), g.dst.data_ptr(), dmsg.data_ptr(), H,
                  gW[0].data_ptr(), gW[1].data_ptr(), gW[2].data_ptr(), gW[3].data_ptr(), fws.data_ptr(), fwsb, st)
    print('one-pass', nm, 'fuse', round(bench.time_stage(ffn2), 3), 'ms', flush=True)
