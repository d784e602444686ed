"""HIP-event timing of the one-launch input transform (tmpnn_input_tf_fwd / _bwd) on the C2 batch's calls: per call the launch
durations of the forward and the backward (the wave-owned form unless TMPNN_IT_WAVE=0)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN, _lib

dev = torch.device('cuda:0')
import __graft_entry__
if not os.environ.get('TMPNN_LIB_PATH'):
    __graft_entry__.build()
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
plans, xs, _ = bench.build_batch(16384, 7, 6.0, 20, 8, seed=1, device=dev)
P = dict(model.named_parameters()); B = dict(model.named_buffers())
t = 'input_transforms.0.'
st = torch.cuda.current_stream().cuda_stream
lib = _lib.load()
H, F = 64, 8
out = {}
for c, (plan, x) in enumerate(zip(plans, xs)):
    nd, S = int(plan.new_det_row.numel()), plan.S
    N = plan.graph.N
    h = torch.zeros(N, H, device=dev); dh = torch.randn(N, H, device=dev)
    y_save = torch.empty(nd, H, device=dev); mean = torch.empty(S, H, device=dev); rstd = torch.empty(S, H, device=dev)
    rm, rv = B[t + '1.running_mean'].clone(), B[t + '1.running_var'].clone()
    grads = [torch.zeros_like(P[t + k]) for k in ('0.weight', '0.bias', '1.weight', '1.bias', '3.weight', '3.bias')]
    wsb = int(lib.tmpnn_input_tf_bwd_ws(nd, S, H, F, 1)); ws = torch.empty(wsb // 4 + 1, device=dev)
    xr = plan.new_det_local

    def fwd():
        _lib.call('tmpnn_input_tf_fwd', x.data_ptr(), xr.data_ptr(), F, F, nd, plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(),
                  _lib.ptr(plan.seg_of_det), S, plan.max_seg_nd, H, 1, P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                  P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(), rm.data_ptr(), rv.data_ptr(), P[t + '3.weight'].data_ptr(),
                  P[t + '3.bias'].data_ptr(), y_save.data_ptr(), mean.data_ptr(), rstd.data_ptr(), plan.new_det_row.data_ptr(),
                  h.data_ptr(), H, st)

    def bwd():
        _lib.call('tmpnn_input_tf_bwd', x.data_ptr(), xr.data_ptr(), F, F, nd, plan.seg_ptr.data_ptr(), plan.seg_cnt.data_ptr(),
                  _lib.ptr(plan.seg_of_det), S, plan.max_seg_nd, H, 1, P[t + '0.weight'].data_ptr(), P[t + '0.bias'].data_ptr(),
                  P[t + '1.weight'].data_ptr(), P[t + '1.bias'].data_ptr(), P[t + '3.weight'].data_ptr(), y_save.data_ptr(),
                  mean.data_ptr(), rstd.data_ptr(), plan.new_det_row.data_ptr(), dh.data_ptr(), H, None, F, None,
                  grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), grads[4].data_ptr(),
                  grads[5].data_ptr(), ws.data_ptr(), wsb, st)
    out[f'call{c}'] = dict(nd=nd, S=S, max_seg=plan.max_seg_nd, fwd_us=round(bench.time_stage(fwd, 20) * 1e3, 1),
                           bwd_us=round(bench.time_stage(bwd, 20) * 1e3, 1))
print(json.dumps(out))
