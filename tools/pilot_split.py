"""Time + check tmpnn_rows_linear (fp32 MFMA vs the bf16x6 split, TMPNN_SPLIT=0/1) on R rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trackmpnn_amd import _lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
H = 64
torch.manual_seed(0)
dev = 'cuda'
x = torch.randn(R, H, device=dev) * torch.exp(torch.randn(R, 1, device=dev))
w = torch.randn(3 * H, H, device=dev) / 8
wt = w.t().contiguous()
rows = torch.randperm(R, device=dev, dtype=torch.int32)
out = torch.empty(R, 3 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.call('tmpnn_rows_linear', rows.data_ptr(), R, x.data_ptr(), H, H, wt.data_ptr(), 3 * H, out.data_ptr(), 3 * H, st)
run(); torch.cuda.synchronize()
n = 200_000
ref = (x[rows[:n].long()].double() @ w.double().t())
scale = (x[rows[:n].long()].double().abs() @ w.double().abs().t())
err = ((out[:n].double() - ref).abs() / scale).max().item()
f32 = (x[rows[:n].long()] @ w.t())
err32 = ((f32.double() - ref).abs() / scale).max().item()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): run()
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'TMPNN_SPLIT={os.environ.get("TMPNN_SPLIT", "(default 1)")} R={R} ms={ms:.3f} TF={2*R*H*3*H/ms/1e9:.1f} '
      f'max|err|/sum|a.b|={err:.3e} (torch f32 matmul: {err32:.3e})')
