"""One window of a dense scene (BDD-like density over 8 frames, rows per call up to ~11 k) through the drop-in call:
fused batch-1 path against the staged path (TMPNN_SMALL_PATH=0 semantics), forward + loss + backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trackmpnn_amd.track_mpnn as tm
from tests.test_small_path_gpu import _dense_window_calls, DEV
from trackmpnn_amd import TrackMPNN

calls = _dense_window_calls(seed=3, frames=8, mean_dets=32, max_dets=45, F=8)
print('rows per call', [int(na.shape[0]) for _, na, _ in calls])
modes = (True,) if '--fused-only' in sys.argv else (True, False)
for small in modes:
    tm.SMALL_PATH = small
    torch.manual_seed(5)
    model = TrackMPNN('2d', 3, 64, 0, 'diff').to(DEV).train()

    def step():
        h, outs = None, []
        for x, na, ea in calls:
            s, l, h, _ = model(x, h, na, ea)
            outs.append(l)
        model.zero_grad(set_to_none=True)
        torch.cat(outs).sum().backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print('fused batch-1 path' if small else 'staged path       ', round((time.perf_counter() - t0) / 30 * 1e3, 3),
          'ms per window (fwd + loss + bwd)')
