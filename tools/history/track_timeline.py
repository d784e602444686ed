"""s_memtime sums per phase of k_track_retire (variant build -DTK_TIMELINE, TMPNN_LIB_PATH) over the greedy / Hungarian inference
loops of bench.py's C2 / C3 sequences: associate | finalize | delete | gather | active, us per launch (s_memtime ticks taken as 1 ns: the phase sums then add up to the kernel's duration in the trace);
the hg: entries split the Hungarian sweep (both launches that run it: k_track_select and k_track_retire, per k_track_retire launch)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trackmpnn_amd import TrackMPNN
from trackmpnn_amd.graph import synth_window
from trackmpnn_amd.loops import infer_sequence

dev = torch.device('cuda:0')
raw = ctypes.CDLL(os.environ['TMPNN_LIB_PATH'])
buf = (ctypes.c_ulonglong * 16)()
names = ['associate', 'finalize', 'delete', 'gather', 'active', 'hg: build problem', 'hg: solve', 'hg: barrier after solve', 'hg: apply + loop top']
for tag in ('C2', 'C3'):
    s = bench.LOOP_SHAPES[tag]
    torch.manual_seed(5)
    model = TrackMPNN('2d', s['ncat'], 64, 0, 'diff').to(dev).eval()
    yy = synth_window(2001, bench.LOOP_INFER_FRAMES, s['mean'], s['mx'])
    y = torch.from_numpy(yy)[None]
    X = torch.randn(1, yy.shape[0], s['ncat'] + 5, generator=torch.Generator().manual_seed(3001))
    for hung in (False, True):
        for _ in range(2):
            infer_sequence(model, X, y, s['win'], 0, hung, dev)
        torch.cuda.synchronize()
        assert raw.tmpnn_debug_tk_timeline(buf, 1) == 0
        for _ in range(5):
            infer_sequence(model, X, y, s['win'], 0, hung, dev)
        torch.cuda.synchronize()
        assert raw.tmpnn_debug_tk_timeline(buf, 1) == 0
        n = max(int(buf[15]), 1)
        print(json.dumps(dict(seq=tag, hungarian=hung, launches=n,
                              us_per_launch={nm: round(buf[i] / n / 1000.0, 2) for i, nm in enumerate(names)})), flush=True)
