"""Per-phase cycle sums of k_gru_bwd_two (a -DTWO_TIMELINE build, tools/build_variant.sh twotl gru -DTWO_TIMELINE): where a
wave's 32-row tile goes, by role (waves 0-3: W_ih side, waves 4-7: W_hh side).
usage: TMPNN_LIB_PATH=.../libtmpnn_twotl.so python3 tools/bwd_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from trackmpnn_amd import TrackMPNN, _lib

dev = torch.device('cuda:0')
torch.manual_seed(5)
model = TrackMPNN('2d', 3, 64, 0, 'diff').to(dev).train()
plans, xs, edge_iters = bench.build_batch(16384, 7, 6.0, 20, 8, seed=1, device=dev)
lib = _lib.load()
fn = lib.tmpnn_debug_two_timeline
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
t, flops, nbytes = bench.stage_profile(model, plans[-1], 64)          # runs every stage kernel incl. the one-pass backward
fn(buf, 0)
names = ['dW operand reads (transposing LDS reads)', 'staging slice + re-requests (waits for its rows)', 'dW: 6 MFMA 32x32x16 + bias dots',
         'row operand reads + 12 MFMA 16x16x32 (data product)', 'next ids + epilogue stores', 'barrier']
print(f"one-pass backward: {t['gru_bwd_one_edge']:.3f} ms per launch (instrumented build)")
for role, rn in ((0, 'W_ih side (waves 0-3)'), (1, 'W_hh side (waves 4-7)')):
    v = [buf[role * 8 + i] for i in range(8)]
    tiles = v[6]
    tot = sum(v[:6])
    print(f'--- {rn}: {tiles} wave-tiles, {tot / max(tiles, 1):.0f} ticks per tile')
    for i in range(6):
        print(f'   {names[i]:58s} {v[i] / max(tiles, 1):8.0f} ticks {100.0 * v[i] / max(tot, 1):5.1f} %')
