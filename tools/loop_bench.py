"""bench.py's loop_batch1 block alone (train chunks, greedy / Hungarian inference loops of C2 / C3 / C4-shaped sequences)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.loop_batch1()
print(json.dumps({k: {kk: (vv if not isinstance(vv, dict) else {a: b for a, b in vv.items() if a in ('ms_per_timestep', 'ms_per_chunk', 'tracks', 'stages_ms')}) for kk, vv in v.items()} for k, v in r.items() if isinstance(v, dict)}))
