import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
r = bench.loop_batch1()
print(json.dumps({k: {kk: (vv if not isinstance(vv, dict) else {a: b for a, b in vv.items() if a in ('ms_per_timestep', 'ms_per_chunk', 'tracks', 'stages_ms')}) for kk, vv in v.items()} for k, v in r.items() if isinstance(v, dict)}))
