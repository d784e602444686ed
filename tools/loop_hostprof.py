"""cProfile of the per-timestep loops at batch 1 (where the host time of a timestep goes).  python tools/loop_hostprof.py [greedy|train]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + ['--mode', sys.argv[1] if len(sys.argv) > 1 else 'greedy', '--reps', '1']
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'timestep_trace.py'))
fn = ns['fn']
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fn()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue()[:9000])
