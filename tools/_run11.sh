cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 300 python3 -c "
import sys; sys.path.insert(0,'.')
import torch
from tests import gpu_stage_checks as c
bad = 0
for H, F_, S in ((64, 8, 4), (64, 8, 150), (32, 13, 70), (64, 128, 40), (32, 2, 4)):
    for training in (True, False):
        r = c.check_input_bn(H, F_, training, S=S, fused=True)
        w = max(r.values()); bad += w > 2e-4
        print(H, F_, S, training, 'worst %.2e' % w, {k: '%.1e' % v for k, v in r.items() if v > 2e-4}, flush=True)
print('BAD' if bad else 'ALL OK')
" > gpurun_out/r03a/tf_check.log 2>&1 ; echo "check rc=$?"; tail -12 gpurun_out/r03a/tf_check.log
for v in 0 1; do TMPNN_INPUT_TF=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-latency --no-loops --no-cpu-baseline > gpurun_out/r03a/bench_tf$v.json 2> gpurun_out/r03a/bench_tf$v.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r03a/bench_tf$v.json') if l.startswith('{\"metric')][0]); print('input_tf=$v', d['value'], d['ms_per_step'])"; done
