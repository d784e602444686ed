cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 400 python3 -c "
import sys, json; sys.path.insert(0,'.')
import torch, bench
import __graft_entry__; __graft_entry__.build()
torch.cuda.set_device(0)
print(json.dumps(bench.loop_batch1(), indent=1))
" > gpurun_out/r03a/loops.json 2> gpurun_out/r03a/loops.err; echo rc=$?; tail -3 gpurun_out/r03a/loops.err; python3 -c "
import json; d=json.load(open('gpurun_out/r03a/loops.json'))
for k,v in d['train'].items(): print('train',k,v)
for k,v in d['infer'].items(): print('infer',k,v)"
