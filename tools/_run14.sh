cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 200 python3 tools/stage_bench.py --windows 16384 2>&1 | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['stages']['gru_bwd_one_edge'])"
TMPNN_LIB_PATH=$PWD/trackmpnn_amd/lib/libtmpnn_twont.so timeout -k 10 200 python3 tools/stage_bench.py --windows 16384 2>&1 | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt loads', d['stages']['gru_bwd_one_edge'])"
done
