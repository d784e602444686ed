timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for v in 0 1; do echo -n "SPLIT=$v "; TMPNN_SPLIT=$v timeout -k 10 200 python tools/stage_bench.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v['ms'] for k,v in d['stages'].items()})" || exit 1; done
