import sys, os, subprocess
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for d in (0, 1, 2, 3):
    env = dict(os.environ, TMPNN_ONE_DBG=str(d))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bwd_variants.py')], env=env, capture_output=True, text=True)
    print('dbg', d, [l for l in r.stdout.splitlines() if 'FUSED' in l], r.stderr[-300:] if r.returncode else '', flush=True)
