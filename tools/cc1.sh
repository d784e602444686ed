#!/bin/bash
# compile ONE kernel file for gfx950 with the ISA kept, and print the register / scratch use of its kernels
# usage: tools/cc1.sh wide [kernel-name-substring] [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
f=$1; shift; pat=${1:-k_}; shift || true
mkdir -p $ROOT/trackmpnn_amd/lib/obj
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$ROOT/include "$@" -c $ROOT/trackmpnn_amd/csrc/$f.hip \
    -o $ROOT/trackmpnn_amd/lib/obj/$f.o -save-temps=obj
S=$ROOT/trackmpnn_amd/lib/obj/$f-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$S" "$pat" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    name, body = m.group(1), m.group(2)
    if sys.argv[2] not in name or name.endswith('.kd'):
        continue
    g = lambda k: re.search(k + r':\s+(\d+)', body)
    vals = {k: (g(k).group(1) if g(k) else '?') for k in ('vgpr_count', 'vgpr_spill_count', 'sgpr_count', 'private_segment_fixed_size', 'group_segment_fixed_size')}
    print(f"{name[:70]:70s} vgpr={vals['vgpr_count']} spill={vals['vgpr_spill_count']} sgpr={vals['sgpr_count']} scratch={vals['private_segment_fixed_size']}B lds={vals['group_segment_fixed_size']}")
PY
