#!/bin/bash
# The comparison build: every kernel file with -DTMPNN_KEEP_VARIANTS (the superseded forms and their entry points compiled in)
#   -> trackmpnn_amd/lib/libtmpnn_variants.so ; use it with TMPNN_LIB_PATH=... TMPNN_KEEP_VARIANTS=1 TMPNN_TEST_VARIANTS=1
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/trackmpnn_amd/lib/obj_variants
mkdir -p $O
pids=""
for f in $(cd $ROOT/trackmpnn_amd/csrc && ls *.hip | sed s/.hip//); do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$ROOT/include -DTMPNN_KEEP_VARIANTS -c $ROOT/trackmpnn_amd/csrc/$f.hip -o $O/$f.o &
    pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/trackmpnn_amd/lib/libtmpnn_variants.so $O/*.o
echo "built trackmpnn_amd/lib/libtmpnn_variants.so: $(nm -D --defined-only $ROOT/trackmpnn_amd/lib/libtmpnn_variants.so | grep -c ' T tmpnn_') entry points"
