#!/bin/bash
# Build a variant of libtmpnn.so with extra -D flags on ONE kernel file (kernel experiments; loaded with TMPNN_LIB_PATH).
# -DTMPNN_ABLATE is always passed: the wrong-result timing ablations (csrc/common.h) compile only in variant builds.
#   usage: tools/build_variant.sh <tag> <file-without-.hip> -DFLAG=1 ...   ->  trackmpnn_amd/lib/libtmpnn_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; f=$2; shift 2
O=$ROOT/trackmpnn_amd/lib/obj
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$ROOT/include -DTMPNN_ABLATE "$@" -c $ROOT/trackmpnn_amd/csrc/$f.hip -o $O/${f}_$tag.o
objs=""
for x in $(cd $ROOT/trackmpnn_amd/csrc && ls *.hip | sed s/.hip//); do
    if [ "$x" = "$f" ]; then objs="$objs $O/${f}_$tag.o"; else objs="$objs $O/$x.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/trackmpnn_amd/lib/libtmpnn_$tag.so $objs
echo "built trackmpnn_amd/lib/libtmpnn_$tag.so"
