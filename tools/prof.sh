#!/bin/bash
# usage: tools/prof.sh <tag> <python script + args...>   -- rocprofv3 kernel stats of one command -> gpurun_out/<tag>/kernel_stats.csv + top rows
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 "$@" > $O/run.log 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
cp $(ls $O/stats/*kernel_stats.csv $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -rf $O/stats
tail -1 $O/run.log
python3 $R/tools/kstats.py $O ${TOPN:-16}
