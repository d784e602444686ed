#!/bin/bash
# Round measurements on the GPU box (run through gpurun): rocprofv3 kernel stats of bench.py, FETCH_SIZE / WRITE_SIZE
# passes over the stage kernels, the C3-shaped and C5 workloads, and one un-profiled bench line.
#   usage: bash tools/measure_round.sh r02        -> gpurun_out/r02_*; then `python tools/collect_profiles.py r02` here
set -o pipefail
T=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write $R/gpurun_out/${T}_c5prof
echo "[measure] stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 10 --warmup 2 > $R/gpurun_out/${T}_bench.log 2> $R/gpurun_out/${T}_bench.err || exit 1
echo "[measure] fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_fetch.log 2>&1 || exit 1
echo "[measure] write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_write.log 2>&1 || exit 1
cd $R
[ "$2" = "prof" ] && { echo "[measure] done (profiles only)"; exit 0; }
echo "[measure] c3"; python3 tools/c3_profile.py > gpurun_out/${T}_c3.json 2> gpurun_out/${T}_c3.err || exit 1
echo "[measure] c4"; python3 tools/c4_profile.py > gpurun_out/${T}_c4.json 2> gpurun_out/${T}_c4.err || exit 1
echo "[measure] c5"; python3 tools/c5_bench.py --steps 3 > gpurun_out/${T}_c5.json 2> gpurun_out/${T}_c5.err || exit 1
echo "[measure] plain bench"; python3 bench.py > gpurun_out/${T}_bench_plain.json 2> gpurun_out/${T}_bench_plain.err || exit 1
echo "[measure] done"
