#!/bin/bash
# Round-4 measurements on the GPU box (through gpurun, one part per call):
#   bash tools/measure_r04.sh bench   rocprofv3 kernel stats of bench.py; FETCH / WRITE passes over the stage kernels (stage_bench.py)
#                                     and over the attention stage (att_bench.py); the plain bench line
#   bash tools/measure_r04.sh var     base / K = 2 / concat / G = 3 steps with kernel stats, one KITTI window eager / captured for the
#                                     models outside the plain fused path, C3 / C4-shaped steps, the C5 step
#   bash tools/c5_profile_r04.sh      C5: one stream (default) / auxiliary stream, kernel stats + FETCH / WRITE passes
#   bash tools/timestep_profile.sh    dispatches per model call of the two batch-1 loops
# then here: python tools/collect_r04.py   -> profiles/r04_*
set -o pipefail
R=$GRAFT_REPO_ROOT
T=r04
part=${1:-bench}
cd /tmp && export TMPDIR=/tmp
case $part in
bench)
  rm -rf $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write $R/gpurun_out/${T}_afetch $R/gpurun_out/${T}_awrite
  echo "[measure] stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 10 --warmup 2 > $R/gpurun_out/${T}_bench.log 2> $R/gpurun_out/${T}_bench.err || { tail -5 $R/gpurun_out/${T}_bench.err; exit 1; }
  echo "[measure] fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_fetch.log 2>&1 || exit 1
  echo "[measure] write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_write.log 2>&1 || exit 1
  echo "[measure] att fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_afetch -- python3 $R/tools/att_bench.py --windows 16384 --train --iters 3 > $R/gpurun_out/${T}_afetch.log 2>&1 || exit 1
  echo "[measure] att write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_awrite -- python3 $R/tools/att_bench.py --windows 16384 --train --iters 3 > $R/gpurun_out/${T}_awrite.log 2>&1 || exit 1
  echo "[measure] att stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_astats -- python3 $R/tools/att_bench.py --windows 16384 --train --iters 10 > $R/gpurun_out/${T}_astats.log 2>&1 || exit 1
  cd $R
  echo "[measure] plain bench"; python3 bench.py > gpurun_out/${T}_bench_plain.json 2> gpurun_out/${T}_bench_plain.err || { tail -5 gpurun_out/${T}_bench_plain.err; exit 1; }
  find gpurun_out/${T}_stats gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_afetch gpurun_out/${T}_awrite gpurun_out/${T}_astats -name '*kernel_trace.csv' -delete
  tail -c 400 gpurun_out/${T}_bench_plain.json
  ;;
var)
  O=$R/gpurun_out/${T}_var
  rm -rf $O; mkdir -p $O
  cd $R
  python3 tools/variants_bench.py > $O/variants.log 2>&1 || { tail -5 $O/variants.log; exit 1; }
  cd /tmp
  for v in att_k2 concat; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o r -- python3 $R/tools/variants_bench.py --only $v --steps 2 > $O/prof_$v.log 2>&1 || { echo "profile failed: $v"; continue; }
    cp $(ls $O/prof_$v/*kernel_stats.csv $O/prof_$v/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats_$v.csv
    rm -rf $O/prof_$v
  done
  cd $R
  python3 tools/staged_window.py > $O/staged_window.log 2>&1 || tail -3 $O/staged_window.log
  python3 tools/c3_profile.py > $O/c3.json 2> $O/c3.err || tail -3 $O/c3.err
  python3 tools/c4_profile.py > $O/c4.json 2> $O/c4.err || tail -3 $O/c4.err
  python3 tools/c5_bench.py > $O/c5.log 2>&1 || tail -3 $O/c5.log
  grep "^{" $O/variants.log | tail -1 | head -c 900; tail -3 $O/c5.log; cat $O/staged_window.log
  ;;
*) echo "unknown part $part"; exit 2;;
esac
echo "[measure] $part done"
