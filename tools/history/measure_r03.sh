#!/bin/bash
# Round-3 measurements on the GPU box (run through gpurun, one part per call: each fits a 1200 s limit):
#   bash tools/measure_r03.sh bench   rocprofv3 kernel stats of bench.py, FETCH/WRITE passes over the stage kernels, plain bench line
#   bash tools/measure_r03.sh c3      SURVEY 8(d) C3 sweep: B in {1, 64, 1024, 16384}, HIP events + FETCH/WRITE per B
#   bash tools/measure_r03.sh c5      C5: kernel stats, FETCH/WRITE per kernel, un-profiled step, SQ counters of the ring forward, CPU oracle sample
#   bash tools/measure_r03.sh var     K = 2 / concat / G = 3 variants with kernel stats, ragged 16 384-window batch, C3 / C4 shaped steps
# then here: python tools/collect_r03.py   -> profiles/r03_*
set -o pipefail
R=$GRAFT_REPO_ROOT
T=r03
part=${1:-bench}
cd /tmp && export TMPDIR=/tmp
case $part in
bench)
  rm -rf $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write
  echo "[measure] stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 10 --warmup 2 > $R/gpurun_out/${T}_bench.log 2> $R/gpurun_out/${T}_bench.err || { tail -5 $R/gpurun_out/${T}_bench.err; exit 1; }
  echo "[measure] fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_fetch.log 2>&1 || exit 1
  echo "[measure] write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/tools/stage_bench.py --windows 16384 > $R/gpurun_out/${T}_write.log 2>&1 || exit 1
  cd $R
  echo "[measure] plain bench"; python3 bench.py > gpurun_out/${T}_bench_plain.json 2> gpurun_out/${T}_bench_plain.err || { tail -5 gpurun_out/${T}_bench_plain.err; exit 1; }
  # keep the merged-back volume small: the per-dispatch traces are not needed, the stats / counter CSVs are
  find gpurun_out/${T}_stats gpurun_out/${T}_fetch gpurun_out/${T}_write -name '*kernel_trace.csv' -delete
  tail -c 600 gpurun_out/${T}_bench_plain.json
  ;;
c3)
  bash $R/tools/c3_sweep.sh
  ;;
c5)
  bash $R/tools/c5_profile.sh || exit 1
  bash $R/tools/wide_pmc.sh k_wide_g > $R/gpurun_out/${T}_c5/sq_counters.txt 2>&1
  tail -40 $R/gpurun_out/${T}_c5/sq_counters.txt
  ;;
var)
  O=$R/gpurun_out/${T}_var
  rm -rf $O; mkdir -p $O
  python3 $R/tools/variants_bench.py > $O/variants.log 2>&1 || { tail -5 $O/variants.log; exit 1; }
  for v in att_k2 concat g3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o r -- python3 $R/tools/variants_bench.py --only $v --steps 2 > $O/prof_$v.log 2>&1 || { echo "profile failed: $v"; continue; }
    cp $(ls $O/prof_$v/*kernel_stats.csv $O/prof_$v/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats_$v.csv
    rm -rf $O/prof_$v
  done
  cd $R
  python3 tools/ragged_bench.py > $O/ragged.log 2>&1 || tail -3 $O/ragged.log
  python3 tools/c3_profile.py > $O/c3.json 2> $O/c3.err || tail -3 $O/c3.err
  python3 tools/c4_profile.py > $O/c4.json 2> $O/c4.err || tail -3 $O/c4.err
  grep "^{" $O/variants.log | tail -1 | head -c 600; grep "^{" $O/ragged.log | tail -1
  ;;
*) echo "unknown part $part"; exit 2;;
esac
echo "[measure] $part done"
