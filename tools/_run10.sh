cd $GRAFT_REPO_ROOT
bash tools/c5_profile.sh 2>&1 | tail -6
mkdir -p gpurun_out/r03_var
timeout -k 10 500 python3 tools/variants_bench.py > gpurun_out/r03_var/variants.log 2>&1; echo "variants rc=$?"; tail -5 gpurun_out/r03_var/variants.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for v in att_k2 concat g3; do rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_var/prof_$v -o r -- python3 $GRAFT_REPO_ROOT/tools/variants_bench.py --only $v --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r03_var/prof_$v.log 2>&1; cp $(ls $GRAFT_REPO_ROOT/gpurun_out/r03_var/prof_$v/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r03_var/prof_$v/*/*kernel_stats.csv 2>/dev/null | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_var/kernel_stats_$v.csv; rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03_var/prof_$v; done
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/ragged_bench.py > gpurun_out/r03_var/ragged.log 2>&1; echo "ragged rc=$?"; tail -2 gpurun_out/r03_var/ragged.log
