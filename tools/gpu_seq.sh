#!/bin/bash
# Run the given commands (one per argument) one after another on the GPU box, each under its own `timeout -k 10`, logging to
# gpurun_out/<tag>_<i>.log.  An ordinary failure (a test's assertion) does not stop the sequence; a timeout / kill does:
# no further GPU step starts behind a step that had to be killed.
#   bash tools/gpu_seq.sh TAG SECONDS 'cmd 1' 'cmd 2' ...
tag=$1; lim=$2; shift 2
mkdir -p gpurun_out
i=0
for c in "$@"; do
  i=$((i + 1))
  echo "[seq $tag $i] $c"
  timeout -k 10 $lim bash -c "$c" > gpurun_out/${tag}_$i.log 2>&1
  rc=$?
  echo "[seq $tag $i] rc=$rc"; tail -3 gpurun_out/${tag}_$i.log | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[seq] step $i was killed at its limit: stopping"; exit $rc; fi
done
exit 0
