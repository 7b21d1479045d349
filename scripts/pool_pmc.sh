#!/bin/bash
# GPU box: instruction counts, lanes per vector instruction and wait shares of the persistent kernel with the
# lane-by-lane walk and with the pooled walk (one rocprofv3 --pmc pass each). Output: gpurun_out/pool_pmc.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pool_pmc.txt
: > $OUT
for case in ${CASES:-random final million}; do for mode in "0 1" "1 1" "1 2" "1 4"; do
  set -- $mode
  export MIRT_POOL=$1 MIRT_POOL_STEPS=$2 MIRT_POOL_STOP=${STOP:-48}
  bash scripts/pmc_kernels.sh pool_${1}_${2}_$case $GRAFT_REPO_ROOT/scripts/ab.py --reps 1 $case > /dev/null 2>&1 || { echo "pass $mode $case failed" >> $OUT; continue; }
  echo "== $case MIRT_POOL=$1 MIRT_POOL_STEPS=$2 (two renders)" >> $OUT
  grep "Msamples" $GRAFT_REPO_ROOT/gpurun_out/pmck_pool_${1}_${2}_$case/a.log >> $OUT
  grep "render_queue" $GRAFT_REPO_ROOT/gpurun_out/pmck_pool_${1}_${2}_$case/summary.txt >> $OUT
done; done
cat $OUT
