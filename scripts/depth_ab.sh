#!/bin/bash
# usage (GPU box, repository root): scripts/depth_ab.sh — lanes active per vector instruction of the persistent kernel
# with primary rays only (max depth 1) against full paths, on the two sphere-hierarchy workloads: one rocprofv3 --pmc
# pass per case (scripts/pmc_kernels.sh), summaries concatenated into gpurun_out/depth_ab.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/depth_ab.txt
: > $OUT
for case in "random 1" "random 50" "million4 1" "million4 10" "final 1" "final 50"; do
  set -- $case
  bash scripts/pmc_kernels.sh d_$1_$2 $GRAFT_REPO_ROOT/scripts/ab.py --reps 1 --depth $2 $1 > /dev/null 2>&1 || { echo "pass $case failed" >> $OUT; continue; }
  echo "== $1 max depth $2" >> $OUT
  grep "Msamples" $GRAFT_REPO_ROOT/gpurun_out/pmck_d_$1_$2/a.log >> $OUT
  grep render_queue $GRAFT_REPO_ROOT/gpurun_out/pmck_d_$1_$2/summary.txt >> $OUT
done
cat $OUT
