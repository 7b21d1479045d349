#!/bin/bash
# GPU box: instruction counts and lanes per vector instruction of the wavefront kernel set, one queue against one
# queue per class, on Cornell and final (one rocprofv3 --pmc pass each). Output: gpurun_out/wf_classes_pmc.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/wf_classes_pmc.txt
: > $OUT
for k in wavefront wavefront_classes; do for case in cornell final; do
  bash scripts/pmc_kernels.sh wfc_${k}_$case $GRAFT_REPO_ROOT/scripts/ab.py --reps 1 --spp 16 --kernel $k $case > /dev/null 2>&1 || { echo "pass $k $case failed" >> $OUT; continue; }
  echo "== $k $case (16 spp, two renders)" >> $OUT
  grep "Msamples" $GRAFT_REPO_ROOT/gpurun_out/pmck_wfc_${k}_$case/a.log >> $OUT
  grep "wf_" $GRAFT_REPO_ROOT/gpurun_out/pmck_wfc_${k}_$case/summary.txt >> $OUT
done; done
cat $OUT
