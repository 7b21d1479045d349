#!/bin/bash
# A/B of the pooled hierarchy walk (MIRT_POOL=1) against the lane-by-lane walk on one box: scripts/ab.py per setting.
# usage: [CASES="random final256"] [STOPS="32 64"] [STEPS="1 2 4"] bash scripts/pool_ab.sh [output file]
set -e
out=${1:-gpurun_out/pool_ab.txt}
: > $out
for case in ${CASES:-random final256 million million4}; do
  MIRT_POOL=0 timeout -k 10 120 python scripts/ab.py --reps 2 $case >> $out 2>&1
  for stop in ${STOPS:-32 64 128}; do
    for steps in ${STEPS:-1}; do
      echo "# MIRT_POOL=1 MIRT_POOL_STOP=$stop MIRT_POOL_STEPS=$steps" >> $out
      MIRT_POOL=1 MIRT_POOL_STOP=$stop MIRT_POOL_STEPS=$steps timeout -k 10 120 python scripts/ab.py --reps 2 $case >> $out 2>&1
    done
  done
done
cat $out
