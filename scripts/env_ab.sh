#!/bin/bash
# GPU box: one environment switch of the library off (0) and on (1) on one box, twice each, interleaved.
# usage: [CASES="..."] [VALUES="0 1"] bash scripts/env_ab.sh VARIABLE ; output: gpurun_out/env_ab_VARIABLE.txt
var=$1
out=gpurun_out/env_ab_$var.txt
: > $out
for rep in 1 2; do
  for v in ${VALUES:-0 1}; do
    echo "# $var=$v" >> $out
    env $var=$v timeout -k 10 300 python scripts/ab.py --reps 2 $AB_FLAGS ${CASES:-random final256 million million4} >> $out 2>&1 || exit 1
  done
done
cat $out
