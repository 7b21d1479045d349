#!/bin/bash
# GPU box: the in-tree library against another build of it (default ray_tracing_amd/libmirt_prev.so) on one box, twice
# each, interleaved. usage: [CASES="..."] bash scripts/lib_ab.sh [other library] ; output: gpurun_out/lib_ab.txt
other=${1:-ray_tracing_amd/libmirt_prev.so}
out=gpurun_out/lib_ab.txt
: > $out
for rep in 1 2; do
  MIRT_LIB=$PWD/$other timeout -k 10 300 python scripts/ab.py --reps 2 ${CASES:-random final256 million million4} >> $out 2>&1 || exit 1
  timeout -k 10 300 python scripts/ab.py --reps 2 ${CASES:-random final256 million million4} >> $out 2>&1 || exit 1
done
cat $out
