#!/bin/bash
# GPU box: several builds of the library on one box, twice each, interleaved. usage: [CASES="..."] bash scripts/libs_ab.sh lib1.so lib2.so ...
# (paths relative to the repository root); output: gpurun_out/libs_ab.txt
out=gpurun_out/libs_ab.txt
: > $out
for rep in 1 2; do
  for lib in "$@"; do
    MIRT_LIB=$PWD/$lib timeout -k 10 300 python scripts/ab.py --reps 2 ${CASES:-random final256 million million4} >> $out 2>&1 || exit 1
  done
done
cat $out
