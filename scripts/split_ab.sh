#!/bin/bash
# GPU box: sample chunks per pixel and launch (MIRT_SPLIT) on the bench configurations. Output: gpurun_out/split_ab.txt
out=gpurun_out/split_ab.txt
: > $out
for split in 8 16 32 64; do
  echo "# MIRT_SPLIT=$split" >> $out
  MIRT_SPLIT=$split timeout -k 10 200 python scripts/ab.py --reps 2 cornell1024 final256 random million4 >> $out 2>&1
done
cat $out
