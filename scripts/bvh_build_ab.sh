#!/bin/bash
# GPU box: builder settings of the hierarchies on one box, twice each, interleaved. A setting is
# bins:bins_of_large_ranges:large_from (MIRT_BVH_BINS, MIRT_BVH_BINS_LARGE, MIRT_BVH_LARGE_FROM).
# usage: [CASES="..."] [SETTINGS="..."] bash scripts/bvh_build_ab.sh ; output: gpurun_out/bvh_build_ab.txt
out=gpurun_out/bvh_build_ab.txt
: > $out
for rep in 1 2; do
  for s in ${SETTINGS:-16:0:4096 16:128:4096}; do
    IFS=: read bins large from <<< "$s"
    echo "# bins=$bins bins_large=$large large_from=$from" >> $out
    MIRT_BVH_BINS=$bins MIRT_BVH_BINS_LARGE=$large MIRT_BVH_LARGE_FROM=$from timeout -k 10 300 python scripts/ab.py --reps 2 $AB_FLAGS ${CASES:-random final256 million} >> $out 2>&1 || exit 1
  done
done
cat $out
