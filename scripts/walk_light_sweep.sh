#!/bin/bash
# GPU box: the light-scene walk thresholds over chosen values. usage: [B="48 64"] [L="8 12 16"] [N="12 16"] [CASES="random million"] bash scripts/walk_light_sweep.sh
out=gpurun_out/walk_light_sweep.txt
: > $out
for b in ${B:-48 64}; do for l in ${L:-8 12 16}; do for n in ${N:-12 16}; do
  echo "# budget=$b lanes=$l node_lanes=$n" >> $out
  MIRT_WALK_BUDGET=$b MIRT_WALK_LANES=$l MIRT_NODE_LANES=$n timeout -k 10 120 python scripts/ab.py --reps 2 ${CASES:-random million} >> $out 2>&1 || exit 1
done; done; done
grep "#\|Msamples" $out | awk '/^#/{h=$2" "$3" "$4; next} {print h, $3, $(NF-1)}'
