#!/bin/bash
# GPU box: the shared-walk thresholds again (MIRT_WALK_BUDGET / MIRT_WALK_LANES / MIRT_NODE_LANES) on the hierarchy scenes.
out=gpurun_out/walk_resweep.txt
: > $out
run() { echo "# budget=$1 lanes=$2 node_lanes=$3" >> $out; MIRT_WALK_BUDGET=$1 MIRT_WALK_LANES=$2 MIRT_NODE_LANES=$3 timeout -k 10 120 python scripts/ab.py --reps 2 $4 >> $out 2>&1; }
for b in ${BUDGETS:-64 96 128}; do for l in ${LANES:-4 6 8 12}; do for n in ${NODES:-6 8 12}; do run $b $l $n final256; done; done; done
[ -n "$FINAL_ONLY" ] || for b in 32 48 64; do for l in 16 24 32; do for n in 12 16 24; do run $b $l $n random; run $b $l $n million; done; done; done
python3 - <<'PY'
import re
best={}
cur=None
for l in open("gpurun_out/walk_resweep.txt"):
    if l.startswith("#"): cur=l.strip(); continue
    m=re.search(r"queue\s+(\S+).*?(\d+) Msamples",l)
    if m:
        best.setdefault(m.group(1),[]).append((int(m.group(2)),cur))
for k,v in best.items():
    v.sort(reverse=True)
    print(k, v[:5], "worst", v[-1])
PY
