#!/bin/bash
# usage: scripts/pmc_traffic.sh <tag> <scene> <spp> — HBM traffic of the render kernel: FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 --pmc passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2).
TAG=$1; SCENE=${2:-cornell}; SPP=${3:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/scripts/prof_render.py $SCENE $SPP > $OUT/$c.log 2>&1
  echo "pass $c rc=$?"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] or "reduce_chunks" in r["Kernel_Name"]:
            key = r["Counter_Name"] + " " + r["Kernel_Name"].split("<")[0].split("(")[0][-24:]; tot[key] += float(r["Counter_Value"]); n[key] += 1
for k in sorted(tot): print(k, "sum", tot[k], "dispatch rows", n[k])
PY
