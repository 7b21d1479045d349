#!/bin/bash
# usage: scripts/pmc_cache.sh <tag> <scene> <spp>  — vector L1 / L2 hit counters of the render kernel (own rocprofv3 passes)
TAG=$1; SCENE=${2:-final}; SPP=${3:-16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/scripts/prof_render.py $SCENE $SPP > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass l2 TCC_HIT_sum TCC_MISS_sum
pass sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
