#!/bin/bash
# usage: scripts/pmc.sh <tag> <scene> <spp>   — separate rocprofv3 --pmc passes (<= 8 SQ counters each)
set -e
TAG=$1; SCENE=${2:-cornell}; SPP=${3:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/scripts/prof_render.py $SCENE $SPP > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
pass c GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] or "wavefront" in r["Kernel_Name"] or "extend" in r["Kernel_Name"] or "shade" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
