#!/bin/bash
# usage: scripts/pmc_ta.sh <scene> <spp> — how busy the vector-memory address / L1 / data-return units are
SCENE=${1:-final}; SPP=${2:-16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ta_$SCENE
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/ta -- python3 $GRAFT_REPO_ROOT/scripts/prof_render.py $SCENE $SPP > $OUT/ta.log 2>&1
echo "rc=$?"; tail -1 $OUT/ta.log
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
