#!/bin/bash
CUT=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cut_$CUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/prof_final_cut.py $CUT > $OUT/sq.log 2>&1
echo "rc=$?"; tail -1 $OUT/sq.log
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
