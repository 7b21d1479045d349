#!/bin/bash
# GPU box: one rocprofv3 --pmc pass with the vector-L1 path's counters (2 TA, 4 TCP, 2 GRBM) of a bench.py configuration.
# usage: scripts/pmc_l1.sh <config> [steps] ; output: gpurun_out/pmc_l1_c<config>.txt
CFG=$1; STEPS=${2:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l1_c$CFG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-count-pass > $OUT/bench.json 2> $OUT/log.txt || { echo "pass failed"; tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("$GRAFT_REPO_ROOT/gpurun_out/pmc_l1_c$CFG.txt", "w") as o:
    o.write("# bench.py --config $CFG --steps $STEPS --warmup 1: render kernel, per dispatch (sums over the device's units)\n")
    for k in sorted(tot):
        o.write(f"{k} {tot[k] / n[k]:.6g} (dispatches {n[k]})\n")
    g = tot.get("GRBM_GUI_ACTIVE", 0) / max(n.get("GRBM_GUI_ACTIVE", 1), 1) / 8   # summed over the 8 XCDs
    if g:
        o.write(f"device clocks per dispatch (GRBM_GUI_ACTIVE / 8 XCDs) {g:.6g}\n")
        for k in ("TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"):
            if k in tot:
                o.write(f"{k} per unit (256 CUs) and device clock {tot[k] / n[k] / 256 / g:.3f}\n")
        if tot.get("TCP_TCC_READ_REQ_sum"):
            o.write(f"mean latency of an L1 read request to L2 (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) {tot['TCP_TCC_READ_REQ_LATENCY_sum'] / tot['TCP_TCC_READ_REQ_sum']:.1f} clocks\n")
print(open("$GRAFT_REPO_ROOT/gpurun_out/pmc_l1_c$CFG.txt").read())
PY
