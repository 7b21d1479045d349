#!/bin/bash
# usage: scripts/pmc_kernels.sh <tag> <python script and args...> — one rocprofv3 --pmc pass (8 SQ counters) of a
# command, summed per kernel name. Output: gpurun_out/pmck_<tag>/summary.txt
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/a -- python3 "$@" > $OUT/a.log 2>&1 || { echo pass failed; tail -5 $OUT/a.log; exit 1; }
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mirt::", "").replace("mirt::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/summary.txt", "w") as o:
    for k, c in sorted(tot.items()):
        if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
        line = (f"{k}: VALU {c['SQ_INSTS_VALU']:.4g} SALU {c['SQ_INSTS_SALU']:.4g} lanes/VALU {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_INSTS_VALU']):.3f} "
                f"wave_cycles {c['SQ_WAVE_CYCLES']:.4g} wait_any/wave_cycles {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f} active_valu/wave_cycles {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f} busy_cycles {c['SQ_BUSY_CYCLES']:.4g} waves {c['SQ_WAVES']:.0f}")
        o.write(line + "\n")
print(open("$OUT/summary.txt").read())
PY
