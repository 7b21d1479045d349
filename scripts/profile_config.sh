#!/bin/bash
# usage (on the GPU box, from the repository root): scripts/profile_config.sh <tag> <config> [steps] [extra bench args...]
# rocprofv3 evidence for one bench.py configuration: a --kernel-trace --stats pass of the bench command itself, then
# separate --pmc passes (instruction mix and lane activity, HBM traffic) of the same command with few steps.
# Everything lands under gpurun_out/prof_<tag>/; the summaries that are kept get copied to profiles/ by hand.
TAG=$1; CFG=$2; STEPS=${3:-3}; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-count-pass $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_stats.json 2> $OUT/stats.log || { echo "stats pass failed"; exit 1; }
pass() {
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.json 2> $OUT/$name.log || { echo "pass $name failed"; exit 1; }
  echo "pass $name ok"
}
pass pmc_a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass pmc_c GRBM_GUI_ACTIVE
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
tot = collections.defaultdict(float); rows = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        short = "render" if "render_" in k else ("reduce_chunks" if "reduce_chunks" in k else ("wf" if "wf" in k else None))
        if short:
            tot[(short, r["Counter_Name"])] += float(r["Counter_Value"]); rows[(short, r["Counter_Name"])] += 1
with open(out + "/pmc_summary.txt", "w") as o:
    o.write("# bench.py --config $CFG --steps $STEPS --warmup 1 (sums over all dispatches of the run, warm-up included)\n")
    for k in sorted(tot):
        o.write(f"{k[0]} {k[1]} sum {tot[k]} dispatches {rows[k]}\n")
    for kern in ("render", "wf"):
        v, t = tot.get((kern, "SQ_INSTS_VALU")), tot.get((kern, "SQ_THREAD_CYCLES_VALU"))
        a = tot.get((kern, "SQ_ACTIVE_INST_VALU"))
        if v and t:
            o.write(f"{kern} lanes active per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU) = {t / (64 * v):.4f}\n")
        g, ng = tot.get((kern, "GRBM_GUI_ACTIVE")), rows.get((kern, "GRBM_GUI_ACTIVE"))
        nv = rows.get((kern, "SQ_INSTS_VALU"))
        if v and g and ng and nv:
            # clocks per vector instruction and SIMD: device clocks of a dispatch x 1024 SIMDs / its wave-level vector
            # instructions. The counter comes back summed over the 8 XCDs (a 127.6 ms launch reads 2.435e9 = 8 x 127.6 ms
            # x 2.385 GHz), hence the division by 8.
            o.write(f"{kern} clocks per VALU instruction and SIMD = (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs / SQ_INSTS_VALU (per dispatch) = {(g / ng / 8) * 1024 / (v / nv):.3f}\n")
            o.write(f"{kern} VALU instructions per dispatch = {v / nv:.4g}\n")
        f_, w_ = tot.get((kern, "FETCH_SIZE")), tot.get((kern, "WRITE_SIZE"))
        n = rows.get((kern, "FETCH_SIZE"))
        if f_ is not None and w_ is not None and n:
            # gfx950: FETCH_SIZE counts half of the bytes of a wide read (in KiB): traffic = 2 * FETCH_SIZE + WRITE_SIZE
            o.write(f"{kern} HBM bytes per dispatch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / dispatches = {(2 * f_ + w_) * 1024 / n:.0f}\n")
print(open(out + "/pmc_summary.txt").read())
PY
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
head -5 $OUT/kernel_stats.csv
