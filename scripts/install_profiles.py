"""Copies what scripts/r3_profiles.sh left under gpurun_out/ into profiles/<round>/ (kernel stats, counter summary, the
bench line under rocprofv3 and the plain one, per configuration) and refreshes profiles/hbm_traffic.json, the table
bench.py reads its traffic / lane / issue figures from. Registers and scratch come from a device assembly listing
(hipcc -S --cuda-device-only of kernels.hip).  usage: python scripts/install_profiles.py r3 /tmp/isa/kernels.s"""
import json
import re
import shutil
import sys

ROUND, ASM = sys.argv[1], sys.argv[2]
KEYS = {2: "cornell 1920x1080 1024spp depth50", 3: "final 1920x1080 256spp depth50", 5: "random 3840x2160 1spp depth10",
        1: "random 400x225 10spp depth50"}

# registers / scratch of every kernel in the listing
meta, name = {}, None
for line in open(ASM):
    s = line.strip()
    if s.startswith(".name:"):
        name = s.split()[-1]
        meta[name] = {}
    elif name and re.match(r"\.(vgpr_count|private_segment_fixed_size):", s):
        k, v = s.split(":")
        meta[name][k.strip(".")] = int(v)

table = json.load(open("profiles/hbm_traffic.json"))
for c, key in KEYS.items():
    src = f"gpurun_out/prof_c{c}"
    dst = f"profiles/{ROUND}/config{c}_megakernel"
    shutil.copy(f"{src}/kernel_stats.csv", dst + "_kernel_stats.csv")
    shutil.copy(f"{src}/pmc_summary.txt", dst + "_pmc.txt")
    shutil.copy(f"{src}/bench_under_stats.json", dst + "_bench_under_rocprof.json")
    open(dst + "_bench.json", "w").write(open(f"gpurun_out/bench_c{c}.json").read().strip().splitlines()[-1] + "\n")
    summary = open(f"{src}/pmc_summary.txt").read()
    num = lambda pat: float(re.search(pat, summary).group(1))
    kernel = re.search(r'"void mirt::(render_queue_kernel<[^>]*>)', open(f"{src}/kernel_stats.csv").read()).group(1)
    args = re.findall(r"(\d+)u|(true|false)", kernel)
    mangled = "_ZN4mirt19render_queue_kernelILj%sE" % args[0][0] + "".join("Lb%dE" % (a[1] == "true") for a in args[1:]) + "EEvNS_10RenderArgsE"
    e = table[key]
    e["bytes_per_launch"] = int(num(r"render HBM bytes per dispatch = .* = (\d+)"))
    e["lanes_per_valu"] = round(num(r"render lanes active per VALU instruction = .* = ([\d.]+)"), 4)
    e["valu_issue_clk"] = round(num(r"render clocks per VALU instruction and SIMD = .* = ([\d.]+)"), 3)
    valu, salu, disp = num(r"render SQ_INSTS_VALU sum ([\d.e+]+)"), num(r"render SQ_INSTS_SALU sum ([\d.e+]+)"), num(r"render SQ_INSTS_VALU sum [\d.e+]+ dispatches (\d+)")
    e["valu_per_launch"], e["salu_per_launch"] = valu / disp, salu / disp
    e["vgpr"], e["scratch_bytes"] = meta[mangled]["vgpr_count"], meta[mangled]["private_segment_fixed_size"]
    e["kernel"] = kernel
    e["profile"] = f"profiles/{ROUND}/config{c}_megakernel_pmc.txt"
    print(key, {k: e[k] for k in ("lanes_per_valu", "valu_issue_clk", "vgpr", "scratch_bytes", "valu_per_launch", "salu_per_launch", "kernel")})
json.dump(table, open("profiles/hbm_traffic.json", "w"), indent=1)
