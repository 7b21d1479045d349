"""Static instruction histogram of one compiled kernel by the source function each instruction was inlined from.

    cd /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -g -save-temps \
        -c <repo>/ray_tracing_amd/csrc/kernels.hip -o k.o
    llvm-dwarfdump --debug-info kernels-hip-amdgcn-amd-amdhsa-gfx950.out > dwarf.txt
    llvm-objdump -d --no-show-raw-insn kernels-hip-amdgcn-amd-amdhsa-gfx950.out > dis.txt
    python scripts/isa_histogram.py dwarf.txt dis.txt _ZN4mirt19render_queue_kernelILj5643ELb1ELb1EEEvNS_10RenderArgsE

(-g does not change the generated code: the instruction counts equal the production build's.) Every instruction of
the kernel is given the chain of inlined calls that contains its address (DW_TAG_inlined_subroutine records), and
is booked under the innermost function of that chain that appears in PHASES — so a dot product
inside the light pdf inside scatter counts as "lights_pdf", not as "dot". Instructions outside every listed function
are the kernel's own body. Classes: VALU, SALU, scalar memory, vector memory, LDS, branches, waits.
The counts are static (code size per phase, every branch of it once); what runs per segment depends on the scene."""
import bisect
import collections
import re
import sys

# phase functions, matched against the demangled-ish name inside the abstract origin (order = priority, outermost first)
PHASES = ["copy_to_lds", "take_items", "flush_pixel", "next_pixel", "medium_hit", "rect_pair_hit", "rect_rec_hit", "face_pair_hit", "sphere_hit_c", "list_scan", "world_lists",
          "walk_node", "walk_leaf", "walk_some", "walk_run", "finalize_hit", "emitted_radiance", "philox4x32_10", "camera_ray",
          "lights_pdf", "light_draw", "axis_draw_direction", "cosine_draw", "texture_value", "scatter"]
# lambdas of the kernel body: _ZZN...ENKUlbE_clEb = next_pixel(bool), ...ENKUlvE_clEv = flush_pixel()
LAMBDAS = {"ENKUlbE_clEb": "next_pixel", "ENKUlvE_clEv": "flush_pixel"}


def classify(mn):
    if mn.startswith("v_"):
        return "VALU"
    if mn.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
        return "SMEM"
    if mn.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if mn.startswith("s_"):
        return "SALU"
    if mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if mn.startswith("ds_"):
        return "LDS"
    return "other"


def phase_name(origin):
    for tag, name in LAMBDAS.items():
        if origin.endswith(tag):
            return name
    for p in PHASES:
        if re.search(r"\d+" + re.escape(p) + r"(?![a-z_0-9])", origin) or origin == p:
            return p
    return None


def parse_inlines(dwarf_path, kernel):
    """[(depth, name, [(lo, hi), ...])] of the inlined subroutines inside the kernel's subprogram."""
    out = []
    in_kernel, kernel_depth = False, None
    cur = None
    die = re.compile(r"^0x[0-9a-f]+:(\s+)(DW_TAG_\w+|NULL)")
    pending_sub = None
    for raw in open(dwarf_path):
        m = die.match(raw)
        if m:
            depth, tag = len(m.group(1)), m.group(2)
            if cur:
                out.append(cur)
                cur = None
            if in_kernel and depth <= kernel_depth and tag != "NULL":
                in_kernel = False
            if tag == "DW_TAG_subprogram":
                pending_sub = depth
            else:
                if tag != "NULL" and pending_sub is not None and depth <= pending_sub:
                    pending_sub = None
            if in_kernel and tag == "DW_TAG_inlined_subroutine":
                cur = [depth, None, []]
            continue
        s = raw.strip()
        if pending_sub is not None and s.startswith("DW_AT_linkage_name") and f'"{kernel}"' in s:
            in_kernel, kernel_depth = True, pending_sub
        if cur is None:
            continue
        if s.startswith("DW_AT_abstract_origin"):
            mm = re.search(r'"([^"]*)"', s)
            cur[1] = mm.group(1) if mm else "?"
        elif s.startswith("DW_AT_low_pc"):
            cur[2].append([int(re.search(r"0x([0-9a-f]+)", s).group(1), 16), None])
        elif s.startswith("DW_AT_high_pc") and cur[2] and cur[2][-1][1] is None:
            cur[2][-1][1] = int(re.search(r"0x([0-9a-f]+)", s).group(1), 16)
        else:
            mm = re.match(r"\[?\s*\[?0x([0-9a-f]+),\s*0x([0-9a-f]+)\)", s.replace("DW_AT_ranges", "").strip(" \t(") )
            if mm:
                cur[2].append([int(mm.group(1), 16), int(mm.group(2), 16)])
    if cur:
        out.append(cur)
    return [(d, n, [(lo, hi) for lo, hi in r if hi is not None]) for d, n, r in out if n]


def main():
    dwarf, dis, kernel = sys.argv[1:4]
    inl = parse_inlines(dwarf, kernel)
    # per phase-named inline record: intervals with depth
    spans = []
    for depth, name, ranges in inl:
        p = phase_name(name)
        if p:
            for lo, hi in ranges:
                spans.append((lo, hi, depth, p))
    spans.sort()
    starts = [s[0] for s in spans]
    counts = collections.defaultdict(collections.Counter)
    inside = False
    for raw in open(dis):
        if raw.startswith("0") and raw.rstrip().endswith(">:"):
            inside = f"<{kernel}>:" in raw
            continue
        if not inside:
            continue
        m = re.match(r"\s+(\S+).*//\s*([0-9A-Fa-f]+):", raw)
        if not m:
            continue
        mn, addr = m.group(1), int(m.group(2), 16)
        best = None
        i = bisect.bisect_right(starts, addr)
        for lo, hi, depth, p in spans[max(0, i - 4000):i]:
            if lo <= addr < hi and (best is None or depth > best[0]):
                best = (depth, p)
        counts[best[1] if best else "(kernel body)"][classify(mn)] += 1
    cols = ["VALU", "SALU", "SMEM", "VMEM", "LDS", "branch", "wait", "other"]
    print(f"# {kernel}")
    print(f"{'phase (outermost listed function)':36s} " + " ".join(f"{c:>6s}" for c in cols) + "   total")
    tot = collections.Counter()
    for region, c in sorted(counts.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{region:36s} " + " ".join(f"{c[k]:6d}" for k in cols) + f"  {sum(c.values()):6d}")
        tot.update(c)
    print(f"{'all':36s} " + " ".join(f"{tot[k]:6d}" for k in cols) + f"  {sum(tot.values()):6d}")


if __name__ == "__main__":
    main()
