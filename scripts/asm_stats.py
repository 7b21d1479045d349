"""Static figures of the kernels in a device assembly listing, for comparing builds on a machine without a GPU.

    hipcc --offload-arch=gfx950 $CXXFLAGS --cuda-device-only -S ray_tracing_amd/csrc/kernels.hip -o /tmp/isa/kernels.s
    python scripts/asm_stats.py /tmp/isa/kernels.s [substring of the kernel names to list]

Per kernel: registers, spills, scratch bytes, and instruction counts by class (all vector ALU, moves, fused
multiply-adds, separate multiplies / adds, packed forms, lane reads / writes that spill scalar registers, scratch
accesses). The counts are of the code, every branch once; what runs per segment depends on the scene."""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "render_queue_kernel"
    body = collections.defaultdict(collections.Counter)
    meta = {}
    cur = None
    name = None
    for raw in open(path):
        s = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m and not raw.startswith("\t"):
            cur = m.group(1)
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            cur = None
            continue
        if s.startswith(".name:"):
            name = s.split()[-1]
            meta.setdefault(name, {})
        elif name and re.match(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):", s):
            k, v = s.split(":")
            meta[name][k.strip(".")] = int(v)
        if cur is None or not s or s.startswith((".", ";", "//")):
            continue
        mn = s.split()[0]
        c = body[cur]
        if mn.startswith("v_"):
            c["valu"] += 1
            if mn.startswith("v_mov_b32") or mn.startswith("v_accvgpr") or mn.startswith("v_mov_b64"):
                c["mov"] += 1
            elif mn.startswith(("v_fma_f32", "v_fmac_f32", "v_pk_fma_f32")):
                c["fma"] += 1
            elif mn.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32")):
                c["mul_add"] += 1
            if mn.startswith("v_pk_"):
                c["pk"] += 1
            if mn.startswith(("v_readlane", "v_writelane")):
                c["lane_rw"] += 1
            if mn.startswith(("v_cndmask",)):
                c["cndmask"] += 1
            if mn.startswith(("v_cmp",)):
                c["cmp"] += 1
        elif mn.startswith("scratch_"):
            c["scratch"] += 1
        elif mn.startswith("s_"):
            c["salu"] += 1
    cols = ["valu", "mov", "fma", "mul_add", "pk", "cmp", "cndmask", "lane_rw", "scratch", "salu"]
    print(f"{'kernel':58s} vgpr sgsp vgsp scr " + " ".join(f"{c:>7s}" for c in cols))
    for k in body:
        if want not in k:
            continue
        md = meta.get(k, {})
        print(f"{k[:58]:58s} {md.get('vgpr_count', 0):4d} {md.get('sgpr_spill_count', 0):4d} {md.get('vgpr_spill_count', 0):4d} "
              f"{md.get('private_segment_fixed_size', 0):3d} " + " ".join(f"{body[k][c]:7d}" for c in cols))


if __name__ == "__main__":
    main()
