"""Benchmark of the hot path: Msamples/s of the per-pixel sample loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4|5] [--scaling weak|strong]

Default workload (config 2, the one the metric is quoted on): Cornell box 1920x1080, 1024 spp, depth 50.
A step is one full pass of the sample loop over one frame (W*H*spp primary paths) followed by the RGBA8
resolve, with scene and camera already resident on the GPU.

With N > 1 there is one process per GPU. Started by torch.distributed.run (RANK / WORLD_SIZE in the
environment) this file is a rank; started plainly with --gpus N it launches the N ranks itself, as child
processes, before anything in this process touches a GPU. Every rank renders the whole frame for its own range
of samples, the float4 accumulation buffers are summed onto rank 0 by the library's RCCL reduce (ncclReduce over
xGMI, on the render stream), and rank 0 resolves. torch.distributed (gloo) is the control plane only: it carries
the RCCL id and the barriers. Rank 0 prints ONE JSON line; for N > 1 it carries `per_rank`: the render time of
every rank (HIP events), the slowest and fastest, and the reduce time on the root, so a poor scaling curve can be
read (unbalanced renders against a slow exchange). The run fails if the communicator does not span all N ranks.

Two splits:
  weak (default, so that N = 1 is the headline number):  python bench.py --gpus N            `spp` samples per rank,
                                                         N * spp per pixel in total
  strong (the 8-GPU configuration of the metric):        python bench.py --gpus N --config 4  16 384 spp of a
                                                         3840x2160 Cornell frame split over the N ranks
The reduce over two or more GPUs has not run on hardware yet (the boxes this was built on have one GPU; on one
GPU RCCL refuses a second rank): it is covered by world-size 2, 3 and 8 gloo tests of the same split on the CPU and
by the library's RCCL path at world size 1.

Other configurations of the metric's family (parity-tested, reported for their rooflines):
  1  random 400x225, 10 spp, depth 50            3  final 1920x1080, 256 spp, depth 50
  4  cornell 3840x2160, 16384 spp, strong split   5  random num=500 (1 M spheres) 3840x2160, 1 spp per pass, depth 10
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic FLOP per unit of work (fixed convention: add = mul = compare-select = 1, fma = 2, div = sqrt = 1):
# rect test 12 (geo.rs:198-204), sphere test 30 (geo.rs:90-105), box test 24 (aabb.rs:50-59), instance transform
# in + out 15, diffuse shade with the mixture pdf 200 (renderer.rs:207-221). A Cornell segment is 12 rects +
# 1 sphere + 1 transform + shade = 389, rounded to 400 by the convention the metric was defined with.
FLOP_RECT, FLOP_SPHERE, FLOP_BOX, FLOP_TRANSFORM, FLOP_SHADE = 12.0, 30.0, 24.0, 15.0, 200.0
FLOP_PER_CORNELL_SEGMENT = 400.0
# MI355X FP32 vector peak (packed v_pk_fma_f32): the `peak` of the roofline object. Plain v_fma_f32 code was
# measured at 111.8 TFLOP/s (scripts/calib/valu_count.hip); the fraction of that rate is reported beside `frac`.
PEAK_FP32_VALU_TFLOPS = 157.3
UNPACKED_FP32_TFLOPS = 111.8
# HBM bytes per launch of the render kernel, from rocprofv3 PMC passes of this same command (separate runs, as the
# counters demand); the file names the profile each figure comes from.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")

CONFIGS = {
    1: dict(scene="random", width=400, height=225, spp=10, depth=50),
    2: dict(scene="cornell", width=1920, height=1080, spp=1024, depth=50),
    3: dict(scene="final", width=1920, height=1080, spp=256, depth=50),
    4: dict(scene="cornell", width=3840, height=2160, spp=16384, depth=50, scaling="strong"),
    5: dict(scene="random", width=3840, height=2160, spp=1, depth=10, random_num=500),
}


def cpu_baseline(scene, width, height, depth, random_num, target_seconds=10.0):
    """The oracle (CPU restatement of the reference renderer, rows handed to threads as rayon does) on this host,
    on a bounded sample of the same workload: the same scene and depth on a frame of the same shape at a sample
    count (and, for the million-sphere scene, a size) that fits the time budget. Timed at two thread counts — every
    CPU the process may run on, and 16 — because a container may show more CPUs than it may use; the faster one is
    reported with its thread count."""
    import oracle
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    w, h = width, height
    while w * h > 1920 * 1080 or (random_num > 11 and w * h > 64 * 36):   # a linear scan of 10^6 spheres per ray
        w, h = max(w // 2, 2), max(h // 2, 2)
    r = oracle.OracleRenderer(scene, w, h, random_num=random_num)
    best = None
    for threads in sorted({min(visible, 256), min(visible, 16)}, reverse=True):
        t0 = time.time()
        r.render_sums(1, depth, seed=0, threads=threads)
        dt1 = time.time() - t0
        spp = max(1, min(16, int(target_seconds / max(dt1, 1e-3))))
        t0 = time.time()
        r.render_sums(spp, depth, seed=1, threads=threads)
        dt = time.time() - t0
        v = w * h * spp / dt / 1e6
        if best is None or v > best["value"]:
            best = {"value": v, "unit": "Msamples/s", "cores": threads, "kind": "port",
                    "sample": f"{scene} {w}x{h}, {spp} spp, depth {depth}, {dt:.1f} s on {threads} threads",
                    "cpus_visible": visible, "os_cpu_count": os.cpu_count()}
    return best


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """--gpus N without a launcher: start the N ranks as children of this process (which has not touched a GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--scene")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--depth", type=int)
    ap.add_argument("--random-num", type=int, dest="random_num")
    ap.add_argument("--scaling", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count-pass", action="store_true",
                    help="skip the extra render that tallies the hierarchy walks (profiling runs: every launch of the trace is then a timed step or the warm-up)")
    args = ap.parse_args()
    cfg = dict(scaling="weak", random_num=11)
    cfg.update(CONFIGS[args.config])
    for k in ("scene", "width", "height", "spp", "depth", "random_num", "scaling"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import torch  # before the library: one HIP runtime per process, whichever is loaded first serves both
    import ray_tracing_amd as rt
    from ray_tracing_amd.distributed import join_comm, sample_range, weak_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")     # control plane; the accumulation buffers travel over the library's RCCL reduce
    n_dev = max(torch.cuda.device_count(), 1)
    device = local_rank % n_dev

    W, H, spp, depth, scene_name = cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], cfg["scene"]
    ctx = rt.Context(device)
    scene = rt.select_scene(scene_name, random_num=cfg["random_num"])
    ctx.scene_upload(scene)
    ctx.camera_set(rt.Camera.select_camera(W / H, scene_name))
    if world > 1:
        join_comm(ctx, rank, world)
    begin, count = sample_range(spp, world, rank) if cfg["scaling"] == "strong" else weak_range(spp, rank)
    total_spp = spp if cfg["scaling"] == "strong" else spp * world
    jitter = total_spp != 1                 # renderer.rs:175

    def step():
        ctx.frame_begin(W, H)
        if count > 0:                       # (a strong split over more ranks than samples leaves some with none)
            ctx.render(count, depth, spp_begin=begin, seed=0, jitter=jitter)
        if world > 1:
            ctx.comm_reduce(0)              # RCCL sum of the float4 buffers onto rank 0, behind the render on its stream
        if rank == 0:
            ctx.resolve_rgba8_device(None, W * H * 4)
        ctx.synchronize()

    def barrier():
        ctx.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches, reduce_ms = 0.0, 0, 0.0
    tot = {k: 0 for k in ("segments", "bvh_node_steps", "bvh_box_tests", "bvh_sphere_tests", "bvh_rect_tests")}   # bvh_*: filled by the counting pass below
    st = {}
    for _ in range(args.steps):
        step()
        st = ctx.stats()                    # HIP-event time of this step's render launches, on the render stream
        kernel_ms += st["kernel_ms"]
        launches += st["kernel_launches"]
        reduce_ms += st["reduce_ms"]
        tot["segments"] += st["segments"]
        if st["device_faults"]:
            raise SystemExit(f"device faults {st['device_faults']}")
    barrier()
    elapsed = time.perf_counter() - t0
    # one more pass, outside the timed region, that also tallies what the hierarchy walks cost (nodes visited, leaf
    # primitives intersected): the per-segment work the roofline figure is priced with
    if rank == 0 and not args.no_count_pass and st.get("kernel_used") in (rt.KERNEL_QUEUE, rt.KERNEL_WAVEFRONT) and scene.desc.n_prims > 32:
        ctx.frame_begin(W, H)
        ctx.render(min(count, 16), depth, spp_begin=begin, seed=0, jitter=jitter, kernel=st["kernel_used"] | rt.KERNEL_COUNT_WALKS)
        ct = ctx.stats()
        for k in ("bvh_node_steps", "bvh_box_tests", "bvh_sphere_tests", "bvh_rect_tests"):
            tot[k] = ct[k] * tot["segments"] // max(ct["segments"], 1)      # scaled to the timed steps' segments
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = {"rank": rank, "device": device, "spp": count, "render_ms_per_step": kernel_ms / args.steps,
                "reduce_ms_per_step": reduce_ms / args.steps, "rccl_ranks": ctx.comm_ranks()}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        bad = [r for r in everyone if r["rccl_ranks"] != world]
        if bad:
            raise SystemExit(f"the RCCL communicator spans {bad[0]['rccl_ranks']} ranks on rank {bad[0]['rank']}, not {world}")
        renders = [r["render_ms_per_step"] for r in everyone]
        per_rank = {"render_ms_max": max(renders), "render_ms_min": min(renders), "reduce_ms_root": everyone[0]["reduce_ms_per_step"],
                    "note": "reduce_ms is measured on the root's stream and includes its wait for the slowest rank's render",
                    "ranks": everyone}

    if rank == 0:
        samples_per_step = W * H * total_spp
        value = samples_per_step * args.steps / elapsed / 1e6
        seg = max(tot["segments"], 1)
        n_box, n_sph, n_rect = tot["bvh_box_tests"] / seg, tot["bvh_sphere_tests"] / seg, tot["bvh_rect_tests"] / seg
        per_segment = {"boxes": n_box, "spheres": n_sph + st["list_spheres"], "rects": n_rect + st["list_rects"],
                       "transforms": st["list_transforms"]}
        if scene_name == "cornell":
            flop_per_segment = FLOP_PER_CORNELL_SEGMENT
        else:
            flop_per_segment = (FLOP_BOX * per_segment["boxes"] + FLOP_SPHERE * per_segment["spheres"] + FLOP_RECT * per_segment["rects"]
                                + FLOP_TRANSFORM * per_segment["transforms"] + FLOP_SHADE)
        achieved = flop_per_segment * tot["segments"] / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        traffic, traffic_src, prof = None, None, {}
        try:
            with open(TRAFFIC_FILE) as f:
                prof = json.load(f).get(f"{scene_name} {W}x{H} {count}spp depth{depth}") or {}
            if prof:
                traffic, traffic_src = prof["bytes_per_launch"], prof["profile"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "Msamples/sec at 1920x1080 Cornell box; % of FP32-VALU/HBM roofline",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": cfg["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_name} {W}x{H}, {spp} spp {'in total' if cfg['scaling'] == 'strong' else 'per GPU'}, max depth {depth}"
                                   + (", light+cosine mixture pdf" if scene_name == "cornell" else "")
                                   + (f", random_num {cfg['random_num']} ({scene.desc.n_prims} spheres)" if cfg["random_num"] != 11 else ""),
                       "config_id": args.config, "samples_per_step": samples_per_step,
                       "parallelism": f"spp-range x{world}" + (" + RCCL reduce" if world > 1 else ""), "rccl_ranks": ctx.comm_ranks()},
            # compute-bound on the vector ALUs (a path tracer has no matrix products: no MFMA), not HBM-bound
            "roofline": {"bound": "valu", "bound_note": "bound by the instructions a wave issues (vector and scalar alike: 100 more of either per loop iteration cost the same, "
                                                      "profiles/r3/probe_instruction_cost.txt); peak = packed f32 FMA rate of the vector ALUs",
                         "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "frac_of_unpacked_fma_rate": achieved / UNPACKED_FP32_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, separate run)", "traffic_profile": traffic_src,
                         # from the committed profile of this same command (rocprofv3 --pmc passes, code object notes)
                         "lanes_per_valu": prof.get("lanes_per_valu"), "valu_issue_clk": prof.get("valu_issue_clk"),
                         "vgpr": prof.get("vgpr"), "scratch_bytes": prof.get("scratch_bytes"),
                         "valu_per_launch": prof.get("valu_per_launch"), "salu_per_launch": prof.get("salu_per_launch"),
                         "kernel": "render_queue_kernel" if st.get("kernel_used") == rt.KERNEL_QUEUE else "render_fused_kernel",
                         "avg_launch_ms": kernel_ms / max(launches, 1),
                         "segments_per_sample": tot["segments"] / (W * H * count * args.steps),
                         "gsegments_per_s": tot["segments"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                         "flop_per_segment": flop_per_segment, "tests_per_segment": per_segment},
        }
        if per_rank:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_name, W, H, depth, cfg["random_num"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
