"""Benchmark of the hot path: Msamples/s on the 1920x1080 Cornell box, 1024 spp, depth 50.

    python bench.py --gpus N --steps K --warmup W

A step is one full pass of the per-pixel sample loop over one frame (W*H*spp primary paths) followed by
the RGBA8 resolve, with scene and camera already resident on the GPU. With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank renders the whole frame for its own range of `spp`
samples (weak scaling: N*spp samples per pixel in total), the float4 accumulation buffers are summed onto
rank 0 with one RCCL reduce, and rank 0 resolves. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SEGMENT = 400.0       # algorithmic FLOP of one Cornell path segment (12 rect tests x 12 + sphere 30 +
                               # instance transform 15 + diffuse shade with the 2-light mixture pdf 200)
# MI355X FP32 vector peak (spec, packed v_pk_fma_f32): the `peak` of the roofline object. Plain v_fma_f32 code was
# measured at 111.8 TFLOP/s (scripts/calib/valu_count.hip: a wave64 v_fma_f32 per 2.8 cycles and SIMD with 8
# waves resident); the render kernel is scalar-per-lane code except for the rect pairs, so the fraction of that
# measured rate is reported beside `frac`.
PEAK_FP32_VALU_TFLOPS = 157.3
UNPACKED_FP32_TFLOPS = 111.8
# HBM bytes of one launch of the render kernel on the default workload, from rocprofv3 PMC passes
# (profiles/r1/cornell_1080p_1024spp_v4_hbm_traffic_pmc.txt): 2 x FETCH_SIZE (gfx950 counts half of a wide read)
# + WRITE_SIZE, in KiB. The persistent kernel writes one float4 per pixel and sample chunk (16 chunks x 33.2 MB)
# and reads next to nothing; the pass that adds the chunks (reduce_chunks_kernel, 0.1 ms) reads them back once.
MEASURED_TRAFFIC_BYTES = {("cornell", 1920, 1080, 1024): (2 * 6922.375 + 544951.65625) * 1024}

def cpu_baseline(scene, width, height, depth, target_seconds=12.0):
    """The oracle (CPU restatement of the reference renderer) on this host's cores, on a bounded sample of
    the same workload: the same frame at a reduced sample count."""
    import oracle
    threads = min(os.cpu_count() or 1, 16)
    r = oracle.OracleRenderer(scene, width, height)
    t0 = time.time()
    r.render_sums(1, depth, seed=0, threads=threads)
    dt1 = time.time() - t0
    spp = max(1, min(16, int(target_seconds / max(dt1, 1e-3))))
    t0 = time.time()
    r.render_sums(spp, depth, seed=1, threads=threads)
    dt = time.time() - t0
    return {"value": width * height * spp / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{scene} {width}x{height}, {spp} spp, depth {depth}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the control flow)")
    args = ap.parse_args()

    import torch
    import ray_tracing_amd as rt
    from ray_tracing_amd.distributed import reduce_accum, weak_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    W, H, spp = args.width, args.height, args.spp
    ctx = rt.Context(local_rank)
    scene = rt.select_scene(args.scene)
    ctx.scene_upload(scene)
    ctx.camera_set(rt.Camera.select_camera(W / H, args.scene))
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)     # caller-owned, reducible by RCCL
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()   # the fills above run on torch's stream, the renderer on its own

    def step():
        ctx.frame_begin(W, H, accum.data_ptr())
        begin, count = weak_range(spp, rank)
        ctx.render(count, args.depth, spp_begin=begin, seed=0, jitter=True)
        if world > 1:
            ctx.synchronize()                       # the render runs on the library's own stream
            reduce_accum(accum, dst=0)              # RCCL sum of the float4 buffers onto rank 0
            torch.cuda.synchronize()
        if rank == 0:
            ctx.resolve_rgba8_device(frame.data_ptr(), frame.numel())
        ctx.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, segments, launches = 0.0, 0, 0
    for _ in range(args.steps):
        step()
        st = ctx.stats()                            # HIP-event time of this step's render launches
        kernel_ms += st["kernel_ms"]
        segments += st["segments"]
        launches += st["kernel_launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples_per_step = W * H * spp * world
        value = samples_per_step * args.steps / elapsed / 1e6
        achieved = FLOP_PER_SEGMENT * segments / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        out = {
            "metric": "Msamples/sec at 1920x1080 Cornell box; % of FP32-VALU/HBM roofline",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {spp} spp per GPU, max depth {args.depth}, light+cosine mixture pdf",
                       "samples_per_step": samples_per_step, "parallelism": f"spp-range x{world}" + (" + RCCL reduce" if world > 1 else "")},
            # compute-bound, not HBM-bound: of the two roofline categories this is the arithmetic one. A path tracer
            # has no matrix products, so the arithmetic runs on the vector ALUs; for f32 the vector peak and the
            # dense MFMA peak of the MI355X are the same 157.3 TFLOP/s.
            "roofline": {"bound": "mfma", "bound_note": "f32 vector-ALU issue bound (no matrix ops in this path); f32 VALU peak = f32 MFMA peak",
                         "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "frac_of_unpacked_fma_rate": achieved / UNPACKED_FP32_TFLOPS,
                         "traffic": MEASURED_TRAFFIC_BYTES.get((args.scene, W, H, spp)), "traffic_unit": "bytes per launch (PMC, separate run)",
                         "kernel": "render_queue_kernel", "avg_launch_ms": kernel_ms / max(launches, 1),
                         "segments_per_sample": segments / (W * H * spp * args.steps),
                         "gsegments_per_s": segments / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                         "flop_per_segment": FLOP_PER_SEGMENT},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scene, W, H, args.depth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
