"""Image mode of the reference (src/cli.rs:21-38) on the MI355X backend.

    python -m ray_tracing_amd.cli -s 1024 -d 1 --scene cornell

takes the reference's flags with the reference's defaults (-m/--max-depth 50, -s/--sample-count 100,
-d/--down-scale 10, --scene random) and writes screenshot.png. Additions: --width/--height (instead of
1920/down_scale x 1080/down_scale), --seed, --scene-seed, --output, --device, --pass-spp (render in passes
and report progress), --checkpoint (save the accumulation buffer after every pass; an existing file of the
same shape, scene and seed is resumed). Under `python -m torch.distributed.run --nproc-per-node N` every
rank renders its share of the samples on its own GPU and rank 0 writes the image.
"""
import argparse
import os
import sys
import time

import numpy as np

import ray_tracing_amd as rt
from ray_tracing_amd.distributed import join_comm, sample_range

WIDTH, HEIGHT = 1920, 1080  # lib.rs:42-43


def parse(argv=None):
    p = argparse.ArgumentParser(prog="ray_tracing_amd.cli", description="Get ray tracing parameters")  # cli.rs:7
    p.add_argument("-m", "--max-depth", type=int, default=50)       # cli.rs:11
    p.add_argument("-s", "--sample-count", type=int, default=100)   # cli.rs:13
    p.add_argument("-d", "--down-scale", type=int, default=10)      # cli.rs:15
    p.add_argument("--scene", default="random")                     # cli.rs:17
    p.add_argument("--width", type=int, default=0)
    p.add_argument("--height", type=int, default=0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--scene-seed", type=int, default=1)
    p.add_argument("--output", default="screenshot.png")            # cli.rs:32
    p.add_argument("--device", type=int, default=None)
    p.add_argument("--pass-spp", type=int, default=0, help="samples per pixel per pass (0 = one pass)")
    p.add_argument("--checkpoint", default="", help=".npz file holding the accumulation buffer between passes")
    a = p.parse_args(argv)
    if a.down_scale <= 0 or a.sample_count < 0 or a.max_depth < 0:
        p.error("down-scale must be positive, sample-count and max-depth non-negative")
    return a


def _load_checkpoint(path, shape, tag):
    if not path or not os.path.exists(path):
        return None, 0
    with np.load(path, allow_pickle=False) as z:
        if str(z["tag"]) != tag or z["accum"].shape != shape:
            raise SystemExit(f"{path} was written for a different render ({z['tag']}); remove it or change --checkpoint")
        return z["accum"].astype(np.float32), int(z["spp_done"])


def image_mode(argv=None):
    a = parse(argv)
    width = a.width or WIDTH // a.down_scale      # cli.rs:24
    height = a.height or HEIGHT // a.down_scale
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0"))

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")      # control plane only; the buffers travel over the library's RCCL reduce
        if a.checkpoint:
            raise SystemExit("--checkpoint saves one accumulation buffer and is for single-GPU runs; drop it or run one process")

    scene = rt.select_scene(a.scene, scene_seed=a.scene_seed)
    camera = rt.Camera.select_camera(width / height, a.scene)   # cli.rs:25
    ctx = rt.Context(device)
    ctx.scene_upload(scene)
    ctx.camera_set(camera)

    if world > 1:
        join_comm(ctx, rank, world)
    ctx.frame_begin(width, height)

    jitter = a.sample_count != 1   # renderer.rs:175
    tag = f"{a.scene}/{a.scene_seed}/{a.seed}/{a.max_depth}/{world}/{width}x{height}/{'jitter' if jitter else 'centre'}"
    done = 0
    if world == 1:
        saved, done = _load_checkpoint(a.checkpoint, (height, width, 4), tag)
        if saved is not None:
            ctx.write_accum(saved)
            print(f"resumed {a.checkpoint}: {done} of {a.sample_count} samples per pixel", file=sys.stderr)

    begin, count = sample_range(a.sample_count, world, rank)
    step = a.pass_spp if a.pass_spp > 0 else max(count, 1)
    t0 = time.time()
    local_done = min(done, count)
    while local_done < count:
        n = min(step, count - local_done)
        ctx.render(n, a.max_depth, spp_begin=begin + local_done, seed=a.seed, jitter=jitter)
        ctx.synchronize()
        local_done += n
        if rank == 0 and a.pass_spp > 0:
            print(f"{local_done}/{count} spp  {time.time() - t0:.2f} s", file=sys.stderr)
        if world == 1 and a.checkpoint:
            tmp = a.checkpoint + ".tmp.npz"
            np.savez(tmp, accum=ctx.read_accum(), spp_done=np.int64(local_done), tag=np.str_(tag))
            os.replace(tmp, a.checkpoint)
    ctx.synchronize()
    if world > 1:
        ctx.comm_reduce(0)                  # RCCL sum of the float4 buffers onto rank 0, behind the render on its stream
        ctx.synchronize()
    if rank == 0:
        frame = ctx.resolve_rgba8()
        dt = time.time() - t0
        rt.save_png(a.output, frame)   # cli.rs:31-38
        st = ctx.stats()
        total = width * height * max(a.sample_count - done, 0)
        print(f"{a.scene} {width}x{height} {a.sample_count} spp depth {a.max_depth} on {world} GPU(s): {dt:.3f} s, "
              f"{total / max(dt, 1e-9) / 1e6:.1f} Msamples/s, {st['segments'] / max(st['samples'], 1):.2f} segments/sample "
              f"-> {a.output}", file=sys.stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(image_mode())
