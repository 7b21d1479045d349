"""Throughput of the hierarchy scenes (random, final, random with num=500) with each of the render
kernels named on the command line (auto, fused, queue; default: all three). MIRT_LIB selects the library build,
so variants can be compared by running this once per build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

KERNELS = {"auto": rt.KERNEL_AUTO, "fused": rt.KERNEL_FUSED, "queue": rt.KERNEL_QUEUE}
out = [os.path.basename(rt.LIB_PATH)]
for kname in (sys.argv[1:] or list(KERNELS)):
  for scene, spp, depth, (W, H), kw in [("random", 32, 50, (1920, 1080), {}), ("final", 16, 50, (1920, 1080), {}),
                                      ("random", 2, 10, (3840, 2160), {"random_num": 500})]:
    sc = rt.select_scene(scene, **kw)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    best = 0.0
    for rep in range(2):
        ctx.frame_begin(W, H)
        ctx.render(spp, depth, seed=rep, kernel=KERNELS[kname])
        ctx.synchronize()
        st = ctx.stats()
        best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
    extra = f", loop occupancy {st['segments'] / max(st['lane_slots'], 1):.2f}"
    out.append(f"{kname:5s} {scene}{'-1M' if kw else ''} {best:8.1f} Msamples/s ({st['segments'] / st['kernel_ms'] / 1e6:.2f} Gseg/s{extra})")
print("\n   ".join(out), flush=True)
