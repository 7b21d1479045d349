"""Throughput of the hierarchy scenes (random, final, random with num=500) with the fused kernel set.
MIRT_LIB selects the library build, so variants can be compared by running this once per build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

out = [os.path.basename(rt.LIB_PATH)]
for scene, spp, depth, (W, H), kw in [("random", 32, 50, (1920, 1080), {}), ("final", 16, 50, (1920, 1080), {}),
                                      ("random", 2, 10, (3840, 2160), {"random_num": 500})]:
    sc = rt.select_scene(scene, **kw)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    best = 0.0
    for rep in range(2):
        ctx.frame_begin(W, H)
        ctx.render(spp, depth, seed=rep, kernel=rt.KERNEL_FUSED)
        ctx.synchronize()
        st = ctx.stats()
        best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
    extra = f", loop occupancy {st['segments'] / max(st['lane_slots'], 1):.2f}"
    out.append(f"{scene}{'-1M' if kw else ''} {best:8.1f} Msamples/s ({st['segments'] / st['kernel_ms'] / 1e6:.2f} Gseg/s{extra})")
print("\n   ".join(out), flush=True)
