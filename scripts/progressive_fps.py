"""Progressive use (lib.rs:158 replaced by passes): k samples per pixel per frame + RGBA8 resolve, frames per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
W, H = 1920, 1080
for scene, k, depth in (("cornell", 4, 10), ("random", 1, 10), ("final", 1, 10)):
    sc = rt.select_scene(scene); ctx = rt.Context(0); ctx.scene_upload(sc); ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    frame = np.zeros(W * H * 4, dtype=np.uint8)
    ctx.frame_begin(W, H)
    n = 60
    for warm in range(3):
        ctx.render(k, depth, spp_begin=warm * k); ctx.resolve_rgba8(frame)
    ctx.frame_begin(W, H)
    t0 = time.perf_counter()
    for f in range(n):
        ctx.render(k, depth, spp_begin=f * k, seed=1)
        ctx.resolve_rgba8(frame)          # blocks; copies 8.3 MB to the host
    dt = time.perf_counter() - t0
    print(f"{scene}: {k} spp per frame, depth {depth}, 1920x1080 with host copy: {n / dt:.0f} frames/s ({dt / n * 1e3:.2f} ms per frame)")
