"""How long one loop iteration of the persistent kernel takes when a wave has the device to itself: `random` at
8x8 .. 400x225 pixels, 1 spp, depth 50 — kernel time against the iterations of the longest wave (lane_slots / 64
summed over waves; for 8x8 there is one wave). Small frames are bound by the longest path times this latency."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracing_amd as rt
sc = rt.select_scene("random")
ctx = rt.Context(0); ctx.scene_upload(sc); ctx.camera_set(rt.Camera.select_camera(400 / 225, "random"))
for (w, h, spp) in ((8, 8, 1), (8, 8, 1), (16, 8, 1), (64, 64, 1), (400, 225, 1), (400, 225, 10)):
    for seed in (1, 2):
        ctx.frame_begin(w, h)
        ctx.render(spp, 50, seed=seed, jitter=True, kernel=rt.KERNEL_QUEUE)
        ctx.synchronize()
        st = ctx.stats()
        print(w, h, spp, "seed", seed, "kernel_ms %.3f" % st["kernel_ms"], "segments", st["segments"], "wave iterations (lane_slots/64)", st["lane_slots"] // 64, flush=True)
