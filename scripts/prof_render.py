"""Small fixed workload for rocprofv3 counter passes: one render of `scene` (default Cornell 1080p, 64 spp)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
ctx = rt.Context(0)
sc = rt.select_scene(scene)
ctx.scene_upload(sc)
ctx.camera_set(rt.Camera.select_camera(W / H, scene))
ctx.frame_begin(W, H)
ctx.render(spp, 50, seed=0)
ctx.synchronize()
st = ctx.stats()
print(scene, W, H, spp, "kernel_ms", st["kernel_ms"], "Msamples/s", st["samples"] / st["kernel_ms"] / 1e3,
      "Gseg/s", st["segments"] / st["kernel_ms"] / 1e6, "seg/sample", st["segments"] / st["samples"], "lane occupancy of loop", st["segments"] / max(st["lane_slots"], 1))
