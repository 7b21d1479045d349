"""Counter-pass workload: `final` at 1080p with the sphere cluster (world entry 2) cut to argv[1] spheres."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
cut = int(sys.argv[1]); spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 1920, 1080
sc = rt.select_scene("final"); d = sc.desc
d.world[2].prim_count = cut
ctx = rt.Context(0)
assert rt.lib().mirt_scene_upload(ctx._c, C.byref(d)) == 0
ctx.camera_set(rt.Camera.select_camera(W / H, "final"))
ctx.frame_begin(W, H); ctx.render(spp, 50); ctx.synchronize()
st = ctx.stats()
print("cut", cut, "kernel_ms", st["kernel_ms"], "segments", st["segments"], "rounds*64", st["lane_slots"])
