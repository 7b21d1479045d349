"""Cornell 1080p 1024 spp, best and mean kernel time of 4 renders — for comparing library builds (MIRT_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
W, H = 1920, 1080
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = rt.select_scene(scene); ctx = rt.Context(0); ctx.scene_upload(sc); ctx.camera_set(rt.Camera.select_camera(W / H, scene))
ts = []
for rep in range(5):
    ctx.frame_begin(W, H); ctx.render(spp, 50, seed=rep); ctx.synchronize(); ts.append(ctx.stats()["kernel_ms"])
ts = ts[1:]
print(os.path.basename(rt.LIB_PATH), scene, spp, "kernel ms best %.2f mean %.2f -> %.0f Msamples/s" % (min(ts), sum(ts) / len(ts), W * H * spp / min(ts) / 1e3))
