import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
import oracle as O

def check_scene(name, W=64, H=36, **kw):
    earth = rt.load_earthmap() if name in ("earth", "final") else None
    orc = O.OracleRenderer(name, W, H, earth_rgba=earth, **kw)
    sc = rt.select_scene(name, **kw)
    cam = rt.Camera.select_camera(W / H, name)
    print(name, "camera diff", np.abs(cam.as_array() - orc.camera()).max())
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    rays = orc.camera_rays(0.5)
    oi, ot, oa = orc.first_hit(rays, aov=True)
    gi, gt, ga = ctx.first_hit(rays, aov=True)
    same = oi == gi
    hit = oi >= 0
    rel = np.abs(ot[hit & same] - gt[hit & same]) / np.abs(ot[hit & same])
    print(name, "id mismatches", (~same).sum(), "of", len(oi), "max rel t", rel.max() if rel.size else 0, "aov max abs diff", np.abs(oa - ga)[same].max())
    if (~same).any():
        idx = np.nonzero(~same)[0][:5]
        print("  mismatch", idx, oi[idx], gi[idx], ot[idx], gt[idx])
    # per-sample radiance
    oc = orc.ray_color(rays, sample=3, max_depth=50, seed=7)
    gc = ctx.ray_color(rays, sample=3, max_depth=50, seed=7)
    bad = ~np.isclose(oc, gc, rtol=1e-3, atol=1e-4, equal_nan=True)
    print(name, "ray_color mismatching rays", bad.any(axis=1).sum(), "of", len(oc))
    return ctx

for nm in sys.argv[1:] or ["cornell"]:
    check_scene(nm)

W, H = 1920, 1080
sc = rt.select_scene("cornell"); cam = rt.Camera.select_camera(W / H, "cornell")
ctx = rt.Context(0); ctx.scene_upload(sc); ctx.camera_set(cam)
for spp in (4, 64):
    ctx.frame_begin(W, H)
    t0 = time.time(); ctx.render(spp, 50); ctx.synchronize(); dt = time.time() - t0
    st = ctx.stats()
    print("cornell 1080p spp", spp, "wall", dt, "kernel_ms", st["kernel_ms"], "Msamples/s", W * H * spp / st["kernel_ms"] / 1e3, "segments/sample", st["segments"] / st["samples"])
acc = ctx.read_accum()
print("mean", (acc[..., :3] / acc[..., 3:4]).mean(axis=(0, 1)))
