"""`final` with its noise / image textures or media replaced one at a time: what each costs per render."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
W, H, spp = 1920, 1080, 32
cam = rt.Camera.select_camera(W / H, "final")

def run(mod):
    sc = rt.select_scene("final"); d = sc.desc
    mod(d)
    ctx = rt.Context(0)
    assert rt.lib().mirt_scene_upload(ctx._c, C.byref(d)) == 0, rt.lib().mirt_last_error(ctx._c)
    ctx.camera_set(cam)
    best = 1e9
    for rep in range(3):
        ctx.frame_begin(W, H); ctx.render(spp, 50, seed=rep); ctx.synchronize(); best = min(best, ctx.stats()["kernel_ms"])
    return best

def tex_solid(kind):
    def f(d):
        for i in range(d.n_textures):
            if d.textures[i].kind == kind: d.textures[i].kind = 0
    return f
def no_media(d):
    for i in range(d.n_prims):
        if d.prims[i].kind == 2: d.prims[i].neg_inv_density = -1e30   # density ~0: the medium never scatters
def metal_to_lambert(d):
    for i in range(d.n_materials):
        if d.materials[i].kind == 1: d.materials[i].kind = 0
base = run(lambda d: None)
print(f"final {spp} spp: {base:.2f} ms")
for label, m in [("noise texture -> solid", tex_solid(2)), ("image texture -> solid", tex_solid(3)), ("media never scatter", no_media), ("metal -> lambertian", metal_to_lambert)]:
    t = run(m); print(f"  {label}: {t:.2f} ms ({t - base:+.2f})")
