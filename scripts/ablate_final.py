"""Where the time of a scene goes: renders it with one world entry left out at a time and prints the change of
kernel time. usage: ablate_final.py [scene] [spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

name = sys.argv[1] if len(sys.argv) > 1 else "final"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 1920, 1080
sc = rt.select_scene(name)
d = sc.desc
n = d.n_world
full = [rt.mirt_entry.from_buffer_copy(d.world[i]) for i in range(n)]
cam = rt.Camera.select_camera(W / H, name)


def run(keep):
    arr = (rt.mirt_entry * max(len(keep), 1))(*[full[i] for i in keep])
    d.world = C.cast(arr, C.POINTER(rt.mirt_entry))
    d.n_world = len(keep)
    ctx = rt.Context(0)
    rc = rt.lib().mirt_scene_upload(ctx._c, C.byref(d))
    assert rc == 0, rt.lib().mirt_last_error(ctx._c)
    ctx.camera_set(cam)
    best = None
    for rep in range(2):
        ctx.frame_begin(W, H)
        ctx.render(spp, 50, seed=rep)
        ctx.synchronize()
        st = ctx.stats()
        if best is None or st["kernel_ms"] < best[0]:
            best = (st["kernel_ms"], st["segments"] / st["samples"], st["segments"] / max(st["lane_slots"], 1))
    ctx.close()
    return best


base = run(list(range(n)))
print(f"{name} all {n} entries: {base[0]:.2f} ms, {base[1]:.2f} seg/sample, occupancy {base[2]:.2f}", flush=True)
for i in range(n):
    e = full[i]
    p = d.prims[e.first_prim]
    r = run([j for j in range(n) if j != i])
    print(f"  without entry {i} (kind {e.kind}, {e.prim_count} prims, first prim kind {p.kind}, inst {e.instance}): "
          f"{r[0]:.2f} ms ({r[0] - base[0]:+.2f}), {r[1]:.2f} seg/sample, occupancy {r[2]:.2f}", flush=True)

if name == "final":
    import copy
    for label, mod in [("entry 2 cut to 2 spheres", dict(prim_count=2)), ("entry 2 cut to 64 spheres", dict(prim_count=64)),
                       ("entry 2 without its instance", dict(instance=0)), ("entry 0 cut to 12 cuboids", None)]:
        saved = rt.mirt_entry.from_buffer_copy(full[2]), rt.mirt_entry.from_buffer_copy(full[0])
        if mod is None:
            full[0].prim_count = 72
        else:
            for k, v in mod.items():
                setattr(full[2], k, v)
        r = run(list(range(n)))
        print(f"  {label}: {r[0]:.2f} ms ({r[0] - base[0]:+.2f}), {r[1]:.2f} seg/sample, occupancy {r[2]:.2f}", flush=True)
        full[2], full[0] = saved
