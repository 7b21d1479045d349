"""Persistent kernel against the wavefront kernel set on the hierarchy scenes (and Cornell for reference):
Msamples/s from the HIP events around each render call (read-backs of the wavefront driver included), second call
of each kind (the first one allocates). usage: python scripts/wf_ab.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

CASES = {
    "random": ("random", 1920, 1080, 32, 50, {}),
    "final": ("final", 1920, 1080, 16, 50, {}),
    "final256": ("final", 1920, 1080, 256, 50, {}),
    "million": ("random", 3840, 2160, 1, 10, {"random_num": 500}),
    "million4": ("random", 3840, 2160, 4, 10, {"random_num": 500}),
    "cornell": ("cornell", 1920, 1080, 64, 50, {}),
    "config1": ("random", 400, 225, 10, 50, {}),
}
def run_case(case):
    scene, W, H, spp, depth, kw = CASES[case]
    sc = rt.select_scene(scene, **kw)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    row = []
    for name, kern in (("queue", rt.KERNEL_QUEUE), ("wavefront", rt.KERNEL_WAVEFRONT)):
        best = 0.0
        for rep in range(3):
            ctx.frame_begin(W, H)
            ctx.render(spp, depth, seed=rep, jitter=spp != 1, kernel=kern)
            ctx.synchronize()
            st = ctx.stats()
            if rep:
                best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
        ctx.frame_begin(W, H)
        ctx.render(spp, depth, seed=0, jitter=spp != 1, kernel=kern | rt.KERNEL_COUNT_WALKS)
        ct = ctx.stats()
        row.append(f"{name} {best:.0f} ({st['kernel_launches']} launches; counted pass: {ct['bvh_node_steps'] / max(ct['segments'], 1):.1f} nodes ({ct['bvh_box_tests'] / max(ct['segments'], 1):.1f} boxes) + "
                   f"{(ct['bvh_sphere_tests'] + ct['bvh_rect_tests']) / max(ct['segments'], 1):.2f} leaf tests per segment, {ct['samples'] / ct['kernel_ms'] / 1e3:.0f} Msamples/s)")
    print(f"{case:9s} {W}x{H} spp={spp} depth={depth} prims={sc.desc.n_prims} deepest leaf {st['bvh_deepest_leaf']} seg/sample={st['segments'] / st['samples']:.2f}  Msamples/s: "
          + ", ".join(row), flush=True)


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["random", "final", "million", "cornell"]):
        run_case(c)
