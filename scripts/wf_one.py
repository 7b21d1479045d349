"""One render with a chosen kernel set, for rocprofv3 runs: python scripts/wf_one.py <case> <queue|wavefront|fused> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ray_tracing_amd as rt
from wf_ab import CASES

case, kind = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
scene, W, H, spp, depth, kw = CASES[case]
kern = {"queue": rt.KERNEL_QUEUE, "wavefront": rt.KERNEL_WAVEFRONT, "fused": rt.KERNEL_FUSED}[kind]
ctx = rt.Context(0)
ctx.scene_upload(rt.select_scene(scene, **kw))
ctx.camera_set(rt.Camera.select_camera(W / H, scene))
for rep in range(reps):
    ctx.frame_begin(W, H)
    ctx.render(spp, depth, seed=rep, jitter=spp != 1, kernel=kern)
    ctx.synchronize()
    st = ctx.stats()
    print(case, kind, "rep", rep, "kernel_ms", round(st["kernel_ms"], 3), "Msamples/s", round(st["samples"] / st["kernel_ms"] / 1e3), "launches", st["kernel_launches"], flush=True)
