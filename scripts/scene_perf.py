"""Throughput of every preset scene at 1920x1080 (spp chosen per scene) with the three forms of the sample loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
W, H = 1920, 1080
for scene, spp, kw in [("cornell", 256, {}), ("smoke", 64, {}), ("random", 32, {}), ("final", 16, {}), ("2psp", 32, {}), ("earth", 64, {}),
                       ("random", 4, {"random_num": 500})]:
    sc = rt.select_scene(scene, **kw)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    row = []
    for name, kern in (("queue", rt.KERNEL_QUEUE), ("fused", rt.KERNEL_FUSED), ("wavefront", rt.KERNEL_WAVEFRONT)):
      ctx.frame_begin(W, H)
      ctx.render(spp, 50 if not kw else 10, seed=0, kernel=kern)
      ctx.synchronize()
      st = ctx.stats()
      row.append(f"{name} {st['samples'] / st['kernel_ms'] / 1e3:.0f}")
    print(f"{scene:8s} {kw} prims={sc.desc.n_prims} spp={spp} seg/sample={st['segments'] / st['samples']:.2f}  Msamples/s: " + ", ".join(row), flush=True)
