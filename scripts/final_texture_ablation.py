#!/usr/bin/env python3
"""GPU box: what the rare texture branches of `final` cost the waves that run them. The scene is rendered as it is, then with
its noise texture, its image texture, and both replaced by a solid colour (the same kernel form serves all four: the
hierarchy and the media keep it at the full feature set). usage: python scripts/final_texture_ablation.py [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt  # noqa: E402

TK_SOLID, TK_NOISE, TK_IMAGE = 0, 2, 3


def run(label, drop, spp):
    sc = rt.select_scene("final")
    d = sc.desc
    for i in range(d.n_textures):
        if d.textures[i].kind in drop:
            d.textures[i].kind = TK_SOLID
            d.textures[i].color[0] = d.textures[i].color[1] = d.textures[i].color[2] = 0.5
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(16 / 9, "final"))
    best, st = 0.0, None
    for rep in range(3):
        ctx.frame_begin(1920, 1080)
        ctx.render(spp, 50, seed=0, jitter=True, kernel=rt.KERNEL_QUEUE)
        ctx.synchronize()
        st = ctx.stats()
        if rep:
            best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
    print(f"{label:28s} seg/sample {st['segments'] / st['samples']:.3f} loop occupancy {st['segments'] / max(st['lane_slots'], 1):.3f} "
          f"{best:7.0f} Msamples/s  {st['kernel_ms'] / st['segments'] * 1e6:.4f} ns/segment", flush=True)


if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    run("final", (), spp)
    run("noise -> solid", (TK_NOISE,), spp)
    run("image -> solid", (TK_IMAGE,), spp)
    run("noise and image -> solid", (TK_NOISE, TK_IMAGE), spp)
    run("final", (), spp)
