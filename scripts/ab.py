"""Throughput of named workloads with the library MIRT_LIB names (default: the in-tree build), persistent kernel:
one line per case, best of `reps` renders after a warm-up. For A/B runs of two builds on one box call it once per
library. usage: python scripts/ab.py [--depth D] [--spp S] [--reps N] case [case ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

CASES = {
    "cornell": ("cornell", 1920, 1080, 256, 50, {}),
    "cornell1024": ("cornell", 1920, 1080, 1024, 50, {}),
    "smoke": ("smoke", 1920, 1080, 64, 50, {}),
    "random": ("random", 1920, 1080, 32, 50, {}),
    "final": ("final", 1920, 1080, 16, 50, {}),
    "final256": ("final", 1920, 1080, 256, 50, {}),
    "2psp": ("2psp", 1920, 1080, 32, 50, {}),
    "earth": ("earth", 1920, 1080, 64, 50, {}),
    "million": ("random", 3840, 2160, 1, 10, {"random_num": 500}),
    "million4": ("random", 3840, 2160, 4, 10, {"random_num": 500}),
    "config1": ("random", 400, 225, 10, 50, {}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kernel", default="queue", choices=["queue", "fused", "wavefront", "wavefront_classes", "auto"])
    ap.add_argument("--count", action="store_true", help="one more launch that tallies the walks: node steps and leaf tests per segment")
    ap.add_argument("cases", nargs="+")
    args = ap.parse_args()
    kern = {"queue": rt.KERNEL_QUEUE, "fused": rt.KERNEL_FUSED, "wavefront": rt.KERNEL_WAVEFRONT, "wavefront_classes": rt.KERNEL_WAVEFRONT | rt.KERNEL_WF_CLASSES,
            "auto": rt.KERNEL_AUTO}[args.kernel]
    tag = os.path.basename(os.environ.get("MIRT_LIB", "libmirt.so"))
    for case in args.cases:
        scene, W, H, spp, depth, kw = CASES[case]
        spp = args.spp or spp
        depth = args.depth or depth
        sc = rt.select_scene(scene, **kw)
        ctx = rt.Context(0)
        ctx.scene_upload(sc)
        ctx.camera_set(rt.Camera.select_camera(W / H, scene))
        best, st = 0.0, None
        for rep in range(args.reps + 1):
            ctx.frame_begin(W, H)
            ctx.render(spp, depth, seed=0, jitter=spp != 1, kernel=kern)
            ctx.synchronize()
            st = ctx.stats()
            if rep:
                best = max(best, st["samples"] / st["kernel_ms"] / 1e3)
        print(f"{tag:18s} {args.kernel:17s} {case:11s} {W}x{H} spp={spp} depth={depth} seg/sample={st['segments'] / st['samples']:.3f} "
              f"loop occupancy {st['segments'] / max(st['lane_slots'], 1):.3f} stack {st['bvh_stack_entries']}  {best:8.0f} Msamples/s", flush=True)
        if args.count:
            ctx.frame_begin(W, H)
            ctx.render(min(spp, 16), depth, seed=0, jitter=spp != 1, kernel=kern | rt.KERNEL_COUNT_WALKS)
            ctx.synchronize()
            ct = ctx.stats()
            seg = max(ct["segments"], 1)
            print(f"{'':18s} per segment: node steps {ct['bvh_node_steps'] / seg:.2f}, box tests {ct['bvh_box_tests'] / seg:.2f}, leaf spheres "
                  f"{ct['bvh_sphere_tests'] / seg:.2f}, leaf rects {ct['bvh_rect_tests'] / seg:.2f}", flush=True)


if __name__ == "__main__":
    main()
