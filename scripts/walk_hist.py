"""Reads the counters of the statistics build of the library (make -C ray_tracing_amd/csrc stats; run with
MIRT_LIB=ray_tracing_amd/libmirt_stats.so): hierarchy steps per lane and per wave, shader clocks per phase of an
iteration, wave lifetimes against the longest wave. CUT=n cuts the sphere cluster of `final` to n spheres."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import ray_tracing_amd as rt
L = rt.lib()
cut = int(os.environ.get("CUT", "0"))
for scene, spp, depth, (W, H), kw in [("cornell", 64, 50, (1920, 1080), {}), ("random", 8, 50, (1920, 1080), {}), ("final", 8, 50, (1920, 1080), {}),
                                      ("random", 2, 10, (3840, 2160), {"random_num": 500})]:
    sc = rt.select_scene(scene, **kw)
    if cut and scene == "final":
        sc.desc.world[2].prim_count = cut
    ctx = rt.Context(0); ctx.scene_upload(sc); ctx.camera_set(rt.Camera.select_camera(W / H, scene))
    ctx.frame_begin(W, H); ctx.render(spp, depth, kernel=rt.KERNEL_QUEUE); ctx.synchronize(); print('kernel_ms', ctx.stats()['kernel_ms'])
    h = (C.c_uint64 * 24)(); L.mirt_debug_walk_hist(h); h = list(h)
    nl, nw = max(sum(h[0:7]), 1), max(sum(h[7:14]), 1)
    print(scene, kw, "lane walks", nl, "mean steps/lane", h[14] / nl, "mean wave max", h[15] / nw)
    print("  lane hist <8,<16,<32,<64,<128,<256,more:", [round(v / nl, 3) for v in h[0:7]])
    print("  wave-max hist:", [round(v / nw, 3) for v in h[7:14]])
    print("  cycles per wave iteration: lists %.0f, walk %.0f, shade %.0f (iterations %d)" % (h[16] / h[19], h[17] / h[19], h[18] / h[19], h[19]))
    print("  waves %d, mean wave lifetime %.0f cycles, longest %.0f" % (h[22], h[20] / h[22], h[21]))
