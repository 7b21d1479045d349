"""GPU tests of the scaled random-spheres configuration: the generator of scene.rs:46-52 with num = 500
(1 000 004 spheres in one world list, which the reference scans linearly and the device walks through a hierarchy),
3840x2160, depth 10.

The closest geometric hit is the same either way, but the reference's f32 sphere test (geo.rs:89-105:
half_b^2 - a*c) has no significant bits left once t / r reaches a few thousand, and then reports hits on spheres
the ray passes at more than a radius ("phantom hits"). A linear scan offers every sphere to that test and finds
them; a hierarchy never offers a sphere whose box the ray misses. The first-hit test therefore allows an id to
differ from the oracle's linear scan only where a float64 evaluation shows the oracle's sphere to be such a phantom,
and bounds their share."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, NUM = 3840, 2160, 500


def _oracle_first_hit_parallel(orc, rays, workers):
    """The oracle's scan is one thread per call; ctypes releases the GIL, so slices of the rays run side by side
    (no media in this scene, so the per-ray draw position, which follows the index inside a call, is not used)."""
    parts = np.array_split(np.arange(len(rays)), workers * 4)
    with ThreadPoolExecutor(workers) as ex:
        res = list(ex.map(lambda idx: orc.first_hit(rays[idx]), parts))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def _centres_at(prims, time):
    """Sphere centres at a ray time (geo.rs:62-69), float32 like the reference."""
    c0, c1 = prims[:, 1:4], prims[:, 5:8]
    moving, t0, t1 = prims[:, 8] != 0, prims[:, 9], prims[:, 10]
    with np.errstate(divide="ignore", invalid="ignore"):
        f = ((np.float32(time) - t0) / (t1 - t0)).astype(np.float32)
    c = (c0 + f[:, None] * (c1 - c0)).astype(np.float32)
    return np.where(moving[:, None], c, c0).astype(np.float32)


def _closest_without_phantoms(ray, centres, radius):
    """Closest hit of one ray over all spheres by the reference's f32 formula (geo.rs:89-105), leaving out the
    spheres a float64 evaluation says the ray does not touch; a later sphere wins a tie (ray.rs:127)."""
    f = np.float32
    o, d = ray[0:3].astype(f), ray[3:6].astype(f)
    oc = (o[None, :] - centres).astype(f)
    dot = lambda a, b: ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]).astype(f) + a[..., 2] * b[..., 2]).astype(f)
    a = dot(d, d)
    hb = dot(oc, d[None, :])
    c = (dot(oc, oc) - (radius * radius).astype(f)).astype(f)
    disc = ((hb * hb).astype(f) - (a * c).astype(f)).astype(f)
    oc64, d64 = o.astype(np.float64)[None, :] - centres.astype(np.float64), d.astype(np.float64)
    hb64 = oc64 @ d64
    disc64 = hb64 * hb64 - (d64 @ d64) * ((oc64 * oc64).sum(axis=1) - radius.astype(np.float64) ** 2)
    ok = (disc >= 0) & (disc64 >= 0)
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(np.where(ok, disc, 0)).astype(f)
        r0, r1 = ((-hb - sq) / a).astype(f), ((-hb + sq) / a).astype(f)
    t = np.where((r0 >= f(0.001)), r0, np.where(r1 >= f(0.001), r1, np.inf))
    t = np.where(ok, t, np.inf)
    if not np.isfinite(t).any():
        return -1, np.inf
    best = t.min()
    return int(np.flatnonzero(t == best).max()), float(best)


def test_first_hit_million_spheres_against_linear_scan(rt, oracle):
    orc = oracle.OracleRenderer("random", W, H, random_num=NUM)
    n_prims = orc.num_leaf_ids
    assert n_prims > 990_000
    sc = rt.select_scene("random", random_num=NUM)
    assert sc.desc.n_prims == n_prims
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    rays = orc.camera_rays(0.5)
    pick = np.random.default_rng(7).choice(len(rays), 6000, replace=False)
    pick.sort()
    rays = np.ascontiguousarray(rays[pick])
    oi, ot = _oracle_first_hit_parallel(orc, rays, min(os.cpu_count() or 1, 16))
    gi, gt = ctx.first_hit(rays)
    assert (oi >= 0).mean() > 0.3
    same = oi == gi
    hit = same & (oi >= 0)
    assert (np.abs(ot[hit] - gt[hit]) / np.abs(ot[hit])).max() < 1e-5
    assert np.isinf(gt[same & (oi < 0)]).all()
    assert ctx.stats()["bvh_deepest_leaf"] - 1 <= 32
    # every difference is a phantom hit of the f32 sphere test that only the linear scan can find
    bad = np.flatnonzero(~same)
    assert len(bad) <= 0.001 * (oi >= 0).sum(), len(bad)
    if len(bad):
        prims = orc.dump_prims()
        assert (prims[:, 0] == 0).all()
        centres, radius = _centres_at(prims, 0.5), prims[:, 4].astype(np.float32)
        for i in bad:
            o64, d64 = rays[i, 0:3].astype(np.float64), rays[i, 3:6].astype(np.float64)
            oc = o64 - centres[oi[i]].astype(np.float64)
            hb = oc @ d64
            disc = hb * hb - (d64 @ d64) * (oc @ oc - float(radius[oi[i]]) ** 2)
            assert oi[i] >= 0 and disc < 0, (int(i), int(oi[i]), int(gi[i]), disc)      # the oracle's sphere is not touched
            assert gt[i] >= ot[i]                                                          # and the device's hit lies behind it
            want_id, want_t = _closest_without_phantoms(rays[i], centres, radius)
            assert gi[i] == want_id, (int(i), int(gi[i]), want_id)
            if want_id >= 0:
                assert abs(gt[i] - want_t) <= 1e-5 * want_t


def test_full_size_pass_properties(rt, oracle):
    """One progressive pass at full size (3840x2160, 1 sample per pixel, depth 10 — the viewer's default depth,
    gui.rs:156): every pixel gets its sample, the persistent kernel and the one-pixel-per-lane kernel agree bit for
    bit (one sample per call = one chunk), no device fault, and the segments per sample agree with a low-resolution
    render of the same view by the oracle's linear scan."""
    sc = rt.select_scene("random", random_num=NUM)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, "random"))
    out = {}
    for k in (rt.KERNEL_QUEUE, rt.KERNEL_FUSED):
        ctx.frame_begin(W, H)
        ctx.render(1, 10, seed=6, jitter=False, kernel=k)
        st = ctx.stats()
        out[k] = (ctx.read_accum(), st)
        assert st["device_faults"] == 0 and st["kernel_used"] == k
    q, f = out[rt.KERNEL_QUEUE], out[rt.KERNEL_FUSED]
    assert (q[0][..., 3] == 1).all() and (q[0] == f[0]).all()
    assert q[1]["segments"] == f[1]["segments"]
    assert np.isfinite(q[0]).all() and q[0][..., :3].mean() > 0.05
    assert q[1]["bvh_node_steps"] == 0                     # nothing is tallied unless asked for
    # a counting pass: same image, and plausible walk statistics — tens of interior nodes per segment, a leaf sphere
    # for every few of them — identical from the persistent kernel and the wavefront walk kernel (same walks)
    tallies = []
    for k in (rt.KERNEL_QUEUE, rt.KERNEL_WAVEFRONT):
        ctx.frame_begin(W, H)
        ctx.render(1, 10, seed=6, jitter=False, kernel=k | rt.KERNEL_COUNT_WALKS)
        st = ctx.stats()
        assert (ctx.read_accum() == q[0]).all() and st["segments"] == q[1]["segments"]
        tallies.append((st["bvh_node_steps"], st["bvh_box_tests"], st["bvh_sphere_tests"], st["bvh_rect_tests"]))
        nodes = st["bvh_node_steps"] / st["segments"]
        assert 5 < nodes < 60 and 0.5 * st["segments"] < st["bvh_sphere_tests"] < st["bvh_node_steps"] and st["bvh_rect_tests"] == 0
        assert 2 * st["bvh_node_steps"] <= st["bvh_box_tests"] <= 4 * st["bvh_node_steps"]       # four-wide nodes hold 2-4 children
        assert 0 < st["bvh_stack_entries"] <= 64
    assert tallies[0] == tallies[1]
    # a second pass on top: progressive refinement keeps adding
    ctx.frame_begin(W, H)
    for s in range(2):
        ctx.render(1, 10, spp_begin=s, seed=6, jitter=True)
    assert (ctx.read_accum()[..., 3] == 2).all()
    small = oracle.OracleRenderer("random", 192, 108, random_num=NUM)
    _, cnt = small.render_sums(1, 10, seed=6)
    want = int(cnt[0]) / (192 * 108)
    got = q[1]["segments"] / (W * H)
    assert abs(got - want) <= 0.02 * want, (got, want)
