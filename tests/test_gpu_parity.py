"""GPU parity tests: the HIP path (through the C ABI of libmirt.so) against the CPU oracle on the same
seeded inputs. Because both sides address their random draws the same way (Philox key = pixel, sample;
counter = block, bounce, stream, seed), a path on the GPU follows the same decisions as in the oracle
and differs only by the last bits of sin / cos / log / division; the tolerances below are for that."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import kat_rays, observe

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "smoke", "random", "final", "2sp", "2psp", "earth", "simplelight"]
W, H = 64, 36
# Tolerance tables. A sample or pixel falls outside the tolerance only when a last-bit difference between the device's
# arithmetic (hardware reciprocal / square root / sine, fused multiply-adds in shading) and the oracle's diverts a
# path: another side of an edge, a rejection-sampling trial accepted on one side only — and, among the sphere clusters
# of `final` and `random`, because reflections off spheres amplify a last-bit difference of the camera ray from bounce
# to bounce until the two paths part. Each bound is what was observed on the MI355X (every run appends what it sees to
# gpurun_out/observed_parity.jsonl) with a margin of half the observed shortfall again, at least 0.2 % of the pixels
# (0.05 % of the samples), so a change that diverts noticeably more paths than today fails.
# Observed share of diverted single samples: cornell 0.00043, random 0.00022, all others 0.
SAMPLE_DIVERTED_MAX = {"cornell": 0.00115, "random": 0.00083, "smoke": 0.0005, "final": 0.0005, "2sp": 0.0005, "2psp": 0.0005,
                       "earth": 0.0005, "simplelight": 0.0005}
# Observed share of close pixels: cornell 0.9952, random 0.9918, final 0.9705 (0.9731 before shading used fused
# multiply-adds), earth 0.9983, smoke / 2sp / 2psp / simplelight 1.
IMAGE_CLOSE_MIN = {"cornell": 0.9928, "random": 0.9876, "final": 0.9557, "earth": 0.9974, "smoke": 0.998, "2sp": 0.998, "2psp": 0.998,
                   "simplelight": 0.998}
# Observed: one pixel of the 37 x 19 frame is not close (0.9986), none of the others.
RAGGED_CLOSE_MIN = {(37, 19): 0.997, (8, 8): 0.98, (33, 9): 0.993, (2, 2): 1.0}


def sanitise(c):
    c = np.where(np.isnan(c), 0.0, c)
    return np.where(np.isinf(c), 1.0, c).astype(np.float32)


def resample_onto(gh, gw, img):
    """A large frame averaged over the pixel footprints of a small frame of the same view: pixel X of a W-wide frame
    covers u in [X, X + 1) / (W - 1) (renderer.rs:181-182), so the weights are the exact overlaps of the footprints.
    The small frame's last column and row reach beyond u = 1, where the large frame has nothing: they are left out
    (the result is (gh - 1) x (gw - 1))."""
    h, w = img.shape[:2]

    def overlap(n_small, n_big):
        lo = np.arange(n_small)[:, None] * (n_big - 1) / (n_small - 1)     # footprint of small pixel X in big-pixel units
        hi = lo + (n_big - 1) / (n_small - 1)
        x = np.arange(n_big)[None, :]
        wgt = np.clip(np.minimum(hi, x + 1) - np.maximum(lo, x), 0, None)
        return (wgt / (hi - lo)).astype(np.float64)

    rows = (overlap(gh, h) @ img.astype(np.float64).reshape(h, w * 3)).reshape(gh, w, 3)   # two matrix products, rows first
    return np.einsum("Xx,Yxc->YXc", overlap(gw, w), rows)[:gh - 1, :gw - 1]


def make(rt, oracle, earth, name, w=W, h=H, **kw):
    e = earth if name in ("earth", "final") else None
    orc = oracle.OracleRenderer(name, w, h, earth_rgba=e, **kw)
    sc = rt.select_scene(name, earth_rgba=e, **kw)
    cam = rt.Camera.select_camera(w / h, name)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(cam)
    return orc, ctx, sc


def test_cornell_known_answers_on_gpu(rt, kat):
    ctx = rt.Context(0)
    sc = rt.select_scene("cornell")
    ctx.scene_upload(sc)
    ids, t = ctx.first_hit(kat_rays(kat))
    for row, i, tt in zip(kat["rays"], ids, t):
        assert i == row["prim_id"], row
        if row["prim_id"] >= 0:
            assert abs(tt - row["hit_t"]) / row["hit_t"] < 1e-5, row


@pytest.mark.parametrize("name", SCENES)
def test_first_hit_matches_oracle(rt, oracle, earth, name):
    """Deterministic check: primary-ray first-hit ids equal, t within 1e-5 relative, on the full grid of
    fixed camera rays (lens offset 0, time 0.5)."""
    orc, ctx, _ = make(rt, oracle, earth, name)
    rays = orc.camera_rays(0.5)
    oi, ot, oa = orc.first_hit(rays, seed=11, aov=True)
    gi, gt, ga = ctx.first_hit(rays, seed=11, aov=True)
    assert (oi == gi).all(), f"{(oi != gi).sum()} id mismatches"
    hit = oi >= 0
    assert hit.any()
    assert np.isinf(gt[~hit]).all()
    rel = np.abs(ot[hit] - gt[hit]) / np.abs(ot[hit])
    assert rel.max() < 1e-5
    # hit record: point, normal, front_face, material colour (uv only where an image texture needs it)
    scale = np.abs(oa[hit, 0:3]).max()
    assert np.abs(oa[hit, 0:3] - ga[hit, 0:3]).max() < 2e-5 * max(scale, 1.0)
    assert np.abs(oa[hit, 3:6] - ga[hit, 3:6]).max() < 1e-5
    assert (oa[hit, 6] == ga[hit, 6]).all()
    if name in ("earth", "final"):
        assert np.abs(oa[hit, 7:9] - ga[hit, 7:9]).max() < 1e-4
    col_bad = np.abs(oa[hit, 9:12] - ga[hit, 9:12]).max(axis=1) > 2e-3
    # a texel / checker boundary may flip on the last bit of uv or sin(); never more than a few rays
    assert col_bad.mean() < 0.01, col_bad.sum()


def test_first_hit_full_resolution_cornell(rt, oracle, earth):
    orc, ctx, _ = make(rt, oracle, earth, "cornell", 1920, 1080)
    rays = orc.camera_rays(0.25)
    oi, ot = orc.first_hit(rays)
    gi, gt = ctx.first_hit(rays)
    assert (oi == gi).all()
    hit = oi >= 0
    assert (np.abs(ot[hit] - gt[hit]) / ot[hit]).max() < 1e-5
    assert set(np.unique(oi)) == {-1, 0, 1, 2, 3, 4, 5, 10, 11, 12}   # box sides 6-9 face away


def test_bvh_equals_linear_scan_on_random_rays(rt, oracle, earth):
    """The flattened BVH (final: 400 cuboids, 1000 instanced spheres; random: auto-built) returns what the
    oracle's list / pointer-BVH traversal returns for arbitrary rays, not only camera rays."""
    rng = np.random.default_rng(4)
    for name, centre, spread in (("final", [200.0, 250.0, 200.0], 400.0), ("random", [0.0, 1.0, 0.0], 8.0)):
        orc, ctx, _ = make(rt, oracle, earth, name)
        n = 20000
        o = (np.asarray(centre) + rng.uniform(-1, 1, (n, 3)) * spread).astype(np.float32)
        if name == "final":
            # Neighbouring ground cuboids share whole faces (scene.rs:230-232): a ray that starts inside
            # one hits two coincident rects at the same t, and which of the two ids wins depends on the
            # tree shape (random in the reference, aabb.rs:73). Start above the boxes to keep such exact
            # ties out of the id statistics; they remain covered by the equal-t assertion below.
            o[:, 1] = np.abs(o[:, 1] - 110.0) + 110.0
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d, rng.uniform(0, 1, (n, 1)).astype(np.float32)], axis=1)
        oi, ot = orc.first_hit(rays, seed=3)
        gi, gt = ctx.first_hit(rays, seed=3)
        same = oi == gi
        # an id may differ only where two surfaces are hit at (numerically) the same t
        assert (np.abs(ot[~same] - gt[~same]) <= 1e-5 * np.abs(ot[~same])).all(), name
        assert same.mean() > 0.999, name
        hit = same & (oi >= 0)
        assert (np.abs(ot[hit] - gt[hit]) / np.abs(ot[hit])).max() < 1e-5, name


@pytest.mark.parametrize("name", SCENES)
def test_single_samples_match_oracle(rt, oracle, earth, name):
    """ray_color for the same ray with the same draws: the whole path (scatter decisions, mixture pdf,
    media, textures) must agree sample by sample for all but the rare path a last-bit difference diverts."""
    orc, ctx, _ = make(rt, oracle, earth, name)
    rays = orc.camera_rays(0.5)
    bad_total = 0
    for sample in (0, 5):
        oc = orc.ray_color(rays, sample=sample, max_depth=50, seed=7)
        gc = ctx.ray_color(rays, sample=sample, max_depth=50, seed=7)
        # compare what the renderer keeps of a sample: NaN -> 0, +-inf -> 1 (renderer.rs:107). `random`
        # does produce NaN samples: small diffuse spheres reach inside the glass light sphere, where
        # Sphere::pdf_val takes the square root of a negative number (geo.rs:138).
        oc, gc = sanitise(oc), sanitise(gc)
        bad = ~np.isclose(oc, gc, rtol=2e-3, atol=1e-4)
        bad_total += bad.any(axis=1).sum()
    observe("single_samples", scene=name, diverted_share=bad_total / (2 * len(rays)))
    assert bad_total <= SAMPLE_DIVERTED_MAX[name] * 2 * len(rays), bad_total


@pytest.mark.parametrize("name,spp,depth", [("cornell", 64, 50), ("smoke", 32, 50), ("random", 32, 50), ("final", 16, 50),
                                            ("2psp", 32, 50), ("earth", 32, 50), ("simplelight", 32, 50), ("2sp", 32, 50)])
def test_accumulated_image_matches_oracle(rt, oracle, earth, name, spp, depth):
    """Sum of sanitised samples per pixel (renderer.rs:100-110) with identical draws."""
    orc, ctx, _ = make(rt, oracle, earth, name)
    want, cnt = orc.render_sums(spp, depth, seed=3)
    ctx.frame_begin(W, H)
    ctx.render(spp, depth, seed=3, jitter=True)
    acc = ctx.read_accum()
    assert (acc[..., 3] == spp).all()
    got = acc[..., :3]
    assert np.isfinite(got).all()
    st = ctx.stats()
    assert st["samples"] == W * H * spp
    if name == "random":
        # Shading points inside the glass light sphere give a NaN pdf and a NaN sampled direction
        # (geo.rs:80,138). The reference keeps tracing that NaN ray until max_depth (every NaN comparison
        # "hits"); the sample is NaN -> 0 either way (renderer.rs:107), so the GPU ends the path at once
        # and traces fewer segments for the same image.
        assert 0.90 * int(cnt[0]) <= st["segments"] <= 1.002 * int(cnt[0])
    else:
        assert abs(st["segments"] - int(cnt[0])) <= 0.002 * int(cnt[0]) + 2  # same paths -> same segment count
    close = np.isclose(got, want, rtol=5e-3, atol=5e-3 * spp * 0.05)
    observe("accumulated_image", scene=name, close_share=close.all(axis=2).mean(), worst=np.abs(got - want).max(),
            mean_rel=abs(got.mean() - want.mean()) / want.mean())
    assert close.all(axis=2).mean() >= IMAGE_CLOSE_MIN[name], close.all(axis=2).mean()
    # A pixel that is not close holds a path that a last-bit difference diverted (another side of an edge, a free
    # path that ends a hair earlier): its SUM then differs by what one sample can carry, not by an arbitrary amount.
    # One sample carries at most the brightest source of the scene times a weight of order one: measured worst cases
    # over four seeds are 6.1 (cornell, emitter 15), 2.0 (random, sky 1), 15 (final, emitter 15); the cap is 4 sources.
    brightest = {"cornell": 15.0, "smoke": 7.0, "final": 15.0, "simplelight": 4.0}.get(name, 1.0)
    assert np.abs(got - want).max() <= 4.0 * brightest, np.abs(got - want).max()
    mean_g, mean_o = got.mean() / spp, want.mean() / spp
    assert abs(mean_g - mean_o) <= 0.01 * mean_o + 1e-4


def test_converged_cornell_statistical_parity(rt, oracle, earth):
    """Statistical check with DIFFERENT seeds: converged images (4096 spp) agree within the Monte-Carlo
    noise floor measured between two oracle runs. Tolerance: per-channel RMSE of the GPU image against
    oracle run A is at most 1.5x the RMSE between oracle runs A and B; mean luminance within 1%."""
    w, h, spp = 48, 27, 4096
    orc, ctx, _ = make(rt, oracle, earth, "cornell", w, h)
    a = orc.render_sums(spp, 50, seed=101)[0] / spp
    b = orc.render_sums(spp, 50, seed=202)[0] / spp
    ctx.frame_begin(w, h)
    ctx.render(spp, 50, seed=303)
    acc = ctx.read_accum()
    g = acc[..., :3] / acc[..., 3:4]
    for c in range(3):
        floor = math.sqrt(((a[..., c] - b[..., c]) ** 2).mean())
        rmse = math.sqrt(((g[..., c] - a[..., c]) ** 2).mean())
        assert rmse <= 1.5 * floor + 1e-4, (c, rmse, floor)
    lum = lambda im: (0.2126 * im[..., 0] + 0.7152 * im[..., 1] + 0.0722 * im[..., 2]).mean()
    assert abs(lum(g) - lum(a)) <= 0.01 * lum(a)


def test_renderer_draw_matches_oracle_draw(rt, oracle, earth):
    """The drop-in boundary: Renderer::new + multisample / max_depth + draw(frame) (cli.rs:25-30)."""
    for name, spp in (("cornell", 16), ("random", 8)):
        sc = rt.select_scene(name)
        cam = rt.Camera.select_camera(W / H, name)
        r = rt.Renderer(W, H, sc, cam)
        assert (r.multisample, r.max_depth, r.dirty) == (4, 10, True)       # renderer.rs:50-52
        r.multisample, r.max_depth, r.seed = spp, 50, 9
        frame = np.zeros(W * H * 4, dtype=np.uint8)
        r.draw(frame)
        assert not r.dirty
        want = oracle.OracleRenderer(name, W, H).draw(spp, 50, seed=9).reshape(-1)
        diff = np.abs(frame.astype(np.int32) - want.astype(np.int32))
        assert (frame.reshape(-1, 4)[:, 3] == 255).all()
        assert (diff <= 1).mean() > 0.98 and diff.mean() < 0.5, (name, diff.mean())
        # not dirty -> draw is a no-op (renderer.rs:57-60)
        before = frame.copy()
        frame[:] = 7
        r.draw(frame)
        assert (frame == 7).all()
        frame[:] = before
        with pytest.raises(ValueError):
            r.dirty = True
            r.draw(np.zeros(W * H * 4 - 4, dtype=np.uint8))               # renderer.rs:65


def test_no_jitter_when_single_sample(rt, oracle, earth):
    # renderer.rs:175: multisample == 1 -> pixel centres on the (W-1, H-1) lattice, no jitter draws
    orc, ctx, _ = make(rt, oracle, earth, "cornell")
    want, _ = orc.render_sums(1, 50, seed=4)
    ctx.frame_begin(W, H)
    ctx.render(1, 50, seed=4, jitter=False)
    got = ctx.read_accum()[..., :3]
    assert np.isclose(got, want, rtol=5e-3, atol=1e-4).all(axis=2).mean() > 0.99


def test_sample_ranges_add_up_and_are_deterministic(rt, oracle, earth):
    """Multi-GPU split by sample range: ranges [0,a) and [a,n) accumulate to the image of [0,n)."""
    _, ctx, _ = make(rt, oracle, earth, "cornell")
    ctx.frame_begin(W, H)
    ctx.render(24, 50, seed=5)
    full = ctx.read_accum()
    ctx.frame_begin(W, H)
    ctx.render(10, 50, spp_begin=0, seed=5)
    ctx.render(14, 50, spp_begin=10, seed=5)
    parts = ctx.read_accum()
    np.testing.assert_allclose(parts, full, rtol=1e-5, atol=1e-5)
    ctx.frame_begin(W, H)
    ctx.render(24, 50, seed=5)
    again = ctx.read_accum()
    assert (again == full).all()                                            # bitwise reproducible
    # more than one launch per call (chunks of 256 spp) gives the same sums as the oracle convention
    ctx.frame_begin(W, H)
    ctx.render(300, 6, seed=5)
    a = ctx.read_accum()
    assert (a[..., 3] == 300).all()


def test_full_size_properties(rt, oracle, earth):
    """At the benchmark size (1920x1080) the oracle is too slow for a pixel-wise check, so check
    size-independent properties: mean radiance and segments per sample agree with a low-resolution oracle
    render of the same view, the open front of the box is black, the emitter saturates."""
    w, h, spp = 1920, 1080, 16
    orc, ctx, _ = make(rt, oracle, earth, "cornell", w, h)
    small = oracle.OracleRenderer("cornell", 192, 108)
    ref, cnt = small.render_sums(64, 50, seed=1)
    ctx.frame_begin(w, h)
    ctx.render(spp, 50, seed=2)
    acc = ctx.read_accum()
    img = acc[..., :3] / acc[..., 3:4]
    assert abs(img.mean() - ref.mean() / 64) < 0.02 * ref.mean() / 64
    st = ctx.stats()
    assert abs(st["segments"] / st["samples"] - cnt[0] / (192 * 108 * 64)) < 0.07   # ~3.5 segments per sample
    frame = ctx.resolve_rgba8()
    assert frame.shape == (h, w, 4)
    assert (frame[:, :40, :3] == 0).all()          # rays left of the box miss everything: background 0
    ly, lx = int(h * (1 - 0.85108)), w // 2         # the light (tests/golden: s=0.5, t=0.85108), top of the frame
    assert (frame[ly, lx, :3] == 255).all()


def test_4k_frame_into_caller_owned_buffer(rt, oracle, earth):
    """3840x2160 (the size of the multi-GPU configuration) rendered into a caller-owned device buffer, as the
    multi-GPU split does before its reduce: the persistent kernel's chunk slots (16 x 8.3 M pixels) and the
    one-pixel-per-lane kernel give the same image, the caller's tensor receives it, and the row sums show the
    box where it should be (nothing left of it, the emitter at the top)."""
    w, h, spp = 3840, 2160, 3
    _, ctx, _ = make(rt, oracle, earth, "cornell", w, h)
    owner = rt.Context(0)                   # stands for the caller: its frame buffer is device memory ctx does not own
    owner.frame_begin(w, h)
    ctx.frame_begin(w, h, owner.accum_device_ptr())
    ctx.render(spp, 50, seed=4, kernel=rt.KERNEL_QUEUE)
    ctx.synchronize()
    got = owner.read_accum()
    assert (ctx.read_accum() == got).all()
    ctx.frame_begin(w, h)
    ctx.render(spp, 50, seed=4, kernel=rt.KERNEL_FUSED)
    want = ctx.read_accum()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    assert (got[..., 3] == spp).all()
    assert (got[:, :80, :3] == 0).all() and got[:, w // 2 - 100:w // 2 + 100, :3].sum() > 0


def test_eight_sample_ranges_of_the_4k_configuration(rt, oracle, earth):
    """The multi-GPU configuration at its full size on one device: Cornell 3840x2160, 16 384 samples per pixel,
    split into eight ranges of 2 048 (what eight ranks render, renderer.rs:97-119 with the sample loop cut by
    range). The eight accumulation buffers, added in rank order as the exchange step adds them, equal one render of
    all 16 384 samples up to the association of the f32 sums; every pixel has its count; no device fault; and the
    image, resampled to 96x54, agrees with the committed 16 384-sample mean image of the oracle (another seed) within
    that image's own noise."""
    import os
    from conftest import ROOT
    w, h, ranks, share, depth, seed = 3840, 2160, 8, 2048, 50, 11
    _, ctx, _ = make(rt, oracle, earth, "cornell", w, h)
    total = np.zeros((h, w, 4), np.float32)
    segments = 0
    for r in range(ranks):
        ctx.frame_begin(w, h)
        ctx.render(share, depth, spp_begin=r * share, seed=seed)
        part = ctx.read_accum()
        st = ctx.stats()
        assert st["device_faults"] == 0 and (part[..., 3] == share).all()
        segments += st["segments"]
        total += part
    ctx.frame_begin(w, h)
    ctx.render(ranks * share, depth, seed=seed)
    whole = ctx.read_accum()
    st = ctx.stats()
    assert st["device_faults"] == 0 and st["segments"] == segments          # the same paths either way
    assert (whole[..., 3] == ranks * share).all() and (total[..., 3] == ranks * share).all()
    np.testing.assert_allclose(total[..., :3], whole[..., :3], rtol=2e-5, atol=1e-3)
    # against the oracle's committed mean image (96x54), resampled onto its pixel footprints
    z = np.load(os.path.join(ROOT, "tests", "golden", "converged_cornell.npz"))
    mean, half = z["mean"], z["half_rmse"]
    gh, gw = mean.shape[:2]

    small = resample_onto(gh, gw, whole[..., :3] / whole[..., 3:4])
    want = mean[:gh - 1, :gw - 1].astype(np.float64)
    for c in range(3):
        rmse = math.sqrt(((small[..., c] - want[..., c]) ** 2).mean())
        assert rmse <= 1.0 * half[c] + 1e-4, (c, rmse, half[c])
    lum = lambda im: (0.2126 * im[..., 0] + 0.7152 * im[..., 1] + 0.0722 * im[..., 2]).mean()
    assert abs(lum(small) - lum(want)) <= 0.01 * lum(want), (lum(small), lum(want))


@pytest.mark.parametrize("name,w,h,spp", [("random", 400, 225, 10), ("cornell", 1920, 1080, 1024), ("final", 1920, 1080, 256)])
def test_benchmark_configurations_against_committed_means(rt, earth, name, w, h, spp):
    """The single-GPU benchmark configurations at their full sizes (depth 50) against fixed data: the render,
    resampled onto the footprints of tests/golden/converged_<scene>.npz (the oracle's mean image of N samples at
    96x54 and the RMSE between the means of its two halves, half = 2 sigma / sqrt(N)). With n device samples per
    small pixel the difference should have an RMSE of sigma sqrt(1/n + 1/N); the bar is 1.5 times that per channel
    (1.0 half for n >> N, as in the test at the fixture's own size) and 1 % of mean luminance (2 % for the
    10-sample configuration)."""
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", f"converged_{name}.npz"))
    mean, half, n_fix = z["mean"], z["half_rmse"], int(z["spp"])
    gh, gw = mean.shape[:2]
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene(name, earth_rgba=earth if name == "final" else None))
    ctx.camera_set(rt.Camera.select_camera(w / h, name))
    ctx.frame_begin(w, h)
    ctx.render(spp, int(z["depth"]), seed=777)
    acc = ctx.read_accum()
    st = ctx.stats()
    assert st["device_faults"] == 0 and (acc[..., 3] == spp).all()
    small = resample_onto(gh, gw, acc[..., :3] / acc[..., 3:4])
    want = mean[:gh - 1, :gw - 1].astype(np.float64)
    n_dev = spp * ((w - 1) / (gw - 1)) * ((h - 1) / (gh - 1))
    for c in range(3):
        sigma = half[c] * math.sqrt(n_fix) / 2
        rmse = math.sqrt(((small[..., c] - want[..., c]) ** 2).mean())
        assert rmse <= 1.5 * sigma * math.sqrt(1 / n_dev + 1 / n_fix) + 1e-4, (c, rmse, half[c], n_dev)
    lum = lambda im: (0.2126 * im[..., 0] + 0.7152 * im[..., 1] + 0.0722 * im[..., 2]).mean()
    assert abs(lum(small) - lum(want)) <= (0.02 if spp < 64 else 0.01) * lum(want), (lum(small), lum(want))


def test_error_paths(rt):
    ctx = rt.Context(0)
    with pytest.raises(rt.MirtError, match="no scene"):
        ctx.first_hit(np.zeros((1, 7), dtype=np.float32))
    sc = rt.select_scene("cornell")
    ctx.scene_upload(sc)
    with pytest.raises(rt.MirtError):
        ctx.frame_begin(1, 1)
    ctx.frame_begin(8, 8)
    with pytest.raises(rt.MirtError, match="camera"):
        ctx.render(1, 5)
    ctx.camera_set(rt.Camera.select_camera(1.0, "cornell"))
    ctx.render(1, 5)
    with pytest.raises(rt.MirtError, match="kernel"):
        ctx.render(1, 5, kernel=rt.KERNEL_QUEUE + 1)
    with pytest.raises(rt.MirtError, match="len"):
        ctx.resolve_rgba8(np.zeros(8 * 8 * 4 - 1, dtype=np.uint8))
    ctx.width = 9
    with pytest.raises(rt.MirtError, match="differ"):
        ctx.render(1, 5)
    # a broken descriptor is rejected before anything is uploaded
    d = sc.desc
    bad = rt.mirt_scene_desc.from_buffer_copy(d)
    bad.n_lights = 1
    arr = (rt.C.c_uint32 * 1)(999)
    bad.lights = arr
    with pytest.raises(rt.MirtError, match="light"):
        ctx.scene_upload(bad)
    with pytest.raises(rt.MirtError):
        rt.Context(99)


@pytest.mark.parametrize("by_class", [False, True])
@pytest.mark.parametrize("name,spp", [("cornell", 16), ("smoke", 8), ("random", 8), ("final", 4), ("2psp", 8)])
def test_wavefront_kernel_set_matches_fused(rt, oracle, earth, name, spp, by_class):
    """The queue-based kernel set (generate -> extend -> shade per ray segment) draws from the same Philox
    addresses as the fused kernel, so both produce the same samples; only the order of the float additions
    into the accumulation buffer differs. by_class: the paths are filed into one queue per class of what they do
    next (material.rs:19-23 ScatterType, or the path ends) and shaded class by class — the same samples again, and
    the same image bit for bit as the one-queue form."""
    wavefront = rt.KERNEL_WAVEFRONT | (rt.KERNEL_WF_CLASSES if by_class else 0)
    orc, ctx, _ = make(rt, oracle, earth, name)
    ctx.frame_begin(W, H)
    ctx.render(spp, 50, seed=6, kernel=rt.KERNEL_FUSED)
    fused = ctx.read_accum()
    seg_fused = ctx.stats()["segments"]
    ctx.frame_begin(W, H)
    ctx.render(spp, 50, seed=6, kernel=wavefront)
    wave = ctx.read_accum()
    st = ctx.stats()
    assert st["kernel_used"] == rt.KERNEL_WAVEFRONT and st["segments"] == seg_fused
    assert (wave[..., 3] == spp).all()
    if by_class:
        ctx.frame_begin(W, H)
        ctx.render(spp, 50, seed=6, kernel=rt.KERNEL_WAVEFRONT)
        assert (ctx.read_accum() == wave).all()
    np.testing.assert_allclose(wave[..., :3], fused[..., :3], rtol=2e-5, atol=1e-5)
    want, _ = orc.render_sums(spp, 50, seed=6)
    share = np.isclose(wave[..., :3], want, rtol=5e-3, atol=5e-3 * spp * 0.05).all(axis=2).mean()
    observe("wavefront_vs_oracle", scene=name, close_share=share)
    assert share >= IMAGE_CLOSE_MIN[name] - 0.005, share     # (other sample counts and seed than the table's test)
    # a sample range on top of an existing accumulation (progressive refinement)
    ctx.render(3, 50, spp_begin=spp, seed=6, kernel=wavefront)
    more = ctx.read_accum()
    assert (more[..., 3] == spp + 3).all()


def test_cli_image_mode_and_checkpoint_resume(rt, oracle, tmp_path):
    """cli.rs:21-38: flags -> Renderer -> screenshot.png; plus the pass-wise render with a saved accumulation
    buffer: a run that stops after some passes and is started again ends with the image of one run."""
    from PIL import Image
    from ray_tracing_amd import cli
    out = tmp_path / "shot.png"
    cli.image_mode(["-s", "16", "-m", "50", "-d", "30", "--scene", "cornell", "--seed", "9", "--output", str(out)])
    im = np.asarray(Image.open(out))
    assert im.shape == (36, 64, 4)
    want = oracle.OracleRenderer("cornell", 64, 36).draw(16, 50, seed=9)
    diff = np.abs(im.astype(np.int32) - want.astype(np.int32))
    assert (diff <= 1).mean() > 0.98 and diff.mean() < 0.5

    ck = tmp_path / "ck.npz"
    common = ["-m", "50", "-d", "30", "--scene", "cornell", "--seed", "9", "--pass-spp", "4", "--checkpoint", str(ck)]
    cli.image_mode(["-s", "8", "--output", str(tmp_path / "half.png")] + common)     # "interrupted" after 8 of 16
    with np.load(ck) as z:
        assert int(z["spp_done"]) == 8 and (z["accum"][..., 3] == 8).all()
    cli.image_mode(["-s", "16", "--output", str(tmp_path / "resumed.png")] + common)
    with np.load(ck) as z:
        assert int(z["spp_done"]) == 16 and (z["accum"][..., 3] == 16).all()
    resumed = np.asarray(Image.open(tmp_path / "resumed.png")).astype(np.int32)
    assert np.abs(resumed - im.astype(np.int32)).max() <= 1
    with pytest.raises(SystemExit):
        cli.image_mode(["-s", "16", "--seed", "10", "--output", str(out)] + common[:6] + ["--checkpoint", str(ck)])


@pytest.mark.parametrize("name", SCENES)
def test_first_hit_matches_committed_grid(rt, earth, name):
    """The device against fixed data: tests/golden/first_hit_<scene>.npz (made by make_oracle_fixtures.py)."""
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", f"first_hit_{name}.npz"))
    sc = rt.select_scene(name, earth_rgba=earth if name in ("earth", "final") else None)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    cam = rt.Camera.select_camera(W / H, name).c
    ys, xs = np.mgrid[0:H, 0:W]
    s, t = (xs / np.float32(W - 1)).astype(np.float32), (ys / np.float32(H - 1)).astype(np.float32)
    f = lambda v: np.asarray(list(v), dtype=np.float32)
    d = (f(cam.lower_left_corner) + s[..., None] * f(cam.horizontal) + t[..., None] * f(cam.vertical) - f(cam.origin)).astype(np.float32)
    d /= np.sqrt((d * d).sum(axis=2, keepdims=True))     # Ray::new normalises the direction (ray.rs:14-20)
    rays = np.concatenate([np.broadcast_to(f(cam.origin), d.shape), d, np.full(d.shape[:2] + (1,), 0.5, np.float32)], axis=2)
    gi, gt = ctx.first_hit(rays.reshape(-1, 7), seed=11)
    wi, wt = z["prim_id"].reshape(-1).astype(np.int32), z["t"].reshape(-1)
    same = gi == wi
    # the rays are recomputed here in numpy; where their last bit differs from the generator's a silhouette
    # pixel may fall on the other side of an edge
    assert same.mean() > 0.995, (~same).sum()
    hit = same & (wi >= 0)
    assert (np.abs(gt[hit] - wt[hit]) / np.abs(wt[hit])).max() < 2e-5


@pytest.mark.parametrize("name", ["cornell", "smoke", "random", "final"])
def test_converged_image_matches_committed_mean(rt, earth, name):
    """Statistical parity against fixed data: tests/golden/converged_<scene>.npz holds the oracle's mean image
    of N samples and the RMSE between the means of its two halves, 2 sigma / sqrt(N). A device render of N
    samples with another seed should differ from the full mean by sqrt(2) sigma / sqrt(N) = 0.71 half_rmse;
    the bar is 1.0 x half_rmse per channel and 1 % of mean luminance."""
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", f"converged_{name}.npz"))
    mean, half, spp = z["mean"], z["half_rmse"], int(z["spp"])
    h, w = mean.shape[:2]
    sc = rt.select_scene(name, earth_rgba=earth if name == "final" else None)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(w / h, name))
    ctx.frame_begin(w, h)
    ctx.render(spp, int(z["depth"]), seed=4242)
    acc = ctx.read_accum()
    g = acc[..., :3] / acc[..., 3:4]
    for c in range(3):
        rmse = math.sqrt(((g[..., c] - mean[..., c]) ** 2).mean())
        assert rmse <= 1.0 * half[c] + 1e-4, (c, rmse, half[c])
    lum = lambda im: (0.2126 * im[..., 0] + 0.7152 * im[..., 1] + 0.0722 * im[..., 2]).mean()
    assert abs(lum(g) - lum(mean)) <= 0.01 * lum(mean)


@pytest.mark.parametrize("name", ["random", "final", "smoke"])
def test_first_hit_half_resolution_hierarchy_scenes(rt, oracle, earth, name):
    """960x540 fixed camera rays through the flattened hierarchies (random: auto-built over 485 spheres;
    final: 400 cuboids + 1000 instanced spheres + media) against the oracle's list / pointer-BVH walk."""
    orc, ctx, _ = make(rt, oracle, earth, name, 960, 540)
    rays = orc.camera_rays(0.5)
    oi, ot = orc.first_hit(rays, seed=5)
    gi, gt = ctx.first_hit(rays, seed=5)
    same = oi == gi
    # an id may differ only where two surfaces are hit at (numerically) the same t: shared cuboid faces
    assert (np.abs(ot[~same] - gt[~same]) <= 1e-5 * np.abs(ot[~same])).all()
    assert same.mean() > 0.9999, (~same).sum()
    hit = same & (oi >= 0)
    assert (np.abs(ot[hit] - gt[hit]) / np.abs(ot[hit])).max() < 1e-5


@pytest.mark.parametrize("w,h", [(37, 19), (8, 8), (33, 9), (2, 2)])
def test_ragged_frame_sizes(rt, oracle, earth, w, h):
    """Frames that are not multiples of the 32x8 workgroup tile (and smaller than one tile): every pixel is
    rendered exactly once per sample, by both kernel sets, and matches the oracle."""
    orc, ctx, _ = make(rt, oracle, earth, "cornell", w, h)
    want, _ = orc.render_sums(8, 50, seed=2)
    for kernel in (rt.KERNEL_FUSED, rt.KERNEL_QUEUE, rt.KERNEL_WAVEFRONT):
        ctx.frame_begin(w, h)
        ctx.render(8, 50, seed=2, kernel=kernel)
        acc = ctx.read_accum()
        assert acc.shape == (h, w, 4) and (acc[..., 3] == 8).all()
        share = np.isclose(acc[..., :3], want, rtol=5e-3, atol=2e-3).all(axis=2).mean()
        observe("ragged_frame", size=[w, h], kernel=int(kernel), close_share=share)
        assert share >= RAGGED_CLOSE_MIN[(w, h)], share
    frame = ctx.resolve_rgba8()
    assert frame.shape == (h, w, 4) and (frame[..., 3] == 255).all()


@pytest.mark.parametrize("name,spp,size", [("cornell", 24, (64, 36)), ("smoke", 8, (64, 36)), ("random", 8, (64, 36)),
                                           ("final", 4, (64, 36)), ("2psp", 8, (37, 19)), ("random", 3, (130, 70)),
                                           ("cornell", 17, (40, 24)), ("random", 33, (24, 16))])   # ragged last chunk
def test_pixel_queue_kernel_equals_fused_kernel(rt, oracle, earth, name, spp, size):
    """The persistent kernel hands (pixel, sample chunk) items to its lanes from a shared counter. Every sample
    is the same path as in the one-pixel-per-lane kernel (equal segment counts); a chunk's samples are added
    in order by one lane and the chunk sums of a pixel in chunk order, so the result does not depend on which
    lane took what and differs from the other kernel's only by the regrouping of the float additions. With one
    sample per call there is one chunk and the sums are equal bit for bit. AUTO picks the persistent
    kernel."""
    w, h = size
    orc, ctx, _ = make(rt, oracle, earth, name, w, h)
    out = {}
    for kernel in (rt.KERNEL_FUSED, rt.KERNEL_QUEUE, rt.KERNEL_AUTO, "queue again"):
        k = rt.KERNEL_QUEUE if kernel == "queue again" else kernel
        ctx.frame_begin(w, h)
        ctx.render(spp, 50, seed=8, kernel=k)
        ctx.render(2, 50, spp_begin=spp, seed=8, kernel=k)     # a second sample range on top
        st = ctx.stats()
        out[kernel] = (ctx.read_accum(), st["segments"], st["kernel_used"])
    assert out[rt.KERNEL_FUSED][2] == rt.KERNEL_FUSED and out[rt.KERNEL_QUEUE][2] == rt.KERNEL_QUEUE
    assert out[rt.KERNEL_AUTO][2] == rt.KERNEL_QUEUE
    np.testing.assert_allclose(out[rt.KERNEL_QUEUE][0], out[rt.KERNEL_FUSED][0], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out[rt.KERNEL_AUTO][0], out[rt.KERNEL_FUSED][0], rtol=2e-6, atol=1e-6)
    assert (out["queue again"][0] == out[rt.KERNEL_QUEUE][0]).all()      # scheduling does not enter the sums
    assert out[rt.KERNEL_QUEUE][1] == out[rt.KERNEL_FUSED][1]
    assert (out[rt.KERNEL_FUSED][0][..., 3] == spp + 2).all() and (out[rt.KERNEL_QUEUE][0][..., 3] == spp + 2).all()
    one = {}
    for kernel in (rt.KERNEL_FUSED, rt.KERNEL_QUEUE):
        ctx.frame_begin(w, h)
        for k in range(3):
            ctx.render(1, 50, spp_begin=k, seed=8, kernel=kernel)
        one[kernel] = ctx.read_accum()
    assert (one[rt.KERNEL_QUEUE] == one[rt.KERNEL_FUSED]).all()


@pytest.mark.parametrize("budget,lanes,node_lanes", [(1, 0, 0), (2, 64, 64), (3, 63, 1), (1000, 1, 63)])
def test_walks_resume_correctly_whatever_cuts_them(budget, lanes, node_lanes):
    """The persistent kernel cuts hierarchy walks into turns (a step budget, a minimum of lanes still walking, a
    minimum of lanes still descending) and resumes them in later iterations. The limits only change when a walk
    pauses, never what it finds: with extreme settings (a turn of one step; every round cut; never cut) the images
    of the hierarchy scenes still equal the one-pixel-per-lane kernel's, which walks without pausing. The limits
    are read from the environment once per process, hence the child process."""
    import os, subprocess, sys
    code = (
        "import numpy as np, ray_tracing_amd as rt\n"
        "for name, spp in (('random', 6), ('final', 3)):\n"
        "    sc = rt.select_scene(name); ctx = rt.Context(0); ctx.scene_upload(sc)\n"
        "    ctx.camera_set(rt.Camera.select_camera(64 / 36, name))\n"
        "    out = []\n"
        "    for k in (rt.KERNEL_FUSED, rt.KERNEL_QUEUE):\n"
        "        ctx.frame_begin(64, 36); ctx.render(spp, 50, seed=9, kernel=k); out.append((ctx.read_accum(), ctx.stats()['segments']))\n"
        "    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-6, atol=1e-6)\n"
        "    assert out[1][1] == out[0][1]\n"
        "print('ok')\n")
    env = dict(os.environ, MIRT_WALK_BUDGET=str(budget), MIRT_WALK_LANES=str(lanes), MIRT_NODE_LANES=str(node_lanes))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def _sphere_scene(rt, centres, radius, entry_kind):
    """A hand-made descriptor: spheres with one Lambertian material in a single world entry of the given kind."""
    C = rt.C
    n = len(centres)
    prims = (rt.mirt_prim * n)()
    for p, c in zip(prims, centres):
        p.kind, p.material, p.radius = 0, 0, radius
        p.center0[0], p.center0[1], p.center0[2] = c
    mats = (rt.mirt_material * 1)()
    texs = (rt.mirt_texture * 1)()
    texs[0].color[0] = texs[0].color[1] = texs[0].color[2] = 0.5
    world = (rt.mirt_entry * 1)()
    world[0].kind, world[0].first_prim, world[0].prim_count, world[0].prims_per_object = entry_kind, 0, n, 1
    inst = (rt.mirt_instance * 1)()
    d = rt.mirt_scene_desc()
    d.prims, d.n_prims = prims, n
    d.world, d.n_world = world, 1
    d.materials, d.n_materials = mats, 1
    d.textures, d.n_textures = texs, 1
    d.instances, d.n_instances = inst, 1
    d.background[0] = d.background[1] = d.background[2] = 1.0
    d._keep = (prims, mats, texs, world, inst)
    return d


@pytest.mark.parametrize("n", [1, 2, 3, 5, 40])
def test_small_hierarchies_equal_lists(rt, n):
    """A hierarchy entry returns what the same primitives return as a plain list, down to trees of a single
    leaf (one or two members) and up to a list long enough to be given a hierarchy automatically."""
    rng = np.random.default_rng(n)
    centres = rng.uniform(-3, 3, (n, 3)).astype(np.float32).tolist()
    o = rng.uniform(-6, 6, (4000, 3)).astype(np.float32)
    d = -o + rng.normal(scale=1.5, size=o.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.zeros((len(o), 1), np.float32)], axis=1).astype(np.float32)
    res = []
    for kind in (0, 1):   # MIRT_ENTRY_LIST, MIRT_ENTRY_BVH
        ctx = rt.Context(0)
        ctx.scene_upload(_sphere_scene(rt, centres, 0.8, kind))
        res.append(ctx.first_hit(rays))
    (li, lt), (bi, bt) = res
    assert (li >= 0).mean() > 0.02
    assert (li == bi).all() and (lt == bt).all()


def test_empty_world_and_zero_samples(rt):
    """An empty world (scene.rs:290 with no objects) returns the background for every ray; zero samples or
    zero depth leave the sums untouched (renderer.rs:100, 198-200)."""
    C = rt.C
    d = rt.mirt_scene_desc()
    d.background[0], d.background[1], d.background[2] = 0.25, 0.5, 0.75
    ctx = rt.Context(0)
    ctx.scene_upload(d)
    ctx.camera_set(rt.Camera.select_camera(2.0, "cornell"))
    ctx.frame_begin(16, 8)
    ctx.render(3, 5, seed=1)
    acc = ctx.read_accum()
    assert np.allclose(acc[..., :3], 3 * np.array([0.25, 0.5, 0.75], dtype=np.float32)) and (acc[..., 3] == 3).all()
    assert ctx.stats()["segments"] == 16 * 8 * 3
    ids, t = ctx.first_hit(np.array([[0, 0, 0, 0, 0, 1, 0.5]], dtype=np.float32))
    assert ids[0] == -1 and np.isinf(t[0])
    ctx.frame_begin(16, 8)
    ctx.render(0, 5, seed=1)
    ctx.render(4, 0, seed=1)            # depth 0: ray_color returns black at once
    acc = ctx.read_accum()
    assert (acc[..., :3] == 0).all() and (acc[..., 3] == 4).all()
    assert C.sizeof(d) > 0


def test_world_box_of_a_small_hierarchy_loses_no_hit(rt, oracle, earth):
    """`final` keeps its thousand small spheres in a hierarchy of their own under a rotation and a translation
    (scene.rs:259-268); the device tests a ray against that hierarchy's padded world box before it changes frames and
    fetches the root, and skips the hierarchy when the ray misses the box. Rays aimed at points on and just around
    the box — the cluster occupies x in about [-112, 116], y in [258, 447], z in [345, 576] — must name the same
    primitive as the oracle's scan of every member (aabb.rs:112-126 has no such test)."""
    orc, ctx, _ = make(rt, oracle, earth, "final")
    rng = np.random.default_rng(11)
    n = 30000
    lo, hi = np.array([-142.0, 228.0, 315.0]), np.array([146.0, 477.0, 606.0])
    target = lo + rng.random((n, 3)) * (hi - lo)
    # a third of the targets snapped onto a face of the (approximate) box, where a ray grazes it
    face = rng.integers(0, 3, n)
    snap = rng.random(n) < 0.33
    side = rng.random(n) < 0.5
    box_lo, box_hi = np.array([-112.0, 258.0, 345.0]), np.array([116.0, 447.0, 576.0])
    for a in range(3):
        m = snap & (face == a)
        target[m, a] = np.where(side[m], box_lo[a], box_hi[a])
    # (origins above the ground boxes: a ray that starts between them meets the coinciding faces of two neighbours at
    # one t, a tie that a hierarchy of another shape than the reference's resolves differently)
    origin = np.array([278.0, 350.0, -300.0]) + (rng.random((n, 3)) - 0.5) * np.array([1200.0, 400.0, 600.0])
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([origin, d, rng.random((n, 1))], axis=1).astype(np.float32)
    oi, ot = orc.first_hit(rays)
    gi, gt = ctx.first_hit(rays)
    cluster = (oi >= 0) & (gi != oi)
    assert not cluster.any(), (int(cluster.sum()), rays[cluster][:3], oi[cluster][:3], gi[cluster][:3])
    hit = oi >= 0
    np.testing.assert_allclose(gt[hit], ot[hit], rtol=1e-5)


def test_second_set_of_noise_tables(rt):
    """The persistent kernel keeps the scene's first set of noise tables (perlin.rs:9-21) in LDS and reads any other set
    from global memory. `2psp` with a second set in front of its own, and its texture pointed at index 1, must render
    the image of the unchanged scene bit for bit — through the other fetch path."""
    w, h = 48, 27
    images = []
    for second in (False, True):
        sc = rt.select_scene("2psp")
        d = sc.desc
        keep = None
        if second:
            assert d.n_perlins == 1
            keep = (rt.mirt_perlin * 2)()
            C.memmove(C.addressof(keep[1]), C.addressof(d.perlins[0]), C.sizeof(rt.mirt_perlin))   # [0] stays all zero
            d.perlins = C.cast(keep, C.POINTER(rt.mirt_perlin))
            d.n_perlins = 2
            moved = 0
            for i in range(d.n_textures):
                if d.textures[i].kind == 2:      # TK_NOISE
                    d.textures[i].perlin = 1
                    moved += 1
            assert moved >= 1
        ctx = rt.Context(0)
        ctx.scene_upload(sc)
        ctx.camera_set(rt.Camera.select_camera(w / h, "2psp"))
        ctx.frame_begin(w, h)
        ctx.render(6, 50, seed=3, kernel=rt.KERNEL_QUEUE)
        st = ctx.stats()
        assert st["device_faults"] == 0
        images.append(ctx.read_accum())
        del keep
    assert np.isfinite(images[0]).all() and images[0][..., :3].std() > 0.01
    assert (images[0] == images[1]).all()


@pytest.mark.parametrize("counters", [1, 2, 32])
def test_item_counters_do_not_enter_the_image(counters):
    """The persistent kernel's tile runs are shared out to 8 or 16 counters, a contiguous share each
    (MIRT_ITEM_COUNTERS overrides; read once per process, hence the child process). With one counter, two, or thirty-two
    — more shares than some frames have tile runs, and a share boundary inside a frame of ragged size — every pixel
    still gets each of its samples exactly once: the image equals the one-pixel-per-lane kernel's."""
    import os, subprocess, sys
    code = (
        "import numpy as np, ray_tracing_amd as rt\n"
        "for name, spp, (w, h) in (('cornell', 40, (37, 19)), ('random', 6, (64, 36)), ('final', 3, (130, 70)), ('cornell', 3, (8, 8))):\n"
        "    sc = rt.select_scene(name); ctx = rt.Context(0); ctx.scene_upload(sc)\n"
        "    ctx.camera_set(rt.Camera.select_camera(w / h, name))\n"
        "    out = []\n"
        "    for k in (rt.KERNEL_FUSED, rt.KERNEL_QUEUE):\n"
        "        ctx.frame_begin(w, h); ctx.render(spp, 50, seed=4, kernel=k); st = ctx.stats(); assert st['device_faults'] == 0\n"
        "        out.append((ctx.read_accum(), st['segments']))\n"
        "    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-6, atol=1e-6)\n"
        "    assert out[1][1] == out[0][1] and (out[1][0][..., 3] == spp).all()\n"
        "print('ok')\n")
    env = dict(os.environ, MIRT_ITEM_COUNTERS=str(counters))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
