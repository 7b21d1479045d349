"""CPU checks that pin the oracle: Philox known answers, the analytic Cornell first-hit table
(tests/golden/cornell_first_hit_kat.json, derived from the reference's constants) and the numeric
properties of the reference's samplers (src/rand_gen.rs, src/pdf.rs, src/onb.rs, src/geo.rs)."""
import math

import numpy as np
import pytest

from conftest import kat_rays

# Random123 kat_vectors, philox4x32-10
PHILOX_KAT = [
    ([0, 0], [0, 0, 0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 2, [0xffffffff] * 4, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers(oracle):
    for key, ctr, want in PHILOX_KAT:
        assert [int(x) for x in oracle.philox(key, ctr)] == want


def test_cornell_camera_matches_analytic(oracle, kat):
    r = oracle.OracleRenderer("cornell", 1920, 1080)
    cam = r.camera()
    c = kat["camera"]
    want = np.array(c["origin"] + c["lower_left_corner"] + c["horizontal"] + c["vertical"])
    np.testing.assert_allclose(cam[:12], want, rtol=2e-7, atol=1e-5)


def test_cornell_first_hit_known_answers(oracle, kat):
    r = oracle.OracleRenderer("cornell", 1920, 1080)
    ids, t = r.first_hit(kat_rays(kat))
    for row, i, tt in zip(kat["rays"], ids, t):
        assert i == row["prim_id"], row
        if row["prim_id"] >= 0:
            assert abs(tt - row["hit_t"]) / row["hit_t"] < 1e-5, row
        else:
            assert np.isinf(tt)


def test_cosine_lobe_is_truncated(oracle, kat):
    d = oracle.sampler(0, 200000, seed=3)
    assert d[:, 2].min() >= math.sqrt(0.1) - 1e-6               # rand_gen.rs:82: r2 in [0, 0.9)
    assert abs((d[:, 2] ** 2).mean() - kat["sampler"]["cosine_Ez2"]) < 2e-3
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)


def test_rejection_samplers(oracle):
    s = oracle.sampler(1, 100000, seed=1)
    assert (np.einsum("ij,ij->i", s, s) < 1).all()
    assert np.abs(s.mean(axis=0)).max() < 0.01
    assert abs((np.linalg.norm(s, axis=1) ** 3).mean() - 0.5) < 0.01  # uniform in the ball
    d = oracle.sampler(2, 100000, seed=2)
    assert (d[:, 2] == 0).all() and (d[:, 0] ** 2 + d[:, 1] ** 2 < 1).all()
    u = oracle.sampler(4, 100000, seed=5)[:, 0]
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.005
    assert np.all(u * 2 ** 24 == np.floor(u * 2 ** 24))             # 24-bit mantissa grid


def test_cosine_pdf_value(oracle, kat):
    assert oracle.cosine_pdf_value([0, 1, 0], [0, -1, 0]) == np.float32(1e-4)   # pdf.rs:27
    assert oracle.cosine_pdf_value([0, 1, 0], [1, 0, 0]) == np.float32(1e-4)
    assert abs(oracle.cosine_pdf_value([0, 1, 0], [0, 1, 0]) - 1 / math.pi) < 1e-7


def test_onb_is_not_orthonormal(oracle):
    # onb.rs:38-49: |u| = |v| = sin(angle(w, a)); exact only for w perpendicular to a
    m = oracle.onb([0, 1, 0])
    np.testing.assert_allclose(np.linalg.norm(m, axis=1), [1, 1, 1], atol=1e-7)
    w = np.array([0.6, 0.8, 0.0], dtype=np.float32)
    m = oracle.onb(w)
    np.testing.assert_allclose(np.linalg.norm(m[:2], axis=1), [0.8, 0.8], atol=1e-6)
    w = np.array([0.95, math.sqrt(1 - 0.95 ** 2), 0.0], dtype=np.float32)   # |w.x| > 0.9 -> a = y
    m = oracle.onb(w)
    np.testing.assert_allclose(np.linalg.norm(m[:2], axis=1), [0.95, 0.95], atol=1e-6)


def test_light_list_pdf(oracle, kat):
    r = oracle.OracleRenderer("cornell", 64, 36)
    origin = np.array([190.0, 0.0, 100.0])
    to_sphere = np.array([190.0, 90.0, 190.0]) - origin
    p = r.lights_pdf(origin, to_sphere / np.linalg.norm(to_sphere))
    # mean over {rect (missed: 0), glass sphere} — ray.rs:149-151
    assert abs(p - 0.5 * kat["sampler"]["glass_sphere_pdf_from_floor_190_0_100"]) < 1e-5
    # rect: t^2 |v|^2 / (|v.n| area), geo.rs:223-233, straight up from under the light centre
    o2 = np.array([278.0, 0.0, 279.5])
    p2 = r.lights_pdf(o2, [0, 1, 0])
    assert abs(p2 - 0.5 * 554.0 ** 2 / kat["sampler"]["light_rect_area"]) / p2 < 1e-5
    # non-unit direction is used as is (|v|^2 and |v.n| scale, t shrinks)
    p3 = r.lights_pdf(o2, [0, 2, 0])
    assert abs(p3 - 0.5 * (277.0 ** 2 * 4) / (2 * kat["sampler"]["light_rect_area"])) / p3 < 1e-5
    # inside the light sphere the pdf is NaN (geo.rs:138)
    assert math.isnan(r.lights_pdf([190.0, 90.0, 190.0], [0, 0, 1]))


def test_light_list_random(oracle):
    r = oracle.OracleRenderer("cornell", 64, 36)
    origin = np.array([278.0, 100.0, 278.0], dtype=np.float32)
    d = r.lights_random(origin, 20000, seed=9)
    # half of the picks go to each member (ray.rs:152-154); rect picks hit the plane y = 554 inside the light
    t = (554.0 - origin[1]) / d[:, 1]
    px, pz = origin[0] + t * d[:, 0], origin[2] + t * d[:, 2]
    on_rect = (d[:, 1] > 0) & (px >= 213 - 1e-2) & (px <= 343 + 1e-2) & (pz >= 227 - 1e-2) & (pz <= 332 + 1e-2)
    assert 0.47 < on_rect.mean() < 0.53


def test_float_to_rgb(oracle):
    # renderer.rs:135-137
    assert oracle.float_to_rgb(-1.0) == 0 and oracle.float_to_rgb(0.0) == 0
    assert oracle.float_to_rgb(1.0) == int(math.sqrt(np.float32(0.999)) * 256) == 255
    assert oracle.float_to_rgb(0.25) == 128
    assert oracle.float_to_rgb(float("nan")) == 0


def test_draw_layout_and_sanitise(oracle):
    # renderer.rs:97: frame row 0 is the TOP of the image; Cornell's light is near the top
    r = oracle.OracleRenderer("cornell", 64, 36)
    frame = r.draw(spp=8, max_depth=6, seed=1)
    assert frame.shape == (36, 64, 4) and (frame[..., 3] == 255).all()
    lum = frame[..., :3].astype(np.float32).mean(axis=2)
    assert lum[:6].max() == lum.max() and lum.max() >= 250       # emitter row band is saturated
    sums, cnt = r.render_sums(8, 6, seed=1)
    assert np.isfinite(sums).all() and cnt[0] > 64 * 36 * 8
    # renderer row y is frame row H-1-y
    want = np.array([[oracle.float_to_rgb(v / 8) for v in px] for px in sums[::-1].reshape(-1, 3)], dtype=np.uint8)
    assert (want.reshape(36, 64, 3) == frame[..., :3]).all()


def test_sample_range_split_is_exact(oracle):
    # samples are addressed by their global index, so two half-ranges equal one full range
    r = oracle.OracleRenderer("cornell", 32, 18)
    full, _ = r.render_sums(6, 8, seed=2)
    a, _ = r.render_sums(3, 8, seed=2, spp_begin=0)
    b, _ = r.render_sums(3, 8, seed=2, spp_begin=3)
    np.testing.assert_allclose(a + b, full, rtol=1e-5, atol=1e-6)
    c, _ = r.render_sums(6, 8, seed=2, threads=1)
    assert (c == full).all()                                        # independent of the thread count


def test_all_scenes_build_and_render(oracle, earth):
    for name, n_ids in [("random", None), ("2sp", 2), ("2psp", 2), ("earth", 1), ("simplelight", 3), ("cornell", 13),
                        ("smoke", 20), ("final", 2400 + 6 + 1000 + 2 + 2)]:
        r = oracle.OracleRenderer(name, 24, 16, earth_rgba=earth if name in ("earth", "final") else None)
        if n_ids is not None:
            assert r.num_leaf_ids == n_ids, name
        sums, _ = r.render_sums(2, 5, seed=0)
        assert np.isfinite(sums).all() and sums.sum() > 0, name


@pytest.mark.parametrize("name", ["cornell", "smoke", "random", "final", "2sp", "2psp", "earth", "simplelight"])
def test_oracle_reproduces_committed_first_hit_grid(oracle, earth, name):
    """tests/golden/first_hit_<scene>.npz freezes the oracle: a change to it (or to the compiler flags it is
    built with) that moves a first hit shows up here before it can move the device tests' target."""
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", f"first_hit_{name}.npz"))
    orc = oracle.OracleRenderer(name, 64, 36, earth_rgba=earth if name in ("earth", "final") else None)
    ids, t = orc.first_hit(orc.camera_rays(0.5), seed=11)
    assert (ids == z["prim_id"].reshape(-1)).all()
    hit = ids >= 0
    assert (t[hit] == z["t"].reshape(-1)[hit]).all()
