"""CPU checks of the host side: the C-ABI library loads and exports every declared symbol, and the
host scene / camera builders (mirror of src/scene.rs, src/camera.rs) declare exactly the scenes the
oracle's independent restatement declares. No compute entry point is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["random", "2sp", "2psp", "earth", "simplelight", "cornell", "smoke", "final"]


def test_library_exports_every_declared_symbol(rt):
    declared = set()
    for hdr in ("mirt.h", "mirt_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared |= set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", text))
    assert declared == set(rt.EXPORTS)
    L = ctypes.CDLL(rt.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_struct_sizes_match_header(rt):
    # the ctypes mirrors must have the C layout of include/mirt.h
    assert ctypes.sizeof(rt.mirt_prim) == 88
    assert ctypes.sizeof(rt.mirt_camera) == 84
    assert ctypes.sizeof(rt.mirt_render_params) == 32
    assert ctypes.sizeof(rt.mirt_entry) == 20
    assert ctypes.sizeof(rt.mirt_instance) == 20
    assert ctypes.sizeof(rt.mirt_material) == 24
    assert ctypes.sizeof(rt.mirt_texture) == 40
    assert ctypes.sizeof(rt.mirt_perlin) == 256 * 12 + 768
    assert ctypes.sizeof(rt.mirt_stats) == 3 * 8 + 2 * 4 + 8 + 3 * 8 + 6 * 4 + 8 + 8
    assert rt.mirt_stats.device_faults.offset == 76 and rt.mirt_stats.bvh_box_tests.offset == 88 and rt.mirt_stats.reduce_ms.offset == 96
    assert rt.mirt_stats.device_faults.offset == 76


def test_host_philox_matches_oracle(rt, oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        key = rng.integers(0, 2 ** 32, 2, dtype=np.uint64).astype(np.uint32)
        ctr = rng.integers(0, 2 ** 32, 4, dtype=np.uint64).astype(np.uint32)
        assert (rt.philox4x32_10(key, ctr) == oracle.philox(key, ctr)).all()


@pytest.mark.parametrize("name", SCENES)
def test_camera_presets_match_oracle(rt, oracle, name):
    for w, h in ((1920, 1080), (400, 225), (192, 108)):
        cam = rt.Camera.select_camera(w / h, name)
        o = oracle.OracleRenderer("cornell" if name == "cornell" else name, w, h).camera() if name not in ("earth", "final") else \
            oracle.OracleRenderer(name, w, h).camera()
        assert (cam.as_array() == o).all(), (name, w, h)


def flat_table(desc):
    """The flattened scene in the oracle's dump format (one row of 12 floats per leaf id)."""
    out = np.zeros((desc.n_prims, 12), dtype=np.float32)
    for i in range(desc.n_prims):
        p = desc.prims[i]
        if p.kind == 0:
            out[i, :11] = [0, *p.center0, p.radius, *p.center1, 1.0 if p.flags & 2 else 0.0, p.time0, p.time1]
        elif p.kind == 1:
            out[i, :8] = [1, p.axis, p.k, p.lo[0], p.lo[1], p.hi[0], p.hi[1], 1.0 if p.flags & 1 else 0.0]
        else:
            out[i, :2] = [2, p.neg_inv_density]
    return out


@pytest.mark.parametrize("name", SCENES)
def test_flattened_scene_equals_oracle_scene(rt, oracle, earth, name):
    e = earth if name in ("earth", "final") else None
    sc = rt.select_scene(name, scene_seed=5, earth_rgba=e)
    orc = oracle.OracleRenderer(name, 64, 36, scene_seed=5, earth_rgba=e)
    d = sc.desc
    assert d.n_prims == orc.num_leaf_ids
    got, want = flat_table(d), orc.dump_prims()
    assert (got == want).all(), np.argwhere(got != want)[:5]


def test_cornell_flattening(rt):
    d = rt.select_scene("cornell").desc
    assert d.n_prims == 13 and d.n_world == 1 and d.n_lights == 2
    assert [d.lights[i] for i in range(2)] == [11, 12]
    w = d.world[0]
    assert (w.kind, w.first_prim, w.prim_count) == (0, 0, 13)
    assert d.n_instances == 2
    inst = d.instances[1]
    assert abs(inst.sin_theta - np.sin(np.float32(15.0) / np.float32(180.0) * np.float32(np.pi))) < 1e-7
    assert list(inst.offset) == [265.0, 0.0, 295.0]
    assert [d.prims[i].instance for i in range(13)] == [0] * 5 + [1] * 6 + [0, 0]
    assert d.prims[11].flags & 1 and not d.prims[12].flags & 1      # FlipFace on the light only
    kinds = [d.materials[d.prims[i].material].kind for i in range(13)]
    assert kinds == [0] * 11 + [3, 2]
    # shared materials are emitted once: green, red, white, light, glass
    assert d.n_materials == 5


def test_final_scene_flattening(rt, earth):
    d = rt.select_scene("final", earth_rgba=earth).desc
    kinds = [(d.world[i].kind, d.world[i].prim_count, d.world[i].prims_per_object, d.world[i].instance) for i in range(d.n_world)]
    assert kinds == [(1, 2400, 6, 0), (0, 6, 1, 0), (1, 1000, 1, 1), (0, 2, 1, 0)]
    assert d.n_perlins == 1 and d.n_images == 1
    assert (d.images[0].width, d.images[0].height) == (1024, 512)
    # media: boundary spheres live after the world primitives
    med = [d.prims[i] for i in range(d.n_prims) if d.prims[i].kind == 2]
    assert [(m.boundary_first, m.boundary_count) for m in med] == [(3408, 1), (3409, 1)]
    assert abs(med[1].neg_inv_density + 1e5) < 1e-1
    # vertical flip of texture.rs:84: library row 0 is the file's bottom row
    row0 = np.ctypeslib.as_array(d.images[0].rgba, shape=(512, 1024, 4))[0]
    assert (row0 == earth[-1]).all()


def test_scaled_random_scene_size(rt):
    d = rt.select_scene("random", random_num=40).desc      # 80*80 grid
    assert 6000 < d.n_prims <= 6404
    assert d.n_lights == 1 and d.lights[0] == d.n_prims - 1


def test_create_without_gpu_reports_cleanly(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.MirtError):
        rt.Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ray_tracing_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and "import oracle" not in text, os.path.join(dirpath, f)


def test_png_writer_round_trips(rt, tmp_path):
    """image::save_buffer(.., Rgba8) of cli.rs:31-38: the file decodes to the frame that was written."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for shape in ((1, 1, 4), (7, 13, 4), (108, 192, 4)):
        frame = rng.integers(0, 256, shape, dtype=np.uint8)
        path = tmp_path / f"f{shape[0]}.png"
        rt.save_png(str(path), frame)
        im = Image.open(path)
        assert im.mode == "RGBA" and im.size == (shape[1], shape[0])
        assert (np.asarray(im) == frame).all()
    with pytest.raises(ValueError):
        rt.save_png(str(tmp_path / "bad.png"), np.zeros((4, 4, 3), dtype=np.uint8))
    with pytest.raises(rt.MirtError):
        rt.save_png(str(tmp_path / "no_such_dir" / "x.png"), np.zeros((2, 2, 4), dtype=np.uint8))


def test_cli_flags_and_defaults_follow_the_reference():
    from ray_tracing_amd import cli
    a = cli.parse([])
    assert (a.max_depth, a.sample_count, a.down_scale, a.scene) == (50, 100, 10, "random")     # cli.rs:11-18
    a = cli.parse(["-m", "5", "-s", "1024", "-d", "1", "--scene", "cornell"])
    assert (a.max_depth, a.sample_count, a.down_scale, a.scene) == (5, 1024, 1, "cornell")
    assert (cli.WIDTH, cli.HEIGHT) == (1920, 1080)                                               # lib.rs:42-43
    with pytest.raises(SystemExit):
        cli.parse(["-d", "0"])
