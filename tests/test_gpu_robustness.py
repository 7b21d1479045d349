"""GPU tests of the corners the parity suite does not reach: rays parallel to an axis through the hierarchies,
the device's own fault reports (traversal stack, work items that were never stored), frames wider than the
persistent kernel's packed pixel position, the one-call host entry point, and the RCCL exchange step at the
world sizes this box has."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


DEBUG_LIB = os.path.join(ROOT, "ray_tracing_amd", "libmirt_debug.so")   # the build with the test hooks (MIRT_DEBUG_*)


def _run_child(code, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    if any(k.startswith("MIRT_DEBUG_") for k in env):
        assert os.path.exists(DEBUG_LIB), "build ray_tracing_amd/libmirt_debug.so (make -C ray_tracing_amd/csrc)"
        e["MIRT_LIB"] = DEBUG_LIB
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("name", ["random", "final"])
def test_axis_parallel_rays_through_hierarchies(rt, oracle, earth, name):
    """Directions with one or two components exactly +0 or -0 (reciprocal +-inf): the slab test of the hierarchy
    walk must enter every box such a ray really passes through (aabb.rs:43-60 yields (-inf, +inf) for a slab that
    contains the origin), from origins inside and outside the slabs, also through the instanced hierarchy of
    `final`. Ids equal the oracle's list / pointer-BVH scan, t within 1e-5."""
    e = earth if name == "final" else None
    orc = oracle.OracleRenderer(name, 64, 36, earth_rgba=e)
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene(name, earth_rgba=e))
    rng = np.random.default_rng(12)
    n = 6000
    if name == "random":
        o = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        o[:, 1] = rng.uniform(0.05, 3.0, n).astype(np.float32)         # among the small spheres, above the ground
        o[: n // 6] = np.round(o[: n // 6])                             # origins on integer coordinates (box planes)
    else:
        o = (np.array([200.0, 250.0, 200.0]) + rng.uniform(-1, 1, (n, 3)) * 420.0).astype(np.float32)
        o[:, 1] = np.abs(o[:, 1] - 110.0) + 110.0                        # above the ground cuboids (shared faces tie)
        # a quarter unit beside the x planes of the cuboid grid. (Not ON them: a ray that lies in the plane of a rect
        # has t = 0 / 0 there, and the reference's rejecting comparisons, geo.rs:199-204, let a NaN t through as a
        # hit; the device's accepting form calls it a miss. Only a ray constructed for it meets that case.)
        o[: n // 6, 0] = np.round(o[: n // 6, 0] / 100.0) * 100.0 + 0.25
    d = np.zeros((n, 3), dtype=np.float32)
    kind = rng.integers(0, 3, n)
    axis = rng.integers(0, 3, n)
    sign = np.where(rng.integers(0, 2, n) == 0, -1.0, 1.0).astype(np.float32)
    for i in range(n):
        a = axis[i]
        if kind[i] == 0:                        # along one axis: two zero components
            d[i, a] = sign[i]
        else:                                   # in an axis plane: one zero component
            v = rng.normal(size=2)
            v /= np.linalg.norm(v)
            d[i, (a + 1) % 3], d[i, (a + 2) % 3] = v
    negzero = rng.integers(0, 2, (n, 3)) == 1
    d = np.where((d == 0) & negzero, np.float32(-0.0), d).astype(np.float32)
    down = rng.integers(0, 4, n) == 0           # a good share straight down at the ground / the boxes
    d[down] = np.array([0.0, -1.0, -0.0], dtype=np.float32)
    rays = np.concatenate([o, d, rng.uniform(0, 1, (n, 1)).astype(np.float32)], axis=1).astype(np.float32)
    assert ((rays[:, 3:6] == 0).sum(axis=1) >= 1).all()
    oi, ot = orc.first_hit(rays, seed=3)
    gi, gt = ctx.first_hit(rays, seed=3)
    assert (oi >= 0).mean() > 0.3
    same = oi == gi
    # an id may differ only where two surfaces are hit at (numerically) the same t
    assert (np.abs(ot[~same] - gt[~same]) <= 1e-5 * np.abs(ot[~same])).all(), (name, int((~same).sum()))
    assert same.mean() > 0.998, (name, int((~same).sum()))
    hit = same & (oi >= 0)
    assert (np.abs(ot[hit] - gt[hit]) / np.abs(ot[hit])).max() < 1e-5
    # the same rays as paths: the render loop's resumable walk sees them too
    oc = orc.ray_color(rays[:1500], sample=0, max_depth=6, seed=2)
    gc = ctx.ray_color(rays[:1500], sample=0, max_depth=6, seed=2)
    fix = lambda c: np.where(np.isinf(c), 1.0, np.where(np.isnan(c), 0.0, c))
    bad = ~np.isclose(fix(oc), fix(gc), rtol=2e-3, atol=1e-4)
    assert bad.any(axis=1).mean() < 0.01


def test_traversal_stack_overflow_is_reported(rt):
    """A walk that finds its stack full drops a subtree; the kernels raise MIRT_FAULT_STACK_OVERFLOW instead of
    losing the hit silently: visible in mirt_stats.device_faults, and the blocking resolve fails with
    MIRT_ERR_DEVICE. A normal upload never gets there (the builder's depth is checked against the stack at upload:
    bvh_deepest_leaf - 1 <= 32), so the test lowers the limit through the debug variable."""
    st = rt.Context(0)
    st.scene_upload(rt.select_scene("random"))
    st.camera_set(rt.Camera.select_camera(64 / 36, "random"))
    st.frame_begin(64, 36)
    st.render(4, 50, seed=1)
    s = st.stats()
    assert s["device_faults"] == 0 and 2 <= s["bvh_deepest_leaf"] <= 33
    st.resolve_rgba8()
    code = (
        "import numpy as np, ray_tracing_amd as rt\n"
        "ctx = rt.Context(0); ctx.scene_upload(rt.select_scene('random'))\n"
        "ctx.camera_set(rt.Camera.select_camera(64 / 36, 'random'))\n"
        "for k in (rt.KERNEL_QUEUE, rt.KERNEL_FUSED):\n"
        "    ctx.frame_begin(64, 36); ctx.render(4, 50, seed=1, kernel=k)\n"
        "    assert ctx.stats()['device_faults'] & 1, ctx.stats()\n"
        "    try:\n"
        "        ctx.resolve_rgba8(); raise SystemExit('resolve did not fail')\n"
        "    except rt.MirtError as e:\n"
        "        assert 'traversal stack' in str(e), e\n"
        "ctx.frame_begin(64, 36)\n"
        "assert ctx.stats()['device_faults'] == 0\n"      # a new frame starts clean
        "print('ok')\n")
    _run_child(code, MIRT_DEBUG_STACK_LIMIT=2)


def test_dropped_work_item_is_reported(rt):
    """The persistent kernel stores one slot per (pixel, sample chunk) item and a second pass adds a pixel's slots.
    A slot that was not stored by this launch (here: the debug variable makes one lane skip its store) must not pass
    off an earlier launch's sums: its samples are not counted, and MIRT_FAULT_STALE_CHUNK is raised."""
    code = (
        "import numpy as np, ray_tracing_amd as rt\n"
        "ctx = rt.Context(0); ctx.scene_upload(rt.select_scene('cornell'))\n"
        "ctx.camera_set(rt.Camera.select_camera(64 / 36, 'cornell'))\n"
        "for spp, short in ((32, 2), (1, 1)):\n"            # 16 chunks of 2 samples; one chunk straight into the sums
        "    for rep in range(2):\n"                          # the second launch finds the first one's slots in place
        "        ctx.frame_begin(64, 36); ctx.render(spp, 50, seed=1, kernel=rt.KERNEL_QUEUE)\n"
        "        w = ctx.read_accum()[..., 3].reshape(-1)\n"
        "        bad = np.flatnonzero(w != spp)\n"
        "        assert list(bad) == [777] and w[777] == spp - short, (spp, rep, bad, w[777])\n"
        "        if spp > 1: assert ctx.stats()['device_faults'] & 2, ctx.stats()\n"
        "print('ok')\n")
    _run_child(code, MIRT_DEBUG_DROP_PIXEL=777)


def test_frames_wider_than_65535_take_the_one_pixel_per_lane_kernel(rt, oracle):
    """The persistent kernel packs a pixel position into 16 + 16 bits; a 70000x2 frame is rendered by the other
    kernel (whatever the caller selected) and every pixel gets its samples."""
    w, h = 70000, 2
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell"))
    ctx.camera_set(rt.Camera.select_camera(16 / 9, "cornell"))
    ctx.frame_begin(w, h)
    ctx.render(2, 50, seed=3, kernel=rt.KERNEL_QUEUE)
    acc = ctx.read_accum()
    st = ctx.stats()
    assert st["kernel_used"] == rt.KERNEL_FUSED and (acc[..., 3] == 2).all()
    ctx.frame_begin(w, h)
    ctx.render(2, 50, seed=3, kernel=rt.KERNEL_FUSED)
    assert (ctx.read_accum() == acc).all()
    assert acc[..., :3].sum() > 0 and np.isfinite(acc).all()


def test_auto_picks_the_cheaper_loop_for_a_lone_primitive(rt, earth):
    """`earth` is one textured sphere without lights: AUTO renders it with one pixel per lane (nothing for the item
    queue to even out); any scene with more in it goes through the queue."""
    for name, want in (("earth", rt.KERNEL_FUSED), ("2sp", rt.KERNEL_QUEUE), ("cornell", rt.KERNEL_QUEUE)):
        ctx = rt.Context(0)
        ctx.scene_upload(rt.select_scene(name, earth_rgba=earth if name == "earth" else None))
        ctx.camera_set(rt.Camera.select_camera(64 / 36, name))
        ctx.frame_begin(64, 36)
        ctx.render(4, 50, seed=1)
        assert ctx.stats()["kernel_used"] == want, name


def test_host_draw_entry_point(rt, oracle):
    """mirt_host_draw: select_scene + select_camera + Renderer::new + multisample / max_depth + draw (cli.rs:25-30)
    in one C call, against the oracle's draw."""
    w, h, spp = 64, 36, 16
    frame = np.zeros((h, w, 4), dtype=np.uint8)
    rc = rt.lib().mirt_host_draw(0, b"cornell", w, h, spp, 50, 9, 1, 11, None, 0, 0, frame.ctypes.data_as(rt.C.c_void_p), frame.size)
    assert rc == 0, rt.lib().mirt_host_last_error()
    want = oracle.OracleRenderer("cornell", w, h).draw(spp, 50, seed=9)
    diff = np.abs(frame.astype(np.int32) - want.astype(np.int32))
    assert (frame[..., 3] == 255).all() and (diff <= 1).mean() > 0.98 and diff.mean() < 0.5
    rc = rt.lib().mirt_host_draw(0, b"cornell", w, h, spp, 50, 9, 1, 11, None, 0, 0, frame.ctypes.data_as(rt.C.c_void_p), frame.size - 4)
    assert rc != 0 and b"frame" in rt.lib().mirt_host_last_error()
    # an unknown name is the two-spheres scene, as in scene.rs:21 (`"2sp" | _ => two_spheres(name)`)
    other = np.zeros_like(frame)
    assert rt.lib().mirt_host_draw(0, b"no such scene", w, h, spp, 50, 9, 1, 11, None, 0, 0, other.ctypes.data_as(rt.C.c_void_p), other.size) == 0
    assert rt.lib().mirt_host_draw(0, b"2sp", w, h, spp, 50, 9, 1, 11, None, 0, 0, frame.ctypes.data_as(rt.C.c_void_p), frame.size) == 0
    assert (other == frame).all()


def _device_count(rt):
    """HIP devices this process sees (asked of the library itself: a context on a device that does not exist fails)."""
    n = 1
    while n < 16:
        try:
            rt.Context(n).close()
        except rt.MirtError:
            break
        n += 1
    return n


def _render_range(rt, ctx, w, h, begin, count, seed):
    ctx.frame_begin(w, h)
    ctx.render(count, 50, spp_begin=begin, seed=seed)


def test_rccl_reduce_single_process(rt):
    """mirt_reduce over the contexts of one process: with one context the sum is the buffer itself (it still goes
    through ncclCommInitAll + ncclReduce); with two devices, two half ranges summed onto the root equal the full
    range rendered by one context up to the grouping of the float additions."""
    w, h, spp = 96, 54, 12
    one = rt.Context(0)
    one.scene_upload(rt.select_scene("cornell"))
    one.camera_set(rt.Camera.select_camera(w / h, "cornell"))
    _render_range(rt, one, w, h, 0, spp, 5)
    full = one.read_accum()
    rt.reduce([one], root=0)
    one.synchronize()
    assert one.comm_ranks() == 1
    assert (one.read_accum() == full).all()
    rt.reduce([one], root=0)                      # the communicator is kept
    one.synchronize()
    assert (one.read_accum() == full).all()
    with pytest.raises(rt.MirtError, match="root"):
        rt.reduce([one], root=1)
    if _device_count(rt) >= 2:
        ctxs = [rt.Context(i) for i in range(2)]
        for i, c in enumerate(ctxs):
            c.scene_upload(rt.select_scene("cornell"))
            c.camera_set(rt.Camera.select_camera(w / h, "cornell"))
            _render_range(rt, c, w, h, i * spp // 2, spp // 2, 5)
        rt.reduce(ctxs, root=0)
        ctxs[0].synchronize()
        np.testing.assert_allclose(ctxs[0].read_accum(), full, rtol=1e-5, atol=1e-5)


def test_rccl_reduce_one_process_per_gpu(rt):
    """The multi-process form (unique id -> init_rank -> reduce) at the world sizes this box offers: world 1 in
    this process; world 2 as two child processes under torch.distributed.run when two GPUs are present."""
    w, h, spp = 96, 54, 12
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell"))
    ctx.camera_set(rt.Camera.select_camera(w / h, "cornell"))
    with pytest.raises(rt.MirtError, match="init_rank"):
        ctx.comm_reduce(0)
    ctx.comm_init_rank(1, 0, rt.comm_unique_id())
    _render_range(rt, ctx, w, h, 0, spp, 5)
    full = ctx.read_accum()
    ctx.comm_reduce(0)
    ctx.synchronize()
    assert ctx.comm_ranks() == 1 and (ctx.read_accum() == full).all()
    if _device_count(rt) >= 2:
        out = os.path.join(ROOT, "gpurun_out", "reduce2.npy")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        code = (
            "import os, numpy as np, torch.distributed as dist, ray_tracing_amd as rt\n"
            "from ray_tracing_amd.distributed import join_comm, sample_range\n"
            "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            "dist.init_process_group('gloo')\n"
            "ctx = rt.Context(int(os.environ['LOCAL_RANK'])); ctx.scene_upload(rt.select_scene('cornell'))\n"
            f"ctx.camera_set(rt.Camera.select_camera({w} / {h}, 'cornell')); join_comm(ctx, rank, world)\n"
            f"b, c = sample_range({spp}, world, rank); ctx.frame_begin({w}, {h}); ctx.render(c, 50, spp_begin=b, seed=5)\n"
            "ctx.comm_reduce(0); ctx.synchronize()\n"
            f"if rank == 0: np.save(r'{out}', ctx.read_accum())\n"
            "dist.barrier(); dist.destroy_process_group()\n")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), "--no-python", sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=110)
        assert r.returncode == 0, r.stdout + r.stderr
        np.testing.assert_allclose(np.load(out), full, rtol=1e-5, atol=1e-5)
