"""The pooled form of the hierarchy walk (MIRT_KERNEL_POOL: the walks of a workgroup's 256 lanes advance together in
LDS, one step per ray and round, node steps and leaf tests in separate lists) against the lane-by-lane form. Every ray
takes the same steps of BVHNode::hit (aabb.rs:112-126) in the same order on its own stack, whichever lane executes
them, so the two forms must give the same image bit for bit and count the same segments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,W,H,spp,depth,kw", [
    ("random", 200, 120, 4, 50, {}),
    ("final", 200, 120, 4, 50, {}),
    ("random", 333, 77, 1, 10, {"random_num": 40}),
    ("random", 1920, 1080, 2, 50, {}),
    ("final", 1920, 1080, 2, 50, {}),
])
def test_pooled_walk_equals_lane_walk(rt, name, W, H, spp, depth, kw):
    """Small frames (fewer work items than lanes: waves that only help), a frame whose edge tiles are ragged, a deeper
    hierarchy (random_num = 40: 6 500 spheres), and full frames of the two benchmark scenes with hierarchies, `final`
    with two of them (the second under a rotation and a translation, started by whichever lane finds the first
    exhausted)."""
    sc = rt.select_scene(name, **kw)
    ctx = rt.Context(0)
    ctx.scene_upload(sc)
    ctx.camera_set(rt.Camera.select_camera(W / H, name))
    ctx.frame_begin(W, H)
    ctx.render(spp, depth, seed=3, jitter=spp != 1, kernel=rt.KERNEL_QUEUE)
    ctx.synchronize()
    lane = ctx.read_accum()
    st0 = ctx.stats()
    ctx.frame_begin(W, H)
    ctx.render(spp, depth, seed=3, jitter=spp != 1, kernel=rt.KERNEL_QUEUE | rt.KERNEL_POOL)
    ctx.synchronize()
    pool = ctx.read_accum()
    st1 = ctx.stats()
    assert st1["device_faults"] == 0
    assert st1["segments"] == st0["segments"]
    assert (pool[..., 3] == spp).all()
    assert (pool == lane).all(), float(np.abs(pool - lane).max())
