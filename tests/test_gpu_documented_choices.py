"""The two places where the device knowingly answers differently from the reference, each pinned to the documented
choice. Both concern rays that only a constructed input produces (a set of measure zero among camera and scattered
rays); the tests exist so that the choice cannot change unnoticed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ray_in_the_plane_of_a_rect_misses_it(rt, oracle):
    """A ray that lies exactly IN the plane of an axis-aligned rect has t = (k - origin) / direction = 0 / 0 = NaN
    there. The reference's rect test rejects with `t < t_min || t > t_max` and four more comparisons of that kind
    (geo.rs:199-204); every one of them is false for NaN, so the reference reports a hit with t = NaN, and from then
    on `closest` is NaN and every later list member whose test runs is accepted as well (ray.rs:127-129) — the
    oracle restates exactly that and names the last such member. The device tests in accepting form
    (t >= t_min & t <= t_max & ...): NaN is a miss, and the ray goes on to the surface it really meets.
    Cornell box: origin on the floor (y = 0, primitive 2; the tall box's bottom face, primitive 7, lies in the same
    plane), direction +x: the wall x = 555 (primitive 0) at t = 455."""
    rays = np.array([[100, 0, 100, 1, 0, 0, 0.5]], dtype=np.float32)
    ref_id, ref_t = oracle.OracleRenderer("cornell", 64, 36).first_hit(rays)
    assert np.isnan(ref_t[0]) and ref_id[0] in (2, 7)              # the reference's answer: a NaN hit of an in-plane rect
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell"))
    gi, gt = ctx.first_hit(rays)
    assert gi[0] == 0 and abs(gt[0] - 455.0) <= 1e-3, (gi, gt)     # the documented choice: the in-plane rects are missed
    # the same ray through the whole estimator: a finite sample (the reference's NaN path would be zeroed, renderer.rs:107)
    assert np.isfinite(ctx.ray_color(rays, sample=0, max_depth=50, seed=1)).all()


def _two_entry_scene(rt, hierarchy_spheres, list_sphere):
    """World = [BVHNode over `hierarchy_spheres`, HittableList of `list_sphere`], in this order."""
    spheres = list(hierarchy_spheres) + [list_sphere]
    ps = (rt.mirt_prim * len(spheres))()
    for p, (c, r) in zip(ps, spheres):
        p.kind, p.material, p.radius = 0, 0, r
        for j in range(3):
            p.center0[j] = c[j]
    mats = (rt.mirt_material * 1)()
    texs = (rt.mirt_texture * 1)()
    for j in range(3):
        texs[0].color[j] = 0.5
    world = (rt.mirt_entry * 2)()
    world[0].kind, world[0].first_prim, world[0].prim_count, world[0].prims_per_object = 1, 0, len(hierarchy_spheres), 1
    world[1].kind, world[1].first_prim, world[1].prim_count, world[1].prims_per_object = 0, len(hierarchy_spheres), 1, 1
    inst = (rt.mirt_instance * 1)()
    d = rt.mirt_scene_desc()
    d.prims, d.n_prims = ps, len(spheres)
    d.world, d.n_world = world, 2
    d.materials, d.n_materials = mats, 1
    d.textures, d.n_textures = texs, 1
    d.instances, d.n_instances = inst, 1
    d._keep = (ps, mats, texs, world, inst)
    return d


def test_exact_tie_between_list_and_hierarchy_goes_to_the_hierarchy_member(rt):
    """HittableList::hit visits the world's members in order and accepts `t <= closest`, so of two surfaces hit at
    exactly the same t the LATER world member wins (ray.rs:127-129 with the inclusive range of geo.rs:100). The device
    scans all list entries first and walks the hierarchy entries after them, so that the walks start with the
    lists' closest hit as far bound; an exact tie between a list member and a hierarchy member therefore goes to the
    hierarchy member even when the hierarchy entry stands BEFORE the list in the world — where the reference names
    the list member. World here: a hierarchy over three spheres (primitives 0-2), then a list holding a copy of
    sphere 0 (primitive 3): every ray that hits sphere 0 hits primitive 3 at the same t, bit for bit (same
    arithmetic on the same numbers). Reference: 3. Device, documented: 0. With the order reversed — list first — both
    name the hierarchy member, and rays that hit one of the other spheres are not affected."""
    twin = ((0.0, 0.0, 0.0), 1.0)
    others = [((3.0, 0.0, 0.0), 1.0), ((-3.0, 0.5, 0.0), 0.5)]
    ctx = rt.Context(0)
    ctx.scene_upload(_two_entry_scene(rt, [twin] + others, twin))
    g = np.random.default_rng(5)
    n = 2000
    origins = np.tile(np.array([[0.0, 0.0, -10.0]]), (n, 1)) + g.normal(scale=0.05, size=(n, 3))
    targets = g.uniform(-0.6, 0.6, (n, 3)) * np.array([1.0, 1.0, 0.0])          # through the unit sphere at the origin
    d = targets - origins
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([origins, d, np.full((n, 1), 0.5)], axis=1).astype(np.float32)
    gi, gt = ctx.first_hit(rays)
    assert (gi == 0).all(), np.unique(gi)              # never 3, the list's twin, although it is the later world member
    # the same sphere alone in a list gives the same t: the tie is exact
    alone = rt.Context(0)
    alone.scene_upload(_two_entry_scene(rt, others, twin))
    ai, at = alone.first_hit(rays)
    assert (ai == len(others)).all() and (at == gt).all()
    # rays at the other spheres are nobody's tie
    origins2 = np.tile(np.array([[3.0, 0.0, -10.0]]), (50, 1))
    rays2 = np.concatenate([origins2, np.tile(np.array([[0.0, 0.0, 1.0]]), (50, 1)), np.full((50, 1), 0.5)], axis=1).astype(np.float32)
    assert (ctx.first_hit(rays2)[0] == 1).all()
