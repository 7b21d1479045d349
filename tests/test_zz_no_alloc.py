"""GPU test of mirt_render_static, the render entry point that allocates nothing: through the C ABI, against
mirt_render on the same inputs, on two calls in a row — the second with a smaller frame, so it finds the first
call's sums and counters in the static scratch. Where the frame fits sixteen chunk images (256 Ki pixels) both entry points add
the same floats in the same order: the accumulation buffers are equal bit for bit. A larger frame gets fewer,
longer chunks from the static scratch, i.e. another grouping of the same additions: rtol 2e-6, the bar the two
render loops are held to among themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(rt, ctx, w, h, spp, depth, seed, begin=0):
    ctx.frame_begin(w, h)
    ctx.render(spp, depth, spp_begin=begin, seed=seed, kernel=rt.KERNEL_QUEUE)
    want, seg = ctx.read_accum(), ctx.stats()["segments"]
    ctx.frame_begin(w, h)
    ctx.render_static(spp, depth, spp_begin=begin, seed=seed)
    st = ctx.stats()
    assert st["device_faults"] == 0 and st["kernel_used"] == rt.KERNEL_QUEUE and st["segments"] == seg
    return ctx.read_accum(), want


@pytest.mark.parametrize("name", ["cornell", "final"])
def test_static_scratch_render_equals_allocating_render(rt, earth, name):
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene(name, earth_rgba=earth if name == "final" else None))
    ctx.camera_set(rt.Camera.select_camera(16 / 9, name))
    got, want = _pair(rt, ctx, 320, 180, 40, 50, 3)                 # first call
    assert (got[..., 3] == 40).all() and (got == want).all()
    got, want = _pair(rt, ctx, 64, 36, 5, 50, 4, begin=7)           # second call, smaller input, ragged chunks
    assert (got[..., 3] == 5).all() and (got == want).all()
    got, want = _pair(rt, ctx, 64, 36, 1, 50, 4)                    # one sample: one chunk, straight into the sums
    assert (got == want).all()
    # 1920x1080: two chunk images fit, mirt_render uses sixteen
    got, want = _pair(rt, ctx, 1920, 1080, 32 if name == "cornell" else 16, 50, 5)
    assert (got[..., 3] == want[..., 3]).all()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    # and a small one again after the large one
    got, want = _pair(rt, ctx, 37, 19, 9, 50, 6)
    assert (got == want).all()


def test_static_scratch_render_rejects_what_does_not_fit(rt):
    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell"))
    ctx.camera_set(rt.Camera.select_camera(2.0, "cornell"))
    ctx.frame_begin(4200, 2100)                                      # 8.82 M pixels > 8 Mi
    with pytest.raises(rt.MirtError, match="MIRT_STATIC_MAX_PIXELS"):
        ctx.render_static(2, 5)
    ctx.frame_begin(64, 32)
    with pytest.raises(rt.MirtError, match="persistent"):
        ctx.render_static(2, 5, kernel=rt.KERNEL_FUSED)
    ctx.render_static(2, 5)
    assert (ctx.read_accum()[..., 3] == 2).all()
    # 3840x2160: not even two chunk images fit: whole pixels as work items, straight into the sums
    ctx.frame_begin(3840, 2160)
    ctx.render_static(2, 5, seed=1)
    a = ctx.read_accum()
    ctx.frame_begin(3840, 2160)
    ctx.render(2, 5, seed=1)
    assert (a == ctx.read_accum()).all()


def test_static_scratch_render_allocates_nothing(rt):
    """What the entry point exists for: no device memory is allocated or freed by the call. A fresh context that never
    went through mirt_render, a caller-owned accumulation buffer (so mirt_frame_begin allocates nothing either), one
    warm-up call (the first launch of a kernel makes the runtime load it), then free device memory (hipMemGetInfo)
    before and after calls with more samples and with another frame size: unchanged to the byte."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    ctx = rt.Context(0)
    ctx.scene_upload(rt.select_scene("cornell"))
    ctx.camera_set(rt.Camera.select_camera(16 / 9, "cornell"))
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), 1920 * 1080 * 16) == 0
    try:
        ctx.frame_begin(320, 180, buf)
        ctx.render_static(4, 10, seed=1)                             # warm-up
        ctx.synchronize()
        before = free_bytes()
        ctx.frame_begin(320, 180, buf)
        ctx.render_static(40, 50, seed=2)                            # more samples: sixteen chunks
        ctx.frame_begin(1920, 1080, buf)
        ctx.render_static(8, 50, seed=3)                             # larger frame: two chunk images
        ctx.frame_begin(64, 36, buf)
        ctx.render_static(3, 50, seed=4)
        ctx.synchronize()
        assert free_bytes() == before
        assert ctx.stats()["device_faults"] == 0
        ctx.frame_begin(64, 36, buf)
        ctx.render_static(3, 50, seed=4)
        a = ctx.read_accum()
        assert (a[..., 3] == 3).all() and np.isfinite(a).all()
    finally:
        ctx.synchronize()
        del ctx
        hip.hipFree(buf)
