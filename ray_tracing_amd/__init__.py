"""ray_tracing_amd — MI355X (gfx950) path-tracing backend behind the reference's Renderer API.

Python mirror of the reference's host interface for the per-pixel sample loop
(KaminariOS/ray_tracing src/renderer.rs:41 `Renderer::new`, :56 `Renderer::draw`,
src/scene.rs:11 `select_scene`, src/camera.rs:96 `Camera::select_camera`), bound with ctypes to the
C ABI of libmirt.so (include/mirt.h, include/mirt_host.h). All rendering happens in the HIP library;
there is no CPU fallback: importing the binding without libmirt.so raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRT_LIB", os.path.join(_HERE, "libmirt.so"))  # MIRT_LIB: A/B builds of the library
ASSETS = os.path.join(_HERE, "assets")

MIRT_OK = 0
KERNEL_AUTO, KERNEL_FUSED, KERNEL_WAVEFRONT, KERNEL_QUEUE = 0, 1, 2, 3
KERNEL_WF_CLASSES = 0x200    # OR-ed into KERNEL_WAVEFRONT: one queue per class of what a path does next, shading class by class
KERNEL_POOL = 0x400          # OR-ed into KERNEL_AUTO / KERNEL_QUEUE: a workgroup's hierarchy walks pooled in LDS, advanced by its waves together
KERNEL_COUNT_WALKS = 0x100   # OR-ed into a kernel selector: also tally the hierarchy walks (stats bvh_*), a few percent slower

# every symbol include/mirt.h and include/mirt_host.h declare
EXPORTS = [
    "mirt_create", "mirt_destroy", "mirt_last_error", "mirt_scene_upload", "mirt_camera_set", "mirt_frame_begin",
    "mirt_render", "mirt_render_static", "mirt_resolve_rgba8", "mirt_resolve_rgba8_device", "mirt_read_accum", "mirt_write_accum",
    "mirt_accum_device_ptr",
    "mirt_first_hit", "mirt_ray_color", "mirt_synchronize", "mirt_get_stats", "mirt_philox4x32_10",
    "mirt_reduce", "mirt_comm_unique_id", "mirt_comm_init_rank", "mirt_comm_reduce", "mirt_comm_ranks",
    "mirt_host_select_scene", "mirt_host_scene_desc", "mirt_host_scene_free", "mirt_host_select_camera",
    "mirt_host_draw", "mirt_host_save_png", "mirt_host_last_error",
]


class MirtError(RuntimeError):
    pass


class mirt_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class mirt_render_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint32), ("jitter", C.c_uint32), ("kernel", C.c_uint32)]


class mirt_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("kernel_ms", C.c_double),
                ("kernel_launches", C.c_uint32), ("kernel_used", C.c_uint32), ("lane_slots", C.c_uint64),
                ("bvh_node_steps", C.c_uint64), ("bvh_sphere_tests", C.c_uint64), ("bvh_rect_tests", C.c_uint64),
                ("list_rects", C.c_uint32), ("list_spheres", C.c_uint32), ("list_transforms", C.c_uint32),
                ("device_faults", C.c_uint32), ("bvh_deepest_leaf", C.c_uint32),
                ("bvh_stack_entries", C.c_uint32), ("bvh_box_tests", C.c_uint64), ("reduce_ms", C.c_double)]


class mirt_prim(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("instance", C.c_uint32), ("flags", C.c_uint32),
                ("center0", C.c_float * 3), ("radius", C.c_float), ("center1", C.c_float * 3),
                ("time0", C.c_float), ("time1", C.c_float),
                ("axis", C.c_uint32), ("k", C.c_float), ("lo", C.c_float * 2), ("hi", C.c_float * 2),
                ("boundary_first", C.c_uint32), ("boundary_count", C.c_uint32), ("neg_inv_density", C.c_float)]


class mirt_instance(C.Structure):
    _fields_ = [("sin_theta", C.c_float), ("cos_theta", C.c_float), ("offset", C.c_float * 3)]


class mirt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", C.c_float * 3), ("param", C.c_float)]


class mirt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("color2", C.c_float * 3), ("scale", C.c_float),
                ("perlin", C.c_uint32), ("image", C.c_uint32)]


class mirt_perlin(C.Structure):
    _fields_ = [("grad", (C.c_float * 3) * 256), ("perm", (C.c_uint8 * 256) * 3)]


class mirt_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba", C.POINTER(C.c_uint8))]


class mirt_entry(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("first_prim", C.c_uint32), ("prim_count", C.c_uint32),
                ("prims_per_object", C.c_uint32), ("instance", C.c_uint32)]


class mirt_scene_desc(C.Structure):
    _fields_ = [("prims", C.POINTER(mirt_prim)), ("n_prims", C.c_uint32),
                ("world", C.POINTER(mirt_entry)), ("n_world", C.c_uint32),
                ("lights", C.POINTER(C.c_uint32)), ("n_lights", C.c_uint32),
                ("materials", C.POINTER(mirt_material)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(mirt_texture)), ("n_textures", C.c_uint32),
                ("perlins", C.POINTER(mirt_perlin)), ("n_perlins", C.c_uint32),
                ("images", C.POINTER(mirt_image)), ("n_images", C.c_uint32),
                ("instances", C.POINTER(mirt_instance)), ("n_instances", C.c_uint32),
                ("background", C.c_float * 3)]


_lib = None


def build(force=False):
    """Compile libmirt.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load libmirt.so. Raises if the HIP library has not been built — there is no fallback.
    A process that also uses torch on the GPU has to import torch first: torch ships its own HIP runtime under the
    same library name, and if the system's one is already loaded for libmirt torch finds no device."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MirtError(f"{LIB_PATH} is missing: build it with ray_tracing_amd.build() (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.mirt_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.mirt_destroy.argtypes = [vp]
        L.mirt_destroy.restype = None
        L.mirt_last_error.argtypes = [vp]
        L.mirt_last_error.restype = C.c_char_p
        L.mirt_scene_upload.argtypes = [vp, C.POINTER(mirt_scene_desc)]
        L.mirt_camera_set.argtypes = [vp, C.POINTER(mirt_camera)]
        L.mirt_frame_begin.argtypes = [vp, u32, u32, vp]
        L.mirt_render.argtypes = [vp, C.POINTER(mirt_render_params)]
        L.mirt_render_static.argtypes = [vp, C.POINTER(mirt_render_params)]
        L.mirt_resolve_rgba8.argtypes = [vp, vp, sz]
        L.mirt_resolve_rgba8_device.argtypes = [vp, vp, sz]
        L.mirt_read_accum.argtypes = [vp, vp, sz]
        L.mirt_write_accum.argtypes = [vp, vp, sz]
        L.mirt_accum_device_ptr.argtypes = [vp, C.POINTER(vp)]
        L.mirt_first_hit.argtypes = [vp, vp, sz, u32, vp, vp, vp]
        L.mirt_ray_color.argtypes = [vp, vp, sz, u32, u32, u32, vp]
        L.mirt_synchronize.argtypes = [vp]
        L.mirt_get_stats.argtypes = [vp, C.POINTER(mirt_stats)]
        L.mirt_philox4x32_10.argtypes = [vp, vp, vp]
        L.mirt_philox4x32_10.restype = None
        L.mirt_reduce.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        L.mirt_comm_unique_id.argtypes = [vp]
        L.mirt_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp]
        L.mirt_comm_reduce.argtypes = [vp, C.c_int]
        L.mirt_comm_ranks.argtypes = [vp]
        L.mirt_host_select_scene.argtypes = [C.c_char_p, u32, C.c_int, vp, u32, u32, C.POINTER(vp)]
        L.mirt_host_scene_desc.argtypes = [vp]
        L.mirt_host_scene_desc.restype = C.POINTER(mirt_scene_desc)
        L.mirt_host_scene_free.argtypes = [vp]
        L.mirt_host_scene_free.restype = None
        L.mirt_host_select_camera.argtypes = [C.c_float, C.c_char_p, C.POINTER(mirt_camera)]
        L.mirt_host_draw.argtypes = [C.c_int, C.c_char_p, u32, u32, u32, u32, u32, u32, C.c_int, vp, u32, u32, vp, sz]
        L.mirt_host_save_png.argtypes = [C.c_char_p, vp, u32, u32]
        L.mirt_host_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def philox4x32_10(key, ctr):
    k = np.asarray(key, dtype=np.uint32)
    c = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().mirt_philox4x32_10(_ptr(k), _ptr(c), _ptr(out))
    return out


def save_png(path, frame):
    """image::save_buffer(path, &pixels, w, h, Rgba8) (cli.rs:31): frame is uint8 [H, W, 4], top row first."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    if frame.ndim != 3 or frame.shape[2] != 4:
        raise ValueError("frame must be uint8 [H, W, 4]")
    if lib().mirt_host_save_png(os.fsencode(path), _ptr(frame), frame.shape[1], frame.shape[0]) != MIRT_OK:
        raise MirtError(lib().mirt_host_last_error().decode())


def load_earthmap():
    """Decoded static/earthmap.jpg of the reference (its only data asset): RGBA8, top row first."""
    path = os.path.join(ASSETS, "earthmap.jpg")
    if not os.path.exists(path):
        return None
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8))


class Scene:
    """scene.rs:277 `Scene`, held in flattened form (mirt_scene_desc)."""

    def __init__(self, handle, label, keep=None):
        self._h = handle
        self.label = label
        self._keep = keep

    @property
    def desc(self):
        d = lib().mirt_host_scene_desc(self._h).contents
        d._owner = self  # the arrays the descriptor points into live as long as this Scene
        return d

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mirt_host_scene_free(self._h)
            self._h = None


def select_scene(name, scene_seed=1, random_num=11, earth_rgba="auto"):
    """scene.rs:11 `select_scene(name)`: random | 2psp | earth | simplelight | cornell | final | smoke | 2sp."""
    if isinstance(earth_rgba, str):
        earth_rgba = load_earthmap() if name in ("earth", "final") else None
    ep, ew, eh = None, 0, 0
    if earth_rgba is not None:
        earth_rgba = np.ascontiguousarray(earth_rgba, dtype=np.uint8)
        eh, ew = earth_rgba.shape[:2]
        ep = _ptr(earth_rgba)
    h = C.c_void_p()
    rc = lib().mirt_host_select_scene(name.encode(), scene_seed, random_num, ep, ew, eh, C.byref(h))
    if rc != MIRT_OK:
        raise MirtError(lib().mirt_host_last_error().decode())
    return Scene(h, name, keep=earth_rgba)


class Camera:
    """camera.rs:5 `Camera`."""

    def __init__(self, c_struct, aspect_ratio):
        self.c = c_struct
        self.aspect_ratio = aspect_ratio

    @staticmethod
    def select_camera(aspect_ratio, scene):
        """camera.rs:96 `Camera::select_camera(aspect_ratio, scene)`."""
        c = mirt_camera()
        rc = lib().mirt_host_select_camera(float(aspect_ratio), scene.encode(), C.byref(c))
        if rc != MIRT_OK:
            raise MirtError(lib().mirt_host_last_error().decode())
        return Camera(c, aspect_ratio)

    def as_array(self):
        c = self.c
        return np.array(list(c.origin) + list(c.lower_left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u)
                        + list(c.v) + [c.lens_radius, c.time0, c.time1], dtype=np.float32)


class Context:
    """One mirt_ctx: a HIP device, its stream, the uploaded scene and the accumulation buffer."""

    def __init__(self, device=0):
        self._c = C.c_void_p()
        rc = lib().mirt_create(device, C.byref(self._c))
        if rc != MIRT_OK:
            raise MirtError(lib().mirt_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_c", None):
            lib().mirt_destroy(self._c)
            self._c = None

    __del__ = close

    def _check(self, rc):
        if rc != MIRT_OK:
            raise MirtError(f"mirt error {rc}: {lib().mirt_last_error(self._c).decode()}")

    def scene_upload(self, scene):
        desc = scene.desc if isinstance(scene, Scene) else scene
        self._check(lib().mirt_scene_upload(self._c, C.byref(desc)))

    def camera_set(self, camera):
        c = camera.c if isinstance(camera, Camera) else camera
        self._check(lib().mirt_camera_set(self._c, C.byref(c)))

    def frame_begin(self, width, height, accum_device_ptr=None):
        self._check(lib().mirt_frame_begin(self._c, width, height, accum_device_ptr))
        self.width, self.height = width, height

    def render(self, spp_count, max_depth, spp_begin=0, seed=0, jitter=True, kernel=KERNEL_AUTO):
        p = mirt_render_params(self.width, self.height, spp_begin, spp_count, max_depth, seed, 1 if jitter else 0, kernel)
        self._check(lib().mirt_render(self._c, C.byref(p)))

    def render_static(self, spp_count, max_depth, spp_begin=0, seed=0, jitter=True, kernel=KERNEL_AUTO):
        """mirt_render_static: the same samples without any device allocation (static scratch, bounded frame size)."""
        p = mirt_render_params(self.width, self.height, spp_begin, spp_count, max_depth, seed, 1 if jitter else 0, kernel)
        self._check(lib().mirt_render_static(self._c, C.byref(p)))

    def resolve_rgba8(self, frame=None):
        if frame is None:
            frame = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(lib().mirt_resolve_rgba8(self._c, _ptr(frame), frame.size))
        return frame

    def resolve_rgba8_device(self, frame_device_ptr, nbytes):
        self._check(lib().mirt_resolve_rgba8_device(self._c, frame_device_ptr, nbytes))

    def read_accum(self):
        """float32 [H, W, 4] (sum r, g, b, sample count); row 0 = image bottom."""
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._check(lib().mirt_read_accum(self._c, _ptr(out), out.size))
        return out

    def write_accum(self, accum):
        """Restore a saved read_accum() image (checkpoint resume); continue with spp_begin = its sample count."""
        a = np.ascontiguousarray(accum, dtype=np.float32)
        if a.shape != (self.height, self.width, 4):
            raise ValueError(f"accum must be float32 [{self.height}, {self.width}, 4]")
        self._check(lib().mirt_write_accum(self._c, _ptr(a), a.size))

    def accum_device_ptr(self):
        p = C.c_void_p()
        self._check(lib().mirt_accum_device_ptr(self._c, C.byref(p)))
        return p.value

    def first_hit(self, rays, seed=0, aov=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        ids = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        a = np.zeros((n, 12), dtype=np.float32) if aov else None
        self._check(lib().mirt_first_hit(self._c, _ptr(rays), n, seed, _ptr(ids), _ptr(t), _ptr(a) if aov else None))
        return (ids, t, a) if aov else (ids, t)

    def ray_color(self, rays, sample=0, max_depth=50, seed=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros((rays.shape[0], 3), dtype=np.float32)
        self._check(lib().mirt_ray_color(self._c, _ptr(rays), rays.shape[0], sample, max_depth, seed, _ptr(out)))
        return out

    def synchronize(self):
        self._check(lib().mirt_synchronize(self._c))

    def stats(self):
        s = mirt_stats()
        self._check(lib().mirt_get_stats(self._c, C.byref(s)))
        return {"samples": s.samples, "segments": s.segments, "kernel_ms": s.kernel_ms,
                "kernel_launches": s.kernel_launches, "kernel_used": s.kernel_used, "lane_slots": s.lane_slots,
                "device_faults": s.device_faults, "bvh_deepest_leaf": s.bvh_deepest_leaf,
                "bvh_node_steps": s.bvh_node_steps, "bvh_sphere_tests": s.bvh_sphere_tests, "bvh_rect_tests": s.bvh_rect_tests,
                "bvh_box_tests": s.bvh_box_tests, "bvh_stack_entries": s.bvh_stack_entries, "reduce_ms": s.reduce_ms,
                "list_rects": s.list_rects, "list_spheres": s.list_spheres, "list_transforms": s.list_transforms}

    # ---- the exchange step of the sample-range split (RCCL sum of the accumulation buffers onto `root`)
    def comm_init_rank(self, n_ranks, rank, comm_id):
        """Join the communicator identified by the 128 bytes of `comm_id` (made by comm_unique_id() on one rank)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(comm_id))
        self._check(lib().mirt_comm_init_rank(self._c, n_ranks, rank, buf))

    def comm_reduce(self, root=0):
        self._check(lib().mirt_comm_reduce(self._c, root))

    def comm_ranks(self):
        return lib().mirt_comm_ranks(self._c)


def comm_unique_id():
    """128 bytes that identify a new communicator; carry them to every rank and pass them to Context.comm_init_rank."""
    buf = (C.c_uint8 * 128)()
    if lib().mirt_comm_unique_id(buf) != MIRT_OK:
        raise MirtError(lib().mirt_last_error(None).decode())
    return bytes(buf)


def reduce(contexts, root=0):
    """mirt_reduce: sum the accumulation buffers of contexts on different devices of this process onto contexts[root]."""
    arr = (C.c_void_p * len(contexts))(*[c._c for c in contexts])
    rc = lib().mirt_reduce(arr, len(contexts), root)
    if rc != MIRT_OK:
        raise MirtError(f"mirt error {rc}: {lib().mirt_last_error(contexts[0]._c).decode()}")


class Renderer:
    """renderer.rs:24 `Renderer`: Renderer(width, height, scene, camera); set `multisample` / `max_depth`
    (cli.rs:27-28); `draw(frame)` fills a caller-owned RGBA8 buffer of width*height*4 bytes, top row first,
    and is a no-op unless `dirty` (renderer.rs:57-60)."""

    def __init__(self, width, height, scene, camera, device=0):
        self.width, self.height = int(width), int(height)
        self.scene, self.camera = scene, camera
        self.multisample = 4   # renderer.rs:50
        self.max_depth = 10    # renderer.rs:51
        self.dirty = True
        self.seed = 0
        self._ctx = Context(device)
        self._ctx.scene_upload(scene)

    def draw(self, frame):
        if not self.dirty:
            return
        self.dirty = False
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or not frame.flags["C_CONTIGUOUS"]:
            raise ValueError("frame must be a C-contiguous uint8 buffer")
        if frame.size // 4 != self.width * self.height or frame.size % 4:
            raise ValueError("frame.len()/4 != width*height (renderer.rs:65)")
        ctx = self._ctx
        ctx.camera_set(self.camera)
        ctx.frame_begin(self.width, self.height)
        ctx.render(self.multisample, self.max_depth, seed=self.seed, jitter=self.multisample != 1)
        ctx.resolve_rgba8(frame)

    @property
    def context(self):
        return self._ctx
