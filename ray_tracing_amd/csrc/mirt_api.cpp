// C-ABI layer of libmirt.so (see include/mirt.h): context, scene flattening + upload, BVH build,
// frame / accumulation management and the launches. No exception crosses the boundary.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "bvh_build.h"
#include "dev_types.h"
#include "kernels.h"
#include "mirt_ctx.h"
#include "wavefront.h"

using namespace mirt;

namespace {
thread_local std::string g_create_err;

int fail(mirt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}
// Runs the body of an entry point; an exception (std::bad_alloc from a scene table, std::system_error from a
// builder thread) ends at the boundary as an error code, as include/mirt.h promises.
template <class Body>
int guarded(mirt_ctx* c, const char* where, Body&& body) {
    try {
        return body();
    } catch (const std::exception& e) {
        return fail(c, MIRT_ERR_INVALID, std::string(where) + ": " + e.what());
    } catch (...) {
        return fail(c, MIRT_ERR_INVALID, std::string(where) + ": unknown exception");
    }
}
}  // namespace
int mirt::ctx_fail(mirt_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }
namespace {
#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail((ctx), MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

std::string fault_text(uint32_t faults) {
    std::string m = "the render kernels reported";
    if (faults & FAULT_STACK_OVERFLOW) m += " [a hierarchy walk overflowed its traversal stack: hits may be missing]";
    if (faults & FAULT_STALE_CHUNK) m += " [a (pixel, sample chunk) item was never stored: sample counts are short]";
    return m;
}
uint32_t* fault_word(mirt_ctx* c) { return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->seg_counter) + kFaultOffset); }
uint32_t* work_counters(mirt_ctx* c) { return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->seg_counter) + kWorkCounterOffset); }

void free_scene(mirt_ctx* c) {
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->scene = DevScene{};
    c->has_scene = false;
}

template <class T>
int upload(mirt_ctx* c, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) {
        // keep every table pointer valid so a kernel never dereferences null
        void* p = nullptr;
        HIP_TRY(c, hipMalloc(&p, 64));
        HIP_TRY(c, hipMemset(p, 0, 64));
        c->scene_allocs.push_back(p);
        *out = static_cast<const T*>(p);
        return MIRT_OK;
    }
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, v.size() * sizeof(T)));
    c->scene_allocs.push_back(p);
    HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return MIRT_OK;
}

void rot_y_host(const mirt_instance& in, const float p[3], float out[3]) {
    out[0] = in.cos_theta * p[0] + in.sin_theta * p[2];
    out[1] = p[1];
    out[2] = -in.sin_theta * p[0] + in.cos_theta * p[2];
}

// True when the six primitives at p are the faces of one Cuboid as geo.rs:253-283 lists them: XY at
// z lo / hi, XZ at y lo / hi, YZ at x lo / hi, every face spanning the box's extents on its two axes.
bool is_cuboid(const mirt_prim* p) {
    static const uint32_t axes[6] = {MIRT_AXIS_XY, MIRT_AXIS_XY, MIRT_AXIS_XZ, MIRT_AXIS_XZ, MIRT_AXIS_YZ, MIRT_AXIS_YZ};
    for (int i = 0; i < 6; ++i)
        if (p[i].kind != MIRT_PRIM_RECT || p[i].axis != axes[i] || p[i].instance != p[0].instance) return false;
    const float lo[3] = {p[4].k, p[2].k, p[0].k}, hi[3] = {p[5].k, p[3].k, p[1].k};
    static const int first[6] = {0, 0, 0, 0, 1, 1}, second[6] = {1, 1, 2, 2, 2, 2};
    for (int i = 0; i < 6; ++i)
        if (p[i].lo[0] != lo[first[i]] || p[i].hi[0] != hi[first[i]] || p[i].lo[1] != lo[second[i]] || p[i].hi[1] != hi[second[i]]) return false;
    return true;
}

// Bounding box of one leaf primitive over shutter times [0,1] (geo.rs:116-130, 218-221), in the
// coordinate frame of its entry (a primitive's own instance applied, geo.rs:340-368, 317-323).
void prim_bounds(const mirt_prim& p, const mirt_instance* instances, float lo[3], float hi[3]) {
    float l[3], h[3];
    if (p.kind == MIRT_PRIM_SPHERE) {
        for (int a = 0; a < 3; ++a) {
            float c0 = p.center0[a], c1 = c0;
            if (p.flags & MIRT_PRIM_MOVING) {
                float d = p.center1[a] - p.center0[a];
                c0 = p.center0[a] + (0.f - p.time0) / (p.time1 - p.time0) * d;
                c1 = p.center0[a] + (1.f - p.time0) / (p.time1 - p.time0) * d;
            }
            float r = std::fabs(p.radius);
            l[a] = std::min(c0, c1) - r;
            h[a] = std::max(c0, c1) + r;
        }
    } else {
        int xi = p.axis == MIRT_AXIS_YZ ? 1 : 0;
        int yi = p.axis == MIRT_AXIS_XY ? 1 : 2;
        int zi = p.axis == MIRT_AXIS_XY ? 2 : (p.axis == MIRT_AXIS_XZ ? 1 : 0);
        l[xi] = p.lo[0]; h[xi] = p.hi[0];
        l[yi] = p.lo[1]; h[yi] = p.hi[1];
        l[zi] = p.k - 0.0001f; h[zi] = p.k + 0.0001f;
    }
    if (p.instance) {
        const mirt_instance& in = instances[p.instance];
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < 8; ++i) {
            float c[3] = {(i & 1) ? h[0] : l[0], (i & 2) ? h[1] : l[1], (i & 4) ? h[2] : l[2]}, q[3];
            rot_y_host(in, c, q);
            for (int a = 0; a < 3; ++a) {
                q[a] += in.offset[a];
                mn[a] = std::min(mn[a], q[a]);
                mx[a] = std::max(mx[a], q[a]);
            }
        }
        for (int a = 0; a < 3; ++a) { l[a] = mn[a]; h[a] = mx[a]; }
    }
    for (int a = 0; a < 3; ++a) { lo[a] = l[a]; hi[a] = h[a]; }
}
}  // namespace

extern "C" {

const char* mirt_last_error(const mirt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mirt_create(int device, mirt_ctx** out) {
    if (!out) return fail(nullptr, MIRT_ERR_INVALID, "mirt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(nullptr, MIRT_ERR_NO_DEVICE, "mirt_create: no HIP device available");
    if (device < 0 || device >= n) return fail(nullptr, MIRT_ERR_INVALID, "mirt_create: device index out of range");
    mirt_ctx* c = new (std::nothrow) mirt_ctx();
    if (!c) return fail(nullptr, MIRT_ERR_INVALID, "mirt_create: out of host memory");
    c->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        mirt_destroy(c);
        return MIRT_ERR_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_r0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_r1)) != hipSuccess) return bail("hipEventCreate", e);
    // kCounters statistics words, then (from byte 128) the item counters of the persistent render kernel
    static_assert(kCounters * sizeof(unsigned long long) <= kFaultOffset && kFaultOffset + 4 <= kWorkCounterOffset, "block layout");
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->seg_counter), kWorkCounterOffset + kWorkCounterBytes)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->seg_counter, 0, kWorkCounterOffset + kWorkCounterBytes)) != hipSuccess) return bail("hipMemset", e);
    // this device's copy of the static scratch array (mirt_render_static renders without allocating)
    // (a device on which the static scratch cannot be found still renders through mirt_render; mirt_render_static
    // reports the missing scratch when it is called)
    if (static_scratch_address(&c->static_scratch) != hipSuccess) {
        c->static_scratch = nullptr;
        (void)hipGetLastError();
    }
    *out = c;
    return MIRT_OK;
}

void mirt_destroy(mirt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    free_scene(c);
    if (c->accum_owned) (void)hipFree(c->accum_owned);
    if (c->frame_dev) (void)hipFree(c->frame_dev);
    if (c->seg_counter) (void)hipFree(c->seg_counter);
    if (c->partial) (void)hipFree(c->partial);
    for (void* p : c->wf_allocs) (void)hipFree(p);
    if (c->wf_counts_host) (void)hipHostFree(c->wf_counts_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_r0) (void)hipEventDestroy(c->ev_r0);
    if (c->ev_r1) (void)hipEventDestroy(c->ev_r1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int scene_upload_impl(mirt_ctx* c, const mirt_scene_desc* d);
int mirt_scene_upload(mirt_ctx* c, const mirt_scene_desc* d) {
    if (!c) return MIRT_ERR_INVALID;
    if (!d) return fail(c, MIRT_ERR_INVALID, "mirt_scene_upload: desc is NULL");
    return guarded(c, "mirt_scene_upload", [&] { return scene_upload_impl(c, d); });
}

static int scene_upload_impl(mirt_ctx* c, const mirt_scene_desc* d) {
    auto bad = [&](const std::string& m) { return fail(c, MIRT_ERR_INVALID, "mirt_scene_upload: " + m); };
    if (d->n_prims && !d->prims) return bad("prims is NULL");
    if (d->n_world && !d->world) return bad("world is NULL");
    if (d->n_lights && !d->lights) return bad("lights is NULL");
    if (d->n_materials && !d->materials) return bad("materials is NULL");
    if (d->n_prims && d->n_materials == 0) return bad("no materials");
    if (d->n_textures && !d->textures) return bad("textures is NULL");
    if (d->n_perlins && !d->perlins) return bad("perlins is NULL");
    if (d->n_images && !d->images) return bad("images is NULL");
    if (d->n_instances && !d->instances) return bad("instances is NULL");

    uint32_t features = 0;
    // ---- validate + convert primitives
    std::vector<DevPrim> prims(d->n_prims);
    uint32_t n_media = 0;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const mirt_prim& p = d->prims[i];
        DevPrim q{};
        if (p.material >= d->n_materials) return bad("prim " + std::to_string(i) + ": material index out of range");
        if (p.instance && p.instance >= d->n_instances) return bad("prim " + std::to_string(i) + ": instance index out of range");
        q.material = p.material;
        q.instance = p.instance;
        if (p.instance) features |= F_INSTANCE;
        if (p.kind == MIRT_PRIM_SPHERE) {
            features |= F_SPHERE;
            q.meta = PK_SPHERE;
            for (int a = 0; a < 3; ++a) { q.a[a] = p.center0[a]; q.b[a] = p.center1[a]; }
            q.a[3] = p.radius;
            q.b[3] = p.time0;
            q.c0 = p.time1;
            if (p.flags & MIRT_PRIM_MOVING) {
                if (p.time0 == p.time1) return bad("prim " + std::to_string(i) + ": moving sphere with time0 == time1");
                q.meta |= PM_MOVING;
                features |= F_MOVING;
            }
        } else if (p.kind == MIRT_PRIM_RECT) {
            features |= F_RECT;
            if (p.axis > MIRT_AXIS_YZ) return bad("prim " + std::to_string(i) + ": rect axis");
            if (!(p.lo[0] < p.hi[0] && p.lo[1] < p.hi[1])) return bad("prim " + std::to_string(i) + ": rect extents (geo.rs:182)");
            q.meta = PK_RECT | (p.axis << PM_AXIS_SHIFT);
            q.a[0] = p.k; q.a[1] = p.lo[0]; q.a[2] = p.hi[0]; q.a[3] = p.lo[1];
            q.b[0] = p.hi[1];
            // geo.rs:187-188, 239: the light sample plane is the mid-point of the padded corners
            float k0 = p.k - 0.0001f, k1 = p.k + 0.0001f;
            q.b[1] = (k0 + k1) / 2.f;
        } else if (p.kind == MIRT_PRIM_MEDIUM) {
            features |= F_MEDIUM;
            if (p.instance) return bad("prim " + std::to_string(i) + ": a medium takes its transform from its boundary primitives");
            if ((uint64_t)p.boundary_first + p.boundary_count > d->n_prims || p.boundary_count == 0)
                return bad("prim " + std::to_string(i) + ": medium boundary range");
            for (uint32_t b = p.boundary_first; b < p.boundary_first + p.boundary_count; ++b)
                if (d->prims[b].kind == MIRT_PRIM_MEDIUM) return bad("prim " + std::to_string(i) + ": medium boundary must be spheres / rects");
            q.meta = PK_MEDIUM;
            uint32_t idx = n_media++;
            std::memcpy(&q.a[0], &p.boundary_first, 4);
            std::memcpy(&q.a[1], &p.boundary_count, 4);
            q.a[2] = p.neg_inv_density;
            std::memcpy(&q.a[3], &idx, 4);
        } else {
            return bad("prim " + std::to_string(i) + ": unknown kind");
        }
        if (p.flags & MIRT_PRIM_FLIP_FACE) q.meta |= PM_FLIP;
        prims[i] = q;
    }
    // ---- primitive records: 32 bytes per hierarchy-leaf member; list scans use 64-byte records that hold
    //      one list member or two consecutive rects of the same axis and instance, in list order
    std::vector<DevScanRec> scan;
    auto scan_record = [&](uint32_t i) {
        const mirt_prim& p = d->prims[i];
        DevLeafRec r{};
        uint32_t code;
        if (p.kind == MIRT_PRIM_SPHERE) {
            code = (p.flags & MIRT_PRIM_MOVING) ? SC_SPHERE_MOVING : SC_SPHERE;
            for (int a = 0; a < 3; ++a) r.f[a] = p.center0[a];
            r.f[3] = p.radius;
        } else if (p.kind == MIRT_PRIM_RECT) {
            code = SC_RECT_XY + p.axis;
            r.f[0] = p.k; r.f[1] = p.lo[0]; r.f[2] = p.hi[0]; r.f[3] = p.lo[1]; r.f[4] = p.hi[1];
        } else {
            code = SC_MEDIUM;
        }
        r.code_inst = code | (p.instance << 8);
        r.prim = i;
        return r;
    };
    // merge_cuboids: three consecutive rect-pair records XY, XZ, YZ of one instance whose planes and extents are the six
    // faces of one box, in the order geo.rs:253-268 builds them, become one SC_CUBOID record of two corners (world lists
    // only: a medium's boundary keeps its three records, whose shape medium_hit recognises).
    auto make_scan = [&](uint32_t first, uint32_t count, uint32_t* scan_first, uint32_t* scan_count, bool merge_cuboids) {
        *scan_first = (uint32_t)scan.size();
        for (uint32_t i = first; i < first + count; ++i) {
            const DevLeafRec one = scan_record(i);
            DevScanRec r{};
            r.code_inst = one.code_inst;
            r.prim = i;
            const mirt_prim& p = d->prims[i];
            const bool pair = p.kind == MIRT_PRIM_RECT && i + 1 < first + count && d->prims[i + 1].kind == MIRT_PRIM_RECT &&
                              d->prims[i + 1].axis == p.axis && d->prims[i + 1].instance == p.instance;
            if (pair) {
                const DevLeafRec two = scan_record(i + 1);
                r.code_inst = (SC_RECT2_XY + p.axis) | (p.instance << 8);
                r.prim2 = i + 1;
                for (int a = 0; a < 5; ++a) { r.f[2 * a] = one.f[a]; r.f[2 * a + 1] = two.f[a]; }
                ++i;
            } else {
                for (int a = 0; a < 6; ++a) r.f[a] = one.f[a];
            }
            if (p.kind == MIRT_PRIM_SPHERE && (p.flags & MIRT_PRIM_MOVING)) {
                for (int a = 0; a < 3; ++a) r.f[4 + a] = p.center1[a];
                r.f[7] = p.time0;
                r.f[8] = p.time1;
            }
            if (p.kind == MIRT_PRIM_MEDIUM) {
                // (the media's primitives were completed above: boundary records, shape, index)
                const DevPrim& q = prims[i];
                uint32_t bfirst, shape;
                std::memcpy(&bfirst, &q.a[0], 4);
                std::memcpy(&shape, &q.b[0], 4);
                for (int a = 0; a < 12; ++a) r.f[a] = 0.f;
                r.f[4] = q.a[2];
                r.f[5] = q.a[3];
                r.f[6] = q.b[0];
                r.f[8] = q.a[0];
                r.f[9] = q.a[1];
                if (shape == MS_SPHERE) {
                    const DevScanRec b = scan[bfirst];
                    for (int a = 0; a < 4; ++a) r.f[a] = b.f[a];
                    const uint32_t inst = b.code_inst >> 8;
                    std::memcpy(&r.f[7], &inst, 4);
                }
            }
            scan.push_back(r);
        }
        if (merge_cuboids) {
            std::vector<DevScanRec> merged;
            uint32_t prev_inst = 0;
            for (size_t k = *scan_first; k < scan.size(); ++k) {
                const DevScanRec& a = scan[k];
                bool box = false;
                if (k + 2 < scan.size() && (a.code_inst & 0xffu) == SC_RECT2_XY && (scan[k + 1].code_inst & 0xffu) == SC_RECT2_XZ &&
                    (scan[k + 2].code_inst & 0xffu) == SC_RECT2_YZ && (a.code_inst >> 8) == (scan[k + 1].code_inst >> 8) &&
                    (a.code_inst >> 8) == (scan[k + 2].code_inst >> 8) && a.prim2 == a.prim + 1 && scan[k + 1].prim == a.prim + 2 &&
                    scan[k + 1].prim2 == a.prim + 3 && scan[k + 2].prim == a.prim + 4 && scan[k + 2].prim2 == a.prim + 5) {
                    const float* xy = a.f; const float* xz = scan[k + 1].f; const float* yz = scan[k + 2].f;
                    // pair record: f = (k_a, k_b, lo0_a, lo0_b, hi0_a, hi0_b, lo1_a, lo1_b, hi1_a, hi1_b)
                    const float lx = xy[2], hx = xy[4], ly = xy[6], hy = xy[8], lz = xy[0], hz = xy[1];
                    auto same = [](const float* f, float k_a, float k_b, float l0, float h0, float l1, float h1) {
                        return f[0] == k_a && f[1] == k_b && f[2] == l0 && f[3] == l0 && f[4] == h0 && f[5] == h0 && f[6] == l1 && f[7] == l1 &&
                               f[8] == h1 && f[9] == h1;
                    };
                    box = same(xy, lz, hz, lx, hx, ly, hy) && same(xz, ly, hy, lx, hx, lz, hz) && same(yz, lx, hx, ly, hy, lz, hz);
                    if (box) {
                        DevScanRec c{};
                        c.code_inst = SC_CUBOID | (prev_inst << 8);  // (the scan's frame stays what it was: list_scan)
                        c.prim = a.prim;
                        c.prim2 = a.code_inst >> 8;  // the box's own instance
                        c.f[0] = lx; c.f[1] = ly; c.f[2] = lz; c.f[3] = hx; c.f[4] = hy; c.f[5] = hz;
                        merged.push_back(c);
                        k += 2;
                    }
                }
                if (!box) {
                    merged.push_back(a);
                    prev_inst = a.code_inst >> 8;
                }
            }
            scan.resize(*scan_first);
            scan.insert(scan.end(), merged.begin(), merged.end());
        }
        *scan_count = (uint32_t)scan.size() - *scan_first;
    };
    if (d->n_instances >= (1u << 24)) return bad("too many instances");
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const mirt_prim& p = d->prims[i];
        if (p.kind != MIRT_PRIM_MEDIUM) continue;
        uint32_t rf, rc;
        make_scan(p.boundary_first, p.boundary_count, &rf, &rc, false);
        std::memcpy(&prims[i].a[0], &rf, 4);
        std::memcpy(&prims[i].a[1], &rc, 4);
        // boundary shapes whose two crossings come from one evaluation (medium_hit in dev_core.h)
        uint32_t shape = MS_LIST;
        auto code_of = [&](uint32_t k) { return scan[rf + k].code_inst & 0xffu; };
        if (rc == 1 && code_of(0) == SC_SPHERE) shape = MS_SPHERE;
        if (rc == 3 && code_of(0) == SC_RECT2_XY && code_of(1) == SC_RECT2_XZ && code_of(2) == SC_RECT2_YZ &&
            (scan[rf].code_inst >> 8) == (scan[rf + 1].code_inst >> 8) && (scan[rf].code_inst >> 8) == (scan[rf + 2].code_inst >> 8))
            shape = MS_CUBOID;
        std::memcpy(&prims[i].b[0], &shape, 4);
    }
    // ---- materials / textures
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mirt_material& m = d->materials[i];
        switch (m.kind) {
            case MIRT_MAT_LAMBERTIAN: break;
            case MIRT_MAT_METAL: features |= F_MAT_METAL; break;
            case MIRT_MAT_DIELECTRIC:
                features |= F_MAT_DIELECTRIC;
                if (!(m.param > 0.f)) return bad("material " + std::to_string(i) + ": index of refraction must be > 0 (material.rs:109)");
                break;
            case MIRT_MAT_DIFFUSE_LIGHT: features |= F_MAT_LIGHT; break;
            case MIRT_MAT_ISOTROPIC: features |= F_MAT_ISOTROPIC; break;
            default: return bad("material " + std::to_string(i) + ": unknown kind");
        }
        bool textured = m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_DIFFUSE_LIGHT || m.kind == MIRT_MAT_ISOTROPIC;
        if (textured && m.texture >= d->n_textures) return bad("material " + std::to_string(i) + ": texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i)
        if (d->textures[i].kind > MIRT_TEX_IMAGE) return bad("texture " + std::to_string(i) + ": unknown kind");
    // per-primitive shading record: material kind / parameter and, where it is a constant, the colour
    std::vector<DevShade> shade(d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const mirt_material& m = d->materials[d->prims[i].material];
        DevShade q{};
        q.mat_kind = m.kind;
        q.param = m.param;
        q.tex_kind = TK_SOLID;
        if (m.kind == MIRT_MAT_METAL) {
            for (int a = 0; a < 3; ++a) q.color[a] = m.albedo[a];
        } else if (m.kind == MIRT_MAT_DIELECTRIC) {
            for (int a = 0; a < 3; ++a) q.color[a] = 1.f;
        } else {
            const mirt_texture& t = d->textures[m.texture];
            q.tex_kind = t.kind;
            q.texture = m.texture;
            for (int a = 0; a < 3; ++a) q.color[a] = t.color[a];
        }
        shade[i] = q;
    }
    std::vector<DevTexture> texs(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mirt_texture& t = d->textures[i];
        DevTexture q{};
        q.kind = t.kind;
        q.perlin = t.perlin;
        q.image = t.image;
        for (int a = 0; a < 3; ++a) { q.c0[a] = t.color[a]; q.c1[a] = t.color2[a]; }
        q.c1[3] = t.scale;
        switch (t.kind) {
            case MIRT_TEX_SOLID: break;
            case MIRT_TEX_CHECKER: features |= F_TEX_CHECKER; break;
            case MIRT_TEX_NOISE:
                features |= F_TEX_NOISE;
                if (t.perlin >= d->n_perlins) return bad("texture " + std::to_string(i) + ": perlin index out of range");
                break;
            case MIRT_TEX_IMAGE:
                features |= F_TEX_IMAGE;
                if (t.image >= d->n_images) return bad("texture " + std::to_string(i) + ": image index out of range");
                break;
            default: return bad("texture " + std::to_string(i) + ": unknown kind");
        }
        texs[i] = q;
    }
    std::vector<DevPerlin> perlins(d->n_perlins);
    for (uint32_t i = 0; i < d->n_perlins; ++i) {
        for (int j = 0; j < 256; ++j) {
            for (int a = 0; a < 3; ++a) perlins[i].grad[j][a] = d->perlins[i].grad[j][a];
            perlins[i].grad[j][3] = 0.f;
        }
        std::memcpy(perlins[i].perm, d->perlins[i].perm, sizeof(perlins[i].perm));
    }
    std::vector<DevInstance> insts(std::max<uint32_t>(d->n_instances, 1u));
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        DevInstance q{};
        q.sin_t = d->instances[i].sin_theta; q.cos_t = d->instances[i].cos_theta;
        q.ox = d->instances[i].offset[0]; q.oy = d->instances[i].offset[1]; q.oz = d->instances[i].offset[2];
        insts[i] = q;
    }
    // ---- lights: un-instanced spheres / rects (HittableList of scene.rs:286)
    std::vector<uint32_t> lights(d->lights, d->lights + d->n_lights);
    for (uint32_t li : lights) {
        if (li >= d->n_prims) return bad("light index out of range");
        if (d->prims[li].kind == MIRT_PRIM_MEDIUM || d->prims[li].instance)
            return bad("lights must be un-instanced spheres or rects");
    }
    std::vector<DevPrim> light_prims;
    for (uint32_t li : lights) light_prims.push_back(prims[li]);
    // ---- world entries + BVHs
    std::vector<DevEntry> entries(d->n_world);
    std::vector<DevBvhNode4> nodes;   // four-wide nodes of all hierarchies
    uint32_t stack_need = 1;
    struct EntryBox { float lo[3], hi[3]; };
    std::vector<EntryBox> entry_boxes;    // world boxes of the hierarchies, in world order
    std::vector<uint32_t> slots;          // first primitive of every leaf record, in tree order
    std::vector<uint8_t> slot_is_cuboid;  // the record stands for the six rects of a cuboid
    int deepest_leaf = 0;
    for (uint32_t e = 0; e < d->n_world; ++e) {
        const mirt_entry& w = d->world[e];
        DevEntry q{};
        if ((uint64_t)w.first_prim + w.prim_count > d->n_prims) return bad("entry " + std::to_string(e) + ": prim range");
        if (w.instance && w.instance >= d->n_instances) return bad("entry " + std::to_string(e) + ": instance index out of range");
        q.first_prim = w.first_prim;
        q.prim_count = w.prim_count;
        q.prims_per_object = std::max(1u, w.prims_per_object);
        q.instance = w.instance;
        uint32_t kind = w.kind;
        if (kind == MIRT_ENTRY_LIST) {
            if (w.instance) return bad("entry " + std::to_string(e) + ": a list entry carries instances on its prims");
            // A long plain list (the reference scans `random`'s ~488 spheres linearly, ray.rs:123-133)
            // is traversed through a hierarchy too: the closest hit is the same; only the winner of
            // an exact tie between two different primitives could differ.
            bool plain = w.prim_count > 32;
            for (uint32_t k = 0; plain && k < w.prim_count; ++k) {
                const mirt_prim& p = d->prims[w.first_prim + k];
                plain = p.kind != MIRT_PRIM_MEDIUM && p.instance == 0;
            }
            if (plain) {
                kind = MIRT_ENTRY_BVH;
                q.prims_per_object = 1;
            }
        }
        if (kind == MIRT_ENTRY_LIST) {
            q.kind = EK_LIST;
            make_scan(w.first_prim, w.prim_count, &q.scan_first, &q.scan_count, true);
        } else if (kind == MIRT_ENTRY_BVH) {
            if (w.prim_count == 0) return bad("entry " + std::to_string(e) + ": empty BVH (aabb.rs:98)");
            if (w.prim_count % q.prims_per_object) return bad("entry " + std::to_string(e) + ": prim_count is not a multiple of prims_per_object");
            q.kind = EK_BVH;
            features |= F_BVH;
            if (w.instance) features |= F_INSTANCE;
            std::vector<BuildObject> objs(w.prim_count / q.prims_per_object);
            for (uint32_t j = 0; j < objs.size(); ++j) {
                BuildObject& o = objs[j];
                o.first_prim = w.first_prim + j * q.prims_per_object;
                for (int a = 0; a < 3; ++a) { o.lo[a] = INFINITY; o.hi[a] = -INFINITY; }
                for (uint32_t k = 0; k < q.prims_per_object; ++k) {
                    const mirt_prim& p = d->prims[o.first_prim + k];
                    if (p.kind == MIRT_PRIM_MEDIUM) return bad("entry " + std::to_string(e) + ": media cannot be BVH members");
                    if (p.instance) return bad("entry " + std::to_string(e) + ": BVH members take the entry's instance");
                    float lo[3], hi[3];
                    prim_bounds(p, d->instances, lo, hi);
                    for (int a = 0; a < 3; ++a) { o.lo[a] = std::min(o.lo[a], lo[a]); o.hi[a] = std::max(o.hi[a], hi[a]); }
                }
            }
            {
                // the hierarchy's box in the world: the eight corners of its members' union through the entry's
                // transform (geo.rs:349-362 does the same for RotationY's own box), padded
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (const BuildObject& o : objs)
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); }
                EntryBox box;
                for (int a = 0; a < 3; ++a) { box.lo[a] = INFINITY; box.hi[a] = -INFINITY; }
                for (int corner = 0; corner < 8; ++corner) {
                    float p[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
                    if (w.instance) {
                        const mirt_instance& in = d->instances[w.instance];
                        const float x = in.cos_theta * p[0] + in.sin_theta * p[2], z = -in.sin_theta * p[0] + in.cos_theta * p[2];
                        p[0] = x + in.offset[0]; p[1] = p[1] + in.offset[1]; p[2] = z + in.offset[2];
                    }
                    for (int a = 0; a < 3; ++a) { box.lo[a] = std::min(box.lo[a], p[a]); box.hi[a] = std::max(box.hi[a], p[a]); }
                }
                for (int a = 0; a < 3; ++a) {
                    const float pad = 1e-4f * std::max(std::max(std::fabs(box.lo[a]), std::fabs(box.hi[a])), box.hi[a] - box.lo[a]) + 1e-4f;
                    box.lo[a] -= pad; box.hi[a] += pad;
                }
                entry_boxes.push_back(box);
            }
            // Objects that are cuboids (six rects in the order and with the extents of geo.rs:253-283) become
            // one leaf record each; other multi-primitive objects keep one record per primitive.
            bool cuboids = q.prims_per_object == 6;
            for (uint32_t j = 0; cuboids && j < objs.size(); ++j) cuboids = is_cuboid(d->prims + objs[j].first_prim);
            const uint32_t records_per_object = cuboids ? 1u : q.prims_per_object;
            // One object per leaf (the builder's rule; MIRT_BVH_LEAF=n for measurement runs). On the four-wide trees 2 / 3 / 4
            // objects per leaf measured `random` 3 050 -> 2 815 / 2 475 / 2 225 Msamples/s, 10^6 spheres 1 290 -> 1 110 / 975 /
            // 845, `final` 2 705 -> 2 500 / 2 325 / 2 190: half a node step saved per segment against one and a half more
            // sphere tests, each a dependent fetch of its own (the box test that replaces it is a lane of the node step).
            if (records_per_object > 16u) return bad("entry " + std::to_string(e) + ": prims_per_object is limited to 16");
            // Box padding for walk_node's fused plane distances: 2^-21 of the largest coordinate magnitude of the
            // hierarchy (the evaluation errs by 2^-24 |origin| in position for origins inside the scene; farther
            // origins are covered by the relative widening of the far bound).
            float reach = 0.f;
            for (const BuildObject& o : objs)
                for (int a = 0; a < 3; ++a) reach = std::max(reach, std::max(std::fabs(o.lo[a]), std::fabs(o.hi[a])));
            // a binary tree first (binned SAH, depth <= kBvhStackDepth), then two levels of it per four-wide node
            std::vector<DevBvhNode> binary;
            BvhBuilder builder(binary, slots, kBvhStackDepth, records_per_object, std::isfinite(reach) ? reach * 4.76837158e-7f : 0.f);
            const uint32_t root2 = builder.build(objs);
            uint32_t need = 0;
            static const bool by_area = [] { const char* v = std::getenv("MIRT_BVH_BY_AREA"); return v ? std::atoi(v) != 0 : true; }();
            q.bvh_root = collapse_to_four(binary, root2, nodes, &need, by_area);
            {
                static const bool report = std::getenv("MIRT_BVH_REPORT") != nullptr;
                if (report) {
                    double node_cost = 0.0, leaf_cost = 0.0;
                    tree_cost(nodes, q.bvh_root, &node_cost, &leaf_cost);
                    std::fprintf(stderr, "mirt: hierarchy of entry %u: %zu objects, %zu four-wide nodes so far, stack need %u, expected node steps %.3f, leaf steps %.3f\n",
                                 e, objs.size(), nodes.size(), need, node_cost, leaf_cost);
                }
            }
            deepest_leaf = std::max(deepest_leaf, builder.deepest_leaf());
            stack_need = std::max(stack_need, need);
            if (need > (uint32_t)kBvhStackMax)
                return bad("entry " + std::to_string(e) + ": a walk of this hierarchy can hold " + std::to_string(need) +
                           " references, the traversal stack at most " + std::to_string(kBvhStackMax));
            slot_is_cuboid.resize(slots.size(), cuboids ? 1 : 0);  // marks the slots this entry added
            // the hit record of a BVH member is rebuilt with the entry's transform
            for (uint32_t k = 0; k < w.prim_count; ++k) prims[w.first_prim + k].instance = w.instance;
        } else {
            return bad("entry " + std::to_string(e) + ": unknown kind");
        }
        entries[e] = q;
    }
    // nodes of all hierarchies renumbered breadth first from the roots: the most visited ones first (kernels keep the
    // first few in LDS)
    {
        std::vector<uint32_t*> roots;
        for (DevEntry& q : entries)
            if (q.kind == EK_BVH) roots.push_back(&q.bvh_root);
        if (!roots.empty()) order_nodes_breadth_first(nodes, roots);
    }
    // leaf records of the hierarchies, in tree order (a leaf reference holds a 27-bit record index)
    if (slots.size() >= (1u << 27) - 1u) return bad("too many hierarchy members");
    if (nodes.size() >= (1u << 25)) return bad("too many hierarchy nodes (a walk addresses a node by a 32-bit byte offset: at most 2^25 nodes of 128 bytes)");
    std::vector<DevLeafRec> leaves(slots.size());
    std::vector<DevLeafMoving> leaf_moving;   // beside the records, only when one of them is a moving sphere
    for (size_t i = 0; i < slots.size(); ++i) {
        if (!slot_is_cuboid[i]) {
            leaves[i] = scan_record(slots[i]);
            if ((leaves[i].code_inst & 0xffu) == SC_SPHERE_MOVING) {
                const mirt_prim& p = d->prims[slots[i]];
                leaf_moving.resize(slots.size());
                DevLeafMoving& m = leaf_moving[i];
                for (int a = 0; a < 3; ++a) m.center1[a] = p.center1[a];
                m.time0 = p.time0;
                m.time1 = p.time1;
            }
            continue;
        }
        const mirt_prim* p = d->prims + slots[i];  // XY(z lo), XY(z hi), XZ(y lo), XZ(y hi), YZ(x lo), YZ(x hi)
        DevLeafRec r{};
        r.code_inst = SC_CUBOID;
        r.prim = slots[i];
        r.f[0] = p[4].k; r.f[1] = p[2].k; r.f[2] = p[0].k;
        r.f[3] = p[5].k; r.f[4] = p[3].k; r.f[5] = p[1].k;
        leaves[i] = r;
    }
    // ---- images
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    std::vector<DevImage> images(d->n_images);
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const mirt_image& im = d->images[i];
        images[i].width = im.rgba ? im.width : 0;
        images[i].height = im.rgba ? im.height : 0;
        images[i].rgba = nullptr;
        if (im.rgba && im.width && im.height) {
            void* p = nullptr;
            size_t bytes = (size_t)im.width * im.height * 4;
            HIP_TRY(c, hipMalloc(&p, bytes));
            c->scene_allocs.push_back(p);
            HIP_TRY(c, hipMemcpy(p, im.rgba, bytes, hipMemcpyHostToDevice));
            images[i].rgba = static_cast<const uint8_t*>(p);
        } else {
            images[i].width = images[i].height = 0;
        }
    }
    DevScene s{};
    int rc;
    prims.push_back(DevPrim{});  // the list scan reads one record ahead
    if ((rc = upload(c, prims, &s.prims))) return rc;
    {
        DevScanRec end{};
        end.code_inst = SC_END;
        scan.push_back(end);  // the list scan reads one record ahead
    }
    if ((rc = upload(c, scan, &s.scan))) return rc;
    if ((rc = upload(c, shade, &s.shade))) return rc;
    if ((rc = upload(c, light_prims, &s.light_prims))) return rc;
    if ((rc = upload(c, entries, &s.entries))) return rc;
    if ((rc = upload(c, texs, &s.textures))) return rc;
    if ((rc = upload(c, images, &s.images))) return rc;
    if ((rc = upload(c, perlins, &s.perlins))) return rc;
    if ((rc = upload(c, insts, &s.instances))) return rc;
    if ((rc = upload(c, nodes, &s.bvh_nodes))) return rc;
    c->n_bvh_nodes = (uint32_t)nodes.size();
    c->n_scan_records = (uint32_t)scan.size() - 1u;  // without the SC_END record
    if ((rc = upload(c, leaves, &s.bvh_leaves))) return rc;
    s.bvh_leaf_moving = nullptr;
    if (!leaf_moving.empty() && (rc = upload(c, leaf_moving, &s.bvh_leaf_moving))) return rc;
    std::vector<uint32_t> bvh_entries;
    for (uint32_t e = 0; e < d->n_world; ++e)
        if (entries[e].kind == EK_BVH) bvh_entries.push_back(e);
    if ((rc = upload(c, bvh_entries, &s.bvh_entries))) return rc;
    s.n_bvh_entries = (uint32_t)bvh_entries.size();
    {
        std::vector<DevBvhStart> starts(bvh_entries.size());
        for (size_t k = 0; k < bvh_entries.size(); ++k) {
            const DevEntry& en = entries[bvh_entries[k]];
            DevBvhStart st{};
            st.cos_t = 1.f;
            if (en.instance) {
                const DevInstance& in = insts[en.instance];
                st.sin_t = in.sin_t; st.cos_t = in.cos_t; st.ox = in.ox; st.oy = in.oy; st.oz = in.oz;
            }
            st.root = en.bvh_root;
            st.instance = en.instance;
            starts[k] = st;
        }
        if ((rc = upload(c, starts, &s.bvh_starts))) return rc;
    }
    {
        // Hierarchies whose world box is small against the union of all of them (under an eighth of its volume, all
        // boxes finite) are tested against their box before a walk starts them (walk_skip_missed).
        std::vector<float> bounds(entry_boxes.size() * 8, 0.f);
        float ulo[3] = {INFINITY, INFINITY, INFINITY}, uhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool finite = true;
        for (const EntryBox& b : entry_boxes)
            for (int a = 0; a < 3; ++a) {
                ulo[a] = std::min(ulo[a], b.lo[a]); uhi[a] = std::max(uhi[a], b.hi[a]);
                finite = finite && std::isfinite(b.lo[a]) && std::isfinite(b.hi[a]);
            }
        bool any = false;
        const double uvol = entry_boxes.empty() ? 0.0 : (double)(uhi[0] - ulo[0]) * (uhi[1] - ulo[1]) * (uhi[2] - ulo[2]);
        for (size_t k = 0; k < entry_boxes.size(); ++k) {
            const EntryBox& b = entry_boxes[k];
            const double vol = (double)(b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
            const bool flag = finite && entry_boxes.size() >= 2 && vol * 8.0 < uvol;
            for (int a = 0; a < 3; ++a) { bounds[8 * k + a] = b.lo[a]; bounds[8 * k + 4 + a] = b.hi[a]; }
            bounds[8 * k + 3] = flag ? 1.f : 0.f;
            any = any || flag;
        }
        s.bvh_bounds = nullptr;
        if (any && (rc = upload(c, bounds, &s.bvh_bounds))) return rc;
    }
    s.n_prims = d->n_prims;
    s.n_entries = d->n_world;
    s.n_lights = d->n_lights;
    s.n_media = n_media;
    s.n_instances = (uint32_t)insts.size();
    s.n_textures = d->n_textures;
    s.n_perlins = d->n_perlins;
    s.features = features;
    for (int a = 0; a < 3; ++a) s.background[a] = d->background[a];
    s.fault = fault_word(c);
    s.stack_limit = stack_need;   // what the deepest walk can need: the kernels' LDS stacks are launched this deep
    s.lds_nodes = (uint32_t)nodes.size();  // nodes there are; the persistent kernel's launch keeps as many in LDS as fit
    // (the test build of the library lets a test lower the limit to see the overflow report of walk_node)
#ifdef MIRT_DEBUG_HOOKS
    static const long stack_env = [] { const char* e = std::getenv("MIRT_DEBUG_STACK_LIMIT"); return e ? std::atol(e) : -1l; }();
    if (stack_env >= 0) s.stack_limit = (uint32_t)std::min<long>(std::max<long>(stack_env, 1), (long)stack_need);
#endif
    c->bvh_deepest_leaf = deepest_leaf;
    // what one scan of the world's list entries tests (a medium's boundary list is scanned at least once per scan
    // of the list that holds the medium): the per-segment list term of the algorithmic work
    c->list_rects = c->list_spheres = c->list_transforms = 0;
    for (uint32_t e = 0; e < d->n_world; ++e) {
        if (entries[e].kind != EK_LIST) continue;
        uint32_t last_inst = 0;
        auto count_range = [&](uint32_t first, uint32_t count, auto&& self) -> void {
            for (uint32_t i = first; i < first + count; ++i) {
                const mirt_prim& p = d->prims[i];
                if (p.instance && p.instance != last_inst) ++c->list_transforms;
                last_inst = p.instance;
                if (p.kind == MIRT_PRIM_RECT) ++c->list_rects;
                else if (p.kind == MIRT_PRIM_SPHERE) ++c->list_spheres;
                else self(p.boundary_first, p.boundary_count, self);
            }
        };
        count_range(entries[e].first_prim, entries[e].prim_count, count_range);
    }
    c->scene = s;
    c->has_scene = true;
    return MIRT_OK;
}

int mirt_camera_set(mirt_ctx* c, const mirt_camera* cam) {
    if (!c) return MIRT_ERR_INVALID;
    if (!cam) return fail(c, MIRT_ERR_INVALID, "mirt_camera_set: cam is NULL");
    DevCamera q{};
    for (int a = 0; a < 3; ++a) {
        q.origin[a] = cam->origin[a]; q.llc[a] = cam->lower_left_corner[a];
        q.horizontal[a] = cam->horizontal[a]; q.vertical[a] = cam->vertical[a];
        q.u[a] = cam->u[a]; q.v[a] = cam->v[a];
    }
    q.lens_radius = cam->lens_radius; q.time0 = cam->time0; q.time1 = cam->time1;
    c->cam = q;
    c->has_camera = true;
    return MIRT_OK;
}

int mirt_frame_begin(mirt_ctx* c, uint32_t width, uint32_t height, float* accum_device) {
    if (!c) return MIRT_ERR_INVALID;
    if (width < 2 || height < 2) return fail(c, MIRT_ERR_INVALID, "mirt_frame_begin: width and height must be >= 2 (renderer.rs:181-182 divides by width-1, height-1)");
    if ((uint64_t)width * height > 0x7fffffffull) return fail(c, MIRT_ERR_INVALID, "mirt_frame_begin: frame too large");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t pixels = (size_t)width * height;
    if (accum_device) {
        c->accum = reinterpret_cast<float4*>(accum_device);
    } else {
        if (c->accum_owned_pixels < pixels) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->accum_owned) (void)hipFree(c->accum_owned);
            c->accum_owned = nullptr;
            c->accum_owned_pixels = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->accum_owned), pixels * sizeof(float4)));
            c->accum_owned_pixels = pixels;
        }
        c->accum = c->accum_owned;
    }
    c->width = width;
    c->height = height;
    HIP_TRY(c, hipMemsetAsync(c->accum, 0, pixels * sizeof(float4), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->seg_counter, 0, kWorkCounterOffset, c->stream));  // statistics and the fault word
    c->stats = mirt_stats{};
    c->reduce_timed = false;
    return MIRT_OK;
}

}  // extern "C"

namespace {
template <class T>
int wf_alloc(mirt_ctx* c, T** out, size_t n) {
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, n * sizeof(T)));
    c->wf_allocs.push_back(p);
    *out = static_cast<T*>(p);
    return MIRT_OK;
}

// Path pool and queues of the wavefront kernel set for `slots` paths (grown on demand, never shrunk).
uint32_t wf_lane_capacity(uint32_t slots) {
    const uint32_t blocks = (slots + 255u) / 256u;
    return ((blocks + WF_LANES - 1u) / WF_LANES) * 256u;
}
int wf_ensure(mirt_ctx* c, uint32_t slots) {
    if (c->wf_capacity >= slots) return MIRT_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (void* p : c->wf_allocs) (void)hipFree(p);
    c->wf_allocs.clear();
    c->wf_capacity = 0;
    int rc;
    WfState st{};
    if ((rc = wf_alloc(c, &st.thr, slots))) return rc;
    if ((rc = wf_alloc(c, &st.acc, slots))) return rc;
    if ((rc = wf_alloc(c, &st.ids, slots))) return rc;
    const size_t entries = (size_t)WF_LANES * wf_lane_capacity(slots);
    for (int k = 0; k < 2; ++k) {
        if ((rc = wf_alloc(c, &st.q_o[k], entries))) return rc;
        if ((rc = wf_alloc(c, &st.q_d[k], entries))) return rc;
        if ((rc = wf_alloc(c, &st.q_h[k], entries))) return rc;
    }
    if ((rc = wf_alloc(c, &st.cls, (size_t)WC_COUNT * entries))) return rc;
    if ((rc = wf_alloc(c, &st.ctl, (size_t)WF_CTL_WORDS))) return rc;
    if ((rc = wf_alloc(c, &st.cost, (size_t)WF_MAX_WALK_GROUPS * 4))) return rc;
    HIP_TRY(c, hipMemset(st.cost, 0, (size_t)WF_MAX_WALK_GROUPS * 4 * sizeof(unsigned long long)));
    if (!c->wf_counts_host) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->wf_counts_host), WF_CTL_WORDS * sizeof(uint32_t)));
    c->wf = st;
    c->wf_capacity = slots;
    return MIRT_OK;
}

// One launch of the sample loop (a.spp_count samples per pixel, chunked as a.n_chunks says) with the wavefront kernel
// set: rounds are enqueued in batches without reading anything back; after a batch the live count of the pool comes
// to the host, and the render is over when it is zero. Rounds enqueued past the end find empty queues and cost a
// few microseconds each; the batches grow with the render so that neither they nor the read-backs matter.
int render_wavefront_launch(mirt_ctx* c, const RenderArgs& a, bool by_class, uint32_t* launches) {
    static const uint32_t pool = [] { const char* e = std::getenv("MIRT_WF_POOL"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : (1u << 22); }();
    static const uint32_t refill = [] { const char* e = std::getenv("MIRT_WF_REFILL"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)std::min(v, 64l) : 16u; }();
    static const uint32_t fetch_run = [] { const char* e = std::getenv("MIRT_WF_RUN"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : 128u; }();
    const uint64_t items = (uint64_t)((a.width + 7u) / 8u) * ((a.height + 7u) / 8u) * 64u * a.n_chunks;
    if (items >= 0xffffff00ull) return fail(c, MIRT_ERR_INVALID, "mirt_render: too many work items for one launch");
    const uint32_t capacity = (uint32_t)std::min<uint64_t>(pool, items);
    int rc = wf_ensure(c, capacity);
    if (rc) return rc;
    WfArgs w{};
    w.r = a;
    w.r.walk_budget = 0x7fffffffu;  // a walk's turn ends when a refill is due, not after a number of steps
    w.st = c->wf;
    w.capacity = capacity;
    w.lane_capacity = wf_lane_capacity(capacity);
    w.refill_lanes = refill;
    w.fetch_run = fetch_run;
    w.by_class = by_class ? 1u : 0u;
    HIP_TRY(c, wf_launch_init(w, c->stream));
    ++*launches;
    uint32_t round = 0, batch = 8;
    for (;;) {
        for (uint32_t k = 0; k < batch; ++k) {
            w.round = round++;
            HIP_TRY(c, wf_launch_round(w, c->stream));
            *launches += 4;
        }
        HIP_TRY(c, hipMemcpyAsync(c->wf_counts_host, w.st.ctl, WF_CTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint64_t live = 0;
        for (uint32_t k = 0; k < WF_LANES; ++k) live += c->wf_counts_host[k * WF_CTL_STRIDE + ((round & 1u) ? WF_CTL_N1 : WF_CTL_N0)];
        if (live == 0) break;
        batch = std::min(64u, std::max(8u, round / 4u));
    }
    HIP_TRY(c, wf_launch_finish(w, c->stream));
    if (a.n_chunks > 1) HIP_TRY(c, launch_reduce_chunks(a, c->stream));
    return MIRT_OK;
}
}  // namespace

extern "C" {

// Where a render call keeps what it needs beyond inputs and outputs: the item counters of the persistent kernel
// and the chunk sums. mirt_render uses the context's own block and a buffer it grows on demand; mirt_render_static
// uses the library's static device array and never allocates.
struct RenderScratch {
    uint32_t* work_counter;
    float4* partial;       // nullptr: take (and grow) the context's buffer
    size_t partial_bytes;  // capacity behind `partial`
    bool may_allocate;
};

static int render_impl(mirt_ctx* c, const mirt_render_params* p, const char* who, RenderScratch sc) {
    const std::string w(who);
    if (!p) return fail(c, MIRT_ERR_INVALID, w + ": params is NULL");
    if (!c->has_scene) return fail(c, MIRT_ERR_STATE, w + ": no scene uploaded");
    if (!c->has_camera) return fail(c, MIRT_ERR_STATE, w + ": no camera set");
    if (!c->accum || p->width != c->width || p->height != c->height)
        return fail(c, MIRT_ERR_STATE, w + ": width/height differ from mirt_frame_begin");
    if ((uint64_t)p->spp_begin + p->spp_count > 0xffffffffull) return fail(c, MIRT_ERR_INVALID, w + ": sample range overflows 32 bits");
    if ((p->kernel & 0xffu) > MIRT_KERNEL_QUEUE || (p->kernel & ~(0xffu | MIRT_KERNEL_COUNT_WALKS | MIRT_KERNEL_WF_CLASSES | MIRT_KERNEL_POOL))) return fail(c, MIRT_ERR_INVALID, w + ": unknown kernel selector");
    const uint32_t selector = p->kernel & 0xffu;
    HIP_TRY(c, hipSetDevice(c->device));
    RenderArgs a{};
    a.scene = c->scene;
    a.cam = c->cam;
    a.width = p->width; a.height = p->height;
    a.max_depth = p->max_depth; a.seed = p->seed; a.jitter = p->jitter;
    a.accum = c->accum;
    a.segment_counter = c->seg_counter;
    a.work_counter = sc.work_counter;
    a.loop = selector == MIRT_KERNEL_FUSED ? LOOP_FUSED : (selector == MIRT_KERNEL_QUEUE ? LOOP_QUEUE : LOOP_AUTO);
    a.count_cost = (p->kernel & MIRT_KERNEL_COUNT_WALKS) ? 1u : 0u;
    // depth 0 traces nothing (renderer.rs:198-200): the other kernels just count the samples
    const bool wavefront = selector == MIRT_KERNEL_WAVEFRONT && p->max_depth > 0 && sc.may_allocate;
    // the persistent kernel packs a pixel position into 16 + 16 bits: larger frames go to the one-pixel-per-lane kernel
    if (p->width > 0xffffu || p->height > 0xffffu) a.loop = LOOP_FUSED;
    a.debug_drop_pixel = 0xffffffffu;
#ifdef MIRT_DEBUG_HOOKS
    static const long drop_env = [] { const char* e = std::getenv("MIRT_DEBUG_DROP_PIXEL"); return e ? std::atol(e) : -1l; }();
    if (drop_env >= 0) a.debug_drop_pixel = (uint32_t)drop_env;  // tests of the stale-slot report
#endif
    {
        // How the persistent kernel shares a wave between its walks (kernels.h). Measured at 1080p with
        // (lanes walking, lanes descending) out of 64: `random` 1 740 Msamples/s with whole walks, 2 300 with
        // (24, 16); 1 M spheres at 4K 500 / 1 007; `final` 1 430 with whole walks, 1 285 with (24, 16), 1 610
        // with (8, 8). Where an iteration carries long list scans, media or procedural / image textures
        // (`final`), every extra iteration costs more, so walks are cut less: with the four-wide hierarchy `final`
        // at 256 samples per pixel gives 1 849 with (budget 48; 8, 8) and 1 913 with (96; 6, 8); with the world-box test
        // before a small hierarchy is started (walk_skip_missed) 2 364 with (96; 6, 8) and 2 397 with (96; 8, 8).
        auto env = [](const char* name) { const char* e = std::getenv(name); return e ? std::atol(e) : -1l; };
        static const long budget_env = env("MIRT_WALK_BUDGET"), lanes_env = env("MIRT_WALK_LANES"), node_env = env("MIRT_NODE_LANES");
        const bool heavy_iteration = (c->scene.features & (F_MEDIUM | F_TEX_NOISE | F_TEX_IMAGE)) != 0 || c->n_scan_records > 8u;
        a.walk_budget = budget_env > 0 ? (uint32_t)budget_env : (heavy_iteration ? 96u : 48u);
        a.walk_min_lanes = lanes_env >= 0 ? (uint32_t)lanes_env : (heavy_iteration ? 8u : 16u);  // (24 until the leaf step became one fetch: 10^6 spheres +1.5 %)
        a.walk_node_lanes = node_env >= 0 ? (uint32_t)node_env : (heavy_iteration ? 8u : 16u);
        // counters the items are shared out to (kernels.hip, take_items); MIRT_ITEM_COUNTERS for measurement runs
        static const long counters_env = env("MIRT_ITEM_COUNTERS");
        a.item_counters = (heavy_iteration && (c->scene.features & F_BVH)) ? 16u : 8u;
        if (counters_env > 0) {
            uint32_t n = 1;
            while (n * 2u <= (uint32_t)std::min<long>(counters_env, (long)kMaxItemCounters)) n *= 2u;
            a.item_counters = n;
        }
        // the pooled form of the walk (MIRT_KERNEL_POOL; MIRT_POOL=0/1 overrides for A/B runs of whole programs)
        static const long pool_env = env("MIRT_POOL"), pool_stop_env = env("MIRT_POOL_STOP"), pool_steps_env = env("MIRT_POOL_STEPS");
        a.pool = pool_env >= 0 ? (pool_env ? 1u : 0u) : ((p->kernel & MIRT_KERNEL_POOL) ? 1u : 0u);
        a.pool_stop = pool_stop_env > 0 ? (uint32_t)pool_stop_env : 32u;
        a.pool_steps = pool_steps_env > 0 ? (uint32_t)pool_steps_env : 2u;
    }
    // Long renders are cut into launches of at most kChunk samples per pixel: lanes of a wave
    // then finish within a few percent of each other and no launch runs for seconds.
    // measured on Cornell 1080p: 64 -> 9.5, 256 -> 10.6, 1024 -> 11.2 Gsamples/s (fewer launch tails)
    static const uint32_t kChunk = [] { const char* e = std::getenv("MIRT_CHUNK_SPP"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : 1024u; }();
    static const uint32_t split = [] {
        const char* e = std::getenv("MIRT_SPLIT");
        const long v = e ? std::atol(e) : 0;
        return v > 0 ? (uint32_t)std::min(v, 1024l) : 16u;
    }();
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    uint32_t launches = 0;
    for (uint32_t done = 0; done < p->spp_count; done += kChunk) {
        a.spp_begin = p->spp_begin + done;
        a.spp_count = std::min(kChunk, p->spp_count - done);
        a.n_chunks = 1;
        a.chunk_spp = a.spp_count;
        if ((render_uses_queue(a) || wavefront) && a.max_depth > 0 && split > 1 && a.spp_count > 1) {
            // The persistent kernel hands out (pixel, sample chunk) items: with whole pixels as items the launch
            // lasts as long as its most expensive pixel (measured on `final`: the average wave was alive for 46 %
            // of the kernel's duration, the rest was the tail of the pixels on the sphere cluster).
            const size_t slot_bytes = (size_t)a.width * a.height * sizeof(float4);
            uint32_t want = split;
            // a caller-provided scratch holds a fixed number of chunk images: fewer, longer chunks then
            if (!sc.may_allocate) want = (uint32_t)std::min<size_t>(want, sc.partial_bytes / slot_bytes);
            a.chunk_spp = (a.spp_count + want - 1) / std::max(want, 1u);
            a.n_chunks = (a.spp_count + a.chunk_spp - 1) / a.chunk_spp;
            const uint64_t items = (uint64_t)(((a.width + 7u) / 8u) * ((a.height + 7u) / 8u)) * 64u * a.n_chunks;
            const size_t bytes = (size_t)a.n_chunks * slot_bytes;
            if (want < 2 || items >= 0xffffff00ull || (uint64_t)a.n_chunks * a.width * a.height >= 0xffffffffull) {
                a.n_chunks = 1;
                a.chunk_spp = a.spp_count;
            } else if (sc.may_allocate && a.n_chunks > 1 && bytes > c->partial_bytes) {
                if (c->partial) HIP_TRY(c, hipFree(c->partial));
                c->partial = nullptr;
                c->partial_bytes = 0;
                HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->partial), bytes));
                c->partial_bytes = bytes;
            }
        }
        a.partial = sc.may_allocate ? c->partial : sc.partial;
        // (a stamp is a float's bit pattern in a slot nobody reads as a float; it only has to differ from launch to
        // launch among the launches that fill the same buffer: the context's own, or the device's static scratch)
        if (sc.may_allocate) {
            if (++c->stamp == 0) c->stamp = 1;
            a.stamp = c->stamp;
        } else {
            a.stamp = static_scratch_next_stamp();
        }
        if (wavefront) {
            const int rc = render_wavefront_launch(c, a, (p->kernel & MIRT_KERNEL_WF_CLASSES) != 0, &launches);
            if (rc) return rc;
            continue;
        }
        HIP_TRY(c, hipMemsetAsync(a.work_counter, 0, kWorkCounterBytes, c->stream));
        HIP_TRY(c, launch_render_fused(a, c->stream));
        ++launches;
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->stats.samples += (uint64_t)p->width * p->height * p->spp_count;
    c->stats.kernel_launches = launches;
    c->stats.kernel_used = wavefront ? MIRT_KERNEL_WAVEFRONT : (render_uses_queue(a) ? MIRT_KERNEL_QUEUE : MIRT_KERNEL_FUSED);
    return MIRT_OK;
}

int mirt_render(mirt_ctx* c, const mirt_render_params* p) {
    if (!c) return MIRT_ERR_INVALID;
    return render_impl(c, p, "mirt_render", RenderScratch{work_counters(c), nullptr, 0, true});
}

int mirt_render_static(mirt_ctx* c, const mirt_render_params* p) {
    if (!c) return MIRT_ERR_INVALID;
    if (!p) return fail(c, MIRT_ERR_INVALID, "mirt_render_static: params is NULL");
    if (p->kernel != MIRT_KERNEL_AUTO && p->kernel != MIRT_KERNEL_QUEUE)
        return fail(c, MIRT_ERR_INVALID, "mirt_render_static: only the persistent kernel (MIRT_KERNEL_AUTO / MIRT_KERNEL_QUEUE) renders from the static scratch");
    if (p->width > 0xffffu || p->height > 0xffffu || (uint64_t)p->width * p->height > MIRT_STATIC_MAX_PIXELS)
        return fail(c, MIRT_ERR_INVALID, "mirt_render_static: frame exceeds MIRT_STATIC_MAX_PIXELS (" + std::to_string(MIRT_STATIC_MAX_PIXELS) +
                                             " pixels, two chunk images in the static scratch) or 65535 in one dimension");
    if (!c->static_scratch) return fail(c, MIRT_ERR_STATE, "mirt_render_static: no static scratch on this device");
    char* base = static_cast<char*>(c->static_scratch);
    mirt_render_params q = *p;
    q.kernel = MIRT_KERNEL_QUEUE;  // AUTO would hand a lone-primitive scene to the one-pixel-per-lane kernel
    p = &q;
    return render_impl(c, p, "mirt_render_static",
                       RenderScratch{reinterpret_cast<uint32_t*>(base), reinterpret_cast<float4*>(base + kStaticPartialOffset), kStaticPartialBytes, false});
}

int mirt_resolve_rgba8_device(mirt_ctx* c, uint8_t* frame_device, size_t len) {
    if (!c) return MIRT_ERR_INVALID;
    if (!c->accum) return fail(c, MIRT_ERR_STATE, "mirt_resolve_rgba8: no frame begun");
    if (len != (size_t)c->width * c->height * 4) return fail(c, MIRT_ERR_INVALID, "mirt_resolve_rgba8: len != width*height*4 (renderer.rs:65)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!frame_device) {
        // the library's own device frame (the one the host variant stages through)
        if (c->frame_dev_bytes < len) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->frame_dev) (void)hipFree(c->frame_dev);
            c->frame_dev = nullptr;
            c->frame_dev_bytes = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->frame_dev), len));
            c->frame_dev_bytes = len;
        }
        frame_device = c->frame_dev;
    }
    HIP_TRY(c, launch_resolve_rgba8(c->accum, frame_device, c->width, c->height, c->stream));
    return MIRT_OK;
}

int mirt_resolve_rgba8(mirt_ctx* c, uint8_t* frame, size_t len) {
    if (!c) return MIRT_ERR_INVALID;
    if (!c->accum) return fail(c, MIRT_ERR_STATE, "mirt_resolve_rgba8: no frame begun");
    if (!frame) return fail(c, MIRT_ERR_INVALID, "mirt_resolve_rgba8: frame is NULL");
    if (len != (size_t)c->width * c->height * 4) return fail(c, MIRT_ERR_INVALID, "mirt_resolve_rgba8: len != width*height*4 (renderer.rs:65)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->frame_dev_bytes < len) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->frame_dev) (void)hipFree(c->frame_dev);
        c->frame_dev = nullptr;
        c->frame_dev_bytes = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->frame_dev), len));
        c->frame_dev_bytes = len;
    }
    HIP_TRY(c, launch_resolve_rgba8(c->accum, c->frame_dev, c->width, c->height, c->stream));
    HIP_TRY(c, hipMemcpyAsync(frame, c->frame_dev, len, hipMemcpyDeviceToHost, c->stream));
    uint32_t faults = 0;
    HIP_TRY(c, hipMemcpyAsync(&faults, fault_word(c), sizeof(faults), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (faults) return fail(c, MIRT_ERR_DEVICE, "mirt_resolve_rgba8: " + fault_text(faults));
    return MIRT_OK;
}

int mirt_read_accum(mirt_ctx* c, float* rgba, size_t n) {
    if (!c) return MIRT_ERR_INVALID;
    if (!c->accum) return fail(c, MIRT_ERR_STATE, "mirt_read_accum: no frame begun");
    if (!rgba || n != (size_t)c->width * c->height * 4) return fail(c, MIRT_ERR_INVALID, "mirt_read_accum: n != width*height*4");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(rgba, c->accum, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIRT_OK;
}

int mirt_write_accum(mirt_ctx* c, const float* rgba, size_t n) {
    if (!c) return MIRT_ERR_INVALID;
    if (!c->accum) return fail(c, MIRT_ERR_STATE, "mirt_write_accum: no frame begun");
    if (!rgba || n != (size_t)c->width * c->height * 4) return fail(c, MIRT_ERR_INVALID, "mirt_write_accum: n != width*height*4");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->accum, rgba, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIRT_OK;
}

int mirt_accum_device_ptr(mirt_ctx* c, float** out) {
    if (!c || !out) return MIRT_ERR_INVALID;
    if (!c->accum) return fail(c, MIRT_ERR_STATE, "mirt_accum_device_ptr: no frame begun");
    *out = reinterpret_cast<float*>(c->accum);
    return MIRT_OK;
}

static int ray_batch(mirt_ctx* c, const mirt_ray* rays, size_t n, bool color, uint32_t sample, uint32_t max_depth, uint32_t seed,
                     int32_t* prim_id, float* t, float* aov, float* out) {
    if (!c->has_scene) return fail(c, MIRT_ERR_STATE, "no scene uploaded");
    if (n == 0) return MIRT_OK;
    if (!rays) return fail(c, MIRT_ERR_INVALID, "rays is NULL");
    if (n > 0x7fffffffull) return fail(c, MIRT_ERR_INVALID, "too many rays");
    HIP_TRY(c, hipSetDevice(c->device));
    static_assert(sizeof(mirt_ray) == 7 * sizeof(float), "mirt_ray is 7 packed floats");
    float *d_rays = nullptr, *d_t = nullptr, *d_aov = nullptr, *d_out = nullptr;
    int32_t* d_id = nullptr;
    int rc = MIRT_OK;
    auto cleanup = [&]() {
        if (d_rays) (void)hipFree(d_rays);
        if (d_t) (void)hipFree(d_t);
        if (d_aov) (void)hipFree(d_aov);
        if (d_out) (void)hipFree(d_out);
        if (d_id) (void)hipFree(d_id);
    };
#define RB_TRY(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { cleanup(); return fail(c, MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    RB_TRY(hipMalloc(reinterpret_cast<void**>(&d_rays), n * sizeof(mirt_ray)));
    RB_TRY(hipMemcpyAsync(d_rays, rays, n * sizeof(mirt_ray), hipMemcpyHostToDevice, c->stream));
    RayBatchArgs a{};
    a.scene = c->scene;
    a.rays = d_rays;
    a.n = (uint32_t)n;
    a.sample = sample; a.max_depth = max_depth; a.seed = seed;
    if (color) {
        RB_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n * 3 * sizeof(float)));
        a.color = d_out;
        RB_TRY(launch_ray_color(a, c->stream));
        RB_TRY(hipMemcpyAsync(out, d_out, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    } else {
        RB_TRY(hipMalloc(reinterpret_cast<void**>(&d_id), n * sizeof(int32_t)));
        RB_TRY(hipMalloc(reinterpret_cast<void**>(&d_t), n * sizeof(float)));
        if (aov) RB_TRY(hipMalloc(reinterpret_cast<void**>(&d_aov), n * 12 * sizeof(float)));
        a.prim_id = d_id; a.t = d_t; a.aov = d_aov;
        RB_TRY(launch_first_hit(a, c->stream));
        RB_TRY(hipMemcpyAsync(prim_id, d_id, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        RB_TRY(hipMemcpyAsync(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (aov) RB_TRY(hipMemcpyAsync(aov, d_aov, n * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    RB_TRY(hipStreamSynchronize(c->stream));
#undef RB_TRY
    cleanup();
    return rc;
}

int mirt_first_hit(mirt_ctx* c, const mirt_ray* rays, size_t n, uint32_t seed, int32_t* prim_id, float* t, float* aov) {
    if (!c) return MIRT_ERR_INVALID;
    if (n && (!prim_id || !t)) return fail(c, MIRT_ERR_INVALID, "mirt_first_hit: output is NULL");
    return ray_batch(c, rays, n, false, 0, 0, seed, prim_id, t, aov, nullptr);
}

int mirt_ray_color(mirt_ctx* c, const mirt_ray* rays, size_t n, uint32_t sample, uint32_t max_depth, uint32_t seed, float* out) {
    if (!c) return MIRT_ERR_INVALID;
    if (n && !out) return fail(c, MIRT_ERR_INVALID, "mirt_ray_color: out is NULL");
    return ray_batch(c, rays, n, true, sample, max_depth, seed, nullptr, nullptr, nullptr, out);
}

int mirt_synchronize(mirt_ctx* c) {
    if (!c) return MIRT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MIRT_OK;
}

int mirt_get_stats(mirt_ctx* c, mirt_stats* out) {
    if (!c || !out) return MIRT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long seg[kCounters] = {};
    HIP_TRY(c, hipMemcpy(seg, c->seg_counter, sizeof(seg), hipMemcpyDeviceToHost));
    uint32_t faults = 0;
    HIP_TRY(c, hipMemcpy(&faults, fault_word(c), sizeof(faults), hipMemcpyDeviceToHost));
    c->stats.segments = seg[0];
    c->stats.lane_slots = seg[1];
    c->stats.bvh_node_steps = seg[2];
    c->stats.bvh_sphere_tests = seg[3];
    c->stats.bvh_rect_tests = seg[4];
    c->stats.bvh_box_tests = seg[5];
    c->stats.bvh_stack_entries = c->scene.stack_limit;
    c->stats.list_rects = c->list_rects;
    c->stats.list_spheres = c->list_spheres;
    c->stats.list_transforms = c->list_transforms;
    c->stats.device_faults = faults;
    c->stats.bvh_deepest_leaf = (uint32_t)c->bvh_deepest_leaf;
    if (faults) c->err = fault_text(faults);
    if (c->stats.kernel_launches) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.kernel_ms = ms;
    }
    if (c->reduce_timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_r0, c->ev_r1) == hipSuccess) c->stats.reduce_ms = ms;
    }
    *out = c->stats;
    return MIRT_OK;
}

void mirt_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // extern "C"
