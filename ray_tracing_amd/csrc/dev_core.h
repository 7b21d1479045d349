// Device code of the path tracer: vector math, the Philox draw streams, primitive intersection,
// list / BVH traversal, hit reconstruction, textures, materials and the light-mixture sampling.
// Every function states which part of the reference renderer it computes.
//
// Template parameter F (a mask of mirt::F_* bits) strips the code a scene does not need, so the
// Cornell-box kernel carries no BVH stack, medium, Perlin or image-texture code.
//
// Arithmetic policy. Everything that decides WHICH primitive a ray hits and at what t (ray
// transforms, plane / sphere roots, extent tests) is evaluated with separately rounded IEEE
// operations in the reference's order, so first-hit ids agree exactly with a CPU evaluation; the
// one liberty is that the plane division uses a shared reciprocal plus one Newton step (result
// correctly rounded except with probability ~2^-21), and the free path drawn inside a constant medium takes
// its logarithm from v_log_f32 (the draw is random; t stays within 1e-6). Shading (sampling directions, pdfs,
// throughput) uses the hardware approximations v_rcp / v_rsq / v_sqrt / v_sin / v_cos and fused multiply-adds
// written out by hand (fdot, fmadd, fcross, fma_), which perturb a sample by ~1e-6 relative and leave the
// estimator's expectation unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace mirt {

#define MIRT_DEV __device__ __forceinline__
#define MIRT_INF __builtin_huge_valf()
constexpr float kPi = 3.14159265358979323846f;

// ---------------------------------------------------------------------------------------------
// vector math — same operation order as the reference's nalgebra Vector3<f32>
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
MIRT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
MIRT_DEV V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
MIRT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MIRT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MIRT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
MIRT_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
MIRT_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
MIRT_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
MIRT_DEV V3 cmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
MIRT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MIRT_DEV float norm_squared(V3 a) { return dot(a, a); }
MIRT_DEV V3 normalize(V3 a) { return a / sqrtf(norm_squared(a)); }
MIRT_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MIRT_DEV float comp(V3 v, uint32_t i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
// per-component select (a struct-valued ?: makes hipcc index a two-element array in scratch memory)
MIRT_DEV V3 select(bool c, V3 a, V3 b) { return V3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
// A record of a scene table read through the constant address space. The scene arrays are never written
// while a kernel runs, but the compiler cannot know that once the kernel also stores to global memory
// inside its loop (the persistent render kernel writes finished pixels as it goes): reads it cannot prove
// unclobbered become per-lane vector loads even when their address is wave-uniform. Constant-address-space
// reads with a uniform address stay scalar loads (s_load_dwordx*), the data in SGPRs.
template <class T>
MIRT_DEV T load_const(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    constexpr int kWords = (int)(sizeof(T) / 4);
    const __attribute__((address_space(4))) uint32_t* q = (const __attribute__((address_space(4))) uint32_t*)p;
    union {
        T value;
        uint32_t words[kWords];
    } u;
#pragma unroll
    for (int i = 0; i < kWords; ++i) u.words[i] = q[i];
    return u.value;
}
// A launch argument read again from the kernel-argument segment where it is used (p points into that segment: a
// scalar load). The empty asm makes the pointer a new value to the compiler at every use, so it can neither hoist
// the load out of the render loop nor keep the value in scalar registers across it: arguments needed once in a
// few hundred iterations (the item bookkeeping) or only by some branch (the camera) would otherwise take scalar
// registers for the whole loop, and what no longer fits is spilled to vector-register lanes — v_readlane /
// v_writelane, which compete with the arithmetic for issue slots where a scalar load does not.
template <class T>
MIRT_DEV T reload(const __attribute__((address_space(4))) T* p) {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" : "+s"(p));
#endif
    return load_const((const T*)p);
}
// A pointer read from memory is of no known address space to the compiler: what it addresses is fetched with flat
// instructions, each with a 64-bit address put together in vector registers first (the node step of the variant that
// reloads its arguments: seven 64-bit adds and three moves in front of its seven fetches). The scene's arrays are all
// in global memory; said so once, here, the fetches are global loads from a scalar base and a 32-bit offset again.
template <class T>
MIRT_DEV T* in_global_memory(T* p) {
#ifdef __HIP_DEVICE_COMPILE__
    return (T*)(__attribute__((address_space(1))) T*)(uintptr_t)p;  // (by way of the integer: a cast between the spaces alone is folded away)
#else
    return p;
#endif
}
MIRT_DEV DevScene in_global_memory(DevScene s) {
    s.prims = in_global_memory(s.prims);
    s.shade = in_global_memory(s.shade);
    s.scan = in_global_memory(s.scan);
    s.entries = in_global_memory(s.entries);
    s.light_prims = in_global_memory(s.light_prims);
    s.textures = in_global_memory(s.textures);
    s.images = in_global_memory(s.images);
    s.perlins = in_global_memory(s.perlins);
    s.instances = in_global_memory(s.instances);
    s.bvh_nodes = in_global_memory(s.bvh_nodes);
    s.bvh_leaves = in_global_memory(s.bvh_leaves);
    s.bvh_leaf_moving = in_global_memory(s.bvh_leaf_moving);
    s.bvh_entries = in_global_memory(s.bvh_entries);
    s.bvh_starts = in_global_memory(s.bvh_starts);
    s.bvh_bounds = in_global_memory(s.bvh_bounds);
    s.fault = in_global_memory(s.fault);
    return s;
}

// Two floats in a register pair: + - * and fma on them are single packed instructions (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32), each half rounded exactly as the scalar operation would be.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// hardware approximations (about 1 ulp) for the shading code
MIRT_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
MIRT_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
MIRT_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
MIRT_DEV V3 fast_normalize(V3 a) { return a * fast_rsq(norm_squared(a)); }
// Fused forms for the shading code: a product that feeds a sum is one fused multiply-add, written out. (The
// library is built with contraction off, so that the intersection code rounds every operation as the reference does;
// letting the compiler contract the shading code instead of writing the fused operations out would fuse differently
// from one kernel to the next, and the kernels could no longer be compared bit for bit.)
MIRT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MIRT_DEV float fdot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
MIRT_DEV float fnorm_squared(V3 a) { return fdot(a, a); }
MIRT_DEV V3 fmadd(V3 a, float t, V3 b) { return V3{fma_(t, b.x, a.x), fma_(t, b.y, a.y), fma_(t, b.z, a.z)}; }  // a + t * b
MIRT_DEV V3 fcross(V3 a, V3 b) {
    return V3{fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
MIRT_DEV V3 fnormalize(V3 a) { return a * fast_rsq(fnorm_squared(a)); }
// sin / cos of 2*pi*u for u in [0, 1): v_sin_f32 / v_cos_f32 take their argument in revolutions
MIRT_DEV void sincos_2pi(float u, float& sn, float& cs) {
    sn = __builtin_amdgcn_sinf(u);
    cs = __builtin_amdgcn_cosf(u);
}
// sin(x) from v_sin_f32 (argument in revolutions, reduced to one turn first): absolute error about 1e-6 up to
// |x| of a few hundred — for texture patterns, where the reference's sin() only shapes a colour
MIRT_DEV float fast_sin(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
// n / d from r = v_rcp(d) and one Newton step; NaN when d == 0 (callers reject NaN)
MIRT_DEV float div_nr(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-q, d, n);
    return __builtin_fmaf(e, r, q);
}
// sqrt(x) from v_rsq and one Newton step (x >= 0; the x == 0 case would be 0 * inf)
MIRT_DEV float sqrt_nr(float x) {
    float r = fast_rsq(x);
    float s = x * r;
    float e = __builtin_fmaf(-s, s, x);
    float y = __builtin_fmaf(e, 0.5f * r, s);
    return x == 0.f ? 0.f : y;
}

struct Ray {
    V3 o, d;
    float time;
};
MIRT_DEV V3 ray_at(const Ray& r, float t) { return r.o + t * r.d; }

// A ray in the frame of the primitives being tested plus the reciprocals of its direction.
struct LRay {
    Ray r;
    V3 rinv;
    V3 nc;  // -origin * rinv: the distance to the plane at coordinate p of an axis is p * rinv + nc (walk_node)
};
MIRT_DEV LRay make_lray(const Ray& r) {
    LRay l;
    l.r = r;
    l.rinv = v3(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
    l.nc = cmul(-r.o, l.rinv);
    return l;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 draw streams (stands in for rand::random::<f32>() of rand_gen.rs:7-17)
//   key = (pixel, sample)   ctr = (block, bounce, stream, seed)
// draw i of a stream is word (i & 3) of block (i >> 2), mapped to [0,1) as (w >> 8) * 2^-24.
// ---------------------------------------------------------------------------------------------
enum : uint32_t { STREAM_CAMERA = 0, STREAM_SCATTER = 1, STREAM_MEDIUM = 2 };

// a ^ b ^ c in one instruction (v_bitop3_b32 with the truth table of the three-input xor)
MIRT_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
// A round is two 32 x 32 -> 64 bit products, two key additions and two three-input xors: six vector instructions.
MIRT_DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t& o0,
                            uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

MIRT_DEV float unit_f32(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// Sequential draws of one stream. `have` is the index of the block held in w0..w3.
struct Rng {
    uint32_t k0, k1, c1, c2, c3;
    uint32_t pos, have;
    uint32_t w0, w1, w2, w3;
    MIRT_DEV void init(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream, uint32_t seed) {
        k0 = pixel; k1 = sample; c1 = bounce; c2 = stream; c3 = seed; pos = 0; have = 0xffffffffu;
    }
    // start the stream with block 0 already generated by the caller
    MIRT_DEV void init_with_block0(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream, uint32_t seed, uint32_t b0,
                                   uint32_t b1, uint32_t b2, uint32_t b3) {
        init(pixel, sample, bounce, stream, seed);
        have = 0; w0 = b0; w1 = b1; w2 = b2; w3 = b3;
    }
    MIRT_DEV float next() {
        uint32_t blk = pos >> 2, i = pos & 3u;
        if (blk != have) {
            philox4x32_10(k0, k1, blk, c1, c2, c3, w0, w1, w2, w3);
            have = blk;
        }
        uint32_t w = i == 0 ? w0 : (i == 1 ? w1 : (i == 2 ? w2 : w3));
        ++pos;
        return unit_f32(w);
    }
    // rand_gen.rs:20-23
    MIRT_DEV float range(float mn, float mx) { return fma_(mx - mn, next(), mn); }
};

// rand_gen.rs:45-52 — rejection from the cube [-1,1)^3, three draws per trial.
MIRT_DEV V3 rand_in_unit_sphere(Rng& g) {
    for (;;) {
        float x = g.range(-1.f, 1.f);
        float y = g.range(-1.f, 1.f);
        float z = g.range(-1.f, 1.f);
        V3 p = v3(x, y, z);
        if (fnorm_squared(p) < 1.f) return p;
    }
}
MIRT_DEV float unit_range(uint32_t w) { return fma_(2.f, unit_f32(w), -1.f); }  // range(-1, 1) of the draw that took word w
MIRT_DEV bool sphere_trial(uint32_t wx, uint32_t wy, uint32_t wz, V3& p) {
    p = v3(unit_range(wx), unit_range(wy), unit_range(wz));
    return fnorm_squared(p) < 1.f;
}
// rand_in_unit_sphere for a stream whose next draw is word 0 of a block, which is where the scatter stream of a bounce
// stands when a metal or an isotropic material takes its only draws from it. Four trials take twelve words, three whole
// blocks, so the word of every draw is fixed when the code is compiled: the same draws in the same order as the loop
// above, without the bookkeeping of a position (some fourteen scalar instructions a draw) and without a block generated
// for a wave whose lanes are all done. `g` is left as it was: the caller draws nothing from it afterwards.
MIRT_DEV V3 rand_in_unit_sphere_last(const Rng& g) {
    uint32_t blk = g.pos >> 2;
    uint32_t a0 = g.w0, a1 = g.w1, a2 = g.w2, a3 = g.w3;
    bool have = g.have == blk;
    for (;;) {
        V3 p;
        if (!have) philox4x32_10(g.k0, g.k1, blk, g.c1, g.c2, g.c3, a0, a1, a2, a3);
        have = false;
        if (sphere_trial(a0, a1, a2, p)) return p;
        uint32_t b0, b1, b2, b3;
        philox4x32_10(g.k0, g.k1, blk + 1, g.c1, g.c2, g.c3, b0, b1, b2, b3);
        if (sphere_trial(a3, b0, b1, p)) return p;
        philox4x32_10(g.k0, g.k1, blk + 2, g.c1, g.c2, g.c3, a0, a1, a2, a3);
        if (sphere_trial(b2, b3, a0, p)) return p;
        if (sphere_trial(a1, a2, a3, p)) return p;
        blk += 3;
    }
}
// rand_gen.rs:70-78 — two draws per trial.
MIRT_DEV V3 rand_in_unit_disk(Rng& g) {
    for (;;) {
        float x = g.range(-1.f, 1.f);
        float y = g.range(-1.f, 1.f);
        V3 p = v3(x, y, 0.f);
        if (fma_(y, y, x * x) >= 1.f) continue;
        return p;
    }
}
// onb.rs:38-49 / onb.rs:34-36 — u and v are not normalised.
struct Onb {
    V3 u, v, w;
};
MIRT_DEV Onb onb_from_w(V3 w) {
    // v = w x a with a = (0,1,0) when |w.x| > 0.9, else (1,0,0). The products with a's zeros are written
    // out: for finite w they only contribute signed zeros that vanish in the sums (a NaN w gives NaN
    // directions either way).
    const bool ay = fabsf(w.x) > 0.9f;
    Onb o;
    o.v = v3(ay ? -w.z : 0.f, ay ? 0.f : w.z, ay ? w.x : -w.y);
    o.u = fcross(w, o.v);
    o.w = w;
    return o;
}
MIRT_DEV V3 onb_local(const Onb& o, V3 c) { return fmadd(fmadd(o.u * c.x, c.y, o.v), c.z, o.w); }

// ---------------------------------------------------------------------------------------------
// instances: Translation(RotationY(obj)) — geo.rs:306-315, 372-385
// ---------------------------------------------------------------------------------------------
MIRT_DEV V3 rot_y(const DevInstance& in, V3 p) {
    return v3(in.cos_t * p.x + in.sin_t * p.z, p.y, -in.sin_t * p.x + in.cos_t * p.z);
}
MIRT_DEV V3 inv_rot_y(const DevInstance& in, V3 p) {
    return v3(in.cos_t * p.x - in.sin_t * p.z, p.y, in.sin_t * p.x + in.cos_t * p.z);
}
MIRT_DEV Ray to_local(const DevInstance& in, const Ray& r) {
    Ray l;
    l.o = inv_rot_y(in, r.o - v3(in.ox, in.oy, in.oz));
    l.d = inv_rot_y(in, r.d);
    l.time = r.time;
    return l;
}

// Tables that shading reads with a per-lane (divergent) index. They point either at the scene's
// global arrays or at the block's LDS copies (small scenes).
struct Tables {
    const DevPrim* prims;
    const DevShade* shade;
    const DevInstance* instances;
    const DevPrim* light_prims;
};

// ---------------------------------------------------------------------------------------------
// primitives
// ---------------------------------------------------------------------------------------------
// geo.rs:62-69
template <uint32_t F>
MIRT_DEV V3 sphere_center(const DevPrim& p, float time) {
    V3 c0 = v3(p.a[0], p.a[1], p.a[2]);
    if ((F & F_MOVING) && (p.meta & PM_MOVING)) {
        V3 c1 = v3(p.b[0], p.b[1], p.b[2]);
        float t0 = p.b[3], t1 = p.c0;
        return c0 + (time - t0) / (t1 - t0) * (c1 - c0);
    }
    return c0;
}

// geo.rs:89-105 — a = |d|^2; nearest root in [t_min, t_max], both ends inclusive.
MIRT_DEV bool sphere_hit_c(V3 center, float radius, const Ray& r, float t_min, float t_max, float& t_out) {
    V3 oc = r.o - center;
    float a = norm_squared(r.d);
    float half_b = dot(oc, r.d);
    float c = norm_squared(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.f) return false;
    // square root and the two quotients from the hardware estimates plus one Newton step each:
    // correctly rounded except with probability ~2^-21
    float sqrtd = sqrt_nr(discriminant);
    float ra = fast_rcp(a);
    // Accepting form: a NaN root (NaN ray) is a miss. In the reference a NaN root passes both range
    // tests of geo.rs:100-104 and the NaN path is traced on to max_depth; a NaN ray only arises together
    // with a NaN throughput (geo.rs:80,138), so the sample is NaN -> 0 (renderer.rs:107) either way.
    float root = div_nr(-half_b - sqrtd, a, ra);
    if (!(root >= t_min && root <= t_max)) {
        root = div_nr(-half_b + sqrtd, a, ra);
        if (!(root >= t_min && root <= t_max)) return false;
    }
    t_out = root;
    return true;
}

// geo.rs:160-166
MIRT_DEV void axis_indexes(uint32_t axis, uint32_t& xi, uint32_t& yi, uint32_t& zi) {
    xi = axis == 2 ? 1u : 0u;
    yi = axis == 0 ? 1u : 2u;
    zi = axis == 0 ? 2u : (axis == 1 ? 1u : 0u);
}

// Where the medium free-path draws of one ray come from.
struct DrawKey {
    uint32_t pixel, sample, bounce, seed;
};

template <uint32_t F>
MIRT_DEV bool medium_hit(const DevScene& s, const DevScanRec& m, const Ray& r, float t_min, float t_max, const DrawKey& key,
                         float& t_out);

// closest = t, best = prim in the lanes where  t_min <= t <= closest,  lo0 <= x <= hi0  and  lo1 <= y <= hi1  — the
// range and extent tests of geo.rs:199-204 in accepting form (a NaN fails every compare) — written with compares that
// narrow the execution mask one after the other (v_cmpx) and two moves under the narrowed mask. The compiler's code for
// the same expression is six compares into scalar pairs, five s_and_b64 to join them and two selects, thirteen
// instructions against ten here; the render kernels are bound by the NUMBER of instructions a wave issues, scalar ones
// included (profiles/r3/probe_instruction_cost.txt), so the three are worth having. The first instruction saves the
// execution mask and also separates the compares from whatever packed instruction produced t, x or y.
// EXT: "s" when the extents and the id are wave-uniform (a scan record in scalar registers), "v" when they differ
// from lane to lane (a leaf record). TMIN: the same for t_min (the scan of a medium's boundary list starts its second
// pass at a per-lane t_min, geo.rs:416).
#ifdef __HIP_DEVICE_COMPILE__
#define MIRT_RECT_ACCEPT(EXT, TMIN, t, x, y, t_min, lo0, hi0, lo1, hi1, prim, closest, best)                                      \
    do {                                                                                                                  \
        unsigned long long mirt_saved_exec;                                                                               \
        asm volatile(                                                                                                     \
            "s_mov_b64 %[sv], exec\n"                                                                                     \
            "v_cmpx_le_f32 %[tm], %[tt]\n"                                                                                \
            "v_cmpx_ge_f32 %[cl], %[tt]\n"                                                                                \
            "v_cmpx_le_f32 %[l0], %[xx]\n"                                                                                \
            "v_cmpx_ge_f32 %[h0], %[xx]\n"                                                                                \
            "v_cmpx_le_f32 %[l1], %[yy]\n"                                                                                \
            "v_cmpx_ge_f32 %[h1], %[yy]\n"                                                                                \
            "v_mov_b32 %[cl], %[tt]\n"                                                                                    \
            "v_mov_b32 %[bb], %[pp]\n"                                                                                    \
            "s_mov_b64 exec, %[sv]\n"                                                                                     \
            : [cl] "+v"(closest), [bb] "+v"(best), [sv] "=&s"(mirt_saved_exec)                                            \
            : [tt] "v"(t), [xx] "v"(x), [yy] "v"(y), [tm] TMIN(t_min), [l0] EXT(lo0), [h0] EXT(hi0), [l1] EXT(lo1),        \
              [h1] EXT(hi1), [pp] EXT(prim)                                                                               \
            : "vcc");                                                                                                     \
    } while (0)
#else
#define MIRT_RECT_ACCEPT(EXT, TMIN, t, x, y, t_min, lo0, hi0, lo1, hi1, prim, closest, best)                                      \
    do {                                                                                                                  \
        if (((t) >= (t_min)) & ((t) <= (closest)) & ((x) >= (lo0)) & ((x) <= (hi0)) & ((y) >= (lo1)) & ((y) <= (hi1))) {  \
            closest = (t);                                                                                                \
            best = (int)(prim);                                                                                           \
        }                                                                                                                 \
    } while (0)
#endif

// geo.rs:198-204 on one scan record (f = k, lo0, hi0, lo1, hi1), written in accepting form so that a
// NaN t (direction parallel to the plane) is a miss. XI, YI, ZI = geo.rs:160-166.
template <uint32_t XI, uint32_t YI, uint32_t ZI, bool UNIFORM_TMIN>
MIRT_DEV void rect_rec_hit(const float* f, uint32_t prim, const LRay& l, float t_min, float& closest, int& best) {
    float t = div_nr(f[0] - comp(l.r.o, ZI), comp(l.r.d, ZI), comp(l.rinv, ZI));
    float x = comp(l.r.o, XI) + t * comp(l.r.d, XI);
    float y = comp(l.r.o, YI) + t * comp(l.r.d, YI);
    if constexpr (UNIFORM_TMIN) MIRT_RECT_ACCEPT("s", "s", t, x, y, t_min, f[1], f[2], f[3], f[4], prim, closest, best);
    else MIRT_RECT_ACCEPT("s", "v", t, x, y, t_min, f[1], f[2], f[3], f[4], prim, closest, best);
}

// Two consecutive rects of one axis (f = k, lo0, hi0, lo1, hi1, each as the pair a, b): the same
// operations as two rect_rec_hit calls, in the same order and with the same roundings, but the plane
// distances and hit coordinates of both rects come from packed two-float instructions (8 instead of
// 16). Rect b is tested against the range rect a has already narrowed, as in the sequential scan.
template <uint32_t XI, uint32_t YI, uint32_t ZI, bool UNIFORM_TMIN>
MIRT_DEV void rect_pair_hit(const float* f, uint32_t prim_a, uint32_t prim_b, V3 o, V3 d, V3 rinv, float t_min, float& closest, int& best) {
    const float oz = comp(o, ZI), dz = comp(d, ZI), rz = comp(rinv, ZI);
    const float ox = comp(o, XI), dx = comp(d, XI), oy = comp(o, YI), dy = comp(d, YI);
    const f32x2 n = f32x2{f[0], f[1]} - oz;
    const f32x2 q = n * rz;
    const f32x2 e = __builtin_elementwise_fma(-q, (f32x2)dz, n);
    const f32x2 t = __builtin_elementwise_fma(e, (f32x2)rz, q);  // div_nr on both halves
    const f32x2 x = ox + t * dx;
    const f32x2 y = oy + t * dy;
    if constexpr (UNIFORM_TMIN) {
        MIRT_RECT_ACCEPT("s", "s", t.x, x.x, y.x, t_min, f[2], f[4], f[6], f[8], prim_a, closest, best);
        MIRT_RECT_ACCEPT("s", "s", t.y, x.y, y.y, t_min, f[3], f[5], f[7], f[9], prim_b, closest, best);
    } else {
        MIRT_RECT_ACCEPT("s", "v", t.x, x.x, y.x, t_min, f[2], f[4], f[6], f[8], prim_a, closest, best);
        MIRT_RECT_ACCEPT("s", "v", t.y, x.y, y.y, t_min, f[3], f[5], f[7], f[9], prim_b, closest, best);
    }
}

// The two opposite faces of a cuboid on one axis (geo.rs:253-283: planes k_lo and k_hi with the same
// extents [lo0, hi0] x [lo1, hi1]); ids prim and prim + 1. Same arithmetic as rect_pair_hit.
// UNIFORM: the cuboid is a scan record (corners, id and t_min in scalar registers) rather than a leaf record.
template <uint32_t XI, uint32_t YI, uint32_t ZI, bool UNIFORM = false>
MIRT_DEV void face_pair_hit(float k_lo, float k_hi, float lo0, float hi0, float lo1, float hi1, uint32_t prim, V3 o, V3 d, V3 rinv,
                            float t_min, float& closest, int& best) {
    const float oz = comp(o, ZI), dz = comp(d, ZI), rz = comp(rinv, ZI);
    const float ox = comp(o, XI), dx = comp(d, XI), oy = comp(o, YI), dy = comp(d, YI);
    const f32x2 n = f32x2{k_lo, k_hi} - oz;
    const f32x2 q = n * rz;
    const f32x2 e = __builtin_elementwise_fma(-q, (f32x2)dz, n);
    const f32x2 t = __builtin_elementwise_fma(e, (f32x2)rz, q);
    const f32x2 x = ox + t * dx;
    const f32x2 y = oy + t * dy;
    const uint32_t prim_b = prim + 1u;
    if constexpr (UNIFORM) {
        MIRT_RECT_ACCEPT("s", "s", t.x, x.x, y.x, t_min, lo0, hi0, lo1, hi1, prim, closest, best);
        MIRT_RECT_ACCEPT("s", "s", t.y, x.y, y.y, t_min, lo0, hi0, lo1, hi1, prim_b, closest, best);
    } else {
        MIRT_RECT_ACCEPT("v", "v", t.x, x.x, y.x, t_min, lo0, hi0, lo1, hi1, prim, closest, best);
        MIRT_RECT_ACCEPT("v", "v", t.y, x.y, y.y, t_min, lo0, hi0, lo1, hi1, prim_b, closest, best);
    }
}

// closest = t, best = prim in the lanes where  t_min <= t <= limit  (the range test of geo.rs:100-104 in accepting
// form), as two narrowing compares and two moves under the narrowed mask; see MIRT_RECT_ACCEPT.
#ifdef __HIP_DEVICE_COMPILE__
#define MIRT_RANGE_ACCEPT(TMIN, PRIM, t, t_min, limit, prim, closest, best)                                                \
    do {                                                                                                                  \
        unsigned long long mirt_saved_exec;                                                                               \
        asm volatile(                                                                                                     \
            "s_mov_b64 %[sv], exec\n"                                                                                     \
            "v_cmpx_le_f32 %[tm], %[tt]\n"                                                                                \
            "v_cmpx_ge_f32 %[lm], %[tt]\n"                                                                                \
            "v_mov_b32 %[cl], %[tt]\n"                                                                                    \
            "v_mov_b32 %[bb], %[pp]\n"                                                                                    \
            "s_mov_b64 exec, %[sv]\n"                                                                                     \
            : [cl] "+v"(closest), [bb] "+v"(best), [sv] "=&s"(mirt_saved_exec)                                            \
            : [tt] "v"(t), [lm] "v"(limit), [tm] TMIN(t_min), [pp] PRIM(prim)                                             \
            : "vcc");                                                                                                     \
    } while (0)
#else
#define MIRT_RANGE_ACCEPT(TMIN, PRIM, t, t_min, limit, prim, closest, best)                                                \
    do {                                                                                                                  \
        if (((t) >= (t_min)) & ((t) <= (limit))) {                                                                        \
            closest = (t);                                                                                                \
            best = (int)(prim);                                                                                           \
        }                                                                                                                 \
    } while (0)
#endif

// Sphere::hit (geo.rs:89-105) with the result taken at once: closest = root, best = prim where the nearest root in
// [t_min, closest] exists. The same arithmetic as sphere_hit_c without its branches: a negative discriminant makes the
// square root, and with it both roots, NaN, and a NaN fails every compare. geo.rs:100-104 tries the smaller root first
// and the larger one only if the smaller is out of range; here the larger root is taken first and the smaller one —
// tested against the range as it was, not as the larger root left it — overrides it.
template <bool UNIFORM>
MIRT_DEV void sphere_accept(V3 center, float radius, uint32_t prim, const Ray& r, float t_min, float& closest, int& best) {
    V3 oc = r.o - center;
    float a = norm_squared(r.d);
    float half_b = dot(oc, r.d);
    float c = norm_squared(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    float sqrtd = sqrt_nr(discriminant);
    float ra = fast_rcp(a);
    const float root0 = div_nr(-half_b - sqrtd, a, ra), root1 = div_nr(-half_b + sqrtd, a, ra);
    const float limit = closest;
    if constexpr (UNIFORM) {
        MIRT_RANGE_ACCEPT("s", "s", root1, t_min, limit, prim, closest, best);
        MIRT_RANGE_ACCEPT("s", "s", root0, t_min, limit, prim, closest, best);
    } else {
        MIRT_RANGE_ACCEPT("v", "v", root1, t_min, limit, prim, closest, best);
        MIRT_RANGE_ACCEPT("v", "v", root0, t_min, limit, prim, closest, best);
    }
}

// One scan record as sixteen dwords in one scalar load (a 64-byte aligned record: s_load_dwordx16).
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
MIRT_DEV DevScanRec load_scan_record(const DevScanRec* p) {
    const u32x16 v = *(const __attribute__((address_space(4))) u32x16*)p;  // (a C cast, as in load_const: the address space changes)
    union {
        DevScanRec rec;
        u32x16 words;
    } u;
    u.words = v;
    return u.rec;
}

// HittableList::hit (ray.rs:123-133) over the scan records [first, first + count) of a list:
// primitives are visited in list order and `t <= closest` accepts, so the later primitive wins a
// tie. Records are wave-uniform: one 64-byte scalar load each (the other waves of the SIMD cover its
// latency; holding the next record in registers as well costs more than it hides), and the kind /
// axis dispatch is a scalar branch. Every kind updates closest / best itself and ends the record: a common tail that
// selects "hit ? t : closest" would run for the rect records too, which are most of a scan.
template <uint32_t F, bool ALLOW_MEDIUM>
MIRT_DEV void list_scan(const DevScene& s, uint32_t first, uint32_t count, const Ray& r, float t_min, const DrawKey& key,
                        float& closest, int& best) {
    uint32_t cur_inst = 0;
    LRay lr = make_lray(r);
    const DevScanRec* p = s.scan + first;
    for (uint32_t j = 0; j < count; ++j) {
        const DevScanRec rec = load_scan_record(p + j);
        const uint32_t code = rec.code_inst & 0xffu;
        if (F & F_INSTANCE) {
            const uint32_t inst = rec.code_inst >> 8;
            if (inst != cur_inst) {
                cur_inst = inst;
                lr = make_lray(inst ? to_local(load_const(s.instances + inst), r) : r);
            }
        }
        switch (code) {
            case SC_RECT2_XY: if (F & F_RECT) rect_pair_hit<0, 1, 2, ALLOW_MEDIUM>(rec.f, rec.prim, rec.prim2, lr.r.o, lr.r.d, lr.rinv, t_min, closest, best); break;
            case SC_RECT2_XZ: if (F & F_RECT) rect_pair_hit<0, 2, 1, ALLOW_MEDIUM>(rec.f, rec.prim, rec.prim2, lr.r.o, lr.r.d, lr.rinv, t_min, closest, best); break;
            case SC_RECT2_YZ: if (F & F_RECT) rect_pair_hit<1, 2, 0, ALLOW_MEDIUM>(rec.f, rec.prim, rec.prim2, lr.r.o, lr.r.d, lr.rinv, t_min, closest, best); break;
            case SC_RECT_XY: if (F & F_RECT) rect_rec_hit<0, 1, 2, ALLOW_MEDIUM>(rec.f, rec.prim, lr, t_min, closest, best); break;
            case SC_RECT_XZ: if (F & F_RECT) rect_rec_hit<0, 2, 1, ALLOW_MEDIUM>(rec.f, rec.prim, lr, t_min, closest, best); break;
            case SC_RECT_YZ: if (F & F_RECT) rect_rec_hit<1, 2, 0, ALLOW_MEDIUM>(rec.f, rec.prim, lr, t_min, closest, best); break;
            case SC_CUBOID:
                // The six rects of a cuboid (geo.rs:245-289) that the upload found as three consecutive rect pairs of one
                // instance: one record of two corners, ids prim .. prim + 5, its instance in prim2. The frame is set up
                // here and forgotten again, so the records around it keep theirs (the instance bits of this record repeat
                // the instance current before it).
                if ((F & F_RECT) && ALLOW_MEDIUM) {
                    const uint32_t box_inst = (F & F_INSTANCE) ? rec.prim2 : 0u;
                    LRay cl;
                    if (box_inst) cl = make_lray(to_local(load_const(s.instances + box_inst), r));
                    else if (cur_inst == 0u) cl = lr;
                    else cl = make_lray(r);
                    const float lx = rec.f[0], ly = rec.f[1], lz = rec.f[2], hx = rec.f[3], hy = rec.f[4], hz = rec.f[5];
                    face_pair_hit<0, 1, 2, true>(lz, hz, lx, hx, ly, hy, rec.prim, cl.r.o, cl.r.d, cl.rinv, t_min, closest, best);
                    face_pair_hit<0, 2, 1, true>(ly, hy, lx, hx, lz, hz, rec.prim + 2u, cl.r.o, cl.r.d, cl.rinv, t_min, closest, best);
                    face_pair_hit<1, 2, 0, true>(lx, hx, ly, hy, lz, hz, rec.prim + 4u, cl.r.o, cl.r.d, cl.rinv, t_min, closest, best);
                }
                break;
            case SC_SPHERE: if (F & F_SPHERE) sphere_accept<ALLOW_MEDIUM>(v3(rec.f), rec.f[3], rec.prim, lr.r, t_min, closest, best); break;
            case SC_SPHERE_MOVING:
                if (F & F_MOVING) {
                    // geo.rs:29-33 (the expression of sphere_center) from the record itself: f = (centre 0, radius, centre 1,
                    // time 0, time 1)
                    const V3 c0 = v3(rec.f), c1 = v3(rec.f + 4);
                    const V3 center = c0 + (lr.r.time - rec.f[7]) / (rec.f[8] - rec.f[7]) * (c1 - c0);
                    sphere_accept<ALLOW_MEDIUM>(center, rec.f[3], rec.prim, lr.r, t_min, closest, best);
                }
                break;
            case SC_MEDIUM:
                if (ALLOW_MEDIUM && (F & F_MEDIUM)) {
                    float t = 0.f;
                    if (medium_hit<F>(s, rec, r, t_min, closest, key, t)) {
                        closest = t;
                        best = (int)rec.prim;
                    }
                }
                break;
            default:
                // (an empty statement that "redefines" closest / best like every other case does: with all paths through
                // the switch ending in a definition of their own the register allocator keeps the two in place, where a
                // path that passes them through unchanged made it copy them in and out of every record)
#ifdef __HIP_DEVICE_COMPILE__
                asm volatile("" : "+v"(closest), "+v"(best));
#endif
                break;
        }
    }
}

// The two crossings ConstantMedium::hit asks its boundary for (geo.rs:415-416: hit(-inf, inf), then
// hit(t1 + 0.0001, inf)) when the boundary is one sphere: both calls of Sphere::hit (geo.rs:89-105) compute the same
// discriminant and the same two roots and differ only in the range they accept, so the roots are computed once and
// the two selections made on them — the same values bit for bit as two calls of sphere_hit_c.
MIRT_DEV bool sphere_crossings(V3 center, float radius, const Ray& r, float& t1, float& t2) {
    V3 oc = r.o - center;
    float a = norm_squared(r.d);
    float half_b = dot(oc, r.d);
    float c = norm_squared(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.f) return false;
    float sqrtd = sqrt_nr(discriminant);
    float ra = fast_rcp(a);
    const float root0 = div_nr(-half_b - sqrtd, a, ra), root1 = div_nr(-half_b + sqrtd, a, ra);
    // first call, range (-inf, inf): only a NaN root is outside (accepting form, see sphere_hit_c)
    const bool in0 = (root0 >= -MIRT_INF) & (root0 <= MIRT_INF), in1 = (root1 >= -MIRT_INF) & (root1 <= MIRT_INF);
    if (!(in0 | in1)) return false;
    t1 = in0 ? root0 : root1;
    const float lim = t1 + 0.0001f;
    const bool s0 = (root0 >= lim) & (root0 <= MIRT_INF), s1 = (root1 >= lim) & (root1 <= MIRT_INF);
    if (!(s0 | s1)) return false;
    t2 = s0 ? root0 : root1;
    return true;
}

// The distances at which a ray meets two consecutive rects of one axis (a scan record of two), +inf for a rect it
// misses: the arithmetic of rect_pair_hit without the range test. A NaN or infinite distance puts the hit point
// outside every extent, so it comes out as +inf as well.
template <uint32_t XI, uint32_t YI, uint32_t ZI>
MIRT_DEV void rect_pair_distances(const float* f, V3 o, V3 d, V3 rinv, float& ta, float& tb) {
    const float oz = comp(o, ZI), dz = comp(d, ZI), rz = comp(rinv, ZI);
    const float ox = comp(o, XI), dx = comp(d, XI), oy = comp(o, YI), dy = comp(d, YI);
    const f32x2 n = f32x2{f[0], f[1]} - oz;
    const f32x2 q = n * rz;
    const f32x2 e = __builtin_elementwise_fma(-q, (f32x2)dz, n);
    const f32x2 t = __builtin_elementwise_fma(e, (f32x2)rz, q);  // div_nr on both halves
    const f32x2 x = ox + t * dx;
    const f32x2 y = oy + t * dy;
    const bool ha = (x.x >= f[2]) & (x.x <= f[4]) & (y.x >= f[6]) & (y.x <= f[8]);
    const bool hb = (x.y >= f[3]) & (x.y <= f[5]) & (y.y >= f[7]) & (y.y <= f[9]);
    ta = ha ? t.x : MIRT_INF;
    tb = hb ? t.y : MIRT_INF;
}

// The same two crossings for a boundary of three rect-pair records in one frame (a cuboid, geo.rs:245-289): the
// six plane distances are computed once. The first scan of the list (range (-inf, inf), `t <= closest` narrowing as
// it goes) ends with the smallest distance among the rects the ray meets; the second with the smallest that is
// >= t1 + 0.0001 — two selections over the same six values the two scans would compute.
MIRT_DEV bool cuboid_crossings(const DevScene& s, uint32_t first, const Ray& r, float& t1, float& t2) {
    const DevScanRec r0 = load_const(s.scan + first);
    const uint32_t inst = r0.code_inst >> 8;
    const LRay lr = make_lray(inst ? to_local(load_const(s.instances + inst), r) : r);
    float c[6];
    rect_pair_distances<0, 1, 2>(r0.f, lr.r.o, lr.r.d, lr.rinv, c[0], c[1]);
    const DevScanRec r1 = load_const(s.scan + first + 1);
    rect_pair_distances<0, 2, 1>(r1.f, lr.r.o, lr.r.d, lr.rinv, c[2], c[3]);
    const DevScanRec r2 = load_const(s.scan + first + 2);
    rect_pair_distances<1, 2, 0>(r2.f, lr.r.o, lr.r.d, lr.rinv, c[4], c[5]);
    t1 = fminf(fminf(fminf(c[0], c[1]), fminf(c[2], c[3])), fminf(c[4], c[5]));
    if (!(t1 < MIRT_INF)) return false;
    const float lim = t1 + 0.0001f;
    t2 = MIRT_INF;
#pragma unroll
    for (int i = 0; i < 6; ++i) t2 = fminf(t2, c[i] >= lim ? c[i] : MIRT_INF);
    return t2 < MIRT_INF;
}

// ConstantMedium::hit (geo.rs:413-435). The boundary's two crossings come from one evaluation of its primitives
// where its shape allows (DevPrim::b[0] of a medium, set at upload): one static sphere, or the three rect pairs of a
// cuboid; any other boundary list is scanned twice, as the reference does.
template <uint32_t F>
MIRT_DEV bool medium_hit(const DevScene& s, const DevScanRec& m, const Ray& r, float t_min, float t_max, const DrawKey& key,
                         float& t_out) {
    // The medium's scan record carries what the test needs (filled at upload from the medium's primitive and its
    // boundary): f[0..3] the boundary sphere where the boundary is one, f[4] -1 / density, and as bits f[5] the medium's
    // index, f[6] the boundary's shape, f[7] the boundary sphere's instance, f[8], f[9] first record and number of
    // records of the boundary list — read through the primitive and the boundary's own record these were two more
    // dependent loads per medium and segment.
    const uint32_t bfirst = __float_as_uint(m.f[8]), bcount = __float_as_uint(m.f[9]);
    const uint32_t shape = __float_as_uint(m.f[6]);  // wave-uniform, like the record itself
    float t1 = MIRT_INF, t2 = MIRT_INF;
    if ((F & F_SPHERE) && shape == MS_SPHERE) {
        const uint32_t inst = (F & F_INSTANCE) ? __float_as_uint(m.f[7]) : 0u;
        if (!sphere_crossings(v3(m.f), m.f[3], inst ? to_local(load_const(s.instances + inst), r) : r, t1, t2)) return false;
    } else if ((F & F_RECT) && shape == MS_CUBOID) {
        if (!cuboid_crossings(s, bfirst, r, t1, t2)) return false;
    } else {
        int b = -1;
        list_scan<F, false>(s, bfirst, bcount, r, -MIRT_INF, key, t1, b);
        if (b < 0) return false;
        b = -1;
        list_scan<F, false>(s, bfirst, bcount, r, t1 + 0.0001f, key, t2, b);
        if (b < 0) return false;
    }
    t1 = fmaxf(fmaxf(t_min, t1), 0.f);
    t2 = fminf(t_max, t2);
    if (t1 >= t2) return false;
    float distance_inside_boundary = t2 - t1;
    uint32_t w0, w1, w2, w3;
    philox4x32_10(key.pixel, key.sample, 0u, key.bounce, STREAM_MEDIUM + __float_as_uint(m.f[5]), key.seed, w0, w1, w2, w3);
    // ln from the hardware log2 (1 ulp): the free path is a random draw, its last bit decides nothing
    float hit_distance = m.f[4] * (__builtin_amdgcn_logf(unit_f32(w0)) * 0.6931471805599453f);
    if (hit_distance > distance_inside_boundary) return false;
    t_out = t1 + hit_distance;
    return true;
}

// ---------------------------------------------------------------------------------------------
// flattened BVH (the device form of aabb.rs BVHNode): 128-byte nodes holding up to four child boxes,
// leaves as runs of 32-byte primitive records, per-lane traversal stack in LDS.
//
// Traversal is written as a resumable walk: `Walk` holds everything a lane needs to continue
// (current reference, stack depth, hierarchies started, closest hit so far), walk_node()
// consumes one interior reference and walk_leaf() one leaf reference; walk_run() loops until the
// walk is idle. (Interleaving the walk steps of a wave's lanes with shading rounds, so that lanes
// whose walk ended early need not wait for the longest one, was measured and lost 5-20 %: a shading
// round costs as much as ~25 node steps, and every extra round runs that code again.)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kWalkIdle = 0xffffffffu;   // no reference: the walk is over
constexpr uint32_t kRefLeaf = 0x80000000u;    // leaf reference: bit 31, (records - 1) << 27, first record
constexpr uint32_t kRefLeafIndexMask = (1u << 27) - 1u;

MIRT_DEV bool ref_is_leaf(uint32_t ref) { return ref != kWalkIdle && (ref & kRefLeaf); }
MIRT_DEV bool ref_is_node(uint32_t ref) { return !(ref & kRefLeaf); }

// What a lane's walk costs is tallied in one register while walk_some() runs: leaf steps in bits 0-9, leaf spheres
// intersected in bits 10-20, leaf rects intersected (six per cuboid record) in bits 21-31; with the lane's step count
// that also gives the interior nodes it visited. Only a launch that was asked to count adds the tallies to the wave's
// three words in LDS (64 lanes adding to one word is slow enough to keep out of the timed launches).
constexpr uint32_t kTallyLeafStep = 1u, kTallySphere = 1u << 10, kTallyRect = 1u << 21;

struct Walk {
    uint32_t ref;    // current reference (interior node index or leaf run) or kWalkIdle
    uint32_t sp;     // entries on this lane's stack
    uint32_t next;   // how many of the scene's hierarchies this ray has started
    float closest;   // t of the closest hit so far (starts at the list scans' result)
    int best;        // its primitive, -1 = none
};
// The ray in the frame of hierarchy number k (world order). It is derived from the path's ray whenever a walk
// starts or resumes instead of being carried in `Walk`: a walk that pauses over the shading code of an
// iteration then holds five registers, not fifteen.
MIRT_DEV DevBvhStart walk_start(const DevScene& s, uint32_t k) {
    // a scene with one hierarchy: the same record for every lane, a scalar load
    if (s.n_bvh_entries == 1u) return load_const(s.bvh_starts);
    union {
        DevBvhStart value;
        uint4 quads[2];
    } u;
    const uint4* q = reinterpret_cast<const uint4*>(s.bvh_starts + k);
    u.quads[0] = q[0];
    u.quads[1] = q[1];
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" : "+v"(u.quads[0].x), "+v"(u.quads[1].x));  // (whole here: no fetch moved into a branch)
#endif
    return u.value;
}
template <uint32_t F>
MIRT_DEV LRay walk_ray(const Ray& r, const DevBvhStart& st) {
    // 1-ulp reciprocals; walk_node() widens the far bound by more than that
    const DevInstance frame{st.sin_t, st.cos_t, st.ox, st.oy, st.oz, 0.f, 0.f, 0.f};
    LRay l = make_lray(((F & F_INSTANCE) && st.instance) ? to_local(frame, r) : r);
    // A direction component of (+-)0 has the reciprocal +-inf, and walk_node's  p * rinv + (-origin * rinv)  would
    // be inf - inf = NaN for every plane on the far side of the origin: min / max drop the NaN and a slab the ray
    // lies inside is taken for one it misses. With the reciprocal limited to +-2^100 the two products stay finite
    // for any coordinate below 2^27 and keep their signs, so an axis-parallel ray sees (-huge, +huge) for a slab
    // that contains it and two huge distances of one sign otherwise, which is what aabb.rs:43-60 computes with
    // infinities. (The leaf tests read the same reciprocal: a rect parallel to the ray gets a huge finite t
    // instead of inf or NaN and misses either way, since one of its two in-plane coordinates moves with t.)
    constexpr float kRinvMax = 1.2676506e30f;
    l.rinv = v3(__builtin_amdgcn_fmed3f(l.rinv.x, -kRinvMax, kRinvMax), __builtin_amdgcn_fmed3f(l.rinv.y, -kRinvMax, kRinvMax),
                __builtin_amdgcn_fmed3f(l.rinv.z, -kRinvMax, kRinvMax));
    l.nc = cmul(-l.r.o, l.rinv);
    return l;
}

// Number of hierarchies the ray has to walk. A ray with a NaN component hits no primitive (see
// sphere_hit_c) but would pass every box test (min / max drop NaN operands) and visit whole trees: it
// walks none.
MIRT_DEV uint32_t walk_count(const DevScene& s, const Ray& r) {
    const float nan_probe = (r.o.x + r.o.y + r.o.z) + (r.d.x + r.d.y + r.d.z);
    return nan_probe != nan_probe ? 0u : s.n_bvh_entries;
}

// Start the walk of the scene's hierarchy number w.next (world order) and count it as started.
template <uint32_t F>
MIRT_DEV void walk_begin(const DevScene& s, const Ray& r, Walk& w, LRay& lr) {
    const uint32_t k = w.next++;
    const DevBvhStart st = walk_start(s, k);
    lr = walk_ray<F>(r, st);
    w.ref = st.root;
    w.sp = 0;
}

// The hierarchies a ray starts next, without those whose world box it misses (DevScene::bvh_bounds; the slab test of
// aabb.rs:43-60 against [t_min, closest] on a padded box with the widened far bound of walk_node, so it errs on the
// side of walking). Most rays of `final` miss its cluster of a thousand spheres: they no longer change frames and
// fetch its root to find that out, and — the lanes of a wave reach their second hierarchy at different times — no
// longer cost the wave a node step each time one of them does.
MIRT_DEV void walk_skip_missed(const DevScene& s, const Ray& r, float t_min, Walk& w, uint32_t n) {
    constexpr float kRinvMax = 1.2676506e30f;
    const V3 rinv = v3(__builtin_amdgcn_fmed3f(fast_rcp(r.d.x), -kRinvMax, kRinvMax), __builtin_amdgcn_fmed3f(fast_rcp(r.d.y), -kRinvMax, kRinvMax),
                       __builtin_amdgcn_fmed3f(fast_rcp(r.d.z), -kRinvMax, kRinvMax));
    while (w.next < n) {
        const float4 lo = *reinterpret_cast<const float4*>(s.bvh_bounds + 8u * w.next);
        if (lo.w == 0.f) break;
        const float4 hi = *reinterpret_cast<const float4*>(s.bvh_bounds + 8u * w.next + 4u);
        const float ax = (lo.x - r.o.x) * rinv.x, bx = (hi.x - r.o.x) * rinv.x;
        const float ay = (lo.y - r.o.y) * rinv.y, by = (hi.y - r.o.y) * rinv.y;
        const float az = (lo.z - r.o.z) * rinv.z, bz = (hi.z - r.o.z) * rinv.z;
        const float tn = fmaxf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), t_min);
        const float tf = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), fminf(w.closest, 1e38f));
        if (tn <= tf * 1.000002f) break;
        ++w.next;
    }
}

// Next reference from the lane's stack; idle when the hierarchy is exhausted.
MIRT_DEV void walk_pop(Walk& w, const uint32_t* stack, uint32_t stride) {
    w.ref = kWalkIdle;
    if (w.sp) w.ref = stack[(--w.sp) * stride];
}

// BVHNode::hit (aabb.rs:112-126) for one four-wide node, i.e. for a node of the reference's binary tree and its
// interior children at once: test the (up to) four child boxes against [t_min, closest] (the slab test of
// aabb.rs:43-60), continue with the nearest hit child and push the other hit ones.
// The node stores its coordinates by plane (DevBvhNode4): per axis the lane loads the quad of the planes its ray
// enters through and the quad of those it leaves through — low and high bounds when the direction component is
// positive, the other way round when it is negative; the choice is a 16-byte address offset taken from the sign of
// the reciprocal, the same for every node of the walk. The 24 plane distances are twelve packed fused multiply-adds,
// t = p * rinv + (-origin * rinv), and a child's entry and exit distances are one three-way max and one three-way
// min of them: no min / max pairs to sort each slab's two distances.
// The fused form errs by 2^-24 (|origin * rinv| + |t|) instead of a few ulp of t; the boxes are stored padded by
// 2^-21 of the hierarchy's largest coordinate, which covers the first term for origins within the scene, and the far
// bound is widened by 16 ulp, which covers the second, the first for origins far outside (there |origin * rinv| is
// about |t|) and the 1-ulp reciprocals — so the test stays conservative and the closest hit is unchanged.
// The nearest child comes out of a tournament of three comparisons on (entry distance, reference) pairs — (0, 1),
// (2, 3), then the winners — whose losers are the references to push: no sorting network, no index arithmetic. The
// pair losers go on the stack first and the loser of the final last, so the second-nearest is usually popped next.
// `stack` is this lane's column of an LDS array laid out [level][lane] (stride = blockDim.x words): pushes and pops
// are conflict-free.
// Entry distance of one child box from its three entry and three exit distances, +inf when the ray misses it or the
// child does not exist. A missing child is stored as the point (+inf, +inf, +inf): its distances are +inf on the
// axes the ray ascends and -inf on the others, so its entry distance is +inf or its exit distance -inf, and with the
// far bound `closest` capped at a finite value by the caller neither passes — no test of the reference is needed.
MIRT_DEV float child_entry(float nx, float ny, float nz, float fx, float fy, float fz, float t_min, float closest) {
    const float tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), t_min);
    const float tf = fminf(fminf(fminf(fx, fy), fz), closest);
    return tn <= tf * 1.000002f ? tn : MIRT_INF;
}
// A conditional push without a branch: the reference is always stored at the top of the stack and the stack pointer
// moves only when `on` is set (three pushes of a node step are three stores and three predicated increments
// instead of three divergent regions). The stack has three levels beyond s.stack_limit for the stores of a step
// that starts at the limit; walk_node() checks the limit once, after the last push.
MIRT_DEV void walk_push(Walk& w, uint32_t* stack, uint32_t stride, bool on, uint32_t ref) {
    stack[w.sp * stride] = ref;
    w.sp += on ? 1u : 0u;
}
// `top`: the workgroup's LDS copy of nodes [0, s.lds_nodes), kLdsNodeStride bytes apart, or nullptr. The array is
// ordered breadth first, so these are the nodes nearest the roots; a lane whose node is among them reads it from LDS
// and leaves the vector L1 — whose throughput, at 112 bytes per lane and step, bounds the walks of large trees beside
// instruction issue — to the lanes deeper down.
template <bool COUNT = false>
MIRT_DEV void walk_node(const DevScene& s, float t_min, V3 rinv, V3 nc, Walk& w, uint32_t* stack, uint32_t stride, uint32_t* wave_cost,
                        const uint32_t* top) {
    // Byte offset of the entry quad inside an axis' pair of quads: 0 (low planes) where the ray ascends, 16 (high
    // planes) where it descends. Invariant over the walk: the compiler keeps the three values across the node loop.
    // (A reciprocal of -0 was limited to -2^100 by walk_ray: its sign picks the high planes, and the products say the
    // same: the distances to the high planes are the smaller ones.)
    // (each with the offset of its axis' pair of quads inside the node: only bit operations between the node's offset and
    // these, so every fetch is base + one 32-bit offset register)
    const uint32_t sx = (__float_as_uint(rinv.x) >> 31) << 4, sy = ((__float_as_uint(rinv.y) >> 31) << 4) | 32u, sz = ((__float_as_uint(rinv.z) >> 31) << 4) | 64u;
    const char* const nodes = reinterpret_cast<const char*>(s.bvh_nodes);
    const uint32_t at = w.ref << 7;  // byte offset of the node: the upload refuses more than 2^25 nodes
    const uint32_t ax = at | sx, ay = at | sy, az = at | sz;
    float4 px, qx, py, qy, pz, qz;
    uint4 refs;
    if (top && w.ref < s.lds_nodes) {
#ifdef __HIP_DEVICE_COMPILE__
        // (`top` arrives as a pointer of no known address space; the low half of such a pointer into LDS is the LDS
        // address, and with it the fetches are LDS reads instead of flat ones with 64-bit addresses)
        typedef const __attribute__((address_space(3))) char* LdsBytes;
        typedef const __attribute__((address_space(3))) float4* LdsQuad;
        LdsBytes const n = (LdsBytes)(uintptr_t)top + w.ref * kLdsNodeStride;
        px = *(LdsQuad)(n + sx); qx = *(LdsQuad)(n + (sx ^ 16u));
        py = *(LdsQuad)(n + sy); qy = *(LdsQuad)(n + (sy ^ 16u));
        pz = *(LdsQuad)(n + sz); qz = *(LdsQuad)(n + (sz ^ 16u));
        refs = *(const __attribute__((address_space(3))) uint4*)(n + 96u);
        // (keeps the compiler from merging these fetches with the other branch's into one set of flat loads behind the
        // branches, each with an address of either kind chosen by selects)
        asm volatile("" : "+v"(refs.x));
#else
        const char* const n = reinterpret_cast<const char*>(top) + w.ref * kLdsNodeStride;
        px = *reinterpret_cast<const float4*>(n + sx); qx = *reinterpret_cast<const float4*>(n + (sx ^ 16u));
        py = *reinterpret_cast<const float4*>(n + sy); qy = *reinterpret_cast<const float4*>(n + (sy ^ 16u));
        pz = *reinterpret_cast<const float4*>(n + sz); qz = *reinterpret_cast<const float4*>(n + (sz ^ 16u));
        refs = *reinterpret_cast<const uint4*>(n + 96u);
#endif
    } else {
        px = *reinterpret_cast<const float4*>(nodes + ax); qx = *reinterpret_cast<const float4*>(nodes + (ax ^ 16u));
        py = *reinterpret_cast<const float4*>(nodes + ay); qy = *reinterpret_cast<const float4*>(nodes + (ay ^ 16u));
        pz = *reinterpret_cast<const float4*>(nodes + az); qz = *reinterpret_cast<const float4*>(nodes + (az ^ 16u));
        refs = *reinterpret_cast<const uint4*>(nodes + at + 96u);
    }
#ifdef MIRT_PROBE_LOADS  // measurement builds only: what three more 16-byte loads per lane and node step cost (same line: no new data)
    if (!(top && w.ref < s.lds_nodes)) {
        float4 e0 = *reinterpret_cast<const float4*>(nodes + (ax ^ 16u)), e1 = *reinterpret_cast<const float4*>(nodes + (ay ^ 16u)),
               e2 = *reinterpret_cast<const float4*>(nodes + (az ^ 16u));
        asm volatile("" : "+v"(e0.x), "+v"(e1.x), "+v"(e2.x), "+v"(px.x), "+v"(refs.x));
        asm volatile("" : : "v"(e0.y), "v"(e1.y), "v"(e2.y));
    }
#endif
    // a counting launch tallies the child boxes the node really has: its first two, and the third and fourth where
    // they exist (COUNT is a template parameter: a test of the pointer here is three instructions per step for every
    // launch — a branch on the execution mask, since the compiler cannot know the pointer is the same in all lanes)
    if constexpr (COUNT) atomicAdd(wave_cost + 3, 2u + (refs.z != kWalkIdle ? 1u : 0u) + (refs.w != kWalkIdle ? 1u : 0u));
    const f32x2 rx = {rinv.x, rinv.x}, ry = {rinv.y, rinv.y}, rz = {rinv.z, rinv.z};
    const f32x2 cx = {nc.x, nc.x}, cy = {nc.y, nc.y}, cz = {nc.z, nc.z};
    // entry (n) and exit (f) distances of children (0, 1) and (2, 3) on each axis
    const f32x2 nx01 = __builtin_elementwise_fma(f32x2{px.x, px.y}, rx, cx), nx23 = __builtin_elementwise_fma(f32x2{px.z, px.w}, rx, cx);
    const f32x2 fx01 = __builtin_elementwise_fma(f32x2{qx.x, qx.y}, rx, cx), fx23 = __builtin_elementwise_fma(f32x2{qx.z, qx.w}, rx, cx);
    const f32x2 ny01 = __builtin_elementwise_fma(f32x2{py.x, py.y}, ry, cy), ny23 = __builtin_elementwise_fma(f32x2{py.z, py.w}, ry, cy);
    const f32x2 fy01 = __builtin_elementwise_fma(f32x2{qy.x, qy.y}, ry, cy), fy23 = __builtin_elementwise_fma(f32x2{qy.z, qy.w}, ry, cy);
    const f32x2 nz01 = __builtin_elementwise_fma(f32x2{pz.x, pz.y}, rz, cz), nz23 = __builtin_elementwise_fma(f32x2{pz.z, pz.w}, rz, cz);
    const f32x2 fz01 = __builtin_elementwise_fma(f32x2{qz.x, qz.y}, rz, cz), fz23 = __builtin_elementwise_fma(f32x2{qz.z, qz.w}, rz, cz);
    // (no hit lies beyond 1e38 for coordinates below 2^27 and the limited reciprocals of walk_ray)
    const float far_bound = fminf(w.closest, 1e38f);
    const float k0 = child_entry(nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x, t_min, far_bound);
    const float k1 = child_entry(nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y, t_min, far_bound);
    const float k2 = child_entry(nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x, t_min, far_bound);
    const float k3 = child_entry(nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y, t_min, far_bound);
    const bool f01 = k0 <= k1, f23 = k2 <= k3;
    const float ka = f01 ? k0 : k1, la = f01 ? k1 : k0, kb = f23 ? k2 : k3, lb = f23 ? k3 : k2;
    const uint32_t ra = f01 ? refs.x : refs.y, pa = f01 ? refs.y : refs.x, rb = f23 ? refs.z : refs.w, pb = f23 ? refs.w : refs.z;
    const bool fab = ka <= kb;
    const float kn = fab ? ka : kb, lf = fab ? kb : ka;
    const uint32_t rn = fab ? ra : rb, pf = fab ? rb : ra;
    walk_push(w, stack, stride, la < MIRT_INF, pa);
    walk_push(w, stack, stride, lb < MIRT_INF, pb);
    walk_push(w, stack, stride, lf < MIRT_INF, pf);
    // (the host sizes the stacks by the exact need of every hierarchy it builds; a stack beyond its limit means that
    // promise was broken, and the dropped children are reported instead of silently missed)
    // (asked of the whole wave first: a compare and a scalar branch in the step instead of a region of its own)
    if (__builtin_expect(__ballot(w.sp > s.stack_limit) != 0, 0)) {
        if (w.sp > s.stack_limit) {
            w.sp = s.stack_limit;
            atomicOr(s.fault, (uint32_t)FAULT_STACK_OVERFLOW);
        }
    }
    if (kn < MIRT_INF) w.ref = rn;
    else walk_pop(w, stack, stride);
}

// One primitive record of a leaf (the record's kind differs from lane to lane, so the rectangle's axes
// are chosen with per-lane selects). Same tests as the list scan: geo.rs:89-105, geo.rs:196-204. `mv` is the
// DevLeafMoving beside the record (read only for a moving sphere).
template <uint32_t F>
MIRT_DEV bool leaf_rec_hit(const DevLeafRec& rec, const DevLeafMoving& mv, bool any_moving, const LRay& l, float t_min, float t_max,
                           float& t_out) {
    const uint32_t code = rec.code_inst & 0xffu;
    if ((F & F_SPHERE) && (F & F_MOVING) && !(F & F_RECT)) {
        // The variants of spheres only: one sphere test for the still and the moving spheres of a wave (most waves of
        // the reference's sphere scenes hold both, and ran the test twice). Where the scene has moving spheres in a
        // hierarchy (`any_moving`, the same for every lane) the centre at the ray's time — geo.rs:29-33, the expression
        // of sphere_center — is computed for every lane and kept by those whose record is a moving sphere (a still
        // sphere's DevLeafMoving is all zero: its quotient is a NaN that the select drops). `random` +4.5 %; the
        // variant with everything keeps the two branches below (one more spilled register there: -1 %).
        V3 center = v3(rec.f);
        if (any_moving) {
            const V3 c1 = v3(mv.center1);
            const V3 at_time = center + (l.r.time - mv.time0) / (mv.time1 - mv.time0) * (c1 - center);
            center = select(code == SC_SPHERE_MOVING, at_time, center);
        }
        return sphere_hit_c(center, rec.f[3], l.r, t_min, t_max, t_out);
    }
    if ((F & F_SPHERE) && (!(F & (F_RECT | F_MOVING)) || code == SC_SPHERE))
        return sphere_hit_c(v3(rec.f), rec.f[3], l.r, t_min, t_max, t_out);
    if ((F & F_MOVING) && code == SC_SPHERE_MOVING) {
        // geo.rs:29-33, the expression of sphere_center
        const V3 c0 = v3(rec.f), c1 = v3(mv.center1);
        const V3 center = c0 + (l.r.time - mv.time0) / (mv.time1 - mv.time0) * (c1 - c0);
        return sphere_hit_c(center, rec.f[3], l.r, t_min, t_max, t_out);
    }
    if ((F & F_RECT) && code >= SC_RECT_XY && code <= SC_RECT_YZ) {
        uint32_t xi, yi, zi;
        axis_indexes(code - SC_RECT_XY, xi, yi, zi);
        float t = div_nr(rec.f[0] - comp(l.r.o, zi), comp(l.r.d, zi), comp(l.rinv, zi));
        float x = comp(l.r.o, xi) + t * comp(l.r.d, xi);
        float y = comp(l.r.o, yi) + t * comp(l.r.d, yi);
        t_out = t;
        return (t >= t_min) & (t <= t_max) & (x >= rec.f[1]) & (x <= rec.f[2]) & (y >= rec.f[3]) & (y <= rec.f[4]);
    }
    return false;
}

// A leaf's 32-byte record — and, where the scene has moving spheres in a hierarchy, the 32 bytes beside it — fetched
// whole, as 16-byte loads issued together, into values: a record read field by field through a reference is fetched
// piece by piece where the pieces are used — the kind first, the rest behind the branch on the kind, a moving sphere's
// primitive behind that — each a round trip to memory of its own in a step that consists of little else.
template <bool WITH_MOVING>
MIRT_DEV void load_leaf_record(const DevLeafRec* p, const DevLeafMoving* pm, DevLeafRec& rec, DevLeafMoving& mv) {
    union {
        DevLeafRec value;
        uint4 quads[2];
    } u;
    union {
        DevLeafMoving value;
        uint4 quads[2];
    } um;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    u.quads[0] = q[0];
    u.quads[1] = q[1];
    um.quads[0] = um.quads[1] = make_uint4(0u, 0u, 0u, 0u);
    if (WITH_MOVING && pm) {  // (the same pointer in every lane)
        const uint4* qm = reinterpret_cast<const uint4*>(pm);
        um.quads[0] = qm[0];
        um.quads[1] = qm[1];
    }
#ifdef __HIP_DEVICE_COMPILE__
    // (the values are complete here: the compiler may not move a quad's fetch down into the branch that reads it)
    if (WITH_MOVING) asm volatile("" : "+v"(u.quads[0].x), "+v"(u.quads[1].x), "+v"(um.quads[0].x), "+v"(um.quads[1].x));
    else asm volatile("" : "+v"(u.quads[0].x), "+v"(u.quads[1].x));
#endif
    rec = u.value;
    mv = um.value;
}

// The records of one leaf, in object order (a later record wins a tie, as in ray.rs:123-133).
template <uint32_t F>
MIRT_DEV void walk_leaf(const DevScene& s, float t_min, const LRay& lr, Walk& w, const uint32_t* stack, uint32_t stride, uint32_t& tally) {
    const uint32_t first = w.ref & kRefLeafIndexMask, n = ((w.ref >> 27) & 15u) + 1u;
    for (uint32_t j = 0; j < n; ++j) {
        DevLeafRec rec;
        DevLeafMoving mv;
        load_leaf_record<(F & F_MOVING) != 0>(s.bvh_leaves + first + j, s.bvh_leaf_moving ? s.bvh_leaf_moving + first + j : nullptr, rec, mv);
        if ((F & F_RECT) && (rec.code_inst & 0xffu) == SC_CUBOID) {
            tally += 6u * kTallyRect;
            // Cuboid (geo.rs:245-289): its six rects in list order — XY at z lo / hi, XZ at y lo / hi, YZ at
            // x lo / hi — from the two corners f = (lo.xyz, hi.xyz); ids prim .. prim + 5
            const float lx = rec.f[0], ly = rec.f[1], lz = rec.f[2], hx = rec.f[3], hy = rec.f[4], hz = rec.f[5];
            face_pair_hit<0, 1, 2>(lz, hz, lx, hx, ly, hy, rec.prim, lr.r.o, lr.r.d, lr.rinv, t_min, w.closest, w.best);
            face_pair_hit<0, 2, 1>(ly, hy, lx, hx, lz, hz, rec.prim + 2u, lr.r.o, lr.r.d, lr.rinv, t_min, w.closest, w.best);
            face_pair_hit<1, 2, 0>(lx, hx, ly, hy, lz, hz, rec.prim + 4u, lr.r.o, lr.r.d, lr.rinv, t_min, w.closest, w.best);
            continue;
        }
        const uint32_t code = rec.code_inst & 0xffu;
        tally += ((F & F_RECT) && code >= SC_RECT_XY && code <= SC_RECT_YZ) ? kTallyRect : kTallySphere;
        float t;
        if (leaf_rec_hit<F>(rec, mv, s.bvh_leaf_moving != nullptr, lr, t_min, w.closest, t)) {
            w.closest = t;
            w.best = (int)rec.prim;
        }
    }
    walk_pop(w, stack, stride);
}

// Walk every hierarchy of the scene to the end.
template <uint32_t F>
MIRT_DEV void walk_run(const DevScene& s, const Ray& r, float t_min, Walk& w, uint32_t* stack, uint32_t stride, const uint32_t* top = nullptr) {
    const uint32_t n = walk_count(s, r);
    w.next = 0;
    w.ref = kWalkIdle;
    LRay lr;
    uint32_t tally = 0;  // not reported by the callers of whole walks
    for (;;) {
        if (s.bvh_bounds) walk_skip_missed(s, r, t_min, w, n);
        if (w.next >= n) break;
        walk_begin<F>(s, r, w, lr);
        // Two nested loops rather than one step of either kind per iteration: the lanes of a wave
        // descend through interior nodes until each of them holds a leaf (or is done), then all
        // intersect their leaves together, so the long leaf code runs with many lanes active.
        while (w.ref != kWalkIdle) {
            while (ref_is_node(w.ref)) walk_node(s, t_min, lr.rinv, lr.nc, w, stack, stride, nullptr, top);
            if (ref_is_leaf(w.ref)) walk_leaf<F>(s, t_min, lr, w, stack, stride, tally);
        }
    }
}

// A bounded piece of the same walk, continuing where the lane's last call stopped (w.next hierarchies started,
// w.ref / stack as left); returns true when the walk is complete. A render loop that calls this once per
// iteration lets the lanes whose walk is over shade and start their next segment while a lane with a long
// walk keeps its place and goes on in the next iteration, instead of the whole wave waiting for it. The walk
// code is bound by instruction issue at a fifth of the lanes active, so three things end a turn early:
//   budget      the lane has taken this many node / leaf steps;
//   min_lanes   fewer lanes of the wave than this are still walking: they stop too, and continue together
//               with the walks the other lanes will have started by the next iteration;
//   node_lanes  fewer lanes than this are still descending through interior nodes: the descent pauses and the
//               lanes that hold leaves intersect them now rather than after the last straggler.
// `steps` receives the number of steps taken (statistics builds read it; otherwise it folds away).
template <uint32_t F, bool COUNT = false>
MIRT_DEV bool walk_some(const DevScene& s, const Ray& r, float t_min, Walk& w, uint32_t n, uint32_t budget, uint32_t min_lanes,
                        uint32_t node_lanes, uint32_t* stack, uint32_t stride, uint32_t& steps, uint32_t* wave_cost,
                        const uint32_t* top = nullptr) {
    steps = 0;
    uint32_t tally = 0, tallied_steps = 0;
    // (a counting launch only) the lane's tally goes to the wave's words: at the end of the turn, and whenever its
    // leaf-step field is a sixteenth full — a turn without a step budget can outlast the fields otherwise
    auto flush_tally = [&]() {
        atomicAdd(wave_cost + 0, (steps - tallied_steps) - (tally & 1023u));
        atomicAdd(wave_cost + 1, (tally >> 10) & 2047u);
        atomicAdd(wave_cost + 2, tally >> 21);
        tallied_steps = steps;
        tally = 0;
    };
    LRay lr = make_lray(r);
    if (w.ref != kWalkIdle) lr = walk_ray<F>(r, walk_start(s, w.next - 1u));  // resuming inside hierarchy w.next - 1
    for (;;) {
        if (s.bvh_bounds && w.ref == kWalkIdle) walk_skip_missed(s, r, t_min, w, n);
        if (w.ref == kWalkIdle && w.next < n) walk_begin<F>(s, r, w, lr);
        while (w.ref != kWalkIdle && steps < budget) {
            while (ref_is_node(w.ref) && steps < budget) {
                walk_node<COUNT>(s, t_min, lr.rinv, lr.nc, w, stack, stride, wave_cost, top);
                ++steps;
                if ((uint32_t)__popcll(__ballot(ref_is_node(w.ref))) < node_lanes) break;
            }
            if (ref_is_leaf(w.ref)) {
                walk_leaf<F>(s, t_min, lr, w, stack, stride, tally);
                tally += kTallyLeafStep;
                ++steps;
                if (COUNT && (tally & 1023u) >= 64u) flush_tally();
            }
            // few lanes left walking: they stop here too and go on next iteration, together with the walks
            // the other lanes will have started by then
            if ((uint32_t)__popcll(__ballot(w.ref != kWalkIdle)) < min_lanes) budget = steps;
        }
        if (w.ref != kWalkIdle || w.next >= n) break;  // out of budget, or every hierarchy walked
    }
    if (COUNT) flush_tally();
    return w.ref == kWalkIdle;
}

// The list entries of the world, in world order (HittableList::hit, ray.rs:123-133).
template <uint32_t F>
MIRT_DEV void world_lists(const DevScene& s, const Ray& r, const DrawKey& key, float& closest, int& best) {
    for (uint32_t e = 0; e < s.n_entries; ++e) {
        const DevEntry en = load_const(s.entries + e);
        if (!(F & F_BVH) || en.kind != EK_BVH) list_scan<F, true>(s, en.scan_first, en.scan_count, r, 0.001f, key, closest, best);
    }
}

// scene.world.hit(r, 0.001, inf) — renderer.rs:201. The list entries are scanned first and the
// hierarchy entries walked after them, each group in world order, so the hierarchies start with the
// lists' closest hit as their far bound. The closest hit does not depend on the order; only an exact
// tie between a list member and a hierarchy member placed before it would go to the other primitive.
template <uint32_t F>
MIRT_DEV void world_hit(const DevScene& s, const Ray& r, const DrawKey& key, uint32_t* stack, uint32_t stride, float& t_out,
                        int& prim_out) {
    float closest = MIRT_INF;
    int best = -1;
    world_lists<F>(s, r, key, closest, best);
    if (F & F_BVH) {
        Walk w;
        w.closest = closest;
        w.best = best;
        walk_run<F>(s, r, 0.001f, w, stack, stride);
        closest = w.closest;
        best = w.best;
    }
    t_out = closest;
    prim_out = best;
}

// ---------------------------------------------------------------------------------------------
// hit record reconstruction (HitRecord of ray.rs:50-81 as filled by geo.rs)
// ---------------------------------------------------------------------------------------------
struct HitInfo {
    V3 p, n;
    float u, v;
    bool front;
};

template <uint32_t F>
// `need_uv`: the surface's texture reads u, v (only image textures do); the record keeps 0, 0 otherwise, so the
// many hits that never look at them do not pay for acos / atan2.
MIRT_DEV HitInfo finalize_hit(const Tables& tb, const Ray& r, float t, int prim, bool need_uv = true) {
    // (the variants with hierarchies read the primitive from global memory: whole, its three quads fetched together —
    // field by field a moving sphere is three dependent fetches, flags, then second centre, then radius)
    DevPrim whole;
    if constexpr ((F & F_BVH) != 0) {
        union {
            DevPrim value;
            uint4 quads[3];
        } u;
        const uint4* q = reinterpret_cast<const uint4*>(tb.prims + prim);
        u.quads[0] = q[0];
        u.quads[1] = q[1];
        u.quads[2] = q[2];
#ifdef __HIP_DEVICE_COMPILE__
        asm volatile("" : "+v"(u.quads[0].x), "+v"(u.quads[1].x), "+v"(u.quads[2].x));
#endif
        whole = u.value;
    }
    const DevPrim& p = (F & F_BVH) ? whole : tb.prims[prim];
    HitInfo h;
    h.u = 0.f;
    h.v = 0.f;
    uint32_t kind = p.meta & PM_KIND_MASK;
    if ((F & F_MEDIUM) && kind == PK_MEDIUM) {
        // geo.rs:427-431: default record — normal +y, front_face false, uv 0
        h.p = ray_at(r, t);
        h.n = v3(0.f, 1.f, 0.f);
        h.front = false;
        return h;
    }
    Ray lr = r;
    uint32_t inst = (F & F_INSTANCE) ? p.instance : 0u;
    if (inst) lr = to_local(tb.instances[inst], r);
    V3 pl = ray_at(lr, t);
    V3 nl;
    bool front;
    if ((F & F_SPHERE) && (!(F & F_RECT) || kind == PK_SPHERE)) {
        // geo.rs:108-111
        V3 outward = (pl - sphere_center<F>(p, lr.time)) * fast_rcp(p.a[3]);
        if ((F & F_TEX_IMAGE) && need_uv) {
            float theta = acosf(-outward.y);                    // geo.rs:71-75
            float phi = atan2f(-outward.z, outward.x) + kPi;
            h.u = phi / (2.f * kPi);
            h.v = theta / kPi;
        }
        V3 on = fnormalize(outward);
        front = fdot(lr.d, on) < 0.f;
        nl = select(front, on, -on);
    } else {
        // geo.rs:205-211
        uint32_t xi, yi, zi;
        axis_indexes((p.meta & PM_AXIS_MASK) >> PM_AXIS_SHIFT, xi, yi, zi);
        if ((F & F_TEX_IMAGE) && need_uv) {
            h.u = (comp(pl, xi) - p.a[1]) / (p.a[2] - p.a[1]);
            h.v = (comp(pl, yi) - p.a[3]) / (p.b[0] - p.a[3]);
        }
        front = comp(lr.d, zi) < 0.f;
        float sgn = front ? 1.f : -1.f;
        nl = v3(zi == 0 ? sgn : 0.f, zi == 1 ? sgn : 0.f, zi == 2 ? sgn : 0.f);
    }
    if (p.meta & PM_FLIP) front = !front;  // geo.rs:455-459: the flag flips, the normal does not
    if (inst) {
        const DevInstance& in = tb.instances[inst];
        h.p = rot_y(in, pl) + v3(in.ox, in.oy, in.oz);  // geo.rs:377-380 then geo.rs:309
        h.n = rot_y(in, nl);
    } else {
        h.p = pl;
        h.n = nl;
    }
    h.front = front;
    return h;
}

// ---------------------------------------------------------------------------------------------
// textures (texture.rs, perlin.rs)
// ---------------------------------------------------------------------------------------------
// perlin.rs:24-63. The reference's corner weight  i*u + (1-i)*(1-u)  and offset  u - i  are written out for
// i = 0 and i = 1 (1 - u, u and u, u - 1): the same values bit for bit (0*x + y and 1*x are exact for finite
// x), without the multiplications by constants 0 and 1 that IEEE semantics forbid the compiler to drop, and
// with the product of the first two weights shared by the two corners that differ in the third index only.
MIRT_DEV float perlin_noise(const DevPerlin& pl, V3 p) {
    const float fl[3] = {floorf(p.x), floorf(p.y), floorf(p.z)};
    const int ix = (int)fl[0], iy = (int)fl[1], iz = (int)fl[2];
    float uvw[3];
    const float pc[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float x = pc[a] - fl[a];
        uvw[a] = x * x * fma_(-2.f, x, 3.f);
    }
    const float wu[2] = {1.f - uvw[0], uvw[0]}, wv[2] = {1.f - uvw[1], uvw[1]}, ww[2] = {1.f - uvw[2], uvw[2]};
    const float du[2] = {uvw[0], uvw[0] - 1.f}, dv[2] = {uvw[1], uvw[1] - 1.f}, dw[2] = {uvw[2], uvw[2] - 1.f};
    const int px[2] = {pl.perm[0][ix & 0xff], pl.perm[0][(ix + 1) & 0xff]};
    const int py[2] = {pl.perm[1][iy & 0xff], pl.perm[1][(iy + 1) & 0xff]};
    const int pz[2] = {pl.perm[2][iz & 0xff], pl.perm[2][(iz + 1) & 0xff]};
    // four gradients at a time, the arithmetic after them: four loads in flight instead of four round trips (all
    // eight at once cost more registers than the hierarchy variants of the render kernel have to spare)
    float accm = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 grad[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) grad[c] = *reinterpret_cast<const float4*>(pl.grad[px[i] ^ py[c >> 1] ^ pz[c & 1]]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float puv = wu[i] * wv[j];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 c = grad[j * 2 + k];
                accm = fma_(fma_(c.z, dw[k], fma_(c.y, dv[j], c.x * du[i])), puv * ww[k], accm);
            }
        }
    }
    return accm;
}
// perlin.rs:65-71
MIRT_DEV float perlin_turb(const DevPerlin& pl, V3 p) {
    float accum = 0.f, weight = 1.f;
    V3 tp = p;
    for (int i = 0; i < 7; ++i) {
        accum = fma_(weight, perlin_noise(pl, tp), accum);
        weight = 0.5f * weight;
        tp = 2.f * tp;
    }
    return fabsf(accum);
}

// `lds_perlin`: the workgroup's copy of the scene's first set of noise tables, where the kernel keeps one (nullptr
// elsewhere; a compile-time fact to every caller).
template <uint32_t F>
MIRT_DEV V3 texture_value(const DevScene& s, uint32_t tex, float u, float v, V3 p, const DevPerlin* lds_perlin = nullptr) {
    const DevTexture& t = s.textures[tex];
    uint32_t kind = t.kind;
    if ((F & F_TEX_CHECKER) && kind == TK_CHECKER) {
        // texture.rs:42-46
        float sines = 1.0f;
        sines = sines * fast_sin(10.f * p.x);
        sines = sines * fast_sin(10.f * p.y);
        sines = sines * fast_sin(10.f * p.z);
        return sines < 0.f ? v3(t.c1) : v3(t.c0);
    }
    if ((F & F_TEX_NOISE) && kind == TK_NOISE) {
        // texture.rs:64-68
        const float turb = (lds_perlin && t.perlin == 0u) ? perlin_turb(*lds_perlin, p) : perlin_turb(s.perlins[t.perlin], p);
        float g = fma_(0.5f, fast_sin(fma_(t.c1[3], p.z, 10.f * turb)), 0.5f);
        return v3(g, g, g);
    }
    if ((F & F_TEX_IMAGE) && kind == TK_IMAGE) {
        // texture.rs:91-102
        const DevImage& im = s.images[t.image];
        if (im.width == 0) return v3(0.f, 1.f, 1.f);
        float cu = fminf(fmaxf(u, 0.f), 1.f), cv = fminf(fmaxf(v, 0.f), 1.f);
        uint32_t x = min((uint32_t)(cu * (float)im.width), im.width - 1);
        uint32_t y = min((uint32_t)(cv * (float)im.height), im.height - 1);
        uchar4 px = in_global_memory(reinterpret_cast<const uchar4*>(im.rgba))[(size_t)y * im.width + x];
        const float sc = 1.f / 255.f;
        return v3((float)px.x * sc, (float)px.y * sc, (float)px.z * sc);
    }
    return v3(t.c0);  // texture.rs:21-25
}

// ---------------------------------------------------------------------------------------------
// light list: HittableList::{pdf_val, random} (ray.rs:149-154) over scene.lights
// ---------------------------------------------------------------------------------------------
// geo.rs:223-233 (rect) and geo.rs:136-143 (sphere). Instances are not applied and FlipFace is
// transparent, as in geo.rs:324-329 and geo.rs:465-470. `p` is wave-uniform (scalar registers).
// geo.rs:223-233 for a rect of one axis (XI, YI, ZI = geo.rs:160-166 as compile-time constants: the light is the same for
// the whole wave, so its axis is a scalar branch and the components need no per-lane selects).
template <uint32_t XI, uint32_t YI, uint32_t ZI>
MIRT_DEV float rect_light_pdf(const DevPrim& p, V3 origin, V3 v, V3 vinv) {
    float t = (p.a[0] - comp(origin, ZI)) * comp(vinv, ZI);
    float x = fma_(t, comp(v, XI), comp(origin, XI));
    float y = fma_(t, comp(v, YI), comp(origin, YI));
    bool h = (t >= 0.001f) & (x >= p.a[1]) & (x <= p.a[2]) & (y >= p.a[3]) & (y <= p.b[0]);
    float area = fabsf((p.a[2] - p.a[1]) * (p.b[0] - p.a[3]));
    float dis_squared = t * t * fnorm_squared(v);
    float cosine = fabsf(comp(v, ZI));
    return h ? dis_squared * fast_rcp(cosine * area) : 0.f;
}
template <uint32_t F>
MIRT_DEV float light_pdf_one(const DevPrim& p, V3 origin, V3 v, V3 vinv) {
    uint32_t kind = p.meta & PM_KIND_MASK;
    if ((F & F_RECT) && kind == PK_RECT) {
        const uint32_t axis = (p.meta & PM_AXIS_MASK) >> PM_AXIS_SHIFT;
        if (axis == 0u) return rect_light_pdf<0, 1, 2>(p, origin, v, vinv);
        if (axis == 1u) return rect_light_pdf<0, 2, 1>(p, origin, v, vinv);
        return rect_light_pdf<1, 2, 0>(p, origin, v, vinv);
    }
    if ((F & F_SPHERE) && kind == PK_SPHERE) {
        // probe ray at time 0 (geo.rs:137); nearest root in [0.001, inf)
        V3 c = sphere_center<F>(p, 0.f);
        float radius = p.a[3];
        V3 oc = origin - c;
        float a = fnorm_squared(v);
        float half_b = fdot(oc, v);
        float cc = fma_(-radius, radius, fnorm_squared(oc));
        float disc = fma_(half_b, half_b, -(a * cc));
        float sqrtd = fast_sqrt(disc);
        float ra = fast_rcp(a);
        float root0 = (-half_b - sqrtd) * ra, root1 = (-half_b + sqrtd) * ra;
        bool h = (disc >= 0.f) & ((root0 >= 0.001f) | (root1 >= 0.001f));
        V3 c0 = v3(p.a[0], p.a[1], p.a[2]);
        float cos_theta_max = fast_sqrt(fma_(-(radius * radius), fast_rcp(fnorm_squared(c0 - origin)), 1.f));
        float solid_angle = 2.f * kPi * (1.f - cos_theta_max);
        return h ? fast_rcp(solid_angle) : 0.f;  // NaN when the origin is inside the sphere (geo.rs:138)
    }
    return 0.f;
}

template <uint32_t F>
MIRT_DEV float lights_pdf(const DevScene& s, V3 origin, V3 v) {
    V3 vinv = v3(fast_rcp(v.x), fast_rcp(v.y), fast_rcp(v.z));
    float sum = 0.f;
    for (uint32_t i = 0; i < s.n_lights; ++i) sum += light_pdf_one<F>(load_const(s.light_prims + i), origin, v, vinv);
    return sum * fast_rcp((float)s.n_lights);
}

// A direction drawn around an axis: (cos(2 pi u) * rad, sin(2 pi u) * rad, z) in the basis built from
// `axis` (onb.rs:34-49). Both the cosine lobe (rand_gen.rs:80-90) and the cone towards a sphere
// (geo.rs:77-85) have this form, so they share the basis construction and the composition.
struct AxisDraw {
    V3 axis;
    float u, rad, z;
};
MIRT_DEV V3 axis_draw_direction(const AxisDraw& a) {
    float sn, cs;
    sincos_2pi(a.u, sn, cs);
    return onb_local(onb_from_w(a.axis), v3(cs * a.rad, sn * a.rad, a.z));
}
// rand_gen.rs:80-90 around `normal` — r2 in [0, 0.9): z >= sqrt(0.1).
MIRT_DEV AxisDraw cosine_draw(V3 normal, float xi1, float xi2) {
    float r2 = 0.9f * xi2;  // rand_gen.rs:20-23 with min 0: 0 + (0.9 - 0) * xi
    return AxisDraw{normal, xi1, fast_sqrt(r2), fast_sqrt(1.f - r2)};
}

// HittableList::random's member (ray.rs:152-154) for a light primitive, two draws: geo.rs:234-242
// (rect: a point on it) gives the direction itself and returns false; geo.rs:144-149 with geo.rs:77-85
// (sphere: a direction inside its cone) returns true and leaves the direction to axis_draw_direction().
template <uint32_t F>
MIRT_DEV bool light_draw(const DevPrim& p, V3 origin, float xi1, float xi2, V3& dir, AxisDraw& ad) {
    uint32_t kind = p.meta & PM_KIND_MASK;
    if ((F & F_RECT) && kind == PK_RECT) {
        uint32_t xi, yi, zi;
        axis_indexes((p.meta & PM_AXIS_MASK) >> PM_AXIS_SHIFT, xi, yi, zi);
        float a = fma_(p.a[2] - p.a[1], xi1, p.a[1]);
        float b = fma_(p.b[0] - p.a[3], xi2, p.a[3]);
        float k = p.b[1];
        V3 pt = v3(xi == 0 ? a : (yi == 0 ? b : k), xi == 1 ? a : (yi == 1 ? b : k), xi == 2 ? a : (yi == 2 ? b : k));
        dir = fnormalize(pt - origin);
        return false;
    }
    if ((F & F_SPHERE) && kind == PK_SPHERE) {
        V3 c0 = v3(p.a[0], p.a[1], p.a[2]);
        V3 direction = c0 - origin;
        float d = fnorm_squared(direction);
        float z = fma_(xi2, fast_sqrt(fma_(-(p.a[3] * p.a[3]), fast_rcp(d), 1.f)) - 1.f, 1.f);
        ad = AxisDraw{direction * fast_rsq(d), xi1, fast_sqrt(fma_(-z, z, 1.f)), z};
        return true;
    }
    dir = v3(1.f, 0.f, 0.f);  // ray.rs:41-43
    return false;
}

// ---------------------------------------------------------------------------------------------
// one level of ray_color (renderer.rs:197-237) after world.hit found a primitive
// ---------------------------------------------------------------------------------------------
// Value of the material's texture at the hit (texture.rs): the colour resolved at upload when the
// texture is a solid colour, the general evaluation otherwise.
template <uint32_t F>
MIRT_DEV V3 shade_color(const DevScene& s, const DevShade& m, const HitInfo& h, const DevPerlin* lds_perlin = nullptr) {
    if ((F & (F_TEX_CHECKER | F_TEX_NOISE | F_TEX_IMAGE)) && m.tex_kind != TK_SOLID) return texture_value<F>(s, m.texture, h.u, h.v, h.p, lds_perlin);
    return v3(m.color);
}

// material.rs:167-175: a DiffuseLight never scatters and emits toward front_face only.
template <uint32_t F>
MIRT_DEV V3 emitted_radiance(const DevScene& s, const DevShade& m, const HitInfo& h, const DevPerlin* lds_perlin = nullptr) {
    if (h.front) return shade_color<F>(s, m, h, lds_perlin);
    return v3(0.f, 0.f, 0.f);
}

// Material::scatter + the pdf handling of renderer.rs:205-227 for every material that can scatter.
// Returns false when the ray is absorbed. On true, `ray` is the scattered ray and `thr` has been
// multiplied by attenuation / divisor. `g` is the bounce's scatter stream.
template <uint32_t F>
MIRT_DEV bool scatter(const DevScene& s, const Tables& tb, const DevShade& m, const HitInfo& h, Rng& g, Ray& ray, V3& thr,
                      const DevPerlin* lds_perlin = nullptr) {
    uint32_t mk = m.mat_kind;
    if ((F & F_MAT_METAL) && mk == MK_METAL) {
        // material.rs:86-100
        V3 reflected = fmadd(ray.d, -2.f * fdot(h.n, ray.d), h.n);
        // (the bounce's first and only draws. The kernel forms without hierarchy and media run at 80 registers, where
        // the fixed-word sampler costs a spilled pair in the loop: Cornell 19 270 -> 18 960 Msamples/s; they keep the loop.)
        V3 sd = fmadd(reflected, m.param, (F & (F_BVH | F_MEDIUM)) ? rand_in_unit_sphere_last(g) : rand_in_unit_sphere(g));
        if (!(fdot(h.n, sd) > 0.f)) return false;
        ray.o = h.p;
        ray.d = fnormalize(sd);
        thr = cmul(thr, v3(m.color));
        return true;
    }
    if ((F & F_MAT_DIELECTRIC) && mk == MK_DIELECTRIC) {
        // material.rs:129-151
        float ratio = h.front ? fast_rcp(m.param) : m.param;
        float cos_theta = fminf(-fdot(ray.d, h.n), 1.f);
        float sin_theta = fast_sqrt(fma_(-cos_theta, cos_theta, 1.f));
        bool reflect = ratio * sin_theta > 1.f;
        if (!reflect) {
            float r0 = (1.f - ratio) * fast_rcp(1.f + ratio);
            r0 = r0 * r0;
            float x = 1.f - cos_theta;
            float x2 = x * x;
            float refl = fma_(1.f - r0, x2 * x2 * x, r0);  // material.rs:121-125
            reflect = refl > g.next();                      // drawn only when refraction is possible
        }
        V3 perp = ratio * fmadd(ray.d, cos_theta, h.n);     // material.rs:114-119
        V3 refr = fnormalize(fmadd(perp, -fast_sqrt(fabsf(1.f - fnorm_squared(perp))), h.n));  // perp + par
        V3 refl_d = fmadd(ray.d, -2.f * fdot(h.n, ray.d), h.n);  // material.rs:75-77
        ray.o = h.p;
        ray.d = select(reflect, refl_d, refr);
        return true;                                        // attenuation 1, divisor 1
    }
    if ((F & F_MAT_ISOTROPIC) && mk == MK_ISOTROPIC) {
        // material.rs:194-199
        V3 dir = fnormalize(rand_in_unit_sphere_last(g));  // the bounce's first and only draws
        V3 att = shade_color<F>(s, m, h, lds_perlin);
        ray.o = h.p;
        ray.d = dir;
        thr = cmul(thr, att);
        return true;
    }
    // Lambertian — material.rs:51-61 with the mixture of renderer.rs:207-221
    V3 att = shade_color<F>(s, m, h, lds_perlin);
    V3 dir;
    float pdf_val;
    if (s.n_lights == 0) {
        float a = g.next(), b = g.next();
        dir = axis_draw_direction(cosine_draw(h.n, a, b));          // pdf.rs:30-32
        float c = fdot(dir, h.n);
        pdf_val = c <= 0.f ? 0.0001f : c * (1.f / kPi);             // pdf.rs:25-28
    } else {
        float sel = g.next();
        float a = g.next(), b = g.next();
        AxisDraw ad = cosine_draw(h.n, a, b);                       // pdf.rs:77-83: the cosine half ...
        bool around_axis = true;
        if (!(sel < 0.5f)) {                                        // ... or a member of the light list
            float c = g.next();
            // ray.rs:152-154: member floor(n * xi)
            uint32_t li = (uint32_t)(0.f + ((float)s.n_lights - 0.f) * a);
            li = min(li, s.n_lights - 1);
            around_axis = light_draw<F>(tb.light_prims[li], h.p, b, c, dir, ad);
        }
        if (around_axis) dir = axis_draw_direction(ad);
        float c = fdot(dir, h.n);
        float p0 = c <= 0.f ? 0.0001f : c * (1.f / kPi);
        float p1 = lights_pdf<F>(s, h.p, dir);
        pdf_val = fma_(0.5f, p0, 0.5f * p1);                        // pdf.rs:73-75
    }
    float cosine = fmaxf(fdot(dir, h.n), 0.0001f);                  // renderer.rs:219
    // renderer.rs:221,227: attenuation / (pdf_val / cosine * 2 pi)
    float w = cosine * fast_rcp(pdf_val * (2.f * kPi));
    ray.o = h.p;
    ray.d = dir;
    thr = cmul(thr, att) * w;
    return true;
}

// cal_norm_coords (renderer.rs:174-184) + Camera::get_ray (camera.rs:86-94). When the lens radius
// is 0 and nothing in the scene moves, the lens-disk and shutter draws cannot influence the ray
// and are skipped (PINHOLE).
template <bool PINHOLE>
MIRT_DEV Ray camera_ray(const DevCamera& cam, uint32_t x, uint32_t y, uint32_t width, uint32_t height, bool jitter, Rng& g) {
    float xo = 0.f, yo = 0.f;
    if (jitter) {
        xo = g.next();
        yo = g.next();
    }
    const float wm1 = (float)(width - 1), hm1 = (float)(height - 1);
    float s = div_nr((float)x + xo, wm1, fast_rcp(wm1));
    float t = div_nr((float)y + yo, hm1, fast_rcp(hm1));
    Ray r;
    V3 target = fmadd(fmadd(v3(cam.llc), s, v3(cam.horizontal)), t, v3(cam.vertical)) - v3(cam.origin);
    if (PINHOLE) {
        r.o = v3(cam.origin);
        r.d = fnormalize(target);
        r.time = cam.time0;
        return r;
    }
    V3 disk;
    float shutter;
    if (jitter) {
        // The stream stands at draw 2 of block 0. Block 1 serves every lane (the second and third trial of the disk,
        // or the shutter draw of a lane whose first trial held), so it is generated once, for all of them, and the
        // word of each draw is fixed when the code is compiled: trial 1 = words 2, 3 of block 0, trial 2 = words 0, 1
        // and trial 3 = words 2, 3 of block 1, the shutter draw = the word after the trial that held. A lane that is
        // past its second trial (one in 23) goes on in the loop that keeps a position.
        uint32_t b0, b1, b2, b3;
        philox4x32_10(g.k0, g.k1, 1u, g.c1, g.c2, g.c3, b0, b1, b2, b3);
        const float x1 = unit_range(g.w2), y1 = unit_range(g.w3);
        const float x2 = unit_range(b0), y2 = unit_range(b1);
        const bool in1 = !(fma_(y1, y1, x1 * x1) >= 1.f);
        const bool in2 = !(fma_(y2, y2, x2 * x2) >= 1.f);
        disk = v3(in1 ? x1 : x2, in1 ? y1 : y2, 0.f);
        shutter = unit_f32(in1 ? b0 : b2);
        if (!(in1 || in2)) {
            g.pos = 6;
            g.have = 1;
            g.w0 = b0; g.w1 = b1; g.w2 = b2; g.w3 = b3;
            disk = rand_in_unit_disk(g);
            shutter = g.next();
        }
    } else {
        disk = rand_in_unit_disk(g);
        shutter = g.next();
    }
    V3 rd = cam.lens_radius * disk;
    V3 offset = fmadd(v3(cam.u) * rd.x, rd.y, v3(cam.v));
    r.o = v3(cam.origin) + offset;
    r.d = fnormalize(target - offset);
    r.time = fma_(cam.time1 - cam.time0, shutter, cam.time0);  // rand_gen.rs:20-23
    return r;
}

// renderer.rs:107 — NaN -> 0, +-inf -> 1.
MIRT_DEV float sanitize(float c) { return isnan(c) ? 0.f : (isinf(c) ? 1.f : c); }
// The three channels of a sample at once. Almost every sample is finite: one sum of the magnitudes tells (it is finite
// only if all three are; a sum that overflows merely sends finite channels through the exact path), and a wave none of
// whose lanes holds anything else skips the per-channel selects.
MIRT_DEV V3 sanitize3(V3 e) {
    const float m = (fabsf(e.x) + fabsf(e.y)) + fabsf(e.z);
    if (__ballot(!(m < MIRT_INF)) == 0) return e;
    return v3(sanitize(e.x), sanitize(e.y), sanitize(e.z));
}

}  // namespace mirt
