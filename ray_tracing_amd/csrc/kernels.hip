// gfx950 kernels of the path-tracing backend.
//
//   render_queue_kernel  the per-pixel sample loop of Renderer::draw (renderer.rs:97-110) as a persistent
//                        kernel: lanes draw (pixel, chunk of samples) items from one counter; each loop
//                        iteration is one ray segment of ray_color (renderer.rs:197-237) — trace, shade,
//                        scatter — with ray, throughput and RNG position in registers. A lane that
//                        finishes a path starts its next sample in the same iteration, one that finishes
//                        its item takes the next, so the 64 lanes of a wave stay busy although path
//                        lengths differ from 1 to max_depth and pixels differ in cost.
//   reduce_chunks_kernel adds the chunk sums of every pixel, in chunk order, to the accumulation buffer.
//   render_fused_kernel  the same loop with one lane = one pixel for the whole launch.
//   first_hit_kernel     world.hit(ray, 0.001, inf) for explicit rays (deterministic check).
//   ray_color_kernel     one un-sanitised radiance sample per explicit ray.
//   resolve_kernel       mean, clamp, sqrt, *256 -> RGBA8 with the row flip (renderer.rs:97,109-116).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include "kernels.h"
#include "dev_core.h"
#include "dev_pool.h"

namespace mirt {

constexpr int kBlock = 256;
// Waves per SIMD the register allocator is asked to fit (launch bound). Small list-only scenes are
// VALU-issue bound and gain from 6 resident waves (79 VGPRs, no scratch); the hierarchy and texture
// variants need their registers.
#ifndef MIRT_WAVES
#define MIRT_WAVES 6
#endif
#ifndef MIRT_WAVES_BVH
#define MIRT_WAVES_BVH 4
#endif
#ifndef MIRT_WAVES_TEX
#define MIRT_WAVES_TEX 4
#endif
constexpr int waves_for(uint32_t features) {
    return (features & (F_TEX_NOISE | F_TEX_IMAGE)) ? MIRT_WAVES_TEX : ((features & F_BVH) ? MIRT_WAVES_BVH : MIRT_WAVES);
}

// Statistics build (-DMIRT_WALK_STATS, read out by scripts/walk_hist.py through mirt_debug_walk_hist): per
// iteration of the persistent kernel the shader clocks spent in the list scans, the hierarchy walk and the
// rest (lane 0 of every wave), the lifetime of every wave against the longest one (how much of the launch
// is its tail), and histograms of the hierarchy steps per lane and of the longest lane of the wave. The
// histogram atomics slow the kernel several times: -DMIRT_PHASE_ONLY leaves them out for the clock figures.
#ifdef MIRT_WALK_STATS
#define MIRT_STATS(...) __VA_ARGS__
static __device__ unsigned long long g_walk_hist[16];  // [0..6] lanes by steps, [7..13] waves by longest lane, [14] / [15] sums
static __device__ unsigned long long g_phase[4];       // clocks: lists, walk, shading; [3] wave iterations
static __device__ unsigned long long g_life[4];        // sum of wave lifetimes, longest, number of waves
__device__ __forceinline__ void walk_histogram(bool alive, uint32_t steps) {
#ifndef MIRT_PHASE_ONLY
    auto bin = [](uint32_t v) { return v < 8 ? 0u : v < 16 ? 1u : v < 32 ? 2u : v < 64 ? 3u : v < 128 ? 4u : v < 256 ? 5u : 6u; };
    uint32_t mx = steps;
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off));
    if (alive) atomicAdd(&g_walk_hist[bin(steps)], 1ull);
    if (alive) atomicAdd(&g_walk_hist[14], (unsigned long long)steps);
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&g_walk_hist[7 + bin(mx)], 1ull);
        atomicAdd(&g_walk_hist[15], (unsigned long long)mx);
    }
#endif
}
__device__ __forceinline__ void wave_totals(long long life, long long lists, long long walk, long long shade, uint32_t rounds) {
    if ((threadIdx.x & 63u) != 0) return;
    atomicAdd(&g_life[0], (unsigned long long)life); atomicMax(&g_life[1], (unsigned long long)life); atomicAdd(&g_life[2], 1ull);
    atomicAdd(&g_phase[0], (unsigned long long)lists); atomicAdd(&g_phase[1], (unsigned long long)walk);
    atomicAdd(&g_phase[2], (unsigned long long)shade); atomicAdd(&g_phase[3], (unsigned long long)rounds);
}
#else
#define MIRT_STATS(...)
#endif

// capacity of the per-block LDS copies of the tables shading indexes per lane
constexpr uint32_t kLdsPrims = 64, kLdsInstances = 16, kLdsLights = 8;

// Pixel owned by a thread: a 256-thread block covers 32x8 pixels, each wave an 8x8 tile, so the
// lanes of a wave trace neighbouring paths.
__device__ __forceinline__ void thread_pixel(uint32_t width, uint32_t& x, uint32_t& y) {
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t tiles_x = (width + 31u) / 32u;
    uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    x = bx * 32u + wave * 8u + (lane & 7u);
    y = by * 8u + (lane >> 3);
}

template <class T>
__device__ __forceinline__ void copy_to_lds(T* dst, const T* src, uint32_t n) {
    const uint32_t words = n * (uint32_t)(sizeof(T) / 4);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (uint32_t i = threadIdx.x; i < words; i += kBlock) d[i] = s[i];
}

// Everything of one ray_color level after world.hit (renderer.rs:203-235) for the lane's current path,
// plus the start of its next sample when the path ends: classify the hit, close paths that end without
// drawing (miss, emitter), then ONE Philox block serves whichever branch the lane takes — the scatter
// stream of this bounce or the camera stream of its next sample — so the generator runs once at full
// lane occupancy instead of once per divergent branch. On return `alive` tells whether the lane has a
// ray to trace.
template <uint32_t F, bool PINHOLE>
__device__ __forceinline__ void shade_segment(const RenderArgs& a, const Tables& tb, uint32_t pixel, uint32_t x, uint32_t y, bool jitter,
                                              uint32_t sample_end, float t, int prim, Ray& ray, V3& thr, V3& acc, uint32_t& sample,
                                              uint32_t& depth_left, uint32_t& bounce, bool& alive) {
    HitInfo h;
    DevShade m;
    bool ended = false;
    V3 e = v3(0.f, 0.f, 0.f);
    if (prim < 0) {
        e = cmul(thr, v3(a.scene.background));  // renderer.rs:233-235
        ended = true;
    } else {
        m = tb.shade[prim];
        h = finalize_hit<F>(tb, ray, t, prim, m.tex_kind == TK_IMAGE);
        if ((F & F_MAT_LIGHT) && m.mat_kind == MK_DIFFUSE_LIGHT) {
            e = cmul(thr, emitted_radiance<F>(a.scene, m, h));
            ended = true;
        }
    }
    if (ended) {
        acc = acc + sanitize3(e);  // renderer.rs:107-109
        ++sample;
        alive = sample < sample_end;
    }
    const uint32_t r_bounce = ended ? 0u : bounce;
    const uint32_t r_stream = ended ? (uint32_t)STREAM_CAMERA : (uint32_t)STREAM_SCATTER;
    uint32_t w0, w1, w2, w3;
    philox4x32_10(pixel, sample, 0u, r_bounce, r_stream, a.seed, w0, w1, w2, w3);
    Rng g;
    g.init_with_block0(pixel, sample, r_bounce, r_stream, a.seed, w0, w1, w2, w3);

    bool restart = false;  // path closed after its scatter draws: needs a camera ray from a fresh stream
    if (!ended) {
        bool cont = scatter<F>(a.scene, tb, m, h, g, ray, thr);
        --depth_left;
        ++bounce;
        if (!cont || depth_left == 0) {  // absorbed, or renderer.rs:198-200: the next level returns 0
            ++sample;
            alive = sample < sample_end;
            restart = alive;
        }
    } else if (alive) {
        ray = camera_ray<PINHOLE>(a.cam, x, y, a.width, a.height, jitter, g);
        thr = v3(1.f, 1.f, 1.f);
        depth_left = a.max_depth;
        bounce = 0;
    }
    if (restart) {
        Rng gc;
        gc.init(pixel, sample, 0u, STREAM_CAMERA, a.seed);
        ray = camera_ray<PINHOLE>(a.cam, x, y, a.width, a.height, jitter, gc);
        thr = v3(1.f, 1.f, 1.f);
        depth_left = a.max_depth;
        bounce = 0;
    }
}

// Adds the lane's sample sums to the accumulation buffer and the wave's counters to the statistics:
// one atomic per wave for the segment count, one for 64 x `rounds` (the wave's loop iterations).
__device__ __forceinline__ void finish_pixel(const RenderArgs& a, bool valid, uint32_t pixel, V3 acc, uint32_t segments, uint32_t rounds) {
    if (valid) {
        float4 v = a.accum[pixel];
        v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += (float)a.spp_count;
        a.accum[pixel] = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        segments += __shfl_down(segments, off);
        rounds = max(rounds, (uint32_t)__shfl_down(rounds, off));
    }
    if ((threadIdx.x & 63u) == 0 && a.segment_counter) {
        atomicAdd(a.segment_counter, (unsigned long long)segments);
        atomicAdd(a.segment_counter + 1, 64ull * rounds);
    }
}

// F: feature mask. LDS_TABLES: primitives / instances / materials / textures are copied to LDS once
// per block and shading reads them there. PINHOLE: lens radius 0 and nothing moves.
template <uint32_t F, bool LDS_TABLES, bool PINHOLE>
__global__ __launch_bounds__(kBlock, waves_for(F)) void render_fused_kernel(RenderArgs a) {
    extern __shared__ uint32_t bvh_stack[];  // [scene.stack_limit + kBvhStackSlack][kBlock], sized at the launch
    __shared__ DevPrim s_prims[LDS_TABLES ? kLdsPrims : 1];
    __shared__ DevShade s_shade[LDS_TABLES ? kLdsPrims : 1];
    __shared__ DevInstance s_inst[LDS_TABLES ? kLdsInstances : 1];
    __shared__ DevPrim s_lights[LDS_TABLES ? kLdsLights : 1];
    uint32_t* stack = bvh_stack + threadIdx.x;

    Tables tb;
    if (LDS_TABLES) {
        copy_to_lds(s_prims, a.scene.prims, a.scene.n_prims);
        copy_to_lds(s_shade, a.scene.shade, a.scene.n_prims);
        copy_to_lds(s_inst, a.scene.instances, a.scene.n_instances);
        copy_to_lds(s_lights, a.scene.light_prims, a.scene.n_lights);
        __syncthreads();
        tb = Tables{s_prims, s_shade, s_inst, s_lights};
    } else {
        tb = Tables{a.scene.prims, a.scene.shade, a.scene.instances, a.scene.light_prims};
    }

    uint32_t x, y;
    thread_pixel(a.width, x, y);
    const bool valid = x < a.width && y < a.height;
    const uint32_t pixel = y * a.width + x;
    const bool jitter = a.jitter != 0;

    V3 acc = v3(0.f, 0.f, 0.f);
    uint32_t segments = 0;
    uint32_t sample = a.spp_begin;
    const uint32_t sample_end = a.spp_begin + a.spp_count;
    bool alive = valid && a.max_depth > 0 && sample < sample_end;
    uint32_t depth_left = a.max_depth, bounce = 0;
    Ray ray{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 1.f), 0.f};
    V3 thr = v3(1.f, 1.f, 1.f);
    if (alive) {
        Rng g;
        g.init(pixel, sample, 0u, STREAM_CAMERA, a.seed);
        ray = camera_ray<PINHOLE>(a.cam, x, y, a.width, a.height, jitter, g);
    }
    while (alive) {
        // ---- trace: scene.world.hit (renderer.rs:201)
        DrawKey key{pixel, sample, bounce, a.seed};
        float t;
        int prim;
        world_hit<F>(a.scene, ray, key, stack, kBlock, t, prim);
        ++segments;
        shade_segment<F, PINHOLE>(a, tb, pixel, x, y, jitter, sample_end, t, prim, ray, thr, acc, sample, depth_left, bounce, alive);
    }
    finish_pixel(a, valid, pixel, acc, segments, segments);
}

// The sample loop as a persistent kernel. The grid is only as large as the device holds at once. The work
// items are (pixel, chunk of its samples); global counters hand them out a wave-load at a time — tile run
// c stands for the 64 pixel slots of 8x8 tile (c mod tiles) in sample chunk (c / tiles) — and a lane
// that has finished its item stores the sums and takes the next slot in the same iteration. Nothing waits
// for the most expensive pixel of a tile or the most expensive tile of a launch: lanes idle only while the
// last items of the whole launch finish. An item is rendered by exactly one lane, sample after sample.
struct ItemQueue {
    // all wave-uniform, packed because the list kernels have no scalar registers to spare:
    uint32_t chunk_pos;  // sample chunk << 7 | slots of the wave's current tile already handed out (64: none left)
    uint32_t origin;     // that tile's origin, x | y << 16
    uint32_t dead;       // bit i: counter i has passed its last tile (all of them: the launch has no work left)
};
// The tile runs are dealt to RenderArgs::item_counters counters, each in its own 128-byte line: counter i hands out the i-th
// contiguous share of them, in order. A wave draws from the counter of its workgroup (blockIdx mod their number) and moves on to
// the others when that one is exhausted. One counter for the whole device costs nothing while items are long, but
// with almost free samples (one sphere: 0.5 M trips in 6 ms) every trip queues behind all the others'.
// (Contiguous shares rather than every eighth run: the runs a wave takes one after the other then lie a few tile rows
// apart instead of a sixth of the frame, the lanes it holds at any time see the same few surfaces, and fewer waves pay
// for a branch that one lane in fifty needs — `final` +3.3 %, `random` +1.2 %, none of the scenes slower. Sixteen or
// thirty-two shares help `final` more, +5.5 %, and cost the 10^6-sphere pass and `smoke` 1-3 %: the host asks for
// sixteen where iterations are heavy, eight elsewhere.)
constexpr uint32_t kItemCounterStride = 32;  // in 32-bit words

// Hands an item to every lane with `need`: pixel position and chunk; false once the launch has none left.
// One trip to a counter fetches one wave-load. Longer runs (fixed, or a share of what is left) were measured:
// they only pay on scenes with almost free samples, and cost 3-50 % elsewhere, because neighbouring items cost
// alike and a wave that holds a long run of expensive ones becomes the tail of the launch.
__device__ __forceinline__ bool take_items(bool need, ItemQueue& q, uint32_t* counters, uint32_t n_counters, uint32_t n_tiles,
                                           uint32_t tiles_x, uint32_t n_chunks, uint32_t& x, uint32_t& y, uint32_t& chunk) {
    const uint32_t lane = threadIdx.x & 63u;
    bool got = false;
    for (;;) {
        const unsigned long long m = __ballot(need && !got);
        if (m == 0) break;
        if ((q.chunk_pos & 127u) == 64u) {
            uint32_t run = 0xffffffffu;
            for (uint32_t a = 0; a < n_counters && run == 0xffffffffu; ++a) {
                const uint32_t qi = (blockIdx.x + a) & (n_counters - 1u);
                if (q.dead & (1u << qi)) continue;
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(counters + qi * kItemCounterStride, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
const uint32_t band = (n_tiles * n_chunks + n_counters - 1u) / n_counters;
                const uint32_t r = qi * band + c;
                if (c < band && r < n_tiles * n_chunks) run = r;
                else q.dead |= 1u << qi;
            }
            if (run == 0xffffffffu) break;  // every counter is exhausted
            const uint32_t ch = run / n_tiles;  // wave-uniform arithmetic, once per 64 items
            const uint32_t tile = run - ch * n_tiles, ty = tile / tiles_x;
            q.origin = ((tile - ty * tiles_x) * 8u) | (ty * 8u) << 16;
            q.chunk_pos = ch << 7;
        }
        const uint32_t pos = q.chunk_pos & 127u, avail = 64u - pos;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (need && !got && rank < avail) {
            const uint32_t in = pos + rank;
            x = (q.origin & 0xffffu) + (in & 7u);
            y = (q.origin >> 16) + (in >> 3);
            chunk = q.chunk_pos >> 7;
            got = true;
        }
        q.chunk_pos += min((uint32_t)__popcll(m), avail);
    }
    return got;
}

// TOP_NODES: the hierarchies' nodes nearest the roots are copied to LDS behind the stacks and fetched from there
// (walk_node). Launched wherever at least a root and its children fit (launch_persistent): every walk starts with
// them, and the second fetch path in the node step costs next to nothing since it is a branch of LDS reads (while the
// compiler merged it with the global fetches into flat loads it cost 2-4 %, and the form was kept to scenes most of
// whose nodes fit). The 10^6-sphere scene has room for seven nodes beside its stacks: +1.4 % at 4 spp, even at 1 spp.
// POOL: the hierarchy walks of the workgroup's 256 lanes advance together in LDS (dev_pool.h) instead of lane by lane.
// COUNT: the hierarchy walks tally their node steps and leaf tests (a.count_cost launches; a form of its own so that the
// timed form carries neither the tallies nor a test for them in its node step).
template <uint32_t F, bool LDS_TABLES, bool PINHOLE, bool TOP_NODES = false, bool POOL = false, bool COUNT = false>
__global__ __launch_bounds__(kBlock, waves_for(F)) void render_queue_kernel(RenderArgs a) {
    static_assert(!COUNT || (!LDS_TABLES && !TOP_NODES && !POOL && (F & F_BVH) != 0), "the counting form is a form of the plain hierarchy variants");
    static_assert(!POOL || (!LDS_TABLES && !TOP_NODES && (F & F_BVH) != 0), "the pooled walk is a form of the hierarchy variants");
    // the LDS-table variants are only launched for scenes without a hierarchy: no traversal code, no stack
    constexpr uint32_t FT = LDS_TABLES ? (F & ~(uint32_t)F_BVH) : F;
    extern __shared__ uint32_t bvh_stack[];  // [scene.stack_limit + kBvhStackSlack][kBlock], sized at the launch
    __shared__ DevPrim s_prims[LDS_TABLES ? kLdsPrims : 1];
    __shared__ DevShade s_shade[LDS_TABLES ? kLdsPrims : 1];
    __shared__ DevInstance s_inst[LDS_TABLES ? kLdsInstances : 1];
    __shared__ DevPrim s_lights[LDS_TABLES ? kLdsLights : 1];
    uint32_t* stack = bvh_stack + threadIdx.x;
    // The scene's first set of noise tables (perlin.rs:9-21: 256 gradients and three permutations, 4 864 bytes): an
    // evaluation of the marble texture is seven octaves of three dependent fetches each, taken by the few lanes of a
    // wave that hit the textured surface while the others wait — from LDS they come back in a fifth of the time of
    // the vector L1.
    constexpr bool kLdsPerlin = (F & F_TEX_NOISE) != 0 && !LDS_TABLES;
    __shared__ __align__(16) uint32_t s_perlin[kLdsPerlin ? sizeof(DevPerlin) / 4 : 4];
    if (kLdsPerlin && a.scene.n_perlins != 0u) {
        const uint32_t* const src = reinterpret_cast<const uint32_t*>(a.scene.perlins);
        for (uint32_t i = threadIdx.x; i < sizeof(DevPerlin) / 4; i += kBlock) s_perlin[i] = src[i];
        __syncthreads();
    }
    const DevPerlin* const lds_perlin = kLdsPerlin ? reinterpret_cast<const DevPerlin*>(s_perlin) : nullptr;

    Tables tb;
    if (LDS_TABLES) {
        copy_to_lds(s_prims, a.scene.prims, a.scene.n_prims);
        copy_to_lds(s_shade, a.scene.shade, a.scene.n_prims);
        copy_to_lds(s_inst, a.scene.instances, a.scene.n_instances);
        copy_to_lds(s_lights, a.scene.light_prims, a.scene.n_lights);
        __syncthreads();
        tb = Tables{s_prims, s_shade, s_inst, s_lights};
    } else {
        tb = Tables{a.scene.prims, a.scene.shade, a.scene.instances, a.scene.light_prims};
    }

    // The launch arguments where they lie in memory. The variant that carries every feature (`final`: media, noise and
    // image textures, two hierarchies) reads them from there at the point of use — the item bookkeeping when an item
    // ends, the camera when a path starts, the scene's tables once per phase of an iteration — instead of holding them
    // in scalar registers for the whole loop: it has far more live scalars than registers, and what does not fit is
    // spilled to vector-register lanes, v_readlane / v_writelane, which compete with the arithmetic for issue slots
    // (342 of them in its code; 60 with the reloads; `final` +5 %). The hierarchy forms of the smaller variants do the
    // same (some 90 lane reads per iteration otherwise: `random` +1.9 %, 10^6 spheres +1.6 % — once the reloaded
    // pointers were known to be global memory, see in_global_memory; with flat fetches behind them the reloads lost
    // 1-2 % there). The list forms fit and keep their arguments in registers.
    constexpr bool kReloadArgs = (F & (F_TEX_NOISE | F_TEX_IMAGE)) != 0 || (!LDS_TABLES && (F & F_BVH) != 0);
    const __attribute__((address_space(4))) RenderArgs* const ka = (const __attribute__((address_space(4))) RenderArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    auto scene_now = [&]() -> DevScene {
        if constexpr (kReloadArgs) return in_global_memory(reload(&ka->scene));
        else return a.scene;
    };
    auto tables_now = [&]() -> Tables {
        if constexpr (kReloadArgs && !LDS_TABLES) {
            const DevScene sc = in_global_memory(reload(&ka->scene));
            return Tables{sc.prims, sc.shade, sc.instances, sc.light_prims};
        } else {
            return tb;
        }
    };
    auto cam_now = [&]() -> DevCamera {
        if constexpr (kReloadArgs) return reload(&ka->cam);
        else return a.cam;
    };
#define MIRT_ITEM_ARG(field) (kReloadArgs ? reload(&ka->field) : a.field)

    const bool jitter = a.jitter != 0;
    const uint32_t max_depth = a.max_depth;
    ItemQueue q{64u, 0u, 0u};

    // the lane's work item (a pixel and a run of its samples) and its path
    // (position packed x | y << 16, chunk slot derived when the item is stored: the list variants run at 80 VGPRs)
    uint32_t pixel = 0, xy = 0, sample = a.spp_begin, sample_end = a.spp_begin;
    V3 acc = v3(0.f, 0.f, 0.f);
    bool alive = false;
    uint32_t bounce = 0;  // scatters of the lane's path so far = draws' bounce index (renderer.rs:198: depth - 1 each)
    Ray ray{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 1.f), 0.f};
    V3 thr = v3(1.f, 1.f, 1.f);
    uint32_t segments = 0, rounds = 0;  // of the whole wave (scalar registers)
    __shared__ uint32_t s_walk_cost[(FT & F_BVH) ? (kBlock / 64) * 4 : 1];  // per wave: node steps, leaf spheres, leaf rects, child boxes
    uint32_t* const wave_words = s_walk_cost + ((FT & F_BVH) ? (threadIdx.x >> 6) * 4u : 0u);
    if ((FT & F_BVH) && (threadIdx.x & 63u) == 0u) wave_words[0] = wave_words[1] = wave_words[2] = wave_words[3] = 0u;
    // the hierarchies' nodes nearest the roots, copied behind the stacks (walk_node)
    const uint32_t* top = nullptr;
    if constexpr ((FT & F_BVH) != 0 && TOP_NODES) {
        uint32_t* const dst = bvh_stack + (a.scene.stack_limit + kBvhStackSlack) * kBlock;
        const uint32_t* const src = reinterpret_cast<const uint32_t*>(a.scene.bvh_nodes);
        constexpr uint32_t kNodeWords = (uint32_t)sizeof(DevBvhNode4) / 4u, kStrideWords = kLdsNodeStride / 4u;
        for (uint32_t i = threadIdx.x; i < a.scene.lds_nodes * kNodeWords; i += kBlock) dst[(i / kNodeWords) * kStrideWords + (i % kNodeWords)] = src[i];
        __syncthreads();
        top = dst;
    }
    MIRT_STATS(const long long t_birth = clock64(); long long ph0 = 0, ph1 = 0, ph2 = 0);
    Walk wk;               // the lane's walk of the hierarchies, kept from iteration to iteration
    bool walking = false;  // ... which is not over yet
    bool closed = false;   // the lane's path ended in its last scatter; no ray to trace this iteration
    uint32_t walk_n = 0;
    const uint32_t walk_budget = a.walk_budget, walk_min_lanes = a.walk_min_lanes, walk_node_lanes = a.walk_node_lanes;
    wk.ref = kWalkIdle;
    wk.next = wk.sp = 0;
    // the workgroup's pool of walks, behind the stacks (POOL form)
    Pool pool{};
    PoolState pool_state{0u, 0u, 0u, 0u, 0u};
    if constexpr (POOL) {
        pool = pool_at(bvh_stack + (a.scene.stack_limit + kBvhStackSlack) * kBlock);
        pool_clear(pool);
        __syncthreads();
    }
    const uint32_t pool_stop = a.pool_stop, pool_steps = a.pool_steps;

    // Gives the lanes with `need` their next pixel: position, first sample, empty sums. Indices of the edge
    // tiles that lie outside the frame are skipped. A lane that gets none is done (alive = false).
    // (the closures below read the launch arguments they need when they run — an item ends once in a few hundred
    // iterations)
    auto next_pixel = [&q, &alive, &xy, &pixel, &sample, &sample_end, &acc, &a, ka](bool need) {
        if (__ballot(need) == 0) return;
        (void)ka;
        uint32_t* const work_counter = in_global_memory(MIRT_ITEM_ARG(work_counter));
        const uint32_t width = MIRT_ITEM_ARG(width), height = MIRT_ITEM_ARG(height), spp_begin = MIRT_ITEM_ARG(spp_begin);
        const uint32_t spp_last = spp_begin + MIRT_ITEM_ARG(spp_count), chunk_spp = MIRT_ITEM_ARG(chunk_spp), n_chunks = MIRT_ITEM_ARG(n_chunks);
        const uint32_t tiles_x = (width + 7u) / 8u, n_tiles = tiles_x * ((height + 7u) / 8u);
        for (;;) {
            if (__ballot(need) == 0) break;
            uint32_t chunk = 0, x = 0, y = 0;
            const bool got = take_items(need, q, work_counter, MIRT_ITEM_ARG(item_counters), n_tiles, tiles_x, n_chunks, x, y, chunk);
            if (need) {
                alive = got;
                if (alive) {
                    pixel = y * width + x;
                    xy = x | y << 16;
                    sample = spp_begin + chunk * chunk_spp;
                    sample_end = min(sample + chunk_spp, spp_last);
                    acc = v3(0.f, 0.f, 0.f);
                }
                need = alive && !(x < width && y < height);  // outside the frame: take another
            }
        }
    };
    // The lane's work item has all its samples (renderer.rs:107-110). With one chunk per pixel the sums go
    // straight to the accumulation buffer; otherwise to the item's own slot, and reduce_chunks_kernel adds
    // the slots of a pixel in chunk order — no floating-point atomics, so the result does not depend on
    // which lane rendered what, or when.
    const uint32_t stamp = a.stamp;
#ifdef MIRT_DEBUG_HOOKS
    const uint32_t drop_pixel = a.debug_drop_pixel;
#else
    constexpr uint32_t drop_pixel = 0xffffffffu;  // the hook exists only in the test build of the library
#endif
    auto flush_pixel = [&pixel, &sample_end, &acc, &a, ka, stamp, drop_pixel]() {
        const uint32_t spp_begin = MIRT_ITEM_ARG(spp_begin), chunk_spp = MIRT_ITEM_ARG(chunk_spp), n_chunks = MIRT_ITEM_ARG(n_chunks);
        const uint32_t n_pixels = MIRT_ITEM_ARG(width) * MIRT_ITEM_ARG(height);
        const float spp_count = (float)MIRT_ITEM_ARG(spp_count);
        float4* const accum = in_global_memory(MIRT_ITEM_ARG(accum));
        float4* const partial = in_global_memory(MIRT_ITEM_ARG(partial));
        (void)ka;
        const uint32_t chunk = (sample_end - 1u - spp_begin) / chunk_spp;
#ifdef MIRT_DEBUG_HOOKS
        if (pixel == drop_pixel && chunk == 0u) return;
#endif
        (void)drop_pixel;
        if (n_chunks == 1u) {
            float4 v = accum[pixel];
            v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += spp_count;
            accum[pixel] = v;
        } else {
            partial[chunk * n_pixels + pixel] = make_float4(acc.x, acc.y, acc.z, __uint_as_float(stamp));
        }
    };

    next_pixel(true);
    if (alive) {
        Rng g;
        g.init(pixel, sample, 0u, STREAM_CAMERA, a.seed);
        ray = camera_ray<PINHOLE>(cam_now(), xy & 0xffffu, xy >> 16, a.width, a.height, jitter, g);
    }
    while (POOL || __ballot(alive) != 0) {
        ++rounds;
#ifdef MIRT_PROBE_SALU  // measurement builds only: what 100 more scalar / vector instructions per iteration cost
        { uint32_t dummy; asm volatile("s_mov_b32 %0, 0\n .rept 100\n s_add_u32 %0, %0, 1\n .endr" : "=s"(dummy) : : "scc"); asm volatile("" : : "s"(dummy)); }
#endif
#ifdef MIRT_PROBE_VALU_FAST
        { uint32_t d0 = threadIdx.x, d1 = rounds, d2 = 3, d3 = 7; asm volatile(".rept 25\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n .endr" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); if (d0 + d1 + d2 + d3 == 0xdeadbeefu) ++segments; }
#endif
#ifdef MIRT_PROBE_VALU_SLOW
        { uint32_t d0 = threadIdx.x, d1 = rounds, d2 = 3, d3 = 7; asm volatile(".rept 25\n v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %0\n .endr" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); if (d0 + d1 + d2 + d3 == 0xdeadbeefu) ++segments; }
#endif
        // ---- trace: scene.world.hit (renderer.rs:201). The lists are scanned in the iteration that starts
        //      the segment; the hierarchies are walked a bounded number of steps per iteration, and a lane
        //      whose walk is not over sits out the rest of the iteration and continues in the next one.
        float t = MIRT_INF;
        int prim = -1;
        bool ready = alive;
        MIRT_STATS(const long long tc0 = clock64(); long long tc1 = tc0, tc2 = tc0);
        if constexpr ((FT & F_BVH) != 0 && POOL) {
            // The lanes that start a segment scan the lists and hand their ray to the workgroup's pool; then all four
            // waves advance the pool's walks together, round by round (dev_pool.h). A lane whose ray is not through
            // when the rounds stop keeps waiting (walking) and sits out the rest of the iteration.
            const bool start = alive && !walking && !closed;
            bool submit = false;
            if (start) {
                DrawKey key{pixel, sample, bounce, a.seed};
                wk.closest = MIRT_INF;
                wk.best = -1;
                world_lists<FT>(scene_now(), ray, key, wk.closest, wk.best);
                submit = walk_count(scene_now(), ray) != 0u;
            }
            segments += (uint32_t)__popcll(__ballot(start));
            MIRT_STATS(tc1 = clock64());
            pool_submit<FT>(scene_now(), pool, pool_state, submit, ray, wk.closest, wk.best);
            walking = walking || submit;
            // (a workgroup that still iterates after this many iterations has lost a ray: leave instead of hanging)
            if (!pool_rounds<FT>(scene_now(), pool, pool_state, alive && rounds < (1u << 20), pool_stop, pool_steps, bvh_stack)) break;
            if (walking && (pool.meta[threadIdx.x] & kMetaDone)) {
                wk.closest = pool.a[threadIdx.x].w;
                wk.best = __float_as_int(pool.d[threadIdx.x].w);
                walking = false;
            }
            MIRT_STATS(tc2 = clock64());
            ready = alive && !walking;
            t = wk.closest;
            prim = wk.best;
        } else if constexpr ((FT & F_BVH) != 0) {
            if (alive && !walking && !closed) {
                DrawKey key{pixel, sample, bounce, a.seed};
                wk.closest = MIRT_INF;
                wk.best = -1;
                world_lists<FT>(scene_now(), ray, key, wk.closest, wk.best);
                wk.next = 0;
                wk.ref = kWalkIdle;
                walk_n = walk_count(scene_now(), ray);
            }
            segments += (uint32_t)__popcll(__ballot(alive && !walking && !closed));
            uint32_t wsteps = 0;
            MIRT_STATS(tc1 = clock64());
            // the lane thresholds are stated for a full wave: scale them to the lanes this wave still has
            const uint32_t live = (uint32_t)__popcll(__ballot(alive));
            if (alive && !closed)
                walking = !walk_some<FT, COUNT>(scene_now(), ray, 0.001f, wk, walk_n, walk_budget, (walk_min_lanes * live) >> 6,
                                                (walk_node_lanes * live) >> 6, stack, kBlock, wsteps, COUNT ? wave_words : nullptr, top);
            MIRT_STATS(tc2 = clock64());
            MIRT_STATS(walk_histogram(alive, wsteps));
            ready = alive && !walking;
            t = wk.closest;
            prim = wk.best;
        } else if (alive && !closed) {
            DrawKey key{pixel, sample, bounce, a.seed};
            world_hit<FT>(scene_now(), ray, key, stack, kBlock, t, prim);
        }
        if constexpr ((FT & F_BVH) == 0) segments += (uint32_t)__popcll(__ballot(alive && !closed));
        if constexpr ((FT & F_BVH) == 0) { MIRT_STATS(tc1 = tc2 = clock64()); }
        // ---- classify; paths that end here without drawing (miss, emitter) are closed first
        HitInfo h;
        DevShade m;
        bool ended = false;
        if (ready && closed) {
            ended = true;  // closed by its last scatter (sample already counted): straight to the next sample
            closed = false;
        } else if (ready) {
            V3 e = v3(0.f, 0.f, 0.f);
            if (prim < 0) {
                e = cmul(thr, v3(scene_now().background));  // renderer.rs:233-235
                ended = true;
            } else {
                const Tables tbn = tables_now();
                m = tbn.shade[prim];
                h = finalize_hit<F>(tbn, ray, t, prim, m.tex_kind == TK_IMAGE);
                if ((F & F_MAT_LIGHT) && m.mat_kind == MK_DIFFUSE_LIGHT) {
                    e = cmul(thr, emitted_radiance<F>(scene_now(), m, h, lds_perlin));
                    ended = true;
                }
            }
            if (ended) {
                acc = acc + sanitize3(e);  // renderer.rs:107-109
                ++sample;
            }
        }
        // ---- pixels that are complete go out, their lanes take the next pixel of the launch
        {
            const bool done = ready && ended && sample == sample_end;
            if (__ballot(done) != 0) {
                if (done) flush_pixel();
                next_pixel(done);
            }
        }
        // ---- one Philox block per lane and iteration serves whichever branch the lane takes next:
        //      the camera stream of its next sample, or the scatter stream of this bounce
        const uint32_t r_bounce = ended ? 0u : bounce;
        const uint32_t r_stream = ended ? (uint32_t)STREAM_CAMERA : (uint32_t)STREAM_SCATTER;
        uint32_t w0, w1, w2, w3;
        philox4x32_10(pixel, sample, 0u, r_bounce, r_stream, a.seed, w0, w1, w2, w3);
        Rng g;
        g.init_with_block0(pixel, sample, r_bounce, r_stream, a.seed, w0, w1, w2, w3);

        if (ready && alive && !ended) {
            const bool cont = scatter<F>(scene_now(), tables_now(), m, h, g, ray, thr, lds_perlin);
            ++bounce;
            // absorbed, or renderer.rs:198-200: the next level returns 0. The sample is complete; its lane sits
            // out the trace of the next iteration and gets there what every path that ends gets: the item
            // check and a camera ray from that iteration's Philox block. (Doing both here instead means a
            // second Philox block and camera ray issued for the one or two lanes of a wave in this state.)
            if (!cont || bounce == max_depth) {
                ++sample;
                closed = true;
            }
        } else if (ready && alive) {
            ray = camera_ray<PINHOLE>(cam_now(), xy & 0xffffu, xy >> 16, a.width, a.height, jitter, g);
            thr = v3(1.f, 1.f, 1.f);
            bounce = 0;
        }
        MIRT_STATS(const long long tc3 = clock64(); ph0 += tc1 - tc0; ph1 += tc2 - tc1; ph2 += tc3 - tc2);
    }
    MIRT_STATS(wave_totals(clock64() - t_birth, ph0, ph1, ph2, rounds));
    if constexpr (POOL) {
        if (threadIdx.x == 0 && rounds >= (1u << 20)) atomicOr(a.scene.fault, (uint32_t)FAULT_POOL_STALL);
    }
    if ((threadIdx.x & 63u) == 0 && a.segment_counter) {
        atomicAdd(a.segment_counter, (unsigned long long)segments);
        atomicAdd(a.segment_counter + 1, 64ull * rounds);
        if constexpr ((FT & F_BVH) != 0 && COUNT) {
            {
                atomicAdd(a.segment_counter + 2, (unsigned long long)wave_words[0]);
                atomicAdd(a.segment_counter + 3, (unsigned long long)wave_words[1]);
                atomicAdd(a.segment_counter + 4, (unsigned long long)wave_words[2]);
                atomicAdd(a.segment_counter + 5, (unsigned long long)wave_words[3]);
            }
        }
    }
}

hipError_t launch_reduce_chunks(const RenderArgs& a, hipStream_t stream);
// max_depth == 0: ray_color returns black at once (renderer.rs:198-200); the samples still count.
__global__ __launch_bounds__(kBlock) void count_samples_kernel(float4* accum, uint32_t n, float spp) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) accum[i].w += spp;
}

// Second step of a persistent launch that split every pixel's samples into chunks: the chunk sums of a
// pixel, in chunk order, onto the accumulation buffer.
// A slot counts only when it carries the launch's stamp (RenderArgs::stamp): the sample count of the pixel then
// grows by the chunk's samples; a slot without it was never stored by this launch and is reported.
__global__ __launch_bounds__(kBlock) void reduce_chunks_kernel(float4* accum, const float4* partial, uint32_t n, uint32_t n_chunks,
                                                               uint32_t chunk_spp, uint32_t spp_count, uint32_t stamp, uint32_t* fault) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = accum[i];
    uint32_t counted = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const float4 p = partial[(size_t)c * n + i];
        if (__float_as_uint(p.w) == stamp) {
            v.x += p.x; v.y += p.y; v.z += p.z;
            counted += min(chunk_spp, spp_count - c * chunk_spp);
        }
    }
    if (counted != spp_count) atomicOr(fault, (uint32_t)FAULT_STALE_CHUNK);
    v.w += (float)counted;
    accum[i] = v;
}

template <uint32_t F>
__global__ __launch_bounds__(kBlock) void first_hit_kernel(RayBatchArgs a) {
    extern __shared__ uint32_t bvh_stack[];  // [scene.stack_limit + kBvhStackSlack][kBlock], sized at the launch
    uint32_t* stack = bvh_stack + threadIdx.x;
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const Tables tb{a.scene.prims, a.scene.shade, a.scene.instances, a.scene.light_prims};
    const float* q = a.rays + (size_t)i * 7;
    Ray ray{v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
    DrawKey key{i, 0u, 0u, a.seed};
    float t;
    int prim;
    world_hit<F>(a.scene, ray, key, stack, kBlock, t, prim);
    a.prim_id[i] = prim;
    a.t[i] = t;
    if (a.aov) {
        float* o = a.aov + (size_t)i * 12;
        for (int k = 0; k < 12; ++k) o[k] = 0.f;
        if (prim >= 0) {
            HitInfo h = finalize_hit<F>(tb, ray, t, prim);
            const DevShade m = a.scene.shade[prim];
            V3 c = v3(1.f, 1.f, 1.f);
            if (m.mat_kind != MK_DIELECTRIC) c = shade_color<F>(a.scene, m, h);
            o[0] = h.p.x; o[1] = h.p.y; o[2] = h.p.z;
            o[3] = h.n.x; o[4] = h.n.y; o[5] = h.n.z;
            o[6] = h.front ? 1.f : 0.f; o[7] = h.u; o[8] = h.v;
            o[9] = c.x; o[10] = c.y; o[11] = c.z;
        }
    }
}

template <uint32_t F>
__global__ __launch_bounds__(kBlock) void ray_color_kernel(RayBatchArgs a) {
    extern __shared__ uint32_t bvh_stack[];  // [scene.stack_limit + kBvhStackSlack][kBlock], sized at the launch
    uint32_t* stack = bvh_stack + threadIdx.x;
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const Tables tb{a.scene.prims, a.scene.shade, a.scene.instances, a.scene.light_prims};
    const float* q = a.rays + (size_t)i * 7;
    Ray ray{v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
    V3 thr = v3(1.f, 1.f, 1.f);
    V3 e = v3(0.f, 0.f, 0.f);
    for (uint32_t bounce = 0; bounce < a.max_depth; ++bounce) {
        DrawKey key{i, a.sample, bounce, a.seed};
        float t;
        int prim;
        world_hit<F>(a.scene, ray, key, stack, kBlock, t, prim);
        if (prim < 0) {
            e = cmul(thr, v3(a.scene.background));
            break;
        }
        HitInfo h = finalize_hit<F>(tb, ray, t, prim);
        DevShade m = tb.shade[prim];
        if ((F & F_MAT_LIGHT) && m.mat_kind == MK_DIFFUSE_LIGHT) {
            e = cmul(thr, emitted_radiance<F>(a.scene, m, h));
            break;
        }
        Rng g;
        g.init(i, a.sample, bounce, STREAM_SCATTER, a.seed);
        if (!scatter<F>(a.scene, tb, m, h, g, ray, thr)) break;
    }
    a.color[(size_t)i * 3 + 0] = e.x;
    a.color[(size_t)i * 3 + 1] = e.y;
    a.color[(size_t)i * 3 + 2] = e.z;
}

// renderer.rs:135-137
__device__ __forceinline__ uint8_t float_to_rgb(float num) {
    float c = fminf(fmaxf(num, 0.f), 0.999f);
    return (uint8_t)(sqrtf(c) * 256.f);
}

__global__ __launch_bounds__(kBlock) void resolve_kernel(const float4* accum, uchar4* frame, uint32_t width, uint32_t height) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= width * height) return;
    uint32_t fy = i / width, x = i - fy * width;
    uint32_t y = height - 1u - fy;  // renderer.rs:97: renderer row 0 is the LAST frame row
    float4 s = accum[y * width + x];
    float n = s.w;
    uchar4 px;
    px.x = float_to_rgb(s.x / n);
    px.y = float_to_rgb(s.y / n);
    px.z = float_to_rgb(s.z / n);
    px.w = 0xff;
    frame[i] = px;
}

// ---------------------------------------------------------------------------------------------
// kernel specialisations: the smallest feature mask that covers the scene is launched
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kVariantBox = F_SPHERE | F_RECT | F_INSTANCE | F_MAT_DIELECTRIC | F_MAT_LIGHT | F_MAT_METAL;
constexpr uint32_t kVariantSmoke = kVariantBox | F_MEDIUM | F_MAT_ISOTROPIC;
// (spheres alone, as in `random` and its scaled form: the leaf and hit-record code without the per-lane choice between
// sphere and rect is 2-5 % faster there)
constexpr uint32_t kVariantSpheresOnly = F_SPHERE | F_MOVING | F_BVH | F_TEX_CHECKER | F_MAT_METAL | F_MAT_DIELECTRIC | F_MAT_LIGHT;
constexpr uint32_t kVariantSpheres = kVariantSpheresOnly | F_RECT;
constexpr uint32_t kVariantAll = F_ALL;

template <template <uint32_t> class Launcher, class Args>
static hipError_t dispatch(uint32_t features, const Args& a, hipStream_t stream) {
    if ((features & ~kVariantBox) == 0) return Launcher<kVariantBox>::run(a, stream);
    if ((features & ~kVariantSmoke) == 0) return Launcher<kVariantSmoke>::run(a, stream);
    if ((features & ~kVariantSpheresOnly) == 0) return Launcher<kVariantSpheresOnly>::run(a, stream);
    if ((features & ~kVariantSpheres) == 0) return Launcher<kVariantSpheres>::run(a, stream);
    return Launcher<kVariantAll>::run(a, stream);
}

// A persistent launch: as many workgroups as the device keeps resident at once (never more than the
// frame has work for); the waves share out the pixels through a.work_counter, which the caller zeroed.
// LDS bytes of the per-lane traversal stacks of one workgroup: [scene.stack_limit + kBvhStackSlack][kBlock] words,
// none for a scene without a hierarchy.
static uint32_t stack_bytes(const DevScene& s) {
    return (s.features & F_BVH) ? (s.stack_limit + kBvhStackSlack) * kBlock * (uint32_t)sizeof(uint32_t) : 0u;
}

// `kernel_top` (may be the same as `kernel`): the TOP_NODES form of the kernel, launched instead when at least half of
// the scene's hierarchy nodes fit into the LDS the resident workgroups leave free.
// `kernel_pool` (may be the same as `kernel`): the POOL form, launched when the arguments ask for it (a.pool) and the
// launch does not tally its walks; it takes kPoolBytes of LDS more than the stacks and leaves the nodes in memory.
// `kernel_count` (may be the same as `kernel`): the COUNT form, launched when the arguments ask for the walks' tallies.
template <class Kernel, class KernelTop, class KernelPool, class KernelCount>
static hipError_t launch_persistent(Kernel kernel, KernelTop kernel_top, KernelPool kernel_pool, KernelCount kernel_count, const RenderArgs& a_in,
                                    hipStream_t stream) {
    // (the kernels index the counters by blockIdx masked with their number minus one)
    if (a_in.item_counters == 0u || (a_in.item_counters & (a_in.item_counters - 1u)) != 0u || a_in.item_counters > kMaxItemCounters) return hipErrorInvalidValue;
    if ((void*)kernel_count != (void*)kernel && a_in.count_cost) return launch_persistent(kernel_count, kernel_count, kernel_count, kernel_count, a_in, stream);
    int device = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return e;
    RenderArgs a = a_in;
    if ((void*)kernel_pool != (void*)kernel && a.pool && !a.count_cost && (a.scene.features & F_BVH) && a.scene.stack_limit + kBvhStackSlack < 256u) {
        const size_t lds_bytes = stack_bytes(a.scene) + kPoolBytes;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_pool, kBlock, lds_bytes)) != hipSuccess) return e;
        if (per_cu > 0) {
            const uint64_t items = (uint64_t)((a.width + 7u) / 8u) * ((a.height + 7u) / 8u) * 64u * a.n_chunks;
            if (a.n_chunks == 0 || items >= 0xffffff00ull || (uint64_t)a.n_chunks * a.width * a.height >= 0xffffffffull ||
                (a.n_chunks > 1 && !a.partial))
                return hipErrorInvalidValue;
            a.scene.lds_nodes = 0;
            const uint32_t want = (uint32_t)((items + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(kernel_pool, dim3(std::min(want, (uint32_t)(cus * per_cu))), dim3(kBlock), lds_bytes, stream, a);
            if ((e = hipGetLastError()) != hipSuccess || a.n_chunks == 1) return e;
            return launch_reduce_chunks(a, stream);
        }
    }
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, stack_bytes(a.scene))) != hipSuccess) return e;
    // Nodes of the hierarchies kept in LDS behind the stacks: as many as fit without costing a resident workgroup
    // (a.scene.lds_nodes arrives as the number of nodes the scene has).
    uint32_t lds_nodes = 0;
    // (not for a launch of under 4 M samples — under a millisecond: there the copy into LDS is what shows)
    const bool long_launch = (uint64_t)a.width * a.height * a.spp_count >= (4ull << 20);
    if ((void*)kernel_top != (void*)kernel && (a.scene.features & F_BVH) && per_cu > 0 && long_launch) {
        int lds_cu = 0, with_nodes = 0;
        if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess) lds_cu = 0;
        const long room = std::min<long>((long)lds_cu / per_cu, 65536l) - (long)stack_bytes(a.scene) - 2048l;  // (the kernel's own few words, and slack)
        if (room > 0) lds_nodes = (uint32_t)std::min<long>(room / (long)kLdsNodeStride, (long)a.scene.lds_nodes);
        static const long lds_env = [] { const char* ev = std::getenv("MIRT_LDS_NODES"); return ev ? std::atol(ev) : -1l; }();
        if (lds_env >= 0) lds_nodes = std::min(lds_nodes, (uint32_t)lds_env);
        else if (lds_nodes < 5u && lds_nodes < a.scene.lds_nodes) lds_nodes = 0;  // (less than a root and its children: not worth the form)
        if (lds_nodes && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&with_nodes, kernel_top, kBlock, stack_bytes(a.scene) + lds_nodes * kLdsNodeStride) != hipSuccess ||
                          with_nodes < per_cu))
            lds_nodes = 0;
    }
    a.scene.lds_nodes = lds_nodes;
    const size_t lds_bytes = stack_bytes(a.scene) + (size_t)lds_nodes * kLdsNodeStride;
    const uint64_t items = (uint64_t)((a.width + 7u) / 8u) * ((a.height + 7u) / 8u) * 64u * a.n_chunks;
    if (a.n_chunks == 0 || items >= 0xffffff00ull || (uint64_t)a.n_chunks * a.width * a.height >= 0xffffffffull ||
        (a.n_chunks > 1 && !a.partial))
        return hipErrorInvalidValue;
    const uint32_t want = (uint32_t)((items + kBlock - 1) / kBlock);
    const uint32_t resident = (uint32_t)std::max(cus * per_cu, 1);
    if (lds_nodes) hipLaunchKernelGGL(kernel_top, dim3(std::min(want, resident)), dim3(kBlock), lds_bytes, stream, a);
    else hipLaunchKernelGGL(kernel, dim3(std::min(want, resident)), dim3(kBlock), lds_bytes, stream, a);
    if ((e = hipGetLastError()) != hipSuccess || a.n_chunks == 1) return e;
    return launch_reduce_chunks(a, stream);
}

template <uint32_t F>
struct RenderLauncher {
    static hipError_t run(const RenderArgs& a, hipStream_t stream) {
        const uint32_t tiles = ((a.width + 31u) / 32u) * ((a.height + 7u) / 8u);
        const DevScene& s = a.scene;
        const bool lds = s.n_prims <= kLdsPrims && s.n_instances <= kLdsInstances && s.n_lights <= kLdsLights;
        const bool pinhole = a.cam.lens_radius == 0.f && !(s.features & F_MOVING);
        dim3 grid(tiles), block(kBlock);
        // Which loop: the persistent kernel unless the caller asks for the other. Its (pixel, sample chunk)
        // items keep every wave busy until the launch ends, where the one-pixel-per-lane kernel waits for
        // the most expensive tile of the launch and, inside a wave, for the most expensive pixel of the
        // tile (measured at 1080p: Cornell 13.3 -> 15.6 Gsamples/s, `final` 0.80 -> 1.44).
        if (render_uses_queue(a)) {
            if (a.max_depth == 0) {
                const uint32_t n = a.width * a.height;
                hipLaunchKernelGGL(count_samples_kernel, dim3((n + kBlock - 1) / kBlock), block, 0, stream, a.accum, n, (float)a.spp_count);
                return hipGetLastError();
            }
            const bool lq = lds && !(s.features & F_BVH);  // the LDS-table variants carry no traversal stack
            if (lq && pinhole) return launch_persistent(render_queue_kernel<F, true, true>, render_queue_kernel<F, true, true>, render_queue_kernel<F, true, true>, render_queue_kernel<F, true, true>, a, stream);
            if (lq) return launch_persistent(render_queue_kernel<F, true, false>, render_queue_kernel<F, true, false>, render_queue_kernel<F, true, false>, render_queue_kernel<F, true, false>, a, stream);
            if constexpr ((F & F_BVH) != 0) {
                if (pinhole)
                    return launch_persistent(render_queue_kernel<F, false, true>, render_queue_kernel<F, false, true, true>,
                                             render_queue_kernel<F, false, true, false, true>, render_queue_kernel<F, false, true, false, false, true>, a, stream);
                return launch_persistent(render_queue_kernel<F, false, false>, render_queue_kernel<F, false, false, true>,
                                         render_queue_kernel<F, false, false, false, true>, render_queue_kernel<F, false, false, false, false, true>, a, stream);
            }
            if (pinhole) return launch_persistent(render_queue_kernel<F, false, true>, render_queue_kernel<F, false, true>, render_queue_kernel<F, false, true>, render_queue_kernel<F, false, true>, a, stream);
            return launch_persistent(render_queue_kernel<F, false, false>, render_queue_kernel<F, false, false>, render_queue_kernel<F, false, false>, render_queue_kernel<F, false, false>, a, stream);
        }
        if (lds && pinhole) hipLaunchKernelGGL((render_fused_kernel<F, true, true>), grid, block, stack_bytes(s), stream, a);
        else if (lds) hipLaunchKernelGGL((render_fused_kernel<F, true, false>), grid, block, stack_bytes(s), stream, a);
        else if (pinhole) hipLaunchKernelGGL((render_fused_kernel<F, false, true>), grid, block, stack_bytes(s), stream, a);
        else hipLaunchKernelGGL((render_fused_kernel<F, false, false>), grid, block, stack_bytes(s), stream, a);
        return hipGetLastError();
    }
};
template <uint32_t F>
struct FirstHitLauncher {
    static hipError_t run(const RayBatchArgs& a, hipStream_t stream) {
        hipLaunchKernelGGL(first_hit_kernel<F>, dim3((a.n + kBlock - 1) / kBlock), dim3(kBlock), stack_bytes(a.scene), stream, a);
        return hipGetLastError();
    }
};
template <uint32_t F>
struct RayColorLauncher {
    static hipError_t run(const RayBatchArgs& a, hipStream_t stream) {
        hipLaunchKernelGGL(ray_color_kernel<F>, dim3((a.n + kBlock - 1) / kBlock), dim3(kBlock), stack_bytes(a.scene), stream, a);
        return hipGetLastError();
    }
};

#ifdef MIRT_WALK_STATS
extern "C" void mirt_debug_walk_hist(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_hist), sizeof(g_walk_hist));
    (void)hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_phase), sizeof(g_phase));
    (void)hipMemcpyFromSymbol(out + 20, HIP_SYMBOL(g_life), sizeof(g_life));
    { unsigned long long z4[4] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_life), z4, sizeof(z4)); }
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_hist), z, sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(g_phase));
}
#endif
// The static scratch of mirt_render_static (kernels.h). In host code the array's name only stands for the symbol: the
// memory is reached through the per-device address below, never through the name.
static __device__ __align__(256) unsigned char scratch[kStaticPartialOffset + kStaticPartialBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_of_device[kMaxDevices] = {};
uint32_t g_scratch_stamp[kMaxDevices] = {};  // stamp of the last launch that filled the device's scratch
}  // namespace

hipError_t static_scratch_address(void** out) {
    *out = nullptr;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_of_device[device]) {
        void* p = nullptr;
        if ((e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))) != hipSuccess) return e;
        g_scratch_of_device[device] = p;
    }
    *out = g_scratch_of_device[device];
    return hipSuccess;
}

uint32_t static_scratch_next_stamp() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) return 1u;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (++g_scratch_stamp[device] == 0) g_scratch_stamp[device] = 1;
    return g_scratch_stamp[device];
}

bool render_uses_queue(const RenderArgs& a) {
    // the environment variable overrides an AUTO choice for A/B runs of whole programs
    static const int env = [] { const char* e = std::getenv("MIRT_QUEUE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (!a.work_counter || a.loop == LOOP_FUSED) return false;
    if (a.loop == LOOP_QUEUE) return true;
    if (env >= 0) return env == 1;
    // A lone primitive without lights (one textured sphere: 1.3 segments per sample, every pixel costs the same)
    // leaves the item queue nothing to even out, and the trips to its counters are then the larger cost
    // (measured at 1080p: 62 Gsamples/s with one pixel per lane, 46 through the queue).
    return !(a.scene.n_prims <= 1u && a.scene.n_lights == 0u);
}
hipError_t launch_render_fused(const RenderArgs& a, hipStream_t stream) {
    if (a.width == 0 || a.height == 0 || a.spp_count == 0) return hipSuccess;
    return dispatch<RenderLauncher>(a.scene.features, a, stream);
}
hipError_t launch_reduce_chunks(const RenderArgs& a, hipStream_t stream) {
    const uint32_t n = a.width * a.height;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, a.accum, a.partial, n, a.n_chunks,
                       a.chunk_spp, a.spp_count, a.stamp, a.scene.fault);
    return hipGetLastError();
}
hipError_t launch_first_hit(const RayBatchArgs& a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    return dispatch<FirstHitLauncher>(a.scene.features, a, stream);
}
hipError_t launch_ray_color(const RayBatchArgs& a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    return dispatch<RayColorLauncher>(a.scene.features, a, stream);
}
hipError_t launch_resolve_rgba8(const float4* accum, uint8_t* frame, uint32_t width, uint32_t height, hipStream_t stream) {
    uint32_t n = width * height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, accum,
                       reinterpret_cast<uchar4*>(frame), width, height);
    return hipGetLastError();
}

}  // namespace mirt
