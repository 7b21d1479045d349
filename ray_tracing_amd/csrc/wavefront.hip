// Wavefront form of the sample loop (the second kernel set; the persistent kernel of kernels.hip is the first).
//
// A pool of path slots lives in device memory and a render is a sequence of rounds, each one ray segment of every
// live path. The queue of a round holds one 40-byte record per live path — ray, closest hit so far, slot — in queue
// order, so the intersection stages stream through it:
//
//   wf_lists   scene.world.hit over the list entries of the world (renderer.rs:201, ray.rs:123-133): one lane per
//              record, primitives through scalar registers as in the persistent kernel
//   wf_walk    the hierarchy entries: a persistent kernel whose lanes FETCH records from the round's queue — a lane whose
//              walk is over stores its hit, and as soon as enough lanes of its wave are idle the wave hands them the
//              next records of the run it has reserved (one atomic per run), so the traversal code runs with most
//              lanes active however much the walks differ in length; the kernel holds nothing but a ray and a walk,
//              and its LDS stacks are only as deep as the scene's hierarchies, so more waves stay resident than of
//              the kernel that also carries the shading state
//   wf_shade   classify the hit, close paths that end (miss, emitter, absorbed, depth), add finished samples to the
//              slot's work item, store finished items, draw the next item, generate the camera ray of the next sample
//              or scatter (renderer.rs:203-235, renderer.rs:100-110); the records of the paths that go on are
//              appended to the next round's queue with a wave ballot + prefix count and one atomic per wave
//
// Work items are the persistent kernel's: (pixel, chunk of its samples), numbered chunk-major and tile by tile; a
// slot adds its item's samples in order and stores the sum in the item's own place, so the image does not depend on
// which slot rendered what (equal bit for bit to the persistent kernel's). Draws are addressed exactly as there
// (Philox key = pixel, sample; counter = block, bounce, stream, seed).
//
// Every counter the waves of a round share exists WF_LANES times (wavefront.h): a read-modify-write on one address
// costs about 35 ns on this device however many waves wait for it. The host never reads anything back inside a
// render: every kernel takes its sizes from the control words on the device, grids are sized for the pool, and the
// host only fetches the live counts after each batch of rounds.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "wavefront.h"
#include "dev_core.h"

namespace mirt {

constexpr int kWfBlock = 256;
constexpr uint32_t kNoSlot = 0xffffffffu;

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Reserves one queue position for every lane where `pred` holds (wave-collective, all lanes present): lanes are
// ranked with a ballot and a prefix population count and the wave reserves its range with a single atomic.
__device__ __forceinline__ uint32_t queue_reserve(bool pred, uint32_t* count) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return 0u;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __builtin_amdgcn_readfirstlane(base);
    return base + lane_rank(mask);
}

// Geometry of the work items of a launch: item i is pixel slot (i mod 64) of tile run (i / 64); run c stands for
// 8x8 tile (c mod n_tiles) in sample chunk (c / n_tiles). Slots of edge tiles outside the frame are no items.
struct ItemSpace {
    uint32_t n_items, n_tiles, tiles_x, width, height;
};
__device__ __forceinline__ ItemSpace item_space(const RenderArgs& r) {
    ItemSpace s;
    s.tiles_x = (r.width + 7u) / 8u;
    s.n_tiles = s.tiles_x * ((r.height + 7u) / 8u);
    s.n_items = s.n_tiles * 64u * r.n_chunks;
    s.width = r.width;
    s.height = r.height;
    return s;
}
// The tile runs are dealt to WF_LANES counters: run r belongs to counter r mod WF_LANES, which hands out the 64 slots
// of its runs in order. Slots counter k owns in all:
__device__ __forceinline__ uint32_t item_slots_of(const ItemSpace& sp, uint32_t k) { return (((sp.n_items >> 6) + WF_LANES - 1u - k) / WF_LANES) * 64u; }

// First counter at or after `home` (cyclically) that is not in `spent`; WF_LANES when all are.
__device__ __forceinline__ uint32_t next_counter(unsigned long long spent, uint32_t home) {
    const unsigned long long avail = ~spent;
    if (avail == 0ull) return WF_LANES;
    const unsigned long long rot = home ? ((avail >> home) | (avail << (64u - home))) : avail;
    return (home + (uint32_t)__ffsll((long long)rot) - 1u) & (WF_LANES - 1u);
}

// Hands the next work item to every lane with `need` (wave-collective, all lanes present): its pixel position and
// chunk. False for the lanes that find the launch's items exhausted. A wave draws from the counter of its workgroup,
// or from the next one that still has slots; which counters were spent when the round began comes from the control
// words (one scalar load of a line nobody writes during the round), so that at the end of a render the waves do not
// walk from counter to counter.
__device__ __forceinline__ bool take_item(bool need, uint32_t* ctl, const ItemSpace& sp, uint32_t& x, uint32_t& y, uint32_t& chunk) {
    bool got = false;
    if (__ballot(need) == 0) return false;
    const uint32_t home = blockIdx.x & (WF_LANES - 1u);
    unsigned long long spent = (unsigned long long)ctl[WF_CTL_ITEMS_SPENT] | (unsigned long long)ctl[WF_CTL_ITEMS_SPENT + 1] << 32;
    for (;;) {
        const unsigned long long m = __ballot(need && !got);
        if (m == 0) break;
        const uint32_t k = next_counter(spent, home);
        if (k == WF_LANES) break;
        const uint32_t own = item_slots_of(sp, k);
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0u) base = atomicAdd(ctl + k * WF_CTL_STRIDE + WF_CTL_ITEM, (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t j = base + lane_rank(m);
        if (need && !got && j < own) {
            const uint32_t run = (j >> 6) * WF_LANES + k, in = j & 63u;
            const uint32_t ch = run / sp.n_tiles, tile = run - ch * sp.n_tiles, ty = tile / sp.tiles_x;
            const uint32_t px = (tile - ty * sp.tiles_x) * 8u + (in & 7u), py = ty * 8u + (in >> 3);
            if (px < sp.width && py < sp.height) {
                x = px; y = py; chunk = ch;
                got = true;
            }
        }
        if (base + (uint32_t)__popcll(m) >= own) spent |= 1ull << k;
    }
    return got;
}

__device__ __forceinline__ uint32_t* live_count(const WfArgs& a, uint32_t parity, uint32_t ql) { return a.st.ctl + ql * WF_CTL_STRIDE + (parity ? WF_CTL_N1 : WF_CTL_N0); }

// The lane's path: what the shading stage reads and writes back.
struct PathRegs {
    uint32_t pixel, sample, sample_end, bounce, xy;
    Ray ray;
    V3 thr, acc;
};

// A slot starts a work item: sample range and empty sums.
__device__ __forceinline__ void begin_item(const RenderArgs& r, uint32_t x, uint32_t y, uint32_t chunk, PathRegs& p) {
    p.pixel = y * r.width + x;
    p.xy = x | y << 16;
    p.sample = r.spp_begin + chunk * r.chunk_spp;
    p.sample_end = min(p.sample + r.chunk_spp, r.spp_begin + r.spp_count);
    p.acc = v3(0.f, 0.f, 0.f);
}
// The item has all its samples (renderer.rs:107-110): with one chunk per pixel the sums go straight to the
// accumulation buffer, otherwise to the item's own slot of the chunk sums, stamped (see RenderArgs::stamp).
__device__ __forceinline__ void store_item(const RenderArgs& r, const PathRegs& p) {
    const uint32_t chunk = (p.sample_end - 1u - r.spp_begin) / r.chunk_spp;
#ifdef MIRT_DEBUG_HOOKS
    if (p.pixel == r.debug_drop_pixel && chunk == 0u) return;
#endif
    if (r.n_chunks == 1u) {
        float4 v = r.accum[p.pixel];
        v.x += p.acc.x; v.y += p.acc.y; v.z += p.acc.z; v.w += (float)r.spp_count;
        r.accum[p.pixel] = v;
    } else {
        r.partial[(size_t)chunk * r.width * r.height + p.pixel] = make_float4(p.acc.x, p.acc.y, p.acc.z, __uint_as_float(r.stamp));
    }
}
template <bool PINHOLE>
__device__ __forceinline__ void begin_sample(const RenderArgs& r, PathRegs& p, Rng& g) {
    p.ray = camera_ray<PINHOLE>(r.cam, p.xy & 0xffffu, p.xy >> 16, r.width, r.height, r.jitter != 0, g);
    p.thr = v3(1.f, 1.f, 1.f);
    p.bounce = 0;
}
// The slot's state, and the path's record in lane `ql` of the queue of the given parity.
__device__ __forceinline__ void store_path(const WfArgs& a, uint32_t parity, uint32_t ql, bool alive, uint32_t slot, const PathRegs& p) {
    const uint32_t pos = queue_reserve(alive, live_count(a, parity, ql));
    if (!alive) return;
    a.st.thr[slot] = make_float4(p.thr.x, p.thr.y, p.thr.z, 0.f);
    a.st.acc[slot] = make_float4(p.acc.x, p.acc.y, p.acc.z, __uint_as_float(p.xy));
    a.st.ids[slot] = make_uint4(p.pixel, p.sample, p.sample_end, p.bounce);
    const size_t e = (size_t)ql * a.lane_capacity + pos;
    a.st.q_o[parity][e] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, p.ray.time);
    a.st.q_d[parity][e] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, MIRT_INF);
    a.st.q_h[parity][e] = make_uint2(0xffffffffu, slot);
}

// Every slot of the pool draws its first item and the camera ray of its first sample; the records of the slots
// that got one make up the first queue.
template <bool PINHOLE>
__global__ __launch_bounds__(kWfBlock) void wf_init_kernel(WfArgs a) {
    const uint32_t slot = blockIdx.x * kWfBlock + threadIdx.x;
    const uint32_t ql = blockIdx.x & (WF_LANES - 1u);
    const ItemSpace sp = item_space(a.r);
    PathRegs p;
    uint32_t x = 0, y = 0, chunk = 0;
    const bool got = take_item(slot < a.capacity, a.st.ctl, sp, x, y, chunk);
    if (got) {
        begin_item(a.r, x, y, chunk, p);
        Rng g;
        g.init(p.pixel, p.sample, 0u, STREAM_CAMERA, a.r.seed);
        begin_sample<PINHOLE>(a.r, p, g);
    }
    store_path(a, 0u, ql, got, slot, p);
}

// Start of round r (one wave, lane k of it for queue lane k): the other queue is empty, the walk kernel fetches
// from the front, the round's rays are counted as segments, and the item counters that are spent are noted.
__global__ __launch_bounds__(64) void wf_prep_kernel(WfArgs a) {
    const uint32_t cur = a.round & 1u, k = threadIdx.x;
    uint32_t n = *live_count(a, cur, k);
    *live_count(a, cur ^ 1u, k) = 0u;
    a.st.ctl[k * WF_CTL_STRIDE + WF_CTL_CURSOR] = 0u;
    for (uint32_t c = 0; c < WC_COUNT; ++c) a.st.ctl[k * WF_CTL_STRIDE + WF_CTL_CLASS + c] = 0u;
    const ItemSpace sp = item_space(a.r);
    const unsigned long long spent = __ballot(a.st.ctl[k * WF_CTL_STRIDE + WF_CTL_ITEM] >= item_slots_of(sp, k));
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if (k == 0u) {
        a.st.ctl[WF_CTL_ITEMS_SPENT] = (uint32_t)spent;
        a.st.ctl[WF_CTL_ITEMS_SPENT + 1] = (uint32_t)(spent >> 32);
        if (a.r.segment_counter) atomicAdd(a.r.segment_counter, (unsigned long long)n);
    }
}

// The list entries of the world for every live path (HittableList::hit, ray.rs:123-133): closest t and primitive
// go into the path's record; the hierarchy walk continues from them.
template <uint32_t F>
__global__ __launch_bounds__(kWfBlock) void wf_lists_kernel(WfArgs a) {
    const uint32_t cur = a.round & 1u, ql = blockIdx.x & (WF_LANES - 1u), first = (blockIdx.x / WF_LANES) * kWfBlock;
    const uint32_t n = *live_count(a, cur, ql);
    if (first >= n) return;
    const bool on = first + threadIdx.x < n;
    const size_t e = (size_t)ql * a.lane_capacity + (on ? first + threadIdx.x : first);
    const float4 o = a.st.q_o[cur][e], d = a.st.q_d[cur][e];
    Ray ray{v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w};
    DrawKey key{0u, 0u, 0u, a.r.seed};
    if constexpr ((F & F_MEDIUM) != 0) {  // only a medium draws while it is being intersected
        const uint4 id = a.st.ids[a.st.q_h[cur][e].y];
        key = DrawKey{id.x, id.y, id.w, a.r.seed};
    }
    float closest = MIRT_INF;
    int best = -1;
    if (on) {
        world_lists<F>(a.r.scene, ray, key, closest, best);
        a.st.q_d[cur][e].w = closest;
        a.st.q_h[cur][e].x = (uint32_t)best;
    }
}

// The hierarchy entries of the world for every live path. Persistent: the grid is what the device holds at once.
// A wave reserves runs of `fetch_run` records of a queue lane (its workgroup's own first, then the next few) with one
// atomic each and hands records of its run to idle lanes as soon as `refill_lanes` of them are idle; all lanes walk
// on otherwise. Once nothing is left to fetch the remaining walks run to their end.
template <uint32_t F>
__global__ __launch_bounds__(kWfBlock, 5) void wf_walk_kernel(WfArgs a) {
    // per-lane traversal stacks, [level][lane], only as many levels as the scene's hierarchies need
    // (a.r.scene.stack_limit, sized at launch): the LDS a workgroup takes decides how many are resident
    extern __shared__ uint32_t bvh_stack[];
    __shared__ uint32_t s_walk_cost[(kWfBlock / 64) * 4];
    uint32_t* stack = bvh_stack + threadIdx.x;
    uint32_t* const wave_words = s_walk_cost + (threadIdx.x >> 6) * 4u;
    if ((threadIdx.x & 63u) == 0u) wave_words[0] = wave_words[1] = wave_words[2] = wave_words[3] = 0u;
    uint32_t* const wave_cost = a.r.count_cost ? wave_words : nullptr;  // only a counting launch tallies (walk_some)
    const uint32_t cur = a.round & 1u;
    const uint32_t refill = max(a.refill_lanes, 1u), run_len = max(a.fetch_run, 64u);
    constexpr uint32_t kLanesTried = 4;  // queue lanes a wave draws from: they hold equal shares, a few even out the rest
    uint32_t ql = blockIdx.x & (WF_LANES - 1u), tried = 0;
    uint32_t n = *live_count(a, cur, ql);
    uint32_t run_pos = 0, run_end = 0;  // the wave's reserved run [run_pos, run_end) of queue lane ql

    uint32_t entry = kNoSlot, n_h = 0;  // the record this lane is walking (index into the queue arrays)
    Ray ray{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 1.f), 0.f};
    Walk w;
    w.ref = kWalkIdle;
    w.sp = w.next = 0;
    w.closest = MIRT_INF;
    w.best = -1;
    bool exhausted = false;
    for (;;) {
        const unsigned long long idle_m = __ballot(entry == kNoSlot);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if (!exhausted && n_idle >= refill) {
            while (run_pos == run_end && !exhausted) {  // reserve the next run
                uint32_t base = 0xffffffffu;
                if ((threadIdx.x & 63u) == 0u) {
                    uint32_t* const cursor = a.st.ctl + ql * WF_CTL_STRIDE + WF_CTL_CURSOR;
                    if (__hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) base = atomicAdd(cursor, run_len);
                }
                base = __builtin_amdgcn_readfirstlane(base);
                if (base < n) {
                    run_pos = base;
                    run_end = min(base + run_len, n);
                } else if (++tried == kLanesTried) {
                    exhausted = true;
                } else {
                    ql = (ql + 1u) & (WF_LANES - 1u);
                    n = *live_count(a, cur, ql);
                }
            }
            if (run_pos < run_end) {
                const uint32_t take = min(n_idle, run_end - run_pos);
                const uint32_t rank = lane_rank(idle_m);
                if (entry == kNoSlot && rank < take) {
                    entry = ql * a.lane_capacity + run_pos + rank;
                    const float4 o = a.st.q_o[cur][entry], d = a.st.q_d[cur][entry];
                    ray = Ray{v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w};
                    w.closest = d.w;
                    w.best = (int)a.st.q_h[cur][entry].x;
                    w.ref = kWalkIdle;
                    w.sp = w.next = 0;
                    n_h = walk_count(a.r.scene, ray);
                }
                run_pos += take;
                // (lanes a short run leaves idle get theirs from the next run in the next iteration)
            }
        }
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(entry != kNoSlot));
        if (n_busy == 0u) {
            if (exhausted) break;
            continue;
        }
        if (entry != kNoSlot) {
            // this turn ends when a refill is due: fewer than 64 - refill + 1 lanes still walking means >= refill idle
            // (after a short fetch enough lanes are idle already: one step, then the next fetch)
            uint32_t steps;
            const uint32_t min_lanes = exhausted ? 0u : (n_busy > 64u - refill ? 64u - refill + 1u : n_busy + 1u);
            const bool done = a.r.count_cost ? walk_some<F, true>(a.r.scene, ray, 0.001f, w, n_h, a.r.walk_budget, min_lanes, a.r.walk_node_lanes, stack, kWfBlock, steps, wave_cost)
                                             : walk_some<F, false>(a.r.scene, ray, 0.001f, w, n_h, a.r.walk_budget, min_lanes, a.r.walk_node_lanes, stack, kWfBlock, steps, nullptr);
            if (done) {
                a.st.q_d[cur][entry].w = w.closest;
                a.st.q_h[cur][entry].x = (uint32_t)w.best;
                entry = kNoSlot;
            }
        }
    }
    // what the walks cost goes to the workgroup's own words (added to the statistics when the render ends): the
    // thousands of waves of every round adding to three shared words was a quarter of a short round
    __syncthreads();
    if (threadIdx.x < 4u && a.st.cost && a.r.count_cost) {
        unsigned long long s = 0;
        for (uint32_t wv = 0; wv < kWfBlock / 64; ++wv) s += s_walk_cost[wv * 4u + threadIdx.x];
        a.st.cost[(size_t)blockIdx.x * 4u + threadIdx.x] += s;
    }
}

__global__ __launch_bounds__(256) void wf_finish_kernel(WfArgs a) {
    __shared__ unsigned long long part[256][4];
    unsigned long long s[4] = {0, 0, 0, 0};
    for (uint32_t b = threadIdx.x; b < WF_MAX_WALK_GROUPS; b += 256u)
        for (int j = 0; j < 4; ++j) {
            s[j] += a.st.cost[(size_t)b * 4u + j];
            a.st.cost[(size_t)b * 4u + j] = 0ull;
        }
    for (int j = 0; j < 4; ++j) part[threadIdx.x][j] = s[j];
    __syncthreads();
    if (threadIdx.x < 4u && a.r.segment_counter) {
        unsigned long long t = 0;
        for (uint32_t i = 0; i < 256u; ++i) t += part[i][threadIdx.x];
        atomicAdd(a.r.segment_counter + 2 + threadIdx.x, t);
    }
}

// Files every record of the round's queue under the class of what its path does next (WC_*): one ballot and prefix
// count per class and one atomic per wave and class that has members (queue_reserve); the entry is the record's
// index within its queue lane. A workgroup files the records of its lane under the same lane of the class queues,
// so the four class lanes together hold exactly what the record lane held.
__global__ __launch_bounds__(kWfBlock) void wf_classify_kernel(WfArgs a) {
    const uint32_t cur = a.round & 1u, ql = blockIdx.x & (WF_LANES - 1u), first = (blockIdx.x / WF_LANES) * kWfBlock;
    const uint32_t n = *live_count(a, cur, ql);
    if (first >= n) return;
    const bool on = first + threadIdx.x < n;
    const uint32_t index = on ? first + threadIdx.x : first;
    const int prim = (int)a.st.q_h[cur][(size_t)ql * a.lane_capacity + index].x;
    uint32_t cls = WC_ENDED;
    if (prim >= 0) {
        const uint32_t mk = a.r.scene.shade[prim].mat_kind;
        cls = mk == MK_DIFFUSE_LIGHT ? WC_ENDED : (mk == MK_LAMBERTIAN ? WC_DIFFUSE : (mk == MK_ISOTROPIC ? WC_ISO : WC_SPECULAR));
    }
    for (uint32_t c = 0; c < WC_COUNT; ++c) {
        const bool mine = on && cls == c;
        const uint32_t pos = queue_reserve(mine, a.st.ctl + ql * WF_CTL_STRIDE + WF_CTL_CLASS + c);
        if (mine) a.st.cls[((size_t)c * WF_LANES + ql) * a.lane_capacity + pos] = index;
    }
}

// One level of ray_color after world.hit (renderer.rs:203-235) for every live path, then whatever follows for its
// slot: the next sample of its item (renderer.rs:100-110), the next item, or nothing more. CLASS = WC_ALL: the
// records of the round's queue in queue order, whatever they do next. CLASS = one of WC_*: the records filed under
// that class, and only the code that class runs.
template <uint32_t F, bool PINHOLE, uint32_t CLASS>
__global__ __launch_bounds__(kWfBlock) void wf_shade_kernel(WfArgs a) {
    const uint32_t cur = a.round & 1u, ql = blockIdx.x & (WF_LANES - 1u), first = (blockIdx.x / WF_LANES) * kWfBlock;
    const uint32_t n = CLASS == WC_ALL ? *live_count(a, cur, ql) : a.st.ctl[ql * WF_CTL_STRIDE + WF_CTL_CLASS + (CLASS & 3u)];
    if (first >= n) return;
    const bool on = first + threadIdx.x < n;
    const Tables tb{a.r.scene.prims, a.r.scene.shade, a.r.scene.instances, a.r.scene.light_prims};
    const ItemSpace sp = item_space(a.r);
    uint32_t index = on ? first + threadIdx.x : first;
    if (CLASS != WC_ALL) index = a.st.cls[((size_t)(CLASS & 3u) * WF_LANES + ql) * a.lane_capacity + index];
    const size_t e = (size_t)ql * a.lane_capacity + index;
    // the scatter code a class can reach: the other materials' branches are compiled out of its kernel
    constexpr uint32_t FS = CLASS == WC_DIFFUSE ? (F & ~(uint32_t)(F_MAT_METAL | F_MAT_DIELECTRIC | F_MAT_ISOTROPIC))
                          : CLASS == WC_SPECULAR ? (F & ~(uint32_t)F_MAT_ISOTROPIC)
                          : CLASS == WC_ISO ? (F & ~(uint32_t)(F_MAT_METAL | F_MAT_DIELECTRIC)) : F;
    PathRegs p;
    float t;
    int prim;
    uint32_t slot;
    {
        const float4 o = a.st.q_o[cur][e], d = a.st.q_d[cur][e];
        const uint2 hs = a.st.q_h[cur][e];
        slot = hs.y;
        const float4 th = a.st.thr[slot], ac = a.st.acc[slot];
        const uint4 id = a.st.ids[slot];
        p.ray = Ray{v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w};
        p.thr = v3(th.x, th.y, th.z);
        p.acc = v3(ac.x, ac.y, ac.z);
        p.xy = __float_as_uint(ac.w);
        p.pixel = id.x; p.sample = id.y; p.sample_end = id.z; p.bounce = id.w;
        t = d.w;
        prim = (int)hs.x;
    }
    // ---- classify; paths that end here without drawing (miss, emitter) are closed first
    HitInfo h;
    DevShade m;
    bool ended = false;
    if (on) {
        V3 e_ = v3(0.f, 0.f, 0.f);
        if (prim < 0) {
            e_ = cmul(p.thr, v3(a.r.scene.background));  // renderer.rs:233-235
            ended = true;
        } else {
            m = tb.shade[prim];
            h = finalize_hit<F>(tb, p.ray, t, prim, m.tex_kind == TK_IMAGE);
            if ((F & F_MAT_LIGHT) && m.mat_kind == MK_DIFFUSE_LIGHT) {
                e_ = cmul(p.thr, emitted_radiance<F>(a.r.scene, m, h));
                ended = true;
            }
        }
        if (ended) {
            p.acc = p.acc + v3(sanitize(e_.x), sanitize(e_.y), sanitize(e_.z));  // renderer.rs:107-109
            ++p.sample;
        }
    }
    // ---- items that are complete go out, their slots draw the next item of the launch
    bool alive = on;
    {
        const bool done = on && ended && p.sample == p.sample_end;
        if (__ballot(done) != 0) {
            if (done) store_item(a.r, p);
            uint32_t x = 0, y = 0, chunk = 0;
            const bool got = take_item(done, a.st.ctl, sp, x, y, chunk);
            if (done) {
                alive = got;
                if (got) begin_item(a.r, x, y, chunk, p);
            }
        }
    }
    // ---- one Philox block serves whichever branch the lane takes: the camera stream of its next sample, or the
    //      scatter stream of this bounce
    const uint32_t r_bounce = ended ? 0u : p.bounce;
    const uint32_t r_stream = ended ? (uint32_t)STREAM_CAMERA : (uint32_t)STREAM_SCATTER;
    uint32_t w0, w1, w2, w3;
    philox4x32_10(p.pixel, p.sample, 0u, r_bounce, r_stream, a.r.seed, w0, w1, w2, w3);
    Rng g;
    g.init_with_block0(p.pixel, p.sample, r_bounce, r_stream, a.r.seed, w0, w1, w2, w3);
    bool restart = false;  // closed by its scatter: the sample is complete and the slot needs a camera ray from a fresh stream
    if (CLASS != WC_ENDED && alive && !ended) {
        const bool cont = scatter<FS>(a.r.scene, tb, m, h, g, p.ray, p.thr);
        ++p.bounce;
        if (!cont || p.bounce == a.r.max_depth) {  // absorbed, or renderer.rs:198-200: the next level returns 0
            ++p.sample;
            restart = true;
        }
    } else if ((CLASS == WC_ALL || CLASS == WC_ENDED) && alive) {
        begin_sample<PINHOLE>(a.r, p, g);
    }
    if (CLASS != WC_ENDED && __ballot(restart) != 0) {
        const bool done = restart && p.sample == p.sample_end;
        if (__ballot(done) != 0) {
            if (done) store_item(a.r, p);
            uint32_t x = 0, y = 0, chunk = 0;
            const bool got = take_item(done, a.st.ctl, sp, x, y, chunk);
            if (done) {
                alive = got;
                if (got) begin_item(a.r, x, y, chunk, p);
            }
        }
        if (restart && alive) {
            Rng gc;
            gc.init(p.pixel, p.sample, 0u, STREAM_CAMERA, a.r.seed);
            begin_sample<PINHOLE>(a.r, p, gc);
        }
    }
    store_path(a, cur ^ 1u, ql, alive, slot, p);
}

// ---------------------------------------------------------------------------------------------
constexpr uint32_t kVariantBox = F_SPHERE | F_RECT | F_INSTANCE | F_MAT_DIELECTRIC | F_MAT_LIGHT | F_MAT_METAL;
constexpr uint32_t kVariantSmoke = kVariantBox | F_MEDIUM | F_MAT_ISOTROPIC;
constexpr uint32_t kVariantSpheresOnly = F_SPHERE | F_MOVING | F_BVH | F_TEX_CHECKER | F_MAT_METAL | F_MAT_DIELECTRIC | F_MAT_LIGHT;
constexpr uint32_t kVariantSpheres = kVariantSpheresOnly | F_RECT;

static bool is_pinhole(const WfArgs& a) { return a.r.cam.lens_radius == 0.f && !(a.r.scene.features & F_MOVING); }

hipError_t wf_launch_init(const WfArgs& a, hipStream_t stream) {
    hipError_t e;
    if ((e = hipMemsetAsync(a.st.ctl, 0, WF_CTL_WORDS * sizeof(uint32_t), stream)) != hipSuccess) return e;
    dim3 grid((a.capacity + kWfBlock - 1) / kWfBlock), block(kWfBlock);
    if ((size_t)((grid.x + WF_LANES - 1) / WF_LANES) * kWfBlock > a.lane_capacity) return hipErrorInvalidValue;
    if (is_pinhole(a)) hipLaunchKernelGGL(wf_init_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(wf_init_kernel<false>, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t wf_launch_finish(const WfArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(wf_finish_kernel, dim3(1), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// Workgroups the device keeps resident of the walk kernel with `lds_bytes` of stack (asked once per variant, device
// and size).
template <uint32_t F>
static hipError_t walk_grid(size_t lds_bytes, uint32_t* out) {
    static std::mutex mu;
    static uint32_t cached = 0;
    static size_t cached_for = 0;
    static int cached_device = -1;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (!cached || cached_for != lds_bytes || cached_device != device) {
        int cus = 0, per_cu = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wf_walk_kernel<F>, kWfBlock, lds_bytes)) != hipSuccess) return e;
        cached = (uint32_t)(cus * per_cu > 0 ? cus * per_cu : 1);
        cached_for = lds_bytes;
        cached_device = device;
        if (std::getenv("MIRT_WF_DEBUG")) fprintf(stderr, "wf_walk_kernel: %d CUs x %d workgroups resident with %zu bytes of LDS stack\n", cus, per_cu, lds_bytes);
    }
    *out = cached;
    return hipSuccess;
}

template <uint32_t F>
static hipError_t launch_round(const WfArgs& a, hipStream_t stream) {
    // a workgroup per 256 entries of every queue lane (workgroup b: lane b mod WF_LANES, entries 256 * (b / WF_LANES) ...)
    const dim3 pool(WF_LANES * (a.lane_capacity / kWfBlock)), block(kWfBlock);
    hipLaunchKernelGGL(wf_prep_kernel, dim3(1), dim3(WF_LANES), 0, stream, a);
    hipLaunchKernelGGL(wf_lists_kernel<F>, pool, block, 0, stream, a);
    if constexpr ((F & F_BVH) != 0) {
        if (a.r.scene.n_bvh_entries) {
            uint32_t resident = 0;
            const size_t lds_bytes = (size_t)(a.r.scene.stack_limit + kBvhStackSlack) * kWfBlock * sizeof(uint32_t);
            hipError_t e = walk_grid<F>(lds_bytes, &resident);
            if (e != hipSuccess) return e;
            // At least WF_LANES workgroups, resident at once or not: a workgroup starts at queue lane blockIdx mod
            // WF_LANES and draws from a few lanes from there on, so fewer workgroups than lanes (a small device or
            // partition) would leave some lanes' records unwalked. (pool.x >= WF_LANES.)
            hipLaunchKernelGGL(wf_walk_kernel<F>, dim3(std::min(std::min(std::max(resident, (uint32_t)WF_LANES), pool.x), WF_MAX_WALK_GROUPS)), block, lds_bytes,
                               stream, a);
        }
    }
    const bool pin = is_pinhole(a);
    if (!a.by_class) {
        if (pin) hipLaunchKernelGGL((wf_shade_kernel<F, true, WC_ALL>), pool, block, 0, stream, a);
        else hipLaunchKernelGGL((wf_shade_kernel<F, false, WC_ALL>), pool, block, 0, stream, a);
        return hipGetLastError();
    }
    // one shading launch per class the scene's materials can produce; every launch covers the pool (a workgroup whose
    // part of its class lane is empty returns at once) and all of them append to the next round's queue
    hipLaunchKernelGGL(wf_classify_kernel, pool, block, 0, stream, a);
    const uint32_t f = a.r.scene.features;
#define MIRT_SHADE_CLASS(C)                                                                              \
    do {                                                                                                 \
        if (pin) hipLaunchKernelGGL((wf_shade_kernel<F, true, C>), pool, block, 0, stream, a);           \
        else hipLaunchKernelGGL((wf_shade_kernel<F, false, C>), pool, block, 0, stream, a);              \
    } while (0)
    MIRT_SHADE_CLASS(WC_ENDED);
    MIRT_SHADE_CLASS(WC_DIFFUSE);
    if constexpr ((F & (F_MAT_METAL | F_MAT_DIELECTRIC)) != 0) {
        if (f & (F_MAT_METAL | F_MAT_DIELECTRIC)) MIRT_SHADE_CLASS(WC_SPECULAR);
    }
    if constexpr ((F & F_MAT_ISOTROPIC) != 0) {
        if (f & F_MAT_ISOTROPIC) MIRT_SHADE_CLASS(WC_ISO);
    }
#undef MIRT_SHADE_CLASS
    return hipGetLastError();
}

hipError_t wf_launch_round(const WfArgs& a, hipStream_t stream) {
    const uint32_t f = a.r.scene.features;
    if ((f & ~kVariantBox) == 0) return launch_round<kVariantBox>(a, stream);
    if ((f & ~kVariantSmoke) == 0) return launch_round<kVariantSmoke>(a, stream);
    if ((f & ~kVariantSpheresOnly) == 0) return launch_round<kVariantSpheresOnly>(a, stream);
    if ((f & ~kVariantSpheres) == 0) return launch_round<kVariantSpheres>(a, stream);
    return launch_round<F_ALL>(a, stream);
}

}  // namespace mirt
