// Hierarchy walks of a whole workgroup pooled in LDS (the POOL form of the persistent render kernel).
//
// In the default form every lane walks its own ray and the 64 lanes of a wave alternate between node steps and leaf
// tests, each running with the lanes that happen to need it: a node step issues its ~85 vector instructions for
// 0.3-0.5 of the lanes. Here a ray that has to walk is a SLOT of the workgroup's pool — ray, reciprocals, closest hit,
// current reference, stack — and the walk advances in ROUNDS: the slots whose next step is an interior node are listed
// in one list, those whose next step is a leaf (or the start of their next hierarchy) in another, and the four waves
// take the lists 64 entries at a time, so a wave's step runs with as many lanes as the whole workgroup has rays of that
// kind, whoever owns them. A wave without a chunk waits at the round's barrier and issues nothing. The owner of a slot
// (the lane whose path the ray belongs to) sees the result when the slot is done; a phase of rounds ends when fewer
// than `stop_below` rays are left, and those stay in the pool and go on in the next iteration's rounds together with
// the rays submitted by then.
//
// Every ray is walked exactly as walk_run() walks it — same steps in the same order on its own stack (walk_node,
// walk_leaf) — so the closest hit is the same bit for bit; only which lane executes a step differs.
//
// Synchronisation: one workgroup barrier per round. Lists and their counts exist three times: round r reads buffer
// r mod 3, appends to (r + 1) mod 3 and clears the counts of (r + 2) mod 3, which every wave finished reading before
// it arrived at this round's barrier. All branches that lead to a barrier depend only on values read from a buffer
// nobody writes in that round, so the four waves always take them alike.
#pragma once
#include "dev_core.h"

namespace mirt {

constexpr uint32_t kPoolSlots = 256;  // = the workgroup size: slot i belongs to thread i
constexpr uint32_t kPoolListN = 0, kPoolListL = 1, kPoolDone = 2;
// LDS words of the pool behind the traversal stacks
constexpr uint32_t kPoolQuadWords = kPoolSlots * 4;                 // one float4 array
constexpr uint32_t kPoolMetaAt = 4 * kPoolQuadWords;                // uint32 meta[256]
constexpr uint32_t kPoolListAt = kPoolMetaAt + kPoolSlots;          // uint8 list[3][2][256] = 384 words
constexpr uint32_t kPoolCountAt = kPoolListAt + 3 * 2 * kPoolSlots / 4;  // uint32 cnt[3][4]
constexpr uint32_t kPoolAliveAt = kPoolCountAt + 12;                // uint32 alive[2][4]
constexpr uint32_t kPoolSubmitAt = kPoolAliveAt + 8;                // uint32 submitted[2]
constexpr uint32_t kPoolWords = kPoolSubmitAt + 4;
constexpr uint32_t kPoolBytes = kPoolWords * 4;

constexpr uint32_t kMetaDone = 0x80000000u;  // meta: sp in bits 0-7, hierarchies started in bits 8-15, bit 31 = walk complete

struct Pool {
    float4* a;       // (rinv.xyz, closest)
    float4* b;       // (nc.xyz, bits of the current reference; kWalkIdle = the next hierarchy has to be started)
    float4* c;       // (world origin, time)
    float4* d;       // (world direction, bits of the closest primitive)
    uint32_t* meta;
    uint8_t* list;
    uint32_t* cnt;
    uint32_t* alive;
    uint32_t* submitted;
};
// The workgroup's counters, the same in every thread (scalar registers), kept by the render loop across iterations.
struct PoolState {
    uint32_t round;      // rounds processed so far: round r reads buffer r mod 3
    uint32_t phase;      // calls of pool_rounds so far
    uint32_t left_node;  // entries of the node list of buffer round mod 3 that the last phase left unprocessed
    uint32_t seen0, seen1;  // value of submitted[parity] when a phase of that parity last read it (two scalars: an array indexed by the parity would live in scratch memory)
};
MIRT_DEV Pool pool_at(uint32_t* base) {
    Pool p;
    p.a = reinterpret_cast<float4*>(base);
    p.b = p.a + kPoolSlots;
    p.c = p.b + kPoolSlots;
    p.d = p.c + kPoolSlots;
    p.meta = base + kPoolMetaAt;
    p.list = reinterpret_cast<uint8_t*>(base + kPoolListAt);
    p.cnt = base + kPoolCountAt;
    p.alive = base + kPoolAliveAt;
    p.submitted = base + kPoolSubmitAt;
    return p;
}
// (one thread per word; the caller synchronises)
MIRT_DEV void pool_clear(const Pool& p) {
    if (threadIdx.x < 12) p.cnt[threadIdx.x] = 0;
    if (threadIdx.x < 8) p.alive[threadIdx.x] = 0;
    if (threadIdx.x < 2) p.submitted[threadIdx.x] = 0;
    p.meta[threadIdx.x] = kMetaDone;
}

MIRT_DEV uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The lanes with `pred` append their slot to list `kind` of buffer `buf`: one LDS atomic per wave.
MIRT_DEV void pool_append(const Pool& p, uint32_t buf, uint32_t kind, bool pred, uint32_t slot) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(p.cnt + buf * 4u + kind, (uint32_t)__popcll(m));
    base = wave_uniform(base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (pred) p.list[(buf * 2u + kind) * kPoolSlots + ((base + rank) & (kPoolSlots - 1u))] = (uint8_t)slot;
}
MIRT_DEV void pool_count_done(const Pool& p, uint32_t buf, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m != 0 && (threadIdx.x & 63u) == 0u) atomicAdd(p.cnt + buf * 4u + kPoolDone, (uint32_t)__popcll(m));
}

// The ray in the frame of hierarchy k without its reciprocals (walk_ray computes both).
template <uint32_t F>
MIRT_DEV Ray pool_local_ray(const DevScene& s, const Ray& r, uint32_t k) {
    if (!(F & F_INSTANCE)) return r;
    const uint32_t inst = s.entries[s.bvh_entries[k]].instance;
    return inst ? to_local(s.instances[inst], r) : r;
}

// Where a slot goes after a step: the node list, the leaf list (also for the start of its next hierarchy: reference
// idle, hierarchies left), or done.
MIRT_DEV void pool_file(const Pool& p, uint32_t buf, bool active, uint32_t slot, uint32_t ref, uint32_t started, uint32_t n) {
    const bool idle = ref == kWalkIdle;
    const bool done = active && idle && started >= n;
    pool_append(p, buf, kPoolListN, active && !idle && ref_is_node(ref), slot);
    pool_append(p, buf, kPoolListL, active && !done && (idle || !ref_is_node(ref)), slot);
    pool_count_done(p, buf, done);
}

// The owner of slot threadIdx.x hands its ray to the pool: hierarchy 0 is started here, at the occupancy of the lanes
// that start a segment (the root of a hierarchy is always an interior node, so the slot goes to the node list). The
// entries go behind the ones the last phase left in the node list of the buffer the next round reads. Their number is
// NOT added to that buffer's count but to a counter of its own, one per phase parity: a wave that left the last phase
// early may submit while a slow one is still reading the buffer's counts for its own decision to leave, which must
// come out the same. The next phase adds the two once its first barrier has passed.
template <uint32_t F>
MIRT_DEV void pool_submit(const DevScene& s, const Pool& p, const PoolState& st, bool pred, const Ray& r, float closest, int best) {
    const uint32_t slot = threadIdx.x;
    if (pred) {
        const LRay lr = walk_ray<F>(r, walk_start(s, 0u));
        const uint32_t ref = s.entries[s.bvh_entries[0]].bvh_root;
        p.a[slot] = make_float4(lr.rinv.x, lr.rinv.y, lr.rinv.z, closest);
        p.b[slot] = make_float4(lr.nc.x, lr.nc.y, lr.nc.z, __uint_as_float(ref));
        p.c[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
        p.d[slot] = make_float4(r.d.x, r.d.y, r.d.z, __int_as_float(best));
        p.meta[slot] = 1u << 8;  // empty stack, one hierarchy started
    }
    const unsigned long long m = __ballot(pred);
    if (m == 0) return;
    const uint32_t par = st.phase & 1u;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(p.submitted + par, (uint32_t)__popcll(m));
    base = wave_uniform(base) - (par ? st.seen1 : st.seen0) + st.left_node;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (pred) p.list[((st.round % 3u) * 2u + kPoolListN) * kPoolSlots + ((base + rank) & (kPoolSlots - 1u))] = (uint8_t)slot;
}

// One node step (walk_node) for up to 64 entries of the node list.
MIRT_DEV void pool_node_chunk(const DevScene& s, const Pool& p, uint32_t cur, uint32_t nxt, uint32_t first, uint32_t count, uint32_t n,
                              uint32_t* stacks, uint32_t node_steps) {
    const uint32_t i = first + (threadIdx.x & 63u);
    const bool active = i < count;
    uint32_t slot = 0, started = 0;
    Walk w;
    w.ref = kWalkIdle;
    if (active) {
        slot = p.list[(cur * 2u + kPoolListN) * kPoolSlots + i];
        const float4 a = p.a[slot], b = p.b[slot];
        const uint32_t m = p.meta[slot];
        w.ref = __float_as_uint(b.w);
        w.sp = m & 0xffu;
        w.closest = a.w;
        started = (m >> 8) & 0xffu;
        // up to node_steps steps while the ray keeps descending (more than one trades lanes per step for fewer rounds)
        for (uint32_t k = 0; k < node_steps && ref_is_node(w.ref); ++k)
            walk_node(s, 0.001f, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), w, stacks + slot, kPoolSlots, nullptr, nullptr);
        p.b[slot].w = __uint_as_float(w.ref);
        p.meta[slot] = w.sp | started << 8 | ((w.ref == kWalkIdle && started >= n) ? kMetaDone : 0u);
    }
    pool_file(p, nxt, active, slot, w.ref, started, n);
}

// One leaf step (walk_leaf) — or, for a slot whose hierarchy is exhausted, the start of its next one — for up to 64
// entries of the leaf list.
template <uint32_t F>
MIRT_DEV void pool_leaf_chunk(const DevScene& s, const Pool& p, uint32_t cur, uint32_t nxt, uint32_t first, uint32_t count, uint32_t n,
                              uint32_t* stacks) {
    const uint32_t i = first + (threadIdx.x & 63u);
    const bool active = i < count;
    uint32_t slot = 0, started = 0;
    Walk w;
    w.ref = kWalkIdle;
    if (active) {
        slot = p.list[(cur * 2u + kPoolListL) * kPoolSlots + i];
        const float4 c = p.c[slot], d = p.d[slot];
        const uint32_t m = p.meta[slot];
        const Ray r{v3(c.x, c.y, c.z), v3(d.x, d.y, d.z), c.w};
        w.ref = __float_as_uint(p.b[slot].w);
        w.sp = m & 0xffu;
        started = (m >> 8) & 0xffu;
        if (w.ref == kWalkIdle) {
            // walk_begin for hierarchy number `started`
            const LRay lr = walk_ray<F>(r, walk_start(s, started));
            w.ref = s.entries[s.bvh_entries[started]].bvh_root;
            w.sp = 0;
            ++started;
            const float closest = p.a[slot].w;
            p.a[slot] = make_float4(lr.rinv.x, lr.rinv.y, lr.rinv.z, closest);
            p.b[slot] = make_float4(lr.nc.x, lr.nc.y, lr.nc.z, __uint_as_float(w.ref));
        } else {
            const float4 a = p.a[slot];
            LRay lr;
            lr.r = pool_local_ray<F>(s, r, started - 1u);
            lr.rinv = v3(a.x, a.y, a.z);
            lr.nc = v3(0.f, 0.f, 0.f);  // (only node steps read it)
            w.closest = a.w;
            w.best = __float_as_int(d.w);
            uint32_t tally = 0;
            walk_leaf<F>(s, 0.001f, lr, w, stacks + slot, kPoolSlots, tally);
            p.a[slot].w = w.closest;
            p.d[slot].w = __int_as_float(w.best);
            p.b[slot].w = __uint_as_float(w.ref);
        }
        p.meta[slot] = w.sp | started << 8 | ((w.ref == kWalkIdle && started >= n) ? kMetaDone : 0u);
    }
    pool_file(p, nxt, active, slot, w.ref, started, n);
}

// The rounds of one iteration of the render loop. Returns false when no lane of the workgroup has work left (the
// render loop ends), true otherwise. The phase ends when the lists are empty, or when fewer than `stop_below` rays are
// left and at least one slot was completed in this phase (its owner has something to shade).
template <uint32_t F>
MIRT_DEV bool pool_rounds(const DevScene& s, const Pool& p, PoolState& st, bool alive, uint32_t stop_below, uint32_t node_steps, uint32_t* stacks) {
    const uint32_t wave = wave_uniform(threadIdx.x >> 6);
    const uint32_t n = s.n_bvh_entries;
    const uint32_t par = st.phase & 1u;
    // which waves still have a live lane: written before the phase's first barrier, one set of words per phase parity
    // (a wave may reach its next phase before a slow one has read this phase's words)
    const uint32_t wave_alive = __ballot(alive) != 0 ? 1u : 0u;  // (all lanes vote: not inside the branch below)
    if ((threadIdx.x & 63u) == 0u) p.alive[par * 4u + wave] = wave_alive;
    uint32_t done_in_phase = 0;
    for (uint32_t k = 0;; ++k) {
        __syncthreads();
        const uint32_t cur = st.round % 3u, nxt = (st.round + 1u) % 3u, old = (st.round + 2u) % 3u;
        uint32_t n_node = wave_uniform(p.cnt[cur * 4u + kPoolListN]);
        const uint32_t n_leaf = wave_uniform(p.cnt[cur * 4u + kPoolListL]);
        if (k == 0) {
            const uint32_t* al = p.alive + par * 4u;
            const uint32_t any = wave_uniform(al[0] | al[1] | al[2] | al[3]);
            const uint32_t sub = wave_uniform(p.submitted[par]);
            n_node += sub - (par ? st.seen1 : st.seen0);  // the rays submitted since the last phase, behind the ones it left
            if (par) st.seen1 = sub;
            else st.seen0 = sub;
            ++st.phase;
            if (!any) return false;
        } else {
            done_in_phase += wave_uniform(p.cnt[cur * 4u + kPoolDone]);
        }
        const uint32_t total = n_node + n_leaf;
        if (total == 0u || (total < stop_below && done_in_phase > 0u)) {
            st.left_node = n_node;
            return true;
        }
        if (threadIdx.x < 3u) p.cnt[old * 4u + threadIdx.x] = 0u;
        const uint32_t cn = (n_node + 63u) >> 6, cl = (n_leaf + 63u) >> 6;
        for (uint32_t c = wave; c < cn + cl; c += 4u) {
            if (c < cn) pool_node_chunk(s, p, cur, nxt, c * 64u, n_node, n, stacks, node_steps);
            else pool_leaf_chunk<F>(s, p, cur, nxt, (c - cn) * 64u, n_leaf, n, stacks);
        }
        ++st.round;
    }
}

}  // namespace mirt
