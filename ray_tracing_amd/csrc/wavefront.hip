// Wavefront form of the sample loop (the second kernel set; the fused kernel of kernels.hip is the
// first). Path state lives in device memory, one slot per pixel, and every ray segment goes through
// three launches:
//
//   wf_generate  slots that start a sample: cal_norm_coords + Camera::get_ray (renderer.rs:102-103)
//   wf_extend    scene.world.hit for every live path (renderer.rs:201): list scan / BVH traversal with
//                the per-lane stack in LDS; misses are closed here, hits are appended to the queue of
//                their material kind (wave ballot + prefix count, one atomic per wave and queue)
//   wf_shade     one material kind per workgroup: emit / scatter / mixture pdf (renderer.rs:203-227);
//                finished paths are accumulated and their slot is queued for its next sample
//
// A launch only carries the registers of its own stage (the traversal kernel has no shading state),
// which is what a hierarchy-bound scene needs: more resident waves to hide node-fetch latency.
// Draws are addressed exactly as in the fused kernel (Philox key = pixel, sample; counter = block,
// bounce, stream, seed), so both kernel sets produce the same samples.
#include "wavefront.h"
#include "dev_core.h"

namespace mirt {

constexpr int kWfBlock = 256;

// Append `slot` to the queue `q` for the lanes where `pred` holds: lanes are ranked with a ballot and
// a prefix population count and the wave reserves its range with a single atomic.
__device__ __forceinline__ void queue_push(bool pred, uint32_t slot, uint32_t* q, uint32_t* count) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (pred) q[base + rank] = slot;
}

// renderer.rs:107-110 for one finished sample of the pixel that owns `slot` (no other path of that
// pixel is in flight, so the read-modify-write needs no atomic).
__device__ __forceinline__ void accumulate(float4* accum, uint32_t slot, V3 e) {
    float4 v = accum[slot];
    v.x += sanitize(e.x); v.y += sanitize(e.y); v.z += sanitize(e.z); v.w += 1.f;
    accum[slot] = v;
}

__global__ __launch_bounds__(kWfBlock) void wf_reset_kernel(WfState st, uint32_t n, uint32_t spp_begin) {
    uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
    if (i >= n) return;
    st.q_new[i] = i;
    st.ids[i] = make_uint4(spp_begin, 0u, 0u, 0u);
}

template <bool PINHOLE>
__global__ __launch_bounds__(kWfBlock) void wf_generate_kernel(WfArgs a) {
    const uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
    const bool on = i < a.n_new;
    uint32_t slot = 0;
    if (on) {
        slot = a.st.q_new[i];
        const uint32_t sample = a.st.ids[slot].x;
        const uint32_t x = slot % a.width, y = slot / a.width;
        Rng g;
        g.init(slot, sample, 0u, STREAM_CAMERA, a.seed);
        Ray r = camera_ray<PINHOLE>(a.cam, x, y, a.width, a.height, a.jitter != 0, g);
        a.st.rayo[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
        a.st.rayd[slot] = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
        a.st.thr[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
        a.st.ids[slot] = make_uint4(sample, 0u, a.max_depth, 0u);
    }
    queue_push(on, slot, a.st.q_cur, a.st.counts + WF_N_CUR);
}

template <uint32_t F>
__global__ __launch_bounds__(kWfBlock) void wf_extend_kernel(WfArgs a) {
    __shared__ uint32_t bvh_stack[(F & F_BVH) ? kBvhStackDepth * kWfBlock : 1];
    uint32_t* stack = bvh_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
    const bool on = i < a.n_cur;
    uint32_t slot = 0, kind = 0xffffffffu;
    bool renew = false;
    if (on) {
        slot = a.st.q_cur[i];
        const float4 o = a.st.rayo[slot], d = a.st.rayd[slot];
        uint4 id = a.st.ids[slot];
        Ray ray{v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w};
        DrawKey key{slot, id.x, id.y, a.seed};
        float t;
        int prim;
        world_hit<F>(a.scene, ray, key, stack, kWfBlock, t, prim);
        if (prim < 0) {
            const float4 th = a.st.thr[slot];
            accumulate(a.accum, slot, cmul(v3(th.x, th.y, th.z), v3(a.scene.background)));  // renderer.rs:233-235
            id.x += 1u;
            a.st.ids[slot] = id;
            renew = id.x < a.sample_end;
        } else {
            a.st.hit_t[slot] = t;
            id.w = (uint32_t)prim;
            a.st.ids[slot] = id;
            kind = a.scene.shade[prim].mat_kind;
        }
    }
    uint32_t segs = on ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) segs += __shfl_down(segs, off);
    if ((threadIdx.x & 63u) == 0 && segs) atomicAdd(a.segment_counter, (unsigned long long)segs);
    queue_push(renew, slot, a.st.q_new, a.st.counts + WF_N_NEW);
#pragma unroll
    for (uint32_t k = 0; k < WF_KINDS; ++k) queue_push(kind == k, slot, a.st.q_mat + (size_t)k * a.capacity, a.st.counts + WF_N_MAT + k);
}

// One material kind per workgroup: workgroup b serves queue k when it falls into k's range of
// ceil(count_k / 256) workgroups.
template <uint32_t F>
__global__ __launch_bounds__(kWfBlock) void wf_shade_kernel(WfArgs a) {
    uint32_t b = blockIdx.x, k = 0, cnt = 0;
    for (; k < WF_KINDS; ++k) {
        cnt = a.st.counts[WF_N_MAT + k];
        const uint32_t blocks = (cnt + kWfBlock - 1) / kWfBlock;
        if (b < blocks) break;
        b -= blocks;
    }
    if (k == WF_KINDS) return;
    const uint32_t i = b * kWfBlock + threadIdx.x;
    const bool on = i < cnt;
    const Tables tb{a.scene.prims, a.scene.shade, a.scene.instances, a.scene.light_prims};
    uint32_t slot = 0;
    bool cont = false, renew = false;
    if (on) {
        slot = a.st.q_mat[(size_t)k * a.capacity + i];
        const float4 o = a.st.rayo[slot], d = a.st.rayd[slot], th = a.st.thr[slot];
        uint4 id = a.st.ids[slot];
        Ray ray{v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w};
        V3 thr = v3(th.x, th.y, th.z);
        const int prim = (int)id.w;
        const DevShade m = tb.shade[prim];
        const HitInfo h = finalize_hit<F>(tb, ray, a.st.hit_t[slot], prim);
        bool ended;
        V3 e = v3(0.f, 0.f, 0.f);
        if ((F & F_MAT_LIGHT) && m.mat_kind == MK_DIFFUSE_LIGHT) {
            e = cmul(thr, emitted_radiance<F>(a.scene, m, h));
            ended = true;
        } else {
            Rng g;
            g.init(slot, id.x, id.y, STREAM_SCATTER, a.seed);
            const bool scattered = scatter<F>(a.scene, tb, m, h, g, ray, thr);
            id.y += 1u;
            id.z -= 1u;
            ended = !scattered || id.z == 0u;  // absorbed, or renderer.rs:198-200: the next level returns 0
        }
        if (ended) {
            accumulate(a.accum, slot, e);
            id.x += 1u;
            renew = id.x < a.sample_end;
        } else {
            a.st.rayo[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time);
            a.st.rayd[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.f);
            a.st.thr[slot] = make_float4(thr.x, thr.y, thr.z, 0.f);
            cont = true;
        }
        a.st.ids[slot] = id;
    }
    queue_push(cont, slot, a.st.q_next, a.st.counts + WF_N_NEXT);
    queue_push(renew, slot, a.st.q_new, a.st.counts + WF_N_NEW);
}

// ---------------------------------------------------------------------------------------------
constexpr uint32_t kVariantBox = F_SPHERE | F_RECT | F_INSTANCE | F_MAT_DIELECTRIC | F_MAT_LIGHT | F_MAT_METAL;
constexpr uint32_t kVariantSmoke = kVariantBox | F_MEDIUM | F_MAT_ISOTROPIC;
constexpr uint32_t kVariantSpheres = F_SPHERE | F_MOVING | F_BVH | F_TEX_CHECKER | F_MAT_METAL | F_MAT_DIELECTRIC | F_MAT_LIGHT | F_RECT;

template <uint32_t F>
static hipError_t launch_stage(int stage, const WfArgs& a, uint32_t threads, hipStream_t stream) {
    dim3 grid((threads + kWfBlock - 1) / kWfBlock), block(kWfBlock);
    if (grid.x == 0) return hipSuccess;
    if (stage == 1) hipLaunchKernelGGL(wf_extend_kernel<F>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(wf_shade_kernel<F>, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t wf_launch_reset(const WfState& st, uint32_t n, uint32_t spp_begin, hipStream_t stream) {
    hipLaunchKernelGGL(wf_reset_kernel, dim3((n + kWfBlock - 1) / kWfBlock), dim3(kWfBlock), 0, stream, st, n, spp_begin);
    return hipGetLastError();
}

hipError_t wf_launch_generate(const WfArgs& a, hipStream_t stream) {
    if (a.n_new == 0) return hipSuccess;
    dim3 grid((a.n_new + kWfBlock - 1) / kWfBlock), block(kWfBlock);
    const bool pinhole = a.cam.lens_radius == 0.f && !(a.scene.features & F_MOVING);
    if (pinhole) hipLaunchKernelGGL(wf_generate_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(wf_generate_kernel<false>, grid, block, 0, stream, a);
    return hipGetLastError();
}

static hipError_t dispatch_stage(int stage, const WfArgs& a, uint32_t threads, hipStream_t stream) {
    const uint32_t f = a.scene.features;
    if ((f & ~kVariantBox) == 0) return launch_stage<kVariantBox>(stage, a, threads, stream);
    if ((f & ~kVariantSmoke) == 0) return launch_stage<kVariantSmoke>(stage, a, threads, stream);
    if ((f & ~kVariantSpheres) == 0) return launch_stage<kVariantSpheres>(stage, a, threads, stream);
    return launch_stage<F_ALL>(stage, a, threads, stream);
}

hipError_t wf_launch_extend(const WfArgs& a, hipStream_t stream) { return dispatch_stage(1, a, a.n_cur, stream); }

// One workgroup range per material queue: at most ceil(n_cur / 256) + WF_KINDS workgroups have work.
hipError_t wf_launch_shade(const WfArgs& a, hipStream_t stream) {
    if (a.n_cur == 0) return hipSuccess;
    return dispatch_stage(2, a, a.n_cur + WF_KINDS * kWfBlock, stream);
}

}  // namespace mirt
