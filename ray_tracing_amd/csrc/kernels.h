// Launch interface between the C-ABI layer (mirt_api.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace mirt {

struct RenderArgs {
    DevScene scene;
    DevCamera cam;
    uint32_t width, height;
    uint32_t spp_begin, spp_count;
    uint32_t max_depth, seed, jitter;
    float4* accum;                      // width*height, pixel y*width + x, y = 0 at the bottom
    unsigned long long* segment_counter;
    uint32_t* work_counter;             // item counters of the persistent kernel (kWorkCounterBytes): zero before every launch
    uint32_t loop;                      // LOOP_*
    // Hierarchy walks in the persistent kernel, per iteration: a lane takes at most walk_budget node / leaf steps;
    // the walk phase ends for the whole wave when fewer than walk_min_lanes lanes are still walking; the descent
    // through interior nodes pauses for a leaf round when fewer than walk_node_lanes lanes are still descending.
    // A lane whose walk is not over continues in the next iteration.
    uint32_t walk_budget, walk_min_lanes, walk_node_lanes;
    // pool != 0: the hierarchy walks of a workgroup are pooled in LDS and advanced by its four waves together, in rounds
    // of one step per ray (dev_pool.h); the rounds of an iteration stop when fewer than pool_stop rays are left.
    // pool_steps: node steps a lane takes on its ray per round while the ray keeps descending (1 = one step per round).
    uint32_t pool, pool_stop, pool_steps;
    uint32_t item_counters;             // counters the persistent kernel's items are shared out to: a power of two, 1 .. kMaxItemCounters
    // The persistent kernel's work items are (pixel, chunk of chunk_spp samples): n_chunks per pixel. With
    // more than one chunk the item sums go to partial[chunk * width * height + pixel] and are added to
    // accum in chunk order by a second kernel.
    uint32_t n_chunks, chunk_spp;
    float4* partial;
    // Every slot of `partial` a launch fills carries this launch's stamp in its fourth word; the pass that adds the
    // slots up counts a chunk's samples only when the stamp is there, so an item that was never stored shows as a
    // short sample count (and FAULT_STALE_CHUNK) instead of passing off an earlier launch's sums. Never 0.
    uint32_t stamp;
    uint32_t debug_drop_pixel;  // test build (-DMIRT_DEBUG_HOOKS) only: the item of this pixel's first chunk is not stored (0xffffffff: none)
    uint32_t count_cost;        // tally the hierarchy walks' node steps and leaf tests into the statistics (slower: not for timed launches)
};
enum : uint32_t { LOOP_AUTO = 0, LOOP_FUSED = 1, LOOP_QUEUE = 2 };
constexpr uint32_t kMaxItemCounters = 32;                 // one bit each in the kernel's ItemQueue::dead
constexpr size_t kWorkCounterBytes = kMaxItemCounters * 128;  // the item counters, one per 128-byte line
// Whether launch_render_fused() renders these arguments with the persistent pixel-queue kernel.
bool render_uses_queue(const RenderArgs& a);

// The library's static scratch: a file-scope device array (kernels.hip) that mirt_render_static renders from instead of
// allocated buffers — item counters in its first kWorkCounterBytes, chunk sums from kStaticPartialOffset on. Every
// device the library runs on holds its own copy at its own address; static_scratch_address() returns the copy of the
// CURRENT device (looked up once per device with hipGetSymbolAddress and kept).
constexpr size_t kStaticPartialOffset = 4096;
constexpr size_t kStaticPartialBytes = 64u << 20;  // two chunk images of a 1920x1080 frame (33.2 MB each); a frame whose two images do not fit renders unchunked
hipError_t static_scratch_address(void** out);
// Stamp for the next launch that fills the CURRENT device's static scratch (RenderArgs::stamp): counted per device,
// because every context on a device shares that device's array — a per-context count would let two contexts issue
// the same stamp and take each other's slots for their own. Never 0.
uint32_t static_scratch_next_stamp();

struct RayBatchArgs {
    DevScene scene;
    const float* rays;  // n x 7: origin, direction, time
    uint32_t n;
    uint32_t sample, max_depth, seed;
    int32_t* prim_id;   // first_hit
    float* t;           // first_hit
    float* aov;         // first_hit, optional, n x 12
    float* color;       // ray_color, n x 3
};

// Renders a sample range into a.accum with the persistent kernel or its one-pixel-per-lane form (a.loop): camera
// ray, then one ray segment per loop iteration (trace + shade) with the path state in registers. With
// a.n_chunks > 1 the launch is followed by the pass that adds the chunk sums in a.partial to a.accum.
hipError_t launch_render_fused(const RenderArgs& a, hipStream_t stream);
// The pass that adds the stamped chunk sums in a.partial to a.accum (after a launch with a.n_chunks > 1).
hipError_t launch_reduce_chunks(const RenderArgs& a, hipStream_t stream);
hipError_t launch_first_hit(const RayBatchArgs& a, hipStream_t stream);
hipError_t launch_ray_color(const RayBatchArgs& a, hipStream_t stream);
hipError_t launch_resolve_rgba8(const float4* accum, uint8_t* frame, uint32_t width, uint32_t height, hipStream_t stream);

}  // namespace mirt
