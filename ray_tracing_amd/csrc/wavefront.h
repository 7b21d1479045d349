// Launch interface of the wavefront kernel set (wavefront.hip): the sample loop as rounds of small kernels over a
// pool of path slots in device memory, driven from the device — the host enqueues rounds without reading anything
// back and only looks at the live count once per batch of rounds.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "kernels.h"

namespace mirt {

// One path per slot, structure of arrays (80 bytes per slot).
struct WfState {
    float4* rayo;  // origin.xyz, time
    float4* rayd;  // direction.xyz, t of the closest hit so far
    float4* thr;   // throughput.xyz, bits of the hit primitive (int, -1 = none)
    float4* acc;   // sample sums of the slot's work item .xyz, bits of its pixel position x | y << 16
    uint4* ids;    // pixel, sample, end of the item's sample range, bounce
    uint32_t* q[2];  // live slots: round r reads q[r & 1] and fills q[(r + 1) & 1]
    uint32_t* ctl;   // control words, WF_CTL_* (each in its own 128-byte line)
};
// control words (indices of 32-bit words; a stride of 32 words keeps the contended ones in separate lines)
enum : uint32_t {
    WF_CTL_STRIDE = 32,
    WF_CTL_N0 = 0 * WF_CTL_STRIDE,      // entries in q[0]
    WF_CTL_N1 = 1 * WF_CTL_STRIDE,      // entries in q[1]
    WF_CTL_CURSOR = 2 * WF_CTL_STRIDE,  // fetch position of the walk kernel in this round's queue
    WF_CTL_ITEM = 3 * WF_CTL_STRIDE,    // next work item to hand out
    WF_CTL_WORDS = 4 * WF_CTL_STRIDE,
};

struct WfArgs {
    RenderArgs r;       // scene, camera, frame, sample range, chunking (n_chunks, chunk_spp, partial, stamp), statistics
    WfState st;
    uint32_t capacity;  // slots in the pool
    uint32_t round;     // parity selects the queues
    uint32_t refill_lanes;  // the walk kernel fetches new rays when this many lanes of a wave are idle
};

// Slots 0 .. capacity - 1 take the first work items, get their camera rays and fill q[0].
hipError_t wf_launch_init(const WfArgs& a, hipStream_t stream);
// One round: closest hit of every live path (list scan, then the persistent hierarchy walk), then shading /
// scattering / next sample / next item, which fills the other queue.
hipError_t wf_launch_round(const WfArgs& a, hipStream_t stream);

}  // namespace mirt
