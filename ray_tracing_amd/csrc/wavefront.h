// Launch interface of the wavefront kernel set (wavefront.hip): the sample loop as rounds of small kernels over a
// pool of path slots in device memory, driven from the device — the host enqueues rounds without reading anything
// back and only looks at the live count once per batch of rounds.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "kernels.h"

namespace mirt {

// One path per slot, structure of arrays (48 bytes per slot): what only the shading stage reads.
// The ray itself travels in the queue: an entry of a round's queue is the record of one live path —
// its ray, the closest hit found so far, its slot (40 bytes, three arrays) — stored in queue order, so the
// intersection stages read and write it with coalesced 16-byte accesses and never follow a slot index.
struct WfState {
    float4* thr;   // throughput.xyz, -
    float4* acc;   // sample sums of the slot's work item .xyz, bits of its pixel position x | y << 16
    uint4* ids;    // pixel, sample, end of the item's sample range, bounce
    // queues: round r reads [r & 1] and fills [(r + 1) & 1]; WF_LANES lanes of `lane_capacity` entries each
    float4* q_o[2];  // origin.xyz, time
    float4* q_d[2];  // direction.xyz, t of the closest hit so far
    uint2* q_h[2];   // bits of the hit primitive (int, -1 = none), slot
    uint32_t* ctl;   // control words, WF_CTL_*
    // Class queues (WfArgs::by_class): after the intersection stages every record of the round's queue is filed under
    // what its path does next — ScatterType of material.rs:19-23 plus the paths that end: WC_* below — as an index
    // into the round's queue; cls[(class * WF_LANES + lane) * lane_capacity + i], counts in the control words. The
    // shading stage then runs once per class, each launch over lanes that all take the same branch.
    uint32_t* cls;
    unsigned long long* cost;  // per workgroup of the walk kernel: node steps, leaf spheres, leaf rects, child boxes (summed at the end)
};
constexpr uint32_t WF_MAX_WALK_GROUPS = 4096;
// Every queue is WF_LANES independent lanes, and so is the hand-out of work items: a read-modify-write on one
// address costs about 35 ns on this device however many waves queue for it (measured: 32 768 waves appending to one
// counter made a kernel over 2 M paths last 1.2 ms), so the appends of a round are spread over 64 counters in
// separate 128-byte lines. A workgroup reads lane (blockIdx mod WF_LANES) and appends to the same lane of the other
// queue, so a lane never receives more than it held and `lane_capacity` bounds it for good.
constexpr uint32_t WF_LANES = 64;
// control words of lane k: ctl[k * WF_CTL_STRIDE + ...]
enum : uint32_t {
    WF_CTL_STRIDE = 32,  // 128 bytes
    WF_CTL_N0 = 0,       // entries in lane k of q[0]
    WF_CTL_N1 = 1,       // entries in lane k of q[1]
    WF_CTL_CURSOR = 2,   // fetch position of the walk kernel in lane k of this round's queue
    WF_CTL_ITEM = 3,     // work items handed out by counter k
    WF_CTL_CLASS = 4,    // four words: records of lane k filed under class 0 .. 3 this round
    WF_CTL_WORDS = (WF_LANES + 1) * WF_CTL_STRIDE,
    WF_CTL_ITEMS_SPENT = WF_LANES * WF_CTL_STRIDE,  // two words: which item counters were spent when the round began
};

// What a path does after its hit is known: it ends (miss, or an emitter: material.rs:167-175 never scatters), or its
// material scatters as ScatterType::Diffuse (Lambertian), ::Specular (Metal, Dielectric) or ::ISO (Isotropic).
enum : uint32_t { WC_ENDED = 0, WC_DIFFUSE = 1, WC_SPECULAR = 2, WC_ISO = 3, WC_COUNT = 4, WC_ALL = 0xffffffffu };

struct WfArgs {
    RenderArgs r;       // scene, camera, frame, sample range, chunking (n_chunks, chunk_spp, partial, stamp), statistics
    WfState st;
    uint32_t capacity;  // slots in the pool
    uint32_t lane_capacity;  // entries per queue lane: 256 x the workgroups of the pool that map to one lane
    uint32_t round;     // parity selects the queues
    uint32_t refill_lanes;  // the walk kernel fetches new rays when this many lanes of a wave are idle
    uint32_t fetch_run;     // ... out of a run of this many queue entries the wave reserves with one atomic
    uint32_t by_class;      // file the records by class after the intersection stages and shade class by class
};

// Slots 0 .. capacity - 1 take the first work items, get their camera rays and fill q[0].
hipError_t wf_launch_init(const WfArgs& a, hipStream_t stream);
// One round: closest hit of every live path (list scan, then the persistent hierarchy walk), then shading /
// scattering / next sample / next item, which fills the other queue.
hipError_t wf_launch_round(const WfArgs& a, hipStream_t stream);
// End of a render: the walk kernel's per-workgroup cost words go to the statistics block.
hipError_t wf_launch_finish(const WfArgs& a, hipStream_t stream);

}  // namespace mirt
