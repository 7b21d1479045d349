// The context behind the opaque mirt_ctx handle of include/mirt.h. Private to the library's host code.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "dev_types.h"
#include "wavefront.h"

// Layout of the small device block every context owns (seg_counter): kCounters 64-bit statistics words (segments,
// lane slots, hierarchy node steps, leaf sphere tests, leaf rect tests, child boxes of the visited nodes), the fault
// word the kernels OR FAULT_* bits into (byte kFaultOffset), and from byte kWorkCounterOffset the item counters of
// the persistent render kernel, each in its own 128-byte line.
constexpr int kCounters = 6;
constexpr size_t kFaultOffset = 64, kWorkCounterOffset = 128;

struct mirt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // around the launches of the last mirt_render
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;  // around the last reduce of the accumulation buffer (comm.cpp)
    bool reduce_timed = false;                  // ... which was recorded since mirt_frame_begin
    std::string err;
    std::vector<void*> scene_allocs;
    mirt::DevScene scene{};
    mirt::DevCamera cam{};
    bool has_scene = false, has_camera = false;
    float4* accum = nullptr;
    float4* accum_owned = nullptr;
    size_t accum_owned_pixels = 0;
    uint32_t width = 0, height = 0;
    uint8_t* frame_dev = nullptr;
    size_t frame_dev_bytes = 0;
    uint32_t n_bvh_nodes = 0;                   // nodes of all hierarchies of the scene
    uint32_t n_scan_records = 0;                // list-scan records of all list entries
    int bvh_deepest_leaf = 0;                   // level of the deepest hierarchy leaf (root node = level 1)
    uint32_t list_rects = 0, list_spheres = 0, list_transforms = 0;  // what one scan of all list entries tests
    float4* partial = nullptr;                  // chunk sums of the persistent kernel (grown on demand)
    size_t partial_bytes = 0;
    unsigned long long* seg_counter = nullptr;  // the device block described above
    uint32_t stamp = 0;                         // stamp of the last persistent launch (RenderArgs::stamp)
    void* static_scratch = nullptr;             // this device's copy of the library's static scratch array
    mirt_stats stats{};
    // wavefront kernel set: path state + queues for `wf_capacity` slots, counters mirrored in pinned memory
    mirt::WfState wf{};
    uint32_t wf_capacity = 0;
    std::vector<void*> wf_allocs;
    uint32_t* wf_counts_host = nullptr;
    // RCCL communicator this context is a rank of (comm.cpp); ncclComm_t kept as an untyped pointer
    void* comm = nullptr;
    int comm_ranks = 0, comm_rank = -1;
    unsigned long long comm_group = 0;          // contexts initialised together by mirt_reduce share a group number
};

namespace mirt {
int ctx_fail(mirt_ctx* c, int code, const std::string& msg);  // records msg as the context's last error, returns code
void comm_release(mirt_ctx* c);                                // destroys the context's communicator, if any
}  // namespace mirt
