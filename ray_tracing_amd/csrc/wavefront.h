// Launch interface of the wavefront kernel set (wavefront.hip): path state in device memory, one slot
// per pixel, queues of slot indices between the stages.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace mirt {

constexpr uint32_t WF_KINDS = 5;  // MK_LAMBERTIAN .. MK_ISOTROPIC
// counters (uint32 each)
enum : uint32_t { WF_N_NEW = 0, WF_N_CUR = 1, WF_N_NEXT = 2, WF_N_MAT = 3, WF_N_COUNTS = WF_N_MAT + WF_KINDS };

struct WfState {
    float4* rayo;   // origin.xyz, time
    float4* rayd;   // direction.xyz, -
    float4* thr;    // throughput.xyz, -
    uint4* ids;     // sample, bounce, depth_left, hit primitive
    float* hit_t;
    uint32_t* q_new;   // slots that start their next sample
    uint32_t* q_cur;   // live paths of this iteration
    uint32_t* q_next;  // paths that scattered and go on
    uint32_t* q_mat;   // WF_KINDS queues of `capacity` entries: hits by material kind
    uint32_t* counts;  // WF_N_COUNTS counters
};

struct WfArgs {
    DevScene scene;
    DevCamera cam;
    WfState st;
    float4* accum;
    unsigned long long* segment_counter;
    uint32_t width, height, capacity;
    uint32_t n_new, n_cur;
    uint32_t sample_end, max_depth, seed, jitter;
};

hipError_t wf_launch_reset(const WfState& st, uint32_t n, uint32_t spp_begin, hipStream_t stream);
hipError_t wf_launch_generate(const WfArgs& a, hipStream_t stream);
hipError_t wf_launch_extend(const WfArgs& a, hipStream_t stream);
hipError_t wf_launch_shade(const WfArgs& a, hipStream_t stream);

}  // namespace mirt
