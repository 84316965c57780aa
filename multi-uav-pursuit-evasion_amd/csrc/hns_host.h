// hns_host.h — host-side state shared by the translation units of libhns.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hns.h"

namespace hns {
struct Params;
}
// the step kernels' signature (hns_common.h: HNS_STEP_PARAMS)
typedef void (*hns_step_fn)(const hns::Params *, const float *, float *, float *, const void *, float *, float *, float *);

void hns_set_error(const std::string &m);

// trajectory-predictor binding (hns_tp.hip)
struct hns_tp_state {
    hns_tp_buffers buf;
    int history_step = 0, future_step = 0;
    bool bound = false;
    bool dirty = true;     // operand image out of date (parameters changed)
};

struct hns_env {
    hns_cfg cfg;
    hns_buffers buf;
    bool bound = false;
    int device = 0;          // the HIP device that was current at hns_create; bound buffers must live there
    uint32_t epoch = 0;
    int grid = 0, threads = 0;   // threads: the reset kernel's workgroup
    int threads_step = 0;        // the step kernel's workgroup (the tile mapping: as the reset kernel's; the small-batch mapping: 2 A + 1 waves)
    int cus = 0;                 // compute units of `device`
    int small_mapping = 0;       // 1: hns_step_small_kernel serves this env (hns_inst.hip)
    int prio_boost = 0;          // 1: the tile mapping's pursuer waves start at priority 1 (hns_inst.hip; Params::prio_boost)
    size_t lds_step = 0, lds_reset = 0;
    void (*reset_fn)(const hns::Params) = nullptr;
    hns_step_fn step_args_fn = nullptr;        // the step kernel instantiation serving this env (hns_inst.hip)
    hns_step_fn step_args_prof_fn = nullptr;   // ... compiled with the per-wave phase stamps (hns_set_phase_profile); whole tiles, k <= 4 only
    // device copy of the step launch's Params (allocated by hns_create) and what feeds it: a ring of pinned host images, one
    // stream-ordered hipMemcpyAsync per change on `last_stream` (the stream of the latest step / reset / observe call)
    static constexpr int kParamRing = 8;
    hns::Params *params_dev = nullptr, *params_host = nullptr;   // params_host: the image last enqueued (ordinary memory, for comparison)
    hns::Params *params_ring = nullptr;                          // pinned, [kParamRing]
    hipEvent_t ring_events[kParamRing] = {};                     // recorded behind each slot's copy
    bool ring_pending[kParamRing] = {};
    int ring_next = 0;
    // pinned images for changes made INSIDE a stream capture (they must outlive the graph, so they are never reused): a fixed pool made
    // by hns_create — hipHostMalloc is illegal while a capture is active in the default (global) capture mode
    static constexpr int kCaptureImages = 16;
    hns::Params *capture_pool = nullptr;                         // pinned, [kCaptureImages]
    int capture_used = 0;
    hipStream_t last_stream = nullptr;
    bool params_valid = false;
    unsigned long long *prof = nullptr;
    uint32_t cyl_magic = 0;
    int timing = 0;          // 0 = off, n = bracket every n-th step launch with hipEvents
    uint64_t step_count = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded, not yet harvested
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;     // free event pairs
    hipEvent_t region_ev[2] = {};                            // hns_region_begin / hns_region_end (made by the first hns_region_begin)
    int region_state = 0;                                    // 0 none, 1 begun, 2 complete
    hns_tp_state tp;
};

// true iff `ptr` is device (or managed) memory of the GPU the env was created on
inline bool hns_on_env_device(const hns_env *env, const void *ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
    return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) && attr.device == env->device;
}

#define HNS_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            hns_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));      \
            return HNS_ERR_DEVICE;                                                 \
        }                                                                          \
    } while (0)
