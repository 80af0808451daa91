// The scene handle of the C ABI (include/rrt.h: rrt_scene) and what the host-side sources that work on it share.
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "rrt_error.hpp"
#include "rrt_launch.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
// the same for functions that own device buffers: `cleanup()` runs before the return
#define HIP_TRY_OR(cleanup, expr)                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { cleanup(); return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// What rrt_render_multi keeps per scene handle from call to call: the handle's frame, counters, packed tiles, stream and events on
// the handle's own device, and the buffer its share lands in on the first handle's device.
struct rrt_multi_state {
    float* d_img = nullptr;
    size_t img_bytes = 0;
    uint64_t* d_cnt = nullptr;
    float* d_pack = nullptr;
    size_t pack_bytes = 0;
    float* landing = nullptr;
    size_t landing_bytes = 0;
    int landing_device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t begin = nullptr, end = nullptr, landed = nullptr;
};

struct rrt_scene {
    rrt::SceneData d{};
    rrt_multi_state multi;
    std::vector<void*> owned;
    float* slab = nullptr;  // per-sample radiance staging, allocated on first render
    uint32_t* d_items = nullptr;   // item counters of the persistent kernels (k_trace_pool: one word; k_trace_stream: eight, one per XCD band), zeroed on the stream before every launch
    uint32_t wave_places = 0;      // k_trace_pool: waves the device holds at kPoolWaves per SIMD (the persistent grid's size)
    bool pooling = true;           // shallow plain scenes take k_trace_pool (rrt_scene_set_pooling)
    size_t slab_bytes = 0;
    size_t slab_cap = rrt::kSlabCapBytes;
    uint32_t res_x = 0, res_y = 0, samples = 0, stack_depth = 1;
    uint32_t lds_stack = rrt::kLdsStackEntries;  // deep trees: stack entries per lane that k_trace_stream keeps in LDS
    uint32_t features = 0;  // F_* bits the scene needs
    bool deep = false;      // many-node tree: the streaming kernels (per-lane walk, cooperative leaf phases)
    bool clean = false;     // tree built with RRT_AGGREGATE_BVH_CLEAN: traversal mode kAabbClean, whatever the caller asks for
    bool harvest = false;   // single leaf, 1..kHarvestLights lights, each one a primitive of the leaf: light densities come out of the closest-hit query
    uint32_t n_spheres = 0, n_media = 0, n_btris = 0, n_materials = 0, n_textures = 0;
    int device = 0;
    double upload_seconds = 0.0;  // host wall clock of rrt_scene_create
    double upload_parts[4] = {0.0, 0.0, 0.0, 0.0};  // of which: the first device call | the code object | host-side preparation | allocations and copies
    uint64_t desc_hash = 0;       // of the description's arrays and camera (rrt_render_multi compares its handles)
    // optional per-kernel timing: three events per pass (before k_trace, between the kernels, after k_accumulate)
    bool timing = false;
    std::vector<hipEvent_t> events;  // pool
    size_t events_used = 0;          // multiples of three
};

namespace rrt {

constexpr size_t kStaticSlabBytes = RRT_NOALLOC_STAGING_BYTES;  // the staging area of rrt_render_device_noalloc (rrt_noalloc_area.hip)

// A scene's buffers live on the device that was current in rrt_scene_create; launching from another device would hand
// the kernels pointers that device cannot reach.
int check_device(const rrt_scene* s);
// Enqueues the passes of one render (rrt_render.cpp).
int render_passes(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream, bool static_staging);
// Sums the (trace, accumulate) durations of the passes recorded since `mark` and forgets them again unless the caller's own
// timing is on (then they stay for rrt_scene_kernel_times).  The stream must have been synchronised.
int passes_since(rrt_scene* s, size_t mark, bool keep, double* trace_s, double* accumulate_s);

}  // namespace rrt
