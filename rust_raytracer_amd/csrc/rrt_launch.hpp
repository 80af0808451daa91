// What the two translation units of kernels share: launch constants, the argument block of the trace kernels, and the launchers
// of the deep-tree kernels (rrt_stream.hip), which are compiled apart from the shallow-tree kernels (rrt_api.hip) because the two
// want different compiler settings: the SLP vectoriser helps the node walk of the deep-tree kernels (+4 %) and costs the shading
// and sifting code of the others 1 - 9 % (it packs scalar arithmetic into two-lane instructions that need extra register moves).
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_device.hpp"

namespace rrt {

constexpr int kBlock = 256;              // k_intersect: 4 waves per block
constexpr int kTraceBlock = 64;          // k_trace: one wave per block, the unit the hardware dispatcher balances
#ifndef RRT_CHUNK
#define RRT_CHUNK 64
#endif
constexpr uint32_t kChunk = RRT_CHUNK;   // samples of each pixel handled by one wave item
constexpr int kWalkSteps = 4;            // k_intersect_stream: node steps per turn of the wave's loop (TraceArgs::walk_steps for k_trace_stream)
constexpr uint32_t kMaxStackDepth = 64;  // traversal stack: entries per lane
constexpr uint32_t kLdsStackEntries = 32;  // k_trace_stream keeps this many of them in LDS (10 KB per wave with the cooperative area: 16 waves
                                           // per compute unit), the rest in private memory
constexpr uint32_t kDeepTreeNodes = 16;  // trees with more nodes than this take the cooperative-leaf walk
constexpr size_t kMaxTimedPasses = 4096;   // render passes whose events a scene handle keeps between two rrt_scene_kernel_times calls
constexpr size_t kSlabCapBytes = 16ull << 30;  // default budget of the per-sample radiance staging buffer (per scene handle, per GPU):
                                               // sized for 288 GB of HBM, so that a 1024x1024 frame at 1024 spp is a single pass


struct TraceArgs {
    uint32_t W, H;
    uint32_t first_sample;  // stream index of the pass's first sample
    uint32_t n_samples;     // samples per pixel in this pass
    uint32_t seed, bg, max_depth;
    uint32_t tile_count, tile_index, tiles_x, n_tiles;  // 32x32-pixel tiles (multi-GPU sharding unit)
    uint32_t n_t8;                                       // 8x8-pixel wave tiles owned by this call
    uint32_t stack_depth;                                // LDS stack entries per lane (the cooperative-leaf area follows the stack)
    uint32_t chunk;                                      // samples of each pixel handled by one wave item (divides kChunk)
    uint32_t coop_lanes, shade_lanes;                    // k_trace_stream: paused / finished lanes that trigger a leaf / shading phase
    uint32_t walk_steps;                                 // k_trace_stream: node steps per turn of the wave's loop
    uint32_t n_items;                                    // k_trace_pool: (tile, chunk) items of the pass, handed out through an atomic counter
};

#ifndef RRT_TRACE_MIN_WAVES
#define RRT_TRACE_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif

// k_trace_stream<mode, features>: mode RRT_AABB_TIGHT / RRT_AABB_REFERENCE / kAabbClean; `features` = the F_* bits the scene needs (three variants:
// triangles and spheres with constant textures, everything but nested media, everything).  The grid is persistent: the wave items are handed out through item_counter[0..7] (one counter per XCD's
// band of the frame), which must be zero when the kernel starts.  Returns hipGetLastError() of the launch.
hipError_t launch_trace_stream(int mode, uint32_t features, dim3 grid, size_t lds, hipStream_t st, const DScene& sc, const TraceArgs& ta, float* slab,
                               unsigned long long* counters, uint32_t* item_counter);
// k_intersect_stream<mode>
hipError_t launch_intersect_stream(int mode, dim3 grid, size_t lds, hipStream_t st, const DScene& sc, const float* o, const float* d, size_t n, float t_min,
                                   float t_max, uint32_t seed, uint32_t lds_entries, uint32_t run, uint32_t coop_lanes, int* prim, float* t_hit);

// The current device's copy of the built-in staging area of rrt_render_device_noalloc (rrt_noalloc_area.hip); returns an RRT_* code.
int noalloc_area(int device, float** out);

#ifdef RRT_PHASE_PROFILE
// profiling build only: reads and clears the counters of the deep-tree kernels
hipError_t profile_take_stream(uint64_t* out8);
#endif

}  // namespace rrt
