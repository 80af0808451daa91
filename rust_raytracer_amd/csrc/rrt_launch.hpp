// What the host half of the library (scene handles, render planning, the multi-GPU gather: plain C++) knows of the kernels: launch
// constants, the argument block of the trace kernels, and one launcher per kernel family.  The kernels live in two translation
// units because they want different compiler settings: the SLP vectoriser helps the node walk of the deep-tree kernels
// (rrt_stream.hip, +4 %) and costs the shading and sifting code of the shallow-tree kernels (rrt_kernels_shallow.hip) 1 - 9 % (it
// packs scalar arithmetic into two-lane instructions that need extra register moves).  Every launcher picks the compiled variant,
// sizes the launch's LDS and returns hipGetLastError() of the launch.
#pragma once
#include <hip/hip_runtime_api.h>

#include "rrt_layout.hpp"

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

// ---- shallow trees (rrt_kernels_shallow.hip) ----------------------------------------------------------------------------------
// The first use of anything in that file's code object loads it onto the current device; this asks for exactly that.
hipError_t load_shallow_kernels();
// k_trace<mode, features, harvest>, one block per wave item: mode RRT_AABB_TIGHT / RRT_AABB_REFERENCE / kAabbExhaustive;
// `features` = the F_* bits the scene needs; harvest: rrt_scene::harvest
hipError_t launch_trace(int aabb_mode, uint32_t features, bool harvest, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& ta, float* slab,
                        unsigned long long* counters);
// k_trace_pool<mode, features, harvest>: a persistent grid that hands the items out itself through *item_counter, which must be zero
// when the kernel starts.  pool_takes(): whether a pass fits the pool's packed path state; pool_waves_per_simd(): the grid's
// size is that many waves on every SIMD of the device at most.
hipError_t launch_trace_pool(int aabb_mode, uint32_t features, bool harvest, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& ta, float* slab,
                             unsigned long long* counters, uint32_t* item_counter);
bool pool_takes(uint32_t max_depth, uint32_t n_samples, uint32_t n_lights, uint32_t n_slots);
uint32_t pool_waves_per_simd();
// k_intersect<mode>
hipError_t launch_intersect(int aabb_mode, hipStream_t st, const SceneData& scene, const float* o, const float* d, size_t n, float t_min, float t_max,
                            uint32_t seed, uint32_t stack_depth, int* prim, float* t_hit);
// k_accumulate / k_accumulate4: adds the pass's samples (ta.n_samples of them in `slab`) to rgb_sum in sample order
hipError_t launch_accumulate(hipStream_t st, const float* slab, const TraceArgs& ta, int first_pass, float* rgb_sum);
hipError_t launch_resolve(hipStream_t st, const float* sum, size_t n, float spp, uint8_t* out);
hipError_t launch_probe(const SceneData& scene, uint32_t which, uint32_t arg, uint32_t W, uint32_t H, const uint32_t* in, size_t n, float* out);
hipError_t launch_selftest_rcp(unsigned long long* mismatches);
// the multi-GPU gather on the first device: a landed frame added, a device's tiles packed / unpacked (32x32-pixel tiles, tile t of the
// frame belongs to device t % tile_count)
hipError_t launch_add_frame(hipStream_t st, float* into, const float* from, size_t n);
hipError_t launch_pack_tiles(hipStream_t st, const float* frame, float* packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tile_count, uint32_t tile_index);
hipError_t launch_unpack_tiles(hipStream_t st, float* frame, const float* packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tile_count, uint32_t tile_index);

// ---- deep trees (rrt_stream.hip) ------------------------------------------------------------------------------------------------
// k_trace_stream<mode, features>: mode RRT_AABB_TIGHT / RRT_AABB_REFERENCE / kAabbClean; `features` = the F_* bits the scene needs (three variants:
// triangles and spheres with constant textures, everything but nested media, everything).  The grid is persistent: the wave items are handed out through item_counter[0..7] (one counter per XCD's
// band of the frame), which must be zero when the kernel starts.  ta.stack_depth = the stack entries per lane kept in LDS.
hipError_t launch_trace_stream(int mode, uint32_t features, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& ta, float* slab,
                               unsigned long long* counters, uint32_t* item_counter);
// k_intersect_stream<mode>: a run of `run` rays per wave, lds_entries stack entries per lane in LDS
hipError_t launch_intersect_stream(int mode, dim3 grid, hipStream_t st, const SceneData& scene, const float* o, const float* d, size_t n, float t_min,
                                   float t_max, uint32_t seed, uint32_t lds_entries, uint32_t run, uint32_t coop_lanes, int* prim, float* t_hit);

// The current device's copy of the built-in staging area of rrt_render_device_noalloc (rrt_noalloc_area.hip); returns an RRT_* code.
int noalloc_area(int device, float** out);

#ifdef RRT_PHASE_PROFILE
// profiling build only: reads and clears the counters of the deep-tree kernels
hipError_t profile_take_stream(uint64_t* out8);
#endif

}  // namespace rrt
