// Measurement probes of the profiling build (-DRRT_PHASE_PROFILE; tools/perf_pool_phase.py, tools/perf_phase.py).
// The kernels hold one probe object each and call its hooks; in the product build every probe is an empty struct whose hooks
// are empty inline functions, so the kernels carry no measurement code and no #ifdef.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rrt {

#ifdef RRT_PHASE_PROFILE
// One copy per translation unit of kernels; rrt_profile_read / rrt_profile_read_pool (rrt_kernels_shallow.hip) add the copies up.
static __device__ unsigned long long g_prof[8];       // stream kernels: node visits, pair tests, node steps, leaf steps, cycles shade / walk / leaf, shade phases
                                                      // sifting query ([4..7], shallow-tree kernels): filter steps, items, dense groups, phases
static __device__ unsigned long long g_pool_prof[15]; // k_trace_pool, per kind of work: steps | paths served | cycles
static __device__ unsigned long long g_sift_hist[20]; // sifting query: flushes by list size in sixteenths of 256 ([0..15], last = more), cycles sifting [16] and in dense groups [17], lanes with a ray [18]

struct SiftProbe {
    unsigned long long filter = 0, items = 0, groups = 0, phases = 0, cyc_sift = 0, cyc_dense = 0, owners = 0;
    unsigned long long tick = 0;
    uint32_t hist[16] = {};
    __device__ void pair_tested() { filter += 1; }
    __device__ void sift_begins(uint32_t n_owners) {
        tick = __builtin_readcyclecounter();
        owners += n_owners;
    }
    __device__ void flushed(uint32_t n_items) {
        const unsigned long long now = __builtin_readcyclecounter();
        cyc_sift += now - tick;
        tick = now;
        items += n_items;
        groups += (n_items + 63u) / 64u;
        const uint32_t b = n_items / 16u < 15u ? n_items / 16u : 15u;
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
            if (b == i) hist[i] += 1;
    }
    __device__ void dense_done() {
        const unsigned long long now = __builtin_readcyclecounter();
        cyc_dense += now - tick;
        tick = now;
    }
    __device__ void phase_done() { phases += 1; }
    __device__ void store(uint32_t lane) {
        if (lane != 0) return;  // wave-level counts
        atomicAdd(&g_prof[4], filter);
        atomicAdd(&g_prof[5], items);
        atomicAdd(&g_prof[6], groups);
        atomicAdd(&g_prof[7], phases);
        for (uint32_t i = 0; i < 16; ++i) atomicAdd(&g_sift_hist[i], (unsigned long long)hist[i]);
        atomicAdd(&g_sift_hist[16], cyc_sift);
        atomicAdd(&g_sift_hist[17], cyc_dense);
        atomicAdd(&g_sift_hist[18], owners);
    }
};

template <uint32_t KINDS>
struct PoolProbe {
    unsigned long long steps[KINDS] = {}, lanes[KINDS] = {}, cycles[KINDS] = {};
    unsigned long long tick = __builtin_readcyclecounter();
    uint32_t phase = KINDS;
    __device__ void turn() {  // top of the wave's loop: the time since the last turn belongs to the kind that ran
        const unsigned long long now = __builtin_readcyclecounter();
#pragma unroll
        for (uint32_t c = 0; c < KINDS; ++c)
            if (phase == c) cycles[c] += now - tick;
        tick = now;
    }
    __device__ void chosen(uint32_t kind, uint32_t n) {
        phase = kind;
#pragma unroll
        for (uint32_t c = 0; c < KINDS; ++c)
            if (kind == c) {
                steps[c] += 1;
                lanes[c] += n;
            }
    }
    __device__ void store(uint32_t lane) {
        if (lane != 0) return;
        for (uint32_t c = 0; c < KINDS; ++c) {
            atomicAdd(&g_pool_prof[c], steps[c]);
            atomicAdd(&g_pool_prof[KINDS + c], lanes[c]);
            atomicAdd(&g_pool_prof[2 * KINDS + c], cycles[c]);
        }
    }
};

struct StreamProbe {
    unsigned long long nodes = 0, pairs = 0, steps = 0, leaf_steps = 0, cyc_shade = 0, cyc_walk = 0, cyc_leaf = 0, n_shade = 0;
    unsigned long long tick = __builtin_readcyclecounter();
    __device__ unsigned long long lap() {
        const unsigned long long now = __builtin_readcyclecounter(), d = now - tick;
        tick = now;
        return d;
    }
    __device__ void shaded() {
        cyc_shade += lap();
        n_shade += 1;
    }
    __device__ void node_visited() { nodes += 1; }
    __device__ void leaf_reached(uint32_t n_pairs) { pairs += n_pairs; }
    __device__ void walked(uint32_t lane) {
        if (lane == 0) ++steps;
        cyc_walk += lap();
    }
    __device__ void leaf_phase_begins(uint32_t lane) {
        if (lane == 0) ++leaf_steps;
    }
    __device__ void leaf_phase_done() { cyc_leaf += lap(); }
    __device__ void store(uint32_t lane) {
        if (lane == 0) {
            atomicAdd(&g_prof[4], cyc_shade);
            atomicAdd(&g_prof[5], cyc_walk);
            atomicAdd(&g_prof[6], cyc_leaf);
            atomicAdd(&g_prof[7], n_shade);
        }
        atomicAdd(&g_prof[0], nodes);  // per lane
        atomicAdd(&g_prof[1], pairs);
        atomicAdd(&g_prof[2], steps);
        atomicAdd(&g_prof[3], leaf_steps);
    }
};
#else
struct SiftProbe {
    __device__ void pair_tested() {}
    __device__ void sift_begins(uint32_t) {}
    __device__ void dense_done() {}
    __device__ void flushed(uint32_t) {}
    __device__ void phase_done() {}
    __device__ void store(uint32_t) {}
};
template <uint32_t KINDS>
struct PoolProbe {
    __device__ void turn() {}
    __device__ void chosen(uint32_t, uint32_t) {}
    __device__ void store(uint32_t) {}
};
struct StreamProbe {
    __device__ void shaded() {}
    __device__ void node_visited() {}
    __device__ void leaf_reached(uint32_t) {}
    __device__ void walked(uint32_t) {}
    __device__ void leaf_phase_begins(uint32_t) {}
    __device__ void leaf_phase_done() {}
    __device__ void store(uint32_t) {}
};
#endif

}  // namespace rrt
