// Kernels and the device half of the C ABI (include/rrt.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rrt_device.hpp"
#include "rrt_error.hpp"
#include "rrt_launch.hpp"

using namespace rrt;

namespace {

constexpr size_t kStaticSlabBytes = RRT_NOALLOC_STAGING_BYTES;  // the staging area of rrt_render_device_noalloc (rrt_noalloc_area.hip)


// ---- the trace kernel for shallow trees (at most kDeepTreeNodes nodes; deep trees: k_trace_stream) ------------------------
// ≙ main.rs:200-257 (thread_work) + aggregate.rs:16-105 (trace), for one pass of `n_samples` samples per pixel.
//
// A wave owns an 8x8-pixel tile and a chunk of up to 64 consecutive samples of each of its pixels: a pool of
// 64 x chunk camera samples.  Lanes draw samples from that pool one at a time (ballot + prefix count, no memory
// traffic): when a lane's path ends it stores the path's radiance and immediately starts the next undrawn sample, so
// lanes never wait for a neighbour's longer path, and the dispatcher balances the (tile, chunk) items across the chip.
// Path state (ray, throughput, RNG) lives in registers for the whole path; the BVH short stack is in LDS.
// Every sample's radiance goes to slab[sample][y][x] (12 bytes per path); k_accumulate then adds a pixel's samples in
// sample order, which makes the image independent of how samples were scheduled.
template <int AABB_MODE, uint32_t FEAT, bool HARVEST = false>
__global__ __launch_bounds__(kTraceBlock, RRT_TRACE_MIN_WAVES) void k_trace(DScene sc, TraceArgs ta, float* __restrict__ slab, unsigned long long* __restrict__ counters) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x;
    const uint32_t chunk = blockIdx.x / ta.n_t8, j = blockIdx.x % ta.n_t8;
    const uint32_t tile32 = ta.tile_index + (j >> 4) * ta.tile_count;
    const uint32_t x0 = (tile32 % ta.tiles_x) * 32u + ((j & 3u) << 3);
    const uint32_t y0 = (tile32 / ta.tiles_x) * 32u + (((j >> 2) & 3u) << 3);
    const uint32_t s0 = chunk * ta.chunk;
    const uint32_t ns = ta.n_samples - s0 < ta.chunk ? ta.n_samples - s0 : ta.chunk;
    const uint32_t pool = ns * 64u;
    uint32_t* stack = lds_stack + lane;

    uint32_t next = 64u;  // next undrawn pool index (wave-uniform)
    uint32_t idx = lane;
    bool alive = idx < pool;
    bool fresh = true;
    bool valid = false;
    unsigned long long rays = 0, paths = 0;
    uint32_t px = 0, py = 0, smp = 0, depth = 0;
    Ray r;
    f3 thr;
    Rng g;
    r.o = r.d = thr = mk3(0.0f, 0.0f, 0.0f);
    g.s0 = g.s1 = 0;
    PendingPdf pend;  // HARVEST: the last diffuse bounce's throughput, waiting for the light densities of the ray it scattered
    pend.active = false;
    pend.numerator = mk3(0.0f, 0.0f, 0.0f);
    pend.scatter_pdf = 0.0f;
    // The closest-hit query is a whole-wave operation, so the loop runs until every lane's pool share is done and lanes
    // without a path keep taking part (valid = false).
    while (RRT_BALLOT(alive) != 0ull) {
        if (fresh && alive) {
            px = x0 + (idx & 7u);
            py = y0 + ((idx >> 3) & 7u);
            smp = s0 + (idx >> 6);
            valid = px < ta.W && py < ta.H;
            if (valid) {
                g = rng_seed(ta.seed, py * ta.W + px, ta.first_sample + smp);
                r = camera_ray(sc.cam, px, py, ta.W, ta.H, g);
                thr = mk3(1.0f, 1.0f, 1.0f);
                depth = 0;
                pend.active = false;
                ++paths;
            }
            fresh = false;
        }
        bool done = true;
        {
            Hit hit;
            unsigned char* area = (unsigned char*)(lds_stack + kTraceBlock * ta.stack_depth);
            float harvest[kHarvestLights];
            hit = closest_hit_sift<AABB_MODE, FEAT, HARVEST, kSiftCap, true, false>(sc, valid, r, RRT_T_MIN, RRT_T_MAX, g, depth, stack, kTraceBlock, area, harvest);
            if (valid) {
                ++rays;
                f3 radiance;
                if (HARVEST && pend.active) {  // aggregate.rs:79-89 for the previous bounce, now that this ray's light distances are known
                    const float pdf_val = mixture_value_harvested(sc, r, harvest) * 0.5f + pend.scatter_pdf * 0.5f;  // pair_value, pdf.rs:117-119
                    thr = pend.numerator / pdf_val;
                    pend.active = false;
                }
                done = shade<FEAT, HARVEST>(sc, hit, r, thr, g, depth, ta.max_depth, ta.bg, &radiance, &pend);
                if (done) {
                    f3 c = thr * radiance;
                    float* q = slab + (((size_t)smp * ta.H + py) * ta.W + px) * 3;
                    q[0] = c.x;
                    q[1] = c.y;
                    q[2] = c.z;
                } else {
                    ++depth;
                }
            }
        }
        const unsigned long long finished = RRT_BALLOT(done);
        if (done) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(finished >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)finished, 0u));
            idx = next + rank;
            alive = idx < pool;
            fresh = true;
            if (!alive) valid = false;
        }
        next += (uint32_t)__popcll(finished);
    }
    if (counters) {
        for (int off = 32; off > 0; off >>= 1) {
            rays += __shfl_down(rays, off);
            paths += __shfl_down(paths, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], rays);
            atomicAdd(&counters[1], paths);
        }
    }
}


// ---- the trace kernel for shallow trees, pooled ---------------------------------------------------------------------------
// Same paths, same arithmetic, same results as k_trace; what differs is who runs when.  In k_trace each lane owns one path
// and the wave steps query - shade - query in lockstep, so every branch of the shading code (light hit, glass, Lambert towards a
// triangle light, Lambert cosine lobe, fresh camera ray ...) runs in every turn of the loop for the handful of lanes that need it:
// 0.455 of the lanes of an average vector instruction do useful work.  Here a wave keeps a POOL of kPool paths in LDS (15 dwords
// each), every path tagged with what it waits for, and each turn of the loop picks ONE kind of work that has 64 paths waiting,
// pulls those 64 into registers, runs that one piece of code at full width and files the paths under what they wait for next:
//   FREE   a slot without a path            -> draw a sample, make the camera ray                     -> TRACE
//   TRACE  has a ray                        -> closest hit (sifting query), finish the last bounce's density, look up the
//                                              material; a path that missed, hit a light or reached the depth limit stores
//                                              its sample; for Lambert make the light-or-lobe draws  -> FREE / GEN / LTRI / LCS
//   GEN    glass, metal                     -> shade                                                  -> TRACE
//   LTRI   Lambert, towards a triangle light-> shade                                                  -> TRACE
//   LCS    Lambert, cosine lobe or sphere light (one sincos / basis block, see sample_scatter_direction) -> TRACE
// The draws of a path are made in the reference's order whatever the schedule (the light-or-lobe draws are the first of their
// bounce either way), every sample still goes to slab[sample][y][x], and k_accumulate adds in sample order: bit-identical frames.
// Waves are persistent: the grid fills the chip once and a wave that has drawn all samples of its (tile, chunk) item takes the
// next item from an atomic counter, so a launch has no ramp of short-lived blocks and its tail is one item long.
#ifndef RRT_POOL
#define RRT_POOL 128
#endif
#ifndef RRT_POOL_WAVES
#define RRT_POOL_WAVES 4
#endif
constexpr uint32_t kPool = RRT_POOL;    // paths per wave: 15 x 128 x 4 B = 7.5 KB; with the query's work area 9.5 KB: four waves per SIMD
#ifndef RRT_POOL_SIFT_CAP
#define RRT_POOL_SIFT_CAP 320
#endif
constexpr uint32_t kPoolSiftCap = RRT_POOL_SIFT_CAP;  // candidate list of the sifting query: flushed early beyond 256 candidates (a step lists about 110)
constexpr uint32_t kPoolWaves = RRT_POOL_WAVES;  // waves per SIMD the LDS request leaves room for
// P_PDF_T: the cosine lobe's density of a pending bounce on the way from shading to the query, the hit distance on the way back
// P_HIT: 0 for a miss, else bit 0 = negative determinant, bits 1..25 slot + 1, bits 26..31 the light the light-or-lobe draw chose
enum : uint32_t { P_OX, P_OY, P_OZ, P_DX, P_DY, P_DZ, P_TX, P_TY, P_TZ, P_G0, P_G1, P_PIX, P_META, P_PDF_T, P_HIT, kPoolFields };
constexpr uint32_t kPoolMaxLights = 64;
enum : uint32_t { C_FREE = 0, C_TRACE = 1, C_GEN = 2, C_LTRI = 3, C_LCS = 4, kClasses = 5 };
// P_META: bits 0..2 class, 3 a diffuse bounce's density is pending, 4 the light-or-lobe draw chose a light, 5..12 depth, 13..31 sample
constexpr uint32_t kPoolMaxDepth = 255, kPoolMaxSamples = 1u << 19, kPoolMaxSlots = (1u << 25) - 1u;
constexpr size_t pool_bytes_per_wave(bool harvest) { return sift_bytes(kPoolSiftCap) + (harvest ? kHarvestBytesPerWave : 0u) + (size_t)kPoolFields * kPool * 4; }

// The kernel's argument block as it lies in memory.  The wave's loop runs for the whole launch, and an argument the compiler has
// in a scalar register at its top stays there to its end - 76 of the 102 scalar registers, were all of them held, so most were
// parked in lanes of a vector register and fetched back (one vector instruction each) in the middle of the hot steps.  What only the
// sample-drawing step and the kernel's last lines need (the camera, the frame's and the pass's numbers, the counters) is therefore
// read from the argument block where it is used, through a pointer the compiler cannot see through.
struct PoolKernArgs {
    DScene sc;
    TraceArgs ta;
    float* slab;
    unsigned long long* counters;
    uint32_t* item_counter;
};
typedef const PoolKernArgs RRT_CONST_AS* pool_args_p;
template <class T>
RRT_DEV T load_words(const T RRT_CONST_AS* p) {  // a struct of 32-bit words out of the constant address space (scalar loads)
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T v;
    const uint32_t RRT_CONST_AS* q = (const uint32_t RRT_CONST_AS*)p;
    uint32_t w[sizeof(T) / 4];
    for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = q[i];
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
RRT_DEV pool_args_p pool_args_now() {
    pool_args_p p = (pool_args_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // (loads through p stay where they are written: not hoisted out of the loop, not held across it)
    return p;
}

template <int AABB_MODE, uint32_t FEAT, bool HARVEST>
__global__ __launch_bounds__(kTraceBlock, kPoolWaves) void k_trace_pool(DScene sc, TraceArgs ta, float* __restrict__ slab, unsigned long long* __restrict__,
                                                                        uint32_t* __restrict__) {
    extern __shared__ uint32_t lds_stack[];
    constexpr bool kPlainRecord = (FEAT & (F_TEX | F_BUMP | F_MEDIUM)) == 0;  // one shading record per slot: see DScene::plain
    const uint32_t lane = threadIdx.x;
    uint32_t* stack = lds_stack + lane;
    unsigned char* area = (unsigned char*)(lds_stack + kTraceBlock * ta.stack_depth);
    uint32_t* pool = (uint32_t*)(area + sift_bytes(kPoolSiftCap) + (HARVEST ? kHarvestBytesPerWave : 0u));
    uint32_t* list = (uint32_t*)(area + 64 * 8);  // the step's slot list lives where the query will keep its candidate list: read before that
#define POOL(f, slot) pool[(f) * kPool + (slot)]
    for (uint32_t s = lane; s < kPool; s += 64u) POOL(P_META, s) = C_FREE;
    __threadfence_block();
    uint32_t cnt[kClasses] = {kPool, 0u, 0u, 0u, 0u};  // wave-uniform
    uint32_t x0 = 0, y0 = 0, s0 = 0, item_size = 0, next = 0;  // the item samples are drawn from, and the next undrawn one
    bool exhausted = false;                                    // no items left in the frame
    unsigned long long rays = 0, paths = 0;
    const size_t frame_pixels = (size_t)ta.W * ta.H;
    PoolProbe<kClasses> probe;
    for (;;) {
        probe.turn();
        // ---- which kind of work: a kind with a full wave of paths waiting (ending paths first: they free slots), else fresh
        // samples while slots are free, else the fullest kind
        uint32_t phase = kClasses;
        const bool can_draw = !exhausted && cnt[C_FREE] != 0u;
        if (cnt[C_LCS] >= 64u) phase = C_LCS;
        else if (cnt[C_LTRI] >= 64u) phase = C_LTRI;
        else if (cnt[C_GEN] >= 64u) phase = C_GEN;
        else if (can_draw && cnt[C_FREE] >= 64u) phase = C_FREE;
        else if (cnt[C_TRACE] >= 64u) phase = C_TRACE;
        else {
            uint32_t most = can_draw ? cnt[C_FREE] : 0u;
            if (most != 0u) phase = C_FREE;
#pragma unroll
            for (uint32_t c = C_TRACE; c < kClasses; ++c)
                if (cnt[c] > most) {
                    most = cnt[c];
                    phase = c;
                }
            if (phase == kClasses) break;  // no path in flight and nothing left to draw
        }
        // ---- up to 64 slots of that kind, in slot order
        uint32_t n = 0;
#pragma unroll
        for (uint32_t base = 0; base < kPool; base += 64u) {
            const uint32_t s = base + lane;
            const bool is = s < kPool && (POOL(P_META, s) & 7u) == phase;
            const unsigned long long bal = RRT_BALLOT(is);
            const uint32_t rank = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (is && rank < 64u) list[rank] = s;
            n += (uint32_t)__popcll(bal);
        }
        if (n > 64u) n = 64u;
        probe.chosen(phase, n);
        __threadfence_block();
        bool have = lane < n;
        const uint32_t slot = have ? list[lane] : 0u;

        if (phase == C_FREE) {
            // ≙ main.rs:150-157, 227-229: the next samples of the wave's item (a new item when this one is used up)
            const pool_args_p cold = pool_args_now();
            if (next >= item_size) {
                const TraceArgs t = load_words(&cold->ta);
                uint32_t item = 0;
                if (lane == 0) item = atomicAdd(cold->item_counter, 1u);
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                if (item >= t.n_items) {
                    exhausted = true;
                    continue;
                }
                const uint32_t chunk = item / t.n_t8, j = item % t.n_t8;
                const uint32_t tile32 = t.tile_index + (j >> 4) * t.tile_count;
                x0 = (tile32 % t.tiles_x) * 32u + ((j & 3u) << 3);
                y0 = (tile32 / t.tiles_x) * 32u + (((j >> 2) & 3u) << 3);
                s0 = chunk * t.chunk;
                item_size = (t.n_samples - s0 < t.chunk ? t.n_samples - s0 : t.chunk) * 64u;
                next = 0;
            }
            const uint32_t take = n < item_size - next ? n : item_size - next;
            const uint32_t idx = next + lane;
            const uint32_t px = x0 + (idx & 7u), py = y0 + ((idx >> 3) & 7u), smp = s0 + (idx >> 6);
            const uint32_t W = cold->ta.W, H = cold->ta.H;
            const bool valid = lane < take && px < W && py < H;  // pixels beyond the frame's edge use up their samples, no slot
            if (valid) {
                const rrt_camera cam = load_words(&cold->sc.cam);
                Rng g = rng_seed(cold->ta.seed, py * W + px, cold->ta.first_sample + smp);
                const Ray r = camera_ray(cam, px, py, W, H, g);
                POOL(P_OX, slot) = __float_as_uint(r.o.x); POOL(P_OY, slot) = __float_as_uint(r.o.y); POOL(P_OZ, slot) = __float_as_uint(r.o.z);
                POOL(P_DX, slot) = __float_as_uint(r.d.x); POOL(P_DY, slot) = __float_as_uint(r.d.y); POOL(P_DZ, slot) = __float_as_uint(r.d.z);
                POOL(P_TX, slot) = __float_as_uint(1.0f); POOL(P_TY, slot) = __float_as_uint(1.0f); POOL(P_TZ, slot) = __float_as_uint(1.0f);
                POOL(P_G0, slot) = g.s0; POOL(P_G1, slot) = g.s1;
                POOL(P_PIX, slot) = py * W + px;
                POOL(P_META, slot) = C_TRACE | (smp << 13);
                ++paths;
            }
            next += take;
            const uint32_t made = (uint32_t)__popcll(RRT_BALLOT(valid));
            cnt[C_FREE] -= made;
            cnt[C_TRACE] += made;
            __threadfence_block();
            continue;
        }

        Ray r;
        r.o = r.d = mk3(0.0f, 0.0f, 0.0f);
        Rng g = {0u, 0u};
        uint32_t meta = 0;
        if (have) {
            r.o = mk3(__uint_as_float(POOL(P_OX, slot)), __uint_as_float(POOL(P_OY, slot)), __uint_as_float(POOL(P_OZ, slot)));
            r.d = mk3(__uint_as_float(POOL(P_DX, slot)), __uint_as_float(POOL(P_DY, slot)), __uint_as_float(POOL(P_DZ, slot)));
            g.s0 = POOL(P_G0, slot);
            g.s1 = POOL(P_G1, slot);
            meta = POOL(P_META, slot);
        }
        const uint32_t depth = (meta >> 5) & 255u;

        if (phase == C_TRACE) {
            // ≙ aggregate.rs:24: the closest hit; then whatever decides which shading code the path needs
            float harvest[kHarvestLights];
            const Hit hit = closest_hit_sift<AABB_MODE, FEAT, HARVEST, kPoolSiftCap, false, false>(sc, have, r, RRT_T_MIN, RRT_T_MAX, g, depth, stack, kTraceBlock, area, harvest);
            uint32_t cls = C_FREE;
            if (have) {
                ++rays;
                f3 thr = mk3(__uint_as_float(POOL(P_TX, slot)), __uint_as_float(POOL(P_TY, slot)), __uint_as_float(POOL(P_TZ, slot)));
                const bool pending = HARVEST && (meta & 8u) != 0u;
#ifdef RRT_EXP_NO_PENDING
                if (false) {
#else
                if (pending) {  // aggregate.rs:79-89 for the previous bounce, now that this ray's light distances are known (thr held the numerator)
#endif
                    const float scatter_pdf = __uint_as_float(POOL(P_PDF_T, slot));
                    const float pdf_val = mixture_value_harvested(sc, r, harvest) * 0.5f + scatter_pdf * 0.5f;  // pair_value, pdf.rs:117-119
                    thr = thr / pdf_val;
                }
                uint32_t towards = 0u, light = 0u;
                f3 radiance = mk3(0.0f, 0.0f, 0.0f);
                if (depth < ta.max_depth && hit.slot >= 0) {
                    uint32_t type;  // Shape::get_material's type
                    if (kPlainRecord) type = RRT_PLAIN_TYPE(__float_as_uint(sc.c_plain()[2 * hit.slot].w));
                    else type = sc.c_materials()[RRT_INFO_MATERIAL(__float_as_uint(sc.c_slots()[3 * hit.slot + 1].w))].type;
                    if (kPlainRecord && type == RRT_MAT_DIFFUSE_LIGHT) {  // material.rs:229-231: emission ends the path
                        radiance = xyz(sc.c_plain()[2 * hit.slot + 1]);  // (a textured light's emission needs its hit properties: GEN)
                    } else if (type == RRT_MAT_LAMBERT) {
                        const LightChoice ch = draw_light_choice(sc, g);  // pdf.rs:110, 122: the bounce's first draws
                        cls = C_LCS;
                        if (ch.towards_light) {
                            towards = 16u;
                            if (sc.c_lights()[ch.k].kind == RRT_KIND_TRIANGLE) cls = C_LTRI;
                        }
                        light = ch.k;
                        POOL(P_G0, slot) = g.s0;
                        POOL(P_G1, slot) = g.s1;
                    } else {
                        cls = C_GEN;
                    }
                } else {
                    radiance = background(ta.bg, r);  // aggregate.rs:104: a miss, or the depth limit
                }
                if (cls == C_FREE) {  // the path ends here and its sample is stored (main.rs:244-245): no shading step needed
                    const f3 c = thr * radiance;
                    float* q = slab + ((size_t)(meta >> 13) * frame_pixels + POOL(P_PIX, slot)) * 3;
                    q[0] = c.x;
                    q[1] = c.y;
                    q[2] = c.z;
                    POOL(P_META, slot) = C_FREE;
                } else {
                    if (pending) {
                        POOL(P_TX, slot) = __float_as_uint(thr.x); POOL(P_TY, slot) = __float_as_uint(thr.y); POOL(P_TZ, slot) = __float_as_uint(thr.z);
                    }
                    POOL(P_PDF_T, slot) = __float_as_uint(hit.t);
                    POOL(P_HIT, slot) = (light << 26) | (((uint32_t)hit.slot + 1u) << 1) | (hit.det < 0.0f ? 1u : 0u);
                    POOL(P_META, slot) = (meta & ~(7u | 8u | 16u)) | cls | towards;
                }
            }
            cnt[C_TRACE] -= n;
            cnt[C_FREE] += (uint32_t)__popcll(RRT_BALLOT(have && cls == C_FREE));
            cnt[C_GEN] += (uint32_t)__popcll(RRT_BALLOT(have && cls == C_GEN));
            cnt[C_LTRI] += (uint32_t)__popcll(RRT_BALLOT(have && cls == C_LTRI));
            cnt[C_LCS] += (uint32_t)__popcll(RRT_BALLOT(have && cls == C_LCS));
            __threadfence_block();
            continue;
        }

        // ---- one of the shading kinds: aggregate.rs:26-104 after the query
        bool done = false;
        if (have) {
            f3 thr = mk3(__uint_as_float(POOL(P_TX, slot)), __uint_as_float(POOL(P_TY, slot)), __uint_as_float(POOL(P_TZ, slot)));
            const uint32_t hw = POOL(P_HIT, slot);
            Hit hit;
            hit.slot = (int)((hw >> 1) & 0x01ffffffu) - 1;
            hit.t = __uint_as_float(POOL(P_PDF_T, slot));
            hit.u = hit.v = 0.0f;
            hit.det = (hw & 1u) ? -1.0f : 1.0f;  // only the sign is used (by the hit properties)
            if (FEAT & (F_TEX | F_BUMP)) {  // Triangle::get_hit_properties recomputes the barycentrics from the ray (shape.rs:384-400)
                const cv4p cslots = sc.c_slots();
                const v4f q0 = cslots[3 * hit.slot], q1 = cslots[3 * hit.slot + 1], q2 = cslots[3 * hit.slot + 2];
                if (RRT_INFO_KIND(__float_as_uint(q1.w)) == RRT_KIND_TRIANGLE) triangle_uvdet(xyz(q0), xyz(q1), xyz(q2), r, &hit.u, &hit.v, &hit.det);
            }
            const LightChoice chosen = {(meta & 16u) != 0u, hw >> 26};
            PendingPdf pend;
            pend.active = false;
            pend.numerator = mk3(0.0f, 0.0f, 0.0f);
            pend.scatter_pdf = 0.0f;
            f3 radiance;
            done = shade<FEAT, HARVEST>(sc, hit, r, thr, g, depth, ta.max_depth, ta.bg, &radiance, &pend, &chosen);
            if (done) {
                const f3 c = thr * radiance;
                float* q = slab + ((size_t)(meta >> 13) * frame_pixels + POOL(P_PIX, slot)) * 3;
                q[0] = c.x;
                q[1] = c.y;
                q[2] = c.z;
                POOL(P_META, slot) = C_FREE;
            } else {
                const bool pending = HARVEST && pend.active;
                const f3 keep = pending ? pend.numerator : thr;  // a pending bounce parks its numerator where the throughput lives
                POOL(P_OX, slot) = __float_as_uint(r.o.x); POOL(P_OY, slot) = __float_as_uint(r.o.y); POOL(P_OZ, slot) = __float_as_uint(r.o.z);
                POOL(P_DX, slot) = __float_as_uint(r.d.x); POOL(P_DY, slot) = __float_as_uint(r.d.y); POOL(P_DZ, slot) = __float_as_uint(r.d.z);
                POOL(P_TX, slot) = __float_as_uint(keep.x); POOL(P_TY, slot) = __float_as_uint(keep.y); POOL(P_TZ, slot) = __float_as_uint(keep.z);
                POOL(P_G0, slot) = g.s0; POOL(P_G1, slot) = g.s1;
                if (pending) POOL(P_PDF_T, slot) = __float_as_uint(pend.scatter_pdf);
                POOL(P_META, slot) = (meta & ~(7u | 8u | 16u | (255u << 5))) | C_TRACE | (pending ? 8u : 0u) | ((depth + 1u) << 5);
            }
        }
        const uint32_t ended = (uint32_t)__popcll(RRT_BALLOT(have && done));
#pragma unroll
        for (uint32_t c = C_GEN; c < kClasses; ++c)  // (constant indices: the counts stay in scalar registers)
            if (phase == c) cnt[c] -= n;
        cnt[C_FREE] += ended;
        cnt[C_TRACE] += n - ended;
        __threadfence_block();
    }
#undef POOL
    probe.store(lane);
    unsigned long long* const counters = pool_args_now()->counters;
    if (counters) {
        for (int off = 32; off > 0; off >>= 1) {
            rays += __shfl_down(rays, off);
            paths += __shfl_down(paths, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], rays);
            atomicAdd(&counters[1], paths);
        }
    }
}

// Adds each pixel's samples of one pass in sample order (main.rs:244-245: colors[p] = colors[p] + pixel_color).
// One thread per float of the frame; reads are coalesced across the frame for every sample.
__global__ void k_accumulate(const float* __restrict__ slab, uint32_t n_samples, size_t frame_floats, uint32_t W, uint32_t tiles_x, uint32_t tile_count,
                             uint32_t tile_index, int first_pass, float* __restrict__ rgb_sum) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame_floats) return;
    if (tile_count > 1) {
        size_t pixel = i / 3;
        uint32_t x = (uint32_t)(pixel % W), y = (uint32_t)(pixel / W);
        if (((y >> 5) * tiles_x + (x >> 5)) % tile_count != tile_index) return;  // another rank's pixel: leave untouched
    }
    float sum = first_pass ? 0.0f : rgb_sum[i];
    for (uint32_t s = 0; s < n_samples; ++s) sum = sum + slab[(size_t)s * frame_floats + i];
    rgb_sum[i] = sum;
}

// The same for whole frames whose float count is a multiple of four: one thread per FOUR consecutive floats, 16-byte loads (each
// float is still the sum of its own samples in sample order: the same bits, at 1.4 times the bandwidth of 4-byte loads).
__global__ void k_accumulate4(const v4f* __restrict__ slab, uint32_t n_samples, size_t frame_quads, int first_pass, v4f* __restrict__ rgb_sum) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame_quads) return;
    v4f sum = first_pass ? v4f{0.0f, 0.0f, 0.0f, 0.0f} : rgb_sum[i];
    for (uint32_t s = 0; s < n_samples; ++s) sum = sum + __builtin_nontemporal_load(&slab[(size_t)s * frame_quads + i]);  // (read once: no reason to keep it in a cache)
    rgb_sum[i] = sum;
}

// ≙ Aggregate::hit (aggregate.rs:118-124) for a batch of rays on a shallow tree, one lane per ray.
template <int AABB_MODE>
__global__ __launch_bounds__(kBlock) void k_intersect(DScene sc, const float* __restrict__ o, const float* __restrict__ d, size_t n, float t_min,
                                                      float t_max, uint32_t seed, uint32_t stack_depth, int* __restrict__ prim, float* __restrict__ t_hit) {
    extern __shared__ uint32_t lds_stack[];
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < n;
    Ray r;
    r.o = r.d = mk3(0.0f, 0.0f, 1.0f);
    if (active) {
        r.o = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        r.d = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    }
    Rng g = rng_seed(seed, (uint32_t)i, 0);
    unsigned char* area = (unsigned char*)(lds_stack + kBlock * stack_depth) + (threadIdx.x >> 6) * kSiftBytesPerWave;
    const Hit h = closest_hit_sift<AABB_MODE, F_EVERY>(sc, active, r, t_min, t_max, g, 0, lds_stack + threadIdx.x, kBlock, area);
    if (!active) return;
    prim[i] = h.slot >= 0 ? (int)__float_as_uint(sc.slots[3 * h.slot].w) : -1;
    t_hit[i] = h.t;
}

// ≙ main.rs:173-186
__global__ void k_resolve(const float* __restrict__ sum, size_t n, float spp, uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = __builtin_sqrtf(sum[i] / spp);
    float b = c * 255.99f;
    uint8_t v;
    if (b != b) v = 0;
    else if (b <= 0.0f) v = 0;
    else if (b >= 255.0f) v = 255;
    else v = (uint8_t)b;
    out[i] = v;
}

// Unit probes (see rrt_probe in rrt.h).
__global__ void k_probe(DScene sc, uint32_t which, uint32_t arg, uint32_t W, uint32_t H, const uint32_t* __restrict__ in, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) {
        uint32_t px = in[3 * i], py = in[3 * i + 1], sample = in[3 * i + 2];
        Rng g = rng_seed(arg, py * W + px, sample);
        Ray r = camera_ray(sc.cam, px, py, W, H, g);
        float* q = out + 6 * i;
        q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = r.d.x; q[4] = r.d.y; q[5] = r.d.z;
    } else if (which == 1) {
        const float* f = (const float*)(in + 8 * i);
        Ray r;
        r.o = mk3(f[0], f[1], f[2]);
        r.d = mk3(f[3], f[4], f[5]);
        int prim = (int)in[8 * i + 6];
        float t = f[7];
        float* q = out + 14 * i;
        int slot = -1;
        for (uint32_t s = 0; s < sc.n_slots && prim >= 0; ++s)
            if (__float_as_uint(sc.slots[3 * s].w) == (uint32_t)prim) slot = (int)s;
        if (slot < 0) {
            for (int k = 0; k < 14; ++k) q[k] = 0.0f;
            return;
        }
        uint32_t mat;
        Hit hit;
        hit.slot = slot;
        hit.t = t;
        hit.u = hit.v = hit.det = 0.0f;
        if (RRT_INFO_KIND(__float_as_uint(sc.slots[3 * slot + 1].w)) == RRT_KIND_TRIANGLE)
            triangle_uvdet(xyz(sc.slots[3 * slot]), xyz(sc.slots[3 * slot + 1]), xyz(sc.slots[3 * slot + 2]), r, &hit.u, &hit.v, &hit.det);
        HitProps h = slot_props<F_EVERY>(sc, hit, r, &mat);
        q[0] = h.p.x; q[1] = h.p.y; q[2] = h.p.z; q[3] = h.n.x; q[4] = h.n.y; q[5] = h.n.z; q[6] = h.u; q[7] = h.v;
        q[8] = h.pu.x; q[9] = h.pu.y; q[10] = h.pu.z; q[11] = h.pv.x; q[12] = h.pv.y; q[13] = h.pv.z;
    } else if (which == 2) {
        const float* f = (const float*)(in + 5 * i);
        f3 c = texture_value(sc, (int)arg, f[0], f[1], mk3(f[2], f[3], f[4]));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    } else if (which == 3) {
        const float* f = (const float*)(in + 2 * i);
        float* q = out + 5 * i;
        q[0] = det_sin(f[0]); q[1] = det_cos(f[0]); q[2] = det_asin(f[0]); q[3] = det_log(f[0]); q[4] = det_atan2(f[0], f[1]);
    } else {
        Rng g = rng_seed(arg, in[2 * i], in[2 * i + 1]);
        for (int k = 0; k < 8; ++k) out[8 * i + k] = rng_f32(g);
    }
}

// Exhaustive check of exact_rcp against IEEE division over all 2^32 bit patterns.
__global__ void k_selftest_rcp(unsigned long long* __restrict__ mismatches) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float ref = 1.0f / x;
        const float got = exact_rcp(x);
        if (__float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

thread_local double t_copy_seconds = 0.0;  // rrt_scene_create: the time its device allocations and copies have taken so far
template <typename T>
int upload(const T* src, size_t count, const void** dst, std::vector<void*>* owned) {
    *dst = nullptr;
    size_t bytes = count * sizeof(T);
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // keep every pointer valid (kernels may form addresses for empty arrays but never dereference them)
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    owned->push_back(p);
    if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    t_copy_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *dst = p;
    return RRT_OK;
}

}  // namespace

// What rrt_render_multi keeps per scene handle from call to call: the handle's frame, counters, packed tiles, stream and events on
// the handle's own device, and the buffer its share lands in on the first handle's device.
struct MultiState {
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
    DScene d{};
    MultiState multi;
    std::vector<void*> owned;
    float* slab = nullptr;  // per-sample radiance staging, allocated on first render
    uint32_t* d_items = nullptr;   // item counters of the persistent kernels (k_trace_pool: one word; k_trace_stream: eight, one per XCD band), zeroed on the stream before every launch
    uint32_t wave_places = 0;      // k_trace_pool: waves the device holds at kPoolWaves per SIMD (the persistent grid's size)
    bool pooling = true;           // shallow plain scenes take k_trace_pool (rrt_scene_set_pooling)
    size_t slab_bytes = 0;
    size_t slab_cap = kSlabCapBytes;
    uint32_t res_x = 0, res_y = 0, samples = 0, stack_depth = 1;
    uint32_t lds_stack = kLdsStackEntries;  // deep trees: stack entries per lane that k_trace_stream keeps in LDS
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

static int validate_desc(const rrt_scene_desc* d, uint32_t* tree_depth) {
#define BAD(msg) return rrt_set_error(RRT_ERR_INVALID, std::string("scene description: ") + msg)
    if (d->n_nodes && !d->nodes) BAD("nodes is NULL");
    if (d->n_slots && (!d->slots || !d->tri_shade)) BAD("slots/tri_shade is NULL");
    if (d->n_spheres && !d->spheres) BAD("spheres is NULL");
    if (d->n_media && !d->media) BAD("media is NULL");
    if (d->n_boundary_tris && (!d->boundary_tris || !d->boundary_tri_shade)) BAD("boundary_tris is NULL");
    if (d->n_materials && !d->materials) BAD("materials is NULL");
    if (d->n_textures && !d->textures) BAD("textures is NULL");
    if (d->texel_bytes && !d->texels) BAD("texels is NULL");
    if (d->n_lights && !d->lights) BAD("lights is NULL");
    if (d->n_slots && !d->n_nodes) BAD("slots without nodes");
    if (d->aggregate_kind > RRT_AGGREGATE_BVH_CLEAN) BAD("unknown aggregate kind");
    // Every index a kernel will follow is checked here, once, so that no launch can read out of bounds.
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const rrt_node& n = d->nodes[i];
        if (n.b & RRT_LEAF_FLAG) {
            uint64_t cnt = n.b & ~RRT_LEAF_FLAG;
            if ((uint64_t)n.a + cnt > d->n_slots) BAD("leaf slot range out of bounds");
        } else {
            if (n.b > 2) BAD("inner node cut axis out of range");
            if (i + 1 >= d->n_nodes || n.a <= i + 1 || n.a >= d->n_nodes) BAD("inner node child out of bounds");
        }
    }
    for (uint32_t i = 0; i < d->n_slots; ++i) {
        const rrt_slot& s = d->slots[i];
        uint32_t kind = RRT_INFO_KIND(s.info);
        if (RRT_INFO_MATERIAL(s.info) >= d->n_materials) BAD("slot material out of range");
        if (kind == RRT_KIND_SPHERE && s.aux >= d->n_spheres) BAD("slot sphere index out of range");
        if (kind == RRT_KIND_MEDIUM && s.aux >= d->n_media) BAD("slot medium index out of range");
        if (kind > RRT_KIND_MEDIUM) BAD("unknown slot kind");
    }
    for (uint32_t i = 0; i < d->n_media; ++i) {
        const rrt_medium& m = d->media[i];
        if (m.material >= d->n_materials) BAD("medium phase function out of range");
        if (m.boundary_kind == RRT_KIND_SPHERE) { if (m.boundary_index >= d->n_spheres) BAD("medium boundary sphere out of range"); }
        else if (m.boundary_kind == RRT_KIND_TRIANGLE) { if (m.boundary_index >= d->n_boundary_tris) BAD("medium boundary triangle out of range"); }
        else if (m.boundary_kind == RRT_KIND_MEDIUM) {  // an earlier record, so chains end; kernels follow RRT_MAX_MEDIUM_NESTING records
            if (m.boundary_index >= i) BAD("medium boundary medium must be an earlier record");
            uint32_t chain = 2;
            for (uint32_t k = m.boundary_index; d->media[k].boundary_kind == RRT_KIND_MEDIUM; k = d->media[k].boundary_index) ++chain;
            if (chain > RRT_MAX_MEDIUM_NESTING) BAD("media nested deeper than RRT_MAX_MEDIUM_NESTING");
        } else BAD("unknown medium boundary kind");
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rrt_material& m = d->materials[i];
        if (m.type > RRT_MAT_ISOTROPIC) BAD("unknown material type");
        if (m.type != RRT_MAT_DIELECTRIC && (m.albedo < 0 || (uint32_t)m.albedo >= d->n_textures)) BAD("material texture out of range");
        if (m.bump >= (int32_t)d->n_textures) BAD("material bump map out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rrt_texture& t = d->textures[i];
        if (t.type > RRT_TEX_TURBULENCE) BAD("unknown texture type");
        // children precede their parent, which also rules out cycles in the Checker chain the device loops over
        if (t.type == RRT_TEX_CHECKER && (t.u0 >= i || t.u1 >= i)) BAD("Checker child must precede its parent");
        if (t.type == RRT_TEX_IMAGE) {
            if (t.u0 == 0 || t.u1 == 0 || (t.u2 != 1 && t.u2 != 3)) BAD("Image texture dimensions");
            if ((uint64_t)t.u3 + (uint64_t)t.u0 * t.u1 * t.u2 > d->texel_bytes) BAD("Image texels out of bounds");
        }
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const rrt_light& l = d->lights[i];
        if (l.kind == RRT_KIND_SPHERE) { if (l.index >= d->n_spheres) BAD("light sphere out of range"); }
        else if (l.kind != RRT_KIND_TRIANGLE) BAD("unknown light kind");
    }
    // The traversal stack is sized from the tree itself, never from a number the caller supplies.  The walk also proves that the
    // links form a tree: every node is reached exactly once (a node with two parents would make this walk, and the bottom-up
    // box pass, exponential; an unreachable one has no place in a pre-order layout).
    {
        std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, inner ancestors)
        std::vector<bool> seen(d->n_nodes, false);
        uint32_t deepest = 0, reached = 0;
        if (d->n_nodes) todo.emplace_back(0u, 0u);
        while (!todo.empty()) {
            auto [i, depth] = todo.back();
            todo.pop_back();
            if (seen[i]) BAD("a node is the child of two nodes");
            seen[i] = true;
            ++reached;
            if (depth > deepest) deepest = depth;
            const rrt_node& n = d->nodes[i];
            if (!(n.b & RRT_LEAF_FLAG)) {
                todo.emplace_back(i + 1, depth + 1);
                todo.emplace_back(n.a, depth + 1);
            }
        }
        if (reached != d->n_nodes) BAD("a node is not reachable from the root");
        if (deepest + 1 > kMaxStackDepth) return rrt_set_error(RRT_ERR_LIMIT, "BVH needs more than 64 pending stack entries per ray");
        if (d->bvh_depth < deepest + 1) BAD("bvh_depth is smaller than the tree requires");
        *tree_depth = deepest + 1;
    }
#undef BAD
    return RRT_OK;
}

extern "C" {

// The reference's node boxes all contain the world origin (every union in new_bvh_helper starts from the point box at
// the origin, aggregate.rs:195, 423-428), so they overlap heavily and cull little.  This computes, bottom-up, the boxes
// the primitives actually occupy, padded by 1e-5 of the scene's largest coordinate.  The TIGHT traversal accepts a node
// only if the reference's test accepts the reference's box AND the ray overlaps this box: a node rejected by the second
// condition holds nothing the ray can hit, so the closest hit (and every later decision, which depends only on the
// closest distance so far) is the reference's.  Two float4 per node: (lo, 0), (hi, 0).
// Shallow trees (the sifting query's) get a much wider padding along the triangles' planes, and with it the grazing guard
// (grazes_a_plane in rrt_traversal.hpp): the wider the padding, the closer to a triangle's plane a ray may run before the guard has
// to take its shortcuts away.  Their boxes are few and large, so the width costs next to nothing there; on a mesh it would swallow
// the triangles.
#ifndef RRT_SHALLOW_PAD
#define RRT_SHALLOW_PAD 1.6e-2
#endif
static bool is_shallow(const rrt_scene_desc* d) { return d->aggregate_kind != RRT_AGGREGATE_BVH_CLEAN && d->n_nodes <= kDeepTreeNodes; }
// across: the padding of every box face (deep trees), and in shallow trees the padding of a triangle's box ACROSS the triangle's
// plane; along: shallow trees, the padding of a triangle's box along its plane.  The wide padding goes along the plane only: a wall's
// box stays a thin slab, which a ray that leaves the wall has left before its span begins.  It is a fraction of the radius of the
// sphere around the scene's triangles (not of the scene's extent: a small mesh on a huge ground sphere keeps boxes of its own size).
struct Pads { float across, along; };
// Centre and radius of a sphere around every triangle vertex of the scene (the leaves' triangles and those that bound media);
// false when the scene has no triangle.
static bool triangle_sphere(const rrt_scene_desc* d, double centre[3], double* radius) {
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool any = false;
    auto each_vertex = [&](auto&& f) {
        auto of = [&](const rrt_slot& t) {
            for (int k = 0; k < 3; ++k) {
                double v[3];
                for (int c = 0; c < 3; ++c) v[c] = (double)t.v0[c] + (k == 1 ? (double)t.e1[c] : (k == 2 ? (double)t.e2[c] : 0.0));
                f(v);
            }
        };
        for (uint32_t i = 0; i < d->n_slots; ++i)
            if (RRT_INFO_KIND(d->slots[i].info) == RRT_KIND_TRIANGLE) of(d->slots[i]);
        for (uint32_t i = 0; i < d->n_boundary_tris; ++i) of(d->boundary_tris[i]);
    };
    each_vertex([&](const double* v) {
        for (int c = 0; c < 3; ++c) {
            if (!any || v[c] < lo[c]) lo[c] = v[c];
            if (!any || v[c] > hi[c]) hi[c] = v[c];
        }
        any = true;
    });
    if (!any) return false;
    for (int c = 0; c < 3; ++c) centre[c] = (double)(float)(0.5 * lo[c] + 0.5 * hi[c]);  // (the device holds it as floats)
    double far2 = 0.0;
    each_vertex([&](const double* v) {
        double s = 0.0;
        for (int c = 0; c < 3; ++c) s += (v[c] - centre[c]) * (v[c] - centre[c]);
        far2 = std::fmax(far2, s);
    });
    *radius = std::sqrt(far2);
    return true;
}
static Pads box_pads(const rrt_scene_desc* d) {
    float scene_max = 0.0f;
    for (uint32_t i = 0; i < d->n_nodes; ++i)
        for (int c = 0; c < 3; ++c) scene_max = std::fmax(scene_max, std::fmax(std::fabs(d->nodes[i].bmin[c]), std::fabs(d->nodes[i].bmax[c])));
    for (int c = 0; c < 3; ++c) scene_max = std::fmax(scene_max, std::fabs(d->camera.origin[c]));  // ray origins: the camera or a surface
    if (!is_shallow(d)) return {1.0e-5f * scene_max + 1.0e-30f, 1.0e-5f * scene_max + 1.0e-30f};
    double centre[3], radius = 0.0;
    if (!triangle_sphere(d, centre, &radius)) radius = 0.0;
    return {2.0e-5f * scene_max + 1.0e-30f, (float)(RRT_SHALLOW_PAD * radius) + 2.0e-5f * scene_max + 1.0e-30f};
}
static void pair_box(const rrt_scene_desc* d, uint32_t first, uint32_t count, Pads pads, float* b);

// The table of the grazing guard (layout and meaning: grazes_a_plane in rrt_traversal.hpp): a sphere around all triangle vertices
// of the scene (the leaves' triangles and those that bound media), and one scaled normal per distinct plane direction.
static std::vector<float> guard_table(const rrt_scene_desc* d, Pads pads) {
    struct Plane { double n[3], scale; };
    std::vector<Plane> planes;
    std::vector<const rrt_slot*> tris;
    for (uint32_t i = 0; i < d->n_slots; ++i)
        if (RRT_INFO_KIND(d->slots[i].info) == RRT_KIND_TRIANGLE) tris.push_back(&d->slots[i]);
    for (uint32_t i = 0; i < d->n_boundary_tris; ++i) tris.push_back(&d->boundary_tris[i]);
    auto put_u = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    const float inf = std::numeric_limits<float>::infinity();
    if (tris.empty()) return {0.0f, 0.0f, 0.0f, 0.0f, inf, inf, inf, put_u(2u), inf, 0.0f, 0.0f, 0.0f, inf, inf, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // nothing to graze
    double centre_d[3], radius = 0.0;
    triangle_sphere(d, centre_d, &radius);
    const float centre[3] = {(float)centre_d[0], (float)centre_d[1], (float)centre_d[2]};
    const float rho = std::nextafter((float)radius, inf);
    // how far from that sphere's centre a ray of the render kernels can start: at the camera (camera.rs:54-56: within the lens) or
    // on a primitive
    double start2 = 0.0;
    for (int c = 0; c < 3; ++c) start2 += ((double)d->camera.origin[c] - centre_d[c]) * ((double)d->camera.origin[c] - centre_d[c]);
    double start = std::sqrt(start2) + 2.0 * std::fabs((double)d->camera.lens_radius);
    for (uint32_t i = 0; i < d->n_slots; ++i) {
        float b[8];
        pair_box(d, i, 1, Pads{0.0f, 0.0f}, b);
        for (int k = 0; k < 8; ++k) {
            const double p[3] = {(double)b[(k & 1) ? 4 : 0], (double)b[(k & 2) ? 5 : 1], (double)b[(k & 4) ? 6 : 2]};
            double s2 = 0.0;
            for (int c = 0; c < 3; ++c) s2 += (p[c] - centre_d[c]) * (p[c] - centre_d[c]);
            start = std::fmax(start, std::sqrt(s2));
        }
    }
    constexpr double kUnitRoundoff = 5.9604644775390625e-08;  // 2^-24
    double longest_reach = 3.0e38;
    // directions whose unit normals differ by less (sine of the angle) share an entry; the 1.05 below covers it: no guarded cone
    // is narrower than 27 u * rho / (0.95 pads.along), and the tolerance is a twenty-fifth of that at most
    const double merge = std::fmin(2.0e-6, 0.04 * 27.0 * kUnitRoundoff * (double)rho / (0.95 * (double)pads.along));
    const size_t window = tris.size() <= 4096 ? tris.size() : 64;  // (a huge List over a mesh: only recent entries are looked through)
    for (const rrt_slot* t : tris) {
        const double e1[3] = {t->e1[0], t->e1[1], t->e1[2]}, e2[3] = {t->e2[0], t->e2[1], t->e2[2]};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double a2 = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        double scale = 0.0;  // a triangle without area: an entry of zero length, which every ray "grazes" (no shortcuts in such a scene)
        if (a2 > 0.0 && std::isfinite(a2) && std::isfinite(l1 * l2)) {
            for (int c = 0; c < 3; ++c) n[c] /= a2;
            // C = (1 + 2 r) (14 + 6 L / reach) + 7 + 6 (1 + L / reach), with L / reach <= L / rho (every reach is at least rho) and <= 2
            const double ratio = std::fmax(l1 / l2, l2 / l1), rel = std::fmin(std::fmax(l1, l2) / (double)rho, 2.0);
            scale = (a2 / (l1 * l2)) / ((1.0 + 2.0 * ratio) * (14.0 + 6.0 * rel) + 13.0 + 6.0 * rel);
            longest_reach = std::fmin(longest_reach, 0.95 * (double)pads.across * (a2 / (l1 * l2)) / ((13.0 + 6.0 * rel) * 1.05 * kUnitRoundoff));
            // |det| <= |d| |e1| |e2| must stay below the reference's threshold (f32::EPSILON, shape.rs:346-352) for directions whose
            // squared length is not a normal float (|d| < 1.1e-19): the guard's arithmetic loses its precision there
            if (!(l1 * l2 < 1.0e12)) longest_reach = 0.0;
        } else {
            n[0] = n[1] = n[2] = 0.0;
            longest_reach = 0.0;
        }
        bool merged = false;
        for (size_t j = planes.size() > window ? planes.size() - window : 0; j < planes.size() && !merged; ++j) {
            Plane& q = planes[j];
            const double x[3] = {n[1] * q.n[2] - n[2] * q.n[1], n[2] * q.n[0] - n[0] * q.n[2], n[0] * q.n[1] - n[1] * q.n[0]};
            const bool both_flat = scale == 0.0 && q.scale == 0.0;
            if (both_flat || (scale != 0.0 && q.scale != 0.0 && std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) <= merge)) {
                q.scale = std::fmin(q.scale, scale);
                merged = true;
            }
        }
        if (!merged) planes.push_back({{n[0], n[1], n[2]}, scale});
    }
    const double unit = 0.95 * (double)pads.along / (1.05 * kUnitRoundoff);
    float axis[3] = {inf, inf, inf};
    std::vector<float> general;
    for (const Plane& q : planes) {
        int along = -1;  // perpendicular to an axis (to within the merging tolerance): that axis' entry
        for (int c = 0; c < 3 && q.scale != 0.0; ++c)
            if (std::sqrt(q.n[(c + 1) % 3] * q.n[(c + 1) % 3] + q.n[(c + 2) % 3] * q.n[(c + 2) % 3]) <= merge) along = c;
        if (along >= 0) {
            axis[along] = std::fmin(axis[along], (float)(q.scale * unit));
            continue;
        }
        for (int c = 0; c < 3; ++c) {
            const double g = q.n[c] * q.scale * unit;
            general.push_back((float)(std::fabs(g) < 3.0e38 ? g : 0.0));
        }
        general.push_back(0.0f);
    }
    while (general.size() < 8 || (general.size() / 4) % 2 != 0) {  // (the device takes two per turn, and the first two without asking)
        const float inert[4] = {inf, 0.0f, 0.0f, 0.0f};  // d . G is infinite, or not a number: no ray is close to it
        general.insert(general.end(), inert, inert + 4);
    }
    // The render kernels take ONE reach for all their rays when that costs little: when the farthest start is within four radii of
    // the triangles' sphere (the Cornell boxes).  A small mesh in a wide scene (a cube on a ground sphere of a hundred times its
    // size) would have nearly every ray "graze" with the far ground's reach; there every ray works out its own.
    const double render_reach = 1.001 * (start + (double)rho);
    const bool own_reach = !(render_reach <= 4.0 * (double)rho) || !(render_reach <= longest_reach);
    const float render_reach2 = std::nextafter((float)(render_reach * render_reach * 1.0001), inf);
    std::vector<float> out = {centre[0], centre[1], centre[2], rho, axis[0], axis[1], axis[2], put_u((uint32_t)(general.size() / 4)),
                              (float)longest_reach, render_reach2, put_u(own_reach ? 1u : 0u), 0.0f};
    for (size_t k = 0; k + 8 <= general.size(); k += 8) {  // two entries side by side: (x, x, y, y), (z, z, -, -)
        const float* a = &general[k];
        const float* b = &general[k + 4];
        const float two[8] = {a[0], b[0], a[1], b[1], a[2], b[2], 0.0f, 0.0f};
        out.insert(out.end(), two, two + 8);
    }
    return out;
}
// b = (lo.xyz, -, hi.xyz, -)
static void grow_point(float* b, const float* p) {
    for (int c = 0; c < 3; ++c) {
        b[c] = std::fmin(b[c], p[c]);
        b[4 + c] = std::fmax(b[4 + c], p[c]);
    }
}
static void grow_sphere(float* b, const rrt_sphere& sp) {
    // corners of the local cube [-|r|, |r|]^3 through local_to_world: their box contains the transformed sphere
    const float r = std::fabs(sp.radius);
    for (int k = 0; k < 8; ++k) {
        const float x = (k & 1) ? r : -r, y = (k & 2) ? r : -r, z = (k & 4) ? r : -r;
        float p[3];
        for (int c = 0; c < 3; ++c) p[c] = sp.l2w[4 * c] * x + sp.l2w[4 * c + 1] * y + sp.l2w[4 * c + 2] * z + sp.l2w[4 * c + 3];
        grow_point(b, p);
    }
}
// A triangle's box, padded by `along` in the triangle's plane and by `across` at right angles to it: on axis c that is
// along * sqrt(1 - n_c^2) + across * |n_c| for the unit normal n.
static void grow_triangle(float* b, const rrt_slot& t, Pads pads) {
    float v1[3], v2[3];
    for (int c = 0; c < 3; ++c) {
        v1[c] = t.v0[c] + t.e1[c];
        v2[c] = t.v0[c] + t.e2[c];
    }
    float tb[8];
    for (int c = 0; c < 3; ++c) {
        tb[c] = std::fmin(t.v0[c], std::fmin(v1[c], v2[c]));
        tb[4 + c] = std::fmax(t.v0[c], std::fmax(v1[c], v2[c]));
    }
    const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int c = 0; c < 3; ++c) {
        float pad = std::fmax(pads.along, pads.across);  // (a triangle without area has no plane)
        if (pads.along != pads.across && len > 0.0 && std::isfinite(len)) {
            const double nc = std::fmin(std::fabs(n[c] / len), 1.0);
            pad = std::nextafter((float)((double)pads.along * std::sqrt(1.0 - nc * nc) + (double)pads.across * nc), std::numeric_limits<float>::infinity());
        }
        const float at[3] = {tb[0 + c] - pad, tb[4 + c] + pad, 0.0f};
        b[c] = std::fmin(b[c], at[0]);
        b[4 + c] = std::fmax(b[4 + c], at[1]);
    }
}
// The padded box of `count` consecutive slots starting at `first`.
static void pair_box(const rrt_scene_desc* d, uint32_t first, uint32_t count, Pads pads, float* b) {
    const float inf = std::numeric_limits<float>::infinity();
    b[0] = b[1] = b[2] = inf;
    b[4] = b[5] = b[6] = -inf;
    b[3] = b[7] = 0.0f;
    for (uint32_t k = 0; k < count; ++k) {
        const rrt_slot& sl = d->slots[first + k];
        const uint32_t kind = RRT_INFO_KIND(sl.info);
        float pb[8] = {inf, inf, inf, 0.0f, -inf, -inf, -inf, 0.0f};
        bool padded = false;
        if (kind == RRT_KIND_TRIANGLE) {
            grow_triangle(pb, sl, pads);
            padded = true;
        } else if (kind == RRT_KIND_SPHERE) grow_sphere(pb, d->spheres[sl.aux]);
        else {
            uint32_t mi = sl.aux;  // every distance a medium reports lies between its boundary's two (volume.rs:79-101)
            while (d->media[mi].boundary_kind == RRT_KIND_MEDIUM) mi = d->media[mi].boundary_index;
            const rrt_medium& m = d->media[mi];
            if (m.boundary_kind == RRT_KIND_SPHERE) grow_sphere(pb, d->spheres[m.boundary_index]);
            else {
                grow_triangle(pb, d->boundary_tris[m.boundary_index], pads);
                padded = true;
            }
        }
        for (int c = 0; c < 3; ++c) {
            b[c] = std::fmin(b[c], pb[c] - (padded ? 0.0f : pads.across));
            b[4 + c] = std::fmax(b[4 + c], pb[4 + c] + (padded ? 0.0f : pads.across));
        }
    }
}
static std::vector<float> tight_boxes(const rrt_scene_desc* d) {
    const uint32_t n = d->n_nodes;
    std::vector<float> tb((size_t)n * 8, 0.0f);
    const Pads pad = box_pads(d);
    for (uint32_t i = n; i-- > 0;) {  // pre-order layout: children follow their parent
        const rrt_node& nd = d->nodes[i];
        float* b = &tb[(size_t)i * 8];
        if (nd.b & RRT_LEAF_FLAG) {
            pair_box(d, nd.a, nd.b & ~RRT_LEAF_FLAG, pad, b);
        } else {
            const float* l = &tb[(size_t)(i + 1) * 8];
            const float* r = &tb[(size_t)nd.a * 8];
            for (int c = 0; c < 3; ++c) {
                b[c] = std::fmin(l[c], r[c]);
                b[4 + c] = std::fmax(l[4 + c], r[4 + c]);
            }
        }
    }
    return tb;
}

static int scene_create(const rrt_scene_desc* desc, rrt_scene** out) {
    if (!desc || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_create: NULL argument");
    const auto started = std::chrono::steady_clock::now();
    uint32_t tree_depth = 1;
    int rc = validate_desc(desc, &tree_depth);
    if (rc) return rc;
    // owns the handle (and through rrt_scene_destroy its device buffers) until the last step has succeeded
    struct Destroy { void operator()(rrt_scene* p) const { rrt_scene_destroy(p); } };
    std::unique_ptr<rrt_scene, Destroy> owner(new rrt_scene());
    rrt_scene* s = owner.get();
    // Where a process's first handle spends its time: the first call that needs the device (a process that has not used the GPU yet
    // creates its context here), then the first access to something in this library's code object (the runtime loads the code object
    // onto the device at that moment: every kernel variant of this translation unit and the 128 MiB staging area).
    const auto t_validated = std::chrono::steady_clock::now();
    HIP_TRY(hipGetDevice(&s->device));
    HIP_TRY(hipFree(nullptr));
    const auto t_context = std::chrono::steady_clock::now();
    {
        hipFuncAttributes attr;  // (asks the runtime about one of this file's kernels: that loads the file's code object)
        HIP_TRY(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_resolve)));
    }
    const auto t_code = std::chrono::steady_clock::now();
    t_copy_seconds = 0.0;
    uint32_t clean_depth = 0;  // clean trees: most stack entries a walk can have pending (sizes the traversal stack)
#define UP(field, src, count)                                                        \
    do {                                                                             \
        rc = upload(src, count, (const void**)&s->d.field, &s->owned);               \
        if (rc) return rc;                                                           \
    } while (0)
    // Device copy of the tree: leaves reference pair records (two primitives interleaved component-wise plus their box,
    // 128 bytes).  Which primitives share a record is free (every comparison on the device orders hits by distance, then
    // slot), so each leaf's triangles are matched greedily into the pairs whose common box is smallest: the sifting
    // traversal tests a pair only when the ray overlaps that box.  Spheres and media get records of their own.
    {
        std::vector<rrt_node> dn(desc->nodes, desc->nodes + desc->n_nodes);
        std::vector<float> pairs;
        const Pads pad = box_pads(desc);
        auto put_u = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
        // Harvesting (see HARVEST in rrt_device.hpp) needs a single leaf and every light to be one of its primitives:
        // a triangle light is the slot with the same vertex data and cull flag, a sphere light the slot of that sphere.
        std::vector<uint32_t> light_of_slot(desc->n_slots, 0u);  // light number + 1
        bool harvest = desc->n_nodes == 1 && desc->n_lights >= 1 && desc->n_lights <= kHarvestLights && desc->aggregate_kind != RRT_AGGREGATE_BVH_CLEAN;
        for (uint32_t i = 0; harvest && i < desc->n_lights; ++i) {
            const rrt_light& l = desc->lights[i];
            bool matched = false;
            for (uint32_t j = 0; j < desc->n_slots && !matched; ++j) {
                const rrt_slot& sl = desc->slots[j];
                if (light_of_slot[j] != 0u || RRT_INFO_KIND(sl.info) != l.kind) continue;
                if (l.kind == RRT_KIND_SPHERE) matched = sl.aux == l.index;
                else
                    matched = std::memcmp(sl.v0, l.v0, 12) == 0 && std::memcmp(sl.e1, l.e1, 12) == 0 && std::memcmp(sl.e2, l.e2, 12) == 0 &&
                              ((sl.info & RRT_INFO_CULL) != 0) == (l.cull != 0);
                if (matched) light_of_slot[j] = i + 1u;
            }
            harvest = matched;
        }
        if (!harvest) std::fill(light_of_slot.begin(), light_of_slot.end(), 0u);
        s->harvest = harvest;
        const bool clean = desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN;
        auto emit = [&](uint32_t ia, uint32_t ib, bool has_b) {  // ia < ib: A precedes B in leaf order
            if (clean && has_b && desc->slots[ib].prim < desc->slots[ia].prim) std::swap(ia, ib);  // clean trees: A first in SCENE order
            const rrt_slot& A = desc->slots[ia];
            const rrt_slot& B = desc->slots[has_b ? ib : ia];
            const float* av[3] = {A.v0, A.e1, A.e2};
            const float* bv[3] = {B.v0, B.e1, B.e2};
            for (int v = 0; v < 3; ++v)
                for (int c = 0; c < 3; ++c) {
                    pairs.push_back(av[v][c]);
                    pairs.push_back(bv[v][c]);
                }
            pairs.push_back(put_u((A.info & ~(15u << 24)) | (light_of_slot[ia] << 24)));
            pairs.push_back(put_u(has_b ? ((B.info & ~(15u << 24)) | (light_of_slot[ib] << 24)) : (RRT_KIND_NONE << 30)));
            pairs.push_back(put_u(ia));
            pairs.push_back(put_u(has_b ? ib : ia));
            pairs.push_back(put_u(A.prim));
            pairs.push_back(put_u(B.prim));
            float box[8], box_b[8];
            pair_box(desc, ia, 1, pad, box);
            if (has_b) {
                pair_box(desc, ib, 1, pad, box_b);
                for (int c = 0; c < 3; ++c) {
                    box[c] = std::fmin(box[c], box_b[c]);
                    box[4 + c] = std::fmax(box[4 + c], box_b[4 + c]);
                }
            }
            const bool lit = light_of_slot[ia] != 0u || (has_b && light_of_slot[ib] != 0u);
            const float lit_lanes = put_u(lit ? kPairLitMask : 0u);
            const float inter[8] = {box[0], box[4], box[1], box[5], box[2], box[6], lit_lanes, lit_lanes};  // (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, lane mask in two words)
            pairs.insert(pairs.end(), inter, inter + 8);
        };
        auto half_area = [](const float* b) {
            const float x = b[4] - b[0], y = b[5] - b[1], z = b[6] - b[2];
            return x * y + y * z + z * x;
        };
        std::vector<uint32_t> tri, partner;
        std::vector<float> boxes;
        struct Cand { float gain; uint32_t i, j; };
        std::vector<Cand> cands;
        for (rrt_node& n : dn) {
            if (!(n.b & RRT_LEAF_FLAG)) continue;
            const uint32_t first = n.a, count = n.b & ~RRT_LEAF_FLAG;
            const uint32_t first_pair = (uint32_t)(pairs.size() / (4 * kPairQuads));
            tri.clear();
            for (uint32_t k = 0; k < count; ++k)
                if (RRT_INFO_KIND(desc->slots[first + k].info) == RRT_KIND_TRIANGLE) tri.push_back(first + k);
            const uint32_t nt = (uint32_t)tri.size();
            // spheres and media get records of their own, at the front of the leaf; a leaf with exactly one of them (too few to fill
            // a dense group of the sifting query; with two, Materials.json renders 1.5 % slower) has it tested by the rays' own lanes
            // (closest_hit_sift), and says so in its node
            const bool shallow = is_shallow(desc);  // the sifting query's trees
            const bool direct = shallow && count - nt == 1u;
            for (uint32_t k = 0; k < count; ++k)
                if (RRT_INFO_KIND(desc->slots[first + k].info) != RRT_KIND_TRIANGLE) emit(first + k, first + k, false);
            boxes.resize((size_t)nt * 8);
            for (uint32_t i = 0; i < nt; ++i) pair_box(desc, tri[i], 1, pad, &boxes[(size_t)i * 8]);
            // gain of sharing a record = area of the two boxes minus the area of their union (a ray meets a box about in
            // proportion to its area); best gains first, each triangle used once
            cands.clear();
            const uint32_t window = nt <= 256 ? nt : 16;  // huge leaves (a List aggregate over a mesh): nearby slots only
            for (uint32_t i = 0; i < nt; ++i)
                for (uint32_t j = i + 1; j < nt && j - i <= window; ++j) {
                    float u[8];
                    for (int c = 0; c < 3; ++c) {
                        u[c] = std::fmin(boxes[(size_t)i * 8 + c], boxes[(size_t)j * 8 + c]);
                        u[4 + c] = std::fmax(boxes[(size_t)i * 8 + 4 + c], boxes[(size_t)j * 8 + 4 + c]);
                    }
                    const float gain = half_area(&boxes[(size_t)i * 8]) + half_area(&boxes[(size_t)j * 8]) - half_area(u);
                    cands.push_back({gain, i, j});
                }
            std::stable_sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.gain > y.gain; });
            partner.assign(nt, 0xffffffffu);
            float leaf_box[8];
            pair_box(desc, first, count, pad, leaf_box);
            const float min_gain = -0.1f * half_area(leaf_box);  // a slightly larger common box still beats a second record
            for (const Cand& c : cands) {
                if (c.gain < min_gain) break;
                if (partner[c.i] != 0xffffffffu || partner[c.j] != 0xffffffffu) continue;
                partner[c.i] = c.j;
                partner[c.j] = c.i;
            }
            for (uint32_t i = 0; i < nt; ++i) {
                if (partner[i] == 0xffffffffu) emit(tri[i], tri[i], false);
                else if (partner[i] > i) emit(tri[i], tri[partner[i]], true);
            }
            if (shallow && (((uint32_t)(pairs.size() / (4 * kPairQuads)) - first_pair - (direct ? 1u : 0u)) & 1u) != 0u) {
                // the sifting loop takes two records per turn: an odd leaf gets a record whose box no ray meets (its half-widths are
                // negative).  A ray the grazing guard has taken the box tests from lists it all the same: its two primitives are
                // triangles without area (all zeros, slot 0), which the exact test turns down at the determinant
                const size_t at = pairs.size();
                pairs.resize(at + 4 * kPairQuads, 0.0f);
                for (int c = 0; c < 3; ++c) {  // written as (lo, hi) = (1, -1): becomes half-width -1, centre 0 below
                    pairs[at + 24 + 2 * c] = 1.0f;
                    pairs[at + 24 + 2 * c + 1] = -1.0f;
                }
            }
            n.a = first_pair;
            n.b = RRT_LEAF_FLAG | ((uint32_t)(pairs.size() / (4 * kPairQuads)) - first_pair);
            if (direct) n.b |= kLeafDirect;  // (a shallow tree holds at most 2^25 pair records: the bit is free)
        }
        if (is_shallow(desc)) {
            // the sifting query (shallow trees) reads a pair's box as half-widths and centres: (h.x, c.x, h.y, c.y | h.z, c.z, ...);
            // a half-width is rounded up past the larger of the two distances, so the stored box contains the padded one
            for (size_t r = 0; r + 4 * kPairQuads <= pairs.size(); r += 4 * kPairQuads) {
                float* b = &pairs[r + 24];
                for (int c = 0; c < 3; ++c) {
                    const float lo = b[2 * c], hi = b[2 * c + 1];
                    const float mid = 0.5f * lo + 0.5f * hi;
                    const float half = std::nextafter(std::fmax(hi - mid, mid - lo), std::numeric_limits<float>::infinity());
                    b[2 * c] = lo > hi ? -1.0f : half;  // (a padding record's box: see kPadBoxBits)
                    b[2 * c + 1] = mid;
                }
            }
        }
        const std::vector<float> tb = tight_boxes(desc);
        if (clean) {
            // Inner nodes only, 64 bytes each, with up to FOUR children (see walk_step).  The children's primitive boxes are stored
            // as bytes on a grid of the node's own: dwords 0..2 the grid's origin (the low corner of the children's common box),
            // dword 3 the grid steps as three power-of-two exponents, dwords 4..9 the planes (lo.x of the four children, hi.x,
            // lo.y, hi.y, lo.z, hi.z; lows rounded down, highs up, so a stored box contains the exact one), dwords 10..13 the
            // links: an inner node's number, kLinkLeaf | (pairs - 1) << 28 | first pair, or kLinkNone for an unused place.
            // Half the bytes of float planes and, what counts for a walk whose lanes all sit in different nodes, four 16-byte
            // loads per visit instead of seven.  A node starts from the sweep's two children and replaces the child with the
            // largest box by that child's own two until it has four; a leaf of more than 8 pair records (shapes the sweep would
            // not separate) is split by halving its range.
            std::vector<float> rec;
            auto pair_range_box = [&](uint32_t first, uint32_t count, float* b) {  // pair boxes are stored (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z)
                const float inf = std::numeric_limits<float>::infinity();
                b[0] = b[1] = b[2] = inf;
                b[4] = b[5] = b[6] = -inf;
                for (uint32_t k = first; k < first + count; ++k) {
                    const float* pb = &pairs[(size_t)k * 4 * kPairQuads + 24];
                    for (int c = 0; c < 3; ++c) {
                        b[c] = std::fmin(b[c], pb[2 * c]);
                        b[4 + c] = std::fmax(b[4 + c], pb[2 * c + 1]);
                    }
                }
            };
            struct Child { uint32_t node; bool is_range; uint32_t first, count; float box[8]; };
            auto child_of_node = [&](uint32_t node) {
                const rrt_node& c = dn[node];
                const bool leaf = (c.b & RRT_LEAF_FLAG) != 0;
                Child ch = {node, leaf, leaf ? c.a : 0u, leaf ? (c.b & ~RRT_LEAF_FLAG) : 0u, {}};
                std::memcpy(ch.box, &tb[(size_t)node * 8], 32);
                return ch;
            };
            auto link_of = [&](const Child& c, uint32_t* out_link) -> bool {  // true when the child is a leaf small enough for a direct link
                if (!c.is_range || c.count > 8) return false;
                // a leaf without shapes (the description may hold one) gets the link of an unused place: nothing to fetch
                *out_link = c.count == 0u ? kLinkNone : (kLinkLeaf | ((c.count - 1u) << 28) | c.first);
                return true;
            };
            auto split = [&](const Child& c, Child* out) {
                if (c.is_range) {  // an oversized leaf: halve its pair range
                    const uint32_t half = c.count / 2;
                    out[0] = {0u, true, c.first, half, {}};
                    out[1] = {0u, true, c.first + half, c.count - half, {}};
                    pair_range_box(out[0].first, out[0].count, out[0].box);
                    pair_range_box(out[1].first, out[1].count, out[1].box);
                } else {
                    out[0] = child_of_node(c.node + 1u);
                    out[1] = child_of_node(dn[c.node].a);
                }
            };
            auto new_inner = [&]() {
                const size_t at = rec.size();
                rec.resize(at + 16, 0.0f);
                for (int k = 10; k < 14; ++k) rec[at + k] = put_u(kLinkNone);
                return (uint32_t)(at / 16);
            };
            if ((uint64_t)pairs.size() / (4 * kPairQuads) >= 0x0fffffffull)  // keeps every leaf link apart from kLinkNone
                return rrt_set_error(RRT_ERR_LIMIT, "clean tree: more than 2^28 - 2 pair records");
            struct Pending { Child c; uint32_t self, above; };  // above: stack entries that can be pending when the walk reaches this node
            std::vector<Pending> todo;
            uint32_t root_link = 0;
            if (!dn.empty()) {
                const Child root = child_of_node(0);
                uint32_t link;
                if (link_of(root, &link)) root_link = link;
                else {
                    root_link = new_inner();
                    todo.push_back({root, root_link, 0u});
                }
            }
            while (!todo.empty()) {
                const Pending it = todo.back();
                todo.pop_back();
                Child kids[5];
                uint32_t n_kids = 2;
                split(it.c, kids);
                while (n_kids < 4) {
                    int widest = -1;
                    uint32_t link;
                    for (uint32_t k = 0; k < n_kids; ++k)
                        if (!link_of(kids[k], &link) && (widest < 0 || half_area(kids[k].box) > half_area(kids[widest].box))) widest = (int)k;
                    if (widest < 0) break;
                    Child two[2];
                    split(kids[widest], two);
                    kids[widest] = two[0];
                    kids[n_kids++] = two[1];
                }
                const uint32_t pending_here = it.above + n_kids - 1u;  // the walk pushes every child it meets but the nearest
                if (pending_here > clean_depth) clean_depth = pending_here;
                // the node's grid: origin = low corner of the children's common box, step 2^e per axis with 255 steps spanning it
                float lo[3], hi[3];
                uint32_t expo[3], planes[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                bool empty[5];  // a leaf without shapes (the description may hold one): no box, the link of an unused place
                for (uint32_t k = 0; k < n_kids; ++k) empty[k] = kids[k].is_range && kids[k].count == 0u;
                for (int c = 0; c < 3; ++c) {
                    lo[c] = std::numeric_limits<float>::infinity();
                    hi[c] = -std::numeric_limits<float>::infinity();
                    for (uint32_t k = 0; k < n_kids; ++k) {
                        if (empty[k]) continue;
                        lo[c] = std::fmin(lo[c], kids[k].box[c]);
                        hi[c] = std::fmax(hi[c], kids[k].box[4 + c]);
                    }
                    if (!(lo[c] <= hi[c])) lo[c] = hi[c] = 0.0f;  // (only empty children)
                    if (!std::isfinite(lo[c]) || !std::isfinite(hi[c])) return rrt_set_error(RRT_ERR_INVALID, "clean tree: a primitive's box is not finite");
                    int e = 0;
                    const double span = (double)hi[c] - (double)lo[c];
                    if (!(span > 0.0)) e = -126;
                    else {
                        (void)std::frexp(span / 255.0, &e);  // span / 255 = m * 2^e with m in [0.5, 1): 2^e >= span / 255
                        if (e < -126) e = -126;
                        if (e > 127) return rrt_set_error(RRT_ERR_LIMIT, "clean tree: a node's box is too large for its byte grid");
                    }
                    expo[c] = (uint32_t)(e + 127);
                    const double step = std::ldexp(1.0, e);
                    for (uint32_t k = 0; k < n_kids; ++k) {
                        if (empty[k]) continue;
                        double ql = std::floor(((double)kids[k].box[c] - (double)lo[c]) / step);
                        double qh = std::ceil(((double)kids[k].box[4 + c] - (double)lo[c]) / step);
                        ql = ql < 0.0 ? 0.0 : ql;
                        qh = qh > 255.0 ? 255.0 : qh;
                        // (the device forms plane = origin + q * step inside one fused multiply-add with the ray's terms: held here
                        // to the exact planes in double, its rounding being a hundredth of the boxes' padding)
                        if ((double)lo[c] + ql * step > (double)kids[k].box[c] || (double)lo[c] + qh * step < (double)kids[k].box[4 + c])
                            return rrt_set_error(RRT_ERR_LIMIT, "clean tree: a child's box does not fit its node's byte grid");
                        planes[2 * c] |= (uint32_t)ql << (8 * k);
                        planes[2 * c + 1] |= (uint32_t)qh << (8 * k);
                    }
                }
                float* r = &rec[(size_t)it.self * 16];
                for (int c = 0; c < 3; ++c) r[c] = lo[c];
                r[3] = put_u(expo[0] | (expo[1] << 8) | (expo[2] << 16));
                for (int q = 0; q < 6; ++q) r[4 + q] = put_u(planes[q]);
                for (uint32_t k = 0; k < n_kids; ++k) {
                    uint32_t link;
                    if (!link_of(kids[k], &link)) {
                        link = new_inner();
                        todo.push_back({kids[k], link, pending_here});
                        r = &rec[(size_t)it.self * 16];  // (new_inner may have moved the array)
                    }
                    r[10 + k] = put_u(link);
                }
            }
            s->d.root_link = root_link;
            if (clean_depth + 1u > kMaxStackDepth) return rrt_set_error(RRT_ERR_LIMIT, "clean tree needs more than 64 pending stack entries per ray");
            UP(nodes, (const float4*)rec.data(), rec.size() / 4);
        } else {
            std::vector<float> rec((size_t)dn.size() * 16);  // node (32 bytes) + primitive box (32 bytes) per 64-byte record
            for (size_t i = 0; i < dn.size(); ++i) {
                std::memcpy(&rec[i * 16], &dn[i], 32);
                std::memcpy(&rec[i * 16 + 8], &tb[i * 8], 32);
            }
            UP(nodes, (const float4*)rec.data(), rec.size() / 4);
        }
        // the sifting traversal of shallow trees packs (pair, lane) into 32 bits
        if (is_shallow(desc) && pairs.size() / (4 * kPairQuads) >= (1u << 25))  // (pair number x 128, the record's byte offset, stays below 2^32)
            return rrt_set_error(RRT_ERR_LIMIT, "a tree of at most 16 nodes (or a List) with 2^25 pair records or more");
        UP(pairs, (const float4*)pairs.data(), pairs.size() / 4);
    }
    {
        // Scenes whose textures are all Constant, without bump maps and media: one 32-byte record per slot with all that shading a
        // hit needs (DScene::plain); other scenes do not use it and get an empty array.
        bool plain = true;  // exactly when the plain kernel variants are chosen: see `features` below
        for (uint32_t i = 0; plain && i < desc->n_slots; ++i) plain = RRT_INFO_KIND(desc->slots[i].info) != RRT_KIND_MEDIUM;
        for (uint32_t i = 0; plain && i < desc->n_textures; ++i) plain = desc->textures[i].type == RRT_TEX_CONSTANT;
        for (uint32_t i = 0; plain && i < desc->n_materials; ++i) plain = desc->materials[i].bump < 0 && desc->materials[i].type != RRT_MAT_ISOTROPIC;
        std::vector<float> rec;
        if (plain) {
            rec.resize((size_t)desc->n_slots * 8, 0.0f);
            for (uint32_t i = 0; i < desc->n_slots; ++i) {
                const rrt_slot& sl = desc->slots[i];
                const rrt_material& m = desc->materials[RRT_INFO_MATERIAL(sl.info)];
                const uint32_t kind = RRT_INFO_KIND(sl.info);
                float* q = &rec[(size_t)i * 8];
                if (kind == RRT_KIND_TRIANGLE) std::memcpy(q, desc->tri_shade[i].nrm, 12);
                if (kind == RRT_KIND_SPHERE && sl.aux >= (1u << 27)) return rrt_set_error(RRT_ERR_LIMIT, "more than 2^27 spheres");
                const uint32_t bits = RRT_PLAIN_BITS(m.type, kind, kind == RRT_KIND_SPHERE ? sl.aux : 0u);
                std::memcpy(q + 3, &bits, 4);
                if (m.albedo >= 0) {  // Dielectric has no texture (its attenuation is 1: the colour is never read)
                    const rrt_texture& t = desc->textures[m.albedo];
                    q[4] = t.a; q[5] = t.b; q[6] = t.c;
                }
                q[7] = m.param;
            }
        }
        UP(plain, (const float4*)rec.data(), rec.size() / 4);
    }
    {
        std::vector<float> guards;
        if (is_shallow(desc)) guards = guard_table(desc, box_pads(desc));
        UP(guards, (const float4*)guards.data(), guards.size() / 4);
    }
    UP(slots, (const float4*)desc->slots, (size_t)desc->n_slots * 3);
    UP(tri_shade, (const float4*)desc->tri_shade, (size_t)desc->n_slots * 4);
    {
        // (the device copy marks spheres whose world_to_local matrix only translates: see sphere_hit)
        std::vector<rrt_sphere> sp(desc->spheres, desc->spheres + desc->n_spheres);
        for (rrt_sphere& q : sp) {
            const float* m = q.w2l;
            const bool only = m[0] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[5] == 1.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f &&
                              m[10] == 1.0f && std::isfinite(m[3]) && std::isfinite(m[7]) && std::isfinite(m[11]);
            q.pad[0] = only ? kSphereTranslationOnly : 0u;
        }
        UP(spheres, sp.data(), sp.size());
    }
    UP(media, desc->media, desc->n_media);
    UP(btris, (const float4*)desc->boundary_tris, (size_t)desc->n_boundary_tris * 3);
    UP(btri_shade, (const float4*)desc->boundary_tri_shade, (size_t)desc->n_boundary_tris * 4);
    UP(materials, desc->materials, desc->n_materials);
    UP(textures, desc->textures, desc->n_textures);
    UP(texels, desc->texels, desc->texel_bytes);
    UP(lights, desc->lights, desc->n_lights);
#undef UP
    s->d.n_nodes = desc->n_nodes;
    s->d.n_slots = desc->n_slots;
    s->d.n_lights = desc->n_lights;
    s->d.is_list = desc->aggregate_kind == RRT_AGGREGATE_LIST ? 1u : 0u;
    s->clean = desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN;
    s->d.cam = desc->camera;
    s->res_x = desc->resolution_x;
    s->res_y = desc->resolution_y;
    s->samples = desc->samples;
    s->stack_depth = tree_depth;  // what validate_desc measured on the tree, not the caller's bvh_depth (which may only be larger)
    if (desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN) s->stack_depth = clean_depth + 1u;
    s->n_textures = desc->n_textures;
    s->deep = desc->n_nodes > kDeepTreeNodes || s->clean;  // the clean mode's tie rule lives in the cooperative-leaf walk only
    // which code the scene needs (see F_* in rrt_device.hpp)
    for (uint32_t i = 0; i < desc->n_slots; ++i) {
        uint32_t kind = RRT_INFO_KIND(desc->slots[i].info);
        if (kind == RRT_KIND_SPHERE) s->features |= F_SPHERE;
        if (kind == RRT_KIND_MEDIUM) s->features |= F_MEDIUM | F_SPHERE;
    }
    for (uint32_t i = 0; i < desc->n_media; ++i) {
        if (desc->media[i].boundary_kind == RRT_KIND_MEDIUM) s->features |= F_NESTED;
    }
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind == RRT_KIND_SPHERE) s->features |= F_SPHERE;
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        if (desc->textures[i].type != RRT_TEX_CONSTANT) s->features |= F_TEX;
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        if (desc->materials[i].bump >= 0) s->features |= F_BUMP | F_TEX;
        if (desc->materials[i].type == RRT_MAT_ISOTROPIC) s->features |= F_MEDIUM;
    }
    {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, 256));
        s->owned.push_back(p);
        s->d_items = (uint32_t*)p;
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
        s->wave_places = (uint32_t)(cus > 0 ? cus : 256) * 4u * kPoolWaves;
    }
    {
        uint64_t h = 1469598103934665603ull;  // FNV-1a over everything a kernel reads
        auto mix = [&h](const void* p, size_t n) {
            const unsigned char* b = (const unsigned char*)p;
            for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
        };
        mix(desc->nodes, (size_t)desc->n_nodes * sizeof(rrt_node));
        mix(desc->slots, (size_t)desc->n_slots * sizeof(rrt_slot));
        mix(desc->spheres, (size_t)desc->n_spheres * sizeof(rrt_sphere));
        mix(desc->media, (size_t)desc->n_media * sizeof(rrt_medium));
        mix(desc->materials, (size_t)desc->n_materials * sizeof(rrt_material));
        mix(desc->textures, (size_t)desc->n_textures * sizeof(rrt_texture));
        mix(desc->lights, (size_t)desc->n_lights * sizeof(rrt_light));
        mix(&desc->camera, sizeof desc->camera);
        mix(&desc->aggregate_kind, sizeof desc->aggregate_kind);
        s->desc_hash = h;
    }
    const auto t_done = std::chrono::steady_clock::now();
    s->upload_seconds = std::chrono::duration<double>(t_done - started).count();
    s->upload_parts[0] = std::chrono::duration<double>(t_context - t_validated).count();
    s->upload_parts[1] = std::chrono::duration<double>(t_code - t_context).count();
    s->upload_parts[3] = t_copy_seconds;
    s->upload_parts[2] = s->upload_seconds - s->upload_parts[0] - s->upload_parts[1] - s->upload_parts[3];
    *out = owner.release();
    return RRT_OK;
}

int rrt_scene_create(const rrt_scene_desc* desc, rrt_scene** out) {
    return rrt_guard("rrt_scene_create", [&]() -> int { return scene_create(desc, out); });
}

int rrt_scene_create_on(const rrt_scene_desc* desc, int device, rrt_scene** out) {
    int before = -1, count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_create_on: no device " + std::to_string(device) + " (" + std::to_string(count) + " visible)");
    HIP_TRY(hipGetDevice(&before));
    HIP_TRY(hipSetDevice(device));
    const int rc = rrt_scene_create(desc, out);
    (void)hipSetDevice(before);
    return rc;
}

int rrt_device_count(int* count) {
    if (!count) return rrt_set_error(RRT_ERR_INVALID, "rrt_device_count: NULL argument");
    HIP_TRY(hipGetDeviceCount(count));
    return RRT_OK;
}

int rrt_scene_upload_seconds(const rrt_scene* s, double* seconds) {
    if (!s || !seconds) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_upload_seconds: NULL argument");
    *seconds = s->upload_seconds;
    return RRT_OK;
}

int rrt_scene_upload_breakdown(const rrt_scene* s, double* first_device_call, double* code_object, double* host_preparation, double* copies) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_upload_breakdown: NULL scene");
    if (first_device_call) *first_device_call = s->upload_parts[0];
    if (code_object) *code_object = s->upload_parts[1];
    if (host_preparation) *host_preparation = s->upload_parts[2];
    if (copies) *copies = s->upload_parts[3];
    return RRT_OK;
}

int rrt_scene_set_staging_limit(rrt_scene* s, uint64_t bytes) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_set_staging_limit: NULL scene");
    s->slab_cap = bytes ? (size_t)bytes : kSlabCapBytes;
    return RRT_OK;
}

int rrt_scene_set_pooling(rrt_scene* s, int enable) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_set_pooling: NULL scene");
    s->pooling = enable != 0;
    return RRT_OK;
}

int rrt_scene_set_lds_stack(rrt_scene* s, uint32_t entries) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_set_lds_stack: NULL scene");
    if (entries > kMaxStackDepth) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_set_lds_stack: at most 64 entries");
    s->lds_stack = entries ? entries : kLdsStackEntries;
    return RRT_OK;
}

int rrt_scene_set_timing(rrt_scene* s, int enable) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_set_timing: NULL scene");
    s->timing = enable != 0;
    return RRT_OK;
}

int rrt_scene_kernel_times(rrt_scene* s, double* trace_ms, double* accumulate_ms, uint32_t* passes) {
    if (!s) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_kernel_times: NULL scene");
    double t = 0.0, a = 0.0;
    for (size_t i = 0; i + 3 <= s->events_used; i += 3) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(s->events[i + 2]));
        HIP_TRY(hipEventElapsedTime(&ms, s->events[i], s->events[i + 1]));
        t += ms;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[i + 1], s->events[i + 2]));
        a += ms;
    }
    if (trace_ms) *trace_ms = t;
    if (accumulate_ms) *accumulate_ms = a;
    if (passes) *passes = (uint32_t)(s->events_used / 3);
    s->events_used = 0;
    return RRT_OK;
}

void rrt_scene_destroy(rrt_scene* s) {
    if (!s) return;
    int before = -1;
    (void)hipGetDevice(&before);
    MultiState& m = s->multi;
    if (m.landing) {  // (lives on the first handle's device)
        (void)hipSetDevice(m.landing_device);
        (void)hipFree(m.landing);
    }
    (void)hipSetDevice(s->device);
    if (m.stream) (void)hipStreamSynchronize(m.stream);
    (void)hipFree(m.d_img);
    (void)hipFree(m.d_cnt);
    (void)hipFree(m.d_pack);
    for (hipEvent_t e : {m.begin, m.end, m.landed})
        if (e) (void)hipEventDestroy(e);
    if (m.stream) (void)hipStreamDestroy(m.stream);
    if (before >= 0) (void)hipSetDevice(before);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->slab) (void)hipFree(s->slab);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    delete s;
}

// A scene's buffers live on the device that was current in rrt_scene_create; launching from another device would hand
// the kernels pointers that device cannot reach.
static int check_device(const rrt_scene* s) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != s->device)
        return rrt_set_error(RRT_ERR_INVALID, "the scene was created on device " + std::to_string(s->device) + " but the current device is " + std::to_string(cur));
    return RRT_OK;
}

int rrt_intersect_batch_device(rrt_scene* s, const float* d_o, const float* d_d, size_t n, float t_min, float t_max, uint32_t aabb_mode,
                               uint32_t seed, int32_t* d_prim, float* d_t, void* stream) {
    if (!s || (n && (!d_o || !d_d || !d_prim || !d_t))) return rrt_set_error(RRT_ERR_INVALID, "rrt_intersect_batch: NULL argument");
    if (aabb_mode > RRT_AABB_EXHAUSTIVE) return rrt_set_error(RRT_ERR_INVALID, "rrt_intersect_batch: unknown aabb_mode");
    if (int rc = check_device(s)) return rc;
    if (n == 0) return RRT_OK;
    if (n > (size_t)0x7fffffff * kBlock) return rrt_set_error(RRT_ERR_LIMIT, "rrt_intersect_batch: too many rays for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (s->deep) {
        // a run of up to 1024 rays per wave (16 per lane to refill from), shorter while the batch has fewer than 16384 runs
        uint32_t run = 1024;
        while (run > 64u && (n + run - 1) / run < 16384u) run /= 2u;
        const uint32_t lds_entries = s->stack_depth > s->lds_stack ? s->lds_stack : s->stack_depth;  // the rest in private memory
        const size_t lds = (size_t)kTraceBlock * lds_entries * sizeof(uint32_t) + kCoopBytesPerWave;
        const dim3 grid((unsigned)((n + run - 1) / run));
        // deep trees: the reference's test alone IS exhaustive (leaves are tested whole)
        const int mode = s->clean ? kAabbClean : (aabb_mode != RRT_AABB_TIGHT ? (int)RRT_AABB_REFERENCE : (int)RRT_AABB_TIGHT);
        HIP_TRY(launch_intersect_stream(mode, grid, lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, lds_entries, run, s->clean ? 2u : 24u, d_prim, d_t));
    } else {
        const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
        const size_t lds = (size_t)kBlock * s->stack_depth * sizeof(uint32_t) + (size_t)(kBlock / 64) * kSiftBytesPerWave;
        if (aabb_mode == RRT_AABB_EXHAUSTIVE)
            hipLaunchKernelGGL((k_intersect<kAabbExhaustive>), grid, dim3(kBlock), lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, s->stack_depth, d_prim, d_t);
        else if (aabb_mode == RRT_AABB_REFERENCE)
            hipLaunchKernelGGL((k_intersect<RRT_AABB_REFERENCE>), grid, dim3(kBlock), lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, s->stack_depth, d_prim, d_t);
        else
            hipLaunchKernelGGL((k_intersect<RRT_AABB_TIGHT>), grid, dim3(kBlock), lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, s->stack_depth, d_prim, d_t);
    }
    HIP_TRY(hipGetLastError());
    return RRT_OK;
}

int rrt_intersect_batch(rrt_scene* s, const float* o, const float* d, size_t n, float t_min, float t_max, uint32_t aabb_mode, uint32_t seed,
                        int32_t* prim, float* t_hit) {
    if (!s || (n && (!o || !d || !prim || !t_hit))) return rrt_set_error(RRT_ERR_INVALID, "rrt_intersect_batch: NULL argument");
    if (n == 0) return RRT_OK;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int32_t* d_p = nullptr;
    int rc = RRT_OK;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_t); (void)hipFree(d_p); };
#define TRY2(expr)                                                                                               \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) { cleanup(); return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    TRY2(hipMalloc((void**)&d_o, n * 12));
    TRY2(hipMalloc((void**)&d_d, n * 12));
    TRY2(hipMalloc((void**)&d_t, n * 4));
    TRY2(hipMalloc((void**)&d_p, n * 4));
    TRY2(hipMemcpy(d_o, o, n * 12, hipMemcpyHostToDevice));
    TRY2(hipMemcpy(d_d, d, n * 12, hipMemcpyHostToDevice));
    rc = rrt_intersect_batch_device(s, d_o, d_d, n, t_min, t_max, aabb_mode, seed, d_p, d_t, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY2(hipDeviceSynchronize());
    TRY2(hipMemcpy(prim, d_p, n * 4, hipMemcpyDeviceToHost));
    TRY2(hipMemcpy(t_hit, d_t, n * 4, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

struct RenderPlan {
    TraceArgs ta;
    uint32_t spp;
};

static int plan_render(rrt_scene* s, const rrt_render_params* p, RenderPlan* plan) {
    if (!s || !p) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL argument");
    if (int rc = check_device(s)) return rc;
    if (p->aabb_mode > RRT_AABB_EXHAUSTIVE) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: unknown aabb_mode");
    if (p->background > RRT_BG_SKY) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: unknown background");
    if (p->flags & ~(uint32_t)RRT_RENDER_ACCUMULATE) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: unknown flags");
    TraceArgs& ta = plan->ta;
    ta.W = p->width ? p->width : s->res_x;
    ta.H = p->height ? p->height : s->res_y;
    plan->spp = p->samples ? p->samples : s->samples;
    if (ta.W == 0 || ta.H == 0) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: zero resolution");
    if ((uint64_t)ta.W * ta.H > 0xffffffffull) return rrt_set_error(RRT_ERR_LIMIT, "rrt_render: more than 2^32 pixels");
    ta.first_sample = p->first_sample;
    ta.n_samples = 0;
    ta.seed = p->seed;
    ta.bg = p->background;
    ta.max_depth = p->max_depth < 0 ? 50u : (uint32_t)p->max_depth;
    ta.tile_count = p->tile_count ? p->tile_count : 1;
    ta.tile_index = p->tile_index;
    if (ta.tile_index >= ta.tile_count) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: tile_index >= tile_count");
    ta.tiles_x = (ta.W + 31) / 32;
    ta.n_tiles = ta.tiles_x * ((ta.H + 31) / 32);
    uint32_t mine = ta.n_tiles > ta.tile_index ? (ta.n_tiles - ta.tile_index + ta.tile_count - 1) / ta.tile_count : 0;
    ta.n_t8 = mine * 16u;
    return RRT_OK;
}

// Enqueues the passes of one render.  static_staging = false: the slab is the scene handle's own buffer, (re)allocated here when
// the frame needs more; true: the slab is the current device's copy of the built-in staging area (rrt_noalloc_area.hip) and nothing
// here allocates, frees or creates anything (no event is recorded either: the per-kernel timing of rrt_scene_set_timing does not cover these renders).
static int render_passes(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream, bool static_staging) {
    RenderPlan plan;
    int rc = plan_render(s, p, &plan);
    if (rc) return rc;
    if (!d_rgb_sum) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL output");
    TraceArgs ta = plan.ta;
    if (ta.n_t8 == 0 || plan.spp == 0) return RRT_OK;
    const size_t frame_floats = (size_t)ta.W * ta.H * 3;
    const size_t frame_bytes = frame_floats * sizeof(float);
    if (static_staging && frame_bytes > kStaticSlabBytes)
        return rrt_set_error(RRT_ERR_LIMIT, "rrt_render_device_noalloc: a frame of " + std::to_string(ta.W) + " x " + std::to_string(ta.H) + " pixels needs " +
                                                std::to_string(frame_bytes) + " bytes of staging per sample, the built-in staging area holds " +
                                                std::to_string(kStaticSlabBytes) + " (at most " + std::to_string(kStaticSlabBytes / 12) + " pixels)");
    // samples per pass: as many as fit the staging buffer, in whole chunks when possible
    uint64_t per_pass = (static_staging ? kStaticSlabBytes : s->slab_cap) / frame_bytes;
    if (per_pass == 0) per_pass = 1;
    if (per_pass > plan.spp) per_pass = plan.spp;
    if (per_pass >= kChunk) per_pass -= per_pass % kChunk;
    // equal passes (in whole chunks) rather than full ones plus a short tail: every pass then fills the chip for as long
    const uint64_t n_passes = (plan.spp + per_pass - 1) / per_pass;
    const uint64_t even = ((plan.spp + n_passes - 1) / n_passes + kChunk - 1) / kChunk * kChunk;
    if (even < per_pass) per_pass = even;
    float* slab = nullptr;
    if (static_staging) {
        if (int rc2 = noalloc_area(s->device, &slab)) return rc2;
    } else if (s->slab_bytes < (size_t)per_pass * frame_bytes) {
        if (s->slab) {
            HIP_TRY(hipDeviceSynchronize());  // earlier renders on any stream may still read the old buffer
            HIP_TRY(hipFree(s->slab));
            s->slab = nullptr;
            s->slab_bytes = 0;
        }
        // A device short of memory (other handles, other processes, a smaller board) gets more, smaller passes instead of an error:
        // the frame is the same bits whatever the split (k_accumulate adds in sample order).
        for (;;) {
            const size_t want = (size_t)per_pass * frame_bytes;
            const hipError_t e = hipMalloc((void**)&s->slab, want);
            if (e == hipSuccess) {
                s->slab_bytes = want;
                break;
            }
            (void)hipGetLastError();  // the failed allocation must not show up as the next launch's error
            s->slab = nullptr;
            if (e != hipErrorOutOfMemory || per_pass == 1)
                return rrt_set_error(RRT_ERR_DEVICE, "rrt_render: no device memory for the staging buffer of one sample per pixel (" + std::to_string(want) +
                                                         " bytes): " + hipGetErrorString(e));
            per_pass = per_pass / 2 >= kChunk ? per_pass / 2 / kChunk * kChunk : (per_pass + 1) / 2;
        }
    }
    ta.stack_depth = s->deep && s->stack_depth > s->lds_stack ? s->lds_stack : s->stack_depth;
    const size_t lds = (size_t)kTraceBlock * ta.stack_depth * sizeof(uint32_t) + (s->deep ? kCoopBytesPerWave : kSiftBytesPerWave + kHarvestBytesPerWave);
    // Lanes at a leaf / finished lanes that trigger a leaf / shading phase.  Reference trees (a lane waits at its leaf), measured on a
    // 500 000-triangle mesh, 1024x1024 @ 64 spp: 8/16 -> 172 ms, 16/16 -> 147 ms, 24/16 -> 140 ms, 32/16 -> 143 ms, 16/8 -> 154 ms,
    // 16/32 -> 152 ms.  Clean trees (a lane keeps its leaf and walks on, see walk_step): the sooner a held leaf is tested the fewer
    // nodes are visited with a stale closest distance; 500 000 triangles 1024x1024 @ 16 / 10 M triangles 3840x2160 @ 8:
    // 2 -> 37.2 / 267.6 ms, 8 -> 38.0 / 274.9, 24 -> 39.6 / 280.4, 40 -> 42.9 / 301.7 (without holding: 40.0 / 286.1).
    ta.coop_lanes = s->clean ? 2 : 24;
    ta.shade_lanes = 16;
    // Node steps per turn of k_trace_stream's loop (the bookkeeping of a turn - the ballots, the phase decisions - costs a quarter of a
    // node step).  10 M triangles at 3840x2160@8 / 500 000 triangles at 1024x1024@32, persistent grid: 2 -> 275.9 / 57.4 ms, 3 -> 246.8 /
    // 50.7, 4 -> 241.2 / 48.6, 6 -> 233.4 / 48.6, 8 -> 231.1 / 51.0.  (Shading at 8 / 16 / 24 / 32 finished lanes: 244.6 / 241.2 / 237.6 /
    // 243.5 and 49.4 / 48.6 / 49.6 / 52.5 ms.)
    ta.walk_steps = 6;
    hipStream_t st = (hipStream_t)stream;
    // compiled variants: plain (triangles + spheres, constant textures), textured (+ textures, bump maps), full (+ media), and
    // full with media that bound media (rare, and the code for it costs the scenes with ordinary media 5 % when it rides along)
    constexpr uint32_t kPlain = F_SPHERE, kTextured = F_SPHERE | F_TEX | F_BUMP;
    const uint32_t need = s->features;
    unsigned long long* cnt = (unsigned long long*)d_counters;
    for (uint32_t done = 0; done < plan.spp; done += (uint32_t)per_pass) {
        hipEvent_t* ev = nullptr;
        if (s->timing && !static_staging && s->events_used + 3 <= kMaxTimedPasses * 3) {  // passes beyond the cap run untimed until rrt_scene_kernel_times is called
            while (s->events.size() < s->events_used + 3) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                s->events.push_back(e);
            }
            ev = &s->events[s->events_used];
            s->events_used += 3;
            HIP_TRY(hipEventRecord(ev[0], st));
        }
        ta.n_samples = plan.spp - done < per_pass ? plan.spp - done : (uint32_t)per_pass;
        ta.first_sample = plan.ta.first_sample + done;
        // Wave items are (8x8 pixels) x `chunk` samples, handed out to persistent waves through an atomic counter.  The launch ends
        // when the last item does, so an item must be short against the launch: chunks are halved until the pass has enough items.
        // Shallow trees (k_trace_pool; an item of 64 samples per pixel takes ~0.2 ms): 163 840 items, chunks of 4 at least -
        // Cornell_Box 512x512@64 2.97 -> 2.62 ms, headline scene 1024x1024@128 29.6 -> 28.2 ms, Materials 1920x1080@64 12.4 -> 12.5 ms
        // (655 360 items: Materials 13.7 ms, Globe 3840x2160@32 8.4 -> 13.3 ms: the per-item work then shows).  Deep trees
        // (k_trace_stream; a 10 M-triangle mesh at 3840x2160@16 spends 46 ms on an item of 16 samples per pixel): 655 360 items,
        // chunks down to 1 - 40 960 / 163 840 / 655 360 / 2 621 440 items: 474.7 / 443.0 / 437.1 / 457.4 ms there, 101.5 / 92.4 / 90.1 /
        // 90.6 ms on 500 000 triangles at 1024x1024@64.
        // (k_trace, one block per item: 40 960 items, chunks of 16 at least - shorter blocks cost more in ramp-up and drain than they balance)
        const bool exhaustive = p->aabb_mode == RRT_AABB_EXHAUSTIVE && !s->deep;
        // (the pool packs slot + 1 into 25 bits, the depth into 8, the sample into 19, the chosen light into 6)
        const bool pooled = !exhaustive && s->pooling && !s->deep && ta.max_depth <= kPoolMaxDepth && ta.n_samples <= kPoolMaxSamples &&
                            s->d.n_lights <= kPoolMaxLights && s->d.n_slots < kPoolMaxSlots;
        const uint64_t want = s->deep ? 655360 : (pooled ? 163840 : 40960);
        const uint32_t least = s->deep ? 1u : (pooled ? 4u : 16u);
        ta.chunk = kChunk;
        while (ta.chunk > least && (uint64_t)((ta.n_samples + ta.chunk - 1) / ta.chunk) * ta.n_t8 < want) ta.chunk /= 2u;
        const uint64_t items = (uint64_t)((ta.n_samples + ta.chunk - 1) / ta.chunk) * ta.n_t8;
        if (items > 0x7fffffffull) return rrt_set_error(RRT_ERR_LIMIT, "rrt_render: too many work items in one pass");
        const dim3 grid((unsigned)items);
        float* const pass_slab = static_staging ? slab : s->slab;
        // shallow trees: the pooled schedule (k_trace_pool) - a persistent grid that hands the items out itself
        if (exhaustive) {
            // the yardstick mode on a shallow tree: one kernel variant (every feature, one path per lane); speed is not its point
            if (need & F_NESTED) hipLaunchKernelGGL((k_trace<kAabbExhaustive, F_EVERY>), grid, dim3(kTraceBlock), lds, st, s->d, ta, pass_slab, cnt);
            else hipLaunchKernelGGL((k_trace<kAabbExhaustive, F_ALL>), grid, dim3(kTraceBlock), lds, st, s->d, ta, pass_slab, cnt);
        } else if (pooled) {
            ta.n_items = (uint32_t)items;
            const dim3 pgrid((unsigned)(items < s->wave_places ? items : s->wave_places));
            if (s->harvest) ta.stack_depth = 0;  // a single leaf: the walk never pushes
            const size_t plds = (size_t)kTraceBlock * ta.stack_depth * sizeof(uint32_t) + pool_bytes_per_wave(s->harvest);
            HIP_TRY(hipMemsetAsync(s->d_items, 0, sizeof(uint32_t), st));
#define RRT_LAUNCH_POOL(MODE, FEAT, HARV) hipLaunchKernelGGL((k_trace_pool<MODE, FEAT, HARV>), pgrid, dim3(kTraceBlock), plds, st, s->d, ta, pass_slab, cnt, s->d_items)
#define RRT_PICK_POOL(MODE)                                                  \
    do {                                                                     \
        if ((need & ~kPlain) == 0) {                                         \
            if (s->harvest) RRT_LAUNCH_POOL(MODE, kPlain, true);             \
            else RRT_LAUNCH_POOL(MODE, kPlain, false);                       \
        } else if ((need & ~kTextured) == 0) RRT_LAUNCH_POOL(MODE, kTextured, false); \
        else if (need & F_NESTED) RRT_LAUNCH_POOL(MODE, F_EVERY, false);     \
        else RRT_LAUNCH_POOL(MODE, F_ALL, false);                            \
    } while (0)
            if (p->aabb_mode == RRT_AABB_REFERENCE) RRT_PICK_POOL(RRT_AABB_REFERENCE);
            else RRT_PICK_POOL(RRT_AABB_TIGHT);
#undef RRT_PICK_POOL
#undef RRT_LAUNCH_POOL
        } else {
#define RRT_LAUNCH(MODE, FEAT) hipLaunchKernelGGL((k_trace<MODE, FEAT>), grid, dim3(kTraceBlock), lds, st, s->d, ta, pass_slab, cnt)
#define RRT_PICK(MODE)                                                               \
    do {                                                                             \
        if ((need & ~kPlain) == 0) {                                                 \
            if (s->harvest) hipLaunchKernelGGL((k_trace<MODE, kPlain, true>), grid, dim3(kTraceBlock), lds, st, s->d, ta, pass_slab, cnt); \
            else RRT_LAUNCH(MODE, kPlain);                                           \
        }                                                                            \
        else if ((need & ~kTextured) == 0) RRT_LAUNCH(MODE, kTextured);              \
        else if (need & F_NESTED) RRT_LAUNCH(MODE, F_EVERY);                         \
        else RRT_LAUNCH(MODE, F_ALL);                                                \
    } while (0)
        if (s->deep) {  // (clean trees are deep by definition)
            const int mode = s->clean ? kAabbClean : (p->aabb_mode != RRT_AABB_TIGHT ? (int)RRT_AABB_REFERENCE : (int)RRT_AABB_TIGHT);
            ta.n_items = (uint32_t)items;
            const dim3 sgrid((unsigned)(items < s->wave_places ? items : s->wave_places));  // one round of the chip, see k_trace_stream
            HIP_TRY(hipMemsetAsync(s->d_items, 0, 8 * sizeof(uint32_t), st));
            HIP_TRY(launch_trace_stream(mode, need, sgrid, lds, st, s->d, ta, pass_slab, cnt, s->d_items));
        } else if (p->aabb_mode != RRT_AABB_TIGHT) RRT_PICK(RRT_AABB_REFERENCE);
        else RRT_PICK(RRT_AABB_TIGHT);
        }
#undef RRT_PICK
#undef RRT_LAUNCH
        HIP_TRY(hipGetLastError());
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        const int first_pass = (done == 0 && !(p->flags & RRT_RENDER_ACCUMULATE)) ? 1 : 0;
        if (ta.tile_count <= 1 && frame_floats % 4 == 0 && ((uintptr_t)pass_slab | (uintptr_t)d_rgb_sum) % 16 == 0)
            hipLaunchKernelGGL(k_accumulate4, dim3((unsigned)((frame_floats / 4 + 255) / 256)), dim3(256), 0, st, (const v4f*)pass_slab, ta.n_samples,
                               frame_floats / 4, first_pass, (v4f*)d_rgb_sum);
        else
            hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((frame_floats + 255) / 256)), dim3(256), 0, st, pass_slab, ta.n_samples, frame_floats, ta.W,
                               ta.tiles_x, ta.tile_count, ta.tile_index, first_pass, d_rgb_sum);
        HIP_TRY(hipGetLastError());
        if (ev) HIP_TRY(hipEventRecord(ev[2], st));
    }
    return RRT_OK;
}

int rrt_render_device(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream) {
    return render_passes(s, p, d_rgb_sum, d_counters, stream, false);
}

// The same render with the slab in the built-in staging area.  Every (sample, pixel) record a pass reads was written by that
// pass's trace kernel (pixels of other ranks' tiles and beyond the frame's edge are neither written nor read), so nothing in the
// area has to start at zero and the previous call's contents are simply overwritten.
int rrt_render_device_noalloc(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream) {
    return render_passes(s, p, d_rgb_sum, d_counters, stream, true);
}

// Sums the (trace, accumulate) durations of the passes recorded since `mark` and forgets them again unless the caller's own
// timing is on (then they stay for rrt_scene_kernel_times).  The stream must have been synchronised.
static int passes_since(rrt_scene* s, size_t mark, bool keep, double* trace_s, double* accumulate_s) {
    double t = 0.0, a = 0.0;
    for (size_t i = mark; i + 3 <= s->events_used; i += 3) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[i], s->events[i + 1]));
        t += ms;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[i + 1], s->events[i + 2]));
        a += ms;
    }
    if (!keep) s->events_used = mark;
    *trace_s = t * 1e-3;
    *accumulate_s = a * 1e-3;
    return RRT_OK;
}

int rrt_render(rrt_scene* s, const rrt_render_params* p, float* rgb_sum, rrt_stats* stats) {
    RenderPlan plan;
    int rc = plan_render(s, p, &plan);
    if (rc) return rc;
    if (!rgb_sum) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL output");
    size_t bytes = (size_t)plan.ta.W * plan.ta.H * 3 * sizeof(float);
    float* d_img = nullptr;
    uint64_t* d_cnt = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // before the copy in, before the render, after it, after the copy out
    const bool user_timing = s->timing;
    const size_t mark = s->events_used;
    auto cleanup = [&]() {
        (void)hipFree(d_img);
        (void)hipFree(d_cnt);
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        s->timing = user_timing;
    };
    TRY2(hipMalloc((void**)&d_img, bytes));
    TRY2(hipMalloc((void**)&d_cnt, 16));
    for (hipEvent_t& e : ev) TRY2(hipEventCreate(&e));
    TRY2(hipEventRecord(ev[0], nullptr));
    // pixels of tiles this call does not own, and sums a call with RRT_RENDER_ACCUMULATE adds to, are the caller's values
    const bool keeps_values = (p->flags & RRT_RENDER_ACCUMULATE) != 0 || plan.ta.tile_count > 1;
    if (keeps_values) TRY2(hipMemcpyAsync(d_img, rgb_sum, bytes, hipMemcpyHostToDevice, nullptr));
    TRY2(hipMemsetAsync(d_cnt, 0, 16, nullptr));
    TRY2(hipEventRecord(ev[1], nullptr));
    s->timing = stats != nullptr || user_timing;
    rc = rrt_render_device(s, p, d_img, d_cnt, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY2(hipEventRecord(ev[2], nullptr));
    TRY2(hipMemcpyAsync(rgb_sum, d_img, bytes, hipMemcpyDeviceToHost, nullptr));
    TRY2(hipEventRecord(ev[3], nullptr));
    TRY2(hipEventSynchronize(ev[3]));
    if (stats) {
        uint64_t c[2];
        TRY2(hipMemcpy(c, d_cnt, 16, hipMemcpyDeviceToHost));
        float ms_in = 0.0f, ms = 0.0f, ms_out = 0.0f;
        TRY2(hipEventElapsedTime(&ms_in, ev[0], ev[1]));
        TRY2(hipEventElapsedTime(&ms, ev[1], ev[2]));
        TRY2(hipEventElapsedTime(&ms_out, ev[2], ev[3]));
        std::memset(stats, 0, sizeof *stats);
        stats->rays = c[0];
        stats->paths = c[1];
        stats->render_seconds = (double)ms * 1e-3;
        stats->copy_seconds = ((double)ms_in + (double)ms_out) * 1e-3;
        rc = passes_since(s, mark, user_timing, &stats->trace_seconds, &stats->accumulate_seconds);
        if (rc) { cleanup(); return rc; }
    }
    cleanup();
    return RRT_OK;
}

// ---- several GPUs in one process ------------------------------------------------------------------------------------------
// ≙ main.rs:136-165 (N worker threads, one colour buffer).  One host thread per scene handle: it makes the handle's device
// current, traces that device's share into a frame in that device's HBM and, on the same stream and therefore as soon as that
// device's render is done (no device waits for a slower one), PUSHES its share into a landing buffer on the first device
// (hipMemcpyPeerAsync; peer access between the two devices is switched on once, so the copy is one device writing into the other's
// memory over their xGMI link).  Sample ranges move the whole frame, tiles only the owner's tiles, packed (an N-th of the frame).
// The first device then, in device order, adds the landed frames to its own (sample ranges) or unpacks the tiles into it, one float
// per thread.  Frames, landing buffers, streams and events belong to the scene handles and are kept from call to call.
namespace {
struct MultiWorker {
    rrt_scene* scene = nullptr;
    rrt_render_params params{};
    size_t mark = 0;
    bool user_timing = false;
    int rc = RRT_OK;
    std::string error;
    uint64_t counters[2] = {0, 0};
    double render_s = 0.0, trace_s = 0.0, accumulate_s = 0.0;
};
constexpr uint32_t kTileFloats = 32u * 32u * 3u;
}  // namespace

__global__ void k_add_frame(float* __restrict__ into, const float* __restrict__ from, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) into[i] = into[i] + from[i];
}

// Tile sharding.  Tile t of the frame (32x32 pixels, numbered row by row) belongs to device t % tile_count and is the (t / tile_count)-th
// of that device's tiles; packed, a device's tiles lie one after the other, 32 x 32 x 3 floats each (pixels beyond the frame's edge
// keep their place and are never written or read).  Pixels are taken over as they are: the frame stays bit-identical to one device's.
__global__ void k_pack_tiles(const float* __restrict__ frame, float* __restrict__ packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tiles_x,
                             uint32_t n_tiles, uint32_t tile_count, uint32_t tile_index) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_packed) return;
    const uint32_t t = (uint32_t)(i / kTileFloats) * tile_count + tile_index, in_tile = (uint32_t)(i % kTileFloats);
    const uint32_t x = (t % tiles_x) * 32u + (in_tile / 3u) % 32u, y = (t / tiles_x) * 32u + in_tile / 96u;
    if (t < n_tiles && x < W && y < H) packed[i] = frame[((size_t)y * W + x) * 3 + in_tile % 3u];
}
__global__ void k_unpack_tiles(float* __restrict__ frame, const float* __restrict__ packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tiles_x,
                               uint32_t n_tiles, uint32_t tile_count, uint32_t tile_index) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_packed) return;
    const uint32_t t = (uint32_t)(i / kTileFloats) * tile_count + tile_index, in_tile = (uint32_t)(i % kTileFloats);
    const uint32_t x = (t % tiles_x) * 32u + (in_tile / 3u) % 32u, y = (t / tiles_x) * 32u + in_tile / 96u;
    if (t < n_tiles && x < W && y < H) frame[((size_t)y * W + x) * 3 + in_tile % 3u] = packed[i];
}

// One device writing into another's memory needs peer access switched on once per ordered pair (from -> to).
static int enable_peer_access(int from, int to, bool* direct) {
    static std::mutex lock;
    static std::vector<std::pair<int, int>> done;  // pairs already handled, with what came of it in `direct_ok`
    static std::vector<bool> direct_ok;
    std::lock_guard<std::mutex> hold(lock);
    for (size_t k = 0; k < done.size(); ++k)
        if (done[k] == std::make_pair(from, to)) {
            *direct = direct_ok[k];
            return RRT_OK;
        }
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, from, to));
    bool ok = false;
    if (can) {
        int before = -1;
        HIP_TRY(hipGetDevice(&before));
        HIP_TRY(hipSetDevice(from));
        const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        (void)hipSetDevice(before);
    }
    done.emplace_back(from, to);
    direct_ok.push_back(ok);  // (false: hipMemcpyPeerAsync still works, staged by the runtime)
    *direct = ok;
    return RRT_OK;
}

// Makes sure the handle's multi-GPU buffers can hold this call: its frame and counters on its own device, and - for every handle
// but the first - what it pushes (on its own device: the packed tiles) and where that lands (on the first handle's device).
static int multi_prepare(rrt_scene* s, size_t frame_bytes, size_t push_bytes, int first_device, bool is_first) {
    MultiState& m = s->multi;
    HIP_TRY(hipSetDevice(s->device));
    if (!m.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&m.begin));
        HIP_TRY(hipEventCreate(&m.end));
        HIP_TRY(hipEventCreate(&m.landed));
        HIP_TRY(hipMalloc((void**)&m.d_cnt, 16));
    }
    if (m.img_bytes < frame_bytes) {
        if (m.d_img) HIP_TRY(hipFree(m.d_img));
        m.d_img = nullptr;
        m.img_bytes = 0;
        HIP_TRY(hipMalloc((void**)&m.d_img, frame_bytes));
        m.img_bytes = frame_bytes;
    }
    if (is_first) return RRT_OK;
    if (m.pack_bytes < push_bytes) {
        if (m.d_pack) HIP_TRY(hipFree(m.d_pack));
        m.d_pack = nullptr;
        m.pack_bytes = 0;
        HIP_TRY(hipMalloc((void**)&m.d_pack, push_bytes));
        m.pack_bytes = push_bytes;
    }
    if (m.landing_bytes < push_bytes || m.landing_device != first_device) {
        if (m.landing) {
            HIP_TRY(hipSetDevice(m.landing_device));
            HIP_TRY(hipFree(m.landing));
        }
        m.landing = nullptr;
        m.landing_bytes = 0;
        HIP_TRY(hipSetDevice(first_device));
        HIP_TRY(hipMalloc((void**)&m.landing, push_bytes));
        m.landing_bytes = push_bytes;
        m.landing_device = first_device;
    }
    return RRT_OK;
}

static int render_multi(rrt_scene* const* scenes, uint32_t n, uint32_t shard_mode, const rrt_render_params* p, float* rgb_sum, rrt_stats* stats) {
    if (!scenes || n == 0 || !p || !rgb_sum) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: NULL argument");
    if (n > 64) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: more than 64 scene handles");
    if (shard_mode > RRT_SHARD_TILES) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: unknown shard_mode");
    if (p->tile_count > 1 || p->tile_index != 0) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: tile_count / tile_index must be 0 (the call shards the frame itself)");
    for (uint32_t i = 0; i < n; ++i)
        if (!scenes[i]) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: NULL scene handle");
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (scenes[i] == scenes[j]) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: a scene handle appears twice (one render at a time per handle)");
    // every handle must hold the same scene: the shares of different scenes would make one frame without a word of warning
    for (uint32_t i = 1; i < n; ++i) {
        const rrt_scene *a = scenes[0], *b = scenes[i];
        if (a->res_x != b->res_x || a->res_y != b->res_y || a->samples != b->samples || a->d.n_slots != b->d.n_slots || a->d.n_nodes != b->d.n_nodes ||
            a->d.n_lights != b->d.n_lights || a->features != b->features || a->clean != b->clean || a->desc_hash != b->desc_hash)
            return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: scene handle " + std::to_string(i) + " does not hold the same scene description as handle 0");
    }
    const uint32_t W = p->width ? p->width : scenes[0]->res_x, H = p->height ? p->height : scenes[0]->res_y;
    const uint32_t spp = p->samples ? p->samples : scenes[0]->samples;
    if (W == 0 || H == 0) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_multi: zero resolution");
    const size_t frame_floats = (size_t)W * H * 3, bytes = frame_floats * sizeof(float);
    const uint32_t tiles_x = (W + 31) / 32, n_tiles = tiles_x * ((H + 31) / 32);
    const bool tiles = shard_mode == RRT_SHARD_TILES;
    const bool accumulate = (p->flags & RRT_RENDER_ACCUMULATE) != 0;
    int caller_device = -1;
    HIP_TRY(hipGetDevice(&caller_device));
    const int first_device = scenes[0]->device;
    struct Restore {  // whatever happens below, the caller's device and the handles' timing switches come back
        int device;
        std::vector<MultiWorker>* w;
        ~Restore() {
            for (MultiWorker& k : *w) k.scene->timing = k.user_timing;
            (void)hipSetDevice(device);
        }
    };

    std::vector<MultiWorker> w(n);
    // shares: contiguous sample ranges (the remainder goes to the first devices, one sample each) or interleaved tiles
    const uint32_t base = spp / n, extra = spp % n;
    uint32_t next_sample = p->first_sample;
    for (uint32_t i = 0; i < n; ++i) {
        w[i].scene = scenes[i];
        w[i].user_timing = scenes[i]->timing;
        w[i].params = *p;
        w[i].params.width = W;
        w[i].params.height = H;
        // sample ranges: the caller's sums go into the first device's share only (the others' shares are added to it);
        // tiles: every device continues the sums of its own pixels
        w[i].params.flags = (accumulate && (i == 0 || tiles)) ? RRT_RENDER_ACCUMULATE : 0u;
        if (!tiles) {
            const uint32_t mine = base + (i < extra ? 1u : 0u);
            w[i].params.samples = mine;
            w[i].params.first_sample = next_sample;
            next_sample += mine;
        } else {
            w[i].params.samples = spp;
            w[i].params.tile_count = n;
            w[i].params.tile_index = i;
        }
    }
    Restore restore{caller_device, &w};
    // what handle i pushes to the first device: its frame (sample ranges) or its tiles, packed
    auto push_floats = [&](uint32_t i) -> size_t {
        if (!tiles) return frame_floats;
        const uint32_t owned = n_tiles > i ? (n_tiles - i + n - 1) / n : 0;
        return (size_t)owned * kTileFloats;
    };
    for (uint32_t i = 0; i < n; ++i) {
        if (int rc = multi_prepare(scenes[i], bytes, std::max<size_t>(push_floats(i), 4) * sizeof(float), first_device, i == 0)) return rc;
        bool direct = true;
        if (i > 0 && scenes[i]->device != first_device)
            if (int rc = enable_peer_access(scenes[i]->device, first_device, &direct)) return rc;
    }
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // copy in (2), reduce (2), copy out (2): on the first device
    struct Events {
        hipEvent_t* e;
        int device;
        ~Events() {
            (void)hipSetDevice(device);
            for (int k = 0; k < 6; ++k)
                if (e[k]) (void)hipEventDestroy(e[k]);
        }
    } events{ev, first_device};
    HIP_TRY(hipSetDevice(first_device));
    for (hipEvent_t& e : ev) HIP_TRY(hipEventCreate(&e));
    MultiState& m0 = scenes[0]->multi;
    // with ACCUMULATE the first device's frame starts from the caller's sums (tiles: every device's frame does, see the workers)
    HIP_TRY(hipEventRecord(ev[0], m0.stream));
    if (accumulate) HIP_TRY(hipMemcpyAsync(m0.d_img, rgb_sum, bytes, hipMemcpyHostToDevice, m0.stream));
    HIP_TRY(hipEventRecord(ev[1], m0.stream));

    // one host thread per handle (main.rs:138-147)
    auto work = [&](uint32_t i) {
        MultiWorker& k = w[i];
        MultiState& m = k.scene->multi;
        auto fail = [&](hipError_t e, const char* what) {
            k.rc = RRT_ERR_DEVICE;
            k.error = std::string(what) + ": " + hipGetErrorString(e);
        };
        hipError_t e = hipSetDevice(k.scene->device);
        if (e != hipSuccess) return fail(e, "hipSetDevice");
        if ((e = hipMemsetAsync(m.d_cnt, 0, 16, m.stream)) != hipSuccess) return fail(e, "hipMemsetAsync");
        if (i > 0 && tiles && accumulate && (e = hipMemcpyAsync(m.d_img, rgb_sum, bytes, hipMemcpyHostToDevice, m.stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync");
        if ((e = hipEventRecord(m.begin, m.stream)) != hipSuccess) return fail(e, "hipEventRecord");
        k.mark = k.scene->events_used;
        k.scene->timing = true;
        if (k.params.samples != 0) {
            k.rc = render_passes(k.scene, &k.params, m.d_img, m.d_cnt, m.stream, false);
            if (k.rc) {
                k.error = rrt_last_error();
                return;
            }
        } else if (!(accumulate && (i == 0 || tiles))) {
            if ((e = hipMemsetAsync(m.d_img, 0, bytes, m.stream)) != hipSuccess) return fail(e, "hipMemsetAsync");  // nothing to trace: a frame of zeros
        }
        if ((e = hipEventRecord(m.end, m.stream)) != hipSuccess) return fail(e, "hipEventRecord");
        if (i > 0) {
            // the push follows the render on the same stream: it starts the moment THIS device is done, whatever the others do
            const size_t floats = push_floats(i);
            const float* from = m.d_img;
            if (tiles) {
                if (floats) hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)((floats + 255) / 256)), dim3(256), 0, m.stream, m.d_img, m.d_pack, floats, W, H, tiles_x, n_tiles, n, i);
                if ((e = hipGetLastError()) != hipSuccess) return fail(e, "k_pack_tiles");
                from = m.d_pack;
            }
            if (floats && (e = hipMemcpyPeerAsync(m.landing, first_device, from, k.scene->device, floats * sizeof(float), m.stream)) != hipSuccess)
                return fail(e, "hipMemcpyPeerAsync");
            if ((e = hipEventRecord(m.landed, m.stream)) != hipSuccess) return fail(e, "hipEventRecord");
        }
        if ((e = hipEventSynchronize(m.end)) != hipSuccess) return fail(e, "hipEventSynchronize");
        float ms = 0.0f;
        if ((e = hipEventElapsedTime(&ms, m.begin, m.end)) != hipSuccess) return fail(e, "hipEventElapsedTime");
        k.render_s = (double)ms * 1e-3;
        k.rc = passes_since(k.scene, k.mark, k.user_timing, &k.trace_s, &k.accumulate_s);
        if (k.rc) k.error = rrt_last_error();
        if ((e = hipMemcpyAsync(k.counters, m.d_cnt, 16, hipMemcpyDeviceToHost, m.stream)) != hipSuccess) return fail(e, "hipMemcpyAsync");
        if ((e = hipStreamSynchronize(m.stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    };
    {
        std::vector<std::thread> pool;
        for (uint32_t i = 1; i < n; ++i) pool.emplace_back(work, i);
        work(0);
        for (std::thread& t : pool) t.join();
    }
    for (const MultiWorker& k : w)
        if (k.rc) return rrt_set_error(k.rc, "rrt_render_multi, device " + std::to_string(k.scene->device) + ": " + k.error);
    // gather on the first device, in device order
    HIP_TRY(hipSetDevice(first_device));
    HIP_TRY(hipEventRecord(ev[2], m0.stream));
    for (uint32_t i = 1; i < n; ++i) {
        const MultiState& m = scenes[i]->multi;
        const size_t floats = push_floats(i);
        HIP_TRY(hipStreamWaitEvent(m0.stream, m.landed, 0));
        if (!floats) continue;
        const dim3 grid((unsigned)((floats + 255) / 256));
        if (tiles) hipLaunchKernelGGL(k_unpack_tiles, grid, dim3(256), 0, m0.stream, m0.d_img, m.landing, floats, W, H, tiles_x, n_tiles, n, i);
        else hipLaunchKernelGGL(k_add_frame, grid, dim3(256), 0, m0.stream, m0.d_img, m.landing, floats);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev[3], m0.stream));
    HIP_TRY(hipEventRecord(ev[4], m0.stream));
    HIP_TRY(hipMemcpyAsync(rgb_sum, m0.d_img, bytes, hipMemcpyDeviceToHost, m0.stream));
    HIP_TRY(hipEventRecord(ev[5], m0.stream));
    HIP_TRY(hipEventSynchronize(ev[5]));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (const MultiWorker& k : w) {
            stats->rays += k.counters[0];
            stats->paths += k.counters[1];
            if (k.render_s >= stats->render_seconds) {
                stats->render_seconds = k.render_s;
                stats->trace_seconds = k.trace_s;
                stats->accumulate_seconds = k.accumulate_s;
            }
        }
        float ms_in = 0.0f, ms_red = 0.0f, ms_out = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms_in, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms_red, ev[2], ev[3]));
        HIP_TRY(hipEventElapsedTime(&ms_out, ev[4], ev[5]));
        stats->reduce_seconds = (double)ms_red * 1e-3;
        stats->copy_seconds = ((double)ms_in + (double)ms_out) * 1e-3;
    }
    return RRT_OK;
}

int rrt_render_multi(rrt_scene* const* scenes, uint32_t n, uint32_t shard_mode, const rrt_render_params* p, float* rgb_sum, rrt_stats* stats) {
    return rrt_guard("rrt_render_multi", [&]() -> int { return render_multi(scenes, n, shard_mode, p, rgb_sum, stats); });
}

int rrt_resolve_rgb8_device(const float* d_sum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* d_out, void* stream) {
    if (!d_sum || !d_out) return rrt_set_error(RRT_ERR_INVALID, "rrt_resolve_rgb8: NULL argument");
    size_t n = (size_t)W * H * 3;
    if (n == 0) return RRT_OK;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_sum, n, (float)spp, d_out);
    HIP_TRY(hipGetLastError());
    return RRT_OK;
}

int rrt_resolve_rgb8(const float* rgb_sum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* out) {
    if (!rgb_sum || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_resolve_rgb8: NULL argument");
    size_t n = (size_t)W * H * 3;
    if (n == 0) return RRT_OK;
    float* d_sum = nullptr;
    uint8_t* d_out = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_sum); (void)hipFree(d_out); };
    TRY2(hipMalloc((void**)&d_sum, n * 4));
    TRY2(hipMalloc((void**)&d_out, n));
    TRY2(hipMemcpy(d_sum, rgb_sum, n * 4, hipMemcpyHostToDevice));
    int rc = rrt_resolve_rgb8_device(d_sum, W, H, spp, d_out, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY2(hipDeviceSynchronize());
    TRY2(hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

#ifdef RRT_PHASE_PROFILE
// profiling build only: reads and clears k_trace_pool's per-kind counters (steps[5], paths served[5], cycles[5])
extern "C" int rrt_profile_read_pool(uint64_t* out18) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out18, HIP_SYMBOL(g_pool_prof), 3 * kClasses * 8));
    uint64_t zero[3 * kClasses] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pool_prof), zero, 3 * kClasses * 8));
    return RRT_OK;
}
// profiling build only: reads and clears the sifting query's list-size histogram and cycle split (20 values)
extern "C" int rrt_profile_read_sift(uint64_t* out20) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out20, HIP_SYMBOL(g_sift_hist), 160));
    uint64_t zero[20] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_sift_hist), zero, 160));
    return RRT_OK;
}
// profiling build only: reads and clears the traversal counters
extern "C" int rrt_profile_read(uint64_t* out8) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_prof), 64));
    uint64_t zero[8] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, 64));
    uint64_t deep[8];  // the deep-tree kernels count in their own translation unit's copy
    HIP_TRY(profile_take_stream(deep));
    for (int i = 0; i < 8; ++i) out8[i] += deep[i];
    return RRT_OK;
}
#endif

int rrt_selftest_reciprocal(uint64_t* mismatches) {
    if (!mismatches) return rrt_set_error(RRT_ERR_INVALID, "rrt_selftest_reciprocal: NULL argument");
    unsigned long long* d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d); };
    TRY2(hipMalloc((void**)&d, 8));
    TRY2(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_selftest_rcp, dim3(65536), dim3(256), 0, nullptr, d);
    TRY2(hipGetLastError());
    TRY2(hipDeviceSynchronize());
    TRY2(hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

int rrt_probe(rrt_scene* s, uint32_t which, uint32_t arg, const void* in, size_t n, float* out) {
    if (!s || !in || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_probe: NULL argument");
    static const size_t in_words[5] = {3, 8, 5, 2, 2};
    static const size_t out_words[5] = {6, 14, 3, 5, 8};
    if (which > 4) return rrt_set_error(RRT_ERR_INVALID, "rrt_probe: unknown probe");
    if (int rc = check_device(s)) return rc;
    if (which == 2 && arg >= s->n_textures) return rrt_set_error(RRT_ERR_INVALID, "rrt_probe: texture index out of range");
    if (n == 0) return RRT_OK;
    uint32_t* d_in = nullptr;
    float* d_out = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_in); (void)hipFree(d_out); };
    TRY2(hipMalloc((void**)&d_in, n * in_words[which] * 4));
    TRY2(hipMalloc((void**)&d_out, n * out_words[which] * 4));
    TRY2(hipMemcpy(d_in, in, n * in_words[which] * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, s->d, which, arg, s->res_x, s->res_y, d_in, n, d_out);
    TRY2(hipGetLastError());
    TRY2(hipDeviceSynchronize());
    TRY2(hipMemcpy(out, d_out, n * out_words[which] * 4, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

}  // extern "C"
