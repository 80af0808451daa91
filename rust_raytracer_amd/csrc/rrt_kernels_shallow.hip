// The kernels for shallow trees (at most 16 nodes, or a List) and the frame kernels (accumulate, resolve, the multi-GPU gather), with
// their launchers: everything in the library that is device code and wants the shallow-tree compiler settings (see build.py).
#include <hip/hip_runtime.h>

#include <string>

#include "rrt_device.hpp"
#include "rrt_error.hpp"
#include "rrt_launch.hpp"

using namespace rrt;

namespace {

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
                if (pending) {  // aggregate.rs:79-89 for the previous bounce, now that this ray's light distances are known (thr held the numerator)
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

constexpr uint32_t kTileFloats = 32u * 32u * 3u;
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

DScene view_of(const SceneData& d) {
    DScene v;
    static_cast<SceneData&>(v) = d;
    return v;
}
dim3 per_float(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

namespace rrt {

hipError_t load_shallow_kernels() {
    hipFuncAttributes attr;  // (asks the runtime about one of this file's kernels: that loads the file's code object)
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_resolve));
}

uint32_t pool_waves_per_simd() { return kPoolWaves; }

// (the pool packs slot + 1 into 25 bits, the depth into 8, the sample into 19, the chosen light into 6)
bool pool_takes(uint32_t max_depth, uint32_t n_samples, uint32_t n_lights, uint32_t n_slots) {
    return max_depth <= kPoolMaxDepth && n_samples <= kPoolMaxSamples && n_lights <= kPoolMaxLights && n_slots < kPoolMaxSlots;
}

// compiled variants: plain (triangles + spheres, constant textures), textured (+ textures, bump maps), full (+ media), and
// full with media that bound media (rare, and the code for it costs the scenes with ordinary media 5 % when it rides along)
constexpr uint32_t kPlain = F_SPHERE, kTextured = F_SPHERE | F_TEX | F_BUMP;

hipError_t launch_trace(int aabb_mode, uint32_t need, bool harvest, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& ta, float* slab,
                        unsigned long long* cnt) {
    const DScene sc = view_of(scene);
    const size_t lds = (size_t)kTraceBlock * ta.stack_depth * sizeof(uint32_t) + kSiftBytesPerWave + kHarvestBytesPerWave;
    if (aabb_mode == kAabbExhaustive) {
        // the yardstick mode on a shallow tree: one kernel variant (every feature, one path per lane); speed is not its point
        if (need & F_NESTED) hipLaunchKernelGGL((k_trace<kAabbExhaustive, F_EVERY>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, cnt);
        else hipLaunchKernelGGL((k_trace<kAabbExhaustive, F_ALL>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, cnt);
        return hipGetLastError();
    }
#define RRT_LAUNCH(MODE, FEAT) hipLaunchKernelGGL((k_trace<MODE, FEAT>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, cnt)
#define RRT_PICK(MODE)                                                               \
    do {                                                                             \
        if ((need & ~kPlain) == 0) {                                                 \
            if (harvest) hipLaunchKernelGGL((k_trace<MODE, kPlain, true>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, cnt); \
            else RRT_LAUNCH(MODE, kPlain);                                           \
        }                                                                            \
        else if ((need & ~kTextured) == 0) RRT_LAUNCH(MODE, kTextured);              \
        else if (need & F_NESTED) RRT_LAUNCH(MODE, F_EVERY);                         \
        else RRT_LAUNCH(MODE, F_ALL);                                                \
    } while (0)
    if (aabb_mode != RRT_AABB_TIGHT) RRT_PICK(RRT_AABB_REFERENCE);
    else RRT_PICK(RRT_AABB_TIGHT);
#undef RRT_PICK
#undef RRT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_trace_pool(int aabb_mode, uint32_t need, bool harvest, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& args, float* slab,
                             unsigned long long* cnt, uint32_t* item_counter) {
    const DScene sc = view_of(scene);
    TraceArgs ta = args;
    if (harvest) ta.stack_depth = 0;  // a single leaf: the walk never pushes
    const size_t lds = (size_t)kTraceBlock * ta.stack_depth * sizeof(uint32_t) + pool_bytes_per_wave(harvest);
#define RRT_LAUNCH_POOL(MODE, FEAT, HARV) hipLaunchKernelGGL((k_trace_pool<MODE, FEAT, HARV>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, cnt, item_counter)
#define RRT_PICK_POOL(MODE)                                                  \
    do {                                                                     \
        if ((need & ~kPlain) == 0) {                                         \
            if (harvest) RRT_LAUNCH_POOL(MODE, kPlain, true);                \
            else RRT_LAUNCH_POOL(MODE, kPlain, false);                       \
        } else if ((need & ~kTextured) == 0) RRT_LAUNCH_POOL(MODE, kTextured, false); \
        else if (need & F_NESTED) RRT_LAUNCH_POOL(MODE, F_EVERY, false);     \
        else RRT_LAUNCH_POOL(MODE, F_ALL, false);                            \
    } while (0)
    if (aabb_mode == RRT_AABB_REFERENCE) RRT_PICK_POOL(RRT_AABB_REFERENCE);
    else RRT_PICK_POOL(RRT_AABB_TIGHT);
#undef RRT_PICK_POOL
#undef RRT_LAUNCH_POOL
    return hipGetLastError();
}

hipError_t launch_intersect(int aabb_mode, hipStream_t st, const SceneData& scene, const float* o, const float* d, size_t n, float t_min, float t_max,
                            uint32_t seed, uint32_t stack_depth, int* prim, float* t_hit) {
    const DScene sc = view_of(scene);
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    const size_t lds = (size_t)kBlock * stack_depth * sizeof(uint32_t) + (size_t)(kBlock / 64) * kSiftBytesPerWave;
    if (aabb_mode == kAabbExhaustive) hipLaunchKernelGGL((k_intersect<kAabbExhaustive>), grid, dim3(kBlock), lds, st, sc, o, d, n, t_min, t_max, seed, stack_depth, prim, t_hit);
    else if (aabb_mode == RRT_AABB_REFERENCE) hipLaunchKernelGGL((k_intersect<RRT_AABB_REFERENCE>), grid, dim3(kBlock), lds, st, sc, o, d, n, t_min, t_max, seed, stack_depth, prim, t_hit);
    else hipLaunchKernelGGL((k_intersect<RRT_AABB_TIGHT>), grid, dim3(kBlock), lds, st, sc, o, d, n, t_min, t_max, seed, stack_depth, prim, t_hit);
    return hipGetLastError();
}

hipError_t launch_accumulate(hipStream_t st, const float* slab, const TraceArgs& ta, int first_pass, float* rgb_sum) {
    const size_t frame_floats = (size_t)ta.W * ta.H * 3;
    if (ta.tile_count <= 1 && frame_floats % 4 == 0 && ((uintptr_t)slab | (uintptr_t)rgb_sum) % 16 == 0)
        hipLaunchKernelGGL(k_accumulate4, per_float(frame_floats / 4), dim3(256), 0, st, (const v4f*)slab, ta.n_samples, frame_floats / 4, first_pass, (v4f*)rgb_sum);
    else
        hipLaunchKernelGGL(k_accumulate, per_float(frame_floats), dim3(256), 0, st, slab, ta.n_samples, frame_floats, ta.W, ta.tiles_x, ta.tile_count, ta.tile_index,
                           first_pass, rgb_sum);
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t st, const float* sum, size_t n, float spp, uint8_t* out) {
    hipLaunchKernelGGL(k_resolve, per_float(n), dim3(256), 0, st, sum, n, spp, out);
    return hipGetLastError();
}

hipError_t launch_probe(const SceneData& scene, uint32_t which, uint32_t arg, uint32_t W, uint32_t H, const uint32_t* in, size_t n, float* out) {
    hipLaunchKernelGGL(k_probe, per_float(n), dim3(256), 0, nullptr, view_of(scene), which, arg, W, H, in, n, out);
    return hipGetLastError();
}

hipError_t launch_selftest_rcp(unsigned long long* mismatches) {
    hipLaunchKernelGGL(k_selftest_rcp, dim3(65536), dim3(256), 0, nullptr, mismatches);
    return hipGetLastError();
}

hipError_t launch_add_frame(hipStream_t st, float* into, const float* from, size_t n) {
    hipLaunchKernelGGL(k_add_frame, per_float(n), dim3(256), 0, st, into, from, n);
    return hipGetLastError();
}

hipError_t launch_pack_tiles(hipStream_t st, const float* frame, float* packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tile_count, uint32_t tile_index) {
    const uint32_t tiles_x = (W + 31) / 32, n_tiles = tiles_x * ((H + 31) / 32);
    hipLaunchKernelGGL(k_pack_tiles, per_float(n_packed), dim3(256), 0, st, frame, packed, n_packed, W, H, tiles_x, n_tiles, tile_count, tile_index);
    return hipGetLastError();
}

hipError_t launch_unpack_tiles(hipStream_t st, float* frame, const float* packed, size_t n_packed, uint32_t W, uint32_t H, uint32_t tile_count, uint32_t tile_index) {
    const uint32_t tiles_x = (W + 31) / 32, n_tiles = tiles_x * ((H + 31) / 32);
    hipLaunchKernelGGL(k_unpack_tiles, per_float(n_packed), dim3(256), 0, st, frame, packed, n_packed, W, H, tiles_x, n_tiles, tile_count, tile_index);
    return hipGetLastError();
}

}  // namespace rrt

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

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
