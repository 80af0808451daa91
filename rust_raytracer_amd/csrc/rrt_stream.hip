// The kernels for deep trees (more than 16 nodes, or a clean tree): the streaming path tracer and the streaming ray batch.
#include <hip/hip_runtime.h>

#include "rrt_device.hpp"
#include "rrt_launch.hpp"

using namespace rrt;

namespace {

// ---- the trace kernel for deep trees ------------------------------------------------------------------------
// Same work and same results as k_trace, other schedule.  On a large mesh the rays of a wave need very
// different numbers of node visits (a mean of 300 against 2 000 for the slowest lane of a wave on a 10 M-triangle mesh), and a
// loop that runs query - shade - query in step leaves most lanes idle most of the time.  Here every lane is a small state
// machine - DEAD (no path: takes the next undrawn sample), FRESH (has a sample, no ray yet), WALK (visiting nodes), LEAF (stands
// at a leaf), READY (query finished) - and the wave's loop does, each turn: hand samples to the DEAD lanes; a few node steps for
// the WALK lanes; a cooperative leaf phase when enough lanes stand at (or hold) a leaf; a shading phase (shade, store the sample
// of a path that ended, make the camera ray of a fresh one) when enough lanes are READY or FRESH.  A lane that finishes its query
// early is shaded and sent walking again while its neighbours are still in their queries.
template <int AABB_MODE, uint32_t FEAT>
__global__ __launch_bounds__(kTraceBlock, RRT_TRACE_MIN_WAVES) void k_trace_stream(DScene sc, TraceArgs ta, float* __restrict__ slab, unsigned long long* __restrict__ counters,
                                                                                   uint32_t* __restrict__ item_counter) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x;
    // Waves are persistent: the grid is one round of the chip, and a wave whose item (a tile of 8x8 pixels x `chunk` samples) has no
    // undrawn sample left takes the next item from an atomic counter, so lanes never stand idle while their neighbours finish the
    // item's last paths, and the launch's tail is one path long.
    // The eight XCDs have an L2 each, and workgroups go to them round-robin by number.  The frame's tiles are therefore cut into
    // eight bands with a counter each: a wave draws from its own XCD's band - consecutive draws are neighbouring tiles, so the
    // waves of one XCD work side by side in the image and share the tree's nodes in that XCD's L2 - and, when its band is used
    // up, from the next XCD's (bands cost differently: sky above, mesh below).
    // Within a band the 32x32 tiles are numbered column by column, so that the ~1500 waves of an XCD cover a compact block of the
    // image (about 12 x 8 tiles) and not a strip one tile high.  (Interleaved tiles of a multi-GPU render keep the plain order.)
    const uint32_t xcd = blockIdx.x & 7u;
    // Only when a band is several times what an XCD's waves cover at once (10 M triangles at 3840x2160: 351.7 -> 345.8 ms; on a
    // 1024x1024 frame, whose band is four rows of 32 tiles, the order is all the same and the plain one ends more evenly).
    const uint32_t tiles_y = ta.n_t8 / 16u / ta.tiles_x;
    const uint32_t band_rows = (tiles_y + 7u) / 8u;  // rows of 32x32 tiles per band
    const bool by_columns = ta.tile_count == 1u && band_rows * ta.tiles_x >= 384u;
    const uint32_t band_tiles = by_columns ? band_rows * ta.tiles_x * 16u : ((ta.n_t8 + 7u) / 8u + 15u) & ~15u;  // 8x8 tiles per band: whole 32x32 tiles
    const uint32_t n_chunks = (ta.n_samples + ta.chunk - 1u) / ta.chunk;
    uint32_t bands_done = 0;                    // bands, its own first, that this wave has found used up
    uint32_t x0 = 0, y0 = 0, s0 = 0, pool = 0;  // the current item (wave-uniform)
    bool exhausted = false;                     // no items left in the pass
    uint32_t beyond[kMaxStackDepth];  // private memory; sized for any split of the stack the host may choose
    const WalkStack stack = {lds_stack + lane, kTraceBlock, ta.stack_depth, beyond};
    unsigned char* coop = (unsigned char*)(lds_stack + kTraceBlock * ta.stack_depth);
    enum : uint32_t { DEAD = 0, FRESH = 1, WALK = 2, LEAF = 3, READY = 4 };

    uint32_t next = 0u;  // next undrawn sample of the item (wave-uniform)
    uint32_t st = DEAD;  // DEAD: the lane has no path and takes the next undrawn sample
    unsigned long long rays = 0, paths = 0;
    uint32_t px = 0, py = 0, smp = 0, depth = 0;
    Ray r;
    f3 thr;
    Rng g;
    r.o = r.d = thr = mk3(0.0f, 0.0f, 0.0f);
    g.s0 = g.s1 = 0;
    Walk w;
    walk_begin<AABB_MODE>(sc, w, false, r, RRT_T_MAX);
    StreamProbe probe;
    for (;;) {
        // ---- lanes without a path draw the next samples (≙ main.rs:150-157: the queue of (pixel, sample) work items)
        const unsigned long long idle = RRT_BALLOT(st == DEAD);
        if (idle != 0ull && !exhausted) {
            if (next >= pool) {
                uint32_t chunk = 0, j = 0;
                for (;;) {
                    if (bands_done == 8u) {
                        exhausted = true;
                        break;
                    }
                    const uint32_t band = (xcd + bands_done) & 7u;
                    const uint32_t first = band * band_tiles;
                    const uint32_t count = first < ta.n_t8 ? (ta.n_t8 - first < band_tiles ? ta.n_t8 - first : band_tiles) : 0u;
                    uint32_t c = 0;
                    if (count != 0u) {
                        if (lane == 0) c = atomicAdd(item_counter + band, 1u);
                        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                    }
                    if (count == 0u || c / count >= n_chunks) {
                        ++bands_done;
                        continue;
                    }
                    chunk = c / count;  // a band's items: all its tiles for the first samples, then for the next ...
                    j = first + c % count;
                    if (by_columns) {
                        const uint32_t q = c % count, rows = count / 16u / ta.tiles_x;  // (the last band may have fewer rows)
                        const uint32_t t32 = q >> 4, col = t32 / rows, row = t32 % rows;
                        j = first + ((row * ta.tiles_x + col) << 4) + (q & 15u);
                    }
                    break;
                }
                if (!exhausted) {
                    const uint32_t tile32 = ta.tile_index + (j >> 4) * ta.tile_count;
                    x0 = (tile32 % ta.tiles_x) * 32u + ((j & 3u) << 3);
                    y0 = (tile32 / ta.tiles_x) * 32u + (((j >> 2) & 3u) << 3);
                    s0 = chunk * ta.chunk;
                    pool = (ta.n_samples - s0 < ta.chunk ? ta.n_samples - s0 : ta.chunk) * 64u;
                    next = 0;
                }
            }
            if (!exhausted) {
                const uint32_t idx = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (st == DEAD && idx < pool) {  // (lanes beyond the item's end are served from the next item, next turn)
                    px = x0 + (idx & 7u);
                    py = y0 + ((idx >> 3) & 7u);
                    smp = s0 + (idx >> 6);
                    st = FRESH;
                }
                const uint32_t drawn = (uint32_t)__popcll(idle);
                next = pool - next < drawn ? pool : next + drawn;
            }
        }
        const unsigned long long walking = RRT_BALLOT(st == WALK), paused = RRT_BALLOT(st == LEAF), waiting = RRT_BALLOT(st == READY || st == FRESH);
        if ((walking | paused | waiting) == 0ull) {
            if (exhausted) break;
            continue;  // every lane was beyond the item's end: next item
        }
        if (waiting != 0ull && ((uint32_t)__popcll(waiting) >= ta.shade_lanes || (walking | paused) == 0ull)) {
            bool done = false;  // this lane's path has ended
            bool launch = false;  // this lane has a new ray to trace
            if (st == READY) {
                ++rays;
                f3 radiance;
                done = shade<FEAT>(sc, w.h, r, thr, g, depth, ta.max_depth, ta.bg, &radiance);
                if (done) {
                    f3 c = thr * radiance;
                    float* q = slab + (((size_t)smp * ta.H + py) * ta.W + px) * 3;
                    q[0] = c.x;
                    q[1] = c.y;
                    q[2] = c.z;
                } else {
                    ++depth;
                    launch = true;
                }
            } else if (st == FRESH) {
                if (px < ta.W && py < ta.H) {
                    g = rng_seed(ta.seed, py * ta.W + px, ta.first_sample + smp);
                    r = camera_ray(sc.cam, px, py, ta.W, ta.H, g);
                    thr = mk3(1.0f, 1.0f, 1.0f);
                    depth = 0;
                    ++paths;
                    launch = true;
                } else {
                    done = true;  // a pixel beyond the frame's edge: nothing to trace
                }
            }
            if (launch) {
                walk_begin<AABB_MODE>(sc, w, true, r, RRT_T_MAX);
                st = w.running ? WALK : READY;  // an empty scene answers at once
            }
            if (done) st = DEAD;
            probe.shaded();
            continue;
        }
        // several node steps before the wave looks at its lanes' states again: the bookkeeping of a turn (five ballots, the
        // phase decisions) costs a quarter of a node step
#pragma unroll 1
        for (uint32_t rep = 0; rep < ta.walk_steps; ++rep) {
            if (st == WALK) {
                probe.node_visited();
                walk_step<AABB_MODE>(sc, w, r, RRT_T_MIN, stack);
                if (w.pending) {
                    st = LEAF;
                    probe.leaf_reached(AABB_MODE == kAabbClean ? link_pairs(w.cur) : w.n_pairs);
                } else if (!w.running) st = (AABB_MODE == kAabbClean && w.held != 0u) ? LEAF : READY;  // (a held leaf is still to be tested)
            }
        }
        probe.walked(lane);
        // (clean trees: lanes that hold a leaf count, whether they still walk or not)
        // (clean trees: a phase tests every held leaf; what calls for one are the lanes that cannot take another leaf - two held, or
        // standing at a third, or at the end of their walk)
        const unsigned long long at_leaf = RRT_BALLOT(AABB_MODE == kAabbClean ? w.held != 0u : st == LEAF);
        const unsigned long long pressing = AABB_MODE == kAabbClean ? RRT_BALLOT(w.held2 != 0u || st == LEAF) : at_leaf;
        if (at_leaf != 0ull && ((uint32_t)__popcll(pressing) >= ta.coop_lanes || RRT_BALLOT(st == WALK) == 0ull)) {
            probe.leaf_phase_begins(lane);
            if (AABB_MODE == kAabbClean) {  // every lane's held leaf, see walk_step
                const bool has = w.held != 0u;
                coop_leaves<FEAT, true>(sc, r, RRT_T_MIN, g, depth, has, link_first_pair(w.held), link_pairs(w.held), link_first_pair(w.held2),
                                        w.held2 != 0u ? link_pairs(w.held2) : 0u, w.h, w.hprim, coop);
                if (has) st = walk_after_held(w, stack) ? (w.running ? WALK : LEAF) : READY;
            } else {
                coop_leaves<FEAT, false>(sc, r, RRT_T_MIN, g, depth, st == LEAF, w.first_pair, w.n_pairs, 0u, 0u, w.h, w.hprim, coop);
                if (st == LEAF) {
                    walk_resume(w, stack);
                    st = w.running ? WALK : READY;
                }
            }
            probe.leaf_phase_done();
        }
    }
    probe.store(lane);
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

// The same for deep trees, on the schedule of k_trace_stream without the shading: a wave owns a run of consecutive rays, and a
// lane whose query has ended writes its answer and fetches the next ray of the run at once (FRESH / READY lanes are served
// when 16 of them wait), so no lane idles until the slowest query of the wave is through.
template <int AABB_MODE>
__global__ __launch_bounds__(kTraceBlock) void k_intersect_stream(DScene sc, const float* __restrict__ o, const float* __restrict__ d, size_t n, float t_min,
                                                                  float t_max, uint32_t seed, uint32_t lds_entries, uint32_t run, uint32_t coop_lanes,
                                                                  int* __restrict__ prim, float* __restrict__ t_hit) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * run;
    const uint32_t pool = n - first < (size_t)run ? (uint32_t)(n - first) : run;
    uint32_t beyond[kMaxStackDepth];  // private memory: stack entries past the lds_entries ones kept in LDS
    const WalkStack stack = {lds_stack + lane, kTraceBlock, lds_entries, beyond};
    unsigned char* coop = (unsigned char*)(lds_stack + kTraceBlock * lds_entries);
    enum : uint32_t { DEAD = 0, FRESH = 1, WALK = 2, LEAF = 3, READY = 4 };
    uint32_t next = 64u;  // next unfetched ray of the run (wave-uniform)
    uint32_t idx = lane;
    uint32_t st = idx < pool ? FRESH : DEAD;
    Ray r;
    r.o = r.d = mk3(0.0f, 0.0f, 1.0f);
    Rng g;
    g.s0 = g.s1 = 0;
    Walk w;
    walk_begin<AABB_MODE>(sc, w, false, r, t_max);
    for (;;) {
        const unsigned long long walking = RRT_BALLOT(st == WALK), paused = RRT_BALLOT(st == LEAF), waiting = RRT_BALLOT(st == READY || st == FRESH);
        if ((walking | paused | waiting) == 0ull) break;
        if (waiting != 0ull && ((uint32_t)__popcll(waiting) >= 16u || (walking | paused) == 0ull)) {
            const bool done = st == READY;
            if (done) {
                const size_t i = first + idx;
                prim[i] = w.h.slot >= 0 ? (int)__float_as_uint(sc.slots[3 * w.h.slot].w) : -1;
                t_hit[i] = w.h.t;
            }
            const unsigned long long finished = RRT_BALLOT(done);
            if (done) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(finished >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)finished, 0u));
                idx = next + rank;
                st = idx < pool ? FRESH : DEAD;
            }
            next += (uint32_t)__popcll(finished);
            if (st == FRESH) {
                const size_t i = first + idx;
                r.o = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
                r.d = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
                g = rng_seed(seed, (uint32_t)i, 0);
                walk_begin<AABB_MODE>(sc, w, true, r, t_max);
                st = w.running ? WALK : READY;  // an empty scene answers at once
            }
            continue;
        }
#pragma unroll 1
        for (int rep = 0; rep < kWalkSteps; ++rep) {
            if (st == WALK) {
                walk_step<AABB_MODE>(sc, w, r, t_min, stack);
                if (w.pending) st = LEAF;
                else if (!w.running) st = (AABB_MODE == kAabbClean && w.held != 0u) ? LEAF : READY;
            }
        }
        const unsigned long long at_leaf = RRT_BALLOT(AABB_MODE == kAabbClean ? w.held != 0u : st == LEAF);
        if (at_leaf != 0ull && ((uint32_t)__popcll(at_leaf) >= coop_lanes || RRT_BALLOT(st == WALK) == 0ull)) {
            if (AABB_MODE == kAabbClean) {
                const bool has = w.held != 0u;
                coop_leaves<F_EVERY, true>(sc, r, t_min, g, 0, has, link_first_pair(w.held), link_pairs(w.held), link_first_pair(w.held2),
                                           w.held2 != 0u ? link_pairs(w.held2) : 0u, w.h, w.hprim, coop);
                if (has) st = walk_after_held(w, stack) ? (w.running ? WALK : LEAF) : READY;
            } else {
                coop_leaves<F_EVERY, false>(sc, r, t_min, g, 0, st == LEAF, w.first_pair, w.n_pairs, 0u, 0u, w.h, w.hprim, coop);
                if (st == LEAF) {
                    walk_resume(w, stack);
                    st = w.running ? WALK : READY;
                }
            }
        }
    }
}

}  // namespace

namespace rrt {

static DScene view_of(const SceneData& d) {
    DScene v;
    static_cast<SceneData&>(v) = d;
    return v;
}

hipError_t launch_trace_stream(int mode, uint32_t features, dim3 grid, hipStream_t st, const SceneData& scene, const TraceArgs& ta, float* slab,
                               unsigned long long* counters, uint32_t* item_counter) {
    const DScene sc = view_of(scene);
    const size_t lds = (size_t)kTraceBlock * ta.stack_depth * sizeof(uint32_t) + kCoopBytesPerWave;
    constexpr uint32_t kPlain = F_SPHERE;
#define RRT_GO(MODE)                                                                                                                       \
    do {                                                                                                                                   \
        if (features & F_NESTED) hipLaunchKernelGGL((k_trace_stream<MODE, F_EVERY>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, counters, item_counter); \
        else if (features & ~kPlain) hipLaunchKernelGGL((k_trace_stream<MODE, F_ALL>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, counters, item_counter); \
        else hipLaunchKernelGGL((k_trace_stream<MODE, kPlain>), grid, dim3(kTraceBlock), lds, st, sc, ta, slab, counters, item_counter);                \
    } while (0)
    if (mode == kAabbClean) RRT_GO(kAabbClean);
    else if (mode == RRT_AABB_REFERENCE) RRT_GO(RRT_AABB_REFERENCE);
    else RRT_GO(RRT_AABB_TIGHT);
#undef RRT_GO
    return hipGetLastError();
}

hipError_t launch_intersect_stream(int mode, dim3 grid, hipStream_t st, const SceneData& scene, const float* o, const float* d, size_t n, float t_min,
                                   float t_max, uint32_t seed, uint32_t lds_entries, uint32_t run, uint32_t coop_lanes, int* prim, float* t_hit) {
    const DScene sc = view_of(scene);
    const size_t lds = (size_t)kTraceBlock * lds_entries * sizeof(uint32_t) + kCoopBytesPerWave;
#define RRT_GO(MODE) hipLaunchKernelGGL((k_intersect_stream<MODE>), grid, dim3(kTraceBlock), lds, st, sc, o, d, n, t_min, t_max, seed, lds_entries, run, coop_lanes, prim, t_hit)
    if (mode == kAabbClean) RRT_GO(kAabbClean);
    else if (mode == RRT_AABB_REFERENCE) RRT_GO(RRT_AABB_REFERENCE);
    else RRT_GO(RRT_AABB_TIGHT);
#undef RRT_GO
    return hipGetLastError();
}

#ifdef RRT_PHASE_PROFILE
hipError_t profile_take_stream(uint64_t* out8) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_prof), 64);
    const uint64_t zero[8] = {0};
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, 64);
    return e;
}
#endif

}  // namespace rrt
