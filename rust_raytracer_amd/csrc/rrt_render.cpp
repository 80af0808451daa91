// Rendering through the C ABI (include/rrt.h): pass planning, the passes of one render on one device, the host-buffer wrappers, ray
// batches, the 8-bit resolve, the unit probes.
#include <cstring>
#include <string>
#include <vector>

#include "rrt_scene_handle.hpp"

using namespace rrt;

extern "C" {

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
        const dim3 grid((unsigned)((n + run - 1) / run));
        // deep trees: the reference's test alone IS exhaustive (leaves are tested whole)
        const int mode = s->clean ? kAabbClean : (aabb_mode != RRT_AABB_TIGHT ? (int)RRT_AABB_REFERENCE : (int)RRT_AABB_TIGHT);
        HIP_TRY(launch_intersect_stream(mode, grid, st, s->d, d_o, d_d, n, t_min, t_max, seed, lds_entries, run, s->clean ? 2u : 24u, d_prim, d_t));
    } else {
        HIP_TRY(launch_intersect(aabb_mode == RRT_AABB_EXHAUSTIVE ? kAabbExhaustive : (int)aabb_mode, st, s->d, d_o, d_d, n, t_min, t_max, seed, s->stack_depth, d_prim, d_t));
    }
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
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_o, n * 12));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_d, n * 12));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_t, n * 4));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_p, n * 4));
    HIP_TRY_OR(cleanup, hipMemcpy(d_o, o, n * 12, hipMemcpyHostToDevice));
    HIP_TRY_OR(cleanup, hipMemcpy(d_d, d, n * 12, hipMemcpyHostToDevice));
    rc = rrt_intersect_batch_device(s, d_o, d_d, n, t_min, t_max, aabb_mode, seed, d_p, d_t, nullptr);
    if (rc) { cleanup(); return rc; }
    HIP_TRY_OR(cleanup, hipDeviceSynchronize());
    HIP_TRY_OR(cleanup, hipMemcpy(prim, d_p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY_OR(cleanup, hipMemcpy(t_hit, d_t, n * 4, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

}  // extern "C"

namespace {

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

}  // namespace

// Enqueues the passes of one render.  static_staging = false: the slab is the scene handle's own buffer, (re)allocated here when
// the frame needs more; true: the slab is the current device's copy of the built-in staging area (rrt_noalloc_area.hip) and nothing
// here allocates, frees or creates anything (no event is recorded either: the per-kernel timing of rrt_scene_set_timing does not cover these renders).
int rrt::render_passes(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream, bool static_staging) {
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
    // Lanes at a leaf / finished lanes that trigger a leaf / shading phase.  Reference trees (a lane waits at its leaf), measured on a
    // 500 000-triangle mesh, 1024x1024 @ 64 spp: 8/16 -> 172 ms, 16/16 -> 147 ms, 24/16 -> 140 ms, 32/16 -> 143 ms, 16/8 -> 154 ms,
    // 16/32 -> 152 ms.  Clean trees (a lane keeps up to two leaves and walks on, see walk_step): the number counts the lanes that cannot
    // take another leaf (two held, or at the end of their walk with one).  About four walks end per turn of the wave's loop, so nearly
    // every turn has a phase whatever the number is, and making those lanes wait costs more than fuller phases save: 10 M triangles
    // at 3840x2160 @ 16: 1 -> 319.0 ms, 2 -> 321.1, 4 -> 320.2, 8 -> 321.6, 16 -> 335.5 (one held leaf, phases at two holders: 341).
    ta.coop_lanes = s->clean ? 1 : 24;
    ta.shade_lanes = 16;
    // Node steps per turn of k_trace_stream's loop (the bookkeeping of a turn - the ballots, the phase decisions - costs a quarter of a
    // node step).  10 M triangles at 3840x2160@8 / 500 000 triangles at 1024x1024@32, persistent grid: 2 -> 275.9 / 57.4 ms, 3 -> 246.8 /
    // 50.7, 4 -> 241.2 / 48.6, 6 -> 233.4 / 48.6, 8 -> 231.1 / 51.0.  (Shading at 8 / 16 / 24 / 32 finished lanes: 244.6 / 241.2 / 237.6 /
    // 243.5 and 49.4 / 48.6 / 49.6 / 52.5 ms.)
    ta.walk_steps = 6;
    hipStream_t st = (hipStream_t)stream;
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
        const bool pooled = !exhaustive && s->pooling && !s->deep && pool_takes(ta.max_depth, ta.n_samples, s->d.n_lights, s->d.n_slots);
        const uint64_t want = s->deep ? 655360 : (pooled ? 163840 : 40960);
        const uint32_t least = s->deep ? 1u : (pooled ? 4u : 16u);
        ta.chunk = kChunk;
        while (ta.chunk > least && (uint64_t)((ta.n_samples + ta.chunk - 1) / ta.chunk) * ta.n_t8 < want) ta.chunk /= 2u;
        const uint64_t items = (uint64_t)((ta.n_samples + ta.chunk - 1) / ta.chunk) * ta.n_t8;
        if (items > 0x7fffffffull) return rrt_set_error(RRT_ERR_LIMIT, "rrt_render: too many work items in one pass");
        const dim3 grid((unsigned)items);
        float* const pass_slab = static_staging ? slab : s->slab;
        if (s->deep) {  // (clean trees are deep by definition)
            const int mode = s->clean ? kAabbClean : (p->aabb_mode != RRT_AABB_TIGHT ? (int)RRT_AABB_REFERENCE : (int)RRT_AABB_TIGHT);
            ta.n_items = (uint32_t)items;
            const dim3 sgrid((unsigned)(items < s->wave_places ? items : s->wave_places));  // one round of the chip, see k_trace_stream
            HIP_TRY(hipMemsetAsync(s->d_items, 0, 8 * sizeof(uint32_t), st));
            HIP_TRY(launch_trace_stream(mode, need, sgrid, st, s->d, ta, pass_slab, cnt, s->d_items));
        } else if (pooled) {
            // shallow trees: the pooled schedule (k_trace_pool) - a persistent grid that hands the items out itself
            ta.n_items = (uint32_t)items;
            const dim3 pgrid((unsigned)(items < s->wave_places ? items : s->wave_places));
            HIP_TRY(hipMemsetAsync(s->d_items, 0, sizeof(uint32_t), st));
            HIP_TRY(launch_trace_pool((int)p->aabb_mode, need, s->harvest, pgrid, st, s->d, ta, pass_slab, cnt, s->d_items));
        } else {
            HIP_TRY(launch_trace(exhaustive ? kAabbExhaustive : (int)p->aabb_mode, need, s->harvest, grid, st, s->d, ta, pass_slab, cnt));
        }
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        const int first_pass = (done == 0 && !(p->flags & RRT_RENDER_ACCUMULATE)) ? 1 : 0;
        HIP_TRY(launch_accumulate(st, pass_slab, ta, first_pass, d_rgb_sum));
        if (ev) HIP_TRY(hipEventRecord(ev[2], st));
    }
    return RRT_OK;
}

int rrt::passes_since(rrt_scene* s, size_t mark, bool keep, double* trace_s, double* accumulate_s) {
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

extern "C" {

int rrt_render_device(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream) {
    return render_passes(s, p, d_rgb_sum, d_counters, stream, false);
}

// The same render with the slab in the built-in staging area.  Every (sample, pixel) record a pass reads was written by that
// pass's trace kernel (pixels of other ranks' tiles and beyond the frame's edge are neither written nor read), so nothing in the
// area has to start at zero and the previous call's contents are simply overwritten.
int rrt_render_device_noalloc(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream) {
    return render_passes(s, p, d_rgb_sum, d_counters, stream, true);
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
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_img, bytes));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_cnt, 16));
    for (hipEvent_t& e : ev) HIP_TRY_OR(cleanup, hipEventCreate(&e));
    HIP_TRY_OR(cleanup, hipEventRecord(ev[0], nullptr));
    // pixels of tiles this call does not own, and sums a call with RRT_RENDER_ACCUMULATE adds to, are the caller's values
    const bool keeps_values = (p->flags & RRT_RENDER_ACCUMULATE) != 0 || plan.ta.tile_count > 1;
    if (keeps_values) HIP_TRY_OR(cleanup, hipMemcpyAsync(d_img, rgb_sum, bytes, hipMemcpyHostToDevice, nullptr));
    HIP_TRY_OR(cleanup, hipMemsetAsync(d_cnt, 0, 16, nullptr));
    HIP_TRY_OR(cleanup, hipEventRecord(ev[1], nullptr));
    s->timing = stats != nullptr || user_timing;
    rc = rrt_render_device(s, p, d_img, d_cnt, nullptr);
    if (rc) { cleanup(); return rc; }
    HIP_TRY_OR(cleanup, hipEventRecord(ev[2], nullptr));
    HIP_TRY_OR(cleanup, hipMemcpyAsync(rgb_sum, d_img, bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY_OR(cleanup, hipEventRecord(ev[3], nullptr));
    HIP_TRY_OR(cleanup, hipEventSynchronize(ev[3]));
    if (stats) {
        uint64_t c[2];
        HIP_TRY_OR(cleanup, hipMemcpy(c, d_cnt, 16, hipMemcpyDeviceToHost));
        float ms_in = 0.0f, ms = 0.0f, ms_out = 0.0f;
        HIP_TRY_OR(cleanup, hipEventElapsedTime(&ms_in, ev[0], ev[1]));
        HIP_TRY_OR(cleanup, hipEventElapsedTime(&ms, ev[1], ev[2]));
        HIP_TRY_OR(cleanup, hipEventElapsedTime(&ms_out, ev[2], ev[3]));
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

int rrt_resolve_rgb8_device(const float* d_sum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* d_out, void* stream) {
    if (!d_sum || !d_out) return rrt_set_error(RRT_ERR_INVALID, "rrt_resolve_rgb8: NULL argument");
    size_t n = (size_t)W * H * 3;
    if (n == 0) return RRT_OK;
    HIP_TRY(launch_resolve((hipStream_t)stream, d_sum, n, (float)spp, d_out));
    return RRT_OK;
}

int rrt_resolve_rgb8(const float* rgb_sum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* out) {
    if (!rgb_sum || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_resolve_rgb8: NULL argument");
    size_t n = (size_t)W * H * 3;
    if (n == 0) return RRT_OK;
    float* d_sum = nullptr;
    uint8_t* d_out = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_sum); (void)hipFree(d_out); };
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_sum, n * 4));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_out, n));
    HIP_TRY_OR(cleanup, hipMemcpy(d_sum, rgb_sum, n * 4, hipMemcpyHostToDevice));
    int rc = rrt_resolve_rgb8_device(d_sum, W, H, spp, d_out, nullptr);
    if (rc) { cleanup(); return rc; }
    HIP_TRY_OR(cleanup, hipDeviceSynchronize());
    HIP_TRY_OR(cleanup, hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

int rrt_selftest_reciprocal(uint64_t* mismatches) {
    if (!mismatches) return rrt_set_error(RRT_ERR_INVALID, "rrt_selftest_reciprocal: NULL argument");
    unsigned long long* d = nullptr;
    auto cleanup = [&]() { (void)hipFree(d); };
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d, 8));
    HIP_TRY_OR(cleanup, hipMemset(d, 0, 8));
    HIP_TRY_OR(cleanup, launch_selftest_rcp(d));
    HIP_TRY_OR(cleanup, hipDeviceSynchronize());
    HIP_TRY_OR(cleanup, hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost));
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
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_in, n * in_words[which] * 4));
    HIP_TRY_OR(cleanup, hipMalloc((void**)&d_out, n * out_words[which] * 4));
    HIP_TRY_OR(cleanup, hipMemcpy(d_in, in, n * in_words[which] * 4, hipMemcpyHostToDevice));
    HIP_TRY_OR(cleanup, launch_probe(s->d, which, arg, s->res_x, s->res_y, d_in, n, d_out));
    HIP_TRY_OR(cleanup, hipDeviceSynchronize());
    HIP_TRY_OR(cleanup, hipMemcpy(out, d_out, n * out_words[which] * 4, hipMemcpyDeviceToHost));
    cleanup();
    return RRT_OK;
}

}  // extern "C"
