// rrt_render_multi (include/rrt.h): one frame on several GPUs from one process.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rrt_scene_handle.hpp"

using namespace rrt;

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
    rrt_multi_state& m = s->multi;
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
    rrt_multi_state& m0 = scenes[0]->multi;
    // with ACCUMULATE the first device's frame starts from the caller's sums (tiles: every device's frame does, see the workers)
    HIP_TRY(hipEventRecord(ev[0], m0.stream));
    if (accumulate) HIP_TRY(hipMemcpyAsync(m0.d_img, rgb_sum, bytes, hipMemcpyHostToDevice, m0.stream));
    HIP_TRY(hipEventRecord(ev[1], m0.stream));

    // one host thread per handle (main.rs:138-147)
    auto work = [&](uint32_t i) {
        MultiWorker& k = w[i];
        rrt_multi_state& m = k.scene->multi;
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
                if (floats && (e = launch_pack_tiles(m.stream, m.d_img, m.d_pack, floats, W, H, n, i)) != hipSuccess) return fail(e, "k_pack_tiles");
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
        const rrt_multi_state& m = scenes[i]->multi;
        const size_t floats = push_floats(i);
        HIP_TRY(hipStreamWaitEvent(m0.stream, m.landed, 0));
        if (!floats) continue;
        if (tiles) HIP_TRY(launch_unpack_tiles(m0.stream, m0.d_img, m.landing, floats, W, H, n, i));
        else HIP_TRY(launch_add_frame(m0.stream, m0.d_img, m.landing, floats));
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

extern "C" int rrt_render_multi(rrt_scene* const* scenes, uint32_t n, uint32_t shard_mode, const rrt_render_params* p, float* rgb_sum, rrt_stats* stats) {
    return rrt_guard("rrt_render_multi", [&]() -> int { return render_multi(scenes, n, shard_mode, p, rgb_sum, stats); });
}
