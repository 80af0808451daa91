// Scene handles of the C ABI (include/rrt.h): creation (the CPU work is rrt_scene_prepare.cpp; here the device half: context, code
// object, allocations and copies), settings, timing, destruction.
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "rrt_scene_handle.hpp"
#include "rrt_scene_prepare.hpp"

using namespace rrt;

namespace {

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

int rrt::check_device(const rrt_scene* s) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != s->device)
        return rrt_set_error(RRT_ERR_INVALID, "the scene was created on device " + std::to_string(s->device) + " but the current device is " + std::to_string(cur));
    return RRT_OK;
}

extern "C" {

static int scene_create(const rrt_scene_desc* desc, rrt_scene** out) {
    if (!desc || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_create: NULL argument");
    const auto started = std::chrono::steady_clock::now();
    PreparedScene prep;
    if (int rc0 = prepare_scene(desc, &prep)) return rc0;
    int rc = RRT_OK;
    // owns the handle (and through rrt_scene_destroy its device buffers) until the last step has succeeded
    struct Destroy { void operator()(rrt_scene* p) const { rrt_scene_destroy(p); } };
    std::unique_ptr<rrt_scene, Destroy> owner(new rrt_scene());
    rrt_scene* s = owner.get();
    // Where a process's first handle spends its time: the first call that needs the device (a process that has not used the GPU yet
    // creates its context here), then the first access to something in this library's code object (the runtime loads the code object
    // onto the device at that moment: every kernel variant of rrt_kernels_shallow.hip).
    const auto t_prepared = std::chrono::steady_clock::now();
    HIP_TRY(hipGetDevice(&s->device));
    HIP_TRY(hipFree(nullptr));
    const auto t_context = std::chrono::steady_clock::now();
    HIP_TRY(load_shallow_kernels());
    const auto t_code = std::chrono::steady_clock::now();
    t_copy_seconds = 0.0;
#define UP(field, src, count)                                                        \
    do {                                                                             \
        rc = upload(src, count, (const void**)&s->d.field, &s->owned);               \
        if (rc) return rc;                                                           \
    } while (0)
    UP(nodes, (const float4*)prep.nodes.data(), prep.nodes.size() / 4);
    UP(pairs, (const float4*)prep.pairs.data(), prep.pairs.size() / 4);
    UP(plain, (const float4*)prep.plain.data(), prep.plain.size() / 4);
    UP(guards, (const float4*)prep.guards.data(), prep.guards.size() / 4);
    UP(slots, (const float4*)desc->slots, (size_t)desc->n_slots * 3);
    UP(tri_shade, (const float4*)desc->tri_shade, (size_t)desc->n_slots * 4);
    UP(spheres, prep.spheres.data(), prep.spheres.size());
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
    s->d.root_link = prep.root_link;
    s->clean = prep.clean;
    s->deep = prep.deep;
    s->harvest = prep.harvest;
    s->features = prep.features;
    s->stack_depth = prep.stack_depth;
    s->desc_hash = prep.desc_hash;
    s->d.cam = desc->camera;
    s->res_x = desc->resolution_x;
    s->res_y = desc->resolution_y;
    s->samples = desc->samples;
    s->n_textures = desc->n_textures;
    {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, 256));
        s->owned.push_back(p);
        s->d_items = (uint32_t*)p;
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
        s->wave_places = (uint32_t)(cus > 0 ? cus : 256) * 4u * pool_waves_per_simd();
    }
    const auto t_done = std::chrono::steady_clock::now();
    s->upload_seconds = std::chrono::duration<double>(t_done - started).count();
    s->upload_parts[0] = std::chrono::duration<double>(t_context - t_prepared).count();
    s->upload_parts[1] = std::chrono::duration<double>(t_code - t_context).count();
    s->upload_parts[3] = t_copy_seconds;
    s->upload_parts[2] = s->upload_seconds - s->upload_parts[0] - s->upload_parts[1] - s->upload_parts[3];
    *out = owner.release();
    return RRT_OK;
}

int rrt_scene_check(const rrt_scene_desc* desc, uint64_t* layout) {
    return rrt_guard("rrt_scene_check", [&]() -> int {
        if (!desc) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_check: NULL argument");
        PreparedScene prep;
        if (int rc = prepare_scene(desc, &prep)) return rc;
        if (layout) {
            layout[0] = prep.nodes.size() / 16;
            layout[1] = prep.pairs.size() / (4 * kPairQuads);
            layout[2] = prep.plain.size() / 8;
            layout[3] = prep.guards.size() / 4;
            layout[4] = prep.stack_depth;
            layout[5] = prep.features;
            layout[6] = (prep.harvest ? 1u : 0u) | (prep.deep ? 2u : 0u) | (prep.clean ? 4u : 0u);
            layout[7] = prep.desc_hash;
        }
        return RRT_OK;
    });
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
    rrt_multi_state& m = s->multi;
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


}  // extern "C"
