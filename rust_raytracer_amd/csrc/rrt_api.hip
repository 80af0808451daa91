// Kernels and the device half of the C ABI (include/rrt.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "rrt_device.hpp"
#include "rrt_error.hpp"

using namespace rrt;

namespace {

constexpr int kBlock = 256;            // 4 waves: a 16x16-pixel quadrant of a 32x32 tile, one 8x8 tile per wave
constexpr uint32_t kMaxStackDepth = 64;  // LDS short stack: entries per lane (16 KiB per block at 64)

struct RenderArgs {
    uint32_t W, H, spp, first_sample, seed, bg, max_depth;
    uint32_t tile_count, tile_index, tiles_x, n_tiles;  // 32x32 tiles
};

// ---- the render kernel -------------------------------------------------------------------------------
// One lane owns one pixel and walks that pixel's samples in order; when a path ends the lane adds its
// radiance to the pixel sum held in registers and immediately starts the next sample (path regeneration),
// so lanes stay busy regardless of how long neighbouring paths are.  Path state (ray, throughput, RNG)
// never leaves registers; the only HBM traffic is the scene (cache resident) and one 12-byte store per pixel.
// ≙ main.rs:200-257 (thread_work) + aggregate.rs:16-105 (trace).
template <int AABB_MODE>
__global__ __launch_bounds__(kBlock) void k_render(DScene sc, RenderArgs ra, float* __restrict__ rgb_sum, unsigned long long* __restrict__ counters) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t tile = ra.tile_index + (blockIdx.x >> 2) * ra.tile_count;  // this block's 32x32 tile
    const uint32_t quad = blockIdx.x & 3u;
    const uint32_t tx = tile % ra.tiles_x, ty = tile / ra.tiles_x;
    const uint32_t x = tx * 32u + (quad & 1u) * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = ty * 32u + (quad >> 1) * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool valid = tile < ra.n_tiles && x < ra.W && y < ra.H;
    uint32_t* stack = lds_stack + tid;

    f3 sum = mk3(0.0f, 0.0f, 0.0f);
    unsigned long long rays = 0;
    uint32_t s = 0;
    bool alive = valid && ra.spp > 0;
    bool fresh = true;
    Ray r;
    f3 thr;
    Rng g;
    uint32_t depth = 0;
    r.o = r.d = thr = mk3(0.0f, 0.0f, 0.0f);
    g.s0 = g.s1 = 0;
    while (alive) {
        if (fresh) {
            g = rng_seed(ra.seed, y * ra.W + x, ra.first_sample + s);
            r = camera_ray(sc.cam, x, y, ra.W, ra.H, g);
            thr = mk3(1.0f, 1.0f, 1.0f);
            depth = 0;
            fresh = false;
        }
        Hit hit = closest_hit<AABB_MODE>(sc, r, RRT_T_MIN, RRT_T_MAX, g, depth, stack, kBlock);
        ++rays;
        f3 radiance;
        if (shade(sc, hit, r, thr, g, depth, ra.max_depth, ra.bg, &radiance)) {
            sum = sum + thr * radiance;  // main.rs:244-245
            ++s;
            fresh = true;
            alive = s < ra.spp;
        } else {
            ++depth;
        }
    }
    if (valid) {
        float* q = rgb_sum + ((size_t)y * ra.W + x) * 3;
        q[0] = sum.x;
        q[1] = sum.y;
        q[2] = sum.z;
    }
    if (counters) {
        unsigned long long paths = valid ? (unsigned long long)ra.spp : 0ull;
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

// ≙ Aggregate::hit (aggregate.rs:118-124) for a batch of rays, one lane per ray.
template <int AABB_MODE>
__global__ __launch_bounds__(kBlock) void k_intersect(DScene sc, const float* __restrict__ o, const float* __restrict__ d, size_t n, float t_min,
                                                      float t_max, uint32_t seed, int* __restrict__ prim, float* __restrict__ t_hit) {
    extern __shared__ uint32_t lds_stack[];
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Rng g = rng_seed(seed, (uint32_t)i, 0);
    Hit h = closest_hit<AABB_MODE>(sc, r, t_min, t_max, g, 0, lds_stack + threadIdx.x, kBlock);
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
        HitProps h = slot_props(sc, slot, r, t, &mat);
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

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int upload(const T* src, size_t count, const void** dst, std::vector<void*>* owned) {
    *dst = nullptr;
    size_t bytes = count * sizeof(T);
    void* p = nullptr;
    // keep every pointer valid (kernels may form addresses for empty arrays but never dereference them)
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    owned->push_back(p);
    if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return RRT_OK;
}

}  // namespace

struct rrt_scene {
    DScene d{};
    std::vector<void*> owned;
    uint32_t res_x = 0, res_y = 0, samples = 0, stack_depth = 1;
    uint32_t n_spheres = 0, n_media = 0, n_btris = 0, n_materials = 0, n_textures = 0;
    int device = 0;
};

static int validate_desc(const rrt_scene_desc* d) {
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
        else BAD("unknown medium boundary kind");
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
    // The traversal stack is sized from the tree itself, never from a number the caller supplies.
    {
        std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, inner ancestors)
        uint32_t deepest = 0;
        if (d->n_nodes) todo.emplace_back(0u, 0u);
        while (!todo.empty()) {
            auto [i, depth] = todo.back();
            todo.pop_back();
            if (depth > deepest) deepest = depth;
            const rrt_node& n = d->nodes[i];
            if (!(n.b & RRT_LEAF_FLAG)) {
                todo.emplace_back(i + 1, depth + 1);
                todo.emplace_back(n.a, depth + 1);
            }
        }
        if (deepest + 1 > kMaxStackDepth) return rrt_set_error(RRT_ERR_LIMIT, "BVH needs more than 64 pending stack entries per ray");
        if (d->bvh_depth < deepest + 1) BAD("bvh_depth is smaller than the tree requires");
    }
#undef BAD
    return RRT_OK;
}

extern "C" {

int rrt_scene_create(const rrt_scene_desc* desc, rrt_scene** out) {
    if (!desc || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_scene_create: NULL argument");
    int rc = validate_desc(desc);
    if (rc) return rc;
    rrt_scene* s = new rrt_scene();
    HIP_TRY(hipGetDevice(&s->device));
#define UP(field, src, count)                                                        \
    do {                                                                             \
        rc = upload(src, count, (const void**)&s->d.field, &s->owned);               \
        if (rc) { rrt_scene_destroy(s); return rc; }                                 \
    } while (0)
    UP(nodes, (const float4*)desc->nodes, (size_t)desc->n_nodes * 2);
    UP(slots, (const float4*)desc->slots, (size_t)desc->n_slots * 3);
    UP(tri_shade, (const float4*)desc->tri_shade, (size_t)desc->n_slots * 3);
    UP(spheres, desc->spheres, desc->n_spheres);
    UP(media, desc->media, desc->n_media);
    UP(btris, (const float4*)desc->boundary_tris, (size_t)desc->n_boundary_tris * 3);
    UP(btri_shade, (const float4*)desc->boundary_tri_shade, (size_t)desc->n_boundary_tris * 3);
    UP(materials, desc->materials, desc->n_materials);
    UP(textures, desc->textures, desc->n_textures);
    UP(texels, desc->texels, desc->texel_bytes);
    UP(lights, desc->lights, desc->n_lights);
#undef UP
    s->d.n_nodes = desc->n_nodes;
    s->d.n_slots = desc->n_slots;
    s->d.n_lights = desc->n_lights;
    s->d.is_list = desc->aggregate_is_list;
    s->d.cam = desc->camera;
    s->res_x = desc->resolution_x;
    s->res_y = desc->resolution_y;
    s->samples = desc->samples;
    s->stack_depth = desc->bvh_depth ? desc->bvh_depth : 1;
    s->n_textures = desc->n_textures;
    *out = s;
    return RRT_OK;
}

void rrt_scene_destroy(rrt_scene* s) {
    if (!s) return;
    for (void* p : s->owned) (void)hipFree(p);
    delete s;
}

int rrt_intersect_batch_device(rrt_scene* s, const float* d_o, const float* d_d, size_t n, float t_min, float t_max, uint32_t aabb_mode,
                               uint32_t seed, int32_t* d_prim, float* d_t, void* stream) {
    if (!s || (n && (!d_o || !d_d || !d_prim || !d_t))) return rrt_set_error(RRT_ERR_INVALID, "rrt_intersect_batch: NULL argument");
    if (aabb_mode > RRT_AABB_REFERENCE) return rrt_set_error(RRT_ERR_INVALID, "rrt_intersect_batch: unknown aabb_mode");
    if (n == 0) return RRT_OK;
    if (n > (size_t)0x7fffffff * kBlock) return rrt_set_error(RRT_ERR_LIMIT, "rrt_intersect_batch: too many rays for one launch");
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    size_t lds = (size_t)kBlock * s->stack_depth * sizeof(uint32_t);
    hipStream_t st = (hipStream_t)stream;
    if (aabb_mode == RRT_AABB_REFERENCE)
        hipLaunchKernelGGL(k_intersect<RRT_AABB_REFERENCE>, grid, dim3(kBlock), lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, d_prim, d_t);
    else
        hipLaunchKernelGGL(k_intersect<RRT_AABB_TIGHT>, grid, dim3(kBlock), lds, st, s->d, d_o, d_d, n, t_min, t_max, seed, d_prim, d_t);
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

static int fill_render_args(rrt_scene* s, const rrt_render_params* p, RenderArgs* ra, unsigned* grid) {
    if (!s || !p) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL argument");
    if (p->aabb_mode > RRT_AABB_REFERENCE) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: unknown aabb_mode");
    if (p->background > RRT_BG_SKY) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: unknown background");
    ra->W = p->width ? p->width : s->res_x;
    ra->H = p->height ? p->height : s->res_y;
    ra->spp = p->samples ? p->samples : s->samples;
    if (ra->W == 0 || ra->H == 0) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: zero resolution");
    if ((uint64_t)ra->W * ra->H > 0xffffffffull) return rrt_set_error(RRT_ERR_LIMIT, "rrt_render: more than 2^32 pixels");
    ra->first_sample = p->first_sample;
    ra->seed = p->seed;
    ra->bg = p->background;
    ra->max_depth = p->max_depth < 0 ? 50u : (uint32_t)p->max_depth;
    ra->tile_count = p->tile_count ? p->tile_count : 1;
    ra->tile_index = p->tile_index;
    if (ra->tile_index >= ra->tile_count) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: tile_index >= tile_count");
    ra->tiles_x = (ra->W + 31) / 32;
    uint32_t tiles_y = (ra->H + 31) / 32;
    ra->n_tiles = ra->tiles_x * tiles_y;
    uint32_t mine = ra->n_tiles > ra->tile_index ? (ra->n_tiles - ra->tile_index + ra->tile_count - 1) / ra->tile_count : 0;
    *grid = mine * 4;
    return RRT_OK;
}

int rrt_render_device(rrt_scene* s, const rrt_render_params* p, float* d_rgb_sum, uint64_t* d_counters, void* stream) {
    RenderArgs ra;
    unsigned grid;
    int rc = fill_render_args(s, p, &ra, &grid);
    if (rc) return rc;
    if (!d_rgb_sum) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL output");
    if (grid == 0) return RRT_OK;
    size_t lds = (size_t)kBlock * s->stack_depth * sizeof(uint32_t);
    hipStream_t st = (hipStream_t)stream;
    if (p->aabb_mode == RRT_AABB_REFERENCE)
        hipLaunchKernelGGL(k_render<RRT_AABB_REFERENCE>, dim3(grid), dim3(kBlock), lds, st, s->d, ra, d_rgb_sum, (unsigned long long*)d_counters);
    else
        hipLaunchKernelGGL(k_render<RRT_AABB_TIGHT>, dim3(grid), dim3(kBlock), lds, st, s->d, ra, d_rgb_sum, (unsigned long long*)d_counters);
    HIP_TRY(hipGetLastError());
    return RRT_OK;
}

int rrt_render(rrt_scene* s, const rrt_render_params* p, float* rgb_sum, rrt_stats* stats) {
    RenderArgs ra;
    unsigned grid;
    int rc = fill_render_args(s, p, &ra, &grid);
    if (rc) return rc;
    if (!rgb_sum) return rrt_set_error(RRT_ERR_INVALID, "rrt_render: NULL output");
    size_t bytes = (size_t)ra.W * ra.H * 3 * sizeof(float);
    float* d_img = nullptr;
    uint64_t* d_cnt = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_img);
        (void)hipFree(d_cnt);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    TRY2(hipMalloc((void**)&d_img, bytes));
    TRY2(hipMalloc((void**)&d_cnt, 16));
    // pixels of tiles this call does not own keep the caller's values
    TRY2(hipMemcpy(d_img, rgb_sum, bytes, hipMemcpyHostToDevice));
    TRY2(hipMemset(d_cnt, 0, 16));
    TRY2(hipEventCreate(&e0));
    TRY2(hipEventCreate(&e1));
    TRY2(hipEventRecord(e0, nullptr));
    rc = rrt_render_device(s, p, d_img, d_cnt, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY2(hipEventRecord(e1, nullptr));
    TRY2(hipEventSynchronize(e1));
    TRY2(hipMemcpy(rgb_sum, d_img, bytes, hipMemcpyDeviceToHost));
    if (stats) {
        uint64_t c[2];
        TRY2(hipMemcpy(c, d_cnt, 16, hipMemcpyDeviceToHost));
        float ms = 0.0f;
        TRY2(hipEventElapsedTime(&ms, e0, e1));
        stats->rays = c[0];
        stats->paths = c[1];
        stats->render_seconds = (double)ms * 1e-3;
    }
    cleanup();
    return RRT_OK;
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

int rrt_probe(rrt_scene* s, uint32_t which, uint32_t arg, const void* in, size_t n, float* out) {
    if (!s || !in || !out) return rrt_set_error(RRT_ERR_INVALID, "rrt_probe: NULL argument");
    static const size_t in_words[5] = {3, 8, 5, 2, 2};
    static const size_t out_words[5] = {6, 14, 3, 5, 8};
    if (which > 4) return rrt_set_error(RRT_ERR_INVALID, "rrt_probe: unknown probe");
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
