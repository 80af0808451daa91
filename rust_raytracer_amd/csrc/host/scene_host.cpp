// Scene construction on the host: shapes, light list, SAH BVH, and the flattened device layout.
#include "scene_host.hpp"
#include "../rrt_bvh_build.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>

#include "../rrt_error.hpp"

namespace rrt_host {

static const float kEpsilon = 1.1920929e-7f;  // f32::EPSILON

static void put3(float* dst, Vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

std::string SceneBuild::add_texture(const rrt_texture& t, const uint8_t* px, int32_t* id) {
    rrt_texture r = t;
    switch (t.type) {
        case RRT_TEX_CONSTANT:
        case RRT_TEX_TEST:
        case RRT_TEX_NOISE:
            break;
        case RRT_TEX_CHECKER:
            if (t.u0 >= textures.size() || t.u1 >= textures.size()) return "Checker child texture index out of range";
            break;
        case RRT_TEX_IMAGE: {
            if (!px || t.u0 == 0 || t.u1 == 0 || (t.u2 != 1 && t.u2 != 3)) return "Image texture needs texels with 1 or 3 channels";
            size_t bytes = (size_t)t.u0 * t.u1 * t.u2;
            size_t off = (texels.size() + 15) & ~(size_t)15;
            if (off + bytes > 0xffffffffull) return "texel storage exceeds 4 GiB";
            texels.resize(off + bytes);
            std::memcpy(texels.data() + off, px, bytes);
            r.u3 = (uint32_t)off;
            break;
        }
        case RRT_TEX_TURBULENCE:  // texture.rs:162-173
            if (t.b > 1.0f) return "Turbulence omega is greater than 1.";
            if (t.b < 0.0f) return "Turbulence omega is less than 0.";
            break;
        default:
            return "Unsupported texture type";
    }
    textures.push_back(r);
    if (id) *id = (int32_t)textures.size() - 1;
    return "";
}

std::string SceneBuild::add_material(uint32_t type, int32_t albedo, int32_t bump, float param, int32_t* id) {
    if (type > RRT_MAT_ISOTROPIC) return "Unsupported material type";
    if (type != RRT_MAT_DIELECTRIC && (albedo < 0 || (size_t)albedo >= textures.size())) return "material texture index out of range";
    if (bump >= (int32_t)textures.size()) return "bump map texture index out of range";
    rrt_material m{type, albedo, bump < 0 ? -1 : bump, param};
    if (type == RRT_MAT_METAL) {  // material.rs:129-135
        if (m.param < 0.0f) m.param = 0.0f;
        else if (m.param > 1.0f) m.param = 1.0f;
    }
    materials.push_back(m);
    if (id) *id = (int32_t)materials.size() - 1;
    return "";
}

static bool make_sphere_record(Vec3 tr, Vec3 rot, Vec3 scl, float radius, uint32_t material, rrt_sphere* out) {
    Mat4 l2w = compose(tr, rot, scl);
    Mat4 w2l;
    if (!invert(l2w, &w2l)) return false;  // shape.rs:56
    std::memset(out, 0, sizeof *out);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            out->w2l[4 * r + c] = w2l.m[r][c];
            out->l2w[4 * r + c] = l2w.m[r][c];
        }
    out->radius = radius;
    out->material = material;
    return true;
}
static Box sphere_box(const rrt_sphere& s) {  // shape.rs:129-139
    Vec3 c = mk(s.l2w[0] * 0.0f + s.l2w[1] * 0.0f + s.l2w[2] * 0.0f + s.l2w[3] * 1.0f,
                s.l2w[4] * 0.0f + s.l2w[5] * 0.0f + s.l2w[6] * 0.0f + s.l2w[7] * 1.0f,
                s.l2w[8] * 0.0f + s.l2w[9] * 0.0f + s.l2w[10] * 0.0f + s.l2w[11] * 1.0f);
    Vec3 r = mk(s.radius, s.radius, s.radius);
    Vec3 a = c - r, b = c + r;
    return Box{min3(a, b), max3(a, b)};
}

std::string SceneBuild::add_sphere(Vec3 tr, Vec3 rot, Vec3 scl, float radius, int32_t material) {
    if (material < 0 || (size_t)material >= materials.size()) return "Missing Material for Sphere.";
    rrt_sphere rec;
    if (!make_sphere_record(tr, rot, scl, radius, (uint32_t)material, &rec)) return "Singular matrix";
    spheres.push_back(rec);
    HostShape s{};
    s.kind = RRT_KIND_SPHERE;
    s.index = (uint32_t)spheres.size() - 1;
    s.material = (uint32_t)material;
    s.box = sphere_box(rec);
    shapes.push_back(s);
    return "";
}

// Triangle::new (shape.rs:225-328) for world-space vertices and resolved texture coordinates.
static std::string finish_triangle(HostShape* t) {
    Vec3 pu = mk(0, 0, 0), pv = mk(0, 0, 0);
    float du02 = t->uv[0] - t->uv[4], dv02 = t->uv[1] - t->uv[5];
    float du12 = t->uv[2] - t->uv[4], dv12 = t->uv[3] - t->uv[5];
    Vec3 dp02 = t->v0 - t->v2;
    Vec3 dp12 = t->v1 - t->v2;
    float det = du02 * dv12 - dv02 * du12;
    bool degenerate = std::fabs(det) < kEpsilon;
    if (!degenerate) {
        float inv = 1.0f / det;
        pu = (dv12 * dp02 - dv02 * dp12) * inv;
        pv = (-du12 * dp02 + du02 * dp12) * inv;
    }
    if (degenerate || squared_length(cross(pu, pv)) == 0.0f) {
        Vec3 ng = cross(t->v2 - t->v0, t->v1 - t->v0);
        if (squared_length(ng) == 0.0f) {
            char buf[256];
            std::snprintf(buf, sizeof buf, "Triangle texture coordinates are degenerate. UV1: (%g, %g), UV2: (%g, %g), UV3: (%g, %g)", t->uv[0],
                          t->uv[1], t->uv[2], t->uv[3], t->uv[4], t->uv[5]);
            return buf;
        }
        ng = normalized(ng);
        if (std::fabs(ng.x) > std::fabs(ng.y)) pu = mk(-ng.z, 0.0f, ng.x) / std::sqrt(ng.x * ng.x + ng.z * ng.z);
        else pu = mk(0.0f, ng.z, -ng.y) / std::sqrt(ng.y * ng.y + ng.z * ng.z);
        pv = cross(ng, pu);
    }
    t->pu = pu;
    t->pv = pv;
    t->box = Box{min3(t->v0, min3(t->v1, t->v2)), max3(t->v0, max3(t->v1, t->v2))};  // shape.rs:430-439
    return "";
}

std::string SceneBuild::add_mesh(Vec3 tr, Vec3 rot, Vec3 scl, uint32_t nv, const double* v, uint32_t nt, const double* tc, uint32_t ntri,
                                 const int32_t* vidx, const int32_t* tidx, bool cull, int32_t material, bool as_boundary) {
    if (material < 0 || (size_t)material >= materials.size()) return "Missing Material for Mesh.";
    Mat4 l2w = compose(tr, rot, scl);
    std::vector<Vec3> world(nv);
    for (uint32_t i = 0; i < nv; ++i) world[i] = xform_point(l2w, mk((float)v[3 * i], (float)v[3 * i + 1], (float)v[3 * i + 2]));
    std::vector<HostShape>& dst = as_boundary ? boundary_shapes : shapes;
    char buf[256];
    for (uint32_t k = 0; k < ntri; ++k) {
        int32_t a = vidx[3 * k], b = vidx[3 * k + 1], c = vidx[3 * k + 2];
        if (nv == 0 || a < 0 || b < 0 || c < 0 || (uint32_t)a > nv - 1 || (uint32_t)b > nv - 1 || (uint32_t)c > nv - 1) {
            std::snprintf(buf, sizeof buf, "Triangle mesh has length %u but attempted to make a Triangle with indices %d, %d, %d.", nv, a, b, c);
            return buf;
        }
        HostShape t{};
        t.kind = RRT_KIND_TRIANGLE;
        t.v0 = world[a]; t.v1 = world[b]; t.v2 = world[c];
        const float dflt[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 1.0f};
        for (int j = 0; j < 3; ++j) {
            int32_t ti = tidx ? tidx[3 * k + j] : -1;
            if (ti >= 0) {
                if ((uint32_t)ti >= nt) {
                    std::snprintf(buf, sizeof buf,
                                  "Triangle texture coordinates have length %u but attempted to make a Triangle with texture index %d.", nt, ti);
                    return buf;
                }
                t.uv[2 * j] = (float)tc[2 * ti];
                t.uv[2 * j + 1] = (float)tc[2 * ti + 1];
            } else {
                t.uv[2 * j] = dflt[2 * j];
                t.uv[2 * j + 1] = dflt[2 * j + 1];
            }
        }
        t.cull = cull;
        t.material = (uint32_t)material;
        std::string e = finish_triangle(&t);
        if (!e.empty()) return e;
        dst.push_back(t);
    }
    return "";
}

std::string SceneBuild::add_medium_sphere(Vec3 tr, Vec3 rot, Vec3 scl, float radius, int32_t boundary_material, float density, int32_t phase) {
    if (phase < 0 || (size_t)phase >= materials.size()) return "Missing Phase Function for ConstantMedium.";
    if (boundary_material < 0 || (size_t)boundary_material >= materials.size()) return "Missing Material for Sphere.";
    rrt_sphere rec;
    if (!make_sphere_record(tr, rot, scl, radius, (uint32_t)boundary_material, &rec)) return "Singular matrix";
    spheres.push_back(rec);
    media.push_back(rrt_medium{RRT_KIND_SPHERE, (uint32_t)spheres.size() - 1, density, (uint32_t)phase});
    HostShape s{};
    s.kind = RRT_KIND_MEDIUM;
    s.index = (uint32_t)media.size() - 1;
    s.material = (uint32_t)phase;
    s.box = sphere_box(rec);  // volume.rs:112-114
    shapes.push_back(s);
    return "";
}

std::string SceneBuild::add_medium_last_boundary_triangle(float density, int32_t phase) {
    if (phase < 0 || (size_t)phase >= materials.size()) return "Missing Phase Function for ConstantMedium.";
    if (boundary_shapes.empty()) return "Only single shapes are allowed for constant mediums at the moment.";
    media.push_back(rrt_medium{RRT_KIND_TRIANGLE, (uint32_t)boundary_shapes.size() - 1, density, (uint32_t)phase});
    HostShape s{};
    s.kind = RRT_KIND_MEDIUM;
    s.index = (uint32_t)media.size() - 1;
    s.material = (uint32_t)phase;
    s.box = boundary_shapes.back().box;
    shapes.push_back(s);
    return "";
}

// scene.rs:588-600 with a ConstantMedium for a boundary: the medium added last stops being a shape of the scene and becomes the
// boundary of a new one (same box: volume.rs:112-114 passes the boundary's on)
std::string SceneBuild::wrap_last_medium(float density, int32_t phase) {
    if (phase < 0 || (size_t)phase >= materials.size()) return "Missing Phase Function for ConstantMedium.";
    if (shapes.empty() || shapes.back().kind != RRT_KIND_MEDIUM) return "Only single shapes are allowed for constant mediums at the moment.";
    const uint32_t inner = shapes.back().index;
    uint32_t chain = 2;
    for (uint32_t m = inner; media[m].boundary_kind == RRT_KIND_MEDIUM; m = media[m].boundary_index) ++chain;
    if (chain > RRT_MAX_MEDIUM_NESTING)
        return "A ConstantMedium inside a ConstantMedium inside ... : this build follows " + std::to_string(RRT_MAX_MEDIUM_NESTING) + " of them at most.";
    media.push_back(rrt_medium{RRT_KIND_MEDIUM, inner, density, (uint32_t)phase});
    shapes.back().index = (uint32_t)media.size() - 1;
    shapes.back().material = (uint32_t)phase;
    return "";
}

void SceneBuild::set_camera(Vec3 pos, Vec3 look_at, Vec3 up, float fov, float aspect, float aperture, float focus) {
    // camera.rs:22-51
    float theta = to_radians(fov);
    float half_height = std::tan(theta / 2.0f);
    float half_width = aspect * half_height;
    Vec3 w = normalized(pos - look_at);
    Vec3 u = normalized(cross(up, w));
    Vec3 v = cross(w, u);
    put3(camera.origin, pos);
    put3(camera.lower_left_corner, pos - (half_width * focus * u) - (half_height * focus * v) - (w * focus));
    put3(camera.horizontal, 2.0f * half_width * focus * u);
    put3(camera.vertical, 2.0f * half_height * focus * v);
    put3(camera.u, u);
    put3(camera.v, v);
    camera.lens_radius = aperture / 2.0f;
}

// ---- SAH BVH (aggregate.rs:193-322) over shape indices ---------------------------------------
namespace {
struct BuildCtx {
    const std::vector<HostShape>* shapes;
    std::vector<uint32_t> order;  // permutation being partitioned in place
    std::vector<rrt_node> nodes;
    std::vector<uint32_t> leaf_order;
    std::vector<Box> suffix;      // scratch
    uint32_t max_pending = 0;
};

void emit_leaf(BuildCtx& cx, const Box& bounds, size_t lo, size_t hi) {
    rrt_node n;
    put3(n.bmin, bounds.lo);
    put3(n.bmax, bounds.hi);
    n.a = (uint32_t)cx.leaf_order.size();
    n.b = RRT_LEAF_FLAG | (uint32_t)(hi - lo);
    for (size_t i = lo; i < hi; ++i) cx.leaf_order.push_back(cx.order[i]);
    cx.nodes.push_back(n);
}

void build_range(BuildCtx& cx, size_t lo, size_t hi, uint32_t inner_depth) {
    const std::vector<HostShape>& sh = *cx.shapes;
    size_t n = hi - lo;
    if (inner_depth > cx.max_pending) cx.max_pending = inner_depth;
    Box total = origin_box();  // every union starts from the point box at the origin (aggregate.rs:195)
    for (size_t i = lo; i < hi; ++i) total = merge(total, sh[cx.order[i]].box);
    if (n <= 2) return emit_leaf(cx, total, lo, hi);
    Box cbox = origin_box();
    for (size_t i = lo; i < hi; ++i) cbox = merge_point(cbox, centre(sh[cx.order[i]].box));
    int axis = widest_axis(cbox);
    if (cbox.hi[axis] == cbox.lo[axis]) return emit_leaf(cx, total, lo, hi);
    std::stable_sort(cx.order.begin() + lo, cx.order.begin() + hi,
                     [&](uint32_t p, uint32_t q) { return centre(sh[p].box)[axis] < centre(sh[q].box)[axis]; });
    // Suffix bounds: the loop (0..n-1).rev() never writes the last entry, which stays the origin point box.
    if (cx.suffix.size() < n) cx.suffix.resize(n);
    cx.suffix[n - 1] = origin_box();
    for (size_t k = n - 1; k-- > 0;) cx.suffix[k] = merge(sh[cx.order[lo + k]].box, cx.suffix[k + 1]);
    Box prefix = origin_box();
    float best = std::numeric_limits<float>::max();
    size_t best_i = 0;
    float total_area = area(total);
    for (size_t i = 0; i + 1 < n; ++i) {
        prefix = merge(prefix, sh[cx.order[lo + i]].box);
        float cost = 1.0f + ((area(prefix) / total_area) * (float)(i + 1)) + ((area(cx.suffix[i + 1]) / total_area) * (float)(n - (i + 1)));
        if (cost < best) {
            best = cost;
            best_i = i;
        }
    }
    if (best < (float)n) {
        size_t node_idx = cx.nodes.size();
        rrt_node inner;
        put3(inner.bmin, total.lo);
        put3(inner.bmax, total.hi);
        inner.a = 0;
        inner.b = (uint32_t)axis;
        cx.nodes.push_back(inner);
        build_range(cx, lo, lo + best_i + 1, inner_depth + 1);
        cx.nodes[node_idx].a = (uint32_t)cx.nodes.size();  // node_idx + right_offset
        build_range(cx, lo + best_i + 1, hi, inner_depth + 1);
        return;
    }
    emit_leaf(cx, total, lo, hi);
}

void fill_tri_slot(const HostShape& s, uint32_t prim, rrt_slot* slot, rrt_tri_shade* sd) {
    put3(slot->v0, s.v0);
    put3(slot->e1, s.v1 - s.v0);
    put3(slot->e2, s.v2 - s.v0);
    slot->prim = prim;
    slot->info = (RRT_KIND_TRIANGLE << 30) | (s.cull ? RRT_INFO_CULL : 0u) | (s.material & 0x00ffffffu);
    slot->aux = 0;
    put3(sd->nrm, normalized(cross(s.v1 - s.v0, s.v2 - s.v0)));
    sd->pad = 0;
    std::memcpy(sd->uv0, s.uv, sizeof(float) * 6);
    put3(sd->pu, s.pu);
    put3(sd->pv, s.pv);
}
}  // namespace

std::string SceneBuild::finalize(int aggregate) {
    const auto started = std::chrono::steady_clock::now();
    struct Stopwatch {  // build_seconds on every way out
        SceneBuild* b;
        std::chrono::steady_clock::time_point t0;
        ~Stopwatch() { b->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } stopwatch{this, started};
    if (materials.size() > 0x00ffffffu) return "too many materials";
    const bool bvh = aggregate != RRT_AGGREGATE_LIST;
    BuildCtx cx;
    cx.shapes = &shapes;
    cx.order.resize(shapes.size());
    std::iota(cx.order.begin(), cx.order.end(), 0u);
    if (aggregate == RRT_AGGREGATE_BVH_GPU || aggregate == RRT_AGGREGATE_BVH_CLEAN) {
        // the same tree, built level by level on the current device (rrt_bvh_build.hip)
        std::vector<float> boxes(shapes.size() * 6);
        for (size_t i = 0; i < shapes.size(); ++i) {
            put3(&boxes[i * 6], shapes[i].box.lo);
            put3(&boxes[i * 6 + 3], shapes[i].box.hi);
        }
        if (rrt_bvh_build_gpu(boxes.data(), (uint32_t)shapes.size(), aggregate == RRT_AGGREGATE_BVH_CLEAN, &cx.nodes, &cx.leaf_order, &cx.max_pending) != RRT_OK)
            return rrt_last_error();
    } else if (bvh) {
        if (!shapes.empty()) build_range(cx, 0, shapes.size(), 0);  // an empty BVH has no nodes (aggregate.rs:339-341)
    } else {
        // "List" (scene.rs:610): one leaf, scene order, no bounding-box test
        Box total = origin_box();
        for (const HostShape& s : shapes) total = merge(total, s.box);
        emit_leaf(cx, total, 0, shapes.size());
    }
    nodes = cx.nodes;

    slots.assign(cx.leaf_order.size(), rrt_slot{});
    tri_shade.assign(cx.leaf_order.size(), rrt_tri_shade{});
    for (size_t i = 0; i < cx.leaf_order.size(); ++i) {
        uint32_t prim = cx.leaf_order[i];
        const HostShape& s = shapes[prim];
        if (s.kind == RRT_KIND_TRIANGLE) {
            fill_tri_slot(s, prim, &slots[i], &tri_shade[i]);
        } else {
            slots[i].prim = prim;
            slots[i].info = (s.kind << 30) | (s.material & 0x00ffffffu);
            slots[i].aux = s.index;
        }
    }
    boundary_tris.assign(boundary_shapes.size(), rrt_slot{});
    boundary_tri_shade.assign(boundary_shapes.size(), rrt_tri_shade{});
    for (size_t i = 0; i < boundary_shapes.size(); ++i) fill_tri_slot(boundary_shapes[i], 0xffffffffu, &boundary_tris[i], &boundary_tri_shade[i]);

    // Light mixture members (scene.rs:130-140): DiffuseLight, Metal, Dielectric are "important".
    lights.clear();
    if (use_importance) {
        for (const HostShape& s0 : shapes) {
            uint32_t mt = materials[s0.material].type;
            if (!(mt == RRT_MAT_METAL || mt == RRT_MAT_DIELECTRIC || mt == RRT_MAT_DIFFUSE_LIGHT)) continue;
            const HostShape* s = &s0;
            rrt_light l{};
            if (s->kind == RRT_KIND_MEDIUM) {  // pdf / random_dir_towards delegate to the boundary (volume.rs:116-122)
                uint32_t mi = s->index;
                while (media[mi].boundary_kind == RRT_KIND_MEDIUM) mi = media[mi].boundary_index;
                const rrt_medium& m = media[mi];
                if (m.boundary_kind == RRT_KIND_SPHERE) {
                    l.kind = RRT_KIND_SPHERE;
                    l.index = m.boundary_index;
                    lights.push_back(l);
                    continue;
                }
                s = &boundary_shapes[m.boundary_index];
            }
            if (s->kind == RRT_KIND_SPHERE) {
                l.kind = RRT_KIND_SPHERE;
                l.index = s->index;
            } else {
                l.kind = RRT_KIND_TRIANGLE;
                l.cull = s->cull ? 1u : 0u;
                Vec3 e1 = s->v1 - s->v0, e2 = s->v2 - s->v0;
                put3(l.v0, s->v0); put3(l.v1, s->v1); put3(l.v2, s->v2);
                put3(l.e1, e1); put3(l.e2, e2);
                put3(l.nrm, normalized(cross(e1, e2)));
                l.area = 0.5f * length(cross(e1, e2));
            }
            lights.push_back(l);
        }
    }

    desc = rrt_scene_desc{};
    desc.n_nodes = (uint32_t)nodes.size(); desc.nodes = nodes.data();
    desc.n_slots = (uint32_t)slots.size(); desc.slots = slots.data(); desc.tri_shade = tri_shade.data();
    desc.n_spheres = (uint32_t)spheres.size(); desc.spheres = spheres.data();
    desc.n_media = (uint32_t)media.size(); desc.media = media.data();
    desc.n_boundary_tris = (uint32_t)boundary_tris.size(); desc.boundary_tris = boundary_tris.data(); desc.boundary_tri_shade = boundary_tri_shade.data();
    desc.n_materials = (uint32_t)materials.size(); desc.materials = materials.data();
    desc.n_textures = (uint32_t)textures.size(); desc.textures = textures.data();
    desc.texel_bytes = texels.size(); desc.texels = texels.data();
    desc.n_lights = (uint32_t)lights.size(); desc.lights = lights.data();
    desc.camera = camera;
    desc.resolution_x = res_x; desc.resolution_y = res_y; desc.samples = samples;
    desc.bvh_depth = cx.max_pending + 1;
    desc.aggregate_kind = aggregate == RRT_AGGREGATE_BVH_GPU ? (uint32_t)RRT_AGGREGATE_BVH : (uint32_t)aggregate;
    finalized = true;
    return "";
}

}  // namespace rrt_host

// ---- C ABI ------------------------------------------------------------------------------------
using rrt_host::SceneBuild;
using rrt_host::Vec3;
struct rrt_host_scene {
    SceneBuild b;
};
static Vec3 v3p(const float* p) { return rrt_host::mk(p[0], p[1], p[2]); }
static int fail_scene(const std::string& e) { return rrt_set_error(RRT_ERR_SCENE, e); }
#define RRT_NEED(cond)                                                  \
    do {                                                                \
        if (!(cond)) return rrt_set_error(RRT_ERR_INVALID, "invalid argument: " #cond); \
    } while (0)

extern "C" {

int rrt_host_scene_new(rrt_host_scene** out) {
    RRT_NEED(out);
    return rrt_guard("rrt_host_scene_new", [&]() -> int {
        *out = new rrt_host_scene();
        return RRT_OK;
    });
}
void rrt_host_scene_destroy(rrt_host_scene* s) { delete s; }

int rrt_host_scene_load(const char* path, uint32_t ow, uint32_t oh, uint32_t os, float oaspect, rrt_host_scene** out) {
    return rrt_host_scene_load_as(path, ow, oh, os, oaspect, RRT_AGGREGATE_BVH, out);
}
int rrt_host_scene_load_as(const char* path, uint32_t ow, uint32_t oh, uint32_t os, float oaspect, int tree, rrt_host_scene** out) {
    RRT_NEED(path && out);
    RRT_NEED(tree == RRT_AGGREGATE_BVH || tree == RRT_AGGREGATE_BVH_GPU || tree == RRT_AGGREGATE_BVH_CLEAN);
    return rrt_guard("rrt_host_scene_load", [&]() -> int {
        std::unique_ptr<rrt_host_scene> s(new rrt_host_scene());
        std::string e = rrt_host::load_scene_file(path, ow, oh, os, oaspect, tree, &s->b);
        if (!e.empty()) return fail_scene(e);
        *out = s.release();
        return RRT_OK;
    });
}
int rrt_host_add_texture(rrt_host_scene* s, const rrt_texture* t, const uint8_t* texels, int32_t* id) {
    RRT_NEED(s && t);
    return rrt_guard("rrt_host_add_texture", [&]() -> int {
        std::string e = s->b.add_texture(*t, texels, id);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_add_material(rrt_host_scene* s, uint32_t type, int32_t albedo, int32_t bump, float param, int32_t* id) {
    RRT_NEED(s);
    return rrt_guard("rrt_host_add_material", [&]() -> int {
        std::string e = s->b.add_material(type, albedo, bump, param, id);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_add_sphere(rrt_host_scene* s, const float* tr, const float* rot, const float* scl, float radius, int32_t material) {
    RRT_NEED(s && tr && rot && scl);
    return rrt_guard("rrt_host_add_sphere", [&]() -> int {
        std::string e = s->b.add_sphere(v3p(tr), v3p(rot), v3p(scl), radius, material);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_add_mesh(rrt_host_scene* s, const float* tr, const float* rot, const float* scl, uint32_t nv, const double* v, uint32_t nt,
                      const double* t, uint32_t ntri, const int32_t* vidx, const int32_t* tidx, int cull, int32_t material) {
    RRT_NEED(s && tr && rot && scl && (nv == 0 || v) && (ntri == 0 || vidx));
    return rrt_guard("rrt_host_add_mesh", [&]() -> int {
        std::string e = s->b.add_mesh(v3p(tr), v3p(rot), v3p(scl), nv, v, nt, t, ntri, vidx, tidx, cull != 0, material, false);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_add_medium_sphere(rrt_host_scene* s, const float* tr, const float* rot, const float* scl, float radius, int32_t bmat, float density,
                               int32_t phase) {
    RRT_NEED(s && tr && rot && scl);
    return rrt_guard("rrt_host_add_medium_sphere", [&]() -> int {
        std::string e = s->b.add_medium_sphere(v3p(tr), v3p(rot), v3p(scl), radius, bmat, density, phase);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_wrap_last_medium(rrt_host_scene* s, float density, int32_t phase) {
    RRT_NEED(s);
    return rrt_guard("rrt_host_wrap_last_medium", [&]() -> int {
        std::string e = s->b.wrap_last_medium(density, phase);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_set_camera(rrt_host_scene* s, const float* pos, const float* look, const float* up, float fov, float aspect, float aperture,
                        float focus) {
    RRT_NEED(s && pos && look && up);
    s->b.set_camera(v3p(pos), v3p(look), v3p(up), fov, aspect, aperture, focus);
    return RRT_OK;
}
int rrt_host_set_logistics(rrt_host_scene* s, uint32_t rx, uint32_t ry, uint32_t samples, int use_is) {
    RRT_NEED(s);
    s->b.res_x = rx; s->b.res_y = ry; s->b.samples = samples; s->b.use_importance = use_is != 0;
    return RRT_OK;
}
int rrt_host_finalize(rrt_host_scene* s, int aggregate) {
    RRT_NEED(s);
    RRT_NEED(aggregate >= RRT_AGGREGATE_LIST && aggregate <= RRT_AGGREGATE_BVH_CLEAN);
    return rrt_guard("rrt_host_finalize", [&]() -> int {
        std::string e = s->b.finalize(aggregate);
        return e.empty() ? RRT_OK : fail_scene(e);
    });
}
int rrt_host_scene_times(const rrt_host_scene* s, double* load_seconds, double* build_seconds) {
    RRT_NEED(s);
    if (load_seconds) *load_seconds = s->b.load_seconds;
    if (build_seconds) *build_seconds = s->b.build_seconds;
    return RRT_OK;
}
const rrt_scene_desc* rrt_host_desc(const rrt_host_scene* s) {
    if (!s || !s->b.finalized) {
        rrt_set_error(RRT_ERR_INVALID, "scene is not finalized");
        return nullptr;
    }
    return &s->b.desc;
}

}  // extern "C"
