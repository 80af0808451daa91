// The CPU half of rrt_scene_create (see rrt_scene_prepare.hpp).
#include "rrt_scene_prepare.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rrt_error.hpp"
#include "rrt_launch.hpp"

namespace rrt {
namespace {

int validate_desc(const rrt_scene_desc* d, uint32_t* tree_depth) {
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
bool is_shallow(const rrt_scene_desc* d) { return d->aggregate_kind != RRT_AGGREGATE_BVH_CLEAN && d->n_nodes <= kDeepTreeNodes; }
// across: the padding of every box face (deep trees), and in shallow trees the padding of a triangle's box ACROSS the triangle's
// plane; along: shallow trees, the padding of a triangle's box along its plane.  The wide padding goes along the plane only: a wall's
// box stays a thin slab, which a ray that leaves the wall has left before its span begins.  It is a fraction of the radius of the
// sphere around the scene's triangles (not of the scene's extent: a small mesh on a huge ground sphere keeps boxes of its own size).
struct Pads { float across, along; };
// Centre and radius of a sphere around every triangle vertex of the scene (the leaves' triangles and those that bound media);
// false when the scene has no triangle.
bool triangle_sphere(const rrt_scene_desc* d, double centre[3], double* radius) {
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
Pads box_pads(const rrt_scene_desc* d) {
    float scene_max = 0.0f;
    for (uint32_t i = 0; i < d->n_nodes; ++i)
        for (int c = 0; c < 3; ++c) scene_max = std::fmax(scene_max, std::fmax(std::fabs(d->nodes[i].bmin[c]), std::fabs(d->nodes[i].bmax[c])));
    for (int c = 0; c < 3; ++c) scene_max = std::fmax(scene_max, std::fabs(d->camera.origin[c]));  // ray origins: the camera or a surface
    if (!is_shallow(d)) return {1.0e-5f * scene_max + 1.0e-30f, 1.0e-5f * scene_max + 1.0e-30f};
    double centre[3], radius = 0.0;
    if (!triangle_sphere(d, centre, &radius)) radius = 0.0;
    return {2.0e-5f * scene_max + 1.0e-30f, (float)(RRT_SHALLOW_PAD * radius) + 2.0e-5f * scene_max + 1.0e-30f};
}
void pair_box(const rrt_scene_desc* d, uint32_t first, uint32_t count, Pads pads, float* b);

// The table of the grazing guard (layout and meaning: grazes_a_plane in rrt_traversal.hpp): a sphere around all triangle vertices
// of the scene (the leaves' triangles and those that bound media), and one scaled normal per distinct plane direction.
std::vector<float> guard_table(const rrt_scene_desc* d, Pads pads) {
    struct Plane { double n[3], scale; };
    std::vector<Plane> planes;
    std::vector<const rrt_slot*> tris;
    for (uint32_t i = 0; i < d->n_slots; ++i)
        if (RRT_INFO_KIND(d->slots[i].info) == RRT_KIND_TRIANGLE) tris.push_back(&d->slots[i]);
    for (uint32_t i = 0; i < d->n_boundary_tris; ++i) tris.push_back(&d->boundary_tris[i]);
    auto put_u = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    const float inf = std::numeric_limits<float>::infinity();
    if (tris.empty()) return {0.0f, 0.0f, 0.0f, 0.0f, inf, inf, inf, put_u(0u), inf, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // nothing to graze
    double centre_d[3], radius = 0.0;
    triangle_sphere(d, centre_d, &radius);
    const float centre[3] = {(float)centre_d[0], (float)centre_d[1], (float)centre_d[2]};
    const float rho = std::nextafter((float)radius, inf);
    // how far from that sphere's centre a ray of the render kernels can start: at the camera (camera.rs:54-56: within the lens) or
    // on a primitive
    double start2 = 0.0;
    for (int c = 0; c < 3; ++c) start2 += ((double)d->camera.origin[c] - centre_d[c]) * ((double)d->camera.origin[c] - centre_d[c]);
    const double start_camera = std::sqrt(start2) + 2.0 * std::fabs((double)d->camera.lens_radius);
    double start = 0.0;  // on a primitive
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
    while ((general.size() / 4) % 2 != 0) {  // (the device takes two per turn)
        const float inert[4] = {inf, 0.0f, 0.0f, 0.0f};  // d . G is infinite, or not a number: no ray is close to it
        general.insert(general.end(), inert, inert + 4);
    }
    // The render kernels take ONE reach for all their rays when that costs little: when the farthest start is within four radii of
    // the triangles' sphere (the Cornell boxes).  A small mesh in a wide scene (a cube on a ground sphere of a hundred times its
    // size) would have nearly every ray "graze" with the far ground's reach; there every ray works out its own.
    // Two of them: camera rays start at the camera (often far outside the scene: the Cornell boxes' stands 1.7 box widths in front),
    // all later rays on a primitive, and the shorter reach of those lets a third fewer of them count as grazing.
    const double render_reach = 1.001 * (std::fmax(start, start_camera) + (double)rho), bounce_reach = 1.001 * (start + (double)rho);
    const bool own_reach = !(render_reach <= 4.0 * (double)rho) || !(render_reach <= longest_reach);
    const float render_reach2 = std::nextafter((float)(render_reach * render_reach * 1.0001), inf);
    const float bounce_reach2 = std::nextafter((float)(bounce_reach * bounce_reach * 1.0001), inf);
    std::vector<float> out = {centre[0], centre[1], centre[2], rho, axis[0], axis[1], axis[2], put_u((uint32_t)(general.size() / 4)),
                              (float)longest_reach, render_reach2, put_u(own_reach ? 1u : 0u), bounce_reach2};
    if (general.empty()) out.insert(out.end(), 8, 0.0f);  // (room for the two entries the device fetches with the header)
    for (size_t k = 0; k + 8 <= general.size(); k += 8) {  // two entries side by side: (x, x, y, y), (z, z, -, -)
        const float* a = &general[k];
        const float* b = &general[k + 4];
        const float two[8] = {a[0], b[0], a[1], b[1], a[2], b[2], 0.0f, 0.0f};
        out.insert(out.end(), two, two + 8);
    }
    return out;
}
// b = (lo.xyz, -, hi.xyz, -)
void grow_point(float* b, const float* p) {
    for (int c = 0; c < 3; ++c) {
        b[c] = std::fmin(b[c], p[c]);
        b[4 + c] = std::fmax(b[4 + c], p[c]);
    }
}
void grow_sphere(float* b, const rrt_sphere& sp) {
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
void grow_triangle(float* b, const rrt_slot& t, Pads pads) {
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
void pair_box(const rrt_scene_desc* d, uint32_t first, uint32_t count, Pads pads, float* b) {
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
std::vector<float> tight_boxes(const rrt_scene_desc* d) {
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


}  // namespace

int prepare_scene(const rrt_scene_desc* desc, PreparedScene* out, unsigned max_threads) {
    *out = PreparedScene();
    uint32_t tree_depth = 1;
    if (int rc = validate_desc(desc, &tree_depth)) return rc;
    uint32_t clean_depth = 0;  // clean trees: most stack entries a walk can have pending (sizes the traversal stack)
    // Device copy of the tree: leaves reference pair records (two primitives interleaved component-wise plus their box,
    // 128 bytes).  Which primitives share a record is free (every comparison on the device orders hits by distance, then
    // slot), so each leaf's triangles are matched greedily into the pairs whose common box is smallest: the sifting
    // traversal tests a pair only when the ray overlaps that box.  Spheres and media get records of their own.
    {
        std::vector<rrt_node> dn(desc->nodes, desc->nodes + desc->n_nodes);
        std::vector<float> pairs;
        const Pads pad = box_pads(desc);
        auto put_u = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
        // Harvesting (see HARVEST in rrt_traversal.hpp) needs a single leaf and every light to be one of its primitives:
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
        out->harvest = harvest;
        const bool clean = desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN;
        auto half_area = [](const float* b) {
            const float x = b[4] - b[0], y = b[5] - b[1], z = b[6] - b[2];
            return x * y + y * z + z * x;
        };
        // The leaves are independent of one another: large trees are matched by several threads, each over a run of consecutive
        // nodes into a list of its own (a leaf's first pair counted within that list), and the lists are joined in node order -
        // the same records in the same places as one thread would write.
        auto match_leaves = [&](size_t node_lo, size_t node_hi, std::vector<float>& pairs) {
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
            std::vector<uint32_t> tri, partner;
            std::vector<float> boxes;
            struct Cand { float gain; uint32_t i, j; };
            std::vector<Cand> cands;
            for (size_t node = node_lo; node < node_hi; ++node) {
                rrt_node& n = dn[node];
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
        };
        {
            const size_t n_nodes = dn.size();
            unsigned n_threads = 1;
            if (n_nodes >= 65536) n_threads = std::min(max_threads ? max_threads : 16u, std::max(1u, std::thread::hardware_concurrency()));
            if (n_threads == 1) match_leaves(0, n_nodes, pairs);
            else {
                std::vector<std::vector<float>> part(n_threads);
                std::vector<std::thread> pool;
                auto lo_of = [&](unsigned t) { return n_nodes * t / n_threads; };
                for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back([&, t]() { match_leaves(lo_of(t), lo_of(t + 1), part[t]); });
                for (std::thread& t : pool) t.join();
                size_t total = 0;
                for (const std::vector<float>& q : part) total += q.size();
                pairs.reserve(total);
                for (unsigned t = 0; t < n_threads; ++t) {
                    const uint32_t base = (uint32_t)(pairs.size() / (4 * kPairQuads));
                    for (size_t node = lo_of(t); node < lo_of(t + 1); ++node)
                        if (dn[node].b & RRT_LEAF_FLAG) dn[node].a += base;
                    pairs.insert(pairs.end(), part[t].begin(), part[t].end());
                    std::vector<float>().swap(part[t]);
                }
            }
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
            out->root_link = root_link;
            if (clean_depth + 1u > kMaxStackDepth) return rrt_set_error(RRT_ERR_LIMIT, "clean tree needs more than 64 pending stack entries per ray");
            out->nodes = std::move(rec);
        } else {
            std::vector<float> rec((size_t)dn.size() * 16);  // node (32 bytes) + primitive box (32 bytes) per 64-byte record
            for (size_t i = 0; i < dn.size(); ++i) {
                std::memcpy(&rec[i * 16], &dn[i], 32);
                std::memcpy(&rec[i * 16 + 8], &tb[i * 8], 32);
            }
            out->nodes = std::move(rec);
        }
        // the sifting traversal of shallow trees packs (pair, lane) into 32 bits
        if (is_shallow(desc) && pairs.size() / (4 * kPairQuads) >= (1u << 25))  // (pair number x 128, the record's byte offset, stays below 2^32)
            return rrt_set_error(RRT_ERR_LIMIT, "a tree of at most 16 nodes (or a List) with 2^25 pair records or more");
        out->pairs = std::move(pairs);
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
        out->plain = std::move(rec);
    }
    {
        if (is_shallow(desc)) out->guards = guard_table(desc, box_pads(desc));
    }
    {
        // (the device copy marks spheres whose world_to_local matrix only translates: see sphere_hit)
        std::vector<rrt_sphere> sp(desc->spheres, desc->spheres + desc->n_spheres);
        for (rrt_sphere& q : sp) {
            const float* m = q.w2l;
            const bool only = m[0] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[5] == 1.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f &&
                              m[10] == 1.0f && std::isfinite(m[3]) && std::isfinite(m[7]) && std::isfinite(m[11]);
            q.pad[0] = only ? kSphereTranslationOnly : 0u;
        }
        out->spheres = std::move(sp);
    }
    out->clean = desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN;
    out->stack_depth = tree_depth;  // what validate_desc measured on the tree, not the caller's bvh_depth (which may only be larger)
    if (desc->aggregate_kind == RRT_AGGREGATE_BVH_CLEAN) out->stack_depth = clean_depth + 1u;
    out->deep = desc->n_nodes > kDeepTreeNodes || out->clean;  // the clean mode's tie rule lives in the cooperative-leaf walk only
    // which code the scene needs (see F_* in rrt_layout.hpp)
    for (uint32_t i = 0; i < desc->n_slots; ++i) {
        uint32_t kind = RRT_INFO_KIND(desc->slots[i].info);
        if (kind == RRT_KIND_SPHERE) out->features |= F_SPHERE;
        if (kind == RRT_KIND_MEDIUM) out->features |= F_MEDIUM | F_SPHERE;
    }
    for (uint32_t i = 0; i < desc->n_media; ++i) {
        if (desc->media[i].boundary_kind == RRT_KIND_MEDIUM) out->features |= F_NESTED;
    }
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind == RRT_KIND_SPHERE) out->features |= F_SPHERE;
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        if (desc->textures[i].type != RRT_TEX_CONSTANT) out->features |= F_TEX;
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        if (desc->materials[i].bump >= 0) out->features |= F_BUMP | F_TEX;
        if (desc->materials[i].type == RRT_MAT_ISOTROPIC) out->features |= F_MEDIUM;
    }
    {
        uint64_t h = 1469598103934665603ull;  // FNV-1a's constants over 8-byte words of everything a kernel reads (a gigabyte for 10 M triangles)
        auto mix = [&h](const void* p, size_t n) {
            const unsigned char* b = (const unsigned char*)p;
            size_t i = 0;
            for (; i + 8 <= n; i += 8) {
                uint64_t w;
                std::memcpy(&w, b + i, 8);
                h = (h ^ w) * 1099511628211ull;
                h ^= h >> 29;
            }
            for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
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
        out->desc_hash = h;
    }
    return RRT_OK;
}

}  // namespace rrt
