// Hit properties (shape.rs), the light mixture (pdf.rs), the materials (material.rs) and one step of aggregate::trace.
// Each function names the reference code it computes (paths relative to the reference root).
#pragma once
#include "rrt_traversal.hpp"

namespace rrt {

// ---- hit properties (shape.rs:87-123, 374-424) ---------------------------------------------------------------------
struct HitProps {
    f3 p, n;
    float u, v;
    f3 pu, pv;
};
RRT_DEV float signum_ref(float f) {  // f32::signum
    if (f != f) return f;
    return (__float_as_uint(f) >> 31) ? -1.0f : 1.0f;
}
RRT_DEV HitProps sphere_props(csphere_p sp, const Ray& r, float t_hit, bool want_uv) {
    const float one_over_2pi = 1.0f / (2.0f * RRT_PI);  // shape.rs:63
    struct { cf1p w2l, l2w; float radius; } s = {(cf1p)sp->w2l, (cf1p)sp->l2w, sp->radius};
    Ray lr = xform_ray(s.w2l, r);
    f3 lp = point_at(lr, t_hit);
    HitProps h;
    h.p = point_at(r, t_hit);
    h.n = norm3(xform_vec(s.l2w, (lp - mk3(0.0f, 0.0f, 0.0f)) / s.radius));
    if (want_uv) {  // uv, pu, pv feed non-constant textures and bump maps only
        f3 hp = lp * (__builtin_fabsf(s.radius) / len3(lp - mk3(0.0f, 0.0f, 0.0f)));
        float theta = det_asin(clampf(hp.y / s.radius, -1.0f, 1.0f));
        float inv_y_radius = (signum_ref(s.radius) * 1.0f) / __builtin_sqrtf(hp.x * hp.x + hp.z * hp.z);
        f3 pu = mk3(2.0f * RRT_PI * hp.z, 0.0f, -2.0f * RRT_PI * hp.x);
        f3 pv = (-RRT_PI) * mk3(hp.y * hp.x * inv_y_radius, (-s.radius) * det_cos(theta), hp.y * hp.z * inv_y_radius);
        h.u = 1.0f - ((det_atan2(hp.z, hp.x) + RRT_PI) * one_over_2pi);
        h.v = (theta + RRT_FRAC_PI_2) * RRT_FRAC_1_PI;
        h.pu = xform_vec(s.l2w, pu);
        h.pv = xform_vec(s.l2w, pv);
    } else {
        h.u = h.v = 0.0f;
        h.pu = h.pv = mk3(0.0f, 0.0f, 0.0f);
    }
    return h;
}
// tri_shade record: s0 = (n.xyz, -), s1 = (uv0, uv1), s2 = (uv2, pu.xy), s3 = (pu.z, pv.xyz); n = normalized(e1 x e2)
// as Triangle::get_hit_properties forms it (shape.rs:390), computed once on the host with the same operations.
RRT_DEV HitProps triangle_props(cv4p shade4, const Ray& r, float t_hit, float u, float v, float det, bool want_uv) {
    const v4f s0 = shade4[0];
    f3 normal = xyz(s0);
    if (det < 0.0f) normal = -normal;
    HitProps h;
    h.p = point_at(r, t_hit);
    h.n = normal;
    if (want_uv) {
        const v4f s1 = shade4[1], s2 = shade4[2], s3 = shade4[3];
        float w = 1.0f - u - v;
        h.u = s1.x * u + s1.z * v + s2.x * w;  // uv0 pairs with u, uv1 with v, uv2 with w (shape.rs:407-410)
        h.v = s1.y * u + s1.w * v + s2.y * w;
        f3 pu = mk3(s2.z, s2.w, s3.x);
        if (det < 0.0f) pu = -pu;
        h.pu = pu;
        h.pv = mk3(s3.y, s3.z, s3.w);
    } else {
        h.u = h.v = 0.0f;
        h.pu = h.pv = mk3(0.0f, 0.0f, 0.0f);
    }
    return h;
}
// Shape::get_hit_properties for whatever sits in `hit.slot`; *material receives Shape::get_material.
template <uint32_t FEAT>
RRT_DEV HitProps slot_props(const DScene& sc, const Hit& hit, const Ray& r, uint32_t* material) {
    const int slot = hit.slot;
    const cv4p cslots = sc.c_slots();
    const uint32_t info = __float_as_uint(cslots[3 * slot + 1].w);
    *material = RRT_INFO_MATERIAL(info);
    const uint32_t kind = RRT_INFO_KIND(info);
    const bool want_uv = (FEAT & (F_TEX | F_BUMP)) != 0;
    if (!(FEAT & (F_SPHERE | F_MEDIUM)) || kind == RRT_KIND_TRIANGLE) return triangle_props(sc.c_tri_shade() + 4 * slot, r, hit.t, hit.u, hit.v, hit.det, want_uv);
    const uint32_t aux = __float_as_uint(cslots[3 * slot + 2].w);
    if ((FEAT & F_SPHERE) && kind == RRT_KIND_SPHERE) return sphere_props(sc.c_spheres() + aux, r, hit.t, want_uv);
    cmedium_p m = sc.c_media() + aux;  // volume.rs:104-106: the boundary's properties, whatever the boundary is
    if (FEAT & F_NESTED)
        for (int i = 1; i < RRT_MAX_MEDIUM_NESTING && m->boundary_kind == RRT_KIND_MEDIUM; ++i) m = sc.c_media() + m->boundary_index;
    if (m->boundary_kind == RRT_KIND_SPHERE) return sphere_props(sc.c_spheres() + m->boundary_index, r, hit.t, want_uv);
    const uint32_t bi = m->boundary_index;
    const cv4p bt = sc.c_btris() + 3 * bi;
    float u, v, det;
    triangle_uvdet(xyz(bt[0]), xyz(bt[1]), xyz(bt[2]), r, &u, &v, &det);
    return triangle_props(sc.c_btri_shade() + 4 * bi, r, hit.t, u, v, det, want_uv);
}

// ---- light mixture (pdf.rs:85-128; shape.rs:141-166, 441-471) -----------------------------------------------------------
RRT_DEV float light_pdf(const DScene& sc, clight_p l, const Ray& r) {
    if (l->kind == RRT_KIND_SPHERE) {  // Sphere::pdf
        const csphere_p s = sc.c_spheres() + l->index;
        float t;
        if (!sphere_hit(s, r, RRT_T_MIN, RRT_T_MAX, &t)) return 0.0f;
        Ray lr = xform_ray((cf1p)s->w2l, r);
        const float radius = s->radius;
        float cos_theta_max = __builtin_sqrtf(fmax_ref(1.0f - radius * radius / sqlen3(mk3(0.0f, 0.0f, 0.0f) - lr.o), 0.0f));
        float solid_angle = 2.0f * RRT_PI * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    // Triangle::pdf
    float t_hit;
    if (!triangle_hit(ld3((cf1p)l->v0), ld3((cf1p)l->e1), ld3((cf1p)l->e2), l->cull != 0, r, RRT_T_MIN, RRT_T_MAX, &t_hit)) return 0.0f;
    float dist_squared = t_hit * t_hit * sqlen3(r.d);
    // the hit normal is +-nrm; only |cos| is used, and negation is exact
    float cosine = __builtin_fabsf(dot3(r.d, ld3((cf1p)l->nrm)) / len3(r.d));
    return dist_squared / (cosine * l->area);
}
RRT_DEV float mixture_value(const DScene& sc, const Ray& r) {  // Mixture::value
    float weight = 1.0f / (float)sc.n_lights;
    float sum = 0.0f;
    const clight_p lights = sc.c_lights();
    for (uint32_t i = 0; i < sc.n_lights; ++i) sum += weight * light_pdf(sc, lights + i, r);
    return sum;
}
// Mixture::value from harvested light distances: light_pdf without the hit tests, which the closest-hit query of this
// same ray has already run (harvest[i] = the distance, -1 when the ray misses light i's primitive).
RRT_DEV float mixture_value_harvested(const DScene& sc, const Ray& r, const float* harvest) {
    float weight = 1.0f / (float)sc.n_lights;
    float sum = 0.0f;
    const clight_p lights = sc.c_lights();
    for (uint32_t i = 0; i < kHarvestLights; ++i) {
        if (i >= sc.n_lights) break;
        const clight_p l = lights + i;
        float p = 0.0f;
        if (harvest[i] > 0.0f) {
            if (l->kind == RRT_KIND_SPHERE) {
                const csphere_p s = sc.c_spheres() + l->index;
                Ray lr = xform_ray((cf1p)s->w2l, r);
                const float radius = s->radius;
                float cos_theta_max = __builtin_sqrtf(fmax_ref(1.0f - radius * radius / sqlen3(mk3(0.0f, 0.0f, 0.0f) - lr.o), 0.0f));
                float solid_angle = 2.0f * RRT_PI * (1.0f - cos_theta_max);
                p = 1.0f / solid_angle;
            } else {
                const float t_hit = harvest[i];
                float dist_squared = t_hit * t_hit * sqlen3(r.d);
                float cosine = __builtin_fabsf(dot3(r.d, ld3((cf1p)l->nrm)) / len3(r.d));
                p = dist_squared / (cosine * l->area);
            }
        }
        sum += weight * p;
    }
    return sum;
}
RRT_DEV float cosine_value(f3 normal, const Ray& r) {  // pdf.rs:51-58
    float cosine = dot3(norm3(r.d), normal);
    return cosine < 0.0f ? 0.0f : cosine / RRT_PI;
}
// The scattered direction of the PDF branch of trace (aggregate.rs:58-77): pair_generate (pdf.rs:121-128) picks the light
// mixture or the cosine lobe; Mixture::generate (pdf.rs:109-114) picks a member.  Sphere lights (random_to_sphere) and the
// cosine lobe (random_cosine_direction) are both "two draws, one sincos, one orthonormal basis", so they share that code:
// only the basis axis and the (z, sin-radius) pair differ.  Draw order is the reference's: [pair], [member], r1, r2.
// The first draws of that branch - pair_generate's coin (pdf.rs:122) and Mixture::generate's member (pdf.rs:110) - as a step of
// its own, for callers that sort paths by the outcome before they shade them.
struct LightChoice {
    bool towards_light;
    uint32_t k;
};
RRT_DEV LightChoice draw_light_choice(const DScene& sc, Rng& g) {
    LightChoice c = {false, 0u};
    if (sc.n_lights > 0) {
        float xi = rng_f32(g);
        if (xi < 0.5f) {
            c.towards_light = true;
            c.k = rng_index(g, sc.n_lights);
        }
    }
    return c;
}
// `chosen`: the outcome of draw_light_choice() when the caller has made those draws already (nullptr: they are made here).
RRT_DEV f3 sample_scatter_direction(const DScene& sc, f3 n, f3 from, Rng& g, const LightChoice* chosen = nullptr) {
    const LightChoice choice = chosen ? *chosen : draw_light_choice(sc, g);
    const bool towards_light = choice.towards_light;
    const uint32_t k = choice.k;
    float r1 = rng_f32(g);
    float r2 = rng_f32(g);
    bool tri_light = false;
    bool sphere_light = false;
    if (towards_light) {
        uint32_t kind = sc.c_lights()[k].kind;
        tri_light = kind == RRT_KIND_TRIANGLE;
        sphere_light = !tri_light;
    }
    if (tri_light) {  // Triangle::random_dir_towards
        const clight_p l = sc.c_lights() + k;
        f3 random_point = ld3((cf1p)l->v0) * (1.0f - __builtin_sqrtf(r1)) + ld3((cf1p)l->v1) * (__builtin_sqrtf(r1) * (1.0f - r2)) +
                          ld3((cf1p)l->v2) * (r2 * __builtin_sqrtf(r1));
        return random_point - from;
    }
    f3 axis = n;
    float z, rad;  // local direction = (cos(phi) * rad, sin(phi) * rad, z)
    csphere_p s = sc.c_spheres();
    if (sphere_light) {  // Sphere::random_dir_towards + random_to_sphere
        s = sc.c_spheres() + sc.c_lights()[k].index;
        f3 local_point = xform_point((cf1p)s->w2l, from);
        axis = mk3(0.0f, 0.0f, 0.0f) - local_point;
        float distance_squared = sqlen3(axis);
        const float radius = s->radius;
        z = 1.0f + r2 * (__builtin_sqrtf(fmax_ref(1.0f - radius * radius / distance_squared, 0.0f)) - 1.0f);
        rad = __builtin_sqrtf(fmax_ref(1.0f - z * z, 0.0f));
    } else {  // random_cosine_direction
        z = __builtin_sqrtf(1.0f - r2);
        rad = __builtin_sqrtf(r2);
    }
    float phi = 2.0f * RRT_PI * r1;
    float sp, cp;
    det_sincos(phi, &sp, &cp);
    f3 d = onb_local(axis, mk3(cp * rad, sp * rad, z));
    if (sphere_light) d = xform_vec((cf1p)s->l2w, d);
    return d;
}


// ---- materials (material.rs:15-54) -----------------------------------------------------------------------------------------
RRT_DEV f3 reflect_ref(f3 v, f3 n) { return v - 2.0f * dot3(v, n) * n; }
RRT_DEV f3 refract_ref(f3 v, f3 n, float index) {
    float cos_theta = dot3(-v, n);
    f3 r_par = index * (v + cos_theta * n);
    f3 r_perp = (-__builtin_sqrtf(1.0f - sqlen3(r_par))) * n;
    return r_par + r_perp;
}
RRT_DEV float schlick_ref(float cosine, float index) {
    float r0 = (1.0f - index) / (1.0f + index);
    r0 = r0 * r0;
    float a = 1.0f - cosine;
    float a2 = a * a;
    float a4 = a2 * a2;
    return r0 + (1.0f - r0) * (a * a4);  // f32::powi(5) = a * ((a*a)*(a*a))
}
// Texture::value; without F_TEX every texture is Constant.
template <uint32_t FEAT>
RRT_DEV f3 albedo_value(const DScene& sc, int tex, float u, float v, f3 p) {
    if (FEAT & F_TEX) return texture_value(sc, tex, u, v, p);
    const ctexture_p t = sc.c_textures() + tex;
    return mk3(t->a, t->b, t->c);
}
RRT_DEV f3 bump_normal(const DScene& sc, const HitProps& hp, int bump_tex) {
    const float delta = 0.005f;
    float d0 = texture_bump_value(sc, bump_tex, hp.u, hp.v, hp.p);
    float du = texture_bump_value(sc, bump_tex, hp.u + delta, hp.v, hp.p + delta * hp.pu);
    float dv = texture_bump_value(sc, bump_tex, hp.u, hp.v + delta, hp.p + delta * hp.pv);
    f3 new_pu = hp.pu + ((du - d0) / delta) * hp.n;
    f3 new_pv = hp.pv + ((dv - d0) / delta) * hp.n;
    return norm3(cross3(new_pu, new_pv));
}

RRT_DEV f3 background(uint32_t bg, const Ray& r) {
    if (bg == RRT_BG_BLACK) return mk3(0.0f, 0.0f, 0.0f);  // main.rs:268-270
    f3 dn = norm3(r.d);                                    // main.rs:259-267
    float t = 0.5f * (dn.y + 1.0f);
    return mk3(1.0f, 1.0f, 1.0f) * (1.0f - t) + mk3(0.5f, 0.7f, 1.0f) * t;
}

// Camera::get_ray (camera.rs:53-63) for pixel (x, y) with the jitter of main.rs:227-228
RRT_DEV Ray camera_ray(const rrt_camera& c, uint32_t x, uint32_t y, uint32_t W, uint32_t H, Rng& g) {
    float u = ((float)x + rng_f32(g)) / (float)W;
    float v = ((float)(H - y) + rng_f32(g)) / (float)H;
    f3 disk = c.lens_radius * random_unit_disk(g);
    f3 cu = ld3(c.u), cv = ld3(c.v);
    f3 offset = cu * disk.x + cv * disk.y;
    Ray r;
    r.o = ld3(c.origin) + offset;
    r.d = ld3(c.lower_left_corner) + (ld3(c.horizontal) * u) + (ld3(c.vertical) * v) - ld3(c.origin) - offset;
    return r;
}

// One step of aggregate::trace (aggregate.rs:26-104) after the closest-hit query, in loop form: either the path ends
// with a radiance value (`*radiance`, to be weighted by the throughput) or the ray and throughput are advanced.
// Returns true when the path ended.
// Pending throughput update of a diffuse bounce whose mixture density is still to come (see HARVEST above).
struct PendingPdf {
    bool active;
    f3 numerator;       // throughput * (attenuation * scattering_pdf)
    float scatter_pdf;  // the cosine lobe's density for the scattered ray
};
template <uint32_t FEAT, bool HARVEST = false>
RRT_DEV bool shade(const DScene& sc, const Hit& hit, Ray& r, f3& thr, Rng& g, uint32_t depth, uint32_t max_depth, uint32_t bg, f3* radiance,
                   PendingPdf* pend = nullptr, const LightChoice* chosen = nullptr) {
    if (!(depth < max_depth && hit.slot >= 0)) {
        *radiance = background(bg, r);
        return true;
    }
    // Plain scenes (constant textures, no bump maps, no media) keep one record per slot with everything below: see DScene::plain.
    constexpr bool kPlainRecord = (FEAT & (F_TEX | F_BUMP | F_MEDIUM)) == 0;
    HitProps hp;
    struct { uint32_t type; int32_t albedo, bump; float param; } m;
    f3 plain_colour = mk3(0.0f, 0.0f, 0.0f);
    if (kPlainRecord) {
        const v4f s0 = sc.c_plain()[2 * hit.slot], s1 = sc.c_plain()[2 * hit.slot + 1];
        const uint32_t bits = __float_as_uint(s0.w);
        m = {RRT_PLAIN_TYPE(bits), -1, -1, s1.w};
        plain_colour = xyz(s1);
        if (!(FEAT & F_SPHERE) || RRT_PLAIN_KIND(bits) == RRT_KIND_TRIANGLE) {  // triangle_props without texture coordinates
            hp.p = point_at(r, hit.t);
            hp.n = hit.det < 0.0f ? -xyz(s0) : xyz(s0);
            hp.u = hp.v = 0.0f;
            hp.pu = hp.pv = mk3(0.0f, 0.0f, 0.0f);
        } else {
            hp = sphere_props(sc.c_spheres() + RRT_PLAIN_AUX(bits), r, hit.t, false);
        }
    } else {
        uint32_t mat_id;
        hp = slot_props<FEAT>(sc, hit, r, &mat_id);
        if (FEAT & (F_TEX | F_BUMP)) {
            hp.u = clampf(hp.u, 0.0f, 1.0f);  // aggregate.rs:31
            hp.v = clampf(hp.v, 0.0f, 1.0f);
        }
        const cmaterial_p mp = sc.c_materials() + mat_id;
        m = {mp->type, mp->albedo, mp->bump, mp->param};
    }
    if (m.type == RRT_MAT_DIFFUSE_LIGHT) {  // material.rs:229-231: emission ends the path
        *radiance = kPlainRecord ? plain_colour : albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
        return true;
    }
    if (m.type == RRT_MAT_DIELECTRIC) {  // material.rs:172-207
        float eta;
        f3 n_use;
        if (dot3(r.d, hp.n) < 0.0f) {
            eta = 1.0f / m.param;
            n_use = hp.n;
        } else {
            eta = m.param;
            n_use = -hp.n;
        }
        f3 unit_dir = norm3(r.d);
        float cos_theta = fmin_ref(dot3(-unit_dir, n_use), 1.0f);
        float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
        f3 out;
        if (eta * sin_theta > 1.0f) {
            out = reflect_ref(unit_dir, n_use);
        } else {
            float reflect_prob = schlick_ref(cos_theta, eta);
            if (rng_f32(g) < reflect_prob) out = reflect_ref(unit_dir, n_use);
            else out = refract_ref(unit_dir, n_use, eta);
        }
        thr = thr * mk3(1.0f, 1.0f, 1.0f);
        r.o = hp.p;
        r.d = out;
        return false;
    }
    if ((FEAT & F_MEDIUM) && m.type == RRT_MAT_ISOTROPIC) {  // volume.rs:27-35
        f3 out = unit_sphere_random(g);
        f3 att = kPlainRecord ? plain_colour : albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
        thr = thr * att;
        r.o = hp.p;
        r.d = out;
        return false;
    }
    f3 n = hp.n;
    if ((FEAT & F_BUMP) && m.bump >= 0) n = bump_normal(sc, hp, m.bump);
    if (m.type == RRT_MAT_METAL) {  // material.rs:145-160
        f3 reflected = reflect_ref(norm3(r.d), n);
        f3 out = reflected + m.param * unit_sphere_random(g);
        f3 att = kPlainRecord ? plain_colour : albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
        thr = thr * att;
        r.o = hp.p;
        r.d = out;
        return false;
    }
    // Lambert (material.rs:98-110) + the PDF branch of trace (aggregate.rs:57-90)
    f3 att = kPlainRecord ? plain_colour : albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
    Ray scattered;
    scattered.o = hp.p;
    scattered.d = sample_scatter_direction(sc, n, hp.p, g, chosen);
    const float scatter_pdf = cosine_value(n, scattered);
    if (HARVEST) {  // n_lights > 0 here; the division by pair_value happens after the next closest-hit query
        pend->active = true;
        pend->numerator = thr * (att * scatter_pdf);
        pend->scatter_pdf = scatter_pdf;
        r = scattered;
        return false;
    }
    float pdf_val;
    if (sc.n_lights > 0) pdf_val = mixture_value(sc, scattered) * 0.5f + scatter_pdf * 0.5f;  // pair_value, pdf.rs:117-119
    else pdf_val = scatter_pdf;
    thr = (thr * (att * scatter_pdf)) / pdf_val;
    r = scattered;
    return false;
}

}  // namespace rrt
