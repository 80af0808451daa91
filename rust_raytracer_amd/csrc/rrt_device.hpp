// Device-side building blocks of the path tracer (gfx950).  Everything here is plain IEEE f32,
// compiled with -ffp-contract=off: the reference (Rust) never fuses a*b+c, and hit/miss decisions
// are only reproducible bit for bit with the same unfused operations in the same operand order.
// Each function names the reference code it computes (paths relative to the reference root).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rrt.h"

#define RRT_DEV __device__ __forceinline__
#define RRT_BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)  // lanes where `pred` holds (the HIP __ballot takes an int and costs two more instructions)

namespace rrt {

// ---- 3-vectors (vector.rs:57-79, base.rs:29-128) ------------------------------------------------
struct f3 {
    float x, y, z;
};
RRT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RRT_DEV f3 xyz(const float4& q) { return f3{q.x, q.y, q.z}; }
RRT_DEV f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
RRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
RRT_DEV float sqlen3(f3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
RRT_DEV float len3(f3 a) { return __builtin_sqrtf(sqlen3(a)); }
RRT_DEV f3 norm3(f3 a) { return a / len3(a); }
RRT_DEV f3 cross3(f3 a, f3 b) { return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }

#define RRT_PI 3.14159265358979323846f
#define RRT_FRAC_PI_2 1.57079632679489661923f
#define RRT_FRAC_1_PI 0.318309886183790671538f
#define RRT_EPS 1.1920929e-7f
#define RRT_T_MIN 0.001f
#define RRT_T_MAX 3.402823466e+38f

RRT_DEV float clampf(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }  // utils.rs:51-59
RRT_DEV float fmax_ref(float f, float m) { return f > m ? f : m; }                             // utils.rs:61-67
RRT_DEV float fmin_ref(float f, float m) { return f < m ? f : m; }                             // utils.rs:69-75
RRT_DEV float lerpf(float t, float a, float b) { return (1.0f - t) * a + t * b; }              // utils.rs:77-79

// ---- deterministic elementary functions --------------------------------------------------------
// sin/cos/asin/atan2/log built from single correctly-rounded f32 operations only, so the device and
// any IEEE host produce identical bits (the platform libm / OCML versions differ in the last ulp).
RRT_DEV void det_sincos(float x, float* s_out, float* c_out) {
    const float two_over_pi = 0.63661977236758134308f;
    const float c1 = 1.5703125f;
    const float c2 = 4.837512969970703125e-4f;
    const float c3 = 7.54978995489188e-8f;
    float k = __builtin_floorf(x * two_over_pi + 0.5f);
    float r = ((x - k * c1) - k * c2) - k * c3;
    float r2 = r * r;
    float sp = -1.9841270114e-4f + r2 * 2.7557314297e-6f;
    sp = 8.3333337680e-3f + r2 * sp;
    sp = -1.6666667163e-1f + r2 * sp;
    float sr = r + (r * r2) * sp;
    float cp = -1.3888889225e-3f + r2 * 2.4801587642e-5f;
    cp = 4.1666667908e-2f + r2 * cp;
    float cr = (1.0f - 0.5f * r2) + (r2 * r2) * cp;
    float q = k - 4.0f * __builtin_floorf(k * 0.25f);
    float s, c;
    if (q == 0.0f) { s = sr; c = cr; }
    else if (q == 1.0f) { s = cr; c = -sr; }
    else if (q == 2.0f) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
    *s_out = s;
    *c_out = c;
}
RRT_DEV float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }
RRT_DEV float det_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }
RRT_DEV float det_atan_poly(float a) {
    float z = a * a;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    return a + (a * z) * p;
}
RRT_DEV float det_atan2(float y, float x) {
    float ax = x < 0.0f ? -x : x;
    float ay = y < 0.0f ? -y : y;
    float hi = ax > ay ? ax : ay;
    float lo = ax > ay ? ay : ax;
    float r;
    if (hi == 0.0f) {
        r = 0.0f;
    } else {
        float a = lo / hi;
        if (a > 0.4142135623730950f) r = 0.785398163397448309616f + det_atan_poly((a - 1.0f) / (a + 1.0f));
        else r = det_atan_poly(a);
    }
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    if (y < 0.0f) r = -r;
    return r;
}
RRT_DEV float det_asin_poly(float z) {
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return p;
}
RRT_DEV float det_asin(float x) {
    float a = x < 0.0f ? -x : x;
    float r;
    if (a > 1.0f) {
        r = __builtin_nanf("");
    } else if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = __builtin_sqrtf(z);
        float as = s + (s * z) * det_asin_poly(z);
        r = 1.57079632679489661923f - 2.0f * as;
    } else {
        float z = a * a;
        r = a + (a * z) * det_asin_poly(z);
    }
    return x < 0.0f ? -r : r;
}
RRT_DEV float det_log(float x) {
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xffu) - 127;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e = e + 1;
    }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.2222222222f + s2 * 0.1818181818f;
    p = 0.2857142857f + s2 * p;
    p = 0.4f + s2 * p;
    p = 0.6666666667f + s2 * p;
    float lm = 2.0f * s + (s * s2) * p;
    float fe = (float)e;
    return (fe * 0.693359375f + (fe * -2.12194440054690583e-4f)) + lm;
}

// ---- per-path random stream ----------------------------------------------------------------------
// Stands in for rand::random::<f32>() (rand 0.7.3, unseedable in the reference): xoroshiro64* seeded
// by a hash of (seed, pixel, sample); f32 = top 24 bits * 2^-24, the crate's Standard distribution.
RRT_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RRT_DEV uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
struct Rng {
    uint32_t s0, s1;
};
RRT_DEV Rng rng_seed(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t a = mix32(pixel ^ mix32(seed));
    uint32_t b = mix32(sample ^ 0x5bd1e995u);
    Rng r;
    r.s0 = mix32(a + b);
    r.s1 = mix32(b ^ r.s0);
    if ((r.s0 | r.s1) == 0u) r.s1 = 0x9e3779b9u;
    return r;
}
RRT_DEV uint32_t rng_u32(Rng& r) {
    uint32_t s0 = r.s0, s1 = r.s1;
    uint32_t out = s0 * 0x9e3779bbu;
    s1 ^= s0;
    r.s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9);
    r.s1 = rotl32(s1, 13);
    return out;
}
RRT_DEV float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }
RRT_DEV float rng_f32(Rng& r) { return u32_to_unit(rng_u32(r)); }
RRT_DEV uint32_t rng_index(Rng& r, uint32_t n) { return __umulhi(rng_u32(r), n); }  // SliceRandom::choose, pdf.rs:110
// The draw of ConstantMedium::hit (volume.rs:95) is keyed by (stream state, depth, primitive) so that it
// does not depend on which other BVH nodes were visited before.
RRT_DEV float rng_keyed(const Rng& r, uint32_t depth, uint32_t prim) {
    uint32_t h = mix32(r.s1 + depth * 0x9e3779b9u + prim * 0x85ebca6bu);
    return u32_to_unit(mix32(r.s0 ^ h));
}

// ---- read-only scene data is addressed through the constant address space ---------------------------
// Loads whose address is wave-uniform then go through the scalar cache into SGPRs (no VGPRs, no per-lane address
// arithmetic); loads with per-lane addresses become ordinary vector loads that the compiler may hoist and merge,
// because constant-address-space memory cannot be written by the kernel.
#define RRT_CONST_AS __attribute__((address_space(4)))
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f RRT_CONST_AS* cv4p;
typedef const float RRT_CONST_AS* cf1p;
typedef const rrt_sphere RRT_CONST_AS* csphere_p;
typedef const rrt_medium RRT_CONST_AS* cmedium_p;
typedef const rrt_material RRT_CONST_AS* cmaterial_p;
typedef const rrt_texture RRT_CONST_AS* ctexture_p;
typedef const rrt_light RRT_CONST_AS* clight_p;
RRT_DEV f3 ld3(cf1p p) { return f3{p[0], p[1], p[2]}; }
RRT_DEV f3 xyz(v4f q) { return f3{q.x, q.y, q.z}; }

// ---- device view of the scene ----------------------------------------------------------------------
struct DScene {
    const float4* nodes;      // 4 x float4 per node: the rrt_node (a leaf's `a` holds its first PAIR index), then the padded
                              // box its primitives occupy (lo, hi), which the TIGHT traversal tests as well: one 64-byte line.
                              // Clean trees: inner nodes only, 4 x float4 each: (left box lo, left link), (left box hi, right link),
                              // (right box lo, cut axis), (right box hi, -); a link is an inner node's index, or
                              // kLinkLeaf | (pairs - 1) << 28 | first pair
    const float4* pairs;      // kPairQuads x float4 per two consecutive primitives of a leaf (see triangle_pair_eval)
    const float4* slots;      // 3 x float4 per rrt_slot
    const float4* tri_shade;  // 4 x float4 per rrt_tri_shade
    const rrt_sphere* spheres;
    const rrt_medium* media;
    const float4* btris;
    const float4* btri_shade;
    const rrt_material* materials;
    const rrt_texture* textures;
    const uint8_t* texels;
    const rrt_light* lights;
    uint32_t n_nodes, n_slots, n_lights, is_list;
    uint32_t root_link;  // clean trees: where the walk starts (see walk_step)
    rrt_camera cam;

    RRT_DEV cv4p c_nodes() const { return (cv4p)(const void*)nodes; }
    RRT_DEV cv4p c_pairs() const { return (cv4p)(const void*)pairs; }
    RRT_DEV cv4p c_slots() const { return (cv4p)(const void*)slots; }
    RRT_DEV cv4p c_tri_shade() const { return (cv4p)(const void*)tri_shade; }
    RRT_DEV cv4p c_btris() const { return (cv4p)(const void*)btris; }
    RRT_DEV cv4p c_btri_shade() const { return (cv4p)(const void*)btri_shade; }
    RRT_DEV csphere_p c_spheres() const { return (csphere_p)(const void*)spheres; }
    RRT_DEV cmedium_p c_media() const { return (cmedium_p)(const void*)media; }
    RRT_DEV cmaterial_p c_materials() const { return (cmaterial_p)(const void*)materials; }
    RRT_DEV ctexture_p c_textures() const { return (ctexture_p)(const void*)textures; }
    RRT_DEV clight_p c_lights() const { return (clight_p)(const void*)lights; }
};

struct Ray {
    f3 o, d;
};
RRT_DEV f3 point_at(const Ray& r, float t) { return r.o + (r.d * t); }  // ray.rs:17-19

// ---- samplers (utils.rs:10-49) and the orthonormal basis (utils.rs:191-214) ---------------------------
RRT_DEV f3 random_unit_disk(Rng& g) {
    float x = 2.0f * rng_f32(g) - 1.0f;
    float y = __builtin_sqrtf(1.0f - x * x);
    return mk3(x, y, 0.0f);
}
RRT_DEV f3 unit_sphere_random(Rng& g) {
    float azimuth = rng_f32(g) * RRT_PI * 2.0f;
    float y = rng_f32(g);
    float sin_elevation = __builtin_sqrtf(1.0f - y * y);
    float sa, ca;
    det_sincos(azimuth, &sa, &ca);
    return mk3(sin_elevation * ca, y, sin_elevation * sa);
}
RRT_DEV f3 random_cosine_direction(Rng& g) {
    float r1 = rng_f32(g);
    float r2 = rng_f32(g);
    float z = __builtin_sqrtf(1.0f - r2);
    float phi = 2.0f * RRT_PI * r1;
    float sp, cp;
    det_sincos(phi, &sp, &cp);
    float x = cp * __builtin_sqrtf(r2);
    float y = sp * __builtin_sqrtf(r2);
    return mk3(x, y, z);
}
RRT_DEV f3 random_to_sphere(Rng& g, float radius, float distance_squared) {
    float r1 = rng_f32(g);
    float r2 = rng_f32(g);
    float z = 1.0f + r2 * (__builtin_sqrtf(fmax_ref(1.0f - radius * radius / distance_squared, 0.0f)) - 1.0f);
    float phi = 2.0f * RRT_PI * r1;
    float sp, cp;
    det_sincos(phi, &sp, &cp);
    float x = cp * __builtin_sqrtf(fmax_ref(1.0f - z * z, 0.0f));
    float y = sp * __builtin_sqrtf(fmax_ref(1.0f - z * z, 0.0f));
    return mk3(x, y, z);
}
// OrthonormalBasis::new(v).local(d); v is used as given (not normalised), as the reference does.
RRT_DEV f3 onb_local(f3 v, f3 d) {
    f3 a = __builtin_fabsf(v.x) > 0.9f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    f3 axis1 = norm3(cross3(v, a));
    f3 axis0 = cross3(v, axis1);
    return (d.x * axis0) + (d.y * axis1) + (d.z * v);
}

// ---- Perlin noise (utils.rs:82-189) ----------------------------------------------------------------------
__device__ const uint8_t k_noise_perm[256] = {
    63, 147, 186, 78, 92, 53, 229, 76, 14, 204, 183, 99, 237, 241, 59, 167, 118, 23, 29, 44, 82,
    37, 6, 249, 131, 253, 210, 28, 71, 96, 3, 207, 115, 32, 158, 61, 215, 220, 116, 40, 48, 93,
    179, 196, 141, 0, 165, 185, 145, 217, 139, 216, 250, 235, 39, 232, 124, 146, 5, 77, 180, 4, 31,
    203, 154, 178, 226, 25, 20, 130, 22, 240, 252, 163, 75, 90, 51, 89, 151, 193, 33, 69, 21, 149,
    208, 244, 238, 191, 161, 36, 38, 81, 181, 56, 43, 127, 34, 243, 65, 200, 97, 247, 79, 231, 98,
    11, 100, 142, 15, 166, 45, 209, 223, 66, 119, 155, 49, 153, 113, 41, 133, 197, 157, 112, 46,
    91, 74, 27, 128, 228, 16, 248, 174, 187, 87, 95, 30, 110, 212, 175, 144, 135, 225, 172, 221,
    170, 67, 9, 111, 224, 239, 176, 117, 109, 177, 202, 132, 80, 125, 62, 251, 108, 148, 103, 227,
    50, 17, 35, 24, 126, 164, 42, 156, 10, 182, 218, 70, 246, 150, 73, 213, 138, 129, 189, 188, 84,
    160, 134, 105, 83, 169, 121, 233, 194, 19, 114, 55, 211, 58, 104, 254, 57, 18, 123, 102, 140,
    8, 171, 68, 206, 168, 86, 136, 152, 47, 60, 88, 101, 26, 122, 13, 192, 94, 198, 64, 234, 195,
    52, 245, 54, 236, 219, 12, 106, 143, 120, 7, 190, 1, 2, 205, 222, 159, 162, 173, 85, 107, 201,
    184, 214, 137, 230, 255, 242, 72, 199};
RRT_DEV int perm_at(int i) { return (int)k_noise_perm[i & 255]; }  // the reference's 512-entry table is this one twice
RRT_DEV int f32_to_i32_sat(float f) {  // Rust `as i32`
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
RRT_DEV uint32_t f32_to_u32_sat(float f) {  // Rust `as u32`
    if (f != f) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
RRT_DEV float perlin_gradient(int x, int y, int z, float dx, float dy, float dz) {  // utils.rs:149-170
    int val = perm_at(perm_at(perm_at(x) + y) + z) & 15;
    float u = (val < 8 || val == 12 || val == 13) ? dx : dy;
    float v = (val < 4 || val == 12 || val == 13) ? dy : dz;
    if ((val & 1) > 0) u = -u;
    if ((val & 2) > 0) v = -v;
    return u + v;
}
RRT_DEV float perlin_smooth(float f) {  // utils.rs:171-175
    float f_3 = f * f * f;
    float f_4 = f_3 * f;
    return 6.0f * f_4 * f - 15.0f * f_4 + 10.0f * f_3;
}
RRT_DEV float perlin_noise(f3 p) {  // utils.rs:112-148
    int ix = f32_to_i32_sat(__builtin_floorf(p.x));
    int iy = f32_to_i32_sat(__builtin_floorf(p.y));
    int iz = f32_to_i32_sat(__builtin_floorf(p.z));
    float dx = p.x - (float)ix, dy = p.y - (float)iy, dz = p.z - (float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    float w000 = perlin_gradient(ix, iy, iz, dx, dy, dz);
    float w100 = perlin_gradient(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    float w010 = perlin_gradient(ix, iy + 1, iz, dx, dy - 1.0f, dz);
    float w001 = perlin_gradient(ix, iy, iz + 1, dx, dy, dz - 1.0f);
    float w110 = perlin_gradient(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    float w101 = perlin_gradient(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    float w011 = perlin_gradient(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f);
    float w111 = perlin_gradient(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    float wx = perlin_smooth(dx), wy = perlin_smooth(dy), wz = perlin_smooth(dz);
    float x00 = lerpf(wx, w000, w100);
    float x10 = lerpf(wx, w010, w110);
    float x01 = lerpf(wx, w001, w101);
    float x11 = lerpf(wx, w011, w111);
    float y0 = lerpf(wy, x00, x10);
    float y1 = lerpf(wy, x01, x11);
    return lerpf(wz, y0, y1);
}
RRT_DEV float perlin_turbulence(f3 p, uint32_t depth, float omega) {  // utils.rs:177-189
    float sum = 0.0f;
    float weight = 1.0f;
    for (uint32_t i = 0; i < depth; ++i) {
        sum += weight * perlin_noise(p);
        weight *= omega;
        p = p * 1.99f;
    }
    return __builtin_fabsf(sum);
}

// ---- textures (texture.rs:61-179) ------------------------------------------------------------------------
RRT_DEV f3 texture_value(const DScene& sc, int tex, float u, float v, f3 p) {
    for (;;) {
        const ctexture_p tp = sc.c_textures() + tex;
        struct { uint32_t type; float a, b, c; uint32_t u0, u1, u2, u3; } t = {tp->type, tp->a, tp->b, tp->c, tp->u0, tp->u1, tp->u2, tp->u3};
        switch (t.type) {
            case RRT_TEX_CONSTANT:
                return mk3(t.a, t.b, t.c);
            case RRT_TEX_TEST: {
                float w = 1.0f - u - v;
                return mk3(u, v, w < 0.0f ? 0.0f : w);
            }
            case RRT_TEX_CHECKER: {
                float sines = det_sin(t.a * p.x) * det_sin(t.a * p.y) * det_sin(t.a * p.z);
                tex = sines < 0.0f ? (int)t.u0 : (int)t.u1;
                continue;
            }
            case RRT_TEX_IMAGE: {
                uint32_t i = f32_to_u32_sat(u * (float)t.u0) % t.u0;
                uint32_t j = f32_to_u32_sat((1.0f - v) * (float)t.u1) % t.u1;
                const uint8_t* px = sc.texels + t.u3 + ((size_t)j * t.u0 + i) * t.u2;
                uint8_t r = px[0];
                uint8_t g = t.u2 >= 3 ? px[1] : r;
                uint8_t b = t.u2 >= 3 ? px[2] : r;
                f3 c = mk3((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f);
                return c * c;  // color.rs:41-43
            }
            case RRT_TEX_NOISE:
                return mk3(0.5f, 0.5f, 0.5f) * (1.0f + perlin_noise(p * t.a));
            default:  // RRT_TEX_TURBULENCE
                return mk3(1.0f, 1.0f, 1.0f) * perlin_turbulence(p * t.a, t.u0, t.b);
        }
    }
}
RRT_DEV float texture_bump_value(const DScene& sc, int tex, float u, float v, f3 p) {  // texture.rs:63-66
    f3 b = texture_value(sc, tex, u, v, p);
    return (b.x + b.y + b.z) / 3.0f;
}

// ---- exact reciprocal ------------------------------------------------------------------------------------------
// 1.0f / x, correctly rounded: the hardware reciprocal estimate plus one fused Newton step, accepted when the step's
// residual e = 1 - x*r0 is small.  On gfx950 that is exact for every float whose residual passes (all normal x whose
// reciprocal is normal; zero, denormal, huge and non-finite x fail the residual test and take the full division);
// rrt_selftest_reciprocal repeats the exhaustive comparison with IEEE division over all 2^32 bit patterns.
#define RRT_RCP_RESIDUAL 9.5367431640625e-07f  // 2^-20
RRT_DEV float exact_rcp(float x) {
#ifdef RRT_RCP_RANGE
    if (__builtin_fabsf(x) >= 7.888609052210118e-31f && __builtin_fabsf(x) <= 1.2676506002282294e30f) {
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r0, 1.0f);
        return __builtin_fmaf(e, r0, r0);
    }
    return 1.0f / x;
#else
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    if (__builtin_fabsf(e) <= RRT_RCP_RESIDUAL) return __builtin_fmaf(e, r0, r0);
    return 1.0f / x;
#endif
}

// ---- primitives ----------------------------------------------------------------------------------------------
// Triangle::hit (shape.rs:335-372) on a slot holding v0, e1 = v1 - v0, e2 = v2 - v0.
RRT_DEV bool triangle_hit(f3 v0, f3 e1, f3 e2, bool cull, const Ray& r, float t_min, float t_max, float* t_out) {
    f3 p_vec = cross3(r.d, e2);
    float det = dot3(e1, p_vec);
    if (!cull && det > -RRT_EPS && det < RRT_EPS) return false;
    else if (cull && det < RRT_EPS) return false;
    float inv_det = exact_rcp(det);
    f3 t_vec = r.o - v0;
    float u = dot3(t_vec, p_vec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 q_vec = cross3(t_vec, e1);
    float v = dot3(r.d, q_vec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t_hit = dot3(e2, q_vec) * inv_det;
    if (t_hit < t_max && t_hit > t_min) {
        *t_out = t_hit;
        return true;
    }
    return false;
}
// The same test, also returning the determinant and barycentrics it computed; Triangle::get_hit_properties
// (shape.rs:384-400) recomputes exactly these values from the same inputs, so they can be carried instead.
RRT_DEV bool triangle_hit_uv(f3 v0, f3 e1, f3 e2, bool cull, const Ray& r, float t_min, float t_max, float* t_out, float* u_out, float* v_out,
                             float* det_out) {
    f3 p_vec = cross3(r.d, e2);
    float det = dot3(e1, p_vec);
    if (!cull && det > -RRT_EPS && det < RRT_EPS) return false;
    else if (cull && det < RRT_EPS) return false;
    float inv_det = exact_rcp(det);
    f3 t_vec = r.o - v0;
    float u = dot3(t_vec, p_vec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 q_vec = cross3(t_vec, e1);
    float v = dot3(r.d, q_vec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t_hit = dot3(e2, q_vec) * inv_det;
    if (t_hit < t_max && t_hit > t_min) {
        *t_out = t_hit;
        *u_out = u;
        *v_out = v;
        *det_out = det;
        return true;
    }
    return false;
}
// &Matrix4 * &Ray (matrix.rs:154-190) with a row-major 3x4
template <class P>
RRT_DEV Ray xform_ray(P m, const Ray& r) {
    Ray o;
    o.o = mk3(m[0] * r.o.x + m[1] * r.o.y + m[2] * r.o.z + m[3] * 1.0f, m[4] * r.o.x + m[5] * r.o.y + m[6] * r.o.z + m[7] * 1.0f,
              m[8] * r.o.x + m[9] * r.o.y + m[10] * r.o.z + m[11] * 1.0f);
    o.d = mk3(m[0] * r.d.x + m[1] * r.d.y + m[2] * r.d.z, m[4] * r.d.x + m[5] * r.d.y + m[6] * r.d.z,
              m[8] * r.d.x + m[9] * r.d.y + m[10] * r.d.z);
    return o;
}
template <class P>
RRT_DEV f3 xform_vec(P m, f3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
template <class P>
RRT_DEV f3 xform_point(P m, f3 p) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0f, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0f,
               m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0f);
}
// Sphere::hit (shape.rs:65-85)
template <class P>
RRT_DEV bool sphere_hit_m(P w2l, float radius, const Ray& r, float t_min, float t_max, float* t_out) {
    Ray lr = xform_ray(w2l, r);
    f3 oc = lr.o - mk3(0.0f, 0.0f, 0.0f);
    float a = dot3(lr.d, lr.d);
    float b = 2.0f * dot3(oc, lr.d);
    float c = dot3(oc, oc) - (radius * radius);
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float t_hit = (-b - __builtin_sqrtf(disc)) / (2.0f * a);
        if (t_hit >= t_max || t_hit <= t_min) t_hit = (-b + __builtin_sqrtf(disc)) / (2.0f * a);
        if (t_hit < t_max && t_hit > t_min) {
            *t_out = t_hit;
            return true;
        }
    }
    return false;
}
RRT_DEV bool sphere_hit(csphere_p s, const Ray& r, float t_min, float t_max, float* t_out) {
    return sphere_hit_m((cf1p)s->w2l, s->radius, r, t_min, t_max, t_out);
}
// The boundary shape of a medium: a sphere or a single triangle.
RRT_DEV bool boundary_hit(const DScene& sc, cmedium_p m, const Ray& r, float t_min, float t_max, float* t_out) {
    if (m->boundary_kind == RRT_KIND_SPHERE) return sphere_hit(sc.c_spheres() + m->boundary_index, r, t_min, t_max, t_out);
    const cv4p bt = sc.c_btris() + 3 * m->boundary_index;
    const v4f q0 = bt[0], q1 = bt[1], q2 = bt[2];
    return triangle_hit(xyz(q0), xyz(q1), xyz(q2), (__float_as_uint(q1.w) & RRT_INFO_CULL) != 0, r, t_min, t_max, t_out);
}
// ConstantMedium::hit (volume.rs:65-102); xi is the uniform draw of line 95
RRT_DEV bool medium_hit(const DScene& sc, cmedium_p m, const Ray& r, float t_min, float t_max, float xi, float* t_out) {
    float t1, t2;
    if (!boundary_hit(sc, m, r, -RRT_T_MAX, RRT_T_MAX, &t1)) return false;
    if (!boundary_hit(sc, m, r, t1 + RRT_EPS, RRT_T_MAX, &t2)) return false;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    float distance_inside = (t2 - t1) * len3(r.d);
    float hit_dist = (-1.0f / m->density) * det_log(xi);
    if (hit_dist > distance_inside) return false;
    *t_out = t1 + (hit_dist / len3(r.d));
    return true;
}

// ---- bounding boxes (aggregate.rs:471-503) ---------------------------------------------------------------------
RRT_DEV void slab(float lo, float hi, float o, float inv, float* t0, float* t1) {
    float a = (lo - o) * inv;
    float b = (hi - o) * inv;
    bool neg = inv < 0.0f;
    *t0 = neg ? b : a;
    *t1 = neg ? a : b;
}
// AABB::intersect (aggregate.rs:471-503), the reference's test verbatim: an exit is honoured only against the running
// ENTRY bound (`t1 < t_min`), so many boxes the ray misses are still accepted — and a few it hits are rejected, which the
// reference's images depend on.  Both traversal modes apply it to the reference's boxes; TIGHT additionally requires
// box_overlap() with the node's primitive box (see closest_hit).
RRT_DEV bool aabb_reference(f3 bmin, f3 bmax, f3 o, f3 inv, float t_min, float t_max) {
    float t0x, t1x, t0y, t1y, t0z, t1z;
    slab(bmin.x, bmax.x, o.x, inv.x, &t0x, &t1x);
    slab(bmin.y, bmax.y, o.y, inv.y, &t0y, &t1y);
    slab(bmin.z, bmax.z, o.z, inv.z, &t0z, &t1z);
    float lo = t0x > t_min ? t0x : t_min;
    float hi = t1x < t_min ? t1x : t_max;
    if (hi <= lo) return false;
    float lo2 = t0y > lo ? t0y : lo;
    float hi2 = t1y < lo ? t1y : hi;
    if (hi2 <= lo2) return false;
    float lo3 = t0z > lo2 ? t0z : lo2;
    float hi3 = t1z < lo2 ? t1z : hi2;
    return !(hi3 <= lo3);
}

// Does the ray's (t_min, t_max) span overlap the box?  The ordinary three-slab interval test; min/max drop NaNs (0 * inf
// for a ray lying in a face plane), so such an axis constrains nothing and the test errs towards accepting.
RRT_DEV bool box_overlap(f3 lo, f3 hi, f3 o, f3 inv, float t_min, float t_max) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    const float t_near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), t_min));
    const float t_far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fminf(__builtin_fmaxf(az, bz), t_max));
    return t_near <= t_far;
}

// The same interval test against a pair record's box, which is stored interleaved (lo.x, hi.x | lo.y, hi.y | lo.z, hi.z) so
// that both planes of an axis take one packed fused multiply-add: plane * inv + noi with noi = -(origin * inv).  Compared
// with (plane - origin) * inv the result moves by a few ulps of |origin * inv|, i.e. by a few 1e-7 of the origin's
// coordinate in space, far inside the padding of the boxes (1e-5 of the scene's extent, camera position included).
typedef float v2f_ __attribute__((ext_vector_type(2)));
RRT_DEV bool pair_box_overlap(v2f_ x, v2f_ y, v2f_ z, f3 inv, f3 noi, float t_min, float t_max) {
    const v2f_ tx = __builtin_elementwise_fma(x, v2f_{inv.x, inv.x}, v2f_{noi.x, noi.x});
    const v2f_ ty = __builtin_elementwise_fma(y, v2f_{inv.y, inv.y}, v2f_{noi.y, noi.y});
    const v2f_ tz = __builtin_elementwise_fma(z, v2f_{inv.z, inv.z}, v2f_{noi.z, noi.z});
    const float t_near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), t_min));
    const float t_far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fminf(__builtin_fmaxf(tz.x, tz.y), t_max));
    return t_near <= t_far;
}

// Third traversal mode, chosen by the library for trees built with RRT_AGGREGATE_BVH_CLEAN (never by the caller): nodes are
// tested with the ordinary interval test on their primitive boxes alone, and equal distances go to the shape that comes
// first in SCENE order - together exactly what the reference's List aggregate returns (aggregate.rs:142-165).
constexpr int kAabbClean = 2;
constexpr uint32_t kLinkLeaf = 0x80000000u;  // clean trees: a child link that names a leaf's pair records directly

// ---- scene features a kernel variant is compiled for ------------------------------------------------------------------
// The host picks the smallest variant that covers the scene, so scenes without media / image textures / bump maps do not
// carry that code (registers, instruction cache) through the inner loop.
enum : uint32_t {
    F_SPHERE = 1u,  // sphere primitives (or sphere lights)
    F_MEDIUM = 2u,  // ConstantMedium primitives and the Isotropic phase function
    F_TEX = 4u,     // textures other than Constant (needs uv / world point in texture lookups)
    F_BUMP = 8u,    // bump maps (needs pu, pv)
    F_ALL = 15u
};


// Cost-measurement builds only (-DRRT_DUP=<bits>): the named piece is computed a second time on inputs the compiler
// must treat as new, and the copy's result is kept alive; results are unchanged, the time difference is the piece's cost.
#ifndef RRT_DUP
#define RRT_DUP 0
#endif
#define RRT_OPAQUE(x) asm volatile("" : "+v"(x))
#define RRT_KEEP(x) asm volatile("" ::"v"(x))

#ifdef RRT_PHASE_PROFILE
// profiling build only: per-lane node visits, per-lane pair tests, wave-level node steps, wave-level pair steps
__device__ unsigned long long g_prof[8];  // [4..7]: sift phase (filter steps, items, dense groups, phases), counted per wave
#endif

// ---- closest hit: BVH::hit (aggregate.rs:324-401) with List::hit (aggregate.rs:144-165) in the leaves ------------
struct Hit {
    int slot;  // -1 = miss
    float t;
    float u, v, det;  // triangle hits: the values Triangle::hit computed (reused by the hit properties)
};

// One primitive of a leaf against the ray: updates `h` when it is hit closer than h.t (strict, first visited wins ties).
template <uint32_t FEAT>
RRT_DEV void test_slot(const DScene& sc, uint32_t s, const Ray& r, float t_min, const Rng& key, uint32_t depth, Hit& h) {
    const cv4p cslots = sc.c_slots();
    const v4f q0 = cslots[3 * s], q1 = cslots[3 * s + 1], q2 = cslots[3 * s + 2];
    const uint32_t info = __float_as_uint(q1.w);
    const uint32_t kind = RRT_INFO_KIND(info);
    if (!(FEAT & (F_SPHERE | F_MEDIUM)) || kind == RRT_KIND_TRIANGLE) {
        float t, u, v, det;
        if (triangle_hit_uv(xyz(q0), xyz(q1), xyz(q2), (info & RRT_INFO_CULL) != 0, r, t_min, h.t, &t, &u, &v, &det)) {
            h.t = t;
            h.slot = (int)s;
            h.u = u;
            h.v = v;
            h.det = det;
        }
    } else if ((FEAT & F_SPHERE) && kind == RRT_KIND_SPHERE) {
        float t;
        if (sphere_hit(sc.c_spheres() + __float_as_uint(q2.w), r, t_min, h.t, &t)) {
            h.t = t;
            h.slot = (int)s;
        }
    } else if (FEAT & F_MEDIUM) {
        float t;
        float xi = rng_keyed(key, depth, __float_as_uint(q0.w));
        if (medium_hit(sc, sc.c_media() + __float_as_uint(q2.w), r, t_min, h.t, xi, &t)) {
            h.t = t;
            h.slot = (int)s;
        }
    }
}

// Two consecutive triangles of a leaf at once.  The pair record interleaves the two triangles component by component
// (v0x_A, v0x_B, v0y_A, v0y_B, ...), so every multiply / add / subtract of Triangle::hit (shape.rs:335-372) is one packed
// two-lane f32 instruction; each lane of a packed operation rounds exactly like the scalar one, and the two results are
// accepted in leaf order (A, then B against the distance A may have shortened), so the outcome equals two single tests.
typedef float v2f __attribute__((ext_vector_type(2)));
RRT_DEV v2f splat2(float x) { return v2f{x, x}; }
constexpr uint32_t kPairQuads = 8;  // float4 per pair record: six of primitive data, then the padded box of the two primitives (lo, hi)
struct PairEval {
    v2f t, u, v, det;
    bool ok_a, ok_b;  // every acceptance test of Triangle::hit except `t < t_max`
};
// Returns false when both determinants were rejected (nothing else is computed then).
RRT_DEV bool triangle_pair_eval(v4f q0, v4f q1, v4f q2, v4f q3, v4f q4, const Ray& r, float t_min, PairEval& pe) {
    const v2f v0x = q0.xy, v0y = q0.zw, v0z = q1.xy, e1x = q1.zw, e1y = q2.xy, e1z = q2.zw, e2x = q3.xy, e2y = q3.zw, e2z = q4.xy;
    const uint32_t info_a = __float_as_uint(q4.z), info_b = __float_as_uint(q4.w);
    const v2f dx = splat2(r.d.x), dy = splat2(r.d.y), dz = splat2(r.d.z);
    // p_vec = dir x edge_2 ; determinant = edge_1 . p_vec
    const v2f px = (dy * e2z) - (dz * e2y);
    const v2f py = (dz * e2x) - (dx * e2z);
    const v2f pz = (dx * e2y) - (dy * e2x);
    const v2f det = ((e1x * px) + (e1y * py)) + (e1z * pz);
    // the cull flags are wave-uniform in the uniform leaf walk; the common all-culled pair takes a scalar branch to the plain comparisons
    bool ok_a, ok_b;
    if (info_a & info_b & RRT_INFO_CULL) {
        asm volatile("");  // keeps the two arms apart (merged, they cost a per-lane select each)
        ok_a = !(det.x < RRT_EPS);
        ok_b = !(det.y < RRT_EPS);
    } else {
        ok_a = (info_a & RRT_INFO_CULL) ? !(det.x < RRT_EPS) : !(__builtin_fabsf(det.x) < RRT_EPS);
        ok_b = (info_b & RRT_INFO_CULL) ? !(det.y < RRT_EPS) : !(__builtin_fabsf(det.y) < RRT_EPS);
    }
    pe.ok_a = pe.ok_b = false;
    if (!(ok_a || ok_b)) return false;
    // 1 / determinant for both triangles: reciprocal estimate + one fused Newton step (see exact_rcp); when the residual of
    // either is not small (zero, denormal, huge or non-finite determinant) the whole wave takes the division instead
    const v2f r0 = v2f{__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
    const v2f e = __builtin_elementwise_fma(-det, r0, splat2(1.0f));
    v2f inv = __builtin_elementwise_fma(e, r0, r0);
    if (__builtin_expect(__ballot(!(__builtin_fabsf(e.x) <= RRT_RCP_RESIDUAL) || !(__builtin_fabsf(e.y) <= RRT_RCP_RESIDUAL)) != 0ull, 0))
        inv = v2f{1.0f / det.x, 1.0f / det.y};
    const v2f tx = splat2(r.o.x) - v0x, ty = splat2(r.o.y) - v0y, tz = splat2(r.o.z) - v0z;
    const v2f u = (((tx * px) + (ty * py)) + (tz * pz)) * inv;
    // q_vec = t_vec x edge_1
    const v2f qx = (ty * e1z) - (tz * e1y);
    const v2f qy = (tz * e1x) - (tx * e1z);
    const v2f qz = (tx * e1y) - (ty * e1x);
    const v2f v = (((dx * qx) + (dy * qy)) + (dz * qz)) * inv;
    const v2f uv = u + v;
    const v2f t = (((e2x * qx) + (e2y * qy)) + (e2z * qz)) * inv;
    // acceptance in the reference's order of tests; the comparisons are written so that NaNs fall the same way
    pe.ok_a = ok_a & !(u.x < 0.0f) & !(u.x > 1.0f) & !(v.x < 0.0f) & !(uv.x > 1.0f) & (t.x > t_min);
    pe.ok_b = ok_b & !(u.y < 0.0f) & !(u.y > 1.0f) & !(v.y < 0.0f) & !(uv.y > 1.0f) & (t.y > t_min);
    pe.t = t;
    pe.u = u;
    pe.v = v;
    pe.det = det;
    return true;
}
RRT_DEV void triangle_pair_hit(v4f q0, v4f q1, v4f q2, v4f q3, v4f q4, v4f q5, const Ray& r, float t_min, Hit& h) {
    PairEval pe;
    if (!triangle_pair_eval(q0, q1, q2, q3, q4, r, t_min, pe)) return;
    // A first, then B against the distance A may have shortened (leaf order)
    const bool take_a = pe.ok_a && pe.t.x < h.t;
    const bool take_b = pe.ok_b && pe.t.y < (take_a ? pe.t.x : h.t);
    if (take_a) {
        h.t = pe.t.x;
        h.slot = (int)__float_as_uint(q5.x);
        h.u = pe.u.x;
        h.v = pe.v.x;
        h.det = pe.det.x;
    }
    if (take_b) {
        h.t = pe.t.y;
        h.slot = (int)__float_as_uint(q5.y);
        h.u = pe.u.y;
        h.v = pe.v.y;
        h.det = pe.det.y;
    }
}

#define RRT_KIND_NONE 3u  // second half of an odd leaf's last pair

template <uint32_t FEAT>
RRT_DEV void test_pair(const DScene& sc, cv4p rec, const Ray& r, float t_min, const Rng& key, uint32_t depth, Hit& h) {
    const v4f q4 = rec[4], q5 = rec[5];
    const uint32_t kind_a = RRT_INFO_KIND(__float_as_uint(q4.z)), kind_b = RRT_INFO_KIND(__float_as_uint(q4.w));
    if (kind_a == RRT_KIND_TRIANGLE && kind_b == RRT_KIND_TRIANGLE) {
        triangle_pair_hit(rec[0], rec[1], rec[2], rec[3], q4, q5, r, t_min, h);
    } else {
        test_slot<FEAT>(sc, __float_as_uint(q5.x), r, t_min, key, depth, h);
        if (kind_b != RRT_KIND_NONE) test_slot<FEAT>(sc, __float_as_uint(q5.y), r, t_min, key, depth, h);
    }
}

// `stack` arguments below point at the calling lane's first entry of an LDS array laid out [entry][thread]; `stride` =
// threads per block.  Leaves are stored as pair records (kPairQuads x float4 per two primitives); on the device a leaf
// node's `a` is its first pair and the low bits of `b` its number of pairs.  The two primitives of a pair are in leaf
// order (A before B) but pairs need not be: every comparison below orders candidates by (distance, slot), and slots are
// numbered in leaf order.
#ifdef RRT_PHASE_PROFILE
#define RRT_COUNT(x, n) x += (n)
#else
#define RRT_COUNT(x, n)
#endif

// monotone map float -> uint (all non-NaN values), and back
RRT_DEV uint32_t ord_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
RRT_DEV float ord_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

// ---- closest hit for deep trees: wave-cooperative leaves ------------------------------------------------------------
// On a large mesh the lanes of a wave sit in different leaves of very different sizes (1 to 50+ primitives in the
// reference's trees), and a per-lane leaf loop makes the whole wave wait for every visit.  Here a lane that reaches a leaf
// only notes it (first pair, pair count) and pauses; when enough lanes are paused (or nobody can walk on) the wave
// flattens all noted (lane, pair) items into one list and tests them 64 at a time, every lane taking ANY owner's pair:
// it fetches the owner's ray through the lane crossbar, runs the ordinary pair test against the owner's closest distance,
// and reduces with a 64-bit minimum on (distance bits, slot) in the wave's LDS area — the smallest distance wins,
// equal distances go to the earlier slot, which is what testing the leaf front to back does.  The walk order of every
// lane is untouched, so results are those of closest_hit.
//
// `coop` is this wave's LDS area: 64 x u64 keys followed by 64 x (slot, det, u, v).
constexpr uint32_t kCoopBytesPerWave = 64 * 8 + 64 * 16;
#ifndef RRT_COOP_LANES
#define RRT_COOP_LANES 8
#endif
constexpr uint32_t kCoopLanes = RRT_COOP_LANES;  // paused lanes that trigger a cooperative leaf phase

template <uint32_t FEAT, bool CLEAN>
RRT_DEV void coop_leaves(const DScene& sc, const Ray& r, float t_min, const Rng& key, uint32_t depth, bool pending, uint32_t first_pair, uint32_t n_pairs,
                         Hit& h, unsigned char* coop) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* best = (unsigned long long*)coop;
    v4f* payload = (v4f*)(coop + 64 * 8);
    // inclusive prefix sum of the pair counts over the wave (lanes that are not in this call contribute nothing)
    const uint32_t mine = pending ? n_pairs : 0u;
    uint32_t incl = mine;
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    // Ties: in the reference's tree the leaf visited first wins, so what the owner already holds gets position 0; in a clean
    // tree the shape that is first in scene order wins wherever it sits, so positions are shape numbers throughout.
    uint32_t held = 0u;
    if (CLEAN && h.slot >= 0) held = __float_as_uint(sc.c_slots()[3 * h.slot].w) + 1u;
    const unsigned long long init = ((unsigned long long)ord_bits(h.t) << 32) | held;
    best[lane] = init;
    __threadfence_block();
    const cv4p cpairs = sc.c_pairs();
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t item = base + lane;
        const bool have = item < total;
        // owner = first lane whose inclusive sum exceeds the item index
        uint32_t lo = 0, hi = 63;
        for (int it = 0; it < 6; ++it) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t v = (uint32_t)__shfl((int)incl, (int)mid);
            if (v <= item) lo = mid + 1; else hi = mid;
        }
        const int owner = (int)(have ? lo : lane);
        const uint32_t o_incl = (uint32_t)__shfl((int)incl, owner), o_n = (uint32_t)__shfl((int)mine, owner);
        const uint32_t o_first = (uint32_t)__shfl((int)first_pair, owner);
        Ray ro;
        ro.o = mk3(__shfl(r.o.x, owner), __shfl(r.o.y, owner), __shfl(r.o.z, owner));
        ro.d = mk3(__shfl(r.d.x, owner), __shfl(r.d.y, owner), __shfl(r.d.z, owner));
        const float o_t = __shfl(h.t, owner);
        Rng okey = key;
        uint32_t odepth = depth;
        if (FEAT & F_MEDIUM) {  // the medium's distance draw is keyed by the owner's stream and depth
            okey.s0 = (uint32_t)__shfl((int)key.s0, owner);
            okey.s1 = (uint32_t)__shfl((int)key.s1, owner);
            odepth = (uint32_t)__shfl((int)depth, owner);
        }
        Hit hh;
        hh.slot = -1;
        hh.t = o_t;
        if (CLEAN && __shfl(h.slot, owner) >= 0) hh.t = ord_float(ord_bits(o_t) + 1u);  // an equal distance may still win on shape order
        hh.u = hh.v = hh.det = 0.0f;
        unsigned long long mykey = ~0ull;
        if (have) {
            const uint32_t k = item - (o_incl - o_n);  // position of the pair in its leaf
            const cv4p rec = cpairs + kPairQuads * (o_first + k);
            test_pair<FEAT>(sc, rec, ro, t_min, okey, odepth, hh);
            if (hh.slot >= 0) {
                uint32_t pos = (uint32_t)hh.slot;
                if (CLEAN) {
                    const v4f q5 = rec[5];  // (slot A, slot B, shape A, shape B)
                    pos = __float_as_uint(pos == __float_as_uint(q5.x) ? q5.z : q5.w);
                }
                mykey = ((unsigned long long)ord_bits(hh.t) << 32) | (unsigned long long)(pos + 1u);
                atomicMin(&best[owner], mykey);
            }
        }
        __threadfence_block();
        if (hh.slot >= 0 && best[owner] == mykey) payload[owner] = v4f{__uint_as_float((uint32_t)hh.slot), hh.det, hh.u, hh.v};
        __threadfence_block();
    }
    if (pending && best[lane] != init) {
        const v4f w = payload[lane];
        h.t = ord_float((uint32_t)(best[lane] >> 32));
        h.slot = (int)__float_as_uint(w.x);
        h.det = w.y;
        h.u = w.z;
        h.v = w.w;
    }
}

// One lane's walk through a deep tree, kept between steps so that callers can interleave the lanes of a wave freely.
struct Walk {
    uint32_t sp, cur;
    bool running;  // has a node to visit
    bool pending;  // stands at a leaf (first_pair, n_pairs) that the next cooperative phase tests
    uint32_t first_pair, n_pairs;
    f3 inv;
    Hit h;
};
RRT_DEV void walk_begin(const DScene& sc, Walk& w, bool active, const Ray& r, float t_max) {
    w.h.slot = -1;
    w.h.t = t_max;
    w.h.u = w.h.v = w.h.det = 0.0f;
    w.sp = 0;
    w.cur = sc.root_link;  // 0 for the reference's trees
    w.pending = false;
    w.first_pair = w.n_pairs = 0;
    w.running = active && sc.n_nodes != 0;
    w.inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
}
// Visits one node (BVH::hit, aggregate.rs:343-395): an inner node that passes its test pushes one child and descends into the
// other, a leaf that passes pauses the walk, anything else pops.
template <int AABB_MODE>
RRT_DEV void walk_step(const DScene& sc, Walk& w, const Ray& r, float t_min, uint32_t* stack, uint32_t stride) {
    const cv4p cnodes = sc.c_nodes();
    if (AABB_MODE == kAabbClean) {
        // One fetch brings both children's boxes: only children the ray meets are ever fetched themselves, and leaves are
        // not fetched at all (their pair range sits in the link).
        if (w.cur & kLinkLeaf) {
            w.pending = true;
            w.running = false;
            w.first_pair = w.cur & 0x0fffffffu;
            w.n_pairs = ((w.cur >> 28) & 7u) + 1u;
            return;
        }
        const v4f l0 = cnodes[4 * w.cur], l1 = cnodes[4 * w.cur + 1], r0 = cnodes[4 * w.cur + 2], r1 = cnodes[4 * w.cur + 3];
        const bool hit_l = box_overlap(xyz(l0), xyz(l1), r.o, w.inv, t_min, w.h.t);
        const bool hit_r = box_overlap(xyz(r0), xyz(r1), r.o, w.inv, t_min, w.h.t);
        const uint32_t link_l = __float_as_uint(l0.w), link_r = __float_as_uint(l1.w), axis = __float_as_uint(r0.w);
        const float dax = axis == 0u ? r.d.x : (axis == 1u ? r.d.y : r.d.z);
        const bool left_first = !(dax < 0.0f);  // the left child holds the lower centres: nearer when the ray runs upwards
        if (hit_l && hit_r) {
            stack[w.sp * stride] = left_first ? link_r : link_l;
            ++w.sp;
            w.cur = left_first ? link_l : link_r;
        } else if (hit_l || hit_r) {
            w.cur = hit_l ? link_l : link_r;
        } else if (w.sp == 0) {
            w.running = false;
        } else {
            --w.sp;
            w.cur = stack[w.sp * stride];
        }
        return;
    }
    const v4f n0 = cnodes[4 * w.cur];
    const v4f n1 = cnodes[4 * w.cur + 1];
    bool inside;
    if (sc.is_list) inside = true;
    else if (AABB_MODE == RRT_AABB_TIGHT) {
        const v4f b0 = cnodes[4 * w.cur + 2], b1 = cnodes[4 * w.cur + 3];
        inside = box_overlap(xyz(b0), xyz(b1), r.o, w.inv, t_min, w.h.t) && aabb_reference(xyz(n0), xyz(n1), r.o, w.inv, t_min, w.h.t);
    } else inside = aabb_reference(xyz(n0), xyz(n1), r.o, w.inv, t_min, w.h.t);
    const uint32_t a = __float_as_uint(n0.w), b = __float_as_uint(n1.w);
    if (inside && !(b & RRT_LEAF_FLAG)) {
        // aggregate.rs:378-392: with dir[axis] < 0 the left child (cur+1) is popped first, otherwise the right one
        const float dax = b == 0u ? r.d.x : (b == 1u ? r.d.y : r.d.z);
        const uint32_t left = w.cur + 1u, right = a;
        const bool left_first = dax < 0.0f;
        stack[w.sp * stride] = left_first ? right : left;
        ++w.sp;
        w.cur = left_first ? left : right;
    } else if (inside) {
        w.pending = true;  // the leaf is tested in the next cooperative phase; the walk resumes after it
        w.running = false;
        w.first_pair = a;
        w.n_pairs = b & ~RRT_LEAF_FLAG;
    } else if (w.sp == 0) {
        w.running = false;
    } else {
        --w.sp;
        w.cur = stack[w.sp * stride];
    }
}
// After the cooperative phase that tested this lane's leaf.
RRT_DEV void walk_resume(Walk& w, uint32_t* stack, uint32_t stride) {
    w.pending = false;
    if (w.sp != 0) {
        --w.sp;
        w.cur = stack[w.sp * stride];
        w.running = true;
    }
}

template <int AABB_MODE, uint32_t FEAT>
RRT_DEV Hit closest_hit_deep(const DScene& sc, bool active, const Ray& r, float t_min, float t_max, const Rng& key, uint32_t depth, uint32_t* stack,
                             uint32_t stride, unsigned char* coop) {
    // Must be called by all 64 lanes of the wave (the cooperative phase addresses lanes by number); lanes without a ray
    // pass active = false and only lend a hand.
    Walk w;
    walk_begin(sc, w, active, r, t_max);
#ifdef RRT_PHASE_PROFILE
    unsigned long long prof_nodes = 0, prof_pairs = 0, prof_steps = 0, prof_leaf_steps = 0;
#endif
    for (;;) {
        if (w.running) {
            RRT_COUNT(prof_nodes, 1);
            walk_step<AABB_MODE>(sc, w, r, t_min, stack, stride);
            if (w.pending) RRT_COUNT(prof_pairs, w.n_pairs);
        }
        const unsigned long long waiting = RRT_BALLOT(w.pending), walking = RRT_BALLOT(w.running);
#ifdef RRT_PHASE_PROFILE
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) ++prof_steps;
#endif
        if (waiting != 0ull && ((uint32_t)__popcll(waiting) >= kCoopLanes || walking == 0ull)) {
#ifdef RRT_PHASE_PROFILE
            if ((int)(threadIdx.x & 63u) == __ffsll((long long)__ballot(true)) - 1) ++prof_leaf_steps;
#endif
            coop_leaves<FEAT, AABB_MODE == kAabbClean>(sc, r, t_min, key, depth, w.pending, w.first_pair, w.n_pairs, w.h, coop);
            if (w.pending) walk_resume(w, stack, stride);
        } else if (waiting == 0ull && walking == 0ull) {
            break;
        }
    }
#ifdef RRT_PHASE_PROFILE
    atomicAdd(&g_prof[0], prof_nodes);
    atomicAdd(&g_prof[1], prof_pairs);
    atomicAdd(&g_prof[2], prof_steps);
    atomicAdd(&g_prof[3], prof_leaf_steps);
#endif
    return w.h;
}

// Recomputes (u, v, det) the way Triangle::get_hit_properties does (shape.rs:384-400) for callers without a Hit.
RRT_DEV void triangle_uvdet(f3 v0, f3 e1, f3 e2, const Ray& r, float* u, float* v, float* det) {
    f3 p_vec = cross3(r.d, e2);
    float d = dot3(e1, p_vec);
    float inv_det = 1.0f / d;
    f3 t_vec = r.o - v0;
    *u = dot3(t_vec, p_vec) * inv_det;
    f3 q_vec = cross3(t_vec, e1);
    *v = dot3(r.d, q_vec) * inv_det;
    *det = d;
}
// ---- closest hit for shallow trees: sift, then test densely ------------------------------------------------------------
// The reference's trees for small scenes are one or a few big leaves (its SAH rarely splits them), so every ray would run
// the full test against every primitive.  Here a lane that reaches a leaf pauses; once no lane is walking, the wave
//  1. sifts: each paused lane tests its ray against the padded box of every pair of its leaf (wave-uniform pair index, box
//     in scalar registers) and appends an item (pair, lane) to a list in LDS for every box its ray overlaps.  A pair whose
//     box the ray misses within (t_min, closest so far] holds nothing the ray can hit, so nothing is lost;
//  2. tests densely: the items are handed out 64 at a time, any lane taking any owner's item: it fetches the owner's ray
//     through the lane crossbar, runs the exact pair test and posts (distance, slot) with a 64-bit LDS
//     minimum - the smallest distance wins and equal distances go to the earlier primitive, which is what testing the
//     leaf front to back with a strict comparison does (aggregate.rs:144-165).
// The walk order of every lane is the reference's, so results equal closest_hit.
// `area` is this wave's LDS: 64 x u64 keys, then kSiftCap x u32 items (then 64 x kHarvestLights floats when harvesting).
//
// HARVEST (single-leaf scenes with a few lights).  After a diffuse bounce the reference evaluates the light mixture's
// density for the scattered ray, which costs one more intersection test per light (Shape::pdf, shape.rs:141-147, 441-458)
// - against the very primitives the NEXT closest-hit query of that same ray tests anyway.  With HARVEST the query also
// reports, per light, the distance at which the ray meets the light's own primitive (any distance in (t_min, t_max), whether
// or not something nearer hides it, and whether or not the leaf's box test let the ray in: Shape::pdf asks the shape, not
// the tree), and the caller finishes the previous bounce's throughput from that.  Same tests, same numbers, run once.
constexpr uint32_t kSiftCap = 512;
constexpr uint32_t kHarvestLights = 4;
constexpr uint32_t kSiftBytesPerWave = 64 * 8 + kSiftCap * 4;
constexpr uint32_t kHarvestBytesPerWave = 64 * kHarvestLights * 4;
#define RRT_PAIR_LIGHT(info) (((info) >> 24) & 15u)  // device-only bits of a pair record's info words: light number + 1

template <uint32_t FEAT, bool HARVEST>
RRT_DEV void sift_flush(const DScene& sc, const Ray& r, float t_min, float t_max, const Rng& key, uint32_t depth, const uint32_t* items, uint32_t n_items,
                        unsigned long long* best) {
    constexpr uint32_t kLaneShift = HARVEST ? 7 : 6;  // HARVEST items carry one more bit: the owner only wants its lights
    float* light_t = (float*)((unsigned char*)best + kSiftBytesPerWave);
    const uint32_t lane = threadIdx.x & 63u;
    const cv4p cpairs = sc.c_pairs();
    __threadfence_block();
    for (uint32_t base = 0; base < n_items; base += 64u) {
        const uint32_t idx = base + lane;
        const bool have = idx < n_items;
        const uint32_t item = have ? items[idx] : lane;
        const int owner = (int)(item & 63u);
        const uint32_t pair = item >> kLaneShift;
        const bool lights_only = HARVEST && (item & 64u) != 0u;
        Ray ro;
        ro.o = mk3(__shfl(r.o.x, owner), __shfl(r.o.y, owner), __shfl(r.o.z, owner));
        ro.d = mk3(__shfl(r.d.x, owner), __shfl(r.d.y, owner), __shfl(r.d.z, owner));
        Rng okey = key;
        uint32_t odepth = depth;
        if (FEAT & F_MEDIUM) {  // the medium's distance draw is keyed by the owner's stream and depth
            okey.s0 = (uint32_t)__shfl((int)key.s0, owner);
            okey.s1 = (uint32_t)__shfl((int)key.s1, owner);
            odepth = (uint32_t)__shfl((int)depth, owner);
        }
        if (have) {
            const cv4p rec = cpairs + kPairQuads * pair;
            const v4f q4 = rec[4];
            const uint32_t kind_a = RRT_INFO_KIND(__float_as_uint(q4.z)), kind_b = RRT_INFO_KIND(__float_as_uint(q4.w));
            const v4f q5 = rec[5];
            const uint32_t light_a = HARVEST ? RRT_PAIR_LIGHT(__float_as_uint(q4.z)) : 0u, light_b = HARVEST ? RRT_PAIR_LIGHT(__float_as_uint(q4.w)) : 0u;
            bool found = false, second = false, det_neg = false;
            float t_found = t_max;
            if (kind_a == RRT_KIND_TRIANGLE && kind_b == RRT_KIND_TRIANGLE) {
                PairEval pe;
                if (triangle_pair_eval(rec[0], rec[1], rec[2], rec[3], q4, ro, t_min, pe)) {
                    const bool ok_a = pe.ok_a && pe.t.x < t_max, ok_b = pe.ok_b && pe.t.y < t_max;
                    if (HARVEST && (light_a | light_b) != 0u) {  // Triangle::pdf's own hit test (shape.rs:446): this one
                        if (light_a != 0u && ok_a) light_t[(uint32_t)owner * kHarvestLights + light_a - 1u] = pe.t.x;
                        if (light_b != 0u && ok_b) light_t[(uint32_t)owner * kHarvestLights + light_b - 1u] = pe.t.y;
                    }
                    second = ok_b && !(ok_a && !(pe.t.y < pe.t.x));  // A keeps an equal distance (leaf order)
                    found = ok_a || ok_b;
                    t_found = second ? pe.t.y : pe.t.x;
                    det_neg = (second ? pe.det.y : pe.det.x) < 0.0f;
                }
            } else {
                Hit hh;
                hh.slot = -1;
                hh.t = t_max;
                hh.u = hh.v = hh.det = 0.0f;
                test_slot<FEAT>(sc, __float_as_uint(q5.x), ro, t_min, okey, odepth, hh);
                if (HARVEST && light_a != 0u && hh.slot >= 0) light_t[(uint32_t)owner * kHarvestLights + light_a - 1u] = hh.t;  // Sphere::pdf, shape.rs:142
                if (kind_b != RRT_KIND_NONE) {
                    const int before = hh.slot;
                    const float t_before = hh.t;
                    test_slot<FEAT>(sc, __float_as_uint(q5.y), ro, t_min, okey, odepth, hh);
                    second = hh.slot >= 0 && (before < 0 || hh.t < t_before);
                }
                found = hh.slot >= 0;
                t_found = hh.t;
                det_neg = hh.det < 0.0f;
            }
            if (found && !lights_only) {
                const uint32_t slot = __float_as_uint(second ? q5.y : q5.x);
                const uint32_t low = ((slot + 1u) << 1) | (det_neg ? 1u : 0u);
                atomicMin(&best[owner], ((unsigned long long)ord_bits(t_found) << 32) | (unsigned long long)low);
            }
        }
    }
    __threadfence_block();
}

template <int AABB_MODE, uint32_t FEAT, bool HARVEST = false>
RRT_DEV Hit closest_hit_sift(const DScene& sc, bool active, const Ray& r, float t_min, float t_max, const Rng& key, uint32_t depth, uint32_t* stack,
                             uint32_t stride, unsigned char* area, float* harvest = nullptr) {
    // Must be called by all 64 lanes of the wave (the dense phase addresses lanes by number); lanes without a ray pass
    // active = false and only lend a hand.  HARVEST: the tree is a single leaf; harvest[i] receives the distance at which the
    // ray meets light i's primitive, or -1.
    Hit h;
    h.slot = -1;
    h.t = t_max;
    h.u = h.v = h.det = 0.0f;
    if (sc.n_nodes == 0) return h;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* best = (unsigned long long*)area;
    uint32_t* items = (uint32_t*)(area + 64 * 8);
    const f3 inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    // for the pair boxes: plane * finv + noi (see pair_box_overlap); an axis whose product origin * inv is not finite (zero or
    // denormal direction component) gets NaN, which the interval test ignores: that axis then constrains nothing
    f3 finv = inv, noi = mk3(-(r.o.x * inv.x), -(r.o.y * inv.y), -(r.o.z * inv.z));
    if (!(__builtin_fabsf(noi.x) < __builtin_inff())) { finv.x = 0.0f; noi.x = __builtin_nanf(""); }
    if (!(__builtin_fabsf(noi.y) < __builtin_inff())) { finv.y = 0.0f; noi.y = __builtin_nanf(""); }
    if (!(__builtin_fabsf(noi.z) < __builtin_inff())) { finv.z = 0.0f; noi.z = __builtin_nanf(""); }
#ifdef RRT_PHASE_PROFILE
    unsigned long long prof_filter = 0, prof_items = 0, prof_groups = 0, prof_phases = 0;
#endif
    const cv4p cnodes = sc.c_nodes();
    const cv4p cpairs = sc.c_pairs();
    uint32_t sp = 0, cur = 0;
    bool running = active, pending = false;
    bool lights_only = false;  // HARVEST: the leaf's box test failed, so only the lights' own primitives are of interest
    uint32_t first_pair = 0, n_pairs = 0;
    float* light_t = (float*)(area + kSiftBytesPerWave);
    if (HARVEST)
        for (uint32_t i = 0; i < kHarvestLights; ++i) harvest[i] = -1.0f;
    for (;;) {
        if (running) {
            const v4f n0 = cnodes[4 * cur];
            const v4f n1 = cnodes[4 * cur + 1];
            bool inside;
            if (sc.is_list) inside = true;
            else if (AABB_MODE == RRT_AABB_TIGHT) {
                const v4f b0 = cnodes[4 * cur + 2], b1 = cnodes[4 * cur + 3];
                inside = box_overlap(xyz(b0), xyz(b1), r.o, inv, t_min, h.t) && aabb_reference(xyz(n0), xyz(n1), r.o, inv, t_min, h.t);
            } else inside = aabb_reference(xyz(n0), xyz(n1), r.o, inv, t_min, h.t);
            const uint32_t a = __float_as_uint(n0.w), b = __float_as_uint(n1.w);
            if (inside && !(b & RRT_LEAF_FLAG)) {
                // aggregate.rs:378-392: with dir[axis] < 0 the left child (cur+1) is popped first, otherwise the right one
                const float dax = b == 0u ? r.d.x : (b == 1u ? r.d.y : r.d.z);
                const uint32_t left = cur + 1u, right = a;
                const bool left_first = dax < 0.0f;
                stack[sp * stride] = left_first ? right : left;
                ++sp;
                cur = left_first ? left : right;
            } else if (inside || HARVEST) {
                pending = true;  // the leaf is tested in the next sift phase; the walk resumes after it
                running = false;
                lights_only = !inside;
                first_pair = a;
                n_pairs = b & ~RRT_LEAF_FLAG;
            } else if (sp == 0) {
                running = false;
            } else {
                --sp;
                cur = stack[sp * stride];
            }
        }
        if (RRT_BALLOT(running) != 0ull) continue;
        const unsigned long long waiting = RRT_BALLOT(pending);
        if (waiting == 0ull) break;
        best[lane] = (unsigned long long)ord_bits(h.t) << 32;  // position 0: what the owner already holds wins ties
        if (HARVEST)
            for (uint32_t i = 0; i < kHarvestLights; ++i) light_t[lane * kHarvestLights + i] = -1.0f;
        uint32_t n_items = 0;
        unsigned long long todo = waiting;
        while (todo != 0ull) {
            // the paused lanes are grouped by leaf (one group for the single-leaf trees of the small scenes)
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t leaf_first = (uint32_t)__builtin_amdgcn_readlane((int)first_pair, leader);
            const uint32_t leaf_pairs = (uint32_t)__builtin_amdgcn_readlane((int)n_pairs, leader);
            const bool mine = pending && first_pair == leaf_first;
            for (uint32_t k = 0; k < leaf_pairs; ++k) {
                const cv4p rec = cpairs + kPairQuads * (leaf_first + k);
                const v4f bx = rec[6], bz = rec[7];  // (lo.x, hi.x, lo.y, hi.y), (lo.z, hi.z, -, -)
                bool pass = mine && pair_box_overlap(bx.xy, bx.zw, bz.xy, finv, noi, t_min, h.t);
                if (HARVEST && __float_as_uint(bz.z) == 0u) pass = pass && !lights_only;  // bz.z: the pair holds a light's primitive
                const unsigned long long bal = RRT_BALLOT(pass);
                if (pass) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (HARVEST) items[n_items + rank] = ((leaf_first + k) << 7) | (lights_only ? 64u : 0u) | lane;
                    else items[n_items + rank] = ((leaf_first + k) << 6) | lane;
                }
                n_items += (uint32_t)__popcll(bal);
                RRT_COUNT(prof_filter, 1);
                if (n_items > kSiftCap - 64u) {
                    RRT_COUNT(prof_items, n_items);
                    RRT_COUNT(prof_groups, (n_items + 63u) / 64u);
                    sift_flush<FEAT, HARVEST>(sc, r, t_min, t_max, key, depth, items, n_items, best);
                    n_items = 0;
                }
            }
            todo &= ~RRT_BALLOT(mine);
        }
        RRT_COUNT(prof_items, n_items);
        RRT_COUNT(prof_groups, (n_items + 63u) / 64u);
        RRT_COUNT(prof_phases, 1);
        sift_flush<FEAT, HARVEST>(sc, r, t_min, t_max, key, depth, items, n_items, best);
        if (pending) {
            if (HARVEST)
                for (uint32_t i = 0; i < kHarvestLights; ++i) harvest[i] = light_t[lane * kHarvestLights + i];
            const unsigned long long w = best[lane];
            const uint32_t low = (uint32_t)w;
            if (low != 0u) {
                h.t = ord_float((uint32_t)(w >> 32));
                h.slot = (int)((low >> 1) - 1u);
                h.det = (low & 1u) ? -1.0f : 1.0f;  // only the sign is used (by the hit properties)
            }
            pending = false;
            if (sp != 0) {
                --sp;
                cur = stack[sp * stride];
                running = true;
            }
        }
    }
#ifdef RRT_PHASE_PROFILE
    if (lane == 0) {  // wave-level counts
        atomicAdd(&g_prof[4], prof_filter);
        atomicAdd(&g_prof[5], prof_items);
        atomicAdd(&g_prof[6], prof_groups);
        atomicAdd(&g_prof[7], prof_phases);
    }
#endif
    if ((FEAT & (F_TEX | F_BUMP)) && h.slot >= 0) {
        // Triangle::get_hit_properties recomputes the barycentrics from the ray (shape.rs:384-400); so does this path
        const cv4p cslots = sc.c_slots();
        const v4f q0 = cslots[3 * h.slot], q1 = cslots[3 * h.slot + 1], q2 = cslots[3 * h.slot + 2];
        if (RRT_INFO_KIND(__float_as_uint(q1.w)) == RRT_KIND_TRIANGLE) triangle_uvdet(xyz(q0), xyz(q1), xyz(q2), r, &h.u, &h.v, &h.det);
    }
    return h;
}

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
    const cmedium_p m = sc.c_media() + aux;  // volume.rs:104-106
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
RRT_DEV f3 sample_scatter_direction(const DScene& sc, f3 n, f3 from, Rng& g) {
    bool towards_light = false;
    uint32_t k = 0;
    if (sc.n_lights > 0) {
        float xi = rng_f32(g);
        if (xi < 0.5f) {
            towards_light = true;
            k = rng_index(g, sc.n_lights);
        }
    }
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
                   PendingPdf* pend = nullptr) {
    if (!(depth < max_depth && hit.slot >= 0)) {
        *radiance = background(bg, r);
        return true;
    }
    uint32_t mat_id;
    HitProps hp = slot_props<FEAT>(sc, hit, r, &mat_id);
    if (RRT_DUP & 2) {
        Ray r2 = r;
        RRT_OPAQUE(r2.o.x);
        uint32_t m2;
        HitProps h2 = slot_props<FEAT>(sc, hit, r2, &m2);
        RRT_KEEP(h2.p.x); RRT_KEEP(h2.p.y); RRT_KEEP(h2.p.z); RRT_KEEP(h2.n.x); RRT_KEEP(h2.n.y); RRT_KEEP(h2.n.z);
    }
    if (FEAT & (F_TEX | F_BUMP)) {
        hp.u = clampf(hp.u, 0.0f, 1.0f);  // aggregate.rs:31
        hp.v = clampf(hp.v, 0.0f, 1.0f);
    }
    const cmaterial_p mp = sc.c_materials() + mat_id;
    struct { uint32_t type; int32_t albedo, bump; float param; } m = {mp->type, mp->albedo, mp->bump, mp->param};
    if (m.type == RRT_MAT_DIFFUSE_LIGHT) {  // material.rs:229-231: emission ends the path
        *radiance = albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
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
        if (RRT_DUP & 64) {
            f3 u2 = unit_dir;
            RRT_OPAQUE(u2.x);
            float c2 = fmin_ref(dot3(-u2, n_use), 1.0f);
            float s2 = __builtin_sqrtf(1.0f - c2 * c2);
            f3 o2;
            if (eta * s2 > 1.0f) o2 = reflect_ref(u2, n_use);
            else if (schlick_ref(c2, eta) > 0.5f) o2 = reflect_ref(u2, n_use);
            else o2 = refract_ref(u2, n_use, eta);
            RRT_KEEP(o2.x); RRT_KEEP(o2.y); RRT_KEEP(o2.z);
        }
        thr = thr * mk3(1.0f, 1.0f, 1.0f);
        r.o = hp.p;
        r.d = out;
        return false;
    }
    if ((FEAT & F_MEDIUM) && m.type == RRT_MAT_ISOTROPIC) {  // volume.rs:27-35
        f3 out = unit_sphere_random(g);
        f3 att = albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
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
        f3 att = albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
        thr = thr * att;
        r.o = hp.p;
        r.d = out;
        return false;
    }
    // Lambert (material.rs:98-110) + the PDF branch of trace (aggregate.rs:57-90)
    f3 att = albedo_value<FEAT>(sc, m.albedo, hp.u, hp.v, hp.p);
    Ray scattered;
    scattered.o = hp.p;
    if (RRT_DUP & 4) {
        Rng g2 = g;
        RRT_OPAQUE(g2.s0);
        f3 d2 = sample_scatter_direction(sc, n, hp.p, g2);
        RRT_KEEP(d2.x); RRT_KEEP(d2.y); RRT_KEEP(d2.z);
    }
    scattered.d = sample_scatter_direction(sc, n, hp.p, g);
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
    if (RRT_DUP & 8) {
        Ray s2 = scattered;
        RRT_OPAQUE(s2.d.x);
        float m2 = mixture_value(sc, s2);
        RRT_KEEP(m2);
    }
    if (RRT_DUP & 16) {
        Ray s2 = scattered;
        RRT_OPAQUE(s2.d.x);
        float m2 = cosine_value(n, s2);
        RRT_KEEP(m2);
    }
    thr = (thr * (att * scatter_pdf)) / pdf_val;
    r = scattered;
    return false;
}

}  // namespace rrt
