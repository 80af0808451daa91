// The device view of the uploaded scene, the samplers (utils.rs), Perlin noise and the textures (texture.rs).
// Each function names the reference code it computes (paths relative to the reference root).
#pragma once
#include "rrt_layout.hpp"
#include "rrt_math.hpp"

namespace rrt {

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
// The scene's arrays (SceneData, rrt_layout.hpp) as the kernels address them.
struct DScene : SceneData {
    RRT_DEV cv4p c_nodes() const { return (cv4p)(const void*)nodes; }
    RRT_DEV cv4p c_pairs() const { return (cv4p)(const void*)pairs; }
    RRT_DEV cv4p c_slots() const { return (cv4p)(const void*)slots; }
    RRT_DEV cv4p c_plain() const { return (cv4p)(const void*)plain; }
    RRT_DEV cv4p c_tri_shade() const { return (cv4p)(const void*)tri_shade; }
    RRT_DEV cv4p c_btris() const { return (cv4p)(const void*)btris; }
    RRT_DEV cv4p c_btri_shade() const { return (cv4p)(const void*)btri_shade; }
    RRT_DEV csphere_p c_spheres() const { return (csphere_p)(const void*)spheres; }
    RRT_DEV cmedium_p c_media() const { return (cmedium_p)(const void*)media; }
    RRT_DEV cmaterial_p c_materials() const { return (cmaterial_p)(const void*)materials; }
    RRT_DEV ctexture_p c_textures() const { return (ctexture_p)(const void*)textures; }
    RRT_DEV clight_p c_lights() const { return (clight_p)(const void*)lights; }
    RRT_DEV cv4p c_guards() const { return (cv4p)(const void*)guards; }
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
// (Not inlined: the shading code asks for a texture's value in seven places - three per bump map, one per material branch - and
// seven copies of the noise, turbulence, checker and image code made the textured kernel variants larger than the instruction
// cache.  One copy, called.)
__device__ __noinline__ f3 texture_value(const DScene& sc, int tex, float u, float v, f3 p) {
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

}  // namespace rrt
