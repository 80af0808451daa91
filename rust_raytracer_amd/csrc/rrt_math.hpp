// 3-vectors, the deterministic elementary functions and the per-path random stream of the path tracer (gfx950).
// Everything here is plain IEEE f32, compiled with -ffp-contract=off: the reference (Rust) never fuses a*b+c, and hit/miss
// decisions are only reproducible bit for bit with the same unfused operations in the same operand order.
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

}  // namespace rrt
