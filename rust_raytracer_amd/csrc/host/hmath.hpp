// Host-side f32 math used while preparing a scene (transforms, camera, bounding boxes).
// Operand order follows the reference so that prepared data (world-space vertices, sphere
// matrices, camera frame, boxes) carry the bits the reference would compute:
// vector.rs:57-79, base.rs:29-43, matrix.rs:34-190, transform.rs:33-39.
#pragma once
#include <cmath>
#include <cstring>

namespace rrt_host {

struct Vec3 {
    float x, y, z;
    float operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }
};
inline Vec3 mk(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a) { return mk(-a.x, -a.y, -a.z); }
inline Vec3 operator*(Vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(float s, Vec3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator/(Vec3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline Vec3 min3(Vec3 a, Vec3 b) { return mk(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
inline Vec3 max3(Vec3 a, Vec3 b) { return mk(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline float squared_length(Vec3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
inline float length(Vec3 a) { return std::sqrt(squared_length(a)); }
inline Vec3 normalized(Vec3 a) { return a / length(a); }
inline Vec3 cross(Vec3 a, Vec3 b) { return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }

constexpr float kPi = 3.14159265358979323846f;
inline float to_radians(float deg) { return deg * (kPi / 180.0f); }

struct Mat4 {
    float m[4][4];
};
inline Mat4 identity() {
    Mat4 r;
    std::memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
    return r;
}
inline Mat4 mul(const Mat4& a, const Mat4& b) {  // matrix.rs:138-152
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline Mat4 translation(Vec3 t) {
    Mat4 r = identity();
    r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z;
    return r;
}
inline Mat4 scaling(Vec3 s) {
    Mat4 r = identity();
    r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z;
    return r;
}
// which: 0 = x, 1 = y, 2 = z (matrix.rs:42-76); angles in degrees
inline Mat4 rotation(int which, float degrees) {
    float sn = std::sin(to_radians(degrees));
    float cs = std::cos(to_radians(degrees));
    Mat4 r = identity();
    if (which == 0) { r.m[1][1] = cs; r.m[1][2] = -sn; r.m[2][1] = sn; r.m[2][2] = cs; }
    else if (which == 1) { r.m[0][0] = cs; r.m[0][2] = sn; r.m[2][0] = -sn; r.m[2][2] = cs; }
    else { r.m[0][0] = cs; r.m[0][1] = -sn; r.m[1][0] = sn; r.m[1][1] = cs; }
    return r;
}
// transform.rs:33-39 : T * Rx * Ry * Rz * S, multiplied left to right
inline Mat4 compose(Vec3 translate, Vec3 rotate, Vec3 scale) {
    Mat4 r = mul(translation(translate), rotation(0, rotate.x));
    r = mul(r, rotation(1, rotate.y));
    r = mul(r, rotation(2, rotate.z));
    return mul(r, scaling(scale));
}
// matrix.rs:88-134 ; false = "Singular matrix"
inline bool invert(const Mat4& src, Mat4* dst) {
    Mat4 t = src;
    Mat4 r = identity();
    for (int c = 0; c < 4; ++c) {
        if (t.m[c][c] == 0.0f) {
            int big = c;
            for (int row = 0; row < 4; ++row)
                if (std::fabs(t.m[row][c]) > std::fabs(t.m[big][c])) big = row;
            if (big == c) return false;
            for (int j = 0; j < 4; ++j) {
                float tmp = t.m[c][j]; t.m[c][j] = t.m[big][j]; t.m[big][j] = tmp;
                tmp = r.m[c][j]; r.m[c][j] = r.m[big][j]; r.m[big][j] = tmp;
            }
        }
        for (int row = 0; row < 4; ++row) {
            if (row == c) continue;
            float k = t.m[row][c] / t.m[c][c];
            if (k != 0.0f) {
                for (int j = 0; j < 4; ++j) {
                    t.m[row][j] -= k * t.m[c][j];
                    r.m[row][j] -= k * r.m[c][j];
                }
                t.m[row][c] = 0.0f;
            }
        }
    }
    for (int row = 0; row < 4; ++row)
        for (int c = 0; c < 4; ++c) r.m[row][c] /= t.m[row][row];
    *dst = r;
    return true;
}
inline Vec3 xform_point(const Mat4& a, Vec3 p) {  // matrix.rs:165-183
    return mk(a.m[0][0] * p.x + a.m[0][1] * p.y + a.m[0][2] * p.z + a.m[0][3] * 1.0f,
              a.m[1][0] * p.x + a.m[1][1] * p.y + a.m[1][2] * p.z + a.m[1][3] * 1.0f,
              a.m[2][0] * p.x + a.m[2][1] * p.y + a.m[2][2] * p.z + a.m[2][3] * 1.0f);
}

struct Box {
    Vec3 lo, hi;
};
inline Box origin_box() { return Box{mk(0, 0, 0), mk(0, 0, 0)}; }                              // aggregate.rs:423-428
inline Box merge(const Box& a, const Box& b) { return Box{min3(a.lo, b.lo), max3(a.hi, b.hi)}; }  // aggregate.rs:430-435
inline Box merge_point(const Box& a, Vec3 p) { return Box{min3(a.lo, p), max3(a.hi, p)}; }       // aggregate.rs:437-442
inline Vec3 centre(const Box& b) {                                                              // aggregate.rs:444-450
    return mk(b.lo.x * 0.5f + b.hi.x * 0.5f, b.lo.y * 0.5f + b.hi.y * 0.5f, b.lo.z * 0.5f + b.hi.z * 0.5f);
}
inline int widest_axis(const Box& b) {                                                          // aggregate.rs:452-461
    Vec3 d = b.hi - b.lo;
    if (d.x > d.y && d.x > d.z) return 0;
    return d.y > d.z ? 1 : 2;
}
inline float area(const Box& b) {                                                               // aggregate.rs:463-469
    Vec3 d = b.hi - b.lo;
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}

}  // namespace rrt_host
