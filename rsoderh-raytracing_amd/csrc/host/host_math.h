// host_math.h — f32 vector helpers for the host-side preprocessing.
// Plain Rust/glam f32 semantics: every * + - / is a single rounded operation, nothing fused
// (librsrt_host is compiled with -ffp-contract=off), operands in glam's published order.
#pragma once
#include <cmath>
#include <cstdint>

namespace rsrt_host {

struct Vec3 {
    float x, y, z;
    float operator[](unsigned i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 vec3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 vec3(const float *p) { return Vec3{p[0], p[1], p[2]}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 min3(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 max3(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
// glam Vec3::cross / dot (glam 0.30, src/f32/vec3.rs)
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline void store(float *p, Vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

constexpr float kF32Max = 3.40282347e+38f;

// Bounds3 (reference src/scene.rs:60-141)
struct Bounds3 {
    Vec3 min, max;
    static Bounds3 identity() { return {vec3(kF32Max, kF32Max, kF32Max), vec3(-kF32Max, -kF32Max, -kF32Max)}; }
    void grow(const Bounds3 &o) { min = min3(min, o.min); max = max3(max, o.max); }
    void grow(Vec3 p) { min = min3(min, p); max = max3(max, p); }
    Vec3 center() const { return min * 0.5f + max * 0.5f; }
    unsigned max_axis() const
    {
        Vec3 d = max - min;
        if (d.z > d.x && d.z > d.y) return 2;
        return d.y > d.x ? 1 : 0;
    }
    float surface_area() const
    {
        Vec3 d = max - min;
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
};

} // namespace rsrt_host
