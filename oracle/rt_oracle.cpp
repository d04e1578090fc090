/*
 * rt_oracle.cpp — CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY
 * (see rt_oracle.h).  PARITY UNPINNED BY THE REFERENCE (no reference tests / golden vectors
 * exist; pinned only by the hand-derivable KATs of SURVEY.md §8c).
 *
 * Conventions (the arithmetic contract the HIP kernels reproduce bit for bit):
 *  - IEEE-754 binary32 throughout, compiled with -ffp-contract=off: every written * + - /
 *    is one correctly-rounded operation, evaluated left to right as WGSL precedence dictates.
 *  - The ONLY fused operations are inside dot(), cross(), mat3*vec and madd() below (WGSL
 *    permits fusing; fixing where it happens makes CPU and GPU agree).
 *  - normalize(v) = v * (1/sqrt(dot(v,v))); inverseSqrt(x) = 1/sqrt(x); length = sqrt(dot).
 *  - sin/cos/atan2/asin = include/rsrt_detmath.h (the ABI's published numeric contract).  NB: that header is SHARED with the
 *    product — the kernels compile the very same routines — so a mistake in it is common to both sides and invisible to every
 *    oracle-vs-kernel parity test; it is covered only by tests/test_detmath.py (<= 3.1 ulp against a float64 libm on the
 *    ranges the integrator uses) and, end to end, by the float64 second reading in tests/test_independent_*.py.
 *  - max(a,b) = a<b?b:a ; min(a,b) = b<a?b:a ; saturate(x)=min(max(x,0),1) ; abs = sign clear.
 *  - u32(f32): NaN or <=0 -> 0, >= 4294967040 -> 4294967040 (naga's clamp), else truncation.
 *  - texture fetch = software bilinear with full f32 weights, a*(1-f)+b*f, clamp-to-edge
 *    (hardware filters use fewer weight bits — stated difference).
 */
#include "rt_oracle.h"
#include "../include/rsrt_detmath.h"

// -DORC_LIBM (liboracle_libm.so): the same restatement with the numerics an implementation is FREE to choose under
// WGSL chosen differently — the platform libm's sinf / cosf / atan2f / asinf instead of rsrt_detmath.h, and (by the
// Makefile) -ffp-contract=fast, so the compiler fuses multiply-adds wherever it likes.  It is NOT the parity
// reference; tests/test_numerics_sensitivity.py measures how far such an "honest but not bit-compatible"
// implementation lands from the strict one, i.e. what north_star's 1e-3 RMSE tolerance can and cannot absorb.
#ifdef ORC_LIBM
#include <cmath>
#define rsrt_sinf sinf
#define rsrt_cosf cosf
#define rsrt_atan2f atan2f
#define rsrt_asinf asinf
#endif

#ifdef ORC_COUNT_OPS // the detmath routines, counted by what their polynomial path executes (see OPS below)
static inline float orc_counted_sinf(float x);
static inline float orc_counted_cosf(float x);
static inline float orc_counted_atan2f(float y, float x);
static inline float orc_counted_asinf(float x);
#endif

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef uint32_t u32;

// -DORC_COUNT_OPS (liboracle_ops.so): the same restatement counting the f32 operations it executes — the ALGORITHMIC work of
// the reference's integrator, the numerator of bench.py's roofline fraction.  Rules: one per f32 add, sub, mul, div, fma, sqrt,
// floor, min, max and per f32 comparison; nothing for negation, abs, selects, moves and loads; the vector helpers count
// themselves (dot = 3, cross = 6, normalize = 8 ...), every function below adds what its own scalar expressions execute
// (the terms are written out next to the code), the detmath routines count the operations of their polynomial path
// (include/rsrt_detmath.h; the two polynomial branches of sin / cos are averaged).  u32 arithmetic (the PCG draw, texel and
// table indices) and u32 <-> f32 conversions are counted apart, as int_ops.  The counting build computes the same bits.
#ifdef ORC_COUNT_OPS
thread_local uint64_t g_f32_ops = 0, g_int_ops = 0; // (file scope through the unnamed namespace: the detmath wrappers below use them too)
#define OPS(n) (g_f32_ops += (uint64_t)(n))
#define IOPS(n) (g_int_ops += (uint64_t)(n))
#else
#define OPS(n) ((void)0)
#define IOPS(n) ((void)0)
#endif

#ifdef ORC_COUNT_OPS
} // namespace
// sin / cos: 2 comparisons, x * 4/pi, 3 mul + 3 sub of the reduction, r * r, polynomial 7 (sin) or 9 (cos): 8 on average = 17;
// atan2: 4 comparisons + y / x + w + atanf (3 comparisons, ~2 for the range reduction, z, 3 x (mul, add), mul mul add add) = 22;
// asin: 4 comparisons, ~2 (0.5 (1 - a), sqrt | x * x), 4 x (mul, add), mul mul add, ~1 (r + r, pi/2 - r) = 18
static inline float orc_counted_sinf(float x) { g_f32_ops += 17; g_int_ops += 6; return rsrt_sinf(x); }
static inline float orc_counted_cosf(float x) { g_f32_ops += 17; g_int_ops += 6; return rsrt_cosf(x); }
static inline float orc_counted_atan2f(float y, float x) { g_f32_ops += 22; return rsrt_atan2f(y, x); }
static inline float orc_counted_asinf(float x) { g_f32_ops += 18; return rsrt_asinf(x); }
#define rsrt_sinf orc_counted_sinf
#define rsrt_cosf orc_counted_cosf
#define rsrt_atan2f orc_counted_atan2f
#define rsrt_asinf orc_counted_asinf
namespace {
#endif

// ------------------------------------------------------------------ vector helpers
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { OPS(3); return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { OPS(3); return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { OPS(3); return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { OPS(3); return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { OPS(3); return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { OPS(3); return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline float comp(V3 a, u32 i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

inline float fmax_(float a, float b) { OPS(1); return a < b ? b : a; }
inline float fmin_(float a, float b) { OPS(1); return b < a ? b : a; }
inline float saturate(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }
inline float fabs_(float x) { return fabsf(x); }

// fused helpers (see header comment)
inline float dot(V3 a, V3 b) { OPS(3); return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline float dot2(float ax, float ay, float bx, float by) { OPS(2); return fmaf(ay, by, ax * bx); }
inline V3 cross(V3 a, V3 b)
{
    OPS(6); // 3 mul + 3 fma
    return V3{fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y))};
}
inline V3 madd(V3 d, float t, V3 o) { OPS(3); return V3{fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)}; }
struct M3 { V3 c0, c1, c2; };
inline V3 mul(const M3 &m, V3 v)
{
    OPS(9); // 3 x (mul + 2 fma)
    return V3{fmaf(m.c2.x, v.z, fmaf(m.c1.x, v.y, m.c0.x * v.x)), fmaf(m.c2.y, v.z, fmaf(m.c1.y, v.y, m.c0.y * v.x)),
              fmaf(m.c2.z, v.z, fmaf(m.c1.z, v.y, m.c0.z * v.x))};
}
inline float length(V3 a) { OPS(1); return rsrt_sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { OPS(2); return a * (1.0f / rsrt_sqrtf(dot(a, a))); } // sqrt + div (+ dot 3 + scale 3)
inline float inverse_sqrt(float x) { OPS(2); return 1.0f / rsrt_sqrtf(x); }

inline u32 f2u(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967040.0f) return 4294967040u;
    return (u32)x;
}

// ------------------------------------------------------------------ constants (shader.wgsl:232-240)
const float INFINITY_ = 1.70141183460469231732e+38f;
const float PI = (float)3.14159;            // shader.wgsl:239
const float INV_PI = (float)(1.0 / 3.14159); // :240, const-evaluated in abstract float
const float TWO_PI = (float)(2.0 * 3.14159); // `2 * PI` const-evaluated

struct Ray { V3 origin, direction; };
struct HitInfo { bool did_hit; float distance; V3 hit_point; V3 normal; u32 material_id; };
const HitInfo NO_HIT = {false, 0.0f, {0, 0, 0}, {0, 0, 0}, 0u}; // shader.wgsl:27-33

struct Stats : orc_stats {
    Stats() { memset(this, 0, sizeof(orc_stats)); }
    void add(const orc_stats &o)
    {
        const uint64_t *s = (const uint64_t *)&o;
        uint64_t *d = (uint64_t *)(orc_stats *)this;
        for (size_t i = 0; i < sizeof(orc_stats) / 8; i++) d[i] += s[i];
    }
};

struct Ctx {
    const orc_scene *sc;
    const orc_env *env;
    u32 flags;
    Stats *st;
};

// ------------------------------------------------------------------ RNG (shader.wgsl:605-631)
inline u32 random_u32_uniform(u32 *s)
{
    *s = *s * 747796405u + 2891336453u; // :612
    u32 result = ((*s >> ((*s >> 28) + 4u)) ^ *s) * 277803737u; // :613-614
    result = (result >> 22) ^ result;   // :615
    IOPS(9); // mul add | shr add shr xor mul | shr xor
    return result;
}
inline void salt_rng(u32 *s, u32 salt) { IOPS(1); *s = *s ^ salt; random_u32_uniform(s); } // :605-609
inline float random_uniform(u32 *s) { OPS(1); IOPS(1); return (float)random_u32_uniform(s) / 4294967295.0f; } // :621-623 (div; cvt)
inline void random_in_circle_uniform(u32 *s, float *ox, float *oy) // :627-631
{
    OPS(2 + 1 + 2); // angle: 2 mul; sqrt; cx * r, cy * r
    float angle = random_uniform(s) * 2.0f * 3.1415926f;
    float cx = rsrt_cosf(angle), cy = rsrt_sinf(angle);
    float r = rsrt_sqrtf(random_uniform(s));
    *ox = cx * r;
    *oy = cy * r;
}

// ------------------------------------------------------------------ intersection (shader.wgsl:262-466)
// :262-293.  `best` only used when ORC_FLAG_PRUNE is set (SURVEY A7 i).
inline bool ray_intersects_bounds(const Ray &ray, const orc_bvh_node &n, V3 inv, bool prune, float best)
{
    float t_0 = 0.0f, t_1 = INFINITY_;
    for (u32 axis = 0; axis < 3; axis++) {
        OPS(2 + 2 + 4); // 2 sub, 2 mul, 4 comparisons (swap, t_0, t_1, t_0 > t_1)
        float t_near = (n.bmin[axis] - comp(ray.origin, axis)) * comp(inv, axis);
        float t_far = (n.bmax[axis] - comp(ray.origin, axis)) * comp(inv, axis);
        if (t_near > t_far) { float o = t_near; t_near = t_far; t_far = o; }
        if (t_near > t_0) t_0 = t_near;
        if (t_far < t_1) t_1 = t_far;
        if (t_0 > t_1) return false;
    }
    if (prune) OPS(1);
    if (prune && t_0 > best) return false;
    return true;
}

// :295-360
inline HitInfo cast_ray_sphere(const Ray &ray, const orc_sphere &sp)
{
    const float EPSILON = 1.0e-4f;
    V3 pos = v3(sp.pos);
    V3 l = ray.origin - pos;
    float a = dot(ray.direction, ray.direction);
    float b = 2.0f * dot(ray.direction, l);
    float c = dot(l, l) - sp.radius * sp.radius;
    float t;
    OPS(1 + 2 + 4 + 1); // b: mul; c: mul sub; discriminant: 3 mul + sub; < 0
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) {
        return NO_HIT;
    } else if (discriminant == 0.0f) {
        OPS(1 + 2);
        t = -0.5f * b / a;
    } else {
        OPS(1 + 1 + 3 + 2 + 2); // == 0; sqrt; b > 0, add / sub, mul; 2 div; 2 comparisons (+ fmin_)
        float sq = rsrt_sqrtf(discriminant);
        float q = (b > 0.0f) ? -0.5f * (b + sq) : -0.5f * (b - sq); // select(f, t, cond), :311-315
        float t_0 = q / a;
        float t_1 = c / q;
        if (t_0 < EPSILON) t = t_1;
        else if (t_1 < EPSILON) t = t_0;
        else t = fmin_(t_0, t_1);
    }
    OPS(1);
    if (t < EPSILON) return NO_HIT;
    OPS(2 + 1); // radius^2, sub; < 1e-6
    V3 hit_point = madd(ray.direction, t, ray.origin);
    V3 normal = normalize(hit_point - pos);
    V3 co = pos - ray.origin;
    if (dot(co, co) - sp.radius * sp.radius < 1.0e-6f) normal = normal * -1.0f; // :346-351
    return HitInfo{true, t, hit_point, normal, sp.material_id};
}

// :362-406
inline HitInfo cast_ray_plane(const Ray &ray, const orc_plane &pl)
{
    V3 n = v3(pl.normal), pos = v3(pl.pos);
    float denominator = dot(n, ray.direction);
    OPS(1);
    if (fabs_(denominator) < 0.0001f) return NO_HIT;
    OPS(1 + 1); // div; t < 0.001
    float t = dot(n, pos - ray.origin) / denominator;
    if (t < 0.001f) return NO_HIT;
    V3 inter = madd(ray.direction, t, ray.origin);
    V3 inter_local = inter - pos;
    M3 m = {v3(pl.m[0]), v3(pl.m[1]), v3(pl.m[2])};
    V3 ps = mul(m, inter_local);
    OPS(4);
    if (ps.x < 0.0f || 1.0f < ps.x || ps.z < 0.0f || 1.0f < ps.z) return NO_HIT;
    OPS(1); // dot(origin, n) < 0
    V3 normal = n;
    if (dot(ray.origin, normal) < 0.0f) normal = normal * -1.0f; // :394 (origin NOT plane-relative)
    return HitInfo{true, t, inter, normal, pl.material_id};
}

// :409-466
inline HitInfo cast_ray_triangle(const Ray &ray, const orc_scene &sc, const orc_triangle &tr)
{
    V3 a = v3(sc.vertices[tr.v0].v), b = v3(sc.vertices[tr.v1].v), c = v3(sc.vertices[tr.v2].v);
    V3 pos = a;
    V3 edge_0 = b - a, edge_1 = c - a;
    V3 op = ray.origin - pos;
    V3 perp_to_edge_0 = cross(op, edge_0);
    V3 perp_to_edge_1 = cross(ray.direction, edge_1);
    float determinant = dot(edge_0, perp_to_edge_1);
    OPS(1 + 1); // div; |det| < 1e-8
    float inverse_determinant = 1.0f / determinant; // before the test, :421
    if (fabs_(determinant) < 1.0e-8f) return NO_HIT;
    OPS(2 + 2); // u, v: a mul each; u < 0, 1 < u
    float u = dot(op, perp_to_edge_1) * inverse_determinant;
    float v = dot(ray.direction, perp_to_edge_0) * inverse_determinant;
    if (u < 0.0f || 1.0f < u) return NO_HIT;
    if (u < 0.0f || 1.0f < u) {} else OPS(3); // v < 0, u + v, 1 < ..
    if (v < 0.0f || 1.0f < (u + v)) return NO_HIT;
    OPS(1 + 1); // t: mul; t < 1e-5
    float t = dot(edge_1, perp_to_edge_0) * inverse_determinant;
    if (t < 1.0e-5f) return NO_HIT;
    OPS(2 + 1); // 1 - u - v; dot(normal, dir) > 0
    V3 n0 = v3(sc.normals[tr.n0].v), n1 = v3(sc.normals[tr.n1].v), n2 = v3(sc.normals[tr.n2].v);
    V3 normal = normalize((1.0f - u - v) * n0 + u * n1 + v * n2);
    if (dot(normal, ray.direction) > 0.0f) normal = normal * -1.0f;
    return HitInfo{true, t, madd(ray.direction, t, ray.origin), normal, tr.material_id};
}

// :469-564.  `any_hit`: return at the first accepted primitive (only for did_hit consumers).
HitInfo cast_ray_bvh(const Ctx &cx, const Ray &ray, bool any_hit)
{
    const orc_scene &sc = *cx.sc;
    Stats &st = *cx.st;
    const bool prune = (cx.flags & ORC_FLAG_PRUNE) != 0;
    OPS(3);
    V3 inv = v3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    HitInfo result = {false, INFINITY_, {0, 0, 0}, {0, 0, 0}, 0u};
    u32 result_type = 0;
    u32 nodes_to_visit[64];
    u32 stack_length = 0;
    u32 current = 0;
    for (;;) {
        const orc_bvh_node &node = sc.nodes[current];
        st.nodes_visited++;
        if (ray_intersects_bounds(ray, node, inv, prune, result.distance)) {
            if (node.len > 0) {
                for (u32 i = 0; i < node.len; i++) {
                    const orc_prim_info &info = sc.prims[node.idx + i];
                    st.prim_refs++;
                    HitInfo h = NO_HIT;
                    switch (info.type) {
                    case 0: st.sphere_tests++; h = cast_ray_sphere(ray, sc.spheres[info.index]); break;
                    case 1: st.plane_tests++; h = cast_ray_plane(ray, sc.planes[info.index]); break;
                    case 2: st.tri_tests++; h = cast_ray_triangle(ray, sc, sc.triangles[info.index]); break;
                    default: break;
                    }
                    if (h.did_hit) OPS(1);
                    if (h.did_hit && h.distance < result.distance) {
                        result = h;
                        result_type = info.type;
                        if (any_hit) return result;
                    }
                }
                if (stack_length == 0) break;
                stack_length--;
                current = nodes_to_visit[stack_length];
            } else {
                OPS(1);
                if (comp(inv, node.axis) < 0.0f) { // :536
                    nodes_to_visit[stack_length++] = current + 1;
                    current = node.idx;
                } else {
                    nodes_to_visit[stack_length++] = node.idx;
                    current = current + 1;
                }
            }
        } else {
            if (stack_length == 0) break;
            stack_length--;
            current = nodes_to_visit[stack_length];
        }
    }
    if (!result.did_hit) return NO_HIT;
    if (result_type == 2) st.closest_tri++;
    return result;
}

// :567-601
HitInfo cast_ray(const Ctx &cx, const Ray &ray)
{
    const orc_scene &sc = *cx.sc;
    HitInfo result = {false, INFINITY_, {0, 0, 0}, {0, 0, 0}, 0u};
    {
        HitInfo r = cast_ray_bvh(cx, ray, false);
        if (r.did_hit) return r;
    }
    for (u32 i = 0; i < sc.n_spheres; i++) {
        cx.st->fallback_sphere_tests++;
        HitInfo h = cast_ray_sphere(ray, sc.spheres[i]);
        if (h.did_hit) OPS(1);
        if (h.did_hit && h.distance < result.distance) result = h;
    }
    for (u32 i = 0; i < sc.n_planes; i++) {
        cx.st->fallback_plane_tests++;
        HitInfo h = cast_ray_plane(ray, sc.planes[i]);
        if (h.did_hit) OPS(1);
        if (h.did_hit && h.distance < result.distance) result = h;
    }
    return result;
}

// ------------------------------------------------------------------ environment (shader.wgsl:667-831)
inline void direction_to_equirectangular_uv(V3 d, float *u, float *v) // :710-714
{
    OPS(3 + 2);
    *u = rsrt_atan2f(d.z, d.x) * INV_PI * 0.5f + 0.5f;
    *v = 0.5f - rsrt_asinf(d.y) * INV_PI;
}
inline V3 equirectangular_uv_to_direction(float u, float v) // :718-732
{
    OPS(3 + 1 + 2); // phi; theta; sin_theta * cos(phi), * sin(phi)
    float phi = (2.0f * u - 1.0f) * PI;
    float theta = PI * v;
    float sin_theta = rsrt_sinf(theta), cos_theta = rsrt_cosf(theta);
    return v3(sin_theta * rsrt_cosf(phi), cos_theta, sin_theta * rsrt_sinf(phi));
}
inline float environment_pixel_solid_angle(float v, const orc_env &e) // :739-749
{
    OPS(1 + 2 + 2); IOPS(2); // theta; 2 div; 2 mul; 2 cvt
    float theta = PI * v;
    float sin_t = fmax_(1.0e-6f, rsrt_sinf(theta));
    float d_phi = TWO_PI / (float)e.width;
    float d_theta = PI / (float)e.height;
    return d_phi * d_theta * sin_t;
}
inline u32 clamp_texel(float f, u32 n)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= (float)(n - 1)) return n - 1;
    return (u32)f;
}
// textureSampleLevel(.., uv, 0).xyz with the sampler of src/state.rs:134-142 (Appendix A3)
inline V3 sample_env_bilinear(const orc_env &e, float u, float v)
{
    OPS(4 + 2 + 2 + 2 + 8 + 2); IOPS(2 + 4 + 8); // x, y; floor; fx, fy; xf + 1, yf + 1; clamp comparisons; gx, gy | cvt; texel cvt; addresses
    float x = u * (float)e.width - 0.5f, y = v * (float)e.height - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    u32 x0 = clamp_texel(xf, e.width), x1 = clamp_texel(xf + 1.0f, e.width);
    u32 y0 = clamp_texel(yf, e.height), y1 = clamp_texel(yf + 1.0f, e.height);
    const float *t00 = e.rgba + 4 * ((size_t)y0 * e.width + x0), *t10 = e.rgba + 4 * ((size_t)y0 * e.width + x1);
    const float *t01 = e.rgba + 4 * ((size_t)y1 * e.width + x0), *t11 = e.rgba + 4 * ((size_t)y1 * e.width + x1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    V3 top = v3(t00) * gx + v3(t10) * fx;
    V3 bot = v3(t01) * gx + v3(t11) * fx;
    return top * gy + bot * fy;
}
inline V3 sky_light(const Ctx &cx, V3 dir) // :822-831
{
    float u, v;
    direction_to_equirectangular_uv(dir, &u, &v);
    return sample_env_bilinear(*cx.env, u, v);
}
inline float environment_direction_pdf(const Ctx &cx, V3 dir) // :753-769
{
    const orc_env &e = *cx.env;
    float u, v;
    direction_to_equirectangular_uv(dir, &u, &v);
    OPS(2 + 4 + 1); IOPS(4 + 4 + 2); // u * W, v * H; f2u's comparisons; pmf / ..
    u32 x = std::min(f2u(u * (float)e.width), e.width - 1);
    u32 y = std::min(f2u(v * (float)e.height), e.height - 1);
    u32 index = x + y * e.width;
    float pmf = e.alias[index].pmf;
    return pmf / environment_pixel_solid_angle(v, e);
}
inline u32 random_index_in_environment(const Ctx &cx, u32 *rng) // :689-706
{
    const orc_env &e = *cx.env;
    u32 length = e.width * e.height;
    OPS(1 + 2 + 1); IOPS(5); // u * N; f2u; u2 < probability
    u32 index = std::min(f2u(random_uniform(rng) * (float)length), length - 1);
    const orc_alias_entry &entry = e.alias[index];
    float u2 = random_uniform(rng); // select() evaluates all operands: always drawn
    return (u2 < entry.probability) ? index : entry.alias_index;
}
struct EnvironmentSample { V3 direction, radiance; float pdf; };
inline EnvironmentSample sample_environment(const Ctx &cx, u32 *rng) // :782-820
{
    const orc_env &e = *cx.env;
    u32 index = random_index_in_environment(cx, rng);
    u32 x = index % e.width, y = index / e.width;
    float jitter_x = random_uniform(rng), jitter_y = random_uniform(rng);
    OPS(4 + 1); IOPS(2 + 4); // u, v: add + div each; pmf / ..
    float u = ((float)x + jitter_x) / (float)e.width;
    float v = ((float)y + jitter_y) / (float)e.height;
    EnvironmentSample s;
    s.direction = equirectangular_uv_to_direction(u, v);
    s.radiance = sample_env_bilinear(e, u, v);
    float pmf = e.alias[index].pmf;
    s.pdf = pmf / environment_pixel_solid_angle(v, e);
    return s;
}

// ------------------------------------------------------------------ BSDF (shader.wgsl:55-84, 833-1210)
struct Frame { V3 tangent, bitangent, normal; };
inline Frame make_frame(V3 normal) // :55-67
{
    OPS(1);
    V3 helper = (fabs_(normal.z) < 0.999f) ? v3(0, 0, 1) : v3(1, 0, 0);
    V3 tangent = normalize(cross(helper, normal));
    V3 bitangent = cross(normal, tangent);
    return Frame{tangent, bitangent, normal};
}
inline V3 to_frame_local(const Frame &f, V3 w) { return v3(dot(w, f.tangent), dot(w, f.bitangent), dot(w, f.normal)); }
inline V3 to_frame_world(const Frame &f, V3 l) { return normalize(f.tangent * l.x + f.bitangent * l.y + f.normal * l.z); }

struct BsdfMaterial { V3 color; float metallic; float alpha; V3 f0; V3 emission; };
inline V3 lerp_vec3f(V3 a, V3 b, float t) { OPS(1); return (1.0f - t) * a + t * b; } // :242
inline float lerp_f32(float a, float b, float t) { OPS(4); return (1.0f - t) * a + t * b; }
inline float max_component(V3 v) { return fmax_(v.x, fmax_(v.y, v.z)); }
inline float luminance(V3 c) { OPS(5); return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } // :884
inline BsdfMaterial make_bsdf_material(const orc_material &m) // :850-873
{
    BsdfMaterial b;
    b.color = v3(m.color);
    b.metallic = m.metallic;
    OPS(1);
    b.alpha = fmax_(0.001f, m.roughness * m.roughness);
    b.f0 = lerp_vec3f(v3(0.04f, 0.04f, 0.04f), v3(m.color), saturate(m.metallic));
    b.emission = v3(m.emission);
    return b;
}
inline V3 surface_kd(const BsdfMaterial &m) // :878-881
{
    OPS(2);
    V3 kd0 = m.color * (1.0f - saturate(m.metallic));
    return kd0 * (1.0f - max_component(m.f0));
}
inline V3 sample_cosine_hemisphere(float sx, float sy) // :892-901
{
    OPS(1 + 1 + 2 + 4 + 1); // sqrt; phi; x, y; 1 - x^2 - y^2; sqrt
    float r = rsrt_sqrtf(sx);
    float phi = TWO_PI * sy;
    float x = r * rsrt_cosf(phi), y = r * rsrt_sinf(phi);
    float z = rsrt_sqrtf(fmax_(0.0f, 1.0f - x * x - y * y));
    return v3(x, y, z);
}
inline float pdf_cosine_hemisphere(V3 wi) { OPS(2); return wi.z <= 0.0f ? 0.0f : wi.z / PI; } // :914-919
inline float d_ggx(float ndh, float alpha) // :924-928
{
    OPS(1 + 4 + 3);
    float alpha_2 = alpha * alpha;
    float denominator = (ndh * ndh) * (alpha_2 - 1.0f) + 1.0f;
    return alpha_2 / (PI * denominator * denominator);
}
inline float lambda_ggx(float ndv, float alpha) // :1014-1020
{
    OPS(1 + 5 + 1 + 2); // ndv^2; 2 mul, sub, div, add; sqrt; - 1, / 2
    float ndv2 = ndv * ndv;
    return (rsrt_sqrtf(1.0f + alpha * alpha * (1.0f - ndv2) / ndv2) - 1.0f) / 2.0f;
}
inline float g1_ggx(float ndv, float alpha) { OPS(2); return 1.0f / (1.0f + lambda_ggx(ndv, alpha)); } // :1026
inline float g_smith_ggx(float ndo, float ndi, float alpha) { OPS(1); return g1_ggx(ndo, alpha) * g1_ggx(ndi, alpha); }
inline V3 f_schlick(V3 f0, float cos_theta) // :1045-1051
{
    OPS(1 + 1 + 2);
    float x = 1.0f - saturate(cos_theta);
    float x_2 = x * x;
    float x_5 = x_2 * x_2 * x;
    return f0 + (v3(1, 1, 1) - f0) * x_5;
}
inline float pdf_ggx_half_vector_visible(V3 h, V3 wo, float alpha) // :931-945
{
    float ndh = h.z, ndo = wo.z;
    OPS(1);
    if (ndh <= 0.0f) return 0.0f;
    OPS(3);
    return d_ggx(ndh, alpha) * g1_ggx(ndo, alpha) * fmax_(0.0f, dot(wo, h)) / ndo;
}
inline V3 sample_ggx_visible_half_vector(float sx, float sy, V3 wo, float alpha) // :962-1009
{
    V3 vs = normalize(wo * v3(alpha, alpha, 1.0f));
    float length_squared = dot2(vs.x, vs.y, vs.x, vs.y);
    V3 alt = v3(-vs.y, vs.x, 0.0f) * inverse_sqrt(length_squared);
    OPS(1 + 1 + 1 + 2 + 3 + 1 + 5 + 1 + 2); // > 0; sqrt; azimuth; dx, dy; 1 - dx^2, sqrt; 1 - dx^2 - dy^2, sqrt; alpha * hs.x, .y
    V3 tangent_x = (length_squared > 0.0f) ? alt : v3(1, 0, 0);
    V3 tangent_y = cross(vs, tangent_x);
    // sample_uniform_disk :907-911
    float radius = rsrt_sqrtf(sx);
    float azimuth = TWO_PI * sy;
    float dx = radius * rsrt_cosf(azimuth), dy = radius * rsrt_sinf(azimuth);
    dy = lerp_f32(rsrt_sqrtf(fmax_(0.0f, 1.0f - dx * dx)), dy, vs.z);
    V3 hs = dx * tangent_x + dy * tangent_y + rsrt_sqrtf(fmax_(0.0f, 1.0f - dx * dx - dy * dy)) * vs;
    return normalize(v3(alpha * hs.x, alpha * hs.y, fmax_(0.0f, hs.z)));
}
inline V3 bsdf_eval_local(V3 wo, V3 wi, const BsdfMaterial &m) // :1053-1087
{
    OPS(2);
    if (wo.z <= 0.0f || wi.z <= 0.0f) return v3(0, 0, 0);
    OPS(1 + 3); // D * G; / (4 ndo ndi)
    float ndo = wo.z, ndi = wi.z;
    V3 h = normalize(wo + wi);
    float ndh = saturate(h.z);
    float D = d_ggx(ndh, m.alpha);
    float G = g_smith_ggx(ndo, ndi, m.alpha);
    V3 F = f_schlick(m.f0, dot(h, wo));
    V3 fs = (D * G) / (4.0f * ndo * ndi) * F;
    V3 kd = surface_kd(m);
    V3 fd = kd * INV_PI; // kd * (1 / PI), const-evaluated
    return fd + fs;
}
inline float pdf_specular_wi_visible(V3 wo, V3 wi, float alpha) // :1089-1102
{
    OPS(2);
    if (wo.z <= 0.0f || wi.z <= 0.0f) return 0.0f;
    V3 h = normalize(wo + wi);
    float wo_dot_h = fabs_(dot(wo, h));
    OPS(1);
    if (wo_dot_h <= 0.0f) return 0.0f;
    OPS(2);
    return pdf_ggx_half_vector_visible(h, wo, alpha) / (4.0f * wo_dot_h);
}
inline float bsdf_pdf_local(V3 wo, V3 wi, const BsdfMaterial &m) // :1104-1114
{
    OPS(2);
    if (wo.z <= 0.0f || wi.z <= 0.0f) return 0.0f;
    OPS(1 + 3);
    float ps = saturate(luminance(m.f0));
    float pd = 1.0f - ps;
    return pd * pdf_cosine_hemisphere(wi) + ps * pdf_specular_wi_visible(wo, wi, m.alpha);
}
struct BsdfSample { V3 ray_direction, scattering; float pdf; };
inline BsdfSample bsdf_sample(const Ray &ray, V3 n, const BsdfMaterial &m, u32 *rng) // :1116-1202
{
    V3 wo_world = -ray.direction;
    OPS(1);
    if (dot(n, wo_world) <= 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    Frame frame = make_frame(n);
    V3 wo = to_frame_local(frame, wo_world);
    OPS(1);
    if (wo.z <= 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 1, 0), 0.0f};
    OPS(1 + 1); // pd; sample < pd
    float ps = saturate(luminance(m.f0));
    float pd = 1.0f - ps;
    V3 wi;
    float sample = random_uniform(rng);
    if (sample < pd) {
        OPS(1);
        float s0 = sample / fmax_(pd, 1.e-6f);
        float s1 = random_uniform(rng);
        wi = sample_cosine_hemisphere(s0, s1);
    } else {
        OPS(2 + 1 + 1); // s0; 2 * dot; wi.z <= 0
        float s0 = (sample - pd) / fmax_(ps, 1.e-6f);
        float s1 = random_uniform(rng);
        V3 h = sample_ggx_visible_half_vector(s0, s1, wo, m.alpha);
        V3 i = -wo; // reflect(-wo, h) = i - 2*dot(h,i)*h
        wi = i - (2.0f * dot(h, i)) * h;
        if (wi.z <= 0.0f) return BsdfSample{v3(1, 0, 0), v3(1, 0, 0), 0.0f};
    }
    V3 scattering = bsdf_eval_local(wo, wi, m);
    float pdf = bsdf_pdf_local(wo, wi, m);
    V3 wi_world = to_frame_world(frame, wi);
    OPS(1);
    if (dot(n, wi_world) < 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 1, 0), 0.0f};
    return BsdfSample{wi_world, scattering, pdf};
}
inline float power_heuristic(float a, float b) // :1206-1210
{
    OPS(4);
    float a2 = a * a, b2 = b * b;
    return a2 / (a2 + b2);
}

// ------------------------------------------------------------------ trace_ray (shader.wgsl:1213-1303)
V3 trace_ray(const Ctx &cx, Ray ray, u32 *rng, u32 max_bounces)
{
    Stats &st = *cx.st;
    V3 incoming_light = v3(0, 0, 0);
    V3 throughput = v3(1, 1, 1);
    float last_sample_pdf = 1.0f;
    for (u32 bounce = 0; bounce < max_bounces; bounce++) {
        st.ext_rays++;
        HitInfo info = cast_ray(cx, ray);
        if (!info.did_hit) {
            st.escapes++;
            V3 environment_light = sky_light(cx, ray.direction);
            float pdf = environment_direction_pdf(cx, ray.direction);
            float weight = power_heuristic(last_sample_pdf, pdf);
            incoming_light = incoming_light + throughput * environment_light * weight;
            break;
        }
        st.shaded_hits++;
        BsdfMaterial material = make_bsdf_material(cx.sc->materials[info.material_id]);
        incoming_light = incoming_light + throughput * material.emission;
        {
            st.nee_events++;
            EnvironmentSample environment = sample_environment(cx, rng);
            V3 wo_world = -ray.direction;
            V3 wi_world = environment.direction;
            float cos_theta = fmax_(0.0f, dot(info.normal, wi_world));
            bool lit = false;
            OPS(2);
            if (cos_theta > 0.0f && environment.pdf > 0.0f) {
                st.shadow_rays++;
                Ray shadow = {info.hit_point, environment.direction};
                lit = !cast_ray_bvh(cx, shadow, (cx.flags & ORC_FLAG_ANYHIT_SHADOW) != 0).did_hit;
            }
            if (lit) {
                Frame frame = make_frame(info.normal);
                V3 wo = to_frame_local(frame, wo_world);
                V3 wi = to_frame_local(frame, wi_world);
                V3 scattering = bsdf_eval_local(wo, wi, material);
                float pdf_bsdf = bsdf_pdf_local(wo, wi, material);
                float weight = power_heuristic(environment.pdf, pdf_bsdf);
                OPS(0); // (the vector expression below counts itself: 3 mul-vec, 1 scale, 1 div-vec, 1 add)
                incoming_light =
                    incoming_light + throughput * weight * environment.radiance * scattering * cos_theta / environment.pdf;
            }
        }
        {
            BsdfSample sample = bsdf_sample(ray, info.normal, material, rng);
            if (sample.ray_direction.x == 0.0f && sample.ray_direction.y == 0.0f && sample.ray_direction.z == 0.0f) {
                incoming_light = sample.scattering; // :1274 overwrites
                break;
            }
            OPS(3 + 1); // direction == 0 (x3); pdf <= 0
            if (sample.pdf <= 0.0f) break;
            OPS(1 + 1); // cos / pdf; length < 0.001
            float cos_theta = fmax_(0.0f, dot(info.normal, sample.ray_direction));
            throughput = throughput * (sample.scattering * (cos_theta / sample.pdf));
            if (length(throughput) < 0.001f) break;
            last_sample_pdf = sample.pdf;
            ray = Ray{info.hit_point, sample.ray_direction};
        }
    }
    return incoming_light;
}

// shader.wgsl:1305-1373 (dev_index == 1 path)
V3 pixel_sample(const Ctx &cx, const orc_camera &cam, u32 W, u32 H, u32 px, u32 py, u32 sample_index, u32 max_bounces)
{
    u32 pixel_index = py * W + px;
    u32 rng = 0;
    salt_rng(&rng, pixel_index);
    salt_rng(&rng, sample_index);
    float jx, jy;
    random_in_circle_uniform(&rng, &jx, &jy);
    OPS(2 + 4 + 4 + 1 + 1 + 3); IOPS(6); // fx, fy; sx; sy; fov / 2; aspect; ray_camera_space | cvt
    float fx = (float)px + jx, fy = (float)py + jy;
    float sx = ((fx / (float)W) * 2.0f - 1.0f) * 1.0f;
    float sy = ((fy / (float)H) * 2.0f - 1.0f) * -1.0f;
    float max_y_component = rsrt_sinf(cam.fov_y / 2.0f);
    float aspect_ratio = (float)W / (float)H;
    V3 ray_camera_space = v3(sx * max_y_component * aspect_ratio, sy * max_y_component, -1.0f);
    M3 rot = {v3(cam.rot[0]), v3(cam.rot[1]), v3(cam.rot[2])};
    Ray ray = {v3(cam.pos), normalize(mul(rot, ray_camera_space))};
    cx.st->paths++;
    return trace_ray(cx, ray, &rng, max_bounces);
}

// ------------------------------------------------------------------ host preprocessing
// Plain Rust f32 semantics: no fusing anywhere below.
struct B3 { V3 mn, mx; };
const float FMAX = 3.40282347e+38f;
inline V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
inline B3 b_identity() { return B3{v3(FMAX, FMAX, FMAX), v3(-FMAX, -FMAX, -FMAX)}; } // scene.rs:67-72
inline B3 b_union(const B3 &a, const B3 &b) { return B3{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; } // :131-141
inline V3 b_center(const B3 &b) { return b.mn * 0.5f + b.mx * 0.5f; } // :102-104
inline u32 b_max_axis(const B3 &b) // :113-122
{
    V3 d = b.mx - b.mn;
    if (d.z > d.x && d.z > d.y) return 2;
    if (d.y > d.x) return 1;
    return 0;
}
inline float b_surface_area(const B3 &b) // :125-128
{
    V3 d = b.mx - b.mn;
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}

struct PrimInfo { u32 type, index; B3 bounds; };
struct BuildNode {
    B3 bounds; bool leaf; u32 first, count; u32 axis;
    std::unique_ptr<BuildNode> c0, c1;
};

// src/bvh.rs:215-337
std::unique_ptr<BuildNode> build_sah(PrimInfo *prims, size_t n, std::vector<PrimInfo> &ordered)
{
    const size_t MAX_PRIMITIVES_PER_LEAF = 5, BUCKET_COUNT = 12;
    B3 bounds = b_identity();
    for (size_t i = 0; i < n; i++) bounds = b_union(bounds, prims[i].bounds);
    auto make_leaf = [&]() {
        auto node = std::make_unique<BuildNode>();
        node->bounds = bounds; node->leaf = true; node->first = (u32)ordered.size(); node->count = (u32)n; node->axis = 0;
        ordered.insert(ordered.end(), prims, prims + n);
        return node;
    };
    if (n <= MAX_PRIMITIVES_PER_LEAF) return make_leaf();
    B3 cb = b_identity(); // from_points of centres, :234
    for (size_t i = 0; i < n; i++) { V3 c = b_center(prims[i].bounds); cb.mn = vmin(cb.mn, c); cb.mx = vmax(cb.mx, c); }
    u32 axis = b_max_axis(cb);
    float min_c = comp(cb.mn, axis), max_c = comp(cb.mx, axis);
    if (min_c == max_c) return make_leaf(); // :240-244
    auto bucket_index_of = [&](const PrimInfo &p) -> size_t { // :258-269
        float center = comp(b_center(p.bounds), axis);
        float f = (float)BUCKET_COUNT * ((center - min_c) / (max_c - min_c));
        size_t b = (f > 0.0f) ? (size_t)f : 0; // `as usize` saturates; NaN -> 0
        if (b == BUCKET_COUNT) b = BUCKET_COUNT - 1;
        return b;
    };
    struct Bucket { size_t count; B3 bounds; };
    Bucket buckets[12];
    for (auto &b : buckets) { b.count = 0; b.bounds = b_identity(); }
    for (size_t i = 0; i < n; i++) {
        size_t b = bucket_index_of(prims[i]);
        buckets[b].count++;
        buckets[b].bounds = b_union(buckets[b].bounds, prims[i].bounds);
    }
    float costs[11];
    for (size_t index = 0; index < BUCKET_COUNT - 1; index++) { // :280-292
        B3 b0 = b_identity(), b1 = b_identity();
        size_t c0 = 0, c1 = 0;
        for (size_t j = 0; j <= index; j++) { b0 = b_union(b0, buckets[j].bounds); c0 += buckets[j].count; }
        for (size_t j = index + 1; j < BUCKET_COUNT; j++) { b1 = b_union(b1, buckets[j].bounds); c1 += buckets[j].count; }
        costs[index] = 0.125f + ((float)c0 * b_surface_area(b0) + (float)c1 * b_surface_area(b1)) / b_surface_area(bounds);
    }
    size_t min_cost_index = 0; // first minimum wins, :294-300
    for (size_t i = 1; i < BUCKET_COUNT - 1; i++) if (costs[i] < costs[min_cost_index]) min_cost_index = i;
    size_t split_index = 0; // two-pointer in-place partition, :304-315 (NOT stable)
    {
        size_t end_index = n;
        while (split_index < end_index) {
            if (bucket_index_of(prims[split_index]) <= min_cost_index) split_index++;
            else { end_index--; std::swap(prims[split_index], prims[end_index]); }
        }
    }
    if (split_index == 0 || split_index == n) { // :317-326, unreachable (buckets 0 and 11 never empty)
        size_t mid = n / 2;
        std::nth_element(prims, prims + mid, prims + n, [&](const PrimInfo &a, const PrimInfo &b) {
            return comp(b_center(a.bounds), axis) < comp(b_center(b.bounds), axis);
        });
        split_index = mid;
    }
    auto node = std::make_unique<BuildNode>();
    node->leaf = false; node->axis = axis; node->first = node->count = 0;
    node->c0 = build_sah(prims, split_index, ordered);
    node->c1 = build_sah(prims + split_index, n - split_index, ordered);
    node->bounds = b_union(b_union(b_identity(), node->c0->bounds), node->c1->bounds); // :199-205
    return node;
}
// src/bvh.rs:155-178
u32 flatten(const BuildNode &n, std::vector<orc_bvh_node> &out)
{
    orc_bvh_node u;
    memset(&u, 0, sizeof u);
    u.bmin[0] = n.bounds.mn.x; u.bmin[1] = n.bounds.mn.y; u.bmin[2] = n.bounds.mn.z;
    u.bmax[0] = n.bounds.mx.x; u.bmax[1] = n.bounds.mx.y; u.bmax[2] = n.bounds.mx.z;
    if (n.leaf) {
        u.idx = n.first; u.len = n.count; u.axis = 0;
        out.push_back(u);
        return (u32)out.size() - 1;
    }
    u.idx = 0; u.len = 0; u.axis = n.axis;
    out.push_back(u);
    u32 parent = (u32)out.size() - 1;
    flatten(*n.c0, out);
    u32 second = flatten(*n.c1, out);
    out[parent].idx = second;
    return parent;
}
u32 tree_depth(const BuildNode &n) { return n.leaf ? 0 : std::max(tree_depth(*n.c0), tree_depth(*n.c1)) + 1; }

} // namespace

// ====================================================================== C API
extern "C" {

int orc_build_bvh(const orc_sphere *spheres, uint32_t n_spheres, const orc_plane_src *planes, uint32_t n_planes,
                  const orc_vec3 *vertices, const orc_triangle *triangles, uint32_t n_triangles, orc_prim_info *prims_out,
                  orc_bvh_node *nodes_out, uint32_t *depth_out)
{
    std::vector<PrimInfo> prims; // src/bvh.rs:40-72: spheres, planes, triangles
    for (u32 i = 0; i < n_spheres; i++) { // scene.rs:173-180
        V3 p = v3(spheres[i].pos); float r = spheres[i].radius;
        prims.push_back(PrimInfo{0, i, B3{p - v3(r, r, r), p + v3(r, r, r)}});
    }
    for (u32 i = 0; i < n_planes; i++) { // scene.rs:203-207
        V3 p = v3(planes[i].pos), q = p + v3(planes[i].forward) + v3(planes[i].right);
        prims.push_back(PrimInfo{1, i, B3{vmin(vmin(v3(FMAX, FMAX, FMAX), p), q), vmax(vmax(v3(-FMAX, -FMAX, -FMAX), p), q)}});
    }
    for (u32 i = 0; i < n_triangles; i++) { // mesh.rs:143-147
        V3 a = v3(vertices[triangles[i].v0].v), b = v3(vertices[triangles[i].v1].v), c = v3(vertices[triangles[i].v2].v);
        prims.push_back(PrimInfo{2, i, B3{vmin(vmin(a, b), c), vmax(vmax(a, b), c)}});
    }
    if (prims.empty()) return -1; // bvh.rs:222 assert
    std::vector<PrimInfo> ordered;
    auto root = build_sah(prims.data(), prims.size(), ordered);
    std::vector<orc_bvh_node> nodes;
    flatten(*root, nodes);
    for (size_t i = 0; i < ordered.size(); i++) { prims_out[i].type = ordered[i].type; prims_out[i].index = ordered[i].index; }
    memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(orc_bvh_node));
    if (depth_out) *depth_out = tree_depth(*root);
    return (int)nodes.size();
}

// src/environments.rs:96-187
void orc_alias_table(uint32_t width, uint32_t height, const float *rgb, orc_alias_entry *out, uint32_t *leftover_out)
{
    const float PI_TRUE = 3.14159265358979323846f; // std::f32::consts::PI
    size_t length = (size_t)width * height;
    std::vector<float> prob(length);
    for (u32 y = 0; y < height; y++) {
        float angle_y = PI_TRUE * ((float)y + 0.5f) / (float)height;
        float s = rsrt_sinf(angle_y);
        for (u32 x = 0; x < width; x++) {
            const float *c = rgb + 3 * ((size_t)y * width + x);
            float lum = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];
            prob[(size_t)y * width + x] = lum * s;
        }
    }
    float weight_sum = 0.0f; // sequential f32 sum, :110
    for (size_t i = 0; i < length; i++) weight_sum += prob[i];
    for (size_t i = 0; i < length; i++) prob[i] = prob[i] * (float)length / weight_sum;
    std::vector<float> ap(prob);
    std::vector<size_t> small, large;
    for (size_t i = 0; i < length; i++) if (prob[i] < 1.0f) small.push_back(i);
    for (size_t i = 0; i < length; i++) if (prob[i] >= 1.0f) large.push_back(i);
    std::vector<uint8_t> assigned(length, 0);
    for (;;) { // :135-159: pop small first, then large; a popped small is lost if large is empty
        if (small.empty()) break;
        size_t s = small.back(); small.pop_back();
        if (large.empty()) break;
        size_t l = large.back(); large.pop_back();
        out[s].probability = ap[s];
        out[s].alias_index = (u32)l;
        out[s].pmf = prob[s] / (float)length;
        out[s]._pad = 0;
        assigned[s] = 1;
        ap[l] -= 1.0f - ap[s];
        if (ap[l] < 1.0f) small.push_back(l); else large.push_back(l);
    }
    u32 leftover = 0;
    for (size_t i = 0; i < length; i++) {
        if (!assigned[i]) { // :167-176
            leftover++;
            out[i].probability = 1.0f; out[i].alias_index = (u32)i; out[i].pmf = 1.0f / (float)length; out[i]._pad = 0;
        }
    }
    if (leftover_out) *leftover_out = leftover;
}

// src/scene.rs:190-201 with glam 0.30 Vec3::cross / normalize / Mat3::inverse (un-vendored; restated
// from glam's published formulas: cross = (y*rz - ry*z, ...), normalize = v * (1/length),
// inverse = transpose(cols(c1 x c2, c2 x c0, c0 x c1) * (1/det)), det = c2 . (c0 x c1))
void orc_plane_to_uniform(const orc_plane_src *in, orc_plane *out)
{
    auto gcross = [](V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); };
    auto gdot = [](V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); };
    V3 fwd = v3(in->forward), right = v3(in->right);
    V3 n = gcross(fwd, right);
    n = n * (1.0f / sqrtf(gdot(n, n)));
    V3 c0 = right, c1 = n, c2 = fwd;
    V3 t0 = gcross(c1, c2), t1 = gcross(c2, c0), t2 = gcross(c0, c1);
    float det = gdot(c2, t2);
    float inv_det = 1.0f / det;
    V3 r0 = t0 * inv_det, r1 = t1 * inv_det, r2 = t2 * inv_det; // rows of the inverse
    memset(out, 0, sizeof *out);
    memcpy(out->pos, in->pos, 12);
    out->normal[0] = n.x; out->normal[1] = n.y; out->normal[2] = n.z;
    // transpose: column j = (r0[j], r1[j], r2[j])
    out->m[0][0] = r0.x; out->m[0][1] = r1.x; out->m[0][2] = r2.x;
    out->m[1][0] = r0.y; out->m[1][1] = r1.y; out->m[1][2] = r2.y;
    out->m[2][0] = r0.z; out->m[2][1] = r1.z; out->m[2][2] = r2.z;
    out->material_id = in->material_id;
}

// src/camera.rs:26-28, 111-119 with glam Mat3::from_axis_angle (un-vendored; published formula)
void orc_camera_uniform(const float pos[3], float yaw, float pitch, float fov_y, orc_camera *out)
{
    auto axis_angle = [](V3 axis, float angle) {
        float s = rsrt_sinf(angle), c = rsrt_cosf(angle);
        float xs = axis.x * s, ys = axis.y * s, zs = axis.z * s;
        float x = axis.x, y = axis.y, z = axis.z;
        float x2 = x * x, y2 = y * y, z2 = z * z;
        float omc = 1.0f - c;
        float xyomc = x * y * omc, xzomc = x * z * omc, yzomc = y * z * omc;
        return M3{v3(x2 * omc + c, xyomc + zs, xzomc - ys), v3(xyomc - zs, y2 * omc + c, yzomc + xs),
                  v3(xzomc + ys, yzomc - xs, z2 * omc + c)};
    };
    auto gmul = [](const M3 &m, V3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }; // glam Mat3 * Vec3
    M3 ry = axis_angle(v3(0, 1, 0), yaw), rx = axis_angle(v3(1, 0, 0), pitch);
    M3 r = {gmul(ry, rx.c0), gmul(ry, rx.c1), gmul(ry, rx.c2)};
    memset(out, 0, sizeof *out);
    memcpy(out->pos, pos, 12);
    const V3 *cols[3] = {&r.c0, &r.c1, &r.c2};
    for (int j = 0; j < 3; j++) { out->rot[j][0] = cols[j]->x; out->rot[j][1] = cols[j]->y; out->rot[j][2] = cols[j]->z; }
    out->fov_y = fov_y;
}

int orc_render(const orc_scene *scene, const orc_env *env, const orc_camera *cam, uint32_t W, uint32_t H,
               uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces, uint32_t flags, int n_threads,
               float *sum_rgba, orc_stats *stats)
{
    if (!scene || !env || !cam || !sum_rgba || scene->n_nodes == 0) return -1;
    Stats total;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int tiles_x = (int)((W + 15) / 16), tiles_y = (int)((H + 15) / 16);
#pragma omp parallel
    {
        Stats local;
        Ctx cx = {scene, env, flags, &local};
#ifdef ORC_COUNT_OPS
        g_f32_ops = g_int_ops = 0;
#endif
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < tiles_x * tiles_y; tile++) {
            u32 ty = (u32)(tile / tiles_x), tx = (u32)(tile % tiles_x);
            for (u32 py = ty * 16; py < std::min(H, ty * 16 + 16); py++)
                for (u32 px = tx * 16; px < std::min(W, tx * 16 + 16); px++) {
                    float *p = sum_rgba + 4 * ((size_t)py * W + px);
                    for (u32 s = sample_begin; s < sample_begin + sample_count; s++) {
                        V3 light = pixel_sample(cx, *cam, W, H, px, py, s, max_bounces);
                        p[0] = p[0] + light.x; p[1] = p[1] + light.y; p[2] = p[2] + light.z; // :1367-1371
                        p[3] = 1.0f;
                    }
                }
        }
#ifdef ORC_COUNT_OPS
        local.f32_ops = g_f32_ops; local.int_ops = g_int_ops;
#endif
#pragma omp critical
        total.add(local);
    }
    if (stats) *stats = total;
    return 0;
}

uint32_t orc_rng_seed(uint32_t pixel_index, uint32_t sample_index)
{
    u32 s = 0; salt_rng(&s, pixel_index); salt_rng(&s, sample_index); return s;
}
uint32_t orc_rng_next_u32(uint32_t *state) { return random_u32_uniform(state); }
float orc_u32_to_uniform(uint32_t r) { return (float)r / 4294967295.0f; }

static void put_hit(const HitInfo &h, orc_hit *o)
{
    o->did_hit = h.did_hit; o->distance = h.distance;
    o->hit_point[0] = h.hit_point.x; o->hit_point[1] = h.hit_point.y; o->hit_point[2] = h.hit_point.z;
    o->normal[0] = h.normal.x; o->normal[1] = h.normal.y; o->normal[2] = h.normal.z;
    o->material_id = h.material_id;
}
void orc_cast_ray_sphere(const float o[3], const float d[3], const orc_sphere *s, orc_hit *out) { put_hit(cast_ray_sphere(Ray{v3(o), v3(d)}, *s), out); }
void orc_cast_ray_plane(const float o[3], const float d[3], const orc_plane *p, orc_hit *out) { put_hit(cast_ray_plane(Ray{v3(o), v3(d)}, *p), out); }
void orc_cast_ray_triangle(const float o[3], const float d[3], const orc_scene *scene, const orc_triangle *t, orc_hit *out)
{
    put_hit(cast_ray_triangle(Ray{v3(o), v3(d)}, *scene, *t), out);
}
void orc_cast_rays(const orc_scene *scene, uint32_t n, const float *origins, const float *dirs, uint32_t mode, uint32_t flags,
                   orc_hit *out)
{
    Stats st;
    Ctx cx = {scene, nullptr, flags, &st};
    for (u32 i = 0; i < n; i++) {
        Ray r = {v3(origins + 3 * i), v3(dirs + 3 * i)};
        put_hit(mode == 0 ? cast_ray(cx, r) : cast_ray_bvh(cx, r, false), out + i);
    }
}
void orc_bsdf_eval_local(const orc_material *m, const float wo[3], const float wi[3], float out[3])
{
    V3 r = bsdf_eval_local(v3(wo), v3(wi), make_bsdf_material(*m));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_bsdf_pdf_local(const orc_material *m, const float wo[3], const float wi[3])
{
    return bsdf_pdf_local(v3(wo), v3(wi), make_bsdf_material(*m));
}
float orc_bsdf_sample(const orc_material *m, const float ray_dir[3], const float normal[3], uint32_t *rng, float dir_out[3],
                      float scattering_out[3])
{
    BsdfSample s = bsdf_sample(Ray{v3(0, 0, 0), v3(ray_dir)}, v3(normal), make_bsdf_material(*m), rng);
    dir_out[0] = s.ray_direction.x; dir_out[1] = s.ray_direction.y; dir_out[2] = s.ray_direction.z;
    scattering_out[0] = s.scattering.x; scattering_out[1] = s.scattering.y; scattering_out[2] = s.scattering.z;
    return s.pdf;
}
void orc_sky_light(const orc_env *env, const float dir[3], float out[3])
{
    Stats st; Ctx cx = {nullptr, env, 0, &st};
    V3 r = sky_light(cx, v3(dir));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_environment_direction_pdf(const orc_env *env, const float dir[3])
{
    Stats st; Ctx cx = {nullptr, env, 0, &st};
    return environment_direction_pdf(cx, v3(dir));
}
float orc_sample_environment(const orc_env *env, uint32_t *rng, float dir_out[3], float radiance_out[3])
{
    Stats st; Ctx cx = {nullptr, env, 0, &st};
    EnvironmentSample s = sample_environment(cx, rng);
    dir_out[0] = s.direction.x; dir_out[1] = s.direction.y; dir_out[2] = s.direction.z;
    radiance_out[0] = s.radiance.x; radiance_out[1] = s.radiance.y; radiance_out[2] = s.radiance.z;
    return s.pdf;
}
void orc_direction_to_uv(const float dir[3], float uv[2]) { direction_to_equirectangular_uv(v3(dir), &uv[0], &uv[1]); }
float orc_detmath(int fn, float a, float b)
{
    switch (fn) {
    case 0: return rsrt_sinf(a);
    case 1: return rsrt_cosf(a);
    case 2: return rsrt_atan2f(a, b);
    case 3: return rsrt_asinf(a);
    default: return 0.0f;
    }
}

} // extern "C"
