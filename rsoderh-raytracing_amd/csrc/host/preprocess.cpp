// preprocess.cpp — alias table, plane / camera uniforms, synthetic environment.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../include/rsrt_detmath.h"
#include "../../../include/rsrt_host.h"
#include "host_math.h"

using namespace rsrt_host;

// AliasTable::build_by_luminance (reference src/environments.rs:96-187).
// weight = luminance * sin(pi*(y+0.5)/H) with the true f32 pi (:1, :101); sequential f32 sum
// (:110); p = w*N/sum (:115); Vose pairing with `small` / `large` used as LIFO stacks, small
// popped first (:135-159); every never-assigned index keeps {1, self, 1/N} (:163-178) — which
// gives large never-demoted pixels pmf 1/N instead of p/N; kept, it changes pixel values.
extern "C" int rsrt_alias_table_build(uint32_t width, uint32_t height, const float *rgb, rsrt_alias_entry *out,
                                      uint32_t *leftover_out)
{
    if (!rgb || !out || width == 0 || height == 0) return 1;
    const size_t n = (size_t)width * height;
    const float pi = 3.14159265358979323846f;
    std::vector<float> p(n);
    size_t k = 0;
    for (uint32_t y = 0; y < height; y++) {
        const float row_sin = rsrt_sinf(pi * ((float)y + 0.5f) / (float)height);
        for (uint32_t x = 0; x < width; x++, k++) {
            const float *c = rgb + 3 * k;
            p[k] = (0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]) * row_sin;
        }
    }
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) sum += p[i];
    const float nf = (float)n;
    for (size_t i = 0; i < n; i++) p[i] = p[i] * nf / sum;

    std::vector<float> residual(p);
    std::vector<uint32_t> small, large;
    const float default_pmf = 1.0f / nf;
    for (size_t i = 0; i < n; i++) {
        (p[i] < 1.0f ? small : large).push_back((uint32_t)i);
        out[i] = rsrt_alias_entry{1.0f, (uint32_t)i, default_pmf, 0u};
    }
    size_t assigned = 0;
    while (!small.empty()) {
        const uint32_t s = small.back();
        small.pop_back();
        if (large.empty()) break; // the popped small index stays a leftover, as in the reference
        const uint32_t l = large.back();
        large.pop_back();
        out[s] = rsrt_alias_entry{residual[s], l, p[s] / nf, 0u};
        assigned++;
        residual[l] -= 1.0f - residual[s];
        (residual[l] < 1.0f ? small : large).push_back(l);
    }
    if (leftover_out) *leftover_out = (uint32_t)(n - assigned);
    return 0;
}

// Plane::to_uniform (reference src/scene.rs:190-201): n = normalize(forward x right);
// M = inverse(Mat3::from_cols(right, n, forward)).  glam 0.30 (un-vendored) formulas:
// normalize = v * (1/length); inverse = transpose(cols(c1 x c2, c2 x c0, c0 x c1) * (1/det)),
// det = c2 . (c0 x c1).
extern "C" void rsrt_plane_to_uniform(const rsrt_plane_desc *in, rsrt_plane *out)
{
    const Vec3 right = vec3(in->right), fwd = vec3(in->forward);
    Vec3 n = cross(fwd, right);
    n = n * (1.0f / std::sqrt(dot(n, n)));
    const Vec3 col[3] = {right, n, fwd};
    const Vec3 adj[3] = {cross(col[1], col[2]), cross(col[2], col[0]), cross(col[0], col[1])};
    const float inv_det = 1.0f / dot(col[2], adj[2]);
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->pos, in->pos, sizeof in->pos);
    store(out->normal, n);
    for (int r = 0; r < 3; r++) { // row r of the inverse = adj[r] * inv_det; stored column-major
        const Vec3 row = adj[r] * inv_det;
        out->base_change_matrix[0][r] = row.x;
        out->base_change_matrix[1][r] = row.y;
        out->base_change_matrix[2][r] = row.z;
    }
    out->material_id = in->material_id;
}

namespace {
struct Mat3 { Vec3 c[3]; };
// glam Mat3::from_axis_angle (un-vendored; published formula)
Mat3 from_axis_angle(Vec3 a, float angle)
{
    const float s = rsrt_sinf(angle), c = rsrt_cosf(angle);
    const float omc = 1.0f - c;
    const float xy = a.x * a.y * omc, xz = a.x * a.z * omc, yz = a.y * a.z * omc;
    return Mat3{{vec3(a.x * a.x * omc + c, xy + a.z * s, xz - a.y * s), vec3(xy - a.z * s, a.y * a.y * omc + c, yz + a.x * s),
                 vec3(xz + a.y * s, yz - a.x * s, a.z * a.z * omc + c)}};
}
Vec3 mul(const Mat3 &m, Vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
} // namespace

// Camera::rot_transform = Ry(yaw) * Rx(pitch); CameraUniform::new (reference src/camera.rs:26-28, :111-119)
extern "C" void rsrt_camera_uniform(const rsrt_camera_desc *in, rsrt_camera *out)
{
    const Mat3 ry = from_axis_angle(vec3(0, 1, 0), in->yaw), rx = from_axis_angle(vec3(1, 0, 0), in->pitch);
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->pos, in->pos, sizeof in->pos);
    for (int j = 0; j < 3; j++) store(out->rot_transform[j], mul(ry, rx.c[j]));
    out->fov_y = in->fov_y;
}

namespace {
// exp(x) for x <= 0 from f32 add/mul only (Cody-Waite + degree-5 polynomial), so the synthetic
// environment is the same bytes on every IEEE host.
float det_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    float n = std::floor(x * 1.44269504088896341f + 0.5f);
    x = x - n * 0.693359375f;
    x = x - n * -2.12194440e-4f;
    float z = x * x;
    float q = 1.9875691500e-4f;
    q = q * x + 1.3981999507e-3f;
    q = q * x + 8.3334519073e-3f;
    q = q * x + 4.1665795894e-2f;
    q = q * x + 1.6666665459e-1f;
    q = q * x + 5.0000001201e-1f;
    q = q * z + x + 1.0f;
    return std::ldexp(q, (int)n);
}
} // namespace

// Stand-in for assets/hdri/*.hdr (missing blobs).  Frozen formula, all f32:
//   dir = (sin(th)cos(ph), cos(th), sin(th)sin(ph)), ph = (2u-1)pi, th = pi*v at texel centres
//   sky = mix((1,.9,.8), (.25,.45,.9), max(dir.y,0)) * 1.5 above the horizon, (.15,.13,.12) below
//   sun = 5e4 * exp(-(1 - dir.s)/2e-4) * (1,.95,.85), s = normalize(.4,.6,-.7)
extern "C" int rsrt_synth_environment(uint32_t width, uint32_t height, float *rgba_out)
{
    if (!rgba_out || width == 0 || height == 0) return 1;
    const float pi = 3.14159265358979323846f;
    Vec3 s = vec3(0.4f, 0.6f, -0.7f);
    s = s * (1.0f / std::sqrt(dot(s, s)));
    for (uint32_t y = 0; y < height; y++) {
        const float th = pi * (((float)y + 0.5f) / (float)height);
        const float st = rsrt_sinf(th), ct = rsrt_cosf(th);
        for (uint32_t x = 0; x < width; x++) {
            const float ph = (2.0f * (((float)x + 0.5f) / (float)width) - 1.0f) * pi;
            const Vec3 d = vec3(st * rsrt_cosf(ph), ct, st * rsrt_sinf(ph));
            float r, g, b;
            if (d.y > 0.0f) {
                const float t = d.y;
                r = ((1.0f - t) * 1.0f + t * 0.25f) * 1.5f;
                g = ((1.0f - t) * 0.9f + t * 0.45f) * 1.5f;
                b = ((1.0f - t) * 0.8f + t * 0.9f) * 1.5f;
            } else {
                r = 0.15f; g = 0.13f; b = 0.12f;
            }
            const float sun = 5.0e4f * det_expf(-((1.0f - dot(d, s)) / 2.0e-4f));
            float *o = rgba_out + 4 * ((size_t)y * width + x);
            o[0] = r + sun * 1.0f;
            o[1] = g + sun * 0.95f;
            o[2] = b + sun * 0.85f;
            o[3] = 0.0f;
        }
    }
    return 0;
}
