// tools/mfma_det.hip — a measurement behind DESIGN.md §4 "MFMA: not used" (north_star: "MFMA only if a batched ray-direction x triangle-edge
// contraction actually wins in rocprof").  Standalone: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_det tools/mfma_det.hip
//
// The one batched contraction the integrator offers: cast_ray_triangle's determinant (shader.wgsl:409-423), det = dot(e0, cross(d, e1)), for
// all rays x all triangles — as a matrix product it is det[r][t] = d_r . n_t with n_t = cross(e1_t, e0_t) precomputed (the scalar triple product
// rearranged: K = 3, padded to 4).  Three kernels over the same R rays and T = 64 triangles:
//   A  shader_bvh   the shader's expression on the VALU, for the 5 triangles a ray's BVH walk tests (house: 4.8 a ray, oracle count)
//   B  shader_all   the shader's expression on the VALU for all 64 triangles — the bits the reference produces, for the comparison
//   C  mfma_all     v_mfma_f32_32x32x2f32: 32 rays x 32 triangles x K = 4 per two instructions, all 64 triangles
// Printed: how many of C's R x T determinants differ from B's in some bit (the parity bar is bit equality: a different det is a different u, v, t
// and, at the |det| < 1e-8 and u, v edge tests, a different hit), and the time of each kernel (rocprofv3 --kernel-trace --stats agrees).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define T_TRIS 64
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Tri { float e0[3], e1[3], n[4]; };

__device__ __forceinline__ float det_shader(const float *d, const float *e0, const float *e1)
{
    // p1 = cross(d, e1) with one fma per component, det = dot(e0, p1) as two fmas on a product: rt_math.h cross / dot
    const float px = __builtin_fmaf(d[1], e1[2], -(e1[1] * d[2]));
    const float py = __builtin_fmaf(d[2], e1[0], -(e1[2] * d[0]));
    const float pz = __builtin_fmaf(d[0], e1[1], -(e1[0] * d[1]));
    return __builtin_fmaf(e0[2], pz, __builtin_fmaf(e0[1], py, e0[0] * px));
}

__global__ __launch_bounds__(256) void shader_bvh(const float *dirs, const Tri *tris, uint32_t n_rays, float *out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
    float acc = 0.0f;
    for (uint32_t k = 0; k < 5; k++) { // the triangles "its leaves hold": five pseudo-random ones (a dependent gather, as in the walk)
        const Tri &t = tris[(r * 2654435761u + k * 40503u) >> 26];
        acc += det_shader(d, t.e0, t.e1);
    }
    out[r] = acc;
}

__global__ __launch_bounds__(256) void shader_all(const float *dirs, const Tri *tris, uint32_t n_rays, float *dets)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
    for (uint32_t t = 0; t < T_TRIS; t++) dets[(size_t)r * T_TRIS + t] = det_shader(d, tris[t].e0, tris[t].e1);
}

typedef float float16v __attribute__((ext_vector_type(16)));
// One wave: 32 rays x 64 triangles.  A (32 x 2 per instruction): lane l holds A[l % 32][l / 32]; B (2 x 32): lane l holds B[l / 32][l % 32];
// D (32 x 32): register i of lane l is D[8 * (i / 4) + 4 * (l / 32) + i % 4][l % 32].
__global__ __launch_bounds__(256) void mfma_all(const float *dirs, const Tri *tris, uint32_t n_rays, float *dets)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t r0 = wave * 32u;
    if (r0 >= n_rays) return;
    const uint32_t m = lane & 31u, kh = lane >> 5;
    const uint32_t r = r0 + m < n_rays ? r0 + m : n_rays - 1u;
    // k = 0, 1 in the first instruction (kh picks which), k = 2, 3 (3: padding, zero) in the second
    const float a01 = dirs[3 * r + kh], a23 = kh == 0u ? dirs[3 * r + 2] : 0.0f;
    for (uint32_t tb = 0; tb < T_TRIS; tb += 32u) {
        const Tri &t = tris[tb + m];
        const float b01 = t.n[kh], b23 = kh == 0u ? t.n[2] : 0.0f;
        float16v acc = {0};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, b23, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t row = 8u * (uint32_t)(i / 4) + 4u * kh + (uint32_t)(i % 4);
            if (r0 + row < n_rays) dets[(size_t)(r0 + row) * T_TRIS + tb + m] = acc[i];
        }
    }
}

int main(int argc, char **argv)
{
    const uint32_t n_rays = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 20);
    std::vector<float> dirs(3 * (size_t)n_rays);
    std::vector<Tri> tris(T_TRIS);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f * 2.0f - 1.0f; };
    for (uint32_t r = 0; r < n_rays; r++) {
        float d[3] = {rnd(), rnd(), rnd()};
        const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-9f;
        for (int k = 0; k < 3; k++) dirs[3 * (size_t)r + k] = d[k] / len;
    }
    for (auto &t : tris) {
        for (int k = 0; k < 3; k++) { t.e0[k] = rnd() * 2.0f; t.e1[k] = rnd() * 2.0f; }
        // n = cross(e1, e0), so that d . n = dot(e0, cross(d, e1))
        t.n[0] = t.e1[1] * t.e0[2] - t.e1[2] * t.e0[1];
        t.n[1] = t.e1[2] * t.e0[0] - t.e1[0] * t.e0[2];
        t.n[2] = t.e1[0] * t.e0[1] - t.e1[1] * t.e0[0];
        t.n[3] = 0.0f;
    }
    float *d_dirs, *d_a, *d_b, *d_c;
    Tri *d_tris;
    CHECK(hipMalloc(&d_dirs, dirs.size() * 4));
    CHECK(hipMalloc(&d_tris, tris.size() * sizeof(Tri)));
    CHECK(hipMalloc(&d_a, (size_t)n_rays * 4));
    CHECK(hipMalloc(&d_b, (size_t)n_rays * T_TRIS * 4));
    CHECK(hipMalloc(&d_c, (size_t)n_rays * T_TRIS * 4));
    CHECK(hipMemcpy(d_dirs, dirs.data(), dirs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tris, tris.data(), tris.size() * sizeof(Tri), hipMemcpyHostToDevice));
    hipEvent_t e[4];
    for (auto &x : e) CHECK(hipEventCreate(&x));
    const dim3 blk(256), grd((n_rays + 255) / 256), grd_m((n_rays / 32 * 64 + 255) / 256 + 1);
    float ms[3] = {0, 0, 0};
    for (int rep = 0; rep < 4; rep++) { // (the first repetition warms up)
        CHECK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(shader_bvh, grd, blk, 0, 0, d_dirs, d_tris, n_rays, d_a);
        CHECK(hipEventRecord(e[1]));
        hipLaunchKernelGGL(shader_all, grd, blk, 0, 0, d_dirs, d_tris, n_rays, d_b);
        CHECK(hipEventRecord(e[2]));
        hipLaunchKernelGGL(mfma_all, grd_m, blk, 0, 0, d_dirs, d_tris, n_rays, d_c);
        CHECK(hipEventRecord(e[3]));
        CHECK(hipDeviceSynchronize());
        if (rep) for (int k = 0; k < 3; k++) { float t; CHECK(hipEventElapsedTime(&t, e[k], e[k + 1])); ms[k] += t / 3.0f; }
    }
    std::vector<float> b((size_t)n_rays * T_TRIS), c(b.size());
    CHECK(hipMemcpy(b.data(), d_b, b.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c.data(), d_c, c.size() * 4, hipMemcpyDeviceToHost));
    size_t differ = 0, sign = 0, big = 0;
    double worst = 0.0;
    for (size_t i = 0; i < b.size(); i++) {
        uint32_t ub, uc;
        memcpy(&ub, &b[i], 4); memcpy(&uc, &c[i], 4);
        if (ub != uc) differ++;
        if ((b[i] < 0) != (c[i] < 0)) sign++;
        const double rel = std::fabs((double)b[i] - (double)c[i]) / (std::fabs((double)b[i]) + 1e-30);
        if (rel > 1e-3 && std::fabs(b[i]) > 1e-3) big++;
        if (std::fabs(b[i]) > 1e-3 && rel > worst) worst = rel;
    }
    if (big > b.size() / 1000) { // the layout assumed for A / B / D is wrong: the numbers would mean nothing
        fprintf(stderr, "mfma_det: %zu of %zu MFMA determinants are off by more than 1e-3: the register layout assumed for v_mfma_f32_32x32x2f32 is wrong\n", big, b.size());
        return 1;
    }
    printf("rays %u, triangles %d\n", n_rays, T_TRIS);
    printf("A shader_bvh  (VALU, the shader's expression, 5 triangles a ray)   %8.3f ms  %7.2f ns per 1000 rays\n", ms[0], ms[0] * 1e6 / n_rays * 1000 / 1000);
    printf("B shader_all  (VALU, the shader's expression, all 64 triangles)    %8.3f ms\n", ms[1]);
    printf("C mfma_all    (v_mfma_f32_32x32x2f32, d . cross(e1, e0), all 64)   %8.3f ms   = %.2f x kernel A, %.2f x kernel B\n", ms[2], ms[2] / ms[0], ms[2] / ms[1]);
    printf("determinants of C that differ from B in some bit: %zu of %zu (%.1f %%); with another SIGN: %zu; worst relative difference where |det| > 1e-3: %.2e\n",
           differ, b.size(), 100.0 * differ / b.size(), sign, worst);
    return 0;
}
