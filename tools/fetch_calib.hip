// fetch_calib — what FETCH_SIZE / WRITE_SIZE count on gfx950 for the access shapes of rt_render_pool_kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ... -- /tmp/fetch_calib
// Every kernel moves a KNOWN number of bytes from / to a buffer larger than the 256 MiB Infinity Cache, so that the
// counter can be read against it:
//   k_stream16   coalesced 16 B per lane                      (the guide's case: FETCH_SIZE reports 1/2)
//   k_stream12   coalesced 12 B per lane, three dword loads    (rt_resolve_kernel's reads of the sample buffer)
//   k_gather16   one random 16-byte record per lane            (environment texels / alias entries)
//   k_column4    4 B per lane, 64 consecutive dwords per wave  (a cold path-state column)
//   k_store4x3   three 4-byte stores per lane, 12-byte stride  (a finished path's sample)
//   k_store16    coalesced 16 B per lane stores
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_stream16(const float4 *src, size_t n, float *sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i].x;
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_stream12(const float *src, size_t n3, float *sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x)
        acc += src[3 * i] + src[3 * i + 1] + src[3 * i + 2];
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_gather16(const float4 *src, size_t n_records, size_t gathers_per_lane, float *sink)
{
    float acc = 0.0f;
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
    for (size_t k = 0; k < gathers_per_lane; k++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        acc += src[(s >> 20) % n_records].x;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_column4(const float *src, size_t n, float *sink)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_store4x3(float *dst, size_t n3)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x) {
        dst[3 * i] = 1.0f; dst[3 * i + 1] = 2.0f; dst[3 * i + 2] = 3.0f;
    }
}
__global__ void k_store16(float4 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1, 2, 3, 4);
}

int main()
{
    const size_t bytes = 1ull << 30; // 1 GiB, 4x the Infinity Cache
    void *buf = nullptr, *flush = nullptr;
    float *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&flush, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    const dim3 grid(256 * 8), block(256);
    const size_t gathers_per_lane = 64; // 2048 x 256 x 64 = 33.5 M gathers of 16 B
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(flush, rep, bytes)); // push buf out of the caches
        hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4 *)buf, bytes / 16, sink);
        CK(hipMemset(flush, rep + 2, bytes));
        hipLaunchKernelGGL(k_stream12, grid, block, 0, 0, (const float *)buf, bytes / 12, sink);
        CK(hipMemset(flush, rep + 4, bytes));
        hipLaunchKernelGGL(k_gather16, grid, block, 0, 0, (const float4 *)buf, bytes / 16, gathers_per_lane, sink);
        CK(hipMemset(flush, rep + 6, bytes));
        hipLaunchKernelGGL(k_column4, grid, block, 0, 0, (const float *)buf, bytes / 4, sink);
        CK(hipMemset(flush, rep + 8, bytes));
        hipLaunchKernelGGL(k_store4x3, grid, block, 0, 0, (float *)buf, bytes / 12);
        CK(hipMemset(flush, rep + 10, bytes));
        hipLaunchKernelGGL(k_store16, grid, block, 0, 0, (float4 *)buf, bytes / 16);
        CK(hipDeviceSynchronize());
    }
    std::printf("known bytes: stream16 %zu stream12 %zu gather16 %zu records (x16 B = %zu, x64 B lines = %zu, x128 B = %zu) column4 %zu store4x3 %zu store16 %zu\n",
                bytes, bytes / 12 * 12, (size_t)grid.x * block.x * gathers_per_lane, (size_t)grid.x * block.x * gathers_per_lane * 16,
                (size_t)grid.x * block.x * gathers_per_lane * 64, (size_t)grid.x * block.x * gathers_per_lane * 128, bytes, bytes / 12 * 12, bytes);
    return 0;
}
