// gather_rate — what the vector memory path of one CU sustains for the access shapes of the BVH walks (round 3, second session).
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o gpurun_out/gather_rate && gpurun_out/gather_rate
// Every kernel runs 256 workgroups of 1,024 threads (one per CU, 16 waves, the walk kernel's shape) over a table of
// 128-byte "nodes" that fits L2 (2,064 nodes = 258 KB, the 15 k-triangle grid's wide tree), each lane picking a new pseudo-random
// node per trip, and reports lane-loads and wave-instructions per CU-cycle (s_memtime of wave 0 / wall time x clock).
//   node8      a lane reads its node's 8 x 16 B with eight global_load_dwordx4   (trace_wide's fetch of a global node)
//   node1      a lane reads 16 B of its node                                      (a triangle record piece, an environment texel)
//   node2x32   ... 2 x 16 B, 32 B apart                                           (the fixed-order walk's element)
//   coop8      eight neighbouring lanes read the eight 16-B pieces of ONE lane's node (8 lines per instruction), eight instructions cover the
//              wave's 64 nodes; the pieces go to their owner through LDS          (what a cooperative node fetch would cost)
//   lds8       the same 8 x 16 B per lane from an LDS copy of the first 256 nodes (ds_read_b128)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t next_node(uint32_t &s, uint32_t n) { s = s * 747796405u + 2891336453u; return ((s >> 10) * (uint64_t)n) >> 22; }

template <int PIECES, int STRIDE16>
__global__ __launch_bounds__(1024) void k_node(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t trips, float *sink, unsigned long long *cycles)
{
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 1u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < trips; k++) {
        const uint32_t i = next_node(s, n_nodes);
        float4 v[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; p++) v[p] = nodes[(size_t)i * 8u + p * STRIDE16];
#pragma unroll
        for (int p = 0; p < PIECES; p++) acc += v[p].x + v[p].w;
        s += __float_as_uint(acc) & 1u; // the next address depends on the data: a dependent chain, as in a walk
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) *sink = acc;
}

__global__ __launch_bounds__(1024) void k_coop8(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t trips, float *sink, unsigned long long *cycles)
{
    __shared__ float4 stage[16][64 * 8 + 8]; // one node image per lane and wave: 8 KB + pad per wave
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 1u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < trips; k++) {
        const uint32_t i = next_node(s, n_nodes);
        // instruction j: lanes 8g .. 8g+7 fetch the eight pieces of the node wanted by lane 8j + g
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t owner = 8u * j + (lane >> 3);
            const uint32_t want = __builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)i);
            stage[wave][owner * 8u + (lane & 7u)] = nodes[(size_t)want * 8u + (lane & 7u)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < 8; p++) { const float4 v = stage[wave][lane * 8u + p]; acc += v.x + v.w; }
        __builtin_amdgcn_wave_barrier();
        s += __float_as_uint(acc) & 1u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) *sink = acc;
}

__global__ __launch_bounds__(1024) void k_lds8(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t trips, float *sink, unsigned long long *cycles)
{
    __shared__ float4 image[256 * 8];
    for (uint32_t i = threadIdx.x; i < 256u * 8u; i += 1024u) image[i] = nodes[i];
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 1u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < trips; k++) {
        const uint32_t i = next_node(s, 256u);
        float4 v[8];
#pragma unroll
        for (int p = 0; p < 8; p++) v[p] = image[i * 8u + p];
#pragma unroll
        for (int p = 0; p < 8; p++) acc += v[p].x + v[p].w;
        s += __float_as_uint(acc) & 1u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) *sink = acc;
}

int main(int argc, char **argv)
{
    const uint32_t n_nodes = argc > 1 ? (uint32_t)atoi(argv[1]) : 2064u, trips = 2000u;
    float4 *nodes = nullptr;
    float *sink = nullptr;
    unsigned long long *cycles = nullptr;
    CK(hipMalloc(&nodes, (size_t)n_nodes * 128u));
    CK(hipMemset(nodes, 0, (size_t)n_nodes * 128u));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&cycles, 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(256);
    auto report = [&](const char *name, double lane_loads_per_trip, double instr_per_trip, float ms) {
        CK(hipMemcpy(h.data(), cycles, 256 * 8, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (auto c : h) cyc += (double)c;
        cyc /= 256.0; // s_memtime ticks at 100 MHz on this part? print both: the wall time decides
        const double per_cu_trips = 16.0 * trips; // wave-trips per CU
        std::printf("%-9s %8.3f ms  s_memtime %10.0f ticks  | per CU: %6.2f ns per wave-trip, %6.3f lane-loads / ns, %6.3f wave-instr / ns\n", name, ms, cyc,
                    ms * 1e6 / per_cu_trips, per_cu_trips * lane_loads_per_trip / (ms * 1e6), per_cu_trips * instr_per_trip / (ms * 1e6));
        return 0;
    };
    for (int rep = 0; rep < 2; rep++) {
#define RUN(name, kern, ll, ins)                                                                       \
    do {                                                                                               \
        CK(hipEventRecord(e0));                                                                        \
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, nodes, n_nodes, trips, sink, cycles);    \
        CK(hipEventRecord(e1));                                                                        \
        CK(hipEventSynchronize(e1));                                                                   \
        float ms;                                                                                      \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                          \
        if (rep) report(name, ll, ins, ms);                                                            \
    } while (0)
        RUN("node8", (k_node<8, 1>), 512.0, 8.0);
        RUN("node4", (k_node<4, 1>), 256.0, 4.0);
        RUN("node2x32", (k_node<2, 2>), 128.0, 2.0);
        RUN("node1", (k_node<1, 1>), 64.0, 1.0);
        RUN("coop8", k_coop8, 512.0, 8.0);
        RUN("lds8", k_lds8, 512.0, 8.0);
    }
    std::printf("(%u nodes of 128 B = %.0f KB; 256 workgroups x 1,024 threads, %u trips; a CU-cycle at 2.4 GHz is 0.417 ns)\n", n_nodes, n_nodes * 128.0 / 1024.0, trips);
    return 0;
}
