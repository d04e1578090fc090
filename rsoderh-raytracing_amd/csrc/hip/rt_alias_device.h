// rt_alias_device.h — AliasTable::build_by_luminance (reference src/environments.rs:96-187) on the device, bit for bit
// what the host builder rsrt_alias_table_build (csrc/host/preprocess.cpp) produces.
//
// Two of its five steps are sequential BY DEFINITION and stay sequential here, because the result depends on their order:
//   * `sum` is a left-to-right f32 sum over all W*H weights (environments.rs:110): f32 addition is not associative, a
//     tree reduction gives a different sum (tests/test_alias_device.py shows it on the 64x32 environment), every
//     p = w*N/sum changes with it, and with p the small / large split and the whole table;
//   * the Vose pairing pops `small` and `large` as LIFO stacks and subtracts from the current large's residual one
//     small at a time (environments.rs:135-159): which pixel pairs with which depends on every earlier rounding.
// So: weights, normalisation, classification and the two index lists are data-parallel kernels; the sum and the pairing
// are run by ONE wave whose 64 lanes stage the operands through LDS (coalesced loads / parallel gathers) while the
// arithmetic itself advances one element at a time, exactly as the reference's loop.  Timings: the host
// builder is faster at every size the reference ships; this path is for environments that already live on the device.
#pragma once
#include "rt_math.h"

#define RT_ALIAS_CHUNK 2048u // entries staged per refill, per stack
// one-wave kernels: LDS written by all lanes, then read by all lanes of the SAME wave
#define RT_ALIAS_WAVE_SYNC()                                     \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)
RT_DEV uint32_t rt_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); } // the value IS wave-uniform: tell the compiler
RT_DEV float rt_uniform(float v) { return as_f(rt_uniform(as_u(v))); }

// step 1: weight = luminance * sin(pi * (y + 0.5) / H)  (environments.rs:97-107), rgba texels (alpha ignored)
__global__ void rt_alias_weights_kernel(const float4 *rgba, uint32_t width, uint32_t height, float *w)
{
    const size_t n = (size_t)width * height;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t y = (uint32_t)(i / width);
    const float pi = 3.14159265358979323846f;
    const float row_sin = rsrt_sinf(pi * ((float)y + 0.5f) / (float)height);
    const float4 c = rgba[i];
    w[i] = (0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z) * row_sin;
}

// step 2: the sequential f32 sum.  One wave; lanes stage 64 x 16 weights at a time into LDS, then every lane adds the
// same 1024 values in index order (LDS broadcast reads), so the running sum is wave-uniform.
__global__ __launch_bounds__(64) void rt_alias_sum_kernel(const float *w, size_t n, float *sum_out)
{
    __shared__ float buf[2][1024];
    const uint32_t lane = threadIdx.x;
    float sum = 0.0f;
    const size_t n_chunks = (n + 1023) / 1024;
    for (uint32_t k = 0; k < 16; k++) { // chunk 0
        const size_t i = (size_t)k * 64 + lane;
        buf[0][k * 64 + lane] = i < n ? w[i] : 0.0f;
    }
    for (size_t c = 0; c < n_chunks; c++) {
        RT_ALIAS_WAVE_SYNC();
        if (c + 1 < n_chunks) // stage the next chunk while this one is added
            for (uint32_t k = 0; k < 16; k++) {
                const size_t i = (c + 1) * 1024 + (size_t)k * 64 + lane;
                buf[(c + 1) & 1][k * 64 + lane] = i < n ? w[i] : 0.0f;
            }
        const float *b = buf[c & 1];
        const uint32_t m = (uint32_t)((n - c * 1024 < 1024) ? (n - c * 1024) : 1024);
        for (uint32_t j = 0; j < m; j++) sum = sum + b[j]; // (+0.0 padding is never added: m stops at n)
    }
    if (lane == 0) *sum_out = sum;
}

// step 3: p = w * N / sum (environments.rs:115), default entry {1, self, 1/N} (:163-178), small / large flag (:123-133)
__global__ void rt_alias_normalise_kernel(float *w_inout, size_t n, const float *sum, uint4 *out, uint32_t *block_small)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float nf = (float)n;
    bool small = false;
    if (i < n) {
        const float p = w_inout[i] * nf / *sum;
        w_inout[i] = p;
        small = p < 1.0f;
        out[i] = uint4{as_u(1.0f), (uint32_t)i, as_u(1.0f / nf), 0u};
    }
    // per-block count of smalls (256 threads = 4 waves)
    __shared__ uint32_t cnt[4];
    const unsigned long long m = __ballot(small);
    if ((threadIdx.x & 63u) == 0) cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_small[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// exclusive scan of the per-block counts (one workgroup; n_blocks is a few thousand)
__global__ __launch_bounds__(1024) void rt_alias_scan_kernel(uint32_t *block_small, uint32_t n_blocks, uint32_t *total_small)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u, t = threadIdx.x;
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t i = t * per + k; if (i < n_blocks) s += block_small[i]; }
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s; // exclusive prefix of this thread's range
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = t * per + k;
        if (i < n_blocks) { const uint32_t c = block_small[i]; block_small[i] = run; run += c; }
    }
    if (t == 1023u) *total_small = part[1023];
}

// the two index lists, ascending (the reference pushes indices 0..N-1 in order, :123-133)
__global__ void rt_alias_scatter_kernel(const float *p, size_t n, const uint32_t *block_small_prefix, uint32_t *small, uint32_t *large)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const bool is_small = in && p[i] < 1.0f;
    __shared__ uint32_t wave_small[4];
    const unsigned long long m = __ballot(is_small);
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) wave_small[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0; // smalls of this block in earlier waves
    for (uint32_t k = 0; k < wv; k++) before += wave_small[k];
    const uint32_t rank_small = block_small_prefix[blockIdx.x] + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (!in) return;
    if (is_small) small[rank_small] = (uint32_t)i;
    else large[(uint32_t)i - rank_small] = (uint32_t)i; // elements before i that are NOT small
}

// step 5: Vose pairing, LIFO (environments.rs:135-159).  One wave.  Every lane runs the same loop on the same values
// (wave-uniform: LDS broadcast reads), lane 0 stores the entries.  A large that stays >= 1 is pushed back and popped
// again at once; a large that drops below 1 is pushed on `small` and is the very next small popped: both are carried
// in registers, so memory is only touched for NEW stack entries — in stack order, from the back — and those are staged
// RT_ALIAS_CHUNK at a time by all 64 lanes (coalesced index loads, parallel gathers of p) and read one entry AHEAD of
// their use, so that the LDS latency overlaps the arithmetic of the current pair.  The loop stores {probability,
// alias} only; pmf = p / N of the entries that were assigned is a data-parallel pass afterwards (rt_alias_pmf_kernel).
__global__ __launch_bounds__(64) void rt_alias_vose_kernel(const float *p, size_t n, const uint32_t *small, const uint32_t *n_small_ptr,
                                                           const uint32_t *large, uint4 *out, uint32_t *leftover_out)
{
    __shared__ uint32_t s_idx[RT_ALIAS_CHUNK], l_idx[RT_ALIAS_CHUNK];
    __shared__ float s_p[RT_ALIAS_CHUNK], l_p[RT_ALIAS_CHUNK];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_small = rt_uniform(*n_small_ptr), n_large = (uint32_t)n - n_small;
    uint32_t s_left = n_small, l_left = n_large; // entries of the original stacks not yet fetched
    uint32_t s_pos = 0, s_have = 0, l_pos = 0, l_have = 0; // staging windows
    // the next entry of each original stack, fetched ahead
    bool ns_ok = false, nl_ok = false;
    uint32_t ns_i = 0, nl_i = 0;
    float ns_p = 0.0f, nl_p = 0.0f;
#define RT_ALIAS_FETCH(left, pos, have, idx, pv, stack, ok, oi, op)                                                         \
    do {                                                                                                                    \
        if ((left) == 0u) { ok = false; break; }                                                                            \
        if ((pos) == (have)) { /* refill: the next (up to) CHUNK entries from the back of the stack */                      \
            RT_ALIAS_WAVE_SYNC();                                                                                           \
            have = (left) < RT_ALIAS_CHUNK ? (left) : RT_ALIAS_CHUNK;                                                       \
            for (uint32_t j = lane; j < (have); j += 64u) { const uint32_t i = stack[(left) - 1u - j]; idx[j] = i; pv[j] = p[i]; } \
            pos = 0;                                                                                                        \
            RT_ALIAS_WAVE_SYNC();                                                                                           \
        }                                                                                                                   \
        oi = idx[pos]; op = pv[pos]; ok = true;                                                                             \
        pos++; left--;                                                                                                      \
    } while (0)
    RT_ALIAS_FETCH(s_left, s_pos, s_have, s_idx, s_p, small, ns_ok, ns_i, ns_p);
    RT_ALIAS_FETCH(l_left, l_pos, l_have, l_idx, l_p, large, nl_ok, nl_i, nl_p);
    bool have_dem = false, have_cur = false;
    uint32_t dem_i = 0, cur_i = 0;
    float dem_res = 0.0f, cur_res = 0.0f;
    uint32_t assigned = 0;
    uint2 *out2 = reinterpret_cast<uint2 *>(out);
    for (;;) {
        uint32_t s;
        float res_s;
        if (have_dem) {
            s = dem_i; res_s = dem_res; have_dem = false;
        } else {
            if (!ns_ok) break; // small.is_empty()
            s = rt_uniform(ns_i); res_s = rt_uniform(ns_p);
            RT_ALIAS_FETCH(s_left, s_pos, s_have, s_idx, s_p, small, ns_ok, ns_i, ns_p);
        }
        uint32_t l;
        float res_l;
        if (have_cur) {
            l = cur_i; res_l = cur_res;
        } else {
            if (!nl_ok) break; // large.is_empty(): the small just popped keeps its default entry
            l = rt_uniform(nl_i); res_l = rt_uniform(nl_p);
            RT_ALIAS_FETCH(l_left, l_pos, l_have, l_idx, l_p, large, nl_ok, nl_i, nl_p);
        }
        if (lane == 0) out2[2u * (size_t)s] = uint2{as_u(res_s), l}; // {probability, alias_index} (:143-150); pmf: rt_alias_pmf_kernel
        assigned++;
        res_l = rt_uniform(res_l - (1.0f - res_s)); // :152
        if (res_l < 1.0f) { have_dem = true; have_cur = false; dem_i = l; dem_res = res_l; }
        else { have_cur = true; cur_i = l; cur_res = res_l; }
    }
#undef RT_ALIAS_FETCH
    if (lane == 0) *leftover_out = (uint32_t)n - assigned;
}

// pmf of the assigned entries = p / N (:146); an assigned entry's alias is never itself, a leftover's always is
__global__ void rt_alias_pmf_kernel(const float *p, size_t n, uint4 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float nf = (float)n;
    if (out[i].y != (uint32_t)i) out[i].z = as_u(p[i] / nf);
}

// The packed device layout of an environment (rt_device.h, RT_ENV_PACKED): texel alpha := pmf of the texel's own alias entry,
// entry pad := pmf of the entry's alias target.  Copies of f32 values the kernels would otherwise gather from a second table.
// Two passes, because the second reads the z words of OTHER entries.
__global__ void rt_env_pack_texels_kernel(float4 *rgba, const uint4 *alias, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rgba[i].w = as_f(alias[i].z);
}
__global__ void rt_env_pack_alias_kernel(uint4 *alias, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    alias[i].w = alias[alias[i].y].z; // (only .w is written, only .y / .z are read: no race)
}

