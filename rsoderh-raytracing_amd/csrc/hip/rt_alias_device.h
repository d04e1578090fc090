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

// step 2: the sequential f32 sum.  One wave.  The 64 weights of a chunk sit one per lane (a coalesced load, sixteen chunks requested ahead
// of the one being added); the sum itself is a chain of 64 dependent v_add_f32 on a wave-uniform register, each taking its addend from the
// next lane with v_readlane: two instructions an addend, nothing but the adds on the dependent path (the first version read every addend
// back from LDS: 10 ns each; this one ~3).
#define RT_ALIAS_SUM_AHEAD 16
__global__ __launch_bounds__(64) void rt_alias_sum_kernel(const float *w, size_t n, float *sum_out)
{
    const uint32_t lane = threadIdx.x;
    float sum = 0.0f;
    const size_t n_full = n / 64; // whole chunks; the tail (n % 64 addends) is added below, one at a time, and nothing is ever padded
    float ring[RT_ALIAS_SUM_AHEAD];
#pragma unroll
    for (int k = 0; k < RT_ALIAS_SUM_AHEAD; k++) ring[k] = (size_t)k < n_full ? w[(size_t)k * 64 + lane] : 0.0f;
    for (size_t c0 = 0; c0 < n_full; c0 += RT_ALIAS_SUM_AHEAD) {
#pragma unroll
        for (int k = 0; k < RT_ALIAS_SUM_AHEAD; k++) { // (unrolled: ring[k] is a register)
            const size_t c = c0 + k;
            if (c < n_full) {
                const float v = ring[k];
                const size_t nx = c + RT_ALIAS_SUM_AHEAD;
                ring[k] = nx < n_full ? w[nx * 64 + lane] : 0.0f; // the chunk sixteen ahead takes this register
                // (eight addends fetched, then eight adds: the adds wait for each other, not each for its own v_readlane and the hazard behind it)
#pragma unroll
                for (int j = 0; j < 64; j += 8) {
                    float a[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) a[q] = as_f((uint32_t)__builtin_amdgcn_readlane((int)as_u(v), j + q));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; q++) sum = sum + a[q];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    const uint32_t tail = (uint32_t)(n - n_full * 64);
    if (tail) {
        const float v = lane < tail ? w[n_full * 64 + lane] : 0.0f;
        for (uint32_t j = 0; j < tail; j++) sum = sum + as_f((uint32_t)__builtin_amdgcn_readlane((int)as_u(v), (int)j));
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

// step 5: Vose pairing, LIFO (environments.rs:135-159).  One wave, and the loop of the reference seen from the LARGE that is being
// used up: a large that stays >= 1 is pushed back and popped again at once, so it takes small after small from the stack until its
// residual drops below 1; then it is itself the next small, and pairs with the next large.  What is sequential is the residual —
// r <- r - (1 - p_small), one rounding per small, in stack order — and only that: so 64 smalls are taken at a time, one per lane (their
// 1 - p computed in parallel), and the residual runs through them as a chain of dependent v_sub_f32 on a wave-uniform register, each
// step's operand fetched from the next lane with v_readlane and each step's result kept by "its" lane; a ballot then finds the first
// small that brought the residual below 1, the lanes up to it store their entries {p_small, this large} at once, and the rest of the
// 64 stay on the stack for the next large.  The stacks are staged RT_ALIAS_CHUNK entries at a time from their ends by all lanes
// (coalesced index loads, parallel gathers of p).  pmf = p / N of the assigned entries is a data-parallel pass afterwards.
// The first version advanced one small per trip of a ~60-instruction loop (250 ns); a step of the chain is four instructions.
__global__ __launch_bounds__(64) void rt_alias_vose_kernel(const float *p, size_t n, const uint32_t *small, const uint32_t *n_small_ptr,
                                                           const uint32_t *large, uint4 *out, uint32_t *leftover_out)
{
    __shared__ uint32_t s_idx[RT_ALIAS_CHUNK], l_idx[RT_ALIAS_CHUNK];
    __shared__ float s_p[RT_ALIAS_CHUNK], l_p[RT_ALIAS_CHUNK];
    __shared__ uint32_t s_al[RT_ALIAS_CHUNK]; // the large each staged small has gone to: the entries are written out a chunk at a time (a store
                                              // per round would be waited for, scattered as it is, at the next s_waitcnt the round meets)
    const uint32_t lane = threadIdx.x;
    const uint32_t n_small = rt_uniform(*n_small_ptr), n_large = (uint32_t)n - n_small;
    uint2 *out2 = reinterpret_cast<uint2 *>(out);
    uint32_t s_left = n_small, l_left = n_large; // entries of the original stacks not yet consumed
    uint32_t s_pos = 0, s_have = 0, l_pos = 0, l_have = 0; // staging windows: entries [pos, have) of a chunk are staged and not yet consumed
    // refill: the next (up to) CHUNK entries from the back of the stack (only when the window is empty).  All of a lane's index loads go out
    // together, then all of its gathers of p: two memory round trips a refill (a loop that loads an index, waits, gathers, waits, thirty-two
    // times over, was most of this kernel's time)
    static_assert(RT_ALIAS_CHUNK % 64u == 0u, "a refill is whole rows of 64 lanes");
#define RT_ALIAS_REFILL(left, pos, have, idx, pv, stack)                                                                    \
    do {                                                                                                                    \
        RT_ALIAS_WAVE_SYNC();                                                                                               \
        have = (left) < RT_ALIAS_CHUNK ? (left) : RT_ALIAS_CHUNK;                                                           \
        uint32_t ii_[RT_ALIAS_CHUNK / 64u];                                                                                 \
        float pp_[RT_ALIAS_CHUNK / 64u];                                                                                    \
        _Pragma("unroll") for (uint32_t q = 0; q < RT_ALIAS_CHUNK / 64u; q++) { const uint32_t j = lane + 64u * q; ii_[q] = j < (have) ? stack[(left) - 1u - j] : 0u; } \
        _Pragma("unroll") for (uint32_t q = 0; q < RT_ALIAS_CHUNK / 64u; q++) { const uint32_t j = lane + 64u * q; pp_[q] = j < (have) ? p[ii_[q]] : 0.0f; }         \
        _Pragma("unroll") for (uint32_t q = 0; q < RT_ALIAS_CHUNK / 64u; q++) { const uint32_t j = lane + 64u * q; if (j < (have)) { idx[j] = ii_[q]; pv[j] = pp_[q]; } } \
        pos = 0;                                                                                                            \
        RT_ALIAS_WAVE_SYNC();                                                                                               \
    } while (0)
    // {probability, alias_index} (:143-150) of the first `count` staged smalls
#define RT_ALIAS_FLUSH(count)                                                                                              \
    do {                                                                                                                    \
        RT_ALIAS_WAVE_SYNC();                                                                                               \
        for (uint32_t j = lane; j < (count); j += 64u) out2[2u * (size_t)s_idx[j]] = uint2{as_u(s_p[j]), s_al[j]};          \
    } while (0)
    bool have_dem = false, have_cur = false; // a large that has dropped below 1 and is the next small / the large in use (>= 1)
    uint32_t dem_i = 0, cur_i = 0;
    float dem_res = 0.0f, cur_res = 0.0f;
    uint32_t assigned = 0;
    for (;;) {
        if (!have_cur) { // a large is needed: for the demoted one if there is one, else for the smalls of the stack
            if (!have_dem && s_left == 0u) break; // small.is_empty()
            if (l_left == 0u) break;              // large.is_empty(): whatever small was popped keeps its default entry
            if (l_pos == l_have) RT_ALIAS_REFILL(l_left, l_pos, l_have, l_idx, l_p, large);
            const uint32_t l = rt_uniform(l_idx[l_pos]);
            float res_l = rt_uniform(l_p[l_pos]);
            l_pos++; l_left--;
            if (have_dem) { // the demoted large is the small of this pair (:143-152)
                if (lane == 0) out2[2u * (size_t)dem_i] = uint2{as_u(dem_res), l};
                assigned++;
                res_l = rt_uniform(res_l - (1.0f - dem_res));
                have_dem = false;
                if (res_l < 1.0f) { have_dem = true; dem_i = l; dem_res = res_l; continue; }
            }
            have_cur = true; cur_i = l; cur_res = res_l;
        }
        // cur takes smalls from the stack, up to 64 at a time
        if (s_left == 0u) break; // small.is_empty()
        if (s_pos == s_have) {
            if (s_have != 0u) RT_ALIAS_FLUSH(s_have); // the chunk is used up: its entries go out together
            RT_ALIAS_REFILL(s_left, s_pos, s_have, s_idx, s_p, small);
        }
        // Rounds of a large that 64 smalls cannot use up (each takes 1 - p <= 1 from it, and a difference that is >= 1 exactly is >= 1
        // rounded) — nearly all rounds where a few hundred larges take thousands of smalls each: nothing to look for in between, so these
        // run as a loop of their own, the chain and little else, with the next round's operands read from LDS while the chain runs
        // (1 - p <= 1 holds for p >= 0, i.e. for every environment without negative texels; a round that meets anything else — a negative
        // weight — goes through the looked-at blocks below, which need nothing but 1 - p > 0, true of every small)
        bool wild = false;
        if (cur_res >= 65.0f && s_have - s_pos >= 64u) {
            float r = cur_res;
            float dd = 1.0f - s_p[s_pos + lane];
            do {
                const float d_now = dd;
                if (__ballot(!(d_now <= 1.0f)) != 0ull) { wild = true; break; }
                if (s_have - s_pos >= 128u) dd = 1.0f - s_p[s_pos + 64u + lane]; // (the next round's, if it is staged)
                _Pragma("unroll") for (int k = 0; k < 64; k += 8) {
                    float d8[8];
                    _Pragma("unroll") for (int q = 0; q < 8; q++) d8[q] = as_f((uint32_t)__builtin_amdgcn_readlane((int)as_u(d_now), k + q));
                    __builtin_amdgcn_sched_barrier(0);
                    _Pragma("unroll") for (int q = 0; q < 8; q++) r = r - d8[q];
                    __builtin_amdgcn_sched_barrier(0);
                }
                s_al[s_pos + lane] = cur_i;
                s_pos += 64u; s_left -= 64u; assigned += 64u;
            } while (rt_uniform(r) >= 65.0f && s_have - s_pos >= 64u);
            cur_res = r;
            if (!wild) continue;
        }
        // Otherwise (a large that may be used up in this round — r only ever falls, since 1 - p > 0 for every small): the chain in
        // straight-line blocks of 1, 1, 2, 4, 8, 16, 32 steps, two instructions a step (a loop with exits in it is not unrolled, and a
        // rolled loop costs ~200 cycles a step).  Between two blocks the wave looks whether the residual has dropped below 1; the block in
        // which it did is walked again from its starting value, step by step, to find the small that did it.
        const uint32_t m = (s_have - s_pos) < 64u ? (s_have - s_pos) : 64u; // smalls of this round: lane k has the k-th from the top
        const float my_d = lane < m ? 1.0f - s_p[s_pos + lane] : 0.0f;      // (a lane beyond the round's smalls subtracts 0: r - 0 is r, exactly)
        float r = cur_res, r0 = cur_res;
        uint32_t k0 = 0u, k1 = 0u; // the block [k0, k1) in which the residual dropped below 1 (k1 == 0: it did not)
#define RT_ALIAS_CHAIN(K0, K1)                                                                                                            \
        r0 = r;                                                                                                                            \
        _Pragma("unroll") for (int k = (K0); k < (K1); k += 8) { /* (operands fetched eight at a time, then the eight dependent subtractions) */ \
            float dd[8];                                                                                                                   \
            _Pragma("unroll") for (int q = 0; q < 8; q++) dd[q] = (k + q) < (K1) ? as_f((uint32_t)__builtin_amdgcn_readlane((int)as_u(my_d), (k + q) < 64 ? (k + q) : 63)) : 0.0f; \
            __builtin_amdgcn_sched_barrier(0);                                                                                             \
            _Pragma("unroll") for (int q = 0; q < 8; q++) if ((k + q) < (K1)) r = r - dd[q];                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                             \
        }                                                                                                                                  \
        if (rt_uniform(r) < 1.0f) { k0 = (K0); k1 = (K1); }
        RT_ALIAS_CHAIN(0, 1)
        if (k1 == 0u && m > 1u) {
            RT_ALIAS_CHAIN(1, 2)
            if (k1 == 0u && m > 2u) {
                RT_ALIAS_CHAIN(2, 4)
                if (k1 == 0u && m > 4u) {
                    RT_ALIAS_CHAIN(4, 8)
                    if (k1 == 0u && m > 8u) {
                        RT_ALIAS_CHAIN(8, 16)
                        if (k1 == 0u && m > 16u) {
                            RT_ALIAS_CHAIN(16, 32)
                            if (k1 == 0u && m > 32u) { RT_ALIAS_CHAIN(32, 64) }
                        }
                    }
                }
            }
        }
#undef RT_ALIAS_CHAIN
        uint32_t n_abs = m; // smalls this large takes in this round: all of them, or up to the one that brings it below 1
        float r_last = r;
        const bool below = k1 != 0u;
        if (below) {
            r_last = r0;
            for (uint32_t k = k0; k < k1; k++) { // (wave-uniform)
                r_last = r_last - as_f((uint32_t)__builtin_amdgcn_readlane((int)as_u(my_d), (int)k));
                if (rt_uniform(r_last) < 1.0f) { n_abs = k + 1u; break; }
            }
        }
        if (lane < n_abs) s_al[s_pos + lane] = cur_i; // (written out by RT_ALIAS_FLUSH)
        assigned += n_abs;
        s_pos += n_abs; s_left -= n_abs;
        if (below) { have_cur = false; have_dem = true; dem_i = cur_i; dem_res = r_last; } // :153-157: it is a small now
        else cur_res = r_last;
    }
    if (s_pos != 0u) RT_ALIAS_FLUSH(s_pos); // the smalls of the last chunk that were used
#undef RT_ALIAS_FLUSH
#undef RT_ALIAS_REFILL
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

