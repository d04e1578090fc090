// rt_math.h — f32 vector helpers for the gfx950 kernels.
//
// Arithmetic contract (identical on the CPU checker side, see DESIGN.md "Numerics"):
//  * compiled with -ffp-contract=off: each written * + - / sqrt is ONE correctly rounded IEEE
//    binary32 operation (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt, denormals kept);
//  * the only fused multiply-adds are the explicit fmaf() calls in dot / dot2 / cross /
//    mat3_mul / madd below;
//  * normalize(v) = v * (1/sqrt(dot(v,v))), inverse_sqrt(x) = 1/sqrt(x), length = sqrt(dot);
//  * rt_rcp(x) IS the correctly rounded 1.0f / x, computed in 3 instructions instead of the compiler's 11
//    where that is exhaustively verified to give the same bits (rsrt_selftest_numerics);
//  * fmax_/fmin_ are the compare-select forms (a<b?b:a / b<a?b:a), NaN handling included;
//  * transcendental functions come from include/rsrt_detmath.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/rsrt_detmath.h"

// -DRT_FAST_NUMERICS (experiment build only; tools/fast_numerics.py): what the bit-exact numeric contract costs in run
// time.  The hardware's own approximations replace the contract's functions — v_sin_f32 / v_cos_f32, ocml atan2f /
// asinf, and (by the flags the tool adds: -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast) the 2.5-ulp
// division and square root and free fma contraction — everything WGSL would allow.  NOT a product mode: its image
// cannot be validated against the 1e-3 bar ("What parity unpinned costs").
#ifdef RT_FAST_NUMERICS
#define rsrt_sinf __sinf
#define rsrt_cosf __cosf
#define rsrt_atan2f atan2f
#define rsrt_asinf asinf
#endif

#define RT_DEV __device__ __forceinline__

// Region marks for the overhead ledger (tools/ledger.py).  Nothing in the product build.  -DRT_LEDGER (an analysis build that is
// only disassembled, never run): two s_nop in a row mark the border of a code region, so that the VALU instructions of each region can be
// counted in the disassembly.  -DRT_INSTRUMENT (librsrt_instr.so): the lanes that pass a mark are counted (one atomic per wave and mark:
// rsrt_get_region_counters) — static instructions x lanes = lane-instructions by region.  RT_MARK_COLD: the entry of a path that next to no
// lane takes (the full division behind the exact reciprocal, the tree-walk fallback of axis-parallel rays): the count of a region stops there.
#if defined(RT_LEDGER)
#define RT_MARK(k) asm volatile("s_nop 11\n\ts_nop " #k ::: "memory")
#define RT_MARK2(k) asm volatile("s_nop 12\n\ts_nop " #k ::: "memory")
#define RT_MARK_COLD() asm volatile("s_nop 13\n\ts_nop 0" ::: "memory")
#elif defined(RT_INSTRUMENT)
__device__ unsigned long long rt_region_lanes[32];
#define RT_MARK_N(i)                                                                                                                          \
    do {                                                                                                                                      \
        const unsigned long long m_ = __ballot(true);                                                                                         \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m_)) atomicAdd(&rt_region_lanes[i], (unsigned long long)__popcll(m_));           \
    } while (0)
#define RT_MARK(k) RT_MARK_N(k)
#define RT_MARK2(k) RT_MARK_N(16 + (k))
#define RT_MARK_COLD() do { } while (0)
#else
#define RT_MARK(k) do { } while (0)
#define RT_MARK2(k) do { } while (0)
#define RT_MARK_COLD() do { } while (0)
#endif

struct V3 {
    float x, y, z;
};
RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RT_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_DEV float comp(V3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

RT_DEV float fmax_(float a, float b) { return a < b ? b : a; }
RT_DEV float fmin_(float a, float b) { return b < a ? b : a; }
RT_DEV float saturate(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }
RT_DEV float fabs_(float x) { return __builtin_fabsf(x); }

RT_DEV float dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV float dot2(float ax, float ay, float bx, float by) { return __builtin_fmaf(ay, by, ax * bx); }
RT_DEV V3 cross(V3 a, V3 b)
{
    return V3{__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)),
              __builtin_fmaf(a.x, b.y, -(b.x * a.y))};
}
// o + d*t
RT_DEV V3 madd(V3 d, float t, V3 o) { return V3{__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z)}; }
// column-major 3x3 times vector
RT_DEV V3 mat3_mul(V3 c0, V3 c1, V3 c2, V3 v)
{
    return V3{__builtin_fmaf(c2.x, v.z, __builtin_fmaf(c1.x, v.y, c0.x * v.x)),
              __builtin_fmaf(c2.y, v.z, __builtin_fmaf(c1.y, v.y, c0.y * v.x)),
              __builtin_fmaf(c2.z, v.z, __builtin_fmaf(c1.z, v.y, c0.z * v.x))};
}

RT_DEV float as_f(uint32_t u) { return __uint_as_float(u); }
RT_DEV uint32_t as_u(float f) { return __float_as_uint(f); }

// 1.0f / x, correctly rounded.  For an exponent field in [2, 252] (|x| in [2^-125, 2^126): neither x nor
// 1/x is subnormal) v_rcp_f32 plus ONE Newton step in fma arithmetic equals the IEEE quotient for every
// one of the 2 * 251 * 2^23 inputs — checked exhaustively on the device by rsrt_selftest_numerics, which
// tests/test_gpu_parity.py runs.  Everything else (zeros, subnormals, huge, inf, NaN) takes the
// compiler's full division; the empty asm keeps that path a branch instead of a select of both.
RT_DEV float rt_rcp(float x)
{
#if defined(RT_FAST_NUMERICS) || defined(RT_FAST_RCP) // (RT_FAST_RCP: the ledger's ablation build, tools/r04_ledger.sh — what the exact reciprocal costs beyond v_rcp_f32)
    return __builtin_amdgcn_rcpf(x);
#endif
    const uint32_t ex = (as_u(x) >> 23) & 0xffu;
    if (ex - 2u < 251u) {
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r0, 1.0f);
        return __builtin_fmaf(e, r0, r0);
    }
    asm volatile("; rt_rcp: full division");
    RT_MARK_COLD();
    return 1.0f / x;
}

// The three reciprocals of a direction behind ONE branch: the short form for all of them when every exponent allows it, the full division
// for all of them otherwise (it is the IEEE quotient everywhere, so taking it for an in-range value changes nothing).  One
// straight-line block instead of one branch per reciprocal (fixed-order walk: -1 %; the same for the two determinants of a
// triangle pair measured 0.4 % slower on the BASELINE frame and is not used).
RT_DEV bool rt_rcp_short_ok(float x) { return ((as_u(x) >> 23) & 0xffu) - 2u < 251u; }
RT_DEV float rt_rcp_short(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
}
RT_DEV V3 rt_rcp3(V3 d)
{
#if defined(RT_FAST_NUMERICS) || defined(RT_FAST_RCP)
    return V3{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)};
#endif
    if (((int)rt_rcp_short_ok(d.x) & (int)rt_rcp_short_ok(d.y) & (int)rt_rcp_short_ok(d.z)) != 0) return V3{rt_rcp_short(d.x), rt_rcp_short(d.y), rt_rcp_short(d.z)};
    asm volatile("; rt_rcp3: full division");
    RT_MARK_COLD();
    return V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
}

RT_DEV float length(V3 a) { return rsrt_sqrtf(dot(a, a)); }
RT_DEV V3 normalize(V3 a) { return a * rt_rcp(rsrt_sqrtf(dot(a, a))); }
RT_DEV float inverse_sqrt(float x) { return rt_rcp(rsrt_sqrtf(x)); }

// WGSL u32(f32) as naga emits it: NaN / negative -> 0, clamp at the largest f32 below 2^32
RT_DEV uint32_t f2u(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967040.0f) return 4294967040u;
    return (uint32_t)x;
}

