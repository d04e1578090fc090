/*
 * rsrt_detmath.h — deterministic single-precision transcendental functions.
 *
 * Part of the numerical contract of the rsrt C-ABI (include/rsrt.h): every sin / cos /
 * atan2 / asin the integrator evaluates (reference call sites: src/shaders/shader.wgsl:628-630,
 * :711-712, :724-731, :744, :894-897, :909-910, :1353) is evaluated with exactly these
 * routines.  WGSL leaves the precision of these built-ins implementation-defined, so any
 * accurate implementation is a valid restatement; fixing ONE implementation made of IEEE-754
 * f32 add / sub / mul / div / sqrt only (each correctly rounded on x86-64 and on gfx950) is
 * what lets a CPU evaluation and the HIP kernels agree bit for bit, so that a path never
 * forks between the two because of a 1-ulp libm difference.
 *
 * Algorithms: the classic Cody–Waite three-constant range reduction + minimax polynomials
 * of the Cephes single-precision library (sinf/cosf/atanf/asinf), re-derived here in plain
 * f32 operations (no fused multiply-add, no table, no double).  Measured accuracy vs a
 * float64 libm on the ranges the integrator uses: sin/cos <= 1.5 ulp, asin <= 2.4 ulp,
 * atan2 <= 3.1 ulp (tests/test_detmath.py asserts <= 4).
 *
 * Must be compiled with -ffp-contract=off (both gcc and hipcc) — the build scripts do so.
 */
#ifndef RSRT_DETMATH_H
#define RSRT_DETMATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RSRT_HD __host__ __device__ inline
#else
#define RSRT_HD inline
#endif

#include <math.h>
#include <stdint.h>

/* sqrtf is IEEE correctly rounded on both targets (hipcc default
 * -fhip-fp32-correctly-rounded-divide-sqrt). */
RSRT_HD float rsrt_sqrtf(float x) { return sqrtf(x); }

RSRT_HD float rsrt_nanf() { return __builtin_nanf(""); }

/* |x| reduced by multiples of pi/4 with a 3-part constant (y*DP1 and y*DP2 are exact for the
 * small y that occur: DP1 has 8 significant bits, DP2 has 11). */
#define RSRT_DP1 0.78515625f
#define RSRT_DP2 2.4187564849853515625e-4f
#define RSRT_DP3 3.77489497744594108e-8f
#define RSRT_FOPI 1.27323954473516f /* 4/pi */

RSRT_HD float rsrt_sin_poly(float r, float z)
{
    float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
    p = p * z - 1.6666654611e-1f;
    return p * z * r + r;
}

RSRT_HD float rsrt_cos_poly(float z)
{
    float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    p = p * z * z;
    p = p - 0.5f * z;
    return p + 1.0f;
}

RSRT_HD float rsrt_sinf(float x)
{
    float ax = x < 0.0f ? -x : x;
    bool neg = x < 0.0f;
    if (!(ax <= 8192.0f)) return x - x; /* NaN for NaN/inf; arguments this large never occur */
    uint32_t j = (uint32_t)(ax * RSRT_FOPI);
    j += (j & 1u);
    float y = (float)j;
    float r = ((ax - y * RSRT_DP1) - y * RSRT_DP2) - y * RSRT_DP3;
    j &= 7u;
    if (j > 3u) { neg = !neg; j -= 4u; }
    float z = r * r;
    float v = (j == 1u || j == 2u) ? rsrt_cos_poly(z) : rsrt_sin_poly(r, z);
    return neg ? -v : v;
}

RSRT_HD float rsrt_cosf(float x)
{
    float ax = x < 0.0f ? -x : x;
    bool neg = false;
    if (!(ax <= 8192.0f)) return x - x;
    uint32_t j = (uint32_t)(ax * RSRT_FOPI);
    j += (j & 1u);
    float y = (float)j;
    float r = ((ax - y * RSRT_DP1) - y * RSRT_DP2) - y * RSRT_DP3;
    j &= 7u;
    if (j > 3u) { neg = !neg; j -= 4u; }
    if (j > 1u) neg = !neg;
    float z = r * r;
    float v = (j == 1u || j == 2u) ? rsrt_sin_poly(r, z) : rsrt_cos_poly(z);
    return neg ? -v : v;
}

#define RSRT_PIF 3.14159265358979323846f
#define RSRT_PIO2F 1.57079632679489661923f
#define RSRT_PIO4F 0.78539816339744830962f

RSRT_HD float rsrt_atanf(float xx)
{
    bool neg = xx < 0.0f;
    float x = neg ? -xx : xx;
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = RSRT_PIO2F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y = RSRT_PIO4F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    y = y + (p * z * x + x);
    return neg ? -y : y;
}

/* atan2(y, x) in (-pi, pi]; signed zeros are not distinguished (y = -0 behaves as +0). */
RSRT_HD float rsrt_atan2f(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (x == 0.0f) {
        if (y > 0.0f) return RSRT_PIO2F;
        if (y < 0.0f) return -RSRT_PIO2F;
        return 0.0f;
    }
    if (y == 0.0f) return x < 0.0f ? RSRT_PIF : 0.0f;
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -RSRT_PIF : RSRT_PIF;
    return w + rsrt_atanf(y / x);
}

RSRT_HD float rsrt_asinf(float xx)
{
    bool neg = xx < 0.0f;
    float a = neg ? -xx : xx;
    if (!(a <= 1.0f)) return rsrt_nanf(); /* |x| > 1 or NaN: as WGSL asin (shader.wgsl:712) */
    if (a < 1.0e-4f) return xx;
    float x, z;
    bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        x = rsrt_sqrtf(z);
    } else {
        x = a;
        z = x * x;
    }
    float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    float r = p * z * x + x;
    if (big) {
        r = r + r;
        r = RSRT_PIO2F - r;
    }
    return neg ? -r : r;
}

#endif /* RSRT_DETMATH_H */
