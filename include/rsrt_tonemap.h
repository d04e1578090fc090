/*
 * rsrt_tonemap.h — display transform of the reference's second pass, as shared inline code.
 *
 * Reference: src/shaders/hdr.wgsl:3-22 (`aces_tone_map`; negatives -> magenta, :4-7) applied to the
 * RGBA16F mean `out_texture` (shader.wgsl:1369-1372, src/hdr.rs:209-216), written to an *Srgb surface
 * (src/hdr.rs:120), i.e. followed by the sRGB OETF and 8-bit quantisation.  Like rsrt_detmath.h this
 * is part of the published numeric contract: plain f32 add/mul/div (-ffp-contract=off, nothing fused,
 * so a numpy float32 restatement reproduces it bit for bit — tests/test_display.py), the OETF by
 * threshold search.
 */
#ifndef RSRT_TONEMAP_H
#define RSRT_TONEMAP_H

#include "rsrt_detmath.h"
#include "rsrt_srgb_table.h"

/* mat3x3(c0, c1, c2) * v with columns as WGSL lists them (hdr.wgsl:8-17) */
RSRT_HD void rsrt_mat3_mul(const float c0[3], const float c1[3], const float c2[3], const float v[3], float out[3])
{
    for (int i = 0; i < 3; i++) out[i] = (c0[i] * v[0] + c1[i] * v[1]) + c2[i] * v[2];
}

RSRT_HD void rsrt_aces_tone_map(const float hdr[3], float out[3])
{
    if (hdr[0] < 0.0f || hdr[1] < 0.0f || hdr[2] < 0.0f) { out[0] = 1.0f; out[1] = 0.0f; out[2] = 1.0f; return; }
    const float m1c0[3] = {0.59719f, 0.07600f, 0.02840f}, m1c1[3] = {0.35458f, 0.90834f, 0.13383f}, m1c2[3] = {0.04823f, 0.01566f, 0.83777f};
    const float m2c0[3] = {1.60475f, -0.10208f, -0.00327f}, m2c1[3] = {-0.53108f, 1.10813f, -0.07276f}, m2c2[3] = {-0.07367f, -0.00605f, 1.07602f};
    float v[3], q[3], r[3];
    rsrt_mat3_mul(m1c0, m1c1, m1c2, hdr, v);
    for (int i = 0; i < 3; i++) {
        const float a = v[i] * (v[i] + 0.0245786f) - 0.000090537f;
        const float b = v[i] * (0.983729f * v[i] + 0.4329510f) + 0.238081f;
        q[i] = a / b;
    }
    rsrt_mat3_mul(m2c0, m2c1, m2c2, q, r);
    for (int i = 0; i < 3; i++) { /* clamp(x, 0, 1) = min(max(x, 0), 1) */
        float x = r[i] < 0.0f ? 0.0f : r[i];
        out[i] = 1.0f < x ? 1.0f : x;
    }
}

/* number of thresholds <= x (NaN -> 0) */
RSRT_HD unsigned rsrt_srgb8_encode(float x)
{
    unsigned lo = 0, hi = 255; /* invariant: T[lo-1] <= x < T[hi] with T[-1] = -inf, T[255] = +inf */
    while (lo < hi) {
        const unsigned mid = (lo + hi) / 2;
        if (x >= RSRT_SRGB8_THRESHOLDS[mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* IEEE binary16 round-to-nearest-even of an f32, returned as the f32 it represents
 * (what storing to and sampling from an rgba16float texture does) */
RSRT_HD float rsrt_round_to_f16(float f)
{
    union { float f; uint32_t u; } in, out;
    in.f = f;
    const uint32_t sign = in.u & 0x80000000u;
    uint32_t a = in.u & 0x7fffffffu;
    if (a >= 0x7f800000u) { out.u = in.u; return out.f; }            /* inf / NaN stay */
    if (a >= 0x477ff000u) { out.u = sign | 0x7f800000u; return out.f; } /* rounds to >= 65520: +-inf */
    if (a < 0x38800000u) { /* below the smallest normal half: result is a multiple of 2^-24 */
        in.u = a;
        const float scaled = in.f * 16777216.0f; /* exact */
        float r = scaled + 8388608.0f;           /* adding 2^23 rounds to an integer, ties to even */
        r = r - 8388608.0f;
        out.f = r * 5.9604644775390625e-08f;     /* exact */
        out.u |= sign;
        return out.f;
    }
    /* normal half: keep 10 mantissa bits, ties to even */
    const uint32_t lsb = (a >> 13) & 1u;
    a = (a + 0x0fffu + lsb) & 0xffffe000u;
    out.u = sign | a;
    return out.f;
}

/* one pixel: sums (f32) -> display bytes */
RSRT_HD void rsrt_display_pixel(const float sum_rgb[3], float sample_total, unsigned char out_rgb[3])
{
    float mean[3], sdr[3];
    for (int i = 0; i < 3; i++) mean[i] = rsrt_round_to_f16(sum_rgb[i] / sample_total);
    rsrt_aces_tone_map(mean, sdr);
    for (int i = 0; i < 3; i++) out_rgb[i] = (unsigned char)rsrt_srgb8_encode(sdr[i]);
}

#endif
