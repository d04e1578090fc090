// rt_wavepool.h — the stage-scheduled path-tracing kernel (rt_render_pool_kernel).
//
// Why: the first kernel (one lane = one path, all stages in lockstep; rt_render_kernel) keeps only
// ~13 of 64 lanes active per VALU instruction (profiles/r01_v1_*): lanes wait for each other across
// stages of very different length.  Here a WAVE owns a pool of POOL path slots, every slot carries a
// stage tag, and the wave repeatedly
//     1. counts the slots waiting in each stage (wave64 ballots over the tag column),
//     2. picks the fullest stage,
//     3. compacts that stage's slot ids into a dense list (ballot + prefix popcount) — lane i
//        takes list[i] — and runs ONLY that stage's code for up to 64 slots at once.
// A path's arithmetic is untouched (same functions, same order), so the image is bit-identical;
// only which lane executes which step, and when, changes.  No workgroup barrier is used after the
// scene is staged: waves are independent.
//
// Where the state lives (measured: run time scales ~linearly with resident waves, and LDS is what
// limits them): the HOT columns — what the traversal loop touches: ray, traversal cursor, best hit,
// tag — are in LDS, 9 dwords per slot; the COLD columns — throughput, radiance, RNG, NEE carry,
// shading normal, the best hit's record / barycentrics ... only read/written by the shading stages
// (and once per improved hit) — are in a global-memory arena (25 dwords per slot, one contiguous
// column per field and wave, so a stage's accesses coalesce).  POOL = 192 then costs 7.2 KB of LDS
// per wave: 16 waves per CU, with 3 slots per lane to pick a full stage from.
//
// Stages: GEN (take the next (pixel, sample) of the wave's chunk, build the camera ray)
//         TRACE (extension rays: closest hit; NEE shadow rays: any hit — one shared, resumable loop)
//         MISS (cast_ray's brute-force fallback, then the escape to the environment)
//         SHADE (resolve the hit, emission, sample the environment for NEE)
//         BSDF (NEE contribution, BSDF sample, throughput, termination)
#pragma once
#include "rt_device.h"

enum HotField { H_OX, H_OY, H_OZ, H_DX, H_DY, H_DZ, H_T, H_CT, H_COUNT }; // H_CT: traversal cursor << 3 | stage tag
enum ColdField {
    C_TX, C_TY, C_TZ, C_LX, C_LY, C_LZ, C_LASTPDF, C_RNG, C_BOUNCE, C_OUT,
    C_WX, C_WY, C_WZ,             // direction the path arrived with at the current hit (wo = -w)
    C_NX, C_NY, C_NZ, C_MAT,      // shading normal and material of the current hit
    C_ERX, C_ERY, C_ERZ, C_EPDF, C_COS, // NEE sample: radiance, pdf, cos (0 = no contribution)
    C_REF,                        // best hit of the extension ray: record | source << 30 (rewritten only when a TRACE call improves it)
    C_COUNT
};
enum PoolTag { TAG_FREE = 0, TAG_TRACE_EXT = 1, TAG_TRACE_SHADOW = 2, TAG_MISS = 3, TAG_SHADE = 4, TAG_BSDF = 5, TAG_IDLE = 6 };
enum PoolStage { ST_GEN = 0, ST_TRACE = 1, ST_MISS = 2, ST_SHADE = 3, ST_BSDF = 4, ST_COUNT = 5 };

template <uint32_t POOL>
struct PoolLayout {
    static constexpr uint32_t kSlotsPerLane = (POOL + 63u) / 64u;
    static constexpr uint32_t kHotDwords = H_COUNT * POOL;
    static constexpr uint32_t kListDwords = 64u;
    static constexpr uint32_t kWaveLdsDwords = kHotDwords + kListDwords;
    static constexpr uint32_t kWaveColdDwords = C_COUNT * POOL;
};

RT_DEV uint32_t stage_of_tag(uint32_t tag)
{
    // FREE->GEN, TRACE_EXT/TRACE_SHADOW->TRACE, MISS, SHADE, BSDF; IDLE -> none
    return tag == TAG_FREE ? ST_GEN : (tag <= TAG_TRACE_SHADOW ? ST_TRACE : (tag == TAG_IDLE ? ST_COUNT : tag - 1u));
}

#ifndef RT_POOL_WAVES_PER_SIMD
#define RT_POOL_WAVES_PER_SIMD 5
#endif
// TRAV: which traversal TRACE runs — 0 trace_threaded (any BVH), 1 trace_threaded_typed (leaves of <= 8
// primitives), 2 trace_flat (<= 64 primitive records, nested boxes; rays with a non-finite 1/d fall back to 0)
template <bool LDS, uint32_t POOL, int TRAV>
__global__ __launch_bounds__(RT_BLOCK, RT_POOL_WAVES_PER_SIMD) void rt_render_pool_kernel(RenderParams P)
{
    typedef PoolLayout<POOL> L;
    const DevScene &sc = P.scene;
    if (LDS) stage_scene_lds(sc);
    const SceneView<LDS> S = make_view<LDS>(sc);
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t wave = threadIdx.x / RT_WAVE;
    uint32_t *const lds32 = reinterpret_cast<uint32_t *>(rt_smem + sc.lds_float4s);
    uint32_t *const W = lds32 + wave * L::kWaveLdsDwords; // this wave's hot columns
    uint32_t *const list = W + L::kHotDwords;
    uint32_t *const G = P.cold_state + (size_t)(blockIdx.x * (RT_BLOCK / RT_WAVE) + wave) * L::kWaveColdDwords; // cold columns
    const bool prune = (P.flags & RSRT_FLAG_PRUNE) != 0;
    const bool anyhit_shadow = !(P.flags & RSRT_FLAG_REFERENCE_TRAVERSAL);
    const uint32_t tile_px = P.tile_w * P.tile_h;

#define HOT(f, slot) W[(f) * POOL + (slot)]
#define HOTF(f, slot) as_f(W[(f) * POOL + (slot)])
#define SETH(f, slot, val) W[(f) * POOL + (slot)] = as_u(val)
#define TAG_OF(slot) (W[H_CT * POOL + (slot)] & 7u)
#define SET_TAG(slot, tag) W[H_CT * POOL + (slot)] = (uint32_t)(tag)              /* cursor := root (0) */
#define SET_CUR_TAG(slot, cur, tag) W[H_CT * POOL + (slot)] = ((cur) << 3) | (uint32_t)(tag)
#define COLD(f, slot) G[(f) * POOL + (slot)]
#define COLDF(f, slot) as_f(G[(f) * POOL + (slot)])
#define SETC(f, slot, val) G[(f) * POOL + (slot)] = as_u(val)

    for (uint32_t k = 0; k < L::kSlotsPerLane; k++)
        if (lane + 64u * k < POOL) SET_TAG(lane + 64u * k, TAG_FREE);

    uint32_t chunk_next = 0, chunk_left = 0, chunk_tile_slot0 = 0, chunk_tx0 = 0, chunk_ty0 = 0, chunk_s0 = 0, chunk_p0 = 0;
    bool exhausted = false;
    unsigned long long n_paths = 0, n_ext = 0, n_shadow = 0;
#ifdef RT_INSTRUMENT
    DbgCounters dbg;
    for (int i = 0; i < RT_DBG_N; i++) dbg.c[i] = 0;
#endif

#ifdef RT_INSTRUMENT
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define DBG_STAMP(i) do { unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) dbg.c[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define DBG_STAMP(i) do { } while (0)
#endif
    for (;;) {
        DBG_STAMP(21); // previous stage's tail is charged below; this resets the clock for the census
        // ---------------- 1. census of the stage tags (each lane looks at its kSlotsPerLane slots)
        uint32_t tags[L::kSlotsPerLane];
        uint32_t count[ST_COUNT] = {0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < L::kSlotsPerLane; k++) {
            tags[k] = (lane + 64u * k < POOL) ? TAG_OF(lane + 64u * k) : (uint32_t)TAG_IDLE;
            const uint32_t st = stage_of_tag(tags[k]);
            for (uint32_t s = 0; s < ST_COUNT; s++) count[s] += (uint32_t)__popcll(__ballot(st == s));
        }
        if (exhausted) count[ST_GEN] = 0;
        // ---------------- 2. fullest stage (ties: the later stage, which drains paths)
        uint32_t best = ST_COUNT, best_n = 0;
        for (uint32_t s = 0; s < ST_COUNT; s++)
            if (count[s] >= best_n && count[s] > 0) { best = s; best_n = count[s]; }
        if (best == ST_COUNT) break; // nothing left anywhere
        // ---------------- 3. compaction: dense list of the chosen stage's slots
        uint32_t base = 0;
        for (uint32_t k = 0; k < L::kSlotsPerLane; k++) {
            const bool mine = stage_of_tag(tags[k]) == best;
            const unsigned long long m = __ballot(mine);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (mine && pos < 64u) list[pos] = lane + 64u * k;
            base += (uint32_t)__popcll(m);
        }
        const uint32_t n_run = min(best_n, 64u);
        const bool on = lane < n_run;
        const uint32_t slot = on ? list[lane] : 0u;
        if (lane == 0) { DBG_ADD(best, 1); DBG_ADD(5 + best, n_run); }
        DBG_STAMP(22); // census + compaction

        if (best == ST_GEN) {
            // ---------------- GEN: hand out (pixel, sample) items of the wave's chunk
            uint32_t given = 0; // lanes [given, n_run) still need an item
            while (given < n_run && !exhausted) {
                if (chunk_left == 0u) {
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(P.work_counter, 1u);
                    c = __builtin_amdgcn_readfirstlane((int)c);
                    if (c >= P.n_chunks) { exhausted = true; break; }
                    const uint32_t q = c % P.n_subtiles, cb = c / P.n_subtiles; // sub-tile, then (tile, sample block)
                    const uint32_t j = cb / P.n_sblocks, b = cb % P.n_sblocks;
                    const uint32_t t = j * P.world + P.rank;
                    chunk_tx0 = (t % P.tiles_x) * P.tile_w;
                    chunk_ty0 = (t / P.tiles_x) * P.tile_h;
                    chunk_tile_slot0 = j * tile_px;
                    chunk_s0 = b * P.samples_per_chunk;
                    chunk_p0 = q * P.chunk_px;
                    chunk_next = 0;
                    chunk_left = min(P.samples_per_chunk, P.sample_count - chunk_s0) * P.chunk_px;
                }
                const uint32_t take = min(n_run - given, chunk_left);
                if (on && lane >= given && lane < given + take) {
                    const uint32_t item = chunk_next + (lane - given);
                    const uint32_t ks = item / P.chunk_px, p = chunk_p0 + item % P.chunk_px;
                    const uint32_t px = chunk_tx0 + p % P.tile_w, py = chunk_ty0 + p / P.tile_w;
                    if (px < P.width && py < P.height) {
                        const uint32_t srel = chunk_s0 + ks;
                        PathState ps;
                        start_path(P, px, py, P.sample_begin + srel, ps);
                        SETH(H_OX, slot, ps.o.x); SETH(H_OY, slot, ps.o.y); SETH(H_OZ, slot, ps.o.z);
                        SETH(H_DX, slot, ps.d.x); SETH(H_DY, slot, ps.d.y); SETH(H_DZ, slot, ps.d.z);
                        SETH(H_T, slot, RT_INFINITY);
                        SETC(C_TX, slot, 1.0f); SETC(C_TY, slot, 1.0f); SETC(C_TZ, slot, 1.0f);
                        SETC(C_LX, slot, 0.0f); SETC(C_LY, slot, 0.0f); SETC(C_LZ, slot, 0.0f);
                        SETC(C_LASTPDF, slot, 1.0f);
                        COLD(C_RNG, slot) = ps.rng;
                        COLD(C_BOUNCE, slot) = 0u;
                        COLD(C_OUT, slot) = srel * P.n_slots + chunk_tile_slot0 + p;
                        SET_TAG(slot, TAG_TRACE_EXT);
                        n_paths++;
                    }
                    // an out-of-frame pixel of an edge tile: the slot stays FREE and is offered again
                }
                given += take;
                chunk_next += take;
                chunk_left -= take;
            }
            if (exhausted) { // nothing more to hand out: park every FREE slot
                for (uint32_t k = 0; k < L::kSlotsPerLane; k++)
                    if (lane + 64u * k < POOL && TAG_OF(lane + 64u * k) == TAG_FREE) SET_TAG(lane + 64u * k, TAG_IDLE);
            }
        } else if (best == ST_TRACE) {
            // ---------------- TRACE: cast_ray_bvh for extension and shadow rays together.  The slot's
            // o / d ARE the ray to trace (SHADE already moved a shadow ray's origin to the hit point).
            if (on) {
                const uint32_t ct = HOT(H_CT, slot);
                const bool shadow = (ct & 7u) == TAG_TRACE_SHADOW;
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(H_DX, slot), HOTF(H_DY, slot), HOTF(H_DZ, slot));
                // resume (or start: cur = root, best = INFINITY) the threaded traversal for a bounded number of steps
                Hit h;
                uint32_t cur = ct >> 3;
                h.src = SRC_BVH;
                h.t = HOTF(H_T, slot); h.ref = 0; h.u = h.v = 0.0f; // ref of an earlier call stays in the cold column unless beaten
                const float t_in = h.t;
                if (TRAV == 2) {
                    const V3 inv = v3(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z));
                    // 0 * x is NaN exactly when x is infinite or NaN (an overflowing sum only sends a ray the long way round)
                    const float finite = ((inv.x + inv.y) + inv.z) * 0.0f + ((o.x + o.y) + o.z) * 0.0f;
                    if (finite == 0.0f) {
                        trace_flat(DBG_ARG S, sc, o, d, inv, shadow && anyhit_shadow, h);
                        cur = RT_END;
                    } else {
                        trace_threaded(DBG_ARG S, sc.n_nodes, o, d, prune, shadow && anyhit_shadow, 0xffffffffu, cur, h);
                    }
                } else if (TRAV == 1) {
                    trace_threaded_typed(DBG_ARG S, sc.n_nodes, o, d, prune, shadow && anyhit_shadow, P.trace_budget, cur, h);
                } else {
                    trace_threaded(DBG_ARG S, sc.n_nodes, o, d, prune, shadow && anyhit_shadow, P.trace_budget, cur, h);
                }
                const bool done = cur == RT_END;
                SETH(H_T, slot, h.t);
                if (TRAV != 2 && !shadow && h.t < t_in) COLD(C_REF, slot) = h.ref; // this call found a closer hit
                if (done) {
                    if (shadow) { n_shadow++; SET_TAG(slot, TAG_BSDF); } // H_T < INFINITY <=> occluded
                    else if (TRAV == 2) { n_ext++; SET_CUR_TAG(slot, h.ref, h.did_hit() ? TAG_SHADE : TAG_MISS); } // flat: every ray finishes here and records
                    else { n_ext++; SET_TAG(slot, h.did_hit() ? TAG_SHADE : TAG_MISS); }                            // fit the idle cursor bits: no cold column
                } else {
                    SET_CUR_TAG(slot, cur, ct & 7u);
                }
            }
        } else if (best == ST_MISS) {
            // ---------------- MISS: brute-force fallback of cast_ray (shader.wgsl:583-598), then escape
            if (on) {
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(H_DX, slot), HOTF(H_DY, slot), HOTF(H_DZ, slot));
                Hit h;
                h.t = RT_INFINITY; h.ref = 0; h.src = SRC_BVH; h.u = h.v = 0.0f;
                for (uint32_t i = 0; i < sc.n_spheres; i++) {
                    float u, v;
                    float t = test_record(S, i, SRC_FB_SPHERE, o, d, u, v);
                    if (t >= 0.0f && t < h.t) { h.t = t; h.ref = i; h.src = SRC_FB_SPHERE; }
                }
                for (uint32_t i = 0; i < sc.n_planes; i++) {
                    float u, v;
                    float t = test_record(S, i, SRC_FB_PLANE, o, d, u, v);
                    if (t >= 0.0f && t < h.t) { h.t = t; h.ref = i; h.src = SRC_FB_PLANE; }
                }
                if (h.did_hit()) {
                    SETH(H_T, slot, h.t);
                    if (TRAV == 2) {
                        SET_CUR_TAG(slot, h.ref | (h.src << 6), TAG_SHADE);
                    } else {
                        COLD(C_REF, slot) = h.ref | (h.src << 30);
                        SET_TAG(slot, TAG_SHADE);
                    }
                } else { // escaped: shader.wgsl:1222-1231
                    float u, v;
                    direction_to_equirectangular_uv(d, u, v);
                    const V3 sky = sample_env_bilinear(P.env, u, v);
                    const float pdf = environment_direction_pdf(P.env, d, u, v);
                    const float w = power_heuristic(COLDF(C_LASTPDF, slot), pdf);
                    const V3 T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                    V3 Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                    Lr = Lr + T * sky * w;
                    float *dst = P.sample_buf + (size_t)COLD(C_OUT, slot) * 3u;
                    dst[0] = Lr.x; dst[1] = Lr.y; dst[2] = Lr.z;
                    SET_TAG(slot, TAG_FREE);
                }
            }
        } else if (best == ST_SHADE) {
            // ---------------- SHADE: hit attributes, emission, environment sample (shader.wgsl:1233-1247)
            if (on) {
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(H_DX, slot), HOTF(H_DY, slot), HOTF(H_DZ, slot));
                Hit h;
                h.t = HOTF(H_T, slot);
                if (TRAV == 2) {
                    const uint32_t hr = HOT(H_CT, slot) >> 3;
                    h.ref = hr & 63u; h.src = hr >> 6;
                } else {
                    const uint32_t hr = COLD(C_REF, slot);
                    h.ref = hr & 0x3fffffffu; h.src = hr >> 30;
                }
                hit_barycentrics(S, h, o, d); // not carried through the traversal: the same test gives the same bits
                uint32_t rng = COLD(C_RNG, slot);
                const V3 T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                V3 Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                const Surface surf = resolve_hit(S, h, o, d);
                const BsdfMaterial mat = load_material(S, surf.material_id);
                Lr = Lr + T * mat.emission;
                const EnvironmentSample es = sample_environment(P.env, rng);
                const float cos_nee = fmax_(0.0f, dot(surf.normal, es.direction));
                const bool want_shadow = cos_nee > 0.0f && es.pdf > 0.0f;
                SETC(C_LX, slot, Lr.x); SETC(C_LY, slot, Lr.y); SETC(C_LZ, slot, Lr.z);
                COLD(C_RNG, slot) = rng;
                SETC(C_WX, slot, d.x); SETC(C_WY, slot, d.y); SETC(C_WZ, slot, d.z);
                SETC(C_NX, slot, surf.normal.x); SETC(C_NY, slot, surf.normal.y); SETC(C_NZ, slot, surf.normal.z);
                COLD(C_MAT, slot) = surf.material_id;
                SETC(C_ERX, slot, es.radiance.x); SETC(C_ERY, slot, es.radiance.y); SETC(C_ERZ, slot, es.radiance.z);
                SETC(C_EPDF, slot, es.pdf);
                SETC(C_COS, slot, want_shadow ? cos_nee : 0.0f);
                // the next ray — the shadow ray now, the bounce later — starts at the hit point
                SETH(H_OX, slot, surf.point.x); SETH(H_OY, slot, surf.point.y); SETH(H_OZ, slot, surf.point.z);
                SETH(H_DX, slot, es.direction.x); SETH(H_DY, slot, es.direction.y); SETH(H_DZ, slot, es.direction.z);
                SETH(H_T, slot, RT_INFINITY); // BSDF reads "H_T < INFINITY" as "occluded"
                SET_TAG(slot, want_shadow ? TAG_TRACE_SHADOW : TAG_BSDF);
            }
        } else {
            // ---------------- BSDF: NEE contribution, BSDF sample, throughput (shader.wgsl:1251-1299)
            if (on) {
                const V3 point = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 edir = v3(HOTF(H_DX, slot), HOTF(H_DY, slot), HOTF(H_DZ, slot));
                const bool occluded = HOTF(H_T, slot) < RT_INFINITY;
                const V3 d = v3(COLDF(C_WX, slot), COLDF(C_WY, slot), COLDF(C_WZ, slot));
                const V3 normal = v3(COLDF(C_NX, slot), COLDF(C_NY, slot), COLDF(C_NZ, slot));
                const uint32_t mat_id = COLD(C_MAT, slot);
                V3 T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                V3 Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                const float cos_nee = COLDF(C_COS, slot);
                const V3 erad = v3(COLDF(C_ERX, slot), COLDF(C_ERY, slot), COLDF(C_ERZ, slot));
                const float epdf = COLDF(C_EPDF, slot);
                uint32_t rng = COLD(C_RNG, slot);
                const uint32_t bounce = COLD(C_BOUNCE, slot) + 1u;
                const uint32_t out = COLD(C_OUT, slot);
                const BsdfMaterial mat = load_material(S, mat_id);
                const Frame frame = make_frame(normal); // shader.wgsl:1252 and :1133 build the same frame
                const V3 wo = to_frame_local(frame, -d);
                if (cos_nee > 0.0f && !occluded) { // lit: cos > 0, pdf > 0, not occluded
                    const V3 wi = to_frame_local(frame, edir);
                    V3 scattering;
                    float pdf_bsdf;
                    bsdf_eval_pdf_local(wo, wi, mat, scattering, pdf_bsdf);
                    const float w = power_heuristic(epdf, pdf_bsdf);
                    Lr = Lr + T * w * erad * scattering * cos_nee / epdf;
                }
                const BsdfSample bs = bsdf_sample_in_frame(d, normal, frame, wo, mat, rng);
                bool finished = false;
                if (bs.dir.x == 0.0f && bs.dir.y == 0.0f && bs.dir.z == 0.0f) {
                    Lr = bs.scattering;
                    finished = true;
                } else if (bs.pdf <= 0.0f) {
                    finished = true;
                } else {
                    const float c = fmax_(0.0f, dot(normal, bs.dir));
                    T = T * (bs.scattering * (c / bs.pdf));
                    if (length(T) < 0.001f) finished = true;
                }
                if (bounce >= P.max_bounces) finished = true;
                if (finished) {
                    float *dst = P.sample_buf + (size_t)out * 3u;
                    dst[0] = Lr.x; dst[1] = Lr.y; dst[2] = Lr.z;
                    SET_TAG(slot, TAG_FREE);
                } else {
                    SETC(C_LASTPDF, slot, bs.pdf);
                    SETC(C_TX, slot, T.x); SETC(C_TY, slot, T.y); SETC(C_TZ, slot, T.z);
                    SETC(C_LX, slot, Lr.x); SETC(C_LY, slot, Lr.y); SETC(C_LZ, slot, Lr.z);
                    COLD(C_RNG, slot) = rng;
                    COLD(C_BOUNCE, slot) = bounce;
                    SETH(H_DX, slot, bs.dir.x); SETH(H_DY, slot, bs.dir.y); SETH(H_DZ, slot, bs.dir.z); // origin stays the hit point
                    SETH(H_T, slot, RT_INFINITY);
                    SET_TAG(slot, TAG_TRACE_EXT);
                }
                (void)point;
            }
        }
        DBG_STAMP(16 + best); // the stage just run
    }
#undef HOT
#undef HOTF
#undef SETH
#undef TAG_OF
#undef SET_TAG
#undef SET_CUR_TAG
#undef COLD
#undef COLDF
#undef SETC

    for (int off = 32; off > 0; off >>= 1) {
        n_paths += __shfl_down(n_paths, off);
        n_ext += __shfl_down(n_ext, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if (lane == 0) {
        atomicAdd(&P.stats[0], n_paths);
        atomicAdd(&P.stats[1], n_ext);
        atomicAdd(&P.stats[2], n_shadow);
    }
#ifdef RT_INSTRUMENT
    for (int i = 0; i < RT_DBG_N; i++) {
        unsigned long long v = dbg.c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&P.stats[3 + i], v);
    }
#endif
}
