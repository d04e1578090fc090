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
// Where the state lives (measured: run time scales ~linearly with resident waves up to 3 per SIMD, and LDS
// is what limits them): the HOT columns — what a TRACE invocation needs: the vertex the rays start from,
// the extension direction, the shadow direction, best t, and one word of tag / flags / traversal cursor —
// are in LDS, 11 dwords per slot; the COLD columns — throughput, radiance, the pending NEE term, last pdf ... —
// only read / written by the shading stages, are in a global-memory arena (14 dwords per slot for the tree
// walks; 11 for the flat traversal, whose RNG word, bounce count and hit record ride in hot cells that are
// idle at the time, + 2 that only a ray cut short by the triangle-loop vote touches; one contiguous column per
// field and wave, so a stage's accesses coalesce).
//
// Stages: GEN    take the next (pixel, sample) of the wave's chunk, build the camera ray — and trace it (its first
//                round, for the walks) then and there: the lanes are all busy and the rays of a tile coherent
//         TRACE  one ray of the slot: its NEE shadow ray first (any hit), then its extension ray (closest hit)
//         MISS   the extension ray missed the BVH: cast_ray's brute-force fallback, then the escape
//         SHADE  everything the shader does at a hit (shader.wgsl:1233-1299): emission, environment sample,
//                the NEE term, BSDF sample, throughput, termination — and first of all the NEE term of the
//                PREVIOUS vertex, now that its shadow ray has been traced
//         FINISH a path that ended at its last vertex but still had a shadow ray out: add the NEE term, store
// "Deferred NEE": the shader adds a vertex's NEE term right after its shadow ray and before sampling the
// BSDF.  Here SHADE computes the term and the BSDF sample in one go, parks the term (3 cold dwords),
// and both rays of the vertex are traced from hot state alone; the term is added — or dropped, if
// occluded — when the path is next touched (SHADE, MISS or FINISH), i.e. before anything else is added to
// the radiance, so every sum is formed in the shader's order.  One shading stage per bounce instead of
// two: a third fewer stage switches, and the state crosses the memory system once per bounce.
#pragma once
#include "rt_device.h"
#include "rt_coop.h"

enum HotField { H_OX, H_OY, H_OZ, H_EX, H_EY, H_EZ, H_SX, H_SY, H_SZ, H_T, H_CT, H_COUNT };
// H_CT: stage tag (3 bits) | flags (4 bits) | cursor of the traversal in progress, or the record of the hit it found (25 bits)
enum CtBits : uint32_t {
    F_SHADOW = 8u,    // a shadow ray (direction S) is still to be traced
    F_EXT = 16u,      // an extension ray (direction E) is still to be traced
    F_NEE = 32u,      // C_NEE* holds a term that is added unless F_OCCLUDED
    F_OCCLUDED = 64u, // result of the shadow ray
    CT_FLAGS = 120u,
    CT_SHIFT = 7u
};
enum ColdField {
    C_TX, C_TY, C_TZ, C_LX, C_LY, C_LZ, C_NEEX, C_NEEY, C_NEEZ, C_LASTPDF, C_OUT,
    C_RNG, // (the flat traversal keeps it in a hot cell)
    C_BOUNCE,
    C_REF, // tree-walk traversals only: best hit of the extension ray, record | source << 30
    C_COUNT,
    // the flat traversal stops early: its bounce count and hit record ride in the idle cursor bits of H_CT; two columns
    // hold the untested triangles of a ray whose triangle loop was cut short (only such rays touch them)
    C_REM_LO = C_RNG, C_REM_HI = C_BOUNCE,
    C_COUNT_FLAT = C_REF
};
#define RT_FLAT_BOUNCE_SHIFT 8u     // flat traversal, H_CT payload: hit record (6-bit index | 2-bit source) | bounce << 8 | resumed << 24
#define RT_FLAT_BOUNCE_BITS 0xffff00u
#define RT_FLAT_RESUMED (1u << 24)
#define RT_FLAT_MAX_BOUNCES 0xffffu // 16 bits of the 25; rsrt_render picks a tree-walk kernel beyond that
// the wide walk parks an unfinished ray's stack in columns of its own, and a deep tree's walk keeps the bottom of its stack there (RT_WSTATE_WORDS
// columns, rt_device.h WalkState); only such rays touch them
#define C_WIDE_STATE ((uint32_t)C_COUNT)
#define C_COUNT_WIDE ((uint32_t)C_COUNT + 2u + RT_WSTACK)   /* TRAV 4: next node, pending group, the register stack */
#define C_COUNT_WIDE_DEEP ((uint32_t)C_COUNT + RT_WSTATE_WORDS) /* TRAV 5: + overflow count and overflow words */
__host__ __device__ constexpr uint32_t pool_cold_columns(int trav)
{
    // (6, the cooperative walk: no C_REF — the hit's record waits in the slot's hot result cell — and nothing parked)
    return trav == 2 ? (uint32_t)C_COUNT_FLAT : (trav == 4 ? C_COUNT_WIDE : (trav == 5 ? C_COUNT_WIDE_DEEP : (trav == 6 ? (uint32_t)C_REF : (uint32_t)C_COUNT)));
}
enum PoolTag { TAG_FREE = 0, TAG_TRACE = 1, TAG_MISS = 2, TAG_SHADE = 3, TAG_FINISH = 4, TAG_IDLE = 6 };
enum PoolStage { ST_GEN = 0, ST_TRACE = 1, ST_MISS = 2, ST_SHADE = 3, ST_FINISH = 4, ST_COUNT = 5 };

// What a wave keeps in LDS: its hot columns (TRAV 6: twelve, rt_coop.h CoopCols — the result cell is two dwords wide), the compaction list
// (TRAV 6: every waiting ray is traced at once, so the list holds a whole pool — and the walk's leaf stack uses the same words afterwards) and,
// TRAV 6, the walk's node stack; and in the arena: its cold columns and, TRAV 6, the node stack's overflow block
__host__ __device__ constexpr uint32_t pool_hot_columns(int trav) { return trav == 6 ? 12u : (uint32_t)H_COUNT; }
__host__ __device__ constexpr uint32_t pool_list_dwords(int trav) { return trav == 6 ? RT_COOP_LCAP : 64u; }
__host__ __device__ constexpr uint32_t pool_wave_lds_dwords(int trav, uint32_t pool) { return pool_hot_columns(trav) * pool + pool_list_dwords(trav) + (trav == 6 ? RT_COOP_NCAP + RT_COOP_MAP : 0u); }
__host__ __device__ constexpr uint32_t pool_wave_cold_dwords(int trav, uint32_t pool) { return pool_cold_columns(trav) * pool + (trav == 6 ? RT_COOP_GCAP : 0u); }
template <uint32_t POOL, int TRAV>
struct PoolLayout {
    static constexpr uint32_t kSlotsPerLane = (POOL + 63u) / 64u;
    static constexpr uint32_t kHotDwords = pool_hot_columns(TRAV) * POOL; // in LDS: every hot column
    static constexpr uint32_t kListDwords = pool_list_dwords(TRAV);
    static constexpr uint32_t kWaveLdsDwords = pool_wave_lds_dwords(TRAV, POOL);
    static constexpr uint32_t kColdColumns = pool_cold_columns(TRAV);
    static constexpr uint32_t kWaveColdDwords = pool_wave_cold_dwords(TRAV, POOL);
    static constexpr uint32_t kCtColumn = pool_hot_columns(TRAV) - 1u; // the tag word is the last hot column
};

RT_DEV uint32_t stage_of_tag(uint32_t tag) { return tag; } // FREE -> GEN, TRACE, MISS, SHADE, FINISH; IDLE -> none

// The lanes of a wave hand data to each other through memory: the compaction list, the hot columns and the stage tag
// (LDS) and the cold columns (global memory) are written by the lane that runs a stage and read by whichever lane
// picks the slot up next.  DS and VMEM operations of one wave execute in order, so nothing has to be waited for,
// but the memory model still wants the hand-over spelled out: a wavefront-scope release / acquire pair around a
// wave barrier.  It emits no instruction on gfx950; it keeps the compiler from moving or forwarding those accesses.
#define RT_WAVE_HANDOVER()                                        \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

// A finished path's radiance goes to the sample buffer once and is read once, by rt_resolve_kernel, long after: a
// write-once stream that has no business occupying L2 lines the path-state arena wants — one non-temporal 12-byte store
// (three dword stores, a plain 12-byte store and an sc1 write-through were measured against it: profiles/r02_l2_sweep.txt)
typedef float rt_f3v __attribute__((ext_vector_type(3)));
typedef rt_f3v rt_f3v_a4 __attribute__((aligned(4)));
RT_DEV void store_sample(float *dst, V3 L) { __builtin_nontemporal_store(rt_f3v{L.x, L.y, L.z}, reinterpret_cast<rt_f3v_a4 *>(dst)); }

#ifndef RT_POOL_WAVES_PER_SIMD
#define RT_POOL_WAVES_PER_SIMD 4
#endif
// TRAV: which traversal TRACE runs — 0 trace_threaded (any BVH), 1 trace_threaded_typed (leaves of <= 8
// primitives), 2 trace_flat (<= 64 primitive records, nested boxes; rays with a non-finite 1/d fall back to 0),
// 3 trace_preorder (leaves of <= 8 primitives; fixed order, ties by tabulated visiting rank), 4 trace_wide (4-wide nodes
// collapsed from the binary tree, a quarter of the dependent fetches; rays with a non-finite 1/d fall back to 3), 5 trace_wide for a
// tree deeper than the walk's register stack (the bottom of the stack overflows into the slot's arena columns), 6 the cooperative wide walk
// (rt_coop.h: the wave traces all waiting rays together, as (ray, node) and (ray, leaf) items on two LDS stacks; POOL <= 128)
// SV: where the scene is read from — 0 global memory, 1 the whole image in LDS (256-thread workgroups, several per
// CU), 2 nodes + escape links in LDS (SceneViewHybrid; BLOCK = 1024: one workgroup per CU shares the copy)
template <int SV> struct PoolView;
template <> struct PoolView<0> { typedef SceneView<false> type; static __device__ __forceinline__ type make(const DevScene &sc) { return make_view<false>(sc); } };
template <> struct PoolView<1> { typedef SceneView<true> type; static __device__ __forceinline__ type make(const DevScene &sc) { return make_view<true>(sc); } };
template <> struct PoolView<2> { typedef SceneViewHybrid type; static __device__ __forceinline__ type make(const DevScene &sc) { return make_view_hybrid(sc); } };

template <int SV, uint32_t BLOCK, uint32_t POOL, int TRAV>
__global__ __launch_bounds__(BLOCK, RT_POOL_WAVES_PER_SIMD) void rt_render_pool_kernel(RenderParams P)
{
    typedef PoolLayout<POOL, TRAV> L;
    constexpr bool kCoop = TRAV == 6; // the cooperative wide walk (rt_coop.h): TRACE takes every waiting ray of the pool at once, both rays of a vertex in one call
    constexpr bool kBounceInCt = TRAV == 2; // no C_BOUNCE / C_REF column
    // Flat traversal: a ray finishes in one TRACE call — or is cut short by the triangle-loop vote with its best t parked where the
    // result would go — so "best t so far" is INFINITY at every fresh start and the H_T cell is
    // only needed for the RESULT of the extension ray — which fits the shadow direction's first cell, dead by then (the
    // shadow ray is traced first).  H_T then carries the RNG word instead of a cold column: SHADE's first dependent memory
    // access, the alias-table gather, can leave with the cold loads instead of a memory round trip after them.
    constexpr bool kRngHot = TRAV == 2;
    constexpr uint32_t kTCell = kRngHot ? (uint32_t)H_SX : (uint32_t)H_T; // where the extension ray's t waits for SHADE / MISS
    constexpr bool kFlatVote = kRngHot && kBounceInCt; // a TRACE call may return a flat traversal unfinished
    constexpr bool kPackedMiss = TRAV == 2 || TRAV == 6; // MISS reads an escaping ray's pmf from the texels' alpha (rt_device.h)
    constexpr bool kGenTrace = TRAV >= 2 && !kCoop; // GEN traces the camera ray it has built (the near-first tree walks, kept for RSRT_FLAG_PRUNE, would spill)
    const DevScene &sc = P.scene;
    if (SV != 0) stage_scene_lds(sc);
    const typename PoolView<SV>::type S = PoolView<SV>::make(sc);
    const uint32_t lane0 = threadIdx.x & (RT_WAVE - 1);
    // (readfirstlane: the compiler cannot know that threadIdx.x / 64 is wave-uniform; told so, the wave's LDS and arena bases live in scalar
    // registers and a column access is base + 32-bit lane offset instead of a 64-bit address per lane)
    // (only the walk kernels — the flat kernel, which already holds its cull boxes in scalar registers, runs out of them: +1-1.5 %)
    constexpr bool kScalarWave = SV == 2 || TRAV == 6;
    const uint32_t wave = kScalarWave ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / RT_WAVE)) : threadIdx.x / RT_WAVE;
    uint32_t *const lds32 = reinterpret_cast<uint32_t *>(rt_smem + sc.lds_float4s);
    uint32_t *const W = lds32 + wave * L::kWaveLdsDwords; // this wave's hot columns
    uint32_t *const WCT = W + L::kCtColumn * POOL; // the tag column (the census reads all of it every trip)
    uint32_t *const list = W + L::kHotDwords;
    uint32_t *const G = P.cold_state + (size_t)(blockIdx.x * (BLOCK / RT_WAVE) + wave) * L::kWaveColdDwords; // cold columns
    const bool prune = (P.flags & RSRT_FLAG_PRUNE) != 0;
    const bool anyhit_shadow = !(P.flags & RSRT_FLAG_REFERENCE_TRAVERSAL);
    const uint32_t tile_px = P.tile_w * P.tile_h;

#define HOT(f, slot) W[(f) * POOL + (slot)]           /* a hot column other than the tag, as u32 */
#define HOTF(f, slot) as_f(HOT(f, slot))
#define SETHU(f, slot, val) W[(f) * POOL + (slot)] = (val)
#define SETH(f, slot, val) SETHU(f, slot, as_u(val))
#define CT_OF(slot) WCT[(slot)]
#define TAG_OF(slot) (WCT[(slot)] & 7u)
#define SET_TAG(slot, tag) WCT[(slot)] = (uint32_t)(tag)              /* no flags, cursor := root (0) */
#define SET_CT(slot, payload, flags, tag) WCT[(slot)] = ((payload) << CT_SHIFT) | (flags) | (uint32_t)(tag)
// (TRAV 6: the extension ray's result cell, rt_coop.h CoopCols::BEST)
#define BEST_REF(slot) W[9u * POOL + 2u * (slot)]
#define BEST_T(slot) W[9u * POOL + 2u * (slot) + 1u]
#define COLD(f, slot) G[(f) * POOL + (slot)]
#define COLDF(f, slot) as_f(G[(f) * POOL + (slot)])
#define SETC(f, slot, val) G[(f) * POOL + (slot)] = as_u(val)

    for (uint32_t k = 0; k < L::kSlotsPerLane; k++)
        if (lane0 + 64u * k < POOL) SET_TAG(lane0 + 64u * k, TAG_FREE);
    RT_WAVE_HANDOVER();

    uint32_t chunk_next = 0, chunk_left = 0, chunk_tile_slot0 = 0, chunk_tx0 = 0, chunk_ty0 = 0, chunk_s0 = 0, chunk_p0 = 0;
    bool exhausted = false;
    unsigned long long n_paths = 0, n_ext = 0, n_shadow = 0;
    uint32_t n_work = 0; // box steps + primitive tests of the tree walks (per lane; widened in the final reduction)
#ifdef RT_INSTRUMENT
    DbgCounters dbg;
    for (int i = 0; i < RT_DBG_N; i++) dbg.c[i] = 0;
#endif

#ifdef RT_INSTRUMENT
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define DBG_STAMP(i) do { unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane0 == 0) dbg.c[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define DBG_STAMP(i) do { } while (0)
#endif
#if defined(RT_INSTRUMENT) && defined(RT_SHADE_PROFILE)
#define SHADE_STAMP(i) DBG_STAMP(i)
#else
#define SHADE_STAMP(i) do { } while (0)
#endif
    // What a slot is left as once its ray has been traced as far as this call goes (cur == RT_END: to the end): the end of TRACE's body.
    auto ray_end = [&](const uint32_t slot, const uint32_t ct, const Hit &h, const uint32_t cur, const float t_in, const unsigned long long flat_rem) __attribute__((always_inline)) {
        const bool shadow = (ct & F_SHADOW) != 0u;
        const uint32_t bounce_bits = kBounceInCt ? ((ct >> CT_SHIFT) & RT_FLAT_BOUNCE_BITS) : 0u;
        const bool found = TRAV >= 3 ? (h.ref != RT_REF_UNKNOWN) : (h.t < t_in); // this call found a closer (or earlier-ranked) hit
        // (a shadow ray that is cut short with a hit in hand is done whatever the flags say: only did_hit is read)
        const bool done = cur == RT_END || (kFlatVote && shadow && h.t < RT_INFINITY);
        if (!done && kFlatVote) { // flat traversal cut short: the triangles left, best t and record (a shadow ray has none: any hit ends it)
            COLD(C_REM_LO, slot) = (uint32_t)flat_rem;
            COLD(C_REM_HI, slot) = (uint32_t)(flat_rem >> 32);
            if (!shadow) SETH(kTCell, slot, h.t);
            SET_CT(slot, RT_FLAT_RESUMED | bounce_bits | (shadow ? 0u : h.ref), ct & CT_FLAGS, TAG_TRACE);
        } else if (!done) { // to be resumed: best t and cursor
            if (!kRngHot) SETH(H_T, slot, h.t);
            if (TRAV != 2 && !shadow && found) COLD(C_REF, slot) = h.ref;
            SET_CT(slot, cur, ct & CT_FLAGS, TAG_TRACE);
        } else if (shadow) {
            n_shadow++;
            if (!kRngHot) SETH(H_T, slot, RT_INFINITY); // the extension ray starts fresh
            const uint32_t fl = (ct & (F_EXT | F_NEE)) | (h.t < RT_INFINITY ? (uint32_t)F_OCCLUDED : 0u);
            SET_CT(slot, bounce_bits, fl, (ct & F_EXT) ? TAG_TRACE : TAG_FINISH);
        } else {
            n_ext++;
            SETH(kTCell, slot, h.t);
            if (TRAV != 2 && found) COLD(C_REF, slot) = h.ref;
            // the flat traversal's records fit the idle cursor bits: no cold column
            SET_CT(slot, (TRAV == 2 ? h.ref : 0u) | bounce_bits, ct & (F_NEE | F_OCCLUDED), h.did_hit() ? TAG_SHADE : TAG_MISS);
        }
    };
    // One ray of `slot` (tag / flags word `ct`) from o along d, traced or resumed, and what the slot is left as: TRACE's body, as a
    // lambda because GEN runs it too (a new path's camera ray is traced by the lanes that have just built it).
    auto trace_slot = [&](const uint32_t slot, const uint32_t ct, const V3 o, const V3 d, const bool coherent) __attribute__((always_inline)) {
        const bool shadow = (ct & F_SHADOW) != 0u;
        // resume (or start: cur = root, best = INFINITY) the traversal
        Hit h;
        uint32_t cur = kBounceInCt ? (kFlatVote ? (ct >> (CT_SHIFT + 24u)) : 0u) : (ct >> CT_SHIFT); // (flat: the bits carry the bounce count)
        h.src = SRC_BVH;
        h.t = kRngHot ? RT_INFINITY : HOTF(H_T, slot); h.u = h.v = 0.0f;
        unsigned long long flat_rem = 0ull;
        if (kFlatVote && cur != 0u) { // cut short by the vote of an earlier call: the untested triangles, the best hit so far
            flat_rem = ((unsigned long long)COLD(C_REM_HI, slot) << 32) | COLD(C_REM_LO, slot);
            if (!shadow) h.t = HOTF(kTCell, slot);
        }
        // the record of an earlier call's best hit stays in the cold column unless beaten; the fixed-order walk, where
        // an equal t can still replace it, fetches it from there if (and only if) such a tie comes up
        h.ref = (TRAV >= 3 && !shadow) ? RT_REF_UNKNOWN : ((kFlatVote && cur != 0u) ? ((ct >> CT_SHIFT) & 63u) : 0u);
        const float t_in = h.t;
        trace_dispatch<TRAV>(DBG_ARG S, sc, o, d, prune, shadow && anyhit_shadow, P.trace_budget, TRAV == 2 ? (kFlatVote ? P.flat_quorum : 0u) : P.descend_quorum,
                             cur, h, &COLD(C_REF, slot), n_work, flat_rem, TRAV >= 4 ? &G[C_WIDE_STATE * POOL + slot] : nullptr, POOL, P.stop_quorum, coherent);
        ray_end(slot, ct, h, cur, t_in, flat_rem);
    };
    for (;;) {
        // (TRAV 6: the lane index is made opaque once per trip of the scheduler, so that what is derived from it — a dozen LDS and arena
        // addresses per stage — is formed where it is used instead of being hoisted out of this loop into registers that then live through
        // every stage: the kernel stays within 128 registers without scratch memory, tests/test_code_object.py)
        uint32_t lane = lane0;
        if (kCoop) asm volatile("" : "+v"(lane));
        RT_MARK(0);
        DBG_STAMP(21); // previous stage's tail is charged below; this resets the clock for the census
        // ---------------- 1. census of the stage tags (each lane looks at its kSlotsPerLane slots)
        uint32_t tags[L::kSlotsPerLane];
        uint32_t count[ST_COUNT] = {};
        for (uint32_t k = 0; k < L::kSlotsPerLane; k++) {
            tags[k] = (lane + 64u * k < POOL) ? TAG_OF(lane + 64u * k) : (uint32_t)TAG_IDLE;
            const uint32_t st = stage_of_tag(tags[k]);
            for (uint32_t s = 0; s < ST_COUNT; s++) count[s] += (uint32_t)__popcll(__ballot(st == s));
        }
        if (exhausted) count[ST_GEN] = 0;
        // ---------------- 2. which stage: the fullest (ties: the later stage, which drains paths)
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
            if (mine && pos < L::kListDwords) list[pos] = lane + 64u * k;
            base += (uint32_t)__popcll(m);
        }
        RT_WAVE_HANDOVER(); // list[] is read by other lanes than wrote it
        const uint32_t n_run = min(best_n, 64u);
        const bool on = lane < n_run;
        const uint32_t slot = on ? list[lane] : 0u;
        const uint32_t dbg_stage = best;
        (void)dbg_stage;
        if (lane == 0) { DBG_ADD(dbg_stage, 1); DBG_ADD(5 + dbg_stage, (kCoop && best == ST_TRACE) ? best_n : n_run); }
        DBG_STAMP(22); // census + compaction
        RT_MARK(1);

        if (best == ST_GEN) {
            // ---------------- GEN: hand out (pixel, sample) items of the wave's chunk
            RT_MARK(2);
            uint32_t given = 0; // lanes [given, n_run) still need an item
            bool started = false; // this lane has built a camera ray
            V3 cam_o = v3(0.0f, 0.0f, 0.0f), cam_d = v3(0.0f, 0.0f, 0.0f);
            while (given < n_run && !exhausted) {
                if (chunk_left == 0u) {
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(P.work_counter, 1u);
                    c = __builtin_amdgcn_readfirstlane((int)c);
                    if (c >= P.n_chunks) { exhausted = true; break; }
                    const uint32_t q = c % P.n_subtiles, cb = c / P.n_subtiles; // sub-tile, then (tile, sample block)
                    const uint32_t j = cb / P.n_sblocks, b = cb % P.n_sblocks;
                    uint32_t ttx, tty;
                    if (!owned_tile(j, P.rank, P.world, P.skew, P.tiles_x, P.tiles_per_row, ttx, tty)) continue; // padding slot: next chunk
                    chunk_tx0 = ttx * P.tile_w;
                    chunk_ty0 = tty * P.tile_h;
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
                        SETH(H_EX, slot, ps.d.x); SETH(H_EY, slot, ps.d.y); SETH(H_EZ, slot, ps.d.z);
                        if (kRngHot) SETHU(H_T, slot, ps.rng);
                        else if (!kCoop) SETH(H_T, slot, RT_INFINITY);
                        SETC(C_TX, slot, 1.0f); SETC(C_TY, slot, 1.0f); SETC(C_TZ, slot, 1.0f);
                        SETC(C_LX, slot, 0.0f); SETC(C_LY, slot, 0.0f); SETC(C_LZ, slot, 0.0f);
                        SETC(C_LASTPDF, slot, 1.0f);
                        if (!kRngHot) COLD(C_RNG, slot) = ps.rng;
                        if (!kBounceInCt) COLD(C_BOUNCE, slot) = 0u;
                        COLD(C_OUT, slot) = srel * P.n_slots + chunk_tile_slot0 + p;
                        if (kGenTrace) { started = true; cam_o = ps.o; cam_d = ps.d; }
                        else SET_CT(slot, 0u, F_EXT, TAG_TRACE);
                        n_paths++;
                    }
                    // an out-of-frame pixel of an edge tile: the slot stays FREE and is offered again
                }
                given += take;
                chunk_next += take;
                chunk_left -= take;
            }
            // the camera rays are traced here and now: the lanes are all busy, the rays of a tile are coherent, and the
            // path's first trip through the scheduler (a sixth of all its stage switches) is saved
            RT_MARK(3);
            if (kGenTrace && started) trace_slot(slot, (uint32_t)F_EXT | (uint32_t)TAG_TRACE, cam_o, cam_d, true);
            RT_MARK(14);
            if (exhausted) { // nothing more to hand out: park every FREE slot
                for (uint32_t k = 0; k < L::kSlotsPerLane; k++)
                    if (lane + 64u * k < POOL && TAG_OF(lane + 64u * k) == TAG_FREE) SET_TAG(lane + 64u * k, TAG_IDLE);
            }
        } else if (kCoop && best == ST_TRACE) {
            // ---------------- TRACE, cooperative wide walk: EVERY slot that waits in TRACE (the list holds them all), its shadow ray and its
            // extension ray at once, as items on the wave's two stacks (rt_coop.h); when the stacks are empty every ray is done
            if constexpr (kCoop) {
                CoopStacks cs;
                cs.ns = list + L::kListDwords; cs.ls = list; cs.gs = G + L::kColdColumns * POOL;
                cs.map = reinterpret_cast<uint16_t *>(cs.ns + RT_COOP_NCAP);
                cs.ns_h = cs.ns_n = cs.ls_n = cs.gs_n = 0u;
                cs.lds_cap = P.coop_lds_cap; cs.lifo_at = P.coop_lifo_at; cs.narrow_at = P.coop_narrow_at;
                for (uint32_t i0 = 0; i0 < best_n; i0 += 64u) { // (the list is read to the end before the first leaf item lands in the same words)
                    const bool valid = i0 + lane < best_n;
                    const uint32_t s = list[valid ? i0 + lane : 0u];
                    coop_push_rays<POOL>(W, cs, valid, s, CT_OF(s), (uint32_t)F_EXT, (uint32_t)F_SHADOW);
                }
                coop_trace<POOL>(DBG_ARG S, W, cs, anyhit_shadow, lane, n_work);
                RT_WAVE_HANDOVER();
                // what each slot is left as (the census's tags are still good: nothing else ran).  First the rays the walk handed to the exact
                // fixed-order walk: a non-finite 1/d, two records at the same closest t (next to none)
                uint32_t my_slot[L::kSlotsPerLane];
                bool mine[L::kSlotsPerLane];
                for (uint32_t k = 0; k < L::kSlotsPerLane; k++) { my_slot[k] = lane + 64u * k; mine[k] = tags[k] == (uint32_t)TAG_TRACE; }
                coop_slow_rays<POOL, L::kSlotsPerLane>(DBG_ARG S, sc, W, mine, my_slot, anyhit_shadow, n_work);
                for (uint32_t k = 0; k < L::kSlotsPerLane; k++) {
                    if (!mine[k]) continue;
                    const uint32_t s = my_slot[k], ct = CT_OF(s);
                    const uint32_t fl = ct & (uint32_t)(F_NEE | F_OCCLUDED);
                    if (ct & F_SHADOW) n_shadow++;
                    if (ct & F_EXT) {
                        n_ext++;
                        SET_CT(s, 0u, fl, as_f(BEST_T(s)) < RT_INFINITY ? TAG_SHADE : TAG_MISS);
                    } else {
                        SET_CT(s, 0u, fl, TAG_FINISH);
                    }
                }
            }
        } else if (best == ST_TRACE) {
            // ---------------- TRACE: one ray of the slot from its vertex O — the shadow ray (direction S, any
            // hit) while one is pending, else the extension ray (direction E, closest hit): cast_ray_bvh
            RT_MARK(12);
            if (on) {
                const uint32_t ct = CT_OF(slot);
                const uint32_t dcol = (ct & F_SHADOW) ? (uint32_t)H_SX : (uint32_t)H_EX;
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(dcol, slot), HOTF(dcol + 1u, slot), HOTF(dcol + 2u, slot));
                trace_slot(slot, ct, o, d, false);
            }
        } else if (best == ST_MISS) {
            // ---------------- MISS: brute-force fallback of cast_ray (shader.wgsl:583-598), then escape
            RT_MARK2(0);
            if (on) {
                const uint32_t ct = CT_OF(slot);
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(H_EX, slot), HOTF(H_EY, slot), HOTF(H_EZ, slot));
                // Memory first, as in SHADE: what the escape needs — four texels, the alias entry of the texel under the ray, the
                // cold columns — depends on the ray alone and is asked for before the fallback loops, not after them.  (A ray
                // that does hit a fallback primitive has asked in vain: no scene the reference ships has one outside its BVH.)
                float env_u, env_v;
                direction_to_equirectangular_uv(d, env_u, env_v);
                // (flat kernel and cooperative walk: the texel's alpha carries its pmf, no alias-table gather; the per-lane walks, a register
                // short of their 128, keep the gather — tests/test_code_object.py)
                const EnvBilinearFetch sky_fetch = kPackedMiss ? sample_env_bilinear_begin_pmf(P.env, env_u, env_v) : sample_env_bilinear_begin(P.env, env_u, env_v);
                float sky_pmf = 0.0f;
                if (!kPackedMiss) sky_pmf = environment_direction_pmf(P.env, env_u, env_v);
                const float last_pdf = COLDF(C_LASTPDF, slot);
                const V3 T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                V3 Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                const V3 nee_prev = v3(COLDF(C_NEEX, slot), COLDF(C_NEEY, slot), COLDF(C_NEEZ, slot));
                const uint32_t out_slot = COLD(C_OUT, slot);
                Hit h;
                h.t = RT_INFINITY; h.ref = 0; h.src = SRC_BVH; h.u = h.v = 0.0f;
                for (uint32_t i = 0; i < sc.n_spheres; i++) {
                    RT_MARK2(1);
                    float u, v;
                    float t = test_record(S, i, SRC_FB_SPHERE, o, d, u, v);
                    if (t >= 0.0f && t < h.t) { h.t = t; h.ref = i; h.src = SRC_FB_SPHERE; }
                }
                for (uint32_t i = 0; i < sc.n_planes; i++) {
                    RT_MARK2(2);
                    float u, v;
                    float t = test_record(S, i, SRC_FB_PLANE, o, d, u, v);
                    if (t >= 0.0f && t < h.t) { h.t = t; h.ref = i; h.src = SRC_FB_PLANE; }
                }
                RT_MARK2(3);
                if (h.did_hit()) { // SHADE takes it from here (and settles the pending NEE term)
                    if (kCoop) BEST_T(slot) = as_u(h.t); else SETH(kTCell, slot, h.t);
                    if (kCoop) {
                        BEST_REF(slot) = h.ref | (h.src << 30);
                        SET_CT(slot, 0u, ct & (F_NEE | F_OCCLUDED), TAG_SHADE);
                    } else if (TRAV == 2) {
                        SET_CT(slot, h.ref | (h.src << 6) | (kBounceInCt ? ((ct >> CT_SHIFT) & RT_FLAT_BOUNCE_BITS) : 0u), ct & (F_NEE | F_OCCLUDED), TAG_SHADE);
                    } else {
                        COLD(C_REF, slot) = h.ref | (h.src << 30);
                        SET_CT(slot, 0u, ct & (F_NEE | F_OCCLUDED), TAG_SHADE);
                    }
                } else { // escaped: shader.wgsl:1222-1231
                    const V3 sky = sample_env_bilinear_finish(sky_fetch);
                    if (kPackedMiss) sky_pmf = env_bilinear_pmf(P.env, sky_fetch, env_u, env_v);
                    const float pdf = sky_pmf / environment_pixel_solid_angle(env_v, P.env);
                    const float w = power_heuristic(last_pdf, pdf);
                    if ((ct & (F_NEE | F_OCCLUDED)) == F_NEE) Lr = Lr + nee_prev; // previous vertex, lit
                    Lr = Lr + T * sky * w;
                    store_sample(P.sample_buf + (size_t)out_slot * 3u, Lr);
                    SET_TAG(slot, TAG_FREE);
                }
            }
        } else if (best == ST_SHADE) {
            // ---------------- SHADE: one whole iteration of trace_ray's loop body at a hit (shader.wgsl:1233-1299)
            RT_MARK2(4);
            if (on) {
                const uint32_t ct = CT_OF(slot);
                const V3 o = v3(HOTF(H_OX, slot), HOTF(H_OY, slot), HOTF(H_OZ, slot));
                const V3 d = v3(HOTF(H_EX, slot), HOTF(H_EY, slot), HOTF(H_EZ, slot));
                Hit h;
                h.t = kCoop ? as_f(BEST_T(slot)) : HOTF(kTCell, slot);
                if (TRAV == 2) {
                    const uint32_t hr = (ct >> CT_SHIFT) & 0xffu;
                    h.ref = hr & 63u; h.src = hr >> 6;
                } else {
                    const uint32_t hr = kCoop ? BEST_REF(slot) : COLD(C_REF, slot);
                    h.ref = hr & 0x3fffffffu; h.src = hr >> 30;
                }
                DBG_ADD(27, (as_u(hit_record(S, h, 0).w) & 3u) == PRIM_TRIANGLE ? 1 : 0); DBG_ADD(31, (as_u(hit_record(S, h, 0).w) & 3u) == PRIM_SPHERE ? 1 : 0);
                uint32_t rng, bounce;
                V3 T, Lr;
                Surface surf;
                BsdfMaterial mat;
                EnvironmentSample es;
                if constexpr (kRngHot) {
                    // Memory first: the alias-table gather of this vertex's environment sample (its address needs the RNG word
                    // only, and that is an LDS read away) and the cold columns are requested before anything is computed,
                    // the pending NEE term unconditionally — a load under a branch would have to be waited for, with
                    // everything else in flight, right here.  (Not for the walks: their RNG word is one of the cold loads,
                    // and the longer live ranges push the 128-register kernel into scratch — measured 4 % slower.)
                    rng = HOT(H_T, slot);
                    const EnvironmentPick pick = sample_environment_begin(P.env, rng);
                    T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                    Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                    const V3 nee_prev = v3(COLDF(C_NEEX, slot), COLDF(C_NEEY, slot), COLDF(C_NEEZ, slot));
                    bounce = (kBounceInCt ? (ct >> (CT_SHIFT + RT_FLAT_BOUNCE_SHIFT)) : COLD(C_BOUNCE, slot)) + 1u;
                    SHADE_STAMP(10);
                    hit_barycentrics(S, h, o, d); // not carried through the traversal: the same test gives the same bits
                    surf = resolve_hit(S, h, o, d);
                    mat = load_material(S, surf.material_id);
                    SHADE_STAMP(11);
                    es = sample_environment_finish<true>(P.env, rng, pick);
                    SHADE_STAMP(12);
                    // the previous vertex's NEE term, lit: :1246-1249 of the previous iteration
                    if ((ct & (F_NEE | F_OCCLUDED)) == F_NEE) Lr = Lr + nee_prev;
                    Lr = Lr + T * mat.emission;
                } else {
                    hit_barycentrics(S, h, o, d); // not carried through the traversal: the same test gives the same bits
                    rng = COLD(C_RNG, slot);
                    T = v3(COLDF(C_TX, slot), COLDF(C_TY, slot), COLDF(C_TZ, slot));
                    Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                    bounce = (kBounceInCt ? (ct >> (CT_SHIFT + RT_FLAT_BOUNCE_SHIFT)) : COLD(C_BOUNCE, slot)) + 1u;
                    // the previous vertex's NEE term, lit: :1246-1249 of the previous iteration
                    if ((ct & (F_NEE | F_OCCLUDED)) == F_NEE) Lr = Lr + v3(COLDF(C_NEEX, slot), COLDF(C_NEEY, slot), COLDF(C_NEEZ, slot));
                    surf = resolve_hit(S, h, o, d);
                    mat = load_material(S, surf.material_id);
                    Lr = Lr + T * mat.emission;
                    es = sample_environment(P.env, rng);
                }
                const float cos_nee = fmax_(0.0f, dot(surf.normal, es.direction));
                const bool want_shadow = cos_nee > 0.0f && es.pdf > 0.0f; // a shadow ray is cast: :1246
                const Frame frame = make_frame(surf.normal); // shader.wgsl:1252 and :1133 build the same frame
                const V3 wo = to_frame_local(frame, -d);
                V3 nee = v3(0.0f, 0.0f, 0.0f);
                if (want_shadow) { // what :1247-1249 adds if the shadow ray comes back unoccluded
                    RT_MARK2(5);
                    DBG_WAVE_TICK(23); DBG_ADD(24, 1);
                    const V3 wi = to_frame_local(frame, es.direction);
                    V3 scattering;
                    float pdf_bsdf;
                    bsdf_eval_pdf_local(wo, wi, mat, scattering, pdf_bsdf);
                    const float w = power_heuristic(es.pdf, pdf_bsdf);
                    nee = T * w * es.radiance * scattering * cos_nee / es.pdf;
                }
                RT_MARK2(6);
                SHADE_STAMP(13);
                const BsdfSample bs = bsdf_sample_in_frame(d, surf.normal, frame, wo, mat, rng);
                RT_MARK2(15);
                SHADE_STAMP(14);
                bool finished = false, nee_counts = want_shadow;
                if (bs.dir.x == 0.0f && bs.dir.y == 0.0f && bs.dir.z == 0.0f) {
                    Lr = bs.scattering; // the shader's debug colours overwrite the radiance, NEE term included
                    nee_counts = false; // (the shadow ray is still cast, as in the shader)
                    finished = true;
                } else if (bs.pdf <= 0.0f) {
                    finished = true;
                } else {
                    const float c = fmax_(0.0f, dot(surf.normal, bs.dir));
                    T = T * (bs.scattering * (c / bs.pdf));
                    if (length(T) < 0.001f) finished = true;
                }
                if (bounce >= P.max_bounces) finished = true;
                if (finished && !want_shadow) {
                    store_sample(P.sample_buf + (size_t)COLD(C_OUT, slot) * 3u, Lr);
                    SET_TAG(slot, TAG_FREE);
                } else {
                    SETC(C_LX, slot, Lr.x); SETC(C_LY, slot, Lr.y); SETC(C_LZ, slot, Lr.z);
                    if (nee_counts) { SETC(C_NEEX, slot, nee.x); SETC(C_NEEY, slot, nee.y); SETC(C_NEEZ, slot, nee.z); }
                    if (!finished) {
                        SETC(C_LASTPDF, slot, bs.pdf);
                        SETC(C_TX, slot, T.x); SETC(C_TY, slot, T.y); SETC(C_TZ, slot, T.z);
                        if (kRngHot) SETHU(H_T, slot, rng); else COLD(C_RNG, slot) = rng;
                        if (!kBounceInCt) COLD(C_BOUNCE, slot) = bounce;
                        SETH(H_EX, slot, bs.dir.x); SETH(H_EY, slot, bs.dir.y); SETH(H_EZ, slot, bs.dir.z);
                    }
                    if (want_shadow) { SETH(H_SX, slot, es.direction.x); SETH(H_SY, slot, es.direction.y); SETH(H_SZ, slot, es.direction.z); }
                    SETH(H_OX, slot, surf.point.x); SETH(H_OY, slot, surf.point.y); SETH(H_OZ, slot, surf.point.z); // both rays start at the hit point
                    if (!kRngHot && !kCoop) SETH(H_T, slot, RT_INFINITY);
                    SET_CT(slot, kBounceInCt ? (bounce << RT_FLAT_BOUNCE_SHIFT) : 0u,
                           (want_shadow ? (uint32_t)F_SHADOW : 0u) | (finished ? 0u : (uint32_t)F_EXT) | (nee_counts ? (uint32_t)F_NEE : 0u), TAG_TRACE);
                }
            }
        } else {
            // ---------------- FINISH: the path ended at its last vertex; its shadow ray is back
            RT_MARK2(7);
            if (on) {
                const uint32_t ct = CT_OF(slot);
                V3 Lr = v3(COLDF(C_LX, slot), COLDF(C_LY, slot), COLDF(C_LZ, slot));
                if ((ct & (F_NEE | F_OCCLUDED)) == F_NEE) Lr = Lr + v3(COLDF(C_NEEX, slot), COLDF(C_NEEY, slot), COLDF(C_NEEZ, slot));
                store_sample(P.sample_buf + (size_t)COLD(C_OUT, slot) * 3u, Lr);
                SET_TAG(slot, TAG_FREE);
            }
        }
        RT_MARK(15);
        RT_WAVE_HANDOVER(); // tags, hot and cold columns: the next census / stage reads them from other lanes
        DBG_STAMP(16 + dbg_stage); // the stage just run
    }
#undef HOT
#undef HOTF
#undef SETHU
#undef SETH
#undef CT_OF
#undef TAG_OF
#undef SET_TAG
#undef SET_CT
#undef COLD
#undef BEST_REF
#undef BEST_T
#undef COLDF
#undef SETC

    RT_MARK(11);
    unsigned long long n_work64 = n_work;
    for (int off = 32; off > 0; off >>= 1) {
        n_paths += __shfl_down(n_paths, off);
        n_ext += __shfl_down(n_ext, off);
        n_shadow += __shfl_down(n_shadow, off);
        if (TRAV != 2) n_work64 += __shfl_down(n_work64, off);
    }
    if (lane0 == 0) {
        atomicAdd(&P.stats[0], n_paths);
        atomicAdd(&P.stats[1], n_ext);
        atomicAdd(&P.stats[2], n_shadow);
        if (TRAV != 2) atomicAdd(&P.stats[3], n_work64);
    }
#ifdef RT_INSTRUMENT
    for (int i = 0; i < RT_DBG_N; i++) {
        unsigned long long v = dbg.c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane0 == 0 && v) atomicAdd(&P.stats[4 + i], v);
    }
#endif
}
