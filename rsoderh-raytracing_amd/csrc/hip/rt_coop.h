// rt_coop.h — the cooperative wide walk (TRAV 6): a WAVE traces a batch of rays together.
//
// The per-lane walks (trace_preorder, trace_wide) give every ray a lane and let the lane walk its tree: rays need 3-25 node visits and
// hold 0-15 triangles, so half of the lanes idle in every loop, and half of what the busy ones issue is bookkeeping for rounds, votes,
// windows and a register stack (profiles/r03_walk_bounds.txt).  What the result depends on leaves the SCHEDULE free: the slab tests
// decide which leaves a ray meets (boxes nest, rounding is monotone: a leaf's own box is hit only if every ancestor's is), every record of
// such a leaf is tested, and the winner is the minimum of (t, visiting rank) whatever the order (rt_device.h, "wide walk").  So here the unit
// of work is not a ray but an ITEM:
//     node item  (ray, wide node)          -> four exact box tests; a hit interior child is a new node item, a hit leaf a leaf item
//     leaf item  (ray, first record, n)    -> n primitive tests (n <= 8: one leaf of the binary tree)
// and the wave keeps two lists of them in LDS: a QUEUE of node items (a ring: oldest first) and a stack of leaf items.  A node trip pops up to
// 64 items — one per lane, any ray — and pushes what they turn up with wave64 ballots + prefix popcounts (the compaction north_star names, at
// item granularity; a slot's push is skipped when no lane of the wave has it).  A leaf trip pops up to 64 leaf items and spreads their RECORDS
// over the lanes (a prefix sum of the record counts: leaves hold 1-8 records, and one item a lane would leave the lanes with short leaves idle):
// up to 128 records a trip, two a lane, any item, any ray.  A lane is idle only when a list runs low, i.e. at the very end of a batch; there
// are no rounds, no per-ray stack, nothing to park: every ray of the batch is finished when both lists are empty.  Node items go oldest first
// because that keeps the end of a batch short: the shallow items, whose subtrees take the most trips, are worked off while there is plenty
// beside them, and what is left at the end are the deep ones, one trip from their leaves (newest first, the last old item's whole subtree was
// walked alone: 72-78 % of the lanes busy in the node trips instead of 90 %, tools/coop_sim.py and profiles/r04_coop_walk.txt).  The rays live
// where TRACE's rays always lived, in the pool's hot LDS columns, addressed by slot; a lane reads the ray of its item from there (six LDS dwords)
// instead of keeping one ray in registers.
//
// Results.  Extension ray: one 64-bit LDS cell per slot, t's bits << 32 | record, folded with ds_min_u64 — accepted t are positive floats,
// so unsigned order is float order and the cell ends as the closest hit, lowest record among equal t.  The reference wants the lowest
// visiting RANK among equal t (rt_device.h, prim_rank): an equal t is seen by whichever atomic comes second (it returns the cell it met),
// which flags the slot, and a flagged ray is walked again, alone, by the exact fixed-order walk (trace_preorder).  So are rays with a
// non-finite or out-of-range 1/d (axis-parallel: the nesting argument needs a finite reciprocal).  Both are next to none.
// Shadow ray: only did_hit is read (shader.wgsl:1249), so a hit ORs F_OCCLUDED into the slot's tag word and the ray's other items are
// dropped as they surface.
//
// The node queue can outgrow LDS (a batch of 256 rays in a deep tree).  Three steps keep it bounded: beyond coop_lifo_at outstanding items the
// wave pops the NEWEST 64 instead of the oldest (depth first: the frontier stops growing with the breadth of the tree); what still does not fit
// the ring is spilled to the wave's arena block, newest 64 items at a time, and comes back when the ring runs low; and should even that block
// fill up (coop_narrow_at), the wave pops ONE newest item a trip — a plain depth-first walk, which adds at most three items a level — until it
// has room again (tests/test_gpu_parity.py forces all three).
#pragma once
#include "rt_device.h"

#define RT_COOP_NCAP 320u   // node-queue entries in LDS (a ring): 63 may wait, one node trip adds at most 4 x 64
#define RT_COOP_MAP 64u     // dwords of the leaf trip's map: 128 records x 16 bits
#define RT_COOP_LCAP 320u   // leaf-stack entries in LDS: 63 may wait, one node trip adds at most 4 x 64; the pool kernel's compaction list lives here too
#define RT_COOP_GCAP 4096u  // node items a wave may spill to its arena block
#define RT_COOP_MIN_LDS_CAP 320u // (run-time cap of the LDS part, tests: 63 + 256 must fit after the spills)
#define RT_COOP_NARROW_AT 3072u  // outstanding node items beyond which a wave pops one item a trip: GCAP - 3072 - 256 - 64 >= 3 x (wide levels <= 25)
#define RT_COOP_LIFO_AT 512u     // ... beyond which it pops the newest 64 instead of the oldest (0: always — the first version, A/B)
#ifndef RT_COOP_SLOT_BITS
#define RT_COOP_SLOT_BITS 7u // a pool of up to 128 slots (build-time A/B: 8 with -DRT_COOP_POOL=192u -DRT_COOP_BLOCK=768)
#endif
// item = slot << RT_COOP_SLOT_SHIFT | kind (1: the slot's shadow ray) | payload; node item payload = wide node, leaf item payload = first record << 3 | (records - 1)
#define RT_COOP_SLOT_SHIFT (32u - RT_COOP_SLOT_BITS)
#define RT_COOP_KIND (1u << (RT_COOP_SLOT_SHIFT - 1u))
#define RT_COOP_HEAD (~(RT_COOP_KIND - 1u))
#define RT_COOP_MAX_NODES RT_COOP_KIND           // a node item names its node in the 24 (23) bits below the kind bit
#define RT_COOP_MAX_RECORDS (RT_COOP_KIND >> 3)  // a leaf item names its first record in 21 (20) bits

// the pool's hot columns as the cooperative walk lays them out (dword offsets from the wave's base; POOL slots a column)
template <uint32_t POOL>
struct CoopCols {
    static constexpr uint32_t O = 0u, E = 3u * POOL, S = 6u * POOL;
    static constexpr uint32_t BEST = 9u * POOL; // u64[POOL]: record (low dword) | t bits (high dword) of the extension ray's closest hit
    static constexpr uint32_t CT = 11u * POOL;  // stage tag | flags (rt_wavepool.h CtBits) | the walk's own flags below
    static constexpr uint32_t DWORDS = 12u * POOL;
    static_assert(POOL <= (1u << RT_COOP_SLOT_BITS) && (BEST % 2u) == 0u, "slot ids of RT_COOP_SLOT_BITS bits; 8-byte aligned result cells");
};
enum CoopFlags : uint32_t {
    CF_OCCLUDED = 64u,   // == F_OCCLUDED (rt_wavepool.h): the shadow ray has hit something
    CF_TIE = 1u << 7,    // two records gave the extension ray the same closest t: redo with the exact walk
    CF_SLOW_E = 1u << 8, // the extension ray's 1/d is not finite / in range: the exact walk
    CF_SLOW_S = 1u << 9, // the shadow ray's
    CF_ALL = CF_TIE | CF_SLOW_E | CF_SLOW_S
};

RT_DEV uint32_t coop_lanes_below(unsigned long long m, uint32_t base = 0u) // base + how many lanes below this one are in m (the count starts AT base: no add)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}
// can the walk take this ray?  1/d by the short reciprocal (exactly the IEEE quotient there, rt_math.h) and finite, origin finite
RT_DEV bool coop_ray_ok(V3 o, V3 d)
{
    const bool short_ok = ((int)rt_rcp_short_ok(d.x) & (int)rt_rcp_short_ok(d.y) & (int)rt_rcp_short_ok(d.z)) != 0;
    const float finite = ((o.x + o.y) + o.z) * 0.0f; // NaN exactly when a component is infinite or NaN (an overflowing sum only sends a ray the long way round)
    return short_ok & (finite == 0.0f);
}

// The wave's two lists.  Every member is wave-uniform (scalar registers): counts come from ballots.
struct CoopStacks {
    uint32_t *ns, *ls, *gs; // node queue (LDS ring of RT_COOP_NCAP entries), leaf stack (LDS), the node queue's overflow (global, this wave's arena block)
    uint16_t *map;          // the leaf trip's map (LDS, 128 entries): which item, which of its records, each test slot takes
    uint32_t ns_h, ns_n;    // the ring's head (oldest item) and fill
    uint32_t ls_n, gs_n;
    uint32_t lds_cap, lifo_at, narrow_at;
    RT_DEV uint32_t ring(uint32_t i) const // the ring's i-th entry, counted from the head (i < 2 * RT_COOP_NCAP - head)
    {
        const uint32_t k = ns_h + i;
        return k >= RT_COOP_NCAP ? k - RT_COOP_NCAP : k;
    }
    RT_DEV uint32_t ring_new(unsigned long long m) const // where this lane's new entry goes: behind the fill, one place per lane of m
    {
        const uint32_t k = coop_lanes_below(m, ns_h + ns_n);
        return k >= RT_COOP_NCAP ? k - RT_COOP_NCAP : k;
    }
};

// Room for `n_new` (<= 256) more node items in the ring: the newest go to the arena, 64 items at a time.
RT_DEV void coop_make_room(CoopStacks &st, uint32_t n_new, uint32_t lane)
{
    while (st.ns_n + n_new > st.lds_cap && st.ns_n >= 64u && st.gs_n + 64u <= RT_COOP_GCAP) { // (wave-uniform)
        st.ns_n -= 64u;
        st.gs[st.gs_n + lane] = st.ns[st.ring(st.ns_n + lane)];
        st.gs_n += 64u;
    }
}

// Root items for up to 64 slots (one per lane; `valid` lanes name a slot whose tag word `ct` says which rays to trace: F_EXT 16, F_SHADOW 8).
// Sets up the result cell and the walk's flags; a ray the walk cannot take is flagged for coop_slow_rays instead of being pushed.
template <uint32_t POOL>
RT_DEV void coop_push_rays(uint32_t *W, CoopStacks &st, bool valid, uint32_t slot, uint32_t ct, uint32_t f_ext, uint32_t f_shadow)
{
    typedef CoopCols<POOL> C;
    const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
    const V3 de = v3(as_f(W[C::E + slot]), as_f(W[C::E + POOL + slot]), as_f(W[C::E + 2u * POOL + slot]));
    const V3 ds = v3(as_f(W[C::S + slot]), as_f(W[C::S + POOL + slot]), as_f(W[C::S + 2u * POOL + slot]));
    if (st.ns_n + 128u > st.lds_cap) coop_make_room(st, 128u, coop_lanes_below(~0ull)); // (pools of more than 128 slots: a third chunk of root items may not fit the ring)
    const bool want_e = valid & ((ct & f_ext) != 0u), want_s = valid & ((ct & f_shadow) != 0u);
    const bool push_e = want_e & coop_ray_ok(o, de), push_s = want_s & coop_ray_ok(o, ds);
    if (valid) {
        uint32_t c = ct & ~(uint32_t)(CF_ALL | CF_OCCLUDED);
        c |= (want_e & !push_e) ? (uint32_t)CF_SLOW_E : 0u;
        c |= (want_s & !push_s) ? (uint32_t)CF_SLOW_S : 0u;
        W[C::CT + slot] = c;
        if (want_e) { W[C::BEST + 2u * slot] = 0u; W[C::BEST + 2u * slot + 1u] = as_u(RT_INFINITY); }
    }
    const unsigned long long be = __ballot(push_e), bs = __ballot(push_s);
    if (push_e) st.ns[st.ring_new(be)] = slot << RT_COOP_SLOT_SHIFT;
    st.ns_n += (uint32_t)__popcll(be);
    if (push_s) st.ns[st.ring_new(bs)] = (slot << RT_COOP_SLOT_SHIFT) | RT_COOP_KIND;
    st.ns_n += (uint32_t)__popcll(bs);
}

// One plane or sphere record (rare inside a mesh's tree): the types the triangle path of the leaf trip does not handle
template <class View>
RT_DEV float coop_test_other(const View &S, uint32_t rec, uint32_t type, const float4 (&r)[3], V3 o, V3 d)
{
    if (type == PRIM_SPHERE) return sphere_t(o, d, v3(r[0].x, r[0].y, r[0].z), r[1].y);
    const float4 r3 = S.prim(4u * rec + 3u);
    return plane_t(o, d, v3(r[0].x, r[0].y, r[0].z), v3(r[1].x, r[1].y, r[1].z), v3(r[2].x, r[2].y, r[2].z), v3(r3.x, r3.y, r3.z));
}

// The test passes of a leaf trip: lane p tests record p of the trip's dense sequence (N = 2: and record p + 64).  map[p] = lane that popped the
// record's item | which of the item's records << 6; the popped items are still where they were, `top` entries up the leaf stack.
template <uint32_t POOL, uint32_t N, class View>
RT_DEV void coop_test_records(DBG_DECL const View &S, uint32_t *W, const CoopStacks &st, uint32_t top, uint32_t n_tests, uint32_t lane, uint32_t &work)
{
    typedef CoopCols<POOL> C;
    bool on[N];
    uint32_t e[N], item[N], rec[N];
    float4 r[N][3];
#pragma unroll
    for (uint32_t h = 0; h < N; h++) {
        on[h] = 64u * h + lane < n_tests;
        e[h] = st.map[on[h] ? 64u * h + lane : 0u];
    }
#pragma unroll
    for (uint32_t h = 0; h < N; h++) item[h] = st.ls[top - (e[h] & 63u)]; // (still there: nothing is pushed during a leaf trip)
#pragma unroll
    for (uint32_t h = 0; h < N; h++) {
        rec[h] = on[h] ? ((item[h] >> 3) & (RT_COOP_MAX_RECORDS - 1u)) + (e[h] >> 6) : 0u;
        S.template prim_rec<3>(rec[h], r[h]);
    }
#pragma unroll
    for (uint32_t h = 0; h < N; h++) {
        const uint32_t slot = item[h] >> RT_COOP_SLOT_SHIFT;
        const bool shadow = (item[h] & RT_COOP_KIND) != 0u;
        const uint32_t dcol = shadow ? C::S : C::E;
        const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
        const V3 d = v3(as_f(W[dcol + slot]), as_f(W[dcol + POOL + slot]), as_f(W[dcol + 2u * POOL + slot]));
        const float best_t = as_f(W[C::BEST + 2u * slot + 1u]); // (a filter only: the atomic decides)
        DBG_WAVE_TICK(12);
        DBG_ADD(13, on[h] ? 1 : 0);
        work += on[h] ? 1u : 0u;
        const uint32_t ty = as_u(r[h][0].w) & 3u;
        float u, v;
        float t = triangle_t(o, d, v3(r[h][0].x, r[h][0].y, r[h][0].z), v3(r[h][1].x, r[h][1].y, r[h][1].z), v3(r[h][2].x, r[h][2].y, r[h][2].z), u, v);
        if (ty != PRIM_TRIANGLE) t = coop_test_other(S, rec[h], ty, r[h], o, d);
        // a record that repeats an earlier record of its leaf bit for bit can never win (the reference keeps the first of equals):
        // the upload marks it (r2.w), it counts as a miss — otherwise every hit of doubled geometry would be a tie
        if (!on[h] || (as_u(r[h][2].w) & 1u) != 0u) t = RT_NO_HIT;
        if (shadow) {
            if (t >= 0.0f) atomicOr(&W[C::CT + slot], (uint32_t)CF_OCCLUDED);
        } else if ((t >= 0.0f) & (t <= best_t)) {
            unsigned long long *const cell = reinterpret_cast<unsigned long long *>(W + C::BEST + 2u * slot);
            const unsigned long long old = atomicMin(cell, ((unsigned long long)as_u(t) << 32) | rec[h]);
            if (((uint32_t)(old >> 32) == as_u(t)) & ((uint32_t)old != rec[h])) atomicOr(&W[C::CT + slot], (uint32_t)CF_TIE);
        }
    }
}

// Runs the stacks dry.  W: the wave's hot columns (CoopCols<POOL>); `work` += node items + records tested by this lane.
template <uint32_t POOL, class View>
RT_DEV void coop_trace(DBG_DECL const View &S, uint32_t *W, CoopStacks &st, bool anyhit_shadow, uint32_t lane, uint32_t &work)
{
    typedef CoopCols<POOL> C;
#ifdef RT_INSTRUMENT // (diagnostic build: wave time of the node trips / of the leaf trips, counters 25 / 26)
    unsigned long long t_trip = __builtin_amdgcn_s_memtime();
#define COOP_STAMP(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0u) dbg.c[i] += t_now - t_trip; t_trip = t_now; } while (0)
#else
#define COOP_STAMP(i) do { } while (0)
#endif
    for (;;) {
        // the hand-over between trips: items, result cells and flags are written by one lane and read by another
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (st.ls_n >= 64u || (st.ns_n == 0u && st.gs_n == 0u && st.ls_n != 0u)) {
            // ---------------- leaf trip: up to 128 RECORDS, one a lane and pass, of as many leaf items as hold them
            // Leaves hold 1-8 records (suzanne: 1-5, a fifth of them five).  One item a lane left the lanes with short leaves idle while the
            // long ones were worked off (60 % of the test slots used, profiles/r04_coop_walk.txt); so the items are spread out first: every
            // lane pops an item, an exclusive prefix sum of the record counts (four ballots: the counts are 4-bit numbers) gives each item its
            // place in a dense sequence of records, the items whose records end within the first 128 are taken — a run from the top of the
            // stack, the rest stays — and each writes "record j of the item lane i popped" into the map at its places.  Then lane p tests
            // record p (and p + 64): whichever item it belongs to, whichever ray.
            DBG_WAVE_TICK(14); // (diagnostic build: 14 / 28 leaf trips and their items, 12 / 13 test passes and records, 10 / 11 node trips and items)
            const uint32_t n_take = min(st.ls_n, 64u), top = st.ls_n - 1u;
            const bool act0 = lane < n_take;
            uint32_t n_rec;
            {
                const uint32_t item = st.ls[act0 ? top - lane : 0u];
                const bool shadow = (item & RT_COOP_KIND) != 0u;
                const uint32_t ct = W[C::CT + (item >> RT_COOP_SLOT_SHIFT)];
                // a shadow ray that is already occluded needs nothing more (any hit: only did_hit is read): its item is popped and dropped
                n_rec = (act0 & !(shadow & anyhit_shadow & ((ct & CF_OCCLUDED) != 0u))) ? (item & 7u) + 1u : 0u;
            }
            DBG_ADD(28, n_rec != 0u ? 1 : 0); DBG_ADD(29, act0 ? 1 : 0);
            uint32_t first = 0u; // records of the lanes below
#pragma unroll
            for (uint32_t b = 0; b < 4u; b++) first += coop_lanes_below(__ballot(((n_rec >> b) & 1u) != 0u)) << b;
            const bool taken = act0 & (first + n_rec <= 128u); // (a run of lanes from 0: the prefix sums only grow; lane 0 always)
            const uint32_t n_pop = (uint32_t)__popcll(__ballot(taken));
            const uint32_t n_tests = (uint32_t)__builtin_amdgcn_readlane((int)(first + n_rec), (int)(n_pop - 1u));
            st.ls_n -= n_pop;
            for (uint32_t j = 0; j < 8u; j++) { // (wave-uniform: as many rounds as the longest leaf taken has records)
                const bool w = taken & (j < n_rec);
                if (__ballot(w) == 0ull) break;
                if (w) st.map[first + j] = (uint16_t)(lane | (j << 6));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (both records of a lane are fetched before the first is tested: two passes of one record each would wait for memory twice)
            if (n_tests > 64u) coop_test_records<POOL, 2u>(DBG_ARG S, W, st, top, n_tests, lane, work);
            else if (n_tests != 0u) coop_test_records<POOL, 1u>(DBG_ARG S, W, st, top, n_tests, lane, work); // (0: every item popped was a dropped one)
            COOP_STAMP(26);
            continue;
        }
        // ---------------- node trip
        if (st.ns_n < 64u && st.gs_n != 0u) { // what was spilled comes back, a block at a time (order is free)
            st.gs_n -= 64u;
            st.ns[st.ring(st.ns_n + lane)] = st.gs[st.gs_n + lane];
            st.ns_n += 64u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (st.ns_n == 0u) break; // both stacks empty, nothing spilled: every ray of the batch is done
        DBG_WAVE_TICK(10);
        const uint32_t n_out = st.ns_n + st.gs_n;
        const uint32_t n_take = n_out > st.narrow_at ? 1u : min(st.ns_n, 64u);
        const bool act0 = lane < n_take;
        uint32_t item;
        if (n_out > st.lifo_at) { // (wave-uniform) the newest: depth first
            item = st.ns[st.ring(act0 ? st.ns_n - 1u - lane : 0u)];
        } else { // the oldest
            item = st.ns[st.ring(act0 ? lane : 0u)];
            st.ns_h = st.ring(n_take);
        }
        st.ns_n -= n_take;
        const uint32_t slot = item >> RT_COOP_SLOT_SHIFT;
        const bool shadow = (item & RT_COOP_KIND) != 0u;
        const uint32_t ct = W[C::CT + slot];
        const bool act = act0 & !(shadow & anyhit_shadow & ((ct & CF_OCCLUDED) != 0u));
        const uint32_t dcol = shadow ? C::S : C::E;
        const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
        const V3 d = v3(as_f(W[dcol + slot]), as_f(W[dcol + POOL + slot]), as_f(W[dcol + 2u * POOL + slot]));
        const V3 inv = v3(rt_rcp_short(d.x), rt_rcp_short(d.y), rt_rcp_short(d.z)); // (coop_ray_ok: the short form is the quotient for these)
        float4 n[8];
        S.wnode(act ? (item & (RT_COOP_MAX_NODES - 1u)) : 0u, n);
        // the node's eight .w words (DevScene::wnodes): [0] first interior child | interior-slot mask << 26, [1] first record of the leaf
        // children, [4 + k] slot k's records as a mask from there (0: not a leaf)
        uint32_t hm = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 n0 = n[2 * k], n1 = n[2 * k + 1];
            const float ax = (n0.x - o.x) * inv.x, bx = (n1.x - o.x) * inv.x;
            const float ay = (n0.y - o.y) * inv.y, by = (n1.y - o.y) * inv.y;
            const float az = (n0.z - o.z) * inv.z, bz = (n1.z - o.z) * inv.z;
            const float t_0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz)), 0.0f);
            const float t_1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz)), RT_INFINITY);
            hm |= !(t_0 > t_1) ? (1u << k) : 0u;
        }
        hm = act ? hm : 0u;
        DBG_ADD(11, act ? 1 : 0); DBG_ADD(30, act0 ? 1 : 0);
        const uint32_t wa = as_u(n[0].w), head = item & RT_COOP_HEAD;
        const uint32_t im = hm & (wa >> 26), child0 = wa & 0x3ffffffu, rec_base = as_u(n[1].w);
        work += act ? 1u : 0u;
        // ---- push: the hit interior children (consecutive nodes: slot k is child0 + k) ...
        unsigned long long bi[4];
        uint32_t n_new = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bi[k] = __ballot(((im >> k) & 1u) != 0u);
            n_new += (uint32_t)__popcll(bi[k]);
        }
        if (st.ns_n + n_new > st.lds_cap) coop_make_room(st, n_new, lane);
        // (each slot's push behind a wave-uniform test: near the root no lane has a leaf in any slot, near the leaves few have interior
        // children in the later slots — the slot's prefix count, item and address are then never formed)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (bi[k] != 0ull) {
                if (((im >> k) & 1u) != 0u) st.ns[st.ring_new(bi[k])] = head | (child0 + (uint32_t)k);
                st.ns_n += (uint32_t)__popcll(bi[k]);
            }
        }
        // ---- ... and the hit leaves: one item each, first record and count (a leaf's records are a run of its node's)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t m = ((hm >> k) & 1u) != 0u ? as_u(n[4 + k].w) : 0u;
            const unsigned long long bl = __ballot(m != 0u);
            if (bl != 0ull) {
                if (m != 0u) st.ls[coop_lanes_below(bl, st.ls_n)] = head | ((rec_base + (uint32_t)__builtin_ctz(m)) << 3) | ((uint32_t)__popc(m) - 1u);
                st.ls_n += (uint32_t)__popcll(bl);
            }
        }
        COOP_STAMP(25);
    }
#undef COOP_STAMP
}

// The rays the walk could not take or could not decide (CF_SLOW_E / CF_SLOW_S / CF_TIE in the slot's tag word) by the exact fixed-order walk,
// one ray per lane and pass; a lane answers for N slots of the batch (`mine[k]`: slot[k] was in it).  Leaves CF_OCCLUDED and the result cell as
// the walk itself would have.
template <uint32_t POOL, uint32_t N, class View>
RT_DEV void coop_slow_rays(DBG_DECL const View &S, const DevScene &sc, uint32_t *W, const bool (&mine)[N], const uint32_t (&slots)[N], bool anyhit_shadow, uint32_t &work)
{
    typedef CoopCols<POOL> C;
    uint32_t jobs[N];
#pragma unroll
    for (uint32_t k = 0; k < N; k++) jobs[k] = mine[k] ? (W[C::CT + slots[k]] & (uint32_t)CF_ALL) : 0u;
    for (;;) {
        uint32_t any = 0u;
#pragma unroll
        for (uint32_t k = 0; k < N; k++) any |= jobs[k];
        if (__ballot(any != 0u) == 0ull) break; // (wave-uniform; the loop is next to never entered)
        uint32_t which = 0u, job = jobs[0], slot = slots[0];
#pragma unroll
        for (uint32_t k = 1; k < N; k++)
            if (job == 0u) { which = k; job = jobs[k]; slot = slots[k]; }
        const bool shadow = (job & (uint32_t)(CF_TIE | CF_SLOW_E)) == 0u; // the extension ray first
        const uint32_t dcol = shadow ? C::S : C::E;
        const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
        const V3 d = v3(as_f(W[dcol + slot]), as_f(W[dcol + POOL + slot]), as_f(W[dcol + 2u * POOL + slot]));
        Hit h;
        h.t = RT_INFINITY; h.ref = 0u; h.src = SRC_BVH; h.u = h.v = 0.0f;
        uint32_t cur = job != 0u ? 0u : RT_END;
        if (job != 0u) { RT_MARK(4); } // (diagnostic build: how many rays come here — region counter 4, which only the flat kernel uses otherwise)
        trace_preorder(DBG_ARG S, sc, o, d, false, shadow & anyhit_shadow, 0xffffffffu, 0u, cur, h, nullptr, work);
        if (job != 0u) {
            if (shadow) {
                if (h.t < RT_INFINITY) W[C::CT + slot] |= (uint32_t)CF_OCCLUDED;
                job = 0u;
            } else {
                W[C::BEST + 2u * slot] = h.ref;
                W[C::BEST + 2u * slot + 1u] = as_u(h.t);
                job &= (uint32_t)CF_SLOW_S;
            }
#pragma unroll
            for (uint32_t k = 0; k < N; k++)
                if (which == k) jobs[k] = job;
        }
    }
}
