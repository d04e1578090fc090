// rt_coop.h — the cooperative wide walk (TRAV 6): a WAVE traces a batch of rays together.
//
// The per-lane walks (trace_preorder, trace_wide) give every ray a lane and let the lane walk its tree: rays need 3-25 node visits and
// hold 0-15 triangles, so half of the lanes idle in every loop, and half of what the busy ones issue is bookkeeping for rounds, votes,
// windows and a register stack (profiles/r03_walk_bounds.txt).  What the result depends on leaves the SCHEDULE free: the slab tests
// decide which leaves a ray meets (boxes nest, rounding is monotone: a leaf's own box is hit only if every ancestor's is), every record of
// such a leaf is tested, and the winner is the minimum of (t, visiting rank) whatever the order (rt_device.h, "wide walk").  So here the unit
// of work is not a ray but an ITEM:
//     node item  (ray, wide node)          -> four exact box tests; a hit interior child is a new node item, a hit leaf a leaf item
//     leaf item  (ray, first record, n)    -> n primitive tests (n <= 8: one leaf of the reference's tree)
// and the wave keeps two lists of them in LDS: a QUEUE of node items (a ring: oldest first) and a stack of leaf items.  A trip pops up to 64
// items — one per lane, any ray — and pushes what they turn up with wave64 ballots + prefix popcounts (the compaction north_star names, at item
// granularity).  A lane is idle only when a list holds fewer than 64 items, i.e. at the very end of a batch; there are no rounds, no per-ray
// stack, nothing to park: every ray of the batch is finished when both lists are empty.  (Two wave votes keep the trips dense: a slot's push is
// skipped when no lane has it, and a leaf trip hands its stragglers back as items of their own — see the leaf trip.)  Node items go oldest first because that keeps
// the end of a batch short: the shallow items, whose subtrees take the most trips, are worked off while there is plenty beside them, and what is
// left at the end are the deep ones, one trip from their leaves (newest first, the last old item's whole subtree was walked alone: 72-78 % of
// the lanes busy in the node trips instead of 90 %, tools/coop_sim.py and profiles/r04_coop_walk.txt).  The rays live where TRACE's rays always lived, in the pool's hot LDS columns, addressed by
// slot; a lane reads the ray of its item from there (six LDS dwords) instead of keeping one ray in registers.
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

#define RT_COOP_NCAP 384u   // node-queue entries in LDS (a ring)
#define RT_COOP_LCAP 320u   // leaf-stack entries in LDS: 63 may wait, one node trip adds at most 4 x 64; the pool kernel's compaction list lives here too
#define RT_COOP_GCAP 4096u  // node items a wave may spill to its arena block
#define RT_COOP_MIN_LDS_CAP 320u // (run-time cap of the LDS part, tests: 63 + 256 must fit after the spills)
#define RT_COOP_NARROW_AT 3072u  // outstanding node items beyond which a wave pops one item a trip: GCAP - 3072 - 256 - 64 >= 3 x (wide levels <= 25)
#define RT_COOP_LIFO_AT 512u     // ... beyond which it pops the newest 64 instead of the oldest (0: always — the first version, A/B)
#define RT_COOP_MAX_RECORDS (1u << 21) // a leaf item names its first record in 21 bits
#define RT_COOP_MAX_NODES (1u << 24)   // a node item names its node in 24 bits
// item = slot << 25 | kind << 24 | payload (kind 1: the slot's shadow ray); node item payload = wide node, leaf item payload = first record << 3 | (records - 1)
#define RT_COOP_KIND 0x01000000u
#define RT_COOP_HEAD 0xff000000u

// the pool's hot columns as the cooperative walk lays them out (dword offsets from the wave's base; POOL slots a column)
template <uint32_t POOL>
struct CoopCols {
    static constexpr uint32_t O = 0u, E = 3u * POOL, S = 6u * POOL;
    static constexpr uint32_t BEST = 9u * POOL; // u64[POOL]: record (low dword) | t bits (high dword) of the extension ray's closest hit
    static constexpr uint32_t CT = 11u * POOL;  // stage tag | flags (rt_wavepool.h CtBits) | the walk's own flags below
    static constexpr uint32_t DWORDS = 12u * POOL;
    static_assert(POOL <= 128u && (BEST % 2u) == 0u, "7-bit slot ids; 8-byte aligned result cells");
};
enum CoopFlags : uint32_t {
    CF_OCCLUDED = 64u,   // == F_OCCLUDED (rt_wavepool.h): the shadow ray has hit something
    CF_TIE = 1u << 7,    // two records gave the extension ray the same closest t: redo with the exact walk
    CF_SLOW_E = 1u << 8, // the extension ray's 1/d is not finite / in range: the exact walk
    CF_SLOW_S = 1u << 9, // the shadow ray's
    CF_ALL = CF_TIE | CF_SLOW_E | CF_SLOW_S
};

RT_DEV uint32_t coop_lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
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
    uint32_t ns_h, ns_n;    // the ring's head (oldest item) and fill
    uint32_t ls_n, gs_n;
    uint32_t lds_cap, lifo_at, narrow_at;
    uint32_t leaf_quorum; // a leaf trip hands its stragglers back once fewer than this percentage of the wave's lanes still hold records (0: never)
    RT_DEV uint32_t ring(uint32_t i) const // the ring's i-th entry, counted from the head (i < 2 * RT_COOP_NCAP - head)
    {
        const uint32_t k = ns_h + i;
        return k >= RT_COOP_NCAP ? k - RT_COOP_NCAP : k;
    }
};

// Root items for up to 64 slots (one per lane; `valid` lanes name a slot whose tag word `ct` says which rays to trace: F_EXT 16, F_SHADOW 8).
// Sets up the result cell and the walk's flags; a ray the walk cannot take is flagged for coop_slow_rays instead of being pushed.
template <uint32_t POOL>
RT_DEV void coop_push_rays(uint32_t *W, CoopStacks &st, bool valid, uint32_t slot, uint32_t ct, uint32_t f_ext, uint32_t f_shadow)
{
    typedef CoopCols<POOL> C;
    const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
    const V3 de = v3(as_f(W[C::E + slot]), as_f(W[C::E + POOL + slot]), as_f(W[C::E + 2u * POOL + slot]));
    const V3 ds = v3(as_f(W[C::S + slot]), as_f(W[C::S + POOL + slot]), as_f(W[C::S + 2u * POOL + slot]));
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
    if (push_e) st.ns[st.ring(st.ns_n + coop_lanes_below(be))] = slot << 25;
    st.ns_n += (uint32_t)__popcll(be);
    if (push_s) st.ns[st.ring(st.ns_n + coop_lanes_below(bs))] = (slot << 25) | RT_COOP_KIND;
    st.ns_n += (uint32_t)__popcll(bs);
}

// Room for `n_new` (<= 256) more node items in the ring: the newest go to the arena, 64 items at a time.
RT_DEV void coop_make_room(CoopStacks &st, uint32_t n_new, uint32_t lane)
{
    while (st.ns_n + n_new > st.lds_cap && st.ns_n >= 64u && st.gs_n + 64u <= RT_COOP_GCAP) { // (wave-uniform)
        st.ns_n -= 64u;
        st.gs[st.gs_n + lane] = st.ns[st.ring(st.ns_n + lane)];
        st.gs_n += 64u;
    }
}

// One plane or sphere record (rare inside a mesh's tree): the types the triangle path of the leaf trip does not handle
template <class View>
RT_DEV float coop_test_other(const View &S, uint32_t rec, uint32_t type, const float4 (&r)[3], V3 o, V3 d)
{
    if (type == PRIM_SPHERE) return sphere_t(o, d, v3(r[0].x, r[0].y, r[0].z), r[1].y);
    const float4 r3 = S.prim(4u * rec + 3u);
    return plane_t(o, d, v3(r[0].x, r[0].y, r[0].z), v3(r[1].x, r[1].y, r[1].z), v3(r[2].x, r[2].y, r[2].z), v3(r3.x, r3.y, r3.z));
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
            // ---------------- leaf trip: one leaf item a lane, its records two at a time
            DBG_WAVE_TICK(14); // (diagnostic build: 14 / 28 leaf trips and their items, 12 / 13 record-loop trips and tests, 10 / 11 node trips and items)
            const uint32_t n_take = min(st.ls_n, 64u);
            const bool act0 = lane < n_take;
            const uint32_t item = st.ls[act0 ? st.ls_n - 1u - lane : 0u];
            st.ls_n -= n_take;
            const uint32_t slot = item >> 25;
            const bool shadow = (item & RT_COOP_KIND) != 0u;
            const uint32_t ct = W[C::CT + slot];
            const uint32_t dcol = shadow ? C::S : C::E;
            const V3 o = v3(as_f(W[C::O + slot]), as_f(W[C::O + POOL + slot]), as_f(W[C::O + 2u * POOL + slot]));
            const V3 d = v3(as_f(W[dcol + slot]), as_f(W[dcol + POOL + slot]), as_f(W[dcol + 2u * POOL + slot]));
            float best_t = as_f(W[C::BEST + 2u * slot + 1u]); // (a filter only: the atomic decides)
            unsigned long long *const cell = reinterpret_cast<unsigned long long *>(W + C::BEST + 2u * slot);
            uint32_t rec = (item >> 3) & (RT_COOP_MAX_RECORDS - 1u);
            // a shadow ray that is already occluded needs nothing more (any hit: only did_hit is read)
            uint32_t left = (act0 & !(shadow & anyhit_shadow & ((ct & CF_OCCLUDED) != 0u))) ? (item & 7u) + 1u : 0u;
            const uint32_t n_started = (uint32_t)__popcll(__ballot(left != 0u));
            DBG_ADD(28, left != 0u ? 1 : 0); DBG_ADD(29, act0 ? 1 : 0);
            // (a wave-uniform loop: every lane stays until the wave is through, so that the stack's fill — a scalar — is only ever changed by
            // all lanes together; a lane without records sits the trips out)
            for (;;) {
                const unsigned long long more = __ballot(left != 0u);
                if (more == 0ull) break;
                // Leaves hold 1-5 records (a fifth of them five): the third pair trip would run for a fifth of the lanes.  Once fewer than
                // leaf_quorum percent of the lanes that came in with records still hold some, what they hold goes back on the stack as items of
                // its own — to be tested in a later, fuller trip — and this trip ends.  (Never before the first pair trip: an item shrinks each
                // time round; a lane's leftover is part of one leaf: the stack's bound stands.)
                if ((uint32_t)__popcll(more) * 100u < n_started * st.leaf_quorum) {
                    if (left != 0u) st.ls[st.ls_n + coop_lanes_below(more)] = (item & RT_COOP_HEAD) | (rec << 3) | (left - 1u);
                    st.ls_n += (uint32_t)__popcll(more);
                    break;
                }
                if (left == 0u) continue;
                DBG_WAVE_TICK(12);
                DBG_ADD(13, left >= 2u ? 2 : 1);
                const bool two = left >= 2u;
                work += two ? 2u : 1u;
                const uint32_t rec_a = rec, rec_b = two ? rec + 1u : rec;
                float4 ra[3], rb[3];
                S.template prim_rec<3>(rec_a, ra);
                S.template prim_rec<3>(rec_b, rb);
                const uint32_t ty_a = as_u(ra[0].w) & 3u, ty_b = as_u(rb[0].w) & 3u;
                float u, v;
                float ta = triangle_t(o, d, v3(ra[0].x, ra[0].y, ra[0].z), v3(ra[1].x, ra[1].y, ra[1].z), v3(ra[2].x, ra[2].y, ra[2].z), u, v);
                float tb = triangle_t(o, d, v3(rb[0].x, rb[0].y, rb[0].z), v3(rb[1].x, rb[1].y, rb[1].z), v3(rb[2].x, rb[2].y, rb[2].z), u, v);
                if (ty_a != PRIM_TRIANGLE) ta = coop_test_other(S, rec_a, ty_a, ra, o, d);
                if (ty_b != PRIM_TRIANGLE) tb = coop_test_other(S, rec_b, ty_b, rb, o, d);
                // a record that repeats an earlier record of its leaf bit for bit can never win (the reference keeps the first of equals):
                // the upload marks it (r2.w), it counts as a miss — otherwise every hit of doubled geometry would be a tie
                if ((as_u(ra[2].w) & 1u) != 0u) ta = RT_NO_HIT;
                if (!two || (as_u(rb[2].w) & 1u) != 0u) tb = RT_NO_HIT;
                if (shadow) {
                    if ((ta >= 0.0f) | (tb >= 0.0f)) {
                        atomicOr(&W[C::CT + slot], (uint32_t)CF_OCCLUDED);
                        if (anyhit_shadow) left = 0u;
                    }
                } else {
                    if ((ta >= 0.0f) & (ta <= best_t)) {
                        const unsigned long long old = atomicMin(cell, ((unsigned long long)as_u(ta) << 32) | rec_a);
                        if (((uint32_t)(old >> 32) == as_u(ta)) & ((uint32_t)old != rec_a)) atomicOr(&W[C::CT + slot], (uint32_t)CF_TIE);
                        best_t = ta;
                    }
                    if ((tb >= 0.0f) & (tb <= best_t)) {
                        const unsigned long long old = atomicMin(cell, ((unsigned long long)as_u(tb) << 32) | rec_b);
                        if (((uint32_t)(old >> 32) == as_u(tb)) & ((uint32_t)old != rec_b)) atomicOr(&W[C::CT + slot], (uint32_t)CF_TIE);
                        best_t = tb;
                    }
                }
                rec += 2u;
                left = left > 2u ? left - 2u : 0u;
            }
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
        const uint32_t slot = item >> 25;
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
                if (((im >> k) & 1u) != 0u) st.ns[st.ring(st.ns_n + coop_lanes_below(bi[k]))] = head | (child0 + (uint32_t)k);
                st.ns_n += (uint32_t)__popcll(bi[k]);
            }
        }
        // ---- ... and the hit leaves: one item each, first record and count (a leaf's records are a run of its node's)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t m = ((hm >> k) & 1u) != 0u ? as_u(n[4 + k].w) : 0u;
            const unsigned long long bl = __ballot(m != 0u);
            if (bl != 0ull) {
                if (m != 0u) st.ls[st.ls_n + coop_lanes_below(bl)] = head | ((rec_base + (uint32_t)__builtin_ctz(m)) << 3) | ((uint32_t)__popc(m) - 1u);
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
