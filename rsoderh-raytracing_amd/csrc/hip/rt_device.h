// rt_device.h — device-side scene layout, BVH traversal, intersection, environment and BSDF.
//
// What it computes is the reference's WGSL integrator (src/shaders/shader.wgsl; the functions
// are cited one by one below); how it is organised is for CDNA4:
//  * the eight storage buffers are re-laid out at upload into 16-byte records that a lane
//    fetches with ds_read_b128 / global_load_dwordx4:
//      node      2 x float4  {min.xyz, idx}{max.xyz, len | axis<<16}           (32 B, was 48 B)
//      prim ref  4 x float4  one record per entry of `primitives`, in leaf order, holding the
//                            primitive itself (no second indirection) with per-primitive
//                            constants precomputed (triangle edges, sphere r^2, the two matrix
//                            rows the plane test uses)
//      material  4 x float4  BsdfMaterial incl. kd, lobe probabilities (make_bsdf_material,
//                            surface_kd, luminance are pure functions of the material)
//      tri normals 3 x float4, fetched once per closest hit, not per test
//  * hit attributes (normal, hit point) are resolved once after traversal from (t,u,v,prim),
//    not per candidate — same values, computed once;
//  * the NEE shadow query stops at its first hit (exactly result-preserving: the shader only reads
//    did_hit); pruning nodes entered beyond the current best t is opt-in (RSRT_FLAG_PRUNE) because
//    it is NOT exactly result-preserving (2 of 5.3e8 paths differ on house 1080p x 256 spp).
#pragma once
#include "rt_math.h"

// Diagnostic build only (-DRT_INSTRUMENT, librsrt_instr.so): per-lane counters of loop trips, folded
// into SIMD-efficiency figures by tools/simd_efficiency.py.  No counter exists in the product build.
#ifdef RT_INSTRUMENT
#define RT_DBG_N 32
struct DbgCounters { unsigned long long c[RT_DBG_N]; };
#define DBG_DECL DbgCounters &dbg,
#define DBG_ARG dbg,
#ifdef RT_SHADE_PROFILE // (tools/shade_profile.py) counters 10..15 hold the wave time of SHADE's sections instead of the loop trips
#define DBG_ADD(i, v) do { if ((i) < 10 || (i) > 15) dbg.c[i] += (v); } while (0)
#define DBG_WAVE_TICK(i) do { } while (0)
#else
#define DBG_ADD(i, v) dbg.c[i] += (v)
#define DBG_WAVE_TICK(i) do { if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) dbg.c[i] += 1; } while (0)
#endif
#else
#define DBG_DECL
#define DBG_ARG
#define DBG_ADD(i, v) do { } while (0)
#define DBG_WAVE_TICK(i) do { } while (0)
#endif

#define RT_INFINITY 1.70141183460469231732e+38f // shader.wgsl:235
#define RT_PI ((float)3.14159)                   // shader.wgsl:239
#define RT_INV_PI ((float)(1.0 / 3.14159))       // :240, const-evaluated in abstract float
#define RT_TWO_PI ((float)(2.0 * 3.14159))       // `2 * PI`

enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_TRIANGLE = 2 };
enum { SRC_BVH = 0, SRC_FB_SPHERE = 1, SRC_FB_PLANE = 2 };

struct DevScene {
    const float4 *nodes;       // 2 per node
    const float4 *prims;       // 4 per primitive reference, leaf order
    const float4 *tri_normals; // 3 per triangle
    const float4 *materials;   // 4 per material
    const float4 *fb_spheres;  // 4 per sphere, scene order (cast_ray's brute-force loop)
    const float4 *fb_planes;   // 4 per plane
    const float4 *escape;      // 8 octants x n_nodes u32 'next node when this subtree is done', packed 4 per float4
    const float4 *flat_leaves; // 2 per leaf: {min.xyz, record mask lo}{max.xyz, record mask hi} (trace_flat)
    uint32_t n_nodes, n_prims, n_tris, n_materials, n_spheres, n_planes;
    uint32_t stack_entries;    // per-lane traversal stack entries (tree depth + 1)
    uint32_t lds_float4s;      // float4 count of the LDS image (0 = scene stays in global memory)
    uint32_t lds_hybrid;       // the LDS image is only a traversal's head (SceneViewHybrid, one 1024-thread workgroup per CU): 1 nodes | escape
                               // links (tree walks), 2 the top block of the fixed-order walk's nodes, 3 the top block of the wide walk's
    uint32_t typed_leaves;     // no leaf has more than 8 primitives: leaf node words carry triangle / plane masks (trace_threaded_typed)
    // flat small-scene traversal (trace_flat): at most 64 primitive records, every child box inside its parent's
    uint32_t flat_ok, n_leaves;
    uint32_t tri_mask_lo, tri_mask_hi, plane_mask_lo, plane_mask_hi; // which records are triangles / planes
    const uint32_t *flat_rank; // [8 octants][16]: byte p = position of record p in that octant's depth-first visiting order
    // two-level cull of the flat loop for COHERENT batches (GEN's fused trace): the interior nodes two levels below the
    // root, each with the leaves under it as a mask; leaves above that level are always tested
    float cull_min[8][3], cull_max[8][3];
    uint32_t cull_mask[8], cull_always, n_cull;
    // fixed-order traversal (trace_preorder): its own node array — interior {min, link when hit}{max, link when missed},
    // leaf {min, first record | 1 << 31}{max, len | masks}, successor = next element — laid out top block first
    // (rsrt_upload_scene); [8 octants][n_prims] visiting ranks
    const float4 *pnodes;
    uint32_t n_pnodes;
    const uint32_t *prim_rank;
    // wide walk (trace_wide): 4-wide nodes collapsed from the binary tree, a node's interior children consecutive (groups of
    // siblings laid out hottest-first: the array's head is what LDS stages), 8 float4 per node: slot k's EXACT box = {[2k].xyz, [2k + 1].xyz}; the .w words are the node's: [0] first
    // interior child's node index | interior-slot mask << 26, [1] first record of the node's leaf children (contiguous), [2] / [3]
    // which of the 32 records from there are triangles / planes, [4 + k] slot k's records as a mask from there (0: not a leaf)
    const float4 *wnodes;
    uint32_t n_wnodes, wide_ok;
    uint32_t wide_deep; // the wide tree has more levels than the walk's register stack holds (RT_WSTACK + 1): TRAV 5
    uint32_t coop_ok;   // ... and the cooperative walk (rt_coop.h, TRAV 6) can name every record and node in its 32-bit work items
    // the wide walk's LDS image (lds_hybrid == 3): the first lds_wnodes wide nodes; everything else is read from global memory
    uint32_t lds_wnodes;
    const float4 *lds_src; // what stage_scene_lds copies (lds_float4s float4s): the image, nodes | escape links, or the pre-order nodes
};

struct DevEnv {
    const float4 *rgba;
    const uint4 *alias; // {probability bits, alias_index, pmf bits, pad}
    uint32_t width, height;
    // uniforms the shader recomputes per call; same f32 operations, done once at upload
    float wf, hf;           // f32(width), f32(height)
    float dphi_dtheta;      // (2*PI / f32(width)) * (PI / f32(height)), shader.wgsl:746-748
    uint32_t width_shift;   // log2(width) when width is a power of two, else 0xffffffff
};

// Scene accessor: either the LDS image (offsets in float4 units) or global memory.
template <bool LDS>
struct SceneView;

extern __shared__ float4 rt_smem[];

template <>
struct SceneView<true> {
    uint32_t o_nodes, o_prims, o_esc, o_trin, o_mats, o_fbs, o_fbp, o_flat;
    const float4 *pnodes; // (not part of the LDS image: small scenes run the flat loop, the pre-order walk is an A/B there)
    const float4 *wnodes; // (likewise)
    RT_DEV void pnode_pair(uint32_t e, float4 &n0, float4 &n1) const { n0 = pnodes[2u * e]; n1 = pnodes[2u * e + 1u]; }
    RT_DEV void wnode(uint32_t i, float4 (&n)[8]) const
    {
#pragma unroll
        for (int k = 0; k < 8; k++) n[k] = wnodes[8u * i + k];
    }
    RT_DEV float4 node(uint32_t i) const { return rt_smem[o_nodes + i]; }
    RT_DEV float4 flat(uint32_t i) const { return rt_smem[o_flat + i]; }
    RT_DEV float4 prim(uint32_t i) const { return rt_smem[o_prims + i]; }
    template <int N>
    RT_DEV void prim_rec(uint32_t rec, float4 (&r)[N]) const
    {
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = rt_smem[o_prims + 4u * rec + (uint32_t)k];
    }
    RT_DEV float4 trin(uint32_t i) const { return rt_smem[o_trin + i]; }
    RT_DEV float4 mat(uint32_t i) const { return rt_smem[o_mats + i]; }
    // record k of the primitive array named by src (SRC_BVH / SRC_FB_SPHERE / SRC_FB_PLANE)
    // (written as a sum of one-sided selects: a select BETWEEN two members makes the compiler index this
    // struct in scratch memory)
    RT_DEV float4 rec(uint32_t src, uint32_t i) const
    {
        return rt_smem[o_prims + (src == SRC_FB_SPHERE ? o_fbs - o_prims : 0u) + (src == SRC_FB_PLANE ? o_fbp - o_prims : 0u) + i];
    }
    RT_DEV uint32_t esc(uint32_t i) const { return reinterpret_cast<const uint32_t *>(rt_smem + o_esc)[i]; }
};
template <>
struct SceneView<false> {
    const float4 *nodes, *prims, *tri_normals, *materials, *fb_spheres, *fb_planes, *escape, *flat_leaves, *pnodes, *wnodes;
    RT_DEV void pnode_pair(uint32_t e, float4 &n0, float4 &n1) const { n0 = pnodes[2u * e]; n1 = pnodes[2u * e + 1u]; }
    RT_DEV void wnode(uint32_t i, float4 (&n)[8]) const
    {
#pragma unroll
        for (int k = 0; k < 8; k++) n[k] = wnodes[8u * i + k];
    }
    RT_DEV float4 node(uint32_t i) const { return nodes[i]; }
    RT_DEV float4 flat(uint32_t i) const { return flat_leaves[i]; }
    RT_DEV float4 prim(uint32_t i) const { return prims[i]; }
    template <int N>
    RT_DEV void prim_rec(uint32_t rec, float4 (&r)[N]) const
    {
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = prims[4u * rec + (uint32_t)k];
    }
    RT_DEV float4 trin(uint32_t i) const { return tri_normals[i]; }
    RT_DEV float4 mat(uint32_t i) const { return materials[i]; }
    RT_DEV float4 rec(uint32_t src, uint32_t i) const
    {
        const ptrdiff_t ds = fb_spheres - prims, dp = fb_planes - prims; // all three live in one allocation
        return prims[(src == SRC_FB_SPHERE ? ds : 0) + (src == SRC_FB_PLANE ? dp : 0) + (ptrdiff_t)i];
    }
    RT_DEV uint32_t esc(uint32_t i) const { return reinterpret_cast<const uint32_t *>(escape)[i]; }
};

// Mid-size and big scenes: the head of the chosen traversal's node array in LDS, shared by ONE 1024-thread workgroup per CU
// (DevScene::lds_hybrid says which: 1 nodes | escape links of the tree walks, 2 the top block of the fixed-order walk's elements, 3 the
// first n_w wide nodes); primitive records and the shading arrays are read from global memory.
struct SceneViewHybrid {
    uint32_t o_nodes, o_esc;
    const float4 *prims, *tri_normals, *materials, *fb_spheres, *fb_planes;
    const float4 *pnodes;
    uint32_t lds_f4;
    const float4 *wnodes;
    uint32_t head; // what the staged head holds (DevScene::lds_hybrid): the accessors of the other traversals read global memory
    uint32_t n_w;  // head == 3: wide nodes staged (a prefix of the array)
    // One array element: from LDS when it lies in the staged prefix, else from global memory.  Written as an unconditional ds_read (of
    // element 0 for the lanes that are past the prefix) plus a global load under a branch: a select between the two POINTERS makes the
    // compiler emit flat loads, which take the texture path even for LDS and are waited for one by one (measured on the fixed-order
    // walk: 12 % slower than three plain loads per step).
    RT_DEV void wnode(uint32_t i, float4 (&n)[8]) const
    {
        const bool in_lds = i < n_w;
        const uint32_t k0 = in_lds ? 8u * i : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) n[k] = rt_smem[k0 + (uint32_t)k];
        if (!in_lds) {
#pragma unroll
            for (int k = 0; k < 8; k++) n[k] = wnodes[8u * i + k];
        }
    }
    // Element e of the fixed-order walk: from LDS when it is in the top block, else from global memory.
    RT_DEV void pnode_pair(uint32_t e, float4 &n0, float4 &n1) const
    {
        const bool in_lds = (head == 2u) & (2u * e < lds_f4); // (the wide walk's fallback for rays with a non-finite 1/d comes here with head == 3)
        const uint32_t k = in_lds ? 2u * e : 0u;
        n0 = rt_smem[k];
        n1 = rt_smem[k + 1u];
        if (!in_lds) {
            n0 = pnodes[2u * e];
            n1 = pnodes[2u * e + 1u];
        }
    }
    RT_DEV float4 node(uint32_t i) const { return rt_smem[o_nodes + i]; }
    RT_DEV float4 prim(uint32_t i) const { return prims[i]; }
    template <int N>
    RT_DEV void prim_rec(uint32_t rec, float4 (&r)[N]) const
    {
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = prims[4u * rec + (uint32_t)k];
    }
    RT_DEV float4 trin(uint32_t i) const { return tri_normals[i]; }
    RT_DEV float4 mat(uint32_t i) const { return materials[i]; }
    RT_DEV float4 rec(uint32_t src, uint32_t i) const
    {
        const ptrdiff_t ds = fb_spheres - prims, dp = fb_planes - prims; // all three live in one allocation
        return prims[(src == SRC_FB_SPHERE ? ds : 0) + (src == SRC_FB_PLANE ? dp : 0) + (ptrdiff_t)i];
    }
    RT_DEV float4 flat(uint32_t) const { return float4{0.0f, 0.0f, 0.0f, 0.0f}; } // (the flat traversal needs the whole image in LDS)
    RT_DEV uint32_t esc(uint32_t i) const { return reinterpret_cast<const uint32_t *>(rt_smem + o_esc)[i]; }
};

// ------------------------------------------------------------------ RNG (shader.wgsl:605-631)
RT_DEV uint32_t random_u32_uniform(uint32_t &s)
{
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}
RT_DEV void salt_rng(uint32_t &s, uint32_t salt)
{
    s ^= salt;
    random_u32_uniform(s);
}
// f32(r) / 4294967295.0: the divisor rounds to 2^32, so this is the exact scaling by 2^-32
RT_DEV float random_uniform(uint32_t &s) { return (float)random_u32_uniform(s) * 2.3283064365386963e-10f; }

// ------------------------------------------------------------------ primitive tests
// Each returns the hit distance or a negative value for "no hit" (all accepted t are >= 1e-5).
#define RT_NO_HIT (-1.0f)

// cast_ray_sphere, shader.wgsl:295-333
RT_DEV float sphere_t(V3 o, V3 d, V3 pos, float r2)
{
    const float EPSILON = 1.0e-4f;
    V3 l = o - pos;
    float a = dot(d, d);
    float b = 2.0f * dot(d, l);
    float c = dot(l, l) - r2;
    float disc = b * b - 4.0f * a * c;
    float t;
    if (disc < 0.0f) return RT_NO_HIT;
    if (disc == 0.0f) {
        t = -0.5f * b / a;
    } else {
        float sq = rsrt_sqrtf(disc);
        float q = (b > 0.0f) ? -0.5f * (b + sq) : -0.5f * (b - sq);
        float t0 = q / a, t1 = c / q;
        if (t0 < EPSILON) t = t1;
        else if (t1 < EPSILON) t = t0;
        else t = fmin_(t0, t1);
    }
    return (t < EPSILON) ? RT_NO_HIT : t;
}
// cast_ray_plane, shader.wgsl:362-391; rx / rz = the x and z rows of base_change_matrix
RT_DEV float plane_t(V3 o, V3 d, V3 pos, V3 n, V3 rx, V3 rz)
{
    float den = dot(n, d);
    if (fabs_(den) < 0.0001f) return RT_NO_HIT;
    float t = dot(n, pos - o) / den;
    if (t < 0.001f) return RT_NO_HIT;
    V3 il = madd(d, t, o) - pos;
    float px = __builtin_fmaf(rx.z, il.z, __builtin_fmaf(rx.y, il.y, rx.x * il.x));
    float pz = __builtin_fmaf(rz.z, il.z, __builtin_fmaf(rz.y, il.y, rz.x * il.x));
    if (px < 0.0f || 1.0f < px || pz < 0.0f || 1.0f < pz) return RT_NO_HIT;
    return t;
}
// cast_ray_triangle, shader.wgsl:409-444.  Predicated instead of early returns (the lanes of a wave
// run in lockstep anyway): every comparison is the shader's, with its NaN behaviour (a NaN compares
// false and falls through, exactly as the chain of `if .. return` does).
RT_DEV float triangle_t(V3 o, V3 d, V3 a, V3 e0, V3 e1, float &u, float &v)
{
    const V3 op = o - a;
    const V3 p0 = cross(op, e0);
    const V3 p1 = cross(d, e1);
    const float det = dot(e0, p1);
    const float inv = rt_rcp(det);
    u = dot(op, p1) * inv;
    v = dot(d, p0) * inv;
    const float t = dot(e1, p0) * inv;
    const bool reject = (fabs_(det) < 1.0e-8f) | (u < 0.0f) | (1.0f < u) | (v < 0.0f) | (1.0f < (u + v)) | (t < 1.0e-5f);
    return reject ? RT_NO_HIT : t;
}

struct Hit {
    float t;       // RT_INFINITY while nothing is hit
    uint32_t ref;  // record index in the array named by src
    uint32_t src;  // SRC_*
    float u, v;
    RT_DEV bool did_hit() const { return t < RT_INFINITY; }
};

struct Surface {
    V3 point, normal;
    uint32_t material_id;
};

template <class View>
RT_DEV float4 hit_record(const View &S, const Hit &h, uint32_t k)
{
    return S.rec(h.src, 4u * h.ref + k);
}

// The HitInfo fields the shader fills at every accepted test (shader.wgsl:335-359, :393-405,
// :446-465), computed once for the closest hit.
template <class View>
RT_DEV Surface resolve_hit(const View &S, const Hit &h, V3 o, V3 d)
{
    Surface s;
    float4 r0 = hit_record(S, h, 0);
    uint32_t tag = as_u(r0.w);
    uint32_t type = tag & 3u;
    s.material_id = tag >> 2;
    s.point = madd(d, h.t, o);
    V3 p0 = v3(r0.x, r0.y, r0.z);
    if (type == PRIM_TRIANGLE) {
        RT_MARK2(8);
        uint32_t tri = as_u(hit_record(S, h, 1).w);
        float4 a = S.trin(3u * tri), b = S.trin(3u * tri + 1u), c = S.trin(3u * tri + 2u);
        V3 n0 = v3(a.x, a.y, a.z), n1 = v3(a.w, b.x, b.y), n2 = v3(b.z, b.w, c.x);
        V3 n = normalize((1.0f - h.u - h.v) * n0 + h.u * n1 + h.v * n2);
        if (dot(n, d) > 0.0f) n = n * -1.0f;
        s.normal = n;
    } else if (type == PRIM_SPHERE) {
        RT_MARK2(9);
        float r2 = hit_record(S, h, 1).y;
        V3 n = normalize(s.point - p0);
        V3 co = p0 - o;
        if (dot(co, co) - r2 < 1.0e-6f) n = n * -1.0f;
        s.normal = n;
    } else {
        RT_MARK2(10);
        float4 r1 = hit_record(S, h, 1);
        V3 n = v3(r1.x, r1.y, r1.z);
        if (dot(o, n) < 0.0f) n = n * -1.0f; // origin NOT plane-relative, shader.wgsl:394
        s.normal = n;
    }
    RT_MARK2(11);
    return s;
}

// u, v of a triangle hit, recomputed from the ray and the record (cast_ray_triangle computes them with t;
// the same operations on the same inputs give the same bits, so they need not travel with the hit)
template <class View>
RT_DEV void hit_barycentrics(const View &S, Hit &h, V3 o, V3 d)
{
    h.u = h.v = 0.0f;
    const float4 r0 = hit_record(S, h, 0);
    if ((as_u(r0.w) & 3u) == PRIM_TRIANGLE) {
        const float4 r1 = hit_record(S, h, 1), r2 = hit_record(S, h, 2);
        (void)triangle_t(o, d, v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), h.u, h.v);
    }
}

template <class View>
RT_DEV float test_record(const View &S, uint32_t rec, uint32_t src, V3 o, V3 d, float &u, float &v)
{
    uint32_t i = 4u * rec;
    float4 r0 = S.rec(src, i);
    float4 r1 = S.rec(src, i + 1);
    uint32_t type = as_u(r0.w) & 3u;
    V3 p0 = v3(r0.x, r0.y, r0.z);
    if (type == PRIM_TRIANGLE) {
        float4 r2 = S.rec(src, i + 2);
        return triangle_t(o, d, p0, v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), u, v);
    } else if (type == PRIM_SPHERE) {
        return sphere_t(o, d, p0, r1.y);
    } else {
        float4 r2 = S.rec(src, i + 2);
        float4 r3 = S.rec(src, i + 3);
        return plane_t(o, d, p0, v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), v3(r3.x, r3.y, r3.z));
    }
}

// cast_ray_bvh, shader.wgsl:469-564 (+ ray_intersects_bounds :262-293).  Visit order, test
// order and the strict `<` replacement are the reference's, so ties resolve identically.
// `stack` points at this lane's column of the LDS stack, entries `stride` dwords apart.
template <bool ANYHIT, class View>
RT_DEV void trace_bvh(const View &S, V3 o, V3 d, bool prune, uint32_t *stack, uint32_t stride, Hit &h)
{
    V3 inv = rt_rcp3(d);
    h.t = RT_INFINITY;
    h.ref = 0;
    h.src = SRC_BVH;
    h.u = h.v = 0.0f;
    uint32_t sp = 0, cur = 0;
    for (;;) {
        float4 n0 = S.node(2u * cur), n1 = S.node(2u * cur + 1u);
        float t_0 = 0.0f, t_1 = RT_INFINITY;
        bool inside = true;
        {
            float tn = (n0.x - o.x) * inv.x, tf = (n1.x - o.x) * inv.x;
            if (tn > tf) { float s = tn; tn = tf; tf = s; }
            if (tn > t_0) t_0 = tn;
            if (tf < t_1) t_1 = tf;
            if (t_0 > t_1) inside = false;
        }
        if (inside) {
            float tn = (n0.y - o.y) * inv.y, tf = (n1.y - o.y) * inv.y;
            if (tn > tf) { float s = tn; tn = tf; tf = s; }
            if (tn > t_0) t_0 = tn;
            if (tf < t_1) t_1 = tf;
            if (t_0 > t_1) inside = false;
        }
        if (inside) {
            float tn = (n0.z - o.z) * inv.z, tf = (n1.z - o.z) * inv.z;
            if (tn > tf) { float s = tn; tn = tf; tf = s; }
            if (tn > t_0) t_0 = tn;
            if (tf < t_1) t_1 = tf;
            if (t_0 > t_1) inside = false;
        }
        if (inside && prune && t_0 > h.t) inside = false;
        uint32_t idx = as_u(n0.w), la = as_u(n1.w);
        uint32_t len = la & 0xffffu, axis = la >> 16;
        bool pop = true;
        if (inside) {
            if (len > 0u) {
                for (uint32_t i = 0; i < len; i++) {
                    float u, v;
                    float t = test_record(S, idx + i, SRC_BVH, o, d, u, v);
                    if (t >= 0.0f && t < h.t) {
                        h.t = t;
                        h.ref = idx + i;
                        h.u = u;
                        h.v = v;
                        if (ANYHIT) return;
                    }
                }
            } else {
                uint32_t near_child, far_child;
                if (comp(inv, axis) < 0.0f) { near_child = idx; far_child = cur + 1u; }
                else { near_child = cur + 1u; far_child = idx; }
                stack[sp * stride] = far_child;
                sp++;
                cur = near_child;
                pop = false;
            }
        }
        if (pop) {
            if (sp == 0u) break;
            sp--;
            cur = stack[sp * stride];
        }
    }
}

// cast_ray, shader.wgsl:567-601: BVH, and only if it missed, every sphere then every plane.
template <class View>
RT_DEV void trace_closest(const View &S, const DevScene &sc, V3 o, V3 d, bool prune, uint32_t *stack, uint32_t stride, Hit &h)
{
    trace_bvh<false>(S, o, d, prune, stack, stride, h);
    if (h.did_hit()) return;
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
}

// ------------------------------------------------------------------ environment (shader.wgsl:689-831)
RT_DEV void direction_to_equirectangular_uv(V3 d, float &u, float &v) // :710-714
{
    u = rsrt_atan2f(d.z, d.x) * RT_INV_PI * 0.5f + 0.5f;
    v = 0.5f - rsrt_asinf(d.y) * RT_INV_PI;
}
RT_DEV V3 equirectangular_uv_to_direction(float u, float v) // :718-732
{
    float phi = (2.0f * u - 1.0f) * RT_PI;
    float theta = RT_PI * v;
    float st = rsrt_sinf(theta), ct = rsrt_cosf(theta);
    return v3(st * rsrt_cosf(phi), ct, st * rsrt_sinf(phi));
}
RT_DEV float environment_pixel_solid_angle(float v, const DevEnv &e) // :739-749
{
    float theta = RT_PI * v;
    float sin_t = fmax_(1.0e-6f, rsrt_sinf(theta));
    return e.dphi_dtheta * sin_t; // (d_phi * d_theta) * sin_t
}
// Environment gathers (texels, alias entries): 64 MiB of tables read at random, 128 bytes fetched per 16-byte record, default cache
// policy (non-temporal gathers were measured in round 2, profiles/r02_l2_sweep.txt: L2 hit rate up, run time +3 .. +10 %)
typedef float rt_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u4v __attribute__((ext_vector_type(4)));
RT_DEV float4 env_texel(const DevEnv &e, size_t i)
{
    return e.rgba[i];
}
RT_DEV uint4 env_alias_stream(const DevEnv &e, size_t i) // the uniformly random slot of the alias method
{
    return e.alias[i];
}
RT_DEV uint4 env_alias(const DevEnv &e, size_t i)
{
    return e.alias[i];
}
RT_DEV uint32_t clamp_texel(float f, uint32_t n)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= (float)(n - 1u)) return n - 1u;
    return (uint32_t)f;
}
// textureSampleLevel(env, sampler, uv, 0).xyz — linear magnification, clamp-to-edge
// (src/state.rs:134-142), full-precision f32 weights a*(1-f) + b*f
// (in two halves, so that a caller can put work between the four gathers and their use)
struct EnvBilinearFetch {
    float4 t00, t10, t01, t11;
    float fx, fy;
    uint32_t sel; // (sample_env_bilinear_begin_pmf only) which of the four texels lies under (u, v): bit 0 column x1, bit 1 row y1, bit 2 none of them
};
RT_DEV EnvBilinearFetch sample_env_bilinear_begin(const DevEnv &e, float u, float v)
{
    float x = u * e.wf - 0.5f, y = v * e.hf - 0.5f;
    float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
    EnvBilinearFetch b;
    b.fx = x - xf; b.fy = y - yf;
    uint32_t x0 = clamp_texel(xf, e.width), x1 = clamp_texel(xf + 1.0f, e.width);
    uint32_t y0 = clamp_texel(yf, e.height), y1 = clamp_texel(yf + 1.0f, e.height);
    b.t00 = env_texel(e, (size_t)y0 * e.width + x0); b.t10 = env_texel(e, (size_t)y0 * e.width + x1);
    b.t01 = env_texel(e, (size_t)y1 * e.width + x0); b.t11 = env_texel(e, (size_t)y1 * e.width + x1);
    return b;
}
RT_DEV V3 sample_env_bilinear_finish(const EnvBilinearFetch &b)
{
    float gx = 1.0f - b.fx, gy = 1.0f - b.fy;
    V3 top = v3(b.t00.x, b.t00.y, b.t00.z) * gx + v3(b.t10.x, b.t10.y, b.t10.z) * b.fx;
    V3 bot = v3(b.t01.x, b.t01.y, b.t01.z) * gx + v3(b.t11.x, b.t11.y, b.t11.z) * b.fx;
    return top * gy + bot * b.fy;
}
RT_DEV V3 sample_env_bilinear(const DevEnv &e, float u, float v) { return sample_env_bilinear_finish(sample_env_bilinear_begin(e, u, v)); }
// environment_direction_pdf, :753-769 (uv already computed), likewise: the probability mass of the texel under (u, v) ...
RT_DEV float environment_direction_pmf(const DevEnv &e, float u, float v)
{
    uint32_t x = min(f2u(u * e.wf), e.width - 1u);
    uint32_t y = min(f2u(v * e.hf), e.height - 1u);
    uint32_t index = x + y * e.width;
    return as_f(env_alias(e, index).z);
}
// Packed environment (device layout only, rt_env_pack_kernel at upload): a texel's alpha — 0 in the reference's texture
// (src/texture.rs:112-115) and never sampled (`.xyz`, shader.wgsl:825-830) — carries the pmf of that texel's alias entry, and an
// alias entry's pad word the pmf of its alias TARGET.  An escaping ray's pdf (environment_direction_pdf) then needs no
// alias-table gather of its own: the texel under (u, v) is one of the four the bilinear fetch reads anyway (floor(p) is
// floor(p - 0.5) or that + 1, and both sides clamp alike); should it ever not be, the gather is still there.  The same copies
// of the same f32 values, so the bits cannot change; one 128-byte line request less per escape and per aliased NEE pick.
RT_DEV EnvBilinearFetch sample_env_bilinear_begin_pmf(const DevEnv &e, float u, float v)
{
    float x = u * e.wf - 0.5f, y = v * e.hf - 0.5f;
    float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
    EnvBilinearFetch b;
    b.fx = x - xf; b.fy = y - yf;
    uint32_t x0 = clamp_texel(xf, e.width), x1 = clamp_texel(xf + 1.0f, e.width);
    uint32_t y0 = clamp_texel(yf, e.height), y1 = clamp_texel(yf + 1.0f, e.height);
    b.t00 = env_texel(e, (size_t)y0 * e.width + x0); b.t10 = env_texel(e, (size_t)y0 * e.width + x1);
    b.t01 = env_texel(e, (size_t)y1 * e.width + x0); b.t11 = env_texel(e, (size_t)y1 * e.width + x1);
    const uint32_t px = min(f2u(u * e.wf), e.width - 1u), py = min(f2u(v * e.hf), e.height - 1u); // environment_direction_pmf's texel
    b.sel = (px != x0 ? 1u : 0u) | (py != y0 ? 2u : 0u) | ((((px != x0) & (px != x1)) | ((py != y0) & (py != y1))) ? 4u : 0u);
    return b;
}
RT_DEV float env_bilinear_pmf(const DevEnv &e, const EnvBilinearFetch &b, float u, float v)
{
    const float top = (b.sel & 1u) ? b.t10.w : b.t00.w, bot = (b.sel & 1u) ? b.t11.w : b.t01.w;
    float pmf = (b.sel & 2u) ? bot : top;
    if (b.sel & 4u) pmf = environment_direction_pmf(e, u, v); // (never taken: see above)
    return pmf;
}
// ... and the density it stands for
RT_DEV float environment_direction_pdf(const DevEnv &e, V3 dir, float u, float v)
{
    return environment_direction_pmf(e, u, v) / environment_pixel_solid_angle(v, e);
}
struct EnvironmentSample {
    V3 direction, radiance;
    float pdf;
};
// sample_environment (:782-820 + :689-706) in two halves, so that a caller can put work between the alias-table gather —
// a random 16-byte read of a 32 MB table, the first of two dependent memory round trips — and its use.
struct EnvironmentPick {
    uint32_t index;
    uint4 entry;
};
RT_DEV EnvironmentPick sample_environment_begin(const DevEnv &e, uint32_t &rng) // first draw, gather issued
{
    const uint32_t length = e.width * e.height;
    EnvironmentPick p;
    p.index = min(f2u(random_uniform(rng) * (float)length), length - 1u);
    p.entry = env_alias_stream(e, p.index);
    return p;
}
// PACKED: the alias target's pmf is read from the entry's pad word instead of the target's own entry (the flat
// kernel only — the walks have no register left for the fourth word of the entry, tests/test_code_object.py)
template <bool PACKED = false>
RT_DEV EnvironmentSample sample_environment_finish(const DevEnv &e, uint32_t &rng, const EnvironmentPick &p) // draws two to four
{
    const uint32_t index = p.index;
    const uint4 entry = p.entry;
    float u2 = random_uniform(rng);
    uint32_t pick = (u2 < as_f(entry.x)) ? index : entry.y;
    uint32_t x, y; // pick % width, pick / width
    if (e.width_shift != 0xffffffffu) { x = pick & (e.width - 1u); y = pick >> e.width_shift; }
    else { y = pick / e.width; x = pick - y * e.width; }
    float jx = random_uniform(rng), jy = random_uniform(rng);
    float u = ((float)x + jx) / e.wf;
    float v = ((float)y + jy) / e.hf;
    EnvironmentSample s;
    s.direction = equirectangular_uv_to_direction(u, v);
    s.radiance = sample_env_bilinear(e, u, v);
    float pmf;
    if (PACKED) pmf = (pick == index) ? as_f(entry.z) : as_f(entry.w); // .w: the pmf of the alias target (rt_env_pack_alias_kernel)
    else pmf = (pick == index) ? as_f(entry.z) : as_f(env_alias(e, pick).z);
    s.pdf = pmf / environment_pixel_solid_angle(v, e);
    return s;
}
RT_DEV EnvironmentSample sample_environment(const DevEnv &e, uint32_t &rng)
{
    const EnvironmentPick p = sample_environment_begin(e, rng);
    return sample_environment_finish(e, rng, p);
}

// ------------------------------------------------------------------ BSDF (shader.wgsl:55-84, 850-1210)
struct BsdfMaterial {
    V3 f0, kd, emission;
    float alpha, spec_prob, diff_prob;
};
template <class View>
RT_DEV BsdfMaterial load_material(const View &S, uint32_t id)
{
    float4 m1 = S.mat(4u * id + 1u), m2 = S.mat(4u * id + 2u), m3 = S.mat(4u * id + 3u);
    BsdfMaterial m;
    m.f0 = v3(m1.x, m1.y, m1.z);
    m.alpha = m1.w;
    m.emission = v3(m2.x, m2.y, m2.z);
    m.spec_prob = m2.w;
    m.kd = v3(m3.x, m3.y, m3.z);
    m.diff_prob = m3.w;
    return m;
}
struct Frame {
    V3 tangent, bitangent, normal;
};
RT_DEV Frame make_frame(V3 n) // :55-67
{
    V3 helper = (fabs_(n.z) < 0.999f) ? v3(0, 0, 1) : v3(1, 0, 0);
    V3 t = normalize(cross(helper, n));
    V3 b = cross(n, t);
    return Frame{t, b, n};
}
RT_DEV V3 to_frame_local(const Frame &f, V3 w) { return v3(dot(w, f.tangent), dot(w, f.bitangent), dot(w, f.normal)); }
RT_DEV V3 to_frame_world(const Frame &f, V3 l) { return normalize(f.tangent * l.x + f.bitangent * l.y + f.normal * l.z); }

RT_DEV float d_ggx(float ndh, float alpha) // :924-928
{
    float a2 = alpha * alpha;
    float den = (ndh * ndh) * (a2 - 1.0f) + 1.0f;
    return a2 / (RT_PI * den * den);
}
RT_DEV float lambda_ggx(float ndv, float alpha) // :1014-1020
{
    float ndv2 = ndv * ndv;
    return (rsrt_sqrtf(1.0f + alpha * alpha * (1.0f - ndv2) / ndv2) - 1.0f) / 2.0f;
}
RT_DEV float g1_ggx(float ndv, float alpha) { return rt_rcp(1.0f + lambda_ggx(ndv, alpha)); } // :1026-1028
RT_DEV V3 f_schlick(V3 f0, float c) // :1045-1051
{
    float x = 1.0f - saturate(c);
    float x2 = x * x;
    float x5 = x2 * x2 * x;
    return f0 + (v3(1, 1, 1) - f0) * x5;
}
RT_DEV V3 bsdf_eval_local(V3 wo, V3 wi, const BsdfMaterial &m) // :1053-1087
{
    if (wo.z <= 0.0f || wi.z <= 0.0f) return v3(0, 0, 0);
    V3 h = normalize(wo + wi);
    float ndh = saturate(h.z);
    float D = d_ggx(ndh, m.alpha);
    float G = g1_ggx(wo.z, m.alpha) * g1_ggx(wi.z, m.alpha);
    V3 F = f_schlick(m.f0, dot(h, wo));
    V3 fs = (D * G) / (4.0f * wo.z * wi.z) * F;
    V3 fd = m.kd * RT_INV_PI;
    return fd + fs;
}
RT_DEV float bsdf_pdf_local(V3 wo, V3 wi, const BsdfMaterial &m) // :1104-1114 with :1089-1102, :931-945, :914-919
{
    if (wo.z <= 0.0f || wi.z <= 0.0f) return 0.0f;
    float pdf_cos = wi.z / RT_PI; // wi.z > 0 here
    float pdf_spec;
    {
        V3 h = normalize(wo + wi);
        float wo_dot_h = fabs_(dot(wo, h));
        if (wo_dot_h <= 0.0f) pdf_spec = 0.0f;
        else {
            float hv;
            if (h.z <= 0.0f) hv = 0.0f;
            else hv = d_ggx(h.z, m.alpha) * g1_ggx(wo.z, m.alpha) * fmax_(0.0f, dot(wo, h)) / wo.z;
            pdf_spec = hv / (4.0f * wo_dot_h);
        }
    }
    return m.diff_prob * pdf_cos + m.spec_prob * pdf_spec;
}
RT_DEV V3 sample_cosine_hemisphere(float sx, float sy) // :892-901
{
    float r = rsrt_sqrtf(sx);
    float phi = RT_TWO_PI * sy;
    float x = r * rsrt_cosf(phi), y = r * rsrt_sinf(phi);
    float z = rsrt_sqrtf(fmax_(0.0f, 1.0f - x * x - y * y));
    return v3(x, y, z);
}
RT_DEV V3 sample_ggx_visible_half_vector(float sx, float sy, V3 wo, float alpha) // :962-1009
{
    V3 vs = normalize(wo * v3(alpha, alpha, 1.0f));
    float len2 = dot2(vs.x, vs.y, vs.x, vs.y);
    V3 alt = v3(-vs.y, vs.x, 0.0f) * inverse_sqrt(len2);
    V3 tx = (len2 > 0.0f) ? alt : v3(1, 0, 0);
    V3 ty = cross(vs, tx);
    float radius = rsrt_sqrtf(sx);
    float az = RT_TWO_PI * sy;
    float dx = radius * rsrt_cosf(az), dy = radius * rsrt_sinf(az);
    float a = rsrt_sqrtf(fmax_(0.0f, 1.0f - dx * dx));
    dy = (1.0f - vs.z) * a + vs.z * dy; // lerp_f32(a, dy, vs.z)
    V3 hs = dx * tx + dy * ty + rsrt_sqrtf(fmax_(0.0f, 1.0f - dx * dx - dy * dy)) * vs;
    return normalize(v3(alpha * hs.x, alpha * hs.y, fmax_(0.0f, hs.z)));
}
// bsdf_eval_local and bsdf_pdf_local (shader.wgsl:1053-1114) in one pass: they share h, D and G1(wo);
// every expression is the one the two separate functions evaluate.
RT_DEV void bsdf_eval_pdf_local(V3 wo, V3 wi, const BsdfMaterial &m, V3 &f, float &pdf)
{
    if (wo.z <= 0.0f || wi.z <= 0.0f) { f = v3(0, 0, 0); pdf = 0.0f; return; }
    const V3 h = normalize(wo + wi);
    const float g1_o = g1_ggx(wo.z, m.alpha);
    {
        const float ndh = saturate(h.z);
        const float D = d_ggx(ndh, m.alpha);
        const float G = g1_o * g1_ggx(wi.z, m.alpha);
        const V3 F = f_schlick(m.f0, dot(h, wo));
        const V3 fs = (D * G) / (4.0f * wo.z * wi.z) * F;
        f = m.kd * RT_INV_PI + fs;
    }
    {
        const float pdf_cos = wi.z / RT_PI;
        const float wo_dot_h_signed = dot(wo, h);
        const float wo_dot_h = fabs_(wo_dot_h_signed);
        float pdf_spec;
        if (wo_dot_h <= 0.0f) pdf_spec = 0.0f;
        else {
            const float hv = (h.z <= 0.0f) ? 0.0f : d_ggx(h.z, m.alpha) * g1_o * fmax_(0.0f, wo_dot_h_signed) / wo.z;
            pdf_spec = hv / (4.0f * wo_dot_h);
        }
        pdf = m.diff_prob * pdf_cos + m.spec_prob * pdf_spec;
    }
}

struct BsdfSample {
    V3 dir, scattering;
    float pdf;
};
// `frame` = make_frame(n) and `wo` = to_frame_local(frame, -ray_dir): callers that already have them
// (the NEE evaluation uses the same two) pass them in; the values are what the shader recomputes.
RT_DEV BsdfSample bsdf_sample_in_frame(V3 ray_dir, V3 n, const Frame &frame, V3 wo, const BsdfMaterial &m, uint32_t &rng) // :1116-1202
{
    V3 wo_world = -ray_dir;
    if (dot(n, wo_world) <= 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    if (wo.z <= 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 1, 0), 0.0f};
    V3 wi;
    float s = random_uniform(rng);
    if (s < m.diff_prob) {
        RT_MARK2(12);
        float s0 = s / fmax_(m.diff_prob, 1.e-6f);
        float s1 = random_uniform(rng);
        wi = sample_cosine_hemisphere(s0, s1);
    } else {
        RT_MARK2(13);
        float s0 = (s - m.diff_prob) / fmax_(m.spec_prob, 1.e-6f);
        float s1 = random_uniform(rng);
        V3 h = sample_ggx_visible_half_vector(s0, s1, wo, m.alpha);
        V3 i = -wo;
        wi = i - (2.0f * dot(h, i)) * h; // reflect(-wo, h)
        if (wi.z <= 0.0f) return BsdfSample{v3(1, 0, 0), v3(1, 0, 0), 0.0f};
    }
    RT_MARK2(14);
    V3 scattering;
    float pdf;
    bsdf_eval_pdf_local(wo, wi, m, scattering, pdf);
    V3 wi_world = to_frame_world(frame, wi);
    if (dot(n, wi_world) < 0.0f) return BsdfSample{v3(0, 0, 0), v3(0, 1, 0), 0.0f};
    return BsdfSample{wi_world, scattering, pdf};
}
RT_DEV BsdfSample bsdf_sample(V3 ray_dir, V3 n, const BsdfMaterial &m, uint32_t &rng)
{
    const Frame frame = make_frame(n);
    return bsdf_sample_in_frame(ray_dir, n, frame, to_frame_local(frame, -ray_dir), m, rng);
}
RT_DEV float power_heuristic(float a, float b) // :1206-1210
{
    float a2 = a * a, b2 = b * b;
    return a2 / (a2 + b2);
}

// ------------------------------------------------------------------ while-while traversal
// Same visit order, same tests, same strict-< replacement as trace_bvh (so the same result),
// organised for wave64 lockstep: every lane first walks interior nodes until it HOLDS a leaf
// (lanes that found theirs wait), then all holders test their leaf's primitives together.
// `anyhit` is per lane: an extension ray and an NEE shadow ray can share the loop.
template <class View>
RT_DEV void trace_ww(DBG_DECL const View &S, V3 o, V3 d, bool prune, bool anyhit, uint32_t *stack, uint32_t stride, Hit &h)
{
    const V3 inv = rt_rcp3(d);
    h.t = RT_INFINITY;
    h.ref = 0;
    h.src = SRC_BVH;
    h.u = h.v = 0.0f;
    uint32_t sp = 0, cur = 0;
    bool alive = true;
    while (alive) {
        uint32_t leaf_idx = 0, leaf_len = 0;
        // ---- descend until this lane holds a leaf or its stack runs dry
        DBG_WAVE_TICK(14);
        while (alive && leaf_len == 0u) {
            DBG_WAVE_TICK(10);
            DBG_ADD(11, 1);
            float4 n0 = S.node(2u * cur), n1 = S.node(2u * cur + 1u);
            float t_0 = 0.0f, t_1 = RT_INFINITY;
            bool inside = true;
            {
                float tn = (n0.x - o.x) * inv.x, tf = (n1.x - o.x) * inv.x;
                if (tn > tf) { float s = tn; tn = tf; tf = s; }
                if (tn > t_0) t_0 = tn;
                if (tf < t_1) t_1 = tf;
                if (t_0 > t_1) inside = false;
            }
            if (inside) {
                float tn = (n0.y - o.y) * inv.y, tf = (n1.y - o.y) * inv.y;
                if (tn > tf) { float s = tn; tn = tf; tf = s; }
                if (tn > t_0) t_0 = tn;
                if (tf < t_1) t_1 = tf;
                if (t_0 > t_1) inside = false;
            }
            if (inside) {
                float tn = (n0.z - o.z) * inv.z, tf = (n1.z - o.z) * inv.z;
                if (tn > tf) { float s = tn; tn = tf; tf = s; }
                if (tn > t_0) t_0 = tn;
                if (tf < t_1) t_1 = tf;
                if (t_0 > t_1) inside = false;
            }
            if (inside && prune && t_0 > h.t) inside = false;
            const uint32_t idx = as_u(n0.w), la = as_u(n1.w);
            const uint32_t len = la & 0xffffu, axis = la >> 16;
            if (inside && len > 0u) {
                leaf_idx = idx;
                leaf_len = len;
            } else if (inside) {
                const bool far_first = comp(inv, axis) < 0.0f;
                stack[sp * stride] = far_first ? cur + 1u : idx;
                sp++;
                cur = far_first ? idx : cur + 1u;
            } else if (sp == 0u) {
                alive = false;
            } else {
                sp--;
                cur = stack[sp * stride];
            }
        }
        // ---- test the held leaf
        if (leaf_len != 0u) {
            for (uint32_t i = 0; i < leaf_len; i++) {
                DBG_WAVE_TICK(12);
                DBG_ADD(13, 1);
                float u, v;
                float t = test_record(S, leaf_idx + i, SRC_BVH, o, d, u, v);
                if (t >= 0.0f && t < h.t) {
                    h.t = t;
                    h.ref = leaf_idx + i;
                    h.u = u;
                    h.v = v;
                    if (anyhit) { alive = false; break; }
                }
            }
            if (alive) {
                if (sp == 0u) alive = false;
                else { sp--; cur = stack[sp * stride]; }
            }
        }
    }
}

// ray_intersects_bounds (shader.wgsl:262-293) without branches.  Exactly equivalent:
//  * t_0 / t_1 start at 0 / INFINITY and are only ever assigned non-NaN values, so
//    `if (tn > t_0) t_0 = tn` == maxNum(t_0, tn) and `if (tf < t_1) t_1 = tf` == minNum(t_1, tf)
//    (a NaN tn / tf — 0 * inf for an axis-parallel ray on a box face — is ignored by both forms);
//  * t_0 only grows and t_1 only shrinks, so "t_0 > t_1 after some axis" == "t_0 > t_1 at the end";
//  * the near/far swap keeps the shader's compare-and-select (a min/max pair would move a NaN).
// Returns the slab entry distance through t_entry.
RT_DEV bool slab_test(float4 n0, float4 n1, V3 o, V3 inv, float &t_entry)
{
    float t_0 = 0.0f, t_1 = RT_INFINITY;
    {
        const float a = (n0.x - o.x) * inv.x, b = (n1.x - o.x) * inv.x;
        const bool sw = a > b;
        t_0 = __builtin_fmaxf(t_0, sw ? b : a);
        t_1 = __builtin_fminf(t_1, sw ? a : b);
    }
    {
        const float a = (n0.y - o.y) * inv.y, b = (n1.y - o.y) * inv.y;
        const bool sw = a > b;
        t_0 = __builtin_fmaxf(t_0, sw ? b : a);
        t_1 = __builtin_fminf(t_1, sw ? a : b);
    }
    {
        const float a = (n0.z - o.z) * inv.z, b = (n1.z - o.z) * inv.z;
        const bool sw = a > b;
        t_0 = __builtin_fmaxf(t_0, sw ? b : a);
        t_1 = __builtin_fminf(t_1, sw ? a : b);
    }
    t_entry = t_0;
    return !(t_0 > t_1);
}

// ------------------------------------------------------------------ threaded (stackless) traversal
// For a fixed sign octant of the ray direction the reference's depth-first order (near child by
// sign(inv_dir[split_axis]), shader.wgsl:536-547) is a FIXED sequence, so it can be threaded at
// upload: escape[octant][node] = the node the stack would pop to once `node`'s subtree is done
// (the far sibling for a near child, the parent's escape for a far child, RT_END at the end).
// Same boxes, same primitives, same order, same strict-< replacement as cast_ray_bvh — but the only
// traversal state is `cur`, so a traversal can stop after `budget` steps and be resumed later by
// any lane (h carries the best hit so far).
#define RT_END 0x1ffffffu // 25 bits: the pool kernel keeps cursor, flags and stage tag in one word
#ifndef RT_LEAFQ
#define RT_LEAFQ 4
#endif
// A lane keeps descending until it holds RT_LEAFQ leaves (or its traversal ends) before the wave
// switches to primitive testing: fewer and better-filled rounds than one leaf per round.  Leaves are
// tested in the order they were found, so the winner of a tie is unchanged, and without
// RSRT_FLAG_PRUNE the set of visited nodes does not depend on the best t at all, so queueing is
// exactly result-preserving.  (With the opt-in pruning, nodes entered while leaves wait are checked
// against a slightly older best t.)
template <class View>
RT_DEV void trace_threaded(DBG_DECL const View &S, uint32_t n_nodes, V3 o, V3 d, bool prune, bool anyhit, uint32_t budget,
                           uint32_t &cur, Hit &h, uint32_t &work)
{
    const V3 inv = rt_rcp3(d);
    const uint32_t octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    const uint32_t ebase = octant * n_nodes;
    uint32_t steps = 0;
    while (cur != RT_END && steps < budget) {
        DBG_WAVE_TICK(14);
        uint32_t qi[RT_LEAFQ], ql[RT_LEAFQ]; // held leaves: first primitive record, primitive count
        uint32_t nq = 0;
#pragma unroll
        for (int j = 0; j < RT_LEAFQ; j++) qi[j] = ql[j] = 0u;
        while (cur != RT_END && nq < RT_LEAFQ) {
            DBG_WAVE_TICK(10);
            DBG_ADD(11, 1);
            steps++;
            const float4 n0 = S.node(2u * cur), n1 = S.node(2u * cur + 1u);
            const uint32_t esc = S.esc(ebase + cur); // needed on every path but one: fetch it alongside the node
            float t_0;
            bool inside = slab_test(n0, n1, o, inv, t_0);
            inside = inside & !(prune & (t_0 > h.t));
            const uint32_t idx = as_u(n0.w), la = as_u(n1.w);
            const uint32_t len = la & 0xffffu, axis = la >> 16;
            const bool descend = inside & (len == 0u);
            const uint32_t near_child = ((octant >> axis) & 1u) ? idx : cur + 1u; // near child first
            if (inside & (len != 0u)) {
#pragma unroll
                for (int j = 0; j < RT_LEAFQ; j++) {
                    qi[j] = (nq == (uint32_t)j) ? idx : qi[j];
                    ql[j] = (nq == (uint32_t)j) ? len : ql[j];
                }
                nq++;
            }
            cur = descend ? near_child : esc; // subtree (or leaf) done: continue where the stack would pop to
        }
        // ---- test the held leaves in the order found
        uint32_t j = 0, i = 0;
        uint32_t leaf_i = qi[0], leaf_l = ql[0];
        while (j < nq) {
            DBG_WAVE_TICK(12);
            DBG_ADD(13, 1);
            steps++;
            const uint32_t rec = leaf_i + i;
            float u, v;
            const float t = test_record(S, rec, SRC_BVH, o, d, u, v);
            const bool better = (t >= 0.0f) & (t < h.t); // strict <: the first of equals keeps winning
            h.t = better ? t : h.t;
            h.ref = better ? rec : h.ref;
            h.u = better ? u : h.u;
            h.v = better ? v : h.v;
            i++;
            const bool next_leaf = i == leaf_l;
            i = next_leaf ? 0u : i;
            j += next_leaf ? 1u : 0u;
#pragma unroll
            for (int k = 1; k < RT_LEAFQ; k++) {
                leaf_i = (next_leaf & (j == (uint32_t)k)) ? qi[k] : leaf_i;
                leaf_l = (next_leaf & (j == (uint32_t)k)) ? ql[k] : leaf_l;
            }
            if (better & anyhit) { // the shadow query only wants to know whether anything is hit
                cur = RT_END;
                j = nq;
            }
        }
    }
    work += steps; // box steps + primitive tests: rsrt_stats.traversal_steps
}

// ------------------------------------------------------------------ typed leaf loops
// In trace_threaded's leaf loop the lanes of a wave hold primitives of all three types, so nearly every
// trip runs the triangle, the sphere AND the plane routine (~190 VALU) for ~27 tests.  When no leaf has
// more than 8 primitives (the reference's builder stops at 5) rsrt_upload_scene packs a triangle mask
// and a plane mask into the leaf's node word; a lane then ORs the masks of the leaves it holds into
// three 8-bit-per-leaf masks and the wave runs three HOMOGENEOUS loops — triangles, planes, spheres —
// each lane walking its own mask with ctz.  Testing out of depth-first order needs the tie rule spelled
// out: position p (leaf * 8 + index) IS the depth-first order within the round, the incumbent of earlier
// rounds counts as position -1, and of equal t the earlier position wins — exactly what the
// reference's strict `<` in visiting order decides.  Triangles go first, in order, so plain `<` does it
// there.
RT_DEV uint32_t take_lowest(uint32_t &mask)
{
    const uint32_t p = (uint32_t)__builtin_ctz(mask);
    mask &= mask - 1u;
    return p;
}

template <class View>
RT_DEV void trace_threaded_typed(DBG_DECL const View &S, uint32_t n_nodes, V3 o, V3 d, bool prune, bool anyhit, uint32_t budget, uint32_t &cur,
                                 Hit &h, uint32_t &work)
{
    static_assert(RT_LEAFQ <= 4, "one byte of each 32-bit mask per held leaf");
    const V3 inv = rt_rcp3(d);
    const uint32_t octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    const uint32_t ebase = octant * n_nodes;
    uint32_t steps = 0;
    while (cur != RT_END && steps < budget) {
        DBG_WAVE_TICK(14);
        uint32_t qi[RT_LEAFQ];
        uint32_t nq = 0, all_m = 0, tri_m = 0, pl_m = 0;
#pragma unroll
        for (int j = 0; j < RT_LEAFQ; j++) qi[j] = 0u;
        while (cur != RT_END && nq < RT_LEAFQ) {
            DBG_WAVE_TICK(10);
            DBG_ADD(11, 1);
            steps++;
            const float4 n0 = S.node(2u * cur), n1 = S.node(2u * cur + 1u);
            const uint32_t esc = S.esc(ebase + cur);
            float t_0;
            bool inside = slab_test(n0, n1, o, inv, t_0);
            inside = inside & !(prune & (t_0 > h.t));
            const uint32_t idx = as_u(n0.w), la = as_u(n1.w);
            const uint32_t len = la & 0xffffu, hi = la >> 16; // hi: split axis (interior) / triangle mask | plane mask << 8 (leaf)
            const bool descend = inside & (len == 0u);
            const uint32_t near_child = ((octant >> (hi & 3u)) & 1u) ? idx : cur + 1u;
            if (inside & (len != 0u)) {
#pragma unroll
                for (int j = 0; j < RT_LEAFQ; j++) qi[j] = (nq == (uint32_t)j) ? idx : qi[j];
                const uint32_t sh = 8u * nq;
                all_m |= ((1u << len) - 1u) << sh;
                tri_m |= (hi & 0xffu) << sh;
                pl_m |= (hi >> 8) << sh;
                steps += len;
                nq++;
            }
            cur = descend ? near_child : esc;
        }
        uint32_t sp_m = all_m & ~(tri_m | pl_m);
        uint32_t best_p = 0xffffffffu; // position of the round's best so far; the incumbent is "before everything"
        // ---- triangles, in visiting order
        while (tri_m != 0u) {
            DBG_WAVE_TICK(12);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(tri_m);
            uint32_t base = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) base = ((p >> 3) == (uint32_t)j) ? qi[j] : base;
            const uint32_t rec = base + (p & 7u);
            const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u), r2 = S.prim(4u * rec + 2u);
            float u, v;
            const float t = triangle_t(o, d, v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), u, v);
            const bool better = (t >= 0.0f) & (t < h.t);
            h.t = better ? t : h.t;
            h.ref = better ? rec : h.ref;
            best_p = better ? p : best_p;
            if (better & anyhit) { cur = RT_END; tri_m = pl_m = sp_m = 0u; }
        }
        // ---- planes, then spheres: out of order, so equal t is decided by position
        while (pl_m != 0u) {
            DBG_WAVE_TICK(15);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(pl_m);
            uint32_t base = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) base = ((p >> 3) == (uint32_t)j) ? qi[j] : base;
            const uint32_t rec = base + (p & 7u);
            const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u), r2 = S.prim(4u * rec + 2u), r3 = S.prim(4u * rec + 3u);
            const float t = plane_t(o, d, v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), v3(r3.x, r3.y, r3.z));
            const bool better = (t >= 0.0f) & ((t < h.t) | ((t == h.t) & (p < best_p) & (best_p != 0xffffffffu)));
            h.t = better ? t : h.t;
            h.ref = better ? rec : h.ref;
            best_p = better ? p : best_p;
            if (better & anyhit) { cur = RT_END; pl_m = sp_m = 0u; }
        }
        while (sp_m != 0u) {
            DBG_WAVE_TICK(28);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(sp_m);
            uint32_t base = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) base = ((p >> 3) == (uint32_t)j) ? qi[j] : base;
            const uint32_t rec = base + (p & 7u);
            const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u);
            const float t = sphere_t(o, d, v3(r0.x, r0.y, r0.z), r1.y);
            const bool better = (t >= 0.0f) & ((t < h.t) | ((t == h.t) & (p < best_p) & (best_p != 0xffffffffu)));
            h.t = better ? t : h.t;
            h.ref = better ? rec : h.ref;
            best_p = better ? p : best_p;
            if (better & anyhit) { cur = RT_END; sp_m = 0u; }
        }
    }
    work += steps;
}

// ------------------------------------------------------------------ flat traversal of small scenes
// f32 rounding is monotone, so with a FINITE reciprocal direction the slab interval of a box contains
// the slab interval of every box inside it: a ray that hits a node's box has hit all its ancestors'
// (tests/test_box_containment.py).  cast_ray_bvh's unpruned walk therefore tests exactly the leaves
// whose OWN boxes are hit, and the tree above them only decides the order.  For scenes of at most 64
// primitive records (all three scenes the reference ships) the walk is replaced by
//   1. one wave-uniform loop over the leaf boxes — no cursor, no escape links, no per-lane node
//      fetch: the box is the same for all lanes (an LDS broadcast read), every lane with a ray
//      is busy in every trip, and min/max replace the compare-and-swap (no NaN can occur);
//   2. a 64-bit mask of the records to test (OR of the hit leaves' masks), split by type, walked with
//      ctz in three homogeneous loops as in trace_threaded_typed.
// Order only matters for equal t: then the record that comes first in this octant's depth-first order
// wins (flat_rank, built at upload), which is what the reference's strict `<` in visiting order does.
// Rays with a zero / subnormal direction component (1/d infinite: 0 * inf = NaN makes a child box
// "hit" where its parent "misses") and scenes whose boxes do not nest keep the tree walk.
RT_DEV uint32_t flat_rank_of(const DevScene &sc, uint32_t octant, uint32_t rec)
{
    return (sc.flat_rank[octant * 16u + (rec >> 2)] >> (8u * (rec & 3u))) & 0xffu;
}

template <class View>
RT_DEV void trace_flat(DBG_DECL const View &S, const DevScene &sc, V3 o, V3 d, V3 inv, bool anyhit, uint32_t quorum, uint32_t &cur, unsigned long long &rem, Hit &h,
                       bool coherent = false)
{
    // cur != 0: a ray whose triangle loop was cut short by the vote below comes back with the triangles it has not
    // tested yet (`rem`) and its best hit so far (`h`); it needs no box test.  A batch of such rays only skips the loop.
    RT_MARK(4);
    const bool resumed = cur != 0u;
    uint32_t all_lo = 0u, all_hi = 0u;
    const uint32_t n_leaves = __ballot(!resumed) != 0ull ? sc.n_leaves : 0u; // (wave-uniform)
    // Two-level cull for a coherent batch (the camera rays of a 16 x 4 patch that GEN has just built): a leaf's box is hit only if
    // every ancestor's is (the containment the flat loop rests on), so the leaves under an interior node that NO lane of the wave
    // hits need no test at all.  Wave-uniform: the group boxes come from the kernel arguments (scalar registers), the skip is a
    // scalar branch.
    uint32_t active = 0xffffffffu;
    if (coherent && n_leaves != 0u) { // (measured, profiles/r03_fusion_ab.txt: -4.3 .. -5.3 % on the BASELINE frame)
        active = sc.cull_always;
        for (uint32_t g = 0; g < sc.n_cull; g++) {
            const float ax = (sc.cull_min[g][0] - o.x) * inv.x, bx = (sc.cull_max[g][0] - o.x) * inv.x;
            const float ay = (sc.cull_min[g][1] - o.y) * inv.y, by = (sc.cull_max[g][1] - o.y) * inv.y;
            const float az = (sc.cull_min[g][2] - o.z) * inv.z, bz = (sc.cull_max[g][2] - o.z) * inv.z;
            const float t_0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz)), 0.0f);
            const float t_1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz)), RT_INFINITY);
            if (__ballot(!resumed & !(t_0 > t_1)) != 0ull) active |= sc.cull_mask[g];
        }
    }
#pragma unroll 4 // ( -1.3 % against 1 with the 1024-thread workgroups; it measured the same with 256-thread ones)
    for (uint32_t L = 0; L < n_leaves; L++) {
        if (!((active >> L) & 1u)) continue; // (wave-uniform)
        RT_MARK(5);
        DBG_WAVE_TICK(10);
        DBG_ADD(11, 1);
        const float4 n0 = S.flat(2u * L), n1 = S.flat(2u * L + 1u);
        const float ax = (n0.x - o.x) * inv.x, bx = (n1.x - o.x) * inv.x;
        const float ay = (n0.y - o.y) * inv.y, by = (n1.y - o.y) * inv.y;
        const float az = (n0.z - o.z) * inv.z, bz = (n1.z - o.z) * inv.z;
        const float t_0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz)), 0.0f);
        const float t_1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz)), RT_INFINITY);
        const bool miss = t_0 > t_1;
        all_lo |= miss ? 0u : as_u(n0.w); // .w: the leaf's records as a 64-bit mask
        all_hi |= miss ? 0u : as_u(n1.w);
    }
    RT_MARK(6);
    const unsigned long long all_m = resumed ? rem : (((unsigned long long)all_hi << 32) | all_lo);
    const unsigned long long tri_all = ((unsigned long long)sc.tri_mask_hi << 32) | sc.tri_mask_lo;
    const unsigned long long pl_all = ((unsigned long long)sc.plane_mask_hi << 32) | sc.plane_mask_lo;
    unsigned long long tri_m = all_m & tri_all, pl_m = all_m & pl_all, sp_m = all_m & ~(tri_all | pl_all);
    const uint32_t octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
#define RT_FLAT_ACCEPT(t, rec)                                                                                         \
    bool better = ((t) >= 0.0f) & ((t) < h.t);                                                                         \
    if (((t) == h.t) & (h.t < RT_INFINITY)) better = flat_rank_of(sc, octant, (rec)) < flat_rank_of(sc, octant, h.ref); \
    h.t = better ? (t) : h.t;                                                                                          \
    h.ref = better ? (rec) : h.ref;
    // The vote: rays hold very different numbers of triangles (3.2 on average, 11 for the unluckiest lane of a wave on the
    // BASELINE scene), and every lane that is through idles until the last one is.  Once fewer than `quorum` percent of the
    // lanes that entered the loop still hold triangles the wave leaves it; what a lane has not tested goes back to the
    // scheduler in `rem` (every lane has tested at least one triangle by then, so a ray always advances).
    const uint32_t tri_started = (uint32_t)__popcll(__ballot(tri_m != 0ull));
    // Two triangles per trip: both records are requested together and the two tests are independent instruction streams
    // until their results are taken in (in record order, so that the any-hit exit sees the same first hit)
    while (tri_m != 0ull) {
        RT_MARK(7);
        DBG_WAVE_TICK(12);
        DBG_ADD(13, 1); DBG_ADD(25, 1); // (25: lane-trips of the pair loop, for tools/ledger.py)
        const uint32_t rec_a = (uint32_t)__builtin_ctzll(tri_m);
        tri_m &= tri_m - 1ull;
        const bool two = tri_m != 0ull;
        const uint32_t rec_b = two ? (uint32_t)__builtin_ctzll(tri_m) : rec_a;
        tri_m &= tri_m - 1ull; // (0 & anything = 0)
        const float4 a0 = S.prim(4u * rec_a), a1 = S.prim(4u * rec_a + 1u), a2 = S.prim(4u * rec_a + 2u);
        const float4 b0 = S.prim(4u * rec_b), b1 = S.prim(4u * rec_b + 1u), b2 = S.prim(4u * rec_b + 2u);
        float u, v;
        const float ta = triangle_t(o, d, v3(a0.x, a0.y, a0.z), v3(a1.x, a1.y, a1.z), v3(a2.x, a2.y, a2.z), u, v);
        const float tb = triangle_t(o, d, v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z), v3(b2.x, b2.y, b2.z), u, v);
        bool stop;
        {
            const float t = ta; // (the macro's parameter is spelled like the member)
            RT_FLAT_ACCEPT(t, rec_a)
            stop = better & anyhit;
        }
        if (two & !stop) {
            DBG_ADD(13, 1);
            const float t = tb;
            RT_FLAT_ACCEPT(t, rec_b)
            stop = better & anyhit;
        }
        if (stop) tri_m = pl_m = sp_m = 0ull;
        if ((uint32_t)__popcll(__ballot(tri_m != 0ull)) * 100u < tri_started * quorum) break; // wave-uniform (quorum 0: never)
    }
    RT_MARK(8);
    const unsigned long long tri_left = tri_m; // (the planes and spheres of a ray that is cut short are still tested in this call)
    while (pl_m != 0ull) {
        RT_MARK(9);
        DBG_WAVE_TICK(15);
        DBG_ADD(13, 1); DBG_ADD(29, 1);
        const uint32_t rec = (uint32_t)__builtin_ctzll(pl_m);
        pl_m &= pl_m - 1ull;
        const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u), r2 = S.prim(4u * rec + 2u), r3 = S.prim(4u * rec + 3u);
        const float t = plane_t(o, d, v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), v3(r3.x, r3.y, r3.z));
        RT_FLAT_ACCEPT(t, rec)
        if (better & anyhit) pl_m = sp_m = 0ull;
    }
    RT_MARK(8);
    while (sp_m != 0ull) {
        RT_MARK(10);
        DBG_WAVE_TICK(28);
        DBG_ADD(13, 1); DBG_ADD(30, 1);
        const uint32_t rec = (uint32_t)__builtin_ctzll(sp_m);
        sp_m &= sp_m - 1ull;
        const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u);
        const float t = sphere_t(o, d, v3(r0.x, r0.y, r0.z), r1.y);
        RT_FLAT_ACCEPT(t, rec)
        if (better & anyhit) sp_m = 0ull;
    }
#undef RT_FLAT_ACCEPT
    RT_MARK(13);
    rem = (anyhit & (h.t < RT_INFINITY)) ? 0ull : tri_left; // any hit ends a shadow ray, whatever it still holds
    cur = rem != 0ull ? 1u : RT_END;
}

// ------------------------------------------------------------------ fixed-order (pre-order) traversal
// What a ray's result depends on: WHICH nodes it visits (every node all of whose ancestors' boxes it hits — the slab
// tests decide that, not the order), which primitives it therefore tests, and — only when two of them give the very same
// t — which one the reference meets first.  The last is a property of the ray's sign octant alone, so it is tabulated
// at upload: prim_rank[octant][record] = the record's position in the reference's near-child-first walk.  The winner is
// the lexicographic minimum of (t, rank) over the tested primitives, and that no longer depends on the order they are
// tested in.  So the walk can take the ONE order that needs no side table: the node array's own pre-order.
//   interior node, box hit    -> its first child           } both links are IN the node record (index words of its two
//   interior node, box missed -> its subtree's successor   } float4s), so the array can be laid out in any order:
//   leaf                      -> the next array element     the hottest nodes first, kept in LDS (rsrt_upload_scene)
// Two 16-byte loads per step instead of three (no escape link), no octant-dependent addressing, and for a scene of any
// size the top of the tree — where most box tests happen — is served by LDS instead of the vector L1, which is what
// bounds the all-global walk (profiles/r02_bvh_*); same boxes, same primitives, same winner as cast_ray_bvh
// (shader.wgsl:469-564).
// `ref_mem`: where the incumbent's record index lives when a resumed traversal has not loaded it (pool kernel: the
// slot's C_REF column); read only if a primitive ties with the incumbent.  h.ref == RT_REF_UNKNOWN until then.
#define RT_REF_UNKNOWN 0xffffffffu
RT_DEV uint32_t prim_rank_of(const DevScene &sc, uint32_t octant, uint32_t rec) { return sc.prim_rank[(size_t)octant * sc.n_prims + rec]; }

template <class View>
RT_DEV void trace_preorder(DBG_DECL const View &S, const DevScene &sc, V3 o, V3 d, bool prune, bool anyhit, uint32_t budget, uint32_t quorum, uint32_t &cur,
                           Hit &h, const uint32_t *ref_mem, uint32_t &work)
{
    static_assert(RT_LEAFQ >= 2 && RT_LEAFQ <= 4, "one byte of each 32-bit mask per held leaf; the leaf-loop vote rewinds to a second or later leaf");
    const V3 inv = rt_rcp3(d);
    const uint32_t octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    const uint32_t n_elems = sc.n_pnodes;
    uint32_t steps = 0;
    while (cur != RT_END && steps < budget) {
        DBG_WAVE_TICK(14);
        uint32_t qi[RT_LEAFQ], qe[RT_LEAFQ]; // held leaves: first record, and the walk element they were found at
        uint32_t nq = 0, all_m = 0, tri_m = 0, pl_m = 0;
#pragma unroll
        for (int j = 0; j < RT_LEAFQ; j++) qi[j] = qe[j] = 0u;
        // Lanes descend until they hold RT_LEAFQ leaves — but not for ever: rays need very different numbers of box steps
        // to get there (suzanne grid: 22 on average, 60 for the slowest lane of a wave), and every lane that is done
        // idles until the last one is.  So the wave takes a vote each trip and stops descending once fewer than
        // `quorum` percent of the lanes that started the round are still at it; whatever leaves are held are tested, and
        // the stragglers descend on in the next round TOGETHER with everyone else (the cursor is all the state there is).
        const uint32_t started = (uint32_t)__popcll(__ballot(true));
        while (cur != RT_END && nq < RT_LEAFQ) {
            DBG_WAVE_TICK(10);
            DBG_ADD(11, 1);
            steps++;
            float4 n0, n1;
            S.pnode_pair(cur, n0, n1);
            float t_0;
            bool inside = slab_test(n0, n1, o, inv, t_0);
            inside = inside & !(prune & (t_0 > h.t));
            const uint32_t w0 = as_u(n0.w), w1 = as_u(n1.w);
            const bool leaf = (w0 >> 31) != 0u;
            if (inside & leaf) {
                const uint32_t idx = w0 & 0x7fffffffu, len = w1 & 0xffffu, hi = w1 >> 16; // triangle mask | plane mask << 8
#pragma unroll
                for (int j = 0; j < RT_LEAFQ; j++) {
                    qi[j] = (nq == (uint32_t)j) ? idx : qi[j];
                    qe[j] = (nq == (uint32_t)j) ? cur : qe[j];
                }
                const uint32_t sh = 8u * nq;
                all_m |= ((1u << len) - 1u) << sh;
                tri_m |= (hi & 0xffu) << sh;
                pl_m |= (hi >> 8) << sh;
                steps += len;
                nq++;
            }
            // leaf: the next element; interior: first child when hit, the subtree's successor when missed (a jump
            // element has both links equal, so its never-valid box does not matter)
            const uint32_t next = leaf ? cur + 1u : (inside ? w0 : w1);
            cur = next >= n_elems ? RT_END : next;
            if ((uint32_t)__popcll(__ballot((cur != RT_END) & (nq < RT_LEAFQ))) * 100u < started * quorum) break; // wave-uniform
        }
        uint32_t sp_m = all_m & ~(tri_m | pl_m);
        // equal t: the record the reference meets first wins (rare: coincident geometry)
#define RT_PRE_ACCEPT(t, rec)                                                                                   \
        bool better = ((t) >= 0.0f) & ((t) < h.t);                                                                  \
        if (((t) == h.t) & (h.t < RT_INFINITY)) {                                                                   \
            if (h.ref == RT_REF_UNKNOWN) h.ref = *ref_mem;                                                          \
            better = prim_rank_of(sc, octant, (rec)) < prim_rank_of(sc, octant, h.ref);                            \
        }                                                                                                           \
        h.t = better ? (t) : h.t;                                                                                   \
        h.ref = better ? (rec) : h.ref;
        // Two triangles per trip: the walk is bound by the latency of these dependent gathers, not by instruction issue,
        // so both records are requested together and tested one after the other (the order is free: ties go by rank).
        // The same vote ends the triangle loop: lanes hold between none and twenty triangles, and the few with long
        // lists would keep the whole wave (29 % of the lanes busy, measured).  A lane that is cut short needs no state
        // saved: it REWINDS its cursor to the walk element of its first leaf with untested primitives and finds that
        // leaf (and the ones after it) again when it is next scheduled; a triangle tested twice gives the same t and
        // loses the tie against itself.
        const uint32_t tri_started = (uint32_t)__popcll(__ballot(tri_m != 0u));
        while (tri_m != 0u) {
            DBG_WAVE_TICK(12);
            DBG_ADD(13, 1);
            const uint32_t pa = take_lowest(tri_m);
            const bool two = tri_m != 0u;
            const uint32_t pb = two ? take_lowest(tri_m) : pa;
            uint32_t base_a = qi[0], base_b = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) {
                base_a = ((pa >> 3) == (uint32_t)j) ? qi[j] : base_a;
                base_b = ((pb >> 3) == (uint32_t)j) ? qi[j] : base_b;
            }
            const uint32_t rec_a = base_a + (pa & 7u), rec_b = base_b + (pb & 7u);
            const float4 a0 = S.prim(4u * rec_a), a1 = S.prim(4u * rec_a + 1u), a2 = S.prim(4u * rec_a + 2u);
            const float4 b0 = S.prim(4u * rec_b), b1 = S.prim(4u * rec_b + 1u), b2 = S.prim(4u * rec_b + 2u);
            float u, v;
            bool stop = false;
            {
                const float t = triangle_t(o, d, v3(a0.x, a0.y, a0.z), v3(a1.x, a1.y, a1.z), v3(a2.x, a2.y, a2.z), u, v);
                RT_PRE_ACCEPT(t, rec_a)
                stop = better & anyhit;
            }
            if (two & !stop) {
                DBG_ADD(13, 1);
                const float t = triangle_t(o, d, v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z), v3(b2.x, b2.y, b2.z), u, v);
                RT_PRE_ACCEPT(t, rec_b)
                stop = better & anyhit;
            }
            if (stop) { cur = RT_END; tri_m = pl_m = sp_m = 0u; }
            // (wave-uniform.)  Never before every lane is through the triangles of its FIRST leaf: each round then completes
            // at least one leaf per lane, so a ray that is cut short again and again still advances
            if (__ballot((tri_m & 0xffu) != 0u) == 0ull && (uint32_t)__popcll(__ballot(tri_m != 0u)) * 100u < tri_started * quorum) break;
        }
        if (tri_m != 0u) { // cut short: back to the first leaf with untested triangles (>= 1); the planes / spheres of the leaves before it are still tested below
            const uint32_t first = (uint32_t)__builtin_ctz(tri_m) >> 3;
            uint32_t back = qe[1];
#pragma unroll
            for (int j = 2; j < RT_LEAFQ; j++) back = (first == (uint32_t)j) ? qe[j] : back;
            cur = back;
            const uint32_t keep = (1u << (8u * first)) - 1u;
            tri_m = 0u;
            pl_m &= keep;
            sp_m &= keep;
        }
        while (pl_m != 0u) {
            DBG_WAVE_TICK(15);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(pl_m);
            uint32_t base = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) base = ((p >> 3) == (uint32_t)j) ? qi[j] : base;
            const uint32_t rec = base + (p & 7u);
            const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u), r2 = S.prim(4u * rec + 2u), r3 = S.prim(4u * rec + 3u);
            const float t = plane_t(o, d, v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z), v3(r3.x, r3.y, r3.z));
            RT_PRE_ACCEPT(t, rec)
            if (better & anyhit) { cur = RT_END; pl_m = sp_m = 0u; }
        }
        while (sp_m != 0u) {
            DBG_WAVE_TICK(28);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(sp_m);
            uint32_t base = qi[0];
#pragma unroll
            for (int j = 1; j < RT_LEAFQ; j++) base = ((p >> 3) == (uint32_t)j) ? qi[j] : base;
            const uint32_t rec = base + (p & 7u);
            const float4 r0 = S.prim(4u * rec), r1 = S.prim(4u * rec + 1u);
            const float t = sphere_t(o, d, v3(r0.x, r0.y, r0.z), r1.y);
            RT_PRE_ACCEPT(t, rec)
            if (better & anyhit) { cur = RT_END; sp_m = 0u; }
        }
#undef RT_PRE_ACCEPT
    }
    work += steps;
}

// ------------------------------------------------------------------ wide walk
// The fixed-order walk above takes one dependent 32-byte fetch per box (38 a ray on the 15 k-triangle scene) and spends half
// of its wave time waiting for them.  What the result depends on — the slab tests decide which leaves are tested, the ranks
// decide equal t — leaves the SHAPE of the walk free, so the binary tree is collapsed at upload into 4-wide nodes (a node's
// children are the binary node's children, the largest-area interior one replaced by its own two until there are four):
// one fetch brings the EXACT boxes of four children, a quarter of the round trips for the same number of box tests
// (tools/walk_sim.py).  Skipping the binary nodes in between is exact for the same reason the flat loop is: boxes nest and
// f32 rounding is monotone, so with a finite 1/d a leaf's own box is hit only if every ancestor's is
// (tests/test_box_containment.py); rays with a non-finite 1/d take the fixed-order walk, scenes whose boxes do not nest
// never get here (rsrt_upload_scene, wide_ok).
// Order: any (ties go by rank), so depth-first with the pending children of a level as ONE word — first child's node index
// << 4 | 4-bit mask, a node's interior children being consecutive — and a stack of such words: the innermost RT_WSTACK in registers, what
// a deeper tree pushes beyond them in the slot's own columns of the arena (RT_WSPILL more: the bottom of the stack, touched again only when
// the walk comes back up to the top of the tree; wide depth <= RT_WSTACK + RT_WSPILL + 1, i.e. any scene that fits the device).  Leaves: the records of a node's leaf children are contiguous
// (the upload permutes whole leaves), so what a lane holds for the primitive loops is one base index + three 32-bit masks.
// A round = lanes visit nodes until they hold primitives (or a vote ends the wait) -> typed primitive loops.  A ray that is
// not done when the wave stops (budget, or too few lanes left) parks its stack in the slot's cold columns (`wmem`).
#define RT_WSTACK 8
#define RT_WSPILL 16 // stack words beyond the registers, in memory
// a ray's walk state in memory (`wmem`, words `wstride` apart): [0] next node, [1] pending group, [2 .. 9] the register stack (written when a ray
// is parked), [10] how many words have overflowed, [11 ..] those words, oldest first (written when they overflow, parked or not)
#define RT_WSTATE_WORDS (2u + RT_WSTACK + 1u + RT_WSPILL)
#define RT_WIDE_EMPTY 0xffffffffu
struct WalkState {
    uint32_t cur;  // node to visit next, RT_END: none
    uint32_t grp;  // pending children of the current level: first child's index << 4 | mask
    uint32_t s0, s1, s2, s3, s4, s5, s6, s7; // pending children of the levels above, s0 the innermost; 0 = none (a shift register: named
                                             // scalars and whole-stack moves, because an indexed array would live in scratch memory)
    uint32_t spill; // words that have left the registers at the s7 end for memory (> 0 only while all eight registers are in use)
    RT_DEV void fresh() { cur = 0u; grp = 0u; s0 = s1 = s2 = s3 = s4 = s5 = s6 = s7 = 0u; spill = 0u; }
};
static_assert(RT_WSTACK == 8, "WalkState names its eight stack words");
// DEEP: the tree has more wide levels than RT_WSTACK + 1, the stack may overflow (a kernel variant of its own, TRAV 5: the walk of a
// shallow tree carries neither the test nor the pointer)
template <bool DEEP>
RT_DEV void wstack_push(WalkState &w, uint32_t v, uint32_t *m, uint32_t stride)
{
    if (DEEP && w.s7 != 0u) { // the registers are full: the oldest word goes to memory (trees deeper than RT_WSTACK + 1 wide levels only)
        m[(2u + RT_WSTACK + 1u + w.spill) * stride] = w.s7;
        w.spill++;
    }
    w.s7 = w.s6; w.s6 = w.s5; w.s5 = w.s4; w.s4 = w.s3; w.s3 = w.s2; w.s2 = w.s1; w.s1 = w.s0; w.s0 = v;
}
template <bool DEEP>
RT_DEV uint32_t wstack_pop(WalkState &w, const uint32_t *m, uint32_t stride) // 0 when the stack is empty
{
    const uint32_t v = w.s0;
    w.s0 = w.s1; w.s1 = w.s2; w.s2 = w.s3; w.s3 = w.s4; w.s4 = w.s5; w.s5 = w.s6; w.s6 = w.s7; w.s7 = 0u;
    if (DEEP && w.spill != 0u) { // (then the registers were full: the youngest word in memory takes the place that came free)
        w.spill--;
#ifdef RT_MUTATE_DROP_SPILL // (mutation check of tests/test_gpu_parity.py::test_wide_walk_stack_overflow_on_a_chain_tree: the overflowed words are lost)
        w.s7 = 0u;
#else
        w.s7 = m[(2u + RT_WSTACK + 1u + w.spill) * stride];
#endif
    }
    return v;
}
template <bool DEEP>
RT_DEV void wstate_load(WalkState &w, const uint32_t *m, uint32_t stride)
{
    w.cur = m[0]; w.grp = m[stride];
    w.s0 = m[2u * stride]; w.s1 = m[3u * stride]; w.s2 = m[4u * stride]; w.s3 = m[5u * stride];
    w.s4 = m[6u * stride]; w.s5 = m[7u * stride]; w.s6 = m[8u * stride]; w.s7 = m[9u * stride];
    w.spill = DEEP ? m[10u * stride] : 0u;
}
template <bool DEEP>
RT_DEV void wstate_store(const WalkState &w, uint32_t *m, uint32_t stride)
{
    m[0] = w.cur; m[stride] = w.grp;
    m[2u * stride] = w.s0; m[3u * stride] = w.s1; m[4u * stride] = w.s2; m[5u * stride] = w.s3;
    m[6u * stride] = w.s4; m[7u * stride] = w.s5; m[8u * stride] = w.s6; m[9u * stride] = w.s7;
    if (DEEP) m[10u * stride] = w.spill;
}

#ifndef RT_WIDE_HOLD
#define RT_WIDE_HOLD 6 // triangles a lane wants to hold before it stops visiting nodes (a shadow ray: any; its first hit ends it)
#endif
// What a lane carries for ONE ray of the wide walk: the ray, its stack, and what it holds for the triangle loop — a 64-record WINDOW of
// the (permuted) record array (the records of sibling nodes are neighbours there, so the leaves of several nodes usually share one
// window) plus, once a node's triangles fall outside it, that node's group as overflow; then the lane stops visiting nodes and waits for
// the wave's triangle loop.  A lane without a ray has w.cur == RT_END and holds nothing: it falls through both loops.
struct WideRay {
    V3 o, d, inv;
    bool anyhit;
    uint32_t octant;
    WalkState w;
    const uint32_t *ref_mem; // the slot's cold column with the record of an earlier call's best hit (read only on an equal t)
    uint32_t *wmem;          // the slot's walk state in memory (RT_WSTATE_WORDS words, wstride apart): the stack's overflow, and where a ray is parked
    uint32_t wstride;
    uint32_t win_base, ovf_base, ovf_tri;
    unsigned long long tri_m;
    RT_DEV void start(V3 o_, V3 d_, V3 inv_, bool anyhit_, const uint32_t *ref_mem_, uint32_t *wmem_, uint32_t wstride_)
    {
        o = o_; d = d_; inv = inv_; anyhit = anyhit_; ref_mem = ref_mem_; wmem = wmem_; wstride = wstride_;
        octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
        win_base = ovf_base = ovf_tri = 0u;
        tri_m = 0ull;
    }
    RT_DEV void idle() // a lane without a ray
    {
        w.cur = RT_END; w.grp = 0u;
        win_base = ovf_base = ovf_tri = 0u;
        tri_m = 0ull;
        anyhit = false; octant = 0u; ref_mem = nullptr; wmem = nullptr; wstride = 0u;
        o = d = inv = v3(0.0f, 0.0f, 0.0f);
    }
    RT_DEV bool holds() const { return tri_m != 0ull || ovf_tri != 0u; }
    RT_DEV bool done() const { return w.cur == RT_END && !holds(); }
    RT_DEV bool looking() const { return (w.cur != RT_END) & (ovf_tri == 0u) & ((uint32_t)__popcll(tri_m) < (anyhit ? 1u : (uint32_t)RT_WIDE_HOLD)); }
};
// equal t: the record the reference meets first wins (rare: coincident geometry)
#define RT_WIDE_ACCEPT(t, rec)                                                                                   \
    bool better = ((t) >= 0.0f) & ((t) < h.t);                                                                      \
    if (((t) == h.t) & (h.t < RT_INFINITY)) {                                                                       \
        if (h.ref == RT_REF_UNKNOWN) h.ref = *r.ref_mem;                                                            \
        better = prim_rank_of(sc, r.octant, (rec)) < prim_rank_of(sc, r.octant, h.ref);                            \
    }                                                                                                               \
    h.t = better ? (t) : h.t;                                                                                       \
    h.ref = better ? (rec) : h.ref;

// One plane or sphere record of the wide walk (record `rec`; is_plane from the node's type mask)
#define RT_WIDE_TEST_OTHER(rec, is_plane)                                                                                                             \
    float4 q[4];                                                                                                                                         \
    S.template prim_rec<4>((rec), q);                                                                                                                     \
    const float t = (is_plane) ? plane_t(o, d, v3(q[0].x, q[0].y, q[0].z), v3(q[1].x, q[1].y, q[1].z), v3(q[2].x, q[2].y, q[2].z), v3(q[3].x, q[3].y, q[3].z)) \
                               : sphere_t(o, d, v3(q[0].x, q[0].y, q[0].z), q[1].y);                                                                    \
    RT_WIDE_ACCEPT(t, (rec))

// One round's node visits: every lane visits nodes until it holds enough triangles (a shadow ray: any); the wave stops waiting once
// fewer than `quorum` percent of the `n_started` lanes that hold a ray are still looking.
template <bool DEEP, class View>
RT_DEV void wide_nodes(DBG_DECL const View &S, const DevScene &sc, WideRay &r, Hit &h, uint32_t quorum, uint32_t n_started, uint32_t &steps)
{
    const V3 o = r.o, d = r.d, inv = r.inv;
    WalkState &w = r.w;
    while (r.looking()) {
        DBG_WAVE_TICK(10);
        DBG_ADD(11, 1);
        steps++;
        float4 n[8];
        S.wnode(w.cur, n);
        // the eight .w words are the node's, not the slots': [0] first interior child's node index | interior-slot mask << 26,
        // [1] first record of the node's leaf children, [2] / [3] which of the 32 records from there are triangles / planes,
        // [4 + k] the records of slot k (0 unless it is a leaf) — so a hit turns into masks with a select and an OR
        uint32_t hm = 0u, lm = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 n0 = n[2 * k], n1 = n[2 * k + 1];
            const float ax = (n0.x - o.x) * inv.x, bx = (n1.x - o.x) * inv.x;
            const float ay = (n0.y - o.y) * inv.y, by = (n1.y - o.y) * inv.y;
            const float az = (n0.z - o.z) * inv.z, bz = (n1.z - o.z) * inv.z;
            const float t_0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz)), 0.0f);
            const float t_1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz)), RT_INFINITY);
            const bool hit = !(t_0 > t_1);
            hm |= hit ? (1u << k) : 0u;
            lm |= hit ? as_u(n[4 + k].w) : 0u;
        }
        const uint32_t wa = as_u(n[0].w);
        const uint32_t im = hm & (wa >> 26); // (an empty slot is in neither mask: its "hit" goes nowhere)
        const uint32_t rec_base = as_u(n[1].w), tri32 = as_u(n[2].w), pl32 = as_u(n[3].w);
        steps += (uint32_t)__popc(lm);
        bool stop = false;
        // planes and spheres (rare inside a mesh's tree): tested here and now, one at a time (held to the end of the round's node visits and
        // tested together: 15 k-triangle scene -2 %, suzanne +1 %, profiles/r03_walk_bounds.txt — not kept)
        uint32_t oth = lm & ~tri32;
        while (oth != 0u) {
            DBG_WAVE_TICK(15);
            DBG_ADD(13, 1);
            const uint32_t p = take_lowest(oth);
            RT_WIDE_TEST_OTHER(rec_base + p, (pl32 >> p) & 1u)
            if (better & r.anyhit) { stop = true; oth = 0u; }
        }
        // triangles: into the window if they fall inside it, else this node's group is the overflow
        const uint32_t ltri = lm & tri32;
        if (ltri != 0u) {
            const uint32_t shift = rec_base - r.win_base; // (wraps to a huge number when the group lies below the window)
            if (r.tri_m == 0ull) { r.win_base = rec_base; r.tri_m = ltri; }
            else if (shift <= 32u) r.tri_m |= (unsigned long long)ltri << shift;
            else { r.ovf_base = rec_base; r.ovf_tri = ltri; }
        }
        // where next: the hit interior children become the pending group of a new level (the old one goes on the stack)
        if (im != 0u) {
            if ((w.grp & 15u) != 0u) wstack_push<DEEP>(w, w.grp, r.wmem, r.wstride);
            w.grp = ((wa & 0x3ffffffu) << 4) | im; // interior children are consecutive: slot k is node (first child) + k
        } else if ((w.grp & 15u) == 0u) {
            w.grp = wstack_pop<DEEP>(w, r.wmem, r.wstride);
        }
        if ((w.grp & 15u) != 0u) {
            w.cur = (w.grp >> 4) + (uint32_t)__builtin_ctz(w.grp);
            w.grp &= w.grp - 1u;
        } else {
            w.cur = RT_END;
        }
        if (stop) { w.cur = RT_END; r.tri_m = 0ull; r.ovf_tri = 0u; }
        if ((uint32_t)__popcll(__ballot(r.looking())) * 100u < n_started * quorum) break; // wave-uniform
    }
}

// One round's triangle loop: the triangles of the window, two records in flight per trip.  Lanes hold very different numbers of them, so
// the wave votes here too: once fewer than `quorum` percent of the lanes that entered still hold triangles the loop ends, and what a lane
// has left simply stays in its window for the next round (nothing to save, nothing tested twice) — unless `all`: the wave is about to
// stop, everything held is tested.  Afterwards the overflow group, if any, opens the next window.
template <class View>
RT_DEV void wide_tris(DBG_DECL const View &S, const DevScene &sc, WideRay &r, Hit &h, uint32_t quorum, bool all)
{
    const V3 o = r.o, d = r.d;
    const uint32_t tri_started = (uint32_t)__popcll(__ballot(r.tri_m != 0ull));
    while (r.tri_m != 0ull) {
        DBG_WAVE_TICK(12);
        DBG_ADD(13, 1);
        const uint32_t rec_a = r.win_base + (uint32_t)__builtin_ctzll(r.tri_m);
        r.tri_m &= r.tri_m - 1ull;
        const bool two = r.tri_m != 0ull;
        const uint32_t rec_b = two ? r.win_base + (uint32_t)__builtin_ctzll(r.tri_m) : rec_a;
        r.tri_m &= r.tri_m - 1ull; // (0 & anything = 0)
        float4 ra[3], rb[3];
        S.template prim_rec<3>(rec_a, ra);
        S.template prim_rec<3>(rec_b, rb);
        float u, v;
        bool stop = false;
        {
            const float t = triangle_t(o, d, v3(ra[0].x, ra[0].y, ra[0].z), v3(ra[1].x, ra[1].y, ra[1].z), v3(ra[2].x, ra[2].y, ra[2].z), u, v);
            RT_WIDE_ACCEPT(t, rec_a)
            stop = better & r.anyhit;
        }
        if (two & !stop) {
            DBG_ADD(13, 1);
            const float t = triangle_t(o, d, v3(rb[0].x, rb[0].y, rb[0].z), v3(rb[1].x, rb[1].y, rb[1].z), v3(rb[2].x, rb[2].y, rb[2].z), u, v);
            RT_WIDE_ACCEPT(t, rec_b)
            stop = better & r.anyhit;
        }
        if (stop) { r.w.cur = RT_END; r.tri_m = 0ull; r.ovf_tri = 0u; }
        if (!all && (uint32_t)__popcll(__ballot(r.tri_m != 0ull)) * 100u < tri_started * quorum) break; // wave-uniform
    }
    if (r.ovf_tri != 0u && r.tri_m == 0ull) { r.win_base = r.ovf_base; r.tri_m = r.ovf_tri; r.ovf_tri = 0u; } // the overflow group opens the next window
}
#undef RT_WIDE_TEST_OTHER
#undef RT_WIDE_ACCEPT

// One ray per lane, no refill (GEN's fused first trace, the probe): rounds of node visits and triangle tests until the ray is done, the
// round budget is spent, or fewer than `stop_quorum` percent of the lanes that came in still hold a ray — then what is held is tested
// and the lane returns with its stack in `w` (w.cur != RT_END) for the caller to park.
template <bool DEEP, class View>
RT_DEV void trace_wide(DBG_DECL const View &S, const DevScene &sc, V3 o, V3 d, V3 inv, bool anyhit, uint32_t budget, uint32_t quorum, uint32_t stop_quorum,
                       WalkState &w, Hit &h, const uint32_t *ref_mem, uint32_t *wmem, uint32_t wstride, uint32_t &work)
{
    WideRay r;
    r.start(o, d, inv, anyhit, ref_mem, wmem, wstride);
    r.w = w;
    uint32_t steps = 0, rounds = 0;
    const uint32_t started = (uint32_t)__popcll(__ballot(r.w.cur != RT_END));
    bool stopping = false;
    for (;;) {
        DBG_WAVE_TICK(14);
        if (!stopping) wide_nodes<DEEP>(DBG_ARG S, sc, r, h, quorum, (uint32_t)__popcll(__ballot(true)), steps);
        wide_tris(DBG_ARG S, sc, r, h, quorum, stopping);
        // ---- go on?  A lane leaves when it is done, or when the wave stops and it holds nothing (what it holds is tested first)
        if (stopping) {
            if (r.tri_m == 0ull) break;
            continue;
        }
        rounds++; // (wave-uniform: every lane still here has run the same number of rounds)
        stopping = rounds >= budget || (uint32_t)__popcll(__ballot((r.w.cur != RT_END) | (r.tri_m != 0ull))) * 100u < started * stop_quorum;
        if (r.tri_m == 0ull && (stopping || r.w.cur == RT_END)) break;
    }
    w = r.w;
    work += steps;
}

// ------------------------------------------------------------------ which traversal runs
// TRAV: 0 trace_threaded (any BVH), 1 trace_threaded_typed (no leaf longer than 8 primitives), 2 trace_flat (<= 64
// records, nested boxes), 3 trace_preorder (no leaf longer than 8 primitives), 4 trace_wide (nested boxes, leaves that
// share no record, <= 8 primitives a leaf, wide tree no deeper than RT_WSTACK + RT_WSPILL + 1).  One function so that the production
// kernel's TRACE stage and the ray-query probe (rsrt_cast_rays) run the very same code.  `cur` is the traversal cursor
// (0 = start at the root, RT_END = done), `h` the best hit so far; the tree walks stop after ~`budget` steps and are
// resumed by calling again.  `ref_mem`: see trace_preorder.  `work` += box steps + primitive tests of the tree walks
// (the flat loop, whose work per ray is fixed by the scene, adds nothing).
// TRAV 4 / 5 (trace_wide; 5 = a tree deeper than the walk's register stack, whose bottom then lives in `wmem`): `cur` is 0 for a fresh ray, RT_END when done, and otherwise says that the walk's stack waits in `wmem`
// (this slot's cold columns, `wstride` dwords apart: RT_WSTATE_WORDS words, see WalkState) — 1.
// Rays with a non-finite 1/d take the fixed-order walk, for which `cur` is that walk's cursor; which of the two a ray takes
// is a function of the ray alone, so a resumed ray reads its `cur` the way it was written.
template <int TRAV, class View>
RT_DEV void trace_dispatch(DBG_DECL const View &S, const DevScene &sc, V3 o, V3 d, bool prune, bool anyhit, uint32_t budget, uint32_t quorum, uint32_t &cur,
                           Hit &h, const uint32_t *ref_mem, uint32_t &work, unsigned long long &flat_rem, uint32_t *wmem = nullptr, uint32_t wstride = 0,
                           uint32_t stop_quorum = 0, bool coherent = false)
{
    if (TRAV == 4 || TRAV == 5) { // (5: a tree deeper than the register stack)
        constexpr bool DEEP = TRAV == 5;
        const V3 inv = rt_rcp3(d);
        const float finite = ((inv.x + inv.y) + inv.z) * 0.0f + ((o.x + o.y) + o.z) * 0.0f; // (as for the flat loop below)
        if (finite == 0.0f) {
            WalkState w;
            w.fresh();
            if (cur != 0u) wstate_load<DEEP>(w, wmem, wstride); // parked by an earlier call
            trace_wide<DEEP>(DBG_ARG S, sc, o, d, inv, anyhit, budget, quorum, stop_quorum, w, h, ref_mem, wmem, wstride, work);
            if (w.cur == RT_END) {
                cur = RT_END;
            } else {
                wstate_store<DEEP>(w, wmem, wstride);
                cur = 1u;
            }
        } else {
            trace_preorder(DBG_ARG S, sc, o, d, prune, anyhit, 0xffffffffu, 0u, cur, h, ref_mem, work);
        }
    } else if (TRAV == 2) {
        const V3 inv = rt_rcp3(d);
        // 0 * x is NaN exactly when x is infinite or NaN (an overflowing sum only sends a ray the long way round)
        const float finite = ((inv.x + inv.y) + inv.z) * 0.0f + ((o.x + o.y) + o.z) * 0.0f;
        if (finite == 0.0f) {
            trace_flat(DBG_ARG S, sc, o, d, inv, anyhit, quorum, cur, flat_rem, h, coherent);
        } else {
            RT_MARK_COLD();
            trace_threaded(DBG_ARG S, sc.n_nodes, o, d, prune, anyhit, 0xffffffffu, cur, h, work);
        }
    } else if (TRAV == 3) {
        trace_preorder(DBG_ARG S, sc, o, d, prune, anyhit, budget, quorum, cur, h, ref_mem, work);
    } else if (TRAV == 1) {
        trace_threaded_typed(DBG_ARG S, sc.n_nodes, o, d, prune, anyhit, budget, cur, h, work);
    } else {
        trace_threaded(DBG_ARG S, sc.n_nodes, o, d, prune, anyhit, budget, cur, h, work);
    }
}
