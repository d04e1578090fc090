// rt_bvh_device.h — build_bvh (reference src/bvh.rs:13-337) ON THE DEVICE, bit-identical to the host builder
// (csrc/host/bvh_build.cpp, rsrt_build_bvh).  SURVEY.md §8(f)3.
//
// What makes a parallel build of a binned-SAH tree reproducible to the bit:
//  * every quantity the reference computes per node is order-independent or integer — the node's box and the centroid
//    box are min / max reductions (exact in any order), the 12 bucket counts are integers, the bucket boxes again min / max,
//    and the eleven candidate costs are evaluated by ONE thread with the host's own f32 expression, first minimum winning;
//  * the one order-DEPENDENT step, the reference's in-place two-pointer partition (`split`/`end` swap loop, :304-315, not
//    stable: leaf order depends on it), is a fixed permutation of the node's range given only each item's class, and it
//    has a closed form (checked against the sequential loop on random class strings, tests/test_bvh_device.py):
//    with L = number of LEFT items, holes h_1 < h_2 < ... the RIGHT items at positions < L, and b_1 > b_2 > ... the LEFT
//    items at positions >= L (as many as holes), b_0 = n:
//        LEFT  at i < L   stays            | RIGHT at h_k       goes to b_(k-1) - 1
//        LEFT  at b_k     goes to h_k      | RIGHT at p >= L    goes to p - 1, except p == L, which goes to b_m - 1
//    so one prefix count of the LEFT flags gives every item its place.
// Shape: level-synchronous.  One workgroup per open node of the level (reductions, buckets, scan, scatter into the other
// copy of the items).  The top of a big tree is ONE workgroup walking a long range, so the passes are written for that case:
// min / max in registers, combined across a wave by shuffles and across waves by one LDS atomic each (the first version
// sent every item's twelve keys to the same twelve LDS words: 256 threads taking turns); bucket tables in sixteen private
// copies; the LEFT-flag prefix from wave ballots, one barrier per 256 items (was a 17-barrier Hillis-Steele scan); the nodes are numbered breadth-first as they are allocated, and three small passes (subtree sizes
// bottom-up, positions top-down, emit) turn that into the reference's pre-order array (first child = parent + 1,
// second child's index in the parent, :155-178).  A leaf's primitives are its range of the final item order, so
// `primitives` is simply the items' (type, index) at the end.
#pragma once

#define RT_BVH_BLOCK 256
#define RT_BVH_BUCKETS 12u
#define RT_BVH_MAX_LEAF 5u

struct BvhItems {
    float4 *bmin; // xyz, .w = type bits
    float4 *bmax; // xyz, .w = index bits
    float4 *cen;  // xyz
};
struct BvhNodeB { // breadth-first build node
    float bmin[3], bmax[3];
    uint32_t begin, count; // item range
    uint32_t left, right;  // build-node ids of the children (0: leaf)
    uint32_t axis, size, pos, pad;
};
struct BvhTask { uint32_t begin, end, node; };

// monotone float <-> uint key, for min / max reductions with integer atomics in LDS.  (-0 orders below +0 here, so a bound that is a zero comes
// out as -0 from a min and +0 from a max whatever the operand order; the host's std::fmin / fmax — like the reference's f32::min / max — may return
// either zero.  The one exception to "the same tree bit for bit": the SIGN of a zero bound; same values, and no slab test can tell them apart.
// tests/test_bvh_device.py builds a scene with +0 / -0 vertices on a symmetry plane.)
RT_DEV uint32_t bvh_key(float f) { const uint32_t b = as_u(f); return (b >> 31) ? ~b : (b | 0x80000000u); }
RT_DEV float bvh_unkey(uint32_t k) { return as_f((k >> 31) ? (k & 0x7fffffffu) : ~k); }

// Sphere::bounds (scene.rs:173-180), Plane::bounds (:203-207), HittableTriangle::bounds (mesh.rs:143-147); Bounds3::center
__global__ void rt_bvh_items_kernel(const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                                    const rsrt_vec3 *vertices, const rsrt_triangle *triangles, uint32_t n_triangles, BvhItems it)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = n_spheres + n_planes + n_triangles;
    if (i >= n) return;
    const float FMAX = 3.40282347e+38f;
    float mn[3] = {FMAX, FMAX, FMAX}, mx[3] = {-FMAX, -FMAX, -FMAX};
    uint32_t type, index;
    auto grow = [&](float x, float y, float z) {
        mn[0] = __builtin_fminf(mn[0], x); mn[1] = __builtin_fminf(mn[1], y); mn[2] = __builtin_fminf(mn[2], z);
        mx[0] = __builtin_fmaxf(mx[0], x); mx[1] = __builtin_fmaxf(mx[1], y); mx[2] = __builtin_fmaxf(mx[2], z);
    };
    if (i < n_spheres) {
        type = 0; index = i;
        const rsrt_sphere &s = spheres[i];
        for (int k = 0; k < 3; k++) { mn[k] = s.pos[k] - s.radius; mx[k] = s.pos[k] + s.radius; }
    } else if (i < n_spheres + n_planes) {
        type = 1; index = i - n_spheres;
        const rsrt_plane_desc &p = planes[index];
        grow(p.pos[0], p.pos[1], p.pos[2]);
        grow((p.pos[0] + p.forward[0]) + p.right[0], (p.pos[1] + p.forward[1]) + p.right[1], (p.pos[2] + p.forward[2]) + p.right[2]);
    } else {
        type = 2; index = i - n_spheres - n_planes;
        const rsrt_triangle &t = triangles[index];
        for (uint32_t v : {t.vertex_0, t.vertex_1, t.vertex_2}) grow(vertices[v].v[0], vertices[v].v[1], vertices[v].v[2]);
    }
    it.bmin[i] = float4{mn[0], mn[1], mn[2], as_f(type)};
    it.bmax[i] = float4{mx[0], mx[1], mx[2], as_f(index)};
    it.cen[i] = float4{mn[0] * 0.5f + mx[0] * 0.5f, mn[1] * 0.5f + mx[1] * 0.5f, mn[2] * 0.5f + mx[2] * 0.5f, 0.0f};
}

RT_DEV uint32_t bvh_bucket_of(float c, float lo, float hi) // :258-269
{
    const float f = 12.0f * ((c - lo) / (hi - lo));
    const uint32_t b = f > 0.0f ? (uint32_t)f : 0u; // Rust `as usize`: saturating, NaN -> 0
    return b >= RT_BVH_BUCKETS ? RT_BVH_BUCKETS - 1u : b;
}
RT_DEV float bvh_area(const float *mn, const float *mx) // Bounds3::surface_area, scene.rs:125-128
{
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

// One open node per workgroup.  counters: [0] build nodes allocated, [1] tasks of the next level, [2] error flag
__global__ __launch_bounds__(RT_BVH_BLOCK) void rt_bvh_level_kernel(const BvhTask *tasks, uint32_t n_tasks, BvhItems src, BvhItems dst, BvhNodeB *nodes,
                                                                     BvhTask *next_tasks, uint32_t *counters, uint32_t *scratch_pre, uint32_t *scratch_holes,
                                                                     uint32_t *scratch_backs)
{
    if (blockIdx.x >= n_tasks) return;
    const BvhTask t = tasks[blockIdx.x];
    const uint32_t n = t.end - t.begin, tid = threadIdx.x;
    constexpr uint32_t kPriv = 16u; // private copies of the bucket tables, by tid >> 4
    __shared__ uint32_t s_box[6], s_cbox[6];             // node box / centroid box as keys
    __shared__ uint32_t s_cnt[RT_BVH_BUCKETS], s_bbox[RT_BVH_BUCKETS][6];
    __shared__ uint32_t p_cnt[kPriv][RT_BVH_BUCKETS], p_bbox[kPriv][RT_BVH_BUCKETS][6];
    __shared__ uint32_t s_wsum[2][RT_BVH_BLOCK / 64];
    __shared__ float pre_mn[RT_BVH_BUCKETS][3], pre_mx[RT_BVH_BUCKETS][3], suf_mn[RT_BVH_BUCKETS][3], suf_mx[RT_BVH_BUCKETS][3]; // (thread 0's sweeps)
    __shared__ uint32_t pre_n[RT_BVH_BUCKETS], suf_n[RT_BVH_BUCKETS];
    __shared__ uint32_t s_misc[8]; // [0] leaf, [1] axis, [2] lo bits, [3] hi bits, [4] best bucket, [5] L, [6] running prefix, [7] pre[L]
    if (tid < 6) { s_box[tid] = tid < 3 ? 0xffffffffu : 0u; s_cbox[tid] = tid < 3 ? 0xffffffffu : 0u; }
    for (uint32_t j = tid; j < kPriv * RT_BVH_BUCKETS; j += RT_BVH_BLOCK) {
        p_cnt[j / RT_BVH_BUCKETS][j % RT_BVH_BUCKETS] = 0u;
        for (int k = 0; k < 6; k++) p_bbox[j / RT_BVH_BUCKETS][j % RT_BVH_BUCKETS][k] = k < 3 ? 0xffffffffu : 0u;
    }
    __syncthreads();
    // ---- node box, centroid box (:222-234): keys in registers, wave shuffles, one LDS atomic per wave and word
    {
        uint32_t lo[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}; // min keys: box min xyz, centroid xyz
        uint32_t hi[6] = {0u, 0u, 0u, 0u, 0u, 0u};                                                         // max keys: box max xyz, centroid xyz
        for (uint32_t i = tid; i < n; i += RT_BVH_BLOCK) {
            const float4 a = src.bmin[t.begin + i], b = src.bmax[t.begin + i], c = src.cen[t.begin + i];
            const uint32_t ka[3] = {bvh_key(a.x), bvh_key(a.y), bvh_key(a.z)}, kb[3] = {bvh_key(b.x), bvh_key(b.y), bvh_key(b.z)}, kc[3] = {bvh_key(c.x), bvh_key(c.y), bvh_key(c.z)};
            for (int k = 0; k < 3; k++) {
                lo[k] = min(lo[k], ka[k]); hi[k] = max(hi[k], kb[k]);
                lo[3 + k] = min(lo[3 + k], kc[k]); hi[3 + k] = max(hi[3 + k], kc[k]);
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 6; k++) {
                lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], off));
                hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], off));
            }
        if ((tid & 63u) == 0u)
            for (int k = 0; k < 3; k++) {
                atomicMin(&s_box[k], lo[k]); atomicMax(&s_box[3 + k], hi[k]);
                atomicMin(&s_cbox[k], lo[3 + k]); atomicMax(&s_cbox[3 + k], hi[3 + k]);
            }
    }
    __syncthreads();
    if (tid == 0) {
        BvhNodeB &nd = nodes[t.node];
        for (int k = 0; k < 3; k++) { nd.bmin[k] = bvh_unkey(s_box[k]); nd.bmax[k] = bvh_unkey(s_box[3 + k]); }
        nd.begin = t.begin; nd.count = n; nd.left = nd.right = 0u; nd.axis = 0u; nd.size = 1u; nd.pos = 0u;
        bool leaf = n <= RT_BVH_MAX_LEAF;
        uint32_t axis = 0;
        float lo = 0.0f, hi = 0.0f;
        if (!leaf) {
            const float dx = bvh_unkey(s_cbox[3]) - bvh_unkey(s_cbox[0]), dy = bvh_unkey(s_cbox[4]) - bvh_unkey(s_cbox[1]), dz = bvh_unkey(s_cbox[5]) - bvh_unkey(s_cbox[2]);
            axis = (dz > dx && dz > dy) ? 2u : (dy > dx ? 1u : 0u); // Bounds3::max_axis, scene.rs:113-122
            lo = bvh_unkey(s_cbox[axis]); hi = bvh_unkey(s_cbox[3 + axis]);
            leaf = lo == hi; // :241-244
        }
        s_misc[0] = leaf ? 1u : 0u; s_misc[1] = axis; s_misc[2] = as_u(lo); s_misc[3] = as_u(hi);
    }
    __syncthreads();
    if (s_misc[0]) return; // a leaf: its items are final where they are (both copies hold them)
    const uint32_t axis = s_misc[1];
    const float lo = as_f(s_misc[2]), hi = as_f(s_misc[3]);
    // ---- buckets (:258-276): sixteen private tables (neighbouring items tend to share a bucket), folded into one afterwards
    {
        const uint32_t pv = tid >> 4;
        for (uint32_t i = tid; i < n; i += RT_BVH_BLOCK) {
            const float4 a = src.bmin[t.begin + i], b = src.bmax[t.begin + i], c = src.cen[t.begin + i];
            const uint32_t k = bvh_bucket_of(axis == 0 ? c.x : (axis == 1 ? c.y : c.z), lo, hi);
            atomicAdd(&p_cnt[pv][k], 1u);
            atomicMin(&p_bbox[pv][k][0], bvh_key(a.x)); atomicMin(&p_bbox[pv][k][1], bvh_key(a.y)); atomicMin(&p_bbox[pv][k][2], bvh_key(a.z));
            atomicMax(&p_bbox[pv][k][3], bvh_key(b.x)); atomicMax(&p_bbox[pv][k][4], bvh_key(b.y)); atomicMax(&p_bbox[pv][k][5], bvh_key(b.z));
        }
        __syncthreads();
        if (tid < RT_BVH_BUCKETS * 7u) { // (bucket, word): count or one of six box keys
            const uint32_t bk = tid / 7u, wd = tid % 7u;
            uint32_t acc = wd == 0u ? 0u : (wd <= 3u ? 0xffffffffu : 0u);
            for (uint32_t q = 0; q < kPriv; q++) {
                const uint32_t v = wd == 0u ? p_cnt[q][bk] : p_bbox[q][bk][wd - 1u];
                acc = wd == 0u ? acc + v : (wd <= 3u ? min(acc, v) : max(acc, v));
            }
            if (wd == 0u) s_cnt[bk] = acc; else s_bbox[bk][wd - 1u] = acc;
        }
    }
    __syncthreads();
    if (tid == 0) { // ---- costs (:279-300): prefix / suffix unions, the host's f32 expression, first minimum wins
        const float FMAX = 3.40282347e+38f;
        float amn[3] = {FMAX, FMAX, FMAX}, amx[3] = {-FMAX, -FMAX, -FMAX};
        uint32_t an = 0;
        for (uint32_t i = 0; i < RT_BVH_BUCKETS; i++) {
            for (int k = 0; k < 3; k++) { amn[k] = __builtin_fminf(amn[k], bvh_unkey(s_bbox[i][k])); amx[k] = __builtin_fmaxf(amx[k], bvh_unkey(s_bbox[i][3 + k])); }
            an += s_cnt[i];
            for (int k = 0; k < 3; k++) { pre_mn[i][k] = amn[k]; pre_mx[i][k] = amx[k]; }
            pre_n[i] = an;
        }
        for (int k = 0; k < 3; k++) { amn[k] = FMAX; amx[k] = -FMAX; }
        an = 0;
        for (uint32_t i = RT_BVH_BUCKETS; i-- > 0;) {
            for (int k = 0; k < 3; k++) { amn[k] = __builtin_fminf(amn[k], bvh_unkey(s_bbox[i][k])); amx[k] = __builtin_fmaxf(amx[k], bvh_unkey(s_bbox[i][3 + k])); }
            an += s_cnt[i];
            for (int k = 0; k < 3; k++) { suf_mn[i][k] = amn[k]; suf_mx[i][k] = amx[k]; }
            suf_n[i] = an;
        }
        float nmn[3], nmx[3];
        for (int k = 0; k < 3; k++) { nmn[k] = bvh_unkey(s_box[k]); nmx[k] = bvh_unkey(s_box[3 + k]); }
        const float total_area = bvh_area(nmn, nmx);
        uint32_t best = 0;
        float best_cost = 0.0f;
        for (uint32_t i = 0; i + 1 < RT_BVH_BUCKETS; i++) {
            const float cost = 0.125f + ((float)pre_n[i] * bvh_area(pre_mn[i], pre_mx[i]) + (float)suf_n[i + 1] * bvh_area(suf_mn[i + 1], suf_mx[i + 1])) / total_area;
            if (i == 0 || cost < best_cost) { best = i; best_cost = cost; }
        }
        s_misc[4] = best;
        s_misc[5] = pre_n[best]; // L: the items of buckets <= best
    }
    __syncthreads();
    const uint32_t best = s_misc[4], L = s_misc[5];
    // ---- the partition (:304-315) as its closed-form permutation.  pre[i] = LEFT items in [0, i): within a wave from the ballot of the
    // flags, across waves from the four wave totals (double-buffered: one barrier per 256 items); every thread keeps the running count
    uint32_t run = 0u;
    for (uint32_t base = 0, it = 0; base < n; base += RT_BVH_BLOCK, it ^= 1u) {
        const uint32_t i = base + tid;
        bool flag = false;
        if (i < n) {
            const float4 c = src.cen[t.begin + i];
            flag = bvh_bucket_of(axis == 0 ? c.x : (axis == 1 ? c.y : c.z), lo, hi) <= best;
        }
        const unsigned long long m = __ballot(flag);
        const uint32_t wv = tid >> 6, ln = tid & 63u;
        if (ln == 0u) s_wsum[it][wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0u, total = 0u;
        for (uint32_t q = 0; q < RT_BVH_BLOCK / 64u; q++) { const uint32_t v = s_wsum[it][q]; before += q < wv ? v : 0u; total += v; }
        if (i < n) scratch_pre[t.begin + i] = (run + before + (uint32_t)__popcll(m & ((1ull << ln) - 1ull))) | (flag ? 0x80000000u : 0u);
        run += total;
    }
    __syncthreads(); // (scratch_pre is read below by other threads than wrote it: the barrier also orders the global stores within the workgroup)
    if (L == 0u || L == n) { // the reference's median fallback (:317-326): unreachable (buckets 0 and 11 are never empty); not built here
        if (tid == 0) atomicExch(&counters[2], 1u);
        return;
    }
    if (tid == 0) s_misc[7] = scratch_pre[t.begin + L] & 0x7fffffffu; // LEFT items in [0, L)  (L < n here)
    __syncthreads();
    const uint32_t preL = s_misc[7], m = L - preL; // m holes, m LEFT items at the back
    for (uint32_t i = tid; i < n; i += RT_BVH_BLOCK) { // the tables h_k, b_k (k from 1)
        const uint32_t w = scratch_pre[t.begin + i], left = w >> 31, pre = w & 0x7fffffffu;
        if (i < L && !left) scratch_holes[t.begin + (i - pre)] = i;                       // k - 1 = RIGHT items before i
        if (i >= L && left) scratch_backs[t.begin + (m - 1u - (pre - preL))] = i;          // k = m - LEFT items in [L, i)
    }
    __syncthreads();
    const uint32_t b_m = m ? scratch_backs[t.begin + m - 1u] : n;
    for (uint32_t i = tid; i < n; i += RT_BVH_BLOCK) {
        const uint32_t w = scratch_pre[t.begin + i], left = w >> 31, pre = w & 0x7fffffffu;
        uint32_t to;
        if (i < L) {
            if (left) to = i;
            else { const uint32_t k = i - pre + 1u; to = (k == 1u ? n : scratch_backs[t.begin + k - 2u]) - 1u; }
        } else {
            if (left) { const uint32_t k = m - (pre - preL); to = scratch_holes[t.begin + k - 1u]; }
            else to = (i == L) ? b_m - 1u : i - 1u;
        }
        dst.bmin[t.begin + to] = src.bmin[t.begin + i];
        dst.bmax[t.begin + to] = src.bmax[t.begin + i];
        dst.cen[t.begin + to] = src.cen[t.begin + i];
    }
    if (tid == 0) { // children: left [begin, begin + L), right [begin + L, end) — recursed left first by the reference, which only fixes the final numbering
        const uint32_t id = atomicAdd(&counters[0], 2u);
        const uint32_t slot = atomicAdd(&counters[1], 2u);
        nodes[t.node].left = id; nodes[t.node].right = id + 1u; nodes[t.node].axis = axis;
        next_tasks[slot] = BvhTask{t.begin, t.begin + L, id};
        next_tasks[slot + 1u] = BvhTask{t.begin + L, t.end, id + 1u};
    }
}
// A level's leaves keep their items where they are: the other copy needs them too (the next level reads that one)
__global__ void rt_bvh_copy_leaves_kernel(const BvhTask *tasks, uint32_t n_tasks, const BvhNodeB *nodes, BvhItems src, BvhItems dst)
{
    if (blockIdx.x >= n_tasks) return;
    const BvhTask t = tasks[blockIdx.x];
    if (nodes[t.node].left != 0u) return;
    for (uint32_t i = t.begin + threadIdx.x; i < t.end; i += blockDim.x) { dst.bmin[i] = src.bmin[i]; dst.bmax[i] = src.bmax[i]; dst.cen[i] = src.cen[i]; }
}
// subtree sizes, bottom-up: the nodes [first, first + count) of one level
__global__ void rt_bvh_sizes_kernel(BvhNodeB *nodes, uint32_t first, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    BvhNodeB &nd = nodes[first + i];
    nd.size = nd.left ? 1u + nodes[nd.left].size + nodes[nd.right].size : 1u;
}
// pre-order positions, top-down (:155-178: first child right after its parent, second child after the first child's subtree)
__global__ void rt_bvh_positions_kernel(BvhNodeB *nodes, uint32_t first, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const BvhNodeB &nd = nodes[first + i];
    if (!nd.left) return;
    nodes[nd.left].pos = nd.pos + 1u;
    nodes[nd.right].pos = nd.pos + 1u + nodes[nd.left].size;
}
__global__ void rt_bvh_emit_kernel(const BvhNodeB *nodes, uint32_t n_nodes, rsrt_bvh_node *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BvhNodeB &nd = nodes[i];
    rsrt_bvh_node o;
    memset(&o, 0, sizeof o);
    for (int k = 0; k < 3; k++) { o.bounds_min[k] = nd.bmin[k]; o.bounds_max[k] = nd.bmax[k]; }
    if (nd.left) { o.primitives_or_second_child_index = nodes[nd.right].pos; o.primitives_len = 0u; o.split_axis = nd.axis; }
    else { o.primitives_or_second_child_index = nd.begin; o.primitives_len = nd.count; o.split_axis = 0u; }
    out[nd.pos] = o;
}
__global__ void rt_bvh_prims_kernel(BvhItems it, uint32_t n, rsrt_primitive_info *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i].primitive_type = as_u(it.bmin[i].w);
    out[i].index = as_u(it.bmax[i].w);
}
