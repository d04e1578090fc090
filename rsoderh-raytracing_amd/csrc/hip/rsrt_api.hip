// rsrt_api.hip — kernels + C-ABI of librsrt.so (include/rsrt.h).  gfx950 only.
//
// Kernel design:
//  rt_render_kernel   persistent path-tracing kernel.  A workgroup is 4 waves; every WAVE pulls
//                     chunks (one owned framebuffer tile x a block of sample indices) from one
//                     global atomic counter.  Lanes own one path each; when a lane's path ends it
//                     is refilled with the next (pixel, sample) of the wave's chunk — work is
//                     handed to the idle lanes with a wave64 ballot + prefix popcount, no LDS or
//                     atomics involved.  Each finished path stores its radiance to the sample
//                     buffer [sample][pixel slot]; nothing is accumulated in flight, so any lane
//                     may take any sample and the result cannot depend on scheduling.
//  rt_resolve_kernel  adds the buffered samples of every owned pixel into the RGBA32F
//                     accumulator in increasing sample order — the exact f32 sum that
//                     `sample_count` successive reference frames produce (shader.wgsl:1367-1371).
//  rt_cast_rays_kernel the ray-query probe.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rsrt.h"
#include "../../../include/rsrt_tonemap.h"
#include "rt_device.h"

#define RT_BLOCK 256
#define RT_STATS_WORDS 40 // paths, ext, shadow, traversal steps + 32 diagnostic words (zero in the product build)
#define RT_WAVE 64
#ifndef RT_WALK_POOL
#define RT_WALK_POOL 192u // path slots per wave of the 1024-thread walk kernels (hybrid scene view): measured 128 / 160 / 192 (profiles/r03_wide_walk.txt): the
                          // bigger pool wins even where it takes LDS from the top block (15 k-triangle scene: 161 instead of 352 wide nodes staged, same time)
#endif
#ifndef RT_BIG_POOL
#define RT_BIG_POOL 192u // RSRT_KERNEL=4: 1024-thread workgroups, this many slots per wave
#endif

struct RenderParams {
    DevScene scene;
    DevEnv env;
    float cam_pos[3];
    float cam_rot[9]; // columns
    float fov_y;
    uint32_t width, height;
    uint32_t sample_begin, sample_count; // of this pass
    uint32_t max_bounces, flags;
    uint32_t tile_w, tile_h, tiles_x, n_tiles, rank, world, n_owned_tiles;
    uint32_t tiles_per_row, skew; // partition arithmetic, see owned_tile()
    uint32_t samples_per_chunk, n_sblocks, n_chunks;
    uint32_t chunk_px, n_subtiles; // a chunk covers chunk_px consecutive pixels of a tile (tile_px / n_subtiles)
    uint32_t n_slots; // n_owned_tiles * tile_w * tile_h
    uint32_t trace_budget, descend_quorum, flat_quorum, stop_quorum;
    uint32_t coop_lds_cap, coop_lifo_at, coop_narrow_at; // cooperative walk (rt_coop.h): node-queue entries kept in LDS, outstanding items at which a wave pops newest first / one item a trip
    uint32_t *cold_state; // wave-pool kernel: global arena of the cold path-state columns
    float *sample_buf;
    unsigned int *work_counter;
    unsigned long long *stats; // paths, ext_rays, shadow_rays
};

// ------------------------------------------------------------------ framebuffer partition
// The frame is cut into tile_w x tile_h tiles; tile (tx, ty) belongs to rank (tx + ty * skew) % world: interleaved in x, and
// every row shifted by `skew` against the one above, so that a rank's tiles form a lattice whatever tiles_x is (plain
// t % world degenerates to fixed column stripes whenever tiles_x is a multiple of world: 1920 / 16 = 120 tiles, 8 GPUs).
// skew = the smallest odd number >= 3 that has no factor in common with world (3 for 2 / 4 / 8 GPUs).  Every rank owns
// tiles_per_row = ceil(tiles_x / world) tile slots of every tile row — the same number for all ranks, which is what lets the
// frame be GATHERED from equally sized compact buffers (rsrt_comm.h); a slot whose tx falls beyond the frame is padding.
__host__ __device__ inline uint32_t partition_skew(uint32_t world)
{
    for (uint32_t s = 3;; s += 2) {
        uint32_t a = s, b = world;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1u) return s;
    }
}
// rank's j-th tile slot -> tile coordinates; false: a padding slot (nothing to render)
__host__ __device__ inline bool owned_tile(uint32_t j, uint32_t rank, uint32_t world, uint32_t skew, uint32_t tiles_x, uint32_t tiles_per_row,
                                           uint32_t &tx, uint32_t &ty)
{
    ty = j / tiles_per_row;
    const uint32_t k = j - ty * tiles_per_row;
    tx = (rank + world - ((ty % world) * (skew % world)) % world) % world + k * world; // (32-bit: world <= 64)
    return tx < tiles_x;
}

template <bool LDS>
__device__ __forceinline__ SceneView<LDS> make_view(const DevScene &sc);

template <>
__device__ __forceinline__ SceneView<true> make_view<true>(const DevScene &sc)
{
    SceneView<true> v; // image order: nodes | escape links | primitive records | triangle normals | materials | fallback records | flat leaves
    v.o_nodes = 0;
    v.o_esc = v.o_nodes + 2u * sc.n_nodes;
    v.o_prims = v.o_esc + (8u * sc.n_nodes + 3u) / 4u;
    v.o_trin = v.o_prims + 4u * sc.n_prims;
    v.o_mats = v.o_trin + 3u * sc.n_tris;
    v.o_fbs = v.o_mats + 4u * sc.n_materials;
    v.o_fbp = v.o_fbs + 4u * sc.n_spheres;
    v.o_flat = v.o_fbp + 4u * sc.n_planes;
    v.pnodes = sc.pnodes;
    v.wnodes = sc.wnodes;
    return v;
}
template <>
__device__ __forceinline__ SceneView<false> make_view<false>(const DevScene &sc)
{
    return SceneView<false>{sc.nodes, sc.prims, sc.tri_normals, sc.materials, sc.fb_spheres, sc.fb_planes, sc.escape, sc.flat_leaves, sc.pnodes, sc.wnodes};
}

__device__ __forceinline__ SceneViewHybrid make_view_hybrid(const DevScene &sc)
{
    const bool wide = sc.lds_hybrid == 3u; // the wide walk's image: a prefix of the wide nodes
    return SceneViewHybrid{0u, 2u * sc.n_nodes, sc.prims, sc.tri_normals, sc.materials, sc.fb_spheres, sc.fb_planes, sc.pnodes, sc.lds_float4s, sc.wnodes, sc.lds_hybrid,
                           wide ? sc.lds_wnodes : 0u};
}

// Copies the scene image into LDS (the arrays are contiguous in one device allocation, in the order
// make_view<true> assumes).
__device__ __forceinline__ void stage_scene_lds(const DevScene &sc)
{
    for (uint32_t i = threadIdx.x; i < sc.lds_float4s; i += blockDim.x) rt_smem[i] = sc.lds_src[i];
    __syncthreads();
}

struct PathState {
    V3 o, d, throughput, light;
    float last_pdf;
    uint32_t rng, bounce, slot; // slot = float index / 3 into the sample buffer
};

// shader.wgsl:1305-1364: seed, jitter, camera ray
__device__ __forceinline__ void start_path(const RenderParams &P, uint32_t px, uint32_t py, uint32_t sample, PathState &s)
{
    uint32_t pixel_index = py * P.width + px;
    uint32_t rng = 0;
    salt_rng(rng, pixel_index);
    salt_rng(rng, sample);
    float angle = random_uniform(rng) * 2.0f * 3.1415926f; // random_in_circle_uniform :627-631
    float cx = rsrt_cosf(angle), cy = rsrt_sinf(angle);
    float rad = rsrt_sqrtf(random_uniform(rng));
    float fx = (float)px + cx * rad, fy = (float)py + cy * rad;
    float sx = ((fx / (float)P.width) * 2.0f - 1.0f) * 1.0f;
    float sy = ((fy / (float)P.height) * 2.0f - 1.0f) * -1.0f;
    float m = rsrt_sinf(P.fov_y / 2.0f);
    float aspect = (float)P.width / (float)P.height;
    V3 rcs = v3(sx * m * aspect, sy * m, -1.0f);
    V3 c0 = v3(P.cam_rot[0], P.cam_rot[1], P.cam_rot[2]), c1 = v3(P.cam_rot[3], P.cam_rot[4], P.cam_rot[5]),
       c2 = v3(P.cam_rot[6], P.cam_rot[7], P.cam_rot[8]);
    s.o = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    s.d = normalize(mat3_mul(c0, c1, c2, rcs));
    s.throughput = v3(1, 1, 1);
    s.light = v3(0, 0, 0);
    s.last_pdf = 1.0f;
    s.rng = rng;
    s.bounce = 0;
}

template <bool LDS>
__global__ __launch_bounds__(RT_BLOCK) void rt_render_kernel(RenderParams P)
{
    const DevScene &sc = P.scene;
    if (LDS) stage_scene_lds(sc);
    const SceneView<LDS> S = make_view<LDS>(sc);
    uint32_t *stack = reinterpret_cast<uint32_t *>(rt_smem + sc.lds_float4s) + threadIdx.x;
    const uint32_t stride = RT_BLOCK;
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const bool prune = (P.flags & RSRT_FLAG_PRUNE) != 0;
    const bool anyhit_shadow = !(P.flags & RSRT_FLAG_REFERENCE_TRAVERSAL);
    const uint32_t tile_px = P.tile_w * P.tile_h;

    // wave-uniform chunk cursor
    uint32_t chunk_next = 0, chunk_left = 0, chunk_tile_slot0 = 0, chunk_tx0 = 0, chunk_ty0 = 0, chunk_s0 = 0, chunk_p0 = 0;
    bool exhausted = false;

    PathState ps;
    bool active = false;
    unsigned long long n_paths = 0, n_ext = 0, n_shadow = 0;

    for (;;) {
        // ---------------- refill idle lanes from the wave's chunk (ballot + prefix popcount)
        while (!exhausted) {
            unsigned long long need = __ballot(!active);
            if (need == 0ull) break;
            if (chunk_left == 0u) {
                uint32_t c = 0;
                if (lane == (uint32_t)__builtin_ctzll(need)) c = atomicAdd(P.work_counter, 1u);
                c = __builtin_amdgcn_readlane((int)c, __builtin_ctzll(need));
                if (c >= P.n_chunks) { exhausted = true; break; }
                const uint32_t q = c % P.n_subtiles, cb = c / P.n_subtiles; // sub-tile, then (tile, sample block)
                uint32_t j = cb / P.n_sblocks, b = cb % P.n_sblocks; // owned-tile ordinal, sample block
                uint32_t ttx, tty;
                if (!owned_tile(j, P.rank, P.world, P.skew, P.tiles_x, P.tiles_per_row, ttx, tty)) continue; // padding slot: next chunk
                chunk_tx0 = ttx * P.tile_w;
                chunk_ty0 = tty * P.tile_h;
                chunk_tile_slot0 = j * tile_px;
                chunk_s0 = b * P.samples_per_chunk;
                uint32_t ns = min(P.samples_per_chunk, P.sample_count - chunk_s0);
                chunk_p0 = q * P.chunk_px;
                chunk_next = 0;
                chunk_left = ns * P.chunk_px;
            }
            uint32_t n_need = (uint32_t)__popcll(need);
            uint32_t take = min(n_need, chunk_left);
            uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!active && rank < take) {
                uint32_t item = chunk_next + rank;
                uint32_t k = item / P.chunk_px, p = chunk_p0 + item % P.chunk_px;
                uint32_t px = chunk_tx0 + p % P.tile_w, py = chunk_ty0 + p / P.tile_w;
                if (px < P.width && py < P.height) {
                    uint32_t srel = chunk_s0 + k;
                    start_path(P, px, py, P.sample_begin + srel, ps);
                    ps.slot = srel * P.n_slots + chunk_tile_slot0 + p;
                    active = true;
                    n_paths++;
                }
            }
            chunk_next += take;
            chunk_left -= take;
        }
        if (__ballot(active) == 0ull) break;

        // ---------------- extension ray: cast_ray (shader.wgsl:1221)
        Hit hit;
        hit.t = RT_INFINITY;
        if (active) {
            trace_closest(S, sc, ps.o, ps.d, prune, stack, stride, hit);
            n_ext++;
        }
        bool finished = false;
        bool want_shadow = false;
        Surface surf;
        BsdfMaterial mat;
        EnvironmentSample es;
        float cos_nee = 0.0f;
        if (active) {
            if (!hit.did_hit()) { // escaped: shader.wgsl:1222-1231
                float u, v;
                direction_to_equirectangular_uv(ps.d, u, v);
                V3 sky = sample_env_bilinear(P.env, u, v);
                float pdf = environment_direction_pdf(P.env, ps.d, u, v);
                float w = power_heuristic(ps.last_pdf, pdf);
                ps.light = ps.light + ps.throughput * sky * w;
                finished = true;
            } else {
                surf = resolve_hit(S, hit, ps.o, ps.d);
                mat = load_material(S, surf.material_id);
                ps.light = ps.light + ps.throughput * mat.emission; // :1236
                es = sample_environment(P.env, ps.rng);             // :1240
                cos_nee = fmax_(0.0f, dot(surf.normal, es.direction));
                want_shadow = cos_nee > 0.0f && es.pdf > 0.0f;      // :1246-1247
            }
        }
        // ---------------- NEE shadow query: cast_ray_bvh from the hit point (:1249)
        bool occluded = false;
        if (active && want_shadow) {
            Hit sh;
            if (anyhit_shadow) trace_bvh<true>(S, surf.point, es.direction, prune, stack, stride, sh);
            else trace_bvh<false>(S, surf.point, es.direction, prune, stack, stride, sh);
            occluded = sh.did_hit();
            n_shadow++;
        }
        if (active && !finished) {
            if (want_shadow && !occluded) { // :1251-1265
                Frame frame = make_frame(surf.normal);
                V3 wo = to_frame_local(frame, -ps.d);
                V3 wi = to_frame_local(frame, es.direction);
                V3 scattering = bsdf_eval_local(wo, wi, mat);
                float pdf_bsdf = bsdf_pdf_local(wo, wi, mat);
                float w = power_heuristic(es.pdf, pdf_bsdf);
                ps.light = ps.light + ps.throughput * w * es.radiance * scattering * cos_nee / es.pdf;
            }
            BsdfSample bs = bsdf_sample(ps.d, surf.normal, mat, ps.rng); // :1270
            if (bs.dir.x == 0.0f && bs.dir.y == 0.0f && bs.dir.z == 0.0f) {
                ps.light = bs.scattering; // debug colour overwrites, :1274
                finished = true;
            } else if (bs.pdf <= 0.0f) {
                finished = true;
            } else {
                float c = fmax_(0.0f, dot(surf.normal, bs.dir));
                ps.throughput = ps.throughput * (bs.scattering * (c / bs.pdf));
                if (length(ps.throughput) < 0.001f) finished = true;
                else {
                    ps.last_pdf = bs.pdf;
                    ps.o = surf.point;
                    ps.d = bs.dir;
                }
            }
            ps.bounce++;
            if (ps.bounce >= P.max_bounces) finished = true;
        }
        if (active && finished) {
            float *dst = P.sample_buf + (size_t)ps.slot * 3u;
            dst[0] = ps.light.x;
            dst[1] = ps.light.y;
            dst[2] = ps.light.z;
            active = false;
        }
    }

    // per-wave reduction of the counters, one atomic per wave and counter
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
}

#include "rt_wavepool.h"
// LDS of the 1024-thread walk kernels (hybrid scene view): sixteen waves' pools and beside them the traversal's part of the scene
static constexpr size_t kHybridPoolBytes = (size_t)(1024 / RT_WAVE) * 4u * ((size_t)H_COUNT * RT_WALK_POOL + pool_list_dwords(4));
static constexpr uint32_t kHybridRoomF4 = (uint32_t)((160 * 1024 - kHybridPoolBytes) / sizeof(float4));
// ... and of the cooperative walk's kernel (TRAV 6): 128-slot pools with their two work stacks
#ifndef RT_COOP_POOL
#define RT_COOP_POOL 128u // (build-time A/B: -DRT_COOP_POOL=96u with RSRT_WPS=5 fits a fifth wave per SIMD beside 256-thread workgroups)
#endif
#ifndef RT_COOP_BLOCK
#define RT_COOP_BLOCK 1024 // threads of the one workgroup a CU holds beside the staged node prefix
#endif
static constexpr uint32_t kCoopRoomF4 = (uint32_t)((160 * 1024 - (size_t)(RT_COOP_BLOCK / RT_WAVE) * 4u * pool_wave_lds_dwords(6, RT_COOP_POOL)) / sizeof(float4));
#include "rt_alias_device.h"
#include "rt_bvh_device.h"

// total = textureLoad(cumulative) + sample, once per sample in order (shader.wgsl:1367-1371)
__global__ __launch_bounds__(RT_BLOCK) void rt_resolve_kernel(RenderParams P, float4 *accum)
{
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= P.n_slots) return;
    uint32_t tile_px = P.tile_w * P.tile_h;
    uint32_t j = slot / tile_px, p = slot % tile_px;
    uint32_t ttx, tty;
    if (!owned_tile(j, P.rank, P.world, P.skew, P.tiles_x, P.tiles_per_row, ttx, tty)) return;
    uint32_t px = ttx * P.tile_w + p % P.tile_w, py = tty * P.tile_h + p / P.tile_w;
    if (px >= P.width || py >= P.height) return;
    float4 a = accum[(size_t)py * P.width + px];
    const float *src = P.sample_buf + (size_t)slot * 3u;
    const size_t step = (size_t)P.n_slots * 3u;
    for (uint32_t k = 0; k < P.sample_count; k++, src += step) {
        const rt_f3v v = __builtin_nontemporal_load(reinterpret_cast<const rt_f3v_a4 *>(src)); // read once, never again
        a.x = a.x + v.x;
        a.y = a.y + v.y;
        a.z = a.z + v.z;
    }
    a.w = 1.0f;
    accum[(size_t)py * P.width + px] = a;
}

// The ray-query probe.  SV / TRAV as in rt_render_pool_kernel (where the scene is read from, which traversal
// runs); TRAV == 4 is the first kernel's stack walk.  mode bit 0: cast_ray_bvh only (no brute-force fallback).
template <int SV, int TRAV>
__global__ __launch_bounds__(RT_BLOCK) void rt_cast_rays_kernel(DevScene sc, uint32_t n, const float *origins, const float *dirs,
                                                                uint32_t mode, uint32_t flags, uint32_t repeat, rsrt_hit *out)
{
    if (SV != 0) stage_scene_lds(sc);
    const typename PoolView<SV>::type S = PoolView<SV>::make(sc);
    uint32_t *stack = reinterpret_cast<uint32_t *>(rt_smem + sc.lds_float4s) + threadIdx.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
    V3 d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const bool prune = (flags & RSRT_FLAG_PRUNE) != 0;
    Hit h;
    h.t = RT_INFINITY; h.ref = 0; h.src = SRC_BVH; h.u = h.v = 0.0f;
    if (TRAV == 4) { // the stack traversal of the first kernel
        trace_bvh<false>(S, o, d, prune, stack, RT_BLOCK, h);
    } else { // what the production kernel's TRACE stage runs, resumed until done as the scheduler would
#ifdef RT_INSTRUMENT
        DbgCounters dbg;
#endif
        uint32_t cur = 0, work = 0;
        unsigned long long flat_rem = 0ull;
        uint32_t wmem[RT_WSTATE_WORDS]; // (the wide walk parks its stack here between calls, as the pool kernel does in the slot's cold columns)
        constexpr int T = TRAV; // (probe numbering: 4 is the first kernel's stack walk, handled above; 5 the wide walk — the form whose stack may overflow, which takes any tree)
        while (cur != RT_END) trace_dispatch<T>(DBG_ARG S, sc, o, d, prune, false, T >= 4 ? 2u : 12u, 50u, cur, h, nullptr, work, flat_rem, wmem, 1u, 60u);
        // RSRT_PROBE_REPEAT (tools/trace_rate.py): the same query again and again, so that a timing of this kernel is a timing
        // of the traversal and not of staging the scene for 256 rays; the result does not change
        for (uint32_t k = 1; k < repeat; k++) {
            asm volatile("" : "+v"(o.x), "+v"(d.x));
            Hit h2;
            h2.t = RT_INFINITY; h2.ref = 0; h2.src = SRC_BVH; h2.u = h2.v = 0.0f;
            cur = 0;
            while (cur != RT_END) trace_dispatch<T>(DBG_ARG S, sc, o, d, prune, false, T >= 4 ? 2u : 12u, 50u, cur, h2, nullptr, work, flat_rem, wmem, 1u, 60u);
            h.t = h2.t; h.ref = h2.ref; h.src = h2.src;
        }
        if (h.did_hit()) hit_barycentrics(S, h, o, d); // as SHADE does: the traversals do not carry u, v
    }
    if ((mode & 1u) == 0 && !h.did_hit()) { // cast_ray's brute-force fallback (the MISS stage)
        for (uint32_t k = 0; k < sc.n_spheres; k++) {
            float u, v;
            float t = test_record(S, k, SRC_FB_SPHERE, o, d, u, v);
            if (t >= 0.0f && t < h.t) { h.t = t; h.ref = k; h.src = SRC_FB_SPHERE; }
        }
        for (uint32_t k = 0; k < sc.n_planes; k++) {
            float u, v;
            float t = test_record(S, k, SRC_FB_PLANE, o, d, u, v);
            if (t >= 0.0f && t < h.t) { h.t = t; h.ref = k; h.src = SRC_FB_PLANE; }
        }
    }
    rsrt_hit r;
    memset(&r, 0, sizeof r);
    if (h.did_hit()) {
        Surface s = resolve_hit(S, h, o, d);
        r.did_hit = 1;
        r.distance = h.t;
        r.hit_point[0] = s.point.x; r.hit_point[1] = s.point.y; r.hit_point[2] = s.point.z;
        r.normal[0] = s.normal.x; r.normal[1] = s.normal.y; r.normal[2] = s.normal.z;
        r.material_id = s.material_id;
    } else if ((mode & 1u) == 0) {
        r.distance = RT_INFINITY; // cast_ray returns its `result` initialiser on a total miss (shader.wgsl:568-574, :600)
    }
    out[i] = r;
}

// The ray-query probe of the cooperative walk (TRAV 6, rt_coop.h): a wave takes 64 rays into a one-column "pool" of its own (slot = lane) and runs
// the very functions the production kernel's TRACE stage runs — coop_push_rays, coop_trace, coop_slow_rays.  gstack: RT_COOP_GCAP dwords a wave.
template <int SV>
__global__ __launch_bounds__(RT_BLOCK) void rt_cast_rays_coop_kernel(DevScene sc, uint32_t n, const float *origins, const float *dirs,
                                                                     uint32_t mode, uint32_t flags, uint32_t repeat, rsrt_hit *out, uint32_t *gstack,
                                                                     uint32_t lds_cap, uint32_t lifo_at, uint32_t narrow_at)
{
    if (SV != 0) stage_scene_lds(sc);
    const typename PoolView<SV>::type S = PoolView<SV>::make(sc);
    constexpr uint32_t kPool = 64u;
    typedef CoopCols<kPool> C;
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1u);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / RT_WAVE));
    uint32_t *const W = reinterpret_cast<uint32_t *>(rt_smem + sc.lds_float4s) + wave * pool_wave_lds_dwords(6, kPool);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const V3 o = valid ? v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]) : v3(0.0f, 0.0f, 0.0f);
    const V3 d = valid ? v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]) : v3(0.0f, 0.0f, 1.0f);
#ifdef RT_INSTRUMENT
    DbgCounters dbg;
#endif
    Hit h;
    h.t = RT_INFINITY; h.ref = 0; h.src = SRC_BVH; h.u = h.v = 0.0f;
    uint32_t work = 0;
    for (uint32_t k = 0; k < repeat; k++) { // (RSRT_PROBE_REPEAT: the same queries again; the result does not change)
        W[C::O + lane] = as_u(o.x); W[C::O + kPool + lane] = as_u(o.y); W[C::O + 2u * kPool + lane] = as_u(o.z);
        W[C::E + lane] = as_u(d.x); W[C::E + kPool + lane] = as_u(d.y); W[C::E + 2u * kPool + lane] = as_u(d.z);
        W[C::S + lane] = 0u; W[C::S + kPool + lane] = 0u; W[C::S + 2u * kPool + lane] = 0u;
        W[C::CT + lane] = (uint32_t)F_EXT | (uint32_t)TAG_TRACE;
        RT_WAVE_HANDOVER();
        CoopStacks cs;
        cs.ls = W + C::DWORDS; cs.ns = cs.ls + RT_COOP_LCAP; cs.map = reinterpret_cast<uint16_t *>(cs.ns + RT_COOP_NCAP);
        cs.gs = gstack + (size_t)(blockIdx.x * (RT_BLOCK / RT_WAVE) + wave) * RT_COOP_GCAP;
        cs.ns_h = cs.ns_n = cs.ls_n = cs.gs_n = 0u;
        cs.lds_cap = lds_cap; cs.lifo_at = lifo_at; cs.narrow_at = narrow_at;
        coop_push_rays<kPool>(W, cs, valid, lane, W[C::CT + lane], (uint32_t)F_EXT, (uint32_t)F_SHADOW);
        coop_trace<kPool>(DBG_ARG S, W, cs, false, lane, work);
        RT_WAVE_HANDOVER();
        const bool mine[1] = {valid};
        const uint32_t slots[1] = {lane};
        coop_slow_rays<kPool, 1u>(DBG_ARG S, sc, W, mine, slots, false, work);
        h.t = as_f(W[C::BEST + 2u * lane + 1u]);
        h.ref = W[C::BEST + 2u * lane];
        RT_WAVE_HANDOVER();
    }
    if (!valid) return; // (nothing wave-wide from here on)
    if (h.did_hit()) hit_barycentrics(S, h, o, d); // as SHADE does: the traversals do not carry u, v
    if ((mode & 1u) == 0 && !h.did_hit()) { // cast_ray's brute-force fallback (the MISS stage)
        for (uint32_t k = 0; k < sc.n_spheres; k++) {
            float u, v;
            float t = test_record(S, k, SRC_FB_SPHERE, o, d, u, v);
            if (t >= 0.0f && t < h.t) { h.t = t; h.ref = k; h.src = SRC_FB_SPHERE; }
        }
        for (uint32_t k = 0; k < sc.n_planes; k++) {
            float u, v;
            float t = test_record(S, k, SRC_FB_PLANE, o, d, u, v);
            if (t >= 0.0f && t < h.t) { h.t = t; h.ref = k; h.src = SRC_FB_PLANE; }
        }
    }
    rsrt_hit r;
    memset(&r, 0, sizeof r);
    if (h.did_hit()) {
        Surface s = resolve_hit(S, h, o, d);
        r.did_hit = 1;
        r.distance = h.t;
        r.hit_point[0] = s.point.x; r.hit_point[1] = s.point.y; r.hit_point[2] = s.point.z;
        r.normal[0] = s.normal.x; r.normal[1] = s.normal.y; r.normal[2] = s.normal.z;
        r.material_id = s.material_id;
    } else if ((mode & 1u) == 0) {
        r.distance = RT_INFINITY; // cast_ray returns its `result` initialiser on a total miss (shader.wgsl:568-574, :600)
    }
    out[i] = r;
}

__global__ void rt_mean_f16_kernel(const float4 *accum, size_t n, float inv_is_unused, uint32_t sample_total, ushort4 *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i];
    float cnt = (float)sample_total; // total_light / f32(sample_count + 1), shader.wgsl:1369
    __half hx = __float2half_rn(a.x / cnt), hy = __float2half_rn(a.y / cnt), hz = __float2half_rn(a.z / cnt), hw = __float2half_rn(1.0f);
    out[i] = make_ushort4(__half_as_ushort(hx), __half_as_ushort(hy), __half_as_ushort(hz), __half_as_ushort(hw));
}

// The reference's developer views (shader.wgsl:1314-1338: what `main` writes to out_texture instead of a render when dev_index is 2 or 3).
// View 3 ("display HDRI", :1333-1338): out[x, y] = saturate(environment texel (x, y)), alpha 0; a pixel outside the map reads zeros.
__global__ void rt_dev_view_hdri_kernel(const float4 *env_rgba, uint32_t env_w, uint32_t env_h, uint32_t width, uint32_t height, ushort4 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t x = (uint32_t)(i % width), y = (uint32_t)(i / width);
    float3 c = make_float3(0.0f, 0.0f, 0.0f);
    if (x < env_w && y < env_h) { const float4 t = env_rgba[(size_t)y * env_w + x]; c = make_float3(t.x, t.y, t.z); } // (the texel's alpha is the device's own: not looked at)
    auto sat = [](float v) { return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f); };
    out[i] = make_ushort4(__half_as_ushort(__float2half_rn(sat(c.x))), __half_as_ushort(__float2half_rn(sat(c.y))), __half_as_ushort(__float2half_rn(sat(c.z))), __half_as_ushort(__float2half_rn(0.0f)));
}
// View 2 ("draw pixels based on distribution", :1314-1332): every pixel draws 20 indices from the environment's alias table, seeded as a
// render's pixel is (:1309-1312), and adds 0.1 / 20 to THAT texel's position of out_texture.  In the shader the invocations read-modify-write
// the binary16 texture unordered — which draws survive a frame depends on the GPU's scheduling; this is the frame in which every draw
// lands: first the draws are counted per texel (integer atomics), then each texel takes its count of `v <- f16(f32(v) + 0.1 / 20)` steps
// (equal addends: the order of the draws cannot matter), stopping early once a step no longer changes it.
__global__ void rt_dev_view_count_kernel(const uint4 *alias, uint32_t env_w, uint32_t env_h, uint32_t width, uint32_t height, uint32_t sample_count, uint32_t *counts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * height) return;
    uint32_t rng = 0;
    salt_rng(rng, (uint32_t)i);      // pixel_index = y * resolution.x + x
    salt_rng(rng, sample_count);
    const uint32_t length = env_w * env_h;
    for (uint32_t k = 0; k < 20u; k++) { // random_index_in_environment, :689-706
        const uint32_t index = min(f2u(random_uniform(rng) * (float)length), length - 1u);
        const uint4 entry = alias[index];
        const uint32_t pick = random_uniform(rng) < as_f(entry.x) ? index : entry.y;
        const uint32_t x = pick % env_w, y = pick / env_w;
        if (x < width && y < height) atomicAdd(&counts[(size_t)y * width + x], 1u); // (a store outside the texture is dropped)
    }
}
__global__ void rt_dev_view_apply_kernel(const uint32_t *counts, size_t n, ushort4 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = counts[i];
    if (c == 0u) return;
    const float step = 0.1f / 20.0f; // vec3(0.1, 0.1, 0.1) / f32(count)
    ushort4 v = out[i];
    unsigned short *ch[3] = {&v.x, &v.y, &v.z};
    for (int k = 0; k < 3; k++) {
        unsigned short h = *ch[k];
        for (uint32_t j = 0; j < c; j++) {
            const unsigned short nx = __half_as_ushort(__float2half_rn(__half2float(__ushort_as_half(h)) + step));
            if (nx == h) break; // (a fixed point: the remaining steps change nothing)
            h = nx;
        }
        *ch[k] = h;
    }
    v.w = __half_as_ushort(__float2half_rn(0.0f)); // textureStore(out_texture, .., vec4(color, 0.))
    out[i] = v;
}

// hdr.wgsl fs_main + the *Srgb surface write: mean (through binary16) -> ACES -> sRGB 8-bit
__global__ void rt_display_kernel(const float4 *accum, size_t n, uint32_t sample_total, uchar4 *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    const float sum[3] = {a.x, a.y, a.z};
    unsigned char rgb[3];
    rsrt_display_pixel(sum, (float)sample_total, rgb);
    out[i] = make_uchar4(rgb[0], rgb[1], rgb[2], 255);
}

// =================================================================== host side
namespace {

thread_local std::string g_create_error;

struct Env {
    float4 *rgba = nullptr;
    uint4 *alias = nullptr;
    uint32_t width = 0, height = 0;
};

} // namespace

// Kernel variants (RSRT_KERNEL): 0 = lockstep megakernel (first kernel); 1, 2, 3 = stage-scheduled wave-pool kernel
// with 192 / 160 / 128 path slots per wave in 256-thread workgroups, each with its own LDS copy of a small scene (160: four
// workgroups per CU fit in LDS); 4 (default) = for a scene whose whole image fits LDS, ONE 1024-thread workgroup per CU — one
// scene copy instead of four, which is what makes room for 192 slots per wave (-0.8 % on the BASELINE frame); anything else as 2.
#define RT_N_VARIANTS 5
static const uint32_t kVariantPool[RT_N_VARIANTS] = {0, 192, 160, 128, RT_BIG_POOL};
template <int SV, uint32_t BLOCK, uint32_t POOL>
static const void *pool_function(int trav)
{
    switch (trav) {
    case 0: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 0>);
    case 1: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 1>);
    case 3: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 3>);
    case 4: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 4>);
    case 5: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 5>);
    default: return reinterpret_cast<const void *>(&rt_render_pool_kernel<SV, BLOCK, POOL, 2>);
    }
}
// sv: 0 scene in global memory, 1 whole image in LDS, 2 hybrid (nodes + escape links in LDS, 1024-thread workgroups)
static const void *variant_function(int kv, int sv, int trav)
{
    if (kv == 0) return sv == 1 ? reinterpret_cast<const void *>(&rt_render_kernel<true>) : reinterpret_cast<const void *>(&rt_render_kernel<false>);
    if (trav == 6) // the cooperative walk: 128-slot pools; one 1024-thread workgroup per CU beside the staged node prefix, else 256-thread workgroups
        return sv == 2 ? reinterpret_cast<const void *>(&rt_render_pool_kernel<2, RT_COOP_BLOCK, RT_COOP_POOL, 6>)
                       : (sv == 1 ? reinterpret_cast<const void *>(&rt_render_pool_kernel<1, RT_BLOCK, RT_COOP_POOL, 6>) : reinterpret_cast<const void *>(&rt_render_pool_kernel<0, RT_BLOCK, RT_COOP_POOL, 6>));
    if (sv == 2) return pool_function<2, 1024, RT_WALK_POOL>(trav == 2 ? 1 : trav); // (the flat loop needs the whole image: never asked for here)
    if (kv == 1) return sv == 1 ? pool_function<1, RT_BLOCK, 192>(trav) : pool_function<0, RT_BLOCK, 192>(trav);
    if (kv == 3) return sv == 1 ? pool_function<1, RT_BLOCK, 128>(trav) : pool_function<0, RT_BLOCK, 128>(trav);
    if (kv == 4 && sv == 1) return pool_function<1, 1024, RT_BIG_POOL>(trav);
    return sv == 1 ? pool_function<1, RT_BLOCK, 160>(trav) : pool_function<0, RT_BLOCK, 160>(trav);
}

template <int SV>
static const void *probe_function_sv(int trav)
{
    switch (trav) {
    case 0: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, 0>);
    case 1: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, 1>);
    case 2: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, (SV == 2 ? 1 : 2)>); // (flat needs the whole image: never asked for with SV 2)
    case 3: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, 3>);
    case 5: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, 5>);
    case 6: return reinterpret_cast<const void *>(&rt_cast_rays_coop_kernel<SV>);
    default: return reinterpret_cast<const void *>(&rt_cast_rays_kernel<SV, 4>);
    }
}
static const void *probe_function(int sv, int trav)
{
    return sv == 0 ? probe_function_sv<0>(trav) : (sv == 1 ? probe_function_sv<1>(trav) : probe_function_sv<2>(trav));
}

struct rsrt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    std::string error;
    std::string description;
    int cus = 0;
    // scene
    float4 *scene_blob = nullptr;
    DevScene scene{};
    bool scene_ready = false;
    uint32_t hybrid_head_f4 = 0, hybrid_pnode_f4 = 0; // mid-size scenes: float4s of nodes + escape links / of the pre-order nodes' top block (0 = none)
    // ... and the wide walks' LDS image: a prefix of the wimg_nodes wide nodes, as long as the launch has room for (0 = no image)
    uint32_t wimg_nodes = 0;
    // environments
    std::vector<Env> envs;
    // partition
    uint32_t rank = 0, world = 1, tile_w = 16, tile_h = 16;
    // accumulator
    float4 *accum = nullptr;      // bound or owned
    float4 *accum_owned = nullptr;
    uint32_t acc_w = 0, acc_h = 0;
    // work buffers
    // Two sets of work buffers ("lanes"), used in turn by successive passes: a pass's path-tracing kernel runs on its lane's own
    // stream and touches nothing but its lane's buffers, so the kernel of call k + 1 fills the CUs that call k's tail is leaving
    // (a launch ends with ~0.5 ms of pipeline drain: at one sample per call — the reference's interactive mode, src/state.rs:
    // 760-833 — that was half the frame time).  Only the small resolve kernels, which add into the accumulator in sample
    // order, stay chained on the caller's stream.  RSRT_OVERLAP=0: one lane, everything on the caller's stream (A/B).
    struct Lane {
        hipStream_t stream = nullptr;
        unsigned int *work_counter = nullptr;
        float *sample_buf = nullptr;
        size_t sample_buf_bytes = 0;
        uint32_t *cold_state = nullptr;
        size_t cold_bytes = 0;
        hipEvent_t resolved = nullptr; // recorded after the resolve that read this lane's sample buffer last
        bool resolved_valid = false;
        hipEvent_t traced = nullptr;   // recorded after this lane's last path-tracing kernel (asked, not waited for: is the GPU busy?)
        bool traced_valid = false;
        hipEvent_t caller_at = nullptr; // where the caller's stream stood when this lane's last pass was enqueued
    };
    // Ordinary jobs use lane 0.  SMALL jobs (at most `small_paths` paths a call: the reference's one sample per
    // frame) that arrive while an earlier kernel is still running are PIPELINED: they take turns over all four lanes and run the
    // 256-thread form of the kernel with one workgroup per CU, so that up to four calls are resident side by side and one call's
    // ~0.5 ms of pipeline fill and drain is covered by its neighbours' steady state (a call alone on the GPU keeps the full grid:
    // its latency is what counts then).
    Lane lanes[8];
    uint32_t n_lanes = 4;   // lanes in use (RSRT_PIPE_LANES: 2 .. 8)
    uint32_t pipe_div = 1;  // a pipelined small job's grid is a CU's worth of workgroups / pipe_div (RSRT_PIPE_DIV: with 8 lanes and 2, each of eight resident jobs has half as many waves that each run twice as long)
    uint32_t next_small_lane = 1;
    bool overlap = true;
    uint64_t small_paths = 4ull << 20;
    unsigned long long *dev_stats = nullptr;
    // stats
    rsrt_stats stats{};
    struct PassEvents { hipEvent_t begin, traced, end; };
    std::vector<PassEvents> pending_events;
    double cum_trace_ms = 0, cum_resolve_ms = 0, base_trace_ms = 0, base_resolve_ms = 0;
    unsigned long long base_counts[4] = {0, 0, 0, 0};
    uint32_t cum_launches = 0, base_launches = 0;
    std::vector<hipEvent_t> event_pool;
    // Work buffers (work_counter, sample_buf, cold_state, scratch) and the accumulator are per-context
    // singletons, so everything the context enqueues is ONE chain whatever streams the caller passes:
    // `last_event` is recorded after every enqueue, and an enqueue on a different stream first waits for it.
    hipEvent_t last_event = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    // multi-GPU (rsrt_comm.h): this context's RCCL communicator (an ncclComm_t), NULL in a world of one
    void *comm = nullptr;
    bool comm_owned = false;
    bool comm_dense_mode = false; // RSRT_COMM_MODE=reduce / rsrt_comm_set_mode: the exchange is a dense ncclReduce instead of the gather of compact tile buffers
    uint32_t comm_rank = 0, comm_world = 1;
    struct ReduceEvents { hipEvent_t begin, end; };
    std::vector<ReduceEvents> pending_reduce;
    double cum_reduce_ms = 0, base_reduce_ms = 0;
    uint32_t cum_reduces = 0;
    void *comm_buf = nullptr; // compact tile buffers of the exchange step (rsrt_comm.h): one per rank on the root, one elsewhere
    size_t comm_buf_bytes = 0;
    // scratch for rsrt_resolve_mean_f16 / rsrt_display_srgb8 (grow-only; no per-frame hipMalloc)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int blocks_per_cu[21][RT_N_VARIANTS] = {}; // [scene view * 7 + traversal][kernel variant]
    int kernel_variant = 4; // index into kVariantPool
    int max_traversal = 6; // most specialised traversal to use where the scene allows it (rt_wavepool.h, TRAV)
    uint32_t coop_lds_cap = RT_COOP_NCAP, coop_lifo_at = RT_COOP_LIFO_AT, coop_narrow_at = RT_COOP_NARROW_AT; // RSRT_COOP_LDS_CAP / _LIFO_AT / _NARROW_AT (tests: force the node queue's spill / newest-first / one-item trips)
    bool allow_flat = true;
    bool allow_hybrid = true;
    uint32_t trace_budget = 0; // traversal steps per TRACE invocation before a ray is re-queued (0: 6 for the fixed-order walk, 12 for the tree walks)
    uint32_t descend_quorum = 30; // fixed-order / wide walk: a descending round ends once fewer than this percentage of its lanes are still descending
    uint32_t stop_quorum = 40; // wide walk: a TRACE call ends (the unfinished rays park their stacks) once fewer than this percentage of its lanes are still walking
    uint32_t chunks_per_wave = 32; // work chunks a resident wave should get at least (RSRT_CHUNKS_PER_WAVE): sets samples per chunk, and sub-tiles for small jobs
    uint32_t flat_quorum = 20; // flat traversal: the triangle loop ends once fewer than this percentage of its lanes still hold triangles (0: never)
    // (every RSRT_* environment knob is read ONCE, in rsrt_context_create)
    size_t sample_buffer_budget = 16ull << 30; // RSRT_SAMPLE_BUFFER_MB: bytes of sample buffer per pass
    int pipe_blocks = 1;     // RSRT_PIPE_BLOCKS: workgroups per CU of a pipelined small job (four such jobs fill a CU)
    int max_blocks_per_cu = 0; // RSRT_BLOCKS_PER_CU: cap on the occupancy the runtime reports (0: none; experiment knob)
    uint32_t probe_repeat = 1; // RSRT_PROBE_REPEAT: rsrt_cast_rays runs every query this many times (tools/trace_rate.py)
    // Pipelined small jobs want four kernels of one context resident at a time, each from a stream of its own; the HIP runtime maps a
    // process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share one run one after the other —
    // measured: 0.85 against 0.63 ms per single-sample call.  The HOST sets that variable, before its first HIP call; the
    // library only looks at it, and says so once when it pipelines over fewer queues than it has lanes.
    int hw_queues = 4;
    bool hw_queue_warned = false;
    unsigned long long debug_words[32] = {0};
};

namespace {

rsrt_status fail(rsrt_context *ctx, rsrt_status st, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_create_error = buf;
    return st;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RSRT_ERR_OUT_OF_MEMORY : RSRT_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                              \
    } while (0)

struct DeviceGuard {
    explicit DeviceGuard(int dev) { (void)hipSetDevice(dev); }
};

inline float fmaxh(float a, float b) { return a < b ? b : a; }
inline float fminh(float a, float b) { return b < a ? b : a; }
inline float saturateh(float x) { return fminh(fmaxh(x, 0.0f), 1.0f); }
inline float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// record of one primitive (4 float4), see rt_device.h
void make_sphere_record(const rsrt_sphere &s, float4 *r)
{
    r[0] = f4(s.pos[0], s.pos[1], s.pos[2], u2f(PRIM_SPHERE | (s.material_id << 2)));
    r[1] = f4(s.radius, s.radius * s.radius, 0, 0);
    r[2] = r[3] = f4(0, 0, 0, 0);
}
void make_plane_record(const rsrt_plane &p, float4 *r)
{
    r[0] = f4(p.pos[0], p.pos[1], p.pos[2], u2f(PRIM_PLANE | (p.material_id << 2)));
    r[1] = f4(p.normal[0], p.normal[1], p.normal[2], 0);
    r[2] = f4(p.base_change_matrix[0][0], p.base_change_matrix[1][0], p.base_change_matrix[2][0], 0); // row x
    r[3] = f4(p.base_change_matrix[0][2], p.base_change_matrix[1][2], p.base_change_matrix[2][2], 0); // row z
}
void make_triangle_record(const rsrt_triangle &t, uint32_t index, const rsrt_vec3 *v, float4 *r)
{
    const float *a = v[t.vertex_0].v, *b = v[t.vertex_1].v, *c = v[t.vertex_2].v;
    r[0] = f4(a[0], a[1], a[2], u2f(PRIM_TRIANGLE | (t.material_id << 2)));
    r[1] = f4(b[0] - a[0], b[1] - a[1], b[2] - a[2], u2f(index)); // edge_0, shader.wgsl:415
    r[2] = f4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0);          // edge_1, :416
    r[3] = f4(0, 0, 0, 0);
}
// Do two primitive records put the same numbers into their intersection test (type and geometry words; not the material, not a
// triangle's normal index)?  Then every ray gets the same t from both.
bool same_test(const float4 *a, const float4 *b)
{
    uint32_t ta, tb;
    memcpy(&ta, &a[0].w, 4); memcpy(&tb, &b[0].w, 4);
    if ((ta & 3u) != (tb & 3u) || memcmp(&a[0], &b[0], 12) != 0) return false;
    if ((ta & 3u) == PRIM_TRIANGLE) return memcmp(&a[1], &b[1], 12) == 0 && memcmp(&a[2], &b[2], 12) == 0;
    return memcmp(a + 1, b + 1, 3 * sizeof(float4)) == 0;
}
// make_bsdf_material / surface_f0 / surface_kd / lobe probabilities (shader.wgsl:850-881, :1147-1148)
void make_material_record(const rsrt_material &m, float4 *r)
{
    const float t = saturateh(m.metallic);
    float f0[3], kd[3];
    for (int k = 0; k < 3; k++) f0[k] = (1.0f - t) * 0.04f + t * m.color[k];
    const float maxf0 = fmaxh(f0[0], fmaxh(f0[1], f0[2]));
    for (int k = 0; k < 3; k++) kd[k] = (m.color[k] * (1.0f - t)) * (1.0f - maxf0);
    const float alpha = fmaxh(0.001f, m.roughness * m.roughness);
    const float ps = saturateh(0.2126f * f0[0] + 0.7152f * f0[1] + 0.0722f * f0[2]);
    r[0] = f4(m.color[0], m.color[1], m.color[2], m.metallic);
    r[1] = f4(f0[0], f0[1], f0[2], alpha);
    r[2] = f4(m.emission[0], m.emission[1], m.emission[2], ps);
    r[3] = f4(kd[0], kd[1], kd[2], 1.0f - ps);
}

// ---- wide walk (rt_device.h, trace_wide): the binary tree collapsed into 4-wide nodes, laid out hottest-first (build_wide_tree)
struct WideNode { uint32_t ch[4]; uint32_t n_ch, n_int, first_child; }; // binary nodes of the children (interior ones first), index of the first interior child's wide node
// false: the scene does not qualify (see rsrt_upload_scene).  prims_out / nodes_out: `prims` with whole leaves reordered so that
// the records of a wide node's leaf children are contiguous, and `nodes` with the leaves' first indices pointing there;
// old_of_new (may be NULL): for every new record index the old one.
bool build_wide_tree(const rsrt_bvh_node *nodes, uint32_t n_nodes, const rsrt_primitive_info *prims, uint32_t n_prims, std::vector<WideNode> &wide,
                     std::vector<rsrt_primitive_info> &prims_out, std::vector<rsrt_bvh_node> &nodes_out, std::vector<uint32_t> *old_of_new, uint32_t *wide_depth = nullptr)
{
    wide.clear();
    bool ok = n_nodes >= 3 && nodes[0].primitives_len == 0;
    {
        std::vector<uint8_t> covered(n_prims, 0);
        for (uint32_t i = 0; i < n_nodes && ok; i++) {
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len > 8) ok = false;
            for (uint32_t k = 0; k < nd.primitives_len && ok; k++) {
                uint8_t &c = covered[nd.primitives_or_second_child_index + k];
                ok = c == 0;
                c = 1;
            }
            if (nd.primitives_len == 0)
                for (uint32_t c : {i + 1u, nd.primitives_or_second_child_index})
                    for (int k = 0; k < 3; k++)
                        ok = ok && nodes[c].bounds_min[k] >= nd.bounds_min[k] && nodes[c].bounds_max[k] <= nd.bounds_max[k];
        }
        for (uint32_t p = 0; p < n_prims; p++) ok = ok && covered[p]; // the permutation below must be total
    }
    if (!ok) return false;
    auto area = [&](uint32_t i) {
        const double dx = (double)nodes[i].bounds_max[0] - nodes[i].bounds_min[0], dy = (double)nodes[i].bounds_max[1] - nodes[i].bounds_min[1],
                     dz = (double)nodes[i].bounds_max[2] - nodes[i].bounds_min[2];
        const double a = dx * dy + dy * dz + dz * dx;
        return a == a ? a : 0.0;
    };
    // Array order = the order in which nodes are ALLOCATED, and a node's interior children are allocated together, when the node is
    // expanded (child k = first child + k).  Nodes are expanded largest box first (a ray meets a node about as often as its box is large, and
    // a child's box lies inside its parent's), so that the array's head — the part the kernel stages in LDS — is the part of the tree the
    // rays visit most.
    std::vector<uint32_t> queue{0u}, level{0u}; // binary roots of the wide nodes, in array order
    uint32_t wdepth = 1;
    const bool by_area = true; // (breadth-first instead: the first version, 0-1.5 % slower, profiles/r03_walk_bounds.txt)
    std::vector<std::pair<double, uint32_t>> heap{{area(0), 0u}}; // (box area, array index) of the nodes still to expand
    size_t bfs_next = 0;
    while (by_area ? !heap.empty() : bfs_next < queue.size()) {
        size_t qi;
        if (by_area) {
            std::pop_heap(heap.begin(), heap.end(), [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) { return a.first < b.first || (a.first == b.first && a.second > b.second); });
            qi = heap.back().second;
            heap.pop_back();
        } else {
            qi = bfs_next++;
        }
        const uint32_t r = queue[qi];
        std::vector<uint32_t> ch{r + 1u, nodes[r].primitives_or_second_child_index};
        while (ch.size() < 4) { // open the interior child with the largest box
            int best = -1;
            for (size_t k = 0; k < ch.size(); k++)
                if (nodes[ch[k]].primitives_len == 0 && (best < 0 || area(ch[k]) > area(ch[best]))) best = (int)k;
            if (best < 0) break;
            const uint32_t c = ch[best];
            ch[best] = c + 1u;
            ch.insert(ch.begin() + best + 1, nodes[c].primitives_or_second_child_index);
        }
        WideNode w{};
        for (uint32_t c : ch) if (nodes[c].primitives_len == 0) w.ch[w.n_ch++] = c; // interior children first ...
        w.n_int = w.n_ch;
        for (uint32_t c : ch) if (nodes[c].primitives_len != 0) w.ch[w.n_ch++] = c; // ... then the leaves
        w.first_child = (uint32_t)queue.size(); // consecutive: allocated here and now
        for (uint32_t k = 0; k < w.n_int; k++) {
            const uint32_t idx = (uint32_t)queue.size();
            queue.push_back(w.ch[k]);
            level.push_back(level[qi] + 1u);
            wdepth = std::max(wdepth, level[qi] + 2u);
            if (by_area) {
                heap.push_back({area(w.ch[k]), idx});
                std::push_heap(heap.begin(), heap.end(), [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) { return a.first < b.first || (a.first == b.first && a.second > b.second); });
            }
        }
        if (wide.size() < queue.size()) wide.resize(queue.size());
        wide[qi] = w;
    }
    wide.resize(queue.size());
    if (wdepth > RT_WSTACK + RT_WSPILL + 1u || wide.size() >= (1u << 27)) { wide.clear(); return false; } // (deeper than the walk's stack: registers + overflow columns)
    if (wide_depth) *wide_depth = wdepth;
    // whole leaves, in the order the wide nodes list them
    prims_out.resize(n_prims);
    nodes_out.assign(nodes, nodes + n_nodes);
    if (old_of_new) old_of_new->resize(n_prims);
    uint32_t at = 0;
    for (const WideNode &w : wide)
        for (uint32_t k = w.n_int; k < w.n_ch; k++) {
            const rsrt_bvh_node &lf = nodes[w.ch[k]];
            for (uint32_t j = 0; j < lf.primitives_len; j++) {
                prims_out[at + j] = prims[lf.primitives_or_second_child_index + j];
                if (old_of_new) (*old_of_new)[at + j] = lf.primitives_or_second_child_index + j;
            }
            nodes_out[w.ch[k]].primitives_or_second_child_index = at;
            at += lf.primitives_len;
        }
    return true;
}
// the device records of the wide nodes (8 float4 each; rt_device.h, DevScene::wnodes) from the PERMUTED nodes / primitives
void fill_wide_nodes(const std::vector<WideNode> &wide, const rsrt_bvh_node *nodes, const rsrt_primitive_info *prims, float4 *out)
{
    for (size_t wi = 0; wi < wide.size(); wi++) {
        const WideNode &w = wide[wi];
        float4 *q = out + 8 * wi;
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // see DevScene::wnodes
        words[0] = w.first_child | (((1u << w.n_int) - 1u) << 26);
        if (w.n_ch > w.n_int) words[1] = nodes[w.ch[w.n_int]].primitives_or_second_child_index; // the leaves' records are contiguous from here
        for (uint32_t k = 0; k < 4; k++) {
            if (k >= w.n_ch) { q[2 * k] = q[2 * k + 1] = f4(0, 0, 0, 0); continue; } // (an empty slot's box may "hit": no mask names it)
            const rsrt_bvh_node &nd = nodes[w.ch[k]]; // the child's EXACT box
            q[2 * k] = f4(nd.bounds_min[0], nd.bounds_min[1], nd.bounds_min[2], 0);
            q[2 * k + 1] = f4(nd.bounds_max[0], nd.bounds_max[1], nd.bounds_max[2], 0);
            if (k < w.n_int) continue;
            const uint32_t off = nd.primitives_or_second_child_index - words[1]; // < 32: four leaves of at most eight records
            words[4 + k] = ((1u << nd.primitives_len) - 1u) << off;
            for (uint32_t j = 0; j < nd.primitives_len; j++) {
                const uint32_t ty = prims[nd.primitives_or_second_child_index + j].primitive_type;
                if (ty >= 2) words[2] |= 1u << (off + j);
                else if (ty == 1) words[3] |= 1u << (off + j);
            }
        }
        for (int k = 0; k < 8; k++) q[k].w = u2f(words[k]);
    }
}

rsrt_status ensure_accumulator(rsrt_context *ctx, uint32_t w, uint32_t h)
{
    if (ctx->accum && ctx->acc_w == w && ctx->acc_h == h) return RSRT_OK;
    if (ctx->accum && ctx->accum != ctx->accum_owned)
        return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bound accumulator is %ux%u but %ux%u was requested", ctx->acc_w, ctx->acc_h, w, h);
    if (ctx->accum_owned) { (void)hipFree(ctx->accum_owned); ctx->accum_owned = nullptr; ctx->accum = nullptr; }
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMalloc(&ctx->accum_owned, (size_t)w * h * sizeof(float4)));
    HIP_TRY(ctx, hipMemset(ctx->accum_owned, 0, (size_t)w * h * sizeof(float4)));
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->accum = ctx->accum_owned;
    ctx->acc_w = w;
    ctx->acc_h = h;
    return RSRT_OK;
}

// Orders `stream` after everything this context has enqueued so far (no-op on the same stream).
rsrt_status begin_work(rsrt_context *ctx, hipStream_t stream)
{
    if (ctx->last_valid && ctx->last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->last_event, 0));
    return RSRT_OK;
}
// Marks the end of what was just enqueued on `stream`.
rsrt_status end_work(rsrt_context *ctx, hipStream_t stream)
{
    HIP_TRY(ctx, hipEventRecord(ctx->last_event, stream));
    ctx->last_stream = stream;
    ctx->last_valid = true;
    return RSRT_OK;
}
// Waits for everything the context has enqueued, on whatever stream.
rsrt_status sync_all(rsrt_context *ctx)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &L : ctx->lanes)
        if (L.stream) HIP_TRY(ctx, hipStreamSynchronize(L.stream));
    if (ctx->last_valid) HIP_TRY(ctx, hipEventSynchronize(ctx->last_event));
    return RSRT_OK;
}
rsrt_status ensure_scratch(rsrt_context *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return RSRT_OK;
    rsrt_status st = sync_all(ctx);
    if (st) return st;
    if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return RSRT_OK;
}

hipEvent_t get_event(rsrt_context *ctx)
{
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Folds finished launches' HIP-event times into the cumulative totals (may run at any time, e.g. to
// bound the event pool) — it never touches the "since the previous rsrt_get_stats" window.
rsrt_status collect_events(rsrt_context *ctx)
{
    if (ctx->pending_events.empty() && ctx->pending_reduce.empty()) return RSRT_OK;
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    for (auto &re : ctx->pending_reduce) {
        float t = 0;
        HIP_TRY(ctx, hipEventSynchronize(re.end));
        HIP_TRY(ctx, hipEventElapsedTime(&t, re.begin, re.end));
        ctx->cum_reduce_ms += t;
        ctx->event_pool.push_back(re.begin);
        ctx->event_pool.push_back(re.end);
    }
    ctx->pending_reduce.clear();
    for (auto &pe : ctx->pending_events) {
        float t1 = 0, t2 = 0;
        HIP_TRY(ctx, hipEventSynchronize(pe.end));
        HIP_TRY(ctx, hipEventElapsedTime(&t1, pe.begin, pe.traced));
        HIP_TRY(ctx, hipEventElapsedTime(&t2, pe.traced, pe.end));
        ctx->cum_trace_ms += t1;
        ctx->cum_resolve_ms += t2;
        ctx->event_pool.push_back(pe.begin);
        ctx->event_pool.push_back(pe.traced);
        ctx->event_pool.push_back(pe.end);
    }
    ctx->pending_events.clear();
    return RSRT_OK;
}

// rsrt_get_stats: cumulative totals now, and the difference to the totals at the previous call.
rsrt_status collect_stats(rsrt_context *ctx)
{
    rsrt_status st = sync_all(ctx);
    if (st) return st;
    st = collect_events(ctx);
    if (st) return st;
    unsigned long long c[RT_STATS_WORDS] = {0}; // device counters are cumulative
    HIP_TRY(ctx, hipMemcpy(c, ctx->dev_stats, sizeof c, hipMemcpyDeviceToHost));
    rsrt_stats &s = ctx->stats;
    s.paths = c[0] - ctx->base_counts[0];
    s.ext_rays = c[1] - ctx->base_counts[1];
    s.shadow_rays = c[2] - ctx->base_counts[2];
    s.traversal_steps = c[3] - ctx->base_counts[3];
    s.trace_kernel_ms = ctx->cum_trace_ms - ctx->base_trace_ms;
    s.resolve_kernel_ms = ctx->cum_resolve_ms - ctx->base_resolve_ms;
    s.kernel_ms = s.trace_kernel_ms + s.resolve_kernel_ms;
    s.launches = ctx->cum_launches - ctx->base_launches;
    s.total_paths = c[0];
    s.total_ext_rays = c[1];
    s.total_shadow_rays = c[2];
    s.total_kernel_ms = ctx->cum_trace_ms + ctx->cum_resolve_ms;
    s.reduce_ms = ctx->cum_reduce_ms - ctx->base_reduce_ms;
    ctx->base_reduce_ms = ctx->cum_reduce_ms;
    for (int i = 0; i < 4; i++) ctx->base_counts[i] = c[i];
    ctx->base_trace_ms = ctx->cum_trace_ms;
    ctx->base_resolve_ms = ctx->cum_resolve_ms;
    ctx->base_launches = ctx->cum_launches;
    memcpy(ctx->debug_words, c + 4, sizeof ctx->debug_words);
    return RSRT_OK;
}


// Mid-size and big scenes (no whole image in LDS): what the chosen traversal keeps in LDS beside the pools (SceneViewHybrid) — the wide walk
// (trav 4) its image of nodes | records | shading arrays, the fixed-order walk (3) the top block of its elements (any prefix will do), the
// tree walks their nodes + escape links (all or nothing).  Sets the scene's lds_* fields; false: nothing is staged.
static bool hybrid_stage(const rsrt_context *ctx, DevScene &sc, int trav, uint32_t room_f4)
{
    sc.lds_wnodes = 0u;
    if (trav >= 4) {
        sc.lds_wnodes = std::min(ctx->wimg_nodes, room_f4 / 8u);
        if (sc.lds_wnodes == 0) return false;
        sc.lds_float4s = 8u * sc.lds_wnodes;
        sc.lds_hybrid = 3u;
        sc.lds_src = sc.wnodes; // a prefix of the array itself
        return true;
    }
    uint32_t head = trav == 3 ? ctx->hybrid_pnode_f4 : ctx->hybrid_head_f4;
    if (trav == 3) head = std::min(head, room_f4 / 2u * 2u);
    else if (head > room_f4) head = 0u;
    if (head == 0u) return false;
    sc.lds_float4s = head;
    sc.lds_hybrid = trav == 3 ? 2u : 1u;
    sc.lds_src = trav == 3 ? sc.pnodes : sc.nodes;
    return true;
}

// Which traversal TRACE runs (rt_wavepool.h, TRAV): the flat loop where the scene qualifies, else the fixed-order walk,
// else (leaves longer than 8 primitives) the generic tree walk; RSRT_TRAVERSAL / RSRT_FLAT cap the choice for A/B runs.
int select_traversal(const rsrt_context *ctx, const DevScene &sc, uint32_t max_bounces, uint32_t flags)
{
    if (ctx->max_traversal >= 2 && ctx->allow_flat && sc.flat_ok && max_bounces <= RT_FLAT_MAX_BOUNCES) return 2;
    // RSRT_FLAG_PRUNE wants the reference's near-child-first order: a close hit found early is what lets later boxes be
    // skipped (the fixed-order walk prunes 4 % of suzanne's steps, the near-first walk 8 %)
    if (ctx->max_traversal >= 6 && sc.wide_ok && sc.coop_ok && !(flags & RSRT_FLAG_PRUNE)) return 6; // the cooperative walk (rt_coop.h)
    if (ctx->max_traversal >= 4 && sc.wide_ok && !(flags & RSRT_FLAG_PRUNE)) return sc.wide_deep ? 5 : 4; // (5: the same walk with a stack that may overflow into memory)
    if (ctx->max_traversal >= 3 && sc.typed_leaves && !(flags & RSRT_FLAG_PRUNE)) return 3;
    if (ctx->max_traversal >= 1 && sc.typed_leaves) return 1;
    return 0;
}

// One pass of rsrt_render: the path-tracing kernel over P.sample_count samples, then the ordered resolve.
rsrt_status enqueue_pass(rsrt_context *ctx, RenderParams &P, const rsrt_context::PassEvents &pe, const void *kfn, uint32_t block, int bpc,
                         size_t smem, size_t per_sample, uint32_t max_bounces, hipStream_t caller_stream, rsrt_context::Lane &lane, uint32_t grid_div = 1)
{
    const uint32_t tile_px = P.tile_w * P.tile_h;
    // the path-tracing kernel: on the lane's stream (behind the resolve that last read this lane's sample buffer: same stream)
    const hipStream_t stream = ctx->overlap ? lane.stream : caller_stream;
    HIP_TRY(ctx, hipEventRecord(pe.begin, stream));
    if (max_bounces > 0) {
        // chunk = one tile x samples_per_chunk samples.  Up to 8 samples per chunk (2048 paths: the
        // counter is touched rarely), fewer when the job is small — e.g. one GPU's share of a
        // partitioned frame — so that every resident wave still gets >= ~32 chunks and the tail,
        // where waves run out of work at different times, stays a few percent.
        {
            const uint64_t waves = (uint64_t)ctx->cus * bpc * (block / RT_WAVE) / grid_div;
            const uint64_t want_chunks = (uint64_t)ctx->chunks_per_wave * waves;
            const uint64_t total_tile_samples = (uint64_t)P.n_owned_tiles * P.sample_count;
            uint64_t spc = total_tile_samples / std::max<uint64_t>(want_chunks, 1);
            spc = std::min<uint64_t>(std::max<uint64_t>(spc, 1), std::max<uint32_t>(1u, 2048u / tile_px));
            P.samples_per_chunk = (uint32_t)std::min<uint64_t>(spc, P.sample_count);
            // very small jobs (one sample per call, the reference's interactive mode): split tiles too,
            // down to one wave's worth of pixels per chunk
            P.n_subtiles = 1;
            while (P.samples_per_chunk == 1 && total_tile_samples * P.n_subtiles < want_chunks && tile_px / (P.n_subtiles * 2) >= RT_WAVE &&
                   tile_px % (P.n_subtiles * 2) == 0)
                P.n_subtiles *= 2;
            P.chunk_px = tile_px / P.n_subtiles;
        }
        P.n_sblocks = (P.sample_count + P.samples_per_chunk - 1) / P.samples_per_chunk;
        const uint64_t n_chunks = (uint64_t)P.n_owned_tiles * P.n_sblocks * P.n_subtiles;
        if (n_chunks > 0xffffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "too many work chunks");
        P.n_chunks = (uint32_t)n_chunks;
        HIP_TRY(ctx, hipMemsetAsync(lane.work_counter, 0, sizeof(unsigned int), stream));
        const uint32_t waves_wanted = (uint32_t)std::min<uint64_t>(n_chunks, 0x7fffffffull);
        const uint32_t wpb = block / RT_WAVE;
        uint32_t grid = std::min<uint32_t>((waves_wanted + wpb - 1) / wpb, std::max<uint32_t>(1u, (uint32_t)(ctx->cus * bpc) / grid_div));
        grid = std::max(grid, 1u);
        void *kargs[] = {&P};
        HIP_TRY(ctx, hipLaunchKernel(kfn, dim3(grid), dim3(block), kargs, smem, stream));
        ctx->cum_launches++;
    } else {
        HIP_TRY(ctx, hipMemsetAsync(lane.sample_buf, 0, per_sample * P.sample_count, stream));
    }
    HIP_TRY(ctx, hipEventRecord(pe.traced, stream));
    if (ctx->overlap) {
        HIP_TRY(ctx, hipEventRecord(lane.traced, stream));
        lane.traced_valid = true;
    }
    // The ordered resolve, behind the kernel on the lane's stream.  It adds into the accumulator, so it comes after (a) what the
    // caller's stream holds at this moment (a clear, a caller's own kernel on a bound accumulator) and (b) whatever the context
    // enqueued last on any stream — the previous pass's resolve above all: samples are added in increasing order.  Nothing is
    // put on the caller's stream that could wait there (several streams share a hardware queue: a wait parked in the caller's
    // queue holds up the lane that shares it — measured, 0.81 against 0.60 ms per single-sample call); the caller's stream is
    // ordered behind the resolve by rsrt_render once per call, if the caller named a stream of its own.
    if (ctx->overlap) {
        HIP_TRY(ctx, hipEventRecord(lane.caller_at, caller_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(stream, lane.caller_at, 0));
        if (ctx->last_valid && ctx->last_stream != stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->last_event, 0));
    }
    hipLaunchKernelGGL(rt_resolve_kernel, dim3((P.n_slots + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, stream, P, ctx->accum);
    HIP_TRY(ctx, hipGetLastError());
    ctx->cum_launches++;
    HIP_TRY(ctx, hipEventRecord(pe.end, stream));
    if (ctx->overlap) {
        HIP_TRY(ctx, hipEventRecord(lane.resolved, stream));
        lane.resolved_valid = true;
        rsrt_status st = end_work(ctx, stream); // the context's chain now ends on the lane's stream
        if (st) return st;
    }
    return RSRT_OK;
}

// the device-only packing of an uploaded environment (rt_alias_device.h, rt_env_pack_*): pmf copies in the texels' alpha and
// in the alias entries' pad words
rsrt_status pack_environment(rsrt_context *ctx, Env &e)
{
    const size_t n = (size_t)e.width * e.height;
    rsrt_status st = begin_work(ctx, ctx->stream);
    if (st) return st;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(rt_env_pack_texels_kernel, dim3(nb), dim3(256), 0, ctx->stream, e.rgba, e.alias, n);
    hipLaunchKernelGGL(rt_env_pack_alias_kernel, dim3(nb), dim3(256), 0, ctx->stream, e.alias, n);
    HIP_TRY(ctx, hipGetLastError());
    if ((st = end_work(ctx, ctx->stream))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RSRT_OK;
}

// rsrt_display_srgb8 on any W*H RGBA32F sum that lives on ctx's device (the multi-GPU frame buffer uses it too)
rsrt_status display_from(rsrt_context *ctx, const float4 *sum, uint32_t sample_total, uint8_t *host_rgba8, size_t n_bytes)
{
    DeviceGuard g(ctx->device);
    const size_t n = (size_t)ctx->acc_w * ctx->acc_h;
    if (!host_rgba8 || n_bytes != n * 4 || sample_total == 0) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "display_srgb8: expected %zu bytes and sample_total > 0", n * 4);
    rsrt_status st = ensure_scratch(ctx, n * sizeof(uchar4));
    if (st || (st = begin_work(ctx, ctx->stream))) return st;
    uchar4 *tmp = static_cast<uchar4 *>(ctx->scratch);
    hipLaunchKernelGGL(rt_display_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sum, n, sample_total, tmp);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(host_rgba8, tmp, n * sizeof(uchar4), hipMemcpyDeviceToHost, ctx->stream));
    if ((st = end_work(ctx, ctx->stream))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RSRT_OK;
}

} // namespace

extern "C" {

rsrt_status rsrt_context_create(int device_index, rsrt_context **out)
{
    if (!out) return fail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, RSRT_ERR_NO_DEVICE, "no HIP device available (%s); librsrt has no CPU path", hipGetErrorString(e));
    if (device_index < 0 || device_index >= n) return fail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "device_index %d out of range [0,%d)", device_index, n);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_index)) != hipSuccess) return fail(nullptr, RSRT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RSRT_ERR_NO_DEVICE, "device %d is %s; librsrt is built for gfx950 (MI355X) only", device_index, prop.gcnArchName);
    rsrt_context *ctx = new rsrt_context();
    ctx->device = device_index;
    ctx->cus = prop.multiProcessorCount;
    DeviceGuard g(device_index);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->last_event, hipEventDisableTiming)) != hipSuccess ||
        (e = [&] { hipError_t r = hipSuccess;
                   for (auto &L : ctx->lanes) {
                       if (r == hipSuccess) r = hipMalloc(&L.work_counter, sizeof(unsigned int));
                       if (r == hipSuccess) r = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
                       if (r == hipSuccess) r = hipEventCreateWithFlags(&L.resolved, hipEventDisableTiming);
                       if (r == hipSuccess) r = hipEventCreateWithFlags(&L.traced, hipEventDisableTiming);
                       if (r == hipSuccess) r = hipEventCreateWithFlags(&L.caller_at, hipEventDisableTiming);
                   }
                   return r; }()) != hipSuccess ||
        (e = hipMalloc(&ctx->dev_stats, RT_STATS_WORDS * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(ctx->dev_stats, 0, RT_STATS_WORDS * sizeof(unsigned long long))) != hipSuccess) {
        fail(nullptr, RSRT_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
        delete ctx;
        return RSRT_ERR_HIP;
    }
    for (int kv = 0; kv < RT_N_VARIANTS; kv++)
        for (int m = 0; m < 21; m++) (void)hipFuncSetAttribute(variant_function(kv, m / 7, m % 7), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (const char *kv = getenv("RSRT_KERNEL")) { // see kVariantPool
        int v = atoi(kv);
        if (v >= 0 && v < RT_N_VARIANTS) ctx->kernel_variant = v;
    }
    if (const char *sp = getenv("RSRT_SMALL_PATHS")) { long long v = atoll(sp); if (v >= 0) ctx->small_paths = (uint64_t)v; } // 0: no pipelining of small jobs (A/B)
    if (const char *pl = getenv("RSRT_PIPE_LANES")) { int v = atoi(pl); if (v >= 2 && v <= 8) ctx->n_lanes = (uint32_t)v; } // experiment knobs
    if (const char *pd = getenv("RSRT_PIPE_DIV")) { int v = atoi(pd); if (v >= 1 && v <= 8) ctx->pipe_div = (uint32_t)v; }
    if (const char *ov = getenv("RSRT_OVERLAP")) ctx->overlap = atoi(ov) != 0; // 0: one set of work buffers, every kernel on the caller's stream (A/B)
    if (const char *hy = getenv("RSRT_HYBRID")) ctx->allow_hybrid = atoi(hy) != 0; // 0: mid-size scenes read everything from global memory (A/B)
    if (const char *ty = getenv("RSRT_TRAVERSAL")) ctx->max_traversal = atoi(ty); // cap: 0 generic tree walk, 1 typed leaf loops, 2 + flat small-scene loop, 3 + fixed-order walk (A/B)
    if (const char *fl = getenv("RSRT_FLAT")) ctx->allow_flat = atoi(fl) != 0; // 0: small scenes take the walk a big scene would (A/B)
    if (const char *tb = getenv("RSRT_TRACE_BUDGET")) { int v = atoi(tb); if (v > 0) ctx->trace_budget = (uint32_t)v; }
    if (const char *dq = getenv("RSRT_DESCEND_QUORUM")) { int v = atoi(dq); if (v >= 0 && v <= 100) ctx->descend_quorum = (uint32_t)v; }
    if (const char *sq = getenv("RSRT_STOP_QUORUM")) { int v = atoi(sq); if (v >= 0 && v <= 100) ctx->stop_quorum = (uint32_t)v; }
    if (const char *cw = getenv("RSRT_CHUNKS_PER_WAVE")) { int v = atoi(cw); if (v >= 1 && v <= 4096) ctx->chunks_per_wave = (uint32_t)v; }
    if (const char *fq = getenv("RSRT_FLAT_QUORUM")) { int v = atoi(fq); if (v >= 0 && v <= 100) ctx->flat_quorum = (uint32_t)v; }
    if (const char *e2 = getenv("RSRT_SAMPLE_BUFFER_MB")) { long v = atol(e2); if (v > 0) ctx->sample_buffer_budget = (size_t)v << 20; }
    if (const char *pb = getenv("RSRT_PIPE_BLOCKS")) { int v = atoi(pb); if (v >= 1 && v <= 4) ctx->pipe_blocks = v; } // experiment knob
    if (const char *o = getenv("RSRT_BLOCKS_PER_CU")) { int v = atoi(o); if (v > 0) ctx->max_blocks_per_cu = v; } // experiment knob
    if (const char *pr = getenv("RSRT_PROBE_REPEAT")) { int v = atoi(pr); if (v > 1 && v <= 4096) ctx->probe_repeat = (uint32_t)v; }
    if (const char *cl = getenv("RSRT_COOP_LDS_CAP")) { int v = atoi(cl); if (v >= (int)RT_COOP_MIN_LDS_CAP && v <= (int)RT_COOP_NCAP) ctx->coop_lds_cap = (uint32_t)v; }
    if (const char *cf = getenv("RSRT_COOP_LIFO_AT")) { int v = atoi(cf); if (v >= 0 && v <= (int)RT_COOP_NARROW_AT) ctx->coop_lifo_at = (uint32_t)v; }
    if (const char *cn = getenv("RSRT_COOP_NARROW_AT")) { int v = atoi(cn); if (v >= 0 && v <= (int)RT_COOP_NARROW_AT) ctx->coop_narrow_at = (uint32_t)v; }
    if (const char *cm = getenv("RSRT_COMM_MODE")) ctx->comm_dense_mode = strcmp(cm, "reduce") == 0;
    if (const char *hq = getenv("GPU_MAX_HW_QUEUES")) { int v = atoi(hq); if (v > 0) ctx->hw_queues = v; }
    for (int m = 0; m < 21; m++) (void)hipFuncSetAttribute(probe_function(m / 7, m % 7), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    char buf[256];
    snprintf(buf, sizeof buf, "librsrt 0.1; %s (%s); %d CUs", prop.name, prop.gcnArchName, ctx->cus);
    ctx->description = buf;
    *out = ctx;
    return RSRT_OK;
}

void rsrt_context_destroy(rsrt_context *ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)sync_all(ctx);
    (void)rsrt_comm_destroy(ctx);
    for (auto &re : ctx->pending_reduce) { (void)hipEventDestroy(re.begin); (void)hipEventDestroy(re.end); }
    if (ctx->last_event) (void)hipEventDestroy(ctx->last_event);
    (void)hipFree(ctx->scratch);
    (void)hipFree(ctx->comm_buf);
    for (auto &pe : ctx->pending_events) { (void)hipEventDestroy(pe.begin); (void)hipEventDestroy(pe.traced); (void)hipEventDestroy(pe.end); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipFree(ctx->scene_blob);
    for (auto &e : ctx->envs) { (void)hipFree(e.rgba); (void)hipFree(e.alias); }
    (void)hipFree(ctx->accum_owned);
    for (auto &L : ctx->lanes) {
        (void)hipFree(L.sample_buf);
        (void)hipFree(L.cold_state);
        (void)hipFree(L.work_counter);
        if (L.resolved) (void)hipEventDestroy(L.resolved);
        if (L.traced) (void)hipEventDestroy(L.traced);
        if (L.caller_at) (void)hipEventDestroy(L.caller_at);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    (void)hipFree(ctx->dev_stats);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rsrt_last_error(const rsrt_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

#ifndef RSRT_BUILD_ID
#define RSRT_BUILD_ID "unknown"
#endif
const char *rsrt_build_id(void) { return RSRT_BUILD_ID; }

const char *rsrt_describe(rsrt_context *ctx) { return ctx ? ctx->description.c_str() : "librsrt 0.1"; }

rsrt_status rsrt_upload_scene(rsrt_context *ctx, const rsrt_material *materials, uint32_t n_materials, const rsrt_sphere *spheres,
                              uint32_t n_spheres, const rsrt_plane *planes, uint32_t n_planes, const rsrt_vec3 *vertices,
                              uint32_t n_vertices, const rsrt_vec3 *normals, uint32_t n_normals, const rsrt_triangle *triangles,
                              uint32_t n_triangles, const rsrt_primitive_info *primitives_in, uint32_t n_primitives,
                              const rsrt_bvh_node *nodes_in, uint32_t n_nodes)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    const rsrt_primitive_info *primitives = primitives_in; // (re-pointed below at a copy with whole leaves reordered, when the wide walk is built)
    const rsrt_bvh_node *nodes = nodes_in;
    if (n_nodes == 0 || !nodes) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh_nodes is empty");
    if (n_nodes >= RT_END) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh_nodes: %u nodes exceed the 25-bit traversal cursor", n_nodes);
    if ((n_materials && !materials) || (n_spheres && !spheres) || (n_planes && !planes) || (n_vertices && !vertices) ||
        (n_normals && !normals) || (n_triangles && !triangles) || (n_primitives && !primitives))
        return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "a non-empty array has a NULL pointer");
    if (n_materials >= (1u << 30)) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "too many materials");
    // ---- validate every index the kernels will follow
    for (uint32_t i = 0; i < n_spheres; i++)
        if (spheres[i].material_id >= n_materials) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "sphere %u: material_id %u out of range", i, spheres[i].material_id);
    for (uint32_t i = 0; i < n_planes; i++)
        if (planes[i].material_id >= n_materials) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "plane %u: material_id %u out of range", i, planes[i].material_id);
    for (uint32_t i = 0; i < n_triangles; i++) {
        const rsrt_triangle &t = triangles[i];
        if (t.material_id >= n_materials) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "triangle %u: material_id %u out of range", i, t.material_id);
        if (t.vertex_0 >= n_vertices || t.vertex_1 >= n_vertices || t.vertex_2 >= n_vertices)
            return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "triangle %u: vertex index out of range", i);
        if (t.normal_0 >= n_normals || t.normal_1 >= n_normals || t.normal_2 >= n_normals)
            return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "triangle %u: normal index out of range", i);
    }
    for (uint32_t i = 0; i < n_primitives; i++) {
        const rsrt_primitive_info &p = primitives[i];
        uint32_t lim = p.primitive_type == 0 ? n_spheres : (p.primitive_type == 1 ? n_planes : (p.primitive_type == 2 ? n_triangles : 0));
        if (p.index >= lim) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "primitive %u: type %u index %u out of range", i, p.primitive_type, p.index);
    }
    // tree shape: pre-order layout (children after their parent) => traversal terminates
    uint32_t depth = 0;
    {
        std::vector<std::pair<uint32_t, uint32_t>> st; // node, depth
        std::vector<uint8_t> seen(n_nodes, 0);
        st.push_back({0u, 0u});
        while (!st.empty()) {
            auto [i, d] = st.back();
            st.pop_back();
            if (seen[i]) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh node %u reachable twice", i);
            seen[i] = 1;
            depth = std::max(depth, d);
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len > 0) {
                if (nd.primitives_len > 0xffffu || (uint64_t)nd.primitives_or_second_child_index + nd.primitives_len > n_primitives)
                    return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh leaf %u: primitive range out of bounds", i);
            } else {
                uint32_t second = nd.primitives_or_second_child_index;
                if (nd.split_axis > 2 || i + 1 >= n_nodes || second <= i + 1 || second >= n_nodes)
                    return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh interior node %u: bad child index / axis", i);
                st.push_back({second, d + 1});
                st.push_back({i + 1, d + 1});
            }
        }
        for (uint32_t i = 0; i < n_nodes; i++) // (the tables built below are indexed by every node)
            if (!seen[i]) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh node %u is not reachable from the root", i);
    }
    // ---- wide walk (rt_device.h, trace_wide): the binary tree collapsed into 4-wide nodes, a node's interior children consecutive.  Needs what makes
    // skipping binary nodes exact — every child box inside its parent's — plus leaves of <= 8 records that share none (the
    // records of a wide node's leaf children are made contiguous by moving whole leaves: `primitives` and the leaves' first
    // indices are re-pointed at permuted copies, and everything below is built from those; a record's index is internal to
    // the device image, ties are decided by the visiting ranks computed from the tree) and a wide tree of at most
    // RT_WSTACK + RT_WSPILL + 1 levels (the walk's stack: eight registers and sixteen overflow words).
    std::vector<WideNode> wide;
    std::vector<rsrt_primitive_info> prims_perm;
    std::vector<rsrt_bvh_node> nodes_perm;
    uint32_t wide_depth = 0;
    const bool wide_ok = build_wide_tree(nodes, n_nodes, primitives, n_primitives, wide, prims_perm, nodes_perm, nullptr, &wide_depth);
    if (wide_ok) {
        primitives = prims_perm.data();
        nodes = nodes_perm.data();
    }
    // ---- threaded traversal links: escape[octant][node] (rt_device.h, trace_threaded)
    std::vector<uint32_t> escape(8ull * n_nodes, RT_END);
    for (uint32_t q = 0; q < 8; q++) {
        std::vector<std::pair<uint32_t, uint32_t>> st; // node, its escape
        st.push_back({0u, RT_END});
        while (!st.empty()) {
            auto [i, esc] = st.back();
            st.pop_back();
            escape[(size_t)q * n_nodes + i] = esc;
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len == 0) {
                const bool far_first = (q >> nd.split_axis) & 1u; // sign(inv_dir[axis]) < 0: second child is nearer
                const uint32_t first = i + 1, second = nd.primitives_or_second_child_index;
                const uint32_t near_c = far_first ? second : first, far_c = far_first ? first : second;
                st.push_back({near_c, far_c});
                st.push_back({far_c, esc});
            }
        }
    }
    const size_t esc_f4 = (8ull * n_nodes + 3) / 4;
    // ---- flat small-scene traversal (rt_device.h, trace_flat): leaf list, record masks, per-octant visiting ranks
    std::vector<uint32_t> leaf_nodes;
    // ... at most 64 records of each kind (a hit is carried as a 6-bit record index + 2-bit source in the idle cursor
    // bits), few enough leaves that testing every leaf box beats the tree walk, and (below) nested boxes and leaves
    // that share no record: the rank table gives every record ONE position in the visiting order
    bool flat_ok = n_primitives <= 64 && n_spheres <= 64 && n_planes <= 64;
    {
        std::vector<uint8_t> covered(n_primitives, 0);
        for (uint32_t i = 0; i < n_nodes && flat_ok; i++)
            for (uint32_t k = 0; k < nodes[i].primitives_len && flat_ok; k++) {
                uint8_t &c = covered[nodes[i].primitives_or_second_child_index + k];
                flat_ok = c == 0;
                c = 1;
            }
    }
    for (uint32_t i = 0; i < n_nodes; i++) {
        const rsrt_bvh_node &nd = nodes[i];
        if (nd.primitives_len != 0) { leaf_nodes.push_back(i); continue; }
        for (uint32_t c : {i + 1u, nd.primitives_or_second_child_index}) // every child box inside its parent's
            for (int k = 0; k < 3; k++)
                flat_ok = flat_ok && nodes[c].bounds_min[k] >= nd.bounds_min[k] && nodes[c].bounds_max[k] <= nd.bounds_max[k];
    }
    flat_ok = flat_ok && leaf_nodes.size() <= 32;
    std::vector<uint32_t> flat_rank(8 * 16, 0u);
    uint64_t tri_mask = 0, plane_mask = 0;
    if (flat_ok) {
        for (uint32_t p = 0; p < n_primitives; p++) {
            if (primitives[p].primitive_type >= 2) tri_mask |= 1ull << p;
            else if (primitives[p].primitive_type == 1) plane_mask |= 1ull << p;
        }
        for (uint32_t q = 0; q < 8; q++) { // the reference's near-child-first walk for this sign octant, every box "hit"
            std::vector<uint32_t> st{0u};
            uint32_t pos = 0;
            while (!st.empty()) {
                const uint32_t i = st.back();
                st.pop_back();
                const rsrt_bvh_node &nd = nodes[i];
                if (nd.primitives_len != 0) {
                    for (uint32_t k = 0; k < nd.primitives_len; k++, pos++) {
                        const uint32_t rec = nd.primitives_or_second_child_index + k;
                        flat_rank[q * 16 + (rec >> 2)] |= pos << (8 * (rec & 3));
                    }
                } else {
                    const bool far_first = (q >> nd.split_axis) & 1u;
                    const uint32_t first = i + 1, second = nd.primitives_or_second_child_index;
                    st.push_back(far_first ? first : second); // far child: visited after the near one
                    st.push_back(far_first ? second : first);
                }
            }
        }
    }
    const size_t flat_f4 = flat_ok ? 2 * leaf_nodes.size() : 0, rank_f4 = flat_ok ? 32 : 0;
    // ---- fixed-order traversal (rt_device.h, trace_preorder): per-octant visiting rank of EVERY primitive record — the
    // position at which the reference's near-child-first walk meets it — which decides equal t whatever order the records
    // are really tested in; and the skip links of the pre-order node array (next node once an interior node is missed).
    std::vector<uint32_t> prim_rank(8ull * n_primitives, 0xffffffffu); // (a record that several leaves share keeps the rank of its FIRST visit: the reference's strict < keeps the first of equals)
    for (uint32_t q = 0; q < 8; q++) {
        std::vector<uint32_t> st{0u};
        uint32_t pos = 0;
        while (!st.empty()) {
            const uint32_t i = st.back();
            st.pop_back();
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len != 0) {
                for (uint32_t k = 0; k < nd.primitives_len; k++, pos++) {
                    uint32_t &rk = prim_rank[(size_t)q * n_primitives + nd.primitives_or_second_child_index + k];
                    rk = std::min(rk, pos);
                }
            } else {
                const bool far_first = (q >> nd.split_axis) & 1u;
                const uint32_t first = i + 1, second = nd.primitives_or_second_child_index;
                st.push_back(far_first ? first : second);
                st.push_back(far_first ? second : first);
            }
        }
    }
    // The walk's node array ("pnodes").  Every interior node carries BOTH its links explicitly — where to go when its box
    // is hit (its first child) and when it is missed (the pre-order successor of its subtree) — and a leaf is followed by
    // its successor, so the array may be laid out in any order.  It is laid out in two levels: first a TOP BLOCK, the
    // nodes a ray is most likely to meet (greedy by box surface area from the root: a child's box lies inside its
    // parent's, so the set is closed under "parent of"), in their pre-order; then everything else, in pre-order.  The
    // kernel keeps the top block in LDS (one copy per CU) and reads the rest from global memory: on the 15,488-triangle
    // grid scene 1,400 of 8,731 nodes take ~80 % of the box tests.  Where a leaf's successor is not the next array
    // element (a subtree leaves or re-enters the top block) a JUMP element is inserted: an interior-type record whose
    // two links are equal and whose box is never tested.
    struct PNode { uint32_t old; uint32_t jump_to; }; // old == UINT32_MAX: a jump element to old node `jump_to` (n_nodes = end)
    std::vector<uint32_t> succ(n_nodes, n_nodes); // pre-order successor of node i's subtree
    {
        std::vector<std::pair<uint32_t, uint32_t>> st; // node, what follows its subtree
        st.push_back({0u, n_nodes});
        while (!st.empty()) {
            auto [i, after] = st.back();
            st.pop_back();
            succ[i] = after;
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len == 0) {
                st.push_back({i + 1, nd.primitives_or_second_child_index}); // the first child's subtree ends where the second child begins
                st.push_back({nd.primitives_or_second_child_index, after});
            }
        }
    }
    const uint32_t kTopMax = kHybridRoomF4 / 2u; // top-block elements that fit beside sixteen path pools (32 B each)
    std::vector<PNode> plist;
    uint32_t top_elems = 0;
    for (uint32_t want = std::min<uint32_t>(n_nodes, kTopMax);; want = want * 7 / 8) {
        std::vector<uint8_t> in_top(n_nodes, 0);
        { // greedy by surface area
            auto area = [&](uint32_t i) {
                const double dx = (double)nodes[i].bounds_max[0] - nodes[i].bounds_min[0], dy = (double)nodes[i].bounds_max[1] - nodes[i].bounds_min[1],
                             dz = (double)nodes[i].bounds_max[2] - nodes[i].bounds_min[2];
                const double a = dx * dy + dy * dz + dz * dx;
                return a == a ? a : 0.0;
            };
            std::vector<std::pair<double, uint32_t>> heap;
            heap.push_back({area(0), 0u});
            uint32_t taken = 0;
            while (!heap.empty() && taken < want) {
                std::pop_heap(heap.begin(), heap.end());
                const uint32_t i = heap.back().second;
                heap.pop_back();
                in_top[i] = 1;
                taken++;
                if (nodes[i].primitives_len == 0)
                    for (uint32_t c : {i + 1u, nodes[i].primitives_or_second_child_index}) {
                        heap.push_back({area(c), c});
                        std::push_heap(heap.begin(), heap.end());
                    }
            }
        }
        plist.clear();
        for (int pass = 0; pass < 2; pass++) { // top block, then the rest; both in pre-order (= index order)
            std::vector<uint32_t> order;
            for (uint32_t i = 0; i < n_nodes; i++)
                if ((in_top[i] != 0) == (pass == 0)) order.push_back(i);
            for (size_t k = 0; k < order.size(); k++) {
                const uint32_t i = order[k];
                plist.push_back({i, 0u});
                const uint32_t follows = k + 1 < order.size() ? order[k + 1] : 0xffffffffu; // (the rest's first element never follows the top block's last)
                if (nodes[i].primitives_len != 0 && succ[i] != follows) plist.push_back({0xffffffffu, succ[i]});
            }
            if (pass == 0) top_elems = (uint32_t)plist.size();
        }
        if (top_elems <= kTopMax || want <= 1) break;
    }
    const uint32_t n_pnodes = (uint32_t)plist.size();
    if (n_pnodes >= RT_END) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh_nodes: %u traversal elements exceed the 25-bit cursor", n_pnodes);
    std::vector<uint32_t> new_id(n_nodes + 1, n_pnodes); // old node -> element; [n_nodes] = end
    for (uint32_t e = 0; e < n_pnodes; e++)
        if (plist[e].old != 0xffffffffu) new_id[plist[e].old] = e;
    const size_t pnode_f4 = 2ull * n_pnodes, prank_f4 = (8ull * n_primitives + 3) / 4, wnode_f4 = 8ull * wide.size();
    // the wide walks' LDS image: the head of the wide-node array (hottest first), as far as the room beside the launch's pools goes (hybrid_stage)
    const uint32_t wimg_nodes = wide_ok ? (uint32_t)wide.size() : 0u;
    // ---- build the device image: nodes | prims | escape links | tri normals | materials | fb spheres | fb planes
    const size_t n_f4 = 2ull * n_nodes + 4ull * n_primitives + 3ull * n_triangles + 4ull * n_materials + 4ull * n_spheres + 4ull * n_planes + esc_f4 + flat_f4;
    std::vector<float4> img(n_f4 + rank_f4 + pnode_f4 + prank_f4 + wnode_f4); // what follows the LDS image: flat ranks | pre-order nodes | record ranks | wide nodes
    float4 *p = img.data();
    float4 *p_nodes = p;
    bool typed_leaves = true;
    for (uint32_t i = 0; i < n_nodes; i++) typed_leaves = typed_leaves && nodes[i].primitives_len <= 8;
    for (uint32_t i = 0; i < n_nodes; i++, p += 2) {
        const rsrt_bvh_node &nd = nodes[i];
        p[0] = f4(nd.bounds_min[0], nd.bounds_min[1], nd.bounds_min[2], u2f(nd.primitives_or_second_child_index));
        uint32_t hi = nd.split_axis; // interior: split axis; leaf (when every leaf has <= 8 primitives): triangle mask | plane mask << 8
        if (nd.primitives_len != 0) {
            hi = 0;
            if (typed_leaves)
                for (uint32_t k = 0; k < nd.primitives_len; k++) {
                    const uint32_t ty = primitives[nd.primitives_or_second_child_index + k].primitive_type;
                    if (ty >= 2) hi |= 1u << k;
                    else if (ty == 1) hi |= 1u << (8 + k);
                }
        }
        p[1] = f4(nd.bounds_max[0], nd.bounds_max[1], nd.bounds_max[2], u2f(nd.primitives_len | (hi << 16)));
    }
    float4 *p_esc = p;
    memcpy(p_esc, escape.data(), escape.size() * sizeof(uint32_t));
    p += esc_f4;
    const size_t traversal_head_f4 = (size_t)(p - img.data()); // nodes | escape links: what the hybrid view keeps in LDS
    float4 *p_prims = p;
    for (uint32_t i = 0; i < n_primitives; i++, p += 4) {
        const rsrt_primitive_info &pi = primitives[i];
        if (pi.primitive_type == 0) make_sphere_record(spheres[pi.index], p);
        else if (pi.primitive_type == 1) make_plane_record(planes[pi.index], p);
        else make_triangle_record(triangles[pi.index], pi.index, vertices, p);
    }
    float4 *p_trin = p;
    for (uint32_t i = 0; i < n_triangles; i++, p += 3) {
        const float *a = normals[triangles[i].normal_0].v, *b = normals[triangles[i].normal_1].v, *c = normals[triangles[i].normal_2].v;
        p[0] = f4(a[0], a[1], a[2], b[0]);
        p[1] = f4(b[1], b[2], c[0], c[1]);
        p[2] = f4(c[2], 0, 0, 0);
    }
    float4 *p_mats = p;
    for (uint32_t i = 0; i < n_materials; i++, p += 4) make_material_record(materials[i], p);
    float4 *p_fbs = p;
    for (uint32_t i = 0; i < n_spheres; i++, p += 4) make_sphere_record(spheres[i], p);
    float4 *p_fbp = p;
    for (uint32_t i = 0; i < n_planes; i++, p += 4) make_plane_record(planes[i], p);
    float4 *p_flat = p;
    if (flat_ok) {
        for (uint32_t i : leaf_nodes) {
            const rsrt_bvh_node &nd = nodes[i];
            uint64_t lm = (nd.primitives_len >= 64 ? ~0ull : ((1ull << nd.primitives_len) - 1ull)) << nd.primitives_or_second_child_index;
            // A record whose geometry is bit for bit that of an EARLIER record of the same leaf can never win: it gives the very
            // same t, and the reference keeps the first of equals (strict <, a leaf's records in index order whatever the
            // octant).  It is left out of the leaf's mask — house.toml lists its ground plane twice.
            for (uint32_t b = 1; b < nd.primitives_len; b++)
                for (uint32_t a = 0; a < b; a++)
                    if (same_test(p_prims + 4 * (nd.primitives_or_second_child_index + a), p_prims + 4 * (nd.primitives_or_second_child_index + b))) {
                        lm &= ~(1ull << (nd.primitives_or_second_child_index + b));
                        break;
                    }
            p[0] = f4(nd.bounds_min[0], nd.bounds_min[1], nd.bounds_min[2], u2f((uint32_t)lm));
            p[1] = f4(nd.bounds_max[0], nd.bounds_max[1], nd.bounds_max[2], u2f((uint32_t)(lm >> 32)));
            p += 2;
        }
        memcpy(p, flat_rank.data(), flat_rank.size() * sizeof(uint32_t));
    }
    // The cooperative walk (rt_coop.h) folds an extension ray's hits with an atomic minimum and sends a ray that sees two records at the same
    // closest t through the exact walk once more.  A record that repeats an EARLIER record of its leaf bit for bit gives that record's t to every
    // ray and can never win (the reference keeps the first of equals; a leaf's records are met in index order whatever the octant): it is
    // marked in a word no test reads (r2.w) and counts as a miss there, so that doubled geometry does not make every hit a tie.
    {
        std::vector<uint32_t> twins;
        for (uint32_t i = 0; i < n_nodes; i++) {
            const rsrt_bvh_node &nd = nodes[i];
            for (uint32_t b = 1; b < nd.primitives_len; b++)
                for (uint32_t a = 0; a < b; a++)
                    if (same_test(p_prims + 4 * (nd.primitives_or_second_child_index + a), p_prims + 4 * (nd.primitives_or_second_child_index + b))) {
                        twins.push_back(nd.primitives_or_second_child_index + b);
                        break;
                    }
        }
        for (uint32_t r : twins) p_prims[4 * r + 2].w = u2f(1u); // (afterwards: same_test compares these words too)
    }
    float4 *p_pnodes = img.data() + n_f4 + rank_f4;
    for (uint32_t e = 0; e < n_pnodes; e++) {
        if (plist[e].old == 0xffffffffu) { // jump element: both links equal, box never looked at
            const uint32_t to = new_id[plist[e].jump_to];
            p_pnodes[2 * e] = f4(0, 0, 0, u2f(to));
            p_pnodes[2 * e + 1] = f4(0, 0, 0, u2f(to));
            continue;
        }
        const uint32_t i = plist[e].old;
        p_pnodes[2 * e] = p_nodes[2 * i];
        p_pnodes[2 * e + 1] = p_nodes[2 * i + 1];
        if (nodes[i].primitives_len == 0) { // interior: {min, link when hit}{max, link when missed}
            p_pnodes[2 * e].w = u2f(new_id[i + 1]);
            p_pnodes[2 * e + 1].w = u2f(new_id[succ[i]]);
        } else { // leaf: first record | leaf flag; length | triangle mask << 16 | plane mask << 24 (as in `nodes`)
            p_pnodes[2 * e].w = u2f(nodes[i].primitives_or_second_child_index | 0x80000000u);
        }
    }
    if (n_primitives) memcpy(p_pnodes + pnode_f4, prim_rank.data(), prim_rank.size() * sizeof(uint32_t));
    fill_wide_nodes(wide, nodes, primitives, p_pnodes + pnode_f4 + prank_f4);

    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    if (ctx->scene_blob) { (void)hipFree(ctx->scene_blob); ctx->scene_blob = nullptr; }
    ctx->scene_ready = false;
    HIP_TRY(ctx, hipMalloc(&ctx->scene_blob, img.size() * sizeof(float4)));
    HIP_TRY(ctx, hipMemcpy(ctx->scene_blob, img.data(), img.size() * sizeof(float4), hipMemcpyHostToDevice));
    DevScene &sc = ctx->scene;
    sc.nodes = ctx->scene_blob + (p_nodes - img.data());
    sc.prims = ctx->scene_blob + (p_prims - img.data());
    sc.tri_normals = ctx->scene_blob + (p_trin - img.data());
    sc.materials = ctx->scene_blob + (p_mats - img.data());
    sc.fb_spheres = ctx->scene_blob + (p_fbs - img.data());
    sc.fb_planes = ctx->scene_blob + (p_fbp - img.data());
    sc.escape = ctx->scene_blob + (p_esc - img.data());
    sc.flat_leaves = ctx->scene_blob + (p_flat - img.data());
    sc.flat_rank = reinterpret_cast<const uint32_t *>(ctx->scene_blob + n_f4);
    sc.pnodes = ctx->scene_blob + n_f4 + rank_f4;
    sc.n_pnodes = n_pnodes;
    sc.prim_rank = reinterpret_cast<const uint32_t *>(ctx->scene_blob + n_f4 + rank_f4 + pnode_f4);
    sc.wnodes = ctx->scene_blob + n_f4 + rank_f4 + pnode_f4 + prank_f4;
    sc.n_wnodes = (uint32_t)wide.size();
    sc.wide_ok = wide_ok ? 1u : 0u;
    sc.wide_deep = (wide_ok && wide_depth > RT_WSTACK + 1u) ? 1u : 0u;
    sc.coop_ok = (wide_ok && n_primitives < RT_COOP_MAX_RECORDS && wide.size() < RT_COOP_MAX_NODES) ? 1u : 0u; // (build_wide_tree already bounds the depth at 25 wide levels)
    sc.lds_wnodes = 0u; // (set per launch, hybrid_stage)
    ctx->wimg_nodes = wimg_nodes;
    sc.flat_ok = flat_ok ? 1u : 0u;
    sc.n_cull = 0; sc.cull_always = 0xffffffffu;
    if (flat_ok) { // the interior nodes two levels below the root and the leaves under each (rt_device.h, RT_FLAT_CULL)
        std::vector<uint32_t> leaf_index(n_nodes, 0u);
        for (size_t k = 0; k < leaf_nodes.size(); k++) leaf_index[leaf_nodes[k]] = (uint32_t)k;
        std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}}; // node, depth
        uint32_t always = 0u;
        const uint32_t cull_depth = 2u; // (1 / 2 / 3 below the root measured alike, profiles/r03_fusion_ab.txt)
        while (!st.empty()) {
            auto [i, d] = st.back();
            st.pop_back();
            const rsrt_bvh_node &nd = nodes[i];
            if (nd.primitives_len != 0) { always |= 1u << leaf_index[i]; continue; }
            if (d == cull_depth && sc.n_cull < 8) {
                uint32_t m = 0;
                std::vector<uint32_t> sub{i};
                while (!sub.empty()) {
                    const uint32_t j = sub.back();
                    sub.pop_back();
                    if (nodes[j].primitives_len != 0) m |= 1u << leaf_index[j];
                    else { sub.push_back(j + 1); sub.push_back(nodes[j].primitives_or_second_child_index); }
                }
                for (int k = 0; k < 3; k++) { sc.cull_min[sc.n_cull][k] = nd.bounds_min[k]; sc.cull_max[sc.n_cull][k] = nd.bounds_max[k]; }
                sc.cull_mask[sc.n_cull++] = m;
                continue;
            }
            st.push_back({i + 1, d + 1});
            st.push_back({nd.primitives_or_second_child_index, d + 1});
        }
        sc.cull_always = always;
    }
    sc.n_leaves = (uint32_t)leaf_nodes.size();
    sc.tri_mask_lo = (uint32_t)tri_mask; sc.tri_mask_hi = (uint32_t)(tri_mask >> 32);
    sc.plane_mask_lo = (uint32_t)plane_mask; sc.plane_mask_hi = (uint32_t)(plane_mask >> 32);
    sc.n_nodes = n_nodes; sc.n_prims = n_primitives; sc.n_tris = n_triangles; sc.n_materials = n_materials;
    sc.n_spheres = n_spheres; sc.n_planes = n_planes;
    sc.stack_entries = depth + 1;
    sc.typed_leaves = typed_leaves ? 1u : 0u;
    const size_t stack_bytes = (size_t)sc.stack_entries * RT_BLOCK * sizeof(uint32_t);
    if (stack_bytes > 128 * 1024) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bvh depth %u exceeds the supported traversal stack", depth);
    sc.lds_float4s = (n_f4 * sizeof(float4) <= 24 * 1024) ? (uint32_t)n_f4 : 0u; // whole image in LDS only while it leaves room for the path pools
    sc.lds_hybrid = 0;
    sc.lds_src = sc.nodes; // the image starts with the nodes
    ctx->hybrid_head_f4 = ctx->hybrid_pnode_f4 = 0;
    if (sc.lds_float4s == 0) { // mid-size: what every box step touches goes to LDS, one big workgroup per CU shares the copy
        if (traversal_head_f4 * sizeof(float4) <= 40 * 1024) ctx->hybrid_head_f4 = (uint32_t)traversal_head_f4; // nodes + escape links (tree walks; the render call checks the room beside its pools)
        ctx->hybrid_pnode_f4 = 2u * top_elems; // the top block of the fixed-order walk's nodes (all of them when the scene is mid-size)
    } else {
        ctx->wimg_nodes = 0; // (a small scene runs from its whole image)
    }
    memset(ctx->blocks_per_cu, 0, sizeof ctx->blocks_per_cu); // the kernels' dynamic LDS size depends on the scene: occupancy is asked for again
    ctx->scene_ready = true;
    return RSRT_OK;
}

// build_bvh (reference src/bvh.rs:13-337) on the device: csrc/hip/rt_bvh_device.h.  Same arguments and outputs as the host
// builder rsrt_build_bvh (include/rsrt_host.h) — host arrays in, host arrays out; the inputs are uploaded, the tree is built by
// kernels, the result is downloaded — plus the time the device spent building (events around the kernels, uploads excluded).
rsrt_status rsrt_build_bvh_device(rsrt_context *ctx, const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                                  const rsrt_vec3 *vertices, uint32_t n_vertices, const rsrt_triangle *triangles, uint32_t n_triangles,
                                  rsrt_primitive_info *primitives_out, rsrt_bvh_node *nodes_out, uint32_t *n_nodes_out, uint32_t *depth_out, double *build_ms_out)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    const uint64_t n64 = (uint64_t)n_spheres + n_planes + n_triangles;
    if (n64 == 0) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: empty scene"); // (the reference asserts, src/bvh.rs:222)
    if (n64 > 0x3fffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: too many primitives");
    if (!primitives_out || !nodes_out || (n_spheres && !spheres) || (n_planes && !planes) || (n_triangles && (!triangles || !vertices)))
        return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: NULL array");
    for (uint32_t i = 0; i < n_triangles; i++)
        if (triangles[i].vertex_0 >= n_vertices || triangles[i].vertex_1 >= n_vertices || triangles[i].vertex_2 >= n_vertices)
            return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: triangle %u: vertex index out of range", i);
    const uint32_t n = (uint32_t)n64;
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    // one allocation: inputs | items x 2 (3 float4 arrays each) | build nodes | tasks x 2 | pre, holes, backs | counters | outputs
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_sph = take((size_t)n_spheres * sizeof(rsrt_sphere)), o_pl = take((size_t)n_planes * sizeof(rsrt_plane_desc)),
                 o_v = take((size_t)n_vertices * sizeof(rsrt_vec3)), o_tri = take((size_t)n_triangles * sizeof(rsrt_triangle));
    size_t o_items[6];
    for (auto &o : o_items) o = take((size_t)n * sizeof(float4));
    const size_t o_nodes = take((size_t)2 * n * sizeof(BvhNodeB)), o_t0 = take((size_t)n * sizeof(BvhTask)), o_t1 = take((size_t)n * sizeof(BvhTask));
    const size_t o_pre = take((size_t)n * 4), o_holes = take((size_t)n * 4), o_backs = take((size_t)n * 4), o_cnt = take(16);
    const size_t o_outn = take((size_t)2 * n * sizeof(rsrt_bvh_node)), o_outp = take((size_t)n * sizeof(rsrt_primitive_info));
    char *blob = nullptr;
    HIP_TRY(ctx, hipMalloc(&blob, off));
    struct Free { char *p; ~Free() { (void)hipFree(p); } } free_blob{blob};
    hipStream_t q = ctx->stream;
    if (n_spheres) HIP_TRY(ctx, hipMemcpyAsync(blob + o_sph, spheres, (size_t)n_spheres * sizeof(rsrt_sphere), hipMemcpyHostToDevice, q));
    if (n_planes) HIP_TRY(ctx, hipMemcpyAsync(blob + o_pl, planes, (size_t)n_planes * sizeof(rsrt_plane_desc), hipMemcpyHostToDevice, q));
    if (n_vertices) HIP_TRY(ctx, hipMemcpyAsync(blob + o_v, vertices, (size_t)n_vertices * sizeof(rsrt_vec3), hipMemcpyHostToDevice, q));
    if (n_triangles) HIP_TRY(ctx, hipMemcpyAsync(blob + o_tri, triangles, (size_t)n_triangles * sizeof(rsrt_triangle), hipMemcpyHostToDevice, q));
    BvhItems A{reinterpret_cast<float4 *>(blob + o_items[0]), reinterpret_cast<float4 *>(blob + o_items[1]), reinterpret_cast<float4 *>(blob + o_items[2])};
    BvhItems B{reinterpret_cast<float4 *>(blob + o_items[3]), reinterpret_cast<float4 *>(blob + o_items[4]), reinterpret_cast<float4 *>(blob + o_items[5])};
    BvhNodeB *bnodes = reinterpret_cast<BvhNodeB *>(blob + o_nodes);
    BvhTask *tasks[2] = {reinterpret_cast<BvhTask *>(blob + o_t0), reinterpret_cast<BvhTask *>(blob + o_t1)};
    uint32_t *counters = reinterpret_cast<uint32_t *>(blob + o_cnt);
    hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
    struct Back { rsrt_context *c; hipEvent_t a, b; ~Back() { c->event_pool.push_back(a); c->event_pool.push_back(b); } } back{ctx, e0, e1};
    HIP_TRY(ctx, hipEventRecord(e0, q));
    hipLaunchKernelGGL(rt_bvh_items_kernel, dim3((n + 255) / 256), dim3(256), 0, q, reinterpret_cast<const rsrt_sphere *>(blob + o_sph), n_spheres,
                       reinterpret_cast<const rsrt_plane_desc *>(blob + o_pl), n_planes, reinterpret_cast<const rsrt_vec3 *>(blob + o_v),
                       reinterpret_cast<const rsrt_triangle *>(blob + o_tri), n_triangles, A);
    HIP_TRY(ctx, hipMemcpyAsync(B.bmin, A.bmin, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, q)); // (a root that is a leaf never writes the other copy)
    HIP_TRY(ctx, hipMemcpyAsync(B.bmax, A.bmax, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, q));
    HIP_TRY(ctx, hipMemcpyAsync(B.cen, A.cen, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, q));
    const BvhTask root{0u, n, 0u};
    uint32_t cnt[4] = {1u, 0u, 0u, 0u}; // build nodes allocated, tasks of the next level, error
    HIP_TRY(ctx, hipMemcpyAsync(tasks[0], &root, sizeof root, hipMemcpyHostToDevice, q));
    HIP_TRY(ctx, hipMemcpyAsync(counters, cnt, sizeof cnt, hipMemcpyHostToDevice, q));
    std::vector<std::pair<uint32_t, uint32_t>> levels; // first build node, count
    uint32_t n_tasks = 1, first = 0, n_build = 1;
    BvhItems *src = &A, *dst = &B;
    for (int lvl = 0; n_tasks > 0; lvl++) {
        if (lvl > 4096) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: tree deeper than 4096 levels");
        levels.push_back({first, n_tasks});
        hipLaunchKernelGGL(rt_bvh_level_kernel, dim3(n_tasks), dim3(RT_BVH_BLOCK), 0, q, tasks[lvl & 1], n_tasks, *src, *dst, bnodes, tasks[(lvl + 1) & 1], counters,
                           reinterpret_cast<uint32_t *>(blob + o_pre), reinterpret_cast<uint32_t *>(blob + o_holes), reinterpret_cast<uint32_t *>(blob + o_backs));
        hipLaunchKernelGGL(rt_bvh_copy_leaves_kernel, dim3(n_tasks), dim3(64), 0, q, tasks[lvl & 1], n_tasks, bnodes, *src, *dst);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(cnt, counters, sizeof cnt, hipMemcpyDeviceToHost, q));
        HIP_TRY(ctx, hipStreamSynchronize(q));
        if (cnt[2]) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_bvh: a split left one side empty (the reference's median fallback, src/bvh.rs:317-326, is not built on the device): use rsrt_build_bvh");
        first = n_build;
        n_tasks = cnt[1];
        n_build = cnt[0];
        cnt[1] = 0;
        HIP_TRY(ctx, hipMemcpyAsync(counters + 1, &cnt[1], 4, hipMemcpyHostToDevice, q));
        std::swap(src, dst);
    }
    // (after the swap `src` names the copy the last level wrote: every range is final in it)
    for (size_t d = levels.size(); d-- > 0;)
        hipLaunchKernelGGL(rt_bvh_sizes_kernel, dim3((levels[d].second + 255) / 256), dim3(256), 0, q, bnodes, levels[d].first, levels[d].second);
    for (size_t d = 0; d < levels.size(); d++)
        hipLaunchKernelGGL(rt_bvh_positions_kernel, dim3((levels[d].second + 255) / 256), dim3(256), 0, q, bnodes, levels[d].first, levels[d].second);
    hipLaunchKernelGGL(rt_bvh_emit_kernel, dim3((n_build + 255) / 256), dim3(256), 0, q, bnodes, n_build, reinterpret_cast<rsrt_bvh_node *>(blob + o_outn));
    hipLaunchKernelGGL(rt_bvh_prims_kernel, dim3((n + 255) / 256), dim3(256), 0, q, *src, n, reinterpret_cast<rsrt_primitive_info *>(blob + o_outp));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(e1, q));
    HIP_TRY(ctx, hipMemcpyAsync(nodes_out, blob + o_outn, (size_t)n_build * sizeof(rsrt_bvh_node), hipMemcpyDeviceToHost, q));
    HIP_TRY(ctx, hipMemcpyAsync(primitives_out, blob + o_outp, (size_t)n * sizeof(rsrt_primitive_info), hipMemcpyDeviceToHost, q));
    HIP_TRY(ctx, hipStreamSynchronize(q));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (n_nodes_out) *n_nodes_out = n_build;
    if (depth_out) *depth_out = (uint32_t)levels.size() - 1u;
    if (build_ms_out) *build_ms_out = ms;
    return RSRT_OK;
}

// The wide walk's tree for a BVH, on the host (no GPU needed): what rsrt_upload_scene builds for the device, exposed so that a CPU
// test can walk it.  wnodes_out: 32 floats per wide node (DevScene::wnodes); *n_wnodes: capacity in, count out; old_of_new: for
// every record index the walk uses, the index into `primitives` as given.  RSRT_ERR_INVALID_ARGUMENT: the scene does not qualify.
rsrt_status rsrt_wide_tree_build(const rsrt_primitive_info *primitives, uint32_t n_primitives, const rsrt_bvh_node *nodes, uint32_t n_nodes,
                                 float *wnodes_out, uint32_t *n_wnodes, uint32_t *old_of_new)
{
    if (!primitives || !nodes || !n_wnodes || n_nodes == 0) return RSRT_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < n_nodes; i++) { // (the checks of rsrt_upload_scene that keep the builder's indexing in bounds)
        const rsrt_bvh_node &nd = nodes[i];
        if (nd.primitives_len > 0) { if ((uint64_t)nd.primitives_or_second_child_index + nd.primitives_len > n_primitives) return RSRT_ERR_INVALID_ARGUMENT; }
        else if (i + 1 >= n_nodes || nd.primitives_or_second_child_index <= i + 1 || nd.primitives_or_second_child_index >= n_nodes) return RSRT_ERR_INVALID_ARGUMENT;
    }
    std::vector<WideNode> wide;
    std::vector<rsrt_primitive_info> pp;
    std::vector<rsrt_bvh_node> np;
    std::vector<uint32_t> oon;
    if (!build_wide_tree(nodes, n_nodes, primitives, n_primitives, wide, pp, np, &oon)) return RSRT_ERR_INVALID_ARGUMENT;
    const uint32_t cap = *n_wnodes;
    *n_wnodes = (uint32_t)wide.size();
    if (wnodes_out) {
        if (cap < wide.size()) return RSRT_ERR_INVALID_ARGUMENT;
        fill_wide_nodes(wide, np.data(), pp.data(), reinterpret_cast<float4 *>(wnodes_out));
    }
    if (old_of_new) memcpy(old_of_new, oon.data(), oon.size() * sizeof(uint32_t));
    return RSRT_OK;
}

rsrt_status rsrt_upload_environment(rsrt_context *ctx, uint32_t slot, uint32_t width, uint32_t height, const float *rgba,
                                    const rsrt_alias_entry *alias)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!rgba || width == 0 || height == 0) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "environment: NULL data or zero size");
    if ((uint64_t)width * height > 0x7fffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "environment too large");
    if (slot > 63) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "environment slot %u > 63", slot);
    const size_t n = (size_t)width * height;
    if (alias)
        for (size_t i = 0; i < n; i++)
            if (alias[i].alias_index >= n) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "alias entry %zu: alias_index %u out of range", i, alias[i].alias_index);
    if (ctx->envs.size() <= slot) ctx->envs.resize(slot + 1);
    Env &e = ctx->envs[slot];
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    if (e.rgba) { (void)hipFree(e.rgba); e.rgba = nullptr; }
    if (e.alias) { (void)hipFree(e.alias); e.alias = nullptr; }
    e.width = e.height = 0;
    HIP_TRY(ctx, hipMalloc(&e.rgba, n * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc(&e.alias, n * sizeof(uint4)));
    HIP_TRY(ctx, hipMemcpy(e.rgba, rgba, n * sizeof(float4), hipMemcpyHostToDevice));
    if (alias) {
        HIP_TRY(ctx, hipMemcpy(e.alias, alias, n * sizeof(uint4), hipMemcpyHostToDevice));
        e.width = width;
        e.height = height;
        return pack_environment(ctx, e);
    }
    e.width = width;
    e.height = height;
    const rsrt_status st = rsrt_environment_build_alias(ctx, slot, nullptr, 0, nullptr); // AliasTable::build_by_luminance on the device
    if (st) { e.width = e.height = 0; }
    return st;
}

// AliasTable::build_by_luminance (environments.rs:96-187) for the texels ALREADY in `slot`, on the device
// (rt_alias_device.h): same bits as the host builder rsrt_alias_table_build.  Replaces the slot's alias table.
rsrt_status rsrt_environment_build_alias(rsrt_context *ctx, uint32_t slot, rsrt_alias_entry *host_out, size_t n_entries, uint32_t *leftover_out)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (slot >= ctx->envs.size() || !ctx->envs[slot].rgba || !ctx->envs[slot].alias || ctx->envs[slot].width == 0)
        return fail(ctx, RSRT_ERR_NOT_READY, "environment %u not uploaded", slot);
    Env &e = ctx->envs[slot];
    const size_t n = (size_t)e.width * e.height;
    if (host_out && n_entries != n) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "build_alias: expected %zu entries", n);
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    // scratch: p[n] | small[n] | large[n] | block counts[nb] | {sum, n_small, leftover}
    const size_t bytes = n * 4 * 3 + (size_t)nb * 4 + 16;
    rsrt_status st = ensure_scratch(ctx, bytes);
    if (st || (st = begin_work(ctx, ctx->stream))) return st;
    float *p = static_cast<float *>(ctx->scratch);
    uint32_t *small = reinterpret_cast<uint32_t *>(p + n), *large = small + n, *blocks = large + n, *scalars = blocks + nb;
    hipStream_t q = ctx->stream;
    hipLaunchKernelGGL(rt_alias_weights_kernel, dim3(nb), dim3(256), 0, q, e.rgba, e.width, e.height, p);
    hipLaunchKernelGGL(rt_alias_sum_kernel, dim3(1), dim3(64), 0, q, p, n, reinterpret_cast<float *>(scalars));
    hipLaunchKernelGGL(rt_alias_normalise_kernel, dim3(nb), dim3(256), 0, q, p, n, reinterpret_cast<const float *>(scalars), e.alias, blocks);
    hipLaunchKernelGGL(rt_alias_scan_kernel, dim3(1), dim3(1024), 0, q, blocks, nb, scalars + 1);
    hipLaunchKernelGGL(rt_alias_scatter_kernel, dim3(nb), dim3(256), 0, q, p, n, blocks, small, large);
    hipLaunchKernelGGL(rt_alias_vose_kernel, dim3(1), dim3(64), 0, q, p, n, small, scalars + 1, large, e.alias, scalars + 2);
    hipLaunchKernelGGL(rt_alias_pmf_kernel, dim3(nb), dim3(256), 0, q, p, n, e.alias);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t left = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&left, scalars + 2, 4, hipMemcpyDeviceToHost, q));
    if (host_out) HIP_TRY(ctx, hipMemcpyAsync(host_out, e.alias, n * sizeof(uint4), hipMemcpyDeviceToHost, q)); // (before the pad words are packed)
    if ((st = end_work(ctx, q))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(q));
    if (leftover_out) *leftover_out = left;
    return pack_environment(ctx, e);
}

rsrt_status rsrt_set_partition(rsrt_context *ctx, uint32_t rank, uint32_t world_size, uint32_t tile_w, uint32_t tile_h)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    if (world_size == 0 || world_size > 65535u || rank >= world_size) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "rank %u not in [0,%u) (or more than 65535 ranks)", rank, world_size);
    if (tile_w == 0 || tile_h == 0 || tile_w * tile_h > 4096 || (tile_w * tile_h) % RT_WAVE != 0)
        return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "tile %ux%u: pixel count must be a multiple of 64 and at most 4096", tile_w, tile_h);
    ctx->rank = rank; ctx->world = world_size; ctx->tile_w = tile_w; ctx->tile_h = tile_h;
    return RSRT_OK;
}

rsrt_status rsrt_accumulator_resize(rsrt_context *ctx, uint32_t width, uint32_t height)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bad resolution %ux%u", width, height);
    if (ctx->accum && ctx->accum != ctx->accum_owned) { ctx->accum = nullptr; ctx->acc_w = ctx->acc_h = 0; }
    return ensure_accumulator(ctx, width, height);
}

rsrt_status rsrt_accumulator_bind(rsrt_context *ctx, void *device_rgba32f, uint32_t width, uint32_t height)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    if (!device_rgba32f) {
        ctx->accum = ctx->accum_owned;
        if (!ctx->accum) ctx->acc_w = ctx->acc_h = 0;
        return RSRT_OK;
    }
    if (width == 0 || height == 0) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bad resolution %ux%u", width, height);
    if ((uintptr_t)device_rgba32f % 16) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "accumulator pointer must be 16-byte aligned");
    if (ctx->accum_owned) { (void)hipFree(ctx->accum_owned); ctx->accum_owned = nullptr; }
    ctx->accum = static_cast<float4 *>(device_rgba32f);
    ctx->acc_w = width;
    ctx->acc_h = height;
    return RSRT_OK;
}

rsrt_status rsrt_accumulator_clear(rsrt_context *ctx)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    rsrt_status st = begin_work(ctx, ctx->stream);
    if (st) return st;
    HIP_TRY(ctx, hipMemsetAsync(ctx->accum, 0, (size_t)ctx->acc_w * ctx->acc_h * sizeof(float4), ctx->stream));
    return end_work(ctx, ctx->stream);
}

rsrt_status rsrt_accumulator_download(rsrt_context *ctx, float *host_rgba, size_t n_floats)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    if (!host_rgba || n_floats != (size_t)ctx->acc_w * ctx->acc_h * 4) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "download: expected %zu floats", (size_t)ctx->acc_w * ctx->acc_h * 4);
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    HIP_TRY(ctx, hipMemcpy(host_rgba, ctx->accum, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return RSRT_OK;
}

rsrt_status rsrt_resolve_mean_f16(rsrt_context *ctx, uint32_t sample_total, uint16_t *host, size_t n_halfs)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    const size_t n = (size_t)ctx->acc_w * ctx->acc_h;
    if (!host || n_halfs != n * 4 || sample_total == 0) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "resolve_mean_f16: expected %zu halfs and sample_total > 0", n * 4);
    rsrt_status st = ensure_scratch(ctx, n * sizeof(ushort4));
    if (st || (st = begin_work(ctx, ctx->stream))) return st;
    ushort4 *tmp = static_cast<ushort4 *>(ctx->scratch);
    hipLaunchKernelGGL(rt_mean_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->accum, n, 0.0f, sample_total, tmp);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(host, tmp, n * sizeof(ushort4), hipMemcpyDeviceToHost, ctx->stream));
    if ((st = end_work(ctx, ctx->stream))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RSRT_OK;
}

rsrt_status rsrt_debug_view_f16(rsrt_context *ctx, uint32_t dev_index, uint32_t environment_index, uint32_t sample_count, uint16_t *host_inout, size_t n_halfs)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    if (dev_index != 2u && dev_index != 3u) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "debug_view: dev_index %u (2: draws of the alias table, 3: the HDRI)", dev_index);
    if (environment_index >= ctx->envs.size() || !ctx->envs[environment_index].rgba) return fail(ctx, RSRT_ERR_NOT_READY, "debug_view: environment slot %u is empty", environment_index);
    const Env &e = ctx->envs[environment_index];
    const size_t n = (size_t)ctx->acc_w * ctx->acc_h;
    if (!host_inout || n_halfs != n * 4) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "debug_view: expected %zu halfs", n * 4);
    rsrt_status st = ensure_scratch(ctx, n * sizeof(ushort4) + n * sizeof(uint32_t));
    if (st || (st = begin_work(ctx, ctx->stream))) return st;
    ushort4 *tex = static_cast<ushort4 *>(ctx->scratch);
    uint32_t *counts = reinterpret_cast<uint32_t *>(tex + n);
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dev_index == 3u) {
        hipLaunchKernelGGL(rt_dev_view_hdri_kernel, dim3(nb), dim3(256), 0, ctx->stream, e.rgba, e.width, e.height, ctx->acc_w, ctx->acc_h, tex);
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(tex, host_inout, n * sizeof(ushort4), hipMemcpyHostToDevice, ctx->stream)); // out_texture as the last frame left it
        HIP_TRY(ctx, hipMemsetAsync(counts, 0, n * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(rt_dev_view_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, e.alias, e.width, e.height, ctx->acc_w, ctx->acc_h, sample_count, counts);
        hipLaunchKernelGGL(rt_dev_view_apply_kernel, dim3(nb), dim3(256), 0, ctx->stream, counts, n, tex);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(host_inout, tex, n * sizeof(ushort4), hipMemcpyDeviceToHost, ctx->stream));
    if ((st = end_work(ctx, ctx->stream))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RSRT_OK;
}

rsrt_status rsrt_display_srgb8(rsrt_context *ctx, uint32_t sample_total, uint8_t *host_rgba8, size_t n_bytes)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    return display_from(ctx, ctx->accum, sample_total, host_rgba8, n_bytes);
}

rsrt_status rsrt_render(rsrt_context *ctx, const rsrt_camera *camera, uint32_t width, uint32_t height, uint32_t sample_begin,
                        uint32_t sample_count, uint32_t max_bounces, uint32_t environment_index, uint32_t flags, void *hip_stream)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!camera) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (!ctx->scene_ready) return fail(ctx, RSRT_ERR_NOT_READY, "no scene uploaded");
    if (environment_index >= ctx->envs.size() || !ctx->envs[environment_index].rgba) return fail(ctx, RSRT_ERR_NOT_READY, "environment %u not uploaded", environment_index);
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "bad resolution %ux%u", width, height);
    if ((uint64_t)sample_begin + sample_count > 0xffffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "sample range overflows u32");
    if (flags & ~(uint32_t)(RSRT_FLAG_REFERENCE_TRAVERSAL | RSRT_FLAG_PRUNE)) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    rsrt_status st = ensure_accumulator(ctx, width, height);
    if (st) return st;
    if (ctx->pending_events.size() + ctx->pending_reduce.size() >= 32) { st = collect_events(ctx); if (st) return st; } // bounds the event pool
    if (sample_count == 0) return RSRT_OK;

    RenderParams P;
    memset(&P, 0, sizeof P);
    P.scene = ctx->scene;
    const Env &env = ctx->envs[environment_index];
    P.env.rgba = env.rgba; P.env.alias = env.alias; P.env.width = env.width; P.env.height = env.height;
    P.env.wf = (float)env.width;
    P.env.hf = (float)env.height;
    P.env.dphi_dtheta = (RT_TWO_PI / (float)env.width) * (RT_PI / (float)env.height);
    P.env.width_shift = (env.width & (env.width - 1)) == 0 ? (uint32_t)__builtin_ctz(env.width) : 0xffffffffu;
    memcpy(P.cam_pos, camera->pos, 12);
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) P.cam_rot[3 * j + k] = camera->rot_transform[j][k];
    P.fov_y = camera->fov_y;
    P.width = width; P.height = height;
    P.max_bounces = max_bounces; P.flags = flags;
    P.tile_w = ctx->tile_w; P.tile_h = ctx->tile_h;
    P.tiles_x = (width + P.tile_w - 1) / P.tile_w;
    const uint32_t tiles_y = (height + P.tile_h - 1) / P.tile_h;
    P.n_tiles = P.tiles_x * tiles_y;
    P.rank = ctx->rank; P.world = ctx->world;
    P.skew = partition_skew(P.world);
    P.tiles_per_row = (P.tiles_x + P.world - 1) / P.world;
    P.n_owned_tiles = tiles_y * P.tiles_per_row; // tile slots, the same for every rank (some may be padding, see owned_tile)
    const uint32_t tile_px = P.tile_w * P.tile_h;
    if ((uint64_t)P.n_owned_tiles * tile_px > 0x7fffffffull) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "frame too large");
    P.n_slots = P.n_owned_tiles * tile_px;
    P.descend_quorum = ctx->descend_quorum;
    P.flat_quorum = ctx->flat_quorum;
    P.stop_quorum = ctx->stop_quorum;
    P.stats = ctx->dev_stats;
    if (P.n_slots == 0) return RSRT_OK;

    // sample buffer: [samples of this pass][slot][3]; passes bound its size
    const size_t budget = ctx->sample_buffer_budget;
    const size_t per_sample = (size_t)P.n_slots * 3 * sizeof(float);
    uint32_t pass_samples = (uint32_t)std::max<size_t>(1, std::min<size_t>(sample_count, budget / per_sample));
    pass_samples = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(pass_samples, 0xffffffffull / P.n_slots)); // slot ids are 32-bit
    const size_t need = per_sample * pass_samples;

    const int kv = ctx->kernel_variant;
    const int trav = select_traversal(ctx, P.scene, max_bounces, flags);
    int sv = P.scene.lds_float4s != 0 ? 1 : 0;
    if (sv == 0 && kv != 0 && ctx->allow_hybrid) { // mid-size scene: what the chosen traversal's box steps touch, in LDS (the first kernel has no hybrid form)
        if (hybrid_stage(ctx, P.scene, trav, trav == 6 ? kCoopRoomF4 : kHybridRoomF4)) sv = 2;
    }
    // measured on suzanne and the 15 k-triangle grid (profiles/r02_bvh_knobs.txt): with the quorum vote a round is short, and
    // handing the slot back to the scheduler after about one round beats running several rounds with thinning lanes
    P.trace_budget = ctx->trace_budget ? ctx->trace_budget : (trav >= 4 ? 4u : (trav == 3 ? 6u : 12u)); // (wide walk: rounds, not steps)
    P.coop_lds_cap = ctx->coop_lds_cap;
    P.coop_lifo_at = ctx->coop_lifo_at;
    P.coop_narrow_at = ctx->coop_narrow_at;
    // a small job behind a kernel that is still running: the 256-thread form, one workgroup per CU, on one of four lanes (see Lane)
    bool pipelined = false;
    if (ctx->overlap && kv == 4 && sv == 1 && trav != 6 && (uint64_t)P.n_slots * sample_count <= ctx->small_paths && sample_count <= pass_samples)
        for (auto &L : ctx->lanes) pipelined = pipelined || (L.traced_valid && hipEventQuery(L.traced) == hipErrorNotReady);
    const int kv_eff = pipelined ? 2 : kv;
    const bool big = kv_eff == 4 && sv == 1 && trav != 6; // one workgroup per CU shares the scene copy
    const uint32_t pool = trav == 6 ? RT_COOP_POOL : (sv == 2 ? RT_WALK_POOL : ((kv_eff == 4 && !big) ? 160u : kVariantPool[kv_eff]));
    const uint32_t block = (sv == 2 && trav == 6) ? (uint32_t)RT_COOP_BLOCK : ((sv == 2 || big) ? 1024u : (uint32_t)RT_BLOCK);
    const size_t scene_bytes = (size_t)P.scene.lds_float4s * sizeof(float4);
    const size_t smem = kv == 0 ? scene_bytes + (size_t)P.scene.stack_entries * RT_BLOCK * sizeof(uint32_t)
                                : scene_bytes + (size_t)(block / RT_WAVE) * 4u * pool_wave_lds_dwords(trav, pool);
    if (smem > 160 * 1024) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "kernel needs %zu bytes of LDS (> 160 KiB): bvh too deep for this pool size", smem);
    const void *kfn = variant_function(kv_eff, sv, trav);
    int pipe_blocks = ctx->pipe_blocks; // workgroups per CU of a pipelined small job: four such jobs fill a CU
    if (pipelined && !ctx->hw_queue_warned && ctx->hw_queues < (int)ctx->n_lanes + 1) { // (the lanes' streams and the caller's)
        ctx->hw_queue_warned = true;
        fprintf(stderr, "librsrt: pipelining single-sample calls over %u streams on %d hardware queues: set GPU_MAX_HW_QUEUES=8 before the first HIP call (INTEGRATION.md)\n",
                ctx->n_lanes, ctx->hw_queues);
    }
    int &bpc = pipelined ? pipe_blocks : ctx->blocks_per_cu[sv * 7 + trav][kv];
    if (bpc == 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, (int)block, smem);
        if (e != hipSuccess || nb <= 0) nb = 1;
        bpc = std::min(nb, 8);
        if (ctx->max_blocks_per_cu > 0) bpc = std::min(bpc, ctx->max_blocks_per_cu);
    }

    const size_t need_cold = kv != 0 ? (size_t)ctx->cus * bpc * (block / RT_WAVE) * pool_wave_cold_dwords(trav, pool) * sizeof(uint32_t) : 0; // cold path-state arena: one block of columns per wave that can be resident
    // the context's buffers are shared by every call: order this stream after whatever ran last (another stream's
    // render, rsrt_accumulator_clear on the context's own stream, ...) — with lanes, enqueue_pass orders each resolve itself
    if (!ctx->overlap && (st = begin_work(ctx, stream))) return st;
    rsrt_context::Lane *last_lane = nullptr;
    for (uint32_t done = 0; done < sample_count; done += pass_samples) {
        P.sample_begin = sample_begin + done;
        P.sample_count = std::min(pass_samples, sample_count - done);
        uint32_t li = 0; // ordinary jobs: lane 0, one after the other (a frame's kernel time is then its own, not its neighbour's too)
        if (ctx->overlap && pipelined) { li = ctx->next_small_lane; ctx->next_small_lane = (li + 1u) % ctx->n_lanes; }
        else ctx->next_small_lane = 1u;
        rsrt_context::Lane &lane = ctx->lanes[li];
        if (need > lane.sample_buf_bytes || need_cold > lane.cold_bytes) { // (grow-only; a reallocation waits for everything in flight)
            if ((st = sync_all(ctx))) { if (!ctx->overlap) (void)end_work(ctx, stream); return st; }
            if (need > lane.sample_buf_bytes) {
                if (lane.sample_buf) { (void)hipFree(lane.sample_buf); lane.sample_buf = nullptr; lane.sample_buf_bytes = 0; }
                HIP_TRY(ctx, hipMalloc(&lane.sample_buf, need));
                lane.sample_buf_bytes = need;
            }
            if (need_cold > lane.cold_bytes) {
                if (lane.cold_state) { (void)hipFree(lane.cold_state); lane.cold_state = nullptr; lane.cold_bytes = 0; }
                HIP_TRY(ctx, hipMalloc(&lane.cold_state, need_cold));
                lane.cold_bytes = need_cold;
            }
        }
        P.sample_buf = lane.sample_buf;
        P.cold_state = lane.cold_state;
        P.work_counter = lane.work_counter;
        rsrt_context::PassEvents pe = {get_event(ctx), get_event(ctx), get_event(ctx)};
        const rsrt_status pst = enqueue_pass(ctx, P, pe, kfn, block, bpc, smem, per_sample, max_bounces, stream, lane, pipelined ? ctx->pipe_div : 1u);
        if (pst != RSRT_OK) { // nothing of this pass is pending: the three events go back to the pool
            ctx->event_pool.push_back(pe.begin); ctx->event_pool.push_back(pe.traced); ctx->event_pool.push_back(pe.end);
            // earlier passes may be in flight.  With lanes the chain's end stays where the last successful pass recorded it (its lane's
            // resolve): recording it on the caller's stream here would drop the dependency on those resolves
            if (!ctx->overlap) (void)end_work(ctx, stream);
            return pst;
        }
        ctx->pending_events.push_back(pe);
        last_lane = &lane;
    }
    if (!ctx->overlap) return end_work(ctx, stream);
    // a caller that named a stream of its own gets that stream's semantics: what it enqueues there next comes after this render
    if (hip_stream && last_lane) HIP_TRY(ctx, hipStreamWaitEvent(stream, last_lane->resolved, 0));
    return RSRT_OK;
}

rsrt_status rsrt_synchronize(rsrt_context *ctx)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    return sync_all(ctx);
}

rsrt_status rsrt_get_stats(rsrt_context *ctx, rsrt_stats *out)
{
    if (!ctx || !out) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    rsrt_status st = collect_stats(ctx);
    if (st) return st;
    *out = ctx->stats;
    return RSRT_OK;
}

// Exhaustive check of rt_rcp against the compiler's correctly rounded division, all 2^32 bit patterns.
__global__ void rt_selftest_rcp_kernel(unsigned long long *out)
{
    unsigned long long bad = 0, raw_differs = 0, fast_path = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = as_f((uint32_t)i);
        const float ref = 1.0f / x;
        if (as_u(rt_rcp(x)) != as_u(ref)) {
            bad++;
            atomicMin(&out[3], (unsigned long long)i);
        }
        const uint32_t ex = ((uint32_t)i >> 23) & 0xffu;
        if (ex - 2u < 251u) {
            fast_path++;
            if (as_u(__builtin_amdgcn_rcpf(x)) != as_u(ref)) raw_differs++; // sanity: the Newton step is doing something
        }
    }
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], fast_path);
    atomicAdd(&out[2], raw_differs);
}

rsrt_status rsrt_selftest_numerics(rsrt_context *ctx, uint64_t out[4])
{
    if (!ctx || !out) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    unsigned long long *d = nullptr, h[4] = {0, 0, 0, ~0ull};
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    HIP_TRY(ctx, hipMalloc(&d, sizeof h));
    hipError_t e = hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_selftest_rcp_kernel, dim3(ctx->cus * 8), dim3(256), 0, ctx->stream, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, RSRT_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    for (int i = 0; i < 4; i++) out[i] = h[i];
    return RSRT_OK;
}

rsrt_status rsrt_get_debug_counters(rsrt_context *ctx, uint64_t out[32])
{
    if (!ctx || !out) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    rsrt_status st = collect_stats(ctx);
    if (st) return st;
    for (int i = 0; i < 32; i++) out[i] = ctx->debug_words[i];
    return RSRT_OK;
}

rsrt_status rsrt_get_region_counters(rsrt_context *ctx, uint64_t out[32])
{
    if (!ctx || !out) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    for (int i = 0; i < 32; i++) out[i] = 0;
#ifdef RT_INSTRUMENT
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    unsigned long long h[32], z[32] = {0};
    HIP_TRY(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(rt_region_lanes), sizeof h));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(rt_region_lanes), z, sizeof z));
    for (int i = 0; i < 32; i++) out[i] = h[i];
#endif
    return RSRT_OK;
}

rsrt_status rsrt_cast_rays(rsrt_context *ctx, uint32_t n, const float *origins, const float *dirs, uint32_t mode, uint32_t flags,
                           rsrt_hit *out)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->scene_ready) return fail(ctx, RSRT_ERR_NOT_READY, "no scene uploaded");
    if (n == 0) return RSRT_OK;
    if (!origins || !dirs || !out || mode > 31 || ((mode >> 1) & 7u) > 6u) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: bad arguments");
    // mode: bit 0 = cast_ray_bvh only; bits 1-3 = traversal (0 threaded, 1 stack, 2 typed leaf loops, 3 flat, 4 fixed-order, 5 wide, 6 cooperative wide);
    // bit 4 = scene read from LDS exactly as the production kernel stages it for that traversal (whole image, or for a
    // mid-size scene the nodes + escape links / the pre-order nodes)
    DevScene sc = ctx->scene;
    const uint32_t sel = (mode >> 1) & 7u;
    const int trav = sel == 0 ? 0 : (sel == 1 ? 4 : (sel == 2 ? 1 : (sel == 3 ? 2 : (sel == 4 ? 3 : (sel == 5 ? 5 : 6))))); // (probe numbering: 4 = stack walk, 5 = wide walk, 6 = cooperative wide walk)
    const size_t coop_lds = trav == 6 ? (size_t)(RT_BLOCK / RT_WAVE) * 4u * pool_wave_lds_dwords(6, 64u) : 0u; // the probe's four one-column pools with their stacks
    int sv = 0;
    if (mode & 16u) {
        if (sc.lds_float4s != 0) {
            sv = 1;
        } else {
            // (what the production kernel stages for that traversal — the probe has no pools beside it; only the stack walk needs a stack)
            const uint32_t all_f4 = 160u * 1024u / (uint32_t)sizeof(float4);
            const uint32_t stack_f4 = trav == 4 ? (uint32_t)(((size_t)sc.stack_entries * RT_BLOCK * sizeof(uint32_t) + 15u) / 16u) : 0u;
            const uint32_t coop_f4 = (uint32_t)((coop_lds + 15u) / 16u);
            if (!hybrid_stage(ctx, sc, trav == 5 ? 4 : (trav == 4 ? 0 : trav), all_f4 > stack_f4 + coop_f4 ? all_f4 - stack_f4 - coop_f4 : 0u))
                return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: this scene is not staged in LDS by the production kernel");
            sv = 2;
        }
    } else {
        sc.lds_float4s = 0;
    }
    if (trav == 6 && !(sc.wide_ok && sc.coop_ok)) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: the cooperative walk needs what the wide walk needs, fewer than 2^21 records and 2^24 wide nodes");
    if (trav == 5 && !sc.wide_ok) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: the wide walk needs nested boxes, leaves of at most 8 primitives that share no record, and a wide tree of at most 25 levels");
    if ((trav == 1 || trav == 3) && !sc.typed_leaves) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: typed leaf loops need leaves of at most 8 primitives");
    if (trav == 2 && (!sc.flat_ok || sv == 2)) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: the flat traversal needs a scene of at most 64 records with nested boxes");
    { rsrt_status st0 = sync_all(ctx); if (st0) return st0; }
    float *d_o = nullptr, *d_d = nullptr;
    rsrt_hit *d_h = nullptr;
    uint32_t *d_g = nullptr; // (cooperative walk: the waves' overflow blocks)
    const uint32_t n_blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
    hipError_t e = hipMalloc(&d_o, (size_t)n * 12);
    if (e == hipSuccess && trav == 6) e = hipMalloc(&d_g, (size_t)n_blocks * (RT_BLOCK / RT_WAVE) * RT_COOP_GCAP * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_d, (size_t)n * 12);
    if (e == hipSuccess) e = hipMalloc(&d_h, (size_t)n * sizeof(rsrt_hit));
    if (e == hipSuccess) e = hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, dirs, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t smem = (size_t)sc.lds_float4s * sizeof(float4) + (trav == 4 ? (size_t)sc.stack_entries * RT_BLOCK * sizeof(uint32_t) : 0u) + coop_lds; // (only the stack walk has a stack)
        if (smem > 160 * 1024) { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_h); (void)hipFree(d_g); return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "cast_rays: needs %zu bytes of LDS", smem); }
        uint32_t repeat = ctx->probe_repeat;
        void *kargs[] = {&sc, &n, &d_o, &d_d, &mode, &flags, &repeat, &d_h, &d_g, &ctx->coop_lds_cap, &ctx->coop_lifo_at, &ctx->coop_narrow_at}; // (the last three: the cooperative walk's probe only)
        e = hipLaunchKernel(probe_function(sv, trav), dim3(n_blocks), dim3(RT_BLOCK), kargs, smem, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_h, (size_t)n * sizeof(rsrt_hit), hipMemcpyDeviceToHost);
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_h); (void)hipFree(d_g);
    if (e != hipSuccess) return fail(ctx, RSRT_ERR_HIP, "cast_rays: %s", hipGetErrorString(e));
    return RSRT_OK;
}

} // extern "C"

#include "rsrt_comm.h"
