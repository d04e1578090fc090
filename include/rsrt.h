/*
 * rsrt.h — C-ABI of the MI355X path-tracing integrator (librsrt.so).
 *
 * Drop-in boundary.  The reference has no FFI: its integrator is the WGSL compute shader
 * src/shaders/shader.wgsl (entry `main` :1305, `trace_ray` :1213) reached through three wgpu
 * bind groups that `State::new` builds (src/state.rs:60-649) and `State::render` dispatches
 * once per displayed frame (src/state.rs:760-833).  The entry points below are that bind-group
 * contract restated as plain C calls — what a Rust `extern "C"` block in src/state.rs would
 * bind (INTEGRATION.md shows the stub):
 *
 *   bind group 2 (scene storage buffers, state.rs:394-458)        -> rsrt_upload_scene
 *   bind group 0 bindings 2-5 (sampler, HDRIs, metadata, alias)   -> rsrt_upload_environment
 *   bind group 1 (camera/resolution/sample_count/env index)       -> arguments of rsrt_render
 *   bind group 0 binding 1 (cumulative_light_texture, RGBA32F sum) -> the accumulator
 *   bind group 0 binding 0 (out_texture, RGBA16F mean)             -> rsrt_resolve_mean_f16
 *   compute_pass.dispatch_workgroups (state.rs:808-824)            -> rsrt_render
 *   encoder.clear_texture(cumulative) on scene-hash change (:778-786) -> rsrt_accumulator_clear
 *
 * Semantics that differ from the reference on purpose:
 *   - one rsrt_render call may add MANY samples (the reference adds exactly one per frame);
 *     sample k of pixel p always uses the reference's RNG seed (pixel_index, k)
 *     (shader.wgsl:1309-1312), so any split over calls / tiles / GPUs gives the same image;
 *   - max_bounces is a run-time argument (the reference's constant is 10, shader.wgsl:232);
 *   - dev_index 1 (normal render) is rsrt_render; the developer views 2 / 3 (shader.wgsl:1314-1338) are rsrt_debug_view_f16.
 *
 * Threading: a context is used from one thread at a time, like `State`.  All functions return
 * an rsrt_status (0 = ok) and never throw/abort across the boundary; rsrt_last_error() gives
 * the message.  No CPU fallback exists: without a gfx950 device rsrt_context_create fails.
 */
#ifndef RSRT_H
#define RSRT_H

#include <stddef.h>
#include <stdint.h>

#include "rsrt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rsrt_status {
    RSRT_OK = 0,
    RSRT_ERR_INVALID_ARGUMENT = 1,
    RSRT_ERR_NO_DEVICE = 2,
    RSRT_ERR_HIP = 3,
    RSRT_ERR_NOT_READY = 4, /* scene or environment not uploaded / accumulator missing */
    RSRT_ERR_OUT_OF_MEMORY = 5,
    RSRT_ERR_COMM = 6 /* RCCL missing or a collective failed (multi-GPU only) */
} rsrt_status;

typedef struct rsrt_context rsrt_context;

/* flags of rsrt_render */
enum {
    /* Default (0): exactly the primitives the reference's unpruned walk tests are tested
     * (shader.wgsl:469-564), and of equal distances the one it visits first wins; only the NEE
     * shadow query, of which the shader reads nothing but `.did_hit` (:1249), stops at its
     * first hit.  That is exactly result-preserving.
     *
     * REFERENCE_TRAVERSAL: also run the shadow query to the end, as the shader literally does. */
    RSRT_FLAG_REFERENCE_TRAVERSAL = 1u,
    /* PRUNE: skip nodes whose slab entry lies beyond the current best hit.  NOT exactly
     * result-preserving: the slab entry t and a primitive's own t are rounded differently, and a
     * primitive a hair closer than the current best can sit in a skipped node.  Measured on
     * house.toml 1920x1080 x 256 spp: 2 of 5.3e8 paths change (per-channel RMSE 4e-7).  Opt-in;
     * honoured by the tree-walk traversals only (scenes of up to 64 primitives are not walked). */
    RSRT_FLAG_PRUNE = 2u
};

/* Counters of the work submitted since the previous rsrt_get_stats call (and cumulative since
 * context creation).  Times are HIP-event times on the stream the kernels were launched on. */
typedef struct rsrt_stats {
    uint64_t paths;        /* camera paths started */
    uint64_t ext_rays;     /* calls of cast_ray (shader.wgsl:1221) */
    uint64_t shadow_rays;  /* NEE queries actually issued (shader.wgsl:1246-1250) */
    double kernel_ms;      /* HIP-event time of the integrator kernel(s) on the launch stream */
    uint64_t total_paths, total_ext_rays, total_shadow_rays;
    double total_kernel_ms;
    uint32_t launches;     /* kernel launches since the previous rsrt_get_stats */
    uint32_t _pad;
    double trace_kernel_ms;   /* part of kernel_ms spent in the path-tracing kernel (rt_render_kernel) */
    double resolve_kernel_ms; /* part spent in the ordered sample resolve (rt_resolve_kernel) */
    double reduce_ms;         /* HIP-event time of rsrt_comm_reduce calls (the RCCL reduce of the accumulators) */
    uint64_t traversal_steps; /* box tests + primitive tests of the BVH walks (0 where the flat small-scene loop runs:
                                 its work per ray is fixed by the scene); what RSRT_FLAG_PRUNE reduces */
} rsrt_stats;

/* -- context: State::new's device acquisition (state.rs:60-98) ------------------------------ */
rsrt_status rsrt_context_create(int device_index, rsrt_context **out);
void rsrt_context_destroy(rsrt_context *ctx);
/* Message of the last failing call on ctx; with ctx == NULL, of the last failing
 * rsrt_context_create on this thread. Never NULL. */
const char *rsrt_last_error(const rsrt_context *ctx);

/* -- scene: bind group 2, the eight storage buffers (state.rs:394-458) -----------------------
 * Arrays are in the layouts of rsrt_types.h; the library copies and re-lays them out for the
 * device, the caller keeps ownership.  Any array may be empty (pointer ignored when count 0)
 * except bvh_nodes.  Indices are validated; an out-of-range index is RSRT_ERR_INVALID_ARGUMENT. */
rsrt_status rsrt_upload_scene(rsrt_context *ctx,
                              const rsrt_material *materials, uint32_t n_materials,
                              const rsrt_sphere *spheres, uint32_t n_spheres,
                              const rsrt_plane *planes, uint32_t n_planes,
                              const rsrt_vec3 *vertices, uint32_t n_vertices,
                              const rsrt_vec3 *normals, uint32_t n_normals,
                              const rsrt_triangle *triangles, uint32_t n_triangles,
                              const rsrt_primitive_info *primitives, uint32_t n_primitives,
                              const rsrt_bvh_node *bvh_nodes, uint32_t n_bvh_nodes);

/* -- environment: bind group 0 bindings 3-5 (state.rs:119-132, environments.rs:19-64) --------
 * slot = index into the reference's binding_array / `environments` array.  rgba: width*height*4
 * f32, rows top to bottom, alpha ignored (texture.rs:112-115 writes 0).  alias: width*height
 * entries as AliasTable::build_by_luminance produces (rsrt_host.h has that builder), or NULL to have the
 * library build the table on the device (rsrt_environment_build_alias below). */
rsrt_status rsrt_upload_environment(rsrt_context *ctx, uint32_t slot, uint32_t width, uint32_t height,
                                    const float *rgba, const rsrt_alias_entry *alias);
/* AliasTable::build_by_luminance (src/environments.rs:96-187) ON THE DEVICE, for the texels already uploaded in `slot`
 * (SURVEY.md §8 f3): the same bits as the host builder rsrt_alias_table_build (rsrt_host.h) — the sequential f32 sum and
 * the LIFO Vose pairing are kept sequential, one wave runs them (csrc/hip/rt_alias_device.h).  Replaces the slot's
 * table; host_out (may be NULL) receives the width*height entries, leftover_out (may be NULL) the count of entries
 * that kept the default {1, self, 1/N}.  rsrt_upload_environment with alias == NULL uploads the texels and calls this. */
rsrt_status rsrt_environment_build_alias(rsrt_context *ctx, uint32_t slot, rsrt_alias_entry *host_out, size_t n_entries,
                                         uint32_t *leftover_out);

/* build_bvh (src/bvh.rs:13-337) ON THE DEVICE (csrc/hip/rt_bvh_device.h): the arguments and outputs of the host
 * builder rsrt_build_bvh (rsrt_host.h) — host arrays in, host arrays out: primitives_out holds n_spheres + n_planes + n_triangles
 * entries, nodes_out room for twice that — and the same bits in both: level-parallel binned SAH with integer bucket counts,
 * min / max bounds, the host's f32 cost expression, and the reference's unstable two-pointer partition reproduced as its
 * closed-form permutation.  build_ms_out (may be NULL): device time of the build, uploads and downloads excluded.  A split that
 * leaves one side empty (the reference's median fallback, unreachable for finite input) is RSRT_ERR_INVALID_ARGUMENT: use the
 * host builder then. */
rsrt_status rsrt_build_bvh_device(rsrt_context *ctx, const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                                  const rsrt_vec3 *vertices, uint32_t n_vertices, const rsrt_triangle *triangles, uint32_t n_triangles,
                                  rsrt_primitive_info *primitives_out, rsrt_bvh_node *nodes_out, uint32_t *n_nodes_out, uint32_t *depth_out,
                                  double *build_ms_out);

/* -- multi-GPU framebuffer ownership (no reference counterpart) ---------------
 * The frame is cut into tile_w x tile_h pixel tiles; this context renders tile (tx, ty) iff (tx + ty * skew) % world_size ==
 * rank — interleaved in x, each tile row shifted by `skew` (the smallest odd number >= 3 coprime to world_size: 3 for 2, 4, 8
 * GPUs) against the row above, so that a rank's tiles form a lattice whatever the frame width (t % world_size would give every
 * rank fixed column stripes whenever the tiles per row are a multiple of world_size) — and leaves every other pixel of the
 * accumulator untouched.  Default: rank 0 of 1 (whole frame). */
rsrt_status rsrt_set_partition(rsrt_context *ctx, uint32_t rank, uint32_t world_size, uint32_t tile_w, uint32_t tile_h);

/* Pure host arithmetic of that partition (no GPU needed): the rank that renders pixel (x, y) (UINT32_MAX for bad
 * arguments), and a width*height byte mask (1 = rendered by `rank`; mask may be NULL) with the pixel count. */
uint32_t rsrt_partition_owner(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world_size, uint32_t x, uint32_t y);
rsrt_status rsrt_partition_mask(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world_size,
                                uint8_t *mask, uint64_t *owned_pixels);
/* The compact tile buffer the exchange step moves: every rank owns *n_tile_slots = tiles_y * ceil(tiles_x / world_size) tile
 * slots — the same number for all ranks — of tile_w * tile_h RGBA32F pixels each, row-major inside the tile.  tiles_xy (may be
 * NULL): 2 words per slot, the tile's (tx, ty), or UINT32_MAX twice for a padding slot (a slot beyond the right edge of the
 * frame; it holds zeros). */
rsrt_status rsrt_partition_tiles(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world_size,
                                 uint32_t *tiles_xy, uint32_t *n_tile_slots);

/* -- multi-GPU, form 1: one process (or thread) per GPU ----------------------------------------
 * The one exchange step of the path — the sum of the ranks' W*H*4 accumulators onto a root, once per frame — runs on RCCL
 * over xGMI INSIDE the library (librccl is dlopen'ed on first use; a single-GPU caller never needs it).  Every pixel has
 * exactly one owner, so that sum is a GATHER: each rank packs its tiles into the compact buffer above (1 / world of the
 * frame), the root receives world - 1 of them point to point (grouped ncclSend / ncclRecv) and scatters them into the frame;
 * nothing is added, and the N-GPU frame equals the 1-GPU frame bit for bit.  (RSRT_COMM_MODE=reduce in the environment when the
 * context is created, or rsrt_comm_set_mode(ctx, 1): the dense ncclReduce(sum, f32) of the full accumulators instead — the
 * fallback until the gather has run on a multi-GPU box; bench.py switches to it by itself if the gathered frame is not the 1-GPU frame.)
 *   rank 0:      rsrt_comm_unique_id(&id); hand the 128 bytes to the other ranks (file, socket, MPI, a torch store ...)
 *   every rank:  rsrt_comm_init(ctx, rank, world, &id)   -- creates the communicator (collective call) and sets the tile
 *                                                            partition (rank, world, current tile size)
 *   per frame:   rsrt_accumulator_clear; rsrt_render ...; rsrt_comm_reduce(ctx, root, recv, stream)
 * rsrt_comm_reduce is asynchronous on `hip_stream` (NULL = the context's stream) and ordered after the context's
 * earlier work.  recv_device_rgba32f (root only, ignored elsewhere): where the full frame goes; NULL = in place, i.e.
 * the root's accumulator becomes the full frame (clear it before rendering further samples); a progressive caller
 * passes a separate W*H*4 f32 device buffer so that its accumulator keeps holding only its own tiles.
 * Without rsrt_comm_init the world is one rank and the reduce is a (device) copy or nothing.
 * N > 1 over RCCL is unverified on hardware so far. */
#define RSRT_UNIQUE_ID_BYTES 128
typedef struct rsrt_unique_id { char bytes[RSRT_UNIQUE_ID_BYTES]; } rsrt_unique_id;
int rsrt_comm_available(void); /* 1: librccl could be loaded (a dlopen, nothing else: safe to ask on every rank BEFORE the collective rsrt_comm_init) */
rsrt_status rsrt_comm_unique_id(rsrt_unique_id *out); /* error text: rsrt_last_error(NULL) */
rsrt_status rsrt_comm_init(rsrt_context *ctx, uint32_t rank, uint32_t world_size, const rsrt_unique_id *id);
rsrt_status rsrt_comm_reduce(rsrt_context *ctx, uint32_t root, void *recv_device_rgba32f, void *hip_stream);
rsrt_status rsrt_comm_set_mode(rsrt_context *ctx, uint32_t dense_reduce); /* 0 (default; RSRT_COMM_MODE unset): gather of compact tile buffers, 1 (RSRT_COMM_MODE=reduce):
                                                                             * dense ncclReduce(sum) of the full accumulators; every rank the same, between frames */
rsrt_status rsrt_comm_destroy(rsrt_context *ctx); /* also done by rsrt_context_destroy */

/* -- multi-GPU, form 2: one caller, a list of devices -----------------------
 * What the reference's single-threaded `State` (src/state.rs:60-98 device acquisition, :760-833 render) would bind:
 * one handle over 1/2/4/8 GPUs of the node.  Inside: one rsrt_context per device (rsrt_multi_context gives access,
 * e.g. for per-device stats), ncclCommInitAll for a list of two or more, device i renders the tiles of rank i, and the frame
 * is gathered onto devices[0] (one RCCL group per frame, into a frame buffer, so the per-device accumulators stay progressive) whenever
 * it is asked for.  Calls mirror the single-device ones; errors: rsrt_multi_last_error (NULL handle: of the last
 * failing rsrt_multi_create on this thread). */
typedef struct rsrt_multi rsrt_multi;
rsrt_status rsrt_multi_create(const int *devices, uint32_t n_devices, rsrt_multi **out);
void rsrt_multi_destroy(rsrt_multi *m);
const char *rsrt_multi_last_error(const rsrt_multi *m);
uint32_t rsrt_multi_size(const rsrt_multi *m);
rsrt_context *rsrt_multi_context(rsrt_multi *m, uint32_t i);
/* 1: the frame is gathered by RCCL; 0: by peer copies of the compact tile buffers onto devices[0] — a list of one device (no
 * exchange at all), librccl could not be loaded or its communicators did not come up, or the list names one device twice,
 * which is accepted only with RSRT_MULTI_ALLOW_SAME_DEVICE=1 (a rehearsal of N devices on one GPU). */
int rsrt_multi_uses_rccl(const rsrt_multi *m);
rsrt_status rsrt_multi_upload_scene(rsrt_multi *m,
                                    const rsrt_material *materials, uint32_t n_materials,
                                    const rsrt_sphere *spheres, uint32_t n_spheres,
                                    const rsrt_plane *planes, uint32_t n_planes,
                                    const rsrt_vec3 *vertices, uint32_t n_vertices,
                                    const rsrt_vec3 *normals, uint32_t n_normals,
                                    const rsrt_triangle *triangles, uint32_t n_triangles,
                                    const rsrt_primitive_info *primitives, uint32_t n_primitives,
                                    const rsrt_bvh_node *bvh_nodes, uint32_t n_bvh_nodes);
rsrt_status rsrt_multi_upload_environment(rsrt_multi *m, uint32_t slot, uint32_t width, uint32_t height,
                                          const float *rgba, const rsrt_alias_entry *alias);
rsrt_status rsrt_multi_resize(rsrt_multi *m, uint32_t width, uint32_t height);
rsrt_status rsrt_multi_clear(rsrt_multi *m);
rsrt_status rsrt_multi_render(rsrt_multi *m, const rsrt_camera *camera, uint32_t width, uint32_t height,
                              uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces,
                              uint32_t environment_index, uint32_t flags);
rsrt_status rsrt_multi_synchronize(rsrt_multi *m);
rsrt_status rsrt_multi_download(rsrt_multi *m, float *host_rgba, size_t n_floats);           /* reduce, then copy */
rsrt_status rsrt_multi_display_srgb8(rsrt_multi *m, uint32_t sample_total, uint8_t *host_rgba8, size_t n_bytes);
rsrt_status rsrt_multi_get_stats(rsrt_multi *m, rsrt_stats *out); /* counters summed, times = max over devices */

/* -- accumulator: cumulative_light_texture (hdr.rs:217-223) ----------------------------------
 * Library-owned W*H RGBA32F sum on the device, (re)allocated and zeroed when the resolution
 * changes (State::resize).  rsrt_accumulator_bind lets the caller own it instead (a device
 * pointer of width*height*4 floats, e.g. a torch tensor that a RCCL reduce will read); pass NULL
 * to go back to the internal one. */
rsrt_status rsrt_accumulator_resize(rsrt_context *ctx, uint32_t width, uint32_t height);
rsrt_status rsrt_accumulator_bind(rsrt_context *ctx, void *device_rgba32f, uint32_t width, uint32_t height);
rsrt_status rsrt_accumulator_clear(rsrt_context *ctx);
rsrt_status rsrt_accumulator_download(rsrt_context *ctx, float *host_rgba, size_t n_floats);
/* out_texture: mean = sum / sample_total rounded to binary16 (shader.wgsl:1369-1372) */
rsrt_status rsrt_resolve_mean_f16(rsrt_context *ctx, uint32_t sample_total, uint16_t *host_rgba16f, size_t n_halfs);

/* The reference's developer views — what `main` writes to out_texture instead of a render when `dev_index` (bind group 1 binding 3; key
 * bindings src/camera.rs:283) is 2 or 3, shader.wgsl:1314-1338 — for the bound accumulator's width x height, as RGBA binary16:
 *   3  the HDRI: texel (x, y) of environment `environment_index`, saturated, alpha 0 (zeros outside the map); the buffer's contents going in
 *      are ignored;
 *   2  draws of the alias table: every pixel draws 20 texel indices (seeded by its pixel index and `sample_count`, like a render's pixel) and
 *      each draw adds 0.1 / 20 to THAT texel's position in out_texture — read, add in f32, store as binary16, alpha 0.  host_rgba16f_inout holds
 *      out_texture as the previous frame left it and receives the new one.  The shader's invocations race on the texture (which draws survive
 *      is up to the GPU); this call returns the frame in which every draw lands.
 * The accumulator is not touched. */
rsrt_status rsrt_debug_view_f16(rsrt_context *ctx, uint32_t dev_index, uint32_t environment_index, uint32_t sample_count,
                                uint16_t *host_rgba16f_inout, size_t n_halfs);

/* The display pass (src/shaders/hdr.wgsl `fs_main`, src/hdr.rs:162-200): mean through binary16 ->
 * ACES fit (negatives -> magenta) -> sRGB 8-bit, as the *Srgb surface stores it.  RGBA8, alpha 255.
 * Arithmetic published in include/rsrt_tonemap.h. */
rsrt_status rsrt_display_srgb8(rsrt_context *ctx, uint32_t sample_total, uint8_t *host_rgba8, size_t n_bytes);

/* -- render: State::render's compute pass (state.rs:808-824), batched over samples -----------
 * Adds samples [sample_begin, sample_begin + sample_count) of every owned pixel into the
 * accumulator (alpha := 1), in increasing sample order per pixel, on `hip_stream`
 * (a hipStream_t, NULL = the context's own stream).  Asynchronous; rsrt_synchronize or
 * rsrt_accumulator_download/rsrt_get_stats wait for it. */
rsrt_status rsrt_render(rsrt_context *ctx, const rsrt_camera *camera, uint32_t width, uint32_t height,
                        uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces,
                        uint32_t environment_index, uint32_t flags, void *hip_stream);
rsrt_status rsrt_synchronize(rsrt_context *ctx);
rsrt_status rsrt_get_stats(rsrt_context *ctx, rsrt_stats *out);

/* -- ray-query probe: cast_ray / cast_ray_bvh for a batch of rays (shader.wgsl:469-601) -------
 * Exists for parity tests of traversal + intersection without the RNG: out records are
 * {did_hit u32, distance f32, hit_point 3xf32, normal 3xf32, material_id u32} = 36 bytes.
 * mode bit 0: 0 = cast_ray (BVH, then the brute-force fallback), 1 = cast_ray_bvh only;
 * mode bits 1-3, the traversal: 0 = threaded tree walk, 1 = the first kernel's stack walk, 2 = tree walk with typed leaf
 *   loops, 3 = flat loop over the leaf boxes (house, default, cube in production), 4 = fixed-order walk with ties
 *   decided by tabulated visiting ranks, 5 = wide walk (4-wide nodes, one ray a lane), 6 = cooperative wide walk (the same
 *   nodes; a wave's rays as (ray, node) / (ray, leaf) work items: suzanne and anything bigger in production) — the very device
 *   functions rt_render_pool_kernel's TRACE stage calls; a scene that does not qualify is RSRT_ERR_INVALID_ARGUMENT;
 * mode bit 4: read the scene from LDS exactly as the production kernel stages it for that traversal (whole image, or
 *   for mid-size scenes the nodes + escape links / the pre-order nodes) instead of from global memory.  Host pointers. */
typedef struct rsrt_hit {
    uint32_t did_hit;
    float distance;
    float hit_point[3];
    float normal[3];
    uint32_t material_id;
} rsrt_hit;
rsrt_status rsrt_cast_rays(rsrt_context *ctx, uint32_t n_rays, const float *origins_xyz, const float *directions_xyz,
                           uint32_t mode, uint32_t flags, rsrt_hit *out);

/* The wide walk's tree (csrc/hip/rt_device.h, trace_wide) for a BVH, built on the HOST exactly as rsrt_upload_scene builds it
 * for the device (no GPU needed; tests/test_wide_tree.py walks it on the CPU): the binary tree collapsed into 4-wide nodes, node 0
 * the root, a node's interior children consecutive, groups of siblings in the order their parents are expanded (largest box first).  wnodes_out (may be NULL): 32 floats per wide node, 8 x {x, y, z, word} — slot k's exact box is {[2k].xyz,
 * [2k + 1].xyz}; the words are the node's: [0] first interior child's node index (interior children come first and are consecutive)
 * | interior-slot mask << 26, [1] first record of the node's leaf children, [2] / [3] which of the 32 records from there are
 * triangles / planes, [4 + k] slot k's records as a mask from there (0: not a leaf); *n_wnodes: capacity in, count out; old_of_new (may be NULL):
 * for every record index the walk uses (whole leaves are reordered so that a wide node's leaf records are contiguous), the
 * index into `primitives` as given.  RSRT_ERR_INVALID_ARGUMENT: the BVH does not qualify (boxes that do not nest, leaves of
 * more than 8 records or that share records, a wide tree of more than 25 levels: the walk's eight stack registers + sixteen overflow words) and keeps the fixed-order walk. */
rsrt_status rsrt_wide_tree_build(const rsrt_primitive_info *primitives, uint32_t n_primitives, const rsrt_bvh_node *bvh_nodes, uint32_t n_bvh_nodes,
                                 float *wnodes_out, uint32_t *n_wnodes, uint32_t *old_of_new);

/* Diagnostic words of an instrumented build (-DRT_INSTRUMENT: loop-trip counters behind
 * tools/simd_efficiency.py); all zero in the product build. Cumulative since context creation. */
rsrt_status rsrt_get_debug_counters(rsrt_context *ctx, uint64_t out[32]);
/* ... and the lanes that passed each region mark (rt_math.h RT_MARK; tools/ledger.py sets them against the regions' instruction counts).
 * Read and reset; all zero in the product build. */
rsrt_status rsrt_get_region_counters(rsrt_context *ctx, uint64_t out[32]);

/* Exhaustive device self-test of the numeric contract's one shortcut: the 3-instruction reciprocal used for
 * 1/x (rt_math.h, rt_rcp) against the compiler's correctly rounded division, over all 2^32 f32 bit patterns
 * (~0.1 s).  out[0] = inputs where the bits differ (must be 0), out[1] = inputs that took the short path,
 * out[2] = of those, how many the bare v_rcp_f32 gets wrong (shows the comparison is live), out[3] = smallest
 * failing bit pattern or UINT64_MAX. */
rsrt_status rsrt_selftest_numerics(rsrt_context *ctx, uint64_t out[4]);

/* Library / device description, for logs: "librsrt <version>; <device name>; <CUs> CUs". */
const char *rsrt_describe(rsrt_context *ctx);
/* 16 hex digits: sha256 over the kernel sources this library was compiled from (plus any experiment knob).  The
 * rocprofv3 summaries under profiles/ carry the id of the library they were taken on; bench.py attaches a profile
 * to its roofline object only when the ids agree. */
const char *rsrt_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* RSRT_H */
