/*
 * rt_oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (rsoderh-raytracing_amd/, include/rsrt.h)
 * never links, imports or calls anything under oracle/.
 *
 * What it is: a scalar CPU restatement of the reference's per-pixel radiance integrator
 * (/root/reference/src/shaders/shader.wgsl, WGSL) and of the host preprocessing whose output
 * the integrator consumes (src/bvh.rs, src/scene.rs, src/mesh.rs, src/environments.rs,
 * src/camera.rs).  Every function in rt_oracle.cpp cites the reference file:line it follows.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests, no golden vectors and no CPU
 * path, and it can be neither compiled (no Rust/WGSL toolchain, HDRIs missing) nor imported
 * here.  The only external pins are the hand-derivable known-answer
 * values of SURVEY.md §8(c), checked in tests/test_oracle_kat.py.
 *
 * All array layouts are the encase std430/std140 layouts the reference uploads
 * (src/state.rs:394-458), so a fixture dumped from the reference would load unchanged.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shader.wgsl:86-91, stride 48 */
typedef struct { float color[3]; float roughness; float metallic; float _p0[3]; float emission[3]; float _p1; } orc_material;
/* shader.wgsl:93-97, stride 32 */
typedef struct { float pos[3]; float radius; uint32_t material_id; uint32_t _p[3]; } orc_sphere;
/* shader.wgsl:99-110, stride 96; m = base_change_matrix columns (vec4 stride) */
typedef struct { float pos[3]; float _p0; float normal[3]; float _p1; float m[3][4]; uint32_t material_id; uint32_t _p2[3]; } orc_plane;
/* array<vec3<f32>>, stride 16 */
typedef struct { float v[3]; float _p; } orc_vec3;
/* shader.wgsl:112-126, stride 28 */
typedef struct { uint32_t v0, v1, v2, n0, n1, n2, material_id; } orc_triangle;
/* shader.wgsl:128-133 */
typedef struct { uint32_t type, index; } orc_prim_info;
/* shader.wgsl:135-146, stride 48 */
typedef struct { float bmin[3]; float _p0; float bmax[3]; float _p1; uint32_t idx, len, axis, _p2; } orc_bvh_node;
/* shader.wgsl:158-170 */
typedef struct { float probability; uint32_t alias_index; float pmf; uint32_t _pad; } orc_alias_entry;
/* shader.wgsl:1-6, 80 B */
typedef struct { float pos[3]; float _p0; float rot[3][4]; float fov_y; float _p1[3]; } orc_camera;
/* src/scene.rs:182-188 (the non-uniform plane the BVH builder sees) */
typedef struct { float pos[3]; float forward[3]; float right[3]; uint32_t material_id; } orc_plane_src;

typedef struct {
    const orc_material *materials; uint32_t n_materials;
    const orc_sphere *spheres;     uint32_t n_spheres;
    const orc_plane *planes;       uint32_t n_planes;
    const orc_vec3 *vertices;      uint32_t n_vertices;
    const orc_vec3 *normals;       uint32_t n_normals;
    const orc_triangle *triangles; uint32_t n_triangles;
    const orc_prim_info *prims;    uint32_t n_prims;
    const orc_bvh_node *nodes;     uint32_t n_nodes;
} orc_scene;

typedef struct {
    uint32_t width, height;
    const float *rgba;             /* width*height*4, alpha 0 (src/texture.rs:112-115) */
    const orc_alias_entry *alias;  /* width*height */
} orc_env;

typedef struct {
    uint64_t paths, ext_rays, shadow_rays;
    uint64_t nodes_visited, prim_refs, sphere_tests, plane_tests, tri_tests;
    uint64_t closest_tri, nee_events, escapes, shaded_hits;
    uint64_t fallback_sphere_tests, fallback_plane_tests;
    uint64_t f32_ops, int_ops; /* liboracle_ops.so (-DORC_COUNT_OPS) only, else 0: the f32 operations (add, sub, mul, div, fma, sqrt,
                                  floor, min, max, comparisons) and the u32 operations / conversions the integrator executed */
} orc_stats;

enum {
    ORC_FLAG_PRUNE = 1,        /* skip nodes whose slab entry t > current best (SURVEY A7 i).  NOT exactly
                                  result-preserving in f32: 2 of 5.3e8 paths differ on house 1080p x 256 spp */
    ORC_FLAG_ANYHIT_SHADOW = 2 /* NEE shadow query exits at the first hit (SURVEY A7 ii); exact */
};

typedef struct { uint32_t did_hit; float distance; float hit_point[3]; float normal[3]; uint32_t material_id; } orc_hit;

/* ---- host preprocessing (src/bvh.rs, src/scene.rs, src/environments.rs, src/camera.rs) ---- */
/* Returns node count; prims_out must hold n_spheres+n_planes+n_triangles, nodes_out 2*that. */
int orc_build_bvh(const orc_sphere *spheres, uint32_t n_spheres, const orc_plane_src *planes, uint32_t n_planes,
                  const orc_vec3 *vertices, const orc_triangle *triangles, uint32_t n_triangles,
                  orc_prim_info *prims_out, orc_bvh_node *nodes_out, uint32_t *depth_out);
void orc_alias_table(uint32_t width, uint32_t height, const float *rgb /* w*h*3 */, orc_alias_entry *out,
                     uint32_t *leftover_out);
void orc_plane_to_uniform(const orc_plane_src *in, orc_plane *out);
void orc_camera_uniform(const float pos[3], float yaw, float pitch, float fov_y, orc_camera *out);

/* ---- the integrator (shader.wgsl main / trace_ray) ---- */
/* Adds samples [sample_begin, sample_begin+sample_count) of every pixel into sum_rgba (W*H*4,
 * alpha set to 1), exactly as `sample_count` successive reference frames would. */
int orc_render(const orc_scene *scene, const orc_env *env, const orc_camera *cam, uint32_t width, uint32_t height,
               uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces, uint32_t flags, int n_threads,
               float *sum_rgba, orc_stats *stats);

/* ---- pieces, exported for known-answer tests ---- */
uint32_t orc_rng_seed(uint32_t pixel_index, uint32_t sample_index);
uint32_t orc_rng_next_u32(uint32_t *state);
float orc_u32_to_uniform(uint32_t r);
void orc_cast_ray_sphere(const float o[3], const float d[3], const orc_sphere *s, orc_hit *out);
void orc_cast_ray_plane(const float o[3], const float d[3], const orc_plane *p, orc_hit *out);
void orc_cast_ray_triangle(const float o[3], const float d[3], const orc_scene *scene, const orc_triangle *t, orc_hit *out);
/* mode 0: cast_ray (BVH + brute-force fallback); mode 1: cast_ray_bvh only */
void orc_cast_rays(const orc_scene *scene, uint32_t n, const float *origins, const float *dirs, uint32_t mode,
                   uint32_t flags, orc_hit *out);
/* BSDF pieces in the local frame (shader.wgsl:1053-1202) */
void orc_bsdf_eval_local(const orc_material *m, const float wo[3], const float wi[3], float out[3]);
float orc_bsdf_pdf_local(const orc_material *m, const float wo[3], const float wi[3]);
/* returns pdf; dir_out zero vector on failure, scattering_out as the shader returns it */
float orc_bsdf_sample(const orc_material *m, const float ray_dir[3], const float normal[3], uint32_t *rng,
                      float dir_out[3], float scattering_out[3]);
void orc_sky_light(const orc_env *env, const float dir[3], float out[3]);
float orc_environment_direction_pdf(const orc_env *env, const float dir[3]);
float orc_sample_environment(const orc_env *env, uint32_t *rng, float dir_out[3], float radiance_out[3]);
void orc_direction_to_uv(const float dir[3], float uv[2]);
float orc_detmath(int fn, float a, float b); /* 0 sin, 1 cos, 2 atan2(a,b), 3 asin */

#ifdef __cplusplus
}
#endif
#endif
