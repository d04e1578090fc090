/*
 * rsrt_host.h — C-ABI of the host-side preprocessing (librsrt_host.so, CPU only).
 *
 * The reference does this work in Rust before it uploads anything;
 * a Rust host that keeps its own Scene/BVH/alias code does not need this library at all — it
 * passes its buffers straight to rsrt.h.  It exists so that a C/C++/Python host can produce
 * the SAME buffers: each function restates one reference function in C++ with the same f32
 * operation order.
 *
 *   rsrt_scene_load_toml     Scene::load_toml            src/scene.rs:233-441, src/mesh.rs:29-113
 *   rsrt_build_bvh           build_bvh                   src/bvh.rs:13-337
 *   rsrt_plane_to_uniform    Plane::to_uniform           src/scene.rs:190-201
 *   rsrt_camera_uniform      CameraUniform::new          src/camera.rs:26-28, :111-119
 *   rsrt_alias_table_build   AliasTable::build_by_luminance  src/environments.rs:96-187
 *   rsrt_load_hdr            image::load_from_memory(..).into_rgb32f() for Radiance .hdr  src/state.rs:119-132
 *   rsrt_camera_(de)serialize Camera::serialize / deserialize (--state)  src/camera.rs:30-89
 *   rsrt_display_srgb8_host  hdr.wgsl aces_tone_map + sRGB surface write  src/shaders/hdr.wgsl:3-22
 *   rsrt_write_png/_pfm      image output (the reference only presents to a window)
 *   rsrt_synth_environment   stand-in for the two HDRIs the checkout lacks (state.rs:119-122;
 *                            .MISSING_LARGE_BLOBS) — deterministic formula, DESIGN.md §env
 */
#ifndef RSRT_HOST_H
#define RSRT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rsrt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsrt_scene rsrt_scene;

typedef struct rsrt_scene_counts {
    uint32_t n_materials, n_spheres, n_planes, n_vertices, n_normals, n_triangles, n_primitives, n_bvh_nodes;
    uint32_t bvh_depth;
} rsrt_scene_counts;

/* Camera as the scene file / --state describe it (src/camera.rs:14-21); angles in radians. */
typedef struct rsrt_camera_desc {
    float pos[3];
    float yaw, pitch, fov_y;
} rsrt_camera_desc;

/* Scene::load_toml + PackedMeshes::pack_meshes + build_bvh.  On failure returns non-zero and
 * writes the reference's error text (src/scene.rs:236-249, :334-351, :413-427) into err. */
int rsrt_scene_load_toml(const char *path, rsrt_scene **out, char *err, size_t err_len);
void rsrt_scene_free(rsrt_scene *scene);
void rsrt_scene_get_counts(const rsrt_scene *scene, rsrt_scene_counts *out);
const rsrt_material *rsrt_scene_materials(const rsrt_scene *scene);
const rsrt_sphere *rsrt_scene_spheres(const rsrt_scene *scene);
const rsrt_plane_desc *rsrt_scene_plane_descs(const rsrt_scene *scene);
const rsrt_plane *rsrt_scene_planes(const rsrt_scene *scene);
const rsrt_vec3 *rsrt_scene_vertices(const rsrt_scene *scene);
const rsrt_vec3 *rsrt_scene_normals(const rsrt_scene *scene);
const rsrt_triangle *rsrt_scene_triangles(const rsrt_scene *scene);
const rsrt_primitive_info *rsrt_scene_primitives(const rsrt_scene *scene);
const rsrt_bvh_node *rsrt_scene_bvh_nodes(const rsrt_scene *scene);
void rsrt_scene_get_camera(const rsrt_scene *scene, rsrt_camera_desc *out);

/* build_bvh.  primitives_out: n_spheres+n_planes+n_triangles entries; nodes_out: room for
 * 2*that-1.  Returns 0 and the node count / tree depth; non-zero for an empty scene
 * (the reference asserts, src/bvh.rs:222). */
int rsrt_build_bvh(const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                   const rsrt_vec3 *vertices, uint32_t n_vertices, const rsrt_triangle *triangles, uint32_t n_triangles,
                   rsrt_primitive_info *primitives_out, rsrt_bvh_node *nodes_out, uint32_t *n_nodes_out,
                   uint32_t *depth_out);

void rsrt_plane_to_uniform(const rsrt_plane_desc *in, rsrt_plane *out);
void rsrt_camera_uniform(const rsrt_camera_desc *in, rsrt_camera *out);

/* rgb: width*height*3 f32 (image::Rgb32FImage), rows top to bottom. out: width*height entries. */
int rsrt_alias_table_build(uint32_t width, uint32_t height, const float *rgb, rsrt_alias_entry *out, uint32_t *leftover_out);

/* Deterministic synthetic equirectangular sky (gradient + sun lobe); rgba_out: w*h*4, alpha 0. */
int rsrt_synth_environment(uint32_t width, uint32_t height, float *rgba_out);

/* Radiance RGBE (.hdr): rgb_out = malloc'ed width*height*3 f32, rows top to bottom; free with rsrt_free.
 * value = mantissa * 2^(e-136), (0,0,0) when e == 0 — the `image` 0.25 hdr decoder's rule. */
int rsrt_load_hdr(const char *path, uint32_t *width, uint32_t *height, float **rgb_out, char *err, size_t err_len);
void rsrt_free(void *p);

/* 24 bytes [pos.xyz, yaw, pitch, fov_y] little-endian f32 <-> standard base64 (32 chars + NUL).
 * Deserialize errors carry the reference's text ("Couldn't deserialize camera: binary data (N bytes) not 24 bytes"). */
void rsrt_camera_serialize(const rsrt_camera_desc *cam, char out[33]);
int rsrt_camera_deserialize(const char *encoded, rsrt_camera_desc *out, char *err, size_t err_len);

/* Display transform on the CPU (same inline code as the HIP kernel behind rsrt_display_srgb8). */
void rsrt_display_srgb8_host(const float *sum_rgba, size_t n_pixels, uint32_t sample_total, uint8_t *out_rgba8);

/* Image files: 8-bit RGBA PNG (stored deflate), 32-bit float PFM from a buffer with `stride_floats` per pixel. */
int rsrt_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8);
int rsrt_write_pfm(const char *path, uint32_t width, uint32_t height, const float *rgb, uint32_t stride_floats);

#ifdef __cplusplus
}
#endif
#endif /* RSRT_HOST_H */
