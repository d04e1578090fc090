/*
 * rsrt_types.h — plain-old-data layouts crossing the rsrt C-ABI.
 *
 * These are byte-for-byte the encase std430 / std140 layouts the reference host writes into
 * its wgpu buffers (reference: src/state.rs:249-287 uniforms, :394-458 scene storage buffers,
 * src/environments.rs:30-55 environment buffers; WGSL declarations src/shaders/shader.wgsl:1-170),
 * so the Rust host can hand the very same `Vec<u8>` it gives `create_buffer_init` to this
 * library (INTEGRATION.md shows the binding).
 */
#ifndef RSRT_TYPES_H
#define RSRT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material — shader.wgsl:86-91, src/scene.rs:16-23; stride 48 */
typedef struct rsrt_material {
    float color[3];
    float roughness;
    float metallic;
    float _pad0[3];
    float emission[3];
    float _pad1;
} rsrt_material;

/* Sphere — shader.wgsl:93-97, src/scene.rs:166-171; stride 32 */
typedef struct rsrt_sphere {
    float pos[3];
    float radius;
    uint32_t material_id;
    uint32_t _pad[3];
} rsrt_sphere;

/* UniformPlane — shader.wgsl:99-110, src/scene.rs:209-222; stride 96.
 * base_change_matrix is column-major with vec4 column stride. */
typedef struct rsrt_plane {
    float pos[3];
    float _pad0;
    float normal[3];
    float _pad1;
    float base_change_matrix[3][4];
    uint32_t material_id;
    uint32_t _pad2[3];
} rsrt_plane;

/* array<vec3<f32>> element (vertices, normals) — shader.wgsl:217-221; stride 16 */
typedef struct rsrt_vec3 {
    float v[3];
    float _pad;
} rsrt_vec3;

/* TriangleUniform — shader.wgsl:112-126, src/mesh.rs:150-165; stride 28 */
typedef struct rsrt_triangle {
    uint32_t vertex_0, vertex_1, vertex_2;
    uint32_t normal_0, normal_1, normal_2;
    uint32_t material_id;
} rsrt_triangle;

/* PrimitiveInfoUniform — shader.wgsl:128-133, src/bvh.rs:81-87; type 0 sphere, 1 plane, 2 triangle */
typedef struct rsrt_primitive_info {
    uint32_t primitive_type;
    uint32_t index;
} rsrt_primitive_info;

/* BvhNodeUniform — shader.wgsl:135-146, src/bvh.rs:89-99; stride 48 */
typedef struct rsrt_bvh_node {
    float bounds_min[3];
    float _pad0;
    float bounds_max[3];
    float _pad1;
    uint32_t primitives_or_second_child_index;
    uint32_t primitives_len;
    uint32_t split_axis;
    uint32_t _pad2;
} rsrt_bvh_node;

/* AliasEntry — shader.wgsl:158-170, src/environments.rs:200-213 */
typedef struct rsrt_alias_entry {
    float probability;
    uint32_t alias_index;
    float pmf;
    uint32_t _pad;
} rsrt_alias_entry;

/* CameraUniform — shader.wgsl:1-6, src/camera.rs:103-119; 80 bytes, rot_transform column-major */
typedef struct rsrt_camera {
    float pos[3];
    float _pad0;
    float rot_transform[3][4];
    float fov_y;
    float _pad1[3];
} rsrt_camera;

/* Plane as the scene file describes it — src/scene.rs:182-188 (input of the BVH builder and of
 * Plane::to_uniform) */
typedef struct rsrt_plane_desc {
    float pos[3];
    float forward[3];
    float right[3];
    uint32_t material_id;
} rsrt_plane_desc;

#ifdef __cplusplus
}
#endif
#endif /* RSRT_TYPES_H */
