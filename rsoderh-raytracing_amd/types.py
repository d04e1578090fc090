"""numpy views of the POD layouts in include/rsrt_types.h (the reference's encase layouts,
src/state.rs:394-458)."""
import numpy as np

MATERIAL = np.dtype({"names": ["color", "roughness", "metallic", "emission"],
                     "formats": [("f4", 3), "f4", "f4", ("f4", 3)], "offsets": [0, 12, 16, 32], "itemsize": 48})
SPHERE = np.dtype({"names": ["pos", "radius", "material_id"], "formats": [("f4", 3), "f4", "u4"],
                   "offsets": [0, 12, 16], "itemsize": 32})
PLANE = np.dtype({"names": ["pos", "normal", "base_change_matrix", "material_id"],
                  "formats": [("f4", 3), ("f4", 3), ("f4", (3, 4)), "u4"], "offsets": [0, 16, 32, 80], "itemsize": 96})
VEC3 = np.dtype({"names": ["v"], "formats": [("f4", 3)], "offsets": [0], "itemsize": 16})
TRIANGLE = np.dtype([("vertex_0", "u4"), ("vertex_1", "u4"), ("vertex_2", "u4"), ("normal_0", "u4"), ("normal_1", "u4"),
                     ("normal_2", "u4"), ("material_id", "u4")])
PRIMITIVE_INFO = np.dtype([("primitive_type", "u4"), ("index", "u4")])
BVH_NODE = np.dtype({"names": ["bounds_min", "bounds_max", "primitives_or_second_child_index", "primitives_len", "split_axis"],
                     "formats": [("f4", 3), ("f4", 3), "u4", "u4", "u4"], "offsets": [0, 16, 32, 36, 40], "itemsize": 48})
ALIAS_ENTRY = np.dtype([("probability", "f4"), ("alias_index", "u4"), ("pmf", "f4"), ("_pad", "u4")])
CAMERA = np.dtype({"names": ["pos", "rot_transform", "fov_y"], "formats": [("f4", 3), ("f4", (3, 4)), "f4"],
                   "offsets": [0, 16, 64], "itemsize": 80})
PLANE_DESC = np.dtype([("pos", "f4", 3), ("forward", "f4", 3), ("right", "f4", 3), ("material_id", "u4")])
CAMERA_DESC = np.dtype([("pos", "f4", 3), ("yaw", "f4"), ("pitch", "f4"), ("fov_y", "f4")])
HIT = np.dtype([("did_hit", "u4"), ("distance", "f4"), ("hit_point", "f4", 3), ("normal", "f4", 3), ("material_id", "u4")])

assert (MATERIAL.itemsize, SPHERE.itemsize, PLANE.itemsize, VEC3.itemsize, TRIANGLE.itemsize, PRIMITIVE_INFO.itemsize,
        BVH_NODE.itemsize, ALIAS_ENTRY.itemsize, CAMERA.itemsize, PLANE_DESC.itemsize, HIT.itemsize) == \
    (48, 32, 96, 16, 28, 8, 48, 16, 80, 40, 36)
