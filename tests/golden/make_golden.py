"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle (strict build).

    python tests/golden/make_golden.py

The reference itself can be neither built nor run here, so these vectors are the
oracle's own outputs, frozen: they pin the oracle against accidental change and let the GPU box
check the HIP path without /root/reference.  Inputs (scene files, the 64x32 environment) are
committed next to them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import util  # noqa: E402
import rsoderh_raytracing_amd as R  # noqa: E402


def main():
    env = R.Environment.synthetic(64, 32)
    oal, left = oracle.alias_table(env.rgba[:, :, :3])
    np.savez_compressed(os.path.join(HERE, "env_64x32.npz"), rgba=env.rgba, alias=oal.view(np.uint32).reshape(-1, 4),
                        leftover=left)
    env8 = R.Environment.synthetic(8, 4)
    al8, left8 = oracle.alias_table(env8.rgba[:, :, :3])
    np.savez_compressed(os.path.join(HERE, "alias_8x4.npz"), rgba=env8.rgba, alias=al8.view(np.uint32).reshape(-1, 4),
                        leftover=left8)
    oenv = oracle.Env(env.rgba, oal)
    rng = np.random.default_rng(20260204)
    for name in ["default", "house", "cube", "suzanne", "spheres_only"]:
        sc = R.Scene.load_toml(util.scene_path(name))
        osc = util.oracle_scene(sc)
        prims, nodes, depth = oracle.build_bvh(osc.spheres, sc.plane_descs.view(oracle.PLANE_SRC), osc.vertices, osc.triangles)
        out = dict(bvh_prims=prims.view(np.uint32).reshape(-1, 2), bvh_nodes=nodes.view(np.uint32).reshape(-1, 12), depth=depth)
        cam = sc.camera_uniform().view(oracle.CAMERA)
        for spp, mb in [(4, 3), (16, 10)]:
            if name in ("cube", "suzanne", "spheres_only") and spp == 16:
                continue
            img, st = oracle.render(osc, oenv, cam, 64, 64, 0, spp, mb)
            out["sum_%dspp_%db" % (spp, mb)] = img
            out["rays_%dspp_%db" % (spp, mb)] = np.array([st["paths"], st["ext_rays"], st["shadow_rays"]], np.uint64)
        # 1k-ray batch: half from the camera, half from random points, random directions
        o = np.concatenate([np.tile(np.float32(sc.camera_desc["pos"][0]), (512, 1)),
                            rng.uniform(-3, 3, size=(512, 3)).astype(np.float32)])
        d = rng.normal(size=(1024, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        hits = oracle.cast_rays(osc, o, d, 0, 0)
        hits_bvh = oracle.cast_rays(osc, o, d, 1, 0)
        out.update(ray_o=o, ray_d=d, hits=hits.view(np.uint32).reshape(-1, 9), hits_bvh=hits_bvh.view(np.uint32).reshape(-1, 9))
        np.savez_compressed(os.path.join(HERE, "scene_%s.npz" % name), **out)
        print(name, "nodes", len(nodes), "depth", depth, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})


if __name__ == "__main__":
    main()
