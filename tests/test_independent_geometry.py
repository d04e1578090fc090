"""A SECOND, independent reading of the reference's intersection code, to catch a common-mode misreading shared by the
oracle and the HIP kernel (VERDICT r1 "What's weak" #1): cast_ray_sphere / cast_ray_plane / cast_ray_triangle
(/root/reference/src/shaders/shader.wgsl:295-466) and Plane::to_uniform (src/scene.rs:190-201) restated here in numpy
float64, straight from the WGSL text, with no BVH at all (closest hit over ALL primitives — what cast_ray computes
whenever the boxes are conservative) and numpy's own matrix inverse for the plane basis.  Compared with the oracle's
cast_ray on random and on adversarial rays: same hit / miss, same primitive's material, t, hit point and normal to 1e-4,
except where float64 and f32 can legitimately disagree (a decision within 1e-5 of its threshold, two candidates within
1e-4 of each other)."""
import numpy as np
import pytest

import oracle
import util
import rsoderh_raytracing_amd as R

INF = 1.70141183460469231732e38


def sphere_hit(o, d, pos, radius):  # shader.wgsl:295-360
    l = o - pos
    a = d @ d
    b = 2 * (d @ l)
    c = l @ l - radius * radius
    disc = b * b - 4 * a * c
    if disc < 0:
        return None
    if disc == 0:
        t = -0.5 * b / a
    else:
        q = -0.5 * (b + np.sqrt(disc)) if b > 0 else -0.5 * (b - np.sqrt(disc))  # select(f, t, cond): t when cond
        t0, t1 = q / a, c / q
        if t0 < 1e-4:
            t = t1
        elif t1 < 1e-4:
            t = t0
        else:
            t = min(t0, t1)
    if t < 1e-4:
        return None
    p = o + d * t
    n = (p - pos) / np.linalg.norm(p - pos)
    if (pos - o) @ (pos - o) - radius * radius < 1e-6:
        n = -n
    return t, p, n, (abs(t - 1e-4), abs(disc))


def plane_hit(o, d, pos, forward, right):  # shader.wgsl:362-406 + scene.rs:190-201
    n = np.cross(forward, right)
    n = n / np.linalg.norm(n)
    m = np.linalg.inv(np.stack([right, n, forward], axis=1))  # inverse of the matrix whose COLUMNS are right, n, forward
    den = n @ d
    if abs(den) < 1e-4:
        return None
    t = (n @ (pos - o)) / den
    if t < 1e-3:
        return None
    inter = o + d * t
    q = m @ (inter - pos)
    if q[0] < 0 or 1 < q[0] or q[2] < 0 or 1 < q[2]:
        return None
    nn = -n if o @ n < 0 else n  # the origin is NOT made relative to the plane (kept quirk)
    margin = min(abs(abs(den) - 1e-4), abs(t - 1e-3), abs(q[0]), abs(1 - q[0]), abs(q[2]), abs(1 - q[2]), abs(o @ n) if abs(o @ n) > 0 else 1.0)
    return t, inter, nn, (margin, 1.0)


def triangle_hit(o, d, a, b, c, n0, n1, n2):  # shader.wgsl:409-466
    e0, e1 = b - a, c - a
    p0 = np.cross(o - a, e0)
    p1 = np.cross(d, e1)
    det = e0 @ p1
    if abs(det) < 1e-8:
        return None
    inv = 1.0 / det
    u = ((o - a) @ p1) * inv
    v = (d @ p0) * inv
    if u < 0 or 1 < u:
        return None
    if v < 0 or 1 < u + v:
        return None
    t = (e1 @ p0) * inv
    if t < 1e-5:
        return None
    n = (1 - u - v) * n0 + u * n1 + v * n2
    n = n / np.linalg.norm(n)
    if n @ d > 0:
        n = -n
    margin = min(abs(u), abs(1 - u), abs(v), abs(1 - u - v), abs(t - 1e-5) * 1e3, abs(n @ d) * 10)
    return t, o + d * t, n, (margin, 1.0)


def brute_force(scene, o, d):
    """closest hit over every primitive, float64; returns (hit tuple or None, ambiguous flag)"""
    cands = []
    for s in scene.spheres:
        h = sphere_hit(o, d, s["pos"].astype(np.float64), float(s["radius"]))
        if h:
            cands.append(h + (int(s["material_id"]),))
    for p in scene.plane_descs:
        h = plane_hit(o, d, p["pos"].astype(np.float64), p["forward"].astype(np.float64), p["right"].astype(np.float64))
        if h:
            cands.append(h + (int(p["material_id"]),))
    V, N = scene.vertices["v"].astype(np.float64), scene.normals["v"].astype(np.float64)
    for t in scene.triangles:
        h = triangle_hit(o, d, V[t["vertex_0"]], V[t["vertex_1"]], V[t["vertex_2"]], N[t["normal_0"]], N[t["normal_1"]], N[t["normal_2"]])
        if h:
            cands.append(h + (int(t["material_id"]),))
    if not cands:
        return None, False
    cands.sort(key=lambda h: h[0])
    best = cands[0]
    # (house.toml lists its ground plane twice: two coincident candidates of one material are one answer, not an ambiguity)
    rivals = [h for h in cands[1:] if h[0] - best[0] < 1e-4 * max(1.0, best[0]) and (h[4] != best[4] or not np.allclose(h[2], best[2], atol=1e-6))]
    ambiguous = best[3][0] < 1e-5 or len(rivals) > 0
    return best, ambiguous


def near_misses(scene, o, d):
    """is some primitive within a hair of being hit / missed?  (then float64 and f32 may disagree on hit-or-miss)"""
    eps = 2e-5
    for delta in (np.array([eps, 0, 0]), np.array([0, eps, 0]), np.array([0, 0, eps])):
        for sgn in (1, -1):
            a, _ = brute_force(scene, o, d)
            b, _ = brute_force(scene, o + sgn * delta, d)
            if (a is None) != (b is None) or (a is not None and (a[4] != b[4] or abs(a[0] - b[0]) > 1e-3)):
                return True
    return False


@pytest.mark.parametrize("name", ["house", "default"])
def test_oracle_cast_ray_agrees_with_an_independent_float64_reading(name):
    sc = R.Scene.load_toml(util.scene_path(name))
    osc = util.oracle_scene(sc)
    rng = np.random.default_rng(7)
    n = 600
    # camera-like rays, rays from inside the geometry, rays starting ON surfaces (the acne thresholds)
    o = np.concatenate([np.tile([0.0, 1.0, 3.0], (n // 3, 1)), rng.uniform(-3, 3, (n // 3, 3)), rng.uniform(-2, 2, (n - 2 * (n // 3), 3)) * [1, 0, 1]])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    hits = oracle.cast_rays(osc, o32, d32, 0, 0)
    checked = agree = 0
    for i in range(n):
        oi, di = o32[i].astype(np.float64), d32[i].astype(np.float64)
        ref, ambiguous = brute_force(sc, oi, di)
        if ambiguous or near_misses(sc, oi, di):
            continue
        checked += 1
        h = hits[i]
        if ref is None:
            assert h["did_hit"] == 0, (i, h)
            assert h["distance"] == np.float32(INF)  # cast_ray's `result` initialiser on a total miss
        else:
            t, p, nrm, _, mat = ref
            assert h["did_hit"] == 1, (i, ref, h)
            assert h["material_id"] == mat, (i, ref, h)
            assert abs(h["distance"] - t) <= 1e-4 * max(1.0, t), (i, t, h["distance"])
            assert np.allclose(h["hit_point"], p, atol=2e-4 * max(1.0, t)), (i, p, h["hit_point"])
            assert np.allclose(h["normal"], nrm, atol=2e-3), (i, nrm, h["normal"])
        agree += 1
    assert checked > 0.7 * n and agree == checked, (checked, agree)
