"""The property the flat small-scene traversal (trace_flat, rt_device.h) rests on: with the reference's slab
test (ray_intersects_bounds, shader.wgsl:262-293) a ray whose reciprocal direction is FINITE and that hits a
box hits every box that contains it — f32 rounding is monotone, so the parent's slab interval contains the
child's.  Hence cast_ray_bvh's unpruned walk tests exactly the leaves whose own boxes are hit, whatever the
tree above them looks like.  With an infinite reciprocal (a zero or subnormal direction component) 0 * inf
= NaN breaks this — the second test shows it does — so those rays keep the tree walk."""
import numpy as np


def slab_hit(bmin, bmax, o, d):
    """Literal numpy-f32 restatement; arrays [..., 3]."""
    with np.errstate(all="ignore"):
        inv = np.float32(1.0) / d
        t0 = np.zeros(o.shape[:-1], np.float32)
        t1 = np.full(o.shape[:-1], np.inf, np.float32)
        alive = np.ones(o.shape[:-1], bool)
        for a in range(3):
            tn = (bmin[..., a] - o[..., a]) * inv[..., a]
            tf = (bmax[..., a] - o[..., a]) * inv[..., a]
            sw = tn > tf
            tn, tf = np.where(sw, tf, tn), np.where(sw, tn, tf)
            t0 = np.where(tn > t0, tn, t0)
            t1 = np.where(tf < t1, tf, t1)
            alive &= ~(t0 > t1)
        return alive


def test_a_ray_that_hits_a_box_hits_every_enclosing_box():
    rng = np.random.default_rng(7)
    n = 400_000
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, -2.0, 1e-38, -1e-38, 1e-45, 3.0, -3.0, 1e30, -1e30, np.inf, -np.inf], np.float32)
    grid = np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 3.0], np.float32)
    n_violations_nonfinite = 0
    for mode in range(4):
        # boxes on a small grid (equalities with the origin are common), child inside parent
        c = np.sort(rng.choice(grid, (n, 3, 4)), axis=2).astype(np.float32)  # pmin <= cmin <= cmax <= pmax per axis
        pmin, cmin, cmax, pmax = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
        if mode >= 2:  # generic floats
            c = np.sort(rng.normal(0, 2, (n, 3, 4)).astype(np.float32), axis=2)
            pmin, cmin, cmax, pmax = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
        o = rng.choice(grid, (n, 3)).astype(np.float32) if mode % 2 == 0 else rng.normal(0, 2, (n, 3)).astype(np.float32)
        d = rng.normal(0, 1, (n, 3)).astype(np.float32)
        m = rng.random((n, 3)) < 0.35
        d = np.where(m, rng.choice(special, (n, 3)), d).astype(np.float32)
        child = slab_hit(cmin, cmax, o, d)
        parent = slab_hit(pmin, pmax, o, d)
        with np.errstate(all="ignore"):
            finite = np.isfinite(np.float32(1.0) / d).all(axis=1)
        bad = child & ~parent
        n_violations_nonfinite += int((bad & ~finite).sum())
        bad &= finite
        assert not bad.any(), (mode, int(bad.sum()), o[bad][:3], d[bad][:3], pmin[bad][:3], cmin[bad][:3], cmax[bad][:3], pmax[bad][:3])
        assert (child & finite).sum() > 1000 and (parent & ~child & finite).sum() > 1000  # the sample exercises both outcomes
    assert n_violations_nonfinite > 0  # ... and the exclusion is needed: with 1/0 = inf the implication does fail
