"""CPU count model of the cooperative wide walk (TRAV 6, rt_coop.h) — no GPU: a wave traces a BATCH of rays through the 4-wide tree with
two wave-shared work stacks, (ray, node) items and (ray, leaf) items, 64 items a trip.  Counts node trips, leaf trips, lanes busy in each,
and the stacks' high-water marks, against the per-lane walk's figures.
    python tools/coop_sim.py [suzanne|grid] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ctypes as C
import numpy as np
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import state


def wide_tree(sc):
    L = state.lib()
    n = C.c_uint32(0)
    prims, nodes = np.ascontiguousarray(sc.primitives), np.ascontiguousarray(sc.bvh_nodes)
    args = (prims.ctypes.data_as(C.c_void_p), len(prims), nodes.ctypes.data_as(C.c_void_p), len(nodes))
    assert L.rsrt_wide_tree_build(*args, None, C.byref(n), None) == 0
    wn = np.zeros((n.value, 8, 4), np.float32)
    oon = np.zeros(len(prims), np.uint32)
    assert L.rsrt_wide_tree_build(*args, wn.ctypes.data_as(C.c_void_p), C.byref(n), oon.ctypes.data_as(C.c_void_p)) == 0
    return wn, oon


def load(name):
    if name == 'grid':
        import make_big_scene
        return R.Scene.load_toml(make_big_scene.make(4))
    import util
    return R.Scene.load_toml(util.scene_path(name))


def rays_of(sc, rng, n_cam_w=96, n_cam_h=54, sec_per_cam=2):
    cam = sc.camera_uniform()
    W, H = n_cam_w, n_cam_h
    ys, xs = np.mgrid[0:H, 0:W]
    fx = xs.ravel() + rng.random(W * H) - 0.5; fy = ys.ravel() + rng.random(W * H) - 0.5
    m = np.sin(float(cam['fov_y'][0]) / 2)
    rcs = np.stack([((fx / W) * 2 - 1) * m * W / H, -((fy / H) * 2 - 1) * m, -np.ones(W * H)], 1)
    rot = np.asarray(cam['rot_transform'][0], np.float64).reshape(3, -1)[:, :3]
    d = rcs @ rot; d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.asarray(cam['pos'][0], np.float64)[:3], (W * H, 1))
    nodes = sc.bvh_nodes
    n = len(nodes)
    bmin = np.asarray(nodes['bounds_min'], np.float32).reshape(n, -1)[:, :3]
    bmax = np.asarray(nodes['bounds_max'], np.float32).reshape(n, -1)[:, :3]
    leaf = nodes['primitives_len'] > 0
    lc = (bmin[leaf] + bmax[leaf]) / 2
    k = rng.integers(0, len(lc), sec_per_cam * W * H)
    o2 = lc[k] + rng.normal(0, 0.02, (len(k), 3)); d2 = rng.normal(size=(len(k), 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    return np.concatenate([o, o2]).astype(np.float32), np.concatenate([d, d2]).astype(np.float32)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'grid'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    fifo = len(sys.argv) > 3 and sys.argv[3] == 'fifo'  # node items oldest first (a queue) instead of newest first (a stack)
    sc = load(name)
    wn, _ = wide_tree(sc)
    w = wn.view(np.uint32)
    nw = len(wn)
    bmin = wn[:, 0::2, :3]; bmax = wn[:, 1::2, :3]  # [node][slot][3]
    wa = w[:, 0, 3]; first_child = (wa & 0x3FFFFFF).astype(np.int64); imask = (wa >> 26).astype(np.int64)
    lmask = w[:, 4:8, 3].astype(np.int64)  # [node][slot]
    leaf_len = np.array([[bin(int(m)).count('1') for m in row] for row in lmask])
    rng = np.random.default_rng(3)
    o, d = rays_of(sc, rng)
    with np.errstate(divide='ignore'):
        inv = (np.float32(1.0) / d).astype(np.float32)
    nr = len(o)
    perm = rng.permutation(nr)  # incoherent batches: a TRACE list mixes bounces and patches
    tot = dict(node_items=0, node_trips=0, leaf_items=0, leaf_trips=0, leaf_rec=0, leaf_rec_slots=0, leaf_pair_slots=0, max_ns=0, max_ls=0, batches=0)
    for b0 in range(0, nr - batch + 1, batch):
        rays = perm[b0:b0 + batch]
        ns = [(int(r), 0) for r in rays]  # node stack: (ray, node)
        ls = []  # leaf stack: (ray, n_records)
        tot['batches'] += 1
        while ns or ls:
            if len(ls) >= 64 or (not ns and ls):
                take = ls[-64:]; del ls[-64:]
                cnt = np.array([c for _, c in take])
                tot['leaf_trips'] += 1; tot['leaf_items'] += len(take); tot['leaf_rec'] += int(cnt.sum())
                tot['leaf_rec_slots'] += int(cnt.max()) * 64            # one record a trip
                tot['leaf_pair_slots'] += int(((cnt + 1) // 2).max()) * 64  # two records a trip (slots in units of a PAIR trip)
                continue
            if fifo: take = ns[:64]; del ns[:64]
            else: take = ns[-64:]; del ns[-64:]
            tot['node_trips'] += 1; tot['node_items'] += len(take)
            r = np.array([t[0] for t in take]); nd = np.array([t[1] for t in take])
            a = (bmin[nd] - o[r][:, None, :]) * inv[r][:, None, :]; bb = (bmax[nd] - o[r][:, None, :]) * inv[r][:, None, :]
            t0 = np.maximum(np.minimum(a, bb).max(axis=2), 0.0); t1 = np.maximum(a, bb).min(axis=2)
            hit = ~(t0 > t1)  # [item][slot]
            for j in range(len(take)):
                for k in range(4):
                    if not hit[j, k]: continue
                    if (imask[nd[j]] >> k) & 1: ns.append((int(r[j]), int(first_child[nd[j]] + k)))
                    elif leaf_len[nd[j], k]: ls.append((int(r[j]), int(leaf_len[nd[j], k])))
            tot['max_ns'] = max(tot['max_ns'], len(ns)); tot['max_ls'] = max(tot['max_ls'], len(ls))
    nrays = tot['batches'] * batch
    print('%s: %d wide nodes; %d rays in batches of %d (incoherent), node items %s' % (name, nw, nrays, batch, 'oldest first' if fifo else 'newest first'))
    print('  node items / ray %.2f, node trips / batch %.1f, lanes busy %.1f %%' % (tot['node_items'] / nrays, tot['node_trips'] / tot['batches'], 100.0 * tot['node_items'] / (64 * tot['node_trips'])))
    print('  leaf items / ray %.2f, records / ray %.2f (%.2f a leaf), leaf trips / batch %.1f, lanes busy %.1f %% of the item slots' %
          (tot['leaf_items'] / nrays, tot['leaf_rec'] / nrays, tot['leaf_rec'] / max(1, tot['leaf_items']), tot['leaf_trips'] / tot['batches'], 100.0 * tot['leaf_items'] / (64 * tot['leaf_trips'])))
    print('  record loop: one record a trip %.1f %% of the test slots used; pairs: %.1f %%' %
          (100.0 * tot['leaf_rec'] / tot['leaf_rec_slots'], 100.0 * tot['leaf_rec'] / (2 * tot['leaf_pair_slots'])))
    print('  high-water marks: node stack %d, leaf stack %d items' % (tot['max_ns'], tot['max_ls']))


if __name__ == '__main__':
    main()
