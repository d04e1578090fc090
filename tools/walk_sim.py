"""CPU count model of candidate BVH walks (no GPU): for a sample of rays of a scene, how many box steps, cluster visits,
leaf-box tests and primitive tests each design needs.  Rays: the scene's camera rays + secondary rays from their hit
points (cosine-ish random directions) — close enough to the integrator's mix to compare designs by counts.
    python tools/walk_sim.py [scene|grid] [max_leaves_per_cluster]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import rsoderh_raytracing_amd as R

def load(name):
    if name == 'grid':
        import make_big_scene
        return R.Scene.load_toml(make_big_scene.make(4))
    import util
    return R.Scene.load_toml(util.scene_path(name))

def slab(bmin, bmax, o, inv):
    a = (bmin - o) * inv; b = (bmax - o) * inv
    t0 = np.maximum(np.minimum(a, b).max(axis=1), 0.0); t1 = np.maximum(a, b).min(axis=1)
    return t0 <= t1

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'grid'
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sc = load(name)
    nodes = sc.bvh_nodes
    n = len(nodes)
    bmin = np.stack([nodes['bounds_min'][:, k] for k in range(3)], 1).astype(np.float32) if nodes['bounds_min'].ndim == 2 else None
    bmin = np.asarray(nodes['bounds_min'], np.float32).reshape(n, -1)[:, :3]
    bmax = np.asarray(nodes['bounds_max'], np.float32).reshape(n, -1)[:, :3]
    idx = nodes['primitives_or_second_child_index'].astype(np.int64); ln = nodes['primitives_len'].astype(np.int64)
    leaf = ln > 0
    # subtree leaf counts
    nleaves = np.zeros(n, np.int64); nprims = np.zeros(n, np.int64)
    for i in range(n - 1, -1, -1):
        if leaf[i]: nleaves[i] = 1; nprims[i] = ln[i]
        else: nleaves[i] = nleaves[i + 1] + nleaves[idx[i]]; nprims[i] = nprims[i + 1] + nprims[idx[i]]
    parent = np.full(n, -1, np.int64)
    for i in range(n):
        if not leaf[i]: parent[i + 1] = i; parent[idx[i]] = i
    # rays
    rng = np.random.default_rng(1)
    cam = sc.camera_uniform()
    W, H = 160, 90
    ys, xs = np.mgrid[0:H, 0:W]
    fx = xs.ravel() + rng.random(W * H) - 0.5; fy = ys.ravel() + rng.random(W * H) - 0.5
    m = np.sin(float(cam['fov_y'][0]) / 2)
    rcs = np.stack([((fx / W) * 2 - 1) * m * W / H, -((fy / H) * 2 - 1) * m, -np.ones(W * H)], 1)
    rot = np.asarray(cam['rot_transform'][0], np.float64).reshape(3, -1)[:, :3]  # columns
    d = rcs @ rot; d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.asarray(cam['pos'][0], np.float64)[:3], (W * H, 1))
    # secondary rays: start on random triangle surfaces-ish (leaf box centres), random directions
    lc = (bmin[leaf] + bmax[leaf]) / 2
    k = rng.integers(0, len(lc), 3 * W * H)
    o2 = lc[k] + rng.normal(0, 0.02, (len(k), 3)); d2 = rng.normal(size=(len(k), 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o = np.concatenate([o, o2]).astype(np.float32); d = np.concatenate([d, d2]).astype(np.float32)
    inv = (1.0 / d).astype(np.float32)
    nr = len(o)
    global RAYS
    RAYS = (o, inv)
    V = np.zeros(n, np.int64); Hh = np.zeros(n, np.int64)
    stack = [(0, np.arange(nr))]
    hit_leaf_sets = {}
    while stack:
        i, rays = stack.pop()
        V[i] = len(rays)
        if len(rays) == 0: continue
        h = slab(bmin[i], bmax[i], o[rays], inv[rays])
        hr = rays[h]; Hh[i] = len(hr)
        if not leaf[i]:
            stack.append((i + 1, hr)); stack.append((idx[i], hr))
    print('%s: %d nodes, %d leaves, %d prims; %d rays' % (name, n, leaf.sum(), nprims[0], nr))
    print('binary walk: %.1f box steps/ray, %.2f leaves hit/ray, %.2f prim tests/ray' % (V.sum() / nr, Hh[leaf].sum() / nr, (Hh[leaf] * ln[leaf]).sum() / nr))
    for K in ([K] if len(sys.argv) > 2 else [4, 8, 16, 32]):
        for maxp in (64,):
            # cluster roots: maximal subtrees with <= K leaves and <= maxp prims
            is_root = np.zeros(n, bool)
            for i in range(n):
                ok = nleaves[i] <= K and nprims[i] <= maxp
                pok = parent[i] >= 0 and nleaves[parent[i]] <= K and nprims[parent[i]] <= maxp
                is_root[i] = ok and not pok
            roots = np.nonzero(is_root)[0]
            inside = np.zeros(n, bool)  # strictly inside a cluster
            for r in roots:
                end = r + 1
                # subtree range in pre-order: [r, r + size)
            size = np.zeros(n, np.int64)
            for i in range(n - 1, -1, -1):
                size[i] = 1 if leaf[i] else 1 + size[i + 1] + size[idx[i]]
            for r in roots: inside[r + 1:r + size[r]] = True
            upper = ~inside  # upper interior nodes + cluster roots
            up_steps = V[upper].sum() / nr
            cl_visits = Hh[roots].sum() / nr
            leafbox = (Hh[roots] * nleaves[roots]).sum() / nr
            n_upper = upper.sum()
            print('K=%2d maxp=%d: %4d clusters (%.1f leaves avg), %5d upper elements (%.0f KB @32B, %.0f KB @16B) | per ray: %.1f upper steps, %.2f cluster visits, %.1f leaf-box tests (= %.0f B), binary-inside steps replaced %.1f'
                  % (K, maxp, len(roots), nleaves[roots].mean(), n_upper, n_upper * 32 / 1024, n_upper * 16 / 1024, up_steps, cl_visits, leafbox, leafbox * 32, V[inside].sum() / nr))



def wide(name, widths=(2, 4, 8)):
    """Wide-node collapse of the binary BVH (children = greedy expansion of the largest-area interior child until `w`
    children): node visits (= dependent round trips) and child-box tests per ray; share of visits served by the first
    N nodes in breadth-first order (an LDS top block)."""
    sc = load(name)
    nodes = sc.bvh_nodes; n = len(nodes)
    bmin = np.asarray(nodes['bounds_min'], np.float32).reshape(n, -1)[:, :3]; bmax = np.asarray(nodes['bounds_max'], np.float32).reshape(n, -1)[:, :3]
    idx = nodes['primitives_or_second_child_index'].astype(np.int64); ln = nodes['primitives_len'].astype(np.int64); leaf = ln > 0
    ext = (bmax - bmin).astype(np.float64); area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    o, inv = RAYS
    nr = len(o)
    # per binary node: which rays hit its box (as index arrays), by the nested-box property = rays that reach & hit it
    hits = {}
    stack = [(0, np.arange(nr))]
    while stack:
        i, rays = stack.pop()
        h = slab(bmin[i], bmax[i], o[rays], inv[rays]); hr = rays[h]; hits[i] = hr
        if not leaf[i] and len(hr): stack.append((i + 1, hr)); stack.append((idx[i], hr))
        elif not leaf[i]: hits[i + 1] = hits[idx[i]] = np.zeros(0, np.int64)
    def nhit(i): return len(hits.get(i, ()))
    for w in widths:
        wnodes = []  # (binary root, children list) in BFS order
        queue = [0]
        while queue:
            nxt = []
            for r in queue:
                ch = [r + 1, idx[r]]
                while len(ch) < w:
                    cand = [c for c in ch if not leaf[c]]
                    if not cand: break
                    c = max(cand, key=lambda c: area[c]); k = ch.index(c); ch[k:k + 1] = [c + 1, idx[c]]
                wnodes.append((r, ch)); nxt += [c for c in ch if not leaf[c]]
            queue = nxt
        visits = np.array([nhit(r) for r, ch in wnodes]); nch = np.array([len(ch) for r, ch in wnodes])
        leafhits = sum(nhit(c) for r, ch in wnodes for c in ch if leaf[c])
        tot_v = visits.sum()
        depth = 0; lvl = {0: 0}
        for r, ch in wnodes:
            for c in ch:
                if not leaf[c]: lvl[c] = lvl[r] + 1; depth = max(depth, lvl[c])
        cum = np.cumsum(visits) / tot_v
        msg = ', '.join('first %d nodes: %.0f%%' % (k, 100 * cum[min(k, len(cum)) - 1]) for k in (64, 160, 300, 560))
        print('w=%d: %4d wide nodes (%.1f children avg), depth %d | per ray: %.1f node visits, %.1f child-box tests, %.2f leaf hits | %s'
              % (w, len(wnodes), nch.mean(), depth + 1, tot_v / nr, (visits * nch).sum() / nr, leafhits / nr, msg))

RAYS = None
main()
wide(sys.argv[1] if len(sys.argv) > 1 else 'grid')
