// bvh_build.cpp — binned-SAH BVH builder producing the reference's flattened node array.
//
// Restates build_bvh (reference src/bvh.rs:13-337): primitive order spheres, planes, triangles
// (:40-72); leaf when <= 5 primitives or the centroid box is flat on its longest axis (:227-244);
// 12 centroid buckets on that axis (:258-276); cost_i = 0.125 + (n0*SA0 + n1*SA1)/SA (:279-292),
// first minimum wins (:294-300); unstable two-pointer partition (:304-315); pre-order layout,
// first child = parent + 1, second child index stored in the parent (:155-178).
//
// Unlike the reference (recursive build tree, then a flatten pass) this builder emits the
// flattened nodes directly from an explicit work stack; bucket bounds are swept as prefix /
// suffix unions.  min/max are exact, so both changes leave every f32 identical.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../../include/rsrt_host.h"
#include "host_math.h"

namespace rsrt_host {

namespace {

constexpr size_t kMaxLeaf = 5;
constexpr size_t kBuckets = 12;

struct Item {
    uint32_t type, index;
    Bounds3 box;
    Vec3 centroid;
};

struct Task {
    size_t begin, end;
    int64_t patch_parent; // node whose second-child index is this task's node, or -1
    uint32_t depth;
};

inline size_t bucket_of(float c, float lo, float hi)
{
    float f = (float)kBuckets * ((c - lo) / (hi - lo));
    size_t b = f > 0.0f ? (size_t)f : 0; // Rust `as usize`: saturating, NaN -> 0
    return b == kBuckets ? kBuckets - 1 : b;
}

void put_bounds(rsrt_bvh_node &n, const Bounds3 &b)
{
    store(n.bounds_min, b.min);
    store(n.bounds_max, b.max);
}

} // namespace

int build_bvh(std::vector<Item> &items, std::vector<rsrt_primitive_info> &prims_out, std::vector<rsrt_bvh_node> &nodes,
              uint32_t &depth_out)
{
    if (items.empty()) return 1;
    prims_out.clear();
    nodes.clear();
    depth_out = 0;
    std::vector<Task> stack;
    stack.push_back({0, items.size(), -1, 0});
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        const size_t n = t.end - t.begin;
        Item *p = items.data() + t.begin;
        const uint32_t self = (uint32_t)nodes.size();
        if (t.patch_parent >= 0) nodes[(size_t)t.patch_parent].primitives_or_second_child_index = self;
        depth_out = std::max(depth_out, t.depth);

        Bounds3 box = Bounds3::identity();
        for (size_t i = 0; i < n; i++) box.grow(p[i].box);
        rsrt_bvh_node node;
        std::memset(&node, 0, sizeof node);
        put_bounds(node, box);

        bool leaf = n <= kMaxLeaf;
        unsigned axis = 0;
        float lo = 0, hi = 0;
        if (!leaf) {
            Bounds3 cbox = Bounds3::identity();
            for (size_t i = 0; i < n; i++) cbox.grow(p[i].centroid);
            axis = cbox.max_axis();
            lo = cbox.min[axis];
            hi = cbox.max[axis];
            leaf = (lo == hi);
        }
        if (leaf) {
            node.primitives_or_second_child_index = (uint32_t)prims_out.size();
            node.primitives_len = (uint32_t)n;
            node.split_axis = 0;
            for (size_t i = 0; i < n; i++) prims_out.push_back({p[i].type, p[i].index});
            nodes.push_back(node);
            continue;
        }

        size_t count[kBuckets] = {0};
        Bounds3 bbox[kBuckets];
        for (auto &b : bbox) b = Bounds3::identity();
        for (size_t i = 0; i < n; i++) {
            size_t b = bucket_of(p[i].centroid[axis], lo, hi);
            count[b]++;
            bbox[b].grow(p[i].box);
        }
        // prefix (buckets 0..=i) and suffix (i+1..) unions / counts
        Bounds3 pre[kBuckets], suf[kBuckets];
        size_t pre_n[kBuckets], suf_n[kBuckets];
        Bounds3 acc = Bounds3::identity();
        size_t acc_n = 0;
        for (size_t i = 0; i < kBuckets; i++) { acc.grow(bbox[i]); acc_n += count[i]; pre[i] = acc; pre_n[i] = acc_n; }
        acc = Bounds3::identity();
        acc_n = 0;
        for (size_t i = kBuckets; i-- > 0;) { acc.grow(bbox[i]); acc_n += count[i]; suf[i] = acc; suf_n[i] = acc_n; }
        const float total_area = box.surface_area();
        size_t best = 0;
        float best_cost = 0;
        for (size_t i = 0; i + 1 < kBuckets; i++) {
            float cost = 0.125f + ((float)pre_n[i] * pre[i].surface_area() + (float)suf_n[i + 1] * suf[i + 1].surface_area()) / total_area;
            if (i == 0 || cost < best_cost) { best = i; best_cost = cost; }
        }
        // in-place two-pointer partition (not stable — leaf order depends on it)
        size_t split = 0, end = n;
        while (split < end) {
            if (bucket_of(p[split].centroid[axis], lo, hi) <= best) split++;
            else std::swap(p[split], p[--end]);
        }
        if (split == 0 || split == n) { // reference's median fallback (:317-326); unreachable in practice
            split = n / 2;
            std::nth_element(p, p + split, p + n, [axis](const Item &a, const Item &b) { return a.centroid[axis] < b.centroid[axis]; });
        }
        node.primitives_len = 0;
        node.split_axis = axis;
        nodes.push_back(node);
        // right first so the left child is popped next and lands at self + 1
        stack.push_back({t.begin + split, t.end, (int64_t)self, t.depth + 1});
        stack.push_back({t.begin, t.begin + split, -1, t.depth + 1});
    }
    return 0;
}

int build_bvh_from_arrays(const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                          const rsrt_vec3 *vertices, uint32_t n_vertices, const rsrt_triangle *triangles, uint32_t n_triangles,
                          std::vector<rsrt_primitive_info> &prims_out, std::vector<rsrt_bvh_node> &nodes, uint32_t &depth)
{
    std::vector<Item> items;
    items.reserve((size_t)n_spheres + n_planes + n_triangles);
    for (uint32_t i = 0; i < n_spheres; i++) { // Sphere::bounds, scene.rs:173-180
        Vec3 c = vec3(spheres[i].pos);
        float r = spheres[i].radius;
        Bounds3 b{c - vec3(r, r, r), c + vec3(r, r, r)};
        items.push_back({0, i, b, b.center()});
    }
    for (uint32_t i = 0; i < n_planes; i++) { // Plane::bounds, scene.rs:203-207
        Vec3 a = vec3(planes[i].pos);
        Vec3 far_corner = a + vec3(planes[i].forward) + vec3(planes[i].right);
        Bounds3 b = Bounds3::identity();
        b.grow(a);
        b.grow(far_corner);
        items.push_back({1, i, b, b.center()});
    }
    for (uint32_t i = 0; i < n_triangles; i++) { // HittableTriangle::bounds, mesh.rs:143-147
        const rsrt_triangle &t = triangles[i];
        if (t.vertex_0 >= n_vertices || t.vertex_1 >= n_vertices || t.vertex_2 >= n_vertices) return 2;
        Bounds3 b = Bounds3::identity();
        b.grow(vec3(vertices[t.vertex_0].v));
        b.grow(vec3(vertices[t.vertex_1].v));
        b.grow(vec3(vertices[t.vertex_2].v));
        items.push_back({2, i, b, b.center()});
    }
    return build_bvh(items, prims_out, nodes, depth);
}

} // namespace rsrt_host

extern "C" int rsrt_build_bvh(const rsrt_sphere *spheres, uint32_t n_spheres, const rsrt_plane_desc *planes, uint32_t n_planes,
                              const rsrt_vec3 *vertices, uint32_t n_vertices, const rsrt_triangle *triangles,
                              uint32_t n_triangles, rsrt_primitive_info *primitives_out, rsrt_bvh_node *nodes_out,
                              uint32_t *n_nodes_out, uint32_t *depth_out)
{
    std::vector<rsrt_primitive_info> prims;
    std::vector<rsrt_bvh_node> nodes;
    uint32_t depth = 0;
    int rc = rsrt_host::build_bvh_from_arrays(spheres, n_spheres, planes, n_planes, vertices, n_vertices, triangles, n_triangles,
                                              prims, nodes, depth);
    if (rc) return rc;
    std::memcpy(primitives_out, prims.data(), prims.size() * sizeof(rsrt_primitive_info));
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rsrt_bvh_node));
    if (n_nodes_out) *n_nodes_out = (uint32_t)nodes.size();
    if (depth_out) *depth_out = depth;
    return 0;
}
