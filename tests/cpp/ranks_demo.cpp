// ranks_demo — the "one process per GPU" form of the multi-GPU boundary, written only against the C-ABI
// (include/rsrt.h): N copies of this program, one per GPU of a node, render ONE frame between them.
//   ranks_demo <scene.toml> <w> <h> <spp> <bounces> <env_w> <env_h> <out.f32> <world> <rank> <id-file> [device]
// Rank 0 creates the RCCL unique id and leaves it in <id-file> (the hand-over channel is the host's business: here a
// file); every rank renders the tiles t % world == rank and one rsrt_comm_reduce brings the frame to rank 0, which
// writes the RGBA32F sums.  The frame must equal the single-GPU frame bit for bit (tests/test_multi_gpu.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "rsrt.h"
#include "rsrt_host.h"

#define CHECK(ctx, expr)                                                                        \
    do {                                                                                        \
        if ((expr) != RSRT_OK) { std::fprintf(stderr, "rank %u: %s: %s\n", rank, #expr, rsrt_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 12 && argc != 13) { std::fprintf(stderr, "usage: ranks_demo scene.toml w h spp bounces env_w env_h out.f32 world rank id-file [device]\n"); return 2; }
    const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]), spp = (uint32_t)std::atoi(argv[4]);
    const uint32_t bounces = (uint32_t)std::atoi(argv[5]), ew = (uint32_t)std::atoi(argv[6]), eh = (uint32_t)std::atoi(argv[7]);
    const uint32_t world = (uint32_t)std::atoi(argv[9]), rank = (uint32_t)std::atoi(argv[10]);
    const std::string id_file = argv[11];
    const int device = argc == 13 ? std::atoi(argv[12]) : (int)rank;

    char err[2048] = {0};
    rsrt_scene *scene = nullptr;
    if (rsrt_scene_load_toml(argv[1], &scene, err, sizeof err) != 0) { std::fprintf(stderr, "%s\n", err); return 1; }
    rsrt_scene_counts c;
    rsrt_scene_get_counts(scene, &c);
    rsrt_camera_desc cam_desc;
    rsrt_scene_get_camera(scene, &cam_desc);
    rsrt_camera cam;
    rsrt_camera_uniform(&cam_desc, &cam);
    std::vector<float> rgba((size_t)ew * eh * 4), rgb((size_t)ew * eh * 3);
    std::vector<rsrt_alias_entry> alias((size_t)ew * eh);
    if (rsrt_synth_environment(ew, eh, rgba.data()) != 0) return 1;
    for (size_t i = 0; i < (size_t)ew * eh; i++) for (int k = 0; k < 3; k++) rgb[3 * i + k] = rgba[4 * i + k];
    if (rsrt_alias_table_build(ew, eh, rgb.data(), alias.data(), nullptr) != 0) return 1;

    rsrt_context *ctx = nullptr;
    if (rsrt_context_create(device, &ctx) != RSRT_OK) { std::fprintf(stderr, "rank %u: %s\n", rank, rsrt_last_error(nullptr)); return 1; }
    CHECK(ctx, rsrt_upload_scene(ctx, rsrt_scene_materials(scene), c.n_materials, rsrt_scene_spheres(scene), c.n_spheres, rsrt_scene_planes(scene),
                                 c.n_planes, rsrt_scene_vertices(scene), c.n_vertices, rsrt_scene_normals(scene), c.n_normals,
                                 rsrt_scene_triangles(scene), c.n_triangles, rsrt_scene_primitives(scene), c.n_primitives,
                                 rsrt_scene_bvh_nodes(scene), c.n_bvh_nodes));
    CHECK(ctx, rsrt_upload_environment(ctx, 0, ew, eh, rgba.data(), alias.data()));

    // the id: rank 0 makes it, the others wait for the file
    rsrt_unique_id id;
    if (rank == 0) {
        if (rsrt_comm_unique_id(&id) != RSRT_OK) { std::fprintf(stderr, "rank 0: %s\n", rsrt_last_error(nullptr)); return 1; }
        const std::string tmp = id_file + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id.bytes, 1, sizeof id.bytes, f) != sizeof id.bytes) return 1;
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return 1;
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(id_file.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(id.bytes, 1, sizeof id.bytes, f) != sizeof id.bytes) { std::fprintf(stderr, "rank %u: no id file\n", rank); return 1; }
        std::fclose(f);
    }
    CHECK(ctx, rsrt_comm_init(ctx, rank, world, &id)); // collective; also sets the tile partition (rank, world)
    CHECK(ctx, rsrt_accumulator_resize(ctx, w, h));
    CHECK(ctx, rsrt_render(ctx, &cam, w, h, 0, spp, bounces, 0, 0, nullptr));
    CHECK(ctx, rsrt_comm_reduce(ctx, 0, nullptr, nullptr)); // x + 0 + ... + 0: exact
    rsrt_stats st;
    CHECK(ctx, rsrt_get_stats(ctx, &st));
    std::printf("rank %u of %u: paths %llu rays %llu trace_ms %.3f reduce_ms %.3f\n", rank, world, (unsigned long long)st.paths,
                (unsigned long long)(st.ext_rays + st.shadow_rays), st.trace_kernel_ms, st.reduce_ms);
    if (rank == 0) {
        std::vector<float> sums((size_t)w * h * 4);
        CHECK(ctx, rsrt_accumulator_download(ctx, sums.data(), sums.size()));
        FILE *f = std::fopen(argv[8], "wb");
        if (!f || std::fwrite(sums.data(), sizeof(float), sums.size(), f) != sums.size()) return 1;
        std::fclose(f);
    }
    rsrt_context_destroy(ctx);
    rsrt_scene_free(scene);
    return 0;
}
