// state_demo — a host written only against the C/C++ boundary (include/rsrt_state.hpp): loads a scene
// TOML, renders progressively like the reference (one sample per frame), then a batch, and writes the
// RGBA32F sums + the display PNG.  tests/test_cpp_host.py compares the sums with the oracle.
//   state_demo <scene.toml> <w> <h> <frames> <batch> <bounces> <env_w> <env_h> <out.f32> <out.png> [devices, e.g. 0,1,2,3]
// With a device list the same `State` runs over several GPUs of the node (rsrt_multi_*: tiles interleaved over the
// devices, RCCL reduce inside the library); the output must not change by a bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rsrt_state.hpp"

int main(int argc, char **argv)
{
    if (argc != 11 && argc != 12) { std::fprintf(stderr, "usage: state_demo scene.toml w h frames batch bounces env_w env_h out.f32 out.png [devices]\n"); return 2; }
    try {
        rsrt::Scene scene(argv[1]);
        const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]);
        const uint32_t frames = (uint32_t)std::atoi(argv[4]), batch = (uint32_t)std::atoi(argv[5]);
        rsrt::Environment env = rsrt::Environment::synthetic((uint32_t)std::atoi(argv[7]), (uint32_t)std::atoi(argv[8]));
        std::vector<int> devices;
        if (argc == 12)
            for (const char *p = argv[11]; *p;) { devices.push_back(std::atoi(p)); p = std::strchr(p, ','); if (!p) break; p++; }
        if (devices.empty()) devices.push_back(0);
        rsrt::State state(scene, {&env}, w, h, devices);
        state.max_bounces = (uint32_t)std::atoi(argv[6]);
        for (uint32_t i = 0; i < frames; i++) state.render(); // the reference's frame loop
        if (batch) state.render_samples(batch);
        std::vector<float> sums = state.download();
        FILE *f = std::fopen(argv[9], "wb");
        if (!f || std::fwrite(sums.data(), sizeof(float), sums.size(), f) != sums.size()) { std::fprintf(stderr, "cannot write %s\n", argv[9]); return 1; }
        std::fclose(f);
        std::vector<uint8_t> img = state.display();
        if (rsrt_write_png(argv[10], w, h, img.data()) != 0) { std::fprintf(stderr, "cannot write %s\n", argv[10]); return 1; }
        rsrt_stats st = state.stats();
        std::printf("samples %u paths %llu rays %llu kernel_ms %.3f devices %u reduce_ms %.3f\n", state.sample_count(), (unsigned long long)st.paths,
                    (unsigned long long)(st.ext_rays + st.shadow_rays), st.kernel_ms, state.device_count(), st.reduce_ms);
        // a camera change restarts accumulation (scene hash)
        rsrt_camera_desc cam = state.camera();
        cam.yaw += 0.1f;
        state.update(cam);
        state.render();
        if (state.sample_count() != 1) { std::fprintf(stderr, "scene-hash reset failed\n"); return 1; }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
