// rsrt_state.hpp — header-only C++17 mirror of the reference's `State` (src/state.rs:29-834) over the
// C-ABI of rsrt.h / rsrt_host.h.  What `State::new / resize / update / render` do with wgpu, this does
// with librsrt: same progressive semantics (scene-hash reset, sample_count += 1 per frame,
// src/state.rs:775-794), plus the batched render the C-ABI adds.  Errors become rsrt::Error (the
// reference's anyhow::Error / unwrap at init).
#pragma once
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rsrt.h"
#include "rsrt_host.h"

namespace rsrt {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Scene::load_toml (src/scene.rs:235) + the uploads State::new derives from it
class Scene {
public:
    explicit Scene(const std::string &toml_path)
    {
        char err[2048] = {0};
        if (rsrt_scene_load_toml(toml_path.c_str(), &s_, err, sizeof err) != 0) throw Error(err);
        rsrt_scene_get_counts(s_, &counts_);
        rsrt_scene_get_camera(s_, &camera_);
    }
    ~Scene() { rsrt_scene_free(s_); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    const rsrt_scene *handle() const { return s_; }
    const rsrt_scene_counts &counts() const { return counts_; }
    const rsrt_camera_desc &camera() const { return camera_; }

private:
    rsrt_scene *s_ = nullptr;
    rsrt_scene_counts counts_{};
    rsrt_camera_desc camera_{};
};

// one HDRI + its alias table (EnvironmentMaps::new, src/environments.rs:19-64)
struct Environment {
    uint32_t width = 0, height = 0;
    std::vector<float> rgba;
    std::vector<rsrt_alias_entry> alias;

    static Environment synthetic(uint32_t w, uint32_t h)
    {
        Environment e;
        e.width = w; e.height = h;
        e.rgba.resize((size_t)w * h * 4);
        if (rsrt_synth_environment(w, h, e.rgba.data()) != 0) throw Error("rsrt_synth_environment failed");
        e.build_alias();
        return e;
    }
    static Environment from_hdr(const std::string &path)
    {
        Environment e;
        float *rgb = nullptr;
        char err[512] = {0};
        if (rsrt_load_hdr(path.c_str(), &e.width, &e.height, &rgb, err, sizeof err) != 0) throw Error(err);
        e.rgba.assign((size_t)e.width * e.height * 4, 0.0f);
        for (size_t i = 0; i < (size_t)e.width * e.height; i++) std::memcpy(&e.rgba[4 * i], &rgb[3 * i], 12);
        rsrt_free(rgb);
        e.build_alias();
        return e;
    }

private:
    void build_alias()
    {
        std::vector<float> rgb((size_t)width * height * 3);
        for (size_t i = 0; i < (size_t)width * height; i++) std::memcpy(&rgb[3 * i], &rgba[4 * i], 12);
        alias.resize((size_t)width * height);
        if (rsrt_alias_table_build(width, height, rgb.data(), alias.data(), nullptr) != 0) throw Error("rsrt_alias_table_build failed");
    }
};

class State {
public:
    // State::new (src/state.rs:60-649) over a LIST of devices of one node (rsrt_multi_*): device i renders the tiles
    // t % n == i, the frame is reduced onto devices[0] by RCCL inside librsrt whenever it is asked for.  {0} = one GPU.
    State(const Scene &scene, const std::vector<const Environment *> &environments, uint32_t width, uint32_t height,
          const std::vector<int> &devices = {0})
    {
        if (rsrt_multi_create(devices.data(), (uint32_t)devices.size(), &m_) != RSRT_OK) throw Error(rsrt_multi_last_error(nullptr));
        try {
            const rsrt_scene *s = scene.handle();
            const rsrt_scene_counts &c = scene.counts();
            check(rsrt_multi_upload_scene(m_, rsrt_scene_materials(s), c.n_materials, rsrt_scene_spheres(s), c.n_spheres, rsrt_scene_planes(s),
                                          c.n_planes, rsrt_scene_vertices(s), c.n_vertices, rsrt_scene_normals(s), c.n_normals,
                                          rsrt_scene_triangles(s), c.n_triangles, rsrt_scene_primitives(s), c.n_primitives, rsrt_scene_bvh_nodes(s),
                                          c.n_bvh_nodes));
            for (size_t i = 0; i < environments.size(); i++)
                check(rsrt_multi_upload_environment(m_, (uint32_t)i, environments[i]->width, environments[i]->height, environments[i]->rgba.data(),
                                                    environments[i]->alias.data()));
            camera_ = scene.camera();
            resize(width, height);
        } catch (...) {
            rsrt_multi_destroy(m_);
            throw;
        }
    }
    ~State() { rsrt_multi_destroy(m_); }
    State(const State &) = delete;
    State &operator=(const State &) = delete;

    uint32_t max_bounces = 10;      // MAX_BOUNCES (shader.wgsl:232)
    uint32_t environment_index = 0; // src/state.rs:638
    uint32_t flags = 0;

    // State::resize (src/state.rs:651-666)
    void resize(uint32_t width, uint32_t height)
    {
        check(rsrt_multi_resize(m_, width, height));
        width_ = width; height_ = height;
        have_hash_ = false;
    }
    // State::update (src/state.rs:722-758): the controller's result
    void update(const rsrt_camera_desc &camera) { camera_ = camera; }
    const rsrt_camera_desc &camera() const { return camera_; }
    uint32_t sample_count() const { return sample_count_; }
    uint32_t device_count() const { return rsrt_multi_size(m_); }

    // State::render (src/state.rs:760-833): one more sample per pixel; restart when the scene hash changed
    void render() { render_samples(1); }
    void render_samples(uint32_t n)
    {
        const size_t h = scene_hash();
        if (!have_hash_ || h != last_hash_) { // src/state.rs:778-786
            last_hash_ = h; have_hash_ = true;
            check(rsrt_multi_clear(m_));
            sample_count_ = 0;
        }
        rsrt_camera cam;
        rsrt_camera_uniform(&camera_, &cam);
        check(rsrt_multi_render(m_, &cam, width_, height_, sample_count_, n, max_bounces, environment_index, flags));
        sample_count_ += n;
    }
    std::vector<float> download() // cumulative_light_texture, all devices' tiles
    {
        std::vector<float> out((size_t)width_ * height_ * 4);
        check(rsrt_multi_download(m_, out.data(), out.size()));
        return out;
    }
    std::vector<uint8_t> display() // what the window shows (hdr.wgsl + sRGB surface)
    {
        std::vector<uint8_t> out((size_t)width_ * height_ * 4);
        check(rsrt_multi_display_srgb8(m_, sample_count_, out.data(), out.size()));
        return out;
    }
    // scene.dev_index 2 / 3 (shader.wgsl:1314-1338): out_texture (RGBA binary16) as `main` leaves it for the developer views — 3: the HDRI
    // under the frame; 2: twenty draws of the alias table per pixel added onto `out_texture` (the previous frame's; zeros when empty)
    std::vector<uint16_t> debug_view(uint32_t dev_index, std::vector<uint16_t> out_texture = {})
    {
        out_texture.resize((size_t)width_ * height_ * 4, 0);
        const rsrt_status st = rsrt_debug_view_f16(context(0), dev_index, environment_index, sample_count_, out_texture.data(), out_texture.size());
        if (st != RSRT_OK) throw Error(rsrt_last_error(context(0)));
        return out_texture;
    }
    rsrt_stats stats()
    {
        rsrt_stats s;
        check(rsrt_multi_get_stats(m_, &s));
        return s;
    }
    rsrt_context *context(uint32_t i = 0) { return rsrt_multi_context(m_, i); }

private:
    void check(rsrt_status st)
    {
        if (st != RSRT_OK) throw Error(rsrt_multi_last_error(m_));
    }
    size_t scene_hash() const // SceneState: camera bits + environment index (src/scene.rs:255-262, src/camera.rs:92-100)
    {
        std::string bytes(reinterpret_cast<const char *>(&camera_), sizeof camera_);
        bytes.append(reinterpret_cast<const char *>(&environment_index), sizeof environment_index);
        return std::hash<std::string>()(bytes);
    }
    rsrt_multi *m_ = nullptr;
    rsrt_camera_desc camera_{};
    uint32_t width_ = 0, height_ = 0, sample_count_ = 0;
    size_t last_hash_ = 0;
    bool have_hash_ = false;
};

} // namespace rsrt
