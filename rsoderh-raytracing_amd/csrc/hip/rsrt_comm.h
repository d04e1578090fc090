// rsrt_comm.h — multi-GPU behind the C-ABI (included at the end of rsrt_api.hip; see include/rsrt.h, "multi-GPU").
//
// The path shards by pixel: tile t of the frame belongs to rank t % world, scene and environment are
// replicated, and ONE RCCL reduce(sum, f32) of the W*H*4 accumulators per frame brings the image to the root.  Every
// pixel has exactly one non-zero contributor, so x + 0 + ... + 0 is exact and the N-GPU image equals the 1-GPU image
// bit for bit whatever order RCCL adds in.  There is no other exchange step, hence no other collective.
//
// Two forms, one implementation:
//   rsrt_comm_*   one process (or thread) per GPU: rank 0 makes a 128-byte id (rsrt_comm_unique_id), hands it to the
//                 other ranks by whatever channel the host has, every rank calls rsrt_comm_init on its own context.
//   rsrt_multi_*  one single-threaded caller, a LIST of devices — what the reference's `State` (src/state.rs:60-98,
//                 :760-833) would bind: one rsrt_context per device inside, ncclCommInitAll, grouped ncclReduce.
//
// RCCL is loaded lazily with dlopen (librccl.so.1): a single-GPU user never needs it, and a process that already
// has an RCCL (torch's) gets that same copy instead of a second one.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error; // why it could not be loaded
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *names[] = {getenv("RSRT_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
            api.error = dlerror();
        }
        if (!h) return;
#define RSRT_SYM(name)                                                                     \
    api.name = reinterpret_cast<decltype(api.name)>(dlsym(h, "nccl" #name));               \
    if (!api.name) { api.error = "librccl: symbol nccl" #name " not found"; return; }
        RSRT_SYM(GetUniqueId) RSRT_SYM(CommInitRank) RSRT_SYM(CommInitAll) RSRT_SYM(CommDestroy) RSRT_SYM(Reduce)
        RSRT_SYM(GroupStart) RSRT_SYM(GroupEnd) RSRT_SYM(GetErrorString) RSRT_SYM(GetVersion)
#undef RSRT_SYM
        api.ok = true;
    });
    return api;
}

#define RCCL_TRY(ctx, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(ctx, RSRT_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

// rank that renders pixel (x, y): the arithmetic rsrt_render's kernels use (tile t = ty * tiles_x + tx -> t % world)
inline uint32_t tile_owner(uint32_t width, uint32_t tile_w, uint32_t tile_h, uint32_t world, uint32_t x, uint32_t y)
{
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w;
    return ((y / tile_h) * tiles_x + x / tile_w) % world;
}

bool partition_args_ok(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world)
{
    return width && height && world && tile_w && tile_h && tile_w * tile_h <= 4096 && (tile_w * tile_h) % RT_WAVE == 0;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ partition arithmetic (pure host code, no GPU needed)
uint32_t rsrt_partition_owner(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world_size, uint32_t x, uint32_t y)
{
    if (!partition_args_ok(width, height, tile_w, tile_h, world_size) || x >= width || y >= height) return 0xffffffffu;
    return tile_owner(width, tile_w, tile_h, world_size, x, y);
}

rsrt_status rsrt_partition_mask(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world_size,
                                uint8_t *mask, uint64_t *owned_pixels)
{
    if (!partition_args_ok(width, height, tile_w, tile_h, world_size) || rank >= world_size) return RSRT_ERR_INVALID_ARGUMENT;
    uint64_t n = 0;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const bool mine = tile_owner(width, tile_w, tile_h, world_size, x, y) == rank;
            n += mine;
            if (mask) mask[(size_t)y * width + x] = mine;
        }
    if (owned_pixels) *owned_pixels = n;
    return RSRT_OK;
}

// ------------------------------------------------------------------ one process per GPU
rsrt_status rsrt_comm_unique_id(rsrt_unique_id *out)
{
    if (!out) return fail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "out is NULL");
    static_assert(sizeof(rsrt_unique_id) == sizeof(ncclUniqueId), "rsrt_unique_id carries an ncclUniqueId");
    if (!rccl().ok) return fail(nullptr, RSRT_ERR_COMM, "RCCL is not available: %s", rccl().error.c_str());
    ncclUniqueId id;
    RCCL_TRY(nullptr, rccl().GetUniqueId(&id));
    memcpy(out->bytes, &id, sizeof id);
    return RSRT_OK;
}

rsrt_status rsrt_comm_destroy(rsrt_context *ctx)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (ctx->comm) {
        (void)sync_all(ctx);
        if (ctx->comm_owned) (void)rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    return RSRT_OK;
}

rsrt_status rsrt_comm_init(rsrt_context *ctx, uint32_t rank, uint32_t world_size, const rsrt_unique_id *id)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!id || world_size == 0 || rank >= world_size) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "comm_init: rank %u not in [0,%u) or id NULL", rank, world_size);
    if (!rccl().ok) return fail(ctx, RSRT_ERR_COMM, "RCCL is not available: %s", rccl().error.c_str());
    (void)rsrt_comm_destroy(ctx);
    ncclUniqueId nid;
    memcpy(&nid, id->bytes, sizeof nid);
    ncclComm_t comm = nullptr;
    RCCL_TRY(ctx, rccl().CommInitRank(&comm, (int)world_size, nid, (int)rank));
    ctx->comm = comm;
    ctx->comm_owned = true;
    ctx->comm_rank = rank;
    ctx->comm_world = world_size;
    return rsrt_set_partition(ctx, rank, world_size, ctx->tile_w, ctx->tile_h); // this rank renders tiles t % world == rank
}

// One ncclReduce(sum, f32) of the W*H*4 accumulator onto `root`.  recv_device_rgba32f (root only; ignored elsewhere):
// where the full frame goes — NULL = in place, the root's own accumulator becomes the full frame (fine when it is
// cleared before the next render, as a batch renderer does; a progressive caller passes a separate buffer so that
// its accumulator keeps holding its own tiles only).
rsrt_status rsrt_comm_reduce(rsrt_context *ctx, uint32_t root, void *recv_device_rgba32f, void *hip_stream)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    if (root >= ctx->comm_world) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "comm_reduce: root %u not in [0,%u)", root, ctx->comm_world);
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    const size_t count = (size_t)ctx->acc_w * ctx->acc_h * 4;
    void *recv = (ctx->comm_rank == root && recv_device_rgba32f) ? recv_device_rgba32f : static_cast<void *>(ctx->accum);
    rsrt_status st = begin_work(ctx, stream);
    if (st) return st;
    hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
    hipError_t he = hipEventRecord(e0, stream);
    ncclResult_t nr = ncclSuccess;
    if (he == hipSuccess) {
        if (ctx->comm)
            nr = rccl().Reduce(ctx->accum, recv, count, ncclFloat, ncclSum, (int)root, static_cast<ncclComm_t>(ctx->comm), stream);
        else if (recv != ctx->accum) // world of one, no communicator: the frame is the accumulator
            he = hipMemcpyAsync(recv, ctx->accum, count * sizeof(float), hipMemcpyDeviceToDevice, stream);
    }
    if (he == hipSuccess && nr == ncclSuccess) he = hipEventRecord(e1, stream);
    if (he != hipSuccess || nr != ncclSuccess) { // nothing pending: the two events go back to the pool
        ctx->event_pool.push_back(e0);
        ctx->event_pool.push_back(e1);
        (void)end_work(ctx, stream);
        if (nr != ncclSuccess) return fail(ctx, RSRT_ERR_COMM, "ncclReduce failed: %s", rccl().GetErrorString(nr));
        return fail(ctx, RSRT_ERR_HIP, "comm_reduce: %s", hipGetErrorString(he));
    }
    ctx->pending_reduce.push_back({e0, e1});
    ctx->cum_reduces++;
    return end_work(ctx, stream);
}

} // extern "C"

// ------------------------------------------------------------------ one caller, a list of devices
struct rsrt_multi {
    std::vector<rsrt_context *> ctx;
    float4 *frame = nullptr; // on device ctx[0]: the reduced W*H RGBA32F sum
    float4 *stage = nullptr; // on device ctx[0]: landing buffer of the peer-copy reduce (no RCCL)
    uint32_t frame_w = 0, frame_h = 0;
    std::string error;
};

// frame += part (the reduce without RCCL; every pixel has one non-zero contributor, so the order does not matter)
__global__ void rt_add_frame_kernel(float4 *frame, const float4 *part, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = frame[i], b = part[i];
    frame[i] = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
}

namespace {

thread_local std::string g_multi_error;

rsrt_status mfail(rsrt_multi *m, rsrt_status st, const char *what, const char *msg)
{
    std::string s = std::string(what) + ": " + msg;
    if (m) m->error = s;
    else g_multi_error = s;
    return st;
}

// runs f on every context; the first failure is reported with its device
template <class F>
rsrt_status for_each_ctx(rsrt_multi *m, const char *what, F f)
{
    if (!m) return RSRT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        const rsrt_status st = f(m->ctx[i], (uint32_t)i);
        if (st != RSRT_OK) {
            char buf[64];
            snprintf(buf, sizeof buf, "%s (device %d)", what, m->ctx[i]->device);
            return mfail(m, st, buf, rsrt_last_error(m->ctx[i]));
        }
    }
    return RSRT_OK;
}

rsrt_status multi_ensure_frame(rsrt_multi *m)
{
    rsrt_context *c0 = m->ctx[0];
    if (m->frame && m->frame_w == c0->acc_w && m->frame_h == c0->acc_h) return RSRT_OK;
    DeviceGuard g(c0->device);
    if (m->frame) { (void)hipDeviceSynchronize(); (void)hipFree(m->frame); (void)hipFree(m->stage); m->frame = m->stage = nullptr; }
    if (hipMalloc(&m->frame, (size_t)c0->acc_w * c0->acc_h * sizeof(float4)) != hipSuccess) return mfail(m, RSRT_ERR_OUT_OF_MEMORY, "multi frame buffer", "hipMalloc failed");
    if (m->ctx.size() > 1 && !c0->comm && hipMalloc(&m->stage, (size_t)c0->acc_w * c0->acc_h * sizeof(float4)) != hipSuccess)
        return mfail(m, RSRT_ERR_OUT_OF_MEMORY, "multi stage buffer", "hipMalloc failed");
    m->frame_w = c0->acc_w;
    m->frame_h = c0->acc_h;
    return RSRT_OK;
}

// accumulators of all devices -> m->frame on device 0 (grouped: one launch per device, no deadlock with one caller thread)
rsrt_status multi_reduce(rsrt_multi *m)
{
    rsrt_status st = multi_ensure_frame(m);
    if (st) return st;
    if (m->ctx.size() > 1 && !m->ctx[0]->comm) { // no RCCL (not installed, or the list names one device twice): peer copies + adds on devices[0]
        rsrt_context *c0 = m->ctx[0];
        DeviceGuard g(c0->device);
        const size_t n = (size_t)c0->acc_w * c0->acc_h, bytes = n * sizeof(float4);
        hipStream_t q = c0->stream;
        if ((st = begin_work(c0, q))) return mfail(m, st, "multi reduce", rsrt_last_error(c0));
        hipError_t e = hipMemcpyAsync(m->frame, c0->accum, bytes, hipMemcpyDeviceToDevice, q);
        for (size_t i = 1; i < m->ctx.size() && e == hipSuccess; i++) {
            rsrt_context *ci = m->ctx[i];
            if (ci->last_valid) e = hipStreamWaitEvent(q, ci->last_event, 0); // after device i's renders
            if (e == hipSuccess) e = hipMemcpyPeerAsync(m->stage, c0->device, ci->accum, ci->device, bytes, q);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(rt_add_frame_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, m->frame, m->stage, n);
                e = hipGetLastError();
            }
        }
        if (e != hipSuccess) { (void)end_work(c0, q); return mfail(m, RSRT_ERR_HIP, "multi reduce (peer copies)", hipGetErrorString(e)); }
        if ((st = end_work(c0, q))) return mfail(m, st, "multi reduce", rsrt_last_error(c0));
        for (size_t i = 1; i < m->ctx.size(); i++) { // device i must not overwrite its accumulator before it has been copied
            DeviceGuard gi(m->ctx[i]->device);
            if (hipStreamWaitEvent(m->ctx[i]->stream, c0->last_event, 0) != hipSuccess) return mfail(m, RSRT_ERR_HIP, "multi reduce", "hipStreamWaitEvent failed");
        }
        return RSRT_OK;
    }
    const bool grouped = m->ctx[0]->comm != nullptr;
    if (grouped && rccl().GroupStart() != ncclSuccess) return mfail(m, RSRT_ERR_COMM, "multi reduce", "ncclGroupStart failed");
    rsrt_status first = RSRT_OK;
    for (rsrt_context *c : m->ctx) {
        st = rsrt_comm_reduce(c, 0, m->frame, nullptr);
        if (st && !first) { first = st; mfail(m, st, "multi reduce", rsrt_last_error(c)); }
    }
    if (grouped && rccl().GroupEnd() != ncclSuccess && !first) first = mfail(m, RSRT_ERR_COMM, "multi reduce", "ncclGroupEnd failed");
    return first;
}

} // namespace

extern "C" {

const char *rsrt_multi_last_error(const rsrt_multi *m) { return m ? m->error.c_str() : g_multi_error.c_str(); }
uint32_t rsrt_multi_size(const rsrt_multi *m) { return m ? (uint32_t)m->ctx.size() : 0u; }
rsrt_context *rsrt_multi_context(rsrt_multi *m, uint32_t i) { return (m && i < m->ctx.size()) ? m->ctx[i] : nullptr; }
int rsrt_multi_uses_rccl(const rsrt_multi *m) { return (m && !m->ctx.empty() && m->ctx[0]->comm) ? 1 : 0; }

void rsrt_multi_destroy(rsrt_multi *m)
{
    if (!m) return;
    for (rsrt_context *c : m->ctx) (void)rsrt_synchronize(c);
    if (m->frame && !m->ctx.empty()) { DeviceGuard g(m->ctx[0]->device); (void)hipFree(m->frame); (void)hipFree(m->stage); }
    for (rsrt_context *c : m->ctx) rsrt_context_destroy(c); // destroys its communicator too
    delete m;
}

rsrt_status rsrt_multi_create(const int *devices, uint32_t n_devices, rsrt_multi **out)
{
    if (!out) return mfail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "rsrt_multi_create", "out is NULL");
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return mfail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "rsrt_multi_create", "device list empty (or longer than 64)");
    // A device may appear twice only for rehearsals (RSRT_MULTI_ALLOW_SAME_DEVICE=1: N "devices" on one GPU; RCCL refuses
    // two ranks on one device, so the frame is then reduced by peer copies + adds, as it is when librccl is missing).
    bool duplicates = false;
    for (uint32_t i = 0; i < n_devices; i++)
        for (uint32_t j = 0; j < i; j++) duplicates = duplicates || devices[i] == devices[j];
    const char *allow = getenv("RSRT_MULTI_ALLOW_SAME_DEVICE");
    if (duplicates && !(allow && atoi(allow) != 0)) return mfail(nullptr, RSRT_ERR_INVALID_ARGUMENT, "rsrt_multi_create", "a device appears twice in the list");
    rsrt_multi *m = new rsrt_multi();
    for (uint32_t i = 0; i < n_devices; i++) {
        rsrt_context *c = nullptr;
        const rsrt_status st = rsrt_context_create(devices[i], &c);
        if (st) {
            mfail(nullptr, st, "rsrt_multi_create", rsrt_last_error(nullptr));
            rsrt_multi_destroy(m);
            return st;
        }
        m->ctx.push_back(c);
    }
    if (rccl().ok && !duplicates) { // (a list of one device gets a communicator too: the same calls run whatever the list length)
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t r = rccl().CommInitAll(comms.data(), (int)n_devices, devices);
        if (r != ncclSuccess) { mfail(nullptr, RSRT_ERR_COMM, "ncclCommInitAll", rccl().GetErrorString(r)); rsrt_multi_destroy(m); return RSRT_ERR_COMM; }
        for (uint32_t i = 0; i < n_devices; i++) {
            m->ctx[i]->comm = comms[i];
            m->ctx[i]->comm_owned = true;
            m->ctx[i]->comm_rank = i;
            m->ctx[i]->comm_world = n_devices;
        }
    }
    for (uint32_t i = 0; i < n_devices; i++) (void)rsrt_set_partition(m->ctx[i], i, n_devices, 16, 16);
    *out = m;
    return RSRT_OK;
}

rsrt_status rsrt_multi_upload_scene(rsrt_multi *m, const rsrt_material *materials, uint32_t n_materials, const rsrt_sphere *spheres,
                                    uint32_t n_spheres, const rsrt_plane *planes, uint32_t n_planes, const rsrt_vec3 *vertices,
                                    uint32_t n_vertices, const rsrt_vec3 *normals, uint32_t n_normals, const rsrt_triangle *triangles,
                                    uint32_t n_triangles, const rsrt_primitive_info *primitives, uint32_t n_primitives,
                                    const rsrt_bvh_node *nodes, uint32_t n_nodes)
{
    return for_each_ctx(m, "upload_scene", [&](rsrt_context *c, uint32_t) {
        return rsrt_upload_scene(c, materials, n_materials, spheres, n_spheres, planes, n_planes, vertices, n_vertices, normals, n_normals,
                                 triangles, n_triangles, primitives, n_primitives, nodes, n_nodes);
    });
}

rsrt_status rsrt_multi_upload_environment(rsrt_multi *m, uint32_t slot, uint32_t width, uint32_t height, const float *rgba,
                                          const rsrt_alias_entry *alias)
{
    return for_each_ctx(m, "upload_environment", [&](rsrt_context *c, uint32_t) { return rsrt_upload_environment(c, slot, width, height, rgba, alias); });
}

rsrt_status rsrt_multi_resize(rsrt_multi *m, uint32_t width, uint32_t height)
{
    return for_each_ctx(m, "accumulator_resize", [&](rsrt_context *c, uint32_t) { return rsrt_accumulator_resize(c, width, height); });
}

rsrt_status rsrt_multi_clear(rsrt_multi *m)
{
    return for_each_ctx(m, "accumulator_clear", [&](rsrt_context *c, uint32_t) { return rsrt_accumulator_clear(c); });
}

// State::render's compute pass over all devices: each renders the samples of ITS tiles into its own accumulator
// (asynchronous: the N kernels run side by side).  Nothing is exchanged until the frame is asked for.
rsrt_status rsrt_multi_render(rsrt_multi *m, const rsrt_camera *camera, uint32_t width, uint32_t height, uint32_t sample_begin,
                              uint32_t sample_count, uint32_t max_bounces, uint32_t environment_index, uint32_t flags)
{
    return for_each_ctx(m, "render", [&](rsrt_context *c, uint32_t) {
        return rsrt_render(c, camera, width, height, sample_begin, sample_count, max_bounces, environment_index, flags, nullptr);
    });
}

rsrt_status rsrt_multi_synchronize(rsrt_multi *m)
{
    return for_each_ctx(m, "synchronize", [&](rsrt_context *c, uint32_t) { return rsrt_synchronize(c); });
}

// The frame: one RCCL reduce(sum) of the accumulators onto device 0, then the copy to the host.
rsrt_status rsrt_multi_download(rsrt_multi *m, float *host_rgba, size_t n_floats)
{
    if (!m || m->ctx.empty()) return RSRT_ERR_INVALID_ARGUMENT;
    rsrt_context *c0 = m->ctx[0];
    if (!c0->accum) return mfail(m, RSRT_ERR_NOT_READY, "multi download", "no accumulator");
    if (!host_rgba || n_floats != (size_t)c0->acc_w * c0->acc_h * 4) return mfail(m, RSRT_ERR_INVALID_ARGUMENT, "multi download", "wrong float count");
    rsrt_status st = multi_reduce(m);
    if (st) return st;
    if ((st = rsrt_multi_synchronize(m))) return st;
    DeviceGuard g(c0->device);
    if (hipMemcpy(host_rgba, m->frame, n_floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return mfail(m, RSRT_ERR_HIP, "multi download", "hipMemcpy failed");
    return RSRT_OK;
}

// out_texture / the display pass of the reduced frame (device 0 runs the small kernels)
rsrt_status rsrt_multi_display_srgb8(rsrt_multi *m, uint32_t sample_total, uint8_t *host_rgba8, size_t n_bytes)
{
    if (!m || m->ctx.empty()) return RSRT_ERR_INVALID_ARGUMENT;
    rsrt_status st = multi_reduce(m);
    if (st) return st;
    rsrt_context *c0 = m->ctx[0];
    st = display_from(c0, m->frame, sample_total, host_rgba8, n_bytes);
    return st ? mfail(m, st, "multi display", rsrt_last_error(c0)) : RSRT_OK;
}

// sums of the per-device counters; times are the maximum over devices (they run side by side)
rsrt_status rsrt_multi_get_stats(rsrt_multi *m, rsrt_stats *out)
{
    if (!m || !out) return RSRT_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof *out);
    return for_each_ctx(m, "get_stats", [&](rsrt_context *c, uint32_t) {
        rsrt_stats s;
        const rsrt_status st = rsrt_get_stats(c, &s);
        if (st) return st;
        out->paths += s.paths; out->ext_rays += s.ext_rays; out->shadow_rays += s.shadow_rays; out->traversal_steps += s.traversal_steps;
        out->total_paths += s.total_paths; out->total_ext_rays += s.total_ext_rays; out->total_shadow_rays += s.total_shadow_rays;
        out->launches += s.launches;
        out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);
        out->total_kernel_ms = std::max(out->total_kernel_ms, s.total_kernel_ms);
        out->trace_kernel_ms = std::max(out->trace_kernel_ms, s.trace_kernel_ms);
        out->resolve_kernel_ms = std::max(out->resolve_kernel_ms, s.resolve_kernel_ms);
        out->reduce_ms = std::max(out->reduce_ms, s.reduce_ms);
        return RSRT_OK;
    });
}

} // extern "C"
