// rsrt_comm.h — multi-GPU behind the C-ABI (included at the end of rsrt_api.hip; see include/rsrt.h, "multi-GPU").
//
// The path shards by pixel: tile (tx, ty) of the frame belongs to rank (tx + ty * skew) % world (owned_tile,
// rsrt_api.hip), scene and environment are replicated, and ONE exchange step per frame brings the image to the root.  Every
// pixel has exactly one owner, so the sum of the ranks' accumulators — x + 0 + ... + 0 — is a GATHER of every rank's own
// tiles: each rank packs them into a compact buffer (1 / world of the frame: 16.6 MB instead of 133 MB per rank at 4K on 8
// GPUs), the root receives world - 1 of them (grouped ncclSend / ncclRecv, point to point over xGMI) and scatters them into
// the frame.  No addition happens, so the N-GPU image equals the 1-GPU image bit for bit by construction.  The dense
// ncclReduce(sum, f32) of the W*H*4 accumulators that round 2 used is kept behind RSRT_COMM_MODE=reduce for A/B.  There is
// no other exchange step, hence no other collective.
// N > 1 over RCCL has not run on hardware yet (one-GPU boxes; RCCL refuses two ranks on one device): the peer-copy form of
// the same gather runs lists of 2 / 3 / 8 "devices" on one GPU, and the world-2/4/8 tests are the gate wherever GPUs are.
//
// Two forms, one implementation:
//   rsrt_comm_*   one process (or thread) per GPU: rank 0 makes a 128-byte id (rsrt_comm_unique_id), hands it to the
//                 other ranks by whatever channel the host has, every rank calls rsrt_comm_init on its own context.
//   rsrt_multi_*  one single-threaded caller, a LIST of devices — what the reference's `State` (src/state.rs:60-98,
//                 :760-833) would bind: one rsrt_context per device inside, ncclCommInitAll, one RCCL group per frame.
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
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
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
        RSRT_SYM(GetUniqueId) RSRT_SYM(CommInitRank) RSRT_SYM(CommInitAll) RSRT_SYM(CommDestroy) RSRT_SYM(Reduce) RSRT_SYM(Send) RSRT_SYM(Recv)
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

// rank that renders pixel (x, y): the arithmetic rsrt_render's kernels use (owned_tile, rsrt_api.hip)
inline uint32_t tile_owner(uint32_t width, uint32_t tile_w, uint32_t tile_h, uint32_t world, uint32_t x, uint32_t y)
{
    (void)width;
    return (x / tile_w + ((y / tile_h) % world) * (partition_skew(world) % world)) % world;
}

bool partition_args_ok(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world)
{
    return width && height && world && world <= 65535u && tile_w && tile_h && tile_w * tile_h <= 4096 && (tile_w * tile_h) % RT_WAVE == 0;
}

// geometry of the compact per-rank tile buffers of a frame
struct TileGeom {
    uint32_t width, height, tile_w, tile_h, tiles_x, tiles_y, per_row, skew, world, n_slots; // n_slots: pixels of one rank's compact buffer
};
TileGeom tile_geom(const rsrt_context *ctx)
{
    TileGeom g;
    g.width = ctx->acc_w; g.height = ctx->acc_h; g.tile_w = ctx->tile_w; g.tile_h = ctx->tile_h; g.world = ctx->world;
    g.tiles_x = (g.width + g.tile_w - 1) / g.tile_w;
    g.tiles_y = (g.height + g.tile_h - 1) / g.tile_h;
    g.per_row = (g.tiles_x + g.world - 1) / g.world;
    g.skew = partition_skew(g.world);
    g.n_slots = g.tiles_y * g.per_row * g.tile_w * g.tile_h;
    return g;
}

} // namespace

// rank's tiles of a W x H RGBA32F image -> its compact buffer [tile slot][pixel of the tile] (padding and out-of-frame pixels: 0)
__global__ void rt_pack_tiles_kernel(TileGeom g, uint32_t rank, const float4 *image, float4 *compact)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= g.n_slots) return;
    const uint32_t tile_px = g.tile_w * g.tile_h, j = slot / tile_px, p = slot % tile_px;
    uint32_t tx, ty;
    float4 v = float4{0.0f, 0.0f, 0.0f, 0.0f};
    if (owned_tile(j, rank, g.world, g.skew, g.tiles_x, g.per_row, tx, ty)) {
        const uint32_t px = tx * g.tile_w + p % g.tile_w, py = ty * g.tile_h + p / g.tile_w;
        if (px < g.width && py < g.height) v = image[(size_t)py * g.width + px];
    }
    compact[slot] = v;
}
// the compact buffers of ranks [0, world) laid end to end -> the frame (every pixel is written exactly once)
__global__ void rt_unpack_tiles_kernel(TileGeom g, const float4 *gathered, float4 *frame)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.n_slots * g.world) return;
    const uint32_t rank = (uint32_t)(i / g.n_slots), slot = (uint32_t)(i % g.n_slots);
    const uint32_t tile_px = g.tile_w * g.tile_h, j = slot / tile_px, p = slot % tile_px;
    uint32_t tx, ty;
    if (!owned_tile(j, rank, g.world, g.skew, g.tiles_x, g.per_row, tx, ty)) return;
    const uint32_t px = tx * g.tile_w + p % g.tile_w, py = ty * g.tile_h + p / g.tile_w;
    if (px < g.width && py < g.height) frame[(size_t)py * g.width + px] = gathered[i];
}

namespace {

// grow-only exchange buffer of a context: `world` compact buffers on the root (the gather lands there), one elsewhere
rsrt_status ensure_comm_buf(rsrt_context *ctx, size_t bytes)
{
    if (bytes <= ctx->comm_buf_bytes) return RSRT_OK;
    rsrt_status st = sync_all(ctx);
    if (st) return st;
    if (ctx->comm_buf) { (void)hipFree(ctx->comm_buf); ctx->comm_buf = nullptr; ctx->comm_buf_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->comm_buf, bytes));
    ctx->comm_buf_bytes = bytes;
    return RSRT_OK;
}

// The exchange step in two halves, so that a caller with several communicators (rsrt_multi) can put all the RCCL calls of
// a frame into ONE group: RCCL enqueues nothing before ncclGroupEnd, so whatever has to come AFTER the collective on the
// stream — the unpack kernel, the timing event, the context's last_event — is enqueued by the second half.
struct ExchangeInFlight {
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float4 *recv = nullptr;   // root: where the frame goes
    bool root = false, dense = false, begun = false;
};

// the dense W*H*4 ncclReduce of round 2 instead of the gather of compact tile buffers (RSRT_COMM_MODE=reduce, read once when the context is created;
// rsrt_comm_set_mode switches it — the documented fallback until the gather has run on a multi-GPU box)
bool comm_dense(const rsrt_context *ctx) { return ctx->comm_dense_mode; }

// first half: pack this rank's tiles and post its RCCL calls (root: world - 1 receives; others: one send)
rsrt_status exchange_begin(rsrt_context *ctx, uint32_t root, void *recv_device_rgba32f, hipStream_t stream, ExchangeInFlight &x)
{
    DeviceGuard g(ctx->device);
    if (!ctx->accum) return fail(ctx, RSRT_ERR_NOT_READY, "no accumulator");
    if (root >= ctx->comm_world) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "comm_reduce: root %u not in [0,%u)", root, ctx->comm_world);
    if (ctx->comm && (ctx->comm_world != ctx->world || ctx->comm_rank != ctx->rank))
        return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "comm_reduce: the partition (rank %u of %u) is not the communicator's (%u of %u)", ctx->rank, ctx->world, ctx->comm_rank, ctx->comm_world);
    x.stream = stream;
    x.root = ctx->comm_rank == root;
    x.recv = (x.root && recv_device_rgba32f) ? static_cast<float4 *>(recv_device_rgba32f) : ctx->accum;
    x.dense = comm_dense(ctx);
    const TileGeom tg = tile_geom(ctx);
    const size_t seg = (size_t)tg.n_slots * sizeof(float4);
    rsrt_status st = RSRT_OK;
    if (ctx->comm && !x.dense && ctx->comm_world > 1 && (st = ensure_comm_buf(ctx, x.root ? seg * tg.world : seg))) return st;
    if ((st = begin_work(ctx, stream))) return st;
    x.e0 = get_event(ctx); x.e1 = get_event(ctx);
    x.begun = true;
    HIP_TRY(ctx, hipEventRecord(x.e0, stream));
    if (!ctx->comm || ctx->comm_world == 1) { // a world of one: the frame is the accumulator
        if (x.recv != ctx->accum) HIP_TRY(ctx, hipMemcpyAsync(x.recv, ctx->accum, (size_t)ctx->acc_w * ctx->acc_h * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        return RSRT_OK;
    }
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    if (x.dense) {
        RCCL_TRY(ctx, rccl().Reduce(ctx->accum, x.recv, (size_t)ctx->acc_w * ctx->acc_h * 4, ncclFloat, ncclSum, (int)root, comm, stream));
        return RSRT_OK;
    }
    // gather of the compact buffers: each pixel has ONE owner, so "sum of the accumulators" is "every rank's own tiles, side
    // by side" — 1 / world of the bytes of a dense reduce, point to point over xGMI, and no addition at all
    float4 *mine = static_cast<float4 *>(ctx->comm_buf) + (x.root ? (size_t)ctx->comm_rank * tg.n_slots : 0);
    hipLaunchKernelGGL(rt_pack_tiles_kernel, dim3((tg.n_slots + 255) / 256), dim3(256), 0, stream, tg, ctx->comm_rank, ctx->accum, mine);
    HIP_TRY(ctx, hipGetLastError());
    if (x.root) {
        for (uint32_t r = 0; r < ctx->comm_world; r++)
            if (r != root) RCCL_TRY(ctx, rccl().Recv(static_cast<float4 *>(ctx->comm_buf) + (size_t)r * tg.n_slots, (size_t)tg.n_slots * 4, ncclFloat, (int)r, comm, stream));
    } else {
        RCCL_TRY(ctx, rccl().Send(mine, (size_t)tg.n_slots * 4, ncclFloat, (int)root, comm, stream));
    }
    return RSRT_OK;
}

// second half (after ncclGroupEnd when grouped): unpack on the root, timing event, end of the context's chain.  `ok` false:
// the first half (or the group) failed — the events go back to the pool and the chain is closed.
rsrt_status exchange_end(rsrt_context *ctx, ExchangeInFlight &x, bool ok)
{
    if (!x.begun) return RSRT_OK;
    DeviceGuard g(ctx->device);
    hipError_t he = hipSuccess;
    if (ok && ctx->comm && ctx->comm_world > 1 && !x.dense && x.root) {
        const TileGeom tg = tile_geom(ctx);
        const size_t n = (size_t)tg.n_slots * tg.world;
        hipLaunchKernelGGL(rt_unpack_tiles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, x.stream, tg, static_cast<const float4 *>(ctx->comm_buf), x.recv);
        he = hipGetLastError();
    }
    if (ok && he == hipSuccess) he = hipEventRecord(x.e1, x.stream);
    if (!ok || he != hipSuccess) {
        ctx->event_pool.push_back(x.e0);
        ctx->event_pool.push_back(x.e1);
        (void)end_work(ctx, x.stream);
        return he != hipSuccess ? fail(ctx, RSRT_ERR_HIP, "comm_reduce: %s", hipGetErrorString(he)) : RSRT_OK;
    }
    ctx->pending_reduce.push_back({x.e0, x.e1});
    ctx->cum_reduces++;
    return end_work(ctx, x.stream);
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

// The compact buffer of `rank`: *n_slots = its tile slots (the same for every rank); tiles_xy (may be NULL) receives 2 words
// per slot, the tile's (tx, ty), or 0xffffffff twice for a padding slot.  Pixel p of slot j is pixel (tx * tile_w + p % tile_w,
// ty * tile_h + p / tile_w) of the frame.
rsrt_status rsrt_partition_tiles(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world_size,
                                 uint32_t *tiles_xy, uint32_t *n_slots)
{
    if (!partition_args_ok(width, height, tile_w, tile_h, world_size) || rank >= world_size || !n_slots) return RSRT_ERR_INVALID_ARGUMENT;
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w, tiles_y = (height + tile_h - 1) / tile_h;
    const uint32_t per_row = (tiles_x + world_size - 1) / world_size, skew = partition_skew(world_size);
    *n_slots = tiles_y * per_row;
    if (tiles_xy)
        for (uint32_t j = 0; j < *n_slots; j++) {
            uint32_t tx, ty;
            const bool real = owned_tile(j, rank, world_size, skew, tiles_x, per_row, tx, ty);
            tiles_xy[2 * j] = real ? tx : 0xffffffffu;
            tiles_xy[2 * j + 1] = real ? ty : 0xffffffffu;
        }
    return RSRT_OK;
}

// ------------------------------------------------------------------ one process per GPU
int rsrt_comm_available(void) { return rccl().ok ? 1 : 0; } // (only dlopens librccl: no collective, no GPU work)

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

// 0: the gather of compact tile buffers (default), 1: the dense ncclReduce(sum) of the full accumulators.  Every rank must choose the same.
rsrt_status rsrt_comm_set_mode(rsrt_context *ctx, uint32_t dense_reduce)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    if (dense_reduce > 1u) return fail(ctx, RSRT_ERR_INVALID_ARGUMENT, "comm_set_mode: 0 (gather of compact tile buffers) or 1 (dense reduce)");
    DeviceGuard g(ctx->device);
    rsrt_status st = sync_all(ctx); // (an exchange in flight keeps its mode)
    if (st) return st;
    ctx->comm_dense_mode = dense_reduce != 0u;
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
    return rsrt_set_partition(ctx, rank, world_size, ctx->tile_w, ctx->tile_h); // this rank renders the tiles owned_tile() gives it
}

// The exchange step: the frame — the sum of the ranks' accumulators, i.e. every rank's own tiles — onto `root`.
// recv_device_rgba32f (root only; ignored elsewhere): where the full frame goes — NULL = in place, the root's own
// accumulator becomes the full frame (fine when it is cleared before the next render, as a batch renderer does; a
// progressive caller passes a separate buffer so that its accumulator keeps holding its own tiles only).
// Every rank packs its tiles into a compact buffer (1 / world of the frame), the root receives world - 1 of them
// (grouped ncclSend / ncclRecv: point to point, all of the root's xGMI links at once) and scatters them into the frame.
rsrt_status rsrt_comm_reduce(rsrt_context *ctx, uint32_t root, void *recv_device_rgba32f, void *hip_stream)
{
    if (!ctx) return RSRT_ERR_INVALID_ARGUMENT;
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    ExchangeInFlight x;
    const bool grouped = ctx->comm && ctx->comm_world > 1 && !comm_dense(ctx); // the root's receives must be posted together
    if (grouped && rccl().GroupStart() != ncclSuccess) return fail(ctx, RSRT_ERR_COMM, "ncclGroupStart failed");
    rsrt_status st = exchange_begin(ctx, root, recv_device_rgba32f, stream, x);
    if (grouped) {
        const ncclResult_t r = rccl().GroupEnd();
        if (r != ncclSuccess && !st) st = fail(ctx, RSRT_ERR_COMM, "ncclGroupEnd failed: %s", rccl().GetErrorString(r));
    }
    const rsrt_status st2 = exchange_end(ctx, x, st == RSRT_OK);
    return st ? st : st2;
}

} // extern "C"

// ------------------------------------------------------------------ one caller, a list of devices
struct rsrt_multi {
    std::vector<rsrt_context *> ctx;
    float4 *frame = nullptr; // on device ctx[0]: the W*H RGBA32F sum of the whole frame
    uint32_t frame_w = 0, frame_h = 0;
    std::string error;
};

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
    if (m->frame) { (void)hipDeviceSynchronize(); (void)hipFree(m->frame); m->frame = nullptr; }
    if (hipMalloc(&m->frame, (size_t)c0->acc_w * c0->acc_h * sizeof(float4)) != hipSuccess) return mfail(m, RSRT_ERR_OUT_OF_MEMORY, "multi frame buffer", "hipMalloc failed");
    m->frame_w = c0->acc_w;
    m->frame_h = c0->acc_h;
    return RSRT_OK;
}

// accumulators of all devices -> m->frame on device 0
rsrt_status multi_reduce(rsrt_multi *m)
{
    rsrt_status st = multi_ensure_frame(m);
    if (st) return st;
    rsrt_context *c0 = m->ctx[0];
    const size_t n_dev = m->ctx.size();
    if (n_dev == 1) { // one device: the frame is its accumulator (no communicator exists, none is needed)
        DeviceGuard g(c0->device);
        if ((st = begin_work(c0, c0->stream))) return mfail(m, st, "multi reduce", rsrt_last_error(c0));
        const hipError_t e = hipMemcpyAsync(m->frame, c0->accum, (size_t)c0->acc_w * c0->acc_h * sizeof(float4), hipMemcpyDeviceToDevice, c0->stream);
        st = end_work(c0, c0->stream);
        if (e != hipSuccess) return mfail(m, RSRT_ERR_HIP, "multi reduce", hipGetErrorString(e));
        return st ? mfail(m, st, "multi reduce", rsrt_last_error(c0)) : RSRT_OK;
    }
    if (!c0->comm) { // no RCCL (not installed, failed to come up, or the list names one device twice): peer copies of the compact buffers
        const TileGeom tg = tile_geom(c0);
        const size_t seg = (size_t)tg.n_slots * sizeof(float4);
        { DeviceGuard g(c0->device); if ((st = ensure_comm_buf(c0, seg * tg.world))) return mfail(m, st, "multi reduce", rsrt_last_error(c0)); }
        for (size_t i = 0; i < n_dev; i++) { // every device packs its own tiles, on its own stream, after its renders
            rsrt_context *ci = m->ctx[i];
            DeviceGuard gi(ci->device);
            float4 *mine = static_cast<float4 *>(c0->comm_buf) + i * tg.n_slots; // device 0 packs straight into the gather buffer
            if (i != 0) {
                if ((st = ensure_comm_buf(ci, seg))) return mfail(m, st, "multi reduce", rsrt_last_error(ci));
                mine = static_cast<float4 *>(ci->comm_buf);
            }
            if ((st = begin_work(ci, ci->stream))) return mfail(m, st, "multi reduce", rsrt_last_error(ci));
            hipLaunchKernelGGL(rt_pack_tiles_kernel, dim3((tg.n_slots + 255) / 256), dim3(256), 0, ci->stream, tg, (uint32_t)i, ci->accum, mine);
            const hipError_t e = hipGetLastError();
            st = end_work(ci, ci->stream);
            if (e != hipSuccess) return mfail(m, RSRT_ERR_HIP, "multi reduce (pack)", hipGetErrorString(e));
            if (st) return mfail(m, st, "multi reduce", rsrt_last_error(ci));
        }
        DeviceGuard g(c0->device);
        hipStream_t q = c0->stream;
        if ((st = begin_work(c0, q))) return mfail(m, st, "multi reduce", rsrt_last_error(c0));
        hipError_t e = hipSuccess;
        for (size_t i = 1; i < n_dev && e == hipSuccess; i++) {
            rsrt_context *ci = m->ctx[i];
            e = hipStreamWaitEvent(q, ci->last_event, 0); // after device i's pack
            if (e == hipSuccess) e = hipMemcpyPeerAsync(static_cast<float4 *>(c0->comm_buf) + i * tg.n_slots, c0->device, ci->comm_buf, ci->device, seg, q);
        }
        if (e == hipSuccess) {
            const size_t n = (size_t)tg.n_slots * tg.world;
            hipLaunchKernelGGL(rt_unpack_tiles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, tg, static_cast<const float4 *>(c0->comm_buf), m->frame);
            e = hipGetLastError();
        }
        if (e != hipSuccess) { (void)end_work(c0, q); return mfail(m, RSRT_ERR_HIP, "multi reduce (peer copies)", hipGetErrorString(e)); }
        if ((st = end_work(c0, q))) return mfail(m, st, "multi reduce", rsrt_last_error(c0));
        for (size_t i = 1; i < n_dev; i++) { // device i must not pack again before its buffer has been copied
            DeviceGuard gi(m->ctx[i]->device);
            if (hipStreamWaitEvent(m->ctx[i]->stream, c0->last_event, 0) != hipSuccess) return mfail(m, RSRT_ERR_HIP, "multi reduce", "hipStreamWaitEvent failed");
        }
        return RSRT_OK;
    }
    // RCCL, one group for the whole list: nothing is enqueued before ncclGroupEnd, so every context's unpack / timing
    // event / last_event is recorded AFTER it (exchange_end) — a render that follows on another stream waits for the
    // exchange, not for the stream position in front of it
    std::vector<ExchangeInFlight> x(n_dev);
    if (rccl().GroupStart() != ncclSuccess) return mfail(m, RSRT_ERR_COMM, "multi reduce", "ncclGroupStart failed");
    rsrt_status first = RSRT_OK;
    for (size_t i = 0; i < n_dev; i++) {
        st = exchange_begin(m->ctx[i], 0, m->frame, m->ctx[i]->stream, x[i]);
        if (st && !first) { first = st; mfail(m, st, "multi reduce", rsrt_last_error(m->ctx[i])); }
    }
    const ncclResult_t r = rccl().GroupEnd();
    if (r != ncclSuccess && !first) first = mfail(m, RSRT_ERR_COMM, "multi reduce: ncclGroupEnd", rccl().GetErrorString(r));
    for (size_t i = 0; i < n_dev; i++) {
        st = exchange_end(m->ctx[i], x[i], first == RSRT_OK);
        if (st && !first) { first = st; mfail(m, st, "multi reduce", rsrt_last_error(m->ctx[i])); }
    }
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
    if (m->frame && !m->ctx.empty()) { DeviceGuard g(m->ctx[0]->device); (void)hipFree(m->frame); }
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
    // A list of one device needs no communicator (and a single-GPU caller no RCCL at all: librccl is not even loaded).  Should
    // RCCL be there but fail to bring the communicators up, the list still works: the frame is then brought together by peer
    // copies of the compact tile buffers (rsrt_multi_uses_rccl() says which).
    if (n_devices > 1 && !duplicates && rccl().ok) {
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t r = rccl().CommInitAll(comms.data(), (int)n_devices, devices);
        if (r != ncclSuccess) {
            fprintf(stderr, "librsrt: ncclCommInitAll failed (%s); the frame will be gathered by peer copies instead\n", rccl().GetErrorString(r));
        } else {
            for (uint32_t i = 0; i < n_devices; i++) {
                m->ctx[i]->comm = comms[i];
                m->ctx[i]->comm_owned = true;
                m->ctx[i]->comm_rank = i;
                m->ctx[i]->comm_world = n_devices;
            }
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
