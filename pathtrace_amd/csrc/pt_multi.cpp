// pt_multi.cpp -- the multi-GPU leg of the C ABI: ONE process, one PtContext per device, interleaved
// row bands, and ONE RCCL gather of the film to the first device over xGMI.
//
// Every pixel is an independent unit whose RNG key is a pure function of (x, y) (src/main.rs:51) and whose result
// lands in its own film slot (src/main.rs:58, world.rs:318); the scene is tiny and replicated.  So device g renders
// the bands b with b % n == g (pt_render_device, band_* of PtRenderParams) with no data-path collective, packs its
// tile to 16 B per pixel (linear RGB + RGBA8) and all devices meet in one ncclGather (rccl.h:745) inside an
// ncclGroupStart/End of the one host thread; the root then puts the gathered rows in image order.  The frame is
// bitwise independent of the number of devices.
//
// RCCL is opened with dlopen the first time a multi-device object is created: a host that renders on one GPU never
// needs the library, and a process that already holds an RCCL (torch) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pathtrace_amd.h"
#include "pt_kernels.h"

// defined in pt_api.cpp
int pt_internal_fail(int code, const char* fmt, ...);
void pt_internal_register_atexit(void);
hipStream_t pt_internal_stream(PtContext* c);

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return PT_OK;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return pt_internal_fail(PT_ERR_UNSUPPORTED, "multi-GPU: cannot load RCCL (%s)", dlerror());
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.Gather = (decltype(r.Gather))dlsym(h, "ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.GetErrorString)
        return pt_internal_fail(PT_ERR_UNSUPPORTED, "multi-GPU: the RCCL library lacks ncclCommInitAll / ncclGather");
    g_rccl = r;
    return PT_OK;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return pt_internal_fail(e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                    hipGetErrorString(e_), __FILE__, __LINE__);                               \
    } while (0)
#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return pt_internal_fail(PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct DevMem {
    void* p = nullptr;
    size_t cap = 0;
    int device = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PT_OK;
        HIP_TRY(hipSetDevice(device));
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return PT_OK;
    }
    void release() { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } p = nullptr; cap = 0; }
};

// Band height giving every device about eight interleaved bands (the cost of a pixel depends on what it sees)
uint32_t default_band_rows(uint32_t height, uint32_t n) { return std::max(1u, height / std::max(1u, n * 8u)); }

}  // namespace

struct PtMulti {
    std::vector<int> devices;
    std::vector<PtContext*> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<DevMem> lin, rgba, packed;     // per device: its tile (f32 RGB, RGBA8) and the 16 B/pixel send buffer
    DevMem recv, out_lin, out_rgba;            // root: gathered tiles; frame staging of the host entry
};

extern "C" {

int pt_multi_destroy(PtMulti* m) {
    if (!m) return PT_OK;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (m->ctx[i]) (void)pt_sync(m->ctx[i]);
    }
    for (ncclComm_t c : m->comm) if (c) (void)g_rccl.CommDestroy(c);
    for (auto& b : m->lin) b.release();
    for (auto& b : m->rgba) b.release();
    for (auto& b : m->packed) b.release();
    m->recv.release(); m->out_lin.release(); m->out_rgba.release();
    for (PtContext* c : m->ctx) if (c) (void)pt_context_destroy(c);
    delete m;
    return PT_OK;
}

int pt_multi_create(const int* devices, uint32_t n, PtMulti** out) {
    if (!out || !devices || n == 0) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_create: null argument or no device");
    *out = nullptr;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_create: device %d listed twice", devices[i]);
    int rc = load_rccl();
    if (rc) return rc;
    PtMulti* m = new PtMulti();
    m->devices.assign(devices, devices + n);
    m->ctx.assign(n, nullptr); m->comm.assign(n, nullptr);
    m->lin.resize(n); m->rgba.resize(n); m->packed.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        if ((rc = pt_context_create(devices[i], &m->ctx[i]))) { pt_multi_destroy(m); return rc; }
        m->lin[i].device = m->rgba[i].device = m->packed[i].device = devices[i];
    }
    m->recv.device = m->out_lin.device = m->out_rgba.device = devices[0];
    const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), (int)n, devices);     // rccl.h:236
    if (r != ncclSuccess) {
        for (auto& c : m->comm) c = nullptr;
        pt_multi_destroy(m);
        return pt_internal_fail(PT_ERR_HIP, "ncclCommInitAll over %u device(s) failed: %s", n, g_rccl.GetErrorString(r));
    }
    *out = m;
    return PT_OK;
}

uint32_t pt_multi_device_count(const PtMulti* m) { return m ? (uint32_t)m->devices.size() : 0u; }

int pt_multi_scene_upload(PtMulti* m, const PtObject* objs, uint32_t n_objs) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    for (PtContext* c : m->ctx) {
        const int rc = pt_scene_upload(c, objs, n_objs);     // the scene is replicated (<= 160 KB for the reference's scenes)
        if (rc) return rc;
    }
    return PT_OK;
}

int pt_multi_set_tuning(PtMulti* m, const PtTuning* t) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    for (PtContext* c : m->ctx) { const int rc = pt_context_set_tuning(c, t); if (rc) return rc; }
    return PT_OK;
}

// Enqueue the whole frame: every device renders its bands, then the one gather, then the row permutation on the
// first device, whose stream is complete when the frame is.  d_linear_rgb / d_rgba8: buffers on the FIRST device,
// H*W*3 floats / H*W*4 bytes (d_rgba8 may be NULL).  params->band_rows = 0 picks about eight bands per device;
// band_index / band_count of params are ignored (the object owns the partition).
int pt_multi_render_device(PtMulti* m, const PtCamera* cam, const PtRenderParams* prm, float* d_linear, uint8_t* d_rgba) {
    if (!m || !cam || !prm || !d_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_render_device: null argument");
    const uint32_t n = (uint32_t)m->devices.size();
    const uint32_t W = cam->width, H = cam->height;
    const uint32_t band_rows = prm->band_rows ? prm->band_rows : default_band_rows(H, n);
    uint32_t max_rows = 0;
    for (uint32_t g = 0; g < n; ++g) max_rows = std::max(max_rows, pt_tile_rows(H, band_rows, g, n));
    const size_t tile_px = (size_t)max_rows * W;
    int rc;
    // 1. every device renders its interleaved bands into its own tile (no collective on the data path)
    for (uint32_t g = 0; g < n; ++g) {
        PtRenderParams p = *prm;
        p.band_rows = band_rows; p.band_index = g; p.band_count = n;
        const size_t px = (size_t)pt_tile_rows(H, band_rows, g, n) * W;
        if ((rc = m->lin[g].ensure(std::max<size_t>(px, 1) * 3 * sizeof(float))) || (rc = m->rgba[g].ensure(std::max<size_t>(px, 1) * 4)) ||
            (rc = m->packed[g].ensure(std::max<size_t>(tile_px, 1) * 16)))
            return rc;
        if ((rc = pt_render_device(m->ctx[g], cam, &p, (float*)m->lin[g].p, d_rgba ? (uint8_t*)m->rgba[g].p : nullptr))) return rc;
        HIP_TRY(hipSetDevice(m->devices[g]));
        ptk::launch_film_pack((const float*)m->lin[g].p, d_rgba ? (const uint8_t*)m->rgba[g].p : nullptr, (uint32_t)px, m->packed[g].p,
                              pt_internal_stream(m->ctx[g]));
        HIP_TRY(hipGetLastError());
    }
    if ((rc = m->recv.ensure(std::max<size_t>(tile_px, 1) * 16 * n))) return rc;
    // 2. ONE gather of the padded tiles to the first device (ncclGather, rccl.h:745), every device on its own stream
    NCCL_TRY(g_rccl.GroupStart());
    for (uint32_t g = 0; g < n; ++g) {
        const ncclResult_t r = g_rccl.Gather(m->packed[g].p, m->recv.p, tile_px * 16, ncclUint8, 0, m->comm[g], pt_internal_stream(m->ctx[g]));
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return pt_internal_fail(PT_ERR_HIP, "ncclGather failed on device %d: %s", m->devices[g], g_rccl.GetErrorString(r)); }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    // 3. rows into image order on the first device
    HIP_TRY(hipSetDevice(m->devices[0]));
    ptk::launch_film_unpack(m->recv.p, W, H, band_rows, n, max_rows, d_linear, d_rgba, pt_internal_stream(m->ctx[0]));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_multi_sync(PtMulti* m) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    for (PtContext* c : m->ctx) { const int rc = pt_sync(c); if (rc) return rc; }
    return PT_OK;
}

// Counters of the last frame summed over the devices (times: the slowest device).
int pt_multi_get_stats(PtMulti* m, PtStats* out) {
    if (!m || !out) return pt_internal_fail(PT_ERR_INVALID_ARG, "null argument");
    PtStats t{};
    for (PtContext* c : m->ctx) {
        PtStats s{};
        const int rc = pt_get_stats(c, &s);
        if (rc) return rc;
        t.samples += s.samples; t.vertices += s.vertices; t.shadow_rays += s.shadow_rays;
        t.bounce_launches += s.bounce_launches; t.batches = std::max(t.batches, s.batches);
        t.max_depth_reached = std::max(t.max_depth_reached, s.max_depth_reached);
        t.bounce_kernel_ms = std::max(t.bounce_kernel_ms, s.bounce_kernel_ms); t.total_ms = std::max(t.total_ms, s.total_ms);
        t.primary_vertices += s.primary_vertices; t.primary_kernel_ms = std::max(t.primary_kernel_ms, s.primary_kernel_ms);
        t.primary_launches += s.primary_launches;
    }
    *out = t;
    return PT_OK;
}

int pt_multi_render_host(PtMulti* m, const PtCamera* cam, const PtRenderParams* prm, float* out_linear, uint8_t* out_rgba) {
    if (!m || !cam || !prm || !out_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_render_host: null argument");
    const size_t px = (size_t)cam->width * cam->height;
    int rc;
    if ((rc = m->out_lin.ensure(std::max<size_t>(px, 1) * 3 * sizeof(float))) || (out_rgba && (rc = m->out_rgba.ensure(std::max<size_t>(px, 1) * 4)))) return rc;
    if ((rc = pt_multi_render_device(m, cam, prm, (float*)m->out_lin.p, out_rgba ? (uint8_t*)m->out_rgba.p : nullptr))) return rc;
    if ((rc = pt_multi_sync(m))) return rc;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpy(out_linear, m->out_lin.p, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_rgba) HIP_TRY(hipMemcpy(out_rgba, m->out_rgba.p, px * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Debug / parity entry: the frame of an n_virtual-device render produced on ONE context -- the n tiles are rendered one
// after another on ctx's device, packed, placed in the gather buffer by device-to-device copies (where pt_multi_*
// runs ncclGather) and put in image order by the same kernel.  Exercises partition, pack and unpack for any n on a
// one-GPU box; host output buffers, blocking.
int pt_debug_multi_emulate(PtContext* ctx, uint32_t n_virtual, const PtCamera* cam, const PtRenderParams* prm, float* out_linear,
                           uint8_t* out_rgba) {
    if (!ctx || !cam || !prm || !out_linear || n_virtual == 0) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_multi_emulate: bad argument");
    const uint32_t n = n_virtual, W = cam->width, H = cam->height;
    const uint32_t band_rows = prm->band_rows ? prm->band_rows : default_band_rows(H, n);
    uint32_t max_rows = 0;
    for (uint32_t g = 0; g < n; ++g) max_rows = std::max(max_rows, pt_tile_rows(H, band_rows, g, n));
    const size_t tile_px = std::max<size_t>((size_t)max_rows * W, 1), px = std::max<size_t>((size_t)W * H, 1);
    DevMem lin, rgba, packed, recv, olin, orgba;
    struct Free { DevMem* m[6]; ~Free() { for (DevMem* x : m) x->release(); } } guard{{&lin, &rgba, &packed, &recv, &olin, &orgba}};
    int rc;
    if ((rc = lin.ensure(tile_px * 12)) || (rc = rgba.ensure(tile_px * 4)) || (rc = packed.ensure(tile_px * 16)) ||
        (rc = recv.ensure(tile_px * 16 * n)) || (rc = olin.ensure(px * 12)) || (rc = orgba.ensure(px * 4)))
        return rc;
    hipStream_t st = pt_internal_stream(ctx);
    for (uint32_t g = 0; g < n; ++g) {
        PtRenderParams p = *prm;
        p.band_rows = band_rows; p.band_index = g; p.band_count = n;
        const size_t tp = (size_t)pt_tile_rows(H, band_rows, g, n) * W;
        if ((rc = pt_render_device(ctx, cam, &p, (float*)lin.p, out_rgba ? (uint8_t*)rgba.p : nullptr))) return rc;
        ptk::launch_film_pack((const float*)lin.p, out_rgba ? (const uint8_t*)rgba.p : nullptr, (uint32_t)tp, packed.p, st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync((char*)recv.p + (size_t)g * tile_px * 16, packed.p, tile_px * 16, hipMemcpyDeviceToDevice, st));
        if ((rc = pt_sync(ctx))) return rc;          // the tile buffers are reused by the next virtual device
    }
    ptk::launch_film_unpack(recv.p, W, H, band_rows, n, max_rows, (float*)olin.p, out_rgba ? (uint8_t*)orgba.p : nullptr, st);
    HIP_TRY(hipGetLastError());
    if ((rc = pt_sync(ctx))) return rc;
    HIP_TRY(hipMemcpy(out_linear, olin.p, (size_t)W * H * 12, hipMemcpyDeviceToHost));
    if (out_rgba) HIP_TRY(hipMemcpy(out_rgba, orgba.p, (size_t)W * H * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"

// One shot: a cached multi-device object per device list (pt_shutdown frees it).
static std::mutex g_multi_mu;
static PtMulti* g_multi = nullptr;
void pt_internal_multi_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_multi_mu);
    if (g_multi) pt_multi_destroy(g_multi);
    g_multi = nullptr;
}

extern "C" {

int pt_render_multi(const int* devices, uint32_t n_devices, const PtCamera* cam, const PtObject* objs, uint32_t n_objs,
                    const PtRenderParams* prm, float* out_linear, uint8_t* out_rgba) {
    if (!devices || n_devices == 0 || !cam || !prm || !out_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_render_multi: null argument");
    std::lock_guard<std::mutex> lk(g_multi_mu);
    int rc;
    if (g_multi && (g_multi->devices.size() != n_devices || !std::equal(devices, devices + n_devices, g_multi->devices.begin()))) {
        pt_multi_destroy(g_multi);
        g_multi = nullptr;
    }
    if (!g_multi) {
        if ((rc = pt_multi_create(devices, n_devices, &g_multi))) return rc;
        pt_internal_register_atexit();       // a process that only ever renders on several devices frees its comms too
    }
    if ((rc = pt_multi_scene_upload(g_multi, objs, n_objs))) return rc;
    return pt_multi_render_host(g_multi, cam, prm, out_linear, out_rgba);
}

// The kernels around the gather, for hosts with their own collective (pathtrace_amd/dist.py).
int pt_film_pack(void* hip_stream, const float* d_linear, const uint8_t* d_rgba, uint32_t n_pixels, void* d_packed) {
    if (n_pixels == 0) return PT_OK;
    if (!d_linear || !d_packed) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_film_pack: null buffer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return pt_internal_fail(PT_ERR_NO_DEVICE, "pt_film_pack: no HIP device");
    ptk::launch_film_pack(d_linear, d_rgba, n_pixels, d_packed, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}
int pt_film_unpack(void* hip_stream, const void* d_gathered, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t n_ranks,
                   uint32_t max_rows, float* d_linear, uint8_t* d_rgba) {
    if ((uint64_t)width * height == 0) return PT_OK;
    if (!d_gathered || !d_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_film_unpack: null buffer");
    if ((uint64_t)width * height > 0xFFFFFFFFull) return pt_internal_fail(PT_ERR_UNSUPPORTED, "pt_film_unpack: more than 2^32 pixels");
    if (n_ranks == 0) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_film_unpack: n_ranks is 0");
    if (band_rows == 0) band_rows = height;                      // one band = the whole image (PtRenderParams convention)
    // every image row must lie inside its rank's padded tile (rank 0 owns the largest one)
    if (max_rows < pt_tile_rows(height, band_rows, 0, n_ranks))
        return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_film_unpack: max_rows %u is smaller than the largest tile", max_rows);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return pt_internal_fail(PT_ERR_NO_DEVICE, "pt_film_unpack: no HIP device");
    ptk::launch_film_unpack(d_gathered, width, height, band_rows, n_ranks, max_rows, d_linear, d_rgba, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

}  // extern "C"
