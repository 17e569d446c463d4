// pt_multi.cpp -- the multi-GPU leg of the C ABI: ONE process, one PtContext per device, interleaved
// row bands, and ONE RCCL gather of the film to the first device over xGMI.
//
// Every pixel is an independent unit whose RNG key is a pure function of (x, y) (src/main.rs:51) and whose result
// lands in its own film slot (src/main.rs:58, world.rs:318); the scene is tiny and replicated.  So device g renders
// the bands b with b % n == g (band_* of PtRenderParams) with no data-path collective, its film resolve writes the tile
// as 16 B per pixel (linear RGB + RGBA8: pt_render_device_packed) straight into the send buffer, and all devices meet in
// one ncclGather (rccl.h:745); the root then puts the gathered rows in image order.  The frame is bitwise independent
// of the number of devices.
//
// Host side (round 4): pt_multi_render_device only ENQUEUES -- per device the render, its ncclGather call, on the root the row
// permutation -- and returns; a caller may post frame k + 1 while the devices still run frame k.  Every device has two
// streams: its context's (the render: regenerating launches on the context's lanes, the resolve that writes the send buffer)
// and an EXCHANGE stream (its ncclGather call; on the root also the row permutation), and a ring of send buffers between them
// (kSendSlots) -- frame k's resolve records `ready` for the exchange stream, its gather records `sent` for the resolve of
// frame k + kSendSlots.  The gather is kept out of the render stream because its kernel can be LATE: RCCL's kernel is ONE workgroup
// of 4 x 132 VGPRs (~20 us of work), and with three regenerating launches of a device in flight (pt_api.cpp: lanes; six 80-VGPR
// waves per SIMD, a pending launch taking every slot that frees) it is not placed until they run out of work -- measured: 10-30 ms
// (profiles/r04/gather_kernel_beside_lanes.txt; not for want of free compute units, priority or scratch:
// reserved_cus_rejected.txt).  In the render stream that wait held back the next frame's resolve and with it the lanes: one
// pipeline drain per frame (6.10 ms per C2 frame on one device against 5.77 without the gather).  Behind the ring the render
// stream waits for a gather only when it wants that gather's send buffer back, and the gathers of several frames complete in one
// stall.  pt_multi_set_exchange(m, PT_EXCHANGE_COPY) takes the kernel out altogether (one DMA copy per device).  On the root the
// receive buffer is written by a gather that follows the previous frame's row permutation in the exchange stream.
// By default the calling thread enqueues the devices in turn and the n gather calls form one ncclGroupStart/End;
// pt_multi_set_threads(m, 1) gives every device its own host thread instead (pt_feeder.h: its launches, its ncclGather call
// on its own communicator -- the one-thread-per-device use of RCCL), the call then returns as soon as the frame is posted
// and pt_multi_sync() waits for the threads, then for the streams.
//
// RCCL is opened with dlopen the first time a multi-device object is created: a host that renders on one GPU never
// needs the library, and a process that already holds an RCCL (torch) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pathtrace_amd.h"
#include "pt_feeder.h"
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
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.Gather = (decltype(r.Gather))dlsym(h, "ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.GetErrorString)
        return pt_internal_fail(PT_ERR_UNSUPPORTED, "multi-GPU: the RCCL library lacks ncclCommInitAll / ncclGather");
    g_rccl = r;       // ncclCommCount / ncclGetVersion are optional (pt_multi_info reports 0 without them)
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

// Geometry of one frame over n devices
struct FrameShape {
    uint32_t W, H, n, band_rows, max_rows;
    size_t tile_px;              // pixels of the padded tile every device sends
};
FrameShape frame_shape(const PtCamera* cam, const PtRenderParams* prm, uint32_t n) {
    FrameShape f;
    f.W = cam->width; f.H = cam->height; f.n = n;
    f.band_rows = prm->band_rows ? prm->band_rows : default_band_rows(f.H, n);
    f.max_rows = 0;
    for (uint32_t g = 0; g < n; ++g) f.max_rows = std::max(f.max_rows, pt_tile_rows(f.H, f.band_rows, g, n));
    f.tile_px = std::max<size_t>((size_t)f.max_rows * f.W, 1);
    return f;
}

}  // namespace

// Send buffers per device.  The gather of a frame completes when the device next has room for its kernel, which with frames
// posted back to back is when the ring is full and the lanes run out of work (pt_multi.cpp head): one such stall per
// kSendSlots frames.  A frame uses fewer of them when kSendRingBytes would not hold that many of its tiles.
#ifndef PT_SEND_SLOTS
#define PT_SEND_SLOTS 8
#endif
constexpr uint32_t kSendSlots = PT_SEND_SLOTS;
constexpr size_t kSendRingBytes = (size_t)1 << 30;

struct PtMulti {
    std::vector<int> devices;
    std::vector<PtContext*> ctx;
    std::vector<ncclComm_t> comm;              // shared-device debug objects: none
    std::vector<DevMem> packed;                // per device: kSendSlots send buffers (16 B/pixel) its film resolves write in turn
    std::vector<hipStream_t> xs;               // per device: the exchange stream (gather; on the root the row permutation)
    std::vector<hipEvent_t> ev_ready, ev_sent; // per send buffer: written by the resolve / read by the gather
    std::vector<uint8_t> slot_used;
    uint32_t ring = kSendSlots;                // send buffers in use (<= kSendSlots: the frame size decides)
    DevMem recv, out_lin, out_rgba;            // root: gathered tiles; frame staging of the host entry
    std::unique_ptr<ptfeed::Feeder> feeder;    // one host thread per device; null: the caller's thread enqueues every device
    uint64_t frames = 0;                       // frames posted since creation
    std::mutex info_mu;
    std::vector<double> enqueue_us;            // per device: host time spent enqueueing its share, summed over the frames
    // Debug object (pt_debug_multi_create_shared): every context on ONE device, no RCCL -- each context copies its tile
    // into the receive buffer on its own stream where the real object runs ncclGather.  Host latches keep the emulation's
    // events in frame order (the collective does that by itself).
    bool shared = false;
    // Exchange by copies instead of ncclGather (pt_multi_set_exchange; always for the debug object): device g's exchange stream
    // copies its tile into the root's receive buffer (hipMemcpyPeerAsync: the DMA engines over xGMI, no kernel on any CU), the
    // root's waits for the n `copied` events.  The same events and latches as the debug object.
    bool copy = false;
    std::vector<hipEvent_t> ev_copied;
    hipEvent_t ev_unpacked = nullptr;
    std::mutex lat_mu;
    std::condition_variable lat_cv;
    std::vector<uint64_t> copied_frame;
    uint64_t unpacked_frame = 0;
};

namespace {

int multi_drain(PtMulti* m) {
    if (!m->feeder) return PT_OK;
    std::string err;
    const int rc = m->feeder->drain(&err);
    if (rc) return pt_internal_fail(rc, "%s", err.c_str());
    return PT_OK;
}

// Every posted frame is enqueued and complete -- WITHOUT collecting the statistics (pt_sync does that: a re-allocation between
// two frames must not cut the sums a caller reads after the last one)
int multi_quiesce(PtMulti* m) {
    int rc = multi_drain(m);
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        if (hipSetDevice(m->devices[g]) != hipSuccess || hipStreamSynchronize(pt_internal_stream(m->ctx[g])) != hipSuccess ||
            hipStreamSynchronize(m->xs[g]) != hipSuccess)
            if (!rc) rc = pt_internal_fail(PT_ERR_HIP, "multi-GPU: synchronising device %d failed", m->devices[g]);
    }
    return rc;
}
int sync_exchange(PtMulti* m) {
    for (size_t g = 0; g < m->xs.size(); ++g) {
        if (!m->xs[g]) continue;
        HIP_TRY(hipSetDevice(m->devices[g]));
        HIP_TRY(hipStreamSynchronize(m->xs[g]));
    }
    return PT_OK;
}

// Device g's share of frame `frame`: the render into the frame's send buffer and its part of the gather.  Runs on g's feeder
// thread (or on the caller's).  defer_gather: the caller issues the n ncclGather calls itself, inside one group, and records
// the `sent` events after it.
int enqueue_device(PtMulti* m, uint32_t g, const PtCamera& cam, const PtRenderParams& prm, const FrameShape& fs, uint64_t frame,
                   bool defer_gather) {
    // Exchange by copies: the host latches count FRAMES and must advance whatever happens in here -- a frame whose render was
    // refused (spp = 0, a tile larger than max_paths_in_flight, out of memory) or whose enqueue failed half-way still "copies"
    // and is still "unpacked" as far as the latches go, or the next frame's enqueue would wait for it for ever.
    struct Latch {
        PtMulti* m; uint32_t g; uint64_t frame;
        ~Latch() {
            if (!m->copy) return;
            { std::lock_guard<std::mutex> lk(m->lat_mu); if (m->copied_frame[g] < frame) m->copied_frame[g] = frame; }
            m->lat_cv.notify_all();
        }
    } latch{m, g, frame};
    HIP_TRY(hipSetDevice(m->devices[g]));
    PtRenderParams p = prm;
    p.band_rows = fs.band_rows; p.band_index = g; p.band_count = fs.n;
    const size_t slot = (size_t)g * kSendSlots + (size_t)((frame - 1) % m->ring);
    hipStream_t st = pt_internal_stream(m->ctx[g]), xs = m->xs[g];
    if (m->slot_used[slot]) HIP_TRY(hipStreamWaitEvent(st, m->ev_sent[slot], 0));     // the gather of frame - kSendSlots has read it
    m->slot_used[slot] = 1;
    const int rc = pt_render_device_packed(m->ctx[g], &cam, &p, m->packed[slot].p);   // an empty tile renders nothing
    std::string render_err;
    if (rc) render_err = pt_last_error();
    // (a failed render still takes part in the gather below: the other devices' gather calls would wait for this one for ever)
    HIP_TRY(hipEventRecord(m->ev_ready[slot], st));
    HIP_TRY(hipStreamWaitEvent(xs, m->ev_ready[slot], 0));
    if (m->copy) {
        // gather by copies: the tile goes to its place in the receive buffer once the row permutation of the previous
        // frame has read it
        if (frame > 1) {
            std::unique_lock<std::mutex> lk(m->lat_mu);
            m->lat_cv.wait(lk, [&] { return m->unpacked_frame + 1 >= frame; });
            lk.unlock();
            HIP_TRY(hipStreamWaitEvent(xs, m->ev_unpacked, 0));
        }
        char* const dst = (char*)m->recv.p + (size_t)g * fs.tile_px * 16;
        if (m->devices[g] == m->devices[0]) HIP_TRY(hipMemcpyAsync(dst, m->packed[slot].p, fs.tile_px * 16, hipMemcpyDeviceToDevice, xs));
        else HIP_TRY(hipMemcpyPeerAsync(dst, m->devices[0], m->packed[slot].p, m->devices[g], fs.tile_px * 16, xs));
        HIP_TRY(hipEventRecord(m->ev_copied[g], xs));
        HIP_TRY(hipEventRecord(m->ev_sent[slot], xs));
        // (copied_frame[g] = frame: the latch above, on every way out)
    } else if (!defer_gather) {
        // this device's call of THE gather (ncclGather, rccl.h:745): its communicator, its exchange stream, its thread
        NCCL_TRY(g_rccl.Gather(m->packed[slot].p, m->recv.p, fs.tile_px * 16, ncclUint8, 0, m->comm[g], xs));
        HIP_TRY(hipEventRecord(m->ev_sent[slot], xs));
    }
    if (rc) return pt_internal_fail(rc, "%s", render_err.c_str());
    return PT_OK;
}

// The root's tail of a frame: rows into image order, behind the gather in the root's exchange stream
int enqueue_root_tail(PtMulti* m, const FrameShape& fs, uint64_t frame, float* d_linear, uint8_t* d_rgba) {
    struct Latch {             // as in enqueue_device: the frame counts as unpacked on every way out
        PtMulti* m; uint64_t frame;
        ~Latch() {
            if (!m->copy) return;
            { std::lock_guard<std::mutex> lk(m->lat_mu); if (m->unpacked_frame < frame) m->unpacked_frame = frame; }
            m->lat_cv.notify_all();
        }
    } latch{m, frame};
    HIP_TRY(hipSetDevice(m->devices[0]));
    hipStream_t st = m->xs[0];
    if (m->copy) {
        {
            std::unique_lock<std::mutex> lk(m->lat_mu);
            m->lat_cv.wait(lk, [&] { for (uint64_t f : m->copied_frame) if (f < frame) return false; return true; });
        }
        for (uint32_t g = 1; g < fs.n; ++g) HIP_TRY(hipStreamWaitEvent(st, m->ev_copied[g], 0));
    }
    ptk::launch_film_unpack(m->recv.p, fs.W, fs.H, fs.band_rows, fs.n, fs.max_rows, d_linear, d_rgba, st);
    HIP_TRY(hipGetLastError());
    if (m->copy) HIP_TRY(hipEventRecord(m->ev_unpacked, st));
    return PT_OK;
}

// Events and latches of the exchange by copies: `copied` per device (on that device), `unpacked` on the root
int copy_events(PtMulti* m) {
    const size_t n = m->devices.size();
    m->copied_frame.assign(n, 0);
    m->ev_copied.assign(n, nullptr);
    for (size_t g = 0; g < n; ++g) {
        HIP_TRY(hipSetDevice(m->devices[g]));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_copied[g], hipEventDisableTiming));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipEventCreateWithFlags(&m->ev_unpacked, hipEventDisableTiming));
    return PT_OK;
}

int multi_alloc(uint32_t n, const int* devices, bool shared, PtMulti** out) {
    PtMulti* m = new PtMulti();
    m->devices.assign(devices, devices + n);
    m->ctx.assign(n, nullptr);
    m->packed.resize((size_t)n * kSendSlots);
    m->xs.assign(n, nullptr);
    m->ev_ready.assign((size_t)n * kSendSlots, nullptr);
    m->ev_sent.assign((size_t)n * kSendSlots, nullptr);
    m->slot_used.assign((size_t)n * kSendSlots, 0);
    m->enqueue_us.assign(n, 0.0);
    m->shared = m->copy = shared;
    int rc;
    auto streams = [&](uint32_t i) -> int {
        HIP_TRY(hipSetDevice(devices[i]));
        HIP_TRY(hipStreamCreateWithFlags(&m->xs[i], hipStreamNonBlocking));
        for (uint32_t k = 0; k < kSendSlots; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&m->ev_ready[(size_t)i * kSendSlots + k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&m->ev_sent[(size_t)i * kSendSlots + k], hipEventDisableTiming));
        }
        return PT_OK;
    };
    for (uint32_t i = 0; i < n; ++i) {
        if ((rc = pt_context_create(devices[i], &m->ctx[i])) || (rc = streams(i))) { pt_multi_destroy(m); return rc; }
        for (uint32_t k = 0; k < kSendSlots; ++k) m->packed[(size_t)i * kSendSlots + k].device = devices[i];
    }
    m->recv.device = m->out_lin.device = m->out_rgba.device = devices[0];
    *out = m;
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_multi_destroy(PtMulti* m) {
    if (!m) return PT_OK;
    (void)multi_drain(m);
    m->feeder.reset();                          // joins the threads: nothing touches the contexts from here on
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (m->ctx[i]) (void)pt_sync(m->ctx[i]);
    }
    (void)sync_exchange(m);
    for (ncclComm_t c : m->comm) if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < m->xs.size(); ++i) {
        (void)hipSetDevice(m->devices[i]);
        if (m->xs[i]) (void)hipStreamDestroy(m->xs[i]);
        for (uint32_t k = 0; k < kSendSlots; ++k) {
            if (m->ev_ready[i * kSendSlots + k]) (void)hipEventDestroy(m->ev_ready[i * kSendSlots + k]);
            if (m->ev_sent[i * kSendSlots + k]) (void)hipEventDestroy(m->ev_sent[i * kSendSlots + k]);
        }
    }
    for (auto& b : m->packed) b.release();
    m->recv.release(); m->out_lin.release(); m->out_rgba.release();
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    for (hipEvent_t e : m->ev_copied) if (e) (void)hipEventDestroy(e);
    if (m->ev_unpacked) (void)hipEventDestroy(m->ev_unpacked);
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
    PtMulti* m = nullptr;
    if ((rc = multi_alloc(n, devices, false, &m))) return rc;
    m->comm.assign(n, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), (int)n, devices);     // rccl.h:236
    if (r != ncclSuccess) {
        for (auto& c : m->comm) c = nullptr;
        pt_multi_destroy(m);
        return pt_internal_fail(PT_ERR_HIP, "ncclCommInitAll over %u device(s) failed: %s", n, g_rccl.GetErrorString(r));
    }
    // One 16-byte gather now, from this thread, inside a group: RCCL connects a pair of ranks at their first exchange,
    // and doing that here keeps the (blocking) connection set-up out of the per-device threads and out of the first frame.
    {
        auto warm = [&]() -> int {
            int rc2;
            for (uint32_t g = 0; g < n; ++g) {
                if ((rc2 = m->packed[(size_t)g * kSendSlots].ensure(16))) return rc2;
                HIP_TRY(hipSetDevice(devices[g]));
                HIP_TRY(hipMemsetAsync(m->packed[(size_t)g * kSendSlots].p, 0, 16, m->xs[g]));
            }
            if ((rc2 = m->recv.ensure(16 * (size_t)n))) return rc2;
            NCCL_TRY(g_rccl.GroupStart());
            for (uint32_t g = 0; g < n; ++g) {
                const ncclResult_t rg = g_rccl.Gather(m->packed[(size_t)g * kSendSlots].p, m->recv.p, 16, ncclUint8, 0, m->comm[g], m->xs[g]);
                if (rg != ncclSuccess) { (void)g_rccl.GroupEnd(); return pt_internal_fail(PT_ERR_HIP, "ncclGather (connection warm-up) failed on device %d: %s", devices[g], g_rccl.GetErrorString(rg)); }
            }
            NCCL_TRY(g_rccl.GroupEnd());
            for (uint32_t g = 0; g < n; ++g) {
                HIP_TRY(hipSetDevice(devices[g]));
                HIP_TRY(hipStreamSynchronize(m->xs[g]));
            }
            return PT_OK;
        };
        if ((rc = warm())) { pt_multi_destroy(m); return rc; }
    }
    // Default: the calling thread enqueues every device in turn and the n gather calls form one ncclGroup -- the form RCCL's own
    // tests run by default.  A device's share of a frame costs the host 13-14 us (tools/r04/multi_enqueue.py), so with frames
    // posted back to back one thread keeps eight devices fed with room to spare; pt_multi_set_threads(m, 1) gives every device
    // its own host thread (lowest latency of a single frame on many devices).
    *out = m;
    return PT_OK;
}

// Debug object: n contexts on ONE device, the gather emulated by device-to-device copies -- host threads, frame
// pipelining, the packed resolve, the row permutation and the per-device enqueue cost of an n-device frame on a one-GPU box.
int pt_debug_multi_create_shared(int device, uint32_t n, PtMulti** out) {
    if (!out || n == 0 || n > 64) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_multi_create_shared: bad argument");
    *out = nullptr;
    std::vector<int> devs(n, device);
    PtMulti* m = nullptr;
    int rc;
    if ((rc = multi_alloc(n, devs.data(), true, &m))) return rc;
    if ((rc = copy_events(m))) { pt_multi_destroy(m); return rc; }
    *out = m;
    return PT_OK;
}

// PT_EXCHANGE_RCCL (default): ONE ncclGather per frame.  PT_EXCHANGE_COPY: every device copies its tile to the root with the DMA
// engines (hipMemcpyPeerAsync) -- no kernel takes part, so nothing has to find room beside the regenerating launches (RCCL's
// kernel does not: pt_multi.cpp head).  Same frame either way.  Frames in flight are completed first.
int pt_multi_set_exchange(PtMulti* m, uint32_t mode) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    if (mode != PT_EXCHANGE_RCCL && mode != PT_EXCHANGE_COPY) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_set_exchange: unknown mode %u", mode);
    if (m->shared) return mode == PT_EXCHANGE_COPY ? PT_OK : pt_internal_fail(PT_ERR_UNSUPPORTED, "a shared-device debug object has no RCCL communicators");
    int rc = multi_quiesce(m);
    if (rc) return rc;
    if (mode == PT_EXCHANGE_COPY && m->ev_copied.empty()) {
        if ((rc = copy_events(m))) return rc;
        for (size_t g = 1; g < m->devices.size(); ++g) {       // direct copies over xGMI where the devices allow it (else staged by the runtime)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->devices[g], m->devices[0]) == hipSuccess && can) {
                (void)hipSetDevice(m->devices[g]);
                (void)hipDeviceEnablePeerAccess(m->devices[0], 0);
            }
            (void)hipGetLastError();                            // "already enabled" is fine
        }
    }
    // the latches count frames: both forms start from the frames posted so far
    { std::lock_guard<std::mutex> lk(m->lat_mu); m->unpacked_frame = m->frames; for (auto& f : m->copied_frame) f = m->frames; }
    m->copy = mode == PT_EXCHANGE_COPY;
    return PT_OK;
}

// 0 (default): the calling thread enqueues every device in turn and the gather calls form one ncclGroup; 1: one host thread per
// device feeds its stream.  Same frame either way.  (Both forms have only ever run with all contexts on ONE device or over one
// rank: no multi-GPU node was available in rounds 1-5.)
int pt_multi_set_threads(PtMulti* m, int enabled) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    int rc = multi_drain(m);
    if (rc) return rc;
    if (enabled && !m->feeder) m->feeder.reset(new ptfeed::Feeder((unsigned)m->devices.size()));
    if (!enabled) m->feeder.reset();
    return PT_OK;
}

uint32_t pt_multi_device_count(const PtMulti* m) { return m ? (uint32_t)m->devices.size() : 0u; }

int pt_multi_info(PtMulti* m, PtMultiInfo* out) {
    if (!m || !out) return pt_internal_fail(PT_ERR_INVALID_ARG, "null argument");
    int rc = multi_drain(m);
    if (rc) return rc;
    PtMultiInfo info{};
    info.n_devices = (uint32_t)m->devices.size();
    info.threaded = m->feeder ? 1u : 0u;
    info.exchange = m->copy ? PT_EXCHANGE_COPY : PT_EXCHANGE_RCCL;
    info.frames = m->frames;
    if (!m->comm.empty() && m->comm[0]) {
        int cnt = 0, ver = 0;
        if (g_rccl.CommCount && g_rccl.CommCount(m->comm[0], &cnt) == ncclSuccess) info.comm_count = (uint32_t)cnt;
        if (g_rccl.GetVersion && g_rccl.GetVersion(&ver) == ncclSuccess) info.rccl_version = (uint32_t)ver;
    }
    std::lock_guard<std::mutex> lk(m->info_mu);
    for (double us : m->enqueue_us) {
        info.enqueue_us_sum += us;
        info.enqueue_us_max = std::max(info.enqueue_us_max, us);
    }
    if (m->frames) { info.enqueue_us_sum /= (double)m->frames; info.enqueue_us_max /= (double)m->frames; }
    *out = info;
    return PT_OK;
}

int pt_multi_scene_upload(PtMulti* m, const PtObject* objs, uint32_t n_objs) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    int rc = multi_drain(m);
    if (rc) return rc;
    for (PtContext* c : m->ctx) {
        rc = pt_scene_upload(c, objs, n_objs);     // the scene is replicated (<= 160 KB for the reference's scenes)
        if (rc) return rc;
    }
    return PT_OK;
}

int pt_multi_set_tuning(PtMulti* m, const PtTuning* t) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    int rc = multi_drain(m);
    if (rc) return rc;
    for (PtContext* c : m->ctx) { rc = pt_context_set_tuning(c, t); if (rc) return rc; }
    return PT_OK;
}

// Post the whole frame: every device renders its bands, then the one gather, then the row permutation on the first
// device; pt_multi_sync() completes it.  d_linear_rgb / d_rgba8: buffers on the FIRST device,
// H*W*3 floats / H*W*4 bytes (d_rgba8 may be NULL).  params->band_rows = 0 picks about eight bands per device;
// band_index / band_count of params are ignored (the object owns the partition).  With host threads the call returns
// once the frame is posted to them; an error of a device's enqueue is reported by the next pt_multi_sync / _get_stats.
int pt_multi_render_device(PtMulti* m, const PtCamera* cam, const PtRenderParams* prm, float* d_linear, uint8_t* d_rgba) {
    if (!m || !cam || !prm || !d_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_render_device: null argument");
    const uint32_t n = (uint32_t)m->devices.size();
    const FrameShape fs = frame_shape(cam, prm, n);
    int rc;
    // buffers: a larger frame than any before re-allocates -- not under the feet of frames still in flight
    const uint32_t ring = (uint32_t)std::min<size_t>(kSendSlots, std::max<size_t>(2, kSendRingBytes / (fs.tile_px * 16)));
    bool grow = m->recv.cap < fs.tile_px * 16 * n || ring != m->ring;
    for (uint32_t g = 0; g < n; ++g)
        for (uint32_t k = 0; k < ring; ++k) grow = grow || m->packed[(size_t)g * kSendSlots + k].cap < fs.tile_px * 16;
    if (grow) {
        if ((rc = multi_quiesce(m))) return rc;          // (also when the ring changes length: slot numbers start afresh)
        m->ring = ring;
        for (uint32_t g = 0; g < n; ++g)
            for (uint32_t k = 0; k < ring; ++k)
                if ((rc = m->packed[(size_t)g * kSendSlots + k].ensure(fs.tile_px * 16))) return rc;
        if ((rc = m->recv.ensure(fs.tile_px * 16 * n))) return rc;
    }
    const uint64_t frame = ++m->frames;
    const PtCamera cam_v = *cam;
    const PtRenderParams prm_v = *prm;
    auto timed = [m](uint32_t g, std::chrono::steady_clock::time_point t0) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(m->info_mu);
        m->enqueue_us[g] += us;
    };
    if (m->feeder) {
        // one job per device, on that device's thread; the root's job ends with the row permutation
        for (uint32_t g = 0; g < n; ++g) {
            m->feeder->post(g, [=](std::string& err) -> int {
                const auto t0 = std::chrono::steady_clock::now();
                int r = enqueue_device(m, g, cam_v, prm_v, fs, frame, false);
                if (r) err = pt_last_error();
                if (g == 0) {              // the row permutation also behind a failed render: the frame's bookkeeping must complete
                    const int r2 = enqueue_root_tail(m, fs, frame, d_linear, d_rgba);
                    if (!r && r2) { r = r2; err = pt_last_error(); }
                }
                timed(g, t0);
                return r;
            });
        }
        return PT_OK;
    }
    // one thread: 1. every device renders its interleaved bands into its send buffer (no collective on the data path)
    // (a device whose render is refused does not stop the frame: the others still render, the exchange and the row permutation
    // still run -- every device takes part in the gather, every latch advances -- and the first error is returned at the end)
    int first_rc = PT_OK;
    std::string first_err;
    for (uint32_t g = 0; g < n; ++g) {
        const auto t0 = std::chrono::steady_clock::now();
        rc = enqueue_device(m, g, cam_v, prm_v, fs, frame, true);
        if (rc && !first_rc) { first_rc = rc; first_err = pt_last_error(); }
        timed(g, t0);
    }
    // 2. ONE gather of the padded tiles to the first device (ncclGather, rccl.h:745), every device on its exchange stream
    if (!m->copy) {
        const size_t k = (size_t)((frame - 1) % m->ring);
        NCCL_TRY(g_rccl.GroupStart());
        for (uint32_t g = 0; g < n; ++g) {
            const ncclResult_t r = g_rccl.Gather(m->packed[(size_t)g * kSendSlots + k].p, m->recv.p, fs.tile_px * 16, ncclUint8, 0, m->comm[g], m->xs[g]);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return pt_internal_fail(PT_ERR_HIP, "ncclGather failed on device %d: %s", m->devices[g], g_rccl.GetErrorString(r)); }
        }
        NCCL_TRY(g_rccl.GroupEnd());
        for (uint32_t g = 0; g < n; ++g) {
            HIP_TRY(hipSetDevice(m->devices[g]));
            HIP_TRY(hipEventRecord(m->ev_sent[(size_t)g * kSendSlots + k], m->xs[g]));
        }
    }
    // 3. rows into image order on the first device
    rc = enqueue_root_tail(m, fs, frame, d_linear, d_rgba);
    if (first_rc) return pt_internal_fail(first_rc, "%s", first_err.c_str());
    return rc;
}

int pt_multi_sync(PtMulti* m) {
    if (!m) return pt_internal_fail(PT_ERR_INVALID_ARG, "null multi-device object");
    int rc = multi_drain(m);            // every posted frame is enqueued on the streams ...
    for (PtContext* c : m->ctx) {       // ... and now complete on them: the renders,
        const int r2 = pt_sync(c);
        if (!rc) rc = r2;
    }
    const int r3 = sync_exchange(m);    // then the gather and the row permutation behind them
    return rc ? rc : r3;
}

// Counters of the frames since the last collection summed over the devices (times: the slowest device).
int pt_multi_get_stats(PtMulti* m, PtStats* out) {
    if (!m || !out) return pt_internal_fail(PT_ERR_INVALID_ARG, "null argument");
    int rc = multi_drain(m);
    if (rc) return rc;
    PtStats t{};
    for (PtContext* c : m->ctx) {
        PtStats s{};
        rc = pt_get_stats(c, &s);
        if (rc) return rc;
        t.samples += s.samples; t.vertices += s.vertices; t.shadow_rays += s.shadow_rays;
        t.bounce_launches += s.bounce_launches; t.batches = std::max(t.batches, s.batches);
        t.max_depth_reached = std::max(t.max_depth_reached, s.max_depth_reached);
        t.bounce_kernel_ms = std::max(t.bounce_kernel_ms, s.bounce_kernel_ms); t.total_ms = std::max(t.total_ms, s.total_ms);
        t.primary_vertices += s.primary_vertices; t.primary_kernel_ms = std::max(t.primary_kernel_ms, s.primary_kernel_ms);
        t.primary_launches += s.primary_launches;
        t.samples_expected += s.samples_expected;
    }
    *out = t;
    return PT_OK;
}

int pt_multi_render_host(PtMulti* m, const PtCamera* cam, const PtRenderParams* prm, float* out_linear, uint8_t* out_rgba) {
    if (!m || !cam || !prm || !out_linear) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_multi_render_host: null argument");
    const size_t px = (size_t)cam->width * cam->height;
    int rc;
    if (m->out_lin.cap < std::max<size_t>(px, 1) * 12 || (out_rgba && m->out_rgba.cap < std::max<size_t>(px, 1) * 4))
        if ((rc = multi_quiesce(m))) return rc;          // the staging buffers grow: not while a frame may still write them
    if ((rc = m->out_lin.ensure(std::max<size_t>(px, 1) * 3 * sizeof(float))) || (out_rgba && (rc = m->out_rgba.ensure(std::max<size_t>(px, 1) * 4)))) return rc;
    if ((rc = pt_multi_render_device(m, cam, prm, (float*)m->out_lin.p, out_rgba ? (uint8_t*)m->out_rgba.p : nullptr))) return rc;
    if ((rc = pt_multi_sync(m))) return rc;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpy(out_linear, m->out_lin.p, px * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_rgba) HIP_TRY(hipMemcpy(out_rgba, m->out_rgba.p, px * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Debug / parity entry: the frame of an n_virtual-device render produced on ONE context -- the n tiles are rendered one
// after another on ctx's device into the send-buffer form, placed in the gather buffer by device-to-device copies (where
// pt_multi_* runs ncclGather) and put in image order by the same kernel.  Exercises partition, packed resolve and unpack
// for any n on a one-GPU box; host output buffers, blocking.
int pt_debug_multi_emulate(PtContext* ctx, uint32_t n_virtual, const PtCamera* cam, const PtRenderParams* prm, float* out_linear,
                           uint8_t* out_rgba) {
    if (!ctx || !cam || !prm || !out_linear || n_virtual == 0) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_multi_emulate: bad argument");
    const FrameShape fs = frame_shape(cam, prm, n_virtual);
    const uint32_t n = n_virtual;
    const size_t px = std::max<size_t>((size_t)fs.W * fs.H, 1);
    DevMem packed, recv, olin, orgba;
    struct Free { DevMem* m[4]; ~Free() { for (DevMem* x : m) x->release(); } } guard{{&packed, &recv, &olin, &orgba}};
    int rc;
    if ((rc = packed.ensure(fs.tile_px * 16)) || (rc = recv.ensure(fs.tile_px * 16 * n)) || (rc = olin.ensure(px * 12)) || (rc = orgba.ensure(px * 4)))
        return rc;
    hipStream_t st = pt_internal_stream(ctx);
    for (uint32_t g = 0; g < n; ++g) {
        PtRenderParams p = *prm;
        p.band_rows = fs.band_rows; p.band_index = g; p.band_count = n;
        if ((rc = pt_render_device_packed(ctx, cam, &p, packed.p))) return rc;
        HIP_TRY(hipMemcpyAsync((char*)recv.p + (size_t)g * fs.tile_px * 16, packed.p, fs.tile_px * 16, hipMemcpyDeviceToDevice, st));
        if ((rc = pt_sync(ctx))) return rc;          // the send buffer is reused by the next virtual device
    }
    ptk::launch_film_unpack(recv.p, fs.W, fs.H, fs.band_rows, n, fs.max_rows, (float*)olin.p, out_rgba ? (uint8_t*)orgba.p : nullptr, st);
    HIP_TRY(hipGetLastError());
    if ((rc = pt_sync(ctx))) return rc;
    HIP_TRY(hipMemcpy(out_linear, olin.p, (size_t)fs.W * fs.H * 12, hipMemcpyDeviceToHost));
    if (out_rgba) HIP_TRY(hipMemcpy(out_rgba, orgba.p, (size_t)fs.W * fs.H * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Host-only self test of the per-device feeder threads (pt_feeder.h), no GPU needed: n_workers threads, n_frames jobs
// posted to each in frame order the way pt_multi_render_device posts a frame; job (w, f) appends its tag w << 32 | f to a
// shared log when it STARTS and again (bit 63 set) when it ENDS, after `spin` iterations of busy work.  order_out (2 *
// n_workers * n_frames entries) receives the log; fail_at >= 0: the job with that linear index w * n_frames + f fails, and
// the function returns what drain() reported.  The caller checks per-worker FIFO order and cross-worker overlap.
int pt_debug_feeder_selftest(uint32_t n_workers, uint32_t n_frames, uint32_t spin, int32_t fail_at, uint64_t* order_out, uint32_t* n_out) {
    if (!order_out || !n_out || n_workers == 0 || n_workers > 64) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_feeder_selftest: bad argument");
    std::mutex mu;
    std::vector<uint64_t> log;
    log.reserve(2 * (size_t)n_workers * n_frames);
    int rc;
    {
        ptfeed::Feeder feeder(n_workers);
        for (uint32_t f = 0; f < n_frames; ++f)
            for (uint32_t w = 0; w < n_workers; ++w)
                feeder.post(w, [&, w, f](std::string& err) -> int {
                    { std::lock_guard<std::mutex> lk(mu); log.push_back(((uint64_t)w << 32) | f); }
                    volatile uint64_t x = 0;
                    for (uint32_t k = 0; k < spin * (1u + (w + f) % 3u); ++k) x = x + k;
                    { std::lock_guard<std::mutex> lk(mu); log.push_back((1ull << 63) | ((uint64_t)w << 32) | f); }
                    if (fail_at >= 0 && (uint32_t)fail_at == w * n_frames + f) { err = "job " + std::to_string(fail_at) + " failed as asked"; return PT_ERR_HIP; }
                    return PT_OK;
                });
        std::string err;
        rc = feeder.drain(&err);
        if (rc) (void)pt_internal_fail(rc, "%s", err.c_str());
        std::string again;
        if (feeder.drain(&again) != 0) return pt_internal_fail(PT_ERR_UNSUPPORTED, "feeder: a failure was reported twice");
    }
    *n_out = (uint32_t)log.size();
    std::copy(log.begin(), log.end(), order_out);
    return rc;
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
