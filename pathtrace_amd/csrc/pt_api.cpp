// pt_api.cpp -- C ABI (include/pathtrace_amd.h) and the wavefront driver.
//
// render() == everything src/main.rs:43-60 does: enumerate the tile's pixels, give
// every pixel the RNG key (x, y) (main.rs:51), run SAMPLE_NUM paths per pixel
// (world.rs:296) and fill the two film buffers (world.rs:55-57).  Here the pixel
// x sample loop is a queue of paths in HBM advanced one vertex per kernel launch.
//
// There is no CPU fallback: without a HIP device every rendering entry point
// fails with PT_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pathtrace_amd.h"
#include "pt_bvh.h"
#include "pt_kernels.h"
#include "pt_sched.h"

static thread_local std::string g_err;
namespace {

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                     \
    } while (0)

template <class T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return PT_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_TRY(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return PT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr uint64_t kDefaultMaxPaths = 1ull << 26;
// ... where the level-0 launch keeps its paths in registers (k_paths_regen) a path in flight costs 12 bytes of sample buffer
// instead of 76, and every batch ends in a ~0.3 ms tail: larger batches (C3: 16 instead of 64)
constexpr uint64_t kDefaultMaxPathsRegen = 1ull << 28;
// Default number of workgroups.  What matters is that the queue segments of the RESIDENT waves fit the 256 MiB
// Infinity Cache (and that the dispatcher has enough workgroups to balance the end of the launch; much smaller
// segments pay more low-occupancy tail passes).  Scene in LDS, 6 waves/SIMD = 6144 resident waves: ~672 paths
// (42 KiB of queue) per wave -- measured at 67 M paths: 8192 workgroups 10.31 ms, 12288 10.12, 16384 9.86,
// 24576 9.61 (best), 32768 9.69.  Tiled scan and BVH kernels (4-5 waves/SIMD): ~1024 paths per wave (measured at
// 4 waves/SIMD: 2048 workgroups 11.6 ms, 8192 10.7, 16384 10.2 (best), 32768 10.6, 65536 11.7).
constexpr uint32_t kPathsPerWaveLds = 672;
constexpr uint32_t kPathsPerWave = 1024;
constexpr uint32_t kMinGrid = 256 * 8;         // level-0 launches: at least this many workgroups (small renders)
// continuation launches: their paths are sparse survivors, and a wave-iteration costs the same however few of its
// lanes carry a path, so more, emptier segments cost time -- and so do too few waves (latency of ~35 dependent passes).
// Measured on C2 / C1, ms of the continuation launch: 96 workgroups 1.75 / 5.0, 192 0.98 / 2.7, 384 0.60 / 1.6,
// 768 0.44 / 1.16, 1024 0.46, 1536 0.51 / 1.19, 3072 0.69 (round 1: 6144 -> +0.8).
constexpr uint32_t kContGrid = 1024;
#ifndef PT_REGEN_EXPORT
#define PT_REGEN_EXPORT 1
#endif
// lanes, buffer sets and the core size of overlapping launches: pt_sched.h (the scheduler's constants)
using ptsched::kLanes;
using ptsched::kSets;
constexpr uint32_t kStatsWords = 32;            // 16 x u64 at the front of the counter buffer: 8 render statistics, 8 words for measurement builds (PT_DRAIN_TIMING)
constexpr uint32_t kCountStride = (1 + ptk::kRegenCounters) * ptk::kRegenCounterStride;   // uint32 per batch parity: leftover count + chunk counters
#ifndef PT_SPLIT_BY_DEFAULT
#define PT_SPLIT_BY_DEFAULT 1
#endif
constexpr bool kSplitByDefault = PT_SPLIT_BY_DEFAULT != 0;   // scenes with a minority of Mirror objects: k_paths_regen_split (else the queue form)
#ifndef PT_REGEN_STATIC16
#define PT_REGEN_STATIC16 4
#endif
constexpr uint32_t kRegenStatic16 = PT_REGEN_STATIC16;   // k_paths_regen: sixteenths of a batch's chunks dealt round-robin, the rest by the counters
constexpr uint32_t kRegenExportBelow = PT_REGEN_EXPORT;  // ... and its waves hand over once the batch is used up and fewer paths than this are alive
                                               // (1: they run dry themselves and no continuation launch follows)
constexpr uint32_t kExportSmall = 64;          // a wave hands its segment over when fewer paths than this are left
                                               // (measured 32 ... 256: no difference beyond noise on C1 and C2)
// Tail hand-off: in launches of more than kExportMinPaths paths a wave whose segment falls below one chunk
// exports its leftovers to the overflow queue instead of walking them alone; the next launch takes them up.
// Measured (C2 / C1 ms per 1024^2 x 64 render): no hand-off 10.33 / 18.5; threshold 2^18 (3-4 levels) 11.2 / 16.9;
// 2^22 (2 levels) 10.28 / 15.65; exporting below 32 or 16 paths instead of 64 is slower.
#ifndef PT_EXPORT_MIN_LOG2
#define PT_EXPORT_MIN_LOG2 22
#endif
constexpr uint32_t kExportMinPaths = 1u << PT_EXPORT_MIN_LOG2;
// batches of more paths than this take the regenerating level-0 kernel (where the scene allows it)
#ifndef PT_REGEN_MIN_LOG2
#define PT_REGEN_MIN_LOG2 17
#endif
constexpr uint32_t kRegenMinPaths = 1u << PT_REGEN_MIN_LOG2;
constexpr uint32_t kWavesPerBlock = ptk::kBlock / 64;
// PT_ACCEL_AUTO: the BVH when the scene is larger than one LDS blob and spheres + 2.5 x triangles > 512 (C4-like
// scenes: the tiled scan costs ~0.11 ms per sphere and 67 M samples -- a Moeller-Trumbore test 2.5x that --, the BVH
// ~70 ms flat -> break-even near 600 sphere tests)
constexpr uint32_t kAutoBvhWeight = 512;

}  // namespace

// shared with pt_multi.cpp
int pt_internal_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
void pt_internal_multi_shutdown(void);

struct PtContext {
    int device = 0;
    uint32_t n_cus = 256;             // compute units of the device (grid of the regenerating level-0 launch)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // scene
    DevBuf<float4> scan, shape, mat, blob;
    DevBuf<float4> shape_x, mat_x, blob_x;   // the exact_math = 1 copies: the records carry per-object constants evaluated in that mode (k_scene_setup)
    DevBuf<ptk::Run> runs;
    DevBuf<uint32_t> lights;
    ptk::SceneView view{};
    bool has_scene = false;
    // BVH (PtRenderParams.accel): built from the host copy of the shape records at first use
    std::vector<float4> h_shape;
    std::vector<uint32_t> h_shape_tag;
    DevBuf<uint4> bvh_nodes;
    DevBuf<float4> bvh_rec, bvh_lead;
    DevBuf<uint32_t> bvh_ids;
    bool has_bvh = false;
    bool bvh_refused = false;         // the scene has a non-finite object: PT_ACCEL_AUTO stays with the linear scan
    bool auto_bvh = false;            // PT_ACCEL_AUTO would take the BVH for this scene (size rule above)
    uint32_t bvh_depth = 0;
    bool split_ok = false;            // a minority of the objects is Mirror: the regenerating form that batches their vertices pays
    uint32_t scan_counts[3] = {0, 0, 0};   // entries of the scan array by kind: spheres, single triangles, triangle pairs (pt_debug_scan_layout)
    // wavefront state
    DevBuf<float4> xchg;              // k_paths_regen_split: exchange stacks of every wave, one region per lane (stride: sched.xchg_stride)
    DevBuf<float4> queue[4];
    DevBuf<float4> bvh_aux, bvh_sray[2];   // accel = 1: per-slot scratch of the staged passes (k_paths_bvh)
    DevBuf<float4> ovf[2][2][4];      // overflow queues of the tail hand-off: per batch parity: leftover count, chunk counters (kCountStride)[plane]
    DevBuf<uint32_t> ovf_count;       // per batch parity: leftover count, chunk counters (kCountStride)
    // multi-batch renders: the continuation launches and the film resolve of batch k run on side_stream while the
    // level-0 launch of batch k + 1 runs on the caller's stream (their own queue and a second sample buffer)
    hipStream_t side_stream = nullptr;
    // regenerating launches: kLanes LANES (streams of their own) taken in turn by consecutive sample batches -- of one render or
    // of renders enqueued back to back --, so that the launches of batches k + 1 and k + 2 fill the device while the last waves
    // of batch k run dry; the resolves stay in order on the caller's stream
    hipStream_t lane_stream[kLanes] = {};
    hipEvent_t lane_done[kLanes] = {}, lane_begun[kLanes] = {}, ev_pre = nullptr, ev_switch = nullptr;
    // ... and kSets buffer sets (sample buffer + launch counters) taken in turn: a resolve gets few wave slots beside resident
    // regenerating launches (146 us of work take ~0.9 ms: measured), so the launch of batch k + kSets is the first to wait for
    // the resolve of batch k
    hipEvent_t set_free[kSets] = {};
    DevBuf<ptk::Rgb> lsamp3, lsamp4;
    // Which lane / buffer set comes next, which events have been recorded, which device-side words are known to be zero: the
    // scheduling state.  render_impl plans on a copy (ptsched::plan, pure) and commits it after the last operation was enqueued.
    ptsched::State sched;
    uint64_t expected_samples = 0;    // tile pixels x spp of the renders enqueued since the statistics were last collected (pt_sync compares)
    uint64_t capture_gcd = 0;         // gcd of the sample counts of the renders captured into graphs (replays add multiples of them)
    int64_t debug_fail_at = -1;       // pt_debug_fail_after: the stream operation of the next render that fails (test hook)
    DevBuf<float4> cqueue[4];
    DevBuf<float4> caux, csray[2];    // ... and, for accel = 1, its own staged-pass scratch
    DevBuf<ptk::Rgb> lsamp2;
    hipEvent_t ev_l0[2] = {nullptr, nullptr}, ev_resolved[2] = {nullptr, nullptr};
    uint32_t* h_ovf = nullptr;        // pinned read-back of one counter
    DevBuf<ptk::Rgb> lsamp;
    DevBuf<double> film;
    DevBuf<float> host_lin;       // device staging of pt_render_host
    DevBuf<uint8_t> host_rgba;
    unsigned long long* h_dstats = nullptr;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    PtStats stats{};
    uint32_t regen_occ[2][2][2] = {};          // cached occupancy query [exact_math][integrator][split] of this scene (0: not asked yet)
    std::vector<uint32_t> primary_events;      // slots of the level-0 launches
    PtTuning tuning{};                         // pt_context_set_tuning; 0 = library default
    uint32_t* h_posted = nullptr;              // host memory the device reads: number of the last lanes launch enqueued (BounceArgs.posted)
    uint32_t* d_posted = nullptr;              // ... its device address
    bool bvh_failed = false;                   // the BVH builder refused this scene (depth): PT_ACCEL_AUTO stays with the scan
    // pixel-list entries (pt_render_pixels, pt_ray_color)
    DevBuf<uint2> pixel_list;
    DevBuf<float4> inject[4];
    DevBuf<float> fn_in, fn_out;               // pt_debug_* function entries
    DevBuf<uint32_t> fn_words;
};

namespace {

// What a render does with the f64 film sums (pt_render_progressive carries them across calls).
struct FilmState {
    bool load = false;        // start from the sums in c->film
    bool store = false;       // keep the sums (more samples follow in a later call)
    uint32_t div = 0;         // samples the mean is taken over (0: this call's spp)
};
// Pixel-list render: film slot i <-> image pixel d_pixels[i]; inject: the paths are given (pt_ray_color) instead of
// generated by the camera.
struct ListRender {
    const uint2* d_pixels = nullptr;
    uint32_t n = 0;
    const float4* inject[4] = {nullptr, nullptr, nullptr, nullptr};
};

// The lanes' streams are created with a priority other than the default: the runtime keeps a pool of hardware queues per
// priority level and deals a level's streams over its pool, so the lanes then never share a hardware queue with a
// default-priority stream -- the caller's, on which this library puts the resolves and its waits for the lanes.  (A wait
// sitting in a shared hardware queue holds back whatever another stream put behind it there, e.g. the next lane launch.)
#ifndef PT_LANE_PRIORITY
#define PT_LANE_PRIORITY 1        // 1: the lowest priority the device offers (resolves go first), -1: the highest, 0: default
#endif
int lane_priority() {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 0;
    return PT_LANE_PRIORITY > 0 ? least : PT_LANE_PRIORITY < 0 ? greatest : 0;
}

int ensure_events(PtContext* c, size_t n) {
    while (c->ev_pool.size() < n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    return PT_OK;
}

std::vector<uint32_t> tile_row_list(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count) {
    std::vector<uint32_t> rows;
    if (band_rows == 0) band_rows = height ? height : 1;
    if (band_count == 0) band_count = 1;
    for (uint32_t y = 0; y < height; ++y)
        if ((y / band_rows) % band_count == band_index) rows.push_back(y);
    return rows;
}

float4 f4(double a, double b, double c, double d) { return make_float4((float)a, (float)b, (float)c, (float)d); }

// Shape records of one object: gather form (3 float4, pt_device.h) and scan records (1 float4 for a sphere, 3 for a triangle)
void shape_records(const PtObject& o, float4 gather[3], float4 scan[3], int* n_scan) {
    if (o.shape_tag == PT_SHAPE_SPHERE) {
        float4 s = f4(o.shape[0], o.shape[1], o.shape[2], o.shape[3]);
        gather[0] = s;
        gather[1] = make_float4(1.0f / s.w, 0, 0, 0);       // 1/radius (shape.rs:86)
        gather[2] = make_float4(0, 0, 0, 0);
        s.w = s.w * s.w;                                     // scan record carries r^2 (shape.rs:63)
        scan[0] = s;
        *n_scan = 1;
    } else {
        float v0[3], v1[3], v2[3];
        for (int k = 0; k < 3; ++k) { v0[k] = (float)o.shape[k]; v1[k] = (float)o.shape[3 + k]; v2[k] = (float)o.shape[6 + k]; }
        gather[0] = make_float4(v0[0], v0[1], v0[2], 0.f);
        gather[1] = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.f);   // edge1, shape.rs:163
        gather[2] = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.f);   // edge2, shape.rs:164
        ptbvh::triangle_scan_record(gather[0], gather[1], gather[2], scan);       // plane + barycentric gradients (pt_bvh.h)
        *n_scan = 3;
    }
}

// The scene as a launch in the given arithmetic mode sees it (the records carry constants evaluated in that mode)
ptk::SceneView view_for(const PtContext* c, uint32_t exact_math) {
    ptk::SceneView v = c->view;
    if (exact_math) {
        v.shape = c->shape_x.p; v.mat = c->mat_x.p;
        if (v.blob) v.blob = c->blob_x.p;
    }
    return v;
}

// Build and upload the BVH of the uploaded scene (once per scene).
int ensure_bvh(PtContext* c) {
    if (c->has_bvh) return PT_OK;
    if (c->bvh_refused) return fail(PT_ERR_UNSUPPORTED, "accel: the scene has object(s) with a NaN/inf coordinate; use the linear scan");
    if (c->bvh_failed) return fail(PT_ERR_UNSUPPORTED, "accel: the BVH of this scene is deeper than the traversal stack; use the linear scan");
    if (c->view.n_objs >= (1u << 28)) return fail(PT_ERR_UNSUPPORTED, "accel: %u objects exceed the 2^28 leaf slots", c->view.n_objs);
    ptbvh::Built b = ptbvh::build(c->h_shape.data(), c->h_shape_tag.data(), c->view.n_objs);
    static_assert(ptbvh::kStackDepth == ptk::kBvhStack, "traversal stack depth");
    static_assert(ptbvh::kMaxLeaf == ptk::kBvhMaxLeaf, "leaf size the traversal unrolls for");
    if (b.non_finite) {
        c->bvh_refused = true;
        return fail(PT_ERR_UNSUPPORTED, "accel: %u object(s) with a NaN/inf coordinate; the linear scan's answer for them "
                                        "depends on the scan order, use the linear scan", b.non_finite);
    }
    if (b.depth + 2u > ptbvh::kStackDepth || b.stack_need > ptbvh::kStackDepth) {
        c->bvh_failed = true;        // a property of the scene: do not rebuild on every render
        return fail(PT_ERR_UNSUPPORTED, "accel: BVH (depth %u, stack need %u) exceeds the traversal stack", b.depth, b.stack_need);
    }
    int rc;
    if ((rc = c->bvh_nodes.ensure(b.qnodes.size() + 2)) || (rc = c->bvh_rec.ensure(b.leaf_rec.size() + 3)) ||
        (rc = c->bvh_ids.ensure(b.leaf_ids.size() + 4)) || (rc = c->bvh_lead.ensure(b.leaf_lead.size() + 4)))
        return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!b.qnodes.empty()) HIP_TRY(hipMemcpy(c->bvh_nodes.p, b.qnodes.data(), b.qnodes.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!b.leaf_rec.empty()) HIP_TRY(hipMemcpy(c->bvh_rec.p, b.leaf_rec.data(), b.leaf_rec.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (!b.leaf_ids.empty()) HIP_TRY(hipMemcpy(c->bvh_ids.p, b.leaf_ids.data(), b.leaf_ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!b.leaf_lead.empty()) HIP_TRY(hipMemcpy(c->bvh_lead.p, b.leaf_lead.data(), b.leaf_lead.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->view.bvh.nodes = c->bvh_nodes.p; c->view.bvh.rec = c->bvh_rec.p; c->view.bvh.ids = c->bvh_ids.p; c->view.bvh.lead = c->bvh_lead.p;
    c->view.bvh.root = b.root;
    c->view.bvh.scene_abs = b.scene_abs;
    for (int k = 0; k < 3; ++k) { c->view.bvh.grid_min[k] = b.grid_min[k]; c->view.bvh.grid_cell[k] = b.grid_cell[k]; }
    c->bvh_depth = b.depth;
    c->has_bvh = true;
    return PT_OK;
}

}  // namespace

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }
uint32_t pt_abi_version(void) { return PT_ABI_VERSION; }

void pt_default_params(PtRenderParams* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->spp = 3000;          // world.rs:18
    p->spp_offset = 0;
    p->min_depth = 4;       // rendering.rs:6
    p->max_depth = 50;      // rendering.rs:7
    p->integrator = PT_INTEGRATOR_MIS;   // Cargo.toml:7 default feature
    p->t_min = 0.001;       // rendering.rs:41
    p->band_rows = 0;
    p->band_index = 0;
    p->band_count = 1;
    p->max_paths_in_flight = 0;
    p->profile = 0;
    p->accel = PT_ACCEL_AUTO;
    p->n_devices = 1;
}

uint32_t pt_tile_rows(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count) {
    return (uint32_t)tile_row_list(height, band_rows, band_index, band_count).size();
}

int pt_context_create(int device, PtContext** out) {
    if (!out) return fail(PT_ERR_INVALID_ARG, "pt_context_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PT_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PT_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    PtContext* c = new PtContext();
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = (uint32_t)cus;
    }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "hipStreamCreateWithFlags failed");
    }
    c->stream = c->own_stream;
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "hipStreamCreateWithFlags failed");
    }
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&c->ev_l0[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_resolved[k], hipEventDisableTiming) != hipSuccess) {
            delete c;
            return fail(PT_ERR_HIP, "hipEventCreate failed");
        }
    for (int k = 0; k < kLanes; ++k)
        if (hipStreamCreateWithPriority(&c->lane_stream[k], hipStreamNonBlocking, lane_priority()) != hipSuccess ||
            hipEventCreateWithFlags(&c->lane_done[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->lane_begun[k], hipEventDisableTiming) != hipSuccess) {
            delete c;
            return fail(PT_ERR_HIP, "lane stream / event creation failed");
        }
    for (int k = 0; k < kSets; ++k)
        if (hipEventCreateWithFlags(&c->set_free[k], hipEventDisableTiming) != hipSuccess) {
            delete c;
            return fail(PT_ERR_HIP, "hipEventCreate failed");
        }
    if (hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "hipEventCreate failed");
    }
    if (hipHostMalloc((void**)&c->h_posted, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_posted, c->h_posted, 0) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "context allocation failed (mapped host word)");
    }
    *c->h_posted = 0u;
    if (hipHostMalloc((void**)&c->h_dstats, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_ovf, 4 * sizeof(uint32_t)) != hipSuccess ||
        hipEventCreate(&c->ev_begin) != hipSuccess || hipEventCreate(&c->ev_end) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "context allocation failed");
    }
    *out = c;
    return PT_OK;
}

int pt_context_destroy(PtContext* c) {
    if (!c) return PT_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    for (int k = 0; k < kLanes; ++k) if (c->lane_stream[k]) (void)hipStreamSynchronize(c->lane_stream[k]);
    c->scan.release(); c->shape.release(); c->mat.release(); c->blob.release(); c->runs.release(); c->lights.release();
    c->bvh_nodes.release(); c->bvh_rec.release(); c->bvh_ids.release(); c->bvh_lead.release();
    c->bvh_aux.release(); c->bvh_sray[0].release(); c->bvh_sray[1].release();
    for (auto& b : c->queue) b.release();
    for (auto& par : c->ovf) for (auto& q : par) for (auto& b : q) b.release();
    for (auto& b : c->cqueue) b.release();
    c->caux.release(); c->csray[0].release(); c->csray[1].release();
    c->lsamp2.release(); c->lsamp3.release(); c->lsamp4.release();
    for (int k = 0; k < 2; ++k) {
        if (c->ev_l0[k]) (void)hipEventDestroy(c->ev_l0[k]);
        if (c->ev_resolved[k]) (void)hipEventDestroy(c->ev_resolved[k]);
    }
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    for (int k = 0; k < kLanes; ++k) {
        if (c->lane_done[k]) (void)hipEventDestroy(c->lane_done[k]);
        if (c->lane_begun[k]) (void)hipEventDestroy(c->lane_begun[k]);

        if (c->lane_stream[k]) (void)hipStreamDestroy(c->lane_stream[k]);
    }
    for (int k = 0; k < kSets; ++k) if (c->set_free[k]) (void)hipEventDestroy(c->set_free[k]);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
    c->xchg.release();
    c->ovf_count.release();
    if (c->h_ovf) (void)hipHostFree(c->h_ovf);
    if (c->h_posted) (void)hipHostFree(c->h_posted);
    c->lsamp.release(); c->film.release(); c->host_lin.release(); c->host_rgba.release();
    c->pixel_list.release(); c->fn_in.release(); c->fn_out.release(); c->fn_words.release();
    for (auto& b : c->inject) b.release();
    if (c->h_dstats) (void)hipHostFree(c->h_dstats);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PT_OK;
}

int pt_context_set_stream(PtContext* c, void* hip_stream) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "null context");
    hipStream_t next = hip_stream == PT_STREAM_LEGACY_DEFAULT ? nullptr                       // HIP's legacy default stream (handle 0)
                                                               : hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (next != c->stream && c->ev_switch) {
        // The context's buffers (sample buffers, counters, film sums, statistics) are handed from render to render in the order of
        // ONE stream: what is already enqueued on the old stream comes before anything the new one gets.
        (void)hipSetDevice(c->device);
        if (hipEventRecord(c->ev_switch, c->stream) == hipSuccess) (void)hipStreamWaitEvent(next, c->ev_switch, 0);
        (void)hipGetLastError();
    }
    c->stream = next;
    return PT_OK;
}

int pt_context_set_tuning(PtContext* c, const PtTuning* t) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "null context");
    c->tuning = t ? *t : PtTuning{};
    return PT_OK;
}

// World::new's tail (world.rs:213-225) + flattening of Box<dyn Shape>/Box<dyn Material>.
int pt_scene_upload(PtContext* c, const PtObject* objs, uint32_t n) {
    if (!c || (!objs && n)) return fail(PT_ERR_INVALID_ARG, "pt_scene_upload: null argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> scan, shape(3 * (size_t)n + 1), mat(2 * (size_t)n + 1), obj_scan(3 * (size_t)n + 1);
    std::vector<int> obj_ns(n + 1, 0);
    std::vector<ptk::Run> runs;
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < n; ++i) {
        const PtObject& o = objs[i];
        if (o.shape_tag > PT_SHAPE_TRIANGLE) return fail(PT_ERR_INVALID_ARG, "object %u: bad shape_tag %u", i, o.shape_tag);
        if (o.mat_tag > PT_MAT_OREN_NAYAR) return fail(PT_ERR_INVALID_ARG, "object %u: bad mat_tag %u", i, o.mat_tag);
        shape_records(o, &shape[3 * (size_t)i], &obj_scan[3 * (size_t)i], &obj_ns[i]);
        float p[6] = {(float)o.mat[0], (float)o.mat[1], (float)o.mat[2], (float)o.mat[3], (float)o.mat[4], (float)o.mat[5]};
        uint32_t emits = 0;
        if (o.mat_tag == PT_MAT_EMISSIVE) {
            // emit().length() > 0 (world.rs:219-222), evaluated in f32
            float l2 = std::fmaf(p[2], p[2], std::fmaf(p[1], p[1], p[0] * p[0]));
            emits = std::sqrt(l2) > 0.0f ? 1u : 0u;
        }
        if (o.mat_tag == PT_MAT_OREN_NAYAR) {
            float s2 = p[3] * p[3];                              // OrenNayar::new, material.rs:182-193
            float A = 1.0f - 0.5f * s2 / (s2 + 0.33f);
            float B = 0.45f * s2 / (s2 + 0.09f);
            p[3] = A; p[4] = B; p[5] = 0.f;
        }
        uint32_t bits = o.mat_tag | (o.shape_tag << 8) | (emits << 16);
        float fb;
        std::memcpy(&fb, &bits, 4);
        mat[2 * i] = make_float4(fb, p[0], p[1], p[2]);
        mat[2 * i + 1] = make_float4(p[3], p[4], p[5], 0.f);
        if (emits) lights.push_back(i);
    }
    // Scan array: runs of same-kind primitives in object order (the order decides closest-hit ties, world.rs:281-287).
    // Two consecutive triangles whose records carry the SAME vertex v0 and the SAME plane normal bit for bit -- the two
    // halves of a parallelogram fanned from one corner, like every wall of World::new() (world.rs:82-182) -- form a PAIR:
    // determinant, t, the range test and the hit point are then literally the same numbers for both, and the scan
    // computes them once (tripair_test, pt_kernels.hip).  Nothing changes in any result.
    for (uint32_t i = 0; i < n;) {
        const bool tri = objs[i].shape_tag == PT_SHAPE_TRIANGLE;
        bool pair = false;
        if (tri && i + 1 < n && objs[i + 1].shape_tag == PT_SHAPE_TRIANGLE) {
            const float4 *a = &obj_scan[3 * (size_t)i], *b = &obj_scan[3 * (size_t)i + 3];
            pair = std::memcmp(&a[0], &b[0], 3 * sizeof(float)) == 0 && std::memcmp(&a[1], &b[1], 3 * sizeof(float)) == 0;   // n, v0
        }
        const uint32_t tag = !tri ? (uint32_t)ptk::kRunSphere : pair ? (uint32_t)ptk::kRunTrianglePair : (uint32_t)ptk::kRunTriangle;
        if (runs.empty() || runs.back().tag != tag) {
            ptk::Run r;
            r.tag = tag; r.first_obj = i; r.count = 0; r.off4 = (uint32_t)scan.size();
            runs.push_back(r);
        }
        runs.back().count++;
        if (!pair) scan.insert(scan.end(), &obj_scan[3 * (size_t)i], &obj_scan[3 * (size_t)i] + obj_ns[i]);
        if (pair) {
            // pair record, 5 float4 in the order tripair_test reads them: (n, -) (v0, -) and then the four barycentric gradients
            // back to back from a 16-byte boundary -- (N1, N2.x) (N2.y, N2.z, N1'.x, N1'.y) (N1'.z, N2') -- so that the part only
            // rays inside the pair's t range read is three aligned 16-byte reads (round 5; before: five 8-byte pieces)
            const float4 *a = &obj_scan[3 * (size_t)i], *b = &obj_scan[3 * (size_t)i + 3];
            scan.push_back(make_float4(a[0].x, a[0].y, a[0].z, 0.f));
            scan.push_back(make_float4(a[1].x, a[1].y, a[1].z, 0.f));
            scan.push_back(make_float4(a[0].w, a[1].w, a[2].x, a[2].y));
            scan.push_back(make_float4(a[2].z, a[2].w, b[0].w, b[1].w));
            scan.push_back(make_float4(b[2].x, b[2].y, b[2].z, b[2].w));
        }
        i += pair ? 2u : 1u;
    }
    c->scan_counts[0] = c->scan_counts[1] = c->scan_counts[2] = 0;
    for (const ptk::Run& r : runs) c->scan_counts[r.tag == ptk::kRunSphere ? 0 : r.tag == ptk::kRunTriangle ? 1 : 2] += r.count;
    int rc;
    if ((rc = c->scan.ensure(scan.size() + 1))) return rc;
    if ((rc = c->shape.ensure(shape.size()))) return rc;
    if ((rc = c->mat.ensure(mat.size()))) return rc;
    if ((rc = c->shape_x.ensure(shape.size()))) return rc;
    if ((rc = c->mat_x.ensure(mat.size()))) return rc;
    if ((rc = c->runs.ensure(runs.size() + 1))) return rc;
    if ((rc = c->lights.ensure(lights.size() + 1))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));   // the previous scene may still be in use
    ptsched::on_scene(c->sched);                 // statistics of renders of the previous scene do not carry over
    c->expected_samples = 0;
    // (the statistics words are zero whenever no render is pending; on the context's stream, which is idle here: a plain hipMemset
    // runs on the legacy default stream, which a non-blocking stream does not wait for)
    if (c->ovf_count.p && hipMemsetAsync(c->ovf_count.p, 0, kStatsWords * sizeof(uint32_t), c->stream) == hipSuccess &&
        hipStreamSynchronize(c->stream) == hipSuccess)
        c->sched.stats_clean = 1;
    c->capture_gcd = 0;                          // (graphs captured over the previous scene must not be replayed any more: its buffers are gone)
    std::memset(c->regen_occ, 0, sizeof c->regen_occ);
    if (!scan.empty()) HIP_TRY(hipMemcpy(c->scan.p, scan.data(), scan.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->shape.p, shape.data(), shape.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mat.p, mat.data(), mat.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->shape_x.p, shape.data(), shape.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mat_x.p, mat.data(), mat.size() * sizeof(float4), hipMemcpyHostToDevice));
    // a triangle's unit normal and 1 / area, once per object and arithmetic mode, by the device's own expressions
    ptk::launch_scene_setup_fast(c->shape.p, c->mat.p, n, c->stream);
    ptk::launch_scene_setup_exact(c->shape_x.p, c->mat_x.p, n, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!runs.empty()) HIP_TRY(hipMemcpy(c->runs.p, runs.data(), runs.size() * sizeof(ptk::Run), hipMemcpyHostToDevice));
    if (!lights.empty()) HIP_TRY(hipMemcpy(c->lights.p, lights.data(), lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->view.scan = c->scan.p; c->view.shape = c->shape.p; c->view.mat = c->mat.p;
    c->view.runs = c->runs.p; c->view.lights = c->lights.p;
    c->view.blob = nullptr; c->view.blob_f4 = 0;
    if (n <= ptk::kSmallObjs) {
        // LDS image of a small scene: [scan | shape 3n | mat 2n | runs | lights (padded to 16 B)]
        std::vector<float4> blob(scan.begin(), scan.end());
        blob.insert(blob.end(), shape.begin(), shape.begin() + 3 * (size_t)n);
        blob.insert(blob.end(), mat.begin(), mat.begin() + 2 * (size_t)n);
        static_assert(sizeof(ptk::Run) == sizeof(float4), "Run must be one float4");
        for (const ptk::Run& r : runs) { float4 f; std::memcpy(&f, &r, sizeof f); blob.push_back(f); }
        for (size_t i = 0; i < lights.size(); i += 4) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (size_t k = 0; k < 4 && i + k < lights.size(); ++k) w[k] = lights[i + k];
            float4 f; std::memcpy(&f, w, sizeof f); blob.push_back(f);
        }
        if ((rc = c->blob.ensure(blob.size() + 1)) || (rc = c->blob_x.ensure(blob.size() + 1))) return rc;
        for (float4* dst : {c->blob.p, c->blob_x.p}) {
            if (!blob.empty()) HIP_TRY(hipMemcpy(dst, blob.data(), blob.size() * sizeof(float4), hipMemcpyHostToDevice));
            // shape and material records as k_scene_setup left them in this mode's arrays
            const bool x = dst == c->blob_x.p;
            if (n) HIP_TRY(hipMemcpy(dst + scan.size(), x ? c->shape_x.p : c->shape.p, 3 * (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice));
            if (n) HIP_TRY(hipMemcpy(dst + scan.size() + 3 * (size_t)n, x ? c->mat_x.p : c->mat.p, 2 * (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice));
        }
        c->view.blob = c->blob.p;
        c->view.blob_f4 = (uint32_t)blob.size();
    }
    c->view.scan_f4 = (uint32_t)scan.size();
    c->view.n_runs = (uint32_t)runs.size(); c->view.n_objs = n; c->view.n_lights = (uint32_t)lights.size();
    c->view.diffuse_only = 1u;
    for (uint32_t i = 0; i < n; ++i)
        if (objs[i].mat_tag != PT_MAT_LAMBERT && objs[i].mat_tag != PT_MAT_EMISSIVE) c->view.diffuse_only = 0u;
    {   // k_paths_regen_split sets the Mirror vertices aside: worth it while they are the exception
        uint32_t n_mirror = 0;
        for (uint32_t i = 0; i < n; ++i) n_mirror += objs[i].mat_tag == PT_MAT_MIRROR;
        c->split_ok = n_mirror != 0 && 2 * n_mirror <= n;
        c->view.no_mirror = n_mirror == 0 ? 1u : 0u;
        c->view.no_oren_nayar = 1u;
        for (uint32_t i = 0; i < n; ++i)
            if (objs[i].mat_tag == PT_MAT_OREN_NAYAR) c->view.no_oren_nayar = 0u;
    }
    c->view.bvh = ptk::BvhView{};
    c->has_bvh = false;
    c->bvh_refused = false;
    c->bvh_failed = false;
    {
        uint64_t tris = 0;
        for (uint32_t i = 0; i < n; ++i) tris += objs[i].shape_tag == PT_SHAPE_TRIANGLE;
        c->auto_bvh = n > ptk::kSmallObjs && 2 * (uint64_t)(n - tris) + 5 * tris > 2 * (uint64_t)kAutoBvhWeight;
    }
    c->h_shape.assign(shape.begin(), shape.begin() + 3 * (size_t)n);
    c->h_shape_tag.resize(n);
    for (uint32_t i = 0; i < n; ++i) c->h_shape_tag[i] = objs[i].shape_tag;
    c->has_scene = true;
    return PT_OK;
}

}  // extern "C"

namespace {

// ---- the launch scheduler: plan (pt_sched.h, pure) -> execute (here) -> commit
hipStream_t sched_stream(const PtContext* c, hipStream_t caller, uint32_t id) {
    return id == ptsched::kStreamCaller ? caller : id == ptsched::kStreamSide ? c->side_stream : c->lane_stream[id - ptsched::kStreamLane0];
}
hipEvent_t sched_event(const PtContext* c, const ptsched::Op& o) {
    using namespace ptsched;
    const uint32_t e = o.event;
    if (e == kEvBegin) return c->ev_begin;
    if (e == kEvEnd) return c->ev_end;
    if (e == kEvPre) return c->ev_pre;
    if (e >= kEvL0 && e < kEvResolved) return c->ev_l0[e - kEvL0];
    if (e >= kEvResolved && e < kEvLaneDone) return c->ev_resolved[e - kEvResolved];
    if (e >= kEvLaneDone && e < kEvLaneBegun) return c->lane_done[e - kEvLaneDone];
    if (e >= kEvLaneBegun && e < kEvSetFree) return c->lane_begun[e - kEvLaneBegun];
    if (e >= kEvSetFree && e < kEvPool) return c->set_free[e - kEvSetFree];
    return c->ev_pool[o.pool];
}
DevBuf<ptk::Rgb>& sample_buffer(PtContext* c, uint32_t set) { return set == 3 ? c->lsamp4 : set == 2 ? c->lsamp3 : set ? c->lsamp2 : c->lsamp; }

// An operation of a render could not be enqueued: what was enqueued before it runs to its end (or fails with the device), then
// the scheduling state starts over with nothing known to be clean (ptsched::on_failure).  The render's outputs are undefined;
// the statistics of the renders since the last collection are dropped with it.
void sched_recover(PtContext* c, hipStream_t st, const ptsched::State& planned) {
    for (int k = 0; k < kLanes; ++k) (void)hipStreamSynchronize(c->lane_stream[k]);
    (void)hipStreamSynchronize(c->side_stream);
    (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    ptsched::on_failure(c->sched, planned);
    c->expected_samples = 0;
    if (c->ovf_count.p && hipMemsetAsync(c->ovf_count.p, 0, kStatsWords * sizeof(uint32_t), st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
        c->sched.stats_clean = 1;
    (void)hipGetLastError();
}

// The render driver behind every rendering entry: everything src/main.rs:43-60 does for the tile (or, with `list`,
// for a pixel list / a set of given rays).  Enqueues on the context's streams and returns; no host synchronisation.
// Three steps: PREPARE (validate, size, allocate every buffer the render will touch -- nothing is enqueued yet, so an
// error here leaves the context as it was), PLAN (ptsched::plan on a copy of the scheduling state: pure), EXECUTE (the
// plan's stream operations in order) and COMMIT of the copy.  A failure during EXECUTE: sched_recover.
int render_impl(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, const FilmState& fs, const ListRender* list,
                float* d_linear, uint8_t* d_rgba, void* d_packed = nullptr) {
    if (!c || !cam || !prm) return fail(PT_ERR_INVALID_ARG, "render: null argument");
    if (!c->has_scene) return fail(PT_ERR_INVALID_ARG, "render: no scene uploaded");
    const bool inject = list && list->inject[0];
    if (!inject && (cam->width < 2 || cam->height < 2))   // get_ray_with_offset divides by width-1 / height-1 (camera.rs:140-141)
        return fail(PT_ERR_INVALID_ARG, "camera %ux%u: width and height must be >= 2", cam->width, cam->height);
    if (prm->spp == 0) return fail(PT_ERR_INVALID_ARG, "spp must be > 0");
    if (prm->integrator > PT_INTEGRATOR_BRDF_ONLY) return fail(PT_ERR_INVALID_ARG, "unknown integrator %u", prm->integrator);
    if (prm->accel > PT_ACCEL_AUTO) return fail(PT_ERR_INVALID_ARG, "unknown accel %u", prm->accel);
    const uint32_t band_count = prm->band_count ? prm->band_count : 1;
    if (prm->band_index >= band_count) return fail(PT_ERR_INVALID_ARG, "band_index %u >= band_count %u", prm->band_index, band_count);
    if (list && band_count != 1) return fail(PT_ERR_INVALID_ARG, "pixel-list renders take the whole image (band_count = 1)");
    // the resolve stores a pixel's RGBA8 as one 32-bit word (a byte-offset pointer would fault on the device)
    if (d_rgba && (uintptr_t)d_rgba % 4u) return fail(PT_ERR_INVALID_ARG, "render: the RGBA8 output buffer must be 4-byte aligned");
    if (d_linear && (uintptr_t)d_linear % 4u) return fail(PT_ERR_INVALID_ARG, "render: the linear output buffer must be 4-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    PtRenderParams resolved = *prm;      // PT_ACCEL_AUTO -> what actually runs
    if (prm->accel == PT_ACCEL_AUTO) {
        resolved.accel = PT_ACCEL_LINEAR;
        if (c->auto_bvh && !c->bvh_refused && !c->bvh_failed) {
            const std::string keep = g_err;              // a refused BVH is not an error of this call
            if (ensure_bvh(c) == PT_OK) resolved.accel = PT_ACCEL_BVH;
            else g_err = keep;
        }
    }
    prm = &resolved;

    const uint64_t np64 = list ? list->n : (uint64_t)pt_tile_rows(cam->height, prm->band_rows, prm->band_index, band_count) * cam->width;
    const uint64_t tile_rows = list ? ((np64 + 65535u) >> 16) : np64 / cam->width;
    // Statistics belong to the renders enqueued since they were last collected (pt_sync / pt_get_stats): a caller that
    // pipelines several renders behind one synchronisation gets their sums (vertices, launches, kernel times), total_ms from
    // the first one's start to the last one's end.  One render per synchronisation: the statistics of that render, as ever.
    auto begin_period = [c]() {        // the first render since the last collection
        std::memset(&c->stats, 0, sizeof c->stats);
        c->primary_events.clear();
        c->expected_samples = 0;
    };
    if (np64 == 0) {                   // empty tile: nothing to render
        if (!c->sched.stats_pending) begin_period();
        return PT_OK;
    }
    if (!d_linear && !d_packed) return fail(PT_ERR_INVALID_ARG, "render: the linear output buffer is null");
    if (!list && (cam->width > 65535u || tile_rows > 65535u))   // (tile row, x) share one word of the path state
        return fail(PT_ERR_UNSUPPORTED, "tile %ux%llu: width and tile rows must be < 65536", cam->width, (unsigned long long)tile_rows);
    // scene / job that can take the regenerating level-0 kernel (decided below, once the batch size is known)
    // level0_form 3 / default for scenes with a minority of Mirror objects: the regenerating form with the Mirror vertices batched
    const bool lds_job = !list && !prm->accel && c->view.n_objs <= ptk::kSmallObjs && c->view.blob_f4 != 0;
    const bool split = lds_job && (c->tuning.level0_form == 3 || (c->tuning.level0_form == 0 && c->split_ok && kSplitByDefault));
    // default (round 3): every scene in LDS takes a regenerating form -- compiled for its material set (diffuse only; no
    // Mirror; everything); with the Mirror vertices batched where they are the exception (split, above).  The queue form
    // is level0_form = 1 (all-Mirror Cornell scene: queue form 12.6 ms, regenerating 10.0, split 13.8; tools/r03/mirror_forms.py)
    const bool regen_scene = split || ((c->tuning.level0_form == 2 || c->tuning.level0_form == 0) && lds_job);
    uint64_t cap = prm->max_paths_in_flight ? prm->max_paths_in_flight : (regen_scene ? kDefaultMaxPathsRegen : kDefaultMaxPaths);
    if (cap > (1ull << 30)) cap = 1ull << 30;
    if (np64 > cap)
        return fail(PT_ERR_UNSUPPORTED, "tile of %llu pixels exceeds max_paths_in_flight %llu; use more bands",
                    (unsigned long long)np64, (unsigned long long)cap);
    const uint32_t np = (uint32_t)np64;
    const uint32_t spp = inject ? 1u : prm->spp;
    uint32_t nb_max = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(cap / np, 65535u), spp);
    if (nb_max == 0) nb_max = 1;
    const size_t n_paths_max = (size_t)np * nb_max;
    const uint32_t n_batches = (spp + nb_max - 1) / nb_max;

    // Queue segments: one per wave of the (fixed) grid; pass 0 deals 64-path chunks round-robin, so a segment holds
    // at most ceil(chunks / waves) chunks.
    const uint32_t export_small = c->tuning.export_below ? std::min(256u, std::max(1u, c->tuning.export_below)) : kExportSmall;
    const uint32_t chunks_max = (uint32_t)((n_paths_max + 63) / 64);
    uint32_t grid = prm->workgroups;
    const bool small_scene = c->view.n_objs <= ptk::kSmallObjs || prm->accel;     // the tiled scan exports per workgroup (< 256 paths)
    const uint32_t paths_per_wave = (c->view.n_objs <= ptk::kSmallObjs && !prm->accel) ? kPathsPerWaveLds : kPathsPerWave;
    if (!grid) {
        const uint64_t want = (n_paths_max + (uint64_t)paths_per_wave * kWavesPerBlock - 1) / ((uint64_t)paths_per_wave * kWavesPerBlock);
        grid = (uint32_t)std::max<uint64_t>(want, kMinGrid);
    }
    grid = std::min<uint32_t>(grid, (chunks_max + kWavesPerBlock - 1) / kWavesPerBlock);
    if (grid == 0) grid = 1;
    const uint32_t nw = grid * kWavesPerBlock;
    const uint32_t seg_cap = ((chunks_max + nw - 1) / nw) * 64u;
    // Tail hand-off: the level-0 launch of a large batch exports what its waves have left below one chunk; ONE
    // continuation launch of fixed size takes that queue up.  It reads the count on the device.
    // (scenes that take a regenerating level-0 kernel have their own threshold: that launch needs no continuation launch)
    const bool hand_off = !inject && n_paths_max > (regen_scene ? kRegenMinPaths : kExportMinPaths);
    // Such a batch over a scene in LDS takes the regenerating level-0 kernel: paths live in registers, a lane whose
    // path ends takes the next one of the batch; grid = the waves the device holds at once (k_paths_regen).
    const bool regen = regen_scene && hand_off;
    // its grid = the workgroups the device holds at once: the kernel's occupancy with THIS scene's LDS blob (a 128-object
    // scene leaves room for fewer workgroups per CU than the compile-time figure)
    uint32_t regen_per_cu = split ? ptk::kRegenWavesSplit : c->view.diffuse_only ? ptk::kRegenWavesDiffuse : ptk::kRegenWavesGeneric;
    if (regen_scene) {
        ptk::BounceArgs q{};
        q.sc = view_for(c, prm->exact_math);
        q.integrator = prm->integrator;
        q.xchg = split ? reinterpret_cast<float4*>(1) : nullptr;       // selects the kernel only
        uint32_t& occ = c->regen_occ[prm->exact_math ? 1 : 0][prm->integrator ? 1 : 0][split ? 1 : 0];   // per scene (pt_scene_upload clears it)
        if (occ == 0u) occ = prm->exact_math ? ptk::regen_blocks_per_cu_exact(q) : ptk::regen_blocks_per_cu_fast(q);
        if (occ != 0u) regen_per_cu = std::min(regen_per_cu, occ);
    }
    const uint32_t regen_capacity = c->n_cus * regen_per_cu;          // workgroups the device holds at once
    const uint32_t regen_grid = std::min(65536u, prm->workgroups ? prm->workgroups : c->tuning.regen_workgroups ? c->tuning.regen_workgroups : regen_capacity);
    const uint32_t cont_grid = c->tuning.cont_workgroups ? std::min(65536u, c->tuning.cont_workgroups) : kContGrid;
    const uint32_t nw_cont = cont_grid * kWavesPerBlock;
    // leftovers per wave of the level-0 launch: < export_small from a wave-private segment, < 256 per workgroup
    // (= 64 per wave) from a workgroup-shared one
    const size_t ovf_slots = (size_t)(regen ? regen_grid * kWavesPerBlock : nw) * std::max(export_small, 64u) + 64u;
    const uint32_t seg_cap_cont = (uint32_t)((((ovf_slots + 63) / 64 + nw_cont - 1) / nw_cont) * 64u);
    const size_t q_slots_cont = hand_off ? (size_t)nw_cont * seg_cap_cont : 0;
    const size_t q_slots = regen ? std::max<size_t>(q_slots_cont, 64) : std::max((size_t)nw * seg_cap, q_slots_cont);
    const uint32_t regen_export = split ? 1u : c->tuning.export_below ? std::min(export_small, 64u) : kRegenExportBelow;   // the split form has no hand-over
    // (sized by the grid the launch really takes: never more workgroups than the largest batch has chunks for)
    const uint32_t regen_launch_grid = std::min<uint32_t>(regen_grid, (chunks_max + kWavesPerBlock - 1) / kWavesPerBlock);

    // ---- the job as the scheduler sees it (pt_sched.h)
    ptsched::Job job;
    job.n_batches = n_batches;
    job.regen = regen; job.split = regen && split; job.hand_off = hand_off; job.regen_export = regen_export;
    job.profile = prm->profile != 0; job.in_order = c->tuning.in_order != 0;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        job.capturing = !(hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone);
        (void)hipGetLastError();
    }
    job.grid = grid; job.regen_grid = regen_launch_grid; job.cont_grid = cont_grid;
    job.regen_capacity = regen_capacity;
    job.fixed_grid = prm->workgroups != 0 || c->tuning.regen_workgroups != 0;
    job.counter_words = kCountStride;
    job.xchg_need = job.split ? (size_t)regen_launch_grid * kWavesPerBlock * ptk::kRegenSplitF4PerWave + 1024 : 0;   // + slack: a violated stack invariant (reported) stays inside the buffer

    // ---- PREPARE: every buffer the render will touch, before anything is enqueued
    int rc;
    if (prm->accel && (rc = ensure_bvh(c))) return rc;
    for (int k = 0; k < 4; ++k)
        if ((rc = c->queue[k].ensure(q_slots))) return rc;
    if (prm->accel && ((rc = c->bvh_aux.ensure(q_slots)) || (rc = c->bvh_sray[0].ensure(q_slots)) ||
                       (rc = c->bvh_sray[1].ensure(q_slots))))
        return rc;
    {
        // The buffer sets the render rotates through (with lanes: up to three sample buffers of 12 B per path in flight -- 3 x 3.2 GB
        // for the reference's 400 x 400 x 3000 job at the default cap).  All of them now: an allocation in the middle of the batch
        // loop would drain the pipeline the lanes keep full, and fail with half of the render enqueued.  If the device cannot hold
        // them, the render falls back to ONE set and launches in order -- slower, same film -- instead of failing.
        uint32_t sets[4], n_sets = 0;
        ptsched::sets_of(c->sched, job, sets, &n_sets);
        for (uint32_t k = 0; k < n_sets; ++k) {
            rc = sample_buffer(c, sets[k]).ensure(n_paths_max);
            if (rc == PT_ERR_OOM && ptsched::takes_lanes(job)) {
                job.in_order = 1;
                ptsched::sets_of(c->sched, job, sets, &n_sets);
                k = (uint32_t)-1;                       // start over with the one set of the in-order form
                continue;
            }
            if (rc) return rc;
        }
    }
    const bool overlap = n_batches > 1 && !regen;
    if (overlap && hand_off)
        for (int k = 0; k < 4; ++k)
            if ((rc = c->cqueue[k].ensure(q_slots_cont))) return rc;
    if (overlap && hand_off && prm->accel && ((rc = c->caux.ensure(q_slots_cont)) || (rc = c->csray[0].ensure(q_slots_cont)) ||
                                              (rc = c->csray[1].ensure(q_slots_cont))))
        return rc;
    if (!c->ovf_count.p) {             // [render statistics | launch counters of buffer set 0 | 1 | ...], zero from the start
        if ((rc = c->ovf_count.ensure(kStatsWords + kSets * kCountStride))) return rc;
        // (on the caller's stream and waited for: a plain hipMemset runs on the legacy default stream, which the context's
        // non-blocking streams do not wait for -- the first render's counts raced with it, caught by pt_sync's own sample check)
        HIP_TRY(hipMemsetAsync(c->ovf_count.p, 0, (kStatsWords + kSets * kCountStride) * sizeof(uint32_t), st));
        HIP_TRY(hipStreamSynchronize(st));
        c->sched.stats_clean = 1;
        for (int k = 0; k < kSets; ++k) c->sched.counters_clean[k] = 1;
    }
    if (hand_off)
        for (int par = 0; par < (overlap ? 2 : 1); ++par)
            for (int k = 0; k < 4; ++k)
                if ((rc = c->ovf[par][0][k].ensure(ovf_slots))) return rc;
    if ((n_batches > 1 || fs.load || fs.store) && (rc = c->film.ensure((size_t)np * 3))) return rc;

    ptk::BounceArgs a{};
    unsigned long long* const d_stats = reinterpret_cast<unsigned long long*>(c->ovf_count.p);   // hipMalloc alignment: fine for u64
    a.stats = d_stats;
    if (!list) {   // tile row -> image row without a table (ptk::TileMap)
        const uint32_t br = prm->band_rows ? prm->band_rows : cam->height;
        a.tile.band_rows = br;
        a.tile.band_magic = br > 1 ? (uint32_t)(((1ull << 32) + br - 1) / br) : 0u;
        a.tile.band_stride = br * band_count;
        a.tile.band_first = prm->band_index * br;
    }
    a.sc = view_for(c, prm->exact_math);
    for (int k = 0; k < 3; ++k) {
        a.cam.origin[k] = (float)cam->origin[k]; a.cam.lower_left[k] = (float)cam->lower_left[k];
        a.cam.horizontal[k] = (float)cam->horizontal[k]; a.cam.vertical[k] = (float)cam->vertical[k];
    }
    a.cam.width = cam->width; a.cam.height = cam->height;
    a.pixels = list ? list->d_pixels : nullptr;
    a.film_w = list ? 65536u : cam->width;
    a.np = np;
    auto magic = [](uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d); };
    a.film_w_magic = magic(a.film_w);
    a.np_magic = magic(np);
    a.min_depth = prm->min_depth; a.max_depth = prm->max_depth;
    a.t_min = (float)prm->t_min;
    a.integrator = prm->integrator;
    a.accel = prm->accel;
    a.bvh_refill = c->tuning.bvh_refill ? std::min(64u, c->tuning.bvh_refill) : ptk::kRefillBelow;
    a.bvh_leaf = c->tuning.bvh_leaf ? c->tuning.bvh_leaf : ptk::kLeafBatch;

    // ---- PLAN (pure; on a copy of the state)
    ptsched::State next = c->sched;
#ifndef PT_SCHED_FAULTS
#define PT_SCHED_FAULTS 0        // measurement / demonstration builds only: ptsched::Faults switched on in the product path
#endif
    const ptsched::Plan plan = ptsched::plan(next, job, PT_SCHED_FAULTS);
    if (plan.profile) {
        uint32_t top = 0;
        for (const ptsched::Op& o : plan.ops) if (o.kind == ptsched::kOpRecord && o.event == ptsched::kEvPool) top = std::max(top, o.pool + 1u);
        if ((rc = ensure_events(c, top))) return rc;
    }

    // ---- EXECUTE
    std::vector<uint32_t> primary_events;
    uint32_t pool_begin = 0;
    auto exec = [&](const ptsched::Op& o, size_t index) -> int {
        using namespace ptsched;
        if (c->debug_fail_at >= 0 && (size_t)c->debug_fail_at == index) {
            c->debug_fail_at = -1;
            return fail(PT_ERR_HIP, "injected failure at stream operation %zu of the render (pt_debug_fail_after)", index);
        }
        hipStream_t s = sched_stream(c, st, o.stream);
        switch (o.kind) {
        case kOpHostSync:
            for (int k = 0; k < kLanes; ++k) HIP_TRY(hipStreamSynchronize(c->lane_stream[k]));
            HIP_TRY(hipStreamSynchronize(st));
            if (int r2 = c->xchg.ensure((size_t)next.xchg_stride * kLanes)) return r2;
            return PT_OK;
        case kOpMemsetStats:
            HIP_TRY(hipMemsetAsync(c->ovf_count.p, 0, kStatsWords * sizeof(uint32_t), s));
            return PT_OK;
        case kOpMemsetCounters:
            HIP_TRY(hipMemsetAsync(c->ovf_count.p + kStatsWords + kCountStride * o.set, 0, kCountStride * sizeof(uint32_t), s));
            return PT_OK;
        case kOpRecord:
            if (o.event == kEvPool && !(o.pool & 1u)) pool_begin = o.pool;
            HIP_TRY(hipEventRecord(sched_event(c, o), s));
            return PT_OK;
        case kOpWait:
            HIP_TRY(hipStreamWaitEvent(s, sched_event(c, o), 0));
            return PT_OK;
        case kOpPost:
            __atomic_store_n(c->h_posted, o.seq, __ATOMIC_RELEASE);      // before the launch is handed to the device
            return PT_OK;
        case kOpLaunch: {
            const uint32_t s0 = o.batch * nb_max, nb = std::min(nb_max, spp - s0);
            uint32_t* const d_count = c->ovf_count.p + kStatsWords + kCountStride * o.set;   // [0] leftovers handed over, [64 ...] chunk counters of k_paths_regen
            const bool own = o.own_queue != 0;
            a.s_base = prm->spp_offset + s0;
            a.lsamp = sample_buffer(c, o.set).p;
            // Level 0 traces the batch's paths (every bounce, see k_paths); in a large batch its waves hand their sparse
            // tails to the overflow queue, which level 1 -- same kernel, fixed grid, count read on the device -- finishes.
            // (a regenerating launch whose waves run dry themselves leaves nothing for a continuation launch)
            a.n_first = np * nb;
            a.n_first_dev = nullptr;
            a.seg_cap = ((((a.n_first + 63u) / 64u) + nw - 1) / nw) * 64u;
            a.src_mode = inject ? 1u : 0u;
            a.export_below = hand_off ? (small_scene ? export_small : ptk::kBlock) : 1u;
            if (o.level > 0) {
                a.n_first = 0; a.n_first_dev = d_count;
                a.seg_cap = seg_cap_cont;
                a.src_mode = 1u;
                a.export_below = 1u;
            }
            a.aux = own ? c->caux.p : c->bvh_aux.p;
            a.sray0 = own ? c->csray[0].p : c->bvh_sray[0].p;
            a.sray1 = own ? c->csray[1].p : c->bvh_sray[1].p;
            for (int k = 0; k < 4; ++k) {
                a.q.q[k] = own ? c->cqueue[k].p : c->queue[k].p;
                a.ovf_out.q[k] = hand_off ? c->ovf[o.ovf_par][0][k].p : nullptr;
                a.ovf_in.q[k] = inject ? const_cast<float4*>(list->inject[k]) : (hand_off ? c->ovf[o.ovf_par][0][k].p : nullptr);
            }
            a.ovf_out_count = d_count;
            a.debug_tag = o.set;
            a.chunk_counter = nullptr;
            a.xchg = nullptr;
            a.posted = nullptr; a.seq = 0; a.core_blocks = 0;
            a.regen_static = 0;
            if (o.flags & kLaunchRegen) {
                a.chunk_counter = d_count + ptk::kRegenCounterStride;
                if (o.flags & kLaunchSplit) a.xchg = c->xchg.p + o.xchg_off;
                a.export_below = regen_export;
                // the first kRegenStatic16 / 16 of the chunks are dealt statically (none with lanes: a workgroup of this launch that
                // only finds room when the previous launch's last waves end would carry its dealt share as a serial tail)
                const uint32_t nwr = o.grid * kWavesPerBlock, nch = (a.n_first + 63u) / 64u;
                if (o.flags & kLaunchStaticDeal) a.regen_static = (uint32_t)(((uint64_t)nch * kRegenStatic16 / 16) / nwr) * nwr;
                if (o.core) { a.posted = c->d_posted; a.seq = o.seq; a.core_blocks = o.core; }
            }
            if (prm->exact_math) ptk::launch_paths_exact(a, o.grid, s);
            else ptk::launch_paths_fast(a, o.grid, s);
            HIP_TRY(hipGetLastError());
            if ((o.flags & kLaunchPrimary) && plan.profile) primary_events.push_back(pool_begin / 2u);
            return PT_OK;
        }
        case kOpResolve: {
            const uint32_t s0 = o.batch * nb_max, nb = std::min(nb_max, spp - s0);
            ptk::ResolveArgs r{};
            r.lsamp = sample_buffer(c, o.set).p;
            r.film = c->film.p;
            r.out_linear = d_linear;
            r.out_rgba = d_rgba;
            r.np = np; r.nb = nb;
            r.load_film = o.batch > 0 || fs.load;
            r.store_film = o.batch + 1 < n_batches || fs.store;
            r.finalize = o.batch + 1 == n_batches;
            r.spp_div = fs.div ? fs.div : spp;
            r.out_packed = d_packed;
            if (o.zero_words) { r.zero_words = c->ovf_count.p + kStatsWords + kCountStride * o.set; r.n_zero = o.zero_words; }
            ptk::launch_resolve(r, s);
            HIP_TRY(hipGetLastError());
            return PT_OK;
        }
        default:
            return fail(PT_ERR_HIP, "internal: unknown stream operation %u", o.kind);
        }
    };
    for (size_t i = 0; i < plan.ops.size(); ++i)
        if ((rc = exec(plan.ops[i], i))) {
            const std::string keep = g_err;
            sched_recover(c, st, next);
            g_err = keep;
            return rc;
        }

    c->debug_fail_at = -1;              // (the hook is for ONE render)

    // ---- COMMIT
    if (!plan.accumulate) begin_period();
    c->sched = next;
    // (the device-side statistics are read when they are collected -- pt_sync --, not copied back per render)
    if (job.capturing) {
        // nothing ran, and the graph may be replayed any number of times: the device will have counted a multiple of these
        const uint64_t n = (uint64_t)np * spp;
        uint64_t x = c->capture_gcd, y = n;
        while (y) { const uint64_t t = x % y; x = y; y = t; }
        c->capture_gcd = x;
    } else {
        c->expected_samples += (uint64_t)np * spp;
    }
    c->stats.bounce_launches += plan.launches;
    c->stats.batches += n_batches;
    c->stats.primary_launches += plan.primary_launches;
    c->primary_events.insert(c->primary_events.end(), primary_events.begin(), primary_events.end());
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_render_device(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, float* d_linear, uint8_t* d_rgba) {
    return render_impl(c, cam, prm, FilmState{}, nullptr, d_linear, d_rgba);
}
// The same render with the film written as ONE 16-byte record per tile pixel (12 B linear RGB + 4 B RGBA8): the send buffer of
// the multi-GPU gather, straight from the resolve (what pt_film_pack makes of the two planes in a launch of its own).
int pt_render_device_packed(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, void* d_packed) {
    if (!d_packed) return fail(PT_ERR_INVALID_ARG, "pt_render_device_packed: the output buffer is null");
    if ((uintptr_t)d_packed % 16u) return fail(PT_ERR_INVALID_ARG, "pt_render_device_packed: the output buffer must be 16-byte aligned");
    return render_impl(c, cam, prm, FilmState{}, nullptr, nullptr, nullptr, d_packed);
}

}  // extern "C"
hipStream_t pt_internal_stream(PtContext* c) { return c->stream; }
extern "C" {

int pt_sync(PtContext* c) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const bool collect = c->sched.stats_pending != 0;
    bool cleared = false;
    int rc = PT_OK;
    if (collect) {
        // the device-side statistics of the renders since the last collection: read now (the stream is idle) and cleared
        // for the next ones, so that no render carries a copy or a fill of them in its stream
        HIP_TRY(hipMemcpy(c->h_dstats, c->ovf_count.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        cleared = hipMemsetAsync(c->ovf_count.p, 0, kStatsWords * sizeof(uint32_t), c->stream) == hipSuccess;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->stats.total_ms = ms;
        (void)hipGetLastError();          // (events recorded into a graph have no time)
        double kms = 0.0;
        for (uint32_t b = 0; b < c->sched.profiled; ++b)
            if (hipEventElapsedTime(&ms, c->ev_pool[2 * b], c->ev_pool[2 * b + 1]) == hipSuccess) kms += ms;
        c->stats.bounce_kernel_ms = kms;
        c->stats.shadow_rays = c->h_dstats[0];
        c->stats.vertices = c->h_dstats[1];
        c->stats.primary_vertices = c->h_dstats[3];
        double pms = 0.0;
        if (c->sched.profiled)
            for (uint32_t li : c->primary_events)
                if (hipEventElapsedTime(&ms, c->ev_pool[2 * li], c->ev_pool[2 * li + 1]) == hipSuccess) pms += ms;
        c->stats.primary_kernel_ms = pms;
        c->stats.max_depth_reached = (uint32_t)c->h_dstats[2];
        // PtStats.samples is what the DEVICE counted: a path adds one where its radiance is written to the sample buffer
        // (stats[4]).  The host's own arithmetic -- tile pixels x spp of every render enqueued -- is the expectation; a render
        // that lost or repeated work (a scheduling race, a counter cleared under a running launch) shows up here, not in a film
        // somebody has to look at.  Renders captured into graphs ran zero or more times: multiples of their size are accepted.
        const uint64_t dev = c->h_dstats[4], exp = c->expected_samples;
        c->stats.samples = dev;
        c->stats.samples_expected = exp;
        bool ok = dev == exp;
        if (!ok && c->capture_gcd) ok = dev >= exp && (dev - exp) % c->capture_gcd == 0;
#if defined(PT_COUNT_FINISHED) && !PT_COUNT_FINISHED
        ok = true;                        // measurement build whose kernels do not count (A/B of the counting's cost only)
#endif
        c->expected_samples = 0;
        if (c->h_dstats[7] != 0)     // a kernel found one of its own invariants violated: the film is not to be trusted
            rc = fail(PT_ERR_HIP, "internal: the exchange stacks of k_paths_regen_split overflowed (please report; PtTuning.level0_form = 1 avoids the kernel)");
        else if (!ok)
            rc = fail(PT_ERR_HIP, "internal: the device finished %llu samples where the renders since the last collection asked for %llu "
                                  "(please report; the films of these renders are not to be trusted)", (unsigned long long)dev, (unsigned long long)exp);
    }
    ptsched::on_sync(c->sched, collect, cleared);      // everything enqueued so far is complete: the buffer sets and lanes start over
    return rc;
}

// Test hook: the n-th stream operation (0-based) of the NEXT render on this context fails as if its HIP call had; < 0: none.
int pt_debug_fail_after(PtContext* c, int64_t n) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "null context");
    c->debug_fail_at = n;
    return PT_OK;
}

// Debug: what one linear scan of the uploaded scene tests -- spheres, single triangles, triangle PAIRS (two consecutive
// triangles with the same v0 and plane normal share determinant, t and hit point: tripair_test).
int pt_debug_scan_layout(PtContext* c, uint32_t* n_spheres, uint32_t* n_triangles, uint32_t* n_pairs) {
    if (!c || !c->has_scene) return fail(PT_ERR_INVALID_ARG, "no scene uploaded");
    if (n_spheres) *n_spheres = c->scan_counts[0];
    if (n_triangles) *n_triangles = c->scan_counts[1];
    if (n_pairs) *n_pairs = c->scan_counts[2];
    return PT_OK;
}

// Debug: the 16 raw device-side statistics words as last collected (pt_sync / pt_get_stats).  [0] shadow rays [1] vertices
// [2] deepest vertex [3] level-0 vertices [7] internal error flag; [8..12] only in a PT_DRAIN_TIMING measurement build.
int pt_debug_raw_stats(PtContext* c, uint64_t* out16) {
    if (!c || !out16) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (!c->h_dstats) return fail(PT_ERR_INVALID_ARG, "no statistics yet");
    for (int k = 0; k < 16; ++k) out16[k] = c->h_dstats[k];
    return PT_OK;
}

#ifdef PT_DRAIN_TIMING
// measurement build only: the per-wave records k_paths_regen left in the hand-over queue (4 words per wave)
int pt_debug_wave_dump(PtContext* c, uint32_t* out, uint32_t n_waves) {      // n_waves | lane << 31
    const uint32_t plane = n_waves >> 30; n_waves &= 0x3FFFFFFFu;      // plane = buffer set of the launch (BounceArgs.debug_tag)
    if (!c || !out || !c->ovf[0][0][plane].p || n_waves > c->ovf[0][0][plane].cap) return fail(PT_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipMemcpy(out, c->ovf[0][0][plane].p, (size_t)n_waves * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}
#endif

int pt_get_stats(PtContext* c, PtStats* out) {
    if (!c || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    int rc = pt_sync(c);
    if (rc) return rc;
    *out = c->stats;
    return PT_OK;
}

}  // extern "C"

namespace {

int debug_hit_impl(PtContext* c, const double* rays, uint32_t n, double t_min, double t_max, uint32_t exact_math,
                   uint32_t accel, int32_t* out_id, float* out_t, float* out_rec) {
    if (!c || !rays || !out_id) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (!c->has_scene) return fail(PT_ERR_INVALID_ARG, "no scene uploaded");
    if (accel > PT_ACCEL_AUTO) return fail(PT_ERR_INVALID_ARG, "unknown accel %u", accel);
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (accel == PT_ACCEL_AUTO) {
        const std::string keep = g_err;
        accel = (c->auto_bvh && !c->bvh_refused && !c->bvh_failed && ensure_bvh(c) == PT_OK) ? PT_ACCEL_BVH : PT_ACCEL_LINEAR;
        if (!accel) g_err = keep;
    }
    if (accel) { int rb = ensure_bvh(c); if (rb) return rb; }
    std::vector<float> r6(6 * (size_t)n);
    for (size_t i = 0; i < r6.size(); ++i) r6[i] = (float)rays[i];
    DevBuf<float> d_r, d_t, d_rec;
    DevBuf<int32_t> d_id;
    DevBuf<float4> d_scratch;
    struct Release { DevBuf<float>&a, &b, &e; DevBuf<int32_t>& c; DevBuf<float4>& d; ~Release() { a.release(); b.release(); e.release(); c.release(); d.release(); } }
        guard{d_r, d_t, d_rec, d_id, d_scratch};
    int rc;
    if ((rc = d_r.ensure(r6.size())) || (rc = d_id.ensure(n)) || (rc = d_t.ensure(n))) return rc;
    if (out_rec && (rc = d_rec.ensure(8 * (size_t)n))) return rc;
    if (accel && (rc = d_scratch.ensure(3 * (size_t)n))) return rc;
    HIP_TRY(hipMemcpy(d_r.p, r6.data(), r6.size() * sizeof(float), hipMemcpyHostToDevice));
    if (exact_math) ptk::launch_debug_hit_exact(view_for(c, 1), accel, d_r.p, n, (float)t_min, (float)t_max, d_scratch.p, d_id.p, d_t.p, d_rec.p, c->stream);
    else ptk::launch_debug_hit_fast(c->view, accel, d_r.p, n, (float)t_min, (float)t_max, d_scratch.p, d_id.p, d_t.p, d_rec.p, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_id, d_id.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_t) HIP_TRY(hipMemcpy(out_t, d_t.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (out_rec) HIP_TRY(hipMemcpy(out_rec, d_rec.p, 8 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

// One launch of k_debug_fn: in = n * in_stride floats (host), words = n * 4 raw words or null, out = n * out_stride floats.
int debug_fn(PtContext* c, uint32_t op, uint32_t obj, const std::vector<float>& in, uint32_t in_stride, const uint32_t* words,
             uint32_t n, uint32_t out_stride, uint32_t exact_math, const PtCamera* cam, float* out) {
    if (!c || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (!c->has_scene) return fail(PT_ERR_INVALID_ARG, "no scene uploaded");
    if (op != ptk::kFnLightPoint && op != ptk::kFnCameraRay && obj >= c->view.n_objs)
        return fail(PT_ERR_INVALID_ARG, "object %u out of range (%u objects)", obj, c->view.n_objs);
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->fn_in.ensure(in.size() + 1)) || (rc = c->fn_out.ensure((size_t)n * out_stride)) ||
        (rc = c->fn_words.ensure(4 * (size_t)n)))
        return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!in.empty()) HIP_TRY(hipMemcpy(c->fn_in.p, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice));
    if (words) HIP_TRY(hipMemcpy(c->fn_words.p, words, 4 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    ptk::DebugFnArgs a{};
    a.sc = view_for(c, exact_math);
    if (cam) {
        for (int k = 0; k < 3; ++k) {
            a.cam.origin[k] = (float)cam->origin[k]; a.cam.lower_left[k] = (float)cam->lower_left[k];
            a.cam.horizontal[k] = (float)cam->horizontal[k]; a.cam.vertical[k] = (float)cam->vertical[k];
        }
        a.cam.width = cam->width; a.cam.height = cam->height;
    }
    a.op = op; a.obj = obj; a.n = n; a.in_stride = in_stride; a.out_stride = out_stride;
    a.in = c->fn_in.p; a.words = words ? c->fn_words.p : nullptr; a.out = c->fn_out.p;
    if (exact_math) ptk::launch_debug_fn_exact(a, c->stream); else ptk::launch_debug_fn_fast(a, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->fn_out.p, (size_t)n * out_stride * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}
std::vector<float> to_f32(const double* p, size_t n) {
    std::vector<float> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (float)p[i];
    return v;
}

}  // namespace

extern "C" {

int pt_debug_hit_scene(PtContext* c, const double* rays, uint32_t n, double t_min, double t_max, uint32_t exact_math,
                       uint32_t accel, int32_t* out_id, float* out_t) {
    if (!out_t) return fail(PT_ERR_INVALID_ARG, "null argument");
    return debug_hit_impl(c, rays, n, t_min, t_max, exact_math, accel, out_id, out_t, nullptr);
}
int pt_debug_hit_records(PtContext* c, const double* rays, uint32_t n, double t_min, double t_max, uint32_t exact_math,
                         uint32_t accel, int32_t* out_id, float* out_rec) {
    if (!out_rec) return fail(PT_ERR_INVALID_ARG, "null argument");
    return debug_hit_impl(c, rays, n, t_min, t_max, exact_math, accel, out_id, nullptr, out_rec);
}

int pt_debug_bsdf_eval(PtContext* c, uint32_t obj, const double* in10, uint32_t n, uint32_t exact_math, float* out4) {
    if (!in10 && n) return fail(PT_ERR_INVALID_ARG, "null argument");
    return debug_fn(c, ptk::kFnBsdfEval, obj, to_f32(in10, 10 * (size_t)n), 10, nullptr, n, 4, exact_math, nullptr, out4);
}
int pt_debug_bsdf_sample(PtContext* c, uint32_t obj, const double* in7, const uint32_t* words4, uint32_t n,
                         uint32_t exact_math, float* out8) {
    if ((!in7 || !words4) && n) return fail(PT_ERR_INVALID_ARG, "null argument");
    return debug_fn(c, ptk::kFnBsdfSample, obj, to_f32(in7, 7 * (size_t)n), 7, words4, n, 8, exact_math, nullptr, out8);
}
int pt_debug_shape_sample(PtContext* c, uint32_t obj, const double* from3, const double* target3, const double* r12,
                          uint32_t n, uint32_t exact_math, float* out8) {
    if ((!from3 || (!target3 && !r12)) && n) return fail(PT_ERR_INVALID_ARG, "null argument");
    std::vector<float> in(9 * (size_t)n, 0.0f);
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) in[9 * i + k] = (float)from3[3 * i + k];
        if (target3) { for (int k = 0; k < 3; ++k) in[9 * i + 3 + k] = (float)target3[3 * i + k]; in[9 * i + 8] = 1.0f; }
        else { in[9 * i + 6] = (float)r12[2 * i]; in[9 * i + 7] = (float)r12[2 * i + 1]; }
    }
    return debug_fn(c, ptk::kFnShapeSample, obj, in, 9, nullptr, n, 8, exact_math, nullptr, out8);
}
int pt_debug_light_point(PtContext* c, const double* from3, const uint32_t* words4, uint32_t n, uint32_t exact_math,
                         float* out8) {
    if ((!from3 || !words4) && n) return fail(PT_ERR_INVALID_ARG, "null argument");
    return debug_fn(c, ptk::kFnLightPoint, 0, to_f32(from3, 3 * (size_t)n), 3, words4, n, 8, exact_math, nullptr, out8);
}
int pt_debug_camera_rays(PtContext* c, const PtCamera* cam, const uint32_t* xys, uint32_t n, uint32_t exact_math, float* out8) {
    if ((!cam || !xys) && n) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (cam && (cam->width < 2 || cam->height < 2)) return fail(PT_ERR_INVALID_ARG, "camera %ux%u: width and height must be >= 2", cam->width, cam->height);
    std::vector<uint32_t> w(4 * (size_t)n, 0u);
    for (size_t i = 0; i < n; ++i) { w[4 * i] = xys[3 * i]; w[4 * i + 1] = xys[3 * i + 1]; w[4 * i + 2] = xys[3 * i + 2]; }
    return debug_fn(c, ptk::kFnCameraRay, 0, std::vector<float>(), 1, w.data(), n, 8, exact_math, cam, out8);
}

int pt_debug_bvh_check(const PtObject* objs, uint32_t n, uint32_t* depth, uint32_t* n_nodes, uint32_t* n_leaf_slots) {
    if (!objs && n) return fail(PT_ERR_INVALID_ARG, "pt_debug_bvh_check: null objects");
    if (n >= (1u << 28)) return fail(PT_ERR_UNSUPPORTED, "accel: %u objects exceed the 2^28 leaf slots", n);
    std::vector<float4> shape(3 * (size_t)n + 1), scan(3 * (size_t)n + 1);
    std::vector<uint32_t> tag(n + 1);
    for (uint32_t i = 0; i < n; ++i) {
        if (objs[i].shape_tag > PT_SHAPE_TRIANGLE) return fail(PT_ERR_INVALID_ARG, "object %u: bad shape_tag %u", i, objs[i].shape_tag);
        int ns = 0;
        shape_records(objs[i], &shape[3 * (size_t)i], &scan[3 * (size_t)i], &ns);
        tag[i] = objs[i].shape_tag;
    }
    const ptbvh::Built b = ptbvh::build(shape.data(), tag.data(), n);
    if (depth) *depth = b.depth;
    if (n_nodes) *n_nodes = (uint32_t)b.wide.size();
    if (n_leaf_slots) *n_leaf_slots = b.leaf_prims;          // slots that hold a primitive (leaves are padded to multiples of 4 slots)
    if (b.non_finite) return fail(PT_ERR_UNSUPPORTED, "accel: %u object(s) with a NaN/inf coordinate", b.non_finite);
    if (b.depth + 2u > ptbvh::kStackDepth || b.stack_need > ptbvh::kStackDepth)
        return fail(PT_ERR_UNSUPPORTED, "BVH (depth %u, stack need %u) exceeds the traversal stack", b.depth, b.stack_need);
    if (b.leaf_prims != n || b.leaf_rec.size() != 3 * b.leaf_ids.size() || b.leaf_lead.size() != b.leaf_ids.size() || b.leaf_ids.size() % 4u != 0u)
        return fail(PT_ERR_UNSUPPORTED, "%u primitives in %zu leaf slots for %u objects", b.leaf_prims, b.leaf_ids.size(), n);
    if (n == 0) return b.root == ptbvh::kDone ? PT_OK : fail(PT_ERR_UNSUPPORTED, "empty scene: root is not the sentinel");
    // boxes of the primitives in f64 from the same f32 records the device tests
    auto prim_box = [&](uint32_t o, double lo[3], double hi[3]) {
        const float4 r0 = shape[3 * (size_t)o], r1 = shape[3 * (size_t)o + 1], r2 = shape[3 * (size_t)o + 2];
        if (tag[o] == PT_SHAPE_SPHERE) {
            const double r = std::sqrt((double)(r0.w * r0.w));
            const double c[3] = {r0.x, r0.y, r0.z};
            for (int k = 0; k < 3; ++k) { lo[k] = c[k] - r; hi[k] = c[k] + r; }
        } else {
            const double v0[3] = {r0.x, r0.y, r0.z}, e1[3] = {r1.x, r1.y, r1.z}, e2[3] = {r2.x, r2.y, r2.z};
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(v0[k], std::min(v0[k] + e1[k], v0[k] + e2[k]));
                hi[k] = std::max(v0[k], std::max(v0[k] + e1[k], v0[k] + e2[k]));
            }
        }
    };
    std::vector<uint8_t> seen(n, 0);
    std::string err;
    // returns the exact bounds of the subtree and the stack entries a traversal can need below it (sum over the deepest
    // path of children - 1); checks the bounds against the box the parent stores for the subtree
    struct Walker {
        const ptbvh::Built& b; const std::vector<float4>& scan; const std::vector<uint32_t>& tag; std::vector<uint8_t>& seen;
        decltype(prim_box)& pbox; std::string& err; uint32_t n;
        bool walk(uint32_t code, double lo[3], double hi[3], uint32_t* need) {
            for (int k = 0; k < 3; ++k) { lo[k] = 1e300; hi[k] = -1e300; }
            *need = 0;
            if (code == ptbvh::kDone) { err = "sentinel inside the tree"; return false; }
            if (code & ptbvh::kLeafBit) {
                const uint32_t first = code & 0x0FFFFFFFu, cnt = ((code >> 28) & 7u) + 1u;
                if (cnt > ptbvh::kMaxLeaf || (size_t)first + cnt > b.leaf_ids.size() || first % 4u != 0u) { err = "leaf range out of bounds or not aligned to 4 slots"; return false; }
                for (uint32_t i = first; i < first + cnt; ++i) {
                    const uint32_t w = b.leaf_ids[i], o = w & 0x7FFFFFFFu;
                    if (o >= n || seen[o]) { err = "object missing or in two leaves"; return false; }
                    seen[o] = 1;
                    if (((w >> 31) != 0) != (tag[o] == PT_SHAPE_TRIANGLE)) { err = "leaf tag bit differs from the object's shape"; return false; }
                    const int ns = tag[o] == PT_SHAPE_TRIANGLE ? 3 : 1;
                    if (std::memcmp(&b.leaf_rec[3 * (size_t)i], &scan[3 * (size_t)o], ns * sizeof(float4)) != 0) { err = "leaf record differs from the scan record"; return false; }
                    if (std::memcmp(&b.leaf_lead[i], &scan[3 * (size_t)o], sizeof(float4)) != 0) { err = "lead record differs from the scan record"; return false; }
                    double pl[3], ph[3];
                    pbox(o, pl, ph);
                    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], pl[k]); hi[k] = std::max(hi[k], ph[k]); }
                }
                return true;
            }
            if ((size_t)code >= b.wide.size()) { err = "node index out of bounds"; return false; }
            const ptbvh::WideNode& wn = b.wide[code];
            if (wn.n < 2 || wn.n > ptbvh::kWidth) { err = "node with fewer than 2 or more than 4 children"; return false; }
            // what the device traverses: the boxes decoded from the 16-bit grid; they must contain the f32 boxes
            if (4 * (size_t)code + 3 >= b.qnodes.size()) { err = "quantised node index out of bounds"; return false; }
            const uint4 qa = b.qnodes[4 * (size_t)code], qb = b.qnodes[4 * (size_t)code + 1], qc = b.qnodes[4 * (size_t)code + 2],
                        qd = b.qnodes[4 * (size_t)code + 3];
            const uint32_t qcode[4] = {qd.x, qd.y, qd.z, qd.w};
            const uint32_t qw[4][3] = {{qa.x, qa.y, qa.z}, {qa.w, qb.x, qb.y}, {qb.z, qb.w, qc.x}, {qc.y, qc.z, qc.w}};
            uint32_t need_below = 0;
            for (uint32_t c = 0; c < ptbvh::kWidth; ++c) {
                if (qcode[c] != wn.code[c]) { err = "quantised node carries other child codes"; return false; }
                if (c >= wn.n) {
                    if (wn.code[c] != ptbvh::kDone) { err = "unused child slot without the sentinel code"; return false; }
                    continue;
                }
                const uint32_t q[6] = {qw[c][0] & 0xFFFFu, qw[c][0] >> 16, qw[c][1] & 0xFFFFu, qw[c][1] >> 16, qw[c][2] & 0xFFFFu, qw[c][2] >> 16};
                float blo[3], bhi[3];
                for (int k = 0; k < 3; ++k) {
                    blo[k] = std::fmaf((float)q[k], b.grid_cell[k], b.grid_min[k]);
                    bhi[k] = std::fmaf((float)q[3 + k], b.grid_cell[k], b.grid_min[k]);
                    if (!(blo[k] <= wn.lo[c][k] && bhi[k] >= wn.hi[c][k])) { err = "quantised child box does not contain the f32 box"; return false; }
                }
                double cl[3], ch[3];
                uint32_t nd = 0;
                if (!walk(wn.code[c], cl, ch, &nd)) return false;
                need_below = std::max(need_below, nd);
                for (int k = 0; k < 3; ++k) {
                    if (!((double)blo[k] <= cl[k] && (double)bhi[k] >= ch[k])) { err = "child box does not enclose its subtree"; return false; }
                    lo[k] = std::min(lo[k], cl[k]); hi[k] = std::max(hi[k], ch[k]);
                }
            }
            *need = (wn.n - 1u) + need_below;
            return true;
        }
    } w{b, scan, tag, seen, prim_box, err, n};
    uint32_t need = 0;
    double lo[3], hi[3];
    if (!w.walk(b.root, lo, hi, &need)) return fail(PT_ERR_UNSUPPORTED, "BVH invariant: %s", err.c_str());
    for (uint32_t i = 0; i < n; ++i) if (!seen[i]) return fail(PT_ERR_UNSUPPORTED, "BVH invariant: object %u is in no leaf", i);
    if (1u + need != b.stack_need) return fail(PT_ERR_UNSUPPORTED, "BVH invariant: stack need %u reported, %u found", b.stack_need, 1u + need);
    double amax = 0.0;
    for (int k = 0; k < 3; ++k) amax += std::max(std::fabs(lo[k]), std::fabs(hi[k]));
    if (!((double)b.scene_abs >= amax)) return fail(PT_ERR_UNSUPPORTED, "BVH invariant: scene_abs %g below the scene extent %g", (double)b.scene_abs, amax);
    return PT_OK;
}

}  // extern "C"

namespace {

// render_impl into the context's device staging, then over PCIe into the caller's host buffers (blocking)
int render_to_host(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, const FilmState& fs, const ListRender* list,
                   size_t np, float* out_linear, uint8_t* out_rgba) {
    if (np == 0) return render_impl(c, cam, prm, fs, list, nullptr, nullptr);   // validates, renders nothing
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->host_lin.ensure(np * 3))) return rc;
    if (out_rgba && (rc = c->host_rgba.ensure(np * 4))) return rc;
    rc = render_impl(c, cam, prm, fs, list, c->host_lin.p, out_rgba ? c->host_rgba.p : nullptr);
    if (!rc) rc = pt_sync(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_linear, c->host_lin.p, np * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_rgba) HIP_TRY(hipMemcpy(out_rgba, c->host_rgba.p, np * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}
size_t tile_pixels(const PtCamera* cam, const PtRenderParams* prm) {
    return (size_t)pt_tile_rows(cam->height, prm->band_rows, prm->band_index, prm->band_count ? prm->band_count : 1) * cam->width;
}

// the contexts pt_render() keeps between calls (one per device it has been asked to use)
std::mutex g_render_mu;
std::vector<PtContext*> g_render_ctx;

}  // namespace

// pt_shutdown at exit, registered once by whichever one-shot entry (pt_render, pt_render_multi) creates cached state first
void pt_internal_register_atexit(void) {
    static std::once_flag once;
    std::call_once(once, [] { std::atexit(pt_shutdown); });
}

extern "C" {

int pt_render_host(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, float* out_linear, uint8_t* out_rgba) {
    if (!c || !cam || !prm || !out_linear) return fail(PT_ERR_INVALID_ARG, "pt_render_host: null argument");
    return render_to_host(c, cam, prm, FilmState{}, nullptr, tile_pixels(cam, prm), out_linear, out_rgba);
}

// The progressive preview of the reference (main.rs:79-90 redraws World.data every 16 ms while
// the rayon loop fills it): the same render in increments of spp_step samples; after each increment
// the film holds the mean of the samples so far.  The last frame is bit-identical to pt_render_host.
int pt_render_progressive(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, uint32_t spp_step,
                          PtProgressFn fn, void* user, float* out_linear, uint8_t* out_rgba) {
    if (!c || !cam || !prm || !out_linear || !out_rgba)
        return fail(PT_ERR_INVALID_ARG, "pt_render_progressive: null argument");
    if (prm->spp == 0) return fail(PT_ERR_INVALID_ARG, "spp must be > 0");
    if (spp_step == 0) spp_step = prm->spp;
    const size_t np = tile_pixels(cam, prm);
    int rc = PT_OK;
    for (uint32_t done = 0; done < prm->spp && rc == PT_OK;) {
        const uint32_t n = std::min(spp_step, prm->spp - done);
        PtRenderParams p = *prm;
        p.spp = n;
        p.spp_offset = prm->spp_offset + done;
        FilmState fs;                   // the f64 sums of the film stay on the device between increments
        fs.load = done > 0;
        fs.store = done + n < prm->spp;
        fs.div = done + n;
        rc = render_to_host(c, cam, &p, fs, nullptr, np, out_linear, out_rgba);
        done += n;
        if (rc == PT_OK && np != 0 && fn && fn(user, done, prm->spp, out_rgba, out_linear) != 0) break;   // caller asked to stop
    }
    return rc;
}

int pt_render_pixels(PtContext* c, const PtCamera* cam, const PtRenderParams* prm, const uint32_t* xy, uint32_t n,
                     float* out_linear, uint8_t* out_rgba, float* out_samples) {
    if (!c || !cam || !prm || (n && (!xy || !out_linear))) return fail(PT_ERR_INVALID_ARG, "pt_render_pixels: null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (xy[2 * i] >= cam->width || xy[2 * i + 1] >= cam->height)
            return fail(PT_ERR_INVALID_ARG, "pixel %u = (%u, %u) is outside the %ux%u image", i, xy[2 * i], xy[2 * i + 1], cam->width, cam->height);
    ListRender lr;
    if (n) {
        HIP_TRY(hipSetDevice(c->device));
        int rc = c->pixel_list.ensure(n);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));            // an earlier list may still be in use
        HIP_TRY(hipMemcpy(c->pixel_list.p, xy, (size_t)n * sizeof(uint2), hipMemcpyHostToDevice));
        lr.d_pixels = c->pixel_list.p;
    }
    lr.n = n;
    if (out_samples && n) {
        const uint64_t cap = std::min<uint64_t>(prm->max_paths_in_flight ? prm->max_paths_in_flight : kDefaultMaxPaths, 1ull << 30);
        if ((uint64_t)n * prm->spp > cap || prm->spp > 65535u)
            return fail(PT_ERR_UNSUPPORTED, "pt_render_pixels: out_samples needs n * spp = %llu paths in one sample batch (limit %llu, spp < 65536)",
                        (unsigned long long)n * prm->spp, (unsigned long long)cap);
    }
    int rc = render_to_host(c, cam, prm, FilmState{}, &lr, n, out_linear, out_rgba);
    if (rc || !out_samples || !n) return rc;
    // the batch's per-path radiance buffer, index = sample * n + pixel  ->  [pixel][sample][rgb]
    std::vector<ptk::Rgb> ls((size_t)n * prm->spp);
    HIP_TRY(hipMemcpy(ls.data(), c->lsamp.p, ls.size() * sizeof(ptk::Rgb), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t sidx = 0; sidx < prm->spp; ++sidx) {
            const ptk::Rgb v = ls[(size_t)sidx * n + i];
            float* o = out_samples + ((size_t)i * prm->spp + sidx) * 3;
            o[0] = v.r; o[1] = v.g; o[2] = v.b;
        }
    return PT_OK;
}

int pt_ray_color(PtContext* c, const PtRenderParams* prm, const double* rays, const uint32_t* xy, uint32_t n, float* out_rgb) {
    if (!c || !prm || (n && (!rays || !xy || !out_rgb))) return fail(PT_ERR_INVALID_ARG, "pt_ray_color: null argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    // the paths in queue form (pt_kernels.h): depth 0, throughput 1, no radiance yet, film slot = ray index
    std::vector<float4> plane[4];
    for (auto& pl : plane) pl.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float o[3] = {(float)rays[6 * (size_t)i], (float)rays[6 * (size_t)i + 1], (float)rays[6 * (size_t)i + 2]};
        float d[3] = {(float)rays[6 * (size_t)i + 3], (float)rays[6 * (size_t)i + 4], (float)rays[6 * (size_t)i + 5]};
        // Ray::new normalises (camera.rs:10-16): Vector3::normalize in the f32 arithmetic of the device's exact mode
        const float len = std::sqrt(std::fmaf(d[2], d[2], std::fmaf(d[1], d[1], d[0] * d[0])));
        if (len > 0.0f) { const float inv = 1.0f / len; d[0] *= inv; d[1] *= inv; d[2] *= inv; }
        const uint32_t slot = i, sd = 0u;     // (tile_row << 16 | x) of a 65536-wide film; (s_local << 16 | depth)
        float fslot, fsd;
        std::memcpy(&fslot, &slot, 4); std::memcpy(&fsd, &sd, 4);
        plane[0][i] = make_float4(o[0], o[1], o[2], d[0]);
        plane[1][i] = make_float4(d[1], d[2], fslot, fsd);
        plane[2][i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);      // throughput 1, pdf_prev 0
        plane[3][i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);      // no radiance yet, eta_ratio 1 (camera.rs:14)
    }
    int rc;
    if ((rc = c->pixel_list.ensure(n))) return rc;
    for (int k = 0; k < 4; ++k) if ((rc = c->inject[k].ensure(n))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->pixel_list.p, xy, (size_t)n * sizeof(uint2), hipMemcpyHostToDevice));
    ListRender lr;
    lr.d_pixels = c->pixel_list.p; lr.n = n;
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpy(c->inject[k].p, plane[k].data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
        lr.inject[k] = c->inject[k].p;
    }
    PtCamera cam{};                     // no camera rays are generated
    cam.width = cam.height = 2;
    return render_to_host(c, &cam, prm, FilmState{}, &lr, n, out_rgb, nullptr);
}

void pt_shutdown(void) {
    pt_internal_multi_shutdown();
    std::lock_guard<std::mutex> lk(g_render_mu);
    for (PtContext* c : g_render_ctx) pt_context_destroy(c);
    g_render_ctx.clear();
}

int pt_render(const PtCamera* cam, const PtObject* objs, uint32_t n, const PtRenderParams* prm, float* out_linear,
              uint8_t* out_rgba) {
    if (!cam || !prm || !out_linear) return fail(PT_ERR_INVALID_ARG, "pt_render: null argument");
    if (prm->n_devices > 1) {
        std::vector<int> dev(prm->n_devices);
        for (uint32_t i = 0; i < prm->n_devices; ++i) dev[i] = (int)i;
        return pt_render_multi(dev.data(), prm->n_devices, cam, objs, n, prm, out_linear, out_rgba);
    }
    std::lock_guard<std::mutex> lk(g_render_mu);
    int rc;
    if (g_render_ctx.empty()) {
        PtContext* ctx = nullptr;
        if ((rc = pt_context_create(0, &ctx))) return rc;
        g_render_ctx.push_back(ctx);
        pt_internal_register_atexit();
    }
    PtContext* ctx = g_render_ctx[0];
    if ((rc = pt_scene_upload(ctx, objs, n))) return rc;
    return pt_render_host(ctx, cam, prm, out_linear, out_rgba);
}

}  // extern "C"
