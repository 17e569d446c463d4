/*
 * pathtrace_amd.h -- C ABI of the MI355X-native wavefront path tracer.
 *
 * Drop-in boundary for the per-pixel rendering hot path of roxas1533/pathtrace.
 * The reference has no FFI; the seam it offers is the Rust-internal call
 *     world_clone.render_pixel(x, y, &mut rng) -> Color        (src/main.rs:55,
 *                                                                src/world.rs:293)
 * inside the rayon loop at src/main.rs:43-60, plus the two film buffers
 * World.data (RGBA8, src/world.rs:55) and World.luminance_data (linear RGB,
 * src/world.rs:57).  pt_render() replaces that whole loop: one blocking call,
 * internally asynchronous on HIP streams.
 *
 * Only plain pointers, sizes and POD cross this boundary (no trait objects, no
 * torch types).  Reals are f64 on the boundary because the reference's
 * Vector3 is f64 (src/math.rs:4-8); the device computes in f32.
 *
 * Every function returns 0 on success and a non-zero PtStatus on failure;
 * pt_last_error() returns a thread-local message.  Nothing throws or aborts
 * across the boundary (the reference panics instead: src/main.rs:59,66).
 */
#ifndef PATHTRACE_AMD_H
#define PATHTRACE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 4

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,
    PT_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime failure at init   */
    PT_ERR_HIP = 3,         /* a HIP call failed; see pt_last_error()         */
    PT_ERR_OOM = 4,
    PT_ERR_UNSUPPORTED = 5
} PtStatus;

/* ---- scene description -------------------------------------------------- */

/* = the cached fields of Camera (src/camera.rs:27-39).  Fill with
 * pt_camera_new / pt_camera_look_at, or by hand.                            */
typedef struct {
    double origin[3];
    double lower_left[3];
    double horizontal[3];
    double vertical[3];
    uint32_t width, height;
} PtCamera;

/* Shape tags: which `impl Shape` (src/objects/shape.rs:52,160). */
enum { PT_SHAPE_SPHERE = 0, PT_SHAPE_TRIANGLE = 1 };
/* Material tags: which `impl Material` (src/objects/material.rs:85,138,220;
 * src/objects/mirror.rs:178). */
enum { PT_MAT_LAMBERT = 0, PT_MAT_EMISSIVE = 1, PT_MAT_MIRROR = 2, PT_MAT_OREN_NAYAR = 3 };

/* = Object{shape: Box<dyn Shape>, material: Box<dyn Material>}
 * (src/objects/object.rs:9-14) flattened to POD.  Array order == World.objects
 * order (it decides closest-hit ties, src/world.rs:281-287).
 *   shape:  sphere   = center[3], radius            (shape.rs:38-43)
 *           triangle = v0[3], v1[3], v2[3]          (shape.rs:148-152)
 *   mat:    lambert    = albedo[3]                  (material.rs:67-69)
 *           emissive   = emission[3]                (material.rs:126-129)
 *           mirror     = roughness, color[3], metallic, ior   (mirror.rs:5-14)
 *           oren-nayar = albedo[3], roughness       (material.rs:166-174)     */
typedef struct {
    uint32_t shape_tag;
    uint32_t mat_tag;
    double shape[9];
    double mat[6];
} PtObject;

enum { PT_INTEGRATOR_MIS = 0, PT_INTEGRATOR_BRDF_ONLY = 1 };
enum { PT_ACCEL_LINEAR = 0, PT_ACCEL_BVH = 1, PT_ACCEL_AUTO = 2 };

/* Compile-time constants of the reference made runtime parameters
 * (src/world.rs:16-18, src/rendering.rs:6-10).  pt_default_params() fills the
 * reference's values.                                                        */
typedef struct {
    uint32_t spp;            /* SAMPLE_NUM (world.rs:18)                       */
    uint32_t spp_offset;     /* first sample index; film is linear in spp      */
    uint32_t min_depth;      /* MIN_DEPTH = 4 (rendering.rs:6)                 */
    uint32_t max_depth;      /* MAX_DEPTH = 50 (rendering.rs:7)                */
    uint32_t integrator;     /* PT_INTEGRATOR_*; cargo feature (Cargo.toml:6)  */
    double   t_min;          /* 0.001 (rendering.rs:41,64,105)                 */
    /* Row-band tile of the image rendered by this call: the image rows are cut
     * into bands of band_rows rows; this call renders bands b with
     * b % band_count == band_index.  band_count = 1 renders the whole image.
     * Output buffers hold only this tile's rows, ascending y, row-major.      */
    uint32_t band_rows;
    uint32_t band_index;
    uint32_t band_count;
    /* Wavefront sizing: upper bound on paths resident in HBM at once (0 = library default).  It sizes the
     * context's device buffers, which persist between renders: the default is 2^26 paths (76 B of queue + 12 B of
     * sample buffer each, only as many as the job has) and 2^28 where the level-0 launch keeps its paths in
     * registers (EVERY scene of <= 128 objects, whatever its materials: 12 B of sample buffer per path, i.e. up to 3.2 GB per context --
     * per buffer set: renders of several sample batches, or renders enqueued back to back, rotate through up to three of them --
     * for a render of >= 2^28 samples -- the 400 x 400 x 3000 default job included).  A host that shares the GPU
     * sets a smaller bound: the job is then cut into more sample batches, with the same film.                */
    uint64_t max_paths_in_flight;
    uint32_t profile;        /* 1: time every path-kernel launch with HIP events (the launches of consecutive batches / renders
                                then run strictly one after the other; otherwise one may start while its predecessor's last
                                waves run dry) */
    /* Workgroups (256 threads) of the path kernel; every wave owns one private queue
     * segment.  0 = library default (about 1024 paths per wave).  Results do not
     * depend on it.                                                           */
    uint32_t workgroups;
    /* Device arithmetic.  0 (default): hardware reciprocal / square root (1 ulp).  1: IEEE correctly
     * rounded division and sqrt -- every f32 operation is then reproducible on a host CPU, the film
     * is bit-identical to the f32 CPU oracle, and the render is ~1.3x slower.  Both modes meet the
     * FP32 tolerance against the f64 reference arithmetic.                      */
    uint32_t exact_math;
    /* How World::hit_scene (world.rs:270-290) finds the closest hit.  PT_ACCEL_LINEAR: the reference's linear
     * scan over all objects.  PT_ACCEL_BVH: traversal of a BVH over the objects' bounding boxes, built on the
     * host the first time a render asks for it (beyond the reference).  The BVH only prunes the
     * scan: the primitive tests and the winner (smallest t; among equal t the highest object index) are those of
     * the linear scan, and the film is identical -- so which one runs is a performance decision only.
     * PT_ACCEL_AUTO (default): the BVH where it is faster -- scenes of more than 128 objects whose scan costs
     * more than ~512 sphere tests (a triangle counts 2.5) --, the scan otherwise and for scenes the BVH refuses
     * (an object with a NaN/inf coordinate).                                                                */
    uint32_t accel;
    /* pt_render() only: render on the first n_devices HIP devices (pt_render_multi: row bands, one RCCL
     * gather of the film to device 0).  0 or 1: device 0 alone.  The film does not depend on it.               */
    uint32_t n_devices;
} PtRenderParams;

/* Scheduling knobs of a context (pt_context_set_tuning); results never depend on them.  0 = library default.   */
typedef struct {
    uint32_t export_below;   /* a wave hands its queue segment to the continuation launch below this many paths (64) */
    uint32_t bvh_refill;     /* accel = 1: idle lanes take new rays when fewer lanes than this are tracing (44)      */
    uint32_t bvh_leaf;       /* accel = 1: leaf primitives are tested when this many lanes wait at a leaf (20)       */
    uint32_t cont_workgroups;/* workgroups of the continuation launch that finishes the handed-over tails             */
    uint32_t level0_form;    /* level-0 launch of a large batch over a scene in LDS.  0: paths stay in registers and a
                                lane whose path ends takes the batch's next one (k_paths_regen, compiled for the scene's
                                material set); with the Mirror vertices of a wave set aside and shaded 64 at a time
                                (k_paths_regen_split) if some but at most half of the objects are Mirror (the
                                reference's own scene).  1: the queue form; 2: k_paths_regen; 3: k_paths_regen_split  */
    uint32_t regen_workgroups;/* workgroups of that regenerating launch (0: what the device holds at once)             */
    uint32_t in_order;       /* 1: the regenerating launches of consecutive batches / renders run strictly one after the other (as
                                with profile = 1); 0: the next one may start while the last waves of this one run dry            */
} PtTuning;

/* Counters of the renders enqueued on a context since they were last collected (pt_sync / pt_get_stats; pt_scene_upload
 * starts afresh): normally ONE render -- synchronise after each and these are its counters.  A caller that pipelines several
 * pt_render_device calls behind one synchronisation gets their sums; total_ms then runs from the first one's start to the
 * last one's end. */
typedef struct {
    uint64_t samples;          /* camera samples FINISHED, counted on the device where a path's radiance is written to the
                                  sample buffer.  pt_sync / pt_get_stats fail with PT_ERR_HIP when it differs from
                                  samples_expected: a render lost or repeated work and its film is not to be trusted  */
    uint64_t vertices;         /* path vertices processed (iterations of the
                                  per-vertex loop)                 */
    uint64_t shadow_rays;      /* NEE visibility scans                         */
    uint32_t bounce_launches;  /* path-kernel launches (one per sample batch)  */
    uint32_t batches;          /* sample batches                               */
    uint32_t max_depth_reached;
    uint32_t reserved;
    double   bounce_kernel_ms; /* sum of HIP-event durations of the path-kernel
                                  launches (profile=1), else 0                */
    double   total_ms;         /* HIP-event duration of the whole render      */
    /* The dominant kernel on its own: the level-0 launch of every batch (the launch that generates the
     * camera rays and traces them until its waves hand their sparse tails over; the continuation
     * launches that finish those tails are the rest of bounce_kernel_ms).                          */
    uint64_t primary_vertices; /* vertices processed by the level-0 launches                      */
    double   primary_kernel_ms;/* sum of their HIP-event durations (profile=1)                    */
    uint32_t primary_launches;
    uint32_t reserved2;
    uint64_t samples_expected; /* tile pixels * spp of the renders enqueued (host arithmetic); renders captured into a graph
                                  count by replay: `samples` may then exceed this by multiples of the captured renders' size */
} PtStats;

/* ---- helpers ------------------------------------------------------------ */

/* Camera::new (src/camera.rs:50-82): axis aligned, looks down -Z. */
int pt_camera_new(const double origin[3], uint32_t width, uint32_t height,
                  double screen_distance, double fov_degrees, PtCamera* out);
/* Camera::look_at (src/camera.rs:94-130). */
int pt_camera_look_at(const double origin[3], const double target[3], const double up[3],
                      uint32_t width, uint32_t height, double fov_degrees, PtCamera* out);
/* Reference constants: spp 3000, min_depth 4, max_depth 50, MIS, t_min 1e-3. */
void pt_default_params(PtRenderParams* out);
/* Number of image rows in the tile selected by (band_rows, band_index, band_count). */
uint32_t pt_tile_rows(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count);

/* Built-in scenes: 1 = World::new() Cornell box, verbatim
 * src/world.rs:80-211; 2 = 10-sphere diffuse Cornell; 4 = n random spheres
 * (arg = n, 0 -> 10000).  Writes up to cap objects, returns the count in *n
 * (call with objs = NULL to query).                                          */
int pt_builtin_scene(uint32_t id, uint32_t arg, PtObject* objs, uint32_t cap, uint32_t* n);

/* ---- rendering ---------------------------------------------------------- */

typedef struct PtContext PtContext;

/* One context per process per GPU.  device = HIP device ordinal.
 * Threading: a context is NOT internally synchronised -- use it from one thread at a time (the
 * reference calls render_pixel from every rayon worker; this library is called once from the render
 * thread and parallelises on the GPU).  Different contexts may be used from different threads.
 * pt_last_error() is thread-local.  pt_render() serialises its callers on one cached context.   */
int pt_context_create(int device, PtContext** out);
int pt_context_destroy(PtContext* ctx);
/* Run the library's kernels on a caller-owned hipStream_t (e.g. torch's current
 * stream) instead of the context's own (non-blocking) stream.  NULL restores the context's own
 * stream; PT_STREAM_LEGACY_DEFAULT names HIP's legacy default stream, whose handle is also 0
 * (torch's default stream): pass it when the render must be ordered against work queued there.
 * Renders already enqueued on the previous stream stay ahead: the new stream waits for it once.  */
#define PT_STREAM_LEGACY_DEFAULT ((void*)(uintptr_t)1)
int pt_context_set_stream(PtContext* ctx, void* hip_stream);
int pt_context_set_tuning(PtContext* ctx, const PtTuning* tuning);

/* Copy the scene to the device (the reference's World is immutable while
 * rendering: render_pixel(&self), src/world.rs:293).  Lights are detected as
 * in src/world.rs:214-225: objects whose emit() has non-zero length.          */
int pt_scene_upload(PtContext* ctx, const PtObject* objs, uint32_t n_objs);

/* Render the tile into DEVICE buffers (no host transfer, no host synchronisation inside):
 *   d_linear_rgb: float[tile_rows*W*3], mean linear radiance  (= luminance_data,
 *                 src/world.rs:318-319)
 *   d_rgba8:      uint8[tile_rows*W*4], sqrt-gamma + truncation (= World.data /
 *                 draw(), src/world.rs:322-341); may be NULL; 4-byte aligned (a pixel is one 32-bit store).
 * Work is enqueued on the context's stream and the call returns once the last launch is
 * enqueued; the continuation launch that finishes the sparse tails of a sample batch takes
 * its path count from device memory, so the host never waits inside.  Renders of several
 * sample batches also use a second, context-owned stream for those tails; the context's
 * stream waits for it at the end, so everything is complete when that stream is.  Once the
 * buffers of a given size exist (after the first render of that size) the call allocates
 * nothing and can be captured into a hipGraph.  pt_sync() waits.                          */
int pt_render_device(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params,
                     float* d_linear_rgb, uint8_t* d_rgba8);
/* The same render, the film written as ONE 16-byte record per tile pixel -- float linear RGB (12 B) + RGBA8 (4 B), row-major
 * like the planes -- into d_packed (tile_rows*W*16 bytes, 16-byte aligned): the send buffer of the multi-GPU film gather
 * straight from the film resolve, without the two planes and the pt_film_pack launch in between.  pt_film_unpack (below)
 * turns gathered records into the planes.  Asynchronous like pt_render_device.                                       */
int pt_render_device_packed(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params, void* d_packed);
int pt_sync(PtContext* ctx);
int pt_get_stats(PtContext* ctx, PtStats* out);
/* Debug: the 16 raw 64-bit device-side counters behind PtStats as last collected ([0] shadow rays, [1] vertices, [2] deepest
 * vertex, [3] level-0 vertices, [7] internal error flag, [8..] timing words of measurement builds, else 0).             */
int pt_debug_raw_stats(PtContext* ctx, uint64_t* out16);
/* Debug: the entries one linear scan of the uploaded scene tests: spheres, single triangles, and triangle PAIRS (two consecutive
 * triangles with the same first vertex and plane normal, e.g. the halves of a wall of World::new(), are tested together).     */
int pt_debug_scan_layout(PtContext* ctx, uint32_t* n_spheres, uint32_t* n_triangles, uint32_t* n_pairs);

/* Same render with HOST output buffers (blocking): device staging is owned by the
 * context, results are copied back over PCIe.  out_rgba8 may be NULL.          */
int pt_render_host(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params,
                   float* out_linear_rgb, uint8_t* out_rgba8);

/* Progressive preview (the reference redraws World.data every 16 ms while the rayon loop fills it,
 * src/main.rs:79-90): the same render in increments of spp_step samples.  After each increment the
 * host buffers hold the mean of the samples so far and fn is called; a non-zero return stops early.
 * The final frame is bit-identical to pt_render_host with the same parameters.                    */
typedef int (*PtProgressFn)(void* user, uint32_t spp_done, uint32_t spp_total,
                            const uint8_t* rgba8, const float* linear_rgb);
int pt_render_progressive(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params,
                          uint32_t spp_step, PtProgressFn fn, void* user,
                          float* out_linear_rgb, uint8_t* out_rgba8);

/* One-shot convenience with HOST buffers: create context on device 0 (cached),
 * upload, render, copy back.  = everything src/main.rs:43-60 does.  params->n_devices > 1
 * renders through pt_render_multi on devices 0 .. n_devices-1.  The cached contexts are freed
 * by pt_shutdown() (also registered with atexit).                              */
int pt_render(const PtCamera* cam, const PtObject* objs, uint32_t n_objs,
              const PtRenderParams* params, float* out_linear_rgb, uint8_t* out_rgba8);
void pt_shutdown(void);

/* ---- multi-GPU ------------------------------------------------
 * ONE process drives n devices: one context per device, interleaved row bands (device g renders the bands b with
 * b % n == g; pixels are independent units keyed by (x, y), src/main.rs:51, so there is no collective on the data
 * path), and ONE RCCL gather (ncclGather, rccl.h:745) of the 16 B/pixel film tiles -- written by each device's film
 * resolve straight into its send buffer -- to the first device over xGMI.  The frame is bitwise independent of n.
 * RCCL is loaded with dlopen by pt_multi_create; hosts that render on one GPU never need it.
 * pt_multi_render_device only enqueues and returns: frames posted back to back are kept apart by stream order on every
 * device, and a device's share of a frame costs the host 13-14 us, so ONE host thread (the caller's; the n gather calls inside
 * one ncclGroupStart/End) keeps eight devices fed.  pt_multi_set_threads(m, 1) gives every device its own host thread inside
 * the library instead (its render launches, its ncclGather call on its own communicator): the call then returns as soon as the
 * frame is posted, and an error of a posted frame is reported by the next pt_multi_sync / pt_multi_get_stats.  The object
 * itself is not internally synchronised: call it from one thread at a time.                                          */
typedef struct PtMulti PtMulti;
int pt_multi_create(const int* devices, uint32_t n_devices, PtMulti** out);   /* contexts + ncclCommInitAll (rccl.h:236) + one 16-byte gather that connects the ranks */
int pt_multi_destroy(PtMulti* m);
uint32_t pt_multi_device_count(const PtMulti* m);
int pt_multi_set_threads(PtMulti* m, int enabled);
/* How the tiles reach the first device.  PT_EXCHANGE_RCCL (default): one ncclGather per frame.  PT_EXCHANGE_COPY: every device
 * copies its tile into the root's receive buffer with the DMA engines (hipMemcpyPeerAsync over xGMI) and the root waits for the n
 * copies -- no kernel takes part, so the exchange never waits for wave slots beside the render (RCCL's gather kernel does, see
 * pt_multi.cpp).  Same frame bit for bit.  Exercised on one device only (a same-device copy), like everything with n > 1.   */
#define PT_EXCHANGE_RCCL 0u
#define PT_EXCHANGE_COPY 1u
int pt_multi_set_exchange(PtMulti* m, uint32_t mode);
int pt_multi_scene_upload(PtMulti* m, const PtObject* objs, uint32_t n_objs);  /* replicated on every device */
int pt_multi_set_tuning(PtMulti* m, const PtTuning* tuning);                   /* pt_context_set_tuning on every device's context */
/* Post one frame: d_linear_rgb (H*W*3 floats) / d_rgba8 (H*W*4 bytes or NULL) are buffers on the FIRST device.
 * params->band_rows = 0 picks about eight bands per device; band_index / band_count are ignored.  Asynchronous.   */
int pt_multi_render_device(PtMulti* m, const PtCamera* cam, const PtRenderParams* params,
                           float* d_linear_rgb, uint8_t* d_rgba8);
int pt_multi_sync(PtMulti* m);                       /* every posted frame is complete on every device */
int pt_multi_get_stats(PtMulti* m, PtStats* out);    /* counters of the frames since the last collection summed over the devices, times of the slowest */
int pt_multi_render_host(PtMulti* m, const PtCamera* cam, const PtRenderParams* params,
                         float* out_linear_rgb, uint8_t* out_rgba8);
/* What the object is made of -- a record of an N-device run carries these to show that N ranks took part.          */
typedef struct {
    uint32_t n_devices;
    uint32_t comm_count;      /* ncclCommCount (rccl.h) of the first device's communicator; 0: a debug object without RCCL */
    uint32_t rccl_version;    /* ncclGetVersion, e.g. 22203                                                         */
    uint32_t threaded;        /* 1: one host thread per device                                                      */
    uint64_t frames;          /* frames posted since creation                                                       */
    double   enqueue_us_sum;  /* host time spent enqueueing one frame, summed over the devices (mean per frame) ...  */
    double   enqueue_us_max;  /* ... and the slowest device's share of it: what a frame costs the host with threads  */
    uint32_t exchange;        /* PT_EXCHANGE_*                                                                      */
    uint32_t reserved_;
} PtMultiInfo;
int pt_multi_info(PtMulti* m, PtMultiInfo* out);
/* One shot with host buffers (a cached PtMulti for the device list; pt_shutdown frees it). */
int pt_render_multi(const int* devices, uint32_t n_devices, const PtCamera* cam, const PtObject* objs,
                    uint32_t n_objs, const PtRenderParams* params, float* out_linear_rgb, uint8_t* out_rgba8);

/* The two kernels pt_multi_* runs around its ncclGather, for hosts that bring their own collective (one process per GPU:
 * pathtrace_amd/dist.py over torch.distributed / RCCL, an MPI host, ...).  Both work on DEVICE memory of the calling
 * thread's current HIP device and are enqueued on hip_stream (a hipStream_t; NULL = that device's default stream).
 * The caller guarantees that hip_stream was created on that current device (hipSetDevice before the call in a
 * multi-device host): a stream of another device makes the launch fail with PT_ERR_HIP, it is not redirected.
 *   pt_film_pack:   a rank's tile (d_linear_rgb: n_pixels * 3 floats, d_rgba8: n_pixels * 4 bytes or NULL) ->
 *                   d_packed, 16 bytes per pixel (12 B linear RGB + 4 B RGBA8): both film planes in ONE gather.
 *   pt_film_unpack: the gathered tiles (rank g's tile, padded to max_rows rows, at d_gathered + g * max_rows * width *
 *                   16 bytes) -> the frame in image order (d_linear_rgb: width * height * 3 floats, d_rgba8 or NULL),
 *                   for the interleaved bands of PtRenderParams (band b belongs to rank b % n_ranks).               */
int pt_film_pack(void* hip_stream, const float* d_linear_rgb, const uint8_t* d_rgba8, uint32_t n_pixels, void* d_packed);
int pt_film_unpack(void* hip_stream, const void* d_gathered, uint32_t width, uint32_t height, uint32_t band_rows,
                   uint32_t n_ranks, uint32_t max_rows, float* d_linear_rgb, uint8_t* d_rgba8);

/* Debug / parity entry: the frame of an n_virtual-device render on ONE context (tiles rendered one after another,
 * device-to-device copies where pt_multi_* runs ncclGather): partition, packed resolve and unpack for any n on a one-GPU box. */
int pt_debug_multi_emulate(PtContext* ctx, uint32_t n_virtual, const PtCamera* cam, const PtRenderParams* params,
                           float* out_linear_rgb, uint8_t* out_rgba8);
/* Debug object: a PtMulti of n contexts that all sit on ONE device, without RCCL -- each context copies its tile into the
 * receive buffer on its own stream where the real object calls ncclGather.  Everything else is the real thing: the host
 * threads, frames posted back to back, the packed resolve, the row permutation, the statistics.  Rehearses (and times:
 * pt_multi_info) an n-device frame on a one-GPU box.                                                                */
int pt_debug_multi_create_shared(int device, uint32_t n, PtMulti** out);
/* Host-only self test of the per-device feeder threads (no GPU needed): posts n_frames jobs to each of n_workers threads
 * the way pt_multi_render_device posts frames and returns the start / end log of the jobs in order_out (2 * n_workers *
 * n_frames entries: worker << 32 | frame, bit 63 set on the end record); fail_at >= 0 makes job worker * n_frames + frame
 * fail, and the call then returns the status the drain reported.                                                     */
int pt_debug_feeder_selftest(uint32_t n_workers, uint32_t n_frames, uint32_t spin, int32_t fail_at,
                             uint64_t* order_out, uint32_t* n_out);

/* The launch scheduler of a context, host only (no GPU needed).  A render is planned by a PURE function (csrc/pt_sched.h):
 * (scheduling state of the context, job) -> the list of stream operations the render enqueues -- path-kernel launches and film
 * resolves with their stream, buffer set, lane, exchange region and counters, the event records / waits that order them, the
 * fills of counters and statistics.  pt_render_device* executes exactly such a list.  These entries run the same function on a
 * scheduling state of their own, so that a test can drive random sequences of jobs through it and check the invariants of
 * DESIGN.md 3 with a happens-before simulator (tests/test_sched_cpu.py).  Field meanings: csrc/pt_sched.h (Job, Op).         */
typedef struct {
    uint32_t n_batches, regen, split, hand_off, regen_export, profile, in_order, capturing;
    uint32_t grid, regen_grid, cont_grid, regen_capacity, fixed_grid, counter_words;
    uint64_t xchg_need;
} PtSchedJob;
typedef struct {
    uint32_t kind, stream, event, pool, set, lane, level, own_queue, ovf_par, batch, grid, seq, core, flags, zero_words, reserved;
    uint64_t xchg_off, xchg_len;
} PtSchedOp;
typedef struct PtSched PtSched;
int pt_debug_sched_create(PtSched** out);
void pt_debug_sched_destroy(PtSched* s);
/* Plans one render and advances the state.  faults: bit 0 / bit 1 switch round 4's two scheduling bugs back on (pt_sched.h:
 * Faults; for the test that shows they are caught).  fail_after < the plan's length: the operation of that index fails as a
 * HIP call would -- only the operations before it are returned, followed by the host synchronisation of the recovery, and the
 * state is what render_impl's recovery leaves.  *lanes = 1 if the render took the lanes.                                     */
int pt_debug_sched_render(PtSched* s, const PtSchedJob* job, uint32_t faults, uint32_t fail_after, PtSchedOp* ops, uint32_t cap,
                          uint32_t* n_ops, uint32_t* lanes);
int pt_debug_sched_sync(PtSched* s, uint32_t collect);        /* pt_sync: everything enqueued is complete (collect: statistics read and cleared) */
/* Test hook: the n-th stream operation (0-based) of the NEXT render on this context fails as if its HIP call had (n < 0: none). */
int pt_debug_fail_after(PtContext* ctx, int64_t n);

/* World::render_pixel (src/world.rs:293-333) -- the seam the reference's rayon loop calls at
 * src/main.rs:55 -- for an arbitrary list of n pixels: xy = n * (x, y), y = film row (top-down, the y
 * of the seed (y<<32)|x, main.rs:51).  Every listed pixel gets exactly the samples a full render gives
 * it (same key, same sample indices spp_offset .. spp_offset+spp-1), so out_linear_rgb[3i..] /
 * out_rgba8[4i..] are bit-identical to that pixel of the full film.  out_samples (optional,
 * n * spp * 3 floats, [pixel][sample][rgb]) receives the radiance of every camera sample =
 * RenderingStrategy::ray_color's return value (rendering.rs:34,214; what the reference's pixel
 * diagnostics print, world.rs:378-417); it needs the whole list in one sample batch
 * (n * spp <= max_paths_in_flight).  Host buffers, blocking.  band_* of params must select the
 * whole image.                                                                                */
int pt_render_pixels(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params,
                     const uint32_t* xy, uint32_t n, float* out_linear_rgb, uint8_t* out_rgba8,
                     float* out_samples);

/* RenderingStrategy::ray_color(world, ray, depth = 0, rng, throughput = 1) (src/rendering.rs:34-142,
 * 214-265) for n arbitrary rays: rays = n * (origin3, direction3), the direction is normalised on
 * entry like Ray::new (camera.rs:10-16); xy = n * (x, y) = the RNG key of each ray's stream, and the
 * sample index of every stream is params->spp_offset.  out_rgb = n * 3 floats.  Host buffers,
 * blocking; the paths run through the same kernels as a render.                               */
int pt_ray_color(PtContext* ctx, const PtRenderParams* params, const double* rays,
                 const uint32_t* xy, uint32_t n, float* out_rgb);

/* Debug/parity entry: closest-hit scan of World::hit_scene (src/world.rs:270-290)
 * on the device for n arbitrary rays (host arrays; rays = n*6 doubles o,d; the
 * direction is normalised on entry like Ray::new, src/camera.rs:10-16).
 * out_id[i] = object index or -1, out_t[i] = hit distance.                   */
int pt_debug_hit_scene(PtContext* ctx, const double* rays, uint32_t n,
                       double t_min, double t_max, uint32_t exact_math, uint32_t accel,
                       int32_t* out_id, float* out_t);

/* The same with the HitRecord of every hit (src/objects/base.rs:6-33): out_rec = n * 8 floats
 * (t, point3, face-forwarded normal3, front_face as 0/1); zeros for a miss.   */
int pt_debug_hit_records(PtContext* ctx, const double* rays, uint32_t n,
                         double t_min, double t_max, uint32_t exact_math, uint32_t accel,
                         int32_t* out_id, float* out_rec);

/* Function-level parity entries: the per-vertex device functions of the path kernels on
 * arbitrary inputs, one item per thread.  `obj` = index of an object of the uploaded scene.  Host arrays.
 *   pt_debug_bsdf_eval     Material::bsdf_pdf (material.rs:86-91, 139-148, 221-265; mirror.rs:179-198)
 *                          in = n * (ray dir3, wo3, normal3, ray.eta_ratio) -> out = n * (f3, pdf)
 *   pt_debug_bsdf_sample   Material::bsdf_pdf_sample (material.rs:29-40; mirror.rs:200-305)
 *                          in = n * (ray dir3, normal3, eta_ratio), words = n * 4 raw RNG words (r1, r2, lobe u, -)
 *                          -> out = n * (wo3, f3, pdf, cos)
 *   pt_debug_shape_sample  Shape::sample_surface_from_point (shape.rs:91-145, 200-242) of object obj from
 *                          from3[i]; target3 != NULL: the look-ahead form (point given, no draws), else r12 =
 *                          n * (r1, r2) uniforms -> out = n * (point3, pdf_omega, light_dir3, distance)
 *   pt_debug_light_point   World::sample_light_point (world.rs:251-267) from from3[i]; words = n * 4
 *                          (light-index word, r1 word, r2 word, -) -> out = n * (point3, emission3, pdf, light object)
 *   pt_debug_camera_rays   Camera::get_ray_with_offset with the sample's own jitter draws (camera.rs:139-147,
 *                          world.rs:299); xys = n * (x, y film row, sample) -> out = n * (origin3, direction3, ox, oy) */
int pt_debug_bsdf_eval(PtContext* ctx, uint32_t obj, const double* in10, uint32_t n, uint32_t exact_math, float* out4);
int pt_debug_bsdf_sample(PtContext* ctx, uint32_t obj, const double* in7, const uint32_t* words4, uint32_t n,
                         uint32_t exact_math, float* out8);
int pt_debug_shape_sample(PtContext* ctx, uint32_t obj, const double* from3, const double* target3, const double* r12,
                          uint32_t n, uint32_t exact_math, float* out8);
int pt_debug_light_point(PtContext* ctx, const double* from3, const uint32_t* words4, uint32_t n, uint32_t exact_math,
                         float* out8);
int pt_debug_camera_rays(PtContext* ctx, const PtCamera* cam, const uint32_t* xys, uint32_t n, uint32_t exact_math,
                         float* out8);

/* Debug entry, host only (no GPU needed): build the accel = 1 BVH of a scene and verify it -- every object in
 * exactly one leaf slot with its scan record, every child box encloses the boxes beneath it, depth within the
 * traversal stack.  Returns PT_OK and the tree's size, or PT_ERR_UNSUPPORTED with the violated invariant in
 * pt_last_error().  Any of the three outputs may be NULL.                                                    */
int pt_debug_bvh_check(const PtObject* objs, uint32_t n_objs, uint32_t* depth, uint32_t* n_nodes,
                       uint32_t* n_leaf_slots);

const char* pt_last_error(void);
uint32_t pt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACE_AMD_H */
