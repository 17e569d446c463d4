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

#define PT_ABI_VERSION 1

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
    /* Wavefront sizing: upper bound on paths resident in HBM at once
     * (0 = library default).                                                  */
    uint64_t max_paths_in_flight;
    uint32_t profile;        /* 1: time every path-kernel launch with HIP events */
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
} PtRenderParams;

/* Counters of the last render on a context. */
typedef struct {
    uint64_t samples;          /* camera samples traced = tile pixels * spp    */
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
 * stream) instead of the context's own stream.  NULL restores the default.    */
int pt_context_set_stream(PtContext* ctx, void* hip_stream);

/* Copy the scene to the device (the reference's World is immutable while
 * rendering: render_pixel(&self), src/world.rs:293).  Lights are detected as
 * in src/world.rs:214-225: objects whose emit() has non-zero length.          */
int pt_scene_upload(PtContext* ctx, const PtObject* objs, uint32_t n_objs);

/* Render the tile into DEVICE buffers (no host transfer inside):
 *   d_linear_rgb: float[tile_rows*W*3], mean linear radiance  (= luminance_data,
 *                 src/world.rs:318-319)
 *   d_rgba8:      uint8[tile_rows*W*4], sqrt-gamma + truncation (= World.data /
 *                 draw(), src/world.rs:322-341); may be NULL.
 * Work is enqueued on the context's stream; the call returns once the last launch is
 * enqueued (it synchronises internally once per sample batch, where the tail of the
 * level-0 launch is handed to a continuation launch).  Renders of several sample batches
 * also use a second, context-owned stream for those tails; the context's stream waits
 * for it at the end, so everything is complete when that stream is.  pt_sync() waits.   */
int pt_render_device(PtContext* ctx, const PtCamera* cam, const PtRenderParams* params,
                     float* d_linear_rgb, uint8_t* d_rgba8);
int pt_sync(PtContext* ctx);
int pt_get_stats(PtContext* ctx, PtStats* out);

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
 * upload, render, copy back.  = everything src/main.rs:43-60 does.            */
int pt_render(const PtCamera* cam, const PtObject* objs, uint32_t n_objs,
              const PtRenderParams* params, float* out_linear_rgb, uint8_t* out_rgba8);

/* Debug/parity entry: closest-hit scan of World::hit_scene (src/world.rs:270-290)
 * on the device for n arbitrary rays (host arrays; rays = n*6 doubles o,d; the
 * direction is normalised on entry like Ray::new, src/camera.rs:10-16).
 * out_id[i] = object index or -1, out_t[i] = hit distance.                   */
int pt_debug_hit_scene(PtContext* ctx, const double* rays, uint32_t n,
                       double t_min, double t_max, uint32_t exact_math, uint32_t accel,
                       int32_t* out_id, float* out_t);

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
