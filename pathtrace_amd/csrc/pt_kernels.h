// pt_kernels.h -- launch interface between the host driver (pt_api.cpp) and the
// HIP kernels (pt_kernels.hip).  Plain structs, no HIP types besides float4 and
// hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptk {

// One contiguous run of same-shape objects of World.objects, in object order
// (the order decides closest-hit ties, src/world.rs:281-287).
// kRunTrianglePair: entries of two consecutive triangles that share v0 and the plane normal bit for bit (the halves of a
// parallelogram fanned from one corner): 5 float4 = the first triangle's record + (N1.xyz, N2.x), (N2.yz, -, -) of the
// second; objects first_obj + 2k and first_obj + 2k + 1.
enum { kRunSphere = 0, kRunTriangle = 1, kRunTrianglePair = 2 };
struct Run {
    uint32_t tag;        // kRunSphere / kRunTriangle (= the shape tags) / kRunTrianglePair
    uint32_t first_obj;  // object index of the run's first primitive
    uint32_t count;      // entries in the run (primitives; pairs)
    uint32_t off4;       // offset of the run in the scan array, in float4 units
};
constexpr uint32_t run_entry_f4(uint32_t tag) { return tag == kRunSphere ? 1u : tag == kRunTriangle ? 3u : 5u; }

// Optional BVH over the objects (pt_bvh.h); built on the host the first time a render asks for it.
struct BvhView {
    const uint4* nodes;      // 4 uint4 (64 bytes) per internal node: up to four child boxes on the 16-bit grid + their child codes (pt_bvh.h)
    float grid_min[3], grid_cell[3];   // box coordinate = grid_min + q * grid_cell
    const float4* rec;       // 3 float4 per leaf slot (scan record of the primitive); the traversal reads e1, e2 of triangles here
    const float4* lead;      // 1 float4 per leaf slot = rec[3 * slot]: sphere (c, r^2) / triangle v0 -- a leaf's <= 4 are one 64-byte line
    const uint32_t* ids;     // object index per leaf slot (bit 31: triangle); a leaf starts at a multiple of 4: one 16-byte load
    uint32_t root;           // child code of the root
    float scene_abs;         // scale of the padding the slab test applies (see bvh_scan)
};

struct SceneView {
    const float4* scan;      // scan records, run-packed: sphere = 1 float4 (c, r^2); triangle = 3 float4 (v0, n, N1, N2: pt_bvh.h triangle_scan_record)
    const float4* shape;     // 3 float4 per object (gather form), see pt_device.h
    const float4* mat;       // 2 float4 per object
    const Run* runs;
    const uint32_t* lights;  // object indices of the emitters (world.rs:214-225)
    // scenes of <= kSmallObjs objects: the five arrays above packed back to back
    // [scan | shape | mat | runs | lights], copied whole into LDS by every workgroup
    const float4* blob;
    uint32_t blob_f4;        // float4 count of `blob` (0 for larger scenes)
    uint32_t scan_f4;        // float4 count of `scan`
    uint32_t n_runs, n_objs, n_lights;
    uint32_t diffuse_only;   // every material is Lambertian or emissive: kernels without the GGX / OrenNayar code
    uint32_t no_mirror;      // no Mirror surface (k_paths_regen without the GGX code also for scenes with OrenNayar surfaces)
    uint32_t no_oren_nayar;  // no OrenNayar surface (k_paths_regen_split: its plain iterations are then the diffuse-only code)
    BvhView bvh;             // valid only for launches with accel != 0
};

struct CameraF {             // Camera's cached fields in f32 (camera.rs:36-38)
    float origin[3], lower_left[3], horizontal[3], vertical[3];
    uint32_t width, height;
};

// Path-state queue: 4 float4 planes, index = queue slot (coalesced 16 B/lane).
//   ray part    q0 = (o.x, o.y, o.z, d.x)   q1 = (d.y, d.z, bits(tile_row<<16 | x), bits(s_local<<16 | depth))
//   carry part  q2 = (beta.x, beta.y, beta.z, pdf_prev)   q3 = (L.x, L.y, L.z, eta_in)
// so width, tile rows and samples per batch are each < 65536.  The kernels read the carry part of a path only after
// its vertex's scans (it is not needed before), which keeps eight registers free during them.
struct Queue {
    float4* q[4];
};

// Row-band tile (include/pathtrace_amd.h): tile row yl -> image row
//   y = (yl / band_rows) * band_stride + band_first + yl % band_rows
// with yl / band_rows = umulhi(yl, band_magic) (exact for yl, band_rows < 65536).
struct TileMap {
    uint32_t band_rows, band_magic, band_stride, band_first;
};

// Final radiance of one path (per-sample buffer `lsamp`): 12 bytes, so that the buffer and the resolve's reads are a
// quarter smaller than with a float4.
struct Rgb {
    float r, g, b;
};

struct BounceArgs {
    Queue q;                  // compacted in place, one private segment per wave
    // tail hand-off between launches (small scenes): a wave whose segment falls below export_below paths
    // appends them to ovf_out (slot = atomicAdd(ovf_out_count, n)) and retires; a continuation launch
    // (src_mode = 1) takes its n_first paths from ovf_in instead of generating camera rays
    Queue ovf_in, ovf_out;
    uint32_t* ovf_out_count;
    // non-null: level-0 launch in the regenerating form (k_paths_regen; scene in LDS, src_mode 0, no pixel list): the
    // waves take the batch's 64-path chunks from this counter (zeroed before the launch) and export what is alive
    // when it runs out
    uint32_t* chunk_counter;
    // non-null (with chunk_counter): the regenerating form that batches Mirror vertices (k_paths_regen_split); per wave of the
    // launch kXqF4PerWave float4 of exchange stacks + parking area
    float4* xchg;
    uint32_t regen_static;    // chunks dealt round-robin to the waves (a multiple of the launch's wave count); the rest by the counters
    // non-null (regenerating launches that overlap their neighbours: pt_api.cpp, lanes): the workgroups from core_blocks on are
    // SPARE -- each reads *posted (host memory: launches the host has enqueued so far) when it starts and ends at once if two
    // or more launches follow this one (seq = this launch's number); otherwise it works like any other
    const uint32_t* posted;
    uint32_t seq, core_blocks;
    uint32_t src_mode;        // 0: pass 0 generates camera rays, 1: pass 0 reads ovf_in
    uint32_t export_below;    // >= 1; 1 = never export (a wave runs until its segment is empty)
    uint32_t seg_cap;         // slots per segment (multiple of 64)
    // accel = 1 only (k_paths_bvh): per-slot scratch of the staged passes, same indexing as q
    float4* aux;              // (closest-hit id, t, occluded, -)
    float4* sray0;            // shadow ray (o, d.x)
    float4* sray1;            // (d.y, d.z, t_max, 1 = the slot has a shadow ray)
    Rgb* lsamp;               // per-path final radiance, index = s_local*np + tile_row*width + x
    unsigned long long* stats;  // [0] shadow rays  [1] path vertices  [2] deepest vertex (max)  [3] vertices of level-0 launches  [4] finished samples (lsamp writes)  [7] internal error flag (k_paths_regen_split: exchange stacks met)
    TileMap tile;
    SceneView sc;
    CameraF cam;
    uint32_t n_first;         // paths of the batch
    // continuation launches may take their path count from device memory instead (the counter the previous launch's
    // waves added their leftovers to): the host then never waits for it.  seg_cap is derived on the device in that case.
    const uint32_t* n_first_dev;
    // pixel-list renders (pt_render_pixels / pt_ray_color): film slot i = (tile_row << 16) | x of the path state,
    // RNG key and camera pixel = pixels[i] = (x, y) of the image.  film_w = row pitch of the film-slot arithmetic
    // (camera width, or 65536 for a list).
    const uint2* pixels;
    uint32_t film_w;
    uint32_t film_w_magic;    // floor(2^32 / film_w), 0xFFFFFFFF for 1 (divmod_magic)
    uint32_t np;              // pixels of the tile
    uint32_t np_magic;        // floor(2^32 / np), 0xFFFFFFFF for 1
    uint32_t s_base;          // sample index of s_local = 0 (spp_offset + batch start)
    uint32_t min_depth, max_depth;
    float t_min;
    uint32_t integrator;
    uint32_t bvh_refill, bvh_leaf;   // traverse_segment thresholds (lanes)
    uint32_t accel;           // 0: linear scan (the reference's hit_scene), 1: BVH traversal, same answers
    uint32_t debug_tag;       // measurement builds (PT_DRAIN_TIMING): plane of ovf_out that receives the per-wave stamps
};

constexpr uint32_t kBlock = 256;
// Scenes of at most kSmallObjs objects stay whole in LDS (scan + shape + material + run
// records: at most 9 float4 per object = 18 KiB); larger scenes stream their scan array
// through one LDS tile and gather shape/material records from global memory.
constexpr uint32_t kSmallObjs = 128;
constexpr uint32_t kTileF4 = 1920;         // 30 KiB LDS tile (divisible by 3: whole triangles); 5 workgroups per CU (measured: 2550 / 4 -> 1214 ms, 1920 / 5 -> 1106 ms on C4)
constexpr uint32_t kRefillBelow = 44;      // BVH traversal: hand out new rays when fewer lanes than this are tracing
constexpr uint32_t kLeafBatch = 20;        // BVH traversal: test leaf primitives when at least this many lanes wait at a leaf
constexpr uint32_t kBvhMaxLeaf = 4;        // primitives per BVH leaf the traversal unrolls for (= ptbvh::kMaxLeaf)
#ifndef PT_BVH_STACK
#define PT_BVH_STACK 24
#endif
constexpr uint32_t kBvhStack = PT_BVH_STACK;         // traversal stack entries per lane, in LDS (= ptbvh::kStackDepth): 24 KiB per workgroup, 5 workgroups per CU

// One launch traces every path of a batch to its end.  grid = number of 256-thread workgroups
// (4 queue segments each).  _exact / _fast: the two arithmetic modes of pt_device.h
// (PtRenderParams.exact_math).
// occupancy k_paths_regen is compiled for (waves per SIMD = workgroups per CU): the host launches exactly that many
#ifndef PT_REGEN_WAVES_DIFFUSE
#define PT_REGEN_WAVES_DIFFUSE 6
#endif
#ifndef PT_REGEN_WAVES_SPLIT
#define PT_REGEN_WAVES_SPLIT 5
#endif
constexpr uint32_t kRegenWavesDiffuse = PT_REGEN_WAVES_DIFFUSE, kRegenWavesGeneric = 5, kRegenWavesSplit = PT_REGEN_WAVES_SPLIT;
// k_paths_regen_split: float4 of exchange memory per wave of the launch: 128 stack entries of 5 float4
constexpr uint32_t kRegenSplitF4PerWave = 128u * 5u;
// chunk counters of k_paths_regen: chunk_counter[c * kRegenCounterStride], c < kRegenCounters (256 bytes apart)
constexpr uint32_t kRegenCounters = 8, kRegenCounterStride = 64;
// pt_scene_upload: per-object constants written into the shape / material records (k_scene_setup), one call per arithmetic
// mode on that mode's copy of the records
void launch_scene_setup_exact(float4* shape, float4* mat, uint32_t n_objs, hipStream_t st);
void launch_scene_setup_fast(float4* shape, float4* mat, uint32_t n_objs, hipStream_t st);
// workgroups per CU of the regenerating level-0 kernel `a` selects (sc, integrator, xchg), with the scene's LDS blob; 0 = unknown
uint32_t regen_blocks_per_cu_exact(const BounceArgs& a);
uint32_t regen_blocks_per_cu_fast(const BounceArgs& a);
void launch_paths_exact(const BounceArgs& a, uint32_t grid, hipStream_t st);
void launch_paths_fast(const BounceArgs& a, uint32_t grid, hipStream_t st);
// between the translation units pt_kernels.hip is built as (PT_TU there; not called by the host code): k_paths_regen_split's and the
// BVH form's launchers live with their kernels
int regen_split_blocks_per_cu_exact(const BounceArgs& a, size_t lds);
int regen_split_blocks_per_cu_fast(const BounceArgs& a, size_t lds);
void launch_regen_split_exact(const BounceArgs& b, uint32_t blocks, size_t lds, hipStream_t st);
void launch_regen_split_fast(const BounceArgs& b, uint32_t blocks, size_t lds, hipStream_t st);
void launch_paths_bvh_exact(const BounceArgs& a, uint32_t grid, size_t lds, hipStream_t st, bool diffuse, bool list);
void launch_paths_bvh_fast(const BounceArgs& a, uint32_t grid, size_t lds, hipStream_t st, bool diffuse, bool list);
void launch_debug_hit_bvh_exact(const SceneView& sc, uint32_t grid, size_t lds, const float* rays6, uint32_t n, float t_min, float t_max,
                                float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st);
void launch_debug_hit_bvh_fast(const SceneView& sc, uint32_t grid, size_t lds, const float* rays6, uint32_t n, float t_min, float t_max,
                               float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st);

// Film: sum the nb samples of every tile pixel in sample order into the f64
// accumulator (world.rs:311), and when finalising write mean, sqrt-gamma and
// truncated RGBA8 (world.rs:315-332).
struct ResolveArgs {
    const Rgb* lsamp;
    double* film;             // np*3 doubles (may be null when the render is a single batch)
    float* out_linear;        // np*3
    uint8_t* out_rgba;        // np*4 or null
    uint32_t np, nb;
    uint32_t load_film;       // start from the f64 sums in `film` (not the first samples of the pixel)
    uint32_t store_film;      // keep the sums in `film` (more samples follow)
    uint32_t finalize;        // write the outputs: mean over spp_div samples, gamma, quantisation
    uint32_t spp_div;
    // non-null: the finalising pass writes ONE 16-byte record per pixel here (12 B linear RGB + 4 B RGBA8: the send
    // buffer of the multi-GPU film gather) INSTEAD of the two planes -- what k_film_pack made of them in a second launch
    void* out_packed;
    // non-null: workgroup 0 clears these words (the chunk / hand-over counters of the batch just resolved, so that the
    // next launch that uses them finds them zero without a memset of its own in the stream)
    uint32_t* zero_words;
    uint32_t n_zero;
};
void launch_resolve(const ResolveArgs& a, hipStream_t st);

// Multi-GPU film exchange (pt_multi.cpp): tile -> 16 B per pixel (linear RGB + RGBA8) before the gather, gathered
// padded tiles -> frame in image order after it.
void launch_film_pack(const float* lin, const uint8_t* rgba, uint32_t np, void* packed, hipStream_t st);
void launch_film_unpack(const void* recv, uint32_t W, uint32_t H, uint32_t band_rows, uint32_t n_dev, uint32_t max_rows, float* lin,
                        uint8_t* rgba, hipStream_t st);

// World::hit_scene on arbitrary rays (debug/parity entry).
// scratch: accel = 1 only, 3*n float4 of device memory.  out_rec: optional, 8 floats per ray (t, point3, normal3, front_face).
void launch_debug_hit_exact(const SceneView& sc, uint32_t accel, const float* rays6, uint32_t n, float t_min, float t_max,
                            float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st);
void launch_debug_hit_fast(const SceneView& sc, uint32_t accel, const float* rays6, uint32_t n, float t_min, float t_max,
                           float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st);

// The per-vertex functions on arbitrary inputs (debug/parity entries; layouts at k_debug_fn in pt_kernels.hip).
enum { kFnBsdfEval = 0, kFnBsdfSample = 1, kFnShapeSample = 2, kFnLightPoint = 3, kFnCameraRay = 4 };
struct DebugFnArgs {
    SceneView sc;
    CameraF cam;              // kFnCameraRay only
    uint32_t op, obj, n;
    uint32_t in_stride, out_stride;   // floats per item
    const float* in;
    const uint32_t* words;    // 4 raw words per item, or null
    float* out;
};
void launch_debug_fn_exact(const DebugFnArgs& a, hipStream_t st);
void launch_debug_fn_fast(const DebugFnArgs& a, hipStream_t st);

}  // namespace ptk
