// pt_kernels.hip -- HIP kernels of the wavefront path tracer, gfx950 (wave64).
//
// Every path vertex is one level of MisStrategy::ray_color / BrdfOnlyStrategy::ray_color
// (src/rendering.rs:34-142, 214-265) in the iterative order of SURVEY 3.5:
//     closest hit of the path ray            World::hit_scene      world.rs:270-290
//     miss -> retire; emitter -> credit (MIS look-ahead weight), retire
//     NEE: pick light, sample its surface    World::sample_light_point  world.rs:251-267
//          shadow scan, BSDF eval, MIS weight                      rendering.rs:55-81
//     BSDF sample, throughput, Russian roulette                    rendering.rs:83-102
//     survivors are compacted in place into the wave's own queue segment (wave64 ballot
//     + prefix popcount, no atomics); retired paths store their radiance to lsamp[pid].
// (vertex_begin / vertex_end below; shared by both path kernels.)
//
// Kernels:
//   k_paths<MODE, MIS, OVF>   one launch traces a whole sample batch, every bounce; hit_scene is the reference's
//                             linear scan, out of LDS (MODE = kModeLds, scenes <= 128 objects) or streamed through
//                             an LDS tile (kModeTiled).  Pass 0 generates the camera rays (camera.rs:139-147,
//                             world.rs:299) or, in a continuation launch (OVF), takes over the overflow queue.
//   k_paths_regen<MIS, MATS>  level-0 launch of a batch of > 2^17 paths over a scene in LDS: a path stays in its lane's registers,
//                             a lane whose path ends takes the batch's next one (chunk counters); compiled per material set.
//   k_paths_regen_split<..>   the same for scenes with a few Mirror objects (the reference's own): a wave's Mirror vertices are
//                             set aside on a per-wave stack and shaded 64 at a time.
//   k_paths_bvh<MIS, OVF>     the queue form for PtRenderParams.accel = 1: hit_scene by traversal of a 4-wide BVH, each pass cut
//                             into extend / connect / occlude / shade stages with per-lane ray refill.
//   k_scene_setup             per-object constants (a triangle's unit normal and 1 / area) at pt_scene_upload.
//   k_resolve                 film: per-pixel f64 sum in sample order, mean, gamma, RGBA8 (world.rs:311-332).
//   k_debug_hit[_bvh]         hit_scene on arbitrary rays (parity tests).
//
// Data layout: path state = 4 float4 planes (SoA of float4 -> every lane moves 16 B per instruction, 1 KiB per
// wave-instruction); small scenes live in LDS and are read by all 64 lanes at the same address (broadcast,
// conflict-free).  Grids are persistent (one queue segment per wave), so no host round trip sits between bounces.
#include "pt_device.h"
#include "pt_kernels.h"

using namespace PTD_NS;
using namespace ptk;

#if PT_MATH_EXACT
#define PTK_IMPL ptk_exact_impl
#define PT_LAUNCH(name) name##_exact
#else
#define PTK_IMPL ptk_fast_impl
#define PT_LAUNCH(name) name##_fast
#endif

// Translation units (round 5).  The library builds this file three times per arithmetic mode, each with the compiler options its kernels
// measured best with (Makefile; profiles/r05/ab_noslp.txt, ab_bvh_slp.txt, ab_compiler_flags2.txt):
//   PT_TU = 1  everything but the two below                       -fno-slp-vectorize (C2 launch -2.6 %)
//   PT_TU = 2  k_paths_regen_split and its launcher               -fno-slp-vectorize, scheduling strategy max-ilp (C1 -1.2 %; C2 would pay 1.6 %)
//   PT_TU = 3  the BVH form (k_paths_bvh, k_debug_hit_bvh)        with the SLP vectoriser (its 4-wide box tests pack well: +2 % without)
//   PT_TU = 0  all of it in one unit (tools/resources.py, tools/build_variant.sh)
// Kernel templates are instantiated where their launcher is; the launchers that cross units are declared in pt_kernels.h.
#ifndef PT_TU
#define PT_TU 0
#endif
#define PT_TU_MAIN (PT_TU == 0 || PT_TU == 1)
#define PT_TU_SPLIT (PT_TU == 0 || PT_TU == 2)
#define PT_TU_BVH (PT_TU == 0 || PT_TU == 3)
#ifndef PT_PAIR_PREFETCH
#define PT_PAIR_PREFETCH 1      // 1: the next pair's normal one pair ahead (C1 launch 7.16 -> 7.07 ms); 2: its v0 too (2 spilled registers, 7.17) -- profiles/r05/ab_lds_latency.txt
#endif
#ifndef PT_PAIR_PREFETCH_GENERIC
#define PT_PAIR_PREFETCH_GENERIC 0      // the same in the generic-material kernels (5 waves per SIMD: registers to spare)
#endif
#ifndef PT_SPHERE_REM2
#define PT_SPHERE_REM2 1
#endif
#ifndef PT_RUN0_SGPR
#define PT_RUN0_SGPR 0          // measured (round 5): C1 launch +1.7 %, C2 +0.3 % -- profiles/r05/ab_lds_latency.txt
#endif
#ifndef PT_PAIR_S_EARLY
#define PT_PAIR_S_EARLY 0       // measured (round 5): C1 launch 7.09 -> 7.05 / 7.17 ms without / with the prefetch -- profiles/r05/ab_pair_s_early.txt
#endif
#ifndef PT_U_MED3
#define PT_U_MED3 1             // measured (round 5): C1 launch 7.04-7.07 -> 7.00-7.04 ms (two scalar instructions less per pair test) -- profiles/r05/ab_u_med3.txt
#endif
#ifndef PT_PAIR_FLAT
#define PT_PAIR_FLAT 0          // measured (round 5): C1 7.46 -> 7.92 ms per launch -- profiles/r05/ab_c1_replace_flat.txt
#endif
#ifndef PT_SPHERE_FLAT
#define PT_SPHERE_FLAT 0        // measured (round 5): C2 5.54 -> 5.55 ms per launch, nothing -- profiles/r05/ab_noslp.txt
#endif
namespace PTK_IMPL {

// ------------------------------------------------------------------ primitive tests
// SphereShape::hit (shape.rs:53-82) against the running closest t.  s = (center, r^2).
// Every ray the scans see is a unit vector (Ray::new normalises, camera.rs:10-16; so do the entries that take rays
// from outside), so the reference's a = d.d is 1 up to rounding and the f32 arithmetic specification takes a = 1: no
// multiplication by a or 1/a (SURVEY 8a row a5 prices the test that way: "16 if a = 1 and r^2 cached").
// ORDERED (BVH traversal, which meets the primitives in tree order): among equal t the highest object index
// wins -- what the scan's "accept t <= closest" gives when it walks the objects in index order.
// ANY (visibility scans): only "is anything accepted" is asked (rendering.rs:62-65 tests is_none()), and the first
// accepted object of the shrinking scan is tested against the initial t_max, so every test runs against that fixed
// bound and nothing is tracked but a flag: id >= 0.
template <bool ORDERED = false, bool ANY = false>
PT_DEV void sphere_test(float4 s, f3 o, f3 d, float t_min, float& closest, int& id, int obj) {
    f3 oc = o - mk(s.x, s.y, s.z);
    float half_b = dot(oc, d);
    // half_b^2 - c cancels catastrophically in f32 for a small sphere far from the origin;
    // same quantity, robust form: r^2 - |oc - half_b d|^2  (Ray Tracing Gems ch. 7)
    f3 l = madd(d, -half_b, oc);
    float disc = s.w - dot(l, l);
    if (disc < 0.0f) return;                       // NaN falls through, as in the reference (Q10)
    float sqrtd = pt_sqrt(disc);
    float root1 = -half_b - sqrtd;
    float root2 = -half_b + sqrtd;
    // shape.rs:76-82: take the near root unless it is out of range, then the far one.  root2 >= root1,
    // so "closest < root1" already rejects both; hence the candidate is root2 only when root1 < t_min.
    float c = root1 < t_min ? root2 : root1;
    if (c < t_min || closest < c) return;          // NaN is accepted, as in the reference
    if (ANY) { id = 0; return; }
    if (ORDERED && c == closest && obj < id) return;
    closest = c;
    id = obj;
}
// RangeInclusive(0.0..=1.0).contains(u) (shape.rs:176): true for -0.0, false for NaN.  Evaluated (round 5, PT_U_MED3) as
// "the median of (u, 0, 1) is u" -- one v_med3 + one compare instead of two compares and a scalar AND of their masks (the scalar ALU is
// one per CU).  Equivalent for every input: a NaN is not equal to itself, and med3(-0, 0, 1) compares equal to -0 whichever zero it returns.
PT_DEV bool in_unit_range(float u) {
#if PT_U_MED3
    return __builtin_amdgcn_fmed3f(u, 0.0f, 1.0f) == u;
#else
    return u >= 0.0f && u <= 1.0f;
#endif
}
// TriangleShape::hit (shape.rs:161-192).  The reference runs Moeller-Trumbore per ray (two cross products, three dot
// products with the edges); the f32 specification evaluates the same u, v, t from per-triangle constants built once at
// upload (ptbvh::triangle_scan_record: plane normal n = e1 x e2 and the barycentric gradients N1, N2):
//     a = e1.(d x e2) = -(d.n)        t = f e2.(s x e1) = -(s.n)/(d.n)        u = (s + t d).N1        v = (s + t d).N2
// -- 19 instead of 30 arithmetic instructions and no cross product.  The accept rules are the reference's, predicate
// for predicate: |a| < 1e-8 rejects (:169), u outside [0, 1] rejects, NaN included (RangeInclusive::contains, :176),
// v < 0 or u + v > 1 rejects (:183), t outside [t_min, closest] rejects (:190); t == closest is accepted (last wins).
// They form one conjunction, so testing the t range first (it is known first here) changes nothing.
// Record: r0 = (n, N1.x), r1 = (v0, N1.y), r2 = (N1.z, N2.xyz) -- one 16-byte read per stage of the test.
template <bool ORDERED = false, bool ANY = false>
PT_DEV void triangle_test(float4 r0, float4 r1, float4 r2, f3 o, f3 d, float t_min, float& closest, int& id, int obj) {
    const f3 n = mk(r0.x, r0.y, r0.z);
    const float det = dot(d, n);
    if (__builtin_fabsf(det) < 1e-8f) return;
    const f3 s = o - mk(r1.x, r1.y, r1.z);
    const float t = pt_div(-dot(s, n), det);
#ifdef PT_TRI_BRANCHLESS      // measurement variant: one reject at the end instead of three early-outs
    const f3 p = madd(d, t, s);
    const float u = dot(p, mk(r0.w, r1.w, r2.x)), v = dot(p, mk(r2.y, r2.z, r2.w));
    if (t < t_min || t > closest || !(u >= 0.0f && u <= 1.0f) || v < 0.0f || u + v > 1.0f) return;
#else
    if (t < t_min || t > closest) return;
    const f3 p = madd(d, t, s);                      // hit point relative to v0
    const float u = dot(p, mk(r0.w, r1.w, r2.x));
    if (!in_unit_range(u)) return;                 // RangeInclusive::contains: NaN rejected
    const float v = dot(p, mk(r2.y, r2.z, r2.w));
    if (v < 0.0f || u + v > 1.0f) return;
#endif
    if (ANY) { id = 0; return; }
    if (ORDERED && t == closest && obj < id) return;
    closest = t;
    id = obj;
}

// Two consecutive triangles with the same v0 and the same plane normal bit for bit (kRunTrianglePair, pt_kernels.h): what
// triangle_test would compute twice -- determinant, t, the range test, the hit point -- is computed once.  The second test's
// range check "t <= closest" holds either way: closest is unchanged, or the first triangle was just accepted at this t (and
// the second, accepted too, wins the tie as the later object: world.rs:281-287).  Same results as two triangle_test calls.
// Record (pt_scene_upload): r0 = (n, -), r1 = (v0, -), r2 = (N1, N2.x), r3 = (N2.y, N2.z, N1'.x, N1'.y), r4 = (N1'.z, N2').
template <bool ANY = false>
PT_DEV void tripair_test(float4 r0, float4 r1, float4 r2, float4 r3, float4 r4, f3 o, f3 d, float t_min, float& closest, int& id, int obj) {
    const f3 n = mk(r0.x, r0.y, r0.z);
    const float det = dot(d, n);
#if PT_PAIR_S_EARLY
    // measurement variant: o - v0 before the determinant's test, so that v0 is requested together with the normal (one dependent LDS
    // latency less per pair; three subtractions more for the waves whose rays are all parallel to the plane: none in practice)
    f3 s = o - mk(r1.x, r1.y, r1.z);
    asm volatile("" : "+v"(s.x), "+v"(s.y), "+v"(s.z));
    if (__builtin_fabsf(det) < 1e-8f) return;
#else
    if (__builtin_fabsf(det) < 1e-8f) return;
    const f3 s = o - mk(r1.x, r1.y, r1.z);
#endif
    const float t = pt_div(-dot(s, n), det);
    if (t < t_min || t > closest) return;
    const f3 p = madd(d, t, s);
#if PT_PAIR_FLAT
    // measurement variant: both triangles' barycentric tests without branches (same predicates on the same values; a wave of
    // incoherent rays nearly always has a lane inside each u range, so the branches skip little and cost scalar instructions)
    {
        const float u0 = dot(p, mk(r2.x, r2.y, r2.z)), v0 = dot(p, mk(r2.w, r3.x, r3.y));
        const float u1 = dot(p, mk(r3.z, r3.w, r4.x)), v1 = dot(p, mk(r4.y, r4.z, r4.w));
        const bool acc0 = (u0 >= 0.0f && u0 <= 1.0f) && !(v0 < 0.0f || u0 + v0 > 1.0f);
        const bool acc1 = (u1 >= 0.0f && u1 <= 1.0f) && !(v1 < 0.0f || u1 + v1 > 1.0f);
        if (ANY) { if (acc0 || acc1) id = 0; return; }
        closest = (acc0 || acc1) ? t : closest;
        id = acc1 ? obj + 1 : acc0 ? obj : id;
        return;
    }
#endif
    const float u0 = dot(p, mk(r2.x, r2.y, r2.z));
    if (in_unit_range(u0)) {
        const float v0 = dot(p, mk(r2.w, r3.x, r3.y));
        if (!(v0 < 0.0f || u0 + v0 > 1.0f)) {
            if (ANY) { id = 0; return; }
            closest = t; id = obj;
        }
    }
    const float u1 = dot(p, mk(r3.z, r3.w, r4.x));
    if (in_unit_range(u1)) {
        const float v1 = dot(p, mk(r4.y, r4.z, r4.w));
        if (!(v1 < 0.0f || u1 + v1 > 1.0f)) {
            if (ANY) { id = 0; return; }
            closest = t; id = obj + 1;
        }
    }
}

// sphere_test in two halves: the part every sphere pays (half_b, discriminant) and the part an accepted discriminant
// pays.  Same operations in the same order per sphere; split so that a group of four can run the first halves
// back to back (four independent dependency chains) before the divergent second halves.
PT_DEV void sphere_pre(float4 s, f3 o, f3 d, float& half_b, float& disc) {
    f3 oc = o - mk(s.x, s.y, s.z);
    half_b = dot(oc, d);
    f3 l = madd(d, -half_b, oc);
    disc = s.w - dot(l, l);
}
template <bool ANY>
PT_DEV void sphere_post(float half_b, float disc, float t_min, float& closest, int& id, int obj) {
#if PT_SPHERE_FLAT
    // measurement variant: no branch on the discriminant (the root of a negative one is a NaN nobody reads)
    {
        const float sq = pt_sqrt(disc);
        const float ra = -half_b - sq, rb = -half_b + sq;
        const float cc = ra < t_min ? rb : ra;
        const bool rej = disc < 0.0f || cc < t_min || closest < cc;
        if (ANY) { id = rej ? id : 0; return; }
        closest = rej ? closest : cc;
        id = rej ? id : obj;
        __builtin_amdgcn_sched_barrier(0);      // (one test after the other: interleaved, four of them need 30 registers more)
        return;
    }
#endif
    if (disc < 0.0f) return;
    float sqrtd = pt_sqrt(disc);
    float root1 = -half_b - sqrtd;
    float root2 = -half_b + sqrtd;
    float c = root1 < t_min ? root2 : root1;
    if (c < t_min || closest < c) return;
    if (ANY) { id = 0; return; }
    closest = c;
    id = obj;
}
// disc of SphereShape::hit only (the part every sphere pays), see sphere_test
PT_DEV float sphere_disc(float4 s, f3 o, f3 d) {
    f3 oc = o - mk(s.x, s.y, s.z);
    f3 l = madd(d, -dot(oc, d), oc);
    return s.w - dot(l, l);
}

// GROUPED (large scenes, where a given sphere is rarely hit): four discriminants, ONE wave-uniform
// branch "did any lane hit any of the four?" instead of a divergent branch per sphere; the exact
// sequential tests run only then.  max() drops NaNs unless all four are NaN, which is exactly the
// NaN-ray case the reference lets through (Q10), so a NaN still reaches sphere_test.
template <bool GROUPED, bool ANY = false, int PF = 0>      // PF: pair records requested ahead (PT_PAIR_PREFETCH; the split kernel only)
PT_DEV void scan_run(const float4* __restrict__ p, uint32_t tag, uint32_t n, int first_obj, f3 o, f3 d, float t_min,
                     float& closest, int& id) {
    if (tag == SHAPE_SPHERE) {
        // four LDS reads in flight per wait instead of one
        uint32_t i = 0;
        for (; i + 4u <= n; i += 4u) {
            float4 s0 = p[i], s1 = p[i + 1], s2 = p[i + 2], s3 = p[i + 3];
            if (GROUPED) {
                float m = __builtin_fmaxf(__builtin_fmaxf(sphere_disc(s0, o, d), sphere_disc(s1, o, d)),
                                          __builtin_fmaxf(sphere_disc(s2, o, d), sphere_disc(s3, o, d)));
                if (__ballot(!(m < 0.0f)) == 0ull) continue;
            }
            if (!GROUPED) {
                // the four discriminants first, then the four root parts: four independent dependency chains for the
                // scheduler instead of one test after the other (same arithmetic; same-box A/B on C2: -0.8 %)
                float h0, h1, h2, h3, d0, d1, d2, d3;
                sphere_pre(s0, o, d, h0, d0); sphere_pre(s1, o, d, h1, d1); sphere_pre(s2, o, d, h2, d2); sphere_pre(s3, o, d, h3, d3);
                sphere_post<ANY>(h0, d0, t_min, closest, id, first_obj + (int)i);
                sphere_post<ANY>(h1, d1, t_min, closest, id, first_obj + (int)i + 1);
                sphere_post<ANY>(h2, d2, t_min, closest, id, first_obj + (int)i + 2);
                sphere_post<ANY>(h3, d3, t_min, closest, id, first_obj + (int)i + 3);
                continue;
            }
            sphere_test<false, ANY>(s0, o, d, t_min, closest, id, first_obj + (int)i);
            sphere_test<false, ANY>(s1, o, d, t_min, closest, id, first_obj + (int)i + 1);
            sphere_test<false, ANY>(s2, o, d, t_min, closest, id, first_obj + (int)i + 2);
            sphere_test<false, ANY>(s3, o, d, t_min, closest, id, first_obj + (int)i + 3);
        }
#if PT_SPHERE_REM2
        // of the last (n mod 4) records two reads in flight at once instead of one read per test (round 5: C2's ten spheres are two
        // groups and two; launch 5.58 -> 5.52 ms, profiles/r05/ab_lds_latency.txt; three at once for n mod 4 = 3 spills four registers)
        if (!GROUPED && i + 2u <= n) {
            const float4 s0 = p[i], s1 = p[i + 1];
            float h0, h1, d0, d1;
            sphere_pre(s0, o, d, h0, d0); sphere_pre(s1, o, d, h1, d1);
            sphere_post<ANY>(h0, d0, t_min, closest, id, first_obj + (int)i);
            sphere_post<ANY>(h1, d1, t_min, closest, id, first_obj + (int)i + 1);
            i += 2u;
        }
#endif
        for (; i < n; ++i) sphere_test<false, ANY>(p[i], o, d, t_min, closest, id, first_obj + (int)i);
    } else if (tag == kRunTriangle) {
        for (uint32_t i = 0; i < n; ++i) {
            float4 a0 = p[3 * i], a1 = p[3 * i + 1], a2 = p[3 * i + 2];
            triangle_test<false, ANY>(a0, a1, a2, o, d, t_min, closest, id, first_obj + (int)i);
        }
    } else {
        if (PF > 0) {
            // the next pair's plane normal is requested while this pair is tested (one of the three dependent LDS latencies of a pair
            // test off the critical path, for three registers: kernels with registers to spare only -- k_paths_regen_split)
            float4 a0n = p[0], a1n = p[1];
            for (uint32_t i = 0; i < n; ++i) {
                const float4 a0 = a0n;
                float4 a1 = PF > 1 ? a1n : p[5 * i + 1];
                if (i + 1u < n) { a0n = p[5 * i + 5]; if (PF > 1) a1n = p[5 * i + 6]; }
                float4 a2 = p[5 * i + 2], a3 = p[5 * i + 3], a4 = p[5 * i + 4];
                tripair_test<ANY>(a0, a1, a2, a3, a4, o, d, t_min, closest, id, first_obj + 2 * (int)i);
            }
        } else {
            for (uint32_t i = 0; i < n; ++i) {
                float4 a0 = p[5 * i], a1 = p[5 * i + 1], a2 = p[5 * i + 2], a3 = p[5 * i + 3], a4 = p[5 * i + 4];
                tripair_test<ANY>(a0, a1, a2, a3, a4, o, d, t_min, closest, id, first_obj + 2 * (int)i);
            }
        }
    }
}

// How a kernel finds the closest hit:
//   kModeLds    scenes of <= kSmallObjs objects: everything is in LDS (the blob of SceneView, copied once per
//               workgroup), linear scan
//   kModeTiled  larger scenes: the scan array streams through one LDS tile (block-uniform loop, barriers), the
//               per-object records are gathered from global memory, linear scan
//   kModeBvh    accel = 1: per-lane BVH traversal out of global memory / L2 (stack in LDS, no barriers)
constexpr int kModeLds = 0, kModeTiled = 1, kModeBvh = 2;
struct SceneRef {
    const float4* scan;     // SMALL: LDS scan array; else: the LDS tile buffer
    const float4* shape;
    const float4* mat;
    const Run* runs;
    const uint32_t* lights;
    const float4* scan_global;
    const Run* runs_global;
    uint32_t n_runs, n_lights;
    BvhView bvh;
    uint32_t* stack;        // kModeBvh: LDS traversal stack, entry e of thread t at stack[e * kBlock + t]
#if PT_RUN0_SGPR
    Run run0;               // kModeLds: the first run record, read once per kernel into scalar registers (measured, round 5: C1 +1.7 %, C2 +0.3 %: rejected)
#endif
};
template <int MODE>
PT_DEV SceneRef stage_scene(const SceneView& sc, float4* lds) {
    SceneRef r;
    r.n_runs = sc.n_runs; r.n_lights = sc.n_lights;
    r.scan_global = sc.scan;
    r.runs_global = sc.runs;
    r.bvh = sc.bvh;
    r.stack = reinterpret_cast<uint32_t*>(lds);
    if (MODE == kModeLds) {
        for (uint32_t k = threadIdx.x; k < sc.blob_f4; k += blockDim.x) lds[k] = sc.blob[k];      // (k_paths_regen runs smaller workgroups)
        __syncthreads();
        r.scan = lds;
        r.shape = lds + sc.scan_f4;
        r.mat = lds + sc.scan_f4 + 3u * sc.n_objs;
        r.runs = reinterpret_cast<const Run*>(lds + sc.scan_f4 + 5u * sc.n_objs);
        r.lights = reinterpret_cast<const uint32_t*>(lds + sc.scan_f4 + 5u * sc.n_objs + sc.n_runs);
#if PT_RUN0_SGPR
        r.run0 = r.runs[0];
        r.run0.tag = __builtin_amdgcn_readfirstlane(r.run0.tag); r.run0.first_obj = __builtin_amdgcn_readfirstlane(r.run0.first_obj);
        r.run0.count = __builtin_amdgcn_readfirstlane(r.run0.count); r.run0.off4 = __builtin_amdgcn_readfirstlane(r.run0.off4);
#endif
    } else {
        r.scan = lds;
        r.shape = sc.shape; r.mat = sc.mat; r.runs = sc.runs; r.lights = sc.lights;
    }
    return r;
}

// hit_scene by BVH traversal (traverse_segment below).  Every primitive whose test the linear scan would have
// accepted is still tested: a subtree is skipped only if the ray misses its box, enlarged by `pad` on every
// side, inside [t_min, closest].  pad = 2^-15 (|o|_1 + scene extent) is ~100x the rounding error of the
// primitive tests (their error scales with the distance between ray origin and primitive), so a hit that exists
// only through rounding (a grazing ray) is inside the padded box as well.  Rays with a non-finite coordinate or
// a zero direction (the reference lets NaN through its sphere test, Q10) take the linear scan, from global
// memory (scan_global).  A visibility query only uses "is there a hit": the lane stops at its first accepted
// primitive.
constexpr float kBvhPad = 1.0f / 32768.0f;
PT_DEV void scan_global(const SceneRef& sc, f3 o, f3 d, float t_min, float t_max, int& id_out, float& t_out);

// World::hit_scene (world.rs:270-290): linear scan in object order with a
// shrinking t_max.  kModeLds: the whole scan array already sits in LDS.  kModeTiled:
// every run is streamed through one LDS tile; the loop is block-uniform (all
// threads of the workgroup call this together, active or not).
template <int MODE, bool ANY = false, int PF = 0>
PT_DEV void scan_closest(const SceneRef& sc, f3 o, f3 d, float t_min, float t_max, int& id_out, float& t_out) {
    constexpr bool SMALL = MODE == kModeLds;
    float closest = t_max;
    int id = -1;
    for (uint32_t r = 0; r < sc.n_runs; ++r) {
#if PT_RUN0_SGPR
        Run run = sc.run0;
        if (!SMALL || r != 0u) run = sc.runs[r];
#else
        Run run = sc.runs[r];
#endif
        // the run record is the same in every lane: keep it (and the loop counters and object indices derived
        // from it) in scalar registers
        run.tag = __builtin_amdgcn_readfirstlane(run.tag); run.first_obj = __builtin_amdgcn_readfirstlane(run.first_obj);
        run.count = __builtin_amdgcn_readfirstlane(run.count); run.off4 = __builtin_amdgcn_readfirstlane(run.off4);
        const uint32_t per = run_entry_f4(run.tag);
        if (SMALL) {
            scan_run<false, ANY, PF>(sc.scan + run.off4, run.tag, run.count, (int)run.first_obj, o, d, t_min, closest, id);
        } else {
            float4* tile = const_cast<float4*>(sc.scan);
            const uint32_t tile_prims = kTileF4 / per;
            for (uint32_t p0 = 0; p0 < run.count; p0 += tile_prims) {
                uint32_t np = run.count - p0 < tile_prims ? run.count - p0 : tile_prims;
                __syncthreads();
                const float4* src = sc.scan_global + run.off4 + p0 * per;
                for (uint32_t k = threadIdx.x; k < np * per; k += kBlock) tile[k] = src[k];
                __syncthreads();
                scan_run<true, ANY>(tile, run.tag, np, (int)(run.first_obj + p0 * (run.tag == kRunTrianglePair ? 2u : 1u)), o, d, t_min, closest, id);
            }
        }
    }
    id_out = id;
    t_out = closest;
}
// the same scan with every record read from global memory (no LDS, no barrier: any subset of lanes may call it)
PT_DEV void scan_global(const SceneRef& sc, f3 o, f3 d, float t_min, float t_max, int& id_out, float& t_out) {
    float closest = t_max;
    int id = -1;
    for (uint32_t r = 0; r < sc.n_runs; ++r) {
        const Run run = sc.runs_global[r];
        scan_run<false>(sc.scan_global + run.off4, run.tag, run.count, (int)run.first_obj, o, d, t_min, closest, id);
    }
    id_out = id;
    t_out = closest;
}

// Ray given to lanes that carry no path (or need no shadow ray).  It must FAIL every sphere
// discriminant with a finite negative number: a zero ray gives a = 0, 1/a = inf, disc = NaN, and a
// NaN falls through to the hit branch (reference semantics, Q10) -- one dead lane then drags its
// whole wave through the sqrt/root logic of every sphere (measured on C4: 0.6 transcendental
// instructions per sphere test).  From 3e18 along +x every |oc - (oc.d)d|^2 is ~1.8e37.
PT_DEV f3 parked_origin() { return mk(3e18f, 3e18f, 3e18f); }
PT_DEV f3 parked_dir() { return mk(1.0f, 0.0f, 0.0f); }

// number of set bits of a wave mask below this lane (v_mbcnt: no lane-mask registers to keep)
PT_DEV uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// n / d and n % d for n < 2^32 with the host's magic = floor(2^32 / d) (d >= 2; 0xFFFFFFFF for d = 1): umulhi is at
// most one below the quotient.  Replaces the compiler's division sequence, whose reciprocals sat in VGPRs for the
// whole kernel.
PT_DEV void divmod_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t& q, uint32_t& r) {
    q = __umulhi(n, magic);
    r = n - q * d;
    if (r >= d) { q += 1u; r -= d; }
}
// tile row -> image row (TileMap)
PT_DEV uint32_t image_row(const TileMap& t, uint32_t yl) {
    uint32_t q = t.band_rows == 1u ? yl : __umulhi(yl, t.band_magic);
    return q * t.band_stride + t.band_first + (yl - q * t.band_rows);
}

// End of a wave: its statistics go to the launch's totals.  Every wave adding them to the same five global words itself is
// ~30 000 atomics on ONE cache line per launch, serialised in L2 at the very end of the launch, where every microsecond is tail
// (measured: the fifth word, the finished-sample count, alone cost 1.5 % of a C2 launch: profiles/r05/ab_count_finished.txt).
// So the waves of a workgroup add up in LDS first and the LAST of them to end does the global atomics: a quarter of the traffic.
#ifndef PT_WG_TOTALS
#define PT_WG_TOTALS 1
#endif

struct WgTotals { uint32_t done, shadow, vertices, samples, dmax; };
PT_DEV void wg_totals_init(WgTotals& t) {                 // by one thread, before the workgroup's first barrier
    t.done = 0u; t.shadow = 0u; t.vertices = 0u; t.samples = 0u; t.dmax = 0u;
}
// called by lane 0 of every wave that ran (spare workgroups that end at once never get here); waves_in_block of them
template <bool MIS, bool PRIMARY = true>     // PRIMARY: a level-0 launch (its vertices also count as primary_vertices)
PT_DEV void wave_totals(WgTotals& t, uint32_t waves_in_block, unsigned long long* stats, uint32_t shadow, uint32_t vertices,
                        uint32_t samples, uint32_t dmax) {
#if PT_WG_TOTALS
    if (MIS && shadow != 0u) __hip_atomic_fetch_add(&t.shadow, shadow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (vertices != 0u) __hip_atomic_fetch_add(&t.vertices, vertices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (samples != 0u) __hip_atomic_fetch_add(&t.samples, samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (vertices != 0u) __hip_atomic_fetch_max(&t.dmax, dmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // (acq_rel: the sums of the waves that ended earlier are visible to the one that finds itself last)
    if (__hip_atomic_fetch_add(&t.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u != waves_in_block) return;
    shadow = __hip_atomic_load(&t.shadow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    vertices = __hip_atomic_load(&t.vertices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    samples = __hip_atomic_load(&t.samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    dmax = __hip_atomic_load(&t.dmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    if (MIS && shadow != 0u) atomicAdd(&stats[0], (unsigned long long)shadow);
    if (vertices != 0u) atomicAdd(&stats[1], (unsigned long long)vertices);
    if (PRIMARY && vertices != 0u) atomicAdd(&stats[3], (unsigned long long)vertices);
    if (vertices != 0u) atomicMax(&stats[2], (unsigned long long)dmax);
    if (samples != 0u) atomicAdd(&stats[4], (unsigned long long)samples);
}

// ------------------------------------------------------------------ one path vertex
// The per-vertex body of MisStrategy::ray_color / BrdfOnlyStrategy::ray_color (rendering.rs:34-142,
// 214-265), cut at the visibility scan: vertex_begin (hit record, emitter credit, light sample) -> shadow
// scan -> vertex_end (NEE term, BSDF sample, roulette, next ray).
struct PathState {
    f3 o, d, beta, L;
    float pdf_prev, eta_in;
    uint32_t s_local, depth, px, yl;
};
struct Vertex {
    Hit hit;
    Mat m;
    bool alive;              // the path continues past this vertex (so far)
    bool need_shadow;        // a light point was sampled: visibility of light_dir up to distance is needed
    f3 light_dir, ls_emission;
    float distance, ls_pdf;
    uint32_t w_bsdf1, w_bsdf2;   // the vertex's BSDF words of BLK_SURFACE (drawn together with the light words)
    uint32_t w_lobe;             // its lobe word of BLK_CHOICE, when vertex_begin had to draw that block (several lights)
    uint32_t w_rr;               // its roulette word (made of the BLK_SURFACE bits u01() skips)
    int obj, light_obj;          // object hit (>= 0) and light picked: vertex_end can re-read their records (REMAT)
    bool hit_emitter;            // the path ray reached an emitter: vertex_end credits it (it needs the carry state)
    float emit_pdf_shape;        // ... with the light pdf of that point seen from the previous vertex (MIS, depth > 0)
};

// Queue planes (pt_kernels.h): the RAY part of the state -- what the closest-hit scan and the light sample need --
// is planes 0 and 1; the CARRY part -- throughput, radiance so far, the previous sampling pdf, the incoming eta --
// is planes 2 and 3 and is only looked at once the vertex's scans are through (k_paths loads it that late, so those
// eight values do not occupy registers during the scans).
PT_DEV void unpack_ray(PathState& p, float4 q0, float4 q1) {
    p.o = mk(q0.x, q0.y, q0.z); p.d = mk(q0.w, q1.x, q1.y);
    const uint32_t xy = __float_as_uint(q1.z), sd = __float_as_uint(q1.w);
    p.yl = xy >> 16; p.px = xy & 0xFFFFu;
    p.s_local = sd >> 16; p.depth = sd & 0xFFFFu;
}
PT_DEV void unpack_carry(PathState& p, float4 q2, float4 q3) {
    p.beta = mk(q2.x, q2.y, q2.z); p.pdf_prev = q2.w;
    p.L = mk(q3.x, q3.y, q3.z); p.eta_in = q3.w;
}
PT_DEV PathState unpack_state(float4 q0, float4 q1, float4 q2, float4 q3) {
    PathState p;
    unpack_ray(p, q0, q1);
    unpack_carry(p, q2, q3);
    return p;
}
PT_DEV void store_state(const Queue& q, uint32_t j, const PathState& p) {
    q.q[0][j] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
    q.q[1][j] = make_float4(p.d.y, p.d.z, __uint_as_float((p.yl << 16) | p.px), __uint_as_float((p.s_local << 16) | p.depth));
    q.q[2][j] = make_float4(p.beta.x, p.beta.y, p.beta.z, p.pdf_prev);
    q.q[3][j] = make_float4(p.L.x, p.L.y, p.L.z, p.eta_in);
}
PT_DEV PathState parked_state() {
    PathState p;
    p.o = parked_origin(); p.d = parked_dir(); p.beta = mk(1.f, 1.f, 1.f); p.L = mk(0.f, 0.f, 0.f);
    p.pdf_prev = 0.0f; p.eta_in = 1.0f;
    p.s_local = 0; p.depth = 0; p.px = 0; p.yl = 0;
    return p;
}

// Camera::get_ray_with_offset for sample `sample` of pixel (px, py) (camera.rs:139-147; jitter draws world.rs:299)
PT_DEV void camera_ray(const CameraF& cam, uint32_t sample, uint32_t px, uint32_t py, f3& o, f3& d) {
    uint32_t dc[4];
    philox4x32_draw(px, py, sample, kDepthCamera, BLK_SURFACE, 0u, dc);
    float ox = u01(dc[0]), oy = u01(dc[1]);                               // world.rs:299 (ox first)
    // (The divisors go through an empty asm: the compiler otherwise hoists their reciprocals out of the path loop of the
    // regenerating kernels into two registers that live -- or are spilled -- for the whole kernel; camera rays are generated once
    // per 64-path chunk, two reciprocals there cost nothing.  Same arithmetic.)
    float wm1 = (float)(cam.width - 1u), hm1 = (float)(cam.height - 1u);
    asm volatile("" : "+v"(wm1), "+v"(hm1));
    float u = pt_div((float)px + ox, wm1);                                   // camera.rs:140
    float v = pt_div((float)(cam.height - 1u - py) + oy, hm1);               // world.rs:299 y flip
    const f3 cam_o = mk(cam.origin[0], cam.origin[1], cam.origin[2]);
    f3 dir = mk(cam.lower_left[0], cam.lower_left[1], cam.lower_left[2]) +
             mk(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]) * u +
             mk(cam.vertical[0], cam.vertical[1], cam.vertical[2]) * v - cam_o;   // camera.rs:143-144
    o = cam_o;
    d = normalize(dir);                                                   // Ray::new, camera.rs:13
}

// World::sample_light_point (world.rs:251-267) from `from`: w_index = the light-index word, w_r1 / w_r2 = the surface words.
// n_lights > 0.
// dir / dist: unit direction and distance from `from` to the point (rendering.rs:58-60), from the sampler itself.
// Material sets a kernel (or a part of one) is compiled for: vertex_begin / vertex_end / sample_light_point take one as
// their second template argument (bool DIFFUSE converts: false = every material, true = Lambertian + emissive only).
//   kMatsAll       every material
//   kMatsDiffuse   Lambertian and emissive only (scene property, decided at pt_scene_upload): no GGX, no OrenNayar code
//   kMatsNoMirror  everything but Mirror (the plain iterations of k_paths_regen_split, which hand Mirror vertices on)
//   kMatsMirror    the object HIT is a Mirror (the batches of k_paths_regen_split: every entry of the special stack is one);
//                  says nothing about the light's material
constexpr int kMatsAll = 0, kMatsDiffuse = 1, kMatsNoMirror = 2, kMatsMirror = 3;
template <int MATS>
PT_DEV void assume_mats(uint32_t tag) {
    if (MATS == kMatsDiffuse) __builtin_assume(tag <= MAT_EMISSIVE);
    if (MATS == kMatsNoMirror) __builtin_assume(tag != MAT_MIRROR);
    if (MATS == kMatsMirror) __builtin_assume(tag == MAT_MIRROR);
}
template <int DIFFUSE>
PT_DEV void sample_light_point(const SceneRef& sc, f3 from, uint32_t w_index, uint32_t w_r1, uint32_t w_r2, f3& point,
                               int& lobj, f3& emission, float& pdf, f3& dir, float& dist) {
    const uint32_t li = __umulhi(w_index, sc.n_lights);                           // random_range(0..n), world.rs:255
    lobj = (int)sc.lights[li];
    const Mat lm = load_mat(sc.mat, lobj);
    if (DIFFUSE != kMatsMirror) assume_mats<DIFFUSE>(lm.tag);
    float pdf_shape;
    shape_sample(sc.shape, sc.mat, lobj, lm.shape_tag, from, false, from, u01(w_r1), u01(w_r2), point, pdf_shape, dir, dist);
    emission = lm.color;                                                          // world.rs:259
    pdf = sc.n_lights == 1u ? pdf_shape : pt_div(pdf_shape, (float)sc.n_lights);  // world.rs:260 (x/1 == x)
}

// (id, t) = closest hit of the path ray, id < 0: miss.  Notes an emitter hit and samples the light point.  Reads only the
// RAY part of p (origin, direction, depth, film position).
// DIFFUSE: the scene has Lambertian and emissive materials only (decided at pt_scene_upload); the GGX and
// OrenNayar code is then compiled out of the kernel (same results; smaller code, no spills at 6 waves/SIMD: C2 +2 %).
// (kx, py) = the pixel's RNG key (main.rs:51); it is the path's film position except in pixel-list renders.
template <bool MIS, int DIFFUSE>
PT_DEV void vertex_begin(const SceneRef& sc, PathState& p, bool active, int id, float t, uint32_t sample, uint32_t kx,
                         uint32_t py, Vertex& v) {
    v.alive = active && id >= 0;
    v.obj = id >= 0 ? id : 0; v.light_obj = 0;
    v.hit_emitter = false; v.emit_pdf_shape = 0.0f;
    v.hit.point = p.o; v.hit.normal = p.d; v.hit.t = 0.0f; v.hit.front_face = false;
    v.m.tag = MAT_LAMBERT; v.m.shape_tag = 0; v.m.emits = 0; v.m.color = mk(0.f, 0.f, 0.f);
    v.m.roughness = 0.f; v.m.metallic = 0.f; v.m.ior = 1.f; v.m.on_a = 1.f; v.m.on_b = 0.f;
    if (v.alive) {
        v.m = load_mat(sc.mat, id);
        assume_mats<DIFFUSE>(v.m.tag);
        v.hit = finish_hit(sc.shape, id, v.m.shape_tag, p.o, p.d, t);
        if (v.m.emits) {
            v.hit_emitter = true;
            if (MIS && p.depth != 0u) {
                // emitter reached by a BSDF-sampled ray: its MIS weight (vertex_end) is against the light pdf of
                // this point seen from the previous vertex = this ray's origin (rendering.rs:107-116)
                f3 sp, sd; float sl;
                shape_sample(sc.shape, sc.mat, id, v.m.shape_tag, p.o, true, v.hit.point, 0.f, 0.f, sp, v.emit_pdf_shape, sd, sl);
            }
            v.alive = false;
        }
    }

    // ---- draws of the vertex; NEE: light pick + surface sample (world.rs:251-267)
    v.need_shadow = false;
    v.light_dir = mk(0.f, 0.f, 0.f); v.ls_emission = mk(0.f, 0.f, 0.f);
    v.distance = 0.0f; v.ls_pdf = 1.0f;
    v.w_bsdf1 = v.w_bsdf2 = v.w_lobe = v.w_rr = 0u;
    if (v.alive) {
        uint32_t ds[4];
        philox4x32_draw(kx, py, sample, p.depth, BLK_SURFACE, 0u, ds);
        v.w_bsdf1 = ds[2]; v.w_bsdf2 = ds[3];
        v.w_rr = (ds[0] << 23) | ((ds[1] & 0x1FFu) << 14) | ((ds[2] & 0x1FFu) << 5);   // roulette word: the bits of the block u01() skips
        if (MIS && sc.n_lights > 0u) {
            uint32_t w_index = 0u;                                                // umulhi(u, 1) = 0: one light needs no draw
            if (sc.n_lights > 1u) {
                uint32_t dc[4];
                philox4x32_draw(kx, py, sample, p.depth, BLK_CHOICE, 0u, dc);
                w_index = dc[0]; v.w_lobe = dc[1];
            }
            f3 lp;                                                                // rendering.rs:58-60: direction and distance
            sample_light_point<DIFFUSE>(sc, v.hit.point, w_index, ds[0], ds[1], lp, v.light_obj, v.ls_emission, v.ls_pdf,
                                        v.light_dir, v.distance);                 // to the point, from the sampler itself
            v.need_shadow = true;
        }
    }
}

// visible: the shadow scan found nothing between the vertex and the light point.  Returns "the path goes on";
// p is then the state at the next vertex.
// REMAT (scene in LDS): the material of the hit object and the light's emission are read again here instead of
// being carried across the visibility scan -- two broadcast LDS reads instead of ~6 live registers, which is what
// keeps the kernel at 80 VGPRs without spills.
template <bool MIS, int DIFFUSE, bool REMAT>
PT_DEV bool vertex_end(const SceneRef& sc, PathState& p, const Vertex& vin, bool visible, uint32_t sample, uint32_t kx,
                       uint32_t py, uint32_t min_depth, uint32_t max_depth) {
    const uint32_t n_lights = sc.n_lights;
    Vertex v = vin;
    if (REMAT) {
        asm volatile("" ::: "memory");          // a real re-read, not the values of vertex_begin kept alive
        v.m = load_mat(sc.mat, vin.obj);
        v.ls_emission = load_mat(sc.mat, vin.light_obj).color;
    }
    if (DIFFUSE != kMatsMirror) assume_mats<DIFFUSE>(v.m.tag);
    else if (vin.alive) assume_mats<DIFFUSE>(v.m.tag);         // (a lane without a path re-reads object 0's material)
    if (vin.hit_emitter) {
        if (!MIS || p.depth == 0u) {
            p.L = p.L + p.beta * v.m.color;                                       // rendering.rs:44-45 / :225-227
        } else {
            float w_bsdf = pt_div(p.pdf_prev, p.pdf_prev + vin.emit_pdf_shape);   // :117 (Q2: not / n_lights)
            p.L = p.L + p.beta * v.m.color * w_bsdf;                              // :119-121
        }
    }
    f3 direct = mk(0.f, 0.f, 0.f);
    if (MIS && visible) {
        float cos_theta = __builtin_fabsf(dot(v.hit.normal, v.light_dir));    // rendering.rs:68
        f3 bsdf; float pdf_bsdf;
        bsdf_pdf(v.m, p.d, p.eta_in, v.light_dir, v.hit.normal, bsdf, pdf_bsdf);   // :71-72 (stale eta, Q5)
        float w_nee = pt_div(v.ls_pdf, v.ls_pdf + pdf_bsdf);                       // :73
        direct = w_nee * bsdf * v.ls_emission * cos_theta / v.ls_pdf;         // :75-76
    }

    // ---- BSDF sample, throughput, Russian roulette (rendering.rs:83-102)
    bool alive = v.alive;
    if (alive) {
        // BLK_CHOICE: already drawn by vertex_begin when the scene has several lights; otherwise only a Mirror
        // surface (lobe) reads it
        uint32_t w_lobe = v.w_lobe, w_rr = v.w_rr;
        if (!(MIS && n_lights > 1u) && v.m.tag == MAT_MIRROR) {
            uint32_t dc[4];
            philox4x32_draw(kx, py, sample, p.depth, BLK_CHOICE, 0u, dc);
            w_lobe = dc[1];
        }
        float eta_mat = v.m.tag == MAT_MIRROR ? v.m.ior : 1.0f;               // get_eta, material.rs:50 / mirror.rs:317
        float eta_here = v.hit.front_face ? pt_rcp(eta_mat) : eta_mat;         // rendering.rs:20-25
        f3 wo, bsdf; float pdf, cos_theta;
        bsdf_pdf_sample(v.m, p.d, eta_here, v.hit.normal, v.w_bsdf1, v.w_bsdf2, w_lobe, wo, bsdf, pdf, cos_theta);   // :84-85
        f3 next_tp = p.beta * bsdf * cos_theta / pdf;                         // :89
        float rr = rr_prob(p.depth, min_depth, max_depth, next_tp);           // :91-98
        if (u01(w_rr) > rr) {                                                 // :100-102 (drops direct, Q1)
            alive = false;
        } else {
            p.L = p.L + p.beta * direct;
            p.beta = rr == 1.0f ? next_tp : next_tp / rr;                     // :129 (x * (1/1) == x exactly)
            if (is_zero(p.beta) || p.depth >= 65534u) {                       // Q7: nothing downstream contributes
                alive = false;
            } else {
                p.pdf_prev = pdf;
                p.o = v.hit.point;
                p.d = v.m.tag == MAT_EMISSIVE ? normalize(wo) : wo;           // Ray::new, :86; every sampler but
                                                                              // Emissive's returns a normalised wo
                p.eta_in = eta_here;                                          // :87
                p.depth += 1u;
            }
        }
    }
    return alive;
}

// ------------------------------------------------------------------ the path kernel
// Queue organisation.  The path queue is cut into one PRIVATE segment per wave
// (segment w = slots [w*seg_cap, (w+1)*seg_cap)).  A wave reads its segment 64
// slots at a time (one coalesced 1 KiB access per plane), advances those paths by
// one vertex and writes the survivors back INTO THE SAME SEGMENT at its running
// output position: rank = popcount(ballot(alive) & lanemask_lt), position kept in
// a wave-uniform register.  Writes never pass the read position (out <= in), so
// the compaction is in place, needs no second queue and no global atomic.  (A
// single shared tail counter costs one returning atomic per wave per iteration:
// measured 59 ms of a 60 ms render at 1024^2 x 64 spp.)
//
// Because no wave ever touches another wave's slots, nothing forces the waves to
// advance bounce by bounce in lockstep: ONE launch runs every bounce of a batch.
// Each wave loops { pass over its segment = one more vertex for each of its paths }
// until its segment is empty.  (One launch per bounce cost ~2.5 of 13.2 ms in launch
// gaps, host polling and under-filled tail launches.)  Pass 0 deals 64-path chunks
// round-robin to the waves (chunk k -> wave k % nw) and generates the camera rays
// (Camera::get_ray_with_offset), so every segment samples the whole image and the
// waves finish together.
//
// Memory access of one iteration: the chunk's state is loaded at the top (4 x 16 B per lane, coalesced) and the
// survivors are stored at the bottom; with a scene in LDS nothing else touches global memory.  (An earlier version
// requested the NEXT chunk's state one iteration ahead.  The register allocator had to keep those 16 registers
// somewhere for a whole vertex, placed the copies -- and so the wait -- right after the loads anyway, and the
// pressure cost a wave of occupancy: without it the kernel needs 80-90 VGPRs instead of 115-128 and runs 6 waves
// per SIMD, C2 10.14 -> 9.78 ms, C1 16.1 -> 14.9 ms.)
// minimum waves per SIMD the register allocator must leave room for.  Scene in LDS: 80 VGPRs (the DIFFUSE variant
// without spilling, the generic one with 8 spilled dwords); measured 4 / 5 / 6 / 7 waves: C2 10.14 / 9.84 / 9.78 /
// 10.04 ms, C1 16.1 / 15.2 / 14.9 ms.  Tiled scan: 5 waves (96 VGPRs + 21 spilled dwords, 30 KiB tile so that five
// workgroups fit a CU): C4 1214 -> 1106 ms; 6 waves with a 24 KiB tile: the same.
#ifndef PT_BOUNCE_WAVES_LDS
#define PT_BOUNCE_WAVES_LDS 6
#endif
#ifndef PT_BOUNCE_WAVES_TILED
#define PT_BOUNCE_WAVES_TILED 5
#endif
// film position of a path -> its pixel: the RNG key (x, y) and camera pixel.  LIST: looked up in the pixel list.
// A lane without a path may carry stale slot contents as its film position (k_paths reads whole chunks): it must not
// index the list with them.
template <bool LIST>
PT_DEV void pixel_key(const BounceArgs& a, const PathState& p, bool active, uint32_t& kx, uint32_t& py) {
    if (LIST) { const uint2 k = a.pixels[active ? ((p.yl << 16) | p.px) : 0u]; kx = k.x; py = k.y; }
    else { kx = p.px; py = image_row(a.tile, p.yl); }
}
// a continuation launch whose path count is only known on the device: count, chunks and segment size from there
template <bool OVF>
PT_DEV void launch_shape(const BounceArgs& a, uint32_t nw, uint32_t& n_first, uint32_t& seg_cap) {
    n_first = a.n_first; seg_cap = a.seg_cap;
    if (OVF && a.n_first_dev) {
        n_first = __builtin_amdgcn_readfirstlane(*a.n_first_dev);
        seg_cap = ((((n_first + 63u) >> 6) + nw - 1u) / nw) * 64u;
    }
}
// ... and the generic-material instances of the LDS form (GGX + OrenNayar code in the kernel): what pixel lists, pt_ray_color and
// batches of <= 2^17 paths take on the reference's own scene.  At 6 waves (80 VGPRs) they spill 14-18 registers.
#ifndef PT_BOUNCE_WAVES_LDS_GENERIC
#define PT_BOUNCE_WAVES_LDS_GENERIC 5      // round 5: 93-95 VGPRs, no spills; small jobs on World::new() 6-9 % faster than at 6 waves (profiles/r05/ab_generic_waves.txt)
#endif
template <int MODE, bool MIS, bool OVF, bool DIFFUSE, bool LIST>   // OVF: continuation launch, pass 0 reads the overflow queue
__global__ void __launch_bounds__(kBlock, MODE == kModeLds ? (DIFFUSE ? PT_BOUNCE_WAVES_LDS : PT_BOUNCE_WAVES_LDS_GENERIC) : PT_BOUNCE_WAVES_TILED)
k_paths(BounceArgs a) {
    // SMALL = "the waves of a workgroup are independent" (no barrier inside the scan): wave-private queue
    // segments.  The tiled scan ties the four waves of a workgroup together.  (kModeBvh: k_paths_bvh.)
    static_assert(MODE == kModeLds || MODE == kModeTiled, "linear-scan kernel");
    constexpr bool SMALL = MODE == kModeLds;
    extern __shared__ float4 lds[];
    __shared__ uint32_t s_iters[kBlock / 64];
    __shared__ WgTotals s_totals;
    if (threadIdx.x == 0u) wg_totals_init(s_totals);
    if (MODE != kModeLds) __syncthreads();      // (kModeLds: stage_scene's barrier publishes it)
    const SceneRef sc = stage_scene<MODE>(a.sc, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (kBlock / 64);
    // SMALL: one private segment per wave.  Tiled: the four waves of a workgroup advance in lockstep anyway
    // (barriers in the scan), so they share ONE segment and compact at workgroup level: chunk c of a pass
    // goes to wave c % 4 and only the last chunk of a pass is partial (every pass costs a full scan of the
    // whole scene per wave, however few lanes are alive).
    const uint32_t wib = threadIdx.x >> 6;                       // wave in block
    uint32_t n_first, seg_cap;
    launch_shape<OVF>(a, nw, n_first, seg_cap);
    const uint32_t seg_base = SMALL ? wave * seg_cap : blockIdx.x * (kBlock / 64) * seg_cap;
    const uint32_t n_chunks = (n_first + 63u) >> 6;                // pass 0: 64-path chunks of the batch
    const uint32_t W = a.film_w;
    uint32_t n_in = 0;                     // wave-uniform: queued paths of this wave's segment
    uint32_t wave_shadow = 0, wave_vertices = 0;
    uint32_t wave_samples = 0;             // wave-uniform: paths whose radiance this wave has written to lsamp (finished samples)
    uint32_t wave_depth = 0;               // wave-uniform: deepest vertex this wave has processed
    constexpr bool from_overflow = OVF;

    for (uint32_t pass = 0;; ++pass) {
    const bool first = pass == 0u;
    // SMALL: n_in = paths of this wave's segment; tiled: n_in = paths of the workgroup's segment (same in all waves)
    const uint32_t n_iter = first ? (n_chunks + nw - 1u) / nw : (SMALL ? (n_in + 63u) >> 6 : (n_in + kBlock - 1u) / kBlock);
    if (n_iter == 0u) break;               // SMALL: this wave is done; tiled: the whole workgroup is (uniform)
    uint32_t out_n = 0;                    // wave-uniform: survivors written so far in this pass

    const uint32_t lane_off = SMALL ? lane : wib * 64u + lane;   // position inside a chunk (64 or 256 slots)
    const uint32_t chunk_slots = SMALL ? 64u : kBlock;

    for (uint32_t it = 0; it < n_iter; ++it) {
        bool active;
        PathState p;

        if (first) {
            p = parked_state();
            const uint32_t chunk = it * nw + wave;
            const uint32_t pid = chunk * 64u + lane;
            active = chunk < n_chunks && pid < n_first;
            if (active) {
                if (from_overflow) {
                    // continuation launch: the paths are the leftovers an earlier launch exported
                    unpack_ray(p, a.ovf_in.q[0][pid], a.ovf_in.q[1][pid]);
                } else {
                    uint32_t pix;
                    divmod_magic(pid, a.np, a.np_magic, p.s_local, pix);
                    divmod_magic(pix, W, a.film_w_magic, p.yl, p.px);
                }
            }
        } else {
            // every lane loads its slot (the last chunk of a pass reads stale slots of the segment: in bounds, and a
            // lane without a path only needs a ray that hits nothing -- its other fields are never looked at)
            active = it * chunk_slots + lane_off < n_in;
            const uint32_t s0 = seg_base + it * chunk_slots + lane_off;
            unpack_ray(p, a.q.q[0][s0], a.q.q[1][s0]);
            if (!active) { p.o = parked_origin(); p.d = parked_dir(); }
        }
        uint32_t kx, py;                                  // key of the path's RNG stream = (x, y), main.rs:51
        pixel_key<LIST>(a, p, active, kx, py);
        const uint32_t sample = a.s_base + p.s_local;

        if (first && !from_overflow && active) camera_ray(a.cam, sample, kx, py, p.o, p.d);
        // the CARRY part of the state (throughput, radiance, previous pdf, incoming eta) of the slot this lane works on
        auto load_carry = [&]() {
            if (!first || from_overflow) {
                const Queue& src = first ? a.ovf_in : a.q;
                const uint32_t s1 = first ? (it * nw + wave) * 64u + lane : seg_base + it * chunk_slots + lane_off;
                if (!first || active) unpack_carry(p, src.q[2][s1], src.q[3][s1]);
            }
        };
#ifdef PT_EARLY_CARRY
        load_carry();
#endif

        wave_vertices += (uint32_t)__popcll(__ballot(active));
        // deepest vertex: in a level-0 launch every path of pass p is at depth p; only a continuation launch
        // mixes depths inside a wave and has to look at the lanes
        if (!from_overflow) {
            wave_depth = pass;
        } else if (__ballot(active && p.depth > wave_depth) != 0ull) {
            uint32_t v = active ? p.depth : 0u;
            for (int off = 32; off > 0; off >>= 1) { const uint32_t w2 = (uint32_t)__shfl_xor((int)v, off); v = w2 > v ? w2 : v; }
            wave_depth = __builtin_amdgcn_readfirstlane(v);
        }

        // ---- scan #1: closest hit of the path ray (rendering.rs:41)
        int id; float t;
        scan_closest<MODE, false, (MODE == kModeLds && !DIFFUSE) ? PT_PAIR_PREFETCH_GENERIC : 0>(sc, p.o, p.d, a.t_min, kInf, id, t);
        Vertex v;
        vertex_begin<MIS, DIFFUSE>(sc, p, active, id, t, sample, kx, py, v);

        // ---- scan #2: visibility (rendering.rs:62-65); skipped when no lane needs it
        bool visible = false;
        if (MIS) {
            bool any_shadow = SMALL ? (__ballot(v.need_shadow) != 0ull) : (__syncthreads_or(v.need_shadow) != 0);
            if (any_shadow) {
                // Ray::new (rendering.rs:62) would normalise light_dir a second time; the f32
                // arithmetic specification normalises a direction once
                f3 sdir = v.need_shadow ? v.light_dir : parked_dir();
                f3 sorg = v.need_shadow ? v.hit.point : parked_origin();
                int sid; float st;
                scan_closest<MODE, true, (MODE == kModeLds && !DIFFUSE) ? PT_PAIR_PREFETCH_GENERIC : 0>(sc, sorg, sdir, a.t_min, v.distance - a.t_min, sid, st);   // any-hit form
                visible = v.need_shadow && sid < 0;
                wave_shadow += (uint32_t)__popcll(__ballot(v.need_shadow));
            }
        }
        // ---- the carry part only now: none of it was needed -- or occupied a register -- during the two scans.
        // (The compiler barrier keeps the loads down here.)
#ifndef PT_EARLY_CARRY
        asm volatile("" ::: "memory");
        load_carry();
#endif
        const bool alive = vertex_end<MIS, DIFFUSE, SMALL>(sc, p, v, visible, sample, kx, py, a.min_depth, a.max_depth);

        // ---- retire, or compact in place into the wave's own segment
        if (active && !alive) a.lsamp[p.s_local * a.np + p.yl * W + p.px] = Rgb{p.L.x, p.L.y, p.L.z};
        wave_samples += (uint32_t)__popcll(__ballot(active && !alive));
        const unsigned long long mask = __ballot(alive);
        uint32_t cnt_before = 0, cnt_all = (uint32_t)__popcll(mask);
        if (!SMALL) {
            // workgroup-level prefix of the survivor counts (s_iters is free: the next write to it is an
            // iteration away, behind the barriers of two scans)
            if (lane == 0u) s_iters[wib] = cnt_all;
            __syncthreads();
            cnt_all = 0;
            for (uint32_t k = 0; k < kBlock / 64; ++k) { cnt_before += k < wib ? s_iters[k] : 0u; cnt_all += s_iters[k]; }
        }
        if (alive) store_state(a.q, seg_base + out_n + cnt_before + lane_rank(mask), p);
        out_n += cnt_all;
    }
    n_in = out_n;
    // the next pass reads (from other lanes of this wave -- tiled: of this workgroup) what this pass stored
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (!SMALL) __syncthreads();
    if (n_in < a.export_below) break;      // export_below >= 1: an empty segment always ends the wave (tiled:
    }   // pass loop                       // n_in and export_below are workgroup-uniform)

    // Tail hand-off.  Below one chunk a wave would run every further pass mostly empty (and one path trapped
    // in a glass sphere keeps it alive for 50 passes).  Instead it appends what is left to the global
    // overflow queue -- one atomic per wave per launch -- and retires; the host launches this kernel again
    // on that queue (from_overflow), where the leftovers of ~65 000 waves form dense chunks again.
    if (SMALL && n_in != 0u) {
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(a.ovf_out_count, n_in);
        base = __shfl(base, 0);
        for (uint32_t j = lane; j < n_in; j += 64u) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) { const float4 t = a.q.q[k][seg_base + j]; a.ovf_out.q[k][base + j] = t; }
        }
    }
    if (!SMALL && n_in != 0u) {            // tiled: the workgroup exports its shared segment (< 256 paths)
        if (threadIdx.x == 0u) s_iters[0] = atomicAdd(a.ovf_out_count, n_in);
        __syncthreads();
        const uint32_t base = s_iters[0];
        if (threadIdx.x < n_in) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) { const float4 t = a.q.q[k][seg_base + threadIdx.x]; a.ovf_out.q[k][base + threadIdx.x] = t; }
        }
    }
    // totals for the host: shadow rays, vertices (= loop iterations summed over paths), deepest vertex, finished samples
    if (lane == 0u) wave_totals<MIS, !OVF>(s_totals, kBlock / 64, a.stats, wave_shadow, wave_vertices, wave_samples, wave_depth);
}

// ------------------------------------------------------------------ the path kernel, regenerating form
// Level-0 launch of a large batch over a diffuse scene in LDS (the throughput case: C2, C3, C5).  k_paths keeps a path's
// state in the queue and moves it through HBM once per vertex; here a path stays in its lane's registers from
// its camera ray to its end, and a lane whose path has ended takes the next path of the batch on the spot
// (regeneration; Novak et al. 2010).  So every lane of every wave carries a path until the batch runs out -- no
// partially filled chunks, no queue traffic, no compaction -- and the only global accesses of the loop are the
// 12 bytes a finished sample writes and the chunk counters.
//   * Work: 64-path chunks of the batch (path id = s_local * np + pixel, as in k_paths).  The first regen_static chunks
//     are dealt round-robin (chunk k -> wave k % nw), the rest is handed out by kRegenCounters global counters (counter c
//     owns the chunks = c mod kRegenCounters; one returning atomic per chunk), so that the waves finish together: they
//     do not run equally fast -- a SIMD serves its oldest wave first -- (a static deal of 15/16 of the chunks: 7.63 ms,
//     of 1/2: 6.87, of 1/4: 6.36).  ONE counter for every chunk saturates: the chip consumes ~150 chunks per microsecond and a single address takes ~85 atomics per
//     microsecond (C2 12.3 instead of 8.0 ms).
//   * Camera rays are generated for a whole chunk at a time, all 64 lanes busy, into a per-wave ring in LDS (direction +
//     film position, 20 bytes; the origin is the camera's); a lane that needs a path pops the entry of its rank among
//     the needy lanes.  Generating rays only for the lanes that need one would run the Philox + normalise code at
//     ~20 % lane utilisation in every iteration.
//   * End of the batch: when the counters are used up and a wave's ring is empty, its lanes run dry one by one.  By
//     default (export_below = 1) the wave ends with its last path and no continuation launch follows; with a larger
//     threshold it appends what is alive below it to the overflow queue, as k_paths does (measured: not faster).
//   * The results do not depend on which lane traced which path: the RNG is addressed by (pixel, sample, depth), every
//     sample has its own slot of lsamp, and the statistics are sums.
// Occupancy the variants are compiled for (pt_kernels.h: the host sizes the grid by it): the DIFFUSE variant needs 79 VGPRs
// (6 waves per SIMD), the generic one 93 (5; reached only with PtTuning.level0_form = 2).
// Finished samples (stats[4]; pt_sync compares the sum with pixels x spp).  The queue-form kernels count the lanes that write their
// radiance to the sample buffer (a ballot at the store).  In the regenerating kernels a lane's ONLY transition from "has a path" to
// "has none" is that store, and paths enter a wave only from its ring, so finished = (entries taken from the ring) - (paths handed
// over at the end): one scalar add per iteration on a number the loop computes anyway.  (Counting at the store itself was measured:
// a per-lane count packed into the depth word cost 1.5 % on C2 -- profiles/r05/ab_count_finished.txt; a ballot per iteration
// costs the split form two more spilled registers.)
#ifndef PT_COUNT_FINISHED
#define PT_COUNT_FINISHED 1      // 0: measurement variant without the count (pt_sync's check is compiled out with it: A/B of its cost only)
#endif
constexpr uint32_t kPool = 128;            // ring entries per wave (>= 2 chunks: refilled whenever fewer than 64 are left)
// Workgroup size of k_paths_regen.  Its waves share nothing but the LDS copy of the scene, so a workgroup could be ONE wave --
// a wave that ends would free a slot the next launch (pt_api.cpp, lanes) can take at once, where a four-wave workgroup needs
// four slots of a CU at the same moment.  Measured (round 4, profiles/r04/ab_regen_block_64.txt): one rank's share of C2 at 8
// ranks 0.94 -> 1.00 ms, the whole image unchanged: rejected, 256 stays; the knob remains for measurements.
#ifndef PT_REGEN_BLOCK
#define PT_REGEN_BLOCK 256
#endif
constexpr uint32_t kRegenBlock = PT_REGEN_BLOCK;
#ifndef PT_RING_LAZY
#define PT_RING_LAZY 0
#endif
#ifndef PT_RESOLVE_UNROLL
#define PT_RESOLVE_UNROLL 4     // k_resolve: samples whose loads are in flight together (32 VGPRs: what is free beside six 80-VGPR waves;
#endif                          // 8 -> 56 VGPRs and C1 1.3 % slower; a raised wave priority: nothing.  profiles/r04/ab_resolve_variants.txt)
#ifndef PT_DRAIN_MAIL
#define PT_DRAIN_MAIL 0      // measured (round 4): N = 1 5.96 -> 6.17 ms (4 spilled dwords, the per-iteration checks), one rank's share at 8 ranks
#endif                       // 0.955 -> 0.975 ms: pooling a workgroup's last paths in one wave buys nothing -- kept as a measurement variant
#if PT_DRAIN_MAIL
constexpr uint32_t kMailT = 16;            // a wave with at most this many live paths at the end of the batch donates them
#endif
#ifndef PT_DRAIN_PRIO
#define PT_DRAIN_PRIO 0
#endif
// DIFFUSE = the material set the kernel is compiled for (kMatsDiffuse / kMatsNoMirror / kMatsAll); round 3 added the
// middle one: a scene with OrenNayar but no Mirror surface (material.rs:166-296) takes this kernel too by default.
template <bool MIS, int DIFFUSE>
__global__ void __launch_bounds__(kRegenBlock, DIFFUSE == kMatsDiffuse ? kRegenWavesDiffuse : kRegenWavesGeneric) k_paths_regen(BounceArgs a) {
    extern __shared__ float4 lds[];
    __shared__ float4 s_pool_d[kRegenBlock / 64][kPool];      // (d.x, d.y, d.z, bits(tile_row << 16 | x))
    __shared__ uint32_t s_pool_s[kRegenBlock / 64][kPool];    // s_local << 16 (depth 0)
    __shared__ WgTotals s_totals;
    if (threadIdx.x == 0u) wg_totals_init(s_totals);
#if PT_DRAIN_MAIL
    // End of the batch: a wave left with a handful of paths hands them to a sibling wave of its workgroup and ends (see the
    // main loop).  One region per donor wave, plane-major; cnt = entries published, head = entries taken (atomic), active =
    // waves of the workgroup that have neither ended nor donated.
    __shared__ float4 s_mail[kRegenBlock / 64][4][kMailT];
    __shared__ uint32_t s_mail_cnt[kRegenBlock / 64], s_mail_head[kRegenBlock / 64], s_active;
    if (threadIdx.x < kRegenBlock / 64) { s_mail_cnt[threadIdx.x] = 0u; s_mail_head[threadIdx.x] = 0u; }
    if (threadIdx.x == 0u) s_active = kRegenBlock / 64;
#endif
    // Spare workgroups (BounceArgs.posted): in a sequence of overlapping launches only the first core_blocks of a launch work -- two
    // launches then sit side by side and the third fills the slots the first frees while it runs dry -- but the LAST launches of
    // a sequence, and a launch on its own, would leave half of the device empty.  So every launch brings a full device's worth of
    // workgroups, and a spare one asks, when it gets its slot, whether successors are waiting for it: yes -> it ends at once.
    if (a.posted != nullptr && blockIdx.x >= a.core_blocks) {
        __shared__ uint32_t s_posted;
        if (threadIdx.x == 0u) s_posted = __hip_atomic_load(a.posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (s_posted - a.seq >= 2u) return;
    }
    const SceneRef sc = stage_scene<kModeLds>(a.sc, lds);          // (its barrier also publishes the words above)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = threadIdx.x >> 6;
    float4* const pool_d = s_pool_d[wib];
    uint32_t* const pool_s = s_pool_s[wib];
    const uint32_t n_first = a.n_first;
    const uint32_t n_chunks = (n_first + 63u) >> 6;
    const uint32_t W = a.film_w;
    const f3 cam_o = mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);

    const uint32_t wave = blockIdx.x * (kRegenBlock / 64) + wib, nw = gridDim.x * (kRegenBlock / 64);
    uint32_t st_next = wave;               // wave-uniform: next chunk of the static deal
    uint32_t ctr = blockIdx.x % kRegenCounters, ctr_dry = 0;   // wave-uniform: counter in use, counters found used up
    uint32_t pool_head = 0, pool_cnt = 0;  // wave-uniform: ring read position, entries
    bool exhausted = false;                // wave-uniform: the batch has no more chunks
    uint32_t wave_shadow = 0, wave_vertices = 0;
    uint32_t wave_taken = 0;               // wave-uniform: paths this wave's lanes took from the ring (finished = taken - handed over)
    uint32_t dmax = 0;                     // per lane: deepest vertex of the paths this lane finished
    PathState p = parked_state();
    bool alive = false;
#if PT_DRAIN_MAIL
    bool mail_spent = false;               // wave-uniform: this wave's mail region has been published once and withdrawn
#endif
#ifdef PT_DRAIN_TIMING      // measurement build: when does the batch run out under the waves, when does the last wave end
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_exhausted = 0ull;
#endif

    for (;;) {
#ifdef PT_DRAIN_TIMING
        if (exhausted && t_exhausted == 0ull) t_exhausted = wall_clock64();
#endif
        // ---- keep at least one chunk of camera rays in the ring
#if PT_RING_LAZY
        // ... or rather: only what the lanes without a path ask for now.  The ring then holds 0 .. 63 entries between
        // refills instead of 64 .. 127, and what it holds when the batch runs out is work the wave has to do alone
        const uint32_t ring_want = (uint32_t)__popcll(__ballot(!alive));
        while (!exhausted && pool_cnt < ring_want) {
#else
        while (!exhausted && pool_cnt < 64u) {
#endif
            uint32_t chunk;
            if (st_next < a.regen_static) {            // dealt round-robin, like pass 0 of k_paths
                chunk = st_next; st_next += nw;
            } else {
                // the shared rest: chunk regen_static + ticket * kRegenCounters + c from counter c; a wave starts at the
                // counter of its workgroup and moves on to the next one when that is used up
                for (;;) {
                    uint32_t got = 0;
                    if (lane == 0u) got = atomicAdd(a.chunk_counter + ctr * kRegenCounterStride, 1u);
                    chunk = a.regen_static + __builtin_amdgcn_readfirstlane(got) * kRegenCounters + ctr;
                    if (chunk < n_chunks) break;
                    ctr = ctr + 1u == kRegenCounters ? 0u : ctr + 1u;
                    if (++ctr_dry == kRegenCounters) { exhausted = true; break; }
                }
                if (exhausted) break;
            }
            const uint32_t pid = chunk * 64u + lane;
            const uint32_t valid = n_first - chunk * 64u < 64u ? n_first - chunk * 64u : 64u;
            if (lane < valid) {
                uint32_t s_local, pix, yl, px;
                divmod_magic(pid, a.np, a.np_magic, s_local, pix);
                divmod_magic(pix, W, a.film_w_magic, yl, px);
                f3 o, d;
                camera_ray(a.cam, a.s_base + s_local, px, image_row(a.tile, yl), o, d);
                const uint32_t e = (pool_head + pool_cnt + lane) & (kPool - 1u);
                pool_d[e] = make_float4(d.x, d.y, d.z, __uint_as_float((yl << 16) | px));
                pool_s[e] = s_local << 16;
            }
            pool_cnt += valid;
        }
        __builtin_amdgcn_wave_barrier();
#if PT_DRAIN_PRIO == 1
        // end of the batch: the waves with the most work left go first (they end the launch)
        if (exhausted) {
            if (pool_cnt >= 32u) __builtin_amdgcn_s_setprio(3);
            else if (pool_cnt != 0u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
#elif PT_DRAIN_PRIO == 2
        // a wave with fresh work goes before a wave that runs dry (of this launch or of the previous one, beside which this
        // launch starts: pt_api.cpp, lanes): an issue slot spent on 64 live lanes does more than one spent on a few
        if (exhausted && pool_cnt == 0u) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
#endif
        // ---- lanes without a path take the ring's next entries, in lane order
        {
            const unsigned long long need = __ballot(!alive);
            const uint32_t r = lane_rank(need);
            if (!alive && r < pool_cnt) {
                const uint32_t e = (pool_head + r) & (kPool - 1u);
                const float4 q = pool_d[e];
                const uint32_t sd = pool_s[e];
                p.o = cam_o; p.d = mk(q.x, q.y, q.z);
                const uint32_t xy = __float_as_uint(q.w);
                p.yl = xy >> 16; p.px = xy & 0xFFFFu;
                p.s_local = sd >> 16; p.depth = 0u;
                p.beta = mk(1.f, 1.f, 1.f); p.L = mk(0.f, 0.f, 0.f);
                p.pdf_prev = 0.0f; p.eta_in = 1.0f;
                alive = true;
            }
            const uint32_t n_need = (uint32_t)__popcll(need);
            const uint32_t n_take = n_need < pool_cnt ? n_need : pool_cnt;
            pool_head += n_take; pool_cnt -= n_take;
            if (PT_COUNT_FINISHED) wave_taken += n_take;
        }
        __builtin_amdgcn_wave_barrier();
#if PT_DRAIN_MAIL
        // ---- end of the batch, nothing left in the ring: the workgroup's waves pool their last paths.
        // A wave with at most kMailT live paths publishes them in its region and ends; a sibling with free lanes takes them.
        // No wave ever waits: a donor publishes BEFORE it leaves the count of active waves, and every wave looks at the mail
        // once more AFTER it has left that count (and comes back if there is some) -- so whichever of two leaves first, the
        // other one sees either the mail or that it is the last wave, which keeps (takes back) its paths.
        const bool mail_phase = a.export_below <= 1u && exhausted && pool_cnt == 0u;
        if (mail_phase) {
            const unsigned long long freem = __ballot(!alive);
            uint32_t n_free = (uint32_t)__popcll(freem);
            const uint32_t my_rank = lane_rank(freem);
            uint32_t taken = 0;
            for (uint32_t r = 0; r < kRegenBlock / 64; ++r) {
                if (r == wib || n_free == taken) continue;
                const uint32_t cnt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_mail_cnt[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                const uint32_t head = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_mail_head[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (cnt <= head) continue;
                const uint32_t want = cnt - head < n_free - taken ? cnt - head : n_free - taken;
                uint32_t old = 0;
                if (lane == 0u) old = __hip_atomic_fetch_add(&s_mail_head[r], want, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                const uint32_t got = old < cnt ? (cnt - old < want ? cnt - old : want) : 0u;
                if (!alive && my_rank >= taken && my_rank < taken + got) {
                    const uint32_t e = old + (my_rank - taken);
                    p = unpack_state(s_mail[r][0][e], s_mail[r][1][e], s_mail[r][2][e], s_mail[r][3][e]);
                    alive = true;
                }
                taken += got;
            }
        }
        uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
        // (a wave that once had to take its mail back -- mail_spent -- traces its paths to their end)
        if (mail_phase && n_alive <= kMailT && !(mail_spent && n_alive != 0u)) {
            if (n_alive != 0u) {                         // publish, then leave
                const unsigned long long am = __ballot(alive);
                if (alive) {
                    const uint32_t e = lane_rank(am);
                    s_mail[wib][0][e] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
                    s_mail[wib][1][e] = make_float4(p.d.y, p.d.z, __uint_as_float((p.yl << 16) | p.px), __uint_as_float((p.s_local << 16) | p.depth));
                    s_mail[wib][2][e] = make_float4(p.beta.x, p.beta.y, p.beta.z, p.pdf_prev);
                    s_mail[wib][3][e] = make_float4(p.L.x, p.L.y, p.L.z, p.eta_in);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0u) __hip_atomic_store(&s_mail_cnt[wib], n_alive, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            uint32_t prev = 0;
            if (lane == 0u) prev = __hip_atomic_fetch_sub(&s_active, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = __builtin_amdgcn_readfirstlane(prev);
            // mail of the others that nobody has taken yet?  (also after a donation: it must not be the last look anybody takes)
            bool pending = false;
            for (uint32_t r = 0; r < kRegenBlock / 64; ++r)
                if (r != wib) pending = pending || __hip_atomic_load(&s_mail_cnt[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >
                                                       __hip_atomic_load(&s_mail_head[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pending = __builtin_amdgcn_readfirstlane(pending ? 1u : 0u) != 0u;
            if (n_alive != 0u && prev > 1u && !pending) { alive = false; break; }      // donated: a sibling is (still) there to take them
            if (n_alive == 0u && !pending) break;                                        // nothing left anywhere this wave could see
            // stay: the last wave keeps its paths (takes its own mail back), or there is mail to take in the next iteration
            if (lane == 0u) __hip_atomic_fetch_add(&s_active, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (n_alive != 0u) {
                // un-publish what nobody took (a sibling may have taken some in the meantime: those lanes' paths are gone)
                uint32_t old = 0;
                if (lane == 0u) old = __hip_atomic_fetch_add(&s_mail_head[wib], n_alive, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);             // entries [0, old) were taken by siblings, [old, n_alive) come back
                const unsigned long long am = __ballot(alive);
                if (alive && lane_rank(am) < old) { alive = false; p.o = parked_origin(); p.d = parked_dir(); }
                mail_spent = true;                       // the region is used up for good (cnt <= head from now on)
            }
            n_alive = (uint32_t)__popcll(__ballot(alive));
            if (n_alive == 0u) continue;                 // only mail to fetch: next iteration
        }
#else
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
#endif
        // running dry (only once the batch is exhausted): hand the rest over
        if (n_alive < a.export_below) break;           // export_below >= 1: a wave without paths ends

        const bool active = alive;
        const uint32_t kx = p.px, py = image_row(a.tile, p.yl);
        const uint32_t sample = a.s_base + p.s_local;
        wave_vertices += n_alive;

        // ---- scan #1: closest hit of the path ray (rendering.rs:41)
        int id; float t;
        scan_closest<kModeLds, false, DIFFUSE == kMatsDiffuse ? 0 : PT_PAIR_PREFETCH_GENERIC>(sc, p.o, p.d, a.t_min, kInf, id, t);
        Vertex v;
        vertex_begin<MIS, DIFFUSE>(sc, p, active, id, t, sample, kx, py, v);

        // ---- scan #2: visibility (rendering.rs:62-65)
        bool visible = false;
        if (MIS) {
            const unsigned long long sm = __ballot(v.need_shadow);
            if (sm != 0ull) {
                f3 sdir = v.need_shadow ? v.light_dir : parked_dir();
                f3 sorg = v.need_shadow ? v.hit.point : parked_origin();
                int sid; float st;
                scan_closest<kModeLds, true, DIFFUSE == kMatsDiffuse ? 0 : PT_PAIR_PREFETCH_GENERIC>(sc, sorg, sdir, a.t_min, v.distance - a.t_min, sid, st);
                visible = v.need_shadow && sid < 0;
                wave_shadow += (uint32_t)__popcll(sm);
            }
        }
        alive = vertex_end<MIS, DIFFUSE, true>(sc, p, v, visible, sample, kx, py, a.min_depth, a.max_depth);
        if (active && !alive) {
            a.lsamp[p.s_local * a.np + p.yl * W + p.px] = Rgb{p.L.x, p.L.y, p.L.z};
            dmax = p.depth > dmax ? p.depth : dmax;
            p.o = parked_origin(); p.d = parked_dir();  // until the lane gets its next path (end of the batch: for good)
        }
    }

    // ---- hand-over of the paths still alive (none unless the batch ran out under them)
    {
        const unsigned long long mask = __ballot(alive);
        const uint32_t n_left = (uint32_t)__popcll(mask);
        if (n_left != 0u) {
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(a.ovf_out_count, n_left);
            base = __builtin_amdgcn_readfirstlane(base);
            if (alive) {
                store_state(a.ovf_out, base + lane_rank(mask), p);
                const uint32_t done = p.depth ? p.depth - 1u : 0u;   // deepest vertex it has been through (0: none yet)
                dmax = done > dmax ? done : dmax;
            }
        }
        wave_taken -= n_left;              // (mail pooling, a measurement variant, moves paths between waves: the launch's sum stays right)
    }
    const uint32_t wave_samples = wave_taken;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t w2 = (uint32_t)__shfl_xor((int)dmax, off); dmax = w2 > dmax ? w2 : dmax; }
    if (lane == 0u) {
        wave_totals<MIS>(s_totals, kRegenBlock / 64, a.stats, wave_shadow, wave_vertices, wave_samples, dmax);
#ifdef PT_DRAIN_TIMING      // stats[8..12] (beyond the 8 words the host reads): ~begin (min), ~exhausted (min), exhausted (max), end (max), sum of per-wave drain times
        const unsigned long long t_end = wall_clock64();
        if (t_exhausted == 0ull) t_exhausted = t_end;
        atomicMax(&a.stats[8], ~t_begin); atomicMax(&a.stats[9], ~t_exhausted); atomicMax(&a.stats[10], t_exhausted);
        atomicMax(&a.stats[11], t_end); atomicAdd(&a.stats[12], t_end - t_exhausted);
        // per wave: (begin, out of work, end) stamps and the vertices it processed, into the (unused) hand-over queue
        if (a.ovf_out.q[0]) a.ovf_out.q[a.debug_tag & 3u][wave] = make_float4(__uint_as_float((uint32_t)t_begin), __uint_as_float((uint32_t)t_exhausted),
                                                               __uint_as_float((uint32_t)t_end), __uint_as_float(wave_vertices));
#endif
    }
}

// ------------------------------------------------------------------ the regenerating form for scenes with a few Mirror (GGX) objects
// The reference's own scene (World::new(), world.rs:80-211) is 12 Lambertian / emissive triangles and ONE rough-glass
// sphere.  In k_paths_regen paths of every depth share a wave, so nearly every iteration has a lane or two on the glass
// and the whole GGX code (bsdf_pdf + VNDF sample + the lobe's Philox block, ~350-500 instructions) runs at a few percent
// lane utilisation: C1 costs 37.5 us per million vertices against 26.3 for the same geometry with a Lambertian sphere
// (tools/r03/c1_ggx_cost.py).  This form separates the two populations IN TIME inside each wave:
//   * plain iterations are k_paths_regen's, compiled without the Mirror code (kMatsNoMirror).  A lane whose path ray
//     turns out to hit a Mirror object (known after the closest-hit scan) does not shade it: it pushes the path -- the
//     state BEFORE the vertex plus the scan's (id, t) -- onto the wave's SPECIAL stack and takes a new path like a lane
//     whose path has ended.
//   * when 64 Mirror vertices have piled up (or nothing else is left to do) the wave parks its 64 plain paths in LDS (round 4;
//     round 3: in global memory), pops 64 special entries -- whole, into the registers the parked paths left -- and runs ONE
//     vertex for them with every lane on the GGX code (vertex_begin / scan / vertex_end of kMatsAll), then the closest-hit scan
//     of their NEXT vertex, the survivors staying in registers: Mirror again (a path inside the sphere) -> back onto the special
//     stack with its (id, t); anything else -> onto the wave's PLAIN stack, from which the regeneration step of the plain
//     iterations takes entries before it takes camera rays.  Then the plain paths come back into the lanes.
// Both stacks share one 128-entry region per wave in global memory (L2-resident: 10 KB per wave), special growing up, plain
// growing down.  They cannot collide.  Let S, P be the entries of the two stacks.  At the top of an iteration S <= 63 (batches
// run whenever >= 64 specials wait).  A plain iteration finds f <= 64 Mirror vertices; each of the first min(f, P) takes an
// entry OFF the plain stack as it puts one ON the special stack (PT_SPLIT_REPLACE, round 5: S + P unchanged), the others push
// with the plain stack empty, so afterwards S + P <= max(S + P before, 63 + 64).  A batch takes 64 entries off the special
// stack and puts at most 64 back on either: S + P does not grow.  Hence S + P <= 127 < 128 always.  (Round 3-4 form, without
// the replacement: lanes take plain entries before camera rays, so a push finds the plain stack empty.)  No atomics, no other
// wave involved; a violated invariant sets stats[7] and pt_sync fails.
// Same per-vertex functions on the same inputs as every other form (a path's arithmetic does not depend on which lane
// or in which order it is traced), so the film is bit-identical (test_level0_forms_give_the_same_film, the fuzz tests).
constexpr int kWaitVm0 = 0x0F70;                                 // s_waitcnt vmcnt(0) alone (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
constexpr uint32_t kXq = 128;                                    // exchange entries per wave
constexpr uint32_t kXqEntryF4 = 5;                               // stack entry: 4 float4 of path state (layout of Queue) + (bits(id), t, -, -)
constexpr uint32_t kXqF4PerWave = kXq * kXqEntryF4;               // the stacks (round 3 also parked the wave's 64 plain paths here: LDS since round 4, ab_c1_park_in_lds.txt)
#ifndef PT_SPLIT_REPLACE
#define PT_SPLIT_REPLACE 1       // measured (round 5): 7.40 -> 7.30 ms per C1 launch -- profiles/r05/ab_c1_replace_flat.txt, ab_noslp.txt
#endif
#ifndef PT_SPLIT_BATCH_MATS
#define PT_SPLIT_BATCH_MATS kMatsAll      // kMatsMirror: the batch's vertex code compiled for Mirror hits only (measurement)
#endif
#ifndef PT_SPLIT_STAY
#define PT_SPLIT_STAY 0          // measured (round 5): 7.52 -> 7.62 ms per C1 launch at every threshold tried -- profiles/r05/ab_c1_stay_in_lane.txt
#endif
#ifndef PT_SPLIT_STAY_MIN
#define PT_SPLIT_STAY_MIN 32
#endif
#if PT_SPLIT_STAY
constexpr uint32_t kSplitStayMin = PT_SPLIT_STAY_MIN;             // a batch goes on while at least this many specials are in lanes + waiting
#endif
static_assert(kXqF4PerWave == kRegenSplitF4PerWave, "pt_kernels.h sizes the buffer");
// One wave-uniform base pointer (two scalar registers); entry-major, so the planes of an entry are immediate offsets of
// ONE address -- with plane-major arrays the compiler kept a scalar base per plane (24 SGPRs more than the kernel has).
struct XWave {
    float4* b;
    PT_DEV float4* entry(uint32_t e) const { return b + e * kXqEntryF4; }                  // [0..3] state, [4] = (bits(id), t, -, -) of a special entry's pending vertex
};
PT_DEV XWave xwave(float4* base, uint32_t wave_uniform) {
    XWave x;
    x.b = base + (size_t)wave_uniform * kXqF4PerWave;
    return x;
}
PT_DEV void store_entry(float4* e, const PathState& p) {
    e[0] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
    e[1] = make_float4(p.d.y, p.d.z, __uint_as_float((p.yl << 16) | p.px), __uint_as_float((p.s_local << 16) | p.depth));
    e[2] = make_float4(p.beta.x, p.beta.y, p.beta.z, p.pdf_prev);
    e[3] = make_float4(p.L.x, p.L.y, p.L.z, p.eta_in);
}
PT_DEV bool is_mirror_obj(const SceneRef& sc, int id) { return (__float_as_uint(sc.mat[2 * id].x) & 0xFFu) == MAT_MIRROR; }

// PLAIN = the material set of the plain iterations: kMatsDiffuse when the scene has no OrenNayar surface either (the
// reference scene: exactly k_paths_regen<MIS, DIFFUSE>'s code there), else kMatsNoMirror.
template <bool MIS, int PLAIN>
__global__ void __launch_bounds__(kBlock, kRegenWavesSplit) k_paths_regen_split(BounceArgs a) {
    extern __shared__ float4 lds[];
    __shared__ float4 s_pool_d[kBlock / 64][kPool];
    __shared__ uint32_t s_pool_s[kBlock / 64][kPool];
    __shared__ float4 s_stage[kBlock / 64][4][64];       // the parking area of each wave's plain paths during a batch of specials (plane-major: conflict-free)
    __shared__ WgTotals s_totals;
    if (threadIdx.x == 0u) wg_totals_init(s_totals);
    if (a.posted != nullptr && blockIdx.x >= a.core_blocks) {       // spare workgroup (see k_paths_regen)
        __shared__ uint32_t s_posted;
        if (threadIdx.x == 0u) s_posted = __hip_atomic_load(a.posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (s_posted - a.seq >= 2u) return;
    }
    const SceneRef sc = stage_scene<kModeLds>(a.sc, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = threadIdx.x >> 6;
    float4* const pool_d = s_pool_d[wib];
    uint32_t* const pool_s = s_pool_s[wib];
    float4 (*const park)[64] = s_stage[wib];             // the wave's plain paths while a batch of specials runs
    const uint32_t n_first = a.n_first;
    const uint32_t n_chunks = (n_first + 63u) >> 6;
    const uint32_t W = a.film_w;
    const f3 cam_o = mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);

    const uint32_t wave = blockIdx.x * (kBlock / 64) + wib, nw = gridDim.x * (kBlock / 64);
    const XWave x = xwave(a.xchg, __builtin_amdgcn_readfirstlane(wave));
    uint32_t st_next = wave;
    uint32_t ctr = blockIdx.x % kRegenCounters, ctr_dry = 0;
    uint32_t pool_head = 0, pool_cnt = 0;
    bool exhausted = false;
    uint32_t sq_cnt = 0, pq_cnt = 0;       // wave-uniform: special entries [0, sq_cnt), plain entries [kXq - pq_cnt, kXq)
    bool overflow = false;                 // wave-uniform: the stacks met (cannot happen, see above; reported instead of corrupting paths)
    uint32_t wave_shadow = 0, wave_vertices = 0;
    uint32_t wave_taken = 0;               // wave-uniform: paths taken from the ring; every one of them ends in this wave (no hand-over)
    uint32_t dmax = 0;
    PathState p = parked_state();
    bool alive = false;

    for (;;) {
        // ---- keep at least one chunk of camera rays in the ring (as k_paths_regen)
        while (!exhausted && pool_cnt < 64u) {
            uint32_t chunk;
            if (st_next < a.regen_static) {
                chunk = st_next; st_next += nw;
            } else {
                for (;;) {
                    uint32_t got = 0;
                    if (lane == 0u) got = atomicAdd(a.chunk_counter + ctr * kRegenCounterStride, 1u);
                    chunk = a.regen_static + __builtin_amdgcn_readfirstlane(got) * kRegenCounters + ctr;
                    if (chunk < n_chunks) break;
                    ctr = ctr + 1u == kRegenCounters ? 0u : ctr + 1u;
                    if (++ctr_dry == kRegenCounters) { exhausted = true; break; }
                }
                if (exhausted) break;
            }
            const uint32_t pid = chunk * 64u + lane;
            const uint32_t valid = n_first - chunk * 64u < 64u ? n_first - chunk * 64u : 64u;
            if (lane < valid) {
                uint32_t s_local, pix, yl, px;
                divmod_magic(pid, a.np, a.np_magic, s_local, pix);
                divmod_magic(pix, W, a.film_w_magic, yl, px);
                f3 o, d;
                camera_ray(a.cam, a.s_base + s_local, px, image_row(a.tile, yl), o, d);
                const uint32_t e = (pool_head + pool_cnt + lane) & (kPool - 1u);
                pool_d[e] = make_float4(d.x, d.y, d.z, __uint_as_float((yl << 16) | px));
                pool_s[e] = s_local << 16;
            }
            pool_cnt += valid;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lanes without a path: first the plain stack (paths that left a Mirror surface), then the ring
        // (PT_SPLIT_REPLACE: the ring first -- what waits on the plain stack has its next vertex scanned already and takes the place
        // of the lanes that find a Mirror vertex below; the lanes here take it only when the ring cannot serve them: the end of the batch)
        {
            const unsigned long long need = __ballot(!alive);
            const uint32_t r = lane_rank(need);
            const uint32_t n_need = (uint32_t)__popcll(need);
#if PT_SPLIT_REPLACE
            const uint32_t n_ring = n_need < pool_cnt ? n_need : pool_cnt;
            const uint32_t n_pq = n_need - n_ring < pq_cnt ? n_need - n_ring : pq_cnt;
            const bool from_pq = !alive && r >= n_ring && r - n_ring < n_pq;
            const bool from_ring = !alive && r < n_ring;
            const uint32_t e_pq = kXq - pq_cnt + (r - n_ring), e_ring = (pool_head + r) & (kPool - 1u);
#else
            const uint32_t n_pq = n_need < pq_cnt ? n_need : pq_cnt;
            const uint32_t n_ring = n_need - n_pq < pool_cnt ? n_need - n_pq : pool_cnt;
            const bool from_pq = !alive && r < n_pq;
            const bool from_ring = !alive && r >= n_pq && r - n_pq < pool_cnt;
            const uint32_t e_pq = kXq - pq_cnt + r, e_ring = (pool_head + r - n_pq) & (kPool - 1u);
#endif
            if (from_pq) {
                const float4* src = x.entry(e_pq);
                p = unpack_state(src[0], src[1], src[2], src[3]);
                // all four loads back HERE: otherwise the compiler waits (vmcnt(0)) at the first use of beta / L in the
                // iteration below, on every path -- and on gfx9 that counter also holds the previous iteration's
                // sample stores, a full HBM write latency per iteration (measured on C2, which never takes this
                // branch: 7.5 instead of 6.2 ms)
                __builtin_amdgcn_s_waitcnt(kWaitVm0);
                alive = true;
            } else if (from_ring) {
                const float4 q = pool_d[e_ring];
                const uint32_t sd = pool_s[e_ring];
                p.o = cam_o; p.d = mk(q.x, q.y, q.z);
                const uint32_t xy = __float_as_uint(q.w);
                p.yl = xy >> 16; p.px = xy & 0xFFFFu;
                p.s_local = sd >> 16; p.depth = 0u;
                p.beta = mk(1.f, 1.f, 1.f); p.L = mk(0.f, 0.f, 0.f);
                p.pdf_prev = 0.0f; p.eta_in = 1.0f;
                alive = true;
            }
            pq_cnt -= n_pq;
            pool_head += n_ring; pool_cnt -= n_ring;
            if (PT_COUNT_FINISHED) wave_taken += n_ring;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
        // no plain path although stack and ring were offered: the batch is used up.  Without waiting specials the wave is done.
        if (n_alive == 0u && sq_cnt == 0u) break;

        if (n_alive != 0u) {
            // ---- plain iteration: scan #1 (rendering.rs:41)
            int id; float t;
            scan_closest<kModeLds, false, PT_PAIR_PREFETCH>(sc, p.o, p.d, a.t_min, kInf, id, t);
            // a Mirror vertex is not shaded here: the path waits on the special stack for a batch of its kind
            const bool special = alive && id >= 0 && is_mirror_obj(sc, id);
            const unsigned long long spm = __ballot(special);
            if (spm != 0ull) {
#if PT_SPLIT_REPLACE
                // Round 5: a lane that hands its path to the special stack takes, in the same breath, a path from the plain stack --
                // one that left the glass in an earlier batch, whose next vertex that batch has scanned already: (id, t) travel with
                // the entry.  The lane goes on with vertex_begin at once instead of idling through the rest of the iteration, and the
                // path is not scanned a second time.  The pops are complete before the pushes are issued (with both stacks nearly
                // full the pushed entries may be the popped ones); lanes + stacks stay <= 127 paths: a push without a pop happens
                // only with the plain stack empty, i.e. at <= 63 + 64 entries.
                const uint32_t rk = lane_rank(spm);
                const uint32_t n_sp = (uint32_t)__popcll(spm);
                const uint32_t n_rep = n_sp < pq_cnt ? n_sp : pq_cnt;
                const bool rep = special && rk < n_rep;
                PathState q = p;
                int qid = -1; float qt = 0.0f;
                if (rep) {
                    const float4* src = x.entry(kXq - pq_cnt + rk);
                    q = unpack_state(src[0], src[1], src[2], src[3]);
                    const float2 it = *reinterpret_cast<const float2*>(src + 4);
                    qid = __float_as_int(it.x); qt = it.y;
                    __builtin_amdgcn_s_waitcnt(kWaitVm0);
                }
                if (special) {
                    float4* dst = x.entry(sq_cnt + rk);
                    store_entry(dst, p);
                    *reinterpret_cast<float2*>(dst + 4) = make_float2(__int_as_float(id), t);
                    if (rep) {
                        p = q; id = qid; t = qt;
                    } else {
                        alive = false;
                        p.o = parked_origin(); p.d = parked_dir();
                        id = -1;
                    }
                }
                sq_cnt += n_sp;
                pq_cnt -= n_rep;
#else
                if (special) {
                    float4* dst = x.entry(sq_cnt + lane_rank(spm));
                    store_entry(dst, p);
                    *reinterpret_cast<float2*>(dst + 4) = make_float2(__int_as_float(id), t);     // (8 of the slot's 16 bytes: no padding words to keep in registers)
                    alive = false;
                    p.o = parked_origin(); p.d = parked_dir();
                    id = -1;
                }
                sq_cnt += (uint32_t)__popcll(spm);
#endif
                overflow = overflow || sq_cnt + pq_cnt > kXq;
            }
            const bool active = alive;
            const uint32_t kx = p.px, py = image_row(a.tile, p.yl);
            const uint32_t sample = a.s_base + p.s_local;
            wave_vertices += (uint32_t)__popcll(__ballot(active));
            Vertex v;
            vertex_begin<MIS, PLAIN>(sc, p, active, id, t, sample, kx, py, v);
            bool visible = false;
            if (MIS) {
                const unsigned long long sm = __ballot(v.need_shadow);
                if (sm != 0ull) {
                    f3 sdir = v.need_shadow ? v.light_dir : parked_dir();
                    f3 sorg = v.need_shadow ? v.hit.point : parked_origin();
                    int sid; float st;
                    scan_closest<kModeLds, true, PT_PAIR_PREFETCH>(sc, sorg, sdir, a.t_min, v.distance - a.t_min, sid, st);
                    visible = v.need_shadow && sid < 0;
                    wave_shadow += (uint32_t)__popcll(sm);
                }
            }
            alive = vertex_end<MIS, PLAIN, true>(sc, p, v, visible, sample, kx, py, a.min_depth, a.max_depth);
            if (active && !alive) {
                a.lsamp[p.s_local * a.np + p.yl * W + p.px] = Rgb{p.L.x, p.L.y, p.L.z};
                dmax = p.depth > dmax ? p.depth : dmax;
                p.o = parked_origin(); p.d = parked_dir();
            }
        }

        // ---- batches of Mirror vertices: whenever a full wave of them waits, or nothing else is left to do
        auto plain_work = [&]() { return __ballot(alive) != 0ull || pool_cnt != 0u || !exhausted || pq_cnt != 0u; };
        if (sq_cnt >= 64u || (sq_cnt != 0u && !plain_work())) {
            // park the plain paths in LDS for the batch (round 4): their 16 registers carry the batch's paths instead -- the
            // popped entry WHOLE (one wait per batch iteration instead of one for the ray part and one for the carry part), and
            // the survivors across the scan of their next vertex
            park[0][lane] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
            park[1][lane] = make_float4(p.d.y, p.d.z, __uint_as_float((p.yl << 16) | p.px), __uint_as_float((p.s_local << 16) | p.depth));
            park[2][lane] = make_float4(p.beta.x, p.beta.y, p.beta.z, p.pdf_prev);
            park[3][lane] = make_float4(p.L.x, p.L.y, p.L.z, p.eta_in);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     // entries pushed above are read by other lanes below
#if PT_SPLIT_STAY
            // Round 5: a path whose NEXT vertex is Mirror again (it is inside the glass sphere: most of what a batch refracts)
            // STAYS in its lane with the scan's (id, t) and is shaded by the next iteration of this loop, where the lanes the
            // others freed take what still waits on the stack.  Round 4 pushed it (80 B written) and popped it again in a later
            // batch (80 B read): a third of the form's stack traffic, and a dozen plain iterations of latency for the path.
            PathState q = parked_state();
            bool qa = false;                                           // this lane carries a special waiting for its vertex
            int qid = -1; float qt = 0.0f;
            uint32_t n_stay = 0;                                       // wave-uniform: lanes with qa
            do {
                {   // lanes without a special take the top entries of the special stack, in lane order
                    const unsigned long long need = __ballot(!qa);
                    const uint32_t r = lane_rank(need);
                    const uint32_t n_need = (uint32_t)__popcll(need);
                    const uint32_t n_pop = n_need < sq_cnt ? n_need : sq_cnt;
                    if (!qa && r < n_pop) {
                        const float4* src = x.entry(sq_cnt - 1u - r);
                        q = unpack_state(src[0], src[1], src[2], src[3]);
                        const float2 it = *reinterpret_cast<const float2*>(src + 4);
                        qid = __float_as_int(it.x); qt = it.y;
                        qa = true;
                    }
                    sq_cnt -= n_pop;
                    n_stay += n_pop;
                }
                if (!qa) { q.o = parked_origin(); q.d = parked_dir(); qid = -1; }
                const uint32_t kx = q.px, py = image_row(a.tile, q.yl);
                const uint32_t sample = a.s_base + q.s_local;
                wave_vertices += n_stay;
                Vertex v;
                vertex_begin<MIS, PT_SPLIT_BATCH_MATS>(sc, q, qa, qid, qt, sample, kx, py, v);
                bool visible = false;
                if (MIS) {
                    const unsigned long long sm = __ballot(v.need_shadow);
                    if (sm != 0ull) {
                        f3 sdir = v.need_shadow ? v.light_dir : parked_dir();
                        f3 sorg = v.need_shadow ? v.hit.point : parked_origin();
                        int sid; float st;
                        scan_closest<kModeLds, true, PT_PAIR_PREFETCH>(sc, sorg, sdir, a.t_min, v.distance - a.t_min, sid, st);
                        visible = v.need_shadow && sid < 0;
                        wave_shadow += (uint32_t)__popcll(sm);
                    }
                }
                const bool qalive = vertex_end<MIS, PT_SPLIT_BATCH_MATS, true>(sc, q, v, visible, sample, kx, py, a.min_depth, a.max_depth);
                if (qa && !qalive) {
                    a.lsamp[q.s_local * a.np + q.yl * W + q.px] = Rgb{q.L.x, q.L.y, q.L.z};
                    dmax = q.depth > dmax ? q.depth : dmax;
                }
                // the survivors' next vertex: Mirror again or not?
                const f3 so = qalive ? q.o : parked_origin(), sd = qalive ? q.d : parked_dir();
                asm volatile("" ::: "memory");
                int id2; float t2;
                scan_closest<kModeLds, false, PT_PAIR_PREFETCH>(sc, so, sd, a.t_min, kInf, id2, t2);
                const bool spec2 = qalive && id2 >= 0 && is_mirror_obj(sc, id2);
                const bool plain2 = qalive && !spec2;
                const unsigned long long m_p = __ballot(plain2);
                const uint32_t n_p = (uint32_t)__popcll(m_p);
                if (plain2) store_entry(x.entry(kXq - pq_cnt - n_p + lane_rank(m_p)), q);
                pq_cnt += n_p;
                qa = spec2; qid = id2; qt = t2;
                n_stay = (uint32_t)__popcll(__ballot(spec2));
                overflow = overflow || sq_cnt + pq_cnt + n_stay > kXq;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            } while (n_stay + sq_cnt >= kSplitStayMin || (n_stay + sq_cnt != 0u && !plain_work()));
            // what stays below the threshold waits on the special stack for the next batch, like a special a plain iteration found
            if (n_stay != 0u) {
                const unsigned long long m_s = __ballot(qa);
                if (qa) {
                    float4* de = x.entry(sq_cnt + lane_rank(m_s));
                    store_entry(de, q);
                    *reinterpret_cast<float2*>(de + 4) = make_float2(__int_as_float(qid), qt);
                }
                sq_cnt += n_stay;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            }
#else
            do {
                const uint32_t n = sq_cnt < 64u ? sq_cnt : 64u;
                const bool qa = lane < n;
                const uint32_t e = sq_cnt - n + (qa ? lane : 0u);     // the top n entries
                PathState q = parked_state();
                int qid = -1; float qt = 0.0f;
                if (qa) {
                    const float4* src = x.entry(e);
                    q = unpack_state(src[0], src[1], src[2], src[3]);
                    const float2 it = *reinterpret_cast<const float2*>(src + 4);
                    qid = __float_as_int(it.x); qt = it.y;
                }
                sq_cnt -= n;
                const uint32_t kx = q.px, py = image_row(a.tile, q.yl);
                const uint32_t sample = a.s_base + q.s_local;
                wave_vertices += n;
                Vertex v;
                vertex_begin<MIS, PT_SPLIT_BATCH_MATS>(sc, q, qa, qid, qt, sample, kx, py, v);
                bool visible = false;
                if (MIS) {
                    const unsigned long long sm = __ballot(v.need_shadow);
                    if (sm != 0ull) {
                        f3 sdir = v.need_shadow ? v.light_dir : parked_dir();
                        f3 sorg = v.need_shadow ? v.hit.point : parked_origin();
                        int sid; float st;
                        scan_closest<kModeLds, true, PT_PAIR_PREFETCH>(sc, sorg, sdir, a.t_min, v.distance - a.t_min, sid, st);
                        visible = v.need_shadow && sid < 0;
                        wave_shadow += (uint32_t)__popcll(sm);
                    }
                }
                const bool qalive = vertex_end<MIS, PT_SPLIT_BATCH_MATS, true>(sc, q, v, visible, sample, kx, py, a.min_depth, a.max_depth);
                if (qa && !qalive) {
                    a.lsamp[q.s_local * a.np + q.yl * W + q.px] = Rgb{q.L.x, q.L.y, q.L.z};
                    dmax = q.depth > dmax ? q.depth : dmax;
                }
                // the survivors' next vertex: Mirror again (a path inside the sphere) or not?
                const f3 so = qalive ? q.o : parked_origin(), sd = qalive ? q.d : parked_dir();
                asm volatile("" ::: "memory");
                int id2; float t2;
                scan_closest<kModeLds, false, PT_PAIR_PREFETCH>(sc, so, sd, a.t_min, kInf, id2, t2);
                const bool spec2 = qalive && id2 >= 0 && is_mirror_obj(sc, id2);
                const bool plain2 = qalive && !spec2;
                const unsigned long long m_s = __ballot(spec2), m_p = __ballot(plain2);
                const uint32_t n_p = (uint32_t)__popcll(m_p);
                if (qalive) {
                    const uint32_t dst = spec2 ? sq_cnt + lane_rank(m_s) : kXq - pq_cnt - n_p + lane_rank(m_p);
                    float4* de = x.entry(dst);
                    store_entry(de, q);
                    if (spec2 || PT_SPLIT_REPLACE) *reinterpret_cast<float2*>(de + 4) = make_float2(__int_as_float(id2), t2);
                }
                sq_cnt += (uint32_t)__popcll(m_s);
                pq_cnt += n_p;
                overflow = overflow || sq_cnt + pq_cnt > kXq;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            } while (sq_cnt >= 64u || (sq_cnt != 0u && !plain_work()));
#endif
            p = unpack_state(park[0][lane], park[1][lane], park[2][lane], park[3][lane]);
        }
    }

    const uint32_t wave_samples = wave_taken;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t w2 = (uint32_t)__shfl_xor((int)dmax, off); dmax = w2 > dmax ? w2 : dmax; }
    if (lane == 0u) {
        wave_totals<MIS>(s_totals, kBlock / 64, a.stats, wave_shadow, wave_vertices, wave_samples, dmax);
        if (overflow) atomicMax(&a.stats[7], 1ull);      // pt_sync turns it into an error
    }
}

// ------------------------------------------------------------------ the path kernel, BVH form
// Same organisation as k_paths (one launch per batch, wave-private queue segments compacted in place, tail
// hand-off), but a BVH traversal diverges: rays of one wave need between a handful and a few hundred steps,
// and a wave that traces 64 rays side by side idles most lanes most of the time (measured on C4: 17 % of the
// VALU lane-cycles did work).  So every pass over a segment is cut into stages, and the two traversal stages
// hand a NEW ray to a lane as soon as its ray is done (traverse_segment):
//     stage 1  extend   closest hit of every path ray of the segment          -> aux[slot].xy = (id, t)
//     stage 2  connect  hit record + light sample of every path (vertex_begin) -> shadow ray of the slot
//     stage 3  occlude  any-hit traversal of the shadow rays                   -> aux[slot].z
//     stage 4  shade    vertex_begin again (cheaper than storing it) + vertex_end, in-place compaction
// Pass 0 first writes the camera rays (or the overflow queue's paths) into the segment.

// Rays of a segment: plane0[slot] = (o, d.x), plane1[slot] = (d.y, d.z, t_max, has_ray) when TMAX_IN_RAY (a slot
// with has_ray == 0 is skipped; t_max itself may be anything, also negative or NaN -- the scan's semantics
// decide), else plane1[slot] = (d.y, d.z, -, -) and t_max = inf.  ANY: out[slot].z = 1 if anything is hit, else 0 (visibility).  Otherwise
// out[slot].xy = (id, t) of the closest hit.  Semantics of one ray: bvh_scan.
#ifndef PT_BVH_POSTPONE
#define PT_BVH_POSTPONE 1
#endif
// order: null, or the order in which the slots are handed out -- order[k].w holds (as bits) the slot of the k-th ray
// (sort_segment below); results still land in out[slot].
template <bool TMAX_IN_RAY, bool ANY>
PT_DEV void traverse_segment(const SceneRef& sc, const float4* __restrict__ plane0, const float4* __restrict__ plane1,
                             float4* __restrict__ out, uint32_t n, float t_min, uint32_t refill_below, uint32_t leaf_batch,
                             const float4* order = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* stk = sc.stack + threadIdx.x;
    uint32_t next = 0;                         // wave-uniform: first slot not handed out yet
    bool has = false;                          // this lane is tracing a ray
    uint32_t slot = 0, node = 0xFFFFFFFFu, sp = 1;
    // slab planes as one FMA each: t = plane * inv + b with bp = -(o + pad) * inv for the lower plane of a box and
    // bm = -(o - pad) * inv for the upper one (the rounding of b moves a plane by <= ulp(|o|), far inside pad)
    // ... with the plane on the builder's 16-bit grid, plane = grid_min + q * cell: t = q * (cell * inv) + b, b now from
    // (grid_min - (o +- pad)) * inv.  The folded form rounds differently from "decode, then slab", by ~1e-7 of the
    // scene extent: far inside pad as well.
    f3 o = parked_origin(), d = parked_dir(), inv = mk(0.f, 0.f, 0.f), bp = inv, bm = inv;
    const f3 gmin = mk(sc.bvh.grid_min[0], sc.bvh.grid_min[1], sc.bvh.grid_min[2]);
    const f3 gcell = mk(sc.bvh.grid_cell[0], sc.bvh.grid_cell[1], sc.bvh.grid_cell[2]);
    float closest = 0.0f;
    int id = -1;
#if PT_BVH_POSTPONE
    uint32_t pend = 0xFFFFFFFFu;                // a leaf this lane has put aside (none: the sentinel)
#endif
    for (;;) {
        // ---- hand the next slots to the idle lanes, in lane order
        const unsigned long long idle = __ballot(!has);
        if (next < n && idle != 0ull) {
            uint32_t cand = next + (uint32_t)__popcll(idle & lt);
            if (!has && cand < n) {
                if (order) cand = __float_as_uint(order[cand].w);
                const float4 r0 = plane0[cand], r1 = plane1[cand];
                const float t_max = TMAX_IN_RAY ? r1.z : kInf;
                if (!TMAX_IN_RAY || r1.w != 0.0f) {
                    slot = cand;
                    o = mk(r0.x, r0.y, r0.z); d = mk(r0.w, r1.x, r1.y);
                    const float a = dot(d, d), inv_a = __builtin_amdgcn_rcpf(a);      // only to recognise zero / NaN rays
                    closest = t_max;
                    id = -1;
                    const float o1 = __builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z);
                    const bool regular = o1 + __builtin_fabsf(d.x) + __builtin_fabsf(d.y) + __builtin_fabsf(d.z) + a + inv_a < kInf;
                    if (regular) {
                        const float pad = kBvhPad * (o1 + sc.bvh.scene_abs);
                        // 1/d clamped to +-1e25: a zero (or denormal) component then acts like +-infinity without the
                        // inf - inf = NaN an FMA would make of it (one NaN plane and min/max collapse the slab interval)
                        inv = mk(__builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_rcpf(d.x), -1e25f), 1e25f),
                                 __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_rcpf(d.y), -1e25f), 1e25f),
                                 __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_rcpf(d.z), -1e25f), 1e25f));
                        bp = mk((gmin.x - (o.x + pad)) * inv.x, (gmin.y - (o.y + pad)) * inv.y, (gmin.z - (o.z + pad)) * inv.z);
                        bm = mk((gmin.x - (o.x - pad)) * inv.x, (gmin.y - (o.y - pad)) * inv.y, (gmin.z - (o.z - pad)) * inv.z);
                        inv = inv * gcell;                                    // from here on: per grid step
                        stk[0] = 0xFFFFFFFFu;
                        sp = 1;
                        node = sc.bvh.root;
                    } else {
                        scan_global(sc, o, d, t_min, t_max, id, closest);      // the linear scan's NaN behaviour (rare)
                        node = 0xFFFFFFFFu;
                    }
                    has = true;
                }
            }
            const uint32_t given = (uint32_t)__popcll(idle);
            next = n - next < given ? n : next + given;
        }
        if (__ballot(has) == 0ull) {
            if (next >= n) break;
            continue;
        }
        const uint32_t low_water = next < n ? refill_below : 1u;
        do {
            if (has && (int)node >= 0) {                 // internal node: one 64-byte visit tests its (up to) four child boxes
                const uint4* nd = sc.bvh.nodes + 4u * node;
                const uint4 qa = nd[0], qb = nd[1], qc = nd[2], qd = nd[3];
                // child c = three words (lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16); entry distance of the
                // padded box inside [t_min, closest], or "no hit"
                auto slab = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t code, float& tn) -> bool {
                    const float ax0 = __builtin_fmaf((float)(w0 & 0xFFFFu), inv.x, bp.x), ax1 = __builtin_fmaf((float)(w1 >> 16), inv.x, bm.x);
                    const float ay0 = __builtin_fmaf((float)(w0 >> 16), inv.y, bp.y), ay1 = __builtin_fmaf((float)(w2 & 0xFFFFu), inv.y, bm.y);
                    const float az0 = __builtin_fmaf((float)(w1 & 0xFFFFu), inv.z, bp.z), az1 = __builtin_fmaf((float)(w2 >> 16), inv.z, bm.z);
                    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)),
                                         __builtin_fmaxf(__builtin_fminf(az0, az1), t_min));
                    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)),
                                                     __builtin_fminf(__builtin_fmaxf(az0, az1), closest));
                    return tn <= tf && code != 0xFFFFFFFFu;          // unused child slots carry the sentinel code
                };
                float t0, t1, t2, t3;
                const bool h0 = slab(qa.x, qa.y, qa.z, qd.x, t0), h1 = slab(qa.w, qb.x, qb.y, qd.y, t1);
                const bool h2 = slab(qb.z, qb.w, qc.x, qd.z, t2), h3 = slab(qc.y, qc.z, qc.w, qd.w, t3);
                // Nearest child first, the others onto the stack far to near: a five-comparator network on (key, code) pairs,
                // key = bits of the entry distance (t_min > 0: positive floats order like integers; with an unusual t_min <= 0
                // the order is merely not by distance), misses sort last.  Branch-free: the three far codes are STORED
                // unconditionally at the running stack top (a store above the top is harmless: the stack has three spare
                // rows for it) and only the top moves conditionally -- as divergent branches this part cost more than the
                // slab tests.
                uint32_t k0 = h0 ? __float_as_uint(t0) : 0xFFFFFFFFu, k1 = h1 ? __float_as_uint(t1) : 0xFFFFFFFFu;
                uint32_t k2 = h2 ? __float_as_uint(t2) : 0xFFFFFFFFu, k3 = h3 ? __float_as_uint(t3) : 0xFFFFFFFFu;
                uint32_t c_near = qd.x, c1 = qd.y, c2 = qd.z, c3 = qd.w;
                auto cx = [](uint32_t& kx, uint32_t& cxv, uint32_t& ky, uint32_t& cyv) {
                    const bool sw = ky < kx;
                    const uint32_t ka = sw ? ky : kx, kb = sw ? kx : ky, ca = sw ? cyv : cxv, cb = sw ? cxv : cyv;
                    kx = ka; ky = kb; cxv = ca; cyv = cb;
                };
                cx(k0, c_near, k1, c1); cx(k2, c2, k3, c3); cx(k0, c_near, k2, c2); cx(k1, c1, k3, c3); cx(k1, c1, k2, c2);   // ascending: k0 nearest
                stk[sp * kBlock] = c3; sp += k3 != 0xFFFFFFFFu ? 1u : 0u;
                stk[sp * kBlock] = c2; sp += k2 != 0xFFFFFFFFu ? 1u : 0u;
                stk[sp * kBlock] = c1; sp += k1 != 0xFFFFFFFFu ? 1u : 0u;
                if (k0 != 0xFFFFFFFFu) {
                    node = c_near;
                } else {
                    --sp;
                    node = stk[sp * kBlock];
                }
            }
#if PT_BVH_POSTPONE
            // A lane that reaches a leaf puts it aside (one per lane) and goes on with its stack: it keeps working on inner
            // nodes while the wave collects enough leaves for a dense batch of primitive tests.  (The tests run later than
            // in stack order, so `closest` may shrink later: a few more visits, never another answer.)
            if (has && pend == 0xFFFFFFFFu && (int)node < 0 && node != 0xFFFFFFFFu) {
                pend = node;
                --sp;
                node = stk[sp * kBlock];
            }
            const bool at_leaf = has && pend != 0xFFFFFFFFu;
            const uint32_t leaf = pend;
#else
            const bool at_leaf = has && (int)node < 0 && node != 0xFFFFFFFFu;
            const uint32_t leaf = node;
#endif
            const unsigned long long leafs = __ballot(at_leaf);
            if (leafs != 0ull && ((uint32_t)__popcll(leafs) >= leaf_batch || __ballot(has && (int)node >= 0) == 0ull)) {
                if (at_leaf) {
                    const uint32_t first = leaf & 0x0FFFFFFFu, cnt = ((leaf >> 28) & 7u) + 1u;
                    // all ids (one aligned 16-byte load) and lead records (one 64-byte line) requested before the
                    // first test: one memory latency per leaf
                    const uint4 idv = *reinterpret_cast<const uint4*>(sc.bvh.ids + first);
                    const uint32_t w[kBvhMaxLeaf] = {idv.x, idv.y, idv.z, idv.w};
                    float4 r0[kBvhMaxLeaf];
#pragma unroll
                    for (uint32_t i = 0; i < kBvhMaxLeaf; ++i) r0[i] = sc.bvh.lead[first + (i < cnt ? i : 0u)];
#pragma unroll
                    for (uint32_t i = 0; i < kBvhMaxLeaf; ++i) {
                        if (i < cnt) {
                            if ((int)w[i] >= 0) {
                                sphere_test<true>(r0[i], o, d, t_min, closest, id, (int)w[i]);
                            } else {
                                const float4* rec = sc.bvh.rec + 3u * (first + i);
                                const float4 r1 = rec[1], r2 = rec[2];
                                triangle_test<true>(r0[i], r1, r2, o, d, t_min, closest, id, (int)(w[i] & 0x7FFFFFFFu));
                            }
                        }
                    }
#if PT_BVH_POSTPONE
                    pend = 0xFFFFFFFFu;
                    if (ANY && id >= 0) node = 0xFFFFFFFFu;
#else
                    if (ANY && id >= 0) {
                        node = 0xFFFFFFFFu;
                    } else {
                        --sp;
                        node = stk[sp * kBlock];
                    }
#endif
                }
            }
#if PT_BVH_POSTPONE
            if (has && node == 0xFFFFFFFFu && pend == 0xFFFFFFFFu) {   // this ray is done
#else
            if (has && node == 0xFFFFFFFFu) {            // this ray is done
#endif
                if (ANY) out[slot].z = id >= 0 ? 1.0f : 0.0f;
                else *reinterpret_cast<float2*>(&out[slot]) = make_float2(__int_as_float(id), closest);
                has = false;
            }
        } while ((uint32_t)__popcll(__ballot(has)) >= low_water);
    }
}

// Rays that share a wave should look alike: the lanes of a wave take the segment's rays in handing-out
// order, so that order is made one of similar rays -- a counting sort of the wave's segment by a 6-bit key, direction octant
// (the order in which a traversal visits the children of a node) x octant of the origin about the scene's centre.  Two
// passes over the ray planes: histogram in LDS, wave-wide exclusive prefix, then every slot takes the next position of its
// bucket (LDS atomic; the order inside a bucket is whatever the atomics give -- results are written by slot and the film by
// pixel, so nothing depends on it).  The order goes into the spare .w of the per-slot result records (aux), which the
// traversal does not touch.  Slots without a ray (shadow stage) sort into the last bucket; the traversal skips them as before.
#ifndef PT_BVH_SORT
#define PT_BVH_SORT 0        // measured (round 4, C4 accel 1, ms per launch): no sort 48.1, closest-hit stage sorted 50.4, both stages 53.0 -- rejected, kept as a measurement variant (1: stage 1, 2: both)
#endif
template <bool TMAX_IN_RAY>
PT_DEV void sort_segment(const SceneRef& sc, const float4* __restrict__ plane0, const float4* __restrict__ plane1, float4* __restrict__ aux,
                         uint32_t n, uint32_t* hist) {
    const uint32_t lane = threadIdx.x & 63u;
    const f3 ctr = mk(__builtin_fmaf(32767.5f, sc.bvh.grid_cell[0], sc.bvh.grid_min[0]), __builtin_fmaf(32767.5f, sc.bvh.grid_cell[1], sc.bvh.grid_min[1]),
                      __builtin_fmaf(32767.5f, sc.bvh.grid_cell[2], sc.bvh.grid_min[2]));
    auto key_of = [&](uint32_t s) -> uint32_t {
        const float4 r0 = plane0[s], r1 = plane1[s];
        if (TMAX_IN_RAY && r1.w == 0.0f) return 63u;
        return (r0.w < 0.0f ? 1u : 0u) | (r1.x < 0.0f ? 2u : 0u) | (r1.y < 0.0f ? 4u : 0u) |
               (r0.x > ctr.x ? 8u : 0u) | (r0.y > ctr.y ? 16u : 0u) | (r0.z > ctr.z ? 32u : 0u);
    };
    hist[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s = lane; s < n; s += 64u) atomicAdd(&hist[key_of(s)], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // exclusive prefix over the 64 buckets, one bucket per lane
    uint32_t v = hist[lane], incl = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += up; }
    __builtin_amdgcn_wave_barrier();
    hist[lane] = incl - v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s = lane; s < n; s += 64u) {
        const uint32_t pos = atomicAdd(&hist[key_of(s)], 1u);
        aux[pos].w = __uint_as_float(s);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

#ifndef PT_BVH_WAVES
#define PT_BVH_WAVES 5      // measured on C4: 4 -> 74.0 ms, 5 -> 70.2 ms, 6 (spills) -> 72.8 ms
#endif
template <bool MIS, bool OVF, bool DIFFUSE, bool LIST>
__global__ void __launch_bounds__(kBlock, PT_BVH_WAVES) k_paths_bvh(BounceArgs a) {
    extern __shared__ float4 lds[];
    __shared__ uint32_t s_hist[kBlock / 64][64];                 // sort_segment: bucket counters of each wave
    __shared__ WgTotals s_totals;
    if (threadIdx.x == 0u) wg_totals_init(s_totals);
    __syncthreads();                                             // (the traversal form has no barrier of its own before a wave can end)
    const SceneRef sc = stage_scene<kModeBvh>(a.sc, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    uint32_t* const hist = s_hist[threadIdx.x >> 6];
    const uint32_t nw = gridDim.x * (kBlock / 64);
    uint32_t n_first, seg_cap;
    launch_shape<OVF>(a, nw, n_first, seg_cap);
    const uint32_t seg_base = wave * seg_cap;
    const uint32_t n_chunks = (n_first + 63u) >> 6;
    const uint32_t W = a.film_w;
    const Queue q = {{a.q.q[0] + seg_base, a.q.q[1] + seg_base, a.q.q[2] + seg_base, a.q.q[3] + seg_base}};
    float4* const aux = a.aux + seg_base;
    float4* const sr0 = a.sray0 + seg_base;
    float4* const sr1 = a.sray1 + seg_base;
    uint32_t wave_shadow = 0, wave_vertices = 0, wave_depth = 0, wave_samples = 0;

    // ---- the wave's share of the batch -> its segment (chunk k of the batch belongs to wave k % nw)
    uint32_t n_in = 0;
    for (uint32_t it = 0, n_iter = (n_chunks + nw - 1u) / nw; it < n_iter; ++it) {
        const uint32_t chunk = it * nw + wave;
        const uint32_t pid = chunk * 64u + lane;
        const bool active = chunk < n_chunks && pid < n_first;
        if (active) {
            PathState p = parked_state();
            if (OVF) {
                p = unpack_state(a.ovf_in.q[0][pid], a.ovf_in.q[1][pid], a.ovf_in.q[2][pid], a.ovf_in.q[3][pid]);
            } else {
                uint32_t pix;
                divmod_magic(pid, a.np, a.np_magic, p.s_local, pix);
                divmod_magic(pix, W, a.film_w_magic, p.yl, p.px);
                uint32_t kx, py;
                pixel_key<LIST>(a, p, active, kx, py);
                camera_ray(a.cam, a.s_base + p.s_local, kx, py, p.o, p.d);
            }
            store_state(q, it * 64u + lane, p);     // dense: only the last chunk of the batch can be partial
        }
        n_in += (uint32_t)__popcll(__ballot(active));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

    for (uint32_t pass = 0; n_in != 0u; ++pass) {
        const uint32_t n_iter = (n_in + 63u) >> 6;
        // ---- stage 1: closest hits (rendering.rs:41)
        // (the camera rays of pass 0 of a level-0 launch are consecutive pixels already)
        const bool sorted = PT_BVH_SORT && (OVF || pass != 0u) && n_in > 64u;
        if (sorted) sort_segment<false>(sc, q.q[0], q.q[1], aux, n_in, hist);
        traverse_segment<false, false>(sc, q.q[0], q.q[1], aux, n_in, a.t_min, a.bvh_refill, a.bvh_leaf, sorted ? aux : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (MIS) {
            // ---- stage 2: light samples -> shadow rays (world.rs:251-267, rendering.rs:58-62)
            for (uint32_t it = 0; it < n_iter; ++it) {
                const uint32_t s = it * 64u + lane;
                const bool active = s < n_in;
                PathState p = parked_state();
                float4 h = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
                if (active) { unpack_ray(p, q.q[0][s], q.q[1][s]); h = aux[s]; }
                Vertex v;
                uint32_t kx, py;
                pixel_key<LIST>(a, p, active, kx, py);
                vertex_begin<true, DIFFUSE>(sc, p, active, __float_as_int(h.x), h.y, a.s_base + p.s_local, kx, py, v);
                if (active) {
                    sr0[s] = make_float4(v.hit.point.x, v.hit.point.y, v.hit.point.z, v.light_dir.x);
                    sr1[s] = make_float4(v.light_dir.y, v.light_dir.z, v.distance - a.t_min, v.need_shadow ? 1.0f : 0.0f);
                }
                wave_shadow += (uint32_t)__popcll(__ballot(v.need_shadow));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            // ---- stage 3: visibility (rendering.rs:62-65)
            const bool ssorted = PT_BVH_SORT >= 2 && n_in > 64u;
            if (ssorted) sort_segment<true>(sc, sr0, sr1, aux, n_in, hist);
            traverse_segment<true, true>(sc, sr0, sr1, aux, n_in, a.t_min, a.bvh_refill, a.bvh_leaf, ssorted ? aux : nullptr);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
        // ---- stage 4: shade and compact in place
        uint32_t out_n = 0;
        for (uint32_t it = 0; it < n_iter; ++it) {
            const uint32_t s = it * 64u + lane;
            const bool active = s < n_in;
            PathState p = parked_state();
            float4 h = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
            if (active) { p = unpack_state(q.q[0][s], q.q[1][s], q.q[2][s], q.q[3][s]); h = aux[s]; }
            uint32_t kx, py;
            pixel_key<LIST>(a, p, active, kx, py);
            const uint32_t sample = a.s_base + p.s_local;
            wave_vertices += (uint32_t)__popcll(__ballot(active));
            if (!OVF) {
                wave_depth = pass;
            } else if (__ballot(active && p.depth > wave_depth) != 0ull) {
                uint32_t m = active ? p.depth : 0u;
                for (int off = 32; off > 0; off >>= 1) { const uint32_t w2 = (uint32_t)__shfl_xor((int)m, off); m = w2 > m ? w2 : m; }
                wave_depth = __builtin_amdgcn_readfirstlane(m);
            }
            Vertex v;
            vertex_begin<MIS, DIFFUSE>(sc, p, active, __float_as_int(h.x), h.y, sample, kx, py, v);
            const bool visible = MIS && v.need_shadow && h.z == 0.0f;
            const bool alive = vertex_end<MIS, DIFFUSE, false>(sc, p, v, visible, sample, kx, py, a.min_depth, a.max_depth);
            if (active && !alive) a.lsamp[p.s_local * a.np + p.yl * W + p.px] = Rgb{p.L.x, p.L.y, p.L.z};
            wave_samples += (uint32_t)__popcll(__ballot(active && !alive));
            const unsigned long long mask = __ballot(alive);
            if (alive) store_state(q, out_n + lane_rank(mask), p);
            out_n += (uint32_t)__popcll(mask);
        }
        n_in = out_n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (n_in < a.export_below) break;
    }

    if (n_in != 0u) {                          // tail hand-off, as in k_paths
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(a.ovf_out_count, n_in);
        base = __shfl(base, 0);
        for (uint32_t j = lane; j < n_in; j += 64u) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) { const float4 t = a.q.q[k][seg_base + j]; a.ovf_out.q[k][base + j] = t; }
        }
    }
    if (lane == 0u) wave_totals<MIS, !OVF>(s_totals, kBlock / 64, a.stats, wave_shadow, wave_vertices, wave_samples, wave_depth);
}

[[maybe_unused]] static int scene_mode(const SceneView& sc, uint32_t accel) {
    return accel ? kModeBvh : (sc.n_objs <= kSmallObjs ? kModeLds : kModeTiled);
}
[[maybe_unused]] static size_t scene_lds_bytes(const SceneView& sc, int mode) {
    if (mode == kModeBvh) return (size_t)(kBvhStack + 3u) * kBlock * sizeof(uint32_t);   // + 3 rows: the traversal stores a visit's (up to) three far children before it knows how many there are
    return (mode == kModeLds ? (sc.blob_f4 ? sc.blob_f4 : 1u) : kTileF4) * sizeof(float4);
}
template <int MODE, bool DIFFUSE, bool LIST>
[[maybe_unused]] static void launch_paths_mode(const BounceArgs& a, uint32_t grid, size_t lds, hipStream_t st) {
    const bool mis = a.integrator == 0;
    const bool ovf = a.src_mode != 0u;     // continuation launch
    const dim3 g(grid), b(kBlock);
    if (mis && !ovf) hipLaunchKernelGGL((k_paths<MODE, true, false, DIFFUSE, LIST>), g, b, lds, st, a);
    else if (mis) hipLaunchKernelGGL((k_paths<MODE, true, true, DIFFUSE, LIST>), g, b, lds, st, a);
    else if (!ovf) hipLaunchKernelGGL((k_paths<MODE, false, false, DIFFUSE, LIST>), g, b, lds, st, a);
    else hipLaunchKernelGGL((k_paths<MODE, false, true, DIFFUSE, LIST>), g, b, lds, st, a);
}
#if PT_TU_BVH
template <bool DIFFUSE, bool LIST>
static void launch_paths_bvh_t(const BounceArgs& a, uint32_t grid, size_t lds, hipStream_t st) {
    const bool mis = a.integrator == 0;
    const bool ovf = a.src_mode != 0u;
    const dim3 g(grid), b(kBlock);
    if (mis && !ovf) hipLaunchKernelGGL((k_paths_bvh<true, false, DIFFUSE, LIST>), g, b, lds, st, a);
    else if (mis) hipLaunchKernelGGL((k_paths_bvh<true, true, DIFFUSE, LIST>), g, b, lds, st, a);
    else if (!ovf) hipLaunchKernelGGL((k_paths_bvh<false, false, DIFFUSE, LIST>), g, b, lds, st, a);
    else hipLaunchKernelGGL((k_paths_bvh<false, true, DIFFUSE, LIST>), g, b, lds, st, a);
}
#endif

}  // namespace PTK_IMPL
namespace ptk {
using namespace PTK_IMPL;
#if PT_TU_BVH
void PT_LAUNCH(launch_paths_bvh)(const BounceArgs& a, uint32_t grid, size_t lds, hipStream_t st, bool diffuse, bool list) {
    if (list) launch_paths_bvh_t<false, true>(a, grid, lds, st);        // pixel lists: the generic kernels only
    else if (diffuse) launch_paths_bvh_t<true, false>(a, grid, lds, st);
    else launch_paths_bvh_t<false, false>(a, grid, lds, st);
}
#endif
#if PT_TU_SPLIT
// k_paths_regen_split for the scene's material set (the plain iterations': no OrenNayar either / no Mirror)
typedef void (*RegenSplitKernel)(BounceArgs);
static RegenSplitKernel regen_split_kernel(const BounceArgs& a) {
    const bool mis = a.integrator == 0;
    if (a.sc.no_oren_nayar) return mis ? k_paths_regen_split<true, kMatsDiffuse> : k_paths_regen_split<false, kMatsDiffuse>;
    return mis ? k_paths_regen_split<true, kMatsNoMirror> : k_paths_regen_split<false, kMatsNoMirror>;
}
int PT_LAUNCH(regen_split_blocks_per_cu)(const BounceArgs& a, size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, regen_split_kernel(a), (int)kBlock, lds) != hipSuccess) return -1;
    return n;
}
void PT_LAUNCH(launch_regen_split)(const BounceArgs& b, uint32_t blocks, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL(regen_split_kernel(b), dim3(blocks), dim3(kBlock), lds, st, b);
}
#endif
#if PT_TU_MAIN
// the regenerating level-0 kernel a launch takes: compiled for the scene's material set; with the Mirror vertices batched
// (k_paths_regen_split, its own translation unit) when the host passes exchange memory
typedef void (*RegenKernel)(BounceArgs);
static RegenKernel regen_kernel(const BounceArgs& a) {
    const bool mis = a.integrator == 0;
    if (a.sc.diffuse_only) return mis ? k_paths_regen<true, kMatsDiffuse> : k_paths_regen<false, kMatsDiffuse>;
    if (a.sc.no_mirror) return mis ? k_paths_regen<true, kMatsNoMirror> : k_paths_regen<false, kMatsNoMirror>;
    return mis ? k_paths_regen<true, kMatsAll> : k_paths_regen<false, kMatsAll>;
}
// Workgroups of that kernel one CU holds at once, given the scene's LDS blob (0 if the query fails: the caller falls back
// to the compile-time occupancy).  The launch must not be larger than what is resident: the statically dealt quarter of
// the chunks of a wave that starts late is a serial tail.
uint32_t PT_LAUNCH(regen_blocks_per_cu)(const BounceArgs& a) {
    int n = 0;
    const size_t lds = scene_lds_bytes(a.sc, kModeLds);
    const uint32_t block = a.xchg ? kBlock : kRegenBlock;        // (the form that batches Mirror vertices keeps four waves per workgroup)
    if (a.xchg) n = PT_LAUNCH(regen_split_blocks_per_cu)(a, lds);
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, regen_kernel(a), (int)block, lds) != hipSuccess) n = -1;
    if (n < 0) return 0u;
    return (uint32_t)n * block / kBlock;                         // in units of four waves, like the grid the host passes
}
void PT_LAUNCH(launch_paths)(const BounceArgs& a, uint32_t grid, hipStream_t st) {
    const int mode = scene_mode(a.sc, a.accel);
    const size_t lds = scene_lds_bytes(a.sc, mode);
    const bool diffuse = a.sc.diffuse_only != 0u;
    if (a.pixels) {      // pixel-list renders: the generic kernels only (debug / replay entries, not the throughput path)
        if (mode == kModeLds) launch_paths_mode<kModeLds, false, true>(a, grid, lds, st);
        else if (mode == kModeTiled) launch_paths_mode<kModeTiled, false, true>(a, grid, lds, st);
        else PT_LAUNCH(launch_paths_bvh)(a, grid, lds, st, false, true);
        return;
    }
    if (mode == kModeLds && a.chunk_counter) {   // level-0 launch of a large batch: paths stay in registers (k_paths_regen*)
        const uint32_t block = a.xchg ? kBlock : kRegenBlock;    // grid = number of 4-wave units
        BounceArgs b = a;
        b.core_blocks = a.core_blocks * kBlock / block;          // (given in four-wave units like the grid)
        const uint32_t blocks = std::max(1u, grid * kBlock / block);
        if (a.xchg) PT_LAUNCH(launch_regen_split)(b, blocks, lds, st);
        else hipLaunchKernelGGL(regen_kernel(a), dim3(blocks), dim3(block), lds, st, b);
        return;
    }
    if (mode == kModeLds) { if (diffuse) launch_paths_mode<kModeLds, true, false>(a, grid, lds, st); else launch_paths_mode<kModeLds, false, false>(a, grid, lds, st); }
    else if (mode == kModeTiled) launch_paths_mode<kModeTiled, false, false>(a, grid, lds, st);   // scan-dominated: the variant buys nothing (measured)
    else PT_LAUNCH(launch_paths_bvh)(a, grid, lds, st, diffuse, false);
}
#endif
}  // namespace ptk
#if PT_TU_MAIN
namespace PTK_IMPL {

// ------------------------------------------------------------------ film resolve
// World::render_pixel's tail (world.rs:311-332).  One thread per tile pixel; the
// nb samples of the batch are added in sample order into an f64 sum, so the film
// does not depend on how paths were scheduled.
__global__ void __launch_bounds__(kBlock) k_resolve(ResolveArgs a) {
    // (With lanes -- pt_api.cpp -- a resolve becomes ready while the NEXT batch's regenerating launch holds every wave slot of
    // the device: it gets none until that launch runs dry (22 us of work took ~1 ms; a raised wave priority changes nothing,
    // the workgroups are simply not placed).  Hence the three buffer sets there: nobody waits for the resolve.)
    if (blockIdx.x == 0u && a.zero_words)
        for (uint32_t k = threadIdx.x; k < a.n_zero; k += kBlock) a.zero_words[k] = 0u;
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.np) return;
    double r = 0.0, g = 0.0, b = 0.0;
    if (a.load_film) { r = a.film[3 * (size_t)p]; g = a.film[3 * (size_t)p + 1]; b = a.film[3 * (size_t)p + 2]; }
    uint32_t s = 0;
#if PT_RESOLVE_UNROLL > 1
    // several samples' loads in flight before the (ordered) additions: beside resident path-kernel waves a resolve wave gets few
    // issue slots, and every exposed memory round trip counts
    for (; s + PT_RESOLVE_UNROLL <= a.nb; s += PT_RESOLVE_UNROLL) {
        Rgb v[PT_RESOLVE_UNROLL];
#pragma unroll
        for (int k = 0; k < PT_RESOLVE_UNROLL; ++k) v[k] = a.lsamp[(size_t)(s + k) * a.np + p];
#pragma unroll
        for (int k = 0; k < PT_RESOLVE_UNROLL; ++k) { r += (double)v[k].r; g += (double)v[k].g; b += (double)v[k].b; }
    }
#endif
    for (; s < a.nb; ++s) {
        const Rgb v = a.lsamp[(size_t)s * a.np + p];
        r += (double)v.r; g += (double)v.g; b += (double)v.b;                     // world.rs:311
    }
    if (a.store_film) { a.film[3 * (size_t)p] = r; a.film[3 * (size_t)p + 1] = g; a.film[3 * (size_t)p + 2] = b; }
    if (!a.finalize) return;
    double c[3] = {r / (double)a.spp_div, g / (double)a.spp_div, b / (double)a.spp_div};   // world.rs:315
    const bool want8 = a.out_rgba != nullptr || a.out_packed != nullptr;
    uint32_t q8 = 0xFF000000u;                                                    // alpha 255, world.rs:331
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (want8) {
            double gm = __builtin_sqrt(c[k]);                                     // gamma 2.0, world.rs:322-324
            double cl = gm < 0.0 ? 0.0 : (gm > 1.0 ? 1.0 : gm);                   // clamp keeps NaN
            double q = cl * 255.0;
            q8 |= (uint32_t)((q != q) ? (uint8_t)0 : (uint8_t)q) << (8 * k);      // `as u8`: truncation, NaN -> 0
        }
    }
    if (a.out_packed) {       // the multi-GPU send record: both film planes of the pixel in one 16-byte store
        reinterpret_cast<uint4*>(a.out_packed)[p] = make_uint4(__float_as_uint((float)c[0]), __float_as_uint((float)c[1]),
                                                               __float_as_uint((float)c[2]), q8);
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) a.out_linear[3 * (size_t)p + k] = (float)c[k];   // luminance_data, world.rs:318-319
    if (a.out_rgba) *reinterpret_cast<uint32_t*>(a.out_rgba + 4 * (size_t)p) = q8;
}
}  // namespace PTK_IMPL
#if !PT_MATH_EXACT
namespace PTK_IMPL {
// ------------------------------------------------------------------ multi-GPU film exchange (pt_multi.cpp)
// A device's tile -> one 16-byte record per pixel (12 B linear RGB + 4 B RGBA8), so that both film planes travel in
// ONE gather; rows beyond the tile (tiles are padded to the largest one) are left untouched.
__global__ void __launch_bounds__(kBlock) k_film_pack(const float* __restrict__ lin, const uint8_t* __restrict__ rgba,
                                                      uint32_t np, uint4* __restrict__ packed) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= np) return;
    uint4 v;
    v.x = __float_as_uint(lin[3 * (size_t)p]); v.y = __float_as_uint(lin[3 * (size_t)p + 1]); v.z = __float_as_uint(lin[3 * (size_t)p + 2]);
    v.w = rgba ? *reinterpret_cast<const uint32_t*>(rgba + 4 * (size_t)p) : 0u;
    packed[p] = v;
}
// The gathered tiles (device g's padded tile at recv + g * max_rows * W) -> the frame in image order.  Image row y lies
// in band y / band_rows, which device (band % n_dev) rendered as its tile row (band / n_dev) * band_rows + y % band_rows.
__global__ void __launch_bounds__(kBlock) k_film_unpack(const uint4* __restrict__ recv, uint32_t W, uint32_t H, uint32_t band_rows,
                                                        uint32_t n_dev, uint32_t max_rows, float* __restrict__ lin,
                                                        uint8_t* __restrict__ rgba) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= W * H) return;
    const uint32_t y = p / W, x = p - y * W;
    const uint32_t band = y / band_rows, g = band % n_dev;
    const uint32_t k = (band / n_dev) * band_rows + (y - band * band_rows);
    const uint4 v = recv[((size_t)g * max_rows + k) * W + x];
    lin[3 * (size_t)p] = __uint_as_float(v.x); lin[3 * (size_t)p + 1] = __uint_as_float(v.y); lin[3 * (size_t)p + 2] = __uint_as_float(v.z);
    if (rgba) *reinterpret_cast<uint32_t*>(rgba + 4 * (size_t)p) = v.w;
}
}  // namespace PTK_IMPL
namespace ptk {   // the film kernels have no division or sqrt in f32: one copy serves both modes
void launch_resolve(const ResolveArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(PTK_IMPL::k_resolve, dim3((a.np + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a);
}
void launch_film_pack(const float* lin, const uint8_t* rgba, uint32_t np, void* packed, hipStream_t st) {
    if (np) hipLaunchKernelGGL(PTK_IMPL::k_film_pack, dim3((np + kBlock - 1) / kBlock), dim3(kBlock), 0, st, lin, rgba, np, (uint4*)packed);
}
void launch_film_unpack(const void* recv, uint32_t W, uint32_t H, uint32_t band_rows, uint32_t n_dev, uint32_t max_rows, float* lin,
                        uint8_t* rgba, hipStream_t st) {
    if (W * H) hipLaunchKernelGGL(PTK_IMPL::k_film_unpack, dim3((W * H + kBlock - 1) / kBlock), dim3(kBlock), 0, st, (const uint4*)recv, W, H,
                                  band_rows, n_dev, max_rows, lin, rgba);
}
}  // namespace ptk
#endif
#endif      // PT_TU_MAIN
namespace PTK_IMPL {

// ------------------------------------------------------------------ debug: hit_scene on arbitrary rays
// HitRecord of the winning object (shape.rs:84-88 / 194-197 + base.rs:19-33): rec[8] = (t, point3, normal3, front_face)
PT_DEV void store_hit_record(const SceneRef& sc, int id, f3 o, f3 d, float t, float* rec) {
    Hit h;
    h.t = 0.0f; h.point = mk(0.f, 0.f, 0.f); h.normal = mk(0.f, 0.f, 0.f); h.front_face = false;
    if (id >= 0) h = finish_hit(sc.shape, id, load_mat(sc.mat, id).shape_tag, o, d, t);
    rec[0] = h.t; rec[1] = h.point.x; rec[2] = h.point.y; rec[3] = h.point.z;
    rec[4] = h.normal.x; rec[5] = h.normal.y; rec[6] = h.normal.z; rec[7] = h.front_face ? 1.0f : 0.0f;
}
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_debug_hit(SceneView scv, const float* __restrict__ rays6, uint32_t n,
                                                      float t_min, float t_max, int32_t* out_id, float* out_t, float* out_rec) {
    extern __shared__ float4 lds[];
    const SceneRef sc = stage_scene<MODE>(scv, lds);
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        uint32_t i = base + threadIdx.x;
        bool active = i < n;
        f3 o = parked_origin(), d = parked_dir();
        if (active) {
            o = mk(rays6[6 * (size_t)i], rays6[6 * (size_t)i + 1], rays6[6 * (size_t)i + 2]);
            d = normalize(mk(rays6[6 * (size_t)i + 3], rays6[6 * (size_t)i + 4], rays6[6 * (size_t)i + 5]));
        }
        int id; float t;
        scan_closest<MODE>(sc, o, d, t_min, t_max, id, t);
        if (active) { out_id[i] = id; out_t[i] = id >= 0 ? t : 0.0f; }
        if (active && out_rec) store_hit_record(sc, id, o, d, t, out_rec + 8 * (size_t)i);
    }
}
#if PT_TU_BVH
// the same through the BVH: every wave packs a contiguous slice of the rays into segment form and runs
// traverse_segment (the routine of k_paths_bvh) over it
__global__ void __launch_bounds__(kBlock) k_debug_hit_bvh(SceneView scv, const float* __restrict__ rays6, uint32_t n,
                                                          float t_min, float t_max, float4* p0, float4* p1, float4* res,
                                                          int32_t* out_id, float* out_t, float* out_rec) {
    extern __shared__ float4 lds[];
    const SceneRef sc = stage_scene<kModeBvh>(scv, lds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t nw = gridDim.x * (kBlock / 64);
    const uint32_t slice = (n + nw - 1u) / nw;
    const uint32_t base = wave * slice;
    if (base >= n) return;
    const uint32_t cnt = n - base < slice ? n - base : slice;
    for (uint32_t k = lane; k < cnt; k += 64u) {
        const size_t i = base + k;
        const f3 o = mk(rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]);
        const f3 d = normalize(mk(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]));
        p0[i] = make_float4(o.x, o.y, o.z, d.x);
        p1[i] = make_float4(d.y, d.z, t_max, 1.0f);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    traverse_segment<true, false>(sc, p0 + base, p1 + base, res + base, cnt, t_min, kRefillBelow, kLeafBatch);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (uint32_t k = lane; k < cnt; k += 64u) {
        const float4 r = res[base + k];
        const int id = __float_as_int(r.x);
        out_id[base + k] = id;
        out_t[base + k] = id >= 0 ? r.y : 0.0f;
        if (out_rec) {
            const float4 q0 = p0[base + k], q1 = p1[base + k];
            store_hit_record(sc, id, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), r.y, out_rec + 8 * (size_t)(base + k));
        }
    }
}
#endif      // PT_TU_BVH
#if PT_TU_MAIN
// ------------------------------------------------------------------ debug: the per-vertex functions on arbitrary inputs
// One thread per item; the SAME device functions the path kernels inline (pt_device.h, sample_light_point, camera_ray).
//   kFnBsdfEval    Material::bsdf_pdf (material.rs:86-91,139-148,221-265; mirror.rs:179-198)
//                  in[10] = dir_in3, wo3, normal3, eta -> out[4] = f3, pdf
//   kFnBsdfSample  Material::bsdf_pdf_sample (material.rs:29-40, mirror.rs:200-305)
//                  in[7] = dir_in3, normal3, eta; words[4] = r1, r2, lobe u, - -> out[8] = wo3, f3, pdf, cos
//   kFnShapeSample Shape::sample_surface_from_point (shape.rs:91-145, 200-242)
//                  in[9] = from3, target3, r1, r2, with_target -> out[8] = point3, pdf_omega, light_dir3, distance
//                  (direction and distance as rendering.rs:58-60 forms them)
//   kFnLightPoint  World::sample_light_point (world.rs:251-267)
//                  in[3] = from3; words[4] = index word, r1 word, r2 word, - -> out[8] = point3, emission3, pdf, light object
//   kFnCameraRay   Camera::get_ray_with_offset with the sample's jitter draws (camera.rs:139-147, world.rs:299)
//                  words[4] = x, y (top-down film row), sample, - -> out[8] = origin3, direction3, ox, oy
__global__ void __launch_bounds__(kBlock) k_debug_fn(DebugFnArgs a) {
    extern __shared__ float4 lds[];
    const SceneRef sc = stage_scene<kModeBvh>(a.sc, lds);      // records from global memory, nothing staged
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const float* in = a.in + (size_t)i * a.in_stride;
    const uint32_t* w = a.words ? a.words + 4 * (size_t)i : nullptr;
    float* out = a.out + (size_t)i * a.out_stride;
    if (a.op == kFnBsdfEval) {
        const Mat m = load_mat(sc.mat, (int)a.obj);
        f3 f; float pdf;
        bsdf_pdf(m, mk(in[0], in[1], in[2]), in[9], mk(in[3], in[4], in[5]), mk(in[6], in[7], in[8]), f, pdf);
        out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = pdf;
    } else if (a.op == kFnBsdfSample) {
        const Mat m = load_mat(sc.mat, (int)a.obj);
        f3 wo, f; float pdf, c;
        bsdf_pdf_sample(m, mk(in[0], in[1], in[2]), in[6], mk(in[3], in[4], in[5]), w[0], w[1], w[2], wo, f, pdf, c);
        out[0] = wo.x; out[1] = wo.y; out[2] = wo.z; out[3] = f.x; out[4] = f.y; out[5] = f.z; out[6] = pdf; out[7] = c;
    } else if (a.op == kFnShapeSample) {
        const Mat m = load_mat(sc.mat, (int)a.obj);
        const f3 from = mk(in[0], in[1], in[2]);
        f3 point, dir = mk(0.f, 0.f, 0.f); float pdf, dist = 0.0f;
        const bool with_target = in[8] != 0.0f;
        shape_sample(sc.shape, sc.mat, (int)a.obj, m.shape_tag, from, with_target, mk(in[3], in[4], in[5]), in[6], in[7], point, pdf, dir, dist);
        if (with_target) {                   // look-ahead form: the sampler produces no direction; report the point's
            const f3 to_light = point - from;
            dir = normalize(to_light); dist = length(to_light);
        }
        out[0] = point.x; out[1] = point.y; out[2] = point.z; out[3] = pdf;
        out[4] = dir.x; out[5] = dir.y; out[6] = dir.z; out[7] = dist;
    } else if (a.op == kFnLightPoint) {
        f3 point = mk(0.f, 0.f, 0.f), le = point, ld = point; float pdf = 0.0f, ll = 0.0f; int lobj = -1;
        if (sc.n_lights > 0u) sample_light_point<false>(sc, mk(in[0], in[1], in[2]), w[0], w[1], w[2], point, lobj, le, pdf, ld, ll);
        out[0] = point.x; out[1] = point.y; out[2] = point.z; out[3] = le.x; out[4] = le.y; out[5] = le.z;
        out[6] = pdf; out[7] = (float)lobj;
    } else if (a.op == kFnCameraRay) {
        f3 o, d;
        camera_ray(a.cam, w[2], w[0], w[1], o, d);
        uint32_t dc[4];
        philox4x32_draw(w[0], w[1], w[2], kDepthCamera, BLK_SURFACE, 0u, dc);
        out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
        out[6] = u01(dc[0]); out[7] = u01(dc[1]);
    }
}

// ------------------------------------------------------------------ per-object constants (pt_scene_upload)
// One thread per object: a triangle's unit normal into the spare w components of its shape record, 1 / area into the spare
// component of its material record (pt_device.h "scene records").  Evaluated by the expressions the per-vertex code
// used to run (triangle_constants), in this translation unit's arithmetic mode, on that mode's copy of the records.
__global__ void k_scene_setup(float4* shape, float4* mat, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = __float_as_uint(mat[2 * i].x);
    if (((bits >> 8) & 0xFFu) != SHAPE_TRIANGLE) return;
    float4 r0 = shape[3 * i], r1 = shape[3 * i + 1], r2 = shape[3 * i + 2];
    f3 normal; float pdf_area;
    triangle_constants(mk(r1.x, r1.y, r1.z), mk(r2.x, r2.y, r2.z), normal, pdf_area);
    r0.w = normal.x; r1.w = normal.y; r2.w = normal.z;
    shape[3 * i] = r0; shape[3 * i + 1] = r1; shape[3 * i + 2] = r2;
    mat[2 * i + 1].w = pdf_area;
}
#endif      // PT_TU_MAIN
}  // namespace PTK_IMPL
namespace ptk {
using namespace PTK_IMPL;
#if PT_TU_BVH
void PT_LAUNCH(launch_debug_hit_bvh)(const SceneView& sc, uint32_t grid, size_t lds, const float* rays6, uint32_t n, float t_min, float t_max,
                                     float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st) {
    // scratch: 3 planes of n float4 (two ray planes + result)
    hipLaunchKernelGGL(k_debug_hit_bvh, dim3(grid), dim3(kBlock), lds, st, sc, rays6, n, t_min, t_max, scratch,
                       scratch + n, scratch + 2 * (size_t)n, out_id, out_t, out_rec);
}
#endif
#if PT_TU_MAIN
void PT_LAUNCH(launch_scene_setup)(float4* shape, float4* mat, uint32_t n, hipStream_t st) {
    if (n == 0u) return;
    hipLaunchKernelGGL(k_scene_setup, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, shape, mat, n);
}
void PT_LAUNCH(launch_debug_fn)(const DebugFnArgs& a, hipStream_t st) {
    if (a.n == 0u) return;
    hipLaunchKernelGGL(k_debug_fn, dim3((a.n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a);
}
void PT_LAUNCH(launch_debug_hit)(const SceneView& sc, uint32_t accel, const float* rays6, uint32_t n, float t_min,
                                 float t_max, float4* scratch, int32_t* out_id, float* out_t, float* out_rec, hipStream_t st) {
    const int mode = scene_mode(sc, accel);
    const size_t lds = scene_lds_bytes(sc, mode);
    uint32_t grid = (n + kBlock - 1) / kBlock;
    if (grid > 2048u) grid = 2048u;
    if (grid == 0u) grid = 1u;
    if (mode == kModeLds)
        hipLaunchKernelGGL(k_debug_hit<kModeLds>, dim3(grid), dim3(kBlock), lds, st, sc, rays6, n, t_min, t_max, out_id, out_t, out_rec);
    else if (mode == kModeTiled)
        hipLaunchKernelGGL(k_debug_hit<kModeTiled>, dim3(grid), dim3(kBlock), lds, st, sc, rays6, n, t_min, t_max, out_id, out_t, out_rec);
    else
        PT_LAUNCH(launch_debug_hit_bvh)(sc, grid, lds, rays6, n, t_min, t_max, scratch, out_id, out_t, out_rec, st);
}
#endif      // PT_TU_MAIN

}  // namespace ptk
