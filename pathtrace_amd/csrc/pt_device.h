// pt_device.h -- device-side arithmetic of the wavefront path tracer (gfx950).
//
// FP32 restatement of the reference's per-vertex arithmetic (src/math.rs,
// src/camera.rs, src/objects/*.rs).  The translation unit is compiled with
// -ffp-contract=off: an FMA appears only where this file writes __builtin_fmaf
// (dot, cross, a*s+b).  vector / scalar is vector * (1/scalar) with an IEEE
// reciprocal; sin/cos(2*pi*u) is a fixed polynomial.  sqrtf and '/' are the
// correctly-rounded forms (hipcc default), never -ffast-math: NaN ordering is
// part of the reference's behaviour.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

// The file is compiled twice into one library (see Makefile):
//   PT_MATH_EXACT=1  IEEE correctly-rounded 1/x, a/b, sqrt: every operation is reproducible on the
//                    host, so this mode is bit-identical to the f32 oracle (the parity tests' proof
//                    that the kernel logic is right).  PtRenderParams.exact_math = 1.
//   PT_MATH_EXACT=0  the hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 (default mode, 22 % faster: the
//                    correctly-rounded expansions are ~10 instructions + hazard nops each).  Held to
//                    the FP32 tolerance against the f64 oracle like the exact mode.
#ifndef PT_MATH_EXACT
#define PT_MATH_EXACT 0
#endif
#if PT_MATH_EXACT
#define PTD_NS ptd_exact
#else
#define PTD_NS ptd_fast
#endif

namespace PTD_NS {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;   // x / pi is evaluated as x * kInvPi (f32 arithmetic spec)
constexpr float kInf = __builtin_huge_valf();

// Every division, reciprocal and square root of the device code goes through these three.
#if PT_MATH_EXACT
PT_DEV float pt_rcp(float x) { return 1.0f / x; }
PT_DEV float pt_div(float a, float b) { return a / b; }
PT_DEV float pt_sqrt(float x) { return __builtin_sqrtf(x); }
#else
PT_DEV float pt_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PT_DEV float pt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
PT_DEV float pt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#endif

// ------------------------------------------------------------------ Vector3 (math.rs:3-244)
struct f3 {
    float x, y, z;
};
PT_DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }      // math.rs:139
PT_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }      // math.rs:160
PT_DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }         // math.rs:173
PT_DEV f3 operator*(float s, f3 a) { return mk(a.x * s, a.y * s, a.z * s); }         // math.rs:186
PT_DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }      // math.rs:195
PT_DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }                           // math.rs:234
PT_DEV f3 operator/(f3 a, float s) { float inv = pt_rcp(s); return mk(a.x * inv, a.y * inv, a.z * inv); }   // math.rs:208
PT_DEV float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }     // math.rs:24
PT_DEV float msub(float a, float b, float c, float d) { return __builtin_fmaf(a, b, -(c * d)); }
PT_DEV f3 cross(f3 a, f3 b) {                                                        // math.rs:29
    return mk(msub(a.y, b.z, a.z, b.y), msub(a.z, b.x, a.x, b.z), msub(a.x, b.y, a.y, b.x));
}
PT_DEV float length(f3 a) { return pt_sqrt(dot(a, a)); }                     // math.rs:38
#ifndef PT_RSQ_NORMALIZE
#define PT_RSQ_NORMALIZE 1     // fast arithmetic normalises with one v_rsq_f32 (round 3; same-box A/B on C2: 6.19 -> 6.10 ms per step)
#endif
#if PT_MATH_EXACT || !PT_RSQ_NORMALIZE
PT_DEV f3 normalize(f3 a) { float len = length(a); return len > 0.0f ? a / len : a; }   // math.rs:48-51
// to_light.length() and to_light.normalize() of the same vector (rendering.rs:59-60, shape.rs:218-221)
PT_DEV f3 normalize_len(f3 a, float& len) { len = length(a); return len > 0.0f ? a / len : a; }
#else
// Fast arithmetic: ONE v_rsq_f32 (1 ulp) instead of sqrt + rcp + a compare and three selects.  A zero vector stays
// zero as in math.rs:48-51: 0 * rsq(1e-36) = 0 (the clamp only ever acts on |a| < 1e-18).
PT_DEV f3 normalize(f3 a) { return a * __builtin_amdgcn_rsqf(__builtin_fmaxf(dot(a, a), 1e-36f)); }
PT_DEV f3 normalize_len(f3 a, float& len) {
    const float l2 = dot(a, a), inv = __builtin_amdgcn_rsqf(__builtin_fmaxf(l2, 1e-36f));
    len = l2 * inv;
    return a * inv;
}
#endif
PT_DEV float luminance(f3 a) { return 0.2126f * a.x + 0.7152f * a.y + 0.0722f * a.z; }  // math.rs:133
PT_DEV bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
PT_DEV f3 madd(f3 a, float s, f3 b) {                                                // a*s + b
    return mk(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z));
}
// tangent*x + bitangent*y + normal*z (material.rs:121)
PT_DEV f3 frame3(f3 t, float x, f3 b, float y, f3 n, float z) { return madd(n, z, madd(b, y, t * x)); }
PT_DEV bool finite3(f3 v) { return __builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z); }

// sin/cos(2*pi*u), u in (0,1): quadrant reduction on u, cephes sinf/cosf kernels.  Fast arithmetic: the hardware's
// v_sin_f32 / v_cos_f32, which take their argument in revolutions (two instructions instead of ~25).
PT_DEV void sincos2pi(float u, float& s, float& c) {
#if !PT_MATH_EXACT && !defined(PT_NO_NATIVE_SINCOS)
    s = __builtin_amdgcn_sinf(u); c = __builtin_amdgcn_cosf(u);
    return;
#endif
    float k = __builtin_rintf(u * 4.0f);
    float r = __builtin_fmaf(k, -0.25f, u);
    float t = r * 6.28318530717958647692f;
    float z = t * t;
    float sp = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float st = __builtin_fmaf(sp * z, t, t);
    float cp = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                              4.166664568298827e-2f);
    float ct = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    float ss = (q & 1) ? ct : st;
    float cc = (q & 1) ? st : ct;
    if (q == 1 || q == 2) cc = -cc;
    if (q == 2 || q == 3) ss = -ss;
    s = ss; c = cc;
}

// ------------------------------------------------------------------ counter RNG
// Philox4x32-R (Random123; Salmon et al., SC'11), the round count a compile-time parameter.  The render draws use
// R = 7 -- the paper's "Crush-resistant minimum" (its Table 2: Philox4x32-7 passes SmallCrush, Crush and BigCrush; 10 is
// the default with a safety margin) -- since round 3: three rounds less are 24 VALU instructions per call, C2 -2 % (same-box
// A/B, profiles/r03/).  The round function is pinned by the three Random123 known answers at R = 10 and the zero-input
// one at R = 7 (tests/test_rng.py, through the oracle's independent copy; the device copy is compared bit for bit with the
// oracle by every exact-mode parity test).
// ctr = (x, y, sample, depth) -- (x, y) = the two words of the reference's per-pixel seed
// (y<<32)|x (src/main.rs:51) --, key = (block, 0): the key is a literal at every call, so the round keys are
// constants and the key schedule costs no instruction (with the pixel in the key it was 20 VALU adds per call).
#ifndef PT_PHILOX_ROUNDS
#define PT_PHILOX_ROUNDS 7
#endif
PT_DEV void philox4x32_draw(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < PT_PHILOX_ROUNDS; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Draw blocks of a vertex, ctr = (x, y, sample, depth), key = (block, 0):
//   BLK_SURFACE: [0] light r1 [1] light r2 [2] bsdf r1 [3] bsdf r2     (shape.rs:111-112,211-212; material.rs:100-101,
//                                                                      mirror.rs:42-43)
//   BLK_CHOICE:  [0] light index (world.rs:255) [1] Mirror lobe u (mirror.rs:232) [2], [3] spare
// The Russian-roulette uniform (rendering.rs:100) is made of the bits of BLK_SURFACE that u01() never looks at (the
// low 9 bits of words 0 and 1, bits 8..4 of word 2: vertex_begin, oracle rr_word()).  BLK_CHOICE is generated only
// where it can decide something -- more than one light, or a Mirror surface -- so a vertex of a diffuse one-light scene
// costs ONE Philox call at every depth.
// This addressing -- 7 rounds, ctr (x, y, sample, depth), key (block, 0), the word assignment above, camera jitter at
// depth 0xFFFFFFFF -- is part of the ABI since round 3: spp_offset resume, pt_render_pixels replays and any archived film
// depend on it (DESIGN.md 1 "RNG").
enum { BLK_SURFACE = 0, BLK_CHOICE = 1 };
constexpr uint32_t kDepthCamera = 0xFFFFFFFFu;
// 23-bit uniform on the open interval (0,1): (2k+1)/2^24, exact in f32.
// Evaluated without an integer-to-float conversion: bits(1 + k/2^23) minus (1 - 2^-24); the sum is representable,
// so the one rounding of the add is exact and the value is (2k+1)/2^24 bit for bit.
// (r >> 9) | 0x3F800000 in ONE instruction: v_alignbit_b32 takes bits [40:9] of {0x7F, r}.
PT_DEV float u01(uint32_t r) { return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, r, 9u)) + (-0.99999994f); }

// ------------------------------------------------------------------ scene records
// shape record, 3 float4 per object:
//   sphere   : r0 = (cx, cy, cz, radius), r1 = (1/radius, -, -, -)
//   triangle : r0 = (v0, nx), r1 = (e1 = v1-v0, ny), r2 = (e2 = v2-v0, nz)      n = normalize(e1 x e2), the geometric normal
// material record, 2 float4 per object:
//   m0 = (mat_tag | shape_tag<<8 | emits<<16 as bits, p0, p1, p2), m1 = (p3, p4, p5, triangle: 1/area)
// The triangle's unit normal (shape.rs:195, :213) and 1/area (:214-215, :225) are the same numbers at every hit and
// every light sample: k_scene_setup evaluates them ONCE per object at pt_scene_upload, on the device, with the very
// expressions of this header and in both arithmetic modes (triangle_constants below; each mode has its own copy of
// the records), so they are bit for bit what evaluating them per vertex gave -- 16 + 20 instructions and five
// transcendentals less per vertex of the reference scene.
//   lambert/emissive: p0..2 = colour; mirror: p0 = roughness, p1..3 = colour, p4 = metallic, p5 = ior;
//   oren-nayar: p0..2 = albedo, p3 = A, p4 = B (material.rs:182-193)
enum { SHAPE_SPHERE = 0, SHAPE_TRIANGLE = 1 };
enum { MAT_LAMBERT = 0, MAT_EMISSIVE = 1, MAT_MIRROR = 2, MAT_OREN_NAYAR = 3 };

struct Mat {
    uint32_t tag, shape_tag, emits;
    f3 color;
    float roughness, metallic, ior, on_a, on_b;
};
PT_DEV Mat load_mat(const float4* __restrict__ mat, int id) {
    float4 m0 = mat[2 * id], m1 = mat[2 * id + 1];
    uint32_t bits = __float_as_uint(m0.x);
    Mat m;
    m.tag = bits & 0xFF; m.shape_tag = (bits >> 8) & 0xFF; m.emits = (bits >> 16) & 1;
    m.roughness = 0.0f; m.metallic = 0.0f; m.ior = 1.0f; m.on_a = 1.0f; m.on_b = 0.0f;
    if (m.tag == MAT_MIRROR) {
        m.roughness = m0.y; m.color = mk(m0.z, m0.w, m1.x); m.metallic = m1.y; m.ior = m1.z;
    } else {
        m.color = mk(m0.y, m0.z, m0.w);
        m.on_a = m1.x; m.on_b = m1.y;
    }
    return m;
}

struct Hit {       // HitRecord, base.rs:6-15
    f3 point, normal;
    float t;
    bool front_face;
};
// HitRecord::new, base.rs:19-33
PT_DEV void face_forward(Hit& h, f3 outward, f3 dir) {
    h.front_face = dot(dir, outward) < 0.0f;
    h.normal = h.front_face ? outward : -outward;
}

// Tail of SphereShape::hit / TriangleShape::hit for the winning object of the
// scan (shape.rs:84-88, 194-197): point, outward normal, face-forwarding.
PT_DEV Hit finish_hit(const float4* __restrict__ shape, int id, uint32_t shape_tag, f3 o, f3 d, float t) {
    Hit h;
    h.t = t;
    h.point = madd(d, t, o);                                    // ray.at(t), camera.rs:18-20
    f3 outward;
    float4 r0 = shape[3 * id];
    if (shape_tag == SHAPE_SPHERE) {
        outward = (h.point - mk(r0.x, r0.y, r0.z)) * shape[3 * id + 1].x;   // shape.rs:86; r1.x = 1/radius
    } else {
        outward = mk(r0.w, shape[3 * id + 1].w, shape[3 * id + 2].w);           // shape.rs:195, from triangle_constants
    }
    face_forward(h, outward, d);
    return h;
}

// ------------------------------------------------------------------ shape sampling (shape.rs)
// local frame: material.rs:112-119, mirror.rs:21-27
// up x n written out for the two constant `up` vectors: X x n = (0, -n.z, n.y), Y x n = (n.z, 0, -n.x)
PT_DEV void frame_of(f3 n, f3& tangent, f3& bitangent) {
    const bool use_x = __builtin_fabsf(n.y) > 0.999f;
    const f3 raw = mk(use_x ? 0.0f : n.z, use_x ? -n.z : 0.0f, use_x ? n.y : -n.x);
    const float len2 = __builtin_fmaf(raw.z, raw.z, n.z * n.z);      // raw . raw: the other component is +-n.z in both cases
    const float len = pt_sqrt(len2);
    tangent = len > 0.0f ? raw / len : raw;                          // math.rs:48-51
    bitangent = cross(n, tangent);
}
// SphereShape::sample_surface_from_point, shape.rs:91-145.  with_target: the MIS
// look-ahead form (point given, no draws; dir / dist are not produced).
// Sampled form: dir / dist = the unit direction from `from` to the sampled point and its length -- what
// rendering.rs:58-60 derives from the point again (to_light = point - from, length, normalize).  The f32 specification
// takes them where they come from: the point IS from + t * direction with the unit cone direction, so
// light_dir = direction and distance = t, and t is the near root of the cone ray in the cone's own coordinates,
// t = dc cos(theta) - sqrt(r^2 - dc^2 sin^2(theta)) with dc = |center - from| (same real numbers as the reference's
// quadratic, shape.rs:130-137; no second normalisation, no quadratic on world coordinates: -2 sqrt/rcp, -25 VALU).
// For an observer INSIDE the sphere that root is negative: direction and sign are then flipped at the end, which is
// what the reference's point - from gives there.
PT_DEV void sphere_sample(float4 r0, f3 from, bool with_target, f3 target, float r1, float r2, f3& point,
                          float& pdf_omega, f3& dir, float& dist) {
    f3 center = mk(r0.x, r0.y, r0.z);
    float radius = r0.w;
    f3 to_center = center - from;
    float distance_sq = dot(to_center, to_center);
    float sin_theta_max_sq = pt_div(radius * radius, distance_sq);
    float cos_theta_max = pt_sqrt(__builtin_fmaxf(1.0f - sin_theta_max_sq, 0.0f));
    // 1 - cos_theta_max = 1 - sqrt(1 - s) loses its digits in f32 for a small or distant light;
    // s / (1 + sqrt(1 - s)) is the same number.  Observer inside the sphere: cos_theta_max = 0.
    float omc = pt_div(sin_theta_max_sq, 1.0f + cos_theta_max);
    if (sin_theta_max_sq > 1.0f) omc = 1.0f;
    float solid_angle = 2.0f * kPi * omc;
    pdf_omega = pt_rcp(solid_angle);
    if (with_target) { point = target; return; }
    float x1 = r1 * omc;                                   // 1 - cos_theta  (shape.rs:114)
    float cos_theta = 1.0f - x1;
    float sin2_theta = __builtin_fmaxf(x1 * (2.0f - x1), 0.0f);           // (1-c)(1+c)
    float sin_theta = pt_sqrt(sin2_theta);
    float sphi, cphi;
    sincos2pi(r2, sphi, cphi);
    float dc = pt_sqrt(distance_sq);
    f3 w = dc > 0.0f ? to_center / dc : to_center;                        // normalize(to_center), math.rs:48-51
    f3 u, v;
    frame_of(w, u, v);
    // normalised once (Ray::new, shape.rs:128)
    dir = normalize(frame3(u, sin_theta * cphi, v, sin_theta * sphi, w, cos_theta));
    // deliberate deviation: the discriminant is clamped at 0 (reference: unguarded sqrt, shape.rs:136)
    float disc = __builtin_fmaf(-distance_sq, sin2_theta, radius * radius);
    dist = __builtin_fmaf(dc, cos_theta, -pt_sqrt(__builtin_fmaxf(disc, 0.0f)));
    point = madd(dir, dist, from);
    // `from` inside the sphere (an enclosing light: dc < r): the near root is negative, the point lies BEHIND the cone
    // direction, and what rendering.rs:58-60 derives from it is distance = |t|, light_dir = -direction (shape.rs:139-144)
    if (dist < 0.0f) { dir = -dir; dist = -dist; }
}
// TriangleShape::sample_surface_from_point, shape.rs:200-242; dir / dist as above (here the reference itself forms
// them, :218-221, and rendering.rs:58-60 forms the same values again)
// normal / pdf_area: the triangle's unit normal and 1 / area (shape.rs:213-215,225), from triangle_constants
PT_DEV void triangle_constants(f3 e1, f3 e2, f3& normal, float& pdf_area) {
    f3 cr = cross(e1, e2);
    normal = normalize(cr);
    float area = length(cr) * 0.5f;
    pdf_area = pt_rcp(area);
}
PT_DEV void triangle_sample(f3 v0, f3 e1, f3 e2, f3 normal, float pdf_area, f3 from, bool with_target, f3 target, float r1, float r2,
                            f3& point, float& pdf_omega, f3& dir, float& dist) {
    if (with_target) {
        point = target;
    } else {
        float sqrt_r1 = pt_sqrt(r1);
        float u = 1.0f - sqrt_r1;
        float v = r2 * sqrt_r1;
        point = madd(e2, v, madd(e1, u, v0));
    }
    f3 to_light = point - from;
#if PT_MATH_EXACT || !PT_RSQ_NORMALIZE
    float d = length(to_light);
    f3 light_dir = to_light / d;                     // shape.rs:218-221
#else
    float d;
    f3 light_dir = normalize_len(to_light, d);
#endif
    float cos_light = __builtin_fabsf(dot(normal, -light_dir));
    pdf_omega = cos_light > 1e-8f ? pt_div(pdf_area * (d * d), cos_light) : 1e-8f;
    dist = d;
    dir = d > 0.0f ? light_dir : to_light;           // Vector3::normalize leaves a zero vector as it is (math.rs:48-51)
}
PT_DEV void shape_sample(const float4* __restrict__ shape, const float4* __restrict__ mat, int id, uint32_t shape_tag, f3 from,
                         bool with_target, f3 target, float r1, float r2, f3& point, float& pdf_omega, f3& dir, float& dist) {
    float4 r0 = shape[3 * id];
    if (shape_tag == SHAPE_SPHERE) {
        sphere_sample(r0, from, with_target, target, r1, r2, point, pdf_omega, dir, dist);
    } else {
        float4 q1 = shape[3 * id + 1], q2 = shape[3 * id + 2];
        triangle_sample(mk(r0.x, r0.y, r0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), mk(r0.w, q1.w, q2.w), mat[2 * id + 1].w,
                        from, with_target, target, r1, r2, point, pdf_omega, dir, dist);
    }
}

// ------------------------------------------------------------------ materials
// cosine-weighted direction: material.rs:93-122 / :267-295
PT_DEV f3 cosine_sample(f3 n, float r1, float r2) {
    float sphi, cphi;
    sincos2pi(r1, sphi, cphi);
    float cos_theta = pt_sqrt(r2);
    float sin_theta = pt_sqrt(1.0f - cos_theta * cos_theta);
    float x = sin_theta * cphi, y = sin_theta * sphi, z = cos_theta;
    f3 t, b;
    frame_of(n, t, b);
    return normalize(frame3(t, x, b, y, n, z));
}
PT_DEV float powi5(float x) { float x2 = x * x; return x2 * x2 * x; }

// ---- Mirror (mirror.rs)
PT_DEV f3 mirror_f(const Mat& m, float cos_theta) {                        // mirror.rs:126-132
    float f0d = pt_div(1.0f - m.ior, 1.0f + m.ior);
    f0d = f0d * f0d;
    f3 f0 = mk(f0d, f0d, f0d) * (1.0f - m.metallic) + m.color * m.metallic;
    return f0 + (mk(1.0f, 1.0f, 1.0f) - f0) * powi5(1.0f - cos_theta);
}
PT_DEV float mirror_g1(const Mat& m, float cos_theta) {                    // mirror.rs:136-149
    if (cos_theta <= 0.0f) return 0.0f;
    float alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    float cos2 = cos_theta * cos_theta;
    float term = alpha2 + (1.0f - alpha2) * cos2;
    return pt_div(2.0f * cos_theta, cos_theta + pt_sqrt(term));
}
PT_DEV float mirror_lambda(float alpha2, float c) {                        // mirror.rs:165-172
    float c2 = c * c;
    float num = pt_sqrt(alpha2 + (1.0f - alpha2) * c2);
    return pt_div(num - c, 2.0f * c);
}
PT_DEV float mirror_g(const Mat& m, float ci, float co) {                  // mirror.rs:153-175
    if (ci <= 0.0f || co <= 0.0f) return 0.0f;
    float alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    return pt_rcp(1.0f + mirror_lambda(alpha2, ci) + mirror_lambda(alpha2, co));
}
PT_DEV float ggx_d(float alpha2, float n_h) {                              // mirror.rs:69-70
    float denom = (n_h * n_h) * (alpha2 - 1.0f) + 1.0f;
    return pt_div(alpha2, kPi * denom * denom);
}
// Mirror::brdf (mirror.rs:62-88) and Mirror::btdf (:90-124) in one body: a wave that shades GGX vertices usually has
// reflecting and transmitting lanes side by side, and as two functions behind a divergent branch the parts they share --
// the half vector's normalisation, D, G, F: 55 of ~85 instructions each -- ran twice.  Every lane still evaluates its own
// case's expressions in the reference's order (i * 1 is i exactly, so `i * (1 | eta) + o` is `i + o` or `i * eta + o`).
PT_DEV void mirror_eval(const Mat& m, f3 dir_in, float eta, f3 o, f3 n, bool is_refl, f3& f, float& pdf) {
    f3 i = -dir_in;
    float alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    f3 h = normalize(i * (is_refl ? 1.0f : eta) + o);                          // :64 / :97
    if (!is_refl) h = -h;
    float n_h = dot(n, h);
    float d = ggx_d(alpha2, n_h);
    float ni = dot(n, i), no = dot(n, o);
    float i_n = is_refl ? __builtin_fmaxf(ni, 0.0f) : __builtin_fabsf(ni);      // :71-72 / :104-105
    float o_n = is_refl ? __builtin_fmaxf(no, 0.0f) : __builtin_fabsf(no);
    float g = mirror_g(m, i_n, o_n);
    float i_h = dot(i, h);
    float cos_theta = is_refl ? __builtin_fmaxf(i_h, 0.0f) : __builtin_fabsf(i_h);
    f3 fr = mirror_f(m, cos_theta);
    if (is_refl) {
        float denom_brdf = 4.0f * i_n * o_n;
        f = d * g * fr / denom_brdf;
        pdf = pt_div(d * __builtin_fabsf(n_h), 4.0f * __builtin_fabsf(i_h));
    } else {
        float o_h = dot(o, h);
        float denom_term = eta * i_h + o_h;
        f = (mk(1.0f, 1.0f, 1.0f) - fr) * d * g * __builtin_fabsf(i_h) * __builtin_fabsf(o_h) /
            (i_n * o_n * denom_term * denom_term);
        float jac = pt_div(__builtin_fabsf(o_h), denom_term * denom_term);
        pdf = d * __builtin_fabsf(n_h) * jac;
    }
}
// Mirror::sample_ggx_vndf, mirror.rs:17-60
PT_DEV f3 mirror_vndf(const Mat& m, f3 view, f3 n, float r1, float r2) {
    float alpha = m.roughness * m.roughness;
    f3 tangent, bitangent;
    frame_of(n, tangent, bitangent);
    f3 vl = mk(dot(view, tangent), dot(view, bitangent), dot(view, n));
    f3 vh = normalize(mk(alpha * vl.x, alpha * vl.y, vl.z));
    float lensq = vh.x * vh.x + vh.y * vh.y;
    f3 t1 = lensq > 0.0f ? mk(-vh.y, vh.x, 0.0f) * pt_rcp(pt_sqrt(lensq)) : mk(1.0f, 0.0f, 0.0f);
    f3 t2 = cross(vh, t1);
    float r = pt_sqrt(r1);
    float sphi, cphi;
    sincos2pi(r2, sphi, cphi);
    float p1 = r * cphi;
    float p2 = r * sphi;
    float s = 0.5f * (1.0f + vh.z);
    p2 = (1.0f - s) * pt_sqrt(1.0f - p1 * p1) + s * p2;
    float p3 = pt_sqrt(__builtin_fmaxf(1.0f - p1 * p1 - p2 * p2, 0.0f));
    f3 nh = frame3(t1, p1, t2, p2, vh, p3);
    f3 ne = normalize(mk(alpha * nh.x, alpha * nh.y, __builtin_fmaxf(nh.z, 0.0f)));
    return normalize(frame3(tangent, ne.x, bitangent, ne.y, n, ne.z));
}
// Mirror::bsdf_pdf_sample, mirror.rs:200-305
PT_DEV void mirror_sample(const Mat& m, f3 dir_in, float eta, f3 n, float r1, float r2, float u_lobe, f3& wo, f3& f,
                          float& pdf, float& cos_out) {
    f3 i = -dir_in;
    float i_dot_n = dot(i, n);
    f3 h = mirror_vndf(m, i, n, r1, r2);
    float i_h = dot(i, h);
    wo = n; f = mk(0.0f, 0.0f, 0.0f); pdf = 1.0f; cos_out = 0.0f;      // the failure tuple (mirror.rs:216)
    if (i_h <= 0.0f) return;
    f3 fr = mirror_f(m, i_h);
    float sin2_i = 1.0f - i_h * i_h;
    float cos2_t = 1.0f - (eta * eta) * sin2_i;
    bool tir = cos2_t < 0.0f;
    float rr_f = fr.x;
    if (tir || m.metallic > 0.99f) { rr_f = 1.0f; fr = mk(1.0f, 1.0f, 1.0f); }
    bool is_reflect = u_lobe < rr_f;
    float alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    float n_h = dot(n, h);
    float d = ggx_d(alpha2, n_h);
    // Both lobes in one body (as mirror_eval): the outgoing direction's normalisation, G, G1 and the VNDF pdf are common.
    // cos_t is only read by transmitting lanes (total internal reflection reflects: rr_f = 1).
    float cos_t = pt_sqrt(cos2_t);
    f3 o = is_reflect ? 2.0f * i_h * h - i : h * (eta * i_h - cos_t) - i * eta;      // :240 / :270
    f3 on = normalize(o);
    float n_on = dot(n, on);
    float o_n = is_reflect ? __builtin_fmaxf(n_on, 0.0f) : __builtin_fabsf(n_on);
    float i_n = is_reflect ? __builtin_fmaxf(i_dot_n, 0.0f) : __builtin_fabsf(i_dot_n);
    float g = mirror_g(m, i_n, o_n);
    float g1v = mirror_g1(m, i_n);
    float pdf_vndf = pt_div(g1v * d * __builtin_fmaxf(i_h, 0.0f), i_n);
    f3 val; float p;
    if (is_reflect) {
        float denom_brdf = 4.0f * i_n * o_n;
        val = fr * d * g / (denom_brdf * rr_f);
        p = pt_div(pdf_vndf, 4.0f * __builtin_fabsf(i_h));
    } else {
        float o_h = dot(on, h);
        float denom_term = eta * i_h + o_h;
        f3 one_f = mk(1.0f, 1.0f, 1.0f) - fr;
        val = one_f * d * g * __builtin_fabsf(i_h) * __builtin_fabsf(o_h) /
              (i_n * o_n * denom_term * denom_term * (1.0f - rr_f));
        float jac = pt_div(__builtin_fabsf(o_h), denom_term * denom_term);
        p = pdf_vndf * jac;
    }
    if (!finite3(val) || !__builtin_isfinite(p) || p <= 0.0f) return;
    wo = on; f = val; pdf = p; cos_out = o_n;
}
// OrenNayar::bsdf_pdf, material.rs:221-265.  The reference takes cos(phi_i - phi_o) of two atan2 azimuths
// (:246-249); the f32 specification uses the same number without trigonometry: the cosine of the angle between
// the tangent-plane projections, cos dphi = u_i . u_o with u = p/|p| (and u = (1, 0) for p = 0, as atan2(0, 0) = 0).
PT_DEV void unit_azimuth(float x, float y, float& ux, float& uy) {
    float l2 = __builtin_fmaf(y, y, x * x);
    if (l2 > 0.0f) { float inv = pt_rcp(pt_sqrt(l2)); ux = x * inv; uy = y * inv; }
    else { ux = 1.0f; uy = 0.0f; }
}
PT_DEV void oren_nayar_eval(const Mat& m, f3 dir_in, f3 o, f3 n, f3& f, float& pdf) {
    f3 i = -dir_in;
    float ci = __builtin_fmaxf(dot(i, n), 0.0f), co = __builtin_fmaxf(dot(o, n), 0.0f);
    float si = pt_sqrt(__builtin_fmaxf(1.0f - ci * ci, 0.0f));
    float so = pt_sqrt(__builtin_fmaxf(1.0f - co * co, 0.0f));
    f3 tangent, bitangent;
    frame_of(n, tangent, bitangent);
    float uix, uiy, uox, uoy;
    unit_azimuth(dot(i, tangent), dot(i, bitangent), uix, uiy);
    unit_azimuth(dot(o, tangent), dot(o, bitangent), uox, uoy);
    float cos_phi = __builtin_fmaxf(__builtin_fmaf(uiy, uoy, uix * uox), 0.0f);
    float sin_alpha, tan_beta;
    if (ci > co) { tan_beta = ci > 1e-6f ? pt_div(si, ci) : 0.0f; sin_alpha = so; }
    else { tan_beta = co > 1e-6f ? pt_div(so, co) : 0.0f; sin_alpha = si; }
    float term = m.on_a + m.on_b * cos_phi * sin_alpha * tan_beta;
    f = m.color * (term * kInvPi);
    pdf = __builtin_fmaxf(dot(o, n), 0.0f) * kInvPi;
}
// Object::bsdf_pdf (object.rs:35-43)
PT_DEV void bsdf_pdf(const Mat& m, f3 dir_in, float eta, f3 o, f3 n, f3& f, float& pdf) {
    if (m.tag == MAT_LAMBERT) {                                      // material.rs:86-91
        f = m.color * kInvPi;
        pdf = __builtin_fmaxf(dot(o, n), 0.0f) * kInvPi;
    } else if (m.tag == MAT_EMISSIVE) {                              // material.rs:139-148
        f = mk(0.0f, 0.0f, 0.0f); pdf = 1.0f;
    } else if (m.tag == MAT_MIRROR) {                                // mirror.rs:179-198
        f3 i = -dir_in;
        float i_n = dot(i, n), o_n = dot(o, n);
        bool is_refl = i_n * o_n > 0.0f;
        if (m.metallic > 0.99f && !is_refl) { f = mk(0.0f, 0.0f, 0.0f); pdf = 1.0f; }
        else mirror_eval(m, dir_in, eta, o, n, is_refl, f, pdf);
    } else {
        oren_nayar_eval(m, dir_in, o, n, f, pdf);
    }
}
// Object::bsdf_pdf_sample (object.rs:46-54); w_r1, w_r2 = the vertex's BSDF words of BLK_SURFACE, w_lobe = its
// lobe word of BLK_CHOICE (read by Mirror only)
PT_DEV void bsdf_pdf_sample(const Mat& m, f3 dir_in, float eta, f3 n, uint32_t w_r1, uint32_t w_r2, uint32_t w_lobe,
                            f3& wo, f3& f, float& pdf, float& cos_out) {
    float r1 = u01(w_r1), r2 = u01(w_r2);
    if (m.tag == MAT_MIRROR) {
        mirror_sample(m, dir_in, eta, n, r1, r2, u01(w_lobe), wo, f, pdf, cos_out);
        return;
    }
    wo = (m.tag == MAT_EMISSIVE) ? n : cosine_sample(n, r1, r2);     // material.rs:150-158 / :93-122
    bsdf_pdf(m, dir_in, eta, wo, n, f, pdf);                         // default impl material.rs:29-40
    cos_out = __builtin_fmaxf(dot(wo, n), 0.0f);
}

// Russian-roulette probability, rendering.rs:91-98
PT_DEV float rr_prob(uint32_t depth, uint32_t min_depth, uint32_t max_depth, f3 next_tp) {
    if (depth < min_depth) return 1.0f;
    float l = __builtin_fminf(luminance(next_tp), 1.0f);
    if (depth >= max_depth) return l * __builtin_ldexpf(1.0f, -(int)(depth - min_depth));
    return l;
}

}  // namespace PTD_NS
