// pt_oracle.hpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the hot path of roxas1533/pathtrace (reference at
// /root/reference, Rust, f64, recursive).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, load or call this; the product
// (pathtrace_amd/) never includes or links anything under oracle/.
//
// PARITY PINNING.  The Rust reference cannot be built here (no cargo/rustc, 244
// un-vendored crates, no network -- SURVEY 8c) and its tree holds no golden
// vector for the integrator.  What IS pinned:
//   * Vector3 arithmetic   <- the 18 unit tests of src/math.rs:246-418
//                             (tests/test_oracle_math.py restates every one);
//   * the generator        <- Philox4x32 (7 rounds for the render draws), Random123 known-answer vectors
//                             (tests/test_rng.py).
//   * everything else (intersection, sampling, BSDFs, integrator) is pinned
//     only by source-faithfulness of this restatement: "parity unpinned" in the
//     sense of the task statement.  Each function cites the reference lines it
//     follows so it can be checked by reading.
//
// RNG.  The reference draws from rand 0.9.2 StdRng (ChaCha12), one sequential
// stream per pixel seeded (y<<32)|x (src/main.rs:51-52).  That crate is not in
// the tree.  This build keeps the seeding convention and the draw ORDER but
// addresses every draw as philox(ctr=(x, y, sample, depth), key=(block, 0)) so
// that a wavefront may evaluate vertices in any order.
//
// A second draw source exists for Real=double, recursive form only: StdRngStream below restates the reference's own
// generator (ChaCha12 behind rand's StdRng, seed_from_u64, one sequential stream per pixel) from its published
// algorithm.  It is UNVERIFIED AGAINST THE rand CRATE (not in the tree, nothing fetched); its block function is pinned
// by the RFC 8439 and eSTREAM/Strombergson ChaCha vectors (tests/test_rng.py).  It exists to check the Philox
// ADDRESSING (shared surface block, roulette bits from the low 9) against something other than itself, and to make a
// number-for-number comparison possible the day a Rust-produced luminance.csv exists (examples/luminance_diff).
//
// Two instantiations:
//   Real=double : reference-faithful arithmetic (no FMA, true divisions, libm
//                 sin/cos), recursive integrator exactly as rendering.rs:34-142.
//   Real=float  : the device-equivalent arithmetic: dot/cross/at use FMA in a
//                 fixed order, vector/scalar is multiplication by the IEEE
//                 reciprocal, sin/cos of 2*pi*u is a fixed polynomial.  The HIP
//                 kernels implement the same operation sequence, so GPU vs
//                 float-oracle differences are bugs or compiler reassociation,
//                 not "FP32 tolerance"; the FP32 tolerance proper is float-oracle
//                 vs double-oracle.
// Compile with -ffp-contract=off (the Makefile does) so no other FMA appears.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/pathtrace_amd.h"

namespace orc {

// ------------------------------------------------------------------ Philox
// Philox4x32-R (Salmon et al., SC'11; Random123), R = `rounds`.  KATs in tests/test_rng.py: the three Random123 vectors
// at R = 10 and the zero-input one at R = 7.  The render draws use kDrawRounds = 7, the paper's Crush-resistant minimum
// (its Table 2), like the device (pt_device.h PT_PHILOX_ROUNDS).
constexpr int kDrawRounds = 7;
inline void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds = 10) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draw blocks per vertex (the dimensions of SURVEY Appendix A, regrouped by HOW OFTEN a vertex needs them):
//   block 0 (surface): [0] light r1 [1] light r2 [2] bsdf r1 [3] bsdf r2       -- every non-emitter vertex
//   block 1 (choice):  [0] light index (raw u32) [1] Mirror lobe u [2], [3] spare
// The Russian-roulette uniform of the vertex is made of the bits of block 0 that u01() does not look at (the low 9
// of every word; Philox output bits are independent): rr_word().
// Block 1 decides nothing at a vertex of a Lambertian / OrenNayar surface with one light in the scene
// (index = umulhi(u, 1) = 0): the device does not generate it there (one Philox call per vertex instead of two), the
// oracle always does.
// Camera jitter: depth = 0xFFFFFFFF, block 0: [0] ox [1] oy   (world.rs:299: ox first)
enum { BLK_SURFACE = 0, BLK_CHOICE = 1, DEPTH_CAMERA = 0xFFFFFFFFu };
enum { DIM_LIGHT_R1 = 0, DIM_LIGHT_R2 = 1, DIM_BSDF_R1 = 2, DIM_BSDF_R2 = 3 };       // words of BLK_SURFACE
enum { DIM_LIGHT_INDEX = 0, DIM_LOBE = 1 };                                         // words of BLK_CHOICE

// 23-bit uniform on the OPEN interval (0,1): (2k+1)/2^24, exactly representable
// in f32, so the float and double oracles and the device see the same value.
// The reference's f64 uniforms live on [0,1) (rand: (u64>>11)*2^-53); 0 has
// probability 2^-53 there and would give 0/0 in the cosine sampler
// (material.rs:104, rendering.rs:89), so excluding it changes nothing measurable
// and removes the f32 singularities (SURVEY 8a "unprotected singularities").
inline double u01(uint32_t r) { return (double)(((r >> 9) << 1) | 1u) * (1.0 / 16777216.0); }
// The roulette word of a vertex from its BLK_SURFACE words: u01() of it reads the low 9 bits of word 0, the low 9 of
// word 1 and bits 8..4 of word 2 -- none of which any other draw of the vertex looks at.
inline uint32_t rr_word(const uint32_t ds[4]) { return (ds[0] << 23) | ((ds[1] & 0x1FFu) << 14) | ((ds[2] & 0x1FFu) << 5); }

struct Draws {
    uint32_t key[2];   // (x, y): low/high word of the reference seed (y<<32)|x, main.rs:51 -- the first two counter words
    uint32_t sample;
    void block(uint32_t depth, uint32_t blk, uint32_t out[4]) const {
        uint32_t c[4] = {key[0], key[1], sample, depth};     // counter = (x, y, sample, depth)
        uint32_t k[2] = {blk, 0u};                           // key = (block, 0)
        philox4x32(c, k, out, kDrawRounds);
    }
};

// ------------------------------------------------------------------ the reference's generator, restated
// rand 0.9.2 `StdRng` = rand_chacha 0.9.0 `ChaCha12Rng` (Cargo.lock:1092-1118; crates absent from the tree).  Restated
// from the published algorithms; unverified against the crates themselves:
//   * ChaCha block function (Bernstein 2008; RFC 8439 2.3 for the 20-round form), `rounds` a parameter.  State =
//     "expand 32-byte k" | 8 key words | 64-bit block counter (words 12, 13) | 64-bit stream id = 0 (words 14, 15).
//   * rand_core `SeedableRng::seed_from_u64`: the 32-byte seed is 8 outputs of PCG32 (XSH-RR) run from the u64, each
//     state advance BEFORE its output (mul 6364136223846793005, inc 11634580027462260723), little-endian.
//   * rand_core `BlockRng`: a buffer of 64 words = 4 consecutive blocks; next_u32 = next word; next_u64 = two
//     consecutive words, low word first (straddling a refill when one word is left).
//   * `random::<f64>()` = (next_u64() >> 11) * 2^-53; `random_range(0..n)` for a usize range that fits u32 = one u32
//     draw, widening multiply by n, and Canon's single bias-reducing redraw when the low half exceeds -n mod 2^32.
inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
inline void chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[i];
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 16);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 12);
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 8);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);      // column round
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);      // diagonal round
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}
struct StdRngStream {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[64];
    uint32_t index = 64;          // 64 = buffer used up
    int rounds = 12;
    explicit StdRngStream(uint64_t seed, int rounds_ = 12) : rounds(rounds_) {    // seed_from_u64
        uint64_t state = seed;
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));    // rotate_right
        }
    }
    void refill() {
        for (int b = 0; b < 4; ++b) {
            const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                                     key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
            chacha_block(in, rounds, buf + 16 * b);
            ++counter;
        }
        index = 0;
    }
    uint32_t next_u32() {
        if (index >= 64) refill();
        return buf[index++];
    }
    uint64_t next_u64() {
        if (index < 63) { const uint64_t v = ((uint64_t)buf[index + 1] << 32) | buf[index]; index += 2; return v; }
        if (index >= 64) { refill(); index = 2; return ((uint64_t)buf[1] << 32) | buf[0]; }
        const uint64_t lo = buf[63];
        refill();
        index = 1;
        return ((uint64_t)buf[0] << 32) | lo;
    }
    double f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1), 53 bits
    uint32_t range(uint32_t n) {                                                        // random_range(0..n), n >= 1
        const uint64_t m = (uint64_t)next_u32() * n;
        uint32_t hi = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - n)) {
            const uint32_t new_hi = (uint32_t)(((uint64_t)next_u32() * n) >> 32);
            hi += (uint32_t)(((uint64_t)lo + new_hi) >> 32);                            // carry of lo + new_hi
        }
        return hi;
    }
};

// ------------------------------------------------------------------ draw sources of the integrators
// The integrators below pull their random numbers through one of these, in the reference's program order
// (SURVEY Appendix A): [light index] [light r1, r2] [bsdf r1, r2] [Mirror lobe u, only if i.h > 0] [roulette u].
//   PhiloxSampler: the build's addressing -- philox(ctr = (x, y, sample, depth), key = (block, 0)); a getter returns
//                  the same value however often and in whatever order it is called.
//   StreamSampler: the reference's -- every getter consumes the next draw(s) of the pixel's sequential StdRng stream.
struct PhiloxSampler {
    Draws dr;
    uint32_t ds[4] = {0, 0, 0, 0}, dc[4] = {0, 0, 0, 0};
    explicit PhiloxSampler(const Draws& d) : dr(d) {}
    void vertex(uint32_t depth) { dr.block(depth, BLK_SURFACE, ds); dr.block(depth, BLK_CHOICE, dc); }
    void camera(double& ox, double& oy) const { uint32_t c[4]; dr.block(DEPTH_CAMERA, 0, c); ox = u01(c[0]); oy = u01(c[1]); }
    uint32_t light_index(uint32_t n) const { return (uint32_t)(((uint64_t)dc[DIM_LIGHT_INDEX] * n) >> 32); }
    double light_r1() const { return u01(ds[DIM_LIGHT_R1]); }
    double light_r2() const { return u01(ds[DIM_LIGHT_R2]); }
    double bsdf_r1() const { return u01(ds[DIM_BSDF_R1]); }
    double bsdf_r2() const { return u01(ds[DIM_BSDF_R2]); }
    double lobe() const { return u01(dc[DIM_LOBE]); }
    double roulette() const { return u01(rr_word(ds)); }
};
struct StreamSampler {
    StdRngStream* rng;
    void vertex(uint32_t) {}
    void camera(double& ox, double& oy) { ox = rng->f64(); oy = rng->f64(); }      // world.rs:299: ox first
    uint32_t light_index(uint32_t n) { return rng->range(n); }                     // world.rs:255
    double light_r1() { return rng->f64(); }                                       // shape.rs:111 / :211
    double light_r2() { return rng->f64(); }                                       // shape.rs:112 / :212
    double bsdf_r1() { return rng->f64(); }                                        // material.rs:100, mirror.rs:42
    double bsdf_r2() { return rng->f64(); }                                        // material.rs:101, mirror.rs:43
    double lobe() { return rng->f64(); }                                           // mirror.rs:232
    double roulette() { return rng->f64(); }                                       // rendering.rs:100, :246
};
// the raw words of a debug / fixture entry as a draw source
struct WordSampler {
    uint32_t w_index, w_r1, w_r2, w_b1, w_b2, w_lobe;
    uint32_t light_index(uint32_t n) const { return (uint32_t)(((uint64_t)w_index * n) >> 32); }
    double light_r1() const { return u01(w_r1); }
    double light_r2() const { return u01(w_r2); }
    double bsdf_r1() const { return u01(w_b1); }
    double bsdf_r2() const { return u01(w_b2); }
    double lobe() const { return u01(w_lobe); }
};

// ------------------------------------------------------------------ arithmetic modes
template <class R> struct Ar;
template <> struct Ar<double> {
    static constexpr bool kFloat = false;
    static double dot3(double ax, double ay, double az, double bx, double by, double bz) {
        return ax * bx + ay * by + az * bz;   // math.rs:24-26
    }
    static double msub(double a, double b, double c, double d) { return a * b - c * d; }  // math.rs:31-33
    static double mad(double a, double b, double c) { return a * b + c; }
    static double rcp_div(double num, double den) { return num / den; }
    static void sincos2pi(double u, double& s, double& c) {
        double phi = 2.0 * 3.14159265358979323846 * u;   // "2.0 * PI * r" e.g. material.rs:103
        s = std::sin(phi); c = std::cos(phi);
    }
};
template <> struct Ar<float> {
    static constexpr bool kFloat = true;
    static float dot3(float ax, float ay, float az, float bx, float by, float bz) {
        return std::fmaf(az, bz, std::fmaf(ay, by, ax * bx));
    }
    static float msub(float a, float b, float c, float d) { return std::fmaf(a, b, -(c * d)); }
    static float mad(float a, float b, float c) { return std::fmaf(a, b, c); }
    // x / s is evaluated as x * (1/s) with an IEEE reciprocal (one division per
    // vector instead of three).
    static float rcp_div(float num, float den) { return num * (1.0f / den); }
    // sin/cos(2*pi*u), u in (0,1): quadrant k = rint(4u), r = u - k/4 in [-1/8,1/8],
    // t = 2*pi*r in [-pi/4,pi/4], cephes single-precision kernels, then rotate.
    static void sincos2pi(float u, float& s, float& c) {
        float k = std::rintf(u * 4.0f);
        float r = std::fmaf(k, -0.25f, u);
        float t = r * 6.28318530717958647692f;
        float z = t * t;
        float sp = std::fmaf(std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
        float st = std::fmaf(sp * z, t, t);
        float cp = std::fmaf(std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                             4.166664568298827e-2f);
        float ct = std::fmaf(cp * z, z, std::fmaf(-0.5f, z, 1.0f));
        int q = ((int)k) & 3;
        float ss = (q & 1) ? ct : st;
        float cc = (q & 1) ? st : ct;
        if (q == 1 || q == 2) cc = -cc;
        if (q == 2 || q == 3) ss = -ss;
        s = ss; c = cc;
    }
};

template <class R> constexpr R kPi() { return (R)3.14159265358979323846; }
// f32 mode evaluates x / pi as x * (1/pi) with this constant
constexpr float kInvPiF = 0.31830988618379067154f;
template <class R> inline R div_pi(R x) { return Ar<R>::kFloat ? (R)((float)x * kInvPiF) : x / kPi<R>(); }
template <class R> constexpr R kInf() { return std::numeric_limits<R>::infinity(); }

// ------------------------------------------------------------------ Vector3 (math.rs:3-244)
template <class R> struct V3 {
    R x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(R a, R b, R c) : x(a), y(b), z(c) {}
    static V3 zero() { return V3(0, 0, 0); }            // math.rs:15
    static V3 one() { return V3(1, 1, 1); }             // math.rs:19
    R dot(const V3& o) const { return Ar<R>::dot3(x, y, z, o.x, o.y, o.z); }          // math.rs:24
    V3 cross(const V3& o) const {                                                    // math.rs:29
        return V3(Ar<R>::msub(y, o.z, z, o.y), Ar<R>::msub(z, o.x, x, o.z), Ar<R>::msub(x, o.y, y, o.x));
    }
    R length_squared() const { return dot(*this); }      // math.rs:43
    R length() const { return std::sqrt(dot(*this)); }   // math.rs:38
    V3 normalize() const {                               // math.rs:48-51 (len==0 returns self)
        R len = length();
        return len > 0 ? (*this) / len : *this;
    }
    V3 normalized() const { return normalize(); }        // math.rs:54
    static V3 normal_from_triangle(const V3& v0, const V3& v1, const V3& v2) {   // math.rs:60-64
        return (v1 - v0).cross(v2 - v0).normalize();
    }
    V3 reflect(const V3& n) const { return *this - n * R(2) * dot(n); }           // math.rs:69-71
    bool refract(const V3& n, R eta, V3& out) const {                              // math.rs:77-88
        R cos_i = -dot(n);
        R sin2_t = eta * eta * (R(1) - cos_i * cos_i);
        if (sin2_t > R(1)) return false;
        R cos_t = std::sqrt(R(1) - sin2_t);
        out = *this * eta + n * (eta * cos_i - cos_t);
        return true;
    }
    V3 face_forward(const V3& dir) const { return dot(dir) < 0 ? *this : -(*this); }   // math.rs:92-98
    R max() const { return std::fmax(std::fmax(x, y), z); }                              // math.rs:128
    R luminance() const { return R(0.2126) * x + R(0.7152) * y + R(0.0722) * z; }        // math.rs:133
    bool is_zero() const { return x == 0 && y == 0 && z == 0; }

    V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }   // math.rs:139
    V3& operator+=(const V3& o) { x += o.x; y += o.y; z += o.z; return *this; } // math.rs:151
    V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }   // math.rs:160
    V3 operator*(R s) const { return V3(x * s, y * s, z * s); }                 // math.rs:173
    V3 operator*(const V3& o) const { return V3(x * o.x, y * o.y, z * o.z); }   // math.rs:195
    V3 operator/(R s) const {                                                   // math.rs:208
        if (Ar<R>::kFloat) { R inv = R(1) / s; return V3(x * inv, y * inv, z * inv); }
        return V3(x / s, y / s, z / s);
    }
    V3 div_vec(const V3& o) const { return V3(x / o.x, y / o.y, z / o.z); }     // math.rs:221
    V3 operator-() const { return V3(-x, -y, -z); }                             // math.rs:234
};
template <class R> inline V3<R> operator*(R s, const V3<R>& v) { return v * s; }   // math.rs:186

// a*s + b (used for ray.at and frame combinations)
template <class R> inline V3<R> madd(const V3<R>& a, R s, const V3<R>& b) {
    return V3<R>(Ar<R>::mad(a.x, s, b.x), Ar<R>::mad(a.y, s, b.y), Ar<R>::mad(a.z, s, b.z));
}
// t*x + b*y + n*z: "tangent * x + bitangent * y + *normal * z" (material.rs:121)
template <class R> inline V3<R> frame3(const V3<R>& t, R x, const V3<R>& b, R y, const V3<R>& n, R z) {
    if (Ar<R>::kFloat) return madd(n, z, madd(b, y, t * x));
    return t * x + b * y + n * z;
}

// ------------------------------------------------------------------ Ray / Camera (camera.rs)
template <class R> struct Ray {
    V3<R> origin, direction;
    R eta_ratio;
    Ray() : eta_ratio(1) {}
    Ray(const V3<R>& o, const V3<R>& d) : origin(o), direction(d.normalize()), eta_ratio(1) {}   // camera.rs:10-16
    // f32 mode: a direction that is already normalised is not normalised a second time (the reference's
    // Ray::new does, changing the last bit at most); f64 keeps the reference's double normalisation.
    static Ray from_unit(const V3<R>& o, const V3<R>& d) {
        if (!Ar<R>::kFloat) return Ray(o, d);
        Ray r; r.origin = o; r.direction = d; r.eta_ratio = 1; return r;
    }
    V3<R> at(R t) const {                                                                         // camera.rs:18-20
        if (Ar<R>::kFloat) return madd(direction, t, origin);
        return origin + direction * t;
    }
};

template <class R> struct Camera {
    V3<R> origin, lower_left, horizontal, vertical;
    uint32_t width, height;
    // camera.rs:139-147
    Ray<R> get_ray_with_offset(uint32_t x, uint32_t y, R ox, R oy) const {
        R u = ((R)x + ox) / (R)(width - 1);
        R v = ((R)y + oy) / (R)(height - 1);
        V3<R> dir = lower_left + horizontal * u + vertical * v - origin;
        return Ray<R>(origin, dir);
    }
};

// Camera::new (camera.rs:50-82) in f64 -> PtCamera
inline void camera_new(const double o[3], uint32_t w, uint32_t h, double dist, double fov_deg, PtCamera* out) {
    double fov = fov_deg * (3.14159265358979323846 / 180.0);       // to_radians
    double aspect = (double)w / (double)h;
    double vh = 2.0 * std::tan(fov / 2.0) * dist;
    double vw = vh * aspect;
    V3<double> origin(o[0], o[1], o[2]), hor(vw, 0, 0), ver(0, vh, 0);
    V3<double> llc = origin - hor / 2.0 - ver / 2.0 - V3<double>(0, 0, dist);
    double* f[4] = {out->origin, out->lower_left, out->horizontal, out->vertical};
    const V3<double> v[4] = {origin, llc, hor, ver};
    for (int i = 0; i < 4; ++i) { f[i][0] = v[i].x; f[i][1] = v[i].y; f[i][2] = v[i].z; }
    out->width = w; out->height = h;
}
// Camera::look_at (camera.rs:94-130)
inline void camera_look_at(const double o[3], const double tgt[3], const double upv[3], uint32_t wd, uint32_t h,
                           double fov_deg, PtCamera* out) {
    double fov = fov_deg * (3.14159265358979323846 / 180.0);
    double aspect = (double)wd / (double)h;
    V3<double> origin(o[0], o[1], o[2]), target(tgt[0], tgt[1], tgt[2]), up(upv[0], upv[1], upv[2]);
    V3<double> w = (origin - target).normalize();
    V3<double> u = up.cross(w).normalize();
    V3<double> v = w.cross(u);
    double dist = 1.0;
    double vh = 2.0 * std::tan(fov / 2.0) * dist;
    double vw = vh * aspect;
    V3<double> hor = u * vw, ver = v * vh;
    V3<double> llc = origin - hor / 2.0 - ver / 2.0 - w * dist;
    double* f[4] = {out->origin, out->lower_left, out->horizontal, out->vertical};
    const V3<double> vv[4] = {origin, llc, hor, ver};
    for (int i = 0; i < 4; ++i) { f[i][0] = vv[i].x; f[i][1] = vv[i].y; f[i][2] = vv[i].z; }
    out->width = wd; out->height = h;
}

// ------------------------------------------------------------------ HitRecord (base.rs:6-34)
template <class R> struct Hit {
    V3<R> point, normal;
    R t;
    bool front_face;
    Hit() : t(0), front_face(false) {}
    Hit(const V3<R>& p, const V3<R>& outward, R tt, const Ray<R>& ray) : point(p), t(tt) {   // base.rs:19-33
        front_face = ray.direction.dot(outward) < 0;
        normal = front_face ? outward : -outward;
    }
};

// ------------------------------------------------------------------ scene (object.rs, flattened)
template <class R> struct Obj {
    uint32_t shape_tag, mat_tag;
    // sphere
    V3<R> center; R radius;
    // triangle
    V3<R> v0, v1, v2;
    V3<R> tn, tn1, tn2;   // f32 specification of TriangleShape::hit: plane normal and barycentric gradients (build_scene)
    // material
    V3<R> color;          // albedo / emission / Mirror.color
    R roughness, metallic, ior;
    R on_a, on_b;         // OrenNayar A,B (material.rs:182-193)
    bool emits;           // emit().length() > 0 (world.rs:222, rendering.rs:43)
};

template <class R> struct Scene {
    std::vector<Obj<R>> objs;
    std::vector<uint32_t> lights;   // world.rs:214-225
};

template <class R> inline Scene<R> build_scene(const PtObject* po, uint32_t n) {
    Scene<R> s;
    s.objs.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        Obj<R>& o = s.objs[i];
        const PtObject& p = po[i];
        o.shape_tag = p.shape_tag; o.mat_tag = p.mat_tag;
        o.radius = 0; o.roughness = 0; o.metallic = 0; o.ior = 1; o.on_a = 1; o.on_b = 0;
        if (p.shape_tag == PT_SHAPE_SPHERE) {
            o.center = V3<R>((R)p.shape[0], (R)p.shape[1], (R)p.shape[2]);
            o.radius = (R)p.shape[3];
        } else {
            o.v0 = V3<R>((R)p.shape[0], (R)p.shape[1], (R)p.shape[2]);
            o.v1 = V3<R>((R)p.shape[3], (R)p.shape[4], (R)p.shape[5]);
            o.v2 = V3<R>((R)p.shape[6], (R)p.shape[7], (R)p.shape[8]);
            // constants of the f32 form of TriangleShape::hit (triangle_hit): n = e1 x e2, N1 = (e2 x n)/(n.n),
            // N2 = (n x e1)/(n.n), in f64 from the edges as R holds them, rounded to R -- what the device's upload does
            // (ptbvh::triangle_scan_record), written independently
            const V3<R> e1r = o.v1 - o.v0, e2r = o.v2 - o.v0;
            const V3<double> e1((double)e1r.x, (double)e1r.y, (double)e1r.z), e2((double)e2r.x, (double)e2r.y, (double)e2r.z);
            const V3<double> n(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
            const double nn = n.x * n.x + n.y * n.y + n.z * n.z;
            const V3<double> a(e2.y * n.z - e2.z * n.y, e2.z * n.x - e2.x * n.z, e2.x * n.y - e2.y * n.x);
            const V3<double> b(n.y * e1.z - n.z * e1.y, n.z * e1.x - n.x * e1.z, n.x * e1.y - n.y * e1.x);
            o.tn = V3<R>((R)n.x, (R)n.y, (R)n.z);
            o.tn1 = V3<R>((R)(a.x / nn), (R)(a.y / nn), (R)(a.z / nn));
            o.tn2 = V3<R>((R)(b.x / nn), (R)(b.y / nn), (R)(b.z / nn));
        }
        switch (p.mat_tag) {
            case PT_MAT_LAMBERT:
            case PT_MAT_EMISSIVE:
                o.color = V3<R>((R)p.mat[0], (R)p.mat[1], (R)p.mat[2]);
                break;
            case PT_MAT_MIRROR:
                o.roughness = (R)p.mat[0];
                o.color = V3<R>((R)p.mat[1], (R)p.mat[2], (R)p.mat[3]);
                o.metallic = (R)p.mat[4];
                o.ior = (R)p.mat[5];
                break;
            case PT_MAT_OREN_NAYAR: {
                o.color = V3<R>((R)p.mat[0], (R)p.mat[1], (R)p.mat[2]);
                o.roughness = (R)p.mat[3];
                R s2 = o.roughness * o.roughness;               // material.rs:183-186
                o.on_a = R(1) - R(0.5) * s2 / (s2 + R(0.33));
                o.on_b = R(0.45) * s2 / (s2 + R(0.09));
            } break;
        }
        o.emits = (p.mat_tag == PT_MAT_EMISSIVE) && o.color.length() > 0;   // material.rs:160-162, world.rs:222
        if (o.emits) s.lights.push_back(i);
    }
    return s;
}

template <class R> inline V3<R> emit(const Obj<R>& o) {          // material.rs:62-64,160-162
    return o.mat_tag == PT_MAT_EMISSIVE ? o.color : V3<R>::zero();
}
template <class R> inline R get_eta(const Obj<R>& o) {           // material.rs:50-52, mirror.rs:317-319
    return o.mat_tag == PT_MAT_MIRROR ? o.ior : R(1);
}

// ------------------------------------------------------------------ shapes (shape.rs)
// SphereShape::hit, shape.rs:53-89
template <class R> inline bool sphere_hit(const Obj<R>& s, const Ray<R>& ray, R t_min, R t_max, Hit<R>& out) {
    V3<R> oc = ray.origin - s.center;
    R a = ray.direction.dot(ray.direction);
    R half_b = oc.dot(ray.direction);
    R disc, root;
    if (Ar<R>::kFloat) {
        // f32 arithmetic specification (device-equivalent):
        //  * a = 1: every ray of the path is a unit vector (Ray::new normalises, camera.rs:10-16), so the
        //    multiplications by a and 1/a of the general quadratic are dropped (SURVEY 8a row a5 prices the
        //    test that way: "16 if a = 1 and r^2 cached");
        //  * half_b^2 - c cancels catastrophically for a small sphere far from the origin (abs error
        //    ~|oc|^2 * 2^-24 against r^2): algebraically identical, robust form (Haines et al., Ray Tracing
        //    Gems ch. 7) disc = r^2 - |oc - half_b d|^2.
        V3<R> l = madd(ray.direction, -half_b, oc);
        disc = s.radius * s.radius - l.dot(l);
        if (disc < 0) return false;
        R sqrtd = std::sqrt(disc);
        root = -half_b - sqrtd;
        if (root < t_min || t_max < root) {
            root = -half_b + sqrtd;
            if (root < t_min || t_max < root) return false;
        }
    } else {
        R c = oc.dot(oc) - s.radius * s.radius;
        disc = half_b * half_b - a * c;
        if (disc < 0) return false;
        R sqrtd = std::sqrt(disc);
        root = Ar<R>::rcp_div(-half_b - sqrtd, a);
        if (root < t_min || t_max < root) {
            root = Ar<R>::rcp_div(-half_b + sqrtd, a);
            if (root < t_min || t_max < root) return false;
        }
    }
    V3<R> point = ray.at(root);
    V3<R> outward = (point - s.center) / s.radius;
    out = Hit<R>(point, outward, root, ray);
    return true;
}

// TriangleShape::hit, shape.rs:161-198 (Moeller-Trumbore)
template <class R> inline bool triangle_hit(const Obj<R>& tr, const Ray<R>& ray, R t_min, R t_max, Hit<R>& out) {
    V3<R> e1 = tr.v1 - tr.v0, e2 = tr.v2 - tr.v0;
    if (Ar<R>::kFloat) {
        // f32 arithmetic specification (device-equivalent): the same u, v, t from per-triangle constants instead of two
        // cross products per ray -- a = e1.(d x e2) = -(d.n), t = f e2.(s x e1) = -(s.n)/(d.n), u = (s + t d).N1,
        // v = (s + t d).N2 -- with the reference's accept rules predicate for predicate (:169, :176, :183, :190)
        R det = ray.direction.dot(tr.tn);
        if (std::fabs(det) < R(1e-8)) return false;
        V3<R> s = ray.origin - tr.v0;
        R t = -s.dot(tr.tn) / det;
        if (t < t_min || t > t_max) return false;
        V3<R> p = madd(ray.direction, t, s);
        R u = p.dot(tr.tn1);
        if (!(u >= 0 && u <= R(1))) return false;
        R v = p.dot(tr.tn2);
        if (v < 0 || u + v > R(1)) return false;
        V3<R> point = ray.at(t);
        V3<R> outward = e1.cross(e2).normalize();
        out = Hit<R>(point, outward, t, ray);
        return true;
    }
    V3<R> h = ray.direction.cross(e2);
    R a = e1.dot(h);
    if (std::fabs(a) < R(1e-8)) return false;
    R f = R(1) / a;
    V3<R> s = ray.origin - tr.v0;
    R u = f * s.dot(h);
    if (!(u >= 0 && u <= R(1))) return false;        // !(0.0..=1.0).contains(&u): NaN rejected
    V3<R> q = s.cross(e1);
    R v = f * ray.direction.dot(q);
    if (v < 0 || u + v > R(1)) return false;
    R t = f * e2.dot(q);
    if (t < t_min || t > t_max) return false;
    V3<R> point = ray.at(t);
    V3<R> outward = e1.cross(e2).normalize();
    out = Hit<R>(point, outward, t, ray);
    return true;
}

template <class R> inline bool shape_hit(const Obj<R>& o, const Ray<R>& ray, R t_min, R t_max, Hit<R>& out) {
    return o.shape_tag == PT_SHAPE_SPHERE ? sphere_hit(o, ray, t_min, t_max, out)
                                          : triangle_hit(o, ray, t_min, t_max, out);
}

struct ShapeSample { };
// Shape::sample_surface_from_point (shape.rs:29-34).  target == nullptr: draw r1,r2.
// Returns (point, normal, pdf_omega, dir, dist) through refs.
template <class R>
inline void sphere_sample(const Obj<R>& s, const Hit<R>& from, const Hit<R>* target, R r1, R r2, V3<R>& point,
                          V3<R>& normal, R& pdf_omega, V3<R>& dir, R& dist) {       // shape.rs:91-145
    V3<R> to_center = s.center - from.point;
    R distance_sq = to_center.dot(to_center);
    R sin_theta_max_sq = (s.radius * s.radius) / distance_sq;
    R cos_theta_max = std::sqrt(std::fmax(R(1) - sin_theta_max_sq, R(0)));
    // f32 mode: 1 - cos_theta_max = 1 - sqrt(1 - s) loses all its digits for a small or distant
    // light (s = r^2/d^2 ~ 1e-5 against ulp(1) = 6e-8); s / (1 + sqrt(1 - s)) is the same number.
    R omc = Ar<R>::kFloat ? sin_theta_max_sq / (R(1) + cos_theta_max) : R(1) - cos_theta_max;
    if (Ar<R>::kFloat && sin_theta_max_sq > R(1)) omc = R(1);   // observer inside: cos_theta_max = 0
    R solid_angle = R(2) * kPi<R>() * omc;
    pdf_omega = R(1) / solid_angle;
    if (target) {
        point = target->point;
    } else {
        R cos_theta, sin_theta, sin_theta_sq_f32 = 0;
        if (Ar<R>::kFloat) {
            R x = r1 * omc;                               // 1 - cos_theta
            cos_theta = R(1) - x;
            sin_theta_sq_f32 = std::fmax(x * (R(2) - x), R(0));       // (1-c)(1+c)
            sin_theta = std::sqrt(sin_theta_sq_f32);
        } else {
            cos_theta = R(1) - r1 + r1 * cos_theta_max;
            sin_theta = std::sqrt(std::fmax(R(1) - cos_theta * cos_theta, R(0)));
        }
        R sphi, cphi;
        Ar<R>::sincos2pi(r2, sphi, cphi);
        V3<R> w = to_center.normalize();
        if (Ar<R>::kFloat) {
            // f32 specification: the frame as in frame_of; the direction is normalised once (Ray::new, shape.rs:128).
            // The sampled point is from + t * direction, so the light direction and distance rendering.rs:58-60 derives
            // from the point again ARE direction and t: they are returned as such (dir, dist; the f32 integrators use
            // them), and t is the near root of the cone ray in the cone's own coordinates,
            //     t = dc cos(theta) - sqrt(r^2 - dc^2 sin^2(theta)),  dc = |center - from|
            // -- the same real number as the reference's quadratic on world coordinates (shape.rs:130-137), which mixes
            // the un-normalised direction for a, half_b with the normalised one for ray.at.  Discriminant clamped at 0
            // (unguarded sqrt).
            V3<R> u, v;
            frame_of(w, u, v);
            V3<R> direction = frame3(u, sin_theta * cphi, v, sin_theta * sphi, w, cos_theta).normalize();
            R sin2_theta = sin_theta_sq_f32;
            R dc = std::sqrt(distance_sq);
            R disc = Ar<R>::mad(-distance_sq, sin2_theta, s.radius * s.radius);
            R t = Ar<R>::mad(dc, cos_theta, -std::sqrt(std::fmax(disc, R(0))));
            point = madd(direction, t, from.point);
            normal = (point - s.center).normalize();
            dir = direction;
            dist = t;
            // observer inside the sphere: the near root is negative, the point is behind the cone direction; the reference
            // returns (point - from).normalize() and its length (shape.rs:139-144) = -direction, |t|
            if (dist < R(0)) { dir = -dir; dist = -dist; }
            return;
        } else {
            V3<R> up = std::fabs(w.y) > R(0.999) ? V3<R>(1, 0, 0) : V3<R>(0, 1, 0);
            V3<R> u = up.cross(w).normalize();
            V3<R> v = w.cross(u);
            V3<R> direction = frame3(u, sin_theta * cphi, v, sin_theta * sphi, w, cos_theta);
            Ray<R> sample_ray(from.point, direction);
            V3<R> oc = sample_ray.origin - s.center;
            // NB the reference uses the UN-normalised `direction` for a, half_b but
            // sample_ray.at(t) uses the normalised one (shape.rs:130-137).
            R a = direction.dot(direction);
            R half_b = oc.dot(direction);
            R c = oc.dot(oc) - s.radius * s.radius;
            R disc = half_b * half_b - a * c;
            // DELIBERATE DEVIATION: the reference takes sqrt(disc)
            // unguarded (shape.rs:136); a cone-edge sample whose disc rounds below
            // zero would give a NaN light point.  Clamped at 0 here and on the device.
            R t = (-half_b - std::sqrt(std::fmax(disc, R(0)))) / a;
            point = sample_ray.at(t);
        }
    }
    normal = (point - s.center).normalize();
    V3<R> light_dir = point - from.point;
    dist = light_dir.length();
    dir = light_dir.normalize();
}

template <class R>
inline void triangle_sample(const Obj<R>& tr, const Hit<R>& from, const Hit<R>* target, R r1, R r2, V3<R>& point,
                            V3<R>& normal, R& pdf_omega, V3<R>& dir, R& dist) {     // shape.rs:200-242
    if (target) {
        point = target->point;
    } else {
        R sqrt_r1 = std::sqrt(r1);
        R u = R(1) - sqrt_r1;
        R v = r2 * sqrt_r1;
        if (Ar<R>::kFloat) point = madd(tr.v2 - tr.v0, v, madd(tr.v1 - tr.v0, u, tr.v0));
        else point = tr.v0 + (tr.v1 - tr.v0) * u + (tr.v2 - tr.v0) * v;
    }
    V3<R> e1 = tr.v1 - tr.v0, e2 = tr.v2 - tr.v0;
    V3<R> cr = e1.cross(e2);
    normal = cr.normalize();
    R area = cr.length() * R(0.5);
    V3<R> to_light = point - from.point;
    dist = to_light.length();
    dir = to_light / dist;
    R cos_light = std::fabs(normal.dot(-dir));
    R pdf_area = R(1) / area;
    pdf_omega = cos_light > R(1e-8) ? pdf_area * (dist * dist) / cos_light : R(1e-8);
    if (!(dist > 0)) dir = to_light;      // as returned to the integrator: Vector3::normalize keeps a zero vector (math.rs:48-51)
}

template <class R>
inline void shape_sample(const Obj<R>& o, const Hit<R>& from, const Hit<R>* target, R r1, R r2, V3<R>& point,
                         V3<R>& normal, R& pdf_omega, V3<R>& dir, R& dist) {
    if (o.shape_tag == PT_SHAPE_SPHERE) sphere_sample(o, from, target, r1, r2, point, normal, pdf_omega, dir, dist);
    else triangle_sample(o, from, target, r1, r2, point, normal, pdf_omega, dir, dist);
}

// ------------------------------------------------------------------ materials
// local frame used by every sampler: material.rs:112-119, mirror.rs:21-27
template <class R> inline void frame_of(const V3<R>& n, V3<R>& tangent, V3<R>& bitangent) {
    if (Ar<R>::kFloat) {
        // f32 specification: up x n written out for the two constant `up` vectors -- X x n = (0, -n.z, n.y),
        // Y x n = (n.z, 0, -n.x) -- instead of a general cross product with selected constants: the same values
        // (a zero component may differ in sign), seven operations fewer.
        const bool use_x = std::fabs(n.y) > R(0.999);
        V3<R> raw = use_x ? V3<R>(R(0), -n.z, n.y) : V3<R>(n.z, R(0), -n.x);
        R len2 = Ar<R>::mad(raw.z, raw.z, n.z * n.z);        // = raw . raw: the other component is +-n.z in both cases
        R len = std::sqrt(len2);
        tangent = len > 0 ? raw / len : raw;                 // math.rs:48-51
        bitangent = n.cross(tangent);
        return;
    }
    V3<R> up = std::fabs(n.y) > R(0.999) ? V3<R>(1, 0, 0) : V3<R>(0, 1, 0);
    tangent = up.cross(n).normalize();
    bitangent = n.cross(tangent);
}

// cosine-weighted direction: material.rs:93-122 (Lambert), :267-295 (OrenNayar)
template <class R> inline V3<R> cosine_sample(const V3<R>& n, R r1, R r2) {
    R sphi, cphi;
    Ar<R>::sincos2pi(r1, sphi, cphi);
    R cos_theta = std::sqrt(r2);
    R sin_theta = std::sqrt(R(1) - cos_theta * cos_theta);
    R x = sin_theta * cphi, y = sin_theta * sphi, z = cos_theta;
    V3<R> t, b;
    frame_of(n, t, b);
    return frame3(t, x, b, y, n, z).normalize();
}

template <class R> inline R powi5(R x) { R x2 = x * x; return x2 * x2 * x; }   // powi(5)

// ---- Mirror (mirror.rs)
template <class R> inline V3<R> mirror_f(const Obj<R>& m, R cos_theta) {      // mirror.rs:126-132
    R f0d = (R(1) - m.ior) / (R(1) + m.ior);
    f0d = f0d * f0d;
    V3<R> f0 = V3<R>(f0d, f0d, f0d) * (R(1) - m.metallic) + m.color * m.metallic;
    return f0 + (V3<R>(1, 1, 1) - f0) * powi5(R(1) - cos_theta);
}
template <class R> inline R mirror_g1(const Obj<R>& m, R cos_theta) {         // mirror.rs:136-149
    if (cos_theta <= 0) return 0;
    R alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    R cos2 = cos_theta * cos_theta;
    R term = alpha2 + (R(1) - alpha2) * cos2;
    return R(2) * cos_theta / (cos_theta + std::sqrt(term));
}
template <class R> inline R mirror_g(const Obj<R>& m, R ci, R co) {           // mirror.rs:153-175
    if (ci <= 0 || co <= 0) return 0;
    R alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    auto lambda = [&](R c) {
        R c2 = c * c;
        R num = std::sqrt(alpha2 + (R(1) - alpha2) * c2);
        return (num - c) / (R(2) * c);
    };
    return R(1) / (R(1) + lambda(ci) + lambda(co));
}
template <class R> inline R ggx_d(R alpha2, R n_h) {                          // mirror.rs:69-70,100-101,238-239
    R denom = (n_h * n_h) * (alpha2 - R(1)) + R(1);
    return alpha2 / (kPi<R>() * denom * denom);
}
// Mirror::brdf, mirror.rs:62-88
template <class R> inline void mirror_brdf(const Obj<R>& m, const Ray<R>& ray, const V3<R>& o, const V3<R>& n,
                                           V3<R>& f, R& pdf) {
    V3<R> i = -ray.direction;
    R alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    V3<R> h = (i + o).normalize();
    R n_h = n.dot(h);
    R d = ggx_d(alpha2, n_h);
    R i_n = std::fmax(n.dot(i), R(0));
    R o_n = std::fmax(n.dot(o), R(0));
    R g = mirror_g(m, i_n, o_n);
    R cos_theta = std::fmax(i.dot(h), R(0));
    V3<R> fr = mirror_f(m, cos_theta);
    R denom_brdf = R(4) * i_n * o_n;
    f = d * g * fr / denom_brdf;                 // "d * g * f / denom_brdf": ((d*g)*F)/denom
    R i_h = std::fabs(i.dot(h));
    pdf = d * std::fabs(n_h) / (R(4) * i_h);
}
// Mirror::btdf, mirror.rs:90-124
template <class R> inline void mirror_btdf(const Obj<R>& m, const Ray<R>& ray, const V3<R>& o, const V3<R>& n,
                                           V3<R>& f, R& pdf) {
    V3<R> i = -ray.direction;
    R eta = ray.eta_ratio;
    V3<R> h = -((i * eta + o).normalize());
    R alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    R n_h = n.dot(h);
    R d = ggx_d(alpha2, n_h);
    R i_n = std::fabs(n.dot(i));
    R o_n = std::fabs(n.dot(o));
    R g = mirror_g(m, i_n, o_n);
    R i_h = i.dot(h), o_h = o.dot(h);
    R cos_theta = std::fabs(i_h);
    R denom_term = eta * i_h + o_h;
    V3<R> fr = mirror_f(m, cos_theta);
    // (1-F) * d * g * |i.h| * |o.h| / (i_n * o_n * denom^2)
    f = (V3<R>::one() - fr) * d * g * std::fabs(i_h) * std::fabs(o_h) / (i_n * o_n * denom_term * denom_term);
    R jac = std::fabs(o_h) / (denom_term * denom_term);
    pdf = d * std::fabs(n_h) * jac;
}
// Mirror::sample_ggx_vndf, mirror.rs:17-60
template <class R> inline V3<R> mirror_vndf(const Obj<R>& m, const V3<R>& view, const V3<R>& n, R r1, R r2) {
    R alpha = m.roughness * m.roughness;
    V3<R> tangent, bitangent;
    frame_of(n, tangent, bitangent);
    V3<R> vl(view.dot(tangent), view.dot(bitangent), view.dot(n));
    V3<R> vh = V3<R>(alpha * vl.x, alpha * vl.y, vl.z).normalize();
    R lensq = vh.x * vh.x + vh.y * vh.y;
    V3<R> t1 = lensq > 0 ? V3<R>(-vh.y, vh.x, 0) * (R(1) / std::sqrt(lensq)) : V3<R>(1, 0, 0);
    V3<R> t2 = vh.cross(t1);
    R r = std::sqrt(r1);
    R sphi, cphi;
    Ar<R>::sincos2pi(r2, sphi, cphi);
    R p1 = r * cphi;
    R p2 = r * sphi;
    R s = R(0.5) * (R(1) + vh.z);
    p2 = (R(1) - s) * std::sqrt(R(1) - p1 * p1) + s * p2;
    R p3 = std::sqrt(std::fmax(R(1) - p1 * p1 - p2 * p2, R(0)));
    V3<R> nh = frame3(t1, p1, t2, p2, vh, p3);
    V3<R> ne = V3<R>(alpha * nh.x, alpha * nh.y, std::fmax(nh.z, R(0))).normalize();
    return frame3(tangent, ne.x, bitangent, ne.y, n, ne.z).normalize();
}
template <class R> inline bool finite3(const V3<R>& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }

// Mirror::bsdf_pdf_sample, mirror.rs:200-305.  draw_lobe() is called -- the draw consumed -- only when i.h > 0 (:232).
template <class R, class LobeFn>
inline void mirror_sample(const Obj<R>& m, const Ray<R>& ray, const V3<R>& n, R r1, R r2, LobeFn&& draw_lobe, V3<R>& wo,
                          V3<R>& f, R& pdf, R& cos_out) {
    V3<R> i = -ray.direction;
    R i_dot_n = i.dot(n);
    R eta = ray.eta_ratio;
    V3<R> h = mirror_vndf(m, i, n, r1, r2);
    R i_h = i.dot(h);
    auto fail = [&]() { wo = n; f = V3<R>::zero(); pdf = 1; cos_out = 0; };
    if (i_h <= 0) { fail(); return; }
    V3<R> fr = mirror_f(m, i_h);
    R sin2_i = R(1) - i_h * i_h;
    R cos2_t = R(1) - (eta * eta) * sin2_i;
    bool tir = cos2_t < 0;
    R rr_f = fr.x;
    if (tir || m.metallic > R(0.99)) { rr_f = 1; fr = V3<R>(1, 1, 1); }
    const R u_lobe = (R)draw_lobe();
    bool is_reflect = u_lobe < rr_f;
    R alpha = m.roughness * m.roughness, alpha2 = alpha * alpha;
    R n_h = n.dot(h);
    R d = ggx_d(alpha2, n_h);
    if (is_reflect) {
        V3<R> o = R(2) * i_h * h - i;            // "2.0 * i_h * h - i": ((2*i_h)*h) - i
        V3<R> on = o.normalize();
        R o_n = std::fmax(n.dot(on), R(0));
        R i_n = std::fmax(i_dot_n, R(0));
        R g = mirror_g(m, i_n, o_n);
        R denom_brdf = R(4) * i_n * o_n;
        V3<R> brdf = fr * d * g / (denom_brdf * rr_f);
        R g1v = mirror_g1(m, i_n);
        R pdf_vndf = g1v * d * std::fmax(i_h, R(0)) / i_n;
        R p = pdf_vndf / (R(4) * std::fabs(i_h));
        if (!finite3(brdf) || !std::isfinite(p) || p <= 0) { fail(); return; }
        wo = on; f = brdf; pdf = p; cos_out = o_n;
    } else {
        R cos_t = std::sqrt(cos2_t);
        V3<R> o = h * (eta * i_h - cos_t) - i * eta;
        V3<R> on = o.normalize();
        R o_h = on.dot(h);
        R o_n = std::fabs(n.dot(on));
        R i_n = std::fabs(i_dot_n);
        R denom_term = eta * i_h + o_h;
        R g = mirror_g(m, i_n, o_n);
        V3<R> one_f = V3<R>::one() - fr;
        V3<R> btdf = one_f * d * g * std::fabs(i_h) * std::fabs(o_h) /
                     (i_n * o_n * denom_term * denom_term * (R(1) - rr_f));
        R jac = std::fabs(o_h) / (denom_term * denom_term);
        R g1v = mirror_g1(m, i_n);
        R pdf_vndf = g1v * d * std::fmax(i_h, R(0)) / i_n;
        R p = pdf_vndf * jac;
        if (!finite3(btdf) || !std::isfinite(p) || p <= 0) { fail(); return; }
        wo = on; f = btdf; pdf = p; cos_out = o_n;
    }
}

// ---- OrenNayar eval, material.rs:221-265
// f32 specification of cos(phi_i - phi_o): cosine of the angle between the tangent-plane projections (no
// atan2 / cos, whose float results differ between libm and the device's libdevice); u = (1, 0) for a zero
// projection, as atan2(0, 0) = 0.
inline void unit_azimuth_f32(float x, float y, float& ux, float& uy) {
    float l2 = std::fmaf(y, y, x * x);
    if (l2 > 0.0f) { float inv = 1.0f / std::sqrt(l2); ux = x * inv; uy = y * inv; }
    else { ux = 1.0f; uy = 0.0f; }
}
template <class R>
inline void oren_nayar_eval(const Obj<R>& m, const Ray<R>& ray, const V3<R>& o, const V3<R>& n, V3<R>& f, R& pdf) {
    V3<R> i = -ray.direction;
    R ci = std::fmax(i.dot(n), R(0)), co = std::fmax(o.dot(n), R(0));
    R si = std::sqrt(std::fmax(R(1) - ci * ci, R(0)));
    R so = std::sqrt(std::fmax(R(1) - co * co, R(0)));
    V3<R> tangent, bitangent;
    frame_of(n, tangent, bitangent);     // compute_tangent :210-217, bitangent = n x t :203
    R cos_phi;
    if (Ar<R>::kFloat) {
        float uix, uiy, uox, uoy;
        unit_azimuth_f32((float)i.dot(tangent), (float)i.dot(bitangent), uix, uiy);
        unit_azimuth_f32((float)o.dot(tangent), (float)o.dot(bitangent), uox, uoy);
        cos_phi = (R)std::fmax(std::fmaf(uiy, uoy, uix * uox), 0.0f);
    } else {
        R phi_i = std::atan2(i.dot(bitangent), i.dot(tangent));   // material.rs:246-247
        R phi_o = std::atan2(o.dot(bitangent), o.dot(tangent));
        cos_phi = std::fmax(std::cos(phi_i - phi_o), R(0));       // :249
    }
    R sin_alpha, tan_beta;
    if (ci > co) { tan_beta = ci > R(1e-6) ? si / ci : R(0); sin_alpha = so; }
    else { tan_beta = co > R(1e-6) ? so / co : R(0); sin_alpha = si; }
    R term = m.on_a + m.on_b * cos_phi * sin_alpha * tan_beta;
    f = m.color * div_pi<R>(term);
    pdf = div_pi<R>(std::fmax(o.dot(n), R(0)));
}

// Object::bsdf_pdf (object.rs:35-43) -> Material::bsdf_pdf
template <class R>
inline void bsdf_pdf(const Obj<R>& ob, const Ray<R>& ray, const V3<R>& o, const V3<R>& n, V3<R>& f, R& pdf) {
    switch (ob.mat_tag) {
        case PT_MAT_LAMBERT:                                  // material.rs:86-91, 78-82
            f = Ar<R>::kFloat ? ob.color * (R)kInvPiF : ob.color / kPi<R>();
            pdf = div_pi<R>(std::fmax(o.dot(n), R(0)));
            return;
        case PT_MAT_EMISSIVE:                                 // material.rs:139-148
            f = V3<R>::zero(); pdf = 1;
            return;
        case PT_MAT_MIRROR: {                                 // mirror.rs:179-198
            V3<R> i = -ray.direction;
            R i_n = i.dot(n), o_n = o.dot(n);
            bool is_refl = i_n * o_n > 0;
            if (ob.metallic > R(0.99) && !is_refl) { f = V3<R>::zero(); pdf = 1; return; }
            if (is_refl) mirror_brdf(ob, ray, o, n, f, pdf); else mirror_btdf(ob, ray, o, n, f, pdf);
            return;
        }
        default:
            oren_nayar_eval(ob, ray, o, n, f, pdf);
            return;
    }
}

// Object::bsdf_pdf_sample (object.rs:46-54) -> Material::bsdf_pdf_sample
// (default impl material.rs:29-40; Mirror override mirror.rs:200-305).  Draws through `sm` in the reference's order:
// r1, r2 (cosine sampling material.rs:100-101 / :274-275, VNDF mirror.rs:42-43), then the Mirror lobe draw if it is
// reached; Emissive::sample_direction draws nothing (material.rs:150-158).
template <class R, class S>
inline void bsdf_pdf_sample_s(const Obj<R>& ob, const Ray<R>& ray, const V3<R>& n, S& sm, V3<R>& wo, V3<R>& f, R& pdf,
                              R& cos_out) {
    switch (ob.mat_tag) {
        case PT_MAT_MIRROR: {
            const R r1 = (R)sm.bsdf_r1();
            const R r2 = (R)sm.bsdf_r2();
            mirror_sample(ob, ray, n, r1, r2, [&]() { return sm.lobe(); }, wo, f, pdf, cos_out);
            return;
        }
        case PT_MAT_EMISSIVE:                                 // sample_direction = normal, material.rs:150-158
            wo = n;
            break;
        default: {                                            // Lambert / OrenNayar cosine sampling
            const R r1 = (R)sm.bsdf_r1();
            const R r2 = (R)sm.bsdf_r2();
            wo = cosine_sample(n, r1, r2);
        } break;
    }
    bsdf_pdf(ob, ray, wo, n, f, pdf);
    cos_out = std::fmax(wo.dot(n), R(0));
}
// the same from raw words: w_r1, w_r2 = the vertex's BSDF words of BLK_SURFACE, w_lobe = its lobe word of BLK_CHOICE
template <class R>
inline void bsdf_pdf_sample(const Obj<R>& ob, const Ray<R>& ray, const V3<R>& n, uint32_t w_r1, uint32_t w_r2,
                            uint32_t w_lobe, V3<R>& wo, V3<R>& f, R& pdf, R& cos_out) {
    WordSampler ws{0u, 0u, 0u, w_r1, w_r2, w_lobe};
    bsdf_pdf_sample_s(ob, ray, n, ws, wo, f, pdf, cos_out);
}

// ------------------------------------------------------------------ World (world.rs)
// World::hit_scene, world.rs:270-290.  Returns object index or -1.
template <class R> inline int hit_scene(const Scene<R>& w, const Ray<R>& ray, R t_min, R t_max, Hit<R>& out) {
    int hit_obj = -1;
    R closest = t_max;
    Hit<R> h;
    for (size_t i = 0; i < w.objs.size(); ++i) {
        if (shape_hit(w.objs[i], ray, t_min, closest, h)) {
            closest = h.t;
            out = h;
            hit_obj = (int)i;
        }
    }
    return hit_obj;
}

// world.rs:48-52; dir / dist: direction and distance from the shading point as the sampler itself formed them (the f32
// specification uses them instead of deriving them from the point a second time, rendering.rs:58-60)
template <class R> struct LightSample { V3<R> point, emission; R pdf; V3<R> dir; R dist; };

// World::sample_light_point, world.rs:251-267.  Draws through `sm`: the light index (random_range(0..n), :255), then
// the two surface draws of Shape::sample_surface_from_point (shape.rs:111-112, :211-212); nothing if there is no light.
// PhiloxSampler / WordSampler: random_range = rand's widening multiply without its rare bias-rejection redraw (a second
// draw would break (depth, dim) addressing); StreamSampler keeps the redraw.
template <class R, class S>
inline bool sample_light_point_s(const Scene<R>& w, const Hit<R>& hit, S& sm, LightSample<R>& ls, uint32_t* light_slot = nullptr) {
    if (w.lights.empty()) return false;
    uint32_t n = (uint32_t)w.lights.size();
    const uint32_t li = sm.light_index(n);
    if (light_slot) *light_slot = li;
    const Obj<R>& lo = w.objs[w.lights[li]];
    const R r1 = (R)sm.light_r1();
    const R r2 = (R)sm.light_r2();
    V3<R> normal; R pdf_shape;
    shape_sample<R>(lo, hit, nullptr, r1, r2, ls.point, normal, pdf_shape, ls.dir, ls.dist);
    ls.emission = emit(lo);
    ls.pdf = pdf_shape / (R)n;
    return true;
}
// the same from raw words: w_index = the vertex's light-index word (BLK_CHOICE), w_r1, w_r2 = its light words of BLK_SURFACE
template <class R>
inline bool sample_light_point(const Scene<R>& w, const Hit<R>& hit, uint32_t w_index, uint32_t w_r1, uint32_t w_r2,
                               LightSample<R>& ls) {
    WordSampler ws{w_index, w_r1, w_r2, 0u, 0u, 0u};
    return sample_light_point_s(w, hit, ws, ls);
}

// ------------------------------------------------------------------ integrators (rendering.rs)
struct Params {
    uint32_t min_depth = 4, max_depth = 50;    // rendering.rs:6-7
    uint32_t integrator = PT_INTEGRATOR_MIS;
    double t_min = 0.001;
};
struct Counters { uint64_t vertices = 0, shadow_rays = 0, scans = 0; uint32_t max_depth = 0; };

template <class R> inline R rr_prob(const Params& p, uint32_t depth, const V3<R>& next_tp) {   // rendering.rs:91-98
    if (depth < p.min_depth) return R(1);
    R l = std::fmin(next_tp.luminance(), R(1));
    if (depth >= p.max_depth) return l * std::ldexp(R(1), -(int)(depth - p.min_depth));   // 0.5^(depth-MIN_DEPTH)
    return l;
}
template <class R> inline R eta_from_object(const Obj<R>& o, const Hit<R>& h) {   // rendering.rs:20-25
    return h.front_face ? R(1) / get_eta(o) : get_eta(o);
}

// MisStrategy::ray_color, rendering.rs:34-142 -- RECURSIVE, line for line.  S = the draw source (PhiloxSampler:
// addressed by depth; StreamSampler: the reference's sequential stream, consumed in this function's program order).
template <class R, class S>
V3<R> ray_color_mis_rec(const Scene<R>& w, const Params& p, Ray<R>& ray, uint32_t depth, S& sm,
                        V3<R> throughput, Counters& cn) {
    const R tmin = (R)p.t_min;
    Hit<R> hit;
    int oi = hit_scene(w, ray, tmin, kInf<R>(), hit);                              // :41
    cn.scans++;
    cn.vertices++;   // unit of work = one iteration of the per-vertex loop, misses included
    if (depth > cn.max_depth) cn.max_depth = depth;
    if (oi < 0) return V3<R>::zero();                                              // :141
    const Obj<R>& obj = w.objs[oi];
    V3<R> emitted = emit(obj);                                                     // :42
    if (emitted.length() > 0) return depth == 0 ? emitted : V3<R>::zero();         // :43-49
    V3<R> total = V3<R>::zero(), direct = V3<R>::zero();
    sm.vertex(depth);
    LightSample<R> ls;
    if (sample_light_point_s(w, hit, sm, ls)) {                                    // :56
        V3<R> to_light = ls.point - hit.point;                                     // :58
        R distance = to_light.length();
        V3<R> light_dir = to_light.normalize();
        if (Ar<R>::kFloat) { distance = ls.dist; light_dir = ls.dir; }             // f32 specification: from the sampler
        Ray<R> shadow(hit.point, light_dir);                                       // :62
        Hit<R> sh;
        cn.shadow_rays++; cn.scans++;
        bool visible = hit_scene(w, shadow, tmin, distance - tmin, sh) < 0;        // :63-65 (0.001 both)
        if (visible) {
            R cos_theta = std::fabs(hit.normal.dot(light_dir));                    // :68
            V3<R> bsdf; R pdf_bsdf;
            bsdf_pdf(obj, ray, light_dir, hit.normal, bsdf, pdf_bsdf);             // :71-72 (stale eta, Q5)
            R w_nee = ls.pdf / (ls.pdf + pdf_bsdf);                                // :73
            direct += w_nee * bsdf * ls.emission * cos_theta / ls.pdf;             // :75-76
        }
    }
    total += direct / R(1);                                                        // :81 (NUM_LIGHT_SAMPLES=1)
    ray.eta_ratio = eta_from_object(obj, hit);                                     // :83
    V3<R> wo, bsdf; R pdf, cos_theta;
    bsdf_pdf_sample_s(obj, ray, hit.normal, sm, wo, bsdf, pdf, cos_theta);         // :84-85
    Ray<R> scattered(hit.point, wo);                                               // :86
    scattered.eta_ratio = eta_from_object(obj, hit);                               // :87
    V3<R> next_tp = throughput * bsdf * cos_theta / pdf;                           // :89
    R rr = rr_prob(p, depth, next_tp);                                             // :91-98
    if ((R)sm.roulette() > rr) return V3<R>::zero();                               // :100-102: always drawn; drops `total` (Q1)
    Hit<R> h2;
    int o2 = hit_scene(w, scattered, tmin, kInf<R>(), h2);                         // :104-105
    cn.scans++;
    // The look-ahead scan is the next iteration of the iterative form whenever the
    // recursion is not entered (miss or emitter): count it so both forms agree.
    if (o2 < 0 || w.objs[o2].emits) {
        cn.vertices++;
        if (depth + 1 > cn.max_depth) cn.max_depth = depth + 1;
    }
    if (o2 >= 0) {
        const Obj<R>& ob2 = w.objs[o2];
        if (emit(ob2).length() > 0) {                                              // :107-112
            V3<R> sp, sn, sd; R pdf_shape, dd;
            shape_sample<R>(ob2, hit, &h2, R(0), R(0), sp, sn, pdf_shape, sd, dd);  // :114-116 (no draws)
            R w_bsdf = pdf / (pdf + pdf_shape);                                    // :117 (Q2)
            V3<R> le = emit(ob2);
            total += w_bsdf * bsdf * le * cos_theta / (pdf * rr);                  // :119-121
        } else {
            V3<R> li = ray_color_mis_rec(w, p, scattered, depth + 1, sm, next_tp / rr, cn);   // :124-130
            total += bsdf * li * cos_theta / (pdf * rr);                           // :131-133
        }
    }
    return total;                                                                  // :137
}

// BrdfOnlyStrategy::ray_color, rendering.rs:214-265 -- recursive.
template <class R, class S>
V3<R> ray_color_brdf_rec(const Scene<R>& w, const Params& p, Ray<R>& ray, uint32_t depth, S& sm,
                         V3<R> throughput, Counters& cn) {
    const R tmin = (R)p.t_min;
    Hit<R> hit;
    int oi = hit_scene(w, ray, tmin, kInf<R>(), hit);                              // :221
    cn.scans++;
    cn.vertices++;
    if (depth > cn.max_depth) cn.max_depth = depth;
    if (oi < 0) return V3<R>::zero();
    const Obj<R>& obj = w.objs[oi];
    V3<R> emitted = emit(obj);
    if (emitted.length() > 0) return emitted;                                      // :225-227
    sm.vertex(depth);
    ray.eta_ratio = eta_from_object(obj, hit);                                     // :230
    V3<R> wo, bsdf; R pdf, cos_theta;
    bsdf_pdf_sample_s(obj, ray, hit.normal, sm, wo, bsdf, pdf, cos_theta);         // :231-232
    Ray<R> scattered(hit.point, wo);
    scattered.eta_ratio = eta_from_object(obj, hit);
    V3<R> next_tp = throughput * bsdf * cos_theta / pdf;                           // :236
    R rr = rr_prob(p, depth, next_tp);
    if ((R)sm.roulette() > rr) return V3<R>::zero();                               // :246-248
    V3<R> li = ray_color_brdf_rec(w, p, scattered, depth + 1, sm, next_tp / rr, cn);
    return bsdf * li * cos_theta / (pdf * rr);                                     // :260
}

// ITERATIVE (wavefront-order) form of both integrators, SURVEY 3.5.  One loop
// iteration == one "vertex" == what one thread of the device bounce kernel does:
//   closest hit of the current ray; miss -> stop; emitter -> credit, stop;
//   NEE; BSDF sample; Russian roulette; L += beta*D; beta' ; next ray.
// Decisions (hits, RR) are computed from the same quantities as in the recursive
// form, so both forms visit identical vertices; only the order in which the
// radiance terms are summed differs (sum(beta_k*D_k) vs nested Horner form).
// A path whose next throughput is exactly (0,0,0) is retired.
// Optional per-vertex trace of the iterative form (debugging aid for parity work):
// 24 doubles per vertex, see orc_trace_path in oracle_capi.cpp.
struct Trace { std::vector<double> rec; };

template <class R>
V3<R> ray_color_iter(const Scene<R>& w, const Params& p, Ray<R> ray, const Draws& dr, Counters& cn,
                     Trace* tr = nullptr) {
    PhiloxSampler sm(dr);     // this form retires zero-throughput paths (Q7): legal only with addressed draws
    const R tmin = (R)p.t_min;
    const bool mis = p.integrator == PT_INTEGRATOR_MIS;
    V3<R> L = V3<R>::zero(), beta = V3<R>::one();
    R pdf_prev = 0;
    for (uint32_t depth = 0;; ++depth) {
        Hit<R> hit;
        int oi = hit_scene(w, ray, tmin, kInf<R>(), hit);
        cn.scans++;
        cn.vertices++;
        if (depth > cn.max_depth) cn.max_depth = depth;
        double* T = nullptr;
        if (tr) {
            tr->rec.resize(tr->rec.size() + 24, 0.0);
            T = &tr->rec[tr->rec.size() - 24];
            T[0] = depth; T[1] = oi; T[2] = oi >= 0 ? (double)hit.t : 0.0;
            T[3] = hit.point.x; T[4] = hit.point.y; T[5] = hit.point.z;
            T[6] = hit.normal.x; T[7] = hit.normal.y; T[8] = hit.normal.z;
            T[9] = beta.x; T[10] = beta.y; T[11] = beta.z;
        }
        if (oi < 0) break;
        const Obj<R>& obj = w.objs[oi];
        if (obj.emits) {
            V3<R> le = emit(obj);
            if (!mis || depth == 0) {
                L += beta * le;                          // rendering.rs:44-45 (beta==1), :225-227
            } else {
                // rendering.rs:107-121: from = previous vertex = this ray's origin
                Hit<R> from; from.point = ray.origin;
                V3<R> sp, sn, sd; R pdf_shape, dd;
                shape_sample<R>(obj, from, &hit, R(0), R(0), sp, sn, pdf_shape, sd, dd);
                R w_bsdf = pdf_prev / (pdf_prev + pdf_shape);
                L += beta * le * w_bsdf;
                if (T) { T[12] = pdf_prev; T[13] = pdf_shape; T[14] = w_bsdf; }
            }
            if (T) { T[21] = L.x; T[22] = L.y; T[23] = L.z; }
            break;
        }
        sm.vertex(depth);
        V3<R> direct = V3<R>::zero();
        if (mis) {
            LightSample<R> ls;
            if (sample_light_point_s(w, hit, sm, ls)) {
                V3<R> to_light = ls.point - hit.point;
                R distance = to_light.length();
                V3<R> light_dir = to_light.normalize();
                if (Ar<R>::kFloat) { distance = ls.dist; light_dir = ls.dir; }     // f32 specification: from the sampler
                Ray<R> shadow = Ray<R>::from_unit(hit.point, light_dir);
                Hit<R> sh;
                cn.shadow_rays++; cn.scans++;
                bool visible = hit_scene(w, shadow, tmin, distance - tmin, sh) < 0;
                if (visible) {
                    R cos_theta = std::fabs(hit.normal.dot(light_dir));
                    V3<R> bsdf; R pdf_bsdf;
                    bsdf_pdf(obj, ray, light_dir, hit.normal, bsdf, pdf_bsdf);
                    R w_nee = ls.pdf / (ls.pdf + pdf_bsdf);
                    direct = w_nee * bsdf * ls.emission * cos_theta / ls.pdf;
                    if (T) { T[14] = w_nee; T[15] = pdf_bsdf; }
                }
                if (T) { T[12] = ls.pdf; T[13] = visible ? 1.0 : 0.0; T[16] = distance; T[17] = direct.x; }
            }
        }
        R eta_here = eta_from_object(obj, hit);
        ray.eta_ratio = eta_here;
        V3<R> wo, bsdf; R pdf, cos_theta;
        bsdf_pdf_sample_s(obj, ray, hit.normal, sm, wo, bsdf, pdf, cos_theta);
        V3<R> next_tp = beta * bsdf * cos_theta / pdf;
        R rr = rr_prob(p, depth, next_tp);
        if (T) { T[18] = pdf; T[19] = rr; T[20] = sm.roulette(); }
        if ((R)sm.roulette() > rr) break;                 // drops `direct` too (Q1)
        L += beta * direct;
        if (T) { T[21] = L.x; T[22] = L.y; T[23] = L.z; }
        beta = next_tp / rr;
        if (beta.is_zero()) break;                       // Q7: nothing downstream can contribute
        if (depth >= 65534u) break;                      // device depth counter is 16 bits
        pdf_prev = pdf;
        // every sampler but Emissive's (wo = normal, material.rs:157) returns a normalised direction
        ray = obj.mat_tag == PT_MAT_EMISSIVE ? Ray<R>(hit.point, wo) : Ray<R>::from_unit(hit.point, wo);
        ray.eta_ratio = eta_here;
    }
    return L;
}

// ------------------------------------------------------------------ render_pixel + render()
enum Form { FORM_RECURSIVE = 0, FORM_ITERATIVE = 1 };

// Rows of the tile (band_rows, band_index, band_count), ascending.
inline std::vector<uint32_t> tile_rows(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count) {
    std::vector<uint32_t> rows;
    if (band_rows == 0) band_rows = height;
    if (band_count == 0) band_count = 1;
    for (uint32_t y = 0; y < height; ++y)
        if ((y / band_rows) % band_count == band_index) rows.push_back(y);
    return rows;
}

// World::render_pixel, world.rs:293-333.  The per-sample radiance is computed in
// R; the film sum, mean, gamma and quantisation are f64 in every instantiation
// (as in the reference, and as the device's resolve kernel does).
inline void film_pixel(const double acc[3], uint32_t spp, double out_lin[3], uint8_t out_rgba[4]) {
    for (int k = 0; k < 3; ++k) {
        double m = acc[k] / (double)spp;                                               // world.rs:315
        out_lin[k] = m;                                                                // world.rs:318-319
        double g = std::sqrt(m);                                                       // world.rs:322-324
        double cl = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);                               // clamp; NaN stays NaN
        double q = cl * 255.0;
        out_rgba[k] = (q != q) ? 0 : (uint8_t)q;                                       // `as u8`: trunc, NaN -> 0
    }
    out_rgba[3] = 255;
}
template <class R>
inline void render_pixel(const Scene<R>& w, const Camera<R>& cam, const Params& p, Form form, uint32_t x, uint32_t y,
                         uint32_t spp, uint32_t spp_offset, double out_lin[3], uint8_t out_rgba[4], Counters& cn) {
    double acc[3] = {0, 0, 0};
    for (uint32_t s = 0; s < spp; ++s) {                                               // world.rs:296
        Draws dr{{x, y}, spp_offset + s};                                              // main.rs:51
        PhiloxSampler sm(dr);
        double ox, oy;
        sm.camera(ox, oy);
        Ray<R> ray = cam.get_ray_with_offset(x, cam.height - 1 - y, (R)ox, (R)oy);     // world.rs:297-299
        V3<R> c;
        if (form == FORM_ITERATIVE) c = ray_color_iter(w, p, ray, dr, cn);
        else if (p.integrator == PT_INTEGRATOR_MIS) c = ray_color_mis_rec(w, p, ray, 0, sm, V3<R>::one(), cn);
        else c = ray_color_brdf_rec(w, p, ray, 0, sm, V3<R>::one(), cn);
        acc[0] += (double)c.x; acc[1] += (double)c.y; acc[2] += (double)c.z;           // world.rs:311
    }
    film_pixel(acc, spp, out_lin, out_rgba);
}
// The same with the reference's own draw source: ONE StdRng per pixel, seeded (y << 32) | x (main.rs:51-52), consumed
// sequentially by all samples of the pixel (world.rs:296-312).  f64, recursive form only (the iterative form retires
// zero-throughput paths early, which a sequential stream does not allow).  A sequential stream cannot be
// entered in the middle: spp_offset samples are traced and dropped first (the skip-ahead of world.rs:634-652).
// out_samples: optional, spp * 3 per-sample radiances.
inline void render_pixel_stdrng(const Scene<double>& w, const Camera<double>& cam, const Params& p, uint32_t x, uint32_t y,
                                uint32_t spp, uint32_t spp_offset, double out_lin[3], uint8_t out_rgba[4], Counters& cn,
                                double* out_samples = nullptr) {
    StdRngStream rng(((uint64_t)y << 32) | (uint64_t)x);                               // main.rs:51-52
    StreamSampler sm{&rng};
    double acc[3] = {0, 0, 0};
    Counters skipped;
    for (uint32_t s = 0; s < spp_offset + spp; ++s) {                                  // world.rs:296
        double ox, oy;
        sm.camera(ox, oy);                                                             // world.rs:299 (ox first)
        Ray<double> ray = cam.get_ray_with_offset(x, cam.height - 1 - y, ox, oy);
        Counters& c = s < spp_offset ? skipped : cn;
        V3<double> col = p.integrator == PT_INTEGRATOR_MIS ? ray_color_mis_rec(w, p, ray, 0, sm, V3<double>::one(), c)
                                                           : ray_color_brdf_rec(w, p, ray, 0, sm, V3<double>::one(), c);
        if (s < spp_offset) continue;
        if (out_samples) { double* o = out_samples + 3 * (size_t)(s - spp_offset); o[0] = col.x; o[1] = col.y; o[2] = col.z; }
        acc[0] += col.x; acc[1] += col.y; acc[2] += col.z;                             // world.rs:311
    }
    film_pixel(acc, spp, out_lin, out_rgba);
}

}  // namespace orc
