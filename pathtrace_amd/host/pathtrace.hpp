// pathtrace.hpp -- C++ host mirror of the reference's scene-authoring surface,
// above the C ABI (include/pathtrace_amd.h).
//
// The reference is Rust; no Rust toolchain exists in this image, so the host side
// above the boundary is C++ with the same names, argument meaning and defaults:
//   Vector3                                   src/math.rs:3-8
//   Camera::new / Camera::look_at             src/camera.rs:50-82, 94-130
//   SphereShape::new / TriangleShape::new     src/objects/shape.rs:45-50, 154-158
//   LambertianCosineWeighted::new, Emissive::new, OrenNayar::new, Mirror{..}
//                                             src/objects/material.rs:73, 133, 182; mirror.rs:5-14
//   Object::new(shape, material)              src/objects/object.rs:22-24
//   World::new()  (the Cornell box)           src/world.rs:65-241
//   World::render()  == the closure of main() src/main.rs:43-60  (new name)
//   World::draw(frame)                        src/world.rs:335-341
//   World::export_luminance(path)             src/world.rs:344-369
//   WIDTH, HEIGHT, SAMPLE_NUM                 src/world.rs:16-18
//   Ray::new / at / set_eta_ratio             src/camera.rs:9-25
//   Camera::get_ray_with_offset               src/camera.rs:139-147 (host f64: four multiply-adds, no hot loop)
//   HitRecord                                 src/objects/base.rs:6-33
//   Shape::{hit, sample_surface_from_point}   src/objects/shape.rs:8-35
//   Material::{bsdf_pdf, bsdf_pdf_sample, emit, get_eta}   src/objects/material.rs:5-65
//   Object::{hit, bsdf_pdf, bsdf_pdf_sample}  src/objects/object.rs:27-54
//   World::{hit_scene, sample_light_point, render_pixel}   src/world.rs:251-333
//   MisStrategy / BrdfOnlyStrategy::ray_color src/rendering.rs:34-142, 214-265
//   World::import_luminance + compare_luminance            reader / differ of the format of world.rs:344-369
// Shapes and materials are value types here (the device needs POD, not Box<dyn>);
// Object::new flattens them to the PtObject the ABI takes, preserving World.objects
// order.  The ARITHMETIC of the hot path (hit, bsdf_pdf, sample_surface_from_point, ray_color, ...) is not
// restated on the host: every one of those methods is a call into the GPU library (the pt_debug_* function
// entries, pt_render_pixels, pt_ray_color), i.e. the same device functions the render kernels inline.  This
// header has no CPU rendering path -- every method fails (PtError) if the GPU library reports an error.
// Where the reference threads `rng: &mut impl Rng` through a call, the mirror takes what the counter-based
// generator of the device needs instead: the raw words / uniforms of the draw, or the stream key (x, y, sample).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pathtrace_amd.h"

namespace pathtrace {

constexpr uint32_t WIDTH = 400;        // world.rs:16
constexpr uint32_t HEIGHT = 400;       // world.rs:17
constexpr uint32_t SAMPLE_NUM = 3000;  // world.rs:18

struct Vector3 {
    double x, y, z;
    Vector3() : x(0), y(0), z(0) {}
    Vector3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vector3 zero() { return Vector3(0, 0, 0); }
    static Vector3 one() { return Vector3(1, 1, 1); }
    double length() const { return std::sqrt(x * x + y * y + z * z); }                       // math.rs:38
    Vector3 normalize() const { const double l = length(); return l == 0.0 ? *this : Vector3(x / l, y / l, z / l); }   // math.rs:48-51
    double luminance() const { return 0.2126 * x + 0.7152 * y + 0.0722 * z; }                // math.rs:133-135
    Vector3 operator+(const Vector3& o) const { return Vector3(x + o.x, y + o.y, z + o.z); }
    Vector3 operator-(const Vector3& o) const { return Vector3(x - o.x, y - o.y, z - o.z); }
    Vector3 operator*(double s) const { return Vector3(x * s, y * s, z * s); }
};

struct Color { uint8_t r, g, b, a; };   // world.rs:20-26

class PtError : public std::runtime_error {
public:
    PtError(int code, const char* msg) : std::runtime_error(std::string("pathtrace_amd error ") + std::to_string(code) + ": " + msg), code_(code) {}
    int code() const { return code_; }
private:
    int code_;
};
inline void check(int rc) { if (rc != PT_OK) throw PtError(rc, pt_last_error()); }
// A binary built against another ABI version of the header must not call into the library (struct sizes differ: PtStats grew in
// version 4): every context of this mirror is created through here.
inline void check_abi() {
    if (pt_abi_version() != PT_ABI_VERSION)
        throw std::runtime_error("pathtrace_amd: this program was built against ABI version " + std::to_string(PT_ABI_VERSION) +
                                 " of include/pathtrace_amd.h, the library is version " + std::to_string(pt_abi_version()) + ": rebuild");
}

// ---- camera.rs:3-25
struct Ray {
    Vector3 origin, direction;
    double eta_ratio = 1.0;
    static Ray new_(Vector3 origin, Vector3 direction) { Ray r; r.origin = origin; r.direction = direction.normalize(); r.eta_ratio = 1.0; return r; }
    Vector3 at(double t) const { return origin + direction * t; }
    void set_eta_ratio(double e) { eta_ratio = e; }
};
// ---- base.rs:6-15 (filled by the device: HitRecord::new's face-forwarding runs there)
struct HitRecord {
    Vector3 point, normal;
    double t = 0.0;
    bool front_face = false;
};
// ---- world.rs:48-52
struct LightSample {
    Vector3 point, emission;
    double pdf = 0.0;
    uint32_t light_object = 0;       // index into World.objects (not in the reference's struct; free on the device)
};

namespace detail {
// One lazily created GPU context per thread for the per-object probes (Shape::hit, Material::bsdf_pdf, ...): a scene
// of one object is uploaded and the device function is run on it.
struct Probe {
    PtContext* ctx = nullptr;
    ~Probe() { if (ctx) pt_context_destroy(ctx); }
    PtContext* get(const PtObject& one) {
        check_abi();
        if (!ctx && pt_context_create(0, &ctx) != PT_OK) throw std::runtime_error(std::string("pathtrace_amd: ") + pt_last_error());
        if (pt_scene_upload(ctx, &one, 1) != PT_OK) throw std::runtime_error(std::string("pathtrace_amd: ") + pt_last_error());
        return ctx;
    }
};
inline Probe& probe() { static thread_local Probe p; return p; }
inline void put3(double* dst, const Vector3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
inline Vector3 get3(const float* p) { return Vector3(p[0], p[1], p[2]); }
inline HitRecord record(const float* rec) {
    HitRecord h; h.t = rec[0]; h.point = get3(rec + 1); h.normal = get3(rec + 4); h.front_face = rec[7] != 0.0f; return h;
}
}  // namespace detail

// ---- camera.rs
class Camera {
public:
    static Camera new_(Vector3 origin, uint32_t width, uint32_t height, double screen_distance, double fov_degrees) {
        Camera c;
        const double o[3] = {origin.x, origin.y, origin.z};
        check(pt_camera_new(o, width, height, screen_distance, fov_degrees, &c.pod_));
        return c;
    }
    static Camera look_at(Vector3 origin, Vector3 target, Vector3 up, uint32_t width, uint32_t height, double fov_degrees) {
        Camera c;
        const double o[3] = {origin.x, origin.y, origin.z}, t[3] = {target.x, target.y, target.z}, u[3] = {up.x, up.y, up.z};
        check(pt_camera_look_at(o, t, u, width, height, fov_degrees, &c.pod_));
        return c;
    }
    // camera.rs:139-147 (x, y = pixel, y counted from the bottom row: world.rs:299 passes HEIGHT-1-y)
    Ray get_ray_with_offset(uint32_t x, uint32_t y, double offset_x, double offset_y) const {
        const double u = ((double)x + offset_x) / (double)(pod_.width - 1), v = ((double)y + offset_y) / (double)(pod_.height - 1);
        Vector3 d;
        d.x = pod_.lower_left[0] + pod_.horizontal[0] * u + pod_.vertical[0] * v - pod_.origin[0];
        d.y = pod_.lower_left[1] + pod_.horizontal[1] * u + pod_.vertical[1] * v - pod_.origin[1];
        d.z = pod_.lower_left[2] + pod_.horizontal[2] * u + pod_.vertical[2] * v - pod_.origin[2];
        return Ray::new_(Vector3(pod_.origin[0], pod_.origin[1], pod_.origin[2]), d);
    }
    const PtCamera& pod() const { return pod_; }
    uint32_t width() const { return pod_.width; }
    uint32_t height() const { return pod_.height; }
private:
    PtCamera pod_{};
};

// ---- shape.rs: the Shape trait (shape.rs:8-35), evaluated on the device
namespace detail {
inline PtObject shape_pod(uint32_t tag, const double* v, int n) {
    PtObject o{}; o.shape_tag = tag; o.mat_tag = PT_MAT_LAMBERT;
    for (int k = 0; k < n; ++k) o.shape[k] = v[k];
    o.mat[0] = o.mat[1] = o.mat[2] = 0.5;
    return o;
}
inline std::optional<HitRecord> shape_hit(const PtObject& o, const Ray& ray, double t_min, double t_max, uint32_t exact_math) {
    const double r6[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z};
    int32_t id = -1; float rec[8];
    check(pt_debug_hit_records(probe().get(o), r6, 1, t_min, t_max, exact_math, PT_ACCEL_LINEAR, &id, rec));
    if (id < 0) return std::nullopt;
    return record(rec);
}
struct SurfaceSample { Vector3 point; double pdf_omega; Vector3 light_dir; double distance; };   // shape.rs:30-34 minus the normal
inline SurfaceSample shape_sample(const PtObject& o, const Vector3& from, const Vector3* target, double r1, double r2, uint32_t exact_math) {
    double f[3], t[3], r[2] = {r1, r2}; float out[8];
    put3(f, from); if (target) put3(t, *target);
    check(pt_debug_shape_sample(probe().get(o), 0, f, target ? t : nullptr, target ? nullptr : r, 1, exact_math, out));
    return SurfaceSample{get3(out), out[3], get3(out + 4), out[7]};
}
}  // namespace detail
struct SphereShape {
    Vector3 center; double radius;
    static SphereShape new_(Vector3 center, double radius) { return SphereShape{center, radius}; }
    PtObject pod() const { const double v[4] = {center.x, center.y, center.z, radius}; return detail::shape_pod(PT_SHAPE_SPHERE, v, 4); }
    // Shape::hit (shape.rs:53-89), on the device
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max, uint32_t exact_math = 0) const { return detail::shape_hit(pod(), ray, t_min, t_max, exact_math); }
    // Shape::sample_surface_from_point (shape.rs:91-145): target = the look-ahead form, else (r1, r2) are the two uniforms
    detail::SurfaceSample sample_surface_from_point(const HitRecord& from, const HitRecord* target, double r1, double r2, uint32_t exact_math = 0) const {
        return detail::shape_sample(pod(), from.point, target ? &target->point : nullptr, r1, r2, exact_math);
    }
};
struct TriangleShape {
    Vector3 v0, v1, v2;
    static TriangleShape new_(Vector3 v0, Vector3 v1, Vector3 v2) { return TriangleShape{v0, v1, v2}; }
    PtObject pod() const { const double v[9] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z}; return detail::shape_pod(PT_SHAPE_TRIANGLE, v, 9); }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max, uint32_t exact_math = 0) const { return detail::shape_hit(pod(), ray, t_min, t_max, exact_math); }   // shape.rs:161-198
    detail::SurfaceSample sample_surface_from_point(const HitRecord& from, const HitRecord* target, double r1, double r2, uint32_t exact_math = 0) const {   // shape.rs:200-242
        return detail::shape_sample(pod(), from.point, target ? &target->point : nullptr, r1, r2, exact_math);
    }
};

// ---- material.rs / mirror.rs: the Material trait (material.rs:5-65), evaluated on the device
namespace detail {
struct BsdfSample { Vector3 direction, bsdf; double pdf, cos_theta; };     // material.rs:28-40's tuple
inline PtObject mat_pod(uint32_t tag, const double* m, int n) {
    PtObject o{}; o.shape_tag = PT_SHAPE_SPHERE; o.shape[3] = 1.0; o.mat_tag = tag;
    for (int k = 0; k < n; ++k) o.mat[k] = m[k];
    return o;
}
inline std::pair<Vector3, double> mat_bsdf_pdf(const PtObject& o, const Ray& ray, const Vector3& wo, const Vector3& normal, uint32_t exact_math) {
    double in[10]; float out[4];
    put3(in, ray.direction); put3(in + 3, wo); put3(in + 6, normal); in[9] = ray.eta_ratio;
    check(pt_debug_bsdf_eval(probe().get(o), 0, in, 1, exact_math, out));
    return {get3(out), out[3]};
}
inline BsdfSample mat_bsdf_pdf_sample(const PtObject& o, const Ray& ray, const Vector3& normal, const uint32_t words[3], uint32_t exact_math) {
    double in[7]; float out[8]; const uint32_t w[4] = {words[0], words[1], words[2], 0u};
    put3(in, ray.direction); put3(in + 3, normal); in[6] = ray.eta_ratio;
    check(pt_debug_bsdf_sample(probe().get(o), 0, in, w, 1, exact_math, out));
    return BsdfSample{get3(out), get3(out + 3), out[6], out[7]};
}
}  // namespace detail
#define PT_MATERIAL_METHODS                                                                                                         \
    /* Material::bsdf_pdf(x, ray, o, normal) -> (bsdf, pdf); x is unused by every material of the reference */                     \
    std::pair<Vector3, double> bsdf_pdf(const Ray& ray, const Vector3& o, const Vector3& normal, uint32_t exact_math = 0) const {  \
        return detail::mat_bsdf_pdf(pod(), ray, o, normal, exact_math); }                                                           \
    /* Material::bsdf_pdf_sample(x, ray, normal, rng): words = the raw draws (r1, r2, lobe u) the rng would deliver */              \
    detail::BsdfSample bsdf_pdf_sample(const Ray& ray, const Vector3& normal, const uint32_t words[3], uint32_t exact_math = 0) const { \
        return detail::mat_bsdf_pdf_sample(pod(), ray, normal, words, exact_math); }
struct LambertianCosineWeighted {
    Vector3 albedo;
    static LambertianCosineWeighted new_(Vector3 albedo) { return LambertianCosineWeighted{albedo}; }
    PtObject pod() const { const double m[3] = {albedo.x, albedo.y, albedo.z}; return detail::mat_pod(PT_MAT_LAMBERT, m, 3); }
    PT_MATERIAL_METHODS
    double get_eta() const { return 1.0; }                                // material.rs:50-52
    Vector3 emit() const { return Vector3::zero(); }                      // material.rs:62-64
};
struct Emissive {
    Vector3 emission;
    static Emissive new_(Vector3 emission) { return Emissive{emission}; }
    PtObject pod() const { const double m[3] = {emission.x, emission.y, emission.z}; return detail::mat_pod(PT_MAT_EMISSIVE, m, 3); }
    PT_MATERIAL_METHODS
    double get_eta() const { return 1.0; }
    Vector3 emit() const { return emission; }                             // material.rs:160-162
};
struct OrenNayar {
    Vector3 albedo; double roughness;
    static OrenNayar new_(Vector3 albedo, double roughness) { return OrenNayar{albedo, roughness}; }
    PtObject pod() const { const double m[4] = {albedo.x, albedo.y, albedo.z, roughness}; return detail::mat_pod(PT_MAT_OREN_NAYAR, m, 4); }
    PT_MATERIAL_METHODS
    double get_eta() const { return 1.0; }
    Vector3 emit() const { return Vector3::zero(); }
};
struct Mirror {            // public fields, constructed literally in the reference (world.rs:204-209)
    double roughness; Vector3 color; double metallic; double ior;
    PtObject pod() const { const double m[6] = {roughness, color.x, color.y, color.z, metallic, ior}; return detail::mat_pod(PT_MAT_MIRROR, m, 6); }
    PT_MATERIAL_METHODS
    double get_eta() const { return ior; }                                // mirror.rs:317-319
    Vector3 emit() const { return Vector3::zero(); }
};
#undef PT_MATERIAL_METHODS

// ---- object.rs: Object::new(shape, material) and its forwards (object.rs:27-54)
class Object {
public:
    template <class S, class M> static Object new_(const S& shape, const M& material) {
        Object o;
        const PtObject sp = shape.pod(), mp = material.pod();
        o.pod_.shape_tag = sp.shape_tag; std::memcpy(o.pod_.shape, sp.shape, sizeof sp.shape);
        o.pod_.mat_tag = mp.mat_tag; std::memcpy(o.pod_.mat, mp.mat, sizeof mp.mat);
        return o;
    }
    static Object from_pod(const PtObject& p) { Object o; o.pod_ = p; return o; }
    const PtObject& pod() const { return pod_; }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max, uint32_t exact_math = 0) const { return detail::shape_hit(pod_, ray, t_min, t_max, exact_math); }
    std::pair<Vector3, double> bsdf_pdf(const Ray& ray, const Vector3& o, const Vector3& normal, uint32_t exact_math = 0) const { return detail::mat_bsdf_pdf(pod_, ray, o, normal, exact_math); }
    detail::BsdfSample bsdf_pdf_sample(const Ray& ray, const Vector3& normal, const uint32_t words[3], uint32_t exact_math = 0) const { return detail::mat_bsdf_pdf_sample(pod_, ray, normal, words, exact_math); }
private:
    PtObject pod_{};
};

// ---- world.rs
class World {
public:
    // World::new(): the reference's camera (world.rs:67-73) and Cornell box (world.rs:80-211)
    static World new_() {
        World w(Camera::new_(Vector3(0.0, 0.0, 2.0), WIDTH, HEIGHT, 1.0, 35.0));
        uint32_t n = 0;
        check(pt_builtin_scene(1, 0, nullptr, 0, &n));
        w.objects_.resize(n);
        check(pt_builtin_scene(1, 0, w.objects_.data(), n, &n));
        return w;
    }
    // a World with caller-authored objects (the reference hard-codes its scene)
    World(const Camera& camera, const std::vector<Object>& objects = {}) : camera_(camera) {
        for (const Object& o : objects) objects_.push_back(o.pod());
        pt_default_params(&params_);
        resize_film();
    }
    ~World() { if (ctx_) pt_context_destroy(ctx_); }
    World(const World&) = delete;
    World& operator=(const World&) = delete;
    World(World&& o) noexcept { *this = std::move(o); }
    World& operator=(World&& o) noexcept {
        if (this != &o) {
            if (ctx_) pt_context_destroy(ctx_);
            camera_ = o.camera_; objects_ = std::move(o.objects_); params_ = o.params_;
            data = std::move(o.data); luminance_data = std::move(o.luminance_data);
            ctx_ = o.ctx_; o.ctx_ = nullptr; uploaded_ = o.uploaded_;
        }
        return *this;
    }

    void push(const Object& o) { objects_.push_back(o.pod()); uploaded_ = false; }
    Object object(size_t i) const { return Object::from_pod(objects_.at(i)); }

    // ---- the reference's per-call surface, every call a batch of one on the device (batch forms below)
    // World::hit_scene (world.rs:270-290): closest hit and the index of the object hit
    std::optional<std::pair<HitRecord, size_t>> hit_scene(const Ray& ray, double t_min, double t_max, int device = 0) {
        const double r6[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z};
        int32_t id = -1; float rec[8];
        check(pt_debug_hit_records(scene(device), r6, 1, t_min, t_max, params_.exact_math, params_.accel, &id, rec));
        if (id < 0) return std::nullopt;
        return std::make_pair(detail::record(rec), (size_t)id);
    }
    // ... n rays at once: ids[i] = object index or -1, recs[i] valid where ids[i] >= 0
    void hit_scene_batch(const std::vector<Ray>& rays, double t_min, double t_max, std::vector<int32_t>& ids, std::vector<HitRecord>& recs, int device = 0) {
        std::vector<double> r6(6 * rays.size());
        for (size_t i = 0; i < rays.size(); ++i) { detail::put3(&r6[6 * i], rays[i].origin); detail::put3(&r6[6 * i + 3], rays[i].direction); }
        ids.assign(rays.size(), -1); recs.assign(rays.size(), HitRecord{});
        std::vector<float> rec(8 * rays.size() + 8);
        check(pt_debug_hit_records(scene(device), r6.data(), (uint32_t)rays.size(), t_min, t_max, params_.exact_math, params_.accel, ids.data(), rec.data()));
        for (size_t i = 0; i < rays.size(); ++i) recs[i] = detail::record(&rec[8 * i]);
    }
    // World::sample_light_point (world.rs:251-267): words = the raw draws the rng would deliver (light index, r1, r2);
    // nullopt when the scene has no light (world.rs:252-254)
    std::optional<LightSample> sample_light_point(const HitRecord& hit, const uint32_t words[3], int device = 0) {
        double f[3]; float out[8]; const uint32_t w[4] = {words[0], words[1], words[2], 0u};
        detail::put3(f, hit.point);
        check(pt_debug_light_point(scene(device), f, w, 1, params_.exact_math, out));
        if (out[7] < 0.0f) return std::nullopt;
        LightSample ls; ls.point = detail::get3(out); ls.emission = detail::get3(out + 3); ls.pdf = out[6]; ls.light_object = (uint32_t)out[7];
        return ls;
    }
    // World::render_pixel(x, y, rng) (world.rs:293-333): SAMPLE_NUM = params().spp samples of the pixel with the key
    // (y<<32)|x (main.rs:51); stores the linear mean in luminance_data[y*W+x] (world.rs:318-319) and returns the Color
    Color render_pixel(uint32_t x, uint32_t y, int device = 0) {
        const uint32_t xy[2] = {x, y}; float lin[3]; uint8_t c[4];
        check(pt_render_pixels(scene(device), &camera_.pod(), &whole_image_params(), xy, 1, lin, c, nullptr));
        const size_t idx = (size_t)y * camera_.width() + x;
        luminance_data[idx] = Vector3(lin[0], lin[1], lin[2]);
        return Color{c[0], c[1], c[2], c[3]};
    }
    // ... a pixel list at once, written into data[] and luminance_data[] like the closure of main.rs:48-60; samples
    // (optional) receives ray_color of every camera sample, [pixel][sample]
    void render_pixels(const std::vector<std::pair<uint32_t, uint32_t>>& pixels, std::vector<std::vector<Vector3>>* samples = nullptr, int device = 0) {
        const uint32_t n = (uint32_t)pixels.size();
        std::vector<uint32_t> xy(2 * (size_t)n + 2);
        for (uint32_t i = 0; i < n; ++i) { xy[2 * i] = pixels[i].first; xy[2 * i + 1] = pixels[i].second; }
        std::vector<float> lin(3 * (size_t)n + 3), smp(samples ? 3 * (size_t)n * params_.spp + 3 : 0);
        std::vector<uint8_t> c(4 * (size_t)n + 4);
        check(pt_render_pixels(scene(device), &camera_.pod(), &whole_image_params(), xy.data(), n, lin.data(), c.data(), samples ? smp.data() : nullptr));
        for (uint32_t i = 0; i < n; ++i) {
            const size_t idx = (size_t)pixels[i].second * camera_.width() + pixels[i].first;
            luminance_data[idx] = Vector3(lin[3 * i], lin[3 * i + 1], lin[3 * i + 2]);
            data[idx] = Color{c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]};
        }
        if (samples) {
            samples->assign(n, std::vector<Vector3>(params_.spp));
            for (uint32_t i = 0; i < n; ++i)
                for (uint32_t k = 0; k < params_.spp; ++k) { const float* q = &smp[3 * ((size_t)i * params_.spp + k)]; (*samples)[i][k] = Vector3(q[0], q[1], q[2]); }
        }
    }
    // RenderingStrategy::ray_color(world, ray, 0, rng, Vector3::one()) (rendering.rs:34-142 MIS, 214-265 BRDF-only per
    // params().integrator): the rng is the stream of pixel key (x, y), sample index `sample`
    Vector3 ray_color(const Ray& ray, uint32_t x, uint32_t y, uint32_t sample, int device = 0) {
        const double r6[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z};
        const uint32_t xy[2] = {x, y}; float out[3];
        PtRenderParams p = whole_image_params(); p.spp = 1; p.spp_offset = sample;
        check(pt_ray_color(scene(device), &p, r6, xy, 1, out));
        return Vector3(out[0], out[1], out[2]);
    }
    PtRenderParams& params() { return params_; }        // spp (= SAMPLE_NUM), depth policy, integrator
    const Camera& camera() const { return camera_; }
    size_t object_count() const { return objects_.size(); }

    // The rayon loop of main() (main.rs:43-60) as one call: every pixel gets the RNG key (x, y)
    // (main.rs:51), render_pixel's result lands in data[y*W+x] (main.rs:58-59) and the linear
    // mean in luminance_data[y*W+x] (world.rs:318-319).  Runs on the GPU; throws PtError on failure.
    void render(int device = 0) {
        scene(device);
        resize_film();
        PtRenderParams p = params_;
        p.band_rows = 0; p.band_index = 0; p.band_count = 1;
        render_host(p);
    }
    // Progressive form of render(): after every spp_step samples `data` / `luminance_data` hold the mean so
    // far and on_frame(spp_done) is called -- what the reference's window shows while rendering
    // (main.rs:79-90).  on_frame returning true stops early.  The last frame equals render()'s.
    template <class F> void render_progressive(uint32_t spp_step, F on_frame, int device = 0) {
        scene(device);
        resize_film();
        PtRenderParams p = params_;
        p.band_rows = 0; p.band_index = 0; p.band_count = 1;
        const size_t n = (size_t)camera_.width() * camera_.height();
        std::vector<float> lin(n * 3);
        std::vector<uint8_t> rgba(n * 4);
        struct Ctx { World* w; F* f; std::vector<float>* lin; std::vector<uint8_t>* rgba; } cx{this, &on_frame, &lin, &rgba};
        auto tramp = [](void* u, uint32_t done, uint32_t, const uint8_t*, const float*) -> int {
            Ctx* c = static_cast<Ctx*>(u);
            c->w->unpack(*c->lin, *c->rgba);
            return (*c->f)(done) ? 1 : 0;
        };
        check(pt_render_progressive(ctx_, &camera_.pod(), &p, spp_step, tramp, &cx, lin.data(), rgba.data()));
        unpack(lin, rgba);
    }
    PtStats stats() { PtStats s{}; if (ctx_) check(pt_get_stats(ctx_, &s)); return s; }

    // World::draw (world.rs:335-341): blit RGBA8 into a frame of 4*W*H bytes
    void draw(uint8_t* frame) const {
        for (size_t i = 0; i < data.size(); ++i) { frame[4 * i] = data[i].r; frame[4 * i + 1] = data[i].g; frame[4 * i + 2] = data[i].b; frame[4 * i + 3] = data[i].a; }
    }
    // World::export_luminance (world.rs:344-369): "x,y,r,g,b,luminance", 6 decimals, Rec.709
    void export_luminance(const std::string& path) const {
        FILE* f = std::fopen(path.c_str(), "w");
        if (!f) throw std::runtime_error("cannot create " + path);
        std::fprintf(f, "x,y,r,g,b,luminance\n");
        const uint32_t W = camera_.width(), H = camera_.height();
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                const Vector3& v = luminance_data[(size_t)y * W + x];
                const double lum = 0.2126 * v.x + 0.7152 * v.y + 0.0722 * v.z;
                std::fprintf(f, "%u,%u,%.6f,%.6f,%.6f,%.6f\n", x, y, v.x, v.y, v.z, lum);
            }
        std::fclose(f);
    }
    // Reader of the same format (the reference only writes it): fills luminance_data; the file's size must be the
    // camera's.  Returns the number of pixels read.
    size_t import_luminance(const std::string& path) {
        uint32_t w = 0, h = 0;
        std::vector<Vector3> img = read_luminance(path, w, h);
        if (w != camera_.width() || h != camera_.height())
            throw std::runtime_error(path + ": " + std::to_string(w) + "x" + std::to_string(h) + " does not match the camera");
        luminance_data = std::move(img);
        return luminance_data.size();
    }
    // "x,y,r,g,b,luminance" rows (world.rs:350-366) -> row-major image; the size is taken from the largest x, y
    static std::vector<Vector3> read_luminance(const std::string& path, uint32_t& w, uint32_t& h) {
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) throw std::runtime_error("cannot open " + path);
        char line[256];
        if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "x,y,r,g,b,luminance", 19) != 0) { std::fclose(f); throw std::runtime_error(path + ": not a luminance.csv (header)"); }
        struct Row { uint32_t x, y; double r, g, b; };
        std::vector<Row> rows;
        w = h = 0;
        while (std::fgets(line, sizeof line, f)) {
            Row q; double lum;
            if (std::sscanf(line, "%u,%u,%lf,%lf,%lf,%lf", &q.x, &q.y, &q.r, &q.g, &q.b, &lum) != 6) { std::fclose(f); throw std::runtime_error(path + ": malformed row"); }
            if (q.x + 1 > w) w = q.x + 1;
            if (q.y + 1 > h) h = q.y + 1;
            rows.push_back(q);
        }
        std::fclose(f);
        if (rows.size() != (size_t)w * h) throw std::runtime_error(path + ": " + std::to_string(rows.size()) + " rows for a " + std::to_string(w) + "x" + std::to_string(h) + " image");
        std::vector<Vector3> img((size_t)w * h);
        for (const Row& q : rows) img[(size_t)q.y * w + q.x] = Vector3(q.r, q.g, q.b);
        return img;
    }
    // convenience for headless use: binary PPM of `data`
    void write_ppm(const std::string& path) const {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + path);
        std::fprintf(f, "P6\n%u %u\n255\n", camera_.width(), camera_.height());
        for (const Color& c : data) { const uint8_t px[3] = {c.r, c.g, c.b}; std::fwrite(px, 1, 3, f); }
        std::fclose(f);
    }

    std::vector<Color> data;                 // World.data, world.rs:55
    std::vector<Vector3> luminance_data;     // World.luminance_data, world.rs:57

private:
    Camera camera_;
    std::vector<PtObject> objects_;
    PtRenderParams params_{};
    PtContext* ctx_ = nullptr;
    bool uploaded_ = false;
    PtRenderParams tmp_params_{};

    // the context with this World's objects on it (uploaded again after push())
    PtContext* scene(int device) {
        check_abi();
        if (!ctx_) check(pt_context_create(device, &ctx_));
        if (!uploaded_) { check(pt_scene_upload(ctx_, objects_.data(), (uint32_t)objects_.size())); uploaded_ = true; }
        return ctx_;
    }
    const PtRenderParams& whole_image_params() {
        tmp_params_ = params_;
        tmp_params_.band_rows = 0; tmp_params_.band_index = 0; tmp_params_.band_count = 1;
        return tmp_params_;
    }

    void resize_film() {
        const size_t n = (size_t)camera_.width() * camera_.height();
        data.assign(n, Color{0, 0, 0, 255});                 // world.rs:228-235
        luminance_data.assign(n, Vector3::zero());           // world.rs:236
    }
    void render_host(const PtRenderParams& p) {
        const size_t n = (size_t)camera_.width() * camera_.height();
        std::vector<float> lin(n * 3);
        std::vector<uint8_t> rgba(n * 4);
        check(pt_render_host(ctx_, &camera_.pod(), &p, lin.data(), rgba.data()));
        unpack(lin, rgba);
    }
    void unpack(const std::vector<float>& lin, const std::vector<uint8_t>& rgba) {
        for (size_t i = 0; i < data.size(); ++i) {
            luminance_data[i] = Vector3(lin[3 * i], lin[3 * i + 1], lin[3 * i + 2]);
            data[i] = Color{rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]};
        }
    }
};

// Differ of two luminance images under the FP32 tolerance of SURVEY 8d (ii): per channel |d| <= 1e-3 + 1e-2 |ref| on
// >= 99.5 % of the pixels and image mean within 1e-3 relative.  (A real `cargo run` of the reference uses ChaCha12
// streams, not this library's Philox: against such a file only the image mean and the Monte-Carlo noise level are
// comparable, which `mean_rel` and `rmse` report.)
struct LuminanceDiff {
    size_t pixels = 0, outside = 0;
    double frac_within = 0.0, max_abs = 0.0, rmse = 0.0, mean_a = 0.0, mean_b = 0.0, mean_rel = 0.0;
    bool pass = false;
};
inline LuminanceDiff compare_luminance(const std::vector<Vector3>& a, const std::vector<Vector3>& ref, double abs_tol = 1e-3, double rel_tol = 1e-2,
                                       double frac_needed = 0.995, double mean_tol = 1e-3) {
    if (a.size() != ref.size()) throw std::runtime_error("compare_luminance: image sizes differ");
    LuminanceDiff d;
    d.pixels = a.size();
    double se = 0.0;
    for (size_t i = 0; i < a.size(); ++i) {
        const double da[3] = {a[i].x - ref[i].x, a[i].y - ref[i].y, a[i].z - ref[i].z}, rv[3] = {ref[i].x, ref[i].y, ref[i].z};
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            if (!(std::fabs(da[k]) <= abs_tol + rel_tol * std::fabs(rv[k]))) ok = false;
            if (std::fabs(da[k]) > d.max_abs) d.max_abs = std::fabs(da[k]);
            se += da[k] * da[k];
        }
        d.outside += ok ? 0 : 1;
        d.mean_a += (a[i].x + a[i].y + a[i].z) / 3.0; d.mean_b += (ref[i].x + ref[i].y + ref[i].z) / 3.0;
    }
    if (d.pixels) { d.mean_a /= (double)d.pixels; d.mean_b /= (double)d.pixels; d.rmse = std::sqrt(se / (3.0 * (double)d.pixels)); }
    d.frac_within = d.pixels ? 1.0 - (double)d.outside / (double)d.pixels : 1.0;
    d.mean_rel = d.mean_b != 0.0 ? std::fabs(d.mean_a - d.mean_b) / std::fabs(d.mean_b) : std::fabs(d.mean_a);
    d.pass = d.frac_within >= frac_needed && d.mean_rel <= mean_tol;
    return d;
}

}  // namespace pathtrace
