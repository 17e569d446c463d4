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
// Shapes and materials are value types here (the device needs POD, not Box<dyn>);
// Object::new flattens them to the PtObject the ABI takes, preserving World.objects
// order.  The arithmetic of the hot path (hit, bsdf_pdf, ray_color, ...) is NOT
// mirrored on the host: it lives in the HIP kernels, and this header has no CPU
// rendering path -- render() fails if the GPU library reports an error.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
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
    const PtCamera& pod() const { return pod_; }
    uint32_t width() const { return pod_.width; }
    uint32_t height() const { return pod_.height; }
private:
    PtCamera pod_{};
};

// ---- shape.rs
struct SphereShape {
    Vector3 center; double radius;
    static SphereShape new_(Vector3 center, double radius) { return SphereShape{center, radius}; }
};
struct TriangleShape {
    Vector3 v0, v1, v2;
    static TriangleShape new_(Vector3 v0, Vector3 v1, Vector3 v2) { return TriangleShape{v0, v1, v2}; }
};

// ---- material.rs / mirror.rs
struct LambertianCosineWeighted {
    Vector3 albedo;
    static LambertianCosineWeighted new_(Vector3 albedo) { return LambertianCosineWeighted{albedo}; }
};
struct Emissive {
    Vector3 emission;
    static Emissive new_(Vector3 emission) { return Emissive{emission}; }
};
struct OrenNayar {
    Vector3 albedo; double roughness;
    static OrenNayar new_(Vector3 albedo, double roughness) { return OrenNayar{albedo, roughness}; }
};
struct Mirror {            // public fields, constructed literally in the reference (world.rs:204-209)
    double roughness; Vector3 color; double metallic; double ior;
};

// ---- object.rs: Object::new(shape, material)
class Object {
public:
    template <class S, class M> static Object new_(const S& shape, const M& material) {
        Object o;
        set_shape(o.pod_, shape);
        set_material(o.pod_, material);
        return o;
    }
    const PtObject& pod() const { return pod_; }
private:
    PtObject pod_{};
    static void put(double* dst, Vector3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
    static void set_shape(PtObject& p, const SphereShape& s) { p.shape_tag = PT_SHAPE_SPHERE; put(p.shape, s.center); p.shape[3] = s.radius; }
    static void set_shape(PtObject& p, const TriangleShape& t) { p.shape_tag = PT_SHAPE_TRIANGLE; put(p.shape, t.v0); put(p.shape + 3, t.v1); put(p.shape + 6, t.v2); }
    static void set_material(PtObject& p, const LambertianCosineWeighted& m) { p.mat_tag = PT_MAT_LAMBERT; put(p.mat, m.albedo); }
    static void set_material(PtObject& p, const Emissive& m) { p.mat_tag = PT_MAT_EMISSIVE; put(p.mat, m.emission); }
    static void set_material(PtObject& p, const OrenNayar& m) { p.mat_tag = PT_MAT_OREN_NAYAR; put(p.mat, m.albedo); p.mat[3] = m.roughness; }
    static void set_material(PtObject& p, const Mirror& m) { p.mat_tag = PT_MAT_MIRROR; p.mat[0] = m.roughness; put(p.mat + 1, m.color); p.mat[4] = m.metallic; p.mat[5] = m.ior; }
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
            ctx_ = o.ctx_; o.ctx_ = nullptr;
        }
        return *this;
    }

    void push(const Object& o) { objects_.push_back(o.pod()); }
    PtRenderParams& params() { return params_; }        // spp (= SAMPLE_NUM), depth policy, integrator
    const Camera& camera() const { return camera_; }
    size_t object_count() const { return objects_.size(); }

    // The rayon loop of main() (main.rs:43-60) as one call: every pixel gets the RNG key (x, y)
    // (main.rs:51), render_pixel's result lands in data[y*W+x] (main.rs:58-59) and the linear
    // mean in luminance_data[y*W+x] (world.rs:318-319).  Runs on the GPU; throws PtError on failure.
    void render(int device = 0) {
        if (!ctx_) check(pt_context_create(device, &ctx_));
        check(pt_scene_upload(ctx_, objects_.data(), (uint32_t)objects_.size()));
        resize_film();
        PtRenderParams p = params_;
        p.band_rows = 0; p.band_index = 0; p.band_count = 1;
        render_host(p);
    }
    // Progressive form of render(): after every spp_step samples `data` / `luminance_data` hold the mean so
    // far and on_frame(spp_done) is called -- what the reference's window shows while rendering
    // (main.rs:79-90).  on_frame returning true stops early.  The last frame equals render()'s.
    template <class F> void render_progressive(uint32_t spp_step, F on_frame, int device = 0) {
        if (!ctx_) check(pt_context_create(device, &ctx_));
        check(pt_scene_upload(ctx_, objects_.data(), (uint32_t)objects_.size()));
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

}  // namespace pathtrace
