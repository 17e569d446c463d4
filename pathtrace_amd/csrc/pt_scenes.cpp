// pt_scenes.cpp -- host-side scene authoring helpers behind the C ABI: the camera
// constructors of src/camera.rs and the benchmark scenes of SURVEY 8(d).  Pure
// host code (f64, like the reference's scene setup); nothing here is on the hot path.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/pathtrace_amd.h"

namespace {

struct D3 { double x, y, z; };
D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 mul(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
D3 divs(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
D3 normalize(D3 a) {
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return l > 0.0 ? divs(a, l) : a;
}
void store(double* dst, D3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
const double kPiD = 3.14159265358979323846;

PtObject make_obj(uint32_t shape_tag, uint32_t mat_tag) {
    PtObject o;
    std::memset(&o, 0, sizeof o);
    o.shape_tag = shape_tag; o.mat_tag = mat_tag;
    return o;
}
PtObject sphere(double cx, double cy, double cz, double r, uint32_t mat_tag, double m0, double m1, double m2) {
    PtObject o = make_obj(PT_SHAPE_SPHERE, mat_tag);
    o.shape[0] = cx; o.shape[1] = cy; o.shape[2] = cz; o.shape[3] = r;
    o.mat[0] = m0; o.mat[1] = m1; o.mat[2] = m2;
    return o;
}
PtObject triangle(const double v[9], uint32_t mat_tag, double m0, double m1, double m2) {
    PtObject o = make_obj(PT_SHAPE_TRIANGLE, mat_tag);
    std::memcpy(o.shape, v, 9 * sizeof(double));
    o.mat[0] = m0; o.mat[1] = m1; o.mat[2] = m2;
    return o;
}

// host copy of Philox4x32-10, used only to place the random spheres of scene 4
void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
double u01(uint32_t r) { return (double)(((r >> 9) << 1) | 1u) * (1.0 / 16777216.0); }

// Scene 1: World::new()'s Cornell box (src/world.rs:76-211): five walls of two
// triangles, a two-triangle ceiling light just below the ceiling, one rough-glass
// sphere.  Corners are written as (sx, sy, sz) sign triples of the box
// [-1,1] x [-1,1] x [box_depth-1, box_depth+1].
void scene_cornell(std::vector<PtObject>& out) {
    const double bs = 1.0, bd = -2.0, ls = 0.3;
    auto corner = [&](int sx, int sy, int sz, double* dst) {
        dst[0] = sx * bs; dst[1] = sy * bs; dst[2] = sz > 0 ? bd + bs : bd - bs;
    };
    struct Wall { int c[6][3]; double albedo[3]; };
    // vertex order per triangle as authored in the reference (it fixes the geometric normal)
    const Wall walls[5] = {
        {{{-1, -1, -1}, {-1, 1, -1}, {-1, 1, 1}, {-1, -1, -1}, {-1, 1, 1}, {-1, -1, 1}}, {0.8, 0.1, 0.1}},   // left  :82-97
        {{{1, -1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, -1}, {1, -1, 1}, {1, 1, 1}}, {0.1, 0.8, 0.1}},         // right :99-114
        {{{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, -1, -1}, {1, 1, -1}, {-1, 1, -1}}, {0.2, 0.2, 0.8}},   // back  :116-131
        {{{-1, -1, -1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}, {-1, -1, 1}, {1, -1, 1}}, {0.2, 0.8, 0.8}},   // floor :133-148
        {{{-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, -1}, {1, 1, 1}, {-1, 1, 1}}, {0.8, 0.8, 0.8}},         // ceil  :150-165
    };
    for (const Wall& w : walls)
        for (int t = 0; t < 2; ++t) {
            double v[9];
            for (int k = 0; k < 3; ++k) corner(w.c[3 * t + k][0], w.c[3 * t + k][1], w.c[3 * t + k][2], v + 3 * k);
            out.push_back(triangle(v, PT_MAT_LAMBERT, w.albedo[0], w.albedo[1], w.albedo[2]));
        }
    // ceiling light, world.rs:167-182: y = box_size - 0.01, x,z in +-light_size around (0, box_depth)
    const double ly = bs - 0.01, z0 = bd - ls, z1 = bd + ls;
    const double la[9] = {-ls, ly, z0, ls, ly, z0, ls, ly, z1};
    const double lb[9] = {-ls, ly, z0, ls, ly, z1, -ls, ly, z1};
    out.push_back(triangle(la, PT_MAT_EMISSIVE, 15.0, 15.0, 15.0));
    out.push_back(triangle(lb, PT_MAT_EMISSIVE, 15.0, 15.0, 15.0));
    // glass sphere, world.rs:202-210
    PtObject g = make_obj(PT_SHAPE_SPHERE, PT_MAT_MIRROR);
    g.shape[0] = 0.4; g.shape[1] = -0.6; g.shape[2] = bd; g.shape[3] = 0.4;
    g.mat[0] = 0.3; g.mat[1] = 1.0; g.mat[2] = 1.0; g.mat[3] = 1.0; g.mat[4] = 0.0; g.mat[5] = 1.5;
    out.push_back(g);
}

// Scene 2 (SURVEY 8d "C2"): the same box built from spheres -- five wall spheres
// of radius R tangent to the wall planes with the reference's wall albedos, the
// reference's commented-out sphere light (world.rs:184-190), four diffuse spheres.
// R = 100: with R = 1000 the f32 hit distance on a wall is only good to ~1e-4,
// a tenth of the t_min = 1e-3 self-intersection margin; R = 100 gives ~1e-5
// (DESIGN.md "C2 wall radius").  Frozen; the golden fixtures depend on it.
void scene_cornell_spheres(std::vector<PtObject>& out) {
    const double R = 100.0, bs = 1.0, bd = -2.0;
    out.push_back(sphere(-(bs + R), 0.0, bd, R, PT_MAT_LAMBERT, 0.8, 0.1, 0.1));   // left
    out.push_back(sphere(bs + R, 0.0, bd, R, PT_MAT_LAMBERT, 0.1, 0.8, 0.1));      // right
    out.push_back(sphere(0.0, 0.0, bd - bs - R, R, PT_MAT_LAMBERT, 0.2, 0.2, 0.8));   // back
    out.push_back(sphere(0.0, -(bs + R), bd, R, PT_MAT_LAMBERT, 0.2, 0.8, 0.8));   // floor
    out.push_back(sphere(0.0, bs + R, bd, R, PT_MAT_LAMBERT, 0.8, 0.8, 0.8));      // ceiling
    out.push_back(sphere(0.0, bs - 0.21, bd, 0.2, PT_MAT_EMISSIVE, 36.0, 36.0, 36.0));
    out.push_back(sphere(-0.4, -0.6, bd, 0.4, PT_MAT_LAMBERT, 0.8, 0.8, 0.8));
    out.push_back(sphere(0.4, -0.6, bd, 0.4, PT_MAT_LAMBERT, 0.8, 0.8, 0.8));
    out.push_back(sphere(0.0, -0.8, -1.5, 0.2, PT_MAT_LAMBERT, 0.8, 0.6, 0.2));
    out.push_back(sphere(0.0, 0.1, -2.4, 0.25, PT_MAT_LAMBERT, 0.3, 0.3, 0.9));
}

// Scene 4 (SURVEY 8d "C4"): n random Lambertian spheres in the box volume, every
// 100th one an emitter.  Sphere i draws from philox(key = 0x5EED, ctr = (i, blk)).
void scene_random_spheres(uint32_t n, std::vector<PtObject>& out) {
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t a[4], b[4];
        philox(i, 0, 0, 0, 0x5EEDu, 0u, a);
        philox(i, 1, 0, 0, 0x5EEDu, 0u, b);
        double cx = -1.0 + 2.0 * u01(a[0]), cy = -1.0 + 2.0 * u01(a[1]), cz = -3.0 + 2.0 * u01(a[2]);
        double r = 0.005 + 0.025 * u01(a[3]);
        if (i % 100 == 0) out.push_back(sphere(cx, cy, cz, r, PT_MAT_EMISSIVE, 20.0, 20.0, 20.0));
        else out.push_back(sphere(cx, cy, cz, r, PT_MAT_LAMBERT, 0.2 + 0.7 * u01(b[0]), 0.2 + 0.7 * u01(b[1]),
                                  0.2 + 0.7 * u01(b[2])));
    }
}

}  // namespace

extern "C" {

// Camera::new, src/camera.rs:50-82.  NB the "horizontal" fov sizes the HEIGHT (camera.rs:61-62).
int pt_camera_new(const double origin[3], uint32_t width, uint32_t height, double screen_distance, double fov_degrees,
                  PtCamera* out) {
    if (!origin || !out || width == 0 || height == 0) return PT_ERR_INVALID_ARG;
    double fov = fov_degrees * (kPiD / 180.0);
    double aspect = (double)width / (double)height;
    double vh = 2.0 * std::tan(fov / 2.0) * screen_distance;
    double vw = vh * aspect;
    D3 o{origin[0], origin[1], origin[2]}, hor{vw, 0.0, 0.0}, ver{0.0, vh, 0.0};
    D3 llc = sub(sub(sub(o, divs(hor, 2.0)), divs(ver, 2.0)), D3{0.0, 0.0, screen_distance});
    store(out->origin, o); store(out->lower_left, llc); store(out->horizontal, hor); store(out->vertical, ver);
    out->width = width; out->height = height;
    return PT_OK;
}

// Camera::look_at, src/camera.rs:94-130.
int pt_camera_look_at(const double origin[3], const double target[3], const double up[3], uint32_t width,
                      uint32_t height, double fov_degrees, PtCamera* out) {
    if (!origin || !target || !up || !out || width == 0 || height == 0) return PT_ERR_INVALID_ARG;
    double fov = fov_degrees * (kPiD / 180.0);
    double aspect = (double)width / (double)height;
    D3 o{origin[0], origin[1], origin[2]}, t{target[0], target[1], target[2]}, u0{up[0], up[1], up[2]};
    D3 w = normalize(sub(o, t));
    D3 u = normalize(cross(u0, w));
    D3 v = cross(w, u);
    const double dist = 1.0;
    double vh = 2.0 * std::tan(fov / 2.0) * dist;
    double vw = vh * aspect;
    D3 hor = mul(u, vw), ver = mul(v, vh);
    D3 llc = sub(sub(sub(o, divs(hor, 2.0)), divs(ver, 2.0)), mul(w, dist));
    store(out->origin, o); store(out->lower_left, llc); store(out->horizontal, hor); store(out->vertical, ver);
    out->width = width; out->height = height;
    return PT_OK;
}

int pt_builtin_scene(uint32_t id, uint32_t arg, PtObject* objs, uint32_t cap, uint32_t* n) {
    if (!n) return PT_ERR_INVALID_ARG;
    std::vector<PtObject> v;
    switch (id) {
        case 1: scene_cornell(v); break;
        case 2: scene_cornell_spheres(v); break;
        case 4: scene_random_spheres(arg ? arg : 10000u, v); break;
        default: return PT_ERR_INVALID_ARG;
    }
    *n = (uint32_t)v.size();
    if (objs) {
        uint32_t m = cap < *n ? cap : *n;
        std::memcpy(objs, v.data(), m * sizeof(PtObject));
    }
    return PT_OK;
}

}  // extern "C"
