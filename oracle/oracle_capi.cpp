// oracle_capi.cpp -- C entry points of the CPU ORACLE (test infrastructure only;
// see the header of pt_oracle.hpp for who may use it and what pins it).
//
// Modes of orc_render: precision {f64, f32} x form {recursive, iterative}.
//   f64 recursive  = the reference-faithful restatement (rendering.rs:34-142).
//   f32 iterative  = the device-equivalent computation.
// Threads: a std::thread pool over pixels mirroring rayon's par_iter
// (src/main.rs:48); threads=1 is the scalar port.
#include <atomic>
#include <cstdio>
#include <thread>

#include "pt_oracle.hpp"

using namespace orc;

namespace {

template <class R> Camera<R> make_camera(const PtCamera* c) {
    Camera<R> cam;
    cam.origin = V3<R>((R)c->origin[0], (R)c->origin[1], (R)c->origin[2]);
    cam.lower_left = V3<R>((R)c->lower_left[0], (R)c->lower_left[1], (R)c->lower_left[2]);
    cam.horizontal = V3<R>((R)c->horizontal[0], (R)c->horizontal[1], (R)c->horizontal[2]);
    cam.vertical = V3<R>((R)c->vertical[0], (R)c->vertical[1], (R)c->vertical[2]);
    cam.width = c->width; cam.height = c->height;
    return cam;
}

Params make_params(const PtRenderParams* p) {
    Params q;
    q.min_depth = p->min_depth; q.max_depth = p->max_depth;
    q.integrator = p->integrator; q.t_min = p->t_min;
    return q;
}

template <class R>
int render_impl(const PtCamera* pc, const PtObject* objs, uint32_t n, const PtRenderParams* pp, int form, int threads,
                double* out_lin, uint8_t* out_rgba, uint64_t* out_counters) {
    Scene<R> scene = build_scene<R>(objs, n);
    Camera<R> cam = make_camera<R>(pc);
    Params prm = make_params(pp);
    std::vector<uint32_t> rows = tile_rows(pc->height, pp->band_rows, pp->band_index, pp->band_count);
    const uint32_t W = pc->width;
    const size_t npix = rows.size() * (size_t)W;
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::vector<Counters> cns(threads);
    auto worker = [&](int tid) {
        Counters& cn = cns[tid];
        const size_t chunk = 64;
        for (;;) {
            size_t b = next.fetch_add(chunk);
            if (b >= npix) break;
            size_t e = b + chunk < npix ? b + chunk : npix;
            for (size_t i = b; i < e; ++i) {
                uint32_t yl = (uint32_t)(i / W), x = (uint32_t)(i % W);
                uint32_t y = rows[yl];
                double lin[3]; uint8_t rg[4];
                render_pixel<R>(scene, cam, prm, (Form)form, x, y, pp->spp, pp->spp_offset, lin, rg, cn);
                if (out_lin) { out_lin[i * 3] = lin[0]; out_lin[i * 3 + 1] = lin[1]; out_lin[i * 3 + 2] = lin[2]; }
                if (out_rgba) std::memcpy(out_rgba + i * 4, rg, 4);
            }
        }
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    if (out_counters) {
        uint64_t v = 0, s = 0, sc = 0; uint32_t md = 0;
        for (auto& c : cns) { v += c.vertices; s += c.shadow_rays; sc += c.scans; if (c.max_depth > md) md = c.max_depth; }
        out_counters[0] = v; out_counters[1] = s; out_counters[2] = sc; out_counters[3] = md;
    }
    return 0;
}

}  // namespace


// precision: 64 or 32; form: 0 recursive, 1 iterative.
// out_lin: double[tile_pixels*3]; out_rgba: uint8[tile_pixels*4]; out_counters: u64[4]
// = {vertices, shadow_rays, scans, max_depth}.
extern "C" int orc_render(const PtCamera* cam, const PtObject* objs, uint32_t n, const PtRenderParams* p, int precision, int form,
               int threads, double* out_lin, uint8_t* out_rgba, uint64_t* out_counters) {
    if (!cam || !objs || !p || p->spp == 0) return 1;
    if (precision == 64) return render_impl<double>(cam, objs, n, p, form, threads, out_lin, out_rgba, out_counters);
    return render_impl<float>(cam, objs, n, p, form, threads, out_lin, out_rgba, out_counters);
}

extern "C" void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) { philox4x32(ctr, key, out, rounds); }
extern "C" int orc_draw_rounds(void) { return kDrawRounds; }
extern "C" double orc_u01(uint32_t r) { return u01(r); }
extern "C" uint32_t orc_rr_word(const uint32_t ds[4]) { return rr_word(ds); }

extern "C" void orc_camera_new(const double o[3], uint32_t w, uint32_t h, double dist, double fov, PtCamera* out) {
    camera_new(o, w, h, dist, fov, out);
}
extern "C" void orc_camera_look_at(const double o[3], const double t[3], const double up[3], uint32_t w, uint32_t h, double fov,
                        PtCamera* out) {
    camera_look_at(o, t, up, w, h, fov, out);
}

// Camera::get_ray_with_offset for n (x, y, ox, oy) tuples -> rays n*6 (o, d).
extern "C" void orc_camera_rays(const PtCamera* pc, int precision, uint32_t n, const uint32_t* xy, const double* off,
                     double* rays) {
    for (uint32_t i = 0; i < n; ++i) {
        if (precision == 64) {
            Ray<double> r = make_camera<double>(pc).get_ray_with_offset(xy[2 * i], xy[2 * i + 1], off[2 * i], off[2 * i + 1]);
            double v[6] = {r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z};
            std::memcpy(rays + 6 * i, v, sizeof v);
        } else {
            Ray<float> r = make_camera<float>(pc).get_ray_with_offset(xy[2 * i], xy[2 * i + 1], (float)off[2 * i],
                                                                     (float)off[2 * i + 1]);
            double v[6] = {r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z};
            std::memcpy(rays + 6 * i, v, sizeof v);
        }
    }
}

// World::hit_scene on n rays (o,d as doubles; direction normalised like Ray::new).
// out_id: object index or -1; out_t; out_pn: point(3)+normal(3); out_ff: front_face.
template <class R>
static void hit_scene_batch(const PtObject* objs, uint32_t nobj, const double* rays, uint32_t n, double tmin,
                            double tmax, int32_t* out_id, double* out_t, double* out_pn, uint8_t* out_ff) {
    Scene<R> sc = build_scene<R>(objs, nobj);
    for (uint32_t i = 0; i < n; ++i) {
        const double* r = rays + 6 * i;
        Ray<R> ray(V3<R>((R)r[0], (R)r[1], (R)r[2]), V3<R>((R)r[3], (R)r[4], (R)r[5]));
        Hit<R> h;
        int id = hit_scene<R>(sc, ray, (R)tmin, (R)tmax, h);
        out_id[i] = id;
        if (out_t) out_t[i] = id >= 0 ? (double)h.t : 0.0;
        if (out_pn) {
            double v[6] = {h.point.x, h.point.y, h.point.z, h.normal.x, h.normal.y, h.normal.z};
            if (id < 0) std::memset(v, 0, sizeof v);
            std::memcpy(out_pn + 6 * i, v, sizeof v);
        }
        if (out_ff) out_ff[i] = id >= 0 ? (h.front_face ? 1 : 0) : 0;
    }
}
extern "C" void orc_hit_scene(const PtObject* objs, uint32_t nobj, int precision, const double* rays, uint32_t n, double tmin,
                   double tmax, int32_t* out_id, double* out_t, double* out_pn, uint8_t* out_ff) {
    if (precision == 64) hit_scene_batch<double>(objs, nobj, rays, n, tmin, tmax, out_id, out_t, out_pn, out_ff);
    else hit_scene_batch<float>(objs, nobj, rays, n, tmin, tmax, out_id, out_t, out_pn, out_ff);
}

// Shape::sample_surface_from_point for object `obj` from n points.
// in: from n*3, target n*3 or NULL (then uses r12 n*2).  out: n*11 =
// point3 normal3 pdf dir3 dist.
template <class R>
static void shape_sample_batch(const PtObject* po, const double* from, const double* target, const double* r12,
                               uint32_t n, double* out) {
    Scene<R> sc = build_scene<R>(po, 1);
    for (uint32_t i = 0; i < n; ++i) {
        Hit<R> f; f.point = V3<R>((R)from[3 * i], (R)from[3 * i + 1], (R)from[3 * i + 2]);
        Hit<R> tg;
        if (target) tg.point = V3<R>((R)target[3 * i], (R)target[3 * i + 1], (R)target[3 * i + 2]);
        V3<R> p, nn, d; R pdf, dist;
        shape_sample<R>(sc.objs[0], f, target ? &tg : nullptr, r12 ? (R)r12[2 * i] : R(0), r12 ? (R)r12[2 * i + 1] : R(0),
                        p, nn, pdf, d, dist);
        double v[11] = {p.x, p.y, p.z, nn.x, nn.y, nn.z, (double)pdf, d.x, d.y, d.z, (double)dist};
        std::memcpy(out + 11 * i, v, sizeof v);
    }
}
extern "C" void orc_shape_sample(const PtObject* po, int precision, const double* from, const double* target, const double* r12,
                      uint32_t n, double* out) {
    if (precision == 64) shape_sample_batch<double>(po, from, target, r12, n, out);
    else shape_sample_batch<float>(po, from, target, r12, n, out);
}

// Material::bsdf_pdf for object `obj`: in n*(dir_in3, wo3, normal3, eta); out n*4 (f3, pdf)
template <class R> static void bsdf_eval_batch(const PtObject* po, const double* in, uint32_t n, double* out) {
    Scene<R> sc = build_scene<R>(po, 1);
    for (uint32_t i = 0; i < n; ++i) {
        const double* q = in + 10 * i;
        Ray<R> ray; ray.direction = V3<R>((R)q[0], (R)q[1], (R)q[2]); ray.eta_ratio = (R)q[9];
        V3<R> wo((R)q[3], (R)q[4], (R)q[5]), nn((R)q[6], (R)q[7], (R)q[8]);
        V3<R> f; R pdf;
        bsdf_pdf<R>(sc.objs[0], ray, wo, nn, f, pdf);
        double v[4] = {f.x, f.y, f.z, (double)pdf};
        std::memcpy(out + 4 * i, v, sizeof v);
    }
}
extern "C" void orc_bsdf_eval(const PtObject* po, int precision, const double* in, uint32_t n, double* out) {
    if (precision == 64) bsdf_eval_batch<double>(po, in, n, out); else bsdf_eval_batch<float>(po, in, n, out);
}

// Material::bsdf_pdf_sample: in n*(dir_in3, normal3, eta), draws n*4 u32 = (r1, r2, lobe u, unused) raw words;
// out n*8 (wo3, f3, pdf, cos)
template <class R>
static void bsdf_sample_batch(const PtObject* po, const double* in, const uint32_t* draws, uint32_t n, double* out) {
    Scene<R> sc = build_scene<R>(po, 1);
    for (uint32_t i = 0; i < n; ++i) {
        const double* q = in + 7 * i;
        Ray<R> ray; ray.direction = V3<R>((R)q[0], (R)q[1], (R)q[2]); ray.eta_ratio = (R)q[6];
        V3<R> nn((R)q[3], (R)q[4], (R)q[5]);
        V3<R> wo, f; R pdf, c;
        bsdf_pdf_sample<R>(sc.objs[0], ray, nn, draws[4 * i], draws[4 * i + 1], draws[4 * i + 2], wo, f, pdf, c);
        double v[8] = {wo.x, wo.y, wo.z, f.x, f.y, f.z, (double)pdf, (double)c};
        std::memcpy(out + 8 * i, v, sizeof v);
    }
}
extern "C" void orc_bsdf_sample(const PtObject* po, int precision, const double* in, const uint32_t* draws, uint32_t n,
                     double* out) {
    if (precision == 64) bsdf_sample_batch<double>(po, in, draws, n, out);
    else bsdf_sample_batch<float>(po, in, draws, n, out);
}

// World::sample_light_point (world.rs:251-267) from n points: words n*4 = (light-index word, r1 word, r2 word, -);
// out n*8 = point3, emission3, pdf, light object index (-1: the scene has no light).
template <class R>
static void light_point_batch(const PtObject* objs, uint32_t nobj, const double* from, const uint32_t* words, uint32_t n, double* out) {
    Scene<R> sc = build_scene<R>(objs, nobj);
    for (uint32_t i = 0; i < n; ++i) {
        Hit<R> h; h.point = V3<R>((R)from[3 * i], (R)from[3 * i + 1], (R)from[3 * i + 2]);
        LightSample<R> ls;
        double v[8] = {0, 0, 0, 0, 0, 0, 0, -1};
        if (sample_light_point<R>(sc, h, words[4 * i], words[4 * i + 1], words[4 * i + 2], ls)) {
            const uint32_t li = (uint32_t)(((uint64_t)words[4 * i] * sc.lights.size()) >> 32);
            double w[8] = {ls.point.x, ls.point.y, ls.point.z, ls.emission.x, ls.emission.y, ls.emission.z, (double)ls.pdf,
                           (double)sc.lights[li]};
            std::memcpy(v, w, sizeof v);
        }
        std::memcpy(out + 8 * i, v, sizeof v);
    }
}
extern "C" void orc_light_point(const PtObject* objs, uint32_t nobj, int precision, const double* from, const uint32_t* words,
                                uint32_t n, double* out) {
    if (precision == 64) light_point_batch<double>(objs, nobj, from, words, n, out);
    else light_point_batch<float>(objs, nobj, from, words, n, out);
}

// RenderingStrategy::ray_color(world, ray, 0, rng, 1) (rendering.rs:34, 214) for n arbitrary rays: rays n*6 (Ray::new
// normalises the direction), xy n*2 = RNG key of each ray's stream, `sample` = its sample index.  out n*3.
template <class R>
static void ray_color_batch(const PtObject* objs, uint32_t nobj, const PtRenderParams* pp, int form, const double* rays,
                            const uint32_t* xy, uint32_t n, double* out) {
    Scene<R> sc = build_scene<R>(objs, nobj);
    Params prm = make_params(pp);
    Counters cn;
    for (uint32_t i = 0; i < n; ++i) {
        const double* r = rays + 6 * i;
        Ray<R> ray(V3<R>((R)r[0], (R)r[1], (R)r[2]), V3<R>((R)r[3], (R)r[4], (R)r[5]));
        Draws dr{{xy[2 * i], xy[2 * i + 1]}, pp->spp_offset};
        PhiloxSampler sm(dr);
        V3<R> c;
        if (form == FORM_ITERATIVE) c = ray_color_iter(sc, prm, ray, dr, cn);
        else if (prm.integrator == PT_INTEGRATOR_MIS) c = ray_color_mis_rec(sc, prm, ray, 0, sm, V3<R>::one(), cn);
        else c = ray_color_brdf_rec(sc, prm, ray, 0, sm, V3<R>::one(), cn);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
extern "C" void orc_ray_color(const PtObject* objs, uint32_t nobj, const PtRenderParams* pp, int precision, int form,
                              const double* rays, const uint32_t* xy, uint32_t n, double* out) {
    if (precision == 64) ray_color_batch<double>(objs, nobj, pp, form, rays, xy, n, out);
    else ray_color_batch<float>(objs, nobj, pp, form, rays, xy, n, out);
}

// sin/cos(2*pi*u) in the given arithmetic mode (pins the f32 polynomial against libm).
extern "C" void orc_sincos2pi(int precision, const double* u, uint32_t n, double* out_sc) {
    for (uint32_t i = 0; i < n; ++i) {
        if (precision == 64) { double s, c; Ar<double>::sincos2pi(u[i], s, c); out_sc[2 * i] = s; out_sc[2 * i + 1] = c; }
        else { float s, c; Ar<float>::sincos2pi((float)u[i], s, c); out_sc[2 * i] = s; out_sc[2 * i + 1] = c; }
    }
}

// Vector3 KAT surface (math.rs): op on doubles a,b, scalar s -> out[3], returns scalar result.
// ops: 0 add 1 sub 2 mul_s 3 mul_v 4 div_s 5 neg 6 dot 7 cross 8 length 9 normalize
//      10 normal_from_triangle(a,b,c) 11 reflect 12 refract (ret 0 = None) 13 face_forward
//      14 max 15 luminance 16 div_v 17 length_squared
extern "C" double orc_vec3(int precision, int op, const double* a, const double* b, const double* c, double s, double* out) {
    auto run = [&](auto tag) -> double {
        using R = decltype(tag);
        V3<R> A((R)a[0], (R)a[1], (R)a[2]), B, C, O;
        if (b) B = V3<R>((R)b[0], (R)b[1], (R)b[2]);
        if (c) C = V3<R>((R)c[0], (R)c[1], (R)c[2]);
        double ret = 1.0;
        switch (op) {
            case 0: O = A + B; break;
            case 1: O = A - B; break;
            case 2: O = A * (R)s; break;
            case 3: O = A * B; break;
            case 4: O = A / (R)s; break;
            case 5: O = -A; break;
            case 6: ret = (double)A.dot(B); break;
            case 7: O = A.cross(B); break;
            case 8: ret = (double)A.length(); break;
            case 9: O = A.normalize(); break;
            case 10: O = V3<R>::normal_from_triangle(A, B, C); break;
            case 11: O = A.reflect(B); break;
            case 12: ret = A.refract(B, (R)s, O) ? 1.0 : 0.0; break;
            case 13: O = A.face_forward(B); break;
            case 14: ret = (double)A.max(); break;
            case 15: ret = (double)A.luminance(); break;
            case 16: O = A.div_vec(B); break;
            case 17: ret = (double)A.length_squared(); break;
            default: ret = -1.0;
        }
        if (out) { out[0] = O.x; out[1] = O.y; out[2] = O.z; }
        return ret;
    };
    return precision == 64 ? run(double()) : run(float());
}

extern "C" uint32_t orc_tile_rows(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count) {
    return (uint32_t)tile_rows(height, band_rows, band_index, band_count).size();
}


// Debug: per-vertex records of one camera sample in the iterative form.
// rec (24 doubles per vertex): depth, obj, t, point3, normal3, beta3, [12] ls.pdf | pdf_prev,
// [13] visible | pdf_shape, [14] w_nee | w_bsdf, [15] pdf_bsdf, [16] light distance, [17] direct.x,
// [18] bsdf sample pdf, [19] rr, [20] u_rr, [21..23] L so far.  Returns the number of vertices.
extern "C" int orc_trace_path(const PtCamera* pc, const PtObject* objs, uint32_t n, const PtRenderParams* pp,
                              int precision, uint32_t x, uint32_t y, uint32_t sample, double* rec, int max_vertices) {
    Trace tr;
    Counters cn;
    Params prm = make_params(pp);
    Draws dr{{x, y}, sample};
    uint32_t dc[4];
    dr.block(DEPTH_CAMERA, 0, dc);
    if (precision == 64) {
        Scene<double> sc = build_scene<double>(objs, n);
        Camera<double> cam = make_camera<double>(pc);
        Ray<double> ray = cam.get_ray_with_offset(x, cam.height - 1 - y, u01(dc[0]), u01(dc[1]));
        ray_color_iter<double>(sc, prm, ray, dr, cn, &tr);
    } else {
        Scene<float> sc = build_scene<float>(objs, n);
        Camera<float> cam = make_camera<float>(pc);
        Ray<float> ray = cam.get_ray_with_offset(x, cam.height - 1 - y, (float)u01(dc[0]), (float)u01(dc[1]));
        ray_color_iter<float>(sc, prm, ray, dr, cn, &tr);
    }
    int nv = (int)(tr.rec.size() / 24);
    int m = nv < max_vertices ? nv : max_vertices;
    std::memcpy(rec, tr.rec.data(), (size_t)m * 24 * sizeof(double));
    return nv;
}

// World::render_pixel (world.rs:293-333) for a pixel list, with the radiance of every camera sample (what the
// reference's pixel diagnostics print, world.rs:378-417): xy n*2 = (x, y top-down film row); out_lin n*3 (may be
// null); out_samples n*spp*3 (may be null) in sample order.  The replay tool of the full-size parity tests.
template <class R>
static void render_pixels_impl(const PtCamera* pc, const PtObject* objs, uint32_t nobj, const PtRenderParams* pp, int form,
                               const uint32_t* xy, uint32_t n, double* out_lin, double* out_samples) {
    Scene<R> scene = build_scene<R>(objs, nobj);
    Camera<R> cam = make_camera<R>(pc);
    Params prm = make_params(pp);
    Counters cn;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t x = xy[2 * i], y = xy[2 * i + 1];
        double acc[3] = {0, 0, 0};
        for (uint32_t s = 0; s < pp->spp; ++s) {
            double lin[3]; uint8_t rg[4];
            render_pixel<R>(scene, cam, prm, (Form)form, x, y, 1, pp->spp_offset + s, lin, rg, cn);   // one sample: mean = the sample
            if (out_samples) std::memcpy(out_samples + ((size_t)i * pp->spp + s) * 3, lin, sizeof lin);
            for (int k = 0; k < 3; ++k) acc[k] += lin[k];
        }
        if (out_lin) for (int k = 0; k < 3; ++k) out_lin[3 * (size_t)i + k] = acc[k] / (double)pp->spp;
    }
}
extern "C" int orc_render_pixels(const PtCamera* cam, const PtObject* objs, uint32_t nobj, const PtRenderParams* p, int precision,
                                 int form, const uint32_t* xy, uint32_t n, double* out_lin, double* out_samples) {
    if (!cam || !objs || !p || p->spp == 0 || (n && !xy)) return 1;
    if (precision == 64) render_pixels_impl<double>(cam, objs, nobj, p, form, xy, n, out_lin, out_samples);
    else render_pixels_impl<float>(cam, objs, nobj, p, form, xy, n, out_lin, out_samples);
    return 0;
}

// ------------------------------------------------------------------ the reference's own draw source (StdRng restated)
// orc_render with one sequential ChaCha12 stream per pixel instead of Philox addressing: f64, recursive form.
// Restated from the published algorithms, unverified against the rand crate (pt_oracle.hpp, StdRngStream).
extern "C" int orc_render_stdrng(const PtCamera* pc, const PtObject* objs, uint32_t n, const PtRenderParams* pp, int threads,
                                 double* out_lin, uint8_t* out_rgba, uint64_t* out_counters) {
    if (!pc || !objs || !pp || pp->spp == 0) return 1;
    Scene<double> scene = build_scene<double>(objs, n);
    Camera<double> cam = make_camera<double>(pc);
    Params prm = make_params(pp);
    std::vector<uint32_t> rows = tile_rows(pc->height, pp->band_rows, pp->band_index, pp->band_count);
    const uint32_t W = pc->width;
    const size_t npix = rows.size() * (size_t)W;
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::vector<Counters> cns(threads);
    auto worker = [&](int tid) {
        for (;;) {
            const size_t b = next.fetch_add(16);
            if (b >= npix) break;
            const size_t e = b + 16 < npix ? b + 16 : npix;
            for (size_t i = b; i < e; ++i) {
                double lin[3]; uint8_t rg[4];
                render_pixel_stdrng(scene, cam, prm, (uint32_t)(i % W), rows[i / W], pp->spp, pp->spp_offset, lin, rg, cns[tid]);
                if (out_lin) { out_lin[i * 3] = lin[0]; out_lin[i * 3 + 1] = lin[1]; out_lin[i * 3 + 2] = lin[2]; }
                if (out_rgba) std::memcpy(out_rgba + i * 4, rg, 4);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (out_counters) {
        uint64_t v = 0, s = 0, sc = 0; uint32_t md = 0;
        for (auto& c : cns) { v += c.vertices; s += c.shadow_rays; sc += c.scans; if (c.max_depth > md) md = c.max_depth; }
        out_counters[0] = v; out_counters[1] = s; out_counters[2] = sc; out_counters[3] = md;
    }
    return 0;
}
// per-sample radiances of listed pixels from the sequential stream (the reference's pixel diagnostics, world.rs:378-417)
extern "C" int orc_render_pixels_stdrng(const PtCamera* pc, const PtObject* objs, uint32_t nobj, const PtRenderParams* pp,
                                        const uint32_t* xy, uint32_t n, double* out_lin, double* out_samples) {
    if (!pc || !objs || !pp || pp->spp == 0 || (n && !xy)) return 1;
    Scene<double> scene = build_scene<double>(objs, nobj);
    Camera<double> cam = make_camera<double>(pc);
    Params prm = make_params(pp);
    Counters cn;
    for (uint32_t i = 0; i < n; ++i) {
        double lin[3]; uint8_t rg[4];
        render_pixel_stdrng(scene, cam, prm, xy[2 * i], xy[2 * i + 1], pp->spp, pp->spp_offset, lin, rg, cn,
                            out_samples ? out_samples + (size_t)i * pp->spp * 3 : nullptr);
        if (out_lin) std::memcpy(out_lin + 3 * (size_t)i, lin, sizeof lin);
    }
    return 0;
}
// pins of the generator itself (tests/test_rng.py)
extern "C" void orc_chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) { chacha_block(in, rounds, out); }
extern "C" void orc_stdrng_seed_key(uint64_t seed, uint32_t key_out[8]) { StdRngStream r(seed); std::memcpy(key_out, r.key, sizeof r.key); }
// mode 0: n x next_u32 -> out32; 1: n x next_u64 -> out64; 2: n x random::<f64>() -> outf; 3: n x random_range(0..arg) -> out32;
// 4: alternating next_u32 (-> out32[i]) and next_u64 (-> out64[i]): buffer-straddling reads
extern "C" void orc_stdrng_draw(uint64_t seed, int rounds, int mode, uint32_t arg, uint32_t n, uint32_t* out32, uint64_t* out64, double* outf) {
    StdRngStream r(seed, rounds);
    for (uint32_t i = 0; i < n; ++i) {
        if (mode == 0) out32[i] = r.next_u32();
        else if (mode == 1) out64[i] = r.next_u64();
        else if (mode == 2) outf[i] = r.f64();
        else if (mode == 3) out32[i] = r.range(arg);
        else { out32[i] = r.next_u32(); out64[i] = r.next_u64(); }
    }
}
