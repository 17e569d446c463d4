// mirror_check.cpp -- walks the C++ mirror of the reference's trait surface (pathtrace_amd/host/pathtrace.hpp), every
// call of which runs on the GPU, and prints one "name v0 v1 ..." line per result for tests/test_host_mirror.py to
// compare with the oracle.  Arithmetic mode: exact (bit-comparable with the f32 oracle).
#include <cstdio>

#include "../pathtrace_amd/host/pathtrace.hpp"

using namespace pathtrace;

static void p3(const char* name, const Vector3& v) { std::printf("%s %.9g %.9g %.9g\n", name, v.x, v.y, v.z); }

int main() {
    try {
        World world = World::new_();                       // 400 x 400, the reference scene
        world.params().spp = 4;
        world.params().exact_math = 1;
        // Camera::get_ray_with_offset (host f64) for the reference's diagnostic pixel (79, 176), row flipped as world.rs:299
        const Ray cam_ray = world.camera().get_ray_with_offset(79, HEIGHT - 1 - 176, 0.25, 0.75);
        p3("cam_ray_o", cam_ray.origin); p3("cam_ray_d", cam_ray.direction);
        // World::hit_scene
        auto hs = world.hit_scene(cam_ray, 0.001, INFINITY);
        if (hs) { std::printf("hit_scene %zu %.9g %d\n", hs->second, hs->first.t, (int)hs->first.front_face); p3("hit_point", hs->first.point); p3("hit_normal", hs->first.normal); }
        else std::printf("hit_scene none\n");
        // Shape::hit on single shapes
        const SphereShape sph = SphereShape::new_(Vector3(0.4, -0.6, -2.0), 0.4);
        const Ray r2 = Ray::new_(Vector3(0.0, 0.0, 2.0), Vector3(0.1, -0.15, -1.0));
        if (auto h = sph.hit(r2, 0.001, INFINITY, 1)) { std::printf("sphere_hit %.9g %d\n", h->t, (int)h->front_face); p3("sphere_hit_n", h->normal); }
        else std::printf("sphere_hit none\n");
        const TriangleShape tri = TriangleShape::new_(Vector3(-1, -1, -3), Vector3(1, -1, -3), Vector3(1, 1, -3));
        if (auto h = tri.hit(r2, 0.001, INFINITY, 1)) { std::printf("tri_hit %.9g %d\n", h->t, (int)h->front_face); p3("tri_hit_n", h->normal); }
        else std::printf("tri_hit none\n");
        // Shape::sample_surface_from_point, both forms
        HitRecord from; from.point = Vector3(0.2, -0.9, -1.5);
        auto ss = sph.sample_surface_from_point(from, nullptr, 0.3, 0.6, 1);
        p3("sphere_sample_p", ss.point); std::printf("sphere_sample_pdf %.9g %.9g\n", ss.pdf_omega, ss.distance);
        HitRecord tgt; tgt.point = ss.point;
        auto st = sph.sample_surface_from_point(from, &tgt, 0, 0, 1);
        std::printf("sphere_target_pdf %.9g\n", st.pdf_omega);
        // Material::bsdf_pdf / bsdf_pdf_sample / get_eta / emit
        const Mirror glass{0.3, Vector3(1, 1, 1), 0.0, 1.5};
        const Vector3 n = Vector3(0.2, 0.9, 0.1).normalize();
        Ray in = Ray::new_(Vector3(0, 1, 0), Vector3(0.3, -1.0, 0.2));
        in.set_eta_ratio(1.0 / glass.get_eta());
        auto ev = glass.bsdf_pdf(in, Vector3(-0.1, 0.8, 0.3).normalize(), n, 1);
        p3("glass_f", ev.first); std::printf("glass_pdf %.9g\n", ev.second);
        const uint32_t words[3] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u};
        auto bs = glass.bsdf_pdf_sample(in, n, words, 1);
        p3("glass_wo", bs.direction); p3("glass_sf", bs.bsdf); std::printf("glass_spdf %.9g %.9g\n", bs.pdf, bs.cos_theta);
        const LambertianCosineWeighted lam = LambertianCosineWeighted::new_(Vector3(0.8, 0.6, 0.2));
        auto ls = lam.bsdf_pdf_sample(in, n, words, 1);
        p3("lambert_wo", ls.direction); std::printf("lambert_spdf %.9g %.9g\n", ls.pdf, ls.cos_theta);
        std::printf("eta %.9g %.9g emit %.9g\n", glass.get_eta(), lam.get_eta(), Emissive::new_(Vector3(15, 15, 15)).emit().x);
        // World::sample_light_point
        if (hs) {
            const uint32_t lw[3] = {0xC0000000u, 0x40000000u, 0x80000000u};
            if (auto l = world.sample_light_point(hs->first, lw)) { p3("light_point", l->point); std::printf("light_pdf %.9g %u %.9g\n", l->pdf, l->light_object, l->emission.x); }
        }
        // World::render_pixel: the two pixels the reference's diagnostics replay (world.rs:378,531), and the full film
        const Color c1 = world.render_pixel(79, 176), c2 = world.render_pixel(10, 158);
        const Vector3 l1 = world.luminance_data[176 * WIDTH + 79], l2 = world.luminance_data[158 * WIDTH + 10];
        world.render();
        const Color f1 = world.data[176 * WIDTH + 79], f2 = world.data[158 * WIDTH + 10];
        std::printf("pixel_79_176 %u %u %u  film %u %u %u  same_linear %d\n", c1.r, c1.g, c1.b, f1.r, f1.g, f1.b,
                    (int)(l1.x == world.luminance_data[176 * WIDTH + 79].x && l1.y == world.luminance_data[176 * WIDTH + 79].y && l1.z == world.luminance_data[176 * WIDTH + 79].z));
        std::printf("pixel_10_158 %u %u %u  film %u %u %u  same_linear %d\n", c2.r, c2.g, c2.b, f2.r, f2.g, f2.b,
                    (int)(l2.x == world.luminance_data[158 * WIDTH + 10].x && l2.y == world.luminance_data[158 * WIDTH + 10].y && l2.z == world.luminance_data[158 * WIDTH + 10].z));
        p3("lum_79_176", l1);
        // RenderingStrategy::ray_color on the camera ray above, stream (79, 176), sample 2
        p3("ray_color", world.ray_color(cam_ray, 79, 176, 2));
        // export -> import round trip and the differ
        world.export_luminance("/tmp/pt_mirror_check_luminance.csv");
        World again = World::new_();
        const size_t n_px = again.import_luminance("/tmp/pt_mirror_check_luminance.csv");
        const LuminanceDiff d = compare_luminance(again.luminance_data, world.luminance_data, 0.5e-6 * (1.0 + 1e-6), 0.0, 1.0, 1e-6);   // "%.6f": half a unit of the 6th decimal (a tie such as 15/128 = 0.1171875 is exactly that far off)
        std::printf("roundtrip %zu %d %.3e\n", n_px, (int)d.pass, d.max_abs);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
