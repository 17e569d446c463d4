// cornell.cpp -- what the reference's main() does, minus the window (src/main.rs:39-67):
// build World::new(), render every pixel, export luminance.csv, and write a PPM of
// World.data instead of blitting it to a winit/pixels surface (out of scope).
//
//   ./cornell [width height spp [out_prefix [exact_math]]]        defaults: 400 400 64 cornell 0
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../pathtrace_amd/host/pathtrace.hpp"

using namespace pathtrace;

int main(int argc, char** argv) {
    const uint32_t w = argc > 2 ? (uint32_t)std::atoi(argv[1]) : WIDTH;
    const uint32_t h = argc > 2 ? (uint32_t)std::atoi(argv[2]) : HEIGHT;
    const uint32_t spp = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 64;
    const std::string prefix = argc > 4 ? argv[4] : "cornell";
    const uint32_t exact_math = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 0;
    try {
        // World::new() fixes 400x400; other sizes re-author the same scene with the same camera model
        World world = World::new_();
        if (w != WIDTH || h != HEIGHT) {
            // re-author the same 13 objects through the mirrored constructors for a camera of another size
            World resized(Camera::new_(Vector3(0.0, 0.0, 2.0), w, h, 1.0, 35.0));
            uint32_t n = 0;
            check(pt_builtin_scene(1, 0, nullptr, 0, &n));
            std::vector<PtObject> objs(n);
            check(pt_builtin_scene(1, 0, objs.data(), n, &n));
            world = std::move(resized);
            for (const PtObject& o : objs) {
                if (o.shape_tag == PT_SHAPE_TRIANGLE) {
                    TriangleShape t = TriangleShape::new_(Vector3(o.shape[0], o.shape[1], o.shape[2]),
                                                          Vector3(o.shape[3], o.shape[4], o.shape[5]),
                                                          Vector3(o.shape[6], o.shape[7], o.shape[8]));
                    if (o.mat_tag == PT_MAT_EMISSIVE) world.push(Object::new_(t, Emissive::new_(Vector3(o.mat[0], o.mat[1], o.mat[2]))));
                    else world.push(Object::new_(t, LambertianCosineWeighted::new_(Vector3(o.mat[0], o.mat[1], o.mat[2]))));
                } else {
                    world.push(Object::new_(SphereShape::new_(Vector3(o.shape[0], o.shape[1], o.shape[2]), o.shape[3]),
                                            Mirror{o.mat[0], Vector3(o.mat[1], o.mat[2], o.mat[3]), o.mat[4], o.mat[5]}));
                }
            }
        }
        world.params().spp = spp;
        world.params().exact_math = exact_math;
        const auto t0 = std::chrono::steady_clock::now();
        if (std::getenv("CORNELL_PROGRESSIVE"))      // live-preview form: one line per increment
            world.render_progressive(std::max(1u, spp / 4), [&](uint32_t done) {
                std::printf("  preview after %u spp: centre pixel rgb = %u %u %u\n", done, world.data[(h / 2) * w + w / 2].r,
                            world.data[(h / 2) * w + w / 2].g, world.data[(h / 2) * w + w / 2].b);
                return false; });
        else
            world.render();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const PtStats st = world.stats();
        std::printf("Rendering complete: %ux%u, %u spp, %zu objects, %.3f s (%.1f Msamples/s), %llu vertices\n", w, h, spp,
                    world.object_count(), dt, (double)st.samples / dt / 1e6, (unsigned long long)st.vertices);
        world.export_luminance(prefix + "_luminance.csv");     // main.rs:64-66
        world.write_ppm(prefix + ".ppm");
        std::printf("wrote %s_luminance.csv and %s.ppm\n", prefix.c_str(), prefix.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
