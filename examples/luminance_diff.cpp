// luminance_diff.cpp -- diff two files in the reference's luminance.csv format (World::export_luminance,
// src/world.rs:344-369) under the FP32 tolerance of SURVEY 8d (ii); no GPU involved.  With a CSV written by a real
// `cargo run` of the reference as <ref.csv>, the per-pixel bar cannot hold for a GPU film (the reference's RNG streams
// are ChaCha12, not this library's Philox): then read `mean_rel` and `rmse` -- both films estimate the same image.
// The number-for-number comparison is against the oracle driven by the reference's own draw source:
//     python tools/oracle_luminance.py --rng stdrng > oracle_stdrng.csv          (400 x 400 x 3000 spp, ~3 min on 8 cores)
//     ./luminance_diff oracle_stdrng.csv luminance.csv 1e-5 1e-5 0.999 1e-6
// (StdRng restated from the published algorithm, unverified against the rand crate: tools/oracle_luminance.py.)
//
//   ./luminance_diff <a.csv> <ref.csv> [abs_tol rel_tol frac mean_tol]      exit code 0 = within tolerance
#include <cstdio>
#include <cstdlib>

#include "../pathtrace_amd/host/pathtrace.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s a.csv ref.csv [abs_tol rel_tol frac mean_tol]\n", argv[0]); return 2; }
    try {
        uint32_t wa, ha, wb, hb;
        const auto a = pathtrace::World::read_luminance(argv[1], wa, ha);
        const auto b = pathtrace::World::read_luminance(argv[2], wb, hb);
        if (wa != wb || ha != hb) { std::fprintf(stderr, "sizes differ: %ux%u vs %ux%u\n", wa, ha, wb, hb); return 2; }
        const double abs_tol = argc > 3 ? std::atof(argv[3]) : 1e-3, rel_tol = argc > 4 ? std::atof(argv[4]) : 1e-2;
        const double frac = argc > 5 ? std::atof(argv[5]) : 0.995, mean_tol = argc > 6 ? std::atof(argv[6]) : 1e-3;
        const pathtrace::LuminanceDiff d = pathtrace::compare_luminance(a, b, abs_tol, rel_tol, frac, mean_tol);
        std::printf("{\"width\": %u, \"height\": %u, \"pixels_within\": %.6f, \"outside\": %zu, \"max_abs\": %.6g, \"rmse\": %.6g, "
                    "\"mean_a\": %.9g, \"mean_ref\": %.9g, \"mean_rel\": %.3e, \"pass\": %s}\n",
                    wa, ha, d.frac_within, d.outside, d.max_abs, d.rmse, d.mean_a, d.mean_b, d.mean_rel, d.pass ? "true" : "false");
        return d.pass ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;
    }
}
