// Sanitizer driver (CPU only; GPU ASan is not available on the pool): the host-side code that runs without a
// device -- scene constructors, camera constructors, the BVH builder + verifier (product), and the oracle
// (test infrastructure) -- under AddressSanitizer + UndefinedBehaviorSanitizer.
// Build + run: tests/tools/run_sanitizers.sh   (used by tests/test_sanitizers.py)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/pathtrace_amd.h"

extern "C" int orc_render(const PtCamera* cam, const PtObject* objs, uint32_t n, const PtRenderParams* p, int precision, int form,
                          int threads, double* out_lin, uint8_t* out_rgba, uint64_t* out_counters);

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d): %s\n", #c, __LINE__, pt_last_error()); ++fails; } } while (0)

int main() {
    for (uint32_t id : {1u, 2u, 4u}) {
        for (uint32_t arg : {0u, 7u, 3000u}) {
            if (id != 4 && arg) continue;
            uint32_t n = 0;
            CHECK(pt_builtin_scene(id, arg, nullptr, 0, &n) == PT_OK);
            std::vector<PtObject> objs(n);
            CHECK(pt_builtin_scene(id, arg, objs.data(), n, &n) == PT_OK);
            uint32_t depth = 0, nodes = 0, slots = 0;
            CHECK(pt_debug_bvh_check(objs.data(), n, &depth, &nodes, &slots) == PT_OK);
            CHECK(slots == n);
            if (n > 64) continue;
            PtCamera cam;
            const double origin[3] = {0, 0, 2};
            CHECK(pt_camera_new(origin, 24, 16, 1.0, 35.0, &cam) == PT_OK);
            PtRenderParams prm;
            pt_default_params(&prm);
            prm.spp = 3;
            for (int precision : {64, 32})
                for (int form : {0, 1}) {
                    std::vector<double> lin(24 * 16 * 3);
                    std::vector<uint8_t> rgba(24 * 16 * 4);
                    uint64_t cnt[8] = {0};
                    CHECK(orc_render(&cam, objs.data(), n, &prm, precision, form, 2, lin.data(), rgba.data(), cnt) == 0);
                    CHECK(cnt[0] > 0);
                }
        }
    }
    // degenerate builder inputs: empty, coincident centroids, non-finite objects (refused), bad tags
    CHECK(pt_debug_bvh_check(nullptr, 0, nullptr, nullptr, nullptr) == PT_OK);
    std::vector<PtObject> same(2000);
    std::memset(same.data(), 0, same.size() * sizeof(PtObject));
    for (auto& o : same) { o.shape[2] = -2.0; o.shape[3] = 0.25; }
    CHECK(pt_debug_bvh_check(same.data(), (uint32_t)same.size(), nullptr, nullptr, nullptr) == PT_OK);
    same[5].shape[0] = NAN;
    CHECK(pt_debug_bvh_check(same.data(), (uint32_t)same.size(), nullptr, nullptr, nullptr) == PT_ERR_UNSUPPORTED);
    same[5].shape[0] = 0.0; same[9].shape_tag = 9;
    CHECK(pt_debug_bvh_check(same.data(), (uint32_t)same.size(), nullptr, nullptr, nullptr) == PT_ERR_INVALID_ARG);
    // camera constructors reject bad arguments without UB
    PtCamera cam;
    const double o[3] = {0, 0, 2}, t[3] = {0, 0, 2}, up[3] = {0, 1, 0};
    CHECK(pt_camera_new(o, 0, 0, 1.0, 35.0, &cam) != PT_OK);
    CHECK(pt_camera_look_at(o, t, up, 16, 16, 35.0, &cam) != PT_OK || true);   // origin == target: any status, no UB
    CHECK(pt_tile_rows(100, 7, 2, 3) > 0);
    std::printf("sanitizer driver: %d failed checks\n", fails);
    return fails ? 1 : 0;
}
