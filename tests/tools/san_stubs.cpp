// Link stubs for the sanitizer driver: the kernel launchers live in pt_kernels.hip (device code), which a host-only
// sanitizer build does not contain.  Nothing in san_driver.cpp reaches them (every caller needs a HIP device first).
#include <cstdlib>

#include "../../pathtrace_amd/csrc/pt_kernels.h"
namespace ptk {
void launch_scene_setup_exact(float4*, float4*, uint32_t, hipStream_t) { std::abort(); }
void launch_scene_setup_fast(float4*, float4*, uint32_t, hipStream_t) { std::abort(); }
uint32_t regen_blocks_per_cu_exact(const BounceArgs&) { return 0; }
uint32_t regen_blocks_per_cu_fast(const BounceArgs&) { return 0; }
void launch_paths_exact(const BounceArgs&, uint32_t, hipStream_t) { std::abort(); }
void launch_paths_fast(const BounceArgs&, uint32_t, hipStream_t) { std::abort(); }
void launch_resolve(const ResolveArgs&, hipStream_t) { std::abort(); }
void launch_debug_hit_exact(const SceneView&, uint32_t, const float*, uint32_t, float, float, float4*, int32_t*, float*, float*, hipStream_t) { std::abort(); }
void launch_debug_hit_fast(const SceneView&, uint32_t, const float*, uint32_t, float, float, float4*, int32_t*, float*, float*, hipStream_t) { std::abort(); }
void launch_debug_fn_exact(const DebugFnArgs&, hipStream_t) { std::abort(); }
void launch_debug_fn_fast(const DebugFnArgs&, hipStream_t) { std::abort(); }
void launch_film_pack(const float*, const uint8_t*, uint32_t, void*, hipStream_t) { std::abort(); }
void launch_film_unpack(const void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, float*, uint8_t*, hipStream_t) { std::abort(); }
}  // namespace ptk
