// micro-benchmark 4: is the cost of v_cndmask reading VCC a throughput or a latency cost, and which encodings pay it?
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERN(NAME, ASM, ...)                                                                                         \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float s, uint32_t u) {                        \
        float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; ++i) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                          \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(u) : __VA_ARGS__); \
            }                                                                                                        \
        }                                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
    }
KERN(k_add8, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n", "memory")
// cmp -> vcc, then 3 selects (vcc, e32) each followed by... nothing: 4 VALU
KERN(k_vcc3, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n", "vcc")
// the same with the e64 encoding naming vcc
KERN(k_vcc3_e64, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n", "vcc")
// the same with independent adds between the selects: 4 + 4 VALU
KERN(k_vcc3_mix, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_cndmask_b32 %3, %3, %8, vcc\n", "vcc")
// copy vcc to an SGPR pair first (1 SALU), then e64 selects
KERN(k_vcc_copy, "v_cmp_lt_f32 vcc, %0, %8\n s_mov_b64 s[10:11], vcc\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n", "vcc", "s10", "s11")
KERN(k_sgpr3, "v_cmp_lt_f32 s[10:11], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n", "s10", "s11")
// one select per compare (what scalar code mostly has)
KERN(k_vcc1, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n", "vcc")
// two selects per compare
KERN(k_vcc2, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n", "vcc")
// sign flip by xor instead of a 3-component select
KERN(k_xor3, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %4, 0, %9, vcc\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n", "vcc")
typedef void (*kern_t)(float*, int, float, uint32_t);
static void run(const char* name, kern_t k, int valu_per_group, int groups, int grid, int block, double add_ms_per_valu) {
    static float* d = nullptr;
    if (!d) (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, iters, 0.999f, 0x80000000u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double waves_per_simd = (double)grid * (block / 64) / 1024.0;
    // time of one group of the asm block per wave, in units of one v_add_f32 at full occupancy
    const double per_group = best / (iters * 8.0 * waves_per_simd);
    printf("%-44s grid %5d x %3d  %8.3f ms  one group = %.2f v_add_f32 slots (%d VALU)\n", name, grid, block, best,
           add_ms_per_valu > 0 ? per_group / add_ms_per_valu : 0.0, valu_per_group);
    (void)groups;
}
int main() {
    static float* d = nullptr; (void)hipMalloc(&d, 4096 * 256 * 4);
    // calibrate: ms per (v_add_f32 x wave) per SIMD at 8 waves/SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_add8, dim3(2048), dim3(256), 0, 0, d, 4000, 0.999f, 0u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double slot = best / (4000 * 8.0 * 8.0) / 8.0;     // per VALU instruction (8 per group), per wave, 8 waves per SIMD
    printf("v_add_f32 slot: %.3e ms\n", slot);
    for (int occ = 0; occ < 2; ++occ) {
        const int grid = occ ? 256 : 2048, block = occ ? 256 : 256;      // 8 waves/SIMD, then 1 wave/SIMD
        printf("---- %d wave(s) per SIMD\n", occ ? 1 : 8);
        run("8 v_add_f32", k_add8, 8, 1, grid, block, slot);
        run("cmp->vcc, 1 cndmask (x2)", k_vcc1, 4, 1, grid, block, slot);
        run("cmp->vcc, 2 cndmask (x2)", k_vcc2, 6, 1, grid, block, slot);
        run("cmp->vcc, 3 cndmask e32", k_vcc3, 4, 1, grid, block, slot);
        run("cmp->vcc, 3 cndmask e64 naming vcc", k_vcc3_e64, 4, 1, grid, block, slot);
        run("cmp->vcc, 3 cndmask + 4 adds interleaved", k_vcc3_mix, 8, 1, grid, block, slot);
        run("cmp->vcc, s_mov to sgpr, 3 cndmask e64", k_vcc_copy, 4, 1, grid, block, slot);
        run("cmp->sgpr, 3 cndmask e64", k_sgpr3, 4, 1, grid, block, slot);
        run("cmp->vcc, 1 cndmask mask, 3 xor", k_xor3, 5, 1, grid, block, slot);
    }
    return 0;
}
