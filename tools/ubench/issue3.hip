// micro-benchmark 3: one compare feeding several selects (vcc vs an SGPR pair), and exec-masked moves
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
// 2 compares, each feeding 3 selects (8 instructions)
KERN(k_vcc3, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", "vcc")
KERN(k_sgpr3, "v_cmp_lt_f32 s[10:11], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
              "v_cmp_lt_f32 s[12:13], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[12:13]\n v_cndmask_b32_e64 %6, %6, %8, s[12:13]\n v_cndmask_b32_e64 %7, %7, %8, s[12:13]\n", "s10", "s11", "s12", "s13")
// 1 compare feeding 7 selects
KERN(k_vcc7, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", "vcc")
// exec-masked moves: s_and_saveexec + 3 v_mov + restore (how a divergent `if` assigns 3 registers) x2
KERN(k_exec3, "v_cmp_lt_f32 vcc, %0, %8\n s_and_saveexec_b64 s[10:11], vcc\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n s_mov_b64 exec, s[10:11]\n"
              "v_cmp_lt_f32 vcc, %4, %8\n s_and_saveexec_b64 s[10:11], vcc\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n s_mov_b64 exec, s[10:11]\n", "vcc", "s10", "s11")
// select by arithmetic: mask = cmp ? ~0 : 0 via v_cndmask once, then v_bfi
KERN(k_bfi, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, 0, -1, vcc\n v_bfi_b32 %2, %1, %8, %2\n v_bfi_b32 %3, %1, %8, %3\n"
            "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, 0, -1, vcc\n v_bfi_b32 %6, %5, %8, %6\n v_bfi_b32 %7, %5, %8, %7\n", "vcc")
KERN(k_add8, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n", "memory")
// v_cndmask vcc with vcc written by a SALU instruction each time
KERN(k_vcc_salu, "s_mov_b64 vcc, exec\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                 "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", "vcc")
typedef void (*kern_t)(float*, int, float, uint32_t);
static double run(const char* name, kern_t k, int per_iter, double base) {
    static float* d = nullptr;
    if (!d) (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, grid = 2048;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f, 0x9E3779B9u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr = (double)grid * 4 * iters * per_iter;
    const double rate = instr / best / 1e9;
    printf("%-40s %8.3f ms  %7.3f T wave-instr/s  cost %.2f x v_add_f32 per VALU instruction\n", name, best, rate, base > 0 ? base / rate : 1.0);
    return rate;
}
int main() {
    const double f = run("v_add_f32", k_add8, 64, 0);
    run("cmp->vcc + 3 cndmask vcc", k_vcc3, 64, f);
    run("cmp->sgpr + 3 cndmask sgpr", k_sgpr3, 64, f);
    run("cmp->vcc + 7 cndmask vcc", k_vcc7, 64, f);
    run("s_mov vcc + 7 cndmask vcc", k_vcc_salu, 56, f);
    run("cmp + saveexec + 3 v_mov + restore", k_exec3, 64, f);
    run("cmp + cndmask(mask) + 2 v_bfi", k_bfi, 64, f);
    return 0;
}
