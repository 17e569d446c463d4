// micro-benchmark 2: selects, compares and masks (what the divergent parts of the path kernel are made of)
#include <hip/hip_runtime.h>
#include <cstdio>
#define K8(NAME, PRE, I0, I1, I2, I3, I4, I5, I6, I7, CLOB)                                                          \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float s, uint32_t u) {                        \
        float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        asm volatile(PRE ::"v"(a0), "v"(s) : "vcc", "s10", "s11");                                                   \
        for (int i = 0; i < iters; ++i) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                          \
                asm volatile(I0 I1 I2 I3 I4 I5 I6 I7                                                                 \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(u) : CLOB); \
            }                                                                                                        \
        }                                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
    }
#define R8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define CND_VCC(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n",
#define CND_S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n",
K8(k_cnd_vcc, "v_cmp_lt_f32 vcc, %0, %1\n", "v_cndmask_b32 %0, %0, %8, vcc\n", "v_cndmask_b32 %1, %1, %8, vcc\n", "v_cndmask_b32 %2, %2, %8, vcc\n",
   "v_cndmask_b32 %3, %3, %8, vcc\n", "v_cndmask_b32 %4, %4, %8, vcc\n", "v_cndmask_b32 %5, %5, %8, vcc\n", "v_cndmask_b32 %6, %6, %8, vcc\n",
   "v_cndmask_b32 %7, %7, %8, vcc\n", "memory")
K8(k_cnd_sgpr, "v_cmp_lt_f32 s[10:11], %0, %1\n", "v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n", "v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n",
   "v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n", "v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n", "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n",
   "v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n", "v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n", "v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n", "memory")
// compare + select pairs, as the compiler emits them
K8(k_cmp_cnd, "", "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n", "v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n",
   "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n", "v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n",
   "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n", "v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n",
   "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n", "v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc\n", "vcc")
K8(k_min, "", "v_min_f32 %0, %0, %8\n", "v_min_f32 %1, %1, %8\n", "v_min_f32 %2, %2, %8\n", "v_min_f32 %3, %3, %8\n", "v_min_f32 %4, %4, %8\n",
   "v_min_f32 %5, %5, %8\n", "v_min_f32 %6, %6, %8\n", "v_min_f32 %7, %7, %8\n", "memory")
K8(k_add, "", "v_add_f32 %0, %0, %8\n", "v_add_f32 %1, %1, %8\n", "v_add_f32 %2, %2, %8\n", "v_add_f32 %3, %3, %8\n", "v_add_f32 %4, %4, %8\n",
   "v_add_f32 %5, %5, %8\n", "v_add_f32 %6, %6, %8\n", "v_add_f32 %7, %7, %8\n", "memory")
K8(k_fmac, "", "v_fmac_f32 %0, %8, %8\n", "v_fmac_f32 %1, %8, %8\n", "v_fmac_f32 %2, %8, %8\n", "v_fmac_f32 %3, %8, %8\n", "v_fmac_f32 %4, %8, %8\n",
   "v_fmac_f32 %5, %8, %8\n", "v_fmac_f32 %6, %8, %8\n", "v_fmac_f32 %7, %8, %8\n", "memory")
K8(k_and, "", "v_and_b32 %0, %0, %9\n", "v_and_b32 %1, %1, %9\n", "v_and_b32 %2, %2, %9\n", "v_and_b32 %3, %3, %9\n", "v_and_b32 %4, %4, %9\n",
   "v_and_b32 %5, %5, %9\n", "v_and_b32 %6, %6, %9\n", "v_and_b32 %7, %7, %9\n", "memory")
K8(k_lshl, "", "v_lshlrev_b32 %0, 3, %0\n", "v_lshlrev_b32 %1, 3, %1\n", "v_lshlrev_b32 %2, 3, %2\n", "v_lshlrev_b32 %3, 3, %3\n", "v_lshlrev_b32 %4, 3, %4\n",
   "v_lshlrev_b32 %5, 3, %5\n", "v_lshlrev_b32 %6, 3, %6\n", "v_lshlrev_b32 %7, 3, %7\n", "memory")
K8(k_fma3, "", "v_fma_f32 %0, %0, %8, %1\n", "v_fma_f32 %1, %1, %8, %2\n", "v_fma_f32 %2, %2, %8, %3\n", "v_fma_f32 %3, %3, %8, %4\n", "v_fma_f32 %4, %4, %8, %5\n",
   "v_fma_f32 %5, %5, %8, %6\n", "v_fma_f32 %6, %6, %8, %7\n", "v_fma_f32 %7, %7, %8, %0\n", "memory")
K8(k_max3, "", "v_max3_f32 %0, %0, %8, %1\n", "v_max3_f32 %1, %1, %8, %2\n", "v_max3_f32 %2, %2, %8, %3\n", "v_max3_f32 %3, %3, %8, %4\n", "v_max3_f32 %4, %4, %8, %5\n",
   "v_max3_f32 %5, %5, %8, %6\n", "v_max3_f32 %6, %6, %8, %7\n", "v_max3_f32 %7, %7, %8, %0\n", "memory")
K8(k_cmpx, "", "v_cmp_lt_f32 s[10:11], %0, %8\n", "v_cmp_lt_f32 s[10:11], %1, %8\n", "v_cmp_lt_f32 s[10:11], %2, %8\n", "v_cmp_lt_f32 s[10:11], %3, %8\n",
   "v_cmp_lt_f32 s[10:11], %4, %8\n", "v_cmp_lt_f32 s[10:11], %5, %8\n", "v_cmp_lt_f32 s[10:11], %6, %8\n", "v_cmp_lt_f32 s[10:11], %7, %8\n", "s10")
K8(k_xor3, "", "v_xor_b32 %0, %0, %9\n", "v_xor_b32 %1, %1, %9\n", "v_xor_b32 %2, %2, %9\n", "v_xor_b32 %3, %3, %9\n", "v_xor_b32 %4, %4, %9\n",
   "v_xor_b32 %5, %5, %9\n", "v_xor_b32 %6, %6, %9\n", "v_xor_b32 %7, %7, %9\n", "memory")
K8(k_mulsub, "", "v_mul_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n", "v_mul_f32 %1, %1, %8\n v_sub_f32 %1, %1, %8\n", "v_mul_f32 %2, %2, %8\n v_sub_f32 %2, %2, %8\n",
   "v_mul_f32 %3, %3, %8\n v_sub_f32 %3, %3, %8\n", "v_mul_f32 %4, %4, %8\n v_sub_f32 %4, %4, %8\n", "v_mul_f32 %5, %5, %8\n v_sub_f32 %5, %5, %8\n",
   "v_mul_f32 %6, %6, %8\n v_sub_f32 %6, %6, %8\n", "v_mul_f32 %7, %7, %8\n v_sub_f32 %7, %7, %8\n", "memory")
// what hipcc makes of selects written in C
__global__ void __launch_bounds__(256) k_c_select(float* out, int iters, float s, uint32_t u) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + 1.5f + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = a[k] < s * (float)(r + 1) ? a[k] * 1.0001f : a[k] + 0.5f;
    }
    float t = 0; for (int k = 0; k < 8; ++k) t += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
typedef void (*kern_t)(float*, int, float, uint32_t);
static double run(const char* name, kern_t k, int per_iter, double fma_rate) {
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
    printf("%-28s %8.3f ms  %7.3f T wave-instr/s  cost %.2f x v_add_f32 per instruction\n", name, best, rate, fma_rate > 0 ? fma_rate / rate : 1.0);
    return rate;
}
int main() {
    const double f = run("v_add_f32", k_add, 64, 0);
    run("v_fmac_f32", k_fmac, 64, f);
    run("v_fma_f32 3 distinct srcs", k_fma3, 64, f);
    run("v_min_f32", k_min, 64, f);
    run("v_max3_f32", k_max3, 64, f);
    run("v_and_b32", k_and, 64, f);
    run("v_xor_b32", k_xor3, 64, f);
    run("v_lshlrev_b32", k_lshl, 64, f);
    run("v_cmp_lt_f32 -> sgpr pair", k_cmpx, 64, f);
    run("v_cndmask vcc (set once)", k_cnd_vcc, 64, f);
    run("v_cndmask sgpr (set once)", k_cnd_sgpr, 64, f);
    run("v_cmp + v_cndmask pairs", k_cmp_cnd, 128, f);
    run("v_mul + v_sub pairs", k_mulsub, 128, f);
    run("C select (cmp+cnd+mul+add)", k_c_select, 64 * 4, f);
    return 0;
}
