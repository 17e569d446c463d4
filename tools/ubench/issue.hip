// micro-benchmark: issue cost of the instruction kinds the path kernel is made of, relative to v_fma_f32 (gfx950).
// Each kernel runs 8 independent chains of one instruction, 2048 workgroups x 256 threads, 8 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue.hip -o tools/ubench/issue && tools/ubench/issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY8(INS)                                                                                                   \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                             \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(u))

#define KERNEL(NAME, INS)                                                                                            \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float s, uint32_t u) {                        \
        float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; ++i) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) { BODY8(INS); }                                            \
        }                                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
    }

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, 1.0\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %9\n"
#define I_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %9\n"
#define I_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %9\n"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_RSQ(k) "v_rsq_f32 %" #k ", %" #k "\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_CVT(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define I_ALIGN(k) "v_alignbit_b32 %" #k ", %" #k ", %" #k ", 13\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"

KERNEL(k_fma, I_FMA)
KERNEL(k_mul, I_MUL)
KERNEL(k_xor, I_XOR)
KERNEL(k_addu, I_ADDU)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_mulhi, I_MULHI)
KERNEL(k_mul24, I_MUL24)
KERNEL(k_sqrt, I_SQRT)
KERNEL(k_rcp, I_RCP)
KERNEL(k_rsq, I_RSQ)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_cmp, I_CMP)
KERNEL(k_max, I_MAX)
KERNEL(k_cvt, I_CVT)
KERNEL(k_align, I_ALIGN)
KERNEL(k_mov, I_MOV)

// v_mad_u64_u32 (64-bit result: lo and hi of a 32 x 32 product in one instruction, what Philox compiles to)
__global__ void __launch_bounds__(256) k_mad64(float* out, int iters, float s, uint32_t u) {
    unsigned long long a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n v_mad_u64_u32 %1, vcc, %8, %8, %1\n v_mad_u64_u32 %2, vcc, %8, %8, %2\n"
                         "v_mad_u64_u32 %3, vcc, %8, %8, %3\n v_mad_u64_u32 %4, vcc, %8, %8, %4\n v_mad_u64_u32 %5, vcc, %8, %8, %5\n"
                         "v_mad_u64_u32 %6, vcc, %8, %8, %6\n v_mad_u64_u32 %7, vcc, %8, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
// ds_read_b128, all lanes the same address (the scene broadcast of the path kernel)
__global__ void __launch_bounds__(256) k_lds(float* out, int iters, float s, uint32_t u) {
    __shared__ float4 buf[64];
    if (threadIdx.x < 64) buf[threadIdx.x] = make_float4(s, s, s, s);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n" : "=v"(v) : "v"(0u), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc.x += v.x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x;
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
    const double instr = (double)grid * 4 * iters * per_iter;          // wave-instructions
    const double rate = instr / best / 1e9;                             // T wave-instr / s
    printf("%-16s %8.3f ms  %7.3f T wave-instr/s  cost %.2f x v_fma_f32\n", name, best, rate, fma_rate > 0 ? fma_rate / rate : 1.0);
    return rate;
}
int main() {
    const double f = run("v_fma_f32", k_fma, 64, 0);
    run("v_mul_f32", k_mul, 64, f);
    run("v_max_f32", k_max, 64, f);
    run("v_mov_b32", k_mov, 64, f);
    run("v_xor_b32", k_xor, 64, f);
    run("v_add_u32", k_addu, 64, f);
    run("v_alignbit_b32", k_align, 64, f);
    run("v_cndmask_b32", k_cndmask, 64, f);
    run("v_cmp_lt_f32", k_cmp, 64, f);
    run("v_cvt_f32_u32", k_cvt, 64, f);
    run("v_mul_u32_u24", k_mul24, 64, f);
    run("v_mul_lo_u32", k_mullo, 64, f);
    run("v_mul_hi_u32", k_mulhi, 64, f);
    run("v_mad_u64_u32", k_mad64, 64, f);
    run("v_sqrt_f32", k_sqrt, 64, f);
    run("v_rcp_f32", k_rcp, 64, f);
    run("v_rsq_f32", k_rsq, 64, f);
    run("ds_read_b128 bc", k_lds, 64, f);
    return 0;
}
