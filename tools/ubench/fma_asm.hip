// micro-benchmark: scalar v_fma_f32 (inline asm, cannot be packed) issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n"
                         "v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, grid = 2048;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)grid * 4 * iters * 64;
        printf("scalar v_fma_f32: %.3f ms  %.2f T wave-instr/s  %.1f TFLOP/s\n", ms, instr / ms / 1e9, instr * 64 * 2 / ms / 1e9);
    }
    return 0;
}
