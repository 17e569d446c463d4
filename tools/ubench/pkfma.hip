// micro-benchmark: v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = b0 + 1.f, b5 = b1 + 1.f, b6 = b2 + 1.f, b7 = b3 + 1.f;
    float2v sv = {s, s * 0.5f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a0 = __builtin_fmaf(a0, s, 1.f); a1 = __builtin_fmaf(a1, s, 1.f); a2 = __builtin_fmaf(a2, s, 1.f); a3 = __builtin_fmaf(a3, s, 1.f);
                a4 = __builtin_fmaf(a4, s, 1.f); a5 = __builtin_fmaf(a5, s, 1.f); a6 = __builtin_fmaf(a6, s, 1.f); a7 = __builtin_fmaf(a7, s, 1.f);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                b0 = __builtin_elementwise_fma(b0, sv, sv); b1 = __builtin_elementwise_fma(b1, sv, sv); b2 = __builtin_elementwise_fma(b2, sv, sv); b3 = __builtin_elementwise_fma(b3, sv, sv);
                b4 = __builtin_elementwise_fma(b4, sv, sv); b5 = __builtin_elementwise_fma(b5, sv, sv); b6 = __builtin_elementwise_fma(b6, sv, sv); b7 = __builtin_elementwise_fma(b7, sv, sv);
            }
        }
    }
    float r = MODE == 0 ? a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 : b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y + b4.x + b4.y + b5.x + b5.y + b6.x + b6.y + b7.x + b7.y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 2048;
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)grid * 4 * iters * 64;   // wave-instructions
        double flops = (double)grid * 256 * iters * 64 * 2 * (mode ? 2 : 1);
        printf("mode %s: %.3f ms  %.2f T wave-instr/s  %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms, instr / ms / 1e9, flops / ms / 1e9);
    }
    return 0;
}
