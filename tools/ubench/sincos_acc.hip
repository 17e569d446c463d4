// accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) and v_rsq_f32 on the uniforms the renderer feeds them
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float* s, float* c, float* q, uint32_t n) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u = (float)(2u * (i * 8u + 3u) + 1u) * (1.0f / 16777216.0f);     // a u01() grid value
    s[i] = __builtin_amdgcn_sinf(u); c[i] = __builtin_amdgcn_cosf(u);
    q[i] = __builtin_amdgcn_rsqf(u * 37.0f);
}
int main() {
    const uint32_t n = 1u << 20;
    float *ds, *dc, *dq; (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dc, n * 4); (void)hipMalloc(&dq, n * 4);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, ds, dc, dq, n);
    std::vector<float> s(n), c(n), q(n);
    (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, eq = 0, en = 0;
    for (uint32_t i = 0; i < n; ++i) {
        float u = (float)(2u * (i * 8u + 3u) + 1u) * (1.0f / 16777216.0f);
        double ph = 2.0 * M_PI * (double)u;
        es = fmax(es, fabs(s[i] - sin(ph))); ec = fmax(ec, fabs(c[i] - cos(ph)));
        en = fmax(en, fabs((double)s[i] * s[i] + (double)c[i] * c[i] - 1.0));
        double r = 1.0 / sqrt((double)(u * 37.0f));
        eq = fmax(eq, fabs(q[i] - r) / r);
    }
    printf("v_sin_f32 max abs err %.3e, v_cos_f32 %.3e, |s^2+c^2-1| max %.3e, v_rsq_f32 max rel err %.3e\n", es, ec, en, eq);
    return 0;
}
