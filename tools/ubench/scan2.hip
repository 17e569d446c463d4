// micro-benchmark (round 4): would TWO paths per lane at 3 waves per SIMD beat ONE path per lane at 6 for the
// part of the headline kernel that dominates it -- the linear scan of C2's ten spheres out of LDS?  Two independent rays per
// lane share every LDS broadcast read and every scalar instruction of the scan loop and give the scheduler two dependency
// chains; three waves of 2 x 64 rays keep the same number of rays per SIMD as six waves of 64.
//   k_one : one ray per lane, launch_bounds(256, 6);  k_two : two rays per lane, launch_bounds(256, 3), half the workgroups.
// Each lane runs ITERS closest-hit scans + ITERS any-hit scans (the two scans of a vertex) on rays that change every iteration
// (so nothing hoists), over the C2 scene's scan records (pt_scenes.cpp scene 2), with the kernel's own arithmetic
// (sphere_pre / sphere_post of pt_kernels.hip, groups of four).  Output: ns per ray-scan.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/scan2.hip -o tools/ubench/scan2 && tools/ubench/scan2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 madd(f3 a, float s, f3 b) { return mk(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)); }

__device__ __forceinline__ void sphere_pre(float4 s, f3 o, f3 d, float& half_b, float& disc) {
    f3 oc = o - mk(s.x, s.y, s.z);
    half_b = dot(oc, d);
    f3 l = madd(d, -half_b, oc);
    disc = s.w - dot(l, l);
}
template <bool ANY>
__device__ __forceinline__ void sphere_post(float half_b, float disc, float t_min, float& closest, int& id, int obj) {
    if (disc < 0.0f) return;
    float sqrtd = __builtin_amdgcn_sqrtf(disc);
    float root1 = -half_b - sqrtd, root2 = -half_b + sqrtd;
    float c = root1 < t_min ? root2 : root1;
    if (c < t_min || closest < c) return;
    if (ANY) { id = 0; return; }
    closest = c; id = obj;
}
// one ray: the scan of k_paths_regen (groups of four spheres, then the rest)
template <bool ANY>
__device__ __forceinline__ void scan1(const float4* __restrict__ sc, int n, f3 o, f3 d, float t_max, float& t, int& id) {
    float closest = t_max; int hit = -1;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        float4 s0 = sc[i], s1 = sc[i + 1], s2 = sc[i + 2], s3 = sc[i + 3];
        float h0, h1, h2, h3, d0, d1, d2, d3;
        sphere_pre(s0, o, d, h0, d0); sphere_pre(s1, o, d, h1, d1); sphere_pre(s2, o, d, h2, d2); sphere_pre(s3, o, d, h3, d3);
        sphere_post<ANY>(h0, d0, 1e-3f, closest, hit, i); sphere_post<ANY>(h1, d1, 1e-3f, closest, hit, i + 1);
        sphere_post<ANY>(h2, d2, 1e-3f, closest, hit, i + 2); sphere_post<ANY>(h3, d3, 1e-3f, closest, hit, i + 3);
    }
    for (; i < n; ++i) { float h, dd; sphere_pre(sc[i], o, d, h, dd); sphere_post<ANY>(h, dd, 1e-3f, closest, hit, i); }
    t = closest; id = hit;
}
// two rays: every record is read once and tested against both
template <bool ANY>
__device__ __forceinline__ void scan2(const float4* __restrict__ sc, int n, f3 oa, f3 da, f3 ob, f3 db, float ta_max, float tb_max,
                                      float& ta, int& ia, float& tb, int& ib) {
    float ca = ta_max, cb = tb_max; int ha = -1, hb = -1;
    int i = 0;
    for (; i + 2 <= n; i += 2) {
        float4 s0 = sc[i], s1 = sc[i + 1];
        float h0, h1, h2, h3, d0, d1, d2, d3;
        sphere_pre(s0, oa, da, h0, d0); sphere_pre(s1, oa, da, h1, d1); sphere_pre(s0, ob, db, h2, d2); sphere_pre(s1, ob, db, h3, d3);
        sphere_post<ANY>(h0, d0, 1e-3f, ca, ha, i); sphere_post<ANY>(h1, d1, 1e-3f, ca, ha, i + 1);
        sphere_post<ANY>(h2, d2, 1e-3f, cb, hb, i); sphere_post<ANY>(h3, d3, 1e-3f, cb, hb, i + 1);
    }
    for (; i < n; ++i) {
        float4 s = sc[i]; float h, dd;
        sphere_pre(s, oa, da, h, dd); sphere_post<ANY>(h, dd, 1e-3f, ca, ha, i);
        sphere_pre(s, ob, db, h, dd); sphere_post<ANY>(h, dd, 1e-3f, cb, hb, i);
    }
    ta = ca; ia = ha; tb = cb; ib = hb;
}
__device__ __forceinline__ f3 next_dir(f3 d, float k) {        // a new unit direction every iteration (cheap, data-dependent)
    f3 v = mk(d.y + 0.37f * k, d.z - 0.21f, d.x + 0.11f * k);
    float inv = __builtin_amdgcn_rsqf(dot(v, v));
    return mk(v.x * inv, v.y * inv, v.z * inv);
}

__global__ void __launch_bounds__(256, 6) k_one(const float4* scene, int n, int iters, float* out) {
    __shared__ float4 sc[16];
    if (threadIdx.x < n) sc[threadIdx.x] = scene[threadIdx.x];
    __syncthreads();
    const float u = (float)((blockIdx.x * 256 + threadIdx.x) % 977) * (1.0f / 977.0f);
    f3 o = mk(-0.8f + 1.6f * u, -0.5f + u * 0.7f, -1.2f - u), d = next_dir(mk(u - 0.5f, 0.3f - u, -1.0f), 1.0f);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float t; int id;
        scan1<false>(sc, n, o, d, __builtin_huge_valf(), t, id);
        f3 so = id >= 0 ? madd(d, t, o) : o;
        f3 sd = next_dir(mk(-so.x, 0.79f - so.y, -2.0f - so.z), 0.0f);      // towards the light
        float t2; int id2;
        scan1<true>(sc, n, so, sd, 1.5f, t2, id2);
        acc += (float)id + (id2 < 0 ? t : 0.0f);
        d = next_dir(d, 1.0f + (float)(id & 3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256, 3) k_two(const float4* scene, int n, int iters, float* out) {
    __shared__ float4 sc[16];
    if (threadIdx.x < n) sc[threadIdx.x] = scene[threadIdx.x];
    __syncthreads();
    const float u = (float)((blockIdx.x * 512 + threadIdx.x) % 977) * (1.0f / 977.0f);
    const float w = (float)((blockIdx.x * 512 + 256 + threadIdx.x) % 977) * (1.0f / 977.0f);
    f3 oa = mk(-0.8f + 1.6f * u, -0.5f + u * 0.7f, -1.2f - u), da = next_dir(mk(u - 0.5f, 0.3f - u, -1.0f), 1.0f);
    f3 ob = mk(-0.8f + 1.6f * w, -0.5f + w * 0.7f, -1.2f - w), db = next_dir(mk(w - 0.5f, 0.3f - w, -1.0f), 1.0f);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float ta, tb; int ia, ib;
        scan2<false>(sc, n, oa, da, ob, db, __builtin_huge_valf(), __builtin_huge_valf(), ta, ia, tb, ib);
        f3 sa = ia >= 0 ? madd(da, ta, oa) : oa, sb = ib >= 0 ? madd(db, tb, ob) : ob;
        f3 sda = next_dir(mk(-sa.x, 0.79f - sa.y, -2.0f - sa.z), 0.0f), sdb = next_dir(mk(-sb.x, 0.79f - sb.y, -2.0f - sb.z), 0.0f);
        float t2a, t2b; int i2a, i2b;
        scan2<true>(sc, n, sa, sda, sb, sdb, 1.5f, 1.5f, t2a, i2a, t2b, i2b);
        acc += (float)ia + (i2a < 0 ? ta : 0.0f) + (float)ib + (i2b < 0 ? tb : 0.0f);
        da = next_dir(da, 1.0f + (float)(ia & 3)); db = next_dir(db, 1.0f + (float)(ib & 3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    // C2's scan records (center, r^2): five wall spheres R = 100, the light, four diffuse spheres (pt_scenes.cpp scene_cornell_spheres)
    const float R = 100.f;
    std::vector<float4> sc = {
        {-(1 + R), 0, -2, R * R}, {1 + R, 0, -2, R * R}, {0, 0, -3 - R, R * R}, {0, -(1 + R), -2, R * R}, {0, 1 + R, -2, R * R},
        {0, 0.79f, -2, 0.04f}, {-0.4f, -0.6f, -2, 0.16f}, {0.4f, -0.6f, -2, 0.16f}, {0, -0.8f, -1.5f, 0.04f}, {0, 0.1f, -2.4f, 0.0625f}};
    float4* d_sc; float* d_out;
    hipMalloc(&d_sc, sc.size() * sizeof(float4)); hipMemcpy(d_sc, sc.data(), sc.size() * sizeof(float4), hipMemcpyHostToDevice);
    const int cus = 256, iters = 2000;
    const int grid1 = cus * 6 * 4, grid2 = cus * 3 * 4;          // what the device holds at once, several rounds of it
    hipMalloc(&d_out, (size_t)grid1 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms1, ms2;
        hipEventRecord(e0); hipLaunchKernelGGL(k_one, dim3(grid1), dim3(256), 0, 0, d_sc, (int)sc.size(), iters, d_out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_two, dim3(grid2), dim3(256), 0, 0, d_sc, (int)sc.size(), iters, d_out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms2, e0, e1);
        const double rays1 = (double)grid1 * 256 * iters * 2, rays2 = (double)grid2 * 256 * 2 * iters * 2;       // ray-scans (closest + any)
        std::printf("one ray per lane, 6 waves/SIMD: %.3f ms = %.4f ns per ray-scan | two rays per lane, 3 waves/SIMD: %.3f ms = %.4f ns per ray-scan | ratio %.3f\n",
                    ms1, ms1 * 1e6 / rays1, ms2, ms2 * 1e6 / rays2, (ms2 / rays2) / (ms1 / rays1));
    }
    return 0;
}
