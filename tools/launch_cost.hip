// What does a launch that finds nothing to do cost on MI355X, as a function of grid size and of how fat the kernel is?
// (round 3: k_msm_combine_* take 15-27 us per launch with 256-512 workgroups that all exit at once.)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zkevm-circuits_amd/csrc tools/launch_cost.hip -o tools/launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ff29.hip.hpp"
#include "ec29.hip.hpp"
using namespace zk;

__global__ void __launch_bounds__(256) k_thin(const uint32_t* flag, uint32_t* out) {
    if (*flag) out[blockIdx.x * 256 + threadIdx.x] = threadIdx.x;
}
__global__ void __launch_bounds__(256) k_fat(const uint32_t* flag, G1Xyzz29* pts) {
    const uint32_t n = *flag;
    if (!n) return;
    G1Xyzz29 acc = ldg29(pts + threadIdx.x);
    for (uint32_t i = 0; i < n; ++i) acc = add29pt(acc, ldg29(pts + ((threadIdx.x + i) & 255)));
    stg29(pts + threadIdx.x, acc);
}
__global__ void __launch_bounds__(256) k_fat_lds(const uint32_t* flag, G1Xyzz29* pts) {
    __shared__ G1Xyzz29 sh[256];
    const uint32_t n = *flag;
    if (!n) return;
    sh[threadIdx.x] = ldg29(pts + threadIdx.x);
    __syncthreads();
    G1Xyzz29 acc = sh[threadIdx.x ^ 1];
    for (uint32_t i = 0; i < n; ++i) acc = add29pt(acc, sh[(threadIdx.x + i) & 255]);
    stg29(pts + threadIdx.x, acc);
}
// the same early exit taken by lane 0 of wave 0 only after a second dependent load (what combine_wave does: *nmulti, then toff[M])
__global__ void __launch_bounds__(256) k_fat_2loads(const uint32_t* flag, const uint32_t* tab, G1Xyzz29* pts) {
    const uint32_t m = *flag;
    const uint32_t n = m ? tab[m] : 0u;
    if (!n) return;
    G1Xyzz29 acc = ldg29(pts + threadIdx.x);
    for (uint32_t i = 0; i < n; ++i) acc = add29pt(acc, ldg29(pts + ((threadIdx.x + i) & 255)));
    stg29(pts + threadIdx.x, acc);
}
template <typename F> static float time_launches(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    uint32_t* flag; uint32_t* out; G1Xyzz29* pts; uint32_t* tab;
    hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
    hipMalloc(&tab, 4096); hipMemset(tab, 0, 4096);
    hipMalloc(&out, 4u << 20); hipMalloc(&pts, sizeof(G1Xyzz29) * 256); hipMemset(pts, 0, sizeof(G1Xyzz29) * 256);
    const int grids[] = {1, 16, 64, 128, 256, 512, 1024, 2048, 3158};
    printf("%8s %10s %10s %10s %10s   (us per launch, back to back on one stream, nothing to do)\n", "blocks", "thin", "fat", "fat+lds", "fat2loads");
    for (int g : grids) {
        const float t0 = time_launches([&] { hipLaunchKernelGGL(k_thin, dim3(g), dim3(256), 0, 0, flag, out); }, 300);
        const float t1 = time_launches([&] { hipLaunchKernelGGL(k_fat, dim3(g), dim3(256), 0, 0, flag, pts); }, 300);
        const float t2 = time_launches([&] { hipLaunchKernelGGL(k_fat_lds, dim3(g), dim3(256), 0, 0, flag, pts); }, 300);
        const float t3 = time_launches([&] { hipLaunchKernelGGL(k_fat_2loads, dim3(g), dim3(256), 0, 0, flag, tab, pts); }, 300);
        printf("%8d %10.2f %10.2f %10.2f %10.2f\n", g, t0, t1, t2, t3);
    }
    // alternating thin / fat (the instruction cache sees another kernel in between, as in the MSM chain)
    for (int g : {64, 256, 512}) {
        const float t = time_launches([&] { hipLaunchKernelGGL(k_thin, dim3(1024), dim3(256), 0, 0, flag, out); hipLaunchKernelGGL(k_fat, dim3(g), dim3(256), 0, 0, flag, pts); }, 300);
        printf("thin(1024) + fat(%d): %.2f us per pair\n", g, t);
    }
    // a no-op kernel right behind a kernel that left ~75 MB of dirty lines in the L2s (what k_msm_buckets does)
    {
        uint32_t* big; hipMalloc(&big, 80u << 20);
        hipEvent_t e[5]; for (auto& x : e) hipEventCreate(&x);
        float acc[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 50; ++rep) {
            hipLaunchKernelGGL(k_thin, dim3(1), dim3(256), 0, 0, flag, out);
            hipMemsetAsync(flag, 0xff, 4, 0);                           // flag != 0: k_thin writes
            hipEventRecord(e[0], 0);
            hipLaunchKernelGGL(k_thin, dim3(75u << 10), dim3(256), 0, 0, flag, big);        // 75 Mi words... 300 MB? no: 75 Ki blocks x 256 x 4 B = 75 MiB
            hipEventRecord(e[1], 0);
            hipMemsetAsync(flag, 0, 4, 0);
            hipLaunchKernelGGL(k_fat, dim3(512), dim3(256), 0, 0, flag, pts);
            hipEventRecord(e[2], 0);
            hipLaunchKernelGGL(k_fat, dim3(512), dim3(256), 0, 0, flag, pts);
            hipEventRecord(e[3], 0);
            hipLaunchKernelGGL(k_fat_lds, dim3(256), dim3(256), 0, 0, flag, pts);
            hipEventRecord(e[4], 0);
            hipEventSynchronize(e[4]);
            for (int i = 0; i < 4; ++i) { float ms; hipEventElapsedTime(&ms, e[i], e[i + 1]); acc[i] += ms * 1000.f / 50; }
        }
        printf("writer 75 MiB: %.1f us | memset + no-op fat(512) behind it: %.1f us | second no-op: %.1f us | third (lds): %.1f us\n", acc[0], acc[1], acc[2], acc[3]);
    }
    return 0;
}
