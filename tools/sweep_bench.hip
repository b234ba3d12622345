// Micro-benchmark of the MSM digit sweep (csrc/msm.hip: k_msm_lds_sweep) to find what bounds it.
// Variants: 0 = loads only, 1 = loads + hit mask, 2 = full count pass; block sizes 256/512/1024.
// build: hipcc -O3 --offload-arch=gfx950 tools/sweep_bench.hip -o /tmp/sweep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

constexpr uint32_t DIG_ZERO = 0xFFFFu;

template <int MODE>
__global__ void __launch_bounds__(1024) k_sweep(const uint16_t* __restrict__ dig, uint64_t n_pad, int range_bits, uint32_t B, uint32_t* __restrict__ counts, uint32_t nwin, int xcd_map) {
    __shared__ uint32_t lds[2048];
    const uint32_t nranges = B >> range_bits;
    uint32_t r, w;
    if (xcd_map) { const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3; r = slot % nranges; w = xcd + 8u * (slot / nranges); }
    else { r = blockIdx.x % nranges; w = blockIdx.x / nranges; }
    const uint32_t range = 1u << range_bits, rmask = range - 1;
    if (w >= nwin) return;
    const uint64_t gbase = (uint64_t)w * B + ((uint64_t)r << range_bits);
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) lds[t] = 0u;
    __syncthreads();
    const uint32_t field = (0x7FFFu & ~rmask) * 0x00010001u, want = (r << range_bits) * 0x00010001u;
    const uint4* row = reinterpret_cast<const uint4*>(dig + (uint64_t)w * n_pad);
    const uint64_t nvec = n_pad >> 4;
    uint64_t v = threadIdx.x;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    uint32_t sink = 0;
    if (v < nvec) { q0 = row[2 * v]; q1 = row[2 * v + 1]; }
    while (v < nvec) {
        const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint64_t vn = v + blockDim.x;
        if (vn < nvec) { q0 = row[2 * vn]; q1 = row[2 * vn + 1]; }
        if (MODE == 0) { sink ^= wd[0] ^ wd[1] ^ wd[2] ^ wd[3] ^ wd[4] ^ wd[5] ^ wd[6] ^ wd[7]; v = vn; continue; }
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t z = (wd[j] ^ want) & field;
            const uint32_t nz = (z + 0x7FFF7FFFu) & 0x80008000u;
            m = (m >> 2) | nz;
        }
        uint32_t hits = ~m & 0xAAAAAAAAu;
        if (MODE == 1) { sink ^= hits; v = vn; continue; }
        while (hits) {
            const uint32_t p = (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1;
            const uint32_t j = (p & 15u) >> 1, hi = p >> 4;
            uint32_t word = wd[0];
#pragma unroll
            for (uint32_t t = 1; t < 8; ++t) word = j == t ? wd[t] : word;
            const uint32_t code = (word >> (hi * 16)) & 0xFFFFu;
            atomicAdd(&lds[code & rmask], 1u);
        }
        v = vn;
    }
    __syncthreads();
    if (MODE != 2) { if (sink == 0x12345678u) counts[gbase] = sink; return; }
    for (uint32_t t = threadIdx.x; t < range; t += blockDim.x) counts[gbase + t] = lds[t];
}

int main() {
    const uint64_t n = 1 << 20, n_pad = n;
    const int W = 16, range_bits = 11;
    const uint32_t B = 1u << 15;
    std::vector<uint16_t> h(n_pad * W);
    uint64_t st = 88172645463325252ull;
    for (auto& x : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (uint16_t)(st >> 20); if (x == 0xFFFF) x = 0; }
    uint16_t* d; uint32_t* c;
    hipMalloc(&d, h.size() * 2); hipMalloc(&c, (size_t)W * B * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t grid = 8u * ((W + 7) / 8) * (B >> range_bits);
    auto run = [&](const char* name, auto kern, int threads, int xcd) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, n_pad, range_bits, B, c, (uint32_t)W, xcd);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, n_pad, range_bits, B, c, (uint32_t)W, xcd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s threads=%4d xcd_map=%d  %8.1f us\n", name, threads, xcd, ms * 100);
    };
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int threads : {256, 512, 1024}) {
            run("loads", k_sweep<0>, threads, xcd);
            run("loads+mask", k_sweep<1>, threads, xcd);
            run("count", k_sweep<2>, threads, xcd);
        }
    std::vector<uint32_t> hc((size_t)W * B);
    hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
    uint64_t tot = 0; for (auto x : hc) tot += x;
    printf("total counted %llu of %llu\n", (unsigned long long)tot, (unsigned long long)(n * W));
    return 0;
}
