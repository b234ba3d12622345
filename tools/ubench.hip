// Instruction-throughput microbenchmark for the integer paths that bound 256-bit modular
// arithmetic on gfx950 ("v_mad_u64_u32 / v_mul_hi_u32 rates on gfx950 are not in the
// local guides -- measure first").  Standalone: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip
// Prints one line per op: giga-lane-ops/s over the whole chip and cycles per wave-instruction
// per SIMD (assuming 1024 SIMDs at the measured clock proxy of 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zkevm-circuits_amd/csrc/ff.hip.hpp"
#include "../zkevm-circuits_amd/csrc/ff29.hip.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 512;
constexpr int CH = 8;   // independent chains per thread

#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < CH; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        BODY8(S) BODY8(S)
#undef S
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64_addc(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    uint32_t cnt[CH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < CH; ++i) { acc[i] = i + threadIdx.x; cnt[i] = 0; }
    for (int it = 0; it < ITERS; ++it) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(cnt[i]) : "v"(x), "v"(y) : "vcc");
        BODY8(S) BODY8(S)
#undef S
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i] + cnt[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define K32(NAME, ASMSTR)                                                              \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                      \
        uint32_t acc[CH];                                                              \
        uint32_t x = a + threadIdx.x;                                                  \
        for (int i = 0; i < CH; ++i) acc[i] = i + threadIdx.x + b;                     \
        for (int it = 0; it < ITERS; ++it) {                                           \
            _Pragma("unroll") for (int r = 0; r < 2; ++r) {                            \
                _Pragma("unroll") for (int i = 0; i < CH; ++i)                         \
                    asm volatile(ASMSTR : "+v"(acc[i]) : "v"(x) : "vcc");              \
            }                                                                          \
        }                                                                              \
        uint32_t s = 0;                                                                \
        for (int i = 0; i < CH; ++i) s += acc[i];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
    }
K32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
K32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
K32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %0")
K32(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %1")
K32(k_add, "v_add_u32 %0, %0, %1")
K32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
K32(k_fma32, "v_fma_f32 %0, %0, %1, %0")
K32(k_xor, "v_xor_b32 %0, %0, %1")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")

__global__ void k_fma64(uint64_t* out, uint32_t a, uint32_t b) {
    double acc[CH];
    double x = 1.0 + 1e-9 * threadIdx.x;
    for (int i = 0; i < CH; ++i) acc[i] = i + threadIdx.x + b;
    for (int it = 0; it < ITERS; ++it) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(x));
        BODY8(S) BODY8(S)
#undef S
    }
    double s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_lshladd64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    uint64_t x = a + threadIdx.x;
    for (int i = 0; i < CH; ++i) acc[i] = i + threadIdx.x + b;
    for (int it = 0; it < ITERS; ++it) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(x));
        BODY8(S) BODY8(S)
#undef S
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lshr64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[CH];
    for (int i = 0; i < CH; ++i) acc[i] = ((uint64_t)(i + threadIdx.x + b) << 40) | a;
    for (int it = 0; it < ITERS; ++it) {
#define S(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
        BODY8(S) BODY8(S)
#undef S
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// field-op chains: 4 independent chains/thread
template <class F, int OP>
__global__ void k_field(uint64_t* out, const F* in) {
    F a[4], b = zk::ldg(in + 4);
    for (int i = 0; i < 4; ++i) { a[i] = zk::ldg(in + i); a[i].l[0] ^= threadIdx.x; a[i].l[7] &= 0x0fffffffu; }
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (OP == 0) a[i] = a[i] * b;
            if (OP == 1) a[i] = a[i] + b;
            if (OP == 2) a[i] = a[i] - b;
            if (OP == 3) a[i] = zk::sqr(a[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += a[i].l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 29-bit-limb Montgomery product chains
template <class P, int OP>
__global__ void k_field29(uint64_t* out, const zk::Fp<typename P::P32>* in) {
    zk::F29<P> a[4], b = zk::unpack29<P>(zk::ldg(in + 4));
    for (int i = 0; i < 4; ++i) { auto x = zk::ldg(in + i); x.l[0] ^= threadIdx.x; x.l[7] &= 0x0fffffffu; a[i] = zk::unpack29<P>(x); }
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (OP == 0) a[i] = zk::mul29(a[i], b);
            if (OP == 1) { a[i] = zk::add29(a[i], b); zk::normalize29(a[i]); a[i].l[8] &= 0xffff; }
            if (OP == 2) { a[i] = zk::sub29(a[i], b); zk::normalize29(a[i]); a[i].l[8] &= 0xffff; }
            if (OP == 3) a[i] = zk::mul29(a[i], a[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 9; ++j) s += a[i].l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// correctness dump: o32[i] = a*b (R=2^256 Montgomery), o29[i] = pack(mul29(unpack a, unpack b))
__global__ void k_check29(const zk::Fq* a, const zk::Fq* b, zk::Fq* o32, zk::Fq* o29, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zk::Fq x = zk::ldg(a + i), y = zk::ldg(b + i);
    zk::stg(o32 + i, x * y);
    zk::Fq29 r = zk::mul29(zk::unpack29<zk::Fq29P>(x), zk::unpack29<zk::Fq29P>(y));
    zk::Fq29 d = zk::sub29(zk::add29(r, r), r);     // exercises add/sub: r + r - r + 4p
    zk::normalize29(d);
    zk::stg(o29 + i, zk::pack29(d));
}

template <class K>
double time_kernel(K launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;   // 8 waves/SIMD
    uint64_t* out; CK(hipMalloc(&out, sizeof(uint64_t) * blocks * threads));
    const double lanes = (double)blocks * threads;
    const double nsimd = prop.multiProcessorCount * 4.0;
    auto report = [&](const char* name, double secs, double ops_per_thread) {
        double gops = lanes * ops_per_thread / secs * 1e-9;
        double waveinst = lanes / 64.0 * ops_per_thread;            // wave-instructions total
        double cyc = secs * 2.4e9 / (waveinst / nsimd);             // cycles per wave-inst per SIMD @2.4GHz
        printf("%-22s %10.1f Gop/s   %6.2f cyc/wave-inst/SIMD (@2.4GHz)   t=%.3f ms\n", name, gops, cyc, secs * 1e3);
    };
    const double n16 = (double)ITERS * 16;
#define RUN(K) report(#K, time_kernel([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), n16)
    RUN(k_fma32); RUN(k_add); RUN(k_xor); RUN(k_addco); RUN(k_addc); RUN(k_alignbit);
    RUN(k_mul_lo); RUN(k_mul_hi); RUN(k_mad24); RUN(k_mulhi24);
    RUN(k_mad64); RUN(k_lshladd64); RUN(k_lshr64); RUN(k_fma64);
    report("k_mad64_addc(pair)", time_kernel([&] { hipLaunchKernelGGL(k_mad64_addc, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), n16);

    // field ops
    zk::Fr h[5];
    for (int i = 0; i < 5; ++i) { h[i] = zk::Fr::one(); h[i].l[0] += 17 * i; h[i].l[3] ^= 0x1234567 * (i + 1); }
    zk::Fr* din; CK(hipMalloc(&din, sizeof(h))); CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    const double nf = (double)(ITERS / 8) * 4;
    report("Fr mul", time_kernel([&] { hipLaunchKernelGGL((k_field<zk::Fr, 0>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    report("Fr sqr", time_kernel([&] { hipLaunchKernelGGL((k_field<zk::Fr, 3>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    report("Fr add", time_kernel([&] { hipLaunchKernelGGL((k_field<zk::Fr, 1>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    report("Fr sub", time_kernel([&] { hipLaunchKernelGGL((k_field<zk::Fr, 2>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    report("Fq mul", time_kernel([&] { hipLaunchKernelGGL((k_field<zk::Fq, 0>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 mul", time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fq29P, 0>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 sqr(as mul)", time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fq29P, 3>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 add+norm", time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fq29P, 1>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 sub+norm", time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fq29P, 2>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fr29 mul", time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fr29P, 0>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    {   // correctness dump for offline verification (tools/check29.py)
        const int n = 4096;
        std::vector<zk::Fq> ha(n), hb(n), h32(n), h29(n);
        uint64_t st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
        for (int i = 0; i < n; ++i) { for (int j = 0; j < 8; ++j) { ha[i].l[j] = rnd(); hb[i].l[j] = rnd(); } ha[i].l[7] &= 0x1fffffffu; hb[i].l[7] &= 0x1fffffffu; }
        zk::Fq *da, *db, *d32, *d29;
        CK(hipMalloc(&da, n * 32)); CK(hipMalloc(&db, n * 32)); CK(hipMalloc(&d32, n * 32)); CK(hipMalloc(&d29, n * 32));
        CK(hipMemcpy(da, ha.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), n * 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check29, dim3(n / 256), dim3(256), 0, 0, da, db, d32, d29, n);
        CK(hipMemcpy(h32.data(), d32, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(h29.data(), d29, n * 32, hipMemcpyDeviceToHost));
        FILE* f = fopen("gpurun_out/check29.bin", "wb");
        if (f) { fwrite(ha.data(), 32, n, f); fwrite(hb.data(), 32, n, f); fwrite(h32.data(), 32, n, f); fwrite(h29.data(), 32, n, f); fclose(f); }
    }
    return 0;
}
