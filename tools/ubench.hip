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
#include "../zkevm-circuits_amd/csrc/ff29.hip.hpp"      // the products: one asm statement each (mul29 = mul29_asm on the device), the C forms as mul29_c ...

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

// mixes: one v_mad_u64_u32 and one other instruction per pair, on independent registers -- does the other instruction cost issue time beside the multiply-add?
#define KMIX(NAME, ASMSTR, TY)                                                          \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                      \
        uint64_t acc[CH];                                                              \
        TY z[CH];                                                                      \
        uint32_t x = a + threadIdx.x, y = b + blockIdx.x;                              \
        for (int i = 0; i < CH; ++i) { acc[i] = i + threadIdx.x; z[i] = (TY)(i * 77 + threadIdx.x + b); } \
        for (int it = 0; it < ITERS; ++it) {                                           \
            _Pragma("unroll") for (int r = 0; r < 2; ++r) {                            \
                _Pragma("unroll") for (int i = 0; i < CH; ++i)                         \
                    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t" ASMSTR : "+v"(acc[i]), "+v"(z[i]) : "v"(x), "v"(y) : "vcc"); \
            }                                                                          \
        }                                                                              \
        uint64_t s = 0;                                                                \
        for (int i = 0; i < CH; ++i) s += acc[i] + (uint64_t)z[i];                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
    }
KMIX(k_mix_xor, "v_xor_b32 %1, %1, %2", uint32_t)
KMIX(k_mix_2xor, "v_xor_b32 %1, %1, %2\n\tv_add_u32 %1, %1, %3", uint32_t)
KMIX(k_mix_and, "v_and_b32 %1, 0x1fffffff, %1", uint32_t)
KMIX(k_mix_mullo, "v_mul_lo_u32 %1, %1, %2", uint32_t)
KMIX(k_mix_alignbit, "v_alignbit_b32 %1, %1, %2, 7", uint32_t)
KMIX(k_mix_lshr64, "v_lshrrev_b64 %1, 1, %1", uint64_t)
KMIX(k_mix_lshladd64, "v_lshl_add_u64 %1, %1, 0, %0", uint64_t)

// v_mad_u64_u32 with one factor in a scalar register / both factors the same vector register / a zero addend: what bounds the instruction's rate?
#define KMAD(NAME, ASMSTR, YC)                                                          \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                      \
        uint64_t acc[CH];                                                              \
        uint32_t x = a + threadIdx.x, y = b + blockIdx.x;                              \
        for (int i = 0; i < CH; ++i) acc[i] = i + threadIdx.x;                         \
        for (int it = 0; it < ITERS; ++it) {                                           \
            _Pragma("unroll") for (int r = 0; r < 2; ++r) {                            \
                _Pragma("unroll") for (int i = 0; i < CH; ++i)                         \
                    asm volatile(ASMSTR : "+v"(acc[i]) : "v"(x), YC(y), "s"(b) : "vcc", "s20", "s21"); \
            }                                                                          \
        }                                                                              \
        uint64_t s = 0;                                                                \
        for (int i = 0; i < CH; ++i) s += acc[i];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
    }
KMAD(k_mad64_vs, "v_mad_u64_u32 %0, vcc, %1, %3, %0", "v")
KMAD(k_mad64_xx, "v_mad_u64_u32 %0, vcc, %1, %1, %0", "v")
KMAD(k_mad64_vc, "v_mad_u64_u32 %0, vcc, %1, 17, %0", "v")
KMAD(k_mad64_sgprcarry, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "v")

// the same multiply-add as ONE dependent chain per thread (every instruction accumulates into the register pair the one before it wrote),
// as two and as four chains: the column sums of the Montgomery product are chains of this kind
template <int NCH>
__global__ void k_mad64_chain(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[NCH];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < NCH; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NCH; ++r) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < NCH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Rate of the int8 matrix instruction the round-4 review asked about (tools/ only: the product path keeps MFMA unused, north_star): the
// reduction half of a Montgomery product as a contraction against a constant Toeplitz matrix of the modulus' bytes would be, per 32
// field elements, one 64 x 32 x 32 tile product for m p (two v_mfma_i32_32x32x32_i8) + one 32 x 32 x 32 for m = t (-1/p) mod 2^256 (one).
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
__global__ void k_mfma_i8(uint64_t* out, uint32_t a, uint32_t b) {
    mfma_v4i x = {(int)(a + threadIdx.x), (int)(b ^ threadIdx.x), (int)(a * 3u + blockIdx.x), (int)(b + 7u)}, y = {(int)b, (int)a, (int)(a ^ b), (int)threadIdx.x};
    mfma_v16i c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = i + j;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c[i], 0, 0, 0);
    }
    uint64_t s_ = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s_ += (uint32_t)c[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_;
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
            if (OP == 2) { a[i] = zk::sub29k<4>(a[i], b); zk::normalize29(a[i]); a[i].l[8] &= 0xffff; }
            if (OP == 3) a[i] = zk::mul29(a[i], a[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 9; ++j) s += a[i].l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- the asm products (csrc/mul29_asm.hip.hpp) against the C forms they replace: results on pseudo-random operands at the documented
// limb bounds (mismatches counted on the device), and the rate of each
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <class P>
__global__ void k_cmp29(uint32_t* bad, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    zk::F29<P> a, b, c, d, u;
    for (int i = 0; i < 9; ++i) {
        const uint32_t top = i == 8;
        a.l[i] = hash32(seed + tid * 41 + i) & (top ? 0x3ffffffu : 0x3fffffffu);          // limbs < 2^30, value < 2^258
        b.l[i] = hash32(seed + tid * 43 + i + 100) & (top ? 0xffffffu : 0x1fffffffu);     // normalised, < 2^256
        c.l[i] = hash32(seed + tid * 47 + i + 200) & (top ? 0xffffffu : 0x1fffffffu);
        d.l[i] = hash32(seed + tid * 53 + i + 300) & (top ? 0xffffffu : 0x3fffffffu);     // limbs < 2^30
        u.l[i] = hash32(seed + blockIdx.x * 59 + i + 400) & (top ? 0xffffffu : 0x1fffffffu);   // wave-uniform second factor
        if ((tid & 15) == 3 && i < 8) { a.l[i] = 0x3fffffffu; b.l[i] = 0x1fffffffu; c.l[i] = 0x1fffffffu; d.l[i] = 0x3fffffffu; }     // everything at its bound
    }
    zk::F29<P> an = a;
    for (int i = 0; i < 8; ++i) an.l[i] &= 0x1fffffffu;
    uint32_t diff = 0;
    auto cmp = [&](const zk::F29<P>& x, const zk::F29<P>& y, uint32_t bit) { for (int i = 0; i < 9; ++i) if (x.l[i] != y.l[i]) diff |= bit; };
    cmp(zk::mul29(a, b), zk::mul29_c(a, b), 1u);
    cmp(zk::mul29_ub(a, u), zk::mul29_ub_c(a, u), 2u);
    cmp(zk::sqr29(an), zk::sqr29_c(an), 4u);
    cmp(zk::mul2add29(an, b, c, d), zk::mul2add29_c(an, b, c, d), 8u);
    if (diff) { atomicAdd(bad, 1u); atomicOr(bad + 1, diff); }
}
template <class P, int OP>
__global__ void k_field29c(uint64_t* out, const zk::Fp<typename P::P32>* in) {
    zk::F29<P> a[4], b = zk::unpack29<P>(zk::ldg(in + 4));
    for (int i = 0; i < 4; ++i) { auto x = zk::ldg(in + i); x.l[0] ^= threadIdx.x; x.l[7] &= 0x0fffffffu; a[i] = zk::unpack29<P>(x); }
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (OP == 0) a[i] = zk::mul29_c(a[i], b);
            if (OP == 3) a[i] = zk::sqr29_c(a[i]);
            if (OP == 4) a[i] = zk::sqr29(a[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 9; ++j) s += a[i].l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The radix-4 register step of csrc/ntt.hip (dit_step<2>: four products, lazy butterfly sums, one carry propagation per element) on registers
// only -- no LDS, no barriers, no twiddle loads: what the arithmetic of an NTT step costs by itself, in products per second.
template <int WITH_SUMS>
__global__ void k_ntt_step_alu(uint64_t* out, const zk::Fr* in) {
    using F = zk::Fr29;
    F e[4], w[3];
    for (int i = 0; i < 4; ++i) { auto x = zk::ldg(in + i); x.l[0] ^= threadIdx.x; x.l[7] &= 0x0fffffffu; e[i] = zk::unpack29<zk::Fr29P>(x); }
    for (int i = 0; i < 3; ++i) { auto x = zk::ldg(in + 1 + i); x.l[1] ^= threadIdx.x * 3; x.l[7] &= 0x0fffffffu; w[i] = zk::unpack29<zk::Fr29P>(x); }
    auto bfly = [&](F& u, F& x, const F& tw) {
        const F v = zk::mul29(x, tw);
        if (WITH_SUMS) { const F a0 = zk::add29(u, v), a1 = zk::sub29k<2>(u, v); u = a0; x = a1; }
        else { x = v; }
    };
    for (int it = 0; it < ITERS / 8; ++it) {
        bfly(e[0], e[1], w[0]);
        bfly(e[2], e[3], w[0]);
        bfly(e[0], e[2], w[1]);
        bfly(e[1], e[3], w[2]);
        if (WITH_SUMS) { zk::normalize29(e[0]); zk::normalize29(e[1]); zk::normalize29(e[2]); zk::normalize29(e[3]); for (int i = 0; i < 4; ++i) e[i].l[8] &= 0xffffff; }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 9; ++j) s += e[i].l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// correctness dump: o32[i] = a*b (R=2^256 Montgomery), o29[i] = pack(mul29(unpack a, unpack b))
__global__ void k_check29(const zk::Fq* a, const zk::Fq* b, zk::Fq* o32, zk::Fq* o29, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zk::Fq x = zk::ldg(a + i), y = zk::ldg(b + i);
    zk::stg(o32 + i, x * y);
    zk::Fq29 r = zk::mul29(zk::unpack29<zk::Fq29P>(x), zk::unpack29<zk::Fq29P>(y));
    zk::Fq29 d = zk::sub29k<4>(zk::add29(r, r), r);     // exercises add/sub: r + r - r + 4p
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
    RUN(k_mad64_vs); RUN(k_mad64_xx); RUN(k_mad64_vc); RUN(k_mad64_sgprcarry);
    RUN(k_mad64_chain<1>); RUN(k_mad64_chain<2>); RUN(k_mad64_chain<4>);
    RUN(k_mix_xor); RUN(k_mix_2xor); RUN(k_mix_and); RUN(k_mix_mullo); RUN(k_mix_alignbit); RUN(k_mix_lshr64); RUN(k_mix_lshladd64);
    for (int wps = 1; wps <= 8; wps *= 2) {      // occupancy: waves per SIMD
        const int bl = prop.multiProcessorCount * wps;
        double secs = time_kernel([&] { hipLaunchKernelGGL((k_field29<zk::Fr29P, 0>), dim3(bl), dim3(threads), 0, 0, out, (const zk::Fr*)out); });
        printf("Fr29 mul at %d waves/SIMD: %.1f G products/s\n", wps, (double)bl * threads * (double)(ITERS / 8) * 4 / secs * 1e-9);
    }
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
    report("Fr29 mul (C form)", time_kernel([&] { hipLaunchKernelGGL((k_field29c<zk::Fr29P, 0>), dim3(blocks), dim3(threads), 0, 0, out, din); }), nf);
    report("Fq29 mul (C form)", time_kernel([&] { hipLaunchKernelGGL((k_field29c<zk::Fq29P, 0>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 sqr (C form)", time_kernel([&] { hipLaunchKernelGGL((k_field29c<zk::Fq29P, 3>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    report("Fq29 sqr", time_kernel([&] { hipLaunchKernelGGL((k_field29c<zk::Fq29P, 4>), dim3(blocks), dim3(threads), 0, 0, out, (const zk::Fq*)din); }), nf);
    {
        uint32_t* dbad; CK(hipMalloc(&dbad, 8)); CK(hipMemset(dbad, 0, 8));
        for (uint32_t seed = 1; seed <= 8; ++seed) {
            hipLaunchKernelGGL((k_cmp29<zk::Fr29P>), dim3(1024), dim3(256), 0, 0, dbad, seed * 7919u);
            hipLaunchKernelGGL((k_cmp29<zk::Fq29P>), dim3(1024), dim3(256), 0, 0, dbad, seed * 104729u);
        }
        uint32_t hbad[2]; CK(hipMemcpy(hbad, dbad, 8, hipMemcpyDeviceToHost));
        printf("asm products against the C forms: %u mismatching lanes of %u (mask 0x%x: 1 mul29, 2 mul29_ub, 4 sqr29, 8 mul2add29)\n", hbad[0], 16u * 1024u * 256u, hbad[1]);
    }
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int bl = prop.multiProcessorCount * wps;
        const double np = (double)bl * threads * (double)(ITERS / 8) * 4;
        double s1 = time_kernel([&] { hipLaunchKernelGGL((k_ntt_step_alu<1>), dim3(bl), dim3(threads), 0, 0, out, (const zk::Fr*)din); });
        double s0 = time_kernel([&] { hipLaunchKernelGGL((k_ntt_step_alu<0>), dim3(bl), dim3(threads), 0, 0, out, (const zk::Fr*)din); });
        printf("NTT radix-4 step on registers at %d waves/SIMD: %.1f G products/s with the butterfly sums and carry propagation, %.1f G/s products alone\n", wps, np / s1 * 1e-9, np / s0 * 1e-9);
    }
    {
        const double secs = time_kernel([&] { hipLaunchKernelGGL(k_mfma_i8, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); });
        const double insts = (double)blocks * threads / 64.0 * (double)ITERS * 16.0;          // wave-level instructions
        printf("v_mfma_i32_32x32x32_i8: %.2f G wave-instructions/s (%.1f T int8 multiply-adds/s), %.2f ns per instruction and SIMD; three of them per 32 field elements = %.3f ns per element and SIMD\n",
               insts / secs * 1e-9, insts * 32768.0 / secs * 1e-12, secs * nsimd / insts * 1e9, 3.0 / 32.0 * secs * nsimd / insts * 1e9);
        printf("   against the reduction half on the VALU: 81 v_mad_u64_u32 per element = %.3f ns per element and SIMD (64 elements per wave-instruction)\n", 81.0 * 2.07 / 64.0);
    }
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
