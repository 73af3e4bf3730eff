// What a three-way bf16 split of two float32 costs beside a bf16 MFMA, for three instruction selections:
//   S0  the kernels' form: three v_cvt_pk_bf16_f32, pieces rebuilt as floats by shift / mask, two v_sub_f32 per element
//   S1  as S0, the third piece (exactly representable: nothing to round) packed by v_perm_b32 instead of a conversion
//   S2  round-to-nearest-even in integer arithmetic (v_bfe_u32 + v_add3_u32 + v_and_b32 per element and piece), v_perm_b32 packs:
//       only the two exact subtractions per element are float instructions
// Loop of [1 MFMA + K pair splits]: in one wave, or MFMAs in waves 0-3 and splits in waves 4-7 (same SIMDs).  Ticks of s_memtime per
// iteration (an MFMA alone: 32).  Build: hipcc --offload-arch=gfx950 -O3 -o split_cost tools/split_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned perm_hi(float lo, float hi) {   // the high halves of two floats in one dword
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float rne16(float x) {   // x rounded to nearest-even bf16, as a float (finite x)
    const unsigned u = __float_as_uint(x);
    return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}
template <int S>
__device__ __forceinline__ void split2(float v0, float v1, unsigned &h, unsigned &m, unsigned &l) {
    if constexpr (S == 2) {
        const float h0 = rne16(v0), h1 = rne16(v1);
        const float r0 = v0 - h0, r1 = v1 - h1;
        const float m0 = rne16(r0), m1 = rne16(r1);
        h = perm_hi(h0, h1), m = perm_hi(m0, m1), l = perm_hi(r0 - m0, r1 - m1);
    } else {
        h = pk(v0, v1);
        float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
        m = pk(r0, r1);
        r0 -= __uint_as_float(m << 16), r1 -= __uint_as_float(m & 0xffff0000u);
        l = S == 1 ? perm_hi(r0, r1) : pk(r0, r1);
    }
}

template <int K, int S, int MODE>   // MODE 0 one wave does both, 1 partner waves, 2 MFMA only, 3 splits only
__global__ void __launch_bounds__(512) k(unsigned *out, unsigned long long *cyc, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (1.0f + threadIdx.x * 1.37e-3f + i * 0.61f);
    unsigned sink = 0;
    const int role = __builtin_amdgcn_readfirstlane(MODE == 1 ? (wave < 4 ? 2 : 3) : MODE);
    if (MODE != 1 && wave >= 4) return;
    __syncthreads();
    auto splits = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < K; ++n) {
            float x0 = v[2 * (n & 3)], x1 = v[2 * (n & 3) + 1];
            asm volatile("" : "+v"(x0), "+v"(x1));   // the compiler may not know the values: every split is computed
            unsigned h, m, l;
            split2<S>(x0, x1, h, m, l);
            asm volatile("" ::"v"(h), "v"(m), "v"(l));   // and every piece is used (an LDS write in the kernels)
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                splits();
            }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) splits();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = sink + __float_as_uint(s);
    if (threadIdx.x == 0 || threadIdx.x == 256) cyc[blockIdx.x * 2 + (threadIdx.x >> 8)] = t1 - t0;
}

// the three forms give the same pieces (S2's rounding is round-to-nearest-even like the conversion instruction's)
template <int S>
__global__ void pieces(const float *x, unsigned *o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) split2<S>(x[2 * i], x[2 * i + 1], o[3 * i], o[3 * i + 1], o[3 * i + 2]);
}

template <int K, int S, int MODE>
static void run(unsigned *out, unsigned long long *cyc, const char *mode) {
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<K, S, MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipLaunchKernelGGL((k<K, S, MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, cyc + 16, sizeof(h), hipMemcpyDeviceToHost);
    const double m = (double)h[0] / (iters * 4.0), p = (double)h[1] / (iters * 4.0);
    if (MODE == 1) printf("S%d  %-12s K=%d pair splits per MFMA: MFMA wave %6.1f ticks, split wave %6.1f ticks per iteration\n", S, mode, K, m, p);
    else printf("S%d  %-12s K=%d pair splits per MFMA: %6.1f ticks per iteration\n", S, mode, K, m);
}
#define ROWS(S)                         \
    run<1, S, 3>(out, cyc, "splits only"); \
    run<2, S, 3>(out, cyc, "splits only"); \
    run<4, S, 3>(out, cyc, "splits only"); \
    run<1, S, 0>(out, cyc, "one wave");    \
    run<2, S, 0>(out, cyc, "one wave");    \
    run<4, S, 0>(out, cyc, "one wave");    \
    run<1, S, 1>(out, cyc, "partner");     \
    run<2, S, 1>(out, cyc, "partner");     \
    run<4, S, 1>(out, cyc, "partner");

int main() {
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 2 * 8);
    // agreement of the pieces on 2^20 values over the float range (finite, normal)
    const int n = 1 << 20;
    float *hx = (float *)malloc(n * 4), *dx;
    unsigned *o0, *o1, *o2, *h0 = (unsigned *)malloc(n * 6), *h1 = (unsigned *)malloc(n * 6), *h2 = (unsigned *)malloc(n * 6);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        unsigned u = (unsigned)(st >> 16);
        unsigned e = 1 + (u >> 23) % 253;   // exponent field 1 .. 253
        u = (u & 0x807fffffu) | (e << 23);
        memcpy(&hx[i], &u, 4);
    }
    hipMalloc(&dx, n * 4), hipMalloc(&o0, n * 6), hipMalloc(&o1, n * 6), hipMalloc(&o2, n * 6);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((pieces<0>), dim3(n / 2 / 256), dim3(256), 0, 0, dx, o0, n);
    hipLaunchKernelGGL((pieces<1>), dim3(n / 2 / 256), dim3(256), 0, 0, dx, o1, n);
    hipLaunchKernelGGL((pieces<2>), dim3(n / 2 / 256), dim3(256), 0, 0, dx, o2, n);
    hipMemcpy(h0, o0, n * 6, hipMemcpyDeviceToHost), hipMemcpy(h1, o1, n * 6, hipMemcpyDeviceToHost), hipMemcpy(h2, o2, n * 6, hipMemcpyDeviceToHost);
    long d1 = 0, d2 = 0;
    for (int i = 0; i < n / 2 * 3; ++i) d1 += h0[i] != h1[i], d2 += h0[i] != h2[i];
    printf("pieces of 2^20 random floats (exponent fields 1..253): S1 differs from S0 in %ld dwords, S2 in %ld\n", d1, d2);
    run<1, 0, 2>(out, cyc, "MFMA only");
    ROWS(0)
    ROWS(1)
    ROWS(2)
    return 0;
}
