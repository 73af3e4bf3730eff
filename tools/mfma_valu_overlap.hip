// How many vector instructions issue in the shadow of one v_mfma_f32_32x32x16_bf16 (32 cycles on its SIMD)?  One wave per SIMD (or two,
// the second one doing only the vector work) runs a loop of [1 MFMA + n vector instructions of one kind] and reports cycles per iteration.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/mfma_valu_overlap.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int N, int KIND, int MODE>   // MODE 0: both in one wave; 1: wave w multiplies, wave w + 4 does the vector work; 2: MFMA only; 3: vector only
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
    unsigned u[12];
    for (int i = 0; i < 12; ++i) u[i] = threadIdx.x + i;
    const int role = __builtin_amdgcn_readfirstlane(MODE == 1 ? (wave < 4 ? 2 : 3) : MODE);   // 0 both, 2 MFMA only, 3 vector only (uniform per wave)
    if (MODE != 1 && wave >= 4) return;
    __syncthreads();
    auto vec = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int j = n % 12;
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 5) % 12]));
            if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(v[j]), "v"(v[(j + 1) % 12]));
            if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[j]));
            if (KIND == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[j]) : "v"(u[(j + 3) % 12]));
            if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f32x2 *)&v[2 * (j % 6)]) : "v"(*(f32x2 *)&v[2 * ((j + 2) % 6)]));
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {   // (one loop per role: no branch inside the timed loops)
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                vec();
            }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) vec();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 12; ++i) s += v[i] + (float)u[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 || threadIdx.x == 256) cyc[blockIdx.x * 2 + (threadIdx.x >> 8)] = t1 - t0;
}

template <int N, int KIND, int MODE>
static void run(float *out, unsigned long long *cyc, const char *kind, const char *mode) {
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<N, KIND, MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, KIND, MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, cyc + 16, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-18s %-26s n=%2d  %7.1f ns per [MFMA + n]   (%.1f / %.1f s_memtime ticks)\n", kind, mode, N, ms * 1e6 / (iters * 8.0),
           (double)h[0] / (iters * 8.0), (double)h[1] / (iters * 8.0));
}

#define ROW(KIND, NAME)                                         \
    run<0, KIND, 2>(out, cyc, NAME, "MFMA only");                \
    run<4, KIND, 3>(out, cyc, NAME, "vector only");              \
    run<8, KIND, 3>(out, cyc, NAME, "vector only");              \
    run<2, KIND, 0>(out, cyc, NAME, "one wave");                 \
    run<4, KIND, 0>(out, cyc, NAME, "one wave");                 \
    run<6, KIND, 0>(out, cyc, NAME, "one wave");                 \
    run<8, KIND, 0>(out, cyc, NAME, "one wave");                 \
    run<12, KIND, 0>(out, cyc, NAME, "one wave");                \
    run<4, KIND, 1>(out, cyc, NAME, "partner wave on the SIMD"); \
    run<8, KIND, 1>(out, cyc, NAME, "partner wave on the SIMD"); \
    run<12, KIND, 1>(out, cyc, NAME, "partner wave on the SIMD");

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 2 * 8);
    ROW(0, "v_add_f32")
    ROW(1, "v_cvt_pk_bf16_f32")
    ROW(2, "v_and_b32")
    ROW(3, "v_lshlrev_b32")
    ROW(4, "v_pk_add_f32")
    return 0;
}
