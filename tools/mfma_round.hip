// What rounding does v_mfma_f32_32x32x16_bf16 apply when it adds its 16 products to the accumulator?  (and v_mfma_f32_32x32x2_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); return (unsigned short)(u >> 16); }   // exact inputs only
__global__ void k(const float *prod_a, const float *prod_b, int nprod, float c0, float *out, int use_f32) {
    // lane l: row r = l & 31, half h = l >> 5 holds A[r][8h + j], B[8h + j][c = l & 31].  Put the products in k = 0 .. nprod-1 (half 0).
    const int lane = threadIdx.x, h = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c0;
    if (!use_f32) {
        unsigned short a[8], b[8];
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * h + j;
            a[j] = kk < nprod ? f2bf(prod_a[kk]) : 0;
            b[j] = kk < nprod ? f2bf(prod_b[kk]) : 0;
        }
        bf16x8 av, bv;
        memcpy(&av, a, 16);
        memcpy(&bv, b, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    } else {   // 32x32x2: lane half h holds k = h
        const float a = h < nprod ? prod_a[h] : 0.f, b = h < nprod ? prod_b[h] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
int main() {
    float *da, *db, *dout;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    struct T { const char *name; int n; float a[16], b[16]; float c0; double exact; };
    const float e12 = ldexpf(1.f, -12), e13 = ldexpf(1.f, -13);
    T tests[] = {
        {"1 + (2^-24 + 2^-26): above half an ulp", 2, {e12, e13}, {e12, e13}, 1.f, 1.0 + ldexp(1.0, -24) + ldexp(1.0, -26)},
        {"1 + 2^-24: exactly half an ulp (tie)", 1, {e12}, {e12}, 1.f, 1.0 + ldexp(1.0, -24)},
        {"1 + (2^-24 - tiny): 2^-25 + 2^-26 below half", 2, {e12 , e13}, {e13, e13}, 1.f, 1.0 + ldexp(1.0, -25) + ldexp(1.0, -26)},
        {"-1 - (2^-24 + 2^-26)", 2, {-e12, -e13}, {e12, e13}, -1.f, -1.0 - ldexp(1.0, -24) - ldexp(1.0, -26)},
        {"1 + 3 * 2^-25 as three products", 3, {e12, e12, e12}, {e13, e13, e13}, 1.f, 1.0 + 3 * ldexp(1.0, -25)},
        {"1 + 16 products of 2^-27 (= 2^-23 exactly, one ulp)", 16, {e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13},
         {ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14)}, 1.f, 1.0 + ldexp(1.0, -23)},
        {"1 + 12 products of 2^-27 (= 1.5 ulp/2... 12 * 2^-27 = 0.75 ulp)", 12, {e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13,e13},
         {ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14),ldexpf(1.f,-14)}, 1.f, 1.0 + 12 * ldexp(1.0, -27)},
        {"big + small: 1024 + 2^-13 * 2^-1 ... product 2^-14 vs ulp(1024) = 2^-13: half ulp tie", 1, {ldexpf(1.f,-7)}, {ldexpf(1.f,-7)}, 1024.f, 1024.0 + ldexp(1.0,-14)},
        {"cancel: 1 + (1 * -1) + 2^-30 (product 2^-15 * 2^-15)", 2, {1.f, ldexpf(1.f,-15)}, {-1.f, ldexpf(1.f,-15)}, 1.f, ldexp(1.0,-30)},
    };
    for (auto &t : tests) {
        hipMemcpy(da, t.a, 64, hipMemcpyHostToDevice); hipMemcpy(db, t.b, 64, hipMemcpyHostToDevice);
        float rb = 0, rf = 0;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, t.n, t.c0, dout, 0);
        hipMemcpy(&rb, dout, 4, hipMemcpyDeviceToHost);
        const float rne = (float)t.exact;
        printf("%-80s exact %.12g  RNE %.9g  bf16-mfma %.9g (%+.2f ulp of RNE)", t.name, t.exact, rne, rb, (rb - rne) / ldexpf(1.f, -23) / fabsf(t.c0 == 0 ? 1 : (fabsf(t.c0) >= 1024 ? 1024 : 1)));
        if (t.n <= 2) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, t.n, t.c0, dout, 1);
            hipMemcpy(&rf, dout, 4, hipMemcpyDeviceToHost);
            printf("  f32-mfma %.9g", rf);
        }
        printf("\n");
    }
    return 0;
}
