// Bias and spread of ONE v_mfma_f32_32x32x16_bf16 (D = C + A B, 16 products per output) against the exact sum, in units of ulp(C), for
// products that are small against the accumulator -- the situation of a long accumulation.  Companion of tools/mfma_round.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned short *A, const unsigned short *B, const float *C, float *D) {   // A [32][16], B [16][32] bf16 bits, C/D [32][32]
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned short a[8], b[8];
    for (int j = 0; j < 8; ++j) a[j] = A[r * 16 + 8 * h + j], b[j] = B[(8 * h + j) * 32 + r];
    bf16x8 av, bv;
    memcpy(&av, a, 16);
    memcpy(&bv, b, 16);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float fb(unsigned short s) { unsigned u = (unsigned)s << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    unsigned short hA[512], hB[512], *dA, *dB;
    float hC[1024], hD[1024], *dC, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    srand(7);
    auto rnd = [] { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
    for (int scale = 0; scale <= 24; scale += 4) {        // products ~ 2^-scale of the accumulator
        for (int csign = 0; csign < 2; ++csign) {
            double sum = 0, sum2 = 0, sumr = 0, sumr2 = 0;
            int n = 0;
            for (int rep = 0; rep < 64; ++rep) {
                for (int i = 0; i < 512; ++i) hA[i] = bf((float)rnd()), hB[i] = bf((float)(rnd() * ldexp(1.0, -scale)));
                for (int i = 0; i < 1024; ++i) hC[i] = (float)((1.0 + (rand() / (double)RAND_MAX)) * (csign ? -1.0 : 1.0));
                hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
                hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double ex = hC[i * 32 + j];
                        for (int kk = 0; kk < 16; ++kk) ex += (double)fb(hA[i * 16 + kk]) * (double)fb(hB[kk * 32 + j]);
                        const double ulp = ldexp(1.0, -23) * (fabs(ex) >= 2.0 ? 2.0 : 1.0);
                        const double e = (hD[i * 32 + j] - ex) / ulp, er = ((double)(float)ex - ex) / ulp;   // the instruction; one correct rounding
                        sum += e, sum2 += e * e, sumr += er, sumr2 += er * er, ++n;
                    }
            }
            printf("products ~ 2^-%-2d of C (%c): bf16 MFMA error mean %+.4f ulp, rms %.4f ulp   | one correctly rounded add: mean %+.4f, rms %.4f\n", scale,
                   csign ? '-' : '+', sum / n, sqrt(sum2 / n), sumr / n, sqrt(sumr2 / n));
        }
    }
    return 0;
}
