// fc_api.hip -- fully connected head blocks: Linear -> {BatchNorm1d | LayerNorm | none} -> ReLU -> dropout
// (models/pointnet_pp_vonMises.py:32-35, pointnet_pp_8dir.py:81-85, pointnet_pp_mvM.py:82-83,91-122).
//
// The matrix products run on the same fused MFMA GEMM / dW kernels as the set-abstraction layers;
// the row/column normalisation passes here touch only M x N elements (M = batch), so they are plain
// wave-per-row / lane-per-column kernels with float64 accumulation.
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

#define PNPP_TRY(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != PNPP_OK) return rc_; \
    } while (0)

// y = dropout(relu(z*scale + shift))   (BatchNorm) or   y = dropout(relu(z + b))   (no norm)
__global__ void __launch_bounds__(256) fc_apply_cols_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const uint8_t *__restrict__ mask,
                                                            float drop_scale, int relu, int M, int N, float *__restrict__ y) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i % N);
        float v = fmaf(z[i], scale ? scale[n] : 1.f, shift[n]);
        if (relu) v = fmaxf(v, 0.f);
        if (mask) v = mask[i] ? v * drop_scale : 0.f;
        y[i] = v;
    }
}

// keep-bit of element idx of the dropout draw with stream id sid: one Philox4x32-10 word per element, exactly the draw of the
// BatchNorm epilogue (gemm_smallm_body, E_BN_APPLY) -- counter (idx, "DROP", sid), key = seed
__device__ __forceinline__ bool dropout_keep(unsigned idx, unsigned long long sid, unsigned long long seed, float drop_p) {
    unsigned c0 = idx, c1 = 0x44524f50u /* "DROP" */, c2 = (unsigned)sid, c3 = (unsigned)(sid >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int rd = 0; rd < 10; ++rd) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return (double)c0 >= (double)drop_p * 4294967296.0;
}

// LayerNorm over the feature axis of u = z + b; one 256-thread block per row.  mask_out (round 4): the keep-mask of nn.Dropout is
// DRAWN here (no RNG launch, graph-replayable: the stream id lives in device memory, rng_counter[0], and is bumped by the last row
// block through the ticket word rng_counter[1]) and written out for the backward pass
__global__ void __launch_bounds__(256) fc_apply_ln_kernel(const float *__restrict__ z, const float *__restrict__ b,
                                                          const float *__restrict__ nw, const float *__restrict__ nb,
                                                          const uint8_t *__restrict__ mask, float drop_scale, int relu, int N,
                                                          float eps, float *__restrict__ y, float *__restrict__ mean,
                                                          float *__restrict__ istd, uint8_t *__restrict__ mask_out, float drop_p,
                                                          unsigned long long rng_seed, unsigned long long *__restrict__ rng_counter) {
    __shared__ double red[2][4];
    const int m = blockIdx.x;
    const unsigned long long sid = mask_out ? rng_counter[0] : 0ull;   // requested before the reductions
    const float *zr = z + (size_t)m * N;
    double s1 = 0.0, s2 = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const double u = (double)zr[n] + (double)b[n];
        s1 += u;
        s2 += u * u;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s1 += shfl_xor_f64(s1, k), s2 += shfl_xor_f64(s2, k);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1, red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mu = s1 / N;
    double var = s2 / N - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    if (threadIdx.x == 0) mean[m] = (float)mu, istd[m] = (float)is;
    const float muf = (float)mu, isf = (float)is;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v = ((zr[n] + b[n]) - muf) * isf * nw[n] + nb[n];
        if (relu) v = fmaxf(v, 0.f);
        if (mask) v = mask[(size_t)m * N + n] ? v * drop_scale : 0.f;
        if (mask_out) {
            const bool keep = dropout_keep((unsigned)(m * N + n), sid, rng_seed, drop_p);
            mask_out[(size_t)m * N + n] = keep ? 1 : 0;
            v = keep ? v * drop_scale : 0.f;
        }
        y[(size_t)m * N + n] = v;
    }
    if (mask_out) {   // every row block has read the stream id above before it takes a ticket; the last one bumps it
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(&rng_counter[1], 1ull);
            if (t == (unsigned long long)gridDim.x - 1) {
                rng_counter[1] = 0ull;
                rng_counter[0] += 1ull;
            }
        }
    }
}

// ---- backward, BatchNorm1d / none: block = 32 columns x 8 row lanes --------------------------------
// g = dy * dropout * relu'; BN: dz = gamma*istd*(g - mean(g) - xhat*mean(g*xhat)); none: dz = g
// Row-parallel form of the plain (no normalisation) case for many rows (the per-point linear layers of the transformer,
// M = B * N): dz = dy with the dropout mask and the ReLU gate applied, and one [N] partial of the bias gradient per
// 256-row chunk (slab_reduce adds them).  grid = (N / 64 column blocks, M / 256 row chunks).
__global__ void __launch_bounds__(256)
fc_bwd_rows_kernel(const float *__restrict__ dy, const float *__restrict__ z, const uint8_t *__restrict__ mask, float drop_scale,
                   int relu, const float *__restrict__ bias, int M, int N, float *__restrict__ dz, float *__restrict__ slab) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl, m0 = blockIdx.y * 256;
    const float sh = n < N ? bias[n] : 0.f;
    float acc = 0.f;
    if (n < N)
        for (int r = rl; r < 256; r += 4) {
            const int m = m0 + r;
            if (m >= M) break;
            const size_t i = (size_t)m * N + n;
            float g = dy[i];
            if (mask) g = mask[i] ? g * drop_scale : 0.f;
            if (relu && !(z[i] + sh > 0.f)) g = 0.f;
            dz[i] = g;
            acc += g;
        }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && n < N) slab[(size_t)blockIdx.y * N + n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

__global__ void __launch_bounds__(256)
fc_bwd_cols_kernel(const float *__restrict__ dy, const float *__restrict__ z, const uint8_t *__restrict__ mask,
                   float drop_scale, int relu, int bn, const float *__restrict__ scale, const float *__restrict__ shift,
                   const float *__restrict__ mean, const float *__restrict__ istd, const float *__restrict__ bias, int M, int N,
                   int training, float *__restrict__ dz, float *__restrict__ dnw, float *__restrict__ dnb,
                   float *__restrict__ db, int phase, double *__restrict__ glob, double *__restrict__ local) {
    // phase 0: sums and dz in one pass (every block holds all rows of its columns).  SyncBN cuts it in two around the exchange
    // of the sums over the ranks: phase 1 writes this rank's sums (+ row count) to `glob` and `local`, phase 2 applies the
    // summed `glob` (count at glob[2N]) and takes the parameter gradients from `local`.
    __shared__ double red[8][2][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cl;
    const bool ok = n < N;
    float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f;
    if (ok) {
        if (bn) sc = scale[n], sh = shift[n], mu = mean[n], is = istd[n];
        else sh = bias[n];
    }
    double s1 = 0.0, s2 = 0.0;
    if (ok)
        for (int m = rl; m < M; m += 8) {
            const size_t i = (size_t)m * N + n;
            const float zz = z[i];
            float g = dy[i];
            if (mask) g = mask[i] ? g * drop_scale : 0.f;
            if (relu && !(fmaf(zz, sc, sh) > 0.f)) g = 0.f;
            s1 += (double)g;
            s2 += (double)g * (double)((zz - mu) * is);
        }
    red[rl][0][cl] = s1;
    red[rl][1][cl] = s2;
    __syncthreads();
    s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s1 += red[i][0][cl], s2 += red[i][1][cl];
    if (phase == 1) {
        if (ok && rl == 0) glob[n] = local[n] = s1, glob[N + n] = local[N + n] = s2;
        if (blockIdx.x == 0 && threadIdx.x == 0) glob[2 * N] = local[2 * N] = (double)M;
        return;
    }
    if (!ok) return;
    double count = (double)M;
    if (phase == 2) s1 = glob[n], s2 = glob[N + n], count = glob[2 * N];
    const float c1 = (bn && training) ? (float)(s1 / count) : 0.f, c2 = (bn && training) ? (float)(s2 / count) : 0.f;
    if (phase == 2) s1 = local[n], s2 = local[N + n];
    for (int m = rl; m < M; m += 8) {
        const size_t i = (size_t)m * N + n;
        const float zz = z[i];
        float g = dy[i];
        if (mask) g = mask[i] ? g * drop_scale : 0.f;
        if (relu && !(fmaf(zz, sc, sh) > 0.f)) g = 0.f;
        dz[i] = bn ? sc * (g - c1 - (zz - mu) * is * c2) : g;
    }
    if (rl == 0) {
        if (bn) {
            if (dnw) dnw[n] = (float)s2;
            if (dnb) dnb[n] = (float)s1;
            if (db) db[n] = training ? 0.f : (float)((double)sc * s1);
        } else if (db) {
            db[n] = (float)s1;
        }
    }
}

// ---- backward of a head block with at most 32 rows, ONE launch (round 3) ------------------------------------------------------------
// Linear -> BatchNorm1d (or none) -> ReLU -> Dropout, M <= 32 (models/pointnet_pp_vonMises.py:32-35 at batch 32): fc_bwd_cols_kernel
// (dz and the normalisation's parameter gradients, 5 us) and fc_dx_dw_kernel (dx = dz W, dW = dz^T x, 6 - 7 us) only differ by WHO holds
// dz -- and with 32 rows every workgroup that needs a column of dz can hold all of it: the column sums of BatchNorm-backward are sums
// over the rows of ONE tile.  So dz is never written: the dW workgroups (32 columns n x 128 columns k) rebuild their 32 x 32 tile of it
// from dy, z and the keep-mask, the dx workgroups (32 columns k, the reduction over n split over 16 waves) first rebuild ALL of dz --
// 1024 / N threads per column, each with its share of the 32 rows in registers: three coalesced, L2-resident streams -- into an LDS image
// that feeds their MFMA A operand.  (A first version formed the operand while loading it, lane = row: a 2 KB stride between the lanes of
// every load, 26 - 30 us.)  The parameter gradients of the normalisation come from the dW workgroups of the first k block.
// 5.0 + 5.6 and 5.7 + 7.5 us in four launches -> 9.5 and 11.1 us in two.
#ifndef FCF_EXP   // timing experiments (wrong results): 1 no constants pass in the dx tiles, 2 no reduction loop there, 4 no column pass in the dW blocks
#define FCF_EXP 0
#endif
struct FcFusedArgs {
    const float *dy, *z;
    const uint8_t *mask;
    float drop_scale;
    int relu, bn, training;
    const float *scale, *shift, *mean, *istd, *bias;
    const float *w, *x;
    int M, N, K;
    float *dx, *dw, *dnw, *dnb, *db;
    int g1, gx2;
};

__device__ __forceinline__ float fc_masked_g(const FcFusedArgs &P, size_t i, float zz, float sc, float sh) {
    float g = P.dy[i];
    if (P.mask) g = P.mask[i] ? g * P.drop_scale : 0.f;
    if (P.relu && !(fmaf(zz, sc, sh) > 0.f)) g = 0.f;
    return g;
}

template <int TPC>   // threads per column of dz in the dx tiles' first pass (1024 / TPC columns at a time, 32 / TPC rows per thread)
__global__ void __launch_bounds__(1024) fc_bwd_fused_kernel(const FcFusedArgs P) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int tid = threadIdx.x, M = P.M, N = P.N, K = P.K;
    if ((int)blockIdx.x >= P.g1) {
        // ---- dW block: columns n0 .. n0 + 32 of dz (all rows), columns k0 .. k0 + 128 of x ----
        float (*dzs)[32] = reinterpret_cast<float (*)[32]>(fl);                    // [32][32]
        float (*xs)[128] = reinterpret_cast<float (*)[128]>(fl + 32 * 32);         // [32][128]
        float (*gs)[32] = reinterpret_cast<float (*)[32]>(fl + 32 * 32 + 32 * 128);   // masked upstream gradient
        float (*zs)[32] = gs + 32;                                                  // xhat (bn) of the same elements
        const int bx = ((int)blockIdx.x - P.g1) % P.gx2, by = ((int)blockIdx.x - P.g1) / P.gx2;
        const int k0 = bx * 128, n0 = by * 32;
        {
            const int m = tid >> 5, n = tid & 31, nn = n0 + n;
            float g = 0.f, xh = 0.f;
            if (m < M && nn < N) {
                const size_t i = (size_t)m * N + nn;
                const float zz = P.z[i];
                const float sc = P.bn ? P.scale[nn] : 1.f, sh = P.bn ? P.shift[nn] : P.bias[nn];
                g = fc_masked_g(P, i, zz, sc, sh);
                if (P.bn) xh = (zz - P.mean[nn]) * P.istd[nn];
            }
            gs[m][n] = g, zs[m][n] = xh;
        }
        for (int f = tid; f < 32 * 128; f += 1024) {
            const int m = f >> 7, k = f & 127;
            xs[m][k] = (m < M && k0 + k < K) ? P.x[(size_t)m * K + k0 + k] : 0.f;
        }
        __syncthreads();
        if (tid < 32 && !(FCF_EXP & 4)) {   // one thread per column: the sums over the rows, in row order
            const int nn = n0 + tid;
            double s1 = 0.0, s2 = 0.0;
            for (int m = 0; m < 32; ++m) s1 += (double)gs[m][tid], s2 += (double)gs[m][tid] * (double)zs[m][tid];
            const float sc = (P.bn && nn < N) ? P.scale[nn] : 1.f;
            const float c1 = (P.bn && P.training) ? (float)(s1 / (double)M) : 0.f, c2 = (P.bn && P.training) ? (float)(s2 / (double)M) : 0.f;
            for (int m = 0; m < 32; ++m) dzs[m][tid] = (m < M && nn < N) ? (P.bn ? sc * (gs[m][tid] - c1 - zs[m][tid] * c2) : gs[m][tid]) : 0.f;
            if (bx == 0 && nn < N) {
                if (P.bn) {
                    if (P.dnw) P.dnw[nn] = (float)s2;
                    if (P.dnb) P.dnb[nn] = (float)s1;
                    // a bias in front of a train-mode BatchNorm has exactly zero gradient; with running statistics d(bias) = scale sum g
                    if (P.db) P.db[nn] = P.training ? 0.f : (float)((double)sc * s1);
                } else if (P.db) {
                    P.db[nn] = (float)s1;
                }
            }
        }
        __syncthreads();
        const int kl = tid & 127, nh = tid >> 7;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {
            const float xv = xs[m][kl];
            const float4 d = *reinterpret_cast<const float4 *>(&dzs[m][4 * nh]);
            acc[0] = fmaf(d.x, xv, acc[0]), acc[1] = fmaf(d.y, xv, acc[1]);
            acc[2] = fmaf(d.z, xv, acc[2]), acc[3] = fmaf(d.w, xv, acc[3]);
        }
        if (k0 + kl < K)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 4 * nh + j;
                if (n < N) P.dw[(size_t)n * K + k0 + kl] = acc[j];
            }
        return;
    }
    // ---- dx tile: columns k0 .. k0 + 32 of dx = dz W.  Pass 1: thread n holds column n of all 32 rows (three coalesced streams,
    // eight rows in flight at a time), sums it, and writes its dz column to an LDS image [32][N + 1]; pass 2: wave w takes the
    // reduction indices n = 2 w + 32 j (+ lane half) from that image, W from global memory (coalesced over the 32 output columns) ----
    const int DP = N + 1;
    float *dzs = fl;                                 // [32][N + 1]; afterwards [16 waves][16 registers][64 lanes] partial tiles
    const int k0 = (int)blockIdx.x * 32;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const bool colok = k0 + l31 < K;
    const float *wcol = P.w + k0 + (colok ? l31 : 0);
    constexpr int CPP = 1024 / TPC, R = 32 / TPC;    // columns per pass, rows per thread
    double *ps = reinterpret_cast<double *>(fl + ((32 * DP + 1) & ~1));   // [TPC][2][CPP] partial sums of a pass
    for (int c0 = 0; c0 < ((FCF_EXP & 1) ? 0 : N); c0 += CPP) {
        const int cl = tid % CPP, rg = tid / CPP, n = c0 + cl, nc = min(n, N - 1);
        const float sc = P.bn ? P.scale[nc] : 1.f, sh = P.bn ? P.shift[nc] : P.bias[nc];
        const float mu = P.bn ? P.mean[nc] : 0.f, is = P.bn ? P.istd[nc] : 0.f;
        float g[R], xh[R];
#pragma unroll
        for (int m0 = 0; m0 < R; m0 += 8) {
            float zr[8], gr[8];
            unsigned char kr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned i = (unsigned)min(rg * R + m0 + j, M - 1) * (unsigned)N + (unsigned)nc;
                zr[j] = P.z[i], gr[j] = P.dy[i], kr[j] = P.mask ? P.mask[i] : (unsigned char)1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gg = kr[j] ? gr[j] * (P.mask ? P.drop_scale : 1.f) : 0.f;
                if (P.relu && !(fmaf(zr[j], sc, sh) > 0.f)) gg = 0.f;
                if (rg * R + m0 + j >= M) gg = 0.f;
                g[m0 + j] = gg, xh[m0 + j] = (zr[j] - mu) * is;
            }
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int m = 0; m < R; ++m) s1 += (double)g[m], s2 += (double)g[m] * (double)xh[m];
        if (TPC > 1) {   // the row groups of a column, in row order
            ps[(rg * 2 + 0) * CPP + cl] = s1, ps[(rg * 2 + 1) * CPP + cl] = s2;
            __syncthreads();
            s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int q = 0; q < TPC; ++q) s1 += ps[(q * 2 + 0) * CPP + cl], s2 += ps[(q * 2 + 1) * CPP + cl];
            __syncthreads();
        }
        const float c1 = (P.bn && P.training) ? (float)(s1 / (double)M) : 0.f, c2 = (P.bn && P.training) ? (float)(s2 / (double)M) : 0.f;
        if (n < N)
#pragma unroll
            for (int m = 0; m < R; ++m) dzs[(rg * R + m) * DP + n] = rg * R + m < M ? (P.bn ? sc * (g[m] - c1 - xh[m] * c2) : g[m]) : 0.f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *arow = dzs + l31 * DP + lh;
    // (all sixteen steps' weights requested at the top of the kernel, ahead of pass 1: 11.1 -> 15.0 us -- sixteen more live registers
    // spill at the 128 a 1,024-thread workgroup leaves a lane)
    for (int nb0 = 2 * wave; nb0 < ((FCF_EXP & 2) ? 0 : N); nb0 += 32 * 8) {   // eight steps' weights requested together
        float wr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[j] = wcol[(size_t)min(nb0 + 32 * j + lh, N - 1) * K];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = nb0 + 32 * j + lh;
            const float a = n < N ? arow[min(nb0 + 32 * j, N - 2)] : 0.f, b = (colok && n < N) ? wr[j] : 0.f;
            if (nb0 + 32 * j < N) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);   // (uniform)
        }
    }
    __syncthreads();   // every wave is done with the image: the partial tiles take its place
    float *red = fl;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    {   // one thread per output element: the sixteen partial tiles in wave order
        const int r = tid >> 6, ln = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[(w * 16 + r) * 64 + ln];
        const int row = (r & 3) + 4 * (ln >> 5) + 8 * (r >> 2), col = k0 + (ln & 31);
        if (row < M && col < K) P.dx[(size_t)row * K + col] = t;
    }
}

// ---- backward, LayerNorm: one block per row writes dz and g; column sums in a second kernel ---------
__global__ void __launch_bounds__(256)
fc_bwd_ln_rows_kernel(const float *__restrict__ dy, const float *__restrict__ z, const float *__restrict__ b,
                      const float *__restrict__ nw, const float *__restrict__ nb, const uint8_t *__restrict__ mask,
                      float drop_scale, int relu, const float *__restrict__ mean, const float *__restrict__ istd, int N,
                      float *__restrict__ dz, float *__restrict__ gbuf) {
    __shared__ double red[2][4];
    const int m = blockIdx.x;
    const float mu = mean[m], is = istd[m];
    double s1 = 0.0, s2 = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const size_t i = (size_t)m * N + n;
        const float xh = ((z[i] + b[n]) - mu) * is;
        float g = dy[i];
        if (mask) g = mask[i] ? g * drop_scale : 0.f;
        if (relu && !(xh * nw[n] + nb[n] > 0.f)) g = 0.f;
        gbuf[i] = g;
        const double gh = (double)g * (double)nw[n];
        s1 += gh;
        s2 += gh * (double)xh;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s1 += shfl_xor_f64(s1, k), s2 += shfl_xor_f64(s2, k);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1, red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    const float c1 = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / N);
    const float c2 = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / N);
    for (int n = threadIdx.x; n < N; n += 256) {
        const size_t i = (size_t)m * N + n;
        const float xh = ((z[i] + b[n]) - mu) * is;
        dz[i] = is * (gbuf[i] * nw[n] - c1 - xh * c2);
    }
}

__global__ void __launch_bounds__(256)
fc_bwd_ln_cols_kernel(const float *__restrict__ gbuf, const float *__restrict__ dz, const float *__restrict__ z,
                      const float *__restrict__ b, const float *__restrict__ mean, const float *__restrict__ istd, int M, int N,
                      float *__restrict__ dnw, float *__restrict__ dnb, float *__restrict__ db) {
    __shared__ double red[8][3][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cl;
    double a = 0.0, c = 0.0, e = 0.0;
    if (n < N)
        for (int m = rl; m < M; m += 8) {
            const size_t i = (size_t)m * N + n;
            const float xh = ((z[i] + b[n]) - mean[m]) * istd[m];
            a += (double)gbuf[i] * (double)xh;
            c += (double)gbuf[i];
            e += (double)dz[i];
        }
    red[rl][0][cl] = a, red[rl][1][cl] = c, red[rl][2][cl] = e;
    __syncthreads();
    if (rl == 0 && n < N) {
        a = c = e = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += red[i][0][cl], c += red[i][1][cl], e += red[i][2][cl];
        if (dnw) dnw[n] = (float)a;
        if (dnb) dnb[n] = (float)c;
        if (db) db[n] = (float)e;
    }
}

// ---- narrow output layers (N <= 16, no norm): fc3 / head_pi / head_mu / head_kappa ------------------
// Too narrow for a 32-wide MFMA tile; these are K-long dot products, one wavefront per output element.
constexpr int FC_SMALL_N = 16;

__global__ void __launch_bounds__(256) fc_small_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ b, const uint8_t *__restrict__ mask,
                                                           float drop_scale, int relu, int K, int N, float *__restrict__ z,
                                                           float *__restrict__ y) {
    const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *xr = x + (size_t)m * K;
    for (int n = wave; n < N; n += 4) {
        const float *wr = w + (size_t)n * K;
        double acc = 0.0;
        for (int k = lane; k < K; k += 64) acc += (double)xr[k] * (double)wr[k];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc += shfl_xor_f64(acc, s);
        if (lane == 0) {
            const float zz = (float)acc;  // bias-free pre-activation, like the MFMA path keeps it
            z[(size_t)m * N + n] = zz;
            float v = zz + b[n];
            if (relu) v = fmaxf(v, 0.f);
            if (mask) v = mask[(size_t)m * N + n] ? v * drop_scale : 0.f;
            y[(size_t)m * N + n] = v;
        }
    }
}

__device__ __forceinline__ float fc_small_dz(const float *dy, const float *z, const float *b, const uint8_t *mask,
                                             float drop_scale, int relu, size_t i, int n) {
    float g = dy[i];
    if (mask) g = mask[i] ? g * drop_scale : 0.f;
    if (relu && !(z[i] + b[n] > 0.f)) g = 0.f;
    return g;
}

// One launch for the whole backward of a narrow layer.  Workgroups [0, N * kparts): dW[n][k] = sum_m dz[m][n] x[m][k]
// for one n and one 256-wide k range (the first range of every n also writes db[n] = sum_m dz[m][n]); the dz column
// is staged in LDS once, so the m loop is independent loads.  Workgroups after those: dx[m][:] = sum_n dz[m][n] w[n][:].
__global__ void __launch_bounds__(256) fc_small_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ z,
                                                           const float *__restrict__ b, const uint8_t *__restrict__ mask,
                                                           float drop_scale, int relu, const float *__restrict__ x,
                                                           const float *__restrict__ w, int M, int K, int N, int kparts,
                                                           float *__restrict__ dw, float *__restrict__ db,
                                                           float *__restrict__ dx) {
    __shared__ float dzc[1024];
    const int ndw = N * kparts;
    if ((int)blockIdx.x < ndw) {
        const int n = blockIdx.x / kparts, k = (blockIdx.x % kparts) * 256 + threadIdx.x;
        double acc = 0.0, accb = 0.0;
        for (int m0 = 0; m0 < M; m0 += 1024) {  // M rows in LDS-sized pieces (a training batch is one piece)
            const int mc = min(1024, M - m0);
            __syncthreads();
            for (int m = threadIdx.x; m < mc; m += 256)
                dzc[m] = fc_small_dz(dy, z, b, mask, drop_scale, relu, (size_t)(m0 + m) * N + n, n);
            __syncthreads();
            if (k < K) {
                const float *xc = x + (size_t)m0 * K + k;
#pragma unroll 8
                for (int m = 0; m < mc; ++m) acc += (double)dzc[m] * (double)xc[(size_t)m * K];
            }
            if (threadIdx.x == 0 && blockIdx.x % kparts == 0)
                for (int m = 0; m < mc; ++m) accb += (double)dzc[m];
        }
        if (k < K) dw[(size_t)n * K + k] = (float)acc;
        if (threadIdx.x == 0 && blockIdx.x % kparts == 0) db[n] = (float)accb;
        return;
    }
    if (!dx) return;
    const int m = blockIdx.x - ndw;
    float g[FC_SMALL_N];
#pragma unroll
    for (int n = 0; n < FC_SMALL_N; ++n)
        g[n] = fc_small_dz(dy, z, b, mask, drop_scale, relu, (size_t)m * N + min(n, N - 1), min(n, N - 1)) * (n < N ? 1.f : 0.f);
    for (int k = threadIdx.x; k < K; k += 256) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < FC_SMALL_N; ++n)
            if (n < N) acc = fmaf(g[n], w[(size_t)n * K + k], acc);
        dx[(size_t)m * K + k] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
struct FcSaved {
    float *z, *mean, *istd, *scale, *shift;
    size_t bytes;
};
static FcSaved fc_saved_layout(const pnpp_fc_desc *d, void *base) {
    Carver cv(base);
    FcSaved s;
    const int mx = d->M > d->N ? d->M : d->N;
    s.z = cv.take<float>((size_t)d->M * d->N);
    s.mean = cv.take<float>(mx);
    s.istd = cv.take<float>(mx);
    s.scale = cv.take<float>(d->N);
    s.shift = cv.take<float>(d->N);
    s.bytes = cv.bytes();
    return s;
}
struct FcScratch {
    float *dz, *gbuf, *dwslab;
    double *slab;
    size_t bytes;
};
static FcScratch fc_scratch_layout(const pnpp_fc_desc *d, void *base) {
    Carver cv(base);
    FcScratch s;
    s.slab = cv.take<double>((size_t)kMaxStatBlocks * 2 * d->N);
    s.dz = cv.take<float>((size_t)d->M * d->N);
    s.gbuf = cv.take<float>((size_t)d->M * d->N);
    int nsplit, kp_pad;
    dw_plan(d->M, d->N, d->K, &nsplit, &kp_pad);
    s.dwslab = cv.take<float>((size_t)nsplit * d->N * kp_pad);
    s.bytes = cv.bytes();
    return s;
}

static bool fc_is_small(const pnpp_fc_desc *d) { return d->norm == PNPP_NORM_NONE && d->N <= FC_SMALL_N; }

static int fc_check(const pnpp_fc_desc *d) {
    PNPP_REQUIRE(d, PNPP_ERR_ARG, "fc: null descriptor");
    PNPP_REQUIRE(d->M > 0 && d->K > 0 && d->N > 0, PNPP_ERR_ARG, "fc: non-positive size M=%d K=%d N=%d", d->M, d->K, d->N);
    if (!fc_is_small(d))
        PNPP_REQUIRE(d->K % 4 == 0 && d->N % 4 == 0, PNPP_ERR_ARG, "fc: K=%d and N=%d must be multiples of 4", d->K, d->N);
    PNPP_REQUIRE(d->norm >= PNPP_NORM_NONE && d->norm <= PNPP_NORM_LAYER, PNPP_ERR_ARG, "fc: bad norm kind %d", d->norm);
    return PNPP_OK;
}

static int fc_forward_impl(const pnpp_fc_desc *d, const pnpp_fc_fwd_args *a, hipStream_t st) {
    PNPP_TRY(fc_check(d));
    PNPP_REQUIRE(a && a->x && a->w && a->b && a->y && a->saved && a->scratch, PNPP_ERR_ARG, "fc_forward: null pointer");
    if (d->norm != PNPP_NORM_NONE) PNPP_REQUIRE(a->nw && a->nb, PNPP_ERR_ARG, "fc_forward: norm affine parameters are null");
    if (d->norm == PNPP_NORM_BATCH) PNPP_REQUIRE(a->rm && a->rv, PNPP_ERR_ARG, "fc_forward: running statistics are null");
    const FcSaved sv = fc_saved_layout(d, a->saved);
    const FcScratch sc = fc_scratch_layout(d, a->scratch);
    PNPP_REQUIRE(!a->mask_out || (d->training && !fc_is_small(d) && ((d->norm == PNPP_NORM_BATCH && d->M <= 32 && !stats_sync_on()) ||
                                                                     d->norm == PNPP_NORM_LAYER)),
                 PNPP_ERR_ARG, "fc_forward: the in-kernel dropout draw exists for the BatchNorm epilogue (training, M <= 32, no statistics "
                               "exchange) and for the LayerNorm block (training) only");
    if (a->mask_out)
        PNPP_REQUIRE(!a->mask && a->rng_counter && a->drop_p > 0.f && a->drop_p < 1.f, PNPP_ERR_ARG,
                     "fc_forward: a drawn dropout mask needs rng_counter, 0 < drop_p < 1 and no given mask");

    if (fc_is_small(d)) {
        ProfScope ps(st, "fc_small_fwd_kernel M=%d N=%d K=%d", d->M, d->N, d->K);
        hipLaunchKernelGGL(fc_small_fwd_kernel, dim3(d->M), dim3(256), 0, st, a->x, a->w, a->b, a->mask, d->drop_scale, d->relu,
                           d->K, d->N, sv.z, a->y);
        PNPP_CHECK_LAUNCH("fc_forward(small)");
        return PNPP_OK;
    }

    BOperand W;  // the linear weight (N x K) is read in place
    W.b = a->w;
    W.ldb = d->K;
    W.trans = 1;
    W.rows = d->K;
    AOperand A;
    A.mode = A_PLAIN;
    A.a = a->x;
    A.lda = d->K;
    Epilogue E;
    E.c = sv.z;
    E.ldc = d->N;
    const size_t total = (size_t)d->M * d->N;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (d->norm == PNPP_NORM_BATCH) {
        int nslab = 0;
        if (d->training) PNPP_REQUIRE(d->M > 1, PNPP_ERR_ARG, "Expected more than 1 value per channel when training");  // torch's message
        if (d->training && d->M <= 32 && !stats_sync_on()) {  // the whole batch fits one tile: statistics, affine, ReLU, dropout in the GEMM epilogue
            E.mode = E_BN_APPLY;
            BnTail &T = E.bn;
            T.bias = a->b, T.gamma = a->nw, T.beta = a->nb;
            T.rm = a->rm, T.rv = a->rv, T.nbt = (long long *)a->nbt;
            T.momentum = d->momentum, T.eps = d->eps;
            T.mean = sv.mean, T.istd = sv.istd, T.scale = sv.scale, T.shift = sv.shift;
            T.mask = a->mask, T.drop_scale = d->drop_scale, T.relu = d->relu, T.y = a->y;
            if (a->mask_out) {
                PNPP_REQUIRE(!a->mask && a->rng_counter && a->drop_p > 0.f && a->drop_p < 1.f, PNPP_ERR_ARG,
                             "fc_forward: a drawn dropout mask needs rng_counter, 0 < drop_p < 1 and no given mask");
                T.mask_out = a->mask_out, T.drop_p = a->drop_p, T.rng_seed = a->rng_seed;
                T.rng_counter = reinterpret_cast<unsigned long long *>(a->rng_counter);
            }
            return launch_gemm(A, W, d->M, d->N, d->K, E, nullptr, st);
        }
        if (d->training) {
            E.mode = E_STORE_STATS;
            E.slab = sc.slab;
            PNPP_TRY(launch_gemm(A, W, d->M, d->N, d->K, E, &nslab, st));
        } else {
            E.mode = E_STORE;
            PNPP_TRY(launch_gemm(A, W, d->M, d->N, d->K, E, nullptr, st));
        }
        StatsView V;
        V.slab = sc.slab, V.nslab = nslab;
        if (d->training) PNPP_TRY(stats_exchange(sc.slab, nslab, d->N, (double)d->M, st, &V));
        PNPP_TRY(launch_bn_finalize_fwd(V.slab, V.nslab, d->N, (double)d->M, a->b, a->nw, a->nb, a->rm, a->rv, (long long *)a->nbt, d->momentum, d->eps,
                                        d->training, sv.mean, sv.istd, sv.scale, sv.shift, st, V.count_dev));
        hipLaunchKernelGGL(fc_apply_cols_kernel, dim3(grid), dim3(256), 0, st, sv.z, sv.scale, sv.shift, a->mask, d->drop_scale,
                           d->relu, d->M, d->N, a->y);
    } else {
        E.mode = E_STORE;
        PNPP_TRY(launch_gemm(A, W, d->M, d->N, d->K, E, nullptr, st));
        if (d->norm == PNPP_NORM_LAYER) {
            hipLaunchKernelGGL(fc_apply_ln_kernel, dim3(d->M), dim3(256), 0, st, sv.z, a->b, a->nw, a->nb, a->mask, d->drop_scale,
                               d->relu, d->N, d->eps, a->y, sv.mean, sv.istd, a->mask_out, a->drop_p, (unsigned long long)a->rng_seed,
                               reinterpret_cast<unsigned long long *>(a->rng_counter));
        } else {
            ProfScope ps(st, "fc_apply_cols_kernel M=%d N=%d", d->M, d->N);
            hipLaunchKernelGGL(fc_apply_cols_kernel, dim3(grid), dim3(256), 0, st, sv.z, (const float *)nullptr, a->b, a->mask,
                               d->drop_scale, d->relu, d->M, d->N, a->y);
        }
    }
    PNPP_CHECK_LAUNCH("fc_forward");
    return PNPP_OK;
}

// dW (N,K) = dz^T x for a batch of at most 32 rows (the head layers): an outer-product accumulation with no reduction
// worth an MFMA tile.  Workgroup = 32 output rows (n) x 128 output columns (k): dz tile and x tile in LDS, each thread
// owns one k and sixteen n.  Stores are 512-byte rows.
__global__ void __launch_bounds__(256) dw_fewrows_kernel(const float *__restrict__ dz, const float *__restrict__ x, int M, int N,
                                                         int K, float *__restrict__ dw) {
    __shared__ __attribute__((aligned(16))) float dzs[32][32];
    __shared__ float xs[32][128];
    const int kl = threadIdx.x & 127, nh = threadIdx.x >> 7;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 32;
    for (int f = threadIdx.x; f < 32 * 32; f += 256) {
        const int m = f >> 5, n = f & 31;
        dzs[m][n] = (m < M && n0 + n < N) ? dz[(size_t)m * N + n0 + n] : 0.f;
    }
    for (int f = threadIdx.x; f < 32 * 128; f += 256) {
        const int m = f >> 7, k = f & 127;
        xs[m][k] = (m < M && k0 + k < K) ? x[(size_t)m * K + k0 + k] : 0.f;
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int m = 0; m < 32; ++m) {
        const float xv = xs[m][kl];
        const float4 *d4 = reinterpret_cast<const float4 *>(&dzs[m][16 * nh]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 d = d4[q];
            acc[4 * q] = fmaf(d.x, xv, acc[4 * q]), acc[4 * q + 1] = fmaf(d.y, xv, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(d.z, xv, acc[4 * q + 2]), acc[4 * q + 3] = fmaf(d.w, xv, acc[4 * q + 3]);
        }
    }
    if (k0 + kl < K)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = n0 + 16 * nh + j;
            if (n < N) dw[(size_t)n * K + k0 + kl] = acc[j];
        }
}

static int fc_backward_impl(const pnpp_fc_desc *d, const pnpp_fc_bwd_args *a, hipStream_t st) {
    PNPP_TRY(fc_check(d));
    PNPP_REQUIRE(a && a->x && a->w && a->b && a->dy && a->saved && a->scratch && a->dw && a->db, PNPP_ERR_ARG,
                 "fc_backward: null pointer");
    if (d->norm != PNPP_NORM_NONE) PNPP_REQUIRE(a->nw && a->nb, PNPP_ERR_ARG, "fc_backward: norm affine parameters are null");
    const FcSaved sv = fc_saved_layout(d, const_cast<void *>(a->saved));
    const FcScratch sc = fc_scratch_layout(d, a->scratch);

    if (fc_is_small(d)) {
        ProfScope ps(st, "fc_small_bwd M=%d N=%d K=%d", d->M, d->N, d->K);
        const int kparts = cdiv(d->K, 256);
        hipLaunchKernelGGL(fc_small_bwd_kernel, dim3(d->N * kparts + (a->dx ? d->M : 0)), dim3(256), 0, st, a->dy, sv.z, a->b,
                           a->mask, d->drop_scale, d->relu, a->x, a->w, d->M, d->K, d->N, kparts, a->dw, a->db, a->dx);
        PNPP_CHECK_LAUNCH("fc_backward(small)");
        return PNPP_OK;
    }

    // 1. dz = d loss / d (x W^T)  (M x N) and the parameter gradients of the normalisation
    if (d->norm == PNPP_NORM_LAYER) {
        hipLaunchKernelGGL(fc_bwd_ln_rows_kernel, dim3(d->M), dim3(256), 0, st, a->dy, sv.z, a->b, a->nw, a->nb, a->mask,
                           d->drop_scale, d->relu, sv.mean, sv.istd, d->N, sc.dz, sc.gbuf);
        hipLaunchKernelGGL(fc_bwd_ln_cols_kernel, dim3(cdiv(d->N, 32)), dim3(256), 0, st, sc.gbuf, sc.dz, sv.z, a->b, sv.mean,
                           sv.istd, d->M, d->N, a->dnw, a->dnb, a->db);
    } else if (d->norm == PNPP_NORM_NONE && d->M > 4096) {
        const int chunks = cdiv(d->M, 256);  // sc.gbuf (M x N floats) holds the [chunks][N] partials
        {
            ProfScope ps(st, "fc_bwd_rows_kernel M=%d N=%d", d->M, d->N);
            hipLaunchKernelGGL(fc_bwd_rows_kernel, dim3(cdiv(d->N, 64), chunks), dim3(256), 0, st, a->dy, sv.z, a->mask,
                               d->drop_scale, d->relu, a->b, d->M, d->N, sc.dz, sc.gbuf);
            PNPP_CHECK_LAUNCH("fc_backward(rows)");
        }
        PNPP_TRY(launch_slab_reduce(sc.gbuf, chunks, 1, d->N, d->N, -1, a->db, d->N, st));
    } else {
        const int bn = d->norm == PNPP_NORM_BATCH;
        // head blocks of at most 32 rows with a gradient to pass on: dz is never written (fc_bwd_fused_kernel); PNPP_NO_FC_FUSED=1 keeps
        // the two launches
        static const bool fused_on = !(getenv("PNPP_NO_FC_FUSED") && atoi(getenv("PNPP_NO_FC_FUSED")) != 0);
        if (fused_on && a->dx && d->M <= 32 && d->N >= 256 && d->N % 2 == 0 && !(bn && d->training && stats_sync_on())) {
            FcFusedArgs P;
            P.dy = a->dy, P.z = sv.z, P.mask = a->mask, P.drop_scale = d->drop_scale, P.relu = d->relu, P.bn = bn, P.training = d->training;
            P.scale = sv.scale, P.shift = sv.shift, P.mean = sv.mean, P.istd = sv.istd, P.bias = a->b;
            P.w = a->w, P.x = a->x, P.M = d->M, P.N = d->N, P.K = d->K;
            P.dx = a->dx, P.dw = a->dw, P.dnw = a->dnw, P.dnb = a->dnb, P.db = a->db;
            P.g1 = cdiv(d->K, 32), P.gx2 = cdiv(d->K, 128);
            const int gy2 = cdiv(d->N, 32);
            const int tpc = d->N <= 256 ? 4 : d->N <= 512 ? 2 : 1;
            const size_t img = (size_t)32 * (d->N + 1) + 2 + (size_t)tpc * 2 * (1024 / tpc) * 2, part = (size_t)16 * 16 * 64;
            const size_t lds_dx = (img > part ? img : part) * sizeof(float), lds_dw = (size_t)(32 * 32 + 32 * 128 + 2 * 32 * 32) * sizeof(float);
            const size_t lds = lds_dx > lds_dw ? lds_dx : lds_dw;
            if (lds <= 160 * 1024) {
                static size_t granted[3] = {0, 0, 0};
                auto go = [&](auto kfn, int slot) {
                    if (lds > 48 * 1024 && lds > granted[slot]) {
                        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        granted[slot] = lds;
                    }
                    hipLaunchKernelGGL(kfn, dim3(P.g1 + P.gx2 * gy2), dim3(1024), lds, st, P);
                };
                ProfScope ps(st, "fc_bwd_fused_kernel M=%d N=%d K=%d grid=%d+%d", d->M, d->N, d->K, P.g1, P.gx2 * gy2);
                if (tpc == 4) go(fc_bwd_fused_kernel<4>, 0);
                else if (tpc == 2) go(fc_bwd_fused_kernel<2>, 1);
                else go(fc_bwd_fused_kernel<1>, 2);
                PNPP_CHECK_LAUNCH("fc_backward(fused)");
                return PNPP_OK;
            }
        }
        if (bn && d->training && stats_sync_on()) {   // SyncBN: sums -> exchange over the ranks -> apply
            hipLaunchKernelGGL(fc_bwd_cols_kernel, dim3(cdiv(d->N, 32)), dim3(256), 0, st, a->dy, sv.z, a->mask, d->drop_scale,
                               d->relu, bn, sv.scale, sv.shift, sv.mean, sv.istd, a->b, d->M, d->N, d->training, sc.dz, a->dnw,
                               a->dnb, a->db, 1, stats_buffer_global(), stats_buffer_local());
            PNPP_CHECK_LAUNCH("fc_backward(sums)");
            StatsView V;
            PNPP_TRY(stats_exchange_inplace(d->N, st, &V));
            hipLaunchKernelGGL(fc_bwd_cols_kernel, dim3(cdiv(d->N, 32)), dim3(256), 0, st, a->dy, sv.z, a->mask, d->drop_scale,
                               d->relu, bn, sv.scale, sv.shift, sv.mean, sv.istd, a->b, d->M, d->N, d->training, sc.dz, a->dnw,
                               a->dnb, a->db, 2, stats_buffer_global(), stats_buffer_local());
        } else {
            hipLaunchKernelGGL(fc_bwd_cols_kernel, dim3(cdiv(d->N, 32)), dim3(256), 0, st, a->dy, sv.z, a->mask, d->drop_scale,
                               d->relu, bn, sv.scale, sv.shift, sv.mean, sv.istd, a->b, d->M, d->N, d->training, sc.dz, a->dnw,
                               a->dnb, a->db, 0, (double *)nullptr, (double *)nullptr);
        }
    }
    PNPP_CHECK_LAUNCH("fc_backward");

    // 2. dW = dz^T x
    AOperand dz;
    dz.mode = A_PLAIN;
    dz.a = sc.dz;
    dz.lda = d->N;
    AOperand x;
    x.mode = A_PLAIN;
    x.a = a->x;
    x.lda = d->K;
    int nsplit, kp_pad;
    dw_plan(d->M, d->N, d->K, &nsplit, &kp_pad);
    {
        int rc = PNPP_OK;  // head layers with a gradient to pass on: dW and dx = dz W in one launch
        if (a->dx && try_launch_fc_dx_dw(sc.dz, a->w, a->x, d->M, d->N, d->K, a->dx, a->dw, st, &rc)) return rc;
    }
    if (d->M <= 32) {  // a handful of rows: outer-product kernel, written in place
        ProfScope ps(st, "dw_fewrows_kernel M=%d N=%d K=%d", d->M, d->N, d->K);
        hipLaunchKernelGGL(dw_fewrows_kernel, dim3(cdiv(d->K, 128), cdiv(d->N, 32)), dim3(256), 0, st, sc.dz, a->x, d->M, d->N, d->K,
                           a->dw);
        PNPP_CHECK_LAUNCH("fc_backward(dw)");
    } else if (nsplit == 1 && kp_pad == d->K) {  // a single partial in the gradient's own layout: write it in place
        PNPP_TRY(launch_dw(dz, d->N, x, d->K, d->M, a->dw, 1, kp_pad, st));
    } else {
        PNPP_TRY(launch_dw(dz, d->N, x, d->K, d->M, sc.dwslab, nsplit, kp_pad, st));
        PNPP_TRY(launch_slab_reduce(sc.dwslab, nsplit, d->N, kp_pad, d->K, -1, a->dw, d->K, st));
    }

    // 3. dx = dz W
    if (a->dx) {
        Epilogue E;
        E.mode = E_STORE;
        E.c = a->dx;
        E.ldc = d->K;
        BOperand W;
        W.b = a->w;
        W.ldb = d->K;
        W.rows = d->N;
        PNPP_TRY(launch_gemm(dz, W, d->M, d->K, d->N, E, nullptr, st));
    }
    return PNPP_OK;
}

unsigned fc_build_flags() { return (FCF_EXP != 0) ? 32u : 0u; }

}  // namespace pnpp

using namespace pnpp;

extern "C" size_t pnpp_fc_saved_bytes(const pnpp_fc_desc *d) { return fc_check(d) == PNPP_OK ? fc_saved_layout(d, nullptr).bytes : 0; }
extern "C" size_t pnpp_fc_scratch_bytes(const pnpp_fc_desc *d) {
    return fc_check(d) == PNPP_OK ? fc_scratch_layout(d, nullptr).bytes : 0;
}
extern "C" int pnpp_fc_forward(const pnpp_fc_desc *d, const pnpp_fc_fwd_args *a, void *stream) {
    return fc_forward_impl(d, a, as_stream(stream));
}
extern "C" int pnpp_fc_backward(const pnpp_fc_desc *d, const pnpp_fc_bwd_args *a, void *stream) {
    return fc_backward_impl(d, a, as_stream(stream));
}
