// gemm_kernels.hip -- the grouped per-point MLP of PointNetSetAbstraction as fused GEMMs on the
// gfx950 matrix cores (reference: models/pointnet_pp_8dir.py:29-43 and its autograd backward).
//
// Arithmetic is exact float32: v_mfma_f32_32x32x2_f32 (one rounding per product, k-ordered fmaf
// chain; MI355X_MICROARCH "Matrix cores"), statistics are accumulated in float64.
//
// Layout: every activation tensor is row-major (rows = (cloud, centre, neighbour), channels
// contiguous), so a 1x1 Conv2d over (B,C,npoint,nsample) is C[M x N] = A[M x K] * W^T.
// What would be separate gather / concat / BatchNorm-apply / ReLU / BatchNorm-backward passes over
// HBM is folded into the A-operand loader of the consuming GEMM; column statistics, ReLU masks and
// the stores are folded into the epilogue of the producing GEMM.
//
// MFMA operand maps (wave64, 32x32x2 f32):  A: lane l holds A[i=l&31][k=l>>5]
//                                           B: lane l holds B[k=l>>5][j=l&31]
//                                           D: lane l, reg r: col j=l&31, row i=(r&3)+8*(r>>2)+4*(l>>5)
#include "kernels.h"

namespace pnpp {

constexpr int KC = 32;       // reduction-dim chunk staged in LDS per step
constexpr int APITCH = KC + 1;  // odd pitch: the 32 rows a half-wave reads land on 32 different banks

// ---------------------------------------------------------------------------------------------
// A-operand loaders: four consecutive k of one row, split into a raw fetch (global loads only, so
// the next chunk's loads can be in flight while the current chunk is in the MFMA loop) and a
// transform applied when the chunk is written to LDS.
// ---------------------------------------------------------------------------------------------
struct RawA {
    float4 p, q;  // p: primary values; q: z (A_DZ) or the centre coordinates to subtract (A_GATHER xyz part)
};

template <int MODE>
__device__ __forceinline__ RawA fetch_a4(const AOperand &A, int row, int k, int M, int Kd) {
    RawA r;
    r.p = make_float4(0.f, 0.f, 0.f, 0.f);
    r.q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= M || k >= Kd) return r;
    if constexpr (MODE == A_PLAIN || MODE == A_BNRELU) {
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)row * A.lda + k);
    } else if constexpr (MODE == A_DZ) {
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)row * A.lda + k);
        r.q = *reinterpret_cast<const float4 *>(A.z + (size_t)row * A.lda + k);
    } else {  // A_GATHER / A_CONCAT: features first, then xyz (relative to the centre when gathering)
        size_t prow = (size_t)row, grp = 0;
        if constexpr (MODE == A_GATHER) {
            grp = (size_t)(row / A.K);  // centre row (b*S + s)
            prow = (size_t)(grp / A.S) * A.N + A.idx[row];
        }
        if ((A.D & 3) == 0 && k + 3 < A.D) {
            r.p = *reinterpret_cast<const float4 *>(A.a + prow * A.D + k);
            return r;
        }
        float pv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k + i;
            if (kk < A.D) {
                pv[i] = A.a[prow * A.D + kk];
            } else if (kk < A.D + 3) {
                pv[i] = A.xyz[prow * 3 + (kk - A.D)];
                if constexpr (MODE == A_GATHER) qv[i] = A.new_xyz[grp * 3 + (kk - A.D)];
            }
        }
        r.p = make_float4(pv[0], pv[1], pv[2], pv[3]);
        r.q = make_float4(qv[0], qv[1], qv[2], qv[3]);
    }
    return r;
}

template <int MODE>
__device__ __forceinline__ void xform_a4(const AOperand &A, const RawA &r, int row, int k, int M, int Kd, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (row >= M || k >= Kd) return;  // padding rows/columns must be exact zeros AFTER the transform
    if constexpr (MODE == A_PLAIN) {
        v[0] = r.p.x, v[1] = r.p.y, v[2] = r.p.z, v[3] = r.p.w;
    } else if constexpr (MODE == A_BNRELU) {
        const float4 s = *reinterpret_cast<const float4 *>(A.scale + k);
        const float4 h = *reinterpret_cast<const float4 *>(A.shift + k);
        v[0] = fmaxf(fmaf(r.p.x, s.x, h.x), 0.f);
        v[1] = fmaxf(fmaf(r.p.y, s.y, h.y), 0.f);
        v[2] = fmaxf(fmaf(r.p.z, s.z, h.z), 0.f);
        v[3] = fmaxf(fmaf(r.p.w, s.w, h.w), 0.f);
    } else if constexpr (MODE == A_GATHER || MODE == A_CONCAT) {
        // float32 subtraction of the centre, pointnet_pp_8dir.py:32 (q = 0 for features and for group_all)
        v[0] = __fsub_rn(r.p.x, r.q.x), v[1] = __fsub_rn(r.p.y, r.q.y);
        v[2] = __fsub_rn(r.p.z, r.q.z), v[3] = __fsub_rn(r.p.w, r.q.w);
    } else {  // A_DZ
        const float *c = A.cst + k;
        const float4 g = *reinterpret_cast<const float4 *>(c);
        const float4 mu = *reinterpret_cast<const float4 *>(c + A.C);
        const float4 is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        const float4 c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C);
        const float4 c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
        v[0] = g.x * (r.p.x - c1.x - (r.q.x - mu.x) * is.x * c2.x);
        v[1] = g.y * (r.p.y - c1.y - (r.q.y - mu.y) * is.y * c2.y);
        v[2] = g.z * (r.p.z - c1.z - (r.q.z - mu.z) * is.z * c2.z);
        v[3] = g.w * (r.p.w - c1.w - (r.q.w - mu.w) * is.w * c2.w);
    }
}

// scalar flavour used by the dW kernel (one element per lane: the channel index sits on the lane, the
// per-channel constants are hoisted into registers once per wave)
struct ChanConst {
    float g, mu, is, c1, c2;  // A_DZ
    float sc, sh;             // A_BNRELU
};

template <int MODE>
__device__ __forceinline__ ChanConst load_chan_const(const AOperand &A, int k, int Kvalid) {
    ChanConst c{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k >= Kvalid) return c;
    if constexpr (MODE == A_DZ) {
        const float *p = A.cst + k;
        c.g = p[0], c.mu = p[A.C], c.is = p[2 * A.C], c.c1 = p[3 * A.C], c.c2 = p[4 * A.C];
    } else if constexpr (MODE == A_BNRELU) {
        c.sc = A.scale[k], c.sh = A.shift[k];
    }
    return c;
}

// raw loads of element (row, k): up to two values (second one: z for A_DZ, centre coordinate for A_GATHER)
template <int MODE>
__device__ __forceinline__ float2 fetch_a1(const AOperand &A, int row, int k, int Kvalid, bool ok) {
    float2 r = make_float2(0.f, 0.f);
    if (!ok || k >= Kvalid) return r;
    if constexpr (MODE == A_PLAIN || MODE == A_BNRELU) {
        r.x = A.a[(size_t)row * A.lda + k];
    } else if constexpr (MODE == A_DZ) {
        r.x = A.a[(size_t)row * A.lda + k];
        r.y = A.z[(size_t)row * A.lda + k];
    } else if constexpr (MODE == A_GATHER) {
        const int grp = row / A.K;
        const size_t prow = (size_t)(grp / A.S) * A.N + A.idx[row];
        if (k < A.D) {
            r.x = A.a[prow * A.D + k];
        } else {
            r.x = A.xyz[prow * 3 + (k - A.D)];
            r.y = A.new_xyz[(size_t)grp * 3 + (k - A.D)];
        }
    } else {  // A_CONCAT
        r.x = k < A.D ? A.a[(size_t)row * A.D + k] : A.xyz[(size_t)row * 3 + (k - A.D)];
    }
    return r;
}

template <int MODE>
__device__ __forceinline__ float xform_a1(const float2 r, const ChanConst &c, int k, int Kvalid, bool ok) {
    if (!ok || k >= Kvalid) return 0.f;
    if constexpr (MODE == A_PLAIN) {
        return r.x;
    } else if constexpr (MODE == A_BNRELU) {
        return fmaxf(fmaf(r.x, c.sc, c.sh), 0.f);
    } else if constexpr (MODE == A_DZ) {
        return c.g * (r.x - c.c1 - (r.y - c.mu) * c.is * c.c2);
    } else {
        return __fsub_rn(r.x, r.y);
    }
}

// ---------------------------------------------------------------------------------------------
// fused GEMM: persistent row-tile workers (grid.x) x column tiles (grid.y)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int AMODE, int EMODE>
__global__ void __launch_bounds__(WM * WN * 64, 2)
gemm_kernel(const AOperand A, const float *__restrict__ Bm, int ldb, int M, int Nout, int Kd, const Epilogue E) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int NTHR = WM * WN * 64;
    static_assert(TM % 32 == 0 && TN % 32 == 0 && (BM * (KC / 4)) % NTHR == 0 && (KC * (BN / 4)) % NTHR == 0, "tile configuration");
    __shared__ __attribute__((aligned(16))) float lds[BM * APITCH + KC * BN];
    float *As = lds, *Bs = lds + BM * APITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;

    double s1[NT], s2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) s1[i] = s2[i] = 0.0;

    const int tiles = (M + BM - 1) / BM;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // software pipeline: chunk k0+KC is fetched into registers while chunk k0 is in the MFMA loop
        constexpr int NA = BM * (KC / 4) / NTHR, NB = KC * (BN / 4) / NTHR;
        RawA ra[NA];
        float4 rb[NB];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * NTHR;
                ra[i] = fetch_a4<AMODE>(A, m0 + f / (KC / 4), k0 + 4 * (f % (KC / 4)), M, Kd);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * NTHR;
                const int kk = f / (BN / 4), jq = f % (BN / 4);
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kk < Kd && n0 + 4 * jq < Nout)
                    rb[i] = *reinterpret_cast<const float4 *>(Bm + (size_t)(k0 + kk) * ldb + n0 + 4 * jq);
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < Kd; k0 += KC) {
            __syncthreads();  // the previous chunk has been consumed
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * NTHR;
                const int r = f / (KC / 4), q = f % (KC / 4);
                float v[4];
                xform_a4<AMODE>(A, ra[i], m0 + r, k0 + 4 * q, M, Kd, v);
                float *d = As + r * APITCH + 4 * q;
                d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * NTHR;
                *reinterpret_cast<float4 *>(Bs + (f / (BN / 4)) * BN + 4 * (f % (BN / 4))) = rb[i];
            }
            __syncthreads();
            if (k0 + KC < Kd) fetch(k0 + KC);
            const int ksteps = min(KC, Kd - k0) >> 1;
            const float *ap = As + (wm * TM + l31) * APITCH + lh;
            const float *bp = Bs + lh * BN + wn * TN + l31;
            if (ksteps == KC / 2) {
#pragma unroll 4
                for (int s = 0; s < KC / 2; ++s) {
                    float a[MT], b[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = ap[i * 32 * APITCH + 2 * s];
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j] = bp[2 * s * BN + j * 32];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            } else {
                for (int s = 0; s < ksteps; ++s) {
                    float a[MT], b[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = ap[i * 32 * APITCH + 2 * s];
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j] = bp[2 * s * BN + j * 32];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
        }

        // epilogue: each register is one row; a half-wave writes 32 consecutive floats (128 B)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + wn * TN + j * 32 + l31;
                float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
                if constexpr (EMODE == E_MASK_STATS) {
                    if (col < Nout) sc = E.scale[col], sh = E.shift[col], mu = E.mu[col], is = E.istd[col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < M && col < Nout) {
                        float v = acc[i][j][r];
                        if constexpr (EMODE == E_STORE_STATS) {
                            s1[j] += (double)v;
                            s2[j] += (double)v * (double)v;
                        } else if constexpr (EMODE == E_MASK_STATS) {
                            const float zp = E.zp[(size_t)row * E.ldc + col];
                            v = (fmaf(zp, sc, sh) > 0.f) ? v : 0.f;
                            s1[j] += (double)v;
                            s2[j] += (double)v * (double)((zp - mu) * is);
                        }
                        E.c[(size_t)row * E.ldc + col] = v;
                    }
                }
            }
    }

    if constexpr (EMODE != E_STORE) {
        // column partials: two lane halves -> WM waves (through LDS) -> one slab row per block.x
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);  // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            double a = s1[j] + shfl_xor_f64(s1[j], 32);
            double b = s2[j] + shfl_xor_f64(s2[j], 32);
            if (lh == 0) {
                const int cl = wn * TN + j * 32 + l31;
                red[(wm * 2 + 0) * BN + cl] = a;
                red[(wm * 2 + 1) * BN + cl] = b;
            }
        }
        __syncthreads();
        for (int f = tid; f < 2 * BN; f += NTHR) {
            const int which = f / BN, cl = f % BN;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + cl];
            if (n0 + cl < Nout) E.slab[((size_t)blockIdx.x * 2 + which) * Nout + n0 + cl] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small-M GEMM (fully connected head: M = batch rows).  One 32x32 output tile per workgroup; the
// reduction dimension is split over the 4 waves (chunk-interleaved), so a K=1024 layer is 8 chunks
// deep instead of 32.  A chunks go through wave-private LDS (row-major global -> lane-per-row
// operand), B (weights, [k][n] row-major) is read straight into the MFMA operand layout (the lane
// index is n: one 128-byte segment per half-wave).  Next chunk's loads fly during the MFMA loop.
// ---------------------------------------------------------------------------------------------
template <int EMODE>
__global__ void __launch_bounds__(256)
gemm_smallm_kernel(const float *__restrict__ Am, int lda, const float *__restrict__ Bm, int ldb, int M, int Nout, int Kd,
                   int Kb, const Epilogue E) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * APITCH + 4 * 32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    float *As = lds + wave * 32 * APITCH;
    float *part = lds + 4 * 32 * APITCH;  // [4][32][32]
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int nchunks = (Kd + KC - 1) / KC;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 na[4];
    float nb[KC / 2];
    auto fetch = [&](int c) {
        const int k0 = c * KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + (lane >> 3) + 8 * i, k = k0 + 4 * (lane & 7);
            na[i] = (row < M && k < Kd) ? *reinterpret_cast<const float4 *>(Am + (size_t)row * lda + k)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2) {
            const int k = k0 + 2 * s2 + lh;
            nb[s2] = (k < Kb && n0 + l31 < Nout) ? Bm[(size_t)k * ldb + n0 + l31] : 0.f;
        }
    };
    if (wave < nchunks) fetch(wave);
    for (int c = wave; c < nchunks; c += 4) {
        float cb[KC / 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = As + ((lane >> 3) + 8 * i) * APITCH + 4 * (lane & 7);
            d[0] = na[i].x, d[1] = na[i].y, d[2] = na[i].z, d[3] = na[i].w;
        }
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2) cb[s2] = nb[s2];
        if (c + 4 < nchunks) fetch(c + 4);
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[l31 * APITCH + 2 * s2 + lh], cb[s2], acc, 0, 0, 0);
    }
    // K-split reduction in fixed wave order, then the epilogue on the summed tile
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j;
        v[j] = (part[e] + part[1024 + e]) + (part[2048 + e] + part[3072 + e]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, row = m0 + e / 32, col = n0 + e % 32;
        part[e] = (row < M) ? v[j] : 0.f;
        if (row < M && col < Nout) E.c[(size_t)row * E.ldc + col] = v[j];
    }
    if constexpr (EMODE == E_STORE_STATS) {
        __syncthreads();
        if (tid < 32 && n0 + tid < Nout) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < 32; ++r) {
                const double x = (double)part[r * 32 + tid];
                s1 += x;
                s2 += x * x;
            }
            E.slab[((size_t)blockIdx.y * 2 + 0) * Nout + n0 + tid] = s1;
            E.slab[((size_t)blockIdx.y * 2 + 1) * Nout + n0 + tid] = s2;
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_cfg(const AOperand &A, const float *Bm, int ldb, int M, int Nout, int Kd, const Epilogue &E,
                           int *nslab, hipStream_t st) {
    const int tiles = cdiv(M, BM);
    const int gx = tiles < kMaxStatBlocks ? tiles : kMaxStatBlocks;
    const dim3 grid(gx, cdiv(Nout, BN)), block(WM * WN * 64);
    if (nslab) *nslab = gx;
    ProfScope ps(st, "gemm_kernel<%d,%d,%d,%d,A%d,E%d> M=%d N=%d K=%d", BM, BN, WM, WN, A.mode, E.mode, M, Nout, Kd);
#define PNPP_LAUNCH(AM, EM)                                                                                         \
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AM, EM>), grid, block, 0, st, A, Bm, ldb, M, Nout, Kd, E); \
    break;
#define PNPP_BY_E(AM)                                                          \
    switch (E.mode) {                                                          \
        case E_STORE: PNPP_LAUNCH(AM, E_STORE)                                 \
        case E_STORE_STATS: PNPP_LAUNCH(AM, E_STORE_STATS)                     \
        case E_MASK_STATS: PNPP_LAUNCH(AM, E_MASK_STATS)                       \
        default: set_error("gemm: bad epilogue mode %d", E.mode); return PNPP_ERR_ARG; \
    }                                                                          \
    break;
    switch (A.mode) {
        case A_PLAIN: PNPP_BY_E(A_PLAIN)
        case A_BNRELU: PNPP_BY_E(A_BNRELU)
        case A_GATHER: PNPP_BY_E(A_GATHER)
        case A_CONCAT: PNPP_BY_E(A_CONCAT)
        case A_DZ: PNPP_BY_E(A_DZ)
        default: set_error("gemm: bad A mode %d", A.mode); return PNPP_ERR_ARG;
    }
#undef PNPP_BY_E
#undef PNPP_LAUNCH
    PNPP_CHECK_LAUNCH("gemm");
    return PNPP_OK;
}

int launch_gemm(const AOperand &A, const float *Bm, int ldb, int M, int Nout, int Kd, const Epilogue &E, int *nslab,
                hipStream_t st) {
    PNPP_REQUIRE(M > 0 && Nout > 0 && Kd > 0, PNPP_ERR_ARG, "gemm: non-positive size M=%d N=%d K=%d", M, Nout, Kd);
    if (A.mode == A_PLAIN && E.mode != E_MASK_STATS && M <= 512 && cdiv(M, 32) <= kMaxStatBlocks) {
        PNPP_REQUIRE(Kd % 4 == 0 && A.lda % 4 == 0 && ((uintptr_t)A.a & 15) == 0, PNPP_ERR_ARG, "gemm(small M): A pitch/alignment");
        const dim3 grid(cdiv(Nout, 32), cdiv(M, 32));
        if (nslab) *nslab = grid.y;
        ProfScope ps(st, "gemm_smallm_kernel<E%d> M=%d N=%d K=%d", E.mode, M, Nout, Kd);
        if (E.mode == E_STORE_STATS)
            hipLaunchKernelGGL((gemm_smallm_kernel<E_STORE_STATS>), grid, dim3(256), 0, st, A.a, A.lda, Bm, ldb, M, Nout, Kd, Kd, E);
        else
            hipLaunchKernelGGL((gemm_smallm_kernel<E_STORE>), grid, dim3(256), 0, st, A.a, A.lda, Bm, ldb, M, Nout, Kd, Kd, E);
        PNPP_CHECK_LAUNCH("gemm(small M)");
        return PNPP_OK;
    }
    PNPP_REQUIRE(Kd % 4 == 0 && ldb % 4 == 0 && Nout % 4 == 0, PNPP_ERR_ARG,
                 "gemm: K=%d, N=%d and ldb=%d must be multiples of 4", Kd, Nout, ldb);
    PNPP_REQUIRE(((uintptr_t)Bm & 15) == 0, PNPP_ERR_ARG, "gemm: B operand must be 16-byte aligned");
    if (A.mode == A_PLAIN || A.mode == A_BNRELU || A.mode == A_DZ)
        PNPP_REQUIRE(A.lda % 4 == 0 && ((uintptr_t)A.a & 15) == 0, PNPP_ERR_ARG, "gemm: A operand pitch/alignment");
    // tile shape: tall tiles for the grouped layers (M = B*npoint*nsample), square-ish for small M
    if (M >= 128 * 128) {
        if (Nout % 128 == 0) return launch_gemm_cfg<128, 128, 4, 2>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
        if (Nout % 64 == 0) return launch_gemm_cfg<128, 64, 4, 1>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
        return launch_gemm_cfg<128, 32, 4, 1>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
    }
    if (M > 32) {
        if (Nout % 64 == 0) return launch_gemm_cfg<64, 64, 2, 2>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
        return launch_gemm_cfg<128, 32, 4, 1>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
    }
    if (Nout % 128 == 0) return launch_gemm_cfg<32, 128, 1, 4>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
    return launch_gemm_cfg<128, 32, 4, 1>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
}

// ---------------------------------------------------------------------------------------------
// dW = dZ^T * A2 : both operands are read straight from global memory in MFMA layout -- the lane
// index is the channel, which is the contiguous dimension of every row-major activation, so each
// half-wave load is one 128-byte segment.  Reduction runs over rows; each wave owns one
// (32*CT x 32*KT) output tile and one row range, partial tiles go to a slab (deterministic).
// ---------------------------------------------------------------------------------------------
template <int DZMODE, int A2MODE, int CT, int KT>
__global__ void __launch_bounds__(256)
dw_kernel(const AOperand dz, const AOperand a2, int M, int Nc, int Kp, int tilesC, int tilesK, int rows_per_split,
          int kp_pad, float *__restrict__ slab) {
    constexpr int U = 4;  // row pairs fetched per batch: U*(CT+KT) independent loads in flight per lane
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tiles = tilesC * tilesK;
    const int tile = gw % tiles, split = gw / tiles;
    const int r0 = min(M, split * rows_per_split);  // an empty range still writes its (zero) slab tile
    const int r1 = min(M, r0 + rows_per_split);
    const int c0 = (tile % tilesC) * 32 * CT, k0 = (tile / tilesC) * 32 * KT;

    ChanConst cc[CT], ck[KT];
#pragma unroll
    for (int i = 0; i < CT; ++i) cc[i] = load_chan_const<DZMODE>(dz, c0 + i * 32 + l31, Nc);
#pragma unroll
    for (int j = 0; j < KT; ++j) ck[j] = load_chan_const<A2MODE>(a2, k0 + j * 32 + l31, Kp);

    f32x16 acc[CT][KT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int row = r0; row < r1; row += 2 * U) {
        float2 fa[U][CT], fb[U][KT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = row + 2 * u + lh;
            const bool ok = m < r1;
#pragma unroll
            for (int i = 0; i < CT; ++i) fa[u][i] = fetch_a1<DZMODE>(dz, m, c0 + i * 32 + l31, Nc, ok);
#pragma unroll
            for (int j = 0; j < KT; ++j) fb[u][j] = fetch_a1<A2MODE>(a2, m, k0 + j * 32 + l31, Kp, ok);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = row + 2 * u + lh < r1;
            float a[CT], b[KT];
#pragma unroll
            for (int i = 0; i < CT; ++i) a[i] = xform_a1<DZMODE>(fa[u][i], cc[i], c0 + i * 32 + l31, Nc, ok);
#pragma unroll
            for (int j = 0; j < KT; ++j) b[j] = xform_a1<A2MODE>(fb[u][j], ck[j], k0 + j * 32 + l31, Kp, ok);
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float *o = slab + (size_t)split * Nc * kp_pad;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int k = k0 + j * 32 + l31;
                if (c < Nc && k < kp_pad) o[(size_t)c * kp_pad + k] = acc[i][j][r];
            }
}

void dw_plan(int M, int Nc, int Kp, int *nsplit, int *kp_pad) {
    const int tilesC = cdiv(Nc, 64), tilesK = cdiv(Kp, 64);
    const int tiles = tilesC * tilesK;
    // aim for ~2048 waves (2 per SIMD), at least 64 rows per wave, at most 1024 partial slabs
    int split = cdiv(2048, tiles);
    const int max_split = cdiv(M, 64);
    if (split > max_split) split = max_split;
    if (split > 1024) split = 1024;
    if (split < 1) split = 1;
    *nsplit = split;
    *kp_pad = tilesK * 64;
}

int launch_dw(const AOperand &dz, int Nc, const AOperand &a2, int Kp, int M, float *slab, int nsplit, int kp_pad,
              hipStream_t st) {
    PNPP_REQUIRE(M > 0 && Nc > 0 && Kp > 0 && nsplit > 0, PNPP_ERR_ARG, "dw: non-positive size");
    const int tilesC = cdiv(Nc, 64), tilesK = cdiv(Kp, 64);
    PNPP_REQUIRE(kp_pad == tilesK * 64, PNPP_ERR_ARG, "dw: kp_pad mismatch");
    int rps = cdiv(M, nsplit);
    rps = (rps + 1) & ~1;  // even: a row pair never straddles two splits
    const int waves = tilesC * tilesK * nsplit;
    const dim3 grid(cdiv(waves, 4)), block(256);
    ProfScope ps(st, "dw_kernel<A%d,A%d> M=%d N=%d K=%d split=%d", dz.mode, a2.mode, M, Nc, Kp, nsplit);
#define PNPP_DW(DM, AM)                                                                                              \
    hipLaunchKernelGGL((dw_kernel<DM, AM, 2, 2>), grid, block, 0, st, dz, a2, M, Nc, Kp, tilesC, tilesK, rps, kp_pad, slab); \
    break;
#define PNPP_DW_BY_A(DM)                         \
    switch (a2.mode) {                           \
        case A_PLAIN: PNPP_DW(DM, A_PLAIN)       \
        case A_BNRELU: PNPP_DW(DM, A_BNRELU)     \
        case A_GATHER: PNPP_DW(DM, A_GATHER)     \
        case A_CONCAT: PNPP_DW(DM, A_CONCAT)     \
        default: set_error("dw: bad A2 mode %d", a2.mode); return PNPP_ERR_ARG; \
    }                                            \
    break;
    switch (dz.mode) {
        case A_PLAIN: PNPP_DW_BY_A(A_PLAIN)
        case A_DZ: PNPP_DW_BY_A(A_DZ)
        default: set_error("dw: bad dZ mode %d", dz.mode); return PNPP_ERR_ARG;
    }
#undef PNPP_DW_BY_A
#undef PNPP_DW
    PNPP_CHECK_LAUNCH("dw");
    return PNPP_OK;
}

// out[c][perm(k)] = sum_s slab[s][c][k], fixed summation order: block = 64 outputs x 4 split lanes,
// every lane strides the splits by 4 with four independent partial sums, the 4 lanes are combined in order
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float *__restrict__ slab, int nsplit, int Nc, int kp_pad,
                                                          int Kvalid, int perm_D, float *__restrict__ out, int ldo) {
    __shared__ float red[4][64];
    const int total = Nc * Kvalid;
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    float acc = 0.f;
    int c = 0, k = 0;
    if (i < total) {
        c = i / Kvalid, k = i - c * Kvalid;
        const float *p = slab + (size_t)c * kp_pad + k;
        const size_t stride = (size_t)Nc * kp_pad;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = sl;
        for (; s + 12 < nsplit; s += 16) {
            a0 += p[(size_t)s * stride];
            a1 += p[(size_t)(s + 4) * stride];
            a2 += p[(size_t)(s + 8) * stride];
            a3 += p[(size_t)(s + 12) * stride];
        }
        for (; s < nsplit; s += 4) a0 += p[(size_t)s * stride];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
        int ko = k;
        if (perm_D >= 0) ko = k < perm_D ? k + 3 : k - perm_D;  // features-first -> xyz-first (state_dict order)
        out[(size_t)c * ldo + ko] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

int launch_slab_reduce(const float *slab, int nsplit, int Nc, int kp_pad, int Kvalid, int perm_D, float *out, int ldo,
                       hipStream_t st) {
    const int total = Nc * Kvalid;
    ProfScope ps(st, "slab_reduce_kernel N=%d K=%d split=%d", Nc, Kvalid, nsplit);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, slab, nsplit, Nc, kp_pad, Kvalid, perm_D, out,
                       ldo);
    PNPP_CHECK_LAUNCH("slab_reduce");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// weight preparation: W (Cout x Cin) -> W^T (Kd x Cout) [+ features-first row-major copy]
// ---------------------------------------------------------------------------------------------
struct PrepPack {
    PrepItem it[PNPP_MAX_LAYERS];
    int n;
};

__global__ void __launch_bounds__(256) prep_weights_kernel(const PrepPack P) {
    const PrepItem it = P.it[blockIdx.y];
    const int total = it.Kd * it.Cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int kp = i / it.Cout, n = i - kp * it.Cout;  // writes coalesced along n
        int ksrc = kp;                       // column of the state_dict weight this row comes from
        if (it.perm_D >= 0) ksrc = kp < it.perm_D ? kp + 3 : kp - it.perm_D;  // features first, then xyz
        const float v = kp < it.Cin ? it.w[(size_t)n * it.Cin + ksrc] : 0.f;   // rows Cin..Kd-1 are zero padding
        it.wt[(size_t)kp * it.Cout + n] = v;
        if (it.wperm) it.wperm[(size_t)n * it.Kd + kp] = v;
    }
}

int launch_prep_weights(const PrepItem *items, int n, hipStream_t st) {
    PNPP_REQUIRE(n > 0 && n <= PNPP_MAX_LAYERS, PNPP_ERR_ARG, "prep_weights: bad item count %d", n);
    PrepPack P;
    P.n = n;
    int maxtot = 0;
    for (int i = 0; i < n; ++i) {
        P.it[i] = items[i];
        maxtot = items[i].Kd * items[i].Cout > maxtot ? items[i].Kd * items[i].Cout : maxtot;
    }
    const int gx = cdiv(maxtot, 256) < 256 ? cdiv(maxtot, 256) : 256;
    ProfScope ps(st, "prep_weights_kernel n=%d", n);
    hipLaunchKernelGGL(prep_weights_kernel, dim3(gx, n), dim3(256), 0, st, P);
    PNPP_CHECK_LAUNCH("prep_weights");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics finalisation (float64 reduction of the slab partials, fixed order)
// block = 8 columns x 32 slab lanes (each lane owns every 32nd slab; fixed-order tree afterwards)
constexpr int FIN_COLS = 8;
__device__ __forceinline__ void slab_column_sums(const double *__restrict__ slab, int nslab, int C, int c, double &o1,
                                                 double &o2, double (*red)[2][FIN_COLS]) {
    const int g = threadIdx.x / FIN_COLS, cl = threadIdx.x % FIN_COLS;
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    if (c < C) {
        int s = g;
        for (; s + 32 < nslab; s += 64) {
            a0 += slab[((size_t)s * 2 + 0) * C + c];
            b0 += slab[((size_t)s * 2 + 1) * C + c];
            a1 += slab[((size_t)(s + 32) * 2 + 0) * C + c];
            b1 += slab[((size_t)(s + 32) * 2 + 1) * C + c];
        }
        for (; s < nslab; s += 32) {
            a0 += slab[((size_t)s * 2 + 0) * C + c];
            b0 += slab[((size_t)s * 2 + 1) * C + c];
        }
    }
    red[g][0][cl] = a0 + a1;
    red[g][1][cl] = b0 + b1;
    __syncthreads();
    o1 = 0.0, o2 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) o1 += red[i][0][cl], o2 += red[i][1][cl];
}

__global__ void __launch_bounds__(256)
bn_finalize_fwd_kernel(const double *__restrict__ slab, int nslab, int C, double count, const float *__restrict__ bias,
                       const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ rm,
                       float *__restrict__ rv, float momentum, float eps, int training, float *__restrict__ mean,
                       float *__restrict__ istd, float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ double red[32][2][FIN_COLS];
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    double mu, var;
    if (training) {
        double s1, s2;
        slab_column_sums(slab, nslab, C, c, s1, s2, red);
        mu = s1 / count;
        var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
    } else {
        if (c >= C) return;
        // eval: normalise z + bias with the running statistics  ->  "mean" of the bias-free z is rm - bias
        mu = (double)rm[c] - (bias ? (double)bias[c] : 0.0);
        var = (double)rv[c];
    }
    if (threadIdx.x >= FIN_COLS || c >= C) return;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    mean[c] = (float)mu;
    istd[c] = (float)is;
    scale[c] = (float)(g * is);
    shift[c] = (float)(bt - mu * g * is);
    if (training && rm) {
        const double bmean = mu + (bias ? (double)bias[c] : 0.0);  // the conv/linear bias was folded out of z
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * bmean);
        rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * unbiased);
    }
}

__global__ void __launch_bounds__(256)
bn_finalize_bwd_kernel(const double *__restrict__ slab, int nslab, int C, double count, int training,
                       const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ istd,
                       float *__restrict__ cst, float *__restrict__ dgamma, float *__restrict__ dbeta,
                       float *__restrict__ dbias) {
    __shared__ double red[32][2][FIN_COLS];
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    double s1, s2;
    slab_column_sums(slab, nslab, C, c, s1, s2, red);
    if (threadIdx.x >= FIN_COLS || c >= C) return;
    const float g = gamma ? gamma[c] : 1.f;
    cst[c] = g * istd[c];
    cst[C + c] = mean[c];
    cst[2 * C + c] = istd[c];
    cst[3 * C + c] = training ? (float)(s1 / count) : 0.f;
    cst[4 * C + c] = training ? (float)(s2 / count) : 0.f;
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    // a bias in front of a train-mode BatchNorm has exactly zero gradient; with running
    // statistics the layer is affine and d(bias) = sum_m dz = g * sum_m dy
    if (dbias) dbias[c] = training ? 0.f : (float)((double)cst[c] * s1);
}

int launch_bn_finalize_fwd(const double *slab, int nslab, int C, double count, const float *bias, const float *gamma,
                           const float *beta, float *rm, float *rv, float momentum, float eps, int training, float *mean,
                           float *istd, float *scale, float *shift, hipStream_t st) {
    ProfScope ps(st, "bn_finalize_fwd_kernel C=%d", C);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, FIN_COLS)), dim3(256), 0, st, slab, nslab, C, count, bias, gamma, beta,
                       rm, rv, momentum, eps, training, mean, istd, scale, shift);
    PNPP_CHECK_LAUNCH("bn_finalize_fwd");
    return PNPP_OK;
}

int launch_bn_finalize_bwd(const double *slab, int nslab, int C, double count, int training, const float *gamma,
                           const float *mean, const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias,
                           hipStream_t st) {
    ProfScope ps(st, "bn_finalize_bwd_kernel C=%d", C);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, FIN_COLS)), dim3(256), 0, st, slab, nslab, C, count, training, gamma,
                       mean, istd, cst, dgamma, dbeta, dbias);
    PNPP_CHECK_LAUNCH("bn_finalize_bwd");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// max over the nsample axis with BatchNorm apply + ReLU folded in (pointnet_pp_8dir.py:41-42)
// first maximum wins ties (what torch.max does on the CPU)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pool_fwd_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int G, int K, int C,
                                                       float *__restrict__ out, int32_t *__restrict__ arg) {
    const size_t total = (size_t)G * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t g = i / C;
        const int c = (int)(i - g * C);
        const float sc = scale[c], sh = shift[c];
        const float *p = z + g * K * C + c;
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float v = fmaxf(fmaf(p[(size_t)k * C], sc, sh), 0.f);
            if (v > best) best = v, bi = k;
        }
        out[i] = best;
        arg[i] = bi;
    }
}

int launch_pool_fwd(const float *z, const float *scale, const float *shift, int G, int K, int C, float *out, int32_t *arg,
                    hipStream_t st) {
    const size_t total = (size_t)G * C;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ProfScope ps(st, "pool_fwd_kernel G=%d K=%d C=%d", G, K, C);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid), dim3(256), 0, st, z, scale, shift, G, K, C, out, arg);
    PNPP_CHECK_LAUNCH("pool_fwd");
    return PNPP_OK;
}

// backward of max + ReLU: dense dy (zero except at the arg-max row when the pooled value is > 0)
// plus the two BatchNorm-backward column sums.  block = 64 channels x 4 group lanes.
__global__ void __launch_bounds__(256)
pool_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ arg, const float *__restrict__ z,
                const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
                const float *__restrict__ istd, int G, int K, int C, float *__restrict__ dy, double *__restrict__ slab) {
    __shared__ double red[4][2][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float mu = mean[c], is = istd[c], sc = scale[c], sh = shift[c];
        for (int g = blockIdx.y * 4 + gl; g < G; g += gridDim.y * 4) {
            const size_t gi = (size_t)g * C + c;
            const int a = arg[gi];
            const float za = z[((size_t)g * K + a) * C + c];
            const float d = fmaf(za, sc, sh) > 0.f ? dout[gi] : 0.f;  // ReLU'(pooled value), same expression as forward
            float *p = dy + (size_t)g * K * C + c;
            for (int k = 0; k < K; ++k) p[(size_t)k * C] = (k == a) ? d : 0.f;
            s1 += (double)d;
            s2 += (double)d * (double)((za - mu) * is);
        }
    }
    red[gl][0][cl] = s1;
    red[gl][1][cl] = s2;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a += red[i][0][cl], b += red[i][1][cl];
        slab[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        slab[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

int launch_pool_bwd(const float *dout, const int32_t *arg, const float *z, const float *scale, const float *shift,
                    const float *mean, const float *istd, int G, int K, int C, float *dy, double *slab, int *nslab,
                    hipStream_t st) {
    int gy = cdiv(G, 4);
    if (gy > kMaxStatBlocks) gy = kMaxStatBlocks;
    *nslab = gy;
    ProfScope ps(st, "pool_bwd_kernel G=%d K=%d C=%d", G, K, C);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(cdiv(C, 64), gy), dim3(256), 0, st, dout, arg, z, scale, shift, mean, istd, G, K, C,
                       dy, slab);
    PNPP_CHECK_LAUNCH("pool_bwd");
    return PNPP_OK;
}

__global__ void __launch_bounds__(256) fill_zero_kernel(float4 *__restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_fill_zero(void *p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return PNPP_OK;
    hipError_t e = hipMemsetAsync(p, 0, bytes, st);
    if (e != hipSuccess) {
        set_error("memset failed: %s", hipGetErrorString(e));
        return PNPP_ERR_LAUNCH;
    }
    return PNPP_OK;
}

}  // namespace pnpp
