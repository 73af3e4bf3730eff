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
// A-operand loaders: four consecutive k of one row
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void load_a4(const AOperand &A, int row, int k, int M, int Kd, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (row >= M || k >= Kd) return;
    if constexpr (MODE == A_PLAIN) {
        const float4 t = *reinterpret_cast<const float4 *>(A.a + (size_t)row * A.lda + k);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    } else if constexpr (MODE == A_BNRELU) {
        const float4 t = *reinterpret_cast<const float4 *>(A.a + (size_t)row * A.lda + k);
        const float4 s = *reinterpret_cast<const float4 *>(A.scale + k);
        const float4 h = *reinterpret_cast<const float4 *>(A.shift + k);
        v[0] = fmaxf(fmaf(t.x, s.x, h.x), 0.f);
        v[1] = fmaxf(fmaf(t.y, s.y, h.y), 0.f);
        v[2] = fmaxf(fmaf(t.z, s.z, h.z), 0.f);
        v[3] = fmaxf(fmaf(t.w, s.w, h.w), 0.f);
    } else if constexpr (MODE == A_GATHER || MODE == A_CONCAT) {
        size_t prow;  // source point row in (B*N)
        if constexpr (MODE == A_GATHER) {
            const int grp = row / A.K;  // centre row (b*S + s)
            const int b = grp / A.S;
            prow = (size_t)b * A.N + A.idx[row];
            if ((A.D & 3) == 0 && k + 3 < A.D) {
                const float4 t = *reinterpret_cast<const float4 *>(A.a + prow * A.D + k);
                v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i;
                if (kk < A.D) {
                    v[i] = A.a[prow * A.D + kk];
                } else if (kk < A.D + 3) {
                    const int c = kk - A.D;
                    v[i] = __fsub_rn(A.xyz[prow * 3 + c], A.new_xyz[(size_t)grp * 3 + c]);  // pointnet_pp_8dir.py:32
                }
            }
        } else {
            prow = (size_t)row;
            if ((A.D & 3) == 0 && k + 3 < A.D) {
                const float4 t = *reinterpret_cast<const float4 *>(A.a + prow * A.D + k);
                v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i;
                if (kk < A.D) {
                    v[i] = A.a[prow * A.D + kk];
                } else if (kk < A.D + 3) {
                    v[i] = A.xyz[prow * 3 + (kk - A.D)];  // absolute coordinates, pointnet_pp_8dir.py:24-26
                }
            }
        }
    } else {  // A_DZ
        const float4 dy = *reinterpret_cast<const float4 *>(A.a + (size_t)row * A.lda + k);
        const float4 z = *reinterpret_cast<const float4 *>(A.z + (size_t)row * A.lda + k);
        const float *c = A.cst + k;
        const float4 g = *reinterpret_cast<const float4 *>(c);
        const float4 mu = *reinterpret_cast<const float4 *>(c + A.C);
        const float4 is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        const float4 c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C);
        const float4 c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
        v[0] = g.x * (dy.x - c1.x - (z.x - mu.x) * is.x * c2.x);
        v[1] = g.y * (dy.y - c1.y - (z.y - mu.y) * is.y * c2.y);
        v[2] = g.z * (dy.z - c1.z - (z.z - mu.z) * is.z * c2.z);
        v[3] = g.w * (dy.w - c1.w - (z.w - mu.w) * is.w * c2.w);
    }
}

// scalar flavour used by the dW kernel (one element per lane, channel index on the lane)
template <int MODE>
__device__ __forceinline__ float load_a1(const AOperand &A, int row, int k, int Kvalid) {
    if (k >= Kvalid) return 0.f;
    if constexpr (MODE == A_PLAIN) {
        return A.a[(size_t)row * A.lda + k];
    } else if constexpr (MODE == A_BNRELU) {
        return fmaxf(fmaf(A.a[(size_t)row * A.lda + k], A.scale[k], A.shift[k]), 0.f);
    } else if constexpr (MODE == A_GATHER) {
        const int grp = row / A.K;
        const int b = grp / A.S;
        const size_t prow = (size_t)b * A.N + A.idx[row];
        if (k < A.D) return A.a[prow * A.D + k];
        const int c = k - A.D;
        return __fsub_rn(A.xyz[prow * 3 + c], A.new_xyz[(size_t)grp * 3 + c]);
    } else if constexpr (MODE == A_CONCAT) {
        if (k < A.D) return A.a[(size_t)row * A.D + k];
        return A.xyz[(size_t)row * 3 + (k - A.D)];
    } else {
        const float dy = A.a[(size_t)row * A.lda + k], z = A.z[(size_t)row * A.lda + k];
        const float *c = A.cst + k;
        return c[0] * (dy - c[3 * A.C] - (z - c[A.C]) * c[2 * A.C] * c[4 * A.C]);
    }
}

// ---------------------------------------------------------------------------------------------
// fused GEMM: persistent row-tile workers (grid.x) x column tiles (grid.y)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int AMODE, int EMODE>
__global__ void __launch_bounds__(256, 2)
gemm_kernel(const AOperand A, const float *__restrict__ Bm, int ldb, int M, int Nout, int Kd, const Epilogue E) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "tile configuration");
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

        for (int k0 = 0; k0 < Kd; k0 += KC) {
            __syncthreads();
            // stage A' chunk (BM x KC), transform applied on the way in
#pragma unroll
            for (int f = tid; f < BM * (KC / 4); f += 256) {
                const int r = f / (KC / 4), q = f % (KC / 4);
                float v[4];
                load_a4<AMODE>(A, m0 + r, k0 + 4 * q, M, Kd, v);
                float *d = As + r * APITCH + 4 * q;
                d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
            }
            // stage B chunk (KC x BN), 16-byte loads and stores
#pragma unroll
            for (int f = tid; f < KC * (BN / 4); f += 256) {
                const int kk = f / (BN / 4), jq = f % (BN / 4);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kk < Kd && n0 + 4 * jq < Nout)
                    t = *reinterpret_cast<const float4 *>(Bm + (size_t)(k0 + kk) * ldb + n0 + 4 * jq);
                *reinterpret_cast<float4 *>(Bs + kk * BN + 4 * jq) = t;
            }
            __syncthreads();
            const int ksteps = min(KC, Kd - k0) >> 1;
            const float *ap = As + (wm * TM + l31) * APITCH + lh;
            const float *bp = Bs + lh * BN + wn * TN + l31;
            if (ksteps == KC / 2) {
#pragma unroll
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
        for (int f = tid; f < 2 * BN; f += 256) {
            const int which = f / BN, cl = f % BN;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + cl];
            if (n0 + cl < Nout) E.slab[((size_t)blockIdx.x * 2 + which) * Nout + n0 + cl] = t;
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_cfg(const AOperand &A, const float *Bm, int ldb, int M, int Nout, int Kd, const Epilogue &E,
                           int *nslab, hipStream_t st) {
    const int tiles = cdiv(M, BM);
    const int gx = tiles < kMaxStatBlocks ? tiles : kMaxStatBlocks;
    const dim3 grid(gx, cdiv(Nout, BN)), block(256);
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
    PNPP_REQUIRE(Kd % 4 == 0 && ldb % 4 == 0 && Nout % 4 == 0, PNPP_ERR_ARG,
                 "gemm: K=%d, N=%d and ldb=%d must be multiples of 4", Kd, Nout, ldb);
    PNPP_REQUIRE(((uintptr_t)Bm & 15) == 0, PNPP_ERR_ARG, "gemm: B operand must be 16-byte aligned");
    if (A.mode == A_PLAIN || A.mode == A_BNRELU || A.mode == A_DZ)
        PNPP_REQUIRE(A.lda % 4 == 0 && ((uintptr_t)A.a & 15) == 0, PNPP_ERR_ARG, "gemm: A operand pitch/alignment");
    // tile shape: tall tiles for the grouped layers (M = B*npoint*nsample), square-ish for small M
    if (M >= 128 * 128) {
        if (Nout % 128 == 0) return launch_gemm_cfg<128, 128, 4, 1>(A, Bm, ldb, M, Nout, Kd, E, nslab, st);
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
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tiles = tilesC * tilesK;
    const int tile = gw % tiles, split = gw / tiles;
    const int r0 = split * rows_per_split;
    if (r0 >= M) return;
    const int r1 = min(M, r0 + rows_per_split);
    const int c0 = (tile % tilesC) * 32 * CT, k0 = (tile / tilesC) * 32 * KT;

    f32x16 acc[CT][KT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll 4
    for (int row = r0; row < r1; row += 2) {
        const int m = row + lh;
        const bool ok = m < r1;
        float a[CT], b[KT];
#pragma unroll
        for (int i = 0; i < CT; ++i) a[i] = ok ? load_a1<DZMODE>(dz, m, c0 + i * 32 + l31, Nc) : 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j) b[j] = ok ? load_a1<A2MODE>(a2, m, k0 + j * 32 + l31, Kp) : 0.f;
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
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
    // aim for ~2048 waves (2 per SIMD), at least 64 rows per wave
    int split = cdiv(2048, tiles);
    const int max_split = cdiv(M, 64);
    if (split > max_split) split = max_split;
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

// out[c][perm(k)] = sum_s slab[s][c][k], fixed summation order
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float *__restrict__ slab, int nsplit, int Nc, int kp_pad,
                                                          int Kvalid, int perm_D, float *__restrict__ out, int ldo) {
    const int total = Nc * Kvalid;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = i / Kvalid, k = i - c * Kvalid;
        const float *p = slab + (size_t)c * kp_pad + k;
        const size_t stride = (size_t)Nc * kp_pad;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
            a0 += p[(size_t)s * stride];
            a1 += p[(size_t)(s + 1) * stride];
            a2 += p[(size_t)(s + 2) * stride];
            a3 += p[(size_t)(s + 3) * stride];
        }
        for (; s < nsplit; ++s) a0 += p[(size_t)s * stride];
        int ko = k;
        if (perm_D >= 0) ko = k < perm_D ? k + 3 : k - perm_D;  // features-first -> xyz-first (state_dict order)
        out[(size_t)c * ldo + ko] = (a0 + a1) + (a2 + a3);
    }
}

int launch_slab_reduce(const float *slab, int nsplit, int Nc, int kp_pad, int Kvalid, int perm_D, float *out, int ldo,
                       hipStream_t st) {
    const int total = Nc * Kvalid;
    const int grid = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
    ProfScope ps(st, "slab_reduce_kernel N=%d K=%d split=%d", Nc, Kvalid, nsplit);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, st, slab, nsplit, Nc, kp_pad, Kvalid, perm_D, out, ldo);
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
// block = 32 columns x 8 slab strides
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_column_sums(const double *__restrict__ slab, int nslab, int C, int c, double &o1,
                                                 double &o2, double (*red)[2][32]) {
    const int g = threadIdx.x >> 5, cl = threadIdx.x & 31;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int s = g; s < nslab; s += 8) {
            a += slab[((size_t)s * 2 + 0) * C + c];
            b += slab[((size_t)s * 2 + 1) * C + c];
        }
    red[g][0][cl] = a;
    red[g][1][cl] = b;
    __syncthreads();
    o1 = 0.0, o2 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o1 += red[i][0][cl], o2 += red[i][1][cl];
}

__global__ void __launch_bounds__(256)
bn_finalize_fwd_kernel(const double *__restrict__ slab, int nslab, int C, double count, const float *__restrict__ bias,
                       const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ rm,
                       float *__restrict__ rv, float momentum, float eps, int training, float *__restrict__ mean,
                       float *__restrict__ istd, float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ double red[8][2][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
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
    if (threadIdx.x >= 32 || c >= C) return;
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
    __shared__ double red[8][2][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    double s1, s2;
    slab_column_sums(slab, nslab, C, c, s1, s2, red);
    if (threadIdx.x >= 32 || c >= C) return;
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
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, slab, nslab, C, count, bias, gamma, beta,
                       rm, rv, momentum, eps, training, mean, istd, scale, shift);
    PNPP_CHECK_LAUNCH("bn_finalize_fwd");
    return PNPP_OK;
}

int launch_bn_finalize_bwd(const double *slab, int nslab, int C, double count, int training, const float *gamma,
                           const float *mean, const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias,
                           hipStream_t st) {
    ProfScope ps(st, "bn_finalize_bwd_kernel C=%d", C);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, slab, nslab, C, count, training, gamma,
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
