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
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

// A/B switch read once per process (PNPP_NO_MID=1: the group_all level stays on the 32 x 32 split-K kernels)
static bool mid_tiles_on() {
    static int cached = -1;
    if (cached < 0) {
        const char *v = getenv("PNPP_NO_MID");
        cached = (v && atoi(v) != 0) ? 0 : 1;
    }
    return cached != 0;
}

constexpr int KC = 32;       // reduction-dim chunk staged in LDS per step
constexpr int APITCH = KC + 1;  // odd pitch: the 32 rows a half-wave reads land on 32 different banks

// ---------------------------------------------------------------------------------------------
// A-operand loaders: four consecutive k of one row, split into a raw fetch (global loads only, so
// the next chunk's loads can be in flight while the current chunk is in the MFMA loop) and a
// transform applied when the chunk is written to LDS.
// ---------------------------------------------------------------------------------------------
struct RawA {
    float4 p, q;  // p: primary values; q: z (A_DZ*) or the centre coordinates to subtract (A_GATHER xyz part)
    int4 ia;      // A_DZ_POOL: arg-max neighbour of the row's group, per channel
};

// Loads are UNCONDITIONAL on clamped (always valid) addresses and masked afterwards: a load inside a
// per-lane branch makes hipcc branch around it and drain vmcnt(0) per element, which serialises the
// whole prefetch (cdna_hip_programming.md, "three .s-level traps", item c).
template <int MODE>
__device__ __forceinline__ RawA fetch_a4(const AOperand &A, int row, int k, int M, int Kd) {
    RawA r;
    r.p = make_float4(0.f, 0.f, 0.f, 0.f);
    r.q = make_float4(0.f, 0.f, 0.f, 0.f);
    r.ia = make_int4(0, 0, 0, 0);
    const int rc = min(row, M - 1);
    if constexpr (MODE == A_PLAIN || MODE == A_BNRELU) {
        const int kc = min(k, Kd - 4);
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kc);
    } else if constexpr (MODE == A_DZ) {
        const int kc = min(k, Kd - 4);
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kc);
        r.q = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kc);
    } else if constexpr (MODE == A_DZ_POOL) {
        const int kc = min(k, Kd - 4);
        const size_t g = (size_t)(rc / A.K);
        r.p = *reinterpret_cast<const float4 *>(A.a + g * A.lda + kc);
        r.ia = *reinterpret_cast<const int4 *>(A.arg + g * A.lda + kc);
        r.q = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kc);
    } else {  // A_GATHER / A_CONCAT: features first, then xyz (relative to the centre when gathering)
        size_t prow = (size_t)rc, grp = 0;
        if constexpr (MODE == A_GATHER) {
            grp = (size_t)(rc / A.K);  // centre row (b*S + s)
            prow = (size_t)(grp / A.S) * A.N + A.idx[rc];
        }
        if ((A.D & 3) == 0 && A.D >= 4) {
            // whole float4 groups are either features (k < D) or the [x y z 0] tail (k == D)
            const float4 f = *reinterpret_cast<const float4 *>(A.a + prow * A.D + min(k, A.D - 4));
            const float *xp = A.xyz + prow * 3;
            const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if constexpr (MODE == A_GATHER) {
                const float *cp = A.new_xyz + grp * 3;
                c0 = cp[0], c1 = cp[1], c2 = cp[2];
            }
            const bool feat = k < A.D;
            r.p = feat ? f : make_float4(x0, x1, x2, 0.f);
            r.q = feat ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(c0, c1, c2, 0.f);
        } else {
            float pv[4], qv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i;
                const bool feat = kk < A.D;
                const int xc = min(max(kk - A.D, 0), 2);
                const float *src = feat ? A.a + prow * A.D + kk : A.xyz + prow * 3 + xc;
                const float v = *src;
                float cv = 0.f;
                if constexpr (MODE == A_GATHER) cv = A.new_xyz[grp * 3 + xc];
                const bool valid = kk < A.D + 3;
                pv[i] = valid ? v : 0.f;
                qv[i] = (valid && !feat) ? cv : 0.f;
            }
            r.p = make_float4(pv[0], pv[1], pv[2], pv[3]);
            r.q = make_float4(qv[0], qv[1], qv[2], qv[3]);
        }
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
    } else {  // A_DZ / A_DZ_POOL
        float4 dy = r.p;
        if constexpr (MODE == A_DZ_POOL) {
            const int kk = row % A.K;  // neighbour slot of this row inside its group
            dy.x = kk == r.ia.x ? dy.x : 0.f, dy.y = kk == r.ia.y ? dy.y : 0.f;
            dy.z = kk == r.ia.z ? dy.z : 0.f, dy.w = kk == r.ia.w ? dy.w : 0.f;
        }
        const float *c = A.cst + k;
        const float4 g = *reinterpret_cast<const float4 *>(c);
        const float4 mu = *reinterpret_cast<const float4 *>(c + A.C);
        const float4 is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        const float4 c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C);
        const float4 c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
        v[0] = g.x * (dy.x - c1.x - (r.q.x - mu.x) * is.x * c2.x);
        v[1] = g.y * (dy.y - c1.y - (r.q.y - mu.y) * is.y * c2.y);
        v[2] = g.z * (dy.z - c1.z - (r.q.z - mu.z) * is.z * c2.z);
        v[3] = g.w * (dy.w - c1.w - (r.q.w - mu.w) * is.w * c2.w);
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
    if constexpr (MODE == A_DZ || MODE == A_DZ_POOL) {
        const float *p = A.cst + k;
        c.g = p[0], c.mu = p[A.C], c.is = p[2 * A.C], c.c1 = p[3 * A.C], c.c2 = p[4 * A.C];
    } else if constexpr (MODE == A_BNRELU) {
        c.sc = A.scale[k], c.sh = A.shift[k];
    }
    return c;
}

// raw loads of element (row, k): up to two values (second one: z for A_DZ*, centre coordinate for A_GATHER).
// Unconditional loads on clamped indices; xform_a1 masks what was out of range.
template <int MODE>
__device__ __forceinline__ float2 fetch_a1(const AOperand &A, int row, int k, int Kvalid, int M) {
    float2 r = make_float2(0.f, 0.f);
    const int rc = min(row, M - 1), kc = min(k, Kvalid - 1);
    if constexpr (MODE == A_PLAIN || MODE == A_BNRELU) {
        r.x = A.a[(size_t)rc * A.lda + kc];
    } else if constexpr (MODE == A_DZ) {
        r.x = A.a[(size_t)rc * A.lda + kc];
        r.y = A.z[(size_t)rc * A.lda + kc];
    } else if constexpr (MODE == A_DZ_POOL) {
        const int g = rc / A.K;
        const size_t gi = (size_t)g * A.lda + kc;
        const float d = A.a[gi];
        r.x = (rc - g * A.K == A.arg[gi]) ? d : 0.f;
        r.y = A.z[(size_t)rc * A.lda + kc];
    } else if constexpr (MODE == A_GATHER) {
        const int grp = rc / A.K;
        const size_t prow = (size_t)(grp / A.S) * A.N + A.idx[rc];
        const bool feat = kc < A.D;
        const int xc = min(max(kc - A.D, 0), 2);
        const float *src = feat ? A.a + prow * A.D + kc : A.xyz + prow * 3 + xc;
        r.x = *src;
        const float cv = A.new_xyz[(size_t)grp * 3 + xc];
        r.y = feat ? 0.f : cv;
    } else {  // A_CONCAT
        const bool feat = kc < A.D;
        const int xc = min(max(kc - A.D, 0), 2);
        const float *src = feat ? A.a + (size_t)rc * A.D + kc : A.xyz + (size_t)rc * 3 + xc;
        r.x = *src;
    }
    return r;
}

template <int MODE>
__device__ __forceinline__ float xform_a1(const float2 r, const ChanConst &c, int k, int Kvalid, bool ok) {
    // out-of-range lanes were loaded from clamped (valid, finite) addresses and are zeroed by a multiplication: a
    // select here lets hipcc sink the loads into a per-lane branch and wait for each of them separately
    const float m = (ok && k < Kvalid) ? 1.f : 0.f;
    if constexpr (MODE == A_PLAIN) {
        return r.x * m;
    } else if constexpr (MODE == A_BNRELU) {
        return fmaxf(fmaf(r.x, c.sc, c.sh), 0.f) * m;
    } else if constexpr (MODE == A_DZ || MODE == A_DZ_POOL) {
        return c.g * (r.x - c.c1 - (r.y - c.mu) * c.is * c.c2) * m;
    } else {
        return __fsub_rn(r.x, r.y) * m;
    }
}

// ---------------------------------------------------------------------------------------------
// fused GEMM: persistent row-tile workers (grid.x) x column tiles (grid.y)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int AMODE, int EMODE>
__global__ void __launch_bounds__(WM * WN * 64, 2)
gemm_kernel(const AOperand A, const BOperand B, int M, int Nout, int Kd, const Epilogue E) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int NTHR = WM * WN * 64;
    static_assert(TM % 32 == 0 && TN % 32 == 0 && (BM * (KC / 4)) % NTHR == 0 && (KC * (BN / 4)) % NTHR == 0, "tile configuration");
    constexpr int BP = BN;
    __shared__ __attribute__((aligned(16))) float lds[BM * APITCH + KC * BP];
    float *As = lds, *Bs = lds + BM * APITCH;
    const float *__restrict__ Bm = B.b;
    const int ldb = B.ldb;
    const bool bvec = (ldb & 3) == 0 && ((uintptr_t)Bm & 15) == 0 && B.perm_D < 0;

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
                float t[4];
                if (!B.trans) {  // [k][n]: four consecutive n of one reduction row
                    const int kk = k0 + f / (BN / 4), n = n0 + 4 * (f % (BN / 4));
                    const float *src = Bm + (size_t)min(kk, B.rows - 1) * ldb;
                    if (bvec) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + min(n, Nout - 4));
                        t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = src[min(n + e, Nout - 1)];
                    }
                    const bool okk = kk < B.rows;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= (okk && n + e < Nout) ? 1.f : 0.f;
                } else {  // [n][k]: four consecutive reduction indices of one output column; n runs fastest over the
                          // lanes so that the transposing LDS stores below are conflict-free
                    const int n = n0 + f % BN, kq = k0 + 4 * (f / BN);
                    const float *src = Bm + (size_t)min(n, Nout - 1) * ldb;
                    if (bvec) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + min(kq, B.rows - 4));
                        t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kp = min(kq + e, B.rows - 1);
                            int col = kp;
                            if (B.perm_D >= 0) col = kp < B.perm_D ? kp + 3 : kp - B.perm_D;
                            t[e] = src[col];
                        }
                    }
                    const bool okn = n < Nout;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= (okn && kq + e < B.rows) ? 1.f : 0.f;
                }
                rb[i] = make_float4(t[0], t[1], t[2], t[3]);
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
                if (!B.trans) {
                    *reinterpret_cast<float4 *>(Bs + (f / (BN / 4)) * BP + 4 * (f % (BN / 4))) = rb[i];
                } else {
                    float *d = Bs + 4 * (f / BN) * BP + f % BN;
                    d[0] = rb[i].x, d[BP] = rb[i].y, d[2 * BP] = rb[i].z, d[3 * BP] = rb[i].w;
                }
            }
            __syncthreads();
            if (k0 + KC < Kd) fetch(k0 + KC);
            const int ksteps = min(KC, Kd - k0) >> 1;
            const float *ap = As + (wm * TM + l31) * APITCH + lh;
            const float *bp = Bs + lh * BP + wn * TN + l31;
            if (ksteps == KC / 2) {
#pragma unroll 4
                for (int s = 0; s < KC / 2; ++s) {
                    float a[MT], b[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = ap[i * 32 * APITCH + 2 * s];
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j] = bp[2 * s * BP + j * 32];
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
                    for (int j = 0; j < NT; ++j) b[j] = bp[2 * s * BP + j * 32];
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
                    const bool ok = row < M && col < Nout;
                    float v = ok ? acc[i][j][r] : 0.f;  // padding rows are exact zeros: they add nothing to the sums
                    if constexpr (EMODE == E_STORE_STATS) {
                        s1[j] += (double)v;
                        s2[j] += (double)v * (double)v;
                    } else if constexpr (EMODE == E_MASK_STATS) {
                        const float zp = E.zp[(size_t)min(row, M - 1) * E.ldc + min(col, Nout - 1)];
                        v = (fmaf(zp, sc, sh) > 0.f) ? v : 0.f;
                        s1[j] += (double)v;
                        s2[j] += (double)v * (double)((zp - mu) * is);
                    }
                    if (ok) E.c[(size_t)row * E.ldc + col] = v;
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
// weights-stationary GEMM for the grouped layers (M = B*npoint*nsample rows, K and N <= 260):
//   * the whole weight panel W[K x BN] is staged ONCE per workgroup and stays in LDS;
//   * a workgroup is a persistent worker over row tiles; per tile the whole A'[BM x K] panel is staged
//     in one shot (transform applied on the way in), so there are two barriers per tile instead of
//     two per 32-deep K chunk, and the next tile's HBM stream is already in flight (registers) while
//     the current tile runs its K/2 MFMA steps -- a full tile of compute hides the load latency;
//   * only the HBM streams are prefetched; L2-resident side tables (pooled gradient, arg-max,
//     per-channel constants) are read when the tile is written to LDS.
// Layout of the reduction dimension for grouped operands: [features (D) | x y z 0].
// ---------------------------------------------------------------------------------------------
struct RawS {
    float4 p, q;  // q: z stream of A_DZ
};
struct RawTail {
    float x0, x1, x2, c0, c1, c2;
};

template <int MODE>
__device__ __forceinline__ RawS ws_fetch(const AOperand &A, int row, int k, int M, int Kmain) {
    RawS r;
    r.p = make_float4(0.f, 0.f, 0.f, 0.f);
    r.q = r.p;
    const int rc = min(row, M - 1), kc = min(k, Kmain - 4);
    if constexpr (MODE == A_PLAIN || MODE == A_BNRELU) {
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kc);
    } else if constexpr (MODE == A_DZ) {
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kc);
        r.q = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kc);
    } else if constexpr (MODE == A_DZ_POOL) {
        r.p = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kc);  // the only HBM stream
    } else if constexpr (MODE == A_GATHER) {
        const size_t prow = (size_t)((rc / A.K) / A.S) * A.N + A.idx[rc];
        r.p = *reinterpret_cast<const float4 *>(A.a + prow * A.D + kc);
    } else {  // A_CONCAT
        r.p = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.D + kc);
    }
    return r;
}

template <int MODE>
__device__ __forceinline__ void ws_xform(const AOperand &A, const RawS &r, int row, int k, int M, int Kmain, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (row >= M || k >= Kmain) return;
    if constexpr (MODE == A_PLAIN || MODE == A_GATHER || MODE == A_CONCAT) {
        v[0] = r.p.x, v[1] = r.p.y, v[2] = r.p.z, v[3] = r.p.w;
    } else if constexpr (MODE == A_BNRELU) {
        const float4 s = *reinterpret_cast<const float4 *>(A.scale + k);
        const float4 h = *reinterpret_cast<const float4 *>(A.shift + k);
        v[0] = fmaxf(fmaf(r.p.x, s.x, h.x), 0.f);
        v[1] = fmaxf(fmaf(r.p.y, s.y, h.y), 0.f);
        v[2] = fmaxf(fmaf(r.p.z, s.z, h.z), 0.f);
        v[3] = fmaxf(fmaf(r.p.w, s.w, h.w), 0.f);
    } else {  // A_DZ / A_DZ_POOL
        float4 dy, z;
        if constexpr (MODE == A_DZ) {
            dy = r.p, z = r.q;
        } else {
            z = r.p;
            const int g = row / A.K, kk = row - g * A.K;
            const float4 dm = *reinterpret_cast<const float4 *>(A.a + (size_t)g * A.lda + k);
            const int4 ia = *reinterpret_cast<const int4 *>(A.arg + (size_t)g * A.lda + k);
            dy.x = kk == ia.x ? dm.x : 0.f, dy.y = kk == ia.y ? dm.y : 0.f;
            dy.z = kk == ia.z ? dm.z : 0.f, dy.w = kk == ia.w ? dm.w : 0.f;
        }
        const float *c = A.cst + k;
        const float4 g4 = *reinterpret_cast<const float4 *>(c);
        const float4 mu = *reinterpret_cast<const float4 *>(c + A.C);
        const float4 is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        const float4 c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C);
        const float4 c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
        v[0] = g4.x * (dy.x - c1.x - (z.x - mu.x) * is.x * c2.x);
        v[1] = g4.y * (dy.y - c1.y - (z.y - mu.y) * is.y * c2.y);
        v[2] = g4.z * (dy.z - c1.z - (z.z - mu.z) * is.z * c2.z);
        v[3] = g4.w * (dy.w - c1.w - (z.w - mu.w) * is.w * c2.w);
    }
}

template <int MODE>
__device__ __forceinline__ RawTail ws_fetch_tail(const AOperand &A, int row, int M) {
    RawTail t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == A_GATHER || MODE == A_CONCAT) {
        const int rc = min(row, M - 1);
        size_t prow = (size_t)rc;
        if constexpr (MODE == A_GATHER) {
            const size_t grp = (size_t)(rc / A.K);
            prow = (size_t)(grp / A.S) * A.N + A.idx[rc];
            const float *cp = A.new_xyz + grp * 3;
            t.c0 = cp[0], t.c1 = cp[1], t.c2 = cp[2];
        }
        const float *xp = A.xyz + prow * 3;
        t.x0 = xp[0], t.x1 = xp[1], t.x2 = xp[2];
    }
    return t;
}

// KD: reduction length (compile time; for grouped operands KD = D + 4 with the [x y z 0] tail last).
// Thread -> (column group kq = tid % G4, rows tid / G4 + i * RPP): every thread keeps ONE column group
// for the whole kernel, so its per-channel constants live in registers, and with K_nbr == 32 the
// pooled-gradient / arg-max entries of a tile's few neighbour groups are fetched once per tile.
#ifdef PNPP_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ int g_stamp_kd;
#define PNPP_STAMP(i)                                                 \
    if (st_on) {                                                      \
        __builtin_amdgcn_s_waitcnt(0);                                \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime(); \
        if (lane == 0) g_stamps[i] += st_t - st_last;                 \
        st_last = st_t;                                               \
    }
#else
#define PNPP_STAMP(i)
#endif
// A/B switches of the tile loop (tools/ab_trace.sh builds one library per setting):
#ifndef PNPP_WS_DENSE_MINK      // smallest K whose full tiles take the clamp-free staging pass and uniform-pointer streams
#define PNPP_WS_DENSE_MINK 256
#endif
#ifndef PNPP_WS_INTER_MINK      // smallest K whose next-tile loads are issued between the MFMAs of the K loop
#define PNPP_WS_INTER_MINK 256
#endif
#ifndef PNPP_WS_DWTABLE_MINK    // smallest K whose fused dW loop reads its operands through a per-lane address table
#define PNPP_WS_DWTABLE_MINK 256
#endif
#ifndef PNPP_WS_POOL_PREFETCH   // pooled-gradient / arg-max entries of the NEXT tile requested before this tile's stores
#define PNPP_WS_POOL_PREFETCH 1
#endif
// timing experiments only (results are WRONG with either on; never in a shipped build): what is left of a launch without its
// matrix instructions, or without its output stores
#ifndef PNPP_WS_TUNED128        // pooled K = 128 backward launch on the clamp-free staging pass + dW address table
#define PNPP_WS_TUNED128 1
#endif
#ifndef PNPP_WS_DEFER_STORES     // forward kernels: a tile's stores are issued behind the NEXT tile's staging pass
#define PNPP_WS_DEFER_STORES 1
#endif
#ifndef PNPP_WS_EXP_NO_MFMA
#define PNPP_WS_EXP_NO_MFMA 0
#endif
#ifndef PNPP_WS_EXP_NO_STORE
#define PNPP_WS_EXP_NO_STORE 0
#endif
#if PNPP_WS_EXP_NO_MFMA
#define PNPP_WS_MFMA(a, b, c) ((c)[0] += (a) + (b), (c))
#else
#define PNPP_WS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif
template <int KD, int BM, int BN, int WM, int WN, int AMODE, int EMODE, bool FDW>
__global__ void __launch_bounds__(256, (KD >= 256 ? 1 : 2))  // the K=256 panels leave room for one workgroup per CU anyway
gemm_ws_kernel(const AOperand A, const BOperand B, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr bool HAS_TAIL = (AMODE == A_GATHER || AMODE == A_CONCAT);
    constexpr int KMAIN = HAS_TAIL ? KD - 4 : KD;  // columns served by the float4 stream
    constexpr int G4 = KMAIN / 4;                  // column groups per row
    constexpr int RPP = G4 > 0 ? 256 / (G4 > 0 ? G4 : 1) : 1;  // rows staged per pass
    constexpr int NG = G4 > 0 ? BM / RPP : 0;      // passes per tile
    // A tile addressing: element (r, k) lives at r*KP + (k ^ f(r)), f(r) = (r & 15) << 2, when KD is a multiple of 64:
    // an XOR swizzle of whole 16-byte groups, no padding.  The dA operand is read lane-per-row as ONE ds_read_b128 per
    // four MFMA steps (the 16-lane groups of that instruction hold rows that are distinct mod 16, so they land on 16
    // different groups of a 256-byte bank row); the dW operand is read lane-per-column with ds_read_b32 (fixed r: the
    // XOR permutes an aligned block of 32 columns, 32 different banks); a staged float4 is one ds_write_b128 of the
    // registers as loaded.  The weight tile uses the same image, [n][k ^ f(n)].  Otherwise (KD = D + 4) element
    // (r, k) is at r*(KD+1) + k and the weights are [k][n].
    constexpr bool SWZ = (KD % 64 == 0);
    constexpr int KP = SWZ ? KD : KD + 1;
    constexpr int GPT = (BM + 31) / 32;            // neighbour groups per tile when nsample == 32
    constexpr int DW_TILES = FDW ? (KD / 32) * (BN / 32) : 0, DT = FDW ? (DW_TILES + 3) / 4 : 1;
    static_assert(!FDW || (EMODE == E_MASK_STATS && SWZ && DW_TILES % 4 == 0), "fused dW needs the ReLU-mask epilogue");
    static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "tile configuration");
    static_assert(G4 == 0 || (256 % G4 == 0 && BM % RPP == 0 && (32 % RPP == 0 || RPP % 32 == 0)), "staging map");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds;            // [KD][BN], or [BN][KD] swizzled (SWZ)
    float *As = lds + KD * BN;  // [BM][KP]
    float *Ap = As + BM * KP;   // FDW: [BM][BN] = relu(bn(zp)) tile, the dW GEMM's second operand
    auto a_swz = [](int r) { return (r & 15) << 2; };
    auto a_idx = [&](int r, int k) { return r * KP + (SWZ ? (k ^ a_swz(r)) : k); };

    // the wave index is uniform: telling the compiler so moves the tile-row / tile-column arithmetic of every address to the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
#ifdef PNPP_STAMPS
    // selector: KD for the fused pooled backward kernels, 1000 + KD for the forward kernels (BN+ReLU operand, statistics)
    const bool st_on = blockIdx.x == 8 && wave == 0 &&
                       ((FDW && AMODE == A_DZ_POOL && g_stamp_kd == KD) ||
                        (!FDW && AMODE == A_BNRELU && EMODE == E_STORE_STATS && g_stamp_kd == 1000 + KD));
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    // XCD-aware tile map (speed only, never correctness): blocks b and b + 8 share an XCD and with it an L2, so the ncol
    // column blocks of one worker -- which stream the SAME operand rows -- are placed 8 apart: the rows come from HBM once
    // and from that L2 for the other column blocks, instead of once per XCD
    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;
    const int kq = G4 > 0 ? 4 * (tid % (G4 > 0 ? G4 : 1)) : 0;  // this thread's first column
    const int r_base = G4 > 0 ? tid / (G4 > 0 ? G4 : 1) : 0;

    // ---- per-channel constants of this thread's column group: registers for the whole kernel.  Their loads go out FIRST (the table
    // was written by the previous launch: a cold ~1.5 us round trip), then the weight panel's, then the first tile's: one wait
    // covers the three instead of three round trips in a row ----
    float4 c_g = make_float4(0.f, 0.f, 0.f, 0.f), c_mu = c_g, c_is = c_g, c_c1 = c_g, c_c2 = c_g, c_sc = c_g, c_sh = c_g;
    if constexpr (G4 > 0 && (AMODE == A_DZ || AMODE == A_DZ_POOL)) {
        const float *c = A.cst + kq;
        c_g = *reinterpret_cast<const float4 *>(c);
        c_mu = *reinterpret_cast<const float4 *>(c + A.C);
        c_is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        c_c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C);
        c_c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
    } else if constexpr (G4 > 0 && AMODE == A_BNRELU) {
        c_sc = *reinterpret_cast<const float4 *>(A.scale + kq);
        c_sh = *reinterpret_cast<const float4 *>(A.shift + kq);
    }

    const int tiles = (M + BM - 1) / BM;
    const bool pool_fast = (AMODE == A_DZ_POOL) && A.K == 32;  // tiles start on neighbour-group boundaries (BM % 32 == 0)
    float4 rp[NG > 0 ? NG : 1], rq[(AMODE == A_DZ && NG > 0) ? NG : 1];
    RawTail rt;
    // M a multiple of BM (every shape of the training step): no row of a tile needs a clamp or a bounds test, and the operand
    // streams are (uniform tile pointer, advanced by scalar arithmetic) + (one 32-bit lane offset computed once) -- the 64-bit
    // multiply-add, clamp and EXEC branch per 16-byte group were a third of the staging pass, and VALU time is MFMA time
    // (measured: -4.6 % on the K = 256 kernels, which run one wave per SIMD; nothing or a small loss on the K <= 128 kernels with
    // two waves per SIMD, which keep the general path)
    // (round 3 A/B, two traces per variant on one box: for K <= 128 the clamp-free pass alone is +1.4 / +2.2 us on the pooled K = 128
    // and the K = 64 backward launch; together with the dW address table it is -2.4 us on the pooled K = 128 launch and 0 / +0.3
    // on the others -- so that one instantiation takes both)
    constexpr bool TUNED_128 = PNPP_WS_TUNED128 && KD == 128 && AMODE == A_DZ_POOL && FDW;
    constexpr bool DENSE_A = (AMODE == A_PLAIN || AMODE == A_BNRELU || AMODE == A_DZ || AMODE == A_DZ_POOL) &&
                             (KD >= PNPP_WS_DENSE_MINK || TUNED_128);
    const bool full_rows = DENSE_A && (M % BM) == 0 && (AMODE != A_DZ_POOL || A.K == 32);
    const unsigned offA = (unsigned)r_base * (unsigned)A.lda + (unsigned)kq;
    auto fetch = [&](int m0) {
        if constexpr (DENSE_A) {
            if (full_rows) {
                const float *pa = (AMODE == A_DZ_POOL ? A.z : A.a) + (size_t)m0 * A.lda;
                const float *pz = A.z + (size_t)m0 * A.lda;
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    rp[i] = *reinterpret_cast<const float4 *>(pa + (size_t)(i * RPP) * A.lda + offA);
                    if constexpr (AMODE == A_DZ) rq[i] = *reinterpret_cast<const float4 *>(pz + (size_t)(i * RPP) * A.lda + offA);
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int rc = min(m0 + r_base + i * RPP, M - 1);
            if constexpr (AMODE == A_PLAIN || AMODE == A_BNRELU) {
                rp[i] = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kq);
            } else if constexpr (AMODE == A_DZ) {
                rp[i] = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.lda + kq);
                rq[i] = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kq);
            } else if constexpr (AMODE == A_DZ_POOL) {
                rp[i] = *reinterpret_cast<const float4 *>(A.z + (size_t)rc * A.lda + kq);  // the only HBM stream
            } else if constexpr (AMODE == A_GATHER) {
                const size_t prow = (size_t)((rc / A.K) / A.S) * A.N + A.idx[rc];
                rp[i] = *reinterpret_cast<const float4 *>(A.a + prow * A.D + kq);
            } else {
                rp[i] = *reinterpret_cast<const float4 *>(A.a + (size_t)rc * A.D + kq);
            }
        }
        if constexpr (HAS_TAIL) rt = ws_fetch_tail<AMODE>(A, m0 + min(tid, BM - 1), M);
    };
    // vmcnt counts loads and stores in issue order on gfx9: a wait for a load issued AFTER a tile's 16 output stores is a wait
    // for those stores' acknowledgements too.  Everything the next staging pass reads is therefore requested before them.
    float4 gdm[GPT];
    int4 garg[GPT];
    auto fetch_pool = [&](int m0) {
        if constexpr (AMODE == A_DZ_POOL) {
            if (full_rows) {
#pragma unroll
                for (int g = 0; g < GPT; ++g) {
                    const size_t gi = (size_t)(m0 / 32 + g) * A.lda + kq;
                    gdm[g] = *reinterpret_cast<const float4 *>(A.a + gi);
                    garg[g] = *reinterpret_cast<const int4 *>(A.arg + gi);
                }
            } else if (pool_fast) {
#pragma unroll
                for (int g = 0; g < GPT; ++g) {
                    const size_t gi = (size_t)min(m0 / 32 + g, (M - 1) / 32) * A.lda + kq;
                    gdm[g] = *reinterpret_cast<const float4 *>(A.a + gi);
                    garg[g] = *reinterpret_cast<const int4 *>(A.arg + gi);
                }
            }
        }
    };
    int tile = worker;
    bool fetched = false;
    // ---- weights: staged once, [k][n] ----
    {
        const float *__restrict__ Bm = B.b;
        const int ldb = B.ldb;
        const bool bvec = (ldb & 3) == 0 && ((uintptr_t)Bm & 15) == 0 && B.perm_D < 0;
        // two passes: every 16-byte group of the panel this thread owns is REQUESTED first (all loads in flight together: one
        // L2 / HBM round trip for the whole panel instead of one per group of a rolled loop), then masked and written to LDS
        auto wload = [&](int f, float (&t)[4]) {
            if (!B.trans && SWZ) {
                // row-major panel -> [n][k ^ f(n)] image: lane = column (four dword loads of consecutive rows, each 256 B per
                // wave), then ONE conflict-free ds_write_b128 per group.  (The first version read 16 bytes along n and wrote
                // four transposed ds_write_b32 that met 8-way bank conflicts: 5.8 us of a 54 us launch went into this panel.)
                const int nl = f % BN, k4 = 4 * (f / BN), n = min(n0 + nl, Nout - 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = Bm[(size_t)min(k4 + e, B.rows - 1) * ldb + n];
            } else if (!B.trans) {
                const int kk = f / (BN / 4), n = n0 + 4 * (f % (BN / 4));
                const float *src = Bm + (size_t)min(kk, B.rows - 1) * ldb;
                if (bvec) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + min(n, Nout - 4));
                    t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = src[min(n + e, Nout - 1)];
                }
            } else {
                // swizzled image: consecutive lanes take consecutive 16-byte groups of ONE weight row (coalesced 1 KB per wave, and
                // the ds_write_b128 of a row land in distinct slots); the [k][n] image keeps lane = column
                const int nl = SWZ ? f / (KD / 4) : f % BN, k4 = SWZ ? 4 * (f % (KD / 4)) : 4 * (f / BN), n = n0 + nl;
                const float *src = Bm + (size_t)min(n, Nout - 1) * ldb;
                if (bvec) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + min(k4, B.rows - 4));
                    t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kp = min(k4 + e, B.rows - 1);
                        int col = kp;
                        if (B.perm_D >= 0) col = kp < B.perm_D ? kp + 3 : kp - B.perm_D;
                        t[e] = src[col];
                    }
                }
            }
        };
        auto wstore = [&](int f, float (&t)[4]) {
            if (!B.trans && SWZ) {
                const int nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] *= (n0 + nl < Nout && k4 + e < B.rows) ? 1.f : 0.f;
                *reinterpret_cast<float4 *>(Ws + nl * KD + (k4 ^ a_swz(nl))) = make_float4(t[0], t[1], t[2], t[3]);
            } else if (!B.trans) {
                const int kk = f / (BN / 4), n = n0 + 4 * (f % (BN / 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] *= (kk < B.rows && n + e < Nout) ? 1.f : 0.f;
                if constexpr (SWZ) {  // transposed image: four scalar stores (once per workgroup)
                    const int nl = 4 * (f % (BN / 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ws[(nl + e) * KD + (kk ^ a_swz(nl + e))] = t[e];
                } else {
                    *reinterpret_cast<float4 *>(Ws + kk * BN + 4 * (f % (BN / 4))) = make_float4(t[0], t[1], t[2], t[3]);
                }
            } else {
                // swizzled image: consecutive lanes take consecutive 16-byte groups of ONE weight row (coalesced 1 KB per wave, and
                // the ds_write_b128 of a row land in distinct slots); the [k][n] image keeps lane = column
                const int nl = SWZ ? f / (KD / 4) : f % BN, k4 = SWZ ? 4 * (f % (KD / 4)) : 4 * (f / BN), n = n0 + nl;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] *= (n < Nout && k4 + e < B.rows) ? 1.f : 0.f;
                if constexpr (SWZ) {
                    *reinterpret_cast<float4 *>(Ws + nl * KD + (k4 ^ a_swz(nl))) = make_float4(t[0], t[1], t[2], t[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ws[(k4 + e) * BN + nl] = t[e];
                }
            }
        };
        // (row-major panels of the backward kernels; for the [n][k] weights of the forward kernels the same two passes were worth
        // 0.2 us per launch and cost the backward instantiations as much -- A/B on one box, tools/ab_trace.sh -- so they keep the loop)
        constexpr int NWF = (KD * (BN / 4)) / 256;
        bool staged = false;
        if constexpr (SWZ && (KD * (BN / 4)) % 256 == 0 && NWF >= 4 && NWF <= 16) {
            if (!B.trans) {
                staged = true;
                float tw[NWF][4];
#pragma unroll
                for (int j = 0; j < NWF; ++j) wload(tid + 256 * j, tw[j]);
                if (tile < tiles) fetch(tile * BM);   // behind the panel in the queue: its HBM round trip overlaps the LDS writes
                fetched = true;
#pragma unroll
                for (int j = 0; j < NWF; ++j) wstore(tid + 256 * j, tw[j]);
            }
        }
        if (!staged) {
            for (int f = tid; f < KD * (BN / 4); f += 256) {
                float t[4];
                if (!B.trans) {
                    const int kk = f / (BN / 4), n = n0 + 4 * (f % (BN / 4));
                    const float *src = Bm + (size_t)min(kk, B.rows - 1) * ldb;
                    if (bvec) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + min(n, Nout - 4));
                        t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = src[min(n + e, Nout - 1)];
                    }
    #pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= (kk < B.rows && n + e < Nout) ? 1.f : 0.f;
                    if constexpr (SWZ) {  // transposed image: four scalar stores (once per workgroup)
                        const int nl = 4 * (f % (BN / 4));
    #pragma unroll
                        for (int e = 0; e < 4; ++e) Ws[(nl + e) * KD + (kk ^ a_swz(nl + e))] = t[e];
                    } else {
                        *reinterpret_cast<float4 *>(Ws + kk * BN + 4 * (f % (BN / 4))) = make_float4(t[0], t[1], t[2], t[3]);
                    }
                } else {
                    // swizzled image: consecutive lanes take consecutive 16-byte groups of ONE weight row (coalesced 1 KB per wave, and
                // the ds_write_b128 of a row land in distinct slots); the [k][n] image keeps lane = column
                const int nl = SWZ ? f / (KD / 4) : f % BN, k4 = SWZ ? 4 * (f % (KD / 4)) : 4 * (f / BN), n = n0 + nl;
                    const float *src = Bm + (size_t)min(n, Nout - 1) * ldb;
                    if (bvec) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + min(k4, B.rows - 4));
                        t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kp = min(k4 + e, B.rows - 1);
                            int col = kp;
                            if (B.perm_D >= 0) col = kp < B.perm_D ? kp + 3 : kp - B.perm_D;
                            t[e] = src[col];
                        }
                    }
    #pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= (n < Nout && k4 + e < B.rows) ? 1.f : 0.f;
                    if constexpr (SWZ) {
                        *reinterpret_cast<float4 *>(Ws + nl * KD + (k4 ^ a_swz(nl))) = make_float4(t[0], t[1], t[2], t[3]);
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) Ws[(k4 + e) * BN + nl] = t[e];
                    }
                }
            }
        }
    }

    PNPP_STAMP(11)  // prologue: weight panel in LDS
    if (!fetched && tile < tiles) fetch(tile * BM);
    if constexpr (PNPP_WS_POOL_PREFETCH) {
        if (tile < tiles) fetch_pool(tile * BM);
    }
    if constexpr (G4 > 0 && (AMODE == A_DZ || AMODE == A_DZ_POOL)) {
        // dz = g (dy - c1 - (z - mu) istd c2) as two FMAs per element: g dy + (a z + b), a = -g istd c2, b = -g c1 - a mu
        // (c_is keeps a, c_c1 keeps b from here on; six dependent VALU per element otherwise, and VALU time is MFMA time)
        c_is = make_float4(-c_g.x * c_is.x * c_c2.x, -c_g.y * c_is.y * c_c2.y, -c_g.z * c_is.z * c_c2.z, -c_g.w * c_is.w * c_c2.w);
        c_c1 = make_float4(-c_g.x * c_c1.x - c_is.x * c_mu.x, -c_g.y * c_c1.y - c_is.y * c_mu.y, -c_g.z * c_c1.z - c_is.z * c_mu.z,
                           -c_g.w * c_c1.w - c_is.w * c_mu.w);
    }

    PNPP_STAMP(12)  // prologue: per-channel constants
    double s1[NT], s2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) s1[i] = s2[i] = 0.0;
    float pool_sg[NT];   // sign of gamma at this lane's columns (pooling in the epilogue: max z or min z)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        pool_sg[j] = 1.f;
        if constexpr (EMODE == E_STORE_STATS) {
            if (E.pool_ext && E.pool_gamma) pool_sg[j] = E.pool_gamma[min(n0 + wn * TN + j * 32 + l31, Nout - 1)] >= 0.f ? 1.f : -1.f;
        }
    }

    f32x16 dwacc[DT];  // FDW: this wave's (32 x 32) tiles of dW, accumulated over every row tile of the worker
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.f;

    // Forward kernels (one 32 x 32 tile per wave): the tile's output stores are DEFERRED to the next iteration, behind the staging
    // pass.  vmcnt counts loads and stores in issue order, and the compiler cannot prove how many stores separate the next tile's
    // operand loads from the wait in front of the staging pass (the first iteration has none), so it waits with vmcnt(0): every
    // tile then also waits for its predecessor's 16 stores to be ACKNOWLEDGED -- 3 to 6 us per forward launch (a build with the
    // stores compiled out: 30.0 -> 24.0, 19.4 -> 16.0, 23.7 -> 20.8 us).  Held back until the loads have been consumed, the
    // stores have a whole tile to drain before anything waits again.
    constexpr bool DEFER = PNPP_WS_DEFER_STORES && EMODE != E_MASK_STATS && !FDW && MT == 1 && NT == 1;
    f32x16 held;
    float held_ext = 0.f;
    int held_arg = 0, held_m0 = -1;
    auto flush_held = [&]() {
        if constexpr (DEFER) {
            if (held_m0 >= 0) {
                if constexpr (EMODE == E_STORE_STATS) {
                    if (E.pool_ext && lh == 0) {
                        const size_t gi = (size_t)((held_m0 + wm * TM) >> 5) * E.ldc + (n0 + wn * TN + l31);
                        E.pool_ext[gi] = held_ext;
                        E.pool_arg[gi] = held_arg;
                    }
                }
                float *tb = E.c + (size_t)(held_m0 + wm * TM) * E.ldc + (n0 + wn * TN);
                const unsigned lo = (unsigned)(4 * lh) * (unsigned)E.ldc + (unsigned)l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc + lo] = held[r];
                held_m0 = -1;
            }
        }
    };

    // see the K loop: with `inter` the operand loads of the NEXT tile are issued between this tile's MFMAs
    const bool inter = DENSE_A && KD >= PNPP_WS_INTER_MINK && SWZ && EMODE == E_MASK_STATS && !(FDW && NT > 1) && full_rows && n0 + BN <= Nout;
    PNPP_STAMP(8)   // prologue, rest: first tile's loads complete
    for (; tile < tiles; tile += nworkers) {
        const int m0 = tile * BM;
        PNPP_STAMP(0)
        // pooled gradient / arg-max of the tile's neighbour groups at this thread's columns (L2-resident tables): requested one
        // tile ahead (fetch_pool below), so that no load the staging pass waits for is YOUNGER than the previous tile's stores
        if constexpr (!PNPP_WS_POOL_PREFETCH) fetch_pool(m0);
        __syncthreads();  // previous tile's operand reads are done (and, first time, the weights are staged)
        PNPP_STAMP(1)
        // two copies of the staging pass, the compile-time flag FULL picking which one runs (full_rows is uniform): the copy for
        // M % BM == 0 has no bounds test, no zero fill and no EXEC branch per 16-byte group
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
        const bool FULL = pass == 0;
        if (FULL != full_rows) continue;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int r = r_base + i * RPP, row = m0 + r;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (FULL || row < M) {
                if constexpr (AMODE == A_PLAIN || AMODE == A_GATHER || AMODE == A_CONCAT) {
                    v[0] = rp[i].x, v[1] = rp[i].y, v[2] = rp[i].z, v[3] = rp[i].w;
                } else if constexpr (AMODE == A_BNRELU) {
                    v[0] = fmaxf(fmaf(rp[i].x, c_sc.x, c_sh.x), 0.f);
                    v[1] = fmaxf(fmaf(rp[i].y, c_sc.y, c_sh.y), 0.f);
                    v[2] = fmaxf(fmaf(rp[i].z, c_sc.z, c_sh.z), 0.f);
                    v[3] = fmaxf(fmaf(rp[i].w, c_sc.w, c_sh.w), 0.f);
                } else {
                    float4 dy, z;
                    if constexpr (AMODE == A_DZ) {
                        dy = rp[i], z = rq[i];
                    } else {
                        z = rp[i];
                        float4 dm;
                        int4 ia;
                        int kk;
                        if (FULL || pool_fast) {   // (FULL implies pool_fast: no merge with the general path below)
                            // neighbour group of this row inside the tile: a constant per unrolled pass when RPP | 32
                            const int g = (RPP >= 32) ? r / 32 : (i * RPP) / 32;
                            dm = gdm[0], ia = garg[0];
#pragma unroll
                            for (int gg = 1; gg < GPT; ++gg)
                                if (g == gg) dm = gdm[gg], ia = garg[gg];
                            kk = r - g * 32;
                        } else {
                            const int g = row / A.K;
                            kk = row - g * A.K;
                            dm = *reinterpret_cast<const float4 *>(A.a + (size_t)g * A.lda + kq);
                            ia = *reinterpret_cast<const int4 *>(A.arg + (size_t)g * A.lda + kq);
                        }
                        dy.x = kk == ia.x ? dm.x : 0.f, dy.y = kk == ia.y ? dm.y : 0.f;
                        dy.z = kk == ia.z ? dm.z : 0.f, dy.w = kk == ia.w ? dm.w : 0.f;
                    }
                    v[0] = fmaf(c_g.x, dy.x, fmaf(c_is.x, z.x, c_c1.x));
                    v[1] = fmaf(c_g.y, dy.y, fmaf(c_is.y, z.y, c_c1.y));
                    v[2] = fmaf(c_g.z, dy.z, fmaf(c_is.z, z.z, c_c1.z));
                    v[3] = fmaf(c_g.w, dy.w, fmaf(c_is.w, z.w, c_c1.w));
                }
            }
            if constexpr (SWZ) {
                *reinterpret_cast<float4 *>(As + r * KP + (kq ^ a_swz(r))) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[a_idx(r, kq + e)] = v[e];
            }
        }
        }
        if constexpr (HAS_TAIL) {
            if (tid < BM) {
                const bool ok = m0 + tid < M;
                // [x-cx, y-cy, z-cz, 0]: float32 subtraction, pointnet_pp_8dir.py:32
                As[a_idx(tid, KMAIN + 0)] = ok ? __fsub_rn(rt.x0, rt.c0) : 0.f;
                As[a_idx(tid, KMAIN + 1)] = ok ? __fsub_rn(rt.x1, rt.c1) : 0.f;
                As[a_idx(tid, KMAIN + 2)] = ok ? __fsub_rn(rt.x2, rt.c2) : 0.f;
                As[a_idx(tid, KMAIN + 3)] = 0.f;
            }
        }
        flush_held();   // the previous tile's output: this tile's operand loads have just been consumed
        PNPP_STAMP(2)
        __syncthreads();
        PNPP_STAMP(3)
        // K >= 256 (one wave per SIMD), dense operand, full tiles: the next tile's operand loads and this tile's epilogue operand are
        // issued BETWEEN the MFMAs of the unrolled K loop below -- a memory instruction issues while the matrix pipe works
        // on the previous MFMA, whereas 32 loads issued in front of the loop are ~1k cycles in which the pipe idles
        const bool have_next = tile + nworkers < tiles;
        if (!inter && have_next) fetch((tile + nworkers) * BM);  // next tile's HBM stream flies during the MFMA loop
        if constexpr (PNPP_WS_POOL_PREFETCH) {
            if (have_next) fetch_pool((tile + nworkers) * BM);   // (this tile's entries were consumed by the staging pass above)
        }

        // the ReLU-mask operand of the epilogue is fetched now and lands while the MFMA loop runs
        float zp[MT][NT][16];
        if constexpr (EMODE == E_MASK_STATS) {
          if (inter) {
            // (issued inside the K loop)
          } else if (full_rows && n0 + BN <= Nout) {
            const float *pzp = E.zp + (size_t)m0 * E.ldc;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const unsigned oz = (unsigned)(wm * TM + i * 32 + 4 * lh) * (unsigned)E.ldc + (unsigned)(n0 + wn * TN + j * 32 + l31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) zp[i][j][r] = pzp[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc + oz];
                }
          } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int cc = min(n0 + wn * TN + j * 32 + l31, Nout - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        zp[i][j][r] = E.zp[(size_t)min(row, M - 1) * E.ldc + cc];
                    }
                }
          }
        }

        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float *bp = Ws + lh * BN + wn * TN + l31;
        if constexpr (SWZ) {
            // K loop for the swizzled tiles.  Lane (l31, lh) fetches the four reduction indices k = 64 c + 8 t + 4 lh + {0..3}
            // of its row (A) and of its column (W) with one ds_read_b128 each and feeds them to four MFMA steps -- both
            // operands of a step carry the same k for the same lh, which is all the instruction asks for.  The swizzled
            // group is 64 c + ((8 t) ^ (4 lh ^ f)), so for a fixed t the NC = KD / 64 reads of a lane differ by an immediate
            // offset only: one xor + one add of address arithmetic per operand and t (an fp32 MFMA and VALU work of the
            // same SIMD do not overlap, so every VALU in this loop is MFMA time lost).  t is the rolled, software-pipelined
            // loop (reads of t + 1 are issued before the MFMAs of t); c is unrolled.
            constexpr int NC = KD / 64;
            const float *arow[MT], *brow[NT];
            int ga[MT], gb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * TM + i * 32 + l31;
                arow[i] = As + r * KP;
                ga[i] = (4 * lh) ^ a_swz(r);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = wn * TN + j * 32 + l31;
                brow[j] = Ws + n * KD;
                gb[j] = (4 * lh) ^ a_swz(n);
            }
            float4 ra[2][NC][MT], rb[2][NC][NT];
            auto ld = [&](int buf, int t) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float *pa = arow[i] + ((8 * t) ^ ga[i]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) ra[buf][c][i] = *reinterpret_cast<const float4 *>(pa + 64 * c);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float *pb = brow[j] + ((8 * t) ^ gb[j]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) rb[buf][c][j] = *reinterpret_cast<const float4 *>(pb + 64 * c);
                }
            };
            auto mm = [&](int buf) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = PNPP_WS_MFMA(ra[buf][c][i].x, rb[buf][c][j].x, acc[i][j]);
                            acc[i][j] = PNPP_WS_MFMA(ra[buf][c][i].y, rb[buf][c][j].y, acc[i][j]);
                            acc[i][j] = PNPP_WS_MFMA(ra[buf][c][i].z, rb[buf][c][j].z, acc[i][j]);
                            acc[i][j] = PNPP_WS_MFMA(ra[buf][c][i].w, rb[buf][c][j].w, acc[i][j]);
                        }
            };
            if constexpr (FDW && NT > 1) {  // no second operand buffer: the fused dW instantiations with two column tiles per
                                            // wave are at the 256-register budget of two waves per SIMD
#pragma unroll 1
                for (int t = 0; t < 8; ++t) {
                    ld(0, t);
                    mm(0);
                }
            } else {
                bool looped = false;
                if constexpr (DENSE_A && EMODE == E_MASK_STATS) {
                    if (inter) {
                        looped = true;
                        const size_t mn = (size_t)(tile + nworkers) * BM;
                        const float *pa = (AMODE == A_DZ_POOL ? A.z : A.a) + mn * A.lda;
                        const float *pz = A.z + mn * A.lda;
                        const float *pzp = E.zp + (size_t)m0 * E.ldc;
                        auto issue = [&](int u) {  // an eighth of the two load streams
                            constexpr int GP = (NG + 7) / 8, ZP = MT * NT * 2;
                            if (have_next) {
#pragma unroll
                                for (int g = 0; g < GP; ++g) {
                                    const int i = u * GP + g;
                                    if (i < NG) {
                                        rp[i] = *reinterpret_cast<const float4 *>(pa + (size_t)(i * RPP) * A.lda + offA);
                                        if constexpr (AMODE == A_DZ)
                                            rq[i] = *reinterpret_cast<const float4 *>(pz + (size_t)(i * RPP) * A.lda + offA);
                                    }
                                }
                            }
#pragma unroll
                            for (int q = u * ZP; q < (u + 1) * ZP; ++q) {
                                const int r = q & 15, j = (q >> 4) % NT, i = (q >> 4) / NT;
                                const unsigned oz = (unsigned)(wm * TM + i * 32 + 4 * lh) * (unsigned)E.ldc + (unsigned)(n0 + wn * TN + j * 32 + l31);
                                zp[i][j][r] = pzp[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc + oz];
                            }
                        };
                        ld(0, 0);
#pragma unroll
                        for (int t = 0; t < 8; t += 2) {
                            ld(1, t + 1);
                            issue(t);
                            mm(0);
                            if (t + 2 < 8) ld(0, t + 2);
                            issue(t + 1);
                            mm(1);
                        }
                    }
                }
                if (!looped) {
                    ld(0, 0);
#pragma unroll 1
                    for (int t = 0; t < 8; t += 2) {
                        ld(1, t + 1);
                        mm(0);
                        if (t + 2 < 8) ld(0, t + 2);
                        mm(1);
                    }
                }
            }
        } else {
            {
                // K loop, software-pipelined by hand: the LDS operand reads of block b+1 (SB k-steps) are issued before
                // the MFMAs of block b, so an MFMA never waits for a read issued right in front of it
                constexpr int SB = (KD / 2) % 4 == 0 ? 4 : 2, NBLK = (KD / 2) / SB;
                float ra[2][SB][MT], rb[2][SB][NT];
                auto ld = [&](int buf, int s0) {
    #pragma unroll
                    for (int u = 0; u < SB; ++u) {
    #pragma unroll
                        for (int i = 0; i < MT; ++i) ra[buf][u][i] = As[a_idx(wm * TM + i * 32 + l31, 2 * (s0 + u) + lh)];
    #pragma unroll
                        for (int j = 0; j < NT; ++j) rb[buf][u][j] = bp[2 * (s0 + u) * BN + j * 32];
                    }
                };
                auto mm = [&](int buf) {
    #pragma unroll
                    for (int u = 0; u < SB; ++u)
    #pragma unroll
                        for (int i = 0; i < MT; ++i)
    #pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = PNPP_WS_MFMA(ra[buf][u][i], rb[buf][u][j], acc[i][j]);
                };
                ld(0, 0);
    #pragma unroll 1
                for (int blk = 0; blk + 1 < NBLK; blk += 2) {  // rolled: a fully unrolled loop lets the scheduler hoist reads until it spills
                    ld(1, (blk + 1) * SB);
                    mm(0);
                    if (blk + 2 < NBLK) ld(0, (blk + 2) * SB);
                    mm(1);
                }
                if constexpr (NBLK % 2 == 1) mm(0);
            }
        }

        PNPP_STAMP(4)
        // epilogue: each accumulator register is one row; a half-wave writes 32 consecutive floats (128 B)
        bool done = false;
        if constexpr (EMODE != E_MASK_STATS) {
            // interior tiles of the forward kernels: no bounds tests, and every store is (uniform row pointer) + (one 32-bit
            // lane offset) -- scalar address arithmetic instead of a 64-bit multiply-add, a compare and an EXEC branch per row
            if (m0 + BM <= M && n0 + BN <= Nout) {
                done = true;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float *tb = E.c + (size_t)(m0 + wm * TM + i * 32) * E.ldc + (n0 + wn * TN + j * 32);
                        const unsigned lo = (unsigned)(4 * lh) * (unsigned)E.ldc + (unsigned)l31;
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][j][r];
                            float *tr = tb + (size_t)((r & 3) + 8 * (r >> 2)) * E.ldc;
                            if constexpr (!DEFER) {
                                if (!PNPP_WS_EXP_NO_STORE) tr[lo] = v;
                            }
                            t1 += v;
                            t2 = fmaf(v, v, t2);
                        }
                        if constexpr (DEFER) held = acc[i][j], held_m0 = m0;
                        if constexpr (EMODE == E_STORE_STATS) s1[j] += (double)t1, s2[j] += (double)t2;
                        if constexpr (EMODE == E_STORE_STATS) {
                            if (E.pool_ext) {   // (uniform) this 32 x 32 tile is one neighbourhood: its extreme row per column
                                const float sg = pool_sg[j];
                                float mx = sg * acc[i][j][0];
#pragma unroll
                                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sg * acc[i][j][r]);
                                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));        // the other 16 rows of the column sit in lane ^ 32
                                int a = 64;
#pragma unroll
                                for (int r = 15; r >= 0; --r) a = (sg * acc[i][j][r] == mx) ? (r & 3) + 8 * (r >> 2) + 4 * lh : a;
                                a = min(a, __shfl_xor(a, 32, 64));             // first row attaining it
                                if constexpr (DEFER) {   // stored with the tile, behind the next staging pass
                                    held_ext = sg * mx, held_arg = a;
                                } else if (lh == 0) {
                                    const size_t gi = (size_t)((m0 + wm * TM + i * 32) >> 5) * E.ldc + (n0 + wn * TN + j * 32 + l31);
                                    E.pool_ext[gi] = sg * mx;
                                    E.pool_arg[gi] = a;
                                }
                            }
                        }
                    }
            }
        }
        if constexpr (EMODE == E_MASK_STATS && NT == 1) {  // (two column tiles per wave: hipcc hoists the straight-line copy into spills)
            if (m0 + BM <= M && n0 + BN <= Nout) {  // the same for the backward kernels: ReLU mask, BN-backward sums, a_{l-1} tile
                done = true;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int col = n0 + wn * TN + j * 32 + l31;
                        const float sc = E.scale[col], sh = E.shift[col], mu = E.mu[col], is = E.istd[col];
                        float *tb = E.c + (size_t)(m0 + wm * TM + i * 32) * E.ldc + (n0 + wn * TN + j * 32);
                        const unsigned lo = (unsigned)(4 * lh) * (unsigned)E.ldc + (unsigned)l31;
                        float *apb = Ap + (wm * TM + i * 32 + 4 * lh) * BN + wn * TN + j * 32 + l31;
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float z0 = zp[i][j][r];
                            const float a0 = fmaf(z0, sc, sh);
                            const float v = a0 > 0.f ? acc[i][j][r] : 0.f;
                            float *tr = tb + (size_t)((r & 3) + 8 * (r >> 2)) * E.ldc;
                            if (!PNPP_WS_EXP_NO_STORE) tr[lo] = v;
                            t1 += v;
                            t2 = fmaf(v, (z0 - mu) * is, t2);
                            if constexpr (FDW) apb[((r & 3) + 8 * (r >> 2)) * BN] = fmaxf(a0, 0.f);
                        }
                        s1[j] += (double)t1, s2[j] += (double)t2;
                    }
            }
        }
        if (!done)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + wn * TN + j * 32 + l31;
                float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
                if constexpr (EMODE == E_MASK_STATS) {
                    const int cc = min(col, Nout - 1);
                    sc = E.scale[cc], sh = E.shift[cc], mu = E.mu[cc], is = E.istd[cc];
                }
                // statistics: this lane's 16 rows are summed in float32, the tiles of the worker in float64 (a float64
                // add per element costs several VALU slots, and VALU time is MFMA time here)
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = row < M && col < Nout;
                    float v = ok ? acc[i][j][r] : 0.f;
                    if constexpr (EMODE == E_STORE_STATS) {
                        t1 += v;
                        t2 = fmaf(v, v, t2);
                    } else if constexpr (EMODE == E_MASK_STATS) {
                        const float z0 = zp[i][j][r];
                        v = (fmaf(z0, sc, sh) > 0.f) ? v : 0.f;
                        t1 += v;
                        t2 = fmaf(v, (z0 - mu) * is, t2);
                    }
                    if (ok) E.c[(size_t)row * E.ldc + col] = v;
                    if constexpr (FDW) {  // a_{l-1} = relu(bn(z_{l-1})), the operand dW_l is contracted with
                        const int rl = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        Ap[rl * BN + wn * TN + j * 32 + l31] = fmaxf(fmaf(zp[i][j][r], sc, sh), 0.f);
                    }
                }
                if constexpr (EMODE != E_STORE) s1[j] += (double)t1, s2[j] += (double)t2;
            }
        PNPP_STAMP(5)
        if constexpr (FDW) {
            __syncthreads();  // the whole relu(bn(zp)) tile is in LDS; the dZ tile still is
            PNPP_STAMP(6)
            // dW tile (ct, kt) += dZ^T (columns ct*32.. of the A tile) x activation tile (columns kt*32..).  A wave's DT tiles
            // (tile_id = wave + 4 t) share kt, so one activation operand feeds DT MFMAs on DT independent accumulators.
            // The reduction index is the tile row m = 32 c + 2 t2 + lh; f(m) = f(2 t2) | (lh << 2) (the swizzle only looks at
            // m mod 16), so the swizzled column is ((ct*32 + l31) ^ (lh << 2)) ^ F(t2) with F(t2) = (t2 & 7) << 3 uniform:
            // scalar work plus one xor per tile and t2, and the BM / 32 reads of a t2 differ by immediate offsets only.
            static_assert(4 % (BN / 32) == 0, "a wave's dW tiles must share their activation columns");
            constexpr int MC = BM / 32;
            const int kt = wave % (BN / 32);
            int colx[DT];
#pragma unroll
            for (int t = 0; t < DT; ++t) colx[t] = (((wave + 4 * t) / (BN / 32)) * 32 + l31) ^ (lh << 2);
            const float *abase = As + lh * KP, *bbase = Ap + lh * BN + kt * 32 + l31;
            float da[2][MC][DT], db[2][MC];
            auto ld = [&](int buf, int t2) {
                const int F = (t2 & 7) << 3;
                const float *pb = bbase + 2 * t2 * BN;
#pragma unroll
                for (int c = 0; c < MC; ++c) db[buf][c] = pb[32 * c * BN];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const float *pa = abase + 2 * t2 * KP + (colx[t] ^ F);
#pragma unroll
                    for (int c = 0; c < MC; ++c) da[buf][c][t] = pa[32 * c * KP];
                }
            };
            auto mm = [&](int buf) {
#pragma unroll
                for (int c = 0; c < MC; ++c)
#pragma unroll
                    for (int t = 0; t < DT; ++t)
                        dwacc[t] = PNPP_WS_MFMA(da[buf][c][t], db[buf][c], dwacc[t]);
            };
            if constexpr (KD >= PNPP_WS_DWTABLE_MINK || TUNED_128) {
                // one wave per SIMD here, registers to spare: the swizzled operand addresses of the eight F values are a table
                // built once per tile from this lane's colx, and with t2 unrolled every read of the loop is (table entry) +
                // (immediate offset) -- no address arithmetic between the MFMAs (it was 45 VALU per 8 MFMAs)
                const float *pre[DT][8];
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int f = 0; f < 8; ++f) pre[t][f] = abase + 2 * f * KP + (colx[t] ^ (f << 3));
                auto ldt = [&](int buf, int t2) {
                    const int f = t2 & 7, h = t2 >> 3;
                    const float *pb = bbase + 2 * t2 * BN;
#pragma unroll
                    for (int c = 0; c < MC; ++c) db[buf][c] = pb[32 * c * BN];
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const float *pa = pre[t][f] + 16 * h * KP;
#pragma unroll
                        for (int c = 0; c < MC; ++c) da[buf][c][t] = pa[32 * c * KP];
                    }
                };
                ldt(0, 0);
#pragma unroll
                for (int t2 = 0; t2 < 16; t2 += 2) {
                    ldt(1, t2 + 1);
                    mm(0);
                    if (t2 + 2 < 16) ldt(0, t2 + 2);
                    mm(1);
                }
            } else {
                ld(0, 0);
#pragma unroll 1
                for (int t2 = 0; t2 < 16; t2 += 2) {
                    ld(1, t2 + 1);
                    mm(0);
                    if (t2 + 2 < 16) ld(0, t2 + 2);
                    mm(1);
                }
            }
            PNPP_STAMP(7)
        }
    }

    flush_held();
    PNPP_STAMP(9)       // (nothing: closes the last tile)
    if constexpr (FDW) {  // one partial dW per worker: dwslab[worker][c][n0 + k]
      if (n0 + BN <= Nout) {  // (uniform row pointer) + (one lane offset): scalar address arithmetic, no bounds test per element
        float *wb = E.dwslab + (size_t)worker * KD * E.dw_ld + n0;
        const unsigned lo = (unsigned)(4 * lh) * (unsigned)E.dw_ld + (unsigned)l31;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int tile_id = wave + 4 * t, ct = tile_id / (BN / 32), kt = tile_id % (BN / 32);
            float *tb = wb + (size_t)(ct * 32) * E.dw_ld + kt * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.dw_ld + lo] = dwacc[t][r];
        }
      } else
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int tile_id = wave + 4 * t, ct = tile_id / (BN / 32), kt = tile_id % (BN / 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, k = n0 + kt * 32 + l31;
                if (k < Nout) E.dwslab[((size_t)worker * KD + c) * E.dw_ld + k] = dwacc[t][r];
            }
        }
    }

    if constexpr (EMODE != E_STORE) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);  // [WM][2][BN]; the launcher sizes the LDS for it as well
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
            if (n0 + cl < Nout) E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
        }
    }
    PNPP_STAMP(10)      // tail: dW partial, statistics slab (stores complete)
}

template <int KD, int BM, int BN, int WM, int WN, int AM, int EM, bool FDW>
static int launch_ws_one(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                         int *dw_slabs) {
    const int tiles = cdiv(M, BM), ncol = cdiv(Nout, BN);
    size_t lds = ((size_t)KD * BN + (size_t)BM * (KD % 32 == 0 ? KD : KD + 1) + (FDW ? (size_t)BM * BN : 0)) * sizeof(float);
    const size_t red_bytes = (size_t)WM * 2 * BN * sizeof(double);  // column-statistics reduction reuses the LDS
    if (lds < red_bytes) lds = red_bytes;
    // persistent workers: as many workgroups as the LDS lets the chip hold at once (dW slabs and statistic slabs are
    // per worker, so fewer is cheaper), at most three per CU
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    int workers = (256 * per_cu) / ncol;
    if (workers > tiles) workers = tiles;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (workers < 1) workers = 1;
    if (nslab) *nslab = workers;
    if (dw_slabs) *dw_slabs = FDW ? workers : 0;
    ProfScope ps(st, "gemm_ws_kernel<%d,%d,%d,A%d,E%d%s> M=%d N=%d K=%d grid=%dx1", KD, BM, BN, AM, EM, FDW ? ",dW" : "", M, Nout,
                 KD, workers * ncol);
    auto kfn = gemm_ws_kernel<KD, BM, BN, WM, WN, AM, EM, FDW>;
    static size_t lds_granted = 0;  // per instantiation; the attribute call is a host-side setting, made once per size
    if (lds > 48 * 1024 && lds > lds_granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_granted = lds;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(256), lds, st, A, B, M, Nout, ncol, E);
    PNPP_CHECK_LAUNCH("gemm_ws");
    return PNPP_OK;
}

template <int KD, int BM, int BN, int WM, int WN, int AM>
static int launch_ws_e(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                       int *dw_slabs = nullptr) {
    if (dw_slabs) *dw_slabs = 0;
    switch (E.mode) {
        case E_STORE: return launch_ws_one<KD, BM, BN, WM, WN, AM, E_STORE, false>(A, B, M, Nout, E, nslab, st, nullptr);
        case E_STORE_STATS: return launch_ws_one<KD, BM, BN, WM, WN, AM, E_STORE_STATS, false>(A, B, M, Nout, E, nslab, st, nullptr);
        case E_MASK_STATS:
            if constexpr ((AM == A_DZ || AM == A_DZ_POOL) && KD % 32 == 0 && ((KD / 32) * (BN / 32)) % 4 == 0) {
                if (E.dwslab && dw_slabs)
                    return launch_ws_one<KD, BM, BN, WM, WN, AM, E_MASK_STATS, true>(A, B, M, Nout, E, nslab, st, dw_slabs);
            }
            return launch_ws_one<KD, BM, BN, WM, WN, AM, E_MASK_STATS, false>(A, B, M, Nout, E, nslab, st, nullptr);
    }
    set_error("gemm_ws: bad epilogue mode %d", E.mode);
    return PNPP_ERR_ARG;
}

// dense (non-grouped) operands: K in {64, 128, 256}
template <int KD, int BM, int BN, int WM, int WN>
static int launch_ws_dense(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                           int *dw_slabs) {
    switch (A.mode) {
        case A_PLAIN: return launch_ws_e<KD, BM, BN, WM, WN, A_PLAIN>(A, B, M, Nout, E, nslab, st);
        case A_BNRELU: return launch_ws_e<KD, BM, BN, WM, WN, A_BNRELU>(A, B, M, Nout, E, nslab, st);
        case A_DZ: return launch_ws_e<KD, BM, BN, WM, WN, A_DZ>(A, B, M, Nout, E, nslab, st, dw_slabs);
        case A_DZ_POOL: return launch_ws_e<KD, BM, BN, WM, WN, A_DZ_POOL>(A, B, M, Nout, E, nslab, st, dw_slabs);
    }
    set_error("gemm_ws: bad A mode %d", A.mode);
    return PNPP_ERR_ARG;
}

// picks a weights-stationary configuration, or returns false when the shape does not qualify (the chunked kernel
// then handles it): the reference models' grouped layers all qualify
static bool try_launch_ws(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab,
                          hipStream_t st, int *rc, int *dw_slabs) {
    if (M < 8192 || Nout % 64 != 0) return false;
    const bool grouped = A.mode == A_GATHER || A.mode == A_CONCAT;
    if (grouped) {
        if (Kd != A.D + 4) return false;
        if (A.mode != A_GATHER) return false;
        if (A.D == 0) {  // xyz only
            if (Nout % 128 == 0) *rc = launch_ws_e<4, 128, 128, 4, 1, A_GATHER>(A, B, M, Nout, E, nslab, st);
            else *rc = launch_ws_e<4, 128, 64, 4, 1, A_GATHER>(A, B, M, Nout, E, nslab, st);
            return true;
        }
        if (A.D == 128 && ((uintptr_t)A.a & 15) == 0) {
            *rc = launch_ws_e<132, 64, 64, 2, 2, A_GATHER>(A, B, M, Nout, E, nslab, st);
            return true;
        }
        return false;
    }
    if (A.lda % 4 != 0 || ((uintptr_t)A.a & 15) != 0) return false;
    if (Kd == 64) {   // 64 x 64 tiles for every K = 64 launch (measured against 128 x 128 / 128 x 64: forward 30.0 vs 31.9 and 19.3 vs
                      // 19.4 us, backward 36.7 vs 40.5 us on SA1: twice the tiles per worker, column blocks share an XCD's L2)
        *rc = launch_ws_dense<64, 64, 64, 2, 2>(A, B, M, Nout, E, nslab, st, dw_slabs);
        return true;
    }
    if (Kd == 128) {
        // (a 128-row tile with one workgroup per CU and the K = 256 kernel's unrolled, interleaved loops was measured at the same
        // 57.8 us for the SA1 backward launch and 4-7 % slower for the others)
        *rc = launch_ws_dense<128, 64, 64, 2, 2>(A, B, M, Nout, E, nslab, st, dw_slabs);
        return true;
    }
    if (Kd == 256) {
        *rc = launch_ws_dense<256, 64, 64, 2, 2>(A, B, M, Nout, E, nslab, st, dw_slabs);
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// small-M GEMM (fully connected head: M = batch rows).  One 32x32 output tile per workgroup; the
// reduction dimension is split over the 4 waves (chunk-interleaved), so a K=1024 layer is 8 chunks
// deep instead of 32.  A chunks go through wave-private LDS (row-major global -> lane-per-row
// operand), B (weights, [k][n] row-major) is read straight into the MFMA operand layout (the lane
// index is n: one 128-byte segment per half-wave).  Next chunk's loads fly during the MFMA loop.
// ---------------------------------------------------------------------------------------------
template <int AMODE, int EMODE, bool BT, int NW>
__device__ __forceinline__ void gemm_smallm_body(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E,
                                                 int bx, int by, int nblocks) {
    // per wave: A chunk [32][33] and weight chunk [32 n][33] (BT only); after the K loop the first NW x 1024 floats
    // are reused for the K-split partials [NW][32][32] (a wave's partial overwrites only its own A chunk)
    constexpr int NTHR = NW * 64, NJ = 1024 / NTHR;
    __shared__ __attribute__((aligned(16))) float lds[2 * NW * 32 * APITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    float *As = lds + wave * 32 * APITCH;
    float *Ws = lds + (NW + wave) * 32 * APITCH;
    float *part = lds;  // [NW][32][32], wave w at part + w * 1056: inside its own A chunk region (32*33 = 1056 floats)
    const float *__restrict__ Bm = B.b;
    const int ldb = B.ldb;
    const int n0 = bx * 32, m0 = by * 32;
    const int nchunks = (Kd + KC - 1) / KC;
    const bool bvec = (ldb & 3) == 0 && ((uintptr_t)Bm & 15) == 0 && B.perm_D < 0 && B.rows >= 4 && (B.rows & 3) == 0;  // uniform

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // E_BN_APPLY: the epilogue's per-column parameters and the dropout stream id are requested now -- behind the K loop and its
    // barriers each of them would be one more cold round trip (~1.5 us) on the tail of a 10 us launch
    float p_g = 1.f, p_b = 0.f, p_bias = 0.f, p_rm = 0.f, p_rv = 0.f;
    unsigned long long p_sid = 0ull;
    if constexpr (EMODE == E_BN_APPLY) {
        const BnTail &T = E.bn;
        if (tid < 32 && n0 + tid < Nout) {
            const int c = n0 + tid;
            if (T.gamma) p_g = T.gamma[c];
            if (T.beta) p_b = T.beta[c];
            if (T.bias) p_bias = T.bias[c];
            if (T.rm) p_rm = T.rm[c], p_rv = T.rv[c];
        }
        if (T.mask_out) p_sid = T.rng_counter[0];
    }

    RawA na[4];
    float4 nw[4];        // BT: weight rows, same (row, 4k) mapping as the A chunk
    float nb[KC / 2];    // !BT: weights already in operand layout
    auto fetch = [&](int c) {
        const int k0 = c * KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (lane >> 3) + 8 * i, k = k0 + 4 * (lane & 7);
            na[i] = fetch_a4<AMODE>(A, m0 + r, k, M, Kd);
            if constexpr (BT) {
                const float *wrow = Bm + (size_t)min(n0 + r, Nout - 1) * ldb;
                if (bvec) {
                    nw[i] = *reinterpret_cast<const float4 *>(wrow + min(k, B.rows - 4));
                } else {  // odd pitch or the layer-0 column permutation: four scalar loads (clamped; masked when staged)
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kp = min(k + e, B.rows - 1);
                        t[e] = wrow[B.perm_D >= 0 ? (kp < B.perm_D ? kp + 3 : kp - B.perm_D) : kp];
                    }
                    nw[i] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
        if constexpr (!BT) {
#pragma unroll
            for (int s2 = 0; s2 < KC / 2; ++s2) {
                const int k = k0 + 2 * s2 + lh;
                // mask by multiplication, not by a select: hipcc turns "cond ? loaded : 0" into a branch around the load
                // with its own vmcnt(0), which serialises the sixteen operand loads of a chunk
                const float v = Bm[(size_t)min(k, B.rows - 1) * ldb + min(n0 + l31, Nout - 1)];
                nb[s2] = v * ((k < B.rows && n0 + l31 < Nout) ? 1.f : 0.f);
            }
        }
    };
    if (wave < nchunks) fetch(wave);
    for (int c = wave; c < nchunks; c += NW) {
        float cb[KC / 2];
        const int k0 = c * KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (lane >> 3) + 8 * i, k = k0 + 4 * (lane & 7);
            const int o = r * APITCH + 4 * (lane & 7);
            float v[4];
            xform_a4<AMODE>(A, na[i], m0 + r, k, M, Kd, v);
            As[o] = v[0], As[o + 1] = v[1], As[o + 2] = v[2], As[o + 3] = v[3];
            if constexpr (BT) {
                const bool okw = n0 + r < Nout;
                Ws[o] = (okw && k < B.rows) ? nw[i].x : 0.f;
                Ws[o + 1] = (okw && k + 1 < B.rows) ? nw[i].y : 0.f;
                Ws[o + 2] = (okw && k + 2 < B.rows) ? nw[i].z : 0.f;
                Ws[o + 3] = (okw && k + 3 < B.rows) ? nw[i].w : 0.f;
            }
        }
        if constexpr (!BT) {
#pragma unroll
            for (int s2 = 0; s2 < KC / 2; ++s2) cb[s2] = nb[s2];
        }
        if (c + NW < nchunks) fetch(c + NW);
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2) {
            const float bv = BT ? Ws[l31 * APITCH + 2 * s2 + lh] : cb[s2];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[l31 * APITCH + 2 * s2 + lh], bv, acc, 0, 0, 0);
        }
    }
    // K-split reduction in fixed wave order, then the epilogue on the summed tile
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave * 32 * APITCH + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
    __syncthreads();
    float v[NJ], w2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = tid + NTHR * j;
        v[j] = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4)  // groups of four waves, in wave order
            v[j] += (part[w * 32 * APITCH + e] + part[(w + 1) * 32 * APITCH + e]) +
                    (part[(w + 2) * 32 * APITCH + e] + part[(w + 3) * 32 * APITCH + e]);
        w2[j] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e = tid + NTHR * j, row = m0 + e / 32, col = n0 + e % 32;
        const bool ok = row < M && col < Nout;
        float x = ok ? v[j] : 0.f;
        if constexpr (EMODE == E_MASK_STATS) {
            const int cc = min(col, Nout - 1);
            const float zp = E.zp[(size_t)min(row, M - 1) * E.ldc + cc];
            x = (fmaf(zp, E.scale[cc], E.shift[cc]) > 0.f) ? x : 0.f;
            w2[j] = x * ((zp - E.mu[cc]) * E.istd[cc]);
            part[1024 + e] = w2[j];
        }
        part[e] = x;
        if (ok) E.c[(size_t)row * E.ldc + col] = x;
    }
    if constexpr (EMODE == E_BN_APPLY) {
        // the whole batch is in this tile: finish the BatchNorm here (same fp64 sums, in the same order, as the
        // slab + bn_finalize_fwd route), then normalise, ReLU and mask the tile
        const BnTail &T = E.bn;
        float *cs = part + 2048;  // [2][32] scale / shift of this column block
        __syncthreads();
        if (tid < 32 && n0 + tid < Nout) {
            const int c = n0 + tid;
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < 32; ++r) {
                const double x = (double)part[r * 32 + tid];
                s1 += x;
                s2 += x * x;
            }
            const double count = (double)M;
            const double mu = s1 / count;
            double var = s2 / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)T.eps);
            const double g = (double)p_g, bt = (double)p_b;
            const float sc = (float)(g * is), sh = (float)(bt - mu * g * is);
            T.mean[c] = (float)mu;
            T.istd[c] = (float)is;
            T.scale[c] = sc;
            T.shift[c] = sh;
            cs[tid] = sc;
            cs[32 + tid] = sh;
            if (T.rm) {
                const double bmean = mu + (double)p_bias;  // the linear bias was folded out of z
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                T.rm[c] = (float)((1.0 - (double)T.momentum) * (double)p_rm + (double)T.momentum * bmean);
                T.rv[c] = (float)((1.0 - (double)T.momentum) * (double)p_rv + (double)T.momentum * unbiased);
            }
        }
        if (T.nbt && bx == 0 && tid == 0) *T.nbt += 1;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e = tid + NTHR * j, row = m0 + e / 32, col = n0 + e % 32;
            if (row < M && col < Nout) {
                float y = fmaf(part[e], cs[e % 32], cs[32 + e % 32]);
                if (T.relu) y = fmaxf(y, 0.f);
                if (T.mask) y = T.mask[(size_t)row * E.ldc + col] ? y * T.drop_scale : 0.f;
                if (T.mask_out) {  // draw the keep bit of this element: one Philox word per element (a few hundred per workgroup)
                    const unsigned long long sid = p_sid;
                    const unsigned idx = (unsigned)(row * E.ldc + col);
                    unsigned c0 = idx, c1 = 0x44524f50u /* "DROP" */, c2 = (unsigned)sid, c3 = (unsigned)(sid >> 32);
                    unsigned k0 = (unsigned)T.rng_seed, k1 = (unsigned)(T.rng_seed >> 32);
#pragma unroll
                    for (int rd = 0; rd < 10; ++rd) {
                        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
                        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
                        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
                        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
                    }
                    const bool keep = (double)c0 >= (double)T.drop_p * 4294967296.0;
                    T.mask_out[(size_t)row * E.ldc + col] = keep ? 1 : 0;
                    y = keep ? y * T.drop_scale : 0.f;
                }
                T.y[(size_t)row * E.ldc + col] = y;
            }
        }
        if (T.mask_out) {  // every workgroup has read the counter above before it takes a ticket; the last one bumps it
            __syncthreads();
            if (tid == 0) {
                const unsigned long long t = atomicAdd(&T.rng_counter[1], 1ull);
                if (t == (unsigned long long)nblocks - 1) {
                    T.rng_counter[1] = 0ull;
                    T.rng_counter[0] += 1ull;
                }
            }
        }
    } else if constexpr (EMODE != E_STORE) {
        __syncthreads();
        if (tid < 32 && n0 + tid < Nout) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < 32; ++r) {
                const double x = (double)part[r * 32 + tid];
                s1 += x;
                if constexpr (EMODE == E_STORE_STATS) s2 += x * x;
                else s2 += (double)part[1024 + r * 32 + tid];
            }
            E.slab[((size_t)by * 2 + 0) * Nout + n0 + tid] = s1;
            E.slab[((size_t)by * 2 + 1) * Nout + n0 + tid] = s2;
        }
    }
}

template <int AMODE, int EMODE, bool BT, int NW>
__global__ void __launch_bounds__(NW * 64)
gemm_smallm_kernel(const AOperand A, const BOperand B, int M, int Nout, int Kd, const Epilogue E) {
    gemm_smallm_body<AMODE, EMODE, BT, NW>(A, B, M, Nout, Kd, E, blockIdx.x, blockIdx.y, gridDim.x * gridDim.y);
}

template <int AM, int EM>
static void launch_smallm_t(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, dim3 grid,
                            hipStream_t st) {
    if constexpr (AM == A_PLAIN) {
        // one row of tiles (the head, M <= 32): few workgroups and a long reduction, so K is split over 16 waves -- two
        // 32-deep chunks per wave at K = 1024 instead of eight dependent load round trips
        if (grid.y == 1 && Kd >= 256) {
            if (B.trans) hipLaunchKernelGGL((gemm_smallm_kernel<AM, EM, true, 16>), grid, dim3(1024), 0, st, A, B, M, Nout, Kd, E);
            else hipLaunchKernelGGL((gemm_smallm_kernel<AM, EM, false, 16>), grid, dim3(1024), 0, st, A, B, M, Nout, Kd, E);
            return;
        }
    }
    if (B.trans) hipLaunchKernelGGL((gemm_smallm_kernel<AM, EM, true, 4>), grid, dim3(256), 0, st, A, B, M, Nout, Kd, E);
    else hipLaunchKernelGGL((gemm_smallm_kernel<AM, EM, false, 4>), grid, dim3(256), 0, st, A, B, M, Nout, Kd, E);
}
template <int AM>
static int launch_smallm_e(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, dim3 grid,
                           hipStream_t st) {
    switch (E.mode) {
        case E_STORE: launch_smallm_t<AM, E_STORE>(A, B, M, Nout, Kd, E, grid, st); return PNPP_OK;
        case E_STORE_STATS: launch_smallm_t<AM, E_STORE_STATS>(A, B, M, Nout, Kd, E, grid, st); return PNPP_OK;
        case E_MASK_STATS: launch_smallm_t<AM, E_MASK_STATS>(A, B, M, Nout, Kd, E, grid, st); return PNPP_OK;
        case E_BN_APPLY:
            if constexpr (AM == A_PLAIN) {
                PNPP_REQUIRE(grid.y == 1 && E.bn.mean && E.bn.istd && E.bn.scale && E.bn.shift && E.bn.y, PNPP_ERR_ARG,
                             "gemm(small M): the BatchNorm epilogue needs all rows in one tile (M <= 32) and its outputs");
                launch_smallm_t<AM, E_BN_APPLY>(A, B, M, Nout, Kd, E, grid, st);
                return PNPP_OK;
            }
            break;
    }
    set_error("gemm(small M): bad epilogue mode %d", E.mode);
    return PNPP_ERR_ARG;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_cfg(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab,
                           hipStream_t st) {
    const int tiles = cdiv(M, BM);
    const int gx = tiles < kMaxStatBlocks ? tiles : kMaxStatBlocks;
    const dim3 grid(gx, cdiv(Nout, BN)), block(WM * WN * 64);
    if (nslab) *nslab = gx;
    ProfScope ps(st, "gemm_kernel<%d,%d,%d,%d,A%d,E%d> M=%d N=%d K=%d grid=%dx%d", BM, BN, WM, WN, A.mode, E.mode, M, Nout, Kd,
                 grid.x, grid.y);
#define PNPP_LAUNCH(AM, EM)                                                                                   \
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AM, EM>), grid, block, 0, st, A, B, M, Nout, Kd, E); \
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
        case A_DZ_POOL: PNPP_BY_E(A_DZ_POOL)
        default: set_error("gemm: bad A mode %d", A.mode); return PNPP_ERR_ARG;
    }
#undef PNPP_BY_E
#undef PNPP_LAUNCH
    PNPP_CHECK_LAUNCH("gemm");
    return PNPP_OK;
}

// Epilogue::pool_ext is honoured by the interior epilogue of the float32 weights-stationary kernel with one column tile per wave
// (every dense launch of try_launch_ws); the caller asks before it relies on it
bool gemm_pools_in_epilogue(const AOperand &A, int M, int Nout, int Kd, int nsample) {
    if (nsample != 32) return false;
    if (M < 8192) return mid_tiles_on() && mid_gemm_pools(A, M, Nout, Kd);   // group_all levels of 32-point clouds: the 64 x 64 kernel
    if (matmul_precision() != 0) return false;
    if (M % 64 != 0 || Nout % 64 != 0) return false;
    if (!(A.mode == A_PLAIN || A.mode == A_BNRELU)) return false;
    if (A.lda % 4 != 0 || ((uintptr_t)A.a & 15) != 0) return false;
    return Kd == 64 || Kd == 128 || Kd == 256;
}

int launch_gemm(const AOperand &A, const BOperand &Bin, int M, int Nout, int Kd, const Epilogue &E, int *nslab,
                hipStream_t st, int *dw_slabs) {
    if (dw_slabs) *dw_slabs = 0;
    PNPP_REQUIRE(M > 0 && Nout > 0 && Kd > 0, PNPP_ERR_ARG, "gemm: non-positive size M=%d N=%d K=%d", M, Nout, Kd);
    PNPP_REQUIRE(Bin.b && Bin.ldb > 0, PNPP_ERR_ARG, "gemm: null B operand");
    PNPP_REQUIRE(Kd % 4 == 0, PNPP_ERR_ARG, "gemm: K=%d must be a multiple of 4", Kd);
    BOperand B = Bin;
    if (B.rows <= 0 || B.rows > Kd) B.rows = Kd;
    {
        int rc = PNPP_OK;
        if (try_launch_ws_bf16(A, B, M, Nout, Kd, E, nslab, st, &rc, dw_slabs)) return rc;   // only in the opt-in bf16-operand mode
        if (try_launch_wsf3(A, B, M, Nout, Kd, E, nslab, st, &rc)) return rc;  // the same products from exact bf16 splits (gemm_wsf3_kernels.hip)
        if (try_launch_wsf(A, B, M, Nout, Kd, E, nslab, st, &rc)) return rc;   // forward products: wave-private strips
        if (try_launch_wsd3(A, B, M, Nout, Kd, E, nslab, st, &rc, dw_slabs)) return rc;  // fused backward products of the grouped levels, split products, wave pairs
        if (try_launch_wsp(A, B, M, Nout, Kd, E, nslab, st, &rc, dw_slabs)) return rc;   // fused backward products, 64-channel input
        if (try_launch_wsq(A, B, M, Nout, Kd, E, nslab, st, &rc, dw_slabs)) return rc;   // the same for the 256-channel last layer
        if (try_launch_ws(A, B, M, Nout, Kd, E, nslab, st, &rc, dw_slabs)) return rc;
        if (mid_tiles_on() && try_launch_mid_gemm(A, B, M, Nout, Kd, E, nslab, st, &rc)) return rc;
    }
    const bool a_aligned = (A.mode == A_CONCAT || A.mode == A_GATHER) || (A.lda % 4 == 0 && ((uintptr_t)A.a & 15) == 0);
    if (M <= 4096 && cdiv(M, 32) <= kMaxStatBlocks && a_aligned && A.mode != A_GATHER) {
        // split-K 32x32 tiles: the fully connected head (M = batch) and the group_all layers (M = B * 32)
        const dim3 grid(cdiv(Nout, 32), cdiv(M, 32));
        if (nslab) *nslab = grid.y;
        ProfScope ps(st, "gemm_smallm_kernel<A%d,E%d,T%d> M=%d N=%d K=%d grid=%dx%d", A.mode, E.mode, B.trans, M, Nout, Kd, grid.x,
                     grid.y);
        int rc = PNPP_OK;
        switch (A.mode) {
            case A_PLAIN: rc = launch_smallm_e<A_PLAIN>(A, B, M, Nout, Kd, E, grid, st); break;
            case A_BNRELU: rc = launch_smallm_e<A_BNRELU>(A, B, M, Nout, Kd, E, grid, st); break;
            case A_CONCAT: rc = launch_smallm_e<A_CONCAT>(A, B, M, Nout, Kd, E, grid, st); break;
            case A_DZ: rc = launch_smallm_e<A_DZ>(A, B, M, Nout, Kd, E, grid, st); break;
            case A_DZ_POOL: rc = launch_smallm_e<A_DZ_POOL>(A, B, M, Nout, Kd, E, grid, st); break;
            default: set_error("gemm(small M): bad A mode %d", A.mode); return PNPP_ERR_ARG;
        }
        if (rc != PNPP_OK) return rc;
        PNPP_CHECK_LAUNCH("gemm(small M)");
        return PNPP_OK;
    }
    if (A.mode == A_PLAIN || A.mode == A_BNRELU || A.mode == A_DZ || A.mode == A_DZ_POOL)
        PNPP_REQUIRE(A.lda % 4 == 0 && ((uintptr_t)A.a & 15) == 0, PNPP_ERR_ARG, "gemm: A operand pitch/alignment");
    // tile shape: tall tiles for the grouped layers (M = B*npoint*nsample), square-ish for small M
    if (M >= 128 * 128) {
        if (Nout % 128 == 0) return launch_gemm_cfg<128, 128, 4, 2>(A, B, M, Nout, Kd, E, nslab, st);
        if (Nout % 64 == 0) return launch_gemm_cfg<128, 64, 4, 2>(A, B, M, Nout, Kd, E, nslab, st);
        return launch_gemm_cfg<128, 32, 4, 1>(A, B, M, Nout, Kd, E, nslab, st);
    }
    if (M > 32) {
        if (Nout % 64 == 0) return launch_gemm_cfg<64, 64, 2, 2>(A, B, M, Nout, Kd, E, nslab, st);
        return launch_gemm_cfg<128, 32, 4, 1>(A, B, M, Nout, Kd, E, nslab, st);
    }
    if (Nout % 128 == 0) return launch_gemm_cfg<32, 128, 1, 4>(A, B, M, Nout, Kd, E, nslab, st);
    return launch_gemm_cfg<128, 32, 4, 1>(A, B, M, Nout, Kd, E, nslab, st);
}

// ---------------------------------------------------------------------------------------------
// small-M layers (group_all: M = 32 * B): the BatchNorm-backward operand dZ is materialised once (a few MB)
// instead of being rebuilt by every 32 x 32 output tile of the dA and dW GEMMs that consume it
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) dz_materialize_kernel(const AOperand A, int M, int C, float *__restrict__ out) {
    const size_t total = (size_t)M * (C / 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / (C / 4)), k = 4 * (int)(i % (C / 4));
        const RawA r = fetch_a4<MODE>(A, row, k, M, C);
        float v[4];
        xform_a4<MODE>(A, r, row, k, M, C, v);
        *reinterpret_cast<float4 *>(out + (size_t)row * C + k) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

int launch_dz_materialize(const AOperand &dz, int M, int C, float *out, hipStream_t st) {
    PNPP_REQUIRE(C % 4 == 0 && (dz.mode == A_DZ || dz.mode == A_DZ_POOL), PNPP_ERR_ARG, "dz_materialize: bad operand");
    const size_t total = (size_t)M * (C / 4);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    ProfScope ps(st, "dz_materialize_kernel M=%d C=%d", M, C);
    if (dz.mode == A_DZ) hipLaunchKernelGGL(dz_materialize_kernel<A_DZ>, dim3(grid), dim3(256), 0, st, dz, M, C, out);
    else hipLaunchKernelGGL(dz_materialize_kernel<A_DZ_POOL>, dim3(grid), dim3(256), 0, st, dz, M, C, out);
    PNPP_CHECK_LAUNCH("dz_materialize");
    return PNPP_OK;
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

    // A_DZ_POOL: a batch of 2U rows lies inside one group when K % 2U == 0 (row ranges start on multiples of 2U),
    // so the pooled gradient / arg-max of the group are fetched once per batch instead of once per row
    const bool grp_batch = (DZMODE == A_DZ_POOL) && (dz.K % (2 * U) == 0) && (r0 % (2 * U) == 0);
    for (int row = r0; row < r1; row += 2 * U) {
        float2 fa[U][CT], fb[U][KT];
        float gdm[CT];
        int garg[CT], gk0 = 0;
        if (DZMODE == A_DZ_POOL && grp_batch) {
            const int g = row / dz.K;
            gk0 = row - g * dz.K;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int c = min(c0 + i * 32 + l31, Nc - 1);
                gdm[i] = dz.a[(size_t)g * dz.lda + c];
                garg[i] = dz.arg[(size_t)g * dz.lda + c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = row + 2 * u + lh;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                if (DZMODE == A_DZ_POOL && grp_batch) {
                    const int c = min(c0 + i * 32 + l31, Nc - 1);
                    fa[u][i].x = (gk0 + 2 * u + lh == garg[i]) ? gdm[i] : 0.f;
                    fa[u][i].y = dz.z[(size_t)min(m, M - 1) * dz.lda + c];
                } else {
                    fa[u][i] = fetch_a1<DZMODE>(dz, m, c0 + i * 32 + l31, Nc, M);
                }
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) fb[u][j] = fetch_a1<A2MODE>(a2, m, k0 + j * 32 + l31, Kp, M);
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

// ---------------------------------------------------------------------------------------------
// dW of an xyz-only layer 0 (SA1: C x 3): dW[c][k] = sum_m dZ[m][c] * (xyz[nbr(m)][k] - centre(m)[k]).
// The 64 x 64 MFMA tiles of dw_kernel would spend 61 of their 64 reduction columns on padding; this is a streaming
// VALU kernel instead: a workgroup takes 256 consecutive rows, stages their three relative coordinates in LDS (one
// row per thread: neighbour index, gather, float32 subtraction as in the forward) and then walks the rows with
// lane = channel (64 channels x 4 row lanes), so dY and Z are read once, fully coalesced, and dZ is rebuilt on the
// fly.  Output: one partial [C][4] per workgroup in the slab layout slab_reduce expects (pitch 4).
// ---------------------------------------------------------------------------------------------
template <int DZMODE, int A2MODE>  // A2MODE: A_GATHER (neighbourhoods) or A_CONCAT (group_all on raw coordinates: centre = origin)
__global__ void __launch_bounds__(256)
dw_xyz_kernel(const AOperand dz, const AOperand a2, int M, int C, float *__restrict__ slab) {
    __shared__ float rel[256][4];
    __shared__ float red[4][64][3];
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int m0 = blockIdx.x * 256;
    {  // this thread's row: relative coordinates, zero for rows beyond M
        const float2 x = fetch_a1<A2MODE>(a2, m0 + tid, 0, 3, M);
        const float2 y = fetch_a1<A2MODE>(a2, m0 + tid, 1, 3, M);
        const float2 z = fetch_a1<A2MODE>(a2, m0 + tid, 2, 3, M);
        const float okf = (m0 + tid < M) ? 1.f : 0.f;
        rel[tid][0] = __fsub_rn(x.x, x.y) * okf;
        rel[tid][1] = __fsub_rn(y.x, y.y) * okf;
        rel[tid][2] = __fsub_rn(z.x, z.y) * okf;
        rel[tid][3] = 0.f;
    }
    __syncthreads();
    for (int c0 = blockIdx.y * 64; c0 < C; c0 += gridDim.y * 64) {
        const int c = c0 + cl;
        const ChanConst cc = load_chan_const<DZMODE>(dz, min(c, C - 1), C);
        float a0 = 0.f, a1 = 0.f, a2s = 0.f;
#pragma unroll 8
        for (int r = rl; r < 256; r += 4) {
            const float2 f = fetch_a1<DZMODE>(dz, m0 + r, c, C, M);
            const float g = xform_a1<DZMODE>(f, cc, c, C, true);  // rows >= M meet zero coordinates
            const float4 q = *reinterpret_cast<const float4 *>(rel[r]);
            a0 = fmaf(g, q.x, a0), a1 = fmaf(g, q.y, a1), a2s = fmaf(g, q.z, a2s);
        }
        red[rl][cl][0] = a0, red[rl][cl][1] = a1, red[rl][cl][2] = a2s;
        __syncthreads();
        if (tid < 192) {
            const int ch = tid / 3, k = tid - 3 * ch;
            if (c0 + ch < C)
                slab[((size_t)blockIdx.x * C + c0 + ch) * 4 + k] = (red[0][ch][k] + red[1][ch][k]) + (red[2][ch][k] + red[3][ch][k]);
        }
        __syncthreads();
    }
}

// number of partial slabs launch_dw_xyz writes (pitch 4), for sizing
int dw_xyz_splits(int M) { return cdiv(M, 256); }

template <int A2MODE>
static int launch_dw_xyz_a2(const AOperand &dz, int C, const AOperand &a2, int M, float *slab, dim3 grid, hipStream_t st) {
    switch (dz.mode) {
        case A_PLAIN: hipLaunchKernelGGL((dw_xyz_kernel<A_PLAIN, A2MODE>), grid, dim3(256), 0, st, dz, a2, M, C, slab); break;
        case A_DZ: hipLaunchKernelGGL((dw_xyz_kernel<A_DZ, A2MODE>), grid, dim3(256), 0, st, dz, a2, M, C, slab); break;
        case A_DZ_POOL: hipLaunchKernelGGL((dw_xyz_kernel<A_DZ_POOL, A2MODE>), grid, dim3(256), 0, st, dz, a2, M, C, slab); break;
        default: set_error("dw_xyz: bad dZ mode %d", dz.mode); return PNPP_ERR_ARG;
    }
    return PNPP_OK;
}

int launch_dw_xyz(const AOperand &dz, int C, const AOperand &a2, int M, float *slab, hipStream_t st) {
    PNPP_REQUIRE((a2.mode == A_GATHER || a2.mode == A_CONCAT) && a2.D == 0, PNPP_ERR_ARG,
                 "dw_xyz: the second operand must be xyz-only (gathered or whole-cloud)");
    PNPP_REQUIRE(M > 0 && C > 0, PNPP_ERR_ARG, "dw_xyz: non-positive size");
    const dim3 grid(dw_xyz_splits(M), 1);
    ProfScope ps(st, "dw_xyz_kernel<A%d,A%d> M=%d N=%d K=3 grid=%dx1", dz.mode, a2.mode, M, C, grid.x);
    const int rc = a2.mode == A_GATHER ? launch_dw_xyz_a2<A_GATHER>(dz, C, a2, M, slab, grid, st)
                                       : launch_dw_xyz_a2<A_CONCAT>(dz, C, a2, M, slab, grid, st);
    if (rc != PNPP_OK) return rc;
    PNPP_CHECK_LAUNCH("dw_xyz");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// Layer 0 of a grouped set abstraction with input features ("convolve, then gather").  The 1x1 convolution is linear,
// so for row (group s, neighbour k) with source point j = idx[s][k]
//     z = W_xyz (x_j - c_s) + W_f f_j  =  P[j] + W_xyz (x_j - c_s),      P = F W_f^T   (one row per SOURCE point)
// P costs B*N rows of GEMM instead of B*S*K (8x fewer for SA2); the relative-coordinate term keeps the reference's
// float32 subtraction (pointnet_pp_8dir.py:28-31) and is three FMAs per output.  This kernel builds Z (row-major,
// pre-BN, no bias: BatchNorm cancels it) and the per-channel sum / sum of squares partials of the BN statistics.
// Thread = 4 channels of one row; a workgroup walks `rpb` consecutive rows, 256 / (C/4) at a time.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gather_rel_stats_kernel(const float *__restrict__ P, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                        const int32_t *__restrict__ idx, const float *__restrict__ W0, int ldw, int N, int S, int K, int M,
                        int C, int rpb, float *__restrict__ z, double *__restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) double gred[];  // [RPP][2][C]
    const int LPR = C >> 2, RPP = 256 / LPR;
    const int cl = threadIdx.x % LPR, rl = threadIdx.x / LPR, c4 = cl * 4;
    float wx[4], wy[4], wz[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float *w = W0 + (size_t)(c4 + e) * ldw;
        wx[e] = w[0], wy[e] = w[1], wz[e] = w[2];
    }
    const int r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
    double d1[4] = {0.0, 0.0, 0.0, 0.0}, d2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rb = r0 + rl; rb < r1; rb += 4 * RPP) {
        float4 p[4];
        float rx[4], ry[4], rz[4], ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four rows in flight per thread
            const int r = rb + u * RPP;
            const int rc = min(r, r1 - 1);
            ok[u] = r < r1 ? 1.f : 0.f;
            const int grp = rc / K;
            const size_t src = (size_t)(grp / S) * N + idx[rc];
            p[u] = *reinterpret_cast<const float4 *>(P + src * C + c4);
            const float *x = xyz + src * 3, *c = new_xyz + (size_t)grp * 3;
            rx[u] = __fsub_rn(x[0], c[0]), ry[u] = __fsub_rn(x[1], c[1]), rz[u] = __fsub_rn(x[2], c[2]);
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[4] = {p[u].x, p[u].y, p[u].z, p[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaf(rz[u], wz[e], fmaf(ry[u], wy[e], fmaf(rx[u], wx[e], v[e])));
                const float m = v[e] * ok[u];
                s1[e] += m, s2[e] = fmaf(m, m, s2[e]);
            }
            const int r = rb + u * RPP;
            if (r < r1) *reinterpret_cast<float4 *>(z + (size_t)r * C + c4) = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d1[e] += (double)s1[e], d2[e] += (double)s2[e];
    }
    if (slab == nullptr) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gred[(rl * 2 + 0) * C + c4 + e] = d1[e];
        gred[(rl * 2 + 1) * C + c4 + e] = d2[e];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < 2 * C; f += 256) {
        const int which = f / C, c = f - which * C;
        double t = 0.0;
        for (int w = 0; w < RPP; ++w) t += gred[(w * 2 + which) * C + c];
        slab[((size_t)blockIdx.x * 2 + which) * C + c] = t;
    }
}

bool delayed_layer0_ok(int C) {  // C/4 lanes per row must divide the 256-thread workgroup; the scatter holds C <= 512
    return C >= 32 && C <= 512 && (C & (C - 1)) == 0;
}

int launch_gather_rel_stats(const float *P, const AOperand &geo, const float *W0, int ldw, int M, int C, float *z,
                            double *slab, int *nslab, hipStream_t st) {
    PNPP_REQUIRE(delayed_layer0_ok(C) && geo.mode == A_GATHER, PNPP_ERR_ARG, "gather_rel_stats: unsupported width %d", C);
    const int rpp = 256 / (C / 4);
    int rpb = 64;  // rows per workgroup: a multiple of the 4 * rpp rows in flight, at most kMaxStatBlocks workgroups
    while (rpb < 4 * rpp || cdiv(M, rpb) > kMaxStatBlocks) rpb *= 2;
    const int grid = cdiv(M, rpb);
    if (nslab) *nslab = grid;
    ProfScope ps(st, "gather_rel_stats_kernel M=%d C=%d grid=%d", M, C, grid);
    hipLaunchKernelGGL(gather_rel_stats_kernel, dim3(grid), dim3(256), (size_t)rpp * 2 * C * sizeof(double), st, P, geo.xyz,
                       geo.new_xyz, geo.idx, W0, ldw, geo.N, geo.S, geo.K, M, C, rpb, z, slab);
    PNPP_CHECK_LAUNCH("gather_rel_stats");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the same layer: the gradient reaches the feature weights and the source features only through
//     G[j] = sum over the rows r of the cloud with idx[r] == j of dZ[r]            (one row per SOURCE point)
// (dW_f = G^T F and dF = G W_f are then B*N-row GEMMs).  One wavefront per source point scans its cloud's neighbour
// lists in order, 1024 entries per pass, compacts the matching rows into a list (ballot + prefix count, so list order
// is row order) and adds them in that order: a fixed summation order, no atomics.  dZ is rebuilt on the fly from the
// masked upstream gradient and Z (A_DZ): two coalesced row reads per match, up to eight matches in flight.  Lane = 2 channels of each 128-channel chunk.  The same pass accumulates the C x 3 gradient of
// the coordinate columns, dW_xyz = sum_r dZ[r] (x_j - c_s)^T, as one [C][4] partial per workgroup (slab_reduce layout).
// ---------------------------------------------------------------------------------------------
constexpr int SCW = 8;       // wavefronts (= source points) per workgroup
constexpr int SCWIN = 1024;  // neighbour-list entries examined per pass
template <int NCH>
__global__ void __launch_bounds__(SCW * 64)
scatter_dz_kernel(const AOperand dz, const AOperand geo, int Mc, int C, int total, float *__restrict__ G,
                  float *__restrict__ wslab) {
    constexpr int UB = NCH == 1 ? 8 : 4;  // matching rows fetched per batch
    __shared__ int hl[SCW][SCWIN];
    __shared__ float wred[SCW][NCH * 128][3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int N = geo.N;
    const int dst = min(blockIdx.x * SCW + wv, total - 1);  // b * N + n; a surplus wave repeats the last point, writes nothing
    const bool live = blockIdx.x * SCW + wv < total;
    const int b = dst / N, n = dst - b * N;
    const int32_t *ib = geo.idx + (size_t)b * Mc;
    const size_t row0 = (size_t)b * Mc;
    const float px = geo.xyz[(size_t)dst * 3], py = geo.xyz[(size_t)dst * 3 + 1], pz = geo.xyz[(size_t)dst * 3 + 2];
    const float *cb = geo.new_xyz + (size_t)b * geo.S * 3;
    float2 cg[NCH], cmu[NCH], cis[NCH], c1[NCH], c2[NCH], acc[NCH], ax[NCH], ay[NCH], az[NCH];
    int cc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        cc[j] = min(j * 128 + 2 * lane, C - 2);
        cg[j] = make_float2(1.f, 1.f);
        cmu[j] = cis[j] = c1[j] = c2[j] = acc[j] = ax[j] = ay[j] = az[j] = make_float2(0.f, 0.f);
        if (dz.mode == A_DZ) {
            const float *p = dz.cst + cc[j];
            cg[j] = *reinterpret_cast<const float2 *>(p), cmu[j] = *reinterpret_cast<const float2 *>(p + dz.C);
            cis[j] = *reinterpret_cast<const float2 *>(p + 2 * dz.C), c1[j] = *reinterpret_cast<const float2 *>(p + 3 * dz.C);
            c2[j] = *reinterpret_cast<const float2 *>(p + 4 * dz.C);
        }
    }
    const float *zsrc = dz.mode == A_DZ ? dz.z : dz.a;  // a materialised dZ (small levels) passes through: g = 1, c1 = c2 = 0
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int w0 = 0; w0 < Mc; w0 += SCWIN) {  // uniform over the workgroup
        // 1. the rows of this window that point at n, in row order, as a list in LDS
        int v[SCWIN / 64];
#pragma unroll
        for (int q = 0; q < SCWIN / 64; ++q) v[q] = ib[min(w0 + q * 64 + lane, Mc - 1)];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < SCWIN / 64; ++q) {
            const int m = w0 + q * 64 + lane;
            const bool mine = live && m < Mc && v[q] == n;
            const unsigned long long bal = __ballot(mine);
            if (mine) hl[wv][cnt + __popcll(bal & below)] = m;
            cnt += __popcll(bal);
        }
        __syncthreads();
        // 2. their dZ rows, UB at a time, added in list order
        for (int i = 0; i < cnt; i += UB) {  // cnt is wave-uniform
            int pos[UB];
            float mk[UB], rx[UB], ry[UB], rz[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                mk[u] = i + u < cnt ? 1.f : 0.f;
                pos[u] = hl[wv][min(i + u, cnt - 1)];
                const float *c = cb + (size_t)(pos[u] / geo.K) * 3;  // the forward's float32 subtraction
                rx[u] = __fsub_rn(px, c[0]) * mk[u], ry[u] = __fsub_rn(py, c[1]) * mk[u], rz[u] = __fsub_rn(pz, c[2]) * mk[u];
            }
            float2 gy[UB][NCH], gz[UB][NCH];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const size_t o = (row0 + pos[u]) * dz.lda + cc[j];
                    gy[u][j] = *reinterpret_cast<const float2 *>(dz.a + o);
                    gz[u][j] = *reinterpret_cast<const float2 *>(zsrc + o);
                }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float vx = cg[j].x * (gy[u][j].x - c1[j].x - (gz[u][j].x - cmu[j].x) * cis[j].x * c2[j].x);
                    const float vy = cg[j].y * (gy[u][j].y - c1[j].y - (gz[u][j].y - cmu[j].y) * cis[j].y * c2[j].y);
                    acc[j].x = fmaf(vx, mk[u], acc[j].x), acc[j].y = fmaf(vy, mk[u], acc[j].y);
                    ax[j].x = fmaf(vx, rx[u], ax[j].x), ax[j].y = fmaf(vy, rx[u], ax[j].y);
                    ay[j].x = fmaf(vx, ry[u], ay[j].x), ay[j].y = fmaf(vy, ry[u], ay[j].y);
                    az[j].x = fmaf(vx, rz[u], az[j].x), az[j].y = fmaf(vy, rz[u], az[j].y);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 128 + 2 * lane;
        if (live && c < C) *reinterpret_cast<float2 *>(G + (size_t)dst * C + c) = acc[j];
        wred[wv][c][0] = ax[j].x, wred[wv][c][1] = ay[j].x, wred[wv][c][2] = az[j].x;
        wred[wv][c + 1][0] = ax[j].y, wred[wv][c + 1][1] = ay[j].y, wred[wv][c + 1][2] = az[j].y;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < C * 3; f += SCW * 64) {  // this workgroup's share of dW_xyz: [C][4] partial, waves in order
        const int c = f / 3, k = f - 3 * c;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SCW; ++w) t += wred[w][c][k];
        wslab[((size_t)blockIdx.x * C + c) * 4 + k] = t;
    }
}

int scatter_dz_splits(int rows) { return cdiv(rows, SCW); }

int launch_scatter_dz(const AOperand &dz, const AOperand &geo, int B, int Mc, int C, float *G, float *wslab, hipStream_t st) {
    PNPP_REQUIRE((dz.mode == A_PLAIN || (dz.mode == A_DZ && dz.C == C)) && dz.lda == C && delayed_layer0_ok(C), PNPP_ERR_ARG,
                 "scatter_dz: bad operand");
    PNPP_REQUIRE(geo.mode == A_GATHER && geo.S * geo.K == Mc, PNPP_ERR_ARG, "scatter_dz: bad geometry");
    const int total = B * geo.N;
    ProfScope ps(st, "scatter_dz_kernel B=%d N=%d C=%d M=%d", B, geo.N, C, Mc);
    const dim3 grid(scatter_dz_splits(total));
    if (C <= 128) hipLaunchKernelGGL(scatter_dz_kernel<1>, grid, dim3(SCW * 64), 0, st, dz, geo, Mc, C, total, G, wslab);
    else if (C <= 256) hipLaunchKernelGGL(scatter_dz_kernel<2>, grid, dim3(SCW * 64), 0, st, dz, geo, Mc, C, total, G, wslab);
    else hipLaunchKernelGGL(scatter_dz_kernel<4>, grid, dim3(SCW * 64), 0, st, dz, geo, Mc, C, total, G, wslab);
    PNPP_CHECK_LAUNCH("scatter_dz");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// dW for the small-M levels (group_all: M = 32 B rows, wide layers).  dw_kernel's waves each pull their own operand
// rows from L2 one dword per lane; here a workgroup owns a 128 x 128 block of dW over one row range, stages 32-row
// chunks of both operands through LDS with 16-byte loads (8 per thread and chunk instead of 64 dword loads per lane),
// and its four waves (64 x 64 each, 2 x 2 MFMA tiles) read them back lane-per-column -- every staged element feeds two
// MFMA tiles.  Same partial-slab output as dw_kernel (slab[split][c][kp_pad]).
// ---------------------------------------------------------------------------------------------
template <int DZMODE, int A2MODE>
__device__ __forceinline__ void dw_lds_body(const AOperand &dz, const AOperand &a2, int M, int Nc, int Kp, int tilesC, int tilesK, int rps,
                                            int kp_pad, float *__restrict__ slab, int bx) {
    __shared__ __attribute__((aligned(16))) float Dz[32][128];
    __shared__ __attribute__((aligned(16))) float A2[32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles = tilesC * tilesK;
    const int tile = bx % tiles, split = bx / tiles;
    const int c0 = (tile % tilesC) * 128, k0 = (tile / tilesC) * 128;
    const int r0 = min(M, split * rps), r1 = min(M, r0 + rps);  // an empty range still writes its (zero) slab block
    const int wc = wave >> 1, wk = wave & 1;
    const int q4 = 4 * (tid & 31), rb = tid >> 5;  // staging map: 4 columns of rows rb + 8 i

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    RawA nd[4], na[4];
    auto fetch = [&](int m0) {  // rows >= r1 belong to the next split: r1 plays M for the loaders (clamped loads, zeroed values)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            nd[i] = fetch_a4<DZMODE>(dz, m0 + rb + 8 * i, c0 + q4, r1, Nc);
            na[i] = fetch_a4<A2MODE>(a2, m0 + rb + 8 * i, k0 + q4, r1, Kp);
        }
    };
    if (r0 < r1) fetch(r0);
    for (int m0 = r0; m0 < r1; m0 += 32) {
        __syncthreads();  // the previous chunk's operand reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4];
            xform_a4<DZMODE>(dz, nd[i], m0 + rb + 8 * i, c0 + q4, r1, Nc, v);
            *reinterpret_cast<float4 *>(&Dz[rb + 8 * i][q4]) = make_float4(v[0], v[1], v[2], v[3]);
            xform_a4<A2MODE>(a2, na[i], m0 + rb + 8 * i, k0 + q4, r1, Kp, v);
            *reinterpret_cast<float4 *>(&A2[rb + 8 * i][q4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        if (m0 + 32 < r1) fetch(m0 + 32);  // the next chunk's loads fly during the MFMA loop
        const float *pd = &Dz[lh][wc * 64 + l31], *pa = &A2[lh][wk * 64 + l31];
#pragma unroll 4
        for (int rp = 0; rp < 16; ++rp) {  // reduction index = row 2 rp + lh of the chunk
            const float d0 = pd[rp * 256], d1 = pd[rp * 256 + 32], b0 = pa[rp * 256], b1 = pa[rp * 256 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float *o = slab + (size_t)split * Nc * kp_pad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wk * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wc * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c < Nc && k < kp_pad) o[(size_t)c * kp_pad + k] = acc[i][j][r];
            }
        }
}

template <int DZMODE, int A2MODE>
__global__ void __launch_bounds__(256)
dw_lds_kernel(const AOperand dz, const AOperand a2, int M, int Nc, int Kp, int tilesC, int tilesK, int rps, int kp_pad,
              float *__restrict__ slab) {
    dw_lds_body<DZMODE, A2MODE>(dz, a2, M, Nc, Kp, tilesC, tilesK, rps, kp_pad, slab, blockIdx.x);
}

// The two products of a small-M backward layer that only share their input -- dA = dZ W (32 x 32 split-K tiles) and
// dW = dZ^T A (LDS-staged 128 x 128 blocks) -- in ONE launch: the first g1 workgroups take the GEMM tiles, the rest
// the dW blocks.  These launches are latency-bound, so the pair costs about as much as the longer of the two.
template <int EMODE, int A2MODE>
__global__ void __launch_bounds__(256)
da_dw_kernel(const AOperand dzA, const BOperand W, int M, int Nout, int Kd, const Epilogue E, int g1x, int g1, const AOperand a2, int Nc,
             int Kp, int tilesC, int tilesK, int rps, int kp_pad, float *__restrict__ slab) {
    if ((int)blockIdx.x < g1)
        gemm_smallm_body<A_PLAIN, EMODE, false, 4>(dzA, W, M, Nout, Kd, E, blockIdx.x % g1x, blockIdx.x / g1x, g1);
    else
        dw_lds_body<A_PLAIN, A2MODE>(dzA, a2, M, Nc, Kp, tilesC, tilesK, rps, kp_pad, slab, blockIdx.x - g1);
}

void dw_plan(int M, int Nc, int Kp, int *nsplit, int *kp_pad) {
    const int tilesC = cdiv(Nc, 64), tilesK = cdiv(Kp, 64);
    const int tiles = tilesC * tilesK;
    // aim for ~2048 waves (2 per SIMD), at least 64 rows per wave (32 for the small-M layers, whose waves are latency
    // bound: twice the waves in flight beats the doubled slab count), at most 1024 partial slabs
    int split = cdiv(2048, tiles);
    const int max_split = cdiv(M, M <= 4096 ? 32 : 64);
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
    rps = (rps + 7) & ~7;  // multiple of 8: a fetch batch (4 row pairs) never straddles two splits or two groups
    if (M <= 4096 && Nc >= 128 && Kp >= 128 && dz.mode == A_PLAIN && (dz.lda & 3) == 0 && ((uintptr_t)dz.a & 15) == 0 &&
        (a2.mode == A_PLAIN || a2.mode == A_BNRELU || a2.mode == A_CONCAT) &&
        (a2.mode == A_CONCAT || ((a2.lda & 3) == 0 && ((uintptr_t)a2.a & 15) == 0))) {
        // small-M, wide layers: LDS-staged 128 x 128 blocks
        const int tc = cdiv(Nc, 128), tk = cdiv(Kp, 128);
        const dim3 grid(tc * tk * nsplit);
        ProfScope ps(st, "dw_lds_kernel<A%d,A%d> M=%d N=%d K=%d split=%d grid=%d", dz.mode, a2.mode, M, Nc, Kp, nsplit, grid.x);
        switch (a2.mode) {
            case A_PLAIN:
                hipLaunchKernelGGL((dw_lds_kernel<A_PLAIN, A_PLAIN>), grid, dim3(256), 0, st, dz, a2, M, Nc, Kp, tc, tk, rps, kp_pad, slab);
                break;
            case A_BNRELU:
                hipLaunchKernelGGL((dw_lds_kernel<A_PLAIN, A_BNRELU>), grid, dim3(256), 0, st, dz, a2, M, Nc, Kp, tc, tk, rps, kp_pad, slab);
                break;
            default:
                hipLaunchKernelGGL((dw_lds_kernel<A_PLAIN, A_CONCAT>), grid, dim3(256), 0, st, dz, a2, M, Nc, Kp, tc, tk, rps, kp_pad, slab);
                break;
        }
        PNPP_CHECK_LAUNCH("dw(lds)");
        return PNPP_OK;
    }
    const int waves = tilesC * tilesK * nsplit;
    const dim3 grid(cdiv(waves, 4)), block(256);
    ProfScope ps(st, "dw_kernel<A%d,A%d> M=%d N=%d K=%d split=%d grid=%dx1", dz.mode, a2.mode, M, Nc, Kp, nsplit, grid.x);
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
        case A_DZ_POOL: PNPP_DW_BY_A(A_DZ_POOL)
        default: set_error("dw: bad dZ mode %d", dz.mode); return PNPP_ERR_ARG;
    }
#undef PNPP_DW_BY_A
#undef PNPP_DW
    PNPP_CHECK_LAUNCH("dw");
    return PNPP_OK;
}

// The head layers (M <= 32 rows): dx = dz W as one row of 32 x 32 split-K tiles (16 waves) and dW = dz^T x as an outer
// product (no reduction worth an MFMA tile; workgroup = 32 output rows n x 128 columns k, dz and x tiles in LDS, each
// thread one k and four n) only share dz: one launch, the first g1 workgroups take the GEMM tiles.
__device__ __forceinline__ void dw_fewrows_body(const float *__restrict__ dz, const float *__restrict__ x, int M, int N, int K,
                                                float *__restrict__ dw, int bx, int by) {
    __shared__ __attribute__((aligned(16))) float dzs[32][32];
    __shared__ float xs[32][128];
    const int kl = threadIdx.x & 127, nh = threadIdx.x >> 7;  // 1024 threads: 8 groups of 4 output rows
    const int k0 = bx * 128, n0 = by * 32;
    for (int f = threadIdx.x; f < 32 * 32; f += 1024) {
        const int m = f >> 5, n = f & 31;
        dzs[m][n] = (m < M && n0 + n < N) ? dz[(size_t)m * N + n0 + n] : 0.f;
    }
    for (int f = threadIdx.x; f < 32 * 128; f += 1024) {
        const int m = f >> 7, k = f & 127;
        xs[m][k] = (m < M && k0 + k < K) ? x[(size_t)m * K + k0 + k] : 0.f;
    }
    __syncthreads();
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
            if (n < N) dw[(size_t)n * K + k0 + kl] = acc[j];
        }
}

__global__ void __launch_bounds__(1024)
fc_dx_dw_kernel(const AOperand dzA, const BOperand W, int M, int Nout, int Kd, const Epilogue E, int g1, const float *__restrict__ x,
                int N, int K, int gx2, float *__restrict__ dw) {
    if ((int)blockIdx.x < g1) gemm_smallm_body<A_PLAIN, E_STORE, false, 16>(dzA, W, M, Nout, Kd, E, blockIdx.x, 0, g1);
    else dw_fewrows_body(dzA.a, x, M, N, K, dw, (blockIdx.x - g1) % gx2, (blockIdx.x - g1) / gx2);
}

// dz: (M, N) row-major with pitch N; W: (N, K) row-major; dx: (M, K); dw: (N, K).  false = not this form, nothing launched.
bool try_launch_fc_dx_dw(const float *dz, const float *w, const float *x, int M, int N, int K, float *dx, float *dw, hipStream_t st,
                         int *rc) {
    *rc = PNPP_OK;
    if (!(M <= 32 && N >= 256 && N % 4 == 0 && K % 4 == 0 && ((uintptr_t)dz & 15) == 0)) return false;
    AOperand A;
    A.a = dz;
    A.lda = N;
    BOperand B;
    B.b = w;
    B.ldb = K;
    B.rows = N;
    Epilogue E;
    E.mode = E_STORE;
    E.c = dx;
    E.ldc = K;
    const int g1 = cdiv(K, 32), gx2 = cdiv(K, 128), gy2 = cdiv(N, 32);
    ProfScope ps(st, "fc_dx_dw_kernel M=%d N=%d K=%d grid=%d+%d", M, N, K, g1, gx2 * gy2);
    hipLaunchKernelGGL(fc_dx_dw_kernel, dim3(g1 + gx2 * gy2), dim3(1024), 0, st, A, B, M, K, N, E, g1, x, N, K, gx2, dw);
    if (hipGetLastError() != hipSuccess) {
        set_error("fc_dx_dw: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

// dA (+ its epilogue) and dW of one small-M backward layer in one launch; returns false (nothing launched) when the
// pair does not fit that form, and the caller launches the two separately.
bool try_launch_da_dw(const AOperand &dz, const BOperand &Win, int M, int Nout, int Kd, const Epilogue &E, int *nslab, const AOperand &a2,
                      int Kp, float *slab, int *nsplit_io, int *kp_pad_io, hipStream_t st, int *rc, float *dw_direct, int dw_ld) {
    *rc = PNPP_OK;
    if (mid_tiles_on() && try_launch_mid_da_dw(dz, Win, M, Nout, Kd, E, nslab, a2, Kp, slab, nsplit_io, kp_pad_io, st, rc, dw_direct, dw_ld))
        return true;   // wide layers of a group_all level: 64 x 64 tiles over the whole reduction, no 64-row partials
    const int nsplit = *nsplit_io, kp_pad = *kp_pad_io;
    const int Nc = Kd;  // dZ is M x Nc; dA = dZ W contracts over Nc, dW is Nc x Kp
    if (!(M > 32 && M <= 4096 && cdiv(M, 32) <= kMaxStatBlocks && dz.mode == A_PLAIN && (dz.lda & 3) == 0 && ((uintptr_t)dz.a & 15) == 0))
        return false;
    if (!(Nc >= 128 && Kp >= 128 && Kd % 4 == 0 && !Win.trans && (E.mode == E_STORE || E.mode == E_MASK_STATS))) return false;
    if (!(a2.mode == A_PLAIN || a2.mode == A_BNRELU || a2.mode == A_CONCAT)) return false;
    if (a2.mode != A_CONCAT && ((a2.lda & 3) != 0 || ((uintptr_t)a2.a & 15) != 0)) return false;
    if (kp_pad != cdiv(Kp, 64) * 64 || nsplit < 1) return false;
    BOperand W = Win;
    if (W.rows <= 0 || W.rows > Kd) W.rows = Kd;
    const int g1x = cdiv(Nout, 32), g1y = cdiv(M, 32), g1 = g1x * g1y;
    const int tc = cdiv(Nc, 128), tk = cdiv(Kp, 128);
    int rps = cdiv(M, nsplit);
    rps = (rps + 7) & ~7;
    const dim3 grid(g1 + tc * tk * nsplit);
    if (nslab) *nslab = g1y;
    ProfScope ps(st, "da_dw_kernel<E%d,A%d> M=%d | dA N=%d K=%d grid=%d | dW N=%d K=%d split=%d grid=%d", E.mode, a2.mode, M, Nout, Kd, g1,
                 Nc, Kp, nsplit, tc * tk * nsplit);
#define PNPP_DADW(EM, AM) \
    hipLaunchKernelGGL((da_dw_kernel<EM, AM>), grid, dim3(256), 0, st, dz, W, M, Nout, Kd, E, g1x, g1, a2, Nc, Kp, tc, tk, rps, kp_pad, slab)
    if (E.mode == E_STORE) {
        if (a2.mode == A_PLAIN) PNPP_DADW(E_STORE, A_PLAIN);
        else if (a2.mode == A_BNRELU) PNPP_DADW(E_STORE, A_BNRELU);
        else PNPP_DADW(E_STORE, A_CONCAT);
    } else {
        if (a2.mode == A_PLAIN) PNPP_DADW(E_MASK_STATS, A_PLAIN);
        else if (a2.mode == A_BNRELU) PNPP_DADW(E_MASK_STATS, A_BNRELU);
        else PNPP_DADW(E_MASK_STATS, A_CONCAT);
    }
#undef PNPP_DADW
    if (hipGetLastError() != hipSuccess) {
        set_error("da_dw: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

// out[c][perm(k)] = sum_s slab[s][c][k], fixed summation order: block = EPB outputs x (256/EPB) split lanes,
// every lane strides the splits with four independent partial sums, the lanes are combined in lane order.
// Small outputs (a 64 x 3 weight) take 16 outputs per block so that the splits, not the outputs, fill the chip.
struct SlabReduceArgs {
    const float *slab;
    int nsplit, Nc, kp_pad, Kvalid, perm_D;
    float *out;
    int ldo;
};

template <int EPB>
__device__ __forceinline__ void slab_reduce_block(const SlabReduceArgs &R, int bid) {
    constexpr int SL = 256 / EPB;
    __shared__ float red[SL][EPB];
    const int total = R.Nc * R.Kvalid;
    const int e = threadIdx.x % EPB, sl = threadIdx.x / EPB;
    const int i = bid * EPB + e;
    float acc = 0.f;
    int c = 0, k = 0;
    if (i < total) {
        c = i / R.Kvalid, k = i - c * R.Kvalid;
        const float *p = R.slab + (size_t)c * R.kp_pad + k;
        const size_t stride = (size_t)R.Nc * R.kp_pad;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = sl;
        for (; s + 3 * SL < R.nsplit; s += 4 * SL) {
            a0 += p[(size_t)s * stride];
            a1 += p[(size_t)(s + SL) * stride];
            a2 += p[(size_t)(s + 2 * SL) * stride];
            a3 += p[(size_t)(s + 3 * SL) * stride];
        }
        for (; s < R.nsplit; s += SL) a0 += p[(size_t)s * stride];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < SL; ++j) t += red[j][e];
        int ko = k;
        if (R.perm_D >= 0) ko = k < R.perm_D ? k + 3 : k - R.perm_D;  // features-first -> xyz-first (state_dict order)
        R.out[(size_t)c * R.ldo + ko] = t;
    }
}

// the same reduction on groups of four consecutive k (16-byte loads and stores, a quarter of the threads and load
// instructions): block = EPB groups x (256/EPB) split lanes.  Needs Kvalid, kp_pad, ldo multiples of 4, no permutation.
template <int EPB>
__device__ __forceinline__ void slab_reduce_block4(const SlabReduceArgs &R, int bid) {
    constexpr int SL = 256 / EPB;
    __shared__ float4 red4[SL][EPB];
    const int kg = R.Kvalid >> 2, total = R.Nc * kg;
    const int e = threadIdx.x % EPB, sl = threadIdx.x / EPB;
    const int i = bid * EPB + e;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0, k = 0;
    if (i < total) {
        c = i / kg, k = 4 * (i - c * kg);
        const float *p = R.slab + (size_t)c * R.kp_pad + k;
        const size_t stride = (size_t)R.Nc * R.kp_pad;
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = sl;
        for (; s + 3 * SL < R.nsplit; s += 4 * SL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 q = *reinterpret_cast<const float4 *>(p + (size_t)(s + u * SL) * stride);
                a[u].x += q.x, a[u].y += q.y, a[u].z += q.z, a[u].w += q.w;
            }
        }
        for (; s < R.nsplit; s += SL) {
            const float4 q = *reinterpret_cast<const float4 *>(p + (size_t)s * stride);
            a[0].x += q.x, a[0].y += q.y, a[0].z += q.z, a[0].w += q.w;
        }
        acc = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                          (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
    }
    red4[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            const float4 q = red4[j][e];
            t.x += q.x, t.y += q.y, t.z += q.z, t.w += q.w;
        }
        *reinterpret_cast<float4 *>(R.out + (size_t)c * R.ldo + k) = t;
    }
}

static inline bool slab_reduce_vec4(const SlabReduceArgs &R) {  // few, wide partials (measured: no gain once nsplit > 32)
    return R.nsplit <= 32 && R.perm_D < 0 && (R.Kvalid & 3) == 0 && (R.kp_pad & 3) == 0 && (R.ldo & 3) == 0 && (((uintptr_t)R.slab | (uintptr_t)R.out) & 15) == 0;
}
// groups per block for the 16-byte form: few splits -> many groups per block; many splits -> many split lanes
static inline int slab_reduce_epb4(int nsplit) { return nsplit <= 32 ? 64 : nsplit <= 128 ? 16 : 4; }

template <int EPB, bool V4 = false>
__global__ void __launch_bounds__(256) slab_reduce_kernel(SlabReduceArgs R) {
    if constexpr (V4) slab_reduce_block4<EPB>(R, blockIdx.x);
    else slab_reduce_block<EPB>(R, blockIdx.x);
}

static inline bool slab_reduce_wide(int total, int nsplit) { return total >= 16384 || nsplit <= 8; }

// two independent reductions in one launch (scalar form): blocks [0, n1) take R1, the rest R2
template <int EPB1, int EPB2>
__global__ void __launch_bounds__(256) slab_reduce2_kernel(SlabReduceArgs R1, int n1, SlabReduceArgs R2) {
    if ((int)blockIdx.x < n1) slab_reduce_block<EPB1>(R1, blockIdx.x);
    else slab_reduce_block<EPB2>(R2, blockIdx.x - n1);
}

int launch_slab_reduce2(const float *slab1, int nsplit1, int Nc1, int kp_pad1, int Kvalid1, float *out1, int ldo1, const float *slab2,
                        int nsplit2, int Nc2, int kp_pad2, int Kvalid2, float *out2, int ldo2, hipStream_t st) {
    const SlabReduceArgs R1{slab1, nsplit1, Nc1, kp_pad1, Kvalid1, -1, out1, ldo1}, R2{slab2, nsplit2, Nc2, kp_pad2, Kvalid2, -1, out2, ldo2};
    const int t1 = Nc1 * Kvalid1, t2 = Nc2 * Kvalid2;
    const bool w1 = slab_reduce_wide(t1, nsplit1), w2 = slab_reduce_wide(t2, nsplit2);
    const int n1 = cdiv(t1, w1 ? 64 : 16), n2 = cdiv(t2, w2 ? 64 : 16);
    ProfScope ps(st, "slab_reduce2_kernel N=%d K=%d split=%d | N=%d K=%d split=%d", Nc1, Kvalid1, nsplit1, Nc2, Kvalid2, nsplit2);
    if (w1 && w2) hipLaunchKernelGGL((slab_reduce2_kernel<64, 64>), dim3(n1 + n2), dim3(256), 0, st, R1, n1, R2);
    else if (w1) hipLaunchKernelGGL((slab_reduce2_kernel<64, 16>), dim3(n1 + n2), dim3(256), 0, st, R1, n1, R2);
    else if (w2) hipLaunchKernelGGL((slab_reduce2_kernel<16, 64>), dim3(n1 + n2), dim3(256), 0, st, R1, n1, R2);
    else hipLaunchKernelGGL((slab_reduce2_kernel<16, 16>), dim3(n1 + n2), dim3(256), 0, st, R1, n1, R2);
    PNPP_CHECK_LAUNCH("slab_reduce2");
    return PNPP_OK;
}

int launch_slab_reduce(const float *slab, int nsplit, int Nc, int kp_pad, int Kvalid, int perm_D, float *out, int ldo,
                       hipStream_t st) {
    const int total = Nc * Kvalid;
    const SlabReduceArgs R{slab, nsplit, Nc, kp_pad, Kvalid, perm_D, out, ldo};
    ProfScope ps(st, "slab_reduce_kernel N=%d K=%d split=%d", Nc, Kvalid, nsplit);
    if (slab_reduce_vec4(R)) {
        const int groups = total / 4, epb = slab_reduce_epb4(nsplit);
        if (epb == 64) hipLaunchKernelGGL((slab_reduce_kernel<64, true>), dim3(cdiv(groups, 64)), dim3(256), 0, st, R);
        else if (epb == 16) hipLaunchKernelGGL((slab_reduce_kernel<16, true>), dim3(cdiv(groups, 16)), dim3(256), 0, st, R);
        else hipLaunchKernelGGL((slab_reduce_kernel<4, true>), dim3(cdiv(groups, 4)), dim3(256), 0, st, R);
    } else if (slab_reduce_wide(total, nsplit))
        hipLaunchKernelGGL(slab_reduce_kernel<64>, dim3(cdiv(total, 64)), dim3(256), 0, st, R);
    else
        hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3(cdiv(total, 16)), dim3(256), 0, st, R);
    PNPP_CHECK_LAUNCH("slab_reduce");
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
#pragma unroll 4  // 16 independent loads in flight per lane: the reduction is a chain of L2 round trips otherwise
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
                       float *__restrict__ rv, long long *__restrict__ nbt, float momentum, float eps, int training,
                       float *__restrict__ mean, float *__restrict__ istd, float *__restrict__ scale,
                       float *__restrict__ shift, const double *__restrict__ count_dev, const float *__restrict__ pool_ext,
                       float *__restrict__ pool_out, int G, int32_t *__restrict__ pool_arg, float *__restrict__ origin_a,
                       float *__restrict__ origin_b, int norigin) {
    __shared__ double red[32][2][FIN_COLS];
    if (blockIdx.x == 0 && blockIdx.y == 0)   // group_all levels: every cloud's centre is the origin (pointnet_pp_8dir.py:24)
        for (int i = threadIdx.x; i < norigin; i += 256) {
            if (origin_a) origin_a[i] = 0.f;
            if (origin_b) origin_b[i] = 0.f;
        }
    __shared__ float pool_cs[2][FIN_COLS];
    if (count_dev) count = *count_dev;   // SyncBN: the row count of ALL ranks, summed with the statistics
    // pooling in the producer's epilogue (Epilogue::pool_ext): gridDim.y row blocks each redo the slab reduction for their 8
    // channels (identical sums, identical order) and turn their rows of the extreme pre-BN values into the pooled output;
    // the statistics themselves are written by row block 0 only
    const bool writer = blockIdx.y == 0;
    if (training && nbt && blockIdx.x == 0 && writer && threadIdx.x == 0) *nbt += 1;  // num_batches_tracked (nn.BatchNorm forward)
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    // The per-channel parameters are requested BEFORE the slab reduction: at a kernel boundary every line is a cold miss of this
    // XCD's L2 (~1.5 us), the reduction ends in a barrier the compiler will not move loads across, and a launch this short is
    // the sum of its dependent round trips -- one instead of two.
    const bool owner = threadIdx.x < FIN_COLS && c < C;
    float p_bias = 0.f, p_g = 1.f, p_b = 0.f, p_rm = 0.f, p_rv = 0.f;
    if (owner || (!training && c < C)) {
        if (bias) p_bias = bias[c];
        if (rm) p_rm = rm[c], p_rv = rv[c];
    }
    if (owner) {
        if (gamma) p_g = gamma[c];
        if (beta) p_b = beta[c];
    }
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
        mu = (double)p_rm - (double)p_bias;
        var = (double)p_rv;
    }
    if (owner) {
        const double is = 1.0 / sqrt(var + (double)eps);
        const double g = (double)p_g, bt = (double)p_b;
        const float sc = (float)(g * is), sh = (float)(bt - mu * g * is);
        if (pool_out) pool_cs[0][threadIdx.x] = sc, pool_cs[1][threadIdx.x] = sh;
        if (writer) {
            mean[c] = (float)mu;
            istd[c] = (float)is;
            scale[c] = sc;
            shift[c] = sh;
            if (training && rm) {
                const double bmean = mu + (double)p_bias;  // the conv/linear bias was folded out of z
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                rm[c] = (float)((1.0 - (double)momentum) * (double)p_rm + (double)momentum * bmean);
                rv[c] = (float)((1.0 - (double)momentum) * (double)p_rv + (double)momentum * unbiased);
            }
        }
    }
    if (!pool_out) return;
    __syncthreads();
    // out[g][c] = relu(scale * ext + shift): 8 consecutive channels (32 bytes) of 32 rows per pass
    const int cl = threadIdx.x % FIN_COLS, cc = blockIdx.x * FIN_COLS + cl;
    if (cc >= C) return;
    const float sc = pool_cs[0][cl], sh = pool_cs[1][cl];
    const int rows_per = (G + gridDim.y - 1) / gridDim.y, g0 = blockIdx.y * rows_per, g1 = min(G, g0 + rows_per);
    for (int gg = g0 + threadIdx.x / FIN_COLS; gg < g1; gg += 256 / FIN_COLS) {
        const size_t i = (size_t)gg * C + cc;
        const float v = fmaf(pool_ext[i], sc, sh);
        pool_out[i] = fmaxf(v, 0.f);
        // a neighbourhood whose activations are all zero routes (no) gradient through its first row, as torch.max over the
        // post-ReLU values does -- and the backward pass then reads one z row per group instead of a scattered one
        if (pool_arg && !(v > 0.f)) pool_arg[i] = 0;
    }
}

struct BnFinalizeBwdArgs {
    const double *slab;
    int nslab, C;
    double count;
    int training;
    const float *gamma, *mean, *istd;
    float *cst, *dgamma, *dbeta, *dbias;
    // SyncBN: `slab` holds the sums over ALL ranks (one slab), *count_dev their row count; the parameter gradients stay this
    // rank's own sums (`local`: [2][C]) -- the gradient all-reduce adds the ranks up, as it does for every other parameter
    const double *count_dev = nullptr, *local = nullptr;
    // Pooled source (levels with few groups: the group_all level has one per cloud): the column sums are taken straight from the pooled
    // gradient -- sum over the G groups of d = ReLU'(scale zsel + shift) dout and of d xhat(zsel) -- instead of from slabs a pool_bwd
    // launch would have written; the dZ job rebuilds d the same way.  No pool_bwd launch, no dm tensor.
    const float *p_dout = nullptr, *p_zsel = nullptr, *p_scale = nullptr, *p_shift = nullptr;
    int p_G = 0;
};

// column sums of one channel from the pooled source, groups in order
__device__ __forceinline__ void pooled_column_sums(const BnFinalizeBwdArgs &F, int c, double &s1, double &s2) {
    const float sc = F.p_scale[c], sh = F.p_shift[c], mu = F.mean[c], is = F.istd[c];
    s1 = 0.0, s2 = 0.0;
    for (int g0 = 0; g0 < F.p_G; g0 += 8) {   // eight groups' two streams in flight at a time (all 32 at once measured the same)
        float za[8], dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t gi = (size_t)min(g0 + j, F.p_G - 1) * F.C + c;
            za[j] = F.p_zsel[gi], dv[j] = F.p_dout[gi];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = (g0 + j < F.p_G && fmaf(za[j], sc, sh) > 0.f) ? dv[j] : 0.f;
            s1 += (double)d, s2 += (double)d * (double)((za[j] - mu) * is);
        }
    }
}

__device__ __forceinline__ void bn_finalize_bwd_block(const BnFinalizeBwdArgs &F, int bid) {
    __shared__ double red[32][2][FIN_COLS];
    const int C = F.C;
    const int c = bid * FIN_COLS + (threadIdx.x % FIN_COLS);
    // (parameters first, reduction second: see bn_finalize_fwd_kernel)
    const bool owner = threadIdx.x < FIN_COLS && c < C;
    float g = 1.f, p_is = 0.f, p_mu = 0.f;
    if (owner) {
        if (F.gamma) g = F.gamma[c];
        p_is = F.istd[c], p_mu = F.mean[c];
    }
    double s1, s2;
    if (F.p_dout) {
        if (!owner) return;
        pooled_column_sums(F, c, s1, s2);
    } else {
        slab_column_sums(F.slab, F.nslab, C, c, s1, s2, red);
        if (!owner) return;
    }
    const double count = F.count_dev ? *F.count_dev : F.count;
    float *cst = F.cst;
    cst[c] = g * p_is;
    cst[C + c] = p_mu;
    cst[2 * C + c] = p_is;
    cst[3 * C + c] = F.training ? (float)(s1 / count) : 0.f;
    cst[4 * C + c] = F.training ? (float)(s2 / count) : 0.f;
    if (F.local) s1 = F.local[c], s2 = F.local[C + c];
    if (F.dgamma) F.dgamma[c] = (float)s2;
    if (F.dbeta) F.dbeta[c] = (float)s1;
    // a bias in front of a train-mode BatchNorm has exactly zero gradient; with running
    // statistics the layer is affine and d(bias) = sum_m dz = g * sum_m dy
    if (F.dbias) F.dbias[c] = F.training ? 0.f : (float)((double)(g * p_is) * s1);
}

// Small-M levels materialise dZ once per layer (dz_materialize_kernel).  The BatchNorm-backward constants it needs are
// a 32-slab column reduction, so the workgroups that write dZ redo that reduction for their own 64 columns (in their
// own fixed order: the constants can differ from `cst` in the last float32 bit, deterministically) and the materialisation rides in the launch that finalises: no
// launch of its own, no wait for `cst`.
struct DzJob {
    const float *dy = nullptr;   // masked upstream gradient (M x C), or the pooled gradient (G x C) when arg != nullptr
    const float *z = nullptr;    // pre-BN activations (M x C)
    const int32_t *arg = nullptr;  // pooled form: arg-max neighbour per (group, channel)
    int K = 1;                   // pooled form: rows per group
    int M = 0;
    float *out = nullptr;        // dZ (M x C); nullptr = no job
};

__device__ __forceinline__ void dz_fused_block(const BnFinalizeBwdArgs &F, const DzJob &J, int bid) {
    __shared__ double red[4][2][64];
    __shared__ float kc[5][64];  // g, mu, istd, c1, c2 of this block's 64 columns
    const int C = F.C, ncg = (C + 63) / 64;
    const int c0 = (bid % ncg) * 64, r0 = (bid / ncg) * 64;
    {   // column sums of the slabs: 64 columns x 4 slab lanes, every load of a lane in flight at once, lanes combined in order
        const int cl = threadIdx.x & 63, q = threadIdx.x >> 6, c = min(c0 + cl, C - 1);
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
        if (F.p_dout) {   // pooled source: lane 0 of the four takes the whole column (a few dozen groups)
            if (q == 0) pooled_column_sums(F, c, a1, a2);
        }
        int sidx = F.p_dout ? F.nslab : q;
#pragma unroll 4
        for (; sidx + 4 < F.nslab; sidx += 8) {
            a1 += F.slab[((size_t)sidx * 2 + 0) * C + c];
            a2 += F.slab[((size_t)sidx * 2 + 1) * C + c];
            b1 += F.slab[((size_t)(sidx + 4) * 2 + 0) * C + c];
            b2 += F.slab[((size_t)(sidx + 4) * 2 + 1) * C + c];
        }
        for (; sidx < F.nslab; sidx += 4) {
            a1 += F.slab[((size_t)sidx * 2 + 0) * C + c];
            a2 += F.slab[((size_t)sidx * 2 + 1) * C + c];
        }
        red[q][0][cl] = a1 + b1;
        red[q][1][cl] = a2 + b2;
        __syncthreads();
        if (threadIdx.x < 64) {
            const double s1 = (red[0][0][cl] + red[1][0][cl]) + (red[2][0][cl] + red[3][0][cl]);
            const double s2 = (red[0][1][cl] + red[1][1][cl]) + (red[2][1][cl] + red[3][1][cl]);
            const float g = F.gamma ? F.gamma[c] : 1.f;
            kc[0][cl] = g * F.istd[c];
            kc[1][cl] = F.mean[c];
            kc[2][cl] = F.istd[c];
            const double count = F.count_dev ? *F.count_dev : F.count;
            kc[3][cl] = F.training ? (float)(s1 / count) : 0.f;
            kc[4][cl] = F.training ? (float)(s2 / count) : 0.f;
        }
        __syncthreads();
    }
    const int q4 = 4 * (threadIdx.x & 15), c = c0 + q4;
    if (c >= C) return;
    const float4 g = *reinterpret_cast<const float4 *>(&kc[0][q4]), mu = *reinterpret_cast<const float4 *>(&kc[1][q4]);
    const float4 is = *reinterpret_cast<const float4 *>(&kc[2][q4]), c1 = *reinterpret_cast<const float4 *>(&kc[3][q4]);
    const float4 c2 = *reinterpret_cast<const float4 *>(&kc[4][q4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + (threadIdx.x >> 4) + 16 * i;
        if (row >= J.M) continue;
        const float4 z = *reinterpret_cast<const float4 *>(J.z + (size_t)row * C + c);
        float4 dy;
        if (J.arg) {
            const int grp = row / J.K, kk = row - grp * J.K;
            float4 dm = *reinterpret_cast<const float4 *>(J.dy + (size_t)grp * C + c);
            if (F.p_dout) {   // J.dy is dout itself: d = ReLU'(scale zsel + shift) dout, as pool_bwd_kernel writes it
                const float4 zs = *reinterpret_cast<const float4 *>(F.p_zsel + (size_t)grp * C + c);
                const float4 ps = *reinterpret_cast<const float4 *>(F.p_scale + c), ph = *reinterpret_cast<const float4 *>(F.p_shift + c);
                dm.x = fmaf(zs.x, ps.x, ph.x) > 0.f ? dm.x : 0.f, dm.y = fmaf(zs.y, ps.y, ph.y) > 0.f ? dm.y : 0.f;
                dm.z = fmaf(zs.z, ps.z, ph.z) > 0.f ? dm.z : 0.f, dm.w = fmaf(zs.w, ps.w, ph.w) > 0.f ? dm.w : 0.f;
            }
            const int4 ia = *reinterpret_cast<const int4 *>(J.arg + (size_t)grp * C + c);
            dy = make_float4(kk == ia.x ? dm.x : 0.f, kk == ia.y ? dm.y : 0.f, kk == ia.z ? dm.z : 0.f, kk == ia.w ? dm.w : 0.f);
        } else {
            dy = *reinterpret_cast<const float4 *>(J.dy + (size_t)row * C + c);
        }
        float4 o;  // the operand loaders' formula (xform_a4<A_DZ>)
        o.x = g.x * (dy.x - c1.x - (z.x - mu.x) * is.x * c2.x);
        o.y = g.y * (dy.y - c1.y - (z.y - mu.y) * is.y * c2.y);
        o.z = g.z * (dy.z - c1.z - (z.z - mu.z) * is.z * c2.z);
        o.w = g.w * (dy.w - c1.w - (z.w - mu.w) * is.w * c2.w);
        *reinterpret_cast<float4 *>(J.out + (size_t)row * C + c) = o;
    }
}
static inline int dz_job_blocks(const DzJob &J, int C) { return J.out ? ((C + 63) / 64) * ((J.M + 63) / 64) : 0; }

__global__ void __launch_bounds__(256) bn_finalize_bwd_kernel(BnFinalizeBwdArgs F, int nfin, DzJob J) {
    if ((int)blockIdx.x < nfin) bn_finalize_bwd_block(F, blockIdx.x);
    else dz_fused_block(F, J, blockIdx.x - nfin);
}

// the two reductions that follow a backward GEMM -- the weight-gradient partials of layer l and the BatchNorm-backward
// column sums of layer l-1 -- share one launch: the first nfin workgroups finalise, the rest reduce slabs
template <int EPB, bool V4 = false>
__global__ void __launch_bounds__(256) post_gemm_kernel(BnFinalizeBwdArgs F, int nfin, SlabReduceArgs R, int ndz, DzJob J) {
    if ((int)blockIdx.x < nfin) bn_finalize_bwd_block(F, blockIdx.x);
    else if ((int)blockIdx.x < nfin + ndz) dz_fused_block(F, J, blockIdx.x - nfin);
    else if constexpr (V4) slab_reduce_block4<EPB>(R, blockIdx.x - nfin - ndz);
    else slab_reduce_block<EPB>(R, blockIdx.x - nfin - ndz);
}

// SyncBN: the [nslab][2][C] partials of THIS rank reduced to one [2][C] slab followed by the row count, written twice -- `glob`
// is summed over the ranks in place by the registered exchange, `local` keeps this rank's own sums for the parameter gradients
__global__ void __launch_bounds__(256)
slab_sum_kernel(const double *__restrict__ slab, int nslab, int C, double count, double *__restrict__ glob, double *__restrict__ local) {
    __shared__ double red[32][2][FIN_COLS];
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    double s1, s2;
    slab_column_sums(slab, nslab, C, c, s1, s2, red);
    if (threadIdx.x < FIN_COLS && c < C) {
        glob[c] = s1, glob[C + c] = s2;
        local[c] = s1, local[C + c] = s2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) glob[2 * C] = count, local[2 * C] = count;
}

int launch_slab_sum(const double *slab, int nslab, int C, double count, double *glob, double *local, hipStream_t st) {
    ProfScope ps(st, "slab_sum_kernel C=%d", C);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(C, FIN_COLS)), dim3(256), 0, st, slab, nslab, C, count, glob, local);
    PNPP_CHECK_LAUNCH("slab_sum");
    return PNPP_OK;
}

int launch_bn_finalize_fwd(const double *slab, int nslab, int C, double count, const float *bias, const float *gamma,
                           const float *beta, float *rm, float *rv, long long *nbt, float momentum, float eps, int training,
                           float *mean, float *istd, float *scale, float *shift, hipStream_t st, const double *count_dev,
                           const float *pool_ext, float *pool_out, int G, int32_t *pool_arg, float *origin_a, float *origin_b,
                           int norigin) {
    const bool pool = pool_ext && pool_out && G > 0 && training;
    int gy = 1;
    if (pool) {   // enough row blocks to fill the chip, at least 64 rows each
        gy = cdiv(256, cdiv(C, FIN_COLS));
        if (gy > cdiv(G, 64)) gy = cdiv(G, 64);
        if (gy < 1) gy = 1;
    }
    ProfScope ps(st, "bn_finalize_fwd_kernel C=%d%s", C, pool ? " +pool" : "");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, FIN_COLS), gy), dim3(256), 0, st, slab, nslab, C, count, bias, gamma, beta,
                       rm, rv, nbt, momentum, eps, training, mean, istd, scale, shift, count_dev, pool ? pool_ext : nullptr,
                       pool ? pool_out : nullptr, G, pool ? pool_arg : nullptr, pool ? origin_a : nullptr, pool ? origin_b : nullptr,
                       pool ? norigin : 0);
    PNPP_CHECK_LAUNCH("bn_finalize_fwd");
    return PNPP_OK;
}

static DzJob make_dz_job(const AOperand *dz, int M, int C, float *out) {
    DzJob J;
    if (dz && out && (C & 3) == 0 && dz->lda == C && (dz->mode == A_DZ || dz->mode == A_DZ_POOL)) {
        J.dy = dz->a, J.z = dz->z, J.M = M, J.out = out;
        if (dz->mode == A_DZ_POOL) J.arg = dz->arg, J.K = dz->K;
    }
    return J;
}

int launch_bn_finalize_bwd(const double *slab, int nslab, int C, double count, int training, const float *gamma,
                           const float *mean, const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias,
                           hipStream_t st, const AOperand *dz, int M, float *dz_out, const double *count_dev, const double *local,
                           const PooledSource *pooled) {
    BnFinalizeBwdArgs F{slab, nslab, C, count, training, gamma, mean, istd, cst, dgamma, dbeta, dbias, count_dev, local};
    if (pooled) F.p_dout = pooled->dout, F.p_zsel = pooled->zsel, F.p_scale = pooled->scale, F.p_shift = pooled->shift, F.p_G = pooled->G;
    const DzJob J = make_dz_job(dz, M, C, dz_out);
    const int nfin = cdiv(C, FIN_COLS), ndz = dz_job_blocks(J, C);
    ProfScope ps(st, "bn_finalize_bwd_kernel C=%d%s%s", C, ndz ? " +dZ" : "", pooled ? " +pool" : "");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(nfin + ndz), dim3(256), 0, st, F, nfin, J);
    PNPP_CHECK_LAUNCH("bn_finalize_bwd");
    return PNPP_OK;
}

int launch_post_gemm(const double *slab, int nslab, int C, double count, int training, const float *gamma, const float *mean,
                     const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias, const float *dwslab, int nsplit,
                     int Nc, int kp_pad, int Kvalid, int perm_D, float *dw, int ldo, hipStream_t st, const AOperand *dz, int M,
                     float *dz_out, const double *count_dev, const double *local) {
    const BnFinalizeBwdArgs F{slab, nslab, C, count, training, gamma, mean, istd, cst, dgamma, dbeta, dbias, count_dev, local};
    const SlabReduceArgs R{dwslab, nsplit, Nc, kp_pad, Kvalid, perm_D, dw, ldo};
    const DzJob J = make_dz_job(dz, M, C, dz_out);
    const int total = Nc * Kvalid, nfin = cdiv(C, FIN_COLS), ndz = dz_job_blocks(J, C), nf = nfin + ndz;
    ProfScope ps(st, "post_gemm_kernel C=%d%s | N=%d K=%d split=%d", C, ndz ? " +dZ" : "", Nc, Kvalid, nsplit);
    if (nsplit == 0) {   // the weight gradient was written in place by its GEMM (one row range): nothing to reduce
        hipLaunchKernelGGL(post_gemm_kernel<64>, dim3(nf), dim3(256), 0, st, F, nfin, R, ndz, J);
    } else if (slab_reduce_vec4(R)) {
        const int groups = total / 4, epb = slab_reduce_epb4(nsplit);
        if (epb == 64) hipLaunchKernelGGL((post_gemm_kernel<64, true>), dim3(nf + cdiv(groups, 64)), dim3(256), 0, st, F, nfin, R, ndz, J);
        else if (epb == 16) hipLaunchKernelGGL((post_gemm_kernel<16, true>), dim3(nf + cdiv(groups, 16)), dim3(256), 0, st, F, nfin, R, ndz, J);
        else hipLaunchKernelGGL((post_gemm_kernel<4, true>), dim3(nf + cdiv(groups, 4)), dim3(256), 0, st, F, nfin, R, ndz, J);
    } else if (slab_reduce_wide(total, nsplit))
        hipLaunchKernelGGL(post_gemm_kernel<64>, dim3(nf + cdiv(total, 64)), dim3(256), 0, st, F, nfin, R, ndz, J);
    else
        hipLaunchKernelGGL(post_gemm_kernel<16>, dim3(nf + cdiv(total, 16)), dim3(256), 0, st, F, nfin, R, ndz, J);
    PNPP_CHECK_LAUNCH("post_gemm");
    return PNPP_OK;
}

// ---------------------------------------------------------------------------------------------
// max over the nsample axis with BatchNorm apply + ReLU folded in (pointnet_pp_8dir.py:41-42)
// first maximum wins ties (what torch.max does on the CPU)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pool_fwd_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int G, int K, int C,
                                                       float *__restrict__ out, int32_t *__restrict__ arg,
                                                       float *__restrict__ origin_a, float *__restrict__ origin_b, int norigin,
                                                       float *__restrict__ zsel) {
    if (blockIdx.x == 0)  // group_all levels: the centre of every cloud is the origin (pointnet_pp_8dir.py:24); no launch of its own
        for (int i = threadIdx.x; i < norigin; i += 256) {
            if (origin_a) origin_a[i] = 0.f;
            if (origin_b) origin_b[i] = 0.f;
        }
    const size_t total = (size_t)G * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t g = i / C;
        const int c = (int)(i - g * C);
        const float sc = scale[c], sh = shift[c];
        const float *p = z + g * K * C + c;
        float best = -INFINITY, zb = 0.f;
        int bi = 0;
        int k = 0;
        for (; k + 8 <= K; k += 8) {  // eight independent strided loads in flight per lane
            float z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) z[u] = p[(size_t)(k + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = fmaxf(fmaf(z[u], sc, sh), 0.f);
                if (v > best) best = v, bi = k + u, zb = z[u];
            }
        }
        for (; k < K; ++k) {
            const float zk = p[(size_t)k * C];
            const float v = fmaxf(fmaf(zk, sc, sh), 0.f);
            if (v > best) best = v, bi = k, zb = zk;
        }
        out[i] = best;
        arg[i] = bi;
        if (zsel) zsel[i] = zb;   // the pre-BN value the maximum came from: backward reads it instead of gathering z
    }
}

// The same reduction for a level that pools over whole clouds (group_all on raw points: K = N in the thousands, few
// groups): K is cut into gridDim.z chunks, a workgroup = 64 channels x 4 interleaved row lanes reduces one chunk to a
// (value, position) partial, and pool_fwd_merge_kernel takes the first maximum over the chunks in ascending order.
__global__ void __launch_bounds__(256) pool_fwd_split_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int K, int C, int chunk,
                                                             float *__restrict__ pmax, int32_t *__restrict__ parg) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int g = blockIdx.y, s = blockIdx.z;
    const int k0 = s * chunk, k1 = k0 + chunk < K ? k0 + chunk : K;
    float best = -INFINITY;
    int bi = k0;
    if (c < C) {
        const float sc = scale[c], sh = shift[c];
        const float *p = z + (size_t)g * K * C + c;
        int k = k0 + rl;
        for (; k + 12 < k1; k += 16) {  // four independent strided loads in flight per lane
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = p[(size_t)(k + 4 * u) * C];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = fmaxf(fmaf(t[u], sc, sh), 0.f);
                if (v > best) best = v, bi = k + 4 * u;
            }
        }
        for (; k < k1; k += 4) {
            const float v = fmaxf(fmaf(p[(size_t)k * C], sc, sh), 0.f);
            if (v > best) best = v, bi = k;
        }
    }
    sv[rl][cl] = best;
    si[rl][cl] = bi;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const float v = sv[r][cl];
            const int i = si[r][cl];
            if (v > best || (v == best && i < bi)) best = v, bi = i;
        }
        const size_t o = ((size_t)g * gridDim.z + s) * C + c;
        pmax[o] = best;
        parg[o] = bi;
    }
}

__global__ void __launch_bounds__(256) pool_fwd_merge_kernel(const float *__restrict__ pmax, const int32_t *__restrict__ parg, int G,
                                                             int nsplit, int C, float *__restrict__ out, int32_t *__restrict__ arg,
                                                             float *__restrict__ origin_a, float *__restrict__ origin_b, int norigin) {
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < norigin; i += 256) {
            if (origin_a) origin_a[i] = 0.f;
            if (origin_b) origin_b[i] = 0.f;
        }
    const size_t total = (size_t)G * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t g = i / C;
        const int c = (int)(i - g * C);
        float best = -INFINITY;
        int bi = 0;
        for (int s = 0; s < nsplit; ++s) {
            const size_t o = (g * nsplit + s) * C + c;
            const float v = pmax[o];
            if (v > best) best = v, bi = parg[o];
        }
        out[i] = best;
        arg[i] = bi;
    }
}

int pool_fwd_splits(int G, int K, int C) {
    if (K < 512) return 1;  // neighbourhood-sized groups: one thread per (group, channel)
    const long long blocks = (long long)cdiv(C, 64) * G;
    int nsplit = (int)cdiv(2048, blocks);          // >= 2048 workgroups over the chip ...
    const int most = K / 64;                        // ... of at least 64 rows each
    nsplit = nsplit > most ? most : nsplit;
    return nsplit < 1 ? 1 : nsplit;
}

int launch_pool_fwd(const float *z, const float *scale, const float *shift, int G, int K, int C, float *out, int32_t *arg,
                    hipStream_t st, float *origin_a, float *origin_b, int norigin, void *part, float *zsel) {
    const int nsplit = part ? pool_fwd_splits(G, K, C) : 1;
    if (nsplit > 1) {
        const int chunk = (cdiv(K, nsplit) + 3) & ~3;
        float *pmax = (float *)part;
        int32_t *parg = (int32_t *)(pmax + (size_t)G * nsplit * C);
        {
            ProfScope ps(st, "pool_fwd_split_kernel G=%d K=%d C=%d split=%d", G, K, C, nsplit);
            hipLaunchKernelGGL(pool_fwd_split_kernel, dim3(cdiv(C, 64), G, nsplit), dim3(256), 0, st, z, scale, shift, K, C, chunk,
                               pmax, parg);
            PNPP_CHECK_LAUNCH("pool_fwd_split");
        }
        const size_t tot = (size_t)G * C;
        ProfScope ps(st, "pool_fwd_merge_kernel G=%d C=%d split=%d", G, C, nsplit);
        hipLaunchKernelGGL(pool_fwd_merge_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, st, pmax, parg, G, nsplit, C, out, arg,
                           origin_a, origin_b, norigin);
        PNPP_CHECK_LAUNCH("pool_fwd_merge");
        return PNPP_OK;
    }
    const size_t total = (size_t)G * C;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ProfScope ps(st, "pool_fwd_kernel G=%d K=%d C=%d", G, K, C);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid), dim3(256), 0, st, z, scale, shift, G, K, C, out, arg, origin_a, origin_b, norigin,
                       zsel);
    PNPP_CHECK_LAUNCH("pool_fwd");
    return PNPP_OK;
}

// backward of max + ReLU: the dense gradient (zero except at the arg-max row when the pooled value is > 0)
// is NOT written; this kernel emits the masked pooled gradient dm (G x C) and the two BatchNorm-backward
// column sums, and the consumers rebuild dy from (dm, arg) on the fly.  block = 64 channels x 4 group lanes.
__global__ void __launch_bounds__(256)
pool_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ arg, const float *__restrict__ z,
                const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
                const float *__restrict__ istd, int G, int K, int C, float *__restrict__ dm, double *__restrict__ slab,
                const float *__restrict__ zsel) {
    __shared__ double red[4][2][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float mu = mean[c], is = istd[c], sc = scale[c], sh = shift[c];
        for (int g = blockIdx.y * 4 + gl; g < G; g += gridDim.y * 4) {
            const size_t gi = (size_t)g * C + c;
            // the pre-BN value behind the pooled output: kept by the forward pass (zsel, a coalesced stream), or gathered --
            // one 4-byte element per (group, channel) out of a row of Z, a 64-byte line each
            const float za = zsel ? zsel[gi] : z[((size_t)g * K + arg[gi]) * C + c];
            const float d = fmaf(za, sc, sh) > 0.f ? dout[gi] : 0.f;  // ReLU'(pooled value), same expression as forward
            dm[gi] = d;
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
                    const float *mean, const float *istd, int G, int K, int C, float *dm, double *slab, int *nslab,
                    hipStream_t st, const float *zsel) {
    int gy = cdiv(G, 16);  // four groups per lane-row and pass
    if (gy > kMaxStatBlocks) gy = kMaxStatBlocks;
    if (gy < 1) gy = 1;
    *nslab = gy;
    ProfScope ps(st, "pool_bwd_kernel G=%d K=%d C=%d", G, K, C);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(cdiv(C, 64), gy), dim3(256), 0, st, dout, arg, z, scale, shift, mean, istd, G, K, C,
                       dm, slab, zsel);
    PNPP_CHECK_LAUNCH("pool_bwd");
    return PNPP_OK;
}

__global__ void __launch_bounds__(256) fill_zero_kernel(float4 *__restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

#ifdef PNPP_STAMPS
}  // namespace pnpp
extern "C" int pnpp_debug_stamps(unsigned long long *out16, int kd) {  // kd > 0: select + reset; kd == 0: read
    if (kd > 0) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_stamps), z, sizeof(z));
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_stamp_kd), &kd, sizeof(int));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out16, HIP_SYMBOL(pnpp::g_stamps), 16 * sizeof(unsigned long long));
    }
    return 0;
}
namespace pnpp {
#endif
int launch_fill_zero(void *p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return PNPP_OK;
    hipError_t e = hipMemsetAsync(p, 0, bytes, st);
    if (e != hipSuccess) {
        set_error("memset failed: %s", hipGetErrorString(e));
        return PNPP_ERR_LAUNCH;
    }
    return PNPP_OK;
}

#ifdef PNPP_STAMPS
#define PNPP_STAMPS_BIT 64u
#else
#define PNPP_STAMPS_BIT 0u
#endif
unsigned gemm_build_flags() { return ((PNPP_WS_EXP_NO_MFMA != 0) ? 1u : 0u) | PNPP_STAMPS_BIT; }

}  // namespace pnpp
