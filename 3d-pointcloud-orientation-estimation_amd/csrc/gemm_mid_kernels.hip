// gemm_mid_kernels.hip -- the group_all level's wide layers (M = 32 * batch rows, K and N in the hundreds): 64 x 64 output tiles,
// one per workgroup, 64-deep reduction chunks double-buffered through LDS.
//
// Reference: models/pointnet_pp_8dir.py:23-26,40-42 (the group_all PointNetSetAbstraction: 1 x 1 conv -> BatchNorm -> ReLU on
// B x 32 rows) and its autograd backward (dA = dZ W, dW = dZ^T a).
//
// Why a third GEMM form.  The weights-stationary kernel needs tens of thousands of rows to amortise its panel; the 32 x 32 split-K
// kernel (gemm_smallm_kernel) re-reads both operands from L2 once per 32 x 32 tile -- 131 MB of L2 reads for the 512 -> 1024 layer
// of a 1,024-row level, 0.33 of the float32 MFMA roof -- and the 128 x 128 dW blocks it was paired with went out as sixteen 64-row
// splits whose partial sums were 7.5 x the launch's compulsory HBM traffic.  Here a workgroup owns a 64 x 64 tile over the WHOLE
// reduction (no partial sums at all for dA and the forward product, at most two row ranges for dW), a tile's operands are read
// from L2 once per 64 x 64 outputs, and the loop is  [barrier | LDS image of chunk c+1 from registers | loads of chunk c+2 |
// MFMAs of chunk c] with ONE barrier per chunk.  (A 64 x 32-tile form -- 512 workgroups, two or three per CU, the four waves =
// (row half) x (reduction half of a chunk) combined through LDS -- measured the same 26 / 15 us as these 256 workgroups, one per
// CU: DESIGN.md section 9.)
//
// Operand images.  Every operand tile is 64 rows x 64 contiguous floats of a row-major matrix, fetched as four 16-byte loads per
// thread (thread -> column group tid % 16, rows tid / 16 + 16 i).  What differs is how an MFMA consumes it:
//   row operand   (MFMA index = tile row, reduction = tile column): pitch 68 floats, lane (l31, lh) reads k = 8 t + 4 lh + {0..3}
//                 of its row with one ds_read_b128 per four MFMA steps (eight lanes of a read cycle land on eight different
//                 4-bank groups: 68 r mod 32 = 4 r);
//   column operand (MFMA index = tile column, reduction = tile row): pitch 64, one conflict-free ds_read_b32 per MFMA step
//                 (32 consecutive floats per half-wave).
// forward  Z = a W^T : a row operand, W [n][k] row operand.     dA = dZ W : dZ row operand, W [k][n] column operand.
// dW = dZ^T a        : both column operands (reduction over rows).
// Both operands of an MFMA step must carry the same reduction index in the same lane half; k(t, lh, u) = 8 t + 4 lh + u for all.
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

static inline bool mid_ptr_ok_ext(const float *p, int ld) { return p && (ld & 3) == 0 && ((uintptr_t)p & 15) == 0; }

constexpr int MID_T = 64;            // tile edge and chunk depth
constexpr int MID_RP = 68;           // pitch of a row-operand image
constexpr int MID_IMG = MID_T * MID_RP;          // floats per operand image (column operands use the first 64 * 64)
constexpr int MID_STAGE = 2 * MID_IMG;           // A image + B image
constexpr size_t MID_LDS_BYTES = (size_t)2 * MID_STAGE * sizeof(float);   // two stages: 69,632 bytes

struct MidGemm {
    const float *a;      // [M][K] activations (row operand)
    int lda;
    const float *scale, *shift;   // A_BNRELU: per reduction index
    const float *b;      // weights: BT ? [N][K] : [K][N]
    int ldb;
    int M, N, K;
    Epilogue E;
};

struct MidDw {
    const float *dz;     // [M][Nc]
    int ldz;
    const float *a2;     // [M][Kp]
    int lda2;
    const float *scale, *shift;   // A_BNRELU on a2: per column of a2
    int M, Nc, Kp;
    int nsplit, rps;     // row ranges (multiples of 64)
    float *out;          // [nsplit][Nc][ldo]
    int ldo;
};

// Operand streams are BUFFER loads: (resource descriptor in SGPRs = the tile's uniform base) + (one 32-bit lane offset, computed
// once) + (the chunk's byte offset in an SGPR).  With global_load and per-load 64-bit addresses the register allocator recycled
// address registers that were still the DESTINATION of loads in flight -- an s_waitcnt vmcnt in the middle of every batch, one
// exposed L2 round trip per chunk -- and loop strength reduction defeats a hand-made (uniform pointer + lane offset) form.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mid_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), (short)0, 0x7ffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 mid_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned chunk_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)chunk_off, 0));
}

// ---- C tile (64 x 64) = a'[64 x K] * W, one workgroup --------------------------------------------------------------------
template <int AX, bool BT, int EM>
__device__ __forceinline__ void mid_gemm_tile(const MidGemm &G, const int tm, const int tn, float *__restrict__ lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int q4 = 4 * (tid & 15), rb = tid >> 4;
    const int m0 = tm * MID_T, n0 = tn * MID_T, nc = G.K / MID_T;
    const Epilogue &E = G.E;

    // (operand registers are ext-vector values, not float4 structs: a struct that is only copied global -> register -> LDS compiles
    //  to memcpy through PRIVATE memory -- scratch_store / scratch_load per chunk, which count in vmcnt and serialise on the
    //  global loads issued in front of them: 2.2 us per chunk for 1.0 us of MFMAs in the first version of this kernel)
    f32x4 ra[4], rw[4];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    // operand streams = (uniform base, advanced per chunk by scalar arithmetic) + (four 32-bit lane offsets computed once): the
    // loads take the SGPR-base form and need no 64-bit address registers.  (With per-load 64-bit addresses the allocator recycled
    // address registers that were still the DESTINATION of loads in flight: an s_waitcnt vmcnt in the middle of every batch, i.e.
    // one exposed L2 round trip per chunk.)
    const __amdgpu_buffer_rsrc_t resA = mid_rsrc(G.a + (size_t)m0 * G.lda);
    const __amdgpu_buffer_rsrc_t resW = mid_rsrc(BT ? G.b + (size_t)n0 * G.ldb : G.b + n0);
    unsigned oa[4], ow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oa[i] = 4u * ((unsigned)(rb + 16 * i) * (unsigned)G.lda + (unsigned)q4);
        ow[i] = 4u * ((unsigned)(rb + 16 * i) * (unsigned)G.ldb + (unsigned)q4);
    }
    const unsigned stepA = MID_T * (unsigned)sizeof(float), stepW = BT ? MID_T * (unsigned)sizeof(float) : MID_T * (unsigned)G.ldb * (unsigned)sizeof(float);
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = mid_load4(resA, oa[i], (unsigned)c * stepA);
            rw[i] = mid_load4(resW, ow[i], (unsigned)c * stepW);
        }
        if constexpr (AX == A_BNRELU) {
            sc = *reinterpret_cast<const float4 *>(G.scale + c * MID_T + q4);
            sh = *reinterpret_cast<const float4 *>(G.shift + c * MID_T + q4);
        }
    };
    auto lstore = [&](int stage) {
        float *As = lds + stage * MID_STAGE, *Bs = As + MID_IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = ra[i];
            if constexpr (AX == A_BNRELU) {
                v[0] = fmaxf(fmaf(v[0], sc.x, sh.x), 0.f), v[1] = fmaxf(fmaf(v[1], sc.y, sh.y), 0.f);
                v[2] = fmaxf(fmaf(v[2], sc.z, sh.z), 0.f), v[3] = fmaxf(fmaf(v[3], sc.w, sh.w), 0.f);
            }
            *reinterpret_cast<f32x4 *>(As + (rb + 16 * i) * MID_RP + q4) = v;
            *reinterpret_cast<f32x4 *>(Bs + (rb + 16 * i) * (BT ? MID_RP : MID_T) + q4) = rw[i];
        }
    };

    // E_MASK_STATS reads the previous layer's z at the output positions: requested while the last chunk is being multiplied
    float zp[16];
    float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f;
    const int col = n0 + wn * 32 + l31;
    if constexpr (EM == E_MASK_STATS) e_sc = E.scale[col], e_sh = E.shift[col], e_mu = E.mu[col], e_is = E.istd[col];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // Per chunk: barrier | LDS image of chunk c+1 from the registers | global loads of chunk c+2 into them | MFMAs of chunk c.
    // __syncthreads() is a workgroup-scope fence: it waits for EVERY outstanding global load (s_waitcnt vmcnt(0)) before the barrier,
    // so loads issued in front of it are never in flight during the MFMAs.  They are issued right BEHIND it instead and have the
    // whole matrix phase to land (first version, loads in front of the barrier: 2.2 us per chunk for 1.0 us of MFMAs).
    gload(0);
    lstore(0);
    if (nc > 1) gload(1);
    for (int c = 0; c < nc; ++c) {
        __syncthreads();   // stage c & 1 is complete; every wave has finished reading stage (c + 1) & 1
        if (c + 1 < nc) lstore((c + 1) & 1);   // the registers hold chunk c + 1 (requested one iteration ago)
        if (c + 2 < nc) gload(c + 2);
        const float *As = lds + (c & 1) * MID_STAGE, *Bs = As + MID_IMG;
        if constexpr (EM == E_MASK_STATS) {
            if (c == nc - 1) {
                const float *pz = E.zp + (size_t)(m0 + wm * 32 + 4 * lh) * E.ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[r] = pz[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc];
            }
        }
        const float *arow = As + (wm * 32 + l31) * MID_RP + 4 * lh;
        const float *brow = BT ? Bs + (wn * 32 + l31) * MID_RP + 4 * lh : Bs + (4 * lh) * MID_T + wn * 32 + l31;
        float4 fa[2], fb[2];
        auto ld = [&](int buf, int t) {
            fa[buf] = *reinterpret_cast<const float4 *>(arow + 8 * t);
            if constexpr (BT) {
                fb[buf] = *reinterpret_cast<const float4 *>(brow + 8 * t);
            } else {
                const float *p = brow + 8 * t * MID_T;
                fb[buf] = make_float4(p[0], p[MID_T], p[2 * MID_T], p[3 * MID_T]);
            }
        };
        auto mm = [&](int buf) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf].w, acc, 0, 0, 0);
        };
        ld(0, 0);
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            ld(1, t + 1);
            mm(0);
            if (t + 2 < 8) ld(0, t + 2);
            mm(1);
        }
    }

    // epilogue: a lane holds 16 rows of one column; a half-wave store is 32 consecutive floats
    if constexpr (EM == E_STORE_STATS) {
        if (E.pool_ext) {   // (uniform) 32-row neighbourhoods: this wave's 32 x 32 tile is one of them -- its extreme row per column
            const float sg = (E.pool_gamma ? E.pool_gamma[col] : 1.f) >= 0.f ? 1.f : -1.f;   // (see Epilogue::pool_ext, kernels.h)
            float mx = sg * acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sg * acc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            int a = 64;
#pragma unroll
            for (int r = 15; r >= 0; --r) a = (sg * acc[r] == mx) ? (r & 3) + 8 * (r >> 2) + 4 * lh : a;
            a = min(a, __shfl_xor(a, 32, 64));
            if (lh == 0) {
                const size_t gi = (size_t)((m0 + wm * 32) >> 5) * E.ldc + col;
                E.pool_ext[gi] = sg * mx;
                E.pool_arg[gi] = a;
            }
        }
    }
    float *tb = E.c + (size_t)(m0 + wm * 32 + 4 * lh) * E.ldc + col;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if constexpr (EM == E_STORE_STATS) {
            t1 += v;
            t2 = fmaf(v, v, t2);
        } else if constexpr (EM == E_MASK_STATS) {
            const float z0 = zp[r];
            v = fmaf(z0, e_sc, e_sh) > 0.f ? v : 0.f;
            t1 += v;
            t2 = fmaf(v, (z0 - e_mu) * e_is, t2);
        }
        tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc] = v;
    }
    if constexpr (EM != E_STORE) {
        // column sums of the tile: the two row halves of a wave, then the two waves of a column, in fixed order
        __syncthreads();                                 // every operand read of the last chunk is done
        double *red = reinterpret_cast<double *>(lds);   // [2 wm][2][64]
        const double a = (double)t1 + shfl_xor_f64((double)t1, 32), b = (double)t2 + shfl_xor_f64((double)t2, 32);
        if (lh == 0) red[(wm * 2 + 0) * MID_T + wn * 32 + l31] = a, red[(wm * 2 + 1) * MID_T + wn * 32 + l31] = b;
        __syncthreads();
        if (tid < 2 * MID_T) {
            const int which = tid >> 6, cl = tid & 63;
            E.slab[((size_t)tm * 2 + which) * G.N + n0 + cl] = red[(0 * 2 + which) * MID_T + cl] + red[(1 * 2 + which) * MID_T + cl];
        }
    }
}

// ---- dW tile (64 x 64) = dZ^T a2 over one row range, one workgroup -------------------------------------------------------
template <int A2X>
__device__ __forceinline__ void mid_dw_tile(const MidDw &D, const int tc, const int tk, const int split, float *__restrict__ lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, wc = wave >> 1, wk = wave & 1;
    const int q4 = 4 * (tid & 15), rb = tid >> 4;
    const int c0 = tc * MID_T, k0 = tk * MID_T;
    const int r0 = split * D.rps, r1 = min(D.M, r0 + D.rps), nc = (r1 - r0) / MID_T;

    f32x4 rd[4], ra[4];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (A2X == A_BNRELU) {   // this thread's four a2 columns never change
        sc = *reinterpret_cast<const float4 *>(D.scale + k0 + q4);
        sh = *reinterpret_cast<const float4 *>(D.shift + k0 + q4);
    }
    const __amdgpu_buffer_rsrc_t resD = mid_rsrc(D.dz + (size_t)r0 * D.ldz + c0);
    const __amdgpu_buffer_rsrc_t resA = mid_rsrc(D.a2 + (size_t)r0 * D.lda2 + k0);
    unsigned od[4], oa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        od[i] = 4u * ((unsigned)(rb + 16 * i) * (unsigned)D.ldz + (unsigned)q4);
        oa[i] = 4u * ((unsigned)(rb + 16 * i) * (unsigned)D.lda2 + (unsigned)q4);
    }
    const unsigned stepD = MID_T * (unsigned)D.ldz * (unsigned)sizeof(float), stepA = MID_T * (unsigned)D.lda2 * (unsigned)sizeof(float);
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rd[i] = mid_load4(resD, od[i], (unsigned)c * stepD);
            ra[i] = mid_load4(resA, oa[i], (unsigned)c * stepA);
        }
    };
    auto lstore = [&](int stage) {
        float *Ds = lds + stage * MID_STAGE, *As = Ds + MID_IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = ra[i];
            if constexpr (A2X == A_BNRELU) {
                v[0] = fmaxf(fmaf(v[0], sc.x, sh.x), 0.f), v[1] = fmaxf(fmaf(v[1], sc.y, sh.y), 0.f);
                v[2] = fmaxf(fmaf(v[2], sc.z, sh.z), 0.f), v[3] = fmaxf(fmaf(v[3], sc.w, sh.w), 0.f);
            }
            *reinterpret_cast<f32x4 *>(Ds + (rb + 16 * i) * MID_T + q4) = rd[i];
            *reinterpret_cast<f32x4 *>(As + (rb + 16 * i) * MID_T + q4) = v;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (nc > 0) {
        gload(0);
        lstore(0);
        if (nc > 1) gload(1);
    }
    for (int c = 0; c < nc; ++c) {
        __syncthreads();   // (same order as mid_gemm_tile: the loads follow the barrier and fly during the MFMAs)
        if (c + 1 < nc) lstore((c + 1) & 1);
        if (c + 2 < nc) gload(c + 2);
        const float *Ds = lds + (c & 1) * MID_STAGE, *As = Ds + MID_IMG;
        const float *dcol = Ds + lh * MID_T + wc * 32 + l31, *acol = As + lh * MID_T + wk * 32 + l31;
        float fd[2][4], fa[2][4];
        auto ld = [&](int buf, int s4) {   // four reduction steps: rows 2 (4 s4 + u) + lh
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                fd[buf][u] = dcol[(8 * s4 + 2 * u) * MID_T];
                fa[buf][u] = acol[(8 * s4 + 2 * u) * MID_T];
            }
        };
        auto mm = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[buf][u], fa[buf][u], acc, 0, 0, 0);
        };
        ld(0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 8; s4 += 2) {
            ld(1, s4 + 1);
            mm(0);
            if (s4 + 2 < 8) ld(0, s4 + 2);
            mm(1);
        }
    }
    float *o = D.out + ((size_t)split * D.Nc + c0 + wc * 32 + 4 * lh) * D.ldo + k0 + wk * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * D.ldo] = acc[r];
}

// XCD-aware tile map (speed only).  Workgroup b runs on XCD b % 8, each with its own L2: a column of tiles per XCD makes every XCD
// fetch ALL rows of the row operand (4 MB of dZ eight times over).  Dealing the R x C tile grid out as 8 rectangles of (R/4) x (C/2)
// tiles instead gives an XCD R/4 row blocks and C/2 column blocks of operands: the launch's L2 misses drop from ~8 x to ~3 x the
// compulsory bytes.  b -> (row tile, column tile); falls back to row-major when the grid does not divide.
__device__ __forceinline__ void mid_tile_map(int b, int R, int C, int &tr, int &tc) {
    if ((R & 3) == 0 && (C & 1) == 0) {
        const int xcd = b & 7, i = b >> 3, bw = C >> 1;   // rectangle (xcd / 2, xcd % 2), i-th tile inside it (R/4 x C/2, row-major)
        tr = (xcd >> 1) * (R >> 2) + i / bw;
        tc = (xcd & 1) * bw + i % bw;
    } else {
        tr = b / C, tc = b % C;
    }
}

// ---- kernels ---------------------------------------------------------------------------------------------------------------
template <int AX, bool BT, int EM>
__global__ void __launch_bounds__(256) gemm_mid_kernel(const MidGemm G, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tm, tn;
    mid_tile_map(blockIdx.x, G.M / MID_T, tiles_n, tm, tn);
    mid_gemm_tile<AX, BT, EM>(G, tm, tn, lds);
}

// dA (+ ReLU mask and BatchNorm-backward sums) and dW of one backward layer only share dZ: one launch, the first g1 workgroups
// take the dA tiles, the rest the dW tiles
template <int EM, int A2X>
__global__ void __launch_bounds__(256) da_dw_mid_kernel(const MidGemm G, int tiles_n, int g1, const MidDw D, int tiles_c, int tiles_k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < g1) {
        int tm, tn;
        mid_tile_map(blockIdx.x, G.M / MID_T, tiles_n, tm, tn);
        mid_gemm_tile<A_PLAIN, false, EM>(G, tm, tn, lds);
    } else {
        const int b = blockIdx.x - g1, per = tiles_c * tiles_k;
        int tc, tk;
        if ((g1 & 7) == 0 && (per & 7) == 0) mid_tile_map(b % per, tiles_c, tiles_k, tc, tk);   // (b % per) % 8 is still this block's XCD
        else tc = (b % per) / tiles_k, tk = (b % per) % tiles_k;
        mid_dw_tile<A2X>(D, tc, tk, b / per, lds);
    }
}

// ---- the forward product with split products (gemm_wsf3_kernels.hip has the arithmetic): 64 x 64 output tile, eight waves ---------
// Waves 0-3 multiply (one 32 x 32 accumulator each, the leading products and the small ones apart), waves 4-7 -- one per SIMD beside
// them -- stage: they stream the next 64-deep chunks of BOTH operands (a' rows and W rows: the weights are not stationary here), apply
// BatchNorm + ReLU to a', split every element into its three bf16 pieces and write the chunk images ([64 rows][64 k], three planes,
// 128-byte rows, 16-byte groups at g ^ x(r) as in gemm_wsd3_kernels.hip) of the OTHER stage while the multipliers work on this one.  The
// vector work of the split (about as many cycles as the 24 bf16 MFMAs of a chunk) overlaps with the matrix work because it comes from
// another wave; one s_barrier per chunk (no workgroup fence: the stagers' loads stay in flight across it).
// In-kernel phase stamps (-DMID3_STAMPS, tools/mid3_stamps.py): s_memtime in one stager and one multiplier wave of workgroup 8.
#ifdef MID3_STAMPS
__device__ unsigned long long g_mid3_stamps[2][4];   // [stager, multiplier][prologue, work, barrier, epilogue]
#define M3_STAMP(i)                                                    \
    if (st_on) {                                                       \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += st_t - st_last;                                   \
        st_last = st_t;                                                \
    }
#else
#define M3_STAMP(i)
#endif
constexpr int MID3_PLANE = MID_T * 128, MID3_IMG = 3 * MID3_PLANE, MID3_STAGE = 2 * MID3_IMG;   // bytes: plane, image (A or W), stage
constexpr size_t MID3_LDS_BYTES = (size_t)2 * MID3_STAGE;                                    // two stages: 98,304 bytes

typedef __bf16 mid3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mid3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mid3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mid3_pk(float lo, float hi) {
    const mid3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mid3_bf16x2));
}
__device__ __forceinline__ void mid3_split4(const f32x4 v, uint2 &h, uint2 &m, uint2 &l) {
    h.x = mid3_pk(v[0], v[1]), h.y = mid3_pk(v[2], v[3]);
    float r0 = v[0] - __uint_as_float(h.x << 16), r1 = v[1] - __uint_as_float(h.x & 0xffff0000u);
    float r2 = v[2] - __uint_as_float(h.y << 16), r3 = v[3] - __uint_as_float(h.y & 0xffff0000u);
    m.x = mid3_pk(r0, r1), m.y = mid3_pk(r2, r3);
    r0 -= __uint_as_float(m.x << 16), r1 -= __uint_as_float(m.x & 0xffff0000u), r2 -= __uint_as_float(m.y << 16), r3 -= __uint_as_float(m.y & 0xffff0000u);
    l.x = mid3_pk(r0, r1), l.y = mid3_pk(r2, r3);
}
__device__ __forceinline__ mid3_bf16x8 mid3_op(uint4 v) { return __builtin_bit_cast(mid3_bf16x8, v); }
__device__ __forceinline__ void mid3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS only: no vmcnt wait

template <int AX, int EM>
__global__ void __launch_bounds__(512, 1) gemm_mid3_kernel(const MidGemm G, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    int tm, tn;
    mid_tile_map(blockIdx.x, G.M / MID_T, tiles_n, tm, tn);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = tm * MID_T, n0 = tn * MID_T, nc = G.K / MID_T;
    const Epilogue &E = G.E;
    auto xs = [](int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); };   // group XOR of image row r (bits 1-3 of r)
    const bool stager = wave >= 4;
    const int wm = (wave >> 1) & 1, wn = wave & 1;   // multipliers: the wave's 32 x 32 tile
    f32x16 acc, accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f, accs[r] = 0.f;

#ifdef MID3_STAMPS
    const bool st_on = blockIdx.x == 8 && (wave == 0 || wave == 4);
    unsigned long long st_acc[4] = {0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    if (stager) {
        const int t2 = tid - 256, q = t2 & 15, q4 = 4 * q, rb = t2 >> 4;   // k group 4 q .. + 3 of the chunk, rows rb + 16 i
        const __amdgpu_buffer_rsrc_t resA = mid_rsrc(G.a + (size_t)m0 * G.lda), resW = mid_rsrc(G.b + (size_t)n0 * G.ldb);
        const unsigned oa = 4u * ((unsigned)rb * (unsigned)G.lda + (unsigned)q4), ow = 4u * ((unsigned)rb * (unsigned)G.ldb + (unsigned)q4);
        const unsigned wofs = (unsigned)(rb * 128 + 16 * ((q >> 1) ^ xs(rb)) + 8 * (q & 1));   // row rb + 16 i: + 2048 i (x ignores bit 4)
        // four register sets, by chunk index mod 4, named apart (a set indexed by c & 3 would live in private memory): a chunk is requested
        // four chunk times before it is staged -- with two sets the loop ran at the L2 round trip per chunk (13.8 us for 8 chunks)
        f32x4 ra0[4], rw0[4], ra1[4], rw1[4], ra2[4], rw2[4], ra3[4], rw3[4];
        f32x4 sc0, sh0, sc1, sh1, sc2, sh2, sc3, sh3;
        auto gload = [&](int c, f32x4 (&ra)[4], f32x4 (&rw)[4], f32x4 &sc, f32x4 &sh) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = mid_load4(resA, oa, (unsigned)c * 256u + (unsigned)i * (64u * (unsigned)G.lda));
                rw[i] = mid_load4(resW, ow, (unsigned)c * 256u + (unsigned)i * (64u * (unsigned)G.ldb));
            }
            if constexpr (AX == A_BNRELU) {
                sc = *reinterpret_cast<const f32x4 *>(G.scale + c * MID_T + q4);
                sh = *reinterpret_cast<const f32x4 *>(G.shift + c * MID_T + q4);
            }
        };
        auto lstore = [&](int stage, const f32x4 (&ra)[4], const f32x4 (&rw)[4], const f32x4 &sc, const f32x4 &sh) {
            unsigned char *As = lds3 + stage * MID3_STAGE, *Ws = As + MID3_IMG;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = ra[i];
                if constexpr (AX == A_BNRELU) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = fmaxf(fmaf(v[u], sc[u], sh[u]), 0.f);
                }
                uint2 h, m, l;
                mid3_split4(v, h, m, l);
                unsigned char *d = As + wofs + i * 2048;
                *reinterpret_cast<uint2 *>(d) = h, *reinterpret_cast<uint2 *>(d + MID3_PLANE) = m, *reinterpret_cast<uint2 *>(d + 2 * MID3_PLANE) = l;
                mid3_split4(rw[i], h, m, l);
                d = Ws + wofs + i * 2048;
                *reinterpret_cast<uint2 *>(d) = h, *reinterpret_cast<uint2 *>(d + MID3_PLANE) = m, *reinterpret_cast<uint2 *>(d + 2 * MID3_PLANE) = l;
            }
        };
        gload(0, ra0, rw0, sc0, sh0);
        gload(1, ra1, rw1, sc1, sh1);
        gload(2, ra2, rw2, sc2, sh2);
        gload(3, ra3, rw3, sc3, sh3);
        lstore(0, ra0, rw0, sc0, sh0);
        gload(4, ra0, rw0, sc0, sh0);   // K / 64 is a multiple of 4 and at least 8
        M3_STAMP(0)
        mid3_barrier();   // stage 0 is complete
        M3_STAMP(2)
        // Four chunks per pass.  The multipliers are on chunk c (stage 0): the other stage, which they left at the previous barrier, takes
        // chunk c + 1, and so on.  The last two passes are written out apart so that no request sits behind a branch: the compiler counts
        // outstanding loads per path, and a conditional request in the loop made every wait behind it a wait for ALL loads.
        auto pass = [&](const int c, const bool more, const bool more8) __attribute__((always_inline)) {
            lstore(1, ra1, rw1, sc1, sh1);
            if (more) gload(c + 5, ra1, rw1, sc1, sh1);
            M3_STAMP(1)
            mid3_barrier();
            M3_STAMP(2)
            lstore(0, ra2, rw2, sc2, sh2);
            if (more) gload(c + 6, ra2, rw2, sc2, sh2);
            M3_STAMP(1)
            mid3_barrier();
            M3_STAMP(2)
            lstore(1, ra3, rw3, sc3, sh3);
            if (more) gload(c + 7, ra3, rw3, sc3, sh3);
            M3_STAMP(1)
            mid3_barrier();
            M3_STAMP(2)
            if (more) lstore(0, ra0, rw0, sc0, sh0);
            if (more8) gload(c + 8, ra0, rw0, sc0, sh0);
            M3_STAMP(1)
            mid3_barrier();
            M3_STAMP(2)
        };
        for (int c = 0; c + 8 < nc; c += 4) pass(c, true, true);
        pass(nc - 8, true, false);
        pass(nc - 4, false, false);
    } else {
        const unsigned arow = (unsigned)((wm * 32 + l31) * 128), brow = (unsigned)((wn * 32 + l31) * 128);
        const int ax = xs(l31);   // rows 32 wm + l31 and 32 wn + l31: x ignores bits 4, 5
        M3_STAMP(0)
        mid3_barrier();   // stage 0 is complete
        M3_STAMP(2)
        for (int c = 0; c < nc; ++c) {
            const unsigned char *As = lds3 + (c & 1) * MID3_STAGE, *Ws = As + MID3_IMG;
            uint4 fa[2][3], fb[2][3];
            auto ld = [&](int buf, int t) {
                const int g = (2 * t + lh) ^ ax;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[buf][p] = *reinterpret_cast<const uint4 *>(As + p * MID3_PLANE + arow + 16 * g);
                    fb[buf][p] = *reinterpret_cast<const uint4 *>(Ws + p * MID3_PLANE + brow + 16 * g);
                }
            };
            auto mm = [&](int buf) {
                const mid3_bf16x8 ah = mid3_op(fa[buf][0]), am = mid3_op(fa[buf][1]), al = mid3_op(fa[buf][2]);
                const mid3_bf16x8 bh = mid3_op(fb[buf][0]), bm = mid3_op(fb[buf][1]), bl = mid3_op(fb[buf][2]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accs, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accs, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, accs, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, accs, 0, 0, 0);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, accs, 0, 0, 0);
            };
            ld(0, 0);
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                ld(1, t + 1);
                mm(0);
                if (t + 2 < 4) ld(0, t + 2);
                mm(1);
            }
#ifdef MID3_STAMPS
            asm volatile("s_nop 0" ::"v"(acc), "v"(accs));   // the clock is read behind the last MFMA, not behind its issue
#endif
            M3_STAMP(1)
            mid3_barrier();
            M3_STAMP(2)
        }
    }

    // ---- epilogue (multiplier waves; the column sums need two more workgroup barriers, which every wave takes) ----
    const int col = n0 + wn * 32 + l31;
    float t1 = 0.f, t2 = 0.f;
    if (!stager) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += accs[r];
        if constexpr (EM == E_STORE_STATS) {
            if (E.pool_ext) {   // (uniform) 32-row neighbourhoods: this wave's 32 x 32 tile is one of them -- its extreme row per column
                const float sg = (E.pool_gamma ? E.pool_gamma[col] : 1.f) >= 0.f ? 1.f : -1.f;
                float mx = sg * acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sg * acc[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                int a = 64;
#pragma unroll
                for (int r = 15; r >= 0; --r) a = (sg * acc[r] == mx) ? (r & 3) + 8 * (r >> 2) + 4 * lh : a;
                a = min(a, __shfl_xor(a, 32, 64));
                if (lh == 0) {
                    const size_t gi = (size_t)((m0 + wm * 32) >> 5) * E.ldc + col;
                    E.pool_ext[gi] = sg * mx;
                    E.pool_arg[gi] = a;
                }
            }
        }
        float *tb = E.c + (size_t)(m0 + wm * 32 + 4 * lh) * E.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r];
            t1 += v;
            t2 = fmaf(v, v, t2);
            tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc] = v;
        }
    }
    if constexpr (EM == E_STORE_STATS) {
        __syncthreads();                                  // every operand read of the last chunk is done
        double *red = reinterpret_cast<double *>(lds3);   // [2 wm][2][64]
        if (!stager) {
            const double a = (double)t1 + shfl_xor_f64((double)t1, 32), b = (double)t2 + shfl_xor_f64((double)t2, 32);
            if (lh == 0) red[(wm * 2 + 0) * MID_T + wn * 32 + l31] = a, red[(wm * 2 + 1) * MID_T + wn * 32 + l31] = b;
        }
        __syncthreads();
        if (tid < 2 * MID_T) {
            const int which = tid >> 6, cl = tid & 63;
            E.slab[((size_t)tm * 2 + which) * G.N + n0 + cl] = red[(0 * 2 + which) * MID_T + cl] + red[(1 * 2 + which) * MID_T + cl];
        }
    }
#ifdef MID3_STAMPS
    M3_STAMP(3)
    if (st_on && lane == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) g_mid3_stamps[stager ? 0 : 1][i] += st_acc[i];
#endif
}

static bool mid_ptr_ok(const float *p, int ld) { return p && (ld & 3) == 0 && ((uintptr_t)p & 15) == 0; }

template <typename K>
static void mid_grant_lds(K kfn) {
    static bool done = false;   // per instantiation
    if (!done) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MID_LDS_BYTES);
        done = true;
    }
}

static bool mid_gemm_shape_ok(int M, int Nout, int Kd) {
    if (M < 512 || M > 8192 || M % MID_T || Nout % MID_T || Kd % MID_T) return false;
    if ((M / MID_T) * (Nout / MID_T) < 192) return false;   // fewer tiles than compute units: many small workgroups win
    return M / MID_T <= kMaxStatBlocks;
}
// true when try_launch_mid_gemm takes this forward product of a level's last layer (so its epilogue can take the max-pool)
bool mid_gemm_pools(const AOperand &A, int M, int Nout, int Kd) {
    return mid_gemm_shape_ok(M, Nout, Kd) && (A.mode == A_PLAIN || A.mode == A_BNRELU) && mid_ptr_ok_ext(A.a, A.lda);
}

// forward product of a group_all layer: Z = relu(bn(z_prev)) W^T (+ column statistics).  false: the shape stays on the 32 x 32 kernel.
bool try_launch_mid_gemm(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st,
                         int *rc) {
    *rc = PNPP_OK;
    if (!mid_gemm_shape_ok(M, Nout, Kd)) return false;
    if (!(A.mode == A_PLAIN || A.mode == A_BNRELU) || !B.trans || B.perm_D >= 0 || B.rows != Kd) return false;
    if (!(E.mode == E_STORE || E.mode == E_STORE_STATS)) return false;
    if (!mid_ptr_ok(A.a, A.lda) || !mid_ptr_ok(B.b, B.ldb) || !mid_ptr_ok(E.c, E.ldc)) return false;
    MidGemm G{A.a, A.lda, A.scale, A.shift, B.b, B.ldb, M, Nout, Kd, E};
    if (nslab) *nslab = M / MID_T;
    const int tn = Nout / MID_T, grid = (M / MID_T) * tn;
    static const bool mid3_on = !(getenv("PNPP_MID3") && atoi(getenv("PNPP_MID3")) == 0);   // PNPP_MID3=0: the float32-MFMA tile kernel (A/B runs)
    if (mid3_on && split_products() && matmul_precision() == 0 && E.mode == E_STORE_STATS && Kd >= 512 && (Kd & 255) == 0 && (unsigned long long)M * (unsigned)A.lda * 4ull < 0x7ffffff0ull &&
        (unsigned long long)Nout * (unsigned)B.ldb * 4ull < 0x7ffffff0ull) {   // float32 products from exact bf16 splits (the default)
        ProfScope ps(st, "gemm_mid3_kernel<A%d,E%d> M=%d N=%d K=%d grid=%d", A.mode, E.mode, M, Nout, Kd, grid);
#define PNPP_MID3(AX, EMV)                                                                                                       \
    {                                                                                                                            \
        static bool granted = false;                                                                                             \
        if (!granted) {                                                                                                          \
            (void)hipFuncSetAttribute((const void *)gemm_mid3_kernel<AX, EMV>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                      (int)MID3_LDS_BYTES);                                                                      \
            granted = true;                                                                                                      \
        }                                                                                                                        \
        hipLaunchKernelGGL((gemm_mid3_kernel<AX, EMV>), dim3(grid), dim3(512), MID3_LDS_BYTES, st, G, tn);                       \
    }
        if (A.mode == A_BNRELU) PNPP_MID3(A_BNRELU, E_STORE_STATS) else PNPP_MID3(A_PLAIN, E_STORE_STATS)
#undef PNPP_MID3
        if (hipGetLastError() != hipSuccess) {
            set_error("gemm_mid3: launch failed");
            *rc = PNPP_ERR_LAUNCH;
        }
        return true;
    }
    ProfScope ps(st, "gemm_mid_kernel<A%d,E%d,T1> M=%d N=%d K=%d grid=%d", A.mode, E.mode, M, Nout, Kd, grid);
#define PNPP_MID(AX, EMV)                                                                              \
    {                                                                                                  \
        mid_grant_lds(gemm_mid_kernel<AX, true, EMV>);                                                 \
        hipLaunchKernelGGL((gemm_mid_kernel<AX, true, EMV>), dim3(grid), dim3(256), MID_LDS_BYTES, st, G, tn); \
    }
    if (A.mode == A_BNRELU) {
        if (E.mode == E_STORE_STATS) PNPP_MID(A_BNRELU, E_STORE_STATS) else PNPP_MID(A_BNRELU, E_STORE)
    } else {
        if (E.mode == E_STORE_STATS) PNPP_MID(A_PLAIN, E_STORE_STATS) else PNPP_MID(A_PLAIN, E_STORE)
    }
#undef PNPP_MID
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_mid: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

// rows per dW partial for the paired launch (a multiple of 64); the caller sizes the slab for *nsplit partials of Nc x Kp
bool mid_da_dw_plan(int M, int Nout, int Nc, int Kp, int *nsplit) {
    if (M < 512 || M > 8192 || M % MID_T || Nout % MID_T || Nc % MID_T || Kp % MID_T) return false;
    const int ta = (M / MID_T) * (Nout / MID_T), tw = (Nc / MID_T) * (Kp / MID_T);
    if (ta + tw < 192) return false;
    // the dA tiles run the whole reduction (Nc / 64 chunks); a dW tile runs M / 64 chunks per split: balance the two
    int s = 1;
    while (s < 4 && (M / MID_T) / (2 * s) >= (Nc / MID_T) && (M / (2 * s)) % MID_T == 0) s *= 2;
    *nsplit = s;
    return true;
}

bool try_launch_mid_da_dw(const AOperand &dz, const BOperand &W, int M, int Nout, int Kd, const Epilogue &E, int *nslab, const AOperand &a2,
                          int Kp, float *slab, int *nsplit_out, int *kp_pad_out, hipStream_t st, int *rc, float *dw_direct, int dw_ld) {
    *rc = PNPP_OK;
    const int Nc = Kd;
    int nsplit = 1;
    if (!mid_da_dw_plan(M, Nout, Nc, Kp, &nsplit)) return false;
    if (dz.mode != A_PLAIN || W.trans || W.perm_D >= 0 || (W.rows > 0 && W.rows != Kd)) return false;
    if (!(E.mode == E_STORE || E.mode == E_MASK_STATS) || !(a2.mode == A_PLAIN || a2.mode == A_BNRELU)) return false;
    if (!mid_ptr_ok(dz.a, dz.lda) || !mid_ptr_ok(W.b, W.ldb) || !mid_ptr_ok(E.c, E.ldc) || !mid_ptr_ok(a2.a, a2.lda) || !slab) return false;
    if (E.mode == E_MASK_STATS && !mid_ptr_ok(E.zp, E.ldc)) return false;
    if (M / MID_T > kMaxStatBlocks) return false;
    MidGemm G{dz.a, dz.lda, nullptr, nullptr, W.b, W.ldb, M, Nout, Kd, E};
    MidDw D{dz.a, dz.lda, a2.a, a2.lda, a2.scale, a2.shift, M, Nc, Kp, nsplit, M / nsplit, slab, Kp};
    const bool in_place = nsplit == 1 && dw_direct && dw_ld == Kp;   // a single row range: no partial sums, no reduction launch
    if (in_place) D.out = dw_direct, D.ldo = dw_ld;
    if (nslab) *nslab = M / MID_T;
    *nsplit_out = in_place ? 0 : nsplit, *kp_pad_out = Kp;
    const int tn = Nout / MID_T, g1 = (M / MID_T) * tn, tc = Nc / MID_T, tk = Kp / MID_T, g2 = tc * tk * nsplit;
    ProfScope ps(st, "da_dw_mid_kernel<E%d,A%d> M=%d | dA N=%d K=%d grid=%d | dW N=%d K=%d split=%d grid=%d", E.mode, a2.mode, M, Nout, Kd, g1,
                 Nc, Kp, nsplit, g2);
#define PNPP_MIDP(EMV, AX)                                                                                         \
    {                                                                                                              \
        mid_grant_lds(da_dw_mid_kernel<EMV, AX>);                                                                  \
        hipLaunchKernelGGL((da_dw_mid_kernel<EMV, AX>), dim3(g1 + g2), dim3(256), MID_LDS_BYTES, st, G, tn, g1, D, tc, tk); \
    }
    if (E.mode == E_STORE) {
        if (a2.mode == A_BNRELU) PNPP_MIDP(E_STORE, A_BNRELU) else PNPP_MIDP(E_STORE, A_PLAIN)
    } else {
        if (a2.mode == A_BNRELU) PNPP_MIDP(E_MASK_STATS, A_BNRELU) else PNPP_MIDP(E_MASK_STATS, A_PLAIN)
    }
#undef PNPP_MIDP
    if (hipGetLastError() != hipSuccess) {
        set_error("da_dw_mid: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

#ifdef MID3_STAMPS
unsigned mid3_build_flags() { return 512u; }
#else
unsigned mid3_build_flags() { return 0u; }
#endif

}  // namespace pnpp

#ifdef MID3_STAMPS
extern "C" int pnpp_debug_mid3_stamps(unsigned long long *out8, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_mid3_stamps), z, sizeof(z));
    }
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(pnpp::g_mid3_stamps), 8 * sizeof(unsigned long long));
    return 0;
}
#endif
