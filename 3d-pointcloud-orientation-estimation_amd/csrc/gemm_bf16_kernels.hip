// gemm_bf16_kernels.hip -- bf16-OPERAND variant of the weights-stationary GEMM of the grouped layers (throughput mode).
//
// The reference computes in float32 (models/pointnet_pp_8dir.py:40-42 run under torch's default dtype) and so does the
// default path of this library (gemm_kernels.hip, v_mfma_f32_32x32x2_f32).  BASELINE.json's configs[1] names a bf16 mode
// for the metric configuration; this file is that mode, selected explicitly (pnpp_set_matmul_precision(1) /
// PNPP_MATMUL=bf16) and reported under its own metric key, never as the float32 headline:
//   * the two MFMA operands are rounded to bfloat16 (round to nearest even) when they are staged in LDS; products are
//     exact in float32 and accumulated in float32 (v_mfma_f32_32x32x16_bf16, 8x the MAC rate of the f32 instruction);
//   * everything else is unchanged: activations stay float32 in HBM, BatchNorm statistics come from the float32
//     accumulators and are summed in float64, operand transforms (BN apply, ReLU, BatchNorm backward, max backward) and
//     epilogues (ReLU mask, statistics) are the float32 ones.
// Same launch contract as gemm_ws_kernel (persistent workers over 64-row tiles, one weight panel per workgroup, column
// statistics and dW partials per worker), so sa_api.hip and the reductions behind it do not know which one ran.
//
// Tile: 64 rows x 64 columns, four waves (wm, wn) each owning one 32 x 32 accumulator.  LDS images (bf16):
//   Wb [64 n][KD]   weight panel, 16-byte chunks XOR-swizzled per row
//   Ab [64 m][KD]   operand tile, same swizzle (dA / forward product: lane = row, 8 consecutive k per lane and MFMA)
//   AT [KD c][64]   FDW only: the SAME tile transposed, 8 chunks of 8 rows per column (dW product: lane = c, 8 rows per lane)
// The MFMA's row index i is mapped to tile row pi(i) = b2 + 2 b4 + 4 (i & 3) + 16 b3 (bits of i), so that the 16 rows a
// lane holds in the accumulator layout -- registers 8s..8s+7 of lane-half lh -- are the tile rows (2s + lh) + 4j,
// j = 0..7: exactly the rows ONE staging thread loaded (row slot rho = 2s + lh).  Consequences: the transposed image is
// written with one 16-byte store per (column, 8 rows), and the second operand of the fused dW product,
// relu(bn(z_{l-1})), is 8 consecutive epilogue registers per MFMA operand: each lane stores its two operands (16 bytes
// each) into a small [64 k][8 chunks] image of the same form, so that every wave can contract over all 64 rows of the
// tile and a wave owns whole dW tiles (half as many accumulators as with per-half partial sums, one partial per worker).
#include "kernels.h"

namespace pnpp {

static int g_matmul_bf16 = -1;  // -1: not decided yet (environment consulted on first use)

int matmul_precision() {
    if (g_matmul_bf16 < 0) {
        const char *e = getenv("PNPP_MATMUL");
        g_matmul_bf16 = (e && (e[0] == 'b' || e[0] == 'B')) ? 1 : 0;
    }
    return g_matmul_bf16;
}
void set_matmul_precision(int bf16) { g_matmul_bf16 = bf16 ? 1 : 0; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {  // two floats -> one dword of two bf16, round to nearest even
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint2 pk4_bf16(float a, float b, float c, float d) { return make_uint2(pk_bf16(a, b), pk_bf16(c, d)); }
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int KD, int AMODE, int EMODE, bool FDW>
__global__ void __launch_bounds__(256, (KD >= 256 ? 1 : 2))  // K = 256: the three images fill the LDS, one workgroup per CU anyway
gemm_wsb_kernel(const AOperand A, const BOperand B, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int BM = 64, BN = 64;
    constexpr int G4 = KD / 4;             // float4 column groups per row = threads per row
    constexpr int SLOTS = 256 / G4;        // row slots: 4 / 8 / 16 for KD = 256 / 128 / 64
    constexpr int RPT = BM / SLOTS;        // rows per thread and tile: 16 / 8 / 4
    constexpr int CH = KD / 8;             // 16-byte chunks per row of the bf16 images
    constexpr int PITCH = KD * 2;          // bytes per row of Wb / Ab
    constexpr int NCT = KD / 32;           // dW: 32-row tiles of dZ^T per wave
    static_assert(KD == 64 || KD == 128 || KD == 256, "reduction length");
    static_assert(!FDW || EMODE == E_MASK_STATS, "fused dW needs the ReLU-mask epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *Wb = lds, *Ab = lds + BN * PITCH, *AT = Ab + BM * PITCH;  // AT: [KD][128 bytes]
    unsigned char *aT = AT + KD * 128;                                         // FDW: [BN][128 bytes], relu(bn(z_{l-1})) transposed
    auto swz = [](int r) { return CH >= 16 ? (r & 15) : ((r >> 1) & 7); };       // chunk XOR of row r (Wb / Ab)
    auto swzT = [](int c) { return (c >> 1) & 7; };                              // chunk XOR of row c (AT, 8 chunks)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
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
    const int kq = 4 * (tid % G4), q = tid / G4;   // this thread's first column and its row slot
    // rows of the tile this thread stages: 32 h + rho + 4 j
    int hq, rho, j0;
    if constexpr (KD == 256) hq = 0, rho = q, j0 = 0;                       // both halves: i = 8 h + j
    else if constexpr (KD == 128) hq = q >> 2, rho = q & 3, j0 = 0;          // i = j
    else hq = q >> 3, rho = (q >> 1) & 3, j0 = 4 * (q & 1);                 // i = j - j0 (half a pack)
    auto row_of = [&](int i) { return KD == 256 ? 32 * (i >> 3) + rho + 4 * (i & 7) : 32 * hq + rho + 4 * (j0 + i); };

    // ---- weights: staged once per workgroup as bf16 [n][k], swizzled ----
    {
        const float *__restrict__ Bm = B.b;
        const int ldb = B.ldb;
        if (B.trans) {  // b is [Nout][Kd]: rows are output columns, four consecutive k per thread
            const bool bvec = (ldb & 3) == 0 && ((uintptr_t)Bm & 15) == 0 && B.perm_D < 0;
            for (int f = tid; f < BN * G4; f += 256) {
                const int nl = f / G4, k4 = 4 * (f % G4), n = n0 + nl;
                const float *src = Bm + (size_t)min(n, Nout - 1) * ldb;
                float t[4];
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
                *reinterpret_cast<uint2 *>(Wb + nl * PITCH + 16 * ((k4 >> 3) ^ swz(nl)) + 2 * (k4 & 7)) = pk4_bf16(t[0], t[1], t[2], t[3]);
            }
        } else {  // b is [Kd][Nout]: lane = output column -- four coalesced dword loads of consecutive rows k, ONE 8-byte LDS store of
                  // the packed group; all of a thread's loads are requested before the first is used (the first version read 16
                  // bytes along n and scattered four 2-byte stores per group from a rolled loop: one L2 round trip per iteration)
            constexpr int NWF = (KD * (BN / 4)) / 256;
            float tw[NWF][4];
#pragma unroll
            for (int j = 0; j < NWF; ++j) {
                const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN), n = min(n0 + nl, Nout - 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) tw[j][e] = Bm[(size_t)min(k4 + e, B.rows - 1) * ldb + n];
            }
#pragma unroll
            for (int j = 0; j < NWF; ++j) {
                const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
                for (int e = 0; e < 4; ++e) tw[j][e] *= (n0 + nl < Nout && k4 + e < B.rows) ? 1.f : 0.f;
                *reinterpret_cast<uint2 *>(Wb + nl * PITCH + 16 * ((k4 >> 3) ^ swz(nl)) + 2 * (k4 & 7)) =
                    pk4_bf16(tw[j][0], tw[j][1], tw[j][2], tw[j][3]);
            }
        }
    }

    // ---- per-channel constants of this thread's column group ----
    float4 c_g = make_float4(0.f, 0.f, 0.f, 0.f), c_a = c_g, c_b = c_g, c_sc = c_g, c_sh = c_g;
    if constexpr (AMODE == A_DZ || AMODE == A_DZ_POOL) {
        const float *c = A.cst + kq;
        c_g = *reinterpret_cast<const float4 *>(c);
        const float4 mu = *reinterpret_cast<const float4 *>(c + A.C), is = *reinterpret_cast<const float4 *>(c + 2 * A.C);
        const float4 c1 = *reinterpret_cast<const float4 *>(c + 3 * A.C), c2 = *reinterpret_cast<const float4 *>(c + 4 * A.C);
        // dz = g (dy - c1 - (z - mu) istd c2) = g dy + (a z + b)
        c_a = make_float4(-c_g.x * is.x * c2.x, -c_g.y * is.y * c2.y, -c_g.z * is.z * c2.z, -c_g.w * is.w * c2.w);
        c_b = make_float4(-c_g.x * c1.x - c_a.x * mu.x, -c_g.y * c1.y - c_a.y * mu.y, -c_g.z * c1.z - c_a.z * mu.z,
                          -c_g.w * c1.w - c_a.w * mu.w);
    } else if constexpr (AMODE == A_BNRELU) {
        c_sc = *reinterpret_cast<const float4 *>(A.scale + kq);
        c_sh = *reinterpret_cast<const float4 *>(A.shift + kq);
    }

    // MFMA row index -> tile row (see the header): pi(i) = b2 + 2 b4 + 4 (i & 3) + 16 b3
    const int pi_row = ((l31 >> 2) & 1) + 2 * ((l31 >> 4) & 1) + 4 * (l31 & 3) + 16 * ((l31 >> 3) & 1);
    const int arow = 32 * wm + pi_row, brow = 32 * wn + l31;
    const unsigned char *a_base = Ab + arow * PITCH, *b_base = Wb + brow * PITCH;
    const int a_x = swz(arow), b_x = swz(brow);

    double s1 = 0.0, s2 = 0.0;
    constexpr int NDW = FDW ? NCT / 2 : 1;   // dW tiles per wave: c-tiles ct = 2 t + wm, column tile wn
    f32x16 dwacc[NDW];
#pragma unroll
    for (int t = 0; t < NDW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.f;

    const int tiles = (M + BM - 1) / BM;
    float4 rp[RPT], rq[AMODE == A_DZ ? RPT : 1];
    // Addresses: (uniform tile pointer, advanced by scalar arithmetic) + (ONE 32-bit lane offset computed once).  M is a
    // multiple of 64 and Nout of 64 (launcher), so no load or store of this kernel needs a clamp -- 64-bit multiply-adds and
    // clamps per element were half of the vector instructions of the first version, and every one of them is MFMA time.
    const unsigned offA = (unsigned)row_of(0) * (unsigned)A.lda + (unsigned)kq;
    const unsigned offC = (unsigned)(32 * wm + lh) * (unsigned)E.ldc + (unsigned)(n0 + 32 * wn + l31);   // this lane's first output element
    auto fetch = [&](int m0) {
        const float *pa = (AMODE == A_DZ_POOL ? A.z : A.a) + (size_t)m0 * A.lda;
        const float *pz = A.z + (size_t)m0 * A.lda;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int dr = KD == 256 ? 32 * (i >> 3) + 4 * (i & 7) : 4 * i;   // row_of(i) - row_of(0): a compile-time constant
            rp[i] = *reinterpret_cast<const float4 *>(pa + (size_t)dr * A.lda + offA);
            if constexpr (AMODE == A_DZ) rq[i] = *reinterpret_cast<const float4 *>(pz + (size_t)dr * A.lda + offA);
        }
    };
    int tile = worker;
    if (tile < tiles) fetch(tile * BM);
    for (; tile < tiles; tile += nworkers) {
        const int m0 = tile * BM;
        // pooled gradient / arg-max of the neighbour groups this thread's rows lie in (nsample == 32: group = 32-row half)
        float4 gdm[KD == 256 ? 2 : 1];
        int4 garg[KD == 256 ? 2 : 1];
        if constexpr (AMODE == A_DZ_POOL) {
#pragma unroll
            for (int h = 0; h < (KD == 256 ? 2 : 1); ++h) {
                const int g = m0 / 32 + (KD == 256 ? h : hq);
                gdm[h] = *reinterpret_cast<const float4 *>(A.a + (size_t)g * A.lda + kq);
                garg[h] = *reinterpret_cast<const int4 *>(A.arg + (size_t)g * A.lda + kq);
            }
        }
        // rows / column of this lane's 16 accumulator registers: register 8 s + j  <->  tile row 32 wm + (2 s + lh) + 4 j.
        // The epilogue's loads go out at the TOP of the tile: they depend on nothing, the MFMA work of a bf16 tile is far too
        // short to hide a round trip, and vmcnt retires in order -- behind the prefetched operand stream (older: consumed by
        // the staging pass below), ahead of the next tile's prefetch (younger: still in flight when the epilogue waits)
        const int col = n0 + 32 * wn + l31;
        float zp[16];
        if constexpr (EMODE == E_MASK_STATS) {
            const float *pzp = E.zp + (size_t)m0 * E.ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) zp[r] = pzp[(size_t)(2 * (r >> 3) + 4 * (r & 7)) * E.ldc + offC];
        }
        __syncthreads();  // the previous tile's operand reads are done (first time: nothing; the weights are fenced below)
        float v[RPT][4];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = row_of(i);
            if constexpr (AMODE == A_PLAIN) {
                v[i][0] = rp[i].x, v[i][1] = rp[i].y, v[i][2] = rp[i].z, v[i][3] = rp[i].w;
            } else if constexpr (AMODE == A_BNRELU) {
                v[i][0] = fmaxf(fmaf(rp[i].x, c_sc.x, c_sh.x), 0.f);
                v[i][1] = fmaxf(fmaf(rp[i].y, c_sc.y, c_sh.y), 0.f);
                v[i][2] = fmaxf(fmaf(rp[i].z, c_sc.z, c_sh.z), 0.f);
                v[i][3] = fmaxf(fmaf(rp[i].w, c_sc.w, c_sh.w), 0.f);
            } else {
                float4 dy, z;
                if constexpr (AMODE == A_DZ) {
                    dy = rp[i], z = rq[i];
                } else {
                    z = rp[i];
                    const int h = KD == 256 ? (i >> 3) : 0;
                    const int kk = r & 31;   // position inside the neighbourhood
                    const float4 dm = gdm[h];
                    const int4 ia = garg[h];
                    dy.x = kk == ia.x ? dm.x : 0.f, dy.y = kk == ia.y ? dm.y : 0.f;
                    dy.z = kk == ia.z ? dm.z : 0.f, dy.w = kk == ia.w ? dm.w : 0.f;
                }
                v[i][0] = fmaf(c_g.x, dy.x, fmaf(c_a.x, z.x, c_b.x));
                v[i][1] = fmaf(c_g.y, dy.y, fmaf(c_a.y, z.y, c_b.y));
                v[i][2] = fmaf(c_g.z, dy.z, fmaf(c_a.z, z.z, c_b.z));
                v[i][3] = fmaf(c_g.w, dy.w, fmaf(c_a.w, z.w, c_b.w));
            }
            *reinterpret_cast<uint2 *>(Ab + r * PITCH + 16 * ((kq >> 3) ^ swz(r)) + 2 * (kq & 7)) = pk4_bf16(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
        if constexpr (FDW) {  // the transposed image: per column one 16-byte chunk per (half h, row slot rho) = 8 rows rho + 4 j
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = kq + e;
                unsigned char *crow = AT + c * 128;
                if constexpr (KD == 256) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint4 pk = make_uint4(pk_bf16(v[8 * h + 0][e], v[8 * h + 1][e]), pk_bf16(v[8 * h + 2][e], v[8 * h + 3][e]),
                                                    pk_bf16(v[8 * h + 4][e], v[8 * h + 5][e]), pk_bf16(v[8 * h + 6][e], v[8 * h + 7][e]));
                        *reinterpret_cast<uint4 *>(crow + 16 * ((4 * h + rho) ^ swzT(c))) = pk;
                    }
                } else if constexpr (KD == 128) {
                    const uint4 pk = make_uint4(pk_bf16(v[0][e], v[1][e]), pk_bf16(v[2][e], v[3][e]), pk_bf16(v[4][e], v[5][e]),
                                                pk_bf16(v[6][e], v[7][e]));
                    *reinterpret_cast<uint4 *>(crow + 16 * ((4 * hq + rho) ^ swzT(c))) = pk;
                } else {  // four rows = half a chunk
                    *reinterpret_cast<uint2 *>(crow + 16 * ((4 * hq + rho) ^ swzT(c)) + 2 * j0) =
                        make_uint2(pk_bf16(v[0][e], v[1][e]), pk_bf16(v[2][e], v[3][e]));
                }
            }
        }
        __syncthreads();
        if (tile + nworkers < tiles) fetch((tile + nworkers) * BM);  // next tile's HBM stream flies during the rest of this tile

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {   // product over KD: 16 k per MFMA; lane half lh takes the chunk 2 st + lh
            uint4 ra[2], rb[2];
            auto ld = [&](int buf, int st) {
                const int ch = 2 * st + lh;
                ra[buf] = *reinterpret_cast<const uint4 *>(a_base + 16 * (ch ^ a_x));
                rb[buf] = *reinterpret_cast<const uint4 *>(b_base + 16 * (ch ^ b_x));
            };
            ld(0, 0);
#pragma unroll
            for (int st = 0; st < KD / 16; ++st) {
                if (st + 1 < KD / 16) ld((st + 1) & 1, st + 1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ra[st & 1]), as_bf16x8(rb[st & 1]), acc, 0, 0, 0);
            }
        }

        // ---- epilogue ----
        float t1 = 0.f, t2 = 0.f;
        float av[16];   // FDW: relu(bn(z_{l-1})) of this lane's 16 rows, the second operand of the dW product
        {
            float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
            if constexpr (EMODE == E_MASK_STATS) sc = E.scale[col], sh = E.shift[col], mu = E.mu[col], is = E.istd[col];
            float *pc = E.c + (size_t)m0 * E.ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float o = acc[r];
                if constexpr (EMODE == E_STORE_STATS) {
                    t1 += o;
                    t2 = fmaf(o, o, t2);
                } else if constexpr (EMODE == E_MASK_STATS) {
                    const float a0 = fmaf(zp[r], sc, sh);
                    o = a0 > 0.f ? o : 0.f;
                    t1 += o;
                    t2 = fmaf(o, (zp[r] - mu) * is, t2);
                    av[r] = fmaxf(a0, 0.f);
                }
                pc[(size_t)(2 * (r >> 3) + 4 * (r & 7)) * E.ldc + offC] = o;
            }
        }
        if constexpr (EMODE != E_STORE) s1 += (double)t1, s2 += (double)t2;

        if constexpr (FDW) {
            // relu(bn(z_{l-1})) of this lane's column k and its 16 rows -> the transposed image: chunk 4 wm + 2 s + lh of row k
            // holds rows (2 s + lh) + 4 j of half wm, the same slot the dZ image uses for them
            {
                const int k = 32 * wn + l31;
                unsigned char *krow = aT + k * 128;
                const int kx = swzT(k);
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    *reinterpret_cast<uint4 *>(krow + 16 * ((4 * wm + 2 * s + lh) ^ kx)) =
                        make_uint4(pk_bf16(av[8 * s + 0], av[8 * s + 1]), pk_bf16(av[8 * s + 2], av[8 * s + 3]),
                                   pk_bf16(av[8 * s + 4], av[8 * s + 5]), pk_bf16(av[8 * s + 6], av[8 * s + 7]));
            }
            __syncthreads();
            // dW[c][k] += sum over the tile's 64 rows of dZ[row][c] a[row][k]: step p = 2 h + s contracts the 16 rows
            // {(2 s + g) + 4 j of half h : g = lane half}; both operands read chunk 4 h + 2 s + lh of their row
            uint4 bop[4];
            {
                const int k = 32 * wn + l31;
                const unsigned char *krow = aT + k * 128;
                const int kx = swzT(k);
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) bop[p2] = *reinterpret_cast<const uint4 *>(krow + 16 * ((4 * (p2 >> 1) + 2 * (p2 & 1) + lh) ^ kx));
            }
#pragma unroll
            for (int t = 0; t < NDW; ++t) {
                const int c = 32 * (2 * t + wm) + l31;
                const unsigned char *crow = AT + c * 128;
                const int cx = swzT(c);
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const uint4 aop = *reinterpret_cast<const uint4 *>(crow + 16 * ((4 * (p2 >> 1) + 2 * (p2 & 1) + lh) ^ cx));
                    dwacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(aop), as_bf16x8(bop[p2]), dwacc[t], 0, 0, 0);
                }
            }
        }
    }

    if constexpr (FDW) {  // one partial dW per worker: dwslab[worker][c][n0 + k]; standard accumulator layout
#pragma unroll
        for (int t = 0; t < NDW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * (2 * t + wm) + (r & 3) + 8 * (r >> 2) + 4 * lh, k = n0 + 32 * wn + l31;
                if (k < Nout) E.dwslab[((size_t)worker * KD + c) * E.dw_ld + k] = dwacc[t][r];
            }
    }

    if constexpr (EMODE != E_STORE) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);  // [2 wm][2][BN]
        const double a = s1 + shfl_xor_f64(s1, 32), b = s2 + shfl_xor_f64(s2, 32);
        if (lh == 0) {
            const int cl = 32 * wn + l31;
            red[(wm * 2 + 0) * BN + cl] = a;
            red[(wm * 2 + 1) * BN + cl] = b;
        }
        __syncthreads();
        for (int f = tid; f < 2 * BN; f += 256) {
            const int which = f / BN, cl = f % BN;
            const double t = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl];
            if (n0 + cl < Nout) E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
        }
    }
}

template <int KD, int AM, int EM, bool FDW>
static int launch_wsb_one(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                          int *dw_slabs) {
    const int tiles = cdiv(M, 64), ncol = cdiv(Nout, 64);
    size_t lds = (size_t)64 * KD * 2 * 2 + (FDW ? (size_t)KD * 128 + 64 * 128 : 0);
    if (lds < 4096) lds = 4096;  // the column-statistics reduction reuses the LDS: 2 x 2 x 64 doubles
    int per_cu = (int)((160 * 1024) / lds);
    const int cap_cu = FDW ? 2 : 4;          // __launch_bounds__(256, 2): the fused kernels need ~200 registers, the others < 128
    if (per_cu > cap_cu) per_cu = cap_cu;
    int workers = (256 * per_cu) / ncol;
    if (workers > tiles) workers = tiles;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (workers < 1) workers = 1;
    if (nslab) *nslab = workers;
    if (dw_slabs) *dw_slabs = FDW ? workers : 0;
    ProfScope ps(st, "gemm_wsb_kernel<%d,64,64,A%d,E%d%s> M=%d N=%d K=%d grid=%dx1", KD, AM, EM, FDW ? ",dW" : "", M, Nout, KD,
                 workers * ncol);
    auto kfn = gemm_wsb_kernel<KD, AM, EM, FDW>;
    static size_t lds_granted = 0;
    if (lds > 48 * 1024 && lds > lds_granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_granted = lds;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(256), lds, st, A, B, M, Nout, ncol, E);
    PNPP_CHECK_LAUNCH("gemm_wsb");
    return PNPP_OK;
}

template <int KD, int AM>
static int launch_wsb_e(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                        int *dw_slabs) {
    if (dw_slabs) *dw_slabs = 0;
    switch (E.mode) {
        case E_STORE: return launch_wsb_one<KD, AM, E_STORE, false>(A, B, M, Nout, E, nslab, st, nullptr);
        case E_STORE_STATS: return launch_wsb_one<KD, AM, E_STORE_STATS, false>(A, B, M, Nout, E, nslab, st, nullptr);
        case E_MASK_STATS:
            if constexpr (AM == A_DZ || AM == A_DZ_POOL) {
                if (E.dwslab && dw_slabs) return launch_wsb_one<KD, AM, E_MASK_STATS, true>(A, B, M, Nout, E, nslab, st, dw_slabs);
            }
            return launch_wsb_one<KD, AM, E_MASK_STATS, false>(A, B, M, Nout, E, nslab, st, nullptr);
    }
    set_error("gemm_wsb: bad epilogue mode %d", E.mode);
    return PNPP_ERR_ARG;
}

template <int KD>
static int launch_wsb_k(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int *nslab, hipStream_t st,
                        int *dw_slabs) {
    switch (A.mode) {
        case A_PLAIN: return launch_wsb_e<KD, A_PLAIN>(A, B, M, Nout, E, nslab, st, dw_slabs);
        case A_BNRELU: return launch_wsb_e<KD, A_BNRELU>(A, B, M, Nout, E, nslab, st, dw_slabs);
        case A_DZ: return launch_wsb_e<KD, A_DZ>(A, B, M, Nout, E, nslab, st, dw_slabs);
        case A_DZ_POOL: return launch_wsb_e<KD, A_DZ_POOL>(A, B, M, Nout, E, nslab, st, dw_slabs);
    }
    set_error("gemm_wsb: bad A mode %d", A.mode);
    return PNPP_ERR_ARG;
}

// The shapes the weights-stationary float32 kernel takes for dense operands (M >= 8192 rows, K in {64, 128, 256}, N a
// multiple of 64); anything else -- gathers, group_all levels, the head -- stays on the float32 kernels.
bool try_launch_ws_bf16(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st,
                        int *rc, int *dw_slabs) {
    if (!matmul_precision()) return false;
    if (M < 8192 || M % 64 != 0 || Nout % 64 != 0) return false;   // whole 64 x 64 tiles only: the kernel has no bounds tests
    if (!(A.mode == A_PLAIN || A.mode == A_BNRELU || A.mode == A_DZ || A.mode == A_DZ_POOL)) return false;
    if (A.lda != Kd || A.lda % 4 != 0 || ((uintptr_t)A.a & 15) != 0) return false;
    if (A.mode == A_DZ_POOL && (A.K != 32 || M % 32 != 0)) return false;
    if ((A.mode == A_DZ || A.mode == A_DZ_POOL) && A.C != Kd) return false;
    if (E.mode == E_BN_APPLY) return false;
    switch (Kd) {
        case 64: *rc = launch_wsb_k<64>(A, B, M, Nout, E, nslab, st, dw_slabs); return true;
        case 128: *rc = launch_wsb_k<128>(A, B, M, Nout, E, nslab, st, dw_slabs); return true;
        case 256: *rc = launch_wsb_k<256>(A, B, M, Nout, E, nslab, st, dw_slabs); return true;
    }
    return false;
}

}  // namespace pnpp

extern "C" int pnpp_set_matmul_precision(int bf16_operands) {
    pnpp::set_matmul_precision(bf16_operands);
    return PNPP_OK;
}
extern "C" int pnpp_get_matmul_precision(void) { return pnpp::matmul_precision(); }
