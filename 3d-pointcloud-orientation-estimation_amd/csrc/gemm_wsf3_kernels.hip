// gemm_wsf3_kernels.hip -- the forward 1x1 convolutions of the grouped levels (gemm_wsf_kernels.hip: wave-private 32-row strips, no
// barrier in the strip loop) with the float32 products formed on the bf16 matrix pipe from EXACT three-way operand splits.
//
// Reference: models/pointnet_pp_8dir.py:40-42 (conv -> BatchNorm -> ReLU), float32.
//
// Why.  gfx950 has no reduced-width float32 MFMA: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (64 FLOP/clk/SIMD), and on the
// float32 forward kernels the MFMA time and the HBM time ADD.  A float32 number is the exact sum of three
// bfloat16 numbers (24 significand bits = 8 + 8 + 8; bf16 has float32's exponent range):
//     a = a_h + a_m + a_l,   a_h = bf16(a),  a_m = bf16(a - a_h),  a_l = a - a_h - a_m   (every subtraction exact, a_l exact in bf16)
// so  a b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m) + [a_m b_l + a_l b_m + a_l b_l].
// The bracket is at most 2^-25 |a b| -- below the rounding of the float32 accumulation itself (2^-24 per addition) -- and is dropped; the
// six products kept are exact in float32 (8 x 8 significand bits) and are accumulated in float32 by v_mfma_f32_32x32x16_bf16: six
// instructions of 32 cycles for 16 reduction steps against eight of 64 cycles = 2.67 x the float32 MFMA rate.  The leading products go
// to one accumulator and the five small ones to a second, added once per strip, so the rounding of the sum is that of a float32
// accumulation of the leading products.  (Why two accumulators: v_mfma_f32_32x32x16_bf16 aligns its 16 products and the accumulator to
// the largest exponent among them and drops what lies 2^-26 below it -- tools/mfma_round.hip -- so small addends must not meet the
// large sum inside the instruction.)  Not a reduced-precision mode: tests/test_gpu_levels_routed.py holds it to the same gates as the
// float32 MFMA form, and tests/test_gpu_split_products.py measures both against float64.  (Infinities do not survive the split:
// inf - inf; the float32 form gives inf where this one gives NaN.  Operands below 2^-110 lose their low pieces to underflow.)
//
// Layout.  Per wave 3 planes [32 rows][64 k] bf16 of its strip chunk (128-byte rows, 16-byte groups XOR-swizzled by x(r) = 4 bit1(r) +
// bits3:2(r): the four 16-lane groups of a ds_read_b128 then touch all 64 banks once); per workgroup 3 planes [64 n][KD] bf16 of the weight panel.
// K = 64: 4 waves, 24 + 48 = 72 KB, two workgroups per CU; K = 128: 8 waves, 48 + 96 = 144 KB, one workgroup per CU -- eight waves per
// CU either way: while one wave of a SIMD splits and stages (VALU), the other multiplies (an MFMA holds the vector issue for 8 of its
// 32 cycles only).
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned x3_pk(float lo, float hi) {   // two floats -> two bf16 in one dword, round to nearest even
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ float x3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float x3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// four floats -> their three bf16 pieces, four per 8-byte word
__device__ __forceinline__ void x3_split4(const f32x4 v, uint2 &h, uint2 &m, uint2 &l) {
    h.x = x3_pk(v[0], v[1]), h.y = x3_pk(v[2], v[3]);
    float r0 = v[0] - x3_lo(h.x), r1 = v[1] - x3_hi(h.x), r2 = v[2] - x3_lo(h.y), r3 = v[3] - x3_hi(h.y);
    m.x = x3_pk(r0, r1), m.y = x3_pk(r2, r3);
    r0 -= x3_lo(m.x), r1 -= x3_hi(m.x), r2 -= x3_lo(m.y), r3 -= x3_hi(m.y);
    l.x = x3_pk(r0, r1), l.y = x3_pk(r2, r3);
}
__device__ __forceinline__ bf16x8 x3_op(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsf3_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wsf3_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}

// KD in {64, 128}; NW waves per workgroup; NT column tiles of 32 per wave; AX = A_PLAIN or A_BNRELU; EM = E_STORE or E_STORE_STATS
template <int KD, int NW, int NT, int AX, int EM>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1)
gemm_wsf3_kernel(const float *__restrict__ A, int lda, const float *__restrict__ scale, const float *__restrict__ shift,
                 const float *__restrict__ W, int ldw, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int BN = NT * 32, NC = KD / 64, NTHR = NW * 64;
    constexpr int WPITCH = KD * 2, WPLANE = BN * WPITCH;   // bytes: a row and a plane of the weight panel
    constexpr int APLANE = 32 * 128, ASTRIP = 3 * APLANE;  // bytes: a plane of a strip chunk, a wave's three planes
    constexpr int WCH = KD / 8;                            // 16-byte groups per panel row
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    unsigned char *Wp = lds3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *Ap = lds3 + 3 * WPLANE + wave * ASTRIP;
    const int l31 = lane & 31, lh = lane >> 5;
    auto swzA = [](int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); };   // x(r) = 4 bit1(r) + bits3:2(r): conflict-free row reads, and row
                                                                                 // rb + 4 i of the staging map is one register ^ ((i & 3) << 4) + 512 i
    auto swzW = [](int n) { return WCH >= 16 ? (n & 15) : ((n >> 1) & 7); };

    // XCD-aware map (as gemm_wsf_kernel): the column blocks of one worker sit on one XCD and share its L2
    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;

    // per-channel constants of this lane's column group (k = 64 c + 4 q .. + 3)
    const int q = lane & 15, q4 = 4 * q, rb = lane >> 4;   // staging map: column group lane % 16, rows lane / 16 + 4 i
    float4 sc[NC], sh[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sc[c] = make_float4(1.f, 1.f, 1.f, 1.f), sh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (AX == A_BNRELU) {
            sc[c] = *reinterpret_cast<const float4 *>(scale + 64 * c + q4);
            sh[c] = *reinterpret_cast<const float4 *>(shift + 64 * c + q4);
        }
    }
    // weight panel W[n0 .. n0 + 64)[0 .. KD): consecutive lanes take consecutive 16-byte groups of one row; split once per workgroup
    {
        constexpr int NWF = (KD / 4) * BN / NTHR;
        f32x4 tw[NWF];
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + NTHR * j, nl = f / (KD / 4), k4 = 4 * (f % (KD / 4));
            tw[j] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n0 + nl) * ldw + k4);
        }
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + NTHR * j, nl = f / (KD / 4), k4 = 4 * (f % (KD / 4));
            uint2 h, m, l;
            x3_split4(tw[j], h, m, l);
            unsigned char *dst = Wp + nl * WPITCH + 16 * ((k4 >> 3) ^ swzW(nl)) + 2 * (k4 & 7);
            *reinterpret_cast<uint2 *>(dst) = h;
            *reinterpret_cast<uint2 *>(dst + WPLANE) = m;
            *reinterpret_cast<uint2 *>(dst + 2 * WPLANE) = l;
        }
    }

    // strips: this wave takes strip (worker * NW + wave) + i * (nworkers * NW)
    const int nstrips = M / 32, stride = nworkers * NW;
    int strip = worker * NW + wave;
    const __amdgpu_buffer_rsrc_t resA = wsf3_rsrc(A);
    const unsigned oa0 = 4u * ((unsigned)rb * (unsigned)lda + (unsigned)q4);   // rows rb + 4 i: the i part rides in the scalar offset
    f32x4 ra[NC][8];
    auto fetch = [&](int s) {
        const unsigned so = (unsigned)s * 32u * (unsigned)lda * 4u;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[c][i] = wsf3_load4(resA, oa0, so + 256u * (unsigned)c + (unsigned)i * (16u * (unsigned)lda));
    };
    if (strip < nstrips) fetch(strip);

    double s1[NT], s2[NT];
    float sg[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        s1[j] = s2[j] = 0.0;
        sg[j] = 1.f;
        if constexpr (EM == E_STORE_STATS) {
            if (E.pool_ext && E.pool_gamma) sg[j] = E.pool_gamma[n0 + j * 32 + l31] >= 0.f ? 1.f : -1.f;
        }
    }
    __syncthreads();   // the weight panel is complete; from here on the waves run on their own

    // this lane's staging slot in a strip plane: row rb + 4 i, 8 bytes at k = q4 (half q & 1 of group q >> 1)
    const unsigned wofs0 = (unsigned)(rb * 128 + 16 * ((q >> 1) ^ (((rb >> 1) & 1) << 2)) + 8 * (q & 1));   // row rb + 4 i: ^ ((i & 3) << 4), + 512 i
    const unsigned char *arow = Ap + l31 * 128;
    const int ax = swzA(l31);
    const unsigned char *brow[NT];
    int bx[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + l31;
        brow[j] = Wp + n * WPITCH;
        bx[j] = swzW(n);
    }
    // A strip's column tiles are multiplied two at a time (NH passes over the strip image: 64 accumulator registers per pass; the image is
    // staged, i.e. read, transformed and split, ONCE for all NT tiles)
    constexpr int NH = NT / 2;
    static_assert(NT == 2 || (NT == 4 && NC == 1), "four tiles per wave: one chunk (the passes re-read the chunk image)");
    // stage chunk c of the strip: transform and split in registers, three ds_write_b64 per group (in-order per wave: the reads of the
    // previous chunk were issued before these writes, and the reads that follow come after them)
    auto stage = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = ra[c][i];
            if constexpr (AX == A_BNRELU) {
                v[0] = fmaxf(fmaf(v[0], sc[c].x, sh[c].x), 0.f), v[1] = fmaxf(fmaf(v[1], sc[c].y, sh[c].y), 0.f);
                v[2] = fmaxf(fmaf(v[2], sc[c].z, sh[c].z), 0.f), v[3] = fmaxf(fmaf(v[3], sc[c].w, sh[c].w), 0.f);
            }
            uint2 h, m, l;
            x3_split4(v, h, m, l);
            unsigned char *dst = Ap + (wofs0 ^ (unsigned)((i & 3) << 4)) + i * 512;
            *reinterpret_cast<uint2 *>(dst) = h;
            *reinterpret_cast<uint2 *>(dst + APLANE) = m;
            *reinterpret_cast<uint2 *>(dst + 2 * APLANE) = l;
        }
    };
    // tiles 2 hh, 2 hh + 1 of chunk c: step t covers k = 64 c + 16 t + 8 lh + (0 .. 7): group 2 t + lh of the strip chunk, group
    // 8 c + 2 t + lh of the panel rows
    auto product = [&](int c, int hh, f32x16 (&acc)[2], f32x16 (&accl)[2]) {
        uint4 fa[2][3], fb[2][2][3];
        auto ld = [&](int buf, int t) {
            const int g = 2 * t + lh;
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[buf][p] = *reinterpret_cast<const uint4 *>(arow + p * APLANE + 16 * (g ^ ax));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[buf][j][p] = *reinterpret_cast<const uint4 *>(brow[2 * hh + j] + p * WPLANE + 16 * ((8 * c + g) ^ bx[2 * hh + j]));
        };
        auto mm = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 ah = x3_op(fa[buf][0]), am = x3_op(fa[buf][1]), al = x3_op(fa[buf][2]);
                const bf16x8 bh = x3_op(fb[buf][j][0]), bm = x3_op(fb[buf][j][1]), bl = x3_op(fb[buf][j][2]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accl[j], 0, 0, 0);
                accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accl[j], 0, 0, 0);
                accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, accl[j], 0, 0, 0);
                accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, accl[j], 0, 0, 0);
                accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, accl[j], 0, 0, 0);
            }
        };
        ld(0, 0);
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            ld(1, t + 1);
            mm(0);
            if (t + 2 < 4) ld(0, t + 2);
            mm(1);
        }
    };
    // epilogue of tiles 2 hh, 2 hh + 1 of the strip: 16 rows of one column per lane and column tile
    auto epilogue = [&](int hh, f32x16 (&acc)[2], f32x16 (&accl)[2]) {
        float *tb = E.c + (size_t)(strip * 32 + 4 * lh) * E.ldc + n0 + l31;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int j = 2 * hh + j2;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[j2][r] + accl[j2][r];
                acc[j2][r] = v;
                tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc + j * 32] = v;
                t1 += v, t2 = fmaf(v, v, t2);
            }
            if constexpr (EM == E_STORE_STATS) {
                s1[j] += (double)t1, s2[j] += (double)t2;
                if (E.pool_ext) {   // the strip is one neighbourhood: extreme pre-BN value per column and its first row
                    float mx = sg[j] * acc[j2][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sg[j] * acc[j2][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    int a = 64;
#pragma unroll
                    for (int r = 15; r >= 0; --r) a = (sg[j] * acc[j2][r] == mx) ? (r & 3) + 8 * (r >> 2) + 4 * lh : a;
                    a = min(a, __shfl_xor(a, 32, 64));
                    if (lh == 0) {
                        const size_t gi = (size_t)strip * E.ldc + n0 + j * 32 + l31;
                        E.pool_ext[gi] = sg[j] * mx;
                        E.pool_arg[gi] = a;
                    }
                }
            }
        }
    };
    for (; strip < nstrips; strip += stride) {
        if constexpr (NH == 1) {
            f32x16 acc[2], accl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f, accl[j][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                stage(c);
                if (c == NC - 1 && strip + stride < nstrips) fetch(strip + stride);   // the next strip flies during the MFMAs
                product(c, 0, acc, accl);
            }
            epilogue(0, acc, accl);
        } else {
            stage(0);
            if (strip + stride < nstrips) fetch(strip + stride);
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                f32x16 acc[2], accl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f, accl[j][r] = 0.f;
                product(0, hh, acc, accl);
                epilogue(hh, acc, accl);
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler would overlap the passes: their accumulators do not fit together)
            }
        }
    }

    if constexpr (EM == E_STORE_STATS) {
        // per-column sums of this worker: lane halves, then the waves in wave order
        __syncthreads();   // every wave is done with the panel and its strip
        double *red = reinterpret_cast<double *>(lds3);   // [NW waves][2][BN]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const double a = s1[j] + shfl_xor_f64(s1[j], 32), b = s2[j] + shfl_xor_f64(s2[j], 32);
            if (lh == 0) red[(wave * 2 + 0) * BN + j * 32 + l31] = a, red[(wave * 2 + 1) * BN + j * 32 + l31] = b;
        }
        __syncthreads();
        for (int f = tid; f < 2 * BN; f += NTHR) {
            const int which = f / BN, cl = f % BN;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * BN + cl];
            E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
        }
    }
}

// 1 (default): the float32 products of the grouped layers' large GEMMs on v_mfma_f32_32x32x16_bf16 from exact three-way operand splits
// (this file, gemm_wsd3_kernels.hip); 0: on v_mfma_f32_32x32x2_f32 (PNPP_SPLIT_PRODUCTS=0 / pnpp_set_split_products)
static int g_split_products = -1;
int split_products() {
    if (g_split_products < 0) {
        const char *v = getenv("PNPP_SPLIT_PRODUCTS");
        g_split_products = (v && atoi(v) == 0) ? 0 : 1;
    }
    return g_split_products;
}
void set_split_products(int on) { g_split_products = on ? 1 : 0; }

template <int KD, int NW, int NT, int AX, int EM>
static void wsf3_launch(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int workers, int ncol, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * (NT * 32) * KD * 2 + (size_t)NW * 3 * 32 * 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kfn = gemm_wsf3_kernel<KD, NW, NT, AX, EM>;
    static bool granted = false;
    if (lds > 48 * 1024 && !granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(NW * 64), lds, st, A.a, A.lda, A.scale, A.shift, B.b, B.ldb, M, Nout, ncol, E);
}

// the shapes gemm_wsf_kernel takes (wsf_applies), when split products are on
bool try_launch_wsf3(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc) {
    *rc = PNPP_OK;
    if (!split_products() || !wsf_applies(A, B, M, Nout, Kd, E)) return false;
    // two column tiles per wave.  (PNPP_WSF3_NT4=1: four tiles per wave for K = 64 with 128 or more columns -- the strip is read, transformed
    // and split once for 128 columns, eight waves in one workgroup per CU -- measured 23.2 - 23.6 us against 22.6 us on the sa1 launch: the
    // launch is bound by its 67 MB of stores and by latency, not by the vector work the form saves)
    static const bool nt4_on = getenv("PNPP_WSF3_NT4") && atoi(getenv("PNPP_WSF3_NT4")) != 0;
    const bool nt4 = nt4_on && Kd == 64 && Nout % 128 == 0;
    const int BNsel = nt4 ? 128 : 64, ncol = Nout / BNsel, nstrips = M / 32;
    const int NW = (Kd == 64 && !nt4) ? 4 : 8;
    int workers = (NW == 4 ? 512 : 256) / ncol;   // eight waves per CU either way
    if (workers * NW > nstrips) workers = (nstrips + NW - 1) / NW;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (workers < 1) workers = 1;
    if (nslab) *nslab = workers;
    ProfScope ps(st, "gemm_wsf3_kernel<%d,A%d,E%d> M=%d N=%d K=%d grid=%dx1", Kd, A.mode, E.mode, M, Nout, Kd, workers * ncol);
#define PNPP_WSF3(KDV, NWV, NTV)                                                                                                    \
    {                                                                                                                               \
        if (A.mode == A_BNRELU) {                                                                                                   \
            if (E.mode == E_STORE_STATS) wsf3_launch<KDV, NWV, NTV, A_BNRELU, E_STORE_STATS>(A, B, M, Nout, E, workers, ncol, st);  \
            else wsf3_launch<KDV, NWV, NTV, A_BNRELU, E_STORE>(A, B, M, Nout, E, workers, ncol, st);                                \
        } else {                                                                                                                    \
            if (E.mode == E_STORE_STATS) wsf3_launch<KDV, NWV, NTV, A_PLAIN, E_STORE_STATS>(A, B, M, Nout, E, workers, ncol, st);   \
            else wsf3_launch<KDV, NWV, NTV, A_PLAIN, E_STORE>(A, B, M, Nout, E, workers, ncol, st);                                 \
        }                                                                                                                           \
    }
    if (Kd == 64 && nt4) PNPP_WSF3(64, 8, 4)
    else if (Kd == 64) PNPP_WSF3(64, 4, 2)
    else PNPP_WSF3(128, 8, 2)
#undef PNPP_WSF3
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsf3: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

}  // namespace pnpp

extern "C" int pnpp_set_split_products(int on) {
    pnpp::set_split_products(on);
    return PNPP_OK;
}
extern "C" int pnpp_get_split_products(void) { return pnpp::split_products(); }
