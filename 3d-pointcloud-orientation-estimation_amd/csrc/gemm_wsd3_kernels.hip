// gemm_wsd3_kernels.hip -- the fused backward products of the grouped levels (dA + ReLU mask + BatchNorm-backward sums + dW in one
// launch) with the float32 products formed on the bf16 matrix pipe from exact three-way operand splits (gemm_wsf3_kernels.hip has the
// arithmetic), as a PAIR of waves per 32-row strip on one SIMD: a producer that builds dZ and a consumer that multiplies.
//
// Reference: the autograd backward of conv -> BatchNorm -> ReLU (-> max over the 32 neighbours) (models/pointnet_pp_8dir.py:40-42):
//   dZ_l     = BatchNorm-backward(dY_l, Z_l)    A_DZ_POOL: dY_l rebuilt from the pooled gradient and the arg-max rows (a level's last
//                                               layer); A_DZ: dY_l dense (a middle layer)
//   dY_{l-1} = (dZ_l W_l) masked by ReLU'(layer l-1), + its BatchNorm-backward column sums
//   dW_l     = dZ_l^T relu(bn(Z_{l-1}))
//
// Why pairs.  With split products a strip's two products are 2 x 96 bf16 MFMAs = 6,144 cycles of the matrix pipe, and building the
// operands (BatchNorm-backward, three-way splits, ReLU mask, sums) is about as many cycles of vector issue.  A bf16 MFMA holds the
// vector issue for 8 of its 32 cycles, so the two kinds of work CAN overlap on a SIMD -- but only from two waves: one wave per SIMD
// (gemm_wsp_kernel's form; the dW accumulators of a strip owner alone are 128 registers) runs them back to back and waits out every
// LDS latency alone (a first version of this kernel in that form: 48.4 us against the float32 kernel's 50.4; for the dense gradient
// 26.0 against 30.4).  Two waves per SIMD leave 256 registers each, which one wave's share of BOTH products and the operand stream
// does not fit.  So the work is split between wave w (P, waves 0-3) and wave w + 4 (C) of a workgroup, which sit on the same SIMD
// and share two chunk-image buffers (three bf16 planes of [32 rows][64 channels] each) and two counters in LDS: `ready` (chunks P has
// published) and `done` (chunks C has finished reading).  P writes chunk k into buffer k & 1 once done >= k - 1; C reads it once
// ready >= k + 1.  LDS operations of one wave complete in order and the counters only grow, so there is no cycle to wait in; the
// polls are bounded all the same (a poll that gave up sets g_wsd3_timeouts and goes on: wrong numbers, never a hang; the tests check
// the mark).  Every workgroup owns 32 output columns: a wave's share of dW is 32 or 64 registers.
//   P builds the image -- streams Z (and dY) one strip ahead, computes dZ (one compare + select + FMA per element), splits it, writes
//     it: vector work and memory;
//   C multiplies: dA from the image rows x the weight panel (small products in an accumulator of their own), dW with the activation
//     operand relu(bn(z_{l-1})) as eight consecutive accumulator-layout registers per 32x32x16 operand and the dZ operand from the SAME
//     image read transposed (ds_read_b64_tr_b16, rows in the accumulator layout's order: no second image), then mask, store, sums.
//   dW is split between the two waves: the tiles of a strip's first chunks accumulate in C, those of its second half in P, which
//     multiplies its own image for them behind publishing it (a quarter of the strip's MFMAs; LDS is in order per wave) -- at K = 256
//     because all of dW (128 registers) does not fit beside the dA tile, at K = 128 for balance (+0.9 % on the step, one box).
//   K = 128 (SA1's last layer 128 -> 64, SA2's middle layer 128 -> 128): P also builds the strip's activation fragments for C and hands
//     them over through LDS; a second accumulator set for dW's small products where the pooled-gradient form has the registers.
//   K = 256 (SA2's last layer 256 -> 128): no LDS is left for the hand-off, both waves build the activation fragments.  Stamps: P 54 k
//     ticks, C 57 k per launch -- balanced.  (A first arrangement for K = 256 had P multiply dA and run the epilogue and C hold all of
//     dW: P was the critical path, C waited 48 %: 37.4 against 34.6 us on one box.)
// Image layout: 16-byte group g of row r at g ^ x(r), x(r) = 4 bit1(r) + bits3:2(r): conflict-free for the row reads (the four
// 16-lane groups of a ds_read_b128 see eight different x per row parity) and for the transposed reads (rows r and r + 2 of a block
// differ in x's bit 2).  Addresses: ONE lane-offset register per stream, everything uniform in the instructions' scalar offsets -- a
// spilled address register is reloaded behind s_waitcnt vmcnt(0) and serialises every load behind it (measured: 74 % of a wave's time).
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

typedef __bf16 wd3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wd3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wd3_f32x2 __attribute__((ext_vector_type(2)));
typedef short wd3_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned wd3_pk(float lo, float hi) {
    const wd3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wd3_bf16x2));
}
__device__ __forceinline__ float wd3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float wd3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void wd3_split4(const f32x4 v, uint2 &h, uint2 &m, uint2 &l) {
    h.x = wd3_pk(v[0], v[1]), h.y = wd3_pk(v[2], v[3]);
    float r0 = v[0] - wd3_lo(h.x), r1 = v[1] - wd3_hi(h.x), r2 = v[2] - wd3_lo(h.y), r3 = v[3] - wd3_hi(h.y);
    m.x = wd3_pk(r0, r1), m.y = wd3_pk(r2, r3);
    r0 -= wd3_lo(m.x), r1 -= wd3_hi(m.x), r2 -= wd3_lo(m.y), r3 -= wd3_hi(m.y);
    l.x = wd3_pk(r0, r1), l.y = wd3_pk(r2, r3);
}
__device__ __forceinline__ wd3_bf16x8 wd3_op(uint4 v) { return __builtin_bit_cast(wd3_bf16x8, v); }
__device__ __forceinline__ uint2 wd3_tr(const unsigned char *p) {   // ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes, transposed
    const wd3_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wd3_s16x4 *)(p));
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wd3_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wd3_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}
__device__ __forceinline__ float wd3_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}
typedef __attribute__((address_space(3))) volatile unsigned wd3_flag;   // a counter in LDS (ds_read_b32 / ds_write_b32, never flat)
__device__ int g_wsd3_timeouts;   // set by a poll that gave up (pnpp_debug_wsd3_timeouts)
// waits until *f >= target (f: an LDS counter that only grows); false after 2^16 polls (a few milliseconds: a legitimate wait is
// tens of microseconds)
__device__ __forceinline__ bool wd3_wait(wd3_flag *f, unsigned target) {
    bool ok = true;
    unsigned spins = 0;
    while ((unsigned)__builtin_amdgcn_readfirstlane((int)*f) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 16)) {
            ok = false;
            break;
        }
    }
    asm volatile("" ::: "memory");
    return ok;
}
// publishes a counter value after every LDS operation of this wave issued so far has completed
__device__ __forceinline__ void wd3_post(wd3_flag *f, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *f = v;
}

#ifdef PNPP_STAMPS
__device__ unsigned long long g_wsd3_stamps[2][2][8];   // [KD == 256][P, C][phase]: s_memtime ticks of pair 0 of workgroup 8
#define WD3_STAMP(i)                                                   \
    if (st_on) {                                                       \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += st_t - st_last;                                   \
        st_last = st_t;                                                \
    }
#else
#define WD3_STAMP(i)
#endif

// LDS layout of both forms (bytes), ONE definition for the kernels and the launcher: weight panel (three planes of [BN][KD] bf16) |
// four pairs x two chunk-image buffers (three planes of [32][64] bf16) | constant table [3][KD] floats | K = 128: four pairs' activation
// fragments [2 steps][3 pieces][64 lanes] x 16 bytes | eight counters
template <int KD, int BN, bool HANDOFF>
struct Wsd3Lds {
    static constexpr int WPLANE = BN * KD * 2, APLANE = 32 * 128, ABUF = 3 * APLANE, APAIR = 2 * ABUF, BFR = 2 * 3 * 64 * 16;
    static constexpr int OFF_IMG = 3 * WPLANE, OFF_CST = OFF_IMG + 4 * APAIR, OFF_BFR = OFF_CST + 3 * KD * 4;
    static constexpr int OFF_FLG = OFF_BFR + (HANDOFF ? 4 * BFR : 0), END = OFF_FLG + 64;
};

template <int KD, int BN, int AM>
__global__ void __launch_bounds__(512, 1)
gemm_wsd3_kernel(const AOperand A, const float *__restrict__ W, int ldw, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int NC = KD / 64, CT = KD / 32;
    static_assert(BN == 32, "one column tile per workgroup: the consumer wave holds dW (KD x 32) and the dA tile");
    constexpr bool SPLITDW = true;                           // the dW tiles of the chunks' second half live in the PRODUCER wave, which multiplies
                                                             // its own image for them (K = 256: dW would not fit one wave; K = 128: balance)
    constexpr int CTC = SPLITDW ? CT / 2 : CT;               // dW tiles (of 32 channels) the consumer wave holds
    constexpr bool HANDOFF = KD == 128;                      // the producer hands the activation fragments over through LDS (K = 256: no LDS
                                                             // left for them: both waves build them)
    constexpr bool DW2 = KD == 128 && AM == A_DZ_POOL;      // a second dW accumulator set (the small products): where registers allow
    constexpr bool ACC2 = true;                              // a second dA accumulator (the small products)
    constexpr int WPITCH = KD * 2, WPLANE = BN * WPITCH;     // bytes: row and plane of the weight panel [n][k]
    constexpr int APLANE = 32 * 128, ABUF = 3 * APLANE;      // bytes: plane and buffer (three planes) of a dZ chunk image
    constexpr int APAIR = 2 * ABUF;                          // two buffers per wave pair
    using LY = Wsd3Lds<KD, BN, HANDOFF>;
    constexpr int BFR = LY::BFR;                             // bytes: the strip's activation fragments, [step][piece][lane] of 16 bytes
    constexpr int OFF_IMG = LY::OFF_IMG, OFF_CST = LY::OFF_CST, OFF_BFR = LY::OFF_BFR, OFF_FLG = LY::OFF_FLG;
    static_assert(NC % 2 == 0, "chunk k of a strip uses buffer k & 1");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    unsigned char *Wp = lds3;
    float *Tc = reinterpret_cast<float *>(lds3 + OFF_CST);   // [3][KD]: g, a, b of dZ = g dY + a Z + b
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    const int pair = wave & 3;
    unsigned char *Ap = lds3 + OFF_IMG + pair * APAIR;
    uint4 *Bf = reinterpret_cast<uint4 *>(lds3 + OFF_BFR + pair * BFR) + lane;   // this lane's slots: Bf[(2 step + piece... ) * 64]
    wd3_flag *f_ready = (wd3_flag *)(lds3 + OFF_FLG) + 2 * pair, *f_done = f_ready + 1;
    const int l31 = lane & 31, lh = lane >> 5;
    auto xs = [](int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); };   // chunk XOR of image row r
    auto xw = [](int n) { return n & 15; };                                  // chunk XOR of panel row n

    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {   // XCD-aware: the column blocks of one worker share an L2
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;
    const int nstrips = M / 32, stride = nworkers * 4;
    int strip = worker * 4 + pair;
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.z), (short)0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t resP = wd3_rsrc(E.zp);
    // accumulator positions of a strip of layer l-1 (z_{l-1} in, dY_{l-1} out): row 4 lh + (r & 3) + 8 (r >> 2), column n0 + l31; the
    // lane part is ONE register, the register part is uniform and rides in the instruction's scalar offset
    const unsigned oq = 4u * ((unsigned)(4 * lh) * (unsigned)Nout + (unsigned)(n0 + l31));
    auto quni = [&](int r) -> unsigned { return 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * Nout); };
    const float e_sc = E.scale[n0 + l31], e_sh = E.shift[n0 + l31];
    // transposed reads [c-tile it of the chunk][step s][block b]: rows 16 s + 8 b + 4 lh + qq, columns 32 it + l31.  Two registers
    // (b = 0, 1): step s adds 16 rows = 2,048 bytes (bit 4 of the row is not in x(r)), c-tile it adds 4 to the group index, which
    // the XOR with x(r) turns into ^ 64 on the byte offset
    unsigned tbase[2];
    {
        const int gi = lane & 15, qq = gi >> 2, pp = gi & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = 8 * b + 4 * lh + qq, ch = 2 * g1 + (pp >> 1);
            tbase[b] = (unsigned)(r * 128 + 16 * (ch ^ xs(r)) + 8 * (pp & 1));
        }
    }
    auto tofs = [&](int it, int s, int b) -> unsigned { return (tbase[b] ^ (unsigned)(it * 64)) + (unsigned)(s * 2048); };
    // relu(bn(z_{l-1})) at this lane's 16 accumulator positions as the dW product's B fragments: eight consecutive registers are one
    // 32x32x16 operand, in three pieces
    auto act_fragments = [&](const float (&z)[16], uint4 (&bfr)[2][3]) {
        float act[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) act[r] = fmaxf(fmaf(z[r], e_sc, e_sh), 0.f);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 v0, v1;
            v0[0] = act[8 * s + 0], v0[1] = act[8 * s + 1], v0[2] = act[8 * s + 2], v0[3] = act[8 * s + 3];
            v1[0] = act[8 * s + 4], v1[1] = act[8 * s + 5], v1[2] = act[8 * s + 6], v1[3] = act[8 * s + 7];
            uint2 h0, m0, l0, h1, m1, l1;
            wd3_split4(v0, h0, m0, l0);
            wd3_split4(v1, h1, m1, l1);
            bfr[s][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            bfr[s][1] = make_uint4(m0.x, m0.y, m1.x, m1.y);
            bfr[s][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    };
    // one dW tile of a chunk: dw (+ dws) += dZ_chunk[:, 32 it ..]^T act, two steps of 16 rows; `last` runs behind the final read
    auto dw_tile = [&](const unsigned char *Ab, int it, const uint4 (&bfr)[2][3], f32x16 &dwl, f32x16 &dwsm, auto last) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 ta[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint2 lo = wd3_tr(Ab + p * APLANE + tofs(it, s, 0)), hi = wd3_tr(Ab + p * APLANE + tofs(it, s, 1));
                ta[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            if (s == 1) last();
            const wd3_bf16x8 ah = wd3_op(ta[0]), am = wd3_op(ta[1]), al = wd3_op(ta[2]);
            const wd3_bf16x8 bh = wd3_op(bfr[s][0]), bm = wd3_op(bfr[s][1]), bl = wd3_op(bfr[s][2]);
            f32x16 d = DW2 ? dwsm : dwl;   // the small products first, the leading one last
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
            if constexpr (DW2) {
                dwsm = d;
                dwl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, dwl, 0, 0, 0);
            } else {
                dwl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
            }
        }
    };

    // ---- prologue, all eight waves: counters, the constant table, the weight panel ----
    if (tid < 8) ((wd3_flag *)(lds3 + OFF_FLG))[tid] = 0u;
    for (int c = tid; c < KD; c += 512) {
        const float g = A.cst[c], mu = A.cst[A.C + c], is = A.cst[2 * A.C + c], c1 = A.cst[3 * A.C + c], c2 = A.cst[4 * A.C + c];
        const float a = -g * is * c2;
        Tc[c] = g, Tc[KD + c] = a, Tc[2 * KD + c] = -g * c1 - a * mu;
    }
    {   // W is (KD x Nout) row-major; image [n][k] of columns n0 .. n0 + 31 in three bf16 planes (lane = column n: four dword loads of
        // consecutive rows k, split, one 8-byte store per plane)
        constexpr int NWF = (KD / 4) * BN / 512;
        f32x4 tw[NWF];
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 512 * j, nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
            for (int e = 0; e < 4; ++e) tw[j][e] = W[(size_t)(k4 + e) * ldw + n0 + nl];
        }
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 512 * j, nl = f % BN, k4 = 4 * (f / BN);
            uint2 h, m, l;
            wd3_split4(tw[j], h, m, l);
            unsigned char *dst = Wp + nl * WPITCH + 16 * ((k4 >> 3) ^ xw(nl)) + 2 * (k4 & 7);
            *reinterpret_cast<uint2 *>(dst) = h;
            *reinterpret_cast<uint2 *>(dst + WPLANE) = m;
            *reinterpret_cast<uint2 *>(dst + 2 * WPLANE) = l;
        }
    }
    __syncthreads();

#ifdef PNPP_STAMPS
    const bool st_on = blockIdx.x == 8 && pair == 0;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    f32x16 dw[CTC], dws[DW2 ? CTC : 1];   // dW tiles of this wave (C: tiles 0 .. CTC - 1; SPLITDW: P holds CTC .. CT - 1): the leading
                                          // products' sums and (DW2) the small products' sums
    double s1 = 0.0, s2 = 0.0;          // C waves
    bool timed_out = false;

    if (producer) {
        // =========================== P: the dZ image, nothing else (vector work only) ===========================
        const int q = lane & 15, q4 = 4 * q, rb = lane >> 4;   // staging map: channels 64 c + 4 q .. + 3, rows rb + 4 i
        const __amdgpu_buffer_rsrc_t resZ = wd3_rsrc(A.z), resY = wd3_rsrc(A.a), resI = wd3_rsrc(A.arg);
        const unsigned oa0 = 4u * ((unsigned)rb * (unsigned)KD + (unsigned)q4);
        f32x4 rz[2][8], ry[AM == A_DZ ? 2 : 1][8], rdm[2];   // two chunk register sets in flight (A_DZ: the dense gradient beside Z)
        int4 rarg[2];
        auto fetch_chunk = [&](bool have, int s, int c) {   // chunk c of strip s into register set c & 1
            const __amdgpu_buffer_rsrc_t rZ = have ? resZ : resNull, rY = have ? resY : resNull, rI = have ? resI : resNull;
            const unsigned so = (unsigned)s * (32u * KD * 4u) + 256u * (unsigned)c;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rz[c & 1][i] = wd3_load4(rZ, oa0, so + (unsigned)i * (4u * KD * 4u));
                if constexpr (AM == A_DZ) ry[c & 1][i] = wd3_load4(rY, oa0, so + (unsigned)i * (4u * KD * 4u));
            }
            if constexpr (AM == A_DZ_POOL) {
                const unsigned sg = (unsigned)s * (KD * 4u) + 256u * (unsigned)c;   // one row of the pooled tables per strip
                rdm[c & 1] = wd3_load4(rY, 4u * (unsigned)q4, sg);
                rarg[c & 1] = __builtin_bit_cast(int4, wd3_load4(rI, 4u * (unsigned)q4, sg));
            }
        };
        float zn[16];   // z_{l-1} of the next strip at this lane's accumulator positions (the consumer's dW operand is built here)
        {
            const bool have = strip < nstrips;
            fetch_chunk(have, strip, 0);
            fetch_chunk(have, strip, 1);
            const __amdgpu_buffer_rsrc_t rP = have ? resP : resNull;
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[r] = wd3_load1(rP, oq, (unsigned)strip * (32u * (unsigned)Nout * 4u) + quni(r));
        }
        // LDS offsets of this lane inside a chunk-image plane: row rb + 4 i: x(r) = 4 bit1(rb) + (i & 3)
        unsigned wofs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wofs[i] = (unsigned)(rb * 128 + 16 * ((q >> 1) ^ (((rb >> 1) & 1) << 2 | i)) + 8 * (q & 1));
        if constexpr (SPLITDW) {
#pragma unroll
            for (int i = 0; i < CTC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
            if constexpr (DW2) {
#pragma unroll
                for (int i = 0; i < CTC; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dws[i][r] = 0.f;
            }
        }
        unsigned kbase = 0;   // chunks of the strips before this one
        for (; strip < nstrips; strip += stride, kbase += NC) {
            const bool more = strip + stride < nstrips;
            const int snext = strip + stride;
            // ---- relu(bn(z_{l-1})) of the strip as the dW product's B fragments.  HANDOFF: written to LDS ahead of chunk 0 for the partner
            // (it read the previous strip's before it finished the chunk this wait is for -- the one chunk 0's buffer waits for, too);
            // SPLITDW: kept, this wave multiplies with them itself ----
            uint4 bfr[2][3];
            act_fragments(zn, bfr);
            {
                const __amdgpu_buffer_rsrc_t nP = more ? resP : resNull;
                const unsigned sn_off = (unsigned)snext * (32u * (unsigned)Nout * 4u);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = wd3_load1(nP, oq, sn_off + quni(r));
            }
            if constexpr (HANDOFF) {
                if (kbase >= 2) timed_out = timed_out || !wd3_wait(f_done, kbase - 1);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p3 = 0; p3 < 3; ++p3) Bf[(s * 3 + p3) * 64] = bfr[s][p3];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                // ---- stage chunk c (sequence number kbase + c) into buffer c & 1: dZ in registers, split, three 8-byte stores per group ----
                unsigned char *Ab = Ap + (c & 1) * ABUF;
                const float4 cg = *reinterpret_cast<const float4 *>(Tc + 64 * c + q4), ca = *reinterpret_cast<const float4 *>(Tc + KD + 64 * c + q4);
                const float4 cb = *reinterpret_cast<const float4 *>(Tc + 2 * KD + 64 * c + q4);
                f32x4 dm = {0.f, 0.f, 0.f, 0.f};
                int4 ar = make_int4(-1, -1, -1, -1);
                if constexpr (AM == A_DZ_POOL) dm = rdm[c & 1], ar = rarg[c & 1];
                float4 bt;   // A_DZ_POOL: b + g dm, what the arg-max row of a channel starts from
                bt.x = fmaf(cg.x, dm[0], cb.x), bt.y = fmaf(cg.y, dm[1], cb.y), bt.z = fmaf(cg.z, dm[2], cb.z), bt.w = fmaf(cg.w, dm[3], cb.w);
                // the dZ elements of one 16-byte group of the chunk
                auto dz4 = [&](int i) -> f32x4 {
                    const f32x4 z = rz[c & 1][i];
                    const int r = rb + 4 * i;
                    f32x4 v;
                    if constexpr (AM == A_DZ) {
                        const f32x4 dy = ry[c & 1][i];
                        v[0] = fmaf(cg.x, dy[0], fmaf(ca.x, z[0], cb.x)), v[1] = fmaf(cg.y, dy[1], fmaf(ca.y, z[1], cb.y));
                        v[2] = fmaf(cg.z, dy[2], fmaf(ca.z, z[2], cb.z)), v[3] = fmaf(cg.w, dy[3], fmaf(ca.w, z[3], cb.w));
                    } else {
                        v[0] = fmaf(ca.x, z[0], r == ar.x ? bt.x : cb.x), v[1] = fmaf(ca.y, z[1], r == ar.y ? bt.y : cb.y);
                        v[2] = fmaf(ca.z, z[2], r == ar.z ? bt.z : cb.z), v[3] = fmaf(ca.w, z[3], r == ar.w ? bt.w : cb.w);
                    }
                    return v;
                };
                if constexpr (AM == A_DZ_POOL && !SPLITDW) {   // the split goes ahead of the wait for the buffer (48 registers the other forms lack)
                    uint2 ph[8], pm[8], pl[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) wd3_split4(dz4(i), ph[i], pm[i], pl[i]);
                    WD3_STAMP(0)   // dZ and its pieces
                    // this register set is free: the chunk two ahead goes out
                    if (c + 2 < NC) fetch_chunk(true, strip, c + 2);
                    else fetch_chunk(more, snext, c + 2 - NC);
                    if (kbase + c >= 2) timed_out = timed_out || !wd3_wait(f_done, kbase + c - 1);   // the partner has finished with this buffer
                    WD3_STAMP(1)   // wait for the buffer
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        unsigned char *dst = Ab + wofs[i & 3] + i * 512;
                        *reinterpret_cast<uint2 *>(dst) = ph[i];
                        *reinterpret_cast<uint2 *>(dst + APLANE) = pm[i];
                        *reinterpret_cast<uint2 *>(dst + 2 * APLANE) = pl[i];
                    }
                } else {
                    if (kbase + c >= 2) timed_out = timed_out || !wd3_wait(f_done, kbase + c - 1);   // the partner has finished with this buffer
                    WD3_STAMP(1)   // wait for the buffer
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        uint2 h, m, l;
                        wd3_split4(dz4(i), h, m, l);
                        unsigned char *dst = Ab + wofs[i & 3] + i * 512;
                        *reinterpret_cast<uint2 *>(dst) = h;
                        *reinterpret_cast<uint2 *>(dst + APLANE) = m;
                        *reinterpret_cast<uint2 *>(dst + 2 * APLANE) = l;
                    }
                    WD3_STAMP(0)   // dZ, its pieces, the image
                    if (c + 2 < NC) fetch_chunk(true, strip, c + 2);
                    else fetch_chunk(more, snext, c + 2 - NC);
                }
                wd3_post(f_ready, kbase + c + 1);
                WD3_STAMP(2)   // image written and published
                if constexpr (SPLITDW) {   // this wave's share of dW: the chunks of the second half, from its own image (LDS is in order per wave)
                    if (c >= NC / 2) {
#pragma unroll
                        for (int it = 0; it < 2; ++it) dw_tile(Ab, it, bfr, dw[2 * c + it - CTC], dws[DW2 ? 2 * c + it - CTC : 0], [] {});
                    }
                }
            }
        }
    } else {
        // =========================== C: both products and the epilogue (matrix work) ===========================
#pragma unroll
        for (int i = 0; i < CTC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
        if constexpr (DW2) {
#pragma unroll
            for (int i = 0; i < CTC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dws[i][r] = 0.f;
        }
        const __amdgpu_buffer_rsrc_t resC = wd3_rsrc(E.c);
        const float e_mu = E.mu[n0 + l31], e_is = E.istd[n0 + l31];
        float zq[16];   // z_{l-1} of the strip: requested at its start, used in its epilogue (mask, sum v z)
        const unsigned arow = (unsigned)(l31 * 128);
        const int ax = xs(l31);
        const unsigned char *brow = Wp + l31 * WPITCH;
        const int bx = xw(l31);
        float zn[16];   // (!HANDOFF) the next strip's z_{l-1}: this wave builds its fragments itself
        if constexpr (!HANDOFF) {
            const __amdgpu_buffer_rsrc_t rP = strip < nstrips ? resP : resNull;
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[r] = wd3_load1(rP, oq, (unsigned)strip * (32u * (unsigned)Nout * 4u) + quni(r));
        }
        unsigned kbase = 0;
        for (; strip < nstrips; strip += stride, kbase += NC) {
            const bool more = strip + stride < nstrips;
            uint4 bfr[2][3];   // [step][piece]: relu(bn(z_{l-1})) of the strip (HANDOFF: built by the partner, read behind the wait for chunk 0)
            if constexpr (HANDOFF) {   // z_{l-1} is only needed in the epilogue: requested now
                const unsigned sq_off = (unsigned)strip * (32u * (unsigned)Nout * 4u);
#pragma unroll
                for (int r = 0; r < 16; ++r) zq[r] = wd3_load1(resP, oq, sq_off + quni(r));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zq[r] = zn[r];
                const __amdgpu_buffer_rsrc_t nP = more ? resP : resNull;
                const unsigned sn_off = (unsigned)(strip + stride) * (32u * (unsigned)Nout * 4u);
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = wd3_load1(nP, oq, sn_off + quni(r));
                act_fragments(zq, bfr);
            }
            f32x16 acc, accs;   // dA: the leading products and (ACC2) the small ones, added in the epilogue
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (ACC2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[r] = 0.f;
            }
            WD3_STAMP(3)   // activation fragments
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const unsigned char *Ab = Ap + (c & 1) * ABUF;
                timed_out = timed_out || !wd3_wait(f_ready, kbase + c + 1);
                if constexpr (HANDOFF) {
                    if (c == 0) {
#pragma unroll
                        for (int s = 0; s < 2; ++s)
#pragma unroll
                            for (int p3 = 0; p3 < 3; ++p3) bfr[s][p3] = Bf[(s * 3 + p3) * 64];
                    }
                }
                WD3_STAMP(4)   // wait for the chunk
                // ---- dA += dZ_chunk W_chunk: step t covers k = 64 c + 16 t + 8 lh + (0 .. 7) ----
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int g = 2 * t + lh;
                    uint4 fa[3], fb[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        fa[p] = *reinterpret_cast<const uint4 *>(Ab + p * APLANE + arow + 16 * (g ^ ax));
                        fb[p] = *reinterpret_cast<const uint4 *>(brow + p * WPLANE + 16 * ((8 * c + g) ^ bx));
                    }
                    const wd3_bf16x8 ah = wd3_op(fa[0]), am = wd3_op(fa[1]), al = wd3_op(fa[2]);
                    const wd3_bf16x8 bh = wd3_op(fb[0]), bm = wd3_op(fb[1]), bl = wd3_op(fb[2]);
                    f32x16 d = ACC2 ? accs : acc;   // the small products first, the leading one last
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
                    if constexpr (ACC2) {
                        accs = d;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
                    }
                }
                WD3_STAMP(5)   // dA product of the chunk
                // ---- dW[64 c + 32 it + .][n] += dZ_chunk^T act (SPLITDW: the chunks of the first half; the partner has the others); the
                // buffer is released behind this wave's last read of it ----
                if (!SPLITDW || c < NC / 2) {
                    dw_tile(Ab, 0, bfr, dw[2 * c < CTC ? 2 * c : 0], dws[DW2 ? (2 * c < CTC ? 2 * c : 0) : 0], [] {});
                    dw_tile(Ab, 1, bfr, dw[2 * c + 1 < CTC ? 2 * c + 1 : 0], dws[DW2 ? (2 * c + 1 < CTC ? 2 * c + 1 : 0) : 0],
                            [&] { wd3_post(f_done, kbase + c + 1); });
                } else {
                    wd3_post(f_done, kbase + c + 1);   // (behind the dA product's reads: wd3_post waits for them)
                }
                WD3_STAMP(6)   // transposed reads + dW products of the chunk
            }
            // ---- epilogue: mask, store, BatchNorm-backward sums of layer l-1 ----
            {
                const unsigned sc_off = (unsigned)strip * (32u * (unsigned)Nout * 4u);
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaf(zq[r], e_sc, e_sh) > 0.f ? (ACC2 ? acc[r] + accs[r] : acc[r]) : 0.f;   // (the mask is recomputed: a bit per row built
                                                                                                           //  at the strip's start was a chain of 48 dependent instructions)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), resC, (int)oq, (int)(sc_off + quni(r)), 0);
                    t1 += v;
                    t2 = fmaf(v, zq[r], t2);
                }
                const double d1 = (double)t1;   // sum v xhat = istd (sum v z - mu sum v), finished in float64
                s1 += d1, s2 += (double)e_is * ((double)t2 - (double)e_mu * d1);
            }
            WD3_STAMP(7)   // epilogue
        }
    }
#ifdef PNPP_STAMPS
    if (st_on && lane == 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) g_wsd3_stamps[KD == 256][producer ? 0 : 1][i] += st_acc[i];
#endif
    if (timed_out && lane == 0) atomicExch(&g_wsd3_timeouts, 1);

    // ---- tails: one dW partial per workgroup (every tile's four copies -- one per pair -- added through LDS), the column statistics ----
    __syncthreads();   // every wave is done with the panel and the images
    f32x4 *red = reinterpret_cast<f32x4 *>(lds3);                          // [tile][pair][r4][lane]
    double *dred = reinterpret_cast<double *>(lds3 + CT * 4 * 4 * 64 * 16);   // [pair][2][BN]
    if (!producer || SPLITDW) {   // C: tiles 0 .. CTC - 1; P (SPLITDW): tiles CTC .. CT - 1
        const int t0 = producer ? CTC : 0;
#pragma unroll
        for (int i = 0; i < CTC; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = DW2 ? dw[i][4 * r4 + e] + dws[DW2 ? i : 0][4 * r4 + e] : dw[i][4 * r4 + e];
                red[(((t0 + i) * 4 + pair) * 4 + r4) * 64 + lane] = v;
            }
    }
    if (!producer) {
        const double a = s1 + shfl_xor_f64(s1, 32), b = s2 + shfl_xor_f64(s2, 32);
        if (lh == 0) dred[(pair * 2 + 0) * BN + l31] = a, dred[(pair * 2 + 1) * BN + l31] = b;
    }
    __syncthreads();
    {
        float *wb = E.dwslab + (size_t)worker * KD * E.dw_ld + n0;
        for (int t = wave; t < CT; t += 8) {   // c-tile t
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 a0 = red[((t * 4 + 0) * 4 + r4) * 64 + lane], a1 = red[((t * 4 + 1) * 4 + r4) * 64 + lane];
                const f32x4 a2 = red[((t * 4 + 2) * 4 + r4) * 64 + lane], a3 = red[((t * 4 + 3) * 4 + r4) * 64 + lane];
                const int c0 = t * 32 + 8 * r4 + 4 * lh;
                float *o = wb + (size_t)c0 * E.dw_ld + l31;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[(size_t)e * E.dw_ld] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
            }
        }
        if (tid < 2 * BN) {
            const int which = tid / BN, cl = tid % BN;
            const double t = (dred[(0 * 2 + which) * BN + cl] + dred[(1 * 2 + which) * BN + cl]) +
                             (dred[(2 * 2 + which) * BN + cl] + dred[(3 * 2 + which) * BN + cl]);
            E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
        }
    }
}

template <int KD, int BN, int AM>
static void wsd3_launch(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int workers, int ncol, hipStream_t st) {
    // (a first version with the hand-off sized this by hand and forgot the fragments: its counters lay behind the end of the allocation, LDS
    //  drops such writes without a fault, and every poll ran into its bound)
    constexpr size_t main_b = Wsd3Lds<KD, BN, KD != 256>::END;   // (the hand-off area exists for K = 128 only)
    constexpr size_t red_b = (size_t)(KD / 32) * (BN / 32) * 4 * 4 * 64 * 16 + (size_t)4 * 2 * BN * 8;
    constexpr size_t lds = main_b > red_b ? main_b : red_b;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kfn = gemm_wsd3_kernel<KD, BN, AM>;
    static bool granted = false;
    if (!granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(512), lds, st, A, B.b, B.ldb, M, Nout, ncol, E);
}

bool wsd3_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E) {
    if (!split_products() || matmul_precision() != 0) return false;
    // the last layer of a grouped level (pooled gradient): 128 -> 64 (SA1), 256 -> 128 (SA2); a middle layer with a dense gradient: 128 -> 128
    const bool pooled = A.mode == A_DZ_POOL && A.K == 32 && ((Kd == 128 && Nout == 64) || (Kd == 256 && Nout == 128));
    const bool dense = A.mode == A_DZ && Kd == 128 && Nout == 128 && A.a;
    if (M < 8192 || M % 32 != 0 || !(pooled || dense)) return false;
    if (E.mode != E_MASK_STATS || !E.dwslab || E.dw_ld < Nout) return false;
    if (B.trans || B.perm_D >= 0 || (B.rows > 0 && B.rows != Kd) || B.ldb < Nout) return false;
    if (A.lda != Kd || A.C != Kd || E.ldc != Nout) return false;
    if ((((uintptr_t)A.a | (uintptr_t)A.z | (uintptr_t)E.zp | (uintptr_t)E.c | (uintptr_t)A.cst) & 15) != 0) return false;
    if (A.mode == A_DZ_POOL && (((uintptr_t)A.arg) & 15) != 0) return false;
    if ((unsigned long long)M * (unsigned)Kd * 4ull >= 0xfffffff0ull) return false;   // 32-bit buffer offsets
    return true;
}

bool try_launch_wsd3(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                     int *dw_slabs) {
    *rc = PNPP_OK;
    if (!dw_slabs || !wsd3_applies(A, B, M, Nout, Kd, E)) return false;
    const int nstrips = M / 32, ncol = Nout / 32;
    int workers = 256 / ncol;   // one workgroup of eight waves per CU
    if (workers * 4 > nstrips) workers = (nstrips + 3) / 4;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (nslab) *nslab = workers;
    *dw_slabs = workers;
    ProfScope ps(st, "gemm_wsd3_kernel<%d,32,A%d> M=%d N=%d K=%d grid=%dx1", Kd, A.mode, M, Nout, Kd, workers * ncol);
    if (Kd == 256) wsd3_launch<256, 32, A_DZ_POOL>(A, B, M, Nout, E, workers, ncol, st);
    else if (A.mode == A_DZ_POOL) wsd3_launch<128, 32, A_DZ_POOL>(A, B, M, Nout, E, workers, ncol, st);
    else wsd3_launch<128, 32, A_DZ>(A, B, M, Nout, E, workers, ncol, st);
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsd3: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

#ifdef PNPP_STAMPS
#define WD3_STAMPS_BIT 64u
#else
#define WD3_STAMPS_BIT 0u
#endif
#ifdef WD3_PRIO
#define WD3_PRIO_BIT 256u
#else
#define WD3_PRIO_BIT 0u
#endif
unsigned wsd3_build_flags() { return WD3_STAMPS_BIT | WD3_PRIO_BIT; }

int wsd3_timeouts() {
    int v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wsd3_timeouts), sizeof(int));
    return v;
}

}  // namespace pnpp

extern "C" int pnpp_debug_wsd3_timeouts(void) { return pnpp::wsd3_timeouts(); }
#ifdef PNPP_STAMPS
extern "C" int pnpp_debug_wsd3_stamps(unsigned long long *out32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_wsd3_stamps), z, sizeof(z));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out32, HIP_SYMBOL(pnpp::g_wsd3_stamps), 32 * sizeof(unsigned long long));
    }
    return 0;
}
#endif
