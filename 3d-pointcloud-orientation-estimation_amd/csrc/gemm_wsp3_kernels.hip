// gemm_wsp3_kernels.hip -- the fused backward product of a grouped layer with 128 output channels (dA + ReLU mask + BatchNorm-backward
// sums + dW in one launch) on wave-private row strips, with the float32 products formed on the bf16 matrix pipe from exact three-way
// operand splits (gemm_wsf3_kernels.hip has the arithmetic: six bf16 x bf16 products per float32 product, float32 accumulation).
//
// Reference: the autograd backward of conv -> BatchNorm -> ReLU (models/pointnet_pp_8dir.py:40-42) for layer l of a grouped level:
//   dZ_l     = BatchNorm-backward(dY_l, Z_l)              (A_DZ: dY_l dense; A_DZ_POOL: dY_l rebuilt from the pooled gradient and the
//                                                          arg-max rows of the level's max over 32 neighbours)
//   dY_{l-1} = (dZ_l W_l) masked by ReLU'(layer l-1), + its BatchNorm-backward column sums
//   dW_l     = dZ_l^T relu(bn(Z_{l-1}))
//
// gemm_wsp_kernel (float32 MFMA) is built around "every VALU instruction is matrix time lost": it folds the BatchNorm-backward
// transform into the weight panel and patches the one-hot gradient into LDS.  A bf16 MFMA holds the vector issue for 8 of its 32
// cycles only, and the six of them that make a float32 product take 192 cycles where the float32 instructions take 512: here the
// vector work is cheap and the matrix work short, so dZ is simply computed (one compare + select + FMA per element), split, and
// written to the wave's own LDS image; what is left of the old design is the wave-private strip (no barrier in the strip loop), the
// operand streams one strip ahead in registers, and the dW reduction in the order of the dA accumulator layout:
//   * dA:  A fragments = rows of the dZ image (ds_read_b128), B fragments = rows of the [n][k] panel of W_l, split once per workgroup
//   * dW:  B fragments = relu(bn(z_{l-1})) straight from REGISTERS -- z_{l-1} is loaded in accumulator layout (lane = column,
//          registers = rows 4 lh + (r & 3) + 8 (r >> 2)), and eight consecutive registers are one 32x32x16 operand whose reduction
//          index j stands for row 16 s + 8 (j >> 2) + 4 lh + (j & 3); A fragments = the dZ image read TRANSPOSED by ds_read_b64_tr_b16
//          in exactly that row order (two 4-row blocks per fragment), so no second image and no activation image exist
//   * the dZ image is [32 rows][64 channels] bf16 per 64-channel chunk (128-byte rows, three planes, two buffers): 16-byte group g of
//     row r sits at g ^ x(r), x(r) = 4 bit1(r) + bits3:2(r) -- conflict-free for the row reads (the four 16-lane groups of a
//     ds_read_b128 see eight different x per row parity) and for the transposed reads (rows r and r + 2 of a block differ in x's bit 2)
// Shapes: K = C_l = 128, N = C_{l-1} in {64, 128} (64 output columns per workgroup), M a multiple of 32.
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

typedef __bf16 wp3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wp3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wp3_f32x2 __attribute__((ext_vector_type(2)));
typedef short wp3_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned wp3_pk(float lo, float hi) {
    const wp3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wp3_bf16x2));
}
__device__ __forceinline__ float wp3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float wp3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void wp3_split4(const f32x4 v, uint2 &h, uint2 &m, uint2 &l) {
    h.x = wp3_pk(v[0], v[1]), h.y = wp3_pk(v[2], v[3]);
    float r0 = v[0] - wp3_lo(h.x), r1 = v[1] - wp3_hi(h.x), r2 = v[2] - wp3_lo(h.y), r3 = v[3] - wp3_hi(h.y);
    m.x = wp3_pk(r0, r1), m.y = wp3_pk(r2, r3);
    r0 -= wp3_lo(m.x), r1 -= wp3_hi(m.x), r2 -= wp3_lo(m.y), r3 -= wp3_hi(m.y);
    l.x = wp3_pk(r0, r1), l.y = wp3_pk(r2, r3);
}
__device__ __forceinline__ wp3_bf16x8 wp3_op(uint4 v) { return __builtin_bit_cast(wp3_bf16x8, v); }
__device__ __forceinline__ uint2 wp3_tr(const unsigned char *p) {   // ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes, transposed
    const wp3_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp3_s16x4 *)(p));
    return __builtin_bit_cast(uint2, v);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wp3_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wp3_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}
__device__ __forceinline__ float wp3_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}

// KD = C_l = 128; AM = A_DZ or A_DZ_POOL; 64 output columns (of Nout) per workgroup
template <int KD, int AM>
__global__ void __launch_bounds__(256, 1)
gemm_wsp3_kernel(const AOperand A, const float *__restrict__ W, int ldw, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int BN = 64, NC = KD / 64, CT = KD / 32;
    constexpr int WPITCH = KD * 2, WPLANE = BN * WPITCH;     // bytes: row and plane of the weight panel [n][k]
    constexpr int APLANE = 32 * 128, ABUF = 3 * APLANE;      // bytes: plane and buffer (three planes) of a dZ chunk image
    constexpr int AWAVE = 2 * ABUF;                          // two buffers per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    unsigned char *Wp = lds3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *Ap = lds3 + 3 * WPLANE + wave * AWAVE;
    const int l31 = lane & 31, lh = lane >> 5;
    auto xs = [](int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); };   // chunk XOR of image row r
    auto xw = [](int n) { return n & 15; };                                  // chunk XOR of panel row n (256-byte rows)

    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {   // XCD-aware: the column blocks of one worker share an L2
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;

    // staging map of a strip: channels 64 c + 4 q .. + 3, rows rb + 4 i
    const int q = lane & 15, q4 = 4 * q, rb = lane >> 4;
    const int nstrips = M / 32, stride = nworkers * 4;
    int strip = worker * 4 + wave;
    const __amdgpu_buffer_rsrc_t resZ = wp3_rsrc(A.z), resY = wp3_rsrc(A.a), resP = wp3_rsrc(E.zp), resI = wp3_rsrc(A.arg);
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.z), (short)0, 0, 0x00020000);
    const unsigned oa0 = 4u * ((unsigned)rb * (unsigned)KD + (unsigned)q4);   // rows rb + 4 i: the i part rides in the scalar offset
    // accumulator positions of a strip of layer l-1 (z_{l-1} in, dY_{l-1} out): row 4 lh + (r & 3) + 8 (r >> 2), column n0 + 32 j + l31
    const unsigned oq = 4u * ((unsigned)(4 * lh) * (unsigned)Nout + (unsigned)(n0 + l31));
    // (the lane part `oq` is ONE register; the (column tile, register) part is uniform and rides in the instruction's scalar offset --
    //  32 per-position address registers were what this kernel spilled, and a spilled address serialises the loads behind vmcnt(0))
    auto quni = [&](int j, int r) -> unsigned { return 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * Nout + 32 * j); };

    // operand streams: registers, one strip ahead
    f32x4 rz[NC][8], ry[AM == A_DZ ? NC : 1][8];
    f32x4 rdm[NC];
    int4 rarg[NC];
    float zq[2][16], zn[2][16];
    auto fetch_chunk = [&](bool have, int s, int c) {
        const __amdgpu_buffer_rsrc_t rZ = have ? resZ : resNull, rY = have ? resY : resNull, rI = have ? resI : resNull;
        const unsigned so = (unsigned)s * (32u * KD * 4u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rz[c][i] = wp3_load4(rZ, oa0, so + 256u * (unsigned)c + (unsigned)i * (4u * KD * 4u));
            if constexpr (AM == A_DZ) ry[c][i] = wp3_load4(rY, oa0, so + 256u * (unsigned)c + (unsigned)i * (4u * KD * 4u));
        }
        if constexpr (AM == A_DZ_POOL) {   // one row of the pooled tables per strip (nsample = 32: the strip is the group)
            const unsigned sg = (unsigned)s * (KD * 4u);
            rdm[c] = wp3_load4(rY, 4u * (unsigned)q4 + 256u * (unsigned)c, sg);
            rarg[c] = __builtin_bit_cast(int4, wp3_load4(rI, 4u * (unsigned)q4 + 256u * (unsigned)c, sg));
        }
    };
    auto fetch_p = [&](bool have, int s) {
        const __amdgpu_buffer_rsrc_t rP = have ? resP : resNull;
        const unsigned so = (unsigned)s * (32u * (unsigned)Nout * 4u);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) zn[j][r] = wp3_load1(rP, oq, so + quni(j, r));
    };
    {
        const bool have = strip < nstrips;
#pragma unroll
        for (int c = 0; c < NC; ++c) fetch_chunk(have, strip, c);
        fetch_p(have, strip);
    }

    // per-channel constants of this lane's channel groups: dZ = g dY + a Z + b,  a = -g istd c2,  b = -g c1 - a mu
    float4 cg[NC], ca[NC], cb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float *p = A.cst + 64 * c + q4;
        const float4 g = *reinterpret_cast<const float4 *>(p), mu = *reinterpret_cast<const float4 *>(p + A.C);
        const float4 is = *reinterpret_cast<const float4 *>(p + 2 * A.C), c1 = *reinterpret_cast<const float4 *>(p + 3 * A.C);
        const float4 c2 = *reinterpret_cast<const float4 *>(p + 4 * A.C);
        cg[c] = g;
        ca[c] = make_float4(-g.x * is.x * c2.x, -g.y * is.y * c2.y, -g.z * is.z * c2.z, -g.w * is.w * c2.w);
        cb[c] = make_float4(-g.x * c1.x - ca[c].x * mu.x, -g.y * c1.y - ca[c].y * mu.y, -g.z * c1.z - ca[c].z * mu.z,
                            -g.w * c1.w - ca[c].w * mu.w);
    }
    // epilogue constants of this lane's two output columns
    float e_sc[2], e_sh[2];
    double e_mu[2], e_is[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + l31;
        e_sc[j] = E.scale[col], e_sh[j] = E.shift[col], e_mu[j] = (double)E.mu[col], e_is[j] = (double)E.istd[col];
    }
    // weight panel: W is (KD x Nout) row-major; image [n][k] of columns n0 .. n0 + 63 in three bf16 planes (lane = column n: four dword
    // loads of consecutive rows k, split, one 8-byte store per plane)
    {
        constexpr int NWF = (KD / 4) * BN / 256;
        f32x4 tw[NWF];
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
            for (int e = 0; e < 4; ++e) tw[j][e] = W[(size_t)(k4 + e) * ldw + n0 + nl];
        }
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
            uint2 h, m, l;
            wp3_split4(tw[j], h, m, l);
            unsigned char *dst = Wp + nl * WPITCH + 16 * ((k4 >> 3) ^ xw(nl)) + 2 * (k4 & 7);
            *reinterpret_cast<uint2 *>(dst) = h;
            *reinterpret_cast<uint2 *>(dst + WPLANE) = m;
            *reinterpret_cast<uint2 *>(dst + 2 * WPLANE) = l;
        }
    }

    f32x16 dw[CT][2];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[i][j][r] = 0.f;
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    __syncthreads();   // the panel is complete; from here on the waves run on their own

    // LDS offsets of this lane inside a chunk-image plane
    unsigned wofs[4];   // staging: row rb + 4 i (x(r) = 4 bit1(rb) + (i & 3)), 8 bytes at channel q4; + 512 i
#pragma unroll
    for (int i = 0; i < 4; ++i) wofs[i] = (unsigned)(rb * 128 + 16 * ((q >> 1) ^ (((rb >> 1) & 1) << 2 | i)) + 8 * (q & 1));
    const unsigned arow = (unsigned)(l31 * 128);   // row reads: row l31, group g -> 16 (g ^ ax)
    const int ax = xs(l31);
    unsigned tofs[2][2][2];   // transposed reads [c-tile of the chunk][step s][block]: rows 16 s + 8 blk + 4 lh + qq, columns 32 it + l31
    {
        const int gi = lane & 15, qq = gi >> 2, pp = gi & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int r = 16 * s + 8 * b + 4 * lh + qq, ch = 4 * it + 2 * g1 + (pp >> 1);
                    tofs[it][s][b] = (unsigned)(r * 128 + 16 * (ch ^ xs(r)) + 8 * (pp & 1));
                }
    }
    const unsigned char *brow[2];
    int bx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + l31;
        brow[j] = Wp + n * WPITCH;
        bx[j] = xw(n);
    }

    for (; strip < nstrips; strip += stride) {
        const bool more = strip + stride < nstrips;
        const int snext = strip + stride;
        // ---- layer l-1's activations of this strip: mask bits, and the dW product's B fragments straight from the registers ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) zq[j][r] = zn[j][r];
        fetch_p(more, snext);
        uint4 bfr[2][2][3];   // [column tile][step][piece]
        unsigned maskb[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float act[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = fmaf(zq[j][r], e_sc[j], e_sh[j]);
                maskb[j] |= (a0 > 0.f ? 1u : 0u) << r;
                act[r] = fmaxf(a0, 0.f);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 v0, v1;
                v0[0] = act[8 * s + 0], v0[1] = act[8 * s + 1], v0[2] = act[8 * s + 2], v0[3] = act[8 * s + 3];
                v1[0] = act[8 * s + 4], v1[1] = act[8 * s + 5], v1[2] = act[8 * s + 6], v1[3] = act[8 * s + 7];
                uint2 h0, m0, l0, h1, m1, l1;
                wp3_split4(v0, h0, m0, l0);
                wp3_split4(v1, h1, m1, l1);
                bfr[j][s][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                bfr[j][s][1] = make_uint4(m0.x, m0.y, m1.x, m1.y);
                bfr[j][s][2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }

        // the five small products go to an accumulator of their own where the registers allow it (the dense-gradient form streams
        // twice the operands one strip ahead); in one accumulator the sum is rounded six times per 16 reduction steps instead of once --
        // the float32 MFMA form rounds eight times
        constexpr int NACC = AM == A_DZ_POOL ? 2 : 1;
        f32x16 acc[2], accl[NACC == 2 ? 2 : 1];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if constexpr (NACC == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accl[j][r] = 0.f;
        }

#pragma unroll
        for (int c = 0; c < NC; ++c) {
            unsigned char *Ab = Ap + (c & 1) * ABUF;
            // ---- stage chunk c: dZ in registers, split, three 8-byte stores per group ----
            float4 bt = cb[c];   // b + g dm at the arg-max row (A_DZ_POOL)
            if constexpr (AM == A_DZ_POOL) {
                bt.x = fmaf(cg[c].x, rdm[c][0], cb[c].x), bt.y = fmaf(cg[c].y, rdm[c][1], cb[c].y);
                bt.z = fmaf(cg[c].z, rdm[c][2], cb[c].z), bt.w = fmaf(cg[c].w, rdm[c][3], cb[c].w);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 z = rz[c][i];
                f32x4 v;
                if constexpr (AM == A_DZ) {
                    const f32x4 dy = ry[c][i];
                    v[0] = fmaf(cg[c].x, dy[0], fmaf(ca[c].x, z[0], cb[c].x)), v[1] = fmaf(cg[c].y, dy[1], fmaf(ca[c].y, z[1], cb[c].y));
                    v[2] = fmaf(cg[c].z, dy[2], fmaf(ca[c].z, z[2], cb[c].z)), v[3] = fmaf(cg[c].w, dy[3], fmaf(ca[c].w, z[3], cb[c].w));
                } else {
                    const int r = rb + 4 * i;
                    v[0] = fmaf(ca[c].x, z[0], r == rarg[c].x ? bt.x : cb[c].x), v[1] = fmaf(ca[c].y, z[1], r == rarg[c].y ? bt.y : cb[c].y);
                    v[2] = fmaf(ca[c].z, z[2], r == rarg[c].z ? bt.z : cb[c].z), v[3] = fmaf(ca[c].w, z[3], r == rarg[c].w ? bt.w : cb[c].w);
                }
                uint2 h, m, l;
                wp3_split4(v, h, m, l);
                unsigned char *dst = Ab + wofs[i & 3] + i * 512;
                *reinterpret_cast<uint2 *>(dst) = h;
                *reinterpret_cast<uint2 *>(dst + APLANE) = m;
                *reinterpret_cast<uint2 *>(dst + 2 * APLANE) = l;
            }
            fetch_chunk(more, snext, c);   // this chunk's registers are free: the next strip's stream goes out

            // ---- dA += dZ_chunk W_chunk: step t covers k = 64 c + 16 t + 8 lh + (0 .. 7) ----
            {
                uint4 fa[NACC][3], fb[NACC][2][3];
                auto ld = [&](int buf, int t) {
                    const int g = 2 * t + lh;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fa[buf][p] = *reinterpret_cast<const uint4 *>(Ab + p * APLANE + arow + 16 * (g ^ ax));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            fb[buf][j][p] = *reinterpret_cast<const uint4 *>(brow[j] + p * WPLANE + 16 * ((8 * c + g) ^ bx[j]));
                };
                auto mm = [&](int buf) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const wp3_bf16x8 ah = wp3_op(fa[buf][0]), am = wp3_op(fa[buf][1]), al = wp3_op(fa[buf][2]);
                        const wp3_bf16x8 bh = wp3_op(fb[buf][j][0]), bm = wp3_op(fb[buf][j][1]), bl = wp3_op(fb[buf][j][2]);
                        f32x16 d = NACC == 2 ? accl[NACC == 2 ? j : 0] : acc[j];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
                        if constexpr (NACC == 2) {
                            accl[NACC == 2 ? j : 0] = d;
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                        } else {
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
                        }
                    }
                };
                if constexpr (NACC == 2) {
                    ld(0, 0);
#pragma unroll
                    for (int t = 0; t < 4; t += 2) {
                        ld(1, t + 1);
                        mm(0);
                        if (t + 2 < 4) ld(0, t + 2);
                        mm(1);
                    }
                } else {   // one fragment set: the dense-gradient form has no registers for a second
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        ld(0, t);
                        mm(0);
                    }
                }
            }
            // ---- dW[64 c + 32 it + .][n] += dZ_chunk^T act: two steps of 16 rows; the small products first, the leading one last ----
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint4 ta[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint2 lo = wp3_tr(Ab + p * APLANE + tofs[it][s][0]), hi = wp3_tr(Ab + p * APLANE + tofs[it][s][1]);
                        ta[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                    const wp3_bf16x8 ah = wp3_op(ta[0]), am = wp3_op(ta[1]), al = wp3_op(ta[2]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const wp3_bf16x8 bh = wp3_op(bfr[j][s][0]), bm = wp3_op(bfr[j][s][1]), bl = wp3_op(bfr[j][s][2]);
                        f32x16 d = dw[2 * c + it][j];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
                        dw[2 * c + it][j] = d;
                    }
                }
        }

        // ---- epilogue of the dA product: mask, store, BatchNorm-backward sums of layer l-1 ----
        {
            const unsigned sc_off = (unsigned)strip * (32u * (unsigned)Nout * 4u);
            const __amdgpu_buffer_rsrc_t resC = wp3_rsrc(E.c);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ((maskb[j] >> r) & 1u) ? (NACC == 2 ? acc[j][r] + accl[NACC == 2 ? j : 0][r] : acc[j][r]) : 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), resC, (int)oq, (int)(sc_off + quni(j, r)), 0);
                    t1 += v;
                    t2 = fmaf(v, zq[j][r], t2);
                }
                const double d1 = (double)t1;   // sum v xhat = istd (sum v z - mu sum v), finished in float64
                s1[j] += d1, s2[j] += e_is[j] * ((double)t2 - e_mu[j] * d1);
            }
        }
    }

    // ---- one dW partial per workgroup: every wave parks its tiles in LDS (16-byte groups of four accumulator registers = four
    // consecutive dW rows), one barrier, then wave w adds the four copies of tiles w, w + 4 in wave order and stores them ----
    __syncthreads();   // every wave is done with the panel and its strips
    {
        constexpr int NTILE = CT * 2;
        f32x4 *red = reinterpret_cast<f32x4 *>(lds3);   // [tile][wave][r4][lane]
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
                    v[0] = dw[i][j][4 * r4], v[1] = dw[i][j][4 * r4 + 1], v[2] = dw[i][j][4 * r4 + 2], v[3] = dw[i][j][4 * r4 + 3];
                    red[(((i * 2 + j) * 4 + wave) * 4 + r4) * 64 + lane] = v;
                }
        __syncthreads();
        float *wb = E.dwslab + (size_t)worker * KD * E.dw_ld + n0;
#pragma unroll
        for (int tt = 0; tt < NTILE / 4; ++tt) {
            const int t = wave + 4 * tt, i = t >> 1, j = t & 1;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 a0 = red[((t * 4 + 0) * 4 + r4) * 64 + lane], a1 = red[((t * 4 + 1) * 4 + r4) * 64 + lane];
                const f32x4 a2 = red[((t * 4 + 2) * 4 + r4) * 64 + lane], a3 = red[((t * 4 + 3) * 4 + r4) * 64 + lane];
                const int c0 = i * 32 + 8 * r4 + 4 * lh;
                float *o = wb + (size_t)c0 * E.dw_ld + j * 32 + l31;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[(size_t)e * E.dw_ld] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
            }
        }
    }
    // ---- column statistics of the worker ----
    __syncthreads();
    double *dred = reinterpret_cast<double *>(lds3);   // [4 waves][2][64]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double a = s1[j] + shfl_xor_f64(s1[j], 32), b = s2[j] + shfl_xor_f64(s2[j], 32);
        if (lh == 0) dred[(wave * 2 + 0) * BN + j * 32 + l31] = a, dred[(wave * 2 + 1) * BN + j * 32 + l31] = b;
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid / BN, cl = tid % BN;
        const double t = (dred[(0 * 2 + which) * BN + cl] + dred[(1 * 2 + which) * BN + cl]) +
                         (dred[(2 * 2 + which) * BN + cl] + dred[(3 * 2 + which) * BN + cl]);
        E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
    }
}

template <int KD, int AM>
static void wsp3_launch(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int workers, int ncol, hipStream_t st) {
    constexpr size_t main_b = (size_t)3 * 64 * KD * 2 + (size_t)4 * 2 * 3 * 32 * 128, red_b = (size_t)(KD / 32) * 2 * 4 * 4 * 64 * 16;
    constexpr size_t lds = main_b > red_b ? main_b : red_b;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kfn = gemm_wsp3_kernel<KD, AM>;
    static bool granted = false;
    if (!granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(256), lds, st, A, B.b, B.ldb, M, Nout, ncol, E);
}

bool wsp3_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E) {
    if (!split_products() || matmul_precision() != 0) return false;
    if (M < 8192 || M % 32 != 0 || !(Nout == 64 || Nout == 128) || Kd != 128) return false;
    if (!(A.mode == A_DZ || (A.mode == A_DZ_POOL && A.K == 32))) return false;
    if (E.mode != E_MASK_STATS || !E.dwslab || E.dw_ld < Nout) return false;
    if (B.trans || B.perm_D >= 0 || (B.rows > 0 && B.rows != Kd) || B.ldb < Nout) return false;
    if (A.lda != Kd || A.C != Kd || E.ldc != Nout) return false;
    if ((((uintptr_t)A.a | (uintptr_t)A.z | (uintptr_t)E.zp | (uintptr_t)E.c | (uintptr_t)A.cst) & 15) != 0) return false;
    if (A.mode == A_DZ_POOL && (((uintptr_t)A.arg) & 15) != 0) return false;
    if ((unsigned long long)M * (unsigned)Kd * 4ull >= 0xfffffff0ull) return false;   // 32-bit buffer offsets
    return true;
}

bool try_launch_wsp3(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                     int *dw_slabs) {
    *rc = PNPP_OK;
    if (!dw_slabs || !wsp3_applies(A, B, M, Nout, Kd, E)) return false;
    const int nstrips = M / 32, ncol = Nout / 64;
    int workers = 256 / ncol;   // one workgroup per CU, one wave per SIMD
    if (workers * 4 > nstrips) workers = (nstrips + 3) / 4;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (nslab) *nslab = workers;
    *dw_slabs = workers;
    ProfScope ps(st, "gemm_wsp3_kernel<%d,A%d> M=%d N=%d K=%d grid=%dx1", Kd, A.mode, M, Nout, Kd, workers * ncol);
    if (A.mode == A_DZ_POOL) wsp3_launch<128, A_DZ_POOL>(A, B, M, Nout, E, workers, ncol, st);
    else wsp3_launch<128, A_DZ>(A, B, M, Nout, E, workers, ncol, st);
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsp3: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

}  // namespace pnpp
