// gemm_wsp_kernels.hip -- the fused backward product of a grouped layer whose INPUT is 64 channels wide, on wave-private row strips.
//
// Reference: the autograd backward of conv -> BatchNorm -> ReLU (models/pointnet_pp_8dir.py:40-42) for layer l of a grouped level:
//   dZ_l  = BatchNorm-backward(dY_l, Z_l)                       (operand transform; dY_l of the level's last layer is rebuilt from the
//                                                                pooled gradient and the arg-max rows: A_DZ_POOL)
//   dY_{l-1} = (dZ_l W_l) masked by ReLU'(layer l-1), + its BatchNorm-backward column sums            (epilogue E_MASK_STATS)
//   dW_l  = dZ_l^T relu(bn(Z_{l-1}))                                                                  (fused second product)
//
// gemm_ws_kernel<..., dW> (gemm_kernels.hip) does this on 64-row tiles staged by the whole workgroup: three barriers per tile, two
// waves of different workgroups per SIMD.  Round 3's counters say what that form waits for: a third fewer VALU instructions moved
// the same number of cycles from "issuing" to "waiting" (profiles/round3_sq_counters*.txt).  The only cross-row quantity here is the
// SUM over rows in dW, and a sum can be kept per wave: with C_{l-1} = 64 the whole dW (C_l x 64, C_l <= 128) is at most eight
// 32 x 32 accumulator tiles = 128 registers, which one wave per SIMD can afford.  So a wave owns strips of 32 rows end to end --
// operand streams one strip ahead in registers, dZ and z_{l-1} images in ITS OWN LDS, both products, the epilogue -- and the tile
// loop has no barrier at all; the four waves' dW tiles are added through LDS once, at the end (one partial per workgroup: half the
// slabs of the 64-row form).  A strip is one neighbourhood (nsample = 32), so the pooled gradient / arg-max entries of A_DZ_POOL are
// one row of those tables per strip -- and the rebuilt dY has ONE non-zero row per channel: the strip is staged as a z + b (one FMA
// per element) and the 128 entries g dm are added to their rows afterwards (two LDS adds per lane) instead of a compare + select +
// FMA per element.
//
// LDS of one wave executes in order, so the strip images need no barrier between their writes and their reads.
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsp_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wsp_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}
__device__ __forceinline__ float wsp_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}

#ifndef WSP_EXP   // timing experiments (wrong results): 1 no output stores, 2 no dW loop, 4 no dA loop, 8 no epilogue, 16 no staging, 32 no re-fetch
#define WSP_EXP 0
#endif
#ifdef PNPP_STAMPS
__device__ unsigned long long g_wsp_stamps[2][16];   // [KD == 128][phase]: s_memtime ticks of wave 0 of workgroup 8, a full wait at each stamp
#define WSP_STAMP(i)                                                          \
    if (st_on) {                                                              \
        __builtin_amdgcn_s_waitcnt(0);                                        \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime();         \
        st_acc[i] += st_t - st_last;                                          \
        st_last = st_t;                                                       \
    }
#else
#define WSP_STAMP(i)
#endif

// KD = C_l in {64, 128}; C_{l-1} = 64 (BN); AM = A_DZ or A_DZ_POOL
//
// What the kernel is built around (scratch/coissue.hip): on this chip a float32 MFMA and the vector instructions of
// the same SIMD do not overlap -- every VALU instruction of a strip is matrix time lost, whichever wave issues it and wherever it is
// placed -- while LDS and memory instructions issue beside the matrix pipe.  So the strip is arranged to need as few VALU
// instructions as possible:
//   * the BatchNorm-backward transform dZ = g dY + a Z + b (a = -g istd c2, b = -g c1 - a mu) is folded into the products:
//       A_DZ_POOL:  dZ = a (Z + G / a) + b,  G = the one-hot g dm term: the image is the RAW Z (a copy) with g dm / a added to one
//                   element per channel; the weight panel is diag(a) W
//       A_DZ:       dZ = g (dY + k Z) + b,   k = a / g = -istd c2: one FMA per element; the panel is diag(g) W
//     the constant row b W is what the dA accumulators start from, and dW = diag(scale) (X^T A) + b (1^T A): the column sums of the
//     activation operand are one add per operand value, the scaling happens once, in the tail
//     (a live channel whose k is zero cannot be divided by: the whole launch then takes the k-form, dZ = g (dY + k Z) with k Z as
//     one multiply per element -- eval-mode statistics, c1 = c2 = 0, are such a case)
//   * the rows of the dW reduction are visited in the order of the dA accumulator layout (row 4 lh + (s & 3) + 8 (s >> 2) at step
//     s): the activation value a lane feeds to the dW product at step s, relu(bn(z)), is then the value whose sign masks the lane's
//     dA accumulator s -- computed once, from z_{l-1} held in registers in accumulator layout; there is no activation image in LDS
//   * sum v xhat = istd (sum v z - mu sum v): one FMA per element, the rest once per strip in float64
//   * the operand registers are re-loaded for the next strip once their contents have been used -- the dense streams one 16-byte group
//     per step of the dA product, a z_{l-1} value right behind its step of the dW product (a full strip period of latency cover, no
//     second register set, no copies); past the last strip the loads go through a null descriptor (no branch, no traffic)
template <int KD, int AM>
__global__ void __launch_bounds__(256, 1)
gemm_wsp_kernel(const AOperand A, const float *__restrict__ W, int ldw, int M, const Epilogue E) {
    constexpr int BN = 64, NC = KD / 64, DP = KD + 4;      // DP: pitch of the dZ image (16-byte row reads AND column reads conflict-free)
    constexpr int CT = KD / 32;                            // dW row tiles (channels of layer l); two column tiles (64 / 32)
    constexpr int MAIN = BN * KD + 4 * 32 * DP, RED = CT * 2 * 4 * 4 * 64 * 4;   // floats: panel + strips | tail reduction
    constexpr int TAB = MAIN > RED ? MAIN : RED;           // tables behind both
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds;                                       // [BN][KD]: (diag(scale) W)^T image, 16-byte groups swizzled by (n & 15)
    float *Tsc = lds + TAB, *Tb = Tsc + KD, *Tbw = Tb + KD, *Tred = Tbw + BN;   // scale[KD], b[KD], (b W)[64], scratch [4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *Dz = lds + BN * KD + wave * (32 * DP);          // this wave's strip image [32][DP]
    const int l31 = lane & 31, lh = lane >> 5;
    auto swz = [](int r) { return (r & 15) << 2; };
    const int worker = blockIdx.x, nworkers = gridDim.x;
#ifdef PNPP_STAMPS
    const bool st_on = blockIdx.x == 8 && wave == 0;
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // registers: one write per launch, at the end
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

    // staging map of a strip: column group lane % 16 (+ 64 c), rows lane / 16 + 4 i
    const int q4 = 4 * (lane & 15), rb = lane >> 4;
    const int nstrips = M / 32, stride = nworkers * 4;
    int strip = worker * 4 + wave;
    const __amdgpu_buffer_rsrc_t resZ = wsp_rsrc(A.z), resY = wsp_rsrc(A.a), resP = wsp_rsrc(E.zp), resI = wsp_rsrc(A.arg);
    const __amdgpu_buffer_rsrc_t resC = wsp_rsrc(E.c);
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.z), (short)0, 0, 0x00020000);
    unsigned oa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = 4u * ((unsigned)(rb + 4 * i) * (unsigned)KD + (unsigned)q4);
    // accumulator positions of a strip of layer l-1 (z_{l-1} in, dY_{l-1} out): row 4 lh + (r & 3) + 8 (r >> 2), column 32 j + l31;
    // two lane offsets (rows 0..15 / 16..31) keep the per-element part inside the 12-bit immediate
    const unsigned oq0 = 4u * ((unsigned)(4 * lh) * BN + (unsigned)l31), oq1 = oq0 + 16u * BN * 4u;
    auto qoff = [&](int j, int r) -> unsigned { return ((r >> 3) ? oq1 : oq0) + 4u * (unsigned)(((r & 3) + 8 * ((r >> 2) & 1)) * BN + 32 * j); };

    // operand streams: registers, one strip ahead
    f32x4 rz[NC][8], ry[AM == A_DZ ? NC : 1][8];
    float zq[2][16];   // z_{l-1} at this lane's accumulator positions
    float rdm[NC];
    int rarg[NC];
    auto fetch_z = [&](__amdgpu_buffer_rsrc_t rZ, __amdgpu_buffer_rsrc_t rY, int s, int c, int i) {
        const unsigned so = (unsigned)s * (32u * KD * 4u);
        rz[c][i] = wsp_load4(rZ, oa[i] + 256u * (unsigned)c, so);
        if constexpr (AM == A_DZ) ry[c][i] = wsp_load4(rY, oa[i] + 256u * (unsigned)c, so);
    };
    auto fetch_p = [&](__amdgpu_buffer_rsrc_t rP, int s, int j, int r) { zq[j][r] = wsp_load1(rP, qoff(j, r), (unsigned)s * (32u * BN * 4u)); };
    auto fetch_g = [&](__amdgpu_buffer_rsrc_t rY, __amdgpu_buffer_rsrc_t rI, int s, int c) {   // one row of the pooled tables per strip
        if constexpr (AM == A_DZ_POOL) {
            const unsigned sg = (unsigned)s * (KD * 4u);
            rdm[c] = wsp_load1(rY, 4u * (unsigned)lane + 256u * (unsigned)c, sg);
            rarg[c] = __builtin_bit_cast(int, wsp_load1(rI, 4u * (unsigned)lane + 256u * (unsigned)c, sg));
        }
    };
    // the first strip's loads go out before anything else of the prologue (chunk 0 first: the MFMAs start on it)
    {
        const bool have = strip < nstrips;
        const __amdgpu_buffer_rsrc_t z0 = have ? resZ : resNull, y0 = have ? resY : resNull, p0 = have ? resP : resNull, i0 = have ? resI : resNull;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int i = 0; i < 8; ++i) fetch_z(z0, y0, strip, c, i);
            fetch_g(y0, i0, strip, c);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) fetch_p(p0, strip, 0, r), fetch_p(p0, strip, 1, r);
    }

    // ---- per-channel tables: a, b, k; which form the launch takes ----
    float ch_a = 0.f, ch_b = 0.f, ch_g = 0.f, ch_k = 0.f;
    int bad = 0;
    if (tid < KD) {
        const float g = A.cst[tid], mu = A.cst[A.C + tid], is = A.cst[2 * A.C + tid], c1 = A.cst[3 * A.C + tid], c2 = A.cst[4 * A.C + tid];
        ch_g = g, ch_k = -is * c2, ch_a = g * ch_k, ch_b = -g * c1 - ch_a * mu;
        // dividing by k must stay finite for every pooled gradient: |k| >= 1e-30 (or the channel is dead: g == 0, dZ = 0)
        bad = (g != 0.f && !(fabsf(ch_k) >= 1e-30f)) || !(g == g);
    }
    // the weight panel's loads go out here, in front of the first barrier (nothing they need is behind it): one round trip for constants,
    // panel and first strip instead of two in a row
    constexpr int NWF = (KD / 4) * BN / 256;
    f32x4 tw[NWF];
#pragma unroll
    for (int j = 0; j < NWF; ++j) {
        const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
        for (int e = 0; e < 4; ++e) tw[j][e] = W[(size_t)(k4 + e) * ldw + nl];
    }
    const bool fold = (AM == A_DZ_POOL) && !__syncthreads_or(bad);   // image = raw Z, panel = diag(a) W; else k-form, panel = diag(g) W
    if (tid < KD) Tsc[tid] = fold ? ch_a : ch_g, Tb[tid] = ch_b;
    // staging multipliers (k-form) / fix-up multipliers of this lane
    float4 kq[NC];
    float fx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float *p = A.cst + 64 * c + q4;
        const float4 is = *reinterpret_cast<const float4 *>(p + 2 * A.C), c2 = *reinterpret_cast<const float4 *>(p + 4 * A.C);
        kq[c] = make_float4(-is.x * c2.x, -is.y * c2.y, -is.z * c2.z, -is.w * c2.w);
        if constexpr (AM == A_DZ_POOL) {
            const int ch = 64 * c + lane;
            const float is1 = A.cst[2 * A.C + ch], c21 = A.cst[4 * A.C + ch];
            const float k1 = -is1 * c21;
            fx[c] = !fold ? 1.f : (fabsf(k1) >= 1e-30f ? 1.f / k1 : 0.f);   // g dm / a = dm / k (a dead channel adds nothing);  k-form: dm itself
        }
    }
    // epilogue constants of this lane's two output columns (l31 and 32 + l31 of layer l-1)
    float e_sc[2], e_sh[2];
    double e_mu[2], e_is[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + l31;
        e_sc[j] = E.scale[col], e_sh[j] = E.shift[col], e_mu[j] = (double)E.mu[col], e_is[j] = (double)E.istd[col];
    }
    __syncthreads();   // tables

    // weight panel: W is (KD x 64) row-major; image [n][k ^ swz(n)] of diag(scale) W (lane = column n: four dword loads of consecutive
    // rows, one conflict-free 16-byte LDS store per group); the same pass takes this thread's share of b W
    {
        float bwp = 0.f;
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(Tsc + k4), bb = *reinterpret_cast<const f32x4 *>(Tb + k4);
            bwp = fmaf(bb[0], tw[j][0], fmaf(bb[1], tw[j][1], fmaf(bb[2], tw[j][2], fmaf(bb[3], tw[j][3], bwp))));
            f32x4 t;
            t[0] = sc[0] * tw[j][0], t[1] = sc[1] * tw[j][1], t[2] = sc[2] * tw[j][2], t[3] = sc[3] * tw[j][3];
            *reinterpret_cast<f32x4 *>(Ws + nl * KD + (k4 ^ swz(nl))) = t;
        }
        Tred[wave * BN + lane] = bwp;   // nl == lane for every group of this thread
    }
    __syncthreads();
    if (tid < BN) Tbw[tid] = (Tred[tid] + Tred[BN + tid]) + (Tred[2 * BN + tid] + Tred[3 * BN + tid]);

    f32x16 dw[CT][2];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[i][j][r] = 0.f;
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0}, sa[2] = {0.0, 0.0};
    __syncthreads();   // the panel and (b W) are complete; from here on the waves run on their own
    const float bw0 = Tbw[l31], bw1 = Tbw[32 + l31];
    WSP_STAMP(8)   // prologue

    // one 16-byte group of the strip image: rows rb + 4 i of chunk c
    auto stage_z = [&](int c, int i) {
        f32x4 v = rz[c][i];
        if constexpr (AM == A_DZ) {
            const f32x4 dy = ry[c][i];
            v[0] = fmaf(kq[c].x, v[0], dy[0]), v[1] = fmaf(kq[c].y, v[1], dy[1]);
            v[2] = fmaf(kq[c].z, v[2], dy[2]), v[3] = fmaf(kq[c].w, v[3], dy[3]);
        } else if (!fold) {   // a real (uniform) branch: the empty asm keeps hipcc from turning it into a select per element
            asm volatile("" : "+v"(v));
            v[0] *= kq[c].x, v[1] *= kq[c].y, v[2] *= kq[c].z, v[3] *= kq[c].w;
        }
        *reinterpret_cast<f32x4 *>(Dz + (rb + 4 * i) * DP + 64 * c + q4) = v;
    };
    // A_DZ_POOL: dY has one non-zero row per channel: its entry lands on row arg of column 64 c + lane
    auto fixup = [&](int c) {
        if constexpr (AM == A_DZ_POOL) {
            const int a = rarg[c];
            const float dmv = rdm[c];
            if ((unsigned)a < 32u && dmv != 0.f) {
                float *p = Dz + a * DP + 64 * c + lane;
                *p = fmaf(fx[c], dmv, *p);
            }
        }
    };

    const float *arow = Dz + l31 * DP + 4 * lh;
    const float *brow[2];
    int gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + l31;
        brow[j] = Ws + n * KD;
        gb[j] = (4 * lh) ^ swz(n);
    }
    const float *dcol = Dz + 4 * lh * DP + l31;   // dW reduction rows in accumulator order: row 4 lh + (s & 3) + 8 (s >> 2) at step s

    for (; strip < nstrips; strip += stride) {
        WSP_STAMP(0)
        const bool more = !(WSP_EXP & 32) && strip + stride < nstrips;
        const int snext = strip + stride;
        const __amdgpu_buffer_rsrc_t nZ = more ? resZ : resNull, nY = more ? resY : resNull, nP = more ? resP : resNull, nI = more ? resI : resNull;
        // ---- chunk 0 of the strip image (the only stretch of the loop without MFMAs) ----
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_z(0, i);
        fixup(0);
        WSP_STAMP(1)

        // ---- dA = X W', two column tiles, k = 8 t + 4 lh + u; image rows by ds_read_b128 (pitch KD + 4), W'^T image swizzled; the
        // accumulators start from b W ----
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bw0, acc[1][r] = bw1;
        if (!(WSP_EXP & 4)) {
            float4 fa[2], fb[2][2];
            auto ld = [&](int buf, int t) {
                fa[buf] = *reinterpret_cast<const float4 *>(arow + 8 * t);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[buf][j] = *reinterpret_cast<const float4 *>(brow[j] + ((8 * t) & ~63) + (((8 * t) & 63) ^ gb[j]));
            };
            auto mm = [&](int buf) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf][j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf][j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf][j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf][j].w, acc[j], 0, 0, 0);
                }
            };
            constexpr int NTT = KD / 8;
            ld(0, 0);
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                // chunk 1 of the image is complete -- fix-up included -- before step 7 prefetches its first operands
                if (NC == 2 && t < 4) stage_z(NC - 1, 2 * t), stage_z(NC - 1, 2 * t + 1);
                if (NC == 2 && t == 4) fixup(NC - 1);
                // the dense streams of the next strip, one 16-byte group per step: its register was staged at least four steps ago
                // (re-loading a register right behind its staging -- all of a chunk's loads within four steps -- measured +2.3 us)
                if (NC == 2) {
                    if (t < 8) fetch_z(nZ, nY, snext, 0, t);
                    else fetch_z(nZ, nY, snext, 1, t - 8);
                    if (t == 8 || t == 9) fetch_g(nY, nI, snext, t - 8);
                } else {
                    fetch_z(nZ, nY, snext, 0, t);
                    if (t == 0) fetch_g(nY, nI, snext, 0);
                }
                if (t + 1 < NTT) ld((t + 1) & 1, t + 1);
                mm(t & 1);
            }
        }
        WSP_STAMP(2)
        // ---- dW += X^T a, reduction rows in accumulator order, with the epilogue of the dA product: the activation value of step s IS
        // the value whose sign masks accumulator s.  Mask, store, sums; then the z_{l-1} register is re-loaded for the next strip ----
        {
            const unsigned sc_off = (unsigned)strip * (32u * BN * 4u), sn_off = (unsigned)snext;
            float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f}, ta[2] = {0.f, 0.f};
            float fd[2][CT];
            auto ldw = [&](int buf, int s) {
#pragma unroll
                for (int i = 0; i < CT; ++i) fd[buf][i] = dcol[((s & 3) + 8 * (s >> 2)) * DP + 32 * i];
            };
            if (!(WSP_EXP & 2)) ldw(0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float bact[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float z0 = zq[j][s];
                    const float a0 = fmaf(z0, e_sc[j], e_sh[j]);
                    bact[j] = fmaxf(a0, 0.f);
                    if (!(WSP_EXP & 8)) {
                        const float v = a0 > 0.f ? acc[j][s] : 0.f;
                        if (!(WSP_EXP & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), resC, (int)qoff(j, s), (int)sc_off, 0);
                        t1[j] += v;
                        t2[j] = fmaf(v, z0, t2[j]);
                        ta[j] += bact[j];
                    }
                    fetch_p(nP, (int)sn_off, j, s);
                }
                if (!(WSP_EXP & 2)) {
                    if (s + 1 < 16) ldw((s + 1) & 1, s + 1);
#pragma unroll
                    for (int i = 0; i < CT; ++i) {
                        dw[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[s & 1][i], bact[0], dw[i][0], 0, 0, 0);
                        dw[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[s & 1][i], bact[1], dw[i][1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // sum v xhat = istd (sum v z - mu sum v), finished in float64
                const double d1 = (double)t1[j];
                s1[j] += d1, s2[j] += e_is[j] * ((double)t2[j] - e_mu[j] * d1), sa[j] += (double)ta[j];
            }
        }
        WSP_STAMP(4)
    }

    // ---- one dW partial per workgroup: every wave parks its tiles of X^T A in LDS (16-byte groups of four accumulator registers =
    // four consecutive dW rows), one barrier, then wave w adds the four copies of tiles w, w + 4, ... in wave order, applies
    // dW = scale (X^T A) + b (1^T A) and stores them ----
    __syncthreads();   // every wave is done with the panel and its strips
    {
        constexpr int NTILE = CT * 2;
        f32x4 *red = reinterpret_cast<f32x4 *>(lds);   // [tile][wave][r4][lane]
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
        // column sums of the activation operand over the workgroup's rows: lane halves, then the four waves in wave order
        double *ared = reinterpret_cast<double *>(Tred);   // [4 waves][64] doubles = the scratch + what follows it (sized by the launcher)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double a = sa[j] + shfl_xor_f64(sa[j], 32);
            if (lh == 0) ared[wave * BN + j * 32 + l31] = a;
        }
        __syncthreads();
        float asum[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = j * 32 + l31;
            asum[j] = (float)((ared[k] + ared[BN + k]) + (ared[2 * BN + k] + ared[3 * BN + k]));
        }
        float *wb = E.dwslab + (size_t)worker * KD * 64;
#pragma unroll
        for (int tt = 0; tt < NTILE / 4; ++tt) {
            const int t = wave + 4 * tt, i = t >> 1, j = t & 1;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 a0 = red[((t * 4 + 0) * 4 + r4) * 64 + lane], a1 = red[((t * 4 + 1) * 4 + r4) * 64 + lane];
                const f32x4 a2 = red[((t * 4 + 2) * 4 + r4) * 64 + lane], a3 = red[((t * 4 + 3) * 4 + r4) * 64 + lane];
                const int c0 = i * 32 + 8 * r4 + 4 * lh;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(Tsc + c0), bb = *reinterpret_cast<const f32x4 *>(Tb + c0);
                float *o = wb + (size_t)c0 * 64 + j * 32 + l31;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e * 64] = fmaf(sc[e], (a0[e] + a1[e]) + (a2[e] + a3[e]), bb[e] * (j == 0 ? asum[0] : asum[1]));
            }
        }
    }
    // ---- column statistics of the worker ----
    __syncthreads();
    double *dred = reinterpret_cast<double *>(lds);   // [4 waves][2][64]
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
        E.slab[((size_t)worker * 2 + which) * BN + cl] = t;
    }
    WSP_STAMP(9)   // tail: dW partial through LDS, statistics
#ifdef PNPP_STAMPS
    if (st_on && lane == 0)
#pragma unroll
        for (int i = 0; i < 10; ++i) g_wsp_stamps[KD == 128][i] += st_acc[i];
#endif
}

// A/B switch: PNPP_NO_WSP=1 keeps these launches on gemm_ws_kernel<..., dW>
static bool wsp_on() {
    static int cached = -1;
    if (cached < 0) {
        const char *v = getenv("PNPP_NO_WSP");
        cached = (v && atoi(v) != 0) ? 0 : 1;
    }
    return cached != 0;
}

template <int KD, int AM>
static void wsp_launch(const AOperand &A, const BOperand &B, int M, const Epilogue &E, int workers, hipStream_t st) {
    constexpr size_t main_f = (size_t)64 * KD + 4 * 32 * (KD + 4), red_f = (size_t)(KD / 32) * 2 * 4 * 4 * 64 * 4;
    // tables behind the larger of the two: scale[KD], b[KD], (b W)[64], scratch [4][64] floats reused as [4][64] doubles
    constexpr size_t lds = ((main_f > red_f ? main_f : red_f) + 2 * KD + 64 + 2 * 4 * 64) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kfn = gemm_wsp_kernel<KD, AM>;
    static bool granted = false;
    if (lds > 48 * 1024 && !granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers), dim3(256), lds, st, A, B.b, B.ldb, M, E);
}

bool wsp_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E) {
    if (!wsp_on() || matmul_precision() != 0) return false;
    // the last layer of a grouped level, 64 -> 128 channels (SA1 of every reference model).  The template also covers C_l = 64 and
    // the dense-gradient operand (A_DZ); measured on the 64 -> 64 layer of SA1 it loses to gemm_ws_kernel (38.5 - 39.5 against
    // 33.6 us: that launch moves the same bytes for half the products, and one wave per SIMD keeps too few requests in flight),
    // so those shapes stay there
    if (M < 8192 || M % 32 != 0 || Nout != 64 || Kd != 128 || A.mode != A_DZ_POOL || A.K != 32) return false;
    if (E.mode != E_MASK_STATS || !E.dwslab || E.dw_ld != 64) return false;
    if (B.trans || B.perm_D >= 0 || (B.rows > 0 && B.rows != Kd) || B.ldb < 64) return false;
    if (A.lda != Kd || A.C != Kd || E.ldc != 64) return false;
    if ((((uintptr_t)A.a | (uintptr_t)A.z | (uintptr_t)E.zp | (uintptr_t)E.c | (uintptr_t)A.cst) & 15) != 0) return false;
    if ((unsigned long long)M * (unsigned)Kd * 4ull >= 0xfffffff0ull) return false;   // 32-bit buffer offsets
    return true;
}

bool try_launch_wsp(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                    int *dw_slabs) {
    *rc = PNPP_OK;
    if (!dw_slabs || !wsp_applies(A, B, M, Nout, Kd, E)) return false;
    const int nstrips = M / 32;
    int workers = 256;   // one workgroup per CU, one wave per SIMD
    if (workers * 4 > nstrips) workers = (nstrips + 3) / 4;
    if (nslab) *nslab = workers;
    *dw_slabs = workers;
    ProfScope ps(st, "gemm_wsp_kernel<%d,A%d> M=%d N=%d K=%d grid=%dx1", Kd, A.mode, M, Nout, Kd, workers);
    wsp_launch<128, A_DZ_POOL>(A, B, M, E, workers, st);
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsp: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

#ifdef PNPP_STAMPS
#define PNPP_STAMPS_BIT 64u
#else
#define PNPP_STAMPS_BIT 0u
#endif
unsigned wsp_build_flags() { return ((WSP_EXP != 0) ? 2u : 0u) | PNPP_STAMPS_BIT; }

}  // namespace pnpp

#ifdef PNPP_STAMPS
extern "C" int pnpp_debug_wsp_stamps(unsigned long long *out32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_wsp_stamps), z, sizeof(z));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out32, HIP_SYMBOL(pnpp::g_wsp_stamps), 32 * sizeof(unsigned long long));
    }
    return 0;
}
#endif
