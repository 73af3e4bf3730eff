// gemm_wsq_kernels.hip -- the fused backward product of a grouped level's LAST layer when that layer is 256 channels wide (SA2 of every
// reference model: 128 -> 256), on 64-row tiles staged by the workgroup, with the instruction diet of gemm_wsp_kernels.hip.
//
// Reference: the autograd backward of conv -> BatchNorm -> ReLU -> max over the neighbourhood (models/pointnet_pp_8dir.py:40-43) for the
// last layer l of a level:
//   dZ_l  = BatchNorm-backward(dY_l, Z_l),  dY_l rebuilt from the pooled gradient and the arg-max rows (A_DZ_POOL)
//   dY_{l-1} = (dZ_l W_l) masked by ReLU'(layer l-1), + its BatchNorm-backward column sums            (epilogue E_MASK_STATS)
//   dW_l  = dZ_l^T relu(bn(Z_{l-1}))                                                                  (fused second product)
//
// gemm_ws_kernel<256,64,64,A5,E2,dW> (gemm_kernels.hip) does this with ~1,200 vector instructions per wave and 64-row tile beside its 256
// matrix instructions -- and on this chip a float32 MFMA and the vector instructions of the same SIMD do not overlap:
// a launch is MFMA time + the rest.  A wave-private form (gemm_wsp_kernels.hip) does not fit here -- 256 x 128 weight-gradient
// accumulators are 512 registers -- so the tile stays a workgroup's, but the rest of that kernel's diet carries over:
//   * the BatchNorm-backward transform is folded into the products: dZ = a (Z + G / a) + b with G the one-hot pooled-gradient term, so the
//     tile image is the RAW Z (a copy: no arithmetic) plus one read-modify-write per (neighbourhood, channel); the weight panel is
//     diag(a) W, the dA accumulators start from the row b W, and dW = diag(a) (X^T A) + b (1^T A) is finished once, in the tail
//     (k-form dZ = g (k Z + G / g) when some live channel has k = -istd c2 = 0: eval-mode statistics);
//   * a fix-up is applied by the thread whose WAVE staged that element (rows r with r mod 4 == wave): LDS operations of one wave
//     execute in order, so no barrier separates the staging pass from the fix-ups;
//   * the rows of the dW reduction are visited in accumulator order: the activation a lane feeds to the dW product at step s is
//     relu(bn(z)) of the element whose sign masks its dA accumulator s -- computed once, from z_{l-1} in registers; there is no
//     activation image in LDS and no staging pass for it;
//   * every operand register is re-loaded for the next tile right behind its last use (the dense stream one 16-byte group per two
//     steps of the dA product, a z_{l-1} value behind its step of the dW product).
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsq_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wsq_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}
typedef int i32x4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4q wsq_load4i(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(i32x4q, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}
__device__ __forceinline__ float wsq_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}

#ifndef WSQ_PLAIN
#define WSQ_PLAIN 0
#endif
#ifndef WSQ_EXP   // timing experiments (wrong results): 2 no dW loop, 4 no dA loop, 8 no tile at all, 16 no fix-ups, 32 no epilogue arithmetic / stores
#define WSQ_EXP 0
#endif

#ifdef PNPP_STAMPS
__device__ unsigned long long g_wsq_stamps[16];   // s_memtime ticks of wave 0 of workgroup 8, per phase (no extra waits: what the wave itself sees)
#define WSQ_STAMP(i)                                                   \
    if (st_on) {                                                       \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += st_t - st_last;                                   \
        st_last = st_t;                                                \
    }
#else
#define WSQ_STAMP(i)
#endif

// KD = C_l = 256; the workgroup takes 64 of the Nout = C_{l-1} columns; waves (wm, wn) = 2 x 2 tiles of 32 x 32
template <int KD, int NOUT>
__global__ void __launch_bounds__(256, 1)
gemm_wsq_kernel(const AOperand A, const float *__restrict__ W, int ldw, int M, int ncol, const Epilogue E) {
    constexpr int Nout = NOUT;
    constexpr int BM = 64, BN = 64, DP = KD + 4, CT = KD / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds;                  // [BN][KD]: (diag(scale) W)^T image, 16-byte groups swizzled by (n & 15)
    float *Img = lds + BN * KD;       // [BM][DP]: the tile's dZ image (raw Z + fix-ups, or k Z + fix-ups)
    float *Tsc = Img + BM * DP, *Tb = Tsc + KD, *Tbw = Tb + KD, *Tred = Tbw + BN;   // scale[KD], b[KD], (b W)[64], scratch [4][64] floats / doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    auto swz = [](int r) { return (r & 15) << 2; };

    // XCD-aware map (as gemm_ws_kernel): the column blocks of one worker sit on one XCD and share its L2
    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;
    const int ntiles = M / BM;
    int tile = worker;
#ifdef PNPP_STAMPS
    const bool st_on = blockIdx.x == 8 && wave == 0;
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: shader cycles / real time = the clock held
#endif

    // staging map of a tile: column group lane (16 bytes), rows wave + 4 i -- a row's stager is the wave (row mod 4)
    const int q4 = 4 * lane;
    const __amdgpu_buffer_rsrc_t resZ = wsq_rsrc(A.z), resD = wsq_rsrc(A.a), resI = wsq_rsrc(A.arg), resP = wsq_rsrc(E.zp), resC = wsq_rsrc(E.c);
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.z), (short)0, 0, 0x00020000);
    unsigned oa[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) oa[i] = 4u * ((unsigned)(wave + 4 * i) * (unsigned)KD + (unsigned)q4);
    // accumulator positions of this wave's 32 x 32 tile of layer l-1 (z_{l-1} in, dY_{l-1} out): row 32 wm + 4 lh + (r & 3) + 8 (r >> 2),
    // column n0 + 32 wn + l31; one lane offset per group of four registers keeps the per-element part inside the 12-bit immediate
    unsigned oq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) oq[g] = 4u * ((unsigned)(32 * wm + 4 * lh + 8 * g) * (unsigned)Nout + (unsigned)(n0 + 32 * wn + l31));
    constexpr unsigned rowp = 4u * (unsigned)NOUT;   // bytes per row of layer l-1 (a constant: the per-element parts are immediates)

    f32x4 rz[16];      // the dense stream of Z_l, one tile ahead
    float zq[16];      // z_{l-1} at this lane's accumulator positions
    float zn[16];      // ... of the next tile (requested during the dA product, moved over behind the dW product)
    float vh[16];      // this tile's outputs, stored during the NEXT tile's dA product (see the tile loop)
    f32x4 gdm[2];      // pooled gradient / arg-max rows of the tile's two neighbourhoods at this lane's column group
    i32x4q garg[2];
    auto fetch_z = [&](__amdgpu_buffer_rsrc_t rZ, int t, int i) { rz[i] = wsq_load4(rZ, oa[i], (unsigned)t * (BM * KD * 4u)); };
    auto fetch_p = [&](__amdgpu_buffer_rsrc_t rP, int t, int r) {
        zn[r] = wsq_load1(rP, oq[r >> 2] + (unsigned)(r & 3) * rowp, (unsigned)t * (unsigned)BM * rowp);
    };
    auto fetch_g = [&](__amdgpu_buffer_rsrc_t rD, __amdgpu_buffer_rsrc_t rI, int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned so = (unsigned)(2 * t + h) * (KD * 4u);
            gdm[h] = wsq_load4(rD, 4u * (unsigned)q4, so);
            garg[h] = wsq_load4i(rI, 4u * (unsigned)q4, so);
        }
    };
    // ---- everything the prologue reads is requested before its first wait: first tile, constants, weight panel ----
    {
        const bool have = tile < ntiles;
        const __amdgpu_buffer_rsrc_t z0 = have ? resZ : resNull, d0 = have ? resD : resNull, i0 = have ? resI : resNull, p0 = have ? resP : resNull;
        const int t0 = have ? tile : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) fetch_z(z0, t0, i);
        fetch_g(d0, i0, t0);
#pragma unroll
        for (int r = 0; r < 16; ++r) fetch_p(p0, t0, r);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) vh[r] = 0.f;
    float ch_a = 0.f, ch_b = 0.f, ch_g = 0.f;
    int bad = 0;
    {   // KD == 256 == blockDim: one channel per thread
        const float g = A.cst[tid], mu = A.cst[A.C + tid], is = A.cst[2 * A.C + tid], c1 = A.cst[3 * A.C + tid], c2 = A.cst[4 * A.C + tid];
        const float k = -is * c2;
        ch_g = g, ch_a = g * k, ch_b = -g * c1 - ch_a * mu;
        // dividing by k must stay finite for every pooled gradient: |k| >= 1e-30 (or the channel is dead: g == 0, dZ = 0)
        bad = (g != 0.f && !(fabsf(k) >= 1e-30f)) || !(g == g);
    }
    f32x4 kq, fxq;   // k of this lane's column group (k-form staging multiplier), and its fix-up multiplier
    {
        const float *p = A.cst + q4;
        const float4 is = *reinterpret_cast<const float4 *>(p + 2 * A.C), c2 = *reinterpret_cast<const float4 *>(p + 4 * A.C);
        kq[0] = -is.x * c2.x, kq[1] = -is.y * c2.y, kq[2] = -is.z * c2.z, kq[3] = -is.w * c2.w;
    }
    float e_sc, e_sh;
    double e_mu, e_is;
    {
        const int col = n0 + 32 * wn + l31;
        e_sc = E.scale[col], e_sh = E.shift[col], e_mu = (double)E.mu[col], e_is = (double)E.istd[col];
    }
    constexpr int NWF = (KD / 4) * BN / 256;   // 16 groups of four consecutive k for one column n (lane = column: coalesced dword loads)
    f32x4 tw[NWF];
#pragma unroll
    for (int j = 0; j < NWF; ++j) {
        const int f = tid + 256 * j, nl = f % BN, k4 = 4 * (f / BN);
#pragma unroll
        for (int e = 0; e < 4; ++e) tw[j][e] = W[(size_t)(k4 + e) * ldw + n0 + nl];
    }
    const bool fold = !__syncthreads_or(bad);   // image = raw Z, panel = diag(a) W; else k-form, panel = diag(g) W
    Tsc[tid] = fold ? ch_a : ch_g, Tb[tid] = ch_b;
#if WSQ_PLAIN   // debugging: dZ computed element by element while staging, plain panel
    Tsc[tid] = 1.f, Tb[tid] = 0.f;
    f32x4 pg, pa, pb;
    {
        const float *p = A.cst + q4;
        const float4 g4 = *reinterpret_cast<const float4 *>(p), mu4 = *reinterpret_cast<const float4 *>(p + A.C);
        const float4 c14 = *reinterpret_cast<const float4 *>(p + 3 * A.C);
        pg[0] = g4.x, pg[1] = g4.y, pg[2] = g4.z, pg[3] = g4.w;
        pa[0] = g4.x * kq[0], pa[1] = g4.y * kq[1], pa[2] = g4.z * kq[2], pa[3] = g4.w * kq[3];
        pb[0] = -g4.x * c14.x - pa[0] * mu4.x, pb[1] = -g4.y * c14.y - pa[1] * mu4.y, pb[2] = -g4.z * c14.z - pa[2] * mu4.z, pb[3] = -g4.w * c14.w - pa[3] * mu4.w;
    }
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) fxq[e] = !fold ? 1.f : (fabsf(kq[e]) >= 1e-30f ? 1.f / kq[e] : 0.f);   // g dm / a = dm / k;  k-form: dm itself
    __syncthreads();   // tables
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

    f32x16 dw[CT];   // dW rows 32 i .., columns n0 + 32 wn ..: this wave's 32 rows of every tile
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
    double s1 = 0.0, s2 = 0.0, sa = 0.0;
    __syncthreads();   // the panel and (b W) are complete
    const float bw = Tbw[32 * wn + l31];

    const float *arow = Img + (32 * wm + l31) * DP + 4 * lh;
    const float *brow = Ws + (32 * wn + l31) * KD;
    const int gb = (4 * lh) ^ swz(32 * wn + l31);
    const float *dcol = Img + (32 * wm + 4 * lh) * DP + l31;   // dW reduction rows in accumulator order

    if (WSQ_EXP & 8) tile = ntiles;
    // vmcnt counts loads and stores in issue order, and the wait the compiler puts on the loop's back edge covers nearly everything
    // outstanding: memory operations issued at the END of a tile (output stores, z_{l-1} of the next tile) would be waited for -- a full
    // round trip per tile -- right behind their issue.  So a tile issues ALL its memory operations inside its dA product: the previous
    // tile's output stores (held in registers), then the next tile's z_{l-1} (a second register set), the dense stream throughout; the
    // dW product issues none, and whatever the back edge waits for has had that whole product to complete.
    __amdgpu_buffer_rsrc_t pC = resNull;   // the held outputs' destination (nothing is held before the first tile)
    unsigned prev_off = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) zq[r] = zn[r];
    WSQ_STAMP(8)   // prologue
    for (; tile < ntiles; tile += nworkers) {
        WSQ_STAMP(0)   // loop turn-around
        const bool more = tile + nworkers < ntiles;
        const int tnext = more ? tile + nworkers : 0;
        const __amdgpu_buffer_rsrc_t nZ = more ? resZ : resNull, nD = more ? resD : resNull, nI = more ? resI : resNull, nP = more ? resP : resNull;
        // ---- the tile image: this wave's rows (wave + 4 i), then the fix-ups that land on its rows ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 v = rz[i];
#if WSQ_PLAIN
            {
                const int r = wave + 4 * i, h = r >> 5, kk = r & 31;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = (garg[h][e] == kk) ? gdm[h][e] : 0.f;
                    v[e] = fmaf(pg[e], dy, fmaf(pa[e], v[e], pb[e]));
                }
            }
#else
            if (!fold) {   // a real (uniform) branch: the empty asm keeps hipcc from turning it into a select per element
                asm volatile("" : "+v"(v));
                v[0] *= kq[0], v[1] *= kq[1], v[2] *= kq[2], v[3] *= kq[3];
            }
#endif
            *reinterpret_cast<f32x4 *>(Img + (wave + 4 * i) * DP + q4) = v;
        }
#if !WSQ_PLAIN
        {   // dY has one non-zero row per (neighbourhood, channel): row arg of column q4 + e.  The eight candidates of a thread are
            // distinct addresses: all reads first, then all writes -- one LDS round trip instead of eight in a row (the compiler cannot
            // know they do not alias and would keep read-modify-write pairs in order: 0.85 us per tile)
            float cur[2][4];
            bool hit[2][4];
            float *pp[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int a = garg[h][e];
                    hit[h][e] = !(WSQ_EXP & 16) && (unsigned)a < 32u && (a & 3) == wave && gdm[h][e] != 0.f;
                    pp[h][e] = Img + (32 * h + (hit[h][e] ? a : wave)) * DP + q4 + e;   // a miss reads (and does not write) an element of its own
                    cur[h][e] = *pp[h][e];
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (hit[h][e]) *pp[h][e] = fmaf(fxq[e], gdm[h][e], cur[h][e]);
        }
#endif
        fetch_g(nD, nI, tnext);
        WSQ_STAMP(1)   // staging + fix-ups (incl. the wait for the tile's loads)
        __syncthreads();   // the image is complete
        WSQ_STAMP(2)   // barrier

        // ---- dA = X W', this wave's 32 x 32 tile, k = 8 t + 4 lh + u; the accumulator starts from b W ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bw;
        if (!(WSQ_EXP & 4)) {
            float4 fa[2], fb[2];
            auto ld = [&](int buf, int t) {
                fa[buf] = *reinterpret_cast<const float4 *>(arow + 8 * t);
                fb[buf] = *reinterpret_cast<const float4 *>(brow + ((8 * t) & ~63) + (((8 * t) & 63) ^ gb));
            };
            auto mm = [&](int buf) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf].w, acc, 0, 0, 0);
            };
            constexpr int NTT = KD / 8;
            ld(0, 0);
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                if (t < 16) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vh[t]), pC, (int)(oq[t >> 2] + (unsigned)(t & 3) * rowp), (int)prev_off, 0);
                else fetch_p(nP, tnext, t - 16);
                if ((t & 1) == 0) fetch_z(nZ, tnext, t >> 1);   // the next tile's dense stream: its registers were staged above
                if (t + 1 < NTT) ld((t + 1) & 1, t + 1);
                mm(t & 1);
            }
        }
        WSQ_STAMP(3)   // dA product
        // ---- dW += X^T a over this wave's 32 rows, reduction rows in accumulator order, with the epilogue of the dA product ----
        {
            const unsigned sc_off = (unsigned)tile * (unsigned)BM * rowp;
            float t1 = 0.f, t2 = 0.f, ta = 0.f;
            float fd[2][CT];
            auto ldw = [&](int buf, int s) {
#pragma unroll
                for (int i = 0; i < CT; ++i) fd[buf][i] = dcol[((s & 3) + 8 * (s >> 2)) * DP + 32 * i];
            };
            if (!(WSQ_EXP & 2)) ldw(0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float z0 = zq[s];
                const float a0 = fmaf(z0, e_sc, e_sh);
                const float bact = fmaxf(a0, 0.f);
                const float v = a0 > 0.f ? acc[s] : 0.f;
                vh[s] = v;
                t1 += v;
                t2 = fmaf(v, z0, t2);
                ta += bact;
                if (!(WSQ_EXP & 2)) {
                    if (s + 1 < 16) ldw((s + 1) & 1, s + 1);
#pragma unroll
                    for (int i = 0; i < CT; ++i) dw[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[s & 1][i], bact, dw[i], 0, 0, 0);
                }
            }
            const double d1 = (double)t1;   // sum v xhat = istd (sum v z - mu sum v), finished in float64
            s1 += d1, s2 += e_is * ((double)t2 - e_mu * d1), sa += (double)ta;
            pC = resC, prev_off = sc_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) zq[r] = zn[r];
        }
        WSQ_STAMP(4)   // dW product + epilogue
        __syncthreads();   // every wave is done with the image
        WSQ_STAMP(5)   // barrier
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)   // the last tile's outputs
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vh[r]), pC, (int)(oq[r >> 2] + (unsigned)(r & 3) * rowp), (int)prev_off, 0);

    // ---- one dW partial per workgroup: waves (0, wn) and (1, wn) hold the same dW tiles over different rows; each sends the half it does
    // not own (tiles i with (i >> 2) != wm) through LDS, adds the partner's copy of its own half, applies dW = scale (X^T A) + b (1^T A) ----
    {
        f32x4 *red = reinterpret_cast<f32x4 *>(Img);   // [wn][dest wm][tile & 3][r4][lane]
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if ((i >> 2) != wm) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
                    v[0] = dw[i][4 * r4], v[1] = dw[i][4 * r4 + 1], v[2] = dw[i][4 * r4 + 2], v[3] = dw[i][4 * r4 + 3];
                    red[((((wn * 2 + (i >> 2)) * 4 + (i & 3)) * 4 + r4) * 64) + lane] = v;
                }
            }
        }
        // column sums of the activation operand over the workgroup's rows: lane halves, then the two row halves
        double *ared = reinterpret_cast<double *>(Tred);   // [4 waves][32] doubles
        {
            const double a = sa + shfl_xor_f64(sa, 32);
            if (lh == 0) ared[wave * 32 + l31] = a;
        }
        __syncthreads();
        const float asum = (float)(ared[(0 * 2 + wn) * 32 + l31] + ared[(1 * 2 + wn) * 32 + l31]);
        float *wb = E.dwslab + (size_t)worker * KD * E.dw_ld + n0 + 32 * wn + l31;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if ((i >> 2) == wm) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 o = red[((((wn * 2 + wm) * 4 + (i & 3)) * 4 + r4) * 64) + lane];
                    const int c0 = i * 32 + 8 * r4 + 4 * lh;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(Tsc + c0), bb = *reinterpret_cast<const f32x4 *>(Tb + c0);
                    float *op = wb + (size_t)c0 * E.dw_ld;
                    // rows in (wm = 0) + (wm = 1) order whichever wave adds them
                    const float x0 = wm == 0 ? dw[i][4 * r4] + o[0] : o[0] + dw[i][4 * r4];
                    const float x1 = wm == 0 ? dw[i][4 * r4 + 1] + o[1] : o[1] + dw[i][4 * r4 + 1];
                    const float x2 = wm == 0 ? dw[i][4 * r4 + 2] + o[2] : o[2] + dw[i][4 * r4 + 2];
                    const float x3 = wm == 0 ? dw[i][4 * r4 + 3] + o[3] : o[3] + dw[i][4 * r4 + 3];
                    op[0] = fmaf(sc[0], x0, bb[0] * asum);
                    op[(size_t)E.dw_ld] = fmaf(sc[1], x1, bb[1] * asum);
                    op[(size_t)2 * E.dw_ld] = fmaf(sc[2], x2, bb[2] * asum);
                    op[(size_t)3 * E.dw_ld] = fmaf(sc[3], x3, bb[3] * asum);
                }
            }
        }
    }
    // ---- column statistics of the worker ----
    __syncthreads();
    double *dred = reinterpret_cast<double *>(Img);   // [4 waves][2][32]
    {
        const double a = s1 + shfl_xor_f64(s1, 32), b = s2 + shfl_xor_f64(s2, 32);
        if (lh == 0) dred[(wave * 2 + 0) * 32 + l31] = a, dred[(wave * 2 + 1) * 32 + l31] = b;
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid / BN, cl = tid % BN, w2 = cl >> 5, c5 = cl & 31;
        const double t = dred[((0 * 2 + w2) * 2 + which) * 32 + c5] + dred[((1 * 2 + w2) * 2 + which) * 32 + c5];
        E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
    }
    WSQ_STAMP(9)   // tail
#ifdef PNPP_STAMPS
    st_acc[6] = __builtin_amdgcn_s_memrealtime() - st_real0;
    if (st_on && lane == 0)
#pragma unroll
        for (int i = 0; i < 10; ++i) g_wsq_stamps[i] += st_acc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Second form of the same launch: the weight panel in REGISTERS, two tile images, the images filled by LDS-DMA.
// The stamps of gemm_wsq_kernel say where its time outside the matrix instructions goes: staging + fix-ups 8.5 % and the barriers
// around them (a serial phase per tile in which no wave multiplies), a dA product at 82 % of the issue rate (two 16-byte LDS reads per
// four MFMAs), prologue 9.6 %.
//   * A lane's B operand of the dA product never changes -- column n0 + 32 wn + l31, reduction indices 8 t + 4 lh + {0..3}: 128 values --
//     so it is loaded once into 128 registers (one wave per SIMD: 512 are there), scaled there; b W falls out of a lane-half exchange.
//   * The 64 KB of LDS the panel occupied hold a SECOND tile image, and in the folded form the image is a COPY of Z: tile i + 1 goes
//     global -> LDS directly (`buffer_load_dwordx4 ... lds`: a wave-instruction writes one 1-KiB row; no registers, no ds_write, no
//     VALU), issued between the MFMAs of tile i's dA product; its fix-ups follow tile i's dW product; one barrier per tile.
//     (A first version staged through registers: 128 + 64 operand registers pushed the panel into AGPRs, a v_accvgpr_read in front of
//     every MFMA: 53.6 us against 49.5.)
//   * The folded form is always valid here: a channel whose z-coefficient k = -istd c2 is (nearly) zero -- eval-mode statistics -- takes
//     k' = 1e-18 instead: its one-hot term dm / k' times the panel's g k' is g dm to rounding, and the Z term it should not have is
//     1e-18 g Z, far below the last bit of anything it is added to.
template <int KD, int NOUT>
__global__ void __launch_bounds__(256, 1)
gemm_wsq2_kernel(const AOperand A, const float *__restrict__ W, int ldw, int M, int ncol, const Epilogue E) {
    constexpr int Nout = NOUT;
    constexpr int BM = 64, BN = 64, DP = KD + 4, CT = KD / 32, NT4 = KD / 8;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Img0 = lds;                          // [2][BM][DP]: dZ images of two consecutive tiles
    float *Tsc = lds + 2 * BM * DP, *Tb = Tsc + KD, *Tred = Tb + KD;   // scale[KD], b[KD], scratch [4][64] floats / doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;

    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;
    const int ntiles = M / BM;
    int tile = worker;
#ifdef PNPP_STAMPS
    const bool st_on = blockIdx.x == 8 && wave == 0;
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: shader cycles / real time = the clock held
#endif

    const int q4 = 4 * lane;
    const __amdgpu_buffer_rsrc_t resZ = wsq_rsrc(A.z), resD = wsq_rsrc(A.a), resI = wsq_rsrc(A.arg), resP = wsq_rsrc(E.zp), resC = wsq_rsrc(E.c);
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.z), (short)0, 0, 0x00020000);
    unsigned oq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) oq[g] = 4u * ((unsigned)(32 * wm + 4 * lh + 8 * g) * (unsigned)Nout + (unsigned)(n0 + 32 * wn + l31));
    constexpr unsigned rowp = 4u * (unsigned)NOUT;

    float zq[16], zn[16], vh[16];   // z_{l-1} of this / the next tile; this tile's outputs (stored during the next tile's dA product)
    f32x4 gdm[2];                   // pooled gradient / arg-max rows of the tile whose image is being completed
    i32x4q garg[2];
    // row (wave + 4 i) of tile t -> its image: one wave-instruction, 64 lanes x 16 bytes = the row's 1 KiB
    // Issued as an asm statement: through the builtin, hipcc orders every later ds_read behind the DMA with an `s_waitcnt vmcnt(0)` (it
    // cannot see that the image being read and the image being filled are different halves of one LDS object) -- 170 cycles per row in
    // the dA product.  An asm load is absent from hipcc's counter bookkeeping (its own waits can only become stricter: vmcnt counts in
    // issue order); the kernel waits for its rows itself, once, behind the dW product.  M0 (the LDS destination) is written and
    // restored inside the statement.  A descriptor with zero records (past the last tile) drops the load.
    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
    auto dma_desc = [](const void *base, bool on) {
        const unsigned long long a = (unsigned long long)base;
        u32x4q d;
        d[0] = (unsigned)a, d[1] = (unsigned)(a >> 32) & 0xffffu, d[2] = on ? 0xfffffffeu : 0u, d[3] = 0x00020000u;
        return d;
    };
    const unsigned dma_voff = 16u * (unsigned)lane;
    auto dma_row = [&](u32x4q dZ, float *img, int t, int i) {
        const unsigned ldsa = (unsigned)(size_t)(lds_void *)(img + (wave + 4 * i) * DP);
        const unsigned soff = (unsigned)t * (BM * KD * 4u) + (unsigned)(wave + 4 * i) * (KD * 4u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(ldsa), "v"(dma_voff), "s"(dZ), "s"(soff)
                     : "memory");
    };
    const u32x4q dmaZ = dma_desc(A.z, true), dmaNull = dma_desc(A.z, false);
    auto fetch_p = [&](__amdgpu_buffer_rsrc_t rP, int t, int r) {
        zn[r] = wsq_load1(rP, oq[r >> 2] + (unsigned)(r & 3) * rowp, (unsigned)t * (unsigned)BM * rowp);
    };
    auto fetch_g = [&](__amdgpu_buffer_rsrc_t rD, __amdgpu_buffer_rsrc_t rI, int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned so = (unsigned)(2 * t + h) * (KD * 4u);
            gdm[h] = wsq_load4(rD, 4u * (unsigned)q4, so);
            garg[h] = wsq_load4i(rI, 4u * (unsigned)q4, so);
        }
    };
    // ---- everything the prologue reads is requested before its first wait ----
    const bool have0 = tile < ntiles;
    {
        const __amdgpu_buffer_rsrc_t z0 = have0 ? resZ : resNull, d0 = have0 ? resD : resNull, i0 = have0 ? resI : resNull, p0 = have0 ? resP : resNull;
        const int t0 = have0 ? tile : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) dma_row(have0 ? dmaZ : dmaNull, Img0, t0, i);
        fetch_g(d0, i0, t0);
#pragma unroll
        for (int r = 0; r < 16; ++r) fetch_p(p0, t0, r);
    }
    {   // per-channel tables: a' = g k', b (KD == 256 == blockDim: one channel per thread)
        const float g = A.cst[tid], mu = A.cst[A.C + tid], is = A.cst[2 * A.C + tid], c1 = A.cst[3 * A.C + tid], c2 = A.cst[4 * A.C + tid];
        float k = -is * c2;
        k = fabsf(k) >= 1e-18f ? k : 1e-18f;
        const float a = g * k;
        Tsc[tid] = a, Tb[tid] = -g * c1 - a * mu;
    }
    f32x4 fxq;   // fix-up multipliers 1 / k' of this lane's column group
    {
        const float *p = A.cst + q4;
        const float4 is = *reinterpret_cast<const float4 *>(p + 2 * A.C), c2 = *reinterpret_cast<const float4 *>(p + 4 * A.C);
        const float k0 = -is.x * c2.x, k1 = -is.y * c2.y, k2 = -is.z * c2.z, k3 = -is.w * c2.w;
        fxq[0] = 1.f / (fabsf(k0) >= 1e-18f ? k0 : 1e-18f), fxq[1] = 1.f / (fabsf(k1) >= 1e-18f ? k1 : 1e-18f);
        fxq[2] = 1.f / (fabsf(k2) >= 1e-18f ? k2 : 1e-18f), fxq[3] = 1.f / (fabsf(k3) >= 1e-18f ? k3 : 1e-18f);
    }
    float e_sc, e_sh;
    double e_mu, e_is;
    {
        const int col = n0 + 32 * wn + l31;
        e_sc = E.scale[col], e_sh = E.shift[col], e_mu = (double)E.mu[col], e_is = (double)E.istd[col];
    }
    // this lane's B operand: W[8 t + 4 lh + u][n0 + 32 wn + l31] (coalesced over l31)
    f32x4 pb[NT4];
    {
        const float *wc = W + (size_t)(4 * lh) * ldw + n0 + 32 * wn + l31;
#pragma unroll
        for (int t = 0; t < NT4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) pb[t][u] = wc[(size_t)(8 * t + u) * ldw];
    }
    // the fix-ups of a tile, by the wave whose rows they land on (that wave's DMA wrote them: its own counted wait orders them)
    auto fixups = [&](float *img) {
        float cur[2][4];
        bool hit[2][4];
        float *pp[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = garg[h][e];
                hit[h][e] = (unsigned)a < 32u && (a & 3) == wave && gdm[h][e] != 0.f;
                pp[h][e] = img + (32 * h + (hit[h][e] ? a : wave)) * DP + q4 + e;
                cur[h][e] = *pp[h][e];
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (hit[h][e]) *pp[h][e] = fmaf(fxq[e], gdm[h][e], cur[h][e]);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // image 0 has landed (and everything else of the prologue)
    fixups(Img0);
    {
        const bool have1 = tile + nworkers < ntiles;
        fetch_g(have1 ? resD : resNull, have1 ? resI : resNull, have1 ? tile + nworkers : 0);
    }
    __syncthreads();   // tables and image 0
    // scale the panel in registers; b W of this lane's column from its half of the reduction + the other lane half's
    float bw;
    {
        float bwp = 0.f;
#pragma unroll
        for (int t = 0; t < NT4; ++t) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(Tsc + 8 * t + 4 * lh), bb = *reinterpret_cast<const f32x4 *>(Tb + 8 * t + 4 * lh);
            bwp = fmaf(bb[0], pb[t][0], fmaf(bb[1], pb[t][1], fmaf(bb[2], pb[t][2], fmaf(bb[3], pb[t][3], bwp))));
            pb[t][0] *= sc[0], pb[t][1] *= sc[1], pb[t][2] *= sc[2], pb[t][3] *= sc[3];
        }
        const float o = __shfl_xor(bwp, 32, 64);
        bw = lh == 0 ? bwp + o : o + bwp;   // (lh = 0) + (lh = 1) in both halves
    }

    f32x16 dw[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) zq[r] = zn[r], vh[r] = 0.f;
    double s1 = 0.0, s2 = 0.0, sa = 0.0;
    __amdgpu_buffer_rsrc_t pC = resNull;   // the held outputs' destination (nothing is held before the first tile)
    unsigned prev_off = 0;
    int par = 0;
    WSQ_STAMP(8)   // prologue
    for (; tile < ntiles; tile += nworkers) {
        WSQ_STAMP(0)
        float *img = Img0 + par * (BM * DP), *imgn = Img0 + (par ^ 1) * (BM * DP);
        const bool more1 = tile + nworkers < ntiles, more2 = tile + 2 * nworkers < ntiles;
        const int t1 = more1 ? tile + nworkers : 0, t2 = more2 ? tile + 2 * nworkers : 0;
        const __amdgpu_buffer_rsrc_t nZ = more1 ? resZ : resNull, nP = more1 ? resP : resNull, nD2 = more2 ? resD : resNull, nI2 = more2 ? resI : resNull;
        const float *arow = img + (32 * wm + l31) * DP + 4 * lh;
        const float *dcol = img + (32 * wm + 4 * lh) * DP + l31;
        // ---- dA = X W': one 16-byte LDS read per four MFMAs, the B operand from registers.  Every memory operation of the tile is
        // issued here: the previous tile's output stores, the next tile's z_{l-1}, the next tile's image rows (LDS-DMA) ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bw;
        {
            float4 fa[2];
            fa[0] = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
            for (int t = 0; t < NT4; ++t) {
                if (!(WSQ_EXP & 128)) {
                if (t < 16) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vh[t]), pC, (int)(oq[t >> 2] + (unsigned)(t & 3) * rowp), (int)prev_off, 0);
                else fetch_p(nP, t1, t - 16);
                }
                if (!(WSQ_EXP & 64) && (t & 1) == 0) dma_row(more1 ? dmaZ : dmaNull, imgn, t1, t >> 1);
                if (t + 1 < NT4) fa[(t + 1) & 1] = *reinterpret_cast<const float4 *>(arow + 8 * (t + 1));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1].x, pb[t][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1].y, pb[t][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1].z, pb[t][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1].w, pb[t][3], acc, 0, 0, 0);
            }
        }
        WSQ_STAMP(3)   // dA product
        // ---- dW += X^T a with the epilogue (no memory operation in here) ----
        {
            float t1s = 0.f, t2s = 0.f, ta = 0.f;
            float fd[2][CT];
            auto ldw = [&](int buf, int s) {
#pragma unroll
                for (int i = 0; i < CT; ++i) fd[buf][i] = dcol[((s & 3) + 8 * (s >> 2)) * DP + 32 * i];
            };
            ldw(0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float z0 = zq[s];
                const float a0 = fmaf(z0, e_sc, e_sh);
                const float bact = fmaxf(a0, 0.f);
                const float v = a0 > 0.f ? acc[s] : 0.f;
                vh[s] = v;
                t1s += v;
                t2s = fmaf(v, z0, t2s);
                ta += bact;
                if (s + 1 < 16) ldw((s + 1) & 1, s + 1);
#pragma unroll
                for (int i = 0; i < CT; ++i) dw[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[s & 1][i], bact, dw[i], 0, 0, 0);
            }
            const double d1 = (double)t1s;
            s1 += d1, s2 += e_is * ((double)t2s - e_mu * d1), sa += (double)ta;
            pC = resC, prev_off = (unsigned)tile * (unsigned)BM * rowp;
#pragma unroll
            for (int r = 0; r < 16; ++r) zq[r] = zn[r];
        }
        WSQ_STAMP(4)   // dW product + epilogue
        // the next tile's rows were requested a whole dW product ago: this wait is for instructions long complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more1) {
            fixups(imgn);
            fetch_g(nD2, nI2, t2);
        }
        WSQ_STAMP(1)   // wait + fix-ups
        __syncthreads();   // tile i's image is free, tile i + 1's is complete
        WSQ_STAMP(5)   // barrier
        par ^= 1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)   // the last tile's outputs
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vh[r]), pC, (int)(oq[r >> 2] + (unsigned)(r & 3) * rowp), (int)prev_off, 0);

    // ---- tail: as gemm_wsq_kernel (the images are free) ----
    {
        f32x4 *red = reinterpret_cast<f32x4 *>(Img0);   // [wn][dest wm][tile & 3][r4][lane]
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if ((i >> 2) != wm) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
                    v[0] = dw[i][4 * r4], v[1] = dw[i][4 * r4 + 1], v[2] = dw[i][4 * r4 + 2], v[3] = dw[i][4 * r4 + 3];
                    red[((((wn * 2 + (i >> 2)) * 4 + (i & 3)) * 4 + r4) * 64) + lane] = v;
                }
            }
        }
        double *ared = reinterpret_cast<double *>(Tred);   // [4 waves][32] doubles
        {
            const double a = sa + shfl_xor_f64(sa, 32);
            if (lh == 0) ared[wave * 32 + l31] = a;
        }
        __syncthreads();
        const float asum = (float)(ared[(0 * 2 + wn) * 32 + l31] + ared[(1 * 2 + wn) * 32 + l31]);
        float *wb = E.dwslab + (size_t)worker * KD * E.dw_ld + n0 + 32 * wn + l31;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if ((i >> 2) == wm) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 o = red[((((wn * 2 + wm) * 4 + (i & 3)) * 4 + r4) * 64) + lane];
                    const int c0 = i * 32 + 8 * r4 + 4 * lh;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(Tsc + c0), bb = *reinterpret_cast<const f32x4 *>(Tb + c0);
                    float *op = wb + (size_t)c0 * E.dw_ld;
                    op[0] = fmaf(sc[0], dw[i][4 * r4] + o[0], bb[0] * asum);
                    op[(size_t)E.dw_ld] = fmaf(sc[1], dw[i][4 * r4 + 1] + o[1], bb[1] * asum);
                    op[(size_t)2 * E.dw_ld] = fmaf(sc[2], dw[i][4 * r4 + 2] + o[2], bb[2] * asum);
                    op[(size_t)3 * E.dw_ld] = fmaf(sc[3], dw[i][4 * r4 + 3] + o[3], bb[3] * asum);
                }
            }
        }
    }
    __syncthreads();
    double *dred = reinterpret_cast<double *>(Img0);   // [4 waves][2][32]
    {
        const double a = s1 + shfl_xor_f64(s1, 32), b = s2 + shfl_xor_f64(s2, 32);
        if (lh == 0) dred[(wave * 2 + 0) * 32 + l31] = a, dred[(wave * 2 + 1) * 32 + l31] = b;
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid / BN, cl = tid % BN, w2 = cl >> 5, c5 = cl & 31;
        const double t = dred[((0 * 2 + w2) * 2 + which) * 32 + c5] + dred[((1 * 2 + w2) * 2 + which) * 32 + c5];
        E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
    }
    WSQ_STAMP(9)   // tail
#ifdef PNPP_STAMPS
    st_acc[6] = __builtin_amdgcn_s_memrealtime() - st_real0;
    if (st_on && lane == 0)
#pragma unroll
        for (int i = 0; i < 10; ++i) g_wsq_stamps[i] += st_acc[i];
#endif
}

// A/B switch: PNPP_NO_WSQ=1 keeps this launch on gemm_ws_kernel<256, ..., dW>
static bool wsq_on() {
    static int cached = -1;
    if (cached < 0) {
        const char *v = getenv("PNPP_NO_WSQ");
        cached = (v && atoi(v) != 0) ? 0 : 1;
    }
    return cached != 0;
}

bool wsq_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E) {
    if (!wsq_on() || matmul_precision() != 0) return false;
    if (M < 8192 || M % 64 != 0 || Nout != 128 || Kd != 256 || A.mode != A_DZ_POOL || A.K != 32) return false;
    if (E.mode != E_MASK_STATS || !E.dwslab || E.dw_ld != Nout || E.ldc != Nout) return false;
    if (B.trans || B.perm_D >= 0 || (B.rows > 0 && B.rows != Kd) || B.ldb < Nout) return false;
    if (A.lda != Kd || A.C != Kd) return false;
    if ((((uintptr_t)A.a | (uintptr_t)A.arg | (uintptr_t)A.z | (uintptr_t)E.zp | (uintptr_t)E.c | (uintptr_t)A.cst) & 15) != 0) return false;
    if ((unsigned long long)M * (unsigned)Kd * 4ull >= 0xfffffff0ull) return false;   // 32-bit buffer offsets
    return true;
}

bool try_launch_wsq(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                    int *dw_slabs) {
    *rc = PNPP_OK;
    if (!dw_slabs || !wsq_applies(A, B, M, Nout, Kd, E)) return false;
    constexpr int KD = 256;
    const int ncol = Nout / 64, tiles = M / 64;
    int workers = 256 / ncol;   // one workgroup per CU
    if (workers > tiles) workers = tiles;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (workers < 1) workers = 1;
    if (nslab) *nslab = workers;
    *dw_slabs = workers;
    constexpr size_t lds = ((size_t)64 * KD + 64 * (KD + 4) + 2 * KD + 64 + 2 * 4 * 64) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    ProfScope ps(st, "gemm_wsq_kernel<%d,A%d> M=%d N=%d K=%d grid=%dx1", Kd, A.mode, M, Nout, Kd, workers * ncol);
    // 1: weight panel in LDS, one image (default); 2: panel in registers, two images filled by LDS-DMA.  Form 2's stamped wave ends
    // 1.6 us earlier inside the step (5 - 7 % fewer shader cycles at a 1 - 2 % lower clock) and the launch takes the same time at every
    // batch size (49.3 / 49.8 us at 32 clouds, 93.3 k / 93.4 k clouds/s at 512): DESIGN section 9
    static const int form = getenv("PNPP_WSQ_FORM") ? atoi(getenv("PNPP_WSQ_FORM")) : 1;
    if (form == 2) {
        constexpr size_t lds2 = ((size_t)2 * 64 * (KD + 4) + 2 * KD + 2 * 4 * 64) * sizeof(float);
        static_assert(lds2 <= 160 * 1024, "LDS budget");
        auto kfn2 = gemm_wsq2_kernel<KD, 128>;
        static bool granted2 = false;
        if (!granted2) {
            (void)hipFuncSetAttribute((const void *)kfn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            granted2 = true;
        }
        hipLaunchKernelGGL(kfn2, dim3(workers * ncol), dim3(256), lds2, st, A, B.b, B.ldb, M, ncol, E);
    } else {
    auto kfn = gemm_wsq_kernel<KD, 128>;
    static bool granted = false;
    if (!granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(256), lds, st, A, B.b, B.ldb, M, ncol, E);
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsq: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

#ifdef PNPP_STAMPS
#define PNPP_STAMPS_BIT 64u
#else
#define PNPP_STAMPS_BIT 0u
#endif
unsigned wsq_build_flags() { return ((WSQ_EXP != 0) ? 8u : 0u) | ((WSQ_PLAIN != 0) ? 16u : 0u) | PNPP_STAMPS_BIT; }

}  // namespace pnpp

#ifdef PNPP_STAMPS
extern "C" int pnpp_debug_wsq_stamps(unsigned long long *out16, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_wsq_stamps), z, sizeof(z));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out16, HIP_SYMBOL(pnpp::g_wsq_stamps), 16 * sizeof(unsigned long long));
    }
    return 0;
}
#endif
