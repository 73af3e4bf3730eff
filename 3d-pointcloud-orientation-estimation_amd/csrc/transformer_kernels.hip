// transformer_kernels.hip -- forward kernels of the point-transformer configuration (SURVEY section 8 f-4,
// reference models/point_transformer.py:4-20 = nn.TransformerEncoderLayer(d_model=64, nhead=4, batch_first=True) x 6).
//
//   linear_smallk_kernel   input_proj: y = x W^T + b for a reduction length of at most 8 (xyz -> 64 channels)
//   attention_fwd_kernel   softmax(q k^T / sqrt(d_head)) v per (cloud, head), flash style: the N x N score matrix is
//                          never written; float32 MFMA (v_mfma_f32_32x32x2_f32), online softmax, log-sum-exp kept
//   add_layernorm_kernel   y = LayerNorm(x + r)  (post-norm residual blocks)
//   mean_points_kernel     mean over the points of a cloud
//
// The dense projections (in_proj, out_proj, linear1 + ReLU, linear2, fc_out) go through pnpp_fc_forward.
#include "common.h"
#include "kernels.h"

namespace pnpp {

constexpr int SMALLK_MAX = 8;

__global__ void __launch_bounds__(256) linear_smallk_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ b, int M, int K, int N,
                                                            float *__restrict__ y) {
    // lane = output channel (coalesced stores), rows strided over the grid; K <= 8 inputs per row are broadcast loads
    const int n = threadIdx.x % 64, rl = threadIdx.x / 64;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int c = n0 + n;
        float wr[SMALLK_MAX];
#pragma unroll
        for (int k = 0; k < SMALLK_MAX; ++k) wr[k] = (c < N && k < K) ? w[(size_t)c * K + k] : 0.f;
        const float bias = (c < N && b) ? b[c] : 0.f;
        for (int m = blockIdx.x * 4 + rl; m < M; m += gridDim.x * 4) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < SMALLK_MAX; ++k)
                if (k < K) acc = fmaf(x[(size_t)m * K + k], wr[k], acc);
            if (c < N) y[(size_t)m * N + c] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention forward, head dimension 16.  qkv (B, N, 3E) row-major with the in_proj bias already added, E = H * 16:
// q = qkv[..., 0:E], k = qkv[..., E:2E], v = qkv[..., 2E:3E]; head h owns columns h*16 .. h*16+15 of each.
// Workgroup = 128 queries of one (cloud, head): 4 waves x 32 queries.  Keys / values stream through LDS in blocks of 32.
//
// Layouts (v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
// D[row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31] in register r):
//   S^T = K Q^T : A = key block (i = key, k = dim), B = Q^T (k = dim, j = query)  ->  lane = query, register r = key
//                 kappa(r, lh) = (r & 3) + 8 (r >> 2) + 4 lh: a query's row of scores sits in the 16 registers of two lanes,
//                 so the row maximum and row sum are register reductions plus one cross-half shuffle;
//   O^T = V^T P^T: A = V^T (i = dim, k = key), B = P^T (k = key, j = query): the k index of an MFMA step is free to name
//                 any key as long as A and B agree, and step s with k = lh naming key kappa(s, lh) makes the B operand
//                 exactly score register s -- the probabilities never move between lanes.
// ---------------------------------------------------------------------------------------------
// Dropout on the attention weights (nn.MultiheadAttention(dropout=p) in train mode): one keep bit per (cloud, head, query,
// key), Bernoulli(1 - p), generated ONCE per layer call by a counter-based generator (Philox4x32-10, a pure function of
// seed, stream id and the element index -- the backward pass regenerates the identical bits) and stored bit-packed in
// both orientations: mask[bh][query][key / 32] (bit = key % 32) for the kernels whose lane is a query, and
// maskT[bh][key][query / 32] (bit = query % 32) for the dK / dV kernel whose lane is a key.  1/8 byte per score in each
// orientation; the float scores themselves are still never stored.
__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                        unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// one wave per pair of 32 x 32 tiles: lane = (tile half, row); 8 Philox calls give the row's 32 uniform words
__global__ void __launch_bounds__(256)
attention_dropout_mask_kernel(unsigned seed_lo, unsigned seed_hi, unsigned str_lo, unsigned str_hi,
                              unsigned long long *__restrict__ str_dev, int N, unsigned thresh, unsigned *__restrict__ mask,
                              unsigned *__restrict__ maskT) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    // stream id = (str_hi:str_lo) [+ the call counter in device memory: a captured step draws fresh masks on every replay];
    // a call owns the eight Philox counter words 8 * id .. 8 * id + 7, so consecutive calls share none
    unsigned long long sid = ((unsigned long long)str_hi << 32) | str_lo;
    if (str_dev) sid += str_dev[0];
    sid <<= 3;
    const int nw = N / 32;                                  // words per row
    const size_t wave_id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t tile = wave_id * 2 + lh;                   // (bh, query block, key block) flattened
    const size_t tiles_per_bh = (size_t)nw * nw;
    const size_t bh = tile / tiles_per_bh;
    const int qb = (int)((tile % tiles_per_bh) / nw), kb = (int)(tile % nw);
    const int q = qb * 32 + l31;
    unsigned w = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsigned r[4];
        const unsigned long long sc = sid + (unsigned long long)c;
        philox4((unsigned)(q * nw + kb), (unsigned)bh, (unsigned)sc, (unsigned)(sc >> 32), seed_lo, seed_hi, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (r[e] >= thresh ? 1u : 0u) << (4 * c + e);   // keep with probability 1 - p
    }
    mask[(bh * N + q) * nw + kb] = w;
    // transpose the 32 x 32 bit tile of this half-wave: word j of the transposed tile = bit j of every row
    unsigned wt = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const unsigned long long bal = __ballot((w >> j) & 1u);
        const unsigned half = lh ? (unsigned)(bal >> 32) : (unsigned)bal;
        if (l31 == j) wt = half;
    }
    maskT[(bh * N + kb * 32 + l31) * nw + qb] = wt;
    if (str_dev) {   // post-increment: every workgroup has read str_dev[0] before it takes a ticket; the last one bumps the counter
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(&str_dev[1], 1ull);
            if (t == (unsigned long long)gridDim.x - 1ull) {
                str_dev[1] = 0ull;
                str_dev[0] += 1ull;
            }
        }
    }
}

constexpr int ATT_DH = 16;
constexpr int ATT_KP = 17;  // LDS pitch of the key / value tiles

__global__ void __launch_bounds__(256)
attention_fwd_kernel(const float *__restrict__ qkv, int N, int Nv, int H, float scale, const unsigned *__restrict__ mask,
                     float keep_scale, float *__restrict__ out, float *__restrict__ lse) {
    __shared__ float Ks[2][32][ATT_KP], Vs[2][32][ATT_KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, E = H * ATT_DH, ld = 3 * E;
    const float *base = qkv + (size_t)b * N * ld;
    const int q = blockIdx.x * 128 + wave * 32 + l31;  // this lane's query (N % 128 == 0: always valid)

    float qreg[8];  // Q^T operand: dims 2s + lh of this lane's query, pre-scaled (1/sqrt(16) is exact)
#pragma unroll
    for (int s = 0; s < 8; ++s) qreg[s] = base[(size_t)q * ld + h * ATT_DH + 2 * s + lh] * scale;

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging map: threads 0..127 load the key tile, 128..255 the value tile, one float4 each
    const int st_row = (tid & 127) >> 2, st_c4 = 4 * (tid & 3);
    const float *st_src = base + (tid < 128 ? E : 2 * E) + h * ATT_DH + st_c4;
    auto fetch = [&](int kb) { return *reinterpret_cast<const float4 *>(st_src + (size_t)(kb * 32 + st_row) * ld); };
    auto put = [&](const float4 &v, int buf) {
        float *d = (tid < 128 ? &Ks[buf][st_row][st_c4] : &Vs[buf][st_row][st_c4]);
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    };
    // N rows are allocated per cloud (a multiple of 128), the first Nv of them are points: keys beyond Nv get no weight
    // (their scores are set to -inf before the softmax), key blocks that hold no point are not visited at all
    const int nkb = N / 32, nkbv = (Nv + 31) / 32;
    put(fetch(0), 0);
    __syncthreads();
    const float vmask = l31 < ATT_DH ? 1.f : 0.f;  // rows 16..31 of V^T do not exist
    for (int kb = 0; kb < nkbv; ++kb) {
        const int buf = kb & 1;
        const float4 nxt = fetch(min(kb + 1, nkbv - 1));  // in flight while this block is computed
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][l31][2 * t + lh], qreg[t], s, 0, 0, 0);
        if ((kb + 1) * 32 > Nv) {   // the last block of a cloud whose size is not a multiple of 32
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < Nv ? s[r] : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first block
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            rs += s[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;   // the softmax normalisation is over the undropped weights
        m_run = m_new;
        if (mask) {                   // dropout on the weights: keep bit of (query, key kappa(r, lh)), scaled by 1 / (1 - p)
            const unsigned mw = mask[(((size_t)b * H + h) * N + q) * nkb + kb];
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = ((mw >> ((r & 3) + 8 * (r >> 2) + 4 * lh)) & 1u) ? s[r] * keep_scale : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] *= alpha;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int key = (t & 3) + 8 * (t >> 2) + 4 * lh;
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][key][l31 & (ATT_DH - 1)] * vmask, s[t], o, 0, 0, 0);
        }
        if (kb + 1 < nkbv) put(nxt, buf ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    const float inv = 1.f / l_run;
    float *orow = out + ((size_t)b * N + q) * E + h * ATT_DH;
#pragma unroll
    for (int r = 0; r < 8; ++r) orow[(r & 3) + 8 * (r >> 2) + 4 * lh] = o[r] * inv;
    if (lse && lh == 0) lse[((size_t)b * H + h) * N + q] = m_run + logf(l_run);
}

// y = LayerNorm(x + r) over the last dimension E (64 or 128); one wave per row, two-pass statistics in float64
__global__ void __launch_bounds__(256) add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ r,
                                                            const float *__restrict__ w, const float *__restrict__ bias,
                                                            int M, int E, float eps, float *__restrict__ y) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[2];
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < E ? x[(size_t)row * E + c] + (r ? r[(size_t)row * E + c] : 0.f) : 0.f;
        sum += (double)v[j];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += shfl_xor_f64(sum, m);
    const double mu = sum / (double)E;
    double sq = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (lane + 64 * j < E) sq += ((double)v[j] - mu) * ((double)v[j] - mu);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += shfl_xor_f64(sq, m);
    const double is = 1.0 / sqrt(sq / (double)E + (double)eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = lane + 64 * j;
        if (c < E) y[(size_t)row * E + c] = (float)(((double)v[j] - mu) * is * (double)w[c] + (double)bias[c]);
    }
}

// y[b][c] = mean_n x[b][n][c]; one workgroup per cloud, fixed-order float64 sums
__global__ void __launch_bounds__(256) mean_points_kernel(const float *__restrict__ x, int N, int E, float *__restrict__ y) {
    __shared__ double red[4][64];
    const int b = blockIdx.x, c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int c0 = 0; c0 < E; c0 += 64) {
        double acc = 0.0;
        if (c0 + c < E)
            for (int n = rl; n < N; n += 4) acc += (double)x[((size_t)b * N + n) * E + c0 + c];
        red[rl][c] = acc;
        __syncthreads();
        if (rl == 0 && c0 + c < E) y[(size_t)b * E + c0 + c] = (float)(((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / (double)N);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// backward kernels (same MFMA layout algebra as the forward: an accumulator tile with lane = X and register = Y is,
// unchanged, the B operand of a product that contracts over Y and whose output columns are X)
// ---------------------------------------------------------------------------------------------

// input_proj backward: dW (N,K) = dy^T x and db (N) = column sums of dy; one [N][16] partial per 256 rows
// (columns 0..K-1: dW, column K: db), combined by slab_reduce.  x is data: no dx.
__global__ void __launch_bounds__(256) linear_smallk_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                int M, int K, int N, float *__restrict__ slab) {
    __shared__ float red[4][64][SMALLK_MAX + 1];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, m0 = blockIdx.x * 256;
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int c = c0 + cl;
        float acc[SMALLK_MAX + 1];
#pragma unroll
        for (int k = 0; k <= SMALLK_MAX; ++k) acc[k] = 0.f;
        for (int r = rl; r < 256; r += 4) {
            const int m = m0 + r;
            if (m >= M) break;
            const float g = c < N ? dy[(size_t)m * N + c] : 0.f;
#pragma unroll
            for (int k = 0; k < SMALLK_MAX; ++k)
                if (k < K) acc[k] = fmaf(g, x[(size_t)m * K + k], acc[k]);
            acc[SMALLK_MAX] += g;
        }
#pragma unroll
        for (int k = 0; k <= SMALLK_MAX; ++k) red[rl][cl][k] = acc[k];
        __syncthreads();
        for (int f = threadIdx.x; f < 64 * (SMALLK_MAX + 1); f += 256) {
            const int ch = f / (SMALLK_MAX + 1), k = f % (SMALLK_MAX + 1);
            if (c0 + ch < N && (k < K || k == SMALLK_MAX)) {
                const float t = (red[0][ch][k] + red[1][ch][k]) + (red[2][ch][k] + red[3][ch][k]);
                slab[((size_t)blockIdx.x * N + c0 + ch) * 16 + (k == SMALLK_MAX ? K : k)] = t;
            }
        }
        __syncthreads();
    }
}

// dQ: a wave owns 32 queries and walks the key blocks.  S^T and dP^T tiles are lane = query, register = key, so the
// per-query log-sum-exp and D = rowsum(dO * O) are per-lane scalars and dS^T is directly the B operand of dQ^T = K^T dS^T.
// Also writes D (B,H,N) for the dK/dV kernel.
__global__ void __launch_bounds__(256)
attention_bwd_dq_kernel(const float *__restrict__ qkv, const float *__restrict__ o, const float *__restrict__ d_o,
                        const float *__restrict__ lse, int N, int Nv, int H, float scale, const unsigned *__restrict__ mask,
                        float keep_scale, float *__restrict__ dqkv, float *__restrict__ dsum) {
    __shared__ float Ks[2][32][ATT_KP], Vs[2][32][ATT_KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, E = H * ATT_DH, ld = 3 * E;
    const float *base = qkv + (size_t)b * N * ld;
    const int q = blockIdx.x * 128 + wave * 32 + l31;
    float qreg[8], doreg[8], dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int d = h * ATT_DH + 2 * s + lh;
        qreg[s] = base[(size_t)q * ld + d] * scale;
        doreg[s] = d_o[((size_t)b * N + q) * E + d];
        dpart = fmaf(doreg[s], o[((size_t)b * N + q) * E + d], dpart);
    }
    const float D = dpart + __shfl_xor(dpart, 32, 64);
    const float L = lse[((size_t)b * H + h) * N + q];
    if (lh == 0) dsum[((size_t)b * H + h) * N + q] = D;

    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    const int st_row = (tid & 127) >> 2, st_c4 = 4 * (tid & 3);
    const float *st_src = base + (tid < 128 ? E : 2 * E) + h * ATT_DH + st_c4;
    auto fetch = [&](int kb) { return *reinterpret_cast<const float4 *>(st_src + (size_t)(kb * 32 + st_row) * ld); };
    auto put = [&](const float4 &v, int buf) {
        float *d = (tid < 128 ? &Ks[buf][st_row][st_c4] : &Vs[buf][st_row][st_c4]);
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    };
    const int nkb = N / 32, nkbv = (Nv + 31) / 32;   // keys beyond the cloud's Nv points carry no weight (P = 0)
    put(fetch(0), 0);
    __syncthreads();
    const float vmask = l31 < ATT_DH ? 1.f : 0.f;
    for (int kb = 0; kb < nkbv; ++kb) {
        const int buf = kb & 1;
        const float4 nxt = fetch(min(kb + 1, nkbv - 1));
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][l31][2 * t + lh], qreg[t], s, 0, 0, 0);     // S^T  = K Q^T
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][l31][2 * t + lh], doreg[t], dp, 0, 0, 0);  // dP^T = V dO^T
        }
        if (mask) {  // d(dropped weights) -> d(weights): the same keep bits and scale as the forward
            const unsigned mw = mask[(((size_t)b * H + h) * N + q) * nkb + kb];
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = ((mw >> ((r & 3) + 8 * (r >> 2) + 4 * lh)) & 1u) ? dp[r] * keep_scale : 0.f;
        }
        if ((kb + 1) * 32 > Nv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < Nv ? s[r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __expf(s[r] - L) * (dp[r] - D);  // dS^T = P^T * (dP^T - D)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int key = (t & 3) + 8 * (t >> 2) + 4 * lh;
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][key][l31 & (ATT_DH - 1)] * vmask, s[t], dq, 0, 0, 0);
        }
        if (kb + 1 < nkbv) put(nxt, buf ^ 1);
        __syncthreads();
    }
    float *drow = dqkv + ((size_t)b * N + q) * ld + h * ATT_DH;
#pragma unroll
    for (int r = 0; r < 8; ++r) drow[(r & 3) + 8 * (r >> 2) + 4 * lh] = dq[r] * scale;
}

// dK, dV: a wave owns 32 keys and walks the query blocks.  S and dP tiles are lane = key, register = query; P and dS are
// directly the B operands of dV^T = dO^T P and dK^T = (scale Q)^T dS.  Per-query L and D come from LDS (one per register).
__global__ void __launch_bounds__(256)
attention_bwd_dkv_kernel(const float *__restrict__ qkv, const float *__restrict__ d_o, const float *__restrict__ lse,
                         const float *__restrict__ dsum, int N, int Nv, int H, float scale, const unsigned *__restrict__ maskT,
                         float keep_scale, float *__restrict__ dqkv) {
    __shared__ float Qs[2][32][ATT_KP], Gs[2][32][ATT_KP], Ls[2][32], Ds[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, E = H * ATT_DH, ld = 3 * E;
    const float *base = qkv + (size_t)b * N * ld;
    const int key = blockIdx.x * 128 + wave * 32 + l31;
    float kreg[8], vreg[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        kreg[s] = base[(size_t)key * ld + E + h * ATT_DH + 2 * s + lh];
        vreg[s] = base[(size_t)key * ld + 2 * E + h * ATT_DH + 2 * s + lh];
    }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[r] = 0.f, dv[r] = 0.f;
    // staging: threads 0..127 the (scaled) query tile, 128..255 the dO tile; threads 0..31 / 32..63 also L / D
    const int st_row = (tid & 127) >> 2, st_c4 = 4 * (tid & 3);
    const float *q_src = base + h * ATT_DH + st_c4;
    const float *g_src = d_o + (size_t)b * N * E + h * ATT_DH + st_c4;
    const float *l_src = lse + ((size_t)b * H + h) * N, *d_src = dsum + ((size_t)b * H + h) * N;
    auto fetch = [&](int qb) {
        const int row = qb * 32 + st_row;
        return tid < 128 ? *reinterpret_cast<const float4 *>(q_src + (size_t)row * ld) : *reinterpret_cast<const float4 *>(g_src + (size_t)row * E);
    };
    auto fetch_ld = [&](int qb) { return tid < 32 ? l_src[qb * 32 + tid] : (tid < 64 ? d_src[qb * 32 + tid - 32] : 0.f); };
    auto put = [&](const float4 &v, float x, int buf) {
        if (tid < 128) {
            float *d = &Qs[buf][st_row][st_c4];
            d[0] = v.x * scale, d[1] = v.y * scale, d[2] = v.z * scale, d[3] = v.w * scale;
        } else {
            float *d = &Gs[buf][st_row][st_c4];
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
        }
        if (tid < 32) Ls[buf][tid] = x;
        else if (tid < 64) Ds[buf][tid - 32] = x;
    };
    // queries beyond the cloud's Nv points are padding: their dO is zero, so they add nothing and their blocks are skipped;
    // keys beyond Nv got no weight in the forward pass: their dK / dV rows are written as zeros
    const int nqb = N / 32, nqbv = (Nv + 31) / 32;
    put(fetch(0), fetch_ld(0), 0);
    __syncthreads();
    const float vmask = l31 < ATT_DH ? 1.f : 0.f;
    const float kvalid = key < Nv ? 1.f : 0.f;
    for (int qb = 0; qb < nqbv; ++qb) {
        const int buf = qb & 1;
        const float4 nxt = fetch(min(qb + 1, nqbv - 1));
        const float nxt_ld = fetch_ld(min(qb + 1, nqbv - 1));
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[buf][l31][2 * t + lh], kreg[t], s, 0, 0, 0);   // S  = (scale Q) K^T
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[buf][l31][2 * t + lh], vreg[t], dp, 0, 0, 0);  // dP = dO V^T
        }
        const unsigned mw = maskT ? maskT[(((size_t)b * H + h) * N + key) * nqb + qb] : 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = (r & 3) + 8 * (r >> 2) + 4 * lh;       // the query this register belongs to
            const float keep = ((mw >> qi) & 1u) ? keep_scale : 0.f;
            const float pr = __expf(s[r] - Ls[buf][qi]);          // P
            dp[r] = pr * (dp[r] * keep - Ds[buf][qi]);            // dS = P * (d(dropped P) * keep / (1 - p) - D)
            s[r] = pr * keep;                                     // dropped P, the operand of dV
        }
        if ((qb + 1) * 32 > Nv) {   // the cloud's last, partial query block: padding queries contribute nothing
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= Nv) dp[r] = 0.f, s[r] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int qi = (t & 3) + 8 * (t >> 2) + 4 * lh;
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[buf][qi][l31 & (ATT_DH - 1)] * vmask, s[t], dv, 0, 0, 0);   // dV^T += dO^T P
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[buf][qi][l31 & (ATT_DH - 1)] * vmask, dp[t], dk, 0, 0, 0);  // dK^T += (scale Q)^T dS
        }
        if (qb + 1 < nqbv) put(nxt, nxt_ld, buf ^ 1);
        __syncthreads();
    }
    float *drow = dqkv + ((size_t)b * N + key) * ld + h * ATT_DH;
#pragma unroll
    for (int r = 0; r < 8; ++r) {   // padding keys took no part in the forward softmax: their rows are written as zeros
        const int d = (r & 3) + 8 * (r >> 2) + 4 * lh;
        drow[E + d] = dk[r] * kvalid;
        drow[2 * E + d] = dv[r] * kvalid;
    }
}

// LayerNorm(x + r) backward: u = x + r, xh = (u - mu) rstd (recomputed), g = dy * w:
//   du = rstd * (g - mean(g) - xh * mean(g * xh))  (= dx = dr);  dw = sum_rows dy * xh;  db = sum_rows dy
// One wave per row; the column sums go to one [2][E] partial per workgroup of 64 rows (slab_reduce combines them).
__global__ void __launch_bounds__(256) add_layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ r,
                                                                const float *__restrict__ w, const float *__restrict__ dy,
                                                                int M, int E, float eps, float *__restrict__ du,
                                                                float *__restrict__ slab) {
    __shared__ float red[4][2][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float aw[2] = {0.f, 0.f}, ab[2] = {0.f, 0.f};
    for (int it = 0; it < 16; ++it) {
        const int row = blockIdx.x * 64 + it * 4 + wv;
        if (row >= M) break;
        float v[2], g[2];
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < E ? x[(size_t)row * E + c] + (r ? r[(size_t)row * E + c] : 0.f) : 0.f;
            g[j] = c < E ? dy[(size_t)row * E + c] : 0.f;
            sum += (double)v[j];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sum += shfl_xor_f64(sum, m);
        const double mu = sum / (double)E;
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane + 64 * j < E) sq += ((double)v[j] - mu) * ((double)v[j] - mu);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += shfl_xor_f64(sq, m);
        const double rstd = 1.0 / sqrt(sq / (double)E + (double)eps);
        double xh[2], gw[2], s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = lane + 64 * j;
            xh[j] = c < E ? ((double)v[j] - mu) * rstd : 0.0;
            gw[j] = c < E ? (double)g[j] * (double)w[c] : 0.0;
            s1 += gw[j], s2 += gw[j] * xh[j];
            aw[j] += (float)((double)g[j] * xh[j]);
            ab[j] += g[j];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s1 += shfl_xor_f64(s1, m), s2 += shfl_xor_f64(s2, m);
        s1 /= (double)E, s2 /= (double)E;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = lane + 64 * j;
            if (c < E) du[(size_t)row * E + c] = (float)(rstd * (gw[j] - s1 - xh[j] * s2));
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) red[wv][0][lane + 64 * j] = aw[j], red[wv][1][lane + 64 * j] = ab[j];
    __syncthreads();
    for (int f = threadIdx.x; f < 2 * E; f += 256) {
        const int which = f / E, c = f % E;
        slab[((size_t)blockIdx.x * 2 + which) * E + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// x.mean(dim=1) backward: dx[b][n][c] = dy[b][c] / N
__global__ void __launch_bounds__(256) mean_points_bwd_kernel(const float *__restrict__ dy, int N, int E, size_t total,
                                                              float *__restrict__ dx) {
    const float inv = 1.f / (float)N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bn = i / E;
        dx[i] = dy[(bn / N) * E + (i - bn * E)] * inv;
    }
}

}  // namespace pnpp

using namespace pnpp;

extern "C" int pnpp_linear_smallk(const float *x, const float *w, const float *b, int M, int K, int N, float *y, void *stream) {
    PNPP_REQUIRE(x && w && y, PNPP_ERR_ARG, "linear_smallk: null pointer");
    PNPP_REQUIRE(M > 0 && N > 0 && K > 0 && K <= SMALLK_MAX, PNPP_ERR_ARG, "linear_smallk: M=%d N=%d K=%d (K <= %d)", M, N, K,
                 SMALLK_MAX);
    const int grid = cdiv(M, 4) < 4096 ? cdiv(M, 4) : 4096;
    ProfScope ps(as_stream(stream), "linear_smallk_kernel M=%d N=%d K=%d", M, N, K);
    hipLaunchKernelGGL(linear_smallk_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, w, b, M, K, N, y);
    PNPP_CHECK_LAUNCH("linear_smallk");
    return PNPP_OK;
}

static int attention_dropout_mask_impl(uint64_t seed, uint64_t stream_id, uint64_t *stream_id_dev, int B, int N, int H, float p,
                                       uint32_t *mask, uint32_t *maskT, void *stream) {
    PNPP_REQUIRE(mask && maskT, PNPP_ERR_ARG, "attention_dropout_mask: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && H > 0 && N % 128 == 0, PNPP_ERR_ARG, "attention_dropout_mask: bad size (N %% 128 == 0)");
    PNPP_REQUIRE(p >= 0.f && p < 1.f, PNPP_ERR_ARG, "attention_dropout_mask: p=%g outside [0, 1)", (double)p);
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;   // keep iff word >= thresh
    const size_t tiles = (size_t)B * H * (N / 32) * (N / 32);               // even: N / 32 is a multiple of 4
    const size_t blocks = tiles / 8;
    PNPP_REQUIRE(blocks <= 0x7fffffff, PNPP_ERR_ARG, "attention_dropout_mask: too many tiles");
    ProfScope ps(as_stream(stream), "attention_dropout_mask_kernel B=%d N=%d H=%d", B, N, H);
    hipLaunchKernelGGL(attention_dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (unsigned)seed,
                       (unsigned)(seed >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32),
                       reinterpret_cast<unsigned long long *>(stream_id_dev), N, thresh, mask, maskT);
    PNPP_CHECK_LAUNCH("attention_dropout_mask");
    return PNPP_OK;
}

extern "C" int pnpp_attention_dropout_mask(uint64_t seed, uint64_t stream_id, int B, int N, int H, float p, uint32_t *mask,
                                           uint32_t *maskT, void *stream) {
    return attention_dropout_mask_impl(seed, stream_id, nullptr, B, N, H, p, mask, maskT, stream);
}

extern "C" int pnpp_attention_dropout_mask_dev(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N, int H, float p,
                                               uint32_t *mask, uint32_t *maskT, void *stream) {
    PNPP_REQUIRE(stream_id_dev, PNPP_ERR_ARG, "attention_dropout_mask_dev: null counter pointer");
    return attention_dropout_mask_impl(seed, offset, stream_id_dev, B, N, H, p, mask, maskT, stream);
}

extern "C" int pnpp_attention_fwd(const float *qkv, int B, int N, int n_valid, int H, int head_dim, const uint32_t *mask, float p,
                                  float *out, float *lse, void *stream) {
    PNPP_REQUIRE(qkv && out, PNPP_ERR_ARG, "attention_fwd: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && H > 0, PNPP_ERR_ARG, "attention_fwd: non-positive size");
    PNPP_REQUIRE(head_dim == ATT_DH, PNPP_ERR_ARG, "attention_fwd: head dimension %d is not supported (only %d)", head_dim, ATT_DH);
    PNPP_REQUIRE(N % 128 == 0, PNPP_ERR_ARG, "attention_fwd: the row count N=%d must be a multiple of 128 (pad, and pass n_valid)", N);
    PNPP_REQUIRE(n_valid > 0 && n_valid <= N, PNPP_ERR_ARG, "attention_fwd: n_valid=%d outside 1..N=%d", n_valid, N);
    PNPP_REQUIRE(B <= 65535 && H <= 65535, PNPP_ERR_ARG, "attention_fwd: B or H exceeds the grid limit");
    ProfScope ps(as_stream(stream), "attention_fwd_kernel B=%d N=%d H=%d", B, N, H);
    PNPP_REQUIRE(p >= 0.f && p < 1.f, PNPP_ERR_ARG, "attention_fwd: dropout p=%g outside [0, 1)", (double)p);
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(N / 128, H, B), dim3(256), 0, as_stream(stream), qkv, N, n_valid, H,
                       1.0f / sqrtf((float)head_dim), mask, 1.0f / (1.0f - p), out, lse);
    PNPP_CHECK_LAUNCH("attention_fwd");
    return PNPP_OK;
}

extern "C" int pnpp_add_layernorm(const float *x, const float *r, const float *w, const float *b, int M, int E, float eps,
                                  float *y, void *stream) {
    PNPP_REQUIRE(x && w && b && y, PNPP_ERR_ARG, "add_layernorm: null pointer");
    PNPP_REQUIRE(M > 0 && E > 0 && E <= 128, PNPP_ERR_ARG, "add_layernorm: M=%d E=%d (E <= 128)", M, E);
    ProfScope ps(as_stream(stream), "add_layernorm_kernel M=%d E=%d", M, E);
    hipLaunchKernelGGL(add_layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, as_stream(stream), x, r, w, b, M, E, eps, y);
    PNPP_CHECK_LAUNCH("add_layernorm");
    return PNPP_OK;
}

extern "C" int pnpp_mean_points(const float *x, int B, int N, int E, float *y, void *stream) {
    PNPP_REQUIRE(x && y, PNPP_ERR_ARG, "mean_points: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && E > 0, PNPP_ERR_ARG, "mean_points: non-positive size");
    ProfScope ps(as_stream(stream), "mean_points_kernel B=%d N=%d E=%d", B, N, E);
    hipLaunchKernelGGL(mean_points_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, N, E, y);
    PNPP_CHECK_LAUNCH("mean_points");
    return PNPP_OK;
}

extern "C" size_t pnpp_linear_smallk_bwd_scratch_bytes(int M, int N) { return (size_t)cdiv(M, 256) * N * 16 * sizeof(float); }

extern "C" int pnpp_linear_smallk_bwd(const float *x, const float *dy, int M, int K, int N, float *dw, float *db, void *scratch,
                                      void *stream) {
    PNPP_REQUIRE(x && dy && dw && scratch, PNPP_ERR_ARG, "linear_smallk_bwd: null pointer");
    PNPP_REQUIRE(M > 0 && N > 0 && K > 0 && K <= SMALLK_MAX, PNPP_ERR_ARG, "linear_smallk_bwd: M=%d N=%d K=%d (K <= %d)", M, N, K,
                 SMALLK_MAX);
    hipStream_t st = as_stream(stream);
    float *slab = static_cast<float *>(scratch);
    const int nsplit = cdiv(M, 256);
    {
        ProfScope ps(st, "linear_smallk_bwd_kernel M=%d N=%d K=%d", M, N, K);
        hipLaunchKernelGGL(linear_smallk_bwd_kernel, dim3(nsplit), dim3(256), 0, st, x, dy, M, K, N, slab);
        PNPP_CHECK_LAUNCH("linear_smallk_bwd");
    }
    int rc = launch_slab_reduce(slab, nsplit, N, 16, K, -1, dw, K, st);
    if (rc != PNPP_OK) return rc;
    if (db) rc = launch_slab_reduce(slab + K, nsplit, N, 16, 1, -1, db, 1, st);
    return rc;
}

extern "C" int pnpp_attention_bwd(const float *qkv, const float *out, const float *d_out, const float *lse, int B, int N, int n_valid, int H,
                                  int head_dim, const uint32_t *mask, const uint32_t *maskT, float p, float *dqkv, float *dsum,
                                  void *stream) {
    PNPP_REQUIRE((mask == nullptr) == (maskT == nullptr), PNPP_ERR_ARG, "attention_bwd: pass both mask orientations or neither");
    PNPP_REQUIRE(p >= 0.f && p < 1.f, PNPP_ERR_ARG, "attention_bwd: dropout p=%g outside [0, 1)", (double)p);
    const float keep_scale = 1.0f / (1.0f - p);
    PNPP_REQUIRE(qkv && out && d_out && lse && dqkv && dsum, PNPP_ERR_ARG, "attention_bwd: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && H > 0, PNPP_ERR_ARG, "attention_bwd: non-positive size");
    PNPP_REQUIRE(head_dim == ATT_DH, PNPP_ERR_ARG, "attention_bwd: head dimension %d is not supported (only %d)", head_dim, ATT_DH);
    PNPP_REQUIRE(N % 128 == 0, PNPP_ERR_ARG, "attention_bwd: the row count N=%d must be a multiple of 128 (pad, and pass n_valid)", N);
    PNPP_REQUIRE(n_valid > 0 && n_valid <= N, PNPP_ERR_ARG, "attention_bwd: n_valid=%d outside 1..N=%d", n_valid, N);
    PNPP_REQUIRE(B <= 65535 && H <= 65535, PNPP_ERR_ARG, "attention_bwd: B or H exceeds the grid limit");
    hipStream_t st = as_stream(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    {
        ProfScope ps(st, "attention_bwd_dq_kernel B=%d N=%d H=%d", B, N, H);
        hipLaunchKernelGGL(attention_bwd_dq_kernel, dim3(N / 128, H, B), dim3(256), 0, st, qkv, out, d_out, lse, N, n_valid, H, scale, mask,
                           keep_scale, dqkv, dsum);
        PNPP_CHECK_LAUNCH("attention_bwd_dq");
    }
    {
        ProfScope ps(st, "attention_bwd_dkv_kernel B=%d N=%d H=%d", B, N, H);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel, dim3(N / 128, H, B), dim3(256), 0, st, qkv, d_out, lse, dsum, N, n_valid, H, scale, maskT,
                           keep_scale, dqkv);
        PNPP_CHECK_LAUNCH("attention_bwd_dkv");
    }
    return PNPP_OK;
}

extern "C" size_t pnpp_add_layernorm_bwd_scratch_bytes(int M, int E) { return (size_t)cdiv(M, 64) * 2 * E * sizeof(float); }

extern "C" int pnpp_add_layernorm_bwd(const float *x, const float *r, const float *w, const float *dy, int M, int E, float eps,
                                      float *du, float *dwb, void *scratch, void *stream) {
    PNPP_REQUIRE(x && w && dy && du && dwb && scratch, PNPP_ERR_ARG, "add_layernorm_bwd: null pointer");
    PNPP_REQUIRE(M > 0 && E > 0 && E <= 128, PNPP_ERR_ARG, "add_layernorm_bwd: M=%d E=%d (E <= 128)", M, E);
    hipStream_t st = as_stream(stream);
    float *slab = static_cast<float *>(scratch);
    const int nsplit = cdiv(M, 64);
    {
        ProfScope ps(st, "add_layernorm_bwd_kernel M=%d E=%d", M, E);
        hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(nsplit), dim3(256), 0, st, x, r, w, dy, M, E, eps, du, slab);
        PNPP_CHECK_LAUNCH("add_layernorm_bwd");
    }
    return launch_slab_reduce(slab, nsplit, 2, E, E, -1, dwb, E, st);  // dwb (2,E): row 0 = d weight, row 1 = d bias
}

extern "C" int pnpp_mean_points_bwd(const float *dy, int B, int N, int E, float *dx, void *stream) {
    PNPP_REQUIRE(dy && dx, PNPP_ERR_ARG, "mean_points_bwd: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && E > 0, PNPP_ERR_ARG, "mean_points_bwd: non-positive size");
    const size_t total = (size_t)B * N * E;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ProfScope ps(as_stream(stream), "mean_points_bwd_kernel B=%d N=%d E=%d", B, N, E);
    hipLaunchKernelGGL(mean_points_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), dy, N, E, total, dx);
    PNPP_CHECK_LAUNCH("mean_points_bwd");
    return PNPP_OK;
}
