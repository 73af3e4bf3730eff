// index_kernels.hip -- sampling / neighbour search / gather kernels for gfx950 (MI355X).
//
// Everything here is integer-, compare- or copy-bound (HBM/LDS bound, no MFMA): the design rules
// are coalesced xyz reads, candidate tiles staged once per workgroup in LDS (SoA, conflict-free),
// one 64-lane wavefront per query, and wave shuffles for the selection reductions.
//
// Bit-exactness: the float32 arithmetic below reproduces the evaluation order of the reference's
// CPU PyTorch path (oracle/index_ops.c is the CPU restatement).  This file is
// compiled with -ffp-contract=off and uses explicit fmaf where -- and only where -- ATen fuses.
#include "common.h"
#include "sampler_device.h"

#pragma clang fp contract(off)

namespace pnpp {

// ---------------------------------------------------------------------------------------------
// exact float32 recipes
// ---------------------------------------------------------------------------------------------
// models/base.py:25-26  torch.sum(p**2, -1): (x^2 + y^2) + z^2, unfused
__device__ __forceinline__ float sq3_exact(float x, float y, float z) {
    float s = __fmul_rn(x, x);
    s = __fadd_rn(s, __fmul_rn(y, y));
    s = __fadd_rn(s, __fmul_rn(z, z));
    return s;
}
// models/base.py:24-26  -2*matmul (K=3 sgemm: fma chain) then += |a|^2 then += |b|^2
__device__ __forceinline__ float pair_dist_exact(float ax, float ay, float az, float bx, float by, float bz, float sa,
                                                 float sb) {
    float dot = __fmaf_rn(az, bz, __fmaf_rn(ay, by, __fmul_rn(ax, bx)));
    float d = __fmul_rn(-2.0f, dot);
    d = __fadd_rn(d, sa);
    d = __fadd_rn(d, sb);
    return d;
}
// PointNet++Demo.py:25,63  torch.sum((a-b)**2, -1): direct form, unfused
__device__ __forceinline__ float direct_dist_exact(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    return d;
}

// ---------------------------------------------------------------------------------------------
// square_distance: one thread per output element, n fastest (coalesced 4-byte stores)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) square_distance_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, int S, int N,
                                                              float *__restrict__ out) {
    const int b = blockIdx.z, s = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *a = src + ((size_t)b * S + s) * 3;
    const float *q = dst + ((size_t)b * N + n) * 3;
    const float ax = a[0], ay = a[1], az = a[2];
    const float bx = q[0], by = q[1], bz = q[2];
    out[((size_t)b * S + s) * N + n] = pair_dist_exact(ax, ay, az, bx, by, bz, sq3_exact(ax, ay, az), sq3_exact(bx, by, bz));
}

// ---------------------------------------------------------------------------------------------
// kNN grouping (models/base.py:29-35): one wavefront per query centre.
//   * the cloud is staged tile by tile (TILE points) into LDS as x[],y[],z[],|p|^2[] (SoA);
//   * every lane forms 64-bit keys (sortable(d) << 32 | n) for TILE/64 candidates in registers;
//   * prune: the k-th smallest of the 64 lane minima bounds the k-th smallest key of the tile from
//     above (k <= 64), and so does the previous best list's last entry; only keys <= that bound are
//     compacted (ballot + prefix popcount) into a small LDS pool -- typically 40-70 of 1024;
//   * select: every pooled key counts the pooled keys below it (broadcast LDS reads); rank < k
//     writes straight to its sorted slot.  Keys are unique (index in the low word), so ranks are too.
//   * pathological inputs (pool overflow: massive ties, or k > 64) fall back to k rounds of
//     "wave-min, owner retires its key".
// Output: neighbours in ascending (distance, index).
// ---------------------------------------------------------------------------------------------
constexpr int KNN_TILE = 1024;
constexpr int KNN_CPL = KNN_TILE / 64;  // candidates per lane per tile
constexpr int KNN_KMAX = 128;
constexpr int KNN_POOL = 256;
constexpr unsigned long long KEY_MAX = ~0ull;

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned o = (unsigned)__shfl_xor((int)v, m, 64);
        v = o < v ? o : v;
    }
    return v;
}

// centre != nullptr: the query of (b, q) is xyz[b, centre[b, q]] and the kernel also writes it to out_a / out_b (the
// caller's new_xyz and the copy kept for backward) -- the set-abstraction forward then needs no separate gather launch.
// gather != nullptr: the "cloud" of (b) is itself a gathered subset of xyz -- candidate p is xyz[b, gather[b, p]] (the centres
// of the level below, xyz holding Nsrc points per cloud): the neighbour search of a stacked level then needs neither the level
// below's output nor a launch of its own (knn_pair_kernel).
struct KnnJob {
    const float *new_xyz;      // explicit queries, or nullptr with `centre`
    const float *xyz;
    const int32_t *gather;     // (B, N) rows of xyz that form the candidate cloud, or nullptr (the cloud is xyz itself)
    int Nsrc;                  // points per cloud in xyz (== N without gather)
    int S, N, k;
    int32_t *idx;
    const int32_t *centre;
    float *out_a, *out_b;
};

struct KnnLds {   // one copy per workgroup, shared by the two instantiations of the body a pair kernel contains
    float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE], sn[KNN_TILE];
    unsigned long long best[4][2][KNN_KMAX];
    unsigned long long pool[4][KNN_POOL];
};

template <bool GATHER>
__device__ __forceinline__ void knn_body(const KnnJob &J, const int bx, const int b, KnnLds &L) {
    const float *__restrict__ new_xyz = J.new_xyz;
    const float *__restrict__ xyz = J.xyz;
    const int32_t *__restrict__ centre = J.centre;
    const int32_t *__restrict__ gather = GATHER ? J.gather + (size_t)b * J.N : nullptr;
    int32_t *__restrict__ idx = J.idx;
    float *__restrict__ out_a = J.out_a, *__restrict__ out_b = J.out_b;
    const int S = J.S, N = J.N, k = J.k;
    float(&sx)[KNN_TILE] = L.sx, (&sy)[KNN_TILE] = L.sy, (&sz)[KNN_TILE] = L.sz, (&sn)[KNN_TILE] = L.sn;
    unsigned long long(&best)[4][2][KNN_KMAX] = L.best;
    unsigned long long(&pool)[4][KNN_POOL] = L.pool;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = bx * 4 + wave;
    const bool active = q < S;  // wave-uniform
    float ax = 0.f, ay = 0.f, az = 0.f, sa = 0.f;
    if (active) {
        const float *a = new_xyz + ((size_t)b * S + q) * 3;
        if (centre) {
            const int c = centre[(size_t)b * S + q];
            a = xyz + ((size_t)b * J.Nsrc + (GATHER ? gather[c] : c)) * 3;
        }
        ax = a[0], ay = a[1], az = a[2];
        sa = sq3_exact(ax, ay, az);
        if (centre && lane < 3) {
            const float v = lane == 0 ? ax : (lane == 1 ? ay : az);
            out_a[((size_t)b * S + q) * 3 + lane] = v;
            if (out_b) out_b[((size_t)b * S + q) * 3 + lane] = v;
        }
    }
    int cur = 0;         // which half of best[] holds the current list
    bool have = false;   // best[cur][0..k) is valid (wave-uniform)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    const float *cloud = xyz + (size_t)b * J.Nsrc * 3;
    for (int t0 = 0; t0 < N; t0 += KNN_TILE) {
        __syncthreads();  // previous tile fully consumed
        const int cnt = min(KNN_TILE, N - t0);
        // coalesced stage: 3*cnt consecutive floats, de-interleaved into SoA (gathered clouds: three floats per listed row)
        for (int i = threadIdx.x; i < cnt * 3; i += 256) {
            int p = i / 3, c = i - p * 3;
            float v;
            if constexpr (GATHER) v = cloud[(size_t)gather[t0 + p] * 3 + c];
            else v = cloud[(size_t)t0 * 3 + i];
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < cnt; p += 256) sn[p] = sq3_exact(sx[p], sy[p], sz[p]);
        __syncthreads();
        if (!active) continue;

        unsigned long long key[KNN_CPL + 2];
        unsigned long long lmin = KEY_MAX;
#pragma unroll
        for (int j = 0; j < KNN_CPL; ++j) {
            const int p = j * 64 + lane;
            key[j] = KEY_MAX;
            if (p < cnt) {
                float d = pair_dist_exact(ax, ay, az, sx[p], sy[p], sz[p], sa, sn[p]);
                key[j] = ((unsigned long long)f32_sortable(d) << 32) | (unsigned)(t0 + p);
            }
            lmin = key[j] < lmin ? key[j] : lmin;
        }
        // previous best list rides along as up to two more candidates per lane
        key[KNN_CPL] = (have && lane < k) ? best[wave][cur][lane] : KEY_MAX;
        key[KNN_CPL + 1] = (have && lane + 64 < k) ? best[wave][cur][lane + 64] : KEY_MAX;

        // ---- pruning bound on the distance word ----
        unsigned long long T = KEY_MAX;
        if (k <= 64) {
            const unsigned hi = (unsigned)(lmin >> 32);
            int cnt_le = 0;
#pragma unroll
            for (int i = 0; i < 64; ++i) cnt_le += ((unsigned)__builtin_amdgcn_readlane((int)hi, i) <= hi) ? 1 : 0;
            const unsigned thi = wave_min_u32(cnt_le >= k ? hi : 0xffffffffu);
            T = ((unsigned long long)thi << 32) | 0xffffffffull;
        }
        if (have) {
            const unsigned long long last = best[wave][cur][k - 1];
            T = last < T ? last : T;
        }
        // ---- compaction of the survivors into the LDS pool ----
        int P = 0;  // wave-uniform
#pragma unroll
        for (int j = 0; j < KNN_CPL + 2; ++j) {
            const bool in = key[j] <= T && key[j] != KEY_MAX;
            const unsigned long long m = __ballot(in);
            const int pos = P + __popcll(m & lt_mask);
            if (in && pos < KNN_POOL) pool[wave][pos] = key[j];
            P += __popcll(m);
        }
        if (P <= KNN_POOL) {
            // ---- rank selection inside the pool ----
            for (int e0 = 0; e0 < P; e0 += 64) {
                const int e = e0 + lane;
                const unsigned long long mine = e < P ? pool[wave][e] : KEY_MAX;
                int r = 0;
                for (int j = 0; j < P; ++j) r += pool[wave][j] < mine ? 1 : 0;
                if (e < P && r < k) best[wave][cur ^ 1][r] = mine;
            }
        } else {
            // ---- fallback: k rounds of wave-min extraction over everything this lane holds ----
            unsigned long long fmin = KEY_MAX;
#pragma unroll
            for (int j = 0; j < KNN_CPL + 2; ++j) fmin = key[j] < fmin ? key[j] : fmin;
            for (int it = 0; it < k; ++it) {
                const unsigned long long w = wave_min_u64(fmin);
                if (lane == 0) best[wave][cur ^ 1][it] = w;
                if (fmin == w && w != KEY_MAX) {  // keys are unique: exactly one owner
                    unsigned long long m2 = KEY_MAX;
#pragma unroll
                    for (int j = 0; j < KNN_CPL + 2; ++j) {
                        if (key[j] == w) key[j] = KEY_MAX;
                        m2 = key[j] < m2 ? key[j] : m2;
                    }
                    fmin = m2;
                }
            }
        }
        cur ^= 1;
        have = true;
    }
    if (active) {
        int32_t *o = idx + ((size_t)b * S + q) * k;
        for (int j = lane; j < k; j += 64) o[j] = (int32_t)(unsigned)(best[wave][cur][j] & 0xffffffffu);
    }
}

__global__ void __launch_bounds__(256) knn_kernel(const KnnJob J) {
    __shared__ KnnLds L;
    knn_body<false>(J, blockIdx.x, blockIdx.y, L);
}

// The neighbour searches of two stacked levels in ONE launch (they are independent once both levels' centre indices are drawn:
// level 2 searches among level 1's centres, which are rows of the same cloud).  blockIdx.x < nb1: level 1, else level 2.
// 5 workgroups per CU (<= 96 registers): 32 clouds x (32 + 8) workgroups = 1,280 are then all resident at once on 256 CUs -- at 4 per
// CU the last 256 start only when a slot frees up and the launch takes a round and a half (measured: 21.1 us against 12.6 + 7.0 for
// the two separate launches).  The short level-2 workgroups come first.
__global__ void __launch_bounds__(256, 5) knn_pair_kernel(const KnnJob J1, const KnnJob J2, int nb2) {
    __shared__ KnnLds L;
    if ((int)blockIdx.x < nb2) knn_body<true>(J2, blockIdx.x, blockIdx.y, L);
    else knn_body<false>(J1, blockIdx.x - nb2, blockIdx.y, L);
}

// Wave-wide maximum of a 64-bit key, result in every lane.  Six dependent ds_bpermute round trips (what __shfl_xor compiles
// to) are most of a farthest-point round; this is the DPP form: an inclusive row scan (row_shr 1, 2, 4, 8 inside the rows of
// 16 lanes, shifted-in lanes keep their own value), row_bcast15 / row_bcast31 to fold the four rows, lane 63 read back
// through the scalar unit -- VALU latency only.
__device__ __forceinline__ unsigned long long wave_max_u64_dpp(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define PNPP_DPP_MAX_STEP(CTRL, ROWMASK)                                                                              \
    {                                                                                                                 \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROWMASK, 0xf, false);      \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROWMASK, 0xf, false);      \
        const bool take = ohi > hi || (ohi == hi && olo > lo);                                                        \
        lo = take ? olo : lo, hi = take ? ohi : hi;                                                                   \
    }
    PNPP_DPP_MAX_STEP(0x111, 0xf)  // row_shr:1
    PNPP_DPP_MAX_STEP(0x112, 0xf)  // row_shr:2
    PNPP_DPP_MAX_STEP(0x114, 0xf)  // row_shr:4
    PNPP_DPP_MAX_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds its row's maximum
    PNPP_DPP_MAX_STEP(0x142, 0xa)  // row_bcast15 -> rows 1 and 3 fold in the row below
    PNPP_DPP_MAX_STEP(0x143, 0xc)  // row_bcast31 -> rows 2 and 3 fold in lane 31: lane 63 holds the wave's maximum
#undef PNPP_DPP_MAX_STEP
    const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63), rhi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)rhi << 32) | rlo;
}

// ---------------------------------------------------------------------------------------------
// farthest point sampling (PointNet++Demo.py:8-29): one workgroup per cloud.
//
// The algorithm is a chain of npoint dependent rounds (distance update over all N points, arg-max, next centre), so its
// time is npoint x (latency of one round); a round is short enough (< 1 us up to N ~ 16k) that handing the arg-max
// between workgroups through memory (>= 1 us per hop) would lengthen it -- a cloud therefore stays on ONE CU and the
// round is made short instead:
//   * every thread keeps its PPT points (x, y, z, running minimum) in REGISTERS for the whole kernel: a round touches
//     the LDS only for the T/64 per-wave winners, and N is bounded by the register file (T x PPT points), not by 16 bytes
//     of LDS per point; points beyond T x PPT (N > 16384) live in the LDS as before (the "tail", up to 160 KiB / 16 B more);
//   * the winner's coordinates travel with its key: the lane that owns a wave's maximum writes (key, x, y, z) to its
//     wave's slot, so the next centre needs no lookup by index (one barrier per round, slots double-buffered);
//   * ties: key = (distance bits << 32) | ~index, i.e. the FIRST maximum, as torch.max(distance, -1)[1] (line 28).
// Distances are the reference's exact float32 sequence ((dx^2 + dy^2) + dz^2, no FMA) -> indices are bit-exact.
// ---------------------------------------------------------------------------------------------
template <int T, int PPT>
__global__ void __launch_bounds__(T) fps_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                const int32_t *__restrict__ start, int32_t *__restrict__ out, int ntail) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = T / 64;
    // one LDS object (a second __shared__ array beside a dynamic one can cost a vmcnt(0) per read, guide 5.4 item 4a):
    // [2][NW] keys (u64) | [2][NW][4] winner coordinates | tail: x[ntail] y[ntail] z[ntail] d[ntail]
    unsigned long long *wkey = reinterpret_cast<unsigned long long *>(lds);
    float *wxyz = lds + 2 * 2 * NW;
    float *sx = wxyz + 2 * NW * 4, *sy = sx + ntail, *sz = sy + ntail, *sd = sz + ntail;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *cloud = xyz + (size_t)b * N * 3;
    float px[PPT], py[PPT], pz[PPT], pd[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = tid + j * T, pc = p < N ? p : N - 1;
        px[j] = cloud[pc * 3 + 0], py[j] = cloud[pc * 3 + 1], pz[j] = cloud[pc * 3 + 2];
        pd[j] = p < N ? 1e10f : -1.f;   // a slot past the cloud: min(d, -1) = -1 never beats a real (non-negative) distance
    }
    for (int q = tid; q < ntail; q += T) {
        const int p = T * PPT + q;
        sx[q] = cloud[p * 3 + 0], sy[q] = cloud[p * 3 + 1], sz[q] = cloud[p * 3 + 2], sd[q] = 1e10f;
    }
    int far = start[b];
    float cx = cloud[far * 3 + 0], cy = cloud[far * 3 + 1], cz = cloud[far * 3 + 2];
    if (ntail) __syncthreads();

    for (int it = 0; it < npoint; ++it) {
        if (tid == 0) out[(size_t)b * npoint + it] = far;
        // inside a thread the points are visited in ascending index, so a strict float compare keeps the FIRST maximum;
        // distances are non-negative, so their bit patterns order like unsigned integers -- the 64-bit key
        // (distance bits << 32) | ~index is only built once per thread, for the cross-lane reduction
        float bv = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
        int bp = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float cur = fminf(direct_dist_exact(px[j], py[j], pz[j], cx, cy, cz), pd[j]);   // NaN distance: keeps pd, like dist < distance
            pd[j] = cur;
            const bool w = cur > bv;
            bv = w ? cur : bv, bp = w ? tid + j * T : bp, bx = w ? px[j] : bx, by = w ? py[j] : by, bz = w ? pz[j] : bz;
        }
        for (int q = tid; q < ntail; q += T) {
            const float x = sx[q], y = sy[q], z = sz[q];
            const float cur = fminf(direct_dist_exact(x, y, z, cx, cy, cz), sd[q]);
            sd[q] = cur;
            const bool w = cur > bv;
            bv = w ? cur : bv, bp = w ? T * PPT + q : bp, bx = w ? x : bx, by = w ? y : by, bz = w ? z : bz;
        }
        // a thread without a valid point (bv < 0) contributes key 0, below every real key (~index is never 0)
        const unsigned long long bestk =
            bv < 0.f ? 0ull : ((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)(0xffffffffu - (unsigned)bp);
        const unsigned long long wk = wave_max_u64_dpp(bestk);
        if (bestk == wk && (wk != 0ull || lane == 0)) {   // keys carry the point index: exactly one lane of the wave holds the maximum
            wkey[(it & 1) * NW + wave] = wk;
            float *w4 = wxyz + ((it & 1) * NW + wave) * 4;
            w4[0] = bx, w4[1] = by, w4[2] = bz;
        }
        __syncthreads();
        // (folding the NW slots with a second DPP maximum + readlane instead of reading them all was measured slower:
        // 98.6 vs 80.3 us at N = 1024 -- the scalar round trip is longer than two LDS round trips)
        unsigned long long g = 0;
        int gw = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned long long o = wkey[(it & 1) * NW + w];
            const bool win = o > g;
            g = win ? o : g, gw = win ? w : gw;
        }
        far = (int)(0xffffffffu - (unsigned)(g & 0xffffffffu));
        const float *w4 = wxyz + ((it & 1) * NW + gw) * 4;
        cx = w4[0], cy = w4[1], cz = w4[2];
    }
}

// ---------------------------------------------------------------------------------------------
// radius ball query (PointNet++Demo.py:49-70): one wavefront per centre, ascending index scan,
// ballot + prefix popcount compaction, early exit once nsample hits were written.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ball_query_kernel(const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                         int S, int N, float r2, int nsample, int32_t *__restrict__ idx) {
    __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    const bool active = q < S;
    float ax = 0.f, ay = 0.f, az = 0.f;
    int32_t *o = nullptr;
    if (active) {
        const float *a = new_xyz + ((size_t)b * S + q) * 3;
        ax = a[0], ay = a[1], az = a[2];
        o = idx + ((size_t)b * S + q) * nsample;
    }
    int cnt = 0, first = N;  // wave-uniform
    const float *cloud = xyz + (size_t)b * N * 3;
    for (int t0 = 0; t0 < N; t0 += KNN_TILE) {
        __syncthreads();
        const int tc = min(KNN_TILE, N - t0);
        for (int i = threadIdx.x; i < tc * 3; i += 256) {
            float v = cloud[(size_t)t0 * 3 + i];
            int p = i / 3, c = i - p * 3;
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();
        if (!active || cnt >= nsample) continue;
        for (int c0 = 0; c0 < tc && cnt < nsample; c0 += 64) {
            const int p = c0 + lane;
            bool in = false;
            if (p < tc) {
                float d = direct_dist_exact(ax, ay, az, sx[p], sy[p], sz[p]);
                in = !(d > r2);
            }
            const unsigned long long m = __ballot(in);
            if (m == 0ull) continue;
            if (first == N) first = t0 + c0 + (__ffsll((long long)m) - 1);
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < nsample) o[pos] = t0 + p;
            cnt += __popcll(m);
        }
    }
    if (active) {
        cnt = min(cnt, nsample);
        for (int j = cnt + lane; j < nsample; j += 64) o[j] = first;
    }
}

// ---------------------------------------------------------------------------------------------
// device-side centre sampling: uniform random ordered subset (replaces B host randperm calls).
// key(n) = Philox4x32-10(counter = (n, b, stream_lo, stream_hi), key = seed); rank by (key, n);
// out[rank] = n for rank < npoint.  Fully deterministic (a pure function of seed, stream id and cloud).
// ---------------------------------------------------------------------------------------------
// (philox_key and the sampling body: sampler_device.h)
__global__ void __launch_bounds__(256) sample_random_kernel(unsigned seed_lo, unsigned seed_hi, unsigned str_lo,
                                                            unsigned str_hi, unsigned long long *__restrict__ str_dev,
                                                            int N, int npoint, int32_t *__restrict__ out, int B1, int N2,
                                                            int npoint2, int32_t *__restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long cand[];
    __shared__ int nc_s;
    sample_random_body(seed_lo, seed_hi, str_lo, str_hi, str_dev, N, npoint, out, B1, N2, npoint2, out2, blockIdx.x, gridDim.x, cand, nc_s);
}

// ---------------------------------------------------------------------------------------------
// index_points (models/base.py:4-18): row gather, 16-byte vectors when C % 4 == 0
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_rows_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                                          int N, int C, int M, size_t total_vec, int vec,
                                                          float *__restrict__ out) {
    const int cv = C / vec;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cv;
        const int c = (int)(i - row * cv) * vec;
        const int b = (int)(row / M);
        const int n = idx[row];
        const float *s = points + ((size_t)b * N + n) * C + c;
        float *d = out + row * C + c;
        if (vec == 4) {
            *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(s);
        } else {
            *d = *s;
        }
    }
}

// backward of the gather: one wavefront per destination row (b,n); the cloud's M indices are scanned
// in order, so duplicate contributions are added in a fixed order (bitwise reproducible, no atomics).
__global__ void __launch_bounds__(64) scatter_rows_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ idx,
                                                              int N, int C, int M, float *__restrict__ dpoints) {
    const int b = blockIdx.y, n = blockIdx.x, lane = threadIdx.x;
    const int32_t *ib = idx + (size_t)b * M;
    const float *db = dout + (size_t)b * M * C;
    for (int c0 = 0; c0 < C; c0 += 64 * 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int m0 = 0; m0 < M; m0 += 64) {
            const int m = m0 + lane;
            unsigned long long hit = __ballot(m < M && ib[m] == n);
            while (hit) {
                const int p = __ffsll((long long)hit) - 1;
                hit &= hit - 1;
                const float *r = db + (size_t)(m0 + p) * C + c0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = j * 64 + lane;
                    if (c0 + c < C) acc[j] += r[c];
                }
            }
        }
        float *d = dpoints + ((size_t)b * N + n) * C + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = j * 64 + lane;
            if (c0 + c < C) d[c] += acc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// on-device point subsampling for the data path (dataloader_*: sample_pts / np.random.choice(len, num, replace=len<num)):
// a bank of full clouds stays resident in HBM as (n_clouds, Lmax, 3) + lengths; one workgroup per batch slot draws
// `num` rows of its cloud -- an ordered uniform subset without replacement when the cloud has at least `num` points
// (same key + rank scheme as sample_random_kernel), uniform draws with replacement otherwise -- and writes the
// coordinates straight into the batch tensor.  Pure function of (seed, stream id, slot); empty clouds give zeros.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
subsample_points_kernel(unsigned seed_lo, unsigned seed_hi, unsigned str_lo, unsigned str_hi,
                        const float *__restrict__ bank, const int32_t *__restrict__ lengths,
                        const int32_t *__restrict__ cloud_ids, int Lmax, int num, float *__restrict__ out, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long cand[];
    // gridDim.y workgroups share one batch slot: each builds the slot's whole candidate list (cheap: one Philox call per
    // point) and ranks / gathers every gridDim.y-th candidate -- ranks do not depend on the order candidates were compacted in
    const int b = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int cloud = cloud_ids ? cloud_ids[b] : b;
    const int L = min(lengths[cloud], Lmax);
    const float *src = bank + (size_t)cloud * Lmax * 3;
    float *dst = out + (size_t)b * num * 3;
    if (L <= 0) {
        for (int j = part * 256 + threadIdx.x; j < 3 * num; j += 256 * nparts) dst[j] = 0.f;
        return;
    }
    if (L < num) {  // with replacement: index = floor(u * L), u from the 32-bit Philox word
        for (int j = part * 256 + threadIdx.x; j < num; j += 256 * nparts) {
            const unsigned u = philox_key((unsigned)j, (unsigned)b, str_lo, str_hi, seed_lo, seed_hi);
            const int n = (int)(((unsigned long long)u * (unsigned long long)L) >> 32);
            dst[3 * j] = src[3 * n], dst[3 * j + 1] = src[3 * n + 1], dst[3 * j + 2] = src[3 * n + 2];
        }
        return;
    }
    const double keep = (num + 4.0 * sqrt((double)num) + 16.0) / (double)L;
    unsigned cut = keep >= 1.0 ? 0xffffffffu : (unsigned)(keep * 4294967296.0);
    // Compaction without atomics: wave w scans the points [w L/4, (w+1) L/4) and appends its survivors to its own segment
    // of the list, so the list -- positions included -- is the same in every workgroup of the slot: position p of the
    // concatenated segments belongs to part p / 256 % nparts.  Segments are padded to an even length with a word above every key.
    // The table holds `cap` words per segment whatever the cloud's length (clouds of millions of points draw through the same
    // few thousand words): a cut that leaves fewer than num keys, or more than a segment holds, is bisected -- the result (the
    // num smallest (key, index) words) never depends on where the cut ends up.
    __shared__ int cnt_s[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = cap;                                    // capacity of one segment (even)
    const int lo = wave * ((L + 3) / 4), hi = min(L, lo + (L + 3) / 4);
    unsigned long long *mycand = cand + (size_t)wave * seg;
    unsigned cut_lo = 0u, cut_hi = 0xffffffffu;             // keys <= cut_lo are too few, keys <= cut_hi may not fit
    for (int round = 0;; ++round) {
        int cnt = 0;
        for (int n0 = lo; n0 < hi; n0 += 64) {
            const int n = n0 + lane;
            const unsigned k = n < hi ? philox_key((unsigned)n, (unsigned)b, str_lo, str_hi, seed_lo, seed_hi) : 0u;
            const bool keepit = n < hi && k <= cut;
            const unsigned long long vote = __ballot(keepit);
            const int pos = cnt + __popcll(vote & ((1ull << lane) - 1ull));
            if (keepit && pos < seg - 1) mycand[pos] = ((unsigned long long)k << 32) | (unsigned)n;
            cnt += __popcll(vote);
        }
        if (lane == 0) {
            if ((cnt & 1) && cnt < seg) mycand[cnt] = ~0ull;
            cnt_s[wave] = cnt;
        }
        __syncthreads();
        const int total = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
        const bool fits = cnt_s[0] < seg && cnt_s[1] < seg && cnt_s[2] < seg && cnt_s[3] < seg;
        if ((total >= num && fits) || round >= 40) break;   // uniform (round bound: never reached, 32 bisections settle any cut)
        if (total < num) {                                  // too few keys under the cut (a > 5 sigma event): raise it
            cut_lo = cut;
            if (cut_hi != 0xffffffffu) cut = cut_lo + (cut_hi - cut_lo) / 2u;          // between a too-small and a too-large cut
            else if (4 * (seg - 2) >= L + 8) cut = 0xffffffffu;                        // the whole cloud fits the table: take it
            else cut = cut > 0x7fffffffu ? 0xffffffffu : 2u * cut + 1u;                // never overflowed yet: double
        } else {                                            // a segment overflowed: lower it
            cut_hi = cut;
            cut = cut_lo + (cut_hi - cut_lo) / 2u;
        }
        __syncthreads();
    }
    const int c0 = cnt_s[0], c1 = cnt_s[1], c2n = cnt_s[2], c3 = cnt_s[3], nc = c0 + c1 + c2n + c3;
    const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(cand);
    for (int p = part * 256 + threadIdx.x; p < nc; p += 256 * nparts) {
        int w = 0, q = p;                                  // position p -> (segment, index)
        if (q >= c0) { q -= c0, w = 1; if (q >= c1) { q -= c1, w = 2; if (q >= c2n) q -= c2n, w = 3; } }
        const unsigned long long mine = cand[(size_t)w * seg + q];
        int rank = 0;
#pragma unroll
        for (int sgm = 0; sgm < 4; ++sgm) {
            const int np = (cnt_s[sgm] + 1) >> 1;
            const ulonglong2 *c2 = pairs + (size_t)sgm * (seg >> 1);
#pragma unroll 4
            for (int jj = 0; jj < np; ++jj) {             // one 16-byte LDS broadcast per two candidates
                const ulonglong2 o = c2[jj];
                rank += (o.x < mine) + (o.y < mine);
            }
        }
        if (rank < num) {
            const int n = (int)(unsigned)mine;
            dst[3 * rank] = src[3 * n], dst[3 * rank + 1] = src[3 * n + 1], dst[3 * rank + 2] = src[3 * n + 2];
        }
    }
}

// gather of centre coordinates new_xyz[b,s,:] = xyz[b, centre[b,s], :] into two destinations (the caller's output and
// the copy kept for backward); centre == nullptr writes the origin (group_all, pointnet_pp_8dir.py:24)
__global__ void __launch_bounds__(256) gather_centres_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ centre,
                                                             int N, int S, int total, float *__restrict__ out_a,
                                                             float *__restrict__ out_b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float x = 0.f, y = 0.f, z = 0.f;
    if (centre) {
        const int b = i / S;
        const float *s = xyz + ((size_t)b * N + centre[i]) * 3;
        x = s[0], y = s[1], z = s[2];
    }
    out_a[(size_t)i * 3 + 0] = x;
    out_a[(size_t)i * 3 + 1] = y;
    out_a[(size_t)i * 3 + 2] = z;
    if (out_b) {
        out_b[(size_t)i * 3 + 0] = x;
        out_b[(size_t)i * 3 + 1] = y;
        out_b[(size_t)i * 3 + 2] = z;
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers (internal C++ API, used by the C ABI and by the set-abstraction orchestrator)
// ---------------------------------------------------------------------------------------------
int launch_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx, hipStream_t st) {
    PNPP_REQUIRE(new_xyz && xyz && idx, PNPP_ERR_ARG, "knn: null pointer");
    PNPP_REQUIRE(B > 0 && S > 0 && N > 0 && k > 0, PNPP_ERR_ARG, "knn: non-positive size (B=%d S=%d N=%d k=%d)", B, S, N, k);
    PNPP_REQUIRE(k <= N, PNPP_ERR_RANGE, "selected index k out of range (k=%d > N=%d)", k, N);
    PNPP_REQUIRE(k <= KNN_KMAX, PNPP_ERR_ARG, "knn: nsample=%d exceeds the supported maximum %d", k, KNN_KMAX);
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "knn: batch %d exceeds grid limit", B);
    ProfScope ps(st, "knn_kernel B=%d S=%d N=%d k=%d", B, S, N, k);
    const KnnJob J{new_xyz, xyz, nullptr, N, S, N, k, idx, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(knn_kernel, dim3(cdiv(S, 4), B), dim3(256), 0, st, J);
    PNPP_CHECK_LAUNCH("knn");
    return PNPP_OK;
}

// kNN whose queries are gathered centres: also writes the centre coordinates (out_a, optional out_b)
int launch_knn_centres(const float *xyz, const int32_t *centre, int B, int S, int N, int k, int32_t *idx, float *out_a,
                       float *out_b, hipStream_t st) {
    PNPP_REQUIRE(xyz && centre && idx && out_a, PNPP_ERR_ARG, "knn: null pointer");
    PNPP_REQUIRE(B > 0 && S > 0 && N > 0 && k > 0, PNPP_ERR_ARG, "knn: non-positive size (B=%d S=%d N=%d k=%d)", B, S, N, k);
    PNPP_REQUIRE(k <= N, PNPP_ERR_RANGE, "selected index k out of range (k=%d > N=%d)", k, N);
    PNPP_REQUIRE(k <= KNN_KMAX, PNPP_ERR_ARG, "knn: nsample=%d exceeds the supported maximum %d", k, KNN_KMAX);
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "knn: batch %d exceeds grid limit", B);
    ProfScope ps(st, "knn_kernel B=%d S=%d N=%d k=%d", B, S, N, k);
    const KnnJob J{nullptr, xyz, nullptr, N, S, N, k, idx, centre, out_a, out_b};
    hipLaunchKernelGGL(knn_kernel, dim3(cdiv(S, 4), B), dim3(256), 0, st, J);
    PNPP_CHECK_LAUNCH("knn");
    return PNPP_OK;
}

// Both levels' searches in one launch: level 1 = S1 centres (rows centre1 of the N-point cloud), k1 neighbours among the N
// points; level 2 = S2 of those centres (positions centre2 in 0..S1-1), k2 neighbours among the S1 centres.
int launch_knn_pair(const float *xyz, int B, int N, const int32_t *centre1, int S1, int k1, int32_t *idx1, float *a1, float *b1,
                    const int32_t *centre2, int S2, int k2, int32_t *idx2, float *a2, float *b2, hipStream_t st) {
    PNPP_REQUIRE(xyz && centre1 && centre2 && idx1 && idx2 && a1 && a2, PNPP_ERR_ARG, "knn_pair: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && S1 > 0 && S2 > 0 && k1 > 0 && k2 > 0, PNPP_ERR_ARG, "knn_pair: non-positive size");
    PNPP_REQUIRE(k1 <= N && k2 <= S1, PNPP_ERR_RANGE, "selected index k out of range (k=%d > N=%d)", k1 <= N ? k2 : k1, k1 <= N ? S1 : N);
    PNPP_REQUIRE(S1 <= N && S2 <= S1, PNPP_ERR_RANGE, "knn_pair: more centres than points");
    PNPP_REQUIRE(k1 <= KNN_KMAX && k2 <= KNN_KMAX, PNPP_ERR_ARG, "knn: nsample exceeds the supported maximum %d", KNN_KMAX);
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "knn: batch %d exceeds grid limit", B);
    ProfScope ps(st, "knn_pair_kernel B=%d | S=%d N=%d k=%d | S=%d N=%d k=%d", B, S1, N, k1, S2, S1, k2);
    const KnnJob J1{nullptr, xyz, nullptr, N, S1, N, k1, idx1, centre1, a1, b1};
    const KnnJob J2{nullptr, xyz, centre1, N, S2, S1, k2, idx2, centre2, a2, b2};
    const int nb2 = cdiv(S2, 4);
    hipLaunchKernelGGL(knn_pair_kernel, dim3(cdiv(S1, 4) + nb2, B), dim3(256), 0, st, J1, J2, nb2);
    PNPP_CHECK_LAUNCH("knn_pair");
    return PNPP_OK;
}

int launch_gather_centres(const float *xyz, const int32_t *centre, int B, int N, int S, float *out_a, float *out_b,
                          hipStream_t st) {
    const int total = B * S;
    ProfScope ps(st, "gather_centres_kernel B=%d S=%d", B, S);
    hipLaunchKernelGGL(gather_centres_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, xyz, centre, N, S, total, out_a, out_b);
    PNPP_CHECK_LAUNCH("gather_centres");
    return PNPP_OK;
}

int launch_scatter_rows_bwd(const float *dout, const int32_t *idx, int B, int N, int C, int M, float *dpoints, hipStream_t st) {
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "index_points_bwd: batch %d exceeds grid limit", B);
    ProfScope ps(st, "scatter_rows_bwd_kernel B=%d N=%d C=%d M=%d", B, N, C, M);
    hipLaunchKernelGGL(scatter_rows_bwd_kernel, dim3(N, B), dim3(64), 0, st, dout, idx, N, C, M, dpoints);
    PNPP_CHECK_LAUNCH("index_points_bwd");
    return PNPP_OK;
}

}  // namespace pnpp

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
using namespace pnpp;

extern "C" int pnpp_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, void *stream) {
    PNPP_REQUIRE(src && dst && out, PNPP_ERR_ARG, "square_distance: null pointer");
    PNPP_REQUIRE(B > 0 && S > 0 && N > 0, PNPP_ERR_ARG, "square_distance: non-positive size");
    PNPP_REQUIRE(B <= 65535 && S <= 65535, PNPP_ERR_ARG, "square_distance: B or S exceeds grid limit");
    hipLaunchKernelGGL(square_distance_kernel, dim3(cdiv(N, 256), S, B), dim3(256), 0, as_stream(stream), src, dst, S, N, out);
    PNPP_CHECK_LAUNCH("square_distance");
    return PNPP_OK;
}

extern "C" int pnpp_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx, void *stream) {
    return launch_knn(new_xyz, xyz, B, S, N, k, idx, as_stream(stream));
}

extern "C" int pnpp_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, void *stream) {
    PNPP_REQUIRE(xyz && start && out, PNPP_ERR_ARG, "fps: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && npoint > 0, PNPP_ERR_ARG, "fps: non-positive size");
    // npoint > N is legal, as in the reference (PointNet++Demo.py:8-29): once every distance is 0 the first maximum is point 0
    hipStream_t st = as_stream(stream);
    // points per thread live in registers; clouds beyond 1024 x 16 points keep the rest in the LDS (16 bytes per point)
    constexpr int kRegPoints = 1024 * 16, kTailMax = (160 * 1024 - 2048) / 16;   // 1 KiB of winner slots at T = 1024
    PNPP_REQUIRE(N <= kRegPoints + kTailMax, PNPP_ERR_ARG, "fps: N=%d exceeds the %d points one CU can hold (registers + LDS)", N,
                 kRegPoints + kTailMax);
    const int ntail = N > kRegPoints ? N - kRegPoints : 0;
    auto go = [&](auto kfn, int T) {
        const size_t lds = ((size_t)2 * 2 * (T / 64) + (size_t)2 * (T / 64) * 4 + (size_t)4 * ntail) * sizeof(float);
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ProfScope ps(st, "fps_kernel<%d> B=%d N=%d npoint=%d", T, B, N, npoint);
        hipLaunchKernelGGL(kfn, dim3(B), dim3(T), lds, st, xyz, N, npoint, start, out, ntail);
    };
    if (N <= 1024) go(fps_kernel<256, 4>, 256);
    else if (N <= 2048) go(fps_kernel<256, 8>, 256);
    else if (N <= 4096) go(fps_kernel<512, 8>, 512);
    else if (N <= 8192) go(fps_kernel<1024, 8>, 1024);
    else go(fps_kernel<1024, 16>, 1024);
    PNPP_CHECK_LAUNCH("fps");
    return PNPP_OK;
}

extern "C" int pnpp_ball_query(const float *new_xyz, const float *xyz, int B, int S, int N, float radius, int nsample,
                               int32_t *idx, void *stream) {
    PNPP_REQUIRE(new_xyz && xyz && idx, PNPP_ERR_ARG, "ball_query: null pointer");
    PNPP_REQUIRE(B > 0 && S > 0 && N > 0 && nsample > 0, PNPP_ERR_ARG, "ball_query: non-positive size");
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "ball_query: batch exceeds grid limit");
    const float r2 = (float)((double)radius * (double)radius);  // Demo.py:65: python float squared, compared in float32
    ProfScope ps(as_stream(stream), "ball_query_kernel B=%d S=%d N=%d nsample=%d", B, S, N, nsample);
    hipLaunchKernelGGL(ball_query_kernel, dim3(cdiv(S, 4), B), dim3(256), 0, as_stream(stream), new_xyz, xyz, S, N, r2,
                       nsample, idx);
    PNPP_CHECK_LAUNCH("ball_query");
    return PNPP_OK;
}

static int sample_random_impl(uint64_t seed, uint64_t stream_id, uint64_t *stream_id_dev, int B, int N, int npoint,
                              int32_t *out, void *stream);

extern "C" int pnpp_sample_random(uint64_t seed, uint64_t stream_id, int B, int N, int npoint, int32_t *out, void *stream) {
    return sample_random_impl(seed, stream_id, nullptr, B, N, npoint, out, stream);
}

extern "C" int pnpp_subsample_points(uint64_t seed, uint64_t stream_id, const float *bank, const int32_t *lengths,
                                     const int32_t *cloud_ids, int B, int Lmax, int num, float *out, void *stream) {
    PNPP_REQUIRE(bank && lengths && out, PNPP_ERR_ARG, "subsample_points: null pointer");
    PNPP_REQUIRE(B > 0 && Lmax > 0 && num > 0, PNPP_ERR_ARG, "subsample_points: non-positive size");
    // four LDS segments of `cap` words: the whole cloud when it is short, else room for four times the keys the cut is expected
    // to keep (num + 4 sqrt(num) + 16, spread over the segments) -- independent of the cloud's length
    const long long want = 4ll * (long long)(num + 4.0 * sqrt((double)num) + 16.0) + 64;
    long long per_seg = ((long long)Lmax + 3) / 4 + 4;
    if (per_seg > want / 4 + 64) per_seg = want / 4 + 64;
    const int cap = (int)((per_seg + 1) & ~1ll);
    const size_t lds = (size_t)4 * cap * sizeof(unsigned long long);
    PNPP_REQUIRE(lds <= 128 * 1024, PNPP_ERR_ARG, "subsample_points: num=%d needs %zu bytes of LDS (<= 128 KB)", num, lds);
    static size_t granted = 0;
    if (lds > 48 * 1024 && lds > granted) {
        (void)hipFuncSetAttribute((const void *)subsample_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = lds;
    }
    // the rank-by-counting pass is quadratic in the ~num survivors: spread a slot over several workgroups until the chip is full
    int nparts = 1;
    while (nparts < 8 && (long long)B * nparts * 2 <= 512 && num >= 128 * nparts * 2) nparts *= 2;
    ProfScope ps(as_stream(stream), "subsample_points_kernel B=%d Lmax=%d num=%d parts=%d", B, Lmax, num, nparts);
    hipLaunchKernelGGL(subsample_points_kernel, dim3(B, nparts), dim3(256), lds, as_stream(stream), (unsigned)seed,
                       (unsigned)(seed >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), bank, lengths, cloud_ids, Lmax,
                       num, out, cap);
    PNPP_CHECK_LAUNCH("subsample_points");
    return PNPP_OK;
}

extern "C" int pnpp_sample_random_dev(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N, int npoint,
                                      int32_t *out, void *stream) {
    PNPP_REQUIRE(stream_id_dev, PNPP_ERR_ARG, "sample_random_dev: null counter pointer");
    return sample_random_impl(seed, offset, stream_id_dev, B, N, npoint, out, stream);
}

static int sample_random_impl(uint64_t seed, uint64_t stream_id, uint64_t *stream_id_dev, int B, int N, int npoint,
                              int32_t *out, void *stream) {
    PNPP_REQUIRE(out, PNPP_ERR_ARG, "sample_random: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && npoint > 0, PNPP_ERR_ARG, "sample_random: non-positive size");
    PNPP_REQUIRE(npoint <= N, PNPP_ERR_RANGE, "sample_random: npoint=%d > N=%d", npoint, N);
    PNPP_REQUIRE((size_t)(N + 1) * 8 <= 128 * 1024, PNPP_ERR_ARG, "sample_random: N=%d too large", N);
    PNPP_REQUIRE(B <= 65535, PNPP_ERR_ARG, "sample_random: batch exceeds grid limit");
    const size_t lds = (size_t)(N + 1) * sizeof(unsigned long long);
    if (lds > 48 * 1024)
        hipFuncSetAttribute((const void *)sample_random_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ProfScope ps(as_stream(stream), "sample_random_kernel B=%d N=%d npoint=%d", B, N, npoint);
    hipLaunchKernelGGL(sample_random_kernel, dim3(B), dim3(256), lds, as_stream(stream), (unsigned)seed,
                       (unsigned)(seed >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32),
                       reinterpret_cast<unsigned long long *>(stream_id_dev), N, npoint, out, 0, 0, 0, (int32_t *)nullptr);
    PNPP_CHECK_LAUNCH("sample_random");
    return PNPP_OK;
}

extern "C" int pnpp_sample_random_dev2(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N1, int npoint1,
                                       int32_t *out1, int N2, int npoint2, int32_t *out2, void *stream) {
    PNPP_REQUIRE(stream_id_dev && out1 && out2, PNPP_ERR_ARG, "sample_random_dev2: null pointer");
    PNPP_REQUIRE(B > 0 && N1 > 0 && npoint1 > 0 && N2 > 0 && npoint2 > 0, PNPP_ERR_ARG, "sample_random_dev2: non-positive size");
    PNPP_REQUIRE(npoint1 <= N1 && npoint2 <= N2, PNPP_ERR_RANGE, "sample_random_dev2: npoint exceeds N");
    const int Nmax = N1 > N2 ? N1 : N2;
    PNPP_REQUIRE((size_t)(Nmax + 1) * 8 <= 128 * 1024, PNPP_ERR_ARG, "sample_random_dev2: N=%d too large", Nmax);
    PNPP_REQUIRE(2 * B <= 65535, PNPP_ERR_ARG, "sample_random_dev2: batch exceeds grid limit");
    const size_t lds = (size_t)(Nmax + 1) * sizeof(unsigned long long);
    if (lds > 48 * 1024)
        hipFuncSetAttribute((const void *)sample_random_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ProfScope ps(as_stream(stream), "sample_random_kernel B=%d N=%d npoint=%d + N=%d npoint=%d", B, N1, npoint1, N2, npoint2);
    hipLaunchKernelGGL(sample_random_kernel, dim3(2 * B), dim3(256), lds, as_stream(stream), (unsigned)seed, (unsigned)(seed >> 32),
                       (unsigned)offset, (unsigned)(offset >> 32), reinterpret_cast<unsigned long long *>(stream_id_dev), N1, npoint1,
                       out1, B, N2, npoint2, out2);
    PNPP_CHECK_LAUNCH("sample_random_dev2");
    return PNPP_OK;
}

extern "C" int pnpp_index_points(const float *points, const int32_t *idx, int B, int N, int C, int M, float *out, void *stream) {
    PNPP_REQUIRE(points && idx && out, PNPP_ERR_ARG, "index_points: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && C > 0 && M > 0, PNPP_ERR_ARG, "index_points: non-positive size");
    const int vec = (C % 4 == 0 && ((uintptr_t)points % 16 == 0) && ((uintptr_t)out % 16 == 0)) ? 4 : 1;
    const size_t total = (size_t)B * M * (C / vec);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), points, idx, N, C, M, total, vec, out);
    PNPP_CHECK_LAUNCH("index_points");
    return PNPP_OK;
}

extern "C" int pnpp_index_points_bwd(const float *dout, const int32_t *idx, int B, int N, int C, int M, float *dpoints,
                                     void *stream) {
    PNPP_REQUIRE(dout && idx && dpoints, PNPP_ERR_ARG, "index_points_bwd: null pointer");
    PNPP_REQUIRE(B > 0 && N > 0 && C > 0 && M > 0, PNPP_ERR_ARG, "index_points_bwd: non-positive size");
    return launch_scatter_rows_bwd(dout, idx, B, N, C, M, dpoints, as_stream(stream));
}
