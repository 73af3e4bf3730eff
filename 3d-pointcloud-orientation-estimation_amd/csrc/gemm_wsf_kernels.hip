// gemm_wsf_kernels.hip -- forward 1x1 convolutions of the grouped levels, weights-stationary, WAVE-PRIVATE row strips: no workgroup
// barrier inside the tile loop.
//
// Reference: models/pointnet_pp_8dir.py:40-42 (conv -> BatchNorm -> ReLU on B x npoint x nsample rows; the BatchNorm + ReLU of the
// previous layer is applied while the operand is staged, the column statistics of this layer are collected in the epilogue, and for a
// level's last layer the max over the 32-row neighbourhood is taken from the accumulators: kernels.h, Epilogue::pool_ext).
//
// Why a second weights-stationary form.  gemm_ws_kernel (gemm_kernels.hip) stages a 64-row tile cooperatively: three barriers per
// tile couple the four waves of a workgroup, each of which shares its SIMD with a wave of ANOTHER workgroup -- the round-3 counters
// show what that costs: cutting a third of the VALU instructions of such a kernel moved the same number of cycles from "issuing" to
// "waiting" and left the launch time where it was.  The forward product needs no cross-row data at all (the fused dW of the backward
// kernels does), so here a wave owns a strip of 32 rows end to end: it loads the strip (global -> registers, one strip ahead), applies
// scale / shift / ReLU, writes it to ITS OWN 8 KB of LDS, and multiplies it with the shared weight panel for every column tile of the
// workgroup's column block.  LDS operations of one wave execute in order, so the strip needs no barrier between its write and its
// reads; the only barriers of the kernel are the one behind the weight panel and the one in front of the final statistics reduction.
// The operand transform is done once per strip for ALL NT x 32 columns (the 64 x 64 form redoes it per 64-column block), and a strip
// IS a neighbourhood (nsample = 32), so the pooled extreme of a column is a reduction over one accumulator tile.
#include <stdlib.h>

#include "kernels.h"

namespace pnpp {

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsf_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ f32x4 wsf_load4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)s_off, 0));
}

#ifndef WSF_EXP   // timing experiments (wrong results): 1 no output stores, 2 no MFMA loop, 4 no operand transform (copy), 8 no statistics / pooling arithmetic
#define WSF_EXP 0
#endif
#ifdef PNPP_STAMPS
__device__ unsigned long long g_wsf_stamps[4][8];   // [instantiation][phase]: s_memtime ticks of wave 0 of workgroup 8 (no extra waits)
#define WSF_STAMP(i)                                                   \
    if (st_on) {                                                       \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += st_t - st_last;                                   \
        st_last = st_t;                                                \
    }
#else
#define WSF_STAMP(i)
#endif

// KD in {64, 128}; NT column tiles (of 32) per wave; AX = A_PLAIN or A_BNRELU; EM = E_STORE or E_STORE_STATS
template <int KD, int NT, int AX, int EM>
__global__ void __launch_bounds__(256, 2)
gemm_wsf_kernel(const float *__restrict__ A, int lda, const float *__restrict__ scale, const float *__restrict__ shift,
                const float *__restrict__ W, int ldw, int M, int Nout, int ncol, const Epilogue E) {
    constexpr int BN = NT * 32, NC = KD / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds;                                   // [BN][KD], 16-byte groups XOR-swizzled by (n & 15)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *As = lds + BN * KD + wave * (32 * 64);      // this wave's strip chunk: [32][64], swizzled by (r & 15)
    const int l31 = lane & 31, lh = lane >> 5;
    auto swz = [](int r) { return (r & 15) << 2; };

    // XCD-aware map (as gemm_ws_kernel): the column blocks of one worker sit on one XCD and share its L2
    const int nworkers = gridDim.x / ncol;
    int col_blk = blockIdx.x % ncol, worker = blockIdx.x / ncol;
    if ((nworkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        col_blk = i % ncol, worker = (i / ncol) * 8 + xcd;
    }
    const int n0 = col_blk * BN;
#ifdef PNPP_STAMPS
    const bool st_on = blockIdx.x == 8 && wave == 0;
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

    // per-channel constants of this lane's column group (k = 64 c + 4 q .. + 3): registers for the whole kernel
    const int q4 = 4 * (lane & 15), rb = lane >> 4;   // staging map: column group lane % 16, rows lane / 16 + 4 i
    float4 sc[NC], sh[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sc[c] = make_float4(1.f, 1.f, 1.f, 1.f), sh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (AX == A_BNRELU) {
            sc[c] = *reinterpret_cast<const float4 *>(scale + 64 * c + q4);
            sh[c] = *reinterpret_cast<const float4 *>(shift + 64 * c + q4);
        }
    }
    // weight panel W[n0 .. n0 + BN)[0 .. KD): consecutive lanes take consecutive 16-byte groups of one row
    {
        constexpr int NWF = (KD / 4) * BN / 256;
        f32x4 tw[NWF];
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 256 * j, nl = f / (KD / 4), k4 = 4 * (f % (KD / 4));
            tw[j] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n0 + nl) * ldw + k4);
        }
#pragma unroll
        for (int j = 0; j < NWF; ++j) {
            const int f = tid + 256 * j, nl = f / (KD / 4), k4 = 4 * (f % (KD / 4));
            *reinterpret_cast<f32x4 *>(Ws + nl * KD + (k4 ^ swz(nl))) = tw[j];
        }
    }

    // strips: this wave takes strip (worker * 4 + wave) + i * (nworkers * 4)
    const int nstrips = M / 32, stride = nworkers * 4;
    int strip = worker * 4 + wave;
    const __amdgpu_buffer_rsrc_t resA = wsf_rsrc(A);
    unsigned oa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = 4u * ((unsigned)(rb + 4 * i) * (unsigned)lda + (unsigned)q4);
    f32x4 ra[NC][8];
    auto fetch = [&](int s) {
        const unsigned so = (unsigned)s * 32u * (unsigned)lda * 4u;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[c][i] = wsf_load4(resA, oa[i] + 256u * (unsigned)c, so);
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
    WSF_STAMP(0)   // prologue

    const float *arow = As + l31 * 64;
    const int ga = (4 * lh) ^ swz(l31);
    for (; strip < nstrips; strip += stride) {
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // stage chunk c of the strip: transform in registers, one ds_write_b128 per group (in-order per wave: the reads of the
            // previous chunk were issued before these writes, and the reads below follow them)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v = ra[c][i];
                if constexpr (AX == A_BNRELU && !(WSF_EXP & 4)) {
                    v[0] = fmaxf(fmaf(v[0], sc[c].x, sh[c].x), 0.f), v[1] = fmaxf(fmaf(v[1], sc[c].y, sh[c].y), 0.f);
                    v[2] = fmaxf(fmaf(v[2], sc[c].z, sh[c].z), 0.f), v[3] = fmaxf(fmaf(v[3], sc[c].w, sh[c].w), 0.f);
                }
                const int r = rb + 4 * i;
                *reinterpret_cast<f32x4 *>(As + r * 64 + (q4 ^ swz(r))) = v;
            }
            WSF_STAMP(1)   // staging (incl. the wait for the strip's loads)
            if (c == NC - 1 && strip + stride < nstrips) fetch(strip + stride);   // the next strip flies during the MFMAs
            // k = 64 c + 8 t + 4 lh + u: one ds_read_b128 of the strip per t, one of the panel per t and column tile
            const float *brow[NT];
            int gb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = j * 32 + l31;
                brow[j] = Ws + n * KD + 64 * c;
                gb[j] = (4 * lh) ^ swz(n);
            }
            float4 fa[2], fb[2][NT];
            auto ld = [&](int buf, int t) {
                fa[buf] = *reinterpret_cast<const float4 *>(arow + ((8 * t) ^ ga));
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[buf][j] = *reinterpret_cast<const float4 *>(brow[j] + ((8 * t) ^ gb[j]));
            };
            auto mm = [&](int buf) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf][j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf][j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf][j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf][j].w, acc[j], 0, 0, 0);
                }
            };
            if (!(WSF_EXP & 2)) {
                ld(0, 0);
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    ld(1, t + 1);
                    mm(0);
                    if (t + 2 < 8) ld(0, t + 2);
                    mm(1);
                }
            }
            WSF_STAMP(2)   // product
        }
        // epilogue of the strip: 16 rows of one column per lane and column tile.  (Round 4: the same tile sent through the wave's own
        // strip buffer and stored as 16 bytes per lane -- whole 128-byte lines, a quarter of the store instructions -- measured 29.5 - 30.0
        // us against 29.2 us on the sa1 launch: what the stores cost here is their 67 MB, not their issue.)
        float *tb = E.c + (size_t)(strip * 32 + 4 * lh) * E.ldc + n0 + l31;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[j][r];
                if (!(WSF_EXP & 1)) tb[(size_t)((r & 3) + 8 * (r >> 2)) * E.ldc + j * 32] = v;
                if (!(WSF_EXP & 8)) t1 += v, t2 = fmaf(v, v, t2);
            }
            if constexpr (EM == E_STORE_STATS && !(WSF_EXP & 8)) {
                s1[j] += (double)t1, s2[j] += (double)t2;
                if (E.pool_ext) {   // the strip is one neighbourhood: extreme pre-BN value per column and its first row
                    float mx = sg[j] * acc[j][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sg[j] * acc[j][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    int a = 64;
#pragma unroll
                    for (int r = 15; r >= 0; --r) a = (sg[j] * acc[j][r] == mx) ? (r & 3) + 8 * (r >> 2) + 4 * lh : a;
                    a = min(a, __shfl_xor(a, 32, 64));
                    if (lh == 0) {
                        const size_t gi = (size_t)strip * E.ldc + n0 + j * 32 + l31;
                        E.pool_ext[gi] = sg[j] * mx;
                        E.pool_arg[gi] = a;
                    }
                }
            }
        }
        WSF_STAMP(3)   // epilogue: stores, statistics, pooling
    }

    if constexpr (EM == E_STORE_STATS) {
        // per-column sums of this worker: lane halves, then the four waves in wave order
        __syncthreads();   // every wave is done with the panel and its strip
        double *red = reinterpret_cast<double *>(lds);   // [4 waves][2][BN]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const double a = s1[j] + shfl_xor_f64(s1[j], 32), b = s2[j] + shfl_xor_f64(s2[j], 32);
            if (lh == 0) red[(wave * 2 + 0) * BN + j * 32 + l31] = a, red[(wave * 2 + 1) * BN + j * 32 + l31] = b;
        }
        __syncthreads();
        for (int f = tid; f < 2 * BN; f += 256) {
            const int which = f / BN, cl = f % BN;
            const double t = (red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl]) +
                             (red[(2 * 2 + which) * BN + cl] + red[(3 * 2 + which) * BN + cl]);
            E.slab[((size_t)worker * 2 + which) * Nout + n0 + cl] = t;
        }
    }
    WSF_STAMP(4)   // tail
#ifdef PNPP_STAMPS
    if (st_on && lane == 0) {
        const int which = (KD == 128 ? 2 : 0) + (Nout >= 256 || (KD == 64 && Nout >= 128) ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) g_wsf_stamps[which][i] += st_acc[i];
    }
#endif
}

// A/B switch: PNPP_NO_WSF=1 keeps the forward products on gemm_ws_kernel
static bool wsf_on() {
    static int cached = -1;
    if (cached < 0) {
        const char *v = getenv("PNPP_NO_WSF");
        cached = (v && atoi(v) != 0) ? 0 : 1;
    }
    return cached != 0;
}

template <int KD, int NT, int AX, int EM>
static void wsf_launch(const AOperand &A, const BOperand &B, int M, int Nout, const Epilogue &E, int workers, int ncol, hipStream_t st) {
    constexpr size_t lds = ((size_t)NT * 32 * KD + 4 * 32 * 64) * sizeof(float);
    auto kfn = gemm_wsf_kernel<KD, NT, AX, EM>;
    static bool granted = false;
    if (lds > 48 * 1024 && !granted) {
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted = true;
    }
    hipLaunchKernelGGL(kfn, dim3(workers * ncol), dim3(256), lds, st, A.a, A.lda, A.scale, A.shift, B.b, B.ldb, M, Nout, ncol, E);
}

// true when the forward product goes out on gemm_wsf_kernel (and its epilogue then honours Epilogue::pool_ext)
bool wsf_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E) {
    if (!wsf_on() || matmul_precision() != 0) return false;
    if (M < 8192 || M % 32 != 0 || Nout % 64 != 0 || !(Kd == 64 || Kd == 128)) return false;
    if (!(A.mode == A_PLAIN || A.mode == A_BNRELU) || !(E.mode == E_STORE || E.mode == E_STORE_STATS)) return false;
    if (!B.trans || B.perm_D >= 0 || (B.rows > 0 && B.rows != Kd) || (B.ldb & 3) != 0 || ((uintptr_t)B.b & 15) != 0) return false;
    if ((A.lda & 3) != 0 || ((uintptr_t)A.a & 15) != 0 || E.ldc != Nout) return false;
    if ((unsigned long long)M * (unsigned)A.lda * 4ull >= 0xfffffff0ull) return false;   // 32-bit buffer offsets
    return true;
}

bool try_launch_wsf(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc) {
    *rc = PNPP_OK;
    if (!wsf_applies(A, B, M, Nout, Kd, E)) return false;
    // column tiles per wave: 4 (128 columns) for K = 64 when the width allows it, else 2; two workgroups per CU either way (64 KB)
    // (NT = 4 for K = 64, N = 128 -- the operand transform done once for all 128 columns -- measured 32.8 us against 31.1 us for the
    //  64 x 64 kernel: 233 registers and a 64-store epilogue per strip; PNPP_WSF_NT4=1 selects it)
    static const bool nt4 = getenv("PNPP_WSF_NT4") && atoi(getenv("PNPP_WSF_NT4")) != 0;
    const int NTsel = (nt4 && Kd == 64 && Nout % 128 == 0) ? 4 : 2;
    const int ncol = Nout / (NTsel * 32), nstrips = M / 32;
    int workers = 512 / ncol;
    if (workers * 4 > nstrips) workers = (nstrips + 3) / 4;
    if (workers > kMaxStatBlocks) workers = kMaxStatBlocks;
    if (workers < 1) workers = 1;
    if (nslab) *nslab = workers;
    ProfScope ps(st, "gemm_wsf_kernel<%d,%d,A%d,E%d> M=%d N=%d K=%d grid=%dx1", Kd, NTsel, A.mode, E.mode, M, Nout, Kd, workers * ncol);
#define PNPP_WSF(KDV, NTV)                                                                                                   \
    {                                                                                                                        \
        if (A.mode == A_BNRELU) {                                                                                            \
            if (E.mode == E_STORE_STATS) wsf_launch<KDV, NTV, A_BNRELU, E_STORE_STATS>(A, B, M, Nout, E, workers, ncol, st); \
            else wsf_launch<KDV, NTV, A_BNRELU, E_STORE>(A, B, M, Nout, E, workers, ncol, st);                               \
        } else {                                                                                                             \
            if (E.mode == E_STORE_STATS) wsf_launch<KDV, NTV, A_PLAIN, E_STORE_STATS>(A, B, M, Nout, E, workers, ncol, st);  \
            else wsf_launch<KDV, NTV, A_PLAIN, E_STORE>(A, B, M, Nout, E, workers, ncol, st);                                \
        }                                                                                                                    \
    }
    if (Kd == 64 && NTsel == 4) PNPP_WSF(64, 4)
    else if (Kd == 64) PNPP_WSF(64, 2)
    else PNPP_WSF(128, 2)
#undef PNPP_WSF
    if (hipGetLastError() != hipSuccess) {
        set_error("gemm_wsf: launch failed");
        *rc = PNPP_ERR_LAUNCH;
    }
    return true;
}

#ifdef PNPP_STAMPS
#define PNPP_WSF_STAMPS_BIT 64u
#else
#define PNPP_WSF_STAMPS_BIT 0u
#endif
unsigned wsf_build_flags() { return ((WSF_EXP != 0) ? 128u : 0u) | PNPP_WSF_STAMPS_BIT; }

}  // namespace pnpp

#ifdef PNPP_STAMPS
extern "C" int pnpp_debug_wsf_stamps(unsigned long long *out32, int reset) {
    if (reset) {
        unsigned long long z[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_wsf_stamps), z, sizeof(z));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out32, HIP_SYMBOL(pnpp::g_wsf_stamps), 32 * sizeof(unsigned long long));
    }
    return 0;
}
#endif
