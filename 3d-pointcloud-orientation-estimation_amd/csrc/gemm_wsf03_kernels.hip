// gemm_wsf03_kernels.hip -- gemm_wsf0_kernel (gemm_wsx_kernels.hip: layer 1's forward product of a grouped level whose layer 0 convolves
// relative coordinates only, layer 0 rebuilt from the coordinates inside the kernel) with its 64 -> 64 product formed from exact three-way
// bf16 splits on v_mfma_f32_32x32x16_bf16 (gemm_wsf3_kernels.hip has the arithmetic).
//
// Reference: models/pointnet_pp_8dir.py:31-41 (grouped_xyz - new_xyz, conv 3 -> 64, BatchNorm, ReLU, conv 64 -> 64).
//
// What changes against gemm_wsf0_kernel.  There the tile s Z_0 + t of a strip (32 rows x 64 channels: two float32 MFMA steps per 32 channels
// on [x y | z 1] x [s w_x, s w_y | s w_z, t]) comes out in accumulator layout (lane = channel, registers = rows), goes through a wave-private
// LDS image to become a row operand, and is multiplied on the float32 pipe: 64 MFMAs of 64 cycles per strip.  Here the SAME two MFMA steps are
// issued with their operands exchanged -- weights as the row operand, coordinates as the column operand -- so the tile comes out TRANSPOSED:
// lane = row of the strip, registers = channels.  The sums are the same fused multiply-adds in the same order (a product does not depend on
// which factor is called A), so the ReLU mask the backward kernel rebuilds from the untransposed form is still the forward's bit for bit
// (tests/test_gpu_levels_routed.py, tests/test_gpu_sa.py hold it).  Eight consecutive registers of that tile ARE one 32x32x16 row operand
// (channel order 16 s + 8 (e >> 2) + 4 lh + (e & 3), e = 0 .. 7; the weight image is laid out in the same order): ReLU, split into three
// bf16 pieces in registers, six MFMAs of 32 cycles per 16 channels and 32 output columns -- 48 per strip, no strip image, and the
// weight fragments are the only LDS reads of the loop.
#include <stdlib.h>

#include "kernels.h"
#include "wsf0_args.h"

namespace pnpp {

typedef __bf16 w03_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w03_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w03_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned w03_pk(float lo, float hi) {
    const w03_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, w03_bf16x2));
}
// two floats -> their three bf16 pieces, packed (lo in the low half)
__device__ __forceinline__ void w03_split2(float v0, float v1, unsigned &h, unsigned &m, unsigned &l) {
    h = w03_pk(v0, v1);
    float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
    m = w03_pk(r0, r1);
    r0 -= __uint_as_float(m << 16), r1 -= __uint_as_float(m & 0xffff0000u);
    l = w03_pk(r0, r1);
}
__device__ __forceinline__ w03_bf16x8 w03_op(uint4 v) { return __builtin_bit_cast(w03_bf16x8, v); }

constexpr int W03_PLANE = 64 * 128;   // bytes: [64 output channels][64 k] bf16

template <int EM>
__global__ void __launch_bounds__(256, 2)
gemm_wsf03_kernel(const Wsf0Args P) {
    constexpr int BN = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds03[];
    unsigned char *Wp = lds03;                                           // three planes of W_1 (see the header for the k order inside a row)
    float *Tc = reinterpret_cast<float *>(lds03 + 3 * W03_PLANE);        // [2][64]: s, t of layer 0
    double *Rm = reinterpret_cast<double *>(Tc + 128);                   // [4][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    auto xs = [](int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); };   // group XOR of image row r (conflict-free ds_read_b128 row reads)
    const int worker = blockIdx.x, nworkers = gridDim.x;
    const int nstrips = P.M / 32, stride = nworkers * 4;
    int strip = worker * 4 + wave;
    const __amdgpu_buffer_rsrc_t resI = wsx_rsrc(P.idx), resX = wsx_rsrc(P.xyz), resC = wsx_rsrc(P.centres);
    const __amdgpu_buffer_rsrc_t resNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.xyz), (short)0, 0, 0x00020000);
    int nidx;
    float px, py, pz, cx, cy, cz;
    auto fetch_idx = [&](__amdgpu_buffer_rsrc_t rI, int s) { nidx = __builtin_bit_cast(int, wsx_load1(rI, 4u * (unsigned)l31, (unsigned)s * 128u)); };
    auto fetch_geo = [&](__amdgpu_buffer_rsrc_t rX, __amdgpu_buffer_rsrc_t rCn, int s) {
        const unsigned cloud = (unsigned)(s / P.S) * (unsigned)P.N * 12u, po = 12u * (unsigned)nidx;
        px = wsx_load1(rX, po, cloud), py = wsx_load1(rX, po + 4u, cloud), pz = wsx_load1(rX, po + 8u, cloud);
        const unsigned co = (unsigned)s * 12u;
        cx = wsx_load1(rCn, 0u, co), cy = wsx_load1(rCn, 4u, co), cz = wsx_load1(rCn, 8u, co);
    };
    // ---- prologue: as gemm_wsf0_kernel (everything it reads is requested before its first wait) ----
    const bool have = strip < nstrips;
    fetch_idx(have ? resI : resNull, have ? strip : 0);
    double mpart = 0.0;
    const int mq = tid & 15, mqc = mq < 9 ? mq : 8;
    if (P.training) {
#pragma unroll
        for (int it = 0; it < kMomSlabs / 16; ++it) {
            const int s = (tid >> 4) + 16 * it;
            const double v = P.mom[(size_t)(s < P.nmom ? s : 0) * kMomPitch + mqc];
            mpart += s < P.nmom ? v : 0.0;
        }
    }
    float w0[3] = {0.f, 0.f, 0.f}, p_g = 1.f, p_b = 0.f, p_bias = 0.f, p_rm = 0.f, p_rv = 0.f;
    if (tid < 64) {
        w0[0] = P.W0[tid * P.ldw0], w0[1] = P.W0[tid * P.ldw0 + 1], w0[2] = P.W0[tid * P.ldw0 + 2];
        if (P.training) {
            if (P.gamma0) p_g = P.gamma0[tid];
            if (P.beta0) p_b = P.beta0[tid];
            if (worker == 0) {
                if (P.bias0) p_bias = P.bias0[tid];
                if (P.rm0) p_rm = P.rm0[tid], p_rv = P.rv0[tid];
            }
        } else {
            p_g = P.scale0[tid], p_b = P.shift0[tid];
        }
    }
    float wl[2][3];   // W_0 rows of this lane's two layer-0 channels
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = j * 32 + l31;
        wl[j][0] = P.W0[ch * P.ldw0], wl[j][1] = P.W0[ch * P.ldw0 + 1], wl[j][2] = P.W0[ch * P.ldw0 + 2];
    }
    constexpr int NWF = 16 * BN / 256;   // W_1[n][0 .. 64): consecutive lanes take consecutive 16-byte groups of a row
    f32x4 tw[NWF];
#pragma unroll
    for (int j = 0; j < NWF; ++j) {
        const int f = tid + 256 * j, nl = f >> 4, k4 = 4 * (f & 15);
        tw[j] = *reinterpret_cast<const f32x4 *>(P.W1 + (size_t)nl * P.ldw1 + k4);
    }
    if (P.training) {
        mpart += shfl_xor_f64(mpart, 16), mpart += shfl_xor_f64(mpart, 32);
        if (lane < 16) Rm[wave * 16 + mq] = mpart;
    }
    // the weight image: row n = 128 bytes; the four k of (n, k4) go to 16-byte group 2 (k4 / 16) + ((k4 / 4) & 1) -- the lane half that reads
    // them -- at its 8-byte half (k4 / 8) & 1: a half-wave's operand of step (j, s) is k = 32 j + 16 s + 4 lh + {0..3}, + 8 + 4 lh + {0..3}
#pragma unroll
    for (int j = 0; j < NWF; ++j) {
        const int f = tid + 256 * j, nl = f >> 4, kq = f & 15;   // kq = k4 / 4
        const int g = 2 * (kq >> 2) + (kq & 1), sub = (kq >> 1) & 1;
        unsigned h0, m0, l0, h1, m1, l1;
        w03_split2(tw[j][0], tw[j][1], h0, m0, l0);
        w03_split2(tw[j][2], tw[j][3], h1, m1, l1);
        unsigned char *d = Wp + nl * 128 + 16 * (g ^ xs(nl)) + 8 * sub;
        *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(d + W03_PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(d + 2 * W03_PLANE) = make_uint2(l0, l1);
    }
    fetch_geo(have ? resX : resNull, have ? resC : resNull, have ? strip : 0);
    __syncthreads();
    if (tid < 64) {
        float sc = p_g, sh = p_b;
        if (P.training) {
            double R[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) R[q] = (Rm[q] + Rm[16 + q]) + (Rm[32 + q] + Rm[48 + q]);
            const double inv = 1.0 / (double)P.M, wx = (double)w0[0], wy = (double)w0[1], wz = (double)w0[2];
            const double mu = (wx * R[0] + wy * R[1] + wz * R[2]) * inv;
            const double e2 = (wx * (wx * R[3] + wy * R[4] + wz * R[5]) + wy * (wx * R[4] + wy * R[6] + wz * R[7]) +
                               wz * (wx * R[5] + wy * R[7] + wz * R[8])) * inv;
            double var = e2 - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)P.eps);
            sc = (float)((double)p_g * is), sh = (float)((double)p_b - mu * (double)p_g * is);
            if (worker == 0) {
                P.mean0[tid] = (float)mu, P.istd0[tid] = (float)is, P.scale0[tid] = sc, P.shift0[tid] = sh;
                if (P.rm0) {
                    const double cnt = (double)P.M, unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                    P.rm0[tid] = (float)((1.0 - (double)P.momentum) * (double)p_rm + (double)P.momentum * (mu + (double)p_bias));
                    P.rv0[tid] = (float)((1.0 - (double)P.momentum) * (double)p_rv + (double)P.momentum * unbiased);
                }
                if (P.nbt0 && tid == 0) *P.nbt0 += 1;
            }
        }
        Tc[tid] = sc, Tc[64 + tid] = sh;
    }
    __syncthreads();   // the weight image and layer 0's constants are complete; from here on the waves run on their own
    float zb0[2], zb1[2];   // [s w_x, s w_y | s w_z, t] of this lane's layer-0 channels: the ROW operand here
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = j * 32 + l31;
        const float sc = Tc[ch], sh = Tc[64 + ch];
        zb0[j] = lh ? sc * wl[j][1] : sc * wl[j][0];
        zb1[j] = lh ? sh : sc * wl[j][2];
    }
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    const unsigned char *brow = Wp + l31 * 128;   // output channel 32 jn + l31: + 4096 jn (x ignores bit 5)
    const int bx = xs(l31);
    for (; strip < nstrips; strip += stride) {
        const bool more = strip + stride < nstrips;
        const int snext = more ? strip + stride : 0;
        const __amdgpu_buffer_rsrc_t nI = more ? resI : resNull, nX = more ? resX : resNull, nC = more ? resC : resNull;
        const float rx = __fsub_rn(px, cx), ry_ = __fsub_rn(py, cy), rz_ = __fsub_rn(pz, cz);
        const float za0 = lh ? ry_ : rx, za1 = lh ? 1.f : rz_;   // [x y | z 1] of row l31: the COLUMN operand here
        fetch_idx(nI, snext);
        f32x16 zt[2];   // (s Z_0 + t)^T: lane = row l31 of the strip, register r = channel 32 j + (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zt[j][r] = 0.f;
            zt[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(zb0[j], za0, zt[j], 0, 0, 0);
            zt[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(zb1[j], za1, zt[j], 0, 0, 0);
        }
        f32x16 acc[2], accs[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jn][r] = 0.f, accs[jn][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 fb[2][3];
                const int g = (2 * (2 * j + s) + lh) ^ bx;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fb[jn][p] = *reinterpret_cast<const uint4 *>(brow + jn * 4096 + p * W03_PLANE + 16 * g);
                uint4 ah, am, al;
                {
                    unsigned h[4], m[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w03_split2(fmaxf(zt[j][8 * s + 2 * e], 0.f), fmaxf(zt[j][8 * s + 2 * e + 1], 0.f), h[e], m[e], l[e]);
                    ah = make_uint4(h[0], h[1], h[2], h[3]), am = make_uint4(m[0], m[1], m[2], m[3]), al = make_uint4(l[0], l[1], l[2], l[3]);
                }
                const w03_bf16x8 a_h = w03_op(ah), a_m = w03_op(am), a_l = w03_op(al);
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const w03_bf16x8 b_h = w03_op(fb[jn][0]), b_m = w03_op(fb[jn][1]), b_l = w03_op(fb[jn][2]);
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc[jn], 0, 0, 0);
                    accs[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, accs[jn], 0, 0, 0);
                    accs[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, accs[jn], 0, 0, 0);
                    accs[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_m, accs[jn], 0, 0, 0);
                    accs[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_h, accs[jn], 0, 0, 0);
                    accs[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_m, accs[jn], 0, 0, 0);
                }
            }
        fetch_geo(nX, nC, snext);   // the next strip's coordinates: its indices were requested a whole product ago
        float *tb = P.z1 + (size_t)(strip * 32 + 4 * lh) * 64 + l31;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[jn][r] + accs[jn][r];
                tb[(size_t)((r & 3) + 8 * (r >> 2)) * 64 + jn * 32] = v;
                t1 += v;
                t2 = fmaf(v, v, t2);
            }
            if constexpr (EM == E_STORE_STATS) s1[jn] += (double)t1, s2[jn] += (double)t2;
        }
    }
    if constexpr (EM == E_STORE_STATS) {
        __syncthreads();   // every wave is done with the weight image
        double *red = reinterpret_cast<double *>(lds03);   // [4 waves][2][64]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double a = s1[j] + shfl_xor_f64(s1[j], 32), b = s2[j] + shfl_xor_f64(s2[j], 32);
            if (lh == 0) red[(wave * 2 + 0) * BN + j * 32 + l31] = a, red[(wave * 2 + 1) * BN + j * 32 + l31] = b;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, cl = tid % BN;
            const double t = (red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl]) +
                             (red[(2 * 2 + which) * BN + cl] + red[(3 * 2 + which) * BN + cl]);
            P.slab[((size_t)worker * 2 + which) * 64 + cl] = t;
        }
    }
}

bool wsf03_enabled() {
    static const bool on = !(getenv("PNPP_WSF03") && atoi(getenv("PNPP_WSF03")) == 0);   // PNPP_WSF03=0: gemm_wsf0_kernel (float32 MFMA; A/B runs)
    return on && split_products();
}
void launch_wsf03(const Wsf0Args &P, int workers, int epilogue_mode, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * W03_PLANE + 128 * sizeof(float) + 64 * sizeof(double);
    if (epilogue_mode == E_STORE_STATS) hipLaunchKernelGGL(gemm_wsf03_kernel<E_STORE_STATS>, dim3(workers), dim3(256), lds, st, P);
    else hipLaunchKernelGGL(gemm_wsf03_kernel<E_STORE>, dim3(workers), dim3(256), lds, st, P);
}

}  // namespace pnpp
