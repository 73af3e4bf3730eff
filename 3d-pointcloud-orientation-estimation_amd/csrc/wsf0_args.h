// wsf0_args.h -- what gemm_wsf0_kernel (gemm_wsx_kernels.hip) and its split-product form gemm_wsf03_kernel (gemm_wsf03_kernels.hip) share:
// the argument block, the moment-partial layout and the buffer-load helpers.
#pragma once
#include "kernels.h"

namespace pnpp {

constexpr int kMomSlabs = 128, kMomPitch = 16;

struct Wsf0Args {
    const float *xyz, *centres;
    const int32_t *idx;
    int M, N, S;
    const float *W0;   // 64 x 3, pitch ldw0
    int ldw0;
    // layer 0's BatchNorm: train mode finishes the statistics from the moment partials (mom, nmom); eval mode reads scale0 / shift0
    const double *mom;
    int nmom, training;
    const float *bias0, *gamma0, *beta0;
    float *rm0, *rv0;
    long long *nbt0;
    float momentum, eps;
    float *mean0, *istd0, *scale0, *shift0;
    const float *W1;   // C_1 x 64 as stored (row = output channel)
    int ldw1;
    float *z1;         // M x 64
    double *slab;      // [workers][2][64] (EM == E_STORE_STATS)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsx_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, 0xfffffffe, 0x00020000);
}
__device__ __forceinline__ float wsx_load1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned s_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)s_off, 0));
}

// the split-product form (float32 products from exact three-way bf16 splits; on with split_products() unless PNPP_WSF03=0)
bool wsf03_enabled();
void launch_wsf03(const Wsf0Args &P, int workers, int epilogue_mode, hipStream_t st);

}  // namespace pnpp
