// sampler_device.h -- device-side centre sampling body, shared by sample_random_kernel (index_kernels.hip) and the step tail
// (loss_kernels.hip), which draws the next step's centres in the otherwise idle CUs of its launch.
#pragma once
#include "common.h"

namespace pnpp {

__device__ __forceinline__ unsigned philox_key(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

// One workgroup per cloud.  Only the npoint smallest keys matter, so the keys are first cut at a threshold that keeps
// about npoint + 4 sqrt(npoint) + 16 of them (all of them when that is >= N); the survivors go to LDS as 64-bit (key, index) words --
// the lexicographic order of the reference rank -- and each survivor counts the survivors below it.  A cloud whose
// cut kept fewer than npoint keys (a > 5 sigma event) is redone without the cut, so the result never depends on it.
// A second draw (B1 > 0: workgroups B1.. take N2 / npoint2 / out2 with the NEXT stream id) rides in the same launch:
// two consecutive calls' results from one launch, the device counter advances by two.
// `b` = this workgroup's index among the `nblocks` sampling workgroups of the launch (they may share a launch with other work:
// vm_fc_head_kl_step_kernel draws the NEXT step's centres beside the single workgroup of the step's tail); cand / nc_s: LDS.
__device__ __forceinline__ void sample_random_body(unsigned seed_lo, unsigned seed_hi, unsigned str_lo, unsigned str_hi,
                                                   unsigned long long *__restrict__ str_dev, int N, int npoint,
                                                   int32_t *__restrict__ out, int B1, int N2, int npoint2, int32_t *__restrict__ out2,
                                                   int b, const int nblocks, unsigned long long *__restrict__ cand, int &nc_s) {
    const bool second = B1 > 0 && b >= B1;
    if (second) b -= B1, N = N2, npoint = npoint2, out = out2;
    {   // stream id: (str_hi:str_lo) [+ the device counter: graph replays draw fresh centres] [+ 1 for the second draw]
        unsigned long long sid = (((unsigned long long)str_hi << 32) | str_lo) + (second ? 1ull : 0ull);
        if (str_dev) sid += str_dev[0];
        str_lo = (unsigned)sid, str_hi = (unsigned)(sid >> 32);
    }
    const double keep = (npoint + 4.0 * sqrt((double)npoint) + 16.0) / (double)N;  // >= 5 sigma above npoint survivors
    unsigned cut = keep >= 1.0 ? 0xffffffffu : (unsigned)(keep * 4294967296.0);
    int nc;
    for (;;) {
        if (threadIdx.x == 0) nc_s = 0;
        __syncthreads();
        for (int n0 = 0; n0 < N; n0 += 256) {  // wave-level compaction: one LDS atomic per wave and pass
            const int n = n0 + threadIdx.x;
            const unsigned k = n < N ? philox_key((unsigned)n, (unsigned)b, str_lo, str_hi, seed_lo, seed_hi) : 0u;
            const bool keepit = n < N && k <= cut;
            const unsigned long long vote = __ballot(keepit);
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0 && vote) base = atomicAdd(&nc_s, __popcll(vote));
            base = __shfl(base, 0, 64);
            if (keepit) cand[base + __popcll(vote & ((1ull << lane) - 1ull))] = ((unsigned long long)k << 32) | (unsigned)n;
        }
        __syncthreads();
        nc = nc_s;
        if (nc >= npoint || cut == 0xffffffffu) break;  // uniform: nc_s is the same for every lane
        cut = 0xffffffffu;
        __syncthreads();
    }
    if (nc & 1) {  // pad to an even count with a word above every real one
        if (threadIdx.x == 0) cand[nc] = ~0ull;
        __syncthreads();
    }
    const int nc2 = (nc + 1) >> 1;
    const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(cand);
    for (int i = threadIdx.x; i < nc; i += 256) {
        const unsigned long long mine = cand[i];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < nc2; ++j) {  // one 16-byte LDS broadcast per two candidates
            const ulonglong2 o = c2[j];
            rank += (o.x < mine) + (o.y < mine);
        }
        if (rank < npoint) out[(size_t)b * npoint + rank] = (int32_t)(unsigned)mine;
    }
    if (str_dev) {
        // post-increment of the device counter: every workgroup has read str_dev[0] before it takes a ticket, so
        // the workgroup that takes the last ticket can bump the counter (and clear the ticket word for the next launch)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(&str_dev[1], 1ull);
            if (t == (unsigned long long)nblocks - 1ull) {
                str_dev[1] = 0ull;
                str_dev[0] += B1 > 0 ? 2ull : 1ull;
            }
        }
    }
}


struct SampleJob {   // a two-level centre draw riding in another launch (B <= 0: none); arguments as pnpp_sample_random_dev2
    unsigned seed_lo = 0, seed_hi = 0, str_lo = 0, str_hi = 0;
    unsigned long long *str_dev = nullptr;
    int B = 0, N1 = 0, npoint1 = 0, N2 = 0, npoint2 = 0;
    int32_t *out1 = nullptr, *out2 = nullptr;
};

}  // namespace pnpp
