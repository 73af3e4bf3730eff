// common.h -- shared host/device helpers for libpnpp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pnpp_hip.h"

namespace pnpp {

// ---- error plumbing (no exceptions across the C ABI) -----------------------------------
void set_error(const char *fmt, ...);

#define PNPP_REQUIRE(cond, code, ...)       \
    do {                                    \
        if (!(cond)) {                      \
            ::pnpp::set_error(__VA_ARGS__); \
            return (code);                  \
        }                                   \
    } while (0)

#define PNPP_CHECK_LAUNCH(what)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::pnpp::set_error("%s: launch failed: %s", (what), hipGetErrorString(e_)); \
            return PNPP_ERR_LAUNCH;                                                     \
        }                                                                               \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// carve typed, 256-byte aligned regions out of a caller-owned workspace
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T>
    T *take(size_t n) {
        off = align_up(off, 256);
        T *r = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
    size_t bytes() const { return align_up(off, 256); }
};

// ---- opt-in per-launch HIP-event timing (pnpp_profile_enable / pnpp_profile_report) ----
// A ProfScope names the launch that follows it.  While profiling is on, every kernel launch of the library goes out through
// hipExtLaunchKernelGGL with a start and a stop event ATTACHED TO ITS DISPATCH PACKET: the pair brackets the kernel's own begin and
// end on the device -- what rocprofv3's kernel trace reports -- and not the few microseconds of dispatch latency that a
// hipEventRecord in front of the launch would add to it.
bool prof_on();
void prof_begin(hipStream_t st, const char *fmt, ...);
void prof_end(hipStream_t st);
bool prof_take_events(hipEvent_t *a, hipEvent_t *b);   // events for the next launch of the open scope (false: not profiling)
struct ProfScope {
    hipStream_t st;
    bool on;
    template <typename... Args>
    ProfScope(hipStream_t s, const char *fmt, Args... args) : st(s), on(prof_on()) {
        if (on) prof_begin(st, fmt, args...);
    }
    ~ProfScope() {
        if (on) prof_end(st);
    }
};

}  // namespace pnpp
#if defined(__HIPCC__)
#include <hip/hip_ext.h>
namespace pnpp {
template <typename... KArgs, typename... Args>
inline void launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args &&...args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    hipEvent_t ea, eb;
    if (prof_on() && prof_take_events(&ea, &eb))
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, ea, eb, 0u, static_cast<KArgs>(args)...);
    else
        kernel<<<grid, block, lds, st>>>(static_cast<KArgs>(args)...);
}
}  // namespace pnpp
// every launch site of the library is written hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::pnpp::launch_kernel(kernel, grid, block, lds, stream, __VA_ARGS__)
#endif
namespace pnpp {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMaxStatBlocks = 512;  // upper bound on partial-statistic slabs per GEMM

}  // namespace pnpp

// ---- device helpers ---------------------------------------------------------------------
#if defined(__HIPCC__)
namespace pnpp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor((int)lo, mask, 64);
    hi = __shfl_xor((int)hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl((int)lo, src, 64);
    hi = __shfl((int)hi, src, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// wave-wide minimum of a 64-bit key, result in every lane
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = shfl_xor_u64(v, m);
        v = o < v ? o : v;
    }
    return v;
}

// monotone map float -> uint32 (total order of the IEEE values, -0 < +0, NaNs last)
__device__ __forceinline__ unsigned f32_sortable(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace pnpp
#endif
