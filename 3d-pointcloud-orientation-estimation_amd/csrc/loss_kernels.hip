// loss_kernels.hip -- output heads, von-Mises KL losses (value + analytic gradient in one launch),
// soft-label cross entropy, and the flat-buffer optimiser glue.
//
// These are O(batch) scalar kernels: latency-bound, one thread per sample.  All transcendental
// work is done in float64 (fp64 VALU is cheap on CDNA4 and B is tiny) and rounded once to float32.
//
// Bessel functions: exponentially scaled I0e / I1e by the Cephes Chebyshev expansions (the same
// tables ATen's torch.special.i0/i1 evaluate, BSD-licensed Cephes Math Library constants);
// log I0(k) = k + log(i0e(k)) never overflows, A(k) = I1/I0 = i1e/i0e, A'(k) = 1 - A^2 - A/k.
#include "kernels.h"
#include "sampler_device.h"

namespace pnpp {

constexpr double kI0eA[30] = {
    -4.41534164647933937950E-18, 3.33079451882223809783E-17,  -2.43127984654795469359E-16, 1.71539128555513303061E-15,
    -1.16853328779934516808E-14, 7.67618549860493561688E-14,  -4.85644678311192946090E-13, 2.95505266312963983461E-12,
    -1.72682629144155570723E-11, 9.67580903537323691224E-11,  -5.18979560163526290666E-10, 2.65982372468238665035E-9,
    -1.30002500998624804212E-8,  6.04699502254191894932E-8,   -2.67079385394061173391E-7,  1.11738753912010371815E-6,
    -4.41673835845875056359E-6,  1.64484480707288970893E-5,   -5.75419501008210370398E-5,  1.88502885095841655729E-4,
    -5.76375574538582365885E-4,  1.63947561694133579842E-3,   -4.32430999505057594430E-3,  1.05464603945949983183E-2,
    -2.37374148058994688156E-2,  4.93052842396707084878E-2,   -9.49010970480476444210E-2,  1.71620901522208775349E-1,
    -3.04682672343198398683E-1,  6.76795274409476084995E-1};
constexpr double kI0eB[25] = {
    -7.23318048787475395456E-18, -4.83050448594418207126E-18, 4.46562142029675999901E-17,  3.46122286769746109310E-17,
    -2.82762398051658348494E-16, -3.42548561967721913462E-16, 1.77256013305652638360E-15,  3.81168066935262242075E-15,
    -9.55484669882830764870E-15, -4.15056934728722208663E-14, 1.54008621752140982691E-14,  3.85277838274214270114E-13,
    7.18012445138366623367E-13,  -1.79417853150680611778E-12, -1.32158118404477131188E-11, -3.14991652796324136454E-11,
    1.18891471078464383424E-11,  4.94060238822496958910E-10,  3.39623202570838634515E-9,   2.26666899049817806459E-8,
    2.04891858946906374183E-7,   2.89137052083475648297E-6,   6.88975834691682398426E-5,   3.36911647825569408990E-3,
    8.04490411014108831608E-1};
constexpr double kI1eA[29] = {
    2.77791411276104639959E-18, -2.11142121435816608115E-17, 1.55363195773620046921E-16, -1.10559694773538630805E-15,
    7.60068429473540693410E-15, -5.04218550472791168711E-14, 3.22379336594557470981E-13, -1.98397439776494371520E-12,
    1.17361862988909016308E-11, -6.66348972350202774223E-11, 3.62559028155211703701E-10, -1.88724975172282928790E-9,
    9.38153738649577178388E-9,  -4.44505912879632808065E-8,  2.00329475355213526229E-7,  -8.56872026469545474066E-7,
    3.47025130813767847674E-6,  -1.32731636560394358279E-5,  4.78156510755005422638E-5,  -1.61760815825896745588E-4,
    5.12285956168575772895E-4,  -1.51357245063125314899E-3,  4.15642294431288815669E-3,  -1.05640848946261981558E-2,
    2.47264490306265168283E-2,  -5.29459812080949914269E-2,  1.02643658689847095384E-1,  -1.76416518357834055153E-1,
    2.52587186443633654823E-1};
constexpr double kI1eB[25] = {
    7.51729631084210481353E-18,  4.41434832307170791151E-18,  -4.65030536848935832153E-17, -3.20952592199342395980E-17,
    2.96262899764595013876E-16,  3.30820231092092828324E-16,  -1.88035477551078244854E-15, -3.81440307243700780478E-15,
    1.04202769841288027642E-14,  4.27244001671195135429E-14,  -2.10154184277266431302E-14, -4.08355111109219731823E-13,
    -7.19855177624590851209E-13, 2.03562854414708950722E-12,  1.41258074366137813316E-11,  3.25260358301548823856E-11,
    -1.89749581235054123450E-11, -5.58974346219658380687E-10, -3.83538038596423702205E-9,  -2.63146884688951950684E-8,
    -2.51223623787020892529E-7,  -3.88256480887769039346E-6,  -1.10588938762623716291E-4,  -9.76109749136146840777E-3,
    7.78576235018280120474E-1};

// Clenshaw recurrence, fully unrolled over a compile-time table: the coefficients become literals of the instruction stream.
// (Round 4: the tables used to be __device__ arrays read through a pointer inside a runtime loop -- one dependent scalar load per
// term, 25 - 30 terms, four series per KL value: a single kl_multi_eval took 20 - 35 us and the multi-peak loss was a 40 - 100 us
// launch.  Same operations in the same order; the values are unchanged.)
template <int N>
__device__ __forceinline__ double chbevl(double x, const double (&c)[N]) {
    double b0 = c[0], b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int i = 1; i < N; ++i) {
        b2 = b1;
        b1 = b0;
        b0 = x * b1 - b2 + c[i];
    }
    return 0.5 * (b0 - b2);
}
__device__ __forceinline__ double i0e(double x) {  // x >= 0
    return x <= 8.0 ? chbevl(0.5 * x - 2.0, kI0eA) : chbevl(32.0 / x - 2.0, kI0eB) / sqrt(x);
}
__device__ __forceinline__ double i1e(double x) {  // x >= 0
    return x <= 8.0 ? chbevl(0.5 * x - 2.0, kI1eA) * x : chbevl(32.0 / x - 2.0, kI1eB) / sqrt(x);
}
__device__ __forceinline__ double log_i0(double k) { return k + log(i0e(k)); }
__device__ __forceinline__ double bessel_ratio(double k) { return i1e(k) / i0e(k); }
__device__ __forceinline__ double bessel_ratio_prime(double k, double A) {
    return k > 1e-8 ? 1.0 - A * A - A / k : 0.5;
}

constexpr double kPi = 3.14159265358979323846;

// ---- single-peak KL, train_single_peak_vonMises_KL.py:23-28 --------------------------------------
__device__ __forceinline__ void kl_single_eval(double mp, double kp, double mq, double kq, double &kl, double &dmu,
                                               double &dk) {
    const double d = mp - mq;
    const double A = bessel_ratio(kp);
    const double base = log_i0(kq) - log_i0(kp);
    if (kp <= 1e-6) {  // a1 := 0 branch of line 26: only -log I0(kappa_p) depends on the prediction
        kl = base;
        dmu = 0.0;
        dk = -A;
    } else {
        const double cd = cos(d), sd = sin(d);
        kl = base + kp * A - kq * A * cd;
        dmu = kq * A * sd;
        dk = bessel_ratio_prime(kp, A) * (kp - kq * cd);
    }
}

__global__ void __launch_bounds__(256) vm_kl_single_kernel(const float *__restrict__ mu_p, const float *__restrict__ kappa_p,
                                                           const float *__restrict__ mu_q, const float *__restrict__ kappa_q,
                                                           int n, float *__restrict__ kl, float *__restrict__ dmu,
                                                           float *__restrict__ dkappa) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v, a, b;
    kl_single_eval((double)mu_p[i], (double)kappa_p[i], (double)mu_q[i], (double)kappa_q[i], v, a, b);
    kl[i] = (float)v;
    if (dmu) dmu[i] = (float)a;
    if (dkappa) dkappa[i] = (float)b;
}

// fc3 output -> (mu, kappa) -> KL -> d/d(fc3 output); pointnet_pp_vonMises.py:36-37 fused with the loss
__global__ void __launch_bounds__(256) vm_head_kl_kernel(const float *__restrict__ o, const float *__restrict__ mu_gt,
                                                         const float *__restrict__ kappa_gt, int B, float *__restrict__ mu,
                                                         float *__restrict__ kappa, float *__restrict__ loss_vec,
                                                         float *__restrict__ d_o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const double o0 = (double)o[2 * i], o1 = (double)o[2 * i + 1];
    const double th = tanh(o0);
    const float mu_f = (float)(th * kPi);                              // torch.tanh(out[:,0]) * np.pi
    const double sp = o1 > 20.0 ? o1 : log1p(exp(o1));                 // F.softplus (beta 1, threshold 20)
    const float kap_f = (float)sp;
    mu[i] = mu_f;
    kappa[i] = kap_f;
    if (!loss_vec) return;
    double v, a, b;
    kl_single_eval((double)mu_f, (double)kap_f, (double)mu_gt[i], (double)kappa_gt[i], v, a, b);
    loss_vec[i] = (float)v;
    if (d_o) {
        const double sig = o1 > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-o1));
        d_o[2 * i] = (float)(a * kPi * (1.0 - th * th));
        d_o[2 * i + 1] = (float)(b * sig);
    }
}

// The float64 work of a sample is four independent chains of library calls and Chebyshev series (~5 us one after the other on one
// lane).  They cannot be spread over the lanes of a wave (different code per lane = divergence = serial again), so each of a
// 256-thread workgroup's four WAVES evaluates one chain for up to 64 samples, the pieces meet in LDS, and wave 0 combines them with
// the same float64 operations in the same order as kl_single_eval: bitwise the same result, a third of the latency.
struct KlPieces {
    double th[64], cd[64], sd[64], i0e_kp[64], logi0_kp[64], i1e_kp[64], logi0_kq[64], sig[64];
    float mu[64], kap[64];
};
// Called by ALL 256 threads (two barriers inside) for sample i = chunk base + lane; returns true on the wave-0 lane that owns a valid
// sample, with mu, kappa (float32, as the reference's head returns them), the loss value and d loss_mean / d o.
__device__ __forceinline__ bool vm_head_kl_chunk(KlPieces &S, int i, int B, double o0, double o1, const float *__restrict__ mu_gt,
                                                 const float *__restrict__ kappa_gt, double inv_b, float &mu_f, float &kap_f, float &vf,
                                                 float &g0, float &g1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (i < B) {
        if (wave == 0) {
            const double th = tanh(o0);
            const float m = (float)(th * kPi);                               // torch.tanh(out[:,0]) * np.pi
            const double d = (double)m - (double)mu_gt[i];
            S.th[lane] = th, S.cd[lane] = cos(d), S.sd[lane] = sin(d), S.mu[lane] = m;
        } else if (wave == 1) {
            const float k = (float)(o1 > 20.0 ? o1 : log1p(exp(o1)));        // F.softplus (beta 1, threshold 20)
            const double kp = (double)k, e0 = i0e(kp);
            S.i0e_kp[lane] = e0, S.logi0_kp[lane] = kp + log(e0), S.kap[lane] = k;
        } else if (wave == 2) {
            const float k = (float)(o1 > 20.0 ? o1 : log1p(exp(o1)));
            S.i1e_kp[lane] = i1e((double)k);
        } else {
            S.logi0_kq[lane] = log_i0((double)kappa_gt[i]);
            S.sig[lane] = o1 > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-o1));
        }
    }
    __syncthreads();
    bool mine = false;
    if (wave == 0 && i < B) {
        mine = true;
        const double th = S.th[lane], cd = S.cd[lane], sd = S.sd[lane];
        mu_f = S.mu[lane], kap_f = S.kap[lane];
        const double kp = (double)kap_f, kq = (double)kappa_gt[i];
        const double A = S.i1e_kp[lane] / S.i0e_kp[lane];          // bessel_ratio(kp)
        const double basev = S.logi0_kq[lane] - S.logi0_kp[lane];  // log I0(kq) - log I0(kp)
        double v, a, b;
        if (kp <= 1e-6) {
            v = basev, a = 0.0, b = -A;
        } else {
            v = basev + kp * A - kq * A * cd;
            a = kq * A * sd;
            b = bessel_ratio_prime(kp, A) * (kp - kq * cd);
        }
        vf = (float)v;
        g0 = (float)(a * kPi * (1.0 - th * th) * inv_b);
        g1 = (float)(b * S.sig[lane] * inv_b);
    }
    __syncthreads();
    return mine;
}

// head + KL + batch mean + gradient of the mean in ONE single-workgroup launch: the training step's loss tail
// (head, KL, mean, and the three elementwise kernels of their autograd backward) collapses into this and one multiply.
// The mean is a fixed-order fp64 tree, so it is deterministic.
__global__ void __launch_bounds__(256) vm_head_kl_mean_kernel(const float *__restrict__ o, const float *__restrict__ mu_gt,
                                                              const float *__restrict__ kappa_gt, int B, float *__restrict__ mu,
                                                              float *__restrict__ kappa, float *__restrict__ loss_vec,
                                                              float *__restrict__ loss_mean, float *__restrict__ d_o_mean) {
    __shared__ double red[256];
    __shared__ KlPieces S;
    const int lane = threadIdx.x & 63;
    const double inv_b = 1.0 / (double)B;
    double part = 0.0;
    for (int base = 0; base < B; base += 64) {
        const int i = base + lane;
        const double o0 = i < B ? (double)o[2 * i] : 0.0, o1 = i < B ? (double)o[2 * i + 1] : 0.0;
        float mu_f, kap_f, vf, g0, g1;
        if (vm_head_kl_chunk(S, i, B, o0, o1, mu_gt, kappa_gt, inv_b, mu_f, kap_f, vf, g0, g1)) {
            if (mu) mu[i] = mu_f;
            if (kappa) kappa[i] = kap_f;
            if (loss_vec) loss_vec[i] = vf;
            part += (double)vf;
            d_o_mean[2 * i] = g0;
            d_o_mean[2 * i + 1] = g1;
        }
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_mean = (float)(red[0] * inv_b);
}

#ifdef PNPP_STAMPS
__device__ unsigned long long g_tail_stamps[16];   // s_memtime ticks of thread 0 of the tail workgroup, per phase
#define TAIL_STAMP(i)                                                        \
    if (threadIdx.x == 0) {                                                  \
        const unsigned long long st_t = __builtin_amdgcn_s_memtime();        \
        g_tail_stamps[i] += st_t - st_last;                                  \
        st_last = st_t;                                                      \
    }
#else
#define TAIL_STAMP(i)
#endif

// The whole tail of the single-peak training step in ONE single-workgroup launch: o = x W^T + b (the model's fc3,
// pointnet_pp_vonMises.py:35), head activations, KL, batch mean (train_single_peak_vonMises_KL.py:82-84) and their
// backward -- d loss / d W, d b and d x.  x: (B, K) features, W: (2, K).  Eager PyTorch terms: a Linear, the head,
// the loss, a mean and seven autograd nodes.  The mean is a fixed-order fp64 tree; everything is deterministic.
__global__ void __launch_bounds__(256)
vm_fc_head_kl_step_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                          const float *__restrict__ mu_gt, const float *__restrict__ kappa_gt, int B, int K, int x_in_lds,
                          float *__restrict__ loss_mean, float *__restrict__ dW, float *__restrict__ db, float *__restrict__ dx,
                          const SampleJob J) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // o[B][2] (then d_o in place), xs[B][K] when it fits
    // The tail is ONE workgroup on a 256-CU chip.  The centre draw of the NEXT step (models/pointnet_pp_8dir.py:28 of sa1 and sa2)
    // depends on nothing but its counter, so its 2 B workgroups ride in this launch (blocks 1 ..) instead of opening the next step
    // with a launch of their own: same draws, same order, one launch and ~7 us fewer per step.
    if (blockIdx.x > 0) {
        __shared__ int nc_s;
        sample_random_body(J.seed_lo, J.seed_hi, J.str_lo, J.str_hi, J.str_dev, J.N1, J.npoint1, J.out1, J.B, J.N2, J.npoint2, J.out2,
                           (int)blockIdx.x - 1, (int)gridDim.x - 1, reinterpret_cast<unsigned long long *>(sm), nc_s);
        return;
    }
    __shared__ double red[256];
    float *o = sm;
    float *ws = sm + ((2 * B + 3) & ~3);         // W (2 x K) in LDS: with x, ONE round trip for everything the kernel reads
    float *xs = ws + ((2 * K + 3) & ~3);
    const int tid = threadIdx.x;
#ifdef PNPP_STAMPS
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const float *xr = x;  // where the features are read from after staging
    for (int f = tid; f < 2 * K; f += 256) ws[f] = W[f];
    const float b0 = bias[0], b1 = bias[1];
    if (x_in_lds) {       // one pass of independent 16-byte loads; every later read of x is an LDS read
        const int n4 = (B * K) >> 2;
        for (int f = tid; f < n4; f += 256) reinterpret_cast<float4 *>(xs)[f] = reinterpret_cast<const float4 *>(x)[f];
        xr = xs;
    }
    __syncthreads();
    TAIL_STAMP(8)   // staging
    const float *Wl = ws;
    // 1. o = x W^T + b: eight lanes per row, each a strided eighth of the features
    const int part8 = tid & 7;
    for (int i = tid >> 3; i < ((B + 31) & ~31); i += 32) {
        const int ic = min(i, B - 1);
        float a0 = 0.f, a1 = 0.f;
        for (int k = part8; k < K; k += 8) {
            const float xv = xr[(size_t)ic * K + k];
            a0 = fmaf(xv, Wl[k], a0), a1 = fmaf(xv, Wl[K + k], a1);
        }
#pragma unroll
        for (int m = 4; m >= 1; m >>= 1) a0 += __shfl_xor(a0, m), a1 += __shfl_xor(a1, m);
        if (part8 == 0 && i < B) o[2 * i] = a0 + b0, o[2 * i + 1] = a1 + b1;
    }
    __syncthreads();
    TAIL_STAMP(9)   // o = x W^T
    // 2. head + KL + mean (the four float64 chains on the four waves: vm_head_kl_chunk); d loss / d o overwrites o
    __shared__ KlPieces S;
    const double inv_b = 1.0 / (double)B;
    double part = 0.0;
    for (int base = 0; base < B; base += 64) {
        const int i = base + (tid & 63);
        const double o0 = i < B ? (double)o[2 * i] : 0.0, o1 = i < B ? (double)o[2 * i + 1] : 0.0;
        float mu_f, kap_f, vf, g0, g1;
        if (vm_head_kl_chunk(S, i, B, o0, o1, mu_gt, kappa_gt, inv_b, mu_f, kap_f, vf, g0, g1)) {
            part += (double)vf;
            o[2 * i] = g0;      // (every wave read o[2i], o[2i+1] before the first barrier of the chunk)
            o[2 * i + 1] = g1;
        }
    }
    TAIL_STAMP(10)   // head + KL
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) *loss_mean = (float)(red[0] * inv_b);
    TAIL_STAMP(11)   // mean
    // 3. dW = d_o^T x, db = column sums of d_o, dx = d_o W: one thread per feature, rows in order
    for (int k = tid; k < K; k += 256) {
        const float w0 = Wl[k], w1 = Wl[K + k];
        float g0 = 0.f, g1 = 0.f;
#pragma unroll 8
        for (int i = 0; i < B; ++i) {
            const float d0 = o[2 * i], d1 = o[2 * i + 1];
            const float xv = xr[(size_t)i * K + k];
            g0 = fmaf(d0, xv, g0), g1 = fmaf(d1, xv, g1);
            if (dx) dx[(size_t)i * K + k] = fmaf(d0, w0, d1 * w1);
        }
        dW[k] = g0, dW[K + k] = g1;
    }
    if (tid < 2) {
        float g = 0.f;
        for (int i = 0; i < B; ++i) g += o[2 * i + tid];
        db[tid] = g;
    }
    TAIL_STAMP(12)   // dW, db, dx
}

__global__ void __launch_bounds__(256) vm_head_bwd_kernel(const float *__restrict__ o, const float *__restrict__ dmu,
                                                          const float *__restrict__ dkappa, int B, float *__restrict__ d_o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const double o0 = (double)o[2 * i], o1 = (double)o[2 * i + 1];
    const double th = tanh(o0);
    const double sig = o1 > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-o1));
    d_o[2 * i] = (float)((double)dmu[i] * kPi * (1.0 - th * th));
    d_o[2 * i + 1] = (float)((double)dkappa[i] * sig);
}

// ---- multi-peak KL + matching, train_multi_peaks_vonMises_KL.py:38-81 ----------------------------
constexpr int MATCH_KMAX = 8;

__device__ __forceinline__ void kl_multi_eval(double mp, double kp_raw, double mq, double kq_raw, double &kl, double &dmu,
                                              double &dk) {
    const double kp = fmin(fmax(kp_raw, 1e-6), 500.0), kq = fmin(fmax(kq_raw, 1e-6), 500.0);
    double d = fmod(mp - mq + kPi, 2.0 * kPi);  // python %: result takes the sign of the divisor
    if (d < 0.0) d += 2.0 * kPi;
    d -= kPi;
    const double A = bessel_ratio(kp);
    const double cd = cos(d), sd = sin(d);
    kl = (log_i0(kq) - log_i0(kp)) + A * (kp - kq * cd);
    dmu = A * kq * sd;
    const bool pass = kp_raw >= 1e-6 && kp_raw <= 500.0;  // clamp backward
    dk = pass ? bessel_ratio_prime(kp, A) * (kp - kq * cd) : 0.0;
}

// match_loss of one sample in two parts, because the first is what costs: the K x K cost matrix is K^2 independent evaluations of
// kl_multi_eval (two Bessel series, log, cos / sin in float64: ~4 us each on one lane) -- sixteen in a row per sample made the
// one-thread-per-sample kernel a 60 - 100 us launch with 32 lanes of the chip busy (round 4: the largest launch of configs[2]'s step).
// So (1) one THREAD PER MATRIX ENTRY fills cost / gradient tables in LDS, (2) one thread per sample does the assignment and the
// weighted reduction from the tables.  Entry by entry the arithmetic is what the serial form did.
__device__ __forceinline__ void match_cost_entry(float mu_i, float kappa_i, float mq, float kq, float &cost, float &gmu, float &gk) {
    double v, a, c;
    kl_multi_eval((double)mu_i, (double)kappa_i, (double)mq, (double)kq, v, a, c);
    float vf = (float)v;   // rounded to float32 like the reference's cost tensor (line 66-73)
    if (!(fabsf(vf) <= 3.0e38f)) {  // nan_to_num(nan/+-inf -> 1e6): constant, no gradient
        vf = 1e6f;
        a = 0.0;
        c = 0.0;
    }
    cost = vf, gmu = (float)a, gk = (float)c;
}

// cost / gmu / gk: this sample's tables, entry (i, j) at [i * maxK + j], filled for i, j < K.  w (maxK); outputs may be null.
// MK: compile-time bound of maxK.  Everything per-sample lives in REGISTERS: the permutation is eight 4-bit fields of one word and
// every loop over components runs to MK under an `i < K` test, so no array is indexed by a run-time value (round 4: `int perm[8]`
// and friends lived in scratch memory -- a global-memory round trip per access, 24 permutations x ~20 accesses: 42 us of the
// multi-peak step's 71 us tail launch by in-kernel stamps, tools/tail_stamps.py).
template <int MK>
__device__ __forceinline__ void match_assign_sample(const float *cost, const float *gmu, const float *gk, const float *w, int K, int maxK,
                                                    float *loss, float *dmu, float *dkappa, float *dw, int32_t *assign) {
    if (K > maxK) K = maxK;
#pragma unroll
    for (int i = 0; i < MK; ++i)
        if (i < maxK) {
            if (dmu) dmu[i] = 0.f;
            if (dkappa) dkappa[i] = 0.f;
            if (dw) dw[i] = 0.f;
            if (assign) assign[i] = -1;
        }
    if (K <= 0) {
        *loss = 0.f;
        return;
    }
    auto nib = [](unsigned p, int i) -> int { return (int)((p >> (4 * i)) & 15u); };
    auto swp = [](unsigned p, int i, int j) -> unsigned {
        const unsigned x = ((p >> (4 * i)) ^ (p >> (4 * j))) & 15u;
        return p ^ ((x << (4 * i)) | (x << (4 * j)));
    };
    // exhaustive optimal assignment in lexicographic permutation order, first optimum kept
    unsigned perm = 0x76543210u, bestp = perm;
    double best = 1e300;
    while (true) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < MK; ++i)
            if (i < K) t += (double)cost[i * maxK + nib(perm, i)];
        if (t < best) best = t, bestp = perm;
        int i = K - 2;  // next lexicographic permutation of the first K fields
        while (i >= 0 && nib(perm, i) > nib(perm, i + 1)) --i;
        if (i < 0) break;
        int j = K - 1;
        while (nib(perm, j) < nib(perm, i)) --j;
        perm = swp(perm, i, j);
        for (int l = i + 1, r = K - 1; l < r; ++l, --r) perm = swp(perm, l, r);
    }
    // loss_b = sum w_i c_i / (sum w_i + 1e-8)   (float32 arithmetic order of lines 77-80, evaluated in double)
    double sw = 0.0, swc = 0.0;
#pragma unroll
    for (int i = 0; i < MK; ++i)
        if (i < K) {
            sw += (double)w[i];
            swc += (double)w[i] * (double)cost[i * maxK + nib(bestp, i)];
        }
    const double S = sw + 1e-8;
    *loss = (float)(swc / S);
#pragma unroll
    for (int i = 0; i < MK; ++i)
        if (i < K) {
            const double wi = (double)w[i];
            const int j = nib(bestp, i);
            if (assign) assign[i] = j;
            if (dw) dw[i] = (float)(((double)cost[i * maxK + j] * S - swc) / (S * S));
            if (dmu) dmu[i] = (float)(wi / S * (double)gmu[i * maxK + j]);
            if (dkappa) dkappa[i] = (float)(wi / S * (double)gk[i * maxK + j]);
        }
}

// a 256-thread workgroup takes 256 / maxK^2 samples (16 at max_K = 4)
__global__ void __launch_bounds__(256) vm_match_loss_kernel(const float *__restrict__ mu, const float *__restrict__ kappa,
                                                            const float *__restrict__ w, const float *__restrict__ vm_gt,
                                                            const int32_t *__restrict__ K_gt, int B, int maxK, int spb,
                                                            float *__restrict__ loss_vec, float *__restrict__ dmu,
                                                            float *__restrict__ dkappa, float *__restrict__ dw,
                                                            int32_t *__restrict__ assign) {
    __shared__ float tab[3][256];   // cost, gmu, gk: [sample in block][i][j]
    const int kk = maxK * maxK, b0 = blockIdx.x * spb, tid = threadIdx.x;
    if (tid < spb * kk) {
        const int s = tid / kk, ij = tid - s * kk, i = ij / maxK, j = ij - i * maxK, b = b0 + s;
        if (b < B) {
            const int K = min(K_gt[b], maxK);
            if (i < K && j < K)
                match_cost_entry(mu[(size_t)b * maxK + i], kappa[(size_t)b * maxK + i], vm_gt[((size_t)b * maxK + j) * 3],
                                 vm_gt[((size_t)b * maxK + j) * 3 + 1], tab[0][tid], tab[1][tid], tab[2][tid]);
        }
    }
    __syncthreads();
    if (tid < spb && b0 + tid < B) {
        const int b = b0 + tid;
        const size_t o = (size_t)b * maxK;
        match_assign_sample<MATCH_KMAX>(&tab[0][tid * kk], &tab[1][tid * kk], &tab[2][tid * kk], w + o, K_gt[b], maxK, loss_vec + b, dmu ? dmu + o : nullptr,
                            dkappa ? dkappa + o : nullptr, dw ? dw + o : nullptr, assign ? assign + o : nullptr);
    }
}

// ---- multi-peak output head, pointnet_pp_mvM.py:91-125 ---------------------------------------------
// one sample: pi_raw (K), mu_raw (K x 2), kappa_raw (K) -> mu, kappa, weight (K)
__device__ __forceinline__ void mvm_head_sample(const float *pi_raw, const float *mu_raw, const float *kappa_raw, int K, float temp,
                                                float kappa_max, float *mu, float *kappa, float *weight) {
    double mx = -1e300;
    for (int k = 0; k < K; ++k) mx = fmax(mx, (double)pi_raw[k] / (double)temp);
    double se = 0.0;
    for (int k = 0; k < K; ++k) se += exp((double)pi_raw[k] / (double)temp - mx);
    for (int k = 0; k < K; ++k) {
        weight[k] = (float)(exp((double)pi_raw[k] / (double)temp - mx) / se);
        const float c0 = mu_raw[k * 2], s0 = mu_raw[k * 2 + 1];
        const float nrm = fmaxf(sqrtf(c0 * c0 + s0 * s0), 1e-4f);  // F.normalize(eps=1e-4)
        float c = c0 / nrm, s = s0 / nrm;
        if (sqrtf(c * c + s * s) < 1e-3f) c = 1.f, s = 0.f;        // degenerate direction -> mu = 0
        mu[k] = (float)atan2((double)s, (double)c);
        const double kr = (double)kappa_raw[k];
        const double sp = (kr > 20.0 ? kr : log1p(exp(kr))) + 1e-6;
        kappa[k] = (float)fmin(sp, (double)kappa_max);
    }
}

__global__ void __launch_bounds__(64) mvm_head_kernel(const float *__restrict__ pi_raw, const float *__restrict__ mu_raw,
                                                      const float *__restrict__ kappa_raw, int B, int K, float temp,
                                                      float kappa_max, float *__restrict__ mu, float *__restrict__ kappa,
                                                      float *__restrict__ weight) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * K;
    mvm_head_sample(pi_raw + o, mu_raw + 2 * o, kappa_raw + o, K, temp, kappa_max, mu + o, kappa + o, weight + o);
}

// one sample: gradients w.r.t. the raw head outputs from (dmu, dkappa, dweight)
__device__ __forceinline__ void mvm_head_bwd_sample(const float *mu_raw, const float *kappa_raw, const float *weight, const float *dmu,
                                                    const float *dkappa, const float *dweight, int K, float temp, float kappa_max,
                                                    float *dpi_raw, float *dmu_raw, float *dkappa_raw) {
    double dot = 0.0;
    for (int k = 0; k < K; ++k) dot += (double)weight[k] * (double)dweight[k];
    for (int k = 0; k < K; ++k) {
        const double wk = (double)weight[k];
        dpi_raw[k] = (float)(wk * ((double)dweight[k] - dot) / (double)temp);
        // atan2(s, c) with (c, s) = r / max(|r|, eps)
        const double r0 = (double)mu_raw[k * 2], r1 = (double)mu_raw[k * 2 + 1];
        const double n = sqrt(r0 * r0 + r1 * r1);
        const double den = fmax(n, 1e-4);
        const double c = r0 / den, s = r1 / den;
        const double n2 = c * c + s * s;
        double g0 = 0.0, g1 = 0.0;
        if (sqrt(n2) >= 1e-3) {
            const double gm = (double)dmu[k];
            const double dc = -s / n2 * gm, ds = c / n2 * gm;  // d atan2 / d(c, s)
            if (n >= 1e-4) {                                   // u = r/|r|: (I - u u^T)/|r|
                const double proj = dc * c + ds * s;
                g0 = (dc - c * proj) / n;
                g1 = (ds - s * proj) / n;
            } else {                                           // u = r/eps
                g0 = dc / 1e-4;
                g1 = ds / 1e-4;
            }
        }
        dmu_raw[k * 2] = (float)g0;
        dmu_raw[k * 2 + 1] = (float)g1;
        const double kr = (double)kappa_raw[k];
        const double sp = (kr > 20.0 ? kr : log1p(exp(kr))) + 1e-6;
        const double sig = kr > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-kr));
        dkappa_raw[k] = (float)(sp <= (double)kappa_max ? (double)dkappa[k] * sig : 0.0);
    }
}

__global__ void __launch_bounds__(64)
mvm_head_bwd_kernel(const float *__restrict__ pi_raw, const float *__restrict__ mu_raw, const float *__restrict__ kappa_raw,
                    const float *__restrict__ weight, const float *__restrict__ dmu, const float *__restrict__ dkappa,
                    const float *__restrict__ dweight, int B, int K, float temp, float kappa_max,
                    float *__restrict__ dpi_raw, float *__restrict__ dmu_raw, float *__restrict__ dkappa_raw) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    (void)pi_raw;
    const size_t o = (size_t)b * K;
    mvm_head_bwd_sample(mu_raw + 2 * o, kappa_raw + o, weight + o, dmu + o, dkappa + o, dweight + o, K, temp, kappa_max, dpi_raw + o,
                        dmu_raw + 2 * o, dkappa_raw + o);
}

// The whole tail of the multi-peak training step in ONE single-workgroup launch (round 4): the three output heads
// o = x [W_pi; W_mu; W_kappa]^T + b (models/pointnet_pp_mvM.py:91-96), the head activations (:91-125), match_loss
// (train_multi_peaks_vonMises_KL.py:54-81), its batch mean (:229) and their backward -- d loss / d W, d b of the three heads and d x.
// x: (B, K) features; the heads' weights are (KC x K), (2 KC x K), (KC x K).  In eager PyTorch terms: three Linears, the head, the loss, a
// mean and a dozen autograd nodes; as separate launches of this library: 3 x fc_small_fwd, mvm_head, vm_match_loss, torch's mean and
// its backward, three elementwise multiplies, mvm_head_bwd, 3 x fc_small_bwd.  Per-sample arithmetic is that of the standalone kernels
// (the same device functions, float32 values at the same places); the mean is a fixed-order float64 tree.  Workgroups 1 .. carry the
// next step's centre draw, as in vm_fc_head_kl_step_kernel.
template <int KC>
__global__ void __launch_bounds__(256)
mvm_fc_head_match_step_kernel(const float *__restrict__ x, const float *__restrict__ Wpi, const float *__restrict__ bpi,
                              const float *__restrict__ Wmu, const float *__restrict__ bmu, const float *__restrict__ Wkap,
                              const float *__restrict__ bkap, const float *__restrict__ vm_gt, const int32_t *__restrict__ K_gt, int B,
                              int K, float temp, float kappa_max, float *__restrict__ loss_mean, float *__restrict__ dWpi,
                              float *__restrict__ dbpi, float *__restrict__ dWmu, float *__restrict__ dbmu, float *__restrict__ dWkap,
                              float *__restrict__ dbkap, float *__restrict__ dx, float *__restrict__ mu_out, float *__restrict__ kappa_out,
                              float *__restrict__ weight_out, const SampleJob J) {
    constexpr int NO = 4 * KC;   // outputs per sample: pi (KC) | mu (2 KC) | kappa (KC)
    extern __shared__ __attribute__((aligned(16))) float sm[];   // o[B][NO] (then d_o in place), ws[NO][K], xs[B][K]
    if (blockIdx.x > 0) {
        __shared__ int nc_s;
        sample_random_body(J.seed_lo, J.seed_hi, J.str_lo, J.str_hi, J.str_dev, J.N1, J.npoint1, J.out1, J.B, J.N2, J.npoint2, J.out2,
                           (int)blockIdx.x - 1, (int)gridDim.x - 1, reinterpret_cast<unsigned long long *>(sm), nc_s);
        return;
    }
    __shared__ double red[256];
    float *o = sm;
    float *ws = sm + (((size_t)B * NO + 3) & ~(size_t)3);
    float *xs = ws + (size_t)NO * K;
    const int tid = threadIdx.x;
#ifdef PNPP_STAMPS
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    // one round trip for everything the kernel reads: the heads' weight rows in output order, their biases, the features
    for (int f = tid; f < NO * K; f += 256) {   // (the address is selected, not the loaded value: one unconditional load per element)
        const int n = f / K, k = f - n * K;
        const float *src = n < KC ? Wpi + (size_t)n * K + k : n < 3 * KC ? Wmu + (size_t)(n - KC) * K + k : Wkap + (size_t)(n - 3 * KC) * K + k;
        ws[f] = *src;
    }
    float bias_n = 0.f;
    if (tid < NO) bias_n = tid < KC ? bpi[tid] : tid < 3 * KC ? bmu[tid - KC] : bkap[tid - 3 * KC];
    {
        const int n4 = (B * K) >> 2;
        for (int f = tid; f < n4; f += 256) reinterpret_cast<float4 *>(xs)[f] = reinterpret_cast<const float4 *>(x)[f];
    }
    __shared__ float bias_s[NO];
    if (tid < NO) bias_s[tid] = bias_n;
    __syncthreads();
    TAIL_STAMP(0)   // staging
    // 1. o = x W^T + b: eight lanes per row, each a strided eighth of the features, NO accumulators per lane
    const int part8 = tid & 7;
    for (int i = tid >> 3; i < ((B + 31) & ~31); i += 32) {
        const int ic = min(i, B - 1);
        float acc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) acc[n] = 0.f;
        for (int k = part8; k < K; k += 8) {
            const float xv = xs[(size_t)ic * K + k];
#pragma unroll
            for (int n = 0; n < NO; ++n) acc[n] = fmaf(xv, ws[n * K + k], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) {
#pragma unroll
            for (int m = 4; m >= 1; m >>= 1) acc[n] += __shfl_xor(acc[n], m);
            if (part8 == 0 && i < B) o[(size_t)i * NO + n] = acc[n] + bias_s[n];
        }
    }
    __syncthreads();
    // 2. head (one thread per sample), the K x K cost matrices (one thread per ENTRY: match_cost_entry), then per sample the assignment,
    // the mean's factor 1 / B (a float32 multiply, as autograd applies it) and the head's backward; d loss / d o overwrites o.
    // Tables in LDS, reusing nothing the other steps read: hm[B][3 KC] = mu | kappa | weight, tb[3][B][KC^2] = cost | gmu | gk.
    TAIL_STAMP(1)   // o = x W^T
    float *hm = xs + (size_t)B * K;
    float *tb = hm + (size_t)B * 3 * KC;
    const float gf = 1.0f / (float)B;
    for (int b = tid; b < B; b += 256) {
        const float *raw = o + (size_t)b * NO;
        mvm_head_sample(raw, raw + KC, raw + 3 * KC, KC, temp, kappa_max, hm + (size_t)b * 3 * KC, hm + (size_t)b * 3 * KC + KC,
                        hm + (size_t)b * 3 * KC + 2 * KC);
    }
    __syncthreads();
    TAIL_STAMP(2)   // head forward
    for (int e = tid; e < B * KC * KC; e += 256) {
        const int b = e / (KC * KC), ij = e - b * (KC * KC), i = ij / KC, j = ij - i * KC;
        const int Kb = min(K_gt[b], KC);
        if (i < Kb && j < Kb)
            match_cost_entry(hm[(size_t)b * 3 * KC + i], hm[(size_t)b * 3 * KC + KC + i], vm_gt[((size_t)b * KC + j) * 3],
                             vm_gt[((size_t)b * KC + j) * 3 + 1], tb[e], tb[(size_t)B * KC * KC + e], tb[(size_t)2 * B * KC * KC + e]);
    }
    __syncthreads();
    TAIL_STAMP(3)   // cost entries
    double part = 0.0;
    for (int b = tid; b < B; b += 256) {
        float raw[NO], dmu[KC], dk[KC], dw[KC], lv;
#pragma unroll
        for (int n = 0; n < NO; ++n) raw[n] = o[(size_t)b * NO + n];
        const float *mu = hm + (size_t)b * 3 * KC, *kap = mu + KC, *wt = mu + 2 * KC;
        match_assign_sample<KC>(tb + (size_t)b * KC * KC, tb + (size_t)(B + b) * KC * KC, tb + (size_t)(2 * B + b) * KC * KC, wt, K_gt[b], KC, &lv,
                            dmu, dk, dw, nullptr);
        part += (double)lv;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (mu_out) mu_out[(size_t)b * KC + k] = mu[k], kappa_out[(size_t)b * KC + k] = kap[k], weight_out[(size_t)b * KC + k] = wt[k];
            dmu[k] *= gf, dk[k] *= gf, dw[k] *= gf;
        }
        float d[NO];
        mvm_head_bwd_sample(raw + KC, raw + 3 * KC, wt, dmu, dk, dw, KC, temp, kappa_max, d, d + KC, d + 3 * KC);
#pragma unroll
        for (int n = 0; n < NO; ++n) o[(size_t)b * NO + n] = d[n];
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) *loss_mean = (float)(red[0] / (double)B);
    TAIL_STAMP(4)   // assignment + head backward + mean
    // 3. dW = d_o^T x, db = column sums of d_o, dx = d_o W: one thread per feature, rows in order
    for (int k = tid; k < K; k += 256) {
        float wv[NO], g[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) wv[n] = ws[n * K + k], g[n] = 0.f;
        for (int i = 0; i < B; ++i) {
            const float xv = xs[(size_t)i * K + k];
            float dxv = 0.f;
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float dn = o[(size_t)i * NO + n];
                g[n] = fmaf(dn, xv, g[n]);
                dxv = fmaf(dn, wv[n], dxv);
            }
            if (dx) dx[(size_t)i * K + k] = dxv;
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            float *dst = n < KC ? dWpi + (size_t)n * K : n < 3 * KC ? dWmu + (size_t)(n - KC) * K : dWkap + (size_t)(n - 3 * KC) * K;
            dst[k] = g[n];
        }
    }
    if (tid < NO) {
        float gsum = 0.f;
        for (int i = 0; i < B; ++i) gsum += o[(size_t)i * NO + tid];
        float *dst = tid < KC ? dbpi + tid : tid < 3 * KC ? dbmu + (tid - KC) : dbkap + (tid - 3 * KC);
        *dst = gsum;
    }
    TAIL_STAMP(5)   // dW, db, dx
}

// ---- soft-label cross entropy, train_8dir_KL.py:60-68 ----------------------------------------------
__global__ void __launch_bounds__(64) soft_ce_kernel(const float *__restrict__ logits, const float *__restrict__ p, int B,
                                                     int C, float *__restrict__ loss_vec, float *__restrict__ dlogits) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float *l = logits + (size_t)b * C, *pt = p + (size_t)b * C;
    double mx = -1e300;
    for (int c = 0; c < C; ++c) mx = fmax(mx, (double)l[c]);
    double se = 0.0, sp = 0.0;
    for (int c = 0; c < C; ++c) se += exp((double)l[c] - mx), sp += (double)pt[c];
    const double lse = mx + log(se);
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc -= (double)pt[c] * ((double)l[c] - lse);
    loss_vec[b] = (float)acc;
    if (dlogits)
        for (int c = 0; c < C; ++c) dlogits[(size_t)b * C + c] = (float)(exp((double)l[c] - lse) * sp - (double)pt[c]);
}

// ---- flat-buffer Adam (torch.optim.Adam defaults: no amsgrad, no weight decay) ---------------------
// torch.nn.utils.clip_grad_norm_(params, max_norm) (train_multi_peaks_vonMises_KL.py:235) folded into the update: sumsq[0]
// is the sum of squares of the flat gradient buffer as it stands (pnpp_sumsq), gscale the factor that turns the buffer into
// the gradient (1/world under data parallelism), so the gradient's norm is sqrt(sumsq) * gscale and the coefficient is
// min(1, max_norm / (norm + 1e-6)), torch's formula.  A null sumsq means no clipping.  No host round trip.
__device__ __forceinline__ float clip_coef(const double *__restrict__ sumsq, float gscale, float max_norm) {
    if (!sumsq) return 1.f;
    const float norm = (float)(sqrt(sumsq[0]) * (double)gscale);
    return fminf(1.f, max_norm / (norm + 1e-6f));
}
template <bool ZERO>
__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, float lr_over_bc1, float inv_sqrt_bc2,
                                                   float b1, float b2, float eps, float gscale,
                                                   const double *__restrict__ sumsq, float max_norm) {
    gscale *= clip_coef(sumsq, gscale, max_norm);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        if (ZERO) g[i] = 0.f;  // the next iteration's opt.zero_grad(), folded in
    }
}

// The same update with the step count in device memory, for launches that are captured in a hipGraph: state[0] = steps
// taken so far, state[1] = ticket.  Every workgroup reads state[0] before it takes its ticket and the last one
// advances the count, so each replay applies its own bias correction.  zero_grad != 0 clears every gradient once it has
// been consumed (the next step's zero_grad() memset, folded in).
__global__ void __launch_bounds__(256) adam_dev_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                       float *__restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                       float gscale, unsigned long long *__restrict__ state, int zero_grad,
                                                       const double *__restrict__ sumsq, float max_norm) {
    gscale *= clip_coef(sumsq, gscale, max_norm);
    __shared__ float bc[2];
    __shared__ unsigned long long step_s;
    if (threadIdx.x == 0) {
        const unsigned long long step = state[0] + 1ull;
        step_s = step;
        const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
        bc[0] = (float)((double)lr / bc1);
        bc[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float lr_over_bc1 = bc[0], inv_sqrt_bc2 = bc[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        if (zero_grad) g[i] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(&state[1], 1ull);
        if (t == (unsigned long long)gridDim.x - 1ull) {
            state[1] = 0ull;
            state[0] = step_s;
        }
    }
}

__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float *__restrict__ x, size_t n, double *__restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += (double)x[i] * (double)x[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += shfl_xor_f64(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(64) sumsq_final_kernel(const double *__restrict__ part, int nb, double *__restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) acc += part[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += shfl_xor_f64(acc, m);
    if (threadIdx.x == 0) out[0] = acc;
}

// ---- direction-vector heads and their MSE / orthogonality losses (the other set-abstraction models) ----------------
// F.normalize(x, p=2, dim=1, eps): y = x / max(||x||, eps)            (pointnet_pp_Fwd.py:98, Pointnet_pp_xyz.py:84-85)
constexpr int VEC_CMAX = 64;
__global__ void __launch_bounds__(64) l2_normalize_kernel(const float *__restrict__ x, int M, int C, float eps,
                                                          float *__restrict__ y) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const float *xr = x + (size_t)m * C;
    double ss = 0.0;
    for (int c = 0; c < C; ++c) ss += (double)xr[c] * (double)xr[c];
    const double d = fmax(sqrt(ss), (double)eps);
    for (int c = 0; c < C; ++c) y[(size_t)m * C + c] = (float)((double)xr[c] / d);
}

// dx = (dy - y (y . dy)) / ||x||  where ||x|| > eps, dy / eps otherwise (the clamp is then the constant denominator)
__global__ void __launch_bounds__(64) l2_normalize_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                              int M, int C, float eps, float *__restrict__ dx) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const float *xr = x + (size_t)m * C, *gr = dy + (size_t)m * C;
    double ss = 0.0, xg = 0.0;
    for (int c = 0; c < C; ++c) ss += (double)xr[c] * (double)xr[c], xg += (double)xr[c] * (double)gr[c];
    const double nrm = sqrt(ss);
    if (nrm > (double)eps) {
        for (int c = 0; c < C; ++c) dx[(size_t)m * C + c] = (float)(((double)gr[c] - (double)xr[c] * xg / ss) / nrm);
    } else {
        for (int c = 0; c < C; ++c) dx[(size_t)m * C + c] = (float)((double)gr[c] / (double)eps);
    }
}

// nn.MSELoss(): mean over all n elements (train.py:168,183; train_multi_8dir.py:80,100); dp = 2 (p - t) / n.
// One workgroup, fixed-order float64 tree: deterministic.
__global__ void __launch_bounds__(256) mse_kernel(const float *__restrict__ p, const float *__restrict__ t, size_t n,
                                                  float *__restrict__ loss, float *__restrict__ dp) {
    __shared__ double red[256];
    double acc = 0.0;
    const double inv = 1.0 / (double)n;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const double d = (double)p[i] - (double)t[i];
        acc += d * d;
        if (dp) dp[i] = (float)(2.0 * d * inv);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv);
}

// per-row mean squared error: loss_vec[b] = mean_c (p[b,c] - t[b,c])^2 -- the per-sample line of simple_pointnet_train.py:174
// (the mean of these rows is nn.MSELoss() of the batch, :153,182); dp[b,c] = 2 (p - t) / C = d loss_vec[b] / d p[b,c].
__global__ void __launch_bounds__(256) mse_rows_kernel(const float *__restrict__ p, const float *__restrict__ t, int B, int C,
                                                       float *__restrict__ loss_vec, float *__restrict__ dp) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const double inv = 1.0 / (double)C;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)b * C + c;
        const double d = (double)p[i] - (double)t[i];
        acc += d * d;
        if (dp) dp[i] = (float)(2.0 * d * inv);
    }
    loss_vec[b] = (float)(acc * inv);
}

// orthogonality penalty of two predicted axes: mean_b (a_b . b_b)^2   (train.py:184-185)
__global__ void __launch_bounds__(256) orth_loss_kernel(const float *__restrict__ a, const float *__restrict__ b, int B, int C,
                                                        float *__restrict__ loss, float *__restrict__ da,
                                                        float *__restrict__ db) {
    __shared__ double red[256];
    double acc = 0.0;
    const double inv = 1.0 / (double)B;
    for (int m = threadIdx.x; m < B; m += 256) {
        double dot = 0.0;
        for (int c = 0; c < C; ++c) dot += (double)a[(size_t)m * C + c] * (double)b[(size_t)m * C + c];
        acc += dot * dot;
        if (da && db)
            for (int c = 0; c < C; ++c) {
                da[(size_t)m * C + c] = (float)(2.0 * dot * inv * (double)b[(size_t)m * C + c]);
                db[(size_t)m * C + c] = (float)(2.0 * dot * inv * (double)a[(size_t)m * C + c]);
            }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv);
}

// proj_probs (train_multi_8dir.py:41-44): v = normalize(vec); sims = clamp(v D^T, min=0); p = sims / clamp(sum sims, 1e-8)
constexpr int PROJ_DMAX = 16;
__global__ void __launch_bounds__(64) proj_probs_kernel(const float *__restrict__ vec, const float *__restrict__ dirs, int B,
                                                        int D, float *__restrict__ probs) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= B) return;
    const double x0 = vec[3 * m], x1 = vec[3 * m + 1], x2 = vec[3 * m + 2];
    const double d = fmax(sqrt(x0 * x0 + x1 * x1 + x2 * x2), 1e-12);
    const float v0 = (float)(x0 / d), v1 = (float)(x1 / d), v2 = (float)(x2 / d);  // F.normalize's float32 result
    double sims[PROJ_DMAX], sum = 0.0;
    for (int j = 0; j < D; ++j) {
        const double dot = (double)v0 * (double)dirs[3 * j] + (double)v1 * (double)dirs[3 * j + 1] + (double)v2 * (double)dirs[3 * j + 2];
        sims[j] = fmax(dot, 0.0);
        sum += sims[j];
    }
    const double s = fmax(sum, 1e-8);
    for (int j = 0; j < D; ++j) probs[(size_t)m * D + j] = (float)(sims[j] / s);
}

__global__ void __launch_bounds__(64) proj_probs_bwd_kernel(const float *__restrict__ vec, const float *__restrict__ dirs,
                                                            const float *__restrict__ dprobs, int B, int D,
                                                            float *__restrict__ dvec) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= B) return;
    const double x0 = vec[3 * m], x1 = vec[3 * m + 1], x2 = vec[3 * m + 2];
    const double ss = x0 * x0 + x1 * x1 + x2 * x2, nrm = sqrt(ss), d = fmax(nrm, 1e-12);
    const double v0 = x0 / d, v1 = x1 / d, v2 = x2 / d;
    double dots[PROJ_DMAX], sum = 0.0, gs = 0.0;
    for (int j = 0; j < D; ++j) {
        dots[j] = v0 * (double)dirs[3 * j] + v1 * (double)dirs[3 * j + 1] + v2 * (double)dirs[3 * j + 2];
        const double sj = fmax(dots[j], 0.0);
        sum += sj;
        gs += (double)dprobs[(size_t)m * D + j] * sj;
    }
    const double s = fmax(sum, 1e-8);
    const double back = sum >= 1e-8 ? gs / (s * s) : 0.0;  // the clamped denominator is a constant below 1e-8
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;                   // gradient w.r.t. the unit vector
    for (int j = 0; j < D; ++j) {
        if (!(dots[j] >= 0.0)) continue;                    // clamp(min=0) passes the gradient where its input >= 0
        const double gj = (double)dprobs[(size_t)m * D + j] / s - back;
        g0 += gj * (double)dirs[3 * j], g1 += gj * (double)dirs[3 * j + 1], g2 += gj * (double)dirs[3 * j + 2];
    }
    if (nrm > 1e-12) {
        const double vg = v0 * g0 + v1 * g1 + v2 * g2;
        dvec[3 * m] = (float)((g0 - v0 * vg) / nrm), dvec[3 * m + 1] = (float)((g1 - v1 * vg) / nrm);
        dvec[3 * m + 2] = (float)((g2 - v2 * vg) / nrm);
    } else {
        dvec[3 * m] = (float)(g0 / 1e-12), dvec[3 * m + 1] = (float)(g1 / 1e-12), dvec[3 * m + 2] = (float)(g2 / 1e-12);
    }
}

}  // namespace pnpp

using namespace pnpp;

extern "C" int pnpp_vm_kl_single(const float *mu_p, const float *kappa_p, const float *mu_q, const float *kappa_q, int n,
                                 float *kl, float *dmu, float *dkappa, void *stream) {
    PNPP_REQUIRE(mu_p && kappa_p && mu_q && kappa_q && kl, PNPP_ERR_ARG, "vm_kl_single: null pointer");
    PNPP_REQUIRE(n > 0, PNPP_ERR_ARG, "vm_kl_single: n must be positive");
    hipLaunchKernelGGL(vm_kl_single_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), mu_p, kappa_p, mu_q, kappa_q,
                       n, kl, dmu, dkappa);
    PNPP_CHECK_LAUNCH("vm_kl_single");
    return PNPP_OK;
}

extern "C" int pnpp_vm_head_kl(const float *o, const float *mu_gt, const float *kappa_gt, int B, float *mu, float *kappa,
                               float *loss_vec, float *d_o, void *stream) {
    PNPP_REQUIRE(o && mu && kappa, PNPP_ERR_ARG, "vm_head_kl: null pointer");
    PNPP_REQUIRE(!loss_vec || (mu_gt && kappa_gt), PNPP_ERR_ARG, "vm_head_kl: loss requested without ground truth");
    PNPP_REQUIRE(B > 0, PNPP_ERR_ARG, "vm_head_kl: B must be positive");
    hipLaunchKernelGGL(vm_head_kl_kernel, dim3(cdiv(B, 256)), dim3(256), 0, as_stream(stream), o, mu_gt, kappa_gt, B, mu, kappa,
                       loss_vec, d_o);
    PNPP_CHECK_LAUNCH("vm_head_kl");
    return PNPP_OK;
}

extern "C" int pnpp_vm_fc_head_kl_step(const float *x, const float *w, const float *b, const float *mu_gt, const float *kappa_gt, int B,
                                       int K, float *loss_mean, float *dw, float *db, float *dx, void *stream) {
    PNPP_REQUIRE(x && w && b && mu_gt && kappa_gt && loss_mean && dw && db, PNPP_ERR_ARG, "vm_fc_head_kl_step: null pointer");
    PNPP_REQUIRE(B > 0 && K > 0 && B <= 8192, PNPP_ERR_ARG, "vm_fc_head_kl_step: B must be in 1..8192 and K positive");
    const size_t o_floats = ((size_t)2 * B + 3) & ~(size_t)3;
    const int x_in_lds = ((size_t)B * K <= 12288 && (((size_t)B * K) & 3) == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;  // <= 48 KB
    const size_t w_floats = ((size_t)2 * K + 3) & ~(size_t)3;
    const size_t lds = (o_floats + w_floats + (x_in_lds ? (size_t)B * K : 0)) * sizeof(float);
    // fc3's two weight rows and the B x 2 outputs live in LDS beside ~7 KB of static arrays; the 64 KB a launch gets without
    // an opt-in bounds B and K together (the reference head: K = 256).  Wider heads take pnpp_fc_forward + pnpp_vm_head_kl_mean.
    PNPP_REQUIRE(lds <= 56 * 1024, PNPP_ERR_RANGE, "vm_fc_head_kl_step: 2*B + 2*K (+ B*K) floats = %zu bytes of LDS exceed 56 KB (B=%d, K=%d)",
                 lds, B, K);
    hipLaunchKernelGGL(vm_fc_head_kl_step_kernel, dim3(1), dim3(256), lds, as_stream(stream), x, w, b, mu_gt, kappa_gt, B, K, x_in_lds,
                       loss_mean, dw, db, dx, SampleJob());
    PNPP_CHECK_LAUNCH("vm_fc_head_kl_step");
    return PNPP_OK;
}

extern "C" int pnpp_vm_fc_head_kl_step_sample(const float *x, const float *w, const float *b, const float *mu_gt, const float *kappa_gt,
                                              int B, int K, float *loss_mean, float *dw, float *db, float *dx, uint64_t seed,
                                              uint64_t *stream_id_dev, uint64_t offset, int Bs, int N1, int npoint1, int32_t *out1, int N2,
                                              int npoint2, int32_t *out2, void *stream) {
    PNPP_REQUIRE(x && w && b && mu_gt && kappa_gt && loss_mean && dw && db, PNPP_ERR_ARG, "vm_fc_head_kl_step_sample: null pointer");
    PNPP_REQUIRE(B > 0 && K > 0 && B <= 8192, PNPP_ERR_ARG, "vm_fc_head_kl_step_sample: B must be in 1..8192 and K positive");
    PNPP_REQUIRE(stream_id_dev && out1 && out2, PNPP_ERR_ARG, "vm_fc_head_kl_step_sample: null sampler pointer");
    PNPP_REQUIRE(Bs > 0 && N1 > 0 && npoint1 > 0 && N2 > 0 && npoint2 > 0, PNPP_ERR_ARG, "vm_fc_head_kl_step_sample: non-positive sampler size");
    PNPP_REQUIRE(npoint1 <= N1 && npoint2 <= N2, PNPP_ERR_RANGE, "sample_random_dev2: npoint exceeds N");
    const size_t o_floats = ((size_t)2 * B + 3) & ~(size_t)3;
    const int x_in_lds = ((size_t)B * K <= 12288 && (((size_t)B * K) & 3) == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    const size_t w_floats = ((size_t)2 * K + 3) & ~(size_t)3;
    size_t lds = (o_floats + w_floats + (x_in_lds ? (size_t)B * K : 0)) * sizeof(float);
    const int Nmax = N1 > N2 ? N1 : N2;
    const size_t lds_s = (size_t)(Nmax + 1) * sizeof(unsigned long long);   // the sampling workgroups' candidate table
    if (lds_s > lds) lds = lds_s;
    PNPP_REQUIRE(lds <= 56 * 1024, PNPP_ERR_RANGE, "vm_fc_head_kl_step_sample: %zu bytes of LDS exceed 56 KB (B=%d, K=%d, N=%d)", lds, B, K, Nmax);
    SampleJob J;
    J.seed_lo = (unsigned)seed, J.seed_hi = (unsigned)(seed >> 32), J.str_lo = (unsigned)offset, J.str_hi = (unsigned)(offset >> 32);
    J.str_dev = reinterpret_cast<unsigned long long *>(stream_id_dev);
    J.B = Bs, J.N1 = N1, J.npoint1 = npoint1, J.N2 = N2, J.npoint2 = npoint2, J.out1 = out1, J.out2 = out2;
    ProfScope ps(as_stream(stream), "vm_fc_head_kl_step_kernel B=%d K=%d + sample_random B=%d N=%d npoint=%d + N=%d npoint=%d", B, K, Bs, N1,
                 npoint1, N2, npoint2);
    hipLaunchKernelGGL(vm_fc_head_kl_step_kernel, dim3(1 + 2 * Bs), dim3(256), lds, as_stream(stream), x, w, b, mu_gt, kappa_gt, B, K,
                       x_in_lds, loss_mean, dw, db, dx, J);
    PNPP_CHECK_LAUNCH("vm_fc_head_kl_step_sample");
    return PNPP_OK;
}

extern "C" int pnpp_vm_head_kl_mean(const float *o, const float *mu_gt, const float *kappa_gt, int B, float *mu, float *kappa,
                                    float *loss_vec, float *loss_mean, float *d_o_mean, void *stream) {
    PNPP_REQUIRE(o && mu_gt && kappa_gt && loss_mean && d_o_mean, PNPP_ERR_ARG, "vm_head_kl_mean: null pointer");
    PNPP_REQUIRE(B > 0, PNPP_ERR_ARG, "vm_head_kl_mean: B must be positive");
    hipLaunchKernelGGL(vm_head_kl_mean_kernel, dim3(1), dim3(256), 0, as_stream(stream), o, mu_gt, kappa_gt, B, mu, kappa,
                       loss_vec, loss_mean, d_o_mean);
    PNPP_CHECK_LAUNCH("vm_head_kl_mean");
    return PNPP_OK;
}

extern "C" int pnpp_vm_head_bwd(const float *o, const float *dmu, const float *dkappa, int B, float *d_o, void *stream) {
    PNPP_REQUIRE(o && dmu && dkappa && d_o, PNPP_ERR_ARG, "vm_head_bwd: null pointer");
    PNPP_REQUIRE(B > 0, PNPP_ERR_ARG, "vm_head_bwd: B must be positive");
    hipLaunchKernelGGL(vm_head_bwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, as_stream(stream), o, dmu, dkappa, B, d_o);
    PNPP_CHECK_LAUNCH("vm_head_bwd");
    return PNPP_OK;
}

extern "C" int pnpp_vm_match_loss(const float *mu, const float *kappa, const float *w, const float *vm_gt,
                                  const int32_t *K_gt, int B, int maxK, float *loss_vec, float *dmu, float *dkappa, float *dw,
                                  int32_t *assign, void *stream) {
    PNPP_REQUIRE(mu && kappa && w && vm_gt && K_gt && loss_vec, PNPP_ERR_ARG, "vm_match_loss: null pointer");
    PNPP_REQUIRE(B > 0 && maxK > 0, PNPP_ERR_ARG, "vm_match_loss: non-positive size");
    PNPP_REQUIRE(maxK <= MATCH_KMAX, PNPP_ERR_ARG, "vm_match_loss: max_K=%d exceeds the supported maximum %d", maxK, MATCH_KMAX);
    const int spb = 256 / (maxK * maxK);   // samples per workgroup: one thread per cost-matrix entry
    hipLaunchKernelGGL(vm_match_loss_kernel, dim3(cdiv(B, spb)), dim3(256), 0, as_stream(stream), mu, kappa, w, vm_gt, K_gt, B,
                       maxK, spb, loss_vec, dmu, dkappa, dw, assign);
    PNPP_CHECK_LAUNCH("vm_match_loss");
    return PNPP_OK;
}

extern "C" int pnpp_mvm_fc_head_match_step(const float *x, const float *w_pi, const float *b_pi, const float *w_mu, const float *b_mu,
                                           const float *w_kappa, const float *b_kappa, const float *vm_gt, const int32_t *K_gt, int B, int K,
                                           int maxK, float temp, float kappa_max, float *loss_mean, float *dw_pi, float *db_pi, float *dw_mu,
                                           float *db_mu, float *dw_kappa, float *db_kappa, float *dx, float *mu, float *kappa, float *weight,
                                           uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int Bs, int N1, int npoint1,
                                           int32_t *out1, int N2, int npoint2, int32_t *out2, void *stream) {
    PNPP_REQUIRE(x && w_pi && b_pi && w_mu && b_mu && w_kappa && b_kappa && vm_gt && K_gt && loss_mean && dw_pi && db_pi && dw_mu && db_mu &&
                     dw_kappa && db_kappa,
                 PNPP_ERR_ARG, "mvm_fc_head_match_step: null pointer");
    PNPP_REQUIRE(B > 0 && K > 0 && (K & 3) == 0 && ((uintptr_t)x & 15) == 0, PNPP_ERR_ARG,
                 "mvm_fc_head_match_step: B must be positive, K a positive multiple of 4, x 16-byte aligned");
    PNPP_REQUIRE(maxK == 4 || maxK == 8, PNPP_ERR_ARG, "mvm_fc_head_match_step: max_K must be 4 or 8 (got %d); other widths take the separate launches", maxK);
    PNPP_REQUIRE(!mu || (kappa && weight), PNPP_ERR_ARG, "mvm_fc_head_match_step: mu, kappa, weight come together");
    const int NO = 4 * maxK;
    // o[B][NO] | ws[NO][K] | xs[B][K] | mu, kappa, weight [B][3 max_K] | cost, gmu, gk [3][B][max_K^2]
    size_t lds = ((((size_t)B * NO + 3) & ~(size_t)3) + (size_t)NO * K + (size_t)B * K + (size_t)B * 3 * maxK + (size_t)3 * B * maxK * maxK) *
                 sizeof(float);
    SampleJob J;
    int grid = 1;
    if (Bs > 0) {   // the next step's centre draw rides in this launch (sampling.CentreRing)
        PNPP_REQUIRE(stream_id_dev && out1 && out2 && N1 > 0 && npoint1 > 0 && N2 > 0 && npoint2 > 0, PNPP_ERR_ARG,
                     "mvm_fc_head_match_step: bad sampler arguments");
        PNPP_REQUIRE(npoint1 <= N1 && npoint2 <= N2, PNPP_ERR_RANGE, "sample_random_dev2: npoint exceeds N");
        const int Nmax = N1 > N2 ? N1 : N2;
        const size_t lds_s = (size_t)(Nmax + 1) * sizeof(unsigned long long);
        if (lds_s > lds) lds = lds_s;
        J.seed_lo = (unsigned)seed, J.seed_hi = (unsigned)(seed >> 32), J.str_lo = (unsigned)offset, J.str_hi = (unsigned)(offset >> 32);
        J.str_dev = reinterpret_cast<unsigned long long *>(stream_id_dev);
        J.B = Bs, J.N1 = N1, J.npoint1 = npoint1, J.N2 = N2, J.npoint2 = npoint2, J.out1 = out1, J.out2 = out2;
        grid = 1 + 2 * Bs;
    }
    // features, the 4 max_K weight rows, the outputs and the match tables live in LDS (the reference head, B = 32: 59 KB)
    PNPP_REQUIRE(lds <= 96 * 1024, PNPP_ERR_RANGE, "mvm_fc_head_match_step: %zu bytes of LDS exceed 96 KB (B=%d, K=%d, max_K=%d)", lds, B, K, maxK);
    {   // more than the 64 KB a launch gets without asking
        static bool granted[2] = {false, false};
        const int which = maxK == 4 ? 0 : 1;
        if (!granted[which]) {
            const void *fn = maxK == 4 ? (const void *)mvm_fc_head_match_step_kernel<4> : (const void *)mvm_fc_head_match_step_kernel<8>;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            granted[which] = true;
        }
    }
    ProfScope ps(as_stream(stream), "mvm_fc_head_match_step_kernel B=%d K=%d maxK=%d%s", B, K, maxK, Bs > 0 ? " + sample_random" : "");
    if (maxK == 4) {
        hipLaunchKernelGGL(mvm_fc_head_match_step_kernel<4>, dim3(grid), dim3(256), lds, as_stream(stream), x, w_pi, b_pi, w_mu, b_mu, w_kappa,
                           b_kappa, vm_gt, K_gt, B, K, temp, kappa_max, loss_mean, dw_pi, db_pi, dw_mu, db_mu, dw_kappa, db_kappa, dx, mu, kappa,
                           weight, J);
    } else {
        hipLaunchKernelGGL(mvm_fc_head_match_step_kernel<8>, dim3(grid), dim3(256), lds, as_stream(stream), x, w_pi, b_pi, w_mu, b_mu, w_kappa,
                           b_kappa, vm_gt, K_gt, B, K, temp, kappa_max, loss_mean, dw_pi, db_pi, dw_mu, db_mu, dw_kappa, db_kappa, dx, mu, kappa,
                           weight, J);
    }
    PNPP_CHECK_LAUNCH("mvm_fc_head_match_step");
    return PNPP_OK;
}

extern "C" int pnpp_mvm_head(const float *pi_raw, const float *mu_raw, const float *kappa_raw, int B, int K, float temp,
                             float kappa_max, float *mu, float *kappa, float *weight, void *stream) {
    PNPP_REQUIRE(pi_raw && mu_raw && kappa_raw && mu && kappa && weight, PNPP_ERR_ARG, "mvm_head: null pointer");
    PNPP_REQUIRE(B > 0 && K > 0, PNPP_ERR_ARG, "mvm_head: non-positive size");
    hipLaunchKernelGGL(mvm_head_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), pi_raw, mu_raw, kappa_raw, B, K, temp,
                       kappa_max, mu, kappa, weight);
    PNPP_CHECK_LAUNCH("mvm_head");
    return PNPP_OK;
}

extern "C" int pnpp_mvm_head_bwd(const float *pi_raw, const float *mu_raw, const float *kappa_raw, const float *weight,
                                 const float *dmu, const float *dkappa, const float *dweight, int B, int K, float temp,
                                 float kappa_max, float *dpi_raw, float *dmu_raw, float *dkappa_raw, void *stream) {
    PNPP_REQUIRE(pi_raw && mu_raw && kappa_raw && weight && dmu && dkappa && dweight && dpi_raw && dmu_raw && dkappa_raw,
                 PNPP_ERR_ARG, "mvm_head_bwd: null pointer");
    PNPP_REQUIRE(B > 0 && K > 0, PNPP_ERR_ARG, "mvm_head_bwd: non-positive size");
    hipLaunchKernelGGL(mvm_head_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), pi_raw, mu_raw, kappa_raw, weight,
                       dmu, dkappa, dweight, B, K, temp, kappa_max, dpi_raw, dmu_raw, dkappa_raw);
    PNPP_CHECK_LAUNCH("mvm_head_bwd");
    return PNPP_OK;
}

extern "C" int pnpp_l2_normalize(const float *x, int M, int C, float eps, float *y, void *stream) {
    PNPP_REQUIRE(x && y, PNPP_ERR_ARG, "l2_normalize: null pointer");
    PNPP_REQUIRE(M > 0 && C > 0 && C <= VEC_CMAX, PNPP_ERR_ARG, "l2_normalize: M=%d C=%d (C <= %d)", M, C, VEC_CMAX);
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(cdiv(M, 64)), dim3(64), 0, as_stream(stream), x, M, C, eps, y);
    PNPP_CHECK_LAUNCH("l2_normalize");
    return PNPP_OK;
}

extern "C" int pnpp_l2_normalize_bwd(const float *x, const float *dy, int M, int C, float eps, float *dx, void *stream) {
    PNPP_REQUIRE(x && dy && dx, PNPP_ERR_ARG, "l2_normalize_bwd: null pointer");
    PNPP_REQUIRE(M > 0 && C > 0 && C <= VEC_CMAX, PNPP_ERR_ARG, "l2_normalize_bwd: M=%d C=%d (C <= %d)", M, C, VEC_CMAX);
    hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3(cdiv(M, 64)), dim3(64), 0, as_stream(stream), x, dy, M, C, eps, dx);
    PNPP_CHECK_LAUNCH("l2_normalize_bwd");
    return PNPP_OK;
}

extern "C" int pnpp_mse(const float *p, const float *t, size_t n, float *loss, float *dp, void *stream) {
    PNPP_REQUIRE(p && t && loss, PNPP_ERR_ARG, "mse: null pointer");
    PNPP_REQUIRE(n > 0, PNPP_ERR_ARG, "mse: empty input");
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, as_stream(stream), p, t, n, loss, dp);
    PNPP_CHECK_LAUNCH("mse");
    return PNPP_OK;
}

extern "C" int pnpp_mse_rows(const float *p, const float *t, int B, int C, float *loss_vec, float *dp, void *stream) {
    PNPP_REQUIRE(p && t && loss_vec, PNPP_ERR_ARG, "mse_rows: null pointer");
    PNPP_REQUIRE(B > 0 && C > 0, PNPP_ERR_ARG, "mse_rows: B=%d C=%d", B, C);
    hipLaunchKernelGGL(mse_rows_kernel, dim3(cdiv(B, 256)), dim3(256), 0, as_stream(stream), p, t, B, C, loss_vec, dp);
    PNPP_CHECK_LAUNCH("mse_rows");
    return PNPP_OK;
}

extern "C" int pnpp_orth_loss(const float *a, const float *b, int B, int C, float *loss, float *da, float *db, void *stream) {
    PNPP_REQUIRE(a && b && loss, PNPP_ERR_ARG, "orth_loss: null pointer");
    PNPP_REQUIRE((da == nullptr) == (db == nullptr), PNPP_ERR_ARG, "orth_loss: pass both gradient outputs or neither");
    PNPP_REQUIRE(B > 0 && C > 0, PNPP_ERR_ARG, "orth_loss: non-positive size");
    hipLaunchKernelGGL(orth_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, b, B, C, loss, da, db);
    PNPP_CHECK_LAUNCH("orth_loss");
    return PNPP_OK;
}

extern "C" int pnpp_proj_probs(const float *vec, const float *dirs, int B, int D, float *probs, void *stream) {
    PNPP_REQUIRE(vec && dirs && probs, PNPP_ERR_ARG, "proj_probs: null pointer");
    PNPP_REQUIRE(B > 0 && D > 0 && D <= PROJ_DMAX, PNPP_ERR_ARG, "proj_probs: B=%d D=%d (D <= %d)", B, D, PROJ_DMAX);
    hipLaunchKernelGGL(proj_probs_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), vec, dirs, B, D, probs);
    PNPP_CHECK_LAUNCH("proj_probs");
    return PNPP_OK;
}

extern "C" int pnpp_proj_probs_bwd(const float *vec, const float *dirs, const float *dprobs, int B, int D, float *dvec,
                                   void *stream) {
    PNPP_REQUIRE(vec && dirs && dprobs && dvec, PNPP_ERR_ARG, "proj_probs_bwd: null pointer");
    PNPP_REQUIRE(B > 0 && D > 0 && D <= PROJ_DMAX, PNPP_ERR_ARG, "proj_probs_bwd: B=%d D=%d (D <= %d)", B, D, PROJ_DMAX);
    hipLaunchKernelGGL(proj_probs_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), vec, dirs, dprobs, B, D, dvec);
    PNPP_CHECK_LAUNCH("proj_probs_bwd");
    return PNPP_OK;
}

extern "C" int pnpp_soft_ce(const float *logits, const float *p, int B, int C, float *loss_vec, float *dlogits, void *stream) {
    PNPP_REQUIRE(logits && p && loss_vec, PNPP_ERR_ARG, "soft_ce: null pointer");
    PNPP_REQUIRE(B > 0 && C > 0, PNPP_ERR_ARG, "soft_ce: non-positive size");
    hipLaunchKernelGGL(soft_ce_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), logits, p, B, C, loss_vec, dlogits);
    PNPP_CHECK_LAUNCH("soft_ce");
    return PNPP_OK;
}

extern "C" int pnpp_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                              float beta1, float beta2, float eps, float grad_scale, void *stream) {
    PNPP_REQUIRE(param && grad && exp_avg && exp_avg_sq, PNPP_ERR_ARG, "adam_step: null pointer");
    PNPP_REQUIRE(n > 0 && step > 0, PNPP_ERR_ARG, "adam_step: n and step must be positive");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), param, const_cast<float *>(grad), exp_avg,
                       exp_avg_sq, n, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale,
                       (const double *)nullptr, 0.f);
    PNPP_CHECK_LAUNCH("adam_step");
    return PNPP_OK;
}

extern "C" int pnpp_adam_step_zero(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                                   float beta1, float beta2, float eps, float grad_scale, void *stream) {
    PNPP_REQUIRE(param && grad && exp_avg && exp_avg_sq, PNPP_ERR_ARG, "adam_step_zero: null pointer");
    PNPP_REQUIRE(n > 0 && step > 0, PNPP_ERR_ARG, "adam_step_zero: n and step must be positive");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale, (const double *)nullptr, 0.f);
    PNPP_CHECK_LAUNCH("adam_step_zero");
    return PNPP_OK;
}

extern "C" int pnpp_adam_step_clip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                                   float beta1, float beta2, float eps, float grad_scale, const double *grad_sumsq, float max_norm,
                                   int zero_grad, void *stream) {
    PNPP_REQUIRE(param && grad && exp_avg && exp_avg_sq && grad_sumsq, PNPP_ERR_ARG, "adam_step_clip: null pointer");
    PNPP_REQUIRE(n > 0 && step > 0 && max_norm > 0.f, PNPP_ERR_ARG, "adam_step_clip: n, step and max_norm must be positive");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    const float a = (float)((double)lr / bc1), b = (float)(1.0 / sqrt(bc2));
    if (zero_grad)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, a, b,
                           beta1, beta2, eps, grad_scale, grad_sumsq, max_norm);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, a, b,
                           beta1, beta2, eps, grad_scale, grad_sumsq, max_norm);
    PNPP_CHECK_LAUNCH("adam_step_clip");
    return PNPP_OK;
}

extern "C" int pnpp_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, uint64_t *step_state,
                                  float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad, void *stream) {
    PNPP_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_state, PNPP_ERR_ARG, "adam_step_dev: null pointer");
    PNPP_REQUIRE(n > 0, PNPP_ERR_ARG, "adam_step_dev: n must be positive");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, grad_scale, reinterpret_cast<unsigned long long *>(step_state), zero_grad, (const double *)nullptr, 0.f);
    PNPP_CHECK_LAUNCH("adam_step_dev");
    return PNPP_OK;
}

extern "C" int pnpp_adam_step_dev_clip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, uint64_t *step_state,
                                       float lr, float beta1, float beta2, float eps, float grad_scale, const double *grad_sumsq,
                                       float max_norm, int zero_grad, void *stream) {
    PNPP_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_state && grad_sumsq, PNPP_ERR_ARG, "adam_step_dev_clip: null pointer");
    PNPP_REQUIRE(n > 0 && max_norm > 0.f, PNPP_ERR_ARG, "adam_step_dev_clip: n and max_norm must be positive");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, grad_scale, reinterpret_cast<unsigned long long *>(step_state), zero_grad, grad_sumsq, max_norm);
    PNPP_CHECK_LAUNCH("adam_step_dev_clip");
    return PNPP_OK;
}

extern "C" int pnpp_sumsq(const float *x, size_t n, double *out, void *scratch, size_t scratch_bytes, void *stream) {
    PNPP_REQUIRE(x && out && scratch, PNPP_ERR_ARG, "sumsq: null pointer");
    int nb = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    if (nb < 1) nb = 1;
    PNPP_REQUIRE(scratch_bytes >= (size_t)nb * sizeof(double), PNPP_ERR_WORKSPACE, "sumsq: scratch needs %zu bytes",
                 (size_t)nb * sizeof(double));
    double *part = static_cast<double *>(scratch);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), x, n, part);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), part, nb, out);
    PNPP_CHECK_LAUNCH("sumsq");
    return PNPP_OK;
}

#ifdef PNPP_STAMPS
extern "C" int pnpp_debug_tail_stamps(unsigned long long *out16, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pnpp::g_tail_stamps), z, sizeof(z));
    } else {
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(out16, HIP_SYMBOL(pnpp::g_tail_stamps), 16 * sizeof(unsigned long long));
    }
    return 0;
}
#endif
