// kernels.h -- internal launcher interface between the kernel translation units and the C ABI.
#pragma once
#include "common.h"

namespace pnpp {

// ---- index_kernels.hip ----
int launch_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx, hipStream_t st);
int launch_knn_centres(const float *xyz, const int32_t *centre, int B, int S, int N, int k, int32_t *idx, float *out_a,
                       float *out_b, hipStream_t st);
int launch_knn_pair(const float *xyz, int B, int N, const int32_t *centre1, int S1, int k1, int32_t *idx1, float *a1, float *b1,
                    const int32_t *centre2, int S2, int k2, int32_t *idx2, float *a2, float *b2, hipStream_t st);
int launch_gather_centres(const float *xyz, const int32_t *centre, int B, int N, int S, float *out_a, float *out_b, hipStream_t st);
int launch_scatter_rows_bwd(const float *dout, const int32_t *idx, int B, int N, int C, int M, float *dpoints, hipStream_t st);

// ---- gemm_kernels.hip ----
// How the A operand (activation rows) of a fused GEMM is produced on the fly.
enum AMode {
    A_PLAIN = 0,   // A[row][k]
    A_BNRELU = 1,  // relu(A[row][k] * scale[k] + shift[k])           (BatchNorm apply + ReLU of the previous layer)
    A_GATHER = 2,  // [points[b, idx[row], :D] | xyz[b, idx[row]] - new_xyz[row / K] | 0]   (grouping, features first)
    A_CONCAT = 3,  // [points[row, :D] | xyz[row] | 0]                                     (group_all)
    A_DZ = 4,      // g[c] * (dy[row][c] - c1[c] - (Z[row][c] - mu[c]) * istd[c] * c2[c])  (BatchNorm backward)
    A_DZ_POOL = 5  // same, with dy never materialised: dy[row][c] = (row % K == arg[row/K][c]) ? dm[row/K][c] : 0
                   // (backward of max-over-nsample + ReLU; a = dm (G x C), arg = arg-max rows)
};

struct AOperand {
    int mode = A_PLAIN;
    const float *a = nullptr;  // PLAIN/BNRELU: matrix; GATHER/CONCAT: points; DZ: dy
    int lda = 0;
    const float *scale = nullptr, *shift = nullptr;  // BNRELU
    const float *xyz = nullptr, *new_xyz = nullptr;  // GATHER/CONCAT
    const int32_t *idx = nullptr;                    // GATHER
    int D = 0, N = 0, S = 0, K = 0;                  // GATHER/CONCAT geometry (N points, S centres per cloud, K neighbours)
    const int32_t *arg = nullptr;                    // DZ_POOL: arg-max neighbour per (group, channel)
    const float *z = nullptr;                        // DZ: pre-BN activations of the same layer
    const float *cst = nullptr;                      // DZ: [5][C] = g, mu, istd, c1, c2
    int C = 0;                                       // DZ: channel count (row pitch of cst)
};

// What happens to each accumulator element.
enum EMode {
    E_STORE = 0,        // C[row][col] = acc
    E_STORE_STATS = 1,  // + per-column sum / sum of squares (float64 partials -> slab)
    E_MASK_STATS = 2,   // v = acc * [scale*Zp+shift > 0]; C = v; stats: sum v, sum v * xhat(Zp)
    E_BN_APPLY = 3      // small-M kernel, all rows in one tile (M <= 32): train-mode BatchNorm1d statistics, running
                        // update, affine, ReLU and dropout mask in the epilogue (C = pre-BN z, Epilogue::bn.y = output)
};

struct BnTail {  // E_BN_APPLY: the fully connected head's Linear -> BatchNorm1d -> ReLU -> Dropout in one launch
    const float *bias = nullptr, *gamma = nullptr, *beta = nullptr;
    float *rm = nullptr, *rv = nullptr;
    long long *nbt = nullptr;
    float momentum = 0.1f, eps = 1e-5f;
    float *mean = nullptr, *istd = nullptr, *scale = nullptr, *shift = nullptr;  // outputs kept for backward
    const uint8_t *mask = nullptr;   // given keep-mask (parity runs, injected masks)
    float drop_scale = 1.f;
    // or: draw the keep-mask here (training with nn.Dropout): Bernoulli(1 - drop_p) bits from Philox4x32-10 keyed by
    // rng_seed with stream id rng_counter[0] (device memory, post-incremented by the kernel through the ticket word
    // rng_counter[1]: graph replays draw fresh masks); the drawn mask is written to mask_out for the backward pass
    uint8_t *mask_out = nullptr;
    float drop_p = 0.f;
    unsigned long long rng_seed = 0;
    unsigned long long *rng_counter = nullptr;
    int relu = 0;
    float *y = nullptr;
};

struct Epilogue {
    int mode = E_STORE;
    float *c = nullptr;
    int ldc = 0;
    double *slab = nullptr;  // [gridDim.x][2][Nout]
    const float *zp = nullptr;  // MASK_STATS: previous layer's pre-BN activations, pitch ldc
    const float *scale = nullptr, *shift = nullptr, *mu = nullptr, *istd = nullptr;
    // MASK_STATS, optional: fuse dW = dZ^T * relu(bn(zp)) into the same launch (the dZ tile is already in LDS);
    // partial sums go to dwslab[worker][Kd][dw_ld] and are combined by launch_slab_reduce
    float *dwslab = nullptr;
    int dw_ld = 0;
    // STORE_STATS of a level's LAST layer, neighbourhoods of 32 rows (one MFMA tile each), optional: the max over the neighbourhood
    // is taken from the accumulators.  relu(scale z + shift) is monotone in z with the sign of gamma, which is known before the
    // batch statistics are: the epilogue writes the extreme pre-BN value pool_ext[g][c] (max z for gamma >= 0, min z otherwise)
    // and its row pool_arg[g][c] (first such row); bn_finalize_fwd then turns pool_ext into the pooled output -- no pooling pass
    // over Z (models/pointnet_pp_8dir.py:42, torch.max(x, 3)[0]).
    float *pool_ext = nullptr;
    int32_t *pool_arg = nullptr;
    const float *pool_gamma = nullptr;
    BnTail bn;
};

// The B operand (weights) is read in place from its state_dict layout -- no transposed copies are made.
struct BOperand {
    const float *b = nullptr;
    int ldb = 0;
    int trans = 0;     // 0: b is [Kd][Nout] (row = reduction index); 1: b is [Nout][Kd] (a conv / linear weight)
    int perm_D = -1;   // trans only: >= 0 maps reduction index k' to weight column (k' < D ? k'+3 : k'-D) and
                       // treats k' >= D+3 as zero (layer 0: kernels order the operand features-first, weights xyz-first)
    int rows = 0;      // number of valid reduction rows (<= Kd; beyond it B is treated as zero)
};

// C[M x Nout] = A'[M x Kd] * B[Kd x Nout].  Returns the number of statistic slabs written (gridDim.x)
// through *nslab when the epilogue collects statistics.
// *dw_slabs (optional) receives the number of dW partial slabs written when E.dwslab was honoured, else 0.
int launch_gemm(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab,
                hipStream_t st, int *dw_slabs = nullptr);

// gemm_bf16_kernels.hip: the bf16-operand / float32-accumulate variant of the weights-stationary kernel (throughput mode,
// off by default; pnpp_set_matmul_precision / PNPP_MATMUL=bf16).  Returns false when the shape stays on the float32 kernels.
int matmul_precision();
void set_matmul_precision(int bf16);
bool try_launch_ws_bf16(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st,
                        int *rc, int *dw_slabs);

// dW[Nc x Kp] = dZ^T[Nc x M] * A2[M x Kp], split over `nsplit` row ranges into slab[nsplit][Nc][kp_pad].
// dz is produced as in A_DZ (or read directly when dz.mode == A_PLAIN); A2 by its own AOperand.
int launch_dw(const AOperand &dz, int Nc, const AOperand &a2, int Kp, int M, float *slab, int nsplit, int kp_pad,
              hipStream_t st);
// out[M x C] = the A_DZ / A_DZ_POOL operand written out densely (used for small M, see gemm_kernels.hip)
int launch_dz_materialize(const AOperand &dz, int M, int C, float *out, hipStream_t st);
// picks the split count / padded pitch launch_dw will use (so callers can size the slab)
void dw_plan(int M, int Nc, int Kp, int *nsplit, int *kp_pad);
// xyz-only layer 0 (second operand A_GATHER with D == 0): streaming kernel, one [Nc][4] partial per 256 rows
// layer 0 of a grouped level with features, convolved before the gather (see gemm_kernels.hip)
bool delayed_layer0_ok(int C);
int launch_gather_rel_stats(const float *P, const AOperand &geo, const float *W0, int ldw, int M, int C, float *z,
                            double *slab, int *nslab, hipStream_t st);
int scatter_dz_splits(int rows);  // number of [C][4] dW_xyz partials launch_scatter_dz writes for `rows` source points
int launch_scatter_dz(const AOperand &dz, const AOperand &geo, int B, int Mc, int C, float *G, float *wslab, hipStream_t st);
int dw_xyz_splits(int M);
int launch_dw_xyz(const AOperand &dz, int Nc, const AOperand &a2, int M, float *slab, hipStream_t st);
// out[c][perm(k)] = sum_s slab[s][c][k]; perm_D < 0: identity; else feature-first -> xyz-first column order.
bool try_launch_fc_dx_dw(const float *dz, const float *w, const float *x, int M, int N, int K, float *dx, float *dw, hipStream_t st,
                         int *rc);
// *nsplit / *kp_pad: in = what dw_plan chose (the slab is sized for it); out = the partial count and pitch actually written
bool try_launch_da_dw(const AOperand &dz, const BOperand &W, int M, int Nout, int Kd, const Epilogue &E, int *nslab, const AOperand &a2,
                      int Kp, float *slab, int *nsplit, int *kp_pad, hipStream_t st, int *rc, float *dw_direct = nullptr, int dw_ld = 0);
// gemm_mid_kernels.hip: 64 x 64 tiles over the whole reduction for the group_all level's wide layers
bool try_launch_mid_gemm(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st,
                         int *rc);
bool mid_gemm_pools(const AOperand &A, int M, int Nout, int Kd);
// gemm_wsf_kernels.hip: forward products of the grouped levels on wave-private row strips (no barrier in the tile loop)
bool try_launch_wsf(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc);
bool wsf_applies(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E);
// gemm_wsf3_kernels.hip: the same launches with the float32 products formed on the bf16 matrix pipe from exact three-way operand
// splits (six bf16 x bf16 products per float32 product, float32 accumulation); pnpp_set_split_products / PNPP_SPLIT_PRODUCTS
int split_products();
void set_split_products(int on);
bool try_launch_wsf3(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc);
// gemm_wsd3_kernels.hip: the fused backward products (dA + mask + sums + dW) of the grouped levels with split products, a producer and a
// consumer wave per strip: a level's last layer (pooled gradient, 128 -> 64 and 256 -> 128) and a middle layer (dense gradient, 128 -> 128)
bool try_launch_wsd3(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                     int *dw_slabs);
int wsd3_timeouts();
// gemm_wsp_kernels.hip: the fused backward product (dA + mask + sums + dW) of a grouped layer with a 64-channel input, wave-private strips
bool try_launch_wsp(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                    int *dw_slabs);
// gemm_wsq_kernels.hip: the fused backward product of a level's last layer at 256 channels (K = 256, N = 128): 64-row tiles, the
// BatchNorm-backward transform folded into the products, dW rows in accumulator order
bool try_launch_wsq(const AOperand &A, const BOperand &B, int M, int Nout, int Kd, const Epilogue &E, int *nslab, hipStream_t st, int *rc,
                    int *dw_slabs);
// gemm_wsx_kernels.hip: backward through layer 1 of a grouped level whose layer 0 convolves relative coordinates only (D == 0), with
// layer 0's backward folded in: Z_0 is rebuilt from the coordinates, dY_0 is never written; launch_xyz0_post turns the workers' sums
// into dW_1 and layer 0's parameter gradients
bool try_launch_wsx(const AOperand &dz, const BOperand &W, int M, int C1, int C0, const AOperand &geo, const float *W0, int ldw0,
                    const float *scale0, const float *shift0, float *dwslab, double *stat, int *workers_out, hipStream_t st, int *rc);
size_t wsx_stat_doubles(int M);
// the forward half: layer 0's statistics from the moments of the relative coordinates, layer 1's product with its operand built from
// the coordinates -- Z_0 is never stored.  xyz0_applies decides for BOTH directions of a level (forward keeps no Z_0 for a generic backward)
bool xyz0_applies(int M, int D, int K, int group_all, int L, const int *C);
size_t xyz0_moment_doubles();
int launch_rel_moments(const AOperand &geo, int M, double *mom, int *nmom, hipStream_t st);
int launch_wsf0(const AOperand &geo, int M, const float *W0, int ldw0, const double *mom, int nmom, int training, const float *bias0,
                const float *gamma0, const float *beta0, float *rm0, float *rv0, long long *nbt0, float momentum, float eps, float *mean0,
                float *istd0, float *scale0, float *shift0, const float *W1, int ldw1, const Epilogue &E, int *nslab, hipStream_t st);
int launch_xyz0_post(const float *dwslab, int workers, int C1, float *dw1, int ld1, const double *stat, const float *W0, int ldw0,
                     const float *gamma0, const float *mean0, const float *istd0, double count, int training, float *dW0, int ld0,
                     float *dgamma0, float *dbeta0, float *dbias0, hipStream_t st);
// diagnostics (pnpp_sa_saved_relu_mask): the ReLU decisions of a stored layer / of the rebuilt layer 0 of a level on raw coordinates
int launch_relu_mask(const float *z, const float *scale, const float *shift, size_t n, int C, uint8_t *out, hipStream_t st);
int launch_xyz0_mask(const AOperand &geo, int M, const float *W0, int ldw0, const float *scale0, const float *shift0, uint8_t *out,
                     hipStream_t st);
// per translation unit: bits of pnpp_build_flags() (experiment / stamp switches compiled in; zero in a library that ships)
unsigned wsx_build_flags();
unsigned wsf_build_flags();
unsigned wsd3_build_flags();
unsigned mid3_build_flags();   // bit 9: gemm_mid3_kernel's phase stamps compiled in
unsigned wsp_build_flags();
unsigned wsq_build_flags();
unsigned gemm_build_flags();
unsigned fc_build_flags();
bool try_launch_mid_da_dw(const AOperand &dz, const BOperand &W, int M, int Nout, int Kd, const Epilogue &E, int *nslab, const AOperand &a2,
                          int Kp, float *slab, int *nsplit_out, int *kp_pad_out, hipStream_t st, int *rc, float *dw_direct = nullptr,
                          int dw_ld = 0);   // dw_direct (Nc x dw_ld, dw_ld == Kp): written in place when one row range suffices; *nsplit_out = 0 then
int launch_slab_reduce2(const float *slab1, int nsplit1, int Nc1, int kp_pad1, int Kvalid1, float *out1, int ldo1, const float *slab2,
                        int nsplit2, int Nc2, int kp_pad2, int Kvalid2, float *out2, int ldo2, hipStream_t st);
int launch_slab_reduce(const float *slab, int nsplit, int Nc, int kp_pad, int Kvalid, int perm_D, float *out, int ldo,
                       hipStream_t st);

int launch_bn_finalize_fwd(const double *slab, int nslab, int C, double count, const float *bias, const float *gamma,
                           const float *beta, float *rm, float *rv, long long *nbt, float momentum, float eps, int training,
                           float *mean, float *istd, float *scale, float *shift, hipStream_t st, const double *count_dev = nullptr,
                           const float *pool_ext = nullptr, float *pool_out = nullptr, int G = 0, int32_t *pool_arg = nullptr,
                           float *origin_a = nullptr, float *origin_b = nullptr, int norigin = 0);
// true when launch_gemm will honour Epilogue::pool_ext for this shape (the weights-stationary kernel, 32-row neighbourhoods)
bool gemm_pools_in_epilogue(const AOperand &A, int M, int Nout, int Kd, int nsample);
// pooled (optional): levels with few groups take the column sums (and, in the dZ job, the masked pooled gradient) straight from
// dout / zsel -- the pool_bwd launch and its dm tensor are not needed then (dz->a must be dout)
struct PooledSource {
    const float *dout = nullptr, *zsel = nullptr, *scale = nullptr, *shift = nullptr;
    int G = 0;
};
int launch_bn_finalize_bwd(const double *slab, int nslab, int C, double count, int training, const float *gamma,
                           const float *mean, const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias,
                           hipStream_t st, const AOperand *dz = nullptr, int M = 0, float *dz_out = nullptr,
                           const double *count_dev = nullptr, const double *local = nullptr, const PooledSource *pooled = nullptr);

// bn_finalize_bwd of one layer and the weight-gradient slab reduction of the layer above it, in one launch
// dz / M / dz_out (optional): also materialise dZ = BN-backward(dz operand) of the layer being finalised (small-M levels)
int launch_post_gemm(const double *slab, int nslab, int C, double count, int training, const float *gamma, const float *mean,
                     const float *istd, float *cst, float *dgamma, float *dbeta, float *dbias, const float *dwslab, int nsplit,
                     int Nc, int kp_pad, int Kvalid, int perm_D, float *dw, int ldo, hipStream_t st, const AOperand *dz = nullptr, int M = 0,
                     float *dz_out = nullptr, const double *count_dev = nullptr, const double *local = nullptr);

// ---- SyncBN (off unless pnpp_set_stats_exchange has registered a callback; sa_api.hip) ----
// What a BatchNorm finalisation reads: the partial slabs of this rank, or -- after stats_exchange -- one slab of sums over all
// ranks with their row count in device memory, plus this rank's own sums for the parameter gradients.
struct StatsView {
    const double *slab = nullptr;
    int nslab = 0;
    const double *count_dev = nullptr, *local = nullptr;
};
bool stats_sync_on();
int launch_slab_sum(const double *slab, int nslab, int C, double count, double *glob, double *local, hipStream_t st);
// training-mode statistics of one BatchNorm layer: reduce, exchange over the ranks (stream-ordered), return the view to finalise from
int stats_exchange(const double *slab, int nslab, int C, double count, hipStream_t st, StatsView *out);
// the same for sums that are already one [2][C] slab in the exchange buffer's global half (fully connected head)
int stats_exchange_inplace(int C, hipStream_t st, StatsView *out);
double *stats_buffer_global();
double *stats_buffer_local();

// part (optional): workspace of pool_fwd_splits(G, K, C) * G * C * 8 bytes; with it, pooling over whole clouds (K >= 512)
// is split over K into partial maxima that a second launch merges (first maximum wins, as in the one-launch form)
int pool_fwd_splits(int G, int K, int C);
int launch_pool_fwd(const float *z, const float *scale, const float *shift, int G, int K, int C, float *out, int32_t *arg,
                    hipStream_t st, float *origin_a = nullptr, float *origin_b = nullptr, int norigin = 0, void *part = nullptr,
                    float *zsel = nullptr);
// backward of max + ReLU without materialising the dense gradient: writes the masked pooled gradient dm (G x C)
// and collects the BatchNorm-backward column sums; consumers rebuild dy on the fly (A_DZ_POOL)
int launch_pool_bwd(const float *dout, const int32_t *arg, const float *z, const float *scale, const float *shift,
                    const float *mean, const float *istd, int G, int K, int C, float *dm, double *slab, int *nslab,
                    hipStream_t st, const float *zsel = nullptr);

int launch_fill_zero(void *p, size_t bytes, hipStream_t st);

}  // namespace pnpp
