/*
 * pnpp_hip.h -- C ABI of libpnpp_hip.so: the MI355X (gfx950) implementation of the
 * PointNet++ set-abstraction + von-Mises-KL training path of
 * 0xPabloxx/3d-pointcloud-orientation-estimation.
 *
 * The reference has no FFI of its own: its boundary is the Python surface listed in
 * SURVEY.md 8(b).  Each entry point below names the reference interface (file:line under
 * /root/reference) whose work it performs; the Python host in
 * 3d-pointcloud-orientation-estimation_amd/ keeps the reference's names on top of these.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every pointer is a DEVICE pointer unless the name ends in _host.
 *   - Tensors are dense row-major float32; indices are int32 on the device
 *     (the Python host converts to the reference's int64 at its own boundary).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, allocates or frees.  Outputs and workspaces are caller-owned.
 *   - Return value: PNPP_OK or a negative pnpp_status; pnpp_last_error() gives the text.
 *     No exception crosses the ABI.  Functions are re-entrant; the only global state is
 *     the thread-local last-error string, an immutable device-property cache and the two
 *     process-wide switches (matmul precision, BatchNorm statistics exchange).
 */
#ifndef PNPP_HIP_H
#define PNPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PNPP_OK = 0,
    PNPP_ERR_ARG = -1,      /* bad shape / null pointer / unsupported size: Python raises ValueError   */
    PNPP_ERR_RANGE = -2,    /* k > N and similar: Python raises RuntimeError like torch.topk           */
    PNPP_ERR_LAUNCH = -3,   /* hipLaunch / hipGetLastError failure: Python raises RuntimeError         */
    PNPP_ERR_WORKSPACE = -4 /* workspace too small                                                       */
} pnpp_status;

const char *pnpp_last_error(void);
int pnpp_abi_version(void);

/* Opt-in per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * pnpp_profile_enable(1) starts recording one event pair around every kernel the library launches (not
 * thread-safe, adds ~2 us per launch: never on while throughput is being timed); pnpp_profile_report
 * synchronises the recorded events and writes one line per distinct launch tag,
 * "<kernel and shape tag>\t<launches>\t<total ms>\n", into buf (truncated to buflen), then clears.
 * Returns the number of distinct tags, or a negative pnpp_status. */
int pnpp_profile_enable(int on);
int pnpp_profile_report(char *buf_host, size_t buflen);

/* ------------------------------------------------------------------------------------------
 * Index primitives
 * ---------------------------------------------------------------------------------------- */

/* models/base.py:20-27  square_distance(src (B,S,3), dst (B,N,3)) -> out (B,S,N).
 * Bit-equal to the ATen CPU evaluation order (fmaf-chained dot, unfused norms). */
int pnpp_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, void *stream);

/* models/base.py:29-35  query_ball_point(new_xyz, xyz, nsample) == kNN.
 * new_xyz (B,S,3), xyz (B,N,3) -> idx (B,S,k) int32, ascending (distance, index).
 * The k smallest by the bit-exact float32 recipe; lowest index wins ties. */
int pnpp_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx, void *stream);

/* PointNet++Demo.py:8-29  farthest_point_sample with the start indices injected.
 * xyz (B,N,3), start (B) -> out (B,npoint) int32. */
int pnpp_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, void *stream);

/* PointNet++Demo.py:49-70  query_ball_point(radius, nsample, xyz, new_xyz) -> idx (B,S,nsample) int32. */
int pnpp_ball_query(const float *new_xyz, const float *xyz, int B, int S, int N, float radius, int nsample,
                    int32_t *idx, void *stream);

/* models/pointnet_pp_8dir.py:28  per-cloud uniform random subset of size npoint out of N, without
 * replacement, in random order -- the device-side replacement of B host `torch.randperm(N)[:npoint]`
 * calls (throughput mode; parity mode replays the CPU generator on the host and passes the indices in).
 * Counter-based: the result is a pure function of (seed, stream_id, b).  out (B,npoint) int32. */
int pnpp_sample_random(uint64_t seed, uint64_t stream_id, int B, int N, int npoint, int32_t *out, void *stream);
/* dataloader_single_peak_vonMises.py:12-14 (and the two other dataloaders): sample_pts = np.random.choice(len, num,
 * replace=len<num) rows of a cloud -- on the device, for a bank of full clouds resident in HBM: bank (n_clouds,Lmax,3),
 * lengths (n_clouds) valid rows per cloud, cloud_ids (B) bank row of every batch slot (NULL: slot b reads cloud b).
 * out (B,num,3): an ordered uniform subset without replacement where lengths >= num, uniform draws with replacement
 * where 0 < lengths < num, zeros for an empty cloud.  Pure function of (seed, stream_id, slot); any Lmax (the LDS key table is sized by num, not by the cloud). */
int pnpp_subsample_points(uint64_t seed, uint64_t stream_id, const float *bank, const int32_t *lengths,
                          const int32_t *cloud_ids, int B, int Lmax, int num, float *out, void *stream);
/* Same, with the stream id read from DEVICE memory at kernel time: stream_id = stream_id_dev[0] + offset, and the
 * kernel post-increments stream_id_dev[0] once every workgroup has read it -- the launch can be captured into a
 * hipGraph and still draws fresh centres on every replay.  stream_id_dev points at TWO uint64 words: the counter and
 * a ticket word that must be zero before the first call (the kernel leaves it zero). */
int pnpp_sample_random_dev(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N, int npoint,
                           int32_t *out, void *stream);
/* Two consecutive pnpp_sample_random_dev draws (the centres of two stacked set-abstraction levels: pointnet_pp_8dir.py:28
 * of sa1 and sa2) in one launch: out1 (B,npoint1) uses the counter's value, out2 (B,npoint2) the next one, the counter
 * advances by two -- bit-identical to the two separate calls. */
int pnpp_sample_random_dev2(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N1, int npoint1, int32_t *out1,
                            int N2, int npoint2, int32_t *out2, void *stream);

/* models/base.py:4-18  index_points(points (B,N,C), idx (B,M)) -> out (B,M,C); idx is int32, flattened
 * over its trailing dims.  _bwd accumulates dpoints (B,N,C) += scatter(dout) deterministically
 * (dpoints must be zero-initialised by the caller when it wants a pure gradient). */
int pnpp_index_points(const float *points, const int32_t *idx, int B, int N, int C, int M, float *out, void *stream);
int pnpp_index_points_bwd(const float *dout, const int32_t *idx, int B, int N, int C, int M, float *dpoints,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * Set abstraction: grouping + 3 x (1x1 conv -> train/eval BatchNorm2d -> ReLU) + max over nsample
 * models/pointnet_pp_8dir.py:21-43 (PointNetSetAbstraction.forward) and its autograd backward.
 * ---------------------------------------------------------------------------------------- */
#define PNPP_MAX_LAYERS 4

typedef struct {
    int B;             /* clouds                                                                  */
    int N;             /* points per cloud going in                                               */
    int S;             /* centres per cloud (npoint); 1 when group_all                            */
    int K;             /* neighbours per centre (nsample); N when group_all                       */
    int D;             /* feature channels of `points` (0 when points is None)                    */
    int L;             /* number of conv/bn layers (3 in every reference model)                   */
    int C[PNPP_MAX_LAYERS]; /* mlp_channels, each a multiple of 32                                */
    int group_all;     /* pointnet_pp_8dir.py:23-26                                               */
    int training;      /* 1: batch statistics + running-stat update; 0: running statistics        */
    float eps;         /* 1e-5                                                                    */
    float momentum;    /* 0.1                                                                     */
} pnpp_sa_desc;

typedef struct {
    const float *xyz;          /* (B,N,3)                                                          */
    const float *points;       /* (B,N,D) or NULL                                                  */
    const int32_t *centre_idx; /* (B,S) centre indices (unused when group_all)                     */
    const int32_t *neighbour_idx; /* optional (B,S,K): use these neighbours instead of running the kNN
                                  (radius grouping of PointNet++Demo.py:49-70, or injection in tests)  */
    const float *conv_w[PNPP_MAX_LAYERS];   /* (C[l], Cin[l]) = state_dict convs.l.weight, xyz columns first */
    const float *conv_b[PNPP_MAX_LAYERS];   /* (C[l])                                               */
    const float *bn_w[PNPP_MAX_LAYERS];     /* gamma                                                */
    const float *bn_b[PNPP_MAX_LAYERS];     /* beta                                                 */
    float *bn_rm[PNPP_MAX_LAYERS];          /* running_mean (updated in training)                   */
    float *bn_rv[PNPP_MAX_LAYERS];          /* running_var  (updated in training)                   */
    int64_t *bn_nbt[PNPP_MAX_LAYERS];       /* num_batches_tracked (+1 in training) or NULL         */
    float *new_xyz;            /* out (B,S,3)                                                      */
    float *out;                /* out (B,S,C[L-1])                                                 */
    void *saved;               /* activations kept for backward, pnpp_sa_saved_bytes()             */
    void *scratch;             /* transient, pnpp_sa_scratch_bytes()                               */
} pnpp_sa_fwd_args;

typedef struct {
    const float *xyz, *points;
    const float *conv_w[PNPP_MAX_LAYERS];
    const float *bn_w[PNPP_MAX_LAYERS];
    const float *bn_b[PNPP_MAX_LAYERS];
    const float *dout;         /* (B,S,C[L-1]) upstream gradient                                   */
    const void *saved;         /* as written by pnpp_sa_forward                                    */
    void *scratch;             /* pnpp_sa_scratch_bytes()                                          */
    float *d_conv_w[PNPP_MAX_LAYERS];       /* out, same shapes as conv_w                           */
    float *d_conv_b[PNPP_MAX_LAYERS];       /* out (written as exact zeros in training: a bias in front
                                               of a train-mode BatchNorm has zero gradient)         */
    float *d_bn_w[PNPP_MAX_LAYERS], *d_bn_b[PNPP_MAX_LAYERS];
    float *dpoints;            /* out (B,N,D) or NULL                                              */
} pnpp_sa_bwd_args;

size_t pnpp_sa_saved_bytes(const pnpp_sa_desc *d);
size_t pnpp_sa_scratch_bytes(const pnpp_sa_desc *d);
int pnpp_sa_forward(const pnpp_sa_desc *d, const pnpp_sa_fwd_args *a, void *stream);
int pnpp_sa_backward(const pnpp_sa_desc *d, const pnpp_sa_bwd_args *a, void *stream);
/* Grouping of two stacked levels ahead of their forward passes, in ONE launch (models/pointnet_pp_8dir.py:28-31 of sa1 and of
 * sa2: index_points + query_ball_point twice).  Level 2 searches among level 1's centres, which are rows of the same cloud, so
 * once both levels' centre indices are drawn neither search waits for anything: d1 / d2 describe the two levels (d2->N == d1->S),
 * centre1 (B,S1) rows of xyz, centre2 (B,S2) positions among level 1's centres.  Writes the neighbour indices and centre
 * coordinates into saved1 / saved2 (sized by pnpp_sa_saved_bytes) and new_xyz1 (B,S1,3) / new_xyz2 (B,S2,3); pnpp_sa_forward then
 * takes them as already grouped when its neighbour_idx argument IS pnpp_sa_saved_neighbours(d, saved) (and new_xyz is that buffer). */
int pnpp_sa_group_pair(const pnpp_sa_desc *d1, const pnpp_sa_desc *d2, const float *xyz, const int32_t *centre1,
                       const int32_t *centre2, void *saved1, float *new_xyz1, void *saved2, float *new_xyz2, void *stream);
/* read-only view of the neighbour indices kept in `saved` ((B,S,K) int32; NULL when group_all) */
const int32_t *pnpp_sa_saved_neighbours(const pnpp_sa_desc *d, const void *saved);
/* read-only view of the max-pool routing kept in `saved`: (B*S, C_last) int32, the position 0..K-1 inside its group of the
 * row that torch.max(x, 3) selected (pointnet_pp_8dir.py:42; first maximum on ties) */
const int32_t *pnpp_sa_saved_argmax(const pnpp_sa_desc *d, const void *saved);
/* diagnostics for parity tests: the ReLU decisions of layer `layer` (0 .. L-1) of this call exactly as its backward pass takes
 * them -- out[M x C_layer] bytes, 1 where F.relu(bn(conv(x))) of pointnet_pp_8dir.py:40-41 lets the gradient pass: the sign of
 * fmaf(Z_l, scale_l, shift_l) on the kept pre-BN activations; layer 0 of a level on raw coordinates keeps no Z_0 and is rebuilt
 * from xyz (B,N,3) and conv_w0 (C_0 x 3) by the same matrix instructions as the kernels (NULL for other levels / layers).
 * A float64 evaluation that is handed these decisions and the max-pool routing above is a smooth function of rounding. */
int pnpp_sa_saved_relu_mask(const pnpp_sa_desc *d, const void *saved, const float *xyz, const float *conv_w0, int layer,
                            uint8_t *out, void *stream);
/* compile-time experiment switches that were on when the library was built (timing experiments that compute WRONG results,
 * in-kernel stamps): 0 in a library that ships -- tests/test_abi.py holds it to 0.  bit 0 PNPP_WS_EXP_NO_MFMA, 1 WSP_EXP,
 * 2 WSX_EXP, 3 WSQ_EXP, 4 WSQ_PLAIN, 5 FCF_EXP, 6 PNPP_STAMPS, 7 WSF_EXP, 8 WD3_PRIO */
unsigned pnpp_build_flags(void);

/* ------------------------------------------------------------------------------------------
 * Fully connected block: y = act(norm(x W^T + b)) with norm in {BatchNorm1d, LayerNorm, none},
 * optional ReLU and inverted dropout by an explicit {0,1} mask.
 * models/pointnet_pp_vonMises.py:32-35, pointnet_pp_8dir.py:81-85, pointnet_pp_mvM.py:82-83.
 * ---------------------------------------------------------------------------------------- */
typedef enum { PNPP_NORM_NONE = 0, PNPP_NORM_BATCH = 1, PNPP_NORM_LAYER = 2 } pnpp_norm;

typedef struct {
    int M, K, N;        /* rows (batch), in features, out features                               */
    int norm;           /* pnpp_norm                                                              */
    int relu;           /* apply ReLU after norm                                                  */
    int training;
    float eps, momentum;
    float drop_scale;   /* 1/(1-p) applied with `mask` / the drawn mask; ignored without dropout   */
} pnpp_fc_desc;

typedef struct {
    const float *x;      /* (M,K)                                                                  */
    const float *w;      /* (N,K)                                                                  */
    const float *b;      /* (N)                                                                    */
    const float *nw, *nb;/* norm affine (N) or NULL                                                */
    float *rm, *rv;      /* BatchNorm running stats or NULL                                        */
    int64_t *nbt;        /* BatchNorm num_batches_tracked (+1 in training) or NULL                 */
    const uint8_t *mask; /* (M,N) dropout keep-mask or NULL                                        */
    uint8_t *mask_out;   /* or: (M,N) keep-mask DRAWN by the kernel (training, BatchNorm, M <= 32):
                            Bernoulli(1 - drop_p) from Philox keyed by rng_seed, stream id rng_counter[0];
                            the kernel post-increments rng_counter[0] (rng_counter[1] is its ticket word,
                            zero on entry and exit); pass it as `mask` to pnpp_fc_backward.  NULL: unused */
    float drop_p;
    uint64_t rng_seed;
    uint64_t *rng_counter;
    float *y;            /* out (M,N)                                                              */
    void *saved;         /* pnpp_fc_saved_bytes()                                                  */
    void *scratch;       /* pnpp_fc_scratch_bytes()                                                */
} pnpp_fc_fwd_args;

typedef struct {
    const float *x, *w, *b, *nw, *nb;
    const uint8_t *mask;
    const float *dy;     /* (M,N)                                                                  */
    const void *saved;
    void *scratch;
    float *dx;           /* (M,K) or NULL                                                          */
    float *dw, *db;      /* (N,K), (N)                                                             */
    float *dnw, *dnb;    /* (N) or NULL                                                            */
} pnpp_fc_bwd_args;

size_t pnpp_fc_saved_bytes(const pnpp_fc_desc *d);
size_t pnpp_fc_scratch_bytes(const pnpp_fc_desc *d);
int pnpp_fc_forward(const pnpp_fc_desc *d, const pnpp_fc_fwd_args *a, void *stream);
int pnpp_fc_backward(const pnpp_fc_desc *d, const pnpp_fc_bwd_args *a, void *stream);

/* ------------------------------------------------------------------------------------------
 * Output heads and losses (forward value and analytic gradient in one launch)
 * ---------------------------------------------------------------------------------------- */

/* pointnet_pp_vonMises.py:36-37 + train_single_peak_vonMises_KL.py:23-28,83:
 * o (B,2) raw fc3 output -> mu = tanh(o0)*pi, kappa = softplus(o1);
 * loss_vec[b] = KL(vM(mu,kappa) || vM(mu_gt,kappa_gt)) by the reference formula (kappa_p <= 1e-6 branch
 * included, log I0 evaluated as kappa + log(i0e) so kappa >= 89 stays finite where the reference is NaN).
 * d_o (B,2) = d loss_vec[b] / d o[b,:]  (the host scales by the upstream gradient, 1/B for .mean()). */
int pnpp_vm_head_kl(const float *o, const float *mu_gt, const float *kappa_gt, int B, float *mu, float *kappa,
                    float *loss_vec, float *d_o, void *stream);
/* The whole tail of the single-peak step in one launch: o = x W^T + b (the model's fc3: pointnet_pp_vonMises.py:35,
 * W is (2,K)), head activations, KL, .mean() (train_single_peak_vonMises_KL.py:82-84) and their backward:
 * loss_mean[0], dw (2,K), db (2), dx (B,K; may be NULL).  Gradients are written, not accumulated. */
int pnpp_vm_fc_head_kl_step(const float *x, const float *w, const float *b, const float *mu_gt, const float *kappa_gt, int B,
                            int K, float *loss_mean, float *dw, float *db, float *dx, void *stream);
/* The same launch, carrying the NEXT step's centre draw in its idle CUs: workgroup 0 is pnpp_vm_fc_head_kl_step, workgroups 1 .. 2*Bs
 * are pnpp_sample_random_dev2(seed, stream_id_dev, offset, Bs, N1, npoint1, out1, N2, npoint2, out2) -- the draws of
 * models/pointnet_pp_8dir.py:28 for sa1 and sa2 depend on nothing but their counter, so a training loop that keeps (out1, out2)
 * as the centres of its next forward pass draws exactly the sequence it would draw at the start of every step, one launch fewer. */
int pnpp_vm_fc_head_kl_step_sample(const float *x, const float *w, const float *b, const float *mu_gt, const float *kappa_gt, int B,
                                   int K, float *loss_mean, float *dw, float *db, float *dx, uint64_t seed, uint64_t *stream_id_dev,
                                   uint64_t offset, int Bs, int N1, int npoint1, int32_t *out1, int N2, int npoint2, int32_t *out2,
                                   void *stream);
/* The same, followed by the batch mean of train_single_peak_vonMises_KL.py:83 (`loss = kl_von_mises(...).mean()`),
 * in one single-workgroup launch: loss_mean[0] = mean_b loss_vec[b] (fixed-order fp64 sum),
 * d_o_mean (B,2) = d loss_mean / d o.  mu, kappa and loss_vec may be NULL. */
int pnpp_vm_head_kl_mean(const float *o, const float *mu_gt, const float *kappa_gt, int B, float *mu, float *kappa,
                         float *loss_vec, float *loss_mean, float *d_o_mean, void *stream);
/* backward of the activation alone: d_o[b] = (dmu[b]*pi*(1-tanh^2 o0), dkappa[b]*sigmoid(o1)). */
int pnpp_vm_head_bwd(const float *o, const float *dmu, const float *dkappa, int B, float *d_o, void *stream);

/* train_single_peak_vonMises_KL.py:23-28 on its own: values and d/dmu_p, d/dkappa_p. */
int pnpp_vm_kl_single(const float *mu_p, const float *kappa_p, const float *mu_q, const float *kappa_q, int n,
                      float *kl, float *dmu, float *dkappa, void *stream);

/* train_multi_peaks_vonMises_KL.py:38-81 match_loss: per sample K x K cost of the clamped / wrapped KL,
 * nan_to_num(1e6), optimal assignment (exhaustive over <= 8! permutations, first optimum in lexicographic
 * order), loss_b = sum w_i c_i / (sum w_i + 1e-8).  mu, kappa, w (B,maxK); vm_gt (B,maxK,3); K_gt (B) int32.
 * Outputs loss_vec (B), gradients (B,maxK) each, assignment (B,maxK) int32 (-1 beyond K). */
int pnpp_vm_match_loss(const float *mu, const float *kappa, const float *w, const float *vm_gt, const int32_t *K_gt,
                       int B, int maxK, float *loss_vec, float *dmu, float *dkappa, float *dw, int32_t *assign,
                       void *stream);

/* The whole tail of the multi-peak training step in ONE launch: the three output heads o = x [W_pi; W_mu; W_kappa]^T + b
 * (models/pointnet_pp_mvM.py:91-96), the head activations (:91-125), match_loss (train_multi_peaks_vonMises_KL.py:54-81), its batch
 * mean (:229) and their backward.  x (B,K) features; w_pi (max_K,K), w_mu (2 max_K,K), w_kappa (max_K,K); max_K 4 or 8.
 * Writes loss_mean (1), the heads' weight / bias gradients, dx (B,K; may be NULL) and optionally mu, kappa, weight (B,max_K).
 * Bs > 0: the next step's centre draw rides in the same launch (arguments as pnpp_sample_random_dev2); Bs = 0: none. */
int pnpp_mvm_fc_head_match_step(const float *x, const float *w_pi, const float *b_pi, const float *w_mu, const float *b_mu,
                                const float *w_kappa, const float *b_kappa, const float *vm_gt, const int32_t *K_gt, int B, int K,
                                int maxK, float temp, float kappa_max, float *loss_mean, float *dw_pi, float *db_pi, float *dw_mu,
                                float *db_mu, float *dw_kappa, float *db_kappa, float *dx, float *mu, float *kappa, float *weight,
                                uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int Bs, int N1, int npoint1, int32_t *out1,
                                int N2, int npoint2, int32_t *out2, void *stream);
/* pointnet_pp_mvM.py:91-125: raw head outputs -> (mu, kappa, weight) and the backward of that map.
 * pi_raw (B,K), mu_raw (B,2K), kappa_raw (B,K). */
int pnpp_mvm_head(const float *pi_raw, const float *mu_raw, const float *kappa_raw, int B, int K, float temp,
                  float kappa_max, float *mu, float *kappa, float *weight, void *stream);
int pnpp_mvm_head_bwd(const float *pi_raw, const float *mu_raw, const float *kappa_raw, const float *weight,
                      const float *dmu, const float *dkappa, const float *dweight, int B, int K, float temp,
                      float kappa_max, float *dpi_raw, float *dmu_raw, float *dkappa_raw, void *stream);

/* train_8dir_KL.py:60-68: loss_vec[b] = -sum p * log_softmax(logits); dlogits = softmax*sum(p) - p. */
int pnpp_soft_ce(const float *logits, const float *p, int B, int C, float *loss_vec, float *dlogits, void *stream);

/* Heads and losses of the other set-abstraction models.
 * models/pointnet_pp_Fwd.py:98, models/Pointnet_pp_xyz.py:84-85, models/Pointnet_pp_xyz_Schedmit.py:87-88:
 * F.normalize(x, p=2, dim=1, eps): y[m,:] = x[m,:] / max(||x[m,:]||, eps); x, y (M,C), C <= 64. */
int pnpp_l2_normalize(const float *x, int M, int C, float eps, float *y, void *stream);
int pnpp_l2_normalize_bwd(const float *x, const float *dy, int M, int C, float eps, float *dx, void *stream);
/* nn.MSELoss() (train.py:168,183; train_multi_8dir.py:80,100; train_8dir.py:53,67): loss[0] = mean_i (p_i - t_i)^2 over
 * all n elements, dp (optional) = 2 (p - t) / n.  One workgroup, fixed-order float64 sum. */
int pnpp_mse(const float *p, const float *t, size_t n, float *loss, float *dp, void *stream);
/* simple_pointnet_train.py:153,174,182: the per-sample form, loss_vec[b] = mean_c (p[b,c] - t[b,c])^2 (its mean over b is
 * nn.MSELoss() of the batch); dp (optional) [b,c] = 2 (p - t) / C. */
int pnpp_mse_rows(const float *p, const float *t, int B, int C, float *loss_vec, float *dp, void *stream);
/* train.py:184-185: loss[0] = mean_b (a_b . b_b)^2 for two (B,C) sets of axes; da, db optional (both or neither). */
int pnpp_orth_loss(const float *a, const float *b, int B, int C, float *loss, float *da, float *db, void *stream);
/* train_multi_8dir.py:41-44 proj_probs: v = normalize(vec (B,3)); sims = clamp(v dirs^T, min=0) with dirs (D,3), D <= 16;
 * probs (B,D) = sims / clamp(sum_j sims, min=1e-8). */
int pnpp_proj_probs(const float *vec, const float *dirs, int B, int D, float *probs, void *stream);
int pnpp_proj_probs_bwd(const float *vec, const float *dirs, const float *dprobs, int B, int D, float *dvec, void *stream);

/* Point-transformer configuration (models/point_transformer.py:4-20), forward pass.
 * input_proj (nn.Linear with at most 8 inputs): y (M,N) = x (M,K) w^T (N,K) + b. */
int pnpp_linear_smallk(const float *x, const float *w, const float *b, int M, int K, int N, float *y, void *stream);
/* nn.MultiheadAttention core (torch.nn.functional.multi_head_attention_forward between in_proj and out_proj):
 * qkv (B,N,3E) with the in_proj bias added, E = H*head_dim, head h = columns h*head_dim.. of each third;
 * out (B,N,E) = concat_h dropout(softmax(q_h k_h^T / sqrt(head_dim))) v_h; lse (B,H,N) optional log-sum-exp of the
 * scaled scores.  The N x N matrix is never materialised.  head_dim must be 16.  N is the number of ROWS per cloud in
 * qkv / out and must be a multiple of 128; the first n_valid of them are points, the rest padding supplied by the caller:
 * keys >= n_valid get no attention weight, outputs of queries >= n_valid are unspecified (finite) and must be ignored
 * (the reference takes any N: models/point_transformer.py:15-20; the drop-in module pads to the next multiple of 128).
 * Dropout on the attention weights (train mode, nn.MultiheadAttention(dropout=p)): mask = bit-packed keep bits from
 * pnpp_attention_dropout_mask (kept weights are scaled by 1/(1-p)); mask == NULL means no dropout (p is then ignored). */
int pnpp_attention_fwd(const float *qkv, int B, int N, int n_valid, int H, int head_dim, const uint32_t *mask, float p,
                       float *out, float *lse, void *stream);
/* Keep bits, Bernoulli(1-p), a pure function of (seed, stream_id, element): mask (B,H,N,N/32) holds for every query the
 * bits of its keys (bit = key % 32 of word key / 32); maskT (B,H,N,N/32) is the transpose (for every key the bits of the
 * queries), read by the dK/dV kernel. */
int pnpp_attention_dropout_mask(uint64_t seed, uint64_t stream_id, int B, int N, int H, float p, uint32_t *mask,
                                uint32_t *maskT, void *stream);
/* The same with the stream id kept in device memory (stream_id_dev[0] = calls so far, stream_id_dev[1] = ticket word, both
 * zero-initialised by the caller): stream id = offset + stream_id_dev[0], read by the launch, which then adds 1 -- a
 * train step captured in a hipGraph draws fresh masks on every replay (nn.Dropout semantics), reproducibly. */
int pnpp_attention_dropout_mask_dev(uint64_t seed, uint64_t *stream_id_dev, uint64_t offset, int B, int N, int H, float p,
                                    uint32_t *mask, uint32_t *maskT, void *stream);
/* Post-norm residual block: y (M,E) = LayerNorm(x + r) * w + b over the last dimension (r may be NULL), E <= 128. */
int pnpp_add_layernorm(const float *x, const float *r, const float *w, const float *b, int M, int E, float eps, float *y,
                       void *stream);
/* x.mean(dim=1): y (B,E) = mean over the N points of x (B,N,E). */
int pnpp_mean_points(const float *x, int B, int N, int E, float *y, void *stream);
/* Backward passes of the four entries above (what torch.autograd derives for the reference model).
 * linear_smallk: dw (N,K) = dy^T x, db (N) = column sums of dy (db may be NULL); x is data, no dx. */
size_t pnpp_linear_smallk_bwd_scratch_bytes(int M, int N);
int pnpp_linear_smallk_bwd(const float *x, const float *dy, int M, int K, int N, float *dw, float *db, void *scratch,
                           void *stream);
/* attention: dqkv (B,N,3E) from qkv, the forward's out and lse, and d_out (B,N,E); dsum (B,H,N) is scratch
 * (rowsum(d_out * out) per head).  Scores are recomputed, never stored.  mask / maskT / p as in the forward (both NULL:
 * no dropout).  n_valid as in the forward: d_out rows >= n_valid must be zero; dK / dV rows >= n_valid are written as zeros. */
int pnpp_attention_bwd(const float *qkv, const float *out, const float *d_out, const float *lse, int B, int N, int n_valid, int H,
                       int head_dim, const uint32_t *mask, const uint32_t *maskT, float p, float *dqkv, float *dsum,
                       void *stream);
/* add_layernorm: du (M,E) = gradient w.r.t. both x and r; dwb (2,E) = (d weight, d bias). */
size_t pnpp_add_layernorm_bwd_scratch_bytes(int M, int E);
int pnpp_add_layernorm_bwd(const float *x, const float *r, const float *w, const float *dy, int M, int E, float eps,
                           float *du, float *dwb, void *scratch, void *stream);
int pnpp_mean_points_bwd(const float *dy, int B, int N, int E, float *dx, void *stream);

/* ------------------------------------------------------------------------------------------
 * Throughput mode of the grouped layers (BASELINE.json configs[1] names a bf16 mode; the reference itself is float32,
 * models/pointnet_pp_8dir.py:40-42): bf16_operands != 0 makes the large 1x1-conv GEMMs of pnpp_sa_forward / pnpp_sa_backward
 * round their two MFMA operands to bfloat16 (float32 accumulate; activations, statistics, transforms stay float32 / float64).
 * Process-wide, default 0 (exact float32), initial value from the environment variable PNPP_MATMUL=bf16.
 * ---------------------------------------------------------------------------------------- */
int pnpp_set_matmul_precision(int bf16_operands);
int pnpp_get_matmul_precision(void);

/* ------------------------------------------------------------------------------------------
 * How the float32 products of the grouped layers' large GEMMs are formed (ABI 5).  0: v_mfma_f32_32x32x2_f32.  1: on the bf16
 * matrix pipe from EXACT three-way operand splits (a float32 is the sum of three bfloat16 numbers; six bf16 x bf16 products per
 * float32 product, each exact in float32, float32 accumulation; the three products dropped are below 2^-25 of the product, i.e.
 * below the rounding of the accumulation): float32 results to float32 rounding -- held to the same parity gates as mode 0 --
 * at 2.67 x the float32 MFMA rate (csrc/gemm_wsf3_kernels.hip).  Not a reduced-precision mode and unrelated to
 * pnpp_set_matmul_precision (which rounds operands to ONE bf16).  Process-wide; initial value from PNPP_SPLIT_PRODUCTS.
 * ---------------------------------------------------------------------------------------- */
int pnpp_set_split_products(int on);
int pnpp_get_split_products(void);
/* diagnostics: non-zero once a bounded LDS poll of the wave-pair kernel (csrc/gemm_wsd3_kernels.hip) has given up in this process */
int pnpp_debug_wsd3_timeouts(void);

/* ------------------------------------------------------------------------------------------
 * SyncBN (ABI 3; off by default): cross-rank exchange of the train-mode BatchNorm sums under data parallelism, so that a
 * global batch split over ranks normalises exactly like the reference's single process on the concatenated batch
 * (nn.BatchNorm2d / nn.BatchNorm1d in training mode: models/pointnet_pp_8dir.py:40-41, models/pointnet_pp_vonMises.py:32-33).
 * Once a callback is registered, every training-mode BatchNorm of pnpp_sa_forward / pnpp_sa_backward / pnpp_fc_forward /
 * pnpp_fc_backward reduces this rank's per-channel sums -- forward (sum z, sum z^2, rows), backward (sum dy, sum dy*xhat, rows) --
 * into buf[0 .. 2C], calls fn(buf, 2C + 1, stream, user), which must replace those doubles by their SUM OVER ALL RANKS,
 * stream-ordered behind `stream` (RCCL: an all-reduce enqueued on it; gloo: a blocking all-reduce), and finalises from the
 * summed values.  Parameter gradients of the BatchNorm layers stay rank-local sums (the gradient all-reduce adds them up like
 * every other parameter's); running statistics become the global ones on every rank.  buf: caller-owned device memory of
 * buf_doubles >= 2 * (2 * C_max + 1) doubles (second half: the rank's own sums).  fn == NULL unregisters.
 * The head's fused BatchNorm epilogue and its in-kernel dropout draw need the whole batch in one tile and are bypassed while a
 * callback is registered (pass an explicit mask).  Process-wide, like the matmul precision.
 * ---------------------------------------------------------------------------------------- */
typedef int (*pnpp_stats_exchange_fn)(double *buf, size_t n_doubles, void *stream, void *user);
int pnpp_set_stats_exchange(pnpp_stats_exchange_fn fn, void *user, double *buf, size_t buf_doubles);
int pnpp_stats_exchange_enabled(void);

/* ------------------------------------------------------------------------------------------
 * Step glue on one flat parameter / gradient buffer
 * (train_single_peak_vonMises_KL.py:80,85; train_multi_peaks_vonMises_KL.py:221,235-236)
 * ---------------------------------------------------------------------------------------- */
/* torch.optim.Adam(lr, betas, eps, weight_decay=0) single fused update; step is 1-based. */
int pnpp_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                   float beta1, float beta2, float eps, float grad_scale, void *stream);
/* The same update, clearing each gradient once it has been consumed: the opt.zero_grad() of the next iteration
 * (train_single_peak_vonMises_KL.py:80) folded into the optimiser launch. */
int pnpp_adam_step_zero(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                        float beta1, float beta2, float eps, float grad_scale, void *stream);

/* The same update for launches captured in a hipGraph: the step count lives in device memory (step_state[0] = steps
 * taken so far, step_state[1] = ticket word, both zero-initialised by the caller) and is advanced by the launch
 * itself, so every replay applies its own bias correction.  zero_grad != 0 clears each gradient after it has been
 * consumed -- the opt.zero_grad() of the next iteration (train_single_peak_vonMises_KL.py:80), folded in. */
int pnpp_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, uint64_t *step_state,
                       float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad, void *stream);
/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) (train_multi_peaks_vonMises_KL.py:235) folded into the update with
 * no host round trip: grad_sumsq[0] (device memory, written by pnpp_sumsq on the same stream) is the sum of squares of the
 * flat gradient buffer as it stands; the gradient is grad * grad_scale, its norm sqrt(grad_sumsq[0]) * grad_scale, and the
 * update uses grad * grad_scale * min(1, max_norm / (norm + 1e-6)).  zero_grad != 0 clears each gradient once consumed. */
int pnpp_adam_step_clip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, int step, float lr,
                        float beta1, float beta2, float eps, float grad_scale, const double *grad_sumsq, float max_norm,
                        int zero_grad, void *stream);
int pnpp_adam_step_dev_clip(float *param, float *grad, float *exp_avg, float *exp_avg_sq, size_t n, uint64_t *step_state,
                            float lr, float beta1, float beta2, float eps, float grad_scale, const double *grad_sumsq,
                            float max_norm, int zero_grad, void *stream);
/* sum of squares of a flat buffer into out[0] (double), deterministic two-level reduction. */
int pnpp_sumsq(const float *x, size_t n, double *out, void *scratch, size_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PNPP_HIP_H */
