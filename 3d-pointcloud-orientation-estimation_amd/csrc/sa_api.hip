// sa_api.hip -- host-side orchestration of PointNetSetAbstraction forward / backward and of the
// fully connected head blocks, on top of the fused kernels (gemm_kernels.hip, index_kernels.hip).
//
// Reference: models/pointnet_pp_8dir.py:21-43 (forward), its autograd graph and the
// heads models/pointnet_pp_vonMises.py:32-35 / pointnet_pp_mvM.py:82-83.
//
// Everything is stream-ordered: these functions only enqueue work on `stream`; they never allocate,
// free, copy to the host or synchronise (they can be captured into a hipGraph).
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace pnpp {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- opt-in launch timing ----------------------------------------------------------------------
struct ProfRec {
    std::string tag;
    hipEvent_t a, b;
};
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::string g_open_tag;   // tag of the open ProfScope; every launch inside it gets its own record
static bool g_scope_open = false;

bool prof_on() { return g_prof; }
void prof_begin(hipStream_t, const char *fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_open_tag = buf;
    g_scope_open = true;
}
void prof_end(hipStream_t) { g_scope_open = false; }
bool prof_take_events(hipEvent_t *a, hipEvent_t *b) {
    if (!g_prof || !g_scope_open) return false;   // a launch outside any scope is not timed
    ProfRec r;
    r.tag = g_open_tag;
    if (hipEventCreate(&r.a) != hipSuccess) return false;
    if (hipEventCreate(&r.b) != hipSuccess) {
        (void)hipEventDestroy(r.a);
        return false;
    }
    g_recs.push_back(r);
    *a = r.a, *b = r.b;
    return true;
}

// ---- SyncBN: optional cross-rank exchange of the BatchNorm sums (off by default) ----------------------------
// The callback sums a device buffer of doubles over the ranks, stream-ordered (RCCL: an all-reduce enqueued behind `stream`).
// The buffer is the caller's (this library never allocates): [0, half) is exchanged, [half, 2 half) keeps this rank's sums.
struct StatsExchange {
    pnpp_stats_exchange_fn fn = nullptr;
    void *user = nullptr;
    double *buf = nullptr;
    size_t half = 0;   // doubles per half
};
static StatsExchange g_sx;

bool stats_sync_on() { return g_sx.fn != nullptr; }
double *stats_buffer_global() { return g_sx.buf; }
double *stats_buffer_local() { return g_sx.buf + g_sx.half; }

int stats_exchange_inplace(int C, hipStream_t st, StatsView *out) {
    PNPP_REQUIRE((size_t)(2 * C + 1) <= g_sx.half, PNPP_ERR_ARG, "stats exchange: buffer of %zu doubles per half is too small for C=%d",
                 g_sx.half, C);
    const int rc = g_sx.fn(g_sx.buf, (size_t)(2 * C + 1), (void *)st, g_sx.user);
    PNPP_REQUIRE(rc == 0, PNPP_ERR_LAUNCH, "stats exchange: the registered callback returned %d", rc);
    out->slab = g_sx.buf, out->nslab = 1, out->count_dev = g_sx.buf + 2 * C, out->local = g_sx.buf + g_sx.half;
    return PNPP_OK;
}

int stats_exchange(const double *slab, int nslab, int C, double count, hipStream_t st, StatsView *out) {
    if (!g_sx.fn) {
        out->slab = slab, out->nslab = nslab, out->count_dev = nullptr, out->local = nullptr;
        return PNPP_OK;
    }
    PNPP_REQUIRE((size_t)(2 * C + 1) <= g_sx.half, PNPP_ERR_ARG, "stats exchange: buffer of %zu doubles per half is too small for C=%d",
                 g_sx.half, C);
    int rc = launch_slab_sum(slab, nslab, C, count, g_sx.buf, g_sx.buf + g_sx.half, st);
    if (rc != PNPP_OK) return rc;
    return stats_exchange_inplace(C, st, out);
}

#define PNPP_TRY(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != PNPP_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// set abstraction
// ---------------------------------------------------------------------------------------------
struct SaGeom {
    int M, G, Kd0, maxC, L;
    int Cin[PNPP_MAX_LAYERS];  // reduction dim of layer l in state_dict terms (D+3 for l = 0)
    int Kd[PNPP_MAX_LAYERS];   // padded reduction dim used by the kernels
};

static int sa_geom(const pnpp_sa_desc *d, SaGeom *g) {
    PNPP_REQUIRE(d, PNPP_ERR_ARG, "sa: null descriptor");
    PNPP_REQUIRE(d->B > 0 && d->N > 0 && d->S > 0 && d->K > 0 && d->D >= 0, PNPP_ERR_ARG,
                 "sa: bad geometry B=%d N=%d S=%d K=%d D=%d", d->B, d->N, d->S, d->K, d->D);
    PNPP_REQUIRE(d->L >= 1 && d->L <= PNPP_MAX_LAYERS, PNPP_ERR_ARG, "sa: unsupported layer count %d", d->L);
    if (d->group_all) PNPP_REQUIRE(d->S == 1 && d->K == d->N, PNPP_ERR_ARG, "sa: group_all needs S == 1 and K == N");
    PNPP_REQUIRE((long long)d->B * d->S * d->K < (1ll << 31), PNPP_ERR_ARG, "sa: B*S*K overflows int32");
    g->L = d->L;
    g->M = d->B * d->S * d->K;
    g->G = d->B * d->S;
    g->Kd0 = (d->D + 3 + 3) & ~3;
    g->maxC = 0;
    for (int l = 0; l < d->L; ++l) {
        PNPP_REQUIRE(d->C[l] > 0 && d->C[l] % 32 == 0, PNPP_ERR_ARG,
                     "sa: mlp channel %d (=%d) must be a positive multiple of 32 for the MFMA path", l, d->C[l]);
        g->maxC = d->C[l] > g->maxC ? d->C[l] : g->maxC;
        g->Cin[l] = l == 0 ? d->D + 3 : d->C[l - 1];
        g->Kd[l] = l == 0 ? g->Kd0 : d->C[l - 1];
    }
    return PNPP_OK;
}

struct SaSaved {
    int32_t *idx;
    float *new_xyz;
    float *z[PNPP_MAX_LAYERS];
    float *mean[PNPP_MAX_LAYERS], *istd[PNPP_MAX_LAYERS], *scale[PNPP_MAX_LAYERS], *shift[PNPP_MAX_LAYERS];
    int32_t *arg;
    float *zmax;   // (G, C_last): the pre-BN value each pooled output came from (the backward pass's ReLU gate and xhat need it)
    size_t bytes;
};

// whole-cloud pooling in K chunks (pool_fwd_split / merge) does not produce the selected pre-BN values; backward gathers them there
static bool sa_keeps_zmax(const pnpp_sa_desc *d, const SaGeom &g) { return pool_fwd_splits(g.G, d->K, d->C[d->L - 1]) <= 1; }

static SaSaved sa_saved_layout(const pnpp_sa_desc *d, const SaGeom &g, void *base) {
    Carver cv(base);
    SaSaved s;
    s.idx = cv.take<int32_t>(d->group_all ? 0 : (size_t)g.M);
    s.new_xyz = cv.take<float>((size_t)g.G * 3);
    for (int l = 0; l < d->L; ++l) {
        s.z[l] = cv.take<float>((size_t)g.M * d->C[l]);
        s.mean[l] = cv.take<float>(d->C[l]);
        s.istd[l] = cv.take<float>(d->C[l]);
        s.scale[l] = cv.take<float>(d->C[l]);
        s.shift[l] = cv.take<float>(d->C[l]);
    }
    s.arg = cv.take<int32_t>((size_t)g.G * d->C[d->L - 1]);
    s.zmax = cv.take<float>(sa_keeps_zmax(d, g) ? (size_t)g.G * d->C[d->L - 1] : 0);
    s.bytes = cv.bytes();
    return s;
}

// A/B switch (PNPP_NO_POOL_FUSION=1: pooling stays a pass of its own over Z)
static bool pool_fused_on() {
    static int cached = -1;
    if (cached < 0) {
        const char *v = getenv("PNPP_NO_POOL_FUSION");
        cached = (v && atoi(v) != 0) ? 0 : 1;
    }
    return cached != 0;
}

constexpr int kSmallM = 4096;  // at or below this many rows dZ is materialised once per layer (group_all layers)

struct SaScratch {
    double *slab;
    float *dy[2];
    float *dzbuf;  // small-M layers only: materialised dZ
    float *dm;
    float *cst;
    float *dwslab;
    float *src;  // convolve-then-gather layer 0: one C_0-wide row per source point (P forward, G backward)
    float *xslab;  // ... and the [C_0][4] dW_xyz partials of its backward scatter
    double *mom;   // a level on raw coordinates: moment partials of the relative coordinates (gemm_wsx_kernels.hip)
    size_t bytes;
};

// layer 0 is convolved on the B*N source points and gathered afterwards when the level groups points WITH features
static bool sa_delayed(const pnpp_sa_desc *d) {
    return !d->group_all && d->D > 0 && d->D % 4 == 0 && delayed_layer0_ok(d->C[0]) && d->S * d->K > d->N;
}

static SaScratch sa_scratch_layout(const pnpp_sa_desc *d, const SaGeom &g, void *base) {
    Carver cv(base);
    SaScratch s;
    {   // BatchNorm partial sums of one layer; a level on raw coordinates also parks layer 0's backward sums here (gemm_wsx_kernels.hip)
        size_t nd = (size_t)kMaxStatBlocks * 2 * g.maxC;
        if (!d->group_all && d->D == 0 && wsx_stat_doubles(g.M) > nd) nd = wsx_stat_doubles(g.M);
        s.slab = cv.take<double>(nd);
    }
    const int wide = g.maxC > d->D ? g.maxC : d->D;
    s.dy[0] = cv.take<float>((size_t)g.M * wide);
    s.dy[1] = cv.take<float>((size_t)g.M * wide);
    s.dzbuf = cv.take<float>(g.M <= kSmallM ? (size_t)g.M * g.maxC : 0);
    s.dm = cv.take<float>((size_t)g.G * d->C[d->L - 1]);
    s.cst = cv.take<float>((size_t)5 * g.maxC);
    size_t dwmax = 0;
    for (int l = 0; l < d->L; ++l) {
        int nsplit, kp_pad;
        dw_plan(g.M, d->C[l], g.Cin[l], &nsplit, &kp_pad);
        size_t need = (size_t)nsplit * d->C[l] * kp_pad;
        // the fused dA+dW kernels (large M only) write one C_l x C_{l-1} partial per worker, at most kMaxStatBlocks of them
        const size_t fused = (l > 0 && g.M >= 8192) ? (size_t)kMaxStatBlocks * d->C[l] * d->C[l - 1] : 0;
        need = fused > need ? fused : need;
        dwmax = need > dwmax ? need : dwmax;
    }
    if (sa_delayed(d)) {  // the per-source-point dW_f partials
        int nsplit, kp_pad;
        dw_plan(d->B * d->N, d->C[0], d->D, &nsplit, &kp_pad);
        const size_t a = (size_t)nsplit * d->C[0] * kp_pad;
        dwmax = a > dwmax ? a : dwmax;
    }
    s.dwslab = cv.take<float>(dwmax);
    s.src = cv.take<float>(sa_delayed(d) ? (size_t)d->B * d->N * d->C[0] : 0);
    s.xslab = cv.take<float>(sa_delayed(d) ? (size_t)scatter_dz_splits(d->B * d->N) * d->C[0] * 4 : 0);
    s.mom = cv.take<double>((!d->group_all && d->D == 0) ? xyz0_moment_doubles() : 0);
    s.bytes = cv.bytes();
    return s;
}

static AOperand layer0_operand(const pnpp_sa_desc *d, const float *xyz, const float *points, const SaSaved &sv) {
    AOperand A;
    A.mode = d->group_all ? A_CONCAT : A_GATHER;
    A.a = points;
    A.xyz = xyz;
    A.new_xyz = sv.new_xyz;
    A.idx = sv.idx;
    A.D = d->D;
    A.N = d->N;
    A.S = d->S;
    A.K = d->K;
    return A;
}

static int sa_forward_impl(const pnpp_sa_desc *d, const pnpp_sa_fwd_args *a, hipStream_t st) {
    SaGeom g;
    PNPP_TRY(sa_geom(d, &g));
    PNPP_REQUIRE(a && a->xyz && a->new_xyz && a->out && a->saved && a->scratch, PNPP_ERR_ARG, "sa_forward: null pointer");
    PNPP_REQUIRE(d->D == 0 || a->points, PNPP_ERR_ARG, "sa_forward: D=%d but points is null", d->D);
    PNPP_REQUIRE(d->group_all || a->centre_idx, PNPP_ERR_ARG, "sa_forward: centre_idx is null");
    for (int l = 0; l < d->L; ++l)
        PNPP_REQUIRE(a->conv_w[l] && a->conv_b[l] && a->bn_w[l] && a->bn_b[l] && a->bn_rm[l] && a->bn_rv[l], PNPP_ERR_ARG,
                     "sa_forward: null parameter pointer in layer %d", l);
    const SaSaved sv = sa_saved_layout(d, g, a->saved);
    const SaScratch sc = sa_scratch_layout(d, g, a->scratch);

    // 1. centres and neighbours (pointnet_pp_8dir.py:23-31)
    if (d->group_all) {
        // the centres are the origin: written by the pooling launch that ends this forward (nothing in between reads them)
    } else {
        PNPP_REQUIRE(d->S <= d->N, PNPP_ERR_RANGE, "sa_forward: npoint=%d > N=%d", d->S, d->N);
        if (a->neighbour_idx == sv.idx) {
            // grouped ahead of time by pnpp_sa_group_pair: neighbours, saved centres and new_xyz are already in place
        } else if (a->neighbour_idx) {
            PNPP_TRY(launch_gather_centres(a->xyz, a->centre_idx, d->B, d->N, d->S, a->new_xyz, sv.new_xyz, st));
            hipError_t e = hipMemcpyAsync(sv.idx, a->neighbour_idx, (size_t)g.M * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
            PNPP_REQUIRE(e == hipSuccess, PNPP_ERR_LAUNCH, "sa_forward: neighbour copy failed: %s", hipGetErrorString(e));
        } else {  // the neighbour search gathers its own queries and writes the centre coordinates on the way
            PNPP_TRY(launch_knn_centres(a->xyz, a->centre_idx, d->B, d->S, d->N, d->K, sv.idx, a->new_xyz, sv.new_xyz, st));
        }
    }

    // 2. conv -> BN -> ReLU chain; BN apply + ReLU of layer l-1 happen inside layer l's operand loader
    bool pooled = false;
    // a level on raw coordinates (SA1): layer 0 never exists as a tensor -- its statistics come from the moments of the relative
    // coordinates, layer 1's product builds its operand from the coordinates (gemm_wsx_kernels.hip); the backward pass does the same
    const bool xyz0 = xyz0_applies(g.M, d->D, d->K, d->group_all, d->L, d->C);
    int nmom = 0;
    for (int l = 0; l < d->L; ++l) {
        if (xyz0 && l == 0) {
            if (d->training) {
                PNPP_TRY(launch_rel_moments(layer0_operand(d, a->xyz, a->points, sv), g.M, sc.mom, &nmom, st));
            } else {
                PNPP_TRY(launch_bn_finalize_fwd(nullptr, 0, d->C[0], (double)g.M, a->conv_b[0], a->bn_w[0], a->bn_b[0], a->bn_rm[0],
                                                a->bn_rv[0], nullptr, d->momentum, d->eps, 0, sv.mean[0], sv.istd[0], sv.scale[0],
                                                sv.shift[0], st));
            }
            continue;
        }
        // layer 1 of such a level: the product that also finishes layer 0's BatchNorm
        auto gemm_l = [&](const AOperand &Aop, const BOperand &Wop, const Epilogue &Eop, int *ns) -> int {
            if (xyz0 && l == 1)
                return launch_wsf0(layer0_operand(d, a->xyz, a->points, sv), g.M, a->conv_w[0], g.Cin[0], sc.mom, nmom, d->training ? 1 : 0,
                                   a->conv_b[0], a->bn_w[0], a->bn_b[0], a->bn_rm[0], a->bn_rv[0],
                                   d->training ? (long long *)a->bn_nbt[0] : nullptr, d->momentum, d->eps, sv.mean[0], sv.istd[0],
                                   sv.scale[0], sv.shift[0], a->conv_w[1], g.Cin[1], Eop, ns, st);
            return launch_gemm(Aop, Wop, g.M, d->C[l], g.Kd[l], Eop, ns, st);
        };
        AOperand A;
        if (l == 0) {
            A = layer0_operand(d, a->xyz, a->points, sv);
        } else {
            A.mode = A_BNRELU;
            A.a = sv.z[l - 1];
            A.lda = d->C[l - 1];
            A.scale = sv.scale[l - 1];
            A.shift = sv.shift[l - 1];
        }
        Epilogue E;
        E.c = sv.z[l];
        E.ldc = d->C[l];
        BOperand W;  // the conv weight (C_l x Cin_l) is read in place; layer 0 maps features-first k' to xyz-first columns
        W.b = a->conv_w[l];
        W.ldb = g.Cin[l];
        W.trans = 1;
        W.perm_D = l == 0 ? d->D : -1;
        W.rows = g.Cin[l];
        int nslab = 0;
        if (l == 0 && sa_delayed(d)) {  // P = F W_f^T on the source points, then Z = P[idx] + W_xyz (x - c) with statistics
            AOperand F;
            F.a = a->points;
            F.lda = d->D;
            BOperand Wf;
            Wf.b = a->conv_w[0] + 3;
            Wf.ldb = g.Cin[0];
            Wf.trans = 1;
            Wf.rows = d->D;
            Epilogue Ep;
            Ep.mode = E_STORE;
            Ep.c = sc.src;
            Ep.ldc = d->C[0];
            PNPP_TRY(launch_gemm(F, Wf, d->B * d->N, d->C[0], d->D, Ep, nullptr, st));
            PNPP_TRY(launch_gather_rel_stats(sc.src, A, a->conv_w[0], g.Cin[0], g.M, d->C[0], sv.z[0],
                                             d->training ? sc.slab : nullptr, &nslab, st));
            StatsView V;
            if (d->training) PNPP_TRY(stats_exchange(sc.slab, nslab, d->C[0], (double)g.M, st, &V));
            PNPP_TRY(launch_bn_finalize_fwd(V.slab, V.nslab, d->C[0], (double)g.M, a->conv_b[0],
                                            a->bn_w[0], a->bn_b[0], a->bn_rm[0], a->bn_rv[0],
                                            d->training ? (long long *)a->bn_nbt[0] : nullptr, d->momentum, d->eps, d->training ? 1 : 0,
                                            sv.mean[0], sv.istd[0], sv.scale[0], sv.shift[0], st, V.count_dev));
        } else if (d->training) {
            E.mode = E_STORE_STATS;
            E.slab = sc.slab;
            // last layer of a level with 32-row neighbourhoods: the max over the neighbourhood is taken from the GEMM's accumulators
            // and finished by the statistics launch (sv.zmax holds the extreme pre-BN values; the backward pass reads them too)
            const bool pool_here = l == d->L - 1 && l > 0 && pool_fused_on() && gemm_pools_in_epilogue(A, g.M, d->C[l], g.Kd[l], d->K);
            if (pool_here) E.pool_ext = sv.zmax, E.pool_arg = sv.arg, E.pool_gamma = a->bn_w[l];
            PNPP_TRY(gemm_l(A, W, E, &nslab));
            StatsView V;
            PNPP_TRY(stats_exchange(sc.slab, nslab, d->C[l], (double)g.M, st, &V));
            PNPP_TRY(launch_bn_finalize_fwd(V.slab, V.nslab, d->C[l], (double)g.M, a->conv_b[l], a->bn_w[l], a->bn_b[l],
                                            a->bn_rm[l], a->bn_rv[l], (long long *)a->bn_nbt[l], d->momentum, d->eps, 1, sv.mean[l], sv.istd[l],
                                            sv.scale[l], sv.shift[l], st, V.count_dev, pool_here ? sv.zmax : nullptr,
                                            pool_here ? a->out : nullptr, g.G, pool_here ? sv.arg : nullptr,
                                            d->group_all ? a->new_xyz : nullptr, d->group_all ? sv.new_xyz : nullptr,
                                            d->group_all ? g.G * 3 : 0));
            pooled = pool_here;
        } else {
            E.mode = E_STORE;
            PNPP_TRY(launch_bn_finalize_fwd(nullptr, 0, d->C[l], (double)g.M, a->conv_b[l], a->bn_w[l], a->bn_b[l],
                                            a->bn_rm[l], a->bn_rv[l], nullptr, d->momentum, d->eps, 0, sv.mean[l], sv.istd[l],
                                            sv.scale[l], sv.shift[l], st));
            PNPP_TRY(gemm_l(A, W, E, nullptr));
        }
    }

    // 3. max over the neighbourhood (pointnet_pp_8dir.py:42-43), unless the last layer's launches have taken it
    const int Lm = d->L - 1;
    if (!pooled) PNPP_TRY(launch_pool_fwd(sv.z[Lm], sv.scale[Lm], sv.shift[Lm], g.G, d->K, d->C[Lm], a->out, sv.arg, st,
                             d->group_all ? a->new_xyz : nullptr, d->group_all ? sv.new_xyz : nullptr, d->group_all ? g.G * 3 : 0,
                             sc.dy[0],   // dy[0] (M x C floats) is idle in the forward pass: >= the K/64 partials per (group, channel)
                             sa_keeps_zmax(d, g) ? sv.zmax : nullptr));
    return PNPP_OK;
}

static int sa_backward_impl(const pnpp_sa_desc *d, const pnpp_sa_bwd_args *a, hipStream_t st) {
    SaGeom g;
    PNPP_TRY(sa_geom(d, &g));
    PNPP_REQUIRE(a && a->xyz && a->dout && a->saved && a->scratch, PNPP_ERR_ARG, "sa_backward: null pointer");
    for (int l = 0; l < d->L; ++l)
        PNPP_REQUIRE(a->conv_w[l] && a->bn_w[l] && a->d_conv_w[l] && a->d_conv_b[l] && a->d_bn_w[l] && a->d_bn_b[l], PNPP_ERR_ARG,
                     "sa_backward: null pointer in layer %d", l);
    const bool want_dpoints = d->D > 0 && a->dpoints != nullptr;
    if (want_dpoints) PNPP_REQUIRE(d->D % 4 == 0, PNPP_ERR_ARG, "sa_backward: feature width D=%d must be a multiple of 4", d->D);
    const SaSaved sv = sa_saved_layout(d, g, const_cast<void *>(a->saved));
    const SaScratch sc = sa_scratch_layout(d, g, a->scratch);

    const int Lm = d->L - 1;
    int cur = 0, nslab = 0, nslab_next = 0;
    // a level with few groups (group_all: one per cloud) whose dZ is materialised anyway: the finalisation launch takes the pooled
    // gradient as it is -- no pool_bwd launch, no dm tensor (PNPP_NO_POOL_BWD_FUSION=1 keeps the launch)
    static const bool pool_bwd_fused = !(getenv("PNPP_NO_POOL_BWD_FUSION") && atoi(getenv("PNPP_NO_POOL_BWD_FUSION")) != 0);
    const bool pooled_src = pool_bwd_fused && g.M <= kSmallM && g.G <= 64 && sa_keeps_zmax(d, g) && (d->C[Lm] & 3) == 0 &&
                            !(d->training && stats_sync_on());
    if (!pooled_src)
        PNPP_TRY(launch_pool_bwd(a->dout, sv.arg, sv.z[Lm], sv.scale[Lm], sv.shift[Lm], sv.mean[Lm], sv.istd[Lm], g.G, d->K,
                                 d->C[Lm], sc.dm, sc.slab, &nslab, st, sa_keeps_zmax(d, g) ? sv.zmax : nullptr));
    // sc.cst holds the BatchNorm-backward constants of the layer being processed; layer l-1's are produced (together with
    // layer l's weight-gradient reduction) by the post-GEMM launch that ends iteration l
    // small-M levels materialise dZ once per layer; that pass rides in the launch that finalises the layer's BatchNorm sums
    const bool small = g.M <= kSmallM;
    auto dz_operand = [&](int l, int buf) {
        AOperand dz;  // the top layer's dense gradient is never materialised (A_DZ_POOL rebuilds it from dm / arg)
        dz.mode = l == Lm ? A_DZ_POOL : A_DZ;
        dz.a = l == Lm ? sc.dm : sc.dy[buf];
        dz.arg = sv.arg;
        dz.K = d->K;
        dz.lda = d->C[l];
        dz.z = sv.z[l];
        dz.cst = sc.cst;
        dz.C = d->C[l];
        return dz;
    };
    AOperand dz_top = dz_operand(Lm, 0);
    if (pooled_src) {
        dz_top.a = a->dout;   // the dZ job masks it itself
        PooledSource ps;
        ps.dout = a->dout, ps.zsel = sv.zmax, ps.scale = sv.scale[Lm], ps.shift = sv.shift[Lm], ps.G = g.G;
        PNPP_TRY(launch_bn_finalize_bwd(nullptr, 0, d->C[Lm], (double)g.M, d->training, a->bn_w[Lm], sv.mean[Lm], sv.istd[Lm], sc.cst,
                                        a->d_bn_w[Lm], a->d_bn_b[Lm], a->d_conv_b[Lm], st, &dz_top, g.M, sc.dzbuf, nullptr, nullptr, &ps));
    } else {
        StatsView V;
        V.slab = sc.slab, V.nslab = nslab;
        if (d->training) PNPP_TRY(stats_exchange(sc.slab, nslab, d->C[Lm], (double)g.M, st, &V));
        PNPP_TRY(launch_bn_finalize_bwd(V.slab, V.nslab, d->C[Lm], (double)g.M, d->training, a->bn_w[Lm], sv.mean[Lm], sv.istd[Lm],
                                        sc.cst, a->d_bn_w[Lm], a->d_bn_b[Lm], a->d_conv_b[Lm], st, small ? &dz_top : nullptr, g.M,
                                        small ? sc.dzbuf : nullptr, V.count_dev, V.local));
    }
    bool dz_ready = small;  // sc.dzbuf holds dZ of the layer about to be processed
    bool dpoints_done = false;
    for (int l = Lm; l >= 0; --l) {
        const int C = d->C[l];
        AOperand dz = dz_operand(l, cur);
        if (small) {  // every consumer would rebuild dZ per 32 x 32 tile: it was written out once instead
            if (!dz_ready) PNPP_TRY(launch_dz_materialize(dz, g.M, C, sc.dzbuf, st));
            dz_ready = false;
            dz = AOperand();
            dz.mode = A_PLAIN;
            dz.a = sc.dzbuf;
            dz.lda = C;
        }

        if (l == 1 && xyz0_applies(g.M, d->D, d->K, d->group_all, d->L, d->C)) {
            // layer 0 convolves relative coordinates only: layer 1's backward rebuilds Z_0 from them, keeps dY_0 on chip and hands
            // layer 0's parameter gradients to the launch that reduces dW_1 (gemm_wsx_kernels.hip) -- two launches end the level
            const AOperand geo = layer0_operand(d, a->xyz, a->points, sv);
            BOperand W;
            W.b = a->conv_w[1];
            W.ldb = d->C[0];
            W.rows = C;
            int workers = 0, rc = PNPP_OK;
            if (try_launch_wsx(dz, W, g.M, C, d->C[0], geo, a->conv_w[0], g.Cin[0], sv.scale[0], sv.shift[0], sc.dwslab, sc.slab, &workers, st,
                               &rc)) {
                PNPP_TRY(rc);
                PNPP_TRY(launch_xyz0_post(sc.dwslab, workers, C, a->d_conv_w[1], g.Cin[1], sc.slab, a->conv_w[0], g.Cin[0], a->bn_w[0],
                                          sv.mean[0], sv.istd[0], (double)g.M, d->training, a->d_conv_w[0], g.Cin[0], a->d_bn_w[0],
                                          a->d_bn_b[0], a->d_conv_b[0], st));
                break;
            }
            // the forward pass of this level kept no Z_0: there is no generic path to fall back to
            PNPP_REQUIRE(false, PNPP_ERR_ARG, "sa_backward: the coordinate-level backward kernel does not take this call (alignment?)");
        }
        AOperand a2;
        if (l == 0) {
            a2 = layer0_operand(d, a->xyz, a->points, sv);
        } else {
            a2.mode = A_BNRELU;
            a2.a = sv.z[l - 1];
            a2.lda = d->C[l - 1];
            a2.scale = sv.scale[l - 1];
            a2.shift = sv.shift[l - 1];
        }
        int nsplit, kp_pad;
        dw_plan(g.M, C, g.Cin[l], &nsplit, &kp_pad);
        int fused_slabs = 0;
        bool pair_done = false;
        if (l > 0) {
            // dY_{l-1} = (dZ_l * W_l) masked by ReLU'(layer l-1), with layer l-1's BN-backward sums; where the
            // weights-stationary kernel applies, dW_l = dZ_l^T * relu(bn(Z_{l-1})) is accumulated in the same launch
            Epilogue E;
            E.mode = E_MASK_STATS;
            E.c = sc.dy[cur ^ 1];
            E.ldc = d->C[l - 1];
            E.slab = sc.slab;
            E.zp = sv.z[l - 1];
            E.scale = sv.scale[l - 1];
            E.shift = sv.shift[l - 1];
            E.mu = sv.mean[l - 1];
            E.istd = sv.istd[l - 1];
            E.dwslab = sc.dwslab;
            E.dw_ld = d->C[l - 1];
            BOperand W;  // W_l (C_l x C_{l-1}) row-major as stored
            W.b = a->conv_w[l];
            W.ldb = d->C[l - 1];
            W.rows = C;
            int dw_slabs = 0, rc = PNPP_OK;
            if (try_launch_da_dw(dz, W, g.M, d->C[l - 1], C, E, &nslab_next, a2, g.Cin[l], sc.dwslab, &nsplit, &kp_pad, st, &rc,
                                 a->d_conv_w[l], g.Cin[l])) {
                PNPP_TRY(rc);  // small-M level: dA and dW of this layer went out as one launch
                pair_done = true;
            } else {
                PNPP_TRY(launch_gemm(dz, W, g.M, d->C[l - 1], C, E, &nslab_next, st, &dw_slabs));
            }
            fused_slabs = dw_slabs;
        }
        const bool xyz_only = l == 0 && d->D == 0 && (a2.mode == A_GATHER || (a2.mode == A_CONCAT && g.M >= 8192)) &&
                              dw_xyz_splits(g.M) <= nsplit * (kp_pad / 4);
        if (l == 0 && sa_delayed(d)) {
            // G = dZ_0 summed per source point (dW_xyz = dZ_0^T (x - c) in the same pass); dW_f = G^T F, dF = G W_f
            const int R = d->B * d->N;
            PNPP_TRY(launch_scatter_dz(dz, a2, d->B, d->S * d->K, C, sc.src, sc.xslab, st));
            AOperand G, F;
            G.a = sc.src;
            G.lda = C;
            F.a = a->points;
            F.lda = d->D;
            dw_plan(R, C, d->D, &nsplit, &kp_pad);
            bool paired = false;
            if (want_dpoints) {  // dW_f = G^T F and dF = G W_f only share G: one launch
                Epilogue E;
                E.mode = E_STORE;
                E.c = a->dpoints;
                E.ldc = d->D;
                BOperand W;
                W.b = a->conv_w[0] + 3;
                W.ldb = g.Cin[0];
                W.rows = C;
                int rc = PNPP_OK;
                paired = try_launch_da_dw(G, W, R, d->D, C, E, nullptr, F, d->D, sc.dwslab, &nsplit, &kp_pad, st, &rc);
                if (paired) PNPP_TRY(rc);
            }
            if (!paired) PNPP_TRY(launch_dw(G, C, F, d->D, R, sc.dwslab, nsplit, kp_pad, st));
            // both partial sets of W_0 -- coordinate columns 0..2, feature columns 3.. -- in one launch
            PNPP_TRY(launch_slab_reduce2(sc.xslab, scatter_dz_splits(R), C, 4, 3, a->d_conv_w[0], g.Cin[0], sc.dwslab, nsplit, C, kp_pad,
                                         d->D, a->d_conv_w[0] + 3, g.Cin[0], st));
            if (want_dpoints && !paired) {
                Epilogue E;
                E.mode = E_STORE;
                E.c = a->dpoints;
                E.ldc = d->D;
                BOperand W;
                W.b = a->conv_w[0] + 3;
                W.ldb = g.Cin[0];
                W.rows = C;
                PNPP_TRY(launch_gemm(G, W, R, d->D, C, E, nullptr, st));
            }
            break;
        }
        if (xyz_only) {  // C x 3 gradient: streaming kernel (the MFMA tiles would be 95 % padding)
            PNPP_TRY(launch_dw_xyz(dz, C, a2, g.M, sc.dwslab, st));
            nsplit = dw_xyz_splits(g.M), kp_pad = 4;
        } else if (fused_slabs == 0 && !pair_done) {  // dW_l = dZ_l^T * A_l as its own launch (layer 0, group_all layers, odd shapes)
            // group_all layer 0 with a feature gradient to return: dW_0 and dF = dZ_0 W_f only share dZ_0 -- one launch
            bool paired0 = false;
            if (l == 0 && want_dpoints && d->group_all) {
                Epilogue E;
                E.mode = E_STORE;
                E.ldc = d->D;
                E.c = a->dpoints;
                BOperand W;
                W.b = a->conv_w[0] + 3;
                W.ldb = g.Cin[0];
                W.rows = C;
                int rc = PNPP_OK;
                paired0 = try_launch_da_dw(dz, W, g.M, d->D, C, E, nullptr, a2, g.Cin[0], sc.dwslab, &nsplit, &kp_pad, st, &rc);
                if (paired0) PNPP_TRY(rc);
                dpoints_done = paired0;
            }
            if (!paired0) PNPP_TRY(launch_dw(dz, C, a2, g.Cin[l], g.M, sc.dwslab, nsplit, kp_pad, st));
        }
        if (l > 0) {  // reduce dW_l's partials and finalise layer l-1's BatchNorm-backward sums in one launch
            const int Cp = d->C[l - 1];
            const AOperand dz_next = dz_operand(l - 1, cur ^ 1);  // dY_{l-1} was just written to sc.dy[cur ^ 1]
            StatsView V;
            V.slab = sc.slab, V.nslab = nslab_next;
            if (d->training) PNPP_TRY(stats_exchange(sc.slab, nslab_next, Cp, (double)g.M, st, &V));
            PNPP_TRY(launch_post_gemm(V.slab, V.nslab, Cp, (double)g.M, d->training, a->bn_w[l - 1], sv.mean[l - 1],
                                      sv.istd[l - 1], sc.cst, a->d_bn_w[l - 1], a->d_bn_b[l - 1],
                                      a->d_conv_b[l - 1], sc.dwslab, fused_slabs > 0 ? fused_slabs : nsplit, C,
                                      fused_slabs > 0 ? Cp : kp_pad, g.Cin[l], -1, a->d_conv_w[l], g.Cin[l], st,
                                      small ? &dz_next : nullptr, g.M, small ? sc.dzbuf : nullptr, V.count_dev, V.local));
            dz_ready = small;
        } else {
            PNPP_TRY(launch_slab_reduce(sc.dwslab, nsplit, C, kp_pad, g.Cin[l], d->D, a->d_conv_w[l], g.Cin[l], st));
        }

        if (l > 0) {
            nslab = nslab_next;
            cur ^= 1;
        } else if (want_dpoints) {
            Epilogue E;
            E.mode = E_STORE;
            E.ldc = d->D;
            BOperand W;  // feature columns of W_0: (C_0 x (3+D)) row-major, skip the three xyz columns
            W.b = a->conv_w[0] + 3;
            W.ldb = g.Cin[0];
            W.rows = C;
            if (d->group_all) {  // rows are the points themselves
                E.c = a->dpoints;
                if (!dpoints_done) PNPP_TRY(launch_gemm(dz, W, g.M, d->D, C, E, nullptr, st));
            } else {
                E.c = sc.dy[cur ^ 1];
                PNPP_TRY(launch_gemm(dz, W, g.M, d->D, C, E, nullptr, st));
                PNPP_TRY(launch_fill_zero(a->dpoints, (size_t)d->B * d->N * d->D * sizeof(float), st));
                PNPP_TRY(launch_scatter_rows_bwd(sc.dy[cur ^ 1], sv.idx, d->B, d->N, d->D, d->S * d->K, a->dpoints, st));
            }
        }
    }
    return PNPP_OK;
}

}  // namespace pnpp

using namespace pnpp;

extern "C" const char *pnpp_last_error(void) { return g_err; }
extern "C" int pnpp_abi_version(void) { return 5; }

extern "C" int pnpp_set_stats_exchange(pnpp_stats_exchange_fn fn, void *user, double *buf, size_t buf_doubles) {
    if (!fn) {
        g_sx = StatsExchange();
        return PNPP_OK;
    }
    PNPP_REQUIRE(buf && buf_doubles >= 2 * (2 * 32 + 1), PNPP_ERR_ARG, "set_stats_exchange: a device buffer of doubles is required");
    g_sx.fn = fn, g_sx.user = user, g_sx.buf = buf, g_sx.half = buf_doubles / 2;
    return PNPP_OK;
}
extern "C" int pnpp_stats_exchange_enabled(void) { return g_sx.fn ? 1 : 0; }

extern "C" int pnpp_profile_enable(int on) {
    for (auto &r : g_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    g_prof = on != 0;
    return PNPP_OK;
}

extern "C" int pnpp_profile_report(char *buf, size_t buflen) {
    PNPP_REQUIRE(buf && buflen > 0, PNPP_ERR_ARG, "profile_report: null buffer");
    std::map<std::string, std::pair<long, double>> agg;
    std::vector<std::string> order;
    for (auto &r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
            set_error("profile_report: event query failed");
            return PNPP_ERR_LAUNCH;
        }
        if (!agg.count(r.tag)) order.push_back(r.tag);
        agg[r.tag].first += 1;
        agg[r.tag].second += (double)ms;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    size_t off = 0;
    buf[0] = 0;
    for (auto &t : order) {
        int n = snprintf(buf + off, buflen - off, "%s\t%ld\t%.6f\n", t.c_str(), agg[t].first, agg[t].second);
        if (n < 0 || (size_t)n >= buflen - off) break;
        off += (size_t)n;
    }
    return (int)order.size();
}

extern "C" size_t pnpp_sa_saved_bytes(const pnpp_sa_desc *d) {
    SaGeom g;
    if (sa_geom(d, &g) != PNPP_OK) return 0;
    return sa_saved_layout(d, g, nullptr).bytes;
}
extern "C" size_t pnpp_sa_scratch_bytes(const pnpp_sa_desc *d) {
    SaGeom g;
    if (sa_geom(d, &g) != PNPP_OK) return 0;
    return sa_scratch_layout(d, g, nullptr).bytes;
}
extern "C" const int32_t *pnpp_sa_saved_neighbours(const pnpp_sa_desc *d, const void *saved) {
    SaGeom g;
    if (sa_geom(d, &g) != PNPP_OK || d->group_all) return nullptr;
    return sa_saved_layout(d, g, const_cast<void *>(saved)).idx;
}
extern "C" const int32_t *pnpp_sa_saved_argmax(const pnpp_sa_desc *d, const void *saved) {
    SaGeom g;
    if (sa_geom(d, &g) != PNPP_OK) return nullptr;
    return sa_saved_layout(d, g, const_cast<void *>(saved)).arg;
}
extern "C" int pnpp_sa_saved_relu_mask(const pnpp_sa_desc *d, const void *saved, const float *xyz, const float *conv_w0, int layer,
                                       uint8_t *out, void *stream) {
    SaGeom g;
    PNPP_TRY(sa_geom(d, &g));
    PNPP_REQUIRE(saved && out && layer >= 0 && layer < d->L, PNPP_ERR_ARG, "sa_saved_relu_mask: null pointer or layer %d out of range", layer);
    const SaSaved sv = sa_saved_layout(d, g, const_cast<void *>(saved));
    if (layer == 0 && xyz0_applies(g.M, d->D, d->K, d->group_all, d->L, d->C)) {   // never stored: rebuilt as the kernels rebuild it
        PNPP_REQUIRE(xyz && conv_w0, PNPP_ERR_ARG, "sa_saved_relu_mask: layer 0 of a level on raw coordinates needs xyz and conv_w0");
        return launch_xyz0_mask(layer0_operand(d, xyz, nullptr, sv), g.M, conv_w0, g.Cin[0], sv.scale[0], sv.shift[0], out, as_stream(stream));
    }
    return launch_relu_mask(sv.z[layer], sv.scale[layer], sv.shift[layer], (size_t)g.M * d->C[layer], d->C[layer], out, as_stream(stream));
}
extern "C" unsigned pnpp_build_flags(void) {
    return gemm_build_flags() | wsp_build_flags() | wsx_build_flags() | wsq_build_flags() | fc_build_flags() | wsf_build_flags() | wsd3_build_flags() |
           mid3_build_flags();
}
extern "C" int pnpp_sa_group_pair(const pnpp_sa_desc *d1, const pnpp_sa_desc *d2, const float *xyz, const int32_t *centre1,
                                  const int32_t *centre2, void *saved1, float *new_xyz1, void *saved2, float *new_xyz2, void *stream) {
    SaGeom g1, g2;
    PNPP_TRY(sa_geom(d1, &g1));
    PNPP_TRY(sa_geom(d2, &g2));
    PNPP_REQUIRE(!d1->group_all && !d2->group_all, PNPP_ERR_ARG, "sa_group_pair: both levels must group neighbourhoods");
    PNPP_REQUIRE(d1->B == d2->B && d2->N == d1->S, PNPP_ERR_ARG, "sa_group_pair: level 2 must take level 1's %d centres (got N=%d)",
                 d1->S, d2->N);
    PNPP_REQUIRE(xyz && centre1 && centre2 && saved1 && saved2 && new_xyz1 && new_xyz2, PNPP_ERR_ARG, "sa_group_pair: null pointer");
    const SaSaved s1 = sa_saved_layout(d1, g1, saved1), s2 = sa_saved_layout(d2, g2, saved2);
    return launch_knn_pair(xyz, d1->B, d1->N, centre1, d1->S, d1->K, s1.idx, new_xyz1, s1.new_xyz, centre2, d2->S, d2->K, s2.idx,
                           new_xyz2, s2.new_xyz, as_stream(stream));
}

extern "C" int pnpp_sa_forward(const pnpp_sa_desc *d, const pnpp_sa_fwd_args *a, void *stream) {
    return sa_forward_impl(d, a, as_stream(stream));
}
extern "C" int pnpp_sa_backward(const pnpp_sa_desc *d, const pnpp_sa_bwd_args *a, void *stream) {
    return sa_backward_impl(d, a, as_stream(stream));
}
