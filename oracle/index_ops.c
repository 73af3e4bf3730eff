/*
 * oracle/index_ops.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Plain-C restatement of the reference's integer/index-producing primitives, written so
 * that every floating-point operation happens in exactly the order (and with exactly the
 * fused/unfused rounding) that the reference's CPU PyTorch path executes.  The HIP kernels
 * must reproduce these index sets bit-for-bit.
 *
 * Reference call sites restated here (paths relative to /root/reference):
 *   - square_distance            models/base.py:20-27
 *   - query_ball_point (= kNN)   models/base.py:29-35
 *   - farthest_point_sample      PointNet++Demo.py:8-29
 *   - query_ball_point (radius)  PointNet++Demo.py:49-70
 *
 * Pinned by tests/golden/index_*.npz, which oracle/make_golden.py captured from the
 * imported reference itself (see tests/test_oracle_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; contraction MUST stay off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- the ATen-CPU arithmetic of models/base.py:20-27 --------------------------------
 * dist  = -2 * matmul(src, dst^T)        -> K=3 sgemm dot = fma(a2,b2, fma(a1,b1, a0*b0))
 * dist += sum(src**2, -1)[:, :, None]    -> (a0^2 + a1^2) + a2^2, no fusion
 * dist += sum(dst**2, -1)[:, None, :]
 * (bit-equal on every shape/thread-count probed.)                      */
static inline float sq3(const float *p) {
    float s = p[0] * p[0];
    s = s + p[1] * p[1];
    s = s + p[2] * p[2];
    return s;
}

static inline float pair_dist(const float *a, const float *b, float sa, float sb) {
    float dot = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
    float d = -2.0f * dot;
    d = d + sa;
    d = d + sb;
    return d;
}

/* src (B,S,3), dst (B,N,3) -> out (B,S,N) */
void oracle_square_distance(const float *src, const float *dst, int B, int S, int N, float *out) {
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *a = src + ((size_t)b * S + s) * 3;
            float sa = sq3(a);
            for (int n = 0; n < N; ++n) {
                const float *q = dst + ((size_t)b * N + n) * 3;
                out[((size_t)b * S + s) * N + n] = pair_dist(a, q, sa, sq3(q));
            }
        }
}

/* Total order used for neighbour selection: smaller distance first, lower index on ties.
 * (The reference's topk(sorted=False) leaves order and tie choice unspecified; the build
 * fixes both.  Parity is therefore defined on tie-free inputs, as sorted index sets.)   */
typedef struct { float d; int32_t i; } cand_t;

static int cand_cmp(const void *pa, const void *pb) {
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* kNN grouping: new_xyz (B,S,3), xyz (B,N,3) -> idx (B,S,k) ascending by (distance, index).
 * Returns 0, or -1 if k > N. */
int oracle_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx) {
    if (k > N) return -1;
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)N);
    float *sb = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        for (int n = 0; n < N; ++n) sb[n] = sq3(xyz + ((size_t)b * N + n) * 3);
        for (int s = 0; s < S; ++s) {
            const float *a = new_xyz + ((size_t)b * S + s) * 3;
            float sa = sq3(a);
            for (int n = 0; n < N; ++n) {
                c[n].d = pair_dist(a, xyz + ((size_t)b * N + n) * 3, sa, sb[n]);
                c[n].i = n;
            }
            qsort(c, (size_t)N, sizeof(cand_t), cand_cmp);
            for (int j = 0; j < k; ++j) idx[((size_t)b * S + s) * k + j] = c[j].i;
        }
    }
    free(c);
    free(sb);
    return 0;
}

/* Farthest point sampling, PointNet++Demo.py:8-29.
 *   distance = 1e10; farthest = start[b]
 *   repeat npoint times: record farthest; dist = sum((xyz - c)^2, -1)  [(dx^2+dy^2)+dz^2]
 *                        distance = min(distance, dist) (strict <); farthest = first argmax */
void oracle_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out) {
    float *dist = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int n = 0; n < N; ++n) dist[n] = 1e10f;
        int far = start[b];
        for (int i = 0; i < npoint; ++i) {
            out[(size_t)b * npoint + i] = far;
            const float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
            float best = -INFINITY;
            int besti = 0;
            for (int n = 0; n < N; ++n) {
                float dx = p[n * 3 + 0] - cx, dy = p[n * 3 + 1] - cy, dz = p[n * 3 + 2] - cz;
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (d < dist[n]) dist[n] = d;
                if (dist[n] > best) { best = dist[n]; besti = n; }
            }
            far = besti;
        }
    }
    free(dist);
}

/* Radius ball query, PointNet++Demo.py:49-70.
 *   d2 = sum((new - xyz)^2, -1); keep n with !(d2 > r2) in ascending n; first nsample;
 *   missing slots are filled with the first kept index (N when nothing is inside).      */
void oracle_ball_query(const float *new_xyz, const float *xyz, int B, int S, int N, float radius,
                       int nsample, int32_t *idx) {
    const float r2 = (float)((double)radius * (double)radius);
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s) {
            const float *a = new_xyz + ((size_t)b * S + s) * 3;
            int32_t *o = idx + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int n = 0; n < N && cnt < nsample; ++n) {
                const float *q = xyz + ((size_t)b * N + n) * 3;
                float dx = a[0] - q[0], dy = a[1] - q[1], dz = a[2] - q[2];
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (!(d > r2)) o[cnt++] = n;
            }
            int first = cnt ? o[0] : N;
            for (int j = cnt; j < nsample; ++j) o[j] = first;
        }
}
