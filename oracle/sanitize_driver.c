/*
 * oracle/sanitize_driver.c -- TEST INFRASTRUCTURE ONLY.  Exercises every entry point of index_ops.c on random and edge-case
 * inputs (one point, k == N, duplicates, empty radius, N not a multiple of anything) so that a build with
 * -fsanitize=address,undefined (make -C oracle sanitize; tests/test_oracle_golden.py::test_c_oracle_under_sanitizers)
 * reports out-of-bounds accesses, leaks and undefined behaviour of the CPU oracle.  GPU AddressSanitizer is not available
 * on the build pool; the host side is what can be sanitised.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_square_distance(const float *src, const float *dst, int B, int S, int N, float *out);
int oracle_knn(const float *new_xyz, const float *xyz, int B, int S, int N, int k, int32_t *idx);
void oracle_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out);
void oracle_ball_query(const float *new_xyz, const float *xyz, int B, int S, int N, float radius, int nsample, int32_t *idx);

static unsigned lcg = 12345u;
static float rnd(void) {
    lcg = lcg * 1664525u + 1013904223u;
    return (float)(lcg >> 8) / 16777216.0f;
}

static int run(int B, int S, int N, int k, int npoint, float radius, int dup) {
    float *xyz = malloc(sizeof(float) * (size_t)B * N * 3), *q = malloc(sizeof(float) * (size_t)B * S * 3);
    for (int i = 0; i < B * N * 3; ++i) xyz[i] = dup ? (float)((int)(rnd() * 4)) * 0.25f : rnd();
    for (int i = 0; i < B * S * 3; ++i) q[i] = dup ? (float)((int)(rnd() * 4)) * 0.25f : rnd();
    float *d = malloc(sizeof(float) * (size_t)B * S * N);
    oracle_square_distance(q, xyz, B, S, N, d);
    int32_t *idx = malloc(sizeof(int32_t) * (size_t)B * S * k);
    int rc = oracle_knn(q, xyz, B, S, N, k, idx);
    long acc = rc;
    if (rc == 0)
        for (int i = 0; i < B * S * k; ++i) {
            if (idx[i] < 0 || idx[i] >= N) return 1;
            acc += idx[i];
        }
    int32_t *start = malloc(sizeof(int32_t) * (size_t)B), *f = malloc(sizeof(int32_t) * (size_t)B * npoint);
    for (int b = 0; b < B; ++b) start[b] = (int32_t)(rnd() * N) % N;
    oracle_fps(xyz, B, N, npoint, start, f);
    for (int i = 0; i < B * npoint; ++i) {
        if (f[i] < 0 || f[i] >= N) return 2;
        acc += f[i];
    }
    int32_t *bq = malloc(sizeof(int32_t) * (size_t)B * S * k);
    oracle_ball_query(q, xyz, B, S, N, radius, k, bq);
    for (int i = 0; i < B * S * k; ++i) {
        if (bq[i] < 0 || bq[i] > N) return 3;   /* N itself marks "nothing inside the radius" (PointNet++Demo.py:67-69) */
        acc += bq[i];
    }
    free(xyz), free(q), free(d), free(idx), free(start), free(f), free(bq);
    return acc < 0;
}

int main(void) {
    int bad = 0;
    bad |= run(2, 16, 257, 32, 40, 0.2f, 0);
    bad |= run(1, 1, 1, 1, 1, 0.0f, 0);          /* a single point */
    bad |= run(3, 5, 33, 33, 33, 10.0f, 0);      /* k == N, npoint == N, everything inside the radius */
    bad |= run(2, 8, 100, 16, 50, 0.3f, 1);      /* duplicated coordinates: exact ties */
    bad |= run(1, 4, 64, 8, 8, 1e-6f, 0);        /* (almost) empty balls */
    if (oracle_knn((const float *)&bad, (const float *)&bad, 0, 0, 1, 2, NULL) != -1) bad |= 4;   /* k > N is refused */
    printf(bad ? "FAILED %d\n" : "sanitize_driver ok\n", bad);
    return bad;
}
