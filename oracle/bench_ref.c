/*
 * oracle/bench_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Threaded drivers around the CPU oracle, used ONLY by bench.py's cpu_baseline
 * leg (and smoke-level tests).  They parallelise the oracle the way the
 * reference parallelises itself:
 *   - one image per task over a thread pool (rayon par_iter, scanner.rs:1202)
 *   - per-query MIH adjacency with a per-thread SparseBitSet
 *     (into_par_iter().map_init, hamminghash.rs:196-243), then serial greedy
 *     clustering (hamminghash.rs:245-268)
 *   - a brute-force XOR-popcount sweep for an apples-to-apples pairs/s figure.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "oracle_internal.h"

int rph_ref_pdq_features(const uint8_t *px, int w, int h, int stride_bytes, int channels, float *coeffs,
                         float *quality);
void rph_ref_to_hash(const float *coeffs, uint8_t *hash);

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------- PDQ over a batch of images ---------------- */
typedef struct {
    const uint8_t *rgb;
    int n, w, h;
    uint8_t *hashes;
    float *quality;
    atomic_int next;
} pdq_job_t;

static void *pdq_worker(void *arg)
{
    pdq_job_t *j = (pdq_job_t *)arg;
    for (;;) {
        int k = atomic_fetch_add(&j->next, 1);
        if (k >= j->n) break;
        float c[256], q;
        rph_ref_pdq_features(j->rgb + (size_t)k * (size_t)j->w * (size_t)j->h * 3, j->w, j->h, j->w * 3, 3, c, &q);
        rph_ref_to_hash(c, j->hashes + (size_t)k * 32);
        if (j->quality) j->quality[k] = q;
    }
    return NULL;
}

/* Returns elapsed seconds. */
double rph_ref_bench_pdq(const uint8_t *rgb, int n, int w, int h, int nthreads, uint8_t *hashes, float *quality)
{
    pdq_job_t job = {rgb, n, w, h, hashes, quality, 0};
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    double t0 = now_s();
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, pdq_worker, &job);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    double t1 = now_s();
    free(th);
    return t1 - t0;
}

/* ---------------- brute-force all pairs (i<j) ---------------- */
typedef struct {
    const uint64_t *h; /* n x 4 u64 */
    uint32_t n, thr;
    atomic_uint next_row;
    atomic_ullong hits;
} ap_job_t;

static void *ap_worker(void *arg)
{
    ap_job_t *j = (ap_job_t *)arg;
    unsigned long long local = 0;
    for (;;) {
        uint32_t i0 = atomic_fetch_add(&j->next_row, 64);
        if (i0 >= j->n) break;
        uint32_t i1 = i0 + 64 < j->n ? i0 + 64 : j->n;
        for (uint32_t i = i0; i < i1; i++) {
            const uint64_t *a = j->h + (size_t)i * 4;
            for (uint32_t k = i + 1; k < j->n; k++) {
                const uint64_t *b = j->h + (size_t)k * 4;
                uint32_t d = (uint32_t)__builtin_popcountll(a[0] ^ b[0]) + (uint32_t)__builtin_popcountll(a[1] ^ b[1]) +
                             (uint32_t)__builtin_popcountll(a[2] ^ b[2]) + (uint32_t)__builtin_popcountll(a[3] ^ b[3]);
                local += d <= j->thr;
            }
        }
    }
    atomic_fetch_add(&j->hits, local);
    return NULL;
}

/* Counts pairs with d <= thr; returns elapsed seconds. */
double rph_ref_bench_all_pairs256(const uint8_t *hashes, uint32_t n, uint32_t thr, int nthreads, uint64_t *n_hits)
{
    uint64_t *copy = (uint64_t *)malloc((size_t)n * 32);
    memcpy(copy, hashes, (size_t)n * 32);
    ap_job_t job = {copy, n, thr, 0, 0};
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    double t0 = now_s();
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, ap_worker, &job);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    double t1 = now_s();
    if (n_hits) *n_hits = atomic_load(&job.hits);
    free(th);
    free(copy);
    return t1 - t0;
}

/* ---------------- find_groups with parallel adjacency ---------------- */
typedef struct {
    const mih_t *m;
    uint32_t n, max_dist;
    uint32_t q_limit; /* number of queries to run (sampled timing), <= n */
    vec_t *adj;
    atomic_uint next;
} fg_job_t;

static void *fg_worker(void *arg)
{
    fg_job_t *j = (fg_job_t *)arg;
    sbs_t vis;
    vec_t res = {0};
    rph_ref_sbs_init(&vis, j->n);
    for (;;) {
        uint32_t i0 = atomic_fetch_add(&j->next, 256);
        if (i0 >= j->q_limit) break;
        uint32_t i1 = i0 + 256 < j->q_limit ? i0 + 256 : j->q_limit;
        for (uint32_t i = i0; i < i1; i++) {
            rph_ref_query_adjacency(j->m, i, j->max_dist, &vis, &res);
            if (res.n) {
                j->adj[i].p = (uint32_t *)malloc(res.n * sizeof(uint32_t));
                memcpy(j->adj[i].p, res.p, res.n * sizeof(uint32_t));
                j->adj[i].n = j->adj[i].cap = res.n;
            }
        }
    }
    free(res.p);
    rph_ref_sbs_destroy(&vis);
    return NULL;
}

/*
 * MIHIndex::new + find_groups (kind 0 = u64, 1 = [u8;32]) with nthreads workers.
 * q_limit < n times only the first q_limit queries (sampled baseline); groups
 * are then meaningless and not returned.  times[0] = index build s,
 * times[1] = adjacency s, times[2] = greedy s.
 */
uint32_t rph_ref_bench_find_groups(int kind, const uint8_t *hashes, uint32_t n, uint32_t max_dist, int nthreads,
                                   uint32_t q_limit, double *times, uint32_t **members_out, uint32_t **offsets_out)
{
    double t0 = now_s();
    mih_t *m = rph_ref_mih_new(kind, hashes, n);
    double t1 = now_s();
    if (q_limit == 0 || q_limit > n) q_limit = n;
    fg_job_t job;
    job.m = m;
    job.n = n;
    job.max_dist = max_dist;
    job.q_limit = q_limit;
    job.adj = (vec_t *)calloc(n ? n : 1, sizeof(vec_t));
    atomic_init(&job.next, 0);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, fg_worker, &job);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    double t2 = now_s();
    uint32_t ng = 0;
    if (q_limit == n && members_out && offsets_out) ng = rph_ref_greedy_cluster(n, job.adj, members_out, offsets_out);
    double t3 = now_s();
    for (uint32_t i = 0; i < n; i++) free(job.adj[i].p);
    free(job.adj);
    free(th);
    rph_ref_mih_free(m);
    if (times) {
        times[0] = t1 - t0;
        times[1] = t2 - t1;
        times[2] = t3 - t2;
    }
    return ng;
}
