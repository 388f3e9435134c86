/* oracle/oracle_internal.h -- shared between hamming_ref.c and bench_ref.c (test infrastructure). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int kind;
    uint32_t n;
    uint8_t *db_hashes;  /* n * hsize */
    uint32_t *offsets;   /* num_chunks*num_buckets + 1 */
    uint32_t *values;    /* num_chunks * n dense ids */
} mih_t;

typedef struct {
    uint64_t *data;
    size_t *dirty;
    size_t ndirty, cap;
} sbs_t;

typedef struct { uint32_t *p; size_t n, cap; } vec_t;

void rph_ref_sbs_init(sbs_t *s, size_t size);
void rph_ref_sbs_destroy(sbs_t *s);
/* the map closure of find_groups (hamminghash.rs:200-241) for one query i */
void rph_ref_query_adjacency(const mih_t *m, uint32_t i, uint32_t max_dist, sbs_t *visited, vec_t *results);
/* greedy clustering (hamminghash.rs:245-268) over prebuilt adjacency lists */
uint32_t rph_ref_greedy_cluster(uint32_t n, vec_t *adj, uint32_t **members_out, uint32_t **offsets_out);

mih_t *rph_ref_mih_new(int kind, const uint8_t *hashes, uint32_t n);
void rph_ref_mih_free(mih_t *m);
uint32_t rph_ref_hamming256(const uint8_t *a, const uint8_t *b);
#endif
