/*
 * rq_cli.c -- librq.so driven from plain C: no HIP headers, no Python, no torch.
 * Reads one batch of Raft groups as text on stdin, runs ONE fused round on GPU 0
 * and prints every output as text.  It is what a maintainer of a C or cgo host
 * would write first against include/rq.h, and the tests use it to check that
 * the library stands on its own outside a torch process.
 *
 *   rq_cli host      rq_fused_round on host buffers
 *   rq_cli device    rq_device_alloc + rq_copy_to_device + rq_round_dev on
 *                    the default stream + rq_stream_sync + rq_copy_from_device
 *
 * Input (whitespace separated integers):
 *   G n
 *   match[n][G]  commit[G]  leader_term[G]  log_off[G+1]  log_term[log_off[G]]
 *   voter_term[n][G]  voted_for[n][G]  voter_last_log_term[n][G]  cand_term[G]  cand_id[G]
 * Output, one line each: commit, voter_term (n lines), grant_mask, stepdown_mask,
 * elected, max_term, status.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rq.h"

static int32_t *read_i32(size_t count) {
    int32_t *a = (int32_t *)calloc(count ? count : 1, sizeof *a);
    for (size_t i = 0; a && i < count; i++) {
        long long v;
        if (scanf("%lld", &v) != 1) { fprintf(stderr, "rq_cli: short input\n"); exit(2); }
        a[i] = (int32_t)v;
    }
    return a;
}

static void print_i32(const int32_t *a, size_t count) {
    for (size_t i = 0; i < count; i++) printf(i ? " %d" : "%d", a[i]);
    printf("\n");
}

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != RQ_OK) {                                                      \
            fprintf(stderr, "rq_cli: %s -> %s (%d), hip %d\n", #call, rq_strerror(rc_), rc_, \
                    e ? rq_last_hip_error(e) : 0);                               \
            return 1;                                                            \
        }                                                                        \
    } while (0)

/* device copy of a host array */
static int to_device(rq_engine *e, const void *src, size_t bytes, void **out) {
    int rc = rq_device_alloc(e, bytes ? bytes : 4, RQ_MEM_DEFAULT, out);
    if (rc != RQ_OK) return rc;
    return rq_copy_to_device(e, *out, src, bytes);
}

int main(int argc, char **argv) {
    const int device_mode = argc > 1 && strcmp(argv[1], "device") == 0;
    rq_engine *e = NULL;
    long long G_in, n_in;
    if (scanf("%lld %lld", &G_in, &n_in) != 2 || G_in < 1 || n_in < 1 || n_in > RQ_MAX_SLOTS) {
        fprintf(stderr, "rq_cli: bad header\n");
        return 2;
    }
    const uint32_t G = (uint32_t)G_in, n = (uint32_t)n_in;
    const size_t plane = (size_t)n * G;
    int32_t *match = read_i32(plane), *commit = read_i32(G), *leader_term = read_i32(G);
    uint32_t *log_off = (uint32_t *)read_i32((size_t)G + 1);
    const size_t arena = log_off[G];
    int32_t *log_term = read_i32(arena);
    int32_t *voter_term = read_i32(plane), *voted_for = read_i32(plane), *last_log_term = read_i32(plane);
    int32_t *cand_term = read_i32(G), *cand_id = read_i32(G);

    int32_t *commit_out = (int32_t *)calloc(G, 4), *voter_term_out = (int32_t *)calloc(plane, 4);
    int32_t *max_term = (int32_t *)calloc(G, 4);
    uint16_t *grant = (uint16_t *)calloc(G, 2), *stepdown = (uint16_t *)calloc(G, 2);
    uint8_t *elected = (uint8_t *)calloc(G, 1), *status = (uint8_t *)calloc(G, 1);

    CHECK(rq_engine_create(0, G, n, arena + 1, &e));
    if (!device_mode) {
        CHECK(rq_fused_round(e, G, n, G, match, commit, leader_term, log_off, log_term, voter_term,
                             voted_for, last_log_term, cand_term, cand_id, commit_out, voter_term_out,
                             grant, stepdown, elected, max_term, status));
    } else {
        void *d_match, *d_commit, *d_lt, *d_off, *d_log, *d_vt, *d_vf, *d_ll, *d_ct, *d_ci;
        void *d_co, *d_vto, *d_gm, *d_sm, *d_el, *d_mx, *d_st;
        CHECK(to_device(e, match, plane * 4, &d_match));
        CHECK(to_device(e, commit, (size_t)G * 4, &d_commit));
        CHECK(to_device(e, leader_term, (size_t)G * 4, &d_lt));
        CHECK(to_device(e, log_off, ((size_t)G + 1) * 4, &d_off));
        CHECK(to_device(e, log_term, arena * 4, &d_log));
        CHECK(to_device(e, voter_term, plane * 4, &d_vt));
        CHECK(to_device(e, voted_for, plane * 4, &d_vf));
        CHECK(to_device(e, last_log_term, plane * 4, &d_ll));
        CHECK(to_device(e, cand_term, (size_t)G * 4, &d_ct));
        CHECK(to_device(e, cand_id, (size_t)G * 4, &d_ci));
        CHECK(to_device(e, commit_out, (size_t)G * 4, &d_co));
        CHECK(to_device(e, voter_term_out, plane * 4, &d_vto));
        CHECK(to_device(e, grant, (size_t)G * 2, &d_gm));
        CHECK(to_device(e, stepdown, (size_t)G * 2, &d_sm));
        CHECK(to_device(e, elected, G, &d_el));
        CHECK(to_device(e, max_term, (size_t)G * 4, &d_mx));
        CHECK(to_device(e, status, G, &d_st));
        /* stream NULL = the device's default stream; variant 0 = library default build */
        rq_round_desc d;
        memset(&d, 0, sizeof d);
        d.struct_size = sizeof d;
        d.layout = RQ_LAYOUT_I32;
        d.what = RQ_ROUND_COMMIT | RQ_ROUND_VOTE;
        d.G = G; d.n = n; d.ld = G;
        d.log_terms = arena;
        d.match = d_match; d.commit = d_commit; d.leader_term = d_lt;
        d.log_off = (const uint32_t *)d_off; d.log_term = d_log;
        d.voter_term = d_vt; d.voted_for = d_vf; d.voter_last_log_term = d_ll;
        d.cand_term = d_ct; d.cand_id = d_ci;
        d.commit_out = d_co; d.voter_term_out = d_vto;
        d.grant_mask = (uint16_t *)d_gm; d.stepdown_mask = (uint16_t *)d_sm; d.elected = (uint8_t *)d_el;
        d.max_term = d_mx; d.status_out = (uint8_t *)d_st;
        CHECK(rq_round_dev(e, NULL, &d));
        CHECK(rq_stream_sync(e, NULL));
        CHECK(rq_copy_from_device(e, commit_out, d_co, (size_t)G * 4));
        CHECK(rq_copy_from_device(e, voter_term_out, d_vto, plane * 4));
        CHECK(rq_copy_from_device(e, grant, d_gm, (size_t)G * 2));
        CHECK(rq_copy_from_device(e, stepdown, d_sm, (size_t)G * 2));
        CHECK(rq_copy_from_device(e, elected, d_el, G));
        CHECK(rq_copy_from_device(e, max_term, d_mx, (size_t)G * 4));
        CHECK(rq_copy_from_device(e, status, d_st, G));
        void *all[] = {d_match, d_commit, d_lt, d_off, d_log, d_vt, d_vf, d_ll, d_ct, d_ci,
                       d_co, d_vto, d_gm, d_sm, d_el, d_mx, d_st};
        for (size_t i = 0; i < sizeof all / sizeof all[0]; i++) CHECK(rq_device_free(e, all[i]));
    }
    print_i32(commit_out, G);
    for (uint32_t p = 0; p < n; p++) print_i32(voter_term_out + (size_t)p * G, G);
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", grant[g]);
    printf("\n");
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", stepdown[g]);
    printf("\n");
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", elected[g]);
    printf("\n");
    print_i32(max_term, G);
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", status[g]);
    printf("\n");
    CHECK(rq_engine_destroy(e));
    return 0;
}
