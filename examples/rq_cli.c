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
 *   rq_cli segmented the device path for groups of DIFFERENT sizes kept in size order: two more
 *                    input lines (old_mask[G], new_mask[G]); the program works out the size
 *                    table itself (rq_round_desc.n_segments) and launches once
 *   rq_cli rounds    the same input cut into two images at a 64-group boundary, decided by
 *                    ONE launch through rq_plan_rounds / rq_plan_launch
 *   rq_cli replicate one AppendEntries batch through rq_append_check_dev with the
 *                    follower commit rule and the leader's commit round in the same
 *                    launch (its own input format, see replicate())
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

/* Input: G nf extra, then
 *   follower_term[nf][G] leader_term[G] prev_log_index[nf][G] prev_log_term[nf][G] n_new[nf][G]
 *   flog_off[nf*G+1] flog_term[flog_off[nf*G]]
 *   match[nf+extra][G] next_index[nf][G] follower_commit[nf][G] last_new_index[nf][G]
 *   commit[G] log_off[G+1] log_term[log_off[G]]
 * (nf follower slots, extra = 1 if the leader's own slot follows them in match; the leaders'
 * commit array is the RPCs' leaderCommit and is advanced in place by the commit round).
 * Output, one line each: follower_term (nf lines), success_mask, stepdown_mask, status,
 * match (nf + extra lines), next_index (nf), follower_commit (nf), commit, round status. */
static int replicate(void) {
    rq_engine *e = NULL;
    long long G_in, nf_in, extra_in;
    if (scanf("%lld %lld %lld", &G_in, &nf_in, &extra_in) != 3 || G_in < 1 || nf_in < 1 || extra_in < 0 || extra_in > 1 ||
        nf_in + extra_in > RQ_MAX_SLOTS) {
        fprintf(stderr, "rq_cli: bad header\n");
        return 2;
    }
    const uint32_t G = (uint32_t)G_in, nf = (uint32_t)nf_in, nr = nf + (uint32_t)extra_in;
    const size_t fp = (size_t)nf * G, rp = (size_t)nr * G;
    int32_t *ft = read_i32(fp), *lt = read_i32(G), *pi = read_i32(fp), *pt = read_i32(fp), *nn = read_i32(fp);
    uint32_t *foff = (uint32_t *)read_i32(fp + 1);
    const size_t farena = foff[fp];
    int32_t *fterm = read_i32(farena);
    int32_t *match = read_i32(rp), *next = read_i32(fp), *fc = read_i32(fp), *li = read_i32(fp);
    int32_t *commit = read_i32(G);
    uint32_t *log_off = (uint32_t *)read_i32((size_t)G + 1);
    const size_t arena = log_off[G];
    int32_t *log_term = read_i32(arena);
    uint16_t *ok = (uint16_t *)calloc(G, 2), *sd = (uint16_t *)calloc(G, 2);
    uint8_t *st = (uint8_t *)calloc(G, 1), *rst = (uint8_t *)calloc(G, 1);

    CHECK(rq_engine_create(0, 64, 1, 64, &e));          /* binds the device; every buffer below is ours */
    void *d_ft, *d_lt, *d_pi, *d_pt, *d_nn, *d_foff, *d_fterm, *d_match, *d_next, *d_fc, *d_li, *d_commit, *d_off, *d_log;
    void *d_ok, *d_sd, *d_st, *d_rst;
    CHECK(to_device(e, ft, fp * 4, &d_ft));       CHECK(to_device(e, lt, (size_t)G * 4, &d_lt));
    CHECK(to_device(e, pi, fp * 4, &d_pi));       CHECK(to_device(e, pt, fp * 4, &d_pt));
    CHECK(to_device(e, nn, fp * 4, &d_nn));       CHECK(to_device(e, foff, (fp + 1) * 4, &d_foff));
    CHECK(to_device(e, fterm, farena * 4, &d_fterm));
    CHECK(to_device(e, match, rp * 4, &d_match)); CHECK(to_device(e, next, fp * 4, &d_next));
    CHECK(to_device(e, fc, fp * 4, &d_fc));       CHECK(to_device(e, li, fp * 4, &d_li));
    CHECK(to_device(e, commit, (size_t)G * 4, &d_commit));
    CHECK(to_device(e, log_off, ((size_t)G + 1) * 4, &d_off));
    CHECK(to_device(e, log_term, arena * 4, &d_log));
    CHECK(to_device(e, ok, (size_t)G * 2, &d_ok)); CHECK(to_device(e, sd, (size_t)G * 2, &d_sd));
    CHECK(to_device(e, st, G, &d_st));             CHECK(to_device(e, rst, G, &d_rst));

    rq_round_desc r;                                    /* the commit round that would follow the batch */
    memset(&r, 0, sizeof r);
    r.struct_size = sizeof r; r.layout = RQ_LAYOUT_I32; r.what = RQ_ROUND_COMMIT;
    r.G = G; r.n = nr; r.ld = G;
    r.match = d_match;                                  /* the array the batch updates */
    r.commit = d_commit; r.commit_out = d_commit;       /* advanced in place */
    r.leader_term = d_lt; r.log_off = (const uint32_t *)d_off; r.log_term = d_log; r.log_terms = arena;
    r.status_out = (uint8_t *)d_rst;
    rq_append_desc a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a; a.layout = RQ_LAYOUT_I32;
    a.G = G; a.n = nf; a.ld = G; a.flog_terms = farena;
    a.follower_term = d_ft; a.leader_term = d_lt; a.prev_log_index = d_pi; a.prev_log_term = d_pt; a.n_new = d_nn;
    a.flog_off = (const uint32_t *)d_foff; a.flog_term = d_fterm;
    a.follower_term_out = d_ft;                         /* terms in place */
    a.success_mask = (uint16_t *)d_ok; a.stepdown_mask = (uint16_t *)d_sd; a.status_out = (uint8_t *)d_st;
    a.match_index_io = d_match; a.next_index_io = d_next;
    a.leader_commit = d_commit; a.last_new_index = d_li; a.follower_commit_io = d_fc;
    a.commit_round = &r;
    CHECK(rq_append_check_dev(e, NULL, &a));
    CHECK(rq_stream_sync(e, NULL));
    CHECK(rq_copy_from_device(e, ft, d_ft, fp * 4));       CHECK(rq_copy_from_device(e, ok, d_ok, (size_t)G * 2));
    CHECK(rq_copy_from_device(e, sd, d_sd, (size_t)G * 2)); CHECK(rq_copy_from_device(e, st, d_st, G));
    CHECK(rq_copy_from_device(e, match, d_match, rp * 4)); CHECK(rq_copy_from_device(e, next, d_next, fp * 4));
    CHECK(rq_copy_from_device(e, fc, d_fc, fp * 4));       CHECK(rq_copy_from_device(e, commit, d_commit, (size_t)G * 4));
    CHECK(rq_copy_from_device(e, rst, d_rst, G));
    void *all[] = {d_ft, d_lt, d_pi, d_pt, d_nn, d_foff, d_fterm, d_match, d_next, d_fc, d_li, d_commit, d_off, d_log,
                   d_ok, d_sd, d_st, d_rst};
    for (size_t i = 0; i < sizeof all / sizeof all[0]; i++) CHECK(rq_device_free(e, all[i]));
    for (uint32_t p = 0; p < nf; p++) print_i32(ft + (size_t)p * G, G);
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", ok[g]);
    printf("\n");
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", sd[g]);
    printf("\n");
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", st[g]);
    printf("\n");
    for (uint32_t p = 0; p < nr; p++) print_i32(match + (size_t)p * G, G);
    for (uint32_t p = 0; p < nf; p++) print_i32(next + (size_t)p * G, G);
    for (uint32_t p = 0; p < nf; p++) print_i32(fc + (size_t)p * G, G);
    print_i32(commit, G);
    for (uint32_t g = 0; g < G; g++) printf(g ? " %u" : "%u", rst[g]);
    printf("\n");
    CHECK(rq_engine_destroy(e));
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "replicate") == 0) return replicate();
    const int segmented = argc > 1 && strcmp(argv[1], "segmented") == 0;
    const int rounds = argc > 1 && strcmp(argv[1], "rounds") == 0;
    const int device_mode = segmented || rounds || (argc > 1 && strcmp(argv[1], "device") == 0);
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
    uint16_t *old_mask = NULL, *new_mask = NULL;
    if (segmented || rounds) {
        int32_t *om = read_i32(G), *nm = read_i32(G);
        old_mask = (uint16_t *)calloc(G, 2); new_mask = (uint16_t *)calloc(G, 2);
        for (uint32_t g = 0; g < G; g++) { old_mask[g] = (uint16_t)om[g]; new_mask[g] = (uint16_t)nm[g]; }
        free(om); free(nm);
    }

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
        void *d_om = NULL, *d_nm = NULL;
        if (old_mask) {
            CHECK(to_device(e, old_mask, (size_t)G * 2, &d_om));
            CHECK(to_device(e, new_mask, (size_t)G * 2, &d_nm));
            d.old_mask = (const uint16_t *)d_om; d.new_mask = (const uint16_t *)d_nm;
        }
        if (segmented) {
            /* a 64-group tile whose groups are all plain clusters of one size k (old == new ==
             * the k lowest slots) belongs to a segment of kind k; any other tile to one of kind
             * 0, decided through the masks; neighbouring tiles of one kind share a segment */
            uint32_t count = 0, prev = 99;
            for (uint32_t t = 0; t * 64 < G; t++) {
                uint32_t kind = 0;
                for (uint32_t g = t * 64; g < G && g < (t + 1) * 64; g++) {
                    uint32_t k = 0;
                    for (uint32_t c = 1; c <= 9; c++)
                        if (old_mask[g] == new_mask[g] && old_mask[g] == (1u << c) - 1u) k = c;
                    if (g == t * 64) kind = k;
                    else if (k != kind) kind = 0;
                    if (!k) kind = 0;
                }
                if (kind != prev) {
                    if (count == RQ_MAX_SEGMENTS) { fprintf(stderr, "rq_cli: groups are not in size order\n"); return 2; }
                    d.seg_first_group[count] = t * 64;
                    d.seg_n[count] = (uint8_t)kind;
                    count++;
                    prev = kind;
                }
            }
            d.seg_first_group[count] = G;
            d.n_segments = count;
            fprintf(stderr, "rq_cli: %u segments\n", count);
        }
        if (rounds) {
            /* two images of one state: the second starts `mid` groups into every array */
            const uint32_t mid = (G / 2) & ~63u;
            rq_round_desc two[2];
            two[0] = d; two[1] = d;
            uint32_t images = 1;
            if (mid) {
                images = 2;
                two[0].G = mid; two[1].G = G - mid;
#define SHIFT(field, type) two[1].field = (type *)two[1].field + mid
                SHIFT(match, const int32_t); SHIFT(commit, const int32_t); SHIFT(leader_term, const int32_t);
                SHIFT(log_off, const uint32_t);
                SHIFT(voter_term, const int32_t); SHIFT(voted_for, const int32_t); SHIFT(voter_last_log_term, const int32_t);
                SHIFT(cand_term, const int32_t); SHIFT(cand_id, const int32_t);
                SHIFT(old_mask, const uint16_t); SHIFT(new_mask, const uint16_t);
                SHIFT(commit_out, int32_t); SHIFT(voter_term_out, int32_t); SHIFT(max_term, int32_t);
                SHIFT(grant_mask, uint16_t); SHIFT(stepdown_mask, uint16_t);
                SHIFT(elected, uint8_t); SHIFT(status_out, uint8_t);
#undef SHIFT
            }
            rq_plan *plan = NULL;
            CHECK(rq_plan_rounds(e, two, images, &plan));
            CHECK(rq_plan_launch(plan, NULL));
            CHECK(rq_stream_sync(e, NULL));
            CHECK(rq_plan_destroy(plan));
        } else {
            CHECK(rq_round_dev(e, NULL, &d));
        }
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
        CHECK(rq_device_free(e, d_om));
        CHECK(rq_device_free(e, d_nm));
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
