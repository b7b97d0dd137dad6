/*
 * rq_bench.c -- the fused round timed from plain C through the C-ABI alone (no
 * Python, no torch, no HIP headers): device memory from rq_device_alloc, launches
 * through an rq_plan on the default stream, time from the host clock around
 * `steps` launches and one rq_stream_sync.
 *
 *   rq_bench [groups=1048576] [peers=5] [sets=12] [steps=400] [warmup=40] [term_bits=8] [tiles|planes]
 *
 * State lives in 64-group tiles (rq_round_desc.tile_bytes: one input and one output
 * block per set, laid out here) unless "planes" asks for one slot-major plane per array.
 *
 * The batch is drawn here with splitmix64 in the shape of the headline
 * configuration (log lengths U[1,16], matchIndex U[0,len), about half of the
 * groups with a quorum index above commitIndex, voter terms around the
 * candidate's); it is not the generator of the parity tests, this program only
 * measures.  `sets` input/output sets rotate so that neither reads nor writes
 * stay in the 256 MB Infinity Cache.  Prints one line:
 *   groups peers sets steps us_per_launch decisions_per_s algorithmic_GBps pct_of_8TBps
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rq.h"

static uint64_t sm_state;
static uint64_t splitmix(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t below(uint32_t k) { return (uint32_t)(splitmix() % k); }

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != RQ_OK) {                                                           \
            fprintf(stderr, "rq_bench: %s -> %s (%d)\n", #call, rq_strerror(rc_), rc_); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

struct set {
    void *match, *commit, *leader_term, *log_off, *log_term, *voter_term, *voted_for, *last_log_term;
    void *cand_term, *cand_id, *commit_out, *voter_term_out, *grant, *stepdown, *elected, *max_term, *status;
    void *in_block, *out_block;      /* tiled layout: everything but the arena lives in these two */
    rq_plan *plan;
};

/* Tiled layout (rq_round_desc.tile_bytes): rows of 64 elements, `slots` rows per array, arrays back
 * to back inside a tile.  Copies plane array src[slots][ld] (elem bytes each) into its place. */
static void tile_in(uint8_t *block, size_t tile_bytes, size_t field_off, const void *src, uint32_t slots,
                    size_t ld, size_t count, size_t elem) {
    for (uint32_t p = 0; p < slots; p++)
        for (size_t g = 0; g < count; g++)
            memcpy(block + (g / 64) * tile_bytes + field_off + ((size_t)p * 64 + g % 64) * elem,
                   (const uint8_t *)src + ((size_t)p * ld + g) * elem, elem);
}

static int upload(rq_engine *e, const void *src, size_t bytes, void **out) {
    int rc = rq_device_alloc(e, bytes, RQ_MEM_DEFAULT, out);
    return rc != RQ_OK ? rc : rq_copy_to_device(e, *out, src, bytes);
}

static int int_cmp(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

int main(int argc, char **argv) {
    const uint32_t G = argc > 1 ? (uint32_t)atol(argv[1]) : 1u << 20;
    const uint32_t n = argc > 2 ? (uint32_t)atol(argv[2]) : 5;
    const uint32_t S = argc > 3 ? (uint32_t)atol(argv[3]) : 12;
    const uint32_t steps = argc > 4 ? (uint32_t)atol(argv[4]) : 400;
    const uint32_t warmup = argc > 5 ? (uint32_t)atol(argv[5]) : 40;
    const uint32_t term_bits = argc > 6 ? (uint32_t)atol(argv[6]) : 8;   /* log-term arena entry width */
    const int tiled = !(argc > 7 && strcmp(argv[7], "planes") == 0);     /* 64-group tiles unless "planes" */
    if (G < 1 || n < 1 || n > 9 || S < 1 || S > 64 || steps < 1 || (term_bits != 8 && term_bits != 16 && term_bits != 32)) {
        fprintf(stderr, "usage: rq_bench [groups] [peers 1..9] [sets 1..64] [steps] [warmup] [term bits 8|16|32] [tiles|planes]\n");
        return 2;
    }
    const uint32_t ld = G + 256;                 /* keeps the planes off one power-of-two stride */
    const size_t plane = (size_t)n * ld;
    rq_engine *e = NULL;
    CHECK(rq_engine_create(0, 64, 1, 64, &e));   /* the engine only binds the device here */

    int32_t *match = calloc(plane, 4), *vt = calloc(plane, 4), *vf = calloc(plane, 4), *ll = calloc(plane, 4);
    int32_t *commit = calloc(G, 4), *lterm = calloc(G, 4), *cterm = calloc(G, 4), *cid = calloc(G, 4);
    uint32_t *off = calloc((size_t)G + 1, 4);
    int32_t *arena = calloc((size_t)G * 16 + 1, 4);
    uint8_t *narrow8 = calloc((size_t)G * 16 + 1, 1);
    uint16_t *narrow16 = calloc((size_t)G * 16 + 1, 2);
    struct set *sets = calloc(S, sizeof *sets);
    /* tiled layout: (4n + 5) input rows and (n + 2) 4-byte, two 2-byte and two 1-byte output rows per tile */
    const size_t ntiles = ((size_t)G + 1 + 63) / 64;
    const size_t in_tile = (size_t)(4 * n + 5) * 256, out_tile = ((size_t)(n + 2) * 256 + 2 * 128 + 2 * 64 + 255) / 256 * 256;
    uint8_t *in_host = tiled ? calloc(ntiles, in_tile) : NULL;
    if (tiled && !in_host) return 3;
    if (!match || !vt || !vf || !ll || !commit || !lterm || !cterm || !cid || !off || !arena || !sets || !narrow8 || !narrow16) return 3;

    for (uint32_t s = 0; s < S; s++) {
        sm_state = 0x5EED0001ull + 7919ull * s;
        size_t total = 0;
        for (uint32_t g = 0; g < G; g++) {
            const uint32_t len = 1 + below(16), T = 1 + below(8);
            off[g] = (uint32_t)total;
            for (uint32_t i = 0; i < len; i++) arena[total + i] = 1 + (int32_t)(i * T / len);
            total += len;
            int32_t m[9];
            for (uint32_t p = 0; p < n; p++) match[(size_t)p * ld + g] = m[p] = (int32_t)below(len);
            qsort(m, n, sizeof m[0], int_cmp);
            const int32_t N = m[(n - 1) / 2];
            commit[g] = below(2) ? (N > 0 ? (int32_t)below((uint32_t)N) : 0) : N + (int32_t)below(3);
            lterm[g] = below(10) < 7 ? arena[off[g] + (uint32_t)N] : (int32_t)T + 1;
            const int32_t t = 2 + (int32_t)below(6);
            cterm[g] = t;
            cid[g] = 0;
            for (uint32_t p = 0; p < n; p++) {
                const size_t i = (size_t)p * ld + g;
                const uint32_t u = below(10), w = below(4);
                vt[i] = p == 0 ? t : (u < 6 ? t - 1 : (u < 9 ? t : t + 1));
                vf[i] = p == 0 ? 0 : (w < 2 ? -1 : (w == 2 ? 0 : 1 + (int32_t)below(n)));
                ll[i] = (int32_t)below((uint32_t)t + 2);
            }
        }
        off[G] = (uint32_t)total;
        struct set *d = &sets[s];
        /* the arena as a host keeping it resident stores it: terms here are 1..9, one byte each is enough */
        if (term_bits == 32) {
            CHECK(upload(e, arena, (total + 1) * 4, &d->log_term));
        } else {
            for (size_t i = 0; i < total; i++) {
                if (term_bits == 8) narrow8[i] = (uint8_t)arena[i]; else narrow16[i] = (uint16_t)arena[i];
            }
            CHECK(upload(e, term_bits == 8 ? (void *)narrow8 : (void *)narrow16, (total + 1) * (term_bits / 8), &d->log_term));
        }
        if (tiled) {
            size_t f = 0;       /* field offsets inside a tile, in the order below */
            const size_t f_match = f; f += (size_t)n * 256;
            const size_t f_vt = f; f += (size_t)n * 256;
            const size_t f_vf = f; f += (size_t)n * 256;
            const size_t f_ll = f; f += (size_t)n * 256;
            const size_t f_commit = f; f += 256;
            const size_t f_lterm = f; f += 256;
            const size_t f_cterm = f; f += 256;
            const size_t f_cid = f; f += 256;
            const size_t f_off = f; f += 256;
            memset(in_host, 0, ntiles * in_tile);
            tile_in(in_host, in_tile, f_match, match, n, ld, G, 4);
            tile_in(in_host, in_tile, f_vt, vt, n, ld, G, 4);
            tile_in(in_host, in_tile, f_vf, vf, n, ld, G, 4);
            tile_in(in_host, in_tile, f_ll, ll, n, ld, G, 4);
            tile_in(in_host, in_tile, f_commit, commit, 1, G, G, 4);
            tile_in(in_host, in_tile, f_lterm, lterm, 1, G, G, 4);
            tile_in(in_host, in_tile, f_cterm, cterm, 1, G, G, 4);
            tile_in(in_host, in_tile, f_cid, cid, 1, G, G, 4);
            tile_in(in_host, in_tile, f_off, off, 1, (size_t)G + 1, (size_t)G + 1, 4);
            CHECK(upload(e, in_host, ntiles * in_tile, &d->in_block));
            CHECK(rq_device_alloc(e, ntiles * out_tile, RQ_MEM_DEFAULT, &d->out_block));
            uint8_t *ib = d->in_block, *ob = d->out_block;
            d->match = ib + f_match; d->voter_term = ib + f_vt; d->voted_for = ib + f_vf; d->last_log_term = ib + f_ll;
            d->commit = ib + f_commit; d->leader_term = ib + f_lterm; d->cand_term = ib + f_cterm; d->cand_id = ib + f_cid;
            d->log_off = ib + f_off;
            size_t g = 0;       /* output fields, widest elements first */
            d->voter_term_out = ob + g; g += (size_t)n * 256;
            d->commit_out = ob + g; g += 256;
            d->max_term = ob + g; g += 256;
            d->grant = ob + g; g += 128;
            d->stepdown = ob + g; g += 128;
            d->elected = ob + g; g += 64;
            d->status = ob + g; g += 64;
        } else {
        CHECK(upload(e, match, plane * 4, &d->match));
        CHECK(upload(e, commit, (size_t)G * 4, &d->commit));
        CHECK(upload(e, lterm, (size_t)G * 4, &d->leader_term));
        CHECK(upload(e, off, ((size_t)G + 1) * 4, &d->log_off));
        CHECK(upload(e, vt, plane * 4, &d->voter_term));
        CHECK(upload(e, vf, plane * 4, &d->voted_for));
        CHECK(upload(e, ll, plane * 4, &d->last_log_term));
        CHECK(upload(e, cterm, (size_t)G * 4, &d->cand_term));
        CHECK(upload(e, cid, (size_t)G * 4, &d->cand_id));
        CHECK(rq_device_alloc(e, (size_t)G * 4, RQ_MEM_DEFAULT, &d->commit_out));
        CHECK(rq_device_alloc(e, plane * 4, RQ_MEM_DEFAULT, &d->voter_term_out));
        CHECK(rq_device_alloc(e, (size_t)G * 2, RQ_MEM_DEFAULT, &d->grant));
        CHECK(rq_device_alloc(e, (size_t)G * 2, RQ_MEM_DEFAULT, &d->stepdown));
        CHECK(rq_device_alloc(e, G, RQ_MEM_DEFAULT, &d->elected));
        CHECK(rq_device_alloc(e, (size_t)G * 4, RQ_MEM_DEFAULT, &d->max_term));
        CHECK(rq_device_alloc(e, G, RQ_MEM_DEFAULT, &d->status));
        }
        rq_round_desc r;
        memset(&r, 0, sizeof r);
        r.struct_size = sizeof r;
        r.layout = RQ_LAYOUT_I32;
        r.what = RQ_ROUND_COMMIT | RQ_ROUND_VOTE;
        r.G = G; r.n = n; r.ld = ld;
        if (tiled) { r.tile_bytes = in_tile; r.out_tile_bytes = out_tile; }
        r.log_terms = total; r.log_term_bits = term_bits;
        r.match = d->match; r.commit = d->commit; r.leader_term = d->leader_term;
        r.log_off = d->log_off; r.log_term = d->log_term;
        r.voter_term = d->voter_term; r.voted_for = d->voted_for; r.voter_last_log_term = d->last_log_term;
        r.cand_term = d->cand_term; r.cand_id = d->cand_id;
        r.commit_out = d->commit_out; r.voter_term_out = d->voter_term_out;
        r.grant_mask = d->grant; r.stepdown_mask = d->stepdown; r.elected = d->elected;
        r.max_term = d->max_term; r.status_out = d->status;
        CHECK(rq_plan_round(e, &r, &d->plan));
    }
    /* one untimed rotation over every set, then the warm-up */
    for (uint32_t i = 0; i < S; i++) CHECK(rq_plan_launch(sets[i].plan, NULL));
    for (uint32_t i = 0; i < warmup; i++) CHECK(rq_plan_launch(sets[i % S].plan, NULL));
    CHECK(rq_stream_sync(e, NULL));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < steps; i++) CHECK(rq_plan_launch(sets[i % S].plan, NULL));
    CHECK(rq_stream_sync(e, NULL));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    const double us = sec / steps * 1e6;
    const double alg = (double)(20 * n + 37) * G;      /* commit 4n+20, vote 16n+17 bytes per group */

    /* a decision count as a plausibility check of the drawn batch */
    uint8_t *el = malloc(tiled ? ntiles * out_tile : G);
    size_t elected = 0;
    if (tiled) {
        CHECK(rq_copy_from_device(e, el, sets[0].out_block, ntiles * out_tile));
        const size_t f_el = (size_t)((uint8_t *)sets[0].elected - (uint8_t *)sets[0].out_block);
        for (uint32_t g = 0; g < G; g++) elected += el[(size_t)(g / 64) * out_tile + f_el + g % 64];
    } else {
        CHECK(rq_copy_from_device(e, el, sets[0].elected, G));
        for (uint32_t g = 0; g < G; g++) elected += el[g];
    }
    printf("%u %u %u %u %.3f %.4g %.1f %.2f\n", G, n, S, steps, us, (double)G / (us * 1e-6),
           alg / (us * 1e-6) / 1e9, 100.0 * alg / (us * 1e-6) / 8e12);
    fprintf(stderr, "rq_bench: set 0 elected %.1f%% of its groups\n", 100.0 * (double)elected / G);
    for (uint32_t s = 0; s < S; s++) {
        struct set *d = &sets[s];
        CHECK(rq_plan_destroy(d->plan));
        if (tiled) {
            CHECK(rq_device_free(e, d->in_block)); CHECK(rq_device_free(e, d->out_block)); CHECK(rq_device_free(e, d->log_term));
            continue;
        }
        void *all[] = {d->match, d->commit, d->leader_term, d->log_off, d->log_term, d->voter_term, d->voted_for,
                       d->last_log_term, d->cand_term, d->cand_id, d->commit_out, d->voter_term_out, d->grant,
                       d->stepdown, d->elected, d->max_term, d->status};
        for (size_t i = 0; i < sizeof all / sizeof all[0]; i++) CHECK(rq_device_free(e, all[i]));
    }
    CHECK(rq_engine_destroy(e));
    free(el);
    return 0;
}
