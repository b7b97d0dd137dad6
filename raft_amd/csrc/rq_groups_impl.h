// rq_groups_impl.h -- single-group traffic between a host and the resident
// state of a round: a scatter that overwrites chosen groups with packed rows and
// a gather that packs chosen groups' outputs.  A host that keeps a million groups
// in HBM touches a few thousand of them between two rounds (the groups whose Node
// objects took an RPC); moving only those rows keeps a step at the kernel's
// rate instead of the PCIe rate of the whole image.  Written once over the
// element type T like the round kernels.  Internal to librq.so.
#pragma once

#include "rq_round_impl.h"

namespace {

constexpr uint8_t kRowNoRoom = 1, kRowTermWidth = 2, kRowBadGroup = 4;

// v (already relative to the group's base term) as an arena entry of `bits`
// bits; false when it does not fit.
template <typename T>
__device__ __forceinline__ bool term_fits(uint32_t bits, T v) {
    if (bits == 8) return v >= 0 && v <= (T)0xFF;
    if (bits == 16) return v >= 0 && v <= (T)0xFFFF;
    if (bits == 32 && sizeof(T) == 8) return v >= (T)INT32_MIN && v <= (T)INT32_MAX;
    return true;
}
template <typename T>
__device__ __forceinline__ void store_term(void *arena, uint32_t bits, size_t idx, T v) {
    if (bits == 8) static_cast<uint8_t *>(arena)[idx] = (uint8_t)v;
    else if (bits == 16) static_cast<uint16_t *>(arena)[idx] = (uint16_t)v;
    else if (bits == 32 && sizeof(T) == 8) static_cast<int32_t *>(arena)[idx] = (int32_t)v;
    else static_cast<T *>(arena)[idx] = v;
}

// Words of one scatter row: 4 planes of n slots, then commit, leader_term,
// cand_term, cand_id, log_base, member masks (old | new << 16), and the arena
// offset of the group's log (used with a log_len plane: the slice may move).
// With log_term_start in the round's arguments there is no arena: the last word
// is the index where the leader's term starts, row_off still gives every row's
// log length (it goes to log_len) and no terms travel.
__host__ __device__ constexpr uint32_t scatter_row_words(uint32_t n) { return 4 * n + 7; }

// One wave per row.  The row is checked completely before anything is written,
// so a refused row leaves its group as it was.
template <typename T>
__global__ __launch_bounds__(64) void rq_groups_scatter_kernel(rq_round_args_t<T> a, uint32_t n, uint32_t count,
                                                               const uint32_t *gids, const T *rows,
                                                               const uint32_t *row_off, const T *row_terms,
                                                               uint8_t *row_status) {
    typedef typename std::make_unsigned<T>::type U;
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    if (r >= count) return;
    const uint32_t g = gids[r];
    const T *row = rows + (size_t)r * scatter_row_words(n);
    const group_addr ga = addr_of(a, g < a.G ? g : 0u);       // planes or 64-group tiles
    uint32_t bad = 0;
    if (g >= a.G) {
        bad = kRowBadGroup;
    } else if (a.log_term_start) {
        // nothing to check: no arena
    } else if (a.log_off) {
        const uint32_t len = row_off[r + 1] - row_off[r];
        bool room;
        if (a.log_len) {                  // the row names the slice
            room = span_ok((uint32_t)row[4 * n + 6], len, a.log_terms);
        } else {                          // tight CSR: the log keeps its length
            const uint32_t beg = *at(a.log_off, ga), end = *at(a.log_off, addr_of(a, g + 1));
            room = slice_ok(beg, end, a.log_terms) && len == end - beg;
        }
        if (!room) {
            bad = kRowNoRoom;
        } else {
            const T base = a.log_term_base ? *at(a.log_term_base, ga) : (T)0;
            bool fits = true;
            for (uint32_t i = lane; i < len; i += 64)
                fits = fits && term_fits<T>(a.log_term_bits, (T)((U)row_terms[row_off[r] + i] - (U)base));
            if (__ballot(!fits) != 0ull) bad = kRowTermWidth;
        }
    }
    if (row_status && lane == 0) row_status[r] = (uint8_t)bad;
    if (bad) return;
    T *planes[4] = {const_cast<T *>(a.match), const_cast<T *>(a.voter_term), const_cast<T *>(a.voted_for),
                    const_cast<T *>(a.last_log_term)};
    for (uint32_t i = lane; i < 4 * n; i += 64) {
        T *pl = planes[i / n];
        if (pl) *at(pl, ga, i % n) = row[i];
    }
    if (lane == 0) {
        const T *tail = row + 4 * n;
        if (a.commit) *at(const_cast<T *>(a.commit), ga) = tail[0];
        if (a.leader_term) *at(const_cast<T *>(a.leader_term), ga) = tail[1];
        if (a.cand_term) *at(const_cast<T *>(a.cand_term), ga) = tail[2];
        if (a.cand_id) *at(const_cast<T *>(a.cand_id), ga) = tail[3];
        if (a.log_base) *at(const_cast<T *>(a.log_base), ga) = tail[4];
        if (a.old_mask) {
            *at(const_cast<uint16_t *>(a.old_mask), ga) = (uint16_t)((uint64_t)tail[5] & 0xFFFFu);
            *at(const_cast<uint16_t *>(a.new_mask), ga) = (uint16_t)(((uint64_t)tail[5] >> 16) & 0xFFFFu);
        }
    }
    if (a.log_term_start) {
        if (lane == 0) {
            *at(const_cast<T *>(a.log_term_start), ga) = row[4 * n + 6];
            if (a.log_len) *at(const_cast<uint32_t *>(a.log_len), ga) = row_off[r + 1] - row_off[r];
        }
    } else if (a.log_off) {
        const uint32_t len = row_off[r + 1] - row_off[r];
        const uint32_t beg = a.log_len ? (uint32_t)row[4 * n + 6] : *at(a.log_off, ga);
        if (a.log_len && lane == 0) *at(const_cast<uint32_t *>(a.log_off), ga) = beg;
        const T base = a.log_term_base ? *at(a.log_term_base, ga) : (T)0;
        for (uint32_t i = lane; i < len; i += 64)
            store_term<T>(const_cast<void *>(a.log_term), a.log_term_bits, (size_t)beg + i,
                          (T)((U)row_terms[row_off[r] + i] - (U)base));
        if (a.log_len && lane == 0) *at(const_cast<uint32_t *>(a.log_len), ga) = len;
    }
}

// One thread per (row, word).  rows_out[r] = { commit, max_term, voter_term[n] };
// flags_out[r] = grant | stepdown << 16 | elected << 32 | status << 40.
template <typename T>
__global__ __launch_bounds__(256) void rq_groups_gather_kernel(rq_round_args_t<T> a, uint32_t n, uint32_t count,
                                                               const uint32_t *gids, T *rows_out,
                                                               uint64_t *flags_out) {
    const uint32_t words = n + 2;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)count * words) return;
    const uint32_t r = (uint32_t)(i / words), w = (uint32_t)(i % words);
    const uint32_t g = gids[r];
    if (g >= a.G) {                       // an unknown group reads as zeros with every flag bit set
        rows_out[i] = 0;
        if (w == 0) flags_out[r] = ~0ull;
        return;
    }
    // the outputs' own tile block, unless an output is its input updated in place
    const group_addr ga = addr_of(a, g);
    group_addr go = ga;
    if (a.tile_bytes) go.tile_off = (size_t)(g >> 6) * a.out_tile_bytes;
    const group_addr gco = (const void *)a.commit_out == (const void *)a.commit ? ga : go;
    const group_addr gvo = (const void *)a.voter_term_out == (const void *)a.voter_term ? ga : go;
    T v = 0;
    if (w == 0) v = a.commit_out ? *at(a.commit_out, gco) : (T)0;
    else if (w == 1) v = a.max_term ? *at(a.max_term, go) : (T)0;
    else v = a.voter_term_out ? *at(a.voter_term_out, gvo, w - 2) : (T)0;
    rows_out[i] = v;
    if (w == 0) {
        uint64_t f = 0;
        if (a.grant_mask) f |= (uint64_t)*at(a.grant_mask, go);
        if (a.stepdown_mask) f |= (uint64_t)*at(a.stepdown_mask, go) << 16;
        if (a.elected) f |= (uint64_t)*at(a.elected, go) << 32;
        if (a.status_out) f |= (uint64_t)*at(a.status_out, go) << 40;
        flags_out[r] = f;
    }
}

template <typename T>
hipError_t launch_groups_scatter_impl(const rq_round_args_t<T> &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                      const T *rows, const uint32_t *row_off, const T *row_terms,
                                      uint8_t *row_status, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (n < 1 || n > 16 || !gids || !rows) return hipErrorInvalidValue;
    if (a.log_term_start ? !row_off : (a.log_off && (!row_off || !row_terms))) return hipErrorInvalidValue;
    hipLaunchKernelGGL((rq_groups_scatter_kernel<T>), dim3(count), dim3(64), 0, stream, a, n, count, gids, rows,
                       row_off, row_terms, row_status);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_groups_gather_impl(const rq_round_args_t<T> &a, uint32_t n, uint32_t count, const uint32_t *gids,
                                     T *rows_out, uint64_t *flags_out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (n < 1 || n > 16 || !gids || !rows_out || !flags_out) return hipErrorInvalidValue;
    const uint64_t threads = (uint64_t)count * (n + 2);
    hipLaunchKernelGGL((rq_groups_gather_kernel<T>), dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream,
                       a, n, count, gids, rows_out, flags_out);
    return hipGetLastError();
}

}  // namespace
