// The prep record of ONE evolving session, written by a group of PREP_LANES = 8 lanes (lane `sub` of the group; all eight call together): vmis_prep_kernel's body, shared with
// the latency path's fused launch (vmis_fast_kernel<TINY>, srn_fast.hip), where the workgroup's first eight lanes write the record of its query themselves.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "srn_device.h"
#include "srn_kernels.h"

namespace srn {

static constexpr uint32_t PREP_LANES = 8;
__device__ __forceinline__ void prep_group(const DeviceIndex& ix, const uint64_t* __restrict__ items_flat, const uint32_t* __restrict__ q_off, uint32_t q, uint32_t sub, uint32_t m, uint32_t max_len,
                                           char* rec, const IdSlot* __restrict__ loc_table, uint32_t loc_mask, unsigned long long* __restrict__ okeys) {
    const uint32_t qb = q_off[q], L = q_off[q + 1] - qb;
    PrepItem* items = (PrepItem*)(rec + sizeof(PrepHead));
    uint32_t U = 0, rmax = 0, xlo = 0, sumw = 0, P = 0, nruns = 0, n_staged = 0, cur_attr = SRN_ATTR_NONE, my_run_start = 0;
    const bool ok = L != 0 && L <= max_len;
    const uint32_t rounds = ok ? (L + PREP_LANES - 1) / PREP_LANES : 0;
    uint32_t vmax = 0;          // largest FlatIndex::viol among this lane's known items (an index with incomplete lists only)
    uint32_t hot_key2 = kNone;  // ... and the second smallest (SRN_ORDER_KEY2: the order's secondary key)
    uint32_t hot_key = kNone;   // smallest dense idx (of the index the lists come from) among this lane's known items
    uint32_t idx = kNone, len = 0, pre = 0; unsigned long long base = 0;   // this lane's item of the LAST round (a session of <= 8 items has one round: its record is written once, with `kept`)
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t pos = r * PREP_LANES + sub;
        const bool have = pos < L;
        idx = kNone; len = 0; base = 0;
        uint32_t first_i = 0, head_rank = 0, mth_rank = 0;
        if (have) {
            const uint64_t raw = items_flat[qb + (L - 1 - pos)];   // pos 0 = most recent item
            bool first = true;
            for (uint32_t j = 0; j < pos; ++j) first = first && (items_flat[qb + (L - 1 - j)] != raw);   // Q2: most recent occurrence only
            { uint32_t hh = (uint32_t)dev_mix64(raw) & ix.id_mask;
              for (;;) { const IdSlot s = ix.id_table[hh]; if (s.idx == kNone) break; if (s.key == raw) { idx = s.idx; break; } hh = (hh + 1) & ix.id_mask; } }
            if (idx < hot_key) { hot_key2 = hot_key; hot_key = idx; } else if (idx != hot_key) hot_key2 = min(hot_key2, idx);
            if (ix.viol != nullptr && idx != kNone) vmax = max(vmax, ix.viol[idx]);
            first_i = first ? 1u : 0u;   // Q1: distinct raw ids, known or not
            if (first && idx != kNone) {
                const unsigned long long o0 = ix.post_off[idx], o1 = ix.post_off[idx + 1];
                len = (uint32_t)min((unsigned long long)m, o1 - o0); base = o0;
                if (len) { head_rank = ix.post_rank[o0]; if (len >= m) mth_rank = ix.post_rank[o0 + m - 1]; }
            }
            if (pos == 0 && idx != kNone) cur_attr = ix.meta[idx].attr;   // business rules look at the current item's attributes (mod.rs:162-182)
            if (loc_table) {   // (after everything that needs the whole index's idx)
                idx = kNone;
                uint32_t hh = (uint32_t)dev_mix64(raw) & loc_mask;
                for (;;) { const IdSlot s2 = loc_table[hh]; if (s2.idx == kNone) break; if (s2.key == raw) { idx = s2.idx; break; } hh = (hh + 1) & loc_mask; }
            }
        }
        // 8-lane inclusive scans of len and of (len != 0); 8-lane totals
        uint32_t sc_len = len, sc_run = len ? 1u : 0u;
        #pragma unroll
        for (uint32_t d = 1; d < PREP_LANES; d <<= 1) {
            const uint32_t a = __shfl_up(sc_len, d, PREP_LANES), b2 = __shfl_up(sc_run, d, PREP_LANES);
            if (sub >= d) { sc_len += a; sc_run += b2; }
        }
        uint32_t u = first_i, rm = head_rank, xl = mth_rank, sw = len ? L - pos : 0u;
        #pragma unroll
        for (uint32_t d = 1; d < PREP_LANES; d <<= 1) {
            u += __shfl_xor(u, d, PREP_LANES); sw += __shfl_xor(sw, d, PREP_LANES);
            rm = max(rm, __shfl_xor(rm, d, PREP_LANES)); xl = max(xl, __shfl_xor(xl, d, PREP_LANES));
        }
        pre = P + sc_len - len;
        const uint32_t run_idx = nruns + sc_run - (len ? 1u : 0u);   // which non-empty list of the session this one is
        #pragma unroll
        for (uint32_t s = 0; s < PREP_LANES; ++s) {   // lane j keeps run_start[j]
            const uint32_t rp = __shfl(pre, s, PREP_LANES), ri = __shfl(run_idx, s, PREP_LANES), rl = __shfl(len, s, PREP_LANES);
            if (rl != 0u && ri == sub) my_run_start = rp;
        }
        U += u; sumw += sw; rmax = max(rmax, rm); xlo = max(xlo, xl);
        P += __shfl(sc_len, PREP_LANES - 1, PREP_LANES); nruns += __shfl(sc_run, PREP_LANES - 1, PREP_LANES);
        if (have && rounds > 1) items[pos] = PrepItem{idx, len, pre, 0u, base};
    }
    // entries >= x_lo of every list (a prefix: the lists are sorted by rank, descending): what the merge-mode kernels stage
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t pos = r * PREP_LANES + sub;
        const bool have = pos < L;
        uint32_t kept = 0;
        if (have) {
            if (rounds > 1) { len = items[pos].len; base = items[pos].base; }
            if (len != 0) {
                const uint32_t* __restrict__ lst = ix.post_rank + base;
                uint32_t lo = 0, hi = len;   // first index whose entry is < x_lo
                if (xlo == 0u || lst[len - 1] >= xlo) lo = len;   // (the usual case: the whole list is kept -- one look instead of ~11 dependent ones)
                // (round 6) an 8-ary search: seven independent probes per step instead of one -- a list of 2 500 entries takes 4 dependent round trips instead of 12.  On the
                // latency path (one session, this chain in front of everything) that is up to ~4 us of a call; the batch prep kernel does not notice either way
                while (hi - lo > 8u) {
                    const uint32_t w = hi - lo;
                    uint32_t b[7], v[7];
#pragma unroll
                    for (uint32_t j = 0; j < 7u; ++j) { b[j] = lo + (uint32_t)(((unsigned long long)w * (j + 1u)) >> 3); v[j] = lst[b[j]]; }   // (lo < b[0] <= ... <= b[6] < hi)
                    uint32_t c = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 7u; ++j) c += v[j] >= xlo ? 1u : 0u;   // (descending list: the probes at or above x_lo are a prefix)
                    uint32_t nlo = lo, nhi = hi;
#pragma unroll
                    for (uint32_t j = 0; j < 7u; ++j) { if (c == j + 1u) nlo = b[j] + 1u; if (c == j) nhi = b[j]; }
                    lo = nlo; hi = nhi;
                }
                {   // <= 8 entries left: all at once
                    uint32_t v[8], c = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j) v[j] = lst[min(lo + j, len - 1u)];
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j) c += lo + j < hi && v[j] >= xlo ? 1u : 0u;
                    lo += c; hi = lo;
                }
                kept = lo;
            }
            if (rounds > 1) items[pos].kept = kept; else items[pos] = PrepItem{idx, len, pre, kept, base};
        }
        uint32_t ks = kept;
        #pragma unroll
        for (uint32_t d = 1; d < PREP_LANES; d <<= 1) ks += __shfl_xor(ks, d, PREP_LANES);
        n_staged += ks;
    }
    cur_attr = __shfl(cur_attr, 0, PREP_LANES);
    #pragma unroll
    for (uint32_t d = 1; d < PREP_LANES; d <<= 1) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)hot_key, d, PREP_LANES), o2 = (uint32_t)__shfl_xor((int)hot_key2, d, PREP_LANES);
        hot_key2 = o1 == hot_key ? min(hot_key2, o2) : min(max(hot_key, o1), min(hot_key2, o2));   // (the two smallest DISTINCT idx of the group)
        hot_key = min(hot_key, o1);
        vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, d, PREP_LANES));
    }
    // position sets are exact for this query iff no listed session at or above its cut holds one of its items without being in that item's list
    const uint32_t unsafe = vmax > xlo ? 1u : 0u;
#ifndef SRN_ORDER_KEY2
#define SRN_ORDER_KEY2 0   // (experiment, round 6) 8 more key bits: the SECOND most popular item of the session, in buckets of 16 popularity ranks -- fast kernel 21.48 -> 21.39 ms,
                           //  a third radix pass in exchange: nothing left of it in the step (profiles/r06_order_key2_ab.txt); off
#endif
    if (okeys && sub == 0) okeys[q] = SRN_ORDER_KEY2 ? ((unsigned long long)min(hot_key, 0xFFFFu) << 40) | ((unsigned long long)min(hot_key2 >> 4, 0xFFu) << 32) | q :
                                      ((unsigned long long)min(hot_key, 0xFFFFu) << 32) | q;   // (16 key bits: beyond the 65 535 most popular items there is nothing to group -- two radix passes instead of three)
    uint32_t* hw = (uint32_t*)rec;   // PrepHead, word by word: U rmax xlo sumw | P nruns L n_staged | run_start[8] | cur_attr unsafe
    if (sub == 0) {   // (records are 8-byte aligned: 72 + 24 * max_len)
        *(uint2*)hw = make_uint2(U, rmax); *(uint2*)(hw + 2) = make_uint2(xlo, sumw); *(uint2*)(hw + 4) = make_uint2(P, nruns); *(uint2*)(hw + 6) = make_uint2(L, n_staged);
        *(uint2*)(hw + 16) = make_uint2(cur_attr, unsafe);
    }
    hw[8 + sub] = sub < nruns ? my_run_start : 0u;
}

}  // namespace srn
