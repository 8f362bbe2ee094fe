// bc_match_pairs.hip — the 3-mismatch matcher (gfx950 only): a pigeonhole index on PAIRS of pieces.
//
// The d + 1 = 4 segment index of bc_match.hip keys every table on one 5-base segment of a 20-mer: ~G/1024 guides
// per bucket, and a read that matches nothing closer pays for all four buckets (392 candidates at 1e5 guides: the
// quad kernels spend their whole 10 ms per 1e8 reads on the tenth of the reads that reaches this far).  Cut the
// k-mer into FIVE pieces instead: a guide within three substitutions leaves at least TWO pieces clean, so it shares
// one of the C(5,2) = 10 piece pairs with the read.  Table (a, b) is keyed on piece a and the first bases of piece
// b — about log4(G / 6) bases in all: 16 384 buckets of 6 guides at G = 1e5 — and a read visits ten 64-byte slabs
// of fifteen 4-byte rests (61 candidates) instead of four chains of seven 128-byte slabs.
//
// A guide with more than two clean pieces sits in several of the read's buckets; it is counted in ONE: the table of
// the lexicographically first pair of its clean pieces (the entry's mismatch mask tells which pieces are clean), so
// n_best needs no candidate list.  The lookup is exact for any read and any budget <= 3 — best distance, lowest
// index, tie count — and is used as the second stage of the cascades (launch_cascade in bc_match.hip), where only
// the reads the first pass left without a hit arrive.  One lane per read for the arithmetic, one QUAD per slab for
// the loads (pairs_lookup_tables).  Large launches run it as passes over groups of tables from a list of the
// missed reads that the first stage (k_match_wq, LIST form) writes itself; small ones as one kernel that finds
// the misses in best_idx (k_cascade_stage2).  Round 3 at 1e5 guides, 1e8 reads: 1.81 ms (one kernel, a lane
// fetching its own slab) -> 1.06 ms (list 0.08 in the first stage + pack 0.08 + four passes 0.92).
//
// Geometry limits: an entry's rest holds at most 16 bases per plane, so key >= k - 16 bases; the key is
// capped at 9 bases (10 tables x 262 144 slabs).  k from 8 to 25; other lengths keep the four-segment index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "bc_match_dev.h"

namespace bc {

BC_HD static inline uint32_t pair_drop(uint32_t v, uint32_t start, uint32_t len) {  // remove the bit field [start, start + len)
    const uint32_t below = (1u << start) - 1u;
    const uint32_t sh = start + len;
    return (v & below) | ((sh >= 32u ? 0u : (v >> sh)) << start);
}
BC_HD static inline uint32_t pair_rest_plane(const PairTable& T, uint32_t plane) {
    return pair_drop(pair_drop(plane, T.b_start, T.b_len), T.a_start, T.a_len);  // b lies above a
}
// the same with the table's precomputed masks (three and-s, two shifts, two or-s: the device's form)
BC_HD static inline uint32_t pair_rest_masks(const PairTable& T, uint32_t plane) {
    return (plane & T.m_low) | ((plane >> T.a_len) & T.m_mid) | ((plane >> (T.a_len + T.b_len)) & T.m_high);
}
BC_HD static inline uint32_t pair_bucket(const PairTable& T, uint32_t lo, uint32_t hi) {
    const uint32_t ma = (1u << T.a_len) - 1u, mb = (1u << T.b_len) - 1u, kb = T.a_len + T.b_len;
    const uint32_t klo = ((lo >> T.a_start) & ma) | (((lo >> T.b_start) & mb) << T.a_len);
    const uint32_t khi = ((hi >> T.a_start) & ma) | (((hi >> T.b_start) & mb) << T.a_len);
    return klo | (khi << kb);
}

}  // namespace bc

namespace {

using namespace bc;

// one read through the ten tables (lane = read): best distance, tie count, lowest index.  Exact for any read and
// any budget <= 3 (see the header of this file).
struct PairBest { uint32_t mind, cnt, bidx; };

// Slab geometry of the pair tables: 64-byte slabs of fifteen rests + tail (about 6 guides per bucket).  Round 4 built the
// form round 3's notes asked for — 32-byte slabs of seven rests under a key one base longer (1.5 guides per bucket at 1e5:
// a visit scans 8 slots instead of 16, a quad fetches two mates' slabs per turn, two turns instead of four) — and
// measured it SLOWER at every library size: 45.3 against 47.3 G reads/s at 1e5 guides (best grouping), 22.6 / 23.9 at
// 3e5, 8.3 / 10.3 at 1e6 (profiles/r04_d3_slab32_ab.txt): halving the scan doubles the tables (2.1 MB each), and the
// passes' time follows what a group keeps in the L2, not the instruction count.  Kept as the compile-time A/B
// (-DBC_PAIR_SLAB32=1); tests/test_match_gpu.py passes in both forms.
#ifndef BC_PAIR_SLAB32
#define BC_PAIR_SLAB32 0
#endif
constexpr uint32_t PSW = BC_PAIR_SLAB32 ? 8u : 16u;   // words per slab (the last one: {next << 4 | count})

__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t one, uint32_t zero) { return (m & one) | (~m & zero); }
template <int J>
__device__ __forceinline__ uint32_t qb(uint32_t v) { return quad_perm<J * 0x55>(v); }  // lane J of the quad, to the whole quad

// four entries of quad-mate J's slab against its rest: flag bits [4 J, 4 J + 4)
// flag bit i = entry i lies within the budget.  popcount + nb (nb = ~max_mm = -max_mm - 1) is negative exactly then,
// and v_alignbit shifts its sign bit in: five instructions per entry (xor with the DPP operand, shift, and-or,
// bit count with the addend, align)
template <int J>
__device__ __forceinline__ uint32_t scan4(const u32x4& w, uint32_t rest, uint32_t nb) {
    const uint32_t rj = qb<J>(rest);
    const uint32_t e[4] = {w.x, w.y, w.z, w.w};
    uint32_t f = 0;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const uint32_t x = rj ^ e[i];
        const uint32_t mm = (x | (x >> 16)) & 0xFFFFu;
        f = __builtin_amdgcn_alignbit(f, (uint32_t)__popc(mm) + nb, 31);  // (f << 1) | sign
    }
    return f;  // four bits
}

// the same against a rest that is already the right mate's (32-byte slabs: the mate depends on the lane)
__device__ __forceinline__ uint32_t scan4r(const u32x4& w, uint32_t rj, uint32_t nb) {
    const uint32_t e[4] = {w.x, w.y, w.z, w.w};
    uint32_t f = 0;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const uint32_t x = rj ^ e[i];
        const uint32_t mm = (x | (x >> 16)) & 0xFFFFu;
        f = __builtin_amdgcn_alignbit(f, (uint32_t)__popc(mm) + nb, 31);  // (f << 1) | sign
    }
    return f;  // four bits
}

// One read per lane through tables [t0, t1) — EVERY lane of the wave calls this (valid = the lane has a read): the
// arithmetic is per lane, the loads are per quad.  A lane that fetches its own 64-byte slab with four 16-byte loads
// costs the vector L1 four tag look-ups per slab, and that rate (about one look-up per clock and CU), not the L2, not
// the instruction count, set the pace: 1.3 ms per 1e8 slabs whether the tables sat in the L2 or in the Infinity Cache
// and whether the scan took 250 instructions behind 100 branches or 110 in a straight line.  Here the four lanes of a
// quad fetch the slab of quad-mate 0, 1, 2, 3 in turn, 16 bytes each — one look-up per slab — and scan their quarter
// against that mate's rest (two DPP moves per turn); a lane gets the candidate bits of its own slab back with four
// more.  What decides whether a candidate counts runs per candidate (a lane has one for a slab in thirteen).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// hook: called once, right after the first slab's loads are requested — where a caller asks for what it needs NEXT
// (the memory counter is in-order: a request made before the slab's would be waited for together with it)
template <typename Hook = NoHook>
__device__ __forceinline__ void pairs_lookup_tables(const PairParams& pp, int t0, int t1, bool valid, uint32_t rlo, uint32_t rhi,
                                                    uint32_t max_mm, PairBest& b, Hook hook = Hook()) {
    const uint32_t q = threadIdx.x & 3u;
    bool hooked = false;
    for (int t = t0; t < t1; ++t) {
        const PairTable T = pp.t[t];  // (by value: every field in one batch of scalar loads — by reference the candidate path loaded them one by one, a wait each)
        const uint32_t rest = pair_rest_masks(T, rlo) | (pair_rest_masks(T, rhi) << 16);
        uint32_t slab = pair_bucket(T, rlo, rhi);  // (a lane without a read: bucket of the all-A read, a valid address)
        bool live = valid;
        for (;;) {  // uniform over the wave: while any lane has a slab to visit
            // (global, not flat, loads: the tables' pointers come from memory and the compiler cannot tell; a flat load
            // makes every wait a wait for everything in flight, the next round's list entry included)
            typedef const u32x4 __attribute__((address_space(1)))* slab_ptr;
            typedef const uint32_t __attribute__((address_space(1)))* word_ptr;
            uint32_t cand, tail;
            if constexpr (PSW == 8u) {
                // 32-byte slabs: two mates' slabs per turn — in turn 0 lanes 0, 1 take the halves of mate 0's slab and
                // lanes 2, 3 those of mate 1's, in turn 1 mates 2 and 3 (quad_perm [0,0,1,1] / [2,2,3,3]); one tag
                // look-up per 64 bytes requested when the two slabs happen to share a line, two otherwise
                const slab_ptr base = (slab_ptr)(uintptr_t)T.rests + (q & 1u);
                const u32x4 w0 = base[(size_t)quad_perm<0x50>(slab) * 2u], w1 = base[(size_t)quad_perm<0xFA>(slab) * 2u];
                if (!hooked) { hook(); hooked = true; }
                const uint32_t fl = scan4r(w0, quad_perm<0x50>(rest), ~max_mm) | (scan4r(w1, quad_perm<0xFA>(rest), ~max_mm) << 4);
                // this lane's slab came in turn q >> 1, its low half on lane 2 (q & 1), its high half on the lane after
                const uint32_t sh = 4u * (q >> 1);
                cand = ((quad_perm<0x88>(fl) >> sh) & 15u) | (((quad_perm<0xDD>(fl) >> sh) & 15u) << 4);
                // the slab's last word {next << 4 | count} is the high half's fourth
                const uint32_t x0 = quad_perm<0xDD>(w0.w), x1 = quad_perm<0xDD>(w1.w);
                tail = (q >> 1) ? x1 : x0;
            } else {
                const slab_ptr base = (slab_ptr)(uintptr_t)T.rests + q;
                const u32x4 w0 = base[(size_t)qb<0>(slab) * 4u], w1 = base[(size_t)qb<1>(slab) * 4u];
                const u32x4 w2 = base[(size_t)qb<2>(slab) * 4u], w3 = base[(size_t)qb<3>(slab) * 4u];
                if (!hooked) { hook(); hooked = true; }
                const uint32_t fl = scan4<0>(w0, rest, ~max_mm) | (scan4<1>(w1, rest, ~max_mm) << 4) | (scan4<2>(w2, rest, ~max_mm) << 8) |
                                    (scan4<3>(w3, rest, ~max_mm) << 12);
                // bits [4 q, 4 q + 4) of every quad lane's flags belong to this lane's slab: entries 4 Q ... 4 Q + 3 from lane Q
                const uint32_t sh = 4u * q;
                cand = ((qb<0>(fl) >> sh) & 15u) | (((qb<1>(fl) >> sh) & 15u) << 4) | (((qb<2>(fl) >> sh) & 15u) << 8) |
                       (((qb<3>(fl) >> sh) & 15u) << 12);
                // the slab's last word {next << 4 | count} arrived on quad lane 3
                const uint32_t x0 = qb<3>(w0.w), x1 = qb<3>(w1.w), x2 = qb<3>(w2.w), x3 = qb<3>(w3.w);
                const uint32_t q0 = 0u - (q & 1u), q1 = 0u - (q >> 1);
                tail = bit_select(q1, bit_select(q0, x3, x2), bit_select(q0, x1, x0));
            }
            const uint32_t cnt = tail & 15u, next = tail >> 4;
            cand &= (1u << cnt) - 1u;  // (count <= PSW - 1: the last word itself never counts)
            if (!live) cand = 0u;
            while (cand) {
                const uint32_t e = (uint32_t)__ffs((int)cand) - 1u;
                cand &= cand - 1u;
                const uint32_t x = rest ^ ((word_ptr)(uintptr_t)T.rests)[(size_t)slab * PSW + e];  // (the quad has just brought the line into the L1)
                const uint32_t mm = (x | (x >> 16)) & 0xFFFFu;
                // counted here only if this table's pair is the first pair of the guide's clean pieces: what is left
                // of pieces a and b is clean, every earlier piece is dirty
                bool ok = (mm & T.must_clean) == 0u;
#pragma unroll
                for (int c = 0; c < 3; ++c) ok = ok && (c >= (int)T.n_other || (mm & T.other[c]) != 0u);
                if (ok) {  // a guide within the budget — its index comes from the parallel array
                    const uint32_t d = (uint32_t)__popc(mm);
                    const uint32_t g = ((word_ptr)(uintptr_t)T.gidx)[(size_t)slab * PSW + e];
                    if (d < b.mind) { b.mind = d; b.cnt = 1u; b.bidx = g; }
                    else if (d == b.mind) { ++b.cnt; b.bidx = min(b.bidx, g); }
                }
            }
            live = live && next != 0u;
            if (!__any(live)) break;
            if (live) slab = next;  // (the others fetch their last slab again and ignore it)
        }
    }
}
__device__ __forceinline__ PairBest pairs_lookup(const PairParams& pp, bool valid, uint32_t rlo, uint32_t rhi, uint32_t max_mm) {
    PairBest b{0xFFu, 0u, 0xFFFFFFFFu};
    pairs_lookup_tables(pp, 0, pp.ntab, valid, rlo, rhi, max_mm, b);
    return b;
}

// ---- finding the reads the first stage left without a hit -----------------------------------------------------------
// A workgroup takes best_idx in sub-chunks of 4 096 results (sixteen per thread, 16-byte loads).  A thread's misses are
// sixteen flag bits; ballots turn them into queue slots (no per-thread position list: that form lived in scratch
// memory and took 0.36 ms per 1e8 results for a walk the HBM can feed in 0.07).
constexpr int ST2_SUB = 4096;
__device__ __forceinline__ uint32_t miss_flags(const int32_t* __restrict__ best_idx, uint64_t c0, uint64_t n) {
    const u32x4* pv = reinterpret_cast<const u32x4*>(best_idx);  // (16-byte aligned: the cascade's precondition)
    uint32_t fm = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t i = c0 + ((uint64_t)q * BLOCK + threadIdx.x) * 4;
        uint32_t e[4] = {0u, 0u, 0u, 0u};
        if (i + 3 < n) {
            const u32x4 v = pv[i >> 2];
            e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = (i + c < n) ? (uint32_t)best_idx[i + c] : 0u;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) fm |= (uint32_t)(e[c] == (uint32_t)BC_IDX_MISS) << (q * 4 + c);
    }
    return fm;
}
__device__ __forceinline__ uint32_t miss_count_wave(uint32_t fm) {  // uniform over the wave
    uint32_t cnt = 0;
#pragma unroll
    for (int bit = 0; bit < 16; ++bit) cnt += (uint32_t)__popcll(__ballot((fm >> bit) & 1u));
    return cnt;
}
__device__ __forceinline__ void miss_queue_wave(uint32_t fm, uint64_t c0, uint32_t* __restrict__ q, uint32_t at) {
#pragma unroll
    for (int bit = 0; bit < 16; ++bit) {
        const bool f = (fm >> bit) & 1u;
        const unsigned long long m = __ballot(f);
        if (f) q[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                   (uint32_t)(c0 + ((uint64_t)(bit >> 2) * BLOCK + threadIdx.x) * 4 + (uint32_t)(bit & 3));
        at += (uint32_t)__popcll(m);
    }
}
// queue the sub-chunk's misses at q[qn ...) (wave after wave); returns their number.  Two block barriers inside.
__device__ __forceinline__ uint32_t miss_queue_block(const int32_t* __restrict__ best_idx, uint64_t c0, uint64_t n, uint32_t* __restrict__ q,
                                                     uint32_t qn, uint32_t* __restrict__ w_sum) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t fm = miss_flags(best_idx, c0, n);
    const uint32_t mine = miss_count_wave(fm);
    __syncthreads();  // (whoever read the queue or w_sum before is done)
    if (lane == 0u) w_sum[wv] = mine;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { before += (uint32_t)w < wv ? w_sum[w] : 0u; total += w_sum[w]; }
    miss_queue_wave(fm, c0, q, qn + before);
    return total;
}

// Second stage of the d = 3 cascade, fused with the search for its input: a workgroup walks its share of best_idx in
// sub-chunks of 4 096 results (sixteen per thread, 16-byte loads), queues the positions the d <= 2 pass left as
// BC_IDX_MISS in LDS, and whenever 256 are queued every thread takes one through the pair index and writes the
// result IN PLACE; hits are tallied with device atomics (5 % of the benchmark's reads).  No compacted copy of the
// missed reads, no device-side count for the host to wait for, no scatter pass: the launch is asynchronous and needs
// no workspace — the form for small launches (large ones: the list form below).  (Round 2: k_collect_missed 0.40 ms +
// k_match_pairs 1.49 ms + k_scatter_results 0.26 ms per 1e8 reads and a host synchronisation in between.  Measured
// and dropped: a quad-per-read form of the lookup with the key arithmetic in all four lanes, 1.84 ms against 1.49;
// the primary slabs of five tables requested before the first is scanned (144 VGPRs, three waves per SIMD): 2.10
// against 1.67 ms.  Of this kernel's 1.55 ms per 1e8 reads, 0.36 is the walk: best_idx and, for the misses, most
// lines of the read array again.)
__global__ __launch_bounds__(BLOCK) void k_cascade_stage2(const PairParams* __restrict__ ppp, const uint64_t* __restrict__ reads,
                                                         uint64_t n, uint64_t kmask, int max_mm_, int32_t* __restrict__ best_idx,
                                                         uint8_t* __restrict__ best_dist, uint8_t* __restrict__ n_best,
                                                         uint32_t* __restrict__ counts, uint64_t* __restrict__ first_seen,
                                                         uint64_t base) {
    __shared__ uint32_t q_pos[ST2_SUB + BLOCK];
    __shared__ uint32_t w_sum[BLOCK / 64];
    const PairParams& pp = *ppp;
    const uint32_t max_mm = (uint32_t)max_mm_;
    uint32_t qn = 0;  // queued positions (uniform over the block)
    auto serve = [&](uint32_t first, uint32_t count) {  // queue entries [first, first + count), one per thread
        const bool valid = threadIdx.x < count;  // (whole waves go through the lookup: its loads are per quad)
        const uint32_t i = valid ? q_pos[first + threadIdx.x] : 0u;
        const uint64_t r = valid ? reads[i] & kmask : 0ull;
        const PairBest b = pairs_lookup(pp, valid, (uint32_t)r, (uint32_t)(r >> 32), max_mm);
        {
            if (valid && b.mind <= max_mm) {
                const uint32_t g = b.bidx & SLAB_GIDX_MASK;
                best_idx[i] = (int32_t)g;
                if (best_dist) best_dist[i] = (uint8_t)b.mind;
                if (n_best) n_best[i] = (uint8_t)(b.cnt > 255u ? 255u : b.cnt);
                if (counts) atomicAdd(&counts[g], 1u);
                if (first_seen) {
                    const unsigned long long v = base + i;
                    if (v < first_seen[g]) atomicMin((unsigned long long*)&first_seen[g], v);
                }
            }  // else: the BC_IDX_MISS / BC_DIST_NONE / 0 the first stage wrote stay
        }
    };
    const uint64_t n_sub = (n + ST2_SUB - 1) / ST2_SUB;
    for (uint64_t sc = blockIdx.x; sc < n_sub; sc += gridDim.x) {
        qn += miss_queue_block(best_idx, sc * ST2_SUB, n, q_pos, qn, w_sum);
        __syncthreads();
        while (qn >= (uint32_t)BLOCK) {  // full rounds only: every lane busy
            serve(qn - BLOCK, BLOCK);
            qn -= BLOCK;
        }
    }
    __syncthreads();
    if (qn) serve(0, qn);
}

// ---- the second stage from a list, as passes over groups of tables --------------------------------------------------
// The missed reads arrive as a list in tile layout (MatchOut::miss_*: written by the first stage itself when that is
// k_match_wq, else gathered here by k_cascade_collect), are packed (prefix sum of the tiles' counts + k_cascade_pack),
// and every pass takes the packed list through a group of tables (BC_PAIR_GROUP_KB of slabs: three tables at 1e5
// guides) and carries {distance, ties, lowest index} to the next in 8 bytes per read; the last pass writes the
// results in place and tallies.  Launches are sized by the chip, not by the count (which stays on the device):
// asynchronous like the one-kernel form.  What the groups buy is modest (a pass costs 96-100 us per table and 1e7
// reads whether it holds one table or five: the lookup is bound by its instructions and its chain of dependent
// loads, not by where the slabs live); what the list buys is the walk (0.36 ms) and a lookup that runs with every
// lane busy.  (Measured and dropped: two or three tables' slabs requested together and their first candidates decided in
// one batch of loads — one chain slab -> entry -> index per group instead of per table: 252 us per three-table pass
// against 228.  Counters of the pass kernel: 210 vector instructions per slab visit, 56 % of the issue slots, 92 % of
// the lanes active, 1.43 L2 requests per visit, waves waiting on an instruction's operands 28 % of their time —
// `KPREFIX=k_cascade bash profiles/prof_valu.sh --max-mismatch 3`.)
__global__ __launch_bounds__(BLOCK) void k_cascade_collect(const uint64_t* __restrict__ reads, uint64_t n, uint64_t kmask,
                                                          const int32_t* __restrict__ best_idx, uint32_t* __restrict__ l_pos,
                                                          uint64_t* __restrict__ l_read, uint32_t* __restrict__ l_cnt) {
    // a wave per tile of 256 results, four per lane (the list's layout: see MatchOut)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (n + 255) >> 8, n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    const u32x4* pv = reinterpret_cast<const u32x4*>(best_idx);  // (16-byte aligned: the cascade's precondition)
    for (uint64_t tile = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
        const uint64_t i = tile * 256 + lane * 4;
        uint32_t e[4] = {0u, 0u, 0u, 0u};
        if (i + 3 < n) {
            const u32x4 v = pv[i >> 2];
            e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = (i + c < n) ? (uint32_t)best_idx[i + c] : 0u;
        }
        uint32_t tot = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool f = e[c] == (uint32_t)BC_IDX_MISS;
            const unsigned long long m = __ballot(f);
            if (f) {
                const uint64_t at = tile * 256 + tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                l_pos[at] = (uint32_t)(i + c);
                l_read[at] = reads[i + c] & kmask;
            }
            tot += (uint32_t)__popcll(m);
        }
        if (lane == 0u) l_cnt[tile] = tot;
    }
}

__device__ __forceinline__ uint64_t pair_state_pack(const PairBest& b) {  // ties saturate (n_best is capped at 255 anyway)
    return ((uint64_t)(b.mind & 0xFFu) << 48) | ((uint64_t)min(b.cnt, 0xFFFFu) << 32) | (uint64_t)b.bidx;
}

// tile layout -> packed list (offsets: the exclusive prefix sum of the tiles' counts); the last tile leaves the total
__global__ __launch_bounds__(BLOCK) void k_cascade_pack(const uint32_t* __restrict__ l_cnt, const uint32_t* __restrict__ l_off,
                                                       uint64_t n_tiles, const uint32_t* __restrict__ l_pos,
                                                       const uint64_t* __restrict__ l_read, uint32_t* __restrict__ d_pos,
                                                       uint64_t* __restrict__ d_read, unsigned long long* __restrict__ d_count) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    for (uint64_t tile = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
        const uint32_t c = l_cnt[tile], o = l_off[tile];
        for (uint32_t e = lane; e < c; e += 64u) {
            d_pos[(uint64_t)o + e] = l_pos[tile * 256 + e];
            d_read[(uint64_t)o + e] = l_read[tile * 256 + e];
        }
        if (tile == n_tiles - 1 && lane == 0u) *d_count = (unsigned long long)o + c;
    }
}

__global__ __launch_bounds__(BLOCK) void k_cascade_pass(const PairParams* __restrict__ ppp, int t0, int t1, int first, int last,
                                                       const uint64_t* __restrict__ l_read, const uint32_t* __restrict__ l_pos,
                                                       uint64_t* __restrict__ l_state, const unsigned long long* __restrict__ l_count,
                                                       int max_mm_, int32_t* __restrict__ best_idx, uint8_t* __restrict__ best_dist,
                                                       uint8_t* __restrict__ n_best, uint32_t* __restrict__ counts,
                                                       uint64_t* __restrict__ first_seen, uint64_t base) {
    const PairParams& pp = *ppp;
    const uint32_t max_mm = (uint32_t)max_mm_;
    const uint64_t m = *l_count;
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    // the next round's read and state are requested before this round's lookup (the stage is bound by its chains of
    // dependent loads — list entry, slab, candidate's index — not by instructions)
    uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint64_t r_next = j < m ? l_read[j] : 0ull;  // (a lane without an entry looks up some read and drops the result)
    uint64_t s_next = (!first && j < m) ? l_state[j] : 0ull;
    for (uint64_t j0 = (uint64_t)blockIdx.x * BLOCK; j0 < m; j0 += stride, j += stride) {  // whole workgroups: the lookup's loads are per quad
        const bool valid = j < m;
        const uint64_t r = r_next, s = s_next;
        const uint64_t jn = j + stride;
        PairBest b{0xFFu, 0u, 0xFFFFFFFFu};
        if (!first && valid) { b.mind = (uint32_t)(s >> 48) & 0xFFu; b.cnt = (uint32_t)(s >> 32) & 0xFFFFu; b.bidx = (uint32_t)s; }
        const uint64_t jc = jn < m ? jn : j0;  // (an index that exists: the loaded value is not touched before the next round)
        pairs_lookup_tables(pp, t0, t1, valid, (uint32_t)r, (uint32_t)(r >> 32), max_mm, b, [&]() {
            r_next = l_read[jc];
            if (!first) s_next = l_state[jc];
        });
        if (!valid) continue;
        if (!last) { l_state[j] = pair_state_pack(b); continue; }
        if (b.mind <= max_mm) {
            const uint32_t i = l_pos[j];
            const uint32_t g = b.bidx & SLAB_GIDX_MASK;
            best_idx[i] = (int32_t)g;
            if (best_dist) best_dist[i] = (uint8_t)b.mind;
            if (n_best) n_best[i] = (uint8_t)(b.cnt > 255u ? 255u : b.cnt);
            if (counts) atomicAdd(&counts[g], 1u);
            if (first_seen) {
                const unsigned long long v = base + i;
                if (v < first_seen[g]) atomicMin((unsigned long long*)&first_seen[g], v);
            }
        }  // else: the BC_IDX_MISS / BC_DIST_NONE / 0 the first stage wrote stay
    }
}

__global__ __launch_bounds__(BLOCK) void k_match_pairs(const PairParams* __restrict__ ppp, const uint64_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ flags, uint64_t n, int k, uint64_t kmask,
                                                      int max_mm, MatchOut o) {
    const PairParams& pp = *ppp;
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint64_t r = i < n ? reads[i] : 0ull;
    const bool skipped = i < n && read_skipped(r, flags, i, k);
    const bool valid = i < n && !skipped;  // (whole waves go through the lookup: its loads are per quad)
    r &= kmask;
    const PairBest b = pairs_lookup(pp, valid, (uint32_t)r, (uint32_t)(r >> 32), (uint32_t)max_mm);
    if (i >= n) return;
    if (skipped) emit(o, i, BC_IDX_SKIP, BC_DIST_NONE, 0u);
    else if (b.mind <= (uint32_t)max_mm) emit(o, i, (int32_t)(b.bidx & SLAB_GIDX_MASK), b.mind, b.cnt > 255u ? 255u : b.cnt);
    else emit(o, i, BC_IDX_MISS, BC_DIST_NONE, 0u);
}

template <typename T>
int upload(bc_lib* lib, const std::vector<T>& h, T** d) {
    void* p = nullptr;
    BC_HIP(hipMalloc(&p, std::max<size_t>(h.size() * sizeof(T), 16)));
    lib->owned.push_back(p);
    if (!h.empty()) BC_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = (T*)p;
    return BC_OK;
}

}  // namespace

namespace bc {

int build_pair_index(bc_lib* lib) {
    lib->d_pairs = nullptr;
    lib->d_pairs2 = nullptr;
    lib->pair_slabs.clear();
    lib->pair_slabs2.clear();
    const int k = lib->k;
    if (k < 8 || k > 25 || lib->G >= (1u << SLAB_GIDX_BITS)) return BC_OK;  // other lengths: the four-segment index
    // five pieces, the longer ones first
    int p_start[5], p_len[5], at = 0;
    for (int c = 0; c < 5; ++c) { p_len[c] = k / 5 + (c < k % 5 ? 1 : 0); p_start[c] = at; at += p_len[c]; }
    // key length in bases: ~6 guides per bucket, at least k - 16 (the rest must fit 16 bases per plane), at most 9
    int want = 1;
    // (32-byte slabs hold seven: about 1.5 per bucket; 64-byte slabs fifteen: about 6)
    while (want < 9 && ((double)lib->G / (double)(1ull << (2 * want))) > (PSW == 8u ? 2.0 : lib->knobs.pair_bucket_pct / 100.0)) ++want;
    want = std::max(want, k - 16);
    if (want > 9) return BC_OK;
    // the ten tables are independent: one host thread each (50 ms single-threaded at 1e5 guides)
    struct Built { PairTable T{}; std::vector<uint32_t> rests, gidx; bool ok = false; uint32_t other2[3] = {0u, 0u, 0u}, n_other2 = 0; bool in2 = false; };
    Built built[10];
    int pa[10], pb[10], nt = 0;
    for (int a = 0; a < 5; ++a)
        for (int b = a + 1; b < 5; ++b) { pa[nt] = a; pb[nt] = b; ++nt; }
    auto build_one = [&](int t) {
        const int a = pa[t], b = pb[t];
        PairTable T{};
        const int la = std::min(p_len[a], (want + 1) / 2 + std::max(0, (want / 2) - p_len[b]));
        const int lb = std::min(p_len[b], want - la);
        if (la < 1 || lb < 1 || k - la - lb > 16) return;
        T.a_start = (uint32_t)p_start[a]; T.a_len = (uint32_t)la;
        T.b_start = (uint32_t)p_start[b]; T.b_len = (uint32_t)lb;
        T.m_low = (1u << T.a_start) - 1u;
        T.m_mid = ((1u << (T.b_start - T.a_len)) - 1u) & ~T.m_low;
        T.m_high = ~((1u << (T.b_start - T.a_len)) - 1u);
        auto rest_mask = [&](int start, int len) {  // bases [start, start + len) in rest coordinates
            const uint32_t m = len >= 32 ? 0xFFFFFFFFu : (((1u << len) - 1u) << start);
            return pair_rest_plane(T, m);
        };
        T.must_clean = rest_mask(p_start[a] + la, p_len[a] - la) | rest_mask(p_start[b] + lb, p_len[b] - lb);
        T.n_other = 0;
        for (int c = 0; c < b; ++c)
            if (c != a) T.other[T.n_other++] = rest_mask(p_start[c], p_len[c]);
        // budgets <= 2 (three clean pieces at least) visit the tables (0,1) (2,3) (2,4) (3,4) in that order, and a guide
        // counts in the first of them whose pair is clean: (0,1) always; the others only when piece 0 or piece 1 is
        // dirty (ONE mask over both: non-zero = one of them is), and (2,4) / (3,4) only when piece 3 / piece 2 is dirty
        // too — otherwise (2,3) has counted it already
        built[t].in2 = (a == 0 && b == 1) || a >= 2;
        if (a >= 2) {
            built[t].other2[built[t].n_other2++] = rest_mask(p_start[0], p_len[0]) | rest_mask(p_start[1], p_len[1]);
            if (b == 4) built[t].other2[built[t].n_other2++] = rest_mask(p_start[a == 2 ? 3 : 2], p_len[a == 2 ? 3 : 2]);
        }
        const uint32_t nb = 1u << (2 * (la + lb));  // bucket = slab number, overflow slabs appended
        std::vector<uint32_t> cnt(nb + 1, 0u);
        std::vector<uint32_t> bucket_of(lib->G);
        for (uint32_t g = 0; g < lib->G; ++g) {
            const uint64_t v = lib->h_guides[g];
            bucket_of[g] = pair_bucket(T, (uint32_t)v, (uint32_t)(v >> 32));
            if (pair_rest_masks(T, (uint32_t)v) != pair_rest_plane(T, (uint32_t)v) ||
                pair_rest_masks(T, (uint32_t)(v >> 32)) != pair_rest_plane(T, (uint32_t)(v >> 32))) return;  // (never: the two forms agree)
            ++cnt[bucket_of[g] + 1];
        }
        for (uint32_t i = 0; i < nb; ++i) cnt[i + 1] += cnt[i];
        std::vector<uint32_t> order(lib->G), fill(cnt.begin(), cnt.end() - 1);
        for (uint32_t g = 0; g < lib->G; ++g) order[fill[bucket_of[g]]++] = g;  // ascending guide index inside a bucket
        // slabs of 15 rests + tail {next << 4 | count}; the guide indices in a parallel array
        std::vector<uint32_t>& rests = built[t].rests;
        std::vector<uint32_t>& gix = built[t].gidx;
        rests.assign((size_t)nb * PSW, 0u);
        gix.assign((size_t)nb * PSW, 0u);
        for (uint32_t bkt = 0; bkt < nb; ++bkt) {
            size_t cur = bkt;
            uint32_t pos = 0;
            for (uint32_t j = cnt[bkt]; j < cnt[bkt + 1]; ++j) {
                if (pos == PSW - 1u) {
                    const size_t nxt = rests.size() / PSW;
                    rests.resize(rests.size() + PSW, 0u);
                    gix.resize(gix.size() + PSW, 0u);
                    rests[cur * PSW + PSW - 1u] = ((uint32_t)nxt << 4) | (PSW - 1u);
                    cur = nxt;
                    pos = 0;
                }
                const uint32_t g = order[j];
                const uint64_t v = lib->h_guides[g];
                rests[cur * PSW + pos] = pair_rest_plane(T, (uint32_t)v) | (pair_rest_plane(T, (uint32_t)(v >> 32)) << 16);
                gix[cur * PSW + pos] = g;
                ++pos;
            }
            rests[cur * PSW + PSW - 1u] = pos;  // last slab of the bucket: no next
        }
        if ((rests.size() / PSW) >> 28) return;  // slab numbers travel in 28 bits
        built[t].T = T;
        built[t].ok = true;
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 10; ++t) th.emplace_back(build_one, t);
        for (auto& x : th) x.join();
    }
    PairParams pp{};
    pp.ntab = 0;
    for (int t = 0; t < 10; ++t) {
        if (!built[t].ok) return BC_OK;  // a geometry the 8-byte entries cannot hold: the four-segment index stays
        uint32_t *d_rests, *d_gidx;
        int rc = upload(lib, built[t].rests, &d_rests);
        if (rc == BC_OK) rc = upload(lib, built[t].gidx, &d_gidx);
        if (rc != BC_OK) return rc;
        built[t].T.rests = d_rests;
        built[t].T.gidx = d_gidx;
        pp.t[pp.ntab++] = built[t].T;
        lib->pair_slabs.push_back((uint32_t)(built[t].rests.size() / PSW));
    }
    std::vector<PairParams> one(1, pp);
    PairParams* d_pp;
    int rc = upload(lib, one, &d_pp);
    if (rc != BC_OK) return rc;
    PairParams pp2{};
    pp2.ntab = 0;
    for (int t = 0; t < 10; ++t) {
        if (!built[t].in2) continue;
        PairTable T = pp.t[t];
        T.n_other = built[t].n_other2;
        for (int c = 0; c < 3; ++c) T.other[c] = built[t].other2[c];
        pp2.t[pp2.ntab++] = T;
        lib->pair_slabs2.push_back(lib->pair_slabs[(size_t)t]);
    }
    std::vector<PairParams> two(1, pp2);
    PairParams* d_pp2;
    rc = upload(lib, two, &d_pp2);
    if (rc != BC_OK) return rc;
    lib->d_pairs2 = d_pp2;
    lib->d_pairs = d_pp;
    return BC_OK;
}

// the descriptors a budget walks: four tables for <= 2 substitutions (BC_PAIR_FOUR=0: all ten, as for three)
static const PairParams* pairs_for(const bc_lib* lib, int max_mm) {
    return (max_mm <= 2 && lib->d_pairs2 && lib->knobs.pair_four) ? lib->d_pairs2 : lib->d_pairs;
}

int launch_pairs(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n, const MatchOut& o,
                 hipStream_t st) {
    if (!lib->d_pairs) { set_error("bc_match: no pair index for k = %d", lib->k); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    const uint64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks >> 31) { set_error("bc_match: launch too large"); return BC_EINVAL; }
    hipLaunchKernelGGL(k_match_pairs, dim3((unsigned)blocks), dim3(BLOCK), 0, st, pairs_for(lib, max_mm), d_reads, d_flags, n, lib->k,
                       lib->kmask, max_mm, o);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

size_t cascade_scan_bytes(uint64_t tiles) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)tiles, rocprim::plus<uint32_t>(), 0);
    return bytes;
}

// does the second stage of a launch of n reads work from a list of the missed reads (passes over groups of tables)?
// Few reads: the one-kernel form, which needs no workspace.
bool cascade_uses_list(const bc_lib* lib, uint64_t n) {
    return lib->d_pairs && lib->knobs.pair_group_kb > 0 && n >= (uint64_t)std::max(0, lib->knobs.pair_pass_min);
}

// second stage of the cascade, in place over the first stage's results: asynchronous in both forms.  w: the workspace
// of the list form (NULL: the one-kernel form); listed: the first stage has written the list already (k_match_wq)
int launch_cascade_stage2(bc_lib* lib, int max_mm, const uint64_t* d_reads, uint64_t n, const MatchOut& o, hipStream_t st,
                          TallyWs* w, bool listed) {
    if (!lib->d_pairs) { set_error("bc_match: no pair index for k = %d", lib->k); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    const uint64_t n_sub = (n + ST2_SUB - 1) / ST2_SUB;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_sub, 2048));  // 8 workgroups per CU
    if (!w) {
        hipLaunchKernelGGL(k_cascade_stage2, dim3(grid), dim3(BLOCK), 0, st, pairs_for(lib, max_mm), d_reads, n, lib->kmask, max_mm, o.best_idx,
                           o.best_dist, o.n_best, o.counts, o.first_seen, o.base);
        BC_HIP(hipGetLastError());
        return BC_OK;
    }
    const bool four = pairs_for(lib, max_mm) == lib->d_pairs2;
    const std::vector<uint32_t>& slabs = four ? lib->pair_slabs2 : lib->pair_slabs;
    const int ntab = (int)slabs.size();
    const uint64_t budget = (uint64_t)std::max(0, lib->knobs.pair_group_kb) * 1024ull;
    const uint64_t n_tiles = (n + 255) >> 8;
    if (!listed) {
        const unsigned cgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_tiles + BLOCK / 64 - 1) / (BLOCK / 64), 4096));
        hipLaunchKernelGGL(k_cascade_collect, dim3(cgrid), dim3(BLOCK), 0, st, d_reads, n, lib->kmask, (const int32_t*)o.best_idx,
                           w->casc_pos, w->casc_read, w->casc_cnt);
    }
    // tile layout -> packed list: prefix sum of the tiles' counts, one copy (0.05 ms per 1e7 listed reads; the passes
    // ran 40 % slower from the tile layout itself — a search per entry and partly filled rounds)
    {
        size_t bytes = w->casc_scan_bytes;
        BC_HIP(rocprim::exclusive_scan(w->casc_scan_tmp, bytes, (const uint32_t*)w->casc_cnt, w->casc_off, 0u, (size_t)n_tiles,
                                       rocprim::plus<uint32_t>(), st));
        const unsigned kgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_tiles + BLOCK / 64 - 1) / (BLOCK / 64), 4096));
        hipLaunchKernelGGL(k_cascade_pack, dim3(kgrid), dim3(BLOCK), 0, st, (const uint32_t*)w->casc_cnt, (const uint32_t*)w->casc_off,
                           n_tiles, (const uint32_t*)w->casc_pos, (const uint64_t*)w->casc_read, w->casc_dpos, w->casc_dread,
                           w->casc_count);
    }
    const unsigned pgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + BLOCK - 1) / BLOCK, (uint64_t)lib->knobs.pair_grid));  // 8 workgroups per CU
    for (int t0 = 0; t0 < ntab;) {  // greedy groups: tables in order while their slabs fit the budget, at least one
        int t1 = t0 + 1;
        uint64_t bytes = (uint64_t)slabs[(size_t)t0] * (PSW * 4u);
        while (t1 < ntab && bytes + (uint64_t)slabs[(size_t)t1] * (PSW * 4u) <= budget) bytes += (uint64_t)slabs[(size_t)t1++] * (PSW * 4u);
        hipLaunchKernelGGL(k_cascade_pass, dim3(pgrid), dim3(BLOCK), 0, st, pairs_for(lib, max_mm), t0, t1, t0 == 0 ? 1 : 0, t1 == ntab ? 1 : 0,
                           (const uint64_t*)w->casc_dread, (const uint32_t*)w->casc_dpos, w->casc_state,
                           (const unsigned long long*)w->casc_count, max_mm, o.best_idx, o.best_dist, o.n_best, o.counts, o.first_seen,
                           o.base);
        t0 = t1;
    }
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // namespace bc
