// bc_match_dev.h — device-side helpers shared by the match kernels' translation units (gfx950 only):
// bc_match.hip (hash, wave matcher, CSR, brute force, tally) and bc_match_quad.hip (the previous-generation quad
// kernels).  Output descriptors live in namespace bc (they cross the two files); everything else is per-file.
#pragma once
#include <type_traits>

#include "bc_device.h"

namespace bc {

// Tally records (K5', see k_hist_part): while a tile of 256 results is still on chip, the matcher writes, for each
// of the NR guide ranges the tally kernel owns in LDS, one line of up to `cap` two-byte records (guide index inside
// the range, 0xFFFF = empty) to rec[range][tile][cap].  The tally then streams every range's records ONCE instead of
// sweeping best_idx once per range (4 B per read and range).  The line shrinks as the ranges multiply — a tile's 256
// results spread over more lines — so that the records stay a few bytes per read: 64 records (128 B) up to 16
// ranges, 32 up to 32 ranges, 16 up to 64 ranges (1.3e6 guides): 2.5 B per read at five ranges, 8 at 16, 4 at 32,
// 8 at 64.  A tile with more hits in one range than a line holds writes empty lines and raises its overflow flag;
// the tally takes such tiles from best_idx.  (Round 2 had one line size, 64, and records for 5..8 ranges only:
// libraries of 1.7e5 ... 1e6 guides fell back to one sweep of best_idx per range, 0.4 ... 2.3 ms per 1e8 reads.)
constexpr int PART_MAX_NR = 64;      // ranges
constexpr int PART_SLOTS = 1024;     // two-byte slots of a wave's staging area (2 KiB): nr * cap <= PART_SLOTS
struct PartOut {
    uint16_t* rec;        // [nr][n_tiles][cap]; NULL = no records wanted
    uint32_t* ovf_flag;   // [n_tiles] 1: the tile's records did not fit (its lines are empty), 0: they did; written for
                          // every tile (no shared append counter: one took ~90 appends per microsecond, 4 ms when every
                          // tile overflowed)
    uint32_t n_tiles;     // tiles of this launch (record stride between ranges)
    uint32_t nr;          // ranges
    uint32_t R;           // guides per range (<= HIST_R)
    uint32_t magic;       // range of guide g = umulhi(g, magic) (exact for g < G, checked on the host)
    uint32_t cap_log2;    // log2 of the records per line: 6, 5 or 4
};

struct MatchOut {
    int32_t* best_idx;
    uint8_t* best_dist;
    uint8_t* n_best;
    uint32_t* counts;      // NULL when the tally is done by k_hist / k_hist_part afterwards
    uint64_t* first_seen;  // idem
    uint64_t base;
    PartOut part;
    // first stage of a cascade (launch_cascade): the reads left at BC_IDX_MISS also go on a list — position in this
    // launch and masked read — for the second stage.  NULL otherwise.  The list is laid out by TILE of 256 reads:
    // tile t owns entries [256 t, 256 t + miss_cnt[t]) — no atomics, no order between tiles (one shared counter
    // serialised the tiles' atomics: 3.7 ms per 1e8 reads; 1 024 counters still cost every wave an atomic round trip
    // at its end: +0.23 ms).  k_match_wq fills it (its launcher then sets *miss_listed, a HOST flag); after any other
    // kernel the second stage gathers the list from best_idx itself (k_cascade_collect).
    uint32_t* miss_pos = nullptr;
    uint64_t* miss_read = nullptr;
    uint32_t* miss_cnt = nullptr;
    bool* miss_listed = nullptr;
};


// launchers of bc_match_quad.hip (the kernels stay in their own file; see there for when they run)
int launch_quad_fused(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                      const MatchOut& o, hipStream_t st);
int launch_quad_slab(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                     const MatchOut& o, hipStream_t st);

// bc_match_pairs.hip: the pair index of the 3-mismatch matcher and its kernel (exact for any read, budget <= 3)
int build_pair_index(bc_lib* lib);
int launch_pairs(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n, const MatchOut& o,
                 hipStream_t st);
bool cascade_uses_list(const bc_lib* lib, uint64_t n);  // the second stage as passes over table groups (needs the workspace)
size_t cascade_scan_bytes(uint64_t tiles);  // rocprim scratch for the prefix sum over that many tile counts
int cascade_workspace(bc_lib* lib, hipStream_t st, uint64_t n, TallyWs** out);
int launch_cascade_stage2(bc_lib* lib, int max_mm, const uint64_t* d_reads, uint64_t n, const MatchOut& o, hipStream_t st,
                          TallyWs* w, bool listed);

}  // namespace bc

namespace {

using namespace bc;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Result / record stores of the wave matcher: nothing in the kernel touches these lines again.  BC_STORE_POLICY picks the
// cache-policy bits (compile-time experiment, profiles/r04_store_policy_ab.txt): 0 = nt (the compiler's non-temporal
// store), 1 = sc0 sc1 nt (write through to memory, streaming), 2 = sc1 nt, 3 = no bits.
#ifndef BC_STORE_POLICY
#define BC_STORE_POLICY 0
#endif
__device__ __forceinline__ void stream_store(u32x4 v, u32x4* p) {
#if BC_STORE_POLICY == 0 || !defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#elif BC_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
#elif BC_STORE_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void stream_store(uint64_t v, uint64_t* p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_store(uint16_t v, uint16_t* p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_store(uint32_t v, uint32_t* p) {
#if BC_STORE_POLICY == 0 || !defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#elif BC_STORE_POLICY == 1
    asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
#elif BC_STORE_POLICY == 2
    asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
// streaming input: keep it out of the L2's way (the lookup tables live there)
#ifndef BC_LOAD_POLICY
#define BC_LOAD_POLICY 0   // read stream: 0 = non-temporal loads, 1 = plain loads (compile-time experiment)
#endif
__device__ __forceinline__ void load_reads4(const uint64_t* p, uint64_t (&r)[4]) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
#if BC_LOAD_POLICY == 0
    u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
#else
    u32x4 a = q[0], b = q[1];
#endif
    r[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); r[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    r[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); r[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
}

// the same for a whole tile of 256 reads and lane l's reads 2l, 2l + 1, 128 + 2l, 129 + 2l: each of the two load instructions
// covers 1 KiB contiguously, so every line of the stream is requested once (k_match_wq: see there)
__device__ __forceinline__ void load_reads2x2(const uint64_t* tile, uint32_t lane, uint64_t (&r)[4]) {
    const u32x4* q = reinterpret_cast<const u32x4*>(tile);
#if BC_LOAD_POLICY == 0
    const u32x4 a = __builtin_nontemporal_load(q + lane), b = __builtin_nontemporal_load(q + 64 + lane);
#else
    const u32x4 a = q[lane], b = q[64 + lane];
#endif
    r[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); r[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    r[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); r[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
}

// One wave, one tile of 256 consecutive results (4 per lane): stage the tally records in LDS (stage: PART_SLOTS
// two-byte slots, cnt: PART_MAX_NR counters; both private to the wave) and store them as whole lines.
__device__ __forceinline__ void part_emit(const PartOut& P, uint32_t tile, const int32_t (&idx)[4], uint32_t lane,
                                          uint16_t* stage, uint32_t* cnt) {
    const uint32_t cap = 1u << P.cap_log2, used = P.nr << P.cap_log2;  // slots in use (<= PART_SLOTS)
    // empty staging area: 16 slots (32 bytes) per lane and step; counters
    for (uint32_t b = lane * 16u; b < used; b += 64u * 16u) {
        reinterpret_cast<uint4*>(stage + b)[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        reinterpret_cast<uint4*>(stage + b)[1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    if (lane < (uint32_t)PART_MAX_NR) cnt[lane] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (idx[j] >= 0) {
            const uint32_t g = (uint32_t)idx[j];
            const uint32_t p = __umulhi(g, P.magic);
            const uint32_t slot = atomicAdd(&cnt[p], 1u);
            if (slot < cap) stage[(p << P.cap_log2) + slot] = (uint16_t)(g - p * P.R);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t c = lane < P.nr ? cnt[lane] : 0u;
    const bool ovf = __ballot(c > cap) != 0ull;
    // (the flags are zero when a pass starts — k_first_refine clears the ones it finds set — so only an overflowing tile
    // writes: a four-byte store per tile cost the matcher 12 us per 1e8 reads, profiles/r04_part_emit_flag.txt)
    if (ovf && lane == 0) P.ovf_flag[tile] = 1u;
    // lines out, one dword per lane and step: dword d of the staging area belongs to range d / (cap / 2); a range's
    // lines lie one stride (< 2^31 bytes) apart, a tile's line at tile * cap * 2 bytes inside it
    const uint32_t* sw = reinterpret_cast<const uint32_t*>(stage);
    const uint32_t stride = P.n_tiles << (P.cap_log2 + 1u);  // bytes between two ranges' lines of a tile
    const uint32_t dw_log2 = P.cap_log2 - 1u;                // dwords per line
    char* line0 = reinterpret_cast<char*>(P.rec) + ((size_t)tile << (P.cap_log2 + 1u));
    for (uint32_t d = lane; d < (used >> 1); d += 64u) {
        const uint32_t r = d >> dw_log2, within = d & ((1u << dw_log2) - 1u);
        const uint32_t v = ovf ? 0xFFFFFFFFu : sw[d];
        stream_store(v, (uint32_t*)(line0 + (size_t)r * stride + within * 4u));
    }
}

__device__ __forceinline__ bool read_skipped(uint64_t r, const uint8_t* flags, uint64_t i, int k) {
    bool s = (k <= 31) && (r >> 63);
    if (flags) s = s || (flags[i] != 0);
    return s;
}

__device__ __forceinline__ void tally(const MatchOut& o, uint64_t i, int32_t idx) {
    if (idx < 0) return;
    if (o.counts) atomicAdd(&o.counts[idx], 1u);
    if (o.first_seen) {
        unsigned long long v = o.base + i;
        // a stale (cached) value can only be larger than the true one, so the
        // plain pre-check never drops a needed update
        if (v < o.first_seen[idx]) atomicMin((unsigned long long*)&o.first_seen[idx], v);
    }
}

__device__ __forceinline__ void emit(const MatchOut& o, uint64_t i, int32_t idx, uint32_t dist,
                                     uint32_t nb) {
    o.best_idx[i] = idx;
    if (o.best_dist) o.best_dist[i] = (uint8_t)dist;
    if (o.n_best) o.n_best[i] = (uint8_t)(nb > 255u ? 255u : nb);
    tally(o, i, idx);
}

// ---------------------------------------------------------------------------
// K2: exact match.  slot layout (narrow): key bits in place (kmask), guide
// index + 1 spread over the free bits above k in both 32-bit halves; 0 = empty.
// wide (k too long for that): slot = index + 1, verified against guides[].
// ---------------------------------------------------------------------------
template <bool WIDE>
__device__ __forceinline__ int32_t hash_lookup(uint64_t key, const uint64_t* __restrict__ s64,
                                               const uint32_t* __restrict__ s32,
                                               const uint64_t* __restrict__ guides, int k,
                                               int log_slots, uint64_t kmask) {
    const uint32_t smask = (1u << log_slots) - 1u;
    uint32_t h = hash32(key) >> (32 - log_slots);
    for (uint32_t probe = 0; probe <= smask; ++probe) {
        if (WIDE) {
            uint32_t s = s32[h];
            if (s == 0) return BC_IDX_MISS;
            if (guides[s - 1] == key) return (int32_t)(s - 1);
        } else {
            // narrow table: buckets of two 8-byte slots fetched by one 16-byte load (log_slots counts
            // buckets); slot .x fills before .y, so an empty .y ends the probe sequence
            const ulonglong2 b = reinterpret_cast<const ulonglong2*>(s64)[h];
            if ((b.x & kmask) == key && b.x != 0) {
                uint32_t lo = (uint32_t)b.x >> k, hi = (uint32_t)(b.x >> 32) >> k;
                return (int32_t)((lo | (hi << (32 - k))) - 1u);
            }
            if ((b.y & kmask) == key && b.y != 0) {
                uint32_t lo = (uint32_t)b.y >> k, hi = (uint32_t)(b.y >> 32) >> k;
                return (int32_t)((lo | (hi << (32 - k))) - 1u);
            }
            if (b.y == 0) return BC_IDX_MISS;
        }
        h = (h + 1) & smask;
    }
    return BC_IDX_MISS;
}

struct Best {
    int32_t idx;
    uint32_t dist, nb, cur;  // cur = min(dist, max_mm): acceptance threshold
};

__device__ __forceinline__ void consider(Best& b, uint32_t d, uint32_t gi) {
    if (d < b.dist) {
        b.dist = d; b.idx = (int32_t)gi; b.nb = 1; b.cur = d;
    } else {  // d == b.dist (d <= cur <= dist)
        b.nb++;
        if ((int32_t)gi < b.idx) b.idx = (int32_t)gi;
    }
}

__device__ __forceinline__ uint32_t seed_value(uint32_t lo, uint32_t hi, int start, int slen) {
    uint32_t m = (1u << slen) - 1u;
    return ((lo >> start) & m) | (((hi >> start) & m) << slen);
}

// true when the candidate was already enumerated through an earlier seed
__device__ __forceinline__ bool seen_earlier(const SeedParams& sp, int s, uint32_t mm) {
    bool seen = false;
    for (int t = 0; t < s; ++t) seen = seen || (((mm >> sp.start[t]) & ((1u << sp.slen[t]) - 1u)) == 0u);
    return seen;
}

// ---------------------------------------------------------------------------
// K3a': pigeonhole seed index in slab form.  LPR lanes share one read: a bucket
// visit is ONE wave-level load in which the LPR lanes of a read fetch adjacent
// 16-byte pieces of the same slab (one L2 request per 64/128 bytes), entries
// carry the guide index, and a read stops after the first table once it has an
// exact hit (guides are distinct, so nothing can tie with distance 0).
// ---------------------------------------------------------------------------
// cross-lane helpers inside a group of LPR lanes.  For LPR = 4 a group is a DPP
// quad: one v_mov_dpp per exchange instead of a ds_bpermute round trip.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int LPR>
__device__ __forceinline__ uint32_t grp_min(uint32_t v) {
    if constexpr (LPR == 4) {
        v = min(v, quad_perm<0xB1>(v));  // lanes ^1
        v = min(v, quad_perm<0x4E>(v));  // lanes ^2
    } else {
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m));
    }
    return v;
}
template <int LPR>
__device__ __forceinline__ uint32_t grp_sum(uint32_t v) {
    if constexpr (LPR == 4) {
        v += quad_perm<0xB1>(v);
        v += quad_perm<0x4E>(v);
    } else {
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m);
    }
    return v;
}
template <int LPR>
__device__ __forceinline__ uint32_t grp_last(uint32_t v, int last_lane) {  // value held by the group's last lane
    if constexpr (LPR == 4) return quad_perm<0xFF>(v);
    else return (uint32_t)__shfl((int)v, last_lane);
}

constexpr int SLAB_MAX_LOG2S = 5;  // <= 32 entries (256 B) per slab; longer buckets chain

// Running result of one lane: key = dist << 24 | guide index (one v_min keeps the
// minimum distance and, among ties, the lowest index; the slab layout is only
// built for libraries below 2^24 guides), cnt = per-distance candidate counts in
// 16-bit fields, cur = min(max_mm, best distance) = the acceptance threshold.
struct SlabBest {
    uint32_t key;
    uint32_t cur;
    uint64_t cnt;
};

// all of a slab's pieces for this lane, issued back to back (memory-level
// parallelism: the wave has MAXSTEP x 16 requests in flight, not 16)
template <int LPR, int MAXSTEP>
__device__ __forceinline__ void slab_load(const SlabTable& T, uint32_t slab, int sub, uint4 (&v)[MAXSTEP]) {
    const uint4* p = reinterpret_cast<const uint4*>(T.slabs + ((size_t)slab << T.log2S));
    const int nstep = (1 << T.log2S) / (2 * LPR);
#pragma unroll
    for (int st = 0; st < MAXSTEP; ++st)
        if (st < nstep) v[st] = p[st * LPR + sub];
}

// verify the loaded entries; returns the chained slab (0 = none), identical in
// all LPR lanes of the group.  Hot path per entry: xor, shift, or+and, popcount,
// compare.  Empty slots hold rest = 0 and are rejected in the (rare) accept path.
template <int LPR, int MAXSTEP, int S>
__device__ __forceinline__ uint32_t slab_scan(const SlabTable& T, const uint4 (&v)[MAXSTEP], uint32_t rest,
                                              SlabBest& b, int last_lane) {
    const int nstep = (1 << T.log2S) / (2 * LPR);
    const uint32_t hmask = (1u << T.half) - 1u;
    uint32_t lz = 0, lw = 0;
#pragma unroll
    for (int st = 0; st < MAXSTEP; ++st) {
        if (st < nstep) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t er = e ? v[st].z : v[st].x, gw = e ? v[st].w : v[st].y;
                const uint32_t x = rest ^ er;
                const uint32_t mm = (x | (x >> T.half)) & hmask;
                const uint32_t d = __popc(mm);
                if (d <= b.cur) {
                    bool ok = gw < SLAB_LINK;
                    const uint32_t gi = gw & SLAB_GIDX_MASK;
#pragma unroll
                    for (int t = 0; t < S; ++t) ok = ok && ((mm & T.dmask[t]) != 0u);  // not met through an earlier seed
                    if (ok) {
                        b.key = min(b.key, (d << 24) | gi);
                        b.cur = d;  // d <= cur
                        b.cnt += 1ull << (16 * d);
                    }
                }
            }
            if (st == nstep - 1) { lz = v[st].z; lw = v[st].w; }
        }
    }
    // the link is the last entry of the slab: second entry of the last lane's last piece
    lw = grp_last<LPR>(lw, last_lane);
    lz = grp_last<LPR>(lz, last_lane);
    return (lw == SLAB_LINK) ? lz : 0u;
}

// drop the seed field from both planes: rest = lo' | hi' << half  (k <= 28 here, so no shift reaches 32)
__device__ __forceinline__ uint32_t slab_rest(const SlabTable& T, uint32_t rlo, uint32_t rhi) {
    const uint32_t below = (1u << T.start) - 1u;
    const uint32_t sh = T.start + T.slen;
    const uint32_t lo_r = (rlo & below) | ((rlo >> sh) << T.start);
    const uint32_t hi_r = (rhi & below) | ((rhi >> sh) << T.start);
    return lo_r | (hi_r << T.half);
}

// ---------------------------------------------------------------------------
// Branch-free slab scan of the quad kernels:
//   key  = dist << 24 | guide            -> kmin = min key  (best distance, lowest index)
//   key2 = key + (255 - max_mm) << 24    -> kmax = max key2 (largest candidate; distances
//                                           above max_mm wrap below every candidate)
// One distinct candidate <=> kmin == kmax - bias: n_best = 1.  Anything else
// (several candidates within max_mm) goes through the exact slow path.
// ---------------------------------------------------------------------------
template <int LPR, int MAXSTEP>
__device__ __forceinline__ uint32_t slab_scan_fast(const SlabTable& T, const uint4 (&v)[MAXSTEP], uint32_t rest,
                                                   uint32_t bias, uint32_t& kmin, uint32_t& kmax, int last_lane) {
    const int nstep = (1 << T.log2S) / (2 * LPR);
    const uint32_t hmask = (1u << T.half) - 1u;
    uint32_t lz = 0, lw = 0;
#pragma unroll
    for (int st = 0; st < MAXSTEP; ++st) {
        if (st < nstep) {
            const uint32_t x0 = rest ^ v[st].x, x1 = rest ^ v[st].z;
            const uint32_t d0 = __popc((x0 | (x0 >> T.half)) & hmask);
            const uint32_t d1 = __popc((x1 | (x1 >> T.half)) & hmask);
            // key = dist << 24 | flags | guide (empty / link slots stay 0xFFFFFFFx)
            const uint32_t k0 = (d0 << 24) | v[st].y;
            const uint32_t k1 = (d1 << 24) | v[st].w;
            kmin = min(kmin, min(k0, k1));
            kmax = max(kmax, max(k0 + bias, k1 + bias));
            if (st == nstep - 1) { lz = v[st].z; lw = v[st].w; }
        }
    }
    lw = grp_last<LPR>(lw, last_lane);
    lz = grp_last<LPR>(lz, last_lane);
    return (lw == SLAB_LINK) ? lz : 0u;
}

template <int LPR>
__device__ __forceinline__ uint32_t grp_max(uint32_t v) {
    if constexpr (LPR == 4) {
        v = max(v, quad_perm<0xB1>(v));
        v = max(v, quad_perm<0x4E>(v));
    } else {
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
    }
    return v;
}

// exact evaluation of one read over every table (chains followed, duplicates
// across seeds removed, counts per distance): the definition of the result
template <int LPR, int NSEG>
__device__ __forceinline__ void slab_slow(const SlabParams& sp, uint32_t rlo, uint32_t rhi, int max_mm, int sub,
                                       int last_lane, uint32_t& key_out, uint32_t& nb_out) {
    constexpr int MAXSTEP = (1 << SLAB_MAX_LOG2S) / (2 * LPR);
    SlabBest b{0xFFFFFFFFu, (uint32_t)max_mm, 0ull};
    uint4 v[MAXSTEP];
    auto one = [&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        const SlabTable& T = sp.t[s];
        const uint32_t rest = slab_rest(T, rlo, rhi);
        uint32_t slab = seed_value(rlo, rhi, T.start, T.slen);
        do {
            slab_load<LPR, MAXSTEP>(T, slab, sub, v);
            slab = slab_scan<LPR, MAXSTEP, s>(T, v, rest, b, last_lane);
        } while (slab != 0u);
        b.cur = grp_min<LPR>(b.cur);
    };
    one(std::integral_constant<int, 0>{});
    if constexpr (NSEG > 1) one(std::integral_constant<int, 1>{});
    if constexpr (NSEG > 2) one(std::integral_constant<int, 2>{});
    if constexpr (NSEG > 3) one(std::integral_constant<int, 3>{});
    const uint32_t key = grp_min<LPR>(b.key);
    const uint32_t md = key >> 24;
    const uint32_t mine = (md <= 3u) ? (uint32_t)((b.cnt >> (16 * (md & 3u))) & 0xFFFFu) : 0u;
    key_out = key;
    nb_out = grp_sum<LPR>(mine);
}

}  // namespace
