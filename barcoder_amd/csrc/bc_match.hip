// bc_match.hip — read -> library matching on gfx950 (K2 exact, K3 Hamming, K5
// per-guide histogram).  Integer / byte kernels: no MFMA.
//
// Replaces the per-read body of heuristicount.py:428-562 (process_chunk: set
// lookup + Counter update) and generalises it to <= d substitutions, which the
// reference can only obtain from `bowtie -v N` (targets.py:496-509).
//
// Algorithms over the same inputs / outputs (bc_match_dev selects):
//   k_match_hash   : d = 0.  One open-addressing probe (two-slot buckets, one 16-byte
//                    load) per read into a 4 MiB table that lives in L2.
//   k_match_wq     : d = 1, 2, the default.  Pigeonhole slab index, no hash table (table 0 is
//                    the exact stage); a wave owns 256 reads, quad-cooperative slab loads for
//                    the full-width stages, a quad work list for the sparse ones, a filter line
//                    in front of table 2.
//   cascade (d = 3): the whole d <= 2 pass over every read, then k_cascade_stage2 (bc_match_pairs.hip): the reads
//                    still marked BC_IDX_MISS through the pair-of-pieces index, in place; launch_cascade.  Also d = 2 on
//                    libraries whose buckets are too long for the wave matcher (d <= 1 pass + pair index).
//   k_match_fused  : previous generation (bc_match_quad.hip: hash stage + LDS queue + DPP quad per
//                    queued read): buckets above 40 guides (libraries above ~6.5e5 20-mers), BC_LANE=0.
//   k_match_slab   : the slab index as one simple kernel (small / unaligned inputs, d = 0 via seed).
//   k_match_seed   : CSR form of the seed index, lane = read (k - seed > 16 bases, huge libraries).
//   k_match_all    : every (query, guide, dist) within d, optionally gated by a PAM bitmap:
//                    the Bowtie -a -v N bypass of targets.find_hits.
//   k_match_brute  : any d.  Guide tile in LDS, lane = read, all pairs XOR + popcount.
//                    VALU-bound; the in-GPU cross-check.
//   k_hist_part + k_first_refine : per-guide counts / first-seen from the range-partitioned
//                    records the matchers write while a tile's results are on chip;
//   k_hist         : the same from best_idx, one sweep per guide range (other kernels, few ranges).
// Bounds (measured): the lookup kernels sit on the rate of 64-byte L2 requests
// under a concurrent stream (~165 G/s; 1.1 requests per read is the floor: 0.66 ms per 1e8 reads;
// k_match_wq makes 1.79); the tally kernels are HBM streams.
#include <string.h>
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>

#include "bc_match_dev.h"

namespace {


template <bool WIDE, int VEC>
__global__ __launch_bounds__(BLOCK) void k_match_hash(const uint64_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ flags, uint64_t n,
                                                      const uint64_t* __restrict__ s64,
                                                      const uint32_t* __restrict__ s32,
                                                      const uint64_t* __restrict__ guides, int k,
                                                      int log_slots, uint64_t kmask, MatchOut o) {
    __shared__ __attribute__((aligned(16))) uint16_t p_stage[VEC == 4 ? (BLOCK / 64) * PART_SLOTS : 8];
    __shared__ uint32_t p_cnt[VEC == 4 ? (BLOCK / 64) * PART_MAX_NR : 1];
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    const uint64_t ngroups = n / VEC;
    // the loop bound is wave-uniform (g0 = the wave's first group): a wave's 64 groups of 4 reads are one tile of
    // 256 consecutive reads, and the whole wave takes part in the tile's tally records even when the tile is ragged
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t g0 = (uint64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63u); g0 < ngroups; g0 += stride) {
        const uint64_t g = g0 + lane;
        const bool active = g < ngroups;
        uint64_t r[VEC];
        int32_t idx[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { r[j] = 0; idx[j] = BC_IDX_SKIP; }
        // a whole tile (the wave's 64 groups all exist): lane l takes reads 2l, 2l + 1, 128 + 2l, 129 + 2l of it, so that each
        // of the two load instructions covers 1 KiB contiguously and every line of the stream is requested once (see
        // k_match_wq); a ragged last tile keeps four consecutive reads per lane
        const bool whole = VEC == 4 && g0 + 63 < ngroups;  // wave-uniform
        auto index_of = [&](int j) -> uint64_t {
            return whole ? g0 * 4 + (j < 2 ? 0u : 128u) + 2u * lane + (uint32_t)(j & 1) : g * VEC + (uint64_t)j;
        };
        if (active) {
            if constexpr (VEC == 4) {
                if (whole) load_reads2x2(reads + g0 * 4, lane, r);
                else load_reads4(reads + g * 4, r);
            } else {
                r[0] = reads[g];
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (read_skipped(r[j], flags, index_of(j), k)) idx[j] = BC_IDX_SKIP;
                else idx[j] = hash_lookup<WIDE>(r[j] & kmask, s64, s32, guides, k, log_slots, kmask);
            }
        }
        if constexpr (VEC == 4) {
            if (active) {   // results are final: non-temporal stores keep the table in the L2
                uint32_t dd[4], nn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { dd[j] = idx[j] >= 0 ? 0u : (uint32_t)BC_DIST_NONE; nn[j] = idx[j] >= 0 ? 1u : 0u; }
                if (whole) {
                    uint64_t* bi = reinterpret_cast<uint64_t*>(o.best_idx + g0 * 4) + lane;
                    stream_store((uint64_t)(uint32_t)idx[0] | ((uint64_t)(uint32_t)idx[1] << 32), bi);
                    stream_store((uint64_t)(uint32_t)idx[2] | ((uint64_t)(uint32_t)idx[3] << 32), bi + 64);
                    if (o.best_dist) {
                        uint16_t* bd = reinterpret_cast<uint16_t*>(o.best_dist + g0 * 4) + lane;
                        stream_store((uint16_t)(dd[0] | (dd[1] << 8)), bd);
                        stream_store((uint16_t)(dd[2] | (dd[3] << 8)), bd + 64);
                    }
                    if (o.n_best) {
                        uint16_t* bn = reinterpret_cast<uint16_t*>(o.n_best + g0 * 4) + lane;
                        stream_store((uint16_t)(nn[0] | (nn[1] << 8)), bn);
                        stream_store((uint16_t)(nn[2] | (nn[3] << 8)), bn + 64);
                    }
                } else {
                    const u32x4 t = {(uint32_t)idx[0], (uint32_t)idx[1], (uint32_t)idx[2], (uint32_t)idx[3]};
                    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(o.best_idx) + g);
                    if (o.best_dist) __builtin_nontemporal_store(dd[0] | (dd[1] << 8) | (dd[2] << 16) | (dd[3] << 24), reinterpret_cast<uint32_t*>(o.best_dist) + g);
                    if (o.n_best) __builtin_nontemporal_store(nn[0] | (nn[1] << 8) | (nn[2] << 16) | (nn[3] << 24), reinterpret_cast<uint32_t*>(o.n_best) + g);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) tally(o, index_of(j), idx[j]);
            }
            if (o.part.rec) {
                const uint32_t w = threadIdx.x >> 6;
                part_emit(o.part, (uint32_t)((g0 * VEC) >> 8), idx, lane, p_stage + w * PART_SLOTS, p_cnt + w * PART_MAX_NR);
            }
        } else {
            if (active) emit(o, g, idx[0], idx[0] >= 0 ? 0u : (uint32_t)BC_DIST_NONE, idx[0] >= 0 ? 1u : 0u);
        }
    }
    // tail (n % VEC reads) by the first threads of block 0; the tally kernel takes these reads from best_idx
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < (n % VEC)) {
        uint64_t i = ngroups * VEC + threadIdx.x;
        uint64_t r = reads[i];
        int32_t idx = read_skipped(r, flags, i, k)
                          ? BC_IDX_SKIP
                          : hash_lookup<WIDE>(r & kmask, s64, s32, guides, k, log_slots, kmask);
        emit(o, i, idx, idx >= 0 ? 0u : (uint32_t)BC_DIST_NONE, idx >= 0 ? 1u : 0u);
    }
}

// ---------------------------------------------------------------------------
// K3a: pigeonhole seed index.  lane = read.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_match_seed(SeedParams sp, const uint64_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ flags, uint64_t n,
                                                      int k, uint64_t kmask, int max_mm, MatchOut o) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint64_t r = reads[i];
        if (read_skipped(r, flags, i, k)) {
            emit(o, i, BC_IDX_SKIP, BC_DIST_NONE, 0);
            continue;
        }
        r &= kmask;
        const uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
        Best b{BC_IDX_MISS, BC_DIST_NONE, 0u, (uint32_t)max_mm};
        for (int s = 0; s < sp.nseg; ++s) {
            uint32_t sv = seed_value(rlo, rhi, sp.start[s], sp.slen[s]);
            uint32_t j = sp.off[s][sv], e = sp.off[s][sv + 1];
            const uint64_t* __restrict__ keys = sp.keys[s];
            for (; j < e; ++j) {
                uint64_t g = keys[j];
                uint32_t mm = (rlo ^ (uint32_t)g) | (rhi ^ (uint32_t)(g >> 32));
                uint32_t d = __popc(mm);
                if (d <= b.cur && !seen_earlier(sp, s, mm)) consider(b, d, sp.gidx[s][j]);
            }
        }
        emit(o, i, b.idx, b.dist, b.nb);
    }
}

// all hits: one record per (query, guide) pair within max_mm.  With a PAM bitmap (queries = genome windows, query
// index = window position): only windows whose adjacent PAM matches are searched at all — the PAM test of
// targets.py:325-352 (extract_*_pam + pam_matches) moved in front of the search instead of after it.
struct PamGate {
    const uint32_t* bits;  // bit p: the PAM (on this strand) starts at forward position p; NULL = no gate
    int64_t shift;         // PAM position = window position + shift
    uint64_t limit;        // positions >= limit cannot hold a whole PAM
    uint64_t wrap;         // circular sequence length (a negative position wraps once), 0 = linear
};

__device__ __forceinline__ bool pam_gate_open(const PamGate& pg, uint64_t i) {
    if (!pg.bits) return true;
    int64_t p = (int64_t)i + pg.shift;
    if (p < 0) {
        if (!pg.wrap) return false;
        p += (int64_t)pg.wrap;
    }
    if (p < 0 || (uint64_t)p >= pg.limit) return false;
    return (pg.bits[p >> 5] >> (p & 31)) & 1u;
}

__global__ __launch_bounds__(BLOCK) void k_match_all(SeedParams sp, const uint64_t* __restrict__ reads,
                                                     const uint8_t* __restrict__ flags, uint64_t n,
                                                     int k, uint64_t kmask, int max_mm, PamGate pg,
                                                     bc_hit* __restrict__ hits, uint64_t cap,
                                                     unsigned long long* n_hits) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint64_t r = reads[i];
        if (read_skipped(r, flags, i, k) || !pam_gate_open(pg, i)) continue;
        r &= kmask;
        const uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
        for (int s = 0; s < sp.nseg; ++s) {
            uint32_t sv = seed_value(rlo, rhi, sp.start[s], sp.slen[s]);
            uint32_t j = sp.off[s][sv], e = sp.off[s][sv + 1];
            for (; j < e; ++j) {
                uint64_t g = sp.keys[s][j];
                uint32_t mm = (rlo ^ (uint32_t)g) | (rhi ^ (uint32_t)(g >> 32));
                uint32_t d = __popc(mm);
                if (d <= (uint32_t)max_mm && !seen_earlier(sp, s, mm)) {
                    unsigned long long at = atomicAdd(n_hits, 1ULL);
                    if (at < cap) {
                        bc_hit h;
                        h.query = (uint32_t)i; h.guide = sp.gidx[s][j]; h.dist = d;
                        hits[at] = h;
                    }
                }
            }
        }
    }
}




// ---------------------------------------------------------------------------
// Building blocks of the wave matcher (k_match_wq below).  A wave works alone (64-thread workgroups, no barriers)
// on 256 consecutive reads that its lanes own, four each; slabs are fetched QUAD-COOPERATIVELY: for the four reads
// owned by the lanes of a quad, load j brings read j's 64-byte slab with lane s taking quarter s — one L1 access
// per slab; lane s scans its quarter for all four reads (rest of read j by DPP broadcast; xor, shift, and-or,
// popcount, shift-or, min per entry) and a transposed DPP min-reduction hands every owner its minimum.  There is
// no hash table: a guide equal to the read sits in the read's seed-0 bucket, so table 0 doubles as the exact-match
// table.  n_best = 1 is proven by the best guide's isolation flags; otherwise ties are counted from the guide's
// neighbour list; the quad exact path (slab_slow) is left for guides with more than NBR_CAP close neighbours.
// Results are staged in the list entries and leave as full-width coalesced stores by the lane that owns the read.
// ---------------------------------------------------------------------------
constexpr int WAVE_T = 256;  // reads per wave (4 per lane)
// layout of k_match_wq's kernel-argument segment (natural alignment, declaration order)
struct WaveKargs {
    const void* spp; const void* reads; const void* flags; uint64_t n; int k; uint64_t kmask; int max_mm; MatchOut o;
};
static_assert(offsetof(WaveKargs, o) == 56, "k_match_wq kernel-argument layout");

// is the best candidate provably the only guide at its distance?  The index word of an entry carries the guide's
// isolation class c (SLAB_CLASS_*, bc_device.h): the largest e <= the table's budget with nn(guide) > 2e.  A best key at
// distance d <= c has no other guide within d of the read (triangle inequality) — and d <= c implies d <= max_mm.
// (Empty and link slots: 0xFFFFFFFx, distance 255 > any class.)
__device__ __forceinline__ bool key_unique(uint32_t key) { return (key >> 24) <= ((key >> SLAB_CLASS_SHIFT) & 3u); }

// rotation inside a DPP quad: lane s takes the value of lane (s + I + 1) & 3 (I = 3: its own)
template <int I>
__device__ __forceinline__ uint32_t quad_rot(uint32_t v) {
    if constexpr (I == 0) return quad_perm<0x39>(v);       // quad_perm:[1,2,3,0]
    else if constexpr (I == 1) return quad_perm<0x4E>(v);  // quad_perm:[2,3,0,1]
    else if constexpr (I == 2) return quad_perm<0x93>(v);  // quad_perm:[3,0,1,2]
    else return v;
}
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return quad_perm<J * 0x55>(v); }

// 16-byte load, global address space (the table pointer comes out of a descriptor in memory; a generic pointer
// would make this a flat load that also waits on the LDS counter), uniform base + 32-bit lane offset
typedef const u32x4 __attribute__((address_space(1)))* gquarter_t;
__device__ __forceinline__ uint4 load16(const char* base, uint32_t off) {
    const u32x4 t = *(gquarter_t)(uintptr_t)(base + off);
    return make_uint4(t.x, t.y, t.z, t.w);
}
// ... piece PIECE (a constant) of the slab at `off`: the piece's 16 * PIECE bytes are added to the 64-bit address, so
// they end up in the instruction's immediate offset (added to the 32-bit lane offset they cost a v_add per load)
template <int PIECE>
__device__ __forceinline__ uint4 load16p(const char* base, uint32_t off) {
    const u32x4 t = ((gquarter_t)(uintptr_t)(base + off))[PIECE];
    return make_uint4(t.x, t.y, t.z, t.w);
}
// wave-wide vote on a lane predicate, straight from the compare's lane mask (HIP's __ballot takes an int: the
// predicate is first turned into 0 / 1 and compared again, two VALU instructions per vote)
__device__ __forceinline__ unsigned long long vote(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull;
#endif
}
// the same for the filter lines and neighbour lists (pointers out of the same descriptor)
typedef const uint32_t __attribute__((address_space(1)))* gword_t;
__device__ __forceinline__ uint32_t gload32(const uint32_t* p, size_t i) { return ((gword_t)(uintptr_t)p)[i]; }
__device__ __forceinline__ uint4 gload128(const uint4* p, size_t i) {
    const u32x4 t = ((gquarter_t)(uintptr_t)p)[i];
    return make_uint4(t.x, t.y, t.z, t.w);
}

// Two entry formats share the slab layout.  NARROW (k - seed <= 16 bases): 8-byte entries {rest, giw}, both planes of
// the rest packed into one dword (hi plane << half); a 16-byte piece holds two.  WIDE (longer rests, k up to 32):
// 16-byte entries {rest lo plane, giw, rest hi plane, 0}; a piece holds one.  In both the last slot of a slab is
// link-or-empty and its first word is the next slab (0 = none).
struct RestW { uint32_t lo, hi; };  // narrow: lo = the packed rest, hi unused
template <bool WIDE>
__device__ __forceinline__ RestW rest_of(uint32_t start, uint32_t sh, uint32_t half, uint32_t rlo, uint32_t rhi) {
    const uint32_t below = (1u << start) - 1u;
    if constexpr (WIDE) {
        const uint32_t up_lo = sh >= 32u ? 0u : (rlo >> sh), up_hi = sh >= 32u ? 0u : (rhi >> sh);
        return RestW{(rlo & below) | (up_lo << start), (rhi & below) | (up_hi << start)};
    } else {
        (void)half;  // narrow entries keep the high plane in the upper 16 bits, whatever the rest's length
        return RestW{((rlo & below) | ((rlo >> sh) << start)) | (((rhi & below) | ((rhi >> sh) << start)) << 16), 0u};
    }
}
// mismatching bases of a narrow rest: x = read ^ entry, low plane in bits 0-15, high plane in bits 16-31 (unused
// bits are zero on both sides).  The two planes are or-ed by sub-dword operand selection: one instruction instead
// of shift + and-or (the scan is a quarter of this VALU-bound kernel's instructions)
__device__ __forceinline__ uint32_t planes_dist(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m;
    asm("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(m) : "v"(x));
    return (uint32_t)__popc(m);
#else
    return (uint32_t)__builtin_popcount((x | (x >> 16)) & 0xFFFFu);
#endif
}
// best key (distance << 24 | giw) among the entries of one 16-byte piece
template <bool WIDE>
__device__ __forceinline__ uint32_t piece_min(const uint4& w, uint32_t rl, uint32_t rh) {
    if constexpr (WIDE) {
        return ((uint32_t)__popc((rl ^ w.x) | (rh ^ w.z)) << 24) | w.y;
    } else {
        (void)rh;
        return min((planes_dist(rl ^ w.x) << 24) | w.y, (planes_dist(rl ^ w.z) << 24) | w.w);
    }
}
template <bool WIDE>
__device__ __forceinline__ uint32_t link_word(const uint4& last_piece) { return WIDE ? last_piece.x : last_piece.z; }
// the eight-entry form of tables 0/1 (EightTables): the index word keeps a spare low byte, the key comes out of one
// v_alignbyte_b32; link16 (slots 6 and 7: the last piece of the slab) = 1 + overflow slab number - buckets, 0 = none
__device__ __forceinline__ uint32_t key8(uint32_t dist, uint32_t giw8) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(dist, giw8, 1u);
#else
    return (dist << 24) | (giw8 >> 8);
#endif
}
__device__ __forceinline__ uint32_t piece_min8(const uint4& w, uint32_t rl) {
    return min(key8(planes_dist(rl ^ w.x), w.y), key8(planes_dist(rl ^ w.z), w.w));
}
__device__ __forceinline__ uint32_t link16_of(const uint4& last_piece) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(last_piece.w, last_piece.y, 0x0C0C0400u);  // byte 0 of slot 6's index word | byte 0 of slot 7's << 8
#else
    return (last_piece.y & 0xFFu) | ((last_piece.w & 0xFFu) << 8);
#endif
}

// Quad-cooperative visit: the four lanes of a quad own one read each; load j (0..3) brings the slab wanted by the read
// of the quad's lane j, lane s taking piece s (+ 4h for longer slabs) — the quad's four 16-byte accesses of one
// instruction fall into ONE 64-byte line: one vector-L1 tag look-up per slab.  (Round 4 tried the rotated form — in
// slot i lane s fetches piece i of the slab of lane (s + i + 1) & 3, which makes the transposed minimum three DPP
// instructions and hands the link slot to its owner for free: VALU 1 284 -> 1 077 per tile, but every instruction then
// touches 64 lines instead of 16: L1 accesses 2.8 -> 6.3 per read and the kernel 1.00 -> 1.52 ms.  The L1's tag rate is
// the third wall; it was not visible in the instruction counts.)
// off = byte offset of the slab wanted by the read this lane owns; s16 = 16 * (lane & 3).  A read that wants nothing
// is pointed at its table's all-empty slab: skipping the load instead (exec-masked branches per read, results masked
// afterwards) measured 6 % slower once the kernel had become VALU-bound.
// (v_add_u32_dpp written as inline assembly would make this one instruction everywhere — the compiler fuses the
// broadcast into the add in some places and emits v_mov_dpp + v_add in others — but the hazard recogniser does not look
// into inline assembly: a DPP operand read within two wait states of the VALU write that produced it returns stale
// data.  Round 4 measured exactly that: 0.6 % wrong results at d = 1 until the assembly was taken out again.)
template <int J>
__device__ __forceinline__ uint32_t quad_bcast_add(uint32_t v, uint32_t add) { return quad_perm<J * 0x55>(v) + add; }
template <int NQ>
__device__ __forceinline__ void visit_issue(const char* base, uint32_t off, uint32_t s16, uint4 (&w)[4][NQ]) {
    auto one = [&](auto jt) {
        constexpr int j = decltype(jt)::value;
        const uint32_t oj = quad_bcast_add<j>(off, s16);
        w[j][0] = load16p<0>(base, oj);
        if constexpr (NQ > 1) w[j][1] = load16p<4>(base, oj);
        static_assert(NQ <= 2, "table 0/1 slabs: one or two pieces per lane");
    };
    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
}

// The transposed reduction of a visit — every owner needs the minimum over the quad's four lanes of THEIR partial
// minimum for its slab, and the link word that sits in lane 3 — goes through LDS when BC_WQ_LDS_REDUCE is set: lane
// 4q + t writes its four partial minima a[j] to rows[68 j + 4q + t] (two ds_write2_b32), owner 4q + s reads the four
// partials of its slab back as one ds_read_b128 at rows[68 s + 4q] (row pitch 68 dwords: the 64 16-byte reads fall
// evenly on the banks) and takes v_min3 + v_min: two VALU instructions instead of the ten of the DPP butterfly
// (four selects, two v_min_dpp, a select, a DPP move, a select, a min); lane 3 of a quad writes the four link words to
// lk[4q + j], the owner reads lk[lane]: no VALU instead of four broadcasts and three selects.  The data movement
// leaves the VALU (the wall) for the LDS pipe (idle in phase 1), at the price of two LDS round trips per visit.
#ifndef BC_WQ_LDS_REDUCE
#define BC_WQ_LDS_REDUCE 1
#endif
constexpr int TR_PITCH = 68;
constexpr int TR_WORDS = 4 * TR_PITCH + 64;  // transposition rows + link words

// lane s of every quad returns min over the quad's lanes of their a[s] (DPP butterfly)
__device__ __forceinline__ uint32_t quad_tmin(const uint32_t (&a)[4], int s) {
    const bool odd = (s & 1) != 0, hi = (s & 2) != 0;
    const uint32_t b0 = min(odd ? a[1] : a[0], quad_perm<0xB1>(odd ? a[0] : a[1]));
    const uint32_t b1 = min(odd ? a[3] : a[2], quad_perm<0xB1>(odd ? a[2] : a[3]));
    return min(hi ? b1 : b0, quad_perm<0x4E>(hi ? b0 : b1));
}

// scan them: kmin = min(kmin, best key of the read this lane owns); link = first word of the last slot of the lane's
// own slab (0 = none: the slot is link-or-empty).  tr: TR_WORDS dwords of LDS private to the wave, 16-byte aligned
template <int NQ, bool WIDE = false, bool E8 = false>
__device__ __forceinline__ void visit_scan(const uint4 (&w)[4][NQ], RestW rest, uint32_t lane, uint32_t* tr,
                                           uint32_t& kmin, uint32_t& link) {
    const int s = (int)(lane & 3u);
    uint32_t a[4], l[4];
    auto one = [&](auto jt) {
        constexpr int j = decltype(jt)::value;
        const uint32_t rl = quad_bcast<j>(rest.lo), rh = WIDE ? quad_bcast<j>(rest.hi) : 0u;
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < NQ; ++h) m = min(m, E8 ? piece_min8(w[j][h], rl) : piece_min<WIDE>(w[j][h], rl, rh));
        a[j] = m;
        l[j] = E8 ? link16_of(w[j][NQ - 1]) : link_word<WIDE>(w[j][NQ - 1]);
    };
    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
    if (BC_WQ_LDS_REDUCE) {
        uint32_t* lk = tr;          // [64] link words (first: every offset below fits the 8-bit dword offsets of ds_write2_b32)
        uint32_t* rows = tr + 64;   // [4][TR_PITCH]
#pragma unroll
        for (int j = 0; j < 4; ++j) rows[TR_PITCH * j + (int)lane] = a[j];
        if (s == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lk[(int)lane - 3 + j] = l[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint4 t = *reinterpret_cast<const uint4*>(rows + TR_PITCH * s + (int)(lane & ~3u));
        link = lk[lane];
        kmin = min(min(kmin, t.x), min(t.y, min(t.z, t.w)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next visit overwrites the rows)
    } else {
        kmin = min(kmin, quad_tmin(a, s));
        const uint32_t t0 = quad_bcast<3>(l[0]), t1 = quad_bcast<3>(l[1]), t2 = quad_bcast<3>(l[2]), t3 = quad_bcast<3>(l[3]);
        link = s == 0 ? t0 : s == 1 ? t1 : s == 2 ? t2 : t3;  // the owner's overflow link
    }
}

// the best guide g (at distance d1) may tie with close neighbours: count them from g's neighbour list.
// false: the list is not available (more than NBR_CAP neighbours) -> exact path.
__device__ __forceinline__ bool resolve_ties(const SlabParams& sp, uint32_t rlo, uint32_t rhi, uint32_t d1,
                                             uint32_t& idx, uint32_t& nb) {
    const uint32_t slot = sp.nbr_slot[idx];
    nb = 1u;
    if (slot == NBR_NONE) return true;
    if (slot == NBR_OVER) return false;
    // the whole list (NBR_CAP x 16 B = one 128-byte line) is requested before any entry is looked at: one memory
    // round trip, not one per entry (unused slots are zero: e.w == 0)
    uint4 e[NBR_CAP];
#pragma unroll
    for (int i = 0; i < NBR_CAP; ++i) e[i] = sp.nbr[(size_t)slot * NBR_CAP + i];
    bool ok = true;
    uint32_t best = idx;
#pragma unroll
    for (int i = 0; i < NBR_CAP; ++i) {
        const uint32_t d = __popc((rlo ^ e[i].x) | (rhi ^ e[i].y));
        const bool used = e[i].w != 0u;
        if (used && d == d1) { ++nb; best = min(best, e[i].z); }
        if (used && d < d1) ok = false;  // cannot happen (every closer guide has been seen); never trust it silently
    }
    idx = best;
    return ok;
}

#ifdef BC_DIAG_BUILD
__device__ unsigned long long g_phase_cycles[8];  // BC_PHASE_PROF=1 (diag build): summed cycles per stage over all waves
#endif

// single-wave workgroups: LDS operations of one wave execute in issue order, so making another lane's list
// entry visible only takes a compiler barrier and the LDS counter — not __syncthreads(), whose fence also waits
// for every outstanding global load and store
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// exact evaluation of one read by one DPP quad over every table of a WIDE index (chains followed, candidates met
// through an earlier seed skipped, counts per distance): the definition of the result, for the rare read whose
// best guide has more close neighbours than a neighbour list holds
template <int NSEG>
__device__ __forceinline__ void slab_slow_wide(const SlabParams& sp, uint32_t rlo, uint32_t rhi, uint32_t max_mm, int s,
                                               uint32_t& key_out, uint32_t& nb_out) {
    uint32_t key = 0xFFFFFFFFu, cur = max_mm;
    uint64_t cnt = 0;  // candidates per distance, 16-bit fields
    for (int t = 0; t < NSEG; ++t) {
        const SlabTable& T = sp.t[t];
        const RestW rest = rest_of<true>(T.start, T.start + T.slen, T.half, rlo, rhi);
        const char* base = reinterpret_cast<const char*>(T.slabs);
        const uint32_t bytes_log2 = T.log2S + 4u, pieces = 1u << (T.log2S - 2u);  // 16-byte entries, pieces per lane
        uint32_t slab = seed_value(rlo, rhi, T.start, T.slen);
        while (slab != 0xFFFFFFFFu) {
            uint32_t next = 0;
            for (uint32_t h = 0; h < pieces; ++h) {
                const uint4 w = load16(base, (slab << bytes_log2) + (uint32_t)(s + 4 * h) * 16u);
                const uint32_t mm = (rest.lo ^ w.x) | (rest.hi ^ w.z);
                const uint32_t d = __popc(mm);
                bool ok = d <= cur && w.y < SLAB_LINK;
                for (int u = 0; u < t; ++u) ok = ok && (mm & T.dmask[u]) != 0u;  // not met through an earlier seed
                if (ok) {
                    key = min(key, (d << 24) | (w.y & SLAB_GIDX_MASK));
                    cur = d;
                    cnt += 1ull << (16 * d);
                }
                if (h == pieces - 1u) next = w.x;
            }
            next = quad_bcast<3>(next);  // lane 3's last piece is the slab's link-or-empty slot
            slab = next ? next : 0xFFFFFFFFu;
        }
        cur = min(cur, quad_perm<0xB1>(cur));
        cur = min(cur, quad_perm<0x4E>(cur));
    }
    key = min(key, quad_perm<0xB1>(key));
    key = min(key, quad_perm<0x4E>(key));
    const uint32_t md = key >> 24;
    uint32_t mine = md <= 3u ? (uint32_t)((cnt >> (16 * (md & 3u))) & 0xFFFFu) : 0u;
    mine += quad_perm<0xB1>(mine);
    mine += quad_perm<0x4E>(mine);
    key_out = key;
    nb_out = mine;
}

// ---------------------------------------------------------------------------
// K2+K3, the wave matcher (k_match_wq; default for d = 1, 2).
// Full-width work stays full width — table 0 for all 256 reads (phase 1), table 1's primary slab for the first 64
// listed reads, the decision pass — but everything that only a FEW reads of a tile need is NOT run as lane-per-read
// batches (round 1's k_match_wave did: overflow slabs of tables 0/1, a second, mostly empty batch of listed reads,
// table 2's 128-byte slabs and their chains — 64 % of that kernel's cycles for 23 % of the reads, because a batch
// costs a full wave's instructions however few lanes have work).  Such work becomes ITEMS {listed read, table, slab}
// on a stack in LDS that the wave drains sixteen at a time, one DPP quad per item: lane s fetches quarter s of the
// slab (one L1 access), scans its two entries, a two-step DPP minimum gives the quad the slab's best key, lane 0
// folds it into the read's running minimum with an LDS atomic, and a link in the slab's last slot becomes a new item.
// Ties are resolved the same way (one quad per read, the best guide's neighbour list two entries per lane).  A read
// reaches table 2 only when no guide within reach has a clean segment 0 or 1; the bucket's filter line
// (SlabParams::flt) then turns away most of the reads that match nothing with one request instead of a slab chain.
//   phase 1   table 0, every read; reads that are not settled are listed, their table-0 overflow link -> item
//   phase 2   table 1 primary slab: first 64 listed reads as one batch (links -> items), further ones as items;
//             drain; decision per listed read: hit / tie list / candidate for table 2 -> filter -> item
//   phase 3   table 2 items drained (128/256-byte slabs: NQ2 pieces per lane), decisions for the candidates;
//             ties by quads; exact path (slab_slow) for guides with more than NBR_CAP close neighbours
//   phase 4   coalesced result stores, tally records
// Measured (1e8 reads, 1e5 guides, d <= 2): 1.14-1.17 ms against 1.27-1.29 ms for k_match_wave on the same boxes;
// VALU 1420 wave-instructions per tile (was 1650), L2 requests 1.92 per read (was 2.07), 54 VGPRs.
// Tried and dropped this round: two tiles per wave sharing one pass of the later phases (the full-width passes
// scale with the listed reads, only partially filled item rounds amortise: -3 % VALU, and the bigger lists cost
// occupancy: 1.27 ms); forcing 8 waves per SIMD (5 KiB of LDS, 1.17 ms: occupancy is not what limits it).
// ---------------------------------------------------------------------------
constexpr int WL_CAP = 384;  // work items a wave can hold; a step that may add 64 drains first when fewer are free

// (amdgpu_waves_per_eu(6) is a register-budget floor — at least six waves per SIMD, i.e. at most 80 VGPRs — not the
// occupancy: the kernel uses about 50 VGPRs and 5.5 KiB of LDS per single-wave workgroup, so 29 workgroups fit a CU's
// 160 KiB: seven waves per SIMD run.)
// LIST: the first stage of a cascade — the reads left without a hit also go on the second stage's list (MatchOut::miss_*).
// Its own instantiation: the default kernel carries none of it (the list entry kept for a miss, the ballots at the end:
// +0.03 ms per 1e8 reads when they sat in the one kernel).
template <int NSEG, int NQ2, bool PROF, bool WIDE, int NQ01, bool LIST = false, bool E8 = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void k_match_wq(const SlabParams* __restrict__ spp,
                                                   const uint64_t* __restrict__ reads,
                                                   const uint8_t* __restrict__ flags, uint64_t n, int k,
                                                   uint64_t kmask, int max_mm_, MatchOut o_arg) {
    static_assert(NSEG == 2 || NSEG == 3, "wave kernel: d = 1 or 2");
    // NQ01: 16-byte pieces per lane of a table-0/1 slab — 1: 64-byte slabs (8 narrow entries); 2: 128-byte slabs
    // (16 narrow entries for buckets above 16 guides, or 8 wide entries)
    static_assert(NQ01 == 1 || NQ01 == 2, "table 0/1 slabs are 64 or 128 bytes");
    static_assert(!WIDE || NQ01 == 2, "wide entries: 8-entry slabs of 128 bytes");
    static_assert(!LIST || !WIDE, "the list form is instantiated for 8-byte entries (64- and 128-byte slabs)");
    // E8: tables 0/1 in their eight-entry form (EightTables): links are 16-bit numbers relative to the table's first
    // overflow slab (lnk_abs makes a slab number of one), list entries and items keep slab numbers
    static_assert(!E8 || (!WIDE && NQ01 == 1), "the eight-entry form exists for 64-byte slabs of 8-byte entries");
    constexpr uint32_t S01 = NQ01 == 2 ? 7u : 6u;     // log2 of that slab's bytes
    (void)o_arg;  // read from the kernel-argument segment in the last phase, see there
    // The list of the tile's unsettled reads: one 16-byte entry per listed read — {read lo, read hi, best key so far
    // (LDS atomic min), aux} — written with one ds_write_b128 in phase 1 and read back with one ds_read_b128 (round 3
    // kept three arrays: three address computations per push and per visit).  aux: until the listed reads' decisions
    // the table-0 overflow link (its item is pushed only if table 1's primary slab does not settle the read); from
    // then on two 16-bit lists threaded through the same words (low half: list entries that need table 2; high half:
    // entries whose best guide may tie with a neighbour, then — compacted in place behind the position being read —
    // those that need the exact path).  Once a read is decided, words 0-1 of its entry hold its result record.
    __shared__ __attribute__((aligned(16))) uint32_t q_ent[WAVE_T * 4];
    // items: list entry | table << 8 | slab << 10.  Phase 1 and table 1's batch use the same words as the rows of their
    // transposed reduction (visit_scan): no item exists before the batch has been scanned
    __shared__ __attribute__((aligned(16))) uint32_t wl[WL_CAP];
    static_assert(WL_CAP >= TR_WORDS, "the item stack doubles as the transposition rows");
    auto e_read = [&](uint32_t qi) -> uint64_t& { return *reinterpret_cast<uint64_t*>(q_ent + 4u * qi); };
    auto e_key = [&](uint32_t qi) -> uint32_t& { return q_ent[4u * qi + 2u]; };
    auto e_all = [&](uint32_t qi) -> uint4& { return *reinterpret_cast<uint4*>(q_ent + 4u * qi); };
    auto l_t2 = [&](uint32_t t) -> uint16_t& { return reinterpret_cast<uint16_t*>(q_ent)[8u * t + 6u]; };
    auto l_tie = [&](uint32_t t) -> uint16_t& { return reinterpret_cast<uint16_t*>(q_ent)[8u * t + 7u]; };
    const SlabParams& sp = *spp;
    const uint32_t max_mm = (uint32_t)max_mm_ & 0xFFu;
#ifdef BC_DIAG_BUILD  // `make diag` only (build/libbarcoder_hip_diag.so, profiles/diag.sh): stages switched off for timing
    const uint32_t diag = ((uint32_t)max_mm_ >> 8) & 0xFFu;  // attribution — results are wrong with any bit set
#else
    constexpr uint32_t diag = 0u;  // the release library has no such mode: every `diag &` below folds away
#endif
    const uint32_t pf_dist = ((uint32_t)max_mm_ >> 16) << 7;  // read-stream prefetch distance in tiles (0 = none), see phase 1
    (void)pf_dist;
    const uint64_t tile0 = (uint64_t)blockIdx.x * WAVE_T;
    const uint32_t tn = (uint32_t)min((uint64_t)WAVE_T, n - tile0);
    const SlabTable& T0 = sp.t[0];
    const SlabTable& T1 = sp.t[1];
    const SlabTable& T2 = sp.t[NSEG - 1];
    const char* const b0 = E8 ? sp.e8.base[0] : reinterpret_cast<const char*>(T0.slabs);
    const char* const b1 = E8 ? sp.e8.base[1] : reinterpret_cast<const char*>(T1.slabs);
    const uint32_t nbm1_0 = E8 ? sp.e8.nb[0] - 1u : 0u, nbm1_1 = E8 ? sp.e8.nb[1] - 1u : 0u;  // slab = link16 + buckets - 1
    const char* const b2 = reinterpret_cast<const char*>(T2.slabs);
    const uint32_t lane = threadIdx.x;
    const uint32_t half0 = T0.half, half1 = T1.half, half2 = T2.half, sh2 = T2.log2S + (WIDE ? 4u : 3u);
    const uint32_t empty0 = (E8 ? sp.e8.empty[0] : T0.empty) << S01, empty1 = (E8 ? sp.e8.empty[1] : T1.empty) << S01,
                   empty2 = T2.empty << sh2;  // what idle lanes fetch
    const int s = (int)(lane & 3u);
    const uint32_t s16 = (lane & 3u) * 16u;
    unsigned long long t_prev = PROF ? (unsigned long long)clock64() : 0ull, t_acc[4] = {0ull, 0ull, 0ull, 0ull};
    auto phase_mark = [&](int ph) {
        if (PROF) {
            const unsigned long long t = (unsigned long long)clock64();
            t_acc[ph] += t - t_prev;
            t_prev = t;
        }
    };
    auto rank_of = [&](unsigned long long m) {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    auto put = [&](uint32_t qi, uint32_t idx, uint32_t dist, uint32_t nb) {
        e_read(qi) = (uint64_t)idx | ((uint64_t)dist << 32) | ((uint64_t)(nb > 255u ? 255u : nb) << 40);
    };
    // no guide within the budget.  LIST: up to 31 bases the listed read stays where it is with bit 63 set (free: the
    // planes end below it, and a result record has zeros there) — the second stage gets the read from there; a
    // 32-mer has no free bit and takes the plain record
    auto put_miss = [&](uint32_t qi) {
        if (!LIST || k > 31) put(qi, (uint32_t)BC_IDX_MISS, BC_DIST_NONE, 0u);
        else __hip_atomic_fetch_or(&e_read(qi), 1ull << 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    uint32_t qn = 0, t2n = 0, sn = 0, wn = 0, tien = 0;  // wave-uniform list lengths

    // a best key whose guide has no neighbour that close is final whatever has not been visited yet: any other guide h
    // lies further than 2e from it, hence further than e from the read (key_unique: class test, triangle inequality)
    auto settled = [&](uint32_t key) { return key_unique(key); };
    auto push_items = [&](bool has, uint32_t item) {
        const unsigned long long m = vote(has);
        if (has) wl[wn + rank_of(m)] = item;
        wn += (uint32_t)__popcll(m);
    };

    // sixteen items per step, one quad each (lane s: piece s); tables 0 and 1
    auto drain01 = [&]() {
        while (wn) {
            wave_lds_sync();
            const uint32_t take = min(wn, 16u);
            wn -= take;
            const uint32_t q = lane >> 2;
            const bool act = q < take;
            const uint32_t item = act ? wl[wn + q] : 0u;
            const uint32_t qi = item & 255u, slab = item >> 10;
            const bool t1 = (item & 0x100u) != 0u;
            const uint64_t r = e_read(qi);
            const uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
            // the rest of the read with the table picked per lane
            const uint32_t st = t1 ? T1.start : 0u, sh = t1 ? T1.start + T1.slen : T0.slen;  // (table 0's seed starts at base 0)
            const RestW rest = rest_of<WIDE>(st, sh, t1 ? half1 : half0, rlo, rhi);
            const char* base = t1 ? b1 : b0;
            const uint32_t off = act ? (slab << S01) : (t1 ? empty1 : empty0);
            uint4 w[NQ01];
#pragma unroll
            for (int h = 0; h < NQ01; ++h) w[h] = load16(base, off + (uint32_t)(s + 4 * h) * 16u);
            uint32_t m = 0xFFFFFFFFu;
#pragma unroll
            for (int h = 0; h < NQ01; ++h) m = min(m, E8 ? piece_min8(w[h], rest.lo) : piece_min<WIDE>(w[h], rest.lo, rest.hi));
            m = min(m, quad_perm<0xB1>(m));
            m = min(m, quad_perm<0x4E>(m));
            uint32_t lnk = quad_bcast<3>(E8 ? link16_of(w[NQ01 - 1]) : link_word<WIDE>(w[NQ01 - 1]));  // the slab's last slot(s): link or none (0)
            if (E8) lnk = lnk ? lnk + (t1 ? nbm1_1 : nbm1_0) : 0u;
            const bool lead = act && s == 0;
            uint32_t now = 0xFFFFFFFFu;
            if (lead) now = min(m, atomicMin(&e_key(qi), m));
            push_items(lead && lnk != 0u && !settled(now), (item & 0x3FFu) | (lnk << 10));  // a settled read's chain ends here
        }
        wave_lds_sync();
    };

    // table 2: NQ2 pieces per lane (128- or 256-byte slabs)
    auto drain2 = [&]() {
        while (wn) {
            wave_lds_sync();
            const uint32_t take = min(wn, 16u);
            wn -= take;
            const uint32_t q = lane >> 2;
            const bool act = q < take;
            const uint32_t item = act ? wl[wn + q] : 0u;
            const uint32_t qi = item & 255u, slab = item >> 10;
            const uint64_t r = e_read(qi);
            const RestW rest = rest_of<WIDE>(T2.start, T2.start + T2.slen, half2, (uint32_t)r, (uint32_t)(r >> 32));
            const uint32_t off = act ? (slab << sh2) : empty2;
            uint4 w[NQ2];
#pragma unroll
            for (int h = 0; h < NQ2; ++h) w[h] = load16(b2, off + (uint32_t)(s + 4 * h) * 16u);
            uint32_t m = 0xFFFFFFFFu;
#pragma unroll
            for (int h = 0; h < NQ2; ++h) m = min(m, piece_min<WIDE>(w[h], rest.lo, rest.hi));
            m = min(m, quad_perm<0xB1>(m));
            m = min(m, quad_perm<0x4E>(m));
            const uint32_t lnk = quad_bcast<3>(link_word<WIDE>(w[NQ2 - 1]));
            const bool lead = act && s == 0;
            uint32_t now = 0xFFFFFFFFu;
            if (lead) now = min(m, atomicMin(&e_key(qi), m));
            push_items(lead && lnk != 0u && !settled(now), (item & 0x3FFu) | (lnk << 10));
        }
        wave_lds_sync();
    };

    // hit / tie / miss decision for the lane's read once every guide that can matter has been seen.  A best guide
    // whose isolation class does not rule out a second guide at the same distance goes on the tie list (resolved by
    // quads from the guide's neighbour list after all decisions: one pass per tile instead of one per decision batch)
    auto decide = [&](bool mine, bool final_miss, uint32_t qi, uint32_t kmin) {
        const uint32_t d = kmin >> 24;
        const bool inrange = mine && d <= max_mm;
        const bool tie = inrange && !key_unique(kmin) && !(diag & 4u);
        if (inrange && !tie) put(qi, kmin & SLAB_GIDX_MASK, d, 1u);
        if (mine && !inrange && final_miss) put_miss(qi);
        const unsigned long long m = vote(tie);
        if (tie) l_tie(tien + rank_of(m)) = (uint16_t)qi;
        tien += (uint32_t)__popcll(m);
        return inrange;
    };

    // ---- read-stream prefetch (BC_PREFETCH = distance / 128 tiles, default 8; 0 = none).  With the read stream served
    // from L2 the kernel runs 0.84 instead of 1.07 ms (BC_DIAG=16): the stream's HBM round trips cost a fifth of its time,
    // because a line of the stream holds a vector-L1 miss slot for an HBM round trip where a table slab holds one for an
    // L2 hit.  Workgroups start in id order, ~365 per microsecond on the whole chip, so the workgroup pf_dist ids after
    // this one (a multiple of 8: same XCD, same L2 — workgroup i runs on XCD i % 8, profiles/xcc_map.hip) starts
    // pf_dist / 365 us from now: pull its tile into the L2 now, through the SCALAR cache — thirty-two 4-byte loads, one
    // per 64-byte sector (the L2 fills sectors), which occupy no vector-L1 miss slot and are awaited with phase 1's own
    // waits.  Measured (profiles/r02_prefetch_sweep.txt): at 512-1024 tiles 62 % of the stream's sectors are in L2 when
    // the demand load comes (the L2 keeps a line it has seen once for a few microseconds only: 29 % at 2048 tiles, 9 %
    // beyond 4096), mean L1->L2 latency 230 -> 167 cycles, kernel -5...-7 %; the sectors that were evicted again are
    // fetched from HBM twice (+0.3 GB per pass).
    uint32_t pfv[32];
#if defined(__HIP_DEVICE_COMPILE__)
    {
        // (a tile past the end of the input: the last tile again — harmless, and no branch around the loads)
        const uint64_t last = (n - 1) / WAVE_T;
        const uint64_t pt = pf_dist ? min((uint64_t)blockIdx.x + pf_dist, last ? last - 1 : 0) : 0;
        typedef const uint32_t __attribute__((address_space(4)))* cu32_t;
        cu32_t pl = (cu32_t)(uintptr_t)(reads + pt * WAVE_T);
        if (pf_dist) {
#pragma unroll
            for (int i = 0; i < 32; ++i) pfv[i] = pl[i * 16];
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) pfv[i] = 0;
        }
    }
#else
    for (int i = 0; i < 32; ++i) pfv[i] = 0;
#endif
    // ---- phase 1: table 0, primary slab, every read (exact hits and isolated near hits are final)
    // Which four reads of the tile a lane owns.  A FULL tile (every tile but a ragged last one): reads 2l, 2l + 1 and
    // 128 + 2l, 129 + 2l — so that each of the two 16-byte-per-lane load instructions covers 1 KiB of the tile CONTIGUOUSLY
    // and every line of the stream is requested once.  (Round 3 gave lane l reads 4l .. 4l + 3: both load instructions then
    // touch every 64-byte sector of the tile, half of it each, and the vector L1 asked the L2 for every sector twice —
    // 0.125 instead of 0.0625 requests per read of the most expensive kind, the ones that wait for HBM: round 4 measured
    // 1.682 -> 1.619 requests per read and -3 % time with the loads rearranged, profiles/r04_stream_load_contiguous.txt.)
    // A ragged tile keeps the consecutive assignment (element-wise loads and stores).
    const bool whole = tn == (uint32_t)WAVE_T;  // wave-uniform
    const uint32_t p0 = lane * 4u;
    const bool full = whole;
    auto pos_of = [&](int j) -> uint32_t { return whole ? ((j < 2 ? 0u : 128u) + 2u * lane + (uint32_t)(j & 1)) : p0 + (uint32_t)j; };
    int32_t idx[4];
    uint32_t at[4];     // list entry of the lane's reads that were listed
    uint32_t dist4 = 0; // distances of the reads settled in phase 1, one byte each
    {
        uint64_t r[4] = {0, 0, 0, 0};
        if (full) {
            // (BC_DIAG 16: the read stream from 512 KB that stay in L2 — timing experiment)
            load_reads2x2(reads + ((diag & 16u) ? (uint64_t)(blockIdx.x & 255u) * WAVE_T : tile0), lane, r);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < tn) r[j] = reads[tile0 + p0 + j];
        }
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ok[j] = pos_of(j) < tn && !read_skipped(r[j], flags, tile0 + pos_of(j), k);
            r[j] &= kmask;
        }
        // skipped reads fetch bucket 0 (always mapped) and ignore it: no branch around the loads
        // table 0's seed starts at base 0 (build_seed_index; checked when the launch is chosen): with the start a
        // constant the seed value is two and-s and a shift-or, the rest a shift per plane — phase 1 runs for every read
        auto off0 = [&](int j) { return ok[j] ? (seed_value((uint32_t)r[j], (uint32_t)(r[j] >> 32), 0, T0.slen) << S01) : 0u; };
        auto finish = [&](int j, const uint4 (&w)[4][NQ01]) {
            uint32_t lnk, key = 0xFFFFFFFFu;
            visit_scan<NQ01, WIDE, E8>(w, rest_of<WIDE>(0u, T0.slen, half0, (uint32_t)r[j], (uint32_t)(r[j] >> 32)), lane, wl, key, lnk);
            const bool fin = ok[j] && key_unique(key);
            idx[j] = !ok[j] ? BC_IDX_SKIP : fin ? (int32_t)(key & SLAB_GIDX_MASK) : BC_IDX_MISS;
            dist4 |= (fin ? (key >> 24) : 0u) << (8 * j);
            const bool push = ok[j] && !fin;
            const unsigned long long m = vote(push);
            at[j] = push ? qn + rank_of(m) : 0xFFFFFFFFu;
            // lnk: table 0's overflow slab, visited only if table 1 does not settle the read
            if (push) e_all(at[j]) = make_uint4((uint32_t)r[j], (uint32_t)(r[j] >> 32), key, lnk);
            qn += (uint32_t)__popcll(m);
        };
        if constexpr (NQ01 == 1) {
            uint4 wa[4][1], wb[4][1];  // two rounds (2 x 4 reads per quad) in flight
            visit_issue<1>(b0, off0(0), s16, wa);
            visit_issue<1>(b0, off0(1), s16, wb);
            finish(0, wa);
            visit_issue<1>(b0, off0(2), s16, wa);
            finish(1, wb);
            visit_issue<1>(b0, off0(3), s16, wb);
            finish(2, wa);
            finish(3, wb);
        } else {
            uint4 wa[4][2];  // 128-byte slabs: one round in flight (two would take 64 registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                visit_issue<2>(b0, off0(j), s16, wa);
                finish(j, wa);
            }
        }
    }
    wave_lds_sync();
    if (pf_dist) {  // the prefetch loads have long returned: their registers are free again
        uint32_t pf_acc = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) pf_acc ^= pfv[i];
        asm volatile("" ::"s"(pf_acc));
    }
    phase_mark(0);

    // ---- phase 2: table 1's primary slab — the first 64 listed reads as one full-width batch, the rest as items
    if (qn && !(diag & 2u)) {
        const uint32_t qi = lane;
        const bool mine = qi < qn;
        uint32_t kmin = 0xFFFFFFFFu, off1 = empty1, lnk0 = 0;
        RestW rest1{0u, 0u};
        if (mine) {
            const uint4 e = e_all(qi);
            kmin = e.z;
            lnk0 = e.w;
            rest1 = rest_of<WIDE>(T1.start, T1.start + T1.slen, half1, e.x, e.y);
            off1 = seed_value(e.x, e.y, T1.start, T1.slen) << S01;
        }
        uint4 wa[4][NQ01];
        uint32_t lnk;
        visit_issue<NQ01>(b1, off1, s16, wa);
        visit_scan<NQ01, WIDE, E8>(wa, rest1, lane, wl, kmin, lnk);
        // (the item stack shares its words with the transposition rows: the visit above is complete — it ends on its
        // own LDS wait — before the first push below)
        // a read that table 1's primary slab settles (a third of the listed ones) needs neither overflow slab
        const bool more = mine && !settled(kmin);
        if (mine) e_key(qi) = kmin;
        // (E8: the links are 16-bit numbers relative to the table's first overflow slab — the entry kept table 0's as it
        // came: three quarters of the listed reads never turn it into a slab number)
        push_items(more && lnk0 != 0u, qi | ((lnk0 + nbm1_0) << 10));
        push_items(more && lnk != 0u, qi | 0x100u | ((lnk + nbm1_1) << 10));
        for (uint32_t qb = 64; qb < qn; qb += 64) {
            if (wn > WL_CAP - 128) drain01();
            const uint32_t qj = qb + lane;
            const bool mj = qj < qn;
            uint32_t sv = 0, lj = 0;
            if (mj) {
                const uint4 e = e_all(qj);
                sv = seed_value(e.x, e.y, T1.start, T1.slen);
                lj = e.w;
            }
            push_items(mj, qj | 0x100u | (sv << 10));
            push_items(mj && lj != 0u, qj | ((lj + nbm1_0) << 10));
        }
    }
    if (!(diag & 2u)) drain01();
    wn = 0;
    // seen so far: every guide with a clean segment 0 or 1, i.e. every guide within distance 1 (NSEG = 3) /
    // within max_mm (NSEG = 2); unseen guides are at distance >= 2.  A read with nothing in range is a candidate
    // for table 2: its bucket's filter line first (one request) — most such reads match nothing and leave here
    const uint32_t* const flt = (NSEG == 3 && !(diag & 8u)) ? sp.flt : nullptr;
    for (uint32_t qb = 0; qb < qn; qb += 64) {
        if (NSEG == 3 && wn > WL_CAP - 64) drain2();
        const uint32_t qi = qb + lane;
        const bool mine = qi < qn;
        uint32_t rlo = 0, rhi = 0, kmin = 0xFFFFFFFFu;
        if (mine) {
            const uint4 e = e_all(qi);
            rlo = e.x; rhi = e.y;
            kmin = e.z;
        }
        const bool inrange = decide(mine, NSEG != 3, qi, kmin);
        if (NSEG == 3) {
            bool pass = mine && !inrange && !(diag & 1u);
            const uint32_t sv = seed_value(rlo, rhi, T2.start, T2.slen);
            if (pass && flt) {
                const uint32_t* line = flt + (size_t)sv * 16u;
                uint32_t hit[4];
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const uint32_t bit = flt_hash7(seed_value(rlo, rhi, sp.f_pos[hh], sp.f_len[hh]));
                    hit[hh] = (gload32(line, hh * 4 + (bit >> 5)) >> (bit & 31u)) & 1u;
                }
                pass = ((hit[0] | hit[1]) & (hit[2] | hit[3])) != 0u;
            }
            if (mine && !inrange && !pass) put_miss(qi);
            const unsigned long long m = vote(pass);
            if (pass) l_t2(t2n + rank_of(m)) = (uint16_t)qi;
            t2n += (uint32_t)__popcll(m);
            push_items(pass, qi | 0x200u | (sv << 10));
        }
    }
    wave_lds_sync();
    phase_mark(1);

    // ---- phase 3: table 2 items, then the decisions for the candidates
    if (NSEG == 3 && t2n) {
        drain2();
        // every guide within max_mm of these reads has segments 0 and 1 dirty, hence a clean segment 2: all seen
        for (uint32_t tb = 0; tb < t2n; tb += 64) {
            const uint32_t ti = tb + lane;
            const bool mine = ti < t2n;
            uint32_t qi = 0, kmin = 0xFFFFFFFFu;
            if (mine) {
                qi = l_t2(ti);
                kmin = e_key(qi);
            }
            decide(mine, true, qi, kmin);
        }
        wave_lds_sync();
    }
    // ---- ties: one quad per listed read whose best guide has neighbours within reach.  The guide's list (at most
    // NBR_CAP entries, one 128-byte line) is fetched two entries per lane; every guide that ties with it is in there
    for (uint32_t tb = 0; tb < tien; tb += 16) {
        const uint32_t ti = tb + (lane >> 2);
        const bool act = ti < tien;
        const uint32_t qi = act ? l_tie(ti) : 0u;
        const uint4 e = e_all(qi);
        const uint32_t rlo = e.x, rhi = e.y, kmin = e.z;
        const uint32_t d1 = kmin >> 24, g0 = kmin & SLAB_GIDX_MASK;
        uint32_t slot = NBR_NONE;
        if (act && sp.nbr_slot) slot = gload32(sp.nbr_slot, g0);
        const bool listed = slot < NBR_OVER;
        uint32_t cnt = 0, best = g0;
        if (listed) {
            const uint4 e0 = gload128(sp.nbr, (size_t)slot * NBR_CAP + s), e1 = gload128(sp.nbr, (size_t)slot * NBR_CAP + s + 4);
            const bool t0 = e0.w != 0u && (uint32_t)__popc((rlo ^ e0.x) | (rhi ^ e0.y)) == d1;
            const bool t1 = e1.w != 0u && (uint32_t)__popc((rlo ^ e1.x) | (rhi ^ e1.y)) == d1;
            cnt = (t0 ? 1u : 0u) + (t1 ? 1u : 0u);
            best = min(t0 ? e0.z : g0, t1 ? e1.z : g0);
        }
        cnt += quad_perm<0xB1>(cnt); cnt += quad_perm<0x4E>(cnt);
        best = min(best, quad_perm<0xB1>(best)); best = min(best, quad_perm<0x4E>(best));
        // without lists (or with more neighbours than a list holds) the exact path decides
        const bool slow = act && (!sp.nbr_slot || slot == NBR_OVER);
        // (the list of exact-path entries is compacted into the tie list's own words: position sn + rank <= ti, and
        // every quad has read its l_tie word above — LDS operations of one wave execute in order)
        const unsigned long long m = vote(slow && s == 0);
        if (slow && s == 0) l_tie(sn + rank_of(m)) = (uint16_t)qi;
        if (act && !slow && s == 0) put(qi, best, d1, 1u + cnt);
        sn += (uint32_t)__popcll(m);
    }
    wave_lds_sync();
    phase_mark(2);

    // ---- phase 4a: exact evaluation by DPP quads (guides with more than NBR_CAP close neighbours only)
    if (sn) {
        const int last_lane = (int)(lane | 3u);
        for (uint32_t si = lane >> 2; si < sn; si += 16) {
            const uint32_t qi = l_tie(si);
            const uint64_t r = e_read(qi);
            uint32_t key, nb;
            if constexpr (WIDE) slab_slow_wide<NSEG>(sp, (uint32_t)r, (uint32_t)(r >> 32), max_mm, s, key, nb);
            else slab_slow<4, NSEG>(sp, (uint32_t)r, (uint32_t)(r >> 32), (int)max_mm, s, last_lane, key, nb);
            if (s == 0) {
                if ((key >> 24) <= max_mm) put(qi, key & SLAB_GIDX_MASK, key >> 24, nb);
                else put_miss(qi);
            }
        }
        wave_lds_sync();
    }

    // ---- phase 4b: every lane collects its four reads' results; full-width coalesced stores, nothing is patched
    {
        // The output descriptor (22 dwords) is needed from here on only.  Left to itself the compiler loads it at
        // kernel entry and keeps it in SGPRs through every phase, spilling table descriptors around it (v_writelane /
        // v_readlane count against the VALU issue this kernel is bound by): fetch it from the kernel-argument
        // segment now, through a pointer the optimiser cannot see through.
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const MatchOut __attribute__((address_space(4)))* karg_out_t;
        const char __attribute__((address_space(4)))* ka =
            (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const MatchOut o = *(karg_out_t)(ka + offsetof(WaveKargs, o));
#else
        const MatchOut o = o_arg;  // host pass of the single-source compile: never executed
#endif
        uint32_t dist[4], nb[4];
        uint64_t mr[4] = {0, 0, 0, 0};  // the reads that found nothing (narrow entries: see put_miss)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dist[j] = idx[j] >= 0 ? ((dist4 >> (8 * j)) & 0xFFu) : (uint32_t)BC_DIST_NONE;
            nb[j] = idx[j] >= 0 ? 1u : 0u;
            if (at[j] != 0xFFFFFFFFu) {
                const uint64_t e = e_read(at[j]);
                idx[j] = (int32_t)(uint32_t)e; dist[j] = (uint32_t)(e >> 32) & 0xFFu; nb[j] = (uint32_t)(e >> 40) & 0xFFu;
                if (LIST && k <= 31 && (e >> 63)) { idx[j] = BC_IDX_MISS; dist[j] = (uint32_t)BC_DIST_NONE; nb[j] = 0u; mr[j] = e & ~(1ull << 63); }
            }
        }
        if (full) {
            // non-temporal: nothing in this kernel touches these lines again, the tables should keep the L2 (3 % measured).
            // The lane's reads are two pairs (2l, 2l + 1 and 128 + 2l, 129 + 2l): two 8-byte index stores and two 2-byte
            // stores per byte array, each instruction a contiguous 512 / 128 bytes of whole 64-byte sectors
            const uint64_t st0 = (diag & 64u) ? (uint64_t)(blockIdx.x & 255u) * WAVE_T : tile0;  // BC_DIAG 64: results into 0.4 MB (timing experiment)
            uint64_t* bi = reinterpret_cast<uint64_t*>(o.best_idx + st0) + lane;
            stream_store((uint64_t)(uint32_t)idx[0] | ((uint64_t)(uint32_t)idx[1] << 32), bi);
            stream_store((uint64_t)(uint32_t)idx[2] | ((uint64_t)(uint32_t)idx[3] << 32), bi + 64);
            if (o.best_dist) {
                uint16_t* bd = reinterpret_cast<uint16_t*>(o.best_dist + st0) + lane;
                stream_store((uint16_t)(dist[0] | (dist[1] << 8)), bd);
                stream_store((uint16_t)(dist[2] | (dist[3] << 8)), bd + 64);
            }
            if (o.n_best) {
                uint16_t* bn = reinterpret_cast<uint16_t*>(o.n_best + st0) + lane;
                stream_store((uint16_t)(nb[0] | (nb[1] << 8)), bn);
                stream_store((uint16_t)(nb[2] | (nb[3] << 8)), bn + 64);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < tn) {
                    o.best_idx[tile0 + p0 + j] = idx[j];
                    if (o.best_dist) o.best_dist[tile0 + p0 + j] = (uint8_t)dist[j];
                    if (o.n_best) o.n_best[tile0 + p0 + j] = (uint8_t)nb[j];
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (pos_of(j) < tn) tally(o, tile0 + pos_of(j), idx[j]);
        if (o.part.rec) {
            // the lists are dead (every lane has its results in registers): the list entries become the record staging
            // area (the first 2 KiB = PART_SLOTS two-byte slots), the item stack the range counters
            static_assert(sizeof(q_ent) >= PART_SLOTS * sizeof(uint16_t) && WL_CAP >= PART_MAX_NR, "record staging");
            wave_lds_sync();
            part_emit(o.part, blockIdx.x, idx, lane, reinterpret_cast<uint16_t*>(q_ent), wl);
        }
        if (LIST && o.miss_cnt) {
            // first stage of a cascade: the reads left without a hit go on the second stage's list (position, read), in
            // this tile's own 256 entries — no atomic, and the read is still in the list entry (put_miss; a cascade
            // exists for k <= 25 only.  Fetching it from memory again made 24 more requests per tile: +0.2 ms)
            uint32_t off[4], mtot = 0;
            bool ms[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ms[j] = pos_of(j) < tn && idx[j] == BC_IDX_MISS;
                const unsigned long long m = vote(ms[j]);
                off[j] = mtot + rank_of(m);
                mtot += (uint32_t)__popcll(m);
            }
            if (lane == 0) o.miss_cnt[blockIdx.x] = mtot;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ms[j]) {
                    o.miss_pos[tile0 + off[j]] = (uint32_t)(tile0 + pos_of(j));
                    o.miss_read[tile0 + off[j]] = k > 31 ? (reads[tile0 + pos_of(j)] & kmask) : mr[j];
                }
        }
    }
    phase_mark(3);
#ifdef BC_DIAG_BUILD
    if (PROF && threadIdx.x == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_phase_cycles[i], t_acc[i]);
        atomicAdd(&g_phase_cycles[7], 1ull);
    }
#else
    static_assert(!PROF, "per-phase cycle sums exist in the diag build only");
#endif
}

// ---------------------------------------------------------------------------
// K5: per-guide histogram + first-seen from best_idx, privatised in LDS.
// Scattered device-scope atomics leave the L2 (one 32-byte memory-side write
// each, measured: profiles/r01_v1_*); instead each workgroup owns a range of
// HIST_R guides in LDS (count + first read index), sweeps a slice of best_idx,
// and flushes its range with contiguous atomics.  grid = passes x slices.
// ---------------------------------------------------------------------------
constexpr int HIST_R = 20480;          // guides per pass: 2 x 4 B x 20480 = 160 KiB of LDS
constexpr int HIST_BLOCK = 1024;

__global__ __launch_bounds__(HIST_BLOCK) void k_hist(const int32_t* __restrict__ best_idx, uint64_t n,
                                                     uint32_t G, int slices, uint32_t* __restrict__ counts,
                                                     uint64_t* __restrict__ first_seen, uint64_t base) {
    extern __shared__ uint32_t lds[];
    uint32_t* cnt = lds;
    uint32_t* fs = lds + HIST_R;
    const int pass = blockIdx.x / slices, slice = blockIdx.x % slices;
    const uint32_t lo = (uint32_t)pass * HIST_R;
    const uint32_t width = min((uint32_t)HIST_R, G - lo);
    for (uint32_t j = threadIdx.x; j < width; j += HIST_BLOCK) { cnt[j] = 0; fs[j] = 0xFFFFFFFFu; }
    __syncthreads();
    // slice of reads, 4 indices (16 B) per lane per step
    const uint64_t n4 = n / 4;
    const uint64_t per = (n4 + slices - 1) / slices;
    const uint64_t a = (uint64_t)slice * per, e = min(n4, a + per);
    const u32x4* pv = reinterpret_cast<const u32x4*>(best_idx);
    auto take = [&](const u32x4& v, uint64_t q) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t g = vv[c] - lo;  // negative sentinels wrap far above width
            if (g < width) {
                atomicAdd(&cnt[g], 1u);
                if (first_seen) atomicMin(&fs[g], (uint32_t)(q * 4 + c));
            }
        }
    };
    uint64_t q = a + threadIdx.x;
    for (; q + 3 * HIST_BLOCK < e; q += 4 * HIST_BLOCK) {  // four loads in flight
        const u32x4 v0 = __builtin_nontemporal_load(pv + q), v1 = __builtin_nontemporal_load(pv + q + HIST_BLOCK);
        const u32x4 v2 = __builtin_nontemporal_load(pv + q + 2 * HIST_BLOCK), v3 = __builtin_nontemporal_load(pv + q + 3 * HIST_BLOCK);
        take(v0, q);
        take(v1, q + HIST_BLOCK);
        take(v2, q + 2 * HIST_BLOCK);
        take(v3, q + 3 * HIST_BLOCK);
    }
    for (; q < e; q += HIST_BLOCK) take(__builtin_nontemporal_load(pv + q), q);
    if (slice == slices - 1)  // tail reads (n % 4)
        for (uint64_t i = n4 * 4 + threadIdx.x; i < n; i += HIST_BLOCK) {
            uint32_t g = (uint32_t)best_idx[i] - lo;
            if (g < width) {
                atomicAdd(&cnt[g], 1u);
                if (first_seen) atomicMin(&fs[g], (uint32_t)i);
            }
        }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < width; j += HIST_BLOCK) {
        uint32_t c = cnt[j];
        if (c) {
            if (counts) atomicAdd(&counts[lo + j], c);
            if (first_seen) atomicMin((unsigned long long*)&first_seen[lo + j], (unsigned long long)(base + fs[j]));
        }
    }
}

// ---------------------------------------------------------------------------
// K5': the tally from the matcher's range-partitioned records (PartOut).  Workgroup (range p, slice s) owns the
// range's R guides in LDS (count + first TILE seen) and streams the records rec[p][tiles of slice s] — whole
// 128-byte lines, 8 records per lane and load — then the tiles on the overflow list and the reads past the last
// full group of four (both from best_idx, filtered to its range), and flushes with contiguous atomics: counts
// into counts[], first tiles into first_tile[] (a min over slices).  k_first_refine then turns a guide's first
// tile into its first read by scanning that tile's 256 results, and leaves first_tile[] / the overflow counter
// reset for the next launch.  Bytes per read: 2.5 (records) instead of 4 x ranges (five sweeps of best_idx).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(HIST_BLOCK) void k_hist_part(PartOut P, const int32_t* __restrict__ best_idx, uint64_t n,
                                                          uint64_t tail_begin, uint32_t G, int slices,
                                                          uint32_t* __restrict__ counts,
                                                          uint32_t* __restrict__ first_tile) {
    extern __shared__ uint32_t lds[];
    uint32_t* cnt = lds;
    uint32_t* ft = lds + HIST_R;
    const uint32_t range = blockIdx.x / slices, slice = blockIdx.x % slices;
    const uint32_t lo = range * P.R;
    const uint32_t width = min(P.R, G - lo);
    for (uint32_t j = threadIdx.x; j < width; j += HIST_BLOCK) { cnt[j] = 0; ft[j] = 0xFFFFFFFFu; }
    __syncthreads();
    // records: a tile's line is 8, 4 or 2 pieces of 16 bytes (64, 32 or 16 records), tiles of this slice
    const uint64_t per = ((uint64_t)P.n_tiles + slices - 1) / slices;
    const uint64_t t0 = (uint64_t)slice * per, t1 = min((uint64_t)P.n_tiles, t0 + per);
    const uint32_t pc_log2 = P.cap_log2 - 3u;  // log2 of the pieces per line
    const u32x4* pv = reinterpret_cast<const u32x4*>(P.rec + (((size_t)range * P.n_tiles) << P.cap_log2));
    auto take = [&](const u32x4& v, uint32_t tile) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t g = (vv[c] >> (16 * h)) & 0xFFFFu;
                if (g < width) {  // 0xFFFF = empty
                    atomicAdd(&cnt[g], 1u);
                    if (first_tile) atomicMin(&ft[g], tile);
                }
            }
        }
    };
    const uint64_t q0 = t0 << pc_log2, q1 = t1 << pc_log2;  // 16-byte pieces
    uint64_t q = q0 + threadIdx.x;
    for (; q + 3 * HIST_BLOCK < q1; q += 4 * HIST_BLOCK) {  // four loads in flight
        const u32x4 v0 = __builtin_nontemporal_load(pv + q), v1 = __builtin_nontemporal_load(pv + q + HIST_BLOCK);
        const u32x4 v2 = __builtin_nontemporal_load(pv + q + 2 * HIST_BLOCK), v3 = __builtin_nontemporal_load(pv + q + 3 * HIST_BLOCK);
        take(v0, (uint32_t)(q >> pc_log2));
        take(v1, (uint32_t)((q + HIST_BLOCK) >> pc_log2));
        take(v2, (uint32_t)((q + 2 * HIST_BLOCK) >> pc_log2));
        take(v3, (uint32_t)((q + 3 * HIST_BLOCK) >> pc_log2));
    }
    for (; q < q1; q += HIST_BLOCK) take(__builtin_nontemporal_load(pv + q), (uint32_t)(q >> pc_log2));
    // tiles whose records did not fit (overflow flag): straight from best_idx
    auto take_idx = [&](int32_t v, uint32_t tile) {
        const uint32_t g = (uint32_t)v - lo;  // negative sentinels wrap far above width
        if (g < width) {
            atomicAdd(&cnt[g], 1u);
            if (first_tile) atomicMin(&ft[g], tile);
        }
    };
    // every wave walks its share of the slice's tiles and takes the flagged ones (256 results, four per lane) itself
    {
        const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
        for (uint64_t c = t0 + (uint64_t)wv * 64; c < t1; c += HIST_BLOCK) {  // 64 tiles' flags per load
            const uint64_t tl = c + ln;
            unsigned long long m = __ballot(tl < t1 && P.ovf_flag[tl] != 0u);
            while (m) {  // wave-uniform
                const uint64_t t = c + (uint64_t)(__ffsll((long long)m) - 1);
                m &= m - 1;
                const uint64_t a = t << 8, b = min(tail_begin, a + 256);
                for (uint64_t i = a + ln; i < b; i += 64) take_idx(best_idx[i], (uint32_t)t);
            }
        }
    }
    if (slice == (uint32_t)slices - 1)  // reads the matcher's record path did not cover
        for (uint64_t i = tail_begin + threadIdx.x; i < n; i += HIST_BLOCK) take_idx(best_idx[i], (uint32_t)(i >> 8));
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < width; j += HIST_BLOCK) {
        const uint32_t c = cnt[j];
        if (c) {
            if (counts) atomicAdd(&counts[lo + j], c);
            if (first_tile) atomicMin(&first_tile[lo + j], ft[j]);
        }
    }
}

// one quarter-wave (16 lanes x 16 results) per guide: first read of the guide inside its first tile
__global__ __launch_bounds__(BLOCK) void k_first_refine(const int32_t* __restrict__ best_idx, uint64_t n, uint32_t G,
                                                        uint32_t* __restrict__ first_tile,
                                                        uint64_t* __restrict__ first_seen, uint64_t base,
                                                        uint32_t* __restrict__ ovf_count,
                                                        uint32_t* __restrict__ ovf_flag, uint32_t n_tiles) {
    (void)ovf_count;
    // the tiles' overflow flags go back to zero for the next pass (the matcher only ever sets them)
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < n_tiles; t += gridDim.x * BLOCK)
        if (ovf_flag[t] != 0u) ovf_flag[t] = 0u;
    const uint32_t g = (blockIdx.x * BLOCK + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (g >= G) return;
    const uint32_t tile = first_tile[g];
    if (tile == 0xFFFFFFFFu) return;  // uniform over the 16 lanes
    const uint64_t a = (uint64_t)tile << 8;
    uint32_t pos = 0xFFFFFFFFu;
#pragma unroll 4
    for (uint32_t j = 0; j < 16; ++j) {
        const uint64_t i = a + j * 16u + sub;
        if (i < n && best_idx[i] == (int32_t)g) pos = min(pos, j * 16u + sub);
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) pos = min(pos, (uint32_t)__shfl_xor((int)pos, m));
    if (sub == 0) {
        first_tile[g] = 0xFFFFFFFFu;
        if (pos != 0xFFFFFFFFu) atomicMin((unsigned long long*)&first_seen[g], (unsigned long long)(base + a + pos));
    }
}

// ---------------------------------------------------------------------------
// K3b: brute force.  Guide tile in LDS (broadcast reads), RPT reads per lane
// held in registers, 5 VALU per pair (xor, xor, or, bcnt, cmp).
// ---------------------------------------------------------------------------
constexpr int BRUTE_TILE = 2048;  // guides per LDS tile (16 KiB)
constexpr int BRUTE_RPT = 4;      // reads per thread

__global__ __launch_bounds__(BLOCK) void k_match_brute(const uint64_t* __restrict__ guides, uint32_t G,
                                                       const uint64_t* __restrict__ reads,
                                                       const uint8_t* __restrict__ flags, uint64_t n,
                                                       int k, uint64_t kmask, int max_mm, MatchOut o) {
    __shared__ uint64_t tile[BRUTE_TILE];
    const uint64_t per_block = (uint64_t)BLOCK * BRUTE_RPT;
    for (uint64_t blk0 = (uint64_t)blockIdx.x * per_block; blk0 < n; blk0 += (uint64_t)gridDim.x * per_block) {
        uint32_t rlo[BRUTE_RPT], rhi[BRUTE_RPT];
        bool live[BRUTE_RPT];
        Best b[BRUTE_RPT];
#pragma unroll
        for (int q = 0; q < BRUTE_RPT; ++q) {
            uint64_t i = blk0 + (uint64_t)q * BLOCK + threadIdx.x;
            uint64_t r = (i < n) ? reads[i] : 0;
            live[q] = (i < n) && !read_skipped(r, flags, i, k);
            r &= kmask;
            rlo[q] = (uint32_t)r; rhi[q] = (uint32_t)(r >> 32);
            // a dead lane keeps cur = 0 and compares against an impossible word
            b[q] = Best{BC_IDX_MISS, BC_DIST_NONE, 0u, live[q] ? (uint32_t)max_mm : 0u};
            if (!live[q]) { rlo[q] = 0xFFFFFFFFu; rhi[q] = 0xFFFFFFFFu; }
        }
        for (uint32_t t0 = 0; t0 < G; t0 += BRUTE_TILE) {
            uint32_t tn = min((uint32_t)BRUTE_TILE, G - t0);
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < tn; j += BLOCK) tile[j] = guides[t0 + j];
            __syncthreads();
            for (uint32_t j = 0; j < tn; ++j) {
                uint64_t g = tile[j];
                uint32_t glo = (uint32_t)g, ghi = (uint32_t)(g >> 32);
#pragma unroll
                for (int q = 0; q < BRUTE_RPT; ++q) {
                    uint32_t d = __popc((rlo[q] ^ glo) | (rhi[q] ^ ghi));
                    if (d <= b[q].cur && live[q]) consider(b[q], d, t0 + j);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BRUTE_RPT; ++q) {
            uint64_t i = blk0 + (uint64_t)q * BLOCK + threadIdx.x;
            if (i >= n) continue;
            if (live[q]) emit(o, i, b[q].idx, b[q].dist, b[q].nb);
            else emit(o, i, BC_IDX_SKIP, BC_DIST_NONE, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// Library preprocessing: distance from every guide to its nearest other guide
// (LDS-tiled all pairs, the brute-force kernel's inner loop).  A guide whose
// nearest neighbour is farther than 2 * max_mm is "isolated": once a read is
// found within max_mm of it, no other guide can be within max_mm of that read
// (triangle inequality), so the remaining seed tables need not be visited.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_nn_dist(const uint64_t* __restrict__ guides, uint32_t G, uint64_t kmask,
                                                   uint8_t* __restrict__ out) {
    __shared__ uint64_t tile[BRUTE_TILE];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t me = (i < G) ? (guides[i] & kmask) : 0;
    const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
    uint32_t best = 255u;
    for (uint32_t t0 = 0; t0 < G; t0 += BRUTE_TILE) {
        const uint32_t tn = min((uint32_t)BRUTE_TILE, G - t0);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < tn; j += BLOCK) tile[j] = guides[t0 + j] & kmask;
        __syncthreads();
        for (uint32_t j = 0; j < tn; ++j) {
            const uint64_t g = tile[j];
            const uint32_t d = __popc((lo ^ (uint32_t)g) | (hi ^ (uint32_t)(g >> 32)));
            best = (t0 + j != i) ? min(best, d) : best;
        }
    }
    if (i < G) out[i] = (uint8_t)best;
}

// neighbour lists of the guides listed in cand[] (those with another guide within NBR_RADIUS): one thread per
// candidate walks the whole library (the same address in every lane: broadcast loads)
__global__ __launch_bounds__(64) void k_nn_list(const uint64_t* __restrict__ guides, uint32_t G, uint64_t kmask,
                                                const uint32_t* __restrict__ cand, uint32_t ncand,
                                                uint4* __restrict__ nbr, uint32_t* __restrict__ count) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= ncand) return;
    const uint32_t i = cand[c];
    const uint64_t me = guides[i] & kmask;
    const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < G; ++j) {
        const uint64_t g = guides[j] & kmask;
        const uint32_t d = __popc((lo ^ (uint32_t)g) | (hi ^ (uint32_t)(g >> 32)));
        if (d <= (uint32_t)NBR_RADIUS && j != i) {
            if (cnt < (uint32_t)NBR_CAP) nbr[(size_t)c * NBR_CAP + cnt] = make_uint4((uint32_t)g, (uint32_t)(g >> 32), j, 1u);
            ++cnt;
        }
    }
    count[c] = cnt;
}

// ---------------------------------------------------------------------------
// synthetic reads (bench / tests)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_synth_reads(const uint64_t* __restrict__ guides, uint32_t G,
                                                       int k, uint64_t seed, uint64_t first, uint64_t n,
                                                       int mixture, uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += stride) {
        uint64_t i = first + t;
        uint64_t g = guides[synth_guide_index(seed, i, G)];
        out[t] = synth_read(seed, i, g, k, mixture);
    }
}

int grid_for(uint64_t work_items) {
    uint64_t b = (work_items + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    return (int)b;
}

template <typename T>
int dev_upload(bc_lib* lib, const std::vector<T>& h, T** d) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
    BC_HIP(hipMalloc(&p, bytes));
    lib->owned.push_back(p);
    if (!h.empty()) BC_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = (T*)p;
    return BC_OK;
}

int floor_log4(uint32_t g) {
    int l = 0;
    while (g >= 4) { g >>= 2; ++l; }
    return l;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// slab form of table s of the index for max_mismatch = d (see bc_device.h)
int build_slab_table(bc_lib* lib, int d, int s, const std::vector<uint32_t>& off,
                     const std::vector<uint64_t>& keys, const std::vector<uint32_t>& gidx) {
    const SeedParams& cs = lib->seed[d];
    SlabParams& sl = lib->slab[d];
    if (s == 0) { sl.nseg = cs.nseg; lib->slab_ok[d] = true; }
    const int start = cs.start[s], slen = cs.slen[s], half = lib->k - slen;
    // rest in one dword (two planes of `half` bits: narrow entries) or in two (wide entries, k up to 32); index: 22 bits + 2 flags.
    // All tables of an index use one format: the widest rest decides.
    if (lib->G >= (1u << SLAB_GIDX_BITS)) { lib->slab_ok[d] = false; return BC_OK; }
    if (s == 0) {
        bool wide = false;
        for (int t = 0; t < cs.nseg; ++t) wide = wide || 2 * (lib->k - cs.slen[t]) > 32;
        lib->slab_wide[d] = wide;
    }
    const bool wide = lib->slab_wide[d];
    if (!lib->slab_ok[d]) return BC_OK;
    const uint32_t nb = 1u << (2 * slen);
    // slab size: 8 entries (7 + link: 64 bytes narrow, 128 wide) while the mean bucket fits, else 16 and
    // chaining.  Measured on MI355X (G=1e5, k=20, d<=2): sizes (8,8,16) 3.04 ms per 1e8 reads vs
    // (16,16,32) 3.13 ms — short chained slabs beat long ones (smaller tables, fewer bytes per visit).
    const double mu = (double)lib->G / nb;
    int log2S = (mu <= 7.0) ? 3 : 4;
    // 8-byte entries of 16+ base k-mers: 128-byte slabs from 7 guides per bucket on, in every table (k_match_wq takes two
    // pieces per lane then).  Measured, 1e8 reads, d <= 2, G = 1.2e5 / 1.4e5 / 1.6e5 / 2e5: 64-byte slabs with chains
    // 1.38 / 1.56 / 1.81 / 2.35 ms, 128-byte slabs 1.39 / 1.43 / 1.49 / 1.65 ms (at 1e5, 6.1 per bucket: 1.14 / 1.35).
    // Shorter k-mers and 16-byte entries keep 8-entry slabs for tables 0 and 1 up to 16 guides per bucket: the quad
    // kernels (12-mers) and the wide wave kernel are built for that geometry.
    if ((cs.nseg == 2 || cs.nseg == 3) && s < 2 && mu <= 16.0 && (wide || lib->k < 16)) log2S = 3;
    log2S = std::max(log2S, env_int("BC_SLAB_MIN_LOG2S", 3));
    log2S = std::min(std::min(SLAB_MAX_LOG2S, env_int("BC_SLAB_MAX_LOG2S", SLAB_MAX_LOG2S)), log2S);
    if (const char* spec = getenv("BC_SLAB_LOG2S")) {  // tuning aid: explicit per-table sizes "a,b,c"
        int vals[MAX_SEG] = {0, 0, 0, 0};
        int nv = sscanf(spec, "%d,%d,%d,%d", &vals[0], &vals[1], &vals[2], &vals[3]);
        if (s < nv && vals[s] >= 3 && vals[s] <= SLAB_MAX_LOG2S) log2S = vals[s];
    }
    if (wide) log2S = std::min(log2S, 4);  // 16-byte entries: at most 256-byte slabs (four pieces per lane)
    const uint32_t S = 1u << log2S;
    const uint32_t W = wide ? 2u : 1u;  // uint2 words per entry
    // entries as uint2 words: narrow {rest, giw}; wide {rest lo, giw}{rest hi, 0}
    std::vector<uint2> slabs(((size_t)nb << log2S) * W, make_uint2(0u, SLAB_EMPTY));
    if (wide)
        for (size_t e = 0; e < ((size_t)nb << log2S); ++e) slabs[e * 2 + 1] = make_uint2(0u, 0u);
    const uint32_t below = (1u << start) - 1u;
    const int sh = start + slen;
    auto put_entry = [&](size_t slot, uint64_t g, uint32_t giw, bool link, uint32_t next) {
        const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
        const uint32_t lo_r = (lo & below) | ((sh >= 32 ? 0u : (lo >> sh)) << start);
        const uint32_t hi_r = (hi & below) | ((sh >= 32 ? 0u : (hi >> sh)) << start);
        if (link) {
            slabs[slot * W] = make_uint2(next, SLAB_LINK);
            if (wide) slabs[slot * W + 1] = make_uint2(0u, 0u);
        } else if (wide) {
            slabs[slot * 2] = make_uint2(lo_r, giw);
            slabs[slot * 2 + 1] = make_uint2(hi_r, 0u);
        } else {
            slabs[slot] = make_uint2(lo_r | (hi_r << 16), giw);  // narrow: half <= 16 bases per plane
        }
    };
    auto append_slab = [&]() {
        const size_t idx = slabs.size() / W >> log2S;
        slabs.resize(slabs.size() + (size_t)S * W, make_uint2(0u, SLAB_EMPTY));
        if (wide)
            for (uint32_t e = 0; e < S; ++e) slabs[(idx << log2S) * 2 + (size_t)e * 2 + 1] = make_uint2(0u, 0u);
        return idx;
    };
    for (uint32_t b = 0; b < nb; ++b) {
        size_t cur = b;
        uint32_t pos = 0;
        for (uint32_t j = off[b]; j < off[b + 1]; ++j) {
            if (pos == S - 1) {  // chain an overflow slab
                const size_t nxt = append_slab();
                put_entry((cur << log2S) + S - 1, 0, 0, true, (uint32_t)nxt);
                cur = nxt;
                pos = 0;
            }
            // the isolation class rides in the index word (see bc_device.h): how many of the radii 2, 4 .. 2d the
            // guide's nearest neighbour lies beyond
            uint32_t fl = 0;
            if (!lib->h_nn.empty()) {
                const int nn = lib->h_nn[gidx[j]];
                uint32_t c = 0;
                for (int e = 1; e <= d; ++e) c += nn > 2 * e ? 1u : 0u;
                fl = c << SLAB_CLASS_SHIFT;
            }
            put_entry((cur << log2S) + pos++, keys[j], gidx[j] | fl, false, 0);
        }
    }
    const uint32_t empty_slab = (uint32_t)append_slab();
    if ((slabs.size() * sizeof(uint2)) >> 32) { lib->slab_ok[d] = false; return BC_OK; }  // 32-bit byte offsets
    uint2* d_slabs;
    int rc = dev_upload(lib, slabs, &d_slabs);
    if (rc != BC_OK) return rc;
    SlabTable& T = sl.t[s];
    T.empty = empty_slab;
    T.wide = wide ? 1u : 0u;
    T.slabs = d_slabs; T.start = (uint32_t)start; T.slen = (uint32_t)slen; T.half = wide ? (uint32_t)half : 16u;
    T.log2S = (uint32_t)log2S;
    for (int t = 0; t < MAX_SEG; ++t)  // earlier seeds lie below this one: same bit positions in `rest`
        T.dmask[t] = (t < s) ? (((1u << cs.slen[t]) - 1u) << cs.start[t]) : 0u;
    // tables 0 and 1 once more in the eight-entry form for k_match_wq (EightTables, bc_device.h): narrow entries,
    // 64-byte slabs, three spare bits per plane of the rest (the empty slot's sentinel), a budget of at most two
    if (s == 0) sl.e8 = EightTables{};
    if (s < 2 && !wide && log2S == 3 && (cs.nseg == 2 || cs.nseg == 3) && lib->k - slen <= 13 && d <= 2 && lib->knobs.e8 &&
        !lib->h_nn.empty()) {
        std::vector<uint2> e8((size_t)nb * 8, make_uint2(E8_EMPTY_REST, E8_EMPTY_GIW8));
        uint64_t n_ovf = 0;
        auto set_link = [&](size_t slab, uint32_t link16) {
            e8[slab * 8 + 6].y = (e8[slab * 8 + 6].y & 0xFFFFFF00u) | (link16 & 0xFFu);
            e8[slab * 8 + 7].y = (e8[slab * 8 + 7].y & 0xFFFFFF00u) | (link16 >> 8);
        };
        for (uint32_t b = 0; b < nb; ++b) {
            size_t cur = b;
            uint32_t pos = 0;
            for (uint32_t j = off[b]; j < off[b + 1]; ++j) {
                if (pos == 8) {  // chain an overflow slab
                    const size_t nxt = e8.size() / 8;
                    e8.resize(e8.size() + 8, make_uint2(E8_EMPTY_REST, E8_EMPTY_GIW8));
                    ++n_ovf;
                    set_link(cur, (uint32_t)(nxt - nb + 1));
                    cur = nxt;
                    pos = 0;
                }
                const int nn = lib->h_nn[gidx[j]];
                uint32_t c = 0;
                for (int e = 1; e <= d; ++e) c += nn > 2 * e ? 1u : 0u;
                const uint32_t lo = (uint32_t)keys[j], hi = (uint32_t)(keys[j] >> 32);
                const uint32_t lo_r = (lo & below) | ((sh >= 32 ? 0u : (lo >> sh)) << start);
                const uint32_t hi_r = (hi & below) | ((sh >= 32 ? 0u : (hi >> sh)) << start);
                uint2& slot = e8[cur * 8 + pos++];
                slot.x = lo_r | (hi_r << 16);
                slot.y = ((gidx[j] | (c << SLAB_CLASS_SHIFT)) << 8) | (slot.y & 0xFFu);
            }
        }
        const size_t empty8 = e8.size() / 8;
        e8.resize(e8.size() + 8, make_uint2(E8_EMPTY_REST, E8_EMPTY_GIW8));
        if (n_ovf + 1 < 65535 && empty8 < (1u << 22) && ((e8.size() * sizeof(uint2)) >> 32) == 0) {
            uint2* d_e8;
            if ((rc = dev_upload(lib, e8, &d_e8)) != BC_OK) return rc;
            sl.e8.base[s] = reinterpret_cast<const char*>(d_e8);
            sl.e8.nb[s] = nb;
            sl.e8.empty[s] = (uint32_t)empty8;
            if (s == 1 && sl.e8.base[0]) sl.e8.ok = 1u;
        }
    }
    if (s == cs.nseg - 1) {  // descriptor block for the kernel (read through scalar loads)
        sl.nbr_slot = lib->h_nn.empty() ? nullptr : lib->d_nbr_slot;
        sl.nbr = lib->h_nn.empty() ? nullptr : lib->d_nbr;
        sl.flt = nullptr;
        if (cs.nseg == 3 && env_int("BC_T2_FILTER", 1)) {  // table-2 filter (see SlabParams)
            const int k = lib->k;
            int seg_len[3], seg_pos[3], at = 0;
            for (int t = 0; t < 3; ++t) { seg_len[t] = k / 3 + (t < k % 3 ? 1 : 0); seg_pos[t] = at; at += seg_len[t]; }
            for (int t = 0; t < 2; ++t) {
                const int la = (seg_len[t] + 1) / 2;
                sl.f_pos[2 * t] = (uint32_t)seg_pos[t]; sl.f_len[2 * t] = (uint32_t)la;
                sl.f_pos[2 * t + 1] = (uint32_t)(seg_pos[t] + la); sl.f_len[2 * t + 1] = (uint32_t)(seg_len[t] - la);
            }
            bool usable = true;
            for (int h = 0; h < 4; ++h) usable = usable && sl.f_len[h] >= 1 && sl.f_len[h] <= 16;
            if (usable) {
                std::vector<uint32_t> flt((size_t)nb * 16, 0u);
                const uint32_t m2 = (1u << slen) - 1u;
                for (uint32_t g = 0; g < lib->G; ++g) {
                    const uint32_t lo = (uint32_t)lib->h_guides[g], hi = (uint32_t)(lib->h_guides[g] >> 32);
                    const uint32_t bucket = ((lo >> start) & m2) | (((hi >> start) & m2) << slen);
                    for (int h = 0; h < 4; ++h) {
                        const uint32_t mh = (1u << sl.f_len[h]) - 1u;
                        const uint32_t v = ((lo >> sl.f_pos[h]) & mh) | (((hi >> sl.f_pos[h]) & mh) << sl.f_len[h]);
                        const uint32_t bit = flt_hash7(v);
                        flt[(size_t)bucket * 16 + (size_t)h * 4 + (bit >> 5)] |= 1u << (bit & 31u);
                    }
                }
                uint32_t* d_flt;
                if ((rc = dev_upload(lib, flt, &d_flt)) != BC_OK) return rc;
                sl.flt = d_flt;
            }
        }
        std::vector<SlabParams> one(1, sl);
        SlabParams* d_sp;
        if ((rc = dev_upload(lib, one, &d_sp)) != BC_OK) return rc;
        lib->d_slab[d] = d_sp;
    }
    return BC_OK;
}

int build_seed_index(bc_lib* lib, int d) {
    SeedParams& sp = lib->seed[d];
    const int k = lib->k;
    sp.nseg = std::min(d + 1, k);
    // seed length: about 4..16 guides per bucket (G=1e5 -> 7 nt).  Longer seeds mean bigger tables;
    // for d=1 (two 10-nt segments) 7-nt seeds measured 2.10 ms per 1e8 reads against 2.40 ms with 8-nt seeds.
    int want = std::max(4, std::min(12, floor_log4(std::max(lib->G / 4, 1u)) + env_int("BC_SEED_WANT_BIAS", 0)));  // (bias: experiments, profiles/seed_want_ab.sh)
    int pos = 0;
    for (int s = 0; s < sp.nseg; ++s) {
        int seg = k / sp.nseg + (s < k % sp.nseg ? 1 : 0);
        sp.start[s] = pos;
        sp.slen[s] = std::min(seg, want);
        pos += seg;
    }
    for (int s = 0; s < sp.nseg; ++s) {
        const uint32_t nb = 1u << (2 * sp.slen[s]);
        const uint32_t m = (1u << sp.slen[s]) - 1u;
        std::vector<uint32_t> off(nb + 1, 0);
        auto sv = [&](uint64_t g) {
            uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
            return ((lo >> sp.start[s]) & m) | (((hi >> sp.start[s]) & m) << sp.slen[s]);
        };
        for (uint32_t i = 0; i < lib->G; ++i) off[sv(lib->h_guides[i]) + 1]++;
        for (uint32_t b = 0; b < nb; ++b) off[b + 1] += off[b];
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        std::vector<uint64_t> keys(lib->G);
        std::vector<uint32_t> gidx(lib->G);
        for (uint32_t i = 0; i < lib->G; ++i) {  // ascending i -> ascending index inside a bucket
            uint32_t at = cur[sv(lib->h_guides[i])]++;
            keys[at] = lib->h_guides[i];
            gidx[at] = i;
        }
        uint32_t* d_off; uint64_t* d_keys; uint32_t* d_gidx;
        int rc;
        if ((rc = dev_upload(lib, off, &d_off)) != BC_OK) return rc;
        if ((rc = dev_upload(lib, keys, &d_keys)) != BC_OK) return rc;
        if ((rc = dev_upload(lib, gidx, &d_gidx)) != BC_OK) return rc;
        sp.off[s] = d_off; sp.keys[s] = d_keys; sp.gidx[s] = d_gidx;
        if ((rc = build_slab_table(lib, d, s, off, keys, gidx)) != BC_OK) return rc;
    }
    return BC_OK;
}

// nearest-neighbour distances for the isolation flags (all pairs: only for libraries up to 4M guides)
int compute_nn(bc_lib* lib) {
    lib->h_nn.clear();
    if (lib->G < 2 || lib->G > 4000000u || !env_int("BC_ISOLATION", 1)) return BC_OK;
    uint8_t* d_nn = nullptr;
    BC_HIP(hipMalloc((void**)&d_nn, lib->G));
    hipLaunchKernelGGL(k_nn_dist, dim3((lib->G + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, lib->d_guides, lib->G, lib->kmask, d_nn);
    lib->h_nn.resize(lib->G);
    hipError_t e = hipMemcpy(lib->h_nn.data(), d_nn, lib->G, hipMemcpyDeviceToHost);
    (void)hipFree(d_nn);
    if (e != hipSuccess) { lib->h_nn.clear(); set_error("k_nn_dist: %s", hipGetErrorString(e)); return BC_EHIP; }
    // neighbour lists for the guides that are not isolated at NBR_RADIUS
    std::vector<uint32_t> cand;
    for (uint32_t i = 0; i < lib->G; ++i)
        if (lib->h_nn[i] <= NBR_RADIUS) cand.push_back(i);
    std::vector<uint32_t> slot(lib->G, NBR_NONE);
    std::vector<uint4> lists(cand.size() * (size_t)NBR_CAP, make_uint4(0u, 0u, 0u, 0u));
    if (!cand.empty()) {
        uint32_t* d_cand = nullptr; uint32_t* d_cnt = nullptr; uint4* d_l = nullptr;
        const uint32_t nc = (uint32_t)cand.size();
        hipError_t e2 = hipMalloc((void**)&d_cand, nc * sizeof(uint32_t));
        if (e2 == hipSuccess) e2 = hipMalloc((void**)&d_cnt, nc * sizeof(uint32_t));
        if (e2 == hipSuccess) e2 = hipMalloc((void**)&d_l, lists.size() * sizeof(uint4));
        if (e2 == hipSuccess) e2 = hipMemcpy(d_cand, cand.data(), nc * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e2 == hipSuccess) e2 = hipMemset(d_l, 0, lists.size() * sizeof(uint4));
        std::vector<uint32_t> cnt(nc, 0);
        if (e2 == hipSuccess) {
            hipLaunchKernelGGL(k_nn_list, dim3((nc + 63) / 64), dim3(64), 0, 0, lib->d_guides, lib->G, lib->kmask, d_cand, nc, d_l, d_cnt);
            e2 = hipMemcpy(cnt.data(), d_cnt, nc * sizeof(uint32_t), hipMemcpyDeviceToHost);
        }
        if (e2 == hipSuccess) e2 = hipMemcpy(lists.data(), d_l, lists.size() * sizeof(uint4), hipMemcpyDeviceToHost);
        if (d_cand) (void)hipFree(d_cand);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_l) (void)hipFree(d_l);
        if (e2 != hipSuccess) { lib->h_nn.clear(); set_error("k_nn_list: %s", hipGetErrorString(e2)); return BC_EHIP; }
        for (uint32_t c = 0; c < nc; ++c) slot[cand[c]] = (cnt[c] > (uint32_t)NBR_CAP) ? NBR_OVER : c;
    }
    int rc = dev_upload(lib, slot, &lib->d_nbr_slot);
    if (rc == BC_OK) rc = dev_upload(lib, lists, &lib->d_nbr);
    return rc;
}

int build_seed_index(bc_lib* lib, int d);

int build_hash(bc_lib* lib) {
    const int k = lib->k;
    // Slots: a power of two.  A sparse table (five slots per guide) saves probes while it is small anyway; from 2^17 slots
    // (1 MB) on it is its footprint in the L2 that costs, and two slots per guide win: 0.655 -> 0.630 ms per 1e8 reads
    // at 1e5 guides, 1.29 -> 0.83 at 2.5e5, 1.84 -> 1.42 at 1e6 (profiles/r03_hash_slots.txt).  BC_HASH_SLOTS_PCT
    // (tuning aid, tests) sets slots >= pct % of G instead.
    auto log2_at_least = [&](uint64_t want) { int l = 4; while ((1ull << l) < want) ++l; return l; };
    const int pct = env_int("BC_HASH_SLOTS_PCT", 0);
    int ls = pct > 0 ? log2_at_least((uint64_t)lib->G * std::max(110, pct) / 100 + 1)
                     : std::max(log2_at_least((uint64_t)lib->G * 2 + 1), std::min(log2_at_least((uint64_t)lib->G * 5 + 1), 17));
    lib->log_slots = ls;
    const uint32_t nslots = 1u << ls, smask = nslots - 1;
    int free_bits = 2 * (32 - k);
    lib->wide = !(free_bits >= 33 || ((uint64_t)lib->G + 1) < (1ull << free_bits));
    if (k == 32) lib->wide = true;
    if (lib->wide) {
        std::vector<uint32_t> slots(nslots, 0);
        for (uint32_t i = 0; i < lib->G; ++i) {
            uint32_t h = hash32(lib->h_guides[i]) >> (32 - ls);
            while (slots[h]) {
                if (lib->h_guides[slots[h] - 1] == lib->h_guides[i]) {
                    set_error("bc_lib_create: duplicate guide at index %u (first at %u)", i, slots[h] - 1);
                    return BC_EINVAL;
                }
                h = (h + 1) & smask;
            }
            slots[h] = i + 1;
        }
        return dev_upload(lib, slots, &lib->d_slots32);
    }
    // narrow: nslots buckets of two slots each (total slots = 2 x the count chosen above / 2: keep the
    // memory of the single-slot layout by halving the bucket count)
    lib->log_slots = ls = std::max(3, ls - 1);
    const uint32_t nb = 1u << ls, bmask = nb - 1;
    std::vector<uint64_t> slots((size_t)nb * 2, 0);
    const uint32_t fm = (k >= 32) ? 0u : ((1u << (32 - k)) - 1u);
    for (uint32_t i = 0; i < lib->G; ++i) {
        uint64_t key = lib->h_guides[i];
        uint32_t h = hash32(key) >> (32 - ls);
        uint32_t v = i + 1;
        uint64_t lo = (uint64_t)(v & fm) << k;
        uint64_t hi = (uint64_t)((v >> (32 - k)) & fm) << (32 + k);
        const uint64_t word = key | lo | hi;
        for (;;) {
            uint64_t& x = slots[(size_t)h * 2];
            uint64_t& y = slots[(size_t)h * 2 + 1];
            if ((x && (x & lib->kmask) == key) || (y && (y & lib->kmask) == key)) {
                set_error("bc_lib_create: duplicate guide at index %u", i);
                return BC_EINVAL;
            }
            if (!x) { x = word; break; }
            if (!y) { y = word; break; }
            h = (h + 1) & bmask;
        }
    }
    return dev_upload(lib, slots, &lib->d_slots64);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// 16-byte reads / results and 4-byte distance / tie words: what the vectorised kernels need
bool vector_io_ok(const uint64_t* d_reads, const MatchOut& o, uint64_t n) {
    return aligned16(d_reads) && aligned16(o.best_idx) && n >= 4 && (!o.best_dist || (((uintptr_t)o.best_dist) & 3u) == 0) &&
           (!o.n_best || (((uintptr_t)o.n_best) & 3u) == 0);
}

// does the wave matcher (k_match_wq) take a launch with this budget?  nq01 / nq2: 16-byte pieces per lane of a
// table-0/1 and of a table-2 slab
bool wave_kernel_serves(const bc_lib* lib, int max_mm, int algo, bool vec_ok, int* nq01_out, int* nq2_out) {
    if (max_mm < 1 || max_mm >= MAX_SEG) return false;
    const SlabParams& hp = lib->slab[max_mm];
    const int nseg = hp.nseg;
    const bool wide = lib->slab_wide[max_mm];
    // table-0/1 slabs of 64 bytes (8 narrow entries) or 128 bytes (16 narrow entries: buckets above 16 guides; or
    // 8 wide entries): one or two 16-byte pieces per lane
    const int nq01 = (int)(((wide ? 16u : 8u) << hp.t[0].log2S) / 64u);
    bool wave_ok = algo == BC_ALGO_SEED && lib->slab_ok[max_mm] && vec_ok && !lib->knobs.single_slab && hp.t[0].start == 0u &&
                   lib->knobs.lane && (nseg == 2 || nseg == 3) && hp.t[0].log2S == hp.t[1].log2S &&
                   (nq01 == 1 || nq01 == 2) && !lib->h_nn.empty();
    for (int t = 0; t < nseg && wave_ok; ++t) wave_ok = hp.t[t].empty < (1u << 22);  // slab numbers travel in 22-bit item fields
    // narrow 128-byte slabs: measured against the quad kernels, 1e8 reads, d <= 2: 5e5 20-mers (30 per bucket) 14.7 vs
    // 13.1 G reads/s, 1e6 (61 per bucket) 6.1 vs 7.1, 2e4 12-mers (rest of 8 bases: ties everywhere) 9.6 vs 11.0
    if (wave_ok && !wide && nq01 == 2 && (lib->k < 16 || (double)lib->G / (double)(1u << (2 * hp.t[0].slen)) > 40.0)) wave_ok = false;
    const int nq2 = nseg == 3 ? (int)(((wide ? 16u : 8u) << hp.t[2].log2S) / 64u) : nq01;  // pieces per lane, table 2
    if (wave_ok && nseg == 3 && (nq2 > 4 || (nq01 == 2 && nq2 < 2))) wave_ok = false;
    *nq01_out = nq01; *nq2_out = nq2;
    return wave_ok;
}

// does the kernel a launch will run write tally records (PartOut)?  k_match_hash and k_match_wq do
bool kernel_writes_records(const bc_lib* lib, const uint64_t* d_reads, const MatchOut& o, uint64_t n, int max_mm, int algo) {
    const bool vec_ok = vector_io_ok(d_reads, o, n);
    if (algo == BC_ALGO_HASH) return vec_ok && max_mm == 0;
    int a, b;
    return wave_kernel_serves(lib, max_mm, algo, vec_ok, &a, &b);
}
int launch_match_kernel(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                        int max_mm, int algo, const MatchOut& o, hipStream_t st, bool* part_done, uint64_t* tail_begin);

// dynamic LDS above 64 KiB needs the attribute once per device and kernel (a second bc_lib on another device, or a
// second thread, must not launch without it)
template <typename K>
int ensure_big_lds(K kernel, int device, size_t bytes) {
    static std::mutex m;
    static bool done[64] = {};
    std::lock_guard<std::mutex> g(m);
    if (device >= 0 && device < 64 && done[device]) return BC_OK;
    BC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (device >= 0 && device < 64) done[device] = true;
    return BC_OK;
}

constexpr size_t HIST_LDS_BYTES = (size_t)HIST_R * 2 * sizeof(uint32_t);

int launch_hist(bc_lib* lib, const MatchOut& o, uint64_t n, hipStream_t st) {
    int rc = ensure_big_lds(k_hist, lib->device, HIST_LDS_BYTES);
    if (rc != BC_OK) return rc;
    const int passes = (int)((lib->G + HIST_R - 1) / HIST_R);
    const int slices = std::max(1, 256 / passes);
    const uint64_t CH = 0xFFFFFFF0ull;  // first-seen is tracked as a 32-bit offset inside a sweep
    for (uint64_t a = 0; a < n; a += CH) {
        uint64_t m = std::min(CH, n - a);
        hipLaunchKernelGGL(k_hist, dim3(passes * slices), dim3(HIST_BLOCK), HIST_LDS_BYTES, st, o.best_idx + a, m, lib->G,
                           slices, o.counts, o.first_seen, o.base + a);
    }
    BC_HIP(hipGetLastError());
    return BC_OK;
}

// ---- d = 3 cascade ---------------------------------------------------------------------------------------------
// A guide within two substitutions, once found, is the best hit whatever the budget; only reads the d <= 2 pass left
// without a hit need the 3-mismatch search, whose four 5-base seeds put ~G/1024 guides in every bucket (9.4 G reads/s
// over all reads at 1e5 guides against 75 for d <= 2).  The second stage (k_cascade_stage2, bc_match_pairs.hip) finds
// those reads in best_idx itself and writes its results in place: launch_cascade3 below.

// ---- tally records: geometry and per-stream workspace -------------------------------------------------------
// ranges: as few as the LDS allows (HIST_R guides each), equal sizes; range(g) = umulhi(g, magic) must be exact
bool part_geometry_compute(uint32_t G, PartOut& P) {
    uint32_t nr = (G + HIST_R - 1) / HIST_R;
    // one or two ranges: sweeping best_idx costs no more
    if (nr < 3 || nr > (uint32_t)PART_MAX_NR) return false;
    // a line must hold well more than the hits it expects (256 results x 90 % over nr ranges): with three or four
    // ranges and 64-record lines a tile overflows as a rule (77 / 58 hits per range) and falls back to best_idx
    // (measured at 6e4 / 8e4 guides: 7.2 / 4.7 ms per pass instead of 1.2): at least five ranges (<= 51 expected in 64);
    // 17..32 ranges take 32-record lines (<= 13.6 expected), 33..64 take 16-record lines (<= 7 expected)
    nr = std::max(nr, 5u);
    uint32_t R = (G + nr - 1) / nr;
    uint32_t magic = (uint32_t)((0x100000000ull + R - 1) / R);
    bool exact = nr <= 8u;  // (beyond ~1.6e5 guides the 32-bit reciprocal of an arbitrary R stops being exact)
    for (uint32_t g = 0; g < G && exact; ++g) exact = (uint32_t)(((uint64_t)g * magic) >> 32) == g / R;
    if (!exact) {  // ranges of 2^14 guides: range = g >> 14 = umulhi(g, 2^18), exact for every g
        R = 16384u;
        magic = 1u << 18;
        nr = (G + R - 1) / R;
        if (nr > (uint32_t)PART_MAX_NR) return false;
    }
    P.nr = nr; P.R = R; P.magic = magic;
    // records per line: a line must hold well more than the 230 / nr hits it expects (Poisson tails: 64 for up to 14
    // ranges, 32 — expecting at most 15.4 — up to 32, 16 — at most 7 — up to 64)
    P.cap_log2 = nr <= 14u ? 6u : nr <= 32u ? 5u : 4u;
    return (nr << P.cap_log2) <= (uint32_t)PART_SLOTS;
}

bool part_geometry(const bc_lib* lib, PartOut& P) {  // the library's cached geometry (bc_lib_create)
    if (!lib->part_ok) return false;
    P.nr = lib->part_nr; P.R = lib->part_R; P.magic = lib->part_magic; P.cap_log2 = lib->part_cap_log2;
    return true;
}

// launches on one stream are ordered, so one workspace per stream is enough; it only ever grows
static int stream_workspace(bc_lib* lib, hipStream_t st, TallyWs** out) {  // caller holds ws_mutex
    TallyWs* w = nullptr;
    for (auto& e : lib->ws) if (e.stream == (void*)st) w = &e;
    if (!w) {
        if (lib->ws.size() >= 1024) {  // (a caller that rotates streams releases them: bc_lib_release_stream)
            set_error("bc_match: more than 1024 streams hold a workspace on one library (bc_lib_release_stream frees one)");
            return BC_EINVAL;
        }
        lib->ws.emplace_back();
        w = &lib->ws.back();
        w->stream = (void*)st;
        BC_HIP(hipMalloc((void**)&w->first_tile, (size_t)lib->G * 4));
        BC_HIP(hipMalloc((void**)&w->ovf_count, 16));
        BC_HIP(hipMemset(w->first_tile, 0xFF, (size_t)lib->G * 4));
        BC_HIP(hipMemset(w->ovf_count, 0, 16));
    }
    *out = w;
    return BC_OK;
}

int tally_workspace(bc_lib* lib, hipStream_t st, uint64_t n_tiles, uint32_t nr, uint32_t cap_log2, TallyWs** out) {
    std::lock_guard<std::mutex> g(lib->ws_mutex);
    TallyWs* w = nullptr;
    int rc = stream_workspace(lib, st, &w);
    if (rc != BC_OK) return rc;
    if (n_tiles > w->tiles_cap) {
        BC_HIP(hipStreamSynchronize(st));  // nothing of this stream may still use the old buffers
        if (w->rec) (void)hipFree(w->rec);
        if (w->ovf_list) (void)hipFree(w->ovf_list);
        w->rec = nullptr; w->ovf_list = nullptr; w->tiles_cap = 0;
        BC_HIP(hipMalloc((void**)&w->rec, (((size_t)n_tiles * nr) << cap_log2) * sizeof(uint16_t)));
        BC_HIP(hipMalloc((void**)&w->ovf_list, (size_t)n_tiles * 4));
        BC_HIP(hipMemset(w->ovf_list, 0, (size_t)n_tiles * 4));  // (zero between passes: part_emit sets, k_first_refine clears)
        w->tiles_cap = n_tiles;
    }
    *out = w;
    return BC_OK;
}

// the miss list of a cascade's second stage: room for every read of the launch (all of them may miss)
int cascade_workspace_impl(bc_lib* lib, hipStream_t st, uint64_t n, TallyWs** out) {
    std::lock_guard<std::mutex> g(lib->ws_mutex);
    TallyWs* w = nullptr;
    int rc = stream_workspace(lib, st, &w);
    if (rc != BC_OK) return rc;
    const uint64_t tiles = (n + 255) >> 8;
    if (tiles > w->casc_tiles) {
        BC_HIP(hipStreamSynchronize(st));  // nothing of this stream may still use the old buffers
        for (void* p : {(void*)w->casc_read, (void*)w->casc_pos, (void*)w->casc_state, (void*)w->casc_cnt, (void*)w->casc_off,
                        (void*)w->casc_dread, (void*)w->casc_dpos, (void*)w->casc_count, w->casc_scan_tmp}) if (p) (void)hipFree(p);
        w->casc_read = nullptr; w->casc_pos = nullptr; w->casc_state = nullptr; w->casc_cnt = nullptr; w->casc_tiles = 0;
        w->casc_off = nullptr; w->casc_dread = nullptr; w->casc_dpos = nullptr; w->casc_count = nullptr; w->casc_scan_tmp = nullptr;
        BC_HIP(hipMalloc((void**)&w->casc_read, tiles * 256 * 8));
        BC_HIP(hipMalloc((void**)&w->casc_pos, tiles * 256 * 4));
        BC_HIP(hipMalloc((void**)&w->casc_state, tiles * 256 * 8));
        BC_HIP(hipMalloc((void**)&w->casc_cnt, tiles * 4));
        BC_HIP(hipMalloc((void**)&w->casc_off, tiles * 4));
        BC_HIP(hipMalloc((void**)&w->casc_dread, tiles * 256 * 8));
        BC_HIP(hipMalloc((void**)&w->casc_dpos, tiles * 256 * 4));
        BC_HIP(hipMalloc((void**)&w->casc_count, 16));
        w->casc_scan_bytes = cascade_scan_bytes(tiles);
        BC_HIP(hipMalloc(&w->casc_scan_tmp, std::max<size_t>(w->casc_scan_bytes, 16)));
        w->casc_tiles = tiles;
    }
    *out = w;
    return BC_OK;
}

int launch_hist_part(bc_lib* lib, const MatchOut& o, const TallyWs* w, uint64_t n, uint64_t tail_begin, hipStream_t st) {
    int rc = ensure_big_lds(k_hist_part, lib->device, HIST_LDS_BYTES);
    if (rc != BC_OK) return rc;
    const int slices = std::max(1, 256 / (int)o.part.nr);
    hipLaunchKernelGGL(k_hist_part, dim3(o.part.nr * slices), dim3(HIST_BLOCK), HIST_LDS_BYTES, st, o.part, o.best_idx, n,
                       tail_begin, lib->G, slices, o.counts, o.first_seen ? w->first_tile : nullptr);
    hipLaunchKernelGGL(k_first_refine, dim3((lib->G * 16 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, o.best_idx, n,
                       o.first_seen ? lib->G : 0u, w->first_tile, o.first_seen, o.base, w->ovf_count, w->ovf_list,
                       o.part.n_tiles);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

// which kernel ran last (tests assert the path, bc_lib_last_kernel)
void note_kernel(bc_lib* lib, const char* name) { lib->last_kernel.store(name, std::memory_order_relaxed); }

// the pair index is built when a 3-mismatch search first asks for it (ten tables: 13 MB and ~50 ms at 1e5 guides)
int ensure_pair_index(bc_lib* lib) {
    std::lock_guard<std::mutex> g(lib->ws_mutex);
    if (lib->pairs_tried) return BC_OK;
    int rc = build_pair_index(lib);
    if (rc == BC_OK) lib->pairs_tried = true;
    return rc;
}

int launch_match(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                 int max_mm, int algo, const MatchOut& o_in, hipStream_t st);

// A budget as a cascade: the complete pass at max_mm - 1 (results + its tally) runs first — a guide within fewer
// substitutions, once found, is the best hit under any larger budget, with the same tie count — then the second
// stage walks best_idx, takes the reads still marked BC_IDX_MISS through the pair index at max_mm, writes their results
// in place and tallies its hits with device atomics.  Fully asynchronous, no workspace.  Used for max_mm = 3 (the
// four 5-base seeds of a 20-mer put ~G / 1024 guides in a bucket) and for max_mm = 2 on libraries whose single-segment
// buckets are too long for the wave matcher (above ~6.5e5 20-mers: 7 / 7 / 6-base seeds cannot grow with the library,
// the pair index's keys can).  The first pass may itself be a cascade.
int launch_cascade(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n, int max_mm, const MatchOut& o_in,
                   hipStream_t st) {
    // positions travel as 32-bit values inside the second stage; the list form keeps 32 bytes of workspace per read of a
    // launch (every read may miss), so it takes very long inputs 2^27 reads at a time (4.3 GB)
    const uint64_t CH = cascade_uses_list(lib, n) ? (1ull << std::min(31, std::max(8, lib->knobs.pair_chunk_log2))) : (1ull << 31);
    for (uint64_t a = 0; a < n; a += CH) {
        const uint64_t m = std::min(CH, n - a);
        MatchOut o = o_in;
        o.best_idx += a;
        if (o.best_dist) o.best_dist += a;
        if (o.n_best) o.n_best += a;
        o.base += a;
        o.part = PartOut{};
        o.miss_pos = nullptr; o.miss_read = nullptr; o.miss_cnt = nullptr; o.miss_listed = nullptr;  // (an outer cascade's list is not ours)
        TallyWs* w = nullptr;
        bool listed = false;
        MatchOut o1 = o;
        if (cascade_uses_list(lib, m)) {  // the second stage works from a list: the first stage may as well write it
            int rcw = cascade_workspace_impl(lib, st, m, &w);
            if (rcw != BC_OK) return rcw;
            o1.miss_pos = w->casc_pos; o1.miss_read = w->casc_read; o1.miss_cnt = w->casc_cnt; o1.miss_listed = &listed;
        }
        int rc = launch_match(lib, d_reads + a, d_flags ? d_flags + a : nullptr, m, max_mm - 1, BC_ALGO_SEED, o1, st);
        if (rc != BC_OK) return rc;
        rc = launch_cascade_stage2(lib, max_mm, d_reads + a, m, o, st, w, listed);
        if (rc != BC_OK) return rc;
    }
    note_kernel(lib, max_mm == 3 ? "cascade3:d2+k_match_pairs" : "cascade2:d1+k_match_pairs");
    return BC_OK;
}

int launch_match(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                 int max_mm, int algo, const MatchOut& o_in, hipStream_t st) {
    if (n == 0) return BC_OK;
    if (max_mm < 0 || max_mm >= lib->k) {
        // a guide within d substitutions keeps a clean pigeonhole segment only when k > d; and "every k-mer matches" is
        // never what a caller means
        set_error("bc_match: max_mismatch %d must be >= 0 and < k = %d", max_mm, lib->k);
        return BC_EINVAL;
    }
    if (algo == BC_ALGO_AUTO) algo = (max_mm == 0) ? BC_ALGO_HASH : BC_ALGO_SEED;
    // d = 3: the d <= 2 pass, then the pair index for the reads it left (without a pair index — k outside 8..25 — the
    // four-segment kernels take every read: they spend their time on exactly those reads either way)
    if (algo == BC_ALGO_SEED && max_mm == 3 && lib->knobs.cascade) {
        int rc = ensure_pair_index(lib);
        if (rc != BC_OK) return rc;
        if (lib->knobs.cascade == 1 && lib->d_pairs && aligned16(o_in.best_idx)) return launch_cascade(lib, d_reads, d_flags, n, 3, o_in, st);
    }
    // d = 2 on a library the wave matcher cannot serve at d = 2 (buckets above 40 guides) but can at d = 1, or serves
    // more slowly than the cascade does (by bucket load, below; BC_CASCADE2_MIN_G = n: from n guides on, 0: never): the
    // d <= 1 pass, then the pair index for the rest — for budgets <= 2 only FOUR of its ten tables (bc_device.h, d_pairs2).  Measured, kernel ms
    // per 1e8 reads, wave matcher / cascade: 2.43 / 2.52 at 3e5 guides, 3.57 / 2.72 at 4e5, 4.62 / 3.39 at 5e5,
    // 6.13 / 3.70 at 6.5e5; 1e6: 6.54 with ten tables, 4.49 with four (profiles/r04_cascade2_ab.txt)
    if (algo == BC_ALGO_SEED && max_mm == 2 && lib->knobs.cascade == 1 && lib->knobs.lane && aligned16(o_in.best_idx)) {
        int q01, q2;
        const bool vec_ok = vector_io_ok(d_reads, o_in, n);
        // default (-1): by the load of the wave matcher's d = 2 buckets (guides per bucket of tables 0 / 1 / 2), weighted by
        // the share of the benchmark's reads that reach each table: L = load0 + 0.23 load1 + 0.10 load2 >= 36, and only
        // where table 2's seed is shorter than table 0's (load2 >= 3 load0; with three equal seeds — 18-, 21-, 24-mers —
        // the wave matcher wins up to its own limit of 40 per bucket).  Measured by PASS time (matcher + tally), wave
        // matcher / cascade, ms per 1e8 reads (profiles/r04_cascade2_ab_list.txt, r04_cascade2_crossover_k.txt):
        // 20-mers 2.66 / 2.98 at 3e5 guides, 3.87 / 3.30 at 4e5 (L = 36 at 3.6e5); 16-mers 1.98 / 2.23 at 6e4, 2.53 / 2.22
        // at 7.5e4 (L = 36 at 6.35e4); 18-mers 1.79 / 2.30 at 1e5, 2.59 / 2.68 at 1.5e5; 24-mers 3.12 / 5.08 at 1e6
        double ld[3] = {0.0, 0.0, 0.0};
        if (lib->slab_ok[2])
            for (int t = 0; t < 3; ++t) ld[t] = (double)lib->G / (double)(1ull << (2 * lib->slab[2].t[t].slen));
        const bool by_load = ld[2] >= 3.0 * ld[0] && ld[0] + 0.23 * ld[1] + 0.10 * ld[2] >= 36.0;
        const bool big = lib->knobs.cascade2_min_g > 0 ? lib->G >= (uint32_t)lib->knobs.cascade2_min_g
                                                       : (lib->knobs.cascade2_min_g < 0 && by_load);
        if ((big || !wave_kernel_serves(lib, 2, algo, vec_ok, &q01, &q2)) && wave_kernel_serves(lib, 1, algo, vec_ok, &q01, &q2)) {
            int rc = ensure_pair_index(lib);
            if (rc != BC_OK) return rc;
            if (lib->d_pairs) return launch_cascade(lib, d_reads, d_flags, n, 2, o_in, st);
        }
    }
    // tally: in-kernel global atomics for small launches; otherwise LDS-privatised, from the matcher's range-partitioned
    // records (k_hist_part) when the kernel that runs can write them, else by sweeping best_idx once per range (k_hist)
    const int tally_mode = lib->knobs.tally;  // 0 auto, 1 atomics, 2 k_hist, 3 records when possible
    const bool want_tally = o_in.counts || o_in.first_seen;
    // libraries of more than 36 ranges (7.4e5 guides): one sweep of best_idx per range (0.05 ms each per 1e8 reads) costs
    // more than device-scope atomics on counters that spread (1.6 - 2.2 ms per 1e8 reads at 5e5 - 3e6 guides)
    // (records serve up to PART_MAX_NR = 64 ranges — 1.3e6 guides — when the kernel that runs writes them)
    const uint32_t n_ranges = (lib->G + HIST_R - 1) / HIST_R;
    PartOut probe{};
    const bool sweeps_pay = n_ranges <= 36u || (tally_mode != 2 && part_geometry(lib, probe) &&
                                                kernel_writes_records(lib, d_reads, o_in, n, max_mm, algo));
    const bool use_lds = want_tally && aligned16(o_in.best_idx) &&
                         (tally_mode >= 2 || (tally_mode == 0 && n >= 4ull * lib->G && sweeps_pay));
    const uint64_t CH = 1ull << 31;  // reads per launch: 32-bit tile numbers and offsets stay far from wrapping
    for (uint64_t a = 0; a < n; a += CH) {
        const uint64_t m = std::min(CH, n - a);
        MatchOut o = o_in;
        o.best_idx += a;
        if (o.best_dist) o.best_dist += a;
        if (o.n_best) o.n_best += a;
        o.base += a;
        o.part = PartOut{};
        TallyWs* w = nullptr;
        if (use_lds) {
            o.counts = nullptr; o.first_seen = nullptr;
            if (tally_mode != 2 && part_geometry(lib, o.part)) {
                o.part.n_tiles = (uint32_t)((m + 255) >> 8);
                int rc = tally_workspace(lib, st, o.part.n_tiles, o.part.nr, o.part.cap_log2, &w);
                if (rc != BC_OK) return rc;
                o.part.rec = w->rec; o.part.ovf_flag = w->ovf_list;
            }
        }
        uint64_t tail_begin = m;
        bool part_done = false;
        int rc = launch_match_kernel(lib, d_reads + a, d_flags ? d_flags + a : nullptr, m, max_mm, algo, o, st, &part_done,
                                     &tail_begin);
        if (rc != BC_OK) return rc;
        if (use_lds) {
            MatchOut t = o;
            t.counts = o_in.counts; t.first_seen = o_in.first_seen;
            rc = part_done ? launch_hist_part(lib, t, w, m, tail_begin, st) : launch_hist(lib, t, m, st);
            if (rc != BC_OK) return rc;
        }
    }
    return BC_OK;
}

int launch_hash(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n, const MatchOut& o_in,
                hipStream_t st, bool* part_done, uint64_t* tail_begin) {
    bool vec = aligned16(d_reads) && aligned16(o_in.best_idx) && (!o_in.best_dist || (((uintptr_t)o_in.best_dist) & 3u) == 0) &&
               (!o_in.n_best || (((uintptr_t)o_in.n_best) & 3u) == 0) && n >= 4;
    MatchOut o = o_in;
    if (!vec) o.part = PartOut{};
    if (vec) {
        // one row of 256 four-read groups (a tile of 256 reads per wave) per workgroup: measured against four rows per
        // workgroup, 1e8 reads: 0.660 against 0.713 ms (the grid-stride loop only runs again above 2^22 rows)
        const uint64_t rows = (n / 4 + BLOCK - 1) / BLOCK;
        int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(rows, 1u << 22));
        if (lib->wide)
            hipLaunchKernelGGL((k_match_hash<true, 4>), dim3(grid), dim3(BLOCK), 0, st, d_reads, d_flags, n, lib->d_slots64,
                               lib->d_slots32, lib->d_guides, lib->k, lib->log_slots, lib->kmask, o);
        else
            hipLaunchKernelGGL((k_match_hash<false, 4>), dim3(grid), dim3(BLOCK), 0, st, d_reads, d_flags, n, lib->d_slots64,
                               lib->d_slots32, lib->d_guides, lib->k, lib->log_slots, lib->kmask, o);
        if (o.part.rec) { *part_done = true; *tail_begin = n & ~3ull; }
    } else {
        int grid = grid_for(n);
        if (lib->wide)
            hipLaunchKernelGGL((k_match_hash<true, 1>), dim3(grid), dim3(BLOCK), 0, st, d_reads, d_flags, n, lib->d_slots64,
                               lib->d_slots32, lib->d_guides, lib->k, lib->log_slots, lib->kmask, o);
        else
            hipLaunchKernelGGL((k_match_hash<false, 1>), dim3(grid), dim3(BLOCK), 0, st, d_reads, d_flags, n, lib->d_slots64,
                               lib->d_slots32, lib->d_guides, lib->k, lib->log_slots, lib->kmask, o);
    }
    note_kernel(lib, "k_match_hash");
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int launch_match_kernel(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                        int max_mm, int algo, const MatchOut& o, hipStream_t st, bool* part_done, uint64_t* tail_begin) {
    if (algo == BC_ALGO_HASH) {
        if (max_mm != 0) {
            set_error("bc_match: BC_ALGO_HASH requires max_mismatch == 0");
            return BC_EINVAL;
        }
        int rc = launch_hash(lib, d_reads, d_flags, n, o, st, part_done, tail_begin);
        if (rc != BC_OK) return rc;
    } else if (algo == BC_ALGO_SEED || algo == BC_ALGO_SEED_CSR) {
        if (max_mm < 0 || max_mm >= MAX_SEG) {
            set_error("bc_match: the seed index supports 0 <= max_mismatch <= %d", MAX_SEG - 1);
            return BC_EINVAL;
        }
        if (algo == BC_ALGO_SEED && max_mm == 3 && lib->knobs.cascade == 2 && lib->d_pairs) {  // BC_CASCADE=2: tests, timing
            int rc = launch_pairs(lib, max_mm, d_reads, d_flags, n, o, st);
            if (rc != BC_OK) return rc;
            note_kernel(lib, "k_match_pairs");
            return BC_OK;
        }
        // the slab form serves k_match_wq (narrow and wide entries) and, narrow only, the quad / single slab kernels
        const SlabParams& hp = lib->slab[max_mm];
        const int nseg = hp.nseg;
        const bool wide = lib->slab_wide[max_mm];
        const bool vec_ok = vector_io_ok(d_reads, o, n);
        int nq01 = 1, nq2 = 1;
        const bool wave_ok = wave_kernel_serves(lib, max_mm, algo, vec_ok, &nq01, &nq2);
        if (algo == BC_ALGO_SEED && lib->slab_ok[max_mm] && (!wide || wave_ok)) {
            const SlabParams* dp = lib->d_slab[max_mm];
            if (max_mm >= 1 && vec_ok && !lib->knobs.single_slab) {
                // default: the wave kernel (lane-owned reads, quad-cooperative loads, no hash table);
                // BC_LANE=0 selects the quad kernels
                if (wave_ok) {
#ifdef BC_DIAG_BUILD
                    const bool prof = lib->knobs.phase_prof != 0 && !wide && nq01 == 1;  // tuning aid: per-stage cycle sums on stderr
#else
                    constexpr bool prof = false;
                    (void)prof;
#endif
                    const int wgrid = (int)((n + WAVE_T - 1) / WAVE_T);
                    const bool listing = o.miss_cnt != nullptr && !wide;  // (the shapes the list form exists for: 8-byte entries)
#define BC_WAVE_LAUNCH(NSEG_, NQ2_, PROF_, WIDE_, NQ01_)                                                           \
    hipLaunchKernelGGL((k_match_wq<NSEG_, NQ2_, PROF_, WIDE_, NQ01_>), dim3(wgrid), dim3(64), 0, st, dp, d_reads, d_flags, n, \
                       lib->k, lib->kmask, max_mm | ((lib->knobs.diag & 0xFF) << 8) | ((n >= 4u * WAVE_T ? lib->knobs.prefetch : 0) << 16), o)
#define BC_WAVE_LIST(NSEG_, NQ2_, E8_)                                                                               \
    hipLaunchKernelGGL((k_match_wq<NSEG_, NQ2_, false, false, 1, true, E8_>), dim3(wgrid), dim3(64), 0, st, dp, d_reads, d_flags, n, \
                       lib->k, lib->kmask, max_mm | ((lib->knobs.diag & 0xFF) << 8) | ((n >= 4u * WAVE_T ? lib->knobs.prefetch : 0) << 16), o)
#define BC_WAVE_LIST2(NSEG_, NQ2_)  /* 128-byte slabs of tables 0/1 (libraries above 1.2e5 20-mers): a large library's d = 2 cascade starts here */ \
    hipLaunchKernelGGL((k_match_wq<NSEG_, NQ2_, false, false, 2, true, false>), dim3(wgrid), dim3(64), 0, st, dp, d_reads, d_flags, n, \
                       lib->k, lib->kmask, max_mm | ((lib->knobs.diag & 0xFF) << 8) | ((n >= 4u * WAVE_T ? lib->knobs.prefetch : 0) << 16), o)
#define BC_WAVE_E8(NSEG_, NQ2_)                                                                                      \
    hipLaunchKernelGGL((k_match_wq<NSEG_, NQ2_, false, false, 1, false, true>), dim3(wgrid), dim3(64), 0, st, dp, d_reads, d_flags, n, \
                       lib->k, lib->kmask, max_mm | ((lib->knobs.diag & 0xFF) << 8) | ((n >= 4u * WAVE_T ? lib->knobs.prefetch : 0) << 16), o)
#ifdef BC_DIAG_BUILD
#define BC_WAVE_P(NSEG_, NQ2_) do { if (prof) BC_WAVE_LAUNCH(NSEG_, NQ2_, true, false, 1); else if (listing && hp.e8.ok) BC_WAVE_LIST(NSEG_, NQ2_, true); else if (listing) BC_WAVE_LIST(NSEG_, NQ2_, false); else if (hp.e8.ok) BC_WAVE_E8(NSEG_, NQ2_); else BC_WAVE_LAUNCH(NSEG_, NQ2_, false, false, 1); } while (0)
#else
#define BC_WAVE_P(NSEG_, NQ2_) do { if (listing && hp.e8.ok) BC_WAVE_LIST(NSEG_, NQ2_, true); else if (listing) BC_WAVE_LIST(NSEG_, NQ2_, false); else if (hp.e8.ok) BC_WAVE_E8(NSEG_, NQ2_); else BC_WAVE_LAUNCH(NSEG_, NQ2_, false, false, 1); } while (0)
#endif
                    if (!wide && nq01 == 1) {
                        if (nseg == 2) BC_WAVE_P(2, 1);
                        else if (nq2 == 1) BC_WAVE_P(3, 1);
                        else if (nq2 == 2) BC_WAVE_P(3, 2);
                        else BC_WAVE_P(3, 4);
                    } else if (!wide && listing) {
                        if (nseg == 2) BC_WAVE_LIST2(2, 2);
                        else if (nq2 == 2) BC_WAVE_LIST2(3, 2);
                        else BC_WAVE_LIST2(3, 4);
                    } else if (!wide) {
                        if (nseg == 2) BC_WAVE_LAUNCH(2, 2, false, false, 2);
                        else if (nq2 == 2) BC_WAVE_LAUNCH(3, 2, false, false, 2);
                        else BC_WAVE_LAUNCH(3, 4, false, false, 2);
                    } else {
                        if (nseg == 2) BC_WAVE_LAUNCH(2, 2, false, true, 2);
                        else if (nq2 == 2) BC_WAVE_LAUNCH(3, 2, false, true, 2);
                        else BC_WAVE_LAUNCH(3, 4, false, true, 2);
                    }
#undef BC_WAVE_P
#undef BC_WAVE_LIST
#undef BC_WAVE_LIST2
#undef BC_WAVE_E8
#undef BC_WAVE_LAUNCH
                    note_kernel(lib, "k_match_wq");
                    BC_HIP(hipGetLastError());
                    if (o.part.rec) { *part_done = true; *tail_begin = n; }
                    if (listing && o.miss_listed) *o.miss_listed = true;  // (a cascade's first stage: the list is written)
#ifdef BC_DIAG_BUILD
                    if (prof) {
                        unsigned long long c[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        BC_HIP(hipStreamSynchronize(st));
                        BC_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_phase_cycles), sizeof(c)));
                        BC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
                        fprintf(stderr, "match stages (cycles/wave): 1 %.0f  2 %.0f  3 %.0f  4 %.0f  waves %llu\n",
                                (double)c[0] / c[7], (double)c[1] / c[7], (double)c[2] / c[7], (double)c[3] / c[7], c[7]);
                    }
#endif
                    return BC_OK;
                }
                // quad kernels (previous generation, bc_match_quad.hip): d = 3, buckets above 16 guides, BC_LANE=0
                int rc = launch_quad_fused(lib, max_mm, d_reads, d_flags, n, o, st);
                if (rc != BC_OK) return rc;
                note_kernel(lib, "k_match_fused");
            } else {
                int rc = launch_quad_slab(lib, max_mm, d_reads, d_flags, n, o, st);
                if (rc != BC_OK) return rc;
                note_kernel(lib, "k_match_slab");
            }
        } else {
            hipLaunchKernelGGL(k_match_seed, dim3(grid_for(n)), dim3(BLOCK), 0, st, lib->seed[max_mm], d_reads,
                               d_flags, n, lib->k, lib->kmask, max_mm, o);
            note_kernel(lib, "k_match_seed");
        }
    } else if (algo == BC_ALGO_BRUTE) {
        if (max_mm < 0 || max_mm > lib->k) {
            set_error("bc_match: bad max_mismatch %d", max_mm);
            return BC_EINVAL;
        }
        uint64_t per_block = (uint64_t)BLOCK * BRUTE_RPT;
        uint64_t blocks = (n + per_block - 1) / per_block;
        if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
        hipLaunchKernelGGL(k_match_brute, dim3((int)blocks), dim3(BLOCK), 0, st, lib->d_guides, lib->G, d_reads,
                           d_flags, n, lib->k, lib->kmask, max_mm, o);
        note_kernel(lib, "k_match_brute");
    } else {
        set_error("bc_match: unknown algo %d", algo);
        return BC_EINVAL;
    }
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // namespace

namespace bc {
int cascade_workspace(bc_lib* lib, hipStream_t st, uint64_t n, TallyWs** out) { return cascade_workspace_impl(lib, st, n, out); }
}  // namespace bc

constexpr int HOST_WORKERS = 8;  // worker threads of bc_match_host (two buffer sets each)
struct HostSlot {
    uint64_t cap = 0;
    uint64_t* d_r = nullptr;
    uint8_t *d_f = nullptr, *d_bd = nullptr, *d_nb = nullptr;
    int32_t* d_bi = nullptr;
    hipStream_t st = nullptr;
    // page-locked staging of one chunk (pageable caller buffers): reads + flags in, index + distance + ties out
    uint64_t pin_cap = 0;
    uint64_t* h_r = nullptr;
    uint8_t *h_f = nullptr, *h_bd = nullptr, *h_nb = nullptr;
    int32_t* h_bi = nullptr;
};

static void slot_release(HostSlot& s) {
    for (void* p : {(void*)s.d_r, (void*)s.d_f, (void*)s.d_bd, (void*)s.d_nb, (void*)s.d_bi}) if (p) (void)hipFree(p);
    for (void* p : {(void*)s.h_r, (void*)s.h_f, (void*)s.h_bd, (void*)s.h_nb, (void*)s.h_bi}) if (p) (void)hipHostFree(p);
    if (s.st) (void)hipStreamDestroy(s.st);
    s = HostSlot{};
}

static int slot_reserve_pinned(HostSlot& s, uint64_t cap) {
    if (cap <= s.pin_cap) return BC_OK;
    for (void* p : {(void*)s.h_r, (void*)s.h_f, (void*)s.h_bd, (void*)s.h_nb, (void*)s.h_bi}) if (p) (void)hipHostFree(p);
    s.pin_cap = 0;
    s.h_r = nullptr; s.h_bi = nullptr; s.h_f = s.h_bd = s.h_nb = nullptr;
    BC_HIP(hipHostMalloc((void**)&s.h_r, cap * 8, hipHostMallocDefault));
    BC_HIP(hipHostMalloc((void**)&s.h_bi, cap * 4, hipHostMallocDefault));
    BC_HIP(hipHostMalloc((void**)&s.h_f, cap, hipHostMallocDefault));
    BC_HIP(hipHostMalloc((void**)&s.h_bd, cap, hipHostMallocDefault));
    BC_HIP(hipHostMalloc((void**)&s.h_nb, cap, hipHostMallocDefault));
    s.pin_cap = cap;
    return BC_OK;
}

static int slot_reserve(HostSlot& s, uint64_t cap) {
    if (!s.st) BC_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    if (cap <= s.cap) return BC_OK;
    for (void* p : {(void*)s.d_r, (void*)s.d_f, (void*)s.d_bd, (void*)s.d_nb, (void*)s.d_bi}) if (p) (void)hipFree(p);
    s.cap = 0;
    s.d_r = nullptr; s.d_bi = nullptr; s.d_f = s.d_bd = s.d_nb = nullptr;
    BC_HIP(hipMalloc((void**)&s.d_r, cap * 8));
    BC_HIP(hipMalloc((void**)&s.d_bi, cap * 4));
    BC_HIP(hipMalloc((void**)&s.d_f, cap));
    BC_HIP(hipMalloc((void**)&s.d_bd, cap));
    BC_HIP(hipMalloc((void**)&s.d_nb, cap));
    s.cap = cap;
    return BC_OK;
}

namespace {

// scope guards: every exit path of the host entry points releases what it took
struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        BC_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return BC_OK;
    }
};
struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// is this host pointer page-locked memory the runtime knows (hipHostMalloc / hipHostRegister)?  Then DMA can use it as
// it is; pageable memory goes through the worker's page-locked staging buffers
static bool host_pinned(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// Host buffers in, per-read results out to the host; counts / first-seen are left ON THE DEVICE (d_c: uint32[G],
// d_fs: uint64[G], accumulated over the chunks).  The call is cut into chunks of 1M reads that worker threads take in
// turn.  Each worker owns TWO buffer sets (device buffers, page-locked staging, a stream each) and keeps one chunk in
// flight on the link / the device while it copies the results of the chunk before it out and the reads of the chunk
// after it in — round 3 had one set per worker: copy in, upload, kernels, download, wait, copy out, one after the
// other, and only the other workers to fill the link meanwhile (0.54 G reads/s = 7.6 GB/s over a link that carries
// several times that).  Staging is per buffer and only for PAGEABLE memory: a caller buffer that is page-locked
// (bc_host_alloc_pinned, hipHostRegister; barcoder_amd.GpuMatcher hands out its result arrays that way) is the DMA's
// source / target as it is.  h_best_idx == NULL is the counts-only mode (what heuristicount's semantics need,
// heuristicount.py:535-538): 8 bytes per read cross the link once, nothing per read comes back.
// BC_HOST_PIN=0: no staging at all, plain copies from the caller's memory.  The caller holds lib->host_mutex.
int match_host_chunks(bc_lib* lib, const uint64_t* h_reads, const uint8_t* h_flags, uint64_t n, int max_mismatch,
                      int algo, int32_t* h_best_idx, uint8_t* h_best_dist, uint8_t* h_n_best, uint32_t* d_c,
                      uint64_t* d_fs, uint64_t first_index_base) {
    const bool staging = lib->knobs.host_pin != 0 && n >= (1ull << 18);
    // which of the caller's buffers need the page-locked detour
    const bool st_r = staging && !host_pinned(h_reads), st_f = staging && h_flags && !host_pinned(h_flags);
    const bool st_bi = staging && h_best_idx && !host_pinned(h_best_idx), st_bd = staging && h_best_dist && !host_pinned(h_best_dist);
    const bool st_nb = staging && h_n_best && !host_pinned(h_n_best);
    const bool any_stage = st_r || st_f || st_bi || st_bd || st_nb;
    // 1M reads per chunk (8 MB up, 6 MB down: 14 MB of page-locked memory per buffer set); without staging: 8M
    const uint64_t CH = staging ? (1ull << 20) : (1ull << 23);
    const uint64_t n_chunks = (n + CH - 1) / CH;
    const int workers = (int)std::min<uint64_t>(staging ? HOST_WORKERS : 3, (n_chunks + 1) / 2);
    if (!lib->host_slots) lib->host_slots = new HostSlot[2 * HOST_WORKERS];
    HostSlot* slots = (HostSlot*)lib->host_slots;
    for (int w = 0; w < 2 * workers; ++w) {
        int rc = slot_reserve(slots[w], std::min(n, CH));
        if (rc == BC_OK && any_stage) rc = slot_reserve_pinned(slots[w], std::min(n, CH));
        if (rc != BC_OK) return rc;
    }
    BC_HIP(hipDeviceSynchronize());
    std::atomic<uint64_t> next{0};
    std::atomic<int> first_rc{BC_OK};
    std::string err;
    std::mutex err_mutex;
    auto work = [&](int w) {
        auto fail = [&](int rc) {
            int expect = BC_OK;
            if (first_rc.compare_exchange_strong(expect, rc)) {
                std::lock_guard<std::mutex> g(err_mutex);
                err = bc_last_error();
            }
        };
        if (hipSetDevice(lib->device) != hipSuccess) { set_error("hipSetDevice failed in a worker"); fail(BC_EHIP); return; }
        struct InFlight { bool busy = false; uint64_t a = 0, m = 0; } fl[2];
        // wait for the chunk in buffer set b and hand its results to the caller
        auto finish = [&](int b) -> int {
            if (!fl[b].busy) return BC_OK;
            HostSlot& s = slots[2 * w + b];
            fl[b].busy = false;
            BC_HIP(hipStreamSynchronize(s.st));
            const uint64_t a = fl[b].a, m = fl[b].m;
            if (st_bi) memcpy(h_best_idx + a, s.h_bi, m * 4);
            if (st_bd) memcpy(h_best_dist + a, s.h_bd, m);
            if (st_nb) memcpy(h_n_best + a, s.h_nb, m);
            return BC_OK;
        };
        auto start = [&](int b, uint64_t a, uint64_t m) -> int {
            HostSlot& s = slots[2 * w + b];
            const uint64_t* src_r = h_reads + a;
            const uint8_t* src_f = h_flags ? h_flags + a : nullptr;
            if (st_r) { memcpy(s.h_r, h_reads + a, m * 8); src_r = s.h_r; }
            if (st_f) { memcpy(s.h_f, h_flags + a, m); src_f = s.h_f; }
            BC_HIP(hipMemcpyAsync(s.d_r, src_r, m * 8, hipMemcpyHostToDevice, s.st));
            if (h_flags) BC_HIP(hipMemcpyAsync(s.d_f, src_f, m, hipMemcpyHostToDevice, s.st));
            MatchOut o{s.d_bi, h_best_dist ? s.d_bd : nullptr, h_n_best ? s.d_nb : nullptr, d_c, d_fs, first_index_base + a, PartOut{}};
            int rc = launch_match(lib, s.d_r, h_flags ? s.d_f : nullptr, m, max_mismatch, algo, o, s.st);
            if (rc != BC_OK) return rc;
            if (h_best_idx) BC_HIP(hipMemcpyAsync(st_bi ? s.h_bi : h_best_idx + a, s.d_bi, m * 4, hipMemcpyDeviceToHost, s.st));
            if (h_best_dist) BC_HIP(hipMemcpyAsync(st_bd ? s.h_bd : h_best_dist + a, s.d_bd, m, hipMemcpyDeviceToHost, s.st));
            if (h_n_best) BC_HIP(hipMemcpyAsync(st_nb ? s.h_nb : h_n_best + a, s.d_nb, m, hipMemcpyDeviceToHost, s.st));
            fl[b].busy = true; fl[b].a = a; fl[b].m = m;
            return BC_OK;
        };
        int b = 0, rc = BC_OK;
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_chunks || first_rc.load() != BC_OK) break;
            const uint64_t a = c * CH, m = std::min(CH, n - a);
            if ((rc = finish(b)) != BC_OK) break;        // the chunk before last: its buffers are free again
            if ((rc = start(b, a, m)) != BC_OK) break;   // ... and take this one, while the last chunk is still in flight
            b ^= 1;
        }
        for (int q = 0; q < 2; ++q) {  // drain, on every path (an error must not leave copies in flight)
            const int r2 = finish(q);
            if (rc == BC_OK) rc = r2;
        }
        if (rc != BC_OK) fail(rc);
    };
    if (workers == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int w = 0; w < workers; ++w) th.emplace_back(work, w);
        for (auto& t : th) t.join();
    }
    const int rc = first_rc.load();
    if (rc != BC_OK) { set_error("%s", err.c_str()); return rc; }
    BC_HIP(hipDeviceSynchronize());
    return BC_OK;
}

__global__ void k_widen_counts(const uint32_t* __restrict__ c32, uint64_t* __restrict__ c64, uint32_t G) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) c64[g] = c32[g];
}

// tallies of one pass in the form a job-wide reduction wants (64-bit sums; first-seen with the identity of a SIGNED
// 64-bit minimum for "unseen"), optionally leaving the accumulators reset for the next pass: one launch instead of
// five framework kernels between two passes
__global__ void k_tally_export(uint32_t* __restrict__ c32, uint64_t* __restrict__ fs, int64_t* __restrict__ c64,
                               int64_t* __restrict__ fmin, uint32_t G, int reset) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (c32) {
        if (c64) c64[g] = (int64_t)c32[g];
        if (reset) c32[g] = 0u;
    }
    if (fs) {
        const uint64_t f = fs[g];
        if (fmin) fmin[g] = f == 0xFFFFFFFFFFFFFFFFull ? INT64_MAX : (int64_t)f;
        if (reset) fs[g] = 0xFFFFFFFFFFFFFFFFull;
    }
}

}  // namespace

extern "C" {

int bc_device_count(int* n_out) {
    if (!n_out) { set_error("bc_device_count: null"); return BC_EINVAL; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *n_out = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return BC_ENODEV; }
    *n_out = n;
    return BC_OK;
}

int bc_lib_create(const uint64_t* h_guides, uint32_t n_guides, int k, int device, bc_lib** out) {
    if (!out || (!h_guides && n_guides) || k < 1 || k > 32 || n_guides == 0) {
        set_error("bc_lib_create: bad argument (G=%u k=%d)", n_guides, k);
        return BC_EINVAL;
    }
    const uint64_t km = kmer_mask(k);
    for (uint32_t i = 0; i < n_guides; ++i)
        if (h_guides[i] & ~km) {
            set_error("bc_lib_create: guide %u has bits outside a %d-mer (non-ACGT base?)", i, k);
            return BC_EINVAL;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("bc_lib_create: no HIP device visible (the matcher has no CPU path)");
        return BC_ENODEV;
    }
    if (device < 0 || device >= ndev) { set_error("bc_lib_create: device %d of %d", device, ndev); return BC_EINVAL; }
    BC_HIP(hipSetDevice(device));
    bc_lib* lib = new bc_lib();
    lib->device = device; lib->G = n_guides; lib->k = k; lib->kmask = km;
    // experiment knobs: the environment is read here, once per library, never per launch
    lib->knobs.lane = env_int("BC_LANE", 1);
#ifdef BC_DIAG_BUILD
    lib->knobs.diag = env_int("BC_DIAG", 0);
    lib->knobs.phase_prof = env_int("BC_PHASE_PROF", 0);
#endif
    lib->knobs.phase2_sm = env_int("BC_PHASE2_SM", 1);
    lib->knobs.single_slab = env_int("BC_SINGLE_SLAB", 0);
    lib->knobs.slab_lpr = env_int("BC_SLAB_LPR", 4);
    lib->knobs.tally = env_int("BC_TALLY", 0);
    lib->knobs.host_pin = env_int("BC_HOST_PIN", 1);
    lib->knobs.cascade = env_int("BC_CASCADE", 1);
    lib->knobs.e8 = env_int("BC_E8", 1);
    lib->knobs.pair_group_kb = env_int("BC_PAIR_GROUP_KB", 3300);
    lib->knobs.pair_pass_min = env_int("BC_PAIR_PASS_MIN_READS", 65536);
    lib->knobs.cascade2_min_g = env_int("BC_CASCADE2_MIN_G", -1);
    lib->knobs.pair_chunk_log2 = env_int("BC_PAIR_CHUNK_LOG2", 27);
    lib->knobs.pair_grid = std::max(1, env_int("BC_PAIR_GRID", 16384));
    lib->knobs.pair_four = env_int("BC_PAIR_FOUR", 1);
    lib->knobs.pair_bucket_pct = std::max(25, env_int("BC_PAIR_BUCKET_PCT", 800));
    lib->knobs.prefetch = std::max(0, std::min(255, env_int("BC_PREFETCH", 8)));
    {
        PartOut pg{};
        lib->part_ok = part_geometry_compute(n_guides, pg);
        lib->part_nr = pg.nr; lib->part_R = pg.R; lib->part_magic = pg.magic; lib->part_cap_log2 = pg.cap_log2;
    }
    lib->h_guides.assign(h_guides, h_guides + n_guides);
    const bool prof = env_int("BC_LIB_PROF", 0) != 0;  // tuning aid: phase times of this call on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t0 = now();
    int rc = dev_upload(lib, lib->h_guides, &lib->d_guides);
    if (rc == BC_OK) rc = compute_nn(lib);
    const double t_nn = ms_since(t0);
    t0 = now();
    if (rc == BC_OK) rc = build_hash(lib);
    const double t_hash = ms_since(t0);
    double t_seed[MAX_SEG] = {0, 0, 0, 0};
    for (int d = 0; d < MAX_SEG && rc == BC_OK; ++d) {
        t0 = now();
        rc = build_seed_index(lib, d);
        t_seed[d] = ms_since(t0);
    }
    if (prof)
        fprintf(stderr, "bc_lib_create: G=%u k=%d  neighbours %.1f ms, hash table %.1f ms, seed indices d=0..3 %.1f / %.1f / %.1f / %.1f ms\n",
                n_guides, k, t_nn, t_hash, t_seed[0], t_seed[1], t_seed[2], t_seed[3]);
    // (the pair index of the 3-mismatch matcher is built on first use: ensure_pair_index)
    if (rc != BC_OK) { bc_lib_destroy(lib); return rc; }
    *out = lib;
    return BC_OK;
}

int bc_lib_destroy(bc_lib* lib) {
    if (!lib) return BC_OK;
    (void)hipSetDevice(lib->device);
    (void)hipDeviceSynchronize();
    for (void* p : lib->owned) (void)hipFree(p);
    for (auto& w : lib->ws)
        for (void* p : {(void*)w.rec, (void*)w.ovf_list, (void*)w.ovf_count, (void*)w.first_tile, (void*)w.casc_read, (void*)w.casc_pos,
                        (void*)w.casc_state, (void*)w.casc_cnt, (void*)w.casc_off, (void*)w.casc_dread, (void*)w.casc_dpos,
                        (void*)w.casc_count, w.casc_scan_tmp}) if (p) (void)hipFree(p);
    if (lib->host_slots) {
        HostSlot* hs = (HostSlot*)lib->host_slots;
        for (int w = 0; w < 2 * HOST_WORKERS; ++w) slot_release(hs[w]);
        delete[] hs;
    }
    delete lib;
    return BC_OK;
}

int bc_lib_attach_text(bc_lib* lib, const char* h_text, int text_k) {
    if (!lib || !h_text || text_k < 33 || text_k > 1000 || lib->k != 32) {
        set_error("bc_lib_attach_text: a k = 32 library of text hashes and texts of 33 ... 1000 characters are needed");
        return BC_EINVAL;
    }
    BC_HIP(hipSetDevice(lib->device));
    void* p = nullptr;
    BC_HIP(hipMalloc(&p, (size_t)lib->G * (size_t)text_k));
    lib->owned.push_back(p);
    BC_HIP(hipMemcpy(p, h_text, (size_t)lib->G * (size_t)text_k, hipMemcpyHostToDevice));
    lib->d_text = (char*)p;
    lib->text_k = text_k;
    return BC_OK;
}

int bc_lib_release_stream(bc_lib* lib, void* stream) {
    if (!lib) { set_error("bc_lib_release_stream: null"); return BC_EINVAL; }
    std::lock_guard<std::mutex> g(lib->ws_mutex);
    for (auto it = lib->ws.begin(); it != lib->ws.end(); ++it) {
        if (it->stream != stream) continue;
        BC_HIP(hipSetDevice(lib->device));
        BC_HIP(hipStreamSynchronize((hipStream_t)stream));  // nothing of this stream may still use the buffers
        for (void* p : {(void*)it->rec, (void*)it->ovf_list, (void*)it->ovf_count, (void*)it->first_tile, (void*)it->casc_read,
                        (void*)it->casc_pos, (void*)it->casc_state, (void*)it->casc_cnt, (void*)it->casc_off, (void*)it->casc_dread,
                        (void*)it->casc_dpos, (void*)it->casc_count, it->casc_scan_tmp}) if (p) (void)hipFree(p);
        lib->ws.erase(it);
        break;
    }
    return BC_OK;
}

int bc_lib_workspace_count(bc_lib* lib, uint32_t* n_streams) {
    if (!lib || !n_streams) { set_error("bc_lib_workspace_count: null"); return BC_EINVAL; }
    std::lock_guard<std::mutex> g(lib->ws_mutex);
    *n_streams = (uint32_t)lib->ws.size();
    return BC_OK;
}

int bc_lib_info(const bc_lib* lib, uint32_t* n_guides, int* k, int* device) {
    if (!lib) { set_error("bc_lib_info: null"); return BC_EINVAL; }
    if (n_guides) *n_guides = lib->G;
    if (k) *k = lib->k;
    if (device) *device = lib->device;
    return BC_OK;
}

const char* bc_lib_last_kernel(const bc_lib* lib) {
    return lib ? lib->last_kernel.load(std::memory_order_relaxed) : "";
}

int bc_match_dev(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                 int max_mismatch, int algo, int32_t* d_best_idx, uint8_t* d_best_dist,
                 uint8_t* d_n_best, uint32_t* d_counts, uint64_t* d_first_seen,
                 uint64_t first_index_base, void* stream) {
    if (!lib || (n && (!d_reads || !d_best_idx))) { set_error("bc_match_dev: null argument"); return BC_EINVAL; }
    BC_HIP(hipSetDevice(lib->device));
    MatchOut o{d_best_idx, d_best_dist, d_n_best, d_counts, d_first_seen, first_index_base, PartOut{}};
    return launch_match(lib, d_reads, d_flags, n, max_mismatch, algo, o, (hipStream_t)stream);
}


int bc_tally_export_dev(uint32_t* d_counts, uint64_t* d_first_seen, int64_t* d_counts64, int64_t* d_first_min,
                        uint32_t n_guides, int reset, void* stream) {
    if (n_guides == 0) return BC_OK;
    if ((d_counts64 && !d_counts) || (d_first_min && !d_first_seen)) {
        set_error("bc_tally_export_dev: an output without its accumulator");
        return BC_EINVAL;
    }
    hipLaunchKernelGGL(k_tally_export, dim3((n_guides + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_counts,
                       d_first_seen, d_counts64, d_first_min, n_guides, reset);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int bc_match_time_dev(bc_lib* lib, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                      int max_mismatch, int algo, int32_t* d_best_idx, uint8_t* d_best_dist,
                      uint8_t* d_n_best, uint32_t* d_counts, uint64_t* d_first_seen, int iters,
                      void* stream, float* ms_out) {
    if (!lib || !ms_out || iters < 1) { set_error("bc_match_time_dev: bad argument"); return BC_EINVAL; }
    BC_HIP(hipSetDevice(lib->device));
    hipStream_t st = (hipStream_t)stream;
    Events ev;  // destroyed on every path
    BC_HIP(hipEventCreate(&ev.e0));
    BC_HIP(hipEventCreate(&ev.e1));
    MatchOut o{d_best_idx, d_best_dist, d_n_best, d_counts, d_first_seen, 0, PartOut{}};
    BC_HIP(hipEventRecord(ev.e0, st));
    int rc = BC_OK;
    for (int it = 0; it < iters && rc == BC_OK; ++it)
        rc = launch_match(lib, d_reads, d_flags, n, max_mismatch, algo, o, st);
    BC_HIP(hipEventRecord(ev.e1, st));
    BC_HIP(hipEventSynchronize(ev.e1));
    if (rc != BC_OK) return rc;  // *ms_out untouched: there is nothing to report
    float ms = 0;
    BC_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *ms_out = ms / iters;
    return BC_OK;
}

static int match_all_impl(bc_lib* lib, const uint64_t* d_queries, const uint8_t* d_flags, uint64_t n, int max_mismatch,
                          const PamGate& pg, bc_hit* d_hits, uint64_t cap, uint64_t* d_n_hits, void* stream) {
    if (!lib || !d_n_hits || (n && !d_queries) || (cap && !d_hits)) { set_error("bc_match_all_dev: null argument"); return BC_EINVAL; }
    if (max_mismatch < 0 || max_mismatch >= MAX_SEG || max_mismatch >= lib->k) {
        set_error("bc_match_all_dev: 0 <= max_mismatch <= %d and < k", MAX_SEG - 1);
        return BC_EINVAL;
    }
    if (n > 0xFFFFFFFFull) { set_error("bc_match_all_dev: at most 2^32-1 queries per call"); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    BC_HIP(hipSetDevice(lib->device));
    hipLaunchKernelGGL(k_match_all, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, lib->seed[max_mismatch],
                       d_queries, d_flags, n, lib->k, lib->kmask, max_mismatch, pg, d_hits, cap,
                       (unsigned long long*)d_n_hits);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int bc_match_all_dev(bc_lib* lib, const uint64_t* d_queries, const uint8_t* d_flags, uint64_t n,
                     int max_mismatch, bc_hit* d_hits, uint64_t cap, uint64_t* d_n_hits, void* stream) {
    return match_all_impl(lib, d_queries, d_flags, n, max_mismatch, PamGate{nullptr, 0, 0, 0}, d_hits, cap, d_n_hits, stream);
}

int bc_match_all_pam_dev(bc_lib* lib, const uint64_t* d_windows, const uint8_t* d_flags, uint64_t n,
                         int max_mismatch, const uint32_t* d_pam_bits, int64_t pam_shift, uint64_t pam_limit,
                         uint64_t wrap_len, bc_hit* d_hits, uint64_t cap, uint64_t* d_n_hits, void* stream) {
    if (!d_pam_bits) { set_error("bc_match_all_pam_dev: null PAM bitmap"); return BC_EINVAL; }
    return match_all_impl(lib, d_windows, d_flags, n, max_mismatch, PamGate{d_pam_bits, pam_shift, pam_limit, wrap_len},
                          d_hits, cap, d_n_hits, stream);
}

// Host buffers in, host results out (see match_host_chunks).  Counts / first-seen accumulate on the device across
// chunks and come back once.
int bc_match_host(bc_lib* lib, const uint64_t* h_reads, const uint8_t* h_flags, uint64_t n,
                  int max_mismatch, int algo, int32_t* h_best_idx, uint8_t* h_best_dist,
                  uint8_t* h_n_best, uint64_t* h_counts, uint64_t* h_first_seen,
                  uint64_t first_index_base) {
    if (!lib || (n && !h_reads)) { set_error("bc_match_host: null argument"); return BC_EINVAL; }
    if (n && !h_best_idx && (h_best_dist || h_n_best || (!h_counts && !h_first_seen))) {
        // counts-only mode: no per-read output at all, and a tally to return
        set_error("bc_match_host: without h_best_idx (counts-only) pass neither h_best_dist nor h_n_best, and h_counts or h_first_seen");
        return BC_EINVAL;
    }
    if (n == 0) return BC_OK;
    BC_HIP(hipSetDevice(lib->device));
    if (n > (1ull << 31)) {  // keep the 32-bit device counters safe: split very long calls
        const uint64_t half = 1ull << 31;
        for (uint64_t a = 0; a < n; a += half) {
            const uint64_t m = std::min(half, n - a);
            int rc = bc_match_host(lib, h_reads + a, h_flags ? h_flags + a : nullptr, m, max_mismatch, algo,
                                   h_best_idx ? h_best_idx + a : nullptr, h_best_dist ? h_best_dist + a : nullptr,
                                   h_n_best ? h_n_best + a : nullptr, h_counts, h_first_seen, first_index_base + a);
            if (rc != BC_OK) return rc;
        }
        return BC_OK;
    }
    std::lock_guard<std::mutex> guard(lib->host_mutex);  // the staging slots belong to one call at a time
    DevMem d_c, d_fs;  // freed on every path
    if (h_counts) {
        int rc = d_c.alloc((size_t)lib->G * 4);
        if (rc != BC_OK) return rc;
        BC_HIP(hipMemset(d_c.p, 0, (size_t)lib->G * 4));
    }
    if (h_first_seen) {
        int rc = d_fs.alloc((size_t)lib->G * 8);
        if (rc != BC_OK) return rc;
        BC_HIP(hipMemset(d_fs.p, 0xFF, (size_t)lib->G * 8));
    }
    int rc = match_host_chunks(lib, h_reads, h_flags, n, max_mismatch, algo, h_best_idx, h_best_dist, h_n_best,
                               (uint32_t*)d_c.p, (uint64_t*)d_fs.p, first_index_base);
    if (rc != BC_OK) return rc;
    if (d_c.p) {
        std::vector<uint32_t> c32(lib->G);
        BC_HIP(hipMemcpy(c32.data(), d_c.p, (size_t)lib->G * 4, hipMemcpyDeviceToHost));
        for (uint32_t g = 0; g < lib->G; ++g) h_counts[g] += c32[g];
    }
    if (d_fs.p) {
        std::vector<uint64_t> fs(lib->G);
        BC_HIP(hipMemcpy(fs.data(), d_fs.p, (size_t)lib->G * 8, hipMemcpyDeviceToHost));
        for (uint32_t g = 0; g < lib->G; ++g) h_first_seen[g] = std::min(h_first_seen[g], fs[g]);
    }
    return BC_OK;
}

// The host link's copy rates between page-locked memory and HBM: upload alone, download alone, and both at once on two
// streams (PCIe is full duplex) — the roofline bench.py holds bc_match_host against (secondary.host_pcie.roofline).
int bc_pcie_probe(int device, uint64_t bytes, int repeats, double* h2d_GBps, double* d2h_GBps, double* duplex_GBps) {
    if (bytes < 4096 || repeats < 1) { set_error("bc_pcie_probe: bad argument"); return BC_EINVAL; }
    BC_HIP(hipSetDevice(device));
    struct Bufs {
        void *h_up = nullptr, *h_dn = nullptr, *d_up = nullptr, *d_dn = nullptr;
        hipStream_t s_up = nullptr, s_dn = nullptr;
        ~Bufs() {
            if (h_up) (void)hipHostFree(h_up);
            if (h_dn) (void)hipHostFree(h_dn);
            if (d_up) (void)hipFree(d_up);
            if (d_dn) (void)hipFree(d_dn);
            if (s_up) (void)hipStreamDestroy(s_up);
            if (s_dn) (void)hipStreamDestroy(s_dn);
        }
    } B;
    BC_HIP(hipHostMalloc(&B.h_up, bytes, hipHostMallocDefault));
    BC_HIP(hipHostMalloc(&B.h_dn, bytes, hipHostMallocDefault));
    BC_HIP(hipMalloc(&B.d_up, bytes));
    BC_HIP(hipMalloc(&B.d_dn, bytes));
    BC_HIP(hipStreamCreateWithFlags(&B.s_up, hipStreamNonBlocking));
    BC_HIP(hipStreamCreateWithFlags(&B.s_dn, hipStreamNonBlocking));
    memset(B.h_up, 1, bytes);
    memset(B.h_dn, 2, bytes);
    BC_HIP(hipMemset(B.d_dn, 3, bytes));
    auto timed = [&](bool up, bool dn, double* out) -> int {
        for (int warm = 0; warm < 2; ++warm) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < (warm ? repeats : 1); ++r) {
                if (up) BC_HIP(hipMemcpyAsync(B.d_up, B.h_up, bytes, hipMemcpyHostToDevice, B.s_up));
                if (dn) BC_HIP(hipMemcpyAsync(B.h_dn, B.d_dn, bytes, hipMemcpyDeviceToHost, B.s_dn));
            }
            BC_HIP(hipStreamSynchronize(B.s_up));
            BC_HIP(hipStreamSynchronize(B.s_dn));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (warm && out) *out = (double)bytes * repeats * ((up ? 1 : 0) + (dn ? 1 : 0)) / dt / 1e9;
        }
        return BC_OK;
    };
    int rc;
    if ((rc = timed(true, false, h2d_GBps)) != BC_OK) return rc;
    if ((rc = timed(false, true, d2h_GBps)) != BC_OK) return rc;
    return timed(true, true, duplex_GBps);
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU, one process: a replica of the library per device, reads sharded in contiguous ranges, one all-reduce
// (RCCL over xGMI: SUM of the count vectors, MIN of the first-seen indices) at the end — the C-ABI form of what
// barcoder_amd/dist.py does with one process per GPU (heuristicount.py:703-737: Pool fan-out + Counter merge).
// RCCL is resolved at run time (dlopen), so the library loads on hosts without it; it is required only when the
// group spans several distinct devices.  A device listed more than once gets several replicas (rehearsal of the
// sharding on a box with fewer GPUs); RCCL cannot put two ranks on one device, so such a group — and a group of
// one — reduces through the host instead.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    typedef void* comm_t;
    void* h = nullptr;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::mutex m;
    bool load() {
        std::lock_guard<std::mutex> g(m);
        if (h) return true;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (int (*)(comm_t*, int, const int*))dlsym(h, "ncclCommInitAll");
        CommDestroy = (int (*)(comm_t))dlsym(h, "ncclCommDestroy");
        GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        AllReduce = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
        GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        const bool ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce && GetErrorString;
        if (!ok) { dlclose(h); h = nullptr; }
        return ok;
    }
};
Rccl g_rccl;
constexpr int NCCL_UINT64 = 5, NCCL_SUM = 0, NCCL_MIN = 3;  // rccl.h: ncclDataType_t / ncclRedOp_t

}  // namespace

struct bc_multi {
    std::vector<bc_lib*> libs;
    std::vector<int> devices;
    std::vector<Rccl::comm_t> comms;  // empty: reduce through the host
    std::mutex call_mutex;
};

int bc_multi_destroy(bc_multi* m) {
    if (!m) return BC_OK;
    for (auto c : m->comms) if (c) (void)g_rccl.CommDestroy(c);
    for (bc_lib* l : m->libs) bc_lib_destroy(l);
    delete m;
    return BC_OK;
}

int bc_multi_create(const uint64_t* h_guides, uint32_t n_guides, int k, const int* devices, int n_gpus, bc_multi** out) {
    if (!out || n_gpus < 1 || n_gpus > 64) { set_error("bc_multi_create: bad argument (n_gpus=%d)", n_gpus); return BC_EINVAL; }
    bc_multi* m = new bc_multi();
    for (int i = 0; i < n_gpus; ++i) {
        bc_lib* l = nullptr;
        const int dev = devices ? devices[i] : i;
        int rc = bc_lib_create(h_guides, n_guides, k, dev, &l);
        if (rc != BC_OK) { bc_multi_destroy(m); return rc; }
        m->libs.push_back(l);
        m->devices.push_back(dev);
    }
    bool distinct = true;
    for (int i = 0; i < n_gpus; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && m->devices[i] != m->devices[j];
    // (BC_MULTI_RCCL=1: a communicator also for a single device — a group of one still goes through ncclCommInitAll and
    // ncclAllReduce, which is how the RCCL branch is exercised on a one-GPU box: tests/test_dist_gpu.py)
    if ((n_gpus > 1 || env_int("BC_MULTI_RCCL", 0) == 1) && distinct) {
        if (!g_rccl.load()) { bc_multi_destroy(m); set_error("bc_multi_create: librccl is not available"); return BC_ENODEV; }
        m->comms.assign(n_gpus, nullptr);
        int r = g_rccl.CommInitAll(m->comms.data(), n_gpus, m->devices.data());
        if (r != 0) {
            m->comms.clear();
            bc_multi_destroy(m);
            set_error("ncclCommInitAll: %s", g_rccl.GetErrorString(r));
            return BC_EHIP;
        }
    }
    *out = m;
    return BC_OK;
}

int bc_multi_info(const bc_multi* m, int* n_gpus, int* uses_rccl) {
    if (!m) { set_error("bc_multi_info: null"); return BC_EINVAL; }
    if (n_gpus) *n_gpus = (int)m->libs.size();
    if (uses_rccl) *uses_rccl = m->comms.empty() ? 0 : 1;
    return BC_OK;
}

int bc_multi_gpu_match(bc_multi* m, const uint64_t* h_reads, const uint8_t* h_flags, uint64_t n, int max_mismatch,
                       int algo, int32_t* h_best_idx, uint8_t* h_best_dist, uint8_t* h_n_best, uint64_t* h_counts,
                       uint64_t* h_first_seen, uint64_t first_index_base) {
    if (!m || (n && (!h_reads || !h_best_idx))) { set_error("bc_multi_gpu_match: null argument"); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    std::lock_guard<std::mutex> call(m->call_mutex);
    const int P = (int)m->libs.size();
    const uint32_t G = m->libs[0]->G;
    const uint64_t per = (n + P - 1) / P;
    std::vector<DevMem> d_c(P), d_c64(P), d_fs(P);
    std::vector<int> rcs(P, BC_OK);
    std::vector<std::string> errs(P);
    auto shard = [&](int i) {
        bc_lib* lib = m->libs[i];
        auto run = [&]() -> int {
            BC_HIP(hipSetDevice(lib->device));
            std::lock_guard<std::mutex> guard(lib->host_mutex);
            int rc;
            if ((rc = d_c[i].alloc((size_t)G * 4)) != BC_OK) return rc;
            if ((rc = d_c64[i].alloc((size_t)G * 8)) != BC_OK) return rc;
            if ((rc = d_fs[i].alloc((size_t)G * 8)) != BC_OK) return rc;
            BC_HIP(hipMemset(d_c[i].p, 0, (size_t)G * 4));
            BC_HIP(hipMemset(d_fs[i].p, 0xFF, (size_t)G * 8));
            const uint64_t a = std::min(n, (uint64_t)i * per), b = std::min(n, a + per);
            if (b > a) {
                rc = match_host_chunks(lib, h_reads + a, h_flags ? h_flags + a : nullptr, b - a, max_mismatch, algo,
                                       h_best_idx + a, h_best_dist ? h_best_dist + a : nullptr,
                                       h_n_best ? h_n_best + a : nullptr, (uint32_t*)d_c[i].p, (uint64_t*)d_fs[i].p,
                                       first_index_base + a);
                if (rc != BC_OK) return rc;
            }
            // (null stream: match_host_chunks has synchronised the device, so every slot stream's tallies are in d_c)
            hipLaunchKernelGGL(k_widen_counts, dim3((G + 255) / 256), dim3(256), 0, 0, (const uint32_t*)d_c[i].p,
                               (uint64_t*)d_c64[i].p, G);
            BC_HIP(hipGetLastError());
            BC_HIP(hipDeviceSynchronize());
            return BC_OK;
        };
        rcs[i] = run();
        if (rcs[i] != BC_OK) errs[i] = bc_last_error();
    };
    if (P == 1) {
        shard(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < P; ++i) th.emplace_back(shard, i);
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < P; ++i)
        if (rcs[i] != BC_OK) { set_error("GPU %d: %s", m->devices[i], errs[i].c_str()); return rcs[i]; }
    std::vector<uint64_t> c64(G), fs(G);
    if (!m->comms.empty()) {
        // one all-reduce of each vector over RCCL / xGMI; every device ends with the job-wide result
        int r = g_rccl.GroupStart();
        hipError_t dev_err = hipSuccess;
        for (int i = 0; i < P && r == 0 && dev_err == hipSuccess; ++i) {
            dev_err = hipSetDevice(m->devices[i]);  // (no return inside the group: it must always be closed)
            if (dev_err != hipSuccess) break;
            r = g_rccl.AllReduce(d_c64[i].p, d_c64[i].p, G, NCCL_UINT64, NCCL_SUM, m->comms[i], 0);
            if (r == 0) r = g_rccl.AllReduce(d_fs[i].p, d_fs[i].p, G, NCCL_UINT64, NCCL_MIN, m->comms[i], 0);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r == 0) r = r2;
        if (dev_err != hipSuccess) { set_error("bc_multi_gpu_match: hipSetDevice: %s", hipGetErrorString(dev_err)); return BC_EHIP; }
        if (r != 0) { set_error("ncclAllReduce: %s", g_rccl.GetErrorString(r)); return BC_EHIP; }
        for (int i = 0; i < P; ++i) {
            BC_HIP(hipSetDevice(m->devices[i]));
            BC_HIP(hipDeviceSynchronize());
        }
        BC_HIP(hipSetDevice(m->devices[0]));
        BC_HIP(hipMemcpy(c64.data(), d_c64[0].p, (size_t)G * 8, hipMemcpyDeviceToHost));
        BC_HIP(hipMemcpy(fs.data(), d_fs[0].p, (size_t)G * 8, hipMemcpyDeviceToHost));
        for (uint32_t g = 0; g < G; ++g) {
            if (h_counts) h_counts[g] += c64[g];
            if (h_first_seen) h_first_seen[g] = std::min(h_first_seen[g], fs[g]);
        }
    } else {
        for (int i = 0; i < P; ++i) {
            BC_HIP(hipSetDevice(m->devices[i]));
            BC_HIP(hipMemcpy(c64.data(), d_c64[i].p, (size_t)G * 8, hipMemcpyDeviceToHost));
            BC_HIP(hipMemcpy(fs.data(), d_fs[i].p, (size_t)G * 8, hipMemcpyDeviceToHost));
            for (uint32_t g = 0; g < G; ++g) {
                if (h_counts) h_counts[g] += c64[g];
                if (h_first_seen) h_first_seen[g] = std::min(h_first_seen[g], fs[g]);
            }
        }
    }
    return BC_OK;
}

int bc_synth_reads_dev(bc_lib* lib, uint64_t seed, uint64_t first_index, uint64_t n, int mixture,
                       uint64_t* d_reads, void* stream) {
    if (!lib || (n && !d_reads) || mixture < 0 || (mixture > 1 && (mixture < 100 || mixture > 200))) { set_error("bc_synth_reads_dev: bad argument"); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    BC_HIP(hipSetDevice(lib->device));
    hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, lib->d_guides,
                       lib->G, lib->k, seed, first_index, n, mixture, d_reads);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

// ---- raw device memory helpers ----------------------------------------------
int bc_dev_set(int device) { BC_HIP(hipSetDevice(device)); return BC_OK; }
int bc_dev_alloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) { set_error("bc_dev_alloc: null"); return BC_EINVAL; }
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return BC_ENOMEM; }
    return BC_OK;
}
int bc_dev_free(void* d_ptr) { BC_HIP(hipFree(d_ptr)); return BC_OK; }
int bc_dev_memset(void* d_ptr, int value, size_t bytes, void* stream) {
    BC_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return BC_OK;
}
int bc_dev_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    BC_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    BC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BC_OK;
}
int bc_dev_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    BC_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    BC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BC_OK;
}
int bc_host_alloc_pinned(void** h_ptr, size_t bytes) {
    if (!h_ptr) { set_error("bc_host_alloc_pinned: null"); return BC_EINVAL; }
    hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return BC_ENOMEM; }
    return BC_OK;
}
int bc_host_free_pinned(void* h_ptr) { BC_HIP(hipHostFree(h_ptr)); return BC_OK; }
int bc_dev_sync(void* stream) {
    if (stream) BC_HIP(hipStreamSynchronize((hipStream_t)stream));
    else BC_HIP(hipDeviceSynchronize());
    return BC_OK;
}

}  // extern "C"
