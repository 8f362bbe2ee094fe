// bc_match_quad.hip — the previous generation of seed-index matchers (gfx950), kept for what the wave matcher
// (k_match_wq, bc_match.hip) does not cover: d = 3, tables whose buckets outgrow 8-entry slabs (more than 16 guides
// per bucket: libraries above ~2.6e5 20-mers, short k-mers), small or unaligned inputs, and the BC_LANE=0 /
// BC_SINGLE_SLAB=1 cross-checks of the knob tests.  Same slab index (narrow entries only), same outputs.
//   k_match_slab  : one simple kernel, LPR lanes per read, every table visited
//   k_match_fused : hash-table exact stage, LDS queue per 2048-read workgroup, one DPP quad per queued read
#include "bc_match_dev.h"

namespace {

template <int LPR, int NSEG>
__global__ __launch_bounds__(BLOCK) void k_match_slab(const SlabParams* __restrict__ spp,
                                                      const uint64_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ flags, uint64_t n,
                                                      int k, uint64_t kmask, int max_mm, MatchOut o) {
    // table descriptors stay in memory (scalar loads on demand): passing the
    // struct by value made the compiler spill SGPRs through v_writelane/v_readlane
    const SlabParams& sp = *spp;
    constexpr int MAXSTEP = (1 << SLAB_MAX_LOG2S) / (2 * LPR);
    const int sub = threadIdx.x & (LPR - 1);
    const uint64_t stride = (uint64_t)gridDim.x * (BLOCK / LPR);
    const int last_lane = (threadIdx.x & 63) | (LPR - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * (BLOCK / LPR) + (threadIdx.x / LPR); i < n; i += stride) {
        uint64_t r = reads[i];
        if (read_skipped(r, flags, i, k)) {
            if (sub == 0) emit(o, i, BC_IDX_SKIP, BC_DIST_NONE, 0);
            continue;
        }
        r &= kmask;
        const uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
        SlabBest b{0xFFFFFFFFu, (uint32_t)max_mm, 0ull};
        // round 0: the first table alone (an exact hit ends the search: guides are distinct)
        uint4 v0[MAXSTEP];
        {
            const SlabTable& T = sp.t[0];
            const uint32_t rest = slab_rest(T, rlo, rhi);
            uint32_t slab = seed_value(rlo, rhi, T.start, T.slen);
            do {
                slab_load<LPR, MAXSTEP>(T, slab, sub, v0);
                slab = slab_scan<LPR, MAXSTEP, 0>(T, v0, rest, b, last_lane);
            } while (slab != 0u);
        }
        b.cur = grp_min<LPR>(b.cur);
        if (NSEG > 1 && b.cur != 0u) {
            // round 1: every remaining table's primary slab is requested before any is scanned
            uint4 vr[NSEG > 1 ? NSEG - 1 : 1][MAXSTEP];
#pragma unroll
            for (int s = 1; s < NSEG; ++s)
                slab_load<LPR, MAXSTEP>(sp.t[s], seed_value(rlo, rhi, sp.t[s].start, sp.t[s].slen), sub, vr[s - 1]);
            auto round1 = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                const SlabTable& T = sp.t[s];
                const uint32_t rest = slab_rest(T, rlo, rhi);
                uint32_t slab = slab_scan<LPR, MAXSTEP, s>(T, vr[s - 1], rest, b, last_lane);
                while (slab != 0u) {  // overflow chain (rare)
                    slab_load<LPR, MAXSTEP>(T, slab, sub, v0);
                    slab = slab_scan<LPR, MAXSTEP, s>(T, v0, rest, b, last_lane);
                }
                b.cur = grp_min<LPR>(b.cur);
            };
            if constexpr (NSEG > 1) round1(std::integral_constant<int, 1>{});
            if constexpr (NSEG > 2) round1(std::integral_constant<int, 2>{});
            if constexpr (NSEG > 3) round1(std::integral_constant<int, 3>{});
        }
        // combine the LPR partial results: min key = (min distance, lowest index)
        const uint32_t key = grp_min<LPR>(b.key);
        const uint32_t md = key >> 24;
        const uint32_t mine = (md <= 3u) ? (uint32_t)((b.cnt >> (16 * (md & 3u))) & 0xFFFFu) : 0u;
        const uint32_t nb = grp_sum<LPR>(mine);
        if (sub == 0) {
            if (key != 0xFFFFFFFFu) emit(o, i, (int32_t)(key & SLAB_GIDX_MASK), md, nb);
            else emit(o, i, BC_IDX_MISS, BC_DIST_NONE, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// K2+K3 fused: one workgroup owns a tile of FUSED_T consecutive reads.
//   phase 1  exact stage: 4 reads per lane and step through the hash table; results
//            go to LDS, reads without an exact hit to an LDS queue;
//   phase 2  seed stage: DPP quads drain the LDS queue through the slab index.
//            Tables are visited in rounds: after tables 0..j every guide within
//            distance j has been seen (it keeps >= d+1-j clean segments), so a
//            read whose best is <= 1 after tables 0 and 1 never touches the large
//            third table;
//   phase 3  the tile's results leave LDS as full-width coalesced stores.
// No global queue, no placeholder writes, no scattered 6-byte result stores.
// ---------------------------------------------------------------------------
constexpr int FUSED_T = 2048;                 // reads per workgroup
constexpr int FUSED_IT = FUSED_T / (BLOCK * 4);

template <bool WIDE, int NSEG, int MSA, int MSB, bool SM>
__global__ __launch_bounds__(BLOCK) void k_match_fused(const SlabParams* __restrict__ spp,
                                                       const uint64_t* __restrict__ reads,
                                                       const uint8_t* __restrict__ flags, uint64_t n,
                                                       const uint64_t* __restrict__ s64,
                                                       const uint32_t* __restrict__ s32,
                                                       const uint64_t* __restrict__ guides, int k, int log_slots,
                                                       uint64_t kmask, int max_mm, MatchOut o) {
    constexpr int LPR = 4;
    // LDS: only the queue (20 KiB per workgroup, so registers, not LDS, set the occupancy).
    // An entry's 8 bytes hold the read until the seed stage has used it, then its result.
    __shared__ uint64_t q_slot[FUSED_T];
    __shared__ uint16_t q_pos[FUSED_T];
    __shared__ uint32_t q_n, q_head;
    const SlabParams& sp = *spp;
    const uint64_t tile0 = (uint64_t)blockIdx.x * FUSED_T;
    const uint32_t tn = (uint32_t)min((uint64_t)FUSED_T, n - tile0);  // reads in this tile
    if (threadIdx.x == 0) { q_n = 0; q_head = 0; }
    __syncthreads();

    // ---- phase 1: exact stage; exact hits (and skips) are final and stored at once, coalesced
#pragma unroll
    for (int it = 0; it < FUSED_IT; ++it) {
        const uint32_t p0 = (uint32_t)(it * BLOCK + threadIdx.x) * 4u;  // position in tile of this lane's 4 reads
        uint64_t r[4] = {0, 0, 0, 0};
        const bool full = p0 + 3 < tn;
        if (full) {
            load_reads4(reads + tile0 + p0, r);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < tn) r[j] = reads[tile0 + p0 + j];
        }
        int32_t idx[4];
        bool push[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = p0 + j < tn;
            if (!in || read_skipped(r[j], flags, tile0 + p0 + j, k)) idx[j] = BC_IDX_SKIP;
            else idx[j] = hash_lookup<WIDE>(r[j] & kmask, s64, s32, guides, k, log_slots, kmask);
            push[j] = in && idx[j] == BC_IDX_MISS;
        }
        if (full) {
            reinterpret_cast<int4*>(o.best_idx + tile0)[p0 >> 2] = make_int4(idx[0], idx[1], idx[2], idx[3]);
            if (o.best_dist) {
                uchar4 d;
                d.x = idx[0] >= 0 ? 0 : BC_DIST_NONE; d.y = idx[1] >= 0 ? 0 : BC_DIST_NONE;
                d.z = idx[2] >= 0 ? 0 : BC_DIST_NONE; d.w = idx[3] >= 0 ? 0 : BC_DIST_NONE;
                reinterpret_cast<uchar4*>(o.best_dist + tile0)[p0 >> 2] = d;
            }
            if (o.n_best) {
                uchar4 c;
                c.x = idx[0] >= 0; c.y = idx[1] >= 0; c.z = idx[2] >= 0; c.w = idx[3] >= 0;
                reinterpret_cast<uchar4*>(o.n_best + tile0)[p0 >> 2] = c;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < tn) {
                    o.best_idx[tile0 + p0 + j] = idx[j];
                    if (o.best_dist) o.best_dist[tile0 + p0 + j] = idx[j] >= 0 ? 0 : BC_DIST_NONE;
                    if (o.n_best) o.n_best[tile0 + p0 + j] = idx[j] >= 0;
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (p0 + j < tn) tally(o, tile0 + p0 + j, idx[j]);
        // wave-aggregated append to the LDS queue
        unsigned long long mask[4];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { mask[j] = __ballot(push[j]); total += (uint32_t)__popcll(mask[j]); }
        if (total) {
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(&q_n, total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (push[j]) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[j] >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask[j], 0u));
                    q_slot[base + rank] = r[j] & kmask;
                    q_pos[base + rank] = (uint16_t)(p0 + j);
                }
                base += (uint32_t)__popcll(mask[j]);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: seed stage over the LDS queue, one DPP quad per read
    const uint32_t qn = q_n;
    if constexpr (SM) {
        // State machine: every quad owns one queued read at a time and, per step, fetches and scans up to
        // two 64-byte pieces — both first-round slabs of a fresh read, an overflow slab, or the third
        // table's slab of a read that is not settled yet.  Quads in different stages share the same load
        // and scan instructions (only addresses and per-quad constants differ), so a wave never idles
        // through another quad's extra round, and reads are handed out dynamically from the LDS queue.
        // Geometry handled here: 8-entry slabs for tables 0 and 1, 8- or 16-entry slabs for table 2.
        constexpr int LPR = 4;
        const int sub = threadIdx.x & 3;
        const int last_lane = (threadIdx.x & 63) | 3;
        const uint32_t bias = (255u - (uint32_t)max_mm) << 24;
        const SlabTable& T0 = sp.t[0];
        const SlabTable& T1 = sp.t[NSEG > 1 ? 1 : 0];
        const SlabTable& T2 = sp.t[NSEG > 2 ? 2 : 0];
        const bool t2_two = NSEG > 2 && T2.log2S == 4;
        const uint4 none = make_uint4(0u, SLAB_EMPTY, 0u, SLAB_EMPTY);
        int state = 0;  // 0 need a read, 1 first round (tables 0,1 + their chains), 2 table 2 (+ chain), 3 no reads left
        uint32_t qi = 0, rlo = 0, rhi = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
        const uint4 *p0 = nullptr, *p1 = nullptr;
        uint32_t rest0 = 0, rest1 = 0, half0 = 1, half1 = 1;
        bool link0 = false, link1 = false;  // the piece ends with the slab's link slot
        for (;;) {
            if (state == 0) {
                uint32_t mine = 0;
                if (sub == 0) mine = atomicAdd(&q_head, 1u);
                qi = quad_perm<0x00>(mine);
                if (qi < qn) {
                    const uint64_t r = q_slot[qi];
                    rlo = (uint32_t)r; rhi = (uint32_t)(r >> 32);
                    kmin = 0xFFFFFFFFu; kmax = 0u;
                    p0 = reinterpret_cast<const uint4*>(T0.slabs + ((size_t)seed_value(rlo, rhi, T0.start, T0.slen) << 3));
                    rest0 = slab_rest(T0, rlo, rhi); half0 = T0.half; link0 = true;
                    if (NSEG > 1) {
                        p1 = reinterpret_cast<const uint4*>(T1.slabs + ((size_t)seed_value(rlo, rhi, T1.start, T1.slen) << 3));
                        rest1 = slab_rest(T1, rlo, rhi); half1 = T1.half; link1 = true;
                    } else { p1 = nullptr; link1 = false; }
                    state = 1;
                } else {
                    state = 3; p0 = nullptr; p1 = nullptr;
                }
            }
            if (__ballot(state != 3) == 0ull) break;
            // one load instruction per piece for every quad, whatever its stage
            uint4 v0 = none, v1 = none;
            if (p0) v0 = p0[sub];
            if (p1) v1 = p1[sub];
            {
                const uint32_t hm0 = (1u << half0) - 1u, hm1 = (1u << half1) - 1u;
                const uint32_t xa = rest0 ^ v0.x, xb = rest0 ^ v0.z, xc = rest1 ^ v1.x, xd = rest1 ^ v1.z;
                const uint32_t da = __popc((xa | (xa >> half0)) & hm0), db = __popc((xb | (xb >> half0)) & hm0);
                const uint32_t dc = __popc((xc | (xc >> half1)) & hm1), dd = __popc((xd | (xd >> half1)) & hm1);
                const uint32_t ka = (da << 24) | v0.y;
                const uint32_t kb = (db << 24) | v0.w;
                const uint32_t kc = (dc << 24) | v1.y;
                const uint32_t kd = (dd << 24) | v1.w;
                kmin = min(min(kmin, ka), min(kb, min(kc, kd)));
                kmax = max(max(kmax, ka + bias), max(kb + bias, max(kc + bias, kd + bias)));
            }
            // link slots: the second entry of lane 3 of a piece that ends its slab
            const uint32_t l0 = link0 ? quad_perm<0xFF>(v0.w == SLAB_LINK ? v0.z : 0u) : 0u;
            const uint32_t l1 = link1 ? quad_perm<0xFF>(v1.w == SLAB_LINK ? v1.z : 0u) : 0u;
            bool finish = false;
            if (state == 1) {
                if (l0 | l1) {  // overflow slabs of table 0 / 1: same stage, next step
                    p0 = l0 ? reinterpret_cast<const uint4*>(T0.slabs + ((size_t)l0 << 3)) : nullptr;
                    p1 = l1 ? reinterpret_cast<const uint4*>(T1.slabs + ((size_t)l1 << 3)) : nullptr;
                } else {
                    const uint32_t km = grp_min<LPR>(kmin);
                    const bool settled = (km >> 24) <= (uint32_t)max_mm && ((km >> SLAB_CLASS_SHIFT) & 3u) == (uint32_t)max_mm &&
                                         (grp_max<LPR>(kmax) - bias) == km;
                    if (NSEG > 2 && (km >> 24) > 1u && !settled) {
                        const uint4* b = reinterpret_cast<const uint4*>(
                            T2.slabs + ((size_t)seed_value(rlo, rhi, T2.start, T2.slen) << T2.log2S));
                        p0 = b; p1 = t2_two ? b + 4 : nullptr;
                        rest0 = rest1 = slab_rest(T2, rlo, rhi); half0 = half1 = T2.half;
                        link0 = !t2_two; link1 = t2_two;
                        state = 2;
                    } else finish = true;
                }
            } else if (state == 2) {
                const uint32_t l = t2_two ? l1 : l0;
                if (l) {
                    const uint4* b = reinterpret_cast<const uint4*>(T2.slabs + ((size_t)l << T2.log2S));
                    p0 = b; p1 = t2_two ? b + 4 : nullptr;
                } else finish = true;
            }
            if (finish) {
                const uint32_t km = grp_min<LPR>(kmin), kx = grp_max<LPR>(kmax);
                uint32_t key = km, nb = 1u;
                const bool hit = (km >> 24) <= (uint32_t)max_mm;
                if (hit && (kx - bias) != km) slab_slow<LPR, NSEG>(sp, rlo, rhi, max_mm, sub, last_lane, key, nb);
                if (sub == 0) {
                    const uint32_t ri = hit ? (key & SLAB_GIDX_MASK) : (uint32_t)BC_IDX_MISS;
                    const uint32_t rd = hit ? (key >> 24) : (uint32_t)BC_DIST_NONE;
                    const uint32_t rn_ = hit ? (nb > 255u ? 255u : nb) : 0u;
                    q_slot[qi] = (uint64_t)ri | ((uint64_t)rd << 32) | ((uint64_t)rn_ << 40);
                }
                state = 0;
            }
        }
    } else {
        const int sub = threadIdx.x & (LPR - 1);
        const int last_lane = (threadIdx.x & 63) | (LPR - 1);
        const uint32_t bias = (255u - (uint32_t)max_mm) << 24;
        constexpr int RA = NSEG < 2 ? NSEG : 2;  // round A: tables 0 and 1
        // software pipeline: the next queued read's round-A slabs are requested before the current
        // read is scanned, so a quad exposes about one memory round trip per read instead of two
        uint4 v[RA][MSA], vn[RA][MSA];
        uint32_t qi = threadIdx.x / LPR;
        uint64_t r = 0, rn = 0;
        if (qi < qn) {
            r = q_slot[qi];
#pragma unroll
            for (int s = 0; s < RA; ++s)
                slab_load<LPR, MSA>(sp.t[s], seed_value((uint32_t)r, (uint32_t)(r >> 32), sp.t[s].start, sp.t[s].slen), sub, v[s]);
        }
        for (; qi < qn; qi += BLOCK / LPR) {
            const uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
            uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
            const uint32_t qnext = qi + BLOCK / LPR;
            if (qnext < qn) {
                rn = q_slot[qnext];
#pragma unroll
                for (int s = 0; s < RA; ++s)
                    slab_load<LPR, MSA>(sp.t[s], seed_value((uint32_t)rn, (uint32_t)(rn >> 32), sp.t[s].start, sp.t[s].slen), sub, vn[s]);
            }
            {
#pragma unroll
                for (int s = 0; s < RA; ++s) {
                    const SlabTable& T = sp.t[s];
                    const uint32_t rest = slab_rest(T, rlo, rhi);
                    uint32_t slab = slab_scan_fast<LPR, MSA>(T, v[s], rest, bias, kmin, kmax, last_lane);
                    while (slab != 0u) {
                        slab_load<LPR, MSA>(T, slab, sub, v[s]);
                        slab = slab_scan_fast<LPR, MSA>(T, v[s], rest, bias, kmin, kmax, last_lane);
                    }
                }
            }
            kmin = grp_min<LPR>(kmin);
            // a single candidate so far that is isolated at radius 2*max_mm is the answer: nothing else can be
            // within max_mm of this read (k_nn_dist), so the remaining tables are not visited
            const bool settled = (kmin >> 24) <= (uint32_t)max_mm && ((kmin >> SLAB_CLASS_SHIFT) & 3u) == (uint32_t)max_mm &&
                                 (grp_max<LPR>(kmax) - bias) == kmin;
            if (NSEG > 2 && (kmin >> 24) > 1u && !settled) {
                // round B: the remaining tables can only add guides at distance >= 2
                constexpr int RB = NSEG > 2 ? NSEG - 2 : 1;
                uint4 v[RB][MSB];
#pragma unroll
                for (int s = 0; s < RB; ++s)
                    slab_load<LPR, MSB>(sp.t[s + 2], seed_value(rlo, rhi, sp.t[s + 2].start, sp.t[s + 2].slen), sub, v[s]);
#pragma unroll
                for (int s = 0; s < RB; ++s) {
                    const SlabTable& T = sp.t[s + 2];
                    const uint32_t rest = slab_rest(T, rlo, rhi);
                    uint32_t slab = slab_scan_fast<LPR, MSB>(T, v[s], rest, bias, kmin, kmax, last_lane);
                    while (slab != 0u) {
                        slab_load<LPR, MSB>(T, slab, sub, v[s]);
                        slab = slab_scan_fast<LPR, MSB>(T, v[s], rest, bias, kmin, kmax, last_lane);
                    }
                }
                kmin = grp_min<LPR>(kmin);
            }
            kmax = grp_max<LPR>(kmax);
            uint32_t key = kmin, nb = 1u;
            const bool hit = (kmin >> 24) <= (uint32_t)max_mm;
            if (hit && (kmax - bias) != kmin) slab_slow<LPR, NSEG>(sp, rlo, rhi, max_mm, sub, last_lane, key, nb);
            if (sub == 0) {  // the entry now carries its result: index | dist << 32 | n_best << 40
                const uint32_t ri = hit ? (key & SLAB_GIDX_MASK) : (uint32_t)BC_IDX_MISS;
                const uint32_t rd = hit ? (key >> 24) : (uint32_t)BC_DIST_NONE;
                const uint32_t rn_ = hit ? (nb > 255u ? 255u : nb) : 0u;
                q_slot[qi] = (uint64_t)ri | ((uint64_t)rd << 32) | ((uint64_t)rn_ << 40);
            }
            r = rn;
#pragma unroll
            for (int s = 0; s < RA; ++s)
#pragma unroll
                for (int st = 0; st < MSA; ++st) v[s][st] = vn[s][st];
        }
    }
    __syncthreads();

    // ---- phase 3: queued reads' results over the placeholders phase 1 left (same lines, still in L2)
    for (uint32_t qi = threadIdx.x; qi < qn; qi += BLOCK) {
        const uint64_t e = q_slot[qi];
        const int32_t idx = (int32_t)(uint32_t)e;
        if (idx == BC_IDX_MISS) continue;  // the placeholder already says so
        const uint64_t i = tile0 + q_pos[qi];
        o.best_idx[i] = idx;
        if (o.best_dist) o.best_dist[i] = (uint8_t)(e >> 32);
        if (o.n_best) o.n_best[i] = (uint8_t)(e >> 40);
        tally(o, i, idx);
    }
}

int grid_for(uint64_t work_items) {
    uint64_t b = (work_items + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    return (int)b;
}

}  // namespace

namespace bc {

int launch_quad_fused(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                      const MatchOut& o, hipStream_t st) {
    const SlabParams& hp = lib->slab[max_mm];
    const SlabParams* dp = lib->d_slab[max_mm];
    const int nseg = hp.nseg;
    const int grid = (int)((n + FUSED_T - 1) / FUSED_T);
    // steps (16-byte pieces per lane) of the largest slab in round A (tables 0,1) and round B
    int sa = 1, sb = 1;
    for (int s = 0; s < nseg; ++s) {
        int st_ = (1 << hp.t[s].log2S) / 8;
        if (s < 2) sa = std::max(sa, st_); else sb = std::max(sb, st_);
    }
#define BC_FUSED_LAUNCH(W_, NSEG_, MSA_, MSB_)                                                                   \
    hipLaunchKernelGGL((k_match_fused<W_, NSEG_, MSA_, MSB_, false>), dim3(grid), dim3(BLOCK), 0, st, dp,        \
                       d_reads, d_flags, n, lib->d_slots64, lib->d_slots32, lib->d_guides, lib->k, lib->log_slots, \
                       lib->kmask, max_mm, o)
#define BC_FUSED_SM(W_, NSEG_)                                                                                   \
    hipLaunchKernelGGL((k_match_fused<W_, NSEG_, 1, 1, true>), dim3(grid), dim3(BLOCK), 0, st, dp, d_reads,      \
                       d_flags, n, lib->d_slots64, lib->d_slots32, lib->d_guides, lib->k, lib->log_slots,         \
                       lib->kmask, max_mm, o)
#define BC_FUSED_W(NSEG_, MSA_, MSB_) do { if (lib->wide) BC_FUSED_LAUNCH(true, NSEG_, MSA_, MSB_); else BC_FUSED_LAUNCH(false, NSEG_, MSA_, MSB_); } while (0)
#define BC_FUSED_S(NSEG_)                                                        \
    do {                                                                         \
        if (sa <= 1 && sb <= 2) BC_FUSED_W(NSEG_, 1, 2);                         \
        else if (sa <= 2 && sb <= 2) BC_FUSED_W(NSEG_, 2, 2);                    \
        else if (sa <= 2) BC_FUSED_W(NSEG_, 2, 4);                               \
        else BC_FUSED_W(NSEG_, 4, 4);                                            \
    } while (0)
    // the quad state machine handles 8-entry slabs for tables 0,1 and 8/16-entry slabs for table 2
    const bool sm_ok = lib->knobs.phase2_sm && (nseg == 2 || nseg == 3) && hp.t[0].log2S == 3 &&
                       hp.t[1].log2S == 3 && (nseg == 2 || hp.t[2].log2S == 3 || hp.t[2].log2S == 4);
    if (sm_ok && nseg == 2) { if (lib->wide) BC_FUSED_SM(true, 2); else BC_FUSED_SM(false, 2); }
    else if (sm_ok && nseg == 3) { if (lib->wide) BC_FUSED_SM(true, 3); else BC_FUSED_SM(false, 3); }
    else switch (nseg) {
        case 2: BC_FUSED_S(2); break;
        case 3: BC_FUSED_S(3); break;
        default: BC_FUSED_S(4); break;
    }
#undef BC_FUSED_SM
#undef BC_FUSED_S
#undef BC_FUSED_W
#undef BC_FUSED_LAUNCH
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int launch_quad_slab(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n,
                     const MatchOut& o, hipStream_t st) {
    const SlabParams& hp = lib->slab[max_mm];
    const SlabParams* dp = lib->d_slab[max_mm];
    const int nseg = hp.nseg;
    uint32_t minS = 1u << SLAB_MAX_LOG2S;
    for (int s = 0; s < nseg; ++s) minS = std::min(minS, 1u << hp.t[s].log2S);
    const bool l8 = (lib->knobs.slab_lpr == 8 && minS >= 16);
#define BC_SLAB_LAUNCH(LPR_, NSEG_)                                                                          \
    hipLaunchKernelGGL((k_match_slab<LPR_, NSEG_>), dim3(grid_for(n * LPR_)), dim3(BLOCK), 0, st, dp, d_reads, \
                       d_flags, n, lib->k, lib->kmask, max_mm, o)
    switch (nseg) {
        case 1: if (l8) BC_SLAB_LAUNCH(8, 1); else BC_SLAB_LAUNCH(4, 1); break;
        case 2: if (l8) BC_SLAB_LAUNCH(8, 2); else BC_SLAB_LAUNCH(4, 2); break;
        case 3: if (l8) BC_SLAB_LAUNCH(8, 3); else BC_SLAB_LAUNCH(4, 3); break;
        default: if (l8) BC_SLAB_LAUNCH(8, 4); else BC_SLAB_LAUNCH(4, 4); break;
    }
#undef BC_SLAB_LAUNCH
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // namespace bc
