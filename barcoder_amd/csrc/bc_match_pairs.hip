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
// n_best needs no candidate list.  The kernel is exact for any read and any budget <= 3 — best distance, lowest
// index, tie count — and is used as the second stage of the d = 3 cascade (launch_cascade3 in bc_match.hip), where
// only the reads the d <= 2 pass left without a hit arrive.  One lane per read: the stage sees a tenth of the reads
// and is bound by its ten dependent-free L2 / Infinity-Cache requests per read, not by instructions.
//
// Geometry limits: an entry's rest holds at most 16 bases per plane, so key >= k - 16 bases; the key is
// capped at 9 bases (10 tables x 262 144 slabs).  k from 8 to 25; other lengths keep the four-segment index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>

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
__device__ __forceinline__ PairBest pairs_lookup(const PairParams& pp, uint32_t rlo, uint32_t rhi, uint32_t max_mm) {
    PairBest b{0xFFu, 0u, 0xFFFFFFFFu};
    for (int t = 0; t < pp.ntab; ++t) {
        const PairTable& T = pp.t[t];
        const uint32_t rest = pair_rest_plane(T, rlo) | (pair_rest_plane(T, rhi) << 16);
        uint32_t slab = pair_bucket(T, rlo, rhi);
        for (;;) {
            const uint4* p = reinterpret_cast<const uint4*>(T.rests + (size_t)slab * 16u);
            const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
            const uint32_t er[15] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z};
            const uint32_t cnt = w3.w & 15u, next = w3.w >> 4;
#pragma unroll
            for (int e = 0; e < 15; ++e) {
                const uint32_t x = rest ^ er[e];
                const uint32_t mm = (x | (x >> 16)) & 0xFFFFu;
                const uint32_t d = (uint32_t)__popc(mm);
                // counted here only if this table's pair is the first pair of the guide's clean pieces: what is left
                // of pieces a and b is clean, every earlier piece is dirty
                bool ok = (uint32_t)e < cnt && d <= max_mm && (mm & T.must_clean) == 0u;
#pragma unroll
                for (int c = 0; c < 3; ++c) ok = ok && (c >= (int)T.n_other || (mm & T.other[c]) != 0u);
                if (ok) {  // rare: a guide within the budget — its index comes from the parallel array
                    const uint32_t g = T.gidx[(size_t)slab * 16u + (uint32_t)e];
                    if (d < b.mind) { b.mind = d; b.cnt = 1u; b.bidx = g; }
                    else if (d == b.mind) { ++b.cnt; b.bidx = min(b.bidx, g); }
                }
            }
            if (!next) break;
            slab = next;
        }
    }
    return b;
}

// Second stage of the d = 3 cascade, fused with the search for its input: a workgroup walks its share of best_idx in
// sub-chunks of 4 096 results (sixteen per thread, 16-byte loads), queues the positions the d <= 2 pass left as
// BC_IDX_MISS in LDS, and whenever 256 are queued every thread takes one through the pair index and writes the
// result IN PLACE; hits are tallied with device atomics (5 % of the benchmark's reads).  No compacted copy of the
// missed reads, no device-side count for the host to wait for, no scatter pass: the launch is asynchronous and needs
// no workspace.  (Round 2: k_collect_missed 0.40 ms + k_match_pairs 1.49 ms + k_scatter_results 0.26 ms per 1e8
// reads and a host synchronisation in between.  Measured and dropped: a quad-per-read form of the lookup — one
// vector-L1 access per slab instead of four, but four lanes' worth of key arithmetic per read — 1.84 ms against 1.49
// for the lane-per-read form on the round-2 tables; the primary slabs of five tables requested before the first is
// scanned (twenty loads in flight per lane, 144 VGPRs, three waves per SIMD): 2.10 ms against 1.67 for one slab at
// a time at eight waves per SIMD.  The stage sits between three walls at once — L2-missing requests into 10.5 MB
// of tables (1e8 per 1e7 reads), four L1 tag look-ups per slab, and the scan's VALU work.)
constexpr int ST2_SUB = 4096;
__global__ __launch_bounds__(BLOCK) void k_cascade_stage2(const PairParams* __restrict__ ppp, const uint64_t* __restrict__ reads,
                                                         uint64_t n, uint64_t kmask, int max_mm_, int32_t* __restrict__ best_idx,
                                                         uint8_t* __restrict__ best_dist, uint8_t* __restrict__ n_best,
                                                         uint32_t* __restrict__ counts, uint64_t* __restrict__ first_seen,
                                                         uint64_t base) {
    __shared__ uint32_t q_pos[ST2_SUB + BLOCK];
    __shared__ uint32_t w_sum[BLOCK / 64];
    const PairParams& pp = *ppp;
    const uint32_t max_mm = (uint32_t)max_mm_;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32x4* pv = reinterpret_cast<const u32x4*>(best_idx);  // (16-byte aligned: the cascade's precondition)
    uint32_t qn = 0;  // queued positions (uniform over the block)
    auto serve = [&](uint32_t first, uint32_t count) {  // queue entries [first, first + count), one per thread
        if (threadIdx.x < count) {
            const uint32_t i = q_pos[first + threadIdx.x];
            const uint64_t r = reads[i] & kmask;
            const PairBest b = pairs_lookup(pp, (uint32_t)r, (uint32_t)(r >> 32), max_mm);
            if (b.mind <= max_mm) {
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
        const uint64_t c0 = sc * ST2_SUB;
        uint32_t mine = 0, pos[16];
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
            for (int c = 0; c < 4; ++c)
                if (e[c] == (uint32_t)BC_IDX_MISS) pos[mine++] = (uint32_t)(i + c);
        }
        // exclusive scan of `mine` over the block
        uint32_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        __syncthreads();  // (the previous round's serve() has read the queue)
        if (lane == 63u) w_sum[wv] = x;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) { before += (uint32_t)w < wv ? w_sum[w] : 0u; total += w_sum[w]; }
        uint32_t at = qn + before + x - mine;
        for (uint32_t j = 0; j < mine; ++j) q_pos[at + j] = pos[j];
        qn += total;
        __syncthreads();
        while (qn >= (uint32_t)BLOCK) {  // full rounds only: every lane busy
            serve(qn - BLOCK, BLOCK);
            qn -= BLOCK;
        }
    }
    __syncthreads();
    if (qn) serve(0, qn);
}

__global__ __launch_bounds__(BLOCK) void k_match_pairs(const PairParams* __restrict__ ppp, const uint64_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ flags, uint64_t n, int k, uint64_t kmask,
                                                      int max_mm, MatchOut o) {
    const PairParams& pp = *ppp;
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t r = reads[i];
    if (read_skipped(r, flags, i, k)) {
        emit(o, i, BC_IDX_SKIP, BC_DIST_NONE, 0u);
        return;
    }
    r &= kmask;
    const PairBest b = pairs_lookup(pp, (uint32_t)r, (uint32_t)(r >> 32), (uint32_t)max_mm);
    if (b.mind <= (uint32_t)max_mm) emit(o, i, (int32_t)(b.bidx & SLAB_GIDX_MASK), b.mind, b.cnt > 255u ? 255u : b.cnt);
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
    const int k = lib->k;
    if (k < 8 || k > 25 || lib->G >= (1u << SLAB_GIDX_BITS)) return BC_OK;  // other lengths: the four-segment index
    // five pieces, the longer ones first
    int p_start[5], p_len[5], at = 0;
    for (int c = 0; c < 5; ++c) { p_len[c] = k / 5 + (c < k % 5 ? 1 : 0); p_start[c] = at; at += p_len[c]; }
    // key length in bases: ~6 guides per bucket, at least k - 16 (the rest must fit 16 bases per plane), at most 9
    int want = 1;
    while (want < 9 && ((double)lib->G / (double)(1ull << (2 * want))) > 8.0) ++want;
    want = std::max(want, k - 16);
    if (want > 9) return BC_OK;
    // the ten tables are independent: one host thread each (50 ms single-threaded at 1e5 guides)
    struct Built { PairTable T{}; std::vector<uint32_t> rests, gidx; bool ok = false; };
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
        auto rest_mask = [&](int start, int len) {  // bases [start, start + len) in rest coordinates
            const uint32_t m = len >= 32 ? 0xFFFFFFFFu : (((1u << len) - 1u) << start);
            return pair_rest_plane(T, m);
        };
        T.must_clean = rest_mask(p_start[a] + la, p_len[a] - la) | rest_mask(p_start[b] + lb, p_len[b] - lb);
        T.n_other = 0;
        for (int c = 0; c < b; ++c)
            if (c != a) T.other[T.n_other++] = rest_mask(p_start[c], p_len[c]);
        const uint32_t nb = 1u << (2 * (la + lb));  // bucket = slab number, overflow slabs appended
        std::vector<uint32_t> cnt(nb + 1, 0u);
        std::vector<uint32_t> bucket_of(lib->G);
        for (uint32_t g = 0; g < lib->G; ++g) {
            const uint64_t v = lib->h_guides[g];
            bucket_of[g] = pair_bucket(T, (uint32_t)v, (uint32_t)(v >> 32));
            ++cnt[bucket_of[g] + 1];
        }
        for (uint32_t i = 0; i < nb; ++i) cnt[i + 1] += cnt[i];
        std::vector<uint32_t> order(lib->G), fill(cnt.begin(), cnt.end() - 1);
        for (uint32_t g = 0; g < lib->G; ++g) order[fill[bucket_of[g]]++] = g;  // ascending guide index inside a bucket
        // slabs of 15 rests + tail {next << 4 | count}; the guide indices in a parallel array
        std::vector<uint32_t>& rests = built[t].rests;
        std::vector<uint32_t>& gix = built[t].gidx;
        rests.assign((size_t)nb * 16, 0u);
        gix.assign((size_t)nb * 16, 0u);
        for (uint32_t bkt = 0; bkt < nb; ++bkt) {
            size_t cur = bkt;
            uint32_t pos = 0;
            for (uint32_t j = cnt[bkt]; j < cnt[bkt + 1]; ++j) {
                if (pos == 15) {
                    const size_t nxt = rests.size() / 16;
                    rests.resize(rests.size() + 16, 0u);
                    gix.resize(gix.size() + 16, 0u);
                    rests[cur * 16 + 15] = ((uint32_t)nxt << 4) | 15u;
                    cur = nxt;
                    pos = 0;
                }
                const uint32_t g = order[j];
                const uint64_t v = lib->h_guides[g];
                rests[cur * 16 + pos] = pair_rest_plane(T, (uint32_t)v) | (pair_rest_plane(T, (uint32_t)(v >> 32)) << 16);
                gix[cur * 16 + pos] = g;
                ++pos;
            }
            rests[cur * 16 + 15] = pos;  // last slab of the bucket: no next
        }
        if ((rests.size() / 16) >> 28) return;  // slab numbers travel in 28 bits
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
    }
    std::vector<PairParams> one(1, pp);
    PairParams* d_pp;
    int rc = upload(lib, one, &d_pp);
    if (rc != BC_OK) return rc;
    lib->d_pairs = d_pp;
    return BC_OK;
}

int launch_pairs(bc_lib* lib, int max_mm, const uint64_t* d_reads, const uint8_t* d_flags, uint64_t n, const MatchOut& o,
                 hipStream_t st) {
    if (!lib->d_pairs) { set_error("bc_match: no pair index for k = %d", lib->k); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    const uint64_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks >> 31) { set_error("bc_match: launch too large"); return BC_EINVAL; }
    hipLaunchKernelGGL(k_match_pairs, dim3((unsigned)blocks), dim3(BLOCK), 0, st, lib->d_pairs, d_reads, d_flags, n, lib->k,
                       lib->kmask, max_mm, o);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

// second stage of the cascade, in place over the first stage's results (see k_cascade_stage2): asynchronous
int launch_cascade_stage2(bc_lib* lib, int max_mm, const uint64_t* d_reads, uint64_t n, const MatchOut& o, hipStream_t st) {
    if (!lib->d_pairs) { set_error("bc_match: no pair index for k = %d", lib->k); return BC_EINVAL; }
    if (n == 0) return BC_OK;
    const uint64_t n_sub = (n + ST2_SUB - 1) / ST2_SUB;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_sub, 2048));  // 8 workgroups per CU
    hipLaunchKernelGGL(k_cascade_stage2, dim3(grid), dim3(BLOCK), 0, st, lib->d_pairs, d_reads, n, lib->kmask, max_mm, o.best_idx,
                       o.best_dist, o.n_best, o.counts, o.first_seen, o.base);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // namespace bc
