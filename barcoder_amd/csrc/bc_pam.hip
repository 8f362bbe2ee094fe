// bc_pam.hip — PAM site scan over a packed genome (K4) and k-mer window
// extraction, gfx950.  Bit-parallel: one lane tests 32 positions per step.
//
// Replaces the per-position Python loops of design_guides.py:22-49
// (find_sequences_with_barcode_and_pam), the regex scan of
// PAMProcessor.py:34-57 (GuideFinder.find_guides_from_pam) and the per-hit PAM
// test of targets.py:219-266 (pam_matches / extract_downstream_pam).
//
// Genome layout: word w = 32 positions in plane form (bc_common.h), plus one
// "invalid" bit per position.  HBM traffic: 12 B in per 32 positions, 8 B out.
#include "bc_device.h"

using namespace bc;

namespace {

struct PamParams {
    int len;            // pattern length, 1..32
    uint8_t fwd[32];    // IUPAC mask per pattern position, forward strand
    uint8_t rev[32];    // revcomp(pattern) as laid out on the forward strand
};

// 32 bits of plane `v` starting `sh` bits into word pair (w0, w1)
__device__ __forceinline__ uint32_t take32(uint32_t w0, uint32_t w1, int sh) {
    return sh == 0 ? w0 : ((w0 >> sh) | (w1 << (32 - sh)));
}

__device__ __forceinline__ uint32_t allowed(uint32_t lo, uint32_t hi, uint32_t m) {
    uint32_t a = 0;
    if (m & 1u) a |= ~hi & ~lo;  // A
    if (m & 2u) a |= ~hi & lo;   // C
    if (m & 4u) a |= hi & ~lo;   // G
    if (m & 8u) a |= hi & lo;    // T
    return a;
}

// One pattern position j of one strand, branch-free: sel[c] is all ones where the position accepts base code c.
// a = (hi ? (lo ? T : G) : (lo ? C : A)) bit by bit: three v_bfi.
struct PamSel { uint32_t s[4]; };
__device__ __forceinline__ uint32_t allowed_sel(uint32_t lo, uint32_t hi, const PamSel& m) {
    const uint32_t u1 = (lo & m.s[3]) | (~lo & m.s[2]), u0 = (lo & m.s[1]) | (~lo & m.s[0]);
    return (hi & u1) | (~hi & u0);
}
__device__ __forceinline__ PamSel sel_of(uint8_t m) {
    return PamSel{{(m & 1u) ? 0xFFFFFFFFu : 0u, (m & 2u) ? 0xFFFFFFFFu : 0u, (m & 4u) ? 0xFFFFFFFFu : 0u,
                   (m & 8u) ? 0xFFFFFFFFu : 0u}};
}
// (w1:w0) >> sh, low 32 bits; sh in 0..31: v_alignbit_b32
__device__ __forceinline__ uint32_t funnel(uint32_t w0, uint32_t w1, int sh) { return __builtin_amdgcn_alignbit(w1, w0, (uint32_t)sh); }

// pattern bitmaps (forward pattern, reverse-complemented pattern) for the 32 positions of word A; B = the next word.
// inv = not ACGT or past the end (bc_genome_pack sets the bits past the end of the last word; words past the
// last one are passed as all ones), out = past the end (only looked at when a pattern position carries
// BC_PAM_ANY_CHAR, the '.' of targets.py:224: an odd character matches there, the end of the sequence does not).
template <int LEN, bool ANY>
__device__ __forceinline__ void pam_eval(const PamParams& pp, uint32_t loA, uint32_t hiA, uint32_t invA, uint32_t outA,
                                         uint32_t loB, uint32_t hiB, uint32_t invB, uint32_t outB, uint32_t& f,
                                         uint32_t& r) {
    f = 0xFFFFFFFFu; r = 0xFFFFFFFFu;
    auto step = [&](int j) {
        const uint32_t lo = funnel(loA, loB, j), hi = funnel(hiA, hiB, j), odd = funnel(invA, invB, j);
        const uint32_t af = allowed_sel(lo, hi, sel_of(pp.fwd[j])) & ~odd, ar = allowed_sel(lo, hi, sel_of(pp.rev[j])) & ~odd;
        if (ANY) {
            const uint32_t in = ~funnel(outA, outB, j);
            f &= in & (af | ((pp.fwd[j] & 16u) ? odd : 0u));
            r &= in & (ar | ((pp.rev[j] & 16u) ? odd : 0u));
        } else {
            f &= af;
            r &= ar;
        }
    };
    if constexpr (LEN > 0) {  // compiled lengths: the position loop is straight-line code
#pragma unroll
        for (int j = 0; j < LEN; ++j) step(j);
    } else {
        for (int j = 0; j < pp.len; ++j) step(j);
    }
}

// bits of word w that lie at or past total_len
__device__ __forceinline__ uint32_t outside_bits(uint64_t w, uint64_t n_words, uint64_t total_len) {
    if (w + 1 < n_words) return 0u;
    if (w >= n_words) return 0xFFFFFFFFu;
    return (total_len & 31) ? ~((1u << (total_len & 31)) - 1u) : 0u;
}

// mask bit 4 (BC_PAM_ANY_CHAR): the pattern position also accepts a character that is not ACGT — the '.' that
// targets.py:224 (pam_matches) writes for N, as opposed to the [ATCG] of PAMProcessor.py:31 / design_guides.py:26.
// One lane = four consecutive words (128 positions): every access is a 16-byte vector access (two loads of bases,
// one of the invalid bits, one store per bitmap); the word after the four comes from the next lane (ds_bpermute),
// only the wave's last lane loads it.  Round 2's form (one word per lane, every word loaded twice, 8-byte loads,
// run-time pattern loop) reached 3.8 TB/s at 3.1e9 bp; two words per lane with 8-byte stores 4.4-5.0.
template <int LEN, bool ANY>
__global__ __launch_bounds__(BLOCK) void k_pam_scan(const uint64_t* __restrict__ bases,
                                                    const uint32_t* __restrict__ invalid,
                                                    uint64_t total_len, uint64_t n_words, PamParams pp,
                                                    uint32_t* __restrict__ out_fwd,
                                                    uint32_t* __restrict__ out_rev, bool vec) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t n_quads = (n_words + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    const uint32_t lane = threadIdx.x & 63u;
    struct Quad { uint32_t lo[4], hi[4], c[4]; };
    auto fetch = [&](uint64_t t) {
        const uint64_t w = 4 * t;
        Quad q;
#pragma unroll
        for (int j = 0; j < 4; ++j) { q.lo[j] = 0; q.hi[j] = 0; q.c[j] = 0xFFFFFFFFu; }
        if (w + 3 < n_words && vec) {
            const u32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bases) + 2 * t);
            const u32x4 b1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bases) + 2 * t + 1);
            const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(invalid) + t);
            q.lo[0] = b0.x; q.hi[0] = b0.y; q.lo[1] = b0.z; q.hi[1] = b0.w;
            q.lo[2] = b1.x; q.hi[2] = b1.y; q.lo[3] = b1.z; q.hi[3] = b1.w;
            q.c[0] = c.x; q.c[1] = c.y; q.c[2] = c.z; q.c[3] = c.w;
        } else {  // (pointers that are not 16-byte aligned, or the ragged last quad)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w + j < n_words) { const uint64_t b = bases[w + j]; q.lo[j] = (uint32_t)b; q.hi[j] = (uint32_t)(b >> 32); q.c[j] = invalid[w + j]; }
        }
        return q;
    };
    auto finish = [&](uint64_t t, Quad q) {
        const uint64_t w = 4 * t;
        // the word after the four: the next lane's first word
        uint32_t lo4 = (uint32_t)__shfl_down((int)q.lo[0], 1), hi4 = (uint32_t)__shfl_down((int)q.hi[0], 1), c4 = (uint32_t)__shfl_down((int)q.c[0], 1);
        if (lane == 63u) {
            lo4 = 0; hi4 = 0; c4 = 0xFFFFFFFFu;
            if (w + 4 < n_words) { const uint64_t b = bases[w + 4]; lo4 = (uint32_t)b; hi4 = (uint32_t)(b >> 32); c4 = invalid[w + 4]; }
        }
        // positions past total_len never match, whatever the caller left in the invalid bits there
        uint32_t o[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = outside_bits(w + j, n_words, total_len);
#pragma unroll
        for (int j = 0; j < 4; ++j) q.c[j] |= o[j];
        c4 |= o[4];
        uint32_t f[4], r[4];
#pragma unroll
        for (int j = 0; j < 3; ++j) pam_eval<LEN, ANY>(pp, q.lo[j], q.hi[j], q.c[j], o[j], q.lo[j + 1], q.hi[j + 1], q.c[j + 1], o[j + 1], f[j], r[j]);
        pam_eval<LEN, ANY>(pp, q.lo[3], q.hi[3], q.c[3], o[3], lo4, hi4, c4, o[4], f[3], r[3]);
        if (w + 3 < n_words && vec) {
            __builtin_nontemporal_store(u32x4{f[0], f[1], f[2], f[3]}, reinterpret_cast<u32x4*>(out_fwd) + t);
            __builtin_nontemporal_store(u32x4{r[0], r[1], r[2], r[3]}, reinterpret_cast<u32x4*>(out_rev) + t);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w + j < n_words) { out_fwd[w + j] = f[j]; out_rev[w + j] = r[j]; }
        }
    };
    // wave-uniform loop bounds (every lane of a wave takes part in the exchange); two tiles' loads in flight per lane
    uint64_t t0 = (uint64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63u);
    for (; t0 + stride < n_quads; t0 += 2 * stride) {
        const Quad a = fetch(t0 + lane), b = fetch(t0 + stride + lane);
        finish(t0 + lane, a); finish(t0 + stride + lane, b);
    }
    for (; t0 < n_quads; t0 += stride) finish(t0 + lane, fetch(t0 + lane));
}

// ---- bitmap -> ascending position list: popcount, scan, expand ---------------
constexpr int SCAN_WPT = 4;                       // words per thread
constexpr int SCAN_WPB = BLOCK * SCAN_WPT;        // words per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < BLOCK / 64; ++q) {
        if (q < wid) base += wsum[q];
        tot += wsum[q];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(BLOCK) void k_popc_blocks(const uint32_t* __restrict__ bm, uint64_t n_words,
                                                       uint32_t* __restrict__ block_sums) {
    uint64_t w0 = (uint64_t)blockIdx.x * SCAN_WPB + (uint64_t)threadIdx.x * SCAN_WPT;
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < SCAN_WPT; ++q)
        if (w0 + q < n_words) c += __popc(bm[w0 + q]);
    uint32_t tot;
    block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place, grand total to *n_out
__global__ __launch_bounds__(BLOCK) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t n_blocks,
                                                     unsigned long long* __restrict__ n_out) {
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += BLOCK) {
        uint32_t i = b0 + threadIdx.x;
        uint32_t v = i < n_blocks ? sums[i] : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan(v, &tot);
        if (i < n_blocks) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *n_out = carry;
}

__global__ __launch_bounds__(BLOCK) void k_expand(const uint32_t* __restrict__ bm, uint64_t n_words,
                                                  const uint32_t* __restrict__ block_off,
                                                  uint32_t* __restrict__ pos, uint64_t cap) {
    uint64_t w0 = (uint64_t)blockIdx.x * SCAN_WPB + (uint64_t)threadIdx.x * SCAN_WPT;
    uint32_t words[SCAN_WPT];
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < SCAN_WPT; ++q) {
        words[q] = (w0 + q < n_words) ? bm[w0 + q] : 0u;
        c += __popc(words[q]);
    }
    uint32_t tot;
    uint64_t at = (uint64_t)block_off[blockIdx.x] + block_exclusive_scan(c, &tot);
#pragma unroll
    for (int q = 0; q < SCAN_WPT; ++q) {
        uint32_t m = words[q];
        while (m) {
            int b = __ffs(m) - 1;
            m &= m - 1;
            if (at < cap) pos[at] = (uint32_t)((w0 + q) * 32 + b);
            ++at;
        }
    }
}

// ---- k-mer windows -----------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_windows(const uint64_t* __restrict__ bases,
                                                   const uint32_t* __restrict__ invalid,
                                                   uint64_t total_len, uint64_t n_words,
                                                   const uint32_t* __restrict__ pos, uint64_t n_pos,
                                                   int64_t offset, int k, int strand,
                                                   uint64_t* __restrict__ out,
                                                   uint8_t* __restrict__ flags) {
    const uint32_t km = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t < n_pos; t += stride) {
        int64_t sp = (int64_t)(pos ? (uint64_t)pos[t] : t) + offset;
        uint64_t p = (uint64_t)sp;
        uint64_t word = 0;
        bool bad = sp < 0 || (p + (uint64_t)k > total_len);
        if (!bad) {
            uint64_t w = p >> 5;
            int sh = (int)(p & 31);
            uint64_t b0 = bases[w];
            uint32_t i0 = invalid[w];
            uint64_t b1 = 0;
            uint32_t i1 = 0xFFFFFFFFu;
            if (w + 1 < n_words) { b1 = bases[w + 1]; i1 = invalid[w + 1]; }
            uint32_t lo = take32((uint32_t)b0, (uint32_t)b1, sh) & km;
            uint32_t hi = take32((uint32_t)(b0 >> 32), (uint32_t)(b1 >> 32), sh) & km;
            bad = (take32(i0, i1, sh) & km) != 0;
            if (strand) {
                lo = (__brev(~lo) >> (32 - k)) & km;
                hi = (__brev(~hi) >> (32 - k)) & km;
            }
            word = (uint64_t)lo | ((uint64_t)hi << 32);
        }
        if (bad) word = (k <= 31) ? BC_SKIP_BIT : 0;
        out[t] = word;
        if (flags) flags[t] = bad ? 1 : 0;
    }
}

// ---- fused site pipeline: PAM test + spacer window test -> compact site list, two launches ---------------------
// find_sites (pam.py) used to run scan -> popcount -> scan -> expand -> windows per strand (six launches, bitmaps and
// position lists through HBM, flagged windows dropped on the host).  Here a block stages its 1 024 words (+ halo) in
// LDS once and evaluates, per position p, "PAM matches at p AND the k-mer window at p + off is inside the sequence
// and pure ACGT" as bit operations; launch 1 counts such sites per block and strand, launch 2 re-evaluates (12 bytes
// per 32 positions: cheaper than writing and re-reading bitmaps), derives its output offset from the block counts
// and writes {PAM position, oriented spacer word} for every site — forward-strand sites in ascending order, then
// reverse-strand sites in ascending order, which is the order find_sites always had.
constexpr int SITE_WPT = 4;                   // words per thread
constexpr int SITE_WPB = BLOCK * SITE_WPT;    // words per block
constexpr int SITE_BEFORE = 1, SITE_AFTER = 3;  // halo words: windows reach from p - 32 to p + 32 + 31 + 32
constexpr uint32_t SITE_MAX_DIRECT = 4096;    // up to this many blocks every block sums the counts itself (two launches)

struct SiteParams {
    PamParams pp;
    int k;                 // spacer length, 1..32
    int off_fwd, off_rev;  // window start relative to the PAM position, per strand (-k or +pam_len)
};

struct SiteLds {
    uint32_t lo[SITE_WPB + SITE_BEFORE + SITE_AFTER];
    uint32_t hi[SITE_WPB + SITE_BEFORE + SITE_AFTER];
    uint32_t inv[SITE_WPB + SITE_BEFORE + SITE_AFTER];
};

__device__ __forceinline__ void site_stage(const uint64_t* __restrict__ bases, const uint32_t* __restrict__ invalid,
                                           uint64_t total_len, uint64_t n_words, uint64_t w0, SiteLds& L) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)(SITE_WPB + SITE_BEFORE + SITE_AFTER); i += BLOCK) {
        const int64_t w = (int64_t)w0 - SITE_BEFORE + (int64_t)i;
        uint32_t lo = 0, hi = 0, inv = 0xFFFFFFFFu;
        if (w >= 0 && (uint64_t)w < n_words) {
            const uint64_t b = bases[w];
            lo = (uint32_t)b; hi = (uint32_t)(b >> 32);
            inv = invalid[w] | outside_bits((uint64_t)w, n_words, total_len);
        }
        L.lo[i] = lo; L.hi[i] = hi; L.inv[i] = inv;
    }
    __syncthreads();
}

// bit s of the result: the k positions from (32 * word(li) + s + off) on are all valid.  off in [-32, 32].
__device__ __forceinline__ uint32_t window_ok(const SiteLds& L, int li, int off, int k) {
    const int wi = li + (off >> 5), sh = off & 31;  // arithmetic shift: -32..-1 -> -1, 0..31 -> 0, 32 -> 1
    const uint32_t a = ~L.inv[wi], b = ~L.inv[wi + 1], c = ~L.inv[wi + 2];
    const uint64_t V = (uint64_t)funnel(a, b, sh) | ((uint64_t)funnel(b, c, sh) << 32);
    // AND over a run of k by doubling: P(s) = AND of plen bits from s; acc collects the binary digits of k
    uint64_t acc = ~0ull, P = V;
    int pos = 0, plen = 1;
    for (int kk = k; kk; kk >>= 1) {
        if (kk & 1) { acc &= P >> pos; pos += plen; }
        P &= P >> plen;
        plen <<= 1;
    }
    return (uint32_t)acc;  // s + k - 1 <= 62: the zeros shifted in at the top never reach the low word
}

// valid-site bitmaps of the block's word number q (LDS index li)
template <int LEN>
__device__ __forceinline__ void site_bits(const SiteParams& sp, const SiteLds& L, int li, uint32_t& fv, uint32_t& rv) {
    uint32_t f, r;
    pam_eval<LEN, false>(sp.pp, L.lo[li], L.hi[li], L.inv[li], 0u, L.lo[li + 1], L.hi[li + 1], L.inv[li + 1], 0u, f, r);
    fv = f ? (f & window_ok(L, li, sp.off_fwd, sp.k)) : 0u;
    rv = r ? (r & window_ok(L, li, sp.off_rev, sp.k)) : 0u;
}

// launch 1: sites per block and strand -> counts[2 * block], counts[2 * block + 1]
template <int LEN>
__global__ __launch_bounds__(BLOCK) void k_sites_count(const uint64_t* __restrict__ bases, const uint32_t* __restrict__ invalid,
                                                       uint64_t total_len, uint64_t n_words, SiteParams sp,
                                                       uint32_t* __restrict__ counts) {
    __shared__ SiteLds L;
    const uint64_t w0 = (uint64_t)blockIdx.x * SITE_WPB;
    site_stage(bases, invalid, total_len, n_words, w0, L);
    uint32_t cf = 0, cr = 0;
#pragma unroll
    for (int q = 0; q < SITE_WPT; ++q) {
        const int li = SITE_BEFORE + (int)threadIdx.x * SITE_WPT + q;
        if (w0 + (uint64_t)(li - SITE_BEFORE) < n_words) {
            uint32_t fv, rv;
            site_bits<LEN>(sp, L, li, fv, rv);
            cf += __popc(fv); cr += __popc(rv);
        }
    }
    uint32_t tf, tr;
    block_exclusive_scan(cf, &tf);
    block_exclusive_scan(cr, &tr);
    if (threadIdx.x == 0) { counts[2 * blockIdx.x] = tf; counts[2 * blockIdx.x + 1] = tr; }
}

// many blocks (genomes above 134 Mbp): counts -> exclusive prefixes in place, totals in counts[2 * n_blocks ..]
__global__ __launch_bounds__(BLOCK) void k_sites_prefix(uint32_t* __restrict__ counts, uint32_t n_blocks) {
    uint32_t carry_f = 0, carry_r = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += BLOCK) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t vf = i < n_blocks ? counts[2 * i] : 0u, vr = i < n_blocks ? counts[2 * i + 1] : 0u;
        uint32_t tf, tr;
        const uint32_t ef = block_exclusive_scan(vf, &tf), er = block_exclusive_scan(vr, &tr);
        if (i < n_blocks) { counts[2 * i] = carry_f + ef; counts[2 * i + 1] = carry_r + er; }
        carry_f += tf; carry_r += tr;
    }
    if (threadIdx.x == 0) { counts[2 * n_blocks] = carry_f; counts[2 * n_blocks + 1] = carry_r; }
}

// launch 2: the same evaluation, then one record per site
template <int LEN, bool PREFIXED>
__global__ __launch_bounds__(BLOCK) void k_sites_emit(const uint64_t* __restrict__ bases, const uint32_t* __restrict__ invalid,
                                                      uint64_t total_len, uint64_t n_words, SiteParams sp,
                                                      const uint32_t* __restrict__ counts, uint32_t n_blocks,
                                                      uint32_t* __restrict__ out_pos, uint64_t* __restrict__ out_word,
                                                      uint64_t cap, uint64_t* __restrict__ n_out) {
    __shared__ SiteLds L;
    __shared__ uint32_t red[4][BLOCK / 64];
    const uint64_t w0 = (uint64_t)blockIdx.x * SITE_WPB;
    site_stage(bases, invalid, total_len, n_words, w0, L);
    // where this block's records go: forward sites of earlier blocks; all forward sites + reverse sites of earlier blocks
    uint32_t before_f, before_r, total_f, total_r;
    if (PREFIXED) {
        before_f = counts[2 * blockIdx.x]; before_r = counts[2 * blockIdx.x + 1];
        total_f = counts[2 * n_blocks]; total_r = counts[2 * n_blocks + 1];
    } else {
        uint32_t a[4] = {0, 0, 0, 0};  // before_f, before_r, total_f, total_r
        for (uint32_t i = threadIdx.x; i < n_blocks; i += BLOCK) {
            const uint32_t vf = counts[2 * i], vr = counts[2 * i + 1];
            a[2] += vf; a[3] += vr;
            if (i < blockIdx.x) { a[0] += vf; a[1] += vr; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a[c] += (uint32_t)__shfl_xor((int)a[c], d);
            if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = a[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[c] = 0; for (int q = 0; q < BLOCK / 64; ++q) a[c] += red[c][q]; }
        before_f = a[0]; before_r = a[1]; total_f = a[2]; total_r = a[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { n_out[0] = total_f; n_out[1] = total_r; }
    uint32_t fv[SITE_WPT], rv[SITE_WPT], cf = 0, cr = 0;
#pragma unroll
    for (int q = 0; q < SITE_WPT; ++q) {
        const int li = SITE_BEFORE + (int)threadIdx.x * SITE_WPT + q;
        fv[q] = 0; rv[q] = 0;
        if (w0 + (uint64_t)(li - SITE_BEFORE) < n_words) site_bits<LEN>(sp, L, li, fv[q], rv[q]);
        cf += __popc(fv[q]); cr += __popc(rv[q]);
    }
    uint32_t tf, tr;
    uint64_t at_f = (uint64_t)before_f + block_exclusive_scan(cf, &tf);
    uint64_t at_r = (uint64_t)total_f + before_r + block_exclusive_scan(cr, &tr);
    const int k = sp.k;
    const uint32_t km = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    auto emit = [&](uint32_t m, int li, int off, bool rev, uint64_t& at) {
        while (m) {
            const int b = __ffs(m) - 1;
            m &= m - 1;
            if (at < cap) {
                const int s = b + off;  // window start relative to the word, -32 .. 63
                const int wi = li + (s >> 5), sh = s & 31;
                uint32_t lo = funnel(L.lo[wi], L.lo[wi + 1], sh) & km, hi = funnel(L.hi[wi], L.hi[wi + 1], sh) & km;
                if (rev) { lo = (__brev(~lo) >> (32 - k)) & km; hi = (__brev(~hi) >> (32 - k)) & km; }
                out_pos[at] = (uint32_t)((w0 + (uint64_t)(li - SITE_BEFORE)) * 32 + (uint32_t)b);
                out_word[at] = (uint64_t)lo | ((uint64_t)hi << 32);
            }
            ++at;
        }
    };
#pragma unroll
    for (int q = 0; q < SITE_WPT; ++q) {
        const int li = SITE_BEFORE + (int)threadIdx.x * SITE_WPT + q;
        emit(fv[q], li, sp.off_fwd, false, at_f);
        emit(rv[q], li, sp.off_rev, true, at_r);
    }
}

int grid_for(uint64_t items) {
    uint64_t b = (items + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    return (int)b;
}

bool pam_has_any(const PamParams& pp) {
    bool any = false;
    for (int j = 0; j < pp.len; ++j) any = any || (pp.fwd[j] & 16u) || (pp.rev[j] & 16u);
    return any;
}

// pattern lengths 2..8 are compiled with the position loop unrolled; others take the run-time loop
int launch_pam_scan(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len, uint64_t n_words,
                    const PamParams& pp, uint32_t* d_fwd, uint32_t* d_rev, hipStream_t st) {
    const bool vec = (((uintptr_t)d_bases) & 15u) == 0 && (((uintptr_t)d_invalid) & 15u) == 0 &&
                     (((uintptr_t)d_fwd) & 15u) == 0 && (((uintptr_t)d_rev) & 15u) == 0;
    const bool any = pam_has_any(pp);
    const int grid = grid_for((n_words + 3) / 4);
#define BC_PAM_LAUNCH(LEN_)                                                                                            \
    do {                                                                                                               \
        if (any) hipLaunchKernelGGL((k_pam_scan<LEN_, true>), dim3(grid), dim3(BLOCK), 0, st, d_bases, d_invalid,       \
                                    total_len, n_words, pp, d_fwd, d_rev, vec);                                        \
        else hipLaunchKernelGGL((k_pam_scan<LEN_, false>), dim3(grid), dim3(BLOCK), 0, st, d_bases, d_invalid,          \
                                total_len, n_words, pp, d_fwd, d_rev, vec);                                            \
    } while (0)
    switch (pp.len) {
        case 2: BC_PAM_LAUNCH(2); break;
        case 3: BC_PAM_LAUNCH(3); break;
        case 4: BC_PAM_LAUNCH(4); break;
        case 5: BC_PAM_LAUNCH(5); break;
        case 6: BC_PAM_LAUNCH(6); break;
        case 7: BC_PAM_LAUNCH(7); break;
        case 8: BC_PAM_LAUNCH(8); break;
        default: BC_PAM_LAUNCH(0); break;
    }
#undef BC_PAM_LAUNCH
    return BC_OK;
}

uint8_t comp_mask(uint8_t m) {  // A<->T, C<->G; the any-character flag stays
    return (uint8_t)(((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3) | (m & 16));
}

}  // namespace

extern "C" {

int bc_pam_scan_dev(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len,
                    const uint8_t* pam_masks, int pam_len, uint32_t* d_fwd, uint32_t* d_rev,
                    void* stream) {
    if (!d_bases || !d_invalid || !pam_masks || !d_fwd || !d_rev || pam_len < 1 || pam_len > 32) {
        set_error("bc_pam_scan_dev: bad argument (pam_len=%d)", pam_len);
        return BC_EINVAL;
    }
    if (total_len == 0) return BC_OK;
    PamParams pp{};
    pp.len = pam_len;
    for (int j = 0; j < pam_len; ++j) {
        if ((pam_masks[j] & 15) == 0 || pam_masks[j] > 31) {
            set_error("bc_pam_scan_dev: pattern position %d has empty or invalid base set", j);
            return BC_EINVAL;
        }
        pp.fwd[j] = pam_masks[j];
        pp.rev[j] = comp_mask(pam_masks[pam_len - 1 - j]);
    }
    uint64_t n_words = (total_len + 31) / 32;
    int rc = launch_pam_scan(d_bases, d_invalid, total_len, n_words, pp, d_fwd, d_rev, (hipStream_t)stream);
    if (rc != BC_OK) return rc;
    BC_HIP(hipGetLastError());
    return BC_OK;
}

// the scan launched `iters` times back to back from this loop (no per-launch trip through the caller's language),
// HIP events around them on the launch stream: mean milliseconds per launch
int bc_pam_scan_time_dev(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len,
                         const uint8_t* pam_masks, int pam_len, uint32_t* d_fwd, uint32_t* d_rev, int iters,
                         void* stream, float* ms_out) {
    if (!ms_out || iters < 1) { set_error("bc_pam_scan_time_dev: bad argument"); return BC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BC_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return BC_EHIP; }
    int rc = BC_OK;
    hipError_t e = hipEventRecord(e0, st);
    for (int it = 0; it < iters && rc == BC_OK && e == hipSuccess; ++it)
        rc = bc_pam_scan_dev(d_bases, d_invalid, total_len, pam_masks, pam_len, d_fwd, d_rev, stream);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != BC_OK) return rc;
    if (e != hipSuccess) { set_error("bc_pam_scan_time_dev: %s", hipGetErrorString(e)); return BC_EHIP; }
    *ms_out = ms / iters;
    return BC_OK;
}

int bc_pam_scan_host(int device, const uint64_t* h_bases, const uint32_t* h_invalid,
                     uint64_t total_len, const uint8_t* pam_masks, int pam_len, uint32_t* h_fwd,
                     uint32_t* h_rev) {
    if (!h_bases || !h_invalid || !h_fwd || !h_rev) { set_error("bc_pam_scan_host: null argument"); return BC_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("bc_pam_scan_host: no HIP device visible (the scan has no CPU path)");
        return BC_ENODEV;
    }
    BC_HIP(hipSetDevice(device));
    uint64_t nw = (total_len + 31) / 32;
    if (nw == 0) return BC_OK;
    uint64_t* d_b = nullptr;
    uint32_t *d_i = nullptr, *d_f = nullptr, *d_r = nullptr;
    auto body = [&]() -> int {
        BC_HIP(hipMalloc((void**)&d_b, nw * 8));
        BC_HIP(hipMalloc((void**)&d_i, nw * 4));
        BC_HIP(hipMalloc((void**)&d_f, nw * 4));
        BC_HIP(hipMalloc((void**)&d_r, nw * 4));
        BC_HIP(hipMemcpy(d_b, h_bases, nw * 8, hipMemcpyHostToDevice));
        BC_HIP(hipMemcpy(d_i, h_invalid, nw * 4, hipMemcpyHostToDevice));
        int rc = bc_pam_scan_dev(d_b, d_i, total_len, pam_masks, pam_len, d_f, d_r, 0);
        if (rc != BC_OK) return rc;
        BC_HIP(hipDeviceSynchronize());
        BC_HIP(hipMemcpy(h_fwd, d_f, nw * 4, hipMemcpyDeviceToHost));
        BC_HIP(hipMemcpy(h_rev, d_r, nw * 4, hipMemcpyDeviceToHost));
        return BC_OK;
    };
    int rc = body();
    for (void* p : {(void*)d_b, (void*)d_i, (void*)d_f, (void*)d_r}) (void)hipFree(p);
    return rc;
}

int bc_sites_expand_dev(const uint32_t* d_bitmap, uint64_t n_words, uint32_t* d_pos, uint64_t cap,
                        uint64_t* d_n_out, uint32_t* d_tmp, void* stream) {
    if (!d_bitmap || !d_n_out || !d_tmp || (cap && !d_pos)) { set_error("bc_sites_expand_dev: null argument"); return BC_EINVAL; }
    if (n_words * 32 > 0xFFFFFFFFull) { set_error("bc_sites_expand_dev: positions must fit 32 bits"); return BC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (n_words == 0) {
        BC_HIP(hipMemsetAsync(d_n_out, 0, 8, st));
        return BC_OK;
    }
    uint32_t n_blocks = (uint32_t)((n_words + SCAN_WPB - 1) / SCAN_WPB);
    hipLaunchKernelGGL(k_popc_blocks, dim3(n_blocks), dim3(BLOCK), 0, st, d_bitmap, n_words, d_tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, st, d_tmp, n_blocks, (unsigned long long*)d_n_out);
    hipLaunchKernelGGL(k_expand, dim3(n_blocks), dim3(BLOCK), 0, st, d_bitmap, n_words, d_tmp, d_pos, cap);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int bc_windows_dev(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len,
                   const uint32_t* d_pos, uint64_t n_pos, int64_t offset, int k, int strand,
                   uint64_t* d_out, uint8_t* d_flags, void* stream) {
    if (!d_bases || !d_invalid || (n_pos && !d_out) || k < 1 || k > 32 || strand < 0 || strand > 1) {
        set_error("bc_windows_dev: bad argument (k=%d strand=%d)", k, strand);
        return BC_EINVAL;
    }
    if (n_pos == 0) return BC_OK;
    uint64_t n_words = (total_len + 31) / 32;
    hipLaunchKernelGGL(k_windows, dim3(grid_for(n_pos)), dim3(BLOCK), 0, (hipStream_t)stream, d_bases,
                       d_invalid, total_len, n_words, d_pos, n_pos, offset, k, strand, d_out, d_flags);
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // extern "C"

namespace {
int site_params(const uint8_t* pam_masks, int pam_len, int k, int direction, const char* who, SiteParams* sp) {
    if (!pam_masks || pam_len < 1 || pam_len > 32 || k < 1 || k > 32 || direction < 0 || direction > 1) {
        set_error("%s: bad argument (pam_len=%d k=%d direction=%d)", who, pam_len, k, direction);
        return BC_EINVAL;
    }
    *sp = SiteParams{};
    sp->pp.len = pam_len;
    for (int j = 0; j < pam_len; ++j) {
        if ((pam_masks[j] & 15) == 0 || pam_masks[j] > 15) {  // (no BC_PAM_ANY_CHAR here: a spacer next to it could not be ACGT-checked the same way)
            set_error("%s: pattern position %d has an empty or unsupported base set", who, j);
            return BC_EINVAL;
        }
        sp->pp.fwd[j] = pam_masks[j];
        sp->pp.rev[j] = comp_mask(pam_masks[pam_len - 1 - j]);
    }
    sp->k = k;
    // downstream PAM (direction 0): spacer ends where the PAM starts on its own strand (design_guides.py:33-38);
    // on the reverse strand that is the window right after the reverse-complemented pattern.  Upstream: mirrored.
    sp->off_fwd = direction == 0 ? -k : pam_len;
    sp->off_rev = direction == 0 ? pam_len : -k;
    return BC_OK;
}
}  // namespace

extern "C" {

int bc_pam_sites_count_dev(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len,
                           const uint8_t* pam_masks, int pam_len, int k, int direction, uint32_t* d_counts,
                           void* stream) {
    SiteParams sp;
    int rc = site_params(pam_masks, pam_len, k, direction, "bc_pam_sites_count_dev", &sp);
    if (rc != BC_OK) return rc;
    if (!d_bases || !d_invalid || !d_counts) { set_error("bc_pam_sites_count_dev: null argument"); return BC_EINVAL; }
    if (total_len > 0xFFFFFFFFull) { set_error("bc_pam_sites_count_dev: positions must fit 32 bits"); return BC_EINVAL; }
    if (total_len == 0) return BC_OK;
    const uint64_t n_words = (total_len + 31) / 32;
    const uint32_t n_blocks = (uint32_t)((n_words + SITE_WPB - 1) / SITE_WPB);
    hipStream_t st = (hipStream_t)stream;
#define BC_SITES_COUNT(LEN_) hipLaunchKernelGGL((k_sites_count<LEN_>), dim3(n_blocks), dim3(BLOCK), 0, st, d_bases, d_invalid, total_len, n_words, sp, d_counts)
    switch (pam_len) {
        case 2: BC_SITES_COUNT(2); break;
        case 3: BC_SITES_COUNT(3); break;
        case 4: BC_SITES_COUNT(4); break;
        case 5: BC_SITES_COUNT(5); break;
        case 6: BC_SITES_COUNT(6); break;
        default: BC_SITES_COUNT(0); break;
    }
#undef BC_SITES_COUNT
    BC_HIP(hipGetLastError());
    return BC_OK;
}

int bc_pam_sites_emit_dev(const uint64_t* d_bases, const uint32_t* d_invalid, uint64_t total_len,
                          const uint8_t* pam_masks, int pam_len, int k, int direction, uint32_t* d_counts,
                          uint32_t* d_pos, uint64_t* d_words, uint64_t cap, uint64_t* d_n_out, void* stream) {
    SiteParams sp;
    int rc = site_params(pam_masks, pam_len, k, direction, "bc_pam_sites_emit_dev", &sp);
    if (rc != BC_OK) return rc;
    if (!d_bases || !d_invalid || !d_counts || !d_n_out || (cap && (!d_pos || !d_words))) {
        set_error("bc_pam_sites_emit_dev: null argument");
        return BC_EINVAL;
    }
    if (total_len > 0xFFFFFFFFull) { set_error("bc_pam_sites_emit_dev: positions must fit 32 bits"); return BC_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (total_len == 0) { BC_HIP(hipMemsetAsync(d_n_out, 0, 16, st)); return BC_OK; }
    const uint64_t n_words = (total_len + 31) / 32;
    const uint32_t n_blocks = (uint32_t)((n_words + SITE_WPB - 1) / SITE_WPB);
    const bool prefixed = n_blocks > SITE_MAX_DIRECT;
    if (prefixed) hipLaunchKernelGGL(k_sites_prefix, dim3(1), dim3(BLOCK), 0, st, d_counts, n_blocks);
#define BC_SITES_EMIT(LEN_)                                                                                          \
    do {                                                                                                             \
        if (prefixed) hipLaunchKernelGGL((k_sites_emit<LEN_, true>), dim3(n_blocks), dim3(BLOCK), 0, st, d_bases,    \
                                         d_invalid, total_len, n_words, sp, d_counts, n_blocks, d_pos, d_words, cap, d_n_out); \
        else hipLaunchKernelGGL((k_sites_emit<LEN_, false>), dim3(n_blocks), dim3(BLOCK), 0, st, d_bases, d_invalid,  \
                                total_len, n_words, sp, d_counts, n_blocks, d_pos, d_words, cap, d_n_out);           \
    } while (0)
    switch (pam_len) {
        case 2: BC_SITES_EMIT(2); break;
        case 3: BC_SITES_EMIT(3); break;
        case 4: BC_SITES_EMIT(4); break;
        case 5: BC_SITES_EMIT(5); break;
        case 6: BC_SITES_EMIT(6); break;
        default: BC_SITES_EMIT(0); break;
    }
#undef BC_SITES_EMIT
    BC_HIP(hipGetLastError());
    return BC_OK;
}

}  // extern "C"
