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

// mask bit 4 (BC_PAM_ANY_CHAR): the pattern position also accepts a character that is not ACGT — the '.' that
// targets.py:224 (pam_matches) writes for N, as opposed to the [ATCG] of PAMProcessor.py:31 / design_guides.py:26
__global__ __launch_bounds__(BLOCK) void k_pam_scan(const uint64_t* __restrict__ bases,
                                                    const uint32_t* __restrict__ invalid,
                                                    uint64_t total_len, uint64_t n_words, PamParams pp,
                                                    uint32_t* __restrict__ out_fwd,
                                                    uint32_t* __restrict__ out_rev) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t w = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += stride) {
        uint64_t b0 = bases[w];
        uint32_t c0 = invalid[w];          // characters that are not ACGT
        uint32_t t0 = 0, t1 = 0xFFFFFFFFu; // positions at or past total_len
        uint64_t b1 = 0;
        uint32_t c1 = 0xFFFFFFFFu;
        if (w + 1 < n_words) { b1 = bases[w + 1]; c1 = invalid[w + 1]; t1 = 0; }
        if (w == n_words - 1 && (total_len & 31)) t0 = ~((1u << (total_len & 31)) - 1u);
        if (w + 2 == n_words && (total_len & 31)) t1 = ~((1u << (total_len & 31)) - 1u);
        const uint32_t lo0 = (uint32_t)b0, hi0 = (uint32_t)(b0 >> 32);
        const uint32_t lo1 = (uint32_t)b1, hi1 = (uint32_t)(b1 >> 32);
        uint32_t f = 0xFFFFFFFFu, r = 0xFFFFFFFFu;
        for (int j = 0; j < pp.len; ++j) {
            const uint32_t lo = take32(lo0, lo1, j), hi = take32(hi0, hi1, j);
            const uint32_t in = ~take32(t0, t1, j);         // inside the sequence
            const uint32_t odd = take32(c0, c1, j);         // not ACGT
            const uint32_t fa = (pp.fwd[j] & 16u) ? odd : 0u, ra = (pp.rev[j] & 16u) ? odd : 0u;
            f &= in & ((allowed(lo, hi, pp.fwd[j]) & ~odd) | fa);
            r &= in & ((allowed(lo, hi, pp.rev[j]) & ~odd) | ra);
        }
        // a pattern that would run past total_len meets a position outside the sequence: no match
        out_fwd[w] = f;
        out_rev[w] = r;
    }
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

int grid_for(uint64_t items) {
    uint64_t b = (items + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    return (int)b;
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
    PamParams pp;
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
    hipLaunchKernelGGL(k_pam_scan, dim3(grid_for(n_words)), dim3(BLOCK), 0, (hipStream_t)stream, d_bases,
                       d_invalid, total_len, n_words, pp, d_fwd, d_rev);
    BC_HIP(hipGetLastError());
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
