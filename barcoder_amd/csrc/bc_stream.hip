// bc_stream.hip — a reads file straight through the device: the raw bytes of an
// uncompressed .reads / .fastq file are uploaded in chunks and everything process_chunk does per read
// (heuristicount.py:428-562) happens in HBM: line splitting, whitespace strip (:127,:138), N-skip (:524), window
// slice (:527), flank test (:458-460), 2-bit packing, optional reverse complement (:532-533), the library match
// with its per-guide tally, and the collection of the flank-valid windows that match nothing (the "undocumented"
// barcodes, :455-457), which are grouped at the end by one device sort + run-length pass instead of per-chunk
// host work.  Only counts, first-seen keys, the undocumented table and the rare odd windows (flanks present but
// the window short or not pure ACGT: string semantics, handled by the caller) return to the host.
//
// Host side: worker threads take fixed byte chunks of the file in turn — pread into a pinned buffer, count the
// newlines that the chunk owns, upload, launch — each on its own stream with its own device buffers, so file
// reads, PCIe and kernels overlap.  A chunk owns the lines whose preceding newline lies in it (the line at the
// start of the range is owned by the first chunk); it is uploaded with one byte of left context and a slack of
// SLACK bytes on the right so that its last line is complete.
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "bc_device.h"

using namespace bc;

extern "C" {
typedef struct bc_reader bc_reader;
int bc_reader_open(const char* path, bc_reader** out);
int bc_reader_read_raw(bc_reader* r, char* dst, uint64_t cap, uint64_t* got);
int bc_reader_close(bc_reader* r);
}

namespace {

constexpr uint64_t CHUNK_MIN = 16ull << 20;   // bytes of file per device pass: at least (small files, compressed streams) ...
constexpr uint64_t CHUNK_MAX = 32ull << 20;   // ... and at most: see run_file
constexpr uint64_t SLACK = 1ull << 20;    // longest line that may run past a chunk's end

struct Geometry {
    int64_t start;
    int L_len, R_len, k, reverse, fastq;
    int text_k;  // > 0: barcodes of that many characters (more than 32): the window's 64-bit text hash stands for it
    char L[16], R[16];
};

// 1 bit per byte: is it '\n'?  One lane covers 32 bytes (two 16-byte loads).
__global__ __launch_bounds__(BLOCK) void k_nl_bitmap(const uint8_t* __restrict__ text, uint64_t nbytes,
                                                     uint32_t* __restrict__ bits, uint64_t n_words) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t w = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += stride) {
        const uint64_t a = w * 32;
        uint32_t m = 0;
        if (a + 32 <= nbytes) {
            const uint4* p = reinterpret_cast<const uint4*>(text + a);
            const uint4 v0 = p[0], v1 = p[1];
            const uint32_t d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t x = d[q] ^ 0x0A0A0A0Au;  // a zero byte <=> newline
#pragma unroll
                for (int b = 0; b < 4; ++b) m |= (((x >> (8 * b)) & 0xFFu) == 0u ? 1u : 0u) << (4 * q + b);
            }
        } else {
            for (int j = 0; j < 32 && a + j < nbytes; ++j) m |= (text[a + j] == '\n' ? 1u : 0u) << j;
        }
        bits[w] = m;
    }
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
__device__ __forceinline__ int dev_base_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

struct OddRec { uint32_t begin, len; uint64_t key; };

// One thread per owned line.  Line t begins after newline (t - virt) (or at buffer offset 0 for the virtual newline
// in front of the range) and ends at the next newline, or at `nbytes` when the file ends first.
// A FASTQ keeps the lines with (line_base + t) % 4 == 1.  Read r of the chunk = global read - gri0.
// n_nl_p: number of newlines the expand pass found in the buffer (device scalar); more_file: the buffer stops short
// of the end of the file, so a line without a closing newline in it is longer than the slack -> flags[1].
__global__ __launch_bounds__(BLOCK) void k_lines_to_windows(const uint8_t* __restrict__ text, uint64_t nbytes,
                                                            const uint32_t* __restrict__ nl,
                                                            const uint64_t* __restrict__ n_nl_p, uint64_t nl_cap,
                                                            uint64_t n_own, int virt, int more_file, uint64_t line_base,
                                                            uint64_t gri0, Geometry g, uint64_t key0,
                                                            uint64_t* __restrict__ packed, uint8_t* __restrict__ status,
                                                            OddRec* __restrict__ odd, uint32_t odd_cap,
                                                            uint32_t* __restrict__ flags, uint32_t* __restrict__ woff,
                                                            char* __restrict__ otext) {
    // otext (paired runs of barcodes of more than 32 nt): the barcode as the library spells it, kk characters at the
    // read's index — the pairing and the verification run after both files are in
    uint32_t* n_odd = flags;
    const int kk = g.text_k ? g.text_k : g.k;  // characters of a barcode
    const uint64_t n_nl = *n_nl_p;
    if (n_nl > nl_cap) {  // the position list was cut short: refuse
        if (blockIdx.x == 0 && threadIdx.x == 0) flags[1] = 2u;
        return;
    }
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    const int64_t wlen = (int64_t)g.L_len + kk + g.R_len;
    for (uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t < n_own; t += stride) {
        const uint64_t gl = line_base + t;
        uint64_t r;
        if (g.fastq) {
            if ((gl & 3u) != 1u) continue;
            r = (gl >> 2) - gri0;
        } else {
            r = gl - gri0;
        }
        const int64_t bi = (int64_t)t - virt;               // newline in front of the line
        uint64_t a = bi < 0 ? 0 : (uint64_t)nl[bi] + 1;
        uint64_t b = nbytes;
        if ((uint64_t)(bi + 1) < n_nl) b = (uint64_t)nl[bi + 1];
        else if (more_file) flags[1] = 1u;                  // no end of line inside the slack
        while (a < b && is_space(text[a])) ++a;              // str.strip()
        while (b > a && is_space(text[b - 1])) --b;
        const int64_t len = (int64_t)(b - a);
        uint64_t w = BC_SKIP_BIT;
        uint8_t st = BC_WIN_SKIP;
        bool has_n = false;
        for (uint64_t i = a; i < b; ++i) has_n = has_n || text[i] == 'N';
        if (!has_n) {
            const int64_t ws = min((int64_t)g.start, len), we = min((int64_t)g.start + wlen, len);
            const int64_t n = we - ws;
            const uint8_t* win = text + a + ws;
            bool has = n >= g.L_len && n >= g.R_len;
            for (int j = 0; has && j < g.L_len; ++j) has = win[j] == (uint8_t)g.L[j];
            for (int j = 0; has && j < g.R_len; ++j) has = win[n - g.R_len + j] == (uint8_t)g.R[j];
            if (has) {
                st = BC_WIN_ODD;  // flanks present: a short window or an odd character keeps string semantics
                if (n == wlen && g.text_k) {  // more than 32 nt: pure ACGT -> the hash of the barcode as the library spells it
                    const uint8_t* sq = win + g.L_len;
                    bool bad = false;
                    for (int j = 0; j < kk; ++j) bad = bad || dev_base_code(sq[j]) > 3;
                    if (!bad) {
                        if (g.reverse)
                            w = text_hash64_at((size_t)kk, [&](size_t i) {
                                const uint8_t c = sq[(size_t)kk - 1 - i];
                                return (uint8_t)(c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C');
                            });
                        else
                            w = text_hash64_at((size_t)kk, [&](size_t i) { return sq[i]; });
                        woff[r] = (uint32_t)(a + (uint64_t)ws + (uint64_t)g.L_len);
                        if (otext) {
                            char* dst = otext + r * (uint64_t)kk;
                            for (int j = 0; j < kk; ++j) {
                                const uint8_t c = g.reverse ? sq[kk - 1 - j] : sq[j];
                                dst[j] = (char)(!g.reverse ? c : c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C');
                            }
                        }
                        st = BC_WIN_OK;
                    }
                } else if (n == wlen) {
                    uint32_t lo = 0, hi = 0;
                    bool bad = false;
                    for (int j = 0; j < g.k; ++j) {
                        const int c = dev_base_code(win[g.L_len + j]);
                        bad = bad || c > 3;
                        lo |= (uint32_t)(c & 1) << j;
                        hi |= (uint32_t)((c >> 1) & 1) << j;
                    }
                    if (!bad) {
                        if (g.reverse) {
                            const uint32_t km = g.k >= 32 ? 0xFFFFFFFFu : ((1u << g.k) - 1u);
                            lo = (__brev(~lo) >> (32 - g.k)) & km;
                            hi = (__brev(~hi) >> (32 - g.k)) & km;
                        }
                        w = (uint64_t)lo | ((uint64_t)hi << 32);
                        st = BC_WIN_OK;
                    }
                }
                if (st == BC_WIN_ODD) {
                    const uint32_t at = atomicAdd(n_odd, 1u);
                    if (at < odd_cap) odd[at] = OddRec{(uint32_t)a, (uint32_t)len, key0 + r};
                }
            }
        }
        if (st != BC_WIN_OK) w = (g.k <= 31) ? BC_SKIP_BIT : 0;
        packed[r] = w;
        status[r] = st;
    }
}

// flank-valid windows that matched nothing -> (word, key) pairs appended to the run-wide list
__global__ __launch_bounds__(BLOCK) void k_collect_misses(const int32_t* __restrict__ best_idx,
                                                          const uint64_t* __restrict__ packed, uint64_t n, uint64_t kmask,
                                                          uint64_t key0, uint64_t* __restrict__ m_words,
                                                          uint64_t* __restrict__ m_keys,
                                                          unsigned long long* __restrict__ cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    const uint64_t n_round = (n + 63) & ~63ull;  // whole waves stay together for the ballots
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_round; r += stride) {
        const bool miss = r < n && best_idx[r] == BC_IDX_MISS;
        const unsigned long long m = __ballot(miss);
        if (m == 0ull) continue;
        const uint32_t lane = threadIdx.x & 63u;
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if ((int)lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        if (miss) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            m_words[base + rank] = packed[r] & kmask;
            m_keys[base + rank] = key0 + r;
        }
    }
}

// ---- barcodes of more than 32 nt: the exact table matched the window's text HASH; the text decides ---------------------
__device__ __forceinline__ uint8_t oriented_char(const uint8_t* sq, int kk, int reverse, int j) {
    if (!reverse) return sq[j];
    const uint8_t c = sq[kk - 1 - j];
    return (uint8_t)(c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C');
}

// every hit against the library entry's text (a hash collision becomes a miss), then the tally of the verified hits
__global__ __launch_bounds__(BLOCK) void k_long_verify_tally(int32_t* __restrict__ best_idx, const uint8_t* __restrict__ text,
                                                            const uint32_t* __restrict__ woff, const char* __restrict__ lib_text,
                                                            int kk, int reverse, uint64_t n, uint64_t key0,
                                                            uint32_t* __restrict__ counts, uint64_t* __restrict__ first_seen) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
        const int32_t g = best_idx[r];
        if (g < 0) continue;
        const uint8_t* sq = text + (woff ? (uint64_t)woff[r] : r * (uint64_t)kk);  // (no offsets: a store of kk characters per read)
        const char* lt = lib_text + (size_t)g * (size_t)kk;
        bool same = true;
        for (int j = 0; j < kk && same; ++j) same = oriented_char(sq, kk, reverse, j) == (uint8_t)lt[j];
        if (!same) { best_idx[r] = BC_IDX_MISS; continue; }
        atomicAdd(&counts[g], 1u);
        const unsigned long long v = key0 + r;
        if (v < first_seen[g]) atomicMin((unsigned long long*)&first_seen[g], v);
    }
}

// flank-valid windows that matched nothing -> (hash, key) pairs and the window's text
__global__ __launch_bounds__(BLOCK) void k_collect_misses_long(const int32_t* __restrict__ best_idx, const uint64_t* __restrict__ packed,
                                                              const uint8_t* __restrict__ text, const uint32_t* __restrict__ woff,
                                                              int kk, int reverse, uint64_t n, uint64_t key0,
                                                              uint64_t* __restrict__ m_words, uint64_t* __restrict__ m_keys,
                                                              char* __restrict__ m_text, unsigned long long* __restrict__ cursor) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    const uint64_t n_round = (n + 63) & ~63ull;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_round; r += stride) {
        const bool miss = r < n && best_idx[r] == BC_IDX_MISS;
        const unsigned long long m = __ballot(miss);
        if (m == 0ull) continue;
        const uint32_t lane = threadIdx.x & 63u;
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if ((int)lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        if (miss) {
            const uint64_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            m_words[at] = packed[r];
            m_keys[at] = key0 + r;
            const uint8_t* sq = text + (woff ? (uint64_t)woff[r] : r * (uint64_t)kk);
            char* dst = m_text + at * (uint64_t)kk;
            for (int j = 0; j < kk; ++j) dst[j] = (char)oriented_char(sq, kk, reverse, j);
        }
    }
}

// after the sort by hash: neighbours with equal hashes must hold equal texts (else the grouping would merge two
// different windows: reported, never silently done)
__global__ __launch_bounds__(BLOCK) void k_long_check_groups(const uint64_t* __restrict__ sorted_hash, const uint64_t* __restrict__ ord,
                                                            const char* __restrict__ m_text, int kk, uint64_t n,
                                                            uint32_t* __restrict__ clash) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j == 0 || j >= n || sorted_hash[j] != sorted_hash[j - 1]) return;
    const char *x = m_text + ord[j] * (uint64_t)kk, *y = m_text + ord[j - 1] * (uint64_t)kk;
    for (int c = 0; c < kk; ++c)
        if (x[c] != y[c]) { *clash = 1u; return; }
}

__global__ __launch_bounds__(BLOCK) void k_long_gather(const uint64_t* __restrict__ rep, uint64_t n_u, uint64_t n_m, const uint64_t* __restrict__ m_keys,
                                                      const char* __restrict__ m_text, int kk, uint64_t* __restrict__ first,
                                                      char* __restrict__ out_text) {
    const uint64_t u = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (u >= n_u) return;
    const uint64_t o = rep[u];  // (an ordinal: the reduction's operands behind the data never reach an output)
    if (o >= n_m) return;
    first[u] = m_keys[o];
    for (int c = 0; c < kk; ++c) out_text[u * (uint64_t)kk + c] = m_text[o * (uint64_t)kk + c];
}

__global__ __launch_bounds__(BLOCK) void k_iota64(uint64_t* __restrict__ p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) p[i] = i;
}

// of two miss ordinals the one whose read comes first (keys are distinct: one per read).  rocprim's reduce_by_key scans
// whole tiles: behind the last valid item of the last tile it hands the operator UNINITIALISED values (their results
// are dropped behind a head flag), so an operand that is not an ordinal must not be followed into keys[].
struct EarlierMiss {
    const uint64_t* keys;
    uint64_t n;
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const {
        if (a >= n) return b;
        if (b >= n) return a;
        return keys[a] <= keys[b] ? a : b;
    }
};

// paired-end (heuristicount.py:469-518): a pair counts when both mates have their flanks, neither holds an N and
// seq1 == rev_comp(seq2) (:506) — with both windows packed (the reverse mate's already reverse-complemented) that is
// st1 == st2 == OK and p1 == p2; the match then runs on p1.  Pairs with an ODD window keep string semantics on the host.
__global__ __launch_bounds__(BLOCK) void k_pair_flags(const uint64_t* __restrict__ p1, const uint8_t* __restrict__ st1,
                                                      const uint64_t* __restrict__ p2, const uint8_t* __restrict__ st2,
                                                      uint64_t n, uint8_t* __restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride)
        flags[r] = (st1[r] == BC_WIN_OK && st2[r] == BC_WIN_OK && p1[r] == p2[r]) ? 0 : 1;
}

// ... barcodes of more than 32 nt: p1, p2 are text hashes; equal hashes must hold equal texts
__global__ __launch_bounds__(BLOCK) void k_pair_flags_long(const uint64_t* __restrict__ p1, const uint8_t* __restrict__ st1,
                                                           const uint64_t* __restrict__ p2, const uint8_t* __restrict__ st2,
                                                           const char* __restrict__ t1, const char* __restrict__ t2, int kk,
                                                           uint64_t n, uint8_t* __restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
        bool ok = st1[r] == BC_WIN_OK && st2[r] == BC_WIN_OK && p1[r] == p2[r];
        const char *a = t1 + r * (uint64_t)kk, *b = t2 + r * (uint64_t)kk;
        for (int j = 0; ok && j < kk; ++j) ok = a[j] == b[j];
        flags[r] = ok ? 0 : 1;
    }
}
// the kk-character texts of the listed reads (pairs with an odd window: the packed mate's barcode goes back as text)
__global__ __launch_bounds__(BLOCK) void k_gather_texts(const uint64_t* __restrict__ idx, uint64_t m, const char* __restrict__ t,
                                                        int kk, char* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const char* src = t + idx[j] * (uint64_t)kk;
    char* dst = out + j * (uint64_t)kk;
    for (int c = 0; c < kk; ++c) dst[c] = src[c];
}

__global__ __launch_bounds__(BLOCK) void k_gather_pairs(const uint64_t* __restrict__ idx, uint64_t m,
                                                        const uint64_t* __restrict__ p1, const uint8_t* __restrict__ st1,
                                                        const uint64_t* __restrict__ p2, const uint8_t* __restrict__ st2,
                                                        uint64_t* __restrict__ o1, uint8_t* __restrict__ os1,
                                                        uint64_t* __restrict__ o2, uint8_t* __restrict__ os2) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint64_t r = idx[j];
    o1[j] = p1[r]; os1[j] = st1[r]; o2[j] = p2[r]; os2[j] = st2[r];
}

int grid_for(uint64_t items) {
    uint64_t b = (items + BLOCK - 1) / BLOCK;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, 4096));
}

// newlines in [p, p + n): the one pass the host makes over the text (a chunk's line count orders the chunks)
__attribute__((target("avx2"))) static uint64_t count_nl_avx2(const uint8_t* p, uint64_t n) {
    const __m256i nl = _mm256_set1_epi8('\n');
    uint64_t c = 0, i = 0;
    for (; i + 32 <= n; i += 32)
        c += (uint64_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nl)));
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
static uint64_t count_nl(const uint8_t* p, uint64_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return count_nl_avx2(p, n);
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; ++i) c += p[i] == '\n';
    return c;
}

static bool is_compressed(const char* path) {
    const size_t n = strlen(path);
    return (n >= 3 && strcmp(path + n - 3, ".gz") == 0) || (n >= 4 && strcmp(path + n - 4, ".zst") == 0);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int reserve(size_t bytes) {
        if (bytes <= cap) return BC_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        BC_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return BC_OK;
    }
};

struct Slot {  // one worker's buffers
    uint8_t* h_text = nullptr;   // pinned, 1 + chunk + SLACK
    hipStream_t st = nullptr;
    DevBuf text, bits, nl, tmp, packed, status, best, odd;
    DevBuf mw, mk, mt;           // this chunk's undocumented windows (word, key[, text]) before they join the run-wide list
    DevBuf woff;                 // long barcodes: offset of every window's barcode in the chunk text
    uint64_t* d_n = nullptr;     // expand count
    uint32_t* d_nodd = nullptr;  // [0] odd windows, [1] error flag, [2..3] the chunk's miss cursor (64 bits)
    ~Slot() {
        if (h_text) (void)hipHostFree(h_text);
        if (st) (void)hipStreamDestroy(st);
        if (d_n) (void)hipFree(d_n);
        if (d_nodd) (void)hipFree(d_nodd);
    }
};

struct OddOut { uint64_t key; std::string text; };

// run-wide list of undocumented windows (word, key).  A worker collects a chunk's misses into buffers of its own and
// appends them once its stream has finished the chunk: it RESERVES a range of the list under the short mutex `m`,
// then copies into it on its own stream while holding only the SHARED side of `rw` — eight workers' copies and their
// two stream waits overlap (round 3 held one mutex across both hipStreamSynchronize calls: workers serialised there on
// miss-heavy files).  Growing takes the exclusive side: every reserved range has been copied by then (a copier
// waits for its stream before it lets go of the shared side), so the old buffers can move.  (Round 2 let every
// worker's kernel append to the shared list and grew it under a lock the launches did not take: a worker past its
// capacity check could launch with pointers another one was swapping out.)
struct MissList {
    DevBuf words, keys, texts;
    int text_k = 0;  // > 0: every entry also carries its window's text (barcodes of more than 32 nt)
    uint64_t cap = 0, used = 0;
    std::mutex m;             // used / cap
    std::shared_mutex rw;     // shared: copying into a reserved range; exclusive: growing
    int reserve(uint64_t n) {
        int rc;
        if ((rc = words.reserve(n * 8)) != BC_OK) return rc;
        if ((rc = keys.reserve(n * 8)) != BC_OK) return rc;
        if (text_k && (rc = texts.reserve(n * (uint64_t)text_k)) != BC_OK) return rc;
        cap = n;
        return BC_OK;
    }
    int grow(uint64_t need, hipStream_t st) {  // caller holds the exclusive side of rw
        if (need <= cap) return BC_OK;         // somebody else grew it meanwhile
        const uint64_t ncap = std::max<uint64_t>(cap * 2, need + 65536);
        DevBuf nw, nk, nt;
        int rc;
        if ((rc = nw.reserve(ncap * 8)) != BC_OK) return rc;
        if ((rc = nk.reserve(ncap * 8)) != BC_OK) return rc;
        if (text_k && (rc = nt.reserve(ncap * (uint64_t)text_k)) != BC_OK) return rc;
        if (used) {
            BC_HIP(hipMemcpyAsync(nw.p, words.p, used * 8, hipMemcpyDeviceToDevice, st));
            BC_HIP(hipMemcpyAsync(nk.p, keys.p, used * 8, hipMemcpyDeviceToDevice, st));
            if (text_k) BC_HIP(hipMemcpyAsync(nt.p, texts.p, used * (uint64_t)text_k, hipMemcpyDeviceToDevice, st));
        }
        BC_HIP(hipStreamSynchronize(st));  // before the old buffers are freed (nw / nk / nt leave scope)
        std::swap(nw.p, words.p); std::swap(nw.cap, words.cap);
        std::swap(nk.p, keys.p); std::swap(nk.cap, keys.cap);
        std::swap(nt.p, texts.p); std::swap(nt.cap, texts.cap);
        cap = ncap;
        return BC_OK;
    }
    // append n entries from device buffers; the copies run on `st` and are complete on return
    int append(const void* w, const void* k, const void* t, uint64_t n, hipStream_t st) {
        for (;;) {
            {
                std::shared_lock<std::shared_mutex> copying(rw);
                uint64_t at = ~0ull;
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (used + n <= cap) { at = used; used += n; }
                }
                if (at != ~0ull) {
                    BC_HIP(hipMemcpyAsync((uint64_t*)words.p + at, w, n * 8, hipMemcpyDeviceToDevice, st));
                    BC_HIP(hipMemcpyAsync((uint64_t*)keys.p + at, k, n * 8, hipMemcpyDeviceToDevice, st));
                    if (text_k) BC_HIP(hipMemcpyAsync((char*)texts.p + at * (uint64_t)text_k, t, n * (uint64_t)text_k, hipMemcpyDeviceToDevice, st));
                    BC_HIP(hipStreamSynchronize(st));  // complete before anybody may grow (and move) the list
                    return BC_OK;
                }
            }
            std::unique_lock<std::shared_mutex> growing(rw);  // nobody copies now; `used` is stable under rw + m
            int rc = grow(used + n, st);
            if (rc != BC_OK) return rc;
        }
    }
};

// paired-end: the windows of every read of one file, at the read's index (the pairing and the match run afterwards
// over the whole arrays).  A chunk's kernels write under the shared side of `m` until the chunk's stream has
// finished; growing takes the exclusive side, so nothing is in flight while the arrays move.
struct WindowStore {
    DevBuf packed, status;   // uint64[cap], uint8[cap]
    DevBuf texts;            // char[cap][text_k]: barcodes of more than 32 nt, as the library spells them (text_k > 0)
    int text_k = 0;
    uint64_t cap = 0, estimate = 0;
    std::shared_mutex m;
    int grow(uint64_t need) {  // caller holds the exclusive lock
        const uint64_t ncap = std::max<uint64_t>(std::max<uint64_t>(cap * 2, need + (estimate && estimate < need ? 16 : 65536)), estimate);
        DevBuf np_, ns, nt;
        int rc;
        if ((rc = np_.reserve(ncap * 8)) != BC_OK) return rc;
        if ((rc = ns.reserve(ncap)) != BC_OK) return rc;
        if (text_k && (rc = nt.reserve(ncap * (uint64_t)text_k)) != BC_OK) return rc;
        if (cap) {
            BC_HIP(hipMemcpy(np_.p, packed.p, cap * 8, hipMemcpyDeviceToDevice));
            BC_HIP(hipMemcpy(ns.p, status.p, cap, hipMemcpyDeviceToDevice));
            if (text_k) BC_HIP(hipMemcpy(nt.p, texts.p, cap * (uint64_t)text_k, hipMemcpyDeviceToDevice));
        }
        BC_HIP(hipDeviceSynchronize());
        std::swap(np_.p, packed.p); std::swap(np_.cap, packed.cap);
        std::swap(ns.p, status.p); std::swap(ns.cap, status.cap);
        std::swap(nt.p, texts.p); std::swap(nt.cap, texts.cap);
        cap = ncap;
        return BC_OK;
    }
};

struct FileJob {
    bc_lib* lib = nullptr;
    const char* path = nullptr;
    uint64_t byte_begin = 0, byte_end = 0;
    Geometry g{};
    int max_mismatch = 0, algo = 0, n_threads = 0;
    uint64_t key_base = 0;
    // chunk c of the stream is this job's when c % n_parts == part (several ranks over one COMPRESSED file: every rank
    // decodes the stream, the per-read work is split; byte ranges cannot cut a compressed stream)
    uint32_t part = 0, n_parts = 1;
    // match mode (store == nullptr): per-chunk match, tallies accumulate on the device, misses join `misses`
    uint32_t* d_counts = nullptr;
    uint64_t* d_first = nullptr;
    MissList* misses = nullptr;
    // store mode: windows only
    WindowStore* store = nullptr;
    // results
    std::vector<OddOut> odd_all;
    uint64_t n_reads = 0;
};

// every chunk of one file through the device (see the header of this file); returns BC_OK or the first error
int run_file(FileJob& J) {
    uint32_t G = 0; int k = 0, device = 0;
    bc_lib_info(J.lib, &G, &k, &device);
    const char* path = J.path;
    Geometry g = J.g;
    // A compressed file (.gz / .zst) is a sequential stream: whichever worker is next decodes the following 16 MiB
    // (+ slack) into its own page-locked buffer under `src_m` — the decoder runs on one thread at a time, uploads and
    // kernels of the chunks before it run meanwhile on the others' streams — and the stream's length is known at its end
    // only.  A plain file is read with pread by every worker for itself, as before.
    const bool compressed = is_compressed(path);
    bc_reader* rd = nullptr;
    struct ReaderCloser { bc_reader*& r; ~ReaderCloser() { if (r) bc_reader_close(r); } } rd_closer{rd};
    int fd = -1;
    struct Closer { int& fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fd};
    uint64_t fsize_plain = 0;
    if (compressed) {
        if (J.byte_begin != 0 || J.byte_end < (1ull << 62)) {
            set_error("%s is compressed: byte ranges need a plain .fastq / .reads file (several ranks split it by chunk: part / n_parts)", path);
            return BC_EINVAL;
        }
        int rc0 = bc_reader_open(path, &rd);
        if (rc0 != BC_OK) return rc0;
    } else {
        fd = open(path, O_RDONLY);
        if (fd < 0) { set_error("cannot open %s", path); return BC_EINVAL; }
        struct stat sb;
        if (fstat(fd, &sb) != 0) { set_error("cannot stat %s", path); return BC_EINVAL; }
        fsize_plain = (uint64_t)sb.st_size;
    }
    const uint64_t byte_end = compressed ? ~0ull : std::min(J.byte_end, fsize_plain);
    const uint64_t byte_begin = compressed ? 0 : std::min(J.byte_begin, byte_end);
    const uint64_t span = compressed ? ~0ull : byte_end - byte_begin;
    int T = J.n_threads;
    if (T <= 0) {  // default; BC_STREAM_THREADS overrides (tuning aid)
        const char* e = getenv("BC_STREAM_THREADS");
        T = e && atoi(e) > 0 ? atoi(e) : 8;  // measured, 4.6 GB file in the page cache: 4 / 8 / 12 / 16 threads -> 144-161 / 210 / 243 / 222 Mreads/s
    }
    // Chunk size: 32 MiB for a long file, 16 MiB for a short one (under 1 GiB with eight workers: every worker gets several)
    // and for a compressed stream (length unknown, pace the decoder's).  Measured on a 4.6 GB file in the page cache, eight
    // workers (profiles/r04_stream_chunk_ab.txt): 16 / 32 / 64 MiB -> 260 / 283 / 202-213 Mreads/s in the bulk pass.  The
    // workers' time is their pread() into the page-locked buffer (1.2-2.2 s summed over the threads against 0.3-0.5 s of
    // launches): fewer, longer reads help up to 32 MiB; beyond it the last wave of chunks leaves workers idle.
    // BC_STREAM_CHUNK_MB overrides (tests, tuning).
    uint64_t CHUNK = CHUNK_MIN;
    if (!compressed) CHUNK = std::min<uint64_t>(CHUNK_MAX, std::max<uint64_t>(CHUNK_MIN, (span / (4ull * (uint64_t)T)) / CHUNK_MIN * CHUNK_MIN));
    if (const char* e = getenv("BC_STREAM_CHUNK_MB"))
        if (atoi(e) >= 1 && atoi(e) <= 1024) CHUNK = (uint64_t)atoi(e) << 20;
    const uint64_t n_chunks = compressed ? ~0ull : (span + CHUNK - 1) / CHUNK;
    std::mutex src_m;                       // compressed: the decoder, `src_next`, `src_done`, `carry`
    uint64_t src_next = 0;
    bool src_done = false;
    std::vector<uint8_t> carry;             // compressed: decoded bytes from one before the next chunk's core on
    T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)T, std::max<uint64_t>(n_chunks, 1)));
    if (compressed) T = std::min(T, 4);  // one decodes, the others upload / launch / collect
    const uint64_t kmask = kmer_mask(k);
    WindowStore* const store = J.store;

    std::vector<Slot> slots((size_t)T);
    // ordered hand-over of the running line / read totals (a chunk needs the number of lines in front of it)
    std::mutex order_m;
    std::condition_variable order_cv;
    uint64_t next_to_publish = 0, lines_before = 0, reads_launched = 0;
    std::atomic<uint64_t> next_chunk{0};
    std::atomic<int> first_rc{BC_OK};
    std::string err;
    std::mutex err_m, odd_m;

    auto fail = [&](int code) {
        int expect = BC_OK;
        if (first_rc.compare_exchange_strong(expect, code)) {
            std::lock_guard<std::mutex> lk(err_m);
            err = bc_last_error();
        }
        order_cv.notify_all();
    };

    std::atomic<uint64_t> prof_us[5];
    for (auto& x : prof_us) x = 0;
    auto work = [&](int ti) {
        Slot& s = slots[(size_t)ti];
        auto body = [&]() -> int {
            BC_HIP(hipSetDevice(device));
            BC_HIP(hipHostMalloc((void**)&s.h_text, 1 + CHUNK + SLACK + 32, hipHostMallocDefault));
            BC_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
            BC_HIP(hipMalloc((void**)&s.d_n, 16));
            BC_HIP(hipMalloc((void**)&s.d_nodd, 16));
            int rc2;
            if ((rc2 = s.text.reserve(1 + CHUNK + SLACK + 64)) != BC_OK) return rc2;
            auto tick = [] { return std::chrono::steady_clock::now(); };
            auto since = [](std::chrono::steady_clock::time_point a) {
                return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - a).count();
            };
            for (;;) {
                uint64_t c, core_lo, core_hi, base, hi, nbytes, fsize;
                int virt;
                auto t_a = tick();
                if (!compressed) {
                    c = next_chunk.fetch_add(1);
                    if (c >= n_chunks || first_rc.load() != BC_OK) break;
                    // ---- read: bytes [base, hi) of the file; the chunk owns newlines in [own_lo, own_hi)
                    fsize = fsize_plain;
                    core_lo = byte_begin + c * CHUNK; core_hi = std::min(byte_end, core_lo + CHUNK);
                    virt = (c == 0) ? 1 : 0;                 // the line at the start of the range
                    base = virt ? core_lo : core_lo - 1;
                    hi = std::min(fsize, core_hi + SLACK);
                    nbytes = hi - base;
                    uint64_t got = 0;
                    while (got < nbytes) {
                        ssize_t n = pread(fd, s.h_text + got, nbytes - got, (off_t)(base + got));
                        if (n <= 0) { set_error("read error in %s", path); return BC_EINVAL; }
                        got += (uint64_t)n;
                    }
                } else {
                    // ---- decode: the same window [core_lo - 1, core_hi + SLACK) of the DECODED stream; what lies past
                    // this chunk's core (one byte of context + the slack) is kept for the next chunk
                    std::lock_guard<std::mutex> lk(src_m);
                    if (src_done || first_rc.load() != BC_OK) break;
                    c = src_next++;
                    virt = (c == 0) ? 1 : 0;
                    core_lo = c * CHUNK;
                    base = virt ? core_lo : core_lo - 1;
                    const uint64_t left = virt ? 0 : 1;
                    const uint64_t want = left + CHUNK + SLACK;
                    uint64_t have = carry.size();
                    if (have) memcpy(s.h_text, carry.data(), have);
                    bool at_end = false;
                    while (have < want) {
                        uint64_t got = 0;
                        int rr = bc_reader_read_raw(rd, (char*)s.h_text + have, want - have, &got);
                        if (rr != BC_OK) return rr;
                        if (got == 0) { at_end = true; break; }
                        have += got;
                    }
                    nbytes = have;
                    hi = base + have;
                    if (have <= left + CHUNK) {  // the stream ends inside this chunk's core: the last chunk
                        core_hi = base + have;
                        src_done = true;
                        carry.clear();
                    } else {
                        core_hi = core_lo + CHUNK;
                        carry.assign(s.h_text + left + CHUNK - 1, s.h_text + have);
                    }
                    fsize = at_end ? hi : ~0ull;  // (compared with `hi` below: does the stream go on behind the buffer?)
                    if (have == 0) break;         // an empty stream (a later chunk always starts with the kept bytes)
                }
                // owned newlines: file positions [core_lo - 1, core_hi - 1) = buffer offsets [0, own_hi); the first chunk
                // starts at the range's first line instead (a virtual newline in front of it).  The newline at
                // core_hi - 1 belongs to the next chunk — or to the next range, or ends the file.
                prof_us[0] += since(t_a);
                t_a = tick();
                const uint64_t own_hi = core_hi - 1 - base;
                const uint64_t n_real = count_nl(s.h_text, own_hi);
                prof_us[1] += since(t_a);
                t_a = tick();
                const uint64_t n_own = n_real + (uint64_t)virt;
                // ---- position in the run: lines and reads in front of this chunk
                uint64_t line_base;
                {
                    std::unique_lock<std::mutex> lk(order_m);
                    order_cv.wait(lk, [&] { return next_to_publish == c || first_rc.load() != BC_OK; });
                    if (first_rc.load() != BC_OK) return BC_OK;
                    line_base = lines_before;
                    lines_before += n_own;
                    const uint64_t gri0_ = g.fastq ? ((line_base + 2) >> 2) : line_base;
                    const uint64_t gri1_ = g.fastq ? ((line_base + n_own + 2) >> 2) : line_base + n_own;
                    if (c % J.n_parts == J.part) reads_launched += gri1_ - gri0_;
                    if (store && c == 0) {  // size the run-wide arrays from the first chunk's density (+25 %)
                        const char* e = getenv("BC_STREAM_EST_PCT");  // (tests shrink the estimate to exercise the growth path)
                        const double pct = e && atoi(e) > 0 ? atoi(e) / 100.0 : 1.25;
                        // (a compressed stream's length is not known: eight chunks' worth, doubled as it fills)
                        const double chunks = compressed ? 8.0 : (double)span / (double)std::max<uint64_t>(core_hi - core_lo, 1);
                        store->estimate = (uint64_t)((double)(gri1_ - gri0_) * chunks * pct) + (pct < 1.0 ? 16 : 65536);
                    }
                    ++next_to_publish;
                }
                order_cv.notify_all();
                prof_us[2] += since(t_a);
                t_a = tick();
                const uint64_t gri0 = g.fastq ? ((line_base + 2) >> 2) : line_base;
                const uint64_t gri1 = g.fastq ? ((line_base + n_own + 2) >> 2) : line_base + n_own;
                const uint64_t n_reads = gri1 - gri0;
                if (n_own == 0 || n_reads == 0 || c % J.n_parts != J.part) continue;  // (another part's chunk: its lines are counted, no more)
                // ---- device pass
                const uint64_t n_words = (nbytes + 31) / 32;
                const uint64_t nl_cap = n_real + SLACK + 64;
                const uint32_t odd_cap = 1u << 16;
                if ((rc2 = s.bits.reserve(n_words * 4 + 64)) != BC_OK) return rc2;
                if ((rc2 = s.nl.reserve(nl_cap * 4)) != BC_OK) return rc2;
                if ((rc2 = s.tmp.reserve((n_words / 1024 + 2) * 4)) != BC_OK) return rc2;
                if ((rc2 = s.odd.reserve((size_t)odd_cap * sizeof(OddRec))) != BC_OK) return rc2;
                std::shared_lock<std::shared_mutex> window_lock;
                // leaving this chunk early (an error below) with kernels in flight: wait for them BEFORE the window lock
                // goes (declared after it, destroyed before it) — another worker may grow, i.e. move, the arrays they write
                struct DrainOnExit {
                    hipStream_t st; bool armed = true;
                    ~DrainOnExit() { if (armed) (void)hipStreamSynchronize(st); }
                } drain{s.st};
                uint64_t* d_packed;
                uint8_t* d_status;
                char* d_otext = nullptr;
                if (store) {  // the windows go to the run-wide arrays, at the reads' own indices
                    window_lock = std::shared_lock<std::shared_mutex>(store->m);
                    if (gri1 > store->cap) {
                        window_lock.unlock();
                        {
                            std::unique_lock<std::shared_mutex> ul(store->m);
                            if (gri1 > store->cap && (rc2 = store->grow(gri1)) != BC_OK) return rc2;
                        }
                        window_lock.lock();
                    }
                    d_packed = (uint64_t*)store->packed.p + gri0;
                    d_status = (uint8_t*)store->status.p + gri0;
                    if (store->text_k) d_otext = (char*)store->texts.p + gri0 * (uint64_t)store->text_k;
                } else {
                    if ((rc2 = s.packed.reserve(n_reads * 8 + 64)) != BC_OK) return rc2;
                    if ((rc2 = s.status.reserve(n_reads + 64)) != BC_OK) return rc2;
                    if ((rc2 = s.best.reserve(n_reads * 4 + 64)) != BC_OK) return rc2;
                    if ((rc2 = s.mw.reserve(n_reads * 8 + 64)) != BC_OK) return rc2;  // every read could be a miss
                    if ((rc2 = s.mk.reserve(n_reads * 8 + 64)) != BC_OK) return rc2;
                    if (g.text_k && (rc2 = s.mt.reserve(n_reads * (uint64_t)g.text_k + 64)) != BC_OK) return rc2;
                    d_packed = (uint64_t*)s.packed.p;
                    d_status = (uint8_t*)s.status.p;
                }
                if (g.text_k && (rc2 = s.woff.reserve(n_reads * 4 + 64)) != BC_OK) return rc2;
                BC_HIP(hipMemcpyAsync(s.text.p, s.h_text, nbytes, hipMemcpyHostToDevice, s.st));
                BC_HIP(hipMemsetAsync(s.d_nodd, 0, 16, s.st));
                hipLaunchKernelGGL(k_nl_bitmap, dim3(grid_for(n_words)), dim3(BLOCK), 0, s.st, (const uint8_t*)s.text.p, nbytes,
                                   (uint32_t*)s.bits.p, n_words);
                if ((rc2 = bc_sites_expand_dev((const uint32_t*)s.bits.p, n_words, (uint32_t*)s.nl.p, nl_cap, s.d_n,
                                               (uint32_t*)s.tmp.p, s.st)) != BC_OK) return rc2;
                // the device's list holds every newline of the buffer in ascending order: the owned ones come first,
                // the slack's after them (the end of the last owned line is among those)
                const uint64_t key0 = J.key_base + gri0;
                hipLaunchKernelGGL(k_lines_to_windows, dim3(grid_for(n_own)), dim3(BLOCK), 0, s.st, (const uint8_t*)s.text.p,
                                   nbytes, (const uint32_t*)s.nl.p, (const uint64_t*)s.d_n, nl_cap, n_own, virt,
                                   hi < fsize ? 1 : 0, line_base, gri0, g, key0, d_packed, d_status, (OddRec*)s.odd.p,
                                   odd_cap, s.d_nodd, (uint32_t*)s.woff.p, d_otext);
                BC_HIP(hipGetLastError());
                if (!store && g.text_k) {
                    // barcodes of more than 32 nt: the exact table on the text hashes (no tally: a hit is a hint), the
                    // texts decide, the verified hits are tallied, the misses keep their texts
                    if ((rc2 = bc_match_dev(J.lib, d_packed, d_status, n_reads, 0, BC_ALGO_HASH, (int32_t*)s.best.p, nullptr,
                                            nullptr, nullptr, nullptr, key0, s.st)) != BC_OK) return rc2;
                    hipLaunchKernelGGL(k_long_verify_tally, dim3(grid_for(n_reads)), dim3(BLOCK), 0, s.st, (int32_t*)s.best.p,
                                       (const uint8_t*)s.text.p, (const uint32_t*)s.woff.p, J.lib->d_text, g.text_k, g.reverse,
                                       n_reads, key0, J.d_counts, J.d_first);
                    hipLaunchKernelGGL(k_collect_misses_long, dim3(grid_for(n_reads)), dim3(BLOCK), 0, s.st,
                                       (const int32_t*)s.best.p, (const uint64_t*)d_packed, (const uint8_t*)s.text.p,
                                       (const uint32_t*)s.woff.p, g.text_k, g.reverse, n_reads, key0, (uint64_t*)s.mw.p,
                                       (uint64_t*)s.mk.p, (char*)s.mt.p, (unsigned long long*)(s.d_nodd + 2));
                    BC_HIP(hipGetLastError());
                } else if (!store) {
                    if ((rc2 = bc_match_dev(J.lib, d_packed, d_status, n_reads, J.max_mismatch, J.algo, (int32_t*)s.best.p,
                                            nullptr, nullptr, J.d_counts, J.d_first, key0, s.st)) != BC_OK) return rc2;
                    hipLaunchKernelGGL(k_collect_misses, dim3(grid_for(n_reads)), dim3(BLOCK), 0, s.st, (const int32_t*)s.best.p,
                                       (const uint64_t*)d_packed, n_reads, kmask, key0, (uint64_t*)s.mw.p,
                                       (uint64_t*)s.mk.p, (unsigned long long*)(s.d_nodd + 2));
                    BC_HIP(hipGetLastError());
                }
                // ---- odd windows: their text is still in the pinned buffer
                uint32_t fl[4] = {0, 0, 0, 0};
                BC_HIP(hipMemcpyAsync(fl, s.d_nodd, 16, hipMemcpyDeviceToHost, s.st));
                prof_us[3] += since(t_a);
                t_a = tick();
                BC_HIP(hipStreamSynchronize(s.st));
                drain.armed = false;
                if (window_lock.owns_lock()) window_lock.unlock();  // this chunk's writes into the run-wide arrays are done
                prof_us[4] += since(t_a);
                if (fl[1] == 2u) {  // k_lines_to_windows: the newline list was cut short — a sizing bug here, not bad input
                    set_error("%s: internal error: newline list overflow (%llu entries)", path, (unsigned long long)nl_cap);
                    return BC_ECAP;
                }
                if (fl[1]) { set_error("%s: a line longer than %llu bytes", path, (unsigned long long)SLACK); return BC_EINVAL; }
                const uint64_t n_m = (uint64_t)fl[2] | ((uint64_t)fl[3] << 32);
                if (n_m && (rc2 = J.misses->append(s.mw.p, s.mk.p, s.mt.p, n_m, s.st)) != BC_OK) return rc2;
                const uint32_t n_odd = fl[0];
                if (n_odd > odd_cap) { set_error("%s: more than %u odd windows in one chunk", path, odd_cap); return BC_EINVAL; }
                if (n_odd) {
                    std::vector<OddRec> recs(n_odd);
                    BC_HIP(hipMemcpy(recs.data(), s.odd.p, (size_t)n_odd * sizeof(OddRec), hipMemcpyDeviceToHost));
                    std::lock_guard<std::mutex> lk(odd_m);
                    for (const OddRec& o : recs)
                        J.odd_all.push_back(OddOut{o.key, std::string((const char*)s.h_text + o.begin, o.len)});
                }
            }
            return BC_OK;
        };
        int code = body();
        // On EVERY exit — the normal one, an error of this worker, or `return BC_OK` because another worker failed — the
        // stream dies with the slot, so its workspace on the library (tally records, cascade buffers: keyed by the
        // stream handle) goes first; a later stream that reuses the handle value must not inherit it.
        if (s.st) {
            const hipError_t e = hipStreamSynchronize(s.st);
            const int rr = bc_lib_release_stream(J.lib, s.st);
            if (code == BC_OK && e != hipSuccess) { set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); code = BC_EHIP; }
            if (code == BC_OK) code = rr;
        }
        if (code != BC_OK) fail(code);
    };
    const auto t_workers = std::chrono::steady_clock::now();
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int ti = 0; ti < T; ++ti) th.emplace_back(work, ti);
        for (auto& t : th) t.join();
    }
    if (getenv("BC_STREAM_PROF"))
        fprintf(stderr, "stream: %d threads, chunks of %llu MiB, workers %.3f s; per-thread-summed ms: pread %.0f count %.0f order %.0f launch %.0f sync %.0f\n", T, (unsigned long long)(CHUNK >> 20),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_workers).count(), prof_us[0] / 1e3,
                prof_us[1] / 1e3, prof_us[2] / 1e3, prof_us[3] / 1e3, prof_us[4] / 1e3);
    if (first_rc.load() != BC_OK) {
        set_error("%s", err.c_str());
        return first_rc.load();
    }
    J.n_reads = reads_launched;
    std::sort(J.odd_all.begin(), J.odd_all.end(), [](const OddOut& a, const OddOut& b) { return a.key < b.key; });
    return BC_OK;
}

int file_kind(const char* path, const char* who, int* fastq) {
    std::string p(path);
    auto ends = [&](const char* suf) { size_t n = strlen(suf); return p.size() >= n && p.compare(p.size() - n, n, suf) == 0; };
    if (ends(".gz")) p.resize(p.size() - 3);       // heuristicount.py:104-116: the type is read behind an optional .gz / .zst
    else if (ends(".zst")) p.resize(p.size() - 4);
    if (ends(".fastq")) *fastq = 1;
    else if (ends(".reads")) *fastq = 0;
    else { set_error("%s: a .fastq or .reads file (optionally .gz / .zst) is needed, got %s", who, path); return BC_EINVAL; }
    return BC_OK;
}

}  // namespace

struct bc_stream_result {
    uint64_t n_reads = 0;
    double seconds = 0;
    // the table of undocumented windows stays on the device until the caller hands over its arrays
    // (bc_stream_result_undoc copies straight into them: 24 bytes x 2.5e7 entries on the benchmark file)
    int device = 0;
    uint64_t n_undoc = 0;
    uint64_t *d_words = nullptr, *d_counts = nullptr, *d_first = nullptr;
    int text_k = 0;            // > 0: the undocumented windows are texts of that length (d_utext), sorted by their hash
    char* d_utext = nullptr;
    ~bc_stream_result() {
        if (d_words || d_counts || d_first || d_utext) (void)hipSetDevice(device);
        for (void* q : {(void*)d_words, (void*)d_counts, (void*)d_first, (void*)d_utext}) if (q) (void)hipFree(q);
    }
    std::vector<uint64_t> odd_keys, odd_off;  // odd_off: n_odd + 1 offsets into odd_text
    std::string odd_text;
    // paired runs: the pairs with an odd window (see bc_stream_result_pair_odd); odd_keys / odd_text stay empty
    std::vector<uint64_t> pair_keys, pair_p1, pair_p2, pair_off1, pair_off2;
    std::vector<uint8_t> pair_st1, pair_st2;
    std::string pair_text1, pair_text2;
};

namespace {

// undocumented windows: sort the (word, key) pairs by word, then one pass gives every distinct word its count and
// smallest key; the table stays on the device (ownership moves to the result)
int group_misses(MissList& M, int device, bc_stream_result* res) {
    const unsigned long long n_miss = M.used;
    if (!n_miss) return BC_OK;
    DevBuf w2, k2, tmp, uw, uf, uc, un;
    int rc3;
    if ((rc3 = w2.reserve(n_miss * 8)) != BC_OK) return rc3;
    if ((rc3 = k2.reserve(n_miss * 8)) != BC_OK) return rc3;
    if ((rc3 = uw.reserve(n_miss * 8)) != BC_OK) return rc3;
    if ((rc3 = uf.reserve(n_miss * 8)) != BC_OK) return rc3;
    if ((rc3 = uc.reserve(n_miss * 8)) != BC_OK) return rc3;
    if ((rc3 = un.reserve(16)) != BC_OK) return rc3;
    size_t tb = 0;
    BC_HIP(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)M.words.p, (uint64_t*)w2.p, (uint64_t*)M.keys.p,
                                     (uint64_t*)k2.p, (size_t)n_miss, 0, 64, 0));
    if ((rc3 = tmp.reserve(tb)) != BC_OK) return rc3;
    BC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t*)M.words.p, (uint64_t*)w2.p, (uint64_t*)M.keys.p,
                                     (uint64_t*)k2.p, (size_t)n_miss, 0, 64, 0));
    // every run of equal words: its smallest key, and its length (a sum of ones)
    size_t tb2 = 0;
    BC_HIP(rocprim::reduce_by_key(nullptr, tb2, (uint64_t*)w2.p, (uint64_t*)k2.p, (size_t)n_miss, (uint64_t*)uw.p,
                                  (uint64_t*)uf.p, (uint64_t*)un.p, rocprim::minimum<uint64_t>(),
                                  rocprim::equal_to<uint64_t>(), 0));
    if ((rc3 = tmp.reserve(tb2)) != BC_OK) return rc3;
    BC_HIP(rocprim::reduce_by_key(tmp.p, tb2, (uint64_t*)w2.p, (uint64_t*)k2.p, (size_t)n_miss, (uint64_t*)uw.p,
                                  (uint64_t*)uf.p, (uint64_t*)un.p, rocprim::minimum<uint64_t>(),
                                  rocprim::equal_to<uint64_t>(), 0));
    size_t tb4 = 0;
    auto ones_it = rocprim::make_constant_iterator<uint64_t>(1);
    BC_HIP(rocprim::reduce_by_key(nullptr, tb4, (uint64_t*)w2.p, ones_it, (size_t)n_miss, (uint64_t*)uw.p,
                                  (uint64_t*)uc.p, (uint64_t*)un.p, rocprim::plus<uint64_t>(),
                                  rocprim::equal_to<uint64_t>(), 0));
    if ((rc3 = tmp.reserve(tb4)) != BC_OK) return rc3;
    BC_HIP(rocprim::reduce_by_key(tmp.p, tb4, (uint64_t*)w2.p, ones_it, (size_t)n_miss, (uint64_t*)uw.p,
                                  (uint64_t*)uc.p, (uint64_t*)un.p, rocprim::plus<uint64_t>(),
                                  rocprim::equal_to<uint64_t>(), 0));
    BC_HIP(hipDeviceSynchronize());
    uint64_t n_u = 0;
    BC_HIP(hipMemcpy(&n_u, un.p, 8, hipMemcpyDeviceToHost));
    res->device = device;
    res->n_undoc = n_u;
    res->d_words = (uint64_t*)uw.p; res->d_first = (uint64_t*)uf.p; res->d_counts = (uint64_t*)uc.p;
    uw.p = nullptr; uf.p = nullptr; uc.p = nullptr;  // ownership moves to the result
    return BC_OK;
}

// the same for windows of more than 32 nt: the sort key is the text hash, every group's representative (its earliest
// read) lends the text, and equal hashes are checked to hold equal texts
int group_misses_long(MissList& M, int device, bc_stream_result* res) {
    const unsigned long long n_miss = M.used;
    res->text_k = M.text_k;
    if (!n_miss) return BC_OK;
    const unsigned nb = (unsigned)((n_miss + BLOCK - 1) / BLOCK);
    DevBuf ord, h2, o2, tmp, uh, rep, uc, un, clash;
    int rc3;
    if ((rc3 = ord.reserve(n_miss * 8)) != BC_OK || (rc3 = h2.reserve(n_miss * 8)) != BC_OK || (rc3 = o2.reserve(n_miss * 8)) != BC_OK ||
        (rc3 = uh.reserve(n_miss * 8)) != BC_OK || (rc3 = rep.reserve(n_miss * 8)) != BC_OK || (rc3 = uc.reserve(n_miss * 8)) != BC_OK ||
        (rc3 = un.reserve(16)) != BC_OK || (rc3 = clash.reserve(16)) != BC_OK) return rc3;
    hipLaunchKernelGGL(k_iota64, dim3(nb), dim3(BLOCK), 0, 0, (uint64_t*)ord.p, (uint64_t)n_miss);
    BC_HIP(hipGetLastError());
    BC_HIP(hipMemset(clash.p, 0, 16));
    size_t tb = 0;
    BC_HIP(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)M.words.p, (uint64_t*)h2.p, (uint64_t*)ord.p, (uint64_t*)o2.p,
                                     (size_t)n_miss, 0, 64, 0));
    if ((rc3 = tmp.reserve(tb)) != BC_OK) return rc3;
    BC_HIP(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t*)M.words.p, (uint64_t*)h2.p, (uint64_t*)ord.p, (uint64_t*)o2.p,
                                     (size_t)n_miss, 0, 64, 0));
    hipLaunchKernelGGL(k_long_check_groups, dim3(nb), dim3(BLOCK), 0, 0, (const uint64_t*)h2.p, (const uint64_t*)o2.p,
                       (const char*)M.texts.p, M.text_k, (uint64_t)n_miss, (uint32_t*)clash.p);
    BC_HIP(hipGetLastError());
    const EarlierMiss earlier{(const uint64_t*)M.keys.p, (uint64_t)n_miss};
    size_t tb2 = 0;
    BC_HIP(rocprim::reduce_by_key(nullptr, tb2, (uint64_t*)h2.p, (uint64_t*)o2.p, (size_t)n_miss, (uint64_t*)uh.p, (uint64_t*)rep.p,
                                  (uint64_t*)un.p, earlier, rocprim::equal_to<uint64_t>(), 0));
    if ((rc3 = tmp.reserve(tb2)) != BC_OK) return rc3;
    BC_HIP(rocprim::reduce_by_key(tmp.p, tb2, (uint64_t*)h2.p, (uint64_t*)o2.p, (size_t)n_miss, (uint64_t*)uh.p, (uint64_t*)rep.p,
                                  (uint64_t*)un.p, earlier, rocprim::equal_to<uint64_t>(), 0));
    size_t tb4 = 0;
    auto ones_it = rocprim::make_constant_iterator<uint64_t>(1);
    BC_HIP(rocprim::reduce_by_key(nullptr, tb4, (uint64_t*)h2.p, ones_it, (size_t)n_miss, (uint64_t*)uh.p, (uint64_t*)uc.p,
                                  (uint64_t*)un.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), 0));
    if ((rc3 = tmp.reserve(tb4)) != BC_OK) return rc3;
    BC_HIP(rocprim::reduce_by_key(tmp.p, tb4, (uint64_t*)h2.p, ones_it, (size_t)n_miss, (uint64_t*)uh.p, (uint64_t*)uc.p,
                                  (uint64_t*)un.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), 0));
    BC_HIP(hipDeviceSynchronize());
    uint64_t n_u = 0;
    uint32_t bad = 0;
    BC_HIP(hipMemcpy(&n_u, un.p, 8, hipMemcpyDeviceToHost));
    BC_HIP(hipMemcpy(&bad, clash.p, 4, hipMemcpyDeviceToHost));
    if (bad) { set_error("two different undocumented windows share a 64-bit text hash: take the host path (BC_DEVICE_INGEST=0)"); return BC_ECAP; }
    DevBuf uf, ut;
    if ((rc3 = uf.reserve(n_u * 8 + 16)) != BC_OK || (rc3 = ut.reserve(n_u * (uint64_t)M.text_k + 16)) != BC_OK) return rc3;
    hipLaunchKernelGGL(k_long_gather, dim3((unsigned)((n_u + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, (const uint64_t*)rep.p, n_u,
                       (uint64_t)n_miss, (const uint64_t*)M.keys.p, (const char*)M.texts.p, M.text_k, (uint64_t*)uf.p, (char*)ut.p);
    BC_HIP(hipGetLastError());
    BC_HIP(hipDeviceSynchronize());
    res->device = device;
    res->n_undoc = n_u;
    res->d_words = (uint64_t*)uh.p; res->d_first = (uint64_t*)uf.p; res->d_counts = (uint64_t*)uc.p; res->d_utext = (char*)ut.p;
    uh.p = nullptr; uf.p = nullptr; uc.p = nullptr; ut.p = nullptr;  // ownership moves to the result
    return BC_OK;
}

int tallies_to_host(uint32_t G, const DevBuf& d_counts, const DevBuf& d_first, uint64_t* h_counts, uint64_t* h_first_seen) {
    if (h_counts) {
        std::vector<uint32_t> c32(G);
        BC_HIP(hipMemcpy(c32.data(), d_counts.p, (size_t)G * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < G; ++i) h_counts[i] += c32[i];
    }
    if (h_first_seen) {
        std::vector<uint64_t> fs(G);
        BC_HIP(hipMemcpy(fs.data(), d_first.p, (size_t)G * 8, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < G; ++i) h_first_seen[i] = std::min(h_first_seen[i], fs[i]);
    }
    return BC_OK;
}

int fill_geometry(Geometry& g, int k, int64_t start, const char* L, int L_len, const char* R, int R_len, int reverse,
                  const char* who) {
    if (start < 0 || L_len < 0 || R_len < 0 || L_len > 15 || R_len > 15 || (L_len && !L) || (R_len && !R)) {
        set_error("%s: bad argument (flanks are at most 15 characters)", who);
        return BC_EINVAL;
    }
    g.start = start; g.L_len = L_len; g.R_len = R_len; g.k = k; g.reverse = reverse ? 1 : 0;
    memcpy(g.L, L ? L : "", (size_t)L_len);
    memcpy(g.R, R ? R : "", (size_t)R_len);
    return BC_OK;
}

}  // namespace

extern "C" {

int bc_stream_result_free(bc_stream_result* r) {
    delete r;
    return BC_OK;
}

int bc_stream_result_info(const bc_stream_result* r, uint64_t* n_reads, uint64_t* n_undoc, uint64_t* n_odd,
                          uint64_t* odd_bytes, double* seconds) {
    if (!r) { set_error("bc_stream_result_info: null"); return BC_EINVAL; }
    if (n_reads) *n_reads = r->n_reads;
    if (n_undoc) *n_undoc = r->n_undoc;
    if (n_odd) *n_odd = r->odd_keys.size();
    if (odd_bytes) *odd_bytes = r->odd_text.size();
    if (seconds) *seconds = r->seconds;
    return BC_OK;
}

int bc_stream_result_undoc(const bc_stream_result* r, uint64_t* words, uint64_t* counts, uint64_t* first) {
    if (!r || ((!words || !counts || !first) && r->n_undoc)) { set_error("bc_stream_result_undoc: null"); return BC_EINVAL; }
    if (!r->n_undoc) return BC_OK;
    BC_HIP(hipSetDevice(r->device));
    BC_HIP(hipMemcpy(words, r->d_words, r->n_undoc * 8, hipMemcpyDeviceToHost));
    BC_HIP(hipMemcpy(counts, r->d_counts, r->n_undoc * 8, hipMemcpyDeviceToHost));
    BC_HIP(hipMemcpy(first, r->d_first, r->n_undoc * 8, hipMemcpyDeviceToHost));
    return BC_OK;
}

int bc_stream_result_odd(const bc_stream_result* r, uint64_t* keys, uint64_t* off, char* text) {
    if (!r) { set_error("bc_stream_result_odd: null"); return BC_EINVAL; }
    std::copy(r->odd_keys.begin(), r->odd_keys.end(), keys);
    std::copy(r->odd_off.begin(), r->odd_off.end(), off);
    std::copy(r->odd_text.begin(), r->odd_text.end(), text);
    return BC_OK;
}

int bc_stream_result_text_info(const bc_stream_result* r, int* text_k) {
    if (!r || !text_k) { set_error("bc_stream_result_text_info: null"); return BC_EINVAL; }
    *text_k = r->text_k;
    return BC_OK;
}

int bc_stream_result_undoc_text(const bc_stream_result* r, char* text, uint64_t* counts, uint64_t* first) {
    if (!r || !r->text_k || ((!text || !counts || !first) && r->n_undoc)) { set_error("bc_stream_result_undoc_text: no texts / null"); return BC_EINVAL; }
    if (!r->n_undoc) return BC_OK;
    BC_HIP(hipSetDevice(r->device));
    BC_HIP(hipMemcpy(text, r->d_utext, r->n_undoc * (uint64_t)r->text_k, hipMemcpyDeviceToHost));
    BC_HIP(hipMemcpy(counts, r->d_counts, r->n_undoc * 8, hipMemcpyDeviceToHost));
    BC_HIP(hipMemcpy(first, r->d_first, r->n_undoc * 8, hipMemcpyDeviceToHost));
    return BC_OK;
}

int bc_stream_result_pair_info(const bc_stream_result* r, uint64_t* n_pairs, uint64_t* text1_bytes, uint64_t* text2_bytes) {
    if (!r) { set_error("bc_stream_result_pair_info: null"); return BC_EINVAL; }
    if (n_pairs) *n_pairs = r->pair_keys.size();
    if (text1_bytes) *text1_bytes = r->pair_text1.size();
    if (text2_bytes) *text2_bytes = r->pair_text2.size();
    return BC_OK;
}

int bc_stream_result_pair_odd(const bc_stream_result* r, uint64_t* keys, uint64_t* p1, uint8_t* st1, uint64_t* p2,
                              uint8_t* st2, uint64_t* off1, char* text1, uint64_t* off2, char* text2) {
    if (!r) { set_error("bc_stream_result_pair_odd: null"); return BC_EINVAL; }
    std::copy(r->pair_keys.begin(), r->pair_keys.end(), keys);
    std::copy(r->pair_p1.begin(), r->pair_p1.end(), p1);
    std::copy(r->pair_st1.begin(), r->pair_st1.end(), st1);
    std::copy(r->pair_p2.begin(), r->pair_p2.end(), p2);
    std::copy(r->pair_st2.begin(), r->pair_st2.end(), st2);
    std::copy(r->pair_off1.begin(), r->pair_off1.end(), off1);
    std::copy(r->pair_text1.begin(), r->pair_text1.end(), text1);
    std::copy(r->pair_off2.begin(), r->pair_off2.end(), off2);
    std::copy(r->pair_text2.begin(), r->pair_text2.end(), text2);
    return BC_OK;
}

static int stream_count_file_impl(bc_lib* lib, const char* path, uint64_t byte_begin, uint64_t byte_end, uint32_t part,
                                  uint32_t n_parts, int64_t start, const char* L, int L_len, const char* R, int R_len,
                                  int reverse, int max_mismatch, int algo, uint64_t key_base, int n_threads,
                                  uint64_t* h_counts, uint64_t* h_first_seen, bc_stream_result** out);

int bc_stream_count_file(bc_lib* lib, const char* path, uint64_t byte_begin, uint64_t byte_end, int64_t start,
                         const char* L, int L_len, const char* R, int R_len, int reverse, int max_mismatch, int algo,
                         uint64_t key_base, int n_threads, uint64_t* h_counts, uint64_t* h_first_seen,
                         bc_stream_result** out) {
    return stream_count_file_impl(lib, path, byte_begin, byte_end, 0, 1, start, L, L_len, R, R_len, reverse, max_mismatch, algo,
                                  key_base, n_threads, h_counts, h_first_seen, out);
}

// The same for one PART of a file's chunks: chunk c (16 MiB of the decoded stream) belongs to part c % n_parts.  How several
// ranks share a compressed file (every rank decodes the stream; windows, match and tally of a chunk run on one rank): the
// "stride" plan of barcoder_amd.heuristicount.  Keys are key_base + the read's index in the whole file.
int bc_stream_count_file_part(bc_lib* lib, const char* path, uint32_t part, uint32_t n_parts, int64_t start,
                              const char* L, int L_len, const char* R, int R_len, int reverse, int max_mismatch, int algo,
                              uint64_t key_base, int n_threads, uint64_t* h_counts, uint64_t* h_first_seen,
                              bc_stream_result** out) {
    if (n_parts < 1 || part >= n_parts) { set_error("bc_stream_count_file_part: part %u of %u", part, n_parts); return BC_EINVAL; }
    return stream_count_file_impl(lib, path, 0, ~0ull, part, n_parts, start, L, L_len, R, R_len, reverse, max_mismatch, algo,
                                  key_base, n_threads, h_counts, h_first_seen, out);
}

static int stream_count_file_impl(bc_lib* lib, const char* path, uint64_t byte_begin, uint64_t byte_end, uint32_t part,
                                  uint32_t n_parts, int64_t start, const char* L, int L_len, const char* R, int R_len,
                                  int reverse, int max_mismatch, int algo, uint64_t key_base, int n_threads,
                                  uint64_t* h_counts, uint64_t* h_first_seen, bc_stream_result** out) {
    if (!lib || !path || !out) { set_error("bc_stream_count_file: null argument"); return BC_EINVAL; }
    uint32_t G = 0; int k = 0, device = 0;
    bc_lib_info(lib, &G, &k, &device);
    FileJob J;
    int rc;
    if ((rc = fill_geometry(J.g, k, start, L, L_len, R, R_len, reverse, "bc_stream_count_file")) != BC_OK) return rc;
    if ((rc = file_kind(path, "bc_stream_count_file", &J.g.fastq)) != BC_OK) return rc;
    J.g.text_k = lib->text_k;  // > 0: a library of text hashes with its texts attached (barcodes of more than 32 nt)
    if (J.g.text_k && max_mismatch != 0) { set_error("bc_stream_count_file: barcodes of more than 32 nt are matched exactly"); return BC_EINVAL; }
    const auto t_begin = std::chrono::steady_clock::now();
    BC_HIP(hipSetDevice(device));
    struct stat sb;
    if (stat(path, &sb) != 0) { set_error("cannot open %s", path); return BC_EINVAL; }
    uint64_t span = std::min<uint64_t>(byte_end, (uint64_t)sb.st_size) - std::min<uint64_t>(std::min<uint64_t>(byte_begin, byte_end), (uint64_t)sb.st_size);
    if (is_compressed(path)) span *= 4;  // (a guess at the decoded length: it only sizes the first miss list, which grows)
    // run-wide device state
    DevBuf d_counts, d_first;
    MissList misses;
    if ((rc = d_counts.reserve((size_t)G * 4)) != BC_OK) return rc;
    if ((rc = d_first.reserve((size_t)G * 8)) != BC_OK) return rc;
    BC_HIP(hipMemset(d_counts.p, 0, (size_t)G * 4));
    BC_HIP(hipMemset(d_first.p, 0xFF, (size_t)G * 8));
    misses.text_k = J.g.text_k;
    {
        uint64_t est = span / (uint64_t)(24 + J.g.text_k) + 65536;  // grown when the file holds more misses than that
        const char* e = getenv("BC_STREAM_EST_PCT");               // (tests shrink the estimate to exercise the growth path)
        if (e && atoi(e) > 0 && atoi(e) < 100) est = (span / (uint64_t)(24 + J.g.text_k)) * (uint64_t)atoi(e) / 100 + 16;
        if ((rc = misses.reserve(est)) != BC_OK) return rc;
    }
    J.lib = lib; J.path = path; J.byte_begin = byte_begin; J.byte_end = byte_end;
    J.part = part; J.n_parts = n_parts;
    J.max_mismatch = max_mismatch; J.algo = algo; J.n_threads = n_threads; J.key_base = key_base;
    J.d_counts = (uint32_t*)d_counts.p; J.d_first = (uint64_t*)d_first.p; J.misses = &misses;
    if ((rc = run_file(J)) != BC_OK) return rc;
    bc_stream_result* res = new bc_stream_result();
    auto finish = [&]() -> int {
        BC_HIP(hipDeviceSynchronize());
        res->n_reads = J.n_reads;
        int rc3 = tallies_to_host(G, d_counts, d_first, h_counts, h_first_seen);
        if (rc3 != BC_OK) return rc3;
        if ((rc3 = (J.g.text_k ? group_misses_long(misses, device, res) : group_misses(misses, device, res))) != BC_OK) return rc3;
        res->odd_off.push_back(0);
        for (const OddOut& o : J.odd_all) {
            res->odd_keys.push_back(o.key);
            res->odd_text += o.text;
            res->odd_off.push_back(res->odd_text.size());
        }
        return BC_OK;
    };
    rc = finish();
    if (rc != BC_OK) { delete res; return rc; }
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    *out = res;
    return BC_OK;
}

// Paired-end form (heuristicount.py:469-518): both uncompressed files go through the device one after the other,
// each leaving its windows {packed, status} at the read's index in HBM (18 bytes per pair in all); then one pass pairs
// them (st1 == st2 == OK and p1 == p2, the reverse mate's window being reverse-complemented as it is packed),
// the matcher runs once over all pairs, and the unmatched flank-valid pairs are grouped as for a single file.
// path_fwd is the file whose reads carry the barcode in library orientation (the reference's reads1 AFTER its
// need_swap exchange, :444-447), path_rev the other one.  Files with different numbers of reads fail like the
// reference's process_paired_end (:483-486).
int bc_stream_count_pair(bc_lib* lib, const char* path_fwd, const char* path_rev, int64_t start_fwd, const char* L_fwd,
                         int L_fwd_len, const char* R_fwd, int R_fwd_len, int64_t start_rev, const char* L_rev,
                         int L_rev_len, const char* R_rev, int R_rev_len, int max_mismatch, int algo, uint64_t key_base,
                         int n_threads, uint64_t* h_counts, uint64_t* h_first_seen, bc_stream_result** out) {
    return bc_stream_count_pair_range(lib, path_fwd, 0, ~0ull, path_rev, 0, ~0ull, start_fwd, L_fwd, L_fwd_len, R_fwd, R_fwd_len,
                                      start_rev, L_rev, L_rev_len, R_rev, R_rev_len, max_mismatch, algo, key_base, n_threads,
                                      h_counts, h_first_seen, out);
}

// The same over one byte range of each file — ranges that hold the SAME reads (record-aligned, found with
// bc_file_count_newlines / bc_file_seek_newline): one rank's share when the pairs are spread over several GPUs.
int bc_stream_count_pair_range(bc_lib* lib, const char* path_fwd, uint64_t fwd_begin, uint64_t fwd_end, const char* path_rev,
                               uint64_t rev_begin, uint64_t rev_end, int64_t start_fwd, const char* L_fwd, int L_fwd_len,
                               const char* R_fwd, int R_fwd_len, int64_t start_rev, const char* L_rev, int L_rev_len,
                               const char* R_rev, int R_rev_len, int max_mismatch, int algo, uint64_t key_base, int n_threads,
                               uint64_t* h_counts, uint64_t* h_first_seen, bc_stream_result** out) {
    if (!lib || !path_fwd || !path_rev || !out) { set_error("bc_stream_count_pair: null argument"); return BC_EINVAL; }
    // a text-hash library (bc_lib_attach_text: barcodes of more than 32 nt): the stores keep every window's text beside its
    // hash (text_k bytes per read and file), pairs need equal hashes AND equal texts, a hit is verified against the
    // library's text, exact match only
    const int text_k = lib->text_k;
    if (text_k && max_mismatch != 0) { set_error("bc_stream_count_pair: barcodes of more than 32 nt are matched exactly"); return BC_EINVAL; }
    uint32_t G = 0; int k = 0, device = 0;
    bc_lib_info(lib, &G, &k, &device);
    const auto t_begin = std::chrono::steady_clock::now();
    BC_HIP(hipSetDevice(device));
    WindowStore W[2];
    FileJob J[2];
    int rc;
    if ((rc = fill_geometry(J[0].g, k, start_fwd, L_fwd, L_fwd_len, R_fwd, R_fwd_len, 0, "bc_stream_count_pair")) != BC_OK) return rc;
    if ((rc = fill_geometry(J[1].g, k, start_rev, L_rev, L_rev_len, R_rev, R_rev_len, 1, "bc_stream_count_pair")) != BC_OK) return rc;
    const char* paths[2] = {path_fwd, path_rev};
    const uint64_t begins[2] = {fwd_begin, rev_begin}, ends_[2] = {fwd_end, rev_end};
    for (int f = 0; f < 2; ++f) {
        if ((rc = file_kind(paths[f], "bc_stream_count_pair", &J[f].g.fastq)) != BC_OK) return rc;
        J[f].lib = lib; J[f].path = paths[f]; J[f].byte_begin = begins[f]; J[f].byte_end = ends_[f];
        J[f].n_threads = n_threads; J[f].key_base = key_base; J[f].store = &W[f];
        J[f].g.text_k = text_k;
        W[f].text_k = text_k;
    }
    if (is_compressed(paths[0]) || is_compressed(paths[1])) {
        // compressed mates: each file's decoder is one sequential thread at a time — both files at once, so the two
        // decoders run side by side (plain files are read by eight threads each: one after the other, as before)
        int rcs[2] = {BC_OK, BC_OK};
        std::string errs[2];
        std::thread other([&] {
            (void)hipSetDevice(device);
            rcs[1] = run_file(J[1]);
            if (rcs[1] != BC_OK) errs[1] = bc_last_error();
        });
        rcs[0] = run_file(J[0]);
        if (rcs[0] != BC_OK) errs[0] = bc_last_error();
        other.join();
        for (int f = 0; f < 2; ++f)
            if (rcs[f] != BC_OK) { set_error("%s", errs[f].c_str()); return rcs[f]; }
    } else {
        for (int f = 0; f < 2; ++f)
            if ((rc = run_file(J[f])) != BC_OK) return rc;
    }
    if (J[0].n_reads != J[1].n_reads) {
        set_error("Length of reads1 and reads2 must be the same for paired-end data.");  // heuristicount.py:484-486
        return BC_EINVAL;
    }
    const uint64_t n = J[0].n_reads;
    bc_stream_result* res = new bc_stream_result();
    auto body = [&]() -> int {
        res->n_reads = n;
        res->pair_off1.push_back(0);
        res->pair_off2.push_back(0);
        if (n == 0) return BC_OK;
        const uint64_t *p1 = (const uint64_t*)W[0].packed.p, *p2 = (const uint64_t*)W[1].packed.p;
        const uint8_t *s1 = (const uint8_t*)W[0].status.p, *s2 = (const uint8_t*)W[1].status.p;
        // pairs with an odd window: both mates' windows back to the host (a packed mate is rebuilt from its word there —
        // or, barcodes of more than 32 nt, from its text in the store)
        std::vector<uint64_t> keys;
        std::vector<int64_t> ia, ib;  // the pair's entry in each file's odd list (-1: that mate is not odd)
        {
            size_t a = 0, b = 0;
            const auto &A = J[0].odd_all, &B = J[1].odd_all;
            while (a < A.size() || b < B.size()) {
                const uint64_t ka = a < A.size() ? A[a].key : ~0ull, kb = b < B.size() ? B[b].key : ~0ull;
                const uint64_t kk = std::min(ka, kb);
                keys.push_back(kk);
                ia.push_back(ka == kk ? (int64_t)a : -1);
                ib.push_back(kb == kk ? (int64_t)b : -1);
                if (ka == kk) ++a;
                if (kb == kk) ++b;
            }
        }
        const uint64_t m = keys.size();
        std::vector<char> tx1, tx2;  // long barcodes: the stored texts of the listed pairs' mates
        if (m) {
            DevBuf d_idx, o1, o2, os1, os2, ot1, ot2;
            int rc3;
            std::vector<uint64_t> idx(m);
            for (uint64_t j = 0; j < m; ++j) idx[j] = keys[j] - key_base;
            if ((rc3 = d_idx.reserve(m * 8)) != BC_OK || (rc3 = o1.reserve(m * 8)) != BC_OK || (rc3 = o2.reserve(m * 8)) != BC_OK ||
                (rc3 = os1.reserve(m)) != BC_OK || (rc3 = os2.reserve(m)) != BC_OK) return rc3;
            BC_HIP(hipMemcpy(d_idx.p, idx.data(), m * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_gather_pairs, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, (const uint64_t*)d_idx.p, m,
                               p1, s1, p2, s2, (uint64_t*)o1.p, (uint8_t*)os1.p, (uint64_t*)o2.p, (uint8_t*)os2.p);
            BC_HIP(hipGetLastError());
            res->pair_keys = keys;
            res->pair_p1.resize(m); res->pair_p2.resize(m); res->pair_st1.resize(m); res->pair_st2.resize(m);
            BC_HIP(hipMemcpy(res->pair_p1.data(), o1.p, m * 8, hipMemcpyDeviceToHost));
            BC_HIP(hipMemcpy(res->pair_p2.data(), o2.p, m * 8, hipMemcpyDeviceToHost));
            BC_HIP(hipMemcpy(res->pair_st1.data(), os1.p, m, hipMemcpyDeviceToHost));
            BC_HIP(hipMemcpy(res->pair_st2.data(), os2.p, m, hipMemcpyDeviceToHost));
            if (text_k) {
                if ((rc3 = ot1.reserve(m * (uint64_t)text_k)) != BC_OK || (rc3 = ot2.reserve(m * (uint64_t)text_k)) != BC_OK) return rc3;
                hipLaunchKernelGGL(k_gather_texts, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, (const uint64_t*)d_idx.p, m,
                                   (const char*)W[0].texts.p, text_k, (char*)ot1.p);
                hipLaunchKernelGGL(k_gather_texts, dim3((unsigned)((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, (const uint64_t*)d_idx.p, m,
                                   (const char*)W[1].texts.p, text_k, (char*)ot2.p);
                BC_HIP(hipGetLastError());
                tx1.resize(m * (size_t)text_k); tx2.resize(m * (size_t)text_k);
                BC_HIP(hipMemcpy(tx1.data(), ot1.p, tx1.size(), hipMemcpyDeviceToHost));
                BC_HIP(hipMemcpy(tx2.data(), ot2.p, tx2.size(), hipMemcpyDeviceToHost));
            }
        }
        // the mates' texts: an odd mate's stripped read (kept by the host as the chunk went through); a packed mate's is
        // empty — except for barcodes of more than 32 nt, whose packed word is a hash: then the barcode itself, as the
        // library spells it (the caller rebuilds the window from it as it does from a packed word)
        for (uint64_t j = 0; j < m; ++j) {
            if (ia[j] >= 0) res->pair_text1 += J[0].odd_all[(size_t)ia[j]].text;
            else if (text_k && res->pair_st1[j] == BC_WIN_OK) res->pair_text1.append(tx1.data() + j * (size_t)text_k, (size_t)text_k);
            if (ib[j] >= 0) res->pair_text2 += J[1].odd_all[(size_t)ib[j]].text;
            else if (text_k && res->pair_st2[j] == BC_WIN_OK) res->pair_text2.append(tx2.data() + j * (size_t)text_k, (size_t)text_k);
            res->pair_off1.push_back(res->pair_text1.size());
            res->pair_off2.push_back(res->pair_text2.size());
        }
        // pair, match, collect
        DevBuf d_counts, d_first, flags, best, d_cursor;
        MissList misses;
        int rc3;
        if ((rc3 = d_counts.reserve((size_t)G * 4)) != BC_OK) return rc3;
        if ((rc3 = d_first.reserve((size_t)G * 8)) != BC_OK) return rc3;
        if ((rc3 = flags.reserve(n + 64)) != BC_OK) return rc3;
        if ((rc3 = best.reserve(n * 4 + 64)) != BC_OK) return rc3;
        if ((rc3 = d_cursor.reserve(16)) != BC_OK) return rc3;
        BC_HIP(hipMemset(d_counts.p, 0, (size_t)G * 4));
        BC_HIP(hipMemset(d_first.p, 0xFF, (size_t)G * 8));
        BC_HIP(hipMemset(d_cursor.p, 0, 16));
        misses.text_k = text_k;
        if (text_k) {
            // barcodes of more than 32 nt: equal hashes AND equal texts pair; the exact table answers with a hint (no
            // tally), the library's text decides, the verified hits are tallied, the misses keep their texts
            hipLaunchKernelGGL(k_pair_flags_long, dim3(grid_for(n)), dim3(BLOCK), 0, 0, p1, s1, p2, s2, (const char*)W[0].texts.p,
                               (const char*)W[1].texts.p, text_k, n, (uint8_t*)flags.p);
            BC_HIP(hipGetLastError());
            if ((rc3 = bc_match_dev(lib, p1, (const uint8_t*)flags.p, n, 0, BC_ALGO_HASH, (int32_t*)best.p, nullptr, nullptr, nullptr,
                                    nullptr, key_base, nullptr)) != BC_OK) return rc3;
            hipLaunchKernelGGL(k_long_verify_tally, dim3(grid_for(n)), dim3(BLOCK), 0, 0, (int32_t*)best.p, (const uint8_t*)W[0].texts.p,
                               (const uint32_t*)nullptr, lib->d_text, text_k, 0, n, key_base, (uint32_t*)d_counts.p, (uint64_t*)d_first.p);
            BC_HIP(hipGetLastError());
            BC_HIP(hipDeviceSynchronize());
            if ((rc3 = misses.reserve(n)) != BC_OK) return rc3;
            hipLaunchKernelGGL(k_collect_misses_long, dim3(grid_for(n)), dim3(BLOCK), 0, 0, (const int32_t*)best.p, p1,
                               (const uint8_t*)W[0].texts.p, (const uint32_t*)nullptr, text_k, 0, n, key_base, (uint64_t*)misses.words.p,
                               (uint64_t*)misses.keys.p, (char*)misses.texts.p, (unsigned long long*)d_cursor.p);
        } else {
            hipLaunchKernelGGL(k_pair_flags, dim3(grid_for(n)), dim3(BLOCK), 0, 0, p1, s1, p2, s2, n, (uint8_t*)flags.p);
            BC_HIP(hipGetLastError());
            if ((rc3 = bc_match_dev(lib, p1, (const uint8_t*)flags.p, n, max_mismatch, algo, (int32_t*)best.p, nullptr, nullptr,
                                    (uint32_t*)d_counts.p, (uint64_t*)d_first.p, key_base, nullptr)) != BC_OK) return rc3;
            BC_HIP(hipDeviceSynchronize());
            if ((rc3 = misses.reserve(n)) != BC_OK) return rc3;  // every pair could be a miss
            hipLaunchKernelGGL(k_collect_misses, dim3(grid_for(n)), dim3(BLOCK), 0, 0, (const int32_t*)best.p, p1, n, kmer_mask(k),
                               key_base, (uint64_t*)misses.words.p, (uint64_t*)misses.keys.p, (unsigned long long*)d_cursor.p);
        }
        BC_HIP(hipGetLastError());
        unsigned long long used = 0;
        BC_HIP(hipMemcpy(&used, d_cursor.p, 8, hipMemcpyDeviceToHost));
        misses.used = used;
        if ((rc3 = tallies_to_host(G, d_counts, d_first, h_counts, h_first_seen)) != BC_OK) return rc3;
        return text_k ? group_misses_long(misses, device, res) : group_misses(misses, device, res);
    };
    rc = body();
    if (rc != BC_OK) { delete res; return rc; }
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    *out = res;
    return BC_OK;
}

// ---- line arithmetic on an uncompressed file (host): how ranks agree on byte ranges that hold the same reads ------
// newlines in [begin, end) of the file (end clamped to its size); *size_out = file size; *last_byte_out = the file's
// last byte (0 for an empty file): lines = newlines + (size > 0 and last byte != '\n')
int bc_file_count_newlines(const char* path, uint64_t begin, uint64_t end, uint64_t* n_out, uint64_t* size_out, int* last_byte_out) {
    if (!path || !n_out) { set_error("bc_file_count_newlines: null argument"); return BC_EINVAL; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error("cannot open %s", path); return BC_EINVAL; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) { set_error("cannot stat %s", path); return BC_EINVAL; }
    const uint64_t size = (uint64_t)sb.st_size;
    end = std::min(end, size);
    begin = std::min(begin, end);
    std::vector<uint8_t> buf(8u << 20);
    uint64_t n = 0;
    for (uint64_t at = begin; at < end;) {
        const size_t want = (size_t)std::min<uint64_t>(buf.size(), end - at);
        const ssize_t got = pread(fd, buf.data(), want, (off_t)at);
        if (got <= 0) { set_error("read error in %s", path); return BC_EINVAL; }
        n += count_nl(buf.data(), (uint64_t)got);
        at += (uint64_t)got;
    }
    *n_out = n;
    if (size_out) *size_out = size;
    if (last_byte_out) {
        uint8_t c = 0;
        if (size && pread(fd, &c, 1, (off_t)(size - 1)) != 1) { set_error("read error in %s", path); return BC_EINVAL; }
        *last_byte_out = (int)c;
    }
    return BC_OK;
}

// offset of the byte after the nth newline (1-based) at or after `begin`; nth = 0 gives `begin`; past the last newline
// the file size
int bc_file_seek_newline(const char* path, uint64_t begin, uint64_t nth, uint64_t* offset_out) {
    if (!path || !offset_out) { set_error("bc_file_seek_newline: null argument"); return BC_EINVAL; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error("cannot open %s", path); return BC_EINVAL; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) { set_error("cannot stat %s", path); return BC_EINVAL; }
    const uint64_t size = (uint64_t)sb.st_size;
    begin = std::min(begin, size);
    if (nth == 0) { *offset_out = begin; return BC_OK; }
    std::vector<uint8_t> buf(8u << 20);
    uint64_t seen = 0;
    for (uint64_t at = begin; at < size;) {
        const size_t want = (size_t)std::min<uint64_t>(buf.size(), size - at);
        const ssize_t got = pread(fd, buf.data(), want, (off_t)at);
        if (got <= 0) { set_error("read error in %s", path); return BC_EINVAL; }
        const uint64_t c = count_nl(buf.data(), (uint64_t)got);
        if (seen + c >= nth) {
            for (ssize_t i = 0; i < got; ++i)
                if (buf[(size_t)i] == '\n' && ++seen == nth) { *offset_out = at + (uint64_t)i + 1; return BC_OK; }
        }
        seen += c;
        at += (uint64_t)got;
    }
    *offset_out = size;
    return BC_OK;
}

}  // extern "C"
