// bc_host.cpp — host-side half of the C ABI: error string, encoders (K1), the
// process_chunk window/flank front end, genome packing.  No device code here.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "bc_common.h"

namespace bc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace bc

using namespace bc;

extern "C" {

int bc_version(void) { return BC_VERSION; }
const char* bc_last_error(void) { return g_err; }

int bc_pack_kmers(const char* h_seqs, size_t n, int k, size_t stride, uint64_t* h_out,
                  uint8_t* h_flags) {
    if (k < 1 || k > 32 || (!h_seqs && n) || (!h_out && n) || stride < (size_t)k) {
        set_error("bc_pack_kmers: bad argument (k=%d stride=%zu)", k, stride);
        return BC_EINVAL;
    }
    for (size_t i = 0; i < n; ++i) {
        const unsigned char* s = (const unsigned char*)h_seqs + i * stride;
        uint32_t lo = 0, hi = 0;
        int bad = 0;
        for (int j = 0; j < k; ++j) {
            int c = base_code(s[j]);
            if (c > 3) { bad = 1; c = 0; }
            lo |= (uint32_t)(c & 1) << j;
            hi |= (uint32_t)(c >> 1) << j;
        }
        uint64_t w = (uint64_t)lo | ((uint64_t)hi << 32);
        if (bad && k <= 31) w |= BC_SKIP_BIT;
        h_out[i] = w;
        if (h_flags) h_flags[i] = (uint8_t)bad;
    }
    return BC_OK;
}

int bc_unpack_kmers(const uint64_t* h_packed, size_t n, int k, char* h_out) {
    if (k < 1 || k > 32 || (!h_packed && n) || (!h_out && n)) {
        set_error("bc_unpack_kmers: bad argument");
        return BC_EINVAL;
    }
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < n; ++i) {
        uint32_t lo = (uint32_t)h_packed[i], hi = (uint32_t)(h_packed[i] >> 32);
        for (int j = 0; j < k; ++j) h_out[i * k + j] = L[((lo >> j) & 1) | (((hi >> j) & 1) << 1)];
    }
    return BC_OK;
}

int bc_revcomp_kmers(const uint64_t* h_in, size_t n, int k, uint64_t* h_out) {
    if (k < 1 || k > 32 || (!h_in && n) || (!h_out && n)) {
        set_error("bc_revcomp_kmers: bad argument");
        return BC_EINVAL;
    }
    for (size_t i = 0; i < n; ++i) {
        uint64_t skip = (k <= 31) ? (h_in[i] & BC_SKIP_BIT) : 0;
        h_out[i] = revcomp_word(h_in[i] & kmer_mask(k), k) | skip;
    }
    return BC_OK;
}

// ---- process_chunk front half -------------------------------------------------
static void pack_windows_range(const char* text, const uint64_t* off, size_t a, size_t b,
                               int64_t start, const char* L, int L_len, const char* R,
                               int R_len, int k, int reverse, uint64_t* out, uint8_t* status) {
    const int64_t wlen = (int64_t)L_len + k + R_len;
    for (size_t i = a; i < b; ++i) {
        const char* rd = text + off[i];
        const int64_t len = (int64_t)(off[i + 1] - off[i]);
        uint64_t w = BC_SKIP_BIT;
        uint8_t st = BC_WIN_SKIP;
        // heuristicount.py:524 — any 'N' anywhere in the read drops it
        if (len > 0 && memchr(rd, 'N', (size_t)len) != nullptr) {
            out[i] = (k <= 31) ? w : 0;
            status[i] = st;
            continue;
        }
        // heuristicount.py:527 — python slice, clamped to the read
        int64_t ws = std::min<int64_t>(start, len);
        int64_t we = std::min<int64_t>(start + wlen, len);
        int64_t n = we - ws;  // actual window length
        const char* win = rd + ws;
        // heuristicount.py:458-460 — startswith(L) and endswith(R) on the window
        bool has = n >= L_len && n >= R_len && (L_len == 0 || memcmp(win, L, (size_t)L_len) == 0) &&
                   (R_len == 0 || memcmp(win + n - R_len, R, (size_t)R_len) == 0);
        if (has) {
            if (n == wlen) {
                uint32_t lo = 0, hi = 0;
                int bad = 0;
                const unsigned char* s = (const unsigned char*)win + L_len;
                for (int j = 0; j < k; ++j) {
                    int c = base_code(s[j]);
                    if (c > 3) { bad = 1; break; }
                    lo |= (uint32_t)(c & 1) << j;
                    hi |= (uint32_t)(c >> 1) << j;
                }
                if (!bad) {
                    w = (uint64_t)lo | ((uint64_t)hi << 32);
                    if (reverse) w = revcomp_word(w, k);  // heuristicount.py:532-533
                    st = BC_WIN_OK;
                } else {
                    st = BC_WIN_ODD;
                }
            } else {
                st = BC_WIN_ODD;  // truncated window that still shows both flanks
            }
        }
        if (st != BC_WIN_OK) w = (k <= 31) ? BC_SKIP_BIT : 0;
        out[i] = w;
        status[i] = st;
    }
}

int bc_pack_windows(const char* h_text, const uint64_t* h_off, size_t n_reads, int64_t start,
                    const char* L, int L_len, const char* R, int R_len, int k, int reverse,
                    int n_threads, uint64_t* h_out, uint8_t* h_status) {
    if (k < 1 || k > 32 || start < 0 || L_len < 0 || R_len < 0 || (L_len && !L) || (R_len && !R) ||
        (n_reads && (!h_off || !h_out || !h_status))) {
        set_error("bc_pack_windows: bad argument (k=%d start=%lld)", k, (long long)start);
        return BC_EINVAL;
    }
    if (n_reads == 0) return BC_OK;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = n_threads > 0 ? (size_t)n_threads : (hw ? hw : 1);
    nt = std::max<size_t>(1, std::min(nt, n_reads / 8192 + 1));
    if (nt == 1) {
        pack_windows_range(h_text, h_off, 0, n_reads, start, L, L_len, R, R_len, k, reverse, h_out,
                           h_status);
        return BC_OK;
    }
    std::vector<std::thread> th;
    size_t per = (n_reads + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        size_t a = t * per, b = std::min(n_reads, a + per);
        if (a >= b) break;
        th.emplace_back(pack_windows_range, h_text, h_off, a, b, start, L, L_len, R, R_len, k,
                        reverse, h_out, h_status);
    }
    for (auto& t : th) t.join();
    return BC_OK;
}

// ---- barcodes longer than 32 nt (exact match only) --------------------------------
// The reference's set lookup (heuristicount.py:449-453) is length-agnostic up to 1 000 nt (:75-97); the packed matcher
// holds 32 bases.  Longer windows travel as a 64-bit hash of the oriented barcode text (what the GPU's exact table
// looks up, as a pseudo 32-mer) together with the text itself (what the caller verifies a hit against and names an
// undocumented window by): bc_hash_windows is bc_pack_windows' twin for them, bc_hash_kmers hashes the library.
static inline uint64_t text_hash64(const unsigned char* p, size_t n) {
    return text_hash64_at(n, [p](size_t i) { return p[i]; });
}

int bc_hash_kmers(const char* h_seqs, size_t n, int k, uint64_t* h_out) {
    if (k < 1 || (!h_seqs && n) || (!h_out && n)) { set_error("bc_hash_kmers: bad argument"); return BC_EINVAL; }
    for (size_t i = 0; i < n; ++i) h_out[i] = text_hash64((const unsigned char*)h_seqs + i * (size_t)k, (size_t)k);
    return BC_OK;
}

static void hash_windows_range(const char* text, const uint64_t* off, size_t a, size_t b, int64_t start, const char* L,
                               int L_len, const char* R, int R_len, int k, int reverse, uint64_t* out, uint8_t* status,
                               char* wbuf) {
    const int64_t wlen = (int64_t)L_len + k + R_len;
    for (size_t i = a; i < b; ++i) {
        const char* rd = text + off[i];
        const int64_t len = (int64_t)(off[i + 1] - off[i]);
        char* dst = wbuf + i * (size_t)k;
        out[i] = 0;
        status[i] = BC_WIN_SKIP;
        if (len > 0 && memchr(rd, 'N', (size_t)len) != nullptr) continue;  // heuristicount.py:524
        const int64_t ws = std::min<int64_t>(start, len), we = std::min<int64_t>(start + wlen, len);
        const int64_t n = we - ws;
        const char* win = rd + ws;
        const bool has = n >= L_len && n >= R_len && (L_len == 0 || memcmp(win, L, (size_t)L_len) == 0) &&
                         (R_len == 0 || memcmp(win + n - R_len, R, (size_t)R_len) == 0);
        if (!has) continue;
        status[i] = BC_WIN_ODD;  // flanks present; a short window or an odd character keeps string semantics
        if (n != wlen) continue;
        const unsigned char* sq = (const unsigned char*)win + L_len;
        bool bad = false;
        for (int j = 0; j < k && !bad; ++j) bad = base_code(sq[j]) > 3;
        if (bad) continue;
        if (reverse) {  // heuristicount.py:532-533: the barcode as the library spells it
            for (int j = 0; j < k; ++j) {
                const char c = (char)sq[k - 1 - j];
                dst[j] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C';
            }
        } else {
            memcpy(dst, sq, (size_t)k);
        }
        out[i] = text_hash64((const unsigned char*)dst, (size_t)k);
        status[i] = BC_WIN_OK;
    }
}

int bc_hash_windows(const char* h_text, const uint64_t* h_off, size_t n_reads, int64_t start, const char* L, int L_len,
                    const char* R, int R_len, int k, int reverse, int n_threads, uint64_t* h_out, uint8_t* h_status,
                    char* h_windows) {
    if (k < 1 || k > 1000 || start < 0 || L_len < 0 || R_len < 0 || (L_len && !L) || (R_len && !R) ||
        (n_reads && (!h_off || !h_out || !h_status || !h_windows))) {
        set_error("bc_hash_windows: bad argument (k=%d start=%lld)", k, (long long)start);
        return BC_EINVAL;
    }
    if (n_reads == 0) return BC_OK;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = n_threads > 0 ? (size_t)n_threads : (hw ? hw : 1);
    nt = std::max<size_t>(1, std::min(nt, n_reads / 8192 + 1));
    std::vector<std::thread> th;
    const size_t per = (n_reads + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        const size_t a = t * per, b = std::min(n_reads, a + per);
        if (a >= b) break;
        if (nt == 1) hash_windows_range(h_text, h_off, a, b, start, L, L_len, R, R_len, k, reverse, h_out, h_status, h_windows);
        else th.emplace_back(hash_windows_range, h_text, h_off, a, b, start, L, L_len, R, R_len, k, reverse, h_out, h_status, h_windows);
    }
    for (auto& t : th) t.join();
    return BC_OK;
}

// ---- genome packing -------------------------------------------------------------
int bc_genome_pack(const char* h_seq, uint64_t len, uint64_t wrap, uint64_t* h_bases,
                   uint32_t* h_invalid) {
    if ((!h_seq && len) || !h_bases || !h_invalid || (wrap && len == 0)) {
        set_error("bc_genome_pack: bad argument");
        return BC_EINVAL;
    }
    uint64_t total = len + wrap;
    uint64_t nw = (total + 31) / 32;
    for (uint64_t w = 0; w < nw; ++w) {
        uint32_t lo = 0, hi = 0, inv = 0;
        for (int j = 0; j < 32; ++j) {
            uint64_t p = w * 32 + j;
            if (p >= total) { inv |= 1u << j; continue; }
            unsigned char ch = (unsigned char)h_seq[p < len ? p : (p - len) % len];
            int c = base_code_nocase(ch);
            if (c > 3) { inv |= 1u << j; c = 0; }
            lo |= (uint32_t)(c & 1) << j;
            hi |= (uint32_t)(c >> 1) << j;
        }
        h_bases[w] = (uint64_t)lo | ((uint64_t)hi << 32);
        h_invalid[w] = inv;
    }
    return BC_OK;
}

}  // extern "C"
