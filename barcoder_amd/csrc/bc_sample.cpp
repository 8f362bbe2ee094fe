// bc_sample.cpp — native sample_data / find_flanks.
//
// Host logic, no device code.  Restates, with the same observable results, the
// orientation / offset / flank discovery that heuristicount.py runs once per job
// on the first chunks of reads:
//   sample_data   heuristicount.py:156-379
//   find_flanks   heuristicount.py:382-425
// including its quirks (a k-mer is only examined once per chunk, the mate's
// k-mer is skipped whenever the first mate's was already seen, "observed"
// records the first mate's k-mer for both mates, Counter ties resolve to the
// first inserted key).  The Python mirror in barcoder_amd/heuristicount.py is
// the golden-tested statement of the same rules; tests compare the two.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "bc_common.h"

using namespace bc;

extern "C" {
typedef struct bc_reader bc_reader;
int bc_reader_open(const char* path, bc_reader** out);
int bc_reader_next(bc_reader* r, uint64_t max_reads, char* text, uint64_t text_cap, uint64_t* off, uint64_t* n_out);
int bc_reader_close(bc_reader* r);
}

namespace {

// collections.Counter with most_common()'s tie rule: equal counts keep insertion order
template <typename K>
struct OrderedCounter {
    std::vector<std::pair<K, int64_t>> items;
    std::unordered_map<K, size_t> at;
    void add(const K& k, int64_t c = 1) {
        auto it = at.find(k);
        if (it == at.end()) { at.emplace(k, items.size()); items.emplace_back(k, c); }
        else items[it->second].second += c;
    }
    std::vector<std::pair<K, int64_t>> most_common(size_t n) const {
        std::vector<std::pair<K, int64_t>> v(items);
        std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
        if (v.size() > n) v.resize(n);
        return v;
    }
    bool empty() const { return items.empty(); }
};

// Set of strings for the sampler's membership tests: open addressing over {hash, pointer, length} slots (one cache
// line per probe where std::unordered_set walks bucket -> node -> characters), the hash of a string computed once by
// the caller and shared by every set it is looked up in, storage either the caller's (views into the chunk's text:
// the per-chunk sets) or a chunked arena (the sets that outlive a chunk).
static inline uint64_t str_hash(std::string_view v) {
    const unsigned char* p = (const unsigned char*)v.data();
    size_t n = v.size();
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
        p += 8; n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
    }
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

struct StrSet {
    struct Slot { uint64_t h; const char* p; uint64_t len; };  // p == nullptr: empty
    std::vector<Slot> tab;
    size_t count = 0, mask = 0;
    bool own;                                     // copy inserted strings into the arena
    std::vector<std::unique_ptr<char[]>> blocks;
    size_t used = 0, cap = 0;
    explicit StrSet(size_t expect = 0, bool own_ = true) : own(own_) {
        size_t n = 64;
        while (n < expect * 2 + 16) n <<= 1;
        tab.assign(n, Slot{0, nullptr, 0});
        mask = n - 1;
    }
    void clear() {
        std::fill(tab.begin(), tab.end(), Slot{0, nullptr, 0});
        count = 0;
    }
    bool contains(uint64_t h, std::string_view v) const {
        for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
            const Slot& s = tab[i];
            if (!s.p) return false;
            if (s.h == h && s.len == v.size() && memcmp(s.p, v.data(), v.size()) == 0) return true;
        }
    }
    bool contains(std::string_view v) const { return contains(str_hash(v), v); }
    int64_t find_slot(uint64_t h, std::string_view v) const {  // the entry's slot (stable while the set does not grow), -1 = absent
        for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
            const Slot& s = tab[i];
            if (!s.p) return -1;
            if (s.h == h && s.len == v.size() && memcmp(s.p, v.data(), v.size()) == 0) return (int64_t)i;
        }
    }
    void prefetch(uint64_t h) const { __builtin_prefetch(&tab[(size_t)h & mask]); }
    void grow() {
        std::vector<Slot> old;
        old.swap(tab);
        tab.assign(old.size() * 2, Slot{0, nullptr, 0});
        mask = tab.size() - 1;
        for (const Slot& s : old)
            if (s.p) {
                size_t i = (size_t)s.h & mask;
                while (tab[i].p) i = (i + 1) & mask;
                tab[i] = s;
            }
    }
    bool insert(uint64_t h, std::string_view v) { return insert_at(h, v, own) != nullptr; }  // true when new
    // where the string now lives (nullptr: it was there already); `copy`: into the arena, else the caller's bytes stay
    const char* insert_at(uint64_t h, std::string_view v, bool copy) {
        size_t i = (size_t)h & mask;
        for (;; i = (i + 1) & mask) {
            const Slot& s = tab[i];
            if (!s.p) break;
            if (s.h == h && s.len == v.size() && memcmp(s.p, v.data(), v.size()) == 0) return nullptr;
        }
        const char* at = v.data();
        if (copy) {
            const size_t need = std::max<size_t>(v.size(), 1);  // (an empty string still needs a non-null address)
            if (used + need > cap) {
                cap = std::max<size_t>(1u << 22, need);
                blocks.emplace_back(new char[cap]);
                used = 0;
            }
            char* dst = blocks.back().get() + used;
            memcpy(dst, v.data(), v.size());
            used += need;
            at = dst;
        } else if (!at) {
            at = "";
        }
        tab[i] = Slot{h, at, v.size()};
        if (++count * 2 > tab.size()) grow();
        return at;
    }
    bool insert(std::string_view v) { return insert(str_hash(v), v); }
    size_t size() const { return count; }
};

// The library (or its reverse complements) as a set in eight parts by the top bits of the hash: the parts are filled by
// as many threads, never grow afterwards, and every entry has a stable id (part base + slot) — what the per-chunk
// "first occurrence" table of the sampling pass is indexed by.
struct LibSet {
    std::vector<StrSet> part;
    size_t base[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    explicit LibSet(size_t expect) { for (int p = 0; p < 8; ++p) part.emplace_back(expect / 8 + expect / 32 + 16, false); }
    // hashes of the G entries (computed by the caller's threads), entry g = at(g)
    template <typename At>
    void build(const std::vector<uint64_t>& h, uint32_t G, At at, unsigned threads) {
        auto fill = [&](unsigned p0, unsigned step) {
            for (unsigned p = p0; p < 8; p += step)
                for (uint32_t g = 0; g < G; ++g)
                    if ((unsigned)(h[g] >> 61) == p) part[p].insert(h[g], at(g));
        };
        const unsigned step = threads >= 8 ? 8u : threads >= 4 ? 4u : threads >= 2 ? 2u : 1u;
        if (step == 1) fill(0, 1);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < step; ++t) th.emplace_back(fill, t, step);
            for (auto& x : th) x.join();
        }
        for (int p = 0; p < 8; ++p) base[p + 1] = base[p] + part[p].tab.size();
    }
    int64_t find(uint64_t h, std::string_view v) const {
        const unsigned p = (unsigned)(h >> 61);
        const int64_t s = part[p].find_slot(h, v);
        return s < 0 ? -1 : (int64_t)base[p] + s;
    }
    bool contains(uint64_t h, std::string_view v) const { return part[h >> 61].contains(h, v); }
    size_t size() const { size_t n = 0; for (const StrSet& s : part) n += s.size(); return n; }
    size_t slots() const { return base[8]; }
};

// a chunk's text: bytes that nobody zeroes first (a std::vector<char> of 6 MB per chunk costs a pass over them)
struct TextBuf {
    std::unique_ptr<char[]> p;
    size_t n = 0;
    char* data() { return p.get(); }
    const char* data() const { return p.get(); }
    size_t size() const { return n; }
    void resize(size_t m) {  // keeps the first min(n, m) bytes
        if (m == n) return;
        std::unique_ptr<char[]> q(new char[std::max<size_t>(m, 1)]);
        if (n && m) memcpy(q.get(), p.get(), std::min(n, m));
        p = std::move(q);
        n = m;
    }
};

struct Chunk {
    TextBuf text;
    std::vector<uint64_t> off;
    uint64_t n = 0;
    std::string_view read(uint64_t i) const { return std::string_view(text.data() + off[i], off[i + 1] - off[i]); }
};

// exactly `want` reads (or what is left): bc_reader_next stops early when its text buffer fills
int next_chunk(bc_reader* r, uint64_t want, Chunk& c) {
    c.n = 0;
    c.off.assign(want + 1, 0);
    if (c.text.size() < want * 64 + 4096) c.text.resize(want * 64 + 4096);
    uint64_t used = 0;
    std::vector<uint64_t> off(want + 1);
    while (c.n < want) {
        uint64_t got = 0;
        int rc = bc_reader_next(r, want - c.n, c.text.data() + used, c.text.size() - used, off.data(), &got);
        if (rc == BC_ECAP) { c.text.resize(c.text.size() * 2); continue; }  // not even one read fits: grow, retry
        if (rc != BC_OK) return rc;
        if (got == 0) break;  // end of file
        for (uint64_t j = 1; j <= got; ++j) c.off[c.n + j] = used + off[j];
        used += off[got];
        c.n += got;
        if (c.n < want) c.text.resize(c.text.size() * 2);  // stopped for lack of room
    }
    return BC_OK;
}

constexpr int PF_SMALL_BITS = 20;
constexpr int PF_BITS = 24;  // bits of the pre-filter table (2 MiB): 20 / 22 / 24 / 26 measured 1.00 / 0.84 / 0.75 / 0.75 s per sampling pass
// base code by table (a switch on sequence characters mispredicts on nearly every base)
struct CodeTable {
    unsigned char t[256];
    constexpr CodeTable() : t() {
        for (int i = 0; i < 256; ++i) t[i] = 4;
        t[(int)'A'] = 0; t[(int)'C'] = 1; t[(int)'G'] = 2; t[(int)'T'] = 3;
    }
    constexpr unsigned char operator[](unsigned char c) const { return t[c]; }
};
constexpr CodeTable kCode{};

// rolling 2-bit value of the k-mer window at position i of a read (positions visited in ascending order)
struct Roll {
    int k;
    uint64_t v = 0, mask;
    size_t next = 0;   // first character not yet shifted in
    int run = 0;       // ACGT characters in a row up to `next`
    explicit Roll(int k_) : k(k_), mask(k_ >= 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull)) {}
    static uint32_t hash(uint64_t v) { return (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> (64 - PF_BITS)); }
    // first level: 2^20 bits (128 KiB, stays in the core's L2) under an independent hash; at 2e5 keys it turns away
    // five windows in six before the 2 MiB table (an L3 access per window otherwise: two thirds of the sampler's time)
    static uint32_t hash_small(uint64_t v) { return (uint32_t)((v * 0xD6E8FEB86659FD93ull) >> (64 - PF_SMALL_BITS)); }
    // can the window [i, i + k) be a library entry?  (pure ACGT and its hash bit is set)
    bool at(std::string_view s, size_t i, const std::vector<uint64_t>& bits, const std::vector<uint64_t>& small) {
        const size_t end = i + (size_t)k;
        for (; next < end; ++next) {
            const int c = kCode[(unsigned char)s[next]];
            if (c < 4) { v = ((v << 2) | (uint64_t)c) & mask; ++run; }
            else { v = (v << 2) & mask; run = 0; }
        }
        if (run < k) return false;
        const uint32_t hs = hash_small(v);
        if (!((small[hs >> 6] >> (hs & 63)) & 1ull)) return false;
        const uint32_t h = hash(v);
        return (bits[h >> 6] >> (h & 63)) & 1ull;
    }
};

}  // namespace

struct bc_sample {
    int64_t n_novel = 0, off1 = -1, off2 = -1;  // -1 = None
    std::vector<std::string_view> valid1, valid2;  // sample1 / sample2 AFTER the swap rule: views into what follows
    std::vector<std::string> strs[2];              // (the paired / unfiltered path keeps copies)
    std::vector<TextBuf> texts;                    // (single-end fast path: the sampled chunks' text, kept)
    std::vector<StrSet> arenas;                    // (... and the sets whose arenas hold what was copied instead)
    uint64_t n_observed = 0, n_chunks = 0;
    int need_swap = 0, satisfied = 0;
};

// worker threads of the sampling pass for a chunk of nz reads: one per 2 048 reads, at most 16 (32 on a host with 64+
// hardware threads: measured on the 128-thread GPU box); BC_SAMPLE_THREADS overrides (tuning, tests)
static uint64_t sample_threads(unsigned hw, uint64_t nz) {
    if (const char* e = getenv("BC_SAMPLE_THREADS"))
        if (atoi(e) >= 1 && atoi(e) <= 256) return (uint64_t)atoi(e);
    return std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw >= 64 ? 32 : 16, hw), nz / 2048 + 1));
}

extern "C" {

int bc_sample_data(const char* file1, const char* file2, const char* barcodes, uint32_t G, int k, bc_sample** out) {
    if (!file1 || !barcodes || !out || G == 0 || k < 1) { set_error("bc_sample_data: bad argument"); return BC_EINVAL; }
    const bool paired = file2 && *file2;
    const auto t_entry = std::chrono::steady_clock::now();
    // the library and its reverse complements as two contiguous blobs (no per-entry allocation), the sets hold views.
    // Everything per entry — reverse complement, the two string hashes, the purity test and the pre-filter bits — is done
    // by worker threads over ranges of the library; the sets are then filled in eight parts by as many threads.
    std::vector<char> rev_blob((size_t)G * k);
    auto fwd_at = [&](uint32_t g) { return std::string_view(barcodes + (size_t)g * k, (size_t)k); };
    auto rev_at = [&](uint32_t g) { return std::string_view(rev_blob.data() + (size_t)g * k, (size_t)k); };
    const unsigned hw0 = std::max(1u, std::thread::hardware_concurrency());
    const unsigned TS = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(16, hw0), G / 2048 + 1));
    // Pre-filter for the k-mer sweep (most windows of a read are no barcode): 2^24 bits set at a hash of the packed
    // value of every library entry and reverse complement.  Only when every entry is pure upper-case ACGT and fits a
    // 64-bit packing; otherwise the sweep asks the sets for every window, as the reference does.
    std::vector<uint64_t> pf_bits, pf_small, hash_f(G), hash_r(G);
    if (k <= 32) {
        pf_bits.assign((1u << PF_BITS) / 64, 0);
        pf_small.assign((1u << PF_SMALL_BITS) / 64, 0);
    }
    std::atomic<bool> pure{k <= 32};
    auto per_entry = [&](uint32_t g0, uint32_t g1) {
        bool all_pure = true;
        for (uint32_t g = g0; g < g1; ++g) {
            const char* src = barcodes + (size_t)g * k;
            char* dst = rev_blob.data() + (size_t)g * k;
            for (int j = 0; j < k; ++j) {
                char c = src[k - 1 - j];
                switch (c) {
                    case 'A': c = 'T'; break;
                    case 'T': c = 'A'; break;
                    case 'C': c = 'G'; break;
                    case 'G': c = 'C'; break;
                    default: break;
                }
                dst[j] = c;
            }
            hash_f[g] = str_hash(fwd_at(g));
            hash_r[g] = str_hash(rev_at(g));
            if (k > 32) continue;
            for (const std::string_view sp : {fwd_at(g), rev_at(g)}) {
                uint64_t v = 0;
                for (int j = 0; j < k; ++j) {
                    const int c = base_code((unsigned char)sp[j]);
                    all_pure = all_pure && c < 4;
                    v = (v << 2) | (uint64_t)(c & 3);
                }
                // (bits of an impure entry are harmless: one impure entry switches the pre-filter off altogether)
                const uint32_t h = Roll::hash(v), hs = Roll::hash_small(v);
                __atomic_fetch_or(&pf_bits[h >> 6], 1ull << (h & 63), __ATOMIC_RELAXED);
                __atomic_fetch_or(&pf_small[hs >> 6], 1ull << (hs & 63), __ATOMIC_RELAXED);
            }
        }
        if (!all_pure) pure.store(false, std::memory_order_relaxed);
    };
    if (TS == 1) per_entry(0, G);
    else {
        std::vector<std::thread> th;
        const uint32_t per = (G + TS - 1) / TS;
        for (unsigned t = 0; t < TS; ++t) th.emplace_back(per_entry, std::min(G, t * per), std::min(G, (t + 1) * per));
        for (auto& x : th) x.join();
    }
    LibSet fwd(G), rev(G);
    {
        std::thread other([&] { rev.build(hash_r, G, rev_at, std::max(1u, TS / 2)); });
        fwd.build(hash_f, G, fwd_at, std::max(1u, TS / 2));
        other.join();
    }
    const uint64_t n_lib = fwd.size();  // len(barcodes) of the SET (heuristicount.py:174-176)
    const bool prefilter = pure.load();
    bc_reader *r1 = nullptr, *r2 = nullptr;
    int rc = bc_reader_open(file1, &r1);
    if (rc != BC_OK) return rc;
    if (paired && (rc = bc_reader_open(file2, &r2)) != BC_OK) { bc_reader_close(r1); return rc; }

    const bool prof = getenv("BC_SAMPLE_PROF") != nullptr;  // tuning aid: where the sampling pass spends its time
    auto now = [] { return std::chrono::steady_clock::now(); };
    double t_setup = 0, t_read = 0, t_sweep = 0, t_prep = 0, t_rem = 0, t_collect = 0, t_count = 0, t_valid = 0, t_alloc = 0;
    static thread_local std::chrono::steady_clock::time_point t_enter;
    t_setup = std::chrono::duration<double, std::milli>(now() - t_entry).count();
    auto t_mark = now();
    int64_t div[2] = {0, 0};
    OrderedCounter<int> orients[2];      // 0 = "forward", 1 = "reverse"
    OrderedCounter<int64_t> offsets[2];
    std::vector<std::string> valid_store[2];
    const bool lookahead = !paired && prefilter;  // the single-end fast path (below)
    // (single-end runs with the pre-filter count their distinct reads in seen_part instead: no table to clear for nothing)
    StrSet valid[2] = {StrSet(lookahead ? 0 : 2 * (size_t)n_lib), StrSet(paired ? 2 * (size_t)n_lib : 0)},
           seen_reads((lookahead || (paired && prefilter)) ? 0 : 6 * (size_t)n_lib), observed(lookahead ? 0 : 2 * (size_t)n_lib);
    StrSet c_barcodes((size_t)n_lib, false), c_reads((size_t)n_lib, false);  // per chunk: views into the chunk's text
    std::vector<StrSet> seen_part;  // single-end runs with the pre-filter: seen_reads in eight parts (see `remember`)
    for (int p = 0; p < 8; ++p) seen_part.emplace_back((size_t)0);  // (they grow inside their own threads)
    // ---- state of the single-end fast path ------------------------------------------------------------------------
    // valid reads (sample1): a set in eight parts by the hash, filled by as many threads, and the reads in the order the
    // reference would have added them; "observed" barcodes as one bit per library id; the text of the sampled chunks is
    // KEPT (up to KEEP_BYTES: the sets then hold views instead of copies; past that they copy into their arenas)
    size_t KEEP_BYTES = 1ull << 30;
    if (const char* e = getenv("BC_SAMPLE_KEEP_MB"))  // (tests: 0 makes every chunk take the copying form)
        if (atoi(e) >= 0) KEEP_BYTES = (size_t)atoi(e) << 20;
    std::vector<StrSet> vset;
    for (int p = 0; p < 8; ++p) vset.emplace_back(lookahead ? (size_t)n_lib / 4 : 0, true);
    std::vector<std::string_view> valid_views;
    std::vector<TextBuf> kept;
    size_t kept_bytes = 0;
    const size_t fwd_slots = fwd.slots(), id_slots = fwd_slots + rev.slots();
    std::vector<uint64_t> obs_bits(lookahead ? (id_slots + 63) / 64 : 0, 0);
    // first[id] = the earliest (read, candidate) of the CURRENT chunk whose window is library entry `id`.  Keys of later
    // chunks are smaller than any key of an earlier one, so the table is never cleared.
    std::unique_ptr<std::atomic<uint64_t>[]> first(lookahead ? new std::atomic<uint64_t>[id_slots] : nullptr);
    if (lookahead)
        for (size_t i = 0; i < id_slots; ++i) first[i].store(~0ull, std::memory_order_relaxed);
    auto seen_count = [&]() {
        size_t n = seen_reads.size();
        for (const StrSet& p : seen_part) n += p.size();
        return n;
    };
    // what the worker threads note per read of a chunk (see the loop): reused from chunk to chunk, never cleared
    constexpr uint64_t AHEAD = 12;
    constexpr int MAXC = 6;
    struct Ahead { uint64_t h1; int ncand; bool overflow; uint32_t pos[MAXC]; uint32_t id[MAXC]; uint8_t in[MAXC]; };  // in: bit 0 forward set, bit 1 reverse
    std::vector<Ahead> prep;
    // mate pairs with the pre-filter: the windows of both mates that are library entries, found by the worker threads; the
    // reference's loop then runs over those alone (it does nothing at any other position), and the distinct-read count is
    // taken by partitions afterwards (it depends on which pairs the loop skipped)
    const bool lookahead2 = paired && prefilter;
    constexpr int MAXC2 = 8;
    struct Ahead2 { uint64_t h1, h2; int ncand; bool overflow, skipped; uint32_t pos[MAXC2]; uint8_t in1[MAXC2], in2[MAXC2]; };
    std::vector<Ahead2> prep2;
    const uint64_t h_none = str_hash(std::string_view("\x01None"));
    uint64_t all_novel = 0;
    bool satisfied = false;
    uint64_t n_chunks = 0;
    bool last1_truthy = false, last2_truthy = false, any_read = false;
    Chunk c1, c2;
    // single-end fast path: the next chunk is read by a thread of its own while this one is worked on (the stopping rule
    // may make that one read too many; the reader is used by one thread at a time)
    Chunk ahead_chunk;
    std::thread ahead_reader;
    int ahead_rc = BC_OK;
    auto join_reader = [&]() { if (ahead_reader.joinable()) ahead_reader.join(); };
    for (;;) {
        t_mark = now();
        if (lookahead && n_chunks > 0) {
            join_reader();
            rc = ahead_rc;
            if (rc != BC_OK) break;
            std::swap(c1, ahead_chunk);
        } else {
            if ((rc = next_chunk(r1, n_lib, c1)) != BC_OK) break;
            if (paired && (rc = next_chunk(r2, n_lib, c2)) != BC_OK) break;
        }
        t_read += std::chrono::duration<double, std::milli>(now() - t_mark).count();
        if (c1.n == 0) break;
        if (lookahead) ahead_reader = std::thread([&] { ahead_rc = next_chunk(r1, n_lib, ahead_chunk); });
        t_mark = now();
        ++n_chunks;
        std::vector<int> c_or[2];
        std::vector<int64_t> c_off[2];
        if (!lookahead) {  // (the fast path clears them only for a chunk that takes the loop form)
            c_barcodes.clear();
            c_reads.clear();
        }
        // zip(read1_chunk, read2_chunk if read2_chunk else [None]*len): an EMPTY mate chunk means "None" mates
        const bool have2 = paired && c2.n > 0;
        const uint64_t nz = have2 ? std::min(c1.n, c2.n) : c1.n;
        // Single-end input with the pre-filter: the set look-ups of a read are independent cache misses into tables of
        // several MB.  A few reads ahead of the one being processed, hash the read, sweep its windows through the
        // pre-filter, hash the windows that pass and prefetch every slot the read will touch; the processing below then
        // runs the reference's sequential logic unchanged on warm lines.
        // (id: the window's library id — the forward set's when it is there, else fwd_slots + the reverse set's: one
        // id per distinct string, which is what the reference's per-chunk novel_barcodes is keyed by)
        const uint64_t chunk_tag = (uint64_t)(0xFFFFFFull - std::min<uint64_t>(n_chunks, 0xFFFFFFull)) << 40;
        auto key_of = [&](uint64_t q, int c) { return chunk_tag | (q << 3) | (uint64_t)c; };
        std::atomic<bool> any_overflow{false};
        const bool retain = lookahead && kept_bytes + c1.text.size() <= KEEP_BYTES;
        // what a read contributes before any growing set is consulted — its hash, the windows that ARE library entries
        // (pre-filter, then the two library sets, which never change) — is a pure function of the read: computed for the
        // whole chunk by worker threads; the order-dependent logic below then touches only the reads that hold a
        // library k-mer (the reference's loop does nothing for the others except remember the read: see seen_part)
        const auto t_a0 = now();
        if (lookahead && prep.size() < nz) prep.resize(nz);  // (prepare() writes every field it or anybody reads)
        t_alloc += std::chrono::duration<double, std::milli>(now() - t_a0).count();
        std::vector<std::thread> rem_threads;
        auto prepare = [&](uint64_t q) {
            Ahead& a = prep[q];
            const std::string_view rd = c1.read(q);
            a.h1 = str_hash(rd);
            a.ncand = 0;
            a.overflow = false;
            if ((int64_t)rd.size() < k) return;
            // one pass over the characters: rolling packed value, length of the current ACGT run, both filter levels
            const uint64_t kmask2 = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
            const uint64_t* const small = pf_small.data();
            const uint64_t* const big = pf_bits.data();
            uint64_t v = 0;
            int run = 0;
            const size_t len = rd.size();
            for (size_t e = 0; e < len; ++e) {
                const unsigned c = kCode[(unsigned char)rd[e]];
                v = ((v << 2) | (c & 3u)) & kmask2;
                run = c < 4 ? run + 1 : 0;
                if (run < k) continue;
                const uint32_t hs = Roll::hash_small(v);
                if (!((small[hs >> 6] >> (hs & 63)) & 1ull)) continue;
                const uint32_t hb = Roll::hash(v);
                if (!((big[hb >> 6] >> (hb & 63)) & 1ull)) continue;
                const size_t i = e + 1 - (size_t)k;
                const std::string_view kmer = rd.substr(i, k);
                const uint64_t hk = str_hash(kmer);
                const int64_t idf = fwd.find(hk, kmer), idr = rev.find(hk, kmer);
                const uint8_t in = (uint8_t)((idf >= 0 ? 1 : 0) | (idr >= 0 ? 2 : 0));
                if (!in) continue;  // (a false positive of the pre-filter)
                if (a.ncand == MAXC) { a.overflow = true; any_overflow.store(true, std::memory_order_relaxed); break; }
                const uint32_t id = (uint32_t)(idf >= 0 ? (size_t)idf : fwd_slots + (size_t)idr);
                a.pos[a.ncand] = (uint32_t)i; a.id[a.ncand] = id; a.in[a.ncand] = in;
                // earliest occurrence of this library entry in the chunk (reads ascending, then positions)
                const uint64_t key = key_of(q, a.ncand);
                uint64_t cur = first[id].load(std::memory_order_relaxed);
                while (key < cur && !first[id].compare_exchange_weak(cur, key, std::memory_order_relaxed)) {}
                ++a.ncand;
            }
        };
        auto warm = [&](uint64_t q) {  // (loop form only) pull the slot read q will touch towards the core
            const Ahead& a = prep[q];
            if (a.ncand) vset[a.h1 >> 61].prefetch(a.h1);
        };
        // seen_reads only ever answers "how many distinct reads so far": eight sets, one per range of the hash, filled by
        // as many threads (every thread looks at every read's hash and keeps its own)
        auto remember = [&](unsigned part0, unsigned step) {
            for (uint64_t q = 0; q < nz; ++q) {
                const unsigned part = (unsigned)(prep[q].h1 >> 61);
                if (part % step == part0) seen_part[part].insert_at(prep[q].h1, c1.read(q), !retain);
            }
        };
        auto obs_mark = [&](uint32_t id) { obs_bits[id >> 6] |= 1ull << (id & 63); };
        auto valid_add = [&](uint64_t h1, std::string_view rd) {  // (sequential callers only)
            const char* at = vset[h1 >> 61].insert_at(h1, rd, !retain);
            if (at) valid_views.emplace_back(at, rd.size());
        };
        // ---- mate pairs: candidates of both mates, merged by position
        auto scan_read = [&](std::string_view rd, size_t n_windows, uint32_t* pos, uint8_t* in, int cap) -> int {  // -1: more than cap
            int n = 0;
            if ((int64_t)rd.size() < k || n_windows == 0) return 0;
            const uint64_t kmask2 = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
            const uint64_t* const small = pf_small.data();
            const uint64_t* const big = pf_bits.data();
            uint64_t v = 0;
            int run = 0;
            const size_t len = std::min(rd.size(), n_windows - 1 + (size_t)k);
            for (size_t e = 0; e < len; ++e) {
                const unsigned c = kCode[(unsigned char)rd[e]];
                v = ((v << 2) | (c & 3u)) & kmask2;
                run = c < 4 ? run + 1 : 0;
                if (run < k) continue;
                const uint32_t hs = Roll::hash_small(v);
                if (!((small[hs >> 6] >> (hs & 63)) & 1ull)) continue;
                const uint32_t hb = Roll::hash(v);
                if (!((big[hb >> 6] >> (hb & 63)) & 1ull)) continue;
                const size_t i = e + 1 - (size_t)k;
                const std::string_view kmer = rd.substr(i, k);
                const uint64_t hk = str_hash(kmer);
                const uint8_t b = (uint8_t)((fwd.contains(hk, kmer) ? 1 : 0) | (rev.contains(hk, kmer) ? 2 : 0));
                if (!b) continue;
                if (n == cap) return -1;
                pos[n] = (uint32_t)i; in[n] = b; ++n;
            }
            return n;
        };
        auto prepare2 = [&](uint64_t q) {
            Ahead2& a = prep2[q];
            const std::string_view r1 = c1.read(q);
            const std::string_view r2 = have2 ? c2.read(q) : std::string_view();
            a.h1 = str_hash(r1);
            a.h2 = have2 ? str_hash(r2) : h_none;   // (what seen_reads gets beside read1: the mate, or the "None" of an empty mate chunk)
            a.ncand = 0; a.overflow = false; a.skipped = false;
            if ((int64_t)r1.size() < k) return;
            const size_t n_windows = r1.size() - (size_t)k + 1;   // the loop runs over the FIRST mate's windows
            uint32_t p1[MAXC2], p2[MAXC2];
            uint8_t b1[MAXC2], b2[MAXC2];
            const int n1 = scan_read(r1, n_windows, p1, b1, MAXC2);
            const int n2 = have2 ? scan_read(r2, n_windows, p2, b2, MAXC2) : 0;
            if (n1 < 0 || n2 < 0) { a.overflow = true; return; }
            int i1 = 0, i2 = 0;
            while (i1 < n1 || i2 < n2) {
                if (a.ncand == MAXC2) { a.overflow = true; return; }
                const uint32_t pa = i1 < n1 ? p1[i1] : 0xFFFFFFFFu, pb = i2 < n2 ? p2[i2] : 0xFFFFFFFFu;
                const uint32_t pp = std::min(pa, pb);
                a.pos[a.ncand] = pp;
                a.in1[a.ncand] = pa == pp ? b1[i1++] : (uint8_t)0;
                a.in2[a.ncand] = pb == pp ? b2[i2++] : (uint8_t)0;
                ++a.ncand;
            }
        };
        // what the loop did not skip goes into seen_reads: read1 and the mate (or "None"), by hash partition
        auto remember2 = [&](unsigned part0, unsigned step) {
            for (uint64_t q = 0; q < nz; ++q) {
                const Ahead2& a = prep2[q];
                if (a.skipped) continue;
                const unsigned pa = (unsigned)(a.h1 >> 61), pb = (unsigned)(a.h2 >> 61);
                if (pa % step == part0) seen_part[pa].insert(a.h1, c1.read(q));
                if (pb % step == part0) seen_part[pb].insert(a.h2, have2 ? c2.read(q) : std::string_view("\x01None"));
            }
        };
        if (lookahead2) {
            if (prep2.size() < nz) prep2.resize(nz);
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const uint64_t T = sample_threads(hw, nz);
            const auto t_p0 = now();
            if (T == 1) {
                for (uint64_t q = 0; q < nz; ++q) prepare2(q);
            } else {
                std::vector<std::thread> th;
                const uint64_t per = (nz + T - 1) / T;
                for (uint64_t t = 0; t < T; ++t)
                    th.emplace_back([&, t] { for (uint64_t q = t * per; q < std::min(nz, (t + 1) * per); ++q) prepare2(q); });
                for (auto& x : th) x.join();
            }
            t_prep += std::chrono::duration<double, std::milli>(now() - t_p0).count();
        }
        bool chunk_done = false;  // the fast path has dealt with the whole chunk
        if (lookahead) {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const uint64_t T = sample_threads(hw, nz);
            const auto t_p0 = now();
            if (T == 1) {
                for (uint64_t q = 0; q < nz; ++q) prepare(q);
                t_prep += std::chrono::duration<double, std::milli>(now() - t_p0).count();
                remember(0, 1);
            } else {
                std::vector<std::thread> th;
                const uint64_t per = (nz + T - 1) / T;
                for (uint64_t t = 0; t < T; ++t)
                    th.emplace_back([&, t] { for (uint64_t q = t * per; q < std::min(nz, (t + 1) * per); ++q) prepare(q); });
                for (auto& x : th) x.join();
                th.clear();
                t_prep += std::chrono::duration<double, std::milli>(now() - t_p0).count();
                // (they run beside the order-dependent loop below, which touches none of their sets; joined behind it)
                const unsigned step = T >= 8 ? 8u : T >= 4 ? 4u : 2u;  // (eight parts: a divisor of eight threads take them)
                for (unsigned t = 0; t < step; ++t) rem_threads.emplace_back([&, t, step] { remember(t, step); });  // (step by value: its scope ends before the threads do)
            }
            if (!any_overflow.load()) {
                // ---- the chunk without the order-dependent loop.  The reference counts a library k-mer the FIRST time it
                // meets it in a chunk (novel_barcodes) and nothing else of that loop depends on the order: skipping a
                // read equal to an earlier valid one changes nothing (that read's library k-mers are all in
                // novel_barcodes by then).  first[] holds those first meetings; what is left is to walk them in order.
                chunk_done = true;
                const auto t_c0 = now();
                any_read = nz > 0;
                if (nz) last1_truthy = !c1.read(nz - 1).empty();
                last2_truthy = false;
                const uint64_t Tc = std::min<uint64_t>(T, 16);  // (light work: more threads cost more to start than they save)
                const uint64_t per = (nz + Tc - 1) / Tc;
                std::vector<std::vector<uint64_t>> counted(Tc);  // (q << 3 | candidate) of the counted meetings, ascending
                auto collect = [&](uint64_t t) {
                    std::vector<uint64_t>& out_v = counted[t];
                    out_v.reserve((size_t)per);
                    for (uint64_t q = t * per; q < std::min(nz, (t + 1) * per); ++q) {
                        const Ahead& a = prep[q];
                        for (int c = 0; c < a.ncand; ++c)
                            if (first[a.id[c]].load(std::memory_order_relaxed) == key_of(q, c)) out_v.push_back((q << 3) | (uint64_t)c);
                    }
                };
                if (Tc == 1) collect(0);
                else {
                    std::vector<std::thread> th;
                    for (uint64_t t = 0; t < Tc; ++t) th.emplace_back(collect, t);
                    for (auto& x : th) x.join();
                }
                const auto t_c1 = now();
                t_collect += std::chrono::duration<double, std::milli>(t_c1 - t_c0).count();
                // counters in the order of the meetings (Counter ties resolve to the first inserted key): per chunk only the
                // first appearance of a key and its total matter
                std::vector<std::pair<int, int64_t>> or_seq;
                std::vector<std::pair<int64_t, int64_t>> off_seq;
                std::vector<int32_t> off_at;  // position -> index into off_seq, -1
                int or_at[2] = {-1, -1};
                std::vector<uint64_t> vq;     // the reads with a counted meeting, ascending
                for (uint64_t t = 0; t < Tc; ++t)
                    for (const uint64_t e : counted[t]) {
                        const uint64_t q = e >> 3;
                        const Ahead& a = prep[q];
                        const int c = (int)(e & 7);
                        ++all_novel;  // (c_barcodes.size() of the loop form)
                        obs_mark(a.id[c]);
                        if (vq.empty() || vq.back() != q) vq.push_back(q);
                        const uint32_t pos = a.pos[c];
                        if (pos >= off_at.size()) off_at.resize((size_t)pos + 64, -1);
                        for (int lib = 0; lib < 2; ++lib)
                            if (a.in[c] & (1 << lib)) {
                                div[0]++;
                                if (or_at[lib] < 0) { or_at[lib] = (int)or_seq.size(); or_seq.emplace_back(lib, 0); }
                                or_seq[or_at[lib]].second++;
                                if (off_at[pos] < 0) { off_at[pos] = (int32_t)off_seq.size(); off_seq.emplace_back((int64_t)pos, 0); }
                                off_seq[off_at[pos]].second++;
                            }
                    }
                for (const auto& it : or_seq) orients[0].add(it.first, it.second);
                for (const auto& it : off_seq) offsets[0].add(it.first, it.second);
                const auto t_c2 = now();
                t_count += std::chrono::duration<double, std::milli>(t_c2 - t_c1).count();
                // valid reads: each part of the set takes its own reads (a new one leaves its address), then they join
                // sample1 in the order of the meetings
                std::vector<const char*> new_at(vq.size(), nullptr);
                auto add_valid = [&](unsigned p0, unsigned step) {
                    for (size_t j = 0; j < vq.size(); ++j) {
                        const uint64_t h1 = prep[vq[j]].h1;
                        const unsigned part = (unsigned)(h1 >> 61);
                        if (part % step == p0) new_at[j] = vset[part].insert_at(h1, c1.read(vq[j]), !retain);
                    }
                };
                const unsigned vstep = T >= 8 ? 8u : T >= 4 ? 4u : T >= 2 ? 2u : 1u;
                if (vstep == 1 || vq.size() < 4096) add_valid(0, 1);
                else {
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < vstep; ++t) th.emplace_back(add_valid, t, vstep);
                    for (auto& x : th) x.join();
                }
                for (size_t j = 0; j < vq.size(); ++j)
                    if (new_at[j]) valid_views.emplace_back(new_at[j], c1.read(vq[j]).size());
                t_valid += std::chrono::duration<double, std::milli>(now() - t_c2).count();
            } else {
                c_barcodes.clear();
                c_reads.clear();
                for (uint64_t q = 0; q < std::min<uint64_t>(AHEAD, nz); ++q) warm(q);
            }
        }
        for (uint64_t q = 0; q < nz && !chunk_done; ++q) {
            const std::string_view read1 = c1.read(q);
            const std::string_view read2 = have2 ? c2.read(q) : std::string_view();
            any_read = true;
            last1_truthy = !read1.empty();
            last2_truthy = have2 && !read2.empty();
            if (lookahead) {
                const Ahead& a = prep[q];
                if (q + AHEAD < nz) warm(q + AHEAD);
                const uint64_t h1 = a.h1;
                // (the reference skips a read that equals an earlier valid read of the chunk: that read's library
                // k-mers are all in c_barcodes by then, so going through them again changes nothing — no c_reads here;
                // and it remembers every other read in seen_reads: done above, by the partitions)
                if (a.ncand == 0 && !a.overflow) continue;
                if (!a.overflow) {
                    // the windows that are library entries, in ascending position: exactly the iterations of the
                    // reference's loop that do anything
                    for (int c = 0; c < a.ncand; ++c) {
                        const size_t i = a.pos[c];
                        const std::string_view kmer = read1.substr(i, k);
                        const uint64_t hk = str_hash(kmer);
                        if (c_barcodes.contains(hk, kmer)) continue;
                        for (int lib = 0; lib < 2; ++lib)
                            if (a.in[c] & (1 << lib)) {
                                div[0]++; c_barcodes.insert(hk, kmer); obs_mark(a.id[c]);
                                c_or[0].push_back(lib); c_off[0].push_back((int64_t)i);
                                valid_add(h1, read1);
                            }
                    }
                    continue;
                }
                // more candidate windows than the look-ahead keeps: the plain sweep (the sets are idempotent up to here:
                // nothing of this read has been entered yet except seen_reads)
                Roll roll1(k);
                for (size_t i = 0; i + k <= read1.size(); ++i) {
                    const std::string_view kmer = read1.substr(i, k);
                    if (!roll1.at(read1, i, pf_bits, pf_small)) continue;
                    const uint64_t hk = str_hash(kmer);
                    if (c_barcodes.contains(hk, kmer)) continue;
                    const int64_t idf = fwd.find(hk, kmer), idr = rev.find(hk, kmer);
                    for (int lib = 0; lib < 2; ++lib)
                        if ((lib == 0 ? idf : idr) >= 0) {
                            div[0]++; c_barcodes.insert(hk, kmer);
                            obs_mark((uint32_t)(idf >= 0 ? (size_t)idf : fwd_slots + (size_t)idr));
                            c_or[0].push_back(lib); c_off[0].push_back((int64_t)i);
                            valid_add(h1, read1);
                            c_reads.insert(h1, read1);
                        }
                }
                continue;
            }
            const bool la2 = lookahead2;
            const uint64_t h1 = la2 ? prep2[q].h1 : str_hash(read1), h2 = last2_truthy ? (la2 ? prep2[q].h2 : str_hash(read2)) : 0;
            if (c_reads.contains(h1, read1) || (last2_truthy && c_reads.contains(h2, read2))) {
                if (la2) prep2[q].skipped = true;
                continue;
            }
            if (!la2) {
                seen_reads.insert(h1, read1);
                if (paired) seen_reads.insert(have2 ? read2 : std::string_view("\x01None"));
            }
            if ((int64_t)read1.size() < k) continue;
            if (la2 && !prep2[q].overflow) {
                // the positions where either mate's window is a library entry, ascending: the iterations of the
                // reference's loop that do anything (a window already met in this chunk skips its mate's, as there)
                const Ahead2& a = prep2[q];
                for (int c = 0; c < a.ncand; ++c) {
                    const size_t i = a.pos[c];
                    const std::string_view kmer = read1.substr(i, k);
                    if (a.in1[c]) {
                        const uint64_t hk = str_hash(kmer);
                        if (c_barcodes.contains(hk, kmer)) continue;
                        for (int lib = 0; lib < 2; ++lib)
                            if (a.in1[c] & (1 << lib)) {
                                div[0]++; c_barcodes.insert(hk, kmer); observed.insert(hk, kmer);
                                c_or[0].push_back(lib); c_off[0].push_back((int64_t)i);
                                if (valid[0].insert(h1, read1)) valid_store[0].emplace_back(read1);
                                c_reads.insert(h1, read1);
                            }
                    }
                    if (a.in2[c]) {
                        const std::string_view kmer2 = read2.substr(i, k);
                        const uint64_t hk2 = str_hash(kmer2);
                        if (c_barcodes.contains(hk2, kmer2)) continue;
                        for (int lib = 0; lib < 2; ++lib)
                            if (a.in2[c] & (1 << lib)) {
                                div[1]++; c_barcodes.insert(hk2, kmer2); observed.insert(kmer);  // sic: the first mate's k-mer
                                c_or[1].push_back(lib); c_off[1].push_back((int64_t)i);
                                if (valid[1].insert(h2, read2)) valid_store[1].emplace_back(read2);
                                c_reads.insert(h2, read2);
                            }
                    }
                }
                continue;
            }
            Roll roll1(k), roll2(k);
            for (size_t i = 0; i + k <= read1.size(); ++i) {
                const std::string_view kmer = read1.substr(i, k);
                // pre-filter: a window that cannot be a library entry (or its reverse complement) takes none of the
                // set look-ups below — they would all miss (c_barcodes only ever holds k-mers found in a set)
                const bool maybe1 = !prefilter || roll1.at(read1, i, pf_bits, pf_small);
                const uint64_t hk = maybe1 ? str_hash(kmer) : 0;
                if (maybe1 && c_barcodes.contains(hk, kmer)) continue;
                if (maybe1) for (int lib = 0; lib < 2; ++lib)
                    if ((lib == 0 ? fwd : rev).contains(hk, kmer)) {
                        div[0]++; c_barcodes.insert(hk, kmer); observed.insert(hk, kmer);
                        c_or[0].push_back(lib); c_off[0].push_back((int64_t)i);
                        if (valid[0].insert(h1, read1)) valid_store[0].emplace_back(read1);
                        c_reads.insert(h1, read1);
                    }
                if (paired && have2) {
                    const std::string_view kmer2 = i < read2.size() ? read2.substr(i, k) : std::string_view();
                    if (prefilter && (kmer2.size() != (size_t)k || !roll2.at(read2, i, pf_bits, pf_small))) continue;  // same argument
                    const uint64_t hk2 = str_hash(kmer2);
                    if (c_barcodes.contains(hk2, kmer2)) continue;
                    for (int lib = 0; lib < 2; ++lib)
                        if (kmer2.size() == (size_t)k && (lib == 0 ? fwd : rev).contains(hk2, kmer2)) {
                            div[1]++; c_barcodes.insert(hk2, kmer2); observed.insert(kmer);  // sic: the first mate's k-mer
                            c_or[1].push_back(lib); c_off[1].push_back((int64_t)i);
                            if (valid[1].insert(h2, read2)) valid_store[1].emplace_back(read2);
                            c_reads.insert(h2, read2);
                        }
                }
            }
        }
        if (lookahead2) {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const unsigned step = (hw >= 8 && nz >= 8192) ? 8u : (hw >= 4 && nz >= 8192) ? 4u : 1u;
            if (step == 1) remember2(0, 1);
            else for (unsigned t = 0; t < step; ++t) rem_threads.emplace_back([&, t, step] { remember2(t, step); });
        }
        {
            const auto t_r0 = now();
            for (auto& x : rem_threads) x.join();
            t_rem += std::chrono::duration<double, std::milli>(now() - t_r0).count();
        }
        if (retain) {  // the sets hold views into this chunk's text: it stays (next_chunk allocates a new buffer)
            kept_bytes += c1.text.size();
            kept.push_back(std::move(c1.text));
            c1.text = TextBuf();
        }
        t_sweep += std::chrono::duration<double, std::milli>(now() - t_mark).count();
        if (!chunk_done) all_novel += c_barcodes.size();
        for (int s = 0; s < 2; ++s) {
            for (int o : c_or[s]) orients[s].add(o);
            for (int64_t o : c_off[s]) offsets[s].add(o);
        }
        const auto top1 = offsets[0].most_common(2), top2 = offsets[1].most_common(2);
        const int64_t need = 5 * (int64_t)n_lib;
        if (paired ? (div[0] >= need && div[1] >= need) : (div[0] >= need)) satisfied = true;
        if ((int64_t)seen_count() >= need && all_novel > 0) satisfied = true;
        if ((int64_t)all_novel >= need) satisfied = true;
        if (satisfied) {
            auto dominant = [](const std::vector<std::pair<int64_t, int64_t>>& t) {
                return t.size() > 1 && t[0].second >= 2 * t[1].second;
            };
            const bool done = paired ? ((top1.size() == 1 && top2.size() == 1) || (dominant(top1) && dominant(top2)))
                                     : (top1.size() == 1 || dominant(top1));
            if (done) break;
        }
    }
    join_reader();
    bc_reader_close(r1);
    if (r2) bc_reader_close(r2);
    (void)t_enter; (void)t_setup;
    if (prof) fprintf(stderr, "bc_sample_data: %llu chunks of %llu reads: reading %.1f ms, sweep %.1f ms (of it: per-read work on threads %.1f, "
                      "waiting for the distinct-read count %.1f; first meetings collected %.1f, counters %.1f, valid reads %.1f, "
                      "look-ahead array %.1f), sets + filter before the first chunk %.1f ms\n",
                      (unsigned long long)n_chunks, (unsigned long long)n_lib, t_read, t_sweep, t_prep, t_rem, t_collect, t_count,
                      t_valid, t_alloc, t_setup);
    if (rc != BC_OK) return rc;
    if (!any_read) { set_error("Unable to determine orientation of reads. Please check the input files."); return BC_EINVAL; }
    // most_common(1)[0][0] on an empty Counter is an IndexError in the reference (logged by main)
    if ((last1_truthy && orients[0].empty()) || (last2_truthy && orients[1].empty())) {
        set_error("list index out of range");
        return BC_EINVAL;
    }
    const int o1 = last1_truthy ? orients[0].most_common(1)[0].first : -1;
    const int o2 = last2_truthy ? orients[1].most_common(1)[0].first : -1;
    const int64_t f1 = last1_truthy ? offsets[0].most_common(1)[0].first : -1;
    const int64_t f2 = last2_truthy ? offsets[1].most_common(1)[0].first : -1;
    bc_sample* s = new bc_sample();
    s->n_novel = (int64_t)seen_count();
    if (lookahead) {
        uint64_t n_obs = 0;
        for (const uint64_t w : obs_bits) n_obs += (uint64_t)__builtin_popcountll(w);
        s->n_observed = n_obs;
    } else {
        s->n_observed = observed.size();
    }
    s->n_chunks = n_chunks;
    s->satisfied = satisfied;
    // sample1 / sample2 as views: into the kept chunk texts and the sets' arenas (single-end fast path), or into the
    // copies the other paths made
    std::vector<std::string_view> views[2];
    if (lookahead) {
        views[0] = std::move(valid_views);
        s->texts = std::move(kept);
        s->arenas = std::move(vset);
    } else {
        for (int w = 0; w < 2; ++w) {
            s->strs[w] = std::move(valid_store[w]);
            views[w].reserve(s->strs[w].size());
            for (const std::string& x : s->strs[w]) views[w].emplace_back(x);
        }
    }
    if (o1 == 0 || o2 == 1) {
        s->need_swap = 0; s->off1 = f1; s->off2 = f2;
        s->valid1 = std::move(views[0]); s->valid2 = std::move(views[1]);
    } else if (o1 == 1 || o2 == 0) {
        s->need_swap = 1; s->off1 = f2; s->off2 = f1;
        s->valid1 = std::move(views[1]); s->valid2 = std::move(views[0]);
    } else {
        delete s;
        set_error("Unable to determine orientation of reads. Please check the input files.");
        return BC_EINVAL;
    }
    *out = s;
    return BC_OK;
}

int bc_sample_info(const bc_sample* s, int64_t* n_novel, int64_t* off1, int64_t* off2, uint64_t* n_valid1,
                   uint64_t* n_valid2, uint64_t* n_observed, int* need_swap, uint64_t* n_chunks, int* satisfied) {
    if (!s) { set_error("bc_sample_info: null"); return BC_EINVAL; }
    if (n_novel) *n_novel = s->n_novel;
    if (off1) *off1 = s->off1;
    if (off2) *off2 = s->off2;
    if (n_valid1) *n_valid1 = s->valid1.size();
    if (n_valid2) *n_valid2 = s->valid2.size();
    if (n_observed) *n_observed = s->n_observed;
    if (need_swap) *need_swap = s->need_swap;
    if (n_chunks) *n_chunks = s->n_chunks;
    if (satisfied) *satisfied = s->satisfied;
    return BC_OK;
}

// find_flanks over sample `which` (0 = sample1, 1 = sample2): lengths -1 = None
int bc_sample_flanks(const bc_sample* s, int which, int64_t start, int bc_len, int max_flank, char* L_out,
                     int* L_len, char* R_out, int* R_len) {
    if (!s || !L_out || !R_out || !L_len || !R_len || max_flank < 1 || max_flank > 64 || which < 0 || which > 1) {
        set_error("bc_sample_flanks: bad argument");
        return BC_EINVAL;
    }
    const std::vector<std::string_view>& reads = which ? s->valid2 : s->valid1;
    const auto t_begin = std::chrono::steady_clock::now();
    // Counter of the flank candidates: a candidate is at most max_flank <= 64 characters... in practice ten, so up to
    // 15 characters travel as one 128-bit key (characters + length) through an open-addressing table that remembers
    // insertion order (Counter / max() resolve ties towards the first inserted key); longer ones take the string map
    struct Key { uint64_t a, b; };
    struct Entry { Key k; int64_t count; uint32_t order; bool used; };
    struct FlankCounter {
        std::vector<Entry> tab;
        size_t mask, n = 0;
        OrderedCounter<std::string> big;  // candidates of 16+ characters (max_flank > 15 only)
        FlankCounter() : tab(1u << 12, Entry{{0, 0}, 0, 0, false}), mask((1u << 12) - 1) {}
        static Key pack(std::string_view v) {
            Key k{0, 0};
            memcpy(&k.a, v.data(), std::min<size_t>(v.size(), 8));
            if (v.size() > 8) memcpy(&k.b, v.data() + 8, v.size() - 8);
            k.b |= (uint64_t)v.size() << 56;  // at most 15 characters: byte 7 of b is free
            return k;
        }
        void add(std::string_view v) {
            if (v.size() > 15) { big.add(std::string(v)); return; }
            add_packed(pack(v), 1);
        }
        void add_packed(const Key k, int64_t times) {
            size_t i = (size_t)((k.a * 0x9E3779B97F4A7C15ull) ^ (k.b * 0xD6E8FEB86659FD93ull)) >> 20 & mask;
            for (;; i = (i + 1) & mask) {
                Entry& e = tab[i];
                if (!e.used) { e = Entry{k, times, (uint32_t)n++, true}; break; }
                if (e.k.a == k.a && e.k.b == k.b) { e.count += times; return; }
            }
            if (n * 2 > tab.size()) {
                std::vector<Entry> old;
                old.swap(tab);
                tab.assign(old.size() * 2, Entry{{0, 0}, 0, 0, false});
                mask = tab.size() - 1;
                for (const Entry& e : old)
                    if (e.used) {
                        size_t j = (size_t)((e.k.a * 0x9E3779B97F4A7C15ull) ^ (e.k.b * 0xD6E8FEB86659FD93ull)) >> 20 & mask;
                        while (tab[j].used) j = (j + 1) & mask;
                        tab[j] = e;
                    }
            }
        }
        // items in insertion order: (text, count)
        std::vector<std::pair<std::string, int64_t>> items() const {
            std::vector<const Entry*> v;
            for (const Entry& e : tab) if (e.used) v.push_back(&e);
            std::sort(v.begin(), v.end(), [](const Entry* x, const Entry* y) { return x->order < y->order; });
            std::vector<std::pair<std::string, int64_t>> out;
            for (const Entry* e : v) {
                const size_t len = (size_t)(e->k.b >> 56);
                char buf[16];
                memcpy(buf, &e->k.a, 8);
                const uint64_t b = e->k.b & 0x00FFFFFFFFFFFFFFull;
                memcpy(buf + 8, &b, 8);
                out.emplace_back(std::string(buf, len), e->count);
            }
            return out;
        }
    };
    FlankCounter tally[2];
    auto count_range = [&](size_t r0, size_t r1, FlankCounter (&tl)[2]) {
        for (size_t ri = r0; ri < r1; ++ri) {
            const std::string_view read = reads[ri];
            const int64_t n = (int64_t)read.size();
            // python slices: read[start-max_flank:start] (or read[0:start]) and read[start+k : start+k+max_flank]
            int64_t a = start - max_flank >= 0 ? start - max_flank : 0, b = std::min<int64_t>(std::max<int64_t>(start, 0), n);
            a = std::min(a, n);
            std::string_view left = b > a ? read.substr((size_t)a, (size_t)(b - a)) : std::string_view();
            int64_t c = std::min<int64_t>(std::max<int64_t>(start + bc_len, 0), n), d = std::min<int64_t>(std::max<int64_t>(start + bc_len + max_flank, 0), n);
            std::string_view right = d > c ? read.substr((size_t)c, (size_t)(d - c)) : std::string_view();
            for (size_t m = left.size(); m > 0; --m) tl[0].add(left.substr(left.size() - m));
            for (size_t m = right.size(); m > 0; --m) tl[1].add(right.substr(0, m));
        }
    };
    // contiguous ranges of the sample on worker threads, their counters merged in range order: a key's place in the
    // insertion order is where the earliest range that holds it put it — the order one sequential pass gives
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, hw), reads.size() / 8192 + 1));
    if (T == 1) {
        count_range(0, reads.size(), tally);
    } else {
        std::vector<FlankCounter> part(2 * T);
        std::vector<std::thread> th;
        const size_t per = (reads.size() + T - 1) / T;
        for (size_t t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                FlankCounter tl[2];
                count_range(std::min(reads.size(), t * per), std::min(reads.size(), (t + 1) * per), tl);
                part[2 * t] = std::move(tl[0]);
                part[2 * t + 1] = std::move(tl[1]);
            });
        for (auto& x : th) x.join();
        for (size_t t = 0; t < T; ++t)
            for (int side = 0; side < 2; ++side) {
                const FlankCounter& src = part[2 * t + side];
                std::vector<const Entry*> v;
                for (const Entry& e : src.tab) if (e.used) v.push_back(&e);
                std::sort(v.begin(), v.end(), [](const Entry* x, const Entry* y) { return x->order < y->order; });
                for (const Entry* e : v) tally[side].add_packed(e->k, e->count);
                for (const auto& it : src.big.items) tally[side].big.add(it.first, it.second);
            }
    }
    auto best = [&](const FlankCounter& fc, char* out_buf, int* out_len) {
        // (with max_flank <= 15 every candidate is in the packed table; the string map only fills beyond that, and the
        // two never hold candidates of the same length, so per-length maxima need no merged order)
        std::vector<std::pair<std::string, int64_t>> all = fc.items();
        for (const auto& it : fc.big.items) all.push_back(it);
        const std::pair<std::string, int64_t>* chosen = nullptr;
        for (int n = max_flank; n > 0; --n) {
            const std::pair<std::string, int64_t>* top = nullptr;
            for (const auto& it : all)
                if ((int)it.first.size() == n && (!top || it.second > top->second)) top = &it;  // max(): first maximum
            if (!top) continue;
            if (!chosen || top->second > 3 * chosen->second) chosen = top;
        }
        if (!chosen) { *out_len = -1; out_buf[0] = 0; return; }
        *out_len = (int)chosen->first.size();
        memcpy(out_buf, chosen->first.data(), chosen->first.size());
        out_buf[chosen->first.size()] = 0;
    };
    best(tally[0], L_out, L_len);
    best(tally[1], R_out, R_len);
    if (getenv("BC_SAMPLE_PROF"))
        fprintf(stderr, "bc_sample_flanks: %zu reads, %zu threads, %.1f ms\n", reads.size(), T,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return BC_OK;
}

int bc_sample_free(bc_sample* s) {
    delete s;
    return BC_OK;
}

}  // extern "C"
