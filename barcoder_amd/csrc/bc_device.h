// bc_device.h — device-side library object and launch helpers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <list>
#include <mutex>
#include <vector>

#include "bc_common.h"

#define BC_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            bc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                    \
            return BC_EHIP;                                                             \
        }                                                                               \
    } while (0)

namespace bc {

constexpr int MAX_SEG = 4;      // max_mismatch <= 3 -> up to 4 pigeonhole segments
constexpr int BLOCK = 256;      // 4 waves of 64
constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 resident blocks: grid-stride beyond

// hash of a packed k-mer, identical on host (table build) and device (lookup)
BC_HD static inline uint32_t hash32(uint64_t key) {
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu);
    x ^= x >> 15;
    return x * 0x9E3779B1u;
}

// One pigeonhole seed index (for one value of max_mismatch): the k-mer is cut
// into nseg = d+1 disjoint segments; the first seed_len bases of segment s key a
// CSR bucket table.  A guide within d substitutions of a read agrees with it
// exactly on at least one segment, hence on that segment's seed.
struct SeedParams {
    int nseg;
    int start[MAX_SEG];  // first base of the seed
    int slen[MAX_SEG];   // seed length in bases
    const uint32_t* off[MAX_SEG];   // [4^slen + 1] bucket offsets
    const uint64_t* keys[MAX_SEG];  // [G] guides grouped by bucket, guide index ascending
    const uint32_t* gidx[MAX_SEG];  // [G] guide index of each entry
};

// Slab form of the same index (k_match_slab): every bucket is a fixed-size
// slab of 8-byte entries {rest-of-key, guide index} at a computable address, so
// one bucket visit is one 64/128-byte L2 request shared by LPR lanes and needs
// neither an offset lookup nor a separate index lookup.  The last entry of a
// slab is a link {next slab, LINK} to an overflow slab appended after the
// primaries.  Requires 2*(k - slen) <= 32 so that the rest fits one dword.
// Entry = {rest, giw}: giw = guide index (22 bits) | isolation class << 22.  With d = the table's max_mismatch and nn =
// distance to the guide's nearest other guide (k_nn_dist), the class c = the number of e in 1..d with nn > 2e, i.e.
// the largest e <= d with nn > 2e.  A read whose best guide g lies at distance e has no second guide within e when
// nn(g) > 2e (triangle inequality), so `e <= c` proves n_best = 1 without looking any further — one compare on the
// key (distance << 24 | giw); c == d says "isolated at the table's full budget".  (Rounds 1-3 kept two flag bits,
// nn > 2d and nn > 2(d-1), and picked one by the distance: eight instructions per test in the wave matcher.)
constexpr uint32_t SLAB_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t SLAB_LINK = 0xFFFFFFFEu;
constexpr int SLAB_GIDX_BITS = 22;
constexpr uint32_t SLAB_GIDX_MASK = (1u << SLAB_GIDX_BITS) - 1u;
constexpr int SLAB_CLASS_SHIFT = 22;
struct SlabTable {
    const uint2* slabs;
    uint32_t start, slen, half;   // seed position / length; half = bit distance of the rest's two planes inside an
                                  // 8-byte entry's first word (always 16) / k - slen for 16-byte entries
    uint32_t log2S;               // entries per slab (incl. the link entry)
    uint32_t empty;               // index of an all-empty slab appended after the overflow slabs
    uint32_t wide;                // 1: 16-byte entries {rest lo plane, giw, rest hi plane, 0} (k - seed > 16 bases)
    uint32_t dmask[MAX_SEG];      // dmask[t]: seed t's bases in rest coordinates (t < this table)
};
// Neighbour lists: for a guide with another guide within NBR_RADIUS substitutions, the (at most NBR_CAP) guides
// that close, as {key lo, key hi, index, 1}.  Any guide tying with g at distance e <= NBR_RADIUS/2 from a read is
// within 2e of g, hence in g's list: ties are resolved with one small lookup instead of a full re-evaluation.
constexpr int NBR_CAP = 8;
constexpr int NBR_RADIUS = 4;
constexpr uint32_t NBR_NONE = 0xFFFFFFFFu;  // no guide within NBR_RADIUS
constexpr uint32_t NBR_OVER = 0xFFFFFFFEu;  // more than NBR_CAP of them: take the exact path
// Table-2 filter of the d = 2 index (nseg = 3).  A read reaches table 2 only when nothing within distance 2 has a
// clean segment 0 or 1; a guide that can still match has exactly one mismatch in segment 0 and one in segment 1, so it
// agrees with the read on one HALF of each of those segments.  Per table-2 bucket one 64-byte line holds four
// 128-bit sets — the (hashed) values the bucket's guides show in the halves 0a, 0b, 1a, 1b; a read whose own values
// miss (0a and 0b) or (1a and 1b) has no candidate in the bucket and need not fetch its slabs at all.
BC_HD static inline uint32_t flt_hash7(uint32_t v) { return (v * 0x9E3779B1u) >> 25; }
// Second form of tables 0 and 1, read by k_match_wq alone (8-byte entries of k-mers whose rest leaves three spare bits per
// plane, 64-byte slabs): EIGHT entries per slab and no link slot.  The low byte of an entry's index word is spare
// (giw8 = giw << 8 | byte; the key comes out of v_alignbyte_b32(distance, giw8, 1) — the same one instruction as
// distance << 24 | giw); slots 6 and 7 keep the two bytes of link16 = 1 + (number of the bucket's next slab - buckets),
// 0 = none; overflow slabs have the same form.  An empty slot's rest has the three spare bits of both planes set: at
// least three "mismatches" against any read, so it never comes within a budget of two.  A size-biased 27 % of the
// bucket visits continue into an overflow slab instead of 41 % with seven entries + link, and there are fewer overflow
// slabs to keep in the L2 (round 2's FAST01 experiment got the eighth entry by putting the bucket's next entries at an
// implied address in a sparse second region: requests 1.79 -> 1.72 per read, but the region cost L2 capacity and time
// did not move; here the form needs no extra footprint).  The exact path and the quad kernels keep reading t[].
struct EightTables {
    const char* base[2];   // [buckets primaries | overflow slabs | one all-empty slab]
    uint32_t nb[2];        // buckets = number of the first overflow slab
    uint32_t empty[2];     // number of the all-empty slab
    uint32_t ok;           // both tables built
};
constexpr uint32_t E8_EMPTY_REST = 0xE000E000u;
// ... and its index word: class 0 (a key at distance >= 3 with class 0 is neither in range nor "settled"), index all ones
constexpr uint32_t E8_EMPTY_GIW8 = SLAB_GIDX_MASK << 8;
struct SlabParams {
    int nseg;
    SlabTable t[MAX_SEG];
    EightTables e8;
    const uint32_t* nbr_slot;  // [G] list number, NBR_NONE or NBR_OVER (NULL = no lists)
    const uint4* nbr;          // [lists][NBR_CAP]
    const uint32_t* flt;       // [table-2 buckets][16] filter lines, NULL = none
    uint32_t f_pos[4], f_len[4];  // first base / length of the halves 0a, 0b, 1a, 1b
};

// Pair index of the 3-mismatch matcher (bc_match_pairs.hip): the k-mer in five pieces, one table per pair (a < b) of
// pieces, keyed on piece a's first a_len bases and piece b's first b_len bases.  A slab is 64 bytes: fifteen 4-byte
// rests (the other k - a_len - b_len bases: low plane in bits 0-15, high plane in bits 16-31) and a tail word
// {next slab << 4 | entries in this slab}; the guide index of slot (slab, e) sits in a parallel array that only a
// candidate within the budget ever touches.  (Round 2: 8-byte {rest, guide} entries, seven per slab + link: a quarter
// of the bucket visits chained, 12.7 requests per read instead of ten, 13 MB of tables instead of 10.5.)
struct PairTable {
    const uint32_t* rests;        // [slabs][16]: bucket = slab number; overflow slabs behind the primaries
    const uint32_t* gidx;         // [slabs][16]: guide index of each slot (slot 15 unused)
    uint32_t a_start, a_len, b_start, b_len;   // key fields, in bases
    uint32_t m_low, m_mid, m_high;  // rest of a plane p = (p & m_low) | ((p >> a_len) & m_mid) | ((p >> (a_len + b_len)) & m_high)
    uint32_t must_clean;          // rest positions of what is left of pieces a and b: a counted guide is clean there
    uint32_t other[3], n_other;   // rest positions of each piece c < b, c != a: a counted guide is dirty in every one
};
struct PairParams {
    int ntab;
    PairTable t[10];
};

// Tuning / experiment knobs, read from the environment ONCE when a library is created (never per launch).
// Defaults are the measured best; the knob-coverage tests check that none of them changes a result.
struct Knobs {
    int lane = 1;         // BC_LANE=0: quad kernels (k_match_fused) instead of the wave kernel (k_match_wq)
    int phase2_sm = 1;    // BC_PHASE2_SM=0: quad kernels without the state machine
    int single_slab = 0;  // BC_SINGLE_SLAB=1: the one-kernel slab form (k_match_slab) for every d
    int slab_lpr = 4;     // BC_SLAB_LPR=8: eight lanes per read in k_match_slab
    int diag = 0;         // BC_DIAG: bit mask that switches stages of k_match_wq off (timing experiments; results are wrong)
    int tally = 0;        // BC_TALLY: 0 auto, 1 global atomics, 2 k_hist sweeps, 3 range-partitioned records
    int phase_prof = 0;   // BC_PHASE_PROF=1: per-stage cycle sums of the wave kernel on stderr
    int host_pin = 1;     // BC_HOST_PIN=0: bc_match_host does not pin the caller's buffers
    int prefetch = 8;     // BC_PREFETCH: read-stream prefetch distance of k_match_wq in units of 128 tiles (0 = none)
    int pair_group_kb = 3300;  // BC_PAIR_GROUP_KB: pair tables taken together in one pass of the cascade's second stage
                          // (their slabs stay in an XCD's 4 MiB L2); 0: the one-kernel form over all ten tables
    int cascade2_min_g = -1;   // BC_CASCADE2_MIN_G: d = 2 as d <= 1 pass + pair index from that many guides on (0: only where
                          // the wave matcher cannot serve d = 2; -1, the default: by the load of its buckets, see
                          // launch_match — 3.6e5 20-mers, 6.4e4 16-mers, never for 18- / 21- / 24-mers)
    int pair_bucket_pct = 800; // BC_PAIR_BUCKET_PCT: the pair tables' key grows while a bucket would hold more than this many guides / 100
    int pair_four = 1;         // BC_PAIR_FOUR=0: budgets <= 2 walk all ten pair tables instead of the four that cover three clean pieces
    int pair_grid = 16384;     // BC_PAIR_GRID: workgroups of a second-stage pass at most (2 048 = 8 per CU fill the chip; more balance the tail: 226 -> 210 us)
    int pair_chunk_log2 = 27;  // BC_PAIR_CHUNK_LOG2: reads per launch of a cascade in its list form (workspace: 32 bytes per read)
    int pair_pass_min = 65536; // BC_PAIR_PASS_MIN_READS: launches below that take the one-kernel form (tests: 0)
    int e8 = 1;           // BC_E8=0: k_match_wq on the classic tables 0/1 (seven entries + link) instead of the eight-entry form
    int cascade = 1;      // BC_CASCADE=0: d = 3 by the four-segment index over every read (round 1's path); 2: the pair
                          // index over every read (no d <= 2 pass first); 1: cascade
};

// scratch of the record tally (k_hist_part), one per stream that launches on a library
struct TallyWs {
    void* stream = nullptr;
    uint16_t* rec = nullptr;       // [ranges][tiles_cap][64] records
    uint32_t* ovf_list = nullptr;  // [tiles_cap]
    uint32_t* ovf_count = nullptr; // self-resetting (k_first_refine)
    uint32_t* first_tile = nullptr;  // [G], 0xFFFFFFFF between launches
    uint64_t tiles_cap = 0;
    // second stage of a cascade (bc_match_pairs.hip): the reads the first stage left without a hit, as a list
    // (tile t of 256 reads owns entries [256 t, 256 t + casc_cnt[t]))
    uint64_t* casc_read = nullptr;   // [256 casc_tiles] the read (masked)
    uint32_t* casc_pos = nullptr;    // [256 casc_tiles] its position
    uint64_t* casc_state = nullptr;  // [256 casc_tiles] best so far between table groups {distance, ties, lowest index}
    uint32_t* casc_cnt = nullptr;    // [casc_tiles]
    uint64_t casc_tiles = 0;
    // ... and packed for the passes over the table groups: entry j of casc_count[0]
    uint32_t* casc_off = nullptr;    // [casc_tiles] exclusive prefix sum of casc_cnt
    uint64_t* casc_dread = nullptr;  // [256 casc_tiles]
    uint32_t* casc_dpos = nullptr;   // [256 casc_tiles]
    unsigned long long* casc_count = nullptr;
    void* casc_scan_tmp = nullptr;   // rocprim scratch of the prefix sum
    size_t casc_scan_bytes = 0;
};

}  // namespace bc

struct bc_lib {
    int device = 0;
    uint32_t G = 0;
    int k = 0;
    uint64_t kmask = 0;
    std::vector<uint64_t> h_guides;
    std::vector<uint8_t> h_nn;     // distance to the nearest other guide (empty = not computed)
    uint32_t* d_nbr_slot = nullptr;  // neighbour lists (see SlabParams)
    uint4* d_nbr = nullptr;
    uint64_t* d_guides = nullptr;  // [G] original order (also the brute-force tile source)
    // barcodes of more than 32 nt (bc_lib_attach_text): the guides above are 64-bit hashes of the texts, these the texts
    char* d_text = nullptr;        // [G][text_k]
    int text_k = 0;
    // exact-match open-addressing table
    int log_slots = 0;
    bool wide = false;             // true: 32-bit index slots + verify against d_guides
    uint64_t* d_slots64 = nullptr;
    uint32_t* d_slots32 = nullptr;
    // seed indices, entry d for max_mismatch = d
    bc::SeedParams seed[bc::MAX_SEG];
    bc::SlabParams slab[bc::MAX_SEG] = {};
    bool slab_ok[bc::MAX_SEG] = {false, false, false, false};
    bool slab_wide[bc::MAX_SEG] = {false, false, false, false};  // wide entries: only k_match_wq reads them
    const bc::SlabParams* d_slab[bc::MAX_SEG] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint32_t> pair_slabs;          // slabs of each pair table (64 bytes each): the second stage groups by size
    const bc::PairParams* d_pairs = nullptr;   // 3-mismatch pair index (NULL: not built yet, or k outside 8..25)
    // The same tables for budgets <= 2: at most two substitutions leave THREE of the five pieces clean, and three pieces
    // always hold both of {0, 1} or two of {2, 3, 4} — the four tables (0,1) (2,3) (2,4) (3,4) find every such guide
    // (a covering of the 3-subsets by pairs: the complement of a triangle-free K(2,3)); descriptors with their own
    // count-once rule (bc_match_pairs.hip, build_pair_index), slabs shared with d_pairs
    const bc::PairParams* d_pairs2 = nullptr;
    std::vector<uint32_t> pair_slabs2;
    bool pairs_tried = false;                  // under ws_mutex
    bc::Knobs knobs;
    // geometry of the record tally (ranges, records per line), fixed by the library size: computed once at creation
    uint32_t part_nr = 0, part_R = 0, part_magic = 0, part_cap_log2 = 0;
    bool part_ok = false;
    std::list<bc::TallyWs> ws;     // tally / cascade workspaces by stream (list: pointers stay valid as entries come and go)
    std::mutex ws_mutex;
    std::atomic<const char*> last_kernel{""};  // name of the match kernel launched last (bc_lib_last_kernel)
    void* host_slots = nullptr;    // HostSlot[3]: staging of bc_match_host, kept between calls
    std::mutex host_mutex;
    std::vector<void*> owned;      // every device allocation, for destroy
};
