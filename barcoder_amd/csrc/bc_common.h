// bc_common.h — shared definitions for libbarcoder_hip (host + device).
//
// Encoding ("plane form"): a k-mer (k <= 32) is one uint64_t whose low 32 bits
// hold the LOW code bit of every base and whose high 32 bits hold the HIGH code
// bit: base i (0 = leftmost) -> bit i and bit 32+i.  Codes A=0 C=1 G=2 T=3, so
// complement = flip both bits.  Two k-mers differ at base i iff bit i of
//     mm = (lo_a ^ lo_b) | (hi_a ^ hi_b)
// is set: the Hamming distance is one popcount of a 32-bit word, and `mm` is the
// mismatch-position mask the hit record needs (targets.py:184-190 get_diff,
// mismatch.py:29-32).  A packed genome is the same format: word w is the 32-mer
// at position 32*w.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/barcoder.h"

namespace bc {

void set_error(const char* fmt, ...);

static inline uint64_t kmer_mask(int k) {
    uint32_t m = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    return (uint64_t)m | ((uint64_t)m << 32);
}

// code of an ASCII base, 4 = not ACGT (case-sensitive, like the reference's
// string compares: heuristicount.py:449-462).
static inline int base_code(unsigned char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}
static inline int base_code_nocase(unsigned char c) { return base_code((unsigned char)(c & 0xDF)); }

static inline uint32_t bitrev32(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}

// reverse complement of a packed k-mer
static inline uint64_t revcomp_word(uint64_t w, int k) {
    uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    uint32_t m = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    lo = (bitrev32(~lo) >> (32 - k)) & m;
    hi = (bitrev32(~hi) >> (32 - k)) & m;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ---- splitmix64 synthetic streams (mirrored in synth.py) ------
#if defined(__HIPCC__)
#define BC_HD __host__ __device__
#else
#define BC_HD
#endif

BC_HD static inline uint64_t sm64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// j-th draw of item i of stream `seed`
BC_HD static inline uint64_t sm_draw(uint64_t seed, uint64_t i, uint32_t j) {
    return sm64(seed ^ sm64(i * 8ULL + j));
}

// 64-bit hash of a text of n characters, character i = at(i) — what stands for a barcode of more than 32 nt in the
// exact table.  Eight characters per step, assembled little-endian: the same value on the host (bc_hash_kmers /
// bc_hash_windows) and on the device (k_lines_to_windows), whatever orientation `at` reads the text in.
template <typename At>
BC_HD static inline uint64_t text_hash64_at(size_t n, At at) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xD6E8FEB86659FD93ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w = 0;
        for (int b = 0; b < 8; ++b) w |= (uint64_t)(unsigned char)at(i + (size_t)b) << (8 * b);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
    }
    if (i < n) {
        uint64_t w = 0;
        for (int b = 0; i + (size_t)b < n; ++b) w |= (uint64_t)(unsigned char)at(i + (size_t)b) << (8 * b);
        h = (h ^ w) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32;
    }
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// One synthetic read (plane form) given the guide it derives from.
//   mixture 0: u1 < 0.9 -> guide unmodified, else random k-mer
//   mixture 1: 0/1/2/3 substitutions with P = .60/.20/.10/.05, else random
BC_HD static inline uint32_t synth_guide_index(uint64_t seed, uint64_t i, uint32_t G) {
    uint64_t u0 = sm_draw(seed, i, 0);
    return (uint32_t)(((u0 >> 32) * (uint64_t)G) >> 32);
}
BC_HD static inline uint64_t synth_read(uint64_t seed, uint64_t i, uint64_t guide, int k,
                                        int mixture) {
    uint32_t km = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    uint64_t kmask = (uint64_t)km | ((uint64_t)km << 32);
    uint64_t u1 = sm_draw(seed, i, 1);
    uint32_t c = (uint32_t)(u1 & 0xFFFF);
    int nsub;
    if (mixture == 0) {
        if (c < 58982u) return guide;
        return sm_draw(seed, i, 2) & kmask;
    }
    if (mixture >= 100) {
        // read-mix sweep: mixture = 100 + J gives J % uniform-random reads (they match nothing, but for chance); the
        // others carry 0 / 1 / 2 substitutions in the proportion 6 : 2 : 1 of the standard stream
        const uint32_t T = (65536u * (uint32_t)(200 - mixture)) / 100u;
        if (c >= T) return sm_draw(seed, i, 2) & kmask;
        nsub = c < (T * 6u) / 9u ? 0 : c < (T * 8u) / 9u ? 1 : 2;
    } else if (c < 39322u) nsub = 0;
    else if (c < 52429u) nsub = 1;
    else if (c < 58982u) nsub = 2;
    else if (c < 62259u) nsub = 3;
    else return sm_draw(seed, i, 2) & kmask;
    uint64_t r = guide;
    uint64_t u3 = sm_draw(seed, i, 3);
    uint64_t u4 = sm_draw(seed, i, 4);
    int p[3] = {0, 0, 0};
    for (int s = 0; s < nsub; ++s) {
        uint32_t rr = (uint32_t)(u3 >> (16 * s)) & 0xFFFF;
        int pos = (int)(rr % (uint32_t)(k - s));
        // skip already used positions (kept sorted ascending)
        for (int t = 0; t < s; ++t)
            if (pos >= p[t]) ++pos;
        // insert sorted
        int t = s;
        while (t > 0 && p[t - 1] > pos) { p[t] = p[t - 1]; --t; }
        p[t] = pos;
        uint32_t x = 1u + ((uint32_t)(u4 >> (16 * s)) & 0xFFFF) % 3u;  // XOR 1..3: a different base
        r ^= ((uint64_t)(x & 1u) << pos) | ((uint64_t)(x >> 1) << (32 + pos));
    }
    return r;
}

}  // namespace bc
