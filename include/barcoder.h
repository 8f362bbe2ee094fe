/*
 * barcoder.h — C ABI of libbarcoder_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the read->library matching path and the PAM
 * site scan of ryandward/barcoder.  The reference exposes no FFI for this path
 * (it is pure Python); each entry point below cites the reference interface
 * (file:line, relative to the reference checkout) whose per-read / per-base
 * work it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes, no C++ or torch types;
 *   - every function returns BC_OK (0) or a negative BC_E* code; the message of
 *     the last failure on the calling thread is bc_last_error();
 *   - "h_" pointers are host memory, "d_" pointers are device (HBM) memory of
 *     the device the bc_lib lives on; `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream);
 *   - no global state besides the thread-local error string.
 *
 * Sequence encoding (DESIGN.md "Data layout"): codes A=0 C=1 G=2 T=3, two bits
 * per base in "plane form": base i of a k-mer (0 = leftmost / 5', k <= 32) has
 * its low code bit at bit i and its high code bit at bit 32+i of a uint64_t.
 * For k <= 31 bit 63 set means "skip this read"; for any k an optional byte
 * array `flags` (non-zero = skip) carries the same information.
 */
#ifndef BARCODER_H
#define BARCODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BC_VERSION 100 /* 0.1.0 */

/* status codes */
#define BC_OK 0
#define BC_EINVAL (-1)  /* bad argument (maps to ValueError)            */
#define BC_ENOMEM (-2)  /* host or device allocation failed             */
#define BC_EHIP (-3)    /* HIP runtime error (maps to RuntimeError)     */
#define BC_ENODEV (-4)  /* no usable GPU                                */
#define BC_ECAP (-5)    /* caller-provided output capacity too small    */

/* best_idx sentinels */
#define BC_IDX_MISS (-1) /* valid read, no guide within max_mismatch     */
#define BC_IDX_SKIP (-2) /* read flagged skip (N / flank failure / short) */
#define BC_DIST_NONE 255

/* skip flag inside a packed read (k <= 31) */
#define BC_SKIP_BIT (1ULL << 63)

/* match algorithms */
#define BC_ALGO_AUTO 0  /* exact hash when max_mismatch==0, else seed index */
#define BC_ALGO_HASH 1  /* d = 0 only: open-addressing table in L2          */
#define BC_ALGO_SEED 2  /* pigeonhole seed index + verify, d <= 3 and d < k */
                        /* (slab layout when the k-mer allows it, else CSR) */
#define BC_ALGO_BRUTE 3 /* LDS-tiled all-pairs XOR+popcount, any d          */
#define BC_ALGO_SEED_CSR 4 /* the seed index in its CSR layout, always      */

/* PAM pattern position: also accept a genome character that is not ACGT */
#define BC_PAM_ANY_CHAR 16

/* window status written by bc_pack_windows */
#define BC_WIN_OK 0       /* flanks present, k-mer packed                        */
#define BC_WIN_SKIP 1     /* 'N' in read, or flanks absent: contributes nothing  */
#define BC_WIN_ODD 2      /* flanks present but window short / non-ACGT: the     */
                          /* caller keeps the string (undocumented label only)   */

typedef struct bc_lib bc_lib; /* device-resident guide library, one per GPU */

/* ---- misc ---------------------------------------------------------------- */
int bc_version(void);
const char *bc_last_error(void);
int bc_device_count(int *n_out);

/* ---- host-side encoding (K1) ---------------------------------------------
 * bc_pack_kmers: n fixed-length k-mers laid out `stride` bytes apart ->
 * packed words.  flags[i]=1 (and BC_SKIP_BIT for k<=31) where a character is
 * not one of ACGT.  Replaces the string form of library entries and windows
 * (heuristicount.py:41-57 read_fasta, BarCodeLibrary.py:27-45). */
int bc_pack_kmers(const char *h_seqs, size_t n, int k, size_t stride,
                  uint64_t *h_out, uint8_t *h_flags /* may be NULL */);

/* bc_unpack_kmers: inverse (for labels in TSV/JSON output). out is n*k chars. */
int bc_unpack_kmers(const uint64_t *h_packed, size_t n, int k, char *h_out);

/* bc_revcomp_kmers: reverse complement of packed k-mers
 * (heuristicount.py:29-30 rev_comp, PySamParser.py:8-9). */
int bc_revcomp_kmers(const uint64_t *h_in, size_t n, int k, uint64_t *h_out);

/* bc_pack_windows: the per-read front half of process_chunk
 * (heuristicount.py:449-462 validate_read, :520-538 process_single_end):
 * N-skip (:524), window slice at `start` (:527), flank check (:458-460),
 * optional reverse complement of the barcode (:532-533).
 *   text      : concatenated read bytes, read i = text[off[i] .. off[i+1])
 *   L, R      : flank strings (may be empty), k = barcode length
 *   reverse   : non-zero -> emit rev_comp(barcode) (single-end reverse reads)
 *   out       : packed barcode (BC_SKIP_BIT set unless status==BC_WIN_OK)
 *   status    : BC_WIN_* per read
 * n_threads <= 0 means all cores. */
int bc_pack_windows(const char *h_text, const uint64_t *h_off, size_t n_reads,
                    int64_t start, const char *L, int L_len, const char *R, int R_len,
                    int k, int reverse, int n_threads,
                    uint64_t *h_out, uint8_t *h_status);

/* ---- reads-file reader (host) ---------------------------------------------
 * The sequence lines of a .fastq / .reads file (optionally .gz or .zst), as
 * heuristicount.py:100-153 (read_in_chunks) selects and strips them, delivered
 * as one text buffer + offsets in the form bc_pack_windows takes.
 * bc_reader_next returns up to max_reads reads (fewer if text_cap is reached;
 * *n_out == 0 at end of file). */
typedef struct bc_reader bc_reader;
int bc_reader_open(const char *path, bc_reader **out);
/* bc_reader_open_range: only the records of an UNCOMPRESSED file that start in
 * the byte range [begin, end) — one rank's shard when the chunks of
 * heuristicount.py:703-737 (Pool fan-out) are spread over several GPUs.
 * Ranges that tile the file deliver every record exactly once.
 * bc_reader_tell: offset of the next unread byte in the decoded stream; it
 * grows by at least one per read, so (offset at chunk start + i) orders reads
 * like their global index does (first-seen order, heuristicount.py:876-877).
 * A damaged or truncated .gz / .zst stream makes bc_reader_next fail with
 * BC_EINVAL (the reference's gzip / zstandard readers raise there). */
int bc_reader_open_range(const char *path, uint64_t begin, uint64_t end, bc_reader **out);
int bc_reader_tell(const bc_reader *r, uint64_t *offset_out);
int bc_reader_next(bc_reader *r, uint64_t max_reads, char *h_text, uint64_t text_cap,
                   uint64_t *h_off /* max_reads + 1 */, uint64_t *n_out);
/* The decoded byte stream itself, sequentially (what the device ingest uploads of a
 * .gz / .zst file; *got == 0 at the end; not to be mixed with bc_reader_next). */
int bc_reader_read_raw(bc_reader *r, char *dst, uint64_t cap, uint64_t *got);
int bc_reader_close(bc_reader *r);

/* ---- barcodes of 33 ... 1000 nt (exact match) ---------------------------------
 * heuristicount.py:75-97 admits barcodes up to 1 000 nt and its set lookup
 * (:449-453) is length-agnostic; the packed matcher holds 32 bases.  Longer
 * windows are matched through a 64-bit hash of the oriented barcode text — a
 * pseudo 32-mer for the GPU's exact table (bc_lib_create with k = 32 over
 * bc_hash_kmers of the library) — and every hit is verified against the text.
 * bc_hash_windows = bc_pack_windows for such windows: same N-skip / slice / flank
 * rules and status codes; h_out[i] = hash of the barcode as the library spells
 * it (reverse-complemented when `reverse`), h_windows[i*k .. i*k+k) = that text
 * (both only for status BC_WIN_OK). */
int bc_hash_kmers(const char *h_seqs, size_t n, int k, uint64_t *h_out);
int bc_hash_windows(const char *h_text, const uint64_t *h_off, size_t n_reads, int64_t start,
                    const char *L, int L_len, const char *R, int R_len, int k, int reverse,
                    int n_threads, uint64_t *h_out, uint8_t *h_status, char *h_windows);

/* ---- a reads file straight through the device --------------------------------
 * bc_stream_count_file: everything process_chunk does per read
 * (heuristicount.py:428-562) for the records of a .reads / .fastq file that lie in
 * [byte_begin, byte_end) — a range that starts at a record
 * boundary (0, or a bc_reader_open_range / bc_reader_tell position) — with the
 * raw bytes uploaded in chunks and line splitting, strip, N-skip (:524), window
 * slice (:527), flank test (:458-460), packing, reverse complement (:532-533),
 * match, per-guide tally and the grouping of the unmatched flank-valid windows
 * (the "undocumented" barcodes, :455-457) all done in HBM.  Single-end
 * geometry as bc_pack_windows (start, L, R, reverse); k is the library's.
 *   key_base       : first-seen key of read i of the range = key_base + i
 *   h_counts       : uint64[n_guides], ADDED to; h_first_seen: uint64[n_guides], min'ed
 *   n_threads      : host reader / launcher threads (<= 0: 4)
 * The result object holds the read count, the undocumented table (packed word,
 * count, smallest key; sorted by word) and the text + key of the odd windows
 * (status BC_WIN_ODD: flanks present but the window short or not pure ACGT),
 * which keep the reference's string semantics on the caller's side. */
typedef struct bc_stream_result bc_stream_result;
int bc_stream_count_file(bc_lib *lib, const char *path, uint64_t byte_begin, uint64_t byte_end,
                         int64_t start, const char *L, int L_len, const char *R, int R_len,
                         int reverse, int max_mismatch, int algo, uint64_t key_base, int n_threads,
                         uint64_t *h_counts, uint64_t *h_first_seen, bc_stream_result **out);
/* Compressed input (.gz / .zst, heuristicount.py:60-72, :100-153) takes the same path: the
 * stream is decoded on the host, sequentially, in 16 MiB pieces straight into the
 * page-locked upload buffers of the worker threads (whichever worker is next decodes;
 * uploads and kernels of earlier chunks overlap it), so the bulk rate is the decoder's.
 * A compressed stream has no byte ranges: bc_stream_count_file wants (0, UINT64_MAX) for
 * it, and several ranks share it by CHUNK instead — bc_stream_count_file_part: chunk c
 * (16 MiB of the decoded stream) belongs to part c % n_parts; every rank decodes, one rank
 * does a chunk's per-read work; keys are key_base + the read's index in the whole file.
 * (Plain files may be split the same way.)  bc_stream_count_pair accepts compressed mates
 * too (whole files: both decoders run side by side). */
int bc_stream_count_file_part(bc_lib *lib, const char *path, uint32_t part, uint32_t n_parts,
                              int64_t start, const char *L, int L_len, const char *R, int R_len,
                              int reverse, int max_mismatch, int algo, uint64_t key_base,
                              int n_threads, uint64_t *h_counts, uint64_t *h_first_seen,
                              bc_stream_result **out);
int bc_stream_result_info(const bc_stream_result *r, uint64_t *n_reads, uint64_t *n_undoc,
                          uint64_t *n_odd, uint64_t *odd_bytes, double *seconds);
int bc_stream_result_undoc(const bc_stream_result *r, uint64_t *words, uint64_t *counts,
                           uint64_t *first);
int bc_stream_result_odd(const bc_stream_result *r, uint64_t *keys, uint64_t *off /* n_odd + 1 */,
                         char *text);
/* libraries with attached texts (bc_lib_attach_text): the undocumented windows as texts (n_undoc x text_k
 * characters, sorted by their hash) instead of packed words; *text_k = 0 for a packed library */
int bc_stream_result_text_info(const bc_stream_result *r, int *text_k);
int bc_stream_result_undoc_text(const bc_stream_result *r, char *text, uint64_t *counts, uint64_t *first);
int bc_stream_result_free(bc_stream_result *r);

/* bc_stream_count_pair: the paired-end form (heuristicount.py:469-518, reached from
 * :540-554).  Both files (plain: one after the other; .gz / .zst: their decoders side
 * by side) go through the device, each leaving {packed window, status} per read in HBM; then the pairs are formed
 * (neither mate holds an N, both show their flanks, seq1 == rev_comp(seq2), :489-507),
 * matched once over the whole run (documented: both windows in the library,
 * :509-515; undocumented: flanks only, :517-518) and tallied.  path_fwd = the file
 * whose reads carry the barcode in library orientation — reads1 AFTER the
 * reference's need_swap exchange (:444-447) — with geometry (start_fwd, L_fwd,
 * R_fwd); path_rev the other file, geometry (start_rev, L_rev, R_rev), windows
 * reverse-complemented.  Key of pair i = key_base + i.  Files with different
 * numbers of reads fail with BC_EINVAL and the reference's message (:484-486).
 * Pairs with an odd window (flanks present, window short or not pure ACGT) come
 * back through bc_stream_result_pair_*: per pair its key, both mates' packed word
 * and status, and the stripped text of each odd mate (empty for a packed one).
 * A library with attached texts (barcodes of 33 ... 1000 nt, bc_lib_attach_text;
 * max_mismatch must be 0): the per-read stores also keep the window's text as the
 * library spells it (text_k bytes per read and file), a pair needs equal hashes AND
 * equal texts, every hit is verified against the library's text, the undocumented
 * windows come back as texts (bc_stream_result_undoc_text), and a PACKED mate of an
 * odd pair returns its barcode text (text_k characters) where a short barcode returns
 * nothing — its word is a hash and cannot be unpacked. */
int bc_stream_count_pair(bc_lib *lib, const char *path_fwd, const char *path_rev,
                         int64_t start_fwd, const char *L_fwd, int L_fwd_len, const char *R_fwd,
                         int R_fwd_len, int64_t start_rev, const char *L_rev, int L_rev_len,
                         const char *R_rev, int R_rev_len, int max_mismatch, int algo,
                         uint64_t key_base, int n_threads, uint64_t *h_counts,
                         uint64_t *h_first_seen, bc_stream_result **out);
/* The same over one byte range of each file — ranges that hold the SAME reads: one rank's
 * share when the pairs are spread over several GPUs (heuristicount.py:703-737).  The ranges
 * come from line arithmetic both files allow: bc_file_count_newlines over a slice per rank,
 * summed over the ranks, tells every rank which line its share starts at;
 * bc_file_seek_newline turns a line number into a byte offset. */
int bc_stream_count_pair_range(bc_lib *lib, const char *path_fwd, uint64_t fwd_begin, uint64_t fwd_end,
                               const char *path_rev, uint64_t rev_begin, uint64_t rev_end,
                               int64_t start_fwd, const char *L_fwd, int L_fwd_len, const char *R_fwd,
                               int R_fwd_len, int64_t start_rev, const char *L_rev, int L_rev_len,
                               const char *R_rev, int R_rev_len, int max_mismatch, int algo,
                               uint64_t key_base, int n_threads, uint64_t *h_counts,
                               uint64_t *h_first_seen, bc_stream_result **out);
int bc_file_count_newlines(const char *path, uint64_t begin, uint64_t end, uint64_t *n_out,
                           uint64_t *size_out, int *last_byte_out);
int bc_file_seek_newline(const char *path, uint64_t begin, uint64_t nth, uint64_t *offset_out);
int bc_stream_result_pair_info(const bc_stream_result *r, uint64_t *n_pairs, uint64_t *text1_bytes,
                               uint64_t *text2_bytes);
int bc_stream_result_pair_odd(const bc_stream_result *r, uint64_t *keys, uint64_t *p1, uint8_t *st1,
                              uint64_t *p2, uint8_t *st2, uint64_t *off1 /* n_pairs + 1 */,
                              char *text1, uint64_t *off2 /* n_pairs + 1 */, char *text2);

/* ---- sampling (host) --------------------------------------------------------
 * bc_sample_data: orientation / barcode offset discovery on the first chunks of
 * the reads file(s), heuristicount.py:156-379 (sample_data).  `barcodes` = G
 * library entries of k characters.  The result object keeps the sampled reads;
 * bc_sample_info returns the reference's tuple (offsets -1 = None), and
 * bc_sample_flanks runs heuristicount.py:382-425 (find_flanks) on sample1 / sample2
 * (flank length -1 = None; buffers need max_flank + 1 bytes). */
typedef struct bc_sample bc_sample;
int bc_sample_data(const char *file1, const char *file2 /* NULL = single-end */,
                   const char *barcodes, uint32_t n_barcodes, int k, bc_sample **out);
int bc_sample_info(const bc_sample *s, int64_t *n_novel, int64_t *offset1, int64_t *offset2,
                   uint64_t *n_valid1, uint64_t *n_valid2, uint64_t *n_observed, int *need_swap,
                   uint64_t *n_chunks, int *satisfied);
int bc_sample_flanks(const bc_sample *s, int which, int64_t start, int bc_len, int max_flank,
                     char *L_out, int *L_len, char *R_out, int *R_len);
int bc_sample_free(bc_sample *s);

/* ---- library -------------------------------------------------------------
 * bc_lib_create: upload G distinct packed guides (index = position in the
 * array = first-occurrence order of the de-duplicated library,
 * BarCodeLibrary.py:48-93 / heuristicount.py:41-57) to `device`, build the
 * exact-match table and the pigeonhole seed indices for d = 1..3.
 * Duplicate guides are rejected with BC_EINVAL. */
int bc_lib_create(const uint64_t *h_guides, uint32_t n_guides, int k, int device,
                  bc_lib **out);
int bc_lib_destroy(bc_lib *lib);
int bc_lib_info(const bc_lib *lib, uint32_t *n_guides, int *k, int *device);
/* A library keeps a small workspace (tally records, d = 3 cascade lists) per stream
 * that launches on it.  A caller that creates and destroys streams releases a
 * stream's workspace before destroying the stream (waits for the stream). */
int bc_lib_release_stream(bc_lib *lib, void *stream);
/* How many streams hold a workspace on the library right now (at most 1024).  The
 * ingest entry points (bc_stream_count_*) release their worker streams on every exit,
 * failed runs included: the count is back where it was when they return. */
int bc_lib_workspace_count(bc_lib *lib, uint32_t *n_streams);
/* Barcodes of 33 ... 1000 nt: the library was created with k = 32 over bc_hash_kmers of the texts; attaching the
 * texts (n_guides x text_k characters) lets bc_stream_count_file take such a library: windows are hashed on the
 * device, every hit is verified against the text there, the undocumented table carries texts. */
int bc_lib_attach_text(bc_lib *lib, const char *h_text, int text_k);
/* name of the match kernel the last bc_match_* call on this library launched
 * ("k_match_wq", "k_match_hash", "cascade3:d2+k_match_pairs", "k_match_seed", ...): lets a test assert
 * which path a (k, max_mismatch, library size) combination takes. */
const char *bc_lib_last_kernel(const bc_lib *lib);

/* ---- read -> library match (K2, K3, K5) ----------------------------------
 * The device twin of the back half of process_chunk (heuristicount.py:449-453
 * set lookup, :535-538 Counter update) generalised to <= max_mismatch
 * substitutions (the reference delegates that to `bowtie -v N`,
 * targets.py:496-509, BowtieRunner.py:111-125).
 *
 * Semantics (DESIGN.md "Best-hit definition"): best = the guide at minimum
 * Hamming distance <= max_mismatch; ties -> lowest guide index; n_best = number
 * of guides at that distance (saturating at 255).
 *
 *   d_reads      : n packed reads
 *   d_flags      : optional n bytes, non-zero = skip
 *   d_best_idx   : n int32 out (guide index, BC_IDX_MISS, BC_IDX_SKIP)
 *   d_best_dist  : optional n uint8 out (distance, BC_DIST_NONE)
 *   d_n_best     : optional n uint8 out
 *   d_counts     : optional uint32[n_guides], ATOMICALLY INCREMENTED (caller
 *                  zeroes it; several launches may accumulate into it)
 *   d_first_seen : optional uint64[n_guides], atomically min'ed with
 *                  first_index_base + i for the matched read i (caller
 *                  initialises to UINT64_MAX); gives heuristicount.py:876-877's
 *                  first-seen output order
 *
 * Asynchronous on `stream` for every budget.  max_mismatch = 3 runs as a cascade
 * (8 <= k <= 25): the whole <= 2-mismatch pass over every read (kernel
 * k_match_wq, tally included), then the reads still marked BC_IDX_MISS through the
 * pair-of-pieces index, results written in place.  Launches of 65 536 reads or
 * more work from a list of those reads that k_match_wq writes itself (packed by a
 * prefix sum on the device, k_cascade_pack, then k_cascade_pass over groups of
 * tables); smaller ones find them in d_best_idx (k_cascade_stage2).  Either way
 * the count of missed reads stays on the device: nothing the host waits for.
 * max_mismatch = 2 runs the same way — the <= 1-mismatch pass, then the pair index,
 * for this budget only the four of its ten tables that cover three clean pieces —
 * on libraries whose three seed tables are too crowded for k_match_wq to be the
 * faster choice (from about 3.6e5 20-mers or 6.4e4 16-mers on; bc_lib_last_kernel
 * then reports "cascade2:d1+k_match_pairs"; BC_CASCADE2_MIN_G overrides).
 * (The first launch with a tally or a cascade on a stream, or one larger than
 * every earlier launch on it, allocates that stream's workspace — records, and
 * for the list form 32 bytes per read — and waits for the stream once;
 * bc_lib_release_stream frees it.)  0 <= max_mismatch < k; the seed algorithms
 * take max_mismatch <= 3.
 */
int bc_match_dev(bc_lib *lib, const uint64_t *d_reads, const uint8_t *d_flags, uint64_t n,
                 int max_mismatch, int algo,
                 int32_t *d_best_idx, uint8_t *d_best_dist, uint8_t *d_n_best,
                 uint32_t *d_counts, uint64_t *d_first_seen, uint64_t first_index_base,
                 void *stream);

/* Same with host buffers: H2D, match, D2H on the library's device, in chunks of 1M
 * reads that worker threads pipeline (two buffer sets each: one chunk on the link /
 * the device while the one before it is copied out and the next one in), so any n
 * fits.  Buffers in page-locked memory (bc_host_alloc_pinned, hipHostRegister) are
 * the DMA's source / target as they are; pageable ones go through page-locked staging.
 * h_counts (uint64[n_guides]) is ADDED to, h_first_seen (uint64[n_guides]) is min'ed.
 * h_best_idx == NULL (then h_best_dist and h_n_best NULL too, and a tally wanted):
 * counts-only — what heuristicount.py:535-538 needs of a chunk — 8 bytes per read go
 * up the link, nothing per read comes back. */
int bc_match_host(bc_lib *lib, const uint64_t *h_reads, const uint8_t *h_flags, uint64_t n,
                  int max_mismatch, int algo,
                  int32_t *h_best_idx, uint8_t *h_best_dist, uint8_t *h_n_best,
                  uint64_t *h_counts, uint64_t *h_first_seen, uint64_t first_index_base);
/* Copy rates of the host link between page-locked memory and HBM, GB/s: upload alone,
 * download alone, both at once (`bytes` each way, `repeats` times): the roofline of
 * bc_match_host (measurement aid, bench.py secondary.host_pcie). */
int bc_pcie_probe(int device, uint64_t bytes, int repeats, double *h2d_GBps, double *d2h_GBps,
                  double *duplex_GBps);

/* ---- several GPUs in one process --------------------------------------------
 * The data-parallel fan-out of heuristicount.py:703-737 (Pool.starmap over
 * chunks, Counter merge :731-737) for a caller that owns all GPUs of a node
 * from one process: one replica of the library per listed device
 * (devices == NULL: 0..n_gpus-1), reads sharded in contiguous ranges of
 * ceil(n / n_gpus), per-read outputs written to the caller's arrays range by
 * range, and ONE all-reduce of the per-guide vectors at the end — SUM of the
 * counts, MIN of the first-seen indices — over RCCL/xGMI when the devices are
 * distinct (librccl is resolved with dlopen).  A device listed more than once,
 * or n_gpus == 1, reduces through the host (rehearsal on fewer GPUs).
 * Arguments as bc_match_host; h_counts is ADDED to, h_first_seen min'ed.
 * The one-process-per-GPU form of the same job is barcoder_amd/dist.py. */
typedef struct bc_multi bc_multi;
int bc_multi_create(const uint64_t *h_guides, uint32_t n_guides, int k, const int *devices,
                    int n_gpus, bc_multi **out);
int bc_multi_info(const bc_multi *m, int *n_gpus, int *uses_rccl);
int bc_multi_gpu_match(bc_multi *m, const uint64_t *h_reads, const uint8_t *h_flags, uint64_t n,
                       int max_mismatch, int algo,
                       int32_t *h_best_idx, uint8_t *h_best_dist, uint8_t *h_n_best,
                       uint64_t *h_counts, uint64_t *h_first_seen, uint64_t first_index_base);
int bc_multi_destroy(bc_multi *m);

/* ---- all hits (guide -> genome search, the Bowtie bypass) -----------------
 * For every query word q[i] (a genome window or a read) emit every guide
 * within max_mismatch as a record {query index, guide index, distance}; this is
 * what `bowtie -a -v N` returns for spacers against the genome
 * (BowtieRunner.py:111-125, targets.py:496-509) with the roles of query and
 * index swapped.  Records are written in unspecified order; *d_n_hits (uint64,
 * zeroed by the caller) counts ALL hits, also those beyond `cap` (which are
 * dropped) so the caller can retry with a larger buffer. */
typedef struct bc_hit {
    uint32_t query;
    uint32_t guide;
    uint32_t dist;
} bc_hit;
int bc_match_all_dev(bc_lib *lib, const uint64_t *d_queries, const uint8_t *d_flags,
                     uint64_t n, int max_mismatch, bc_hit *d_hits, uint64_t cap,
                     uint64_t *d_n_hits, void *stream);

/* Same search gated by a PAM bitmap: query i is the genome window at position i
 * and is searched only if bit (i + pam_shift) of d_pam_bits is set — a bitmap
 * from bc_pam_scan_dev for the window's strand; positions outside
 * [0, pam_limit) fail, negative ones wrap once by wrap_len when it is non-zero
 * (circular sequence).  This is the PAM test of targets.py:325-352
 * (extract_downstream_pam / extract_upstream_pam + pam_matches) applied before
 * the search instead of to every alignment after it. */
int bc_match_all_pam_dev(bc_lib *lib, const uint64_t *d_windows, const uint8_t *d_flags,
                         uint64_t n, int max_mismatch, const uint32_t *d_pam_bits,
                         int64_t pam_shift, uint64_t pam_limit, uint64_t wrap_len,
                         bc_hit *d_hits, uint64_t cap, uint64_t *d_n_hits, void *stream);

/* ---- genome encoding + PAM site scan (K4) --------------------------------
 * bc_genome_pack: ASCII bases -> one uint64 per 32 positions in the same plane
 * form as a 32-mer (word w = positions 32w..32w+31; low half = low code bits,
 * high half = high code bits) plus an "invalid" bitmap (bit j of word w set
 * where the character is not ACGT, case-insensitive, or past the end).
 * `wrap` > 0 appends the first `wrap` bases again after the end (position
 * len+i = seq[i % len]) so that sites of a circular sequence that span the
 * origin have a linear representation: the device-side replacement for the
 * reference's 100 kb overhang (targets.py:35-56 create_topological_fasta).
 * Outputs hold ceil((len+wrap)/32) words. */
int bc_genome_pack(const char *h_seq, uint64_t len, uint64_t wrap, uint64_t *h_bases,
                   uint32_t *h_invalid);

/* bc_pam_scan_*: for every position p in [0, total_len) report whether the PAM
 * pattern matches genome[p .. p+pam_len) on the forward strand (bitmap fwd) and
 * whether revcomp(PAM) matches there, i.e. a PAM on the reverse strand (bitmap
 * rev).  Bit p%32 of word p/32; ceil(total_len/32) words each.
 * `pam_masks[j]` is the IUPAC set of pattern position j as a 4-bit mask
 * (bit0=A bit1=C bit2=G bit3=T; N = 15).  Positions past total_len never
 * match; a genome character that is not ACGT matches only a pattern position
 * that carries BC_PAM_ANY_CHAR as well (the '.' targets.py:224 writes for N;
 * PAMProcessor.py:31 and design_guides.py:26 write [ATCG], i.e. no flag).
 * This is the per-position test of PAMProcessor.py:34-57
 * (GuideFinder.find_guides_from_pam), design_guides.py:22-49
 * (find_sequences_with_barcode_and_pam) and targets.py:219-224 (pam_matches);
 * the reference-flavoured site lists are selections from these bitmaps
 * (barcoder_amd/pam.py). */
int bc_pam_scan_dev(const uint64_t *d_bases, const uint32_t *d_invalid, uint64_t total_len,
                    const uint8_t *pam_masks, int pam_len, uint32_t *d_fwd, uint32_t *d_rev,
                    void *stream);
int bc_pam_scan_host(int device, const uint64_t *h_bases, const uint32_t *h_invalid,
                     uint64_t total_len, const uint8_t *pam_masks, int pam_len,
                     uint32_t *h_fwd, uint32_t *h_rev);

/* Timing aid for benchmarks: bc_pam_scan_dev launched `iters` times back to back
 * from C, HIP events around them on `stream`; *ms_out = mean ms per launch. */
int bc_pam_scan_time_dev(const uint64_t *d_bases, const uint32_t *d_invalid, uint64_t total_len,
                         const uint8_t *pam_masks, int pam_len, uint32_t *d_fwd, uint32_t *d_rev,
                         int iters, void *stream, float *ms_out);

/* bc_pam_sites_*: the whole site pipeline of design_guides.py:22-49
 * (find_sequences_with_barcode_and_pam) / PAMProcessor.py:34-57 in two launches:
 * every position p where the PAM matches (either strand) AND the k-mer window
 * next to it — [p-k, p) forward / [p+pam_len, p+pam_len+k) reverse for
 * direction 0 (PAM downstream of the spacer), mirrored for direction 1 — lies
 * inside [0, total_len) and is pure ACGT (design_guides.py:36,46).
 *   count: d_counts[2b], d_counts[2b+1] = forward / reverse sites of block b of
 *          1024 words; d_counts holds 2*(ceil(n_words/1024)+1) uint32.  The
 *          caller sums them to size the outputs.
 *   emit:  d_pos[i] = PAM position, d_words[i] = the spacer, oriented 5'->3' on
 *          its own strand (plane form), forward-strand sites first (ascending
 *          position), then reverse-strand sites (ascending); d_n_out[0..1] =
 *          the two totals (uint64).  Records beyond `cap` are not written.
 *          Consumes d_counts (genomes above 134 Mbp prefix-sum it in place).
 * pam_masks as for bc_pam_scan_dev, without BC_PAM_ANY_CHAR. */
int bc_pam_sites_count_dev(const uint64_t *d_bases, const uint32_t *d_invalid, uint64_t total_len,
                           const uint8_t *pam_masks, int pam_len, int k, int direction,
                           uint32_t *d_counts, void *stream);
int bc_pam_sites_emit_dev(const uint64_t *d_bases, const uint32_t *d_invalid, uint64_t total_len,
                          const uint8_t *pam_masks, int pam_len, int k, int direction,
                          uint32_t *d_counts, uint32_t *d_pos, uint64_t *d_words, uint64_t cap,
                          uint64_t *d_n_out, void *stream);

/* bc_sites_expand_dev: bitmap -> ascending position list (uint32).  *d_n_out
 * (uint64) receives the number of set bits; positions beyond `cap` are not
 * written.  d_tmp must hold ceil(n_words/1024)+1 uint32. */
int bc_sites_expand_dev(const uint32_t *d_bitmap, uint64_t n_words, uint32_t *d_pos,
                        uint64_t cap, uint64_t *d_n_out, uint32_t *d_tmp, void *stream);

/* bc_windows_dev: k-mer at p = pos[t] + offset for each listed position of a
 * packed genome, on the forward strand (strand=0: genome[p .. p+k)) or
 * reverse-complemented (strand=1: revcomp(genome[p .. p+k))).  Windows that
 * touch a non-ACGT base, start before 0 or run past total_len get
 * BC_SKIP_BIT (k<=31) and flag 1.
 * Feeds bc_match_dev / bc_match_all_dev (targets.py:354-410 hit records). */
int bc_windows_dev(const uint64_t *d_bases, const uint32_t *d_invalid, uint64_t total_len,
                   const uint32_t *d_pos /* NULL = every position 0..n_pos-1 */,
                   uint64_t n_pos, int64_t offset, int k, int strand, uint64_t *d_out,
                   uint8_t *d_flags, void *stream);

/* ---- mismatch-efficacy score (mismatch.py:15-35 calculate_y_pred) ------------
 * y_pred for n (original, variant) pairs of packed k-mers, float64, bit-identical
 * to the reference's Python arithmetic.  d_weights[50]: [0] intercept, [1+p]
 * position p (0..31), [33 + 4*o + v] substitution o->v (codes A C G T), [49]
 * GC_content.  d_none[i] = 1 where the pair is identical (the reference
 * returns None there); may be NULL. */
int bc_score_mismatch_dev(const uint64_t *d_orig, const uint64_t *d_var, uint64_t n, int k,
                          const double *d_weights, double *d_out, uint8_t *d_none, void *stream);

/* Tallies of a pass in the form the job-wide merge wants (heuristicount.py:726-737 merges per-worker Counters; here
 * the per-GPU vectors are all-reduced: SUM of 64-bit counts, MIN of first-seen read indices).  counts64[g] =
 * counts[g]; first_min[g] = first_seen[g], with INT64_MAX for "unseen" (the identity of a signed 64-bit MIN, which
 * is what torch.distributed reduces).  reset != 0 leaves counts = 0 and first_seen = UINT64_MAX for the next pass.
 * Any of the four pointers may be NULL (an output needs its accumulator). */
int bc_tally_export_dev(uint32_t *d_counts, uint64_t *d_first_seen, int64_t *d_counts64, int64_t *d_first_min,
                        uint32_t n_guides, int reset, void *stream);

/* ---- synthetic workloads (bench / tests) -----------
 * Counter-based splitmix64 streams, identical to barcoder_amd/synth.py.
 * mixture: 0 = exact (90 % library reads, 10 % random),
 *          1 = mismatch (60/20/10/5 % with 0/1/2/3 substitutions, 5 % random). */
int bc_synth_reads_dev(bc_lib *lib, uint64_t seed, uint64_t first_index, uint64_t n,
                       int mixture, uint64_t *d_reads, void *stream);

/* ---- raw device memory helpers (so a ctypes caller needs no torch) -------- */
int bc_dev_set(int device);
int bc_dev_alloc(void **d_ptr, size_t bytes);
int bc_dev_free(void *d_ptr);
int bc_dev_memset(void *d_ptr, int value, size_t bytes, void *stream);
int bc_dev_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int bc_dev_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int bc_dev_sync(void *stream);
/* page-locked host memory: a D2H / H2D copy to or from it runs at the link's rate
 * and needs no staging copy inside the runtime */
int bc_host_alloc_pinned(void **h_ptr, size_t bytes);
int bc_host_free_pinned(void *h_ptr);

/* ---- timing of the dominant kernel (bench.py roofline) --------------------
 * Runs bc_match_dev `iters` times on `stream` bracketed by HIP events recorded
 * on that stream and returns the mean kernel-launch duration in ms. */
int bc_match_time_dev(bc_lib *lib, const uint64_t *d_reads, const uint8_t *d_flags, uint64_t n,
                      int max_mismatch, int algo, int32_t *d_best_idx, uint8_t *d_best_dist,
                      uint8_t *d_n_best, uint32_t *d_counts, uint64_t *d_first_seen,
                      int iters, void *stream, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* BARCODER_H */
