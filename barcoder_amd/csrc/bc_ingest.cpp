// bc_ingest.cpp — native reads-file reader: the
// sequence lines of a FASTQ or .reads file (plain, .gz via zlib, .zst via a
// dlopen'ed libzstd) as one text buffer + offsets, ready for bc_pack_windows.
//
// Follows heuristicount.py:100-153 (read_in_chunks): the file type comes from the
// name after an optional .gz/.zst suffix (".fastq" -> line 2 of every 4,
// ".reads" -> every line; anything else is rejected, :111-116), every kept line
// is stripped of surrounding whitespace (:127,:138), and the sequence line of
// a truncated last record still counts (it is appended before the record's
// remaining lines are skipped).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bc_common.h"

using namespace bc;

namespace {

// minimal libzstd streaming API (resolved at run time; the image ships the
// shared object but no header)
struct ZBuf { void* p; size_t size; size_t pos; };
struct Zstd {
    void* h = nullptr;
    void* (*createDCtx)() = nullptr;
    size_t (*freeDCtx)(void*) = nullptr;
    size_t (*decompressStream)(void*, ZBuf*, ZBuf*) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool load() {
        if (h) return true;
        for (const char* n : {"libzstd.so.1", "libzstd.so"}) {
            h = dlopen(n, RTLD_NOW);
            if (h) break;
        }
        if (!h) return false;
        createDCtx = (void* (*)())dlsym(h, "ZSTD_createDCtx");
        freeDCtx = (size_t (*)(void*))dlsym(h, "ZSTD_freeDCtx");
        decompressStream = (size_t (*)(void*, ZBuf*, ZBuf*))dlsym(h, "ZSTD_decompressStream");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        return createDCtx && freeDCtx && decompressStream && isError;
    }
};
Zstd g_zstd;

}  // namespace

struct bc_reader {
    int kind = 0;        // 0 = .reads (every line), 1 = .fastq (line 2 of 4)
    int comp = 0;        // 0 plain, 1 gz, 2 zst
    FILE* fp = nullptr;
    gzFile gz = nullptr;
    void* zctx = nullptr;
    std::vector<char> zin;
    size_t zin_pos = 0, zin_len = 0;
    std::vector<char> buf;  // decoded bytes not yet consumed
    size_t pos = 0, len = 0;
    bool eof = false;
    bool failed = false;     // the decoder reported an error or the stream was cut off: never mistaken for EOF
    std::string fail_msg;
    size_t zst_last = 0;     // last ZSTD_decompressStream result: 0 = at a frame boundary
    uint64_t line_in_record = 0;
    uint64_t abs0 = 0;       // offset of buf[0] in the decoded stream (= file offset for plain files)
    uint64_t end = UINT64_MAX;  // bc_reader_open_range: records starting at or after this offset are not ours

    // decoded bytes into dst; 0 = end of stream OR failure (then `failed` is set).  The reference's readers
    // raise on a damaged stream (gzip: EOFError / BadGzipFile, zstandard: ZstdError) and main() prints no
    // counts (heuristicount.py:882-888); a shortened read stream must never pass for a complete one.
    size_t fill(char* dst, size_t cap) {
        if (comp == 0) {
            size_t n = fread(dst, 1, cap, fp);
            if (n == 0 && ferror(fp)) { failed = true; fail_msg = "read error"; }
            return n;
        }
        if (comp == 1) {
            int n = gzread(gz, dst, (unsigned)cap);
            int errnum = Z_OK;
            const char* msg = gzerror(gz, &errnum);
            if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) {
                failed = true;
                fail_msg = std::string("gzip stream damaged or truncated (") + (msg && *msg ? msg : "zlib error") + ")";
                return 0;
            }
            return (size_t)n;
        }
        size_t produced = 0;
        while (produced == 0) {
            if (zin_pos == zin_len) {
                zin_len = fread(zin.data(), 1, zin.size(), fp);
                zin_pos = 0;
                if (zin_len == 0) {
                    if (ferror(fp)) { failed = true; fail_msg = "read error"; }
                    else if (zst_last != 0) { failed = true; fail_msg = "zstd stream truncated (input ends inside a frame)"; }
                    return 0;
                }
            }
            ZBuf in{zin.data(), zin_len, zin_pos}, out{dst, cap, 0};
            size_t rc = g_zstd.decompressStream(zctx, &out, &in);
            zin_pos = in.pos;
            if (g_zstd.isError(rc)) { failed = true; fail_msg = "zstd stream damaged"; return 0; }
            zst_last = rc;
            produced = out.pos;
        }
        return produced;
    }

    // index of the '\n' that ends the line starting at `from` (>= pos), or len when the stream ends first;
    // grows the buffer instead of moving it, so indices stay valid (used while synchronising on a record)
    size_t line_end_nomove(size_t from) {
        for (;;) {
            const char* nl = from < len ? (const char*)memchr(buf.data() + from, '\n', len - from) : nullptr;
            if (nl) return (size_t)(nl - buf.data());
            if (eof) return len;
            if (len == buf.size()) buf.resize(buf.size() * 2);
            size_t got = fill(buf.data() + len, buf.size() - len);
            if (got == 0) { eof = true; return len; }
            len += got;
        }
    }
};

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

extern "C" {

int bc_reader_open(const char* path, bc_reader** out) {
    if (!path || !out) { set_error("bc_reader_open: null argument"); return BC_EINVAL; }
    std::string p(path), base(path);
    int comp = 0;
    if (ends_with(p, ".gz")) { comp = 1; base = p.substr(0, p.size() - 3); }
    else if (ends_with(p, ".zst")) { comp = 2; base = p.substr(0, p.size() - 4); }
    int kind;
    if (ends_with(base, ".fastq")) kind = 1;
    else if (ends_with(base, ".reads")) kind = 0;
    else { set_error("Unsupported file type. Must be '.fastq' or '.reads'."); return BC_EINVAL; }
    bc_reader* r = new bc_reader();
    r->kind = kind; r->comp = comp;
    if (comp == 1) {
        r->gz = gzopen(path, "rb");
        if (!r->gz) { delete r; set_error("cannot open %s", path); return BC_EINVAL; }
        gzbuffer(r->gz, 1 << 20);
    } else {
        r->fp = fopen(path, "rb");
        if (!r->fp) { delete r; set_error("cannot open %s", path); return BC_EINVAL; }
        if (comp == 2) {
            if (!g_zstd.load()) { fclose(r->fp); delete r; set_error("%s: libzstd is not available", path); return BC_EINVAL; }
            r->zctx = g_zstd.createDCtx();
            r->zin.resize(1 << 20);
        }
    }
    r->buf.resize(8 << 20);
    *out = r;
    return BC_OK;
}

// The records of an UNCOMPRESSED file that start in the byte range [begin, end): the shard of one rank when a
// file is split over several GPUs (barcoder_amd/dist.py; the data-parallel fan-out of heuristicount.py:703-737
// over independent chunks).  A .reads record starts after every newline; a FASTQ record start is a line that
// begins with '@' whose third line begins with '+' (a quality line may begin with '@', a sequence line never
// begins with '+').  Ranges that tile the file yield every record exactly once.
int bc_reader_open_range(const char* path, uint64_t begin, uint64_t end, bc_reader** out) {
    int rc = bc_reader_open(path, out);
    if (rc != BC_OK) return rc;
    bc_reader* r = *out;
    if (r->comp != 0) {
        bc_reader_close(r); *out = nullptr;
        set_error("bc_reader_open_range: %s is compressed; byte ranges need a plain .fastq / .reads file", path);
        return BC_EINVAL;
    }
    r->end = end;
    if (begin == 0) return BC_OK;
    if (fseeko(r->fp, (off_t)(begin - 1), SEEK_SET) != 0) {
        bc_reader_close(r); *out = nullptr;
        set_error("bc_reader_open_range: cannot seek in %s", path);
        return BC_EINVAL;
    }
    r->abs0 = begin - 1;
    // first line start at or after `begin`: one past the first newline at or after begin - 1
    size_t e = r->line_end_nomove(0);
    r->pos = e < r->len ? e + 1 : r->len;
    if (r->kind == 1) {
        // synchronise on a record header: '@' line whose third line starts with '+'.  When the file ends before
        // that third line, a '@' line is taken as the header of a cut-off last record (its sequence line, if any,
        // still counts: heuristicount.py:127-131).
        for (;;) {
            const size_t e0 = r->line_end_nomove(r->pos);   // end of the candidate line
            if (r->pos >= r->len) break;                    // nothing left in this range
            bool header = r->buf[r->pos] == '@';
            if (header && e0 < r->len) {
                const size_t e1 = r->line_end_nomove(e0 + 1);   // end of the would-be sequence line
                if (e1 < r->len) {
                    r->line_end_nomove(e1 + 1);                 // make the first byte of the third line available
                    if (e1 + 1 < r->len) header = r->buf[e1 + 1] == '+';
                }
            }
            if (header) break;
            r->pos = e0 < r->len ? e0 + 1 : r->len;
        }
    }
    if (r->failed) {
        std::string m = r->fail_msg;
        bc_reader_close(r); *out = nullptr;
        set_error("%s: %s", path, m.c_str());
        return BC_EINVAL;
    }
    return BC_OK;
}

// offset, in the decoded stream, of the next byte bc_reader_next will look at (plain files: the file offset).
// Strictly increasing from chunk to chunk and by at least one per read: usable as a first-seen ordering key.
int bc_reader_tell(const bc_reader* r, uint64_t* offset_out) {
    if (!r || !offset_out) { set_error("bc_reader_tell: null argument"); return BC_EINVAL; }
    *offset_out = r->abs0 + r->pos;
    return BC_OK;
}

int bc_reader_close(bc_reader* r) {
    if (!r) return BC_OK;
    if (r->gz) gzclose(r->gz);
    if (r->fp) fclose(r->fp);
    if (r->zctx) g_zstd.freeDCtx(r->zctx);
    delete r;
    return BC_OK;
}

// The decoded byte stream itself, sequentially: up to `cap` bytes into dst, *got = 0 at the end of the stream.  What the
// device ingest (bc_stream_count_file on a .gz / .zst file) uploads in 16 MiB chunks: the line rules then run on the
// device.  Not to be mixed with bc_reader_next on one reader.  A damaged or truncated stream is an error.
int bc_reader_read_raw(bc_reader* r, char* dst, uint64_t cap, uint64_t* got) {
    if (!r || !dst || !got) { set_error("bc_reader_read_raw: null argument"); return BC_EINVAL; }
    uint64_t have = 0;
    while (have < cap && !r->eof) {
        const size_t n = r->fill(dst + have, (size_t)std::min<uint64_t>(cap - have, 1u << 30));
        if (n == 0) { r->eof = true; break; }
        have += n;
    }
    if (r->failed) {
        *got = 0;
        set_error("reads file: %s", r->fail_msg.c_str());
        return BC_EINVAL;
    }
    *got = have;
    return BC_OK;
}

// Up to max_reads sequence lines (stripped) into text (capacity text_cap) with
// offsets off[0..n]; stops early when the next read would not fit (a single
// read longer than text_cap is BC_ECAP).  *n_out = 0 at end of file.
int bc_reader_next(bc_reader* r, uint64_t max_reads, char* text, uint64_t text_cap, uint64_t* off,
                   uint64_t* n_out) {
    if (!r || !text || !off || !n_out) { set_error("bc_reader_next: null argument"); return BC_EINVAL; }
    uint64_t n = 0, used = 0;
    off[0] = 0;
    const uint64_t per = r->kind ? 4 : 1, seq_line = r->kind ? 1 : 0;
    while (n < max_reads) {
        if (r->line_in_record == 0 && r->abs0 + r->pos >= r->end) break;  // the next record belongs to another range
        const char* nl = (const char*)memchr(r->buf.data() + r->pos, '\n', r->len - r->pos);
        while (!nl && !r->eof) {  // pull more decoded bytes behind the unconsumed tail
            if (r->pos > 0) {
                memmove(r->buf.data(), r->buf.data() + r->pos, r->len - r->pos);
                r->len -= r->pos;
                r->abs0 += r->pos;
                r->pos = 0;
            }
            if (r->len == r->buf.size()) r->buf.resize(r->buf.size() * 2);  // one very long line
            size_t got = r->fill(r->buf.data() + r->len, r->buf.size() - r->len);
            if (got == 0) { r->eof = true; break; }
            nl = (const char*)memchr(r->buf.data() + r->len, '\n', got);
            r->len += got;
        }
        if (!nl && r->pos == r->len) break;  // nothing left
        const char* line = r->buf.data() + r->pos;
        const size_t ll = nl ? (size_t)(nl - line) : (r->len - r->pos);
        if (r->line_in_record == seq_line) {
            size_t a = 0, b = ll;  // python str.strip(): whitespace off both ends
            while (a < b && (line[a] == ' ' || (line[a] >= '\t' && line[a] <= '\r'))) ++a;
            while (b > a && (line[b - 1] == ' ' || (line[b - 1] >= '\t' && line[b - 1] <= '\r'))) --b;
            if (used + (b - a) > text_cap) {
                if (n == 0) { set_error("bc_reader_next: a read of %zu bytes does not fit the buffer", b - a); return BC_ECAP; }
                break;  // not consumed: the next call starts with this line
            }
            memcpy(text + used, line + a, b - a);
            used += b - a;
            off[++n] = used;
        }
        r->line_in_record = (r->line_in_record + 1) % per;
        r->pos += ll + (nl ? 1 : 0);
    }
    if (r->failed) {  // a damaged / truncated stream is an error, not a short file
        *n_out = 0;
        set_error("reads file: %s", r->fail_msg.c_str());
        return BC_EINVAL;
    }
    *n_out = n;
    return BC_OK;
}

}  // extern "C"
