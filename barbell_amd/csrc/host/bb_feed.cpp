// bb_feed.cpp — see bb_feed.hpp.
#include "bb_feed.hpp"
#include "../bb_pack.h"

#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace barbell {


// Packed staging (BB_FASTQ_PACKED, include/barbell_amd_fastq.h): the kernels only ever look at a read character's IUPAC base set, so the
// readers keep exactly that — two 4-bit codes per byte — and the sequence lines cross PCIe at half their size again (~2 KB per 4-kb read;
// annotator.rs:125-127 hands `demux` the bytes, nothing downstream of it reports them).  Pairs are aligned to the START of a line, so that
// the packed chunks still concatenate to the packed stream: a reader knows where in its line the chunk's first byte lies (it looks back in
// the mapped file / inflated image for the line's start), leaves a first byte at an odd position to the chunk before, and pairs a last
// unpaired base with the first byte of the chunk after (two bytes of look-ahead tell a base from a line end).
struct PackCtx {
    size_t line_pos0 = 0;      // index within its line of the chunk's first byte (used when that byte lies in a sequence line)
    size_t after = 0;          // bytes readable beyond the chunk's end (0: the chunk ends the file)
    bool prev_is_cr = false;   // the byte before the chunk is '\r' (a chunk that starts with the '\n' of a "\r\n": the '\r' is not a base of the line)
    bool prev_line_blank = false;   // the line before the one the chunk starts in is empty (blank lines after a file's last record that straddle a chunk start)
    // from the raw text around a chunk that starts at file_base + off
    void look_back(const uint8_t* file_base, uint64_t off) {    // line_pos0 is set
        prev_is_cr = off > 0 && file_base[off - 1] == '\r';
        prev_line_blank = false;
        const uint64_t ls = off - line_pos0;                       // start of the line the chunk begins in
        if (ls > 0 && file_base[ls - 1] == '\n') {
            uint64_t e = ls - 1;                                    // the '\n' that ends the line before
            if (e > 0 && file_base[e - 1] == '\r') --e;
            prev_line_blank = e == 0 || file_base[e - 1] == '\n';
        }
    }
    bool unpackable = false;   // out: two adjacent non-IUPAC characters would pack to '\n'; the run falls back to the plain two-line form
};
// keeps the bytes of the lines in phase 0 and 1 (header, sequence) of a chunk whose first byte lies in a line of phase ph0; in
// place when out == buf (the write position never passes the read position), or straight from a mapping of the file.  nl_kept / nl_all: newlines kept / seen; bad: a line that starts
// inside the chunk in phase 0 / 2 does not start with '@' / '+'.
// pk != nullptr: sequence lines packed (never in place: out and buf must not overlap; buf[n .. n + pk->after) must be readable).
static size_t compact_two_line(uint8_t* out, const uint8_t* buf, size_t n, int ph0, size_t& nl_kept, size_t& nl_all, bool& bad, TwoLineSummary& S,
                               PackCtx* pk = nullptr) {
    size_t d = 0, p = 0;
    // the header line of the record in hand was empty (blank lines after the last record stay blank lines); for a chunk that starts with a
    // sequence-phase line the reader has looked at the line before
    bool hdr_blank = pk && (ph0 & 3) == 1 && pk->prev_line_blank;
    int hdr_state = pk && (ph0 & 3) == 1 ? (pk->prev_line_blank ? 1 : 0) : -1;   // the same for the '+' line's check, in every mode: -1 = began before the chunk, not known
    int ph = ph0 & 3;
    bool line_start = false;  // the first line may be the tail of one that began in the previous chunk
    nl_kept = nl_all = 0; bad = false;
    S = TwoLineSummary();
    int64_t seq_len = -1;     // the sequence line of the record in hand, if it began in this chunk after the first line
    while (p < n) {
        if (line_start && ph == 0) hdr_state = (buf[p] == '\n' || (buf[p] == '\r' && p + 1 < n && buf[p + 1] == '\n')) ? 1 : 0;
        if (line_start && ((ph == 0 && buf[p] != '@' && buf[p] != '\n' && buf[p] != '\r') ||
                           (ph == 2 && buf[p] != '+' && !(hdr_state != 0 && (buf[p] == '\n' || buf[p] == '\r'))))) bad = true;   // a blank '+' line: only below a blank header (blank lines after the last record)
        const uint8_t* q = (const uint8_t*)memchr(buf + p, '\n', n - p);
        const size_t e = q ? (size_t)(q - buf) + 1 : n;
        if (q) {
            const size_t raw = (size_t)(q - buf) - p;   // without the '\n'
            if (nl_all == 0) { S.head_raw = raw; S.head_last = raw ? buf[p + raw - 1] : 0; S.last3[2] = -2; }
            else {
                const int64_t len = (int64_t)raw - (raw && buf[p + raw - 1] == '\r' ? 1 : 0);
                if (nl_all == 1) S.len1 = len;
                if (nl_all == 2) S.len2 = len;
                if (ph == 1) seq_len = len;
                if (ph == 3) {
                    if (seq_len >= 0) { if (seq_len != len) bad = true; }   // both lines of the pair inside the chunk
                    if (nl_all >= 3 || seq_len >= 0) S.pend_cleared = true;
                    seq_len = -1;
                }
                S.last3[0] = S.last3[1]; S.last3[1] = S.last3[2]; S.last3[2] = len;
            }
        } else { S.tail_raw = n - p; S.tail_last = buf[n - 1]; if (nl_all == 0) { S.head_raw = n - p; S.head_last = buf[n - 1]; } }
        if (ph == 1 && pk) {
            // the line's bases inside the chunk: [p, se); a '\r' belongs to the line end if a '\n' (or the end of the file) follows it
            size_t se = q ? (size_t)(q - buf) : n;
            if (se > p && buf[se - 1] == '\r' && (q || pk->after == 0 || buf[n] == '\n')) --se;
            const size_t i0 = nl_all == 0 ? pk->line_pos0 : 0;    // where in its line the segment starts
            size_t b = p;
            if ((i0 & 1u) && se > b) ++b;                          // an odd first base went into the last pair of the chunk before
            uint8_t next = 15;                                     // pairs with a last unpaired base: nothing, unless the line goes on in the next chunk
            if (!q && ((se - b) & 1u) && se == n && pk->after > 0) {
                const bool eol = buf[n] == '\n' || (buf[n] == '\r' && (pk->after == 1 || buf[n + 1] == '\n'));
                if (!eol) next = base_code_table()[buf[n]];
            }
            const size_t cr_split = (nl_all == 0 && i0 > 0 && q == buf + p && pk->prev_is_cr) ? 1 : 0;   // "\r" | "\n" split over two chunks
            if (i0 - cr_split == 0 && se == p && q && hdr_blank) {  // blank line after a blank header line: not a record, stays as it is ("\n" or "\r\n")
                if (cr_split) out[d++] = '\r';                      // (its '\r' ended the chunk before, which left it to this one)
                memcpy(out + d, buf + p, e - p); d += e - p; ++nl_kept;
            } else {
                d += pack_bases(out + d, buf + b, buf + se, next, pk->unpackable);
                if (q || pk->after == 0) {                          // the line ends here (its '\n', or the end of a file without one): parity terminator
                    const size_t cr_before = (nl_all == 0 && i0 > 0 && q == buf + p && pk->prev_is_cr) ? 1 : 0;   // "\r" | "\n" split over two chunks
                    out[d++] = ((i0 - cr_before + (se - p)) & 1u) ? 'O' : 'E';
                    if (q) { out[d++] = '\n'; ++nl_kept; }
                }
            }
        } else if (ph < 2) {
            if (ph == 0 && pk) {   // a header line without a character — counting what the chunk before holds of it (at most the '\r' of its "\r\n")
                const size_t raw = q ? (size_t)(q - buf) - p : 1;
                const size_t len = raw - ((raw && buf[p + raw - 1] == '\r') ? 1 : 0);
                const size_t before = nl_all == 0 ? pk->line_pos0 - ((pk->line_pos0 == 1 && raw == 0 && pk->prev_is_cr) ? 1 : 0) : 0;
                hdr_blank = q && len == 0 && before == 0;
            }
            if (out + d != buf + p) memmove(out + d, buf + p, e - p);
            d += e - p;
            if (q) ++nl_kept;
        }
        if (q) { ++nl_all; ph = (ph + 1) & 3; line_start = true; }
        p = e;
    }
    S.pend = seq_len;
    return d;
}
// The sequencer's half of the two-line mode's record checks: the chunks' summaries in stream order (their first byte in phase ph0 of its file).
void BlockFeeder::stitch(const Slot& sl, int ph0) {
    const TwoLineSummary& S = sl.sum;
    auto fail = [&](const char* what) {
        throw BarbellError(BB_E_FASTQ, "Input FASTQ parsing failed: '" + paths[sl.file] + "' " + what);
    };
    if (sl.raw_nl == 0) {   // no line ends here: the chunk continues the line in progress
        if (sl.raw_len) { st_part += S.head_raw; st_part_last = S.head_last; }
    } else {
        const size_t raw0 = st_part + S.head_raw;
        const uint8_t last0 = S.head_raw ? S.head_last : st_part_last;
        const int64_t L0 = (int64_t)raw0 - (raw0 && last0 == '\r' ? 1 : 0);
        if (ph0 == 1) st_pend = L0;
        if (ph0 == 3) { if (st_pend >= 0 && st_pend != L0) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (ph0 == 2 && sl.raw_nl >= 2) { if (st_pend >= 0 && st_pend != S.len1) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (ph0 == 1 && sl.raw_nl >= 3) { if (L0 != S.len2) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (S.pend_cleared) st_pend = -1;
        if (S.pend >= 0) st_pend = S.pend;
        // the last two lines that have ended, for the check at the end of the file
        const int64_t a = S.last3[0] == -2 ? L0 : S.last3[0], b = S.last3[1] == -2 ? L0 : S.last3[1], c = S.last3[2] == -2 ? L0 : S.last3[2];
        if (sl.raw_nl >= 3) { st_last3[0] = a; st_last3[1] = b; st_last3[2] = c; }
        else if (sl.raw_nl == 2) { st_last3[0] = st_last3[2]; st_last3[1] = b; st_last3[2] = c; }
        else { st_last3[0] = st_last3[1]; st_last3[1] = st_last3[2]; st_last3[2] = c; }
        st_part = S.tail_raw; st_part_last = S.tail_last;
    }
    if (sl.last) {  // the file's end: a last line without '\n' counts; blank lines may follow the last record (the GPU parser ignores them: whole blank records, then the surplus lines)
        uint64_t lines = seq_raw_lines + sl.raw_nl;
        int64_t tail_len = -1;
        if (st_part) { tail_len = (int64_t)st_part - (st_part_last == '\r' ? 1 : 0); ++lines; }
        const int r = (int)(lines & 3u);
        if (r == 0) {
            if (tail_len >= 0 && st_pend >= 0 && st_pend != tail_len) fail("holds a record whose quality line is not as long as its sequence");
        } else {
            // r surplus lines: they must all be blank
            const int64_t l1 = tail_len >= 0 ? tail_len : st_last3[2], l2 = tail_len >= 0 ? st_last3[2] : st_last3[1], l3 = tail_len >= 0 ? st_last3[1] : st_last3[0];
            const bool blank = l1 == 0 && (r < 2 || l2 == 0) && (r < 3 || l3 == 0);
            if (!blank) fail("ends inside a record (truncated file?)");
        }
        st_part = 0; st_part_last = 0; st_pend = -1; st_last3[0] = st_last3[1] = st_last3[2] = -1;
    }
}
// phase of a chunk's first byte, read off the text: the first line that starts with '@' and has a line starting with '+' two
// lines below is a header (phase 0); -1 if no such pair is found among the chunk's first lines
static int guess_phase(const uint8_t* buf, size_t n, bool at_file_start) {
    if (at_file_start) return 0;
    size_t st[16], ns = 0, p = 0;
    while (ns < 16 && p < n) {
        const void* q = memchr(buf + p, '\n', n - p);
        if (!q) break;
        p = (size_t)((const uint8_t*)q - buf) + 1;
        if (p < n) st[ns++] = p;
    }
    for (size_t j = 0; j + 2 < ns; ++j)
        if (buf[st[j]] == '@' && buf[st[j + 2]] == '+') return (int)((4 - ((j + 1) & 3)) & 3);  // line j+1 of the chunk is in phase 0
    return -1;
}
static bool sniff_gzip(const std::string& path) {  // magic bytes, not the file name (the reference's reader sniffs too)
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + path);
    unsigned char m[2] = {0, 0};
    const size_t n = fread(m, 1, 2, f);
    fclose(f);
    return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

// First record start at or after byte `pos` of a plain FASTQ file (--shard-by bytes): the phase of the line `pos` lies in is read off the text
// as the readers do (guess_phase: a line that starts with '@' two lines above one that starts with '+'), then as many line ends are skipped
// as it takes to stand at the start of a header line.  Shard R ends where shard R + 1 begins: both call this with the same `pos`.
uint64_t BlockFeeder::record_start(int fd, uint64_t size, uint64_t pos, const std::string& path) {
    if (pos == 0) return 0;
    if (pos >= size) return size;
    const size_t want = (size_t)std::min<uint64_t>(size - (pos - 1), (32u << 20) + 1);   // from the byte before `pos` on
    std::vector<uint8_t> buf(want);
    size_t got = 0;
    while (got < want) {
        const ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)(pos - 1 + got));
        if (r < 0) { if (errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + path + "'"); }
        if (r == 0) break;
        got += (size_t)r;
    }
    if (got < 2) return size;
    const bool at_line_start = buf[0] == '\n';
    const uint8_t* w = buf.data() + 1;
    const size_t n = got - 1;
    const int ph = guess_phase(w, n, false);
    if (ph < 0) {
        if (pos - 1 + got >= size) return size;   // fewer than three line ends from here to the end of the file: the last record started earlier
        throw BarbellError(BB_E_FASTQ, "--shard-by bytes: no record boundary found in '" + path + "' within 32 MiB of byte " + std::to_string(pos));
    }
    if (at_line_start && ph == 0) return pos;
    size_t skip = (size_t)((4 - ph) & 3);
    if (skip == 0) skip = 4;          // inside a header line: the next record
    size_t p = 0;
    for (size_t k = 0; k < skip; ++k) {
        const void* q = memchr(w + p, '\n', n - p);
        if (!q) return size;          // the file ends first
        p = (size_t)((const uint8_t*)q - w) + 1;
    }
    return pos + p;
}

BlockFeeder::BlockFeeder(int device_, const std::vector<std::string>& files, size_t chunk_bytes, unsigned n_slots, unsigned n_readers, unsigned n_inflate,
                         bool two_line_mode, bool pack_mode, uint32_t byte_shard_rank, uint32_t byte_shard_world)
    : device(device_), paths(files), chunk(chunk_bytes), two_line(two_line_mode), pack(two_line_mode && pack_mode), lpr(two_line_mode ? 2 : 4) {
    shard_rank = byte_shard_rank; shard_world = std::max(1u, byte_shard_world);
    begins.assign(paths.size(), 0); ends.assign(paths.size(), 0);
    if (const char* e = getenv("BARBELL_AMD_HEAD_BYTES")) HEAD = (size_t)std::max(16L, atol(e));
    is_gz.resize(paths.size()); fds.assign(paths.size(), -1); sizes.assign(paths.size(), 0); size_known.assign(paths.size(), 0);
    maps.assign(paths.size(), nullptr);
    std::vector<std::string> gz_paths;
    for (size_t i = 0; i < paths.size(); ++i) {
        // Pipes, process substitutions and /dev/stdin have no size and cannot be read at offsets (and a sniff would eat their
        // first bytes): they are read sequentially, whole, through zlib like a gzip file — gzread passes plain text through
        // and inflates gzip, whichever arrives (the reference's paraseq reader streams both as well, io.rs:29-33).
        struct stat pst;
        if (stat(paths[i].c_str(), &pst) != 0) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + paths[i]);
        if (!S_ISREG(pst.st_mode)) { is_gz[i] = 1; }
        else is_gz[i] = sniff_gzip(paths[i]) ? 1 : 0;
        if (is_gz[i]) {
            if (shard_world > 1) throw BarbellError(BB_E_INVALID, "--shard-by bytes: '" + paths[i] + "' is gzip (or a pipe): only plain files can be cut by byte ranges; shard those by file");
            continue;
        }
        fds[i] = open(paths[i].c_str(), O_RDONLY);
        struct stat st;
        if (fds[i] < 0 || fstat(fds[i], &st) != 0) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + paths[i]);
        sizes[i] = (uint64_t)st.st_size; size_known[i] = 1;
        ends[i] = sizes[i];
        if (shard_world > 1) {   // this process's byte range of the file, widened to record starts
            begins[i] = record_start(fds[i], sizes[i], sizes[i] / shard_world * shard_rank, paths[i]);
            ends[i] = shard_rank + 1 == shard_world ? sizes[i] : record_start(fds[i], sizes[i], sizes[i] / shard_world * (shard_rank + 1), paths[i]);
            if (ends[i] < begins[i]) ends[i] = begins[i];   // (not for FASTQ text: record_start is monotone there)
        }
        if (two_line && st.st_size > 0 && !getenv("BARBELL_AMD_NO_MMAP")) {  // the readers compact straight out of the page cache: one pass over the text, no copy of the dropped half
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fds[i], 0);
            if (m != MAP_FAILED) {
                maps[i] = (const uint8_t*)m;
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            }
        }
    }
    for (size_t i = 0; i < paths.size(); ++i)
        if (!is_gz[i] && !maps[i] && sizes[i] > 0) pack = false;   // a plain file that could not be mapped is read with pread: no look-back, no packing
    bool any_gz = false;
    for (char g : is_gz) any_gz = any_gz || g;
    if (any_gz) inflater = GzInflater::make(paths, is_gz, n_inflate);
    if (!any_gz) {  // every size is known: no slot needs to be larger than the largest file (small inputs do not page-lock gigabytes)
        uint64_t mx = 4096;
        for (uint64_t z : sizes) mx = std::max(mx, z);
        chunk = (size_t)std::min<uint64_t>(chunk, (mx + 4095) & ~(uint64_t)4095);
        HEAD = std::min(HEAD, (chunk + 15) & ~(size_t)15);
    }
    // Page-locking is the expensive part of starting up (6 GB took a second, and as long again to undo): a slot is allocated by the reader
    // that fills it first — in parallel, while the contexts are being created and the first blocks are already on the GPU —, and a short
    // input never touches most of them.
    slots.resize(std::max(3u, n_slots));
    for (size_t i = 0; i < slots.size(); ++i) { slots[i].p = nullptr; slots[i].cap = HEAD + chunk; slots[i].seq = i; }
    for (unsigned i = 0; i < std::max(1u, n_readers); ++i) readers.emplace_back([this]() { reader_loop(); });
}
BlockFeeder::~BlockFeeder() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    if (inflater) inflater->cancel();   // a reader may be waiting for a piece of inflated text
    for (auto& t : readers) if (t.joinable()) t.join();
    inflater.reset();
    if (!keep_slots)
        for (auto& sl : slots) if (sl.p) { if (pageable) free(sl.p); else bb_host_free_on(device, sl.p); }
    if (!keep_slots) {
        for (size_t i = 0; i < maps.size(); ++i) if (maps[i]) munmap((void*)maps[i], (size_t)sizes[i]);
    } else {
        // The process is about to exit and would take the mappings down by itself — on ONE thread: 128 GB of FASTQ are 32 M page-table
        // entries, 1.0 s of a 2.6 s run on 16 M reads.  Thirty-two threads drop them side by side in a few tens of milliseconds (the page cache
        // keeps the data).
        std::vector<std::thread> zap;
        const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
        uint64_t total = 0;
        for (size_t i = 0; i < maps.size(); ++i) if (maps[i]) total += sizes[i];
        if (total >= (256u << 20)) {
            // ranges of at least 64 MiB, at most ~32 of them per run of threads, whatever the number of files
            const uint64_t part = std::max<uint64_t>(64u << 20, ((total / 32) + pg - 1) & ~(uint64_t)(pg - 1));
            std::vector<std::pair<const uint8_t*, uint64_t>> ranges;
            for (size_t i = 0; i < maps.size(); ++i) {
                if (!maps[i]) continue;
                for (uint64_t a0 = 0; a0 < sizes[i]; a0 += part) ranges.emplace_back(maps[i] + a0, std::min<uint64_t>(part, sizes[i] - a0));
            }
            std::atomic<size_t> next_range{0};
            const unsigned nt = (unsigned)std::min<size_t>(32, ranges.size());
            for (unsigned k = 0; k < nt; ++k)
                zap.emplace_back([&ranges, &next_range]() {
                    for (size_t r; (r = next_range.fetch_add(1)) < ranges.size();) (void)madvise((void*)ranges[r].first, (size_t)ranges[r].second, MADV_DONTNEED);
                });
            for (auto& t : zap) t.join();
            zap.clear();
        }
        for (auto& t : zap) t.join();
    }
    for (int fd : fds) if (fd >= 0) close(fd);
}
// next chunk of the stream; gzip files come as record-aligned pieces of inflated text (ParallelInflater), each chunked like a small file
bool BlockFeeder::claim(Task& t) {
    for (;;) {
        size_t f;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (stop) return false;
            if (cur_file >= paths.size()) { if (!claims_done) { claims_done = true; cv.notify_all(); } return false; }
            f = cur_file;
            if (!is_gz[f]) {
                if (cur_off < begins[f]) cur_off = begins[f];
                if (ends[f] == begins[f]) {  // empty file (or an empty byte range of one): an empty last chunk keeps the sequence simple
                    t = Task{f, begins[f], 0, next_seq++, true, nullptr};
                    ++cur_file; cur_off = 0;
                    return true;
                }
                const size_t len = (size_t)std::min<uint64_t>(chunk, ends[f] - cur_off);
                t = Task{f, cur_off, len, next_seq++, cur_off + len == ends[f], nullptr};
                cur_off += len;
                if (t.last) { ++cur_file; cur_off = 0; }
                return true;
            }
            if (cur_piece) {
                const size_t len = (size_t)std::min<uint64_t>(chunk, cur_piece->size - cur_off);
                t = Task{f, cur_off, len, next_seq++, cur_off + len == cur_piece->size, cur_piece};
                cur_off += len;
                if (t.last) { cur_piece.reset(); cur_off = 0; }
                return true;
            }
            if (piece_fetching) { cv.wait(lk, [&]() { return stop || !piece_fetching; }); continue; }
            piece_fetching = true;
        }
        std::shared_ptr<GzPiece> pc;
        std::string e;
        try { pc = inflater->next_piece(f); } catch (const std::exception& ex) { e = ex.what(); }   // blocks until inflated
        {
            std::lock_guard<std::mutex> lk(mu);
            piece_fetching = false;
            if (e.empty()) {
                if (!pc) { ++cur_file; cur_off = 0; }
                else { cur_piece = pc; cur_off = 0; pc->chunks_left = std::max<uint64_t>(1, (pc->size + chunk - 1) / chunk); }
            }
        }
        cv.notify_all();
        if (!e.empty()) throw BarbellError(BB_E_INVALID, e);
    }
}
void BlockFeeder::reader_loop() {
    try {
        Task t;
        while (claim(t)) {
            Slot& sl = slots[t.seq % slots.size()];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return stop || (sl.state == 0 && sl.seq == t.seq); });
                if (stop) return;
                sl.state = 1;
            }
            if (!sl.p) {  // first use of this slot (it is this reader's alone until it is marked full)
                void* q = nullptr;
                if (pageable) {
                    if (posix_memalign(&q, 2u << 20, sl.cap) != 0) throw BarbellError(BB_E_NOMEM, "out of memory (block buffer)");
                    (void)madvise(q, sl.cap, MADV_HUGEPAGE);
                } else if (bb_host_malloc_on(device, sl.cap, &q) != BB_OK) throw BarbellError(BB_E_NOMEM, "bb_host_malloc_on failed (page-locked block buffer)");
                sl.p = (uint8_t*)q;
            }
            uint8_t* dst = sl.p + HEAD;
            const uint8_t* src = nullptr;   // the chunk's raw bytes where they can be read in place (inflated image, mapped file)
            // the file the chunk lies in, for the look-ahead / look-back below: a plain file's range, or the piece of inflated text
            const uint64_t f_begin = t.piece ? 0 : begins[t.file], f_end = t.piece ? t.piece->size : ends[t.file], f_size = t.piece ? t.piece->size : sizes[t.file];
            if (t.piece) src = t.piece->data.data() + t.off;
            else if (maps[t.file]) {
                // a mapped file that has been truncated since it was opened would fault (SIGBUS) when its lost pages are touched: look at its
                // size again before every chunk and fail like the pread path does (a file cut while a chunk is being read is still a race)
                struct stat stn;
                if (fstat(fds[t.file], &stn) != 0 || (uint64_t)stn.st_size < t.off + t.len)
                    throw BarbellError(BB_E_INVALID, "FASTQ file '" + paths[t.file] + "' shrank while it was read");
                src = maps[t.file] + t.off;
            }
            static const bool map_populate = getenv("BARBELL_AMD_MAP_POPULATE") != nullptr;
            if (map_populate && src && maps[t.file] && t.len) {   // experiment: the chunk's pages mapped by one call instead of one fault per 64 KB
                const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
                const uintptr_t a0 = (uintptr_t)src & ~(pg - 1), a1 = ((uintptr_t)src + t.len + pg - 1) & ~(pg - 1);
                (void)madvise((void*)a0, (size_t)(a1 - a0), 22 /* MADV_POPULATE_READ */);
            }
            size_t got_len = t.len, nl = 0, raw_nl = 0;
            int ph0 = 0;
            bool bad = false;
            TwoLineSummary sum;
            bool unpackable = false;
            if (two_line && src) {
                // the phase is read off the first lines from the chunk's start; they may lie beyond its end (a chunk shorter than three lines):
                // the mapped file / inflated image can be read ahead
                ph0 = guess_phase(src, t.len + (size_t)std::min<uint64_t>(f_size - (t.off + t.len), 1u << 20), t.off == f_begin);   // (a shard's range, and a piece, begin at a record start)
                if (ph0 < 0 && t.off > f_begin) {   // too few lines from here to the end of the file: read the phase off the text BEFORE the chunk and count on
                    const uint64_t back = std::min<uint64_t>(t.off - f_begin, 4u << 20);
                    const uint8_t* w = src - back;
                    const int pw = guess_phase(w, (size_t)(f_size - (t.off - back)), t.off - back == f_begin);
                    if (pw >= 0) ph0 = (int)((pw + count_nl(w, (size_t)back)) & 3u);
                }
                if (ph0 >= 0) {
                    PackCtx pk;
                    if (pack) { pk.line_pos0 = line_pos(src - t.off, t.off); pk.after = (size_t)(f_end - (t.off + t.len)); pk.look_back(src - t.off, t.off); }
                    got_len = compact_two_line(dst, src, t.len, ph0, nl, raw_nl, bad, sum, pack ? &pk : nullptr);
                    unpackable = pk.unpackable;
                } else { if (t.len) memcpy(dst, src, t.len); raw_nl = count_nl(dst, t.len); }   // left raw: the sequencer compacts it with the true phase
            } else {
                if (src) { if (t.len) memcpy(dst, src, t.len); }
                else {
                    size_t got = 0;
                    while (got < t.len) {
                        const ssize_t r = pread(fds[t.file], dst + got, t.len - got, (off_t)(t.off + got));
                        if (r < 0) { if (errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + paths[t.file] + "'"); }
                        if (r == 0) throw BarbellError(BB_E_INVALID, "FASTQ file '" + paths[t.file] + "' shrank while it was read");
                        got += (size_t)r;
                    }
                }
                if (two_line) {
                    ph0 = guess_phase(dst, t.len, t.off == f_begin);
                    if (ph0 >= 0) got_len = compact_two_line(dst, dst, t.len, ph0, nl, raw_nl, bad, sum);
                    else raw_nl = count_nl(dst, t.len);
                } else nl = count_nl(dst, t.len);
            }
            static const bool map_drop = getenv("BARBELL_AMD_MAP_DROP") != nullptr;
            if (src && maps[t.file] && t.len && map_drop) {
                // BARBELL_AMD_MAP_DROP=1: this chunk's pages of the mapping dropped as soon as it is staged (a process that must not hold page-table
                // entries for the whole input).  Measured on 8 M reads: steady state 12.5 -> 11.3 M reads/s (the shoot-downs disturb the upload
                // threads); by default the mapping is taken down at the end instead, by all readers at once (~BlockFeeder)
                const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
                const uintptr_t a0 = ((uintptr_t)src + pg - 1) & ~(pg - 1), a1 = ((uintptr_t)src + t.len) & ~(pg - 1);
                if (a1 > a0) (void)madvise((void*)a0, (size_t)(a1 - a0), MADV_DONTNEED);
            }
            if (t.piece) {
                bool last_copy;
                { std::lock_guard<std::mutex> lk(mu); last_copy = --t.piece->chunks_left == 0; }
                if (last_copy) inflater->piece_consumed(t.file, t.piece->size);  // every chunk of the piece has been copied out
                t.piece.reset();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                sl.got = got_len; sl.nl = nl; sl.last = t.last; sl.file = t.file; sl.raw_nl = raw_nl; sl.phase0 = ph0; sl.off = t.off; sl.raw_len = t.len;
                sl.bad = bad; sl.sum = sum; sl.unpackable = unpackable; sl.state = 2;
            }
            cv.notify_all();
        }
    } catch (const std::exception& e) { fail(e.what()); }
}
void BlockFeeder::unref(int i) {
    bool freed = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        Slot& sl = slots[(size_t)i];
        if (--sl.refs == 0) { sl.state = 0; sl.seq += slots.size(); freed = true; }
    }
    if (freed) cv.notify_all();
}
void BlockFeeder::release(int slot) { if (slot >= 0) unref(slot); }

bool BlockFeeder::next(Block& b) {
    for (;;) {
        if (done) return false;
        Slot* sl;
        {
            std::unique_lock<std::mutex> lk(mu);
            sl = &slots[want_seq % slots.size()];
            cv.wait(lk, [&]() { return !err.empty() || (sl->state == 2 && sl->seq == want_seq) || (claims_done && want_seq >= next_seq); });
            if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
            if (!(sl->state == 2 && sl->seq == want_seq)) { done = true; return false; }  // every chunk has been sequenced
            sl->refs = 2;  // the worker that uploads the block + the sequencer (its tail is the next block's carry)
        }
        const int si = (int)(want_seq % slots.size());
        ++want_seq;
        uint8_t* body = sl->p + HEAD;
        if (two_line) {   // the reader's guess of the chunk's first phase against the running line count of the file
            if (sl->file != seq_file) { seq_file = sl->file; seq_raw_lines = 0; }
            const int truth = (int)(seq_raw_lines & 3u);
            if (sl->phase0 != truth && pack) {
                // packed staging: the chunk is redone from the mapped file under the true phase (look-back and look-ahead need the file); a
                // gzip image may be gone by now: the run then falls back to the plain two-line form
                if (is_gz[sl->file] || !maps[sl->file]) sl->unpackable = true;
                else {
                    PackCtx pk;
                    pk.line_pos0 = line_pos(maps[sl->file], sl->off); pk.after = (size_t)(ends[sl->file] - (sl->off + sl->raw_len));
                    pk.look_back(maps[sl->file], sl->off);
                    sl->got = compact_two_line(body, maps[sl->file] + sl->off, sl->raw_len, truth, sl->nl, sl->raw_nl, sl->bad, sl->sum, &pk);
                    sl->unpackable = pk.unpackable;
                }
            } else if (sl->phase0 != truth) {
                if (sl->phase0 >= 0) {  // compacted under a wrong phase: the raw bytes are needed again
                    if (is_gz[sl->file]) throw BarbellError(BB_E_FASTQ, "'" + paths[sl->file] + "': line layout not recognised while dropping quality lines; rerun with --no-compact");
                    if (maps[sl->file]) memcpy(body, maps[sl->file] + sl->off, sl->raw_len);
                    else {
                        size_t got = 0;
                        while (got < sl->raw_len) {
                            const ssize_t r = pread(fds[sl->file], body + got, sl->raw_len - got, (off_t)(sl->off + got));
                            if (r <= 0) { if (r < 0 && errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + paths[sl->file] + "'"); }
                            got += (size_t)r;
                        }
                    }
                }
                sl->got = compact_two_line(body, body, sl->raw_len, truth, sl->nl, sl->raw_nl, sl->bad, sl->sum);
            }
            if (sl->unpackable) {
                if (getenv("BARBELL_AMD_PROFILE")) fprintf(stderr, "profile: chunk at %llu of '%s' (%zu bytes, phase guessed %d, true %d) has no packed form\n",
                                                           (unsigned long long)sl->off, paths[sl->file].c_str(), sl->raw_len, sl->phase0, truth);
                throw PackFallback();
            }
            if (sl->bad) throw BarbellError(BB_E_FASTQ, "Input FASTQ parsing failed: '" + paths[sl->file] + "' holds a record that is not a 4-line FASTQ record");
            stitch(*sl, truth);
            seq_raw_lines += sl->raw_nl;
        }
        size_t cut = sl->got;  // bytes of this chunk that go into this block
        size_t lines_left = 0;
        if (!sl->last) {
            const size_t total = carry_lines + sl->nl;
            const size_t r = total % lpr;          // complete lines after the last complete record
            if (total < lpr || sl->nl <= r) {       // no record ends inside this chunk (a record longer than the block, or a tiny --block-bytes):
                // the whole chunk joins the carry, kept aside, and the next chunk continues the record
                std::vector<uint8_t> nb(carry_len + sl->got);
                if (carry_len) memcpy(nb.data(), carry_ptr, carry_len);
                if (sl->got) memcpy(nb.data() + carry_len, body, sl->got);
                carry_buf.swap(nb);
                if (carry_slot >= 0) unref(carry_slot);
                carry_slot = -1;
                carry_ptr = carry_buf.data(); carry_len = carry_buf.size(); carry_lines = total;
                unref(si); unref(si);  // neither a worker nor the sequencer keeps the slot
                continue;
            }
            // the cut is just after line end number (nl - r) of the chunk: walk back over the partial last line and r lines
            const uint8_t* e = body + sl->got;
            for (size_t k = 0; k <= r; ++k) {
                const void* q = memrchr(body, '\n', (size_t)(e - body));
                e = (const uint8_t*)q;  // not null: nl > r
            }
            cut = (size_t)(e - body) + 1;
            lines_left = r;
        }
        // assemble: carry (in the previous slot's tail, or aside) + chunk[0, cut)
        Block out;
        if (carry_len > HEAD) {  // a carry longer than the headroom (a huge record): assemble aside
            auto big = std::make_shared<std::vector<uint8_t>>(carry_len + cut);
            memcpy(big->data(), carry_ptr, carry_len);
            memcpy(big->data() + carry_len, body, cut);
            out.data = big->data(); out.len = big->size(); out.big = big; out.slot = -1;
            unref(si);  // the worker does not need the slot
        } else {
            if (carry_len) memcpy(body - carry_len, carry_ptr, carry_len);
            out.data = body - carry_len; out.len = carry_len + cut; out.slot = si;
        }
        if (carry_slot >= 0) unref(carry_slot);  // the previous slot's tail has been copied
        // the new carry
        carry_ptr = body + cut; carry_len = sl->got - cut; carry_lines = lines_left;
        if (carry_len) carry_slot = si;
        else { carry_slot = -1; unref(si); }
        if (sl->last) { carry_lines = 0; }
        if (out.len == 0) { if (out.slot >= 0) unref(out.slot); continue; }  // an empty file
        out.index = n_blocks++;
        b = out;
        return true;
    }
}

// `barbell-amd stage` (tests, no GPU): the text the reader threads and the sequencer stage for upload — the blocks of whole records, one after
// the other — written to a file.  Returns the form that was staged: 4 (4-line text), 2 (two-line), 1 (two-line, sequence lines packed); a
// packed run that meets input without a packed form falls back to the two-line form like annotate() does.
int stage_blocks(const std::vector<std::string>& read_files, size_t block_bytes, unsigned n_threads, bool two_line, bool pack, const std::string& out_path,
                 size_t& n_blocks, uint32_t byte_shard_rank, uint32_t byte_shard_world) {
    for (int attempt = 0;; ++attempt) {
        FILE* f = fopen(out_path.c_str(), "wb");
        if (!f) throw BarbellError(BB_E_INVALID, "Failed to create '" + out_path + "'");
        n_blocks = 0;
        try {
            BlockFeeder feeder(-1, read_files, std::max<size_t>(block_bytes, 16), 8, std::max(1u, n_threads), std::max(1u, n_threads), two_line, pack,
                               byte_shard_rank, byte_shard_world);
            const bool packed = feeder.pack;
            BlockFeeder::Block b;
            while (feeder.next(b)) {
                if (b.len && fwrite(b.data, 1, b.len, f) != b.len) { fclose(f); throw BarbellError(BB_E_INVALID, "write failed"); }
                ++n_blocks;
                feeder.release(b.slot);
            }
            fclose(f);
            return packed ? 1 : (two_line ? 2 : 4);
        } catch (const PackFallback&) {
            fclose(f);
            if (attempt) throw BarbellError(BB_E_INVALID, "staging failed twice");
            pack = false;
        } catch (...) { fclose(f); throw; }
    }
}

}  // namespace barbell
