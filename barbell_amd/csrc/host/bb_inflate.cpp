// bb_inflate.cpp — see bb_inflate.hpp: the thread pool that turns gzip files and pipes into record-aligned pieces of text.
#include "bb_inflate.hpp"

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

size_t count_nl(const uint8_t* p, size_t n) {
    size_t c = 0;
    const uint8_t* e = p + n;
    while (p < e) {
        const void* q = memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        ++c;
        p = (const uint8_t*)q + 1;
    }
    return c;
}

namespace {
struct ParallelInflater : GzInflater {
    std::vector<std::string> paths;
    std::vector<char> gz;    // files that are not gzip are skipped (the feeder reads them directly)
    struct FileState { std::deque<std::shared_ptr<GzPiece>> ready; int state = 0; /* 0 not started, 1 being inflated, 2 all pieces made */ size_t out = 0; /* pieces made, not yet consumed */ };
    std::vector<FileState> fs;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv;
    size_t next_file = 0, consumed_upto = 0, ahead;
    size_t piece_bytes = 256u << 20, pieces_ahead = 3;
    std::string err;
    bool cancelled = false;
    std::shared_ptr<BufPool> bufs = std::make_shared<BufPool>();
    unsigned n_threads_total = 1; size_t n_gz_files = 0;
    uint64_t range_bytes = 32u << 20;   // compressed bytes per range of the member-parallel inflate; BARBELL_AMD_GZ_RANGE (tests) fixes it and scales the limits with it
    bool range_cap = false;
    std::atomic<uint64_t> n_ranges_parallel{0};
    uint64_t n_pieces = 0, held = 0, max_held = 0;   // pieces made; bytes of inflated text made and not yet consumed, and the most there ever was (BARBELL_AMD_PROFILE)
    ParallelInflater(std::vector<std::string> p, std::vector<char> is_gz, unsigned n_threads)
        : paths(std::move(p)), gz(std::move(is_gz)), fs(paths.size()) {
        if (const char* e = getenv("BARBELL_AMD_GZ_PIECE")) piece_bytes = (size_t)std::max(64L, atol(e));   // tests: pieces of a few hundred bytes
        if (const char* e = getenv("BARBELL_AMD_GZ_RANGE")) { range_bytes = (uint64_t)std::max(64L, atol(e)); range_cap = true; }  // tests: ranges of a few hundred bytes; 0 threads' worth: BARBELL_AMD_GZ_SERIAL
        n_threads_total = getenv("BARBELL_AMD_GZ_SERIAL") ? 1u : std::max(1u, n_threads);
        for (char g : gz) n_gz_files += g ? 1 : 0;
        const unsigned nt = std::max(1u, std::min<unsigned>(n_threads, (unsigned)paths.size()));
        ahead = nt + 2;  // files being inflated or inflated and not yet consumed: bounds the memory
        for (unsigned i = 0; i < nt; ++i) pool.emplace_back([this]() { work(); });
    }
    // hands a piece over; waits while the file has pieces_ahead of them unconsumed.  false: cancelled
    bool publish(size_t i, std::shared_ptr<GzPiece> pc) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return cancelled || !err.empty() || fs[i].out < pieces_ahead; });
        if (cancelled || !err.empty()) return false;
        ++n_pieces; held += pc->size; max_held = std::max(max_held, held);
        fs[i].ready.push_back(std::move(pc)); ++fs[i].out;
        lk.unlock();
        cv.notify_all();
        return true;
    }
    // Inflated text in, record-aligned pieces out: a piece is published when piece_bytes of text are there, cut after the last whole record
    // (the file starts with one, so the cut is after line 4 * floor(lines / 4) of what the piece holds); a record larger than a piece makes it grow.
    struct PieceSink {
        ParallelInflater& P; size_t file;
        RawBuf buf; size_t n = 0; bool any = false;
        bool no_fast = false;   // a member of this file did not fit libdeflate's buffers: zlib for the rest (inflate_member)
        PieceSink(ParallelInflater& p, size_t f) : P(p), file(f) {}
        size_t first_cap() const { return std::max<size_t>(64, std::min<size_t>(P.piece_bytes, 4u << 20)); }   // (small files do not pay for a piece-sized buffer)
        bool emit(size_t cut) {   // buf[0, cut) goes out as a piece, the rest starts the next
            auto pc = std::make_shared<GzPiece>();
            pc->pool = P.bufs;
            RawBuf next;
            const size_t rest = n - cut;
            if (!P.bufs->take(next, std::max<size_t>(P.piece_bytes, rest + 64))) next.reserve(std::max(first_cap(), rest + 64));
            if (rest) memcpy(next.data(), buf.data() + cut, rest);
            pc->size = cut; pc->data.swap(buf);
            buf.swap(next);
            any = true; n = rest;
            return P.publish(file, std::move(pc));
        }
        bool append(const uint8_t* p, size_t len) {
            while (len) {
                if (n == buf.cap) {
                    const size_t full = std::max<size_t>(P.piece_bytes, 64);
                    if (buf.cap < full) { buf.reserve(buf.cap ? std::min(full, buf.cap * 2) : first_cap()); continue; }
                    const size_t lines = count_nl(buf.data(), n);
                    if (lines < 4) { buf.reserve(buf.cap * 2); continue; }   // a record larger than the piece
                    const uint8_t* end = buf.data() + n;
                    for (size_t k = 0; k <= lines % 4; ++k) {   // back over the partial last line and the lines % 4 whole ones after the last record
                        end = (const uint8_t*)memrchr(buf.data(), '\n', (size_t)(end - buf.data()));   // not null: lines >= 4
                        if (k == lines % 4) ++end;                                                      // ... up to and including the record's last line end
                    }
                    if (!emit((size_t)(end - buf.data()))) return false;
                    continue;
                }
                const size_t take = std::min(len, buf.cap - n);
                memcpy(buf.data() + n, p, take); n += take; p += take; len -= take;
            }
            return true;
        }
        bool finish() {   // the end of the file: whatever is left (a last line without a line end, blank lines); an empty file gives one empty piece
            if (n == 0 && any) return true;
            auto pc = std::make_shared<GzPiece>();
            pc->pool = P.bufs;
            pc->size = n; pc->data.swap(buf);
            any = true; n = 0;
            return P.publish(file, std::move(pc));
        }
    };
    // ---- a regular gzip file of several MEMBERS (what `cat *.fastq.gz` and bgzip make; a sequencing run's files concatenated) on several cores.
    // A member's start cannot be read off the file, so the threads guess: the compressed bytes are cut into ranges, a range's thread looks for
    // the gzip magic from the range's start on and inflates what it finds, member after member, until one starts at or beyond the range's
    // end.  A false start fails within a few KB (or, with probability 2^-64, passes the member's CRC and length).  The coordinator takes the
    // ranges in order and accepts a range's text only if its first member starts exactly where the text accepted so far ended: every byte
    // of the file is then accounted for by members that inflated with their checksums right.  Anything else — a range whose chain does not
    // link up, a member too large to buffer (a single-member file), an error — and the rest of the file is inflated serially from the last
    // accepted position, in bounded pieces as before.
    struct GzRange {
        uint64_t a = 0, b = 0;             // members that START in [a, b)
        uint64_t first = UINT64_MAX, end = 0;   // compressed offsets: start of the first member found, end of the last one inflated
        RawBuf text; size_t text_n = 0;
        bool done = false, failed = false, garbage = false;   // garbage: what follows `end` is not a gzip member (trailing bytes: ignored, as gzread does)
    };
    static bool gz_magic(const uint8_t* p, uint64_t left) { return left >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 0xE0) == 0; }
    // one member at map[pos ..]: its text appended to `out` (or handed to `sink` as it comes); returns the compressed bytes it took, 0 on error
    // or when it grows beyond the limits (max_in compressed bytes / max_out bytes of text in `out`)
    static uint64_t inflate_member(const uint8_t* map, uint64_t size, uint64_t pos, RawBuf* out, size_t* out_n, PieceSink* sink, uint64_t max_in, uint64_t max_out) {
        // libdeflate first: the member's text into a buffer of guessed size (doubled while it says the space does not suffice)
        if (void* dec = (sink && sink->no_fast) ? nullptr : LibDeflate::decompressor()) {
            const uint64_t left = size - pos;
            const size_t in_n = (size_t)std::min<uint64_t>(left, out ? std::min<uint64_t>(max_in, 1ull << 40) + 65536 : (512ull << 20));
            static const uint64_t serial_cap = getenv("BARBELL_AMD_LIBDEFLATE_MAX") ? (uint64_t)atoll(getenv("BARBELL_AMD_LIBDEFLATE_MAX")) : (1024ull << 20);   // (tests: a small one)
            const uint64_t cap_out = out ? max_out + (16ull << 20) : serial_cap;
            static thread_local RawBuf tmp;   // (serial path)
            const size_t have = out ? *out_n : 0;
            RawBuf& dst = out ? *out : tmp;
            // (serial path: a member that does not fit 512 MB of input or 1 GiB of text is zlib's, and so is the rest of its file: a huge single-member
            // file pays for one failed attempt, ~2 s, not for one per member)
            const bool whole = true;
            for (uint64_t space = std::min<uint64_t>(cap_out, std::max<uint64_t>(16ull << 20, 6ull * std::min<uint64_t>(in_n, 64ull << 20))); whole; space = std::min(cap_out, space * 2)) {
                dst.reserve(have + (size_t)space);
                size_t a_in = 0, a_out = 0;
                const int r = LibDeflate::get().gzip_ex(dec, map + pos, in_n, dst.data() + have, (size_t)space, &a_in, &a_out);
                if (r == 0) {
                    if (out) { *out_n = have + a_out; return a_in; }
                    return sink->append(tmp.data(), a_out) ? a_in : 0;
                }
                if (r != 3 || space >= cap_out) break;   // bad data (or cut off by the window), or larger than what may be buffered: zlib decides
            }
            if (sink) sink->no_fast = true;
        }
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return 0;
        uint64_t in_done = 0;
        std::vector<uint8_t> tmp(sink ? (4u << 20) : 0);
        const size_t out0 = out ? *out_n : 0;
        uint64_t result = 0;
        for (;;) {
            if (zs.avail_in == 0) {
                const uint64_t left = size - pos - in_done;
                if (left == 0) break;                       // the file ends inside the member
                zs.next_in = const_cast<Bytef*>(map + pos + in_done);
                zs.avail_in = (uInt)std::min<uint64_t>(left, 1u << 30);
                in_done += zs.avail_in;
            }
            if (out) {
                if (*out_n - out0 > max_out) break;
                const size_t have = *out_n;
                if (out->cap - have < (1u << 20)) out->reserve(std::max<size_t>(out->cap + out->cap / 2, have + (16u << 20)));
                const uInt room = (uInt)std::min<size_t>(out->cap - have, 1u << 30);
                zs.next_out = out->data() + have; zs.avail_out = room;
                const int r = inflate(&zs, Z_NO_FLUSH);
                *out_n = have + (room - zs.avail_out);
                if (r == Z_STREAM_END) { result = in_done - zs.avail_in; break; }
                if (r != Z_OK && r != Z_BUF_ERROR) break;
            } else {
                zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
                const int r = inflate(&zs, Z_NO_FLUSH);
                if (!sink->append(tmp.data(), tmp.size() - zs.avail_out)) break;
                if (r == Z_STREAM_END) { result = in_done - zs.avail_in; break; }
                if (r != Z_OK && r != Z_BUF_ERROR) break;
            }
            if (in_done - zs.avail_in > max_in) break;
        }
        inflateEnd(&zs);
        if (!result && out) *out_n = out0;
        return result;
    }
    void range_work(const uint8_t* map, uint64_t size, GzRange& R, uint64_t range_bytes) {
        uint64_t p = R.a;
        // a member beyond these is left to the serial path (a single-member file: the ranges' threads would buffer all of it)
        const uint64_t max_in = range_cap ? 4 * range_bytes : (256ull << 20), max_out = range_cap ? 64 * range_bytes : (1536ull << 20);
        // the first member: the first candidate from the range's start on that inflates to its end with its checksum right
        while (p < R.b && p < size) {
            const uint8_t* q = (const uint8_t*)memchr(map + p, 0x1f, (size_t)std::min<uint64_t>(R.b, size) - p);
            if (!q) { p = R.b; break; }
            p = (uint64_t)(q - map);
            if (gz_magic(map + p, size - p)) {
                const uint64_t took = inflate_member(map, size, p, &R.text, &R.text_n, nullptr, max_in, max_out);
                if (took) { R.first = p; p += took; break; }
                if (R.a == 0 && p == 0) { R.failed = true; return; }   // the file's own first member does not inflate: leave it to the serial path and its error message
            }
            ++p;
        }
        if (R.first == UINT64_MAX) { R.end = R.a; return; }
        // the chain: member after member until one starts at or beyond the range's end
        while (p < R.b && p < size) {
            if (!gz_magic(map + p, size - p)) { R.garbage = true; break; }
            const uint64_t took = inflate_member(map, size, p, &R.text, &R.text_n, nullptr, max_in, max_out);
            if (!took) { R.failed = true; break; }
            p += took;
            if (R.text_n > max_out) { R.failed = true; break; }
        }
        R.end = p;
    }
    // the rest of the file from compressed offset pos, member after member on this thread, text straight into the sink; "" or an error message
    std::string inflate_serial(const uint8_t* map, uint64_t size, uint64_t pos, PieceSink& sink, const std::string& path) {
        while (pos < size) {
            if (!gz_magic(map + pos, size - pos)) {
                if (pos == 0) return "Error reading FASTQ file '" + path + "'";
                break;   // trailing bytes that are no gzip member: ignored (gzread does the same)
            }
            const uint64_t took = inflate_member(map, size, pos, nullptr, nullptr, &sink, UINT64_MAX, 0);
            if (!took) {
                std::lock_guard<std::mutex> lk(mu);
                return cancelled ? std::string() : "Error reading FASTQ file '" + path + "' (gzip data corrupt or truncated)";
            }
            pos += took;
        }
        return std::string();
    }
    std::string inflate_regular(size_t i, unsigned n_range_threads) {
        const int fd = open(paths[i].c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return "Failed to open FASTQ input: " + paths[i]; }
        const uint64_t size = (uint64_t)st.st_size;
        PieceSink sink(*this, i);
        if (size == 0) { close(fd); sink.finish(); return std::string(); }
        void* m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return "Failed to map FASTQ input: " + paths[i]; }
        (void)madvise(m, (size_t)size, MADV_SEQUENTIAL);
        const uint8_t* map = (const uint8_t*)m;
        uint64_t pos = 0;           // compressed bytes accounted for
        std::string e;
        const auto t_start = std::chrono::steady_clock::now();
        // ranges of 4 .. 32 MB of compressed bytes, two per thread for a mid-sized file; files under 16 MB are not worth the threads
        const uint64_t RB = range_cap ? range_bytes : std::min<uint64_t>(32u << 20, std::max<uint64_t>(4u << 20, size / (2ull * std::max(1u, n_range_threads))));
        if (n_range_threads >= 2 && size >= (range_cap ? 4 * RB : (16ull << 20))) {
            const size_t K = (size_t)((size + RB - 1) / RB);
            std::vector<std::unique_ptr<GzRange>> rg(K);
            std::mutex rmu; std::condition_variable rcv;
            size_t next_job = 0, taken_upto = 0; bool quit = false;
            const size_t window = n_range_threads + 2;   // ranges inflated or being inflated beyond the one the coordinator waits for: bounds the memory
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_range_threads; ++t)
                th.emplace_back([&]() {
                    for (;;) {
                        size_t k;
                        {
                            std::unique_lock<std::mutex> lk(rmu);
                            rcv.wait(lk, [&]() { return quit || (next_job < K && next_job < taken_upto + window); });
                            if (quit) return;
                            k = next_job++;
                            rg[k] = std::make_unique<GzRange>();
                            rg[k]->a = (uint64_t)k * RB; rg[k]->b = std::min<uint64_t>(size, (uint64_t)(k + 1) * RB);
                        }
                        range_work(map, size, *rg[k], RB);
                        { std::lock_guard<std::mutex> lk(rmu); rg[k]->done = true; }
                        rcv.notify_all();
                    }
                });
            bool ok = true, ended = false;
            for (size_t k = 0; k < K && ok && !ended; ++k) {
                std::unique_ptr<GzRange> R;
                {
                    std::unique_lock<std::mutex> lk(rmu);
                    rcv.wait(lk, [&]() { return rg[k] && rg[k]->done; });
                    R = std::move(rg[k]);
                    taken_upto = k + 1;
                }
                rcv.notify_all();
                if (R->failed) { ok = false; break; }
                if (R->first == UINT64_MAX) { ok = pos >= R->b; continue; }   // no member starts here: fine if the chain so far reaches past the range
                if (R->first != pos) { ok = false; break; }                         // does not link up with what has been accepted
                if (!sink.append(R->text.data(), R->text_n)) { ok = false; ended = true; break; }
                pos = R->end;
                ++n_ranges_parallel;
                if (R->garbage) ended = true;
            }
            { std::lock_guard<std::mutex> lk(rmu); quit = true; }
            rcv.notify_all();
            for (auto& t : th) t.join();
            if (ended && ok) pos = size;   // trailing bytes after the last member: ignored
        }
        { std::lock_guard<std::mutex> lk(mu); if (cancelled) { munmap(m, (size_t)size); close(fd); return std::string(); } }
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: '%s': %llu of %llu compressed bytes inflated by ranges in %.3f s\n", paths[i].c_str(), (unsigned long long)pos, (unsigned long long)size,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        if (pos < size) e = inflate_serial(map, size, pos, sink, paths[i]);
        munmap(m, (size_t)size); close(fd);
        if (e.empty()) sink.finish();
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: '%s' inflated in %.3f s (its thread's time, waits for the consumer included)\n", paths[i].c_str(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        return e;
    }
    std::string inflate_stream(size_t i) {   // pipes, process substitutions, /dev/stdin: sequentially through gzread (plain text passes through)
        gzFile f = gzopen(paths[i].c_str(), "rb");
        if (!f) return "Failed to open FASTQ input: " + paths[i];
        gzbuffer(f, 1 << 20);
        PieceSink sink(*this, i);
        std::vector<uint8_t> tmp(4u << 20);
        std::string e;
        for (;;) {
            const int r = gzread(f, tmp.data(), (unsigned)tmp.size());
            if (r < 0) { e = "Error reading FASTQ file '" + paths[i] + "'"; break; }
            if (r == 0) break;
            if (!sink.append(tmp.data(), (size_t)r)) break;
        }
        gzclose(f);
        if (e.empty()) sink.finish();
        return e;
    }
    void work() {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this]() { return cancelled || next_file >= paths.size() || next_file < consumed_upto + ahead || !err.empty(); });
                if (cancelled || next_file >= paths.size() || !err.empty()) return;
                i = next_file++;
                if (!gz[i]) { fs[i].state = 2; continue; }
                fs[i].state = 1;
            }
            struct stat pst;
            const bool regular = stat(paths[i].c_str(), &pst) == 0 && S_ISREG(pst.st_mode);
            // threads for one file's ranges: all of them for a single gzip input, fewer where several files are being inflated side by side
            const unsigned per_file = std::max(1u, n_threads_total / (unsigned)std::max<size_t>(1, std::min<size_t>(n_gz_files, n_threads_total)));
            const std::string e = regular ? inflate_regular(i, per_file) : inflate_stream(i);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty() && err.empty()) err = e;
                fs[i].state = 2;
            }
            cv.notify_all();
        }
    }
    // the file's next piece, in order; blocks until it is there.  nullptr: the file has no more (an empty file gives one empty piece first)
    std::shared_ptr<GzPiece> next_piece(size_t i) override {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return cancelled || !err.empty() || !fs[i].ready.empty() || fs[i].state == 2; });
        if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
        if (cancelled) throw BarbellError(BB_E_INVALID, "cancelled");
        if (fs[i].ready.empty()) {   // every piece has been handed out: the next file may start (its pieces already made stay until consumed)
            consumed_upto = std::max(consumed_upto, i + 1);
            lk.unlock();
            cv.notify_all();
            return nullptr;
        }
        auto pc = fs[i].ready.front();
        fs[i].ready.pop_front();
        return pc;
    }
    // every chunk of a piece has been copied out: its memory goes with the last reference, the file may make another
    void piece_consumed(size_t i, size_t bytes) override {
        { std::lock_guard<std::mutex> lk(mu); --fs[i].out; held -= bytes; }
        cv.notify_all();
    }
    // wakes everything that waits here (workers, and readers inside next_piece): called before the feeder joins its readers
    void cancel() override {
        {
            std::lock_guard<std::mutex> lk(mu);
            next_file = paths.size();
            cancelled = true;
        }
        cv.notify_all();
    }
    ~ParallelInflater() override {
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: gzip / pipe input inflated in %llu piece(s) of at most %zu bytes; at most %llu bytes of inflated text held at once; %llu range(s) of members inflated side by side\n",
                    (unsigned long long)n_pieces, piece_bytes, (unsigned long long)max_held, (unsigned long long)n_ranges_parallel.load());
        if (getenv("BARBELL_AMD_PROFILE")) fprintf(stderr, "profile: gzip members inflated with %s\n", LibDeflate::get().gzip_ex ? "libdeflate (zlib for members too large to buffer)" : "zlib");
        cancel();
        for (auto& t : pool) if (t.joinable()) t.join();
    }
};
}  // namespace

std::unique_ptr<GzInflater> GzInflater::make(std::vector<std::string> paths, std::vector<char> is_gz, unsigned n_threads) {
    return std::make_unique<ParallelInflater>(std::move(paths), std::move(is_gz), n_threads);
}

}  // namespace barbell
