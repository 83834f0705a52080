// bb_inflate.hpp — gzip input and output of the host (round 6: split from bb_host.cpp, no behaviour change): libdeflate bound at run time where the
// system has it, buffers without a zero fill, the record-aligned PIECES a gzip file / pipe is inflated into, and the interface of the thread pool that
// inflates several files — and the members of one multi-member file — side by side (bb_inflate.cpp).  The reference reads gzip through paraseq /
// niffler (src/io/io.rs:29-33) on the calling thread; here zlib's ~0.3 GB/s per core would otherwise be what the GPU waits for.
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bb_host.hpp"

namespace barbell {

size_t count_nl(const uint8_t* p, size_t n);   // line ends in [p, p + n)

// libdeflate, where the system has it (dlopen: no build dependency; BARBELL_AMD_NO_LIBDEFLATE=1 switches it off): inflates a gzip member
// whose text fits a buffer 4-5 x as fast as zlib (320 MB of FASTQ text in 32 members: 0.41 against 1.84 s on one core).  It has no
// streaming form, so members too large to buffer — and systems without the library — take zlib as before.
struct LibDeflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;   // 0 ok, 1 bad data, 3 insufficient space
    // the other direction (the per-label files of --gzip): a span of records -> one gzip member
    void* (*calloc_)(int) = nullptr;
    void (*cfree)(void*) = nullptr;
    size_t (*gzip_bound)(void*, size_t) = nullptr;
    size_t (*gzip_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    static const LibDeflate& get() {
        static const LibDeflate L = []() {
            LibDeflate l;
            if (getenv("BARBELL_AMD_NO_LIBDEFLATE")) return l;
            void* h = nullptr;
            for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) return l;
            l.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
            l.release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
            l.gzip_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_gzip_decompress_ex");
            if (!l.alloc || !l.release || !l.gzip_ex) l.gzip_ex = nullptr;
            l.calloc_ = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
            l.cfree = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
            l.gzip_bound = (size_t (*)(void*, size_t))dlsym(h, "libdeflate_gzip_compress_bound");
            l.gzip_compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_gzip_compress");
            if (!l.calloc_ || !l.cfree || !l.gzip_bound || !l.gzip_compress) l.gzip_compress = nullptr;
            return l;
        }();
        return L;
    }
    struct Comp { void* c = nullptr; ~Comp() { if (c) LibDeflate::get().cfree(c); } };
    static void* compressor() {   // one per thread, level 6 (zlib's default, what gzopen("wb") writes with)
        static thread_local Comp t;
        if (!t.c && get().gzip_compress) t.c = get().calloc_(6);
        return t.c;
    }
    struct Dec { void* d = nullptr; ~Dec() { if (d) LibDeflate::get().release(d); } };
    static void* decompressor() {   // one per thread
        static thread_local Dec t;
        if (!t.d && get().gzip_ex) t.d = get().alloc();
        return t.d;
    }
};
// bytes without the zero-fill a std::vector pays on every growth (the inflaters write every byte they count; at GB/s the fill was a third of the time)
struct RawBuf {
    uint8_t* p = nullptr; size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    ~RawBuf() { free(p); }
    void reserve(size_t c) {   // keeps the contents
        if (c <= cap) return;
        void* q = realloc(p, c);
        if (!q) throw BarbellError(BB_E_NOMEM, "out of memory (inflated text)");
        p = (uint8_t*)q; cap = c;
    }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    void swap(RawBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); }
};
// piece buffers go round: a fresh 256 MiB buffer costs 65 K page faults on first touch (a third of the coordinator's time per piece), a used one none
struct BufPool {
    std::mutex mu;
    std::vector<std::unique_ptr<RawBuf>> spare;
    void give(RawBuf& b) {
        if (b.cap < (16u << 20)) return;
        std::lock_guard<std::mutex> lk(mu);
        if (spare.size() >= 8) return;
        spare.emplace_back(new RawBuf());
        spare.back()->swap(b);
    }
    bool take(RawBuf& into, size_t want) {   // a spare buffer of at least `want` bytes, if there is one
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < spare.size(); ++i)
            if (spare[i]->cap >= want) { into.swap(*spare[i]); spare.erase(spare.begin() + (long)i); return true; }
        return false;
    }
};
struct GzPiece {
    RawBuf data; size_t size = 0; uint64_t chunks_left = 0;
    std::shared_ptr<BufPool> pool;
    ~GzPiece() { if (pool) pool->give(data); }
};
// gzip files (and pipes): zlib inflates one stream on one core (~0.3 GB/s), far below what the GPU takes, so a pool of threads inflates
// several files at once, a few files ahead of the consumer, and hands them over in input order (the TSV keeps the reads' order).  A file
// comes as PIECES of at most `piece_bytes` of text, each cut after a whole record (the file starts with one, so the cut is after line
// 4 * floor(lines / 4) of what has been read): the feeder treats a piece like a small file of its own.  Round 5: before, a file was inflated
// whole — a 50 GB fastq.gz meant 200 GB of text in memory (and its buffer's last doubling as much again); now a file holds at most
// `pieces_ahead` pieces whatever its size.  n_threads comes from -t/--threads like the reference's worker count.
// Inflates the gzip files of a run (files that are not gzip are skipped: the feeder reads them directly) a few files ahead of the consumer and hands
// every file over as pieces, in input order.  bb_inflate.cpp.
struct GzInflater {
    virtual ~GzInflater() = default;
    // the file's next piece, in order; blocks until it is there.  nullptr: the file has no more (an empty file gives one empty piece first)
    virtual std::shared_ptr<GzPiece> next_piece(size_t file) = 0;
    // every chunk of a piece has been copied out: its memory goes with the last reference, the file may make another
    virtual void piece_consumed(size_t file, size_t bytes) = 0;
    // wakes everything that waits here (workers, and readers inside next_piece): called before the feeder joins its readers
    virtual void cancel() = 0;
    static std::unique_ptr<GzInflater> make(std::vector<std::string> paths, std::vector<char> is_gz, unsigned n_threads);
};

}  // namespace barbell
