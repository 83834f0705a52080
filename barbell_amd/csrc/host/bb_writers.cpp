// bb_writers.cpp — see bb_writers.hpp.
#include "bb_writers.hpp"
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

LabelWriters::LabelWriters(std::string f, bool g, unsigned k) : folder(std::move(f)), gz(g) {
    if (k == 0) { const char* e = getenv("BARBELL_AMD_WRITERS"); k = e ? (unsigned)std::max(1, atoi(e)) : 8u; }
    for (unsigned i = 0; i < k; ++i) threads.emplace_back([this]() { run(); });
}

void LabelWriters::run() {
    std::vector<uint8_t> buf;
    for (;;) {
        LabelQ* L;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return stop || !ready.empty(); });
            if (ready.empty()) return;
            L = ready.front();
            ready.pop_front();
            L->listed = false; L->busy = true;
        }
        for (;;) {
            Item it;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (L->q.empty()) { L->busy = false; break; }
                it = std::move(L->q.front());
                L->q.pop_front();
            }
            try {
                if (it.sp.cut) {
                    const Cut& c = *it.sp.cut;
                    if (buf.size() < it.sp.n) buf.resize(it.sp.n + it.sp.n / 4);
                    size_t total = 0;
                    for (uint64_t k = it.sp.first; k < it.sp.first + it.sp.n_records; ++k) {
                        const bb_slice& sl = c.plan.slices[k];
                        const size_t got = render_trim_record(buf.data() + (sl.out_off - it.sp.off), c.text, c.plan, sl, c.cfg);
                        if (got != sl.rec_len) throw BarbellError(BB_E_INVALID, "internal: a record cut on the host differs in length from the GPU's plan");
                        total += got;
                    }
                    if (total != it.sp.n) throw BarbellError(BB_E_INVALID, "internal: records of a label do not fill its span");
                    write(*L, it.sp.label, buf.data(), it.sp.n);
                } else write(*L, it.sp.label, it.sp.p, it.sp.n);
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(mu);
                if (err.empty()) err = e.what();
            }
            it.sp.keep.reset();
            it.sp.cut.reset();
            if (it.left->fetch_sub(1) == 1) {
                { std::lock_guard<std::mutex> lk(mu); --outstanding; }
                cv.notify_all();
            }
        }
    }
}

void LabelWriters::wait(size_t max_outstanding) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return outstanding <= max_outstanding; });
    if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
}

void LabelWriters::submit(std::vector<Span> job) {
    if (job.empty()) return;
    auto left = std::make_shared<std::atomic<int>>((int)job.size());
    {
        std::lock_guard<std::mutex> lk(mu);
        ++outstanding;
        for (auto& sp : job) {
            auto& slot = labels[sp.label];
            if (!slot) slot = std::make_unique<LabelQ>();
            LabelQ* L = slot.get();
            L->q.push_back(Item{std::move(sp), left});
            if (!L->busy && !L->listed) { L->listed = true; ready.push_back(L); }   // a busy label's thread finds the new span itself
        }
    }
    cv.notify_all();
}

void LabelWriters::write(LabelQ& L, const std::string& label, const uint8_t* p, size_t n) {
    const std::string path = folder + "/" + label + (gz ? ".trimmed.fastq.gz" : ".trimmed.fastq");
    if (gz && LibDeflate::compressor()) {
        // a span = one gzip member appended to the label's file (concatenated members are one gzip file): libdeflate compresses 2-3 x as fast
        // as zlib at the same level, and the writer threads are what `kit --gzip` waits for
        if (!L.plain) {
            L.plain = fopen(path.c_str(), "wb");
            if (!L.plain) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
            setvbuf(L.plain, nullptr, _IONBF, 0);
        }
        static thread_local RawBuf zb;
        void* c = LibDeflate::compressor();
        size_t o = 0;
        do {   // (an empty span still leaves a member: the file is a gzip file from its first write on, as with gzopen)
            const size_t chunk = std::min<size_t>(n - o, 256u << 20);
            zb.reserve(LibDeflate::get().gzip_bound(c, chunk) + 64);
            const size_t z = LibDeflate::get().gzip_compress(c, p + o, chunk, zb.data(), zb.cap);
            if (!z || fwrite(zb.data(), 1, z, L.plain) != z) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
            o += chunk;
        } while (o < n);
    } else if (gz) {
        if (!L.gzf) {
            L.gzf = gzopen(path.c_str(), "wb");
            if (!L.gzf) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
        }
        for (size_t o = 0; o < n;) {
            const unsigned chunk = (unsigned)std::min<size_t>(n - o, 1u << 30);
            if (gzwrite(L.gzf, p + o, chunk) <= 0) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
            o += chunk;
        }
    } else {
        if (!L.plain) {
            L.plain = fopen(path.c_str(), "wb");
            if (!L.plain) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
            setvbuf(L.plain, nullptr, _IONBF, 0);  // spans are large and contiguous: straight to write()
        }
        if (n && fwrite(p, 1, n, L.plain) != n) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
    }
}

LabelWriters::~LabelWriters() {
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    for (auto& t : threads) if (t.joinable()) t.join();  // they drain the ready list first
    for (auto& kv : labels) {
        if (kv.second->gzf) gzclose(kv.second->gzf);
        if (kv.second->plain) fclose(kv.second->plain);
    }
}

}  // namespace barbell
