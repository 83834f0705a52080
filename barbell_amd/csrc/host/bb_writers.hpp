// bb_writers.hpp — the per-label FASTQ files of the trim step (round 6: split from bb_host.cpp, no behaviour change): trim_matches' writers
// (src/trim/trim.rs:356-446) as a pool of threads that write — and, with host_cut, first cut — the records of a block.  bb_writers.cpp.
#pragma once
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bb_host.hpp"

namespace barbell {

struct LabelWriters {  // the per-label writers of trim_matches (trim.rs:356-446)
    std::string folder;
    bool gz;
    // Writes run on K threads of their own (one write() stream moves ~3-6 GB/s of page cache, the GPU plans or renders records many
    // times faster).  A label's file is worked on by one thread at a time and its spans in submission order, so the records of a file
    // stay in batch order and no two threads share a handle; WHICH thread is decided when the work is there (labels with queued spans
    // wait in `ready`, a free thread takes the next one and drains it), so that no thread idles while another has twelve labels of
    // a block to itself.  A batch is done when the last of its spans is written; at most `max_outstanding` batches exist (wait()),
    // which bounds the text held in page-locked buffers.
    // A span is either bytes to write as they are (rendered on the GPU) or, with `cut`, the records slices[first, first + n_records) to be cut out
    // of the block's own text first (host_cut): the thread renders them into its buffer — n bytes, laid out by the plan's offsets — and writes that
    struct Cut { const uint8_t* text; TrimPlan plan; bb_trim_config cfg; std::shared_ptr<void> hold; };
    struct Span { std::string label; const uint8_t* p; size_t n; std::shared_ptr<void> keep; std::shared_ptr<const Cut> cut; uint64_t first = 0, off = 0; uint32_t n_records = 0; };
    struct Item { Span sp; std::shared_ptr<std::atomic<int>> left; };
    struct LabelQ {
        std::deque<Item> q;
        bool busy = false, listed = false;
        gzFile gzf = nullptr;
        FILE* plain = nullptr;
    };
    std::map<std::string, std::unique_ptr<LabelQ>> labels;
    std::deque<LabelQ*> ready;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    size_t outstanding = 0;  // batches submitted and not yet fully written
    std::string err;
    LabelWriters(std::string f, bool g, unsigned k = 0);
    ~LabelWriters();
    // waits until at most `max_outstanding` batches are queued or being written, then rethrows a writer error if any
    void wait(size_t max_outstanding);
    void submit(std::vector<Span> job);
private:
    void run();
    void write(LabelQ& L, const std::string& label, const uint8_t* p, size_t n);
};

}  // namespace barbell
