// bb_feed.hpp — FASTQ files -> blocks of whole records in upload buffers, in stream order (round 6: split from bb_host.cpp, no behaviour change):
// what paraseq's reader + `process_parallel` fan-out are to the reference (src/io/io.rs:29-33, annotator.rs:245-280).  The reader threads, the
// sequencer, the two-line / packed staging forms, byte-range shards of one plain file (`--shard-by bytes`): bb_feed.cpp.
#pragma once
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bb_host.hpp"
#include "bb_inflate.hpp"

namespace barbell {

// the automatic flank cutoff (edit_model.rs:2-11) is applied inside bb_create when k_cutoff is unset
// Raw FASTQ text in blocks of whole records, in page-locked memory, in stream order.
//
// Chunks of `chunk` bytes are read by a pool of reader threads (pread at fixed offsets for plain files, a copy out of
// the inflated image for gzip files — ParallelInflater below) into a ring of page-locked slots, each with HEAD bytes of
// headroom in front; every reader also counts its chunk's line ends.  A sequencer (next(), one caller) takes the chunks
// in order and turns them into blocks that hold complete 4-line records only: with the running number of complete lines
// it knows how many trailing lines of a chunk belong to a record that ends in the next chunk, finds that cut by walking
// back over those few lines, and copies the short tail into the next slot's headroom — exactly the `consumed` the GPU
// parser (bb_fastq_ingest) would have reported, without waiting for it, so block i+1 can go to another GPU while block i
// is still being parsed.  The last block of a file is handed over whole (the parser's final-block rules apply to it).
// What a reader reports about the lines of one chunk it compacted (two-line mode), so that the sequencer can check what the GPU parser
// checks in the 4-line form — a record's sequence and quality lines are equally long, the file ends on a record boundary — although the
// quality lines never leave the host.  Pairs that lie inside the chunk are compared by the reader; what crosses a chunk boundary is
// stitched from these fields, in order (BlockFeeder::stitch).  Line lengths exclude the line end ("\n" or "\r\n").
struct TwoLineSummary {
    size_t head_raw = 0; uint8_t head_last = 0;   // bytes before the chunk's first '\n' (the whole chunk if it has none), the last of them
    size_t tail_raw = 0; uint8_t tail_last = 0;   // bytes after its last '\n' (a line that ends in a later chunk, or at the end of the file)
    int64_t len1 = -1, len2 = -1;                 // lengths of the second and third line that END in the chunk
    int64_t pend = -1;                            // length of the last sequence line, not the chunk's first line, whose quality line does not end in the chunk
    bool pend_cleared = false;                    // some quality line other than the chunk's first two lines ends in the chunk: nothing older is pending after it
    int64_t last3[3] = {-1, -1, -1};              // lengths of the last three lines that end in the chunk (-2: that line is the chunk's first)
};

// thrown by the sequencer when a chunk cannot be staged in the packed form (two adjacent non-IUPAC characters in a read, a gzip chunk whose
// line layout the reader could not tell): annotate() starts over with the plain two-line form
struct PackFallback {};

struct BlockFeeder {
    struct Block { uint64_t index = 0; const uint8_t* data = nullptr; size_t len = 0; int slot = -1; std::shared_ptr<std::vector<uint8_t>> big; };
    struct Task { size_t file = 0; uint64_t off = 0; size_t len = 0; uint64_t seq = 0; bool last = false; std::shared_ptr<GzPiece> piece; };   // piece: gzip input, off within it
    struct Slot { uint8_t* p = nullptr; size_t cap = 0, got = 0, nl = 0; bool last = false; int state = 0; uint64_t seq = 0; int refs = 0; size_t file = 0;
                  // two-line mode: raw newlines of the chunk, the phase (line index mod 4) the reader took its first byte to be in
                  // (-1: none recognisable), where the raw bytes came from (to redo the chunk if the guess was wrong), malformed flag
                  size_t raw_nl = 0; int phase0 = 0; uint64_t off = 0; size_t raw_len = 0; bool bad = false; TwoLineSummary sum;
                  bool unpackable = false; };   // packed staging: this chunk cannot be packed (PackCtx::unpackable, or no look-back possible)
    size_t HEAD = 16u << 20;  // BARBELL_AMD_HEAD_BYTES overrides it (tests of the over-long-carry path)
    int device;                // the slots are page-locked for uploads to this device; a slot is allocated by the first reader that fills it
    // Ordinary (pageable, huge-page advised) memory by default: measured on the MI355X box the runtime uploads from it as fast as from
    // page-locked memory (10.0 M reads/s steady state either way, 2 contexts, 128 MiB blocks) and page-locking 2-6 GB cost 0.4-1.0 s of a
    // 1.6-2.9 s run, serialised inside the runtime against the contexts being created.  BARBELL_AMD_PINNED_SLOTS=1: hipHostMalloc.
    bool pageable = getenv("BARBELL_AMD_PINNED_SLOTS") == nullptr;
    bool keep_slots = false;   // the process is about to exit: the destructor leaves the slots to the OS (unpinning 6 GB costs ~0.5 s)
    std::vector<std::string> paths;
    std::vector<char> is_gz;
    std::vector<int> fds;
    std::vector<const uint8_t*> maps; // two-line mode, plain files: the file mapped (the readers compact out of the page cache)
    std::vector<uint64_t> sizes;      // plain: st_size; gzip: inflated size once known
    // the part of each file this process stages: [begins, ends) — the whole file, or (--shard R/W --shard-by bytes) the records that START in
    // the R-th of W equal byte ranges of a plain file: both ends are record starts, found by the same rule from either side (record_start)
    std::vector<uint64_t> begins, ends;
    uint32_t shard_rank = 0, shard_world = 1;   // byte-range sharding (1: off)
    std::vector<char> size_known;
    size_t chunk;
    std::vector<Slot> slots;
    std::vector<std::thread> readers;
    std::unique_ptr<GzInflater> inflater;
    std::mutex mu;
    std::condition_variable cv;
    // claim cursor
    size_t cur_file = 0; uint64_t cur_off = 0, next_seq = 0;
    std::shared_ptr<GzPiece> cur_piece;   // gzip input: the piece being chunked (claim)
    bool piece_fetching = false;                 // a reader is waiting for the inflater's next piece; the others wait for that reader
    bool stop = false, claims_done = false;
    std::string err;
    // sequencer state
    uint64_t want_seq = 0, n_blocks = 0;
    const uint8_t* carry_ptr = nullptr; size_t carry_len = 0, carry_lines = 0; int carry_slot = -1;
    std::vector<uint8_t> carry_buf;  // a carry that spans whole chunks (a record longer than a chunk) is kept here
    bool done = false;
    // Two-line mode (annotate without the trim step: annotator.rs:125-127 never looks at the quality line): the readers drop the
    // '+' and quality lines while they stage a chunk, so half the bytes cross PCIe and the GPU parses 2-line records
    // (BB_FASTQ_TWO_LINE).  Dropping lines is a pure per-byte filter on "index of the byte's line mod 4", so the compacted chunks
    // concatenate to the compacted stream; a reader only has to know the phase of its chunk's first byte.  It reads it off the
    // text ("@..." two lines above "+..."; a sequence line cannot start with '+', so the test is unambiguous for FASTQ) and the
    // sequencer, which knows the true phase from the running line count, checks every guess and redoes a chunk that was wrong.
    bool two_line = false;
    bool pack = false;                    // two-line mode with the sequence lines packed two bases per byte (PackCtx); needs the raw text in memory
    size_t lpr = 4;                       // lines per record in the staged text
    size_t seq_file = (size_t)-1; uint64_t seq_raw_lines = 0;   // sequencer: file in hand, its raw lines so far
    // stitch(): a line in progress across chunk ends, the sequence length waiting for its quality line, the last two lines' lengths
    size_t st_part = 0; uint8_t st_part_last = 0; int64_t st_pend = -1, st_last3[3] = {-1, -1, -1};
    void stitch(const Slot& sl, int ph0);

    BlockFeeder(int device_, const std::vector<std::string>& files, size_t chunk_bytes, unsigned n_slots, unsigned n_readers, unsigned n_inflate,
                bool two_line_mode = false, bool pack_mode = false, uint32_t byte_shard_rank = 0, uint32_t byte_shard_world = 1);
    ~BlockFeeder();
    static uint64_t record_start(int fd, uint64_t size, uint64_t pos, const std::string& path);
    // where in its line byte `off` of a file lies: the readers look back for the line's start (mapped file / inflated image)
    static size_t line_pos(const uint8_t* file_base, uint64_t off) {
        const void* q = off ? memrchr(file_base, '\n', (size_t)off) : nullptr;
        return q ? (size_t)(file_base + off - ((const uint8_t*)q + 1)) : (size_t)off;
    }
    void reader_loop();
    bool claim(Task& t);
    bool next(Block& b);
    void release(int slot);
    void unref(int slot);
    void fail(const std::string& e) { { std::lock_guard<std::mutex> lk(mu); if (err.empty() && !e.empty()) err = e; stop = true; } cv.notify_all(); }
};

}  // namespace barbell
