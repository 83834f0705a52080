// bb_rendezvous.cpp — see bb_rendezvous.hpp.
#include "bb_rendezvous.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "bb_host.hpp"

namespace barbell {
namespace {
constexpr size_t kNonce = 16;
using Clock = std::chrono::steady_clock;

// whole-file write made visible atomically (tmp + rename): a reader never sees half a file
void publish_file(const std::string& path, const void* head, size_t head_bytes, const void* data, size_t bytes) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    const bool ok = f && (!head_bytes || fwrite(head, 1, head_bytes, f) == head_bytes) && (!bytes || fwrite(data, 1, bytes, f) == bytes);
    if (!f || !ok || fclose(f) != 0) { if (f && !ok) fclose(f); throw BarbellError(BB_E_INVALID, "--rccl-id: cannot write '" + tmp + "'"); }
    if (rename(tmp.c_str(), path.c_str()) != 0) throw BarbellError(BB_E_INVALID, "--rccl-id: cannot publish '" + path + "'");
}
// the whole file if it exists and has exactly `bytes` bytes; false = not there (yet), or a file of another size: not this run's (a run of
// another --shard W or of other queries left it; its owner replaces it when it gets here) — *odd says which
bool read_file(const std::string& path, size_t bytes, std::vector<char>& out, bool* odd = nullptr) {
    if (odd) *odd = false;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    out.assign(bytes + 1, 0);
    const size_t got = fread(out.data(), 1, bytes + 1, f);
    fclose(f);
    if (got != bytes) { if (odd) *odd = true; return false; }
    out.resize(bytes);
    return true;
}
void fresh_nonce(char* out) {
    bool ok = false;
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { ok = read(fd, out, kNonce) == (ssize_t)kNonce; close(fd); }
    if (!ok) {   // pid, wall clock and monotonic clock: distinct per process start
        uint64_t w[2] = {(uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)std::chrono::system_clock::now().time_since_epoch().count(),
                         (uint64_t)Clock::now().time_since_epoch().count()};
        memcpy(out, w, kNonce);
    }
}
}  // namespace

double Rendezvous::timeout_s() { const char* e = getenv("BARBELL_AMD_RCCL_TIMEOUT"); return e && atof(e) > 0 ? atof(e) : 600.0; }

std::string Rendezvous::part(int rank, const char* what) const {
    return rank < 0 ? base_ + "." + what : base_ + ".r" + std::to_string(rank) + "." + what;
}

void Rendezvous::hello(const std::string& base, uint32_t rank) {
    for (const char* w : {"hello", "info", "counts", "done"}) (void)unlink((base + ".r" + std::to_string(rank) + "." + w).c_str());
    if (rank == 0) (void)unlink((base + ".id").c_str());
    char nonce[kNonce];
    fresh_nonce(nonce);
    publish_file(base + ".r" + std::to_string(rank) + ".hello", nullptr, 0, nonce, kNonce);
}

Rendezvous::Rendezvous(std::string base, uint32_t rank, uint32_t world) : base_(std::move(base)), rank_(rank), world_(world) {}

// the W hellos as they are now; false (and the first missing file's name) while some rank has not started
bool Rendezvous::read_identity(std::vector<char>& out, std::string& missing) const {
    out.assign(world_ * kNonce, 0);
    std::vector<char> one;
    for (uint32_t r = 0; r < world_; ++r) {
        if (!read_file(part((int)r, "hello"), kNonce, one)) { missing = part((int)r, "hello"); return false; }
        memcpy(out.data() + r * kNonce, one.data(), kNonce);
    }
    return true;
}

std::vector<ShardInfo> Rendezvous::meet(const ShardInfo& me) {
    const auto t0 = Clock::now();
    std::vector<char> published;   // the identity this rank's info was last written under
    std::vector<ShardInfo> all(world_);
    std::string waiting_for;
    {   // this process must have said hello itself (Rendezvous::hello at program start): its nonce is what makes the identity this run's
        std::vector<char> mine;
        if (!read_file(part((int)rank_, "hello"), kNonce, mine))
            throw BarbellError(BB_E_INVALID, "--rccl-id: '" + part((int)rank_, "hello") + "' is missing: Rendezvous::hello was not called at program start");
    }
    for (;;) {
        std::vector<char> id;
        bool complete = read_identity(id, waiting_for);
        if (complete) {
            if (id != published) { publish_file(part((int)rank_, "info"), id.data(), id.size(), &me, sizeof(me)); published = id; }
            std::vector<char> b;
            for (uint32_t r = 0; r < world_ && complete; ++r) {
                const std::string p = part((int)r, "info");
                bool odd = false;
                if (!read_file(p, id.size() + sizeof(ShardInfo), b, &odd)) { complete = false; waiting_for = odd ? p + " (the file there has another size: a stale file of a run with another --shard W?)" : p; break; }
                if (memcmp(b.data(), id.data(), id.size()) != 0) { complete = false; waiting_for = p + " (it carries another run's identity: a stale file of an interrupted run, or its writer has not caught up yet)"; break; }
                memcpy(&all[r], b.data() + id.size(), sizeof(ShardInfo));
                if (all[r].n_counts != me.n_counts || all[r].world != world_ || all[r].rank != r)
                    throw BarbellError(BB_E_INVALID, "--rccl-id: shard " + std::to_string(r) + " runs other queries or another --shard W (histogram of " +
                                                         std::to_string(all[r].n_counts) + " slots, W = " + std::to_string(all[r].world) + ")");
            }
            if (complete) {
                // (a rank that restarted meanwhile would have changed its hello: look once more before trusting the set)
                std::vector<char> again;
                std::string dummy;
                if (read_identity(again, dummy) && again == id) { identity_ = id; return all; }
            }
        }
        if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s())
            throw BarbellError(BB_E_INVALID, "--rccl-id: timed out after " + std::to_string((long)timeout_s()) + " s waiting for '" + waiting_for +
                                                 "' (is every shard of the run started with the same --rccl-id and --shard R/W?  BARBELL_AMD_RCCL_TIMEOUT=seconds waits longer)");
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}

void Rendezvous::publish(const char* what, const void* data, size_t bytes, int of_rank) {
    publish_file(part(of_rank, what), identity_.data(), identity_.size(), data, bytes);
}

std::vector<char> Rendezvous::await(const char* what, size_t bytes, int of_rank) {
    const auto t0 = Clock::now();
    const std::string p = part(of_rank, what);
    std::string why = "'" + p + "'";
    for (;;) {
        std::vector<char> b;
        bool odd = false;
        if (read_file(p, identity_.size() + bytes, b, &odd)) {
            if (memcmp(b.data(), identity_.data(), identity_.size()) == 0) return std::vector<char>(b.begin() + (long)identity_.size(), b.end());
            why = "'" + p + "' (the file there carries another run's identity: stale)";
        } else if (odd) why = "'" + p + "' (the file there has another size: stale, or shards with different queries)";
        if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s())
            throw BarbellError(BB_E_INVALID, "--rccl-id: timed out after " + std::to_string((long)timeout_s()) + " s waiting for " + why);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}

void Rendezvous::finish() {
    publish("done", "1", 1, (int)rank_);
    if (rank_ != 0) return;
    for (uint32_t r = 0; r < world_; ++r) (void)await("done", 1, (int)r);
    if (getenv("BARBELL_AMD_KEEP_RENDEZVOUS")) return;   // test aid: what an interrupted run leaves behind
    for (uint32_t r = 0; r < world_; ++r)
        for (const char* w : {"hello", "info", "counts", "done"}) (void)unlink(part((int)r, w).c_str());
    (void)unlink(part(-1, "id").c_str());
}

void shard_rendezvous_reset(const std::string& base, uint32_t rank) { Rendezvous::hello(base, rank); }

// The histogram over the W processes through the files alone (what processes that share a device do; `barbell-amd rendezvous`, no GPU)
std::vector<uint64_t> rendezvous_sum_counts(const std::string& base, uint32_t rank, uint32_t world, const std::string& bus, const std::vector<uint64_t>& local,
                                            bool* shared_device) {
    Rendezvous rv(base, rank, world);
    ShardInfo me;
    memset(&me, 0, sizeof(me));
    snprintf(me.bus, sizeof(me.bus), "%s", bus.c_str());
    me.n_counts = local.size(); me.world = world; me.rank = rank;
    const std::vector<ShardInfo> all = rv.meet(me);
    bool shared = false;
    for (uint32_t r = 0; r < world; ++r)
        for (uint32_t q = 0; q < r; ++q) shared |= !strcmp(all[q].bus, all[r].bus);
    if (shared_device) *shared_device = shared;
    std::vector<uint64_t> total(local.size(), 0);
    rv.publish("counts", local.data(), local.size() * sizeof(uint64_t), (int)rank);
    for (uint32_t r = 0; r < world; ++r) {
        const std::vector<char> b = rv.await("counts", local.size() * sizeof(uint64_t), (int)r);
        const uint64_t* c = (const uint64_t*)b.data();
        for (size_t i = 0; i < local.size(); ++i) total[i] += c[i];
    }
    rv.finish();
    return total;
}

}  // namespace barbell
