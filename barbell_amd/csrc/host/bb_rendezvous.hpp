// bb_rendezvous.hpp — how the W processes of a `--shard R/W --rccl-id PATH` run find each other: small files next to PATH, no HIP, no RCCL
// (bb_rccl.cpp does the collective; `barbell-amd rendezvous` drives this file alone, without a GPU: tests/test_rendezvous.py).
//
// Run identity (round 6): file names alone said nothing about WHICH run had written a file — after an interrupted
// run, a rank that reached the meeting point before a slower peer had started read that peer's old `.rN.info` / `.counts` / rank 0's old
// ncclUniqueId: stale histograms summed silently, or a hang in ncclCommInitRank on a dead id.  Now every process publishes a fresh random
// nonce when it STARTS (`PATH.rK.hello`, after removing what an earlier run of rank K left), the run's identity is the vector of all W
// nonces, and every later file (info, counts, id, done) begins with the identity its writer saw.  A reader trusts a file only if that
// identity is the current one — which contains the reader's OWN fresh nonce, something no file of an earlier run can hold.  A file of
// another identity is not an error by itself (its writer may simply not have started yet): the reader keeps waiting, re-reading the
// hellos and re-publishing its own files when the identity moves, and the timeout's message names the stale file.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace barbell {

struct ShardInfo { char bus[64]; uint64_t n_counts; uint32_t world, rank; };

class Rendezvous {
public:
    // program start of rank `rank`: removes the rank's files of an earlier run and publishes its hello.  Throws BarbellError.
    static void hello(const std::string& base, uint32_t rank);
    Rendezvous(std::string base, uint32_t rank, uint32_t world);
    // every rank's info of THIS run (publishes `me`; waits until all W infos carry the current identity and agree on n_counts / world)
    std::vector<ShardInfo> meet(const ShardInfo& me);
    // identity-stamped blobs after meet(): `what` is "counts", "id", "done" ...; rank < 0 = a file of the run, not of a rank (rank 0's "id")
    void publish(const char* what, const void* data, size_t bytes, int of_rank);
    std::vector<char> await(const char* what, size_t bytes, int of_rank);
    // the end of the meeting: rank 0 removes every file of the run once all ranks are done with them
    void finish();
    static double timeout_s();   // BARBELL_AMD_RCCL_TIMEOUT, seconds; default 600 (an hour and more is the option, not the default)

private:
    std::string base_;
    uint32_t rank_, world_;
    std::vector<char> identity_;   // W nonces of kNonce bytes, set by meet()
    std::string part(int rank, const char* what) const;
    bool read_identity(std::vector<char>& out, std::string& missing) const;
};

}  // namespace barbell
