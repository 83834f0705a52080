// bb_rccl.cpp — the one collective of the path: the per-(group, barcode) histogram summed over the
// contexts of a run.  Contexts on DISTINCT devices are all-reduced in place with RCCL over xGMI
// (ncclAllReduce, ncclUint64, ncclSum on bb_counts_dev, one communicator per device, single process); contexts that
// share a device are summed on the host first (RCCL cannot put two ranks on one GPU).  No torch, no MPI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "bb_host.hpp"

namespace barbell {

#define RCHK(call)                                                                                             \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) throw BarbellError(BB_E_HIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define HCHK(call)                                                                                            \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) throw BarbellError(BB_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

std::vector<uint64_t> allreduce_counts(const std::vector<Demuxer*>& dms, std::string& how) {
    const size_t n = bb_counts_len(dms[0]->ctx());
    // contexts grouped by device; the first context of a device carries the device's sum
    std::map<int, std::vector<Demuxer*>> by_dev;
    for (Demuxer* d : dms) by_dev[d->device()].push_back(d);
    std::vector<Demuxer*> leaders;
    for (auto& kv : by_dev) {
        std::vector<uint64_t> sum(n, 0);
        for (Demuxer* d : kv.second) { const auto c = d->counts(); for (size_t i = 0; i < n; ++i) sum[i] += c[i]; }
        Demuxer* lead = kv.second[0];
        if (kv.second.size() > 1) {
            const int rc = bb_dev_upload(lead->ctx(), bb_counts_dev(lead->ctx()), sum.data(), n * sizeof(uint64_t));
            if (rc != BB_OK) throw BarbellError(rc, "bb_dev_upload (histogram)");
        }
        leaders.push_back(lead);
    }
    const bool force = getenv("BARBELL_AMD_FORCE_RCCL") != nullptr;  // exercise the RCCL call path with a single rank
    if (leaders.size() == 1 && !force) {
        how = dms.size() > 1 ? "host" : "single";
        return leaders[0]->counts();
    }
    const int R = (int)leaders.size();
    std::vector<int> devs;
    for (Demuxer* d : leaders) devs.push_back(d->device());
    std::vector<ncclComm_t> comms((size_t)R);
    RCHK(ncclCommInitAll(comms.data(), R, devs.data()));
    RCHK(ncclGroupStart());
    for (int i = 0; i < R; ++i) {
        HCHK(hipSetDevice(devs[(size_t)i]));
        uint64_t* p = bb_counts_dev(leaders[(size_t)i]->ctx());
        RCHK(ncclAllReduce(p, p, n, ncclUint64, ncclSum, comms[(size_t)i], (hipStream_t)0));
    }
    RCHK(ncclGroupEnd());
    for (int i = 0; i < R; ++i) { HCHK(hipSetDevice(devs[(size_t)i])); HCHK(hipDeviceSynchronize()); }
    for (auto& c : comms) (void)ncclCommDestroy(c);
    how = "rccl";
    return leaders[0]->counts();
}

}  // namespace barbell
