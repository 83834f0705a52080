// bb_rccl.cpp — the one collective of the path: the per-(group, barcode) histogram summed over the
// contexts of a run.  Contexts on DISTINCT devices are all-reduced in place with RCCL over xGMI
// (ncclAllReduce, ncclUint64, ncclSum on bb_counts_dev, one communicator per device, single process); contexts that
// share a device are summed on the host first (RCCL cannot put two ranks on one GPU).  No torch, no MPI.
// librccl.so is bound at the first all-reduce (dlopen), not at program start: it is a 570 MB library, and a run on one device — most
// runs — never calls it (mapping and relocating it was a fifth of a second of every start-up).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "bb_host.hpp"

namespace barbell {

namespace {
struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
const Rccl& rccl() {
    static const Rccl R = []() {
        Rccl r;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) throw BarbellError(BB_E_HIP, std::string("librccl.so not found (contexts on several devices need RCCL): ") + dlerror());
        auto sym = [&](const char* s) { void* p = dlsym(h, s); if (!p) throw BarbellError(BB_E_HIP, std::string("librccl.so lacks ") + s); return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        return r;
    }();
    return R;
}
}  // namespace

#define RCHK(call)                                                                                             \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) throw BarbellError(BB_E_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
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
    RCHK(rccl().CommInitAll(comms.data(), R, devs.data()));
    RCHK(rccl().GroupStart());
    for (int i = 0; i < R; ++i) {
        HCHK(hipSetDevice(devs[(size_t)i]));
        uint64_t* p = bb_counts_dev(leaders[(size_t)i]->ctx());
        RCHK(rccl().AllReduce(p, p, n, ncclUint64, ncclSum, comms[(size_t)i], (hipStream_t)0));
    }
    RCHK(rccl().GroupEnd());
    for (int i = 0; i < R; ++i) { HCHK(hipSetDevice(devs[(size_t)i])); HCHK(hipDeviceSynchronize()); }
    for (auto& c : comms) (void)rccl().CommDestroy(c);
    how = "rccl";
    return leaders[0]->counts();
}

}  // namespace barbell
