// bb_rccl.cpp — the one collective of the path: the per-(group, barcode) histogram summed over the
// contexts of a run.  Contexts on DISTINCT devices are all-reduced in place with RCCL over xGMI
// (ncclAllReduce, ncclUint64, ncclSum on bb_counts_dev, one communicator per device, single process); contexts that
// share a device are summed on the host first (RCCL cannot put two ranks on one GPU).  No torch, no MPI.
// One process per GPU (`--shard R/W --rccl-id PATH`, the mode DESIGN §6 recommends past two GPUs): the W processes meet through small files
// next to PATH, bootstrap one communicator with ncclCommInitRank (rank 0 publishes the ncclUniqueId) and all-reduce the same buffer; processes
// that share a device (a one-GPU box) are summed through the same files instead — every rank ends with the total (annotator.rs:278-280 is the
// fan-out this replaces: paraseq's worker threads share ONE process, so the reference never needs this step).
// librccl.so is bound at the first all-reduce (dlopen), not at program start: it is a 570 MB library, and a run on one device — most
// runs — never calls it (mapping and relocating it was a fifth of a second of every start-up).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "bb_host.hpp"
#include "bb_rendezvous.hpp"

namespace barbell {

namespace {
struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
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
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        Dl_info di;
        fprintf(stderr, "librccl.so bound: %s\n", dladdr((void*)r.AllReduce, &di) && di.dli_fname ? di.dli_fname : "?");
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


// ---- one process per GPU: the histogram over the W processes of a `--shard R/W` run --------------------------------------------------
// The processes meet through bb_rendezvous.cpp (identity-stamped files: a stale file of an interrupted run is never read as this run's).
std::vector<uint64_t> allreduce_counts_shards(Demuxer* lead, const std::vector<uint64_t>& local, uint32_t rank, uint32_t world,
                                              const std::string& base, std::string& how) {
    const size_t n = local.size();
    Rendezvous rv(base, rank, world);
    ShardInfo me;
    memset(&me, 0, sizeof(me));
    HCHK(hipDeviceGetPCIBusId(me.bus, (int)sizeof(me.bus), lead->device()));
    char host[40] = "";
    (void)gethostname(host, sizeof(host) - 1);
    const size_t bl = strlen(me.bus);
    snprintf(me.bus + bl, sizeof(me.bus) - bl, "@%s", host);   // RCCL's duplicate-GPU rule is per (host, device)
    me.n_counts = n; me.world = world; me.rank = rank;
    const std::vector<ShardInfo> all = rv.meet(me);
    bool shared_device = false;
    for (uint32_t r = 0; r < world; ++r)
        for (uint32_t q = 0; q < r; ++q) shared_device |= !strcmp(all[q].bus, all[r].bus);
    std::vector<uint64_t> total(n, 0);
    if (shared_device) {
        // two ranks of one communicator cannot sit on one GPU: sum through the files (every rank reads every rank's counts)
        rv.publish("counts", local.data(), n * sizeof(uint64_t), (int)rank);
        for (uint32_t r = 0; r < world; ++r) {
            const std::vector<char> b = rv.await("counts", n * sizeof(uint64_t), (int)r);
            const uint64_t* c = (const uint64_t*)b.data();
            for (size_t i = 0; i < n; ++i) total[i] += c[i];
        }
        how = "files (" + std::to_string(world) + " processes share a device)";
    } else {
        ncclUniqueId id;
        if (rank == 0) {
            RCHK(rccl().GetUniqueId(&id));
            rv.publish("id", &id, sizeof(id), -1);
        } else {
            const std::vector<char> b = rv.await("id", sizeof(id), -1);
            memcpy(&id, b.data(), sizeof(id));
        }
        HCHK(hipSetDevice(lead->device()));
        uint64_t* p = bb_counts_dev(lead->ctx());
        const int rc = bb_dev_upload(lead->ctx(), p, local.data(), n * sizeof(uint64_t));   // this process's sum over its contexts
        if (rc != BB_OK) throw BarbellError(rc, "bb_dev_upload (histogram)");
        ncclComm_t comm;
        RCHK(rccl().CommInitRank(&comm, (int)world, id, (int)rank));
        RCHK(rccl().AllReduce(p, p, n, ncclUint64, ncclSum, comm, (hipStream_t)0));
        HCHK(hipDeviceSynchronize());
        (void)rccl().CommDestroy(comm);
        total = lead->counts();
        how = "rccl (" + std::to_string(world) + " processes, ncclCommInitRank)";
    }
    rv.finish();   // rank 0 removes the run's files once every rank has read what it needs
    return total;
}

}  // namespace barbell
