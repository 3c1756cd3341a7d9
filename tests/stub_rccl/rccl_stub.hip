// rccl_stub.hip -- TEST INFRASTRUCTURE, never shipped, never linked into libvb2.
//
// An in-process stand-in for the eight librccl entry points shard.cpp binds with dlopen
// (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy, ncclAllReduce, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString), selected with VB2_RCCL_LIB=<this .so>.  Its "ranks" live in ONE
// process and may share ONE device, so a box with a single MI355X can execute the N > 1 control flow
// of vb2_shard_group_* -- ncclCommInitAll over several shards, ncclCommInitRank with nranks > 1 from
// several threads, the grouped all-reduce loop over more than one communicator, the publish kernel
// behind it -- which real RCCL cannot do there (it refuses two ranks on one device).
//
// Semantics kept from RCCL: the all-reduce is STREAM-ORDERED (a kernel on the caller's stream), every
// rank's call returns at once, every rank receives the same sums.  The sum is taken in rank order
// (0.0 + p0 + p1 + ...), i.e. bit for bit what ShardGroup's host-sum path computes, so the tests can
// demand equality with that path and 1e-12 against the oracle fixtures.
//
// Mechanics: a communicator world owns a block of mapped host memory (visible to every device of the
// process): two generations of [nranks][kMaxCount] slots and one arrival word per rank.  The kernel of
// rank r, generation g: writes its contribution to slot[g & 1][r], fences, stores g to arrive[r], waits
// until every arrive[] >= g, adds the slots in rank order.  A rank can only start generation g + 1 after
// finishing g, and finishes g only after every rank ARRIVED at g -- i.e. finished reading g - 1 -- so two
// slot generations are enough.  The wait is bounded (5 s of the 100 MHz wall clock): a missing rank
// yields NaN, never a hung GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {

constexpr int kMaxCount = 4096;              // doubles per all-reduce (ShardGroup reduces <= kStagePoints)
constexpr unsigned long long kGiveUpTicks = 500000000ull;     // 5 s at 100 MHz

struct World {
    int nranks = 0;
    int refs = 0;
    double* slots = nullptr;                 // mapped host memory [2][nranks][kMaxCount]
    unsigned long long* arrive = nullptr;    // mapped host memory [nranks]
    std::string key;
};

struct StubComm {
    World* w;
    int rank;
    unsigned long long gen;                  // all-reduces issued by this rank
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;      // by unique id (ncclCommInitRank's rendezvous)
unsigned long long g_next_id = 1;

World* new_world(int nranks, const std::string& key)
{
    World* w = new World();
    w->nranks = nranks;
    w->key = key;
    if (hipHostMalloc((void**)&w->slots, sizeof(double) * 2 * (size_t)nranks * kMaxCount,
                      hipHostMallocPortable | hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&w->arrive, sizeof(unsigned long long) * (size_t)nranks,
                      hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        if (w->slots) (void)hipHostFree(w->slots);
        delete w;
        return nullptr;
    }
    std::memset(w->arrive, 0, sizeof(unsigned long long) * (size_t)nranks);
    return w;
}

__global__ void __launch_bounds__(256)
stub_allreduce_sum_f64(const double* send, double* recv, int count, double* slots, unsigned long long* arrive,
                       int rank, int nranks, unsigned long long gen)
{
    const int tid = threadIdx.x;
    double* mine = slots + (((size_t)(gen & 1) * nranks + rank) * kMaxCount);
    for (int i = tid; i < count; i += blockDim.x)
        __hip_atomic_store(&mine[i], send[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    __shared__ int ok;
    if (tid == 0) {
        __hip_atomic_store(&arrive[rank], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        int all = 0;
        for (unsigned it = 0; !all; ++it) {
            all = 1;
            for (int r = 0; r < nranks; ++r)
                if (__hip_atomic_load(&arrive[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gen) all = 0;
            if (!all) {
                if ((it & 63) == 63 && wall_clock64() - t0 > kGiveUpTicks) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        ok = all;
    }
    __syncthreads();
    for (int i = tid; i < count; i += blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < nranks; ++r)
            s += __hip_atomic_load(&slots[((size_t)(gen & 1) * nranks + r) * kMaxCount + i], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        recv[i] = ok ? s : __builtin_nan("");
    }
}

}  // namespace

extern "C" {

// marks this library in a process (tests assert that the stub, not the real thing, was bound)
int vb2_rccl_stub_marker(void) { return 1; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof(*id));
    std::memcpy(id->internal, "VB2STUB", 8);
    const unsigned long long n = g_next_id++;
    std::memcpy(id->internal + 8, &n, sizeof(n));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::memcmp(id.internal, "VB2STUB", 8) != 0)
        return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    const std::string key(id.internal, sizeof(id.internal));
    World* w = nullptr;
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) {
        w = new_world(nranks, key);
        if (!w) return ncclSystemError;
        g_worlds[key] = w;
    } else {
        w = it->second;
        if (w->nranks != nranks) return ncclInvalidArgument;
    }
    ++w->refs;
    *comm = reinterpret_cast<ncclComm_t>(new StubComm{w, rank, 0ull});
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    (void)devlist;                       // the stub's ranks may share a device: that is its point
    if (!comms || ndev < 1) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    World* w = new_world(ndev, "");
    if (!w) return ncclSystemError;
    w->refs = ndev;
    for (int r = 0; r < ndev; ++r) comms[r] = reinterpret_cast<ncclComm_t>(new StubComm{w, r, 0ull});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    std::lock_guard<std::mutex> lk(g_mu);
    if (--c->w->refs == 0) {
        if (!c->w->key.empty()) g_worlds.erase(c->w->key);
        (void)hipHostFree(c->w->slots);
        (void)hipHostFree(c->w->arrive);
        delete c->w;
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff || datatype != ncclDouble || op != ncclSum || count > (size_t)kMaxCount)
        return ncclInvalidArgument;
    StubComm* c = reinterpret_cast<StubComm*>(comm);
    const unsigned long long gen = ++c->gen;
    hipLaunchKernelGGL(stub_allreduce_sum_f64, dim3(1), dim3(256), 0, stream, static_cast<const double*>(sendbuff),
                       static_cast<double*>(recvbuff), (int)count, c->w->slots, c->w->arrive, c->rank, c->w->nranks, gen);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (rccl stub)";
    case ncclSystemError: return "system error (rccl stub)";
    default: return "error (rccl stub)";
    }
}

}  // extern "C"
