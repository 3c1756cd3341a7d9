// shard.cpp -- see shard.h.
#include "tunables.h"
#include "shard.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <set>
#include <string>

#include <rccl/rccl.h>       // types and prototypes only: the library is bound with dlopen below

#include "estimator.h"

namespace vb2 {

#define VB2_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#call) + " failed: " + hipGetErrorString(e_));       \
            return VB2_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

namespace {

// librccl, bound on first use.  (When the process already holds a librccl.so.1 -- PyTorch ships
// its own -- dlopen returns that one, so there is never a second RCCL in the process.)
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    bool stub = false;          // the test-only in-process stand-in (tests/stub_rccl) was bound, not librccl
    std::string why;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        // VB2_RCCL_LIB=<path>: bind THAT library instead (tests/stub_rccl: an in-process stand-in whose ranks may
        // share a device, so that a one-GPU box can execute the N > 1 control flow; never set in production)
        const char* forced = std::getenv("VB2_RCCL_LIB");
        if (forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                r.why = std::string("VB2_RCCL_LIB not loadable: ") + (dlerror() ? dlerror() : "?");
                return;
            }
            r.stub = dlsym(h, "vb2_rccl_stub_marker") != nullptr;
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        }
        if (!h) {
            r.why = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?");
            return;
        }
#define VB2_SYM(field, sym)                                                    \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));              \
    if (!r.field) { r.why = std::string("librccl lacks ") + sym; return; }
        VB2_SYM(GetUniqueId, "ncclGetUniqueId")
        VB2_SYM(CommInitRank, "ncclCommInitRank")
        VB2_SYM(CommInitAll, "ncclCommInitAll")
        VB2_SYM(CommDestroy, "ncclCommDestroy")
        VB2_SYM(AllReduce, "ncclAllReduce")
        VB2_SYM(GroupStart, "ncclGroupStart")
        VB2_SYM(GroupEnd, "ncclGroupEnd")
        VB2_SYM(GetErrorString, "ncclGetErrorString")
#undef VB2_SYM
        r.ok = true;
    });
    return r;
}

#define VB2_NCCL(call)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            set_error(std::string(#call) + " failed: " + rccl().GetErrorString(r_));           \
            return VB2_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

int group_eval_cb(void* user, int32_t n, const double* pc1, const double* pc2, const double* alpha, double* out)
{
    return static_cast<ShardGroup*>(user)->eval(n, pc1, pc2, alpha, out);
}

}  // namespace

int rccl_unique_id(void* id128)
{
    Rccl& r = rccl();
    if (!r.ok) {
        set_error(r.why);
        return VB2_ERR_NO_DEVICE;
    }
    ncclUniqueId id;
    VB2_NCCL(r.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return VB2_OK;
}

bool rccl_is_stub()
{
    Rccl& r = rccl();
    return r.ok && r.stub;
}

void shard_range(const vb2_input* in, int r, int n, int* lo, int* hi)
{
    const int M = in->num_marker;
    if (n <= 1) { *lo = 0; *hi = M; return; }
    const int64_t base = in->read_off[0], total = in->read_off[M] - base;
    auto cut = [&](int q) -> int {
        if (q <= 0) return 0;
        if (q >= n) return M;
        // first marker whose reads start at or beyond q/n of all reads
        const double target = (double)base + (double)total * q / n;
        const int64_t* p = std::lower_bound(in->read_off, in->read_off + M + 1, target,
                                            [](int64_t off, double t) { return (double)off < t; });
        return (int)std::min<int64_t>(p - in->read_off, M);
    };
    *lo = cut(r);
    *hi = std::max(*lo, cut(r + 1));
}

vb2_input shard_view(const vb2_input* in, int lo, int hi)
{
    vb2_input v = *in;
    v.num_marker = hi - lo;
    if (in->ud) v.ud = in->ud + (size_t)lo * in->num_pc;
    if (in->means) v.means = in->means + lo;
    v.read_off = in->read_off + lo;           // absolute offsets: bases/quals stay the whole arrays
    v.alt_base = in->alt_base + lo;
    if (in->known_af) v.known_af = in->known_af + lo;
    return v;
}

ShardGroup::~ShardGroup()
{
    if (resident_) end_resident();
    for (size_t i = 0; i < ctx.size(); ++i) {
        if (ctx[i]) (void)hipSetDevice(ctx[i]->device);
        if (i < comm_.size() && comm_[i]) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_[i]));
        if (i < d_part_.size() && d_part_[i]) (void)hipFree(d_part_[i]);
        delete ctx[i];
    }
}

int ShardGroup::create(const vb2_input* in, const int32_t* devices, int num_device, ShardGroup** out)
{
    *out = nullptr;
    if (!in || !devices || num_device < 1 || num_device > 64) {
        set_error("vb2_shard_group_create: invalid argument");
        return VB2_ERR_INVALID;
    }
    std::unique_ptr<ShardGroup> g(new ShardGroup());
    g->num_pc = in->num_pc;
    g->num_marker = in->num_marker;
    const std::set<int> distinct(devices, devices + num_device);
    // a real all-reduce needs one rank per DEVICE; shards that share a device (tests on one GPU)
    // are summed on the host instead -- unless the test stand-in for librccl is bound (VB2_RCCL_LIB), whose
    // ranks may share a device: then the launch -> grouped all-reduce -> publish path runs with N > 1 there too
    const bool host_forced = tunables().shard_reduce_host != 0;
    // (the stand-in is only ever bound through VB2_RCCL_LIB -- without it the real librccl is not loaded just to
    // learn that it is not the stand-in)
    const bool shared_ok = (int)distinct.size() < num_device && std::getenv("VB2_RCCL_LIB") != nullptr && rccl().ok && rccl().stub;
    g->use_rccl = ((int)distinct.size() == num_device || shared_ok) && !host_forced;
    for (int d = 0; d < num_device; ++d) {
        int lo, hi;
        shard_range(in, d, num_device, &lo, &hi);
        const vb2_input view = shard_view(in, lo, hi);
        vb2_options opt{};
        opt.device = devices[d];
        Context* c = nullptr;
        const int rc = Context::create(&view, &opt, &c);
        if (rc) return rc;
        g->ctx.push_back(c);
        g->lo.push_back(lo);
        g->hi.push_back(hi);
    }
    if (g->use_rccl) {
        Rccl& r = rccl();
        if (!r.ok) {
            set_error(r.why);
            return VB2_ERR_NO_DEVICE;
        }
        std::vector<ncclComm_t> comms(num_device);
        std::vector<int> devs(devices, devices + num_device);
        VB2_NCCL(r.CommInitAll(comms.data(), num_device, devs.data()));
        for (int d = 0; d < num_device; ++d) g->comm_.push_back(comms[d]);
        for (int d = 0; d < num_device; ++d) {
            VB2_HIP(hipSetDevice(devices[d]));
            double* p = nullptr;
            VB2_HIP(hipMalloc((void**)&p, sizeof(double) * kStagePoints));
            g->d_part_.push_back(p);
        }
    }
    *out = g.release();
    return VB2_OK;
}

int ShardGroup::create_rank(const vb2_input* in, int device, int rank, int nranks, const void* id128,
                            ShardGroup** out)
{
    *out = nullptr;
    if (!in || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("vb2_shard_group_create_rank: invalid argument");
        return VB2_ERR_INVALID;
    }
    if (id128 == nullptr && nranks > 1) {
        set_error("vb2_shard_group_create_rank: nranks > 1 needs the communicator id (vb2_rccl_unique_id), or "
                  "VB2_SHARD_PARTIAL_SUMS for a group that returns this rank's partial sums");
        return VB2_ERR_INVALID;
    }
    std::unique_ptr<ShardGroup> g(new ShardGroup());
    g->num_pc = in->num_pc;
    g->num_marker = in->num_marker;
    g->rank = rank;
    g->nranks = nranks;
    int lo, hi;
    shard_range(in, rank, nranks, &lo, &hi);
    const vb2_input view = shard_view(in, lo, hi);
    vb2_options opt{};
    opt.device = device;
    Context* c = nullptr;
    int rc = Context::create(&view, &opt, &c);
    if (rc) return rc;
    g->ctx.push_back(c);
    g->lo.push_back(lo);
    g->hi.push_back(hi);
    // id128: the communicator's id; VB2_SHARD_PARTIAL_SUMS: none, by request (the caller reduces this rank's partial
    // sums over its own transport); NULL: none needed with one rank -- with more it is refused (a forgotten
    // id used to yield plausible but partial LLKs without a word)
    const bool want_partial = id128 == VB2_SHARD_PARTIAL_SUMS;
    g->use_rccl = id128 != nullptr && !want_partial;       // (nranks == 1 with an id: a one-rank communicator, for tests)
    g->partial_sums = nranks > 1 && !g->use_rccl;
    if (g->use_rccl) {
        Rccl& r = rccl();
        if (!r.ok) {
            set_error(r.why);
            return VB2_ERR_NO_DEVICE;
        }
        VB2_HIP(hipSetDevice(c->device));
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ncclComm_t comm = nullptr;
        VB2_NCCL(r.CommInitRank(&comm, nranks, id, rank));
        g->comm_.push_back(comm);
        double* p = nullptr;
        VB2_HIP(hipMalloc((void**)&p, sizeof(double) * kStagePoints));
        g->d_part_.push_back(p);
    }
    *out = g.release();
    return VB2_OK;
}

// ---- evaluation by plain launches: one launch per owned shard, then the all-reduce ----
int ShardGroup::eval_launch(int num_point, const double* pc1, const double* pc2, const double* alpha, double* out)
{
    const int k = num_pc, stride = 2 * k + 1;
    const size_t S = ctx.size();
    for (int done = 0; done < num_point; done += kStagePoints) {
        const int n = std::min(kStagePoints, num_point - done);
        for (int attempt = 0; attempt < 2; ++attempt) {        // (a NaN partial: see Context::eval_host)
            std::vector<unsigned long long> seq(S, 0);
            for (size_t s = 0; s < S; ++s) {
                Context* c = ctx[s];
                VB2_HIP(hipSetDevice(c->device));
                for (int b = 0; b < n; ++b) {
                    double* row = c->h_points + (size_t)b * stride;
                    std::memcpy(row, pc1 + (size_t)(done + b) * k, sizeof(double) * k);
                    std::memcpy(row + k, pc2 + (size_t)(done + b) * k, sizeof(double) * k);
                    row[2 * k] = alpha[done + b];
                }
                // (the result words are NaN until this step's stores land -- a stale word of the previous step would
                // be a finite number and be summed silently)
                for (int b = 0; b < n; ++b) c->h_out[b] = std::numeric_limits<double>::quiet_NaN();
                int rc;
                if (use_rccl) {
                    rc = c->eval_device(n, c->d_points, d_part_[s], c->stream, nullptr, 0, c->h_points, attempt);
                } else {
                    seq[s] = ++c->done_seq_;
                    rc = c->eval_device(n, c->d_points, c->d_out, c->stream, c->L.num_mt > 0 ? c->d_done : nullptr,
                                        seq[s], c->h_points, attempt);
                }
                if (rc) return rc;
            }
            bool any_nan = false;
            if (use_rccl) {
                // ONE all-reduce of n doubles: every rank ends up with the same sums
                Rccl& r = rccl();
                if (S > 1) VB2_NCCL(r.GroupStart());
                for (size_t s = 0; s < S; ++s) {
                    VB2_HIP(hipSetDevice(ctx[s]->device));
                    VB2_NCCL(r.AllReduce(d_part_[s], d_part_[s], (size_t)n, ncclDouble, ncclSum,
                                         static_cast<ncclComm_t>(comm_[s]), ctx[s]->stream));
                }
                if (S > 1) VB2_NCCL(r.GroupEnd());
                ++num_allreduce;
                // Behind the collective, on the same stream: a one-wave kernel moves the reduced sums to
                // mapped host memory and then writes the sequence number this thread spins on -- no
                // device-to-host copy call, no hipStreamSynchronize per shard (round 2: +37 us per step
                // with ONE rank; now +19 us, tools/shard_step_time.py.  Measured and dropped: the mapped buffer as
                // the collective's receive buffer + hipStreamWriteValue64 for the flag, 93.6 vs 91.8 us per
                // 48-point step).  Every owned shard publishes (its stream is then known to be drained);
                // shard 0's copy is the result.
                for (size_t s = 0; s < S; ++s) {
                    Context* c = ctx[s];
                    VB2_HIP(hipSetDevice(c->device));
                    seq[s] = ++c->done_seq_;
                    VB2_HIP(launch_publish(d_part_[s], c->d_out, n, c->d_done, seq[s], c->stream));
                }
                for (size_t s = 0; s < S; ++s) {
                    Context* c = ctx[s];
                    bool seen = false;
                    const auto t0 = std::chrono::steady_clock::now();
                    for (unsigned spins = 0;; ++spins) {
                        if (__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) == seq[s]) { seen = true; break; }
                        if ((spins & 0x3ff) == 0x3ff &&
                            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(8)) break;
                        __builtin_ia32_pause();
                    }
                    if (!seen) {
                        VB2_HIP(hipSetDevice(c->device));
                        VB2_HIP(hipStreamSynchronize(c->stream));
                    }
                }
                (void)settle_results(ctx[0]->h_out, n);        // (a store still on its way behind the flag)
                for (int b = 0; b < n; ++b) {
                    out[done + b] = ctx[0]->h_out[b];
                    any_nan |= std::isnan(out[done + b]);
                }
            } else {
                for (int b = 0; b < n; ++b) out[done + b] = 0.0;
                for (size_t s = 0; s < S; ++s) {               // host sum, in shard order
                    Context* c = ctx[s];
                    VB2_HIP(hipSetDevice(c->device));
                    bool seen = false;
                    if (c->L.num_mt > 0) {
                        const auto t0 = std::chrono::steady_clock::now();
                        for (unsigned spins = 0;; ++spins) {
                            if (__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) == seq[s]) { seen = true; break; }
                            if ((spins & 0x3ff) == 0x3ff &&
                                std::chrono::steady_clock::now() - t0 > std::chrono::seconds(4)) break;
                            __builtin_ia32_pause();
                        }
                    }
                    if (!seen) VB2_HIP(hipStreamSynchronize(c->stream));
                    if (c->L.num_mt > 0) (void)settle_results(c->h_out, n);
                    for (int b = 0; b < n; ++b) {
                        const double v = c->L.num_mt > 0 ? c->h_out[b] : 0.0;
                        any_nan |= std::isnan(v);
                        out[done + b] += v;
                    }
                }
            }
            if (!any_nan) break;
        }
    }
    return VB2_OK;
}

// ---- evaluation against resident kernels (single-process search): same command to every
//      shard's mailbox, partial sums added on the host in shard order ----
int ShardGroup::begin_resident()
{
    if (nranks > 1) return 0;                      // process-per-GPU: the collective needs stream order
    for (size_t s = 0; s < ctx.size(); ++s)
        if (!ctx[s]->resident_begin()) {
            for (size_t t = 0; t < s; ++t) ctx[t]->resident_end();
            return 0;
        }
    resident_ = true;
    return 1;
}

void ShardGroup::end_resident()
{
    for (Context* c : ctx) c->resident_end();
    resident_ = false;
}

int ShardGroup::eval_resident(int num_point, const double* pc1, const double* pc2, const double* alpha, double* out)
{
    const int k = num_pc, stride = 2 * k + 1;
    for (int served = 0; served < num_point; served += 4) {
        const int n = std::min(4, num_point - served);
        double rows[4 * (2 * VB2_MAX_PC + 1)];
        for (int b = 0; b < n; ++b) {
            double* row = rows + (size_t)b * stride;
            std::memcpy(row, pc1 + (size_t)(served + b) * k, sizeof(double) * k);
            std::memcpy(row + k, pc2 + (size_t)(served + b) * k, sizeof(double) * k);
            row[2 * k] = alpha[served + b];
        }
        for (Context* c : ctx) c->resident_submit(n, rows);
        double part[4];
        bool ok = true;
        for (int b = 0; b < n; ++b) out[served + b] = 0.0;
        for (Context* c : ctx) {
            if (!c->resident_collect(n, part)) { ok = false; continue; }     // (keep collecting: every kernel answers or leaves)
            for (int b = 0; b < n; ++b) out[served + b] += part[b];
        }
        if (!ok) {          // a kernel gave up: leave the mode everywhere, redo these rows by launches
            end_resident();
            return eval_launch(num_point - served, pc1 + (size_t)served * k, pc2 + (size_t)served * k,
                               alpha + served, out + served);
        }
    }
    return VB2_OK;
}

int ShardGroup::eval(int num_point, const double* pc1, const double* pc2, const double* alpha, double* llk_out)
{
    if (num_point < 0 || (num_point > 0 && (!pc1 || !pc2 || !alpha || !llk_out))) {
        set_error("vb2_shard_group_eval: invalid argument");
        return VB2_ERR_INVALID;
    }
    if (num_point == 0) return VB2_OK;
    const int rc = resident_ ? eval_resident(num_point, pc1, pc2, alpha, llk_out) : eval_launch(num_point, pc1, pc2, alpha, llk_out);
    if (rc != VB2_OK) return rc;
    // NaN parameters: the reference's rule (context.h: params_hold_nan)
    // (a group without markers answers 0 anyway: no condition on the shards' sizes, which differ between ranks)
    bool known_af = false;
    for (const Context* c : ctx) known_af |= c->L.known_af != nullptr;
    for (int b = 0; b < num_point; ++b)
        if (params_hold_nan(pc1 + (size_t)b * num_pc, pc2 + (size_t)b * num_pc, alpha[b], num_pc, known_af)) llk_out[b] = 0.0;
    return VB2_OK;
}

int ShardGroup::optimize(const vb2_model* model, vb2_estimate* out, vb2_trace* trace)
{
    if (!model || !out) {
        set_error("vb2_shard_group_optimize_llk: invalid argument");
        return VB2_ERR_INVALID;
    }
    if (nranks > 1 && !use_rccl) {
        set_error("vb2_shard_group_optimize_llk: this rank's group has no communicator (created with id128 == NULL): "
                  "it yields partial sums only");
        return VB2_ERR_INVALID;
    }
    // ONE shard in one process is the plain single-context search: each Minimize() on the device itself
    // (resident_kernel.inc), like vb2_ctx_optimize_llk -- round 3 drove the resident kernel from the host optimiser
    // here, one mailbox round trip per iteration: 8.4 ms against 6.5 for the same 780 evaluations.
    const bool single = ctx.size() == 1 && nranks == 1;
    if (single && trace && trace->capacity > 0 && ctx[0]->device_simplex_enabled) (void)ctx[0]->reserve_trace(trace->capacity);
    const bool res = begin_resident() != 0;
    Estimator est(num_pc, group_eval_cb, this);
    apply_model(est, *model, ctx[0]->L.known_af != nullptr);
    est.trace = trace;
    if (trace) trace->count = 0;
    if (single && res && ctx[0]->device_simplex_dim() > 0 && !(trace && ctx[0]->trace_stage_rows < trace->capacity))
        est.dev_ctx = ctx[0];
    const int rc = est.OptimizeLLK();
    if (res && resident_) end_resident();
    if (rc) return rc;
    fill_estimate(est, out);
    return VB2_OK;
}

}  // namespace vb2
