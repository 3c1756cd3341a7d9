// batch.cpp -- see batch.h.
#include "batch.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>

#include "estimator.h"
#include "lockstep.h"

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
constexpr int kSlot = 8;                     // point slots per sample and step (VB2_BATCH_SLOTS)
constexpr int kShapes = Batch::kShapes;
}

Batch::~Batch()
{
    if (device >= 0) (void)hipSetDevice(device);
    if (d_layouts_) (void)hipFree(d_layouts_);
    if (d_sched_) (void)hipFree(d_sched_);
    if (d_partials_) (void)hipFree(d_partials_);
    if (d_tickets_) (void)hipFree(d_tickets_);
    if (d_batch_done_) (void)hipFree(d_batch_done_);
    if (h_points_) (void)hipHostFree(h_points_);
    if (h_out_) (void)hipHostFree(h_out_);
    if (h_nv_) (void)hipHostFree(h_nv_);
    if (h_done_) (void)hipHostFree(h_done_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

int Batch::create(vb2_ctx* const* ctxs, int num_sample, Batch** out)
{
    *out = nullptr;
    if (!ctxs || num_sample < 1) {
        set_error("vb2_batch_create: invalid argument");
        return VB2_ERR_INVALID;
    }
    std::unique_ptr<Batch> b(new Batch());
    b->num_sample = num_sample;
    int max_mt = 0, num_cu = 256;
    std::vector<DeviceLayout> layouts(num_sample);
    for (int s = 0; s < num_sample; ++s) {
        if (!ctxs[s] || !ctxs[s]->impl) {
            set_error("vb2_batch_create: null context");
            return VB2_ERR_INVALID;
        }
        Context* c = ctxs[s]->impl;
        if (s == 0) {
            b->device = c->device;
            b->num_pc = c->num_pc;
            num_cu = c->L.num_cu;
        } else if (c->device != b->device || c->num_pc != b->num_pc) {
            set_error("vb2_batch_create: contexts must share the device and --NumPC");
            return VB2_ERR_INVALID;
        }
        b->ctx_.push_back(c);
        layouts[s] = c->L;
        layouts[s].stamps = nullptr;
        if (c->L.row_bytes != kRowBytesWide) b->wide_rows_ = false;
        max_mt = std::max(max_mt, c->L.num_mt);
    }
    VB2_HIP(hipSetDevice(b->device));

    // geometry: ~one workgroup per CU in total; more workgroups per sample when the per-tile
    // result slots would not fit in LDS
    const int bps = std::max(1, num_cu / num_sample);
    b->bps_ = bps;
    const int tiles_per_block = (max_mt + bps - 1) / bps;
    const int k = b->num_pc;
    const int min_bw = std::max(4, (kSlot * (2 * k + 1) + 127) / 128);
    b->block_waves_ = std::max(min_bw, std::min(kMaxBlockWaves, tiles_per_block));
    // launch shapes of a step: 0 = up to 4 points per sample, 1 = 8 points, 2 = one point, 3 = two
    static const int kShapeNp[kShapes] = {4, 8, 1, 2};
    for (int sh = 0; sh < kShapes; ++sh) {
        size_t need = 0;
        for (int s = 0; s < num_sample; ++s)
            need = std::max(need, eval_shmem_np(layouts[s], kShapeNp[sh], bps, b->block_waves_, 1));
        b->shmem_[sh] = need;
    }
    if (b->shmem_[1] > (size_t)kLdsLimitBytes) {
        set_error("vb2_batch_create: per-workgroup LDS need exceeds 160 KiB");
        return VB2_ERR_INVALID;
    }

    const size_t S = (size_t)num_sample, stride = 2 * (size_t)k + 1;
    VB2_HIP(hipMalloc((void**)&b->d_layouts_, sizeof(DeviceLayout) * S));
    VB2_HIP(hipMemcpy(b->d_layouts_, layouts.data(), sizeof(DeviceLayout) * S, hipMemcpyHostToDevice));
    // static schedules (llk_kernels.h) of every sample for the wave shapes of a step: two micro-tiles
    // per wave for <= 4 points (when paired), one for 8 points, four for one or two points
    if (b->ctx_[0]->sched_enabled) {
        const int tpu[kShapes] = {paired_mode() ? 2 : 1, 1, paired_mode() ? 4 : 1, paired_mode() ? 4 : 1};
        std::vector<char> blob;
        std::vector<size_t> where[kShapes];
        bool ok = true;
        for (int sh = 0; sh < kShapes && ok; ++sh)
            for (int s = 0; s < num_sample && ok; ++s) {
                std::vector<uint32_t> off;
                std::vector<uint16_t> item;
                Context* c = b->ctx_[s];
                if (c->L.num_mt == 0) { where[sh].push_back((size_t)-1); continue; }
                ok = build_schedule(c->h_mt_rows.data(), c->L.num_mt, bps, b->block_waves_, tpu[sh], 1, &off, &item);
                if (!ok) break;
                blob.resize((blob.size() + 15) / 16 * 16);
                where[sh].push_back(blob.size());
                const size_t ob = (off.size() * sizeof(uint32_t) + 15) / 16 * 16;
                blob.resize(blob.size() + ob + item.size() * sizeof(uint16_t));
                std::memcpy(blob.data() + where[sh].back(), off.data(), off.size() * sizeof(uint32_t));
                std::memcpy(blob.data() + where[sh].back() + ob, item.data(), item.size() * sizeof(uint16_t));
            }
        if (ok) {
            blob.resize((blob.size() + 15) / 16 * 16);
            const size_t arr0 = blob.size();
            VB2_HIP(hipMalloc((void**)&b->d_sched_, arr0 + kShapes * S * sizeof(Schedule)));
            std::vector<Schedule> arr(kShapes * S, Schedule{nullptr, nullptr});
            const size_t ob = (((size_t)bps * b->block_waves_ + 1) * sizeof(uint32_t) + 15) / 16 * 16;
            for (int sh = 0; sh < kShapes; ++sh)
                for (size_t s = 0; s < S; ++s) {
                    const size_t w = where[sh][s];
                    if (w == (size_t)-1) continue;
                    const uint32_t* o = reinterpret_cast<const uint32_t*>(b->d_sched_ + w);
                    arr[sh * S + s] = Schedule{o, reinterpret_cast<const uint16_t*>(b->d_sched_ + w + ob)};
                }
            VB2_HIP(hipMemcpy(b->d_sched_, blob.data(), arr0, hipMemcpyHostToDevice));
            VB2_HIP(hipMemcpy(b->d_sched_ + arr0, arr.data(), kShapes * S * sizeof(Schedule), hipMemcpyHostToDevice));
            for (int sh = 0; sh < kShapes; ++sh) b->d_scheds_[sh] = reinterpret_cast<const Schedule*>(b->d_sched_ + arr0) + sh * S;
        }
    }
    VB2_HIP(hipMalloc((void**)&b->d_partials_, sizeof(double) * S * (kSlot + 1) * bps));
    VB2_HIP(hipMemset(b->d_partials_, 0, sizeof(double) * S * (kSlot + 1) * bps));
    VB2_HIP(hipMalloc((void**)&b->d_tickets_, sizeof(unsigned int) * S));
    VB2_HIP(hipMemset(b->d_tickets_, 0, sizeof(unsigned int) * S));
    VB2_HIP(hipMalloc((void**)&b->d_batch_done_, sizeof(unsigned int)));
    VB2_HIP(hipMemset(b->d_batch_done_, 0, sizeof(unsigned int)));
    VB2_HIP(hipHostMalloc((void**)&b->h_points_, sizeof(double) * S * kSlot * stride, hipHostMallocMapped));
    VB2_HIP(hipHostMalloc((void**)&b->h_out_, sizeof(double) * S * kSlot, hipHostMallocMapped));
    VB2_HIP(hipHostMalloc((void**)&b->h_nv_, sizeof(int) * S, hipHostMallocMapped));
    VB2_HIP(hipHostMalloc((void**)&b->h_done_, sizeof(unsigned long long), hipHostMallocMapped));
    *b->h_done_ = 0;
    VB2_HIP(hipHostGetDevicePointer((void**)&b->d_points_, b->h_points_, 0));
    VB2_HIP(hipHostGetDevicePointer((void**)&b->d_out_, b->h_out_, 0));
    VB2_HIP(hipHostGetDevicePointer((void**)&b->d_nv_, b->h_nv_, 0));
    VB2_HIP(hipHostGetDevicePointer((void**)&b->d_done_, b->h_done_, 0));
    VB2_HIP(hipStreamCreateWithFlags(&b->stream_, hipStreamNonBlocking));
    VB2_HIP(hipDeviceSynchronize());
    *out = b.release();
    return VB2_OK;
}

int Batch::eval(const int32_t* num_point, const double* pc1, const double* pc2, const double* alpha,
                double* llk_out)
{
    VB2_HIP(hipSetDevice(device));
    const int k = num_pc, stride = 2 * k + 1;
    int max_n = 0;
    unsigned int active = 0;
    for (int s = 0; s < num_sample; ++s) {
        if (num_point[s] < 0 || num_point[s] > kSlot) {
            set_error("vb2_batch_eval: num_point out of range");
            return VB2_ERR_INVALID;
        }
        max_n = std::max(max_n, (int)num_point[s]);
    }
    if (max_n == 0) return VB2_OK;
    if (max_n > 4 && !wide_rows_) {
        // a sample with a very wide dictionary has narrow table rows: 4 points per launch at most
        const size_t S = (size_t)num_sample;
        std::vector<int32_t> np(S);
        std::vector<double> p1(S * kSlot * k), p2(S * kSlot * k), al(S * kSlot), lo(S * kSlot);
        for (int half = 0; half < 2; ++half) {
            for (size_t s = 0; s < S; ++s) {
                np[s] = std::max(0, std::min(4, (int)num_point[s] - 4 * half));
                for (int j = 0; j < np[s]; ++j) {
                    const size_t src = s * kSlot + 4 * half + j, dst = s * kSlot + j;
                    std::memcpy(&p1[dst * k], pc1 + src * k, sizeof(double) * k);
                    std::memcpy(&p2[dst * k], pc2 + src * k, sizeof(double) * k);
                    al[dst] = alpha[src];
                }
            }
            if (int rc = eval(np.data(), p1.data(), p2.data(), al.data(), lo.data())) return rc;
            for (size_t s = 0; s < S; ++s)
                for (int j = 0; j < np[s]; ++j) llk_out[s * kSlot + 4 * half + j] = lo[s * kSlot + j];
        }
        return VB2_OK;
    }
    // one or two points per sample (a search that speculates little or not at all): the wave takes
    // four micro-tiles, and the step costs a quarter / half of a 4-point step whose other slots
    // would replicate the last point
    const bool small = max_n <= 2 && paired_mode();
    const int NP = max_n > 4 ? 8 : small ? max_n : 4, shape = max_n > 4 ? 1 : small ? (max_n == 1 ? 2 : 3) : 0;
    for (int s = 0; s < num_sample; ++s) {
        int n = num_point[s];
        // a sample without active markers has LLK 0 for every point: answer it here
        if (n > 0 && ctx_[s]->L.num_mt == 0) {
            for (int j = 0; j < n; ++j) llk_out[(size_t)s * kSlot + j] = 0.0;
            n = 0;
        }
        h_nv_[s] = n;
        if (n > 0) ++active;
        for (int j = 0; j < n; ++j) {
            double* row = h_points_ + ((size_t)s * NP + j) * stride;
            std::memcpy(row, pc1 + ((size_t)s * kSlot + j) * k, sizeof(double) * k);
            std::memcpy(row + k, pc2 + ((size_t)s * kSlot + j) * k, sizeof(double) * k);
            row[2 * k] = alpha[(size_t)s * kSlot + j];
        }
    }
    if (active == 0) return VB2_OK;
    MultiLaunch ml{};
    ml.d_layouts = d_layouts_;
    ml.d_scheds = d_scheds_[shape];
    ml.d_points = d_points_;
    ml.d_num_valid = d_nv_;
    ml.d_partials = d_partials_;
    ml.d_out = d_out_;
    ml.d_tickets = d_tickets_;
    ml.d_batch_done = d_batch_done_;
    ml.done_flag = d_done_;
    ml.done_seq = ++seq_;
    ml.batch_active = active;
    ml.num_sample = num_sample;
    ml.bps = bps_;
    ml.block_waves = block_waves_;
    ml.np = NP;
    ml.shmem = shmem_[shape];
    // (a NaN is the tagged hand-off's "a workgroup never reported" marker: redo the step once with
    // the arrival-ticket hand-off, see Context::eval_host -- NaN must not reach the optimisers)
    for (int attempt = 0; attempt < 2; ++attempt) {
        ml.force_ticket = attempt > 0;
        if (attempt > 0) ml.done_seq = ++seq_;
        VB2_HIP(launch_llk_eval_multi(ml, stream_));
        ++num_launch;
        bool seen = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(h_done_, __ATOMIC_ACQUIRE) == ml.done_seq) { seen = true; break; }
            if ((spins & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
            __builtin_ia32_pause();
        }
        if (!seen) VB2_HIP(hipStreamSynchronize(stream_));
        bool any_nan = false;
        for (int s = 0; s < num_sample && !any_nan; ++s)
            for (int j = 0; j < h_nv_[s]; ++j) any_nan |= std::isnan(h_out_[(size_t)s * NP + j]);
        if (!any_nan) break;
    }
    for (int s = 0; s < num_sample; ++s)
        for (int j = 0; j < h_nv_[s]; ++j) llk_out[(size_t)s * kSlot + j] = h_out_[(size_t)s * NP + j];
    return VB2_OK;
}

// ---------------------------------------------------------------------------
// lock-step search: every sample runs the ordinary Estimator (OptimizeLLK with its six models,
// the reference-exact simplex) as a fiber of the calling thread (lockstep.h); the parked
// requests of a step leave as ONE launch of the multi-sample kernel.
// ---------------------------------------------------------------------------
int Batch::optimize(const vb2_model* models, int num_model, vb2_estimate* out)
{
    if (!models || !out || (num_model != 1 && num_model != num_sample)) {
        set_error("vb2_batch_optimize_llk: pass 1 model or one per sample");
        return VB2_ERR_INVALID;
    }
    // Speculation (amoeba.h) trades device work for fewer dependent steps.  A small cohort's step is
    // latency-bound and {R, E, C_A, C_R} per iteration pays (2.5 points evaluated per point the
    // search needs).  A big cohort's step is throughput-bound -- its cost grows with the points in
    // it, down to the floor of streaming every sample's pileup from HBM once per step (5.7 TB/s
    // measured: a 1-point step of 32 C3 samples takes 131 us against 236 us with 4 points) -- and
    // {R, C_R} is the better trade: 1.2 steps per iteration at about half the points.  Same
    // decisions, same trajectory either way.  VB2_COHORT_SPECULATE=1|2|4 forces one.
    constexpr int kPairFrom = 8;
    speculate_ = num_sample < kPairFrom ? 4 : 2;
    if (const char* e = std::getenv("VB2_COHORT_SPECULATE")) speculate_ = std::max(1, std::atoi(e));

    const int S = num_sample, k = num_pc;
    FiberGang gang(S, kSlot);
    std::vector<int> rcs(S, 0);
    std::vector<int32_t> npts(S, 0);
    std::vector<double> pc1((size_t)S * kSlot * k), pc2((size_t)S * kSlot * k), alpha((size_t)S * kSlot),
        llk((size_t)S * kSlot);
    auto body = [&](int s) {
        try {
            const vb2_model& m = models[num_model == 1 ? 0 : s];
            Estimator est(num_pc, FiberGang::eval_cb, gang.user(s));
            apply_model(est, m);
            est.speculate = speculate_;
            if (ctx_[s]->L.known_af) {       // context built with --KnownAF data
                est.isAFknown = true;
                est.isPCFixed = true;
                est.isHeter = false;
            }
            rcs[s] = est.OptimizeLLK();
            fill_estimate(est, &out[s]);
        } catch (const std::bad_alloc&) {
            rcs[s] = VB2_ERR_NOMEM;
        } catch (const std::exception& e) {
            set_error(e.what());
            rcs[s] = VB2_ERR_INVALID;
        } catch (...) {                      // nothing may unwind past the fiber's entry frame
            set_error("vb2_batch_optimize_llk: unknown exception in a sample's search");
            rcs[s] = VB2_ERR_INVALID;
        }
    };
    auto step = [&](std::vector<FiberGang::Request>& req) {
        for (int s = 0; s < S; ++s) {
            const FiberGang::Request& r = req[s];
            npts[s] = r.n;
            if (r.n <= 0) continue;
            std::memcpy(&pc1[(size_t)s * kSlot * k], r.p1, sizeof(double) * r.n * k);
            std::memcpy(&pc2[(size_t)s * kSlot * k], r.p2, sizeof(double) * r.n * k);
            std::memcpy(&alpha[(size_t)s * kSlot], r.a, sizeof(double) * r.n);
        }
        if (const int rc = eval(npts.data(), pc1.data(), pc2.data(), alpha.data(), llk.data())) return rc;
        for (int s = 0; s < S; ++s)
            if (npts[s] > 0) std::memcpy(req[s].out, &llk[(size_t)s * kSlot], sizeof(double) * npts[s]);
        return 0;
    };
    const int rc = gang.run(k, body, step);
    if (rc < 0) {
        set_error("vb2_batch_optimize_llk: getcontext failed");
        return VB2_ERR_INVALID;
    }
    if (rc) return rc;
    for (int s = 0; s < S; ++s)
        if (rcs[s]) return rcs[s];
    return VB2_OK;
}

}  // namespace vb2
