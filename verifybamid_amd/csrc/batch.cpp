// batch.cpp -- see batch.h.
#include "batch.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

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
constexpr int kSlot = 8;                     // point slots per sample and step (VB2_BATCH_SLOTS)
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
    for (int btl = 1; btl <= 2; ++btl) {
        size_t need = 0;
        for (int s = 0; s < num_sample; ++s)
            need = std::max(need, eval_shmem_bytes(layouts[s], btl, bps, b->block_waves_));
        b->shmem_[btl - 1] = need;
    }
    if (b->shmem_[1] > (size_t)kLdsLimitBytes) {
        set_error("vb2_batch_create: per-workgroup LDS need exceeds 160 KiB");
        return VB2_ERR_INVALID;
    }

    const size_t S = (size_t)num_sample, stride = 2 * (size_t)k + 1;
    VB2_HIP(hipMalloc((void**)&b->d_layouts_, sizeof(DeviceLayout) * S));
    VB2_HIP(hipMemcpy(b->d_layouts_, layouts.data(), sizeof(DeviceLayout) * S, hipMemcpyHostToDevice));
    // static schedules (llk_kernels.h) of every sample for the two wave shapes of a step:
    // btl 1 = <= 4 points per sample (two tiles per wave when paired), btl 2 = 8 points
    if (b->ctx_[0]->sched_enabled) {
        std::vector<char> blob;
        std::vector<size_t> where[2];
        bool ok = true;
        for (int btl = 1; btl <= 2 && ok; ++btl)
            for (int s = 0; s < num_sample && ok; ++s) {
                std::vector<uint32_t> off;
                std::vector<uint16_t> item;
                Context* c = b->ctx_[s];
                if (c->L.num_mt == 0) { where[btl - 1].push_back((size_t)-1); continue; }
                ok = build_schedule(c->h_mt_rows.data(), c->L.num_mt, bps, b->block_waves_,
                                    btl == 1 && paired_mode() ? 2 : 1, 1, &off, &item);
                if (!ok) break;
                blob.resize((blob.size() + 15) / 16 * 16);
                where[btl - 1].push_back(blob.size());
                const size_t ob = (off.size() * sizeof(uint32_t) + 15) / 16 * 16;
                blob.resize(blob.size() + ob + item.size() * sizeof(uint16_t));
                std::memcpy(blob.data() + where[btl - 1].back(), off.data(), off.size() * sizeof(uint32_t));
                std::memcpy(blob.data() + where[btl - 1].back() + ob, item.data(), item.size() * sizeof(uint16_t));
            }
        if (ok) {
            blob.resize((blob.size() + 15) / 16 * 16);
            const size_t arr0 = blob.size();
            VB2_HIP(hipMalloc((void**)&b->d_sched_, arr0 + 2 * S * sizeof(Schedule)));
            std::vector<Schedule> arr(2 * S, Schedule{nullptr, nullptr});
            for (int btl = 1; btl <= 2; ++btl)
                for (size_t s = 0; s < S; ++s) {
                    const size_t w = where[btl - 1][s];
                    if (w == (size_t)-1) continue;
                    const Context* c = b->ctx_[s];
                    (void)c;
                    const uint32_t* o = reinterpret_cast<const uint32_t*>(b->d_sched_ + w);
                    const size_t ob = (((size_t)bps * b->block_waves_ + 1) * sizeof(uint32_t) + 15) / 16 * 16;
                    arr[(btl - 1) * S + s] = Schedule{o, reinterpret_cast<const uint16_t*>(b->d_sched_ + w + ob)};
                }
            VB2_HIP(hipMemcpy(b->d_sched_, blob.data(), arr0, hipMemcpyHostToDevice));
            VB2_HIP(hipMemcpy(b->d_sched_ + arr0, arr.data(), 2 * S * sizeof(Schedule), hipMemcpyHostToDevice));
            b->d_scheds_[0] = reinterpret_cast<const Schedule*>(b->d_sched_ + arr0);
            b->d_scheds_[1] = b->d_scheds_[0] + S;
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
    const int btl = max_n > 4 ? 2 : 1, NP = 4 * btl;
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
    ml.d_scheds = d_scheds_[btl - 1];
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
    ml.btl = btl;
    ml.shmem = shmem_[btl - 1];
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
// lock-step search: one host thread per sample runs the ordinary Estimator; their
// evaluation requests meet in a rendezvous and leave as one launch.
// ---------------------------------------------------------------------------
namespace {

struct Rendezvous {
    Batch* batch;
    int S, k;
    std::mutex mu;
    std::condition_variable cv;
    int active, arrived = 0;
    unsigned long long generation = 0;
    int error = 0;
    std::vector<int32_t> npts;
    std::vector<double> pc1, pc2, alpha, out;

    Rendezvous(Batch* b, int s, int kk)
        : batch(b), S(s), k(kk), active(s), npts(s, 0), pc1((size_t)s * kSlot * kk), pc2((size_t)s * kSlot * kk),
          alpha((size_t)s * kSlot), out((size_t)s * kSlot) {}

    void run_locked()
    {
        const int rc = batch->eval(npts.data(), pc1.data(), pc2.data(), alpha.data(), out.data());
        if (rc && !error) error = rc;
        std::fill(npts.begin(), npts.end(), 0);
        arrived = 0;
        ++generation;
        cv.notify_all();
    }

    int submit(int s, int n, const double* p1, const double* p2, const double* a, double* o)
    {
        std::unique_lock<std::mutex> lk(mu);
        std::memcpy(&pc1[(size_t)s * kSlot * k], p1, sizeof(double) * n * k);
        std::memcpy(&pc2[(size_t)s * kSlot * k], p2, sizeof(double) * n * k);
        std::memcpy(&alpha[(size_t)s * kSlot], a, sizeof(double) * n);
        npts[s] = n;
        ++arrived;
        const unsigned long long gen = generation;
        if (arrived == active) run_locked();
        else cv.wait(lk, [&] { return generation != gen; });
        std::memcpy(o, &out[(size_t)s * kSlot], sizeof(double) * n);
        return error;
    }

    void leave()
    {
        std::unique_lock<std::mutex> lk(mu);
        --active;
        if (active > 0 && arrived == active) run_locked();
    }
};

struct SampleCb {
    Rendezvous* rv;
    int s;
};

int sample_eval(void* user, int32_t n, const double* p1, const double* p2, const double* a, double* o)
{
    SampleCb* cb = static_cast<SampleCb*>(user);
    const int k = cb->rv->k;
    for (int done = 0; done < n; done += kSlot) {
        const int m = std::min(kSlot, n - done);
        const int rc = cb->rv->submit(cb->s, m, p1 + (size_t)done * k, p2 + (size_t)done * k, a + done, o + done);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

int Batch::optimize(const vb2_model* models, int num_model, vb2_estimate* out)
{
    if (!models || !out || (num_model != 1 && num_model != num_sample)) {
        set_error("vb2_batch_optimize_llk: pass 1 model or one per sample");
        return VB2_ERR_INVALID;
    }
    Rendezvous rv(this, num_sample, num_pc);
    std::vector<SampleCb> cbs(num_sample);
    std::vector<int> rcs(num_sample, 0);
    std::vector<std::thread> threads;
    threads.reserve(num_sample);
    for (int s = 0; s < num_sample; ++s) {
        cbs[s] = SampleCb{&rv, s};
        threads.emplace_back([&, s]() {
            const vb2_model& m = models[num_model == 1 ? 0 : s];
            Estimator est(num_pc, sample_eval, &cbs[s]);
            apply_model(est, m);
            if (ctx_[s]->L.known_af) {       // context built with --KnownAF data
                est.isAFknown = true;
                est.isPCFixed = true;
                est.isHeter = false;
            }
            rcs[s] = est.OptimizeLLK();
            fill_estimate(est, &out[s]);
            rv.leave();
        });
    }
    for (auto& t : threads) t.join();
    if (rv.error) return rv.error;
    for (int s = 0; s < num_sample; ++s)
        if (rcs[s]) return rcs[s];
    return VB2_OK;
}

}  // namespace vb2
