// cohort.cpp -- vb2_cohort_run: many samples against one panel (BASELINE.json configs[4]).
//
// The reference runs one process per sample (main.cpp:56-414).  Here the panel is read once, a
// pool of host threads reads + flattens + uploads pileups, and every DEVICE of the run has its
// own pipeline thread that searches the samples in lock-step (one kernel launch per Nelder-Mead
// step for all the samples on the device).  Round 4: the device keeps `group_size` slots and a
// sample that has converged hands its slot to the next one that is ready (stream_search.h) --
// before, a device took the next ready GROUP of samples and searched it to its last sample
// (vb2_batch_*; still there as VB2_COHORT_STREAM=0).  Samples are independent, so several devices
// need no collective: sample s runs on devices[s % n] (--Devices a,b,...; groups: group g on
// devices[g % n]), which is the sample-parallel sharding of SURVEY.md 8e(2) inside one process.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "batch.h"
#include "context.h"
#include "estimator.h"
#include "hostio.h"
#include "stream_search.h"
#include "tunables.h"

using vb2::set_error;

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Slot {
    std::unique_ptr<vb2_flat> flat;
    vb2_ctx* ctx = nullptr;
    int rc = VB2_OK;
    bool ready = false;
};

class CohortRunner {
public:
    CohortRunner(const vb2_cohort_args* a, vb2_run_result* out, int32_t* status)
        : a_(a), out_(out), status_(status), S_(a->num_sample)
    {
        if (a->base.devices && a->base.num_device > 0) devices_.assign(a->base.devices, a->base.devices + a->base.num_device);
        else devices_.push_back(a->base.device);
        ndev_ = (int)devices_.size();
        // (round 4: 32, not 64 -- with the finished samples regrouped (Batch::optimize) a group of 32 costs 1.55 ms per C3
        // sample and one of 64 1.5, and the smaller group starts earlier and leaves a shorter tail nothing overlaps:
        // 256 files 479 -> 546 samples/s)
        G_ = std::max(1, std::min(a->group_size != 0 ? std::abs(a->group_size) : 32, 64));
        const int hw = vb2::usable_cpu_count();
        // readers: reading + resolving + creating the context is ~7 ms of one CPU per C3-sized sample (the flatten runs on the
        // device), the device needs ~1.65 ms per sample -> a device keeps 4-5 readers busy.  The default is SIX per device,
        // within the process's allowance (cgroup quota / affinity, not the host's core count) less one CPU per pipeline
        // thread and one for the rest: the thread of a lock-step search spins on its device's flag between steps, and when it
        // has to queue for a CPU -- or the whole cgroup is throttled because fifteen readers woke up at once -- the device
        // idles (16 CPUs, 256 files: 15 readers 562-587 samples/s with runs at 390; 5 to 10 readers 600-633)
        // (round 6: the device needs 1.2 ms per sample, the host 9 ms of one CPU -> EIGHT per device: 4 / 6 / 8 / 10 / 12
        // readers 500 / 705 / 740 / 750 / 750 samples/s on 16 CPUs)
        const int dflt = std::max(2, std::min(hw - 1 - ndev_, 8 * ndev_));
        T_ = std::max(1, std::min({a->num_host_thread > 0 ? a->num_host_thread : dflt, std::max(hw, 1) * 4, S_}));
        slots_.resize(S_);
        cnt_.assign(ndev_, 0);
        stream_ = vb2::tunables().cohort_stream != 0;
        inflight_.assign(ndev_, 0);
        remaining_.assign(ndev_, 0);
        for (int s = 0; s < S_; ++s) ++remaining_[s % ndev_];
        ready_q_.resize(ndev_);
        // Group boundaries, a function of (S, G, devices) only -- never of timing, so a run is
        // reproducible.  A device's first groups are small (16, then 32 samples): it starts searching
        // after a fraction of the reading a full group needs; the later ones have the full size (a
        // bigger group costs less per sample: 3.6 / 3.25 / 3.0 ms at 16 / 32 / 48 C3-sized samples).
        // A remainder of up to half a group joins the last group instead of running alone.
        for (int begin = 0, gi = 0; begin < S_; ++gi) {
            const int nth = gi / ndev_;                       // this group's index on its device
            int want = std::min(G_, nth == 0 ? 16 : nth == 1 ? 32 : G_);     // (G_ > 32: 16, 32, then full groups)
            if (a->group_size < 0) want = G_;                 // (negative group_size: plain equal groups of |group_size|)
            const int left = S_ - begin;
            if (left <= want + want / 2) want = left;
            group_begin_.push_back(begin);
            for (int s = begin; s < begin + want; ++s) group_of_.push_back(gi);
            begin += want;
        }
        group_begin_.push_back(S_);
        ngroup_ = (int)group_begin_.size() - 1;
    }

    ~CohortRunner() { shutdown(); }

    int run()
    {
        for (int s = 0; s < S_; ++s) {
            std::memset(&out_[s], 0, sizeof(out_[s]));
            status_[s] = VB2_ERR_INVALID;
        }
        const bool timing = vb2::tunables().debug_timing != 0;
        const double t_run0 = now_s();
        // HIP runtime start-up (per device) behind the panel reading
        std::vector<std::thread> warm;
        for (int d : devices_)
            warm.emplace_back([d] {
                if (d >= 0) (void)hipSetDevice(d);
                (void)hipFree(nullptr);
                (void)hipGetLastError();
            });
        panel_ = std::make_shared<vb2::Panel>();
        panel_->numPC = a_->base.num_pc;
        int rc = VB2_OK;
        try {
            rc = vb2::load_panel(&a_->base, panel_.get());
        } catch (...) {
            for (auto& t : warm) t.join();
            throw;
        }
        const double t_panel = now_s();
        for (auto& t : warm) t.join();
        if (rc) return rc;
        const double t_warm = now_s();
        model_ = a_->base.model;
        if (panel_->isAFknown) model_.is_af_known = 1;
        vb2::g_flatten_thread_cap.store(std::max(1, 16 / T_));

        // HIP maps a process's streams onto FOUR hardware queues per device, each in order.  A context's creation (uploads, the
        // flatten's kernels, and a barrier that waits for the 10-MB upload) on a stream that shares its hardware queue with a lane
        // of the search holds that lane's next step back for as long -- a few hundred microseconds, for half of the samples when
        // every context and every lane takes whatever stream the cache has (kernel trace of tools/cohort_from_text.py: steps
        // on two queues, the flatten's kernels on all four).  So: per device four streams made back to back, after the idle cached
        // ones are gone (the runtime gives a new stream the least used queue: four queues), two for the lanes and two that every
        // context of this run is created on.
        if (stream_ && vb2::tunables().cohort_own_queues) {
            own_streams_.assign((size_t)ndev_ * 4, nullptr);
            for (int d = 0; d < ndev_; ++d) {
                int dev = devices_[d];
                if (dev < 0) (void)hipGetDevice(&dev);
                if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); continue; }
                vb2::drop_cached_streams(dev);
                for (int q = 0; q < 4; ++q)
                    if (hipStreamCreateWithFlags(&own_streams_[(size_t)d * 4 + q], hipStreamNonBlocking) != hipSuccess) {
                        (void)hipGetLastError();
                        own_streams_[(size_t)d * 4 + q] = nullptr;
                    }
            }
        }
        releaser_ = std::thread([this] { release_loop(); });
        for (int t = 0; t < T_; ++t) pool_.emplace_back([this] { reader_loop(); });
        for (int d = 0; d < ndev_; ++d) dev_threads_.emplace_back([this, d] { if (stream_) stream_loop(d); else device_loop(d); });
        for (auto& t : dev_threads_) t.join();
        dev_threads_.clear();
        const double t_dev = now_s();
        shutdown();
        if (timing)
            std::fprintf(stderr, "vb2_cohort_run: panel %.1f ms, HIP start-up beyond that %.1f ms, pipelines %.1f ms, "
                                 "shutdown (release contexts, join) %.1f ms; releaser: %.2f ms per context, %.2f ms per sample's host arrays\n",
                         1e3 * (t_panel - t_run0), 1e3 * (t_warm - t_panel), 1e3 * (t_dev - t_warm), 1e3 * (now_s() - t_dev),
                         1e3 * rel_s_[0] / S_, 1e3 * rel_s_[1] / S_);
        if (timing)
            std::fprintf(stderr, "vb2_cohort_run: %d reader threads, per sample: read_pileup %.1f ms, sanity + resolve %.1f ms, "
                                 "vb2_ctx_create %.1f ms (wall-clock inside the reader threads)\n", T_,
                         1e3 * phase_s_[0] / S_, 1e3 * phase_s_[1] / S_, 1e3 * phase_s_[2] / S_);
        return rc_all_;
    }

private:
    const vb2_cohort_args* a_;
    vb2_run_result* out_;
    int32_t* status_;
    int S_, G_ = 64, T_ = 1, ndev_ = 1, ngroup_ = 0;
    std::vector<int> group_begin_, group_of_;       // [group] first sample (+ S_ at the end); [sample] group
    std::vector<int> devices_;
    std::shared_ptr<vb2::Panel> panel_;
    vb2_model model_{};
    std::vector<Slot> slots_;
    std::mutex mu_;
    std::condition_variable cv_;
    int next_ = 0;
    std::vector<int> cnt_;                 // groups completed per device
    bool stop_ = false;
    int rc_all_ = VB2_OK;
    std::string err_all_;
    std::vector<hipStream_t> own_streams_;      // [device][4]: two lanes' streams, two creation streams (or empty)
    std::vector<std::thread> pool_, dev_threads_;
    std::thread releaser_;
    std::mutex rel_mu_;
    std::condition_variable rel_cv_;
    std::deque<std::pair<vb2_ctx*, std::unique_ptr<vb2_flat>>> rel_queue_;
    bool rel_stop_ = false;
    bool down_ = false;

    double rel_s_[2] = {0, 0};             // releaser thread: vb2_ctx_destroy, freeing the host arrays
    double phase_s_[3] = {0, 0, 0};        // reader threads' wall-clock: read_pileup, sanity + resolve, vb2_ctx_create

    // streaming (stream_search.h): sample s belongs to device s % ndev_
    bool stream_ = true;
    std::vector<int> inflight_;            // [device] samples handed to a reader and not yet done
    std::vector<int> remaining_;           // [device] samples not yet delivered to the search (or failed before it)
    std::vector<std::deque<int>> ready_q_; // [device] prepared samples, in the order they became ready
    struct Done { int s; vb2_ctx* ctx; std::unique_ptr<vb2_flat> flat; bool write; vb2_estimate est; };
    std::deque<Done> done_queue_;          // (rel_mu_) searched samples: outputs to write, context and arrays to free

    int device_of_group(int gi) const { return gi % ndev_; }
    int device_of_sample(int s) const { return stream_ ? s % ndev_ : device_of_group(group_of_[s]); }

    // every way out (normal end, error, exception in run()) stops and joins all threads and
    // gives back the contexts that were never handed to a device pipeline
    void shutdown()
    {
        if (down_) return;
        down_ = true;
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : dev_threads_)
            if (t.joinable()) t.join();
        for (auto& t : pool_)
            if (t.joinable()) t.join();
        {
            std::lock_guard<std::mutex> lk(rel_mu_);
            rel_stop_ = true;
        }
        rel_cv_.notify_all();
        if (releaser_.joinable()) releaser_.join();
        for (auto& sl : slots_)
            if (sl.ctx) {
                vb2_ctx_destroy(sl.ctx);
                sl.ctx = nullptr;
            }
        vb2::g_flatten_thread_cap.store(0);
        for (hipStream_t& st : own_streams_)             // (every context and batch that used them is gone)
            if (st) {
                (void)hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
                st = nullptr;
            }
    }

    void prepare(int s)
    {
        Slot& sl = slots_[s];
        const double t0 = now_s();
        sl.flat.reset(new vb2_flat(panel_));
        vb2_flat& f = *sl.flat;
        sl.rc = vb2::read_pileup(a_->pileup_paths[s], *panel_, &f.viewer);
        if (sl.rc) return;
        const double t_read = now_s();
        const bool sanity_off = a_->base.disable_sanity != 0;
        f.sanity_disabled = sanity_off;
        const bool sane = sanity_off || vb2::sanity_check(*panel_, &f.viewer);
        f.resolve();
        vb2_flat_stats(&f, &out_[s]);
        const double t_res = now_s();
        const char* prefix = a_->output_prefixes ? a_->output_prefixes[s] : nullptr;
        if (a_->base.output_pileup && prefix) (void)vb2::write_pileup(prefix, f);
        if (!sane) {
            sl.rc = VB2_ERR_SANITY;
            return;
        }
        vb2_options opt{};
        opt.device = devices_[device_of_sample(s)];
        opt.flags = VB2_OPT_COHORT_LAYOUT;       // the lock-step search streams the 16-bit run lists
        if (!own_streams_.empty())               // (one of the device's two creation streams: see run())
            opt.stream = own_streams_[(size_t)device_of_sample(s) * 4 + 2 + (size_t)(s / ndev_) % 2];
        sl.rc = vb2_ctx_create(&f.input, &opt, &sl.ctx);
        if (sl.rc == VB2_OK && sl.ctx && sl.ctx->impl) {
            // the static schedules of the sample's group (and of the batches its lane is regrouped into): here, on one of many
            // reader threads, not on the one thread that feeds the device (a failure only means they are built there)
            const int gi = group_of_[s];
            if (stream_) (void)vb2::prepare_for_stream(sl.ctx->impl, G_);
            else (void)vb2::Batch::prepare_for_cohort(sl.ctx->impl, group_begin_[gi + 1] - group_begin_[gi]);
        }
        // the context holds its own (flattened) copy: the sample's text-sized arrays go back now, on
        // this reader thread, instead of piling up in front of the single releaser (the writers need
        // only the viewer's counters and the panel)
        std::string().swap(f.bases);
        std::string().swap(f.quals);
        std::string().swap(f.viewer.basePool);
        std::string().swap(f.viewer.qualPool);
        std::vector<int64_t>().swap(f.read_off);
        f.input = vb2_input{};
        const double t_end = now_s();
        out_[s].seconds_load = t_end - t0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            phase_s_[0] += t_read - t0;
            phase_s_[1] += t_res - t_read;
            phase_s_[2] += t_end - t_res;
        }
    }

    void reader_loop()
    {
        for (;;) {
            int s;
            {
                std::unique_lock<std::mutex> lk(mu_);
                // stay at most two groups ahead of the group's device (bounds host and device memory)
                cv_.wait(lk, [&] {
                    if (stop_ || next_ >= S_) return true;
                    if (stream_) return inflight_[next_ % ndev_] < 2 * G_ + T_;       // (two slot sets ahead of the device)
                    const int gi = group_of_[next_];
                    return gi / ndev_ < cnt_[device_of_group(gi)] + 2;
                });
                if (stop_ || next_ >= S_) return;
                s = next_++;
                if (stream_) ++inflight_[s % ndev_];
            }
            try {
                prepare(s);
            } catch (const std::bad_alloc&) {
                slots_[s].rc = VB2_ERR_NOMEM;
            } catch (const std::exception&) {
                slots_[s].rc = VB2_ERR_INVALID;
            }
            if (stream_ && !(slots_[s].rc == VB2_OK && slots_[s].ctx)) {
                // never reaches the search: reported here (the pipeline thread may be waiting for this device's last sample)
                status_[s] = slots_[s].rc ? slots_[s].rc : VB2_ERR_INVALID;
                finish_sample(s, false, nullptr, /*delivered=*/false);
                continue;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                slots_[s].ready = true;
                if (stream_) ready_q_[s % ndev_].push_back(s);
            }
            cv_.notify_all();
        }
    }

    // sample s leaves the pipeline: outputs (write) + context + host arrays go to the releaser thread, its place in the
    // readers' window is free again.  delivered: the search took the sample (DeviceSource::next counted it out of
    // remaining_ then) -- whether or not the search then succeeded; a sample that fails on a reader thread never
    // gets that far and is counted out here.
    void finish_sample(int s, bool write, const vb2_estimate* est, bool delivered)
    {
        {
            std::lock_guard<std::mutex> lk(rel_mu_);
            Done d{s, slots_[s].ctx, std::move(slots_[s].flat), write, vb2_estimate{}};
            if (est) d.est = *est;
            slots_[s].ctx = nullptr;
            done_queue_.push_back(std::move(d));
        }
        rel_cv_.notify_one();
        {
            std::lock_guard<std::mutex> lk(mu_);
            const int d = s % ndev_;
            --inflight_[d];
            if (!delivered) --remaining_[d];
        }
        cv_.notify_all();
    }

    // the pipeline thread's view of its device's samples (stream_search.h)
    struct DeviceSource : vb2::StreamSource {
        CohortRunner* r;
        int d;
        DeviceSource(CohortRunner* runner, int dev) : r(runner), d(dev) {}
        int next(bool block, vb2::Context** ctx) override
        {
            std::unique_lock<std::mutex> lk(r->mu_);
            for (;;) {
                if (r->stop_) return kEnd;
                if (!r->ready_q_[d].empty()) {
                    const int s = r->ready_q_[d].front();
                    r->ready_q_[d].pop_front();
                    --r->remaining_[d];
                    *ctx = r->slots_[s].ctx->impl;
                    return s;
                }
                if (r->remaining_[d] <= 0) return kEnd;
                if (!block) return kNone;
                r->cv_.wait(lk);
            }
        }
        const vb2_model& model(int) override { return r->model_; }
        void done(int s, int rc, const vb2_estimate& est, double seconds) override
        {
            r->status_[s] = rc;
            if (!rc) {
                r->out_[s].est = est;
                r->out_[s].seconds_optimize = seconds;
            }
            r->finish_sample(s, rc == VB2_OK, &est, /*delivered=*/true);
        }
    };

    void stream_loop(int d)
    {
        const bool timing = vb2::tunables().debug_timing != 0;
        const double t0 = now_s();
        DeviceSource src(this, d);
        int rc = VB2_OK;
        int num_cu = 256;
        {
            hipDeviceProp_t prop;
            const int dev = devices_[d];
            int cur = dev;
            if (dev < 0) (void)hipGetDevice(&cur);
            if (hipGetDeviceProperties(&prop, cur) == hipSuccess && prop.multiProcessorCount > 0) num_cu = prop.multiProcessorCount;
        }
        try {
            int dev = devices_[d];
            if (dev < 0) (void)hipGetDevice(&dev);
            rc = vb2::stream_search(dev, a_->base.num_pc, num_cu, G_, src, own_streams_.empty() ? nullptr : &own_streams_[(size_t)d * 4]);
        } catch (const std::exception& e) {
            set_error(e.what());
            rc = VB2_ERR_INVALID;
        }
        if (timing)
            std::fprintf(stderr, "vb2_cohort_run: device %d: %d slots, stream over after %.1f ms\n", devices_[d], G_, 1e3 * (now_s() - t0));
        if (rc) {                                // device-level failure: concerns every sample
            std::lock_guard<std::mutex> lk(mu_);
            if (!rc_all_) {
                rc_all_ = rc;
                err_all_ = vb2::g_last_error;
            }
            stop_ = true;
        }
        cv_.notify_all();
    }

    void release_loop()
    {
        for (;;) {
            std::pair<vb2_ctx*, std::unique_ptr<vb2_flat>> item;
            {
                std::unique_lock<std::mutex> lk(rel_mu_);
                rel_cv_.wait(lk, [&] { return rel_stop_ || !rel_queue_.empty() || !done_queue_.empty(); });
                if (!done_queue_.empty()) {
                    Done d = std::move(done_queue_.front());
                    done_queue_.pop_front();
                    lk.unlock();
                    // a searched sample: its output files first (they need the flattened sample's counters)
                    const char* prefix = a_->output_prefixes ? a_->output_prefixes[d.s] : nullptr;
                    if (d.write && prefix && d.flat) {
                        int rw = vb2::write_ancestry(prefix, a_->base.num_pc, d.est.pc, d.est.pc2);
                        if (!rw) rw = vb2::write_selfsm(prefix, *d.flat, d.est, true);   // (cohort input is text pileups: #READS = NA)
                        if (rw) status_[d.s] = rw;
                    }
                    item.first = d.ctx;
                    item.second = std::move(d.flat);
                } else {
                    if (rel_queue_.empty()) return;
                    item = std::move(rel_queue_.front());
                    rel_queue_.pop_front();
                }
            }
            const double t0 = now_s();
            if (item.first) vb2_ctx_destroy(item.first);
            const double t1 = now_s();
            item.second.reset();
            rel_s_[0] += t1 - t0;
            rel_s_[1] += now_s() - t1;
        }
    }

    void device_loop(int d)
    {
        const bool timing = vb2::tunables().debug_timing != 0;
        for (int gi = d; gi < ngroup_; gi += ndev_) {
            const int g0 = group_begin_[gi], g1 = group_begin_[gi + 1];
            const double tg0 = now_s();
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] {
                    if (stop_) return true;
                    for (int s = g0; s < g1; ++s)
                        if (!slots_[s].ready) return false;
                    return true;
                });
                if (stop_) return;
            }
            const double tg_wait = now_s();
            std::vector<vb2_ctx*> ctxs;
            std::vector<int> who;
            for (int s = g0; s < g1; ++s) {
                status_[s] = slots_[s].rc;
                if (slots_[s].rc == VB2_OK && slots_[s].ctx) {
                    ctxs.push_back(slots_[s].ctx);
                    who.push_back(s);
                }
            }
            int rcb = VB2_OK;
            if (!ctxs.empty()) {
                const double t1 = now_s();
                std::vector<vb2_estimate> est(ctxs.size());
                vb2_batch* batch = nullptr;
                try {
                    rcb = vb2_batch_create(ctxs.data(), (int32_t)ctxs.size(), &batch);
                    if (!rcb) rcb = vb2_batch_optimize_llk(batch, &model_, 1, est.data());
                } catch (const std::exception& e) {
                    set_error(e.what());
                    rcb = VB2_ERR_INVALID;
                }
                if (batch) vb2_batch_destroy(batch);
                const double dt = (now_s() - t1) / (double)ctxs.size();
                if (rcb)                    // the group's search failed: its samples must not keep "ok" with a zeroed estimate
                    for (const int s : who) status_[s] = rcb;
                if (!rcb) {
                    for (size_t i = 0; i < who.size(); ++i) {
                        const int s = who[i];
                        out_[s].est = est[i];
                        out_[s].seconds_optimize = dt;
                        const char* prefix = a_->output_prefixes ? a_->output_prefixes[s] : nullptr;
                        if (prefix) {
                            int rw = vb2::write_ancestry(prefix, a_->base.num_pc, est[i].pc, est[i].pc2);
                            if (!rw) rw = vb2::write_selfsm(prefix, *slots_[s].flat, est[i], true);   // (cohort input is text pileups: #READS = NA)
                            if (rw) status_[s] = rw;
                        }
                    }
                }
            }
            const double tg_opt = now_s();
            {   // freeing device and pinned memory synchronises with the device and takes milliseconds
                // per context: hand the group to the releaser thread and move on
                std::lock_guard<std::mutex> lk(rel_mu_);
                for (int s = g0; s < g1; ++s) {
                    rel_queue_.emplace_back(slots_[s].ctx, std::move(slots_[s].flat));
                    slots_[s].ctx = nullptr;
                }
            }
            rel_cv_.notify_one();
            if (timing)
                std::fprintf(stderr, "vb2_cohort_run: device %d group %d (samples %d-%d): waited %.1f ms for the "
                                     "readers, search + outputs %.1f ms\n", devices_[d], gi, g0, g1 - 1,
                             1e3 * (tg_wait - tg0), 1e3 * (tg_opt - tg_wait));
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++cnt_[d];
                if (rcb) {                       // device-level failure: concerns every sample
                    if (!rc_all_) {
                        rc_all_ = rcb;
                        err_all_ = vb2::g_last_error;
                    }
                    stop_ = true;
                }
            }
            cv_.notify_all();
            if (rcb) return;
        }
    }

public:
    const std::string& error() const { return err_all_; }
    int num_reader() const { return T_; }
};

}  // namespace

extern "C" int vb2_cohort_run(const vb2_cohort_args* a, vb2_run_result* out, int32_t* status)
{
    if (!a || !out || !status || a->num_sample < 1 || !a->pileup_paths || !a->base.ud_path ||
        !a->base.mean_path || !a->base.bed_path || a->base.num_pc < 1 || a->base.num_pc > VB2_MAX_PC ||
        a->base.num_device < 0 || a->base.num_device > 64) {
        set_error("vb2_cohort_run: invalid argument");
        return VB2_ERR_INVALID;
    }
    // (Tunables::cohort_dup_devices: a test switch -- the pipelines of a several-device run, their group
    // schedule and the readers' per-device look-ahead exercised on a machine with one GPU)
    const bool dup_ok = vb2::tunables().cohort_dup_devices != 0;
    for (int i = 0; i < a->base.num_device && !dup_ok; ++i)
        for (int j = 0; j < i; ++j)
            if (a->base.devices && a->base.devices[i] == a->base.devices[j]) {
                // (two lock-step pipelines on one device would each launch device-filling grids)
                set_error("vb2_cohort_run: a device is listed twice");
                return VB2_ERR_INVALID;
            }
    try {
        CohortRunner runner(a, out, status);
        const int rc = runner.run();
        if (rc && !runner.error().empty()) set_error(runner.error());
        return rc;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}
