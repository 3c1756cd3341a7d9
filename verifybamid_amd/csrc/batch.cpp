// batch.cpp -- see batch.h.
#include "batch.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>

#include "estimator.h"
#include "lockstep.h"
#include "stream_search.h"
#include "tunables.h"

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
    for (auto& h : half_) h.reset();                            // (they launch on lane_streams_)
    for (hipStream_t& st : lane_streams_)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
            st = nullptr;
        }
    if (stream_) (void)hipStreamSynchronize(stream_);          // nothing of this batch is in flight any more
    if (d_slab_ && !recycle_device_slab(d_slab_, d_slab_bytes_, device)) (void)hipFree(d_slab_);
    if (h_slab_ && !recycle_pinned_slab(h_slab_, h_slab_bytes_, device)) (void)hipHostFree(h_slab_);
    if (stream_ && own_stream_ && !recycle_stream(stream_, device)) (void)hipStreamDestroy(stream_);
}

int Batch::create(vb2_ctx* const* ctxs, int num_sample, Batch** out)
{
    *out = nullptr;
    if (!ctxs || num_sample < 1) {
        set_error("vb2_batch_create: invalid argument");
        return VB2_ERR_INVALID;
    }
    std::vector<Context*> list;
    for (int s = 0; s < num_sample; ++s) {
        if (!ctxs[s] || !ctxs[s]->impl) {
            set_error("vb2_batch_create: null context");
            return VB2_ERR_INVALID;
        }
        list.push_back(ctxs[s]->impl);
    }
    return create(list, out);
}

// Tunables::cohort_w16 = 0 (a test aid): cohort steps stream the 32-bit run lists
bool cohort_w16_enabled() { return tunables().cohort_w16 != 0; }

// Waves per workgroup of a cohort step (round 4, third part): EIGHT, and twice as many workgroups per sample, instead of
// sixteen.  A search runs its samples as two lanes whose steps take turns; with 1 024-thread workgroups a lane's launch fills
// the CUs' wave slots by itself and the other lane's launch overlaps only with its tail, with 512-thread workgroups two
// workgroups of DIFFERENT launches share a CU and one lane's table builds, barriers and hand-off run under the other's
// streaming: search only 1.356 -> 1.243 ms per sample (737 -> 801 samples/s) on the same box; a synchronous step alone
// takes what it took (75.5 / 105 us).  Tunables::cohort_bw = 16 restores the big workgroups (4 and 8 are the other values).
int cohort_waves()
{
    const int k = tunables().cohort_bw;
    return (k == 4 || k == 8 || k == 16) ? k : 8;
}

void Batch::geometry(int num_cu, int num_sample, int max_mt, int num_pc, int bps_in, int* bps, int* block_waves)
{
    // ~one workgroup per CU in total; the waves of a workgroup: one per tile it owns, 1024 threads at most, and enough
    // of them to stage a step's parameter rows
    const int max_bw = cohort_waves();
    const int wg_total = num_cu * kMaxBlockWaves / max_bw;
    *bps = bps_in > 0 ? bps_in * (kMaxBlockWaves / max_bw) : std::max(1, wg_total / std::max(1, num_sample));
    const int tiles_per_block = (max_mt + *bps - 1) / *bps;
    const int min_bw = std::max(4, (kSlot * (2 * num_pc + 1) + 127) / 128);
    *block_waves = std::max(min_bw, std::min(max_bw, tiles_per_block));
}

int Batch::regroup_bps(int num_cu, int active)
{
    int b = 1;
    while (2 * b * active <= num_cu) b *= 2;
    return b;
}

namespace {
constexpr int kSplitFrom = 16;               // a cohort of this many samples is searched as two half-cohorts taking turns
}

int Batch::prepare_for_cohort(Context* c, int group)
{
    if (group < 1 || c->L.num_mt == 0) return VB2_OK;
    int lanes[2] = {group, 0};
    if (group >= kSplitFrom) { lanes[0] = group / 2; lanes[1] = group - group / 2; }
    Schedule sc[kShapes];
    int first_bps = 0;
    for (int l = 0; l < 2; ++l) {
        if (lanes[l] <= 0) continue;
        int bps, bw;
        geometry(c->L.num_cu, lanes[l], c->L.num_mt, c->num_pc, 0, &bps, &bw);
        if (bps == first_bps) continue;
        if (!first_bps) first_bps = bps;
        if (const int rc = c->cohort_schedules(bps, bw, sc)) return rc;
    }
    // the regrouped batches: powers of two above the lane's own (those that take the work queue need no schedule)
    for (int b = 1; b <= c->L.num_cu; b *= 2) {
        int bps, bw;
        geometry(c->L.num_cu, 1, c->L.num_mt, c->num_pc, b, &bps, &bw);      // (bps: b times the workgroups-per-CU factor)
        if (bps <= first_bps) continue;                                       // like with like
        if (eval_takes_the_queue(c->L, bps, bw, 1)) break;
        if (const int rc = c->cohort_schedules(bps, bw, sc)) return rc;
    }
    return VB2_OK;
}

int Batch::create(const std::vector<Context*>& ctxs, Batch** out, int bps_in)
{
    *out = nullptr;
    const int num_sample = (int)ctxs.size();
    if (num_sample < 1) {
        set_error("vb2_batch_create: invalid argument");
        return VB2_ERR_INVALID;
    }
    std::unique_ptr<Batch> b(new Batch());
    b->num_sample = num_sample;
    int max_mt = 0, num_cu = 256;
    std::vector<DeviceLayout> layouts(num_sample);
    for (int s = 0; s < num_sample; ++s) {
        Context* c = ctxs[s];
        if (s == 0) {
            b->device = c->device;
            b->num_pc = c->num_pc;
            num_cu = c->L.num_cu;
        } else if (c->device != b->device || c->num_pc != b->num_pc) {
            set_error("vb2_batch_create: contexts must share the device and --NumPC");
            return VB2_ERR_INVALID;
        }
        b->ctx_.push_back(c);
        // cohort steps stream every sample's run lists from HBM once per step: the 16-bit copy halves
        // those bytes (VB2_COHORT_W16=0: the 32-bit lists, A/B)
        const bool w16_on = cohort_w16_enabled();
        if (w16_on && c->L.num_mt > 0)
            if (const int rc16 = c->ensure_codes16()) return rc16;
        if (!w16_on || (c->L.num_mt > 0 && !c->L.codes16)) b->w16_ = false;
        layouts[s] = c->L;
        layouts[s].stamps = nullptr;
        if (c->L.row_bytes != kRowBytesWide) b->wide_rows_ = false;
        max_mt = std::max(max_mt, c->L.num_mt);
    }
    VB2_HIP(hipSetDevice(b->device));

    int bps = 1;
    geometry(num_cu, num_sample, max_mt, b->num_pc, bps_in, &bps, &b->block_waves_);
    b->bps_ = bps;
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

    b->layouts_ = std::move(layouts);
    *out = b.release();
    return VB2_OK;
}

int Batch::create_slots(int capacity, int device, int num_pc, int num_cu, Batch** out)
{
    *out = nullptr;
    if (capacity < 1 || capacity > 64 || num_pc < 1 || num_pc > VB2_MAX_PC) {
        set_error("streaming batch: invalid argument");
        return VB2_ERR_INVALID;
    }
    std::unique_ptr<Batch> b(new Batch());
    b->num_sample = capacity;
    b->device = device;
    b->num_pc = num_pc;
    b->slots_ = true;
    b->strict_shapes = true;
    b->ctx_.assign(capacity, nullptr);
    DeviceLayout none;
    std::memset(&none, 0, sizeof(none));
    none.num_pc = num_pc;
    none.row_bytes = kRowBytesWide;
    none.num_cu = num_cu;
    b->layouts_.assign(capacity, none);
    b->bps_ = std::max(1, num_cu * (kMaxBlockWaves / cohort_waves()) / capacity);
    b->block_waves_ = cohort_waves();         // whatever the samples: a sample's sums must not depend on its neighbours
    *out = b.release();
    return VB2_OK;
}

int Batch::set_slot(int i, Context* c)
{
    if (!slots_ || i < 0 || i >= num_sample || in_flight_) {
        set_error("streaming batch: set_slot out of place");
        return VB2_ERR_INVALID;
    }
    DeviceLayout lay = layouts_[i];
    Schedule sc[kShapes];
    for (int sh = 0; sh < kShapes; ++sh) sc[sh] = Schedule{nullptr, nullptr};
    if (c) {
        if (c->device != device || c->num_pc != num_pc) {
            set_error("streaming batch: contexts must share the device and --NumPC");
            return VB2_ERR_INVALID;
        }
        if (cohort_w16_enabled() && c->L.num_mt > 0)
            if (const int rc16 = c->ensure_codes16()) return rc16;
        lay = c->L;
        lay.stamps = nullptr;
        static const int kShapeNp[kShapes] = {4, 8, 1, 2};
        size_t need[kShapes];
        for (int sh = 0; sh < kShapes; ++sh) need[sh] = std::max(shmem_[sh], eval_shmem_np(lay, kShapeNp[sh], bps_, block_waves_, 1));
        if (need[1] > (size_t)kLdsLimitBytes) {
            set_error("streaming batch: per-workgroup LDS need exceeds 160 KiB");
            return VB2_ERR_INVALID;
        }
        for (int sh = 0; sh < kShapes; ++sh) shmem_[sh] = need[sh];
        if (const int rc = c->cohort_schedules(bps_, block_waves_, sc, false)) return rc;
    } else {
        std::memset(&lay, 0, sizeof(lay));
        lay.num_pc = num_pc;
        lay.row_bytes = kRowBytesWide;
        lay.num_cu = layouts_[i].num_cu;
    }
    ctx_[i] = c;
    layouts_[i] = lay;
    w16_ = cohort_w16_enabled();
    wide_rows_ = true;
    for (int s = 0; s < num_sample; ++s) {
        if (!ctx_[s]) continue;
        if (ctx_[s]->L.num_mt > 0 && !ctx_[s]->L.codes16) w16_ = false;
        if (ctx_[s]->L.row_bytes != kRowBytesWide) wide_rows_ = false;
    }
    if (!ready_ || !c) return VB2_OK;           // (ensure_resources uploads everything at the first step; an empty slot's
                                                //  workgroups leave before they read its layout)
    VB2_HIP(hipSetDevice(device));
    // slot i's layout and schedules through the slot's own stretch of pinned staging: stream-ordered before the next step;
    // the stretch is not written again before that step is over (the slot cannot change hands sooner)
    char* st = h_slot_stage_ + (size_t)i * kSlotStageBytes;
    std::memcpy(st, &lay, sizeof(lay));
    std::memcpy(st + sizeof(DeviceLayout), sc, sizeof(sc));
    VB2_HIP(hipMemcpyAsync(d_layouts_ + i, st, sizeof(DeviceLayout), hipMemcpyHostToDevice, stream_));
    for (int sh = 0; sh < kShapes; ++sh)
        VB2_HIP(hipMemcpyAsync(const_cast<Schedule*>(d_sched_arr_) + (size_t)sh * num_sample + i,
                               st + sizeof(DeviceLayout) + sh * sizeof(Schedule), sizeof(Schedule), hipMemcpyHostToDevice, stream_));
    return VB2_OK;
}

int Batch::ensure_resources()
{
    if (ready_) return VB2_OK;
    VB2_HIP(hipSetDevice(device));
    const int k = num_pc, bps = bps_;
    const size_t S = (size_t)num_sample, stride = 2 * (size_t)k + 1;
    // static schedules (llk_kernels.h) of every sample for the wave shapes of a step: the contexts keep them
    // (Context::cohort_schedules -- prepared by the reader threads in a vb2_cohort_run)
    std::vector<Schedule> arr(kShapes * S, Schedule{nullptr, nullptr});
    bool sched_ok = false;
    for (int s = 0; s < num_sample; ++s) {
        Schedule sc[kShapes];
        if (!ctx_[s]) continue;                              // (an empty slot: create_slots)
        if (const int rc = ctx_[s]->cohort_schedules(bps, block_waves_, sc, !slots_)) return rc;
        for (int sh = 0; sh < kShapes; ++sh) {
            arr[sh * S + s] = sc[sh];
            sched_ok |= sc[sh].off != nullptr;
        }
    }
    // device slab
    size_t dtot = 0;
    auto carve = [&](size_t bytes) {
        const size_t off = (dtot + 255) & ~(size_t)255;
        dtot = off + bytes;
        return off;
    };
    const size_t o_lay = carve(sizeof(DeviceLayout) * S);
    const size_t o_arr = carve(kShapes * S * sizeof(Schedule));
    const size_t o_zero = carve(0);
    const size_t o_part = carve(sizeof(double) * S * (kSlot + 1) * bps);
    const size_t o_tick = carve(sizeof(unsigned int) * S);
    const size_t o_done = carve(sizeof(unsigned int));
    dtot = (dtot + 255) & ~(size_t)255;
    d_slab_ = cached_device_slab(dtot, device, &d_slab_bytes_);
    if (!d_slab_) {
        VB2_HIP(hipMalloc(&d_slab_, dtot));
        d_slab_bytes_ = dtot;
    }
    char* dbase = static_cast<char*>(d_slab_);
    if (!stream_) {
        stream_ = cached_stream(device);
        if (!stream_) VB2_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
    // host image of the read-only part, uploaded on the batch's stream (the launches follow on it)
    std::vector<char> image(o_zero, 0);
    std::memcpy(image.data() + o_lay, layouts_.data(), sizeof(DeviceLayout) * S);
    std::memcpy(image.data() + o_arr, arr.data(), kShapes * S * sizeof(Schedule));
    VB2_HIP(hipMemcpyAsync(dbase, image.data(), image.size(), hipMemcpyHostToDevice, stream_));
    VB2_HIP(hipMemsetAsync(dbase + o_zero, 0, dtot - o_zero, stream_));
    VB2_HIP(hipStreamSynchronize(stream_));                       // (image is a pageable temporary)
    d_layouts_ = reinterpret_cast<DeviceLayout*>(dbase + o_lay);
    d_sched_arr_ = reinterpret_cast<const Schedule*>(dbase + o_arr);
    for (int sh = 0; sh < kShapes; ++sh)
        d_scheds_[sh] = (sched_ok || slots_) ? d_sched_arr_ + sh * S : nullptr;
    d_partials_ = reinterpret_cast<double*>(dbase + o_part);
    d_tickets_ = reinterpret_cast<unsigned int*>(dbase + o_tick);
    d_batch_done_ = reinterpret_cast<unsigned int*>(dbase + o_done);
    // pinned, device-mapped slab
    size_t htot = 0;
    auto hcarve = [&](size_t bytes) {
        const size_t off = (htot + 127) & ~(size_t)127;
        htot = off + bytes;
        return off;
    };
    const size_t p_pts = hcarve(sizeof(double) * S * kSlot * stride);
    const size_t p_out = hcarve(sizeof(double) * S * kSlot);
    const size_t p_nv = hcarve(sizeof(int) * S);
    const size_t p_done = hcarve(sizeof(unsigned long long) * 8);
    const size_t p_stage = hcarve(slots_ ? S * kSlotStageBytes : 0);
    h_slab_ = cached_pinned_slab(htot, device, &h_slab_bytes_);
    if (!h_slab_) {
        VB2_HIP(hipHostMalloc(&h_slab_, htot, hipHostMallocMapped));
        h_slab_bytes_ = htot;
    }
    std::memset(h_slab_, 0, htot);
    char* hbase = static_cast<char*>(h_slab_);
    char* hdev = nullptr;
    VB2_HIP(hipHostGetDevicePointer((void**)&hdev, h_slab_, 0));
    h_points_ = reinterpret_cast<double*>(hbase + p_pts);
    d_points_ = reinterpret_cast<double*>(hdev + p_pts);
    h_out_ = reinterpret_cast<double*>(hbase + p_out);
    d_out_ = reinterpret_cast<double*>(hdev + p_out);
    h_nv_ = reinterpret_cast<int*>(hbase + p_nv);
    d_nv_ = reinterpret_cast<int*>(hdev + p_nv);
    h_done_ = reinterpret_cast<unsigned long long*>(hbase + p_done);
    d_done_ = reinterpret_cast<unsigned long long*>(hdev + p_done);
    h_slot_stage_ = hbase + p_stage;
    seq_ = 0;
    ready_ = true;
    return VB2_OK;
}

int Batch::eval(const int32_t* num_point, const double* pc1, const double* pc2, const double* alpha,
                double* llk_out)
{
    if (const int rc = eval_begin(num_point, pc1, pc2, alpha, llk_out)) return rc;
    return eval_end();
}

int Batch::eval_begin(const int32_t* num_point, const double* pc1, const double* pc2, const double* alpha,
                      double* llk_out)
{
    if (in_flight_) {
        set_error("vb2_batch_eval: a step is already in flight");
        return VB2_ERR_INVALID;
    }
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
    if (const int rc = ensure_resources()) return rc;
    {
        // Samples of both layouts in one step (a probability-domain context next to one that could not take that layout: deep
        // markers, quality 0): a launch runs ONE kind of kernel, so the probability-domain samples are evaluated first, on their
        // own, and the others go on as this step.  A sample's values do not depend on which launch carries it.
        bool any_pd = false, any_log = false;
        for (int s = 0; s < num_sample; ++s)
            if (num_point[s] > 0 && ctx_[s] && ctx_[s]->L.num_mt > 0) (ctx_[s]->L.pd ? any_pd : any_log) = true;
        if (any_pd && any_log) {
            std::vector<int32_t> np((size_t)num_sample);
            const bool was_split = in_split_;
            for (int s = 0; s < num_sample; ++s) np[s] = (ctx_[s] && ctx_[s]->L.pd) ? num_point[s] : 0;
            if (const int rc = eval(np.data(), pc1, pc2, alpha, llk_out)) return rc;
            for (int s = 0; s < num_sample; ++s) np[s] = (ctx_[s] && ctx_[s]->L.pd) ? 0 : num_point[s];
            const int rc = eval_begin(np.data(), pc1, pc2, alpha, llk_out);
            in_split_ = was_split;
            return rc;
        }
    }
    if (strict_shapes && !in_split_) {
        // A step's wave shape follows from the LARGEST request in it, and under the static deal the shape decides which tiles a
        // wave multiplies together -- so a sample's sums would move in their last bits with what its neighbours happen to ask
        // for (a new sample's initial simplex, a shrink).  Strict: the requests of 1-2, of 3-4 and of 5-8 points are evaluated
        // as separate launches (the one- and two-point shapes deal and multiply alike: bit-identical, tested), so a sample's
        // values depend on its own requests only.  Mixed steps are rare (a sample starts or shrinks in 1-2 % of them): all
        // classes but the most populous are evaluated synchronously first.
        // (a 5-8 point request of a sample with narrow table rows -- more than kMaxWideCodes codes -- is evaluated as two
        // 4-point launches, below: a class of its own, or one such neighbour would change everybody's wave shape: ADVICE r4)
        auto cls = [&](int s) {
            const int n = num_point[s];
            if (n <= 0) return -1;
            if (n <= 2) return 0;
            if (n <= 4) return 1;
            return (ctx_[s] && ctx_[s]->L.row_bytes != kRowBytesWide) ? 3 : 2;
        };
        int pop[4] = {0, 0, 0, 0};
        for (int s = 0; s < num_sample; ++s)
            if (num_point[s] > 0) ++pop[cls(s)];
        if ((pop[0] > 0) + (pop[1] > 0) + (pop[2] > 0) + (pop[3] > 0) > 1) {
            int keep = 0;
            for (int c = 1; c < 4; ++c)
                if (pop[c] > pop[keep]) keep = c;
            std::vector<int32_t> np((size_t)num_sample);
            in_split_ = true;
            int rc = VB2_OK;
            for (int c = 0; c < 4 && !rc; ++c) {
                if (c == keep || pop[c] == 0) continue;
                for (int s = 0; s < num_sample; ++s) np[s] = cls(s) == c ? num_point[s] : 0;
                rc = eval(np.data(), pc1, pc2, alpha, llk_out);
            }
            if (!rc) {
                for (int s = 0; s < num_sample; ++s) np[s] = cls(s) == keep ? num_point[s] : 0;
                rc = eval_begin(np.data(), pc1, pc2, alpha, llk_out);
            }
            in_split_ = false;
            return rc;
        }
    }
    bool narrow_in_step = false;               // (of the samples that take part in THIS step)
    for (int s = 0; s < num_sample && !wide_rows_; ++s)
        narrow_in_step |= num_point[s] > 0 && ctx_[s] && ctx_[s]->L.row_bytes != kRowBytesWide;
    if (max_n > 4 && narrow_in_step) {
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
    const bool small = max_n <= 2;
    const int NP = max_n > 4 ? 8 : small ? max_n : 4, shape = max_n > 4 ? 1 : small ? (max_n == 1 ? 2 : 3) : 0;
    for (int s = 0; s < num_sample; ++s) {
        int n = num_point[s];
        // a sample without active markers has LLK 0 for every point: answer it here
        if (n > 0 && (!ctx_[s] || ctx_[s]->L.num_mt == 0)) {
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
    MultiLaunch& ml = ml_;
    ml = MultiLaunch{};
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
    ml.force_ticket = false;
    ml.w16 = w16_;
    for (int s2 = 0; s2 < num_sample; ++s2)
        if (h_nv_[s2] > 0 && ctx_[s2] && ctx_[s2]->L.pd) ml.pd = true;      // (the step's samples are of one layout: see the top)
    // (probability domain: the short copy is the 8-bit step lists -- they pay where a step is bound by what it streams: 32 C3
    // samples x 1 point 62.4 -> 58.0 us, x 2 points 77.4 -> 76.0, x 4 points 124.6 -> 126.3: the one- and two-point shapes only)
    if (ml.pd && NP > 2) ml.w16 = false;
    {   // (the pipelined item loop is compiled for the static deal: every sample of the cohort must run it)
        bool st = true;
        for (int s2 = 0; s2 < num_sample && st; ++s2)
            st = !ctx_[s2] || ctx_[s2]->L.num_mt == 0 || !eval_takes_the_queue(ctx_[s2]->L, bps_, block_waves_, 1);
        ml.all_static = st;
    }
    {   // (the cohort kernels compiled for --NumPC 2 / 4 without a known-AF column)
        bool plain = num_pc == 2 || num_pc == 4;
        for (int s2 = 0; s2 < num_sample && plain; ++s2) plain = !ctx_[s2] || ctx_[s2]->L.known_af == nullptr;
        ml.ksel = plain ? num_pc : 0;
    }
    // counts and rows as kernel arguments when they fit (saves every workgroup two trips to mapped host memory)
    ml.inl.count = 0;
    if (num_sample <= kMultiInlineSamples && (size_t)num_sample * NP * stride <= (size_t)kMultiInlineDoubles) {
        ml.inl.count = num_sample * NP * stride;
        for (int s2 = 0; s2 < num_sample; ++s2) ml.inl.nv[s2] = (unsigned char)h_nv_[s2];
        std::memcpy(ml.inl.v, h_points_, sizeof(double) * (size_t)ml.inl.count);
    }
    // (results come back as relaxed stores behind a relaxed flag: NaN first, so that a store still on its way when the
    // flag is seen is noticed -- Context::eval_host, settle_results)
    for (int s = 0; s < num_sample; ++s)
        for (int j = 0; j < h_nv_[s]; ++j) h_out_[(size_t)s * NP + j] = std::numeric_limits<double>::quiet_NaN();
    VB2_HIP(launch_llk_eval_multi(ml, stream_));
    ++num_launch;
    in_flight_ = true;
    flight_np_ = NP;
    flight_out_ = llk_out;
    return VB2_OK;
}

int Batch::eval_end()
{
    if (!in_flight_) return VB2_OK;                      // (nothing was launched: answered in eval_begin)
    in_flight_ = false;
    VB2_HIP(hipSetDevice(device));
    MultiLaunch& ml = ml_;
    const int NP = flight_np_;
    // (a NaN is the tagged hand-off's "a workgroup never reported" marker: redo the step once with
    // the arrival-ticket hand-off, see Context::eval_host -- NaN must not reach the optimisers)
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt > 0) {
            ml.force_ticket = true;
            ml.done_seq = ++seq_;
            for (int s = 0; s < num_sample; ++s)
                for (int j = 0; j < h_nv_[s]; ++j) h_out_[(size_t)s * NP + j] = std::numeric_limits<double>::quiet_NaN();
            VB2_HIP(launch_llk_eval_multi(ml, stream_));
            ++num_launch;
        }
        bool seen = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(h_done_, __ATOMIC_ACQUIRE) == ml.done_seq) { seen = true; break; }
            if ((spins & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
            __builtin_ia32_pause();
        }
        if (!seen) VB2_HIP(hipStreamSynchronize(stream_));
        bool any_nan = true;
        const auto ts = std::chrono::steady_clock::now();
        while (any_nan) {                         // (a store still on its way: a few microseconds; then the kernels' own marker)
            any_nan = false;
            for (int s = 0; s < num_sample && !any_nan; ++s)
                for (int j = 0; j < h_nv_[s]; ++j)
                    any_nan |= std::isnan(reinterpret_cast<volatile double*>(h_out_)[(size_t)s * NP + j]);
            if (std::chrono::steady_clock::now() - ts > std::chrono::microseconds(50)) break;
        }
        if (!any_nan) break;
    }
    for (int s = 0; s < num_sample; ++s)
        for (int j = 0; j < h_nv_[s]; ++j) flight_out_[(size_t)s * kSlot + j] = h_out_[(size_t)s * NP + j];
    return VB2_OK;
}

// ---------------------------------------------------------------------------
// lock-step search: every sample runs the ordinary Estimator (OptimizeLLK with its six models,
// the reference-exact simplex) as a fiber of the calling thread (lockstep.h); the parked
// requests of a step leave as ONE launch of the multi-sample kernel.
//
// A cohort of kSplitFrom samples or more is searched as TWO half-cohorts taking turns: while one
// half's step is on the device, the other half's fibers run on the host (64 fiber switches and
// simplex updates are ~60 us, a quarter of the step they would otherwise be added to), and the
// next launch is already queued behind the running one, so the device never waits for the host.
// ---------------------------------------------------------------------------
namespace {

struct Lane {
    Batch* batch = nullptr;                 // the lane's samples, all of them
    std::unique_ptr<FiberGang> gang;
    int base = 0, count = 0;                // samples [base, base + count) of the whole cohort
    std::vector<int32_t> npts;
    std::vector<double> pc1, pc2, alpha, llk;
    bool flying = false;
    std::chrono::steady_clock::time_point t_launch;
    // the samples still searching, regrouped (see Batch::optimize): slot j of `cur` is fiber slot_fiber[j]
    Batch* cur = nullptr;
    std::unique_ptr<Batch> regrouped;
    std::vector<int> slot_fiber;
};

}  // namespace

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
    // decisions, same trajectory either way.  Tunables::cohort_speculate = 1|2|4 forces one.
    const Tunables& tn = tunables();
    constexpr int kPairFrom = 8;
    speculate_ = num_sample < kPairFrom ? 4 : 2;
    if (tn.cohort_speculate > 0) speculate_ = tn.cohort_speculate;
    bool split = num_sample >= kSplitFrom;
    if (tn.cohort_split >= 0) split = tn.cohort_split != 0 && num_sample >= 2;

    const int S = num_sample, k = num_pc;
    Lane lanes[kMaxLanes];
    int nlane = 1;
    if (split) {
        nlane = std::max(2, std::min(std::min(kMaxLanes, tn.cohort_lanes), S));
        if (tn.cohort_own_queues && !lane_streams_[0]) {
            VB2_HIP(hipSetDevice(device));
            for (int h = 0; h < nlane; ++h)
                if (hipStreamCreateWithFlags(&lane_streams_[h], hipStreamNonBlocking) != hipSuccess) {
                    (void)hipGetLastError();
                    lane_streams_[h] = nullptr;
                }
        }
        for (int h = 0; h < nlane; ++h) {
            const int lo = (int)((long)S * h / nlane), hi = (int)((long)S * (h + 1) / nlane);
            if (!half_[h] || half_[h]->num_sample != hi - lo) {
                std::vector<Context*> part(ctx_.begin() + lo, ctx_.begin() + hi);
                Batch* hb = nullptr;
                if (const int rc = Batch::create(part, &hb)) return rc;
                if (lane_streams_[h]) hb->borrow_stream(lane_streams_[h]);
                half_[h].reset(hb);
            }
            lanes[h].batch = half_[h].get(); lanes[h].base = lo; lanes[h].count = hi - lo;
        }
    } else {
        lanes[0].batch = this; lanes[0].base = 0; lanes[0].count = S;
    }
    // Samples finish at different iterations (C3-shaped cohort of 32: the shortest search is over after ~60 % of the
    // longest's steps), and a step gives every sample of its batch 256 / num_sample workgroups whether it still takes part
    // or not: in round 3 a third of all steps served one or two samples on an eighth of the device.  So when half of a
    // lane's samples have finished, the rest are regrouped into a batch of their own -- twice the workgroups per sample; the
    // sums move in their last bits, as they do between group sizes (the static deal multiplies a wave's items in the wave);
    // what regroups when follows from the samples' own trajectories, so a run stays reproducible -- and again at a quarter,
    // an eighth ...
    // Tunables::cohort_regroup = 0: the fixed batch to the end (A/B).
    const bool regroup = tn.cohort_regroup != 0;
    std::vector<int> rcs(S, 0);
    for (int l = 0; l < nlane; ++l) {
        Lane& L = lanes[l];
        L.cur = L.batch;
        L.slot_fiber.resize(L.count);
        for (int i = 0; i < L.count; ++i) L.slot_fiber[i] = i;
        const size_t n = (size_t)L.count;
        L.gang.reset(new FiberGang(L.count, kSlot));
        L.npts.assign(n, 0);
        L.pc1.resize(n * kSlot * k); L.pc2.resize(n * kSlot * k); L.alpha.resize(n * kSlot); L.llk.resize(n * kSlot);
    }
    auto body_of = [&](Lane& L) {
        return [&, this](int i) {
            const int s = L.base + i;
            try {
                const vb2_model& m = models[num_model == 1 ? 0 : s];
                Estimator est(num_pc, FiberGang::eval_cb, L.gang->user(i));
                apply_model(est, m, ctx_[s]->L.known_af != nullptr);
                est.speculate = speculate_;
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
    };
    const bool dbg = tn.debug_lockstep != 0;
    double dbg_regroup_s = 0, dbg_wall_by_slots[65] = {0}; long dbg_steps_by_slots[65] = {0};
    long dbg_steps = 0, dbg_active = 0, dbg_points = 0, dbg_regroups = 0, dbg_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int error = 0;
    auto fail = [&](int rc) {
        if (!error) error = rc;
        for (int l = 0; l < nlane; ++l) lanes[l].gang->fail(rc);
    };
    // the parked requests of a lane -> one launch
    auto launch = [&](Lane& L) {
        if (error || !L.gang->pending()) return;
        std::vector<FiberGang::Request>& req = L.gang->requests();
        if (regroup) {
            int act = 0;
            for (int j : L.slot_fiber) act += req[j].n > 0;
            if (act >= 1 && 2 * act <= (int)L.slot_fiber.size()) {
                std::vector<int> keep;
                std::vector<Context*> part;
                for (int j : L.slot_fiber)
                    if (req[j].n > 0) {
                        keep.push_back(j);
                        part.push_back(ctx_[L.base + j]);
                    }
                Batch* nb = nullptr;
                const auto tr0 = std::chrono::steady_clock::now();
                int rc = Batch::create(part, &nb, regroup_bps(ctx_[0]->L.num_cu, (int)part.size()));
                if (!rc && nlane > 1 && lane_streams_[&L - lanes]) nb->borrow_stream(lane_streams_[&L - lanes]);
                if (!rc) rc = nb->ensure_resources();
                if (rc) {
                    delete nb;
                    fail(rc);
                    return;
                }
                dbg_regroup_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
                L.regrouped.reset(nb);             // (the batch it replaces has no step in flight: its slabs go back to the caches)
                L.cur = nb;
                L.slot_fiber.swap(keep);
                ++dbg_regroups;
            }
        }
        const int nslot = (int)L.slot_fiber.size();
        for (int i = 0; i < nslot; ++i) {
            const FiberGang::Request& r = req[L.slot_fiber[i]];
            L.npts[i] = r.n;
            if (r.n <= 0) continue;
            std::memcpy(&L.pc1[(size_t)i * kSlot * k], r.p1, sizeof(double) * r.n * k);
            std::memcpy(&L.pc2[(size_t)i * kSlot * k], r.p2, sizeof(double) * r.n * k);
            std::memcpy(&L.alpha[(size_t)i * kSlot], r.a, sizeof(double) * r.n);
        }
        if (dbg) {
            int act = 0, pts = 0;
            for (int i = 0; i < nslot; ++i) { act += L.npts[i] > 0; pts += std::max(0, L.npts[i]); }
            ++dbg_steps; dbg_active += act; dbg_points += pts;
            dbg_hist[std::min(8, (8 * act + nslot - 1) / nslot)]++;
        }
        if (const int rc = L.cur->eval_begin(L.npts.data(), L.pc1.data(), L.pc2.data(), L.alpha.data(), L.llk.data())) {
            fail(rc);
            return;
        }
        L.flying = true;
        L.t_launch = std::chrono::steady_clock::now();
    };
    // wait for the lane's step, hand the values out, run its fibers up to their next requests
    auto land = [&](Lane& L) {
        if (L.flying) {
            L.flying = false;
            if (const int rc = L.cur->eval_end()) fail(rc);
            if (dbg) {
                const int ns = std::min(64, (int)L.slot_fiber.size());
                dbg_wall_by_slots[ns] += std::chrono::duration<double>(std::chrono::steady_clock::now() - L.t_launch).count();
                ++dbg_steps_by_slots[ns];
            }
            if (!error) {
                std::vector<FiberGang::Request>& req = L.gang->requests();
                for (int i = 0; i < (int)L.slot_fiber.size(); ++i)
                    if (L.npts[i] > 0)
                        std::memcpy(req[L.slot_fiber[i]].out, &L.llk[(size_t)i * kSlot], sizeof(double) * L.npts[i]);
            }
        }
        if (L.gang->pending()) L.gang->resume_parked();       // (after an error: the fibers unwind)
    };
    for (int l = 0; l < nlane; ++l)
        if (lanes[l].gang->start(k, body_of(lanes[l])) < 0) {
            // the lanes started so far have fibers parked in the middle of OptimizeLLK: tell them, and
            // run them until they have unwound (their Estimators and vectors live on the fiber stacks,
            // which the gang unmaps) before giving up
            for (int m = 0; m < l; ++m) {
                lanes[m].gang->fail(VB2_ERR_INVALID);
                while (lanes[m].gang->pending()) lanes[m].gang->resume_parked();
            }
            set_error("vb2_batch_optimize_llk: could not start the search fibers (mmap / getcontext failed)");
            return VB2_ERR_INVALID;
        }
    const auto t_loop0 = std::chrono::steady_clock::now();
    launch(lanes[0]);
    for (;;) {
        bool busy = false;
        for (int l = 0; l < nlane; ++l) {
            Lane& next = lanes[(l + 1) % nlane];
            if (nlane > 1 && !next.flying) launch(next);       // queued behind the step in flight
            Lane& cur = lanes[l];
            if (cur.flying || cur.gang->pending()) {
                land(cur);
                launch(cur);
            }
            busy |= cur.flying || cur.gang->pending();
        }
        if (!busy) break;
    }
    if (dbg) {
        std::fprintf(stderr, "lock-step loop %.2f ms, of which regrouping %.2f ms\n",
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count(), 1e3 * dbg_regroup_s);
        for (int n = 1; n <= 64; ++n)
            if (dbg_steps_by_slots[n])
                std::fprintf(stderr, "  batches of %d: %ld steps, %.1f us from launch to landing each\n", n, dbg_steps_by_slots[n],
                             1e6 * dbg_wall_by_slots[n] / dbg_steps_by_slots[n]);
    }
    if (dbg)
        std::fprintf(stderr, "lock-step search of %d samples in %d lane(s): %ld steps, %.1f active samples and %.1f points per step "
                             "(a lane holds %d, %ld regroupings); steps by eighths of the batch active: %ld %ld %ld %ld %ld %ld %ld %ld\n",
                     S, nlane, dbg_steps, (double)dbg_active / std::max(1L, dbg_steps), (double)dbg_points / std::max(1L, dbg_steps),
                     lanes[0].count, dbg_regroups, dbg_hist[1], dbg_hist[2], dbg_hist[3], dbg_hist[4], dbg_hist[5], dbg_hist[6], dbg_hist[7], dbg_hist[8]);
    num_regroup = dbg_regroups;
    if (error) return error;
    for (int s = 0; s < S; ++s)
        if (rcs[s]) return rcs[s];
    return VB2_OK;
}

// ---------------------------------------------------------------------------
// stream_search.h: slots that change hands
// ---------------------------------------------------------------------------
namespace {

struct StreamLane {
    std::unique_ptr<Batch> batch;
    std::unique_ptr<FiberGang> gang;
    int count = 0;
    std::vector<int> id;                    // [slot] the sample in it, or -1
    std::vector<int> rc;
    std::vector<vb2_estimate> est;
    std::vector<double> t0;
    std::vector<int32_t> npts;
    std::vector<double> pc1, pc2, alpha, llk;
    bool flying = false;
};

double wall_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void stream_lanes(int capacity, int* n0, int* n1)
{
    *n0 = capacity;
    *n1 = 0;
    if (capacity >= kSplitFrom) {
        *n0 = capacity / 2;
        *n1 = capacity - capacity / 2;
    }
}

}  // namespace

int prepare_for_stream(Context* c, int capacity)
{
    if (!c || c->L.num_mt == 0) return VB2_OK;
    int n[2];
    stream_lanes(capacity, &n[0], &n[1]);
    Schedule sc[Batch::kShapes];
    for (int l = 0; l < 2; ++l) {
        if (n[l] <= 0 || (l == 1 && n[1] == n[0])) continue;
        if (const int rc = c->cohort_schedules(std::max(1, c->L.num_cu * (kMaxBlockWaves / cohort_waves()) / n[l]), cohort_waves(), sc, false)) return rc;
    }
    return VB2_OK;
}

int stream_search(int device, int num_pc, int num_cu, int capacity, StreamSource& src, const hipStream_t* lane_streams)
{
    capacity = std::max(1, std::min(capacity, 128));
    const int k = num_pc;
    int speculate = capacity < 8 ? 4 : 2;                      // (Batch::optimize: kPairFrom)
    if (tunables().cohort_speculate > 0) speculate = tunables().cohort_speculate;
    StreamLane lanes[2];
    int cnt[2];
    stream_lanes(capacity, &cnt[0], &cnt[1]);
    const int nlane = cnt[1] > 0 ? 2 : 1;
    int error = 0;
    for (int l = 0; l < nlane; ++l) {
        StreamLane& L = lanes[l];
        L.count = cnt[l];
        Batch* b = nullptr;
        if (const int rc = Batch::create_slots(L.count, device, k, num_cu, &b)) return rc;
        L.batch.reset(b);
        if (lane_streams && lane_streams[l]) b->borrow_stream(lane_streams[l]);
        // A streamed sample starts whenever a slot falls free, alone.  The ten vertices of its fresh simplex as ONE request would be
        // a class of their own in that step (strict shapes: eval_begin evaluates the minority classes synchronously first): an
        // eight-point launch on that sample's sixteenth of the device, ~120 us during which the pipeline thread serves neither
        // lane.  In pieces of two they ride along with the other slots' steps.
        L.gang.reset(new FiberGang(L.count, std::max(1, std::min(kSlot, tunables().cohort_stream_points))));
        const size_t n = (size_t)L.count;
        L.id.assign(n, -1);
        L.rc.assign(n, 0);
        L.est.resize(n);
        L.t0.assign(n, 0.0);
        L.npts.assign(n, 0);
        L.pc1.resize(n * kSlot * k); L.pc2.resize(n * kSlot * k); L.alpha.resize(n * kSlot); L.llk.resize(n * kSlot);
    }
    for (int l = 0; l < nlane; ++l) {
        StreamLane& L = lanes[l];
        L.gang->open(k, [&L, &src, k, speculate](int i) {
            try {
                Estimator est(k, FiberGang::eval_cb, L.gang->user(i));
                apply_model(est, src.model(L.id[i]), L.batch->slot(i)->L.known_af != nullptr);
                est.speculate = speculate;
                L.rc[i] = est.OptimizeLLK();
                fill_estimate(est, &L.est[i]);
            } catch (const std::bad_alloc&) {
                L.rc[i] = VB2_ERR_NOMEM;
            } catch (const std::exception& e) {
                set_error(e.what());
                L.rc[i] = VB2_ERR_INVALID;
            } catch (...) {                      // nothing may unwind past the fiber's entry frame
                set_error("stream_search: unknown exception in a sample's search");
                L.rc[i] = VB2_ERR_INVALID;
            }
        });
    }
    bool ended = false;
    const bool dbg = tunables().debug_lockstep != 0;
    long dbg_steps = 0, dbg_active = 0, dbg_points = 0, dbg_samples = 0;
    double dbg_blocked = 0, dbg_refill = 0, dbg_wait = 0, dbg_resume = 0, dbg_launch = 0, dbg_retire = 0;
    const double dbg_t0 = wall_s();
    auto fail = [&](int rc) {
        if (!error) error = rc;
        for (int l = 0; l < nlane; ++l) lanes[l].gang->fail(rc);
    };
    auto retire = [&](StreamLane& L, int i) {
        const int rc = error ? error : L.rc[i];
        src.done(L.id[i], rc, L.est[i], wall_s() - L.t0[i]);
        L.id[i] = -1;
        (void)L.batch->set_slot(i, nullptr);
    };
    // free slots of a lane <- samples that are ready; block: wait for the first one (nothing is running anywhere)
    auto refill = [&](StreamLane& L, bool block) {
        if (error) return;
        for (int i = 0; i < L.count && !ended; ++i) {
            if (L.id[i] >= 0) continue;
            Context* c = nullptr;
            const double tb = dbg ? wall_s() : 0.0;
            const int id = src.next(block, &c);
            if (dbg && block) dbg_blocked += wall_s() - tb;
            block = false;
            if (id == StreamSource::kEnd) ended = true;
            if (id < 0) break;
            L.id[i] = id;
            L.rc[i] = 0;
            std::memset(&L.est[i], 0, sizeof(vb2_estimate));
            L.t0[i] = wall_s();
            ++dbg_samples;
            // (Tunables::cohort_fail_sample: a test's way to send one sample down this error path)
            const int rc_slot = id == tunables().cohort_fail_sample ? (set_error("streaming batch: sample refused (test hook)"), VB2_ERR_INVALID)
                                                                   : L.batch->set_slot(i, c);
            if (const int rc = rc_slot) {                       // this sample cannot be searched here: the others can
                L.rc[i] = rc;
                retire(L, i);
                --i;
                continue;
            }
            if (L.gang->spawn(i) < 0) {
                set_error("stream_search: could not start a search fiber (mmap / getcontext failed)");
                L.rc[i] = VB2_ERR_INVALID;
                retire(L, i);
                continue;
            }
            if (L.gang->idle(i)) {                              // over before its first request
                retire(L, i);
                --i;
            }
            if (dbg) dbg_refill += wall_s() - L.t0[i < 0 ? 0 : i];
        }
    };
    auto launch = [&](StreamLane& L) {
        if (error || !L.gang->pending()) return;
        std::vector<FiberGang::Request>& req = L.gang->requests();
        for (int i = 0; i < L.count; ++i) {
            const FiberGang::Request& r = req[i];
            L.npts[i] = (L.id[i] >= 0 && !L.gang->idle(i)) ? r.n : 0;
            if (L.npts[i] <= 0) continue;
            std::memcpy(&L.pc1[(size_t)i * kSlot * k], r.p1, sizeof(double) * r.n * k);
            std::memcpy(&L.pc2[(size_t)i * kSlot * k], r.p2, sizeof(double) * r.n * k);
            std::memcpy(&L.alpha[(size_t)i * kSlot], r.a, sizeof(double) * r.n);
        }
        if (dbg) {
            ++dbg_steps;
            for (int i = 0; i < L.count; ++i) { dbg_active += L.npts[i] > 0; dbg_points += std::max(0, L.npts[i]); }
        }
        const double tl = dbg ? wall_s() : 0.0;
        if (const int rc = L.batch->eval_begin(L.npts.data(), L.pc1.data(), L.pc2.data(), L.alpha.data(), L.llk.data())) {
            fail(rc);
            return;
        }
        if (dbg) dbg_launch += wall_s() - tl;
        L.flying = true;
    };
    auto land = [&](StreamLane& L) {
        if (L.flying) {
            L.flying = false;
            const double tw = dbg ? wall_s() : 0.0;
            const int rc_end = L.batch->eval_end();
            if (dbg) dbg_wait += wall_s() - tw;
            if (const int rc = rc_end) fail(rc);
            if (!error) {
                std::vector<FiberGang::Request>& req = L.gang->requests();
                for (int i = 0; i < L.count; ++i)
                    if (L.npts[i] > 0) std::memcpy(req[i].out, &L.llk[(size_t)i * kSlot], sizeof(double) * L.npts[i]);
            }
        }
        const double tr = dbg ? wall_s() : 0.0;
        if (L.gang->pending()) L.gang->resume_parked();       // (after an error: the fibers unwind)
        const double tr2 = dbg ? wall_s() : 0.0;
        for (int i = 0; i < L.count; ++i)
            if (L.id[i] >= 0 && L.gang->idle(i)) retire(L, i);
        if (dbg) { dbg_resume += tr2 - tr; dbg_retire += wall_s() - tr2; }
    };
    for (;;) {
        bool busy = false;
        for (int l = 0; l < nlane; ++l) {
            StreamLane& next = lanes[(l + 1) % nlane];
            if (nlane > 1 && !next.flying) {                   // queued behind the step in flight
                refill(next, false);
                launch(next);
            }
            StreamLane& cur = lanes[l];
            if (cur.flying || cur.gang->pending()) land(cur);
            refill(cur, false);
            launch(cur);
            busy |= cur.flying || cur.gang->pending();
        }
        if (busy) continue;
        if (error || ended) break;
        refill(lanes[0], true);                                 // nothing in flight, nothing ready: wait for a sample
        launch(lanes[0]);
        if (!lanes[0].flying && !lanes[0].gang->pending() && ended) break;
    }
    if (dbg)
        std::fprintf(stderr, "stream search: %ld samples through %d slots in %.1f ms: %ld steps, %.1f active samples and %.1f points "
                             "per step (a lane holds %d); waited %.1f ms with nothing to do, %.1f ms taking samples in; the pipeline thread per step: "
                             "%.1f us waiting for a step to land, %.1f us in the samples' optimisers, %.1f us retiring, %.1f us launching\n",
                     dbg_samples, capacity, 1e3 * (wall_s() - dbg_t0), dbg_steps, (double)dbg_active / std::max(1L, dbg_steps),
                     (double)dbg_points / std::max(1L, dbg_steps), lanes[0].count, 1e3 * dbg_blocked, 1e3 * dbg_refill,
                     1e6 * dbg_wait / std::max(1L, dbg_steps), 1e6 * dbg_resume / std::max(1L, dbg_steps),
                     1e6 * dbg_retire / std::max(1L, dbg_steps), 1e6 * dbg_launch / std::max(1L, dbg_steps));
    return error;
}

}  // namespace vb2
