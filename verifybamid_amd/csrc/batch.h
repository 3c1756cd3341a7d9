// batch.h -- a cohort of samples evaluated / optimised in lock-step on one GPU
// (BASELINE.json configs[4]): every Nelder-Mead step of every sample
// goes into ONE kernel launch, so the per-launch fixed costs (launch, per-alpha table,
// cross-workgroup reduction) are paid once per step for the whole cohort.
#ifndef VB2_BATCH_H_
#define VB2_BATCH_H_

#include <memory>
#include <vector>

#include "context.h"

namespace vb2 {

int cohort_waves();                   // waves per workgroup of a cohort step: 8 (VB2_COHORT_BW=16|8|4)
bool cohort_w16_enabled();            // cohort steps stream the 16-bit run lists (default; VB2_COHORT_W16=0 turns it off)

class Batch {
public:
    ~Batch();
    static int create(vb2_ctx* const* ctxs, int num_sample, Batch** out);
    static int create(const std::vector<Context*>& ctxs, Batch** out, int bps = 0);   // bps > 0: that many workgroups per sample
    // A batch of `capacity` slots whose samples come and go (the streaming cohort search, stream_search.h): created empty;
    // set_slot(i, c) puts context c -- or nothing (nullptr) -- into slot i, between two steps.  Every slot has
    // 2 x num_cu / capacity workgroups of cohort_waves() = 8 waves whatever the other slots hold: a sample's sums do not depend on its
    // neighbours, nor on when it arrived.
    static int create_slots(int capacity, int device, int num_pc, int num_cu, Batch** out);
    int set_slot(int i, Context* c);
    // the batch's launches go onto a stream it does not own (set before its first step)
    void borrow_stream(hipStream_t s) { stream_ = s; own_stream_ = false; }
    Context* slot(int i) const { return ctx_[i]; }
    // launch geometry of a batch of num_sample samples whose biggest has max_mt micro-tiles: workgroups per sample and waves
    // per workgroup (what create() uses -- and what a reader thread prepares a sample's schedules for: prepare_for_cohort)
    static void geometry(int num_cu, int num_sample, int max_mt, int num_pc, int bps_in, int* bps, int* block_waves);
    // the workgroups per sample of the batch a lane's remaining `active` samples are regrouped into (a power of two)
    static int regroup_bps(int num_cu, int active);
    // schedules of `c` for the lanes of a lock-step search of `group` samples (and for the regrouped ones that need any)
    static int prepare_for_cohort(Context* c, int group);
    // num_point[s] in [0, 8]; pc1/pc2: [S][8][k]; alpha, llk_out: [S][8]
    int eval(const int32_t* num_point, const double* pc1, const double* pc2, const double* alpha,
             double* llk_out);
    // the same in two halves: begin launches and returns, end waits and delivers (into the llk_out
    // given to begin).  One step may be in flight per Batch.
    int eval_begin(const int32_t* num_point, const double* pc1, const double* pc2, const double* alpha,
                   double* llk_out);
    int eval_end();
    // OptimizeLLK for every sample, searches advancing in lock-step; models: 1 or S entries
    int optimize(const vb2_model* models, int num_model, vb2_estimate* out);

    static constexpr int kShapes = 4;       // launch shapes of a step: <= 4, 8, 1, 2 points per sample
    int num_sample = 0, num_pc = 0, device = -1;
    int64_t num_launch = 0;
    bool strict_shapes = false;             // requests of 1-2, 3-4 and 5-8 points leave as separate launches (see eval_begin)
    int64_t num_regroup = 0;                // batches the last optimize() regrouped its unfinished samples into

private:
    std::vector<Context*> ctx_;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = true;
    // device side: ONE slab (layouts, schedules, partial sums, tickets, step counter), pinned side:
    // ONE device-mapped slab (parameter rows, results, point counts, sequence word); both and the
    // stream come from / go back to the process-wide caches (context.h), and none of it exists
    // before the first step -- a Batch whose optimize() hands its samples to two halves never
    // allocates anything itself
    bool ready_ = false;
    int ensure_resources();
    std::vector<DeviceLayout> layouts_;
    void* d_slab_ = nullptr;
    size_t d_slab_bytes_ = 0;
    void* h_slab_ = nullptr;
    size_t h_slab_bytes_ = 0;
    DeviceLayout* d_layouts_ = nullptr;
    const Schedule* d_scheds_[kShapes] = {nullptr, nullptr, nullptr, nullptr};    // [shape]
    const Schedule* d_sched_arr_ = nullptr;                                       // [shape][sample]: what d_scheds_ points into
    bool slots_ = false;                    // create_slots: ctx_ entries may be null, set_slot changes them
    static constexpr size_t kSlotStageBytes = (sizeof(DeviceLayout) + kShapes * sizeof(Schedule) + 63) / 64 * 64;
    char* h_slot_stage_ = nullptr;          // pinned: [sample] layout + schedules on their way to the device
    double* d_partials_ = nullptr;
    unsigned int* d_tickets_ = nullptr;
    unsigned int* d_batch_done_ = nullptr;
    double *h_points_ = nullptr, *d_points_ = nullptr;     // mapped host memory, both views
    double *h_out_ = nullptr, *d_out_ = nullptr;
    int *h_nv_ = nullptr, *d_nv_ = nullptr;
    unsigned long long *h_done_ = nullptr, *d_done_ = nullptr;
    unsigned long long seq_ = 0;
    int bps_ = 1, block_waves_ = 16;
    bool w16_ = true;                       // every sample has the 16-bit run lists: the steps stream those
    bool wide_rows_ = true;                 // every sample has kRowBytesWide table rows (8-point launches allowed)
    size_t shmem_[kShapes] = {0, 0, 0, 0};  // [shape]
    int speculate_ = 4;                     // points a lock-step search evaluates per iteration (amoeba.h)
    // the step in flight (eval_begin .. eval_end)
    bool in_flight_ = false;
    bool in_split_ = false;                 // eval_begin is evaluating one request class of a mixed step
    MultiLaunch ml_{};
    int flight_np_ = 0;
    double* flight_out_ = nullptr;
    // optimize() of a big cohort: two half-cohorts taking turns on the device (see batch.cpp)
    static constexpr int kMaxLanes = 4;
    std::unique_ptr<Batch> half_[kMaxLanes];
    // the lanes' streams of optimize(): made back to back (different hardware queues: see cohort.cpp), lent to the lanes'
    // batches and to the batches their unfinished samples are regrouped into; they outlive half_
    hipStream_t lane_streams_[kMaxLanes] = {};
    int optimize_range(const vb2_model* models, int num_model, vb2_estimate* out);
};

}  // namespace vb2

struct vb2_batch {
    vb2::Batch* impl;
};

#endif
