// context.h -- vb2_ctx internals.
#ifndef VB2_CONTEXT_H_
#define VB2_CONTEXT_H_

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <string>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/vb2_abi.h"
#include "llk_kernels.h"

namespace vb2 {

void set_error(const std::string& msg);
extern thread_local std::string g_last_error;
int usable_device_count();
// CPUs this process may actually use: min(hardware threads, affinity mask, cgroup CPU quota).  A
// container with a CFS quota still sees every core of the host; threads beyond the quota only
// get the whole cgroup throttled (measured on the 256-thread GPU host with a 16-CPU quota).
int usable_cpu_count();
// The host half of vb2_ctx_create (flatten) without a device: timing aid for tools/ubench/host_pipeline.cpp.
int flatten_dry_run(const vb2_input* in, double* ms);
// ... and a digest of what it produced (run words, tile records, panel rows, per-marker constants, dictionary): lets a
// test hold two ways of flattening -- the AVX2 classification and the scalar statements -- to the same bytes without a device.
int flatten_digest(const vb2_input* in, unsigned long long* digest);
// The process-wide recycling of device slabs, pinned device-mapped slabs and streams (context.cpp):
// allocation and release calls cost milliseconds and serialise in the driver.  take: nullptr = none
// cached, allocate yourself; give: false = cache full, release it yourself.
void* cached_device_slab(size_t bytes, int device, size_t* got);
bool recycle_device_slab(void* p, size_t bytes, int device);
void* cached_pinned_slab(size_t bytes, int device, size_t* got);
bool recycle_pinned_slab(void* p, size_t bytes, int device);
hipStream_t cached_stream(int device);
bool recycle_stream(hipStream_t stream, int device);
// destroys the idle streams the cache holds for `device` (the cohort runner, before it makes its own: see cohort.cpp)
void drop_cached_streams(int device);
extern std::atomic<int> g_flatten_thread_cap;   // 0 = no cap on the flatten threads of vb2_ctx_create

constexpr int kStagePoints = 256;   // points per host<->device staging round

// NaN parameters: in the reference every marker's likelihood is then NaN, fails `markerLK > 0` (h:310) and is left out
// -- the sum over no markers, 0 (the kernels' clamps -- v_max / v_min quiet a NaN away -- would answer with the alpha-free part
// of the likelihood instead).  alpha enters every table entry, the PCs every allele frequency unless the frequencies are known.
// One statement of the rule for the host entries that take a caller's points (Context::eval_host, ShardGroup::eval; a
// cohort's steps carry the points of its own optimisers, which do not make NaNs).
inline bool params_hold_nan(const double* pc1, const double* pc2, double alpha, int k, bool known_af)
{
    bool bad = alpha != alpha;
    if (!known_af)
        for (int j = 0; j < k; ++j) bad |= pc1[j] != pc1[j] || pc2[j] != pc2[j];
    return bad;
}

class Context : public ScheduleProvider {
public:
    ~Context();
    // static work schedule of a launch shape, built on first use (llk_kernels.h: Schedule)
    Schedule get(int mode, int ngrp, int grid, int block_waves) override;
    static int sched_slot(int mode, int ngrp) { return mode == 2 ? ngrp - 1 : mode == 1 ? 6 : mode == 3 ? 7 : 8; }
    struct SchedSlot {
        bool tried = false;
        Schedule s{nullptr, nullptr};
        char* d_base = nullptr;              // reserved space in the device slab
        size_t bytes = 0;
    };
    SchedSlot sched_[9];
    bool sched_enabled = true;               // VB2_SCHED=0: in-kernel snake deal
    // Static schedules of a cohort step's four wave shapes (batch.h: <= 4, 8, 1, 2 points per sample) for this sample served
    // by `bps` workgroups of `block_waves` waves: built and uploaded on first use, kept for the context's lifetime (a
    // reader thread of vb2_cohort_run prepares them; Batch::ensure_resources then only collects pointers: building them there
    // cost 0.35 ms per C3 sample on the thread that feeds the device).  {nullptr, nullptr} where the launch takes the work
    // queue (no schedule needed) or none could be built (-> the in-kernel snake deal).  Thread-safe.
    int cohort_schedules(int bps, int block_waves, Schedule out[4], bool full = true);   // !full: the 1- and 2-point shapes only
    struct CohortSched {
        int bps = 0, block_waves = 0;
        bool full = true;
        Schedule s[4] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
        void* d_mem = nullptr;
        size_t bytes = 0;
    };
    std::vector<CohortSched> cohort_sched_;
    std::mutex cohort_mu_;
    std::vector<uint32_t> h_mt_rows;         // rows per micro-tile (host copy: the schedules are built from it)
    std::vector<uint32_t> h_mt_rec_y;        // probability domain: a tile's {ref steps | all steps << 16} (ensure_codes16: the 8-bit lists)
    static int create(const vb2_input* in, const vb2_options* opt, Context** out);
    static int create_impl(const vb2_input* in, const vb2_options* opt, Context** out, bool dry);
    // device pointers, asynchronous on s (nullptr = own stream)
    int eval_device(int num_point, const double* d_points, double* d_llk, hipStream_t s,
                    unsigned long long* done_flag = nullptr, unsigned long long done_seq = 0,
                    const double* h_points = nullptr, int reduce_override = 0);
    // host pointers, synchronous
    int eval_host(int num_point, const double* pc1, const double* pc2, const double* alpha,
                  double* llk_out);
    // Resident search mode (llk_resident_kernel): between begin and end, eval_host posts its
    // batches to the kernel that is already on the CUs instead of launching one per call.
    // begin() returns false (and changes nothing) when the mode is unavailable; any failure
    // later falls back to plain launches on its own.
    bool resident_begin();
    void resident_end();
    // The two halves of a resident evaluation (eval_host = submit + collect): posts <= 4 rows of
    // 2k+1 doubles to the mailbox; then spins for the results.  collect returns false when the
    // kernel did not answer (the mode is then torn down: redo the rows with plain launches).
    void resident_submit(int n, const double* rows);
    bool resident_collect(int n, double* out);
    // One AmoebaMinimizer::Minimize() run entirely on the device (resident_kernel.inc: the simplex
    // logic in workgroup 0 of the resident kernel; one mailbox round trip for the whole search).
    // Available between resident_begin and resident_end when device_simplex_dim() >= the simplex
    // dimension.  Returns VB2_OK and fills req's outputs, or a negative code / req.status == 3 when
    // the search must be (re)done by the host optimiser -- the request's inputs are never modified.
    struct MinimizeRequest {
        // in
        int dim = 0, kind = 0;                  // kind: FullLLKFunc::Evaluate's packing, 0..5 (resident_kernel.inc)
        const double* start = nullptr;          // [dim]
        const double *fix_pc = nullptr, *fix_pc2 = nullptr, *g_pc = nullptr, *g_pc2 = nullptr;   // [k]
        double fix_alpha = 0, g_alpha = 0, llk1 = 0, ftol = 1e-8;
        long cycle_max = 50000;
        vb2_trace* trace = nullptr;             // appended to (like Estimator's record())
        // out
        int status = 0;                         // 1 converged, 2 cycle limit, 3 redo on the host
        double ret = 0, out_llk1 = 0, out_g_alpha = 0;
        long cycle_count = 0, num_eval = 0, num_point = 0;
        double point[kDeviceSimplexMaxDim];
        double out_g_pc[VB2_MAX_PC], out_g_pc2[VB2_MAX_PC];
    };
    int device_simplex_dim() const { return resident_active ? resident_nmax : 0; }
    int device_minimize(MinimizeRequest* req);
    // Mapped staging for the evaluation trace of on-device searches; call BEFORE resident_begin.
    int reserve_trace(int64_t rows);
    double *h_trace_stage = nullptr, *d_trace_stage = nullptr;
    int64_t trace_stage_rows = 0;
    void fill_info(vb2_info* info) const;
    int read_stamps(unsigned long long* out, int max_blocks);
    int layout_digest(unsigned long long* digest);     // (test hook) the flatten_digest of what is ON THE DEVICE
    std::vector<std::pair<size_t, size_t>> dbg_regions;   // (offset, bytes) of the defined data regions, in digest order
    int64_t dbg_counts[4] = {0, 0, 0, 0};
    // The cohort-step copy of the run lists (DeviceLayout::codes16): built on the device from `codes`, in the
    // slab when the context was created with VB2_OPT_COHORT_LAYOUT, else in an allocation of its own here.
    int ensure_codes16();
    void* d_codes16_own = nullptr;
    int64_t cohort_bytes = 0;                 // device bytes one cohort step streams for this sample
    int num_code_seen = 0;                    // distinct (class, quality) codes of the data (vb2_info::num_code)

    int device = -1;
    int num_marker = 0;
    int num_pc = 0;
    DeviceLayout L{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* d_slab = nullptr;                   // the one device allocation the pointers below (and L.*) point into
    void* h_slab = nullptr;                   // the one pinned, device-mapped host allocation
    size_t d_slab_bytes = 0, h_slab_bytes = 0;   // allocated sizes (slabs are recycled through a cache)
    double* d_partials = nullptr;
    unsigned int* d_ticket = nullptr;
    unsigned long long* d_stamps = nullptr;   // VB2_STAMPS profiling aid
    double* h_points = nullptr;   // pinned + device-mapped staging (host view)
    double* h_out = nullptr;
    double* d_points = nullptr;   // device view of the same memory
    double* d_out = nullptr;
    unsigned long long* h_done = nullptr;   // completion sequence number (mapped host memory)
    unsigned long long* d_done = nullptr;
    unsigned long long done_seq_ = 0;
    bool spin_wait = true;
    bool resident_enabled = true;            // Tunables::resident / VB2_OPT_LAUNCH_PER_STEP turn the mode off
    bool resident_active = false;
    bool plain_launch = false;               // the resident kernel goes up with a plain launch (profiler, VB2_OPT_PLAIN_LAUNCH)
    bool resident_cooperative = false;       // ... and the one that is up went up cooperatively
    unsigned long long* h_cmd = nullptr;     // mailbox (mapped host memory) + device view
    unsigned long long* d_cmd = nullptr;
    unsigned long long* d_relay = nullptr;
    unsigned int* h_state = nullptr;
    unsigned int* d_state = nullptr;
    double* h_result = nullptr;              // result block of an on-device Minimize() (mapped host memory)
    double* d_result = nullptr;
    int resident_nmax = 0;                   // largest simplex dimension the running resident kernel supports
    bool device_simplex_enabled = true;      // Tunables::device_simplex / VB2_OPT_HOST_SEARCH: the host optimiser drives every search
    unsigned long long resident_epoch_ = 0;
    int64_t device_minimizes = 0;            // Minimize() calls served on the device
    int64_t resident_evals = 0;              // batches served by the resident kernel
    int64_t nan_retries = 0;                 // plain launches redone in ticket mode after a "never reported" NaN
    // VB2_DEBUG_TIMING: where a resident search's wall-clock goes (host logic vs device round trip)
    bool dbg_timing = false, dbg_have_prev = false;
    int64_t dbg_cmds = 0;
    double dbg_host_ns = 0, dbg_wait_ns = 0;
    std::chrono::steady_clock::time_point dbg_prev_seen, dbg_t_post;
    int64_t num_read = 0, num_read_other = 0, device_bytes = 0, algorithmic_bytes = 0;
    char device_name[64] = {0};
    char arch[32] = {0};
};

}  // namespace vb2

namespace vb2 {
// Results come back through mapped host memory as relaxed stores behind a relaxed flag (context.cpp): callers set the result
// words to NaN before a step and, once the flag is seen, wait here a few microseconds for NaNs still on their way.
// false: a NaN stayed (the kernels' own "a workgroup never reported", or the input's).
bool settle_results(const double* out, int n);
}  // namespace vb2

struct vb2_ctx {
    vb2::Context* impl;
};

#endif
