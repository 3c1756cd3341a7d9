// abi.cpp -- the extern "C" surface declared in include/vb2_abi.h.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include <deque>
#include <functional>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <new>

#include "tunables.h"
#include "batch.h"
#include "context.h"
#include "estimator.h"
#include "line_search.h"
#include "lockstep.h"
#include "hostio.h"
#include "shard.h"

using vb2::set_error;

namespace {

// Joins a helper thread on every path out of a scope: an exception on the main thread must not
// unwind past a joinable std::thread (that is std::terminate, not an error code).
struct JoinGuard {
    std::thread& t;
    explicit JoinGuard(std::thread& th) : t(th) {}
    ~JoinGuard() { if (t.joinable()) t.join(); }
};

int guard_ctx(const vb2_ctx* ctx)
{
    if (!ctx || !ctx->impl) {
        set_error("null vb2_ctx");
        return VB2_ERR_INVALID;
    }
    return VB2_OK;
}

int ctx_eval_cb(void* user, int32_t n, const double* pc1, const double* pc2, const double* alpha,
                double* out)
{
    return static_cast<vb2::Context*>(user)->eval_host(n, pc1, pc2, alpha, out);
}

// which "Estimation from ..." header the reference prints (ContaminationEstimator.cpp:98-150)
const char* estimation_title(const vb2_model& model)
{
    const bool heter = model.is_heter && !model.is_af_known;
    const bool pcfix = (model.is_pc_fixed && model.fix_pc) || model.is_af_known;
    const bool afix = !pcfix && model.is_alpha_fixed;
    if (!heter) return pcfix ? "Estimation from OptimizeHomoFixedPC:" : afix ? nullptr : "Estimation from OptimizeHomo:";
    return pcfix ? "Estimation from OptimizeHeterFixedPC:"
                 : afix ? "Estimation from OptimizeHeterFixedAlpha:" : "Estimation from OptimizeHeter:";
}

// The panel files of a run (reference order of errors: .bed, AF, .UD, .mu); .UD and .mu are parsed
// on helper threads while this one reads the .bed.
}  // namespace

namespace vb2 {
int load_panel(const vb2_run_args* a, vb2::Panel* panel)
{
    int rc_ud = VB2_OK, rc_mu = VB2_OK;
    std::string err_ud, err_mu;
    std::thread t_ud([&] {
        try { rc_ud = vb2::read_ud(a->ud_path, panel); if (rc_ud) err_ud = vb2::g_last_error; }
        catch (const std::exception& e) { rc_ud = VB2_ERR_NOMEM; err_ud = e.what(); }
    });
    std::thread t_mu([&] {
        try { rc_mu = vb2::read_mean(a->mean_path, panel); if (rc_mu) err_mu = vb2::g_last_error; }
        catch (const std::exception& e) { rc_mu = VB2_ERR_NOMEM; err_mu = e.what(); }
    });
    JoinGuard j_ud(t_ud), j_mu(t_mu);
    int rc = vb2::read_bed(a->bed_path, panel);
    if (!rc && a->known_af_path) rc = vb2::read_known_af(a->known_af_path, panel);
    if (!rc) panel->finish();
    const std::string err_main = rc ? vb2::g_last_error : std::string();
    t_ud.join();
    t_mu.join();
    if (rc) { set_error(err_main); return rc; }
    if (rc_ud) { set_error(err_ud); return rc_ud; }
    if (rc_mu) { set_error(err_mu); return rc_mu; }
    if (panel->means.size() < panel->NumMarker || panel->PosVec.size() < panel->NumMarker) {
        set_error(".UD has more rows than .mu/.bed");
        return VB2_ERR_INVALID;
    }
    return VB2_OK;
}
}  // namespace vb2

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int vb2_abi_version(void) { return VB2_ABI_VERSION; }

const char* vb2_last_error(void) { return vb2::g_last_error.c_str(); }

int vb2_device_count(void) { return vb2::usable_device_count(); }

int vb2_ctx_create(const vb2_input* in, const vb2_options* opt, vb2_ctx** out)
{
    if (!out) {
        set_error("vb2_ctx_create: out is NULL");
        return VB2_ERR_INVALID;
    }
    *out = nullptr;
    try {
        vb2::Context* c = nullptr;
        const int rc = vb2::Context::create(in, opt, &c);
        if (rc) return rc;
        vb2_ctx* h = new vb2_ctx{c};
        *out = h;
        return VB2_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

void vb2_ctx_destroy(vb2_ctx* ctx)
{
    if (!ctx) return;
    delete ctx->impl;
    delete ctx;
}

int vb2_ctx_info(const vb2_ctx* ctx, vb2_info* info)
{
    if (int rc = guard_ctx(ctx)) return rc;
    if (!info) return VB2_ERR_INVALID;
    ctx->impl->fill_info(info);
    return VB2_OK;
}

int vb2_llk_eval_batch(vb2_ctx* ctx, int32_t num_point, const double* pc1, const double* pc2,
                       const double* alpha, double* llk_out)
{
    if (int rc = guard_ctx(ctx)) return rc;
    return ctx->impl->eval_host(num_point, pc1, pc2, alpha, llk_out);
}

int vb2_llk_eval_batch_device(vb2_ctx* ctx, int32_t num_point, const double* d_points,
                              double* d_llk_out, void* stream)
{
    if (int rc = guard_ctx(ctx)) return rc;
    if (num_point < 0 || (num_point > 0 && (!d_points || !d_llk_out))) {
        set_error("vb2_llk_eval_batch_device: invalid argument");
        return VB2_ERR_INVALID;
    }
    return ctx->impl->eval_device(num_point, d_points, d_llk_out, (hipStream_t)stream);
}

// Profiling aid (not part of the public header): per-workgroup wall-clock stamps of the
// last launch when the context was created with VB2_STAMPS set.
// (test hook, not in vb2_abi.h) the host half of vb2_ctx_create without a device: digest of the flattened data
int vb2_debug_flatten_digest(const vb2_input* in, unsigned long long* digest)
{
    if (!in || !digest) return VB2_ERR_INVALID;
    try {
        return vb2::flatten_digest(in, digest);
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

// (test hook, not in vb2_abi.h) the same digest over the data arrays as they are on the device
// What the device sustains with the evaluation kernels' instruction mix (calib_kernels.hip): out[0] FP64 FMA
// lane-instructions / s, arithmetic alone; out[1] with the table reads feeding it; out[2] the latter counting every VALU
// instruction of the loop.  bench.py's roofline quotes the kernels against these instead of a nominal clock.
int vb2_debug_issue_ceiling(int device, double* out)
{
    if (!out) return VB2_ERR_INVALID;
    if (vb2::usable_device_count() < 1) return VB2_ERR_NO_DEVICE;
    return vb2::measure_issue_ceiling(device, out) == hipSuccess ? VB2_OK : VB2_ERR_HIP;
}

int vb2_debug_layout_digest(vb2_ctx* ctx, unsigned long long* digest)
{
    if (int rc = guard_ctx(ctx)) return rc;
    if (!digest) return VB2_ERR_INVALID;
    return ctx->impl->layout_digest(digest);
}

int vb2_debug_read_stamps(vb2_ctx* ctx, unsigned long long* out, int max_blocks)
{
    if (guard_ctx(ctx)) return 0;
    return ctx->impl->read_stamps(out, max_blocks);
}

// Test aid (not part of the public header): batches served by the resident search kernel.
long long vb2_debug_resident_evals(vb2_ctx* ctx)
{
    if (guard_ctx(ctx)) return -1;
    return (long long)ctx->impl->resident_evals;
}

int vb2_ctx_search_begin(vb2_ctx* ctx)
{
    if (int rc = guard_ctx(ctx)) return rc;
    (void)ctx->impl->resident_begin();         // unavailable -> plain launches, still a success
    return VB2_OK;
}

void vb2_ctx_search_end(vb2_ctx* ctx)
{
    if (guard_ctx(ctx)) return;
    ctx->impl->resident_end();
}

// Test aids (not part of the public header): Minimize() calls served by the on-device simplex;
// switch that mode off/on for one context (VB2_DEVICE_SIMPLEX does it globally).
long long vb2_debug_device_minimizes(vb2_ctx* ctx)
{
    if (guard_ctx(ctx)) return -1;
    return (long long)ctx->impl->device_minimizes;
}

void vb2_debug_set_device_simplex(vb2_ctx* ctx, int on)
{
    if (guard_ctx(ctx)) return;
    ctx->impl->device_simplex_enabled = on != 0;
}

// Test aid: is the context in resident mode right now?
// test hook (not in vb2_abi.h): `runs` OptimizeLLK searches over a caller's evaluator, advancing in
// lock-step as fibers of this thread (lockstep.h: what cohorts and multi-start searches use), every
// step's requests answered by ONE call of `eval` with all points concatenated.  Run i starts from
// start index i (vb2_search_opts semantics, seed 1).  No device involved: the CPU suite drives it
// with the oracle as evaluator and compares each run with the same search done on its own.
int vb2_debug_lockstep_optimize(vb2_eval_fn eval, void* user, int32_t num_pc, const vb2_model* model, int32_t runs,
                                int32_t speculate, vb2_estimate* out, int64_t* num_step)
{
    if (!eval || !model || !out || runs < 1 || num_pc < 1 || num_pc > VB2_MAX_PC) return VB2_ERR_INVALID;
    try {
        vb2::FiberGang gang(runs, 4);
        std::vector<int> rcs(runs, 0);
        std::vector<double> p1, p2, al, vals;
        auto body = [&](int i) {
            try {
                vb2::Estimator est(num_pc, vb2::FiberGang::eval_cb, gang.user(i));
                vb2::apply_model(est, *model);
                est.speculate = speculate;
                est.start_index = i;
                est.start_seed = 1;
                rcs[i] = est.OptimizeLLK();
                vb2::fill_estimate(est, &out[i]);
            } catch (...) {
                rcs[i] = VB2_ERR_INVALID;
            }
        };
        auto step = [&](std::vector<vb2::FiberGang::Request>& req) {
            p1.clear(); p2.clear(); al.clear();
            for (const auto& r : req) {
                if (r.n <= 0) continue;
                p1.insert(p1.end(), r.p1, r.p1 + (size_t)r.n * num_pc);
                p2.insert(p2.end(), r.p2, r.p2 + (size_t)r.n * num_pc);
                al.insert(al.end(), r.a, r.a + r.n);
            }
            vals.resize(al.size());
            if (const int rc = eval(user, (int32_t)al.size(), p1.data(), p2.data(), al.data(), vals.data())) return rc;
            size_t o = 0;
            for (auto& r : req) {
                if (r.n <= 0) continue;
                std::memcpy(r.out, &vals[o], sizeof(double) * r.n);
                o += (size_t)r.n;
            }
            return 0;
        };
        const int rc = gang.run(num_pc, body, step);
        if (num_step) *num_step = gang.steps;
        if (rc) return rc < 0 ? VB2_ERR_INVALID : rc;
        for (int i = 0; i < runs; ++i)
            if (rcs[i]) return rcs[i];
        return VB2_OK;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

// test hook (not in vb2_abi.h): the library's bracket + Brent (line_search.h) on a caller's scalar
// function; `committed` is called for every evaluation the reference's ScalarMinimizer would make,
// in its order (a speculative batch calls f for more points); out = {min, fmin, a, b, c}.
int vb2_debug_line_search(double (*f)(void* user, double x), void (*committed)(void* user, double x, double y),
                          void* user, double lo, double hi, double tol, int speculate, double* out)
{
    struct Obj : vb2::ScalarObjective {
        double (*f)(void*, double);
        void (*committed)(void*, double, double);
        void* user;
        int EvaluateBatch(int n, const double* x, double* y) override
        {
            for (int i = 0; i < n; ++i) y[i] = f(user, x[i]);
            return 0;
        }
        void Commit(double x, double y) override
        {
            if (committed) committed(user, x, y);
        }
    } obj;
    obj.f = f;
    obj.committed = committed;
    obj.user = user;
    vb2::BrentMinimizer bm;
    bm.func = &obj;
    bm.speculate = speculate != 0;
    bm.Bracket(lo, hi);
    if (!bm.error) bm.Brent(tol);
    out[0] = bm.min; out[1] = bm.fmin; out[2] = bm.a; out[3] = bm.b; out[4] = bm.c;
    return bm.stuck ? 1 : 0;
}

int vb2_debug_resident_active(vb2_ctx* ctx)
{
    if (guard_ctx(ctx)) return 0;
    return ctx->impl->resident_active ? 1 : 0;
}

// Test aid (not part of the public header): batches created from now on stream the 16-bit (1) or the 32-bit (0)
// run lists in their 1- and 2-point steps.
void vb2_debug_set_cohort_w16(int on) { vb2::tunables().cohort_w16 = on != 0; }

// Test aid: calls of more points than one launch's tables hold as the passes of one launch (1, the default) or as separate
// launches (0)
void vb2_debug_set_eval_passes(int on) { vb2::tunables().passes = on != 0; }

// Test / tool aid: the library's run-time switches by name (tunables.h).  0 on success, VB2_ERR_INVALID for an unknown name.
int vb2_debug_set_tunable(const char* name, int value) { return vb2::set_tunable(name, value) ? VB2_OK : VB2_ERR_INVALID; }
int vb2_debug_get_tunable(const char* name, int* value) { return vb2::get_tunable(name, value) ? VB2_OK : VB2_ERR_INVALID; }

// Test aid: turn the resident search mode off/on for one context (VB2_RESIDENT does it globally).
void vb2_debug_set_resident(vb2_ctx* ctx, int on)
{
    if (guard_ctx(ctx)) return;
    ctx->impl->resident_enabled = on != 0;
}

int vb2_optimize_llk(vb2_eval_fn eval, void* user, int32_t num_pc, const vb2_model* model,
                     vb2_estimate* out, vb2_trace* trace)
{
    if (!eval || !model || !out || num_pc < 1 || num_pc > VB2_MAX_PC) {
        set_error("vb2_optimize_llk: invalid argument");
        return VB2_ERR_INVALID;
    }
    try {
        vb2::Estimator est(num_pc, eval, user);
        vb2::apply_model(est, *model);
        est.trace = trace;
        if (trace) trace->count = 0;
        const int rc = est.OptimizeLLK();
        if (rc) return rc;
        vb2::fill_estimate(est, out);
        return VB2_OK;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_ctx_optimize_llk(vb2_ctx* ctx, const vb2_model* model, vb2_estimate* out, vb2_trace* trace)
{
    if (int rc = guard_ctx(ctx)) return rc;
    if (!model || !out) {
        set_error("vb2_ctx_optimize_llk: invalid argument");
        return VB2_ERR_INVALID;
    }
    vb2::Context* c = ctx->impl;
    // the whole search runs against one resident kernel when that mode is available, and each
    // Minimize() of it on the device itself (resident_kernel.inc) when the simplex fits
    if (trace && trace->capacity > 0 && c->device_simplex_enabled) (void)c->reserve_trace(trace->capacity);
    const bool resident = c->resident_begin();
    int rc;
    try {
        vb2::Estimator est(c->num_pc, ctx_eval_cb, c);
        vb2::apply_model(est, *model, c->L.known_af != nullptr);
        est.trace = trace;
        if (trace) trace->count = 0;
        if (resident && c->device_simplex_dim() > 0 && !(trace && c->trace_stage_rows < trace->capacity))
            est.dev_ctx = c;
        rc = est.OptimizeLLK();
        if (!rc) vb2::fill_estimate(est, out);
    } catch (const std::exception& e) {
        set_error(e.what());
        rc = VB2_ERR_INVALID;
    }
    if (resident) c->resident_end();
    return rc;
}

int vb2_ctx_optimize_llk_ex(vb2_ctx* ctx, const vb2_model* model, const vb2_search_opts* opts,
                            vb2_estimate* best, vb2_estimate* all)
{
    if (int rc = guard_ctx(ctx)) return rc;
    if (!model || !best) {
        set_error("vb2_ctx_optimize_llk_ex: invalid argument");
        return VB2_ERR_INVALID;
    }
    const int S = opts ? std::max(1, (int)opts->num_start) : 1;
    const bool line = opts && opts->line_search != 0;
    if (S > 64) {
        set_error("vb2_ctx_optimize_llk_ex: at most 64 starts");
        return VB2_ERR_INVALID;
    }
    if (S == 1 && !line) {
        const int rc = vb2_ctx_optimize_llk(ctx, model, best, nullptr);
        if (!rc && all) all[0] = *best;
        return rc;
    }
    vb2::Context* c = ctx->impl;
    const int k = c->num_pc;
    auto configure = [&](vb2::Estimator& est, int index) {
        vb2::apply_model(est, *model, c->L.known_af != nullptr);
        est.line_search = line;
        est.start_index = index;
        est.start_seed = opts ? opts->seed : 0u;
        if (opts && opts->start_sd > 0) est.start_sd = opts->start_sd;
        if (index > 0) est.notices = false;  // (the reference's phase lines once, for its own start)
    };
    try {
        if (S == 1) {
            // one run, Brent's single-point evaluations served by the resident kernel
            const bool resident = c->resident_begin();
            vb2::Estimator est(k, ctx_eval_cb, c);
            configure(est, 0);
            const int rc = est.OptimizeLLK();
            if (resident) c->resident_end();
            if (rc) return rc;
            vb2::fill_estimate(est, best);
            if (all) all[0] = *best;
            return VB2_OK;
        }
        // S runs as fibers of this thread: a step's points (<= 4 per run) leave as ONE launch
        vb2::FiberGang gang(S, 4);
        std::vector<vb2_estimate> ests(S);
        std::vector<int> rcs(S, 0);
        std::vector<double> p1, p2, al, out;
        auto body = [&](int i) {
            try {
                vb2::Estimator est(k, vb2::FiberGang::eval_cb, gang.user(i));
                configure(est, i);
                est.speculate = S * 4 <= vb2::kMaxPointsPerLaunch ? 4 : 2;     // the whole step in one launch
                rcs[i] = est.OptimizeLLK();
                vb2::fill_estimate(est, &ests[i]);
            } catch (const std::bad_alloc&) {
                rcs[i] = VB2_ERR_NOMEM;
            } catch (const std::exception& e) {
                set_error(e.what());
                rcs[i] = VB2_ERR_INVALID;
            } catch (...) {
                set_error("vb2_ctx_optimize_llk_ex: unknown exception in a search");
                rcs[i] = VB2_ERR_INVALID;
            }
        };
        auto step = [&](std::vector<vb2::FiberGang::Request>& req) {
            p1.clear(); p2.clear(); al.clear();
            for (const auto& r : req) {
                if (r.n <= 0) continue;
                p1.insert(p1.end(), r.p1, r.p1 + (size_t)r.n * k);
                p2.insert(p2.end(), r.p2, r.p2 + (size_t)r.n * k);
                al.insert(al.end(), r.a, r.a + r.n);
            }
            out.resize(al.size());
            if (const int rc = c->eval_host((int)al.size(), p1.data(), p2.data(), al.data(), out.data())) return rc;
            size_t o = 0;
            for (auto& r : req) {
                if (r.n <= 0) continue;
                std::memcpy(r.out, &out[o], sizeof(double) * r.n);
                o += (size_t)r.n;
            }
            return 0;
        };
        const int rc = gang.run(k, body, step);
        if (rc < 0) {
            set_error("vb2_ctx_optimize_llk_ex: getcontext failed");
            return VB2_ERR_INVALID;
        }
        if (rc) return rc;
        int win = -1;
        for (int i = 0; i < S; ++i) {
            if (rcs[i]) return rcs[i];
            // (a run that wandered into NaN territory never wins)
            if (ests[i].llk1 == ests[i].llk1 && (win < 0 || ests[i].llk1 < ests[win].llk1)) win = i;
        }
        if (win < 0) win = 0;
        *best = ests[win];
        best->reserved = win;
        if (all) std::memcpy(all, ests.data(), sizeof(vb2_estimate) * S);
        return VB2_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_batch_create(vb2_ctx* const* ctxs, int32_t num_sample, vb2_batch** out)
{
    if (!out) return VB2_ERR_INVALID;
    *out = nullptr;
    try {
        vb2::Batch* b = nullptr;
        const int rc = vb2::Batch::create(ctxs, num_sample, &b);
        if (rc) return rc;
        *out = new vb2_batch{b};
        return VB2_OK;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_NOMEM;
    }
}

void vb2_batch_destroy(vb2_batch* b)
{
    if (!b) return;
    delete b->impl;
    delete b;
}

int vb2_batch_eval(vb2_batch* b, const int32_t* num_point, const double* pc1, const double* pc2,
                   const double* alpha, double* llk_out)
{
    if (!b || !b->impl || !num_point || !pc1 || !pc2 || !alpha || !llk_out) {
        set_error("vb2_batch_eval: invalid argument");
        return VB2_ERR_INVALID;
    }
    return b->impl->eval(num_point, pc1, pc2, alpha, llk_out);
}

// Test aid: the batch evaluates the request classes of a step as separate launches (Batch::strict_shapes: what the
// streaming cohort search's batches do)
void vb2_debug_batch_set_strict(vb2_batch* b, int on)
{
    if (b && b->impl) b->impl->strict_shapes = on != 0;
}

// Test aid (not part of the public header): batches the last vb2_batch_optimize_llk regrouped its unfinished samples into.
long long vb2_debug_batch_regroups(vb2_batch* b)
{
    return b && b->impl ? (long long)b->impl->num_regroup : -1;
}

int vb2_batch_optimize_llk(vb2_batch* b, const vb2_model* models, int32_t num_model, vb2_estimate* out)
{
    if (!b || !b->impl) {
        set_error("vb2_batch_optimize_llk: null batch");
        return VB2_ERR_INVALID;
    }
    try {
        return b->impl->optimize(models, num_model, out);
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_shard_group_create(const vb2_input* in, const int32_t* devices, int32_t num_device, vb2_shard_group** out)
{
    if (!out) return VB2_ERR_INVALID;
    *out = nullptr;
    try {
        vb2::ShardGroup* g = nullptr;
        const int rc = vb2::ShardGroup::create(in, devices, num_device, &g);
        if (rc) return rc;
        *out = new vb2_shard_group{g};
        return VB2_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_rccl_unique_id(void* id128)
{
    if (!id128) return VB2_ERR_INVALID;
    return vb2::rccl_unique_id(id128);
}

int vb2_shard_group_create_rank(const vb2_input* in, int32_t device, int32_t rank, int32_t nranks,
                                const void* id128, vb2_shard_group** out)
{
    if (!out) return VB2_ERR_INVALID;
    *out = nullptr;
    try {
        vb2::ShardGroup* g = nullptr;
        const int rc = vb2::ShardGroup::create_rank(in, device, rank, nranks, id128, &g);
        if (rc) return rc;
        *out = new vb2_shard_group{g};
        return VB2_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_shard_group_eval(vb2_shard_group* g, int32_t num_point, const double* pc1, const double* pc2,
                         const double* alpha, double* llk_out)
{
    if (!g || !g->impl) {
        set_error("null vb2_shard_group");
        return VB2_ERR_INVALID;
    }
    return g->impl->eval(num_point, pc1, pc2, alpha, llk_out);
}

int vb2_shard_group_optimize_llk(vb2_shard_group* g, const vb2_model* model, vb2_estimate* out, vb2_trace* trace)
{
    if (!g || !g->impl) {
        set_error("null vb2_shard_group");
        return VB2_ERR_INVALID;
    }
    try {
        return g->impl->optimize(model, out, trace);
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

int vb2_shard_group_info(const vb2_shard_group* g, vb2_shard_info* info)
{
    if (!g || !g->impl || !info) return VB2_ERR_INVALID;
    std::memset(info, 0, sizeof(*info));
    const vb2::ShardGroup& sg = *g->impl;
    info->num_shard = (int32_t)sg.ctx.size();
    info->nranks = sg.nranks;
    info->rank = sg.rank;
    info->uses_rccl = sg.use_rccl ? 1 : 0;
    info->num_allreduce = sg.num_allreduce;
    info->partial_sums = sg.partial_sums ? 1 : 0;
    info->rccl_stub = (sg.use_rccl && vb2::rccl_is_stub()) ? 1 : 0;
    for (size_t s = 0; s < sg.ctx.size() && s < 64; ++s) {
        info->marker_lo[s] = sg.lo[s];
        info->marker_hi[s] = sg.hi[s];
        info->num_read[s] = sg.ctx[s]->num_read;
    }
    return VB2_OK;
}

int vb2_shard_range(const vb2_input* in, int32_t rank, int32_t nranks, int32_t* lo, int32_t* hi)
{
    if (!in || !in->read_off || !lo || !hi || nranks < 1 || rank < 0 || rank >= nranks || in->num_marker < 0) {
        set_error("vb2_shard_range: invalid argument");
        return VB2_ERR_INVALID;
    }
    int l = 0, h = 0;
    vb2::shard_range(in, rank, nranks, &l, &h);
    *lo = l;
    *hi = h;
    return VB2_OK;
}

void vb2_shard_group_destroy(vb2_shard_group* g)
{
    if (!g) return;
    delete g->impl;
    delete g;
}

int vb2_flat_load(const vb2_run_args* a, vb2_flat** out)
{
    if (!a || !out || !a->ud_path || !a->mean_path || !a->bed_path || (!a->pileup_path && !a->bam_path) ||
        a->num_pc < 1 || a->num_pc > VB2_MAX_PC) {
        set_error("vb2_flat_load: invalid argument");
        return VB2_ERR_INVALID;
    }
    *out = nullptr;
    try {
        std::unique_ptr<vb2_flat> f(new vb2_flat());
        f->panel.numPC = a->num_pc;
        int rc;
        // constructor reads the .bed (ContaminationEstimator.cpp:47), then ReadSVDMatrix
        // The four files are independent until the markers are resolved: .UD and .mu are parsed
        // on two helper threads while this one reads the .bed (the pileup reader needs it) and
        // the pileup.  Errors are reported in the reference's reading order (.bed, AF, .UD, .mu).
        const bool timing = vb2::tunables().debug_timing != 0;
        const double tl0 = now_s();
        double tl_bed = 0, tl_pile = 0, tl_join = 0;
        int rc_ud = VB2_OK, rc_mu = VB2_OK;
        std::string err_ud, err_mu;
        std::thread t_ud([&] {
            try { rc_ud = vb2::read_ud(a->ud_path, &f->panel); if (rc_ud) err_ud = vb2::g_last_error; }
            catch (const std::exception& e) { rc_ud = VB2_ERR_NOMEM; err_ud = e.what(); }
        });
        std::thread t_mu([&] {
            try { rc_mu = vb2::read_mean(a->mean_path, &f->panel); if (rc_mu) err_mu = vb2::g_last_error; }
            catch (const std::exception& e) { rc_mu = VB2_ERR_NOMEM; err_mu = e.what(); }
        });
        JoinGuard j_ud(t_ud), j_mu(t_mu);
        rc = vb2::read_bed(a->bed_path, &f->panel);
        if (!rc && a->known_af_path) rc = vb2::read_known_af(a->known_af_path, &f->panel);
        if (!rc) f->panel.finish();
        tl_bed = now_s();
        int rc_pile = VB2_OK;
        std::string err_main = rc ? vb2::g_last_error : std::string(), err_pile;
        if (!rc) {
            rc_pile = a->pileup_path ? vb2::read_pileup(a->pileup_path, f->panel, &f->viewer)
                                     : vb2::read_bam(a->bam_path, a->reference_path ? a->reference_path : "",
                                                     f->panel, &f->viewer, &a->mpileup);
            if (rc_pile) err_pile = vb2::g_last_error;
        }
        tl_pile = now_s();
        t_ud.join();
        t_mu.join();
        tl_join = now_s();
        if (rc) { set_error(err_main); return rc; }
        if (rc_ud) { set_error(err_ud); return rc_ud; }
        if (rc_mu) { set_error(err_mu); return rc_mu; }
        if (f->panel.means.size() < f->panel.NumMarker || f->panel.PosVec.size() < f->panel.NumMarker) {
            set_error(".UD has more rows than .mu/.bed");
            return VB2_ERR_INVALID;
        }
        if (rc_pile) { set_error(err_pile); return rc_pile; }
        f->sanity_disabled = a->disable_sanity != 0;
        if (!f->sanity_disabled && !vb2::sanity_check(f->panel, &f->viewer)) {
            set_error("Insufficient Available markers, check input bam depth distribution in "
                      "output pileup file after specifying --OutputPileup");
            f->resolve();
            *out = f.release();
            return VB2_ERR_SANITY;
        }
        const double tl_san = now_s();
        f->resolve();
        if (timing)
            std::fprintf(stderr, "vb2_flat_load: .bed %.1f ms, pileup %.1f ms, wait for .UD/.mu %.1f ms, "
                         "sanity %.1f ms, resolve %.1f ms\n", 1e3 * (tl_bed - tl0), 1e3 * (tl_pile - tl_bed),
                         1e3 * (tl_join - tl_pile), 1e3 * (tl_san - tl_join), 1e3 * (now_s() - tl_san));
        *out = f.release();
        return VB2_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VB2_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return VB2_ERR_INVALID;
    }
}

const vb2_input* vb2_flat_input(const vb2_flat* f) { return f ? &f->input : nullptr; }

int vb2_flat_stats(const vb2_flat* f, vb2_run_result* out)
{
    if (!f || !out) return VB2_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->num_marker = (int32_t)f->panel.NumMarker;
    out->num_site = f->num_site;
    out->num_bases = f->viewer.numBases;
    out->avg_depth = f->viewer.avgDepth;
    out->sd_depth = f->viewer.sdDepth;
    return VB2_OK;
}

void vb2_flat_free(vb2_flat* f) { delete f; }

int vb2_run(const vb2_run_args* a, vb2_run_result* out)
{
    if (!a || !out) {
        set_error("vb2_run: invalid argument");
        return VB2_ERR_INVALID;
    }
    const double t0 = now_s();
    // The HIP runtime takes ~0.2 s to come up in a fresh process: start that now, on a helper
    // thread, while this one reads the input files (errors surface later, in vb2_ctx_create).
    std::thread warm([dev = a->device] {
        if (dev >= 0) (void)hipSetDevice(dev);
        (void)hipFree(nullptr);
        (void)hipGetLastError();
    });
    vb2_flat* flat = nullptr;
    int rc;
    const bool notices = a->model.notices != 0;
    // main.cpp:321-378: the reference times "Load SVD reference data", "Read pileup" and "Marker
    // sanity check" one after the other; here the panel files and the pileup are read
    // concurrently, so the three are one phase
    if (notices) std::fprintf(stderr, "NOTICE - Starting phase: Load SVD reference data + Read pileup + Marker sanity check\n");
    {
        JoinGuard j_warm(warm);
        rc = vb2_flat_load(a, &flat);
    }
    if (notices) {
        std::fprintf(stderr, "NOTICE - Finished phase: Load SVD reference data + Read pileup + Marker sanity check  "
                             "[%.3f seconds]\n", now_s() - t0);
        if (!rc && !a->disable_sanity) std::fprintf(stderr, "NOTICE - Passing Marker Sanity Check...\n");
    }
    std::unique_ptr<vb2_flat> holder(flat);
    if (rc == VB2_ERR_SANITY && flat && a->output_pileup && a->output_prefix)
        vb2::write_pileup(a->output_prefix, *flat);
    if (rc) return rc;
    vb2_flat_stats(flat, out);
    if (a->output_pileup && a->output_prefix && (rc = vb2::write_pileup(a->output_prefix, *flat)))
        return rc;

    vb2_model model = a->model;
    if (flat->panel.isAFknown) model.is_af_known = 1;
    const double t_flat0 = now_s();
    if (a->devices && a->num_device > 1) {
        // --Devices a,b,...: the sample's markers sharded over the devices (vb2_shard_group_*)
        if (a->search.num_start > 1 || a->search.line_search) {      // (never silently ignored)
            set_error("--NumStart / --LineSearch are single-device options: they cannot be combined with marker "
                      "shards over several --Devices");
            return VB2_ERR_INVALID;
        }
        vb2_shard_group* grp = nullptr;
        if ((rc = vb2_shard_group_create(&flat->input, a->devices, a->num_device, &grp))) return rc;
        out->seconds_load = now_s() - t0;
        if (notices)
            std::fprintf(stderr, "NOTICE - Finished phase: Flatten + upload to %d devices (marker shards, %s)  "
                                 "[%.3f seconds]\n", (int)a->num_device,
                         grp->impl->use_rccl ? "RCCL all-reduce" : "host sum", now_s() - t_flat0);
        const double t1 = now_s();
        if (notices) std::fprintf(stderr, "NOTICE - Starting phase: Optimize likelihood\n");
        rc = vb2_shard_group_optimize_llk(grp, &model, &out->est, nullptr);
        out->seconds_optimize = now_s() - t1;
        if (notices)
            std::fprintf(stderr, "NOTICE - Finished phase: Optimize likelihood  [%.3f seconds]\n", out->seconds_optimize);
        vb2_shard_group_destroy(grp);
        if (rc) return rc;
    } else {
        vb2_options opt{};
        opt.device = (a->devices && a->num_device == 1) ? a->devices[0] : a->device;
        vb2_ctx* ctx = nullptr;
        if ((rc = vb2_ctx_create(&flat->input, &opt, &ctx))) return rc;
        out->seconds_load = now_s() - t0;
        if (notices)
            std::fprintf(stderr, "NOTICE - Finished phase: Flatten + upload to %s  [%.3f seconds]\n",
                         ctx->impl->device_name, now_s() - t_flat0);
        const double t1 = now_s();
        if (notices) std::fprintf(stderr, "NOTICE - Starting phase: Optimize likelihood\n");      // main.cpp:382
        rc = (a->search.num_start > 1 || a->search.line_search)
                 ? vb2_ctx_optimize_llk_ex(ctx, &model, &a->search, &out->est, nullptr)
                 : vb2_ctx_optimize_llk(ctx, &model, &out->est, nullptr);
        out->seconds_optimize = now_s() - t1;
        if (notices)
            std::fprintf(stderr, "NOTICE - Finished phase: Optimize likelihood  [%.3f seconds]\n", out->seconds_optimize);
        vb2_ctx_destroy(ctx);
        if (rc) return rc;
    }

    vb2::print_summary(estimation_title(model), a->num_pc, out->est);
    if (a->output_prefix) {
        if ((rc = vb2::write_ancestry(a->output_prefix, a->num_pc, out->est.pc, out->est.pc2))) return rc;
        // main.cpp:398-400: #READS is viewer.numBases for BAM input, "NA" for pileup input
        if ((rc = vb2::write_selfsm(a->output_prefix, *flat, out->est, a->pileup_path != nullptr))) return rc;
    }
    return VB2_OK;
}

}  // extern "C"
