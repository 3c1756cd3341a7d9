// ref_adapter.cpp -- TEST INFRASTRUCTURE ONLY: INTEGRATION.md section A, executed.
//
// The reference-side binding a VerifyBamID2 maintainer would write to put the MI355X likelihood
// behind the reference's OWN optimiser: a subclass of the reference's functor seam `VectorFunc`
// (statgen/MathVector.h:281-308) whose ComputeMixLLKs is one call of the C-ABI
// (vb2_llk_eval_batch, include/vb2_abi.h), installed with `myMinimizer.func = &fn` into the
// reference's own `AmoebaMinimizer` (MathGenMin.h:92-108) exactly as
// ContaminationEstimator.cpp:212,246,279,304,325 do.  It is compiled together with the reference's
// unmodified optimiser sources, read in place from $(REF) by oracle/Makefile, into oracle/_ref/ --
// nothing of the reference is copied into this repository.
//
// What is the reference's and what is restated: AmoebaMinimizer, GeneralMinimizer, Vector, VectorFunc
// are the reference's compiled code.  ContaminationEstimator.h itself cannot be compiled here (it
// includes htslib headers this image lacks; no stand-ins are written), so the thin members a
// maintainer would leave untouched -- Evaluate's six packings and best-so-far rule (h:339-442),
// Initialize / CalculateLLK0 (h:316-337), the OptimizeLLK sequence (cpp:88-155, 192-332) -- are
// restated below around the one line that changes: ComputeMixLLKs -> vb2_llk_eval_batch.
//
// libvb2.so is not linked: the caller passes the three entry points (so this file builds without
// the product, and the test decides which library drives the GPU).
#include "MathGenMin.h"   // reference header, found through -I$(REF)

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {
typedef int (*vb2_eval_batch_fn)(void* ctx, int32_t n, const double* pc1, const double* pc2,
                                 const double* alpha, double* llk_out);          // vb2_llk_eval_batch
typedef int (*vb2_search_begin_fn)(void* ctx);                                   // vb2_ctx_search_begin
typedef void (*vb2_search_end_fn)(void* ctx);                                    // vb2_ctx_search_end

struct vb2ref_adapter_io {
    // in: the model (main.cpp:285-319)
    int32_t num_pc, is_heter, is_pc_fixed, is_alpha_fixed;
    double fix_alpha, epsilon;
    const double* fix_pc;            // [num_pc] or NULL
    // out
    double alpha, llk1, llk0;
    double* pc;                      // [num_pc] fn.globalPC
    double* pc2;                     // [num_pc] fn.globalPC2
    int64_t num_eval;
    // trace of every Evaluate, in order: rows of (alpha, llk, pc1[k], pc2[k]); capacity in rows
    double* trace;
    int64_t trace_capacity, trace_count;
    int32_t error;                   // first non-zero status of the evaluator
};
}

namespace {

struct Estimator;

// FullLLKFunc (ContaminationEstimator.h:76-443) with the likelihood on the GPU.
class GpuLLKFunc : public VectorFunc {
public:
    Estimator* ptr = nullptr;
    double llk1 = 0, llk0 = 0;
    std::vector<double> fixPC, fixPC2, globalPC, globalPC2;
    double fixAlpha = 0, globalAlpha = 0;

    static double InvLogit(double x) { double e = exp(x); return e / (1. + e); }      // h:119-122
    static double Logit(double x) { return log(x / (1. - x)); }                      // h:124-127

    double ComputeMixLLKs(const std::vector<double>& tPC1, const std::vector<double>& tPC2, double alpha);
    int Initialize();
    int CalculateLLK0();
    virtual double Evaluate(Vector& v);
};

struct Estimator {
    vb2_eval_batch_fn eval;
    void* ctx;
    vb2ref_adapter_io* io;
    int numPC;
    bool isPCFixed = false, isAlphaFixed = false, isHeter = true;
    double alpha = 0.5, epsilon = 1e-8;                  // ContaminationEstimator.cpp:48; main.cpp:76
    std::vector<std::vector<double> > PC;
    GpuLLKFunc fn;
};

double GpuLLKFunc::ComputeMixLLKs(const std::vector<double>& tPC1, const std::vector<double>& tPC2, double alpha)
{
    double llk = 0.;
    const int rc = ptr->eval(ptr->ctx, 1, tPC1.data(), tPC2.data(), &alpha, &llk);     // <- the one changed line
    if (rc && !ptr->io->error) ptr->io->error = rc;
    vb2ref_adapter_io* io = ptr->io;
    io->num_eval++;
    if (io->trace && io->trace_count < io->trace_capacity) {
        const int k = ptr->numPC;
        double* row = io->trace + io->trace_count * (2 * k + 2);
        row[0] = alpha;
        row[1] = llk;
        std::memcpy(row + 2, tPC1.data(), sizeof(double) * k);
        std::memcpy(row + 2 + k, tPC2.data(), sizeof(double) * k);
    }
    io->trace_count++;
    return llk;
}

int GpuLLKFunc::Initialize()
{
    globalPC = fixPC = globalPC2 = fixPC2 = ptr->PC[1];
    globalAlpha = fixAlpha = ptr->alpha;
    llk1 = (0 - ComputeMixLLKs(fixPC, fixPC2, fixAlpha));
    for (int k = 0; k < ptr->numPC; ++k) ptr->PC[0][k] = 0.01;
    for (int k = 0; k < ptr->numPC; ++k) ptr->PC[1][k] = 0.01;
    ptr->alpha = 0.03;
    return 0;
}

int GpuLLKFunc::CalculateLLK0()
{
    llk0 = (0 - ComputeMixLLKs(globalPC, globalPC, 0));
    return 0;
}

// h:339-442
double GpuLLKFunc::Evaluate(Vector& v)
{
    double smLLK = 0;
    const int k = ptr->numPC;
    if (!ptr->isHeter) {
        if (ptr->isPCFixed) {
            double tmpAlpha = InvLogit(v[0]);
            smLLK = 0 - ComputeMixLLKs(fixPC, fixPC2, tmpAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalAlpha = tmpAlpha; }
        } else if (ptr->isAlphaFixed) {
            std::vector<double> tmpPC(k, 0.);
            for (int i = 0; i < k; ++i) tmpPC[i] = v[i];
            smLLK = 0 - ComputeMixLLKs(tmpPC, tmpPC, fixAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalPC = tmpPC; globalPC2 = tmpPC; }
        } else {
            std::vector<double> tmpPC(k, 0.);
            for (int i = 0; i < k; ++i) tmpPC[i] = v[i];
            double tmpAlpha = InvLogit(v[k]);
            smLLK = 0 - ComputeMixLLKs(tmpPC, tmpPC, tmpAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalPC = tmpPC; globalPC2 = tmpPC; globalAlpha = tmpAlpha; }
        }
    } else {
        if (ptr->isPCFixed) {
            std::vector<double> tmpPC(k, 0.);
            for (int i = 0; i < k; ++i) tmpPC[i] = v[i];
            double tmpAlpha = InvLogit(v[k]);
            smLLK = 0 - ComputeMixLLKs(tmpPC, fixPC2, tmpAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalPC = tmpPC; globalAlpha = tmpAlpha; }
        } else if (ptr->isAlphaFixed) {
            std::vector<double> tmpPC(k, 0.), tmpPC2(k, 0.);
            for (int i = 0; i < v.dim; ++i) {
                if (i < k) tmpPC[i] = v[i];
                else if (i < 2 * k) tmpPC2[i - k] = v[i];
            }
            smLLK = 0 - ComputeMixLLKs(tmpPC, tmpPC2, fixAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalPC = tmpPC; globalPC2 = tmpPC2; }
        } else {
            std::vector<double> tmpPC(k, 0.), tmpPC2(k, 0.);
            double tmpAlpha = 0.;
            for (int i = 0; i < v.dim; ++i) {
                if (i < k) tmpPC[i] = v[i];
                else if (i < 2 * k) tmpPC2[i - k] = v[i];
                else if (i == 2 * k) tmpAlpha = InvLogit(v[i]);
            }
            smLLK = 0 - ComputeMixLLKs(tmpPC, tmpPC2, tmpAlpha);
            if (smLLK < llk1) { llk1 = smLLK; globalPC = tmpPC; globalPC2 = tmpPC2; globalAlpha = tmpAlpha; }
        }
    }
    return smLLK;
}

// The six Optimize* wrappers (cpp:192-332) differ in the start vector and in what they read back.
void run(Estimator& e, AmoebaMinimizer& mini, int dim, bool with_pc1, bool with_pc2, bool with_alpha)
{
    const int k = e.numPC;
    Vector startingPoint("TestPoint", dim);
    int at = 0;
    if (with_pc1) for (int i = 0; i < k; ++i) startingPoint[at++] = e.PC[0][i];
    if (with_pc2) for (int i = 0; i < k; ++i) startingPoint[at++] = e.PC[1][i];
    if (with_alpha) startingPoint[at++] = GpuLLKFunc::Logit(e.alpha);
    startingPoint.label = "startPoint";
    mini.func = &e.fn;                                    // cpp:212
    mini.Reset(dim);
    mini.point = startingPoint;
    mini.Minimize(e.epsilon);
    at = 0;
    if (with_pc1) for (int i = 0; i < k; ++i) e.PC[0][i] = mini.point[at++];
    if (with_pc2) for (int i = 0; i < k; ++i) e.PC[1][i] = mini.point[at++];
    if (with_alpha) e.alpha = GpuLLKFunc::InvLogit(mini.point[at++]);
}

}  // namespace

// OptimizeLLK (ContaminationEstimator.cpp:88-155) with the reference's AmoebaMinimizer driving the
// GPU through the C-ABI; the whole search is bracketed by vb2_ctx_search_begin / _end
// (single-point calls served by the resident kernel).
extern "C" int vb2ref_adapter_optimize(vb2_eval_batch_fn eval, vb2_search_begin_fn begin, vb2_search_end_fn end,
                                       void* ctx, vb2ref_adapter_io* io)
{
    Estimator e;
    e.eval = eval;
    e.ctx = ctx;
    e.io = io;
    e.numPC = io->num_pc;
    e.PC.assign(2, std::vector<double>(io->num_pc, 0.));
    e.fn.ptr = &e;
    e.isHeter = io->is_heter != 0;
    if (io->epsilon > 0) e.epsilon = io->epsilon;
    if (io->is_pc_fixed && io->fix_pc) {                   // main.cpp:291-308
        for (int i = 0; i < io->num_pc; ++i) e.PC[1][i] = io->fix_pc[i];
        e.isPCFixed = true;
    } else if (io->is_alpha_fixed) {                       // main.cpp:309-313
        e.alpha = io->fix_alpha;
        e.isAlphaFixed = true;
    }
    io->num_eval = 0;
    io->trace_count = 0;
    io->error = 0;
    const int k = io->num_pc;

    AmoebaMinimizer myMinimizer;                           // the REFERENCE's minimiser
    if (begin) begin(ctx);
    e.fn.Initialize();
    if (!e.isHeter) {
        if (e.isPCFixed) run(e, myMinimizer, 1, false, false, true);
        else if (e.isAlphaFixed) run(e, myMinimizer, k, true, false, false);
        else run(e, myMinimizer, k + 1, true, false, true);
    } else {
        if (e.isPCFixed) {
            run(e, myMinimizer, k + 1, true, false, true);                 // OptimizeHeterFixedPC = OptimizeHomo
        } else if (e.isAlphaFixed) {
            e.isHeter = false;
            run(e, myMinimizer, k, true, false, false);
            e.PC[1] = e.PC[0];
            e.fn.globalPC2 = e.fn.globalPC;
            e.isHeter = true;
            run(e, myMinimizer, 2 * k, true, true, false);
        } else {
            e.isHeter = false;
            run(e, myMinimizer, k + 1, true, false, true);
            e.PC[1] = e.PC[0];
            e.fn.globalPC2 = e.fn.globalPC;
            e.isHeter = true;
            run(e, myMinimizer, 2 * k + 1, true, true, true);
        }
        if (e.fn.globalAlpha >= 0.5) {                     // cpp:146-149
            std::swap(e.fn.globalPC[0], e.fn.globalPC2[0]);
            if (k >= 2) std::swap(e.fn.globalPC[1], e.fn.globalPC2[1]);
        }
    }
    e.fn.CalculateLLK0();
    if (end) end(ctx);
    io->alpha = e.fn.globalAlpha;
    io->llk1 = e.fn.llk1;
    io->llk0 = e.fn.llk0;
    for (int i = 0; i < k; ++i) {
        io->pc[i] = e.fn.globalPC[i];
        io->pc2[i] = e.fn.globalPC2[i];
    }
    return io->error;
}
