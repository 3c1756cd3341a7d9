// estimator.cpp -- see estimator.h.
#include "tunables.h"
#include "estimator.h"

#include "context.h"
#include "line_search.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace vb2 {

double FullLLKFunc::InvLogit(double x)
{
    const double e = std::exp(x);
    return e / (1. + e);
}

double FullLLKFunc::Logit(double x) { return std::log(x / (1. - x)); }

int FullLLKFunc::LLK(const double* pc1, const double* pc2, double alpha, double* out)
{
    ptr->num_launch_point += 1;
    return ptr->eval_(ptr->user_, 1, pc1, pc2, &alpha, out);
}

static void record(Estimator* e, const double* pc1, const double* pc2, double alpha, double llk)
{
    e->num_eval++;
    vb2_trace* t = e->trace;
    if (!t) return;
    if (t->count < t->capacity) {
        const int64_t r = t->count;
        const int k = e->npc;
        t->alpha[r] = alpha;
        t->llk[r] = llk;
        std::memcpy(t->pc1 + r * k, pc1, sizeof(double) * k);
        std::memcpy(t->pc2 + r * k, pc2, sizeof(double) * k);
    }
    t->count++;
}

int FullLLKFunc::Initialize()
{
    globalPC = fixPC = globalPC2 = fixPC2 = ptr->coord[1];   // h:317
    globalAlpha = fixAlpha = ptr->mix;                   // h:318
    double v = 0;
    int rc = LLK(fixPC.data(), fixPC2.data(), fixAlpha, &v);
    if (rc) return rc;
    record(ptr, fixPC.data(), fixPC2.data(), fixAlpha, v);
    llk1 = (0 - v);                                        // h:319
    for (int k = 0; k < ptr->npc; ++k) ptr->coord[0][k] = 0.01;
    for (int k = 0; k < ptr->npc; ++k) ptr->coord[1][k] = 0.01;
    ptr->mix = 0.03;
    return 0;
}

int FullLLKFunc::CalculateLLK0()
{
    double v = 0;
    int rc = LLK(globalPC.data(), globalPC.data(), 0, &v);
    if (rc) return rc;
    record(ptr, globalPC.data(), globalPC.data(), 0, v);
    llk0 = (0 - v);
    return 0;
}

// The six packings of h:339-433 (table in SURVEY.md 3.2).
void FullLLKFunc::Unpack(const double* v, int dim, double* pc1, double* pc2, double* a) const
{
    const int k = ptr->npc;
    if (!ptr->isHeter) {
        if (ptr->isPCFixed) {                               // h:342-348
            std::memcpy(pc1, fixPC.data(), sizeof(double) * k);
            std::memcpy(pc2, fixPC2.data(), sizeof(double) * k);
            *a = InvLogit(v[0]);
        } else if (ptr->isAlphaFixed) {                     // h:349-360
            std::memcpy(pc1, v, sizeof(double) * k);
            std::memcpy(pc2, v, sizeof(double) * k);
            *a = fixAlpha;
        } else {                                            // h:361-374
            std::memcpy(pc1, v, sizeof(double) * k);
            std::memcpy(pc2, v, sizeof(double) * k);
            *a = InvLogit(v[k]);
        }
    } else {
        if (ptr->isPCFixed) {                               // h:377-389
            std::memcpy(pc1, v, sizeof(double) * k);
            std::memcpy(pc2, fixPC2.data(), sizeof(double) * k);
            *a = InvLogit(v[k]);
        } else if (ptr->isAlphaFixed) {                     // h:390-409
            for (int i = 0; i < k; ++i) pc1[i] = pc2[i] = 0.;
            for (int i = 0; i < dim; ++i) {
                if (i < k) pc1[i] = v[i];
                else if (i < 2 * k) pc2[i - k] = v[i];
            }
            *a = fixAlpha;
        } else {                                            // h:410-433
            for (int i = 0; i < k; ++i) pc1[i] = pc2[i] = 0.;
            *a = 0.;
            for (int i = 0; i < dim; ++i) {
                if (i < k) pc1[i] = v[i];
                else if (i < 2 * k) pc2[i - k] = v[i];
                else if (i == 2 * k) *a = InvLogit(v[i]);
            }
        }
    }
}

int FullLLKFunc::EvaluateBatch(int n, const double* pts, int dim, double* y)
{
    const int k = ptr->npc;
    std::vector<double> p1((size_t)n * k), p2((size_t)n * k), a(n), llk(n);
    for (int b = 0; b < n; ++b)
        Unpack(pts + (size_t)b * dim, dim, &p1[(size_t)b * k], &p2[(size_t)b * k], &a[b]);
    ptr->num_launch_point += n;
    const int rc = ptr->eval_(ptr->user_, n, p1.data(), p2.data(), a.data(), llk.data());
    if (rc) return rc;
    for (int b = 0; b < n; ++b) y[b] = 0 - llk[b];
    return 0;
}

void FullLLKFunc::Commit(const double* v, int dim, double smLLK)
{
    const int k = ptr->npc;
    double p1[VB2_MAX_PC], p2[VB2_MAX_PC], a;
    Unpack(v, dim, p1, p2, &a);
    record(ptr, p1, p2, a, 0 - smLLK);
    if (smLLK < llk1) {
        llk1 = smLLK;
        // which of (globalPC, globalPC2, globalAlpha) move depends on the model
        // variant exactly as in h:345-432
        if (!ptr->isHeter) {
            if (ptr->isPCFixed) {
                globalAlpha = a;
            } else if (ptr->isAlphaFixed) {
                globalPC.assign(p1, p1 + k);
                globalPC2.assign(p1, p1 + k);
            } else {
                globalPC.assign(p1, p1 + k);
                globalPC2.assign(p1, p1 + k);
                globalAlpha = a;
            }
        } else {
            if (ptr->isPCFixed) {
                globalPC.assign(p1, p1 + k);
                globalAlpha = a;
            } else if (ptr->isAlphaFixed) {
                globalPC.assign(p1, p1 + k);
                globalPC2.assign(p2, p2 + k);
            } else {
                globalPC.assign(p1, p1 + k);
                globalPC2.assign(p2, p2 + k);
                globalAlpha = a;
            }
        }
    }
    if (ptr->verbose)   // h:435-440 (the reference hard-codes PC indices 0 and 1)
        std::fprintf(stderr,
                     "NOTICE - ContaminatingSamplePC1:%f\tContaminatingSamplePC2:%f\t"
                     "IntendedSamplePC1:%f\tIntendedSamplePC2:%f\tFREEMIX(Alpha):%f\tllk:%f\n",
                     globalPC[0], k > 1 ? globalPC[1] : 0.0, globalPC2[0],
                     k > 1 ? globalPC2[1] : 0.0, globalAlpha, llk1);
}

Estimator::Estimator(int nPC, vb2_eval_fn eval, void* user)
    : npc(nPC), coord(2, std::vector<double>(nPC, 0.)), eval_(eval), user_(user)
{
    objective.ptr = this;
    objective.fixPC.assign(nPC, 0.);
    objective.fixPC2 = objective.globalPC = objective.globalPC2 = objective.fixPC;
}

// The whole Minimize() on the device when the context offers it (resident_kernel.inc): same
// decisions, same evaluation batches, bit-identical trajectory; one host round trip.
static bool device_minimizer(Estimator* e, AmoebaMinimizer& m, int dim, const std::vector<double>& start,
                             double* ret, bool* ok)
{
    Context* c = e->dev_ctx;
    if (!c || dim < 1 || dim > c->device_simplex_dim() || e->verbose) return false;
    Context::MinimizeRequest rq;
    rq.dim = dim;
    rq.kind = !e->isHeter ? (e->isPCFixed ? 0 : e->isAlphaFixed ? 1 : 2) : (e->isPCFixed ? 3 : e->isAlphaFixed ? 4 : 5);
    rq.start = start.data();
    rq.fix_pc = e->objective.fixPC.data();
    rq.fix_pc2 = e->objective.fixPC2.data();
    rq.g_pc = e->objective.globalPC.data();
    rq.g_pc2 = e->objective.globalPC2.data();
    rq.fix_alpha = e->objective.fixAlpha;
    rq.g_alpha = e->objective.globalAlpha;
    rq.llk1 = e->objective.llk1;
    rq.ftol = e->epsilon;
    rq.cycle_max = m.cycleMax;
    rq.trace = e->trace;
    const int64_t trace_before = e->trace ? e->trace->count : 0;
    if (c->device_minimize(&rq) != VB2_OK || rq.status == 3) {
        if (e->trace) e->trace->count = trace_before;
        return false;                                     // the host optimiser takes this search
    }
    m.Reset(dim);
    m.point.assign(rq.status == 1 ? rq.point : start.data(), (rq.status == 1 ? rq.point : start.data()) + dim);
    m.cycleCount = rq.cycle_count;
    e->objective.llk1 = rq.out_llk1;
    e->objective.globalAlpha = rq.out_g_alpha;
    e->objective.globalPC.assign(rq.out_g_pc, rq.out_g_pc + e->npc);
    e->objective.globalPC2.assign(rq.out_g_pc2, rq.out_g_pc2 + e->npc);
    e->num_eval += rq.num_eval;
    e->num_launch_point += rq.num_point;
    ++e->num_device_minimize;
    *ok = rq.status == 1;
    *ret = *ok ? rq.ret : std::numeric_limits<double>::max();
    if (*ok) m.fmin = rq.ret;
    return true;
}

static bool run_minimizer(Estimator* e, AmoebaMinimizer& m, int dim,
                          const std::vector<double>& start, double* ret)
{
    bool dev_ok = false;
    if (device_minimizer(e, m, dim, start, ret, &dev_ok)) {
        if (!dev_ok) {                                    // MathGenMin.cpp:381 (statgen warning())
            e->hit_cycle_limit = true;
            if (e->notices)
                std::fprintf(stderr, "WARNING - Amoeba.Minimize - Couldn't converge in %ld cycles\n", m.cycleMax);
        }
        return dev_ok;
    }
    m.func = &e->objective;
    m.speculate = e->speculate;
    if (tunables().speculate > 0) m.speculate = tunables().speculate;     // A/B and test knob: 1, 2, 4
    m.Reset(dim);
    m.point = start;
    *ret = m.Minimize(e->epsilon);
    if (m.error && !e->error) e->error = m.error;
    const bool ok = *ret != std::numeric_limits<double>::max();
    if (!ok) {                                            // MathGenMin.cpp:381 (statgen warning())
        e->hit_cycle_limit = true;
        if (e->notices)
            std::fprintf(stderr, "WARNING - Amoeba.Minimize - Couldn't converge in %ld cycles\n", m.cycleMax);
    }
    return ok;
}

// One simplex search over the parts of (contaminant's PCs | intended sample's PCs | logit of the mixing fraction) that a
// model leaves free -- the vector layout FullLLKFunc::Unpack expects for that model.  The reference has one hand-written
// wrapper per model (ContaminationEstimator.cpp:192-332); what differs between them is only which parts are free and
// whether a search that stopped at the cycle limit counts as a failure (its two --FixAlpha wrappers return true
// regardless: cpp:258, 312).
bool Estimator::Search(AmoebaMinimizer& m, const FreeParts& free_parts)
{
    std::vector<double> start;
    if (free_parts.contaminant) start.insert(start.end(), coord[0].begin(), coord[0].end());
    if (free_parts.intended) start.insert(start.end(), coord[1].begin(), coord[1].end());
    if (free_parts.mixing) start.push_back(FullLLKFunc::Logit(mix));
    const int dim = (int)start.size();
    double ret;
    const bool converged = run_minimizer(this, m, dim, start, &ret);
    int at = 0;
    if (free_parts.contaminant) { std::copy(m.point.begin(), m.point.begin() + npc, coord[0].begin()); at = npc; }
    if (free_parts.intended) { std::copy(m.point.begin() + at, m.point.begin() + at + npc, coord[1].begin()); at += npc; }
    if (free_parts.mixing) mix = FullLLKFunc::InvLogit(m.point[at]);
    return converged || free_parts.always_ok;
}

bool Estimator::OptimizeHeter(AmoebaMinimizer& m) { return Search(m, FreeParts{true, true, true, false}); }
bool Estimator::OptimizeHeterFixedAlpha(AmoebaMinimizer& m) { return Search(m, FreeParts{true, true, false, true}); }
bool Estimator::OptimizeHeterFixedPC(AmoebaMinimizer& m) { return OptimizeHomo(m); }       // (cpp:261-263)
bool Estimator::OptimizeHomo(AmoebaMinimizer& m) { return Search(m, FreeParts{true, false, true, false}); }
bool Estimator::OptimizeHomoFixedAlpha(AmoebaMinimizer& m) { return Search(m, FreeParts{true, false, false, true}); }

namespace {
// the one-parameter objective of the --FixPC models as a function of logit(alpha)
struct AlphaObjective : ScalarObjective {
    FullLLKFunc* llk;
    int EvaluateBatch(int n, const double* x, double* y) override { return llk->EvaluateBatch(n, x, 1, y); }
    void Commit(double x, double y) override { llk->Commit(&x, 1, y); }
};

// splitmix64 -> uniform -> Box-Muller: a self-contained, reproducible N(0, 1) stream per (seed, run)
struct Gauss {
    uint64_t s;
    uint64_t next()
    {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uniform() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    double normal() { return std::sqrt(-2.0 * std::log(uniform())) * std::cos(6.283185307179586 * uniform()); }
};
}  // namespace

// Bracket from the simplex's own two starting vertices (logit(alpha) and logit(alpha) + 1,
// MathGenMin.cpp:335-345 with scale 1), then Brent to --Epsilon (relative, on logit(alpha)).
bool Estimator::LineSearchAlpha()
{
    AlphaObjective obj;
    obj.llk = &objective;
    BrentMinimizer bm;
    bm.func = &obj;
    const double x0 = FullLLKFunc::Logit(mix);
    bm.Bracket(x0, x0 + 1.0);
    if (!bm.error) bm.Brent(epsilon);
    if (bm.error) {
        if (!error) error = bm.error;
        return false;
    }
    mix = FullLLKFunc::InvLogit(bm.min);
    if (bm.stuck) {
        hit_cycle_limit = true;
        if (notices) std::fprintf(stderr, "WARNING - ScalarMinimizer::Brent got stuck\n");
    }
    return !bm.stuck;
}

void Estimator::JitterStart()
{
    if (start_index <= 0) return;
    Gauss g{((uint64_t)start_seed << 32) ^ (0x5851f42d4c957f2dull * (uint64_t)start_index)};
    if (!isPCFixed) {
        for (int k = 0; k < npc; ++k) coord[0][k] += start_sd * g.normal();
        for (int k = 0; k < npc; ++k) coord[1][k] += start_sd * g.normal();
    }
    if (!isAlphaFixed) mix = FullLLKFunc::InvLogit(FullLLKFunc::Logit(mix) + 50.0 * start_sd * g.normal());
}

bool Estimator::OptimizeHomoFixedPC(AmoebaMinimizer& m)
{
    if (line_search) return LineSearchAlpha();
    return Search(m, FreeParts{false, false, true, false});
}

namespace {
// ContaminationEstimator.cpp:10-24: "NOTICE -   Starting phase: ..." / "Finished phase: ...  [s]"
struct PhaseTimer {
    const char* name;
    bool on;
    std::chrono::steady_clock::time_point start;
    PhaseTimer(const char* n, bool enabled) : name(n), on(enabled), start(std::chrono::steady_clock::now())
    {
        if (on) std::fprintf(stderr, "NOTICE -   Starting phase: %s\n", name);
    }
    ~PhaseTimer()
    {
        if (on)
            std::fprintf(stderr, "NOTICE -   Finished phase: %s  [%.3f seconds]\n", name,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());
    }
};
}  // namespace

int Estimator::OptimizeLLK()
{
    AmoebaMinimizer mini;
    int rc;
    {
        PhaseTimer t("Initialize likelihood", notices);
        rc = objective.Initialize();
    }
    if (rc) return rc;
    JitterStart();
    bool ok = true;
    if (!isHeter) {                                       // cpp:98-110
        PhaseTimer t(isPCFixed ? "OptimizeHomoFixedPC" : isAlphaFixed ? "OptimizeHomoFixedAlpha" : "OptimizeHomo",
                     notices);
        if (isPCFixed) ok &= OptimizeHomoFixedPC(mini);
        else if (isAlphaFixed) ok &= OptimizeHomoFixedAlpha(mini);
        else ok &= OptimizeHomo(mini);
    } else {                                              // cpp:111-150
        if (isPCFixed) {
            PhaseTimer t("OptimizeHeterFixedPC", notices);
            ok &= OptimizeHeterFixedPC(mini);
        } else if (isAlphaFixed) {
            {
                PhaseTimer t("OptimizeHomoFixedAlpha (initial)", notices);
                isHeter = false;
                ok &= OptimizeHomoFixedAlpha(mini);
                coord[1] = coord[0];
                objective.globalPC2 = objective.globalPC;
                isHeter = true;
            }
            PhaseTimer t("OptimizeHeterFixedAlpha", notices);
            ok &= OptimizeHeterFixedAlpha(mini);
        } else {
            {
                PhaseTimer t("OptimizeHomo (initial)", notices);
                isHeter = false;
                ok &= OptimizeHomo(mini);
                coord[1] = coord[0];
                objective.globalPC2 = objective.globalPC;
                isHeter = true;
            }
            PhaseTimer t("OptimizeHeter", notices);
            ok &= OptimizeHeter(mini);
        }
        if (objective.globalAlpha >= 0.5) {                      // cpp:146-149 (indices 0,1 hard-coded)
            std::swap(objective.globalPC[0], objective.globalPC2[0]);
            if (npc >= 2) std::swap(objective.globalPC[1], objective.globalPC2[1]);
        }
    }
    if (error) return error;
    converged = !hit_cycle_limit;
    (void)ok;                                             // the reference ignores the wrappers' results
    PhaseTimer t("Calculate null-model LLK", notices);
    return objective.CalculateLLK0();                            // cpp:152-155
}

void apply_model(Estimator& est, const vb2_model& model)
{
    // main.cpp:285-319
    est.verbose = model.verbose != 0;
    est.notices = model.notices != 0;
    est.isHeter = model.is_heter != 0;
    if (model.epsilon > 0) est.epsilon = model.epsilon;
    if (model.is_pc_fixed && model.fix_pc) {
        for (int i = 0; i < est.npc; ++i) est.coord[1][i] = model.fix_pc[i];
        est.isPCFixed = true;
    } else if (model.is_alpha_fixed) {
        est.mix = model.fix_alpha;
        est.isAlphaFixed = true;
    }
    if (model.is_af_known) {
        est.isAFknown = true;
        est.isPCFixed = true;
        est.isHeter = false;
    }
}

void apply_model(Estimator& est, const vb2_model& model, bool data_has_known_af)
{
    apply_model(est, model);
    if (data_has_known_af) {
        est.isAFknown = true;
        est.isPCFixed = true;
        est.isHeter = false;
    }
}

void fill_estimate(const Estimator& est, vb2_estimate* out)
{
    std::memset(out, 0, sizeof(*out));
    out->alpha = est.objective.globalAlpha;
    out->llk1 = est.objective.llk1;
    out->llk0 = est.objective.llk0;
    for (int i = 0; i < est.npc; ++i) {
        out->pc[i] = est.objective.globalPC[i];
        out->pc2[i] = est.objective.globalPC2[i];
    }
    out->num_eval = est.num_eval;
    out->num_launch_point = est.num_launch_point;
    out->converged = est.converged ? 1 : 0;
}

}  // namespace vb2
