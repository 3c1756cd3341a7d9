// estimator.h -- host-side mirror of the reference's optimisation layer:
//   FullLLKFunc   (ContaminationEstimator.h:76-443)   -> vb2::FullLLKFunc
//   OptimizeLLK + Optimize{Homo,Heter}{,FixedPC,FixedAlpha}
//                 (ContaminationEstimator.cpp:88-332) -> vb2::Estimator
// The likelihood itself is NOT computed here: every value comes from the
// evaluator callback (vb2_eval_fn), i.e. the HIP kernels behind the C-ABI.
#ifndef VB2_ESTIMATOR_H_
#define VB2_ESTIMATOR_H_

#include <cstdint>
#include <vector>

#include "../../include/vb2_abi.h"
#include "amoeba.h"

namespace vb2 {

class Estimator;
class Context;

class FullLLKFunc : public BatchObjective {
public:
    Estimator* ptr = nullptr;
    double llk1 = 0., llk0 = 0.;
    std::vector<double> fixPC, fixPC2, globalPC, globalPC2;
    double fixAlpha = 0., globalAlpha = 0.;

    static double InvLogit(double x);      // h:119-122
    static double Logit(double x);         // h:124-127

    int Initialize();                      // h:316-332
    int CalculateLLK0();                   // h:334-337
    int EvaluateBatch(int n, const double* pts, int dim, double* y) override;   // h:339-442 (values)
    void Commit(const double* pt, int dim, double y) override;                  // h:339-442 (bookkeeping)

private:
    // simplex vector -> (pc1, pc2, alpha) for the active model variant
    void Unpack(const double* v, int dim, double* pc1, double* pc2, double* alpha) const;
    int LLK(const double* pc1, const double* pc2, double alpha, double* out);
};

class Estimator {
public:
    Estimator(int nPC, vb2_eval_fn eval, void* user);

    // model flags, same names as the reference members (h:42-52)
    bool isPCFixed = false, isAlphaFixed = false, isAFknown = false, isHeter = true;
    bool verbose = false;
    int npc;
    double epsilon = 1e-8;
    double mix = 0.5;                      // the mixing fraction (the reference's `alpha`, cpp:48)
    std::vector<std::vector<double>> coord;   // [0] the contaminating sample's PCs, [1] the intended sample's (the reference's PC, h:453)
    FullLLKFunc objective;
    int speculate = 4;                  // AmoebaMinimizer::speculate
    // Optimiser variants (vb2_ctx_optimize_llk_ex), both off by default:
    //   line_search: a model with ONE free parameter (--FixPC / --KnownAF: alpha) is minimised by
    //     bracketing + Brent's method (line_search.h) instead of the two-vertex simplex;
    //   start_index > 0: this run is restart `start_index` of a multi-start search -- the free
    //     parameters start from the reference's values (h:328-331) plus seeded Gaussian noise.
    bool line_search = false;
    int start_index = 0;
    uint32_t start_seed = 0;
    double start_sd = 0.02;             // PC coordinates; logit(alpha) gets 50 x this

    int OptimizeLLK();                     // cpp:88-155 (without the writers)

    // bookkeeping
    int64_t num_eval = 0, num_launch_point = 0;
    bool converged = true;            // false iff a Minimize() hit cycleMax (MathGenMin.cpp:380-383)
    bool hit_cycle_limit = false;
    bool notices = false;             // print the reference's PhaseTimer / warning lines to stderr
    int error = 0;
    vb2_trace* trace = nullptr;

    vb2_eval_fn eval_;
    void* user_;
    // A context whose resident kernel can run a whole Minimize() on the device
    // (Context::device_minimize); nullptr = the host optimiser drives every search.
    Context* dev_ctx = nullptr;
    int64_t num_device_minimize = 0;

private:
    // which parts of (contaminant's PCs | intended sample's PCs | logit of the mixing fraction) a search varies, and whether
    // a search that ran into the cycle limit still counts as done (the reference's --FixAlpha wrappers: cpp:258, 312)
    struct FreeParts { bool contaminant, intended, mixing, always_ok; };
    bool Search(AmoebaMinimizer& m, const FreeParts& free_parts);
    bool OptimizeHomoFixedPC(AmoebaMinimizer& m);      // cpp:315-332
    bool LineSearchAlpha();                            // the same model through Brent's method
    void JitterStart();
    bool OptimizeHomoFixedAlpha(AmoebaMinimizer& m);   // cpp:291-313
    bool OptimizeHomo(AmoebaMinimizer& m);             // cpp:265-289
    bool OptimizeHeterFixedPC(AmoebaMinimizer& m);     // cpp:261-263
    bool OptimizeHeterFixedAlpha(AmoebaMinimizer& m);  // cpp:228-259
    bool OptimizeHeter(AmoebaMinimizer& m);            // cpp:192-226
};

void apply_model(Estimator& est, const vb2_model& model);
// The same for an evaluator whose data carry per-marker allele frequencies (a context created from
// --KnownAF input, Context::L.known_af): main.cpp:314-319 then forces isPCFixed and !isHeter whatever the
// model says.  ONE place, used by every context-, batch- and shard-level optimise entry point.
void apply_model(Estimator& est, const vb2_model& model, bool data_has_known_af);
void fill_estimate(const Estimator& est, vb2_estimate* out);

}  // namespace vb2
#endif
