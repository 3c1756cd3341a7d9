// amoeba.h -- downhill-simplex minimiser with the reference's exact decision
// rules (AmoebaMinimizer, MathGenMin.cpp:313-443; MathGenMin.h:92-108), driven
// through a BATCHED objective so one device launch can serve a whole
// Nelder-Mead iteration.
#ifndef VB2_AMOEBA_H_
#define VB2_AMOEBA_H_

#include <cstdint>
#include <vector>

namespace vb2 {

// The objective seam.  Generalises the reference's VectorFunc::Evaluate
// (statgen/MathVector.h:281-308) in two ways:
//   * EvaluateBatch computes raw objective values for several points at once and
//     has NO side effects, so points may be evaluated speculatively;
//   * Commit is called exactly once, in the reference's order, for every point
//     the reference's minimiser would have passed to Evaluate -- that is where
//     FullLLKFunc's best-so-far bookkeeping (ContaminationEstimator.h:345-432)
//     belongs.
class BatchObjective {
public:
    virtual ~BatchObjective() {}
    // pts: n rows of dim doubles (row-major); y: n outputs. Returns 0 on success.
    virtual int EvaluateBatch(int n, const double* pts, int dim, double* y) = 0;
    virtual void Commit(const double* pt, int dim, double y) = 0;
};

class AmoebaMinimizer {
public:
    BatchObjective* func = nullptr;     // GeneralMinimizer::func   (MathGenMin.h:16)
    std::vector<double> point;          // GeneralMinimizer::point  (MathGenMin.h:18)
    double fmin = 1.0e+100;             // FPMAX                    (MathGenMin.cpp:14)
    long cycleCount = 0, cycleMax = 50000;   // MathGenMin.cpp:314
    // Points evaluated per iteration before its outcome is known (the decisions, hence the
    // trajectory, are the same for every setting; Commit sees only what the reference evaluates):
    //   4: {R, E, C_A, C_R} -- every point the iteration can need, one batch per iteration
    //      (latency-bound callers: one sample on a whole GPU);
    //   2: {R, C_R} -- the reflection and the contraction that follows a rejected reflection, by far
    //      the most frequent second evaluation (on the C3-shaped searches: 45 % of the iterations
    //      end in C_R, 33 % need no second point, 22 % need E or C_A and cost a second batch);
    //   1: R alone, then whatever the reference evaluates next (throughput-bound callers).
    int speculate = 4;
    int error = 0;                      // first non-zero EvaluateBatch status

    void Reset(int ndim, double scale = 1.0);   // MathGenMin.cpp:316-324, 17-25
    // Returns fmin, or std::numeric_limits<double>::max() when cycleMax is exceeded
    // (point is then left untouched) -- MathGenMin.cpp:326-423.
    double Minimize(double ftol);

private:
    int dim_ = 0;
    double scale_ = 1.0;
    std::vector<double> simplex_;       // (n+1) x n
    std::vector<double> y_, psum_, ptry_;

    void TryPoint(const double* psum, const double* hi, double factor, double* out) const;
    bool Accept(int ihi, const double* pt, double ytry);
    void RecomputePsum();
};

}  // namespace vb2
#endif
