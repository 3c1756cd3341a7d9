// line_search.h -- one-dimensional minimiser: bracket a minimum by golden-ratio expansion with
// parabolic extrapolation, then Brent's method (golden-section steps + inverse parabolic
// interpolation).  Behavioural restatement of the reference's ScalarMinimizer
// (MathGold.h:8-27; Bracket: MathGold.cpp:27-96, Brent: MathGold.cpp:98-195), which the reference
// links but never calls: here it is the optional optimiser of the one-parameter
// models (--FixPC / --KnownAF: alpha alone is free), SURVEY.md 8f row 4.  Same constants
// (statgen/MathConstant.h:31-39), same decisions: fed the same function it evaluates the same
// abscissae in the same order (tests/test_abi_and_host.py pins that against the reference's own
// compiled ScalarMinimizer in oracle/_ref).
//
// The objective is batched like amoeba.h's: EvaluateBatch has no side effects, so the first three
// points of the bracket -- a, b and BOTH candidates for c, which of the two depending on f(a) < f(b)
// -- go out as one launch; Commit sees exactly the evaluations the reference would make, in order.
#ifndef VB2_LINE_SEARCH_H_
#define VB2_LINE_SEARCH_H_

namespace vb2 {

class ScalarObjective {
public:
    virtual ~ScalarObjective() {}
    virtual int EvaluateBatch(int n, const double* x, double* y) = 0;   // 0 = ok
    virtual void Commit(double x, double y) = 0;
};

class BrentMinimizer {
public:
    ScalarObjective* func = nullptr;
    double a = 0, b = 0, c = 0, min = 0;           // bracket (b between a and c), abscissa of the minimum
    double fa = 0, fb = 0, fc = 0, fmin = 0;
    int error = 0;                                 // first non-zero EvaluateBatch status
    bool stuck = false;                            // Brent ran out of iterations (the reference: numerror)
    bool speculate = true;                         // bracket's first three points in one batch

    void Bracket(double lo, double hi);            // MathGold.cpp:27-96
    double Brent(double tol = 1.0e-6);             // MathGold.cpp:98-195; returns fmin

private:
    double f(double x);                            // one committed evaluation
};

}  // namespace vb2
#endif
