// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin driver around the REFERENCE's own AmoebaMinimizer (MathGenMin.cpp:313-443)
// so the oracle's Nelder-Mead restatement can be checked against the real thing.
// It is compiled together with the reference's unmodified sources, read in place
// from $(REF) (= /root/reference) by oracle/Makefile, into oracle/_ref/ -- the
// sources themselves are never copied into this repository.  Only the optimiser
// half of the path builds this way: ContaminationEstimator.h pulls in htslib
// headers that this image lacks, so ComputeMixLLKs itself is NOT buildable from
// the reference here (no stand-in headers are written).
//
// The signature matches vb2o_minimizer in vb2_oracle.h.
#include "MathGenMin.h"   // reference header, found through -I$(REF)
#include "MathGold.h"     // ScalarMinimizer (Bracket + Brent), linked into the reference but never called there

#include <limits>

namespace {
typedef double (*objective_fn)(void *user, const double *v, int n);

// VectorFunc is the reference's objective seam (statgen/MathVector.h:281-308).
class CallbackFunc : public VectorFunc {
public:
    objective_fn fn;
    void *user;
    virtual double Evaluate(Vector &v) { return fn(user, v.data, v.dim); }
};
}  // namespace

extern "C" double vb2ref_amoeba_minimize(objective_fn f, void *user, int n,
                                         double *point, double ftol)
{
    CallbackFunc cb;
    cb.fn = f;
    cb.user = user;
    AmoebaMinimizer mini;          // same call sequence as ContaminationEstimator.cpp:211-215
    Vector start("startPoint", n);
    for (int i = 0; i < n; ++i) start[i] = point[i];
    mini.func = &cb;
    mini.Reset(n);
    mini.point = start;
    double ret = mini.Minimize(ftol);
    for (int i = 0; i < n; ++i) point[i] = mini.point[i];
    return ret;
}

namespace {
typedef double (*scalar_fn)(void *user, double x);
// ScalarMinimizer::f is virtual (MathGold.h:21): route it to the callback
class CallbackScalar : public ScalarMinimizer {
public:
    scalar_fn fn;
    void *user;
    virtual double f(double x) { return fn(user, x); }
};
}  // namespace

// The reference's own bracket + Brent on a caller's function: out = {min, fmin, a, b, c}.
extern "C" void vb2ref_scalar_minimize(scalar_fn f, void *user, double lo, double hi, double tol, double *out)
{
    CallbackScalar m;
    m.fn = f;
    m.user = user;
    m.Bracket(lo, hi);
    m.Brent(tol);
    out[0] = m.min; out[1] = m.fmin; out[2] = m.a; out[3] = m.b; out[4] = m.c;
}

extern "C" const char *vb2ref_describe(void)
{
    return "reference AmoebaMinimizer (MathGenMin.cpp) compiled in place";
}
