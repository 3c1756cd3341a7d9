// amoeba.cpp -- see amoeba.h.  Decision rules, tie handling (<= / >=), cycle
// accounting and floating-point operation order follow MathGenMin.cpp:326-443.
#include "amoeba.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

namespace vb2 {

namespace {
const double kZeps = 3.0e-10;    // statgen/MathConstant.h:34
const double kFpMax = 1.0e+100;  // statgen/MathConstant.h:36
}

void AmoebaMinimizer::Reset(int ndim, double scale)
{
    dim_ = ndim;
    scale_ = scale;
    point.assign(ndim, 0.0);
    simplex_.assign((size_t)(ndim + 1) * ndim, 0.0);
    y_.assign(ndim + 1, 0.0);
    psum_.assign(ndim, 0.0);
    ptry_.assign(ndim, 0.0);
    fmin = kFpMax;
    error = 0;
}

// ptry = fac*psum, then ptry += (factor-fac)*simplex[ihi]: two passes, like
// Vector::SetMultiple + Vector::AddMultiple (MathGenMin.cpp:429-431).
void AmoebaMinimizer::TryPoint(const double* psum, const double* hi, double factor, double* out) const
{
    const int n = dim_;
    const double fac = (1.0 - factor) / n;
    for (int i = 0; i < n; ++i) out[i] = fac * psum[i];
    for (int i = 0; i < n; ++i) out[i] += (factor - fac) * hi[i];
}

// MathGenMin.cpp:434-440
bool AmoebaMinimizer::Accept(int ihi, const double* pt, double ytry)
{
    if (!(ytry < y_[ihi])) return false;
    const int n = dim_;
    double* hi = &simplex_[(size_t)ihi * n];
    y_[ihi] = ytry;
    for (int i = 0; i < n; ++i) psum_[i] -= hi[i];
    for (int i = 0; i < n; ++i) hi[i] = pt[i];
    for (int i = 0; i < n; ++i) psum_[i] += hi[i];
    return true;
}

void AmoebaMinimizer::RecomputePsum()   // MathGenMin.cpp:350-351, 416-417
{
    const int n = dim_;
    for (int i = 0; i < n; ++i) psum_[i] = simplex_[i];
    for (int m = 1; m <= n; ++m)
        for (int i = 0; i < n; ++i) psum_[i] += simplex_[(size_t)m * n + i];
}

double AmoebaMinimizer::Minimize(double ftol)
{
    const int n = dim_, nvertex = n + 1;
    if (n == 0) {                                       // MathGenMin.cpp:331-332
        double y = 0;
        if ((error = func->EvaluateBatch(1, point.data(), 0, &y))) return fmin;
        func->Commit(point.data(), 0, y);
        return fmin = y;
    }

    // initial simplex: vertex i = point + e_i*scale, vertex n = point (cpp:335-345)
    for (int i = 0; i < n; ++i) {
        double* v = &simplex_[(size_t)i * n];
        for (int j = 0; j < n; ++j) v[j] = point[j] + (i == j ? scale_ : 0.0);
    }
    std::memcpy(&simplex_[(size_t)n * n], point.data(), sizeof(double) * n);
    if ((error = func->EvaluateBatch(nvertex, simplex_.data(), n, y_.data()))) return fmin;
    for (int i = 0; i < nvertex; ++i) {
        func->Commit(&simplex_[(size_t)i * n], n, y_[i]);
        if (y_[i] < fmin) fmin = y_[i];
    }
    cycleCount = nvertex;
    RecomputePsum();

    std::vector<double> cand((size_t)4 * n), ycand(4), psum_acc(n);
    std::vector<double> shrink_pts((size_t)nvertex * n), shrink_y(nvertex);

    for (;;) {
        int ilo, ihi, inhi;                              // cpp:357-370
        if (y_[0] > y_[1]) { ilo = inhi = 1; ihi = 0; }
        else               { ilo = inhi = 0; ihi = 1; }
        for (int i = 2; i < nvertex; ++i) {
            if (y_[i] <= y_[ilo]) ilo = i;
            else if (y_[i] > y_[ihi]) { inhi = ihi; ihi = i; }
            else if (y_[i] > y_[inhi]) inhi = i;
        }
        const double rtol = 2 * std::fabs(y_[ihi] - y_[ilo]) /
                            (std::fabs(y_[ihi]) + std::fabs(y_[ilo]) + kZeps);   // cpp:373
        if (rtol < ftol) {
            std::memcpy(point.data(), &simplex_[(size_t)ilo * n], sizeof(double) * n);
            return fmin = y_[ilo];
        }
        if (cycleCount > cycleMax)                       // cpp:380-383 (point untouched)
            return std::numeric_limits<double>::max();

        cycleCount += 2;                                 // cpp:389
        double* R = &cand[0];
        double* E = &cand[(size_t)n];
        double* CA = &cand[(size_t)2 * n];
        double* CR = &cand[(size_t)3 * n];
        const double* hi = &simplex_[(size_t)ihi * n];
        TryPoint(psum_.data(), hi, -1.0, R);
        if (speculate >= 4) {
            // State the simplex would be in if the reflection is accepted.
            for (int i = 0; i < n; ++i) psum_acc[i] = psum_[i] - hi[i];
            for (int i = 0; i < n; ++i) psum_acc[i] += R[i];
            TryPoint(psum_acc.data(), R, 2.0, E);        // expansion after acceptance
            TryPoint(psum_acc.data(), R, 0.5, CA);       // contraction after acceptance
            TryPoint(psum_.data(), hi, 0.5, CR);         // contraction, reflection rejected
            if ((error = func->EvaluateBatch(4, cand.data(), n, ycand.data()))) return fmin;
        } else if (speculate >= 2) {
            TryPoint(psum_.data(), hi, 0.5, E);          // (second row of the batch: C_R)
            if ((error = func->EvaluateBatch(2, cand.data(), n, ycand.data()))) return fmin;
            std::memcpy(CR, E, sizeof(double) * n);
            ycand[3] = ycand[1];
        } else {
            if ((error = func->EvaluateBatch(1, R, n, ycand.data()))) return fmin;
        }
        double ytry = ycand[0];
        func->Commit(R, n, ytry);
        const bool accepted = Accept(ihi, R, ytry);      // cpp:390 (Amoeba(ihi,-1.0))

        if (ytry <= y_[ilo]) {                           // cpp:392-394
            double yexp;
            const double* pe;
            if (speculate >= 4 && accepted) {
                pe = E;
                yexp = ycand[1];
            } else {
                TryPoint(psum_.data(), &simplex_[(size_t)ihi * n], 2.0, ptry_.data());
                if ((error = func->EvaluateBatch(1, ptry_.data(), n, &yexp))) return fmin;
                pe = ptry_.data();
            }
            func->Commit(pe, n, yexp);
            Accept(ihi, pe, yexp);
        } else if (ytry >= y_[inhi]) {                   // cpp:395-419
            const double ysave = y_[ihi];
            const double* pc;
            if (speculate >= 4 || (speculate >= 2 && !accepted)) {
                pc = accepted ? CA : CR;
                ytry = accepted ? ycand[2] : ycand[3];
            } else {
                TryPoint(psum_.data(), &simplex_[(size_t)ihi * n], 0.5, ptry_.data());
                if ((error = func->EvaluateBatch(1, ptry_.data(), n, &ytry))) return fmin;
                pc = ptry_.data();
            }
            func->Commit(pc, n, ytry);
            Accept(ihi, pc, ytry);
            if (ytry >= ysave) {
                // contract every vertex toward the best one (cpp:406-417)
                const double* lo = &simplex_[(size_t)ilo * n];
                int cnt = 0;
                for (int i = 0; i < nvertex; ++i) {
                    if (i == ilo) continue;
                    double* v = &simplex_[(size_t)i * n];
                    for (int j = 0; j < n; ++j) v[j] += lo[j];
                    for (int j = 0; j < n; ++j) v[j] *= 0.5;
                    std::memcpy(&shrink_pts[(size_t)cnt * n], v, sizeof(double) * n);
                    ++cnt;
                }
                if ((error = func->EvaluateBatch(cnt, shrink_pts.data(), n, shrink_y.data())))
                    return fmin;
                cnt = 0;
                for (int i = 0; i < nvertex; ++i) {
                    if (i == ilo) continue;
                    y_[i] = shrink_y[cnt];
                    func->Commit(&simplex_[(size_t)i * n], n, y_[i]);
                    ++cnt;
                }
                cycleCount += n;
                RecomputePsum();
            }
        } else {
            cycleCount--;                                // cpp:420-421
        }
    }
}

}  // namespace vb2
