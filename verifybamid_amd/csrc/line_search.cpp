// line_search.cpp -- see line_search.h.
//
// Written around (abscissa, value) pairs; every floating-point expression is evaluated in the order
// the reference's ScalarMinimizer evaluates it, which is what makes the two visit the same
// abscissae (tests/test_abi_and_host.py compares them evaluation for evaluation).
#include "line_search.h"

#include <cmath>
#include <utility>

namespace vb2 {

namespace {

// statgen/MathConstant.h:31-39, MathGold.cpp:22
constexpr double kSmallest = 1.0e-30;
constexpr int kMaxIterations = 200;
constexpr double kAbsTolerance = 3.0e-10;
constexpr double kGoldenRatio = 0.61803399;
constexpr double kGoldenComplement = 0.38196601;
constexpr double kGrowthLimit = 100.0;

struct Sample {
    double x, f;
};

inline double magnitude_with_sign(double m, double sign_of) { return sign_of >= 0 ? std::fabs(m) : -std::fabs(m); }

// vertex of the parabola through three samples, guarded against a vanishing denominator
inline double parabola_vertex(const Sample& p, const Sample& q, const Sample& r)
{
    const double t1 = (q.x - p.x) * (q.f - r.f);
    const double t2 = (q.x - r.x) * (q.f - p.f);
    const double den = t2 - t1;
    const double guarded = std::fabs(den) > kSmallest ? std::fabs(den) : kSmallest;
    return q.x - ((q.x - r.x) * t2 - (q.x - p.x) * t1) / (2.0 * magnitude_with_sign(guarded, den));
}

}  // namespace

double BrentMinimizer::f(double x)
{
    double y = 0;
    if (error) return y;
    if ((error = func->EvaluateBatch(1, &x, &y))) return 0;
    func->Commit(x, y);
    return y;
}

// Downhill from (lo, hi) by golden-ratio steps, helped by parabolic extrapolation, until the
// middle one of three samples is the lowest.
void BrentMinimizer::Bracket(double lo, double hi)
{
    const double grow = kGoldenRatio + 1.0;
    Sample left{lo, 0}, mid{hi, 0}, right{0, 0};
    if (speculate && !error) {
        // the third sample lies beyond whichever of the first two is lower: both candidates at once
        double xs[4] = {lo, hi, hi + grow * (hi - lo), lo + grow * (lo - hi)}, ys[4];
        if ((error = func->EvaluateBatch(4, xs, ys))) return;
        func->Commit(xs[0], ys[0]);
        func->Commit(xs[1], ys[1]);
        left.f = ys[0];
        mid.f = ys[1];
        const bool uphill = mid.f > left.f;
        if (uphill) std::swap(left, mid);
        right.x = mid.x + grow * (mid.x - left.x);
        right.f = uphill ? ys[3] : ys[2];
        func->Commit(right.x, right.f);
    } else {
        left.f = f(left.x);
        mid.f = f(mid.x);
        if (mid.f > left.f) std::swap(left, mid);
        right.x = mid.x + grow * (mid.x - left.x);
        right.f = f(right.x);
    }
    bool done = false;
    while (!done && mid.f > right.f && !error) {
        Sample trial{parabola_vertex(left, mid, right), 0};
        const double farthest = mid.x + kGrowthLimit * (right.x - mid.x);
        enum { kInside, kBeyond, kAtLimit, kDefault } where;
        if ((mid.x - trial.x) * (trial.x - right.x) > 0.0) where = kInside;
        else if ((right.x - trial.x) * (trial.x - farthest) > 0.0) where = kBeyond;
        else if ((trial.x - farthest) * (farthest - right.x) >= 0.0) where = kAtLimit;
        else where = kDefault;
        switch (where) {
        case kInside:
            trial.f = f(trial.x);
            if (trial.f < right.f) {                       // a minimum between mid and right
                left = mid;
                mid = trial;
                done = true;
            } else if (trial.f > mid.f) {                  // a minimum between left and trial
                right = trial;
                done = true;
            } else {                                       // the fit was useless: a plain golden step
                trial.x = right.x + grow * (right.x - mid.x);
                trial.f = f(trial.x);
            }
            break;
        case kBeyond:
            trial.f = f(trial.x);
            if (trial.f < right.f) {                       // still going down: shift and step again
                mid = right;
                right = trial;
                trial.x = right.x + grow * (right.x - mid.x);
                trial.f = f(trial.x);
            }
            break;
        case kAtLimit:
            trial.x = farthest;
            trial.f = f(trial.x);
            break;
        case kDefault:
            trial.x = right.x + grow * (right.x - mid.x);
            trial.f = f(trial.x);
            break;
        }
        if (!done) {
            left = mid;
            mid = right;
            right = trial;
        }
    }
    a = left.x; fa = left.f;
    b = mid.x; fb = mid.f;
    c = right.x; fc = right.f;
}

double BrentMinimizer::Brent(double tol)
{
    double lower = a, upper = c;
    if (lower > upper) {
        std::swap(lower, upper);
        std::swap(fa, fc);
        std::swap(a, c);
    }
    Sample best{b, fb}, second = best, third = best;
    double last_step = 0.0;          // the step before the one just taken
    double step = 0.0;
    stuck = false;
    for (int round = 0; round < kMaxIterations && !error; ++round) {
        const double centre = 0.5 * (lower + upper);
        const double tol1 = tol * std::fabs(best.x) + kAbsTolerance;
        const double tol2 = 2.0 * tol1;
        if (std::fabs(best.x - centre) <= (tol2 - 0.5 * (upper - lower))) {
            a = lower; c = upper;
            min = best.x;
            return fmin = best.f;
        }
        bool interpolated = false;
        if (std::fabs(last_step) > tol1) {                 // inverse parabolic interpolation through the three best
            const double r = (best.x - second.x) * (best.f - third.f);
            double q = (best.x - third.x) * (best.f - second.f);
            double p = (best.x - third.x) * q - (best.x - second.x) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = std::fabs(q);
            const double two_back = last_step;
            last_step = step;
            const bool reject = std::fabs(p) >= std::fabs(0.5 * q * two_back) || p <= q * (lower - best.x) ||
                                p >= q * (upper - best.x);
            if (!reject) {
                interpolated = true;
                step = p / q;
                const double at = best.x + step;
                if (at - lower < tol2 || upper - at < tol2) step = magnitude_with_sign(tol1, centre - best.x);
            }
        }
        if (!interpolated) {                               // golden section into the larger segment
            last_step = best.x >= centre ? lower - best.x : upper - best.x;
            step = kGoldenComplement * last_step;
        }
        Sample trial;
        trial.x = std::fabs(step) >= tol1 ? best.x + step : best.x + magnitude_with_sign(tol1, step);   // never closer than tol1
        trial.f = f(trial.x);
        if (error) break;
        if (trial.f <= best.f) {
            (trial.x >= best.x ? lower : upper) = best.x;
            third = second;
            second = best;
            best = trial;
        } else {
            (trial.x < best.x ? lower : upper) = trial.x;
            if (trial.f <= second.f || second.x == best.x) {
                third = second;
                second = trial;
            } else if (trial.f <= third.f || third.x == best.x || third.x == second.x) {
                third = trial;
            }
        }
    }
    a = lower; c = upper;
    min = best.x;
    fmin = best.f;
    if (!error) stuck = true;
    return fmin;
}

}  // namespace vb2
