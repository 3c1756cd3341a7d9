// line_search.cpp -- see line_search.h.
#include "line_search.h"

#include <cmath>
#include <utility>

namespace vb2 {

namespace {
// statgen/MathConstant.h:31-39
const double kTiny = 1.0e-30;
const int kItMax = 200;
const double kZeps = 3.0e-10;
const double kGold = 0.61803399;
const double kCGold = 0.38196601;
const double kMaxMagnification = 100.0;      // MathGold.cpp:22

inline double with_sign_of(double magnitude, double s) { return s >= 0 ? std::fabs(magnitude) : -std::fabs(magnitude); }
inline double larger(double x, double y) { return x > y ? x : y; }
}  // namespace

double BrentMinimizer::f(double x)
{
    double y = 0;
    if (error) return y;
    if ((error = func->EvaluateBatch(1, &x, &y))) return 0;
    func->Commit(x, y);
    return y;
}

// Walk downhill from (lo, hi) until three points a, b, c with f(b) below both ends are found.
void BrentMinimizer::Bracket(double lo, double hi)
{
    a = lo;
    b = hi;
    const double step = kGold + 1.0;
    if (speculate && !error) {
        // c = b' + step * (b' - a') for the ordered pair (a', b'): both orders' candidates at once
        double x[4] = {lo, hi, hi + step * (hi - lo), lo + step * (lo - hi)}, y[4];
        if ((error = func->EvaluateBatch(4, x, y))) return;
        func->Commit(x[0], y[0]);
        func->Commit(x[1], y[1]);
        fa = y[0];
        fb = y[1];
        const bool swapped = fb > fa;
        if (swapped) {
            std::swap(a, b);
            std::swap(fa, fb);
        }
        c = b + step * (b - a);
        fc = swapped ? y[3] : y[2];
        func->Commit(c, fc);
    } else {
        fa = f(a);
        fb = f(b);
        if (fb > fa) {
            std::swap(a, b);
            std::swap(fa, fb);
        }
        c = b + step * (b - a);
        fc = f(c);
    }
    while (fb > fc && !error) {
        // abscissa u of the vertex of the parabola through the three points
        const double r = (b - a) * (fb - fc);
        const double q = (b - c) * (fb - fa);
        double u = b - ((b - c) * q - (b - a) * r) / (2.0 * with_sign_of(larger(std::fabs(q - r), kTiny), q - r));
        const double ulim = b + kMaxMagnification * (c - b);
        double fu;
        if ((b - u) * (u - c) > 0.0) {                     // u between b and c
            fu = f(u);
            if (fu < fc) {                                 // minimum between b and c
                a = b; b = u;
                fa = fb; fb = fu;
                return;
            }
            if (fu > fb) {                                 // minimum between a and u
                c = u;
                fc = fu;
                return;
            }
            u = c + step * (c - b);                        // no use: default magnification
            fu = f(u);
        } else if ((c - u) * (u - ulim) > 0.0) {           // u between c and the limit
            fu = f(u);
            if (fu < fc) {
                b = c; c = u; u = c + step * (c - b);
                fb = fc; fc = fu; fu = f(u);
            }
        } else if ((u - ulim) * (ulim - c) >= 0.0) {       // beyond the limit: clamp
            u = ulim;
            fu = f(u);
        } else {                                           // reject the parabola
            u = c + step * (c - b);
            fu = f(u);
        }
        a = b; b = c; c = u;
        fa = fb; fb = fc; fc = fu;
    }
}

double BrentMinimizer::Brent(double tol)
{
    if (a > c) {
        std::swap(a, c);
        std::swap(fa, fc);
    }
    min = b;
    fmin = fb;
    double w = b, v = b, fw = fb, fv = fb;     // second best, previous second best
    double delta = 0.0;                        // step before last
    double d = 0.0;
    stuck = false;
    for (int iter = 1; iter <= kItMax && !error; ++iter) {
        const double middle = 0.5 * (a + c);
        const double tol1 = tol * std::fabs(min) + kZeps;
        const double tol2 = 2.0 * tol1;
        if (std::fabs(min - middle) <= (tol2 - 0.5 * (c - a))) return fmin;

        bool golden = true;
        if (std::fabs(delta) > tol1) {                     // try the parabola through min, w, v
            const double r = (min - w) * (fmin - fv);
            double q = (min - v) * (fmin - fw);
            double p = (min - v) * q - (min - w) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = std::fabs(q);
            const double before_last = delta;
            delta = d;
            if (!(std::fabs(p) >= std::fabs(0.5 * q * before_last) || p <= q * (a - min) || p >= q * (c - min))) {
                golden = false;
                d = p / q;
                const double u = min + d;
                if (u - a < tol2 || c - u < tol2) d = with_sign_of(tol1, middle - min);
            }
        }
        if (golden) {                                      // golden section into the larger part
            delta = min >= middle ? a - min : c - min;
            d = kCGold * delta;
        }
        const double u = std::fabs(d) >= tol1 ? min + d : min + with_sign_of(tol1, d);   // never closer than tol1
        const double fu = f(u);
        if (error) break;
        if (fu <= fmin) {
            if (u >= min) a = min;
            else c = min;
            v = w; w = min; min = u;
            fv = fw; fw = fmin; fmin = fu;
        } else {
            if (u < min) a = u;
            else c = u;
            if (fu <= fw || w == min) {
                v = w; w = u;
                fv = fw; fw = fu;
            } else if (fu <= fv || v == min || v == w) {
                v = u;
                fv = fu;
            }
        }
    }
    if (!error) stuck = true;
    return fmin;
}

}  // namespace vb2
