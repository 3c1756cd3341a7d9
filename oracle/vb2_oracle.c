/*
 * vb2_oracle.c -- TEST INFRASTRUCTURE ONLY (see vb2_oracle.h).
 *
 * Plain-C, FP64 restatement of the reference's contamination-likelihood path.
 * Operation ORDER follows the reference wherever it can change a rounding, so
 * that trajectories of the simplex search are reproducible bit for bit when
 * compiled without FMA contraction (-ffp-contract=off, see Makefile).
 */
#include "vb2_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* Constants                                                                  */
/* ------------------------------------------------------------------------- */

#define VB2O_NQ 94                 /* Phred 0..93  (ContaminationEstimator.h:65-74) */
static const double kMinAF = 0.00005;  /* h:94,107 */
static const double kMaxAF = 0.99995;  /* h:95,108 */

/* P(base class | genotype, error?)  -- h:164-177; index [err][geno][class],
 * class 0 = ref ('.' ','), 1 = alt, 2 = anything else. */
static const double kCondLK[2][3][3] = {
    {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.0}, {0.0, 1.0, 0.0}},
    {{0.0, 1.0 / 3.0, 2.0 / 3.0},
     {1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0},
     {1.0 / 3.0, 0.0, 2.0 / 3.0}},
};

/* h:65-74: pow(10, i / -10.0) for i = 0..93 */
static void phred_table(double *t)
{
    for (int i = 0; i < VB2O_NQ; ++i) t[i] = pow(10.0, i / -10.0);
}

/* h:180-184 */
static int classify_base(char base, char alt)
{
    if (base == '.' || base == ',') return 0;
    if (toupper((unsigned char)base) == toupper((unsigned char)alt)) return 1;
    return 2;
}

/* h:186-192 */
static void initial_gf(double af, double *gf)
{
    if (af < kMinAF) af = kMinAF;
    if (af > kMaxAF) af = kMaxAF;
    gf[0] = (1 - af) * (1 - af);
    gf[1] = 2 * (af) * (1 - af);
    gf[2] = af * af;
}

/* ------------------------------------------------------------------------- */
/* ComputeMixLLKs -- ContaminationEstimator.h:194-314                          */
/* ------------------------------------------------------------------------- */

double vb2o_compute_mix_llks(const vb2o_data *d, const double *pc1,
                             const double *pc2, double alpha, int num_thread)
{
    /* h:213-229: per-alpha table of log P(base | g1, g2) */
    double phred[VB2O_NQ];
    double tab[3][VB2O_NQ][3][3];
    const double one_minus_alpha = 1.0 - alpha;
    phred_table(phred);
    for (int bc = 0; bc < 3; ++bc) {
        for (int q = 0; q < VB2O_NQ; ++q) {
            const double p_err = phred[q];
            const double p_ok = 1.0 - p_err;
            for (int g1 = 0; g1 < 3; ++g1)
                for (int g2 = 0; g2 < 3; ++g2) {
                    double val =
                        (alpha * kCondLK[1][g1][bc] + one_minus_alpha * kCondLK[1][g2][bc]) * p_err +
                        (alpha * kCondLK[0][g1][bc] + one_minus_alpha * kCondLK[0][g2][bc]) * p_ok;
                    tab[bc][q][g1][g2] = log(val);
                }
        }
    }

    const int k = d->num_pc;
    double sum_llk = 0;
    (void)num_thread;
#ifdef _OPENMP
    /* h:232-235 */
    omp_set_num_threads(num_thread > 0 ? num_thread : 1);
#pragma omp parallel for reduction(+ : sum_llk)
#endif
    for (int32_t i = 0; i < d->num_marker; ++i) {
        const int32_t idx = d->base_info_index[i];
        if (idx < 0) continue;                                    /* h:239 */
        const int64_t beg = d->site_off[idx];
        const int64_t depth64 = d->site_off[idx + 1] - beg;
        if (depth64 == 0) continue;                               /* h:244 */
        if (!d->sanity_disabled &&                                /* h:246-249 */
            ((double)depth64 < (d->avg_depth - 3 * d->sd_depth) ||
             (double)depth64 > (d->avg_depth + 3 * d->sd_depth)))
            continue;

        double af1, af2;                                          /* h:251-267 */
        if (d->af_known) {
            af1 = af2 = d->known_af[i];
        } else {
            af1 = 0.;
            for (int kk = 0; kk < k; ++kk) af1 += d->ud[(size_t)i * k + kk] * pc1[kk];
            af1 += d->means[i];
            af1 /= 2.0;
            af2 = 0.;
            for (int kk = 0; kk < k; ++kk) af2 += d->ud[(size_t)i * k + kk] * pc2[kk];
            af2 += d->means[i];
            af2 /= 2.0;
        }
        double gf[3], gf2[3];
        initial_gf(af1, gf);                                      /* h:273-274 */
        initial_gf(af2, gf2);

        const char alt = d->alt_base[i];
        const int depth = (int)depth64;
        double acc[3][3] = {{0}};
        for (int j = 0; j < depth; ++j) {                         /* h:288-303 */
            const int bc = classify_base(d->bases[beg + j], alt);
            int q = (int)(unsigned char)d->quals[beg + j] - 33;
            if (q < 0) q = 0;
            else if (q > 93) q = 93;
            for (int g1 = 0; g1 < 3; ++g1)
                for (int g2 = 0; g2 < 3; ++g2) acc[g1][g2] += tab[bc][q][g1][g2];
        }
        double marker_lk = 0;                                     /* h:307-311 */
        for (int g1 = 0; g1 < 3; ++g1)
            for (int g2 = 0; g2 < 3; ++g2)
                marker_lk += exp(acc[g1][g2]) * gf[g1] * gf2[g2];
        if (marker_lk > 0) sum_llk += log(marker_lk);
    }
    return sum_llk;
}

/* ------------------------------------------------------------------------- */
/* Nelder-Mead -- MathGenMin.cpp:313-443                                       */
/* ------------------------------------------------------------------------- */

#define VB2O_ZEPS 3.0e-10   /* statgen/MathConstant.h:34 */
#define VB2O_FPMAX 1.0e+100 /* statgen/MathConstant.h:36 */
#define VB2O_CYCLE_MAX 50000 /* MathGenMin.cpp:314 */

typedef struct amoeba_state {
    vb2o_objective f;
    void *user;
    int n;
    double *simplex; /* (n+1) x n */
    double *y;       /* n+1 */
    double *psum;    /* n */
    double *ptry;    /* n */
} amoeba_state;

/* MathGenMin.cpp:425-443; SetMultiple/AddMultiple are two separate passes
 * (statgen/MathVector.cpp:134-176). */
static double amoeba_try(amoeba_state *s, int ihi, double factor)
{
    const int n = s->n;
    double *hi = s->simplex + (size_t)ihi * n;
    const double fac = (1.0 - factor) / n;
    for (int i = 0; i < n; ++i) s->ptry[i] = fac * s->psum[i];
    for (int i = 0; i < n; ++i) s->ptry[i] += (factor - fac) * hi[i];
    const double ytry = s->f(s->user, s->ptry, n);
    if (ytry < s->y[ihi]) {
        s->y[ihi] = ytry;
        for (int i = 0; i < n; ++i) s->psum[i] -= hi[i];
        for (int i = 0; i < n; ++i) hi[i] = s->ptry[i];
        for (int i = 0; i < n; ++i) s->psum[i] += hi[i];
    }
    return ytry;
}

static void amoeba_psum(amoeba_state *s)
{
    const int n = s->n;
    for (int i = 0; i < n; ++i) s->psum[i] = s->simplex[i];
    for (int m = 1; m <= n; ++m)
        for (int i = 0; i < n; ++i) s->psum[i] += s->simplex[(size_t)m * n + i];
}

double vb2o_amoeba_minimize(vb2o_objective f, void *user, int n, double *point,
                            double ftol)
{
    if (n == 0) return f(user, point, 0);                         /* cpp:331-332 */

    const int nvertex = n + 1;
    amoeba_state s;
    s.f = f;
    s.user = user;
    s.n = n;
    s.simplex = (double *)malloc(sizeof(double) * (size_t)nvertex * n);
    s.y = (double *)malloc(sizeof(double) * nvertex);
    s.psum = (double *)malloc(sizeof(double) * n);
    s.ptry = (double *)malloc(sizeof(double) * n);
    double fmin = VB2O_FPMAX;                                     /* cpp:24 */
    double ret;

    /* cpp:335-345: vertex i = point + e_i (directions = identity * 1.0) */
    for (int i = 0; i < n; ++i) {
        double *v = s.simplex + (size_t)i * n;
        for (int j = 0; j < n; ++j) v[j] = point[j] + (i == j ? 1.0 : 0.0);
        s.y[i] = f(user, v, n);
        if (s.y[i] < fmin) fmin = s.y[i];
    }
    memcpy(s.simplex + (size_t)n * n, point, sizeof(double) * n);
    s.y[n] = f(user, s.simplex + (size_t)n * n, n);
    if (s.y[n] < fmin) fmin = s.y[n];

    long cycle_count = nvertex;
    amoeba_psum(&s);

    for (;;) {
        int ilo, ihi, inhi;                                       /* cpp:357-370 */
        if (s.y[0] > s.y[1]) { ilo = inhi = 1; ihi = 0; }
        else                 { ilo = inhi = 0; ihi = 1; }
        for (int i = 2; i < nvertex; ++i) {
            if (s.y[i] <= s.y[ilo]) ilo = i;
            else if (s.y[i] > s.y[ihi]) { inhi = ihi; ihi = i; }
            else if (s.y[i] > s.y[inhi]) inhi = i;
        }
        /* cpp:373-378 */
        const double rtol = 2 * fabs(s.y[ihi] - s.y[ilo]) /
                            (fabs(s.y[ihi]) + fabs(s.y[ilo]) + VB2O_ZEPS);
        if (rtol < ftol) {
            memcpy(point, s.simplex + (size_t)ilo * n, sizeof(double) * n);
            ret = s.y[ilo];
            break;
        }
        if (cycle_count > VB2O_CYCLE_MAX) {                       /* cpp:380-383 */
            ret = DBL_MAX;
            break;
        }
        cycle_count += 2;                                         /* cpp:389-421 */
        double ytry = amoeba_try(&s, ihi, -1.0);
        if (ytry <= s.y[ilo]) {
            amoeba_try(&s, ihi, 2.0);
        } else if (ytry >= s.y[inhi]) {
            const double ysave = s.y[ihi];
            ytry = amoeba_try(&s, ihi, 0.5);
            if (ytry >= ysave) {
                const double *lo = s.simplex + (size_t)ilo * n;
                for (int i = 0; i < nvertex; ++i) {
                    if (i == ilo) continue;
                    double *v = s.simplex + (size_t)i * n;
                    for (int j = 0; j < n; ++j) v[j] += lo[j];
                    for (int j = 0; j < n; ++j) v[j] *= 0.5;
                    s.y[i] = f(user, v, n);
                }
                cycle_count += n;
                amoeba_psum(&s);
            }
        } else {
            cycle_count--;
        }
    }
    free(s.simplex);
    free(s.y);
    free(s.psum);
    free(s.ptry);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* FullLLKFunc::Initialize/Evaluate/CalculateLLK0 + OptimizeLLK                */
/* ------------------------------------------------------------------------- */

#define VB2O_MAX_PC 64

typedef struct est_state {
    const vb2o_data *d;
    int k;
    int num_thread;
    /* ContaminationEstimator members (h:42-53, 451-453) */
    int is_heter, is_pc_fixed, is_alpha_fixed;
    double alpha;
    double pc[2][VB2O_MAX_PC];
    /* FullLLKFunc members (h:78-88) */
    double fix_pc[VB2O_MAX_PC], fix_pc2[VB2O_MAX_PC], fix_alpha;
    double g_pc[VB2O_MAX_PC], g_pc2[VB2O_MAX_PC], g_alpha;
    double llk1, llk0;
    int64_t num_eval;
    vb2o_trace *trace;
    int verbose;
} est_state;

static double est_llk(est_state *e, const double *pc1, const double *pc2, double alpha)
{
    const double v = vb2o_compute_mix_llks(e->d, pc1, pc2, alpha, e->num_thread);
    vb2o_trace *t = e->trace;
    if (t) {
        if (t->count < t->capacity) {
            const int64_t r = t->count;
            t->alpha[r] = alpha;
            t->llk[r] = v;
            memcpy(t->pc1 + r * e->k, pc1, sizeof(double) * e->k);
            memcpy(t->pc2 + r * e->k, pc2, sizeof(double) * e->k);
        }
        t->count++;
    }
    e->num_eval++;
    return v;
}

static double inv_logit(double x) { const double ex = exp(x); return ex / (1. + ex); } /* h:119-122 */
static double logit(double x) { return log(x / (1. - x)); }                          /* h:124-127 */

/* h:339-442 */
static double est_evaluate(void *user, const double *v, int n)
{
    est_state *e = (est_state *)user;
    const int k = e->k;
    double sm;
    if (!e->is_heter) {
        if (e->is_pc_fixed) {                                     /* h:342-348 */
            const double a = inv_logit(v[0]);
            sm = 0 - est_llk(e, e->fix_pc, e->fix_pc2, a);
            if (sm < e->llk1) { e->llk1 = sm; e->g_alpha = a; }
        } else if (e->is_alpha_fixed) {                           /* h:349-360 */
            sm = 0 - est_llk(e, v, v, e->fix_alpha);
            if (sm < e->llk1) {
                e->llk1 = sm;
                memcpy(e->g_pc, v, sizeof(double) * k);
                memcpy(e->g_pc2, v, sizeof(double) * k);
            }
        } else {                                                  /* h:361-374 */
            const double a = inv_logit(v[k]);
            sm = 0 - est_llk(e, v, v, a);
            if (sm < e->llk1) {
                e->llk1 = sm;
                memcpy(e->g_pc, v, sizeof(double) * k);
                memcpy(e->g_pc2, v, sizeof(double) * k);
                e->g_alpha = a;
            }
        }
    } else {
        if (e->is_pc_fixed) {                                     /* h:377-389 */
            const double a = inv_logit(v[k]);
            sm = 0 - est_llk(e, v, e->fix_pc2, a);
            if (sm < e->llk1) {
                e->llk1 = sm;
                memcpy(e->g_pc, v, sizeof(double) * k);
                e->g_alpha = a;
            }
        } else if (e->is_alpha_fixed) {                           /* h:390-409 */
            if (n > 2 * k) abort();
            double p1[VB2O_MAX_PC] = {0}, p2[VB2O_MAX_PC] = {0};
            for (int i = 0; i < n; ++i) {
                if (i < k) p1[i] = v[i];
                else p2[i - k] = v[i];
            }
            sm = 0 - est_llk(e, p1, p2, e->fix_alpha);
            if (sm < e->llk1) {
                e->llk1 = sm;
                memcpy(e->g_pc, p1, sizeof(double) * k);
                memcpy(e->g_pc2, p2, sizeof(double) * k);
            }
        } else {                                                  /* h:410-433 */
            if (n > 2 * k + 1) abort();
            double p1[VB2O_MAX_PC] = {0}, p2[VB2O_MAX_PC] = {0}, a = 0.;
            for (int i = 0; i < n; ++i) {
                if (i < k) p1[i] = v[i];
                else if (i < 2 * k) p2[i - k] = v[i];
                else a = inv_logit(v[i]);
            }
            sm = 0 - est_llk(e, p1, p2, a);
            if (sm < e->llk1) {
                e->llk1 = sm;
                memcpy(e->g_pc, p1, sizeof(double) * k);
                memcpy(e->g_pc2, p2, sizeof(double) * k);
                e->g_alpha = a;
            }
        }
    }
    /* h:435-440: `notice(...)` of libStatGen (statgen/Error.cpp:70-79) = "NOTICE - " + the formatted text + '\n' on
     * stderr; the reference indexes globalPC[0..1] whatever --NumPC is (k = 1 would read past the vector there: 0 here) */
    if (e->verbose)
        fprintf(stderr, "NOTICE - ContaminatingSamplePC1:%f\tContaminatingSamplePC2:%f\tIntendedSamplePC1:%f\t"
                        "IntendedSamplePC2:%f\tFREEMIX(Alpha):%f\tllk:%f\n",
                e->g_pc[0], k > 1 ? e->g_pc[1] : 0.0, e->g_pc2[0], k > 1 ? e->g_pc2[1] : 0.0, e->g_alpha, e->llk1);
    return sm;
}

/* ContaminationEstimator.cpp:265-289 (also OptimizeHeterFixedPC, cpp:261-263) */
static int opt_homo(est_state *e, vb2o_minimizer mini, double eps)
{
    const int k = e->k;
    double p[VB2O_MAX_PC + 1];
    for (int i = 0; i < k; ++i) p[i] = e->pc[0][i];
    p[k] = logit(e->alpha);
    const double ret = mini(est_evaluate, e, k + 1, p, eps);
    e->alpha = inv_logit(p[k]);
    for (int i = 0; i < k; ++i) e->pc[0][i] = p[i];
    return ret != DBL_MAX;
}

/* cpp:192-226 */
static int opt_heter(est_state *e, vb2o_minimizer mini, double eps)
{
    const int k = e->k;
    double p[2 * VB2O_MAX_PC + 1];
    for (int i = 0; i < k; ++i) p[i] = e->pc[0][i];
    for (int i = 0; i < k; ++i) p[k + i] = e->pc[1][i];
    p[2 * k] = logit(e->alpha);
    const double ret = mini(est_evaluate, e, 2 * k + 1, p, eps);
    e->alpha = inv_logit(p[2 * k]);
    for (int i = 0; i < k; ++i) e->pc[0][i] = p[i];
    for (int i = 0; i < k; ++i) e->pc[1][i] = p[k + i];
    return ret != DBL_MAX;
}

/* cpp:291-313 */
static int opt_homo_fixed_alpha(est_state *e, vb2o_minimizer mini, double eps)
{
    const int k = e->k;
    double p[VB2O_MAX_PC];
    for (int i = 0; i < k; ++i) p[i] = e->pc[0][i];
    const double ret = mini(est_evaluate, e, k, p, eps);
    for (int i = 0; i < k; ++i) e->pc[0][i] = p[i];
    return ret != DBL_MAX;
}

/* cpp:228-259 */
static int opt_heter_fixed_alpha(est_state *e, vb2o_minimizer mini, double eps)
{
    const int k = e->k;
    double p[2 * VB2O_MAX_PC];
    for (int i = 0; i < k; ++i) p[i] = e->pc[0][i];
    for (int i = 0; i < k; ++i) p[k + i] = e->pc[1][i];
    const double ret = mini(est_evaluate, e, 2 * k, p, eps);
    for (int i = 0; i < k; ++i) e->pc[0][i] = p[i];
    for (int i = 0; i < k; ++i) e->pc[1][i] = p[k + i];
    return ret != DBL_MAX;
}

/* cpp:315-332 */
static int opt_homo_fixed_pc(est_state *e, vb2o_minimizer mini, double eps)
{
    double p[1];
    p[0] = logit(e->alpha);
    const double ret = mini(est_evaluate, e, 1, p, eps);
    e->alpha = inv_logit(p[0]);
    return ret != DBL_MAX;
}

int vb2o_optimize_llk(const vb2o_data *d, const vb2o_options *opt,
                      vb2o_minimizer mini, vb2o_trace *trace, double *pc_out,
                      double *pc2_out, vb2o_result *res)
{
    const int k = d->num_pc;
    if (k < 1 || k > VB2O_MAX_PC) return -1;
    if (!mini) mini = vb2o_amoeba_minimize;
    est_state *e = (est_state *)calloc(1, sizeof(est_state));
    e->d = d;
    e->k = k;
    e->num_thread = opt->num_thread;
    e->verbose = opt->verbose;
    e->trace = trace;
    if (trace) trace->count = 0;
    /* ctor, cpp:38-51 + main.cpp:287-319 */
    e->is_heter = opt->is_heter;
    e->is_pc_fixed = opt->is_pc_fixed;
    e->is_alpha_fixed = opt->is_alpha_fixed;
    e->alpha = 0.5;
    if (opt->is_pc_fixed && opt->fix_pc)
        for (int i = 0; i < k; ++i) e->pc[1][i] = opt->fix_pc[i];
    else if (opt->is_alpha_fixed)
        e->alpha = opt->fix_alpha;
    if (d->af_known) { e->is_pc_fixed = 1; e->is_heter = 0; }    /* main.cpp:314-319 */

    /* FullLLKFunc::Initialize, h:316-332 */
    for (int i = 0; i < k; ++i)
        e->g_pc[i] = e->fix_pc[i] = e->g_pc2[i] = e->fix_pc2[i] = e->pc[1][i];
    e->g_alpha = e->fix_alpha = e->alpha;
    e->llk1 = 0 - est_llk(e, e->fix_pc, e->fix_pc2, e->fix_alpha);
    for (int i = 0; i < k; ++i) e->pc[0][i] = 0.01;
    for (int i = 0; i < k; ++i) e->pc[1][i] = 0.01;
    e->alpha = 0.03;

    int ok = 1;
    const double eps = opt->epsilon;
    if (!e->is_heter) {                                           /* cpp:98-110 */
        if (e->is_pc_fixed) ok &= opt_homo_fixed_pc(e, mini, eps);
        else if (e->is_alpha_fixed) ok &= opt_homo_fixed_alpha(e, mini, eps);
        else ok &= opt_homo(e, mini, eps);
    } else {                                                      /* cpp:111-150 */
        if (e->is_pc_fixed) {
            ok &= opt_homo(e, mini, eps);
        } else if (e->is_alpha_fixed) {
            e->is_heter = 0;
            ok &= opt_homo_fixed_alpha(e, mini, eps);
            for (int i = 0; i < k; ++i) e->pc[1][i] = e->pc[0][i];
            for (int i = 0; i < k; ++i) e->g_pc2[i] = e->g_pc[i];
            e->is_heter = 1;
            ok &= opt_heter_fixed_alpha(e, mini, eps);
        } else {
            e->is_heter = 0;
            ok &= opt_homo(e, mini, eps);
            for (int i = 0; i < k; ++i) e->pc[1][i] = e->pc[0][i];
            for (int i = 0; i < k; ++i) e->g_pc2[i] = e->g_pc[i];
            e->is_heter = 1;
            ok &= opt_heter(e, mini, eps);
        }
        if (e->g_alpha >= 0.5) {                                  /* cpp:146-149 */
            /* indices 0 and 1 are hard-coded in the reference; index 1 only
             * exists when k >= 2 */
            double t = e->g_pc[0]; e->g_pc[0] = e->g_pc2[0]; e->g_pc2[0] = t;
            if (k >= 2) { t = e->g_pc[1]; e->g_pc[1] = e->g_pc2[1]; e->g_pc2[1] = t; }
        }
    }
    /* CalculateLLK0, h:334-337 */
    e->llk0 = 0 - est_llk(e, e->g_pc, e->g_pc, 0);

    memcpy(pc_out, e->g_pc, sizeof(double) * k);
    memcpy(pc2_out, e->g_pc2, sizeof(double) * k);
    res->alpha = e->g_alpha;
    res->llk1 = e->llk1;
    res->llk0 = e->llk0;
    res->num_eval = e->num_eval;
    res->converged = ok;
    free(e);
    return 0;
}
