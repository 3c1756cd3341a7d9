/*
 * vb2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64) of the VerifyBamID2 contamination-likelihood
 * hot path, used as the parity checker for the HIP implementation.  Nothing in
 * the shipped product (verifybamid_amd/, include/) may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity status: PINNED -- see
 * tests/test_oracle_golden.py (six golden .Ancestry files and two .selfSM
 * files of the reference's own CTest suite, the known-answer LLK values of
 * SURVEY.md section 8c, and bit-exact agreement of the simplex search with the
 * reference's own AmoebaMinimizer compiled in place as oracle/_ref).
 */
#ifndef VB2_ORACLE_H_
#define VB2_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The data ComputeMixLLKs reaches through its back-pointer `ptr`
 * (ContaminationEstimator.h:82, 445-475), flattened to plain arrays but keeping
 * the reference's shapes: panel-ordered markers that point into per-site
 * base/qual character vectors. */
typedef struct vb2o_data {
    int32_t num_marker;            /* NumMarker (ContaminationEstimator.h:446)        */
    int32_t num_pc;                /* numPC                                           */
    const double *ud;              /* UD[i][k], row-major M x k (h:452)               */
    const double *means;           /* means[i]  (h:454)                               */
    const int32_t *base_info_index;/* resolvedMarkers[i].baseInfoIndex, -1 = absent   */
    const char *alt_base;          /* resolvedMarkers[i].altBase                      */
    const double *known_af;        /* resolvedMarkers[i].knownAFValue, or NULL        */
    int32_t num_site;              /* viewer.baseInfo.size()                          */
    const int64_t *site_off;       /* num_site+1 offsets into bases/quals             */
    const char *bases;             /* viewer.baseInfo  concatenated                   */
    const char *quals;             /* viewer.qualInfo  concatenated (ASCII, q+33)     */
    double avg_depth;              /* viewer.avgDepth                                 */
    double sd_depth;               /* viewer.sdDepth                                  */
    int32_t sanity_disabled;       /* isSanityCheckDisabled                           */
    int32_t af_known;              /* isAFknown                                       */
} vb2o_data;

/* ContaminationEstimator.h:194-314.  Returns +LLK (not negated). */
double vb2o_compute_mix_llks(const vb2o_data *d, const double *pc1,
                             const double *pc2, double alpha, int num_thread);

/* Generic objective and minimiser seams (statgen/MathVector.h:281-308,
 * MathGenMin.h:13-35). */
typedef double (*vb2o_objective)(void *user, const double *v, int n);
/* point: in = start, out = final point (untouched on non-convergence).
 * Returns fmin, or DBL_MAX when cycleMax was exceeded. */
typedef double (*vb2o_minimizer)(vb2o_objective f, void *user, int n,
                                 double *point, double ftol);

/* MathGenMin.cpp:313-443 (AmoebaMinimizer::Reset/Minimize/Amoeba). */
double vb2o_amoeba_minimize(vb2o_objective f, void *user, int n, double *point,
                            double ftol);

/* Model selection, as set by main.cpp:283-319. */
typedef struct vb2o_options {
    int32_t is_heter;        /* !--WithinAncestry                                    */
    int32_t is_pc_fixed;     /* --FixPC (or --KnownAF)                               */
    int32_t is_alpha_fixed;  /* --FixAlpha                                           */
    double  fix_alpha;       /* --FixAlpha value (Estimator.alpha, main.cpp:311)     */
    const double *fix_pc;    /* --FixPC values -> PC[1] (main.cpp:304-306), or NULL  */
    double  epsilon;         /* --Epsilon, default 1e-8 (main.cpp:76)                */
    int32_t num_thread;      /* --NumThread                                          */
    int32_t verbose;         /* --Verbose: FullLLKFunc::Evaluate's notice per call (h:435-440) on stderr */
} vb2o_options;

/* One record per ComputeMixLLKs call, in call order. */
typedef struct vb2o_trace {
    int64_t capacity;        /* records the buffers can hold                         */
    int64_t count;           /* records written (may exceed capacity: then truncated)*/
    double *alpha;           /* [capacity]                                           */
    double *pc1;             /* [capacity * num_pc]                                  */
    double *pc2;             /* [capacity * num_pc]                                  */
    double *llk;             /* [capacity]  (+LLK)                                   */
} vb2o_trace;

typedef struct vb2o_result {
    double alpha;            /* fn.globalAlpha                                       */
    double llk1;             /* fn.llk1  (= -LLK at the best point)                  */
    double llk0;             /* fn.llk0                                              */
    int64_t num_eval;        /* ComputeMixLLKs calls                                 */
    int32_t converged;       /* 0 if any Minimize hit cycleMax                       */
} vb2o_result;

/* ContaminationEstimator.cpp:88-190 (+ the six Optimize* wrappers 192-332 and
 * FullLLKFunc::Initialize/Evaluate/CalculateLLK0, h:316-442), without the
 * file/stdout output.  pc_out/pc2_out receive fn.globalPC/globalPC2 (num_pc
 * each).  minimizer = NULL selects vb2o_amoeba_minimize. */
int vb2o_optimize_llk(const vb2o_data *d, const vb2o_options *opt,
                      vb2o_minimizer minimizer, vb2o_trace *trace,
                      double *pc_out, double *pc2_out, vb2o_result *res);

#ifdef __cplusplus
}
#endif
#endif
