/*
 * vb2_abi.h -- C-ABI of the MI355X-native contamination-likelihood core.
 *
 * This is the drop-in boundary for the one hot path of VerifyBamID2
 * (Griffan/VerifyBamID): the genotype-mixture log-likelihood
 * FullLLKFunc::ComputeMixLLKs and the Nelder-Mead search that drives it.
 * The reference has no FFI layer; its seam for this path is the libStatGen
 * functor  VectorFunc::Evaluate(Vector&)  (statgen/MathVector.h:281-308),
 * installed with  myMinimizer.func = &fn  (ContaminationEstimator.cpp:212,246,
 * 279,304,325), and one level below it the pure-compute member
 *   double ComputeMixLLKs(const std::vector<double>& pc1,
 *                         const std::vector<double>& pc2, double alpha)
 * (ContaminationEstimator.h:194-195).  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every function
 * returns 0 on success and a negative vb2_status otherwise and never throws;
 * vb2_last_error() gives the message for the calling thread.  A context is
 * thread-compatible (one thread at a time), like the reference's estimator.
 * All arithmetic on the path is IEEE-754 binary64.
 *
 * There is NO CPU fallback: every compute entry point fails with
 * VB2_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef VB2_ABI_H_
#define VB2_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VB2_ABI_VERSION 7

typedef enum vb2_status {
    VB2_OK = 0,
    VB2_ERR_INVALID = -1,    /* bad argument                                     */
    VB2_ERR_NO_DEVICE = -2,  /* no usable HIP device / kernel image not loadable */
    VB2_ERR_HIP = -3,        /* a HIP runtime call failed                        */
    VB2_ERR_IO = -4,         /* file could not be read / written                 */
    VB2_ERR_NOMEM = -5,
    VB2_ERR_SANITY = -6      /* marker sanity check failed (main.cpp:371-379)    */
} vb2_status;

typedef struct vb2_ctx vb2_ctx;

/* ------------------------------------------------------------------------- *
 * 1. Likelihood context: the data ComputeMixLLKs reaches through `ptr`
 *    (ContaminationEstimator.h:82, 236-276), handed over ONCE.
 *
 *    Markers are in panel order (= row order of .UD/.mu/.bed).  read_off has
 *    num_marker+1 entries; marker i owns bases/quals[read_off[i]..read_off[i+1]).
 *    An empty range means "marker absent from the pileup" (baseInfoIndex < 0,
 *    h:239) or depth 0 (h:244).  bases are the pileup characters
 *    ". , A C G T N a c g t n" and quals are ASCII Phred+33, exactly as
 *    SimplePileupViewer stores them (SimplePileupViewer.h:60-61,94-95).
 *    All arrays are caller-owned host memory and may be freed after
 *    vb2_ctx_create returns.
 * ------------------------------------------------------------------------- */
typedef struct vb2_input {
    int32_t num_marker;        /* NumMarker                         (h:446)        */
    int32_t num_pc;            /* numPC                             (h:49)         */
    const double *ud;          /* UD[i][k], row-major M x num_pc    (h:452)        */
    const double *means;       /* means[i]                          (h:454)        */
    const int64_t *read_off;   /* [M+1]                                            */
    const char *bases;         /* viewer.baseInfo, concatenated in panel order     */
    const char *quals;         /* viewer.qualInfo                                  */
    const char *alt_base;      /* resolvedMarkers[i].altBase        (h:472)        */
    const double *known_af;    /* resolvedMarkers[i].knownAFValue or NULL (h:473)  */
    double avg_depth;          /* viewer.avgDepth                                  */
    double sd_depth;           /* viewer.sdDepth                                   */
    int32_t sanity_disabled;   /* isSanityCheckDisabled: skips the +-3sd filter of h:246-249 */
    int32_t reserved;
} vb2_input;

typedef struct vb2_options {
    int32_t device;            /* HIP device ordinal, -1 = current device          */
    int32_t flags;             /* VB2_OPT_* bits                                   */
#define VB2_OPT_COHORT_LAYOUT 1   /* also keep the 16-bit run lists that vb2_batch_* steps stream (half the bytes
                                   * per sample and step; +28 % device memory).  Without it vb2_batch_create
                                   * builds them on first use. */
/* (ABI 6) how vb2_ctx_optimize_llk runs on this context -- the defaults are the fast way; results are the same bits:   */
#define VB2_OPT_PLAIN_LAUNCH 2    /* the resident search kernel goes up with a plain launch instead of a cooperative one (the
                                   * library does this by itself under rocprofv3, whose exit handler does not survive a
                                   * cooperative launch in ROCm 7.2)                                                     */
#define VB2_OPT_HOST_SEARCH 4     /* the simplex decisions stay on the host (the resident kernel only evaluates)        */
#define VB2_OPT_LAUNCH_PER_STEP 8 /* no resident kernel at all: one launch per search step                               */
    void *stream;              /* hipStream_t to run on; NULL = context-owned      */
} vb2_options;

typedef struct vb2_info {
    int32_t abi_version;
    int32_t device;
    int32_t num_marker;        /* panel markers                                    */
    int32_t num_pc;
    int64_t num_active_marker; /* markers that survive h:239-249                   */
    int64_t num_read;          /* bases of the active markers                      */
    int64_t num_read_other;    /* of those, class "other" (folded into a constant) */
    int32_t num_code;          /* distinct (class, quality) pairs in the data      */
    int32_t num_tile;          /* 16-marker micro-tiles                            */
    int64_t device_bytes;      /* HBM held by the context                          */
    /* SURVEY 8(d) algorithmic bytes of ONE evaluation: 2*R + M_active*(8k+12) */
    int64_t algorithmic_bytes_per_eval;
    char device_name[64];
    char arch[32];
    /* (ABI 4) HBM bytes ONE lock-step cohort step (vb2_batch_*) reads of this sample: run lists (16-bit
     * once built, see VB2_OPT_COHORT_LAYOUT) + tile records + panel rows + per-marker constants */
    int64_t cohort_step_bytes;
    /* (ABI 7) how the sample is laid out in HBM: 1 = probability domain -- the per-alpha table holds the powers P^n of
     * P(read | genotype pair, alpha) per quality and a marker's likelihoods are PRODUCTS of table rows, one step per row: no
     * exp() per genotype pair (ContaminationEstimator.h:288-311 sums logarithms and exponentiates) -- taken when no marker's
     * products can leave the normal double range; 0 = run words and sums of logarithms (any depth, quality 0).  The
     * values agree to rounding either way.  num_table_row: rows of the per-alpha table. */
    int32_t layout;
    int32_t num_table_row;
    /* read-loop steps per evaluation point, summed over the markers, the tiles' padding included: each reads one 48-byte
     * table row from LDS (the LDS side of bench.py's roofline) */
    int64_t num_step;
} vb2_info;

/* Builds the device-resident SoA form of the input (classification, quality
 * clamping, marker filtering and the alpha-independent partial sums happen
 * here, once -- on the device: the arrays of `in` go up as they are, through
 * one pinned staging copy and hipMemcpyAsync on the context's stream, and
 * kernels classify, run-length code and pack them; the host keeps the
 * dictionary and the sort of the markers).  Replaces BuildResolvedMarkers + the
 * per-call prologue of ComputeMixLLKs (ContaminationEstimator.cpp:67-86;
 * h:236-249, 285-299).  `in` is not referenced after the call returns.
 * Limits (VB2_ERR_INVALID beyond them; the reference has none short of memory): 2^29 - 1024 markers and 2^25 rows of
 * run words -- a row is the k-th and (k+1)-th distinct (class, quality) pairs of 16 depth-sorted markers, so the second
 * is about a billion reads of one sample: the kernels address both by 32-bit byte offsets. */
int vb2_ctx_create(const vb2_input *in, const vb2_options *opt, vb2_ctx **out);
void vb2_ctx_destroy(vb2_ctx *ctx);
int vb2_ctx_info(const vb2_ctx *ctx, vb2_info *info);

/* Replaces  ComputeMixLLKs(pc1, pc2, alpha)  (ContaminationEstimator.h:194-314)
 * for B parameter points at once.  pc1/pc2 are B x num_pc row-major, alpha has
 * B entries, llk_out receives B values of +LLK (not negated, like the
 * reference's return value).  Synchronous; host pointers. */
int vb2_llk_eval_batch(vb2_ctx *ctx, int32_t num_point, const double *pc1,
                       const double *pc2, const double *alpha, double *llk_out);

/* Same evaluation with DEVICE pointers, enqueued on `stream` (NULL = the
 * context's stream) without any host synchronisation:
 *   d_points : num_point x (2*num_pc+1) doubles, each row = pc1[0..k) pc2[0..k) alpha
 *   d_llk_out: num_point doubles
 * Used by the multi-GPU path (the caller all-reduces d_llk_out over RCCL) and by
 * the bench. */
int vb2_llk_eval_batch_device(vb2_ctx *ctx, int32_t num_point, const double *d_points,
                              double *d_llk_out, void *stream);

/* Brackets a series of dependent vb2_llk_eval_batch calls on one context -- a search driven by
 * the caller's own optimiser, e.g. the reference's AmoebaMinimizer calling Evaluate once per
 * point (MathGenMin.cpp:389-421).  Between begin and end the evaluations are served by a kernel
 * that stays resident on the device and is fed through pinned host memory, which
 * saves the launch latency of every call (~7 us of ~22 us for one point at 100 k markers).
 * Results are identical to unbracketed calls.  begin never fails the search: when the mode is
 * unavailable it returns VB2_OK and the calls simply launch kernels as usual.  Only
 * vb2_llk_eval_batch may be called on the context between the two; vb2_ctx_optimize_llk does this
 * bracketing itself. */
int vb2_ctx_search_begin(vb2_ctx *ctx);
void vb2_ctx_search_end(vb2_ctx *ctx);

/* ------------------------------------------------------------------------- *
 * 2. Estimator: FullLLKFunc::Initialize/Evaluate/CalculateLLK0 and
 *    ContaminationEstimator::OptimizeLLK with its six Optimize* wrappers
 *    (ContaminationEstimator.h:316-442, ContaminationEstimator.cpp:88-332),
 *    driving AmoebaMinimizer (MathGenMin.cpp:313-443).
 * ------------------------------------------------------------------------- */
typedef struct vb2_model {
    int32_t is_heter;          /* !--WithinAncestry            (main.cpp:287)      */
    int32_t is_pc_fixed;       /* --FixPC                      (main.cpp:291-308)  */
    int32_t is_alpha_fixed;    /* --FixAlpha                   (main.cpp:309-313)  */
    int32_t is_af_known;       /* --KnownAF                    (main.cpp:314-319)  */
    double fix_alpha;          /* --FixAlpha value                                 */
    const double *fix_pc;      /* --FixPC values (num_pc) or NULL                  */
    double epsilon;            /* --Epsilon, 1e-8 by default   (main.cpp:76)       */
    int32_t verbose;           /* --Verbose: per-evaluation notice (h:435-440)     */
    int32_t notices;           /* print the reference's stderr lines: PhaseTimer "Starting/Finished
                                * phase" (ContaminationEstimator.cpp:10-24, 93-155) and the
                                * non-convergence warning (MathGenMin.cpp:381); vb2_run sets it */
} vb2_model;

#define VB2_MAX_PC 64

typedef struct vb2_estimate {
    double alpha;              /* fn.globalAlpha                                   */
    double llk1;               /* fn.llk1 (= -LLK at the best evaluated point)     */
    double llk0;               /* fn.llk0                                          */
    double pc[VB2_MAX_PC];     /* fn.globalPC   (contaminating sample)             */
    double pc2[VB2_MAX_PC];    /* fn.globalPC2  (intended sample)                  */
    int64_t num_eval;          /* likelihood evaluations the reference would make  */
    int64_t num_launch_point;  /* points actually evaluated (speculation included) */
    int32_t converged;         /* 0 if a Minimize() ran out of cycles              */
    int32_t reserved;          /* vb2_ctx_optimize_llk_ex: index of the winning start, else 0 */
} vb2_estimate;

/* The objective seam, batched: the analogue of VectorFunc::Evaluate
 * (statgen/MathVector.h:281-308).  Must write +LLK for each point. */
typedef int (*vb2_eval_fn)(void *user, int32_t num_point, const double *pc1,
                           const double *pc2, const double *alpha, double *llk_out);

/* Optional per-evaluation trace: one record per evaluation the REFERENCE would
 * have made, in its order. */
typedef struct vb2_trace {
    int64_t capacity;
    int64_t count;
    double *alpha;             /* [capacity]            */
    double *pc1;               /* [capacity * num_pc]   */
    double *pc2;               /* [capacity * num_pc]   */
    double *llk;               /* [capacity]            */
} vb2_trace;

/* OptimizeLLK over an arbitrary evaluator (e.g. marker shards + all-reduce). */
int vb2_optimize_llk(vb2_eval_fn eval, void *user, int32_t num_pc, const vb2_model *model,
                     vb2_estimate *out, vb2_trace *trace);
/* OptimizeLLK on a context (evaluator = vb2_llk_eval_batch on ctx).  The search runs against ONE
 * kernel that stays resident on the device and receives each iteration's points through a mailbox
 * in pinned host memory; the calling thread spins until the search is over and
 * the context's stream is busy for that time.  Falls back to one launch per iteration on its own
 * when the mode is unavailable (VB2_RESIDENT=0 forces that).  Same results either way. */
int vb2_ctx_optimize_llk(vb2_ctx *ctx, const vb2_model *model, vb2_estimate *out,
                         vb2_trace *trace);

/* Optimiser variants beyond the reference's single Nelder-Mead run; with
 * num_start <= 1 and line_search == 0 this IS vb2_ctx_optimize_llk.
 *  - Multi-start: num_start searches from different starting points advance in lock-step on the
 *    context, the evaluations of one step leaving as ONE launch (north_star: "objective
 *    evaluations batch across restarts").  Start 0 is the reference's (h:321-331: PCs 0.01, alpha
 *    0.03); starts 1.. add seeded Gaussian noise to the free parameters -- what the reference's
 *    commented-out rand() starts (h:322, 326) and its otherwise unused --Seed were for.  *best is
 *    the run with the smallest llk1; all (optional) receives every run, all[0] = the reference's.
 *  - Line search: a model with ONE free parameter (--FixPC / --KnownAF: alpha alone) is minimised
 *    by golden-ratio bracketing + Brent's method (ScalarMinimizer, MathGold.cpp:27-195, which the
 *    reference links but never calls) instead of the two-vertex simplex. */
typedef struct vb2_search_opts {
    int32_t num_start;         /* >= 1 (0 counts as 1); at most 64                 */
    uint32_t seed;             /* --Seed                                           */
    double start_sd;           /* noise on the PC starts; logit(alpha) gets 50 x; 0 = 0.02 */
    int32_t line_search;       /* Brent for one-parameter models                   */
    int32_t reserved;
} vb2_search_opts;
int vb2_ctx_optimize_llk_ex(vb2_ctx *ctx, const vb2_model *model, const vb2_search_opts *opts,
                            vb2_estimate *best, vb2_estimate *all /* [num_start] or NULL */);

/* ------------------------------------------------------------------------- *
 * 2b. Cohorts: several samples (contexts on one device, same --NumPC) advancing in
 *     lock-step -- every Nelder-Mead step of every sample goes into ONE kernel launch
 *     (BASELINE.json configs[4]; the reference would run them as separate processes).
 * ------------------------------------------------------------------------- */
typedef struct vb2_batch vb2_batch;
#define VB2_BATCH_SLOTS 8      /* parameter points per sample and step */

int vb2_batch_create(vb2_ctx *const *ctxs, int32_t num_sample, vb2_batch **out);
void vb2_batch_destroy(vb2_batch *b);
/* One step: sample s evaluates num_point[s] (0..8) points.  pc1/pc2 are
 * [num_sample][8][num_pc], alpha and llk_out [num_sample][8]; unused slots are ignored. */
int vb2_batch_eval(vb2_batch *b, const int32_t *num_point, const double *pc1, const double *pc2,
                   const double *alpha, double *llk_out);
/* OptimizeLLK for every sample; models has 1 entry (shared) or num_sample entries. */
int vb2_batch_optimize_llk(vb2_batch *b, const vb2_model *models, int32_t num_model,
                           vb2_estimate *out);

/* ------------------------------------------------------------------------- *
 * 2c. Marker shards: ONE sample's markers spread over several GPUs (BASELINE.json configs[3]).
 *     LLK is a sum of independent per-marker terms -- the reference's OpenMP
 *     `reduction(+:sumLLK)` over markers (ContaminationEstimator.h:232-235) -- so each device
 *     evaluates a contiguous, read-balanced marker range and the B partial sums of a batch meet
 *     in ONE ncclAllReduce of B doubles over RCCL/xGMI.  librccl is bound at run time when a
 *     group spans more than one device.
 * ------------------------------------------------------------------------- */
typedef struct vb2_shard_group vb2_shard_group;

/* One process drives every device (VerifyBamID --Devices a,b,...).  `in` is the WHOLE sample;
 * shard d lives on devices[d].  Evaluations: one launch per device + a grouped ncclAllReduce;
 * vb2_shard_group_optimize_llk runs the search against one resident kernel per device and adds
 * the (<= 4) partial sums per device on the host.  A device listed twice gets two shards that
 * are added on the host (no RCCL): that is how a single-GPU machine exercises the code. */
int vb2_shard_group_create(const vb2_input *in, const int32_t *devices, int32_t num_device,
                           vb2_shard_group **out);
/* One process per GPU (python -m torch.distributed.run, mpirun): rank 0 calls vb2_rccl_unique_id,
 * the caller broadcasts the 128 bytes by its own means, then every rank builds the group with the
 * WHOLE sample, its device, its rank.  Every evaluation is launch + ncclAllReduce on the
 * context's stream; all ranks receive identical sums and take identical search decisions.
 * id128 == VB2_SHARD_PARTIAL_SUMS asks for a group WITHOUT a communicator: vb2_shard_group_eval then
 * returns this rank's PARTIAL sums and the caller reduces them over its own transport
 * (torch.distributed, MPI); vb2_shard_group_optimize_llk needs whole sums and refuses such a group.
 * id128 == NULL is accepted with nranks == 1 only (one rank needs no communicator); with more ranks it
 * is VB2_ERR_INVALID -- since ABI 5: a forgotten id used to yield plausible but partial LLKs silently.
 * vb2_shard_info.partial_sums says which kind a group is. */
#define VB2_SHARD_PARTIAL_SUMS ((const void *)(uintptr_t)1)
int vb2_rccl_unique_id(void *id128 /* 128 bytes out */);
int vb2_shard_group_create_rank(const vb2_input *in, int32_t device, int32_t rank, int32_t nranks,
                                const void *id128, vb2_shard_group **out);
/* ComputeMixLLKs over all shards for B points (host pointers, synchronous). */
int vb2_shard_group_eval(vb2_shard_group *g, int32_t num_point, const double *pc1, const double *pc2,
                         const double *alpha, double *llk_out);
int vb2_shard_group_optimize_llk(vb2_shard_group *g, const vb2_model *model, vb2_estimate *out,
                                 vb2_trace *trace);
typedef struct vb2_shard_info {
    int32_t num_shard;         /* shards owned by THIS process                     */
    int32_t nranks;            /* processes in the group (1 = single process)      */
    int32_t rank;
    int32_t uses_rccl;         /* partial sums meet in ncclAllReduce (else: host)  */
    int64_t num_allreduce;     /* collectives issued so far                        */
    int32_t marker_lo[64];     /* marker range of each owned shard                 */
    int32_t marker_hi[64];
    int64_t num_read[64];      /* reads of each owned shard                        */
    int32_t partial_sums;      /* 1: no communicator by request (VB2_SHARD_PARTIAL_SUMS): eval = this rank's part */
    int32_t rccl_stub;         /* 1: the collective library bound at run time is the test stand-in
                                * (VB2_RCCL_LIB=tests/stub_rccl/librccl_stub.so), not librccl                */
} vb2_shard_info;
int vb2_shard_group_info(const vb2_shard_group *g, vb2_shard_info *info);
/* The partition itself (no device needed): shard `rank` of `nranks` owns markers [*lo, *hi), cut so
 * that every shard holds about the same number of READS (balance on R, not M). */
int vb2_shard_range(const vb2_input *in, int32_t rank, int32_t nranks, int32_t *lo, int32_t *hi);
void vb2_shard_group_destroy(vb2_shard_group *g);

/* ------------------------------------------------------------------------- *
 * 3. File level: the --SVDPrefix/--PileupFile flow of execute()
 *    (main.cpp:283-411): panel + pileup readers, sanity check, OptimizeLLK,
 *    <out>.Ancestry and <out>.selfSM writers.
 * ------------------------------------------------------------------------- */
typedef struct vb2_mpileup_opts {
    int32_t given;             /* 0: every field below is ignored, the defaults of main.cpp:81-96 apply */
    int32_t min_bq;            /* --min-BQ      skip bases with baseQ/BAQ below it         (13)   */
    int32_t min_mq;            /* --min-MQ      skip alignments with mapQ below it         (2)    */
    int32_t adjust_mq;         /* --adjust-MQ   mapQ cap coefficient, 0 disables           (40)   */
    int32_t max_depth;         /* --max-depth   per-file depth cap                         (8000) */
    int32_t no_orphans;        /* --no-orphans  drop anomalous read pairs                  (0)    */
    int32_t incl_flags;        /* --incl-flags  the reference stores this in mplp.flag     (REALN | SMART_OVERLAPS) */
    int32_t excl_flags;        /* --excl-flags  skip reads with any of these bits set      (UNMAP|SECONDARY|QCFAIL|DUP) */
} vb2_mpileup_opts;

typedef struct vb2_run_args {
    const char *ud_path;       /* <SVDPrefix>.UD   (main.cpp:229)                  */
    const char *mean_path;     /* <SVDPrefix>.mu                                   */
    const char *bed_path;      /* <SVDPrefix>.bed                                  */
    const char *pileup_path;   /* --PileupFile                                     */
    const char *known_af_path; /* --KnownAF or NULL                                */
    const char *output_prefix; /* --Output (default "result"); NULL = write nothing*/
    int32_t num_pc;            /* --NumPC (default 2)                              */
    int32_t disable_sanity;    /* --DisableSanityCheck                             */
    int32_t output_pileup;     /* --OutputPileup                                   */
    int32_t device;            /* HIP device ordinal, -1 = current                 */
    vb2_model model;
    /* --Devices a,b,...: more than one entry spreads the work over those GPUs -- vb2_run shards
     * the sample's markers (2c), vb2_cohort_run deals whole groups of samples to the devices
     * (no collective).  NULL / 0 = `device` alone. */
    const int32_t *devices;
    int32_t num_device;
    int32_t reserved;
    /* --BamFile / --Reference: BAM or CRAM input through htslib (SimplePileupViewer.cpp:172-557);
     * used when pileup_path is NULL.  A library built without htslib fails with VB2_ERR_IO. */
    const char *bam_path;
    const char *reference_path;
    /* --NumStart / --Seed / --LineSearch: optimiser variants (vb2_search_opts); all zero = the
     * reference's single Nelder-Mead run.  One sample on one device only. */
    vb2_search_opts search;
    /* (ABI 6) The reference's "Pileup Options" (main.cpp:176-187; defaults main.cpp:81-96): they shape what
     * --BamFile input turns into pileup columns (SimplePileupViewer.cpp:172-237, 457-476) and have no effect on
     * --PileupFile input, exactly as in the reference.  given == 0: the defaults. */
    vb2_mpileup_opts mpileup;
} vb2_run_args;

typedef struct vb2_run_result {
    vb2_estimate est;
    int32_t num_marker;        /* #SNPS                                            */
    int32_t num_site;          /* sites shared with the pileup                     */
    int64_t num_bases;         /* viewer.numBases                                  */
    double avg_depth;          /* AVG_DP                                           */
    double sd_depth;
    double seconds_load;       /* wall-clock: readers                              */
    double seconds_optimize;   /* wall-clock: OptimizeLLK (the "converged alpha" time) */
} vb2_run_result;

int vb2_run(const vb2_run_args *args, vb2_run_result *out);

/* Cohort form of vb2_run (BASELINE configs[4]: many samples against one panel).  The reference
 * runs one process per sample; here the panel (.UD/.mu/.bed, optional AF file) is read once, the
 * pileups are read by host threads (and flattened on the device) while the device searches the previous group of
 * samples in lock-step (one kernel launch per search step for the whole group, vb2_batch_*), and
 * every sample gets the outputs vb2_run would write (<prefix>.Ancestry, <prefix>.selfSM, optional
 * <prefix>.Pileup).  args->pileup_path / output_prefix of `base` are ignored.
 * status[s] receives the sample's own VB2_* code (e.g. VB2_ERR_SANITY); the call itself fails only
 * for errors that concern all samples (panel, device). */
typedef struct vb2_cohort_args {
    vb2_run_args base;                   /* panel paths, num_pc, flags, model, device          */
    int32_t num_sample;
    const char *const *pileup_paths;     /* [num_sample]                                      */
    const char *const *output_prefixes;  /* [num_sample], or NULL = write nothing             */
    int32_t group_size;                  /* samples on a device at a time (<= 64); 0 = 32: slots
                                          * that a converged sample hands to the next ready one;
                                          * < 0: |group_size|.  (VB2_COHORT_STREAM=0: groups of that
                                          * size searched one after the other, the first of 16)   */
    int32_t num_host_thread;             /* pileup readers/flatteners; 0 = the CPUs the process may
                                          * use (cgroup quota / affinity) less one, at most 64 per
                                          * device                                                */
} vb2_cohort_args;
int vb2_cohort_run(const vb2_cohort_args *args, vb2_run_result *out /* [num_sample] */,
                   int32_t *status /* [num_sample] */);

/* Host-side flattening only (no device): reads panel + pileup, resolves markers
 * and returns the arrays of vb2_input in library-owned memory; free with
 * vb2_flat_free.  Lets callers (tests, shard planners) inspect or slice them. */
typedef struct vb2_flat vb2_flat;
int vb2_flat_load(const vb2_run_args *args, vb2_flat **out);
const vb2_input *vb2_flat_input(const vb2_flat *f);
int vb2_flat_stats(const vb2_flat *f, vb2_run_result *out);
void vb2_flat_free(vb2_flat *f);

const char *vb2_last_error(void);
int vb2_abi_version(void);
/* Number of usable gfx950 devices (0 = none; compute calls will fail loudly). */
int vb2_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* VB2_ABI_H_ */
