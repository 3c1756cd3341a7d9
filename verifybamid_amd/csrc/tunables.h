// tunables.h -- every run-time switch of the library, in one table.
//
// Production code never needs any of them: the defaults are the measured best, and what a caller may legitimately
// choose is in the C-ABI (vb2_options.flags, vb2_search_opts, vb2_cohort_args).  They exist for the tests (which compare
// two implementations of the same thing: host pack vs device pack, streamed vs grouped cohorts, ...), for the
// measurement scripts under tools/, and as escape hatches on a misbehaving box.
//
// Two ways in, both ending in the same table:
//   * vb2_debug_set_tunable("name", value) / vb2_debug_get_tunable (abi.cpp) -- what the in-process tests use;
//   * the environment, ONCE, at the first use of the table: VB2_<NAME> in upper case (VB2_COHORT_STREAM=0, ...),
//     for tools and tests that start the library in a child process.  This is the library's only reader of
//     behaviour-changing environment variables (shard.cpp reads VB2_RCCL_LIB, a path, on its own).
#ifndef VB2_TUNABLES_H_
#define VB2_TUNABLES_H_

namespace vb2 {

// X(name, default)
#define VB2_TUNABLE_LIST(X)                                                                                              \
    /* ---- launches and hand-off (llk_kernels.hip) ---- */                                                             \
    X(single_launch, 1)    /* 0: evaluation + llk_finalize_kernel instead of the in-launch reduction */                 \
    X(reduce, 0)           /* cross-workgroup hand-off: 0 by size, 1 arrival ticket, 2 tagged sets */                   \
    X(tagged_max, 16)      /* launches of up to this many points hand off through tagged sets */                        \
    X(passes, 1)           /* 0: a launch per table-load of points instead of llk_eval_passes_kernel */                 \
    X(split, 1)            /* probability-domain launches of many points: 0 in passes, 1 split between 2 or 3 workgroups, 2 pairs only */ \
    X(coop, 1)             /* the resident search kernel goes up with hipLaunchCooperativeKernel (0: plain launch;      \
                              the library also launches plainly by itself when a profiler's tool library is loaded) */  \
    X(dyn_tiles, 10)       /* work items per wave up to which a workgroup's waves pull them through the LDS queue */    \
    X(stagger, 0)          /* second half of a workgroup's waves starts its tiles n x 64 cycles late */                 \
    X(sched, 1)            /* 0: in-kernel snake deal instead of the host-built static schedules */                     \
    X(stamps, 0)           /* 1: room for the in-kernel stamps of the profiling builds (libvb2_stamps.so) */            \
    /* ---- context creation / flatten (context.cpp) ---- */                                                            \
    X(flatten_threads, 0)  /* host threads of the flatten, 0 = by input size */                                          \
    X(host_flatten, 0)     /* 1: classify / run-length code / sort on the host (the checker of the device flatten) */   \
    X(host_pack, 0)        /* 1: (with host_flatten) also the kernel-order arrays on the host */                         \
    X(run_sched, -1)       /* order of a tile's runs: -1 scheduled above kSchedMinCodes codes, 0 plain, 1 scheduled */   \
    X(force_narrow, 0)     /* 1: narrow table rows although the dictionary would fit wide ones */                        \
    X(pd, 1)               /* 0: never the probability-domain layout (llk_kernels.h: kMaxPow) */                         \
    X(pd_rows, 0)          /* its table rows: 0 = what the LDS holds for a 48-point launch */                            \
    X(pd_pairs, 1)         /* 0: no window rows in it (llk_kernels.h: PdDict) */                                         \
    X(digest_multiset, 0)  /* 1: vb2_debug_flatten_digest takes a tile's run words as a multiset per lane */             \
    X(slab_cache, 1)       /* 0: freed device / pinned slabs go back to the driver */                                    \
    X(cpus, 0)             /* CPUs the process may use, 0 = cgroup quota / affinity */                                   \
    /* ---- one sample's search (context.cpp, estimator.cpp) ---- */                                                    \
    X(resident, 1)         /* 0: a launch per search step instead of the resident kernel */                              \
    X(ctl_block, 1)        /* 0: the control wave lives in workgroup 0 even where the grid leaves a CU free for a workgroup of its own */ \
    X(device_simplex, 1)   /* 0: the resident kernel evaluates, the host optimiser decides */                            \
    X(spin_wait, 1)        /* 0: hipStreamSynchronize instead of spinning on the mapped flag */                          \
    X(lds_cache, 1)        /* 0: the resident kernel reads its run lists from L2 every round */                          \
    X(speculate, 0)        /* points per Nelder-Mead iteration of the host optimiser: 0 = the caller's, else 1, 2, 4 */  \
    /* ---- cohorts (batch.cpp, cohort.cpp) ---- */                                                                     \
    X(cohort_w16, 1)       /* 0: cohort steps stream the 32-bit run lists */                                             \
    X(cohort_bw, 8)        /* waves per workgroup of a cohort step: 4, 8, 16 */                                          \
    X(cohort_speculate, 0) /* 0 = by cohort size, else 1, 2, 4 */                                                        \
    X(cohort_split, -1)    /* -1 = from 16 samples on, 0 / 1 */                                                          \
    X(cohort_lanes, 2)     /* lanes taking turns */                                                                      \
    X(cohort_regroup, 1)   /* 0: the fixed batch to the end */                                                           \
    X(cohort_stream, 1)    /* 0: groups searched one after the other instead of slots that change hands */              \
    X(cohort_own_queues, 1) /* 0: the streamed search's lanes and the readers' contexts take whatever streams the cache has */ \
    X(cohort_stream_points, 2) /* points per request of a streamed sample (a simplex's first vertices leave in pieces) */   \
    X(cohort_dup_devices, 0) /* 1: vb2_cohort_run accepts a device listed twice (one-GPU test of several pipelines) */   \
    X(cohort_fail_sample, -1) /* test hook: the streamed search refuses this sample at its slot (error path) */         \
    /* ---- marker shards (shard.cpp) ---- */                                                                           \
    X(shard_reduce_host, 0) /* 1: partial sums met on the host although a collective is possible */                     \
    /* ---- readers (hostio.cpp) ---- */                                                                                \
    X(slow_parse, 0)       /* 1: every line through the stringstream statements */                                       \
    X(scalar_parse, 0)     /* 1: no AVX2 scanner */                                                                      \
    /* ---- diagnostics ---- */                                                                                         \
    X(debug_timing, 0)                                                                                                   \
    X(debug_lockstep, 0)

struct Tunables {
#define VB2_X(name, dflt) int name = dflt;
    VB2_TUNABLE_LIST(VB2_X)
#undef VB2_X
};

// The table (environment overrides applied at the first call).  Plain ints: set them only between operations.
Tunables& tunables();
bool set_tunable(const char* name, int value);
bool get_tunable(const char* name, int* value);

}  // namespace vb2
#endif
