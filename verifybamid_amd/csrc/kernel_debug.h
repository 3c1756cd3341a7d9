// kernel_debug.h -- every compile-time switch of the kernels (llk_kernels.hip, resident_kernel.inc).  A shipping build
// defines none of them; the profiling builds are made by csrc/Makefile (libvb2_stamps.so, `make stamps_round`) and
// tools/build_variant.sh.
//
//   VB2_WITH_STAMPS     in-kernel wall-clock stamps per workgroup (tools/stamps.py, tools/stamps_resident.py): the tests they
//                       leave behind in every kernel cost 0.6 % of a 48-point launch and 4 % of a small sample's search
//   VB2_STAMP_ROUND=n   the per-workgroup stamps are those of round n of a resident search (a round that took the short way)
//                       instead of the last evaluation's
//   VB2_STAMP_CTRL      stamp slot 3 of workgroup 0 = the control wave is back from its tile-phase work
//   VB2_ITEM_PROF       with VB2_WITH_STAMPS: where a wave's time per work item goes (tools/item_prof.py)
//   VB2_FLAG_RELEASE=0  the host hand-off's flag as a RELAXED system-scope store behind acknowledged write-through result stores (rounds
//                       4-5; the shipping build stores it with RELEASE semantics, and the host still sets the result words to NaN
//                       before every step and re-reads a NaN)
//   VB2_ABLATE=mask     ablation builds -- parts of a launch compiled out, to price what is left:
//                         1 kAblNoMap     nothing is read from mapped host memory (rows are made up, counts assumed full)
//                         2 kAblNoTable   the per-alpha table is not built
//                         4 kAblNoItems   no work items at all: a launch's fixed cost
//                         8 kAblNoReads   the run words are consumed, the table is not read
//                        16 kAblNoEpi     the sums and constants are consumed, nothing is computed from them
//                        32 kAblNoSignal  no hand-off to the host
//                        64 kAblNoMul     probability domain: the table rows are read, the products not multiplied
//                       128 kAblNoRowLoads  probability domain: the steps are made up in registers, no loads of the lists
#ifndef VB2_KERNEL_DEBUG_H_
#define VB2_KERNEL_DEBUG_H_

#ifndef VB2_ABLATE
#define VB2_ABLATE 0
#endif
#ifndef VB2_FLAG_RELEASE
#define VB2_FLAG_RELEASE 1      // the flag the host spins on is stored with system-scope RELEASE semantics (0: relaxed; A/B: profiles/r06/ab_flag_release.txt -- no measurable cost)
#endif
#ifndef VB2_SOFT_PROLOGUE
#define VB2_SOFT_PROLOGUE 1     // the resident kernel's control wave starts its round's work at once instead of behind the table build (eval_body; 0: A/B)
#endif
#ifndef VB2_CTRL_PRIO
#define VB2_CTRL_PRIO 3         // the control wave's priority during its tile-phase work
#endif
#ifndef VB2_STAMP_ROUND
#define VB2_STAMP_ROUND 0
#endif

namespace vb2 {
constexpr int kAblNoMap = 1, kAblNoTable = 2, kAblNoItems = 4, kAblNoReads = 8, kAblNoEpi = 16, kAblNoSignal = 32, kAblNoMul = 64, kAblNoRowLoads = 128;
constexpr int kAblate = VB2_ABLATE;
}  // namespace vb2

#ifdef VB2_WITH_STAMPS
#define VB2_STAMPS_OF(L) ((L).stamps)
#else
#define VB2_STAMPS_OF(L) (static_cast<unsigned long long*>(nullptr))
#endif

#ifdef VB2_ITEM_PROF
#define VB2_IP_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define VB2_IP_USE(x) asm volatile("" ::"v"(x))
#else
#define VB2_IP_T(var)
#define VB2_IP_USE(x)
#endif

#endif
