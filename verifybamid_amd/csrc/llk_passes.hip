// llk_passes.hip -- the second translation unit of llk_kernels.hip: llk_eval_passes_kernel (wide quality alphabets: the passes
// of one launch) and the cohort kernels llk_eval_multi_kernel with their launcher.  Everything is in llk_kernels.hip; this
// unit exists so that these kernels are compiled under LLVM's default instruction scheduler while the single-sample kernels
// and the resident search kernel take the iterative-ILP one (see there: VB2_TU_PASSES).
#define VB2_TU_PASSES
#include "llk_kernels.hip"
