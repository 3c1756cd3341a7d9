// llk_passes.hip -- the translation unit of llk_eval_passes_kernel (wide quality alphabets: the passes of one launch).
// The kernel and everything it is made of are in llk_kernels.hip; this unit exists so that it is compiled under LLVM's
// default instruction scheduler while the other kernels take the iterative-ILP one (see there: VB2_TU_PASSES).
#define VB2_TU_PASSES
#include "llk_kernels.hip"
