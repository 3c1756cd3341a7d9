"""verifybamid_amd -- MI355X-native contamination-likelihood core for VerifyBamID2.

The product is libvb2.so (HIP kernels + C++ host, C-ABI in include/vb2_abi.h) and the
`bin/VerifyBamID` command line; this package is the ctypes marshalling used by the
tests, the bench and multi-GPU (torch.distributed) drivers.
"""
from .api import (CohortBatch, LikelihoodContext, PileupData, ShardGroup, optimize_with_evaluator,  # noqa: F401
                  run_cohort_files, run_files)
from . import synth  # noqa: F401

__all__ = ["CohortBatch", "LikelihoodContext", "PileupData", "ShardGroup", "optimize_with_evaluator", "run_cohort_files",
           "run_files", "synth"]
