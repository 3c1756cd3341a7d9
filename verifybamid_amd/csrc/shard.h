// shard.h -- one sample's markers sharded over several GPUs (BASELINE.json
// configs[3]).  LLK = sum over markers of independent terms (ContaminationEstimator.h:236-312;
// the reference's OpenMP `reduction(+:sumLLK)`, h:232-235), so every device holds a contiguous,
// read-balanced marker range and evaluates a partial sum; the partial sums of the B points of a
// batch meet in ONE all-reduce of B doubles over RCCL/xGMI.
//
// Two deployments share the class:
//   * one process, several devices (VerifyBamID --Devices a,b,..): contexts on every device,
//     ncclCommInitAll, evaluations = one launch per device + a grouped ncclAllReduce; the
//     Nelder-Mead search runs against one RESIDENT kernel per device, all fed the same command
//     through their mailboxes, and the host adds the partial sums in shard order (4 doubles per
//     device; a persistent kernel cannot take part in a stream-ordered collective);
//   * one process per GPU (torch.distributed.run / mpirun): every rank builds the same group with
//     its own rank's shard and a ncclUniqueId the caller distributed; each evaluation is
//     launch + ncclAllReduce on the context's stream, and every rank's optimiser takes the same
//     decisions on the same all-reduced values.
// librccl is bound at run time (dlopen) the first time a group with more than one distinct
// device is built: the single-GPU product has no RCCL dependency.
#ifndef VB2_SHARD_H_
#define VB2_SHARD_H_

#include <vector>

#include "context.h"

namespace vb2 {

// ncclGetUniqueId through the run-time binding (128 bytes out).
int rccl_unique_id(void* id128);
// The run-time binding found the test stand-in (VB2_RCCL_LIB=tests/stub_rccl/...), not librccl.
bool rccl_is_stub();
// Contiguous marker range [lo, hi) of shard r of n, balanced on READS (not markers).
void shard_range(const vb2_input* in, int r, int n, int* lo, int* hi);
// The sub-view of `in` for markers [lo, hi): pointers into the caller's arrays, nothing copied.
vb2_input shard_view(const vb2_input* in, int lo, int hi);

class ShardGroup {
public:
    ~ShardGroup();
    // single process, `num_device` devices (duplicates allowed: "virtual" shards on one device,
    // reduced on the host -- used by the 1-GPU tests)
    static int create(const vb2_input* in, const int32_t* devices, int num_device, ShardGroup** out);
    // one process per GPU: this process is shard `rank` of `nranks`; id128 = ncclUniqueId bytes
    static int create_rank(const vb2_input* in, int device, int rank, int nranks, const void* id128,
                           ShardGroup** out);
    int eval(int num_point, const double* pc1, const double* pc2, const double* alpha, double* llk_out);
    int optimize(const vb2_model* model, vb2_estimate* out, vb2_trace* trace);

    int num_pc = 0, num_marker = 0;
    int rank = 0, nranks = 1;              // process-per-GPU mode; (0, 1) in single-process mode
    bool use_rccl = false;                 // partial sums meet in ncclAllReduce (else: host sum)
    bool partial_sums = false;             // nranks > 1 without a communicator (by request): eval() yields this rank's part
    std::vector<Context*> ctx;             // the shards this process owns
    std::vector<int> lo, hi;               // their marker ranges
    int64_t num_allreduce = 0;

private:
    std::vector<void*> comm_;              // ncclComm_t per owned shard
    std::vector<double*> d_part_;          // [kStagePoints] partial / reduced LLKs per owned shard
    bool resident_ = false;
    int begin_resident();
    void end_resident();
    int eval_resident(int n, const double* pc1, const double* pc2, const double* alpha, double* out);
    int eval_launch(int n, const double* pc1, const double* pc2, const double* alpha, double* out);
};

}  // namespace vb2

struct vb2_shard_group {
    vb2::ShardGroup* impl;
};

#endif
