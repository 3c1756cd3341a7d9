// llk_kernels.h -- device data layout + kernel launchers (internal).
#ifndef VB2_LLK_KERNELS_H_
#define VB2_LLK_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

namespace vb2 {

constexpr int kNumQual = 94;             // Phred 0..93 (ContaminationEstimator.h:65-74)
constexpr int kMaxCode = 2 * kNumQual;   // classes ref/alt x qualities
constexpr int kPadCode = 255;            // never a real dictionary index
constexpr int kMtMarkers = 16;           // markers per micro-tile (one 16-lane group)
constexpr int kMaxBlockWaves = 16;       // 1024-thread blocks at most
constexpr int kMaxGroups = 6;             // groups of 8 points per launch
constexpr int kMaxPointsPerLaunch = 8 * kMaxGroups;
constexpr int kLdsLimitBytes = 160 * 1024;
constexpr int kInlinePointDoubles = 96;    // parameter rows that travel as kernel arguments (768 B)

// Everything the kernels read, in HBM.  "Sorted order" = active markers sorted by
// (non-"other") depth, descending; position = micro_tile*16 + m.
struct DeviceLayout {
    const uint32_t* codes;        // [sum_t mt_rows[t]][16] dwords = two (code, count) runs each
    const uint2* mt_rec;          // [num_mt] {first row, rows = ceil(most runs in the tile / 2)}
    const double* ud;             // [num_pc][m_pad]  (SoA)
    const double* mu;             // [m_pad]
    const double* ediag;          // [4][m_pad]: c_other, exp(c_other+D[g]) for g = 0,1,2
    const double* known_af;       // [m_pad] or nullptr
    const double* dict_perr;      // [num_code] +10^(-q/10) class ref, -10^(-q/10) class alt
    const double2* prim;          // [num_prim] codes whose table rows are computed: {signed pErr, bits: code |
                                  // twin << 16}, twin = the alt code of the same quality (its row is the mirror
                                  // image of this one) or 0xffff
    int32_t num_code;
    int32_t num_prim;
    int32_t num_mt;
    int32_t num_pc;
    int32_t num_cu;               // compute units of the device
    int32_t ablate;               // profiling aid (VB2_ABLATE): 1 no table math, 2 no read loop, 4 no epilogue math
    unsigned long long* stamps;   // profiling aid: [grid][8] wall-clock stamps, or nullptr
    int64_t num_active;
    int64_t m_pad;                // num_mt * 16
};

struct LaunchGeom { int grid, block_waves; };
// Workgroups / waves per workgroup used for a launch of 4*btl points.
LaunchGeom launch_geom(const DeviceLayout& L, int btl);
constexpr int kMaxGridPerCU = 2;
constexpr int kDynTilesPerWave = 10;      // up to this many tiles per wave: dynamic queue + per-tile slots
inline int max_grid(const DeviceLayout& L) { return kMaxGridPerCU * L.num_cu; }

// Enqueue evaluation of num_point candidate rows (pc1 | pc2 | alpha) on stream.
// d_partials: >= (kMaxPointsPerLaunch + 1) * kMaxGridPerCU * L.num_cu doubles of scratch.
// tag_counter: incremented per kernel launch; tags must never repeat on one partials buffer.
// d_ticket: one zero-initialised unsigned int (arrival counter of the single-launch mode).
// done_flag: optional word in mapped host memory that receives done_seq after the results of
// the LAST launch are written (lets the host wait without hipStreamSynchronize).
// h_points: the same rows readable by the host (or nullptr): small batches then travel as
// kernel arguments instead of being read from (possibly mapped host) memory.
hipError_t launch_llk_eval(const DeviceLayout& L, int num_point, const double* d_points,
                           const double* h_points, double* d_partials, double* d_out,
                           unsigned int* d_ticket,
                           unsigned long long* done_flag, unsigned long long done_seq,
                           unsigned long long* tag_counter, hipStream_t stream,
                           int reduce_override = 0);     // 1 ticket / 2 tagged for this call only
void set_single_launch(bool on);
void set_reduce_mode(int mode);     // 0 auto, 1 arrival ticket, 2 tagged sets (VB2_REDUCE)

// One launch over several samples (contexts on the same device): see llk_eval_multi_kernel.
struct MultiLaunch {
    const DeviceLayout* d_layouts;   // [num_sample] in HBM
    const double* d_points;          // [num_sample][4*btl][2k+1]
    const int* d_num_valid;          // [num_sample] 0 = sample sits this step out
    double* d_partials;              // [num_sample][4*btl + 1][bps]; done_seq doubles as the launch tag
    double* d_out;                   // [num_sample][4*btl]
    unsigned int* d_tickets;         // [num_sample], zero-initialised
    unsigned int* d_batch_done;      // one zero-initialised counter
    unsigned long long* done_flag;   // mapped host word or nullptr
    unsigned long long done_seq;
    unsigned int batch_active;       // samples with num_valid > 0
    int num_sample, bps, block_waves, btl;
    bool force_ticket;               // arrival-ticket hand-off instead of tagged sets (the retry after a NaN)
    size_t shmem;
};
size_t eval_shmem_bytes(const DeviceLayout& L, int btl, int nblk, int block_waves, int ngrp = 1);   // groups of 4*btl points
size_t eval_shmem_np(const DeviceLayout& L, int np, int nblk, int block_waves, int ngrp);
int max_groups(const DeviceLayout& L, int btl, int nblk, int block_waves);
hipError_t launch_llk_eval_multi(const MultiLaunch& ml, hipStream_t stream);   // VB2_SINGLE_LAUNCH=0 -> eval + finalize kernels
hipError_t launch_fill_zero(double* d_out, int n, hipStream_t stream);

// Resident search kernel (llk_resident_kernel): launched once per search, fed through a mailbox.
// Word layout of h_cmd / relay: [0] seq, [1] rows valid (0 = exit), [2..2+4*(2k+1)) rows
// (pc1 | pc2 | alpha), [2+4*(2k+1)] check word: XOR of word_hash(word w, w) over [1, last) ^ resident_mix(seq).
struct ResidentArgs {
    const unsigned long long* h_cmd;     // mailbox in mapped host memory (device view)
    unsigned long long* relay;           // same layout in device memory, zero-initialised
    double* h_out;                       // [4] results, mapped host memory (device view)
    unsigned long long* h_done;          // completion sequence number, mapped host memory
    unsigned int* h_state;               // 1 running, 2 exited on command, 3 gave up (idle timeout)
    unsigned long long first_seq;        // sequence number of the first command
    unsigned long long timeout_ticks;    // idle limit in 100 MHz wall-clock ticks
};
__host__ __device__ inline unsigned long long resident_mix(unsigned long long seq)
{
    return seq * 0x9E3779B97F4A7C15ull;      // spreads consecutive sequence numbers over 64 bits
}
// Position-dependent 64-bit hash of one word (xor-shift, multiply, xor-shift).  The check words of the
// mailbox and of the partial-sum hand-off are the XOR of these over the payload, ^ mix(tag):
// a plain XOR of the payload would be blind to equal words (pc1 == pc2 rows, the four equal
// sums of a one-point launch), i.e. to exactly the stale/new mixtures it has to catch.
__host__ __device__ inline unsigned long long word_hash(unsigned long long v, unsigned int pos)
{
    unsigned long long z = v + 0x9E3779B97F4A7C15ull * (unsigned long long)(pos + 1);
    z = (z ^ (z >> 32)) * 0xBF58476D1CE4E5B9ull;
    return z ^ (z >> 29);
}
inline int resident_words(int num_pc) { return 2 + 4 * (2 * num_pc + 1) + 1; }
void set_paired_mode(bool on);     // VB2_PAIRED=0: 4-point launches use MODE 1 instead of MODE 3
void set_coop_launch(bool on);     // VB2_COOP=1: cooperative launch of the resident kernel
hipError_t launch_llk_resident(const DeviceLayout& L, const ResidentArgs& ra, double* d_partials,
                               unsigned int* d_ticket, hipStream_t stream);
// A/B switch: lanes of one ds_read_b128 service group share a candidate slot (default) or plain
void set_lane_mapping(bool hardware_groups);
void set_geom_override(int btl, int max_waves, int blocks_per_cu);

}  // namespace vb2
#endif
