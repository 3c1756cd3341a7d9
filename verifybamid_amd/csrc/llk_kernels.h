// llk_kernels.h -- device data layout + kernel launchers (internal).
#ifndef VB2_LLK_KERNELS_H_
#define VB2_LLK_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace vb2 {

constexpr int kNumQual = 94;         // Phred 0..93 (ContaminationEstimator.h:65-74)
constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kMaxCode = 2 * kNumQual;   // classes ref/alt x qualities
constexpr int kPadCode = 255;            // never a real code; maps to a zero table row

// Everything the kernel reads, in HBM.  "Sorted order" = active markers sorted by
// (non-"other") depth, descending; position m = tile*64 + lane.
struct DeviceLayout {
    const uint32_t* codes;         // [sum_t tile_rows[t]][64] dwords; byte j of row s = step 4s+j
    const uint32_t* tile_row_off;  // [num_tile] first row of the tile
    const uint32_t* tile_rows;     // [num_tile] rows (= ceil(max depth in tile / 4))
    const double* ud;              // [num_pc][m_pad]  (SoA)
    const double* mu;              // [m_pad]
    const double* ediag;           // [4][m_pad]: c_other, exp(c_other+D[g]) for g = 0,1,2
    const double* known_af;        // [m_pad] or nullptr
    const double* dict_perr;       // [num_code] +10^(-q/10) for class ref, -10^(-q/10) for class alt
    int32_t num_code;
    int32_t num_tile;
    int32_t num_pc;
    int64_t num_active;
    int64_t m_pad;                 // num_tile * 64
};

int max_points_per_launch();
// Enqueue evaluation of num_point candidate rows (pc1 | pc2 | alpha) on stream.
// d_partials: >= max_points_per_launch() * num_blocks doubles of scratch.
hipError_t launch_llk_eval(const DeviceLayout& L, int num_point, const double* d_points,
                           double* d_partials, double* d_out, hipStream_t stream);
hipError_t launch_fill_zero(double* d_out, int n, hipStream_t stream);

inline int num_blocks_for(const DeviceLayout& L)
{
    return (L.num_tile + kWavesPerBlock - 1) / kWavesPerBlock;
}

}  // namespace vb2
#endif
