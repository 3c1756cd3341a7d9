// llk_kernels.hip -- gfx950 kernels for the genotype-mixture log-likelihood
// (reference: FullLLKFunc::ComputeMixLLKs, ContaminationEstimator.h:194-314).
//
// Work decomposition:
//   * one LANE per marker, 64 markers per wave tile; markers are sorted by depth
//     at context creation so a tile's lanes run the same number of steps;
//   * reads are dictionary codes (class x quality), 4 per dword, stored
//     [tile][step/4][lane] so every wave load is one contiguous 256-byte row;
//   * the per-alpha log-likelihood table (h:213-229) is rebuilt per launch in LDS,
//     restricted to the codes that occur in the data and to the six OFF-diagonal
//     genotype pairs: the diagonal (g1==g2) and the "other base" class do not
//     depend on alpha or the PCs and were summed once at context creation;
//   * each lane keeps 6 FP64 accumulators per candidate point in registers and
//     gathers its table row from LDS with ds_read_b128 (row = 6 values per point);
//   * epilogue per lane: UD*PC projection (h:251-267), HWE priors (h:186-192),
//     9-term exp-sum and log with the reference's `> 0` test (h:307-311);
//   * deterministic reduction: wave butterfly -> block -> per-block partial; a
//     one-block finalize kernel sums the partials in a fixed order.
#include "llk_kernels.h"

#include <hip/hip_runtime.h>

namespace vb2 {

// off-diagonal genotype pairs, in the reference's (g1 outer, g2 inner) order
__device__ __forceinline__ void pair_of(int p, int& g1, int& g2)
{
    // p: 0:(0,1) 1:(0,2) 2:(1,0) 3:(1,2) 4:(2,0) 5:(2,1)
    g1 = p >> 1;
    g2 = (p == 0) ? 1 : (p == 1) ? 2 : (p == 2) ? 0 : (p == 3) ? 2 : (p == 4) ? 0 : 1;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One table entry, with the reference's expression order (h:223-225).
// perr_signed = +pErr(q) for class ref, -pErr(q) for class alt (one load per code;
// the alt class is the ref class with genotypes mirrored, g -> 2-g: h:164-177).
__device__ __forceinline__ double table_entry(double alpha, double perr_signed, int g1, int g2)
{
    const bool alt = perr_signed < 0.0;
    const double p_err = fabs(perr_signed);
    const double p_ok = 1.0 - p_err;
    if (alt) { g1 = 2 - g1; g2 = 2 - g2; }
    // class ref: P(ref | g, error) = {0, 1/6, 1/3}[g], P(ref | g, no error) = {1, .5, 0}[g]
    const double e1 = g1 == 0 ? 0.0 : (g1 == 1 ? 1.0 / 6.0 : 1.0 / 3.0);
    const double e2 = g2 == 0 ? 0.0 : (g2 == 1 ? 1.0 / 6.0 : 1.0 / 3.0);
    const double n1 = g1 == 0 ? 1.0 : (g1 == 1 ? 0.5 : 0.0);
    const double n2 = g2 == 0 ? 1.0 : (g2 == 1 ? 0.5 : 0.0);
    const double one_minus_alpha = 1.0 - alpha;
    const double val = (alpha * e1 + one_minus_alpha * e2) * p_err +
                       (alpha * n1 + one_minus_alpha * n2) * p_ok;
    return log(val);
}

__device__ __forceinline__ void initial_gf(double af, double* gf)   // h:186-192
{
    if (af < 0.00005) af = 0.00005;
    if (af > 0.99995) af = 0.99995;
    gf[0] = (1 - af) * (1 - af);
    gf[1] = 2 * (af) * (1 - af);
    gf[2] = af * af;
}

// Row stride (in doubles) of the LDS table: 6 values per candidate point, padded so
// that stride/2 (in 16-byte slots) is odd -> 16 consecutive codes land in 16
// distinct 4-dword bank slots for ds_read_b128 (bank = (addr/4) % 64).
__host__ __device__ constexpr int row_stride(int bt) { return 6 * bt + ((bt % 2 == 0) ? 2 : 0); }

// BT = candidate points evaluated per launch (register-tiled).
// LDS: table[(num_code+1)][row_stride(BT)] doubles (row num_code = zeros = padding code),
//      then BT*kWavesPerBlock doubles of reduction scratch.
template <int BT>
__global__ void __launch_bounds__(kBlockThreads)
llk_eval_kernel(const DeviceLayout L, const double* __restrict__ points,
                double* __restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RS = row_stride(BT);
    const int nrow = L.num_code + 1;
    double* tab = lds;                       // [nrow][RS]
    double* red = lds + nrow * RS;           // [BT][kWavesPerBlock]
    double* pts = red + BT * kWavesPerBlock; // [BT][2k+1] this launch's parameter rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int k = L.num_pc;
    const int stride = 2 * k + 1;

    // parameter rows -> LDS with one coalesced load (they may live in mapped host memory)
    for (int e = tid; e < BT * stride; e += kBlockThreads) pts[e] = points[e];
    __syncthreads();

    // ---- per-alpha table, off-diagonal pairs only (h:213-229) ----
    for (int e = tid; e < nrow * 6 * BT; e += kBlockThreads) {
        const int d = e / (6 * BT);
        const int bp = e - d * (6 * BT);
        const int b = bp / 6, p = bp - b * 6;
        double v = 0.0;
        if (d < L.num_code) {
            int g1, g2;
            pair_of(p, g1, g2);
            v = table_entry(pts[b * stride + 2 * k], L.dict_perr[d], g1, g2);
        }
        tab[d * RS + bp] = v;
    }
    __syncthreads();

    double llk_lane[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) llk_lane[b] = 0.0;

    const int tile = blockIdx.x * kWavesPerBlock + wave;
    if (tile < L.num_tile) {
        double acc[BT * 6];
#pragma unroll
        for (int i = 0; i < BT * 6; ++i) acc[i] = 0.0;

        // ---- per-read accumulate (h:288-303); code rows prefetched two deep ----
        const uint32_t* cp = L.codes + (size_t)L.tile_row_off[tile] * 64 + lane;
        const int rows = L.tile_rows[tile];
        const uint32_t padw = 0x01010101u * (uint32_t)L.num_code;
        uint32_t w_cur = rows > 0 ? cp[0] : padw;
        uint32_t w_nxt = rows > 1 ? cp[64] : padw;
        for (int s = 0; s < rows; ++s) {
            const uint32_t w_n2 = (s + 2 < rows) ? cp[(size_t)(s + 2) * 64] : padw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (w_cur >> (8 * j)) & 0xffu;
                const double2* row = reinterpret_cast<const double2*>(tab + c * RS);
#pragma unroll
                for (int i = 0; i < 3 * BT; ++i) {
                    const double2 t = row[i];
                    acc[2 * i] += t.x;
                    acc[2 * i + 1] += t.y;
                }
            }
            w_cur = w_nxt;
            w_nxt = w_n2;
        }

        // ---- per-marker epilogue ----
        const size_t m = (size_t)tile * 64 + lane;       // position in sorted order
        if (m < (size_t)L.num_active) {
            const size_t mp = L.m_pad;
            const double cst = L.ediag[m];
            const double e0 = L.ediag[mp + m], e1 = L.ediag[2 * mp + m], e2 = L.ediag[3 * mp + m];
            double af1[BT], af2[BT];
            if (L.known_af) {
                const double a = L.known_af[m];
#pragma unroll
                for (int b = 0; b < BT; ++b) af1[b] = af2[b] = a;
            } else {
                // h:251-267: AF = (sum_k UD[i][k]*pc[k] + mean) / 2, same op order
#pragma unroll
                for (int b = 0; b < BT; ++b) af1[b] = af2[b] = 0.;
                for (int kk = 0; kk < k; ++kk) {
                    const double u = L.ud[(size_t)kk * mp + m];
#pragma unroll
                    for (int b = 0; b < BT; ++b) {
                        af1[b] += u * pts[b * stride + kk];
                        af2[b] += u * pts[b * stride + k + kk];
                    }
                }
                const double mu = L.mu[m];
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    af1[b] += mu; af1[b] /= 2.0;
                    af2[b] += mu; af2[b] /= 2.0;
                }
            }
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                double gf[3], gf2[3];
                initial_gf(af1[b], gf);
                initial_gf(af2[b], gf2);
                const double* a = acc + b * 6;
                // h:307-311, (g1 outer, g2 inner) order; the three g1==g2 exponentials
                // do not depend on (alpha, PC) and were taken at context creation
                double lk = 0;
                lk += e0 * gf[0] * gf2[0];
                lk += exp(a[0] + cst) * gf[0] * gf2[1];
                lk += exp(a[1] + cst) * gf[0] * gf2[2];
                lk += exp(a[2] + cst) * gf[1] * gf2[0];
                lk += e1 * gf[1] * gf2[1];
                lk += exp(a[3] + cst) * gf[1] * gf2[2];
                lk += exp(a[4] + cst) * gf[2] * gf2[0];
                lk += exp(a[5] + cst) * gf[2] * gf2[1];
                lk += e2 * gf[2] * gf2[2];
                if (lk > 0) llk_lane[b] = log(lk);
            }
        }
    }

    // ---- deterministic block reduction -> one partial per (point, block) ----
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        const double s = wave_sum(llk_lane[b]);
        if (lane == 0) red[b * kWavesPerBlock + wave] = s;
    }
    __syncthreads();
    if (tid < BT) {
        double s = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) s += red[tid * kWavesPerBlock + w];
        partials[(size_t)tid * gridDim.x + blockIdx.x] = s;
    }
}

// Sums the per-block partials in a fixed order -> bitwise reproducible: wave w takes
// points w, w+4, ...; lane l adds blocks l, l+64, ... (8 independent loads in flight),
// then a butterfly over the wave.
__global__ void __launch_bounds__(kBlockThreads)
llk_finalize_kernel(const double* __restrict__ partials, int nb, int num_point,
                    double* __restrict__ llk_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < num_point; b += kWavesPerBlock) {
        const double* p = partials + (size_t)b * nb;
        double s = 0;
        for (int base = 0; base < nb; base += 8 * 64) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 64 + lane;
                x[u] = i < nb ? p[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        s = wave_sum(s);
        if (lane == 0) llk_out[b] = s;
    }
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
template <int BT>
static hipError_t launch_bt(const DeviceLayout& L, const double* d_points, double* d_partials,
                            hipStream_t stream)
{
    const int nb = num_blocks_for(L);
    const size_t shmem =
        sizeof(double) * (size_t)((L.num_code + 1) * row_stride(BT) + BT * kWavesPerBlock +
                                  BT * (2 * L.num_pc + 1));
    hipLaunchKernelGGL((llk_eval_kernel<BT>), dim3(nb), dim3(kBlockThreads), shmem, stream, L,
                       d_points, d_partials);
    return hipGetLastError();
}

int max_points_per_launch() { return 8; }

hipError_t launch_llk_eval(const DeviceLayout& L, int num_point, const double* d_points,
                           double* d_partials, double* d_out, hipStream_t stream)
{
    const int stride = 2 * L.num_pc + 1;
    const int nb = num_blocks_for(L);
    int done = 0;
    while (done < num_point) {
        const int left = num_point - done;
        hipError_t e;
        const double* p = d_points + (size_t)done * stride;
        int step;
        if (left >= 8)      { e = launch_bt<8>(L, p, d_partials, stream); step = 8; }
        else if (left >= 4) { e = launch_bt<4>(L, p, d_partials, stream); step = 4; }
        else if (left >= 2) { e = launch_bt<2>(L, p, d_partials, stream); step = 2; }
        else                { e = launch_bt<1>(L, p, d_partials, stream); step = 1; }
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(llk_finalize_kernel, dim3(1), dim3(kBlockThreads), 0, stream, d_partials,
                           nb, step, d_out + done);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        done += step;
    }
    return hipSuccess;
}

// Zero-marker case: LLK of an empty sum.
__global__ void fill_zero_kernel(double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0.0;
}

hipError_t launch_fill_zero(double* d_out, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_zero_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_out, n);
    return hipGetLastError();
}

}  // namespace vb2
