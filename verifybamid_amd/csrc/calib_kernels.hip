// calib_kernels.hip -- what this box's FP64 vector units and LDS sustain with the read loop's instruction mix, measured
// on demand (vb2_debug_issue_ceiling; bench.py quotes the headline kernel against it instead of a nominal clock).
//
// The evaluation kernels are bound by VALU issue, not by HBM (kernels): per run of a marker's reads a lane
// issues 6 ds_read_b128 and 12 v_fma_f64 (two points x six genotype pairs) and two integer operations.  One workgroup of 16
// waves per CU runs exactly that, (a) the arithmetic alone, (b) with the table reads feeding it; the answer is VALU
// lane-instructions per second over the whole device (every VALU instruction a wave issues counts 64).
// tools/ubench/lds_fma_mix.hip is the stand-alone version with more variants.
#include <hip/hip_runtime.h>

#include "llk_kernels.h"

namespace vb2 {

namespace {
typedef double __attribute__((ext_vector_type(2))) vdouble2;
typedef __attribute__((address_space(3))) const vdouble2 lds_cdouble2;
constexpr int kCalibRuns = 8192;        // runs per wave and launch
constexpr int kCalibRows = 43, kCalibRowDoubles = 50;

template <bool LDS>
__global__ void __launch_bounds__(1024) calib_kernel(double* out)
{
    __shared__ __attribute__((aligned(16))) double lds[kCalibRows * kCalibRowDoubles];
    for (int i = threadIdx.x; i < kCalibRows * kCalibRowDoubles; i += blockDim.x) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds;
    double acc[12], mul[12];
    for (int j = 0; j < 12; ++j) { acc[j] = 0.0; mul[j] = 1.0 + 0.01 * j + 1e-6 * threadIdx.x; }
    unsigned seq[8];                     // eight per-lane run words: row byte offset | top of double(count)
    const unsigned l = threadIdx.x & 63u;
    for (unsigned u = 0; u < 8; ++u) seq[u] = ((5u * u + (l & 15u) + (l >> 4) * 3u) % 42u) * 400u | 0x40080000u;
    for (int i = 0; i < kCalibRuns / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned rw = seq[u];
            const unsigned off = rw & 0xffffu;
            const double n = __hiloint2double((int)(rw & 0xffff0000u), 0);
            if (LDS) {
                lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(base + off);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const vdouble2 t = row[q];
                    acc[2 * q] = fma(n, t.x, acc[2 * q]);
                    acc[2 * q + 1] = fma(n, t.y, acc[2 * q + 1]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 12; ++q) acc[q] = fma(n, mul[q], acc[q]);
            }
        }
        asm volatile("" : "+v"(seq[0]), "+v"(seq[1]), "+v"(seq[2]), "+v"(seq[3]), "+v"(seq[4]), "+v"(seq[5]), "+v"(seq[6]), "+v"(seq[7]));
    }
    double s = 0;
    for (int j = 0; j < 12; ++j) s += acc[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

// out[0]: FP64 FMA lane-instructions per second, arithmetic alone; out[1]: the same with the table reads (6 ds_read_b128 per
// 12 FMAs); out[2]: VALU lane-instructions per second of the second variant counting its two integer operations per run too
hipError_t measure_issue_ceiling(int device, double out[3])
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return e;
    const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    double* d_out = nullptr;
    if ((e = hipMalloc(&d_out, (size_t)ncu * 1024 * sizeof(double))) != hipSuccess) return e;
    hipStream_t st;
    hipEvent_t e0, e1;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 8;
    for (int variant = 0; variant < 2 && e == hipSuccess; ++variant) {
        float best = 0.f;
        for (int pass = 0; pass < 3; ++pass) {               // (the first pass warms clocks and code)
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) hipLaunchKernelGGL(calib_kernel<false>, dim3(ncu), dim3(1024), 0, st, d_out);
                else hipLaunchKernelGGL(calib_kernel<true>, dim3(ncu), dim3(1024), 0, st, d_out);
            }
            (void)hipEventRecord(e1, st);
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (pass > 0 && (best == 0.f || ms < best)) best = ms;
        }
        const double fma_lane_instr = (double)reps * ncu * 16.0 * kCalibRuns * 12.0 * 64.0;
        out[variant] = best > 0.f ? fma_lane_instr / (best * 1e-3) : 0.0;
        if (variant == 1) out[2] = out[1] * 14.0 / 12.0;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(d_out);
    return e;
}

}  // namespace vb2
