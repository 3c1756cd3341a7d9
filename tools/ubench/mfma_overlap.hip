// Micro-benchmark: do v_mfma_f64_16x16x4_f64 and v_fma_f64 from different waves of one SIMD
// overlap (gfx950)?  1024 threads per workgroup, one workgroup per CU (4 waves per SIMD).
// MODE 0: all waves VALU fma; 1: all waves MFMA; 2: waves alternate (2 MFMA + 2 VALU per SIMD,
// each doing the same per-wave work as in modes 0/1).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_overlap.hip -o gpurun_out/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void __launch_bounds__(1024) k(double* out)
{
    const int wave = threadIdx.x >> 6;
    // waves w, w+4, w+8, w+12 share a SIMD (round-robin placement): role by (wave >> 2) & 1
    const bool mf = MODE == 1 || (MODE == 2 && ((wave >> 2) & 1));
    double s = 0;
    if (mf) {
        d4 c[4];
        for (int j = 0; j < 4; ++j) c[j] = d4{0, 0, 0, 0};
        const double a = 1.0 + 1e-3 * threadIdx.x, b = 0.5;
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    } else {
        double x[8];
        for (int j = 0; j < 8; ++j) x[j] = 1.0 + 1e-3 * (threadIdx.x + j);
        const double b = 0.9999 + 1e-9 * threadIdx.x;
        for (int i = 0; i < ITER * 8; ++i) {          // 64 fma per ITER step = 256 clk (4 clk each)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = fma(x[j], b, 1e-9);
        }
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name)
{
    double* out;
    hipMalloc(&out, 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms\n", name, ms);
    hipFree(out);
}
int main()
{
    // per wave: MFMA waves issue ITER*4 MFMAs (64 clk each if 32 flop/clk/SIMD... measured);
    // VALU waves issue ITER*64 fma (4 clk each)
    run<0>("all 4 waves/SIMD VALU fma");
    run<1>("all 4 waves/SIMD MFMA f64 16x16x4");
    run<2>("2 MFMA + 2 VALU waves per SIMD");
    printf("if the pipes overlap, mode 2 ~ max(mode0, mode1)/2; if they share, ~ (mode0+mode1)/2\n");
    return 0;
}
