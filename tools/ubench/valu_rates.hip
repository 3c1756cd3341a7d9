// Micro-benchmark: issue cost of the FP64 VALU instructions the likelihood kernel uses
// (gfx950).  Each kernel runs ITER iterations of 8 independent chains of one operation,
// 1024 threads per CU-sized block, one block per CU; cycles per wave-instruction per SIMD
// are derived from the elapsed time and the clock measured with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
template <int OP> __device__ __forceinline__ double op(double a, double b, int i)
{
    if (OP == 0) return fma(a, b, 1e-9);
    if (OP == 1) return a * b;
    if (OP == 2) return a + b;
    if (OP == 3) return ldexp(a, (i & 1) ? 1 : -1);
    if (OP == 4) return rint(a * 1.0000001);            // mul + rndne
    if (OP == 5) return fmax(a, b);
    if (OP == 6) return (double)__builtin_amdgcn_frexp_exp(a) + b;   // frexp_exp + cvt + add
    if (OP == 7) return __builtin_amdgcn_frexp_mant(a) + b;          // frexp_mant + add
    if (OP == 8) return __builtin_amdgcn_rcp(a);
    if (OP == 9) return (double)(int)a + b;                          // cvt_i32_f64 + cvt_f64_i32 + add
    return a;
}
template <int OP> __global__ void __launch_bounds__(1024) k(double* out, unsigned long long* cyc)
{
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = 1.0 + 1e-3 * (threadIdx.x + j);
    const double b = 0.9999 + 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = op<OP>(x[j], b, i);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int extra)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= 256;
    // per SIMD: 4 waves x ITER x 8 ops (x extra instructions per op)
    printf("%-28s %.3f ms, %.0f shader-clock ticks, %.2f ticks per wave-op per SIMD (%d instr/op)\n", name, ms, c,
           c / (4.0 * ITER * 8), extra);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("v_fma_f64", 1); run<1>("v_mul_f64", 1); run<2>("v_add_f64", 1); run<3>("v_ldexp_f64", 1);
    run<4>("v_mul+v_rndne_f64", 2); run<5>("v_max_f64 (fmax)", 1); run<6>("frexp_exp+cvt+add", 3);
    run<7>("frexp_mant+add", 2); run<8>("v_rcp_f64", 1); run<9>("cvt_i32+cvt_f64+add", 3);
    return 0;
}
