// Micro-benchmark: issue cost of the integer / conversion instructions a 16-bit run word costs to decode (gfx950):
// v_mul_lo_u32, v_mad_u32_u24, v_bfe_u32, v_and_b32, v_lshl_add_u32, v_cvt_f64_u32, against v_fma_f64.
// 8 independent chains per lane, 1024 threads per CU-sized block, one block per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/int_rates.hip -o gpurun_out/int_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
template <int OP> __device__ __forceinline__ unsigned op(unsigned a, unsigned b)
{
    unsigned r;
    if (OP == 0) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else if (OP == 1) asm volatile("v_mad_u32_u24 %0, %1, %2, %2" : "=v"(r) : "v"(a), "v"(b));
    else if (OP == 2) asm volatile("v_bfe_u32 %0, %1, 8, 8\n\tv_add_u32 %0, %0, %2" : "=&v"(r) : "v"(a), "v"(b));
    else if (OP == 3) asm volatile("v_and_b32 %0, 0xff00ff, %1\n\tv_add_u32 %0, %0, %2" : "=&v"(r) : "v"(a), "v"(b));
    else if (OP == 4) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(a), "v"(b));
    else if (OP == 5) { double d; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a)); r = (unsigned)__double2hiint(d) + b; }
    else if (OP == 6) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else if (OP == 7) asm volatile("v_perm_b32 %0, %1, %2, %2" : "=v"(r) : "v"(a), "v"(b));
    else if (OP == 8) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    else r = a;
    return r;
}
template <int OP> __global__ void __launch_bounds__(1024) k(unsigned* out, unsigned long long* cyc)
{
    unsigned x[8];
    for (int j = 0; j < 8; ++j) x[j] = 1u + threadIdx.x + j;
    const unsigned b = 3u + (threadIdx.x & 7);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = op<OP>(x[j], b);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int instr)
{
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= 256;
    printf("%-34s %.2f ticks per wave-op per SIMD (%d instr per op; the counter runs at the 100 MHz..shader clock ratio of readcyclecounter)\n",
           name, c / (4.0 * ITER * 8), instr);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<6>("v_add_u32 (reference)", 1); run<0>("v_mul_lo_u32", 1); run<1>("v_mad_u32_u24", 1); run<2>("v_bfe_u32 + v_add_u32", 2);
    run<3>("v_and_b32 + v_add_u32", 2); run<4>("v_lshl_add_u32", 1); run<5>("v_cvt_f64_u32 + v_add_u32", 2); run<7>("v_perm_b32", 1);
    run<8>("v_add_u32_sdwa (byte select)", 1);
    return 0;
}
