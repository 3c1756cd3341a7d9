// Micro-benchmark: can the read loop's instruction mix -- per run 6 x ds_read_b128 (per-lane table rows),
// 12 x v_fma_f64, 2 integer ops -- keep the LDS pipe and the FP64 VALU busy at the same time (gfx950)?
// One 1024-thread (or smaller) workgroup per CU; every wave runs ITER "runs"; variants: the FMAs alone, the
// LDS reads alone, both; waves per SIMD 1..4.  Prints shader-clock ticks per run per wave and the implied busy
// fractions (a wave64 FP64 FMA holds its SIMD 4 cycles; a ds_read_b128 holds the CU's LDS 4 cycles).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/lds_fma_mix.hip -o gpurun_out/lds_fma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 16384
typedef double __attribute__((ext_vector_type(2))) vdouble2;
typedef __attribute__((address_space(3))) const vdouble2 lds_cdouble2;

// MIX: 1 = FMAs, 2 = LDS reads, 3 = both (the reads feed the FMAs, as in the kernel)
template <int MIX> __global__ void __launch_bounds__(1024) k(double* out, unsigned long long* cyc, const unsigned* codes)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int i = threadIdx.x; i < 43 * 50; i += blockDim.x) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds;
    double acc[12], mul[12];
    for (int j = 0; j < 12; ++j) { acc[j] = 0.0; mul[j] = 1.0 + 0.01 * j + 1e-6 * threadIdx.x; }
    unsigned seq[8];                                   // eight per-lane "run words": row offset (x400) | count bits
    for (int u = 0; u < 8; ++u) seq[u] = codes[(threadIdx.x & 63) * 8 + u];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned rw = seq[u];
            const unsigned off = (rw & 0xffffu);
            const double n = __hiloint2double((int)(rw & 0xffff0000u), 0);
            if (MIX & 2) {
                lds_cdouble2* row = reinterpret_cast<lds_cdouble2*>(base + off);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const vdouble2 t = row[q];
                    if (MIX & 1) {
                        acc[2 * q] = fma(n, t.x, acc[2 * q]);
                        acc[2 * q + 1] = fma(n, t.y, acc[2 * q + 1]);
                    } else {
                        asm volatile("" ::"v"(t));         // (keeps the load alive)
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 12; ++q) acc[q] = fma(n, mul[q], acc[q]);
            }
        }
        asm volatile("" : "+v"(seq[0]), "+v"(seq[1]), "+v"(seq[2]), "+v"(seq[3]), "+v"(seq[4]), "+v"(seq[5]), "+v"(seq[6]), "+v"(seq[7]));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int j = 0; j < 12; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MIX> void run(const char* name, int waves)
{
    double* out; unsigned long long* cyc; unsigned* codes;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8); hipMalloc(&codes, 64 * 8 * 4);
    std::vector<unsigned> hc(64 * 8);
    // lane l, run u: markers of a wave step through neighbouring codes (like the sorted run lists of the kernel)
    for (int l = 0; l < 64; ++l)
        for (int u = 0; u < 8; ++u) hc[l * 8 + u] = (unsigned)((5 * u + (l & 15) + (l >> 4) * 3) % 42) * 400u | 0x40080000u;
    hipMemcpy(codes, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(64 * waves), 43 * 50 * 8, 0, out, cyc, codes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(64 * waves), 43 * 50 * 8, 0, out, cyc, codes);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= 256;
    const double per_run = c / ITER;                       // ticks per run per wave (all waves run concurrently)
    const double wps = waves / 4.0;                        // waves per SIMD
    // busy fractions: VALU: waves-per-SIMD x 12 FMA x 4 cycles per run-time; LDS: waves x 6 reads x 4 cycles per run-time
    const double ns_per_run = 1e6 * ms / ITER;             // wall-clock (kernel incl. launch) per run
    printf("%-10s %2d waves/CU: %7.1f ticks = %6.1f ns per run per wave (%.2f ticks/ns) | at 2.4 GHz: VALU busy %.2f  LDS busy %.2f\n",
           name, waves, per_run, ns_per_run, per_run / ns_per_run,
           (MIX & 1) ? wps * 12 * 4 / (ns_per_run * 2.4) : 0.0, (MIX & 2) ? waves * 6 * 4 / (ns_per_run * 2.4) : 0.0);
    hipFree(out); hipFree(cyc); hipFree(codes);
}
int main()
{
    for (int w : {4, 8, 12, 16}) run<1>("fma", w);
    for (int w : {4, 8, 12, 16}) run<2>("lds", w);
    for (int w : {4, 8, 12, 16}) run<3>("lds+fma", w);
    return 0;
}
