// Micro-benchmark: is an XCD of this device slower than the others?  One 1024-thread workgroup per CU runs the same FP64 FMA +
// LDS-read loop; every workgroup records its XCC id (HW_REG_XCC_ID) and the wall-clock (s_memtime, 100 MHz) the loop took.
// Printed: median by XCC id, and by workgroup index mod 8 (the dispatcher deals workgroups round-robin over the XCDs).
// Several repetitions, so that a pattern can be told from noise.  (two of eight XCDs finish a 48-point launch's
// tiles 5 % later; which two differs from box to box.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_speed.hip -o gpurun_out/xcd_speed
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) k(double* out, unsigned long long* ticks, unsigned* xcc, int iters)
{
    __shared__ double tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) tab[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double a[6] = {1, 2, 3, 4, 5, 6};
    const double n = 1.0 + 1e-12 * threadIdx.x;
    unsigned idx = threadIdx.x * 7u;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        const double t = tab[idx & 4095u];
        idx = idx * 5u + 1u;
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = fma(n, t, a[j]);
    }
    const unsigned long long t1 = wall_clock64();
    double s = 0;
    for (int j = 0; j < 6; ++j) s += a[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x] = t1 - t0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf;
    }
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int nb = p.multiProcessorCount;
    double* out; unsigned long long* ticks; unsigned* xcc;
    hipMalloc(&out, sizeof(double) * nb * 1024); hipMalloc(&ticks, 8 * nb); hipMalloc(&xcc, 4 * nb);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 0, 0, out, ticks, xcc, 3000);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 4; ++rep) {
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 0, 0, out, ticks, xcc, 3000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(nb); std::vector<unsigned> x(nb);
        hipMemcpy(t.data(), ticks, 8 * nb, hipMemcpyDeviceToHost); hipMemcpy(x.data(), xcc, 4 * nb, hipMemcpyDeviceToHost);
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        std::printf("rep %d, us per workgroup: by XCC id:", rep);
        for (unsigned c = 0; c < 8; ++c) { std::vector<double> v; for (int b = 0; b < nb; ++b) if (x[b] == c) v.push_back(t[b] / 100.0); std::printf(" %.2f(%zu)", med(v), v.size()); }
        std::printf("   by index mod 8:");
        int agree = 0;
        for (int c = 0; c < 8; ++c) { std::vector<double> v; for (int b = c; b < nb; b += 8) { v.push_back(t[b] / 100.0); agree += x[b] == x[c]; } std::printf(" %.2f", med(v)); }
        std::printf("   (workgroups whose XCC id equals that of index mod 8's first: %d of %d)\n", agree, nb);
    }
    return 0;
}
