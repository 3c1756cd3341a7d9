// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the likelihood kernels use: a 1 GiB
// buffer (beyond the 256 MB Infinity Cache) is read once by each kernel -- 16 bytes per lane (the case
// MI355X_MICROARCH.md calibrates: FETCH_SIZE reports half), 8 bytes per lane coalesced (run words, per-marker
// doubles), 4 bytes per lane.  Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// and divide each kernel's FETCH_SIZE (KB) by 1 048 576.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void __launch_bounds__(256) rd(const T* __restrict__ p, size_t n, double* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) {
        const T v = p[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        for (unsigned j = 0; j < sizeof(T) / 4; ++j) s += w[j];
    }
    if (s == 1.2345) out[0] = s;
}
int main()
{
    const size_t bytes = 1ull << 30;
    void* buf; double* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(rd<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(rd<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, out);
    }
    (void)hipDeviceSynchronize();
    printf("read 1 GiB three times with 16, 8 and 4 bytes per lane\n");
    return 0;
}
