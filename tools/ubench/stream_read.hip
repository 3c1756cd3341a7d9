// Micro-benchmark: HBM read bandwidth by access shape, at the occupancy of a cohort step (one 1024-thread workgroup per CU).
// A 384 MB buffer (past the 256 MB Infinity Cache) is read once per launch: each wave walks chunks of its own,
// U loads in flight per lane, either 8 or 16 bytes per lane, chunk order either contiguous per workgroup or interleaved
// across workgroups (the deal of a cohort step).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int BYTES, int U, int WAVES> __global__ void __launch_bounds__(WAVES * 64)
rd(const char* __restrict__ src, size_t bytes, unsigned long long* out, int interleave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t chunk = (size_t)BYTES * 64 * U;                   // bytes one wave takes per step
    const size_t nchunk = bytes / chunk;
    const size_t nwave_total = (size_t)gridDim.x * WAVES;
    unsigned long long acc = 0;
    for (size_t c = (size_t)blockIdx.x * WAVES + wave; c < nchunk; c += nwave_total) {
        // interleave: chunk c as dealt; else: each workgroup owns one contiguous range
        const size_t cc = interleave ? c : ((size_t)blockIdx.x * WAVES + wave) * (nchunk / nwave_total) + c / nwave_total;
        const char* p = src + cc * chunk + (size_t)lane * BYTES;
        if (BYTES == 8) {
            uint2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const uint2*>(p + (size_t)u * 64 * BYTES);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y;
        } else {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const uint4*>(p + (size_t)u * 64 * BYTES);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int BYTES, int U, int WAVES> void run(const char* src, size_t bytes, unsigned long long* out, int grid, int inter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rd<BYTES, U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, bytes, out, inter);
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((rd<BYTES, U, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, bytes, out, inter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%2d B/lane, %2d loads in flight, %2d waves x %4d workgroups, %s: %7.1f us per pass, %6.2f TB/s\n", BYTES, U, WAVES, grid,
           inter ? "interleaved" : "contiguous ", 1e3 * ms / R, bytes / (1e-3 * ms / R) / 1e12);
}
int main()
{
    const size_t bytes = (size_t)384 << 20;
    char* src; unsigned long long* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 8); hipMemset(src, 1, bytes);
    run<8, 8, 16>(src, bytes, out, 256, 1);  run<8, 8, 16>(src, bytes, out, 256, 0);
    run<8, 16, 16>(src, bytes, out, 256, 1); run<16, 4, 16>(src, bytes, out, 256, 1);
    run<16, 8, 16>(src, bytes, out, 256, 1); run<16, 8, 16>(src, bytes, out, 256, 0);
    run<16, 16, 16>(src, bytes, out, 256, 1);
    run<16, 8, 16>(src, bytes, out, 512, 1); run<16, 8, 8>(src, bytes, out, 1024, 1); run<16, 4, 4>(src, bytes, out, 2048, 1);
    run<8, 8, 4>(src, bytes, out, 2048, 1);
    return 0;
}
