// Micro-benchmark for the shared-panel idea: a cohort step streams 356 MB from HBM of which 128 MB are every
// sample's own copy of its panel rows (5 doubles per marker).  Variant B streams 240 MB and GATHERS the rows instead:
// 3.2 M random 64-byte records (5 doubles used) from ONE 6.4 MB table, which should live in the Infinity Cache / L2.
// One 1024-thread workgroup per CU, like a cohort step.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_panel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(1024)
k(const char* __restrict__ stream, size_t stream_bytes, const double* __restrict__ panel, const unsigned* __restrict__ idx,
  size_t ngather, int do_gather, double* out)
{
    const size_t gtid = (size_t)blockIdx.x * 1024 + threadIdx.x, nthr = (size_t)gridDim.x * 1024;
    double acc = 0;
    // streaming part: 8 bytes per lane, 8 loads in flight
    const size_t nword = stream_bytes / 8;
    const uint2* s = reinterpret_cast<const uint2*>(stream);
    size_t gi = gtid;
    for (size_t w = gtid; w + 7 * nthr < nword; w += 8 * nthr) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[w + (size_t)u * nthr];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)(v[u].x ^ v[u].y);
        if (do_gather) {                         // one gather per 8 streamed words keeps the two interleaved
            for (int r = 0; r < 1 && gi < ngather; ++r, gi += nthr) {
                const double* p = panel + (size_t)idx[gi] * 8;
                acc += p[0] + p[1] + p[2] + p[3] + p[4];
            }
        }
    }
    if (do_gather)
        for (; gi < ngather; gi += nthr) {
            const double* p = panel + (size_t)idx[gi] * 8;
            acc += p[0] + p[1] + p[2] + p[3] + p[4];
        }
    if (acc == 1.2345) out[0] = acc;
}
int main()
{
    const size_t M = 100000, S = 32, ngather = M * S;
    const size_t bytes_a = (size_t)356 << 20, bytes_b = (size_t)228 << 20;      // B: minus 128 MB of panel-row copies
    char* stream; double* panel; unsigned* idx; double* out;
    hipMalloc(&stream, bytes_a); hipMemset(stream, 1, bytes_a);
    hipMalloc(&panel, M * 64); hipMemset(panel, 0, M * 64);
    hipMalloc(&out, 8);
    std::vector<unsigned> h(ngather);
    unsigned x = 12345;
    for (size_t i = 0; i < ngather; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x >> 8) % M; }
    hipMalloc(&idx, ngather * 4); hipMemcpy(idx, h.data(), ngather * 4, hipMemcpyHostToDevice);
    auto run = [&](const char* name, size_t sb, int g) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, stream, sb, panel, idx, ngather, g, out);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, stream, sb, panel, idx, ngather, g, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %7.1f us per pass\n", name, 1e3 * ms / 20);
    };
    run("A: stream 356 MB", bytes_a, 0);
    run("B: stream 228 MB + index 12.8 MB + gather 3.2 M x 64 B", bytes_b, 1);
    run("   stream 228 MB alone", bytes_b, 0);
    run("   gather alone (stream 8 MB)", (size_t)8 << 20, 1);
    return 0;
}
