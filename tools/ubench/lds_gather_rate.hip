// Micro-benchmark: what limits a per-lane table gather with ds_read_b128 on gfx950?  16 waves per CU, every wave
// issues READS reads of 16 B per lane per step from LDS rows chosen per lane, in one of several patterns, consumes
// them with one v_or each (no FP64), and the wall time gives LDS cycles per wave-instruction.
//   pattern 0: per-lane rows, neighbouring lanes neighbouring rows (the kernel's sorted run lists)
//   pattern 1: every lane the same row (broadcast)
//   pattern 2: rows identical within each 16-lane read group, different between groups
//   pattern 3: per-lane pseudo-random rows
// READS in flight per wave before the first use: 6 or 12 or 24.  WIDTH: 16 (b128) or 8 (b64, twice the instructions).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_gather_rate.hip -o gpurun_out/lds_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
typedef unsigned __attribute__((ext_vector_type(4))) vuint4;
typedef unsigned __attribute__((ext_vector_type(2))) vuint2;
typedef __attribute__((address_space(3))) const vuint4 lds_cuint4;
typedef __attribute__((address_space(3))) const vuint2 lds_cuint2;

template <int READS, int WIDTH> __global__ void __launch_bounds__(1024) k(unsigned* out, const unsigned* codes, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 128 * 128; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds;
    unsigned seq[8];
    for (int u = 0; u < 8; ++u) seq[u] = codes[(threadIdx.x & 63) * 8 + u] * (unsigned)stride;
    unsigned acc = 0;
    for (int i = 0; i < ITER / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = seq[u];
            if (WIDTH == 16) {
                lds_cuint4* row = reinterpret_cast<lds_cuint4*>(base + off);
                vuint4 t[READS];
#pragma unroll
                for (int q = 0; q < READS; ++q) t[q] = row[q];
#pragma unroll
                for (int q = 0; q < READS; ++q) acc |= t[q].x ^ t[q].w;
            } else {
                lds_cuint2* row = reinterpret_cast<lds_cuint2*>(base + off);
                vuint2 t[2 * READS];
#pragma unroll
                for (int q = 0; q < 2 * READS; ++q) t[q] = row[q];
#pragma unroll
                for (int q = 0; q < 2 * READS; ++q) acc |= t[q].x ^ t[q].y;
            }
        }
        asm volatile("" : "+v"(seq[0]), "+v"(seq[1]), "+v"(seq[2]), "+v"(seq[3]), "+v"(seq[4]), "+v"(seq[5]), "+v"(seq[6]), "+v"(seq[7]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int READS, int WIDTH> void run(int pattern, int waves, int stride)
{
    unsigned *out, *codes;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&codes, 64 * 8 * 4);
    std::vector<unsigned> hc(64 * 8);
    unsigned r = 12345;
    for (int l = 0; l < 64; ++l)
        for (int u = 0; u < 8; ++u) {
            unsigned row;
            switch (pattern) {
            case 0: row = (5 * u + (l & 15) + (l >> 4) * 3) % 42; break;
            case 1: row = (5 * u) % 42; break;
            case 2: row = (5 * u + 7 * (l >> 4)) % 42; break;
            default: r = r * 1664525u + 1013904223u; row = (r >> 8) % 42; break;
            }
            hc[l * 8 + u] = row;
        }
    hipMemcpy(codes, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shm = 128 * 128 * 4;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<READS, WIDTH>), dim3(256), dim3(64 * waves), shm, 0, out, codes, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<READS, WIDTH>), dim3(256), dim3(64 * waves), shm, 0, out, codes, stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = (double)ITER * READS * (WIDTH == 16 ? 1 : 2) * waves;     // wave-instructions per CU
    const double bytes = (double)ITER * READS * 16 * 64 * waves;
    printf("pattern %d  %2d reads/step b%-3d stride %4d  %2d waves/CU: %6.1f us  %5.2f cycles per wave-instr @2.4GHz  %6.1f B/clk/CU\n",
           pattern, READS, WIDTH * 8, stride, waves, 1e3 * ms, 1e-3 * ms * 2.4e9 / instr, bytes / (1e-3 * ms * 2.4e9));
    hipFree(out); hipFree(codes);
}
int main()
{
    for (int p = 0; p < 4; ++p) {
        run<6, 16>(p, 16, 400);
        run<12, 16>(p, 16, 400);
        run<6, 8>(p, 16, 400);
    }
    run<6, 16>(0, 16, 96);  run<6, 16>(0, 16, 112); run<6, 16>(0, 16, 208); run<6, 16>(3, 16, 112);
    run<24, 16>(0, 16, 400);
    for (int w : {4, 8, 12}) run<6, 16>(0, w, 400);
    for (int w : {4, 8, 12}) run<12, 16>(0, w, 400);
    return 0;
}
