// wave_simd.hip -- which SIMD of its CU does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4],
// cu_id [11:8], se_id [15:13]; XCC_ID in HW_REG_XCC_ID [3:0])
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wave_simd wave_simd.hip && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k(unsigned* out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main()
{
    const int nb = 256;
    unsigned* d; hipMalloc(&d, nb * 16 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 120 * 1024, 0, d);
    static unsigned h[256 * 16 * 2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 100, 255}) {
        printf("workgroup %3d (xcc %u, se %u, cu %2u): simd of waves 0..15:", b, h[b * 32 + 1] & 0xf, (h[b * 32] >> 13) & 7, (h[b * 32] >> 8) & 0xf);
        for (int w = 0; w < 16; ++w) printf(" %u", (h[(b * 16 + w) * 2] >> 4) & 3);
        printf("\n");
    }
    int other = 0;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 16; ++w) other += ((h[(b * 16 + w) * 2] >> 4) & 3) != (unsigned)(w & 3);
    printf("waves whose simd is not (wave index mod 4): %d of %d\n", other, nb * 16);
    return 0;
}
