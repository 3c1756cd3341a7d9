#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __global__ void k(int* out) { int x = threadIdx.x; out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xf, 0xf, false); }
__global__ void ksw16(int* out) {
    int a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
__global__ void ksw32(int* out) {
    int a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
template <int CTRL> void run(const char* name) {
    int* d; hipMalloc(&d, 64 * 4); k<CTRL><<<1, 64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", name); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf("\n"); hipFree(d);
}
int main() {
    run<0x124>("row_ror4"); run<0x12C>("row_ror12"); run<0x128>("row_ror8"); run<0x104>("row_shl4"); run<0x114>("row_shr4");
    run<0x140>("row_mirror"); run<0x141>("row_half_mirror");
    int* d; hipMalloc(&d, 128 * 4); int h[128];
    ksw16<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("permlane16_swap r0:"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n r1:"); for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]); printf("\n");
    ksw32<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("permlane32_swap r0:"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n r1:"); for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}
