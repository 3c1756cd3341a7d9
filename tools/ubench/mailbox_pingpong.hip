// Micro-benchmark: host <-> persistent kernel ping-pong latency (gfx950 box).
//   mode A: the command word lives in mapped HOST memory, the kernel polls it over PCIe;
//   mode B: the command word lives in fine-grained DEVICE memory that the host writes through
//           the PCIe BAR (if the platform maps it), the kernel polls local memory.
// The reply always goes to mapped host memory (posted write), where the host spins.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mailbox_pingpong.hip -o tools/ubench/mailbox_pingpong.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <csetjmp>
#include <csignal>
#include <unistd.h>

static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }

__global__ void pong(const unsigned long long* cmd, unsigned long long* reply, unsigned long long n, int sys_scope)
{
    for (unsigned long long seq = 1; seq <= n; ++seq) {
        unsigned long long t0 = wall_clock64();
        for (;;) {
            unsigned long long v = sys_scope
                ? __hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                : __hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == seq) break;
            if (wall_clock64() - t0 > 200000000ull) return;      // 2 s: give up
        }
        __hip_atomic_store(reply, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double run(unsigned long long* h_cmd_hostview, const unsigned long long* cmd_devview, const char* name)
{
    unsigned long long *h_reply, *d_reply;
    hipHostMalloc((void**)&h_reply, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d_reply, h_reply, 0);
    *h_reply = 0;
    const unsigned long long n = 20000;
    __atomic_store_n(h_cmd_hostview, 0ull, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, 0, cmd_devview, d_reply, n, 1);
    usleep(20000);
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long seq = 1; seq <= n; ++seq) {
        __atomic_store_n(h_cmd_hostview, seq, __ATOMIC_RELEASE);
        auto tw = std::chrono::steady_clock::now();
        while (__atomic_load_n(h_reply, __ATOMIC_ACQUIRE) != seq) {
            if (std::chrono::steady_clock::now() - tw > std::chrono::seconds(3)) { printf("%s: timeout\n", name); return -1; }
        }
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    hipDeviceSynchronize();
    printf("%-46s %.2f us per round trip\n", name, us);
    hipHostFree(h_reply);
    return us;
}

int main()
{
    int large_bar = 0;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    unsigned long long *h_cmd, *d_cmd;
    hipHostMalloc((void**)&h_cmd, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d_cmd, h_cmd, 0);
    run(h_cmd, d_cmd, "A: command in host memory (GPU polls PCIe)");

    unsigned long long* fg = nullptr;
    if (hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
        printf("fine-grained device allocation failed\n");
        return 0;
    }
    hipMemset(fg, 0, 4096);
    hipDeviceSynchronize();
    signal(SIGSEGV, on_fault);
    signal(SIGBUS, on_fault);
    if (sigsetjmp(g_jmp, 1) != 0) {
        printf("B: fine-grained device memory is NOT host-accessible here (fault on host write)\n");
        return 0;
    }
    *(volatile unsigned long long*)fg = 7;
    if (*(volatile unsigned long long*)fg != 7) { printf("B: read-back mismatch\n"); return 0; }
    signal(SIGSEGV, SIG_DFL);
    signal(SIGBUS, SIG_DFL);
    run(fg, fg, "B: command in device memory (host writes BAR)");
    return 0;
}
