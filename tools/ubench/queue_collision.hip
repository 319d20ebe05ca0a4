// Microbenchmark: does a kernel that spins on one stream delay the COMPLETION of short kernels on another stream?
// (What the sharded step's collective streams do to the caller's rollout stream.)
//   hipcc --offload-arch=gfx950 -O3 queue_collision.hip -o bin/queue_collision && bin/queue_collision [n_streams] [prio]
// For stream j of n: a one-lane kernel spins on stream j until a flag is raised (or 2 ms); meanwhile M short kernels run
// back to back on stream 0 (the "main" stream), the last one raises the flag.  Prints the time per short kernel for each j.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(const unsigned long long *flag, unsigned long long want, unsigned long long ticks) {
    if (threadIdx.x) return;
    const unsigned long long give_up = wall_clock64() + ticks;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() > give_up) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
__global__ void busy(double *x, int iters) {       // ~ a few us of dependent FMAs on 256 workgroups
    double v = x[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = __builtin_fma(v, 1.0000001, 1e-9);
    x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void raise(unsigned long long *flag, unsigned long long v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 12;
    const int prio_mode = argc > 2 ? atoi(argv[2]) : 0;      // 0: all normal; 1: others high priority
    const int M = 16;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::vector<hipStream_t> st(n);
    for (int j = 0; j < n; ++j) CK(hipStreamCreateWithPriority(&st[j], hipStreamNonBlocking, (prio_mode && j) ? hi : 0));
    unsigned long long *flag; double *x;
    CK(hipMalloc(&flag, 8)); CK(hipMemset(flag, 0, 8));
    CK(hipMalloc(&x, 256 * 256 * 8)); CK(hipMemset(x, 0, 256 * 256 * 8));
    // touch every stream once so each is bound to its hardware queue
    for (int j = 0; j < n; ++j) { hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, st[j], x, 10); CK(hipStreamSynchronize(st[j])); }
    unsigned long long seq = 0;
    for (int rep = 0; rep < 2; ++rep)
    for (int j = 0; j < n; ++j) {
        ++seq;
        auto t0 = std::chrono::steady_clock::now();
        if (j) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], flag, seq, 200000ULL);     // 2 ms of the 100 MHz clock
        for (int m = 0; m < M; ++m) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, st[0], x, 2000);
        hipLaunchKernelGGL(raise, dim3(1), dim3(1), 0, st[0], flag, seq);
        CK(hipStreamSynchronize(st[0]));
        auto t1 = std::chrono::steady_clock::now();
        if (j) CK(hipStreamSynchronize(st[j]));
        printf("rep %d spinner on stream %2d: %7.2f us per short kernel on stream 0\n", rep, j, std::chrono::duration<double, std::micro>(t1 - t0).count() / M);
    }
    return 0;
}
