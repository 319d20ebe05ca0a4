// Microbenchmark: issue/latency of fp64 VALU for ONE wave on a SIMD (what the sequential phase of
// the rollout kernel sees).  hipcc --offload-arch=gfx950 -O3 fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void dep_chain(double *out, unsigned long long *cyc, double a, double b, int iters) {
    double x = out[threadIdx.x];
    unsigned long long w0 = wall_clock64();               // s_memrealtime: constant 100 MHz
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, a, b);
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void indep4(double *out, unsigned long long *cyc, double a, double b, int iters) {
    double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void dep_chain_f32(float *out, unsigned long long *cyc, float a, float b, int iters) {
    float x = out[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fmaf(x, a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; unsigned long long *c; float *f;
    hipMalloc(&d, 64 * 8); hipMalloc(&f, 64 * 4); hipMalloc(&c, 16);
    hipMemset(d, 0, 64 * 8); hipMemset(f, 0, 64 * 4);
    unsigned long long h;
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(dep_chain, 1, 64, 0, 0, d, c, 0.999, 0.001, iters); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) {
            unsigned long long hw[2]; hipMemcpy(hw, c, 16, hipMemcpyDeviceToHost);
            printf("fp64 dependent fma : %.2f s_memtime ticks/op = %.2f ns/op (s_memrealtime, 100 MHz); s_memtime runs at %.0f MHz\n",
                   (double)hw[0] / (iters * 16), 10.0 * hw[1] / (iters * 16), 100.0 * hw[0] / hw[1]);
        }
        hipLaunchKernelGGL(indep4, 1, 64, 0, 0, d, c, 0.999, 0.001, iters); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("fp64 4 indep chains: %.2f cycles/op\n", (double)h / (iters * 16));
        hipLaunchKernelGGL(dep_chain_f32, 1, 64, 0, 0, f, c, 0.999f, 0.001f, iters); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("fp32 dependent fma : %.2f cycles/op\n", (double)h / (iters * 16));
    }
    return 0;
}
