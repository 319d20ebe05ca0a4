// Microbenchmark: the effective shader clock a latency-bound kernel sees.  One wave per workgroup runs a dependent fp64
// FMA chain and reads SHADER_CYCLES (hwreg 29, 20 bits) and the constant 100 MHz clock around it.
//   hipcc --offload-arch=gfx950 -O3 shader_clock.hip -o bin/shader_clock && bin/shader_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <thread>
__global__ void chain(double *out, unsigned long long *rec, double a, double b, int iters) {
    double x = out[threadIdx.x];
    const unsigned long long w0 = wall_clock64();
    const unsigned c0 = __builtin_amdgcn_s_getreg((19 << 11) | 29);      // SHADER_CYCLES[19:0]
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, a, b);
    }
    const unsigned c1 = __builtin_amdgcn_s_getreg((19 << 11) | 29);
    const unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { rec[0] = (c1 - c0) & 0xfffff; rec[1] = w1 - w0; }
}
int main() {
    double *d; unsigned long long *r;
    hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    hipHostMalloc(&r, 16);
    const int iters = 512;      // 8192 dependent FMAs: ~22 us
    auto run = [&](const char *what, int blocks, int reps) {
        double mhz = 0, ns = 0;
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, r, 1.0000001, 1e-9, iters);
            hipDeviceSynchronize();
            mhz = (double)r[0] / ((double)r[1] / 100.0);
            ns = (double)r[1] * 10.0 / (iters * 16);
        }
        printf("%-44s shader clock %7.1f MHz   %5.2f ns per dependent fp64 FMA (%4.2f cycles)\n", what, mhz, ns, ns * mhz / 1000.0);
    };
    run("cold, 1 workgroup, first launch", 1, 1);
    run("1 workgroup, after 200 launches", 1, 200);
    run("256 workgroups, after 200 launches", 256, 200);
    run("1024 workgroups, after 200 launches", 1024, 200);
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    run("after 0.5 s idle, 256 workgroups, 1 launch", 256, 1);
    run("256 workgroups, 2000 launches", 256, 2000);
    return 0;
}
