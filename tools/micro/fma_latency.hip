// Dependent-issue latency of the vector ALU on one wave: CHAINS independent chains of fused multiply-adds, interleaved.
// hipcc -O3 --offload-arch=gfx950 -o fma_latency fma_latency.hip && ./fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int CHAINS, bool SHORT = false>
__global__ void chain_kernel(T *out, unsigned long long *ticks, int iters, T a, T b) {
    T x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = T(threadIdx.x + c) * T(1e-3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (SHORT) {          // VOP2 encodings (4 bytes): x += a * b
                    if constexpr (sizeof(T) == 8) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                    else asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                } else {                        // VOP3 encodings (8 bytes)
                    if constexpr (sizeof(T) == 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    T s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}
template <typename T, int CHAINS, bool SHORT = false> void run(const char *name, int threads) {
    T *out; unsigned long long *ticks, h[2];
    hipMalloc(&out, 64 * sizeof(T) * 8); hipMalloc(&ticks, 16);
    const int iters = 4096;
    for (int r = 0; r < 2; ++r) chain_kernel<T, CHAINS, SHORT><<<1, threads>>>(out, ticks, iters, T(0.999), T(1e-3));
    hipDeviceSynchronize();
    hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * CHAINS;
    printf("%-8s chains %d threads %3d: %6.2f s_memtime ticks / instruction, %7.3f ns / instruction (100 MHz clock)\n", name, CHAINS, threads,
           h[0] / n, h[1] * 10.0 / n);
    hipFree(out); hipFree(ticks);
}
int main() {
    run<double, 1>("fma_f64", 64); run<double, 2>("fma_f64", 64); run<double, 4>("fma_f64", 64);
    run<double, 1>("fma_f64", 16); run<double, 2>("fma_f64", 16);
    run<float, 1>("fma_f32", 64); run<float, 2>("fma_f32", 64); run<float, 4>("fma_f32", 64);
    // 4-byte encodings: is a lone wave's issue rate an instruction-fetch rate?
    run<double, 1, true>("fmac_f64", 64); run<double, 4, true>("fmac_f64", 64); run<double, 8, true>("fmac_f64", 64);
    run<float, 1, true>("fmac_f32", 64); run<float, 4, true>("fmac_f32", 64); run<float, 8, true>("fmac_f32", 64);
    run<double, 8>("fma_f64", 64);
    return 0;
}
