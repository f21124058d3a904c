// Micro-benchmark (diagnostic, not product): what f32-input MFMA rate and shader clock does THIS device
// sustain?  Gives the practical ceiling for the kernels in gcn-bmp_amd/csrc.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_peak(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f + 0.37f, b = 1.0f - threadIdx.x * 2e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256, threads = 512, iters = 20000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[512]; hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
        double flops = (double)blocks * (threads / 64) * iters * 32.0 * 4096.0;
        printf("rep %d: %.3f ms, %.1f TFLOP/s, shader clock %.3f GHz (memtime/memrealtime*0.1)\n", rep, ms,
               flops / ms / 1e9, cyc / real * 0.1);
    }
    return 0;
}
