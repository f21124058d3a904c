// Stand-alone anatomy of the big-tile row GEMM (k_rowgemm_big of gcn-bmp_amd/csrc/bmp_gemm.hip): the same main loop with
// parts switched off, timed with HIP events.  Diagnostic, not product.
//   hipcc --offload-arch=gfx950 -O3 -I gcn-bmp_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench && tools/gemm_bench [K] [Nout]
#include <stdio.h>
#include <stdlib.h>
#include "bmp_common.h"

#define LDA 36
#define A_FLOATS (256 * LDA)
#define B_FLOATS (8 * 256 * 4)
#define SWZ(n) ((n) ^ (((n) >> 4) & 3))

// MODE bits: 1 no global loads, 2 no LDS stores, 4 no LDS reads, 8 no barrier, 16 fragment double-buffer, 32 stagger waves 4-7,
//            64 waves 4-7 at s_setprio 1
template <int MODE>
__global__ __launch_bounds__(512) void k_big(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y, int N,
                                             int K, int Nout) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.y * 256, n0 = blockIdx.x * 256;
    constexpr int BUF = A_FLOATS + B_FLOATS;
    f32x16 acc[2][4];
    for (int rb = 0; rb < 2; ++rb) for (int cb = 0; cb < 4; ++cb) for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
    f32x4 sa[4], sb[4];
    for (int i = 0; i < 4; ++i) { sa[i] = (f32x4){1.f, 2.f, 3.f, 4.f}; sb[i] = (f32x4){.5f, .25f, .125f, 1.f}; }
    if (MODE & 128) {          // pseudo-random register operands: the data-dependent (power / clock) share without any memory traffic
        unsigned h = (tid + 977u * blockIdx.x + 131071u * blockIdx.y) * 2654435761u;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; sa[i][j] = ((h & 0xFFFF) / 32768.0f - 1.0f) * 0.7f;
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; sb[i][j] = ((h & 0xFFFF) / 32768.0f - 1.0f) * 0.7f;
            }
    }
    const int bk4 = tid >> 6, bnq = tid & 63;
    if (MODE & 64) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }
    auto load_chunk = [&](int k0_) {
        if (MODE & 1) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 512;
            const int r = idx >> 3, c4 = idx & 7;
            int row = row0 + r; row = row < N ? row : N - 1;
            sa[it] = *(const f32x4*)(X + (size_t)row * K + k0_ + 4 * c4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) sb[t] = *(const f32x4*)(Wt + (size_t)(k0_ + 4 * bk4 + t) * Nout + n0 + 4 * bnq);
    };
    auto store_chunk = [&](int buf) {
        if (MODE & 2) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 512;
            *(f32x4*)(&lds[buf * BUF + (idx >> 3) * LDA + 4 * (idx & 7)]) = sa[it];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            *(f32x4*)(&lds[buf * BUF + A_FLOATS + ((bk4 * 256) + SWZ(4 * bnq + jj)) * 4]) = (f32x4){sb[0][jj], sb[1][jj], sb[2][jj], sb[3][jj]};
    };
    const int nchunks = K / 32;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    if (MODE & 32) { if (w >= 4) __builtin_amdgcn_s_sleep(100); }
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) load_chunk((c + 1) * 32);
        const float* la = lds + buf * BUF + (wr * 64 + l31) * LDA + 4 * hi;
        const float* lb = lds + buf * BUF + A_FLOATS + (size_t)hi * 256 * 4;
        if (MODE & 16) {
            f32x4 a0[2], a1[2], b0[4], b1[4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) a0[rb] = (MODE & 4) ? sa[rb] : *(const f32x4*)(la + rb * 32 * LDA);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) b0[cb] = (MODE & 4) ? sb[cb] : *(const f32x4*)(lb + ((size_t)SWZ(wc * 128 + cb * 32 + l31)) * 4);
#pragma unroll
            for (int kk = 0; kk < 32; kk += 8) {
                if (kk + 8 < 32) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) a1[rb] = (MODE & 4) ? sa[rb] : *(const f32x4*)(la + rb * 32 * LDA + kk + 8);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        b1[cb] = (MODE & 4) ? sb[cb] : *(const f32x4*)(lb + ((size_t)((kk + 8) >> 2) * 256 + SWZ(wc * 128 + cb * 32 + l31)) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = bmp_mfma(a0[rb][t], b0[cb][t], acc[rb][cb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a0[rb] = a1[rb];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) b0[cb] = b1[cb];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 32; kk += 8) {
                f32x4 a0[2], b0[4];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a0[rb] = (MODE & 4) ? sa[rb] : *(const f32x4*)(la + rb * 32 * LDA + kk);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    b0[cb] = (MODE & 4) ? sb[cb] : *(const f32x4*)(lb + ((size_t)(kk >> 2) * 256 + SWZ(wc * 128 + cb * 32 + l31)) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = bmp_mfma(a0[rb][t], b0[cb][t], acc[rb][cb]);
            }
        }
        if (more) store_chunk(buf ^ 1);
        if (!(MODE & 8)) __syncthreads();
    }
    float s = 0.f;
    for (int rb = 0; rb < 2; ++rb) for (int cb = 0; cb < 4; ++cb) for (int i = 0; i < 16; ++i) s += acc[rb][cb][i];
    Y[(size_t)blockIdx.y * gridDim.x * 512 + blockIdx.x * 512 + tid] = s + sa[0][0] + sb[0][0];
}

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xFFFF) / 32768.0f - 1.0f) * 0.7f;
    }
}

template <int MODE>
static void run(const char* name, const float* X, const float* W, float* Y, int N, int K, int Nout) {
    const size_t lds_bytes = (size_t)2 * (A_FLOATS + B_FLOATS) * sizeof(float);
    hipFuncSetAttribute((const void*)k_big<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    dim3 grid((Nout + 255) / 256, (N + 255) / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_big<MODE>), grid, dim3(512), lds_bytes, 0, X, W, Y, N, K, Nout);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_big<MODE>), grid, dim3(512), lds_bytes, 0, X, W, Y, N, K, Nout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.1f us  %6.1f TFLOP/s\n", name, ms * 1e3, 2.0 * N * K * (double)Nout / ms / 1e9);
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1024, Nout = argc > 2 ? atoi(argv[2]) : 256, N = 455 * 128;
    float *X, *W, *Y;
    hipMalloc(&X, (size_t)N * K * 4); hipMalloc(&W, (size_t)K * Nout * 4); hipMalloc(&Y, (size_t)N * Nout * 4 + (1 << 22));
    // random operands: zero-filled inputs run at a higher clock (MI355X_MICROARCH.md: +19 % on zeros) and flatter the kernel
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, X, (size_t)N * K, 1u);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, W, (size_t)K * Nout, 7u);
    if (argc > 3) { hipMemset(X, 0, (size_t)N * K * 4); hipMemset(W, 0, (size_t)K * Nout * 4); printf("(zero operands)\n"); }
    printf("N=%d K=%d Nout=%d\n", N, K, Nout);
    run<0>("full", X, W, Y, N, K, Nout);
    run<16>("full + fragment double-buffer", X, W, Y, N, K, Nout);
    run<1>("no global loads", X, W, Y, N, K, Nout);
    run<1 | 2>("no global loads, no LDS stores", X, W, Y, N, K, Nout);
    run<1 | 2 | 8>("no loads, no stores, no barrier", X, W, Y, N, K, Nout);
    run<1 | 2 | 4 | 8>("MFMA only", X, W, Y, N, K, Nout);
    run<1 | 2 | 4 | 8 | 16>("MFMA only (double-buffer form)", X, W, Y, N, K, Nout);
    run<1 | 128>("no global loads, RANDOM operands", X, W, Y, N, K, Nout);
    run<1 | 2 | 4 | 8 | 128>("MFMA only, RANDOM operands", X, W, Y, N, K, Nout);
    run<32>("full + stagger", X, W, Y, N, K, Nout);
    run<64>("full + waves 4-7 prio 1", X, W, Y, N, K, Nout);
    run<16 | 64>("full + frag db + prio", X, W, Y, N, K, Nout);
    return 0;
}
