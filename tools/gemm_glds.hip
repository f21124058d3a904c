// Standalone probe: the 128 x 128 row GEMM's staging pipeline, register-staged (what k_rowgemm_db does: global_load ->
// registers -> ds_write, weights transposed in registers) against LDS-DMA (global_load_lds_dwordx4: A rows into an
// XOR-swizzled linear image, the swizzle on the source address; weights from a K4-packed array [k/4][n][4], whose 32-deep
// chunk IS the LDS image).  Same MFMA loop, same fragment reads, same summation order: outputs must agree bitwise.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tmp/gemm_glds tools/gemm_glds.hip && tools/tmp/gemm_glds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define LDA 36
#define A_FLOATS_R (128 * LDA)
#define B_FLOATS (8 * 128 * 4)
#define SWZ(n) ((n) ^ (((n) >> 4) & 3))

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- (a) register staging -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reg(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                             int K, int Nout) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_FLOATS_R + B_FLOATS)];
    constexpr int BUF = A_FLOATS_R + B_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    f32x16 acc[4];
    for (int rb = 0; rb < 4; ++rb) for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
    f32x4 sa[4], sb[4];
    const int bk4 = tid >> 5, bnq = tid & 31;
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, r = idx >> 3, c4 = idx & 7;
            sa[it] = *(const f32x4*)(X + (size_t)(row0 + r) * K + k0 + 4 * c4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) sb[t] = *(const f32x4*)(Wt + (size_t)(k0 + 4 * bk4 + t) * Nout + n0 + 4 * bnq);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            *(f32x4*)(&lds[buf * BUF + (idx >> 3) * LDA + 4 * (idx & 7)]) = sa[it];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            *(f32x4*)(&lds[buf * BUF + A_FLOATS_R + (bk4 * 128 + SWZ(4 * bnq + jj)) * 4]) = (f32x4){sb[0][jj], sb[1][jj], sb[2][jj], sb[3][jj]};
    };
    const int nch = K >> 5;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int bslot = SWZ(wc * 32 + l31);
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nch;
        if (more) load_chunk((c + 1) << 5);
        const float* la = lds + buf * BUF + l31 * LDA + 4 * hi;
        const float* lb = lds + buf * BUF + A_FLOATS_R + ((size_t)hi * 128 + bslot) * 4;
        f32x4 a0[4], a1[4], b0, b1;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a0[rb] = *(const f32x4*)(la + rb * 32 * LDA);
        b0 = *(const f32x4*)lb;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 8) {
            if (kk + 8 < 32) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) a1[rb] = *(const f32x4*)(la + rb * 32 * LDA + kk + 8);
                b1 = *(const f32x4*)(lb + (size_t)((kk + 8) >> 2) * 128 * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb][t], b0[t], acc[rb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) a0[rb] = a1[rb];
            b0 = b1;
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    const int col = n0 + wc * 32 + l31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) Y[(size_t)(row0 + rb * 32 + acc_row(reg, lane)) * Nout + col] = acc[rb][reg];
}

// ---- (b) LDS-DMA staging ------------------------------------------------------------------------------------------
// A image: [128 rows][8 x 16 B], slot (r, c) holds X[r][k0 + 4 (c ^ (r & 7))].  B image: [8 k4][128 n][4] = the K4-packed
// chunk as it lies in memory.  NBUF buffers; chunk c + NBUF - 1 is requested while chunk c computes.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define A_FLOATS_G (128 * 32)

template <int NBUF>
__global__ __launch_bounds__(256) void k_glds(const float* __restrict__ X, const float* __restrict__ W4, float* __restrict__ Y,
                                              int K, int Nout) {
    __shared__ __attribute__((aligned(16))) float lds[NBUF * (A_FLOATS_G + B_FLOATS)];
    constexpr int BUF = A_FLOATS_G + B_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    f32x16 acc[4];
    for (int rb = 0; rb < 4; ++rb) for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
    // this wave's four A pieces (8 rows x 128 B each) and two B pieces (64 columns of one k4 row... x2) per chunk
    const int ar = lane >> 3, ac = lane & 7;
    auto request = [&](int c, int buf) {
        const int k0 = c << 5;
        float* base = lds + buf * BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = (wc * 4 + p) * 8 + ar;                                  // row of the tile
            const float* g = X + (size_t)(row0 + r) * K + k0 + 4 * (ac ^ (r & 7));
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(base + (wc * 4 + p) * 8 * 32), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wc * 4 + p;                                         // 16 pieces of 64 columns: k4 = piece / 2
            const int k4 = piece >> 1, half = piece & 1;
            const float* g = W4 + ((size_t)((k0 >> 2) + k4) * Nout + n0 + half * 64 + lane) * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(base + A_FLOATS_G + (k4 * 128 + half * 64) * 4), 16, 0, 0);
        }
    };
    const int nch = K >> 5;
    for (int c = 0; c < NBUF - 1 && c < nch; ++c) request(c, c);
    for (int c = 0; c < nch; ++c) {
        const int buf = c % NBUF;
        // chunk c must have landed: everything but the NBUF - 2 younger chunks' 8 pieces each
        if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + NBUF - 1 < nch) request(c + NBUF - 1, (c + NBUF - 1) % NBUF);     // its buffer was read in chunk c - 1
        const float* la = lds + buf * BUF + l31 * 32;
        const float* lb = lds + buf * BUF + A_FLOATS_G + ((size_t)hi * 128 + wc * 32 + l31) * 4;
        const int sw = l31 & 7;
        f32x4 a0[4], a1[4], b0, b1;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a0[rb] = *(const f32x4*)(la + rb * 32 * 32 + 4 * ((0 + hi) ^ sw));
        b0 = *(const f32x4*)lb;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 8) {
            if (kk + 8 < 32) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) a1[rb] = *(const f32x4*)(la + rb * 32 * 32 + 4 * ((((kk + 8) >> 2) + hi) ^ sw));
                b1 = *(const f32x4*)(lb + (size_t)((kk + 8) >> 2) * 128 * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb][t], b0[t], acc[rb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) a0[rb] = a1[rb];
            b0 = b1;
        }
    }
    const int col = n0 + wc * 32 + l31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) Y[(size_t)(row0 + rb * 32 + acc_row(reg, lane)) * Nout + col] = acc[rb][reg];
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; }

template <typename F>
static float time_ms(F launch, int iters) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int N = 450 * 128;
    const int shapes[5][2] = {{512, 768}, {1024, 256}, {768, 256}, {256, 1024}, {768, 512}};
    for (int s = 0; s < 5; ++s) {
        const int K = shapes[s][0], Nout = shapes[s][1];
        std::vector<float> hX((size_t)N * K), hW((size_t)K * Nout), hW4((size_t)K * Nout);
        unsigned seed = 17 + s;
        for (auto& v : hX) v = frand(seed);
        for (auto& v : hW) v = frand(seed);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < Nout; ++n) hW4[((size_t)(k >> 2) * Nout + n) * 4 + (k & 3)] = hW[(size_t)k * Nout + n];
        float *X, *W, *W4, *Y1, *Y2;
        CHECK(hipMalloc(&X, hX.size() * 4)); CHECK(hipMalloc(&W, hW.size() * 4)); CHECK(hipMalloc(&W4, hW.size() * 4));
        CHECK(hipMalloc(&Y1, (size_t)N * Nout * 4)); CHECK(hipMalloc(&Y2, (size_t)N * Nout * 4));
        CHECK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(W4, hW4.data(), hW.size() * 4, hipMemcpyHostToDevice));
        const dim3 grid(N / 128, Nout / 128);
        const double gf = 2.0 * N * K * (double)Nout / 1e9;
        const float t_reg = time_ms([&] { hipLaunchKernelGGL(k_reg, grid, dim3(256), 0, 0, X, W, Y1, K, Nout); }, 10);
        const float t_g2 = time_ms([&] { hipLaunchKernelGGL((k_glds<2>), grid, dim3(256), 0, 0, X, W4, Y2, K, Nout); }, 10);
        std::vector<float> y1((size_t)N * Nout), y2((size_t)N * Nout);
        CHECK(hipMemcpy(y1.data(), Y1, y1.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(y2.data(), Y2, y2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad2 = 0;
        for (size_t i = 0; i < y1.size(); ++i) bad2 += y1[i] != y2[i];
        const float t_g3 = time_ms([&] { hipLaunchKernelGGL((k_glds<3>), grid, dim3(256), 0, 0, X, W4, Y2, K, Nout); }, 10);
        CHECK(hipMemcpy(y2.data(), Y2, y2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad3 = 0;
        for (size_t i = 0; i < y1.size(); ++i) bad3 += y1[i] != y2[i];
        printf("K %4d -> %4d: reg %.1f us (%.1f TF) | glds 2 buffers %.1f us (%.1f TF) mismatches %zu | glds 3 buffers %.1f us (%.1f TF) mismatches %zu\n",
               K, Nout, 1e3 * t_reg, gf / t_reg, 1e3 * t_g2, gf / t_g2, bad2, 1e3 * t_g3, gf / t_g3, bad3);
        hipFree(X); hipFree(W); hipFree(W4); hipFree(Y1); hipFree(Y2);
    }
    return 0;
}
