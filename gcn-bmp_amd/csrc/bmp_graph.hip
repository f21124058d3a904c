// Index-driven kernels of the path: embedding gather / scatter-add, CSR neighbour gather-sum
// (forward) and its transpose (backward).  All HBM-bound: 16-B loads, half a wave per row.
#include "bmp_kernels.h"

// ---------------------------------------------------------------------------------------------
// embedding  (EmbedAtomID, models/ggnn.py:85,603): out[row, :] = W[ids[row], :]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_fwd(const int* __restrict__ ids, const float* __restrict__ W, int N, int d,
                                                   float* __restrict__ out) {
    const int d4 = d >> 2;
    const size_t total = (size_t)N * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / d4), c4 = (int)(idx % d4);
        const int id = ids[row];
        *(f32x4*)(out + (size_t)row * d + 4 * c4) = *(const f32x4*)(W + (size_t)id * d + 4 * c4);
    }
}

// dW[id, :] += sum_{rows with ids[row]==id} dout[row, :].  Each workgroup accumulates its row chunk
// in an LDS copy of the table (V x d floats), then flushes the rows it touched with one global
// float atomic per element (most rows are carbon / pad, so per-row global atomics would pile
// onto two table rows).
__global__ __launch_bounds__(256) void k_embed_bwd(const int* __restrict__ ids, const float* __restrict__ dout, int N, int d,
                                                   int V, int rows_per_block, float* dW) {
    extern __shared__ float tab[];                // V*d floats + V flags
    int* touched = (int*)(tab + (size_t)V * d);
    for (int i = threadIdx.x; i < V * d; i += 256) tab[i] = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) touched[i] = 0;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = (r0 + rows_per_block) < N ? (r0 + rows_per_block) : N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = r0 + wave; row < r1; row += 4) {
        const int id = ids[row];
        if (lane == 0) touched[id] = 1;
        for (int c = lane; c < d; c += 64) atomicAdd(&tab[(size_t)id * d + c], dout[(size_t)row * d + c]);
    }
    __syncthreads();
    for (int id = 0; id < V; ++id) {
        if (touched[id])
            for (int c = threadIdx.x; c < d; c += 256) atomicAdd(&dW[(size_t)id * d + c], tab[(size_t)id * d + c]);
    }
}

// ---------------------------------------------------------------------------------------------
// neighbour gather-sum.  Half a wave (32 lanes, float4 each) per destination row.
//   FWD: agg[i, e*d + k] = sum val * x[src, k]          (4 accumulators, one per bond type)
//   BWD: dx[j, k] (+)= sum val * dagg[dst, e*d + k]      (transposed CSR)
// models/ggnn.py:229-242 does this as a dense (mb*4, A, A) x (mb*4, A, d) batched matmul.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_fwd(const float* __restrict__ x, int ldx, int N, int d,
                                                    const int* __restrict__ ptr, const int* __restrict__ col,
                                                    const float* __restrict__ val, float* __restrict__ agg,
                                                    float* __restrict__ wdeg) {
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= N) return;
    const int e0 = ptr[row], e1 = ptr[row + 1];
    const int d4 = d >> 2;
    float wd[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = sub; c4 < d4; c4 += 32) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = e0; e < e1; ++e) {
            const int cv = col[e];
            const float v = val[e];
            const int src = cv >> 2, typ = cv & 3;
            if (c4 == sub) {                       // first pass: weighted degree per bond type
#pragma unroll
                for (int t = 0; t < 4; ++t) wd[t] += (typ == t) ? v : 0.f;
            }
            const f32x4 xv = *(const f32x4*)(x + (size_t)src * ldx + 4 * c4);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += xv * ((typ == t) ? v : 0.f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) *(f32x4*)(agg + (size_t)row * 4 * d + (size_t)t * d + 4 * c4) = acc[t];
    }
    if (sub == 0) *(f32x4*)(wdeg + (size_t)row * 4) = (f32x4){wd[0], wd[1], wd[2], wd[3]};
}

__global__ __launch_bounds__(256) void k_gather_bwd(const float* __restrict__ dagg, int N, int d,
                                                    const int* __restrict__ ptrT, const int* __restrict__ colT,
                                                    const float* __restrict__ valT, float* dx, int lddx, int accumulate) {
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= N) return;
    const int e0 = ptrT[row], e1 = ptrT[row + 1];
    const int d4 = d >> 2;
    for (int c4 = sub; c4 < d4; c4 += 32) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = e0; e < e1; ++e) {
            const int cv = colT[e];
            const int dst = cv >> 2, typ = cv & 3;
            acc += *(const f32x4*)(dagg + (size_t)dst * 4 * d + (size_t)typ * d + 4 * c4) * valT[e];
        }
        f32x4* o = (f32x4*)(dx + (size_t)row * lddx + 4 * c4);
        *o = accumulate ? (*o + acc) : acc;
    }
}

int bmp_launch_gather_fwd(const float* x, int ldx, int N, int d, const int* ptr, const int* col, const float* val,
                          float* agg, float* wdeg, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && (d & 3) == 0 && (ldx & 3) == 0);
    hipLaunchKernelGGL(k_gather_fwd, dim3((N + 7) / 8), dim3(256), 0, st, x, ldx, N, d, ptr, col, val, agg, wdeg);
    BMP_LAUNCH_CHECK();
    return 0;
}

int bmp_launch_gather_bwd(const float* dagg, int N, int d, const int* ptrT, const int* colT, const float* valT,
                          float* dx, int lddx, int accumulate, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && (d & 3) == 0 && (lddx & 3) == 0);
    hipLaunchKernelGGL(k_gather_bwd, dim3((N + 7) / 8), dim3(256), 0, st, dagg, N, d, ptrT, colT, valT, dx, lddx, accumulate);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// C-ABI: embedding
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_embed_fwd(const int* ids, const float* W, int N, int d, float* out, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && (d & 3) == 0);
    size_t total = (size_t)N * (d >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_embed_fwd, dim3(blocks), dim3(256), 0, st, ids, W, N, d, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_embed_bwd(const int* ids, const float* dout, int N, int d, int V, float* dW, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && V > 0);
    const size_t lds_bytes = ((size_t)V * d + V) * sizeof(float);
    BMP_REQUIRE(lds_bytes <= 160 * 1024);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_embed_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int rows_per_block = 128;
    hipLaunchKernelGGL(k_embed_bwd, dim3((N + rows_per_block - 1) / rows_per_block), dim3(256), lds_bytes, st, ids, dout, N,
                       d, V, rows_per_block, dW);
    BMP_LAUNCH_CHECK();
    return 0;
}
