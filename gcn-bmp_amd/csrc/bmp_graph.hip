// Index-driven kernels of the path: embedding gather / scatter-add, CSR neighbour gather-sum
// (forward) and its transpose (backward).  All HBM-bound: 16-B loads, half a wave per row.
#include "bmp_kernels.h"

// ---------------------------------------------------------------------------------------------
// embedding  (EmbedAtomID, models/ggnn.py:85,603): out[row, :] = W[ids[row], :]
// ---------------------------------------------------------------------------------------------
// An id outside [0, V) (the host wrappers reject such batches before any launch) reads row 0 instead of memory that
// is not the table's.
__global__ __launch_bounds__(256) void k_embed_fwd(const int* __restrict__ ids, const float* __restrict__ W, int N, int d, int V,
                                                   float* __restrict__ out) {
    const int d4 = d >> 2;
    const size_t total = (size_t)N * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / d4), c4 = (int)(idx % d4);
        int id = ids[row];
        if ((unsigned)id >= (unsigned)V) id = 0;
        *(f32x4*)(out + (size_t)row * d + 4 * c4) = *(const f32x4*)(W + (size_t)id * d + 4 * c4);
    }
}

// dW[id, :] = sum_{rows with ids[row]==id} dout[row, :], in two deterministic passes.  Pass 1: each workgroup
// accumulates its 128-row chunk in an LDS copy of the table (V x d floats) and writes the table rows it touched,
// plus a touched flag per id, to its own slab.  Pass 2: one workgroup per id sums that row over the slabs in a
// fixed order.  (Most rows are carbon or padding: flushing the LDS tables with global float atomics sent ~450
// memory-side atomic requests to each of a handful of 64-byte lines and took 45 of the kernel's 60 us.)
__global__ __launch_bounds__(256) void k_embed_bwd(const int* __restrict__ ids, const float* __restrict__ dout, int N, int d,
                                                   int V, int rows_per_block, float* __restrict__ slab, int* __restrict__ flags) {
    extern __shared__ float tab[];                // V*d floats + V flags
    int* touched = (int*)(tab + (size_t)V * d);
    for (int i = threadIdx.x; i < V * d; i += 256) tab[i] = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) touched[i] = 0;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = (r0 + rows_per_block) < N ? (r0 + rows_per_block) : N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // eight rows of a wave in flight at a time
    for (int base = r0 + wave; base < r1; base += 32) {
        int id8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = base + 4 * u;
            id8[u] = row < r1 ? ids[row] : -1;
            if (id8[u] >= V) id8[u] = -1;          // out-of-range ids contribute nothing (and touch no LDS)
        }
        for (int c = lane; c < d; c += 64) {
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = id8[u] >= 0 ? dout[(size_t)(base + 4 * u) * d + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (id8[u] >= 0) atomicAdd(&tab[(size_t)id8[u] * d + c], v8[u]);
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (id8[u] >= 0) touched[id8[u]] = 1;
        }
    }
    __syncthreads();
    float* my = slab + (size_t)blockIdx.x * V * d;
    for (int id = 0; id < V; ++id) {
        if (touched[id])
            for (int c = threadIdx.x; c < d; c += 256) my[(size_t)id * d + c] = tab[(size_t)id * d + c];
    }
    for (int id = threadIdx.x; id < V; id += 256) flags[(size_t)id * gridDim.x + blockIdx.x] = touched[id];
}

// pass 2: dW[id, c] = sum over the slabs that touched id (fixed order: four interleaved partial sums, then a tree)
__global__ __launch_bounds__(256) void k_embed_bwd_reduce(const float* __restrict__ slab, const int* __restrict__ flags,
                                                          int nslab, int V, int d, float* __restrict__ dW) {
    __shared__ float red[4][64];
    const int id = blockIdx.x;
    const int c = blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float v = 0.f;
    if (c < d)
        for (int s = g; s < nslab; s += 4)
            if (flags[(size_t)id * nslab + s]) v += slab[((size_t)s * V + id) * d + c];
    red[g][threadIdx.x & 63] = v;
    __syncthreads();
    if (g == 0 && c < d) dW[(size_t)id * d + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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
extern "C" int bmp_embed_fwd(const int* ids, const float* W, int N, int d, int V, float* out, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && (d & 3) == 0 && V > 0 && ids && W && out);
    size_t total = (size_t)N * (d >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_embed_fwd, dim3(blocks), dim3(256), 0, st, ids, W, N, d, V, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

#define EMB_ROWS_PER_BLOCK 128
static size_t emb_table_ws_floats(int N, int d, int V) {
    const size_t nb = (size_t)(N + EMB_ROWS_PER_BLOCK - 1) / EMB_ROWS_PER_BLOCK;
    return nb * V * d + nb * V;                    // table slabs | touched flags (int32)
}

extern "C" size_t bmp_embed_bwd_ws_floats(int N, int d, int V) {
    const size_t a = bmp_wgrad_ws_floats(N, (V + 3) & ~3, d), b = emb_table_ws_floats(N, d, V);
    return a > b ? a : b;
}

// dW[id, :] = sum_{rows with ids[row]==id} dout[row, :] = OneHot(ids)^T . dout: the weight-gradient GEMM with the
// one-hot operand generated in its staging loop (nothing but dout is read; MFMA sums in a fixed order, no atomics).
// dW [V x d] is overwritten.  Row counts that are not a multiple of 32 (never the case for packed row tensors)
// take the two-pass LDS-table kernels above.
extern "C" int bmp_embed_bwd(const int* ids, const float* dout, int N, int d, int V, float* dW, float* ws, size_t ws_floats,
                             hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && V > 0 && ids && dout && dW && ws != nullptr && ws_floats >= bmp_embed_bwd_ws_floats(N, d, V));
    if ((N & 31) == 0 && (d & 3) == 0 && ((uintptr_t)dout & 15) == 0) {
        WGArgs g{nullptr, nullptr, 0, 0, dout, d, V, d, N, dW, d, 0, nullptr, 0, ids};
        return bmp_launch_wgrad(g, ws, st);
    }
    const size_t lds_bytes = ((size_t)V * d + V) * sizeof(float);
    BMP_REQUIRE(lds_bytes <= 160 * 1024);
    if (int rc_attr = bmp_lds_attr((const void*)k_embed_bwd, (size_t)(160 * 1024))) return rc_attr;
    const int nb = (N + EMB_ROWS_PER_BLOCK - 1) / EMB_ROWS_PER_BLOCK;
    float* slab = ws;
    int* flags = (int*)(ws + (size_t)nb * V * d);
    hipLaunchKernelGGL(k_embed_bwd, dim3(nb), dim3(256), lds_bytes, st, ids, dout, N, d, V, EMB_ROWS_PER_BLOCK, slab, flags);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_embed_bwd_reduce, dim3(V, (d + 63) / 64), dim3(256), 0, st, slab, flags, nb, V, d, dW);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Rows by bond type (round 4).  A row has a non-zero gathered message / gathered gradient for bond type e only if its
// CSR row holds an entry of that type: on the DDI batches 73 % / 19 % / 2 % / 52 % of the rows for single / double /
// triple / aromatic bonds -- 1.46 of 4 (row, type) pairs.  The weight-gradient GEMMs of a propagation step multiply the
// other 2.54 as zeros (4 of their 11 column tiles are the per-type blocks G_e).  These lists let them walk the rows
// that count: idx[e * N + p] = the p-th row (ascending) with an entry of type e in the given CSR, cnt[e] = how many.
//   pass 1: per 256-row block and type, the number of such rows;  pass 2: every block sums the counts of the blocks in
//   front of it (at most N / 256 of them) and writes its rows at their ranks.  Fixed order, no atomics.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tr_mask(const int* __restrict__ ptr, const int* __restrict__ col, const int* __restrict__ row_mol,
                                       int row, int N) {
    int m = 0;
    if (row < N) {
        for (int e = ptr[row]; e < ptr[row + 1]; ++e) m |= 1 << (col[e] & 3);
        if (row_mol && row_mol[row] >= 0) m |= 16;          // list 4: the rows of a molecule (real atoms and pad rows)
    }
    return m;
}
template <int NE>
__global__ __launch_bounds__(256) void k_type_rows_count(const int* __restrict__ ptr, const int* __restrict__ col,
                                                         const int* __restrict__ row_mol, int N, int* __restrict__ bcnt) {
    __shared__ int wc[4][NE];
    const int row = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = tr_mask(ptr, col, row_mol, row, N);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int c = __popcll(__ballot((m >> e) & 1));
        if (lane == 0) wc[w][e] = c;
    }
    __syncthreads();
    if (threadIdx.x < NE) bcnt[blockIdx.x * NE + threadIdx.x] = wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x];
}
template <int NE>
__global__ __launch_bounds__(256) void k_type_rows_emit(const int* __restrict__ ptr, const int* __restrict__ col,
                                                        const int* __restrict__ row_mol, int N, const int* __restrict__ bcnt,
                                                        int* __restrict__ idx, int* __restrict__ cnt) {
    __shared__ int base[NE], wc[4][NE], red[4][NE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x * 256 + tid;
    // offsets of this block: the counts of the blocks in front of it, summed in a fixed order
    int part[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) part[e] = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256)
#pragma unroll
        for (int e = 0; e < NE; ++e) part[e] += bcnt[b * NE + e];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        int v = part[e];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) red[w][e] = v;
    }
    const int m = tr_mask(ptr, col, row_mol, row, N);
    unsigned long long bal[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        bal[e] = __ballot((m >> e) & 1);
        if (lane == 0) wc[w][e] = __popcll(bal[e]);
    }
    __syncthreads();
    if (tid < NE) base[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        int off = base[e];
        for (int q = 0; q < w; ++q) off += wc[q][e];
        if ((m >> e) & 1) idx[(size_t)e * N + off + __popcll(bal[e] & ((1ull << lane) - 1ull))] = row;
    }
    if (blockIdx.x == gridDim.x - 1 && tid < NE) cnt[tid] = base[tid] + wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
}

// idx [4 x N] int32, cnt [4] int32; ws: bmp_type_rows_ws_ints(N) ints.  ptr / col: the CSR whose gather the lists describe (the
// TRANSPOSED CSR for the backward's gathered gradients G_e).
extern "C" size_t bmp_type_rows_ws_ints(int N) { return (size_t)((N + 255) / 256) * 8; }
extern "C" int bmp_type_rows(const int* csr_ptr, const int* csr_col, int N, int* idx, int* cnt, int* ws, hipStream_t st) {
    BMP_REQUIRE(csr_ptr && N > 0 && idx && cnt && ws);          // (csr_col may be NULL: a batch without a single bond)
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(k_type_rows_count<4>, dim3(nb), dim3(256), 0, st, csr_ptr, csr_col, (const int*)nullptr, N, ws);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_type_rows_emit<4>, dim3(nb), dim3(256), 0, st, csr_ptr, csr_col, (const int*)nullptr, N, ws, idx, cnt);
    BMP_LAUNCH_CHECK();
    return 0;
}
// The same with a FIFTH list: the rows that belong to a molecule (row_mol >= 0: real atoms and pad rows), ascending -- idx
// [5 x N], cnt [5].  For a batch whose tiles sit at a fixed stride (one molecule per tile: three rows of four belong to no
// molecule) the step's weight-gradient launch walks that list instead of all N rows (bmp_ggnn_step_wgrad: live_rows).
extern "C" int bmp_type_rows_live(const int* csr_ptr, const int* csr_col, const int* row_mol, int N, int* idx, int* cnt, int* ws,
                                  hipStream_t st) {
    BMP_REQUIRE(csr_ptr && row_mol && N > 0 && idx && cnt && ws);
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(k_type_rows_count<5>, dim3(nb), dim3(256), 0, st, csr_ptr, csr_col, row_mol, N, ws);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_type_rows_emit<5>, dim3(nb), dim3(256), 0, st, csr_ptr, csr_col, row_mol, N, ws, idx, cnt);
    BMP_LAUNCH_CHECK();
    return 0;
}
