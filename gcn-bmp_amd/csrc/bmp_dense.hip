// The reference's batch contract on the device: its encoders take a dense zero-padded adjacency
// adj (mb, 4, A, A) float32 (train_ddi_modify.py:296, concat_mols) that is ~99 % zeros -- 65 KB per molecule at A = 64.
// These kernels turn it into the packed CSR of bmp/packed.py without a trip through the host: one pass counts the
// bonds per dense position (the host needs those counts, and the atom ids, to place molecules into tiles), one pass
// per direction writes the entries.  Integer / byte work, HBM bound: every adjacency element is read once per pass.
// Entry order inside a row equals the host packer's (source position ascending, then bond type), so the result is
// bit-identical to pack_from_dense (tests/test_gpu_dense.py).
#include "bmp_common.h"

// row_nnz[b, i] = #{(e, j): adj[b, e, i, j] != 0}, col_nnz[b, j] = #{(e, i): adj[b, e, i, j] != 0}
// One workgroup per (molecule, bond-type plane pair handled by the loop): thread t owns column j = t (+256 k) for
// the column counts and rows are reduced across the workgroup for the row counts.
__global__ __launch_bounds__(256) void k_dense_count(const float* __restrict__ adj, int A, int* __restrict__ row_nnz,
                                                     int* __restrict__ col_nnz) {
    extern __shared__ int cnt[];               // [A] row counts
    const int b = blockIdx.x;
    const float* base = adj + (size_t)b * 4 * A * A;
    for (int i = threadIdx.x; i < A; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < A; j += 256) {        // coalesced along j
        int cj = 0;
        for (int e = 0; e < 4; ++e)
            for (int i = 0; i < A; ++i) {
                const bool nz = base[((size_t)e * A + i) * A + j] != 0.f;
                cj += nz;
                if (nz) atomicAdd(&cnt[i], 1);           // LDS integer add: order-free, exact
            }
        col_nnz[(size_t)b * A + j] = cj;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A; i += 256) row_nnz[(size_t)b * A + i] = cnt[i];
}

// One thread per dense position p = (b, i): writes the entries of packed row rowmap[p] starting at ptr[rowmap[p]].
//   transposed == 0: entries (src j, type e) of destination i, col = rowmap[b, j] << 2 | e, j ascending then e;
//   transposed == 1: entries (dst i', type e) of source i,     col = rowmap[b, i'] << 2 | e.
// Positions without entries write nothing (several padded positions share one virtual pad row; the host only
// uses this path when those have no entries).
__global__ __launch_bounds__(256) void k_dense_fill(const float* __restrict__ adj, int mb, int A, const int* __restrict__ rowmap,
                                                    const int* __restrict__ ptr, int transposed, int* __restrict__ col,
                                                    float* __restrict__ val) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= mb * A) return;
    const int b = p / A, i = p % A;
    const float* base = adj + (size_t)b * 4 * A * A;
    const int* rm = rowmap + (size_t)b * A;
    int w = ptr[rm[i]];
    for (int j = 0; j < A; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = transposed ? base[((size_t)e * A + j) * A + i] : base[((size_t)e * A + i) * A + j];
            if (v != 0.f) {
                col[w] = (rm[j] << 2) | e;
                val[w] = v;
                ++w;
            }
        }
    }
}

extern "C" int bmp_dense_count(const float* adj, int mb, int A, int* row_nnz, int* col_nnz, hipStream_t st) {
    BMP_REQUIRE(adj && row_nnz && col_nnz && mb > 0 && A > 0 && A <= 16384);
    hipLaunchKernelGGL(k_dense_count, dim3(mb), dim3(256), (size_t)A * sizeof(int), st, adj, A, row_nnz, col_nnz);
    BMP_LAUNCH_CHECK();
    return 0;
}

// adj (mb, 4, A, A) -> the entries of the packed CSR (transposed = 0: by destination row; 1: by source row).
// rowmap [mb x A]: packed row of every dense position; ptr [N + 1]: row pointers (the host builds both from the
// counts of bmp_dense_count).  col / val must hold ptr[N] entries.
extern "C" int bmp_dense_to_csr(const float* adj, int mb, int A, const int* rowmap, const int* ptr, int transposed,
                                int* col, float* val, hipStream_t st) {
    BMP_REQUIRE(adj && rowmap && ptr && col && val && mb > 0 && A > 0);
    const int n = mb * A;
    hipLaunchKernelGGL(k_dense_fill, dim3((n + 255) / 256), dim3(256), 0, st, adj, mb, A, rowmap, ptr, transposed, col, val);
    BMP_LAUNCH_CHECK();
    return 0;
}
