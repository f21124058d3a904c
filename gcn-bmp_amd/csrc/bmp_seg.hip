// Per-molecule segment operators on packed rows: the building blocks of the coarse (atom x molecule-vector)
// co-attention family -- ParallelCoattention, AlternatingCoattention, GlobalCoattention, NeuralCoattention
// (models/coattention/parallel_coattention.py:34-84, alternating_coattention.py:36-86,
// global_coattention.py:27-73, neural_coattention.py:27-71).  All HBM-bound and tiny next to the encoder
// (SURVEY.md 8(a) R8': "negligible"); the dense projections of these modules go through the row GEMM.
// A molecule is the row range [mol_row0[m], mol_row0[m] + mol_nrows[m]); w are the row multiplicities.
#include "bmp_kernels.h"

// out[m, c] = sum_rows w[r] * A[r, c (or 0 if ca == 1)] * Y[r, c]
__global__ __launch_bounds__(256) void k_segpool_fwd(const float* __restrict__ A, int ca, const float* __restrict__ Y, int o,
                                                     const float* __restrict__ w, const int* __restrict__ row0,
                                                     const int* __restrict__ nrows, float* __restrict__ out) {
    const int m = blockIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    for (int c = threadIdx.x; c < o; c += 256) {
        float acc = 0.f;
        for (int r = r0; r < r0 + nr; ++r) acc += w[r] * A[(size_t)r * ca + (ca == 1 ? 0 : c)] * Y[(size_t)r * o + c];
        out[(size_t)m * o + c] = acc;
    }
}

// dY[r, c] = w*A*dout[m, c] ; dA[r, c] = w*Y*dout (ca == o)  or  dA[r] = w * sum_c Y*dout (ca == 1).  dY/dA pre-zeroed.
__global__ __launch_bounds__(256) void k_segpool_bwd(const float* __restrict__ dout, const float* __restrict__ A, int ca,
                                                     const float* __restrict__ Y, int o, const float* __restrict__ w,
                                                     const int* __restrict__ row0, const int* __restrict__ nrows,
                                                     float* __restrict__ dA, float* __restrict__ dY) {
    const int m = blockIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = r0 + wave; r < r0 + nr; r += 4) {
        const float wr = w[r];
        float dot = 0.f;
        for (int c = lane; c < o; c += 64) {
            const float g = dout[(size_t)m * o + c];
            const float y = Y[(size_t)r * o + c];
            const float a = A[(size_t)r * ca + (ca == 1 ? 0 : c)];
            dY[(size_t)r * o + c] = wr * a * g;
            if (ca == 1) dot += y * g;
            else dA[(size_t)r * o + c] = wr * y * g;
        }
        if (ca == 1) {
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) dot += __shfl_xor(dot, s);
            if (lane == 0) dA[r] = wr * dot;
        }
    }
}

// alpha[r] = exp(s[r] - max) / sum_rows w * exp(s - max)   (softmax over the molecule's atoms, multiplicities in
// the denominator: chainer softmax over the padded atom axis, alternating_coattention.py:85)
__global__ __launch_bounds__(64) void k_segsoftmax_fwd(const float* __restrict__ s, const float* __restrict__ w,
                                                       const int* __restrict__ row0, const int* __restrict__ nrows,
                                                       float* __restrict__ alpha) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    float mx = -INFINITY;
    for (int k = lane; k < nr; k += 64) if (w[r0 + k] > 0.f) mx = fmaxf(mx, s[r0 + k]);
#pragma unroll
    for (int t = 32; t >= 1; t >>= 1) mx = fmaxf(mx, __shfl_xor(mx, t));
    float sum = 0.f;
    for (int k = lane; k < nr; k += 64) if (w[r0 + k] > 0.f) sum += w[r0 + k] * bmp_exp(s[r0 + k] - mx);
#pragma unroll
    for (int t = 32; t >= 1; t >>= 1) sum += __shfl_xor(sum, t);
    for (int k = lane; k < nr; k += 64) alpha[r0 + k] = w[r0 + k] > 0.f ? bmp_exp(s[r0 + k] - mx) / sum : 0.f;
}

// ds_k = alpha_k * (dalpha_k - w_k * sum_j alpha_j * dalpha_j)
__global__ __launch_bounds__(64) void k_segsoftmax_bwd(const float* __restrict__ dalpha, const float* __restrict__ alpha,
                                                       const float* __restrict__ w, const int* __restrict__ row0,
                                                       const int* __restrict__ nrows, float* __restrict__ ds) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    float t = 0.f;
    for (int k = lane; k < nr; k += 64) t += alpha[r0 + k] * dalpha[r0 + k];
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) t += __shfl_xor(t, q);
    for (int k = lane; k < nr; k += 64) ds[r0 + k] = alpha[r0 + k] * (dalpha[r0 + k] - w[r0 + k] * t);
}

// out[r, :] = q[row_mol[r], :]   (0 for rows of no molecule)
__global__ __launch_bounds__(256) void k_rowbcast_fwd(const float* __restrict__ q, int c, const int* __restrict__ row_mol, int N,
                                                      float* __restrict__ out) {
    const size_t total = (size_t)N * c;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / c), k = (int)(idx % c);
        const int m = row_mol[r];
        out[idx] = m >= 0 ? q[(size_t)m * c + k] : 0.f;
    }
}

// dq[m, :] = sum over the molecule's rows of d[r, :]   (unweighted: every row received the same vector)
__global__ __launch_bounds__(256) void k_rowbcast_bwd(const float* __restrict__ d, int c, const int* __restrict__ row0,
                                                      const int* __restrict__ nrows, float* __restrict__ dq) {
    const int m = blockIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    for (int k = threadIdx.x; k < c; k += 256) {
        float acc = 0.f;
        for (int r = r0; r < r0 + nr; ++r) acc += d[(size_t)r * c + k];
        dq[(size_t)m * c + k] = acc;
    }
}

// s[r] = x[r, :] . u[row_mol[r], :] + s0[row_mol[r]]    (half a wave per row)
__global__ __launch_bounds__(256) void k_rowdot_fwd(const float* __restrict__ x, int d, const float* __restrict__ u,
                                                    const float* __restrict__ s0, const int* __restrict__ row_mol, int N,
                                                    float* __restrict__ s) {
    const int sub = threadIdx.x & 31;
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= N) return;
    const int m = row_mol[r];
    float acc = 0.f;
    if (m >= 0)
        for (int k = sub; k < d; k += 32) acc += x[(size_t)r * d + k] * u[(size_t)m * d + k];
#pragma unroll
    for (int t = 16; t >= 1; t >>= 1) acc += __shfl_xor(acc, t);
    if (sub == 0) s[r] = m >= 0 ? acc + (s0 ? s0[m] : 0.f) : 0.f;
}

// dx[r, :] = ds[r] * u[m, :]
__global__ __launch_bounds__(256) void k_rowdot_bwd_x(const float* __restrict__ ds, const float* __restrict__ u, int d,
                                                      const int* __restrict__ row_mol, int N, float* __restrict__ dx) {
    const size_t total = (size_t)N * d;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d), k = (int)(idx % d);
        const int m = row_mol[r];
        dx[idx] = m >= 0 ? ds[r] * u[(size_t)m * d + k] : 0.f;
    }
}

// du[m, :] = sum_rows ds[r] * x[r, :] ; ds0[m] = sum_rows ds[r]
__global__ __launch_bounds__(256) void k_rowdot_bwd_u(const float* __restrict__ ds, const float* __restrict__ x, int d,
                                                      const int* __restrict__ row0, const int* __restrict__ nrows,
                                                      float* __restrict__ du, float* __restrict__ ds0) {
    const int m = blockIdx.x;
    const int r0 = row0[m], nr = nrows[m];
    for (int k = threadIdx.x; k < d; k += 256) {
        float acc = 0.f;
        for (int r = r0; r < r0 + nr; ++r) acc += ds[r] * x[(size_t)r * d + k];
        du[(size_t)m * d + k] = acc;
    }
    if (threadIdx.x == 0 && ds0) {
        float acc = 0.f;
        for (int r = r0; r < r0 + nr; ++r) acc += ds[r];
        ds0[m] = acc;
    }
}

// ---- circular correlation of atom rows with their molecule's vector (CircularParallelCoattention,
// parallel_coattention.py:139-187): e[r][k] = sum_t a[r][t] * q[m][(t + k) mod o].  One workgroup per molecule; q is
// held twice over in LDS so that the rotation needs no modulo, rows go through LDS ROWS at a time.  The same kernel
// gives the gradient with respect to the rows: da[r][t] = sum_k de[r][k] * q[m][(t + k) mod o].
constexpr int RC_ROWS = 4;
__global__ __launch_bounds__(256) void k_rowcorr(const float* __restrict__ a, int o, const float* __restrict__ q,
                                                 const int* __restrict__ row0, const int* __restrict__ nrows,
                                                 float* __restrict__ e) {
    extern __shared__ float rc_lds[];
    float* qq = rc_lds;                 // [2 o]
    float* ar = rc_lds + 2 * o;         // [RC_ROWS][o]
    const int m = blockIdx.x, r0 = row0[m], nr = nrows[m];
    for (int k = threadIdx.x; k < 2 * o; k += 256) qq[k] = q[(size_t)m * o + (k >= o ? k - o : k)];
    for (int rb = 0; rb < nr; rb += RC_ROWS) {
        const int nb = min(RC_ROWS, nr - rb);
        __syncthreads();
        for (int i = threadIdx.x; i < nb * o; i += 256) ar[i] = a[(size_t)(r0 + rb) * o + i];
        __syncthreads();
        for (int k = threadIdx.x; k < o; k += 256) {
            float acc[RC_ROWS] = {};
            for (int t = 0; t < o; ++t) {
                const float qv = qq[t + k];
#pragma unroll
                for (int j = 0; j < RC_ROWS; ++j) acc[j] += ar[j * o + t] * qv;     // rows past nb read stale LDS, never stored
            }
            for (int j = 0; j < nb; ++j) e[(size_t)(r0 + rb + j) * o + k] = acc[j];
        }
    }
}

// dq[m][s] = sum_rows sum_t a[r][t] * de[r][(s - t) mod o]
__global__ __launch_bounds__(256) void k_rowcorr_bwd_q(const float* __restrict__ a, const float* __restrict__ de, int o,
                                                       const int* __restrict__ row0, const int* __restrict__ nrows,
                                                       float* __restrict__ dq) {
    extern __shared__ float rc_lds[];
    float* dd = rc_lds;                 // [2 o]: de[r] twice over
    float* ar = rc_lds + 2 * o;         // [o]
    const int m = blockIdx.x, r0 = row0[m], nr = nrows[m];
    float acc[4] = {};                  // o <= 1024
    for (int r = r0; r < r0 + nr; ++r) {
        __syncthreads();
        for (int k = threadIdx.x; k < 2 * o; k += 256) dd[k] = de[(size_t)r * o + (k >= o ? k - o : k)];
        for (int k = threadIdx.x; k < o; k += 256) ar[k] = a[(size_t)r * o + k];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s_ = threadIdx.x + i * 256;
            if (s_ < o)
                for (int t = 0; t < o; ++t) acc[i] += ar[t] * dd[s_ - t + o];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s_ = threadIdx.x + i * 256;
        if (s_ < o) dq[(size_t)m * o + s_] = acc[i];
    }
}

static inline int seg_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int bmp_segpool_fwd(const float* A, int ca, const float* Y, int o, const float* w, const int* mol_row0,
                               const int* mol_nrows, int n_mols, float* out, hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && o > 0 && (ca == 1 || ca == o));
    hipLaunchKernelGGL(k_segpool_fwd, dim3(n_mols), dim3(256), 0, st, A, ca, Y, o, w, mol_row0, mol_nrows, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_segpool_bwd(const float* dout, const float* A, int ca, const float* Y, int o, const float* w,
                               const int* mol_row0, const int* mol_nrows, int n_mols, int N, float* dA, float* dY,
                               hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && o > 0 && N > 0 && (ca == 1 || ca == o));
    hipError_t e = hipMemsetAsync(dA, 0, (size_t)N * ca * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(dY, 0, (size_t)N * o * sizeof(float), st)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_segpool_bwd, dim3(n_mols), dim3(256), 0, st, dout, A, ca, Y, o, w, mol_row0, mol_nrows, dA, dY);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_segsoftmax_fwd(const float* s, const float* w, const int* mol_row0, const int* mol_nrows, int n_mols,
                                  int N, float* alpha, hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && N > 0);
    hipError_t e = hipMemsetAsync(alpha, 0, (size_t)N * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_segsoftmax_fwd, dim3(n_mols), dim3(64), 0, st, s, w, mol_row0, mol_nrows, alpha);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_segsoftmax_bwd(const float* dalpha, const float* alpha, const float* w, const int* mol_row0,
                                  const int* mol_nrows, int n_mols, int N, float* ds, hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && N > 0);
    hipError_t e = hipMemsetAsync(ds, 0, (size_t)N * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_segsoftmax_bwd, dim3(n_mols), dim3(64), 0, st, dalpha, alpha, w, mol_row0, mol_nrows, ds);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_rowbcast_fwd(const float* q, int c, const int* row_mol, int N, float* out, hipStream_t st) {
    BMP_REQUIRE(N > 0 && c > 0);
    hipLaunchKernelGGL(k_rowbcast_fwd, dim3(seg_blocks((size_t)N * c)), dim3(256), 0, st, q, c, row_mol, N, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_rowbcast_bwd(const float* d, int c, const int* mol_row0, const int* mol_nrows, int n_mols, float* dq,
                                hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && c > 0);
    hipLaunchKernelGGL(k_rowbcast_bwd, dim3(n_mols), dim3(256), 0, st, d, c, mol_row0, mol_nrows, dq);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_rowdot_fwd(const float* x, int d, const float* u, const float* s0, const int* row_mol, int N, float* s,
                              hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0);
    hipLaunchKernelGGL(k_rowdot_fwd, dim3((N + 7) / 8), dim3(256), 0, st, x, d, u, s0, row_mol, N, s);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_rowdot_bwd(const float* ds, const float* x, int d, const float* u, const int* row_mol, const int* mol_row0,
                              const int* mol_nrows, int n_mols, int N, float* dx, float* du, float* ds0, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && n_mols > 0);
    hipLaunchKernelGGL(k_rowdot_bwd_x, dim3(seg_blocks((size_t)N * d)), dim3(256), 0, st, ds, u, d, row_mol, N, dx);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rowdot_bwd_u, dim3(n_mols), dim3(256), 0, st, ds, x, d, mol_row0, mol_nrows, du, ds0);
    BMP_LAUNCH_CHECK();
    return 0;
}

// Circular correlation rows x molecule vector.  a, e [N x o] packed rows, q [n_mols x o]; rows outside every molecule are
// not written (callers weight them with 0).
extern "C" int bmp_rowcorr_fwd(const float* a, int o, const float* q, const int* mol_row0, const int* mol_nrows, int n_mols,
                               float* e, hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && o > 0 && o <= 1024);
    hipLaunchKernelGGL(k_rowcorr, dim3(n_mols), dim3(256), (size_t)(2 + RC_ROWS) * o * sizeof(float), st, a, o, q, mol_row0,
                       mol_nrows, e);
    BMP_LAUNCH_CHECK();
    return 0;
}

// de [N x o] -> da [N x o] (rows of molecules only) and dq [n_mols x o] (written).
extern "C" int bmp_rowcorr_bwd(const float* de, const float* a, int o, const float* q, const int* mol_row0,
                               const int* mol_nrows, int n_mols, float* da, float* dq, hipStream_t st) {
    BMP_REQUIRE(n_mols > 0 && o > 0 && o <= 1024);
    hipLaunchKernelGGL(k_rowcorr, dim3(n_mols), dim3(256), (size_t)(2 + RC_ROWS) * o * sizeof(float), st, de, o, q, mol_row0,
                       mol_nrows, da);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rowcorr_bwd_q, dim3(n_mols), dim3(256), (size_t)3 * o * sizeof(float), st, a, de, o, mol_row0,
                       mol_nrows, dq);
    BMP_LAUNCH_CHECK();
    return 0;
}
