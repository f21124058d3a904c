// Pair features of the reference's link predictors other than the plain MLP (models/mlp.py:48-193, used at
// train_binary.py:102-116): what is computed from (g1, g2) before the relu-MLP tail.
//   SYM      [g1 + g2 | g1 * g2]                                   SymMLP.__call__            models/mlp.py:104-110
//   HOLE     circular correlation  c[k] = sum_i g1[i] g2[(i+k) % d]  HolE.circular_correlation  models/mlp.py:126-151
//            (the reference takes it through fft/ifft; d is 16..256 here, the direct sum is exact and cheaper)
//   DISTMULT y[o] = sum_p W[o,p] g1[p] g2[p]                        BilinearDiag               models/mlp.py:153-193
//   NTN      y[o] = g1^T W[:,:,o] g2 + g1.V1[:,o] + g2.V2[:,o] + b[o]   links.Bilinear        models/mlp.py:52,66
// B ~ 1000 rows: plain fp32 FMA kernels, fixed summation orders (no atomics).
#include "bmp_common.h"

enum { PF_SYM = 0, PF_HOLE = 1, PF_DISTMULT = 2, PF_NTN = 3 };
#define PF_MAXD 1024
#define PF_MAXK 16
#define PF_RB 16            // rows per workgroup of the bilinear kernels

// ---------------------------------------------------------------------------------------------- SYM
__global__ void k_pf_sym_fwd(const float* __restrict__ x1, const float* __restrict__ x2, int B, int d, float* __restrict__ out) {
    const size_t n = (size_t)B * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / d, c = i % d;
        const float a = x1[i], b = x2[i];
        out[r * 2 * d + c] = a + b;
        out[r * 2 * d + d + c] = a * b;
    }
}
__global__ void k_pf_sym_bwd(const float* __restrict__ g, const float* __restrict__ x1, const float* __restrict__ x2, int B,
                             int d, float* __restrict__ dx1, float* __restrict__ dx2) {
    const size_t n = (size_t)B * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / d, c = i % d;
        const float gs = g[r * 2 * d + c], gp = g[r * 2 * d + d + c];
        dx1[i] = gs + gp * x2[i];
        dx2[i] = gs + gp * x1[i];
    }
}

// ---------------------------------------------------------------------------------------------- HOLE
// one workgroup per row; the three vectors of the row live in LDS
__global__ __launch_bounds__(256) void k_pf_hole_fwd(const float* __restrict__ x1, const float* __restrict__ x2, int d,
                                                     float* __restrict__ out) {
    __shared__ float a[PF_MAXD], b[PF_MAXD];
    const size_t row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 256) { a[i] = x1[row * d + i]; b[i] = x2[row * d + i]; }
    __syncthreads();
    for (int k = threadIdx.x; k < d; k += 256) {
        float acc = 0.f;
        int j = k;
        for (int i = 0; i < d; ++i) { acc += a[i] * b[j]; if (++j == d) j = 0; }
        out[row * d + k] = acc;
    }
}
// dx1[i] = sum_k g[k] x2[(i+k)%d] ; dx2[j] = sum_k g[k] x1[(j-k)%d]
__global__ __launch_bounds__(256) void k_pf_hole_bwd(const float* __restrict__ g, const float* __restrict__ x1,
                                                     const float* __restrict__ x2, int d, float* __restrict__ dx1,
                                                     float* __restrict__ dx2) {
    __shared__ float a[PF_MAXD], b[PF_MAXD], gg[PF_MAXD];
    const size_t row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 256) { a[i] = x1[row * d + i]; b[i] = x2[row * d + i]; gg[i] = g[row * d + i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256) {
        float s1 = 0.f, s2 = 0.f;
        int jp = i, jm = i;
        for (int k = 0; k < d; ++k) {
            s1 += gg[k] * b[jp];
            s2 += gg[k] * a[jm];
            if (++jp == d) jp = 0;
            if (--jm < 0) jm = d - 1;
        }
        dx1[row * d + i] = s1;
        dx2[row * d + i] = s2;
    }
}

// ---------------------------------------------------------------------------------------------- DISTMULT
__global__ __launch_bounds__(256) void k_pf_dm_fwd(const float* __restrict__ x1, const float* __restrict__ x2, int B, int d,
                                                   const float* __restrict__ W, int K, float* __restrict__ out) {
    __shared__ float pr[PF_RB][PF_MAXD];
    const int row0 = blockIdx.x * PF_RB;
    for (int idx = threadIdx.x; idx < PF_RB * d; idx += 256) {
        const int r = idx / d, c = idx % d, row = row0 + r;
        pr[r][c] = row < B ? x1[(size_t)row * d + c] * x2[(size_t)row * d + c] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PF_RB * K; idx += 256) {
        const int r = idx / K, o = idx % K;
        if (row0 + r >= B) continue;
        float acc = 0.f;
        for (int p = 0; p < d; ++p) acc += W[(size_t)o * d + p] * pr[r][p];
        out[(size_t)(row0 + r) * K + o] = acc;
    }
}
__global__ __launch_bounds__(256) void k_pf_dm_bwd_x(const float* __restrict__ g, const float* __restrict__ x1,
                                                     const float* __restrict__ x2, int B, int d, const float* __restrict__ W,
                                                     int K, float* __restrict__ dx1, float* __restrict__ dx2) {
    const size_t n = (size_t)B * d;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / d, p = i % d;
        float dp = 0.f;
        for (int o = 0; o < K; ++o) dp += g[r * K + o] * W[(size_t)o * d + p];
        dx1[i] = dp * x2[i];
        dx2[i] = dp * x1[i];
    }
}
// dW[o, p] = sum_b g[b, o] x1[b, p] x2[b, p]: one thread per (o, p), rows in order
__global__ __launch_bounds__(256) void k_pf_dm_bwd_w(const float* __restrict__ g, const float* __restrict__ x1,
                                                     const float* __restrict__ x2, int B, int d, int K, float* __restrict__ dW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= K * d) return;
    const int o = idx / d, p = idx % d;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += g[(size_t)b * K + o] * x1[(size_t)b * d + p] * x2[(size_t)b * d + p];
    dW[idx] = acc;
}

// ---------------------------------------------------------------------------------------------- NTN
// W [d1 x d2 x K] (chainer Bilinear layout, o fastest), V1 [d1 x K], V2 [d2 x K], b [K]
__global__ __launch_bounds__(256) void k_pf_ntn_fwd(const float* __restrict__ x1, const float* __restrict__ x2, int B, int d1,
                                                    int d2, const float* __restrict__ W, const float* __restrict__ V1,
                                                    const float* __restrict__ V2, const float* __restrict__ bias, int K,
                                                    float* __restrict__ out) {
    __shared__ float a[PF_RB][PF_MAXD / 4], c[PF_RB][PF_MAXD / 4];
    const int row0 = blockIdx.x * PF_RB;
    for (int idx = threadIdx.x; idx < PF_RB * d1; idx += 256) {
        const int r = idx / d1, k = idx % d1;
        a[r][k] = row0 + r < B ? x1[(size_t)(row0 + r) * d1 + k] : 0.f;
    }
    for (int idx = threadIdx.x; idx < PF_RB * d2; idx += 256) {
        const int r = idx / d2, k = idx % d2;
        c[r][k] = row0 + r < B ? x2[(size_t)(row0 + r) * d2 + k] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PF_RB * K; idx += 256) {
        const int r = idx / K, o = idx % K;
        if (row0 + r >= B) continue;
        float y = bias ? bias[o] : 0.f;
        for (int p = 0; p < d1; ++p) {
            float t = V1 ? V1[(size_t)p * K + o] : 0.f;
            const float* w = W + (size_t)p * d2 * K + o;
            for (int q = 0; q < d2; ++q) t += w[(size_t)q * K] * c[r][q];
            y += a[r][p] * t;
        }
        if (V2)
            for (int q = 0; q < d2; ++q) y += c[r][q] * V2[(size_t)q * K + o];
        out[(size_t)(row0 + r) * K + o] = y;
    }
}
// dx1[b,p] = sum_o g[b,o] (sum_q W[p,q,o] x2[b,q] + V1[p,o]) ; dx2[b,q] = sum_o g[b,o] (sum_p W[p,q,o] x1[b,p] + V2[q,o])
__global__ __launch_bounds__(256) void k_pf_ntn_bwd_x(const float* __restrict__ g, const float* __restrict__ x1,
                                                      const float* __restrict__ x2, int B, int d1, int d2,
                                                      const float* __restrict__ W, const float* __restrict__ V1,
                                                      const float* __restrict__ V2, int K, float* __restrict__ dx1,
                                                      float* __restrict__ dx2) {
    __shared__ float a[PF_RB][PF_MAXD / 4], c[PF_RB][PF_MAXD / 4], gg[PF_RB][PF_MAXK];
    const int row0 = blockIdx.x * PF_RB;
    for (int idx = threadIdx.x; idx < PF_RB * d1; idx += 256) {
        const int r = idx / d1, k = idx % d1;
        a[r][k] = row0 + r < B ? x1[(size_t)(row0 + r) * d1 + k] : 0.f;
    }
    for (int idx = threadIdx.x; idx < PF_RB * d2; idx += 256) {
        const int r = idx / d2, k = idx % d2;
        c[r][k] = row0 + r < B ? x2[(size_t)(row0 + r) * d2 + k] : 0.f;
    }
    for (int idx = threadIdx.x; idx < PF_RB * K; idx += 256) {
        const int r = idx / K, o = idx % K;
        gg[r][o] = row0 + r < B ? g[(size_t)(row0 + r) * K + o] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PF_RB * d1; idx += 256) {      // lanes walk p: W[p, q, :] rows of K floats
        const int r = idx / d1, p = idx % d1;
        if (row0 + r >= B) continue;
        float acc = 0.f;
        for (int o = 0; o < K; ++o) acc += gg[r][o] * (V1 ? V1[(size_t)p * K + o] : 0.f);
        const float* w = W + (size_t)p * d2 * K;
        for (int q = 0; q < d2; ++q) {
            float t = 0.f;
            for (int o = 0; o < K; ++o) t += w[(size_t)q * K + o] * gg[r][o];
            acc += t * c[r][q];
        }
        dx1[(size_t)(row0 + r) * d1 + p] = acc;
    }
    for (int idx = threadIdx.x; idx < PF_RB * d2; idx += 256) {      // lanes walk q: adjacent K-float groups
        const int r = idx / d2, q = idx % d2;
        if (row0 + r >= B) continue;
        float acc = 0.f;
        for (int o = 0; o < K; ++o) acc += gg[r][o] * (V2 ? V2[(size_t)q * K + o] : 0.f);
        for (int p = 0; p < d1; ++p) {
            const float* w = W + ((size_t)p * d2 + q) * K;
            float t = 0.f;
            for (int o = 0; o < K; ++o) t += w[o] * gg[r][o];
            acc += t * a[r][p];
        }
        dx2[(size_t)(row0 + r) * d2 + q] = acc;
    }
}
// workgroup p: dW[p, q, o] = sum_b x1[b,p] x2[b,q] g[b,o]; dV1[p, o] = sum_b x1[b,p] g[b,o];
// workgroups p < d2 also dV2[p, o] = sum_b x2[b,p] g[b,o]; workgroup 0 also db[o] = sum_b g[b,o].  Rows in order.
// The rows come through LDS in chunks of PF_WR (x1's column p, the chunk's x2 and g rows): round 3's form walked the B rows
// from global memory inside every thread's sum -- three dependent strided loads per multiply-add, 437 us for 1024 pairs of
// the reference's published model (d1 = d2 = 16, K = 8).  Same sums in the same order.
#define PF_WR 32
__global__ __launch_bounds__(256) void k_pf_ntn_bwd_w(const float* __restrict__ g, const float* __restrict__ x1,
                                                      const float* __restrict__ x2, int B, int d1, int d2, int K,
                                                      float* __restrict__ dW, float* __restrict__ dV1,
                                                      float* __restrict__ dV2, float* __restrict__ db) {
    extern __shared__ float pf_lds[];
    float* sx1 = pf_lds;                       // [PF_WR]           x1[b, p]
    float* sx2 = sx1 + PF_WR;                  // [PF_WR x d2]
    float* sg = sx2 + PF_WR * d2;              // [PF_WR x K]
    const int p = blockIdx.x, tid = threadIdx.x;
    const int nW = p < d1 ? d2 * K : 0;        // outputs of dW owned by this workgroup
    constexpr int MAXO = (PF_MAXD / 4) * PF_MAXK / 256;      // dW outputs per thread
    float accW[MAXO];
#pragma unroll
    for (int t = 0; t < MAXO; ++t) accW[t] = 0.f;
    float accV1 = 0.f, accV2 = 0.f, accB = 0.f;
    const bool small = nW > 0 && nW <= 128;          // (uniform per workgroup)
    for (int b0 = 0; b0 < B; b0 += PF_WR) {
        const int nr = B - b0 < PF_WR ? B - b0 : PF_WR;
        __syncthreads();
        if (p < d1) for (int i = tid; i < nr; i += 256) sx1[i] = x1[(size_t)(b0 + i) * d1 + p];
        for (int i = tid; i < nr * d2; i += 256) sx2[i] = x2[(size_t)b0 * d2 + i];
        for (int i = tid; i < nr * K; i += 256) sg[i] = g[(size_t)b0 * K + i];
        __syncthreads();
        if (small) {          // at most 128 outputs: the two halves of the workgroup take the chunk's even / odd rows
            const int idx = tid & 127, g = tid >> 7;
            if (idx < nW) {
                const int q = idx / K, o = idx % K;
                float acc = accW[0];
                for (int b = g; b < nr; b += 2) acc += sx1[b] * sx2[b * d2 + q] * sg[b * K + o];
                accW[0] = acc;
            }
        } else {
#pragma unroll
        for (int t = 0; t < MAXO; ++t) {
            const int idx = tid + 256 * t;
            if (idx < nW) {
                const int q = idx / K, o = idx % K;
                float acc = accW[t];
                for (int b = 0; b < nr; ++b) acc += sx1[b] * sx2[b * d2 + q] * sg[b * K + o];
                accW[t] = acc;
            }
        }
        }
        if (tid < K) {
            const int o = tid;
            if (dV1 && p < d1) for (int b = 0; b < nr; ++b) accV1 += sx1[b] * sg[b * K + o];
            if (dV2 && p < d2) for (int b = 0; b < nr; ++b) accV2 += sx2[b * d2 + p] * sg[b * K + o];
            if (db && p == 0) for (int b = 0; b < nr; ++b) accB += sg[b * K + o];
        }
    }
    if (small) {              // even-row sums + odd-row sums, through LDS (the staging area is free by now)
        __syncthreads();
        if (tid >= 128) pf_lds[tid - 128] = accW[0];
        __syncthreads();
        if (tid < nW) dW[(size_t)p * d2 * K + tid] = accW[0] + pf_lds[tid];
    } else {
#pragma unroll
    for (int t = 0; t < MAXO; ++t) {
        const int idx = tid + 256 * t;
        if (idx < nW) dW[(size_t)p * d2 * K + idx] = accW[t];
    }
    }
    if (tid < K) {
        if (dV1 && p < d1) dV1[(size_t)p * K + tid] = accV1;
        if (dV2 && p < d2) dV2[(size_t)p * K + tid] = accV2;
        if (db && p == 0) db[tid] = accB;
    }
}

// ---------------------------------------------------------------------------------------------- C ABI
static int pf_blocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : b); }

extern "C" int bmp_pairfeat_cols(int kind, int d, int K) {
    return kind == PF_SYM ? 2 * d : kind == PF_HOLE ? d : K;
}

// out [B x bmp_pairfeat_cols].  x1 [B x d1], x2 [B x d2] (d1 == d2 except for NTN).  W: DISTMULT [K x d]; NTN [d1 x d2 x K]
// with optional V1 [d1 x K], V2 [d2 x K], b [K].
extern "C" int bmp_pairfeat_fwd(int kind, const float* x1, const float* x2, int B, int d1, int d2, const float* W,
                                const float* V1, const float* V2, const float* b, int K, float* out, hipStream_t st) {
    BMP_REQUIRE(B > 0 && d1 > 0 && d2 > 0 && x1 && x2 && out);
    switch (kind) {
        case PF_SYM:
            BMP_REQUIRE(d1 == d2);
            hipLaunchKernelGGL(k_pf_sym_fwd, dim3(pf_blocks((size_t)B * d1)), dim3(256), 0, st, x1, x2, B, d1, out);
            break;
        case PF_HOLE:
            BMP_REQUIRE(d1 == d2 && d1 <= PF_MAXD);
            hipLaunchKernelGGL(k_pf_hole_fwd, dim3(B), dim3(256), 0, st, x1, x2, d1, out);
            break;
        case PF_DISTMULT:
            BMP_REQUIRE(d1 == d2 && d1 <= PF_MAXD && W && K > 0);
            hipLaunchKernelGGL(k_pf_dm_fwd, dim3((B + PF_RB - 1) / PF_RB), dim3(256), 0, st, x1, x2, B, d1, W, K, out);
            break;
        case PF_NTN:
            BMP_REQUIRE(d1 <= PF_MAXD / 4 && d2 <= PF_MAXD / 4 && W && K > 0 && K <= PF_MAXK);
            hipLaunchKernelGGL(k_pf_ntn_fwd, dim3((B + PF_RB - 1) / PF_RB), dim3(256), 0, st, x1, x2, B, d1, d2, W, V1, V2, b, K, out);
            break;
        default: return -1;
    }
    BMP_LAUNCH_CHECK();
    return 0;
}

// Backward from dout [B x cols]: dx1, dx2 and (DISTMULT) dW [K x d]; (NTN) dW [d1 x d2 x K], optional dV1, dV2, db.
extern "C" int bmp_pairfeat_bwd(int kind, const float* dout, const float* x1, const float* x2, int B, int d1, int d2,
                                const float* W, const float* V1, const float* V2, int K, float* dx1, float* dx2, float* dW,
                                float* dV1, float* dV2, float* db, hipStream_t st) {
    BMP_REQUIRE(B > 0 && d1 > 0 && d2 > 0 && dout && x1 && x2 && dx1 && dx2);
    switch (kind) {
        case PF_SYM:
            BMP_REQUIRE(d1 == d2);
            hipLaunchKernelGGL(k_pf_sym_bwd, dim3(pf_blocks((size_t)B * d1)), dim3(256), 0, st, dout, x1, x2, B, d1, dx1, dx2);
            break;
        case PF_HOLE:
            BMP_REQUIRE(d1 == d2 && d1 <= PF_MAXD);
            hipLaunchKernelGGL(k_pf_hole_bwd, dim3(B), dim3(256), 0, st, dout, x1, x2, d1, dx1, dx2);
            break;
        case PF_DISTMULT:
            BMP_REQUIRE(d1 == d2 && W && dW && K > 0);
            hipLaunchKernelGGL(k_pf_dm_bwd_x, dim3(pf_blocks((size_t)B * d1)), dim3(256), 0, st, dout, x1, x2, B, d1, W, K, dx1, dx2);
            BMP_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_pf_dm_bwd_w, dim3((K * d1 + 255) / 256), dim3(256), 0, st, dout, x1, x2, B, d1, K, dW);
            break;
        case PF_NTN: {
            BMP_REQUIRE(d1 <= PF_MAXD / 4 && d2 <= PF_MAXD / 4 && W && dW && K > 0 && K <= PF_MAXK);
            hipLaunchKernelGGL(k_pf_ntn_bwd_x, dim3((B + PF_RB - 1) / PF_RB), dim3(256), 0, st, dout, x1, x2, B, d1, d2, W, V1, V2, K, dx1, dx2);
            BMP_LAUNCH_CHECK();
            const int nb = d1 > d2 ? d1 : d2;
            hipLaunchKernelGGL(k_pf_ntn_bwd_w, dim3(nb), dim3(256), (size_t)(PF_WR * (1 + d2 + K) > 128 ? PF_WR * (1 + d2 + K) : 128) * sizeof(float), st, dout, x1, x2, B, d1, d2,
                               K, dW, dV1, dV2, db);
            break;
        }
        default: return -1;
    }
    BMP_LAUNCH_CHECK();
    return 0;
}
