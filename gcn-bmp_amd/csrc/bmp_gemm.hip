// fp32-MFMA GEMM building blocks for gfx950: the row GEMM (atom-row tiles x weights) and the
// weight-gradient GEMM (reduction over atom rows).  See bmp_kernels.h for the contracts.
#include "bmp_kernels.h"

// ---------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void rg_epilogue(const RGArgs& a, int row, int col, float v) {
    if (a.bias) v += a.bias[col];
    if (EPI == BMP_EPI_GENERIC) {
        if (a.wdeg) {
            const float* wd = a.wdeg + (size_t)row * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v += wd[e] * a.bE[(size_t)e * a.ldbE + col];
        }
        const bool lo = (a.split <= 0) || (col < a.split);
        if (a.add && lo) v += a.add[(size_t)row * a.ldadd + col];
        v = bmp_act(lo ? a.act_lo : a.act_hi, v);
        if (!lo && a.o1) {
            a.o1[(size_t)row * a.ldo1 + (col - a.split)] = v;
        } else {
            float* y = a.Y + (size_t)row * a.ldy + col;
            *y = a.accumulate ? (*y + v) : v;
        }
    } else if (EPI == BMP_EPI_GRU_OUT) {
        // models/ggnn.py:260 -> chainer StatefulGRU: h' = z*h_bar + (1-z)*h ; first call: z*h_bar
        const float c = bmp_tanh(v);
        const float z = a.z[(size_t)row * a.ldz + col];
        a.c_out[(size_t)row * a.ldc + col] = c;
        float hn = z * c;
        if (!a.first) hn += (1.f - z) * a.h[(size_t)row * a.ldh + col];
        a.Y[(size_t)row * a.ldy + col] = hn;
    } else {  // BMP_EPI_GRU_DRH: v = d(r*h)
        const float r = a.r[(size_t)row * a.ldr + col];
        const float h = a.h[(size_t)row * a.ldh + col];
        a.Y[(size_t)row * a.ldy + col] = v * h * r * (1.f - r);
        float* o = a.o1 + (size_t)row * a.ldo1 + col;
        *o += v * r;
    }
}

// ---------------------------------------------------------------------------------------------
// row GEMM kernel.  256 threads = 4 waves laid out WR (rows) x WC (cols); every wave owns
// RB x CBW blocks of 32x32.  R = WR*RB*32 = 128 rows, NT = WC*CBW*32 columns per workgroup.
// ---------------------------------------------------------------------------------------------
template <int WR, int RB, int CBW, int EPI>
__global__ __launch_bounds__(256) void k_rowgemm(RGArgs a) {
    constexpr int WC = 4 / WR;
    constexpr int NT = WC * CBW * 32;
    static_assert(WR * RB * 32 == BMP_R, "tile must be 128 rows");
    __shared__ __attribute__((aligned(16))) float lds[BMP_R * BMP_LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wr = w / WC, wc = w % WC;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * BMP_R;
    const int n0 = blockIdx.y * NT;

    f32x16 acc[RB][CBW];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

    int colc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
        int col = n0 + (wc * CBW + cb) * 32 + l31;
        colc[cb] = col < a.Nout ? col : a.Nout - 1;     // clamp loads, mask stores
    }

    for (int s = 0; s < a.nsrc; ++s) {
        const float* __restrict__ X = a.s[s].X;
        const float* __restrict__ X2 = a.s[s].X2;
        const float* __restrict__ Wt = a.s[s].Wt;
        const int ldx = a.s[s].ldx, ldx2 = a.s[s].ldx2, ldw = a.s[s].ldw, K = a.s[s].K;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kc = (K - k0) < 64 ? (K - k0) : 64;
            const int kc4 = kc >> 2;
            __syncthreads();
            // stage rows [row0, row0+128) x k [k0, k0+kc) : 16 float4 slots per row
#pragma unroll
            for (int it = 0; it < (BMP_R * 16) / 256; ++it) {
                const int idx = tid + it * 256;
                const int r = idx >> 4, c4 = idx & 15;
                if (c4 < kc4) {
                    f32x4 v = *(const f32x4*)(X + (size_t)(row0 + r) * ldx + k0 + 4 * c4);
                    if (X2) {
                        f32x4 u = *(const f32x4*)(X2 + (size_t)(row0 + r) * ldx2 + k0 + 4 * c4);
                        v *= u;
                    }
                    *(f32x4*)(&lds[r * BMP_LDS_LD + 4 * c4]) = v;
                }
            }
            __syncthreads();
            const float* wp = Wt + (size_t)(k0 + 4 * hi) * ldw;
            for (int kk = 0; kk < kc; kk += 8) {
                f32x4 av[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    av[rb] = *(const f32x4*)(&lds[((wr * RB + rb) * 32 + l31) * BMP_LDS_LD + kk + 4 * hi]);
                float bv[CBW][4];
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[cb][t] = wp[(size_t)(kk + t) * ldw + colc[cb]];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = bmp_mfma(av[rb][t], bv[cb][t], acc[rb][cb]);
            }
        }
    }

#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const int col = n0 + (wc * CBW + cb) * 32 + l31;
            if (col < a.Nout) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = row0 + (wr * RB + rb) * 32 + bmp_acc_row(reg, lane);
                    rg_epilogue<EPI>(a, row, col, acc[rb][cb][reg]);
                }
            }
        }
}

template <int EPI>
static int launch_rowgemm_epi(const RGArgs& a, int n_tiles, hipStream_t st) {
    if (a.Nout <= 32) {
        hipLaunchKernelGGL((k_rowgemm<4, 1, 1, EPI>), dim3(n_tiles, 1), dim3(256), 0, st, a);
    } else if (a.Nout <= 64) {
        hipLaunchKernelGGL((k_rowgemm<2, 2, 1, EPI>), dim3(n_tiles, 1), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_rowgemm<1, 4, 1, EPI>), dim3(n_tiles, (a.Nout + 127) / 128), dim3(256), 0, st, a);
    }
    BMP_LAUNCH_CHECK();
    return 0;
}

int bmp_launch_rowgemm(const RGArgs& a, int n_tiles, int epi, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && a.Nout > 0 && a.nsrc >= 1 && a.nsrc <= 3);
    for (int s = 0; s < a.nsrc; ++s) {
        BMP_REQUIRE(a.s[s].K > 0 && (a.s[s].K & 7) == 0 && (a.s[s].ldx & 3) == 0);
        BMP_REQUIRE(((uintptr_t)a.s[s].X & 15) == 0);
        if (a.s[s].X2) BMP_REQUIRE((a.s[s].ldx2 & 3) == 0 && ((uintptr_t)a.s[s].X2 & 15) == 0);
    }
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.s[s].K;
    const double rows = (double)n_tiles * BMP_R;
    BmpProfScope prof(BMP_KCLS_ROWGEMM, 2.0 * rows * ksum * a.Nout, 4.0 * rows * (ksum + a.Nout), st);
    switch (epi) {
        case BMP_EPI_GENERIC: return launch_rowgemm_epi<BMP_EPI_GENERIC>(a, n_tiles, st);
        case BMP_EPI_GRU_OUT: return launch_rowgemm_epi<BMP_EPI_GRU_OUT>(a, n_tiles, st);
        case BMP_EPI_GRU_DRH: return launch_rowgemm_epi<BMP_EPI_GRU_DRH>(a, n_tiles, st);
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM: slab[s][i][j] = sum_{rows of split s} X[row, i] * dY[row, j]
// 4 waves as 2x2, each wave MB x NB blocks of 32x32; operands straight from global (rows are
// 128-B coalesced per half-wave, re-reads across the workgroup's waves hit L1/L2).
// ---------------------------------------------------------------------------------------------
struct WGKArgs {
    const float* X; const float* X2; int ldx, ldx2;
    const float* dY; int ldy;
    int K, Nn, N, rows_per_split;
    float* slab;
};

template <int MB, int NB>
__global__ __launch_bounds__(256) void k_wgrad(WGKArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i0 = blockIdx.x * (64 * MB) + wm * (32 * MB);
    const int j0 = blockIdx.y * (64 * NB) + wn * (32 * NB);
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    int ic[MB], jc[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { int i = i0 + m * 32 + l31; ic[m] = i < a.K ? i : a.K - 1; }
#pragma unroll
    for (int n = 0; n < NB; ++n) { int j = j0 + n * 32 + l31; jc[n] = j < a.Nn ? j : a.Nn - 1; }

    const float* __restrict__ X = a.X;
    const float* __restrict__ X2 = a.X2;
    const float* __restrict__ dY = a.dY;
    for (int r = r_begin + 4 * hi; r < r_end; r += 8) {
        float av[MB][4], bv[NB][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float x = X[(size_t)(r + t) * a.ldx + ic[m]];
                if (X2) x *= X2[(size_t)(r + t) * a.ldx2 + ic[m]];
                av[m][t] = x;
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n][t] = dY[(size_t)(r + t) * a.ldy + jc[n]];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = bmp_mfma(av[m][t], bv[n][t], acc[m][n]);
    }

    float* slab = a.slab + (size_t)s * a.K * a.Nn;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int j = j0 + n * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = i0 + m * 32 + bmp_acc_row(reg, lane);
                if (i < a.K && j < a.Nn) slab[(size_t)i * a.Nn + j] = acc[m][n][reg];
            }
        }
}

// out[i, j] (=|+=) sum_s slab[s][i][j]
// 64 elements x 4 slab lanes per workgroup; fixed summation order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slab, int S, int K, int Nn, float* out,
                                                      int ldo, int accumulate) {
    __shared__ float red[4][64];
    const size_t total = (size_t)K * Nn;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        const size_t idx = base + c;
        float v = 0.f;
        if (idx < total) {
#pragma unroll 8
            for (int s = g; s < S; s += 4) v += slab[(size_t)s * total + idx];
        }
        red[g][c] = v;
        __syncthreads();
        if (g == 0 && idx < total) {
            v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            const int i = (int)(idx / Nn), j = (int)(idx % Nn);
            float* o = out + (size_t)i * ldo + j;
            *o = accumulate ? (*o + v) : v;
        }
        __syncthreads();
    }
}

static void wgrad_plan(int N, int K, int Nn, int& mb, int& nb, int& S, int& rps) {
    mb = K > 32 ? 2 : 1;
    nb = Nn > 32 ? 2 : 1;
    const int tiles = ((K + 64 * mb - 1) / (64 * mb)) * ((Nn + 64 * nb - 1) / (64 * nb));
    int want = (512 + tiles - 1) / tiles;             // ~2 workgroups per CU in total
    int max_s = N / 128;                              // at least 128 rows per split
    if (max_s < 1) max_s = 1;
    S = want < max_s ? want : max_s;
    if (S < 1) S = 1;
    rps = (N + S - 1) / S;
    rps = (rps + 7) & ~7;
    S = (N + rps - 1) / rps;
}

size_t bmp_wgrad_ws_floats(int N, int K, int Nn) {
    int mb, nb, S, rps;
    wgrad_plan(N, K, Nn, mb, nb, S, rps);
    return (size_t)S * K * Nn;
}

int bmp_launch_wgrad(const WGArgs& a, float* ws, hipStream_t st) {
    BMP_REQUIRE(a.N > 0 && (a.N & 7) == 0 && a.K > 0 && a.Nn > 0 && ws != nullptr);
    int mb, nb, S, rps;
    wgrad_plan(a.N, a.K, a.Nn, mb, nb, S, rps);
    WGKArgs k{a.X, a.X2, a.ldx, a.ldx2, a.dY, a.ldy, a.K, a.Nn, a.N, rps, ws};
    dim3 grid((a.K + 64 * mb - 1) / (64 * mb), (a.Nn + 64 * nb - 1) / (64 * nb), S);
    {
    BmpProfScope prof(BMP_KCLS_WGRAD, 2.0 * a.N * (double)a.K * a.Nn, 4.0 * a.N * ((double)a.K + a.Nn), st);
    if (mb == 2 && nb == 2) hipLaunchKernelGGL((k_wgrad<2, 2>), grid, dim3(256), 0, st, k);
    else if (mb == 2) hipLaunchKernelGGL((k_wgrad<2, 1>), grid, dim3(256), 0, st, k);
    else if (nb == 2) hipLaunchKernelGGL((k_wgrad<1, 2>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((k_wgrad<1, 1>), grid, dim3(256), 0, st, k);
    }
    BMP_LAUNCH_CHECK();
    const size_t total = (size_t)a.K * a.Nn;
    int blocks = (int)((total + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, st, ws, S, a.K, a.Nn, a.out, a.ldo, a.accumulate);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): slab[s][n] = sum_{rows of split s} dY[row, n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dY, int ldy, int N, int Nn, int rps, float* slab) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int s = blockIdx.y;
    const int r_begin = s * rps;
    const int r_end = (r_begin + rps) < N ? (r_begin + rps) : N;
    float v = 0.f;
    if (n < Nn)
        for (int r = r_begin + g; r < r_end; r += 4) v += dY[(size_t)r * ldy + n];
    red[g][c] = v;
    __syncthreads();
    if (g == 0 && n < Nn) slab[(size_t)s * Nn + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

static void colsum_plan(int N, int& S, int& rps) {
    S = N / 512;
    if (S < 1) S = 1;
    if (S > 128) S = 128;
    rps = (N + S - 1) / S;
    S = (N + rps - 1) / rps;
}

size_t bmp_colsum_ws_floats(int N, int Nn) {
    int S, rps;
    colsum_plan(N, S, rps);
    return (size_t)S * Nn;
}

int bmp_launch_colsum(const float* dY, int ldy, int N, int Nn, float* out, int accumulate, float* ws, hipStream_t st) {
    BMP_REQUIRE(N > 0 && Nn > 0 && ws != nullptr);
    int S, rps;
    colsum_plan(N, S, rps);
    hipLaunchKernelGGL(k_colsum, dim3((Nn + 63) / 64, S), dim3(256), 0, st, dY, ldy, N, Nn, rps, ws);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_slabs, dim3((Nn + 63) / 64), dim3(256), 0, st, ws, S, 1, Nn, out, Nn, accumulate);
    BMP_LAUNCH_CHECK();
    return 0;
}
