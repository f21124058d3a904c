// Stand-alone anatomy of the weight-gradient GEMM (k_wgrad_lds of gcn-bmp_amd/csrc/bmp_gemm.hip): the same staging
// loop with parts switched off, timed with HIP events.   hipcc --offload-arch=gfx950 -O3 -I gcn-bmp_amd/csrc
// tools/wgrad_bench.hip -o tools/wgrad_bench && tools/wgrad_bench [N] [Nn]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <algorithm>
#include "bmp_common.h"

#define WG_LD 132
struct WGKArgs { const float* X; const float* dY; int ldx, ldy, K, Nn, N, rows_per_split; float* slab; };

// MODE bits: 1 = no MFMA, 2 = no global loads, 4 = no LDS traffic (fragments stay constant)
template <int MODE>
__global__ __launch_bounds__(256) void k_wg(WGKArgs a) {
    __shared__ __attribute__((aligned(16))) float XS[2][32][WG_LD];
    __shared__ __attribute__((aligned(16))) float YS[2][32][WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j_tile = blockIdx.y * 128;
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) >> 5;
    const int c4 = tid & 31, rr = tid >> 5;
    const int colx = 4 * c4, coly = j_tile + 4 * c4;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    f32x4 xr[4], yr[4];
    for (int i = 0; i < 4; ++i) { xr[i] = (f32x4){1.f, 2.f, 3.f, 4.f}; yr[i] = (f32x4){.5f, .25f, .125f, 1.f}; }
#define WG_LOAD(st)                                                                              \
    if (!(MODE & 2)) {                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            const size_t row = (size_t)(r_begin + (st) * 32 + rr + 8 * i);                       \
            xr[i] = *(const f32x4*)(a.X + row * a.ldx + colx);                                   \
            yr[i] = (MODE & 8) ? *(const f32x4*)(a.dY + (size_t)blockIdx.y * a.N * 128 + row * 128 + 4 * c4)   \
                               : *(const f32x4*)(a.dY + row * a.ldy + coly);                     \
        }                                                                                        \
    }
#define WG_STORE(buf)                                                                            \
    if (!(MODE & 4)) {                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
            *(f32x4*)(&XS[buf][rr + 8 * i][4 * c4]) = xr[i];                                     \
            *(f32x4*)(&YS[buf][rr + 8 * i][4 * c4]) = yr[i];                                     \
        }                                                                                        \
    }
    if (nst > 0) { WG_LOAD(0) WG_STORE(0) }
    __syncthreads();
    float keep = 0.f;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) { WG_LOAD(st + 1) }
        float av[2][2][4], bv[2][2][4];
#define WG_FRAG(slot, k0)                                                                          \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                              \
            av[slot][m][t] = (MODE & 4) ? xr[m][t] : XS[buf][(k0) + 4 * hi + t][wm * 64 + m * 32 + l31]; \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                              \
            bv[slot][n][t] = (MODE & 4) ? yr[n][t] : YS[buf][(k0) + 4 * hi + t][wn * 64 + n * 32 + l31]; \
    }
        WG_FRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) { WG_FRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
            if (!(MODE & 1)) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            } else {
                for (int t = 0; t < 4; ++t) keep += av[cur][0][t] + bv[cur][1][t] + av[cur][1][t] + bv[cur][0][t];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st + 1 < nst) { WG_STORE(buf ^ 1) }
        if ((MODE & 4) && (MODE & 2) == 0) keep += xr[0][0] + yr[3][3] + xr[3][1] + yr[0][2] + xr[1][0] + xr[2][0] + yr[1][0] + yr[2][0];
        __syncthreads();
    }
    float* slab = a.slab + (size_t)s * a.K * a.Nn;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
            for (int reg = 0; reg < 16; ++reg) {
                const int i = wm * 64 + m * 32 + bmp_acc_row(reg, lane);
                slab[(size_t)i * a.Nn + j] = acc[m][n][reg] + keep;
            }
        }
}


// variant: two stages of global loads in flight (register ring), otherwise as k_wg<0>
__global__ __launch_bounds__(256) void k_wg_ring(WGKArgs a) {
    __shared__ __attribute__((aligned(16))) float XS[2][32][WG_LD];
    __shared__ __attribute__((aligned(16))) float YS[2][32][WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j_tile = blockIdx.y * 128;
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) >> 5;
    const int c4 = tid & 31, rr = tid >> 5;
    const int colx = 4 * c4, coly = j_tile + 4 * c4;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    f32x4 xa[4], ya[4], xb[4], yb[4];
#define RLOAD(xr, yr, st)                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
        const size_t row = (size_t)(r_begin + (st) * 32 + rr + 8 * i);                           \
        xr[i] = *(const f32x4*)(a.X + row * a.ldx + colx);                                       \
        yr[i] = *(const f32x4*)(a.dY + row * a.ldy + coly);                                      \
    }
#define RSTORE(xr, yr, buf)                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
        *(f32x4*)(&XS[buf][rr + 8 * i][4 * c4]) = xr[i];                                         \
        *(f32x4*)(&YS[buf][rr + 8 * i][4 * c4]) = yr[i];                                         \
    }
#define RCOMPUTE(buf)                                                                              \
    {                                                                                              \
        float av[2][2][4], bv[2][2][4];                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                         \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                        \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) av[0][m][t] = XS[buf][ks * 8 + 4 * hi + t][wm * 64 + m * 32 + l31]; \
                _Pragma("unroll") for (int n = 0; n < 2; ++n) bv[0][n][t] = YS[buf][ks * 8 + 4 * hi + t][wn * 64 + n * 32 + l31]; \
            }                                                                                      \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                          \
                _Pragma("unroll") for (int m = 0; m < 2; ++m)                                      \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[0][m][t], bv[0][n][t], acc[m][n]); \
        }                                                                                          \
    }
    if (nst > 0) { RLOAD(xa, ya, 0) RSTORE(xa, ya, 0) }
    if (nst > 1) { RLOAD(xa, ya, 1) }
    __syncthreads();
    for (int st = 0; st < nst; st += 2) {
        if (st + 2 < nst) { RLOAD(xb, yb, st + 2) }
        RCOMPUTE(0)
        if (st + 1 < nst) { RSTORE(xa, ya, 1) }
        __syncthreads();
        if (st + 1 >= nst) break;
        if (st + 3 < nst) { RLOAD(xa, ya, st + 3) }
        RCOMPUTE(1)
        if (st + 2 < nst) { RSTORE(xb, yb, 0) }
        __syncthreads();
    }
    float* slab = a.slab + (size_t)s * a.K * a.Nn;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
            for (int reg = 0; reg < 16; ++reg) slab[(size_t)(wm * 64 + m * 32 + bmp_acc_row(reg, lane)) * a.Nn + j] = acc[m][n][reg];
        }
}

// variant: 128 x 256 output tile per workgroup, 512 threads (2 x 4 waves of 64 x 64), one workgroup per CU
#define WY_LD 260
__global__ __launch_bounds__(512) void k_wg_wide(WGKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float (*XS)[32][WG_LD] = (float (*)[32][WG_LD])sm;
    float (*YS)[32][WY_LD] = (float (*)[32][WY_LD])(sm + 2 * 32 * WG_LD);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j_tile = blockIdx.y * 256;
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) >> 5;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    // X stage 32 x 128 = 1024 float4 -> 2 per thread; Y stage 32 x 256 = 2048 float4 -> 4 per thread
    f32x4 xr[2], yr[4];
    const int xc4 = tid & 31, xrr = tid >> 5;       // 16 row groups
    const int yc4 = tid & 63, yrr = tid >> 6;       // 8 row groups
    const bool oky = j_tile + 4 * yc4 < a.Nn;
#define WLOAD(st)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) xr[i] = *(const f32x4*)(a.X + (size_t)(r_begin + (st) * 32 + xrr + 16 * i) * a.ldx + 4 * xc4); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) yr[i] = oky ? *(const f32x4*)(a.dY + (size_t)(r_begin + (st) * 32 + yrr + 8 * i) * a.ldy + j_tile + 4 * yc4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#define WSTORE(buf)                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) *(f32x4*)(&XS[buf][xrr + 16 * i][4 * xc4]) = xr[i]; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) *(f32x4*)(&YS[buf][yrr + 8 * i][4 * yc4]) = yr[i];
    if (nst > 0) { WLOAD(0) WSTORE(0) }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) { WLOAD(st + 1) }
        float av[2][2][4], bv[2][2][4];
#define WFRAG(slot, k0)                                                                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) av[slot][m][t] = XS[buf][(k0) + 4 * hi + t][wm * 64 + m * 32 + l31]; \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) bv[slot][n][t] = YS[buf][(k0) + 4 * hi + t][wn * 64 + n * 32 + l31]; \
    }
        WFRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) { WFRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st + 1 < nst) { WSTORE(buf ^ 1) }
        __syncthreads();
    }
    float* slab = a.slab + (size_t)s * a.K * a.Nn;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
            if (j < a.Nn)
                for (int reg = 0; reg < 16; ++reg) slab[(size_t)(wm * 64 + m * 32 + bmp_acc_row(reg, lane)) * a.Nn + j] = acc[m][n][reg];
        }
}


// variant: 128 x 256 output tile, 512 threads, THREE LDS stages filled by LDS-DMA (global_load_lds_dwordx4) two stages
// ahead; counted vmcnt + raw s_barrier, one barrier per 32-row stage, one workgroup per CU (144 KB of LDS)
#define DW_NS 3
__global__ __launch_bounds__(512) void k_wg_dmaw(WGKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int XSZ = 32 * 128, YSZ = 32 * 256, SSZ = XSZ + YSZ;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j_tile = blockIdx.y * 256;
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) >> 5;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    // per-lane source addresses of this wave's six pieces of a stage (row offsets relative to the stage's first row)
    int ycol = j_tile + 4 * lane;
    if (ycol > a.Nn - 4) ycol = a.Nn - 4;                  // columns past the matrix: any valid address, results are masked
    const float* xsrc = a.X + (size_t)(r_begin + 2 * w + (lane >> 5)) * a.ldx + 4 * (lane & 31);
    const float* ysrc = a.dY + (size_t)(r_begin + w) * a.ldy + ycol;
    typedef __attribute__((address_space(3))) float lds_f;
    typedef const __attribute__((address_space(1))) float glb_f;
#define DW_ISSUE(st)                                                                                         \
    {                                                                                                        \
        float* base = sm + ((st) % DW_NS) * SSZ;                                                             \
        const size_t ro = (size_t)(st) * 32;                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
            __builtin_amdgcn_global_load_lds((glb_f*)(xsrc + (ro + 16 * i) * a.ldx), (lds_f*)(base + (2 * w + 16 * i) * 128), 16, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
            __builtin_amdgcn_global_load_lds((glb_f*)(ysrc + (ro + 8 * i) * a.ldy), (lds_f*)(base + XSZ + (w + 8 * i) * 256), 16, 0, 0); \
    }
    if (nst > 0) DW_ISSUE(0)
    if (nst > 1) DW_ISSUE(1)
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nst) DW_ISSUE(st + 2)
        const float* XS = sm + (st % DW_NS) * SSZ;
        const float* YS = XS + XSZ;
        float av[2][2][4], bv[2][2][4];
#define DFRAG(slot, k0)                                                                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) av[slot][m][t] = XS[((k0) + 4 * hi + t) * 128 + wm * 64 + m * 32 + l31]; \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) bv[slot][n][t] = YS[((k0) + 4 * hi + t) * 256 + wn * 64 + n * 32 + l31]; \
    }
        DFRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) { DFRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* slab = a.slab + (size_t)s * a.K * a.Nn;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
            if (j < a.Nn)
                for (int reg = 0; reg < 16; ++reg) slab[(size_t)(wm * 64 + m * 32 + bmp_acc_row(reg, lane)) * a.Nn + j] = acc[m][n][reg];
        }
}
static void launch_dmaw(const WGKArgs& a, int tiles, int S) {
    const size_t lds = (size_t)DW_NS * (32 * 128 + 32 * 256) * 4;
    hipFuncSetAttribute((const void*)k_wg_dmaw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_wg_dmaw, dim3(1, (a.Nn + 255) / 256, S), dim3(512), lds, 0, a);
}

// variant: the product's 128 x 128 tile / 256 threads / two workgroups per CU, stages of RS rows filled by LDS-DMA NS - 1 stages
// ahead (no staging registers, no ds_write phase); counted vmcnt + raw s_barrier, one barrier per stage
template <int RS, int NS>
__global__ __launch_bounds__(256) void k_wg_dma(WGKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int XSZ = RS * 128, SSZ = 2 * XSZ;
    constexpr int NI = RS / 8;                     // DMA instructions per wave, operand and stage (a wave moves 2 rows each)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int j_tile = blockIdx.y * 128;
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) / RS;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    const float* xsrc = a.X + (size_t)(r_begin + 2 * w + hi) * a.ldx + 4 * l31;
    const float* ysrc = a.dY + (size_t)(r_begin + 2 * w + hi) * a.ldy + j_tile + 4 * l31;
    typedef __attribute__((address_space(3))) float lds_f;
    typedef const __attribute__((address_space(1))) float glb_f;
#define DM_ISSUE(st)                                                                                         \
    {                                                                                                        \
        float* base = sm + ((st) % NS) * SSZ;                                                                \
        const size_t ro = (size_t)(st) * RS;                                                                 \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                     \
            __builtin_amdgcn_global_load_lds((glb_f*)(xsrc + (ro + 8 * i) * a.ldx), (lds_f*)(base + (2 * w + 8 * i) * 128), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((glb_f*)(ysrc + (ro + 8 * i) * a.ldy), (lds_f*)(base + XSZ + (2 * w + 8 * i) * 128), 16, 0, 0); \
        }                                                                                                    \
    }
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) if (p < nst) DM_ISSUE(p)
    for (int st = 0; st < nst; ++st) {
        // the oldest stage in flight has landed when at most (stages still behind it) * 2 * NI loads are outstanding
        const int behind = (nst - 1 - st) < (NS - 2) ? (nst - 1 - st) : (NS - 2);
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NI) : "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + NS - 1 < nst) DM_ISSUE(st + NS - 1)
        const float* XS = sm + (st % NS) * SSZ;
        const float* YS = XS + XSZ;
        float av[2][2][4], bv[2][2][4];
#define MFRAG(slot, k0)                                                                            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) av[slot][m][t] = XS[((k0) + 4 * hi + t) * 128 + wm * 64 + m * 32 + l31]; \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) bv[slot][n][t] = YS[((k0) + 4 * hi + t) * 128 + wn * 64 + n * 32 + l31]; \
    }
        MFRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < RS / 8; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < RS / 8) { MFRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* slab = a.slab + (size_t)s * a.K * a.Nn;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
            if (j < a.Nn)
                for (int reg = 0; reg < 16; ++reg) slab[(size_t)(wm * 64 + m * 32 + bmp_acc_row(reg, lane)) * a.Nn + j] = acc[m][n][reg];
        }
}
template <int RS, int NS>
static void launch_dma(const WGKArgs& a, int tiles, int S) {
    const size_t lds = (size_t)NS * 2 * RS * 128 * 4;
    hipFuncSetAttribute((const void*)k_wg_dma<RS, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_wg_dma<RS, NS>), dim3(1, tiles, S), dim3(256), lds, 0, a);
}

static float run_k(void (*launch)(const WGKArgs&, int, int), const WGKArgs& a, int tiles, int S, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(a, tiles, S);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch(a, tiles, S);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}
static void launch_ring(const WGKArgs& a, int tiles, int S) { hipLaunchKernelGGL(k_wg_ring, dim3(1, tiles, S), dim3(256), 0, 0, a); }
static void launch_wide(const WGKArgs& a, int tiles, int S) {
    const size_t lds = (size_t)2 * 32 * (WG_LD + WY_LD) * 4;
    hipFuncSetAttribute((const void*)k_wg_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_wg_wide, dim3(1, (a.Nn + 255) / 256, S), dim3(512), lds, 0, a);
}

template <int MODE>
static float run(const WGKArgs& a, int tiles, int S, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_wg<MODE>), dim3(1, tiles, S), dim3(256), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_wg<MODE>), dim3(1, tiles, S), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 58240, Nn = argc > 2 ? atoi(argv[2]) : 896, K = 128;
    const int tiles = Nn / 128;
    float *X, *dY, *slab;
    hipMalloc(&X, (size_t)N * K * 4); hipMalloc(&dY, (size_t)N * Nn * 4);
    std::vector<float> h((size_t)N * Nn);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(dY, h.data(), (size_t)N * Nn * 4, hipMemcpyHostToDevice);
    hipMemcpy(X, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    const int splits[] = {512 / tiles, 768 / tiles, 640 / tiles};
    for (int si = 0; si < 3; ++si) {
        int S = splits[si];
        int rps = ((N + S - 1) / S + 31) & ~31;
        S = (N + rps - 1) / rps;
        hipMalloc(&slab, (size_t)S * K * Nn * 4);
        WGKArgs a{X, dY, K, Nn, K, Nn, N, rps, slab};
        const double gf = 2.0 * N * K * (double)Nn / 1e9;
        printf("N %d Nn %d tiles %d S %d rows/split %d  (%.1f GFLOP, ideal %.1f us at 157.3 TF)\n", N, Nn, tiles, S, rps, gf, gf / 157.3e3 * 1e6);
        printf("  full            %7.1f us\n", run<0>(a, tiles, S, 20));
        printf("  full, dY column-group-major %7.1f us\n", run<8>(a, tiles, S, 20));
        printf("  loads only, group-major     %7.1f us\n", run<13>(a, tiles, S, 20));
        printf("  no MFMA         %7.1f us\n", run<1>(a, tiles, S, 20));
        printf("  no global loads %7.1f us\n", run<2>(a, tiles, S, 20));
        printf("  no LDS          %7.1f us\n", run<4>(a, tiles, S, 20));
        printf("  MFMA only       %7.1f us\n", run<6>(a, tiles, S, 20));
        printf("  loads only      %7.1f us\n", run<5>(a, tiles, S, 20));
        printf("  ring (2 stages) %7.1f us\n", run_k(launch_ring, a, tiles, S, 20));
        printf("  DMA 32 rows x 2 %7.1f us\n", run_k(launch_dma<32, 2>, a, tiles, S, 20));
        printf("  DMA 16 rows x 4 %7.1f us\n", run_k(launch_dma<16, 4>, a, tiles, S, 20));
        printf("  DMA 16 rows x 3 %7.1f us\n", run_k(launch_dma<16, 3>, a, tiles, S, 20));
        printf("  DMA 8 rows x 8  %7.1f us\n", run_k(launch_dma<8, 8>, a, tiles, S, 20));
        {   // check the 16 x 4 form against the plain kernel
            std::vector<float> r0((size_t)K * Nn), r1((size_t)K * Nn), tmp((size_t)K * Nn);
            for (int form = 0; form < 2; ++form) {
                if (form == 0) hipLaunchKernelGGL((k_wg<0>), dim3(1, tiles, S), dim3(256), 0, 0, a); else launch_dma<16, 4>(a, tiles, S);
                hipDeviceSynchronize();
                std::vector<float>& r = form ? r1 : r0;
                std::fill(r.begin(), r.end(), 0.f);
                for (int q = 0; q < S; ++q) { hipMemcpy(tmp.data(), slab + (size_t)q * K * Nn, tmp.size() * 4, hipMemcpyDeviceToHost); for (size_t e = 0; e < tmp.size(); ++e) r[e] += tmp[e]; }
            }
            double md = 0, mx = 0; for (size_t e = 0; e < r0.size(); ++e) { md = fmax(md, fabs((double)r0[e] - r1[e])); mx = fmax(mx, fabs((double)r0[e])); }
            printf("  DMA 16 x 4 vs plain: max diff %.3e (scale %.3e)\n", md, mx);
        }
        {
            const int wt = (Nn + 255) / 256;
            int Sw = 256 / wt; int rw = ((N + Sw - 1) / Sw + 31) & ~31; Sw = (N + rw - 1) / rw;
            WGKArgs b = a; b.rows_per_split = rw;
            if ((size_t)Sw <= (size_t)S) printf("  wide 128x256 (S %d) %7.1f us\n", Sw, run_k(launch_wide, b, wt, Sw, 20));
            if ((size_t)Sw <= (size_t)S) {
                printf("  wide DMA ring (S %d) %7.1f us\n", Sw, run_k(launch_dmaw, b, wt, Sw, 20));
                // check against the plain kernel on a few slab sums
                std::vector<float> r0((size_t)K * Nn), r1((size_t)K * Nn, 0.f), tmp((size_t)K * Nn);
                hipLaunchKernelGGL((k_wg<0>), dim3(1, tiles, S), dim3(256), 0, 0, a); hipDeviceSynchronize();
                std::fill(r0.begin(), r0.end(), 0.f);
                for (int q = 0; q < S; ++q) { hipMemcpy(tmp.data(), slab + (size_t)q * K * Nn, tmp.size() * 4, hipMemcpyDeviceToHost); for (size_t e = 0; e < tmp.size(); ++e) r0[e] += tmp[e]; }
                launch_dmaw(b, wt, Sw); hipDeviceSynchronize();
                for (int q = 0; q < Sw; ++q) { hipMemcpy(tmp.data(), slab + (size_t)q * K * Nn, tmp.size() * 4, hipMemcpyDeviceToHost); for (size_t e = 0; e < tmp.size(); ++e) r1[e] += tmp[e]; }
                double md = 0, mx = 0; for (size_t e = 0; e < r0.size(); ++e) { md = fmax(md, fabs((double)r0[e] - r1[e])); mx = fmax(mx, fabs((double)r0[e])); }
                printf("  wide DMA ring vs plain: max diff %.3e (scale %.3e)\n", md, mx);
            }
        }
        hipFree(slab);
    }
    return 0;
}
