// Harness-only prototype: the weight-gradient GEMM's inner product as a 3 x bf16 split --
// an f32 value is exactly the sum of three bf16 values (24 = 3 x 8 mantissa bits), so A.B as the six bf16 products of total
// order <= 2 (hh, hm, mh, hl, lh, mm) with f32 accumulation carries ~2^-22 relative error per product while
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32.  NOT part of the product: the library computes in
// exact f32 (parity 1e-4 against an f32 reference, BASELINE.json).  This file measures, on the step's weight-gradient shape
// (N rows x K = 128 times N x Nn = 896 / 1408), (a) the time of the bf16x3 form against the f32 form, both with and without
// memory traffic, and (b) its error against a float64 host product beside the f32 kernel's own error, on synthetic operands
// and -- with a file written by tools/dump_wgrad_operands.py -- on the operands of a real training step.
//   hipcc --offload-arch=gfx950 -O3 -I gcn-bmp_amd/csrc tools/wgrad_bf16x3.hip -o tools/wgrad_bf16x3
//   tools/wgrad_bf16x3 [N] [Nn] [operands.bin]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include <algorithm>
#include "bmp_common.h"

#define WG_LD 132
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct WGKArgs { const float* X; const float* dY; int ldx, ldy, K, Nn, N, rows_per_split; float* slab; };

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// x = h + m + l exactly (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)): 3 converts + 2 subtractions per element
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const __bf16 hh = (__bf16)x[t];
        const float r1 = x[t] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[t] = hh; m[t] = mm; l[t] = (__bf16)r2;
    }
}

// FORM 0: exact f32 (v_mfma_f32_32x32x2_f32), the product's arithmetic.  FORM 1: bf16 x 3, six MFMAs per 16 k.
// MODE bits: 2 = no global loads, 4 = no LDS traffic (constant fragments), 8 = (FORM 1) fragments pre-split: no split arithmetic
template <int FORM, int MODE>
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
    auto load = [&](int st) {
        if (!(MODE & 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t row = (size_t)(r_begin + st * 32 + rr + 8 * i);
                xr[i] = *(const f32x4*)(a.X + row * a.ldx + colx);
                yr[i] = *(const f32x4*)(a.dY + row * a.ldy + coly);
            }
        }
    };
    auto store = [&](int buf) {
        if (!(MODE & 4)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(f32x4*)(&XS[buf][rr + 8 * i][4 * c4]) = xr[i];
                *(f32x4*)(&YS[buf][rr + 8 * i][4 * c4]) = yr[i];
            }
        }
    };
    if (nst > 0) { load(0); store(0); }
    __syncthreads();
    float keep = 0.f;
    bf16x8 ch, cm, cl;          // constant pre-split fragments (MODE 8)
    { float c8[8]; for (int t = 0; t < 8; ++t) c8[t] = 0.37f + 0.01f * t; split3(c8, ch, cm, cl); }
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) load(st + 1);
        if (FORM == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float av[2][4], bv[2][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) av[m][t] = (MODE & 4) ? xr[m][t] : XS[buf][ks * 8 + 4 * hi + t][wm * 64 + m * 32 + l31];
#pragma unroll
                    for (int n = 0; n < 2; ++n) bv[n][t] = (MODE & 4) ? yr[n][t] : YS[buf][ks * 8 + 4 * hi + t][wn * 64 + n * 32 + l31];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[m][t], bv[n][t], acc[m][n]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {        // 16 k per bf16 MFMA: this lane's eight k are ks * 16 + 8 * hi + 0..7
                bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
                if (MODE & 8) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) { ah[m] = ch; am[m] = cm; al[m] = cl; bh[m] = cm; bm[m] = ch; bl[m] = cl; }
                } else {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float xa[8], xb[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            xa[t] = (MODE & 4) ? xr[m][t & 3] : XS[buf][ks * 16 + 8 * hi + t][wm * 64 + m * 32 + l31];
                            xb[t] = (MODE & 4) ? yr[m][t & 3] : YS[buf][ks * 16 + 8 * hi + t][wn * 64 + m * 32 + l31];
                        }
                        split3(xa, ah[m], am[m], al[m]);
                        split3(xb, bh[m], bm[m], bl[m]);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {       // smallest terms first
                        f32x16 c = acc[m][n];
                        c = mfma_bf16(am[m], bm[n], c);
                        c = mfma_bf16(ah[m], bl[n], c);
                        c = mfma_bf16(al[m], bh[n], c);
                        c = mfma_bf16(ah[m], bm[n], c);
                        c = mfma_bf16(am[m], bh[n], c);
                        c = mfma_bf16(ah[m], bh[n], c);
                        acc[m][n] = c;
                    }
            }
        }
        if (st + 1 < nst) store(buf ^ 1);
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

template <int FORM, int MODE>
static float run(const WGKArgs& a, int tiles, int S, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_wg<FORM, MODE>), dim3(1, tiles, S), dim3(256), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_wg<FORM, MODE>), dim3(1, tiles, S), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / reps;
}

// out [K x Nn] = sum of the S slabs (host, double)
static void fold(const float* slab_d, int S, int K, int Nn, std::vector<double>& out) {
    std::vector<float> tmp((size_t)K * Nn);
    out.assign((size_t)K * Nn, 0.0);
    for (int q = 0; q < S; ++q) {
        hipMemcpy(tmp.data(), slab_d + (size_t)q * K * Nn, tmp.size() * 4, hipMemcpyDeviceToHost);
        for (size_t e = 0; e < tmp.size(); ++e) out[e] += tmp[e];
    }
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 58240, Nn = argc > 2 ? atoi(argv[2]) : 896;
    const int K = 128;
    const char* opf = argc > 3 ? argv[3] : nullptr;
    std::vector<float> hx, hy;
    const char* what = "synthetic: X uniform(-1, 1) (tanh-range atom states), dY = normal x lognormal(sigma 2) x 1e-4 (gradient-like dynamic range)";
    if (opf) {       // int32 N, int32 Nn, then X [N x 128] and dY [N x Nn] as float32 (tools/dump_wgrad_operands.py)
        FILE* f = fopen(opf, "rb");
        int hdr[2];
        if (f && fread(hdr, 4, 2, f) == 2) {
            N = hdr[0] & ~31; Nn = hdr[1];
            hx.resize((size_t)hdr[0] * K); hy.resize((size_t)hdr[0] * Nn);
            if (fread(hx.data(), 4, hx.size(), f) != hx.size() || fread(hy.data(), 4, hy.size(), f) != hy.size()) { printf("short operand file\n"); return 1; }
            what = "operands of a real training step (h and gda of the last GGNN step of config C2)";
        }
        if (f) fclose(f);
    }
    if (hx.empty()) {
        hx.resize((size_t)N * K); hy.resize((size_t)N * Nn);
        unsigned long long sd = 88172645463325252ull;
        auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0; };
        auto nrm = [&]() { double u = rnd() + 1e-12, v = rnd(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); };
        for (auto& v : hx) v = (float)(2.0 * rnd() - 1.0);
        for (auto& v : hy) v = (float)(nrm() * exp(2.0 * nrm()) * 1e-4);
    }
    const int tiles = Nn / 128;
    float *X, *dY, *slab;
    hipMalloc(&X, (size_t)N * K * 4); hipMalloc(&dY, (size_t)N * Nn * 4);
    hipMemcpy(X, hx.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(dY, hy.data(), (size_t)N * Nn * 4, hipMemcpyHostToDevice);
    int S = 512 / tiles;
    int rps = ((N + S - 1) / S + 31) & ~31;
    S = (N + rps - 1) / rps;
    hipMalloc(&slab, (size_t)S * K * Nn * 4);
    WGKArgs a{X, dY, K, Nn, K, Nn, N, rps, slab};
    const double gf = 2.0 * N * K * (double)Nn / 1e9;
    printf("N %d K %d Nn %d: %d column tiles x %d row splits, %.1f GFLOP (%.1f us at the f32 MFMA peak 157.3 TF)\noperands: %s\n", N, K, Nn, tiles, S, gf,
           gf / 157.3e3 * 1e6, what);
    printf("  f32   full (register-staged, the product's k_wgrad_lds form)   %7.1f us\n", run<0, 0>(a, tiles, S, 20));
    printf("  f32   MFMA loop alone (constant fragments, no memory traffic)  %7.1f us\n", run<0, 6>(a, tiles, S, 20));
    printf("  bf16x3 full (same staging, split in registers per use)         %7.1f us\n", run<1, 0>(a, tiles, S, 20));
    printf("  bf16x3 split + MFMA, no memory traffic                         %7.1f us\n", run<1, 6>(a, tiles, S, 20));
    printf("  bf16x3 MFMA loop alone (fragments pre-split)                   %7.1f us\n", run<1, 14>(a, tiles, S, 20));
    // ---- accuracy against a float64 host product (a sample of output columns: the host product is O(N K Nn)) ----
    std::vector<double> r32, r16;
    hipLaunchKernelGGL((k_wg<0, 0>), dim3(1, tiles, S), dim3(256), 0, 0, a); hipDeviceSynchronize(); fold(slab, S, K, Nn, r32);
    hipLaunchKernelGGL((k_wg<1, 0>), dim3(1, tiles, S), dim3(256), 0, 0, a); hipDeviceSynchronize(); fold(slab, S, K, Nn, r16);
    const int ncol = 48;
    double scale = 0, e32 = 0, e16 = 0, s32 = 0, s16 = 0, d3216 = 0;
    for (int cc = 0; cc < ncol; ++cc) {
        const int j = (int)(((long long)cc * Nn) / ncol);
        std::vector<double> ref(K, 0.0);
        for (int r = 0; r < N; ++r) {
            const double y = hy[(size_t)r * Nn + j];
            const float* xr = &hx[(size_t)r * K];
            for (int i = 0; i < K; ++i) ref[i] += xr[i] * y;
        }
        for (int i = 0; i < K; ++i) {
            const double v = ref[i], a32 = r32[(size_t)i * Nn + j], a16 = r16[(size_t)i * Nn + j];
            scale = fmax(scale, fabs(v));
            e32 = fmax(e32, fabs(a32 - v)); e16 = fmax(e16, fabs(a16 - v));
            s32 += (a32 - v) * (a32 - v); s16 += (a16 - v) * (a16 - v);
            d3216 = fmax(d3216, fabs(a32 - a16));
        }
    }
    printf("accuracy over %d output columns x %d rows of the [K x Nn] result, against float64 (max |ref| %.3e):\n", ncol, K, scale);
    printf("  f32 kernel     max |err| / max |ref| %.3e   rms %.3e\n", e32 / scale, sqrt(s32 / (ncol * K)) / scale);
    printf("  bf16x3 kernel  max |err| / max |ref| %.3e   rms %.3e\n", e16 / scale, sqrt(s16 / (ncol * K)) / scale);
    printf("  bf16x3 vs f32  max |diff| / max |ref| %.3e\n", d3216 / scale);
    return 0;
}
