// Fine-grained drug-pair co-attention (NieFineCoattention / VQAParallelCoattention,
// models/coattention/nie_coattention.py:335-396, vqa_parallel_coattention.py:42-102) on packed rows.
//
// The reference tiles both atom sets to (mb*N2*N1, d) and lets chainer Bilinear materialise
// (mb*N2*N1, d*d) outer products.  Here:
//   stage A (row GEMMs):  Q2 = X2 . W^T (the bilinear form applied once per atom),
//                         Z_k = X_k . [Wj^T | Wl_k^T | V_k]  -> J (o cols, +bj) | P (H cols) | v (1 col)
//   stage B (one workgroup per pair): S = Q2 . X1^T as 32x32 MFMA tiles, C = act(S + v1 + v2 + c),
//                         both softmaxes of C, the head projections, the atom softmax and the
//                         pooled outputs -- C never leaves LDS except once for the backward.
// Every sum over atoms carries the row multiplicities w (virtual pad rows, bmp/packed.py), which
// reproduces the reference's unmasked zero-padding exactly.
#include <string.h>
#include "bmp_kernels.h"

#define CO_MAXN 128          // rows per molecule never exceed the tile size
#define CO_MAXH 16
#define CO_NT_BIG 1024     // threads per pair of the 96-row class in the backward
#define CO_DQ 4             // rows per wave and iteration in the backward's dot phase
#define CO_NT_FWD 512      // threads per pair in the forward (one launch sized by the largest class)

struct CoArgs {
    const float* X1; const float* X2;          // [N1 x d], [N2 x d]
    const float* Q2;                           // [N2 x d]
    const float* Z1; const float* Z2; int ZC;  // [N x ZC]: J (o) | P (H) | v (1) | pad
    const float* w1; const float* w2;
    const int* r1; const int* n1; const int* r2; const int* n2;   // per pair row range
    const long long* coff;                     // per pair offset into Cbuf
    const int* order; int order_off;           // pair processed by block i = order[order_off + i] (size-class launch)
    int np;                                    // class size: every pair of this launch has n1, n2 <= np (multiple of 32)
    const float* wa1; const float* wa2; const float* cbias;
    int d, o, H, act, ldc;
    int mode;                                  // 0 = Nie/VQA (softmaxes of C + head projections), 1 = Pooling (means of C)
    float* Cbuf;                               // ragged: pair b holds C (n2 x n1), row-major
    float* H1; float* H2;                      // [N x H] tanh outputs
    float* al1; float* al2;                    // [N] attention weights
    float* out1; float* out2;                  // [B x o]
    // backward
    const float* dout1; const float* dout2;
    float* dQ2; float* dX1;                    // [N2 x d], [N1 x d]
    float* dZ1; float* dZ2;                    // [N x ZC]
    float* dpart;                              // [B x (2H + 1)]: dwa1 | dwa2 | dc
    // the oversized class (BIG kernels: a molecule of more than CO_MAXN rows): the arrays the other classes keep in LDS
    // live in a global workspace, one slice of big_stride floats per workgroup
    float* big_ws; long long big_stride;
    // backward: row -> molecule maps of the two sides (-1: a row of no molecule) and their row counts; null = the caller has
    // cleared the arrays
    const int* rm1; const int* rm2; int N1, N2;
    const float* gscale;                       // backward: device scalar dout1 / dout2 are multiplied with on load (null: 1)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// LDS carve-up shared by forward and backward; every extent scales with the launch's size class np
struct CoLds {
    float* Cs;      // [np x ldc]
    float* dSs;     // [np x ldc]   (backward only)
    float* P1s;     // [np x H]
    float* P2s;
    float* dH1s;    // [np x H]     (backward only)
    float* dH2s;
    float* w1s; float* w2s; float* v2s;
    float* cmax; float* invD2; float* rmax; float* invD1;
    float* s1; float* s2;            // scores / alphas
    float* dots1; float* dots2;
    float* H1s; float* H2s;          // [np x H] tanh outputs          (backward only)
    float* do1; float* do2;          // [o] upstream gradients of the pooled outputs (backward only)
    float* Up; float* dPp; float* cmb;   // [256], [256 x H], [np] per-thread partial sums / combined row sums (backward)
};

__device__ __forceinline__ CoLds co_carve(float* base, int np, int ldc, int H, bool bwd, int o = 0, int nt = 256, bool big = false) {
    CoLds L;
    float* p = base;
    L.Cs = p; p += (size_t)np * ldc;
    L.dSs = p; if (bwd) p += (size_t)np * ldc;
    L.P1s = p; p += np * H;
    L.P2s = p; p += np * H;
    L.dH1s = p; if (bwd) p += np * H;
    L.dH2s = p; if (bwd) p += np * H;
    L.w1s = p; p += np;
    L.w2s = p; p += np;
    L.v2s = p; p += np;
    L.cmax = p; p += np; L.invD2 = p; p += np; L.rmax = p; p += np; L.invD1 = p; p += np;
    L.s1 = p; p += np;
    L.s2 = p; p += np;
    L.dots1 = p; p += np; L.dots2 = p; p += np;
    L.H1s = p; if (bwd) p += np * H;
    L.H2s = p; if (bwd) p += np * H;
    L.do1 = p; if (bwd) p += o;
    L.do2 = p; if (bwd) p += o;
    L.cmb = p; if (bwd) p += np;
    // np == 128: SUB = 2, and the slots alias arrays that are dead by then (dots1|dots2 = 256, H1s|H2s = 256 x H)
    if (big) { L.Up = p; if (bwd) p += np; L.dPp = p; if (bwd) p += np * H; }        // one slot per row (no row x chunk split)
    else if (np >= 128) { L.Up = L.dots1; L.dPp = L.H1s; }
    else { L.Up = p; if (bwd) p += nt; L.dPp = p; if (bwd) p += nt * H; }
    return L;
}

static size_t co_lds_floats(int np, int ldc, int H, bool bwd, int o = 0, int nt = 256, bool big = false) {
    return (size_t)np * ldc * (bwd ? 2 : 1) + (size_t)np * H * (bwd ? 6 : 2) + 11 * (size_t)np + (bwd ? 2 * (size_t)o + np : 0) +
           (big ? (bwd ? np + np * (size_t)H : 0) : ((bwd && np < 128) ? nt + nt * (size_t)H : 0));
    // (no slack behind the last array: the backward of the 128-row class at o = 128, H = 8 is EXACTLY the CU's 160 KB -- eight
    //  spare floats used to put it past the limit, and a pair with a molecule of 96..127 atoms failed the launcher's check)
}
// the oversized class: rows rounded up to 32, the forward's pooled-output scratch (2 * NT floats) fits in the C image
static int co_big_np(int np_big) { return (np_big + 31) & ~31; }
static size_t co_big_stride(int np_big, int H, bool bwd, int o) {
    const int np = co_big_np(np_big);
    return (co_lds_floats(np, np + 1, H, bwd, o, 512, true) + 15) & ~(size_t)15;
}

// column / row softmax statistics of C with multiplicities:
//   cmax[j], invD2[j] = 1 / sum_i w2_i exp(C[i,j]-cmax[j])   (softmax over i, nie_coattention.py:347)
//   rmax[i], invD1[i] = 1 / sum_j w1_j exp(C[i,j]-rmax[i])   (softmax over j, :349)
__device__ __forceinline__ void co_stats(const CoLds& L, int n1, int n2, int ldc) {
    const int tid = threadIdx.x;
    if (tid < CO_MAXN) {
        const int j = tid;
        if (j < n1) {
            float mx = -INFINITY;
            for (int i = 0; i < n2; ++i) if (L.w2s[i] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
            float s = 0.f;
            for (int i = 0; i < n2; ++i) if (L.w2s[i] > 0.f) s += L.w2s[i] * bmp_exp(L.Cs[i * ldc + j] - mx);
            L.cmax[j] = mx; L.invD2[j] = 1.f / s;
        }
    } else {
        const int i = tid - CO_MAXN;
        if (i < n2) {
            float mx = -INFINITY;
            for (int j = 0; j < n1; ++j) if (L.w1s[j] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
            float s = 0.f;
            for (int j = 0; j < n1; ++j) if (L.w1s[j] > 0.f) s += L.w1s[j] * bmp_exp(L.Cs[i * ldc + j] - mx);
            L.rmax[i] = mx; L.invD1[i] = 1.f / s;
        }
    }
}

// The same statistics with Q lanes per atom (Q a power of two <= 16, adjacent lanes): lane q walks the other side's
// atoms q, q + Q, ...; maxima and sums meet through shuffles inside the lane group.  A pair of two 28-atom molecules keeps
// 448 of 512 threads busy instead of 56 (one thread per atom walked 28 exponentials twice, the rest waited at the barrier).
__device__ __forceinline__ float grp_max(float v, int Q) {
    for (int m = Q >> 1; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float grp_sum(float v, int Q) {
    for (int m = Q >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int co_split(int n1, int n2, int nt) {
    int Q = 1;
    while (Q < 16 && 2 * Q * (n1 + n2) <= nt) Q *= 2;
    return Q;
}
__device__ __forceinline__ void co_stats_split(const CoLds& L, int n1, int n2, int ldc, int Q) {
    const int a = threadIdx.x / Q, q = threadIdx.x % Q;
    if (a < n1) {
        const int j = a;
        float mx = -INFINITY;
        for (int i = q; i < n2; i += Q) if (L.w2s[i] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
        mx = grp_max(mx, Q);
        float s = 0.f;
        for (int i = q; i < n2; i += Q) if (L.w2s[i] > 0.f) s += L.w2s[i] * bmp_exp(L.Cs[i * ldc + j] - mx);
        s = grp_sum(s, Q);
        if (q == 0) { L.cmax[j] = mx; L.invD2[j] = 1.f / s; }
    } else if (a < n1 + n2) {
        const int i = a - n1;
        float mx = -INFINITY;
        for (int j = q; j < n1; j += Q) if (L.w1s[j] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
        mx = grp_max(mx, Q);
        float s = 0.f;
        for (int j = q; j < n1; j += Q) if (L.w1s[j] > 0.f) s += L.w1s[j] * bmp_exp(L.Cs[i * ldc + j] - mx);
        s = grp_sum(s, Q);
        if (q == 0) { L.rmax[i] = mx; L.invD1[i] = 1.f / s; }
    }
}

// one thread per atom, atoms tid, tid + nt, ... (the oversized class)
__device__ __forceinline__ void co_stats_split_big(const CoLds& L, int n1, int n2, int ldc, int nt) {
    for (int a = threadIdx.x; a < n1 + n2; a += nt) {
        if (a < n1) {
            const int j = a;
            float mx = -INFINITY;
            for (int i = 0; i < n2; ++i) if (L.w2s[i] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
            float s = 0.f;
            for (int i = 0; i < n2; ++i) if (L.w2s[i] > 0.f) s += L.w2s[i] * bmp_exp(L.Cs[i * ldc + j] - mx);
            L.cmax[j] = mx; L.invD2[j] = 1.f / s;
        } else {
            const int i = a - n1;
            float mx = -INFINITY;
            for (int j = 0; j < n1; ++j) if (L.w1s[j] > 0.f) mx = fmaxf(mx, L.Cs[i * ldc + j]);
            float s = 0.f;
            for (int j = 0; j < n1; ++j) if (L.w1s[j] > 0.f) s += L.w1s[j] * bmp_exp(L.Cs[i * ldc + j] - mx);
            L.rmax[i] = mx; L.invD1[i] = 1.f / s;
        }
    }
}

// L2[i,j] (softmax over i) and L1[j,i] (softmax over j); zero-weight rows get weight 0 in every sum,
// so their (possibly huge) exponent is never used.
__device__ __forceinline__ float co_L2(const CoLds& L, int i, int j, int ldc) {
    return L.w2s[i] > 0.f ? bmp_exp(L.Cs[i * ldc + j] - L.cmax[j]) * L.invD2[j] : 0.f;
}
__device__ __forceinline__ float co_L1(const CoLds& L, int i, int j, int ldc) {
    return L.w1s[j] > 0.f ? bmp_exp(L.Cs[i * ldc + j] - L.rmax[i]) * L.invD1[i] : 0.f;
}

// BIG: the class of pairs with a molecule of more than CO_MAXN rows (more than one tile; the reference's preprocessor has
// no size limit, train_ddi_modify.py:256).  Same program; the arrays the other classes keep in LDS live in the
// workgroup's slice of a global workspace (a.big_ws), and every "one thread per atom" phase is a strided loop.
template <int HT, int NT, bool BIG = false>
__global__ __launch_bounds__(NT) void k_coattn_fwd(CoArgs a) {
    constexpr int NW = NT / 64;
    constexpr int HN = HT > 0 ? HT : CO_MAXH;       // head count known at compile time (8, 4) or runtime (<16)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const int b = a.order[a.order_off + blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int r1 = a.r1[b], n1 = a.n1[b], r2 = a.r2[b], n2 = a.n2[b];
    const int d = a.d, o = a.o, H = HT > 0 ? HT : a.H, ZC = a.ZC, ldc = a.ldc;
    const int nb1 = (n1 + 31) >> 5, nb2 = (n2 + 31) >> 5;
    const CoLds L = BIG ? co_carve(a.big_ws + (size_t)blockIdx.x * a.big_stride, a.np, ldc, H, false, 0, NT, true)
                        : co_carve(lds_raw, a.np, ldc, H, false);
    const float cb = a.cbias[0];

    for (int idx = tid; idx < n1 * H; idx += NT) L.P1s[idx] = a.Z1[(size_t)(r1 + idx / H) * ZC + o + idx % H];
    for (int idx = tid; idx < n2 * H; idx += NT) L.P2s[idx] = a.Z2[(size_t)(r2 + idx / H) * ZC + o + idx % H];
    if constexpr (BIG) {
        for (int j = tid; j < n1; j += NT) L.w1s[j] = a.w1[r1 + j];
        for (int i = tid; i < n2; i += NT) { L.w2s[i] = a.w2[r2 + i]; L.v2s[i] = a.Z2[(size_t)(r2 + i) * ZC + o + H]; }
    } else {
    if (tid < n1) L.w1s[tid] = a.w1[r1 + tid];
    if (tid >= CO_MAXN && tid - CO_MAXN < n2) {
        const int i = tid - CO_MAXN;
        L.w2s[i] = a.w2[r2 + i];
        L.v2s[i] = a.Z2[(size_t)(r2 + i) * ZC + o + H];
    }
    }
    __syncthreads();

    // ---- energy tiles: S = Q2 . X1^T on the matrix cores, one 32x32 block per wave at a time ----
    for (int blk = wave; blk < nb2 * nb1; blk += NW) {
        const int bi = blk / nb1, bj = blk % nb1;
        int ia = bi * 32 + l31; ia = ia < n2 ? ia : n2 - 1;
        int ja = bj * 32 + l31; ja = ja < n1 ? ja : n1 - 1;
        const float* qa = a.Q2 + (size_t)(r2 + ia) * d + 4 * hi;
        const float* xb = a.X1 + (size_t)(r1 + ja) * d + 4 * hi;
        f32x16 acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll 8
        for (int k0 = 0; k0 < d; k0 += 8) {          // unrolled: the 16-byte operand loads of 8 k-steps are in flight together
            const f32x4 av = *(const f32x4*)(qa + k0);
            const f32x4 bv = *(const f32x4*)(xb + k0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = bmp_mfma(av[t], bv[t], acc);
        }
        const int j = bj * 32 + l31;
        const float v1j = j < n1 ? a.Z1[(size_t)(r1 + j) * ZC + o + H] : 0.f;
        float* cg = a.Cbuf + a.coff[b];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = bi * 32 + bmp_acc_row(reg, lane);
            if (i < n2 && j < n1) {
                const float cv = bmp_act(a.act, acc[reg] + v1j + L.v2s[i] + cb);
                L.Cs[i * ldc + j] = cv;
                if (a.Cbuf) cg[(size_t)i * n1 + j] = cv;          // (Cbuf == NULL: forward-only evaluation, nothing kept)
            }
        }
    }
    __syncthreads();
    if (a.mode == 1) {
        // PoolingFineCoattention (PoolingFineCoattention.py:40-51): the atom scores are the means of the energy
        // over the OTHER side's padded positions: e1[j] = sum_i w2_i C[i,j] / A2, e2[i] = sum_j w1_j C[i,j] / A1
        for (int t_ = tid; t_ < (BIG ? n1 + n2 : NT); t_ += NT) {
        const bool s1 = BIG ? t_ < n1 : t_ < CO_MAXN;
        if (s1) {
            const int j = t_;
            if (j < n1) {
                float s = 0.f, A2 = 0.f;
                for (int i = 0; i < n2; ++i) { s += L.w2s[i] * L.Cs[i * ldc + j]; A2 += L.w2s[i]; }
                L.s1[j] = s / A2;
            }
        } else {
            const int i = t_ - (BIG ? n1 : CO_MAXN);
            if (i < n2) {
                float s = 0.f, A1 = 0.f;
                for (int j = 0; j < n1; ++j) { s += L.w1s[j] * L.Cs[i * ldc + j]; A1 += L.w1s[j]; }
                L.s2[i] = s / A1;
            }
        }
        }
    } else {
    const int Q = BIG ? 1 : co_split(n1, n2, NT);
    if constexpr (BIG) co_stats_split_big(L, n1, n2, ldc, NT);
    else co_stats_split(L, n1, n2, ldc, Q);
    __syncthreads();
    if (a.Cbuf) {   // the backward reloads these instead of walking C again (17 % of its time): kept behind the pair's C block
        float* st = a.Cbuf + a.coff[b] + (size_t)n2 * n1;
        if constexpr (BIG) {
            for (int j = tid; j < n1; j += NT) { st[j] = L.cmax[j]; st[n1 + j] = L.invD2[j]; }
            for (int i = tid; i < n2; i += NT) { st[2 * n1 + i] = L.rmax[i]; st[2 * n1 + n2 + i] = L.invD1[i]; }
        } else {
        if (tid < CO_MAXN) { if (tid < n1) { st[tid] = L.cmax[tid]; st[n1 + tid] = L.invD2[tid]; } }
        else if (tid < 2 * CO_MAXN) { const int i = tid - CO_MAXN; if (i < n2) { st[2 * n1 + i] = L.rmax[i]; st[2 * n1 + n2 + i] = L.invD1[i]; } }
        }
    }
    __syncthreads();

    // ---- head projections: H1 = tanh(P1 + L1 . P2), H2 = tanh(P2 + L2 . P1)  (:352-362), Q lanes per atom ----
    for (int a_ = tid / Q; a_ < (BIG ? n1 + n2 : tid / Q + 1); a_ += NT) {      // BIG: Q = 1, atoms tid, tid + NT, ...
        const int q = tid % Q;
        const bool side1 = a_ < n1, valid = a_ < n1 + n2;
        const int me = side1 ? a_ : a_ - n1;
        const int no = side1 ? n2 : n1;
        const float* Po = side1 ? L.P2s : L.P1s;
        const float* wo = side1 ? L.w2s : L.w1s;
        float acc[HN];
#pragma unroll
        for (int h = 0; h < HN; ++h) acc[h] = 0.f;
        if (valid) {
            for (int k = q; k < no; k += Q) {
                const float e = wo[k] * (side1 ? co_L1(L, k, me, ldc) : co_L2(L, me, k, ldc));
#pragma unroll
                for (int h = 0; h < HN; ++h) if (h < H) acc[h] += e * Po[k * H + h];
            }
        }
#pragma unroll
        for (int h = 0; h < HN; ++h) if (h < H) acc[h] = grp_sum(acc[h], Q);
        if (valid && q == 0) {
            const float* Pm = side1 ? L.P1s : L.P2s;
            const float* wa = side1 ? a.wa1 : a.wa2;
            float* Hg = a.H1 == nullptr ? nullptr : (side1 ? a.H1 + (size_t)(r1 + me) * H : a.H2 + (size_t)(r2 + me) * H);
            float sc = 0.f;
#pragma unroll
            for (int h = 0; h < HN; ++h) if (h < H) {
                const float hv = bmp_tanh(Pm[me * H + h] + acc[h]);
                if (Hg) Hg[h] = hv;
                sc += hv * wa[h];
            }
            (side1 ? L.s1 : L.s2)[me] = sc;
        }
    }
    }   // mode
    __syncthreads();

    // ---- atom softmax (default axis=1 = atoms, :364-366), wave 0: side 1, wave 1: side 2 ----
    if (wave < 2) {
        float* sc = wave == 0 ? L.s1 : L.s2;
        const float* ww = wave == 0 ? L.w1s : L.w2s;
        const int n = wave == 0 ? n1 : n2;
        float mx = -INFINITY;
        for (int k = lane; k < n; k += 64) if (ww[k] > 0.f) mx = fmaxf(mx, sc[k]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int k = lane; k < n; k += 64) if (ww[k] > 0.f) s += ww[k] * bmp_exp(sc[k] - mx);
        s = wave_sum(s);
        float* alg = a.al1 == nullptr ? nullptr : (wave == 0 ? a.al1 + r1 : a.al2 + r2);
        for (int k = lane; k < n; k += 64) {
            const float al = ww[k] > 0.f ? bmp_exp(sc[k] - mx) / s : 0.f;
            sc[k] = al;
            if (alg) alg[k] = al;
        }
    }
    __syncthreads();

    // ---- pooled outputs: compact_k = sum_atoms w * alpha * j_layer(atoms)  (:368-369) ----
    // thread (k group, side, column): the NT / 256 groups split the atoms, partial sums meet in LDS (C is dead by now)
    {
        constexpr int KG = NT / 256;
        const int kg = tid >> 8, side = (tid >> 7) & 1, t = tid & 127;
        const float* Z = side == 0 ? a.Z1 + (size_t)r1 * ZC : a.Z2 + (size_t)r2 * ZC;
        const float* al = side == 0 ? L.s1 : L.s2;
        const float* ww = side == 0 ? L.w1s : L.w2s;
        const int n = side == 0 ? n1 : n2;
        const int chunk = (n + KG - 1) / KG, k_lo = kg * chunk, k_hi = (k_lo + chunk) < n ? (k_lo + chunk) : n;
        float* out = (side == 0 ? a.out1 : a.out2) + (size_t)b * o;
        float* part = L.Cs;                     // [KG][2][128] per pass over the columns
        for (int c0 = 0; c0 < o; c0 += 128) {
            const int c = c0 + t;
            float acc = 0.f;
            if (c < o) {
#pragma unroll 8
                for (int k = k_lo; k < k_hi; ++k) acc += ww[k] * al[k] * Z[(size_t)k * ZC + c];
            }
            if (KG == 1) {
                if (c < o) out[c] = acc;
            } else {
                __syncthreads();
                part[(kg * 2 + side) * 128 + t] = acc;
                __syncthreads();
                if (kg == 0 && c < o) {
                    float v = part[side * 128 + t];
#pragma unroll
                    for (int g = 1; g < KG; ++g) v += part[(g * 2 + side) * 128 + t];
                    out[c] = v;
                }
            }
        }
    }
}

// NT threads per pair: the phases are strided loops, (row, chunk) decompositions and MFMA blocks per wave, so a bigger
// pair takes more waves (a launch of a size class lasts about one workgroup's latency; a 96-row pair took 120 us with
// 256 threads).  The per-row phases (one thread per atom of either side) use the first 2 * CO_MAXN threads.
template <int HT, int NT, bool BIG = false>
__global__ __launch_bounds__(NT) void k_coattn_bwd(CoArgs a) {
    constexpr int NW = NT / 64;
    constexpr int HN = HT > 0 ? HT : CO_MAXH;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const int b = a.order[a.order_off + blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int r1 = a.r1[b], n1 = a.n1[b], r2 = a.r2[b], n2 = a.n2[b];
    const int d = a.d, o = a.o, H = HT > 0 ? HT : a.H, ZC = a.ZC, ldc = a.ldc;
    const int nb1 = (n1 + 31) >> 5, nb2 = (n2 + 31) >> 5;
    const int n1p = nb1 * 32, n2p = nb2 * 32;
    const CoLds L = BIG ? co_carve(a.big_ws + (size_t)blockIdx.x * a.big_stride, a.np, ldc, H, true, o, NT, true)
                        : co_carve(lds_raw, a.np, ldc, H, true, o, NT);
    // ---- rows of no molecule must read as zero in the GEMMs that follow the pair kernels: the dead rows of a tile sit behind
    // its last molecule, so the pair that owns that molecule clears them (a launch of its own for this was 23 us of the chain)
    if (a.rm1 != nullptr) {
        for (int row = r1 + n1; row < a.N1 && a.rm1[row] < 0; ++row) {
            for (int c = tid; c < d; c += NT) a.dX1[(size_t)row * d + c] = 0.f;
            for (int c = tid; c < ZC; c += NT) a.dZ1[(size_t)row * ZC + c] = 0.f;
        }
        for (int row = r2 + n2; row < a.N2 && a.rm2[row] < 0; ++row) {
            for (int c = tid; c < d; c += NT) a.dQ2[(size_t)row * d + c] = 0.f;
            for (int c = tid; c < ZC; c += NT) a.dZ2[(size_t)row * ZC + c] = 0.f;
        }
    }
    // ---- load the pair's saved state ----
    {   // column / row softmax statistics of C: saved by the forward behind the pair's C block
        const float* st = a.Cbuf + a.coff[b] + (size_t)n2 * n1;
        if constexpr (BIG) {
            for (int j = tid; j < n1; j += NT) { L.cmax[j] = st[j]; L.invD2[j] = st[n1 + j]; }
            for (int i = tid; i < n2; i += NT) { L.rmax[i] = st[2 * n1 + i]; L.invD1[i] = st[2 * n1 + n2 + i]; }
        } else {
        if (tid < CO_MAXN) { if (tid < n1) { L.cmax[tid] = st[tid]; L.invD2[tid] = st[n1 + tid]; } }
        else if (tid < 2 * CO_MAXN) { const int i = tid - CO_MAXN; if (i < n2) { L.rmax[i] = st[2 * n1 + i]; L.invD1[i] = st[2 * n1 + n2 + i]; } }
        }
    }
    for (int idx = tid; idx < n1 * H; idx += NT) L.P1s[idx] = a.Z1[(size_t)(r1 + idx / H) * ZC + o + idx % H];
    for (int idx = tid; idx < n2 * H; idx += NT) L.P2s[idx] = a.Z2[(size_t)(r2 + idx / H) * ZC + o + idx % H];
    for (int idx = tid; idx < n1 * H; idx += NT) L.H1s[idx] = a.H1[(size_t)r1 * H + idx];
    for (int idx = tid; idx < n2 * H; idx += NT) L.H2s[idx] = a.H2[(size_t)r2 * H + idx];
    {
        const float gs = a.gscale ? a.gscale[0] : 1.f;
        for (int c = tid; c < o; c += NT) { L.do1[c] = a.dout1[(size_t)b * o + c] * gs; L.do2[c] = a.dout2[(size_t)b * o + c] * gs; }
    }
    if constexpr (BIG) {
        for (int j = tid; j < n1; j += NT) { L.w1s[j] = a.w1[r1 + j]; L.s1[j] = a.al1[r1 + j]; }
        for (int i = tid; i < n2; i += NT) { L.w2s[i] = a.w2[r2 + i]; L.s2[i] = a.al2[r2 + i]; }
    } else {
    if (tid < n1) { L.w1s[tid] = a.w1[r1 + tid]; L.s1[tid] = a.al1[r1 + tid]; }
    if (tid >= CO_MAXN && tid - CO_MAXN < n2) {
        const int i = tid - CO_MAXN;
        L.w2s[i] = a.w2[r2 + i]; L.s2[i] = a.al2[r2 + i];
    }
    }
    {
        const float* cg = a.Cbuf + a.coff[b];
        for (int idx = tid; idx < n2p * ldc; idx += NT) { L.dSs[idx] = 0.f; }
        for (int idx = tid; idx < n2 * n1; idx += NT) L.Cs[(idx / n1) * ldc + idx % n1] = cg[idx];
    }
    // B operands of this wave's first energy-backward block (rows of X1 or Q2 -- inputs, nothing of this kernel): asked
    // for behind the loads of the saved state, used at the very end; the block otherwise opens with a global round trip while the matrix pipe waits
    const int ncb = (d + 31) >> 5;
    float pre[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) pre[t] = 0.f;
    if (wave < (nb2 + nb1) * ncb) {
        const bool isq = wave < nb2 * ncb;
        const int bc = (isq ? wave : wave - nb2 * ncb) % ncb;
        const int col = bc * 32 + l31;
        const int colc = col < d ? col : d - 1;
        const int kn = isq ? n1 : n2;
        const float* src = isq ? a.X1 + (size_t)r1 * d : a.Q2 + (size_t)r2 * d;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = ks * 8 + 4 * hi + t;
                pre[ks * 4 + t] = src[(size_t)(k < kn ? k : kn - 1) * d + colc];
            }
    }

    __syncthreads();

    // ---- pooled-output backward: dJ = w*alpha*dout ; dot[k] = J[k,:] . dout ----
    {   // each wave takes CO_DQ rows (rbase + q NW) together: that many independent row loads in flight
        const int nrows = n1 + n2;
        for (int rbase = wave; rbase < nrows; rbase += CO_DQ * NW) {
            float dot[CO_DQ];
#pragma unroll
            for (int q = 0; q < CO_DQ; ++q) dot[q] = 0.f;
            for (int c = lane; c < o; c += 64) {
                float zq[CO_DQ];
#pragma unroll
                for (int q = 0; q < CO_DQ; ++q) {
                    const int row = rbase + NW * q;
                    const size_t gr = row < n1 ? (size_t)(r1 + row) : (size_t)(r2 + row - n1);
                    zq[q] = row < nrows ? (row < n1 ? a.Z1 : a.Z2)[gr * ZC + c] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < CO_DQ; ++q) {
                    const int row = rbase + NW * q;
                    if (row < nrows) {
                        const bool s1 = row < n1;
                        const int k = s1 ? row : row - n1;
                        const float g = (s1 ? L.do1 : L.do2)[c];
                        const float wa = s1 ? L.w1s[k] * L.s1[k] : L.w2s[k] * L.s2[k];
                        dot[q] += zq[q] * g;
                        (s1 ? a.dZ1 : a.dZ2)[(size_t)(s1 ? r1 + k : r2 + k) * ZC + c] = wa * g;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < CO_DQ; ++q) {
                const int row = rbase + NW * q;
                const float dsum = wave_sum(dot[q]);
                if (lane == 0 && row < nrows) (row < n1 ? L.dots1 : L.dots2)[row < n1 ? row : row - n1] = dsum;
            }
        }
    }
    __syncthreads();

    // ---- atom softmax backward: ds_k = alpha_k (dalpha_k - w_k T), dalpha_k = w_k dot_k, T = sum alpha dalpha ----
    if (wave < 2) {
        const float* al = wave == 0 ? L.s1 : L.s2;
        const float* ww = wave == 0 ? L.w1s : L.w2s;
        float* dt = wave == 0 ? L.dots1 : L.dots2;
        const int n = wave == 0 ? n1 : n2;
        float tsum = 0.f;
        for (int k = lane; k < n; k += 64) tsum += al[k] * ww[k] * dt[k];
        tsum = wave_sum(tsum);
        for (int k = lane; k < n; k += 64) dt[k] = al[k] * (ww[k] * dt[k] - ww[k] * tsum);     // = ds_k
    }
    __syncthreads();
    if (a.mode == 1) {
        // Pooling: the scores are weighted means of C, so dC[i,j] = w2_i/A2 * ds1[j] + w1_j/A1 * ds2[i]
        float A1 = 0.f, A2 = 0.f;
        for (int j = 0; j < n1; ++j) A1 += L.w1s[j];
        for (int i = 0; i < n2; ++i) A2 += L.w2s[i];
        for (int idx = tid; idx < n2 * n1; idx += NT) {
            const int i = idx / n1, j = idx % n1;
            const float dc = L.w2s[i] / A2 * L.dots1[j] + L.w1s[j] / A1 * L.dots2[i];
            L.dSs[i * ldc + j] = dc * bmp_dact(a.act, L.Cs[i * ldc + j]);
        }
        if (tid < 2 * H) a.dpart[(size_t)b * (2 * H + 1) + tid] = 0.f;
    } else {
    // dHpre[k,h] = ds_k * wa[h] * (1 - H^2) ; per-pair partial of dwa[h] = sum_k ds_k H[k,h]
    for (int idx = tid; idx < (n1 + n2) * H; idx += NT) {
        const int row = idx / H, h = idx % H;
        const int side = row < n1 ? 0 : 1;
        const int k = side == 0 ? row : row - n1;
        const float hv = (side == 0 ? L.H1s : L.H2s)[k * H + h];
        const float ds = (side == 0 ? L.dots1 : L.dots2)[k];
        const float wa = side == 0 ? a.wa1[h] : a.wa2[h];
        (side == 0 ? L.dH1s : L.dH2s)[k * H + h] = ds * wa * (1.f - hv * hv);
    }
    if (tid < 2 * H) {
        const int side = tid / H, h = tid % H;
        const int n = side == 0 ? n1 : n2;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) {
            acc += (side == 0 ? L.dots1 : L.dots2)[k] * (side == 0 ? L.H1s : L.H2s)[k * H + h];
        }
        a.dpart[(size_t)b * (2 * H + 1) + side * H + h] = acc;
    }
    __syncthreads();

    // ---- softmax-of-C backward, all NT threads: thread (row, q) owns the q-th chunk of the other side's atoms.
    //      L1[j,i] = softmax over j (row i fixed):  dC1[i,j] = L1*(w2_i*g1[i,j] - w1_j*U1_i),  g1 = dH1[j,:].P2[i,:],
    //      U1_i = sum_j L1*w2_i*g1 ;  dP2[i,:] = dH2[i,:] + w2_i * sum_j L1 * dH1[j,:]      (and symmetrically L2).
    //      Partial sums go to per-thread LDS slots and are combined in a fixed order (reproducible). ----
    const int np_ = a.np;
    const int SUB = BIG ? 1 : NT / np_;                  // BIG: one slot per row, rows tid, tid + NT, ...
    const int pq = BIG ? 0 : tid / np_;
    const int prow0 = BIG ? tid : tid % np_, prow_step = BIG ? NT : np_ * (SUB > 0 ? SUB : 1) + NT;      // non-BIG: one pass
#define CO_SLOT(row) (BIG ? (row) : tid)
    for (int prow = prow0; prow < (BIG ? n2 : np_); prow += prow_step) {   // L1 path, pass 1
        float U = 0.f, dp[HN];
#pragma unroll
        for (int h = 0; h < HN; ++h) dp[h] = 0.f;
        if (pq < SUB && prow < n2) {
            const int i = prow;
            const int chunk = (n1 + SUB - 1) / SUB, j0 = pq * chunk, j1 = (j0 + chunk) < n1 ? (j0 + chunk) : n1;
            float p2[HN];
#pragma unroll
            for (int h = 0; h < HN; ++h) p2[h] = h < H ? L.P2s[i * H + h] : 0.f;
            const float w2i = L.w2s[i];
            for (int j = j0; j < j1; ++j) {
                const float l1 = co_L1(L, i, j, ldc);
                float g = 0.f;
#pragma unroll
                for (int h = 0; h < HN; ++h) if (h < H) {
                    const float dh = L.dH1s[j * H + h];
                    g += dh * p2[h];
                    dp[h] += w2i * l1 * dh;
                }
                U += l1 * (w2i * g);
            }
        }
        L.Up[CO_SLOT(prow)] = U;
#pragma unroll
        for (int h = 0; h < HN; ++h) if (h < H) L.dPp[CO_SLOT(prow) * H + h] = dp[h];
    }
    __syncthreads();
    for (int i = tid; i < n2; i += NT) {          // U1_i in a fixed order
        float u = 0.f;
        for (int q = 0; q < SUB; ++q) u += L.Up[q * np_ + i];
        L.cmb[i] = u;
    }
    for (int idx = tid; idx < n2 * H; idx += NT) {       // dP2 -> dZ2
        const int i = idx / H, h = idx % H;
        float v = L.dH2s[idx];
        for (int q = 0; q < SUB; ++q) v += L.dPp[(q * np_ + i) * H + h];
        a.dZ2[(size_t)(r2 + i) * ZC + o + h] = v;
    }
    __syncthreads();
    for (int prow = prow0; prow < (BIG ? n2 : np_); prow += prow_step)
    if (pq < SUB && prow < n2) {   // L1 path, pass 2: dS <- dC1
        const int i = prow;
        const int chunk = (n1 + SUB - 1) / SUB, j0 = pq * chunk, j1 = (j0 + chunk) < n1 ? (j0 + chunk) : n1;
        float p2[HN];
#pragma unroll
        for (int h = 0; h < HN; ++h) p2[h] = h < H ? L.P2s[i * H + h] : 0.f;
        const float w2i = L.w2s[i], U = L.cmb[i];
        for (int j = j0; j < j1; ++j) {
            const float l1 = co_L1(L, i, j, ldc);
            float g = 0.f;
#pragma unroll
            for (int h = 0; h < HN; ++h) if (h < H) g += L.dH1s[j * H + h] * p2[h];
            L.dSs[i * ldc + j] = l1 * (w2i * g) - L.w1s[j] * l1 * U;
        }
    }
    __syncthreads();
    for (int prow = prow0; prow < (BIG ? n1 : np_); prow += prow_step) {   // L2 path, pass 1
        float U = 0.f, dp[HN];
#pragma unroll
        for (int h = 0; h < HN; ++h) dp[h] = 0.f;
        if (pq < SUB && prow < n1) {
            const int j = prow;
            const int chunk = (n2 + SUB - 1) / SUB, i0 = pq * chunk, i1 = (i0 + chunk) < n2 ? (i0 + chunk) : n2;
            float p1[HN];
#pragma unroll
            for (int h = 0; h < HN; ++h) p1[h] = h < H ? L.P1s[j * H + h] : 0.f;
            const float w1j = L.w1s[j];
            for (int i = i0; i < i1; ++i) {
                const float l2 = co_L2(L, i, j, ldc);
                float g = 0.f;
#pragma unroll
                for (int h = 0; h < HN; ++h) if (h < H) {
                    const float dh = L.dH2s[i * H + h];
                    g += dh * p1[h];
                    dp[h] += w1j * l2 * dh;
                }
                U += l2 * (w1j * g);
            }
        }
        L.Up[CO_SLOT(prow)] = U;
#pragma unroll
        for (int h = 0; h < HN; ++h) if (h < H) L.dPp[CO_SLOT(prow) * H + h] = dp[h];
    }
    __syncthreads();
    for (int j = tid; j < n1; j += NT) {
        float u = 0.f;
        for (int q = 0; q < SUB; ++q) u += L.Up[q * np_ + j];
        L.cmb[j] = u;
    }
    for (int idx = tid; idx < n1 * H; idx += NT) {       // dP1 -> dZ1
        const int j = idx / H, h = idx % H;
        float v = L.dH1s[idx];
        for (int q = 0; q < SUB; ++q) v += L.dPp[(q * np_ + j) * H + h];
        a.dZ1[(size_t)(r1 + j) * ZC + o + h] = v;
    }
    __syncthreads();
    for (int prow = prow0; prow < (BIG ? n1 : np_); prow += prow_step)
    if (pq < SUB && prow < n1) {   // L2 path, pass 2: dS <- (dC1 + dC2) * act'(C)
        const int j = prow;
        const int chunk = (n2 + SUB - 1) / SUB, i0 = pq * chunk, i1 = (i0 + chunk) < n2 ? (i0 + chunk) : n2;
        float p1[HN];
#pragma unroll
        for (int h = 0; h < HN; ++h) p1[h] = h < H ? L.P1s[j * H + h] : 0.f;
        const float w1j = L.w1s[j], U = L.cmb[j];
        for (int i = i0; i < i1; ++i) {
            const float l2 = co_L2(L, i, j, ldc);
            float g = 0.f;
#pragma unroll
            for (int h = 0; h < HN; ++h) if (h < H) g += L.dH2s[i * H + h] * p1[h];
            const float dc = L.dSs[i * ldc + j] + l2 * (w1j * g) - L.w2s[i] * l2 * U;
            L.dSs[i * ldc + j] = dc * bmp_dact(a.act, L.Cs[i * ldc + j]);
        }
    }
    }   // mode
    __syncthreads();
    // dv1[j] = sum_i dS[i,j] ; dv2[i] = sum_j dS[i,j] ; dc = sum_i dv2[i]
    if constexpr (BIG) {
        for (int j = tid; j < n1; j += NT) {
            float dv1 = 0.f;
            for (int i = 0; i < n2; ++i) dv1 += L.dSs[i * ldc + j];
            { float* dzr = a.dZ1 + (size_t)(r1 + j) * ZC; dzr[o + H] = dv1;
              for (int c = o + H + 1; c < ZC; ++c) dzr[c] = 0.f; }      // the row's padding columns: the same cache line
        }
        float mine = 0.f;
        for (int i = tid; i < n2; i += NT) {
            float dv2 = 0.f;
            for (int j = 0; j < n1; ++j) dv2 += L.dSs[i * ldc + j];
            { float* dzr = a.dZ2 + (size_t)(r2 + i) * ZC; dzr[o + H] = dv2;
              for (int c = o + H + 1; c < ZC; ++c) dzr[c] = 0.f; }      // the row's padding columns: the same cache line
            mine += dv2;
        }
        const float tot = wave_sum(mine);
        if (lane == 0) L.dots1[wave] = tot;          // dots1 is dead by now; np >= 160 > NW
        __syncthreads();
        if (tid == 0) {
            float dc = 0.f;
            for (int q = 0; q < NW; ++q) dc += L.dots1[q];
            a.dpart[(size_t)b * (2 * H + 1) + 2 * H] = dc;
        }
    } else {
    if (tid < CO_MAXN) {
        const int j = tid;
        if (j < n1) {
            float dv1 = 0.f;
            for (int i = 0; i < n2; ++i) dv1 += L.dSs[i * ldc + j];
            { float* dzr = a.dZ1 + (size_t)(r1 + j) * ZC; dzr[o + H] = dv1;
              for (int c = o + H + 1; c < ZC; ++c) dzr[c] = 0.f; }      // the row's padding columns: the same cache line
        }
    } else {
        const int i = tid - CO_MAXN;
        float dv2 = 0.f;
        if (i < n2) {
            for (int j = 0; j < n1; ++j) dv2 += L.dSs[i * ldc + j];
            { float* dzr = a.dZ2 + (size_t)(r2 + i) * ZC; dzr[o + H] = dv2;
              for (int c = o + H + 1; c < ZC; ++c) dzr[c] = 0.f; }      // the row's padding columns: the same cache line
        }
        const float tot = wave_sum(dv2);
        if (lane == 0 && wave < 4) L.dots1[wave - 2] = tot;
    }
    __syncthreads();
    if (tid == 0) a.dpart[(size_t)b * (2 * H + 1) + 2 * H] = L.dots1[0] + L.dots1[1];
    }

    // ---- energy backward on the matrix cores: dQ2 = dS . X1 ; dX1 = dS^T . Q2 ----
    for (int blk = wave; blk < (nb2 + nb1) * ncb; blk += NW) {
        const bool isq = blk < nb2 * ncb;
        const int bl = isq ? blk : blk - nb2 * ncb;
        const int br = bl / ncb, bc = bl % ncb;
        int col = bc * 32 + l31;
        const int colc = col < d ? col : d - 1;
        const int kn = isq ? n1 : n2;                // reduction length (atoms of the other side)
        const int knp = isq ? n1p : n2p;
        const float* src = isq ? a.X1 + (size_t)r1 * d : a.Q2 + (size_t)r2 * d;
        f32x16 acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.f;
        int k_start = 0;
        if (blk == wave) {                           // the rows requested at the top of the kernel
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = ks * 8 + 4 * hi + t;
                    const float av = isq ? L.dSs[(br * 32 + l31) * ldc + k] : L.dSs[k * ldc + br * 32 + l31];
                    acc = bmp_mfma(av, pre[ks * 4 + t], acc);
                }
            k_start = 32;
        }
#pragma unroll 4
        for (int k0 = k_start; k0 < knp; k0 += 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = k0 + 4 * hi + t;
                const int kc = k < kn ? k : kn - 1;          // padded k: dS is 0 there
                const float av = isq ? L.dSs[(br * 32 + l31) * ldc + k] : L.dSs[k * ldc + br * 32 + l31];
                const float bv = src[(size_t)kc * d + colc];
                acc = bmp_mfma(av, bv, acc);
            }
        }
        float* dst = isq ? a.dQ2 + (size_t)r2 * d : a.dX1 + (size_t)r1 * d;
        const int nrow = isq ? n2 : n1;
        if (col < d) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = br * 32 + bmp_acc_row(reg, lane);
                if (rr < nrow) dst[(size_t)rr * d + col] = acc[reg];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static int co_set_lds(const void* fn, size_t bytes) { return bmp_lds_attr(fn, bytes); }      // once per (kernel, device)

// ZC = o + H + 1 rounded up to a multiple of 8: J (o) | P (H) | v | pad
extern "C" int bmp_coattn_zcols(int o, int H) { return (o + H + 1 + 7) & ~7; }

// Workspace (floats) of the oversized class of the pair kernels: nbig pairs whose largest molecule has np_big rows
// (> 128); backward != 0: the backward's share (part of bmp_coattn_nie_bwd_ws_floats).  0 without such pairs.
extern "C" size_t bmp_coattn_big_ws_floats(int np_big, int H, int o, int nbig, int backward) {
    if (nbig <= 0 || np_big <= 0) return 0;
    return (size_t)nbig * co_big_stride(np_big, H, backward != 0, o);
}

// Forward.  WbT [d x d] = W (bilinear form, [p][q]) so that Q2 = X2 . W^T uses it K-major as [q][p]:
// pass WbT[q*d + p] = W[p][q].  ZW1T/ZW2T [d x ZC] K-major columns [Wj^T | Wl_k^T | V_k | 0]; zb [ZC] = [bj | 0].
// mode bit 1 (value 2): one bias row per side, zb / dzb are [2 x ZC] (the Deep* variants, whose folded projections differ by side).
// Which size classes go to the second stream: every class of at most CO_BESIDE_MAX pairs except the most populated one
// (bit c = class c); 0 without a second stream.  (Measured: the 20-pair 96-row class of the DDI batches beside the others
// -1.3 % on the C2 step, -2.4 % on C3; the 510-pair class there as well: +2 %.)
#define CO_BESIDE_MAX 64
static int co_beside_classes(const int* cnt, hipStream_t st, hipStream_t st_b) {
#ifdef BMP_COATTN_NO_BESIDE
    return 0;
#else
    if (!st_b || st_b == st) return 0;
    int big = -1, beside = 0;
    for (int c = 0; c < 4; ++c)
        if (cnt[c] > 0 && (big < 0 || cnt[c] > cnt[big])) big = c;
    for (int c = 0; c < 4; ++c)
        if (cnt[c] > 0 && c != big && cnt[c] <= CO_BESIDE_MAX) beside |= 1 << c;
    return beside;
#endif
}

extern "C" int bmp_coattn_nie_fwd(const float* X1, int n_tiles1, const float* X2, int n_tiles2, int d, int o, int H, int act, int mode,
                                  const float* w1, const int* r1, const int* n1, const float* w2, const int* r2,
                                  const int* n2, const long long* coff, int B, const int* order, int n32, int n64, int n96,
                                  int n128, int nbig, int np_big, const float* WbT,
                                  const float* ZW1T, const float* ZW2T, const float* zb, const float* wa1,
                                  const float* wa2, const float* cbias, float* Q2, float* Z1, float* Z2, float* Cbuf,
                                  float* H1, float* H2, float* al1, float* al2, float* out1, float* out2, float* ws_big,
                                  size_t ws_big_floats, hipStream_t st) {
    BMP_REQUIRE(d > 0 && (d & 7) == 0 && o > 0 && (o & 3) == 0 && H > 0 && H < CO_MAXH && B > 0);
    BMP_REQUIRE(order != nullptr && n32 >= 0 && n64 >= 0 && n96 >= 0 && n128 >= 0 && nbig >= 0 && n32 + n64 + n96 + n128 + nbig == B);
    BMP_REQUIRE(nbig == 0 || (np_big > CO_MAXN && ws_big && ws_big_floats >= bmp_coattn_big_ws_floats(np_big, H, o, nbig, 0)));
    // forward-only evaluation (predict): Cbuf, H1, H2, al1, al2 -- what the backward reloads -- may be NULL together
    BMP_REQUIRE((Cbuf != nullptr) == (H1 != nullptr) && (Cbuf != nullptr) == (H2 != nullptr) && (Cbuf != nullptr) == (al1 != nullptr) &&
                (Cbuf != nullptr) == (al2 != nullptr));
    const int ZC = bmp_coattn_zcols(o, H);
    int rc;
    {   // Q2 = X2 . WbT ; Z1 = X1 . ZW1T + zb ; Z2 = X2 . ZW2T + zb : three projections, one launch
        RGArgs g[3]; memset(g, 0, sizeof(g));
        g[0].s[0] = RGSrc{X2, nullptr, WbT, d, 0, d, d};
        g[0].nsrc = 1; g[0].Nout = d; g[0].Y = Q2; g[0].ldy = d;
        for (int s = 0; s < 2; ++s) {
            g[1 + s].s[0] = RGSrc{s == 0 ? X1 : X2, nullptr, s == 0 ? ZW1T : ZW2T, d, 0, ZC, d};
            g[1 + s].nsrc = 1; g[1 + s].Nout = ZC; g[1 + s].Y = s == 0 ? Z1 : Z2; g[1 + s].ldy = ZC; g[1 + s].bias = zb + ((mode & 2) && s ? ZC : 0);
        }
        const int nt[3] = {n_tiles2, n_tiles1, n_tiles2};
        if ((rc = bmp_launch_rowgemm_multi(g, nt, 3, st))) return rc;
    }
    const void* kf = H == 8 ? (const void*)k_coattn_fwd<8, CO_NT_FWD> : H == 4 ? (const void*)k_coattn_fwd<4, CO_NT_FWD>
                                                                              : (const void*)k_coattn_fwd<0, CO_NT_FWD>;
    if ((rc = co_set_lds(kf, 160 * 1024))) return rc;
    CoArgs a; memset(&a, 0, sizeof(a));
    a.X1 = X1; a.X2 = X2; a.Q2 = Q2; a.Z1 = Z1; a.Z2 = Z2; a.ZC = ZC; a.w1 = w1; a.w2 = w2;
    a.r1 = r1; a.n1 = n1; a.r2 = r2; a.n2 = n2; a.coff = coff; a.wa1 = wa1; a.wa2 = wa2; a.cbias = cbias;
    a.d = d; a.o = o; a.H = H; a.act = act; a.mode = mode & 1; a.order = order;
    a.Cbuf = Cbuf; a.H1 = H1; a.H2 = H2; a.al1 = al1; a.al2 = al2; a.out1 = out1; a.out2 = out2;
    // one launch per size class: LDS (and so the workgroups per CU) follows the pairs' actual sizes.  (The callers hand
    // the forward ONE class sized by the largest pair: its own class for a handful of big pairs, beside the others on a
    // second stream as the backward does below, measured 0.5 % slower on the C2 / C3 steps.)
    // (`order` lists the classes in DESCENDING size for this entry point: the oversized pairs first)
    int off = 0;
    if (nbig > 0) {       // the oversized class: same program out of a global workspace (rare: a handful of pairs per batch)
        a.np = co_big_np(np_big); a.ldc = a.np + 1; a.order_off = 0;
        a.big_ws = ws_big; a.big_stride = (long long)co_big_stride(np_big, H, false, o);
        BmpProfScope prof(BMP_KCLS_COATTN, 0.0, 0.0, st, BMP_KID_COATTN_FWD);
        if (H == 8) hipLaunchKernelGGL((k_coattn_fwd<8, CO_NT_FWD, true>), dim3(nbig), dim3(CO_NT_FWD), 0, st, a);
        else if (H == 4) hipLaunchKernelGGL((k_coattn_fwd<4, CO_NT_FWD, true>), dim3(nbig), dim3(CO_NT_FWD), 0, st, a);
        else hipLaunchKernelGGL((k_coattn_fwd<0, CO_NT_FWD, true>), dim3(nbig), dim3(CO_NT_FWD), 0, st, a);
        BMP_LAUNCH_CHECK();
        off = nbig;
    }
    const int cnt[4] = {n32, n64, n96, n128};
    for (int c = 3; c >= 0; --c) {
        if (cnt[c] == 0) continue;
        a.np = 32 * (c + 1); a.ldc = a.np + 1; a.order_off = off;
        const size_t lds = co_lds_floats(a.np, a.ldc, H, false) * sizeof(float);
        BMP_REQUIRE(lds <= 160 * 1024);
        BmpProfScope prof(BMP_KCLS_COATTN, 0.0, 0.0, st, BMP_KID_COATTN_FWD);
        if (H == 8) hipLaunchKernelGGL((k_coattn_fwd<8, CO_NT_FWD>), dim3(cnt[c]), dim3(CO_NT_FWD), lds, st, a);
        else if (H == 4) hipLaunchKernelGGL((k_coattn_fwd<4, CO_NT_FWD>), dim3(cnt[c]), dim3(CO_NT_FWD), lds, st, a);
        else hipLaunchKernelGGL((k_coattn_fwd<0, CO_NT_FWD>), dim3(cnt[c]), dim3(CO_NT_FWD), lds, st, a);
        BMP_LAUNCH_CHECK();
        off += cnt[c];
    }
    return 0;
}

extern "C" size_t bmp_coattn_nie_bwd_ws_floats(int n_tiles1, int n_tiles2, int d, int o, int H, int B, int nbig, int np_big) {
    const int ZC = bmp_coattn_zcols(o, H);
    const int N1 = n_tiles1 * BMP_R, N2 = n_tiles2 * BMP_R;
    const int Nm = N1 > N2 ? N1 : N2;
    size_t slab = bmp_wgrad_ws_floats(Nm, d, d);
    size_t s2 = bmp_wgrad_ws_floats(Nm, d, ZC);
    if (s2 > slab) slab = s2;
    size_t s3 = bmp_colsum_ws_floats(Nm, ZC);
    if (s3 > slab) slab = s3;
    size_t s4 = bmp_colsum_ws_floats(B, 2 * H + 1);
    if (s4 > slab) slab = s4;
    {
        const WGArgs g[3] = {WGArgs{nullptr, nullptr, d, 0, nullptr, d, d, d, N2, nullptr, d, 0, nullptr, 0, nullptr},
                             WGArgs{nullptr, nullptr, d, 0, nullptr, ZC, d, ZC, N1, nullptr, ZC, 0, (float*)16, 0, nullptr},
                             WGArgs{nullptr, nullptr, d, 0, nullptr, ZC, d, ZC, N2, nullptr, ZC, 0, (float*)16, 1, nullptr}};
        const size_t s5 = bmp_wgrad_multi_ws_floats(g, 3);
        if (s5 > slab) slab = s5;
    }
    // dQ2 [N2 x d] | dZ1 [N1 x ZC] | dZ2 [N2 x ZC] | dpart [B x (2H+1)] | slab | the oversized class's images
    return (size_t)N2 * d + (size_t)(N1 + N2) * ZC + (size_t)B * (2 * H + 1) + slab + 16 + bmp_coattn_big_ws_floats(np_big, H, o, nbig, 1);
}

// Backward.  Wb [d x d] = W natural ([p][q]) (dX2 += dQ2 . W); ZW1/ZW2 [ZC x d] = transposes of ZW*T.
// Outputs: dX1, dX2 (written), dWbT [d x d], dZW1T/dZW2T [d x ZC], dzb [ZC] (sum of both sides),
// dwa [2H + 1] = dwa1 | dwa2 | dc.
extern "C" int bmp_coattn_nie_bwd(const float* dout1, const float* dout2, const float* X1, int n_tiles1, const float* X2,
                                  int n_tiles2, int d, int o, int H, int act, int mode, const float* w1, const int* r1,
                                  const int* n1, const float* w2, const int* r2, const int* n2, const long long* coff,
                                  int B, const int* order, int n32, int n64, int n96, int n128, int nbig, int np_big, const float* Wb,
                                  const float* ZW1, const float* ZW2, const float* wa1,
                                  const float* wa2, const float* Q2, const float* Z1, const float* Z2, const float* Cbuf,
                                  const float* H1, const float* H2, const float* al1, const float* al2, float* dX1,
                                  float* dX2, float* dWbT, float* dZW1T, float* dZW2T, float* dzb, float* dwa, float* ws,
                                  size_t ws_floats, hipStream_t st, hipStream_t st_w, const int* row_mol1,
                                  const int* row_mol2, const float* gscale) {
    BMP_REQUIRE(d > 0 && (d & 7) == 0 && o > 0 && (o & 3) == 0 && H > 0 && H < CO_MAXH && B > 0);
    BMP_REQUIRE(order != nullptr && n32 >= 0 && n64 >= 0 && n96 >= 0 && n128 >= 0 && nbig >= 0 && n32 + n64 + n96 + n128 + nbig == B);
    BMP_REQUIRE(nbig == 0 || np_big > CO_MAXN);
    BMP_REQUIRE(ws_floats >= bmp_coattn_nie_bwd_ws_floats(n_tiles1, n_tiles2, d, o, H, B, nbig, np_big));
    const int ZC = bmp_coattn_zcols(o, H);
    const int N1 = n_tiles1 * BMP_R, N2 = n_tiles2 * BMP_R;
    float* dQ2 = ws;
    float* dZ1 = dQ2 + (size_t)N2 * d;
    float* dZ2 = dZ1 + (size_t)N1 * ZC;
    float* dpart = dZ2 + (size_t)N2 * ZC;
    float* slab = dpart + (size_t)B * (2 * H + 1);
    hipError_t e;
    // rows outside every pair (dead rows) must read as zero in the GEMMs below: the pair kernels write every row of every
    // molecule (its dZ padding columns included) and, with the row -> molecule maps of the packed batches (-1: no molecule),
    // the dead rows behind a tile's last molecule; without the maps everything is cleared first (125 MB of fills at C2)
    if (!(row_mol1 && row_mol2)) {
        if ((e = hipMemsetAsync(dQ2, 0, ((size_t)N2 * d + (size_t)(N1 + N2) * ZC) * sizeof(float), st)) != hipSuccess) return (int)e;
        if ((e = hipMemsetAsync(dX1, 0, (size_t)N1 * d * sizeof(float), st)) != hipSuccess) return (int)e;
    }
    int rc;
    CoArgs a; memset(&a, 0, sizeof(a));
    a.X1 = X1; a.X2 = X2; a.Q2 = Q2; a.Z1 = Z1; a.Z2 = Z2; a.ZC = ZC; a.w1 = w1; a.w2 = w2;
    a.r1 = r1; a.n1 = n1; a.r2 = r2; a.n2 = n2; a.coff = coff; a.wa1 = wa1; a.wa2 = wa2;
    a.d = d; a.o = o; a.H = H; a.act = act; a.mode = mode & 1; a.order = order;
    a.Cbuf = const_cast<float*>(Cbuf); a.H1 = const_cast<float*>(H1); a.H2 = const_cast<float*>(H2);
    a.al1 = const_cast<float*>(al1); a.al2 = const_cast<float*>(al2);
    a.dout1 = dout1; a.dout2 = dout2; a.dQ2 = dQ2; a.dX1 = dX1; a.dZ1 = dZ1; a.dZ2 = dZ2; a.dpart = dpart;
    if (row_mol1 && row_mol2) { a.rm1 = row_mol1; a.rm2 = row_mol2; a.N1 = N1; a.N2 = N2; }
    a.gscale = gscale;
    if (!st_w) st_w = st;
    {
        const int cnt[4] = {n32, n64, n96, n128};
        // (The 32-row class's pairs inside the 64-row class's launch -- one launch of 995 workgroups at 512 threads instead of
        //  575 at 256 and 420 at 512 one after the other -- measured no faster: C2 400.2 / 398.2 against 401.2 / 400.6 k pairs/s,
        //  C3 821 against 827 k.)
        // A size class of a few pairs is a launch of one workgroup's latency on an all but empty device: with a second stream
        // at hand (stream_w) the smallest such class runs there, beside the others, and `st` picks up behind it below.
        const int beside = co_beside_classes(cnt, st, st_w);       // bit c: class c runs on st_w
        if (beside && (rc = bmp_stream_after(st, st_w))) return rc;       // behind the clearing launch above
        int off = 0;
        for (int c = 0; c < 4; ++c) {
            if (cnt[c] == 0) continue;
            hipStream_t cst = ((beside >> c) & 1) ? st_w : st;
            a.np = 32 * (c + 1); a.ldc = a.np + 1; a.order_off = off;
            // threads per pair by size class: a launch lasts about one workgroup's latency, so bigger pairs get more
            // waves (np = 128 stays at 256: its per-thread partial-sum slots alias arrays sized for 256 threads)
            // (the 32-row class at 512 / 1024 threads per pair: its 575 workgroups then take 0.167 / 0.191 ms per step of the
            //  backward's pair kernels instead of 0.156: that launch is bound by the workgroups' resources, not their latency)
            const int nt = c == 0 ? 256 : (c == 1 ? 512 : (c == 2 ? CO_NT_BIG : 256));
            const size_t lds = co_lds_floats(a.np, a.ldc, H, true, o, nt) * sizeof(float);
            BMP_REQUIRE(lds <= 160 * 1024);
            BmpProfScope prof(BMP_KCLS_COATTN, 0.0, 0.0, cst, BMP_KID_COATTN_BWD);
#define CO_BWD_LAUNCH(HT_, NT_)                                                                              \
            {                                                                                                \
                if ((rc = co_set_lds((const void*)k_coattn_bwd<HT_, NT_>, 160 * 1024))) return rc;          \
                hipLaunchKernelGGL((k_coattn_bwd<HT_, NT_>), dim3(cnt[c]), dim3(NT_), lds, cst, a);          \
            }
#define CO_BWD_BY_NT(HT_)                                                                                    \
            if (nt == 256) CO_BWD_LAUNCH(HT_, 256) else if (nt == 512) CO_BWD_LAUNCH(HT_, 512) else CO_BWD_LAUNCH(HT_, CO_NT_BIG)
            if (H == 8) { CO_BWD_BY_NT(8) } else if (H == 4) { CO_BWD_BY_NT(4) } else { CO_BWD_BY_NT(0) }
#undef CO_BWD_BY_NT
#undef CO_BWD_LAUNCH
            BMP_LAUNCH_CHECK();
            off += cnt[c];
        }
        if (nbig > 0) {   // the oversized class (order: classes ascending, so these pairs come last): out of the workspace's tail
            const size_t big_fl = bmp_coattn_big_ws_floats(np_big, H, o, nbig, 1);
            float* big = ws + (((ws_floats - big_fl) / 4) * 4);
            a.np = co_big_np(np_big); a.ldc = a.np + 1; a.order_off = off;
            a.big_ws = big; a.big_stride = (long long)co_big_stride(np_big, H, true, o);
            BmpProfScope prof(BMP_KCLS_COATTN, 0.0, 0.0, st, BMP_KID_COATTN_BWD);
            if (H == 8) hipLaunchKernelGGL((k_coattn_bwd<8, 512, true>), dim3(nbig), dim3(512), 0, st, a);
            else if (H == 4) hipLaunchKernelGGL((k_coattn_bwd<4, 512, true>), dim3(nbig), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_coattn_bwd<0, 512, true>), dim3(nbig), dim3(512), 0, st, a);
            BMP_LAUNCH_CHECK();
        }
        if (beside && (rc = bmp_stream_after(st_w, st))) return rc;
    }
    if ((rc = bmp_fork_to(st, st_w))) return rc;        // dQ2, dZ1, dZ2, dpart are complete: the weight gradients may start
    {   // dX1 += dZ1 . ZW1 (K = ZC) ; dX2 = dQ2 . W + dZ2 . ZW2 : one launch
        RGArgs g[2]; memset(g, 0, sizeof(g));
        g[0].s[0] = RGSrc{dZ1, nullptr, ZW1, ZC, 0, d, ZC};
        g[0].nsrc = 1; g[0].Nout = d; g[0].Y = dX1; g[0].ldy = d; g[0].accumulate = 1;
        g[1].s[0] = RGSrc{dQ2, nullptr, Wb, d, 0, d, d};
        g[1].s[1] = RGSrc{dZ2, nullptr, ZW2, ZC, 0, d, ZC};
        g[1].nsrc = 2; g[1].Nout = d; g[1].Y = dX2; g[1].ldy = d;
        const int nt[2] = {n_tiles1, n_tiles2};
        if ((rc = bmp_launch_rowgemm_multi(g, nt, 2, st))) return rc;
    }
    {   // the three weight gradients: a few tiles each, one launch.  dzb = column sums of dZ1 and dZ2: they ride along
        // with the two GEMMs that read those arrays anyway (the second reduction accumulates into the first's result)
        const WGArgs g[3] = {WGArgs{X2, nullptr, d, 0, dQ2, d, d, d, N2, dWbT, d, 0, nullptr, 0, nullptr},
                             WGArgs{X1, nullptr, d, 0, dZ1, ZC, d, ZC, N1, dZW1T, ZC, 0, dzb, 0, nullptr},
                             WGArgs{X2, nullptr, d, 0, dZ2, ZC, d, ZC, N2, dZW2T, ZC, 0, dzb + ((mode & 2) ? ZC : 0), (mode & 2) ? 0 : 1, nullptr}};
        if (d <= 128 && d >= 64 && ZC >= 64) {
            if ((rc = bmp_launch_wgrad_multi(g, 3, slab, st_w))) return rc;
        } else {
            for (int p = 0; p < 3; ++p)
                if ((rc = bmp_launch_wgrad(g[p], slab, st_w))) return rc;
        }
    }
    return bmp_launch_colsum(dpart, 2 * H + 1, B, 2 * H + 1, dwa, 0, slab, st_w);
}
