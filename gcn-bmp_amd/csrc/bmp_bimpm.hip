// BiMPM drug-pair matching (models/coattention/bimpm.py:45-199, aggr = F.sum; train_binary.py:253-256) on packed rows.
//
// The reference expands both atom sets to (mb, head, N, d) / (mb, N_1, N_2, d) tensors -- head (= out_dim there) times the
// atom states -- and reduces them again.  Here one workgroup walks a drug pair through the same arithmetic without
// materialising any of them: every quantity is a sum or a maximum over one axis of a product of row vectors.
//   matching function      m(u, v) = u.v / ((|u| + eps)(|v| + eps))           (chainer normalize: eps added to the norm)
//   (1) max-pooling        s_ijk = m(P_k*x_i, P_k*y_j);  m1_ik = max_j s_ijk,  m2_jk = max_i s_ijk            (:132-142)
//   (2) attention          att_ij = m(x_i, y_j)                                                              (:107-120)
//   (3) attentive mean     M2_i = sum_j att_ij y_j / max(sum_j att_ij, 1e-4);  mm1_ik = m(Q_k*x_i, Q_0*M2_i)  (:156-167)
//   (4) attentive max      T2_id = max_j att_ij y_jd;                          mx1_ik = m(R_k*x_i, R_0*T2_i)  (:169-182)
//   mol_1 = sum_i [m1_i | mm1_i | mx1_i]  (3 * head columns), side 2 alike.
// (3) and (4) compare perspective k of the atom with perspective 0 of the attended vector: mp_matching_func keeps column 0
// of its (head x head) product (:76-78); restated as written.  Sums over atoms carry the row multiplicities w of the
// packed layout (the virtual pad row stands for all zero-padded positions, which the reference does not mask); maxima
// run over the rows with w > 0.
// Plain fp32 FMA work (no matrix-core shape: the perspectives scale the feature axis inside every dot product), fixed
// summation orders, no atomics: bitwise reproducible.  The backward recomputes the forward of its pair (same code, same
// order, so the same maxima win) and distributes gradients by owner-computes loops.
#include <string.h>
#include "bmp_common.h"

#define BM_NT 256
#define BM_EPS 1e-5f          // chainer.functions.normalize default
#define BM_DIV_EPS 1e-4f      // div_with_small_value, bimpm.py:122-124

struct BmArgs {
    const float* X1; const float* X2; int d, H, B;
    const float* w1; const float* w2;
    const int* r1; const int* n1; const int* r2; const int* n2;
    const float* P; const float* Q; const float* R;     // [H x d] perspectives: max-pooling, attentive mean, attentive max
    float* scratch; size_t scratch_per_wg;              // per-workgroup scratch (floats), sized for the largest pair
    int maxn;                                           // rows of the largest molecule of the batch
    float* out1; float* out2;                           // [B x 3H]
    // backward
    const float* dout1; const float* dout2;
    float* dX1; float* dX2;                             // [N x d], every row of every pair written
    float* wslab;                                       // [grid x 3 x H x d] per-workgroup weight-gradient sums
    // the global-memory class (round 4): a pair whose two molecules' rows do not fit 160 KB of LDS (more than ~150 rows at
    // d = 128; the reference has no size limit, bimpm.py:17-199 / train_ddi_modify.py:256) stages them in the workgroup's
    // scratch slice instead -- same program, same order of operations
    int stage_global; size_t stage_off;                 // floats into the workgroup's scratch slice
};

// per-pair scratch carve-up (float slots; int arrays share the slots)
struct BmScr {
    float *nP1, *nQ1, *nR1, *nP2, *nQ2, *nR2;           // [n x H] |W_k * row|
    float *nx, *ny;                                     // [n]
    float *att;                                         // [n1 x n2]
    float *D2, *D1;                                     // [n]  sums of att
    float *M2, *M1, *T2, *T1;                           // [n x d]
    int *jT2, *iT1;                                     // [n x d] argmax of the attentive max
    float *nzQ2, *nzR2, *nzQ1, *nzR1;                   // [n] |W_0 * attended vector|
    float *m1, *m2, *mm1, *mm2, *mx1, *mx2;             // [n x H]
    int *j1s, *i2s;                                     // [n x H] argmax of the max-pooling matching
    // backward
    float *datt;                                        // [n1 x n2]
    float *dzQ2, *dzR2, *dzQ1, *dzR1;                   // [n x d] gradients w.r.t. W_0 * attended vector
    float *dD2, *dD1;                                   // [n]
    int *head2, *next2, *head1, *next1;                 // [n x H] lists: maxima of the OTHER side that picked this row
};

__host__ __device__ static inline size_t bm_scratch_floats(int maxn, int d, int H, bool bwd) {
    const size_t n = (size_t)maxn;
    size_t s = 6 * n * H + 2 * n + n * n + 2 * n + 4 * n * d + 2 * n * d + 4 * n + 6 * n * H + 2 * n * H;
    if (bwd) s += n * n + 4 * n * d + 2 * n + 4 * n * H;
    return s + 64;
}

__device__ static inline BmScr bm_carve(float* p, int maxn, int d, int H, bool bwd) {
    BmScr s;
    const size_t n = (size_t)maxn, nH = n * H, nd = n * d;
    s.nP1 = p; p += nH; s.nQ1 = p; p += nH; s.nR1 = p; p += nH; s.nP2 = p; p += nH; s.nQ2 = p; p += nH; s.nR2 = p; p += nH;
    s.nx = p; p += n; s.ny = p; p += n;
    s.att = p; p += n * n;
    s.D2 = p; p += n; s.D1 = p; p += n;
    s.M2 = p; p += nd; s.M1 = p; p += nd; s.T2 = p; p += nd; s.T1 = p; p += nd;
    s.jT2 = (int*)p; p += nd; s.iT1 = (int*)p; p += nd;
    s.nzQ2 = p; p += n; s.nzR2 = p; p += n; s.nzQ1 = p; p += n; s.nzR1 = p; p += n;
    s.m1 = p; p += nH; s.m2 = p; p += nH; s.mm1 = p; p += nH; s.mm2 = p; p += nH; s.mx1 = p; p += nH; s.mx2 = p; p += nH;
    s.j1s = (int*)p; p += nH; s.i2s = (int*)p; p += nH;
    if (bwd) {
        s.datt = p; p += n * n;
        s.dzQ2 = p; p += nd; s.dzR2 = p; p += nd; s.dzQ1 = p; p += nd; s.dzR1 = p; p += nd;
        s.dD2 = p; p += n; s.dD1 = p; p += n;
        s.head2 = (int*)p; p += nH; s.next2 = (int*)p; p += nH; s.head1 = (int*)p; p += nH; s.next1 = (int*)p; p += nH;
    }
    return s;
}

// |W_k * row| for the three perspective sets of one side, and the plain row norms
__device__ static void bm_norms(const float* X, int ldx, int n, int d, int H, const float* P, const float* Q, const float* R,
                                float* nP, float* nQ, float* nR, float* nrow) {
    for (int idx = threadIdx.x; idx < n * H; idx += BM_NT) {
        const int k = idx / n, i = idx % n;               // i fastest: conflict-free LDS rows, one weight row per wave
        const float* x = X + (size_t)i * ldx;
        const float* p = P + (size_t)k * d; const float* q = Q + (size_t)k * d; const float* r = R + (size_t)k * d;
        float sp = 0.f, sq = 0.f, sr = 0.f;
        for (int c = 0; c < d; ++c) {
            const float xv = x[c];
            const float a = p[c] * xv, b = q[c] * xv, e = r[c] * xv;
            sp += a * a; sq += b * b; sr += e * e;
        }
        nP[i * H + k] = sqrtf(sp); nQ[i * H + k] = sqrtf(sq); nR[i * H + k] = sqrtf(sr);
    }
    for (int i = threadIdx.x; i < n; i += BM_NT) {
        const float* x = X + (size_t)i * ldx;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += x[c] * x[c];
        nrow[i] = sqrtf(s);
    }
}

// attended vectors of the rows of side A over side B: M = sum_b wB_b att(a,b) B_b / max(sum_b wB_b att, eps);
// T = max over b (wB_b > 0) of att(a,b) B_b (per feature), with its argmax.  att is read as att[a * sa + b * sb].
__device__ static void bm_attend(const float* Bm, int ldb, int na, int nb, int d, const float* att, int sa, int sb,
                                 const float* wB, float* Dsum, float* M, float* T, int* argT) {
    for (int a = threadIdx.x; a < na; a += BM_NT) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += wB[b] * att[a * sa + b * sb];
        Dsum[a] = s;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < na * d; idx += BM_NT) {
        const int c = idx / na, a = idx % na;
        float s = 0.f, mx = -INFINITY;
        int am = -1;
        for (int b = 0; b < nb; ++b) {
            if (wB[b] <= 0.f) continue;
            const float t = att[a * sa + b * sb] * Bm[(size_t)b * ldb + c];
            s += wB[b] * t;
            if (t > mx) { mx = t; am = b; }
        }
        M[(size_t)a * d + c] = s / fmaxf(Dsum[a], BM_DIV_EPS);
        T[(size_t)a * d + c] = mx;
        argT[(size_t)a * d + c] = am;
    }
}

// mp_matching_func (bimpm.py:50-79): mm[a][k] = m(W_k * A_a, W_0 * V_a); also |W_0 * V_a|
__device__ static void bm_match_vec(const float* A, int lda, int na, int d, int H, const float* W, const float* nWA, const float* V,
                                    float* nz, float* mm) {
    for (int a = threadIdx.x; a < na; a += BM_NT) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) { const float z = W[c] * V[(size_t)a * d + c]; s += z * z; }
        nz[a] = sqrtf(s);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < na * H; idx += BM_NT) {
        const int k = idx / na, a = idx % na;
        const float* x = A + (size_t)a * lda;
        const float* wk = W + (size_t)k * d;
        const float* v = V + (size_t)a * d;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += (wk[c] * x[c]) * (W[c] * v[c]);
        mm[a * H + k] = s / ((nWA[a * H + k] + BM_EPS) * (nz[a] + BM_EPS));
    }
}

// max-pooling matching for the rows of side A against side B: m[a][k] = max_b m(P_k*A_a, P_k*B_b), with its argmax
__device__ static void bm_maxpool(const float* A, int lda, int na, const float* Bm, int ldb, int nb, int d, int H, const float* P,
                                  const float* nPA, const float* nPB, const float* wB, float* m, int* arg) {
    for (int idx = threadIdx.x; idx < na * H; idx += BM_NT) {
        const int k = idx / na, a = idx % na;
        const float* x = A + (size_t)a * lda;
        const float* pk = P + (size_t)k * d;
        const float ia = 1.f / (nPA[a * H + k] + BM_EPS);
        float mx = -INFINITY;
        int am = -1;
        for (int b0 = 0; b0 < nb; b0 += 4) {               // four rows of the other side per pass over the features
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* y0 = Bm + (size_t)b0 * ldb;
            const float* y1 = Bm + (size_t)(b0 + 1 < nb ? b0 + 1 : b0) * ldb;
            const float* y2 = Bm + (size_t)(b0 + 2 < nb ? b0 + 2 : b0) * ldb;
            const float* y3 = Bm + (size_t)(b0 + 3 < nb ? b0 + 3 : b0) * ldb;
            for (int c = 0; c < d; ++c) {
                const float u = pk[c] * pk[c] * x[c];
                s0 += u * y0[c]; s1 += u * y1[c]; s2 += u * y2[c]; s3 += u * y3[c];
            }
            const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = b0 + q;
                if (b < nb && wB[b] > 0.f) {
                    const float t = sv[q] * ia / (nPB[b * H + k] + BM_EPS);
                    if (t > mx) { mx = t; am = b; }
                }
            }
        }
        m[a * H + k] = mx;
        arg[a * H + k] = am;
    }
}

// d m(u, v) / du . gamma, for u = W_k * x, given the raw norm |u|: gamma * [v / (a b) - m u / (a |u|)]   (a = |u| + eps)
__device__ __forceinline__ float bm_dmatch(float gamma, float v, float u, float mval, float nu, float nv) {
    const float a = nu + BM_EPS, b = nv + BM_EPS;
    float g = v / (a * b);
    if (nu > 0.f) g -= mval * u / (a * nu);
    return gamma * g;
}

template <bool BWD>
__global__ __launch_bounds__(BM_NT) void k_bimpm(BmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int d = a.d, H = a.H;
    const BmScr S = bm_carve(a.scratch + (size_t)blockIdx.x * a.scratch_per_wg, a.maxn, d, H, BWD);
    const int LDX = d + 1;
    float* wsl = BWD ? a.wslab + (size_t)blockIdx.x * 3 * H * d : nullptr;
    if (BWD) {
        for (int idx = tid; idx < 3 * H * d; idx += BM_NT) wsl[idx] = 0.f;
    }
    for (int pr = blockIdx.x; pr < a.B; pr += gridDim.x) {
        __syncthreads();
        const int r1 = a.r1[pr], n1 = a.n1[pr], r2 = a.r2[pr], n2 = a.n2[pr];
        // ---- stage both molecules' rows (LDS, odd row stride) and their multiplicities ----
        float* Xs = a.stage_global ? a.scratch + (size_t)blockIdx.x * a.scratch_per_wg + a.stage_off : lds;
        float* Ys = Xs + (size_t)n1 * LDX;
        float* wx = Ys + (size_t)n2 * LDX;
        float* wy = wx + n1;
        for (int idx = tid; idx < n1 * d; idx += BM_NT) Xs[(idx / d) * LDX + idx % d] = a.X1[(size_t)(r1 + idx / d) * d + idx % d];
        for (int idx = tid; idx < n2 * d; idx += BM_NT) Ys[(idx / d) * LDX + idx % d] = a.X2[(size_t)(r2 + idx / d) * d + idx % d];
        for (int i = tid; i < n1; i += BM_NT) wx[i] = a.w1[r1 + i];
        for (int j = tid; j < n2; j += BM_NT) wy[j] = a.w2[r2 + j];
        __syncthreads();
        const float* X = Xs; const float* Y = Ys;

        // ---- forward ----
        bm_norms(X, LDX, n1, d, H, a.P, a.Q, a.R, S.nP1, S.nQ1, S.nR1, S.nx);
        bm_norms(Y, LDX, n2, d, H, a.P, a.Q, a.R, S.nP2, S.nQ2, S.nR2, S.ny);
        __syncthreads();
        for (int idx = tid; idx < n1 * n2; idx += BM_NT) {             // attention (:107-120)
            const int j = idx / n1, i = idx % n1;
            const float* x = X + (size_t)i * LDX; const float* y = Y + (size_t)j * LDX;
            float s = 0.f;
            for (int c = 0; c < d; ++c) s += x[c] * y[c];
            S.att[i * n2 + j] = s / ((S.nx[i] + BM_EPS) * (S.ny[j] + BM_EPS));
        }
        __syncthreads();
        bm_attend(Y, LDX, n1, n2, d, S.att, n2, 1, wy, S.D2, S.M2, S.T2, S.jT2);      // side 1 attends over side 2
        bm_attend(X, LDX, n2, n1, d, S.att, 1, n2, wx, S.D1, S.M1, S.T1, S.iT1);
        __syncthreads();
        bm_match_vec(X, LDX, n1, d, H, a.Q, S.nQ1, S.M2, S.nzQ2, S.mm1);
        bm_match_vec(X, LDX, n1, d, H, a.R, S.nR1, S.T2, S.nzR2, S.mx1);
        bm_match_vec(Y, LDX, n2, d, H, a.Q, S.nQ2, S.M1, S.nzQ1, S.mm2);
        bm_match_vec(Y, LDX, n2, d, H, a.R, S.nR2, S.T1, S.nzR1, S.mx2);
        bm_maxpool(X, LDX, n1, Y, LDX, n2, d, H, a.P, S.nP1, S.nP2, wy, S.m1, S.j1s);
        bm_maxpool(Y, LDX, n2, X, LDX, n1, d, H, a.P, S.nP2, S.nP1, wx, S.m2, S.i2s);
        __syncthreads();
        if (!BWD) {
            for (int idx = tid; idx < 6 * H; idx += BM_NT) {           // mol_k = sum over atoms (aggr = F.sum, :190-192)
                const int side = idx / (3 * H), col = idx % (3 * H), part = col / H, k = col % H;
                const float* src = side == 0 ? (part == 0 ? S.m1 : part == 1 ? S.mm1 : S.mx1) : (part == 0 ? S.m2 : part == 1 ? S.mm2 : S.mx2);
                const float* w = side == 0 ? wx : wy;
                const int n = side == 0 ? n1 : n2;
                float s = 0.f;
                for (int i = 0; i < n; ++i) if (w[i] > 0.f) s += w[i] * src[i * H + k];
                (side == 0 ? a.out1 : a.out2)[(size_t)pr * 3 * H + col] = s;
            }
            continue;
        }

        // ---- backward ----
        const float* g1 = a.dout1 + (size_t)pr * 3 * H;                // d loss / d mol_1 [3H]
        const float* g2 = a.dout2 + (size_t)pr * 3 * H;
        // lists: for every row, the maxima of the OTHER side's max-pooling matching that picked it
        for (int idx = tid; idx < n1 * H; idx += BM_NT) S.head2[idx] = -1;
        for (int idx = tid; idx < n2 * H; idx += BM_NT) S.head1[idx] = -1;
        __syncthreads();
        for (int k = tid; k < H; k += BM_NT) {
            for (int j = n2 - 1; j >= 0; --j) {                         // descending: the lists come out ascending
                const int i = wy[j] > 0.f ? S.i2s[j * H + k] : -1;
                if (i >= 0) { S.next2[j * H + k] = S.head2[i * H + k]; S.head2[i * H + k] = j; }
            }
            for (int i = n1 - 1; i >= 0; --i) {
                const int j = wx[i] > 0.f ? S.j1s[i * H + k] : -1;
                if (j >= 0) { S.next1[i * H + k] = S.head1[j * H + k]; S.head1[j * H + k] = i; }
            }
        }
        // gradients w.r.t. the attended vectors' perspective-0 products z = W_0 * V:  dz = sum_k gamma_k [u_k/(a b) - mm z/(b |z|)]
        for (int side = 0; side < 2; ++side) {
            const int n = side == 0 ? n1 : n2;
            const float* A = side == 0 ? X : Y;
            const float* w = side == 0 ? wx : wy;
            const float* g = side == 0 ? g1 : g2;
            for (int which = 0; which < 2; ++which) {                  // 0: attentive mean (Q), 1: attentive max (R)
                const float* W = which == 0 ? a.Q : a.R;
                const float* V = side == 0 ? (which == 0 ? S.M2 : S.T2) : (which == 0 ? S.M1 : S.T1);
                const float* nWA = side == 0 ? (which == 0 ? S.nQ1 : S.nR1) : (which == 0 ? S.nQ2 : S.nR2);
                const float* nz = side == 0 ? (which == 0 ? S.nzQ2 : S.nzR2) : (which == 0 ? S.nzQ1 : S.nzR1);
                const float* mm = side == 0 ? (which == 0 ? S.mm1 : S.mx1) : (which == 0 ? S.mm2 : S.mx2);
                float* dz = side == 0 ? (which == 0 ? S.dzQ2 : S.dzR2) : (which == 0 ? S.dzQ1 : S.dzR1);
                const float* gk = g + (which + 1) * H;
                for (int idx = tid; idx < n * d; idx += BM_NT) {
                    const int c = idx / n, i = idx % n;
                    float s = 0.f;
                    if (w[i] > 0.f) {
                        const float z = W[c] * V[(size_t)i * d + c], nzv = nz[i], x = A[(size_t)i * LDX + c];
                        for (int k = 0; k < H; ++k)
                            s += bm_dmatch(w[i] * gk[k], W[(size_t)k * d + c] * x, z, mm[i * H + k], nzv, nWA[i * H + k]);
                    }
                    dz[(size_t)i * d + c] = s;
                }
            }
        }
        __syncthreads();
        // attentive mean: M = num / den, den = max(D, eps).  dnum = dM / den (kept in place of dz * W_0), dD = -(dM . M) / den
        for (int side = 0; side < 2; ++side) {
            const int n = side == 0 ? n1 : n2;
            const float* M = side == 0 ? S.M2 : S.M1;
            const float* Dm = side == 0 ? S.D2 : S.D1;
            const float* dz = side == 0 ? S.dzQ2 : S.dzQ1;
            float* dD = side == 0 ? S.dD2 : S.dD1;
            for (int i = tid; i < n; i += BM_NT) {
                float s = 0.f;
                for (int c = 0; c < d; ++c) s += a.Q[c] * dz[(size_t)i * d + c] * M[(size_t)i * d + c];
                dD[i] = Dm[i] > BM_DIV_EPS ? -s / Dm[i] : 0.f;
            }
        }
        __syncthreads();
        // d att_ij: from both attentive means (numerators and denominators) and both attentive maxima
        for (int idx = tid; idx < n1 * n2; idx += BM_NT) {
            const int j = idx / n1, i = idx % n1;
            const float* x = X + (size_t)i * LDX; const float* y = Y + (size_t)j * LDX;
            const float id2 = 1.f / fmaxf(S.D2[i], BM_DIV_EPS), id1 = 1.f / fmaxf(S.D1[j], BM_DIV_EPS);
            float s2 = 0.f, s1 = 0.f, t = 0.f;
            for (int c = 0; c < d; ++c) {
                s2 += a.Q[c] * S.dzQ2[(size_t)i * d + c] * y[c];
                s1 += a.Q[c] * S.dzQ1[(size_t)j * d + c] * x[c];
                if (S.jT2[(size_t)i * d + c] == j) t += a.R[c] * S.dzR2[(size_t)i * d + c] * y[c];
                if (S.iT1[(size_t)j * d + c] == i) t += a.R[c] * S.dzR1[(size_t)j * d + c] * x[c];
            }
            S.datt[i * n2 + j] = wy[j] * (s2 * id2 + S.dD2[i]) + wx[i] * (s1 * id1 + S.dD1[j]) + t;
        }
        __syncthreads();
        // ---- d rows: owner (row, feature) ----
        for (int side = 0; side < 2; ++side) {
            const int n = side == 0 ? n1 : n2, no = side == 0 ? n2 : n1;
            const float* A = side == 0 ? X : Y; const float* O = side == 0 ? Y : X;
            const float* w = side == 0 ? wx : wy; const float* wo = side == 0 ? wy : wx;
            const float* g = side == 0 ? g1 : g2; const float* go = side == 0 ? g2 : g1;
            const float* na_ = side == 0 ? S.nx : S.ny; const float* no_ = side == 0 ? S.ny : S.nx;
            const int sa = side == 0 ? n2 : 1, so = side == 0 ? 1 : n2;      // att[a * sa + o * so]
            const float* Do = side == 0 ? S.D1 : S.D2;                       // denominators of the OTHER side's means
            const float* dzQo = side == 0 ? S.dzQ1 : S.dzQ2; const float* dzRo = side == 0 ? S.dzR1 : S.dzR2;
            const int* argTo = side == 0 ? S.iT1 : S.jT2;
            const float* nPa = side == 0 ? S.nP1 : S.nP2; const float* nPo = side == 0 ? S.nP2 : S.nP1;
            const float* nQa = side == 0 ? S.nQ1 : S.nQ2; const float* nRa = side == 0 ? S.nR1 : S.nR2;
            const float* M = side == 0 ? S.M2 : S.M1; const float* T = side == 0 ? S.T2 : S.T1;
            const float* nzQ = side == 0 ? S.nzQ2 : S.nzQ1; const float* nzR = side == 0 ? S.nzR2 : S.nzR1;
            const float* mm = side == 0 ? S.mm1 : S.mm2; const float* mx = side == 0 ? S.mx1 : S.mx2;
            const float* mp = side == 0 ? S.m1 : S.m2; const int* argp = side == 0 ? S.j1s : S.i2s;
            const float* mpo = side == 0 ? S.m2 : S.m1;
            const int* head = side == 0 ? S.head2 : S.head1; const int* next = side == 0 ? S.next2 : S.next1;
            float* dA = (side == 0 ? a.dX1 + (size_t)r1 * d : a.dX2 + (size_t)r2 * d);
            for (int idx = tid; idx < n * d; idx += BM_NT) {
                const int c = idx / n, i = idx % n;
                const float x = A[(size_t)i * LDX + c];
                float acc = 0.f;
                // attention, the other side's attentive mean / max of THIS row
                for (int o = 0; o < no; ++o) {
                    const float at = S.att[i * sa + o * so];
                    acc += bm_dmatch(S.datt[i * sa + o * so], O[(size_t)o * LDX + c], x, at, na_[i], no_[o]);
                    acc += w[i] * at * a.Q[c] * dzQo[(size_t)o * d + c] / fmaxf(Do[o], BM_DIV_EPS);
                    if (argTo[(size_t)o * d + c] == i) acc += at * a.R[c] * dzRo[(size_t)o * d + c];
                }
                if (w[i] > 0.f) {
                    const float zq = a.Q[c] * M[(size_t)i * d + c], zr = a.R[c] * T[(size_t)i * d + c];
                    for (int k = 0; k < H; ++k) {
                        const float qk = a.Q[(size_t)k * d + c], rk = a.R[(size_t)k * d + c], pk = a.P[(size_t)k * d + c];
                        // this row's own matchings: attentive mean, attentive max, max-pooling
                        acc += qk * bm_dmatch(w[i] * g[H + k], zq, qk * x, mm[i * H + k], nQa[i * H + k], nzQ[i]);
                        acc += rk * bm_dmatch(w[i] * g[2 * H + k], zr, rk * x, mx[i * H + k], nRa[i * H + k], nzR[i]);
                        const int js = argp[i * H + k];
                        if (js >= 0)
                            acc += pk * bm_dmatch(w[i] * g[k], pk * O[(size_t)js * LDX + c], pk * x, mp[i * H + k], nPa[i * H + k],
                                                  nPo[js * H + k]);
                        // maxima of the other side that picked this row
                        for (int o = head[i * H + k]; o >= 0; o = next[o * H + k])
                            acc += pk * bm_dmatch(wo[o] * go[k], pk * O[(size_t)o * LDX + c], pk * x, mpo[o * H + k], nPa[i * H + k],
                                                  nPo[o * H + k]);
                    }
                }
                dA[(size_t)i * d + c] = acc;
            }
        }
        // ---- d perspectives: owner (k, feature), accumulated over this workgroup's pairs ----
        for (int idx = tid; idx < H * d; idx += BM_NT) {
            const int k = idx / d, c = idx % d;
            const float pk = a.P[idx], qk = a.Q[idx], rk = a.R[idx];
            float dp = 0.f, dq = 0.f, dr = 0.f;
            for (int side = 0; side < 2; ++side) {
                const int n = side == 0 ? n1 : n2;
                const float* A = side == 0 ? X : Y; const float* O = side == 0 ? Y : X;
                const float* w = side == 0 ? wx : wy; const float* g = side == 0 ? g1 : g2;
                const float* nPa = side == 0 ? S.nP1 : S.nP2; const float* nPo = side == 0 ? S.nP2 : S.nP1;
                const float* nQa = side == 0 ? S.nQ1 : S.nQ2; const float* nRa = side == 0 ? S.nR1 : S.nR2;
                const float* M = side == 0 ? S.M2 : S.M1; const float* T = side == 0 ? S.T2 : S.T1;
                const float* nzQ = side == 0 ? S.nzQ2 : S.nzQ1; const float* nzR = side == 0 ? S.nzR2 : S.nzR1;
                const float* mm = side == 0 ? S.mm1 : S.mm2; const float* mx = side == 0 ? S.mx1 : S.mx2;
                const float* mp = side == 0 ? S.m1 : S.m2; const int* argp = side == 0 ? S.j1s : S.i2s;
                const float* dzQ = side == 0 ? S.dzQ2 : S.dzQ1; const float* dzR = side == 0 ? S.dzR2 : S.dzR1;
                for (int i = 0; i < n; ++i) {
                    if (w[i] <= 0.f) continue;
                    const float x = A[(size_t)i * LDX + c];
                    const float zq = a.Q[c] * M[(size_t)i * d + c], zr = a.R[c] * T[(size_t)i * d + c];
                    dq += x * bm_dmatch(w[i] * g[H + k], zq, qk * x, mm[i * H + k], nQa[i * H + k], nzQ[i]);
                    dr += x * bm_dmatch(w[i] * g[2 * H + k], zr, rk * x, mx[i * H + k], nRa[i * H + k], nzR[i]);
                    if (k == 0) {                                      // perspective 0 also multiplies the attended vectors
                        dq += M[(size_t)i * d + c] * dzQ[(size_t)i * d + c];
                        dr += T[(size_t)i * d + c] * dzR[(size_t)i * d + c];
                    }
                    const int js = argp[i * H + k];
                    if (js >= 0) {                                     // both factors of the winning pair carry P_k
                        const float y = O[(size_t)js * LDX + c];
                        const float gam = w[i] * g[k], mv = mp[i * H + k], nu = nPa[i * H + k], nv = nPo[js * H + k];
                        dp += x * bm_dmatch(gam, pk * y, pk * x, mv, nu, nv) + y * bm_dmatch(gam, pk * x, pk * y, mv, nv, nu);
                    }
                }
            }
            wsl[idx] += dp; wsl[H * d + idx] += dq; wsl[2 * H * d + idx] += dr;
        }
    }
}

// dW[i] = sum over the workgroups' slabs, fixed order
__global__ __launch_bounds__(256) void k_bimpm_reduce(const float* __restrict__ slab, int nslab, int n, float* dP, float* dQ, float* dR,
                                                      int hd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int s = 0; s < nslab; ++s) v += slab[(size_t)s * n + i];
    float* o = i < hd ? dP + i : (i < 2 * hd ? dQ + (i - hd) : dR + (i - 2 * hd));
    *o = v;
}

static int bm_grid(int B) { return B < 512 ? B : 512; }

static size_t bm_lds_bytes(int maxn, int d) { return ((size_t)2 * maxn * (d + 1) + 2 * maxn + 16) * sizeof(float); }

static bool bm_in_lds(int maxn, int d) { return bm_lds_bytes(maxn, d) <= 160 * 1024; }
static size_t bm_stage_floats(int maxn, int d) { return bm_in_lds(maxn, d) ? 0 : (bm_lds_bytes(maxn, d) / sizeof(float) + 15) & ~(size_t)15; }

// Any molecule size: pairs that fit stage their rows in LDS, larger ones in the workspace (bmp_bimpm_ws_floats grows).
extern "C" int bmp_bimpm_supported(int d, int H, int maxn) { return d > 0 && H > 0 && maxn > 0; }

extern "C" size_t bmp_bimpm_ws_floats(int d, int H, int maxn, int B, int backward) {
    return (size_t)bm_grid(B) * (bm_scratch_floats(maxn, d, H, backward != 0) + bm_stage_floats(maxn, d) + (backward ? (size_t)3 * H * d : 0));
}

static int bm_fill(BmArgs& a, const float* X1, const float* X2, int d, int H, const float* w1, const int* r1, const int* n1,
                   const float* w2, const int* r2, const int* n2, int B, int maxn, const float* P, const float* Q, const float* R,
                   float* ws, bool bwd) {
    memset(&a, 0, sizeof(a));
    BMP_REQUIRE(X1 && X2 && w1 && w2 && r1 && n1 && r2 && n2 && P && Q && R && ws && B > 0);
    BMP_REQUIRE(bmp_bimpm_supported(d, H, maxn));
    a.X1 = X1; a.X2 = X2; a.d = d; a.H = H; a.B = B; a.w1 = w1; a.w2 = w2; a.r1 = r1; a.n1 = n1; a.r2 = r2; a.n2 = n2;
    a.P = P; a.Q = Q; a.R = R; a.maxn = maxn;
    a.scratch = ws; a.scratch_per_wg = bm_scratch_floats(maxn, d, H, bwd) + bm_stage_floats(maxn, d);
    a.stage_global = bm_in_lds(maxn, d) ? 0 : 1; a.stage_off = bm_scratch_floats(maxn, d, H, bwd);
    return 0;
}

// mol_1, mol_2 [B x 3H] of B drug pairs.  X1 / X2: packed atom rows of the two sides ([N x d]); w: row multiplicities;
// r / n: first row and row count of every pair's molecule (n <= maxn); P, Q, R [H x d]: max_pooling_W, att_mean_W,
// att_max_W.  ws: bmp_bimpm_ws_floats(d, H, maxn, B, 0) floats.
extern "C" int bmp_bimpm_fwd(const float* X1, const float* X2, int d, int H, const float* w1, const int* r1, const int* n1,
                             const float* w2, const int* r2, const int* n2, int B, int maxn, const float* P, const float* Q,
                             const float* R, float* out1, float* out2, float* ws, size_t ws_floats, hipStream_t st) {
    BmArgs a;
    int rc = bm_fill(a, X1, X2, d, H, w1, r1, n1, w2, r2, n2, B, maxn, P, Q, R, ws, false);
    if (rc) return rc;
    BMP_REQUIRE(out1 && out2 && ws_floats >= bmp_bimpm_ws_floats(d, H, maxn, B, 0));
    a.out1 = out1; a.out2 = out2;
    if (int rc_attr = bmp_lds_attr((const void*)k_bimpm<false>, (size_t)(160 * 1024))) return rc_attr;
    hipLaunchKernelGGL((k_bimpm<false>), dim3(bm_grid(B)), dim3(BM_NT), a.stage_global ? 64 : bm_lds_bytes(maxn, d), st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}

// Gradients of bmp_bimpm_fwd: dX1 / dX2 [N x d] (rows of the B pairs' molecules are written, other rows untouched), dP, dQ,
// dR [H x d] (overwritten).  The forward of every pair is recomputed.  ws: bmp_bimpm_ws_floats(d, H, maxn, B, 1) floats.
extern "C" int bmp_bimpm_bwd(const float* dout1, const float* dout2, const float* X1, const float* X2, int d, int H,
                             const float* w1, const int* r1, const int* n1, const float* w2, const int* r2, const int* n2, int B,
                             int maxn, const float* P, const float* Q, const float* R, float* dX1, float* dX2, float* dP, float* dQ,
                             float* dR, float* ws, size_t ws_floats, hipStream_t st) {
    BmArgs a;
    int rc = bm_fill(a, X1, X2, d, H, w1, r1, n1, w2, r2, n2, B, maxn, P, Q, R, ws, true);
    if (rc) return rc;
    BMP_REQUIRE(dout1 && dout2 && dX1 && dX2 && dP && dQ && dR && ws_floats >= bmp_bimpm_ws_floats(d, H, maxn, B, 1));
    a.dout1 = dout1; a.dout2 = dout2; a.dX1 = dX1; a.dX2 = dX2;
    const int grid = bm_grid(B);
    a.wslab = ws + (size_t)grid * a.scratch_per_wg;
    if (int rc_attr = bmp_lds_attr((const void*)k_bimpm<true>, (size_t)(160 * 1024))) return rc_attr;
    hipLaunchKernelGGL((k_bimpm<true>), dim3(grid), dim3(BM_NT), a.stage_global ? 64 : bm_lds_bytes(maxn, d), st, a);
    BMP_LAUNCH_CHECK();
    const int n = 3 * H * d;
    hipLaunchKernelGGL(k_bimpm_reduce, dim3((n + 255) / 256), dim3(256), 0, st, a.wslab, grid, n, dP, dQ, dR, H * d);
    BMP_LAUNCH_CHECK();
    return 0;
}
