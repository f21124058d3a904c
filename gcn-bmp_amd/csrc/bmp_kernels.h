// Internal launch helpers shared by the op-level C-ABI launchers (bmp_ops.hip, bmp_coattn.hip).
#pragma once
#include "bmp_common.h"

// ---------------------------------------------------------------------------------------------
// Row GEMM:  Y[N x Nout] = epi( sum_s X_s[N x K_s] (* X2_s) . Wt_s[K_s x Nout] )
//   * N is a multiple of BMP_R (packed layout), K_s a multiple of 8, ldx/ldx2 multiples of 4,
//     every X pointer 16-byte aligned.
//   * Wt_s is K-major ("transposed Linear weight"): element (k, n) at Wt[k*ldw + n].
//   * one workgroup = one 128-row tile x NT output columns; the row tile is staged through LDS
//     in 64-wide K chunks, weights stream from L2 straight into MFMA B registers.
// ---------------------------------------------------------------------------------------------
struct RGSrc {
    const float* X;      // [N x K]
    const float* X2;     // optional elementwise multiplicand of X (same shape), or nullptr
    const float* Wt;     // [K x >=Nout]
    int ldx, ldx2, ldw, K;
};

enum {
    BMP_EPI_GENERIC = 0,   // v (+bias) (+wdeg.bE) (+add) -> act -> Y / o1 (column split)
    BMP_EPI_GRU_OUT = 1,   // c = tanh(v + bias); h' = z*c + (1-z)*h  (first: h' = z*c)
    BMP_EPI_GRU_DRH = 2,   // v = dRH: da_r = v*h*r*(1-r) -> Y ; o1 += v*r
};

struct RGArgs {
    RGSrc s[3];
    int nsrc;
    int Nout;
    float* Y; int ldy;
    const float* bias;               // [Nout] or nullptr
    // GENERIC extras
    const float* add; int ldadd;     // added where col < split (or everywhere if split <= 0)
    const float* wdeg;               // [N x 4] per-row per-bond-type weighted degree, or nullptr
    const float* bE; int ldbE;       // [4 x Nout] per-bond-type bias
    int split;                       // column split (<=0: none)
    int act_lo, act_hi;              // activation below / at-or-above split (act_lo everywhere if no split)
    float* o1; int ldo1;             // columns >= split go to o1[row, col - split] when o1 != nullptr
    int accumulate;                  // Y += instead of Y =
    // row LIST (round 4; bmp_launch_rowgemm_listed): the launch computes the rows ridx[0 .. *rcnt) only -- A rows and Y rows are
    // both taken through the list (device arrays; tiles past the end of the list exit) -- for the per-bond-type blocks of the
    // unfused message operator's backward, whose other rows nobody reads.
    const int* ridx; const int* rcnt;
    // GRU extras
    const float* z; int ldz;
    const float* h; int ldh;
    const float* r; int ldr;
    float* c_out; int ldc;
    int first;
    int vec_epi;                     // set by the launchers (rg_vec_ok): the LDS-staged kernels write the tile row-major, 16 bytes per lane
};

int bmp_launch_rowgemm(const RGArgs& a, int n_tiles, int epi, hipStream_t st);
// generic epilogue, rows through a.ridx / a.rcnt; n_tiles_cap: tiles of the longest possible list (all rows)
int bmp_launch_rowgemm_listed(const RGArgs& a, int n_tiles_cap, hipStream_t st);
bool bmp_rowgemm_listed_ok(const RGArgs& a);      // the listed form exists for the LDS-staged kernel only (weight alignment; BMP_ROWGEMM_DIRECT unset)
// n <= 3 independent problems (generic epilogue) in ONE launch
int bmp_launch_rowgemm_multi(const RGArgs* a, const int* n_tiles, int n, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM:  out[K x Nn] (=|+=) sum_rows X[row, k] (* X2) . dY[row, n]
//   split over row chunks into slabs (deterministic), then reduced.
// ---------------------------------------------------------------------------------------------
struct WGArgs {
    const float* X; const float* X2; int ldx, ldx2;
    const float* dY; int ldy;
    int K, Nn, N;
    float* out; int ldo;             // [K x Nn]
    int accumulate;
    float* cs;                       // optional [Nn]: column sums of dY (bias gradient), same accumulate flag
    int cs_accumulate;               // ... or accumulated regardless of `accumulate` when != 0
    const int* onehot;               // optional [N]: X is not read, X[row, k] = (onehot[row] == k)
    // fused launches only (bmp_launch_wgrad_fused): the Nn LOGICAL columns skip the physical columns
    // [skip_at, skip_at + skip_n) of dY and of out / cs (skip_at a multiple of 128 or 0; the skipped output columns
    // are written as zeros unless accumulating); zero_only: no product at all, out is zero-filled unless accumulating
    int skip_at = 0x7fffffff, skip_n = 0;
    int zero_only = 0;
    // ... and weighted column sums riding along: wout[e, j - w_col0] (=|+=) sum_rows wrow[row, e] * dY[row, j] for the
    // logical columns j >= w_col0 (e < 4; wrow [N x 4]): the per-bond-type bias gradient of a RelGCN layer
    const float* wrow = nullptr; int w_col0 = 0; float* wout = nullptr; int ldwo = 0;
    // ... and a row LIST (round 4): the problem sums over the rows ridx[0 .. *rcnt) only (ascending row numbers; the other
    // rows of dY are known to be zero: the per-bond-type blocks G_e of a propagation step, bmp_type_rows).  ridx / rcnt are
    // DEVICE arrays (no host copy of the count exists): the launch splits the list into round(S * rfrac) parts whatever
    // the count turns out to be; rfrac = the expected share of listed rows (only the balance of the launch depends on it).
    const int* ridx = nullptr; const int* rcnt = nullptr; float rfrac = 1.f;
};
size_t bmp_wgrad_ws_floats(int N, int K, int Nn);
int bmp_launch_wgrad(const WGArgs& a, float* ws, hipStream_t st);
// n <= 3 independent problems (K <= 128, no X2) in ONE GEMM launch: for launches of a few tiles each
size_t bmp_wgrad_multi_ws_floats(const WGArgs* a, int n);
int bmp_launch_wgrad_multi(const WGArgs* a, int n, float* ws, hipStream_t st);
// n <= BMP_WG_MAXP problems over the SAME rows (K <= 128; X2, column skips, zero-only problems and row lists allowed) as ONE
// GEMM launch and ONE reduction launch: the weight gradients of a fused GGNN step / RelGCN layer.
#define BMP_WG_MAXP 8
bool bmp_wgrad_fused_lists_ok(int N);        // row lists need the LDS-DMA body (BMP_WGRAD_DMA != 0, N a multiple of 16)
size_t bmp_wgrad_fused_ws_floats(const WGArgs* a, int n);
int bmp_launch_wgrad_fused(const WGArgs* a, int n, float* ws, hipStream_t st, int kid);

// column sums: out[n] (=|+=) sum_rows dY[row, n]
size_t bmp_colsum_ws_floats(int N, int Nn);
int bmp_launch_colsum(const float* dY, int ldy, int N, int Nn, float* out, int accumulate, float* ws, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Graph kernels (bmp_graph.hip)
// ---------------------------------------------------------------------------------------------
// forward gather: agg[i, e*d + k] = sum_{(j,e) in csr(i)} val * x[j, k]; wdeg[i, e] = sum val
int bmp_launch_gather_fwd(const float* x, int ldx, int N, int d, const int* ptr, const int* col, const float* val,
                          float* agg, float* wdeg, hipStream_t st);
// backward gather (transposed CSR): dx[j, k] (=|+=) sum_{(i,e) in csrT(j)} val * dagg[i, e*d + k]
int bmp_launch_gather_bwd(const float* dagg, int N, int d, const int* ptrT, const int* colT, const float* valT,
                          float* dx, int lddx, int accumulate, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Optional per-kernel-class timing with HIP events (bench.py's roofline leg).  Off by default;
// the only process-global state in the library.  While a class is armed, every launch of that
// class is bracketed by two events on the launch stream.
// ---------------------------------------------------------------------------------------------
enum { BMP_KCLS_ROWGEMM = 1, BMP_KCLS_WGRAD = 2, BMP_KCLS_GATHER = 3, BMP_KCLS_COATTN = 4, BMP_KCLS_STEP_FWD = 5,
       BMP_KCLS_STEP_BWD = 6, BMP_KCLS_ALL = -1 };
// kernel ids inside a class (bmp_prof_collect reports per kernel: key = class * 16 + id)
enum { BMP_KID_ROWGEMM = 0, BMP_KID_ROWGEMM_MULTI = 1, BMP_KID_READOUT_TILE = 2,
       BMP_KID_WGRAD = 0, BMP_KID_WGRAD_X2 = 1, BMP_KID_WGRAD_ONEHOT = 2, BMP_KID_WGRAD_DIRECT = 3, BMP_KID_WGRAD_MULTI = 4, BMP_KID_WGRAD_STEP = 5,
       BMP_KID_COATTN_FWD = 0, BMP_KID_COATTN_BWD = 1,
       BMP_KID_GGNN_LATER = 0, BMP_KID_GGNN_FIRST = 1, BMP_KID_RELGCN = 2 };
struct BmpProfScope {
    BmpProfScope(int kclass, double flops, double bytes, hipStream_t st, int kid = 0);
    ~BmpProfScope();
    int slot;
    hipStream_t st;
};

// The weight-gradient launches of a backward entry point may go to a stream of their own (st_w; null or == st: in line):
// nothing downstream in the chain reads them.  st_w picks up after everything st has been given so far.
// (bmp_stream_after: `to` picks up behind what `from` holds now; a null handle there is the device's default stream.)
static inline int bmp_stream_after(hipStream_t from, hipStream_t to) {
    if (to == from) return 0;
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
    e = hipEventRecord(ev, from);
    if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
    (void)hipEventDestroy(ev);                      // released once the record has completed
    return (int)e;
}
static inline int bmp_fork_to(hipStream_t st, hipStream_t st_w) { return st_w ? bmp_stream_after(st, st_w) : 0; }

