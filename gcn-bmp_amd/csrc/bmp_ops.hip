// Op-level C-ABI launchers of the encoder path: message (+RelGCN layer), GRU update, gated
// readout.  Each launcher only enqueues kernels on the caller's stream: no allocation, no
// host sync, no global state; scratch comes from the caller (sizes via *_ws_floats()).
#include <string.h>
#include "bmp_kernels.h"

static inline RGArgs rg_zero() {
    RGArgs a;
    memset(&a, 0, sizeof(a));
    return a;
}

static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------
// small pointwise kernels
// ---------------------------------------------------------------------------------------------
// dpre = dout * act'(out)
__global__ void k_dact(const float* __restrict__ dout, int lddo, const float* __restrict__ out, int ldo, int act, int N,
                       int d, float* __restrict__ dpre) {
    const size_t total = (size_t)N * d;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / d), c = (int)(idx % d);
        dpre[idx] = dout[(size_t)row * lddo + c] * bmp_dact(act, out[(size_t)row * ldo + c]);
    }
}

// GRU backward, gate derivatives (chainer StatefulGRU):
//   h' = z*c + (1-z)*h : dc_pre = dh'*z*(1-c^2) ; dz_pre = dh'*(c-h)*z*(1-z) ; dh_acc = dh'*(1-z)
//   first call (h' = z*c): dz_pre = dh'*c*z*(1-z) ; dh_acc = 0 ; da_r = 0
__global__ void k_gru_bwd_gates(const float* __restrict__ dhn, const float* __restrict__ h, const float* __restrict__ rz,
                                const float* __restrict__ c, int N, int d, int first, float* __restrict__ da,
                                float* __restrict__ dhacc) {
    const size_t total = (size_t)N * d;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / d;
        const int k = (int)(idx % d);
        const float g = dhn[idx];
        const float z = rz[row * 2 * d + d + k];
        const float cv = c[idx];
        const float hv = first ? 0.f : h[idx];
        float* dar = da + row * 3 * d;
        dar[2 * d + k] = g * z * (1.f - cv * cv);
        dar[d + k] = g * (cv - hv) * z * (1.f - z);
        if (first) dar[k] = 0.f;
        dhacc[idx] = first ? 0.f : g * (1.f - z);
    }
}

// readout: g[mol, c] = sum_{rows of mol} w[row] * si[row, c] * jv[row, c]   (models/ggnn.py:337-340)
__global__ __launch_bounds__(256) void k_readout_segsum(const float* __restrict__ ij, int o, const float* __restrict__ w,
                                                        const int* __restrict__ mol_row0, const int* __restrict__ mol_nrows,
                                                        float* __restrict__ g) {
    const int mol = blockIdx.x;
    const int r0 = mol_row0[mol], nr = mol_nrows[mol];
    for (int c = threadIdx.x; c < o; c += 256) {
        float acc = 0.f;
        for (int r = r0; r < r0 + nr; ++r) acc += w[r] * ij[(size_t)r * 2 * o + c] * ij[(size_t)r * 2 * o + o + c];
        g[(size_t)mol * o + c] = acc;
    }
}

// readout backward, per molecule: dIJ[row, 0:o] = w*dg*jv*si*(1-si) ; dIJ[row, o:2o] = w*dg*si*act'(jv)
__global__ __launch_bounds__(256) void k_readout_bwd_rows(const float* __restrict__ dg, const float* __restrict__ ij, int o,
                                                          int act_j, const float* __restrict__ w,
                                                          const int* __restrict__ mol_row0, const int* __restrict__ mol_nrows,
                                                          float* __restrict__ dij) {
    const int mol = blockIdx.x;
    const int r0 = mol_row0[mol], nr = mol_nrows[mol];
    for (int c = threadIdx.x; c < o; c += 256) {
        const float gv = dg[(size_t)mol * o + c];
        for (int r = r0; r < r0 + nr; ++r) {
            const float si = ij[(size_t)r * 2 * o + c], jv = ij[(size_t)r * 2 * o + o + c];
            const float t = w[r] * gv;
            dij[(size_t)r * 2 * o + c] = t * jv * si * (1.f - si);
            dij[(size_t)r * 2 * o + o + c] = t * si * bmp_dact(act_j, jv);
        }
    }
}

static inline int ew_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : b);
}

// ---------------------------------------------------------------------------------------------
// message / RelGCN layer:  out = act( agg(x) . WT + wdeg . bE [+ x . WsT + bs] )
//   GGNN   (models/ggnn.py:215-243, models/update/ggnn_update.py:31-50): WsT = bs = null, act = none
//   RelGCN (models/update/relgcn_update.py:24-44 + tanh of models/relgcn.py:71): WsT, bs, act = tanh
// WT [4*d_in x d_out]: row e*d_in + k, col c  <->  reference W[4*c + e][k];  bE [4 x d_out].
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_msg_fwd(const float* x, int ldx, int n_tiles, int d_in, int d_out, const int* csr_ptr,
                           const int* csr_col, const float* csr_val, const float* WT, const float* bE, const float* WsT,
                           const float* bs, int act, float* agg, float* wdeg, float* out, int ldo, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && (d_in & 7) == 0 && d_in > 0 && d_out > 0);
    const int N = n_tiles * BMP_R;
    int rc = bmp_launch_gather_fwd(x, ldx, N, d_in, csr_ptr, csr_col, csr_val, agg, wdeg, st);
    if (rc) return rc;
    RGArgs a = rg_zero();
    a.s[0] = RGSrc{agg, nullptr, WT, 4 * d_in, 0, d_out, 4 * d_in};
    a.nsrc = 1;
    if (WsT) {
        a.s[1] = RGSrc{x, nullptr, WsT, ldx, 0, d_out, d_in};
        a.nsrc = 2;
    }
    a.Nout = d_out;
    a.Y = out; a.ldy = ldo;
    a.bias = bs;
    a.wdeg = wdeg; a.bE = bE; a.ldbE = d_out;
    a.act_lo = act; a.act_hi = act;
    return bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st);
}

// dWT [4 d_in x d_out] = agg^T . dpre with agg_e (columns [e d_in, (e + 1) d_in)) an exact zero in every row that has no bond of
// type e: per type and 128-wide K slice one problem over the type's row list (bmp_type_rows of the FORWARD CSR) -- 1.46 N rows
// instead of 4 N on the DDI batches.  0: the shape does not fit the fused launch (more than BMP_WG_MAXP problems, K slices
// below 64) and the caller takes the all-rows launch.
static const float kMsgTypeFrac[4] = {0.78f, 0.24f, 0.05f, 0.58f};
static int msg_listed_problems(WGArgs* g, const float* agg, const float* dpre, int lddp, int N, int d_in, int d_out, float* dWT, int acc,
                               const int* type_rows, const int* type_cnt) {
    const int ks = d_in > 128 ? 128 : d_in, nk = (d_in + ks - 1) / ks;
    if (4 * nk > BMP_WG_MAXP || (d_in % ks) != 0 || ks < 64 || d_out < 64 || (d_out & 3) || !bmp_wgrad_fused_lists_ok(N) || (N & 31)) return 0;
    int n = 0;
    for (int e = 0; e < 4; ++e)
        for (int k = 0; k < nk; ++k) {
            g[n] = WGArgs{agg ? agg + e * d_in + k * ks : nullptr, nullptr, 4 * d_in, 0, dpre, lddp, ks, d_out, N,
                          dWT ? dWT + (size_t)(e * d_in + k * ks) * d_out : nullptr, d_out, acc};
            g[n].ridx = type_rows + (size_t)e * N; g[n].rcnt = type_cnt + e; g[n].rfrac = kMsgTypeFrac[e];
            ++n;
        }
    return n;
}

extern "C" size_t bmp_msg_bwd_ws_floats(int n_tiles, int d_in, int d_out) {
    const int N = n_tiles * BMP_R;
    size_t slab = max_sz(bmp_wgrad_ws_floats(N, 4 * d_in, d_out), bmp_wgrad_ws_floats(N, d_in, d_out));
    slab = max_sz(slab, bmp_wgrad_ws_floats(N, 4, d_out));
    slab = max_sz(slab, bmp_colsum_ws_floats(N, d_out));
    {   // the listed form of dWT (msg_listed_problems)
        WGArgs g[BMP_WG_MAXP];
        const int n = msg_listed_problems(g, nullptr, nullptr, 0, N, d_in, d_out, nullptr, 0, (const int*)16, (const int*)16);
        if (n > 0) slab = max_sz(slab, bmp_wgrad_fused_ws_floats(g, n));
    }
    return (size_t)N * 4 * d_in + (size_t)N * d_out + slab;
}

// Wnat [d_out x 4*d_in] = WT^T ; Ws [d_out x d_in] = WsT^T (reference Linear layout).
extern "C" int bmp_msg_bwd(const float* dout, int lddo, const float* out, int ldo, int act, const float* x, int ldx,
                           int n_tiles, int d_in, int d_out, const int* csrT_ptr, const int* csrT_col,
                           const float* csrT_val, const float* Wnat, const float* Ws, const float* agg, const float* wdeg,
                           float* dx, float* dWT, float* dbE, float* dWsT, float* dbs, int accumulate_w, const int* type_rows_f,
                           const int* type_cnt_f, float* ws, size_t ws_floats, hipStream_t st, hipStream_t st_w) {
    BMP_REQUIRE(n_tiles > 0 && (d_in & 7) == 0 && (d_out & 7) == 0 && d_in > 0 && d_out > 0);
    const int acc = accumulate_w ? 1 : 0;          // the weight gradients add into their outputs (a tied layer's later calls)
    BMP_REQUIRE(ws_floats >= bmp_msg_bwd_ws_floats(n_tiles, d_in, d_out));
    const int N = n_tiles * BMP_R;
    float* dagg = ws;
    float* dpre_buf = dagg + (size_t)N * 4 * d_in;
    float* slab = dpre_buf + (size_t)N * d_out;
    const float* dpre = dout;
    int lddp = lddo;
    if (act != BMP_ACT_NONE) {
        hipLaunchKernelGGL(k_dact, dim3(ew_blocks((size_t)N * d_out)), dim3(256), 0, st, dout, lddo, out, ldo, act, N, d_out,
                           dpre_buf);
        BMP_LAUNCH_CHECK();
        dpre = dpre_buf;
        lddp = d_out;
    } else {
        BMP_REQUIRE((lddo & 3) == 0);
    }
    int rc;
    if (!st_w) st_w = st;
    if ((rc = bmp_fork_to(st, st_w))) return rc;        // dpre is complete: the weight gradients may start
    // dagg = dpre . Wnat.  Only the (row, type) blocks that the transposed gather below reads are needed -- a row's block e is read
    // through the row's own bonds of type e -- so with the row lists of the forward CSR the product runs per type over the
    // listed rows (1.46 N instead of 4 N row blocks on the DDI batches); the blocks that are not computed are not read.
    // (The forward product out = sum_e agg_e . W_e was tried the same way -- out cleared, four listed launches adding to their
    //  rows -- and lost 5 % of the C4 step to the read-modify-write of out and to its four part-filled launches per chain.)
    bool listed = type_rows_f && type_cnt_f && (d_in & 3) == 0 && ((uintptr_t)Wnat & 15) == 0 && ((uintptr_t)dpre & 15) == 0 &&
                  (lddp & 3) == 0 && n_tiles >= 64;
    if (listed) {
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{dpre, nullptr, Wnat, lddp, 0, 4 * d_in, d_out};
        a.nsrc = 1; a.Nout = d_in; a.ridx = type_rows_f; a.rcnt = type_cnt_f;
        listed = bmp_rowgemm_listed_ok(a);
    }
    if (listed) {
        for (int e = 0; e < 4; ++e) {
            RGArgs a = rg_zero();
            a.s[0] = RGSrc{dpre, nullptr, Wnat + e * d_in, lddp, 0, 4 * d_in, d_out};
            a.nsrc = 1; a.Nout = d_in; a.Y = dagg + e * d_in; a.ldy = 4 * d_in;
            a.ridx = type_rows_f + (size_t)e * N; a.rcnt = type_cnt_f + e;
            if ((rc = bmp_launch_rowgemm_listed(a, n_tiles, st))) return rc;
        }
    } else {
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{dpre, nullptr, Wnat, lddp, 0, 4 * d_in, d_out};
        a.nsrc = 1; a.Nout = 4 * d_in; a.Y = dagg; a.ldy = 4 * d_in;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    // dx = [dpre . Ws] + gather^T(dagg)
    int accumulate = 0;
    if (Ws) {
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{dpre, nullptr, Ws, lddp, 0, d_in, d_out};
        a.nsrc = 1; a.Nout = d_in; a.Y = dx; a.ldy = d_in;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
        accumulate = 1;
    }
    if ((rc = bmp_launch_gather_bwd(dagg, N, d_in, csrT_ptr, csrT_col, csrT_val, dx, d_in, accumulate, st))) return rc;
    // weight gradients
    {
        WGArgs gl[BMP_WG_MAXP];
        const int nl = (type_rows_f && type_cnt_f && ((uintptr_t)agg & 15) == 0 && ((uintptr_t)dpre & 15) == 0 && (lddp & 3) == 0)
                           ? msg_listed_problems(gl, agg, dpre, lddp, N, d_in, d_out, dWT, acc, type_rows_f, type_cnt_f) : 0;
        if (nl > 0) {
            if ((rc = bmp_launch_wgrad_fused(gl, nl, slab, st_w, BMP_KID_WGRAD_STEP))) return rc;
        } else {
            WGArgs g{agg, nullptr, 4 * d_in, 0, dpre, lddp, 4 * d_in, d_out, N, dWT, d_out, acc};
            if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
        }
    }
    {
        WGArgs g{wdeg, nullptr, 4, 0, dpre, lddp, 4, d_out, N, dbE, d_out, acc};
        if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
    }
    if (Ws) {
        WGArgs g{x, nullptr, ldx, 0, dpre, lddp, d_in, d_out, N, dWsT, d_out, acc, dbs};      // + dbs = column sums of dpre
        if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// GRU update (chainer links.GRU = StatefulGRU, models/ggnn.py:132,254-262).
// The caller folds the state terms into the x-weights (s == h without dropout):
//   AT [2d x 3d] K-major, rows [h-part ; m-part], cols [r | z | c]:
//       later calls: h-part of r,z = W_{r,z}[:, :d]^T + U_{r,z}^T ; b[0:2d] = bW + bU ; b[2d:3d] = bW_c + bU_c
//       first call : plain W^T, b = bW (U terms and U biases do not exist in that branch)
//   UcT [d x d] = U^T  (candidate: + (r*h) . U^T), unused when first.
// Saves rz [N x 2d] (sigmoid outputs r | z) and c [N x d] (tanh output) for the backward.
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_gru_fwd(const float* h, const float* m, int n_tiles, int d, int first, const float* AT,
                           const float* UcT, const float* b, float* rz, float* c, float* hout, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0);
    int rc;
    {   // gates: first call needs z only
        const int off = first ? d : 0;
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{h, nullptr, AT + off, d, 0, 3 * d, d};
        a.s[1] = RGSrc{m, nullptr, AT + (size_t)d * 3 * d + off, d, 0, 3 * d, d};
        a.nsrc = 2;
        a.Nout = first ? d : 2 * d;
        a.Y = rz + off; a.ldy = 2 * d;
        a.bias = b + off;
        a.act_lo = a.act_hi = BMP_ACT_SIGMOID;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {   // candidate + interpolation
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{h, nullptr, AT + 2 * d, d, 0, 3 * d, d};
        a.s[1] = RGSrc{m, nullptr, AT + (size_t)d * 3 * d + 2 * d, d, 0, 3 * d, d};
        a.nsrc = 2;
        if (!first) {
            a.s[2] = RGSrc{rz, h, UcT, 2 * d, d, d, d};        // (r * h) . U^T
            a.nsrc = 3;
        }
        a.Nout = d;
        a.Y = hout; a.ldy = d;
        a.bias = b + 2 * d;
        a.z = rz + d; a.ldz = 2 * d;
        a.h = h; a.ldh = d;
        a.c_out = c; a.ldc = d;
        a.first = first;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GRU_OUT, st))) return rc;
    }
    return 0;
}

extern "C" size_t bmp_gru_bwd_ws_floats(int n_tiles, int d) {
    const int N = n_tiles * BMP_R;
    size_t slab = max_sz(bmp_wgrad_ws_floats(N, d, 3 * d), bmp_wgrad_ws_floats(N, d, d));
    slab = max_sz(slab, bmp_colsum_ws_floats(N, 3 * d));
    return (size_t)N * 3 * d + (size_t)N * d + slab;
}

// A [3d x 2d] = AT^T (rows r|z|c, cols h-part|m-part) ; Uc [d x d] = U (reference layout, out x in).
extern "C" int bmp_gru_bwd(const float* dhout, const float* h, const float* m, const float* rz, const float* c,
                           int n_tiles, int d, int first, const float* A, const float* Uc, float* dh, float* dm,
                           float* dAT, float* dUcT, float* db, int accumulate_w, float* ws, size_t ws_floats,
                           hipStream_t st, hipStream_t st_w) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0);
    BMP_REQUIRE(ws_floats >= bmp_gru_bwd_ws_floats(n_tiles, d));
    const int acc = accumulate_w ? 1 : 0;          // the weight gradients add into their outputs (a tied step's later calls)
    const int N = n_tiles * BMP_R;
    float* da = ws;                              // [N x 3d]  (da_r | da_z | da_c), pre-activation grads
    float* dhacc = da + (size_t)N * 3 * d;       // [N x d]
    float* slab = dhacc + (size_t)N * d;
    int rc;
    hipLaunchKernelGGL(k_gru_bwd_gates, dim3(ew_blocks((size_t)N * d)), dim3(256), 0, st, dhout, h, rz, c, N, d, first, da,
                       dhacc);
    BMP_LAUNCH_CHECK();
    if (!first) {   // d(r*h) = da_c . U ; da_r = d(r*h)*h*r*(1-r) ; dhacc += d(r*h)*r
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{da + 2 * d, nullptr, Uc, 3 * d, 0, d, d};
        a.nsrc = 1; a.Nout = d;
        a.Y = da; a.ldy = 3 * d;
        a.r = rz; a.ldr = 2 * d;
        a.h = h; a.ldh = d;
        a.o1 = dhacc; a.ldo1 = d;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GRU_DRH, st))) return rc;
    }
    if (!st_w) st_w = st;
    if ((rc = bmp_fork_to(st, st_w))) return rc;        // da is complete: the weight gradients may start
    {   // [dh | dm] = da . A ; dh += dhacc
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{da, nullptr, A, 3 * d, 0, 2 * d, 3 * d};
        a.nsrc = 1; a.Nout = 2 * d;
        a.Y = dh; a.ldy = d;
        a.add = dhacc; a.ldadd = d;
        a.split = d;
        a.o1 = dm; a.ldo1 = d;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {   // dAT rows 0..d-1 = h^T . da ; rows d..2d-1 = m^T . da
        // db = column sums of da: carried by the first GEMM (it reads da anyway; a pass of its own re-read N x 3d floats)
        WGArgs g{h, nullptr, d, 0, da, 3 * d, d, 3 * d, N, dAT, 3 * d, acc, db};
        if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
        WGArgs g2{m, nullptr, d, 0, da, 3 * d, d, 3 * d, N, dAT + (size_t)d * 3 * d, 3 * d, acc};
        if ((rc = bmp_launch_wgrad(g2, slab, st_w))) return rc;
    }
    if (!first) {   // dUcT = (r*h)^T . da_c
        WGArgs g{rz, h, 2 * d, d, da + 2 * d, 3 * d, d, d, N, dUcT, d, acc};
        if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
    } else if (!acc) {
        hipError_t e = hipMemsetAsync(dUcT, 0, (size_t)d * d * sizeof(float), st_w);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// GRU update with its state APART from its input -- what dropout on the step output makes of the update
// (models/ggnn.py:626-627: F.dropout(h) after every step, while the stateful GRU keeps the un-dropped h as its state):
//   x = [hd, m] (hd = dropout(s)), state s:   r = sig(W_r x + U_r s + b), z = sig(W_z x + U_z s + b),
//   c = tanh(W x + U (r*s) + b),  s' = z c + (1 - z) s.      Later calls only (the first call has no state: bmp_gru_fwd).
// WT [2d x 3d] = [W_r | W_z | W]^T (rows [hd-part ; m-part]), UrzT [d x 2d] = [U_r | U_z]^T, UcT [d x d] = U^T,
// b [3d] = bW + bU.  Saves rz [N x 2d] and c [N x d].
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_gru_state_fwd(const float* hd, const float* m, const float* s, int n_tiles, int d, const float* WT,
                                 const float* UrzT, const float* UcT, const float* b, float* rz, float* c, float* sout,
                                 hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0 && hd && m && s && WT && UrzT && UcT && b && rz && c && sout);
    int rc;
    {
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{hd, nullptr, WT, d, 0, 3 * d, d};
        a.s[1] = RGSrc{m, nullptr, WT + (size_t)d * 3 * d, d, 0, 3 * d, d};
        a.s[2] = RGSrc{s, nullptr, UrzT, d, 0, 2 * d, d};
        a.nsrc = 3;
        a.Nout = 2 * d;
        a.Y = rz; a.ldy = 2 * d;
        a.bias = b;
        a.act_lo = a.act_hi = BMP_ACT_SIGMOID;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{hd, nullptr, WT + 2 * d, d, 0, 3 * d, d};
        a.s[1] = RGSrc{m, nullptr, WT + (size_t)d * 3 * d + 2 * d, d, 0, 3 * d, d};
        a.s[2] = RGSrc{rz, s, UcT, 2 * d, d, d, d};              // (r * s) . U^T
        a.nsrc = 3;
        a.Nout = d;
        a.Y = sout; a.ldy = d;
        a.bias = b + 2 * d;
        a.z = rz + d; a.ldz = 2 * d;
        a.h = s; a.ldh = d;
        a.c_out = c; a.ldc = d;
        a.first = 0;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GRU_OUT, st))) return rc;
    }
    return 0;
}

extern "C" size_t bmp_gru_state_bwd_ws_floats(int n_tiles, int d) { return bmp_gru_bwd_ws_floats(n_tiles, d); }

// Gradients of bmp_gru_state_fwd.  A [3d x 2d] = WT^T, Urz [2d x d] = UrzT^T, Uc [d x d] = UcT^T (reference layouts).
// Outputs: dhd, dm, ds [N x d]; dWT [2d x 3d], dUrzT [d x 2d], dUcT [d x d], db [3d].
extern "C" int bmp_gru_state_bwd(const float* dsout, const float* hd, const float* m, const float* s, const float* rz,
                                 const float* c, int n_tiles, int d, const float* A, const float* Urz, const float* Uc, float* dhd,
                                 float* dm, float* ds, float* dWT, float* dUrzT, float* dUcT, float* db, float* ws,
                                 size_t ws_floats, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0 && ws_floats >= bmp_gru_state_bwd_ws_floats(n_tiles, d));
    BMP_REQUIRE(dsout && hd && m && s && rz && c && A && Urz && Uc && dhd && dm && ds && dWT && dUrzT && dUcT && db && ws);
    const int N = n_tiles * BMP_R;
    float* da = ws;                              // [N x 3d]  (da_r | da_z | da_c)
    float* dsacc = da + (size_t)N * 3 * d;       // [N x d]   direct part of ds: dsout (1 - z) + d(r*s) r
    float* slab = dsacc + (size_t)N * d;
    int rc;
    hipLaunchKernelGGL(k_gru_bwd_gates, dim3(ew_blocks((size_t)N * d)), dim3(256), 0, st, dsout, s, rz, c, N, d, 0, da, dsacc);
    BMP_LAUNCH_CHECK();
    {   // d(r*s) = da_c . U ; da_r = d(r*s) s r (1 - r) ; dsacc += d(r*s) r
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{da + 2 * d, nullptr, Uc, 3 * d, 0, d, d};
        a.nsrc = 1; a.Nout = d;
        a.Y = da; a.ldy = 3 * d;
        a.r = rz; a.ldr = 2 * d;
        a.h = s; a.ldh = d;
        a.o1 = dsacc; a.ldo1 = d;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GRU_DRH, st))) return rc;
    }
    {   // [dhd | dm] = da . A
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{da, nullptr, A, 3 * d, 0, 2 * d, 3 * d};
        a.nsrc = 1; a.Nout = 2 * d;
        a.Y = dhd; a.ldy = d;
        a.split = d;
        a.o1 = dm; a.ldo1 = d;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {   // ds = dsacc + [da_r | da_z] . Urz
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{da, nullptr, Urz, 3 * d, 0, d, 2 * d};
        a.nsrc = 1; a.Nout = d;
        a.Y = ds; a.ldy = d;
        a.add = dsacc; a.ldadd = d;
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {
        WGArgs g{hd, nullptr, d, 0, da, 3 * d, d, 3 * d, N, dWT, 3 * d, 0, db};       // + db = column sums of da
        if ((rc = bmp_launch_wgrad(g, slab, st))) return rc;
        WGArgs g2{m, nullptr, d, 0, da, 3 * d, d, 3 * d, N, dWT + (size_t)d * 3 * d, 3 * d, 0};
        if ((rc = bmp_launch_wgrad(g2, slab, st))) return rc;
        WGArgs g3{s, nullptr, d, 0, da, 3 * d, d, 2 * d, N, dUrzT, 2 * d, 0};
        if ((rc = bmp_launch_wgrad(g3, slab, st))) return rc;
        WGArgs g4{rz, s, 2 * d, d, da + 2 * d, 3 * d, d, d, N, dUcT, d, 0};
        if ((rc = bmp_launch_wgrad(g4, slab, st))) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// gated readout:  g[mol] = sum_rows w * sigmoid(i([h,h0])) * act_j(j([h,h0]))
//   models/ggnn.py:333-341 (j sees h only: the caller zeroes j's h0 rows of WT)
//   models/readout/ggnn_readout.py:42-57 (both see [h,h0], or h alone when h0 == null)
// WT [(d + d0) x 2o] K-major, cols [i | j]; b [2o] or null.  Saves ij [N x 2o] = (sigmoid(i) | act_j(j)).
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_readout_fwd(const float* h, const float* h0, int n_tiles, int d, int d0, int o, const float* WT,
                               const float* b, int act_j, const float* row_w, const int* mol_row0, const int* mol_nrows,
                               int n_mols, float* ij, float* g, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0 && o > 0 && n_mols > 0);
    BMP_REQUIRE(h0 == nullptr || (d0 > 0 && (d0 & 7) == 0));
    RGArgs a = rg_zero();
    a.s[0] = RGSrc{h, nullptr, WT, d, 0, 2 * o, d};
    a.nsrc = 1;
    if (h0) {
        a.s[1] = RGSrc{h0, nullptr, WT + (size_t)d * 2 * o, d0, 0, 2 * o, d0};
        a.nsrc = 2;
    }
    a.Nout = 2 * o;
    a.Y = ij; a.ldy = 2 * o;
    a.bias = b;
    a.split = o;
    a.act_lo = BMP_ACT_SIGMOID; a.act_hi = act_j;
    int rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_readout_segsum, dim3(n_mols), dim3(256), 0, st, ij, o, row_w, mol_row0, mol_nrows, g);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t bmp_readout_bwd_ws_floats(int n_tiles, int d, int d0, int o) {
    const int N = n_tiles * BMP_R;
    size_t slab = max_sz(bmp_wgrad_ws_floats(N, d, 2 * o), d0 > 0 ? bmp_wgrad_ws_floats(N, d0, 2 * o) : 0);
    slab = max_sz(slab, bmp_colsum_ws_floats(N, 2 * o));
    return (size_t)N * 2 * o + slab;
}

// Wnat [2o x (d + d0)] = WT^T.
extern "C" int bmp_readout_bwd(const float* dg, const float* h, const float* h0, int n_tiles, int d, int d0, int o,
                               const float* Wnat, const float* ij, int act_j, const float* row_w, const int* mol_row0,
                               const int* mol_nrows, int n_mols, float* dh, float* dh0, float* dWT, float* db,
                               int accumulate_w, float* ws, size_t ws_floats, hipStream_t st, hipStream_t st_w) {
    BMP_REQUIRE(n_tiles > 0 && d > 0 && (d & 7) == 0 && o > 0 && (o & 3) == 0 && n_mols > 0);
    if (!h0) d0 = 0;
    BMP_REQUIRE(ws_floats >= bmp_readout_bwd_ws_floats(n_tiles, d, d0, o));
    const int N = n_tiles * BMP_R;
    float* dij = ws;
    float* slab = dij + (size_t)N * 2 * o;
    hipError_t e = hipMemsetAsync(dij, 0, (size_t)N * 2 * o * sizeof(float), st);   // dead rows stay 0
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_readout_bwd_rows, dim3(n_mols), dim3(256), 0, st, dg, ij, o, act_j, row_w, mol_row0, mol_nrows, dij);
    BMP_LAUNCH_CHECK();
    int rc;
    const int acc = accumulate_w ? 1 : 0;
    if (!st_w) st_w = st;
    if ((rc = bmp_fork_to(st, st_w))) return rc;        // dij is complete
    {   // [dh | dh0] = dij . Wnat
        RGArgs a = rg_zero();
        a.s[0] = RGSrc{dij, nullptr, Wnat, 2 * o, 0, d + d0, 2 * o};
        a.nsrc = 1; a.Nout = d + d0;
        a.Y = dh; a.ldy = d;
        if (h0) { a.split = d; a.o1 = dh0; a.ldo1 = d0; }
        if ((rc = bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st))) return rc;
    }
    {
        WGArgs g{h, nullptr, d, 0, dij, 2 * o, d, 2 * o, N, dWT, 2 * o, acc, db};          // + db = column sums of dij (db may be null)
        if ((rc = bmp_launch_wgrad(g, slab, st_w))) return rc;
        if (h0) {
            WGArgs g2{h0, nullptr, d0, 0, dij, 2 * o, d0, 2 * o, N, dWT + (size_t)d * 2 * o, 2 * o, acc};
            if ((rc = bmp_launch_wgrad(g2, slab, st_w))) return rc;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// generic dense layer on row tiles (co-attention projections, tests):
//   Y = act(X . WT + b) ;  backward pieces exposed separately.
// ---------------------------------------------------------------------------------------------
extern "C" int bmp_linear_fwd(const float* X, int ldx, int n_tiles, int K, int Nout, const float* WT, int ldw,
                              const float* b, int act, float* Y, int ldy, hipStream_t st) {
    RGArgs a = rg_zero();
    a.s[0] = RGSrc{X, nullptr, WT, ldx, 0, ldw, K};
    a.nsrc = 1; a.Nout = Nout; a.Y = Y; a.ldy = ldy; a.bias = b;
    a.act_lo = a.act_hi = act;
    return bmp_launch_rowgemm(a, n_tiles, BMP_EPI_GENERIC, st);
}

extern "C" size_t bmp_wgrad_ws_floats_c(int N, int K, int Nn) {
    return max_sz(bmp_wgrad_ws_floats(N, K, Nn), bmp_colsum_ws_floats(N, Nn));
}

// dWT [K x Nn] = X^T . dY over N rows ; db [Nn] = column sums of dY (db may be null)
extern "C" int bmp_linear_wgrad(const float* X, int ldx, const float* dY, int ldy, int N, int K, int Nn, float* dWT,
                                float* db, float* ws, size_t ws_floats, hipStream_t st) {
    BMP_REQUIRE(ws_floats >= bmp_wgrad_ws_floats_c(N, K, Nn));
    WGArgs g{X, nullptr, ldx, 0, dY, ldy, K, Nn, N, dWT, Nn, 0};
    int rc = bmp_launch_wgrad(g, ws, st);
    if (rc) return rc;
    if (db) return bmp_launch_colsum(dY, ldy, N, Nn, db, 0, ws, st);
    return 0;
}

extern "C" int bmp_version(void) { return 408; }   // round 4: bumped whenever a kernel changes (bench.py ties PMC profiles to it)
extern "C" int bmp_tile_rows(void) { return BMP_R; }
