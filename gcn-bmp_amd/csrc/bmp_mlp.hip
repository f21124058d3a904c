// Link-predictor tail of the pair path as a handful of launches: MLP (models/mlp.py:20-45: Linear -> relu -> ... ->
// Linear on [g1 | g2]) forward / backward and sigmoid cross entropy (chainer.functions.sigmoid_cross_entropy,
// train_ddi_modify.py:285).  The arithmetic is tiny (B x 256 -> 32 -> 16 -> C on B ~ 1000 rows); what it cost as
// framework ops was ~40 launches of a few microseconds each per step.  Plain fp32 FMA, fixed summation orders.
#include <string.h>
#include "bmp_common.h"
#include "bmp_kernels.h"

#define MLP_MAXL 4          // Linear layers
#define MLP_MAXW 64         // widest hidden / output layer
#define MLP_MAXIN 1024      // widest input
#define MLP_FR 8            // rows per workgroup, forward
#define MLP_BR 8            // rows per workgroup, backward

struct MlpArgs {
    const float* x1; const float* x2; int d1, d2;        // input row = [x1 row | x2 row]
    int B, nl;
    int dims[MLP_MAXL + 1];
    const float* W[MLP_MAXL]; const float* b[MLP_MAXL];   // W[l] [dims[l+1] x dims[l]] (reference layout), b may be null
    float* act[MLP_MAXL];                                 // act[l] [B x dims[l+1]]: relu outputs, last = logits
    // backward
    const float* dy; float* dx1; float* dx2; float* slab; int slab_stride;
};

__global__ __launch_bounds__(256) void k_mlp_fwd(MlpArgs a) {
    __shared__ float buf[2][MLP_FR][MLP_MAXIN];
    extern __shared__ float wt[];                     // first-layer weights, transposed: wt[k * (no + 1) + j]
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * MLP_FR;
    const int in0 = a.dims[0];
    for (int idx = tid; idx < MLP_FR * in0; idx += 256) {
        const int r = idx / in0, k = idx % in0;
        const int row = row0 + r;
        float v = 0.f;
        if (row < a.B) v = k < a.d1 ? a.x1[(size_t)row * a.d1 + k] : a.x2[(size_t)row * a.d2 + (k - a.d1)];
        buf[0][r][k] = v;
    }
    {   // W[0] is [no x ni] row-major: read along k (coalesced), write transposed with an odd row stride
        const int ni = a.dims[0], no = a.dims[1];
        const float* __restrict__ W = a.W[0];
        for (int idx = tid; idx < no * ni; idx += 256) wt[(idx % ni) * (no + 1) + idx / ni] = W[idx];
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < a.nl; ++l) {
        const int ni = a.dims[l], no = a.dims[l + 1];
        const float* __restrict__ W = a.W[l];
        const float* __restrict__ bb = a.b[l];
        const bool last = l == a.nl - 1;
        for (int idx = tid; idx < MLP_FR * no; idx += 256) {
            const int r = idx / no, j = idx % no;
            float acc = bb ? bb[j] : 0.f;
            const float* x = buf[cur][r];
            if (l == 0) {
                const float* w = wt + j;
#pragma unroll 8
                for (int k = 0; k < ni; ++k) acc += x[k] * w[k * (no + 1)];
            } else {
                const float* w = W + (size_t)j * ni;      // <= 64 x 64: cache resident
#pragma unroll 8
                for (int k = 0; k < ni; ++k) acc += x[k] * w[k];
            }
            if (!last) acc = acc > 0.f ? acc : 0.f;
            buf[cur ^ 1][r][j] = acc;
            if (row0 + r < a.B) a.act[l][(size_t)(row0 + r) * no + j] = acc;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// slab layout per workgroup: for every layer l: dW[l] [no x ni] then db[l] [no]
// DX: the gradient of the MLP's input rows (what the backward chain waits for); DW: the workgroup's weight / bias gradient
// partials (35 KB per workgroup at 256 -> 32 -> 16 -> 1, folded by k_mlp_reduce).  The chain launches <true, false>, the
// stream beside it <false, true>: both walk the few hundred multiply-adds per row of the layers' backprop themselves.
template <bool DX, bool DW>
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpArgs a) {
    __shared__ float xin[MLP_BR][MLP_MAXIN];              // layer-0 input rows
    __shared__ float hid[MLP_MAXL][MLP_BR][MLP_MAXW];     // relu outputs of the hidden layers
    __shared__ float dcur[2][MLP_BR][MLP_MAXW];           // gradient w.r.t. a layer's pre-activation output
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * MLP_BR;
    const int in0 = a.dims[0];
    if (DW)
    for (int idx = tid; idx < MLP_BR * in0; idx += 256) {
        const int r = idx / in0, k = idx % in0, row = row0 + r;
        float v = 0.f;
        if (row < a.B) v = k < a.d1 ? a.x1[(size_t)row * a.d1 + k] : a.x2[(size_t)row * a.d2 + (k - a.d1)];
        xin[r][k] = v;
    }
    for (int l = 0; l + 1 < a.nl; ++l) {
        const int no = a.dims[l + 1];
        for (int idx = tid; idx < MLP_BR * no; idx += 256) {
            const int r = idx / no, j = idx % no, row = row0 + r;
            hid[l][r][j] = row < a.B ? a.act[l][(size_t)row * no + j] : 0.f;
        }
    }
    {
        const int no = a.dims[a.nl];
        for (int idx = tid; idx < MLP_BR * no; idx += 256) {
            const int r = idx / no, j = idx % no, row = row0 + r;
            dcur[0][r][j] = row < a.B ? a.dy[(size_t)row * no + j] : 0.f;
        }
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
    int off_l[MLP_MAXL];
    {
        int o = 0;
        for (int l = 0; l < a.nl; ++l) { off_l[l] = o; o += a.dims[l + 1] * (a.dims[l] + 1); }
    }
    int cur = 0;
    for (int l = a.nl - 1; l >= 0; --l) {
        const int ni = a.dims[l], no = a.dims[l + 1];
        const float* __restrict__ W = a.W[l];
        // weight / bias gradient partials of this workgroup's rows
        float* dW = slab + off_l[l];
        if (DW) {
        for (int k = tid; k < ni; k += 256) {             // thread owns input column k: its MLP_BR inputs stay in registers
            float xk[MLP_BR];
#pragma unroll
            for (int r = 0; r < MLP_BR; ++r) xk[r] = l == 0 ? xin[r][k] : hid[l - 1][r][k];
#pragma unroll 8
            for (int j = 0; j < no; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < MLP_BR; ++r) acc += dcur[cur][r][j] * xk[r];
                dW[j * ni + k] = acc;
            }
        }
        for (int j = tid; j < no; j += 256) {
            float acc = 0.f;
            for (int r = 0; r < MLP_BR; ++r) acc += dcur[cur][r][j];
            dW[no * ni + j] = acc;
        }
        }
        // gradient w.r.t. the layer input
        if (l > 0) {
            for (int idx = tid; idx < MLP_BR * ni; idx += 256) {
                const int r = idx / ni, k = idx % ni;
                float acc = 0.f;
#pragma unroll 8
                for (int j = 0; j < no; ++j) acc += dcur[cur][r][j] * W[(size_t)j * ni + k];
                dcur[cur ^ 1][r][k] = hid[l - 1][r][k] > 0.f ? acc : 0.f;
            }
        } else if (DX) {
            // a thread owns input column k for all rows of the workgroup: W[j][k] is read once per j, not once per row and j
            // (256 loads per thread in the row-major walk: most of this launch's 37 us)
            for (int k = tid; k < ni; k += 256) {
                float acc[MLP_BR];
#pragma unroll
                for (int r = 0; r < MLP_BR; ++r) acc[r] = 0.f;
#pragma unroll 8
                for (int j = 0; j < no; ++j) {
                    const float wv = W[(size_t)j * ni + k];
#pragma unroll
                    for (int r = 0; r < MLP_BR; ++r) acc[r] += dcur[cur][r][j] * wv;
                }
#pragma unroll
                for (int r = 0; r < MLP_BR; ++r) {
                    const int row = row0 + r;
                    if (row >= a.B) continue;
                    if (k < a.d1) a.dx1[(size_t)row * a.d1 + k] = acc[r];
                    else a.dx2[(size_t)row * a.d2 + (k - a.d1)] = acc[r];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// out[i] = sum over workgroup slabs, fixed order; the outputs are scattered to per-layer arrays by the host
// through pointer/offset pairs
struct MlpRedArgs { const float* slab; int nslab, stride; float* dW[MLP_MAXL]; float* db[MLP_MAXL]; int off[MLP_MAXL + 1]; int nw[MLP_MAXL]; int nl;
                    const float* gscale; };        // device scalar every sum is multiplied with (null: 1)
__global__ __launch_bounds__(256) void k_mlp_reduce(MlpRedArgs a) {
    __shared__ float red[4][64];
    const int total = a.off[a.nl];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    float v = 0.f;
    if (i < total)
        for (int s = g; s < a.nslab; s += 4) v += a.slab[(size_t)s * a.stride + i];      // four interleaved partial sums
    red[g][c] = v;
    __syncthreads();
    if (g == 0 && i < total) {
        v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        if (a.gscale) v *= a.gscale[0];
        int l = 0;
        while (l + 1 < a.nl && i >= a.off[l + 1]) ++l;
        const int loc = i - a.off[l];
        if (loc < a.nw[l]) a.dW[l][loc] = v;
        else if (a.db[l]) a.db[l][loc - a.nw[l]] = v;
    }
}

static int mlp_fill(MlpArgs& a, const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                    const float* const* W, const float* const* b, float* const* act) {
    memset(&a, 0, sizeof(a));
    BMP_REQUIRE(nl >= 1 && nl <= MLP_MAXL && B > 0 && x1 && d1 > 0 && d2 >= 0 && (d2 == 0 || x2));
    BMP_REQUIRE(dims[0] == d1 + d2 && dims[0] <= MLP_MAXIN);
    for (int l = 0; l < nl; ++l) {
        BMP_REQUIRE(dims[l + 1] > 0 && dims[l + 1] <= MLP_MAXW && W[l] && act[l]);
        a.W[l] = W[l]; a.b[l] = b ? b[l] : nullptr; a.act[l] = act[l];
    }
    for (int l = 0; l <= nl; ++l) a.dims[l] = dims[l];
    a.x1 = x1; a.x2 = x2; a.d1 = d1; a.d2 = d2; a.B = B; a.nl = nl;
    return 0;
}

// Forward.  x = [x1 (B x d1) | x2 (B x d2)] (x2 may be NULL with d2 = 0); dims[0..nl] = layer widths, dims[0] = d1 + d2;
// W, b, act: HOST arrays of nl device pointers (W[l] [dims[l+1] x dims[l]], b[l] [dims[l+1]] or NULL,
// act[l] [B x dims[l+1]] = relu(...) for l < nl-1, logits for the last).
extern "C" int bmp_mlp_fwd(const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                           const float* const* W, const float* const* b, float* const* act, hipStream_t st) {
    MlpArgs a;
    int rc = mlp_fill(a, x1, d1, x2, d2, B, nl, dims, W, b, act);
    if (rc) return rc;
    const size_t wt_bytes = (size_t)dims[0] * (dims[1] + 1) * sizeof(float);          // <= 1024 x 65 floats
    if (int rc_attr = bmp_lds_attr((const void*)k_mlp_fwd, (size_t)(160 * 1024 - 65536))) return rc_attr;
    BMP_REQUIRE(wt_bytes + 65536 <= 160 * 1024);
    hipLaunchKernelGGL(k_mlp_fwd, dim3((B + MLP_FR - 1) / MLP_FR), dim3(256), wt_bytes, st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t bmp_mlp_bwd_ws_floats(int B, int nl, const int* dims) {
    size_t per = 0;
    for (int l = 0; l < nl; ++l) per += (size_t)dims[l + 1] * (dims[l] + 1);
    return per * ((B + MLP_BR - 1) / MLP_BR);
}

// Backward from dy [B x dims[nl]]: dx1, dx2, and per layer dW[l], db[l] (HOST arrays of device pointers; db[l] may be NULL).
extern "C" int bmp_mlp_bwd(const float* dy, const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                           const float* const* W, float* const* act, float* dx1, float* dx2, float* const* dW,
                           float* const* db, float* ws, size_t ws_floats, hipStream_t st, hipStream_t st_w) {
    MlpArgs a;
    int rc = mlp_fill(a, x1, d1, x2, d2, B, nl, dims, W, nullptr, act);
    if (rc) return rc;
    BMP_REQUIRE(dy && dx1 && (d2 == 0 || dx2) && dW && ws && ws_floats >= bmp_mlp_bwd_ws_floats(B, nl, dims));
    const int nwg = (B + MLP_BR - 1) / MLP_BR;
    MlpRedArgs r; memset(&r, 0, sizeof(r));
    int off = 0;
    for (int l = 0; l < nl; ++l) {
        BMP_REQUIRE(dW[l] != nullptr);
        r.off[l] = off; r.nw[l] = dims[l + 1] * dims[l]; r.dW[l] = dW[l]; r.db[l] = db ? db[l] : nullptr;
        off += dims[l + 1] * (dims[l] + 1);
    }
    r.off[nl] = off; r.nl = nl; r.slab = ws; r.nslab = nwg; r.stride = off;
    a.dy = dy; a.dx1 = dx1; a.dx2 = dx2; a.slab = ws; a.slab_stride = off;
    // the weight-gradient partials and their fold are off the backward chain: stream_w (bmp.h, "stream_w"); the chain's launch
    // only hands the input gradients on
    if (!st_w || st_w == st) {
        hipLaunchKernelGGL((k_mlp_bwd<true, true>), dim3(nwg), dim3(256), 0, st, a);
        BMP_LAUNCH_CHECK();
        st_w = st;
    } else {
        if ((rc = bmp_fork_to(st, st_w))) return rc;          // dy is ready on `st`
        hipLaunchKernelGGL((k_mlp_bwd<true, false>), dim3(nwg), dim3(256), 0, st, a);
        BMP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_mlp_bwd<false, true>), dim3(nwg), dim3(256), 0, st_w, a);
        BMP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_mlp_reduce, dim3((off + 63) / 64), dim3(256), 0, st_w, r);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---- the whole head of a training step in one launch: MLP forward, sigmoid cross entropy, its gradient, MLP backward down
// to the input rows (Classifier(predictor, lossfun=F.sigmoid_cross_entropy), train_ddi_modify.py:284-286, on
// models/mlp.py:40-45).  As five launches (bmp_mlp_fwd, bmp_sce_fwd, the framework's fill of the root gradient, bmp_sce_bwd,
// bmp_mlp_bwd) these few hundred multiply-adds per row were ~100 us of the step's dependent chain.  Same arithmetic and
// summation orders per row as those launches.  dx is the gradient of the MEAN loss for d loss = 1 (the caller scales it, and
// bmp_mlp_bwd_w the weight gradients, with the gradient that arrives at the loss).
struct HeadArgs {
    const int* t;          // labels [B x C] (C = dims[nl]), -1 = not counted
    float* dy;             // [B x C] gradient of the mean loss w.r.t. the logits
    float* part;           // [n workgroups] loss numerators of the workgroups' rows
    unsigned* ticket;      // zero before the launch, zero again after it
    float* loss; float* sums;       // mean loss; numerator | count
};

__global__ __launch_bounds__(256) void k_mlp_sce(MlpArgs a, HeadArgs h) {
    __shared__ float xin[MLP_BR][MLP_MAXIN];              // layer-0 input rows
    __shared__ float hid[MLP_MAXL][MLP_BR][MLP_MAXW];     // layer outputs (relu; the last: logits)
    __shared__ float dcur[2][MLP_BR][MLP_MAXW];
    __shared__ float redf[4];
    __shared__ int redi[4];
    __shared__ int last_flag;
    // ALL weights in LDS: layer 0 transposed, wt[k * (no + 1) + j]; the layers behind it as they are, rows padded by one float,
    // wl[j * (ni + 1) + k].  (Read from global memory inside the layers' loops -- a dependent load per multiply-add, 32 of them
    // per thread in the walk back to the input rows alone -- the launch took 47 us for its few hundred multiply-adds per row.)
    extern __shared__ float wt[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row0 = blockIdx.x * MLP_BR;
    const int in0 = a.dims[0], C = a.dims[a.nl];
    int woff[MLP_MAXL];
    {
        int o = a.dims[0] * (a.dims[1] + 1);
        woff[0] = 0;
        for (int l = 1; l < a.nl; ++l) { woff[l] = o; o += a.dims[l + 1] * (a.dims[l] + 1); }
    }
    for (int idx = tid; idx < MLP_BR * in0; idx += 256) {
        const int r = idx / in0, k = idx % in0, row = row0 + r;
        float v = 0.f;
        if (row < a.B) v = k < a.d1 ? a.x1[(size_t)row * a.d1 + k] : a.x2[(size_t)row * a.d2 + (k - a.d1)];
        xin[r][k] = v;
    }
    {
        const int ni = a.dims[0], no = a.dims[1];
        const float* __restrict__ W = a.W[0];
        for (int idx = tid; idx < no * ni; idx += 256) wt[(idx % ni) * (no + 1) + idx / ni] = W[idx];
    }
    for (int l = 1; l < a.nl; ++l) {
        const int ni = a.dims[l], no = a.dims[l + 1];
        const float* __restrict__ W = a.W[l];
        float* wl = wt + woff[l];
        for (int idx = tid; idx < no * ni; idx += 256) wl[(idx / ni) * (ni + 1) + idx % ni] = W[idx];
    }
    // labels that count (every workgroup counts all of them: an integer, the same in every workgroup)
    int cnt = 0;
    for (int i = tid; i < a.B * C; i += 256) cnt += h.t[i] != -1 ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m);
    if (lane == 0) redi[wv] = cnt;
    __syncthreads();
    const int n_valid = (redi[0] + redi[1]) + (redi[2] + redi[3]);
    const float inv_n = 1.f / (n_valid > 1 ? (float)n_valid : 1.f);
    // ---- forward ----
    for (int l = 0; l < a.nl; ++l) {
        const int ni = a.dims[l], no = a.dims[l + 1];
        const float* __restrict__ bb = a.b[l];
        const bool last = l == a.nl - 1;
        for (int idx = tid; idx < MLP_BR * no; idx += 256) {
            const int r = idx / no, j = idx % no;
            float acc = bb ? bb[j] : 0.f;
            if (l == 0) {
                const float* x = xin[r];
                const float* w = wt + j;
#pragma unroll 8
                for (int k = 0; k < ni; ++k) acc += x[k] * w[k * (no + 1)];
            } else {
                const float* x = hid[l - 1][r];
                const float* w = wt + woff[l] + j * (ni + 1);
#pragma unroll 8
                for (int k = 0; k < ni; ++k) acc += x[k] * w[k];
            }
            if (!last) acc = acc > 0.f ? acc : 0.f;
            hid[l][r][j] = acc;
            if (row0 + r < a.B) a.act[l][(size_t)(row0 + r) * no + j] = acc;
        }
        __syncthreads();
    }
    // ---- loss of the workgroup's rows, gradient w.r.t. the logits ----
    float lsum = 0.f;
    for (int idx = tid; idx < MLP_BR * C; idx += 256) {
        const int r = idx / C, j = idx % C, row = row0 + r;
        float g = 0.f;
        if (row < a.B) {
            const int ti = h.t[(size_t)row * C + j];
            if (ti != -1) {
                const float yi = hid[a.nl - 1][r][j];
                lsum += (yi > 0.f ? yi : 0.f) + log1pf(expf(-fabsf(yi))) - (float)ti * yi;
                g = (1.f / (1.f + expf(-yi)) - (float)ti) * inv_n;
            }
            h.dy[(size_t)row * C + j] = g;
        }
        dcur[0][r][j] = g;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m);
    if (lane == 0) redf[wv] = lsum;
    __syncthreads();
    if (tid == 0) h.part[blockIdx.x] = (redf[0] + redf[1]) + (redf[2] + redf[3]);
    // ---- backward down to the input rows ----
    int cur = 0;
    for (int l = a.nl - 1; l >= 0; --l) {
        const int ni = a.dims[l], no = a.dims[l + 1];
        if (l > 0) {
            const float* wl = wt + woff[l];
            for (int idx = tid; idx < MLP_BR * ni; idx += 256) {
                const int r = idx / ni, k = idx % ni;
                float acc = 0.f;
#pragma unroll 8
                for (int j = 0; j < no; ++j) acc += dcur[cur][r][j] * wl[j * (ni + 1) + k];
                dcur[cur ^ 1][r][k] = hid[l - 1][r][k] > 0.f ? acc : 0.f;
            }
        } else {
            for (int k = tid; k < ni; k += 256) {         // a thread owns input column k for all rows of the workgroup
                float acc[MLP_BR];
#pragma unroll
                for (int r = 0; r < MLP_BR; ++r) acc[r] = 0.f;
                const float* w = wt + k * (no + 1);
#pragma unroll 8
                for (int j = 0; j < no; ++j) {
                    const float wv0 = w[j];
#pragma unroll
                    for (int r = 0; r < MLP_BR; ++r) acc[r] += dcur[cur][r][j] * wv0;
                }
#pragma unroll
                for (int r = 0; r < MLP_BR; ++r) {
                    const int row = row0 + r;
                    if (row >= a.B) continue;
                    if (k < a.d1) a.dx1[(size_t)row * a.d1 + k] = acc[r];
                    else a.dx2[(size_t)row * a.d2 + (k - a.d1)] = acc[r];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- the workgroup that finishes last folds the loss numerators, in workgroup order ----
    if (tid == 0) {
        __threadfence();
        const unsigned tk = atomicAdd(h.ticket, 1u);
        last_flag = tk == gridDim.x - 1;
    }
    __syncthreads();
    if (last_flag && tid == 0) {
        __threadfence();
        float num = 0.f;
        for (unsigned w = 0; w < gridDim.x; ++w) num += __hip_atomic_load(h.part + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h.sums[0] = num; h.sums[1] = (float)n_valid;
        h.loss[0] = num * inv_n;
        *h.ticket = 0u;
    }
}

extern "C" size_t bmp_mlp_sce_ws_floats(int B) { return (size_t)(B + MLP_BR - 1) / MLP_BR; }

// act as bmp_mlp_fwd (the last one: the logits); t [B x dims[nl]] int32; dy [B x dims[nl]]; dx1 / dx2: the gradient of the
// mean loss w.r.t. the input rows; loss [1], sums [2] as bmp_sce_fwd; part: bmp_mlp_sce_ws_floats(B) floats; ticket: one
// unsigned that is zero before the call (and zero again when the launch has finished).
extern "C" int bmp_mlp_sce_fwdbwd(const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                                  const float* const* W, const float* const* b, float* const* act, const int* t, float* dy,
                                  float* dx1, float* dx2, float* loss, float* sums, float* part, unsigned* ticket,
                                  hipStream_t st) {
    MlpArgs a;
    int rc = mlp_fill(a, x1, d1, x2, d2, B, nl, dims, W, b, act);
    if (rc) return rc;
    BMP_REQUIRE(t && dy && dx1 && (d2 == 0 || dx2) && loss && sums && part && ticket);
    a.dx1 = dx1; a.dx2 = dx2;
    HeadArgs h{t, dy, part, ticket, loss, sums};
    size_t wt_bytes = (size_t)dims[0] * (dims[1] + 1) * sizeof(float);
    for (int l = 1; l < nl; ++l) wt_bytes += (size_t)dims[l + 1] * (dims[l] + 1) * sizeof(float);
    if (int rc_attr = bmp_lds_attr((const void*)k_mlp_sce, (size_t)(160 * 1024 - 49152))) return rc_attr;
    BMP_REQUIRE(wt_bytes + 49152 <= 160 * 1024);
    hipLaunchKernelGGL(k_mlp_sce, dim3((B + MLP_BR - 1) / MLP_BR), dim3(256), wt_bytes, st, a, h);
    BMP_LAUNCH_CHECK();
    return 0;
}

// The weight / bias gradients of bmp_mlp_bwd alone (dy as written by bmp_mlp_sce_fwdbwd), every one multiplied with the device
// scalar gscale[0] (NULL: 1): per-workgroup partials and their fixed-order fold, all on `st`.
extern "C" int bmp_mlp_bwd_w(const float* dy, const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                             const float* const* W, float* const* act, float* const* dW, float* const* db, const float* gscale,
                             float* ws, size_t ws_floats, hipStream_t st) {
    MlpArgs a;
    int rc = mlp_fill(a, x1, d1, x2, d2, B, nl, dims, W, nullptr, act);
    if (rc) return rc;
    BMP_REQUIRE(dy && dW && ws && ws_floats >= bmp_mlp_bwd_ws_floats(B, nl, dims));
    const int nwg = (B + MLP_BR - 1) / MLP_BR;
    MlpRedArgs r; memset(&r, 0, sizeof(r));
    int off = 0;
    for (int l = 0; l < nl; ++l) {
        BMP_REQUIRE(dW[l] != nullptr);
        r.off[l] = off; r.nw[l] = dims[l + 1] * dims[l]; r.dW[l] = dW[l]; r.db[l] = db ? db[l] : nullptr;
        off += dims[l + 1] * (dims[l] + 1);
    }
    r.off[nl] = off; r.nl = nl; r.slab = ws; r.nslab = nwg; r.stride = off; r.gscale = gscale;
    a.dy = dy; a.slab = ws; a.slab_stride = off;
    hipLaunchKernelGGL((k_mlp_bwd<false, true>), dim3(nwg), dim3(256), 0, st, a);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mlp_reduce, dim3((off + 63) / 64), dim3(256), 0, st, r);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---- sigmoid cross entropy: loss = mean over elements with t != -1 of softplus(y) - t*y ----
// One workgroup, fixed summation order: every wave folds its lanes with shuffles (no barrier), the 16 wave sums meet in LDS
// once.  sums[0] = loss numerator, sums[1] = count; loss[0] = sums[0] / max(count, 1).  (A ten-level LDS tree over 1024
// threads spent 25 us in its barriers: more than the co-attention's MLP forward.)
__global__ __launch_bounds__(1024) void k_sce_fwd(const float* __restrict__ y, const int* __restrict__ t, int n, float* loss,
                                                  float* sums) {
    __shared__ float s0[16], s1[16];
    float acc = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int ti = t[i];
        if (ti != -1) {
            const float yi = y[i];
            const float sp = (yi > 0.f ? yi : 0.f) + log1pf(expf(-fabsf(yi)));      // softplus, stable
            acc += sp - (float)ti * yi;
            cnt += 1.f;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { acc += __shfl_xor(acc, m); cnt += __shfl_xor(cnt, m); }
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = acc; s1[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, c0 = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { a0 += s0[w]; c0 += s1[w]; }
        sums[0] = a0; sums[1] = c0;
        loss[0] = a0 / (c0 > 1.f ? c0 : 1.f);
    }
}

__global__ __launch_bounds__(256) void k_sce_bwd(const float* __restrict__ y, const int* __restrict__ t, int n,
                                                 const float* __restrict__ sums, const float* __restrict__ gout,
                                                 float* __restrict__ dy) {
    const float scale = gout[0] / (sums[1] > 1.f ? sums[1] : 1.f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int ti = t[i];
        float g = 0.f;
        if (ti != -1) g = (1.f / (1.f + expf(-y[i])) - (float)ti) * scale;
        dy[i] = g;
    }
}

extern "C" int bmp_sce_fwd(const float* y, const int* t, int n, float* loss, float* sums, hipStream_t st) {
    BMP_REQUIRE(n > 0 && y && t && loss && sums);
    hipLaunchKernelGGL(k_sce_fwd, dim3(1), dim3(1024), 0, st, y, t, n, loss, sums);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_sce_bwd(const float* y, const int* t, int n, const float* sums, const float* gout, float* dy,
                           hipStream_t st) {
    BMP_REQUIRE(n > 0 && y && t && sums && gout && dy);
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_sce_bwd, dim3(blocks), dim3(256), 0, st, y, t, n, sums, gout, dy);
    BMP_LAUNCH_CHECK();
    return 0;
}
