// Fused GGNN propagation step for NARROW hidden widths (d = 32) on gfx950: the same step as bmp_fused.hip
// (message = per-bond-type gather-sum + linear, GRU node update; models/ggnn.py:215-263) with the same operands, outputs
// and weight layouts, laid out for a width at which one 32 x 32 MFMA block IS a whole row of the result.
//
// d = 32 is the width of every model the reference publishes figures for (--fp-hidden-dim=32 --conv-layers=8
// --weight-tying=False: DDI.md:6, RECORD.txt:196-202,246-251).  A step is then 26 d^2 = 27 kflop per atom: a 128-row tile
// holds 3.4 MFLOP, microseconds of matrix work, and the operator chain it replaces (gather, message row GEMM, three gate
// row GEMMs with their epilogues: five launches forward, eight backward) is bound by launch latency and by the HBM round
// trips of its intermediates, not by arithmetic.
//
//   workgroup = 256 threads = 4 waves, one 128-row tile; wave w owns the 32-row block w in EVERY phase -- its rows of the
//   gather (2 threads per row, 16 columns each: the wave's 64 lanes are exactly its 32 rows), the A rows of its MFMAs, its
//   epilogue rows -- so the waves of a tile only meet where a phase reads the WHOLE tile: after the tile load (forward) and
//   before the transposed gather of dm (backward).  Everything else is wave-local (LDS operations of one wave complete in
//   order; a compiler fence is all the synchronisation there is), and the waves drift apart freely.
//   LDS = two [128 x 36] f32 tiles + the tile's CSR = 48 KB: three workgroups per CU, whose gathers, epilogues and MFMAs
//   overlap each other.  Weights (K4-packed, 44 KB in all) stream from L2 into MFMA B registers as in the wide kernels.
//   A tile table (encoder layout, bmp/enclayout.py) gives tiles of 1..4 live blocks: the dead blocks' waves leave after the
//   first barrier.
#include "bmp_tile.h"

#define FS_R 128
#define FS_NT 256
#define FS_LOFF(reg) ((((reg) & 3) + 8 * ((reg) >> 2)) * LD)
#define FS_FOR_ACC _Pragma("unroll") for (int reg = 0; reg < 16; ++reg)
// wave-local ordering of LDS traffic (write by some lanes, read by others of the SAME wave)
#define FS_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// This wave's 32 rows of a per-bond-type neighbour gather: dst[row, :] = sum over the CSR entries of `row` with type e of
// val * src[col_local, :].  Two lanes per row, D / 2 columns each.  Returns the row's weighted degree for that type.
template <int D>
__device__ __forceinline__ float fs_gather(const float* src, float* dst, int LD, const int* ptr, const int* col, const float* val,
                                           int row0, int row, int q, int e, int* tmask) {
    constexpr int F = D / 8;                  // float4 per lane
    f32x4 acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wd = 0.f;
    const int e0 = ptr[row], e1 = ptr[row + 1];
    for (int ed = e0; ed < e1; ++ed) {
        const int cv = col[ed];
        *tmask |= 1 << (cv & 3);
        if ((cv & 3) == e) {
            const float v = val[ed];
            const float* s = src + ((cv >> 2) - row0) * LD + q * (D / 2);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += *(const f32x4*)(s + 4 * f) * v;
            wd += v;
        }
    }
    float* o = dst + row * LD + q * (D / 2);
#pragma unroll
    for (int f = 0; f < F; ++f) *(f32x4*)(o + 4 * f) = acc[f];
    return wd;
}
#define FS_GATHER(srcT, dstT, e) (csr_lds ? fs_gather<D>(srcT, dstT, LD, rptr, ecol, evalv, row0, grow, gq, e, &tmask) \
                                          : fs_gather<D>(srcT, dstT, LD, a.ptr + row0, a.col, a.val, row0, grow, gq, e, &tmask))

// bond types present among this wave's rows (bit e), from the lanes' masks after a pass that walked every entry
__device__ __forceinline__ int fs_wave_types(int tmask) {
    int m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= (__ballot((tmask >> e) & 1) != 0ull) ? (1 << e) : 0;
    return m;
}

// Row-major 16-byte access to this wave's 32 rows: slot v of a lane = float4 (row v * (256 / D) + lane / (D / 4), column
// lane % (D / 4)); all slots of an array share one voffset (AccBuf, bmp_tile.h).
template <int D, int LDP>
__device__ __forceinline__ AccBuf fs_rm_buf(const float* base, int tile_row0, int wrow0, int lane) {
    AccBuf b;
    b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)tile_row0 * LDP), 0, 0x7FFFFFFF, 0x00020000);
    b.vo = ((wrow0 + lane / (D / 4)) * LDP + 4 * (lane % (D / 4))) * 4;
    return b;
}
template <int D, int LDP>
__device__ __forceinline__ f32x4 fs_rm_ld(const AccBuf& b, int v, int coff = 0) {
    const int so = (v * (256 / D) * LDP + coff) * 4;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.rs, b.vo, so, 0));
}
template <int D, int LDP>
__device__ __forceinline__ void fs_rm_st(const AccBuf& b, int v, f32x4 x, int coff = 0) {
    const int so = (v * (256 / D) * LDP + coff) * 4;          // in the voffset: see rm_st (bmp_tile.h) for the store hazard
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), b.rs, b.vo + so, 0, 0);
}

template <int D, bool FIRST, bool VAR, bool SAVE>
__global__ __launch_bounds__(FS_NT) void k_ggnn_step_fwd_s(StepArgs a) {
    static_assert(D == 32, "one 32-column MFMA block per row");
    constexpr int LD = D + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Hs = lds;                         // [128 x LD]  h tile (whole step)
    float* As = lds + FS_R * LD;             // [128 x LD]  AGG_e -> M -> r*h   (each wave: its own 32 rows)
    float* wds = As + FS_R * LD;             // [128 x 4]   weighted degree per bond type
    int* rptr = (int*)(wds + FS_R * 4);      // [132]
    int* ecol = rptr + 132;                  // [FZ_ECAP]
    float* evalv = (float*)(ecol + FZ_ECAP); // [FZ_ECAP]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tile = blockIdx.x + a.tile0;
    const int row0 = VAR ? a.mt_row0[tile] : tile * FS_R;
    const int nblk = VAR ? a.mt_nblk[tile] : 4;
    const int nrows = nblk * 32;
    if (VAR && a.tile_stride > nrows) {      // fixed-stride tile table (bmp_tile.h, StepArgs::tile_stride): the blocks without atoms
        const int e = row0 + a.tile_stride;
        fz_clear_rows<FS_NT>(a.hout, D, row0 + nrows, e, tid);
        fz_clear_rows<FS_NT>(a.m, D, row0 + nrows, e, tid);
        fz_clear_rows<FS_NT>(a.rz, 2 * D, row0 + nrows, e, tid);
        fz_clear_rows<FS_NT>(a.c, D, row0 + nrows, e, tid);
    }
    const int col = l31;
    const int wrow0 = w * 32;
    const int lrow = wrow0 + 4 * hi;         // this lane's row for accumulator register 0
    const int grow = wrow0 + (lane >> 1), gq = lane & 1;      // this lane's row and column half in the gathers
    const int rot = (tile * 8) % D;
    const float* Hw = Hs + (wrow0 + l31) * LD + 4 * hi;
    const float* Aw = As + (wrow0 + l31) * LD + 4 * hi;
    float* Hl = Hs + lrow * LD + col;
    float* Al = As + lrow * LD + col;
    constexpr bool save = SAVE;

    for (int idx = tid; idx < nrows * (D / 4); idx += FS_NT) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        *(f32x4*)(Hs + r * LD + 4 * c4) = *(const f32x4*)(a.h + (size_t)(row0 + r) * D + 4 * c4);
    }
    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows, FS_NT);
    __syncthreads();                         // the only workgroup barrier: h and the CSR are in place
    if (w >= nblk) return;                   // a short tile: this block has no rows

    int tmask = 0, types = 0;
    // ---- message: m = sum_e AGG_e . W_e + wdeg_e * b_e   (models/ggnn.py:223-242) ----
    f32x16 acc_m[1][1];
    zero_acc(acc_m[0]);
    for (int e = 0; e < 4; ++e) {
        const float* const Bp[1] = {a.WT + (size_t)(e * D + 4 * hi) * D + 4 * col};
        const int ldw[1] = {D};
        BPre<1> pre;
        tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
        const float wd = FS_GATHER(Hs, As, e);
        if (gq == 0) wds[grow * 4 + e] = wd;
        if (e == 0) types = fs_wave_types(tmask);         // the first pass walks every entry of the wave's rows
        FS_WSYNC();
        if ((types >> e) & 1) tile_mma<1, 1, 1>(acc_m, Aw, LD, Bp, ldw, D, rot, &pre);
        FS_WSYNC();
    }
    constexpr int NG = FIRST ? 2 : 3;                     // first call after reset: z and c only
    int ldwg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ldwg[g] = 3 * D;
    const float* const base_h = a.AT + (size_t)(4 * hi) * 3 * D + 4 * col + (FIRST ? 4 * D : 0);
    const float* const base_m = a.AT + (size_t)(D + 4 * hi) * 3 * D + 4 * col + (FIRST ? 4 * D : 0);
    const float* Bh[NG]; const float* Bm[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { Bh[g] = base_h + 4 * D * g; Bm[g] = base_m + 4 * D * g; }
    BPre<NG> pre_h;
    tile_b_prefetch<NG>(pre_h, (const float* const (&)[NG])Bh, (const int (&)[NG])ldwg, D, rot);
    {   // m -> LDS (A operand of the gates) and HBM (saved for the backward)
        const AccBuf mo = acc_buf<D>(a.m, row0, lrow, col);
        float be[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) be[e] = a.bE[e * D + col];
        FS_FOR_ACC {
            const int r = lrow + (reg & 3) + 8 * (reg >> 2);
            const f32x4 wd4 = *(const f32x4*)(wds + r * 4);
            const float v = __builtin_fmaf(wd4[3], be[3], __builtin_fmaf(wd4[2], be[2], __builtin_fmaf(wd4[1], be[1],
                                           __builtin_fmaf(wd4[0], be[0], acc_m[0][0][reg]))));
            Al[FS_LOFF(reg)] = v;
            if (save) acc_st<D>(mo, 0, reg, v);
        }
    }
    FS_WSYNC();
    // ---- gates: [r | z | c~] = [h, m] . AT   (chainer StatefulGRU) ----
    f32x16 acc_g[3][1];
    zero_acc(acc_g[0]); zero_acc(acc_g[1]); zero_acc(acc_g[2]);
    {
        f32x16 gg[NG][1];
#pragma unroll
        for (int g = 0; g < NG; ++g) zero_acc(gg[g]);
        BPre<NG> pre_m;
        tile_b_prefetch<NG>(pre_m, (const float* const (&)[NG])Bm, (const int (&)[NG])ldwg, D, rot);
        tile_mma<NG, 1, 1>(gg, Hw, LD, (const float* const (&)[NG])Bh, (const int (&)[NG])ldwg, D, rot, &pre_h);
        tile_mma<NG, 1, 1>(gg, Aw, LD, (const float* const (&)[NG])Bm, (const int (&)[NG])ldwg, D, rot, &pre_m);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc_g[g + (FIRST ? 1 : 0)][0] = gg[g][0];
    }
    const float br = a.b[col], bz = a.b[D + col], bcn = a.b[2 * D + col];
    {
        const AccBuf rzo = acc_buf<2 * D>(a.rz, row0, lrow, col);
        FS_FOR_ACC {
            const float zv = bmp_sigmoid(acc_g[1][0][reg] + bz);
            acc_g[1][0][reg] = zv;
            if (save) acc_st<2 * D>(rzo, 0, reg, zv, D);
            if (!FIRST) {
                const float rv = bmp_sigmoid(acc_g[0][0][reg] + br);
                acc_g[0][0][reg] = rv;
                if (save) acc_st<2 * D>(rzo, 0, reg, rv, 0);
            }
        }
    }
    if (!FIRST) {
        const float* const Bu[1] = {a.UcT + (size_t)(4 * hi) * D + 4 * col};
        const int ldu[1] = {D};
        BPre<1> pre_u;
        tile_b_prefetch<1>(pre_u, Bu, ldu, D, rot);
        FS_WSYNC();                          // the wave is done reading M
        FS_FOR_ACC { Al[FS_LOFF(reg)] = acc_g[0][0][reg] * Hl[FS_LOFF(reg)]; }      // r * h
        FS_WSYNC();
        f32x16 gc[1][1];
        gc[0][0] = acc_g[2][0];
        tile_mma<1, 1, 1>(gc, Aw, LD, Bu, ldu, D, rot, &pre_u);
        acc_g[2][0] = gc[0][0];
    }
    {   // ---- h' = z*c + (1-z)*h  (first call: z*c) ----
        const AccBuf co = acc_buf<D>(a.c, row0, lrow, col);
        const AccBuf ho = acc_buf<D>(a.hout, row0, lrow, col);
        FS_FOR_ACC {
            const float cv = bmp_tanh(acc_g[2][0][reg] + bcn);
            const float zv = acc_g[1][0][reg];
            float hn = zv * cv;
            if (!FIRST) hn = __builtin_fmaf(1.f - zv, Hl[FS_LOFF(reg)], hn);
            if (save) acc_st<D>(co, 0, reg, cv);
            acc_st<D>(ho, 0, reg, hn);
        }
    }
}

// Backward-data of one step for one tile (see k_ggnn_step_bwd, bmp_fused.hip): dh and gda [N x 7D] = [G_0..G_3 | da_r | da_z | da_c].
template <int D, bool FIRST, bool VAR>
__global__ __launch_bounds__(FS_NT) void k_ggnn_step_bwd_s(StepArgs a) {
    static_assert(D == 32, "one 32-column MFMA block per row");
    constexpr int LD = D + 4;
    constexpr int F4 = D / 4;                // float4 per row
    constexpr int NV = D / 8;                // float4 slots per lane of a 32-row block (64 lanes)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                         // [128 x LD]  da_c -> d(r*h) -> da_r -> dm
    float* Ys = lds + FS_R * LD;             // [128 x LD]  da_z -> G_e (transposed gather of dm) -> dh
    int* rptr = (int*)(Ys + FS_R * LD + FS_R * 4);
    int* ecol = rptr + 132;
    float* evalv = (float*)(ecol + FZ_ECAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tile = blockIdx.x + a.tile0;
    const int row0 = VAR ? a.mt_row0[tile] : tile * FS_R;
    const int nblk = VAR ? a.mt_nblk[tile] : 4;
    const int nrows = nblk * 32;
    if (VAR && a.tile_stride > nrows) {
        const int e = row0 + a.tile_stride;
        fz_clear_rows<FS_NT>(a.dh, D, row0 + nrows, e, tid);
        fz_clear_rows<FS_NT>(a.gda, 7 * D, row0 + nrows, e, tid);
    }
    const int col = l31;
    const int wrow0 = w * 32;
    const int lrow = wrow0 + 4 * hi;
    const int grow = wrow0 + (lane >> 1), gq = lane & 1;
    const int rot = (tile * 8) % D;
    const float* Xw = Xs + (wrow0 + l31) * LD + 4 * hi;
    const float* Yw = Ys + (wrow0 + l31) * LD + 4 * hi;
    float* Xl = Xs + lrow * LD + col;
    float* Yl = Ys + lrow * LD + col;
    constexpr bool first = FIRST;
#define RM_ROW(v) (wrow0 + (v) * (64 / F4) + lane / F4)
#define RM_C4(v) (lane % F4)
#define RM_LDS(T, v) (*(f32x4*)((T) + RM_ROW(v) * LD + 4 * RM_C4(v)))
    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows, FS_NT);
    if (w >= nblk) {                         // a short tile: this block has no rows; it has staged its share of the CSR, meets
        __syncthreads();                     // the first barrier and leaves (the barrier stops counting waves that ended)
        return;
    }
    const AccBuf b_g = fs_rm_buf<D, D>(a.dhout, row0, wrow0, lane), b_c = fs_rm_buf<D, D>(a.c, row0, wrow0, lane);
    const AccBuf b_h = fs_rm_buf<D, D>(a.h, row0, wrow0, lane), b_rz = fs_rm_buf<D, 2 * D>(a.rz, row0, wrow0, lane);
    const AccBuf b_o = fs_rm_buf<D, 7 * D>(a.gda, row0, wrow0, lane), b_dh = fs_rm_buf<D, D>(a.dh, row0, wrow0, lane);

    // ---- da_c = dh' z (1 - c^2) -> X ; da_z = dh' (c - h) z (1 - z) -> Y ; ex = dh' (1 - z): the direct part of dh ----
    f32x4 ex[NV];
    {
        const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
        f32x4 g4[NV], z4[NV], c4[NV], h4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            g4[v] = fs_rm_ld<D, D>(b_g, v);
            z4[v] = fs_rm_ld<D, 2 * D>(b_rz, v, D);
            c4[v] = fs_rm_ld<D, D>(b_c, v);
            if (!first) h4[v] = fs_rm_ld<D, D>(b_h, v);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 gz = g4[v] * z4[v];
            const f32x4 dac = gz * (one - c4[v] * c4[v]);
            f32x4 dz = gz * (one - z4[v]);
            if (first) { dz *= c4[v]; ex[v] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            else { dz *= (c4[v] - h4[v]); ex[v] = g4[v] - gz; }
            RM_LDS(Xs, v) = dac;
            RM_LDS(Ys, v) = dz;
            fs_rm_st<D, 7 * D>(b_o, v, dac, 6 * D);
            fs_rm_st<D, 7 * D>(b_o, v, dz, 5 * D);
        }
    }
    const float* const Ac_h = a.A + (size_t)(2 * D + 4 * hi) * 2 * D + 4 * col;
    const float* const Az_h = a.A + (size_t)(D + 4 * hi) * 2 * D + 4 * col;
    const float* const Ar_h = a.A + (size_t)(4 * hi) * 2 * D + 4 * col;
    const int ld2[2] = {2 * D, 2 * D};
    const float* const Bc[2] = {Ac_h, Ac_h + 4 * D};
    BPre<2> pre_c;
    tile_b_prefetch<2>(pre_c, Bc, ld2, D, rot);
    __syncthreads();                         // whole workgroup: the staged CSR is visible (X / Y so far are wave-local)

    f32x16 acc_x[2][1];                      // [0] = dh, [1] = dm
    zero_acc(acc_x[0]); zero_acc(acc_x[1]);
    tile_mma<2, 1, 1>(acc_x, Xw, LD, Bc, ld2, D, rot, &pre_c);                 // [dh | dm] += da_c . A_c
    if (!first) {
        f32x16 acc_d[1][1];                  // d(r*h) = da_c . U
        zero_acc(acc_d[0]);
        {
            const float* const Bu[1] = {a.Uc + (size_t)(4 * hi) * D + 4 * col};
            const int ldu[1] = {D};
            tile_mma<1, 1, 1>(acc_d, Xw, LD, Bu, ldu, D, rot);
        }
        FS_WSYNC();                          // the wave is done with da_c in X
        FS_FOR_ACC { Xl[FS_LOFF(reg)] = acc_d[0][0][reg]; }
        f32x4 r4[NV], h4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) { r4[v] = fs_rm_ld<D, 2 * D>(b_rz, v); h4[v] = fs_rm_ld<D, D>(b_h, v); }
        FS_WSYNC();
        // da_r = d(r*h) h r (1-r) -> X (in place) ; ex += d(r*h) r
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 drh = RM_LDS(Xs, v);
            const f32x4 dr = drh * r4[v];
            const f32x4 dar = dr * h4[v] * ((f32x4){1.f, 1.f, 1.f, 1.f} - r4[v]);
            ex[v] += dr;
            RM_LDS(Xs, v) = dar;
            fs_rm_st<D, 7 * D>(b_o, v, dar, 4 * D);
        }
        FS_WSYNC();
        {
            const float* const Br[2] = {Ar_h, Ar_h + 4 * D};
            tile_mma<2, 1, 1>(acc_x, Xw, LD, Br, ld2, D, rot);
        }
    }
    {   // da_z has been waiting in Y since the prologue
        const float* const Bz[2] = {Az_h, Az_h + 4 * D};
        tile_mma<2, 1, 1>(acc_x, Yw, LD, Bz, ld2, D, rot);
    }
    FS_WSYNC();
    FS_FOR_ACC { Xl[FS_LOFF(reg)] = acc_x[1][0][reg]; }        // X <- dm
    __syncthreads();                         // whole workgroup: the transposed gather reads dm of every row of the tile

    // ---- message backward: G_e = gather^T_e(dm) ; dh += G_e . W_e^T ----
    f32x16 acc_h[1][1];
    acc_h[0][0] = acc_x[0][0];
    int tmask = 0, types = 0;
    for (int e = 0; e < 4; ++e) {
        const float* const Bp[1] = {a.Wnat + (size_t)(4 * hi) * 4 * D + 4 * (e * D + col)};
        const int ldw[1] = {4 * D};
        BPre<1> pre;
        tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
        (void)FS_GATHER(Xs, Ys, e);
        {   // G_e -> HBM for the weight-gradient GEMM (the lane's own half row, 16-byte stores)
            const float* s = Ys + grow * LD + gq * (D / 2);
            float* o = a.gda + (size_t)(row0 + grow) * 7 * D + e * D + gq * (D / 2);
#pragma unroll
            for (int f = 0; f < D / 8; ++f) *(f32x4*)(o + 4 * f) = *(const f32x4*)(s + 4 * f);
        }
        if (e == 0) types = fs_wave_types(tmask);
        FS_WSYNC();
        if ((types >> e) & 1) tile_mma<1, 1, 1>(acc_h, Yw, LD, Bp, ldw, D, rot, &pre);
        FS_WSYNC();
    }
    // ---- dh = (MFMA part, via Y) + ex ----
    FS_FOR_ACC { Yl[FS_LOFF(reg)] = acc_h[0][0][reg]; }
    FS_WSYNC();
#pragma unroll
    for (int v = 0; v < NV; ++v) fs_rm_st<D, D>(b_dh, v, RM_LDS(Ys, v) + ex[v]);
#undef RM_ROW
#undef RM_C4
#undef RM_LDS
}

// ---------------------------------------------------------------------------------------------
// launchers (called from the C ABI in bmp_fused.hip)
// ---------------------------------------------------------------------------------------------
static size_t fs_lds_bytes(int D) { return ((size_t)2 * FS_R * (D + 4) + FS_R * 4 + 132 + 2 * FZ_ECAP + 4) * sizeof(float); }

template <int D, bool FIRST, bool VAR>
static int fs_launch3(bool bwd, const StepArgs& a, int n_tiles, double rows, hipStream_t st) {
    const bool keep = a.m != nullptr;
    const double gates = FIRST ? 4.0 : 7.0;
    BmpProfScope prof(bwd ? BMP_KCLS_STEP_BWD : BMP_KCLS_STEP_FWD, 2.0 * rows * (4.0 + gates) * D * D, 4.0 * rows * D * (bwd ? 13.0 : 6.0), st,
                      FIRST ? BMP_KID_GGNN_FIRST : BMP_KID_GGNN_LATER);
    const size_t lds = fs_lds_bytes(D);      // 48 KB: under the 64 KB a launch may ask for without an attribute
    if (bwd) hipLaunchKernelGGL((k_ggnn_step_bwd_s<D, FIRST, VAR>), dim3(n_tiles), dim3(FS_NT), lds, st, a);
    else if (keep) hipLaunchKernelGGL((k_ggnn_step_fwd_s<D, FIRST, VAR, true>), dim3(n_tiles), dim3(FS_NT), lds, st, a);
    else hipLaunchKernelGGL((k_ggnn_step_fwd_s<D, FIRST, VAR, false>), dim3(n_tiles), dim3(FS_NT), lds, st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}

int bmp_launch_step_small(bool bwd, const StepArgs& a, int n_tiles, int d, hipStream_t st) {
    BMP_REQUIRE(d == 32);
    const double rows = a.mt_row0 != nullptr ? (double)a.mt_rows : (double)n_tiles * FS_R;
    if (a.mt_row0 != nullptr)
        return a.first ? fs_launch3<32, true, true>(bwd, a, n_tiles, rows, st) : fs_launch3<32, false, true>(bwd, a, n_tiles, rows, st);
    return a.first ? fs_launch3<32, true, false>(bwd, a, n_tiles, rows, st) : fs_launch3<32, false, false>(bwd, a, n_tiles, rows, st);
}

// ---------------------------------------------------------------------------------------------
// Weight gradients of one d = 32 step (bmp_ggnn_step_wgrad):  o1 [D x 7D] = h^T gda, o2 [D x 3D] = m^T gda[:, 4D:],
// dUcT [D x D] = (r*h)^T gda[:, 6D:], cs [7D] = column sums of gda -- reductions over all N rows of outputs that are ONE
// 32-row MFMA block tall.  The general weight-gradient GEMM (128 x 128 output tiles staged through LDS, bmp_gemm.hip) spends a
// d = 32 step on padding: three launches + three slab folds + a column-sum pass, 170 us per step of the reference's published
// model, more than the step's forward and backward kernels together.  Here the operands go straight from global memory into the
// MFMA registers -- a row of X is the A fragment of the transposed product as it lies in memory, and NB consecutive columns
// of gda per lane are NB B fragments whose output columns are interleaved (column NB * lane + t: undone by the store, which
// is NB consecutive floats per lane again) -- so every byte of gda is read once, by one wave, in 16-byte pieces:
//   wave 0: o1[:, 0:4D]   (A = h,   B = the four G blocks, one dwordx4 per lane and row pair) + their column sums
//   wave 1: o1[:, 4D:7D]  (A = h,   B = da_r | da_z | da_c) + their column sums
//   wave 2: o2            (A = m,   B = the same three blocks)
//   wave 3: dUcT          (A = r*h, B = da_c)
// The first call after reset has no r gate: the da_r columns of gda are not written by the backward and are not read here.
// Rows are split over at most 256 workgroups; partial results go to the workspace in the outputs' own layout and are folded
// in workgroup order by k_step_wgrad_fold_s (bitwise reproducible; accumulate adds into the outputs: tied layers).
// ---------------------------------------------------------------------------------------------
template <int NB> struct FsVec;
template <> struct FsVec<1> { typedef float T; };
template <> struct FsVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct FsVec<4> { typedef f32x4 T; };
struct __attribute__((packed, aligned(4))) FsF3 { float v[3]; };
template <int NB> __device__ __forceinline__ void fs_ldv(const float* p, float (&o)[NB]) {
    if constexpr (NB == 3) { const FsF3 x = *(const FsF3*)p; o[0] = x.v[0]; o[1] = x.v[1]; o[2] = x.v[2]; }
    else if constexpr (NB == 1) { o[0] = *p; }
    else { const typename FsVec<NB>::T x = *(const typename FsVec<NB>::T*)p;
#pragma unroll
           for (int t = 0; t < NB; ++t) o[t] = x[t]; }
}
template <int NB> __device__ __forceinline__ void fs_stv(float* p, const float (&o)[NB]) {
    if constexpr (NB == 3) { FsF3 x; x.v[0] = o[0]; x.v[1] = o[1]; x.v[2] = o[2]; *(FsF3*)p = x; }
    else if constexpr (NB == 1) { *p = o[0]; }
    else { typename FsVec<NB>::T x;
#pragma unroll
           for (int t = 0; t < NB; ++t) x[t] = o[t];
           *(typename FsVec<NB>::T*)p = x; }
}

// out[i, NB*j + t] (i, j < 32) = sum over rows r0 <= r < r1 of X[r, i] (* X2[r, i]) * G[r, NB*j + t];  cs[NB*j + t] = sum of G[r, .]
template <int D, int NB, bool X2, bool CS>
__device__ __forceinline__ void fs_wgrad_job(const float* X, int ldx, const float* Xb, int ldxb, const float* G, int ldg, int r0, int r1,
                                             float* out, int ldo, float* cs) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[NB];
    float csum[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { csum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; }
    for (int r = r0; r < r1; r += 8) {               // four MFMA k-steps (two rows each) per trip, all loads first
        float a[4], b[4][NB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t row = (size_t)(r + 2 * u + hi);
            a[u] = X[row * ldx + l31];
            if (X2) a[u] *= Xb[row * ldxb + l31];
            fs_ldv<NB>(G + row * ldg + NB * l31, b[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                acc[t] = bmp_mfma(a[u], b[u][t], acc[t]);
                if (CS) csum[t] += b[u][t];
            }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
        float o[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) o[t] = acc[t][reg];
        fs_stv<NB>(out + (size_t)i * ldo + NB * l31, o);
    }
    if (CS) {
        float o[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) o[t] = csum[t] + __shfl_xor(csum[t], 32);      // rows of even + odd parity, fixed order
        if (hi == 0) fs_stv<NB>(cs + NB * l31, o);
    }
}

#define FSW_PART(D) ((D) * 7 * (D) + (D) * 3 * (D) + (D) * (D) + 7 * (D))      // o1 | o2 | dUcT | cs

template <int D, bool FIRST>
__global__ __launch_bounds__(256) void k_step_wgrad_s(const float* __restrict__ h, const float* __restrict__ m, const float* __restrict__ rz,
                                                      const float* __restrict__ gda, int N, int rps, float* __restrict__ ws) {
    const int w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rps, r1 = r0 + rps < N ? r0 + rps : N;
    float* part = ws + (size_t)blockIdx.x * FSW_PART(D);
    float* p_o1 = part, *p_o2 = part + D * 7 * D, *p_u = p_o2 + D * 3 * D, *p_cs = p_u + D * D;
    constexpr int LG = 7 * D;
    if (w == 0) fs_wgrad_job<D, 4, false, true>(h, D, nullptr, 0, gda, LG, r0, r1, p_o1, LG, p_cs);
    else if (w == 1) {
        if (FIRST) fs_wgrad_job<D, 2, false, true>(h, D, nullptr, 0, gda + 5 * D, LG, r0, r1, p_o1 + 5 * D, LG, p_cs + 5 * D);
        else fs_wgrad_job<D, 3, false, true>(h, D, nullptr, 0, gda + 4 * D, LG, r0, r1, p_o1 + 4 * D, LG, p_cs + 4 * D);
    } else if (w == 2) {
        if (FIRST) fs_wgrad_job<D, 2, false, false>(m, D, nullptr, 0, gda + 5 * D, LG, r0, r1, p_o2 + D, 3 * D, nullptr);
        else fs_wgrad_job<D, 3, false, false>(m, D, nullptr, 0, gda + 4 * D, LG, r0, r1, p_o2, 3 * D, nullptr);
    } else if (!FIRST) {
        fs_wgrad_job<D, 1, true, false>(rz, 2 * D, h, D, gda + 6 * D, LG, r0, r1, p_u, D, nullptr);       // A = r * h
    }
}

// out (=|+=) sum over the G partials, in workgroup order.  first: the da_r parts (o1[:, 4D:5D], o2[:, 0:D], dUcT, cs[4D:5D]) were
// not computed and count as zeros.
template <int D>
__global__ __launch_bounds__(256) void k_step_wgrad_fold_s(const float* __restrict__ ws, int G, int first, int accumulate,
                                                           float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ dUcT,
                                                           float* __restrict__ cs) {
    // 64 output elements per workgroup, four lanes of partial sums each (partials g, g + 4, ...), joined in a fixed order: a
    // quarter of the dependent chain of one thread per element walking all G partials (that form: 60 us per launch, more than
    // the products' kernel)
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + c;
    float* dst = nullptr; bool live = false;
    if (idx < FSW_PART(D)) {
        live = true;
        if (idx < D * 7 * D) { const int col = idx % (7 * D); dst = o1 + idx; live = !(first && col >= 4 * D && col < 5 * D); }
        else if (idx < D * 10 * D) { const int k = idx - D * 7 * D; dst = o2 + k; live = !(first && (k % (3 * D)) < D); }
        else if (idx < D * 11 * D) { dst = dUcT + (idx - D * 10 * D); live = !first; }
        else { const int col = idx - D * 11 * D; dst = cs + col; live = !(first && col >= 4 * D && col < 5 * D); }
    }
    float acc = 0.f;
    if (live) {
#pragma unroll 8
        for (int g = q; g < G; g += 4) acc += ws[(size_t)g * FSW_PART(D) + idx];
    }
    red[q][c] = acc;
    __syncthreads();
    if (q == 0 && dst != nullptr) {
        const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        *dst = accumulate ? *dst + v : v;
    }
}

static void fsw_plan(int N, int& G, int& rps) {
    rps = 8 * ((N + 8 * 256 - 1) / (8 * 256));
    G = (N + rps - 1) / rps;
}
size_t bmp_step_wgrad_small_ws_floats(int N, int d) {
    int G, rps; fsw_plan(N, G, rps);
    return (size_t)G * FSW_PART(32);
}
int bmp_launch_step_wgrad_small(const float* h, const float* m, const float* rz, const float* gda, int N, int d, int first, float* o1,
                                float* o2, float* dUcT, float* cs, int accumulate, float* ws, hipStream_t st) {
    BMP_REQUIRE(d == 32 && N > 0 && (N & 7) == 0);
    int G, rps; fsw_plan(N, G, rps);
    const double cols = first ? 6.0 + 2.0 : 7.0 + 3.0 + 1.0;
    BmpProfScope prof(BMP_KCLS_WGRAD, 2.0 * N * d * cols * d, 4.0 * N * d * (7.0 + 3.0), st, BMP_KID_WGRAD_STEP);
    if (first) hipLaunchKernelGGL((k_step_wgrad_s<32, true>), dim3(G), dim3(256), 0, st, h, m, rz, gda, N, rps, ws);
    else hipLaunchKernelGGL((k_step_wgrad_s<32, false>), dim3(G), dim3(256), 0, st, h, m, rz, gda, N, rps, ws);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_step_wgrad_fold_s<32>), dim3((FSW_PART(32) + 63) / 64), dim3(256), 0, st, ws, G, first, accumulate, o1, o2, dUcT, cs);
    BMP_LAUNCH_CHECK();
    return 0;
}
