// Fused GGNN propagation step for gfx950: message (per-bond-type gather-sum + linear) and GRU node
// update in ONE kernel per 128-row tile, forward and backward  (models/ggnn.py:215-263).
//
// A tile holds whole molecules (bmp/packed.py), so the neighbour gather is tile-local: the tile's
// atom states live in LDS for the whole step and never round-trip through HBM between the message
// linear, the three gate GEMMs and the candidate GEMM.  Per tile and step the forward reads h once
// (64 KB at d=128) and writes m, r|z, c, h' (the backward's inputs); weights (720 KB fp32 at d=128)
// stream from L2 into MFMA B registers, double-buffered in registers one k-step ahead.
//
//   workgroup = 512 threads = 8 waves (2 per SIMD), LDS = two [128 x (D+4)] f32 tiles (135 KB at D=128)
//   wave (wr, wc) owns rows [wr*RB*32, +RB*32) x cols [wc*32, +32) of every [128 x D] result,
//   D = 128: 2 x 4 waves, RB = 2;  D = 64: 4 x 2 waves, RB = 1.
//   dense work: v_mfma_f32_32x32x2_f32 (exact f32); K order per lane: four consecutive k per 16-B read.
#include <stdlib.h>
#include <type_traits>
#include <string.h>
#include "bmp_kernels.h"

#include "bmp_tile.h"

// SAVE == false: forward-only evaluation (m, r|z, c are not kept).  The epilogues' arithmetic is written with explicit fused
// multiply-adds so that both instances round alike: predict's logits are bit for bit the training forward's.
template <int D, bool FIRST, bool VAR, bool SAVE>
__global__ __launch_bounds__(512) void k_ggnn_step_fwd(StepArgs a) {
    constexpr int LD = D + 4;
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Hs = lds;                         // [128 x LD]  h tile (whole step)
    float* As = lds + FZ_R * LD;             // [128 x LD]  AGG_e -> M -> r*h
    float* wds = As + FZ_R * LD;             // [128 x 4]   weighted degree per bond type
    int* rptr = (int*)(wds + FZ_R * 4);      // [132]       tile-relative CSR row pointers
    int* ecol = rptr + 132;                  // [FZ_ECAP]
    float* evalv = (float*)(ecol + FZ_ECAP); // [FZ_ECAP]
    int* sy = (int*)(evalv + FZ_ECAP);       // [FZ_NSYNC]  group counters, bond-type masks

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w >> 2;                  // half of the tile this wave works on in every phase
    GrpSync gs{sy + grp, 0};
    if (tid < FZ_NSYNC) sy[tid] = 0;
    if (grp == 0) __builtin_amdgcn_s_setprio(2);
    const int wc = w % NCB, wr = w / NCB;
    const int l31 = lane & 31, hi = lane >> 5;
    FZ_TILE_SETUP();
    if (VAR && a.tile_stride > nrows) {      // a tile of a fixed-stride table: its blocks without atoms, in every array written
        const int e = row0 + a.tile_stride;
        fz_clear_rows<512>(a.hout, D, row0 + nrows, e, tid);
        fz_clear_rows<512>(a.m, D, row0 + nrows, e, tid);
        fz_clear_rows<512>(a.rz, 2 * D, row0 + nrows, e, tid);
        fz_clear_rows<512>(a.c, D, row0 + nrows, e, tid);
    }
    const int col = wc * 32 + l31;
    const int wrow0 = wr * RB * 32;
    const int lrow = wrow0 + 4 * hi;         // this lane's row for reg 0 of row block 0
    const int rot = (tile * 8) % D;
    const float* Hw = Hs + (wrow0 + l31) * LD + 4 * hi;
    const float* Aw = As + (wrow0 + l31) * LD + 4 * hi;
    float* Hl = Hs + lrow * LD + col;        // accumulator-layout views of the two LDS tiles
    float* Al = As + lrow * LD + col;
#define LOFF(rb, reg) (((rb) * 32 + ((reg) & 3) + 8 * ((reg) >> 2)) * LD)

    // forward-only evaluation (predict under no-backprop, train_binary.py:120-127): m, r|z and c are the backward's inputs
    // and are not written when the caller passes no arrays for them
    constexpr bool save = SAVE;              // (the launcher picks the instance: a.m != nullptr)
    // ---- h tile -> LDS ----
    for (int idx = tid; idx < nrows * (D / 4); idx += 512) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        *(f32x4*)(Hs + r * LD + 4 * c4) = *(const f32x4*)(a.h + (size_t)(row0 + r) * D + 4 * c4);
    }
    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows);
    __syncthreads();
    if (!grp_live) return;                   // a short tile: this half has no rows
    const RowOrder ro = FZ_ROW_ORDER();      // the half's rows by (rare bond type, row): the order of the message phase
    const int bi = wr & 1;                   // (RB == 1: this wave's block inside its half)

    // one propagation step on the resident tile (Hs): message, gates, h'.  (A lambda over the GRU form: the first call after
    // reset has no r gate and no U term.  A variant running all T steps of a tile in one launch was built in round 3 on this
    // text, measured slower on every leg -- 256 VGPRs and scratch -- and taken out again in round 4: DESIGN.md 3a''.)
    auto step = [&](auto first_c, const float* WTs, const float* bEs, const float* ATs, const float* bvs, float* om, float* orz,
                    float* oc, float* oh) {
        constexpr bool FST = decltype(first_c)::value;
        int tmask = 0;
        // ---- message: m = sum_e AGG_e . W_e + wdeg_e * b_e   (models/ggnn.py:223-242) ----
        f32x16 acc_m[1][RB];
        zero_acc(acc_m[0]);
        for (int e = 0; e < 4; ++e) {
            const float* const Bp[1] = {WTs + (size_t)(e * D + 4 * hi) * D + 4 * col};
            const int ldw[1] = {D};
            BPre<1> pre;
            tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
            float wd;
            FZ_GATHER(Hs, As, e, &wd);
            if ((tid & 3) == 0) wds[(tid >> 2) * 4 + e] = wd;
            // bond types present in this half of the tile: known after the first pass (it walks every entry)
            if (e == 0 && (tid & 3) == 0 && tmask) __hip_atomic_fetch_or(sy + 2 + grp, tmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            grp_sync(gs);
            const int any = (__hip_atomic_load(sy + 2 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> e) & 1;
            if (any) tile_mma_msg<VAR, 1, RB>(nrb, e, ro.nbr, bi, acc_m, Aw, LD, Bp, ldw, D, rot, &pre);
            FZ_GSYNC();
        }
        // first B fragments of the h-part of the gates: requested now, used after the m epilogue and its barrier
        constexpr int NG = FST ? 2 : 3;                     // gates computed: first call after reset z and c only
        int ldwg[NG];
    #pragma unroll
        for (int g = 0; g < NG; ++g) ldwg[g] = 3 * D;
        const float* const base_h = ATs + (size_t)(4 * hi) * 3 * D + 4 * col + (FST ? 4 * D : 0);
        const float* const base_m = ATs + (size_t)(D + 4 * hi) * 3 * D + 4 * col + (FST ? 4 * D : 0);
        const float* Bh[NG]; const float* Bm[NG];
    #pragma unroll
        for (int g = 0; g < NG; ++g) { Bh[g] = base_h + 4 * D * g; Bm[g] = base_m + 4 * D * g; }
        BPre<NG> pre_h;
        tile_b_prefetch<NG>(pre_h, (const float* const (&)[NG])Bh, (const int (&)[NG])ldwg, D, rot);
        // m -> LDS (A operand of the gates) and HBM (saved for the backward).  The accumulators hold the rows in the message
        // phase's order: accumulator position -> row through ro.perm; from here on everything is in packed row order again
        {
            const AccBuf mo = acc_buf<D>(om, row0, 0, col);
            float be[4];
    #pragma unroll
            for (int e = 0; e < 4; ++e) be[e] = bEs[e * D + col];
            FZ_FOR_ACC {
                const int tp = lrow + rb * 32 + (reg & 3) + 8 * (reg >> 2);          // position in the tile
                const int r = (tp & 64) + ro.perm[tp];                                // its row
                const f32x4 wd4 = *(const f32x4*)(wds + r * 4);
                const float v = __builtin_fmaf(wd4[3], be[3], __builtin_fmaf(wd4[2], be[2], __builtin_fmaf(wd4[1], be[1],
                                               __builtin_fmaf(wd4[0], be[0], acc_m[0][rb][reg]))));
                As[r * LD + col] = v;
                if (save) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), mo.rs, mo.vo + r * D * 4, 0, 0);
            }
        }
        FZ_GSYNC();

        // ---- gates: [r | z | c~] = [h, m] . AT   (chainer StatefulGRU) ----
        f32x16 acc_g[3][RB];
        zero_acc(acc_g[0]); zero_acc(acc_g[1]); zero_acc(acc_g[2]);
        {
            f32x16 gg[NG][RB];
    #pragma unroll
            for (int g = 0; g < NG; ++g) zero_acc(gg[g]);
            BPre<NG> pre_m;                                   // in flight under the h-part
            tile_b_prefetch<NG>(pre_m, (const float* const (&)[NG])Bm, (const int (&)[NG])ldwg, D, rot);
            tile_mma_n<VAR, NG, RB>(nrb, gg, Hw, LD, (const float* const (&)[NG])Bh, (const int (&)[NG])ldwg, D, rot, &pre_h);
            tile_mma_n<VAR, NG, RB>(nrb, gg, Aw, LD, (const float* const (&)[NG])Bm, (const int (&)[NG])ldwg, D, rot, &pre_m);
    #pragma unroll
            for (int g = 0; g < NG; ++g)
    #pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc_g[g + (FST ? 1 : 0)][rb] = gg[g][rb];
        }
        const float br = bvs[col], bz = bvs[D + col], bcn = bvs[2 * D + col];
        {   // r, z in place; save them
            const AccBuf rzo = acc_buf<2 * D>(orz, row0, lrow, col);
            FZ_FOR_ACC {
                const float zv = bmp_sigmoid(acc_g[1][rb][reg] + bz);
                acc_g[1][rb][reg] = zv;
                if (save) acc_st<2 * D>(rzo, rb, reg, zv, D);
                if (!FST) {
                    const float rv = bmp_sigmoid(acc_g[0][rb][reg] + br);
                    acc_g[0][rb][reg] = rv;
                    if (save) acc_st<2 * D>(rzo, rb, reg, rv, 0);
                }
            }
        }
        if (!FST) {
            const float* const Bu[1] = {a.UcT + (size_t)(4 * hi) * D + 4 * col};
            const int ldu[1] = {D};
            BPre<1> pre_u;
            tile_b_prefetch<1>(pre_u, Bu, ldu, D, rot);
            FZ_GSYNC();                          // every wave of this half is done reading M
            FZ_FOR_ACC { Al[LOFF(rb, reg)] = acc_g[0][rb][reg] * Hl[LOFF(rb, reg)]; }      // r * h
            FZ_GSYNC();
            f32x16 gc[1][RB];
    #pragma unroll
            for (int rb = 0; rb < RB; ++rb) gc[0][rb] = acc_g[2][rb];
            tile_mma_n<VAR, 1, RB>(nrb, gc, Aw, LD, Bu, ldu, D, rot, &pre_u);
    #pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc_g[2][rb] = gc[0][rb];
        }
        // ---- h' = z*c + (1-z)*h  (first call: z*c) ----
        {
            const AccBuf co = acc_buf<D>(oc, row0, lrow, col);
            const AccBuf ho = acc_buf<D>(oh, row0, lrow, col);
            FZ_FOR_ACC {
                const float cv = bmp_tanh(acc_g[2][rb][reg] + bcn);
                const float zv = acc_g[1][rb][reg];
                float hn = zv * cv;
                if (!FST) hn = __builtin_fmaf(1.f - zv, Hl[LOFF(rb, reg)], hn);
                if (save) acc_st<D>(co, rb, reg, cv);
                acc_st<D>(ho, rb, reg, hn);
            }
        }
    };
    step(std::integral_constant<bool, FIRST>{}, a.WT, a.bE, a.AT, a.b, a.m, a.rz, a.c, a.hout);
}

// Backward of one step for one tile: all of the backward-data path (gate derivatives, the three
// transposed gate GEMMs, the message-linear transpose and the transposed neighbour gather), and the
// per-row pre-activation gradients [G | da] the weight-gradient GEMMs consume.
//
// All HBM traffic is row-major 16-byte accesses (a wave instruction moves 1 KiB of one or two rows): the
// elementwise gate derivatives are computed in that layout straight from the loads and land in LDS as the
// MFMA A operands; what the MFMAs produce in accumulator layout (d(r*h), dm, dh) crosses to row-major
// through the LDS tile it has to visit anyway.  (Accumulator-layout dword loads/stores of the same arrays
// cost 64 + 30 us of a 287 us launch.)
template <int D, bool FIRST, bool VAR>
__global__ __launch_bounds__(512) void k_ggnn_step_bwd(StepArgs a) {
    constexpr int LD = D + 4;
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    constexpr int F4 = D / 4;                // float4 per row
    constexpr int NV = D / 16;               // float4 per thread of a 64-row half (256 threads)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                         // [128 x LD]  da_c -> d(r*h) -> da_r -> dm
    float* Ys = lds + FZ_R * LD;             // [128 x LD]  da_z -> G_e (transposed gather of dm) -> dh
    int* rptr = (int*)(Ys + FZ_R * LD + FZ_R * 4);
    int* ecol = rptr + 132;
    float* evalv = (float*)(ecol + FZ_ECAP);
    int* sy = (int*)(evalv + FZ_ECAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w >> 2;
    GrpSync gs{sy + grp, 0};
    int tmask = 0;
    if (tid < FZ_NSYNC) sy[tid] = 0;
    if (grp == 0) __builtin_amdgcn_s_setprio(2);
    const int wc = w % NCB, wr = w / NCB;
    const int l31 = lane & 31, hi = lane >> 5;
    FZ_TILE_SETUP();
    if (VAR && a.tile_stride > nrows) {      // (fixed-stride tile table: the tile's blocks without atoms read as zero to the
        const int e = row0 + a.tile_stride;  //  weight-gradient GEMMs and to the step below)
        fz_clear_rows<512>(a.dh, D, row0 + nrows, e, tid);
        fz_clear_rows<512>(a.gda, 7 * D, row0 + nrows, e, tid);
    }
    const int col = wc * 32 + l31;
    const int wrow0 = wr * RB * 32;
    const int lrow = wrow0 + 4 * hi;
    const int rot = (tile * 8) % D;
    const float* Xw = Xs + (wrow0 + l31) * LD + 4 * hi;
    const float* Yw = Ys + (wrow0 + l31) * LD + 4 * hi;
    float* Xl = Xs + lrow * LD + col;
    float* Yl = Ys + lrow * LD + col;
    constexpr bool first = FIRST;
    // row-major view of this half: float4 slot v of this thread = (tile row rm_r(v), float4 column rm_c(v))
    const int tg = tid & 255;
#define RM_ROW(v) (grp * 64 + ((v) * 256 + tg) / F4)
#define RM_C4(v) (((v) * 256 + tg) % F4)
#define RM_LDS(T, v) (*(f32x4*)((T) + RM_ROW(v) * LD + 4 * RM_C4(v)))
#define RM_LIVE(v) (!VAR || grp * 64 + (v) * (256 / F4) < nrows)  /* slot v's rows lie in a live 32-row block (uniform per group) */
    const AccBuf b_g = rm_buf<D, D>(a.dhout, row0, grp, tg), b_c = rm_buf<D, D>(a.c, row0, grp, tg);
    const AccBuf b_h = rm_buf<D, D>(a.h, row0, grp, tg), b_rz = rm_buf<D, 2 * D>(a.rz, row0, grp, tg);
    const AccBuf b_o = rm_buf<D, 7 * D>(a.gda, row0, grp, tg), b_dh = rm_buf<D, D>(a.dh, row0, grp, tg);

    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows);     // visible after the first barrier
    if (!grp_live) {                         // a short tile: this half has no rows.  It has staged its share of the CSR; it
        __syncthreads();                     // meets the first workgroup barrier and leaves (the hardware barrier stops
        return;                              // counting waves that have ended)
    }

    // ---- da_c = dh' z (1 - c^2) -> X ; da_z = dh' (c - h) z (1 - z) -> Y ; ex = dh' (1 - z): the direct part of dh ----
    f32x4 ex[NV];
    {
        const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
        f32x4 g4[NV], z4[NV], c4[NV], h4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) {
            g4[v] = rm_ld<D, D>(b_g, v);
            z4[v] = rm_ld<D, 2 * D>(b_rz, v, D);
            c4[v] = rm_ld<D, D>(b_c, v);
            if (!first) h4[v] = rm_ld<D, D>(b_h, v);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) {
            const f32x4 gz = g4[v] * z4[v];
            const f32x4 dac = gz * (one - c4[v] * c4[v]);
            f32x4 dz = gz * (one - z4[v]);
            if (first) { dz *= c4[v]; ex[v] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            else { dz *= (c4[v] - h4[v]); ex[v] = g4[v] - gz; }
            RM_LDS(Xs, v) = dac;
            RM_LDS(Ys, v) = dz;
            rm_st<D, 7 * D>(b_o, v, dac, 6 * D);
            rm_st<D, 7 * D>(b_o, v, dz, 5 * D);
            // (first call after reset: no r gate, da_r = 0 -- its gda columns are neither written here nor read by
            //  bmp_ggnn_step_wgrad)
        }
    }
    const float* const Ac_h = a.A + (size_t)(2 * D + 4 * hi) * 2 * D + 4 * col;
    const float* const Az_h = a.A + (size_t)(D + 4 * hi) * 2 * D + 4 * col;
    const float* const Ar_h = a.A + (size_t)(4 * hi) * 2 * D + 4 * col;
    const int ld2[2] = {2 * D, 2 * D};
    const float* const Bc[2] = {Ac_h, Ac_h + 4 * D};
    BPre<2> pre_c;
    tile_b_prefetch<2>(pre_c, Bc, ld2, D, rot);
    __syncthreads();                         // whole workgroup: the staged CSR and the counters are visible
    // The half's rows by (rare bond type, row): every MFMA of this kernel takes its A rows in that order (aoff), so all the
    // accumulators line up with the message backward's, where the order pays; what leaves an accumulator for a row-indexed
    // LDS tile (d(r*h), dm, dh) looks its row up (FZ_ROWOF).
    const RowOrder ro = FZ_ROW_ORDER();
    const int bi = wr & 1;
    int aoff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { const int tp = wrow0 + rb * 32 + l31; aoff[rb] = (((tp & 64) + ro.perm[tp]) - (wrow0 + l31)) * LD; }
#define FZ_ROWOF(rb, reg) ((lrow & 64) + ro.perm[lrow + (rb) * 32 + ((reg) & 3) + 8 * ((reg) >> 2)])

    f32x16 acc_x[2][RB];                     // [0] = dh, [1] = dm
    zero_acc(acc_x[0]); zero_acc(acc_x[1]);
    tile_mma_n<VAR, 2, RB>(nrb, acc_x, Xw, LD, Bc, ld2, D, rot, &pre_c, aoff); // [dh | dm] += da_c . A_c
    if (!first) {
        f32x16 acc_d[1][RB];                 // d(r*h) = da_c . U
        zero_acc(acc_d[0]);
        {
            const float* const Bu[1] = {a.Uc + (size_t)(4 * hi) * D + 4 * col};
            const int ldu[1] = {D};
            tile_mma_n<VAR, 1, RB>(nrb, acc_d, Xw, LD, Bu, ldu, D, rot, nullptr, aoff);
        }
        FZ_GSYNC();                          // all waves of this half done with da_c in X
        FZ_FOR_ACC { Xs[FZ_ROWOF(rb, reg) * LD + col] = acc_d[0][rb][reg]; }
        f32x4 r4[NV], h4[NV];                // in flight across the group barrier
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) { r4[v] = rm_ld<D, 2 * D>(b_rz, v); h4[v] = rm_ld<D, D>(b_h, v); }
        FZ_GSYNC();
        // da_r = d(r*h) h r (1-r) -> X (in place) ; ex += d(r*h) r
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) {
            const f32x4 drh = RM_LDS(Xs, v);
            const f32x4 dr = drh * r4[v];
            const f32x4 dar = dr * h4[v] * ((f32x4){1.f, 1.f, 1.f, 1.f} - r4[v]);
            ex[v] += dr;
            RM_LDS(Xs, v) = dar;
            rm_st<D, 7 * D>(b_o, v, dar, 4 * D);
        }
        FZ_GSYNC();
        {
            const float* const Br[2] = {Ar_h, Ar_h + 4 * D};
            tile_mma_n<VAR, 2, RB>(nrb, acc_x, Xw, LD, Br, ld2, D, rot, nullptr, aoff);
        }
    }
    {   // da_z has been waiting in Y since the prologue
        const float* const Bz[2] = {Az_h, Az_h + 4 * D};
        tile_mma_n<VAR, 2, RB>(nrb, acc_x, Yw, LD, Bz, ld2, D, rot, nullptr, aoff);
    }
    FZ_GSYNC();                              // all waves of this half done with X and Y
    // ---- X <- dm ----
    FZ_FOR_ACC { Xs[FZ_ROWOF(rb, reg) * LD + col] = acc_x[1][rb][reg]; }
    __syncthreads();                         // whole workgroup: the transposed gather reads dm of every row of the tile

    // ---- message backward: G_e = gather^T_e(dm) ; dh += G_e . W_e^T ----
    f32x16 acc_h[1][RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc_h[0][rb] = acc_x[0][rb];
    for (int e = 0; e < 4; ++e) {
        const float* const Bp[1] = {a.Wnat + (size_t)(4 * hi) * 4 * D + 4 * (e * D + col)};
        const int ldw[1] = {4 * D};
        BPre<1> pre;
        tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
        float wd;
        const bool has_e = FZ_GATHER(Xs, Ys, e, &wd);
        if ((tid >> 2) < nrows && (has_e || !a.skip_zero_g)) {   // G_e -> HBM for the weight-gradient GEMM (row-wise, 16-byte stores)
            const int row = tid >> 2, q = tid & 3;
            const float* s = Ys + ((row & 64) + ro.inv[row]) * LD + q * (D / 4);       // (the gather put the row at its position)
            float* o = a.gda + (size_t)(row0 + row) * 7 * D + e * D + q * (D / 4);
#pragma unroll
            for (int f = 0; f < D / 16; ++f) *(f32x4*)(o + 4 * f) = *(const f32x4*)(s + 4 * f);
        }
        if (e == 0 && (tid & 3) == 0 && tmask) __hip_atomic_fetch_or(sy + 2 + grp, tmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        grp_sync(gs);
        const int any = (__hip_atomic_load(sy + 2 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> e) & 1;
        if (any) tile_mma_msg<VAR, 1, RB>(nrb, e, ro.nbr, bi, acc_h, Yw, LD, Bp, ldw, D, rot, &pre);
        FZ_GSYNC();
    }
    // ---- dh = (MFMA part, via Y: accumulator position -> row) + ex ----
    FZ_FOR_ACC { Ys[FZ_ROWOF(rb, reg) * LD + col] = acc_h[0][rb][reg]; }
    FZ_GSYNC();
#pragma unroll
    for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) rm_st<D, D>(b_dh, v, RM_LDS(Ys, v) + ex[v]);
#undef RM_LIVE
#undef RM_ROW
#undef RM_C4
#undef RM_LDS
#undef FZ_ROWOF
}

// ---------------------------------------------------------------------------------------------
// Fused RelGCN layer (models/update/relgcn_update.py:24-44 + the tanh of models/relgcn.py:71):
//   out = act(h W_s^T + b_s + sum_e adj'_e (W_e h + b_e))
// = the message half of the GGNN step plus one more K = D pass over the resident h tile (the self connection) and the
// activation in the epilogue.  d_in == d_out == D.  The 1/degree of rescale_adj is in the CSR values.
// ---------------------------------------------------------------------------------------------
struct RelArgs {
    const int* ptr; const int* col; const float* val;      // CSR (fwd) or transposed CSR (bwd)
    int tile0;                      // the launch covers tiles tile0 .. tile0 + gridDim.x - 1 (all arrays whole)
    int act;
    const int* mt_row0; const int* mt_nblk;     // optional tile table, as StepArgs
    int mt_rows;
    // forward
    const float* h;                 // [N x D]
    const float* WT;                // [4D x D]  K4-packed, rows e*D + k
    const float* bE;                // [4 x D]
    const float* WsT;               // [D x D]   K4-packed
    const float* bs;                // [D]
    float* out;                     // [N x D]
    float* wdeg;                    // [N x 4]   weighted degree per bond type (the backward's db_e operand)
    // backward
    const float* dout; const float* y;
    const float* Wnat;              // [D x 4D]  (= WT^T) K4-packed
    const float* Ws;                // [D x D]   (= WsT^T) K4-packed
    float* dh;
    float* gda;                     // [N x 5D]: G_0..G_3 (transposed gather of dpre per bond type) | dpre
    int skip_zero_g;                // the G_e block of a row without a bond of type e (an exact zero) is not written
};

template <int D, bool VAR>
__global__ __launch_bounds__(512) void k_relgcn_layer_fwd(RelArgs a) {
    constexpr int LD = D + 4;
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Hs = lds;
    float* As = lds + FZ_R * LD;
    float* wds = As + FZ_R * LD;
    int* rptr = (int*)(wds + FZ_R * 4);
    int* ecol = rptr + 132;
    float* evalv = (float*)(ecol + FZ_ECAP);
    int* sy = (int*)(evalv + FZ_ECAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w >> 2;
    GrpSync gs{sy + grp, 0};
    int tmask = 0;
    if (tid < FZ_NSYNC) sy[tid] = 0;
    if (grp == 0) __builtin_amdgcn_s_setprio(2);
    const int wc = w % NCB, wr = w / NCB;
    const int l31 = lane & 31, hi = lane >> 5;
    FZ_TILE_SETUP();
    const int col = wc * 32 + l31;
    const int wrow0 = wr * RB * 32;
    const int lrow = wrow0 + 4 * hi;
    const int rot = (tile * 8) % D;
    const float* Hw = Hs + (wrow0 + l31) * LD + 4 * hi;
    const float* Aw = As + (wrow0 + l31) * LD + 4 * hi;

    for (int idx = tid; idx < nrows * (D / 4); idx += 512) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        *(f32x4*)(Hs + r * LD + 4 * c4) = *(const f32x4*)(a.h + (size_t)(row0 + r) * D + 4 * c4);
    }
    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows);
    __syncthreads();
    if (!grp_live) return;                   // a short tile: this half has no rows
    const RowOrder ro = FZ_ROW_ORDER();      // the half's rows by (rare bond type, row): the order of this kernel's accumulators
    const int bi = wr & 1;
    int aoff[RB];                            // the self connection reads its h rows in that order too
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { const int tp = wrow0 + rb * 32 + l31; aoff[rb] = (((tp & 64) + ro.perm[tp]) - (wrow0 + l31)) * LD; }

    f32x16 acc[1][RB];
    zero_acc(acc[0]);
    const float* const Bs[1] = {a.WsT + (size_t)(4 * hi) * D + 4 * col};
    const int lds_[1] = {D};
    BPre<1> pre_s;
    for (int e = 0; e < 4; ++e) {
        const float* const Bp[1] = {a.WT + (size_t)(e * D + 4 * hi) * D + 4 * col};
        const int ldw[1] = {D};
        BPre<1> pre;
        tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
        if (e == 3) tile_b_prefetch<1>(pre_s, Bs, lds_, D, rot);
        float wd;
        FZ_GATHER(Hs, As, e, &wd);
        if ((tid & 3) == 0) wds[(tid >> 2) * 4 + e] = wd;
        if (e == 0 && (tid & 3) == 0 && tmask) __hip_atomic_fetch_or(sy + 2 + grp, tmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        grp_sync(gs);
        const int any = (__hip_atomic_load(sy + 2 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> e) & 1;
        if (any) tile_mma_msg<VAR, 1, RB>(nrb, e, ro.nbr, bi, acc, Aw, LD, Bp, ldw, D, rot, &pre);
        FZ_GSYNC();
    }
    tile_mma_n<VAR, 1, RB>(nrb, acc, Hw, LD, Bs, lds_, D, rot, &pre_s, aoff);    // self connection: h . W_s^T
    {
        const AccBuf oo = acc_buf<D>(a.out, row0, 0, col);
        float be[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) be[e] = a.bE[e * D + col];
        const float bsv = a.bs ? a.bs[col] : 0.f;
        const int act = a.act;
        FZ_FOR_ACC {
            const int tp = lrow + rb * 32 + (reg & 3) + 8 * (reg >> 2);              // accumulator position ...
            const int r = (tp & 64) + ro.perm[tp];                                    // ... -> row
            const f32x4 wd4 = *(const f32x4*)(wds + r * 4);
            const float v = acc[0][rb][reg] + bsv + wd4[0] * be[0] + wd4[1] * be[1] + wd4[2] * be[2] + wd4[3] * be[3];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bmp_act(act, v)), oo.rs, oo.vo + r * D * 4, 0, 0);
        }
    }
    if (a.wdeg != nullptr && (tid & 255) < 64 && grp * 64 + (tid & 255) < nrows) {   // this half's weighted degrees (written by this half's threads, group-synced
                                                   // above); the backward's operand: not written in forward-only evaluation
        const int r = grp * 64 + (tid & 255);
        *(f32x4*)(a.wdeg + (size_t)(row0 + r) * 4) = *(const f32x4*)(wds + r * 4);
    }
}

template <int D, bool VAR>
__global__ __launch_bounds__(512) void k_relgcn_layer_bwd(RelArgs a) {
    constexpr int LD = D + 4;
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    constexpr int F4 = D / 4;
    constexpr int NV = D / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                         // [128 x LD]  dpre (read-only after the prologue)
    float* Ys = lds + FZ_R * LD;             // [128 x LD]  G_e -> dh
    int* rptr = (int*)(Ys + FZ_R * LD + FZ_R * 4);
    int* ecol = rptr + 132;
    float* evalv = (float*)(ecol + FZ_ECAP);
    int* sy = (int*)(evalv + FZ_ECAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w >> 2;
    GrpSync gs{sy + grp, 0};
    int tmask = 0;
    if (tid < FZ_NSYNC) sy[tid] = 0;
    if (grp == 0) __builtin_amdgcn_s_setprio(2);
    const int wc = w % NCB, wr = w / NCB;
    const int l31 = lane & 31, hi = lane >> 5;
    FZ_TILE_SETUP();
    const int col = wc * 32 + l31;
    const int wrow0 = wr * RB * 32;
    const int lrow = wrow0 + 4 * hi;
    const int rot = (tile * 8) % D;
    const float* Xw = Xs + (wrow0 + l31) * LD + 4 * hi;
    const float* Yw = Ys + (wrow0 + l31) * LD + 4 * hi;
    float* Yl = Ys + lrow * LD + col;
    const int tg = tid & 255;
#define RM_ROW(v) (grp * 64 + ((v) * 256 + tg) / F4)
#define RM_C4(v) (((v) * 256 + tg) % F4)
#define RM_LDS(T, v) (*(f32x4*)((T) + RM_ROW(v) * LD + 4 * RM_C4(v)))
#define RM_LIVE(v) (!VAR || grp * 64 + (v) * (256 / F4) < nrows)
    const AccBuf b_g = rm_buf<D, D>(a.dout, row0, grp, tg), b_y = rm_buf<D, D>(a.y, row0, grp, tg);
    const AccBuf b_o = rm_buf<D, 5 * D>(a.gda, row0, grp, tg), b_dh = rm_buf<D, D>(a.dh, row0, grp, tg);

    const bool csr_lds = stage_csr(a.ptr, a.col, a.val, row0, rptr, ecol, evalv, nrows);
    if (!grp_live) {                         // a short tile: this half has no rows (see k_ggnn_step_bwd)
        __syncthreads();
        return;
    }

    {   // dpre = dout * act'(out) -> X and gda[:, 4D:]
        f32x4 g4[NV], y4[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) { g4[v] = rm_ld<D, D>(b_g, v); y4[v] = rm_ld<D, D>(b_y, v); }
        const int act = a.act;
#pragma unroll
        for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) {
            f32x4 dp;
#pragma unroll
            for (int t = 0; t < 4; ++t) dp[t] = g4[v][t] * bmp_dact(act, y4[v][t]);
            RM_LDS(Xs, v) = dp;
            rm_st<D, 5 * D>(b_o, v, dp, 4 * D);
        }
    }
    const float* const Bs[1] = {a.Ws + (size_t)(4 * hi) * D + 4 * col};
    const int lds_[1] = {D};
    BPre<1> pre_s;
    tile_b_prefetch<1>(pre_s, Bs, lds_, D, rot);
    __syncthreads();
    const RowOrder ro = FZ_ROW_ORDER();      // the half's rows by (rare bond type, row): the order of this kernel's accumulators
    const int bi = wr & 1;
    int aoff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { const int tp = wrow0 + rb * 32 + l31; aoff[rb] = (((tp & 64) + ro.perm[tp]) - (wrow0 + l31)) * LD; }

    f32x16 acc_h[1][RB];
    zero_acc(acc_h[0]);
    tile_mma_n<VAR, 1, RB>(nrb, acc_h, Xw, LD, Bs, lds_, D, rot, &pre_s, aoff);  // dh = dpre . W_s
    for (int e = 0; e < 4; ++e) {
        const float* const Bp[1] = {a.Wnat + (size_t)(4 * hi) * 4 * D + 4 * (e * D + col)};
        const int ldw[1] = {4 * D};
        BPre<1> pre;
        tile_b_prefetch<1>(pre, Bp, ldw, D, rot);
        float wd;
        const bool has_e = FZ_GATHER(Xs, Ys, e, &wd);
        if ((tid >> 2) < nrows && (has_e || !a.skip_zero_g)) {
            const int row = tid >> 2, q = tid & 3;
            const float* s = Ys + ((row & 64) + ro.inv[row]) * LD + q * (D / 4);       // (the gather put the row at its position)
            float* o = a.gda + (size_t)(row0 + row) * 5 * D + e * D + q * (D / 4);
#pragma unroll
            for (int f = 0; f < D / 16; ++f) *(f32x4*)(o + 4 * f) = *(const f32x4*)(s + 4 * f);
        }
        if (e == 0 && (tid & 3) == 0 && tmask) __hip_atomic_fetch_or(sy + 2 + grp, tmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        grp_sync(gs);
        const int any = (__hip_atomic_load(sy + 2 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> e) & 1;
        if (any) tile_mma_msg<VAR, 1, RB>(nrb, e, ro.nbr, bi, acc_h, Yw, LD, Bp, ldw, D, rot, &pre);
        FZ_GSYNC();
    }
    FZ_FOR_ACC {
        const int tp = lrow + rb * 32 + (reg & 3) + 8 * (reg >> 2);
        Ys[((tp & 64) + ro.perm[tp]) * LD + col] = acc_h[0][rb][reg];
    }
    FZ_GSYNC();
#pragma unroll
    for (int v = 0; v < NV; ++v) if (RM_LIVE(v)) rm_st<D, D>(b_dh, v, RM_LDS(Ys, v));
#undef RM_LIVE
#undef RM_ROW
#undef RM_C4
#undef RM_LDS
}

// ---------------------------------------------------------------------------------------------
// Gated-sum readout forward on the tile machinery (models/ggnn.py:333-341, models/readout/ggnn_readout.py:42-57) for
// d == o in {64, 128}, h0 absent or d wide:  [i | j] = [h, h0] . WT + b ; g[mol] = sum_rows w * sigmoid(i) * act_j(j).
// The tile's rows are resident in LDS, the 2o output columns are two accumulators per wave (same column of i and j, so
// the gate is formed in registers), and because a molecule never leaves its tile the per-molecule sums are taken
// from LDS in a fixed order by the tile's own workgroup: no second kernel, no re-read of ij.
// ---------------------------------------------------------------------------------------------
struct ROArgs {
    const float* h; const float* h0;
    const float* WT;                // [(D + D0) x 2D] K4-packed, columns [i | j]
    const float* b;                 // [2D] or nullptr
    int act_j;
    const float* row_w; const int* row_mol; const int* mol_nrows;
    float* ij;                      // [N x 2D]  sigmoid(i) | act_j(j)   (the backward's input)
    float* g;                       // [n_mols x D]
};

template <int D, bool HAS0>
__global__ __launch_bounds__(512) void k_readout_tile_fwd(ROArgs a) {
    constexpr int LD = D + 4;
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Hs = lds;                         // [128 x LD]  h tile
    float* As = lds + FZ_R * LD;             // [128 x LD]  h0 tile -> w * sigmoid(i) * act_j(j)
    int* mlist = (int*)(As + FZ_R * LD);     // [64 x 2] (first tile row, molecule) + [2] counts + scratch (see below)
    float* rws = (float*)(mlist + 400);      // [128] row multiplicities of the tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wc = w % NCB, wr = w / NCB;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * FZ_R;
    const int col = wc * 32 + l31;
    const int wrow0 = wr * RB * 32;
    const int lrow = wrow0 + 4 * hi;
    const int rot = (blockIdx.x * 8) % D;
    constexpr int nrb = RB;                  // (this kernel works on the per-instance layout: whole 128-row tiles)
    const float* Hw = Hs + (wrow0 + l31) * LD + 4 * hi;
    const float* Aw = As + (wrow0 + l31) * LD + 4 * hi;
    float* Al = As + lrow * LD + col;

    for (int idx = tid; idx < FZ_R * (D / 4); idx += 512) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        *(f32x4*)(Hs + r * LD + 4 * c4) = *(const f32x4*)(a.h + (size_t)(row0 + r) * D + 4 * c4);
        if (HAS0) *(f32x4*)(As + r * LD + 4 * c4) = *(const f32x4*)(a.h0 + (size_t)(row0 + r) * D + 4 * c4);
    }
    if (tid < FZ_R) rws[tid] = a.row_w[row0 + tid];
    const float* const Bh[2] = {a.WT + (size_t)(4 * hi) * 2 * D + 4 * col, a.WT + (size_t)(4 * hi) * 2 * D + 4 * (D + col)};
    const int ldw[2] = {2 * D, 2 * D};
    BPre<2> pre_h;
    tile_b_prefetch<2>(pre_h, Bh, ldw, D, rot);
    __syncthreads();

    f32x16 acc[2][RB];                       // [0] = i, [1] = j
    zero_acc(acc[0]); zero_acc(acc[1]);
    if (HAS0) {
        const float* const B0[2] = {Bh[0] + (size_t)D * 2 * D, Bh[1] + (size_t)D * 2 * D};
        BPre<2> pre_0;
        tile_b_prefetch<2>(pre_0, B0, ldw, D, rot);
        tile_mma<2, RB>(acc, Hw, LD, Bh, ldw, D, rot, &pre_h);
        tile_mma<2, RB>(acc, Aw, LD, B0, ldw, D, rot, &pre_0);
    } else {
        tile_mma<2, RB>(acc, Hw, LD, Bh, ldw, D, rot, &pre_h);
    }
    // molecules of this tile: rows whose molecule differs from the previous row's
    if (tid < FZ_R) {
        const int mol = a.row_mol[row0 + tid];
        const int prev = tid > 0 ? a.row_mol[row0 + tid - 1] : -2;
        const bool head = mol >= 0 && mol != prev;
        const unsigned long long bal = __ballot(head);
        if (lane == 0) mlist[128 + w] = __popcll(bal);         // wave 1 adds wave 0's count after the barrier
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (head) { mlist[130 + tid] = before; }          // provisional index within the wave
        mlist[260 + tid] = head ? mol : -1;
    }
    __syncthreads();                         // also: every wave is done reading h0 from As
    {
        const AccBuf io = acc_buf<2 * D>(a.ij, row0, lrow, col);
        const float bi = a.b ? a.b[col] : 0.f, bj = a.b ? a.b[D + col] : 0.f;
        const int act = a.act_j;
        FZ_FOR_ACC {
            const int r = lrow + rb * 32 + (reg & 3) + 8 * (reg >> 2);
            const float iv = bmp_sigmoid(acc[0][rb][reg] + bi);
            const float jv = bmp_act(act, acc[1][rb][reg] + bj);
            if (a.ij != nullptr) {            // the backward's input: not written in forward-only evaluation
                acc_st<2 * D>(io, rb, reg, iv, 0);
                acc_st<2 * D>(io, rb, reg, jv, D);
            }
            Al[LOFF(rb, reg)] = rws[r] * iv * jv;
        }
    }
    if (tid < FZ_R && mlist[260 + tid] >= 0) {            // compact list: entry = (first row, molecule)
        const int idx = mlist[130 + tid] + (tid >= 64 ? mlist[128] : 0);
        mlist[2 * idx] = tid;
        mlist[2 * idx + 1] = mlist[260 + tid];
    }
    __syncthreads();
    const int nm = mlist[128] + mlist[129];
    for (int task = tid; task < nm * D; task += 512) {
        const int m = task / D, c = task % D;
        const int r0 = mlist[2 * m], mol = mlist[2 * m + 1];
        const int nr = a.mol_nrows[mol];
        float sum = 0.f;
        for (int r = r0; r < r0 + nr; ++r) sum += As[r * LD + c];
        a.g[(size_t)mol * D + c] = sum;
    }
}
#undef LOFF

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static size_t fz_lds_bytes(int D) { return ((size_t)2 * FZ_R * (D + 4) + FZ_R * 4 + 132 + 2 * FZ_ECAP + FZ_NSYNC) * sizeof(float); }

// d = 64 / 128: the 512-thread tile kernels of this file; d = 32: one wave per 32-row block (bmp_fused_small.hip)
static bool fz_wide(int d) { return d == 64 || d == 128; }
int bmp_launch_step_small(bool bwd, const StepArgs& a, int n_tiles, int d, hipStream_t st);
size_t bmp_step_wgrad_small_ws_floats(int N, int d);
int bmp_launch_step_wgrad_small(const float* h, const float* m, const float* rz, const float* gda, int N, int d, int first, float* o1,
                                float* o2, float* dUcT, float* cs, int accumulate, float* ws, hipStream_t st);
extern "C" int bmp_ggnn_step_supported(int d) { return fz_wide(d) || d == 32; }

// `rows`: rows the launch works on (flop / byte accounting of the roofline leg: with a tile table the live rows, passed by the caller).
template <int D, bool FIRST, bool VAR>
static int fz_launch3(bool bwd, const StepArgs& a, int n_tiles, double rows, hipStream_t st) {
    // kind: 0 forward keeping m / rz / c, 1 forward-only evaluation, 2 backward
    const bool keep = a.m != nullptr;
    const int kind = bwd ? 2 : (keep ? 0 : 1);
    const void* fn = bwd ? (const void*)k_ggnn_step_bwd<D, FIRST, VAR>
                   : kind == 0 ? (const void*)k_ggnn_step_fwd<D, FIRST, VAR, true>
                               : (const void*)k_ggnn_step_fwd<D, FIRST, VAR, false>;
    if (int rc_attr = bmp_lds_attr(fn, (size_t)((int)fz_lds_bytes(D)))) return rc_attr;
    const double gates = FIRST ? 4.0 : 7.0;                         // d^2 MACs per row: W-part (+U)
    const double macs = 4.0 + gates;
    BmpProfScope prof(bwd ? BMP_KCLS_STEP_BWD : BMP_KCLS_STEP_FWD, 2.0 * rows * macs * D * D,
                      4.0 * rows * D * (bwd ? 13.0 : 6.0), st, FIRST ? BMP_KID_GGNN_FIRST : BMP_KID_GGNN_LATER);
    if (bwd) hipLaunchKernelGGL((k_ggnn_step_bwd<D, FIRST, VAR>), dim3(n_tiles), dim3(512), fz_lds_bytes(D), st, a);
    else if (kind == 0) hipLaunchKernelGGL((k_ggnn_step_fwd<D, FIRST, VAR, true>), dim3(n_tiles), dim3(512), fz_lds_bytes(D), st, a);
    else hipLaunchKernelGGL((k_ggnn_step_fwd<D, FIRST, VAR, false>), dim3(n_tiles), dim3(512), fz_lds_bytes(D), st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}

template <int D>
static int fz_launch(bool bwd, const StepArgs& a, int n_tiles, hipStream_t st) {
    const double rows = a.mt_row0 != nullptr ? (double)a.mt_rows : (double)n_tiles * FZ_R;
    if (a.mt_row0 != nullptr)
        return a.first ? fz_launch3<D, true, true>(bwd, a, n_tiles, rows, st) : fz_launch3<D, false, true>(bwd, a, n_tiles, rows, st);
    return a.first ? fz_launch3<D, true, false>(bwd, a, n_tiles, rows, st) : fz_launch3<D, false, false>(bwd, a, n_tiles, rows, st);
}

// One GGNN propagation step, forward (models/ggnn.py:215-263): m = message(h), h' = GRU([h, m]).
// Weight layouts as bmp_msg_fwd / bmp_gru_fwd.  Saves m [N x d], rz [N x 2d], c [N x d] -- or none of them when m, rz and c are
// all NULL (forward-only evaluation: predict under no-backprop, train_binary.py:120-127).
extern "C" int bmp_ggnn_step_fwd(const float* h, int tile0, int n_tiles, int d, int first, const int* csr_ptr,
                                 const int* csr_col, const float* csr_val, const float* WT, const float* bE, const float* AT,
                                 const float* UcT, const float* b, float* m, float* rz, float* c, float* hout,
                                 const int* mt_row0, const int* mt_nblk, int mt_rows, int tile_stride, hipStream_t st) {
    BMP_REQUIRE(tile0 >= 0 && n_tiles > 0 && bmp_ggnn_step_supported(d));
    BMP_REQUIRE(tile_stride == 0 || (tile_stride == FZ_R && mt_row0 != nullptr));
    BMP_REQUIRE((m != nullptr) == (rz != nullptr) && (m != nullptr) == (c != nullptr) && hout != nullptr);
    StepArgs a; memset(&a, 0, sizeof(a));
    BMP_REQUIRE((mt_row0 != nullptr) == (mt_nblk != nullptr));
    a.ptr = csr_ptr; a.col = csr_col; a.val = csr_val; a.first = first; a.tile0 = tile0; a.mt_row0 = mt_row0; a.mt_nblk = mt_nblk; a.mt_rows = mt_rows;
    a.h = h; a.WT = WT; a.bE = bE; a.AT = AT; a.UcT = UcT; a.b = b; a.m = m; a.rz = rz; a.c = c; a.hout = hout;
    a.tile_stride = tile_stride;
    if (!fz_wide(d)) return bmp_launch_step_small(false, a, n_tiles, d, st);
    return d == 128 ? fz_launch<128>(false, a, n_tiles, st) : fz_launch<64>(false, a, n_tiles, st);
}


// Backward-data of one step: dh (gradient w.r.t. the step input h) and gda [N x 7d] =
// [G_0..G_3 | da_r | da_z | da_c] for bmp_ggnn_step_wgrad.  Wnat [d x 4d], A [3d x 2d], Uc [d x d].
// skip_zero_g != 0: the caller reads gda's per-type blocks through the batch's row lists only (bmp_ggnn_step_wgrad with
// type_rows, when bmp_step_wgrad_lists_used): the block G_e of a row WITHOUT a bond of type e -- an exact zero, 64 % of the
// blocks of a DDI batch -- is then not written (77 of the 244 MB this launch stores: -4.5 % of its time, -11 % for a RelGCN layer).
extern "C" int bmp_ggnn_step_bwd(const float* dhout, const float* h, const float* rz, const float* c, int n_tiles, int d,
                                 int first, const int* csrT_ptr, const int* csrT_col, const float* csrT_val,
                                 const float* Wnat, const float* A, const float* Uc, float* dh, float* gda,
                                 const int* mt_row0, const int* mt_nblk, int mt_rows, int tile_stride, int skip_zero_g, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && bmp_ggnn_step_supported(d) && (mt_row0 != nullptr) == (mt_nblk != nullptr));
    StepArgs a; memset(&a, 0, sizeof(a));
    a.ptr = csrT_ptr; a.col = csrT_col; a.val = csrT_val; a.first = first; a.mt_row0 = mt_row0; a.mt_nblk = mt_nblk; a.mt_rows = mt_rows;
    a.dhout = dhout; a.h = h; a.rz = const_cast<float*>(rz); a.c = const_cast<float*>(c); a.Wnat = Wnat; a.A = A; a.Uc = Uc; a.dh = dh; a.gda = gda;
    BMP_REQUIRE(tile_stride == 0 || (tile_stride == FZ_R && mt_row0 != nullptr));
    a.tile_stride = tile_stride;
    a.skip_zero_g = skip_zero_g && fz_wide(d);
    if (!fz_wide(d)) return bmp_launch_step_small(true, a, n_tiles, d, st);
    return d == 128 ? fz_launch<128>(true, a, n_tiles, st) : fz_launch<64>(true, a, n_tiles, st);
}

// The weight-gradient problems of one step as ONE fused launch (bmp_launch_wgrad_fused): all of them reduce over the
// same N rows and read column ranges of the same gda rows.
//   later steps:  o1 = h^T gda [d x 7d] (+ cs) | o2 = m^T gda[:, 4d:7d] [d x 3d] | dUcT = (r*h)^T gda[:, 6d:7d] [d x d]
//   first step (no r gate, no U term: da_r = 0 and is neither written nor read): o1 skips columns [4d, 5d), o2 skips
//   [0, d), dUcT is zero -- 6d + 2d instead of 7d + 3d + d columns of products.
// type_rows [4 x N] / type_cnt [4] (bmp_type_rows of the TRANSPOSED CSR; null: none): the rows whose gathered gradient G_e is
// not zero.  With them the per-type blocks of o1 become four problems of their own that walk their lists only -- on the DDI
// batches 1.46 N rows instead of 4 N, 8.5 instead of 11 column tiles of work for the launch -- and o1's da columns a fifth.
static const float kTypeFrac[4] = {0.78f, 0.24f, 0.05f, 0.58f};      // expected shares (single, double, triple, aromatic): balance only
static int step_wgrad_problems(WGArgs* g, const float* h, const float* m, const float* rz, const float* gda, int N, int d,
                               int first, float* o1, float* o2, float* dUcT, float* cs, int accumulate,
                               const int* type_rows = nullptr, const int* type_cnt = nullptr, const int* live_rows = nullptr,
                               const int* live_cnt = nullptr) {
    g[1] = WGArgs{m, nullptr, d, 0, gda + 4 * d, 7 * d, d, first ? 2 * d : 3 * d, N, o2, 3 * d, accumulate};
    g[2] = WGArgs{rz, h, 2 * d, d, gda + 6 * d, 7 * d, d, d, N, dUcT, d, accumulate};
    if (first) {
        g[1].skip_at = 0; g[1].skip_n = d;
        g[2].zero_only = 1;
    }
    if (type_rows == nullptr) {
        g[0] = WGArgs{h, nullptr, d, 0, gda, 7 * d, d, first ? 6 * d : 7 * d, N, o1, 7 * d, accumulate, cs};
        if (first) { g[0].skip_at = 4 * d; g[0].skip_n = d; }
        return 3;
    }
    g[0] = WGArgs{h, nullptr, d, 0, gda + 4 * d, 7 * d, d, first ? 2 * d : 3 * d, N, o1 + 4 * d, 7 * d, accumulate, cs + 4 * d};
    if (first) { g[0].skip_at = 0; g[0].skip_n = d; }
    for (int e = 0; e < 4; ++e) {
        g[3 + e] = WGArgs{h, nullptr, d, 0, gda + e * d, 7 * d, d, d, N, o1 + e * d, 7 * d, accumulate, cs + e * d};
        g[3 + e].ridx = type_rows + (size_t)e * N; g[3 + e].rcnt = type_cnt + e; g[3 + e].rfrac = kTypeFrac[e];
    }
    // a batch with few live rows (tiles at a fixed stride): the gate blocks through the list of the molecules' rows.  (Not the
    // first call's form -- its skipped da_r columns and a list do not combine in the kernel -- nor dUcT's product operand.)
    if (live_rows && live_cnt && !first)
        for (int q = 0; q < 2; ++q) { g[q].ridx = live_rows; g[q].rcnt = live_cnt; g[q].rfrac = 0.3f; }
    return 7;
}

static bool step_wgrad_fusable(int N, int d) { return (d == 64 || d == 128) && (N & 31) == 0; }
// Will bmp_ggnn_step_wgrad / bmp_relgcn_layer_wgrad, handed row lists, read gda's per-type blocks through them (and through
// them only)?  The caller's licence for skip_zero_g of the backward launches.
extern "C" int bmp_step_wgrad_lists_used(int N, int d) {
    static const bool unfused = getenv("BMP_STEP_WGRAD_UNFUSED") != nullptr;
    return !unfused && step_wgrad_fusable(N, d) && bmp_wgrad_fused_lists_ok(N);
}

extern "C" size_t bmp_ggnn_step_wgrad_ws_floats(int N, int d) {
    size_t a = bmp_wgrad_ws_floats(N, d, 7 * d);
    if (d == 32 && (N & 7) == 0) { const size_t b = bmp_step_wgrad_small_ws_floats(N, d); if (b > a) a = b; }
    if (step_wgrad_fusable(N, d)) {
        WGArgs g[BMP_WG_MAXP];
        int n = step_wgrad_problems(g, nullptr, nullptr, nullptr, nullptr, N, d, 0, nullptr, nullptr, nullptr, (float*)16, 0);
        size_t b = bmp_wgrad_fused_ws_floats(g, n);
        if (b > a) a = b;
        n = step_wgrad_problems(g, nullptr, nullptr, nullptr, nullptr, N, d, 0, nullptr, nullptr, nullptr, (float*)16, 0, (const int*)16, (const int*)16);
        b = bmp_wgrad_fused_ws_floats(g, n);
        if (b > a) a = b;
        n = step_wgrad_problems(g, nullptr, nullptr, nullptr, nullptr, N, d, 0, nullptr, nullptr, nullptr, (float*)16, 0, (const int*)16, (const int*)16,
                                (const int*)16, (const int*)16);
        b = bmp_wgrad_fused_ws_floats(g, n);
        if (b > a) a = b;
    }
    return a;
}

// Weight gradients of one step (reduction over all N = n_tiles*128 rows):
//   o1 [d x 7d]  = h^T . gda        cols [0,4d): dWT as [k][e*d + c];  cols [4d,7d): dAT rows 0..d-1
//   o2 [d x 3d]  = m^T . da         = dAT rows d..2d-1
//   dUcT [d x d] = (r*h)^T . da_c   (zeros when first)
//   cs [7d]      = column sums of gda: [dbE as e*d + c | db]
// accumulate != 0 adds into the outputs (weight tying: one set of buffers for all steps).
// first != 0: the da_r columns of gda are not read (bmp_ggnn_step_bwd does not write them) and count as zeros.
// type_rows / type_cnt: optional row lists of the batch's TRANSPOSED CSR (bmp_type_rows): the per-type blocks then sum over
// the rows that have a bond of the type only (the others' G_e rows are exact zeros: same sums, fewer products).
extern "C" int bmp_ggnn_step_wgrad(const float* h, const float* m, const float* rz, const float* gda, int N, int d,
                                   int first, float* o1, float* o2, float* dUcT, float* cs, int accumulate,
                                   const int* type_rows, const int* type_cnt, const int* live_rows, const int* live_cnt,
                                   float* ws, size_t ws_floats, hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && ws_floats >= bmp_ggnn_step_wgrad_ws_floats(N, d));
    BMP_REQUIRE((live_rows != nullptr) == (live_cnt != nullptr));
    BMP_REQUIRE(h && m && rz && gda && o1 && o2 && dUcT && cs && ws);
    static const bool unfused = getenv("BMP_STEP_WGRAD_UNFUSED") != nullptr;        // A/B switch (tools, tests)
    if (!unfused && d == 32 && (N & 7) == 0 && ((uintptr_t)gda & 15) == 0)          // one 32-row MFMA block per output: bmp_fused_small.hip
        return bmp_launch_step_wgrad_small(h, m, rz, gda, N, d, first, o1, o2, dUcT, cs, accumulate, ws, st);
    if (!unfused && step_wgrad_fusable(N, d) && ((uintptr_t)h & 15) == 0 && ((uintptr_t)m & 15) == 0 && ((uintptr_t)rz & 15) == 0 &&
        ((uintptr_t)gda & 15) == 0) {
        WGArgs g[BMP_WG_MAXP];
        const bool lists = type_rows != nullptr && type_cnt != nullptr && bmp_wgrad_fused_lists_ok(N);
        const int n = step_wgrad_problems(g, h, m, rz, gda, N, d, first, o1, o2, dUcT, cs, accumulate, lists ? type_rows : nullptr,
                                          lists ? type_cnt : nullptr, lists ? live_rows : nullptr, lists ? live_cnt : nullptr);
        return bmp_launch_wgrad_fused(g, n, ws, st, BMP_KID_WGRAD_STEP);
    }
    // one launch per product (first steps: the da_r columns of gda are not written, so they are zeroed here first)
    int rc;
    if (first) {
        hipError_t e = hipMemset2DAsync(const_cast<float*>(gda) + 4 * d, (size_t)7 * d * sizeof(float), 0, (size_t)d * sizeof(float), N, st);
        if (e != hipSuccess) return (int)e;
    }
    WGArgs g1{h, nullptr, d, 0, gda, 7 * d, d, 7 * d, N, o1, 7 * d, accumulate, cs};     // + column sums of gda
    if ((rc = bmp_launch_wgrad(g1, ws, st))) return rc;
    WGArgs g2{m, nullptr, d, 0, gda + 4 * d, 7 * d, d, 3 * d, N, o2, 3 * d, accumulate};
    if ((rc = bmp_launch_wgrad(g2, ws, st))) return rc;
    if (first) {
        if (!accumulate) { hipError_t e = hipMemsetAsync(dUcT, 0, (size_t)d * d * sizeof(float), st); if (e != hipSuccess) return (int)e; }
        return 0;
    }
    WGArgs g3{rz, h, 2 * d, d, gda + 6 * d, 7 * d, d, d, N, dUcT, d, accumulate};
    return bmp_launch_wgrad(g3, ws, st);
}

// ---- fused RelGCN layer (d_in == d_out in {64, 128}) ----
extern "C" int bmp_relgcn_layer_supported(int d_in, int d_out) { return d_in == d_out && fz_wide(d_in); }

template <int D, bool VAR>
static int rel_launch2(bool bwd, const RelArgs& a, int n_tiles, hipStream_t st) {
    const void* fn = bwd ? (const void*)k_relgcn_layer_bwd<D, VAR> : (const void*)k_relgcn_layer_fwd<D, VAR>;
    if (int rc_attr = bmp_lds_attr(fn, (size_t)((int)fz_lds_bytes(D)))) return rc_attr;
    const double rows = VAR ? (double)a.mt_rows : (double)n_tiles * FZ_R;
    BmpProfScope prof(bwd ? BMP_KCLS_STEP_BWD : BMP_KCLS_STEP_FWD, 2.0 * rows * 5.0 * D * D, 4.0 * rows * D * (bwd ? 8.0 : 2.0), st,
                      BMP_KID_RELGCN);
    if (bwd) hipLaunchKernelGGL((k_relgcn_layer_bwd<D, VAR>), dim3(n_tiles), dim3(512), fz_lds_bytes(D), st, a);
    else hipLaunchKernelGGL((k_relgcn_layer_fwd<D, VAR>), dim3(n_tiles), dim3(512), fz_lds_bytes(D), st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}
template <int D>
static int rel_launch(bool bwd, const RelArgs& a, int n_tiles, hipStream_t st) {
    return a.mt_row0 != nullptr ? rel_launch2<D, true>(bwd, a, n_tiles, st) : rel_launch2<D, false>(bwd, a, n_tiles, st);
}

// out = act(h . WsT + bs + sum_e gather_e(h) . WT_e + wdeg_e * bE_e); WT [4d x d] and WsT [d x d] K4-packed
// (bmp/functional.py:pack_k4).  Saves wdeg [N x 4].
extern "C" int bmp_relgcn_layer_fwd(const float* h, int tile0, int n_tiles, int d, const int* csr_ptr, const int* csr_col,
                                    const float* csr_val, const float* WT, const float* bE, const float* WsT, const float* bs,
                                    int act, float* out, float* wdeg, const int* mt_row0, const int* mt_nblk, int mt_rows, hipStream_t st) {
    BMP_REQUIRE(tile0 >= 0 && n_tiles > 0 && fz_wide(d) && (mt_row0 != nullptr) == (mt_nblk != nullptr));
    RelArgs a; memset(&a, 0, sizeof(a));
    a.ptr = csr_ptr; a.col = csr_col; a.val = csr_val; a.act = act; a.tile0 = tile0; a.mt_row0 = mt_row0; a.mt_nblk = mt_nblk; a.mt_rows = mt_rows;
    a.h = h; a.WT = WT; a.bE = bE; a.WsT = WsT; a.bs = bs; a.out = out; a.wdeg = wdeg;
    return d == 128 ? rel_launch<128>(false, a, n_tiles, st) : rel_launch<64>(false, a, n_tiles, st);
}

// dh and gda [N x 5d] = [G_0..G_3 | dpre] for bmp_relgcn_layer_wgrad.  Wnat [d x 4d] = WT^T, Ws [d x d] = WsT^T, K4-packed.
extern "C" int bmp_relgcn_layer_bwd(const float* dout, const float* out, int act, int n_tiles, int d, const int* csrT_ptr,
                                    const int* csrT_col, const float* csrT_val, const float* Wnat, const float* Ws, float* dh,
                                    float* gda, const int* mt_row0, const int* mt_nblk, int mt_rows, int skip_zero_g, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && fz_wide(d) && (mt_row0 != nullptr) == (mt_nblk != nullptr));
    RelArgs a; memset(&a, 0, sizeof(a));
    a.ptr = csrT_ptr; a.col = csrT_col; a.val = csrT_val; a.act = act; a.mt_row0 = mt_row0; a.mt_nblk = mt_nblk; a.mt_rows = mt_rows;
    a.dout = dout; a.y = out; a.Wnat = Wnat; a.Ws = Ws; a.dh = dh; a.gda = gda; a.skip_zero_g = skip_zero_g;
    return d == 128 ? rel_launch<128>(true, a, n_tiles, st) : rel_launch<64>(true, a, n_tiles, st);
}

static int rel_wgrad_problem(WGArgs* g, const float* h, const float* wdeg, const float* gda, int N, int d, float* o1, float* dbE,
                             float* cs, int accumulate, const int* type_rows = nullptr, const int* type_cnt = nullptr) {
    if (type_rows == nullptr) {
        g[0] = WGArgs{h, nullptr, d, 0, gda, 5 * d, d, 5 * d, N, o1, 5 * d, accumulate, cs};
        g[0].wrow = wdeg; g[0].w_col0 = 4 * d; g[0].wout = dbE; g[0].ldwo = d;      // dbE = wdeg^T . dpre rides along
        return 1;
    }
    // the self-connection block (dpre, every row) and the four per-type blocks over their row lists (see step_wgrad_problems)
    g[0] = WGArgs{h, nullptr, d, 0, gda + 4 * d, 5 * d, d, d, N, o1 + 4 * d, 5 * d, accumulate, cs + 4 * d};
    g[0].wrow = wdeg; g[0].w_col0 = 0; g[0].wout = dbE; g[0].ldwo = d;
    for (int e = 0; e < 4; ++e) {
        g[1 + e] = WGArgs{h, nullptr, d, 0, gda + e * d, 5 * d, d, d, N, o1 + e * d, 5 * d, accumulate, cs + e * d};
        g[1 + e].ridx = type_rows + (size_t)e * N; g[1 + e].rcnt = type_cnt + e; g[1 + e].rfrac = kTypeFrac[e];
    }
    return 5;
}

extern "C" size_t bmp_relgcn_layer_wgrad_ws_floats(int N, int d) {
    size_t a = bmp_wgrad_ws_floats(N, d, 5 * d), b = bmp_wgrad_ws_floats(N, 4, d);
    a = a > b ? a : b;
    if (step_wgrad_fusable(N, d)) {
        WGArgs g[BMP_WG_MAXP];
        int n = rel_wgrad_problem(g, nullptr, (const float*)16, nullptr, N, d, nullptr, (float*)16, (float*)16, 0);
        b = bmp_wgrad_fused_ws_floats(g, n);
        if (b > a) a = b;
        n = rel_wgrad_problem(g, nullptr, (const float*)16, nullptr, N, d, nullptr, (float*)16, (float*)16, 0, (const int*)16, (const int*)16);
        b = bmp_wgrad_fused_ws_floats(g, n);
        if (b > a) a = b;
    }
    return a;
}

// Weight gradients of one layer (reduction over all N rows), ONE GEMM launch + ONE reduction:
//   o1 [d x 5d] = h^T . gda     cols [0,4d): dWT as [k][e*d + c];  cols [4d,5d): dWsT
//   dbE [4 x d] = wdeg^T . dpre (weighted column sums of the dpre tile, carried by the same launch)
//   cs [5d]     = column sums of gda; cs[4d:] = dbs
extern "C" int bmp_relgcn_layer_wgrad(const float* h, const float* wdeg, const float* gda, int N, int d, float* o1, float* dbE,
                                      float* cs, int accumulate, const int* type_rows, const int* type_cnt, float* ws, size_t ws_floats,
                                      hipStream_t st) {
    BMP_REQUIRE(N > 0 && d > 0 && ws_floats >= bmp_relgcn_layer_wgrad_ws_floats(N, d));
    BMP_REQUIRE(h && wdeg && gda && o1 && dbE && cs && ws);
    static const bool unfused = getenv("BMP_STEP_WGRAD_UNFUSED") != nullptr;        // A/B switch (tools, tests)
    if (!unfused && step_wgrad_fusable(N, d) && ((uintptr_t)h & 15) == 0 && ((uintptr_t)gda & 15) == 0 && ((uintptr_t)wdeg & 15) == 0) {
        WGArgs g[BMP_WG_MAXP];
        const bool lists = type_rows != nullptr && type_cnt != nullptr && bmp_wgrad_fused_lists_ok(N);
        const int n = rel_wgrad_problem(g, h, wdeg, gda, N, d, o1, dbE, cs, accumulate, lists ? type_rows : nullptr, lists ? type_cnt : nullptr);
        return bmp_launch_wgrad_fused(g, n, ws, st, BMP_KID_WGRAD_STEP);
    }
    int rc;
    WGArgs g1{h, nullptr, d, 0, gda, 5 * d, d, 5 * d, N, o1, 5 * d, accumulate, cs};
    if ((rc = bmp_launch_wgrad(g1, ws, st))) return rc;
    WGArgs g2{wdeg, nullptr, 4, 0, gda + 4 * d, 5 * d, 4, d, N, dbE, d, accumulate};
    return bmp_launch_wgrad(g2, ws, st);
}

// ---- readout forward on the tile machinery (d == o in {64, 128}; h0 absent or d wide) ----
extern "C" int bmp_readout_tile_supported(int d, int d0, int o) {
    return fz_wide(d) && o == d && (d0 == 0 || d0 == d);
}

static size_t ro_lds_bytes(int D) { return ((size_t)2 * FZ_R * (D + 4) + 400 + FZ_R) * sizeof(float); }

template <int D, bool HAS0>
static int ro_launch(const ROArgs& a, int n_tiles, hipStream_t st) {
    if (int rc_attr = bmp_lds_attr((const void*)k_readout_tile_fwd<D, HAS0>, (size_t)((int)ro_lds_bytes(D)))) return rc_attr;
    const double rows = (double)n_tiles * FZ_R;
    BmpProfScope prof(BMP_KCLS_ROWGEMM, 2.0 * rows * (HAS0 ? 2.0 : 1.0) * D * 2.0 * D, 4.0 * rows * D * (HAS0 ? 4.0 : 3.0), st,
                      BMP_KID_READOUT_TILE);
    hipLaunchKernelGGL((k_readout_tile_fwd<D, HAS0>), dim3(n_tiles), dim3(512), ro_lds_bytes(D), st, a);
    BMP_LAUNCH_CHECK();
    return 0;
}

// As bmp_readout_fwd, with WT K4-packed and the row -> molecule map of the packed batch (row_mol [N], -1 = no molecule).
extern "C" int bmp_readout_tile_fwd(const float* h, const float* h0, int n_tiles, int d, const float* WT, const float* b,
                                    int act_j, const float* row_w, const int* row_mol, const int* mol_nrows, float* ij,
                                    float* g, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && fz_wide(d) && row_mol != nullptr);
    ROArgs a{h, h0, WT, b, act_j, row_w, row_mol, mol_nrows, ij, g};
    if (d == 128) return h0 ? ro_launch<128, true>(a, n_tiles, st) : ro_launch<128, false>(a, n_tiles, st);
    return h0 ? ro_launch<64, true>(a, n_tiles, st) : ro_launch<64, false>(a, n_tiles, st);
}
