// Tile machinery shared by the fused tile kernels (bmp_fused.hip: d = 64 / 128, 512 threads per 128-row tile; bmp_fused_small.hip:
// d = 32, one wave per 32-row block): the step's argument block, the LDS-A x streamed-B MFMA loop with its B-fragment
// prefetch, the tile-local neighbour gather, the CSR staging, the half-tile group counters and the buffer-resource views.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include <string.h>
#include "bmp_kernels.h"

#define FZ_R 128

struct StepArgs {
    // graph
    const int* ptr; const int* col; const float* val;      // CSR (fwd) or transposed CSR (bwd)
    int tile0;                      // the launch covers tiles tile0 .. tile0 + gridDim.x - 1 (all arrays whole)
    int first;
    // optional tile table (nullptr: tile t = rows [128 t, 128 t + 128)): tile t = rows [mt_row0[t], + 32 * mt_nblk[t]),
    // mt_nblk in 1..4 -- tiles of fewer live 32-row blocks skip the dead blocks' gathers, MFMAs, loads and stores
    const int* mt_row0; const int* mt_nblk;
    int mt_rows;                    // rows of the launch's tiles when a table is given (host-side accounting only)
    // forward
    const float* h;                 // [N x D] step input
    const float* WT;                // [4D x D]  message weights, K-major (row e*D + k, col c)
    const float* bE;                // [4 x D]
    const float* AT;                // [2D x 3D] gate weights K-major, rows [h ; m], cols [r | z | c]
    const float* UcT;               // [D x D]
    const float* b;                 // [3D]
    float* m; float* rz; float* c; float* hout;
    // backward
    const float* dhout;             // [N x D]
    const float* Wnat;              // [D x 4D]  (= WT^T: row c, col e*D + k)
    const float* A;                 // [3D x 2D] (= AT^T)
    const float* Uc;                // [D x D]   (= UcT^T, reference layout)
    float* dh;                      // [N x D]
    float* gda;                     // [N x 7D]: G (4D: gathered dm per bond type) | da_r | da_z | da_c
    int skip_zero_g;
    // tile table at a FIXED stride (bmp.packed.StaticPairBatch: tile t starts at row t * tile_stride, its mt_nblk[t] live blocks
    // hold its molecule): the rows of the tile's other blocks are cleared in every array the launch writes, so that what an
    // earlier batch left there never enters a GEMM over all rows.  0: dense rows (tile t + 1 follows tile t).
    int tile_stride;
};

// rows [r_begin, r_end) of a row-major [.. x W] array := 0 (all NT threads of the workgroup; W a multiple of 4)
template <int NT>
__device__ __forceinline__ void fz_clear_rows(float* p, int W, int r_begin, int r_end, int tid) {
    if (p == nullptr) return;
    const int w4 = W >> 2;
    f32x4* q = (f32x4*)(p + (size_t)r_begin * W);
    const int n = (r_end - r_begin) * w4;
    for (int i = tid; i < n; i += NT) q[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// acc[nb][rb] += A(rows of this wave, K) . B_nb(K, 32 cols)   with A in LDS, B streamed from global.
//   As_wave = &tile[(wave_row0 + (lane & 31)) * LD + 4 * (lane >> 5)]
//   Bp[nb]  = B_nb + 4 * (lane >> 5) * ldw[nb] + col      (col = this lane's output column)
// `rot` (multiple of 8, < K) rotates the K loop: workgroups walk the shared weight matrices from different
// starting rows, so the 256 CUs do not all request the same L2 lines at the same moment.
// The first two B fragments of a tile_mma call, requested early: every call otherwise opens with an L2 round trip
// during which the matrix pipe has nothing to do (7-8 calls per tile).  The caller issues the prefetch before the
// gather / epilogue / group barrier that precedes the call.
template <int NB>
struct BPre { f32x4 b0[NB], b1[NB]; };
template <int NB>
__device__ __forceinline__ void tile_b_prefetch(BPre<NB>& p, const float* const (&Bp)[NB], const int (&ldw)[NB], int K, int rot) {
    int k1 = rot + 8; if (k1 >= K) k1 -= K;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        p.b0[nb] = *(const f32x4*)(Bp[nb] + (size_t)rot * ldw[nb]);
        p.b1[nb] = *(const f32x4*)(Bp[nb] + (size_t)k1 * ldw[nb]);
    }
}

// NRB (<= RB): live row blocks of this wave (a short tile of the encoder layout); the other blocks' A loads and MFMAs do not
// exist in that instance.
// aoff (optional): float offset of row block rb's A rows from As_wave, per lane -- rows taken in another order than they lie
// in the tile (the type-sorted order of a half, fz_row_order); null: block rb = rows + 32 rb.
template <int NB, int RB, int NRB = RB>
__device__ __forceinline__ void tile_mma(f32x16 (&acc)[NB][RB], const float* As_wave, int LD, const float* const (&Bp)[NB],
                                         const int (&ldw)[NB], int K, int rot, const BPre<NB>* pre = nullptr, const int* aoff = nullptr) {
    // B fragments run two k-steps ahead of the MFMAs (register ring b0 <- b1 <- b2); the load of step s+2 is
    // issued, and pinned by a scheduling barrier, BEFORE the MFMAs of step s, so an L2 round trip hides under
    // two steps of matrix work.  The loop wraps (k mod K), so the look-ahead loads are always in range.
    f32x4 b0[NB], b1[NB], b2[NB];
    int k = rot;
    int k1 = k + 8; if (k1 >= K) k1 -= K;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (pre) { b0[nb] = pre->b0[nb]; b1[nb] = pre->b1[nb]; }
        else {
            b0[nb] = *(const f32x4*)(Bp[nb] + (size_t)k * ldw[nb]);
            b1[nb] = *(const f32x4*)(Bp[nb] + (size_t)k1 * ldw[nb]);
        }
    }
    f32x4 a0[NRB], a1[NRB];          // A fragments (LDS) run one k-step ahead
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) a0[rb] = *(const f32x4*)(As_wave + (aoff ? aoff[rb] : rb * 32 * LD) + k);
    for (int it = 0; it < K; it += 8) {
        int k2 = k1 + 8; if (k2 >= K) k2 -= K;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b2[nb] = *(const f32x4*)(Bp[nb] + (size_t)k2 * ldw[nb]);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a1[rb] = *(const f32x4*)(As_wave + (aoff ? aoff[rb] : rb * 32 * LD) + k1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb][rb] = bmp_mfma(a0[rb][t], b0[nb][t], acc[nb][rb]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { b0[nb] = b1[nb]; b1[nb] = b2[nb]; }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a0[rb] = a1[rb];
        k = k1; k1 = k2;
    }
}

// tile_mma for a wave with `nrb` live row blocks (0: nothing to do).  VAR == false: whole tiles, nrb == RB at compile time.
template <bool VAR, int NB, int RB>
__device__ __forceinline__ void tile_mma_n(int nrb, f32x16 (&acc)[NB][RB], const float* As_wave, int LD, const float* const (&Bp)[NB],
                                           const int (&ldw)[NB], int K, int rot, const BPre<NB>* pre = nullptr, const int* aoff = nullptr) {
    if constexpr (!VAR) {
        tile_mma<NB, RB, RB>(acc, As_wave, LD, Bp, ldw, K, rot, pre, aoff);
    } else if constexpr (RB == 1) {
        if (nrb > 0) tile_mma<NB, 1, 1>(acc, As_wave, LD, Bp, ldw, K, rot, pre, aoff);
    } else {
        if (nrb == RB) tile_mma<NB, RB, RB>(acc, As_wave, LD, Bp, ldw, K, rot, pre, aoff);
        else if (nrb > 0) tile_mma<NB, RB, 1>(acc, As_wave, LD, Bp, ldw, K, rot, pre, aoff);
    }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

// Tile-local neighbour gather for one bond type: dst[row, :] = sum over CSR entries of `row` with type e of
// val * src_tile[col_local, :].  4 threads per row, D/4 columns each.  Returns (per thread) whether it saw a
// matching entry; *wsum gets the row's weighted degree for that type.
// ptr is indexed by the tile-local row; col/val by the entry index ptr yields (either the kernel's global CSR
// arrays, or the copy of the tile's entries staged in LDS -- see stage_csr).
template <int D>
__device__ __forceinline__ bool tile_gather(const float* src_tile, float* dst_tile, int LD, const int* ptr, const int* col,
                                            const float* val, int row0, int e, float* wsum, int* tmask, int nrows = FZ_R,
                                            const unsigned char* inv = nullptr) {
    constexpr int F = D / 16;                 // float4 per thread
    const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
    f32x4 acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wd = 0.f;
    bool any = false;
    *wsum = 0.f;
    if (nrows < FZ_R && row >= nrows) return false;       // a dead block of a short tile: nothing reads its rows
    const int e0 = ptr[row], e1 = ptr[row + 1];
    for (int ed = e0; ed < e1; ++ed) {
        const int cv = col[ed];
        *tmask |= 1 << (cv & 3);
        if ((cv & 3) == e) {
            const float v = val[ed];
            const float* s = src_tile + ((cv >> 2) - row0) * LD + q * (D / 4);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += *(const f32x4*)(s + 4 * f) * v;
            wd += v;
            any = true;
        }
    }
    // (inv: the row goes to its position in the half's type-sorted order, fz_row_order)
    float* o = dst_tile + (inv ? (row & 64) + inv[row] : row) * LD + q * (D / 4);
#pragma unroll
    for (int f = 0; f < F; ++f) *(f32x4*)(o + 4 * f) = acc[f];
    *wsum = wd;
    return any;
}

// The tile's CSR entries -> LDS (the gather loops are chains of dependent loads: from L2 they cost several
// microseconds per bond-type pass).  Returns false (and stages nothing) if the tile has more than FZ_ECAP entries.
#define FZ_ECAP 1024
__device__ __forceinline__ bool stage_csr(const int* ptr, const int* col, const float* val, int row0, int* rptr, int* ecol,
                                          float* evalv, int nrows = FZ_R, int nt = 512) {
    const int ebase = ptr[row0];
    const int ne = ptr[row0 + nrows] - ebase;
    if (ne > FZ_ECAP) return false;
    for (int i = threadIdx.x; i <= nrows; i += nt) rptr[i] = ptr[row0 + i] - ebase;
    for (int i = threadIdx.x; i < ne; i += nt) { ecol[i] = col[ebase + i]; evalv[i] = val[ebase + i]; }
    return true;
}
#define FZ_GATHER(srcT, dstT, e, wdp) (csr_lds ? tile_gather<D>(srcT, dstT, LD, rptr, ecol, evalv, row0, e, wdp, &tmask, nrows, ro.inv) \
                                               : tile_gather<D>(srcT, dstT, LD, a.ptr + row0, a.col, a.val, row0, e, wdp, &tmask, nrows, ro.inv))
#define FZ_ROW_ORDER() fz_row_order(sy, gs, grp, w, lane, csr_lds ? rptr : a.ptr + row0, csr_lds ? ecol : a.col, nrows)

// ---- half-tile groups --------------------------------------------------------------------------------------
// Waves 0-3 own rows [0, 64) of the tile and waves 4-7 rows [64, 128) in every phase (gather rows, MFMA A rows,
// epilogue rows), so between the few points where a phase reads the WHOLE tile the two halves are independent.
// They synchronise separately, on a monotonic LDS counter per group (gfx950 has one hardware barrier per
// workgroup), and group 0 runs at a higher wave priority: with one wave of each group on every SIMD, group 0
// takes the matrix pipe whenever it wants it and group 1 fills the gaps group 0 leaves while it gathers, runs
// epilogues or waits for memory.  In lockstep (one barrier for all eight waves) both waves of a SIMD sit in their
// non-MFMA phases at the same time and the matrix pipe idles for a third of the tile's life.
struct GrpSync {
    int* ctr;       // LDS, zeroed before the first workgroup barrier
    int target;
};
__device__ __forceinline__ void grp_sync(GrpSync& g) {
    g.target += 4;                                   // four waves per group
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(g.ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(g.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < g.target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define FZ_NSYNC 72         // ints of LDS: ctr[2] at +0,+1 ; type masks [2] at +2,+3 ; rare-row counts [2] at +4,+5 ;
                            // row -> position [128 bytes] at +8 ; position -> row [128 bytes] at +40  (fz_row_order)

// ---- type-sorted row order of a half tile, for the message phases (round 4) -------------------------------------
// A row's gathered operand for bond type e (AGG_e forward, G_e backward) is zero unless the row has a bond of that type, and
// double / triple bonds are rare: 19 % / 2 % of the rows, ~12 + 1 of a 64-row half tile -- yet with rows in their packed
// order every 32-row MFMA block holds one, so both blocks of the half were multiplied for every type (the per-type skip
// only fires when a type is absent from the whole half).  For the message phases the half's rows are therefore taken in the
// order (has a double or triple bond, row): the rare types' rows fill the first positions and their MFMAs cover
// ceil(count / 32) blocks -- one instead of two -- while the accumulators of all four types still line up (one order for
// all).  The gather writes its operand rows at their positions, the MFMA loop is unchanged, and the one epilogue that
// leaves the phase (m forward, dh backward) looks its rows up.  On the DDI batches: 5.5 instead of 6.9 blocks per half and
// step, 7 % of the step kernels' MFMAs (measured as an upper bound by skipping the second block blindly: C2 +3.3 %, C3
// +12 %).
#define FZ_RARE_TYPES 0x6       // bit e: type e is a rare type (1 = double, 2 = triple)
struct RowOrder {
    const unsigned char* inv;   // [128] tile row -> position inside its half (0..63)
    const unsigned char* perm;  // [128] half * 64 + position -> row inside the half
    int nbr;                    // 32-row blocks of this half that hold rows with a rare bond type (0, 1 or 2)
};
// Called by every wave of a live group once the tile's CSR is visible; wave 4 g (one lane per row of the half) ranks the
// rows; ends with the group's own sync.
__device__ __forceinline__ RowOrder fz_row_order(int* sy, GrpSync& gs, int grp, int w, int lane, const int* ptr_local, const int* col,
                                                 int nrows) {
    unsigned char* inv = (unsigned char*)(sy + 8);
    unsigned char* perm = (unsigned char*)(sy + 40);
    if ((w & 3) == 0) {
        const int r = grp * 64 + lane;
        int m = 0;
        if (r < nrows)
            for (int ed = ptr_local[r]; ed < ptr_local[r + 1]; ++ed) m |= 1 << (col[ed] & 3);
        const bool rare = (m & FZ_RARE_TYPES) != 0;
        const unsigned long long bal = __ballot(rare), lt = (1ull << lane) - 1ull;
        const int nr = __popcll(bal);
        const int pos = rare ? __popcll(bal & lt) : nr + __popcll(~bal & lt);
        inv[r] = (unsigned char)pos;
        perm[grp * 64 + pos] = (unsigned char)lane;
        if (lane == 0) sy[4 + grp] = nr;
    }
    grp_sync(gs);
    return RowOrder{inv, perm, (sy[4 + grp] + 31) >> 5};
}
// MFMAs of one bond type's message product on this wave's rows: every live block, or -- a rare type -- the blocks that hold
// its rows.  bi: the wave's block inside its half when RB == 1.
template <bool VAR, int NB, int RB>
__device__ __forceinline__ void tile_mma_msg(int nrb, int e, int nbr, int bi, f32x16 (&acc)[NB][RB], const float* As_wave, int LD,
                                             const float* const (&Bp)[NB], const int (&ldw)[NB], int K, int rot, const BPre<NB>* pre) {
    const bool rare = (FZ_RARE_TYPES >> e) & 1;
    if constexpr (RB == 1) {
        if (nrb > 0 && (!rare || bi < nbr)) tile_mma<NB, 1, 1>(acc, As_wave, LD, Bp, ldw, K, rot, pre);
    } else {
        const int nb = rare ? (nbr < nrb ? nbr : nrb) : nrb;
        if (nb >= RB) tile_mma<NB, RB, RB>(acc, As_wave, LD, Bp, ldw, K, rot, pre);
        else if (nb > 0) tile_mma<NB, RB, 1>(acc, As_wave, LD, Bp, ldw, K, rot, pre);
    }
}

// Accumulator-layout access to a row-major [rows x LDC] f32 array through a buffer resource: all 16*RB
// positions of a wave share ONE 32-bit voffset VGPR (the lane's (row, col) byte offset); the per-register
// row offset is a compile-time soffset/immediate.  (Plain pointers cost a 64-bit address pair per element
// here, which the register allocator keeps alive across the MFMA phases and spills.)
struct AccBuf {
    __amdgpu_buffer_rsrc_t rs;
    int vo;
};
template <int LDC>
__device__ __forceinline__ AccBuf acc_buf(const float* base, int tile_row0, int lane_row, int col) {
    AccBuf b;
    b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)tile_row0 * LDC), 0, 0x7FFFFFFF, 0x00020000);
    b.vo = (lane_row * LDC + col) * 4;
    return b;
}
template <int LDC>
__device__ __forceinline__ float acc_ld(const AccBuf& b, int rb, int reg, int coff = 0) {
    const int so = ((rb * 32 + (reg & 3) + 8 * (reg >> 2)) * LDC + coff) * 4;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.rs, b.vo, so, 0));
}
template <int LDC>
__device__ __forceinline__ void acc_st(const AccBuf& b, int rb, int reg, float v, int coff = 0) {
    const int so = ((rb * 32 + (reg & 3) + 8 * (reg >> 2)) * LDC + coff) * 4;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.rs, b.vo, so, 0);
}

// Row-major access to one 64-row half of a tile, 16 bytes per lane: slot v of thread tg (0..255) is the
// float4 at (half row v * (1024 / D) + tg / (D/4), float4 column tg % (D/4)); the per-slot row offset is a
// compile-time soffset, so all slots of one array share one voffset VGPR (see AccBuf).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int D, int LDP>
__device__ __forceinline__ AccBuf rm_buf(const float* base, int tile_row0, int grp, int tg) {
    AccBuf b;
    b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)tile_row0 * LDP), 0, 0x7FFFFFFF, 0x00020000);
    b.vo = ((grp * 64 + tg / (D / 4)) * LDP + 4 * (tg % (D / 4))) * 4;
    return b;
}
template <int D, int LDP>
__device__ __forceinline__ f32x4 rm_ld(const AccBuf& b, int v, int coff = 0) {
    const int so = (v * (1024 / D) * LDP + coff) * 4;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.rs, b.vo, so, 0));
}
template <int D, int LDP>
__device__ __forceinline__ void rm_st(const AccBuf& b, int v, f32x4 x, int coff = 0) {
    // The slot offset rides in the voffset here, not in an SGPR soffset: a 16-byte buffer store with a register
    // soffset still reads its data VGPRs when the next instruction issues, hipcc (ROCm 7.2) schedules a VALU
    // write of those VGPRs right behind it without the wait state, and the stored row arrives corrupted
    // (seen on gfx950: two of the four dwords replaced by the following v_pk_mul's result).
    const int so = (v * (1024 / D) * LDP + coff) * 4;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), b.rs, b.vo + so, 0, 0);
}

#define FZ_FOR_ACC _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) if (rb < nrb) _Pragma("unroll") for (int reg = 0; reg < 16; ++reg)

// This wave's place in a tile of `nblk` live 32-row blocks.  Group g (waves 4 g .. 4 g + 3) owns blocks 2 g, 2 g + 1 in every
// phase; a group without a live block leaves after the tile load (all four waves: the group counters and the row-major
// I/O of a half tile count on whole groups).  nrb = this wave's live row blocks:
//   D = 128: wave row wr in {0, 1} owns blocks 2 wr, 2 wr + 1 (RB = 2);  D = 64: wave row wr in 0..3 owns block wr (RB = 1).
template <int D>
__device__ __forceinline__ int fz_live(int nblk, int wr) {
    constexpr int NCB = D / 32, NRW = 8 / NCB, RB = 4 / NRW;
    const int live = nblk - wr * RB;
    return live < 0 ? 0 : (live > RB ? RB : live);
}
// (VAR == false -- whole 128-row tiles, no table -- folds every one of these to a constant)
#define FZ_TILE_SETUP()                                                          \
    const int tile = blockIdx.x + a.tile0;                                      \
    const int row0 = VAR ? a.mt_row0[tile] : tile * FZ_R;                       \
    const int nblk = VAR ? a.mt_nblk[tile] : 4;                                 \
    const int nrows = VAR ? nblk * 32 : FZ_R;                                   \
    const int nrb = VAR ? fz_live<D>(nblk, wr) : RB;                            \
    const bool grp_live = VAR ? nblk > 2 * grp : true

#define FZ_GSYNC() grp_sync(gs)

