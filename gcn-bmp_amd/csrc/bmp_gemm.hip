// fp32-MFMA GEMM building blocks for gfx950: the row GEMM (atom-row tiles x weights) and the
// weight-gradient GEMM (reduction over atom rows).  See bmp_kernels.h for the contracts.
#include <stdlib.h>
#include <string.h>
#include "bmp_kernels.h"

// ---------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------
// Per-column constants of the epilogue, loaded ONCE per lane and column block: inside the element loop the stores to Y
// may alias them as far as the compiler knows, so it reloaded bias[col] (and waited for it) after every store.
struct RGCol { float bias; float bE[4]; };
__device__ __forceinline__ RGCol rg_col(const RGArgs& a, int col) {
    RGCol c;
    c.bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) c.bE[e] = a.wdeg ? a.bE[(size_t)e * a.ldbE + col] : 0.f;
    return c;
}

template <int EPI>
__device__ __forceinline__ void rg_epilogue(const RGArgs& a, const RGCol& cc, int row, int col, float v) {
#ifdef BMP_PROBE_NO_EPILOGUE       /* timing probe only (tools/build_variant.sh): one store per 16 accumulator registers */
    if ((row & 15) == 0 && v == 12345.678f) a.Y[(size_t)row * a.ldy + col] = v;
    return;
#endif
    v += cc.bias;
    if (EPI == BMP_EPI_GENERIC) {
        if (a.wdeg) {
            const f32x4 wd = *(const f32x4*)(a.wdeg + (size_t)row * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v += wd[e] * cc.bE[e];
        }
        const bool lo = (a.split <= 0) || (col < a.split);
        if (a.add && lo) v += a.add[(size_t)row * a.ldadd + col];
        v = bmp_act(lo ? a.act_lo : a.act_hi, v);
        if (!lo && a.o1) {
            a.o1[(size_t)row * a.ldo1 + (col - a.split)] = v;
        } else {
            float* y = a.Y + (size_t)row * a.ldy + col;
            *y = a.accumulate ? (*y + v) : v;
        }
    } else if (EPI == BMP_EPI_GRU_OUT) {
        // models/ggnn.py:260 -> chainer StatefulGRU: h' = z*h_bar + (1-z)*h ; first call: z*h_bar
        const float c = bmp_tanh(v);
        const float z = a.z[(size_t)row * a.ldz + col];
        a.c_out[(size_t)row * a.ldc + col] = c;
        float hn = z * c;
        if (!a.first) hn += (1.f - z) * a.h[(size_t)row * a.ldh + col];
        a.Y[(size_t)row * a.ldy + col] = hn;
    } else {  // BMP_EPI_GRU_DRH: v = d(r*h)
        const float r = a.r[(size_t)row * a.ldr + col];
        const float h = a.h[(size_t)row * a.ldh + col];
        a.Y[(size_t)row * a.ldy + col] = v * h * r * (1.f - r);
        float* o = a.o1 + (size_t)row * a.ldo1 + col;
        *o += v * r;
    }
}

// The same epilogues on FOUR consecutive columns of one row (col a multiple of 4): 16-byte loads and stores, a wave
// instruction moves whole 512-byte rows.  The caller has checked the operands' strides and alignment (rg_vec_ok).
struct RGCol4 { f32x4 bias; f32x4 bE[4]; };
__device__ __forceinline__ RGCol4 rg_col4(const RGArgs& a, int col) {
    RGCol4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c.bias[j] = a.bias ? a.bias[col + j] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) c.bE[e][j] = a.wdeg ? a.bE[(size_t)e * a.ldbE + col + j] : 0.f;
    }
    return c;
}
template <int EPI>
__device__ __forceinline__ void rg_epilogue4(const RGArgs& a, const RGCol4& cc, int row, int col, f32x4 v) {
    v += cc.bias;
    if (EPI == BMP_EPI_GENERIC) {
        if (a.wdeg) {
            const f32x4 wd = *(const f32x4*)(a.wdeg + (size_t)row * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v += wd[e] * cc.bE[e];
        }
        const bool lo = (a.split <= 0) || (col < a.split);
        if (a.add && lo) v += *(const f32x4*)(a.add + (size_t)row * a.ldadd + col);
        const int act = lo ? a.act_lo : a.act_hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bmp_act(act, v[j]);
        if (!lo && a.o1) {
            *(f32x4*)(a.o1 + (size_t)row * a.ldo1 + (col - a.split)) = v;
        } else {
            f32x4* y = (f32x4*)(a.Y + (size_t)row * a.ldy + col);
            *y = a.accumulate ? (*y + v) : v;
        }
    } else if (EPI == BMP_EPI_GRU_OUT) {
        f32x4 c;
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = bmp_tanh(v[j]);
        const f32x4 z = *(const f32x4*)(a.z + (size_t)row * a.ldz + col);
        *(f32x4*)(a.c_out + (size_t)row * a.ldc + col) = c;
        f32x4 hn = z * c;
        if (!a.first) hn += ((f32x4){1.f, 1.f, 1.f, 1.f} - z) * *(const f32x4*)(a.h + (size_t)row * a.ldh + col);
        *(f32x4*)(a.Y + (size_t)row * a.ldy + col) = hn;
    } else {  // BMP_EPI_GRU_DRH
        const f32x4 r = *(const f32x4*)(a.r + (size_t)row * a.ldr + col);
        const f32x4 h = *(const f32x4*)(a.h + (size_t)row * a.ldh + col);
        *(f32x4*)(a.Y + (size_t)row * a.ldy + col) = v * h * r * ((f32x4){1.f, 1.f, 1.f, 1.f} - r);
        f32x4* o = (f32x4*)(a.o1 + (size_t)row * a.ldo1 + col);
        *o += v * r;
    }
}
// strides and addresses of everything an epilogue touches row by row allow the 16-byte form
static bool rg_vec_ok(const RGArgs& a, int epi) {
    auto al = [](const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld & 3) == 0); };
    if ((a.Nout & 3) != 0 || !al(a.Y, a.ldy)) return false;
    if (epi == BMP_EPI_GENERIC)
        return al(a.add, a.ldadd) && al(a.o1, a.ldo1) && (a.split <= 0 || (a.split & 3) == 0) && (((uintptr_t)a.wdeg) & 15) == 0;
    if (epi == BMP_EPI_GRU_OUT) return al(a.z, a.ldz) && al(a.c_out, a.ldc) && al(a.h, a.ldh);
    return al(a.r, a.ldr) && al(a.h, a.ldh) && al(a.o1, a.ldo1);
}

// ---------------------------------------------------------------------------------------------
// row GEMM kernel.  256 threads = 4 waves laid out WR (rows) x WC (cols); every wave owns
// RB x CBW blocks of 32x32.  R = WR*RB*32 = 128 rows, NT = WC*CBW*32 columns per workgroup.
// ---------------------------------------------------------------------------------------------
template <int WR, int RB, int CBW, int EPI>
__device__ __forceinline__ void rowgemm_body(const RGArgs& a, int bx, int by, float* lds) {
    constexpr int WC = 4 / WR;
    constexpr int NT = WC * CBW * 32;
    constexpr int R = WR * RB * 32;                     // rows per workgroup: 128, or 64 for the small problems
    static_assert(R == BMP_R || R == BMP_R / 2, "a workgroup takes a whole or half a 128-row tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wr = w / WC, wc = w % WC;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = bx * R;
    const int n0 = by * NT;

    f32x16 acc[RB][CBW];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

    int colc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
        int col = n0 + (wc * CBW + cb) * 32 + l31;
        colc[cb] = col < a.Nout ? col : a.Nout - 1;     // clamp loads, mask stores
    }
    // a wave whose 32-column blocks all lie beyond Nout (the remainder tile of e.g. 144 columns keeps one wave of four
    // busy) stages rows and meets the barriers but issues no weight loads and no MFMAs: the matrix pipe of its SIMD is
    // left to the workgroups that share the CU
    const bool wave_active = n0 + wc * CBW * 32 < a.Nout;

    // The K axis of all sources as one sequence of <= 64-wide chunks.  The rows of chunk i+1 are requested into
    // registers before the MFMAs of chunk i (they used to be loaded, waited for and stored between two barriers
    // with the matrix pipe idle: 910 workgroups of the readout GEMM doing that in step ran at 1.3 TB/s).
    constexpr int NLD = (R * 16) / 256;                // float4 per thread and chunk
    f32x4 stage[NLD];
    int cs = 0, ck0 = 0;                               // chunk being prefetched: source, k offset
    auto load_chunk = [&](int s_, int k0_) {
        const float* __restrict__ X = a.s[s_].X;
        const float* __restrict__ X2 = a.s[s_].X2;
        const int ldx = a.s[s_].ldx, ldx2 = a.s[s_].ldx2, K = a.s[s_].K;
        const int kc4 = ((K - k0_) < 64 ? (K - k0_) : 64) >> 2;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid + it * 256;
            const int r = idx >> 4, c4 = idx & 15;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c4 < kc4) {
                v = *(const f32x4*)(X + (size_t)(row0 + r) * ldx + k0_ + 4 * c4);
                if (X2) v *= *(const f32x4*)(X2 + (size_t)(row0 + r) * ldx2 + k0_ + 4 * c4);
            }
            stage[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid + it * 256;
            *(f32x4*)(&lds[(idx >> 4) * BMP_LDS_LD + 4 * (idx & 15)]) = stage[it];
        }
    };
    load_chunk(0, 0);
    store_chunk();
    __syncthreads();
    for (int s = 0; s < a.nsrc; ++s) {
        const float* __restrict__ Wt = a.s[s].Wt;
        const int ldw = a.s[s].ldw, K = a.s[s].K;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kc = (K - k0) < 64 ? (K - k0) : 64;
            // next chunk (possibly the first of the next source)
            cs = s; ck0 = k0 + 64;
            if (ck0 >= K) { cs = s + 1; ck0 = 0; }
            const bool more = cs < a.nsrc;
            if (more) load_chunk(cs, ck0);
            const float* wp = Wt + (size_t)(k0 + 4 * hi) * ldw;
            // operands of k-step s+1 (weights from L2, row fragments from LDS) are requested before the MFMAs
            // of step s issue; scheduling barriers keep the compiler from sinking the loads behind them
            f32x4 a0[RB], a1[RB];
            float b0[CBW][4], b1[CBW][4];
            if (wave_active) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a0[rb] = *(const f32x4*)(&lds[((wr * RB + rb) * 32 + l31) * BMP_LDS_LD + 4 * hi]);
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int t = 0; t < 4; ++t) b0[cb][t] = wp[(size_t)t * ldw + colc[cb]];
            for (int kk = 0; kk < kc; kk += 8) {
                if (kk + 8 < kc) {
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) b1[cb][t] = wp[(size_t)(kk + 8 + t) * ldw + colc[cb]];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        a1[rb] = *(const f32x4*)(&lds[((wr * RB + rb) * 32 + l31) * BMP_LDS_LD + kk + 8 + 4 * hi]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = bmp_mfma(a0[rb][t], b0[cb][t], acc[rb][cb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a0[rb] = a1[rb];
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) b0[cb][t] = b1[cb][t];
            }
            }   // wave_active
            if (more) {
                __syncthreads();               // every wave is done with the chunk in LDS
                store_chunk();
                __syncthreads();
            }
        }
    }

    if (a.vec_epi) {
        // row-major through LDS, 16 bytes per lane (see rowgemm_db_body; the caller's array holds R x max(BMP_LDS_LD, NT + 4) floats)
        constexpr int LDT = NT + 4, P4 = NT / 4;            // pieces of four columns per row
        __syncthreads();                                    // every wave is done with the last chunk in LDS
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    lds[((wr * RB + rb) * 32 + bmp_acc_row(reg, lane)) * LDT + (wc * CBW + cb) * 32 + l31] = acc[rb][cb][reg];
        __syncthreads();
        const int c4 = tid % P4, col4 = n0 + 4 * c4;
        if (col4 < a.Nout) {
            const RGCol4 cc = rg_col4(a, col4);
#pragma unroll 4
            for (int lr = tid / P4; lr < R; lr += 256 / P4)
                rg_epilogue4<EPI>(a, cc, row0 + lr, col4, *(const f32x4*)(&lds[lr * LDT + 4 * c4]));
        }
        return;
    }
    if (wave_active)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const int col = n0 + (wc * CBW + cb) * 32 + l31;
            if (col < a.Nout) {
                const RGCol cc = rg_col(a, col);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = row0 + (wr * RB + rb) * 32 + bmp_acc_row(reg, lane);
                    rg_epilogue<EPI>(a, cc, row, col, acc[rb][cb][reg]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Row GEMM, both operands through LDS: 128 rows x 128 columns per workgroup (4 waves, each all 128 rows x 32 columns).
// The direct form above reads its weight fragments straight from L2 one k-step ahead; vmcnt retires in order, so that
// read queues behind the NEXT chunk's row loads (HBM latency) at every chunk start and the matrix pipe waits
// (SQ_WAIT_INST_ANY: half of the wave cycles, MFMA busy 0.38 on the readout GEMM).  Here a 64-deep chunk of the rows AND
// of the weights is requested at the top of the previous chunk, lands during its MFMAs and is written to LDS at the
// chunk boundary; inside a chunk the waves only wait on LDS.  Needs Nout, ldw multiples of 4 and 16-byte aligned weights.
// ---------------------------------------------------------------------------------------------
#define RGB_LDB 132
template <int EPI>
__device__ __forceinline__ void rowgemm_lds_body(const RGArgs& a, int bx, int by, float* lds) {
    constexpr int RB = 4, R = 128, NT = 128;
    float* la = lds;                         // [128][BMP_LDS_LD]  rows x k
    float* lb = lds + R * BMP_LDS_LD;        // [64][RGB_LDB]      k x columns
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = bx * R, n0 = by * NT;

    f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    f32x4 sa[8], sb[8];                      // the chunk in flight: 128 x 64 rows, 64 x 128 weights
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](int s_, int k0_) {
        const float* __restrict__ X = a.s[s_].X;
        const float* __restrict__ X2 = a.s[s_].X2;
        const float* __restrict__ Wt = a.s[s_].Wt;
        const int ldx = a.s[s_].ldx, ldx2 = a.s[s_].ldx2, ldw = a.s[s_].ldw, K = a.s[s_].K;
        const int kc = (K - k0_) < 64 ? (K - k0_) : 64;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            const int r = idx >> 4, c4 = idx & 15;
            f32x4 v = zero4;
            if (4 * c4 < kc) {
                v = *(const f32x4*)(X + (size_t)(row0 + r) * ldx + k0_ + 4 * c4);
                if (X2) v *= *(const f32x4*)(X2 + (size_t)(row0 + r) * ldx2 + k0_ + 4 * c4);
            }
            sa[it] = v;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            const int k = idx >> 5, c4 = idx & 31;
            const int col = n0 + 4 * c4;
            sb[it] = (k < kc && col < a.Nout) ? *(const f32x4*)(Wt + (size_t)(k0_ + k) * ldw + col) : zero4;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            *(f32x4*)(&la[(idx >> 4) * BMP_LDS_LD + 4 * (idx & 15)]) = sa[it];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            *(f32x4*)(&lb[(idx >> 5) * RGB_LDB + 4 * (idx & 31)]) = sb[it];
        }
    };
    load_chunk(0, 0);
    store_chunk();
    __syncthreads();
    for (int s = 0; s < a.nsrc; ++s) {
        const int K = a.s[s].K;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kc = (K - k0) < 64 ? (K - k0) : 64;
            int cs = s, ck0 = k0 + 64;
            if (ck0 >= K) { cs = s + 1; ck0 = 0; }
            const bool more = cs < a.nsrc;
            if (more) load_chunk(cs, ck0);
            f32x4 a0[RB], a1[RB];
            float b0[4], b1[4];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a0[rb] = *(const f32x4*)(&la[(rb * 32 + l31) * BMP_LDS_LD + 4 * hi]);
#pragma unroll
            for (int t = 0; t < 4; ++t) b0[t] = lb[(4 * hi + t) * RGB_LDB + wc * 32 + l31];
            for (int kk = 0; kk < kc; kk += 8) {
                if (kk + 8 < kc) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) b1[t] = lb[(kk + 8 + 4 * hi + t) * RGB_LDB + wc * 32 + l31];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a1[rb] = *(const f32x4*)(&la[(rb * 32 + l31) * BMP_LDS_LD + kk + 8 + 4 * hi]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb] = bmp_mfma(a0[rb][t], b0[t], acc[rb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a0[rb] = a1[rb];
#pragma unroll
                for (int t = 0; t < 4; ++t) b0[t] = b1[t];
            }
            if (more) {
                __syncthreads();
                store_chunk();
                __syncthreads();
            }
        }
    }
    const int col = n0 + wc * 32 + l31;
    if (col < a.Nout) {
        const RGCol cc = rg_col(a, col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = row0 + rb * 32 + bmp_acc_row(reg, lane);
                rg_epilogue<EPI>(a, cc, row, col, acc[rb][reg]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The same 128 x 128 tile with both operands DOUBLE-buffered in LDS in 32-deep K chunks: chunk c + 1 is requested into
// registers at the top of chunk c and written to the other buffer after its MFMAs -- one barrier per chunk instead of a
// store between two barriers with the matrix pipe idle; the weights sit in LDS as [k/4][column][4] (16-byte slot of column
// n at n ^ ((n >> 4) & 3): the register-transposing store and the fragment reads are both bank-conflict free), so a
// lane's four consecutive k of its column are ONE ds_read_b128 (the form above reads four ds_read_b32).  Same LDS
// footprint (two workgroups per CU), same epilogue.
// ---------------------------------------------------------------------------------------------
#define RGD_LDA 36
#define RGD_A_FLOATS (128 * RGD_LDA)
#define RGD_B_FLOATS (8 * 128 * 4)
#define RGD_SWZ(n) ((n) ^ (((n) >> 4) & 3))
// LISTED: the tile's 128 rows are the list entries [128 bx, 128 bx + 128) (a.ridx; *a.rcnt of them exist): A rows and Y rows go
// through `rows` (LDS, 128 ints), entries past the end re-read the last row and store nothing.
template <int EPI, bool LISTED = false>
__device__ __forceinline__ void rowgemm_db_body(const RGArgs& a, int bx, int by, float* lds, int* rows = nullptr) {
    constexpr int RB = 4, R = 128, NT = 128;
    constexpr int BUF = RGD_A_FLOATS + RGD_B_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row0 = bx * R, n0 = by * NT;
    int count = 0x7fffffff;
    int xrow[4] = {0, 0, 0, 0};              // LISTED: the four A rows this thread stages (rows (tid >> 3) + 32 it of the tile)
    if (LISTED) {
        count = a.rcnt[0];
        if (row0 >= count) return;
        if (tid < R) { const int p = row0 + tid; rows[tid] = a.ridx[p < count ? p : count - 1]; }
#pragma unroll
        for (int it = 0; it < 4; ++it) { const int p = row0 + (tid >> 3) + 32 * it; xrow[it] = a.ridx[p < count ? p : count - 1]; }
    }

    f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    f32x4 sa[4], sb[4];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bk4 = tid >> 5, bnq = tid & 31;                  // weight item of this thread: k rows 4 bk4 .. +3, columns 4 bnq .. +3
    auto load_chunk = [&](int s_, int k0_) {
        const float* __restrict__ X = a.s[s_].X;
        const float* __restrict__ X2 = a.s[s_].X2;
        const float* __restrict__ Wt = a.s[s_].Wt;
        const int ldx = a.s[s_].ldx, ldx2 = a.s[s_].ldx2, ldw = a.s[s_].ldw, K = a.s[s_].K;
        const int kc = (K - k0_) < 32 ? (K - k0_) : 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int r = idx >> 3, c4 = idx & 7;
            const size_t grow = LISTED ? (size_t)xrow[it] : (size_t)(row0 + r);
            f32x4 v = zero4;
            if (4 * c4 < kc) {
                v = *(const f32x4*)(X + grow * ldx + k0_ + 4 * c4);
                if (X2) v *= *(const f32x4*)(X2 + grow * ldx2 + k0_ + 4 * c4);
            }
            sa[it] = v;
        }
        const int col = n0 + 4 * bnq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 4 * bk4 + t;
            sb[t] = (k < kc && col < a.Nout) ? *(const f32x4*)(Wt + (size_t)(k0_ + k) * ldw + col) : zero4;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            *(f32x4*)(&lds[buf * BUF + (idx >> 3) * RGD_LDA + 4 * (idx & 7)]) = sa[it];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)                           // 4 x 4 transpose in registers: [k][n] -> [n][k]
            *(f32x4*)(&lds[buf * BUF + RGD_A_FLOATS + (bk4 * 128 + RGD_SWZ(4 * bnq + jj)) * 4]) =
                (f32x4){sb[0][jj], sb[1][jj], sb[2][jj], sb[3][jj]};
    };
    int nchunks = 0;
    for (int s = 0; s < a.nsrc; ++s) nchunks += (a.s[s].K + 31) >> 5;
    int cs = 0, ck0 = 0;
    auto advance = [&]() { ck0 += 32; if (ck0 >= a.s[cs].K) { ++cs; ck0 = 0; } };
    load_chunk(cs, ck0); advance();
    store_chunk(0);
    __syncthreads();
    const int bslot = RGD_SWZ(wc * 32 + l31);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) { load_chunk(cs, ck0); advance(); }
        const float* la = lds + buf * BUF + l31 * RGD_LDA + 4 * hi;
        const float* lb = lds + buf * BUF + RGD_A_FLOATS + ((size_t)hi * 128 + bslot) * 4;
        f32x4 a0[RB], a1[RB], b0, b1;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a0[rb] = *(const f32x4*)(la + rb * 32 * RGD_LDA);
        b0 = *(const f32x4*)lb;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 8) {
            if (kk + 8 < 32) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a1[rb] = *(const f32x4*)(la + rb * 32 * RGD_LDA + kk + 8);
                b1 = *(const f32x4*)(lb + (size_t)((kk + 8) >> 2) * 128 * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = bmp_mfma(a0[rb][t], b0[t], acc[rb]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a0[rb] = a1[rb];
            b0 = b1;
        }
        if (more) store_chunk(buf ^ 1);        // the other buffer: last read in chunk c - 1, every wave is past that barrier
        __syncthreads();
    }
    if (a.vec_epi) {
        // The tile crosses to row-major through LDS (free by now: every wave is past the last chunk's barrier): the epilogue's
        // loads and stores are 16 bytes per lane, 512-byte rows per half wave.  In accumulator layout a wave instruction moves
        // two 128-byte row pieces; the GRU epilogues' five such accesses per element made them a fifth of the step on the
        // d = 256 configuration (probe without any epilogue: 8.22 -> 6.23 ms of row GEMMs per C4 step).
        constexpr int LDT = NT + 4;
        float* T = lds;                                  // [128][LDT] <= the two staging buffers
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) T[(rb * 32 + bmp_acc_row(reg, lane)) * LDT + wc * 32 + l31] = acc[rb][reg];
        __syncthreads();
        const int c4 = tid & 31, col4 = n0 + 4 * c4;
        if (col4 < a.Nout) {
            const RGCol4 cc = rg_col4(a, col4);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int lr = (tid >> 5) + 8 * it;
                const f32x4 v = *(const f32x4*)(T + lr * LDT + 4 * c4);
                if (LISTED) { if (row0 + lr < count) rg_epilogue4<EPI>(a, cc, rows[lr], col4, v); }
                else rg_epilogue4<EPI>(a, cc, row0 + lr, col4, v);
            }
        }
        return;
    }
    const int col = n0 + wc * 32 + l31;
    if (col < a.Nout) {
        const RGCol cc = rg_col(a, col);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = rb * 32 + bmp_acc_row(reg, lane);
                if (LISTED) { if (row0 + lr < count) rg_epilogue<EPI>(a, cc, rows[lr], col, acc[rb][reg]); }
                else rg_epilogue<EPI>(a, cc, row0 + lr, col, acc[rb][reg]);
            }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void k_rowgemm_db(RGArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (RGD_A_FLOATS + RGD_B_FLOATS)];
    rowgemm_db_body<EPI>(a, blockIdx.x, blockIdx.y, lds);      // (column tiles fastest in the grid: 1.5 % slower on C4)
}
// rows through a list (generic epilogue): the per-bond-type blocks of the unfused message operator's backward
__global__ __launch_bounds__(256) void k_rowgemm_db_listed(RGArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (RGD_A_FLOATS + RGD_B_FLOATS)];
    __shared__ int rows[128];
    rowgemm_db_body<BMP_EPI_GENERIC, true>(a, blockIdx.x, blockIdx.y, lds, rows);
}

template <int EPI>
__global__ __launch_bounds__(256) void k_rowgemm_lds(RGArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[128 * BMP_LDS_LD + 64 * RGB_LDB];
    rowgemm_lds_body<EPI>(a, blockIdx.x, blockIdx.y, lds);
}

static bool rowgemm_lds_ok(const RGArgs& a) {
    static const bool off = getenv("BMP_ROWGEMM_DIRECT") != nullptr;      // tests compare the two forms
    if (off) return false;
    if ((a.Nout & 3) != 0) return false;
    for (int s = 0; s < a.nsrc; ++s)
        if ((a.s[s].ldw & 3) != 0 || ((uintptr_t)a.s[s].Wt & 15) != 0) return false;
    return true;
}

template <int WR, int RB, int CBW, int EPI>
__global__ __launch_bounds__(256) void k_rowgemm(RGArgs a) {
    constexpr int NT = (4 / WR) * CBW * 32;
    __shared__ __attribute__((aligned(16))) float lds[WR * RB * 32 * (NT + 4 > BMP_LDS_LD ? NT + 4 : BMP_LDS_LD)];      // staging | the row-major epilogue's tile
    rowgemm_body<WR, RB, CBW, EPI>(a, blockIdx.x, blockIdx.y, lds);
}

// Up to three independent row GEMMs (generic epilogue) in one launch: x-blocks [bx0[p], bx0[p+1]) belong to problem p.  The
// co-attention's projections are 228 tiles each -- alone, every one of them lasts one workgroup's latency.
// A problem whose last column tile holds at most 32 columns (the co-attention's Z = J | P | v: 128 + 16) is cut in two by the
// launcher: the whole 128-column tiles, and a THIN problem of the remainder whose 64-row workgroups are 2 x 2 waves of 32 rows
// x 32 columns (the waves of the second column block idle): half the MFMAs per wave of a full workgroup's, at the same
// register and LDS footprint (a 128-row x 32-column form took the kernel from four to three waves per SIMD and the launch
// from 57 to 73 us in line).  As a second column tile the remainder kept one wave of four busy in a workgroup that lasted as
// long as a full one: 910 of the 3190 workgroups of that launch.  Measured: 116 against 118.5 us per C2 step -- the launch is
// bound by its rounds of 64-row workgroups, not by the idle waves.
#define RGM_MAXP 6
struct RGMulti { RGArgs p[RGM_MAXP]; int bx0[RGM_MAXP + 1]; int ny[RGM_MAXP]; int thin[RGM_MAXP]; };
__global__ __launch_bounds__(256) void k_rowgemm_multi(RGMulti m) {
    __shared__ __attribute__((aligned(16))) float lds[(BMP_R / 2) * (128 + 4)];       // (64-row staging | the row-major epilogue's 64 x 128 tile)
    const int bx = blockIdx.x;
    int p = 0;
#pragma unroll
    for (int q = 1; q < RGM_MAXP; ++q) p += bx >= m.bx0[q] ? 1 : 0;
    if ((int)blockIdx.y >= m.ny[p]) return;
    if (m.thin[p]) rowgemm_body<2, 1, 1, BMP_EPI_GENERIC>(m.p[p], bx - m.bx0[p], 0, lds);              // 64 rows x (32 | idle) columns
    else rowgemm_body<1, 2, 1, BMP_EPI_GENERIC>(m.p[p], bx - m.bx0[p], blockIdx.y, lds);      // 64-row workgroups: two per tile
}

static bool rg_scalar_epilogue() {
    static const bool on = getenv("BMP_ROWGEMM_SCALAR_EPI") != nullptr;      // A/B: the accumulator-layout epilogue everywhere
    return on;
}

template <int EPI>
static int launch_rowgemm_epi(const RGArgs& a_, int n_tiles, hipStream_t st) {
    RGArgs a = a_;
    a.vec_epi = !rg_scalar_epilogue() && rg_vec_ok(a_, EPI);       // (k_rowgemm_lds keeps the accumulator-layout form)
    if (a.Nout <= 32) {
        hipLaunchKernelGGL((k_rowgemm<4, 1, 1, EPI>), dim3(n_tiles, 1), dim3(256), 0, st, a);
    } else if (a.Nout <= 64) {
        hipLaunchKernelGGL((k_rowgemm<2, 2, 1, EPI>), dim3(n_tiles, 1), dim3(256), 0, st, a);
    } else {
        const int ny = (a.Nout + 127) / 128;
        // problems that do not even give every CU one 128-row workgroup take 64-row workgroups: the launch is one
        // round either way, and its duration is one workgroup's latency
        static const int form = getenv("BMP_ROWGEMM_FORM") ? atoi(getenv("BMP_ROWGEMM_FORM")) : 0;    // 1: single-buffered form
        if (n_tiles * ny <= 256) hipLaunchKernelGGL((k_rowgemm<1, 2, 1, EPI>), dim3(2 * n_tiles, ny), dim3(256), 0, st, a);
        else if (rowgemm_lds_ok(a) && form != 1) hipLaunchKernelGGL((k_rowgemm_db<EPI>), dim3(n_tiles, ny), dim3(256), 0, st, a);
        else if (rowgemm_lds_ok(a)) hipLaunchKernelGGL((k_rowgemm_lds<EPI>), dim3(n_tiles, ny), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_rowgemm<1, 4, 1, EPI>), dim3(n_tiles, ny), dim3(256), 0, st, a);
    }
    BMP_LAUNCH_CHECK();
    return 0;
}

int bmp_launch_rowgemm(const RGArgs& a, int n_tiles, int epi, hipStream_t st) {
    BMP_REQUIRE(n_tiles > 0 && a.Nout > 0 && a.nsrc >= 1 && a.nsrc <= 3);
    for (int s = 0; s < a.nsrc; ++s) {
        BMP_REQUIRE(a.s[s].K > 0 && (a.s[s].K & 7) == 0 && (a.s[s].ldx & 3) == 0);
        BMP_REQUIRE(((uintptr_t)a.s[s].X & 15) == 0);
        if (a.s[s].X2) BMP_REQUIRE((a.s[s].ldx2 & 3) == 0 && ((uintptr_t)a.s[s].X2 & 15) == 0);
    }
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.s[s].K;
    const double rows = (double)n_tiles * BMP_R;
    BmpProfScope prof(BMP_KCLS_ROWGEMM, 2.0 * rows * ksum * a.Nout, 4.0 * rows * (ksum + a.Nout), st);
    switch (epi) {
        case BMP_EPI_GENERIC: return launch_rowgemm_epi<BMP_EPI_GENERIC>(a, n_tiles, st);
        case BMP_EPI_GRU_OUT: return launch_rowgemm_epi<BMP_EPI_GRU_OUT>(a, n_tiles, st);
        case BMP_EPI_GRU_DRH: return launch_rowgemm_epi<BMP_EPI_GRU_DRH>(a, n_tiles, st);
    }
    return -1;
}

bool bmp_rowgemm_listed_ok(const RGArgs& a) { return a.ridx && a.rcnt && rowgemm_lds_ok(a); }

int bmp_launch_rowgemm_listed(const RGArgs& a, int n_tiles_cap, hipStream_t st) {
    BMP_REQUIRE(n_tiles_cap > 0 && a.Nout > 0 && a.nsrc >= 1 && a.nsrc <= 3 && a.ridx && a.rcnt && rowgemm_lds_ok(a));
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        BMP_REQUIRE(a.s[s].K > 0 && (a.s[s].K & 7) == 0 && (a.s[s].ldx & 3) == 0 && ((uintptr_t)a.s[s].X & 15) == 0 && !a.s[s].X2);
        ksum += a.s[s].K;
    }
    // (flop accounting: the listed share is not known on the host; the roofline leg's algorithmic figure counts every row)
    const double rows = (double)n_tiles_cap * BMP_R;
    BmpProfScope prof(BMP_KCLS_ROWGEMM, 2.0 * rows * ksum * a.Nout, 4.0 * rows * (ksum + a.Nout), st);
    RGArgs b = a;
    b.vec_epi = !rg_scalar_epilogue() && rg_vec_ok(a, BMP_EPI_GENERIC);
    hipLaunchKernelGGL(k_rowgemm_db_listed, dim3(n_tiles_cap, (a.Nout + 127) / 128), dim3(256), 0, st, b);
    BMP_LAUNCH_CHECK();
    return 0;
}

int bmp_launch_rowgemm_multi(const RGArgs* a, const int* n_tiles, int n, hipStream_t st) {
    BMP_REQUIRE(n >= 1 && n <= 3);
    static const bool no_thin = getenv("BMP_ROWGEMM_NO_THIN") != nullptr;       // A/B: the remainder as a second column tile
    RGMulti m; memset(&m, 0, sizeof(m));
    int np = 0, nymax = 1;
    int nt_of[RGM_MAXP];
    double flops = 0, bytes = 0;
    for (int p = 0; p < n; ++p) {
        BMP_REQUIRE(n_tiles[p] > 0 && a[p].Nout > 0 && a[p].nsrc >= 1 && a[p].nsrc <= 3);
        double ksum = 0;
        for (int s = 0; s < a[p].nsrc; ++s) {
            BMP_REQUIRE(a[p].s[s].K > 0 && (a[p].s[s].K & 7) == 0 && (a[p].s[s].ldx & 3) == 0 && ((uintptr_t)a[p].s[s].X & 15) == 0);
            if (a[p].s[s].X2) BMP_REQUIRE((a[p].s[s].ldx2 & 3) == 0 && ((uintptr_t)a[p].s[s].X2 & 15) == 0);
            ksum += a[p].s[s].K;
        }
        const double rows = (double)n_tiles[p] * BMP_R;
        flops += 2.0 * rows * ksum * a[p].Nout; bytes += 4.0 * rows * (ksum + a[p].Nout);
        const int rem = a[p].Nout & 127;
        const bool plain = a[p].split <= 0 && !a[p].add && !a[p].wdeg && !a[p].o1 && !a[p].ridx;
        m.p[np] = a[p]; nt_of[np] = n_tiles[p];
        if (!no_thin && plain && a[p].Nout > 128 && rem >= 1 && rem <= 32) m.p[np].Nout = a[p].Nout - rem;
        m.ny[np] = (m.p[np].Nout + 127) / 128;
        if (m.ny[np] > nymax) nymax = m.ny[np];
        ++np;
    }
    for (int p = 0; p < n; ++p) {          // the thin remainders behind the whole tiles: the long workgroups start first
        if (m.p[p].Nout == a[p].Nout) continue;
        const int c0 = m.p[p].Nout;
        RGArgs t = a[p];
        t.Nout = a[p].Nout - c0; t.Y = a[p].Y + c0;
        if (t.bias) t.bias = a[p].bias + c0;
        for (int s = 0; s < t.nsrc; ++s) t.s[s].Wt = a[p].s[s].Wt + c0;
        m.p[np] = t; m.thin[np] = 1; m.ny[np] = 1; nt_of[np] = n_tiles[p];
        ++np;
    }
    int blocks = 0;
    for (int q = 0; q < np; ++q) m.p[q].vec_epi = !rg_scalar_epilogue() && rg_vec_ok(m.p[q], BMP_EPI_GENERIC);
    for (int q = 0; q < np; ++q) { m.bx0[q] = blocks; blocks += 2 * nt_of[q]; }
    for (int q = np; q <= RGM_MAXP; ++q) m.bx0[q] = blocks;
    for (int q = np; q < RGM_MAXP; ++q) m.bx0[q] = 0x7fffffff;      // (the kernel's search never lands behind the last problem)
    m.bx0[RGM_MAXP] = blocks;
    BmpProfScope prof(BMP_KCLS_ROWGEMM, flops, bytes, st, BMP_KID_ROWGEMM_MULTI);
    hipLaunchKernelGGL(k_rowgemm_multi, dim3(blocks, nymax), dim3(256), 0, st, m);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM: slab[s][i][j] = sum_{rows of split s} X[row, i] * dY[row, j]
// 4 waves as 2x2, each wave MB x NB blocks of 32x32; operands straight from global (rows are
// 128-B coalesced per half-wave, re-reads across the workgroup's waves hit L1/L2).
// ---------------------------------------------------------------------------------------------
struct WGKArgs {
    const float* X; const float* X2; int ldx, ldx2;
    const float* dY; int ldy;
    int K, Nn, N, rows_per_split;
    float* slab;
    const int* onehot;      // if set, X is not read: X[row, k] = (onehot[row] == k)  (embedding gradient as a GEMM)
    int skip_at = 0x7fffffff, skip_n = 0;     // logical column j reads dY column j + (j >= skip_at ? skip_n : 0)
    const float* wrow = nullptr; int w_col0 = 0;      // weighted column sums (slab rows K + 1 .. K + 4) for columns >= w_col0
    // row list (LDS-DMA body only): rows ridx[0 .. *rcnt), cut into nsplit parts in the kernel; zrow: >= 128 zero floats, the
    // source of the pieces past the end of the list
    const int* ridx = nullptr; const int* rcnt = nullptr; const float* zrow = nullptr; int nsplit = 0;
};

// 128 zero floats in device memory (zero-initialised when the code object is loaded, per device): what a listed problem of
// the LDS-DMA body reads for the pieces past the end of its list.  (It used to be the head of the launch's workspace, cleared
// by a memset of its own in front of every such launch: four 5 us launches per GGNN step.)
__device__ float bmp_wg_zero_row[128];

template <int MB, int NB>
__global__ __launch_bounds__(256) void k_wgrad(WGKArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i0 = blockIdx.x * (64 * MB) + wm * (32 * MB);
    const int j0 = blockIdx.y * (64 * NB) + wn * (32 * NB);
    const int s = blockIdx.z;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    int ic[MB], jc[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { int i = i0 + m * 32 + l31; ic[m] = i < a.K ? i : a.K - 1; }
#pragma unroll
    for (int n = 0; n < NB; ++n) { int j = j0 + n * 32 + l31; jc[n] = j < a.Nn ? j : a.Nn - 1; }

    const float* __restrict__ X = a.X;
    const float* __restrict__ X2 = a.X2;
    const float* __restrict__ dY = a.dY;
    for (int r = r_begin + 4 * hi; r < r_end; r += 8) {
        float av[MB][4], bv[NB][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float x = X[(size_t)(r + t) * a.ldx + ic[m]];
                if (X2) x *= X2[(size_t)(r + t) * a.ldx2 + ic[m]];
                av[m][t] = x;
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n][t] = dY[(size_t)(r + t) * a.ldy + jc[n]];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = bmp_mfma(av[m][t], bv[n][t], acc[m][n]);
    }

    float* slab = a.slab + (size_t)s * a.K * a.Nn;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int j = j0 + n * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = i0 + m * 32 + bmp_acc_row(reg, lane);
                if (i < a.K && j < a.Nn) slab[(size_t)i * a.Nn + j] = acc[m][n][reg];
            }
        }
}

// out[i, j] (=|+=) sum_s slab[s][i][j]
// 64 elements x 4 slab lanes per workgroup; fixed summation order (bitwise reproducible).
// Slabs are [S][K][Nn]; row `cs_row` (if >= 0) holds column sums and goes to cs_out instead of out.
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slab, int S, int K, int Nn, float* out,
                                                      int ldo, int accumulate, int cs_row, float* cs_out, int cs_accumulate) {
    __shared__ float red[4][64];
    const size_t total = (size_t)K * Nn;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        const size_t idx = base + c;
        float v = 0.f;
        if (idx < total) {
#pragma unroll 8
            for (int s = g; s < S; s += 4) v += slab[(size_t)s * total + idx];
        }
        red[g][c] = v;
        __syncthreads();
        if (g == 0 && idx < total) {
            v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            const int i = (int)(idx / Nn), j = (int)(idx % Nn);
            float* o = (i == cs_row) ? (cs_out + j) : (out + (size_t)i * ldo + j);
            *o = (i == cs_row ? cs_accumulate : accumulate) ? (*o + v) : v;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged weight-gradient GEMM for the big shapes: one 128x128 output tile per workgroup, rows
// streamed in 32-row stages through a double-buffered LDS image (global loads of stage s+1 are in
// flight while stage s feeds the MFMAs), operands shared by the four waves.  Optionally also
// emits the column sums of dY (bias gradients) as slab row K: dY is being read anyway.
// ---------------------------------------------------------------------------------------------
#define WG_LD 132
typedef float WgStage[32][WG_LD];
template <bool HAS_X2>
__device__ __forceinline__ void wgrad_lds_body(const WGKArgs& a, int want_cs, int bx, int by, int bz, WgStage* XS, WgStage* YS) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i_tile = bx * 128, j_tile = by * 128;
    const int s = bz;
    const int r_begin = s * a.rows_per_split;
    const int r_end = (r_begin + a.rows_per_split) < a.N ? (r_begin + a.rows_per_split) : a.N;
    const int nst = (r_end - r_begin) >> 5;
    const int c4 = tid & 31, rr = tid >> 5;
    const int colx = i_tile + 4 * c4, coly = j_tile + 4 * c4;
    const int colyp = coly + (coly >= a.skip_at ? a.skip_n : 0);        // physical dY column
    const bool okx = colx < a.K, oky = coly < a.Nn;
    const bool do_cs = want_cs && bx == 0;
    const bool do_w = do_cs && a.wrow != nullptr && j_tile >= a.w_col0;
    const int Krows = a.K + (want_cs ? 1 : 0) + (a.wrow ? 4 : 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 csw[4] = {csum, csum, csum, csum};
    f32x4 xr[4], yr[4], wr4[4];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

#define WG_LOAD(st)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
        const size_t row = (size_t)(r_begin + (st) * 32 + rr + 8 * i);                               \
        if (a.onehot) {                                                                              \
            const int id = a.onehot[row] - colx;                                                     \
            xr[i] = (f32x4){id == 0 ? 1.f : 0.f, id == 1 ? 1.f : 0.f, id == 2 ? 1.f : 0.f, id == 3 ? 1.f : 0.f}; \
        } else {                                                                                     \
            xr[i] = okx ? *(const f32x4*)(a.X + row * a.ldx + colx) : zero4;                          \
            if (HAS_X2 && okx) xr[i] *= *(const f32x4*)(a.X2 + row * a.ldx2 + colx);                 \
        }                                                                                            \
        yr[i] = oky ? *(const f32x4*)(a.dY + row * a.ldy + colyp) : zero4;                            \
        if (do_w) wr4[i] = *(const f32x4*)(a.wrow + row * 4);                                        \
    }
#define WG_STORE(buf)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
        *(f32x4*)(&XS[buf][rr + 8 * i][4 * c4]) = xr[i];                                             \
        *(f32x4*)(&YS[buf][rr + 8 * i][4 * c4]) = yr[i];                                             \
        csum += yr[i];                                                                               \
        if (do_w) { _Pragma("unroll") for (int e = 0; e < 4; ++e) csw[e] += wr4[i][e] * yr[i]; }     \
    }

    if (nst > 0) {
        WG_LOAD(0)
        WG_STORE(0)
    }
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) { WG_LOAD(st + 1) }
        // fragments of k-step s+1 are read from LDS (pinned by scheduling barriers) before the 16 MFMAs of
        // k-step s issue: otherwise every MFMA waits on its own ds_read
        float av[2][2][4], bv[2][2][4];
#define WG_FRAG(slot, k0)                                                                              \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
            av[slot][m][t] = XS[buf][(k0) + 4 * hi + t][wm * 64 + m * 32 + l31];                       \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                  \
            bv[slot][n][t] = YS[buf][(k0) + 4 * hi + t][wn * 64 + n * 32 + l31];                       \
    }
        WG_FRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) { WG_FRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WG_FRAG
        if (st + 1 < nst) { WG_STORE(buf ^ 1) }
        __syncthreads();
    }
#undef WG_LOAD
#undef WG_STORE

    float* slab = a.slab + (size_t)s * Krows * a.Nn;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = i_tile + wm * 64 + m * 32 + bmp_acc_row(reg, lane);
                if (i < a.K && j < a.Nn) slab[(size_t)i * a.Nn + j] = acc[m][n][reg];
            }
        }
    if (do_cs) {        // column sums of this split's dY rows: reduce the 8 row groups through LDS
        f32x4* red = (f32x4*)&XS[0][0][0];
        red[rr * 32 + c4] = csum;
        __syncthreads();
        if (rr == 0 && oky) {
            f32x4 t = red[c4];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g * 32 + c4];
            *(f32x4*)(slab + (size_t)a.K * a.Nn + coly) = t;
        }
        if (do_w) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __syncthreads();
                red[rr * 32 + c4] = csw[e];
                __syncthreads();
                if (rr == 0 && oky) {
                    f32x4 t = red[c4];
#pragma unroll
                    for (int g = 1; g < 8; ++g) t += red[g * 32 + c4];
                    *(f32x4*)(slab + (size_t)(a.K + 1 + e) * a.Nn + coly) = t;
                }
            }
        }
    }
}

template <bool HAS_X2>
__global__ __launch_bounds__(256) void k_wgrad_lds(WGKArgs a, int want_cs) {
    __shared__ __attribute__((aligned(16))) float XS[2][32][WG_LD];
    __shared__ __attribute__((aligned(16))) float YS[2][32][WG_LD];
    wgrad_lds_body<HAS_X2>(a, want_cs, blockIdx.x, blockIdx.y, blockIdx.z, XS, YS);
}

// Up to three independent problems in one launch (column tiles [ty0[p], ty0[p+1]) belong to problem p; each has its
// own row count, split size and slab).  For problems that are a launch of 1-2 tiles x ~128 splits each: back to back
// every one of them lasts one workgroup's latency, side by side they share it.  ONE staging area: an LDS array per
// instantiated body would halve the workgroups per CU.
// The same 128 x 128 tile with its stages filled by LDS-DMA (global_load_lds_dwordx4: global memory -> LDS without passing
// through registers): stages of WD_RS = 16 rows, WD_NS = 3 of them in LDS, two in flight behind the one that feeds the MFMAs;
// counted vmcnt + one raw s_barrier per stage.  Against the register-staged body: no staging registers, no ds_write phase
// behind the MFMAs of a stage (the wait for the loads sat right in front of it), 48 KB of LDS instead of 67.5 (three
// workgroups per CU).  Same products in the same order: bit for bit the register-staged result (stand-alone harness
// tools/wgrad_bench.hip, the 11-tile shape of a fused step: 218 -> 175 us).  Stage layout: X [16 x 128] | dY [16 x 128] |
// X2 [16 x 128] (only in launches that have an X2 problem); rows are 128 floats without padding (a DMA writes a wave's 64
// pieces back to back), which the fragment reads -- 32 consecutive floats of a row per half-wave -- take without conflicts.
#define WD_RS 16
#define WD_NS 3
template <bool HAS_X2, bool IDX>
__device__ __forceinline__ void wgrad_dma_body(const WGKArgs& a, int want_cs, int bx, int by, int bz, float* sm, int ssz) {
    constexpr int XSZ = WD_RS * 128;
    constexpr int NI = WD_RS / 8;                    // DMA instructions per wave, operand and stage (a wave moves two rows each)
    constexpr int NLD = NI * (HAS_X2 ? 3 : 2);       // loads per lane and stage
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int i_tile = bx * 128, j_tile = by * 128;
    const int s = bz;
    // IDX: the problem walks the row list ridx[0 .. count) (count is on the device): cut into nsplit parts of whole stages;
    // the pieces of the last stage past the end of the list come from the zero row
    const int count = IDX ? a.rcnt[0] : a.N;
    const int rps = IDX ? ((((count + a.nsplit - 1) / a.nsplit) + WD_RS - 1) / WD_RS) * WD_RS : a.rows_per_split;
    const int r_begin = s * rps;
    const int r_end = (r_begin + rps) < count ? (r_begin + rps) : count;
    const int nst = r_end > r_begin ? (r_end - r_begin + WD_RS - 1) / WD_RS : 0;
    const int c4 = tid & 31, rr = tid >> 5;          // column sums: this thread's four columns of the rows rr, rr + 8 of a stage
    const int coly = j_tile + 4 * c4;
    const bool oky = coly < a.Nn;
    const bool do_cs = want_cs != 0 && bx == 0;
    const bool do_w = do_cs && a.wrow != nullptr && j_tile >= a.w_col0;
    const int Krows = a.K + (want_cs ? 1 : 0) + (a.wrow ? 4 : 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 csw[4] = {csum, csum, csum, csum};

    // per-lane sources: columns beyond the matrix read a valid address of the same row instead (their products are never stored)
    int xcol = i_tile + 4 * l31;
    if (xcol > a.K - 4) xcol = a.K - 4;
    int ycol = j_tile + 4 * l31;
    if (ycol > a.Nn - 4) ycol = a.Nn - 4;
    ycol += (ycol >= a.skip_at ? a.skip_n : 0);        // physical dY column
    const float* xsrc = a.X + (size_t)(r_begin + 2 * w + hi) * a.ldx + xcol;
    const float* x2src = HAS_X2 ? a.X2 + (size_t)(r_begin + 2 * w + hi) * a.ldx2 + xcol : nullptr;
    const float* ysrc = a.dY + (size_t)(r_begin + 2 * w + hi) * a.ldy + ycol;
    typedef __attribute__((address_space(3))) float lds_f;
    typedef const __attribute__((address_space(1))) float glb_f;
    // the rows' four weights (weighted column sums) ride along as one more piece of the stage, moved by wave 0 and issued FIRST:
    // the counted wait below leaves the youngest NLD loads of a wave outstanding, i.e. the next stage's X / dY / X2 pieces
    const int WOFF = (HAS_X2 ? 3 : 2) * XSZ;
    const float* wsrc = do_w ? a.wrow + (size_t)(r_begin + (lane & 15)) * 4 : nullptr;
    // IDX: the four row numbers a wave needs per stage (rows 2 w, 2 w + 1 of the stage's two 8-row halves) are wave-uniform.
    // They are read through the SCALAR cache (the list was written by an earlier kernel: a constant-address-space view), one
    // stage ahead of their use: scalar loads count in lgkmcnt, so neither the compiler's wait for them nor the counted vmcnt
    // waits of the DMA pipeline see each other (as per-lane vector loads the compiler drained the DMA queue -- vmcnt(0) --
    // before every use: one stage in flight, 212 us instead of 185).  Operand bases in registers: the waits' memory clobber
    // would re-read the argument block every trip.
    typedef int wd_v2i __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) wd_v2i wd_cv2i;
    typedef const __attribute__((address_space(4))) int wd_cint;
    wd_cint* const ridx_ = IDX ? (wd_cint*)(uintptr_t)a.ridx : nullptr;
    const float* const xb_ = a.X; const float* const yb_ = a.dY; const float* const zr_ = IDX ? bmp_wg_zero_row + 4 * l31 : nullptr;
    const int ldx_ = a.ldx, ldy_ = a.ldy;
    const int cap_ = a.N;
    const int wu_ = __builtin_amdgcn_readfirstlane(w);
    wd_v2i rva = {0, 0}, rvb = {0, 0};
#define WD_FETCH(st)                                                                                                    \
    if (IDX && (st) < nst) {                                                                                            \
        int pw = r_begin + (st) * WD_RS + 2 * wu_;              /* even: 8-byte aligned pairs */                        \
        pw = __builtin_amdgcn_readfirstlane(pw < cap_ - 10 ? pw : cap_ - 10);                                           \
        rva = *(wd_cv2i*)(ridx_ + pw);                                                                                  \
        rvb = *(wd_cv2i*)(ridx_ + pw + 8);                                                                              \
    }
#define WD_ISSUE_IDX(st)                                                                                                \
    {                                                                                                                   \
        float* base = sm + ((st) % WD_NS) * ssz;                                                                        \
        const int p0 = r_begin + (st) * WD_RS + 2 * w + hi;                                                             \
        const int row0 = p0 < count ? (hi ? rva.y : rva.x) : -1, row1 = p0 + 8 < count ? (hi ? rvb.y : rvb.x) : -1;     \
        WD_FETCH((st) + 1)                                                                                              \
        const float* x0 = row0 >= 0 ? xb_ + (size_t)row0 * ldx_ + xcol : zr_;                                           \
        const float* y0 = row0 >= 0 ? yb_ + (size_t)row0 * ldy_ + ycol : zr_;                                           \
        const float* x1 = row1 >= 0 ? xb_ + (size_t)row1 * ldx_ + xcol : zr_;                                           \
        const float* y1 = row1 >= 0 ? yb_ + (size_t)row1 * ldy_ + ycol : zr_;                                           \
        __builtin_amdgcn_global_load_lds((glb_f*)x0, (lds_f*)(base + (2 * w) * 128), 16, 0, 0);                         \
        __builtin_amdgcn_global_load_lds((glb_f*)y0, (lds_f*)(base + XSZ + (2 * w) * 128), 16, 0, 0);                   \
        __builtin_amdgcn_global_load_lds((glb_f*)x1, (lds_f*)(base + (2 * w + 8) * 128), 16, 0, 0);                     \
        __builtin_amdgcn_global_load_lds((glb_f*)y1, (lds_f*)(base + XSZ + (2 * w + 8) * 128), 16, 0, 0);               \
    }
#define WD_ISSUE(st)                                                                                                    \
    if (IDX) { WD_ISSUE_IDX(st) } else                                                                                  \
    {                                                                                                                   \
        float* base = sm + ((st) % WD_NS) * ssz;                                                                        \
        const size_t ro = (size_t)(st) * WD_RS;                                                                         \
        if (do_w && w == 0 && lane < WD_RS)                                                                             \
            __builtin_amdgcn_global_load_lds((glb_f*)(wsrc + ro * 4), (lds_f*)(base + WOFF), 16, 0, 0);                 \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                                \
            __builtin_amdgcn_global_load_lds((glb_f*)(xsrc + (ro + 8 * i) * a.ldx), (lds_f*)(base + (2 * w + 8 * i) * 128), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((glb_f*)(ysrc + (ro + 8 * i) * a.ldy), (lds_f*)(base + XSZ + (2 * w + 8 * i) * 128), 16, 0, 0); \
            if (HAS_X2)                                                                                                 \
                __builtin_amdgcn_global_load_lds((glb_f*)(x2src + (ro + 8 * i) * a.ldx2), (lds_f*)(base + 2 * XSZ + (2 * w + 8 * i) * 128), 16, 0, 0); \
        }                                                                                                               \
    }
    static_assert(!IDX || NI == 2, "the listed form issues the two 8-row halves of a 16-row stage by hand");
    WD_FETCH(0)
#pragma unroll
    for (int p = 0; p < WD_NS - 1; ++p)
        if (p < nst) WD_ISSUE(p)
    for (int st = 0; st < nst; ++st) {
        // stage st has landed once at most the loads of the stages issued behind it are outstanding (loads complete in order)
        if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // every wave's pieces of the stage are in LDS; stage st - 1 has been read
        if (st + WD_NS - 1 < nst) WD_ISSUE(st + WD_NS - 1)
        const float* XS = sm + (st % WD_NS) * ssz;
        const float* YS = XS + XSZ;
        const float* X2S = XS + 2 * XSZ;
        const float* WS = XS + WOFF;
        if (do_cs) {                                 // column sums (and the weighted ones), in the register-staged body's order
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x4 yr = *(const f32x4*)(YS + (rr + 8 * i) * 128 + 4 * c4);
                csum += yr;
                if (do_w) {
                    const f32x4 wr4 = *(const f32x4*)(WS + (rr + 8 * i) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) csw[e] += wr4[e] * yr;
                }
            }
        }
        float av[2][2][4], bv[2][2][4];
#define WD_FRAG(slot, k0)                                                                                               \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                                 \
            av[slot][m][t] = XS[((k0) + 4 * hi + t) * 128 + wm * 64 + m * 32 + l31];                                    \
            if (HAS_X2) av[slot][m][t] *= X2S[((k0) + 4 * hi + t) * 128 + wm * 64 + m * 32 + l31];                      \
        }                                                                                                               \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) bv[slot][n][t] = YS[((k0) + 4 * hi + t) * 128 + wn * 64 + n * 32 + l31]; \
    }
        WD_FRAG(0, 0)
#pragma unroll
        for (int ks = 0; ks < WD_RS / 8; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < WD_RS / 8) { WD_FRAG(cur ^ 1, (ks + 1) * 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = bmp_mfma(av[cur][m][t], bv[cur][n][t], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WD_FRAG
    }
#undef WD_ISSUE
#undef WD_ISSUE_IDX
#undef WD_FETCH

    float* slab = a.slab + (size_t)s * Krows * a.Nn;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int j = j_tile + wn * 64 + n * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = i_tile + wm * 64 + m * 32 + bmp_acc_row(reg, lane);
                if (i < a.K && j < a.Nn) slab[(size_t)i * a.Nn + j] = acc[m][n][reg];
            }
        }
    if (do_cs) {        // column sums of this split's dY rows: reduce the 8 row groups through LDS
        __syncthreads();                             // the last stage has been read by every wave
        f32x4* red = (f32x4*)sm;
        red[rr * 32 + c4] = csum;
        __syncthreads();
        if (rr == 0 && oky) {
            f32x4 t = red[c4];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g * 32 + c4];
            *(f32x4*)(slab + (size_t)a.K * a.Nn + coly) = t;
        }
        if (do_w) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __syncthreads();
                red[rr * 32 + c4] = csw[e];
                __syncthreads();
                if (rr == 0 && oky) {
                    f32x4 t = red[c4];
#pragma unroll
                    for (int g = 1; g < 8; ++g) t += red[g * 32 + c4];
                    *(f32x4*)(slab + (size_t)(a.K + 1 + e) * a.Nn + coly) = t;
                }
            }
        }
    }
}

#define BMP_WG_MAXT 32
struct WGKMulti {
    WGKArgs p[BMP_WG_MAXP]; int want_cs[BMP_WG_MAXP]; int S[BMP_WG_MAXP]; int ty0[BMP_WG_MAXP + 1]; int n, grouped, smax, ssz;
    // grouped == 2: a flat grid of exactly the launch's work items (problems with different part counts: row lists).  Item L
    // is the (L - before(z))-th of the column tiles that have a part z, tiles in descending order of their part counts:
    // ford[k] = the k-th tile, fs[k] = its part count; before(z) = sum_k min(fs[k], z).
    int ft; unsigned char ford[BMP_WG_MAXT]; short fs[BMP_WG_MAXT];
};
// flat grid -> (column tile, part); false: no such item
__device__ __forceinline__ bool wgk_flat(const WGKMulti& m, int L, int& by, int& bz) {
    int lo = 0, hi = m.smax;                 // largest z with before(z) <= L
    while (hi - lo > 1) {
        const int z = (lo + hi) >> 1;
        int b = 0;
        for (int k = 0; k < m.ft; ++k) b += m.fs[k] < z ? m.fs[k] : z;
        if (b <= L) lo = z; else hi = z;
    }
    int b = 0;
    for (int k = 0; k < m.ft; ++k) b += m.fs[k] < lo ? m.fs[k] : lo;
    const int k = L - b;
    if (k >= m.ft || m.fs[k] <= lo) return false;
    by = m.ford[k]; bz = lo;
    return true;
}
// the problem that owns column tile `by` (ty0[p] <= by < ty0[p + 1]; ty0[n ..] = the launch's tile count)
__device__ __forceinline__ int wgk_problem(const WGKMulti& m, int by) {
    int p = 0;
#pragma unroll
    for (int q = 1; q < BMP_WG_MAXP; ++q) p += (q < m.n && by >= m.ty0[q]) ? 1 : 0;
    return p;
}
// STEP = 1: the launch of bmp_launch_wgrad_fused (same code; a separate symbol so that profiles tell the fused step
// weight gradients from the co-attention's small three-problem launch)
template <int STEP>
__global__ __launch_bounds__(256) void k_wgrad_lds_multi(WGKMulti m) {
    __shared__ __attribute__((aligned(16))) float XS[2][32][WG_LD];
    __shared__ __attribute__((aligned(16))) float YS[2][32][WG_LD];
    int by, bz;
    if (m.grouped == 2) { if (!wgk_flat(m, blockIdx.x, by, bz)) return; }
    else if (m.grouped) {
        // XCD-grouped 1-D launch (wgrad_grouped_grid): workgroup L runs on XCD L % 8, and the launch fits the chip's slots, so
        // the workgroups of an XCD are resident together.  All column tiles of a row split then sit on ONE XCD and walk the same
        // rows of X and dY at about the same time: one fetch into that XCD's L2 serves them all (in the (tile, split) grid order
        // the tiles of a split landed on eight different XCDs and every one of them fetched its operands from HBM itself: 658 MB
        // per launch for 291 MB of operands).  G = smax / 8 splits live whole on every XCD; the tiles of the remaining
        // smax % 8 splits fill the XCDs' last slots one by one.
        const int T = m.ty0[BMP_WG_MAXP], L = blockIdx.x, xcd = L & 7, slot = L >> 3, G = m.smax >> 3;
        if (slot < G * T) { bz = xcd * G + slot / T; by = slot % T; }
        else {
            const int r = (slot - G * T) * 8 + xcd;
            if (r >= (m.smax - 8 * G) * T) return;
            bz = 8 * G + r / T; by = r % T;
        }
    } else { by = blockIdx.y; bz = blockIdx.z; }
    const int p = wgk_problem(m, by);
    if (bz >= m.S[p]) return;
    if (m.p[p].X2) wgrad_lds_body<true>(m.p[p], m.want_cs[p], 0, by - m.ty0[p], bz, XS, YS);
    else wgrad_lds_body<false>(m.p[p], m.want_cs[p], 0, by - m.ty0[p], bz, XS, YS);
}

// The same launch on the LDS-DMA body (wgrad_dma_body); m.ssz = floats of one stage in the dynamic LDS block.
template <int STEP>
__global__ __launch_bounds__(256) void k_wgrad_dma_multi(WGKMulti m) {
    extern __shared__ __attribute__((aligned(16))) float wd_sm[];
    int by, bz;
    if (m.grouped == 2) { if (!wgk_flat(m, blockIdx.x, by, bz)) return; }
    else if (m.grouped) {     // see k_wgrad_lds_multi
        const int T = m.ty0[BMP_WG_MAXP], L = blockIdx.x, xcd = L & 7, slot = L >> 3, G = m.smax >> 3;
        if (slot < G * T) { bz = xcd * G + slot / T; by = slot % T; }
        else {
            const int r = (slot - G * T) * 8 + xcd;
            if (r >= (m.smax - 8 * G) * T) return;
            bz = 8 * G + r / T; by = r % T;
        }
    } else { by = blockIdx.y; bz = blockIdx.z; }
    const int p = wgk_problem(m, by);
    if (bz >= m.S[p]) return;
    if (m.p[p].ridx) wgrad_dma_body<false, true>(m.p[p], m.want_cs[p], 0, by - m.ty0[p], bz, wd_sm, m.ssz);
    else if (m.p[p].X2) wgrad_dma_body<true, false>(m.p[p], m.want_cs[p], 0, by - m.ty0[p], bz, wd_sm, m.ssz);
    else wgrad_dma_body<false, false>(m.p[p], m.want_cs[p], 0, by - m.ty0[p], bz, wd_sm, m.ssz);
}

// Launches k_wgrad_dma_multi<STEP> when every problem of the launch can take the DMA body (whole 16-byte pieces in range:
// K and Nn at least 4 and multiples of 4 -- wgrad_use_lds --, no one-hot operand), else k_wgrad_lds_multi<STEP>.
// BMP_WGRAD_DMA=0: always the register-staged kernel.
static bool wgrad_dma_enabled() {
    static const int on = [] { const char* e = getenv("BMP_WGRAD_DMA"); return e ? atoi(e) : 1; }();
    return on != 0;
}
template <int STEP>
static int wgrad_multi_launch(WGKMulti& m, int n, const dim3& grid, hipStream_t st) {
    bool ok = wgrad_dma_enabled(), x2 = false;
    for (int p = 0; p < n; ++p) {
        if (m.S[p] == 0) continue;
        ok = ok && !m.p[p].onehot && m.p[p].K >= 4 && m.p[p].Nn >= 4 && (m.p[p].rows_per_split % WD_RS) == 0 && (m.p[p].N % WD_RS) == 0;
        x2 = x2 || m.p[p].X2 != nullptr;
    }
    if (!ok) {
        for (int p = 0; p < n; ++p) BMP_REQUIRE(m.p[p].ridx == nullptr);       // row lists exist in the LDS-DMA body only
        hipLaunchKernelGGL((k_wgrad_lds_multi<STEP>), grid, dim3(256), 0, st, m);
        return 0;
    }
    m.ssz = WD_RS * 128 * (x2 ? 3 : 2) + WD_RS * 4;          // + the rows' four weights
    const size_t lds = (size_t)WD_NS * m.ssz * sizeof(float);
    if (int rc_attr = bmp_lds_attr((const void*)k_wgrad_dma_multi<STEP>, (size_t)(WD_NS * (WD_RS * 128 * 3 + WD_RS * 4) * (int)sizeof(float)))) return rc_attr;
    hipLaunchKernelGGL((k_wgrad_dma_multi<STEP>), grid, dim3(256), lds, st, m);
    return 0;
}

// Grid of a three-problem launch: XCD-grouped (see the kernel) when the work fits the chip's 512 slots in one round and there
// is more than one column tile to share operands; else the plain (tile, split) grid.  BMP_WGRAD_XCD=0: always the plain grid.
static dim3 wgrad_grouped_grid(WGKMulti& m, int smax) {
    static const int on = [] { const char* e = getenv("BMP_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    const int T = m.ty0[BMP_WG_MAXP];
    m.smax = smax; m.grouped = 0;
    {   // problems with different part counts (row lists): exactly the work items, no holes in the grid -- a (tile, part) grid
        // whose missing parts exit at once leaves some CUs with three resident workgroups and others with one, and the launch
        // lasts as long as the fullest CU (measured: 213 us against 185 for MORE work)
        bool same = true;
        for (int p = 0; p < m.n; ++p) if (m.S[p] != 0 && m.S[p] != smax) same = false;
        if (!same && T <= BMP_WG_MAXT) {
            int tp[BMP_WG_MAXT], nt = 0, total = 0;
            for (int p = 0; p < m.n; ++p)
                for (int t = m.ty0[p]; t < m.ty0[p + 1]; ++t) if (m.S[p] > 0) { tp[nt++] = t; total += m.S[p]; }
            auto sof = [&](int t) { int p = 0; while (p + 1 < m.n && t >= m.ty0[p + 1]) ++p; return m.S[p]; };
            for (int i = 1; i < nt; ++i)                 // insertion sort, descending part count, stable
                for (int j = i; j > 0 && sof(tp[j]) > sof(tp[j - 1]); --j) { const int x = tp[j]; tp[j] = tp[j - 1]; tp[j - 1] = x; }
            m.ft = nt;
            for (int k = 0; k < nt; ++k) { m.ford[k] = (unsigned char)tp[k]; m.fs[k] = (short)sof(tp[k]); }
            m.grouped = 2;
            return dim3(total, 1, 1);
        }
    }
    if (on && T > 1 && smax >= 8) {
        const int G = smax >> 3;
        const int slots = G * T + ((smax - 8 * G) * T + 7) / 8;
        if (slots <= 64) { m.grouped = 1; return dim3(8 * slots, 1, 1); }
    }
    return dim3(1, T, smax);
}

// One reduction launch for the problems of a fused launch.  Walks the PHYSICAL output elements of every problem
// ([Krows x Nn_phys], Krows = K + 1 with column sums): skipped columns and zero-only problems get zeros, the others
// the fixed-order sum over the problem's slabs (same order as k_reduce_slabs: bitwise reproducible).
struct RedProb {
    const float* slab; int S, K, Krows, Nn, Nn_phys, skip_at, skip_n;
    float* out; int ldo, accumulate; float* cs_out; int cs_accumulate;
    float* wout; int ldwo, w_col0;          // rows K + 1 .. K + 4: weighted column sums of the columns >= w_col0
};
struct RedMulti { RedProb p[BMP_WG_MAXP]; int b0[BMP_WG_MAXP + 1]; int n; };
__global__ __launch_bounds__(256) void k_reduce_multi(RedMulti m) {
    __shared__ float red[4][64];
    const int bx = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int k = 1; k < BMP_WG_MAXP; ++k) pi += (k < m.n && bx >= m.b0[k]) ? 1 : 0;
    const RedProb& q = m.p[pi];
    const size_t total = (size_t)q.Krows * q.Nn_phys;
    const size_t slab_sz = (size_t)q.Krows * q.Nn;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int nb = m.b0[pi + 1] - m.b0[pi];
    for (size_t base = (size_t)(bx - m.b0[pi]) * 64; base < total; base += (size_t)nb * 64) {
        const size_t idx = base + c;
        const int i = (int)(idx / q.Nn_phys), jp = (int)(idx % q.Nn_phys);
        const bool skipped = jp >= q.skip_at && jp < q.skip_at + q.skip_n;
        const int jl = jp - (jp >= q.skip_at + q.skip_n ? q.skip_n : 0);
        float v = 0.f;
        if (idx < total && !skipped && !(i > q.K && jp < q.w_col0)) {
#pragma unroll 8
            for (int s = g; s < q.S; s += 4) v += q.slab[(size_t)s * slab_sz + (size_t)i * q.Nn + jl];
        }
        red[g][c] = v;
        __syncthreads();
        if (g == 0 && idx < total) {
            v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (i > q.K) {
                if (jp >= q.w_col0) {
                    float* o = q.wout + (size_t)(i - q.K - 1) * q.ldwo + (jp - q.w_col0);
                    *o = q.accumulate ? (*o + v) : v;
                }
            } else {
                const bool is_cs = i == q.K;
                float* o = is_cs ? (q.cs_out + jp) : (q.out + (size_t)i * q.ldo + jp);
                *o = (is_cs ? q.cs_accumulate : q.accumulate) ? (*o + v) : v;
            }
        }
        __syncthreads();
    }
}

static void wgrad_plan(int N, int K, int Nn, int& mb, int& nb, int& S, int& rps) {
    mb = K > 32 ? 2 : 1;
    nb = Nn > 32 ? 2 : 1;
    const int tiles = ((K + 64 * mb - 1) / (64 * mb)) * ((Nn + 64 * nb - 1) / (64 * nb));
    int want = (512 + tiles - 1) / tiles;             // ~2 workgroups per CU in total
    int max_s = N / 128;                              // at least 128 rows per split
    if (max_s < 1) max_s = 1;
    S = want < max_s ? want : max_s;
    if (S < 1) S = 1;
    rps = (N + S - 1) / S;
    rps = (rps + 7) & ~7;
    S = (N + rps - 1) / rps;
}

static bool wgrad_use_lds(const WGArgs& a) {
    if (a.onehot) return (a.Nn & 3) == 0 && (a.ldy & 3) == 0 && (a.N & 31) == 0 && ((uintptr_t)a.dY & 15) == 0;
    return a.K >= 64 && a.Nn >= 64 && (a.K & 3) == 0 && (a.Nn & 3) == 0 && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 &&
           (a.N & 31) == 0 && ((uintptr_t)a.X & 15) == 0 && ((uintptr_t)a.dY & 15) == 0 &&
           (!a.X2 || ((a.ldx2 & 3) == 0 && ((uintptr_t)a.X2 & 15) == 0));
}

static void wgrad_lds_plan(int N, int K, int Nn, int& S, int& rps) {
    const int tiles = ((K + 127) / 128) * ((Nn + 127) / 128);
    int want = 512 / tiles;                           // 2 workgroups per CU and NO second round: 7 tiles x 74 splits = 518
    if (want < 1) want = 1;                           // workgroups on 512 slots made a launch wait for six stragglers
    int max_s = N / (tiles == 1 ? 128 : 256);         // at least 256 rows (8 stages) per split; a single-tile problem
    if (max_s < 1) max_s = 1;                         // takes 128-row splits so that every CU gets two workgroups
    S = want < max_s ? want : max_s;
    if (S < 1) S = 1;
    rps = (N + S - 1) / S;
    rps = (rps + 31) & ~31;
    S = (N + rps - 1) / rps;
}

size_t bmp_wgrad_ws_floats(int N, int K, int Nn) {
    int mb, nb, S, rps;
    wgrad_plan(N, K, Nn, mb, nb, S, rps);
    size_t a = (size_t)S * K * Nn;
    int S2, rps2;
    wgrad_lds_plan(N, K, Nn, S2, rps2);
    size_t b = (size_t)S2 * (K + 1) * Nn;
    size_t c = bmp_colsum_ws_floats(N, Nn);
    a = a > b ? a : b;
    return a > c ? a : c;
}

int bmp_launch_wgrad(const WGArgs& a, float* ws, hipStream_t st) {
    BMP_REQUIRE(a.N > 0 && (a.N & 7) == 0 && a.K > 0 && a.Nn > 0 && ws != nullptr);
    if (wgrad_use_lds(a)) {
        int S, rps;
        wgrad_lds_plan(a.N, a.K, a.Nn, S, rps);
        const int want_cs = a.cs != nullptr;
        WGKArgs k{a.X, a.X2, a.ldx, a.ldx2, a.dY, a.ldy, a.K, a.Nn, a.N, rps, ws, a.onehot};
        dim3 grid((a.K + 127) / 128, (a.Nn + 127) / 128, S);
        {
            // the one-hot form multiplies by an indicator: a scatter-add done as a GEMM, no algorithmic flops
            BmpProfScope prof(BMP_KCLS_WGRAD, a.onehot ? 0.0 : 2.0 * a.N * (double)a.K * a.Nn,
                              4.0 * a.N * ((a.onehot ? 1.0 : (double)a.K) + a.Nn), st,
                              a.onehot ? BMP_KID_WGRAD_ONEHOT : (a.X2 ? BMP_KID_WGRAD_X2 : BMP_KID_WGRAD));
            // (register-staged body: on the single-problem launches of the d = 256 configuration -- K = 256, 8-14 column tiles --
            //  the LDS-DMA body measured 1.5 % slower for the step; the one-hot operand is made in registers anyway)
            if (a.X2) hipLaunchKernelGGL((k_wgrad_lds<true>), grid, dim3(256), 0, st, k, want_cs);
            else hipLaunchKernelGGL((k_wgrad_lds<false>), grid, dim3(256), 0, st, k, want_cs);
        }
        BMP_LAUNCH_CHECK();
        const int Krows = a.K + want_cs;
        const size_t total = (size_t)Krows * a.Nn;
        int blocks = (int)((total + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, st, ws, S, Krows, a.Nn, a.out, a.ldo, a.accumulate,
                           want_cs ? a.K : -1, a.cs, a.accumulate | a.cs_accumulate);
        BMP_LAUNCH_CHECK();
        return 0;
    }
    int mb, nb, S, rps;
    wgrad_plan(a.N, a.K, a.Nn, mb, nb, S, rps);
    BMP_REQUIRE(a.onehot == nullptr);        // the one-hot form exists in the LDS-staged kernel only
    WGKArgs k{a.X, a.X2, a.ldx, a.ldx2, a.dY, a.ldy, a.K, a.Nn, a.N, rps, ws, nullptr};
    dim3 grid((a.K + 64 * mb - 1) / (64 * mb), (a.Nn + 64 * nb - 1) / (64 * nb), S);
    {
    BmpProfScope prof(BMP_KCLS_WGRAD, 2.0 * a.N * (double)a.K * a.Nn, 4.0 * a.N * ((double)a.K + a.Nn), st, BMP_KID_WGRAD_DIRECT);
    if (mb == 2 && nb == 2) hipLaunchKernelGGL((k_wgrad<2, 2>), grid, dim3(256), 0, st, k);
    else if (mb == 2) hipLaunchKernelGGL((k_wgrad<2, 1>), grid, dim3(256), 0, st, k);
    else if (nb == 2) hipLaunchKernelGGL((k_wgrad<1, 2>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((k_wgrad<1, 1>), grid, dim3(256), 0, st, k);
    }
    BMP_LAUNCH_CHECK();
    const size_t total = (size_t)a.K * a.Nn;
    int blocks = (int)((total + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, st, ws, S, a.K, a.Nn, a.out, a.ldo, a.accumulate, -1,
                       (float*)nullptr, 0);
    BMP_LAUNCH_CHECK();
    if (a.cs) return bmp_launch_colsum(a.dY, a.ldy, a.N, a.Nn, a.cs, a.accumulate | a.cs_accumulate, ws, st);
    return 0;
}

// n <= 3 problems with K <= 128 and no X2, one GEMM launch + one reduction each.  ws: bmp_wgrad_multi_ws_floats.
static void wgrad_multi_plan(const WGArgs* a, int n, int* S, int* rps, int* ty0) {
    int tiles = 0;
    for (int p = 0; p < n; ++p) { ty0[p] = tiles; tiles += (a[p].Nn + 127) / 128; }
    for (int p = n; p <= BMP_WG_MAXP; ++p) ty0[p] = tiles;
    for (int p = 0; p < n; ++p) {
        int s = 512 / tiles;
        int max_s = a[p].N / 256;
        if (max_s < 1) max_s = 1;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
        int r = (a[p].N + s - 1) / s;
        r = (r + 31) & ~31;
        rps[p] = r; S[p] = (a[p].N + r - 1) / r;
    }
}

size_t bmp_wgrad_multi_ws_floats(const WGArgs* a, int n) {
    int S[3], rps[3], ty0[BMP_WG_MAXP + 1];
    wgrad_multi_plan(a, n, S, rps, ty0);
    size_t tot = 0;
    for (int p = 0; p < n; ++p) tot += (size_t)S[p] * (a[p].K + 1) * a[p].Nn;
    return tot;
}

int bmp_launch_wgrad_multi(const WGArgs* a, int n, float* ws, hipStream_t st) {
    BMP_REQUIRE(n >= 1 && n <= 3 && ws != nullptr);
    WGKMulti m; memset(&m, 0, sizeof(m));
    m.n = n;
    int rps[3];
    for (int p = 0; p < n; ++p) BMP_REQUIRE(a[p].K <= 128 && !a[p].X2 && !a[p].onehot && !a[p].ridx && wgrad_use_lds(a[p]));
    wgrad_multi_plan(a, n, m.S, rps, m.ty0);
    float* slab[3];
    size_t off = 0;
    int smax = 0;
    double flops = 0, bytes = 0;
    for (int p = 0; p < n; ++p) {
        slab[p] = ws + off;
        m.want_cs[p] = a[p].cs != nullptr;
        m.p[p] = WGKArgs{a[p].X, nullptr, a[p].ldx, 0, a[p].dY, a[p].ldy, a[p].K, a[p].Nn, a[p].N, rps[p], slab[p], nullptr};
        off += (size_t)m.S[p] * (a[p].K + m.want_cs[p]) * a[p].Nn;
        if (m.S[p] > smax) smax = m.S[p];
        flops += 2.0 * a[p].N * (double)a[p].K * a[p].Nn;
        bytes += 4.0 * a[p].N * ((double)a[p].K + a[p].Nn);
    }
    {
        BmpProfScope prof(BMP_KCLS_WGRAD, flops, bytes, st, BMP_KID_WGRAD_MULTI);
        const dim3 grid = wgrad_grouped_grid(m, smax);
        const int rc = wgrad_multi_launch<0>(m, n, grid, st);
        if (rc) return rc;
    }
    BMP_LAUNCH_CHECK();
    for (int p = 0; p < n; ++p) {
        const int Krows = a[p].K + m.want_cs[p];
        const size_t total = (size_t)Krows * a[p].Nn;
        int blocks = (int)((total + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, st, slab[p], m.S[p], Krows, a[p].Nn, a[p].out, a[p].ldo,
                           a[p].accumulate, m.want_cs[p] ? a[p].K : -1, a[p].cs, a[p].accumulate | a[p].cs_accumulate);
        BMP_LAUNCH_CHECK();
    }
    return 0;
}

// ---- fused form: all problems share the row range, one GEMM launch + one reduction launch ----
// A problem with a row list weighs rfrac of a full one: the launch's 512 slots go to (column tiles x row parts) in proportion,
// so that a part of a listed problem holds about as many rows as a part of a full one -- when the lists are as long as
// expected; any other length is still computed correctly, in parts of another size (wgrad_dma_body).
static void wgrad_fused_plan(const WGArgs* a, int n, int* S, int* rps, int* ty0) {
    int tiles = 0;
    double wtiles = 0.0;
    for (int p = 0; p < n; ++p) {
        ty0[p] = tiles;
        if (!a[p].zero_only) {
            const int t = (a[p].Nn + 127) / 128;
            tiles += t;
            wtiles += t * (a[p].ridx ? (double)a[p].rfrac : 1.0);
        }
    }
    for (int p = n; p <= BMP_WG_MAXP; ++p) ty0[p] = tiles;
    for (int p = 0; p < n; ++p) {
        int s = (int)(512.0 / (wtiles > 0.0 ? wtiles : 1.0));        // 2 workgroups per CU and no second round
        int max_s = a[p].N / 256;                     // at least 8 stages per split
        if (max_s < 1) max_s = 1;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
        int r = (a[p].N + s - 1) / s;
        r = (r + 31) & ~31;
        rps[p] = r;
        const int sfull = (a[p].N + r - 1) / r;
        if (a[p].zero_only) S[p] = 0;
        else if (a[p].ridx) { int sp = (int)(sfull * (double)a[p].rfrac + 0.5); S[p] = sp < 1 ? 1 : (sp > sfull ? sfull : sp); }
        else S[p] = sfull;
    }
}

size_t bmp_wgrad_fused_ws_floats(const WGArgs* a, int n) {
    int S[BMP_WG_MAXP], rps[BMP_WG_MAXP], ty0[BMP_WG_MAXP + 1];
    wgrad_fused_plan(a, n, S, rps, ty0);
    size_t tot = 128;                                // the zero row of the listed problems
    for (int p = 0; p < n; ++p) tot += (size_t)S[p] * (a[p].K + 5) * a[p].Nn;
    return tot;
}

bool bmp_wgrad_fused_lists_ok(int N) { return wgrad_dma_enabled() && N > 0 && (N % WD_RS) == 0; }

int bmp_launch_wgrad_fused(const WGArgs* a, int n, float* ws, hipStream_t st, int kid) {
    BMP_REQUIRE(n >= 1 && n <= BMP_WG_MAXP && ws != nullptr);
    WGKMulti m; memset(&m, 0, sizeof(m));
    RedMulti r; memset(&r, 0, sizeof(r));
    m.n = n; r.n = n;
    int rps[BMP_WG_MAXP];
    bool listed = false;
    for (int p = 0; p < n; ++p) {
        BMP_REQUIRE(a[p].K <= 128 && !a[p].onehot && (a[p].zero_only || wgrad_use_lds(a[p])));
        if (a[p].ridx) {
            BMP_REQUIRE(a[p].rcnt && !a[p].X2 && !a[p].wrow && a[p].skip_n == 0 && !a[p].zero_only && bmp_wgrad_fused_lists_ok(a[p].N));
            listed = true;
        }
        BMP_REQUIRE(a[p].skip_n == 0 || ((a[p].skip_at & 127) == 0 && (a[p].skip_n & 3) == 0));
        BMP_REQUIRE(a[p].N == a[0].N);
        BMP_REQUIRE(!a[p].wrow || (a[p].cs && a[p].wout && a[p].skip_n == 0 && (a[p].w_col0 & 127) == 0));
    }
    wgrad_fused_plan(a, n, m.S, rps, m.ty0);
    float* zrow = ws;                                // (the first 128 floats stay unused: bmp_wgrad_fused_ws_floats counts them)
    (void)listed;
    size_t off = 128;
    int smax = 0, rb = 0;
    double flops = 0, bytes = 0;
    for (int p = 0; p < n; ++p) {
        const int want_cs = a[p].cs != nullptr && !a[p].zero_only;
        m.want_cs[p] = want_cs;
        WGKArgs k{a[p].X, a[p].X2, a[p].ldx, a[p].ldx2, a[p].dY, a[p].ldy, a[p].K, a[p].Nn, a[p].N, rps[p], ws + off, nullptr};
        k.skip_at = a[p].skip_at; k.skip_n = a[p].skip_n;
        m.p[p] = k;
        RedProb& q = r.p[p];
        const int nw = (a[p].wrow && want_cs) ? 4 : 0;
        k.wrow = nw ? a[p].wrow : nullptr; k.w_col0 = a[p].w_col0;
        if (a[p].ridx) { k.ridx = a[p].ridx; k.rcnt = a[p].rcnt; k.zrow = zrow; k.nsplit = m.S[p]; }
        m.p[p] = k;
        q.slab = ws + off; q.S = m.S[p]; q.K = a[p].K; q.Krows = a[p].K + want_cs + nw; q.Nn = a[p].Nn;
        q.wout = a[p].wout; q.ldwo = a[p].ldwo; q.w_col0 = a[p].w_col0;
        q.Nn_phys = a[p].Nn + a[p].skip_n; q.skip_at = a[p].skip_n ? a[p].skip_at : 0x7fffffff; q.skip_n = a[p].skip_n;
        q.out = a[p].out; q.ldo = a[p].ldo; q.accumulate = a[p].accumulate; q.cs_out = a[p].cs;
        q.cs_accumulate = a[p].accumulate | a[p].cs_accumulate;
        off += (size_t)m.S[p] * q.Krows * a[p].Nn;
        if (m.S[p] > smax) smax = m.S[p];
        if (!a[p].zero_only) {
            flops += 2.0 * a[p].N * (double)a[p].K * a[p].Nn;
            bytes += 4.0 * a[p].N * ((double)a[p].K + a[p].Nn);
        }
        r.b0[p] = rb;
        int blocks = (int)(((size_t)q.Krows * q.Nn_phys + 63) / 64);
        if (blocks > 1024) blocks = 1024;
        if (a[p].zero_only && a[p].accumulate) blocks = 0;          // nothing to add
        rb += blocks;
    }
    for (int p = n; p <= BMP_WG_MAXP; ++p) r.b0[p] = rb;
    if (m.ty0[BMP_WG_MAXP] > 0) {
        BmpProfScope prof(BMP_KCLS_WGRAD, flops, bytes, st, kid);
        const dim3 grid = wgrad_grouped_grid(m, smax);
        const int rc = wgrad_multi_launch<1>(m, n, grid, st);
        if (rc) return rc;
    }
    BMP_LAUNCH_CHECK();
    if (rb > 0) hipLaunchKernelGGL(k_reduce_multi, dim3(rb), dim3(256), 0, st, r);
    BMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): slab[s][n] = sum_{rows of split s} dY[row, n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dY, int ldy, int N, int Nn, int rps, float* slab) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int s = blockIdx.y;
    const int r_begin = s * rps;
    const int r_end = (r_begin + rps) < N ? (r_begin + rps) : N;
    float v = 0.f;
    if (n < Nn)
        for (int r = r_begin + g; r < r_end; r += 4) v += dY[(size_t)r * ldy + n];
    red[g][c] = v;
    __syncthreads();
    if (g == 0 && n < Nn) slab[(size_t)s * Nn + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

static void colsum_plan(int N, int& S, int& rps) {
    S = N / 64;                 // a split walks its rows four at a time: keep the dependent chain short
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    rps = (N + S - 1) / S;
    S = (N + rps - 1) / rps;
}

size_t bmp_colsum_ws_floats(int N, int Nn) {
    int S, rps;
    colsum_plan(N, S, rps);
    return (size_t)S * Nn;
}

int bmp_launch_colsum(const float* dY, int ldy, int N, int Nn, float* out, int accumulate, float* ws, hipStream_t st) {
    BMP_REQUIRE(N > 0 && Nn > 0 && ws != nullptr);
    int S, rps;
    colsum_plan(N, S, rps);
    hipLaunchKernelGGL(k_colsum, dim3((Nn + 63) / 64, S), dim3(256), 0, st, dY, ldy, N, Nn, rps, ws);
    BMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_slabs, dim3((Nn + 63) / 64), dim3(256), 0, st, ws, S, 1, Nn, out, Nn, accumulate, -1,
                       (float*)nullptr, 0);
    BMP_LAUNCH_CHECK();
    return 0;
}
