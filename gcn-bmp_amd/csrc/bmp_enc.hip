// The encoder's own row layout and the two index kernels that connect it with the per-instance layout of a pair batch.
//
// The reference zero-pads every molecule of a batch side to the side's largest atom count and masks nothing
// (models/ggnn.py:340,603; nie_coattention.py:347-349); the packed layout (bmp/packed.py) keeps ONE virtual pad row per
// molecule instance, whose multiplicity stands for all of its padded positions.  Inside the encoder that row is the same
// for every molecule of a batch: atom id 0, no bonds, so the same state after every propagation step
// (models/ggnn.py:215-263), and its gradient enters the weight gradients and the embedding only through sums.  The
// ENCODER LAYOUT therefore holds the real atoms of every encoded molecule and one pad row per TILE (the first row behind
// the tile's last molecule), in tiles of 1..4 live 32-row blocks chosen so that the chip's 256 CUs finish together
// (bmp_collate_plan_enc); with de-duplication a molecule that occurs several times in the batch is encoded once
// (SURVEY.md 8(d) caveat).  The consumers of the atom states -- readout, co-attention -- keep the per-instance layout:
//   bmp_encrows_expand : X[instance row]  = h[encoder row of that atom]   (pad row of the instance <- its tile's pad row)
//   bmp_encrows_reduce : dh[encoder row]  = sum of dX over the instance rows that were copied from it, in a fixed order
// No atomics: bitwise reproducible.  HBM-bound index work.
#include "bmp_kernels.h"

// out [N_inst x d], d % 4 == 0.  row_mol [N_inst]: instance of every row (-1: none -> zeros); inst_row0 [I]; uid [I]: encoded
// molecule of every instance; enc_row0 / enc_n / enc_pad [U]: first encoder row, real atoms and the tile's pad row of every
// encoded molecule.
__global__ __launch_bounds__(256) void k_encrows_expand(const float* __restrict__ h, int d4, const int* __restrict__ row_mol,
                                                        const int* __restrict__ inst_row0, const int* __restrict__ uid,
                                                        const int* __restrict__ enc_row0, const int* __restrict__ enc_n,
                                                        const int* __restrict__ enc_pad, int N_inst, float* __restrict__ out) {
    const size_t total = (size_t)N_inst * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d4), c = (int)(idx % d4);
        const int inst = row_mol[r];
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (inst >= 0) {
            const int u = uid[inst], l = r - inst_row0[inst];
            const int src = l < enc_n[u] ? enc_row0[u] + l : enc_pad[u];
            v = *(const f32x4*)(h + ((size_t)src * d4 + c) * 4);
        }
        *(f32x4*)(out + idx * 4) = v;
    }
}

// dh [N_enc x d].  erow_mol [N_enc]: u >= 0: a real atom of encoded molecule u; -2 - t: the pad row of tile t; -1: a dead
// row (zeros).  uptr [U + 1] / uinst: the instances of every encoded molecule, ascending; tptr [T + 1] / tmols: the encoded
// molecules of every tile, ascending.
__global__ __launch_bounds__(256) void k_encrows_reduce(const float* __restrict__ dX, int d4, const int* __restrict__ erow_mol,
                                                        const int* __restrict__ enc_row0, const int* __restrict__ enc_n,
                                                        const int* __restrict__ uptr, const int* __restrict__ uinst,
                                                        const int* __restrict__ inst_row0, const int* __restrict__ tptr,
                                                        const int* __restrict__ tmols, int N_enc, float* __restrict__ dh) {
    const size_t total = (size_t)N_enc * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d4), c = (int)(idx % d4);
        const int m = erow_mol[r];
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (m >= 0) {
            const int l = r - enc_row0[m];
            for (int k = uptr[m]; k < uptr[m + 1]; ++k)
                acc += *(const f32x4*)(dX + ((size_t)(inst_row0[uinst[k]] + l) * d4 + c) * 4);
        } else if (m <= -2) {
            const int t = -2 - m;
            for (int q = tptr[t]; q < tptr[t + 1]; ++q) {
                const int u = tmols[q], l = enc_n[u];          // the instance's pad row follows its real atoms
                for (int k = uptr[u]; k < uptr[u + 1]; ++k)
                    acc += *(const f32x4*)(dX + ((size_t)(inst_row0[uinst[k]] + l) * d4 + c) * 4);
            }
        }
        *(f32x4*)(dh + idx * 4) = acc;
    }
}

static inline int er_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int bmp_encrows_expand(const float* h, int d, const int* row_mol, const int* inst_row0, const int* uid,
                                  const int* enc_row0, const int* enc_n, const int* enc_pad, int N_inst, float* out, hipStream_t st) {
    BMP_REQUIRE(h && row_mol && inst_row0 && uid && enc_row0 && enc_n && enc_pad && out && N_inst > 0 && d > 0 && (d & 3) == 0);
    BMP_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)out & 15) == 0);
    hipLaunchKernelGGL(k_encrows_expand, dim3(er_blocks((size_t)N_inst * (d / 4))), dim3(256), 0, st, h, d / 4, row_mol, inst_row0,
                       uid, enc_row0, enc_n, enc_pad, N_inst, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_encrows_reduce(const float* dX, int d, const int* erow_mol, const int* enc_row0, const int* enc_n, const int* uptr,
                                  const int* uinst, const int* inst_row0, const int* tptr, const int* tmols, int N_enc, float* dh,
                                  hipStream_t st) {
    BMP_REQUIRE(dX && erow_mol && enc_row0 && enc_n && uptr && uinst && inst_row0 && tptr && tmols && dh && N_enc > 0 && d > 0 && (d & 3) == 0);
    BMP_REQUIRE(((uintptr_t)dX & 15) == 0 && ((uintptr_t)dh & 15) == 0);
    hipLaunchKernelGGL(k_encrows_reduce, dim3(er_blocks((size_t)N_enc * (d / 4))), dim3(256), 0, st, dX, d / 4, erow_mol, enc_row0,
                       enc_n, uptr, uinst, inst_row0, tptr, tmols, N_enc, dh);
    BMP_LAUNCH_CHECK();
    return 0;
}
