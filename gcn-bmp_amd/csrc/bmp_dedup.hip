// De-duplicated encoding of a drug-pair batch (SURVEY.md 8(d) "de-duplication caveat"; 7 "the big non-kernel win"): a batch of
// B pairs holds 2B molecule INSTANCES but at most 544 distinct drugs (setting.py:30), and an atom's state after the
// propagation steps depends on its molecule alone (models/ggnn.py:584-654: the padded positions of a batch never touch a real
// atom).  The encoder can therefore run once per DISTINCT molecule; the co-attention, whose softmaxes see the batch side's
// padding through the pad row's multiplicity, keeps the per-instance layout.  Two index kernels connect the layouts:
//   bmp_molrows_expand : X[instance row]  = h[distinct molecule's row]            (forward)
//   bmp_molrows_reduce : dh[distinct row] = sum over the molecule's instances, in instance order, of dX[instance row]
// The reduction walks a molecule's instances in a fixed order (no atomics): bitwise reproducible.  HBM-bound index work.
#include "bmp_kernels.h"

// out [N_inst x d] (d % 4 == 0).  row_mol [N_inst]: instance of every row (-1: none, the row is zero-filled);
// inst_row0 [I]: first row of every instance; uid [I]: its distinct molecule; urow0 [U]: first row of every distinct molecule.
__global__ __launch_bounds__(256) void k_molrows_expand(const float* __restrict__ hU, int d4, const int* __restrict__ row_mol,
                                                        const int* __restrict__ inst_row0, const int* __restrict__ uid,
                                                        const int* __restrict__ urow0, int N_inst, float* __restrict__ out) {
    const size_t total = (size_t)N_inst * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d4), c = (int)(idx % d4);
        const int inst = row_mol[r];
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (inst >= 0) {
            const int src = urow0[uid[inst]] + (r - inst_row0[inst]);
            v = *(const f32x4*)(hU + ((size_t)src * d4 + c) * 4);
        }
        *(f32x4*)(out + idx * 4) = v;
    }
}

// dhU [N_U x d].  urow_mol [N_U]: distinct molecule of every row (-1: none -> 0); uptr [U + 1] / uinst: the instances of
// every distinct molecule, ascending.
__global__ __launch_bounds__(256) void k_molrows_reduce(const float* __restrict__ dX, int d4, const int* __restrict__ urow_mol,
                                                        const int* __restrict__ urow0, const int* __restrict__ uptr,
                                                        const int* __restrict__ uinst, const int* __restrict__ inst_row0, int N_U,
                                                        float* __restrict__ dhU) {
    const size_t total = (size_t)N_U * d4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / d4), c = (int)(idx % d4);
        const int u = urow_mol[r];
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (u >= 0) {
            const int l = r - urow0[u];
            for (int k = uptr[u]; k < uptr[u + 1]; ++k)
                acc += *(const f32x4*)(dX + ((size_t)(inst_row0[uinst[k]] + l) * d4 + c) * 4);
        }
        *(f32x4*)(dhU + idx * 4) = acc;
    }
}

static inline int dd_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int bmp_molrows_expand(const float* hU, int d, const int* row_mol, const int* inst_row0, const int* uid,
                                  const int* urow0, int N_inst, float* out, hipStream_t st) {
    BMP_REQUIRE(hU && row_mol && inst_row0 && uid && urow0 && out && N_inst > 0 && d > 0 && (d & 3) == 0);
    BMP_REQUIRE(((uintptr_t)hU & 15) == 0 && ((uintptr_t)out & 15) == 0);
    hipLaunchKernelGGL(k_molrows_expand, dim3(dd_blocks((size_t)N_inst * (d / 4))), dim3(256), 0, st, hU, d / 4, row_mol, inst_row0,
                       uid, urow0, N_inst, out);
    BMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int bmp_molrows_reduce(const float* dX, int d, const int* urow_mol, const int* urow0, const int* uptr, const int* uinst,
                                  const int* inst_row0, int N_U, float* dhU, hipStream_t st) {
    BMP_REQUIRE(dX && urow_mol && urow0 && uptr && uinst && inst_row0 && dhU && N_U > 0 && d > 0 && (d & 3) == 0);
    BMP_REQUIRE(((uintptr_t)dX & 15) == 0 && ((uintptr_t)dhU & 15) == 0);
    hipLaunchKernelGGL(k_molrows_reduce, dim3(dd_blocks((size_t)N_U * (d / 4))), dim3(256), 0, st, dX, d / 4, urow_mol, urow0, uptr,
                       uinst, inst_row0, N_U, dhU);
    BMP_LAUNCH_CHECK();
    return 0;
}
