// Common definitions for the gfx950 (MI355X / CDNA4) kernels of the GCN-BMP hot path.
// Wave = 64 lanes; fp32-input MFMA v_mfma_f32_32x32x2_f32 is the dense workhorse
// (exact f32, 64 FLOP/clk/SIMD): parity with the reference's fp32 math needs 1e-4,
// so there is no reduced-precision path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BMP_R 128            // rows per tile (bmp.packed.DEFAULT_R)
#define BMP_LDS_LD 68        // LDS row stride (floats) of a 64-wide K chunk: 272 B = 17 x 16 B
                             // -> the 16 lanes of every ds_read_b128 lane group hit 16 distinct slots

#define BMP_LAUNCH_CHECK()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return (int)e__;                   \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set once per (kernel, device), not per
// launch and not once per process (a host that drives two devices from one process would find the second one unset).
// Lock-free table, safe from several host threads (setting an attribute twice is harmless).
#include <atomic>
struct BmpAttrSlot { std::atomic<const void*> fn{nullptr}; std::atomic<unsigned long long> devs{0}; };
inline int bmp_lds_attr(const void* fn, size_t bytes) {
    static BmpAttrSlot tab[96];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    for (int i = 0; i < 96; ++i) {
        const void* cur = tab[i].fn.load(std::memory_order_acquire);
        if (cur == nullptr) {
            const void* expect = nullptr;
            cur = tab[i].fn.compare_exchange_strong(expect, fn, std::memory_order_acq_rel) ? fn : expect;
        }
        if (cur != fn) continue;
        if (tab[i].devs.load(std::memory_order_acquire) & bit) return 0;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        tab[i].devs.fetch_or(bit, std::memory_order_release);
        return 0;
    }
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);      // table full: per launch
}

#define BMP_REQUIRE(cond)                                         \
    do {                                                          \
        if (!(cond)) return -1000 - __LINE__;                     \
    } while (0)

// MFMA 32x32x2 f32 fragment maps (cdna guide section 3):
//   A operand: lane l holds A[i = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D      : lane l, reg r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// The K order inside a sum is free as long as A and B agree, so a lane's four consecutive k
// values (one ds_read_b128 / four row loads) feed four MFMAs: lanes 0-31 carry k0..k0+3,
// lanes 32-63 carry k0+4..k0+7.
__device__ __forceinline__ f32x16 bmp_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int bmp_acc_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// Hardware transcendental forms (v_exp_f32 / v_rcp_f32, ~1e-7 absolute error on the outputs): three
// instructions instead of the ~30 of ocml's expf/tanhf.  The gate nonlinearities run in the VALU shadow of
// the MFMA phases and the co-attention softmaxes are chains of exps, so the libm forms were measurable;
// the 1e-4 parity budget (tests/) is two to three orders of magnitude above this error.
__device__ __forceinline__ float bmp_exp(float x) { return __expf(x); }
__device__ __forceinline__ float bmp_sigmoid(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }
// 1 - 2/(e^{2x}+1): saturates to +-1 without NaN (e -> inf gives 1, e -> 0 gives -1)
__device__ __forceinline__ float bmp_tanh(float x) { return 1.0f - __fdividef(2.0f, __expf(2.0f * x) + 1.0f); }

enum { BMP_ACT_NONE = 0, BMP_ACT_SIGMOID = 1, BMP_ACT_TANH = 2, BMP_ACT_RELU = 3 };

__device__ __forceinline__ float bmp_act(int act, float x) {
    switch (act) {
        case BMP_ACT_SIGMOID: return bmp_sigmoid(x);
        case BMP_ACT_TANH: return bmp_tanh(x);
        case BMP_ACT_RELU: return x > 0.f ? x : 0.f;
        default: return x;
    }
}
// derivative of act expressed through its OUTPUT y
__device__ __forceinline__ float bmp_dact(int act, float y) {
    switch (act) {
        case BMP_ACT_SIGMOID: return y * (1.f - y);
        case BMP_ACT_TANH: return 1.f - y * y;
        case BMP_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}
