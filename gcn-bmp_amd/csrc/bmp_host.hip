// Host-side glue of a training step as three kernels instead of ~100 tiny framework launches:
//   * bmp_gather_sum: every kernel-layout weight array of the model (K-major transposes, K4 packs, folded GRU
//     matrices, concatenated readout / co-attention operands) in ONE launch from the flat parameter buffer, and the
//     reverse -- every parameter gradient from the buffers the weight-gradient kernels wrote -- in one more.
//     Both directions are the same operation: out[i] = sum over <= K table entries of src[idx[i][k]].
//     The tables are built once per model on the host (bmp/plan.py).
//   * bmp_adam_step: chainer.optimizers.Adam's update rule (train_ddi_modify.py:289) over the flat buffers.
#include "bmp_common.h"

__global__ __launch_bounds__(256) void k_gather_sum(float* __restrict__ dst, int n, const float* __restrict__ src,
                                                    const int* __restrict__ idx, int K, int accumulate) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float v = 0.f;
        for (int k = 0; k < K; ++k) {               // fixed order: reproducible sums
            const int j = idx[(size_t)k * n + i];   // table is [K][n]: coalesced per term
            if (j >= 0) v += src[j];
        }
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

// dst[i] (=|+=) sum_k src[idx[k*n + i]] over the entries with idx >= 0.
extern "C" int bmp_gather_sum(float* dst, int n, const float* src, const int* idx, int K, int accumulate, hipStream_t st) {
    BMP_REQUIRE(n >= 0 && K >= 1 && dst && src && idx);
    if (n == 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_gather_sum, dim3(blocks), dim3(256), 0, st, dst, n, src, idx, K, accumulate);
    BMP_LAUNCH_CHECK();
    return 0;
}

// chainer Adam: m += (1-b1)(g-m); v += (1-b2)(g^2-v); p -= alpha_t * m / (sqrt(v) + eps) + wd * p
// (alpha_t = alpha * sqrt(1-b2^t)/(1-b1^t) is computed by the caller; eps sits outside the bias correction).
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int n, float alpha_host, const float* __restrict__ alpha_dev,
                                              float b1, float b2, float eps, float wd, float gscale) {
    const float alpha_t = alpha_dev ? alpha_dev[0] : alpha_host;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float pi = p[i];
        if (wd != 0.f) pi *= (1.f - wd);
        p[i] = pi - alpha_t * mi / (sqrtf(vi) + eps);
    }
}

// alpha_t_dev (optional, device): read instead of alpha_t -- the step-dependent factor of a launch recorded in a HIP graph.
extern "C" int bmp_adam_step(float* p, const float* g, float* m, float* v, int n, float alpha_t, const float* alpha_t_dev,
                             float beta1, float beta2, float eps, float weight_decay_rate, float grad_scale, hipStream_t st) {
    BMP_REQUIRE(n >= 0 && p && g && m && v);
    if (n == 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, alpha_t, alpha_t_dev, beta1, beta2, eps,
                       weight_decay_rate, grad_scale);
    BMP_LAUNCH_CHECK();
    return 0;
}
