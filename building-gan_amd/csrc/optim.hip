// Flat-buffer Adam step (torch.optim.Adam semantics without weight decay / amsgrad).
#include "common.h"

namespace {

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, const int32_t* __restrict__ step_dev) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (step_dev) {   // step count lives on the device (hipGraph replays must not bake it in)
        const double t = (double)step_dev[0];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i];
        float mi = m[i] + (gi - m[i]) * (1.f - b1);          // lerp form used by torch
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int bg_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                           float lr, float beta1, float beta2, float eps, int32_t step,
                           const int32_t* step_dev, bgStream stream) {
    if (n < 0 || (!step_dev && step < 1)) return BG_ERR_ARG;
    if (step_dev) step = 1;
    if (n == 0) return BG_OK;
    if (!param || !grad || !m || !v) return BG_ERR_ARG;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    k_adam<<<bg_grid_for(n, 256, 2048), 256, 0, (hipStream_t)stream>>>(param, grad, m, v, n, lr, beta1, beta2, eps,
                                                                      (float)bc1, (float)sqrt(bc2), step_dev);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// ---------------------------------------------------------------- fills and copies as kernels (see common.h)
namespace {
__global__ void k_fill_zero(uint32_t* p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}
__global__ void k_fill_zero16(uint4* p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
// rows x width dwords, pitches in dwords
__global__ void k_copy2d(uint32_t* dst, int64_t ldd, const uint32_t* src, int64_t lds_, int64_t width, int64_t rows) {
    const int64_t n = width * rows, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / width, c = i - r * width;
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}
}  // namespace

extern "C" int bg_zero_bytes(void* p, int64_t bytes, bgStream stream) {
    if (bytes < 0 || bytes % 4 != 0 || (bytes > 0 && (!p || (reinterpret_cast<uintptr_t>(p) & 3u)))) return BG_ERR_ARG;
    if (bytes == 0) return BG_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0 && bytes % 16 == 0)
        k_fill_zero16<<<bg_grid_for(bytes / 16, 256, 4096), 256, 0, s>>>(static_cast<uint4*>(p), bytes / 16);
    else
        k_fill_zero<<<bg_grid_for(bytes / 4, 256, 4096), 256, 0, s>>>(static_cast<uint32_t*>(p), bytes / 4);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

extern "C" int bg_copy2d_bytes(void* dst, int64_t dpitch, const void* src, int64_t spitch, int64_t width, int64_t rows,
                               bgStream stream) {
    if (width < 0 || rows < 0 || width % 4 != 0 || dpitch % 4 != 0 || spitch % 4 != 0) return BG_ERR_ARG;
    if (width == 0 || rows == 0) return BG_OK;
    if (!dst || !src || dpitch < width || spitch < width || (reinterpret_cast<uintptr_t>(dst) & 3u) ||
        (reinterpret_cast<uintptr_t>(src) & 3u))
        return BG_ERR_ARG;
    k_copy2d<<<bg_grid_for(width / 4 * rows, 256, 4096), 256, 0, (hipStream_t)stream>>>(
        static_cast<uint32_t*>(dst), dpitch / 4, static_cast<const uint32_t*>(src), spitch / 4, width / 4, rows);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

