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
