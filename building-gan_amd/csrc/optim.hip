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
// up to 8 fills / strided copies as ONE launch (blockIdx.y = the job): the composites' preparation steps
struct FillCopyJobs {
    uint32_t* dst[8]; const uint32_t* src[8];           // src == nullptr: zero fill
    int64_t ldd[8], lds_[8], width[8], rows[8];         // dwords
    int vec[8];                                         // every address, pitch and width a multiple of 16 bytes
};
__global__ void k_fill_copy_batch(FillCopyJobs j) {
    const int b = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j.vec[b]) {
        const int64_t w4 = j.width[b] >> 2, n = w4 * j.rows[b], ldd = j.ldd[b] >> 2, lds4 = j.lds_[b] >> 2;
        uint4* d = reinterpret_cast<uint4*>(j.dst[b]);
        const uint4* s = reinterpret_cast<const uint4*>(j.src[b]);
        for (int64_t i = t0; i < n; i += stride) {
            const int64_t r = i / w4, c = i - r * w4;
            d[r * ldd + c] = s ? s[r * lds4 + c] : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        const int64_t w = j.width[b], n = w * j.rows[b];
        for (int64_t i = t0; i < n; i += stride) {
            const int64_t r = i / w, c = i - r * w;
            j.dst[b][r * j.ldd[b] + c] = j.src[b] ? j.src[b][r * j.lds_[b] + c] : 0u;
        }
    }
}
}  // namespace

extern "C" int bg_fill_copy_batch(int32_t n_jobs, void* const* dst, const int64_t* dpitch, const void* const* src,
                                  const int64_t* spitch, const int64_t* width, const int64_t* rows, bgStream stream) {
    if (n_jobs < 0 || n_jobs > 8 || (n_jobs > 0 && (!dst || !dpitch || !src || !spitch || !width || !rows))) return BG_ERR_ARG;
    FillCopyJobs j{};
    int n = 0;
    int64_t most = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (width[i] < 0 || rows[i] < 0 || width[i] % 4 != 0 || dpitch[i] % 4 != 0 || (src[i] && spitch[i] % 4 != 0)) return BG_ERR_ARG;
        if (width[i] == 0 || rows[i] == 0) continue;
        if (!dst[i] || dpitch[i] < width[i] || (src[i] && spitch[i] < width[i]) || (reinterpret_cast<uintptr_t>(dst[i]) & 3u) ||
            (reinterpret_cast<uintptr_t>(src[i]) & 3u))
            return BG_ERR_ARG;
        j.dst[n] = static_cast<uint32_t*>(dst[i]); j.src[n] = static_cast<const uint32_t*>(src[i]);
        j.ldd[n] = dpitch[i] / 4; j.lds_[n] = src[i] ? spitch[i] / 4 : 0; j.width[n] = width[i] / 4; j.rows[n] = rows[i];
        j.vec[n] = (reinterpret_cast<uintptr_t>(dst[i]) % 16 == 0 && reinterpret_cast<uintptr_t>(src[i]) % 16 == 0 &&
                    dpitch[i] % 16 == 0 && (!src[i] || spitch[i] % 16 == 0) && width[i] % 16 == 0) ? 1 : 0;
        const int64_t items = (j.vec[n] ? j.width[n] / 4 : j.width[n]) * j.rows[n];
        if (items > most) most = items;
        ++n;
    }
    if (n == 0) return BG_OK;
    k_fill_copy_batch<<<dim3((unsigned)bg_grid_for(most, 256, 2048), (unsigned)n), 256, 0, (hipStream_t)stream>>>(j);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

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

