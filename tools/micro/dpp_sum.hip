#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
__device__ __forceinline__ float xor_dpp4(float v) {
    int x = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);
    return __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    {   // xor 32
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {   // xor 16
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // xor 8 = row_ror:8
    v += xor_dpp4(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // xor 2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // xor 1
    return v;
}
__device__ __forceinline__ float wave_sum_ref(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ void k(const float* x, float* a, float* b) {
    float v = x[threadIdx.x];
    a[threadIdx.x] = wave_sum_dpp(v);
    b[threadIdx.x] = wave_sum_ref(v);
}
int main() {
    float h[64], ha[64], hb[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f / (i + 1) * ((i % 3) ? 1.f : -1.37f) * 1e3f / (1 + (i * 7919) % 13);
    float *x, *a, *b; hipMalloc(&x, 256); hipMalloc(&a, 256); hipMalloc(&b, 256);
    hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, a, b);
    hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (memcmp(&ha[i], &hb[i], 4)) ++bad;
    printf("mismatching lanes: %d  (%.9g vs %.9g)\n", bad, ha[0], hb[0]);
    return bad != 0;
}
