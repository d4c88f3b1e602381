// Shared helpers for the gfx950 kernels of libbgan_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bgan.h"

#define BG_WAVE 64

#define BG_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return BG_ERR_LAUNCH;        \
    } while (0)

#define BG_HIP(call)                                        \
    do {                                                    \
        if ((call) != hipSuccess) return BG_ERR_LAUNCH;     \
    } while (0)

// Tuning knobs.  The library is stateless and reads NO environment variable in its default build: every knob below
// is the constant `dflt`.  A build with -DBG_TUNING_KNOBS (tools/ A/B runs only: `make tuning`) reads the named
// variable once per process instead; kernel SELECTION may then differ, results never do (every variant is tested).
#ifdef BG_TUNING_KNOBS
#include <stdlib.h>
static inline int bg_tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int bg_tune_int(const char*, int dflt) { return dflt; }
#endif

// Zero fills and device-to-device copies go through bg_zero_bytes / bg_copy2d_bytes (optim.hip, include/bgan.h):
// KERNELS of this library, never hipMemsetAsync / hipMemcpy(2D)Async.  Those become memset / memcpy NODES under stream
// capture, and on this stack (ROCm 7.2, gfx950) a replayed graph was seen to run such a node out of order with the
// kernel nodes around it when other work had been queued on the stream between two graph launches: the coefficient
// table of bg_d_coef_forward_f32 came out zeroed AFTER its first rows were written (tools/lab/nan_probe3.py;
// DESIGN.md section 6, round 3).  Kernel nodes keep their order.
static inline hipError_t bg_zero_async(void* p, size_t bytes, hipStream_t s) {
    return bg_zero_bytes(p, (int64_t)bytes, (bgStream)s) == BG_OK ? hipSuccess : hipErrorLaunchFailure;
}
static inline hipError_t bg_copy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                                         hipStream_t s) {
    return bg_copy2d_bytes(dst, (int64_t)dpitch, src, (int64_t)spitch, (int64_t)width, (int64_t)rows, (bgStream)s) == BG_OK
               ? hipSuccess : hipErrorLaunchFailure;
}
static inline hipError_t bg_copy_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
    return bg_copy2d_async(dst, bytes, src, bytes, bytes, 1, s);
}

static inline bool bg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int bg_grid_for(int64_t work, int per_block, int cap = 1 << 20) {
    int64_t g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

// Cross-lane helpers that stay off the LDS crossbar (`__shfl*` compiles to ds_bpermute_b32, an LDS instruction with LDS
// latency): DPP row operations inside a row of 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap across rows.
//   bg_lane_xor<OFF>(v)[l] = v[l ^ OFF]   for OFF = 1, 2, 4, 8 (DPP: quad_perm, two masked row shifts, row_ror:8)
template <int OFF>
__device__ __forceinline__ float bg_lane_xor(float v) {
    const int x = __float_as_int(v);
    int t;
    if (OFF == 1) t = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
    else if (OFF == 2) t = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);       // quad_perm [2,3,0,1]
    else if (OFF == 4) {
        t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);                     // lanes 0-3, 8-11 of a row: from l + 4
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                     // lanes 4-7, 12-15: from l - 4
    } else t = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);                  // row_ror:8
    return __int_as_float(t);
}
// the two halves of v[l] (op) v[l ^ 16] / v[l ^ 32]: after the swap a holds one partner in every lane, b the other
__device__ __forceinline__ void bg_swap16(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void bg_swap32(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// Sum over the 64 lanes, every lane gets it: the xor butterfly 32, 16, 8, 4, 2, 1 -- the same pairs in the same order as
// `for (off = 32; off; off >>= 1) v += __shfl_xor(v, off)`, hence the same bits (tools/micro/dpp_sum.hip checks that).
__device__ __forceinline__ float bg_wave_sum(float v) {
    float a, b;
    bg_swap32(v, a, b); v = a + b;
    bg_swap16(v, a, b); v = a + b;
    v += bg_lane_xor<8>(v);
    v += bg_lane_xor<4>(v);
    v += bg_lane_xor<2>(v);
    v += bg_lane_xor<1>(v);
    return v;
}
__device__ __forceinline__ float bg_wave_max(float v) {
    float a, b;
    bg_swap32(v, a, b); v = fmaxf(a, b);
    bg_swap16(v, a, b); v = fmaxf(a, b);
    v = fmaxf(v, bg_lane_xor<8>(v));
    v = fmaxf(v, bg_lane_xor<4>(v));
    v = fmaxf(v, bg_lane_xor<2>(v));
    v = fmaxf(v, bg_lane_xor<1>(v));
    return v;
}
// v of lane `lane` (the same, wave-uniform `lane` in every thread) through the scalar unit instead of the LDS crossbar
__device__ __forceinline__ float bg_readlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int32_t bg_readlane(int32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__device__ __forceinline__ float4 bg_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void bg_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 bg_fma4(float s, float4 a, float4 acc) {
    acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y);
    acc.z = fmaf(s, a.z, acc.z); acc.w = fmaf(s, a.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 bg_add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// tanh as (e - 1) / (e + 1), e = e^{2|x|}, on the hardware's exp2 and reciprocal (1 ulp each): seven VALU
// instructions where the library call spends ~27.  e - 1 is exact for e < 2, so the absolute error is that of e
// itself over (e + 1): <= ~6e-8 near zero, <= 1.5 ulp of the result from |x| ~ 0.3 up (the attention logits sum 128
// such terms of mixed sign at arguments of order one).  |x| is clamped at 10 (tanh rounds to 1 there; keeps e
// finite) and the quotient at 1, so |tanh| <= 1 and 1 - t^2 >= 0 in the backward; the sign of zero is kept.
__device__ __forceinline__ float bg_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(fminf(fabsf(x), 10.f) * 2.885390081777927f);
    return copysignf(fminf((e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f), 1.f), x);
}
__device__ __forceinline__ float4 bg_scale4(float s, float4 a) {
    return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}
