// What does a dependency between two phases cost inside ONE launch (XCD-local barrier over the 32 workgroups that
// share an L2) against a kernel boundary?  Question behind VERDICT r2 item 5 (one persistent generator forward with a
// graph per XCD).  P phases; in every phase a workgroup reads the 16 KiB slab its left neighbour (same group) wrote in
// the previous phase, adds 1, writes its own slab.
//   (a) one persistent launch, 8 groups of 32 workgroups (group = blockIdx % 8), release / counter / acquire per phase
//   (b) P launches of the same phase body (256 workgroups), eager and as a hipGraph
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/xcd_barrier_lab.hip -o tools/micro/xcd_barrier_lab.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SLAB_F = 4096;          // floats per workgroup and phase (16 KiB)
constexpr int GROUPS = 8, PER_GROUP = 32, WGS = GROUPS * PER_GROUP;

__device__ __forceinline__ void phase_body(const float* src, float* dst) {
    const float4* s = reinterpret_cast<const float4*>(src);
    float4* d = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int i = 0; i < SLAB_F / 4 / 256; ++i) {
        float4 v = s[threadIdx.x + 256 * i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        d[threadIdx.x + 256 * i] = v;
    }
}

// slabs: two images [2][WGS][SLAB_F]; workgroup w of group g sits at index g * PER_GROUP + w
__global__ __launch_bounds__(256) void k_phase(const float* in, float* out) {
    const int g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;
    const int left = g * PER_GROUP + (w + PER_GROUP - 1) % PER_GROUP;
    phase_body(in + (size_t)left * SLAB_F, out + (size_t)(g * PER_GROUP + w) * SLAB_F);
}

__device__ __forceinline__ float4 load_past_l1(const float4* p) {
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// (a2) the same phases with a barrier that trusts the placement (all 32 workgroups of a group share one L2): stores
// are written through the L1 and acknowledged by that L2 (s_waitcnt vmcnt(0)), the arrive is an atomic performed in
// that L2, consumers read past their L1 (sc1 loads).  No L2 write-back, no L2 invalidate.  Not portable: correct
// only while group == XCD, which main() checks from HW_REG_XCC_ID.
template <int STORE_SC>
__global__ __launch_bounds__(256) void k_persistent_xcd(float* slabs, int* counters, int phases) {
    const int g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;
    const int me = g * PER_GROUP + w, left = g * PER_GROUP + (w + PER_GROUP - 1) % PER_GROUP;
    int* ctr = counters + g * 64;
    for (int p = 0; p < phases; ++p) {
        const float4* src = reinterpret_cast<const float4*>(slabs + ((size_t)(p & 1) * WGS + left) * SLAB_F);
        float4* dst = reinterpret_cast<float4*>(slabs + ((size_t)((p + 1) & 1) * WGS + me) * SLAB_F);
        float4 v[SLAB_F / 4 / 256];
#pragma unroll
        for (int i = 0; i < SLAB_F / 4 / 256; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src + threadIdx.x + 256 * i) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < SLAB_F / 4 / 256; ++i) {
            v[i].x += 1.f; v[i].y += 1.f; v[i].z += 1.f; v[i].w += 1.f;
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t = {v[i].x, v[i].y, v[i].z, v[i].w};
            f4v* q = reinterpret_cast<f4v*>(dst + threadIdx.x + 256 * i);
            if (STORE_SC == 0) *q = t;
            else if (STORE_SC == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(t) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(t) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (STORE_SC == 2) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int target = PER_GROUP * (p + 1);
            int spins = 0, seen;
            do {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(ctr) : "memory");
            } while (seen < target && ++spins < (1 << 22));
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_persistent(float* slabs, int* counters, int phases, int* xcc_seen) {
    const int g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;
    const int me = g * PER_GROUP + w, left = g * PER_GROUP + (w + PER_GROUP - 1) % PER_GROUP;
    int* ctr = counters + g * 64;                      // one counter per group, a line of its own
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_seen[blockIdx.x] = (int)(xcc & 0xf);
    }
    for (int p = 0; p < phases; ++p) {
        const float* src = slabs + ((size_t)(p & 1) * WGS + left) * SLAB_F;
        float* dst = slabs + ((size_t)((p + 1) & 1) * WGS + me) * SLAB_F;
        phase_body(src, dst);
        // ---- group barrier: every store drained, workgroup barrier, one lane releases and arrives, polls, acquires
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int target = PER_GROUP * (p + 1);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 64, reps = 20;
    float* slabs; int* counters; int* xcc;
    CK(hipMalloc(&slabs, sizeof(float) * 2 * WGS * SLAB_F));
    CK(hipMalloc(&counters, sizeof(int) * GROUPS * 64));
    CK(hipMalloc(&xcc, sizeof(int) * WGS));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h(2 * (size_t)WGS * SLAB_F);
    auto check = [&](const char* what) {
        CK(hipMemcpy(h.data(), slabs, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
        const float* img = h.data() + (size_t)(P & 1) * WGS * SLAB_F;
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)WGS * SLAB_F; ++i) bad += img[i] != (float)P;
        printf("   %s: every value == %d after %d phases: %s (%zu wrong)\n", what, P, P, bad ? "NO" : "yes", bad);
    };
    float ms;
    // (a) persistent
    double best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipMemsetAsync(counters, 0, sizeof(int) * GROUPS * 64, s));
        CK(hipEventRecord(e0, s));
        k_persistent<<<WGS, 256, 0, s>>>(slabs, counters, P, xcc);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(a) one launch, %d phases, XCD-group barrier per phase: %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("persistent");
    std::vector<int> hx(WGS);
    CK(hipMemcpy(hx.data(), xcc, sizeof(int) * WGS, hipMemcpyDeviceToHost));
    int mism = 0;
    for (int b = 0; b < WGS; ++b) mism += hx[b] != hx[b % GROUPS];
    printf("   workgroups whose XCC id differs from that of workgroup (blockIdx %% 8): %d of %d\n", mism, WGS);
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipMemsetAsync(counters, 0, sizeof(int) * GROUPS * 64, s));
        CK(hipEventRecord(e0, s));
        k_persistent_xcd<0><<<WGS, 256, 0, s>>>(slabs, counters, P);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(a2/0: plain stores, arrive at workgroup scope) one launch, %d phases, barrier inside the XCD's L2 (no write-back / invalidate): %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("persistent, XCD-scoped");
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipMemsetAsync(counters, 0, sizeof(int) * GROUPS * 64, s));
        CK(hipEventRecord(e0, s));
        k_persistent_xcd<1><<<WGS, 256, 0, s>>>(slabs, counters, P);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(a2/1: sc0 stores, arrive at workgroup scope) one launch, %d phases, barrier inside the XCD's L2 (no write-back / invalidate): %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("persistent, XCD-scoped");
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipMemsetAsync(counters, 0, sizeof(int) * GROUPS * 64, s));
        CK(hipEventRecord(e0, s));
        k_persistent_xcd<2><<<WGS, 256, 0, s>>>(slabs, counters, P);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(a2/2: sc1 stores, arrive at agent scope) one launch, %d phases, barrier inside the XCD's L2 (no write-back / invalidate): %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("persistent, XCD-scoped");
    // (b) launches, eager
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipEventRecord(e0, s));
        for (int p = 0; p < P; ++p) k_phase<<<WGS, 256, 0, s>>>(slabs + (size_t)(p & 1) * WGS * SLAB_F, slabs + (size_t)((p + 1) & 1) * WGS * SLAB_F);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(b) %d launches, eager: %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("launches");
    // (b') the same launches as a hipGraph
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int p = 0; p < P; ++p) k_phase<<<WGS, 256, 0, s>>>(slabs + (size_t)(p & 1) * WGS * SLAB_F, slabs + (size_t)((p + 1) & 1) * WGS * SLAB_F);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(slabs, 0, sizeof(float) * 2 * WGS * SLAB_F, s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("(b') the same %d launches as one hipGraph: %.1f us total, %.2f us per phase\n", P, best * 1e3, best * 1e3 / P);
    check("graph");
    return 0;
}
