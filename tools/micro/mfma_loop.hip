// Inner-loop microbenchmark (tuning aid, not product code): which structure of the
// LDS-read + fp32 MFMA + barrier loop keeps the matrix pipe busy at the train-step GEMM shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PITCH = 36, TILE = 64 * PITCH;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE bits: 1 = LDS reads, 2 = barrier per step, 4 = two accumulator chains, 8 = four chains (k-interleaved),
// 16 = LDS writes of a register tile per step (no global loads), 32 = s_setprio around MFMA
template <int MODE>
__global__ __launch_bounds__(256) void k_loop(float* out, int steps) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 4 * TILE; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    constexpr int NC = (MODE & 8) ? 4 : ((MODE & 4) ? 2 : 1);
    f32x16 acc[NC];
    for (int c = 0; c < NC; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    float4 st0 = make_float4(1.f, 2.f, 3.f, 4.f), st1 = st0;
    float4 a0 = make_float4(0.1f * lane, 0.2f, 0.3f, 0.4f), b0 = make_float4(0.5f, 0.6f, 0.7f, 0.8f);
    for (int step = 0; step < steps; ++step) {
        const float* As = smem + (step & 1) * 2 * TILE;
        const float* Bs = As + TILE;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 af = a0, bf = b0;
            if (MODE & 1) {
                af = *(const float4*)(As + (wm * 32 + r) * PITCH + 8 * kg + 4 * h);
                bf = *(const float4*)(Bs + (wn * 32 + r) * PITCH + 8 * kg + 4 * h);
            }
            if (MODE & 32) __builtin_amdgcn_s_setprio(1);
            acc[(0) % NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[(0) % NC], 0, 0, 0);
            acc[(1) % NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[(1) % NC], 0, 0, 0);
            acc[(2) % NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[(2) % NC], 0, 0, 0);
            acc[(3) % NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[(3) % NC], 0, 0, 0);
            if (MODE & 32) __builtin_amdgcn_s_setprio(0);
        }
        if (MODE & 16) {
            float* An = smem + ((step & 1) ^ 1) * 2 * TILE;
            const int f = threadIdx.x, row = f / 8, kv = f % 8;
            *(float4*)(An + row * PITCH + 4 * kv) = st0;
            *(float4*)(An + (row + 32) * PITCH + 4 * kv) = st1;
            *(float4*)(An + TILE + row * PITCH + 4 * kv) = st0;
            *(float4*)(An + TILE + (row + 32) * PITCH + 4 * kv) = st1;
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < NC; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int blocks, int steps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) k_loop<MODE><<<blocks, 256>>>(out, steps);
    CK(hipEventRecord(a));
    const int it = 50;
    for (int i = 0; i < it; ++i) k_loop<MODE><<<blocks, 256>>>(out, steps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1e3 / it, fl = 2.0 * blocks * 64.0 * 64 * 32 * steps;
    printf("%-34s blocks=%5d steps=%3d  %7.1f us  %6.1f TF\n", name, blocks, steps, us, fl / us / 1e6);
}

int main(int argc, char** argv) {
    float* out; CK(hipMalloc(&out, 8192 * 256 * 4));
    const int cfg[][2] = {{720, 17}, {768, 17}, {1024, 17}, {720, 64}, {2048, 64}};
    for (auto& c : cfg) {
        run<0>("mfma only", out, c[0], c[1]);
        run<4>("mfma only, 2 chains", out, c[0], c[1]);
        run<8>("mfma only, 4 chains", out, c[0], c[1]);
        run<1>("lds reads", out, c[0], c[1]);
        run<3>("lds reads + barrier", out, c[0], c[1]);
        run<7>("lds reads + barrier, 2 chains", out, c[0], c[1]);
        run<11>("lds reads + barrier, 4 chains", out, c[0], c[1]);
        run<19>("lds rd + wr + barrier", out, c[0], c[1]);
        run<27>("lds rd + wr + barrier, 4 chains", out, c[0], c[1]);
        run<51>("lds rd + wr + barrier, setprio", out, c[0], c[1]);
        printf("\n");
    }
    return 0;
}
