// Dense fp32 products on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 fma chains).
//
//   C = epilogue( sum_s opA(A_s) * opB(B)[koff_s : koff_s + K_s, :] )
//
// A workgroup of 4 waves owns a BM x BN tile of C; the K loop walks 32-deep steps over the
// (up to four) A segments, i.e. torch.cat((A_0, A_1, ..), -1) is never materialised.  Operand
// tiles are staged global -> registers -> LDS (double buffered):
//   * K-contiguous operand  (A[M,K], nn.Linear weight W[N,K]):  LDS image [row][32+4], a lane reads
//     its 4 consecutive k with one ds_read_b128 (pitch 36 floats is conflict-free);
//   * K-strided operand     (A^T stored [K,M], B stored [K,N]): LDS image [k][rows+4], ds_read_b32.
// Both feed the same k order to the MFMA: step j of an 8-deep group multiplies k = 8g+j (lanes
// 0-31) and k = 8g+4+j (lanes 32-63) -- a permutation of k that both operands share.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BG_MAX_DEVICES = 64;


struct GemmArgs {
    const float* A[4]; int64_t lda[4]; int64_t Ks[4]; int n_seg;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    const float* bias; const float* row_scale; int act; float slope;
    const float* residual; int64_t ldr;
    const float* pre_add; int64_t ldp;
    int64_t M, N;
    int total_steps;      // sum over segments of ceil(K_s / BK)
    int steps_per_split;  // == total_steps when not split
    float* partial;       // split-K: [splits, M, N]
    // split-K (TN) only: B may be given as column segments  B = [B_0 | B_1 | ..]  (fused cat of the inputs)
    const float* Bseg[4]; int64_t ldbseg[4]; int64_t Nseg[4]; int n_bseg;
    // 1-D launch: gx column tiles x gy row tiles x splits; xcd != 0 remaps the linear block id so that
    // consecutive logical tiles (column tiles of one row panel, tiles of one K split) share an XCD's L2
    int gx, gy, xcd;
};

template <int ROWS, bool KC, bool VEC, int BK>
struct Stage {
    static constexpr int KC_PITCH = BK + 4;
    static constexpr int NF4 = ROWS * BK / 4 / 256;   // float4 per thread per tile
    float4 r[NF4];

    // tile origin: row0 (in the operand's non-reduction dim), k0 (inside the segment, length K)
    __device__ __forceinline__ void load(const float* base, int64_t ld, int64_t row0, int64_t n_rows,
                                         int64_t k0, int64_t K) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            int f = threadIdx.x + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                int row = f / (BK / 4), kv = f % (BK / 4);
                int64_t gr = row0 + row, gk = k0 + 4 * kv;
                if (gr < n_rows) {
                    const float* p = base + gr * ld + gk;
                    if (VEC) {
                        if (gk < K) v = bg_ld4(p);
                    } else {
                        if (gk + 0 < K) v.x = p[0];
                        if (gk + 1 < K) v.y = p[1];
                        if (gk + 2 < K) v.z = p[2];
                        if (gk + 3 < K) v.w = p[3];
                    }
                }
            } else {
                int k = f / (ROWS / 4), r4 = f % (ROWS / 4);
                int64_t gk = k0 + k, gr = row0 + 4 * r4;
                if (gk < K) {
                    const float* p = base + gk * ld + gr;
                    if (VEC) {
                        if (gr < n_rows) v = bg_ld4(p);
                    } else {
                        if (gr + 0 < n_rows) v.x = p[0];
                        if (gr + 1 < n_rows) v.y = p[1];
                        if (gr + 2 < n_rows) v.z = p[2];
                        if (gr + 3 < n_rows) v.w = p[3];
                    }
                }
            }
            r[i] = v;
        }
    }

    __device__ __forceinline__ void store(float* lds) const {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            int f = threadIdx.x + 256 * i;
            if (KC) {
                int row = f / (BK / 4), kv = f % (BK / 4);
                bg_st4(lds + row * KC_PITCH + 4 * kv, r[i]);
            } else {
                int k = f / (ROWS / 4), r4 = f % (ROWS / 4);
                bg_st4(lds + k * (ROWS + 4) + 4 * r4, r[i]);
            }
        }
    }
};

// Vector-path tile loader with incremental addressing: the per-thread pointers are set up once per
// A segment and advanced by a constant per K step, so the steady-state loop issues bare
// global_load_dwordx4 (no 64-bit index arithmetic, no exec-masked branches).  Rows / columns past
// the matrix edge are clamped to a valid row: they only feed accumulators that are never stored.
template <int ROWS, bool KC, int BK>
struct FastLoader {
    static constexpr int NF4 = ROWS * BK / 4 / 256;
    const float* p[NF4];
    const float* safe;
    int64_t stride;

    __device__ __forceinline__ int kofs(int i) const {
        const int f = threadIdx.x + 256 * i;
        return KC ? 4 * (f % (BK / 4)) : f / (ROWS / 4);
    }
    __device__ __forceinline__ void init(const float* base, int64_t ld, int64_t row0, int64_t n_rows, int64_t k0) {
        safe = base;
        stride = KC ? BK : BK * ld;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = threadIdx.x + 256 * i;
            if (KC) {
                int64_t gr = row0 + f / (BK / 4);
                gr = gr < n_rows ? gr : n_rows - 1;
                p[i] = base + gr * ld + k0 + 4 * (f % (BK / 4));
            } else {
                int64_t gr = row0 + 4 * (f % (ROWS / 4));
                gr = gr < n_rows ? gr : n_rows - 4;
                p[i] = base + (k0 + f / (ROWS / 4)) * ld + gr;
            }
        }
    }
    // k_rem = K - k0 of the tile being loaded (> 0)
    __device__ __forceinline__ void load(float4 (&r)[NF4], int64_t k_rem) {
        if (k_rem >= BK) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) r[i] = bg_ld4(p[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const bool in_k = kofs(i) < k_rem;
                float4 v = bg_ld4(in_k ? p[i] : safe);
                r[i] = in_k ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < NF4; ++i) p[i] += stride;
    }
};

// fragment of one 32-row tile for the 8-deep k group g: 4 values, value j feeds MFMA step j
template <int ROWS, bool KC, int BK>
__device__ __forceinline__ float4 read_frag(const float* lds, int tile_row0, int g, int r, int h) {
    constexpr int KC_PITCH = BK + 4;
    if (KC) {
        return bg_ld4(lds + (tile_row0 + r) * KC_PITCH + 8 * g + 4 * h);
    } else {
        const float* p = lds + (8 * g + 4 * h) * (ROWS + 4) + tile_row0 + r;
        return make_float4(p[0], p[ROWS + 4], p[2 * (ROWS + 4)], p[3 * (ROWS + 4)]);
    }
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool VEC, bool SPLIT>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    constexpr int KC_PITCH = BK + 4;
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_SZ = BM * KC_PITCH, B_SZ = BN * KC_PITCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];    // 2 * (A_SZ + B_SZ) floats

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // hardware deals workgroups round-robin over the 8 XCDs: give XCD x a contiguous range of logical tiles
    uint32_t lid = blockIdx.x;
    if (g.xcd) {
        const uint32_t q = gridDim.x >> 3, rem = gridDim.x & 7, x = lid & 7, slot = lid >> 3;
        lid = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + slot;
    }
    const int bx = lid % g.gx, by = (lid / g.gx) % g.gy, bz = lid / (g.gx * g.gy);
    const int64_t m0 = (int64_t)by * BM;
    int64_t n0 = (int64_t)bx * BN;
    // TN mode: locate the B segment this column tile belongs to (tiles never straddle segments)
    const float* Bp = g.B; int64_t ldb = g.ldb, n_lim = g.N, col_base = 0;
    if (SPLIT) {
        int t = bx, s = 0;
        while (s < g.n_bseg - 1) {
            int nt = (int)((g.Nseg[s] + BN - 1) / BN);
            if (t < nt) break;
            t -= nt; col_base += g.Nseg[s]; ++s;
        }
        Bp = g.Bseg[s]; ldb = g.ldbseg[s]; n_lim = g.Nseg[s]; n0 = (int64_t)t * BN;
    }

    int step_begin = 0, step_end = g.total_steps;
    if (SPLIT) {
        step_begin = bz * g.steps_per_split;
        step_end = min(step_begin + g.steps_per_split, g.total_steps);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    Stage<BM, A_KC, VEC, BK> sa;
    Stage<BN, B_KC, VEC, BK> sb;
    FastLoader<BM, A_KC, BK> fa;
    FastLoader<BN, B_KC, BK> fb;
    int it_seg = 0; int64_t it_k0 = (int64_t)step_begin * BK, it_koff = 0;   // position of the NEXT tile to fetch
    if (VEC && step_begin < step_end) {
        fa.init(g.A[0], g.lda[0], m0, g.M, it_k0);
        fb.init(Bp, ldb, n0, n_lim, it_k0);
    }

    // step -> (segment, k offset inside the segment, k offset inside the concatenated K)
    auto locate = [&](int step, int& seg, int64_t& k0, int64_t& koff) {
        seg = 0; koff = 0;
        int s = step;
        while (seg < g.n_seg - 1) {
            int n = (int)((g.Ks[seg] + BK - 1) / BK);
            if (s < n) break;
            s -= n; koff += g.Ks[seg]; ++seg;
        }
        k0 = (int64_t)s * BK;
    };
    auto fetch = [&](int step) {
        if (VEC) {
            const int64_t k_rem = g.Ks[it_seg] - it_k0;
            fa.load(sa.r, k_rem);
            fb.load(sb.r, k_rem);
            it_k0 += BK;
            if (it_k0 >= g.Ks[it_seg] && it_seg + 1 < g.n_seg) {       // next tile starts a new A segment
                it_koff += g.Ks[it_seg]; ++it_seg; it_k0 = 0;
                fa.init(g.A[it_seg], g.lda[it_seg], m0, g.M, 0);
                fb.init(B_KC ? Bp + it_koff : Bp + it_koff * ldb, ldb, n0, n_lim, 0);
            }
            return;
        }
        int seg; int64_t k0, koff;
        locate(step, seg, k0, koff);
        sa.load(g.A[seg], g.lda[seg], m0, g.M, k0, g.Ks[seg]);
        // B rows of this step: concatenated k in [koff + k0, koff + K_seg)
        const float* bb = B_KC ? Bp + koff : Bp + koff * ldb;
        sb.load(bb, ldb, n0, n_lim, k0, g.Ks[seg]);
    };

    if (step_begin < step_end) {
        fetch(step_begin);
        sa.store(smem);
        sb.store(smem + A_SZ);
    }
    __syncthreads();

    for (int step = step_begin; step < step_end; ++step) {
        const int cur = (step - step_begin) & 1;
        const float* As = smem + cur * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
        const bool more = step + 1 < step_end;
        if (more) fetch(step + 1);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_KC, BK>(As, (wm * TM + i) * 32, kg, r, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_KC, BK>(Bs, (wn * TN + j) * 32, kg, r, h);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            float* An = smem + (cur ^ 1) * (A_SZ + B_SZ);
            sa.store(An);
            sb.store(An + A_SZ);
        }
        __syncthreads();
    }

    // epilogue: lane holds rows (q&3) + 8*(q>>2) + 4*h, column r of each 32x32 tile
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int64_t col = n0 + (wn * TN + j) * 32 + r;
            if (col >= n_lim) continue;
            col += col_base;
            const float bv = (!SPLIT && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t row = m0 + (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (row >= g.M) continue;
                float v = acc[i][j][q];
                if (SPLIT) {
                    g.partial[((int64_t)bz * g.M + row) * g.N + col] = v;
                } else {
                    v += bv;
                    if (g.row_scale) v *= g.row_scale[row];
                    if (g.pre_add) v += g.pre_add[row * g.ldp + col];
                    if (g.act == BG_ACT_LEAKY) v = v > 0.f ? v : v * g.slope;
                    if (g.residual) v += g.residual[row * g.ldr + col];
                    g.C[row * g.ldc + col] = v;
                }
            }
        }
}

__global__ void k_splitk_reduce(const float* __restrict__ partial, int splits, int64_t M, int64_t N,
                                float* __restrict__ C, int64_t ldc) {
    // one float4 of C per thread (M*N % 4 == 0 checked by the launcher), splits added in order
    const int64_t MN = M * N;
    int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t >= MN) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
        float4 a = bg_ld4(partial + (int64_t)z * MN + t), b = bg_ld4(partial + (int64_t)(z + 1) * MN + t);
        float4 c = bg_ld4(partial + (int64_t)(z + 2) * MN + t), d = bg_ld4(partial + (int64_t)(z + 3) * MN + t);
        acc = bg_add4(acc, a); acc = bg_add4(acc, b); acc = bg_add4(acc, c); acc = bg_add4(acc, d);
    }
    for (; z < splits; ++z) acc = bg_add4(acc, bg_ld4(partial + (int64_t)z * MN + t));
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) C[((t + j) / N) * ldc + ((t + j) % N)] = v[j];
}

__global__ void k_splitk_reduce_scalar(const float* __restrict__ partial, int splits, int64_t M, int64_t N,
                                       float* __restrict__ C, int64_t ldc) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    float acc = 0.f;
    for (int z = 0; z < splits; ++z) acc += partial[(int64_t)z * M * N + t];
    C[(t / N) * ldc + (t % N)] = acc;
}

// Column sums, phase 1: workgroup (bx, by) owns 64 columns x rows [by*RPB, (by+1)*RPB); its 4 waves
// take every 4th row, 8 loads in flight per lane; the waves' sums are combined through LDS.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int64_t ldx, int64_t M, int64_t N,
                                                        int rows_per_block, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, M);
    float acc = 0.f;
    if (n < N) {
        int64_t m = r0 + wave;
        for (; m + 28 < r1; m += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = x[(m + 4 * j) * ldx + n];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; m < r1; m += 4) acc += x[m * ldx + n];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && n < N) partial[(int64_t)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// phase 2: one wave per column adds the row-block partials (fixed shuffle tree: deterministic)
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int n_blocks, int64_t N,
                                                      float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc = 0.f;
    for (int b = lane; b < n_blocks; b += 64) acc += partial[(int64_t)b * N + n];
    acc = bg_wave_sum(acc);
    if (lane == 0) out[n] = acc;
}

// tuning hook: BG_GEMM_XCD=0 keeps the hardware's round-robin tile order
int xcd_mode() {
    static int cached = -1;
    if (cached < 0) {
        const char* e = getenv("BG_GEMM_XCD");
        cached = e ? (atoi(e) != 0) : 1;
    }
    return cached;
}

template <int BM, int BN, int BK, bool SPLIT>
int launch_gemm(int transA, int transB, bool vec, GemmArgs g, int splits, hipStream_t s, int64_t n_tiles = 0) {
    g.gx = (int)(n_tiles > 0 ? n_tiles : (g.N + BN - 1) / BN);
    g.gy = (int)((g.M + BM - 1) / BM);
    g.xcd = xcd_mode();
    const int64_t blocks = (int64_t)g.gx * g.gy * splits;
    if (blocks > 0x7fffffff) return BG_ERR_ARG;
    dim3 grid((unsigned)blocks);
    constexpr size_t lds_bytes = 2 * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    const bool akc = !transA, bkc = transB != 0;
#define BG_GEMM_CASE(AKC, BKC, VEC)                                                                  \
    if (akc == AKC && bkc == BKC && vec == VEC) {                                                    \
        auto kern = k_gemm<BM, BN, BK, AKC, BKC, VEC, SPLIT>;                                        \
        if (lds_bytes > 65536) {   /* above the static limit: opt in once per device */              \
            static bool raised[BG_MAX_DEVICES] = {};                                                 \
            int dev = 0;                                                                             \
            BG_HIP(hipGetDevice(&dev));                                                        \
            if (dev < 0 || dev >= BG_MAX_DEVICES) return BG_ERR_ARG;                                 \
            if (!raised[dev]) {                                                                      \
                BG_HIP(hipFuncSetAttribute((const void*)kern,                                  \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
                raised[dev] = true;                                                                  \
            }                                                                                        \
        }                                                                                            \
        kern<<<grid, 256, lds_bytes, s>>>(g);                                                        \
        BG_LAUNCH_CHECK();                                                                           \
        return BG_OK;                                                                                \
    }
    BG_GEMM_CASE(true, true, true)
    BG_GEMM_CASE(true, false, true)
    BG_GEMM_CASE(false, true, true)
    BG_GEMM_CASE(false, false, true)
    BG_GEMM_CASE(true, true, false)
    BG_GEMM_CASE(true, false, false)
    BG_GEMM_CASE(false, true, false)
    BG_GEMM_CASE(false, false, false)
#undef BG_GEMM_CASE
    return BG_ERR_ARG;
}

bool operand_vec_ok(const float* p, int64_t ld, bool kc, int64_t K, int64_t rows) {
    if (!bg_aligned16(p) || ld % 4 != 0) return false;
    return kc ? (K % 4 == 0) : (rows % 4 == 0);
}

// tuning hook: BG_GEMM_TILE = 1..4 forces 64x64 / 128x64 / 64x128 / 128x128 (unset: heuristic)
int tile_override() {
    static int cached = -1;
    if (cached < 0) {
        const char* e = getenv("BG_GEMM_TILE");
        cached = e ? atoi(e) : 0;
        if (cached < 0 || cached > 4) cached = 0;
    }
    return cached;
}

// tuning hook: BG_GEMM_BK = 32 / 64 forces the K depth of a 64x64-tile step (unset: 64)
int bk_override() {
    static int cached = -1;
    if (cached < 0) {
        const char* e = getenv("BG_GEMM_BK");
        cached = e ? atoi(e) : 0;
        if (cached != 32 && cached != 64) cached = 0;
    }
    return cached;
}

}  // namespace

extern "C" {

int bg_gemm_f32(int transA, int transB, int64_t M, int64_t N, int n_seg,
                const float* const* A, const int64_t* lda, const int64_t* Ks,
                const float* B, int64_t ldb, float* C, int64_t ldc,
                const float* bias, const float* row_scale, int act, float slope,
                const float* residual, int64_t ldr, const float* pre_add, int64_t ldp, bgStream stream) {
    if (M < 0 || N < 0 || n_seg < 1 || n_seg > 4 || !A || !lda || !Ks) return BG_ERR_ARG;
    if (M == 0 || N == 0) return BG_OK;
    if (!B || !C || ldc < N || (residual && ldr < N) || (pre_add && ldp < N)) return BG_ERR_ARG;
    GemmArgs g{};
    g.n_seg = n_seg;
    g.pre_add = pre_add; g.ldp = ldp;
    int tile = tile_override();
    if (tile == 0) {
        const int64_t big_tiles = ((M + 127) / 128) * ((N + 127) / 128);
        tile = big_tiles >= 512 ? 3 : 0;
    } else {
        tile -= 1;
    }
    // 64x64 tiles run 1-3 workgroups per CU, so a K step is bounded by the latency of its global
    // loads: 64-deep steps halve the number of load -> LDS -> barrier round trips
    int bk = tile == 0 ? 64 : 32;
    if (tile == 0 && bk_override()) bk = bk_override();
    int64_t Ktot = 0;
    bool vec = true;
    int steps = 0;
    for (int s = 0; s < n_seg; ++s) {
        if (!A[s] || Ks[s] < 0) return BG_ERR_ARG;
        if (lda[s] < (transA ? M : Ks[s])) return BG_ERR_ARG;
        g.A[s] = A[s]; g.lda[s] = lda[s]; g.Ks[s] = Ks[s];
        vec = vec && operand_vec_ok(A[s], lda[s], !transA, Ks[s], M);
        // a segment's rows of B start at koff: keep them 16-byte aligned for the vector path
        if (transB && (Ktot % 4 != 0)) vec = false;
        Ktot += Ks[s];
        steps += (int)((Ks[s] + bk - 1) / bk);
    }
    if (ldb < (transB ? Ktot : N)) return BG_ERR_ARG;
    vec = vec && operand_vec_ok(B, ldb, transB != 0, 4, N);
    for (int s = 0; s < n_seg && transB; ++s) if (Ks[s] % 4 != 0) vec = false;
    g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.row_scale = row_scale; g.act = act; g.slope = slope;
    g.residual = residual; g.ldr = ldr; g.M = M; g.N = N;
    g.total_steps = steps; g.steps_per_split = steps; g.partial = nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 3: return launch_gemm<128, 128, 32, false>(transA, transB, vec, g, 1, st);
        case 2: return launch_gemm<64, 128, 32, false>(transA, transB, vec, g, 1, st);
        case 1: return launch_gemm<128, 64, 32, false>(transA, transB, vec, g, 1, st);
        default:
            if (bk == 64) return launch_gemm<64, 64, 64, false>(transA, transB, vec, g, 1, st);
            return launch_gemm<64, 64, 32, false>(transA, transB, vec, g, 1, st);
    }
}

int bg_gemm_tn_splitk_f32(int64_t M, int64_t R, const float* A, int64_t lda,
                          int n_seg, const float* const* B, const int64_t* ldb, const int64_t* Ns,
                          float* C, int64_t ldc, float* partial, int32_t splits, bgStream stream) {
    if (M < 0 || R < 0 || splits < 1 || n_seg < 1 || n_seg > 4 || !B || !ldb || !Ns) return BG_ERR_ARG;
    int64_t N = 0;
    for (int s = 0; s < n_seg; ++s) {
        if (Ns[s] < 0 || (Ns[s] > 0 && (!B[s] || ldb[s] < Ns[s]))) return BG_ERR_ARG;
        N += Ns[s];
    }
    if (M == 0 || N == 0) return BG_OK;
    if (!A || !C || !partial || lda < M || ldc < N) return BG_ERR_ARG;
    GemmArgs g{};
    g.n_seg = 1; g.A[0] = A; g.lda[0] = lda; g.Ks[0] = R;
    g.B = B[0]; g.ldb = ldb[0]; g.C = C; g.ldc = ldc; g.M = M; g.N = N;
    g.n_bseg = n_seg;
    int64_t n_tiles = 0;
    bool vec = operand_vec_ok(A, lda, false, R, M);
    for (int s = 0; s < n_seg; ++s) {
        g.Bseg[s] = B[s]; g.ldbseg[s] = ldb[s]; g.Nseg[s] = Ns[s];
        n_tiles += (Ns[s] + 63) / 64;
        if (Ns[s] > 0) vec = vec && operand_vec_ok(B[s], ldb[s], false, R, Ns[s]);
    }
    constexpr int BK = 32;
    g.total_steps = (int)((R + BK - 1) / BK);
    g.steps_per_split = (g.total_steps + splits - 1) / splits;
    if (g.steps_per_split < 1) g.steps_per_split = 1;
    g.partial = partial;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_gemm<64, 64, 32, true>(1, 0, vec, g, splits, st, n_tiles);
    if (rc != BG_OK) return rc;
    if ((M * N) % 4 == 0 && bg_aligned16(partial))
        k_splitk_reduce<<<bg_grid_for((M * N) / 4, 256, 0x7fffffff), 256, 0, st>>>(partial, splits, M, N, C, ldc);
    else
        k_splitk_reduce_scalar<<<bg_grid_for(M * N, 256, 0x7fffffff), 256, 0, st>>>(partial, splits, M, N, C, ldc);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_colsum_f32(const float* x, int64_t ldx, int64_t M, int64_t N, float* partial, int32_t n_blocks,
                  float* out, bgStream stream) {
    if (M < 0 || N < 0 || n_blocks < 1) return BG_ERR_ARG;
    if (N == 0) return BG_OK;
    if (!out || !partial || (M > 0 && (!x || ldx < N))) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int rows_per_block = (int)((M + n_blocks - 1) / n_blocks);
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)n_blocks);
    k_colsum_partial<<<grid, 256, 0, s>>>(x, ldx, M, N, rows_per_block > 0 ? rows_per_block : 1, partial);
    BG_LAUNCH_CHECK();
    k_colsum_final<<<bg_grid_for(N, 4), 256, 0, s>>>(partial, n_blocks, N, out);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
