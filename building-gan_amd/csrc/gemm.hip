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
#include <type_traits>

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
    const int32_t* pre_rows;   // optional: pre_add row of output row i is pre_rows[i] (a fused gather)
    // k_gemm_panel only (bg_gemm_nbr_f32): the gathered addend row s is formed on load as
    //   (sum_{k in nbr_rowptr[s] .. nbr_rowptr[s+1]} nbr[nbr_col[k], :]) / nbr_count[s] + pre_add[s, :]
    // -- the arithmetic of bg_spmm_add_f32 (slot order from +0, IEEE division, then the addend), so the separate launch
    // that used to write that matrix is not needed
    const float* nbr; int64_t ldn; const int32_t* nbr_rowptr; const int32_t* nbr_col; const float* nbr_count;
    const float* acc0; int64_t ld0;   // optional: the accumulators start from acc0[i, j] instead of zero (k_gemm only)
    float* C2; int64_t ldc2;          // optional: the value BEFORE the residual is added is also written here
    int64_t M, N;
    int total_steps;      // sum over segments of ceil(K_s / BK)
    int steps_per_split;  // == total_steps when not split
    float* partial;       // split-K: [splits, M, N]
    // split-K (TN) only: B may be given as column segments  B = [B_0 | B_1 | ..]  (fused cat of the inputs)
    const float* Bseg[4]; int64_t ldbseg[4]; int64_t Nseg[4]; int n_bseg;
    // 1-D launch: gx column tiles x gy row tiles x splits; xcd != 0 remaps the linear block id so that
    // consecutive logical tiles (column tiles of one row panel, tiles of one K split) share an XCD's L2
    int gx, gy, xcd;
    // k_gemm_skinny only: segment 1 is not read as stored but as the Chebyshev shift of the table A[1] (row blocks of
    // cheb_R rows, cheb_K blocks; cheb_T: the adjoint map) -- the arithmetic of k_cheb_shift (critic.hip), applied on load
    int cheb_R, cheb_K, cheb_T;
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

// Epilogue of one 32x32 accumulator tile: lane holds column `col`, rows row0 + (q&3) + 8*(q>>2) (+4h folded
// into row0).  All addend loads of the 16 rows are issued before the first use (rows past M are clamped for
// the loads and skipped for the store), so the tile pays one memory round trip per addend, not sixteen.
__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16& acc, int64_t row0, int64_t col, float bv) {
    float rs[16], pa[16], rd[16];
    int64_t rows[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t row = row0 + (q & 3) + 8 * (q >> 2);
        rows[q] = row < g.M ? row : g.M - 1;
    }
    if (g.row_scale) {
#pragma unroll
        for (int q = 0; q < 16; ++q) rs[q] = g.row_scale[rows[q]];
    }
    if (g.pre_add) {
        if (g.pre_rows) {
            int32_t pr[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) pr[q] = g.pre_rows[rows[q]];
#pragma unroll
            for (int q = 0; q < 16; ++q) pa[q] = g.pre_add[(int64_t)pr[q] * g.ldp + col];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) pa[q] = g.pre_add[rows[q] * g.ldp + col];
        }
    }
    if (g.residual) {
#pragma unroll
        for (int q = 0; q < 16; ++q) rd[q] = g.residual[rows[q] * g.ldr + col];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float v = acc[q] + bv;
        if (g.row_scale) v *= rs[q];
        if (g.pre_add) v += pa[q];
        if (g.act == BG_ACT_LEAKY) v = v > 0.f ? v : v * g.slope;
        const bool in_range = row0 + (q & 3) + 8 * (q >> 2) < g.M;
        if (g.C2 && in_range) g.C2[rows[q] * g.ldc2 + col] = v;
        if (g.residual) v += rd[q];
        if (in_range) g.C[rows[q] * g.ldc + col] = v;
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
    if (!SPLIT && g.acc0) {
        // C = acc0 + A B: the addend is loaded straight into the accumulator registers (no extra registers, and
        // the loads travel with the first operand tile's) -- a K-independent term costs no K step this way
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int64_t col = n0 + (wn * TN + j) * 32 + r;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int64_t row = m0 + (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (row < g.M && col < g.N) acc[i][j][q] = g.acc0[row * g.ld0 + col];
                }
            }
    }

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
            if (SPLIT) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int64_t row = m0 + (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (row < g.M) g.partial[((int64_t)bz * g.M + row) * g.N + col] = acc[i][j][q];
                }
            } else {
                store_tile(g, acc[i][j], m0 + (wm * TM + i) * 32 + 4 * h, col, g.bias ? g.bias[col] : 0.f);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Row-panel GEMM for a narrow output (N = 128) with K-contiguous operands (A [M,K], nn.Linear weight W [128,K]):
// one workgroup owns 16*TMT rows x all 128 columns, so the whole product is <= 256 panels -- ONE per CU, no
// second round of workgroups (the 64x64-tile kernel runs M = 11,520 as 360 tiles = 1.4 rounds on 256 CUs and is
// bound by that quantisation).  v_mfma_f32_16x16x4_f32: a wave owns 32 columns (2 column tiles) x all
// TMT row tiles = 2*TMT independent accumulators (the 40-cycle dependent latency is covered with 6-8 in
// flight at a 32-cycle issue interval).  K runs in 64-deep steps through a double-buffered LDS image
// [row][64+4] (pitch 68 = 4 mod 64 dwords: the 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank
// slots); one b128 read gives a lane the operands of four MFMAs: MFMA j of a 16-deep group multiplies
// k = 16g + 4*(lane>>4) + j on both operands (a permutation of k the operands share).  Global -> registers
// -> LDS staging of step s+1 overlaps the MFMAs of step s.  Same epilogue terms as k_gemm.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BG_PANEL_PAD = 8;

template <int TMT, int NJ, bool NBR = false>
__global__ __launch_bounds__(256) void k_gemm_panel(GemmArgs g) {
    // PITCH = BK + 8: with the 16x16x4 fragment layout (lane = row r + 16 kq, 16-byte chunk kq of a 16-float group) a
    // ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, .. (MI355X_MICROARCH.md, LDS): the
    // 16-byte slot of lane (r, kq) is (PITCH/4 r + kq) mod 16 = (2 r + kq) mod 16, distinct inside every group.  BK + 4
    // (slot r + kq: right for k_gemm's 32-row layout) put lanes 12 and 27 on one slot: SQ_LDS_BANK_CONFLICT = 34 % of the
    // LDS-active cycles of this kernel (profiles/r02_pmc_sq_panel.txt)
    constexpr int BK = 64, PITCH = BK + BG_PANEL_PAD, PM = 16 * TMT, BN = 64 * NJ;      // a wave owns 16 NJ columns
    constexpr int A_SZ = PM * PITCH, B_SZ = BN * PITCH;
    constexpr int NA = PM * BK / 4 / 256, NB = BN * BK / 4 / 256;       // float4 per thread per step: TMT, 4 NJ
    constexpr int CP = BN + 4;                                           // pitch of the result image
    constexpr int NC = PM * BN / 4 / 256;                                // result float4 per thread: TMT NJ
    constexpr int RL = BN / 4;                                           // lanes per result row
    extern __shared__ __attribute__((aligned(16))) float smem[];         // 2 * (A_SZ + B_SZ) floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * PM;
    const int n0 = blockIdx.y * BN;

    f32x4 acc[TMT][NJ];
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;

    // staging pointers: thread f (+256 i) moves the 16-byte chunk kv = f % 16 of row f / 16
    const float* pa[NA]; const float* pb[NB];
    float4 ra[NA], rb[NB];
    int seg = 0; int64_t k0 = 0, koff = 0;
    auto point = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = threadIdx.x + 256 * i;
            int64_t gr = m0 + f / 16;
            gr = gr < g.M ? gr : g.M - 1;                                 // clamped rows feed results never stored
            pa[i] = g.A[seg] + gr * g.lda[seg] + 4 * (f % 16);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = threadIdx.x + 256 * i;
            pb[i] = g.B + (int64_t)(n0 + f / 16) * g.ldb + koff + 4 * (f % 16);
        }
    };
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) { ra[i] = bg_ld4(pa[i]); pa[i] += BK; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { rb[i] = bg_ld4(pb[i]); pb[i] += BK; }
        k0 += BK;
        if (k0 >= g.Ks[seg] && seg + 1 < g.n_seg) { koff += g.Ks[seg]; ++seg; k0 = 0; point(); }
    };
    auto stash = [&](float* As) {
        float* Bs = As + A_SZ;
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int f = threadIdx.x + 256 * i; bg_st4(As + (f / 16) * PITCH + 4 * (f % 16), ra[i]); }
#pragma unroll
        for (int i = 0; i < NB; ++i) { const int f = threadIdx.x + 256 * i; bg_st4(Bs + (f / 16) * PITCH + 4 * (f % 16), rb[i]); }
    };

    // epilogue operands: thread f (+256 i) finishes the 16-byte chunk c4 = f % 32 of panel row f / 32 (a row = 32
    // lanes = one 512-byte line: every load and store of the epilogue is a full-row float4 access).  They are
    // requested while the LAST K step multiplies, so their latency hides behind its MFMAs.
    const int c4 = threadIdx.x % RL;
    float4 e_acc[NC], e_pre[NC], e_res[NC];
    float e_rs[NC];
    int64_t e_row[NC];
    auto prefetch_epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int64_t row = m0 + (threadIdx.x + 256 * i) / RL;
            e_row[i] = row < g.M ? row : g.M - 1;
        }
        if (g.pre_add && !NBR) {
            int64_t pr[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) pr[i] = g.pre_rows ? (int64_t)g.pre_rows[e_row[i]] : e_row[i];
#pragma unroll
            for (int i = 0; i < NC; ++i) e_pre[i] = bg_ld4(g.pre_add + pr[i] * g.ldp + n0 + 4 * c4);
        }
        if (g.acc0) {
#pragma unroll
            for (int i = 0; i < NC; ++i) e_acc[i] = bg_ld4(g.acc0 + e_row[i] * g.ld0 + n0 + 4 * c4);
        }
        if (g.residual) {
#pragma unroll
            for (int i = 0; i < NC; ++i) e_res[i] = bg_ld4(g.residual + e_row[i] * g.ldr + n0 + 4 * c4);
        }
        if (g.row_scale) {
#pragma unroll
            for (int i = 0; i < NC; ++i) e_rs[i] = g.row_scale[e_row[i]];
        }
    };

    // bg_gemm_nbr_f32: the addend row formed from its neighbour rows.  Four dependent loads (row index -> row pointers
    // -> column indices -> rows): before the first operand request, after the first LDS write, then at K steps 1 and 2,
    // so that none is waited for where it is issued (each has a whole step to arrive); what the K loop is too short for runs after it.  Up to NBR_D neighbours travel through registers, longer rows continue in a loop.
    // The rows a thread finishes are 16 apart and nearly always share their addend row s (voxels of a story are
    // contiguous): then ONE neighbour sum serves all of them, up to NBR_D neighbours in registers (a building's
    // stories: 8-11).  Threads whose rows differ, and longer neighbour lists, take the loop in nbr_finish.
    constexpr int NBR_D = 12;
    int32_t n_s[NC], n_b = 0, n_e = 0, n_c[NBR_D];
    float n_cnt = 1.f;
    float4 n_x[NBR_D];
    bool n_same = true;
    auto nbr_stage = [&](const int st) {
        if (!NBR) return;
        if (st == 0) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int64_t row = m0 + (threadIdx.x + 256 * i) / RL;
                const int64_t rc = row < g.M ? row : g.M - 1;
                n_s[i] = g.pre_rows ? g.pre_rows[rc] : (int32_t)rc;
            }
        } else if (st == 1) {
#pragma unroll
            for (int i = 1; i < NC; ++i) n_same = n_same && n_s[i] == n_s[0];
            n_b = g.nbr_rowptr[n_s[0]]; n_e = g.nbr_rowptr[n_s[0] + 1]; n_cnt = g.nbr_count[n_s[0]];
#pragma unroll
            for (int i = 0; i < NC; ++i)
                e_pre[i] = bg_ld4(g.pre_add + (int64_t)n_s[i] * g.ldp + n0 + 4 * c4);      // the row's own part
        } else if (st == 2) {
#pragma unroll
            for (int j = 0; j < NBR_D; ++j) n_c[j] = n_b + j < n_e ? g.nbr_col[n_b + j] : -1;
        } else if (st == 3) {
#pragma unroll
            for (int j = 0; j < NBR_D; ++j) {
                n_x[j] = make_float4(0.f, 0.f, 0.f, 0.f);          // + 0 leaves a sum that started at +0 unchanged
                if (n_c[j] >= 0) n_x[j] = bg_ld4(g.nbr + (int64_t)n_c[j] * g.ldn + n0 + 4 * c4);
            }
        }
    };
    auto nbr_finish = [&]() {
        if (!NBR) return;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NBR_D; ++j) a = bg_add4(a, n_x[j]);
        for (int32_t k = n_b + NBR_D; k < n_e; ++k)
            a = bg_add4(a, bg_ld4(g.nbr + (int64_t)g.nbr_col[k] * g.ldn + n0 + 4 * c4));
        a.x = a.x / n_cnt; a.y = a.y / n_cnt; a.z = a.z / n_cnt; a.w = a.w / n_cnt;
        e_pre[0] = bg_add4(a, e_pre[0]);
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            float4 t = a;
            if (!n_same && n_s[i] != n_s[0]) {                  // another addend row: its own sum, by the loop
                const int32_t b2 = g.nbr_rowptr[n_s[i]], e2 = g.nbr_rowptr[n_s[i] + 1];
                const float c2 = g.nbr_count[n_s[i]];
                t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int32_t k = b2; k < e2; ++k)
                    t = bg_add4(t, bg_ld4(g.nbr + (int64_t)g.nbr_col[k] * g.ldn + n0 + 4 * c4));
                t.x = t.x / c2; t.y = t.y / c2; t.z = t.z / c2; t.w = t.w / c2;
            }
            e_pre[i] = bg_add4(t, e_pre[i]);
        }
    };

    nbr_stage(0);               // ahead of the operands: it has arrived when they have (loads return in order)
    point();
    fetch();
    stash(smem);
    nbr_stage(1);
    __syncthreads();
    const int steps = g.total_steps;
    // the last step is peeled (MORE = false): the epilogue operands it requests are then not live across the loop
    auto one_step = [&](const int step, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const float* As = smem + (step & 1) * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
        if (more) fetch(); else prefetch_epilogue();
        if (step == 1 || step == 2) nbr_stage(step + 1);      // a whole K step after the level before it
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            float4 af[TMT], bf[NJ];
#pragma unroll
            for (int i = 0; i < TMT; ++i) af[i] = bg_ld4(As + (16 * i + r) * PITCH + 16 * kg + 4 * kq);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = bg_ld4(Bs + (16 * NJ * wave + 16 * j + r) * PITCH + 16 * kg + 4 * kq);
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            // half way through the step's MFMAs the next step's operands have arrived: their LDS writes are issued
            // here, between MFMAs that keep the matrix pipe busy, instead of after the last one
            if (kg == BK / 32 - 1 && more) stash(smem + ((step + 1) & 1) * (A_SZ + B_SZ));
        }
        __syncthreads();
    };
    for (int step = 0; step + 1 < steps; ++step) one_step(step, std::true_type{});
    one_step(steps - 1, std::false_type{});

    for (int st = steps >= 2 ? steps + 1 : 2; st < 4; ++st) nbr_stage(st);
    nbr_finish();

    // the panel's result through LDS (lane holds column (lane & 15), rows 4*(lane>>4) + q of each 16x16 tile), then
    // row-major float4 work: v = act(rs * (acc0 + sum + bias) + pre_add[pr(row)]) (+ residual)
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) Cs[(16 * i + 4 * kq + q) * CP + 16 * NJ * wave + 16 * j + r] = acc[i][j][q];
    __syncthreads();
    const float4 bv = g.bias ? bg_ld4(g.bias + n0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int prow = (threadIdx.x + 256 * i) / RL;
        if (m0 + prow >= g.M) continue;
        float4 v = bg_ld4(Cs + prow * CP + 4 * c4);
        if (g.acc0) v = bg_add4(v, e_acc[i]);
        v = bg_add4(v, bv);
        if (g.row_scale) v = bg_scale4(e_rs[i], v);
        if (g.pre_add) v = bg_add4(v, e_pre[i]);
        if (g.act == BG_ACT_LEAKY) {
            v.x = v.x > 0.f ? v.x : v.x * g.slope; v.y = v.y > 0.f ? v.y : v.y * g.slope;
            v.z = v.z > 0.f ? v.z : v.z * g.slope; v.w = v.w > 0.f ? v.w : v.w * g.slope;
        }
        if (g.C2) bg_st4(g.C2 + e_row[i] * g.ldc2 + n0 + 4 * c4, v);
        if (g.residual) v = bg_add4(v, e_res[i]);
        bg_st4(g.C + e_row[i] * g.ldc + n0 + 4 * c4, v);
    }
}

// ---------------------------------------------------------------------------------------------
// 64x64-tile GEMM for K-contiguous A (vector path): operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no staging registers) through a ring of STAGES buffers, so STAGES-1
// K steps of loads are in flight while one is multiplied.  At the 1-3 workgroups per CU that the
// train step's shapes give, a step is otherwise bounded by the latency of its own loads.
//   * LDS images are lane-linear (an LDS-DMA writes base + 16*lane): [row][32] for a K-contiguous
//     operand with the 16-byte chunk index XOR-ed by (row >> 1) & 7 on the SOURCE address and on
//     the read (conflict-free ds_read_b128, two rows per 256-byte bank row); [k][64] for B[K,N].
//   * one s_barrier per step; loads are counted by hand (vmcnt), the compiler never sees them.
__device__ __attribute__((aligned(16))) float bg_zero_src[4] = {0.f, 0.f, 0.f, 0.f};   // never written

__device__ __forceinline__ void glds16(const float* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int STAGES, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm_dma(GemmArgs g) {
    constexpr int BK = 32, TILE_F = 64 * BK, STAGE_F = 2 * TILE_F;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // STAGES * STAGE_F floats
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t lid = blockIdx.x;
    if (g.xcd) {
        const uint32_t q = gridDim.x >> 3, rem = gridDim.x & 7, x = lid & 7, slot = lid >> 3;
        lid = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + slot;
    }
    const int bx = lid % g.gx, by = lid / g.gx;
    const int64_t m0 = (int64_t)by * 64, n0 = (int64_t)bx * 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
    const int n_steps = g.total_steps;

    // this lane's part of the two A and two B DMA instructions its wave issues per stage
    int a_row[2], a_k[2], b_k[2];
    int64_t a_goff[2], b_goff[2];
    uint32_t a_dst[2], b_dst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave * 2 + j;                       // 8 instructions cover a 64-row tile
        const int row = q * 8 + (lane >> 3), cp = lane & 7;
        a_row[j] = row;
        a_k[j] = 4 * (cp ^ ((row >> 1) & 7));             // source chunk of LDS chunk cp
        int64_t gr = m0 + row;
        a_goff[j] = gr < g.M ? gr : g.M - 1;              // rows past the edge: any valid row, never stored
        a_dst[j] = lds0 + 4u * (uint32_t)(q * 256);
        if (B_KC) {
            int64_t gn = n0 + row;
            gn = gn < g.N ? gn : g.N - 1;
            b_k[j] = a_k[j];
            b_goff[j] = gn * g.ldb;
        } else {
            b_k[j] = q * 4 + (lane >> 4);                 // [k][64]: 4 k rows per instruction
            int64_t gn = n0 + 4 * (lane & 15);
            b_goff[j] = gn + 4 <= g.N ? gn : g.N - 4;
        }
        b_dst[j] = lds0 + 4u * (uint32_t)(TILE_F + q * 256);
    }

    int it_seg = 0; int64_t it_k0 = 0, it_koff = 0;       // position of the next step to fetch
    auto issue = [&](int step) {
        const uint32_t sofs = 4u * (uint32_t)((step % STAGES) * STAGE_F);
        if (step < n_steps) {
            const int64_t K = g.Ks[it_seg];
            const float* Ab = g.A[it_seg];
            const int64_t lda = g.lda[it_seg];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool in_a = it_k0 + a_k[j] < K;
                glds16(in_a ? Ab + a_goff[j] * lda + it_k0 + a_k[j] : bg_zero_src, a_dst[j] + sofs);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool in_b = it_k0 + b_k[j] < K;
                const float* src = B_KC ? g.B + b_goff[j] + it_koff + it_k0 + b_k[j]
                                        : g.B + (it_koff + it_k0 + b_k[j]) * g.ldb + b_goff[j];
                glds16(in_b ? src : bg_zero_src, b_dst[j] + sofs);
            }
            it_k0 += BK;
            if (it_k0 >= K && it_seg + 1 < g.n_seg) { it_koff += K; ++it_seg; it_k0 = 0; }
        } else {
            // past the end: keep every wave's outstanding-load count uniform (target: a buffer nobody reads again)
#pragma unroll
            for (int j = 0; j < 2; ++j) { glds16(bg_zero_src, a_dst[j] + sofs); glds16(bg_zero_src, b_dst[j] + sofs); }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue(s);

    const int arow = wm * 32 + r, brow = wn * 32 + r;
    const int aswz = (arow >> 1) & 7, bswz = (brow >> 1) & 7;
    for (int step = 0; step < n_steps; ++step) {
        // this wave's loads of `step` have landed when at most (STAGES-2) newer stages are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (STAGES - 2)) : "memory");
        __builtin_amdgcn_s_barrier();                     // .. and everyone's; buffer (step-1) is free
        asm volatile("" ::: "memory");
        issue(step + STAGES - 1);
        const float* As = smem + (step % STAGES) * STAGE_F;
        const float* Bs = As + TILE_F;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const int c = 2 * kg + h;
            const float4 af = bg_ld4(As + arow * BK + 4 * (c ^ aswz));
            float4 bf;
            if (B_KC) {
                bf = bg_ld4(Bs + brow * BK + 4 * (c ^ bswz));
            } else {
                const float* p = Bs + (8 * kg + 4 * h) * 64 + brow;
                bf = make_float4(p[0], p[64], p[128], p[192]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
    }
    // no LDS-DMA may outlive the workgroup (its LDS is handed to the next one)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int64_t col = n0 + wn * 32 + r;
    if (col < g.N) store_tile(g, acc, m0 + wm * 32 + 4 * h, col, g.bias ? g.bias[col] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Skinny products (a few dozen output tiles, K of a few hundred: the coefficient recurrence of the
// collapsed discriminator, the program GNN, per-story terms).  With so few tiles a workgroup per 64x64
// tile leaves most CUs idle and walks its K steps one memory round trip at a time.  Here a workgroup
// owns a 32x32 tile, its four waves split K between them in groups of 8, fragments go global ->
// registers in the MFMA layout (no LDS staging, no barrier in the K loop, up to GB groups = 2*GB
// 16-byte loads per lane in flight), and the four partial tiles are summed through LDS.
template <bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_skinny(GemmArgs g) {
    constexpr int GB = 8;
    __shared__ float red[NW][32 * 33];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bx = blockIdx.x % g.gx, by = blockIdx.x / g.gx;
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const int64_t arow = m0 + r < g.M ? m0 + r : g.M - 1;       // edge rows / columns: any valid one, never stored
    const int64_t bcol = n0 + r < g.N ? n0 + r : g.N - 1;

    int n_groups = 0;
    for (int s = 0; s < g.n_seg; ++s) n_groups += (int)((g.Ks[s] + 7) >> 3);
    // shifted second segment: this lane's row takes c_lo * (row one block down) + c_hi * (row one block up)
    float c_lo = 0.f, c_hi = 0.f;
    bool has_lo = false, has_hi = false;
    if (g.cheb_R) {
        const int i = (int)(arow / g.cheb_R);
        has_lo = i >= 1; has_hi = i + 1 < g.cheb_K;
        c_lo = 0.5f; c_hi = 0.5f;
        if (g.cheb_T) { if (i == 0) c_hi = 1.f; } else { if (i == 1) c_lo = 1.f; }
    }

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

    for (int g0 = wave; g0 < n_groups; g0 += NW * GB) {
        float4 af[GB], bf[GB];
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            int gi = g0 + NW * j, seg = 0;
            int64_t koff = 0;
            af[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            bf[j] = af[j];
            if (gi < n_groups) {
                while (seg < g.n_seg - 1) {                       // wave-uniform walk over <= 4 segments
                    const int ng = (int)((g.Ks[seg] + 7) >> 3);
                    if (gi < ng) break;
                    gi -= ng; koff += g.Ks[seg]; ++seg;
                }
                const int64_t k = 8 * (int64_t)gi + 4 * h;
                if (k < g.Ks[seg]) {                              // K_s % 4 == 0: a 4-chunk is all in or all out
                    if (g.cheb_R && seg == 1) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (has_lo) {
                            const float4 a = bg_ld4(g.A[1] + (arow - g.cheb_R) * g.lda[1] + k);
                            v.x = c_lo * a.x; v.y = c_lo * a.y; v.z = c_lo * a.z; v.w = c_lo * a.w;
                        }
                        if (has_hi) {
                            const float4 b = bg_ld4(g.A[1] + (arow + g.cheb_R) * g.lda[1] + k);
                            v.x = __fmaf_rn(c_hi, b.x, v.x); v.y = __fmaf_rn(c_hi, b.y, v.y);
                            v.z = __fmaf_rn(c_hi, b.z, v.z); v.w = __fmaf_rn(c_hi, b.w, v.w);
                        }
                        af[j] = v;
                    } else {
                        af[j] = bg_ld4(g.A[seg] + arow * g.lda[seg] + k);
                    }
                    if (B_KC) {
                        bf[j] = bg_ld4(g.B + bcol * g.ldb + koff + k);
                    } else {
                        const float* bp = g.B + (koff + k) * g.ldb + bcol;
                        bf[j] = make_float4(bp[0], bp[g.ldb], bp[2 * g.ldb], bp[3 * g.ldb]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].x, bf[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].y, bf[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].z, bf[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].w, bf[j].w, acc, 0, 0, 0);
        }
    }
    // lane holds column r, rows (q&3) + 8*(q>>2) + 4*h
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave][((q & 3) + 8 * (q >> 2) + 4 * h) * 33 + r] = acc[q];
    __syncthreads();
    const int col_l = threadIdx.x & 31;
    const int64_t col = n0 + col_l;
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        const int row_l = (threadIdx.x >> 5) + 2 * NW * j;
        const int64_t row = m0 + row_l;
        if (row >= g.M) continue;
        const int o = row_l * 33 + col_l;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][o];
        v += bv;
        if (g.row_scale) v *= g.row_scale[row];
        if (g.pre_add) v += g.pre_add[(g.pre_rows ? (int64_t)g.pre_rows[row] : row) * g.ldp + col];
        if (g.act == BG_ACT_LEAKY) v = v > 0.f ? v : v * g.slope;
        if (g.residual) v += g.residual[row * g.ldr + col];
        g.C[row * g.ldc + col] = v;
    }
}

// The same product with whole-line loads (every skinny shape since round 3; the coefficient recurrence's [<= 520, 256 |
// 256] x [512, 256] is the one it was measured on).  k_gemm_skinny's fragment-shaped loads ask for 32 different 128-byte lines per instruction and use 32
// bytes of each; with ~20 such loads per lane and 8 waves the CU's address / tag path (not the bytes) is what the
// kernel waits for: 3.1 of its 6.9 us (ablations in DESIGN.md section 6, round 3).  Here a wave owns whole 32-k
// CHUNKS (one 128-byte line per operand row): it loads a chunk by full lines (8 lanes per row, 8 rows per
// instruction), turns it into MFMA fragments through a wave-private LDS tile -- LDS instructions of one wave execute
// in order, so no workgroup barrier is involved -- and multiplies.  Chunks go to waves round robin (wave, wave + 8, ..),
// two in flight per wave, so with two equal segments every wave carries one plain and one shifted chunk.
// Epilogue operands are requested before the K loop.  k order inside a chunk as in k_gemm_skinny's groups.
template <bool B_KC>
__global__ __launch_bounds__(512) void k_gemm_skinny_lines(GemmArgs g) {
    constexpr int NW = 8, P = 36, TILE = 32 * P;             // [row][32 + 4] floats: conflict-free ds_read_b128 fragments
    constexpr int PER_WAVE = B_KC ? 2 * TILE : TILE;         // >= 32 * 33, the partial tile of the reduction
    extern __shared__ __align__(16) float sk_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int lr = lane >> 3, lk = 4 * (lane & 7);           // staging: row lr of 8, 16 bytes at k = lk
    const int bx = blockIdx.x % g.gx, by = blockIdx.x / g.gx;
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const int64_t bcol = n0 + r < g.N ? n0 + r : g.N - 1;
    float* As = sk_smem + wave * PER_WAVE;
    float* Bs = As + TILE;

    int n_chunks = 0;
    for (int s = 0; s < g.n_seg; ++s) n_chunks += (int)((g.Ks[s] + 31) >> 5);

    // epilogue operands of this thread's two outputs, requested now
    const int col_l = threadIdx.x & 31;
    const int64_t col = n0 + col_l;
    float e_pre[16 / NW], e_res[16 / NW];
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        const int64_t row = m0 + (threadIdx.x >> 5) + 2 * NW * j;
        e_pre[j] = 0.f; e_res[j] = 0.f;
        if (row < g.M && col < g.N) {
            if (g.pre_add) e_pre[j] = g.pre_add[(g.pre_rows ? (int64_t)g.pre_rows[row] : row) * g.ldp + col];
            if (g.residual) e_res[j] = g.residual[row * g.ldr + col];
        }
    }

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

    for (int c0 = wave; c0 < n_chunks; c0 += 2 * NW) {
        float4 a[2][4], b[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int c = c0 + NW * u, seg = 0;
            int64_t koff = 0;
            if (c >= n_chunks) continue;
            while (seg < g.n_seg - 1) {                       // wave-uniform walk over <= 4 segments
                const int nc = (int)((g.Ks[seg] + 31) >> 5);
                if (c < nc) break;
                c -= nc; koff += g.Ks[seg]; ++seg;
            }
            const int64_t k0 = 32 * (int64_t)c;
            const bool a_in = k0 + lk < g.Ks[seg];            // a segment's last chunk may be short (K_s % 4 == 0: a 16-byte
                                                              // piece is all in or all out); the rest multiplies as zeros
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t rr = 8 * i + lr;
                const int64_t arow = m0 + rr < g.M ? m0 + rr : g.M - 1;       // edge rows: any valid one, never stored
                if (!a_in) {
                    a[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (g.cheb_R && seg == 1) {
                    // c_lo * (row one block down) + c_hi * (row one block up), the arithmetic of k_cheb_shift
                    const int blk = (int)(arow / g.cheb_R);
                    float c_lo = 0.5f, c_hi = 0.5f;
                    if (g.cheb_T) { if (blk == 0) c_hi = 1.f; } else { if (blk == 1) c_lo = 1.f; }
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (blk >= 1) {
                        const float4 x = bg_ld4(g.A[1] + (arow - g.cheb_R) * g.lda[1] + k0 + lk);
                        v.x = c_lo * x.x; v.y = c_lo * x.y; v.z = c_lo * x.z; v.w = c_lo * x.w;
                    }
                    if (blk + 1 < g.cheb_K) {
                        const float4 y = bg_ld4(g.A[1] + (arow + g.cheb_R) * g.lda[1] + k0 + lk);
                        v.x = __fmaf_rn(c_hi, y.x, v.x); v.y = __fmaf_rn(c_hi, y.y, v.y);
                        v.z = __fmaf_rn(c_hi, y.z, v.z); v.w = __fmaf_rn(c_hi, y.w, v.w);
                    }
                    a[u][i] = v;
                } else {
                    a[u][i] = bg_ld4(g.A[seg] + arow * g.lda[seg] + k0 + lk);
                }
                if (B_KC) {
                    const int64_t brow = n0 + rr < g.N ? n0 + rr : g.N - 1;
                    b[u][i] = a_in ? bg_ld4(g.B + brow * g.ldb + koff + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {                                      // B [K, N]: a lane's column, 4 consecutive k -- whole lines already
                    b[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + 8 * i + 4 * h < g.Ks[seg]) {
                        const float* bp = g.B + (koff + k0 + 8 * i + 4 * h) * g.ldb + bcol;
                        b[u][i] = make_float4(bp[0], bp[g.ldb], bp[2 * g.ldb], bp[3 * g.ldb]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (c0 + NW * u >= n_chunks) continue;
            // the previous chunk's fragment reads were issued before these writes (same wave: executed in order)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bg_st4(As + (8 * i + lr) * P + lk, a[u][i]);
                if (B_KC) bg_st4(Bs + (8 * i + lr) * P + lk, b[u][i]);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 af = bg_ld4(As + r * P + 8 * j + 4 * h);
                const float4 bf = B_KC ? bg_ld4(Bs + r * P + 8 * j + 4 * h) : b[u][j];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // partial tiles into the waves' own (now idle) staging areas; lane holds column r, rows (q&3) + 8*(q>>2) + 4*h
#pragma unroll
    for (int q = 0; q < 16; ++q) As[((q & 3) + 8 * (q >> 2) + 4 * h) * 33 + r] = acc[q];
    __syncthreads();
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
        const int row_l = (threadIdx.x >> 5) + 2 * NW * j;
        const int64_t row = m0 + row_l;
        if (row >= g.M) continue;
        const int o = row_l * 33 + col_l;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sk_smem[w * PER_WAVE + o];
        v += bv;
        if (g.row_scale) v *= g.row_scale[row];
        if (g.pre_add) v += e_pre[j];
        if (g.act == BG_ACT_LEAKY) v = v > 0.f ? v : v * g.slope;
        if (g.residual) v += e_res[j];
        g.C[row * g.ldc + col] = v;
    }
}

__global__ void k_splitk_reduce(const float* __restrict__ partial, int splits, int64_t M, int64_t N,
                                float* __restrict__ C, int64_t ldc, int accumulate) {
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
    for (int j = 0; j < 4; ++j) {
        float* c = C + ((t + j) / N) * ldc + ((t + j) % N);
        *c = accumulate ? *c + v[j] : v[j];
    }
}

// Few outputs, many splits (weight gradients of the narrow layers): the serial walk over the splits is the
// whole cost, so 16 waves share it -- wave w adds splits w, w+16, .. for 64 consecutive outputs, the 16
// partial sums are combined through LDS in wave order (deterministic).
__global__ __launch_bounds__(1024) void k_splitk_reduce_small(const float* __restrict__ partial, int splits,
                                                              int64_t M, int64_t N, float* __restrict__ C, int64_t ldc,
                                                              int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t MN = M * N, t = (int64_t)blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (t < MN) {
        int z = wave;
        for (; z + 48 < splits; z += 64) {
            const float a = partial[(int64_t)z * MN + t], b = partial[(int64_t)(z + 16) * MN + t];
            const float c = partial[(int64_t)(z + 32) * MN + t], d = partial[(int64_t)(z + 48) * MN + t];
            acc += a; acc += b; acc += c; acc += d;
        }
        for (; z < splits; z += 16) acc += partial[(int64_t)z * MN + t];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && t < MN) {
        float v = red[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += red[w][lane];
        float* c = C + (t / N) * ldc + (t % N);
        *c = accumulate ? *c + v : v;
    }
}

__global__ void k_splitk_reduce_scalar(const float* __restrict__ partial, int splits, int64_t M, int64_t N,
                                       float* __restrict__ C, int64_t ldc, int accumulate) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    float acc = 0.f;
    for (int z = 0; z < splits; ++z) acc += partial[(int64_t)z * M * N + t];
    float* c = C + (t / N) * ldc + (t % N);
    *c = accumulate ? *c + acc : acc;
}

// Column sums, phase 1: workgroup (bx, by) owns 64 columns x rows [by*RPB, (by+1)*RPB); its 4 waves
// take every 4th row, 8 loads in flight per lane; the waves' sums are combined through LDS.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int64_t ldx, int64_t M, int64_t N,
                                                        int rows_per_block, float* __restrict__ partial, int accumulate) {
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
    if (wave == 0 && n < N) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float* p = partial + (int64_t)blockIdx.y * N + n;
        *p = accumulate ? *p + v : v;       // accumulate only in the single-block form (partial == the result)
    }
}

// phase 2: one wave per column adds the row-block partials (fixed shuffle tree: deterministic)
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int n_blocks, int64_t N,
                                                      float* __restrict__ out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc = 0.f;
    for (int b = lane; b < n_blocks; b += 64) acc += partial[(int64_t)b * N + n];
    acc = bg_wave_sum(acc);
    if (lane == 0) out[n] = accumulate ? out[n] + acc : acc;
}

// tuning hook: BG_GEMM_XCD=0 keeps the hardware's round-robin tile order
int xcd_mode() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_XCD", 1) != 0;
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

// tuning hook: BG_GEMM_PANEL=0 keeps the tile kernels for the narrow (N = 128) products
int panel_mode() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_PANEL", 1) != 0;
    }
    return cached;
}

template <int TMT, int NJ, bool NBR = false>
int launch_gemm_panel(GemmArgs g, hipStream_t s) {
    constexpr size_t lds_bytes = 2 * (size_t)(16 * TMT + 64 * NJ) * (64 + BG_PANEL_PAD) * sizeof(float);
    auto kern = k_gemm_panel<TMT, NJ, NBR>;
    static bool raised[BG_MAX_DEVICES] = {};
    int dev = 0;
    BG_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= BG_MAX_DEVICES) return BG_ERR_ARG;
    if (!raised[dev]) {       // above the 64 KB static limit: opt in once per device
        BG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        raised[dev] = true;
    }
    const int64_t panels = (g.M + 16 * TMT - 1) / (16 * TMT);
    kern<<<dim3((unsigned)panels, (unsigned)(128 / (64 * NJ))), 256, lds_bytes, s>>>(g);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// tuning hook: BG_GEMM_PANEL_NJ=2 gives a panel workgroup all 128 columns (one workgroup per CU); the default 1 splits
// the columns over two workgroups that share a CU, so one's prologue / epilogue overlaps the other's MFMAs
int panel_nj() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_PANEL_NJ", 1) == 2 ? 2 : 1;
    }
    return cached;
}

// tuning hook: BG_GEMM_DMA=0 keeps the register-staged kernel for every shape; 2..4 force a ring depth
int dma_mode() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_DMA", 1);       // 1 = by shape (see bg_gemm_f32)
        if (cached < 0 || cached > 4) cached = 1;
    }
    return cached;
}

template <int STAGES>
int launch_gemm_dma(int transB, GemmArgs g, hipStream_t s) {
    g.gx = (int)((g.N + 63) / 64);
    g.gy = (int)((g.M + 63) / 64);
    g.xcd = xcd_mode();
    const int64_t blocks = (int64_t)g.gx * g.gy;
    if (blocks > 0x7fffffff) return BG_ERR_ARG;
    constexpr size_t lds_bytes = (size_t)STAGES * 2 * 64 * 32 * sizeof(float);
    static_assert(lds_bytes <= 65536, "ring must fit the default dynamic LDS limit");
    if (transB) k_gemm_dma<STAGES, true><<<dim3((unsigned)blocks), 256, lds_bytes, s>>>(g);
    else k_gemm_dma<STAGES, false><<<dim3((unsigned)blocks), 256, lds_bytes, s>>>(g);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// tuning hook: BG_GEMM_SKINNY=0 disables the 32x32-tile split-wave kernel
int skinny_mode() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_SKINNY", 1) != 0;
    }
    return cached;
}

// tuning hook: BG_GEMM_SKINNY_TILES = largest 32x32-tile count that still takes the skinny kernel
int skinny_max_tiles() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_SKINNY_TILES", 256);
    }
    return cached;
}

int launch_gemm_skinny(int transB, GemmArgs g, hipStream_t s) {
    g.gx = (int)((g.N + 31) / 32);
    g.gy = (int)((g.M + 31) / 32);
    const unsigned blocks = (unsigned)(g.gx * g.gy);
    int groups = 0;
    for (int i = 0; i < g.n_seg; ++i) groups += (int)((g.Ks[i] + 7) >> 3);
    static int force = -1;
    if (force < 0) force = bg_tune_int("BG_GEMM_SKINNY_WAVES", 0);
    const bool wide = force ? force == 8 : groups > 32;      // K > 256: eight waves share it, one load round each
    // whole-line loads through wave-private LDS tiles: every shape (the fragment-load kernels below stay for A/B runs in
    // tuning builds, BG_GEMM_SKINNY_LINES=0)
    if (bg_tune_int("BG_GEMM_SKINNY_LINES", 1) != 0) {
        const size_t lds_bytes = (size_t)8 * (transB ? 2 : 1) * 32 * 36 * sizeof(float);
        if (transB) {
            static bool raised[BG_MAX_DEVICES] = {};         // 73,728 bytes: above the 64 KB static limit, opt in per device
            int dev = 0;
            BG_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= BG_MAX_DEVICES) return BG_ERR_ARG;
            if (!raised[dev]) {
                BG_HIP(hipFuncSetAttribute((const void*)k_gemm_skinny_lines<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                raised[dev] = true;
            }
            k_gemm_skinny_lines<true><<<blocks, 512, lds_bytes, s>>>(g);
        } else {
            k_gemm_skinny_lines<false><<<blocks, 512, lds_bytes, s>>>(g);
        }
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (wide) {
        if (transB) k_gemm_skinny<true, 8><<<blocks, 512, 0, s>>>(g);
        else k_gemm_skinny<false, 8><<<blocks, 512, 0, s>>>(g);
    } else {
        if (transB) k_gemm_skinny<true, 4><<<blocks, 256, 0, s>>>(g);
        else k_gemm_skinny<false, 4><<<blocks, 256, 0, s>>>(g);
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

bool operand_vec_ok(const float* p, int64_t ld, bool kc, int64_t K, int64_t rows) {
    if (!bg_aligned16(p) || ld % 4 != 0) return false;
    return kc ? (K % 4 == 0) : (rows % 4 == 0);
}

// tuning hook: BG_GEMM_TILE = 1..4 forces 64x64 / 128x64 / 64x128 / 128x128 (unset: heuristic)
int tile_override() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_TILE", 0);
        if (cached < 0 || cached > 4) cached = 0;
    }
    return cached;
}

// tuning hook: BG_GEMM_BK = 32 / 64 forces the K depth of a 64x64-tile step (unset: 64)
int bk_override() {
    static int cached = -1;
    if (cached < 0) {
        cached = bg_tune_int("BG_GEMM_BK", 0);
        if (cached != 32 && cached != 64) cached = 0;
    }
    return cached;
}

}  // namespace

extern "C" {

// bg_gemm_f32 and bg_gemm_nbr_f32 (nbr != NULL: only the row-panel kernel forms the addend on load; any other shape is
// declined with BG_ERR_SHAPE before anything is launched)
static int gemm_entry(int transA, int transB, int64_t M, int64_t N, int n_seg,
                      const float* const* A, const int64_t* lda, const int64_t* Ks,
                      const float* B, int64_t ldb, float* C, int64_t ldc,
                      const float* bias, const float* row_scale, int act, float slope,
                      const float* residual, int64_t ldr, const float* pre_add, int64_t ldp,
                      const int32_t* pre_add_rows, const float* acc_init, int64_t ld_acc,
                      float* pre_residual_out, int64_t ld_pro,
                      const float* nbr, int64_t ldn, const int32_t* nbr_rowptr, const int32_t* nbr_col,
                      const float* nbr_count, bgStream stream) {
    if (M < 0 || N < 0 || n_seg < 1 || n_seg > 4 || !A || !lda || !Ks) return BG_ERR_ARG;
    if (nbr && (!pre_add || !nbr_rowptr || !nbr_col || !nbr_count || ldn < N)) return BG_ERR_ARG;
    if (M == 0 || N == 0) return BG_OK;
    if (!B || !C || ldc < N || (residual && ldr < N) || (pre_add && ldp < N) || (acc_init && ld_acc < N)) return BG_ERR_ARG;
    GemmArgs g{};
    g.nbr = nbr; g.ldn = ldn; g.nbr_rowptr = nbr_rowptr; g.nbr_col = nbr_col; g.nbr_count = nbr_count;
    g.n_seg = n_seg;
    g.acc0 = acc_init; g.ld0 = ld_acc;
    if (pre_residual_out && ld_pro < N) return BG_ERR_ARG;
    g.C2 = pre_residual_out; g.ldc2 = ld_pro;
    g.pre_add = pre_add; g.ldp = ldp; g.pre_rows = pre_add ? pre_add_rows : nullptr;
    int tile = tile_override();
    if (tile == 0) {
        const int64_t big_tiles = ((M + 127) / 128) * ((N + 127) / 128);
        // narrow outputs (N <= 128: one 128-wide column tile) keep two workgroups per CU with 128x64 tiles:
        // M = 2^20, N = 128: 1031 us (K = 260) / 452 us (K = 128) against 1248 / 707 us with 128x128 (tools/big_gemm.py)
        tile = big_tiles >= 512 ? (N <= 128 ? 1 : 3) : 0;
    } else {
        tile -= 1;
    }
    // 64x64 tiles run 1-3 workgroups per CU, so a K step is bounded by the latency of its global
    // loads: 64-deep steps halve the number of load -> LDS -> barrier round trips
    int bk = tile == 0 ? 64 : 32;
    if (tile == 0 && bk_override()) bk = bk_override();
    // LDS-DMA ring kernel: K-contiguous A on the vector path (decided below), 64x64 tiles
    int dma = (tile == 0 && !transA && M >= 1 && N >= 4 && !acc_init) ? dma_mode() : 0;
    if (dma == 1) {
        // measured on the train step's shapes (tools/gemm_shapes.py): with >= 2 workgroups per CU the
        // register-free ring wins (33 vs 38 us at 720 tiles), below that the 64-deep register-staged
        // steps do (15 vs 16.5 us at 360 tiles)
        const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
        dma = tiles >= 512 ? 2 : 0;
    }
    for (int s = 0; s < n_seg; ++s) if (Ks[s] <= 0) dma = 0;      // its step walk assumes non-empty segments
    if (dma) bk = 32;
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
    // narrow output, K-contiguous operands, a few thousand rows: one row panel per CU (k_gemm_panel)
    if (tile_override() == 0 && panel_mode() && !transA && transB && vec && N == 128 && M >= 2048 && M <= 64 * 256 &&
        !bk_override()) {
        bool whole = true;
        int64_t psteps = 0;
        for (int s = 0; s < n_seg; ++s) { whole = whole && Ks[s] > 0 && Ks[s] % 64 == 0; psteps += Ks[s] / 64; }
        // its epilogue moves whole rows as float4: every epilogue operand 16-byte aligned with a pitch % 4 == 0
        auto al = [](const void* p, int64_t ld) { return !p || (bg_aligned16(p) && ld % 4 == 0); };
        whole = whole && al(C, ldc) && al(residual, ldr) && al(pre_add, ldp) && al(acc_init, ld_acc) &&
                al(pre_residual_out, ld_pro) && al(bias, 0) && al(nbr, ldn);
        if (whole) {
            g.total_steps = (int)psteps; g.steps_per_split = (int)psteps;
            int64_t t = (M + 16 * 256 - 1) / (16 * 256);                 // 16-row tiles per panel so that panels <= 256
            { static int force = -1; if (force < 0) force = bg_tune_int("BG_GEMM_PANEL_T", 0);
              if (force >= 2 && force <= 4) t = force; }
            if (nbr) {
                if (t <= 2) return launch_gemm_panel<2, 1, true>(g, st);
                if (t == 3) return launch_gemm_panel<3, 1, true>(g, st);
                return BG_ERR_SHAPE;      // 64-row panels: the neighbour registers would cost the second workgroup per CU
            }
            if (panel_nj() == 2) {
                if (t <= 2) return launch_gemm_panel<2, 2>(g, st);
                if (t == 3) return launch_gemm_panel<3, 2>(g, st);
                return launch_gemm_panel<4, 2>(g, st);
            }
            if (t <= 2) return launch_gemm_panel<2, 1>(g, st);
            if (t == 3) return launch_gemm_panel<3, 1>(g, st);
            return launch_gemm_panel<4, 1>(g, st);
        }
    }
    if (nbr) return BG_ERR_SHAPE;          // only the panel kernel forms the addend on load: the caller runs two launches
    // few tiles and a moderate K: one 32x32 tile per workgroup, K split over its waves (needs the
    // 16-byte path on A; B is read by 16-byte chunks when K-contiguous, by dwords otherwise)
    if (tile_override() == 0 && !transA && vec && skinny_mode() && Ktot <= 4096 && !acc_init && !pre_residual_out &&
        ((M + 31) / 32) * ((N + 31) / 32) <= skinny_max_tiles()) {
        bool nonempty = true;
        for (int s = 0; s < n_seg; ++s) nonempty = nonempty && Ks[s] > 0;
        if (nonempty) return launch_gemm_skinny(transB, g, st);
    }
    if (dma && vec) {
        switch (dma) {
            case 2: return launch_gemm_dma<2>(transB, g, st);
            case 4: return launch_gemm_dma<4>(transB, g, st);
            default: return launch_gemm_dma<3>(transB, g, st);
        }
    }
    if (dma) {                     // scalar path keeps its own step depth
        steps = 0;
        bk = bk_override() ? bk_override() : 64;
        for (int s = 0; s < n_seg; ++s) steps += (int)((Ks[s] + bk - 1) / bk);
        g.total_steps = steps; g.steps_per_split = steps;
    }
    switch (tile) {
        case 3: return launch_gemm<128, 128, 32, false>(transA, transB, vec, g, 1, st);
        case 2: return launch_gemm<64, 128, 32, false>(transA, transB, vec, g, 1, st);
        case 1: return launch_gemm<128, 64, 32, false>(transA, transB, vec, g, 1, st);
        default:
            if (bk == 64) return launch_gemm<64, 64, 64, false>(transA, transB, vec, g, 1, st);
            return launch_gemm<64, 64, 32, false>(transA, transB, vec, g, 1, st);
    }
}

int bg_gemm_f32(int transA, int transB, int64_t M, int64_t N, int n_seg,
                const float* const* A, const int64_t* lda, const int64_t* Ks,
                const float* B, int64_t ldb, float* C, int64_t ldc,
                const float* bias, const float* row_scale, int act, float slope,
                const float* residual, int64_t ldr, const float* pre_add, int64_t ldp,
                const int32_t* pre_add_rows, const float* acc_init, int64_t ld_acc,
                float* pre_residual_out, int64_t ld_pro, bgStream stream) {
    return gemm_entry(transA, transB, M, N, n_seg, A, lda, Ks, B, ldb, C, ldc, bias, row_scale, act, slope, residual, ldr,
                      pre_add, ldp, pre_add_rows, acc_init, ld_acc, pre_residual_out, ld_pro,
                      nullptr, 0, nullptr, nullptr, nullptr, stream);
}

int bg_gemm_nbr_f32(int64_t M, int64_t N, int n_seg, const float* const* A, const int64_t* lda, const int64_t* Ks,
                    const float* W, int64_t ldw, float* C, int64_t ldc, const float* bias, int act, float slope,
                    const float* residual, int64_t ldr,
                    const float* self_add, int64_t ld_self, const float* nbr, int64_t ld_nbr,
                    const int32_t* nbr_rowptr, const int32_t* nbr_col, const float* nbr_count, const int32_t* rows,
                    const float* acc_init, int64_t ld_acc, float* pre_residual_out, int64_t ld_pro, bgStream stream) {
    if (!nbr || !self_add || !rows) return BG_ERR_ARG;
    if (bg_tune_int("BG_GEMM_NBR", 1) == 0) return BG_ERR_SHAPE;      // tuning builds: A/B against the two launches
    return gemm_entry(0, 1, M, N, n_seg, A, lda, Ks, W, ldw, C, ldc, bias, nullptr, act, slope, residual, ldr,
                      self_add, ld_self, rows, acc_init, ld_acc, pre_residual_out, ld_pro,
                      nbr, ld_nbr, nbr_rowptr, nbr_col, nbr_count, stream);
}

int bg_gemm_cheb_f32(int transB, int64_t M, int64_t N, const float* X, int64_t ldx, int64_t K, int64_t block_rows,
                     int64_t n_blocks, int transpose, const float* B, int64_t ldb, float* C, int64_t ldc,
                     const float* residual, int64_t ldr, const float* pre_add, int64_t ldp, bgStream stream) {
    if (M < 0 || N < 0 || K <= 0 || block_rows <= 0 || n_blocks <= 0) return BG_ERR_ARG;
    if (M == 0 || N == 0) return BG_OK;
    if (!X || !B || !C || ldx < K || ldc < N || (residual && ldr < N) || (pre_add && ldp < N)) return BG_ERR_ARG;
    if (M > block_rows * n_blocks || ldb < (transB ? 2 * K : N)) return BG_ERR_ARG;
    if (C == X) return BG_ERR_ARG;                         // rows one block away are read while others are written
    if (K % 4 != 0 || 2 * K > 4096 || ((M + 31) / 32) * ((N + 31) / 32) > skinny_max_tiles() ||
        block_rows * n_blocks > 0x7fffffff) return BG_ERR_SHAPE;
    if (!operand_vec_ok(X, ldx, true, K, M) || !operand_vec_ok(B, ldb, transB != 0, 4, N)) return BG_ERR_ALIGN;
    GemmArgs g{};
    g.n_seg = 2;
    g.A[0] = X; g.A[1] = X; g.lda[0] = ldx; g.lda[1] = ldx; g.Ks[0] = K; g.Ks[1] = K;
    g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.act = BG_ACT_NONE; g.slope = 0.01f;
    g.residual = residual; g.ldr = ldr; g.pre_add = pre_add; g.ldp = ldp;
    g.M = M; g.N = N;
    g.cheb_R = (int)block_rows; g.cheb_K = (int)n_blocks; g.cheb_T = transpose ? 1 : 0;
    return launch_gemm_skinny(transB, g, (hipStream_t)stream);
}

int bg_gemm_tn_splitk_f32(int64_t M, int64_t R, const float* A, int64_t lda,
                          int n_seg, const float* const* B, const int64_t* ldb, const int64_t* Ns,
                          float* C, int64_t ldc, float* partial, int32_t splits, int accumulate, bgStream stream) {
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
    if (M * N <= 65536 && splits >= 32)
        k_splitk_reduce_small<<<bg_grid_for(M * N, 64, 0x7fffffff), 1024, 0, st>>>(partial, splits, M, N, C, ldc, accumulate);
    else if ((M * N) % 4 == 0 && bg_aligned16(partial))
        k_splitk_reduce<<<bg_grid_for((M * N) / 4, 256, 0x7fffffff), 256, 0, st>>>(partial, splits, M, N, C, ldc, accumulate);
    else
        k_splitk_reduce_scalar<<<bg_grid_for(M * N, 256, 0x7fffffff), 256, 0, st>>>(partial, splits, M, N, C, ldc, accumulate);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_colsum_f32(const float* x, int64_t ldx, int64_t M, int64_t N, float* partial, int32_t n_blocks,
                  float* out, int accumulate, bgStream stream) {
    if (M < 0 || N < 0 || n_blocks < 1) return BG_ERR_ARG;
    if (N == 0) return BG_OK;
    if (!out || !partial || (M > 0 && (!x || ldx < N))) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int rows_per_block = (int)((M + n_blocks - 1) / n_blocks);
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)n_blocks);
    // one row block: its "partial" row is the result (same value as the two-phase form would produce)
    k_colsum_partial<<<grid, 256, 0, s>>>(x, ldx, M, N, rows_per_block > 0 ? rows_per_block : 1, n_blocks == 1 ? out : partial,
                                          n_blocks == 1 ? accumulate : 0);
    BG_LAUNCH_CHECK();
    if (n_blocks == 1) return BG_OK;
    k_colsum_final<<<bg_grid_for(N, 4), 256, 0, s>>>(partial, n_blocks, N, out, accumulate);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
