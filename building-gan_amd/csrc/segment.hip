// Segment kernels: CSR gather-reduce (sum / mean / weighted), long-segment two-phase reductions,
// segment max with arg, arg gather / scatter, sampled dot products.  HBM-bound: 16-byte gathers,
// one float4 per lane, rows owned by lane groups (no atomics), slot order = COO order.
#include "common.h"
#include <mutex>

namespace {

struct SpmmArgs {
    const float* x; int64_t ldx;
    const int32_t* rowptr; const int32_t* col;
    const float* edge_w; const int32_t* eperm;
    const float* row_scale; int row_mode;
    const float* col_scale; int col_mode;
    float* out; int64_t ldo;
    int64_t n_rows; int H4;   // H4 = H / 4 float4 columns (<= 64 per launch)
    const float* add1; int64_t ld1;   // optional addends: out = rs * sum + add1 + add2
    const float* add2; int64_t ld2;
    int nt_store;                     // write the result with non-temporal stores (set by spmm_dispatch for big outputs)
    // optional gathered addend, last in the sum: + gs(grow[i]) applied to gadd[grow[i], :]
    const float* gadd; int64_t ldga; const int32_t* grow; const float* gscale; int gmode;
    // bg_spmm_batch_f32: blockIdx.y = the batch, which reads x + y * x_step and writes out + y * out_step
    int64_t x_step, out_step; int n_batch;
};

__device__ __forceinline__ float apply_scale(float v, float s, int mode) {
    return mode == BG_SCALE_MUL ? v * s : (mode == BG_SCALE_DIV ? v / s : v);
}

// XCD-aware block remap (bijective): consecutive logical blocks stay on one XCD's L2.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb) {
    int64_t q = nb >> 3, r = nb & 7, xcd = b & 7, i = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// LPR lanes own one output row (one float4 each); a wave owns 64/LPR rows.
template <int LPR, bool GENERAL>
__device__ __forceinline__ void spmm_vec_body(const SpmmArgs& a, int64_t blk) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane % LPR;
    const int64_t row = (blk * (blockDim.x >> 6) + wave) * RPW + lane / LPR;
    if (row >= a.n_rows || sub >= a.H4) return;
    const int32_t b = a.rowptr[row], e = a.rowptr[row + 1];
    const float* xb = a.x + 4 * sub;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t k = b;
    if (!GENERAL) {
        for (; k + 4 <= e; k += 4) {
            int32_t c0 = a.col[k], c1 = a.col[k + 1], c2 = a.col[k + 2], c3 = a.col[k + 3];
            float4 v0 = bg_ld4(xb + (int64_t)c0 * a.ldx);
            float4 v1 = bg_ld4(xb + (int64_t)c1 * a.ldx);
            float4 v2 = bg_ld4(xb + (int64_t)c2 * a.ldx);
            float4 v3 = bg_ld4(xb + (int64_t)c3 * a.ldx);
            acc = bg_add4(acc, v0); acc = bg_add4(acc, v1); acc = bg_add4(acc, v2); acc = bg_add4(acc, v3);
        }
        for (; k < e; ++k) acc = bg_add4(acc, bg_ld4(xb + (int64_t)a.col[k] * a.ldx));
    } else {
        for (; k < e; ++k) {
            int32_t c = a.col ? a.col[k] : k;
            float w = 1.f;
            if (a.edge_w) w = a.edge_w[a.eperm ? a.eperm[k] : k];
            if (a.col_mode) w = apply_scale(w, a.col_scale[c], a.col_mode);
            float4 v = bg_ld4(xb + (int64_t)c * a.ldx);
            acc = bg_fma4(w, v, acc);          // explicit fma: k_spmm_chain repeats exactly this arithmetic
        }
    }
    if (a.row_mode) {
        float s = a.row_scale[row];
        acc.x = apply_scale(acc.x, s, a.row_mode); acc.y = apply_scale(acc.y, s, a.row_mode);
        acc.z = apply_scale(acc.z, s, a.row_mode); acc.w = apply_scale(acc.w, s, a.row_mode);
    }
    if (a.add1) acc = bg_add4(acc, bg_ld4(a.add1 + row * a.ld1 + 4 * sub));
    if (a.add2) acc = bg_add4(acc, bg_ld4(a.add2 + row * a.ld2 + 4 * sub));
    if (a.gadd) {
        const int32_t r = a.grow[row];
        float4 t = bg_ld4(a.gadd + (int64_t)r * a.ldga + 4 * sub);
        if (a.gmode) {
            const float s = a.gscale[r];
            t.x = apply_scale(t.x, s, a.gmode); t.y = apply_scale(t.y, s, a.gmode);
            t.z = apply_scale(t.z, s, a.gmode); t.w = apply_scale(t.w, s, a.gmode);
        }
        acc = bg_add4(acc, t);
    }
    if (a.nt_store) {
        // a result far larger than the caches streams past them: the gathered rows keep the L2 / Infinity Cache
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f o = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(a.out + row * a.ldo + 4 * sub));
    } else {
        bg_st4(a.out + row * a.ldo + 4 * sub, acc);
    }
}

template <int LPR, bool GENERAL>
__global__ __launch_bounds__(256) void k_spmm_vec(SpmmArgs a) {
    a.x += (int64_t)blockIdx.y * a.x_step; a.out += (int64_t)blockIdx.y * a.out_step;       // (0 outside bg_spmm_batch_f32)
    spmm_vec_body<LPR, GENERAL>(a, xcd_remap(blockIdx.x, gridDim.x));
}

// The plain gather-sum (no edge weights, no column scale) with its memory round trips batched: a lane group owns R
// consecutive rows and issues, for all of them at once, the row pointers, then up to D column indices per row, then
// the gathers of every slot -- three dependent round trips per R rows where the loop above pays one per slot group
// (index, then row, then the next slot ..).  Sums run in slot order (bit-identical to the loop); slots past D
// (rows longer than D) continue in the loop form.
// Round 3 (tools/micro/spmm_lab.hip, profiles/r03_gather_*): R = 2 rows with D = 6 register slots (voxel grids have
// at most six neighbours; 60 VGPRs, 8 waves per SIMD) against R = 1, D = 8 (40 VGPRs): 10^6 edges streaming from
// HBM 61.9 -> 57.2 us (H = 128), 125.9 -> 114.2 us (H = 256); operand in the Infinity Cache 39.8 -> 34.0 us.  The
// kernel is bound by the CU's queue of outstanding L1 -> L2 read requests (64 lines of 128 bytes per CU: PMC
// TCP_TCC_READ_REQ_LATENCY / kernel cycles / 256 = 45 with R = 1, 51 with R = 2, 62-66 for a persistent pipelined
// form that loses the gain to L2 misses); R = 4 (116 VGPRs, 4 waves per SIMD) is slower than R = 1.
template <int LPR, int R, int D>
__global__ __launch_bounds__(256) void k_spmm_batched(SpmmArgs a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR;
    const int64_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = ((blk * (blockDim.x >> 6) + wave) * RPW + lane / LPR) * R;
    if (row0 >= a.n_rows || sub >= a.H4) return;
    int32_t rp[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) rp[i] = a.rowptr[row0 + i < a.n_rows ? row0 + i : a.n_rows];
    int32_t c[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s = 0; s < D; ++s) c[r][s] = rp[r] + s < rp[r + 1] ? a.col[rp[r] + s] : -1;
    const float* xb = a.x + 4 * sub;
    float4 v[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (c[r][s] >= 0) v[r][s] = bg_ld4(xb + (int64_t)c[r][s] * a.ldx);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        if (row >= a.n_rows) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (c[r][s] >= 0) acc = bg_add4(acc, v[r][s]);
        for (int32_t k = rp[r] + D; k < rp[r + 1]; ++k) acc = bg_add4(acc, bg_ld4(xb + (int64_t)a.col[k] * a.ldx));
        if (a.row_mode) {
            const float sc = a.row_scale[row];
            acc.x = apply_scale(acc.x, sc, a.row_mode); acc.y = apply_scale(acc.y, sc, a.row_mode);
            acc.z = apply_scale(acc.z, sc, a.row_mode); acc.w = apply_scale(acc.w, sc, a.row_mode);
        }
        if (a.add1) acc = bg_add4(acc, bg_ld4(a.add1 + row * a.ld1 + 4 * sub));
        if (a.add2) acc = bg_add4(acc, bg_ld4(a.add2 + row * a.ld2 + 4 * sub));
        if (a.nt_store) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f o = {acc.x, acc.y, acc.z, acc.w};
            __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(a.out + row * a.ldo + 4 * sub));
        } else {
            bg_st4(a.out + row * a.ldo + 4 * sub, acc);
        }
    }
}

template <int R, int D>
int launch_spmm_batched(const SpmmArgs& a, hipStream_t s) {
    int lpr = 1;
    while (lpr < a.H4) lpr <<= 1;
    const int gpb = 4 * (64 / lpr);                                   // lane groups per 256-thread workgroup
    const int grid = bg_grid_for((a.n_rows + R - 1) / R, gpb, 0x7fffffff);
    switch (lpr) {
        case 16: k_spmm_batched<16, R, D><<<grid, 256, 0, s>>>(a); break;
        case 32: k_spmm_batched<32, R, D><<<grid, 256, 0, s>>>(a); break;
        case 64: k_spmm_batched<64, R, D><<<grid, 256, 0, s>>>(a); break;
        default: return BG_ERR_SHAPE;
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// scalar path: one thread per (row, channel)
__global__ void k_spmm_scalar(SpmmArgs a, int H) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_rows * H) return;
    int64_t row = t / H; int h = (int)(t % H);
    float acc = 0.f;
    for (int32_t k = a.rowptr[row]; k < a.rowptr[row + 1]; ++k) {
        int32_t c = a.col ? a.col[k] : k;
        float w = 1.f;
        if (a.edge_w) w = a.edge_w[a.eperm ? a.eperm[k] : k];
        if (a.col_mode) w = apply_scale(w, a.col_scale[c], a.col_mode);
        acc += w * a.x[(int64_t)c * a.ldx + h];
    }
    if (a.row_mode) acc = apply_scale(acc, a.row_scale[row], a.row_mode);
    if (a.add1) acc += a.add1[row * a.ld1 + h];
    if (a.add2) acc += a.add2[row * a.ld2 + h];
    if (a.gadd) {
        const int32_t r = a.grow[row];
        float t = a.gadd[(int64_t)r * a.ldga + h];
        if (a.gmode) t = apply_scale(t, a.gscale[r], a.gmode);
        acc += t;
    }
    a.out[row * a.ldo + h] = acc;
}

template <bool GENERAL>
int launch_spmm_vec(const SpmmArgs& a, hipStream_t s) {
    int lpr = 1;
    while (lpr < a.H4) lpr <<= 1;
    int rpb = 4 * (64 / lpr);
    const dim3 grid((unsigned)bg_grid_for(a.n_rows, rpb, 0x7fffffff), (unsigned)(a.n_batch > 1 ? a.n_batch : 1));
    switch (lpr) {
        case 1: k_spmm_vec<1, GENERAL><<<grid, 256, 0, s>>>(a); break;
        case 2: k_spmm_vec<2, GENERAL><<<grid, 256, 0, s>>>(a); break;
        case 4: k_spmm_vec<4, GENERAL><<<grid, 256, 0, s>>>(a); break;
        case 8: k_spmm_vec<8, GENERAL><<<grid, 256, 0, s>>>(a); break;
        case 16: k_spmm_vec<16, GENERAL><<<grid, 256, 0, s>>>(a); break;
        case 32: k_spmm_vec<32, GENERAL><<<grid, 256, 0, s>>>(a); break;
        default: k_spmm_vec<64, GENERAL><<<grid, 256, 0, s>>>(a); break;
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int spmm_dispatch(SpmmArgs a, int64_t H, hipStream_t s) {
    if (a.n_rows == 0 || H == 0) return BG_OK;
    bool vec = (H % 4 == 0) && (a.ldx % 4 == 0) && (a.ldo % 4 == 0) && bg_aligned16(a.x) && bg_aligned16(a.out) &&
               (!a.add1 || (bg_aligned16(a.add1) && a.ld1 % 4 == 0)) && (!a.add2 || (bg_aligned16(a.add2) && a.ld2 % 4 == 0)) &&
               (!a.gadd || (bg_aligned16(a.gadd) && a.ldga % 4 == 0));
    if (!vec) {
        k_spmm_scalar<<<bg_grid_for(a.n_rows * H, 256, 0x7fffffff), 256, 0, s>>>(a, (int)H);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    const bool general = !a.col || a.edge_w || a.col_mode;
    {
        static const int64_t nt_bytes = (int64_t)bg_tune_int("BG_SPMM_NT_MB", 64) << 20;
        a.nt_store = nt_bytes > 0 && a.n_rows * H * 4 >= nt_bytes;
    }
    for (int64_t h0 = 0; h0 < H; h0 += 256) {   // panels of 64 float4 columns
        SpmmArgs p = a;
        p.x = a.x + h0; p.out = a.out + h0;
        if (a.add1) p.add1 = a.add1 + h0;
        if (a.add2) p.add2 = a.add2 + h0;
        if (a.gadd) p.gadd = a.gadd + h0;
        p.H4 = (int)((H - h0 < 256 ? H - h0 : 256) / 4);
        int rc;
        // large plain gather-sums: the batched form (the indices of two rows, then all their gathers, in flight at
        // once): 70.9 us (loop form) -> 57.2 us at 10^6 edges streaming from HBM.  Tuning builds: BG_SPMM_BATCH = 0
        // keeps the loop form, 1 / 4 select other row counts
        static const int batch_rows = bg_tune_int("BG_SPMM_BATCH", 2);
        if (!general && !p.gadd && p.H4 >= 16 && batch_rows > 0 && p.n_rows >= 4096)
            rc = batch_rows == 1 ? launch_spmm_batched<1, 8>(p, s)
                                 : (batch_rows == 2 ? launch_spmm_batched<2, 6>(p, s) : launch_spmm_batched<4, 6>(p, s));
        else
            rc = general ? launch_spmm_vec<true>(p, s) : launch_spmm_vec<false>(p, s);
        if (rc != BG_OK) return rc;
    }
    return BG_OK;
}

// ---------------------------------------------------------------- two-phase long segments
struct ChunkArgs {
    const float* x; int64_t ldx;
    const int32_t* rowptr; const int32_t* col;
    const float* edge_w; const int32_t* eperm;
    const int32_t* chunk_ptr; int32_t R; int64_t n_rows;
    float* partial; int32_t* partial_arg; int H;
    // direct mode (chunk_ptr == nullptr): one workgroup per ROW, final result written here
    float* out; int64_t ldo; int32_t* arg; const float* row_scale; int row_mode;
    // direct mode, blockIdx.y = b: the same reduction of x + b * x_step into out + b * out_step (arg + b * arg_step)
    int64_t x_step, out_step, arg_step;
    int accumulate;      // direct-mode sum: out += the segment's (scaled) sum, formed first and then added
};

// binary search: the row r with chunk_ptr[r] <= c < chunk_ptr[r+1]
__device__ __forceinline__ int64_t row_of_chunk(const int32_t* chunk_ptr, int64_t n_rows, int32_t c) {
    int64_t lo = 0, hi = n_rows;   // invariant: chunk_ptr[lo] <= c < chunk_ptr[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (chunk_ptr[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// One workgroup per chunk.  LPR lanes own one slot's row (a float4 each); the workgroup keeps
// 256/LPR slots in flight, each lane group walking its slots with stride 256/LPR; the groups'
// partial results are then combined through LDS in group order (deterministic; for max the
// smallest slot wins ties, which is the first-index rule).  T = 1024 serves a few long rows (the critic's pooling and
// the story sums of the generator's backward: 80 stories of ~150 voxels): four times the slots in flight per row,
// and the groups combine as a fixed tree (for the max the tie rule makes the order immaterial; a sum is deterministic
// but rounds in another order than the 256-thread form).
template <bool IS_MAX, int T = 256>
__global__ __launch_bounds__(T) void k_chunk_reduce(ChunkArgs a, int lpr, int hoff) {
    __shared__ float4 sval[T];
    __shared__ int4 sarg[IS_MAX ? T : 1];
    const int32_t c = blockIdx.x;
    const bool direct = a.chunk_ptr == nullptr;
    if (direct && blockIdx.y) {
        a.x += (int64_t)blockIdx.y * a.x_step;
        a.out += (int64_t)blockIdx.y * a.out_step;
        if (IS_MAX) a.arg += (int64_t)blockIdx.y * a.arg_step;
    }
    if (!direct && c >= a.chunk_ptr[a.n_rows]) return;
    const int64_t row = direct ? (int64_t)c : row_of_chunk(a.chunk_ptr, a.n_rows, c);
    const int32_t k0 = direct ? a.rowptr[row] : a.rowptr[row] + (c - a.chunk_ptr[row]) * a.R;
    const int32_t k1 = direct ? a.rowptr[row + 1] : min(k0 + a.R, a.rowptr[row + 1]);
    const int spb = T / lpr;
    const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const int H4 = (a.H - hoff) / 4 < 64 ? (a.H - hoff) / 4 : 64;
    const bool live = sub < H4;
    const float* xb = a.x + hoff + 4 * sub;
    const float lowest = -3.402823466e+38f;
    float4 acc = IS_MAX ? make_float4(lowest, lowest, lowest, lowest) : make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ak = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
    if (live) {
        int32_t k = k0 + grp;
        for (; k + 3 * spb < k1; k += 4 * spb) {
            int32_t ka = k, kb = k + spb, kc = k + 2 * spb, kd = k + 3 * spb;
            int32_t ca = a.col ? a.col[ka] : ka, cb = a.col ? a.col[kb] : kb;
            int32_t cc = a.col ? a.col[kc] : kc, cd = a.col ? a.col[kd] : kd;
            float4 va = bg_ld4(xb + (int64_t)ca * a.ldx), vb = bg_ld4(xb + (int64_t)cb * a.ldx);
            float4 vc = bg_ld4(xb + (int64_t)cc * a.ldx), vd = bg_ld4(xb + (int64_t)cd * a.ldx);
            if (IS_MAX) {
#define BG_UPD(v, kk)                                                    \
    if (v.x > acc.x) { acc.x = v.x; ak.x = kk; }                         \
    if (v.y > acc.y) { acc.y = v.y; ak.y = kk; }                         \
    if (v.z > acc.z) { acc.z = v.z; ak.z = kk; }                         \
    if (v.w > acc.w) { acc.w = v.w; ak.w = kk; }
                BG_UPD(va, ka) BG_UPD(vb, kb) BG_UPD(vc, kc) BG_UPD(vd, kd)
            } else if (a.edge_w) {
                float wa = a.edge_w[a.eperm ? a.eperm[ka] : ka], wb = a.edge_w[a.eperm ? a.eperm[kb] : kb];
                float wc = a.edge_w[a.eperm ? a.eperm[kc] : kc], wd = a.edge_w[a.eperm ? a.eperm[kd] : kd];
                acc = bg_add4(acc, bg_scale4(wa, va)); acc = bg_add4(acc, bg_scale4(wb, vb));
                acc = bg_add4(acc, bg_scale4(wc, vc)); acc = bg_add4(acc, bg_scale4(wd, vd));
            } else {
                acc = bg_add4(acc, va); acc = bg_add4(acc, vb); acc = bg_add4(acc, vc); acc = bg_add4(acc, vd);
            }
        }
        for (; k < k1; k += spb) {
            int32_t cc = a.col ? a.col[k] : k;
            float4 v = bg_ld4(xb + (int64_t)cc * a.ldx);
            if (IS_MAX) {
                BG_UPD(v, k)
            } else {
                float w = a.edge_w ? a.edge_w[a.eperm ? a.eperm[k] : k] : 1.f;
                acc = bg_add4(acc, bg_scale4(w, v));
            }
        }
    }
    sval[threadIdx.x] = acc;
    if (IS_MAX) sarg[threadIdx.x] = ak;
    __syncthreads();
#define BG_CMB(f)                                                                           \
    if (v.f > acc.f || (v.f == acc.f && kk.f < ak.f)) { acc.f = v.f; ak.f = kk.f; }
    if (T > 256) {                    // a fixed tree over the lane groups: deterministic, and order-free for the max
        for (int half = spb >> 1; half >= 1; half >>= 1) {
            if (grp < half) {
                float4 v = sval[(grp + half) * lpr + sub];
                if (IS_MAX) {
                    int4 kk = sarg[(grp + half) * lpr + sub];
                    BG_CMB(x) BG_CMB(y) BG_CMB(z) BG_CMB(w)
                    if (half > 1) sarg[threadIdx.x] = ak;
                } else {
                    acc = bg_add4(acc, v);
                }
                if (half > 1) sval[threadIdx.x] = acc;
            }
            if (half > 1) __syncthreads();
        }
    }
    if (grp == 0 && live) {
        for (int g = 1; g < spb && T <= 256; ++g) {
            float4 v = sval[g * lpr + sub];
            if (IS_MAX) {
                int4 kk = sarg[g * lpr + sub];
                BG_CMB(x) BG_CMB(y) BG_CMB(z) BG_CMB(w)
            } else {
                acc = bg_add4(acc, v);
            }
        }
        int4 r = make_int4(-1, -1, -1, -1);
        if (IS_MAX) {
            r.x = ak.x == 0x7fffffff ? -1 : (a.col ? a.col[ak.x] : ak.x);
            r.y = ak.y == 0x7fffffff ? -1 : (a.col ? a.col[ak.y] : ak.y);
            r.z = ak.z == 0x7fffffff ? -1 : (a.col ? a.col[ak.z] : ak.z);
            r.w = ak.w == 0x7fffffff ? -1 : (a.col ? a.col[ak.w] : ak.w);
        }
        if (direct) {
            if (IS_MAX) {                       // an empty segment gives 0 and arg -1
                acc.x = r.x >= 0 ? acc.x : 0.f; acc.y = r.y >= 0 ? acc.y : 0.f;
                acc.z = r.z >= 0 ? acc.z : 0.f; acc.w = r.w >= 0 ? acc.w : 0.f;
                *reinterpret_cast<int4*>(a.arg + row * a.H + hoff + 4 * sub) = r;
            } else if (a.row_mode) {
                const float sc = a.row_scale[row];
                acc.x = apply_scale(acc.x, sc, a.row_mode); acc.y = apply_scale(acc.y, sc, a.row_mode);
                acc.z = apply_scale(acc.z, sc, a.row_mode); acc.w = apply_scale(acc.w, sc, a.row_mode);
            }
            if (!IS_MAX && a.accumulate) acc = bg_add4(bg_ld4(a.out + row * a.ldo + hoff + 4 * sub), acc);
            bg_st4(a.out + row * a.ldo + hoff + 4 * sub, acc);
        } else {
            bg_st4(a.partial + (int64_t)c * a.H + hoff + 4 * sub, acc);
            if (IS_MAX) *reinterpret_cast<int4*>(a.partial_arg + (int64_t)c * a.H + hoff + 4 * sub) = r;
        }
    }
}

// scalar fallback of the chunk phase (any H / alignment): thread per channel, slots in order
template <bool IS_MAX>
__global__ __launch_bounds__(256) void k_chunk_reduce_scalar(ChunkArgs a) {
    const int32_t c = blockIdx.x;
    if (c >= a.chunk_ptr[a.n_rows]) return;
    const int64_t row = row_of_chunk(a.chunk_ptr, a.n_rows, c);
    const int32_t k0 = a.rowptr[row] + (c - a.chunk_ptr[row]) * a.R;
    const int32_t k1 = min(k0 + a.R, a.rowptr[row + 1]);
    for (int h = threadIdx.x; h < a.H; h += 256) {
        float acc = IS_MAX ? -3.402823466e+38f : 0.f;
        int32_t barg = -1;
        for (int32_t k = k0; k < k1; ++k) {
            int32_t cc = a.col ? a.col[k] : k;
            float v = a.x[(int64_t)cc * a.ldx + h];
            if (IS_MAX) {
                if (v > acc || barg < 0) { acc = v; barg = cc; }
            } else {
                float w = a.edge_w ? a.edge_w[a.eperm ? a.eperm[k] : k] : 1.f;
                acc += w * v;
            }
        }
        a.partial[(int64_t)c * a.H + h] = acc;
        if (IS_MAX) a.partial_arg[(int64_t)c * a.H + h] = barg;
    }
}

template <bool IS_MAX>
int launch_chunk_reduce(const ChunkArgs& a, int64_t n_chunks_ub, hipStream_t s, int64_t n_batch = 1) {
    bool vec = (a.H % 4 == 0) && (a.ldx % 4 == 0) && bg_aligned16(a.x);
    if (a.chunk_ptr) vec = vec && bg_aligned16(a.partial) && (!IS_MAX || bg_aligned16(a.partial_arg));
    else vec = vec && bg_aligned16(a.out) && (a.ldo % 4 == 0) && (!IS_MAX || bg_aligned16(a.arg));
    if (!vec && !a.chunk_ptr) return BG_ERR_ALIGN;      // the one-launch form has no scalar fallback
    if (!vec) {
        k_chunk_reduce_scalar<IS_MAX><<<(int)n_chunks_ub, 256, 0, s>>>(a);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    for (int hoff = 0; hoff < a.H; hoff += 256) {
        int h4 = (a.H - hoff) / 4 < 64 ? (a.H - hoff) / 4 : 64;
        int lpr = 1;
        while (lpr < h4) lpr <<= 1;
        const dim3 grid((unsigned)n_chunks_ub, (unsigned)n_batch);
        if (!a.chunk_ptr && n_chunks_ub * n_batch <= 512)      // a few rows (<= 2 workgroups per CU): the chip is mostly idle, so widen each row
            k_chunk_reduce<IS_MAX, 1024><<<grid, 1024, 0, s>>>(a, lpr, hoff);
        else
            k_chunk_reduce<IS_MAX><<<grid, 256, 0, s>>>(a, lpr, hoff);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

// phase 2 of segmax: combine a row's chunk partials in chunk order (strict > keeps the first)
__global__ void k_segmax_combine(const float* __restrict__ pval, const int32_t* __restrict__ parg,
                                 const int32_t* __restrict__ chunk_ptr, int64_t n_rows, int H,
                                 float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * H) return;
    int64_t row = t / H; int h = (int)(t % H);
    float best = -3.402823466e+38f; int32_t barg = -1;
    for (int32_t c = chunk_ptr[row]; c < chunk_ptr[row + 1]; ++c) {
        float v = pval[(int64_t)c * H + h]; int32_t ar = parg[(int64_t)c * H + h];
        if (ar >= 0 && v > best) { best = v; barg = ar; }
        else if (ar >= 0 && barg < 0) { best = v; barg = ar; }
    }
    out[row * ldo + h] = barg >= 0 ? best : 0.f;
    arg[row * H + h] = barg;
}

__global__ void k_gather_arg(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ arg,
                             float* __restrict__ out, int64_t ldo, int64_t n_seg, int H) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg * H) return;
    int64_t s = t / H; int h = (int)(t % H);
    int32_t a = arg[t];
    out[s * ldo + h] = a >= 0 ? x[(int64_t)a * ldx + h] : 0.f;
}

__global__ void k_scatter_arg(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ arg,
                              float* __restrict__ out, int64_t ldo, int64_t n_seg, int H) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg * H) return;
    int64_t s = t / H; int h = (int)(t % H);
    int32_t a = arg[t];
    if (a >= 0) out[(int64_t)a * ldo + h] = g[s * ldg + h];
}

// The same result written row by row when every row knows its segment (seg_of[r], -1 = none): no zero fill,
// no scatter -- out[r, h] = (arg[seg_of[r], h] == r) ? g[seg_of[r], h] : 0.
__global__ void k_scatter_arg_dense(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ arg,
                                    const int32_t* __restrict__ seg_of, float* __restrict__ out, int64_t ldo,
                                    int64_t n_rows, int H) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * H) return;
    int64_t r = t / H; int h = (int)(t % H);
    const int32_t s = seg_of[r];
    float v = 0.f;
    if (s >= 0 && arg[(int64_t)s * H + h] == (int32_t)r) v = g[(int64_t)s * ldg + h];
    out[r * ldo + h] = v;
}

// the same, four channels per thread (H % 4 == 0, 16-byte aligned rows, fewer than 2^31 elements)
__global__ __launch_bounds__(256) void k_scatter_arg_dense4(const float* __restrict__ g, int64_t ldg,
                                                            const int32_t* __restrict__ arg, const int32_t* __restrict__ seg_of,
                                                            float* __restrict__ out, int64_t ldo, uint32_t n4, uint32_t H4,
                                                            int64_t g_step = 0, int64_t arg_step = 0, int64_t out_step = 0) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n4) return;
    g += (int64_t)blockIdx.y * g_step; arg += (int64_t)blockIdx.y * arg_step; out += (int64_t)blockIdx.y * out_step;   // (batch)
    const uint32_t r = t / H4, h = 4u * (t - r * H4);
    const int32_t s = seg_of[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) {
        const int4 ar = *reinterpret_cast<const int4*>(arg + (int64_t)s * (4 * H4) + h);
        const float4 gv = bg_ld4(g + (int64_t)s * ldg + h);
        v.x = ar.x == (int32_t)r ? gv.x : 0.f; v.y = ar.y == (int32_t)r ? gv.y : 0.f;
        v.z = ar.z == (int32_t)r ? gv.z : 0.f; v.w = ar.w == (int32_t)r ? gv.w : 0.f;
    }
    bg_st4(out + (int64_t)r * ldo + h, v);
}

// two such scatters over the same rows in one launch (blockIdx.y picks the operands): the critic's forward sends the
// building decoder's row and the story decoder's row to their arg-max voxels side by side
struct ScatterPair { const float* g[2]; int64_t ldg[2]; const int32_t* arg[2]; const int32_t* seg_of[2]; float* out[2]; };
__global__ __launch_bounds__(256) void k_scatter_arg_dense4_pair(ScatterPair p, int64_t ldo, uint32_t n4, uint32_t H4) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n4) return;
    const int y = blockIdx.y;
    const uint32_t r = t / H4, h = 4u * (t - r * H4);
    const int32_t s = p.seg_of[y][r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) {
        const int4 ar = *reinterpret_cast<const int4*>(p.arg[y] + (int64_t)s * (4 * H4) + h);
        const float4 gv = bg_ld4(p.g[y] + (int64_t)s * p.ldg[y] + h);
        v.x = ar.x == (int32_t)r ? gv.x : 0.f; v.y = ar.y == (int32_t)r ? gv.y : 0.f;
        v.z = ar.z == (int32_t)r ? gv.z : 0.f; v.w = ar.w == (int32_t)r ? gv.w : 0.f;
    }
    bg_st4(p.out[y] + (int64_t)r * ldo + h, v);
}

// two gathers in one launch: threads [0, n0 H) serve the first, the rest the second
__global__ void k_gather_arg_pair(const float* __restrict__ x0, int64_t ldx0, const int32_t* __restrict__ arg0,
                                  float* __restrict__ out0, int64_t n0, const float* __restrict__ x1, int64_t ldx1,
                                  const int32_t* __restrict__ arg1, float* __restrict__ out1, int64_t n1, int64_t ldo, int H) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (n0 + n1) * H) return;
    const bool second = t >= n0 * H;
    if (second) t -= n0 * H;
    const float* x = second ? x1 : x0;
    const int64_t ldx = second ? ldx1 : ldx0;
    const int32_t* arg = second ? arg1 : arg0;
    float* out = second ? out1 : out0;
    int64_t s = t / H; int h = (int)(t % H);
    int32_t a = arg[t];
    out[s * ldo + h] = a >= 0 ? x[(int64_t)a * ldx + h] : 0.f;
}

// the batches of one row by one thread, and their sum (in batch order: (v_0 + v_1) + ..) written beside them
__global__ __launch_bounds__(256) void k_scatter_arg_dense4_sum(const float* __restrict__ g, int64_t ldg,
                                                                const int32_t* __restrict__ arg, const int32_t* __restrict__ seg_of,
                                                                float* __restrict__ out, int64_t ldo, uint32_t n4, uint32_t H4,
                                                                int64_t g_step, int64_t arg_step, int64_t out_step, int n_batch,
                                                                float* __restrict__ sum_out, int64_t ld_sum) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n4) return;
    const uint32_t r = t / H4, h = 4u * (t - r * H4);
    const int32_t s = seg_of[r];
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < n_batch; ++b) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s >= 0) {
            const int4 ar = *reinterpret_cast<const int4*>(arg + b * arg_step + (int64_t)s * (4 * H4) + h);
            const float4 gv = bg_ld4(g + b * g_step + (int64_t)s * ldg + h);
            v.x = ar.x == (int32_t)r ? gv.x : 0.f; v.y = ar.y == (int32_t)r ? gv.y : 0.f;
            v.z = ar.z == (int32_t)r ? gv.z : 0.f; v.w = ar.w == (int32_t)r ? gv.w : 0.f;
        }
        bg_st4(out + b * out_step + (int64_t)r * ldo + h, v);
        sum = b == 0 ? v : bg_add4(sum, v);
    }
    bg_st4(sum_out + (int64_t)r * ld_sum + h, sum);
}

// one wave per row; 64/LPR slots in flight, LPR lanes per dot product (float4 per lane per step)
template <int LPR>
__global__ __launch_bounds__(256) void k_sddmm(const float* __restrict__ A, int64_t lda,
                                               const float* __restrict__ B, int64_t ldb,
                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                               const int32_t* __restrict__ eperm, float* __restrict__ out,
                                               int64_t n_rows, int H4) {
    constexpr int SPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int sub = lane % LPR, grp = lane / LPR;
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    for (int32_t k0 = b; k0 < e; k0 += SPW) {
        int32_t k = k0 + grp;
        float acc = 0.f;
        if (k < e) {
            int32_t c = col ? col[k] : k;
            for (int j = sub; j < H4; j += LPR) {
                float4 u = bg_ld4(A + row * lda + 4 * j);
                float4 v = bg_ld4(B + (int64_t)c * ldb + 4 * j);
                acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc);
                acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (k < e && sub == 0) out[eperm ? eperm[k] : k] = acc;
    }
}

__global__ void k_sddmm_scalar(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                               const int32_t* __restrict__ eperm, float* __restrict__ out, int64_t n_rows, int H) {
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    for (int32_t k = rowptr[row]; k < rowptr[row + 1]; ++k) {
        int32_t c = col ? col[k] : k;
        float acc = 0.f;
        for (int h = 0; h < H; ++h) acc = fmaf(A[row * lda + h], B[(int64_t)c * ldb + h], acc);
        out[eperm ? eperm[k] : k] = acc;
    }
}

// Forward-only fusion for the generator block's story term (reference models.py:166-170): one workgroup
// per story averages the story's voxel rows (H <= 256 channels) and multiplies the mean by a [2H', H]
// weight held as W[o, 0:H] (first H' outputs) and W[o, H:2H] (second H' outputs) of a row-major
// [H', 2H] matrix -- i.e. out[s] = [ mean W[:, :H]^T | mean W[:, H:]^T ]: the story's own term and its
// contribution to the building term, without the [S, H] mean ever going back to memory.
__device__ __forceinline__ void story_project_body(const float* __restrict__ x, int64_t ldx,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ count,
                                                   const float* __restrict__ W, int64_t ldw, int Hout,
                                                   float* __restrict__ out, int64_t ldo, int H, int64_t row) {
    __shared__ float4 part[1024];
    // the story mean, 4 pad floats after every 16: the projection's P lane groups read segments H / P apart, which without
    // the skew start on one bank (P-way conflict on every read: SQ_LDS_BANK_CONFLICT was 67 % of this kernel's LDS cycles)
    __shared__ __attribute__((aligned(16))) float mean[256 + 64];
    auto mi = [](int h) { return h + 4 * (h >> 4); };
    const int nt = blockDim.x;                  // 256 or 1024 threads (a power of two)
    const int H4 = H / 4;                       // <= 64 lanes per slot row
    int lpr = 1;
    while (lpr < H4) lpr <<= 1;
    const int spb = nt / lpr, sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const int32_t k0 = rowptr[row], k1 = rowptr[row + 1];
    // the projection's weights do not depend on the mean: requested now (up to 8 float4 per thread), they arrive while
    // the mean is formed instead of adding a memory round trip after it
    const int P = nt / (2 * Hout);
    const bool fast = P >= 2 && (P & (P - 1)) == 0 && P <= 8 && P * 2 * Hout == nt && H % (4 * P) == 0;
    const int hs = fast ? H / P : 0;
    const bool pre = fast && hs <= 32;
    float4 wpre[8];
    if (pre) {
        const int o = threadIdx.x / P, pp = threadIdx.x % P;
        const float* w = W + (int64_t)(o < Hout ? o : o - Hout) * ldw + (o < Hout ? 0 : H) + pp * hs;
#pragma unroll
        for (int q = 0; q < 8; ++q) wpre[q] = 4 * q < hs ? bg_ld4(w + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < H4) {
        const float* xb = x + 4 * sub;
        int32_t k = k0 + grp;
        for (; k + 3 * spb < k1; k += 4 * spb) {
            const int32_t ca = col ? col[k] : k, cb = col ? col[k + spb] : k + spb;
            const int32_t cc = col ? col[k + 2 * spb] : k + 2 * spb, cd = col ? col[k + 3 * spb] : k + 3 * spb;
            const float4 va = bg_ld4(xb + (int64_t)ca * ldx), vb = bg_ld4(xb + (int64_t)cb * ldx);
            const float4 vc = bg_ld4(xb + (int64_t)cc * ldx), vd = bg_ld4(xb + (int64_t)cd * ldx);
            acc = bg_add4(acc, va); acc = bg_add4(acc, vb); acc = bg_add4(acc, vc); acc = bg_add4(acc, vd);
        }
        for (; k < k1; k += spb) acc = bg_add4(acc, bg_ld4(xb + (int64_t)(col ? col[k] : k) * ldx));
    }
    // the lane groups' partial sums, combined by a fixed binary tree (deterministic)
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int stride = spb >> 1; stride > 0; stride >>= 1) {
        if (grp < stride) {
            acc = bg_add4(acc, part[(grp + stride) * lpr + sub]);
            part[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (grp == 0 && sub < H4) {
        const float c = count[row];
        *reinterpret_cast<float4*>(mean + mi(4 * sub)) = make_float4(acc.x / c, acc.y / c, acc.z / c, acc.w / c);
    }
    __syncthreads();
    // projection: P = nt / (2 Hout) adjacent lanes share one output, each over a contiguous quarter (..) of H
    if (fast) {
        const int o = threadIdx.x / P, pp = threadIdx.x % P;
        const float* w = W + (int64_t)(o < Hout ? o : o - Hout) * ldw + (o < Hout ? 0 : H) + pp * hs;
        float d = 0.f;
        if (pre) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (4 * q < hs) {
                    const float4 wv = wpre[q];
                    const float4 mv = *reinterpret_cast<const float4*>(mean + mi(pp * hs + 4 * q));
                    d = fmaf(wv.x, mv.x, d); d = fmaf(wv.y, mv.y, d);
                    d = fmaf(wv.z, mv.z, d); d = fmaf(wv.w, mv.w, d);
                }
            }
        } else {
            for (int h = 0; h < hs; h += 4) {
                const float4 wv = bg_ld4(w + h);
                const float4 mv = *reinterpret_cast<const float4*>(mean + mi(pp * hs + h));
                d = fmaf(wv.x, mv.x, d); d = fmaf(wv.y, mv.y, d);
                d = fmaf(wv.z, mv.z, d); d = fmaf(wv.w, mv.w, d);
            }
        }
        d += bg_lane_xor<1>(d);                                   // P in {2, 4, 8}: the same pairs as the shuffle butterfly
        if (P >= 4) d += bg_lane_xor<2>(d);
        if (P >= 8) d += bg_lane_xor<4>(d);
        if (pp == 0) out[row * ldo + o] = d;
        return;
    }
    for (int o = threadIdx.x; o < 2 * Hout; o += nt) {
        const float* w = W + (int64_t)(o < Hout ? o : o - Hout) * ldw + (o < Hout ? 0 : H);
        float d = 0.f;
        for (int h = 0; h < H; h += 4) {
            const float4 wv = bg_ld4(w + h);
            const float4 mv = *reinterpret_cast<const float4*>(mean + mi(h));
            d = fmaf(wv.x, mv.x, d); d = fmaf(wv.y, mv.y, d);
            d = fmaf(wv.z, mv.z, d); d = fmaf(wv.w, mv.w, d);
        }
        out[row * ldo + o] = d;
    }
}

__global__ __launch_bounds__(1024) void k_story_project(const float* __restrict__ x, int64_t ldx,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ count,
                                                       const float* __restrict__ W, int64_t ldw, int Hout,
                                                       float* __restrict__ out, int64_t ldo, int H) {
    story_project_body(x, ldx, rowptr, col, count, W, ldw, Hout, out, ldo, H, blockIdx.x);
}

// Backward companion of the above for the generator block (ops.GBlock): one workgroup per story walks the
// story's voxels once and produces
//   gz[c]    = g[c] * (y[c] > 0 ? 1 : slope)      (the leaky ReLU's backward from its RESULT y, every voxel c)
//   gterm[s] = sum over the story's voxels of gz  (gradient of the per-story term gathered to the voxels)
//   story[s] = mean over the story's voxels of v  (the forward's story mean, recomputed instead of stored)
// in place of one elementwise launch and two segment reductions.  The segments must partition the voxels
// (every voxel in exactly one story) for gz to be complete.  Fixed binary tree over the lane groups.
__global__ __launch_bounds__(1024) void k_story_bwd(const float* __restrict__ g, int64_t ldg,
                                                   const float* __restrict__ y, int64_t ldy,
                                                   const float* __restrict__ v, int64_t ldv,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ count, float slope,
                                                   float* __restrict__ gz, int64_t ldz,
                                                   float* __restrict__ gterm, int64_t ldt,
                                                   float* __restrict__ story, int64_t lds, int H) {
    __shared__ float4 part_g[1024];
    __shared__ float4 part_v[1024];
    const int row = blockIdx.x;
    const int nt = blockDim.x;
    const int H4 = H / 4;
    int lpr = 1;
    while (lpr < H4) lpr <<= 1;
    const int spb = nt / lpr, sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
    const int32_t k0 = rowptr[row], k1 = rowptr[row + 1];
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), av = ag;
    if (sub < H4) {
        for (int32_t k = k0 + grp; k < k1; k += spb) {
            const int64_t c = col ? col[k] : k;
            const float4 gv = bg_ld4(g + c * ldg + 4 * sub), yv = bg_ld4(y + c * ldy + 4 * sub);
            const float4 vv = bg_ld4(v + c * ldv + 4 * sub);
            float4 z;
            z.x = yv.x > 0.f ? gv.x : gv.x * slope; z.y = yv.y > 0.f ? gv.y : gv.y * slope;
            z.z = yv.z > 0.f ? gv.z : gv.z * slope; z.w = yv.w > 0.f ? gv.w : gv.w * slope;
            *reinterpret_cast<float4*>(gz + c * ldz + 4 * sub) = z;
            ag = bg_add4(ag, z);
            av = bg_add4(av, vv);
        }
    }
    part_g[threadIdx.x] = ag;
    part_v[threadIdx.x] = av;
    __syncthreads();
    for (int stride = spb >> 1; stride > 0; stride >>= 1) {
        if (grp < stride) {
            ag = bg_add4(ag, part_g[(grp + stride) * lpr + sub]);
            av = bg_add4(av, part_v[(grp + stride) * lpr + sub]);
            part_g[threadIdx.x] = ag;
            part_v[threadIdx.x] = av;
        }
        __syncthreads();
    }
    if (grp == 0 && sub < H4) {
        const float c = count[row];
        *reinterpret_cast<float4*>(gterm + (int64_t)row * ldt + 4 * sub) = ag;
        *reinterpret_cast<float4*>(story + (int64_t)row * lds + 4 * sub) = make_float4(av.x / c, av.y / c, av.z / c, av.w / c);
    }
}

// Both reductions that a generator block takes of the SAME voxel features, in one launch (they are
// independent: the story workgroups -- the long pole -- are dealt first, the gather-mean rows fill the chip
// beside them).  n_story_blocks is padded to a multiple of 8 so the gather-mean's XCD-aware row order holds.
struct StoryArgs {
    const int32_t* rowptr; const int32_t* col; const float* count;
    const float* W; int64_t ldw; int Hout; float* out; int64_t ldo; int n_story; int n_story_blocks;
};

template <int LPR>
__global__ __launch_bounds__(1024) void k_gather_story(SpmmArgs a, StoryArgs st, int H) {
    if ((int)blockIdx.x < st.n_story_blocks) {
        if ((int)blockIdx.x < st.n_story)
            story_project_body(a.x, a.ldx, st.rowptr, st.col, st.count, st.W, st.ldw, st.Hout, st.out, st.ldo, H, blockIdx.x);
        return;
    }
    spmm_vec_body<LPR, false>(a, xcd_remap(blockIdx.x - st.n_story_blocks, gridDim.x - st.n_story_blocks));
}


// ---------------------------------------------------------------- operator chains
// n_steps successive applications of ONE block-diagonal sparse operator to a narrow matrix in a single
// launch:  y_k = rs * P (cs * y_{k-1}) + add_k,  k = 1..n_steps,  y_0 = x0.  A workgroup owns one diagonal
// block (the voxels of one graph: no edge leaves a graph) and one chunk of four columns; the block's
// rows ping-pong between two LDS images, the block's CSR (local column ids) and scales sit in LDS too,
// and a workgroup barrier separates the steps.  Same slot order and scale arithmetic as k_spmm_vec, so
// every step equals the corresponding separate launch.
// basis != 0 turns the chain into a three-term (Chebyshev) recurrence over the same two LDS images: the image a
// step overwrites still holds y_{k-2}, and each thread reads and writes only its own rows of it.
//   BG_CHAIN_CHEB     y_1 = P y_0,  y_k = 2 P y_{k-1} - y_{k-2}                      (y_k = T_k(P) y_0)
//   BG_CHAIN_CLENSHAW y_k = 2 P y_{k-1} - y_{k-2} + add_k, the last step with factor 1 (y_n = sum_j T_j(P) g_j for
//                     x0 = g_n, add_k = g_{n-k}: Clenshaw's evaluation, the transpose of the line above)
struct ChainArgs {
    const float* x0; int64_t ldx;
    const int32_t* rowptr; const int32_t* col;
    const float* row_scale; int row_mode;
    const float* col_scale; int col_mode;
    const int32_t* blk_ptr; int max_rows; int max_nnz;
    const float* add; int64_t ld_add; int64_t add_step;
    float* out; int64_t ldo; int64_t out_step;
    int n_steps;
    // optional second matrix under the same operator and step count (k_spmm_chain_short only): column chunks
    // blockIdx.y >= split_y belong to it (no addend)
    int split_y; const float* x0b; int64_t ldxb; float* outb; int64_t ldob; int64_t out_stepb;
    int basis;
    // column groups (bg_spmm_chain_grouped_f32): the W columns are groups of grp_w; group g of a matrix with a group
    // step starts g * step elements behind group 0 -- its own [rows, grp_w ..] matrix -- instead of g * grp_w columns
    // further right.  0 = that matrix is one plain [rows, W] matrix.
    int grp_w; int64_t x0_gstep, add_gstep, out_gstep;
};

// a workgroup's column chunk cc lies in one group: move the bases so that `base + row * ld + cc` lands in it
__device__ __forceinline__ void chain_group_bases(ChainArgs& a, int cc) {
    if (!a.grp_w) return;
    const int g = cc / a.grp_w;
    const int64_t back = (int64_t)g * a.grp_w;
    if (a.x0_gstep) a.x0 += g * a.x0_gstep - back;
    if (a.add && a.add_gstep) a.add += g * a.add_gstep - back;
    if (a.out_gstep) a.out += g * a.out_gstep - back;
}

// factor of P y_{k-1} at this step of a three-term chain (the subtraction of y_{k-2} starts at step 2)
__device__ __forceinline__ float chain_factor(int basis, int step, int n_steps) {
    return (basis == BG_CHAIN_CHEB ? step == 1 : step == n_steps) ? 1.f : 2.f;
}

template <bool IDX_LDS>
__global__ __launch_bounds__(1024) void k_spmm_chain(ChainArgs a) {
    extern __shared__ float4 chain_smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int32_t r0 = a.blk_ptr[blockIdx.x];
    const int32_t nr = a.blk_ptr[blockIdx.x + 1] - r0;
    const int cc = 4 * blockIdx.y;
    if (nr <= 0 || nr > a.max_rows) return;                 // uniform per workgroup (host checks the plan)
    chain_group_bases(a, cc);
    float4* buf0 = chain_smem;
    float4* buf1 = chain_smem + a.max_rows;
    float* s_rs = reinterpret_cast<float*>(chain_smem + 2 * (size_t)a.max_rows);
    float* s_cs = s_rs + a.max_rows;
    int32_t* s_rp = reinterpret_cast<int32_t*>(s_cs + a.max_rows);
    int32_t* s_col = s_rp + a.max_rows + 1;
    const int32_t e0 = a.rowptr[r0];
    const int32_t ne = a.rowptr[r0 + nr] - e0;
    if (IDX_LDS && ne > a.max_nnz) return;
    for (int r = tid; r < nr; r += nt) {
        buf0[r] = bg_ld4(a.x0 + (int64_t)(r0 + r) * a.ldx + cc);
        if (a.row_mode) s_rs[r] = a.row_scale[r0 + r];
        if (a.col_mode) s_cs[r] = a.col_scale[r0 + r];
    }
    if (IDX_LDS) {
        for (int r = tid; r <= nr; r += nt) s_rp[r] = a.rowptr[r0 + r] - e0;
        for (int k = tid; k < ne; k += nt) s_col[k] = a.col[e0 + k] - r0;
    }
    __syncthreads();
    for (int step = 1; step <= a.n_steps; ++step) {
        const float4* src = (step & 1) ? buf0 : buf1;
        float4* dst = (step & 1) ? buf1 : buf0;
        const float* addk = a.add ? a.add + (int64_t)(step - 1) * a.add_step : nullptr;
        float* outk = (a.out_step != 0 || step == a.n_steps) ? a.out + (int64_t)(step - 1) * a.out_step : nullptr;
        const float fac = chain_factor(a.basis, step, a.n_steps);
        for (int r = tid; r < nr; r += nt) {
            const int64_t grow = (int64_t)r0 + r;
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
            if (addk) ad = bg_ld4(addk + grow * a.ld_add + cc);
            int32_t kb, ke;
            if (IDX_LDS) { kb = s_rp[r]; ke = s_rp[r + 1]; }
            else { kb = a.rowptr[grow]; ke = a.rowptr[grow + 1]; }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int32_t k = kb; k < ke; ++k) {
                const int32_t c = IDX_LDS ? s_col[k] : a.col[k] - r0;
                if ((uint32_t)c >= (uint32_t)nr) continue;      // never taken on a checked plan
                const float4 v = src[c];
                if (a.col_mode) {
                    float w = 1.f;
                    w = apply_scale(w, s_cs[c], a.col_mode);
                    acc = bg_fma4(w, v, acc);
                } else {
                    acc = bg_add4(acc, v);
                }
            }
            if (a.row_mode) {
                const float s = s_rs[r];
                acc.x = apply_scale(acc.x, s, a.row_mode); acc.y = apply_scale(acc.y, s, a.row_mode);
                acc.z = apply_scale(acc.z, s, a.row_mode); acc.w = apply_scale(acc.w, s, a.row_mode);
            }
            if (a.basis) {
                acc.x *= fac; acc.y *= fac; acc.z *= fac; acc.w *= fac;
                if (step >= 2) { const float4 p = dst[r]; acc.x -= p.x; acc.y -= p.y; acc.z -= p.z; acc.w -= p.w; }
            }
            if (addk) acc = bg_add4(acc, ad);
            dst[r] = acc;
            if (outk) bg_st4(outk + grow * a.ldo + cc, acc);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- program ratio (forward only, two launches)
// program_weight[p] = sum over the cross edges e of program node p of att[e] * (mask[v_e] * area[v_e]): a workgroup per
// program node, threads stride over its edges, partial sums combined in a fixed tree (lane, wave, workgroup)
__global__ __launch_bounds__(256) void k_program_weight(const float* __restrict__ att, const float* __restrict__ mask,
                                                        const float* __restrict__ area, int64_t ld_area,
                                                        const int32_t* __restrict__ voxel_of_edge,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ edge,
                                                        float* __restrict__ program_weight) {
    __shared__ float part[4];
    const int p = blockIdx.x;
    float acc = 0.f;
    for (int32_t k = rowptr[p] + threadIdx.x; k < rowptr[p + 1]; k += 256) {
        const int32_t e = edge[k];
        const int32_t v = voxel_of_edge[e];
        acc += att[e] * (mask[v] * area[(int64_t)v * ld_area]);
    }
    acc = bg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) program_weight[p] = (part[0] + part[1]) + (part[2] + part[3]);
}

// class slots, graphs, ratio: one workgroup (the slots are 6 per graph).  Sums in CSR slot order, as bg_spmm_f32 forms them.
__global__ __launch_bounds__(1024) void k_program_ratio_finish(const float* __restrict__ program_weight,
                                                               const int32_t* __restrict__ slot_rowptr,
                                                               const int32_t* __restrict__ slot_col, int n_slot,
                                                               const int32_t* __restrict__ graph_rowptr,
                                                               const int32_t* __restrict__ graph_col, int B,
                                                               const int32_t* __restrict__ graph_of_slot,
                                                               float* __restrict__ ratio, float* __restrict__ far) {
    extern __shared__ float pr_smem[];                  // class weight [n_slot] | far [B]
    float* cw = pr_smem;
    float* fs = pr_smem + n_slot;
    for (int s = threadIdx.x; s < n_slot; s += blockDim.x) {
        float acc = 0.f;
        for (int32_t k = slot_rowptr[s]; k < slot_rowptr[s + 1]; ++k) acc += program_weight[slot_col[k]];
        cw[s] = acc;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float acc = 0.f;
        for (int32_t k = graph_rowptr[b]; k < graph_rowptr[b + 1]; ++k) acc += cw[graph_col[k]];
        fs[b] = acc;
        far[b] = acc;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slot; s += blockDim.x) ratio[s] = cw[s] / (fs[graph_of_slot[s]] + 1e-16f);
}

constexpr int BG_SEG_MAX_DEVICES = 64;
constexpr size_t BG_CHAIN_LDS_LIMIT = 160 * 1024;

size_t chain_lds_bytes(int64_t max_rows, int64_t max_nnz, bool idx) {
    size_t b = (size_t)max_rows * (2 * 16 + 4 + 4) + (size_t)(max_rows + 1) * 4;
    if (idx) b += (size_t)max_nnz * 4;
    return (b + 15) & ~(size_t)15;
}

template <bool IDX_LDS>
int launch_chain(const ChainArgs& a, int64_t n_blk, int64_t W, size_t lds, hipStream_t s) {
    auto kern = k_spmm_chain<IDX_LDS>;
    if (lds > 65536) {          // above the static limit: opt in once per device
        static bool raised[BG_SEG_MAX_DEVICES] = {};
        int dev = 0;
        BG_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= BG_SEG_MAX_DEVICES) return BG_ERR_ARG;
        if (!raised[dev]) {
            BG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)BG_CHAIN_LDS_LIMIT));
            raised[dev] = true;
        }
    }
    const int nt = a.max_rows > 512 ? 1024 : (a.max_rows > 256 ? 512 : 256);
    dim3 grid((unsigned)n_blk, (unsigned)(W / 4));
    kern<<<grid, nt, lds, s>>>(a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// Step loop of k_spmm_chain_short.  KIND: 1 = rows divided by rs, unit column weights (the mean-aggregation operator);
// 2 = column weights only (its transpose); 0 = whatever the arguments say.  Same arithmetic in every KIND.
template <int RPT, int DEG, bool HAS_ADD, int KIND>
__device__ __forceinline__ void chain_short_steps(const ChainArgs& a, float4* buf0, float4* buf1,
                                                  const int32_t (&cols)[RPT][DEG], const float (&wts)[RPT][DEG],
                                                  const float (&rs)[RPT], int nr, int32_t r0, int cc, int tid, int nt) {
    const bool use_w = KIND == 2 || (KIND == 0 && a.col_mode);
    float4 ad_next[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        ad_next[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = tid + i * nt;
        if (HAS_ADD && r < nr) ad_next[i] = bg_ld4(a.add + ((int64_t)r0 + r) * a.ld_add + cc);
    }
    for (int step = 1; step <= a.n_steps; ++step) {
        const float4* src = (step & 1) ? buf0 : buf1;
        float4* dst = (step & 1) ? buf1 : buf0;
        float* outk = (a.out_step != 0 || step == a.n_steps) ? a.out + (int64_t)(step - 1) * a.out_step : nullptr;
        const float fac = chain_factor(a.basis, step, a.n_steps);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = tid + i * nt;
            if (r < nr) {
                const int64_t grow = (int64_t)r0 + r;
                const float4 ad = ad_next[i];
                if (HAS_ADD && step < a.n_steps)       // the next step's addend travels while this step computes
                    ad_next[i] = bg_ld4(a.add + (int64_t)step * a.add_step + grow * a.ld_add + cc);
                float4 v[DEG];
#pragma unroll
                for (int j = 0; j < DEG; ++j) v[j] = src[cols[i][j]];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (use_w) {
#pragma unroll
                    for (int j = 0; j < DEG; ++j) acc = bg_fma4(wts[i][j], v[j], acc);
                } else {
#pragma unroll
                    for (int j = 0; j < DEG; ++j) acc = bg_add4(acc, v[j]);
                }
                if (KIND == 1) {
                    acc.x = acc.x / rs[i]; acc.y = acc.y / rs[i]; acc.z = acc.z / rs[i]; acc.w = acc.w / rs[i];
                } else if (KIND == 0 && a.row_mode) {
                    acc.x = apply_scale(acc.x, rs[i], a.row_mode); acc.y = apply_scale(acc.y, rs[i], a.row_mode);
                    acc.z = apply_scale(acc.z, rs[i], a.row_mode); acc.w = apply_scale(acc.w, rs[i], a.row_mode);
                }
                if (a.basis) {
                    acc.x *= fac; acc.y *= fac; acc.z *= fac; acc.w *= fac;
                    if (step >= 2) { const float4 p = dst[r]; acc.x -= p.x; acc.y -= p.y; acc.z -= p.z; acc.w -= p.w; }
                }
                if (HAS_ADD) acc = bg_add4(acc, ad);
                dst[r] = acc;
                if (outk) bg_st4(outk + grow * a.ldo + cc, acc);
            }
        }
        __syncthreads();
    }
}

// Fast form for short rows (every row of the operator has at most DEG slots, as on voxel grids: six
// neighbours): each thread keeps the local column ids (and column weights) of its RPT rows in registers,
// rows shorter than DEG are padded with a slot that points at an all-zero LDS row (x + 0 = x exactly),
// so a step is RPT x DEG independent LDS reads per thread, straight-line code and one barrier; no index
// tables in LDS.
template <int RPT, int DEG>
__global__ __launch_bounds__(1024) void k_spmm_chain_short(ChainArgs a) {
    extern __shared__ float4 chain_smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int32_t r0 = a.blk_ptr[blockIdx.x];
    const int32_t nr = a.blk_ptr[blockIdx.x + 1] - r0;
    int cc = 4 * blockIdx.y;
    if (nr <= 0 || nr > a.max_rows || nr > RPT * nt) return;   // uniform per workgroup (host checks the plan)
    if (a.x0b && (int)blockIdx.y >= a.split_y) {                // uniform per workgroup: the second matrix
        cc = 4 * ((int)blockIdx.y - a.split_y);
        a.x0 = a.x0b; a.ldx = a.ldxb; a.out = a.outb; a.ldo = a.ldob; a.out_step = a.out_stepb; a.add = nullptr;
    } else {
        chain_group_bases(a, cc);                               // (never together with a second matrix)
    }
    const int zero_slot = a.max_rows;                           // both images carry one extra, all-zero row
    float4* buf0 = chain_smem;
    float4* buf1 = chain_smem + (a.max_rows + 1);
    int32_t cols[RPT][DEG];
    float wts[RPT][DEG];
    float rs[RPT];
    if (tid == 0) {
        buf0[zero_slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        buf1[zero_slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + i * nt;
        rs[i] = 1.f;
#pragma unroll
        for (int j = 0; j < DEG; ++j) { cols[i][j] = zero_slot; wts[i][j] = 0.f; }
        if (r < nr) {
            const int64_t grow = (int64_t)r0 + r;
            buf0[r] = bg_ld4(a.x0 + grow * a.ldx + cc);
            const int32_t kb = a.rowptr[grow];
            const int d = a.rowptr[grow + 1] - kb;              // <= DEG on a checked plan (max_deg)
            if (a.row_mode) rs[i] = a.row_scale[grow];
#pragma unroll
            for (int j = 0; j < DEG; ++j) {
                if (j < d) {
                    const int32_t c = a.col[kb + j] - r0;
                    if ((uint32_t)c < (uint32_t)nr) {           // always, on a checked plan
                        cols[i][j] = c;
                        if (a.col_mode) wts[i][j] = apply_scale(1.f, a.col_scale[r0 + c], a.col_mode);
                    }
                }
            }
        }
    }
    __syncthreads();
    // the step loop, specialised on what is uniform for the launch (no runtime mode tests, and no load -- hence no
    // wait on the previous step's global stores, which share the load counter -- in a chain without addends)
    const bool row_div_only = a.row_mode == BG_SCALE_DIV && !a.col_mode;
    const bool col_only = !a.row_mode && a.col_mode;
    if (a.add) {
        if (row_div_only) chain_short_steps<RPT, DEG, true, 1>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
        else if (col_only) chain_short_steps<RPT, DEG, true, 2>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
        else chain_short_steps<RPT, DEG, true, 0>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
    } else {
        if (row_div_only) chain_short_steps<RPT, DEG, false, 1>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
        else if (col_only) chain_short_steps<RPT, DEG, false, 2>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
        else chain_short_steps<RPT, DEG, false, 0>(a, buf0, buf1, cols, wts, rs, nr, r0, cc, tid, nt);
    }
}

template <int RPT, int DEG>
int launch_chain_short(const ChainArgs& a, int64_t n_blk, int64_t W, int nt, hipStream_t s) {
    auto kern = k_spmm_chain_short<RPT, DEG>;
    const size_t lds = (size_t)(a.max_rows + 1) * 32;
    if (lds > 65536) {
        static bool raised[BG_SEG_MAX_DEVICES] = {};
        int dev = 0;
        BG_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= BG_SEG_MAX_DEVICES) return BG_ERR_ARG;
        if (!raised[dev]) {
            BG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)BG_CHAIN_LDS_LIMIT));
            raised[dev] = true;
        }
    }
    dim3 grid((unsigned)n_blk, (unsigned)(W / 4));
    kern<<<grid, nt, lds, s>>>(a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

template <int DEG>
int dispatch_chain_short(const ChainArgs& a, int64_t n_blk, int64_t W, hipStream_t s) {
    if (a.max_rows <= 256) return launch_chain_short<1, DEG>(a, n_blk, W, 256, s);
    if (a.max_rows <= 512) return launch_chain_short<1, DEG>(a, n_blk, W, 512, s);
    if (a.max_rows <= 1024) return launch_chain_short<1, DEG>(a, n_blk, W, 1024, s);
    if (a.max_rows <= 2048) return launch_chain_short<2, DEG>(a, n_blk, W, 1024, s);
    if constexpr (DEG <= 6) return launch_chain_short<3, DEG>(a, n_blk, W, 1024, s);   // 8 slots x 3 rows spill
    return BG_ERR_SHAPE;
}

}  // namespace

extern "C" {

int bg_spmm_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                const float* edge_w, const int32_t* eperm,
                const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || ldx < H || ldo < H) return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (col_mode && !col_scale) || (eperm && !edge_w)) return BG_ERR_ARG;
    SpmmArgs a{x, ldx, rowptr, col, edge_w, eperm, row_scale, row_mode, col_scale, col_mode, out, ldo, n_rows, 0,
               nullptr, 0, nullptr, 0};
    return spmm_dispatch(a, H, (hipStream_t)stream);
}

// n_batch matrices under ONE operator in one launch: batch b reads x + b * x_step and writes out + b * out_step
// (the generator stack's backward applies the story -> building mean to every block call's story term at once).
// The 16-byte row form only (H % 4 == 0, H <= 256, aligned operands); other shapes run bg_spmm_f32 per batch.
int bg_spmm_batch_f32(const float* x, int64_t ldx, int64_t x_step, const int32_t* rowptr, const int32_t* col,
                      const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                      float* out, int64_t ldo, int64_t out_step, int64_t n_rows, int64_t H, int64_t n_batch, bgStream stream) {
    if (n_rows < 0 || H < 0 || n_batch < 0 || n_batch > 65535) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0 || n_batch == 0) return BG_OK;
    if (!x || !rowptr || !out || ldx < H || ldo < H) return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (col_mode && !col_scale)) return BG_ERR_ARG;
    const bool vec = H % 4 == 0 && H <= 256 && ldx % 4 == 0 && ldo % 4 == 0 && x_step % 4 == 0 && out_step % 4 == 0 &&
                     bg_aligned16(x) && bg_aligned16(out);
    if (!vec || n_batch == 1) {
        for (int64_t b = 0; b < n_batch; ++b) {
            int rc = bg_spmm_f32(x + b * x_step, ldx, rowptr, col, nullptr, nullptr, row_scale, row_mode, col_scale, col_mode,
                                 out + b * out_step, ldo, n_rows, H, stream);
            if (rc != BG_OK) return rc;
        }
        return BG_OK;
    }
    SpmmArgs a{x, ldx, rowptr, col, nullptr, nullptr, row_scale, row_mode, col_scale, col_mode, out, ldo, n_rows, (int)(H / 4),
               nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, x_step, out_step, (int)n_batch};
    const bool general = !col || col_mode;
    return general ? launch_spmm_vec<true>(a, (hipStream_t)stream) : launch_spmm_vec<false>(a, (hipStream_t)stream);
}

// get_program_ratio without a gradient (evaluation; the generator step when its ratio losses have zero weight) in two
// launches instead of nine: see include/bgan.h
int bg_program_ratio_f32(const float* att, const float* mask, const float* area, int64_t ld_area,
                         const int32_t* voxel_of_edge, const int32_t* prog_rowptr, const int32_t* prog_edge,
                         const int32_t* slot_rowptr, const int32_t* slot_col, const int32_t* graph_rowptr,
                         const int32_t* graph_col, const int32_t* graph_of_slot, float* ratio, float* program_weight,
                         float* far, int64_t Np, int64_t n_slot, int64_t B, bgStream stream) {
    if (Np < 0 || n_slot < 0 || B < 0 || ld_area < 1) return BG_ERR_ARG;
    if (Np == 0 || n_slot == 0 || B == 0) return BG_OK;
    if (!att || !mask || !area || !voxel_of_edge || !prog_rowptr || !prog_edge || !slot_rowptr || !slot_col || !graph_rowptr ||
        !graph_col || !graph_of_slot || !ratio || !program_weight || !far) return BG_ERR_ARG;
    if (Np > 0x7fffffff || n_slot + B > 12000) return BG_ERR_SHAPE;          // class weights and graph sums live in LDS
    hipStream_t s = (hipStream_t)stream;
    k_program_weight<<<(unsigned)Np, 256, 0, s>>>(att, mask, area, ld_area, voxel_of_edge, prog_rowptr, prog_edge, program_weight);
    BG_LAUNCH_CHECK();
    k_program_ratio_finish<<<1, 1024, (size_t)(n_slot + B) * sizeof(float), s>>>(program_weight, slot_rowptr, slot_col, (int)n_slot,
                                                                               graph_rowptr, graph_col, (int)B, graph_of_slot,
                                                                               ratio, far);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_spmm_add_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                    const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                    const float* add1, int64_t ld1, const float* add2, int64_t ld2,
                    const float* gadd, int64_t ldga, const int32_t* grow, const float* gscale, int gmode,
                    float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || ldx < H || ldo < H) return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (col_mode && !col_scale)) return BG_ERR_ARG;
    if ((add1 && ld1 < H) || (add2 && ld2 < H)) return BG_ERR_ARG;
    if (gadd && (!grow || ldga < H || (gmode && !gscale))) return BG_ERR_ARG;
    SpmmArgs a{x, ldx, rowptr, col, nullptr, nullptr, row_scale, row_mode, col_scale, col_mode, out, ldo, n_rows, 0,
               add1, ld1, add2, ld2, 0, gadd, ldga, grow, gscale, gmode};
    return spmm_dispatch(a, H, (hipStream_t)stream);
}

int bg_spmm_chain_grouped_f32(const float* x0, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                              const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                              const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                              const float* add, int64_t ld_add, int64_t add_step,
                              float* out, int64_t ldo, int64_t out_step, int64_t n_steps, int64_t W, int32_t basis,
                              int64_t grp_w, int64_t x0_gstep, int64_t add_gstep, int64_t out_gstep, bgStream stream) {
    if (n_blk < 0 || W < 0 || n_steps < 0 || max_rows < 0 || max_nnz < 0 || max_deg < 0 || grp_w < 0) return BG_ERR_ARG;
    if (basis != BG_CHAIN_POWER && basis != BG_CHAIN_CHEB && basis != BG_CHAIN_CLENSHAW) return BG_ERR_ARG;
    if (n_blk == 0 || W == 0 || n_steps == 0 || max_rows == 0) return BG_OK;
    if (grp_w == 0 && (x0_gstep || add_gstep || out_gstep)) return BG_ERR_ARG;
    if (grp_w > 0 && (grp_w % 4 != 0 || W % grp_w != 0 || grp_w > (1 << 20))) return BG_ERR_SHAPE;
    // a matrix with a group step holds grp_w columns per group, a plain one all W
    const int64_t wx = x0_gstep ? grp_w : W, wa = add_gstep ? grp_w : W, wo = out_gstep ? grp_w : W;
    if (!x0 || !rowptr || !col || !blk_ptr || !out || ldx < wx || ldo < wo || (add && ld_add < wa)) return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (col_mode && !col_scale)) return BG_ERR_ARG;
    if (W % 4 != 0 || n_blk > 65535 || W / 4 > 65535 || n_steps > (1 << 20)) return BG_ERR_SHAPE;
    if (!bg_aligned16(x0) || !bg_aligned16(out) || ldx % 4 != 0 || ldo % 4 != 0 || out_step % 4 != 0 ||
        (add && (!bg_aligned16(add) || ld_add % 4 != 0 || add_step % 4 != 0)) || x0_gstep % 4 != 0 || add_gstep % 4 != 0 ||
        out_gstep % 4 != 0) return BG_ERR_ALIGN;
    if (max_rows > (1 << 20) || max_nnz > (1 << 26)) return BG_ERR_SHAPE;
    ChainArgs a{x0, ldx, rowptr, col, row_scale, row_mode, col_scale, col_mode, blk_ptr, (int)max_rows, (int)max_nnz,
                add, ld_add, add_step, out, ldo, out_step, (int)n_steps, 0, nullptr, 0, nullptr, 0, 0, (int)basis,
                (int)grp_w, x0_gstep, add_gstep, out_gstep};
    hipStream_t s = (hipStream_t)stream;
    if (max_deg <= 8 && max_rows <= (max_deg <= 6 ? 3072 : 2048)) {
        // short rows: indices in registers, 1 to 3 rows per thread (4 rows per thread spills, 3 do with 8 slots)
        return max_deg <= 6 ? dispatch_chain_short<6>(a, n_blk, W, s) : dispatch_chain_short<8>(a, n_blk, W, s);
    }
    size_t lds = chain_lds_bytes(max_rows, max_nnz, true);
    if (lds <= BG_CHAIN_LDS_LIMIT) return launch_chain<true>(a, n_blk, W, lds, s);
    lds = chain_lds_bytes(max_rows, max_nnz, false);
    if (lds <= BG_CHAIN_LDS_LIMIT) return launch_chain<false>(a, n_blk, W, lds, s);
    return BG_ERR_SHAPE;          // a diagonal block does not fit in LDS: run the steps as separate launches
}

int bg_spmm_chain_f32(const float* x0, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                      const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                      const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                      const float* add, int64_t ld_add, int64_t add_step,
                      float* out, int64_t ldo, int64_t out_step, int64_t n_steps, int64_t W, int32_t basis,
                      bgStream stream) {
    return bg_spmm_chain_grouped_f32(x0, ldx, rowptr, col, row_scale, row_mode, col_scale, col_mode, blk_ptr, n_blk, max_rows,
                                     max_nnz, max_deg, add, ld_add, add_step, out, ldo, out_step, n_steps, W, basis, 0, 0, 0, 0,
                                     stream);
}

int bg_spmm_chain_pair_f32(const float* x0, int64_t ldx, float* out, int64_t ldo, int64_t out_step, int64_t W,
                           const float* x0b, int64_t ldxb, float* outb, int64_t ldob, int64_t out_stepb, int64_t Wb,
                           const int32_t* rowptr, const int32_t* col,
                           const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                           const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                           int64_t n_steps, int32_t basis, bgStream stream) {
    if (n_blk < 0 || W <= 0 || Wb <= 0 || n_steps < 0 || max_rows < 0 || max_nnz < 0 || max_deg < 0) return BG_ERR_ARG;
    if (basis != BG_CHAIN_POWER && basis != BG_CHAIN_CHEB) return BG_ERR_ARG;      // no addend here: no Clenshaw form
    if (n_blk == 0 || n_steps == 0 || max_rows == 0) return BG_OK;
    if (!x0 || !out || !x0b || !outb || !rowptr || !col || !blk_ptr || ldx < W || ldo < W || ldxb < Wb || ldob < Wb)
        return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (col_mode && !col_scale)) return BG_ERR_ARG;
    if (W % 4 != 0 || Wb % 4 != 0 || n_blk > 65535 || (W + Wb) / 4 > 65535 || n_steps > (1 << 20)) return BG_ERR_SHAPE;
    if (!bg_aligned16(x0) || !bg_aligned16(out) || !bg_aligned16(x0b) || !bg_aligned16(outb) || ldx % 4 != 0 ||
        ldo % 4 != 0 || out_step % 4 != 0 || ldxb % 4 != 0 || ldob % 4 != 0 || out_stepb % 4 != 0) return BG_ERR_ALIGN;
    if (!(max_deg <= 8 && max_rows <= (max_deg <= 6 ? 3072 : 2048))) return BG_ERR_SHAPE;      // short-row plans only: else two bg_spmm_chain_f32 calls
    ChainArgs a{x0, ldx, rowptr, col, row_scale, row_mode, col_scale, col_mode, blk_ptr, (int)max_rows, (int)max_nnz,
                nullptr, 0, 0, out, ldo, out_step, (int)n_steps, (int)(W / 4), x0b, ldxb, outb, ldob, out_stepb, (int)basis};
    hipStream_t s = (hipStream_t)stream;
    return max_deg <= 6 ? dispatch_chain_short<6>(a, n_blk, W + Wb, s) : dispatch_chain_short<8>(a, n_blk, W + Wb, s);
}

int bg_segreduce_sum_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const float* edge_w, const int32_t* eperm,
                         const float* row_scale, int row_mode,
                         const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                         float* partial, float* out, int64_t ldo, int64_t n_rows, int64_t H,
                         bgStream stream) {
    if (n_rows < 0 || H < 0 || n_chunks_ub < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || ldx < H || ldo < H) return BG_ERR_ARG;
    if ((row_mode && !row_scale) || (eperm && !edge_w)) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!chunk_ptr) {      // one launch: a workgroup per row writes the scaled sum (rows of moderate length)
        ChunkArgs a{x, ldx, rowptr, col, edge_w, eperm, nullptr, 0, n_rows, nullptr, nullptr, (int)H,
                    out, ldo, nullptr, row_scale, row_mode};
        return launch_chunk_reduce<false>(a, n_rows, s);
    }
    if (rows_per_chunk <= 0 || (n_chunks_ub > 0 && !partial)) return BG_ERR_ARG;
    if (n_chunks_ub > 0) {
        ChunkArgs a{x, ldx, rowptr, col, edge_w, eperm, chunk_ptr, rows_per_chunk, n_rows, partial, nullptr, (int)H,
                    nullptr, 0, nullptr, nullptr, 0};
        int rc = launch_chunk_reduce<false>(a, n_chunks_ub, s);
        if (rc != BG_OK) return rc;
    }
    // phase 2: out[r,:] = rs(r) * sum of the row's chunk partials, in chunk order
    SpmmArgs p{partial, H, chunk_ptr, nullptr, nullptr, nullptr, row_scale, row_mode, nullptr, 0, out, ldo, n_rows, 0,
               nullptr, 0, nullptr, 0};
    return spmm_dispatch(p, H, s);
}

// out += the segments' sums (bg_segreduce_sum_f32's one-launch form, no row scale): a segment's sum is formed in the
// same order as there and then added to what out holds -- the bits of that call followed by an elementwise add
int bg_segsum_acc_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, const float* edge_w,
                      const int32_t* eperm, float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || ldx < H || ldo < H || (eperm && !edge_w)) return BG_ERR_ARG;
    ChunkArgs a{x, ldx, rowptr, col, edge_w, eperm, nullptr, 0, n_rows, nullptr, nullptr, (int)H,
                out, ldo, nullptr, nullptr, 0, 0, 0, 0, 1};
    return launch_chunk_reduce<false>(a, n_rows, (hipStream_t)stream);
}

int bg_segmax_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                  const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                  float* partial_val, int32_t* partial_arg,
                  float* out, int64_t ldo, int32_t* arg, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0 || n_chunks_ub < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || !arg || ldx < H || ldo < H) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!chunk_ptr) {
        ChunkArgs a{x, ldx, rowptr, col, nullptr, nullptr, nullptr, 0, n_rows, nullptr, nullptr, (int)H,
                    out, ldo, arg, nullptr, 0};
        return launch_chunk_reduce<true>(a, n_rows, s);
    }
    if (rows_per_chunk <= 0 || (n_chunks_ub > 0 && (!partial_val || !partial_arg))) return BG_ERR_ARG;
    if (n_chunks_ub > 0) {
        ChunkArgs a{x, ldx, rowptr, col, nullptr, nullptr, chunk_ptr, rows_per_chunk, n_rows, partial_val, partial_arg, (int)H,
                    nullptr, 0, nullptr, nullptr, 0};
        int rc = launch_chunk_reduce<true>(a, n_chunks_ub, s);
        if (rc != BG_OK) return rc;
    }
    k_segmax_combine<<<bg_grid_for(n_rows * H, 256, 0x7fffffff), 256, 0, s>>>(partial_val, partial_arg, chunk_ptr,
                                                                                n_rows, (int)H, out, ldo, arg);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// ---------------------------------------------------------------- segment max without a workspace argument
// The chunk partials of bg_segmax_f32's two-phase form in memory the library owns: one array per GPU (every
// device that loads this code object holds its own copy at its own address).  In HOST code the array's name is
// only a handle for hipGetSymbolAddress -- never a device address -- so the kernels get the looked-up pointer.
static __device__ __align__(256) unsigned char bg_segmax_scratch[8u * BG_SEGMAX_STATIC_ELEMS];

#define BG_MAX_DEVICES 64
static void* g_segmax_scratch_ptr[BG_MAX_DEVICES];      // per device, filled on that device's first call
static std::mutex g_segmax_scratch_mutex;

// The buffer belongs to the device the kernels will run on, i.e. the stream's.  Like every launch of this library
// the call needs that device to be the current one (a launch on another device's stream fails in the runtime); here
// a mismatch would also pick another GPU's address, so it is refused before anything is launched.
static int segmax_scratch_for_stream(hipStream_t s, void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BG_MAX_DEVICES) return BG_ERR_LAUNCH;
    if (s) {
        hipDevice_t sd;
        if (hipStreamGetDevice(s, &sd) != hipSuccess) return BG_ERR_LAUNCH;
        if ((int)sd != dev) return BG_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_segmax_scratch_mutex);
    if (!g_segmax_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(bg_segmax_scratch)) != hipSuccess || !p) return BG_ERR_LAUNCH;
        g_segmax_scratch_ptr[dev] = p;
    }
    *out = g_segmax_scratch_ptr[dev];
    return BG_OK;
}

int bg_segmax_static_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                         float* out, int64_t ldo, int32_t* arg, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0 || n_chunks_ub < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!chunk_ptr) return BG_ERR_ARG;
    if (n_chunks_ub > BG_SEGMAX_STATIC_ELEMS || n_chunks_ub * H > (int64_t)BG_SEGMAX_STATIC_ELEMS) return BG_ERR_SCRATCH;
    void* base = nullptr;
    int rc = segmax_scratch_for_stream((hipStream_t)stream, &base);
    if (rc != BG_OK) return rc;
    // nothing has to start at zero: phase 1 writes every partial that phase 2 reads
    float* pval = reinterpret_cast<float*>(base);
    int32_t* parg = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(base) + 4u * BG_SEGMAX_STATIC_ELEMS);
    return bg_segmax_f32(x, ldx, rowptr, col, chunk_ptr, rows_per_chunk, n_chunks_ub, pval, parg, out, ldo, arg,
                         n_rows, H, stream);
}

int bg_gather_arg_f32(const float* x, int64_t ldx, const int32_t* arg, float* out, int64_t ldo,
                      int64_t n_seg, int64_t H, bgStream stream) {
    if (n_seg < 0 || H < 0) return BG_ERR_ARG;
    if (n_seg == 0 || H == 0) return BG_OK;
    if (!x || !arg || !out || ldo < H) return BG_ERR_ARG;
    k_gather_arg<<<bg_grid_for(n_seg * H, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(x, ldx, arg, out, ldo, n_seg, (int)H);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_scatter_arg_f32(const float* g, int64_t ldg, const int32_t* arg, const int32_t* seg_of, float* out, int64_t ldo,
                       int64_t n_seg, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_seg < 0 || H < 0 || n_rows < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!out || ldo < H || (n_seg > 0 && (!g || !arg))) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (seg_of && n_seg > 0) {          // one launch: every row looks its own value up
        if (H % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && bg_aligned16(g) && bg_aligned16(arg) && bg_aligned16(out) &&
            n_rows * (H / 4) < 0x7fffffff) {
            const uint32_t n4 = (uint32_t)(n_rows * (H / 4));
            k_scatter_arg_dense4<<<(n4 + 255u) / 256u, 256, 0, s>>>(g, ldg, arg, seg_of, out, ldo, n4, (uint32_t)(H / 4));
            BG_LAUNCH_CHECK();
            return BG_OK;
        }
        k_scatter_arg_dense<<<bg_grid_for(n_rows * H, 256, 0x7fffffff), 256, 0, s>>>(g, ldg, arg, seg_of, out, ldo, n_rows, (int)H);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (ldo == H) {
        BG_HIP(bg_zero_async(out, sizeof(float) * n_rows * H, s));
    } else {
        for (int64_t r = 0; r < n_rows; ++r) BG_HIP(bg_zero_async(out + r * ldo, sizeof(float) * H, s));   // (pitched output: never on the step's path)
    }
    if (n_seg == 0) return BG_OK;
    k_scatter_arg<<<bg_grid_for(n_seg * H, 256, 0x7fffffff), 256, 0, s>>>(g, ldg, arg, out, ldo, n_seg, (int)H);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// Two gathers of the same width in one launch (the critic's backward reads the penalty's gradient at the arg-max voxels
// of the buildings and of the stories): out_i[s, h] = x_i[arg_i[s, h], h], 0 where arg_i is -1
int bg_gather_arg_pair_f32(const float* x0, int64_t ldx0, const int32_t* arg0, float* out0, int64_t n_seg0,
                           const float* x1, int64_t ldx1, const int32_t* arg1, float* out1, int64_t n_seg1,
                           int64_t ldo, int64_t H, bgStream stream) {
    if (n_seg0 < 0 || n_seg1 < 0 || H < 0) return BG_ERR_ARG;
    if (n_seg0 + n_seg1 == 0 || H == 0) return BG_OK;
    if ((n_seg0 > 0 && (!x0 || !arg0 || !out0)) || (n_seg1 > 0 && (!x1 || !arg1 || !out1)) || ldo < H) return BG_ERR_ARG;
    k_gather_arg_pair<<<bg_grid_for((n_seg0 + n_seg1) * H, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        x0, ldx0, arg0, out0, n_seg0, x1, ldx1, arg1, out1, n_seg1, ldo, (int)H);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// Two scatters over the same rows in one launch, each with its own segments: out_i[r, h] = g_i[seg_of_i[r], h] where
// arg_i[seg_of_i[r], h] == r, else 0 (bg_scatter_arg_f32's one-launch form twice; the shapes it does not take -- no
// segment ids, H % 4 != 0, unaligned rows -- go through two calls of that function)
int bg_scatter_arg_pair_f32(const float* g0, int64_t ldg0, const int32_t* arg0, const int32_t* seg_of0, float* out0, int64_t n_seg0,
                            const float* g1, int64_t ldg1, const int32_t* arg1, const int32_t* seg_of1, float* out1, int64_t n_seg1,
                            int64_t ldo, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_seg0 < 0 || n_seg1 < 0 || H < 0 || n_rows < 0) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!out0 || !out1 || ldo < H) return BG_ERR_ARG;
    const bool vec = n_seg0 > 0 && n_seg1 > 0 && g0 && g1 && arg0 && arg1 && seg_of0 && seg_of1 && H % 4 == 0 &&
                     ldg0 % 4 == 0 && ldg1 % 4 == 0 && ldo % 4 == 0 && bg_aligned16(g0) && bg_aligned16(g1) &&
                     bg_aligned16(arg0) && bg_aligned16(arg1) && bg_aligned16(out0) && bg_aligned16(out1) &&
                     n_rows * (H / 4) < 0x7fffffff;
    if (!vec) {
        int rc = bg_scatter_arg_f32(g0, ldg0, arg0, seg_of0, out0, ldo, n_seg0, n_rows, H, stream);
        if (rc != BG_OK) return rc;
        return bg_scatter_arg_f32(g1, ldg1, arg1, seg_of1, out1, ldo, n_seg1, n_rows, H, stream);
    }
    const uint32_t n4 = (uint32_t)(n_rows * (H / 4));
    ScatterPair p{{g0, g1}, {ldg0, ldg1}, {arg0, arg1}, {seg_of0, seg_of1}, {out0, out1}};
    k_scatter_arg_dense4_pair<<<dim3((n4 + 255u) / 256u, 2), 256, 0, (hipStream_t)stream>>>(p, ldo, n4, (uint32_t)(H / 4));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// n_batch scatters that share the segments (rows' segment ids) in one launch: batch b reads g + b * g_step and
// arg + b * arg_step and writes out + b * out_step (the critic's pooling backward for D(real) and D(fake)); sum_out
// (optional): the elementwise sum of the batches' results, in batch order, written by the same launch
int bg_scatter_arg_batch_f32(const float* g, int64_t ldg, int64_t g_step, const int32_t* arg, int64_t arg_step,
                             const int32_t* seg_of, float* out, int64_t ldo, int64_t out_step, int64_t n_seg, int64_t n_rows,
                             int64_t H, int64_t n_batch, float* sum_out, int64_t ld_sum, bgStream stream) {
    if (n_seg <= 0 || H < 0 || n_rows < 0 || n_batch < 1 || n_batch > 65535) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!out || !g || !arg || !seg_of || ldo < H || (sum_out && ld_sum < H)) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = H % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && g_step % 4 == 0 && arg_step % 4 == 0 && out_step % 4 == 0 &&
                     bg_aligned16(g) && bg_aligned16(arg) && bg_aligned16(out) && n_rows * (H / 4) < 0x7fffffff;
    if (sum_out) {          // the batches' sum as well, by the launch that writes them (16-byte path only, <= 8 batches)
        if (!vec || !bg_aligned16(sum_out) || ld_sum % 4 != 0) return BG_ERR_ALIGN;
        if (n_batch > 8) return BG_ERR_SHAPE;
        const uint32_t n4 = (uint32_t)(n_rows * (H / 4));
        k_scatter_arg_dense4_sum<<<(n4 + 255u) / 256u, 256, 0, s>>>(g, ldg, arg, seg_of, out, ldo, n4, (uint32_t)(H / 4), g_step,
                                                                   arg_step, out_step, (int)n_batch, sum_out, ld_sum);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    if (vec) {
        const uint32_t n4 = (uint32_t)(n_rows * (H / 4));
        k_scatter_arg_dense4<<<dim3((n4 + 255u) / 256u, (unsigned)n_batch), 256, 0, s>>>(g, ldg, arg, seg_of, out, ldo, n4,
                                                                                        (uint32_t)(H / 4), g_step, arg_step, out_step);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    for (int64_t b = 0; b < n_batch; ++b) {
        int rc = bg_scatter_arg_f32(g + b * g_step, ldg, arg + b * arg_step, seg_of, out + b * out_step, ldo, n_seg, n_rows, H, stream);
        if (rc != BG_OK) return rc;
    }
    return BG_OK;
}

// n_batch segment maxima over the SAME segments in one launch (a workgroup per segment and batch; the form of
// bg_segmax_f32 without chunk tables): batch b reduces x + b * x_step into out + b * out_step, arg + b * arg_step
int bg_segmax_batch_f32(const float* x, int64_t ldx, int64_t x_step, const int32_t* rowptr, const int32_t* col,
                        float* out, int64_t ldo, int64_t out_step, int32_t* arg, int64_t arg_step,
                        int64_t n_rows, int64_t H, int64_t n_batch, bgStream stream) {
    if (n_rows < 0 || H < 0 || n_batch < 1 || n_batch > 65535) return BG_ERR_ARG;
    if (n_rows == 0 || H == 0) return BG_OK;
    if (!x || !rowptr || !out || !arg || ldx < H || ldo < H) return BG_ERR_ARG;
    if (x_step % 4 != 0 || out_step % 4 != 0 || arg_step % 4 != 0) return BG_ERR_ALIGN;
    ChunkArgs a{x, ldx, rowptr, col, nullptr, nullptr, nullptr, 0, n_rows, nullptr, nullptr, (int)H,
                out, ldo, arg, nullptr, 0, x_step, out_step, arg_step};
    return launch_chunk_reduce<true>(a, n_rows, (hipStream_t)stream, n_batch);
}

int bg_sddmm_f32(const float* a, int64_t lda, const float* b, int64_t ldb,
                 const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                 float* out, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H < 0) return BG_ERR_ARG;
    if (n_rows == 0) return BG_OK;
    if (!a || !b || !rowptr || !out || lda < H || ldb < H) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bool vec = (H % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && bg_aligned16(a) && bg_aligned16(b) && H >= 4;
    if (!vec) {
        k_sddmm_scalar<<<bg_grid_for(n_rows, 256, 0x7fffffff), 256, 0, s>>>(a, lda, b, ldb, rowptr, col, eperm, out, n_rows, (int)H);
        BG_LAUNCH_CHECK();
        return BG_OK;
    }
    int H4 = (int)(H / 4);
    int grid = bg_grid_for(n_rows, 4, 0x7fffffff);
    if (H4 >= 32) k_sddmm<32><<<grid, 256, 0, s>>>(a, lda, b, ldb, rowptr, col, eperm, out, n_rows, H4);
    else if (H4 >= 8) k_sddmm<8><<<grid, 256, 0, s>>>(a, lda, b, ldb, rowptr, col, eperm, out, n_rows, H4);
    else k_sddmm<1><<<grid, 256, 0, s>>>(a, lda, b, ldb, rowptr, col, eperm, out, n_rows, H4);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_story_project_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const float* count, const float* W, int64_t ldw, int64_t Hout,
                         float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream) {
    if (n_rows < 0 || H <= 0 || Hout <= 0) return BG_ERR_ARG;
    if (n_rows == 0) return BG_OK;
    if (!x || !rowptr || !count || !W || !out || ldx < H || ldw < 2 * H || ldo < 2 * Hout) return BG_ERR_ARG;
    if (H % 4 != 0 || H > 256) return BG_ERR_SHAPE;
    if (!bg_aligned16(x) || !bg_aligned16(W) || ldx % 4 != 0 || ldw % 4 != 0) return BG_ERR_ALIGN;
    k_story_project<<<(unsigned)n_rows, 1024, 0, (hipStream_t)stream>>>(x, ldx, rowptr, col, count, W, ldw, (int)Hout,
                                                                      out, ldo, (int)H);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_story_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* v, int64_t ldv,
                     const int32_t* rowptr, const int32_t* col, const float* count, float slope,
                     float* gz, int64_t ldz, float* gterm, int64_t ldt, float* story, int64_t lds,
                     int64_t n_story, int64_t H, bgStream stream) {
    if (n_story < 0 || H <= 0) return BG_ERR_ARG;
    if (n_story == 0) return BG_OK;
    if (!g || !y || !v || !rowptr || !count || !gz || !gterm || !story) return BG_ERR_ARG;
    if (ldg < H || ldy < H || ldv < H || ldz < H || ldt < H || lds < H) return BG_ERR_ARG;
    if (H % 4 != 0 || H > 256 || n_story > 0x7fffffff) return BG_ERR_SHAPE;
    if (!bg_aligned16(g) || !bg_aligned16(y) || !bg_aligned16(v) || !bg_aligned16(gz) || !bg_aligned16(gterm) ||
        !bg_aligned16(story) || ldg % 4 != 0 || ldy % 4 != 0 || ldv % 4 != 0 || ldz % 4 != 0 || ldt % 4 != 0 || lds % 4 != 0)
        return BG_ERR_ALIGN;
    k_story_bwd<<<(unsigned)n_story, 1024, 0, (hipStream_t)stream>>>(g, ldg, y, ldy, v, ldv, rowptr, col, count, slope,
                                                                    gz, ldz, gterm, ldt, story, lds, (int)H);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_gather_story_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                        const float* row_scale, int row_mode, float* gm, int64_t ldg, int64_t n_rows,
                        const int32_t* s_rowptr, const int32_t* s_col, const float* s_count,
                        const float* W, int64_t ldw, int64_t Hout, float* pq, int64_t ldq, int64_t n_story,
                        int64_t H, bgStream stream) {
    if (n_rows < 0 || n_story < 0 || H <= 0 || Hout <= 0) return BG_ERR_ARG;
    if (n_rows == 0 && n_story == 0) return BG_OK;
    if (!x || !rowptr || !col || !gm || !s_rowptr || !s_count || !W || !pq) return BG_ERR_ARG;
    if (ldx < H || ldg < H || ldw < 2 * H || ldq < 2 * Hout || (row_mode && !row_scale)) return BG_ERR_ARG;
    if (H % 4 != 0 || H > 256 || n_story > (1 << 24)) return BG_ERR_SHAPE;
    if (!bg_aligned16(x) || !bg_aligned16(gm) || !bg_aligned16(W) || ldx % 4 != 0 || ldg % 4 != 0 || ldw % 4 != 0)
        return BG_ERR_ALIGN;
    SpmmArgs a{x, ldx, rowptr, col, nullptr, nullptr, row_scale, row_mode, nullptr, 0, gm, ldg, n_rows, (int)(H / 4),
               nullptr, 0, nullptr, 0};
    int lpr = 1;
    while (lpr < a.H4) lpr <<= 1;
    const int rpb = 16 * (64 / lpr);                   // 1024 threads per workgroup: 16 waves of gather-mean rows
    const int n_story_blocks = (int)((n_story + 7) / 8 * 8);
    StoryArgs st{s_rowptr, s_col, s_count, W, ldw, (int)Hout, pq, ldq, (int)n_story, n_story_blocks};
    const int64_t grid = (int64_t)n_story_blocks + (n_rows + rpb - 1) / rpb;
    if (grid > 0x7fffffff) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (lpr) {
        case 1: k_gather_story<1><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        case 2: k_gather_story<2><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        case 4: k_gather_story<4><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        case 8: k_gather_story<8><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        case 16: k_gather_story<16><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        case 32: k_gather_story<32><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
        default: k_gather_story<64><<<(unsigned)grid, 1024, 0, s>>>(a, st, (int)H); break;
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
