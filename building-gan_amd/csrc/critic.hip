// Small per-voxel kernels around the critic update and the generator's output stage: each replaces a
// chain of stock elementwise / reduction / scatter launches of a few microseconds of work apiece
// (the step is launch-bound at batch 8), keeping the reference's operation order.
#include "common.h"

namespace {

constexpr int MAX_CLASS = 8;

// reference Model/losses.py:76-79: mixed = label*u + out*(1-u); nonzero = (max(mixed) != 0);
// sample = ((onehot(first argmax) - mixed) + mixed) * nonzero.  No fused multiply-adds: every product and
// sum is rounded separately, as the separate elementwise launches of the reference do.
__global__ void k_gp_sample(const float* __restrict__ label, int64_t ldl, const float* __restrict__ out, int64_t ldo,
                            const float* __restrict__ u, float* __restrict__ sample, int64_t lds, int64_t n, int C) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float uv = u[v];
    const float one_minus = __fsub_rn(1.f, uv);
    float m[MAX_CLASS];
    float best = 0.f; int arg = 0;
#pragma unroll
    for (int j = 0; j < MAX_CLASS; ++j) {
        if (j < C) {
            m[j] = __fadd_rn(__fmul_rn(label[v * ldl + j], uv), __fmul_rn(out[v * ldo + j], one_minus));
            if (j == 0 || m[j] > best) { best = m[j]; arg = j; }
        }
    }
    const float nonzero = best != 0.f ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < MAX_CLASS; ++j) {
        if (j < C) {
            const float hard = __fadd_rn(__fsub_rn(j == arg ? 1.f : 0.f, m[j]), m[j]);
            sample[v * lds + j] = __fmul_rn(hard, nonzero);
        }
    }
}

// dst[v, p*group + j] = j < C ? src_p[v, j] : 0   for p < n_src, j < group
struct PackArgs { const float* src[4]; int64_t ld[4]; };
__global__ void k_pack_columns(PackArgs a, int n_src, int C, int group, float* __restrict__ dst, int64_t ldd, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int width = n_src * group;
    if (t >= n * width) return;
    const int64_t v = t / width; const int c = (int)(t % width);
    const int p = c / group, j = c % group;
    dst[v * ldd + c] = j < C ? a.src[p][v * a.ld[p] + j] : 0.f;
}

// reference Model/models.py:128-141 (Attention.construct_output) + util_graph.py:44-48, one thread per voxel
// over its cross edges in slot order (= edge order: the edges are voxel-major): first arg-max slot of
// att_hard * mask_hard, the hard and soft class-label sums, both multiplied by the hard mask.
__global__ void k_construct_output(const float* __restrict__ mask_hard, const float* __restrict__ att_hard,
                                   const float* __restrict__ att_soft, const int32_t* __restrict__ rowptr,
                                   const int32_t* __restrict__ col, const float* __restrict__ cls, int64_t ldc, int C,
                                   float* __restrict__ out, float* __restrict__ soft_out, int64_t* __restrict__ idx,
                                   int64_t n) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float mh = mask_hard[v];
    float hs[MAX_CLASS], ss[MAX_CLASS];
#pragma unroll
    for (int j = 0; j < MAX_CLASS; ++j) { hs[j] = 0.f; ss[j] = 0.f; }
    float best = 0.f; int32_t arg = -1;
    for (int32_t k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const float ah = att_hard[k], as = att_soft[k];
        const float logit = __fmul_rn(ah, mh);
        if (arg < 0 || logit > best) { best = logit; arg = k; }
        const float* c = cls + (int64_t)col[k] * ldc;
#pragma unroll
        for (int j = 0; j < MAX_CLASS; ++j) {
            if (j < C) {
                const float cj = c[j];
                hs[j] = __fadd_rn(hs[j], __fmul_rn(ah, cj));
                ss[j] = __fadd_rn(ss[j], __fmul_rn(as, cj));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAX_CLASS; ++j) {
        if (j < C) {
            out[v * C + j] = __fmul_rn(mh, hs[j]);
            soft_out[v * C + j] = __fmul_rn(mh, ss[j]);
        }
    }
    idx[v] = arg >= 0 ? (int64_t)col[arg] : -1;
}

// WGAN-GP critic objective (reference Model/losses.py:14-19,85-87 with global-count means):
//   loss = <val_b, wb> + <val_s, ws> + coef * sum_r (||g_r|| - 1)^2      over the n_rows rows g_r of `grads`
// Phase 1 writes one partial sum per workgroup (fixed order inside a workgroup), phase 2 adds the
// partials in order: deterministic.
constexpr int LOSS_BLOCKS = 64;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = bg_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) s = (red[0] + red[1]) + (red[2] + red[3]);
    return s;
}

__global__ __launch_bounds__(256) void k_critic_loss_partial(const float* __restrict__ val_b, const float* __restrict__ wb, int64_t nb,
                                                             const float* __restrict__ val_s, const float* __restrict__ ws, int64_t ns,
                                                             const float* __restrict__ grads, int64_t ldg, int64_t n_rows, int W,
                                                             float coef, float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (int64_t i = t0; i < nb; i += stride) acc = fmaf(val_b[i], wb[i], acc);
    for (int64_t i = t0; i < ns; i += stride) acc = fmaf(val_s[i], ws[i], acc);
    float pen = 0.f;
    for (int64_t r = t0; r < n_rows; r += stride) {
        const float* g = grads + r * ldg;
        float sq = 0.f;
        for (int j = 0; j < W; ++j) sq = fmaf(g[j], g[j], sq);
        const float d = sqrtf(sq) - 1.f;
        pen = fmaf(d, d, pen);
    }
    acc = fmaf(coef, pen, acc);
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
    __syncthreads();
    const float sp = block_sum_256(pen, red);
    if (threadIdx.x == 0) partial[LOSS_BLOCKS + blockIdx.x] = sp;
}

__global__ void k_critic_loss_final(const float* __restrict__ partial, int n, float* __restrict__ loss) {
    if (threadIdx.x < 2 && blockIdx.x == 0) {      // loss[0] = the objective, loss[1] = the bare penalty sum
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += partial[threadIdx.x * n + i];
        loss[threadIdx.x] = s;
    }
}

// gradient of the objective, scaled by the upstream scalar go[0]:
//   d val_b = go * wb ; d val_s = go * ws ; d g_r = go * coef * 2 (||g_r|| - 1) / ||g_r|| * g_r  (0 where ||g_r|| = 0)
__global__ void k_critic_loss_bwd(const float* __restrict__ go, const float* __restrict__ wb, int64_t nb,
                                  const float* __restrict__ ws, int64_t ns,
                                  const float* __restrict__ grads, int64_t ldg, int64_t n_rows, int W, float coef,
                                  float* __restrict__ d_val_b, float* __restrict__ d_val_s, float* __restrict__ d_grads) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float g0 = go[0];
    if (t < nb && d_val_b) d_val_b[t] = g0 * wb[t];
    if (t < ns && d_val_s) d_val_s[t] = g0 * ws[t];
    if (t < n_rows && d_grads) {
        const float* g = grads + t * ldg;
        float sq = 0.f;
        for (int j = 0; j < W; ++j) sq = fmaf(g[j], g[j], sq);
        const float nrm = sqrtf(sq);
        const float f = nrm > 0.f ? g0 * coef * 2.f * (nrm - 1.f) / nrm : 0.f;
        for (int j = 0; j < W; ++j) d_grads[t * W + j] = f * g[j];
    }
}

// Mean position difference of every voxel to its in-neighbours, zero-padded to 4 columns:
//   out[v, 0:3] = has(v) * pos[v] - (sum_{j -> v} pos[j]) / max(deg(v), 1) ,  out[v, 3] = 0
// (the mean over in-edges of pos_i - pos_j, reference Model/models.py:160-161, constant per batch);
// slot order, IEEE division, products and differences rounded separately.
__global__ void k_pos_terms(const float* __restrict__ pos, int64_t ldp, const int32_t* __restrict__ rowptr,
                            const int32_t* __restrict__ col, float* __restrict__ out, int64_t n, float pad) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int32_t kb = rowptr[v], ke = rowptr[v + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int32_t k = kb; k < ke; ++k) {
        const float* pj = pos + (int64_t)col[k] * ldp;
        sx = __fadd_rn(sx, pj[0]); sy = __fadd_rn(sy, pj[1]); sz = __fadd_rn(sz, pj[2]);
    }
    const float cnt = (float)(ke - kb > 1 ? ke - kb : 1);
    const float has = ke > kb ? 1.f : 0.f;
    const float* pv = pos + v * ldp;
    float4 o;
    o.x = __fsub_rn(__fmul_rn(has, pv[0]), __fdiv_rn(sx, cnt));
    o.y = __fsub_rn(__fmul_rn(has, pv[1]), __fdiv_rn(sy, cnt));
    o.z = __fsub_rn(__fmul_rn(has, pv[2]), __fdiv_rn(sz, cnt));
    o.w = pad;
    *reinterpret_cast<float4*>(out + v * 4) = o;
}

// Label-free input columns of the collapsed discriminator (models.VoxelGNN_D._static_powers), one row per voxel:
//   [ feat (nf) | onehot(level) (ML) | pdm4 (4) | 1 | 0 .. ]   (ds columns)
__global__ void k_d_static_cols(const float* __restrict__ feat, int64_t ldf, int nf, const int32_t* __restrict__ level,
                                int ML, const float* __restrict__ pdm4, float* __restrict__ out, int64_t ldo, int ds,
                                int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * ds) return;
    const int64_t v = t / ds; const int c = (int)(t % ds);
    float val = 0.f;
    if (c < nf) val = feat[v * ldf + c];
    else if (c < nf + ML) val = (level[v] == c - nf) ? 1.f : 0.f;
    else if (c < nf + ML + 4) val = pdm4[v * 4 + (c - nf - ML)];
    else if (c == nf + ML + 4) val = 1.f;
    out[v * ldo + c] = val;
}

// Parameter-dependent rows of the collapsed discriminator's coefficient recurrence (ops.PolyCoefD): for the
// du = ds + dl input columns and the 2H hidden channels [feature half | label half],
//   x0[r, :]: r < nf: [fe_w[:, r]^T | 0];  nf <= r < nf+ML: [pe[r-nf] | 0];  r == rb = nf+ML+4: [fe_b | le_b];
//             ds <= r < ds+nl: [0 | le_w[:, r-ds]^T];  else 0
//   e[r, :]:  nf+ML <= r < rb: Wc[:, 4H + (r-nf-ML)]^T (the position block);  r == rb: bc;  else 0
__global__ void k_d_coef_init(const float* __restrict__ fe_w, const float* __restrict__ fe_b,
                              const float* __restrict__ le_w, const float* __restrict__ le_b,
                              const float* __restrict__ pe, const float* __restrict__ Wc, int64_t ldw,
                              const float* __restrict__ bc, int64_t ld_bc, int H, int nf, int nl, int ML, int ds, int du,
                              float* __restrict__ x0, float* __restrict__ e) {
    const int H2 = 2 * H;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= du * H2) return;
    const int r = t / H2, c = t % H2;
    const int rb = nf + ML + 4;
    float xv = 0.f, ev = 0.f;
    if (r < nf) { if (c < H) xv = fe_w[c * nf + r]; }
    else if (r < nf + ML) { if (c < H) xv = pe[(r - nf) * H + c]; }
    else if (r < rb) ev = Wc[(int64_t)c * ldw + 2 * H2 + (r - nf - ML)];
    else if (r == rb) { xv = c < H ? fe_b[c] : le_b[c - H]; ev = bc[(int64_t)c * ld_bc]; }
    else if (r >= ds && r < ds + nl) { if (c >= H) xv = le_w[(c - H) * nl + (r - ds)]; }
    x0[t] = xv;
    e[t] = ev;
}

// Its backward: G holds L+1 slabs (pitch `slab` floats) of [.., 2H] gradient states; slab 0 rows [0, du) are
// d/d x0, the sum over slabs 1..L of rows [0, du) is d/d e.  Every output element is written exactly once.
__global__ void k_d_coef_init_bwd(const float* __restrict__ G, int64_t slab, int L, int H, int nf, int nl, int ML,
                                  int ds, int du, float* __restrict__ g_fe_w, float* __restrict__ g_fe_b,
                                  float* __restrict__ g_le_w, float* __restrict__ g_le_b,
                                  float* __restrict__ gw_tail, int64_t ldw, float* __restrict__ g_bc, int64_t ld_gbc,
                                  int zero_pad) {
    const int H2 = 2 * H;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= du * H2) return;
    const int r = t / H2, c = t % H2;
    const int rb = nf + ML + 4;
    const float g0 = G[t];
    if (r < nf) { if (c < H) g_fe_w[c * nf + r] = g0; }
    else if (r == rb) { if (c < H) g_fe_b[c] = g0; else g_le_b[c - H] = g0; }
    else if (r >= ds && r < ds + nl) { if (c >= H) g_le_w[(c - H) * nl + (r - ds)] = g0; }
    if (r >= nf + ML && r <= rb) {
        float s = 0.f;
        for (int m = 1; m <= L; ++m) s = __fadd_rn(s, G[(int64_t)m * slab + t]);
        if (r < rb) gw_tail[(int64_t)c * ldw + (r - nf - ML)] = s;
        else {
            g_bc[(int64_t)c * ld_gbc] = s;
            if (zero_pad) {                          // columns 5..7 after the position block: the pad of a [.., 4H+8] row
                float* q = gw_tail + (int64_t)c * ldw;
                q[5] = 0.f; q[6] = 0.f; q[7] = 0.f;
            }
        }
    }
}

// models.VoxelBlock.compose for the discriminator's block (no story terms), P = 2 H2 + 8 columns per row:
//   wm_ext [H2, P] = [Wm | 0 | bm | 0 0 0]   (second operand of the product Wu2 wm_ext)
//   r      [H2, P] = [Wu1 | 0 .. | bu at column 2 H2 + 4 | 0 0 0]   (what the product's epilogue adds)
// so that  Wu2 wm_ext + r = [Wc (2 H2 + 4 columns) | bc | 0 0 0].
__global__ void k_d_compose_prep(const float* __restrict__ Wm, int64_t ldwm, const float* __restrict__ bm,
                                 const float* __restrict__ Wu, int64_t ldwu, const float* __restrict__ bu, int H2,
                                 float* __restrict__ wm_ext, float* __restrict__ r) {
    const int P = 2 * H2 + 8;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)H2 * P) return;
    const int o = (int)(t / P), c = (int)(t % P);
    wm_ext[t] = c < 2 * H2 + 3 ? Wm[(int64_t)o * ldwm + c] : (c == 2 * H2 + 4 ? bm[o] : 0.f);
    r[t] = c < H2 ? Wu[(int64_t)o * ldwu + c] : (c == 2 * H2 + 4 ? bu[o] : 0.f);
}

// Its backward's last step: gp [H2, P] = d / d [Wc | bc | pad], gwm_ext [H2, P] = Wu2^T gp  ->  the parameters' gradients
// (g_Wu's second half is written by the product gp wm_ext^T itself).
__global__ void k_d_compose_bwd_finish(const float* __restrict__ gp, const float* __restrict__ gwm_ext, int H2,
                                       float* __restrict__ g_Wm, float* __restrict__ g_bm, float* __restrict__ g_Wu,
                                       float* __restrict__ g_bu) {
    const int P = 2 * H2 + 8;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)H2 * P) return;
    const int o = (int)(t / P), c = (int)(t % P);
    if (c < 2 * H2 + 3) g_Wm[(int64_t)o * (2 * H2 + 3) + c] = gwm_ext[t];
    if (c < H2) g_Wu[(int64_t)o * (2 * H2) + c] = gp[t];
    if (c == 2 * H2 + 4) { g_bm[o] = gwm_ext[t]; g_bu[o] = gp[t]; }
}

// Multiplication by x in the Chebyshev basis, on a table of K row blocks of `blk4` float4 each (block j = the
// coefficient of T_j):  x T_0 = T_1,  x T_j = (T_{j+1} + T_{j-1}) / 2.
//   transpose == 0:  out_i = d_i X_{i-1} + X_{i+1} / 2,  d_1 = 1, d_i = 1/2 otherwise   (X_{-1} = X_K = 0)
//   transpose == 1:  out_0 = X_1,  out_j = (X_{j-1} + X_{j+1}) / 2                     (the adjoint)
// Blocks [0, K_out) are written.
__global__ void k_cheb_shift(const float4* __restrict__ X, float4* __restrict__ out, int K, int K_out, int64_t blk4,
                             int transpose) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)K_out * blk4) return;
    const int i = (int)(t / blk4);
    float lo = 0.5f, hi = 0.5f;
    if (transpose) { if (i == 0) hi = 1.f; } else { if (i == 1) lo = 1.f; }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 1) { const float4 a = X[t - blk4]; r.x = lo * a.x; r.y = lo * a.y; r.z = lo * a.z; r.w = lo * a.w; }
    if (i + 1 < K) {
        const float4 b = X[t + blk4];
        r.x = __fmaf_rn(hi, b.x, r.x); r.y = __fmaf_rn(hi, b.y, r.y); r.z = __fmaf_rn(hi, b.z, r.z); r.w = __fmaf_rn(hi, b.w, r.w);
    }
    out[t] = r;
}

// The same map on a run of tables laid out with a period: block b of the run is coefficient block (b % period) - lead of
// its table; blocks outside [0, K) (padding between tables) are written as zeros.
__global__ void k_cheb_shift_slabs(const float4* __restrict__ X, float4* __restrict__ out, int64_t n_blocks, int period,
                                   int lead, int K, int64_t blk4, int transpose) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks * blk4) return;
    const int i = (int)((t / blk4) % period) - lead;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0 && i < K) {
        float lo = 0.5f, hi = 0.5f;
        if (transpose) { if (i == 0) hi = 1.f; } else { if (i == 1) lo = 1.f; }
        if (i >= 1) { const float4 a = X[t - blk4]; r.x = lo * a.x; r.y = lo * a.y; r.z = lo * a.z; r.w = lo * a.w; }
        if (i + 1 < K) {
            const float4 b = X[t + blk4];
            r.x = __fmaf_rn(hi, b.x, r.x); r.y = __fmaf_rn(hi, b.y, r.y); r.z = __fmaf_rn(hi, b.z, r.z); r.w = __fmaf_rn(hi, b.w, r.w);
        }
    }
    out[t] = r;
}

// [n, K, P, W] <-> [P, n, K, W], one float4 per thread (see bg_regroup_f32)
__global__ void k_regroup(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n, int K, int P, int W4,
                          int to_outer) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * K * P * W4;
    if (t >= total) return;
    // t enumerates the INNER layout [n, K, P, W4]
    const int j = (int)(t % W4);
    const int p = (int)((t / W4) % P);
    const int k = (int)((t / ((int64_t)W4 * P)) % K);
    const int64_t v = t / ((int64_t)W4 * P * K);
    const int64_t outer = (((int64_t)p * n + v) * K + k) * W4 + j;
    if (to_outer) dst[outer] = src[t]; else dst[t] = src[outer];
}

// k_regroup and k_iota_mod in one grid (bg_regroup_iota_f32): thread t moves float4 t of the regrouping and, while
// t < idx_n, writes idx[t] = t mod idx_mod -- two independent writes the critic head needs before the same product
__global__ void k_regroup_iota(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n, int K, int P, int W4,
                               int to_outer, int32_t* __restrict__ idx, int64_t idx_n, int32_t idx_mod) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < idx_n) idx[t] = (int32_t)(t % idx_mod);
    const int64_t total = n * K * P * W4;
    if (t >= total) return;
    const int j = (int)(t % W4);
    const int p = (int)((t / W4) % P);
    const int k = (int)((t / ((int64_t)W4 * P)) % K);
    const int64_t v = t / ((int64_t)W4 * P * K);
    const int64_t outer = (((int64_t)p * n + v) * K + k) * W4 + j;
    if (to_outer) dst[outer] = src[t]; else dst[t] = src[outer];
}

// dst[r][i] = src[i] for r < reps (one float4 per thread of dst)
__global__ void k_tile4(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4, int64_t total4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total4) dst[t] = src[t % n4];
}

__global__ void k_fill4(float4* __restrict__ p, int64_t n4, float v) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n4) p[t] = make_float4(v, v, v, v);
}

// out = g where y > 0, slope * g elsewhere (y = the leaky ReLU's RESULT, which has the sign of its input)
__global__ void k_leaky_bwd(const float4* __restrict__ g, const float4* __restrict__ y, float slope, float4* __restrict__ out,
                            int64_t n4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const float4 a = g[t], b = y[t];
    out[t] = make_float4(b.x > 0.f ? a.x : a.x * slope, b.y > 0.f ? a.y : a.y * slope, b.z > 0.f ? a.z : a.z * slope,
                         b.w > 0.f ? a.w : a.w * slope);
}

// out[r, :] = scale[r] * x[r, :]  (x with a row pitch)
__global__ void k_rowscale(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale, float* __restrict__ out,
                           int64_t ldo, int64_t n, int H4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * H4) return;
    const int64_t r = t / H4; const int c = (int)(t % H4);
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + 4 * c);
    const float s = scale[r];
    *reinterpret_cast<float4*>(out + r * ldo + 4 * c) = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}

__global__ void k_iota_mod(int32_t* __restrict__ out, int64_t n, int32_t mod) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (int32_t)(t % mod);
}

__global__ void k_add(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int64_t n4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    const float4 x = a[t], y = b[t];
    out[t] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

}  // namespace

namespace {
// Backward of a two-layer decoder WITHOUT activation, val = (X W1^T + b1) w2^T + b2 (models.py:251-264: the critic's
// building / story heads), for the few rows a batch has (2 B graphs, 2 S stories), together with the gradient-penalty
// path's share.  With dv = d val every gradient is rank one:
//   u = dv^T X,  sdv = sum dv,  c = column sums of dgp (the penalty's d pooled rows; each is W2 W1's gradient)
//   g_w1 = w2 (x) (c + u)     g_b1 = sdv w2     g_w2 = W1 (c + u) + sdv b1     g_b2 = sdv     dX = dv (x) gp (+ add),  gp = w2 W1
// (dv^T h = W1 u + sdv b1 for the hidden rows h = X W1^T + b1, which are therefore never formed, forward or backward)
// One launch instead of ~11 (two split-K products with their reductions, four column sums, four small products).
// Every workgroup forms u, c (all H2 columns) and sdv itself -- a few hundred rows -- then finishes its own rows of g_w1
// (one per wave at a time) and its own rows of dX.  Fixed summation orders: deterministic.
struct DecBwdArgs {
    int R, Rp, H2, Hd;
    const float* dv; const float* X; int64_t ldx; const float* b1; const float* dgp; int64_t ldg;
    const float* w1; int64_t ldw1; const float* w2; const float* gp; const float* add; int64_t lda;
    float* g_w1; int64_t ldgw1; float* g_b1; float* g_w2; float* g_b2; float* dX; int64_t ldd;
};
// 1024 threads: thread (j, q) sums rows r = q (mod 4) of column j with 16 loads in flight (a sequential loop over a few
// hundred rows was ~20 us of dependent round trips), the four partial sums are added in q order through LDS.
constexpr int DEC_T = 1024, DEC_Q = 4, DEC_U = 16;
__global__ __launch_bounds__(DEC_T) void k_decoder_bwd(DecBwdArgs a) {
    extern __shared__ float dec_smem[];                 // c + u [H2] | partial [DEC_Q][H2] | dv [R]
    float* cu = dec_smem;
    float* part = dec_smem + a.H2;
    float* sdvp = part + DEC_Q * a.H2;                  // [DEC_T / 64] wave partials of sum dv
    float* dvs = sdvp + DEC_T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = gridDim.x, b = blockIdx.x;
    float sp = 0.f;
    for (int r = tid; r < a.R; r += DEC_T) { const float d = a.dv[r]; dvs[r] = d; sp += d; }
    sp = bg_wave_sum(sp);
    if (lane == 0) sdvp[wave] = sp;
    __syncthreads();
    float sdv = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_T / 64; ++w) sdv += sdvp[w];
    if (a.g_w1) {                                       // (only dX asked for: no column sums needed)
        const int nj = DEC_T / DEC_Q;                   // columns per pass
        for (int j0 = 0; j0 < a.H2; j0 += nj) {
            const int j = j0 + tid % nj, q = tid / nj;
            float acc = 0.f;
            if (j < a.H2) {
                for (int p0 = q; p0 < a.Rp; p0 += DEC_Q * DEC_U) {
                    float v[DEC_U];
#pragma unroll
                    for (int k = 0; k < DEC_U; ++k) { const int p = p0 + DEC_Q * k; v[k] = p < a.Rp ? a.dgp[(int64_t)p * a.ldg + j] : 0.f; }
#pragma unroll
                    for (int k = 0; k < DEC_U; ++k) acc += v[k];
                }
                for (int r0 = q; r0 < a.R; r0 += DEC_Q * DEC_U) {
                    float v[DEC_U];
#pragma unroll
                    for (int k = 0; k < DEC_U; ++k) { const int r = r0 + DEC_Q * k; v[k] = r < a.R ? a.X[(int64_t)r * a.ldx + j] : 0.f; }
#pragma unroll
                    for (int k = 0; k < DEC_U; ++k) { const int r = r0 + DEC_Q * k; acc = fmaf(r < a.R ? dvs[r] : 0.f, v[k], acc); }
                }
                part[q * a.H2 + j] = acc;
            }
        }
        __syncthreads();
        for (int j = tid; j < a.H2; j += DEC_T) {
            float t = part[j];
#pragma unroll
            for (int q = 1; q < DEC_Q; ++q) t += part[q * a.H2 + j];
            cu[j] = t;
        }
        __syncthreads();
        // rows i = b, b + nb, .. of the first layer's gradients: one per wave at a time
        for (int i = b + nb * wave; i < a.Hd; i += nb * (DEC_T / 64)) {
            float q = 0.f;
            for (int j = lane; j < a.H2; j += 64) q = fmaf(cu[j], a.w1[(int64_t)i * a.ldw1 + j], q);
            q = bg_wave_sum(q);
            const float w2i = a.w2[i];
            for (int j = lane; j < a.H2; j += 64) a.g_w1[(int64_t)i * a.ldgw1 + j] = w2i * cu[j];
            if (lane == 0) { a.g_w2[i] = fmaf(sdv, a.b1[i], q); a.g_b1[i] = sdv * w2i; }
        }
    }
    // rows r = b, b + nb, .. of dX
    {
        const int64_t per = ((int64_t)a.R + nb - 1) / nb, lo = (int64_t)b * per, hi = lo + per < a.R ? lo + per : a.R;
        for (int64_t e = lo * a.H2 + tid; e < hi * a.H2; e += DEC_T) {
            const int64_t r = e / a.H2;
            const int j = (int)(e - r * a.H2);
            float v = dvs[r] * a.gp[j];
            if (a.add) v += a.add[r * a.lda + j];
            a.dX[r * a.ldd + j] = v;
        }
    }
    if (a.g_b2 && b == 0 && tid == 0) a.g_b2[0] = sdv;
}

// The same decoder's forward on those rows: without an activation between the layers val = X (w2 W1)^T + (w2 . b1 + b2),
// so the launch forms gp = w2 W1 [H2] (kept: the backward and the penalty's closed form need it) and one dot product
// per row, instead of two Linear launches and a separate product for gp.  Every workgroup forms gp itself.
struct DecFwdArgs {
    int R, H2, Hd;
    const float* X; int64_t ldx; const float* w1; int64_t ldw1; const float* b1; const float* w2; const float* b2;
    float* val; float* gp;
};
__device__ __forceinline__ void decoder_fwd_body(const DecFwdArgs& a, const int b, const int nb) {
    extern __shared__ float dec_smem[];                 // gp [H2] | partial [DEC_Q][H2]
    float* gps = dec_smem;
    float* part = dec_smem + a.H2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nj = DEC_T / DEC_Q;
    for (int j0 = 0; j0 < a.H2; j0 += nj) {
        const int j = j0 + tid % nj, q = tid / nj;
        float acc = 0.f;
        if (j < a.H2) {
            for (int i0 = q; i0 < a.Hd; i0 += DEC_Q * DEC_U) {
                float v[DEC_U], w[DEC_U];
#pragma unroll
                for (int k = 0; k < DEC_U; ++k) {
                    const int i = i0 + DEC_Q * k;
                    v[k] = i < a.Hd ? a.w1[(int64_t)i * a.ldw1 + j] : 0.f;
                    w[k] = i < a.Hd ? a.w2[i] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < DEC_U; ++k) acc = fmaf(w[k], v[k], acc);
            }
            part[q * a.H2 + j] = acc;
        }
    }
    float c0 = 0.f;
    for (int i = lane; i < a.Hd; i += 64) c0 = fmaf(a.w2[i], a.b1[i], c0);
    c0 = bg_wave_sum(c0) + a.b2[0];
    __syncthreads();
    for (int j = tid; j < a.H2; j += DEC_T) {
        float t = part[j];
#pragma unroll
        for (int q = 1; q < DEC_Q; ++q) t += part[q * a.H2 + j];
        gps[j] = t;
        if (b == 0) a.gp[j] = t;
    }
    __syncthreads();
    for (int r = b * (DEC_T / 64) + wave; r < a.R; r += nb * (DEC_T / 64)) {
        float d = 0.f;
        for (int j = lane; j < a.H2; j += 64) d = fmaf(a.X[(int64_t)r * a.ldx + j], gps[j], d);
        d = bg_wave_sum(d);
        if (lane == 0) a.val[r] = d + c0;
    }
}
__global__ __launch_bounds__(DEC_T) void k_decoder_fwd(DecFwdArgs a) { decoder_fwd_body(a, blockIdx.x, gridDim.x); }
// two decoders side by side: workgroups [0, nb0) run the first exactly as its own launch would, the rest the second
__global__ __launch_bounds__(DEC_T) void k_decoder_fwd_pair(DecFwdArgs a0, DecFwdArgs a1, int nb0) {
    if ((int)blockIdx.x < nb0) decoder_fwd_body(a0, blockIdx.x, nb0);
    else decoder_fwd_body(a1, blockIdx.x - nb0, gridDim.x - nb0);
}
}  // namespace

extern "C" {

int bg_regroup_f32(const float* src, float* dst, int64_t n, int64_t K, int64_t P, int64_t W, int to_outer, bgStream stream) {
    if (n < 0 || K <= 0 || P <= 0 || W <= 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!src || !dst) return BG_ERR_ARG;
    if (W % 4 != 0 || K > (1 << 20) || P > (1 << 20)) return BG_ERR_SHAPE;
    if (!bg_aligned16(src) || !bg_aligned16(dst)) return BG_ERR_ALIGN;
    const int64_t total = n * K * P * (W / 4);
    k_regroup<<<bg_grid_for(total, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n, (int)K, (int)P, (int)(W / 4), to_outer);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// bg_regroup_f32 and bg_iota_mod_i32 (idx[t] = t mod idx_mod, t < idx_n) as ONE launch: see include/bgan.h
int bg_regroup_iota_f32(const float* src, float* dst, int64_t n, int64_t K, int64_t P, int64_t W, int to_outer,
                        int32_t* idx, int64_t idx_n, int64_t idx_mod, bgStream stream) {
    if (n < 0 || K <= 0 || P <= 0 || W <= 0 || idx_n < 0 || idx_mod <= 0 || idx_mod >= ((int64_t)1 << 31)) return BG_ERR_ARG;
    if (n == 0 && idx_n == 0) return BG_OK;
    if ((n > 0 && (!src || !dst)) || (idx_n > 0 && !idx)) return BG_ERR_ARG;
    if (W % 4 != 0 || K > (1 << 20) || P > (1 << 20)) return BG_ERR_SHAPE;
    if (n > 0 && (!bg_aligned16(src) || !bg_aligned16(dst))) return BG_ERR_ALIGN;
    const int64_t total = n * K * P * (W / 4);
    k_regroup_iota<<<bg_grid_for(total > idx_n ? total : idx_n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n, (int)K, (int)P, (int)(W / 4), to_outer, idx, idx_n,
        (int32_t)idx_mod);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_fill_f32(float* p, int64_t n, float value, bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!p) return BG_ERR_ARG;
    if (n % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(p)) return BG_ERR_ALIGN;
    k_fill4<<<bg_grid_for(n / 4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<float4*>(p), n / 4, value);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_leaky_bwd_f32(const float* g, const float* y, float slope, float* out, int64_t n, bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!g || !y || !out) return BG_ERR_ARG;
    if (n % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(g) || !bg_aligned16(y) || !bg_aligned16(out)) return BG_ERR_ALIGN;
    k_leaky_bwd<<<bg_grid_for(n / 4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(y), slope, reinterpret_cast<float4*>(out), n / 4);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_rowscale_f32(const float* x, int64_t ldx, const float* scale, float* out, int64_t ldo, int64_t n, int64_t H,
                    bgStream stream) {
    if (n < 0 || H < 0) return BG_ERR_ARG;
    if (n == 0 || H == 0) return BG_OK;
    if (!x || !scale || !out || ldx < H || ldo < H) return BG_ERR_ARG;
    if (H % 4 != 0 || ldx % 4 != 0 || ldo % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(x) || !bg_aligned16(out)) return BG_ERR_ALIGN;
    k_rowscale<<<bg_grid_for(n * (H / 4), 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(x, ldx, scale, out, ldo, n, (int)(H / 4));
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_iota_mod_i32(int32_t* out, int64_t n, int64_t mod, bgStream stream) {
    if (n < 0 || mod <= 0 || mod >= ((int64_t)1 << 31)) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!out) return BG_ERR_ARG;
    k_iota_mod<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(out, n, (int32_t)mod);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_add_f32(const float* a, const float* b, float* out, int64_t n, bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!a || !b || !out) return BG_ERR_ARG;
    if (n % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(a) || !bg_aligned16(b) || !bg_aligned16(out)) return BG_ERR_ALIGN;
    k_add<<<bg_grid_for(n / 4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n / 4);
    BG_LAUNCH_CHECK();
    return BG_OK;
}


int bg_gp_sample_f32(const float* label, int64_t ldl, const float* out, int64_t ldo, const float* u,
                     float* sample, int64_t lds, int64_t n, int64_t C, bgStream stream) {
    if (n < 0 || C < 0) return BG_ERR_ARG;
    if (n == 0 || C == 0) return BG_OK;
    if (!label || !out || !u || !sample || ldl < C || ldo < C || lds < C) return BG_ERR_ARG;
    if (C > MAX_CLASS) return BG_ERR_SHAPE;
    k_gp_sample<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(label, ldl, out, ldo, u, sample, lds, n, (int)C);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_pack_columns_f32(const float* const* src, const int64_t* ld, int n_src, int64_t C, int64_t group,
                        float* dst, int64_t ldd, int64_t n, bgStream stream) {
    if (n < 0 || C < 0 || group < C || n_src < 1 || n_src > 4) return BG_ERR_ARG;
    if (n == 0 || group == 0) return BG_OK;
    if (!src || !ld || !dst || ldd < n_src * group) return BG_ERR_ARG;
    PackArgs a{};
    for (int p = 0; p < n_src; ++p) {
        if (!src[p] || ld[p] < C) return BG_ERR_ARG;
        a.src[p] = src[p]; a.ld[p] = ld[p];
    }
    k_pack_columns<<<bg_grid_for(n * n_src * group, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(a, n_src, (int)C, (int)group,
                                                                                                    dst, ldd, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_construct_output_f32(const float* mask_hard, const float* att_hard, const float* att_soft,
                            const int32_t* rowptr, const int32_t* col, const float* cls, int64_t ldc, int64_t C,
                            float* out, float* soft_out, int64_t* idx, int64_t n_vox, bgStream stream) {
    if (n_vox < 0 || C < 0) return BG_ERR_ARG;
    if (n_vox == 0) return BG_OK;
    if (!mask_hard || !att_hard || !att_soft || !rowptr || !col || !cls || !out || !soft_out || !idx || ldc < C) return BG_ERR_ARG;
    if (C > MAX_CLASS) return BG_ERR_SHAPE;
    k_construct_output<<<bg_grid_for(n_vox, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(mask_hard, att_hard, att_soft, rowptr, col,
                                                                                            cls, ldc, (int)C, out, soft_out, idx, n_vox);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_critic_loss_f32(const float* val_b, const float* wb, int64_t nb, const float* val_s, const float* ws, int64_t ns,
                       const float* grads, int64_t ldg, int64_t n_rows, int64_t W, float coef,
                       float* partial, float* loss, bgStream stream) {
    if (nb < 0 || ns < 0 || n_rows < 0 || W < 0) return BG_ERR_ARG;
    if (!partial || !loss || (nb > 0 && (!val_b || !wb)) || (ns > 0 && (!val_s || !ws))) return BG_ERR_ARG;
    if (n_rows > 0 && (!grads || ldg < W)) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    k_critic_loss_partial<<<LOSS_BLOCKS, 256, 0, s>>>(val_b, wb, nb, val_s, ws, ns, grads, ldg, n_rows, (int)W, coef, partial);
    BG_LAUNCH_CHECK();
    k_critic_loss_final<<<1, 64, 0, s>>>(partial, LOSS_BLOCKS, loss);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_critic_loss_bwd_f32(const float* go, const float* wb, int64_t nb, const float* ws, int64_t ns,
                           const float* grads, int64_t ldg, int64_t n_rows, int64_t W, float coef,
                           float* d_val_b, float* d_val_s, float* d_grads, bgStream stream) {
    if (nb < 0 || ns < 0 || n_rows < 0 || W < 0 || !go) return BG_ERR_ARG;
    if ((d_val_b && !wb) || (d_val_s && !ws) || (d_grads && (!grads || ldg < W))) return BG_ERR_ARG;
    int64_t work = nb > ns ? nb : ns;
    if (n_rows > work) work = n_rows;
    if (work == 0) return BG_OK;
    k_critic_loss_bwd<<<bg_grid_for(work, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(go, wb, nb, ws, ns, grads, ldg, n_rows, (int)W,
                                                                                          coef, d_val_b, d_val_s, d_grads);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_pos_terms_f32(const float* pos, int64_t ldp, const int32_t* rowptr, const int32_t* col, float* out, int64_t n,
                     float pad, bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!pos || !rowptr || !col || !out || ldp < 3) return BG_ERR_ARG;
    if (!bg_aligned16(out)) return BG_ERR_ALIGN;
    k_pos_terms<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(pos, ldp, rowptr, col, out, n, pad);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_static_cols_f32(const float* feat, int64_t ldf, int64_t nf, const int32_t* level, int64_t ML, const float* pdm4,
                         float* out, int64_t ldo, int64_t ds, int64_t n, bgStream stream) {
    if (n < 0 || nf < 0 || ML < 0 || ds < nf + ML + 5 || ldo < ds) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!feat || !level || !pdm4 || !out || ldf < nf) return BG_ERR_ARG;
    k_d_static_cols<<<bg_grid_for(n * ds, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(feat, ldf, (int)nf, level, (int)ML, pdm4,
                                                                                         out, ldo, (int)ds, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_coef_init_strided_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b,
                               const float* pe, const float* Wc, int64_t ldw, const float* bc, int64_t ld_bc, int64_t H,
                               int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, float* x0, float* e,
                               bgStream stream) {
    if (H <= 0 || nf < 0 || nl < 0 || ML < 0 || ds < nf + ML + 5 || dl < nl || ldw < 4 * H + 4 || ld_bc < 1) return BG_ERR_ARG;
    if (!fe_w || !fe_b || !le_w || !le_b || !pe || !Wc || !bc || !x0 || !e) return BG_ERR_ARG;
    const int64_t n = (ds + dl) * 2 * H;
    if (n > 0x7fffffff) return BG_ERR_SHAPE;
    k_d_coef_init<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        fe_w, fe_b, le_w, le_b, pe, Wc, ldw, bc, ld_bc, (int)H, (int)nf, (int)nl, (int)ML, (int)ds, (int)(ds + dl), x0, e);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_coef_init_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b, const float* pe,
                       const float* Wc, int64_t ldw, const float* bc, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                       int64_t ds, int64_t dl, float* x0, float* e, bgStream stream) {
    return bg_d_coef_init_strided_f32(fe_w, fe_b, le_w, le_b, pe, Wc, ldw, bc, 1, H, nf, nl, ML, ds, dl, x0, e, stream);
}

int bg_d_compose_prep_f32(const float* Wm, int64_t ldwm, const float* bm, const float* Wu, int64_t ldwu, const float* bu,
                          int64_t H2, float* wm_ext, float* r, bgStream stream) {
    if (H2 <= 0 || ldwm < 2 * H2 + 3 || ldwu < 2 * H2) return BG_ERR_ARG;
    if (!Wm || !bm || !Wu || !bu || !wm_ext || !r) return BG_ERR_ARG;
    const int64_t n = H2 * (2 * H2 + 8);
    if (n > 0x7fffffff) return BG_ERR_SHAPE;
    k_d_compose_prep<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(Wm, ldwm, bm, Wu, ldwu, bu, (int)H2,
                                                                                       wm_ext, r);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_compose_bwd_finish_f32(const float* gp, const float* gwm_ext, int64_t H2, float* g_Wm, float* g_bm, float* g_Wu,
                                float* g_bu, bgStream stream) {
    if (H2 <= 0 || !gp || !gwm_ext || !g_Wm || !g_bm || !g_Wu || !g_bu) return BG_ERR_ARG;
    const int64_t n = H2 * (2 * H2 + 8);
    if (n > 0x7fffffff) return BG_ERR_SHAPE;
    k_d_compose_bwd_finish<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(gp, gwm_ext, (int)H2, g_Wm, g_bm,
                                                                                             g_Wu, g_bu);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_tile_f32(const float* src, float* dst, int64_t n, int64_t reps, bgStream stream) {
    if (n < 0 || reps < 0) return BG_ERR_ARG;
    if (n == 0 || reps == 0) return BG_OK;
    if (!src || !dst) return BG_ERR_ARG;
    if (n % 4 != 0 || (n / 4) * reps > 0x7fffffffLL * 256) return BG_ERR_SHAPE;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return BG_ERR_ALIGN;
    const int64_t total4 = (n / 4) * reps;
    k_tile4<<<bg_grid_for(total4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4, total4);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_cheb_shift_f32(const float* x, float* out, int64_t K, int64_t K_out, int64_t block_floats, int32_t transpose,
                      bgStream stream) {
    if (K < 0 || K_out < 0 || K_out > K || block_floats < 0) return BG_ERR_ARG;
    if (K_out == 0 || block_floats == 0) return BG_OK;
    if (!x || !out || x == out) return BG_ERR_ARG;
    if (block_floats % 4 != 0 || K_out * (block_floats / 4) > 0x7fffffff) return BG_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out) & 15) return BG_ERR_ALIGN;
    const int64_t n4 = K_out * (block_floats / 4);
    k_cheb_shift<<<bg_grid_for(n4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), (int)K, (int)K_out, block_floats / 4,
        transpose ? 1 : 0);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_cheb_shift_slabs_f32(const float* x, float* out, int64_t n_blocks, int64_t period, int64_t lead, int64_t K,
                            int64_t block_floats, int32_t transpose, bgStream stream) {
    if (n_blocks < 0 || period <= 0 || lead < 0 || K < 0 || lead + K > period || block_floats < 0) return BG_ERR_ARG;
    if (n_blocks == 0 || block_floats == 0) return BG_OK;
    if (!x || !out || x == out) return BG_ERR_ARG;
    if (block_floats % 4 != 0 || n_blocks * (block_floats / 4) > 0x7fffffff || period > 0x7fffffff) return BG_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out) & 15) return BG_ERR_ALIGN;
    const int64_t n4 = n_blocks * (block_floats / 4);
    k_cheb_shift_slabs<<<bg_grid_for(n4, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), n_blocks, (int)period, (int)lead, (int)K,
        block_floats / 4, transpose ? 1 : 0);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_coef_init_bwd_strided_f32(const float* G, int64_t slab, int64_t L, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                                   int64_t ds, int64_t dl, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b,
                                   float* gw_tail, int64_t ldw, float* g_bc, int64_t ld_gbc, int32_t zero_pad,
                                   bgStream stream) {
    if (H <= 0 || L < 0 || nf < 0 || nl < 0 || ML < 0 || ds < nf + ML + 5 || dl < nl || ldw < (zero_pad ? 8 : 4) || ld_gbc < 1)
        return BG_ERR_ARG;
    if (!G || !g_fe_w || !g_fe_b || !g_le_w || !g_le_b || !gw_tail || !g_bc) return BG_ERR_ARG;
    const int64_t n = (ds + dl) * 2 * H;
    if (n > 0x7fffffff || slab < n) return BG_ERR_SHAPE;
    k_d_coef_init_bwd<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(
        G, slab, (int)L, (int)H, (int)nf, (int)nl, (int)ML, (int)ds, (int)(ds + dl), g_fe_w, g_fe_b, g_le_w, g_le_b,
        gw_tail, ldw, g_bc, ld_gbc, zero_pad ? 1 : 0);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_d_coef_init_bwd_f32(const float* G, int64_t slab, int64_t L, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                           int64_t ds, int64_t dl, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b,
                           float* gw_tail, int64_t ldw, float* g_bc, bgStream stream) {
    return bg_d_coef_init_bwd_strided_f32(G, slab, L, H, nf, nl, ML, ds, dl, g_fe_w, g_fe_b, g_le_w, g_le_b, gw_tail, ldw,
                                          g_bc, 1, 0, stream);
}

int bg_decoder_bwd_f32(int64_t R, int64_t Rp, int64_t H2, int64_t Hd, const float* dv, const float* X, int64_t ldx,
                       const float* b1, const float* dgp, int64_t ldg, const float* w1, int64_t ldw1,
                       const float* w2, const float* gp, const float* add, int64_t lda, float* g_w1, int64_t ldgw1,
                       float* g_b1, float* g_w2, float* g_b2, float* dX, int64_t ldd, bgStream stream) {
    if (R <= 0 || Rp < 0 || H2 <= 0 || Hd <= 0) return BG_ERR_ARG;
    if (!dv || !X || !gp || !dX || (Rp > 0 && !dgp)) return BG_ERR_ARG;
    const bool grads = g_w1 || g_b1 || g_w2 || g_b2;                  // the four parameter gradients: all or none
    if (grads && (!g_w1 || !g_b1 || !g_w2 || !g_b2 || !w1 || !w2 || !b1 || ldw1 < H2 || ldgw1 < H2)) return BG_ERR_ARG;
    if (ldx < H2 || (Rp > 0 && ldg < H2) || ldd < H2 || (add && lda < H2)) return BG_ERR_ARG;
    if (R > 8192 || Rp > (1 << 20) || H2 > 1024 || Hd > (1 << 20)) return BG_ERR_SHAPE;     // a head's sizes, not a layer's (LDS: dv, column sums)
    DecBwdArgs a{(int)R, (int)Rp, (int)H2, (int)Hd, dv, X, ldx, b1, dgp, ldg, w1, ldw1, w2, gp, add, lda,
                 g_w1, ldgw1, g_b1, g_w2, g_b2, dX, ldd};
    if (R > 8192) return BG_ERR_SHAPE;                  // dv is kept in LDS
    int nb = (int)((Hd + DEC_T / 64 - 1) / (DEC_T / 64));               // one first-layer row per wave
    if (nb < 8) nb = 8;
    if (nb > 64) nb = 64;
    const size_t lds = ((size_t)(1 + DEC_Q) * H2 + DEC_T / 64 + R) * sizeof(float);
    k_decoder_bwd<<<nb, DEC_T, lds, (hipStream_t)stream>>>(a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_decoder_fwd_f32(int64_t R, int64_t H2, int64_t Hd, const float* X, int64_t ldx, const float* w1, int64_t ldw1,
                       const float* b1, const float* w2, const float* b2, float* val, float* gp, bgStream stream) {
    if (R <= 0 || H2 <= 0 || Hd <= 0 || !X || !w1 || !b1 || !w2 || !b2 || !val || !gp || ldx < H2 || ldw1 < H2) return BG_ERR_ARG;
    if (R > (1 << 20) || H2 > 1024 || Hd > (1 << 20)) return BG_ERR_SHAPE;
    DecFwdArgs a{(int)R, (int)H2, (int)Hd, X, ldx, w1, ldw1, b1, w2, b2, val, gp};
    int nb = (int)((R + DEC_T / 64 - 1) / (DEC_T / 64));                // one row per wave
    if (nb > 32) nb = 32;
    k_decoder_fwd<<<nb, DEC_T, (size_t)(1 + DEC_Q) * H2 * sizeof(float), (hipStream_t)stream>>>(a);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// Two decoders of the same widths on their own rows in one launch (the critic's story decoder and building decoder
// do not depend on each other): the results of two bg_decoder_fwd_f32 calls, bit for bit
int bg_decoder_fwd_pair_f32(int64_t H2, int64_t Hd, int64_t R0, const float* X0, int64_t ldx0, const float* w1_0, int64_t ldw1_0,
                            const float* b1_0, const float* w2_0, const float* b2_0, float* val0, float* gp0,
                            int64_t R1, const float* X1, int64_t ldx1, const float* w1_1, int64_t ldw1_1,
                            const float* b1_1, const float* w2_1, const float* b2_1, float* val1, float* gp1, bgStream stream) {
    if (R0 <= 0 || R1 <= 0 || H2 <= 0 || Hd <= 0 || !X0 || !w1_0 || !b1_0 || !w2_0 || !b2_0 || !val0 || !gp0 || !X1 || !w1_1 ||
        !b1_1 || !w2_1 || !b2_1 || !val1 || !gp1 || ldx0 < H2 || ldw1_0 < H2 || ldx1 < H2 || ldw1_1 < H2) return BG_ERR_ARG;
    if (R0 > (1 << 20) || R1 > (1 << 20) || H2 > 1024 || Hd > (1 << 20)) return BG_ERR_SHAPE;
    DecFwdArgs a0{(int)R0, (int)H2, (int)Hd, X0, ldx0, w1_0, ldw1_0, b1_0, w2_0, b2_0, val0, gp0};
    DecFwdArgs a1{(int)R1, (int)H2, (int)Hd, X1, ldx1, w1_1, ldw1_1, b1_1, w2_1, b2_1, val1, gp1};
    auto blocks = [](int64_t R) { int nb = (int)((R + DEC_T / 64 - 1) / (DEC_T / 64)); return nb > 32 ? 32 : nb; };   // as the single form
    const int nb0 = blocks(R0), nb1 = blocks(R1);
    k_decoder_fwd_pair<<<nb0 + nb1, DEC_T, (size_t)(1 + DEC_Q) * H2 * sizeof(float), (hipStream_t)stream>>>(a0, a1, nb0);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
