// Voxel -> program cross attention: tanh logits over cross edges (CSR by voxel), the segment
// Gumbel-softmax pointer with its first-index arg-max, and their backward kernels.
#include "common.h"

namespace {

constexpr int kMaxCPL = 8;   // channels per lane: H <= 512

// One wave per voxel.  Lane l owns channels l, l+64, ... of the voxel row; every cross edge of
// the voxel costs one program-row read (L2-resident: a story has a handful of program nodes).
template <int CPL>
__global__ __launch_bounds__(256) void k_cross_logits(const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb,
                                                      const float* __restrict__ theta,
                                                      const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      float* __restrict__ att, int64_t n_vox) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float th[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) th[j] = theta[lane + 64 * j];
    for (int64_t v = wave_id; v < n_vox; v += n_waves) {
        float av[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) av[j] = A[v * lda + lane + 64 * j];
        const int32_t b = rowptr[v], e = rowptr[v + 1];
        for (int32_t k = b; k < e; ++k) {
            const float* bp = B + (int64_t)col[k] * ldb;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc = fmaf(th[j], bg_tanh(av[j] + bp[lane + 64 * j]), acc);
            acc = bg_wave_sum(acc);
            if (lane == 0) att[k] = acc;
        }
    }
}

template <int CPL>
__global__ __launch_bounds__(256) void k_cross_logits_bwd(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          const float* __restrict__ theta,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ dpre, float* __restrict__ dA, int64_t ldda,
                                                          float* __restrict__ dtheta_partial, int64_t n_vox, int H) {
    __shared__ float red[4][64 * kMaxCPL];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float th[CPL], dth[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { th[j] = theta[lane + 64 * j]; dth[j] = 0.f; }
    for (int64_t v = wave_id; v < n_vox; v += n_waves) {
        float av[CPL], da[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) { av[j] = A[v * lda + lane + 64 * j]; da[j] = 0.f; }
        const int32_t b = rowptr[v], e = rowptr[v + 1];
        for (int32_t k = b; k < e; ++k) {
            const float* bp = B + (int64_t)col[k] * ldb;
            const float gk = g[k];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                float t = bg_tanh(av[j] + bp[lane + 64 * j]);
                float d = gk * th[j] * (1.f - t * t);
                dpre[(int64_t)k * H + lane + 64 * j] = d;
                da[j] += d;
                dth[j] = fmaf(gk, t, dth[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) dA[v * ldda + lane + 64 * j] = da[j];
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) red[wave][lane + 64 * j] = dth[j];
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += 256)
        dtheta_partial[(int64_t)blockIdx.x * H + h] = red[0][h] + red[1][h] + red[2][h] + red[3][h];
}

// One thread per segment; segments are short (program nodes of one story).
__global__ void k_seg_gumbel_softmax(const float* __restrict__ att, const float* __restrict__ expn,
                                     const int32_t* __restrict__ rowptr, float tau,
                                     float* __restrict__ soft, float* __restrict__ hard,
                                     int32_t* __restrict__ arg, int64_t n_seg) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_seg) return;
    const int32_t b = rowptr[v], e = rowptr[v + 1];
    if (b >= e) { arg[v] = -1; return; }
    float m = -3.402823466e+38f;
    for (int32_t k = b; k < e; ++k) {
        float y = (att[k] + (-logf(expn[k]))) / tau;
        soft[k] = y;                       // stash y
        m = fmaxf(m, y);
    }
    float s = 0.f;
    for (int32_t k = b; k < e; ++k) {
        float ex = expf(soft[k] - m);
        soft[k] = ex;
        s += ex;
    }
    const float den = s + 1e-16f;
    float best = -3.402823466e+38f; int32_t ba = b;
    for (int32_t k = b; k < e; ++k) {
        float o = soft[k] / den;
        soft[k] = o;
        if (o > best) { best = o; ba = k; }
    }
    arg[v] = ba;
    for (int32_t k = b; k < e; ++k) {
        float o = soft[k];
        float h = (k == ba) ? 1.f : 0.f;
        hard[k] = (h - o) + o;
    }
}

__global__ void k_seg_softmax_bwd(const float* __restrict__ soft, const float* __restrict__ g,
                                  const int32_t* __restrict__ rowptr, float tau,
                                  float* __restrict__ d_att, int64_t n_seg) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_seg) return;
    const int32_t b = rowptr[v], e = rowptr[v + 1];
    float dot = 0.f;
    for (int32_t k = b; k < e; ++k) dot = fmaf(soft[k], g[k], dot);
    for (int32_t k = b; k < e; ++k) d_att[k] = soft[k] * (g[k] - dot) / tau;
}

__global__ void k_seg_argmax(const float* __restrict__ logit, const int32_t* __restrict__ rowptr,
                             int32_t* __restrict__ arg, int64_t n_seg) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_seg) return;
    const int32_t b = rowptr[v], e = rowptr[v + 1];
    float best = -3.402823466e+38f; int32_t ba = -1;
    for (int32_t k = b; k < e; ++k) {
        float o = logit[k];
        if (o > best || ba < 0) { best = o; ba = k; }
    }
    arg[v] = ba;
}

// Two-way Gumbel-softmax of the attention's voxel mask (reference models.py:109-111):
//   z = logits - log(expn);  soft = softmax(z);  hard = (onehot(argmax soft) - soft) + soft;
// only column 0 of either is used downstream.  One thread per voxel.
__global__ void k_mask_gumbel(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ expn,
                              float* __restrict__ soft0, float* __restrict__ hard0, int64_t n) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float2 l = make_float2(logits[v * ldl], logits[v * ldl + 1]);
    const float2 e = *reinterpret_cast<const float2*>(expn + 2 * v);
    const float z0 = l.x - logf(e.x), z1 = l.y - logf(e.y);
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    const float sum = e0 + e1;
    const float s0 = e0 / sum, s1 = e1 / sum;
    const float one0 = s0 >= s1 ? 1.f : 0.f;            // first maximum wins a tie
    soft0[v] = s0;
    hard0[v] = (one0 - s0) + s0;
}

// d logits of  <g_soft, soft0> + <g_hard, hard0>  (the hard output passes its gradient straight through)
__global__ void k_mask_gumbel_bwd(const float* __restrict__ soft0, const float* __restrict__ g_soft,
                                  const float* __restrict__ g_hard, const float* __restrict__ g_more,
                                  float* __restrict__ d_logits, int64_t ldd, int64_t n) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float s = soft0[v];
    const float g = (g_soft ? g_soft[v] : 0.f) + (g_hard ? g_hard[v] : 0.f) + (g_more ? g_more[v] : 0.f);
    const float d = g * s * (1.f - s);
    if (ldd == 2) *reinterpret_cast<float2*>(d_logits + 2 * v) = make_float2(d, -d);
    else *reinterpret_cast<float4*>(d_logits + ldd * v) = make_float4(d, -d, 0.f, 0.f);   // a 4-column slot of a wider row
}

// Forward-only fusion of one Attention call (reference models.py:104-126) for passes that build no
// autograd graph (inference, the generator forward of critic-only iterations).  One wave per voxel does
// what k_mask_gumbel, k_cross_logits, k_seg_gumbel_softmax, the weighted program gather and the
// final v + soft_mask * gathered do as five launches: lane j keeps the logit of the voxel's j-th cross
// edge (degree <= 64, checked by the launcher's caller), max / sum / arg-max run over lanes in slot
// order, so every output equals the unfused kernels' (same operation order).
template <int CPL>
__global__ __launch_bounds__(256) void k_attention_fwd(const float* __restrict__ A, int64_t lda,        // NN_v(v) [Nv, H]
                                                       const float* __restrict__ logits, int64_t ldl,  // mask logits [Nv, 2]
                                                       const float* __restrict__ B, int64_t ldb,        // NN_p(x) [Np, H]
                                                       const float* __restrict__ X, int64_t ldx,        // program features [Np, H]
                                                       const float* __restrict__ V, int64_t ldv,        // voxel features [Nv, H]
                                                       const float* __restrict__ theta,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ exp_mask,              // [Nv, 2]
                                                       const float* __restrict__ exp_att,               // [Ec]
                                                       float tau,
                                                       float* __restrict__ soft0, float* __restrict__ hard0,   // [Nv]
                                                       float* __restrict__ soft, float* __restrict__ hard,     // [Ec]
                                                       float* __restrict__ out, int64_t ldo, int64_t n_vox,
                                                       const float* __restrict__ Wd, int64_t ldwd,      // optional [2, H]: logits = V Wd^T + bd
                                                       const float* __restrict__ bd) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float th[CPL], wd0[CPL], wd1[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        th[j] = theta[lane + 64 * j];
        wd0[j] = Wd ? Wd[lane + 64 * j] : 0.f;
        wd1[j] = Wd ? Wd[ldwd + lane + 64 * j] : 0.f;
    }
    for (int64_t v = wave_id; v < n_vox; v += n_waves) {
        // A voxel's time is its chain of DEPENDENT memory round trips, so the loads are grouped by what they depend
        // on: (1) everything addressed by v alone, (2) the cross edges' program ids and noise, (3) the NN_p rows AND the
        // program feature rows of the first four edges (both need only the ids).  Same arithmetic in the same order.
        float vv[CPL], av[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) { vv[j] = V[v * ldv + lane + 64 * j]; av[j] = A[v * lda + lane + 64 * j]; }
        // (the same row for the whole wave: keep its bounds in scalar registers -- the loops over the edges become scalar
        // branches and the lane indices of bg_readlane scalar operands)
        const int32_t b = __builtin_amdgcn_readfirstlane(rowptr[v]), e = __builtin_amdgcn_readfirstlane(rowptr[v + 1]);
        const float em0 = exp_mask[2 * v], em1 = exp_mask[2 * v + 1];
        float l0 = 0.f, l1 = 0.f;
        if (!Wd) { l0 = logits[v * ldl]; l1 = logits[v * ldl + 1]; }
        const int deg = e - b;
        const int32_t my_col = lane < deg ? col[b + lane] : 0;
        const float my_exp = lane < deg ? exp_att[b + lane] : 1.f;
        float bv[4][CPL], xv0[4][CPL];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t c = bg_readlane(my_col, i < deg ? i : 0);
            const float* bp = B + (int64_t)c * ldb;
            const float* xp = X + (int64_t)c * ldx;
#pragma unroll
            for (int j = 0; j < CPL; ++j) { bv[i][j] = bp[lane + 64 * j]; xv0[i][j] = xp[lane + 64 * j]; }
        }
        // ---- mask logits: given, or the composed two-row decoder applied to the voxel's features here
        if (Wd) {
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) { d0 = fmaf(vv[j], wd0[j], d0); d1 = fmaf(vv[j], wd1[j], d1); }
            l0 = bg_wave_sum(d0) + bd[0]; l1 = bg_wave_sum(d1) + bd[1];
        }
        // ---- two-way mask (k_mask_gumbel)
        const float z0 = l0 - logf(em0), z1 = l1 - logf(em1);
        const float zm = fmaxf(z0, z1);
        const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
        const float esum = e0 + e1;
        const float s0 = e0 / esum, s1 = e1 / esum;
        const float one0 = s0 >= s1 ? 1.f : 0.f;
        if (lane == 0) { soft0[v] = s0; hard0[v] = (one0 - s0) + s0; }
        // ---- logits of the voxel's cross edges (k_cross_logits); lane j keeps edge j's y = (att + gumbel) / tau.
        // The program rows of four edges are requested together, so a voxel pays one memory round trip per four
        // edges instead of two dependent ones (index, then row) per edge.
        float y = -3.402823466e+38f;
        for (int k0 = 0; k0 < deg; k0 += 4) {
            if (k0 > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int32_t c = bg_readlane(my_col, k0 + i < deg ? k0 + i : k0);
                    const float* bp = B + (int64_t)c * ldb;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) bv[i][j] = bp[lane + 64 * j];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (k0 + i < deg) {                                  // wave-uniform
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc = fmaf(th[j], bg_tanh(av[j] + bv[i][j]), acc);
                    acc = bg_wave_sum(acc);
                    if (lane == k0 + i) y = (acc + (-logf(my_exp))) / tau;
                }
            }
        }
        // ---- segment Gumbel-softmax (k_seg_gumbel_softmax): max, then exp / sum in slot order
        const float m = bg_wave_max(y);
        const float ex = lane < deg ? expf(y - m) : 0.f;
        float sum = 0.f;
        for (int j = 0; j < deg; ++j) sum += bg_readlane(ex, j);
        const float o = ex / (sum + 1e-16f);
        float best = -3.402823466e+38f; int ba = 0;
        for (int j = 0; j < deg; ++j) {
            const float oj = bg_readlane(o, j);
            if (oj > best) { best = oj; ba = j; }
        }
        if (lane < deg) {
            soft[b + lane] = o;
            hard[b + lane] = ((lane == ba ? 1.f : 0.f) - o) + o;
        }
        // ---- v + soft_mask * sum_k soft_k x[col_k]   (weighted gather in slot order, then addcmul)
        float g[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) g[j] = 0.f;
        for (int k0 = 0; k0 < deg; k0 += 4) {
            float xv[4][CPL];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (k0 == 0) {                                       // wave-uniform: the rows fetched at the top
#pragma unroll
                    for (int j = 0; j < CPL; ++j) xv[i][j] = xv0[i][j];
                } else {
                    const int32_t c = bg_readlane(my_col, k0 + i < deg ? k0 + i : k0);
                    const float* xp = X + (int64_t)c * ldx;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) xv[i][j] = xp[lane + 64 * j];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (k0 + i < deg) {
                    const float w = bg_readlane(o, k0 + i);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) g[j] += w * xv[i][j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) out[v * ldo + lane + 64 * j] = vv[j] + s0 * g[j];
    }
}

// Head of one Attention call's backward (ops.AttentionNode), one wave per voxel -- what a weighted gather, two
// sampled dot products, an elementwise product, the additions of the incoming edge gradients, the segment
// softmax's backward and the mask's backward do as seven or eight launches.  With out = v + s0 * gathered,
// gathered = sum_k soft_k x[col_k]:
//   gg      = g_out * s0                                     (written: the program-side gather-transpose reads it)
//   g_k     = <gg, x[col_k]> (+ g_soft_k + g_hard_k)         gradient at soft_k (hard passes its own straight on)
//   d_att_k = soft_k (g_k - sum_j soft_j g_j) / tau          (k_seg_softmax_bwd)
//   d_mask  = (g_soft0 + g_hard0 + <g_out, gathered>) s0 (1 - s0)  -> (d, -d, 0, 0) into a 4-column slot (k_mask_gumbel_bwd)
// Slots of a voxel are in edge order (checked plan); degree <= 64.
template <int CPL>
__global__ __launch_bounds__(256) void k_attention_bwd_head(const float* __restrict__ G, int64_t ldg,       // g_out [Nv, H]
                                                            const float* __restrict__ X, int64_t ldx,       // program features
                                                            const float* __restrict__ soft0, const float* __restrict__ soft,
                                                            const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ g_soft, const float* __restrict__ g_hard,
                                                            const float* __restrict__ g_soft0, const float* __restrict__ g_hard0,
                                                            float tau, float* __restrict__ gg, int64_t ldgg,
                                                            float* __restrict__ d_att, float* __restrict__ d_mask, int64_t ldd,
                                                            int64_t n_vox) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vox) return;                                       // wave-uniform
    const float s0 = soft0[v];
    float go[CPL], gs[CPL], gath[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        go[j] = G[v * ldg + lane + 64 * j];
        gs[j] = go[j] * s0;
        gg[v * ldgg + lane + 64 * j] = gs[j];
        gath[j] = 0.f;
    }
    const int32_t b = __builtin_amdgcn_readfirstlane(rowptr[v]), e = __builtin_amdgcn_readfirstlane(rowptr[v + 1]);
    const int deg = e - b;
    const int32_t my_col = lane < deg ? col[b + lane] : 0;
    const float my_soft = lane < deg ? soft[b + lane] : 0.f;
    float my_g = 0.f;
    if (lane < deg) {
        if (g_soft) my_g += g_soft[b + lane];
        if (g_hard) my_g += g_hard[b + lane];
    }
    for (int k0 = 0; k0 < deg; k0 += 4) {
        float xv[4][CPL];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t c = bg_readlane(my_col, k0 + i < deg ? k0 + i : k0);
            const float* xp = X + (int64_t)c * ldx;
#pragma unroll
            for (int j = 0; j < CPL; ++j) xv[i][j] = xp[lane + 64 * j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k0 + i < deg) {                                   // wave-uniform
                const float w = bg_readlane(my_soft, k0 + i);
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    gath[j] = fmaf(w, xv[i][j], gath[j]);
                    acc = fmaf(gs[j], xv[i][j], acc);
                }
                acc = bg_wave_sum(acc);
                if (lane == k0 + i) my_g += acc;
            }
        }
    }
    // segment softmax backward over the voxel's edges, in slot order
    float dot = 0.f;
    for (int k = 0; k < deg; ++k) dot = fmaf(bg_readlane(my_soft, k), bg_readlane(my_g, k), dot);
    if (lane < deg) d_att[b + lane] = my_soft * (my_g - dot) / tau;
    // mask
    float gm = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) gm = fmaf(go[j], gath[j], gm);
    gm = bg_wave_sum(gm);
    if (lane == 0) {
        float g = gm;
        if (g_soft0) g += g_soft0[v];
        if (g_hard0) g += g_hard0[v];
        const float d = g * s0 * (1.f - s0);
        *reinterpret_cast<float4*>(d_mask + v * ldd) = make_float4(d, -d, 0.f, 0.f);
    }
}

}  // namespace

extern "C" {

int bg_cross_logits_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* theta,
                        const int32_t* rowptr, const int32_t* col, float* att,
                        int64_t n_vox, int64_t H, bgStream stream) {
    if (n_vox < 0 || H <= 0) return BG_ERR_ARG;
    if (n_vox == 0) return BG_OK;
    if (!a || !b || !theta || !rowptr || !col || !att || lda < H || ldb < H) return BG_ERR_ARG;
    if (H % 64 != 0 || H > 64 * kMaxCPL) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    int grid = bg_grid_for(n_vox, 4, 256 * 16);
    switch (H / 64) {
        case 1: k_cross_logits<1><<<grid, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, att, n_vox); break;
        case 2: k_cross_logits<2><<<grid, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, att, n_vox); break;
        case 4: k_cross_logits<4><<<grid, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, att, n_vox); break;
        case 8: k_cross_logits<8><<<grid, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, att, n_vox); break;
        default: return BG_ERR_SHAPE;
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_cross_logits_bwd_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* theta,
                            const int32_t* rowptr, const int32_t* col, const float* g,
                            float* dpre, float* da, int64_t ldda, float* dtheta_partial, int32_t n_blocks,
                            int64_t n_vox, int64_t H, bgStream stream) {
    if (n_vox < 0 || H <= 0 || n_blocks <= 0) return BG_ERR_ARG;
    if (!a || !b || !theta || !rowptr || !col || !g || !dpre || !da || !dtheta_partial) return BG_ERR_ARG;
    if (lda < H || ldb < H || ldda < H) return BG_ERR_ARG;
    if (H % 64 != 0 || H > 64 * kMaxCPL) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (H / 64) {
        case 1: k_cross_logits_bwd<1><<<n_blocks, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, g, dpre, da, ldda, dtheta_partial, n_vox, (int)H); break;
        case 2: k_cross_logits_bwd<2><<<n_blocks, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, g, dpre, da, ldda, dtheta_partial, n_vox, (int)H); break;
        case 4: k_cross_logits_bwd<4><<<n_blocks, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, g, dpre, da, ldda, dtheta_partial, n_vox, (int)H); break;
        case 8: k_cross_logits_bwd<8><<<n_blocks, 256, 0, s>>>(a, lda, b, ldb, theta, rowptr, col, g, dpre, da, ldda, dtheta_partial, n_vox, (int)H); break;
        default: return BG_ERR_SHAPE;
    }
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_seg_gumbel_softmax_f32(const float* att, const float* expn, const int32_t* rowptr, float tau,
                              float* soft, float* hard, int32_t* arg, int64_t n_seg, bgStream stream) {
    if (n_seg < 0 || tau == 0.f) return BG_ERR_ARG;
    if (n_seg == 0) return BG_OK;
    if (!att || !expn || !rowptr || !soft || !hard || !arg) return BG_ERR_ARG;
    k_seg_gumbel_softmax<<<bg_grid_for(n_seg, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(att, expn, rowptr, tau, soft, hard, arg, n_seg);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_seg_softmax_bwd_f32(const float* soft, const float* g, const int32_t* rowptr, float tau,
                           float* d_att, int64_t n_seg, bgStream stream) {
    if (n_seg < 0 || tau == 0.f) return BG_ERR_ARG;
    if (n_seg == 0) return BG_OK;
    if (!soft || !g || !rowptr || !d_att) return BG_ERR_ARG;
    k_seg_softmax_bwd<<<bg_grid_for(n_seg, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(soft, g, rowptr, tau, d_att, n_seg);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_seg_argmax_f32(const float* logit, const int32_t* rowptr, int32_t* arg, int64_t n_seg, bgStream stream) {
    if (n_seg < 0) return BG_ERR_ARG;
    if (n_seg == 0) return BG_OK;
    if (!logit || !rowptr || !arg) return BG_ERR_ARG;
    k_seg_argmax<<<bg_grid_for(n_seg, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(logit, rowptr, arg, n_seg);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_mask_gumbel_f32(const float* logits, int64_t ldl, const float* expn, float* soft0, float* hard0, int64_t n,
                       bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!logits || !expn || !soft0 || !hard0 || ldl < 2) return BG_ERR_ARG;
    k_mask_gumbel<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(logits, ldl, expn, soft0, hard0, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_mask_gumbel_bwd_f32(const float* soft0, const float* g_soft, const float* g_hard, const float* g_more,
                           float* d_logits, int64_t ldd, int64_t n, bgStream stream) {
    if (n < 0) return BG_ERR_ARG;
    if (n == 0) return BG_OK;
    if (!soft0 || !d_logits || ldd < 2) return BG_ERR_ARG;
    if (ldd != 2 && (ldd % 4 != 0 || !bg_aligned16(d_logits))) return BG_ERR_ALIGN;
    k_mask_gumbel_bwd<<<bg_grid_for(n, 256, 0x7fffffff), 256, 0, (hipStream_t)stream>>>(soft0, g_soft, g_hard, g_more,
                                                                                         d_logits, ldd, n);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_attention_fwd_f32(const float* a, int64_t lda, const float* logits, int64_t ldl,
                         const float* b, int64_t ldb, const float* x, int64_t ldx,
                         const float* v, int64_t ldv, const float* theta,
                         const int32_t* rowptr, const int32_t* col,
                         const float* exp_mask, const float* exp_att, float tau,
                         float* soft0, float* hard0, float* soft, float* hard,
                         float* out, int64_t ldo, int64_t n_vox, int64_t H,
                         const float* wd, int64_t ldwd, const float* bd, bgStream stream) {
    if (n_vox < 0 || H <= 0 || tau == 0.f) return BG_ERR_ARG;
    if (n_vox == 0) return BG_OK;
    if (!a || !b || !x || !v || !theta || !rowptr || !col || !exp_mask || !exp_att || !soft0 || !hard0 ||
        !soft || !hard || !out || lda < H || ldb < H || ldx < H || ldv < H || ldo < H)
        return BG_ERR_ARG;
    if (wd ? (!bd || ldwd < H) : (!logits || ldl < 2)) return BG_ERR_ARG;   // logits given, or the decoder to apply
    if (H % 64 != 0 || H > 64 * kMaxCPL) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    int grid = bg_grid_for(n_vox, 4, 256 * 16);
#define BG_ATT_CASE(C)                                                                                          \
    case C:                                                                                                     \
        k_attention_fwd<C><<<grid, 256, 0, s>>>(a, lda, logits, ldl, b, ldb, x, ldx, v, ldv, theta, rowptr, col, \
                                                exp_mask, exp_att, tau, soft0, hard0, soft, hard, out, ldo, n_vox, \
                                                wd, ldwd, bd);                                                  \
        break;
    switch ((int)(H / 64)) {
        BG_ATT_CASE(1) BG_ATT_CASE(2) BG_ATT_CASE(3) BG_ATT_CASE(4)
        BG_ATT_CASE(5) BG_ATT_CASE(6) BG_ATT_CASE(7) BG_ATT_CASE(8)
    }
#undef BG_ATT_CASE
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_attention_bwd_head_f32(const float* g_out, int64_t ldg, const float* x, int64_t ldx,
                              const float* soft0, const float* soft, const int32_t* rowptr, const int32_t* col,
                              const float* g_soft, const float* g_hard, const float* g_soft0, const float* g_hard0,
                              float tau, float* gg, int64_t ldgg, float* d_att, float* d_mask, int64_t ldd,
                              int64_t n_vox, int64_t H, bgStream stream) {
    if (n_vox < 0 || H <= 0 || tau == 0.f) return BG_ERR_ARG;
    if (n_vox == 0) return BG_OK;
    if (!g_out || !x || !soft0 || !soft || !rowptr || !col || !gg || !d_att || !d_mask || ldg < H || ldx < H ||
        ldgg < H || ldd < 4) return BG_ERR_ARG;
    if (H % 64 != 0 || H > 64 * kMaxCPL) return BG_ERR_SHAPE;
    if (!bg_aligned16(d_mask) || ldd % 4 != 0) return BG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int grid = bg_grid_for(n_vox, 4, 0x7fffffff);
#define BG_ATTB_CASE(C)                                                                                          \
    case C:                                                                                                      \
        k_attention_bwd_head<C><<<grid, 256, 0, s>>>(g_out, ldg, x, ldx, soft0, soft, rowptr, col, g_soft, g_hard, \
                                                     g_soft0, g_hard0, tau, gg, ldgg, d_att, d_mask, ldd, n_vox); \
        break;
    switch ((int)(H / 64)) {
        BG_ATTB_CASE(1) BG_ATTB_CASE(2) BG_ATTB_CASE(3) BG_ATTB_CASE(4)
        BG_ATTB_CASE(5) BG_ATTB_CASE(6) BG_ATTB_CASE(7) BG_ATTB_CASE(8)
    }
#undef BG_ATTB_CASE
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"
