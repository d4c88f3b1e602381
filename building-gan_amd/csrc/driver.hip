// Whole passes as one host call: the launch sequences of the graph-free paths, issued from C++ instead of from the
// interpreter (a training step over a stream of differently-shaped batches cannot replay a captured hipGraph and
// is bound by host launch time: ~17 us of Python per kernel against ~2 us here).  Every step calls the same
// extern "C" launcher with the same arguments as the op-by-op host path, so results are bit-identical.
#include "common.h"
#include <stdio.h>
#include <initializer_list>

namespace {

struct Arena {
    float* base; int64_t used, cap;
    float* take(int64_t n) {
        int64_t a = (used + 63) / 64 * 64;          // 256-byte pieces
        used = a + n;
        return base ? base + a : nullptr;
    }
};

struct GenBuffers {
    float *x0, *x1, *both, *aggr, *nnp, *v0, *v1, *gm, *a, *posb, *pq, *term, *s0, *h0, *sa, *ha;
};

GenBuffers carve(Arena& ar, int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H) {
    GenBuffers b;
    b.x0 = ar.take(Np * H); b.x1 = ar.take(Np * H); b.both = ar.take(2 * Np * H); b.aggr = ar.take(Np * H);
    b.nnp = ar.take(Np * H);
    b.v0 = ar.take(Nv * H); b.v1 = ar.take(Nv * H); b.gm = ar.take(Nv * H); b.a = ar.take(Nv * H);
    b.posb = ar.take(Nv * H); b.pq = ar.take(S * 2 * H); b.term = ar.take(S * H);
    b.s0 = ar.take(Nv); b.h0 = ar.take(Nv); b.sa = ar.take(Ec); b.ha = ar.take(Ec);
    return b;
}

// bg_gemm_f32 with up to three A segments, nn.Linear weight layout (transB = 1)
int linear(int64_t M, int64_t N, int n_seg, const float* a0, int64_t k0, const float* a1, int64_t k1, const float* a2,
           int64_t k2, int64_t lda, const float* W, int64_t ldw, float* C, const float* bias, const float* row_scale,
           int act, const float* residual, const float* pre_add, int64_t ldp, const int32_t* pre_rows,
           const float* acc_init, bgStream s) {
    const float* A[4] = {a0, a1, a2, nullptr};
    int64_t ld[4] = {lda, lda, lda, 0};
    if (n_seg == 1) ld[0] = lda;
    int64_t Ks[4] = {k0, k1, k2, 0};
    return bg_gemm_f32(0, 1, M, N, n_seg, A, ld, Ks, W, ldw, C, N, bias, row_scale, act, 0.01f,
                       residual, residual ? N : 0, pre_add, ldp, pre_rows, acc_init, acc_init ? N : 0, nullptr, 0, s);
}

}  // namespace

// BG_DRIVER_SYNC=1 (diagnosis, -DBG_TUNING_KNOBS builds only): every step is named on stderr before it is issued and the stream is drained after
// it, so a faulting launch is the last line printed.
static int g_step_sync = -1;
static int step_sync() {
    if (g_step_sync < 0) {
        g_step_sync = bg_tune_int("BG_DRIVER_SYNC", 0) == 1 ? 1 : 0;
    }
    return g_step_sync;
}

#define BG_TRY(call)                                                                    \
    do {                                                                                \
        if (step_sync()) { fprintf(stderr, "[bg driver] %s\n", #call); fflush(stderr); } \
        int rc__ = (call);                                                              \
        if (rc__ != BG_OK) return rc__;                                                 \
        if (step_sync() && hipStreamSynchronize((hipStream_t)s) != hipSuccess) {        \
            fprintf(stderr, "[bg driver] the step above failed\n");                     \
            return BG_ERR_LAUNCH;                                                       \
        }                                                                               \
    } while (0)

extern "C" int64_t bg_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int64_t)sizeof(bgGeneratorWeights);
        case 1: return (int64_t)sizeof(bgGeneratorPlan);
        case 2: return (int64_t)sizeof(bgPlanIn);
        case 3: return (int64_t)sizeof(bgCriticHead);
        case 4: return (int64_t)sizeof(bgGeneratorBackPlan);
        default: return -1;
    }
}

extern "C" int64_t bg_generator_forward_workspace(int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H) {
    if (Np < 0 || Nv < 0 || S < 0 || Ec < 0 || H < 0) return -1;
    Arena ar{nullptr, 0, 0};
    carve(ar, Np, Nv, S, Ec, H);
    return ar.used + 64;
}

extern "C" int bg_generator_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p,
                                        const float* program_in, const float* voxel_in, const float* pdm4,
                                        const float* cls, const float* const* exp_mask, const float* const* exp_att,
                                        int64_t n_calls, float* workspace, int64_t workspace_floats,
                                        float* out, float* soft_out, float* mask_hard, float* mask_soft,
                                        float* att_hard, float* att_soft, int64_t* idx, bgStream s) {
    if (!w || !p || !program_in || !voxel_in || !pdm4 || !cls || !exp_mask || !exp_att || !workspace || !out ||
        !soft_out || !mask_hard || !mask_soft || !att_hard || !att_soft || !idx) return BG_ERR_ARG;
    const int64_t H = w->H, Np = p->Np, Nv = p->Nv, S = p->S, Ec = p->Ec, L = w->voxel_layer;
    if (H <= 0 || Np <= 0 || Nv <= 0 || S <= 0 || Ec <= 0 || L < 1 || w->program_layer < 0) return BG_ERR_ARG;
    if (n_calls != 1 + (L - 1) / 2) return BG_ERR_ARG;
    if (H % 64 != 0 || H > 256 || Nv > 65536 || w->n_class > 8 || w->n_class < 1) return BG_ERR_SHAPE;
    if (!p->vox_all_has || p->cx_max_len < 1 || p->cx_max_len > 64) return BG_ERR_SHAPE;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    GenBuffers b = carve(ar, Np, Nv, S, Ec, H);
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const int64_t kin = w->n_class + 1 + w->noise_dim;

    // ---- program GNN (models.py:83-92): encoder, then program_layer x (x + block(x)) with shared weights
    float *x = b.x0, *xn = b.x1;
    BG_TRY(linear(Np, H, 1, program_in, kin, nullptr, 0, nullptr, 0, kin, w->p_enc_w, kin, x, w->p_enc_b, nullptr,
                  BG_ACT_NONE, nullptr, nullptr, 0, nullptr, nullptr, s));
    for (int64_t l = 0; l < w->program_layer; ++l) {
        // rows [0, Np): mean over in-edges; rows [Np, 2 Np): target ratio x class-slot mean (one operator)
        BG_TRY(bg_spmm_f32(x, H, p->pp_rowptr, p->pp_col, nullptr, nullptr, p->pp_scale, BG_SCALE_MUL, nullptr,
                           BG_SCALE_NONE, b.both, H, 2 * Np, H, s));
        BG_TRY(linear(Np, H, 2, x, H, b.both, H, nullptr, 0, H, w->p_msg_w, 2 * H, b.aggr, w->p_msg_b, p->prog_has,
                      BG_ACT_NONE, nullptr, nullptr, 0, nullptr, nullptr, s));
        BG_TRY(linear(Np, H, 3, x, H, b.aggr, H, b.both + Np * H, H, H, w->p_upd_w, 3 * H, xn, w->p_upd_b, nullptr,
                      BG_ACT_LEAKY, x, nullptr, 0, nullptr, nullptr, s));
        float* t = x; x = xn; xn = t;
    }

    // ---- voxel GNN (models.py:185-203)
    float *v = b.v0, *vn = b.v1;
    BG_TRY(linear(Nv, H, 1, voxel_in, 4, nullptr, 0, nullptr, 0, 4, w->v_enc_w, 4, v, w->v_enc_b, nullptr, BG_ACT_NONE,
                  nullptr, w->pe, H, p->level, nullptr, s));                       // + pe[level] in the epilogue
    BG_TRY(linear(Np, H, 1, x, H, nullptr, 0, nullptr, 0, H, w->nnp_w, H, b.nnp, w->nnp_b, nullptr, BG_ACT_NONE, nullptr,
                  nullptr, 0, nullptr, nullptr, s));                               // NN_p(x): once for all calls
    int64_t call = 0;
    auto attention = [&](bool last) -> int {
        BG_TRY(linear(Nv, H, 1, v, H, nullptr, 0, nullptr, 0, H, w->Wcat, H, b.a, w->bcat, nullptr, BG_ACT_NONE, nullptr,
                      nullptr, 0, nullptr, nullptr, s));                           // NN_v(v)
        BG_TRY(bg_attention_fwd_f32(b.a, H, nullptr, 0, b.nnp, H, x, H, v, H, w->theta, p->cx_rowptr, p->cx_col,
                                    exp_mask[call], exp_att[call], -1.0f, last ? mask_soft : b.s0,
                                    last ? mask_hard : b.h0, last ? att_soft : b.sa, last ? att_hard : b.ha, vn, H, Nv, H,
                                    w->Wcat + H * H, H, w->bcat + H, s));
        float* t = v; v = vn; vn = t;
        ++call;
        return BG_OK;
    };
    BG_TRY(attention(n_calls == 1));
    // the block's position + bias term is the same in all of its calls: once
    BG_TRY(linear(Nv, H, 1, pdm4, 4, nullptr, 0, nullptr, 0, 4, w->Wc + 2 * H, w->ldwc, b.posb, w->bc, nullptr, BG_ACT_NONE,
                  nullptr, nullptr, 0, nullptr, nullptr, s));
    // the last attention call follows block i = L - 1 (even) or L - 2; blocks after it are never read
    int64_t last_att = -1;
    for (int64_t i = 0; i < L; ++i)
        if (i % 2 == 0 && i != 0) last_att = i;
    for (int64_t i = 0; i < L; ++i) {
        if (i > last_att) break;
        BG_TRY(bg_gather_story_f32(v, H, p->vx_rowptr, p->vx_col, p->vx_count, BG_SCALE_DIV, b.gm, H, Nv,
                                   p->st_rowptr, p->st_col, p->st_count, w->Wsb, 2 * H, H, b.pq, 2 * H, S, H, s));
        // the cross-story mean + the block product: one launch where the row-panel kernel takes the shape (the story
        // term is then formed while the product loads it), else the two launches
        const float* A2[4] = {v, b.gm, nullptr, nullptr};
        const int64_t ld2[4] = {H, H, 0, 0}, Ks2[4] = {H, H, 0, 0};
        const int rc_nbr = bg_gemm_nbr_f32(Nv, H, 2, A2, ld2, Ks2, w->Wc, w->ldwc, vn, H, nullptr, BG_ACT_LEAKY, 0.01f, v, H,
                                           b.pq, 2 * H, b.pq + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count,
                                           p->story_of_voxel, b.posb, H, nullptr, 0, s);
        if (rc_nbr == BG_ERR_SHAPE || rc_nbr == BG_ERR_ALIGN) {
            BG_TRY(bg_spmm_add_f32(b.pq + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count, BG_SCALE_DIV, nullptr, BG_SCALE_NONE,
                                   b.pq, 2 * H, nullptr, 0, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE, b.term, H, S, H, s));
            BG_TRY(linear(Nv, H, 2, v, H, b.gm, H, nullptr, 0, H, w->Wc, w->ldwc, vn, nullptr, nullptr, BG_ACT_LEAKY, v, b.term,
                          H, p->story_of_voxel, b.posb, s));
        } else {
            BG_TRY(rc_nbr);
        }
        float* t = v; v = vn; vn = t;
        if (i % 2 == 0 && i != 0) BG_TRY(attention(call == n_calls - 1));
    }
    if (call != n_calls) return BG_ERR_ARG;
    // ---- output stage (models.py:128-141)
    return bg_construct_output_f32(mask_hard, att_hard, att_soft, p->cx_rowptr, p->cx_col, cls, w->n_class, w->n_class,
                                   out, soft_out, idx, Nv, s);
}

// ---------------------------------------------------------------- collapsed discriminator: coefficient recurrence
namespace {
int64_t coef_splits(int64_t M, int64_t R) {          // as ops.gemm_tn_raw: M x (M | M) output, reduction R
    const int64_t tiles = ((M + 63) / 64) * 2 * ((M + 63) / 64);
    const int64_t steps = (R + 31) / 32;
    int64_t sp = (512 + tiles - 1) / tiles;
    if (sp > steps) sp = steps;
    if (sp > 256) sp = 256;
    return sp < 1 ? 1 : sp;
}
}  // namespace

namespace {
// bc / g_bc may be a strided column (ld_bc, ld_gbc) and gW may have any pitch: the composed weights of
// bg_d_table_* live in [2H, 4H+8] rows [Wc | bc | pad]
int d_coef_forward_impl(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b, const float* pe,
                        const float* Wc, int64_t ldw, const float* bc, int64_t ld_bc, int64_t H, int64_t nf, int64_t nl,
                        int64_t ML, int64_t ds, int64_t dl, int64_t L, int32_t basis, float* xall, float* efull, bgStream s) {
    if (!fe_w || !fe_b || !le_w || !le_b || !pe || !Wc || !bc || !xall || !efull) return BG_ERR_ARG;
    if (H <= 0 || L < 0 || ds < nf + ML + 5 || dl < nl || ldw < 4 * H + 4) return BG_ERR_ARG;
    if (basis != BG_CHAIN_POWER && basis != BG_CHAIN_CHEB) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, du = ds + dl, N = (L + 1) * du, S = N + du;
    {   // both workspaces start at zero: one launch
        void* const dst[2] = {xall, efull};
        const void* const src[2] = {nullptr, nullptr};
        const int64_t wd[2] = {(int64_t)sizeof(float) * (L + 1) * S * H2, (int64_t)sizeof(float) * N * H2}, one[2] = {1, 1};
        BG_TRY(bg_fill_copy_batch(2, dst, wd, src, wd, wd, one, s));
    }
    BG_TRY(bg_d_coef_init_strided_f32(fe_w, fe_b, le_w, le_b, pe, Wc, ldw, bc, ld_bc, H, nf, nl, ML, ds, dl, xall + du * H2,
                                      efull, s));
    for (int64_t m = 0; m < L; ++m) {
        const int64_t rows = N < (m + 2) * du ? N : (m + 2) * du;       // blocks above m+1 are still zero
        float* cur = xall + m * S * H2;
        float* nxt = xall + (m + 1) * S * H2 + du * H2;
        if (basis == BG_CHAIN_CHEB) {
            // second operand = x * X^m, formed while the product loads it (no separate shift launch)
            BG_TRY(bg_gemm_cheb_f32(1, rows, H2, cur + du * H2, H2, H2, du, L + 1, 0, Wc, ldw, nxt, H2, cur + du * H2, H2,
                                    efull, H2, s));
            continue;
        }
        const float* A[4] = {cur + du * H2, cur, nullptr, nullptr};     // monomials: one block down = a row offset
        const int64_t ld[4] = {H2, H2, 0, 0}, Ks[4] = {H2, H2, 0, 0};
        BG_TRY(bg_gemm_f32(0, 1, rows, H2, 2, A, ld, Ks, Wc, ldw, nxt, H2, nullptr, nullptr,
                           BG_ACT_NONE, 0.01f, cur + du * H2, H2, efull, H2, nullptr, nullptr, 0, nullptr, 0, s));
    }
    return BG_OK;
}

int d_coef_backward_impl(const float* g, const float* xall, const float* Wc, int64_t ldw, int64_t H, int64_t nf, int64_t nl,
                         int64_t ML, int64_t ds, int64_t dl, int64_t L, int32_t basis, float* shall, float* gall, float* wab,
                         float* partial, float* gW, int64_t ldgw, float* g_bc, int64_t ld_gbc, int32_t zero_pad,
                         float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b, bgStream s) {
    if (!g || !xall || !Wc || !gall || !wab || !partial || !gW || !g_bc || !g_fe_w || !g_fe_b || !g_le_w || !g_le_b)
        return BG_ERR_ARG;
    if (H <= 0 || L < 1 || ldw < 4 * H + 4 || ldgw < 4 * H + 4) return BG_ERR_ARG;
    if ((basis != BG_CHAIN_POWER && basis != BG_CHAIN_CHEB) || (basis == BG_CHAIN_CHEB && !shall)) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, du = ds + dl, N = (L + 1) * du, S = N + du;
    // G^m at rows [0, N) of slab m, du zero rows below it (the shift-up view reads them)
    // one launch: G^m at rows [0, N) of slab m with du zero rows below it (the shift-up view reads them) -- slabs 0 .. L-1
    // and the padding of slab L cleared, G^L = g copied in; Wab = [Wc[:, :2H]; Wc[:, 2H:4H]]  ([4H, 2H]): G Wa + shiftup(G) Wb
    {
        const int64_t f = (int64_t)sizeof(float);
        void* const dst[5] = {gall, gall + L * S * H2 + N * H2, gall + L * S * H2, wab, wab + H2 * H2};
        const void* const src[5] = {nullptr, nullptr, g, Wc, Wc + H2};
        const int64_t dp[5] = {f * L * S * H2, f * (S - N) * H2, f * N * H2, f * H2, f * H2};
        const int64_t sp[5] = {0, 0, f * N * H2, f * ldw, f * ldw};
        const int64_t wd[5] = {f * L * S * H2, f * (S - N) * H2, f * N * H2, f * H2, f * H2};
        const int64_t rw[5] = {1, 1, 1, H2, H2};
        BG_TRY(bg_fill_copy_batch(5, dst, dp, src, sp, wd, rw, s));
    }
    for (int64_t m = L - 1; m >= 0; --m) {
        const float* nxt = gall + (m + 1) * S * H2;
        if (basis == BG_CHAIN_CHEB) {                                   // second operand = the adjoint shift of G^{m+1}
            BG_TRY(bg_gemm_cheb_f32(0, N, H2, nxt, H2, H2, du, L + 1, 1, wab, H2, gall + m * S * H2, H2, nxt, H2, nullptr, 0, s));
            continue;
        }
        const float* A[4] = {nxt, nxt + du * H2, nullptr, nullptr};     // monomials: one block up = a row offset
        const int64_t ld[4] = {H2, H2, 0, 0}, Ks[4] = {H2, H2, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, N, H2, 2, A, ld, Ks, wab, H2, gall + m * S * H2, H2, nullptr, nullptr, BG_ACT_NONE, 0.01f,
                           nxt, H2, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
    }
    // dWc[:, :4H] = sum_m G^{m+1 T} [X^m | shift(X^m)]: one long reduction over the stacked slabs.  Chebyshev: the
    // shifts of ALL layers' states by one launch (slabs of L+2 blocks: one of padding, then the L+1 coefficients)
    if (basis == BG_CHAIN_CHEB)
        BG_TRY(bg_cheb_shift_slabs_f32(xall, shall, L * (L + 2) + 1, L + 2, 1, L + 1, du * H2, 0, s));
    const float* B[4] = {xall + du * H2, basis == BG_CHAIN_CHEB ? shall + du * H2 : xall, nullptr, nullptr};
    const int64_t ldb[4] = {H2, H2, 0, 0}, Ns[4] = {H2, H2, 0, 0};
    BG_TRY(bg_gemm_tn_splitk_f32(H2, L * S, gall + S * H2, H2, 2, B, ldb, Ns, gW, ldgw, partial,
                                 (int32_t)coef_splits(H2, L * S), 0, s));
    return bg_d_coef_init_bwd_strided_f32(gall, S * H2, L, H, nf, nl, ML, ds, dl, g_fe_w, g_fe_b, g_le_w, g_le_b,
                                          gW + 2 * H2, ldgw, g_bc, ld_gbc, zero_pad, s);
}
}  // namespace

extern "C" int bg_d_coef_forward_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b,
                                     const float* pe, const float* Wc, int64_t ldw, const float* bc, int64_t H,
                                     int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L, int32_t basis,
                                     float* xall, float* efull, bgStream s) {
    return d_coef_forward_impl(fe_w, fe_b, le_w, le_b, pe, Wc, ldw, bc, 1, H, nf, nl, ML, ds, dl, L, basis, xall, efull, s);
}

extern "C" int64_t bg_d_coef_backward_workspace(int64_t H, int64_t ds, int64_t dl, int64_t L) {
    if (H <= 0 || ds < 0 || dl < 0 || L < 0) return -1;
    const int64_t H2 = 2 * H, du = ds + dl, S = (L + 2) * du;
    return coef_splits(H2, L * S) * H2 * 2 * H2;
}

extern "C" int bg_d_coef_backward_f32(const float* g, const float* xall, const float* Wc, int64_t ldw,
                                      int64_t H, int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L,
                                      int32_t basis, float* shall, float* gall, float* wab, float* partial, float* gW,
                                      float* g_bc, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b, bgStream s) {
    return d_coef_backward_impl(g, xall, Wc, ldw, H, nf, nl, ML, ds, dl, L, basis, shall, gall, wab, partial, gW,
                                4 * H + 4, g_bc, 1, 0, g_fe_w, g_fe_b, g_le_w, g_le_b, s);
}

// The same with the block's composition (models.VoxelBlock.compose) inside: from messageMLP / updateMLP to the table.
// P = 4H + 8; wm_ext, r, wcb are [2H, P] buffers of the caller (wcb = [Wc | bc | pad] stays for the backward, wm_ext too).
extern "C" int bg_d_table_forward_f32(const float* Wm, int64_t ldwm, const float* bm, const float* Wu, int64_t ldwu,
                                      const float* bu, const float* fe_w, const float* fe_b, const float* le_w,
                                      const float* le_b, const float* pe, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                                      int64_t ds, int64_t dl, int64_t L, int32_t basis, float* wm_ext, float* r, float* wcb,
                                      float* xall, float* efull, bgStream s) {
    if (!Wm || !bm || !Wu || !bu || !wm_ext || !r || !wcb || H <= 0) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, P = 2 * H2 + 8;
    BG_TRY(bg_d_compose_prep_f32(Wm, ldwm, bm, Wu, ldwu, bu, H2, wm_ext, r, s));
    {
        const float* A[4] = {Wu + H2, nullptr, nullptr, nullptr};       // Wu2
        const int64_t ld[4] = {ldwu, 0, 0, 0}, Ks[4] = {H2, 0, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, H2, P, 1, A, ld, Ks, wm_ext, P, wcb, P, nullptr, nullptr, BG_ACT_NONE, 0.01f, r, P, nullptr, 0,
                           nullptr, nullptr, 0, nullptr, 0, s));
    }
    return d_coef_forward_impl(fe_w, fe_b, le_w, le_b, pe, wcb, P, wcb + 2 * H2 + 4, P, H, nf, nl, ML, ds, dl, L, basis, xall,
                               efull, s);
}

// gp, gwm_ext: [2H, P] workspaces.  Writes g_Wm [2H, 4H+3], g_bm, g_Wu [2H, 4H], g_bu and the encoders' gradients.
extern "C" int bg_d_table_backward_f32(const float* g, const float* xall, const float* wcb, const float* wm_ext,
                                       const float* Wu, int64_t ldwu, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                                       int64_t ds, int64_t dl, int64_t L, int32_t basis, float* shall, float* gall,
                                       float* wab, float* partial, float* gp, float* gwm_ext, float* g_Wm, float* g_bm,
                                       float* g_Wu, float* g_bu, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b,
                                       bgStream s) {
    if (!wcb || !wm_ext || !Wu || !gp || !gwm_ext || !g_Wm || !g_bm || !g_Wu || !g_bu || H <= 0 || ldwu < 4 * H) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, P = 2 * H2 + 8;
    BG_TRY(d_coef_backward_impl(g, xall, wcb, P, H, nf, nl, ML, ds, dl, L, basis, shall, gall, wab, partial, gp, P,
                                gp + 2 * H2 + 4, P, 1, g_fe_w, g_fe_b, g_le_w, g_le_b, s));
    {   // d Wu2 = gp wm_ext^T, straight into the second half of g_Wu's rows
        const float* A[4] = {gp, nullptr, nullptr, nullptr};
        const int64_t ld[4] = {P, 0, 0, 0}, Ks[4] = {P, 0, 0, 0};
        BG_TRY(bg_gemm_f32(0, 1, H2, H2, 1, A, ld, Ks, wm_ext, P, g_Wu + H2, 2 * H2, nullptr, nullptr, BG_ACT_NONE, 0.01f,
                           nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
    }
    {   // d wm_ext = Wu2^T gp
        const float* A[4] = {Wu + H2, nullptr, nullptr, nullptr};
        const int64_t ld[4] = {ldwu, 0, 0, 0}, Ks[4] = {H2, 0, 0, 0};
        BG_TRY(bg_gemm_f32(1, 0, H2, P, 1, A, ld, Ks, gp, P, gwm_ext, P, nullptr, nullptr, BG_ACT_NONE, 0.01f, nullptr, 0,
                           nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
    }
    return bg_d_compose_bwd_finish_f32(gp, gwm_ext, H2, g_Wm, g_bm, g_Wu, g_bu, s);
}

// ---------------------------------------------------------------- the critic update below the coefficient table
namespace {

struct CriticBuffers {
    // forward (kept for the backward)
    float *Zp[3], *Sb, *v[3], *dS, *Xs, *Xb, *hs, *hb, *vals, *valb, *hat_s, *hat_b, *gp_b, *gp_s, *G2, *X, *Ys, *acc, *part;
    int32_t *as, *ab, *arg_s, *arg_sb, *arg_b, *idx3;
    // backward
    float *dvb, *dvs, *dacc, *dYs, *dX, *dG2, *dgp_b, *dgp_s, *c_b, *c_s, *dhs, *dhb, *dXs, *dXb, *T, *splitk, *colsum;
};

// workgroups a split-K product aims at: ~2 per CU fill the chip; a long reduction dimension (the products over the
// stacked calls of the generator update: R = 70 k .. 130 k rows) runs faster with shorter K ranges per workgroup
// (tools/gemm_tn_bench.py: R = 126,720: 152 us at 512, 129 at 1,024, 111 at the 256-split cap; R = 11,520 is best at 512)
int64_t tn_target(int64_t R) { return R >= 65536 ? 4096 : (R >= 32768 ? 1024 : 512); }

int32_t tn_splits(int64_t M, int64_t N, int64_t R) {           // ops.gemm_tn_raw's rule (one B segment)
    const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
    const int64_t steps = (R + 31) / 32;
    int64_t s = (tn_target(R) + tiles - 1) / tiles;
    if (s > steps) s = steps;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int32_t)s;
}
int32_t colsum_blocks(int64_t M) {                              // ops.colsum_raw's rule
    int64_t nb = (M + 127) / 128;
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return (int32_t)nb;
}

CriticBuffers carve_critic(Arena& ar, const bgCriticHead& h) {
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, ds = h.ds, dl = h.dl, H2 = h.H2, Hd = h.Hd;
    CriticBuffers b;
    // [4, N, K1 dl] in one piece: the three label sets' powers (bg_regroup_f32 writes them side by side; one product
    // reads them as 3 N rows); in the backward d X [2 N, K1 dl] takes slots 2..3 (the sample's powers are done by then),
    // so that the label rows' gradient is ONE product over the 4 N rows [Zp0; Zp1; dX]^T [dv0; dv1; G2]
    b.Zp[0] = ar.take(4 * N * K1 * dl);
    b.Zp[1] = b.Zp[0] ? b.Zp[0] + N * K1 * dl : nullptr; b.Zp[2] = b.Zp[0] ? b.Zp[0] + 2 * N * K1 * dl : nullptr;
    b.Sb = ar.take(N * H2);
    b.v[0] = ar.take(4 * N * H2);                  // [v0; v1; v2 | G2 (2 N rows, written after v2 has been pooled)]
    b.v[1] = b.v[0] ? b.v[0] + N * H2 : nullptr; b.v[2] = b.v[0] ? b.v[0] + 2 * N * H2 : nullptr;
    b.dS = ar.take(N * H2);
    b.idx3 = reinterpret_cast<int32_t*>(ar.take(3 * N));
    // pooled rows and arg-max ids of the three label sets in one piece each (real, fake, sample: one pooling launch per
    // level serves all three; the decoders read the first two sets' rows)
    b.Xs = ar.take(3 * S * H2); b.Xb = ar.take(3 * B * H2);
    b.as = reinterpret_cast<int32_t*>(ar.take(3 * S * H2)); b.ab = reinterpret_cast<int32_t*>(ar.take(3 * B * H2));
    b.hs = ar.take(2 * S * Hd); b.hb = ar.take(2 * B * Hd); b.vals = ar.take(2 * S); b.valb = ar.take(2 * B);
    b.hat_s = b.Xs ? b.Xs + 2 * S * H2 : nullptr; b.hat_b = b.Xb ? b.Xb + 2 * B * H2 : nullptr;
    b.arg_s = b.as ? b.as + 2 * S * H2 : nullptr; b.arg_sb = b.ab ? b.ab + 2 * B * H2 : nullptr;
    b.arg_b = reinterpret_cast<int32_t*>(ar.take(B * H2));
    b.gp_b = ar.take(H2); b.gp_s = ar.take(H2);
    b.G2 = b.v[2]; b.X = ar.take(2 * N * K1 * dl); b.acc = ar.take(N * 2 * dl);
    b.part = ar.take(128);
    b.dvb = ar.take(2 * B); b.dvs = ar.take(2 * S); b.dacc = ar.take(N * 2 * dl);
    b.dX = b.Zp[2]; b.dG2 = ar.take(2 * N * H2);
    b.dgp_b = ar.take(B * H2); b.dgp_s = ar.take(S * H2); b.c_b = ar.take(H2); b.c_s = ar.take(H2);
    b.dhs = ar.take(2 * S * Hd); b.dhb = ar.take(2 * B * Hd); b.dXs = ar.take(2 * S * H2); b.dXb = ar.take(2 * B * H2);
    b.T = ar.take(2 * S * H2);
    // largest split-K partial of the backward's weight-gradient products
    int64_t sk = 0;
    auto need = [&](int64_t M, int64_t Nn, int64_t R) { const int64_t x = (int64_t)tn_splits(M, Nn, R) * M * Nn; if (x > sk) sk = x; };
    need(K1 * dl, H2, 4 * N); need(K1 * ds, H2, N);
    need(1, Hd, 2 * S); need(Hd, H2, 2 * S); need(1, Hd, 2 * B); need(Hd, H2, 2 * B);
    b.splitk = ar.take(sk);
    b.colsum = ar.take(256 * (H2 > Hd ? H2 : Hd));
    return b;
}

int critic_shape_ok(const bgCriticHead& h) {
    if (h.N <= 0 || h.S <= 0 || h.B <= 0 || h.L < 1 || h.ds <= 0 || h.dl <= 0 || h.H2 <= 0 || h.Hd <= 0) return BG_ERR_ARG;
    if (h.ds % 4 || h.dl % 4 || h.H2 % 4 || h.Hd % 4 || h.H2 > 1024 || h.L > 64) return BG_ERR_SHAPE;
    if (h.blk_n <= 0 || h.blk_max_deg > 8 || h.blk_max_rows <= 0) return BG_ERR_SHAPE;
    if (h.basis != BG_CHAIN_POWER && h.basis != BG_CHAIN_CHEB) return BG_ERR_ARG;
    return BG_OK;
}

// y = x W^T + b for a decoder layer (nn.Linear layout)
int dec_linear(int64_t M, int64_t Nout, int64_t K, const float* x, const float* W, const float* bias, float* y, bgStream s) {
    const float* A[4] = {x, nullptr, nullptr, nullptr};
    int64_t ld[4] = {K, 0, 0, 0}, Ks[4] = {K, 0, 0, 0};
    return bg_gemm_f32(0, 1, M, Nout, 1, A, ld, Ks, W, K, y, Nout, bias, nullptr, BG_ACT_NONE, 0.f, nullptr, 0, nullptr, 0,
                       nullptr, nullptr, 0, nullptr, 0, s);
}
// C [M, N] = A [M, K] (pitch lda) * B, B given as [K, N] (tb = 0) or [N, K] (tb = 1), pitch ldb; optional start matrix
int mm(int64_t M, int64_t Nn, int64_t K, const float* A_, int64_t lda, const float* B_, int64_t ldb, int tb, float* C_,
       const float* pre_add, const float* acc_init, bgStream s) {
    const float* A[4] = {A_, nullptr, nullptr, nullptr};
    int64_t ld[4] = {lda, 0, 0, 0}, Ks[4] = {K, 0, 0, 0};
    return bg_gemm_f32(0, tb, M, Nn, 1, A, ld, Ks, B_, ldb, C_, Nn, nullptr, nullptr, BG_ACT_NONE, 0.f, nullptr, 0,
                       pre_add, pre_add ? Nn : 0, nullptr, acc_init, acc_init ? Nn : 0, nullptr, 0, s);
}
// C [M, N] (+)= A^T B for A [R, M], B [R, N], both contiguous
int tn(int64_t M, int64_t Nn, int64_t R, const float* A_, const float* B_, float* C_, float* partial, int accumulate, bgStream s) {
    const float* Bs[4] = {B_, nullptr, nullptr, nullptr};
    int64_t ldb[4] = {Nn, 0, 0, 0}, Ns[4] = {Nn, 0, 0, 0};
    return bg_gemm_tn_splitk_f32(M, R, A_, M, 1, Bs, ldb, Ns, C_, Nn, partial, tn_splits(M, Nn, R), accumulate, s);
}

}  // namespace

extern "C" int64_t bg_critic_head_workspace(const bgCriticHead* h) {
    if (!h || critic_shape_ok(*h) != BG_OK) return -1;
    Arena ar{nullptr, 0, 0};
    carve_critic(ar, *h);
    return ar.used + 64;
}

extern "C" int bg_critic_head_forward_f32(const bgCriticHead* hp, float coef, float* workspace, int64_t workspace_floats,
                                          float* loss, bgStream s) {
    if (!hp || !workspace || !loss) return BG_ERR_ARG;
    const bgCriticHead& h = *hp;
    int rc = critic_shape_ok(h);
    if (rc != BG_OK) return rc;
    if (!h.Zs || !h.Zk || !h.W || !h.b_w1 || !h.b_b1 || !h.b_w2 || !h.b_b2 || !h.s_w1 || !h.s_b1 || !h.s_w2 || !h.s_b2 ||
        !h.wb || !h.ws || !h.st_rowptr || !h.st_col || !h.story_of_voxel || !h.sg_rowptr || !h.sg_col || !h.graph_of_story ||
        !h.graph_of_voxel || !h.a_rowptr || !h.a_col || !h.a_count || !h.at_rowptr || !h.at_col || !h.blk_ptr) return BG_ERR_ARG;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    CriticBuffers b = carve_critic(ar, h);
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, ds = h.ds, dl = h.dl, H2 = h.H2, Hd = h.Hd;
    const float* Ws = h.W;
    const float* Wl = h.W + K1 * ds * H2;

    // label powers [N, K1, 3, dl] -> one contiguous [N, K1 dl] matrix per label set
    // (and, by the same launch, the row of the label-free part that row r of the 3 N stacked rows adds: r mod N)
    BG_TRY(bg_regroup_iota_f32(h.Zk, b.Zp[0], N, K1, 3, dl, 1, b.idx3, 3 * N, N, s));
    // label-free part once, then one thin product per label set on top of it (models.py collapsed_multi)
    BG_TRY(mm(N, H2, K1 * ds, h.Zs, K1 * ds, Ws, H2, 0, b.Sb, nullptr, nullptr, s));
    {   // all three sets in one product over 3 N rows; row r adds S[r mod N]
        const float* A[4] = {b.Zp[0], nullptr, nullptr, nullptr};
        const int64_t ld[4] = {K1 * dl, 0, 0, 0}, Ks[4] = {K1 * dl, 0, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, 3 * N, H2, 1, A, ld, Ks, Wl, H2, b.v[0], H2, nullptr, nullptr, BG_ACT_NONE, 0.f, nullptr, 0,
                           b.Sb, H2, b.idx3, nullptr, 0, nullptr, 0, s));
    }
    // D(real), D(fake) and the sample point: max over each story, then over the stories of each graph -- the three label
    // sets by ONE launch per level (set 2's rows are hat_s / hat_b, its ids arg_s / arg_sb: which voxel carries each pooled
    // column, losses.py:83-84 in closed form); real and fake share one run of a decoder
    BG_TRY(bg_segmax_batch_f32(b.v[0], H2, N * H2, h.st_rowptr, h.st_col, b.Xs, H2, S * H2, b.as, S * H2, S, H2, 3, s));
    BG_TRY(bg_segmax_batch_f32(b.Xs, H2, S * H2, h.sg_rowptr, h.sg_col, b.Xb, H2, B * H2, b.ab, B * H2, B, H2, 3, s));
    // both decoders: one launch each (no activation between their layers: val = X (w2 W1)^T + w2 . b1 + b2; the row
    // w2 W1 is kept -- d D / d pooled of such a decoder is that constant row)
    BG_TRY(bg_decoder_fwd_pair_f32(H2, Hd, 2 * S, b.Xs, H2, h.s_w1, H2, h.s_b1, h.s_w2, h.s_b2, b.vals, b.gp_s,
                                   2 * B, b.Xb, H2, h.b_w1, H2, h.b_b1, h.b_w2, h.b_b2, b.valb, b.gp_b, s));
    BG_TRY(bg_gather_arg_f32(reinterpret_cast<const float*>(b.arg_s), H2, b.arg_sb, reinterpret_cast<float*>(b.arg_b), H2, B, H2, s));
    // scattered to the arg-max voxels (row pitch 0: every segment reads the same row): building rows and story rows,
    // one launch
    BG_TRY(bg_scatter_arg_pair_f32(b.gp_b, 0, b.arg_b, h.graph_of_voxel, b.G2, B,
                                   b.gp_s, 0, b.arg_s, h.story_of_voxel, b.G2 + N * H2, S, H2, N, H2, s));
    // sum_k (A^T)^k G Wl_k^T: all powers' columns by one product, both outputs side by side through one chain
    BG_TRY(mm(2 * N, K1 * dl, H2, b.G2, H2, Wl, H2, 1, b.X, nullptr, nullptr, s));
    // (the chain's 2 dl columns are two groups of dl: the building half and the story half of X's rows, read where the
    //  product left them -- no regrouping launch in between)
    const int64_t W2c = 2 * dl, ldx = K1 * dl, half = N * ldx;
    BG_TRY(bg_spmm_chain_grouped_f32(b.X + h.L * dl, ldx, h.at_rowptr, h.at_col, nullptr, BG_SCALE_NONE, h.a_count, BG_SCALE_DIV,
                                     h.blk_ptr, h.blk_n, h.blk_max_rows, h.blk_max_nnz, h.blk_max_deg,
                                     b.X + (h.L - 1) * dl, ldx, -dl, b.acc, W2c, 0, h.L, W2c,
                                     h.basis ? BG_CHAIN_CLENSHAW : BG_CHAIN_POWER, dl, half, half, 0, s));
    BG_TRY(bg_critic_loss_f32(b.valb, h.wb, 2 * B, b.vals, h.ws, 2 * S, b.acc, dl, 2 * N, dl, coef, b.part, loss, s));
    return BG_OK;
}

extern "C" int bg_critic_head_backward_f32(const bgCriticHead* hp, float coef, const float* go, float* workspace,
                                           int64_t workspace_floats, float* gW, float* g_b_w1, float* g_b_b1,
                                           float* g_b_w2, float* g_b_b2, float* g_s_w1, float* g_s_b1, float* g_s_w2,
                                           float* g_s_b2, bgStream s) {
    if (!hp || !workspace || !go || !gW || !g_b_w1 || !g_b_b1 || !g_b_w2 || !g_b_b2 || !g_s_w1 || !g_s_b1 || !g_s_w2 ||
        !g_s_b2) return BG_ERR_ARG;
    const bgCriticHead& h = *hp;
    int rc = critic_shape_ok(h);
    if (rc != BG_OK) return rc;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    CriticBuffers b = carve_critic(ar, h);
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, ds = h.ds, dl = h.dl, H2 = h.H2, Hd = h.Hd;
    const float* Wl = h.W + K1 * ds * H2;
    float* gWs = gW;
    float* gWl = gW + K1 * ds * H2;
    const int64_t W2c = 2 * dl, ldx = K1 * dl, half = N * ldx;

    BG_TRY(bg_critic_loss_bwd_f32(go, h.wb, 2 * B, h.ws, 2 * S, b.acc, dl, 2 * N, dl, coef, b.dvb, b.dvs, b.dacc, s));
    // ---- penalty: transpose of the Horner chain = the powers of A applied to d acc
    // d X [2 N, K1 dl] (building rows, then story rows) is filled in place: block 0 of either half = its dl columns
    // of d acc (two strided copies, one launch), blocks 1 .. L by the chain, whose 2 dl columns are written as two
    // groups of dl -- one per half -- instead of a [N, K1, 2 dl] matrix that a regrouping launch would take apart
    {
        const int64_t f = (int64_t)sizeof(float);
        void* const dst[2] = {b.dX, b.dX + half};
        const void* const src[2] = {b.dacc, b.dacc + dl};
        const int64_t dp[2] = {f * ldx, f * ldx}, sp[2] = {f * W2c, f * W2c}, wd[2] = {f * dl, f * dl}, rw[2] = {N, N};
        BG_TRY(bg_fill_copy_batch(2, dst, dp, src, sp, wd, rw, s));
    }
    BG_TRY(bg_spmm_chain_grouped_f32(b.dacc, W2c, h.a_rowptr, h.a_col, h.a_count, BG_SCALE_DIV, nullptr, BG_SCALE_NONE,
                                     h.blk_ptr, h.blk_n, h.blk_max_rows, h.blk_max_nnz, h.blk_max_deg,
                                     nullptr, 0, 0, b.dX + dl, ldx, dl, h.L, W2c, h.basis ? BG_CHAIN_CHEB : BG_CHAIN_POWER,
                                     dl, 0, 0, half, s));
    BG_TRY(mm(2 * N, H2, K1 * dl, b.dX, K1 * dl, Wl, H2, 0, b.dG2, nullptr, nullptr, s));                   // d G2 = dX Wl
    BG_TRY(bg_gather_arg_pair_f32(b.dG2, H2, b.arg_b, b.dgp_b, B, b.dG2 + N * H2, H2, b.arg_s, b.dgp_s, S, H2, H2, s));
    // ---- both decoders' backward, one launch each (bg_decoder_bwd_f32: every gradient is rank one).  Penalty share:
    // every row of d D / d pooled is the same w2 W1, so its gradient is the column sum c of the rows gathered above
    // (g W1 = w2 (x) c, g W2 = W1 c).  D(real), D(fake): building decoder, back through the story level, story decoder.
    BG_TRY(bg_decoder_bwd_f32(2 * B, B, H2, Hd, b.dvb, b.Xb, H2, h.b_b1, b.dgp_b, H2, h.b_w1, H2, h.b_w2, b.gp_b, nullptr, 0,
                              g_b_w1, H2, g_b_b1, g_b_w2, g_b_b2, b.dXb, H2, s));
    BG_TRY(bg_scatter_arg_batch_f32(b.dXb, H2, B * H2, b.ab, B * H2, h.graph_of_story, b.T, H2, S * H2, B, S, H2, 2, nullptr, 0, s));
    BG_TRY(bg_decoder_bwd_f32(2 * S, S, H2, Hd, b.dvs, b.Xs, H2, h.s_b1, b.dgp_s, H2, h.s_w1, H2, h.s_w2, b.gp_s, b.T, H2,
                              g_s_w1, H2, g_s_b1, g_s_w2, g_s_b2, b.dXs, H2, s));          // + what arrived through the buildings
    // to the arg-max voxels (v[0], v[1] are free after the forward's pooling: they take d v), then the coefficient table
    // (the same launch leaves d v0 + d v1 in dS: what reaches the label-free rows, which both sets share)
    BG_TRY(bg_scatter_arg_batch_f32(b.dXs, H2, S * H2, b.as, S * H2, h.story_of_voxel, b.v[0], H2, N * H2, S, N, H2, 2, b.dS, H2, s));
    // label rows: Zp0^T dv0 + Zp1^T dv1 + dX^T G2 as one product over the 4 N stacked rows
    BG_TRY(tn(K1 * dl, H2, 4 * N, b.Zp[0], b.v[0], gWl, b.splitk, 0, s));
    BG_TRY(tn(K1 * ds, H2, N, h.Zs, b.dS, gWs, b.splitk, 0, s));
    return BG_OK;
}

// ---------------------------------------------------------------- generator voxel GNN: forward that saves, backward
namespace {

struct GStack {
    int64_t n_ops;                   // attention / block calls in forward order
    int kind[64];                    // 0 = attention, 1 = block
    int idx[64];                     // call number among its kind
    int64_t n_att, n_blk;
    float *nnp, *posb, *v[65], *gm[32], *y[32], *a[32], *soft0[32], *soft[32];
    float *pq, *term, *h0, *ha;                                                   // forward scratch
    // backward scratch
    // Per-call pieces whose weight-gradient products run ONCE over all calls are carved as contiguous stacks (call i at
    // base + i * rows * width): the block inputs vb, aggregates gm, and in the backward gz / gterm / story / sb; the
    // attention inputs va, G_al, dth_p.  pdm4_rep = the position columns repeated per block call (third operand there).
    float *vb, *va, *gm_all, *gz_all, *gterm_all, *story_all, *sb_all, *G_al_all, *dth_all, *pdm4_rep;
    float *g0, *g1, *gx, *gsb, *g_story, *gg, *d_att, *dpre, *gb_acc, *gx_acc, *tmpNp, *zero, *splitk, *colsum;
};

int32_t tn_splits_multi(int64_t M, const int64_t* Ns, int n_seg, int64_t R) {     // ops.gemm_tn_raw's rule
    int64_t ct = 0;
    for (int i = 0; i < n_seg; ++i) ct += (Ns[i] + 63) / 64;
    const int64_t tiles = ((M + 63) / 64) * ct, steps = (R + 31) / 32;
    int64_t s = (tn_target(R) + tiles - 1) / tiles;
    if (s > steps) s = steps;
    if (s > 256) s = 256;
    return (int32_t)(s < 1 ? 1 : s);
}

int gstack_ops(int64_t L, GStack& g) {
    int64_t last_att = -1;
    for (int64_t i = 0; i < L; ++i) if (i % 2 == 0 && i != 0) last_att = i;
    g.n_ops = 0; g.n_att = 0; g.n_blk = 0;
    g.kind[g.n_ops] = 0; g.idx[g.n_ops++] = (int)g.n_att++;
    for (int64_t i = 0; i <= last_att; ++i) {
        if (g.n_ops + 2 >= 64 || g.n_blk >= 32 || g.n_att >= 32) return BG_ERR_SHAPE;
        g.kind[g.n_ops] = 1; g.idx[g.n_ops++] = (int)g.n_blk++;
        if (i % 2 == 0 && i != 0) { g.kind[g.n_ops] = 0; g.idx[g.n_ops++] = (int)g.n_att++; }
    }
    return BG_OK;
}

int cross_logit_blocks(int64_t Nv) {                       // ops.AttentionNode._backward_tail
    int64_t nb = (Nv + 3) / 4;
    if (nb > 1024) nb = 1024;
    return (int)(nb < 1 ? 1 : nb);
}

int carve_gstack(Arena& ar, int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H, int64_t L, GStack& g) {
    int rc = gstack_ops(L, g);
    if (rc != BG_OK) return rc;
    g.nnp = ar.take(Np * H); g.posb = ar.take(Nv * H);
    const int64_t nb = cross_logit_blocks(Nv);
    // a call's input is the previous call's output: block inputs and attention inputs in one contiguous stack each
    g.vb = ar.take(g.n_blk * Nv * H); g.va = ar.take(g.n_att * Nv * H);
    for (int64_t k = 0; k < g.n_ops; ++k) g.v[k] = (g.kind[k] == 1 ? g.vb : g.va) + (int64_t)g.idx[k] * Nv * H;
    g.v[g.n_ops] = ar.take(Nv * H);
    g.gm_all = ar.take(g.n_blk * Nv * H);
    for (int64_t i = 0; i < g.n_blk; ++i) { g.gm[i] = g.gm_all + i * Nv * H; g.y[i] = ar.take(Nv * H); }
    for (int64_t c = 0; c < g.n_att; ++c) { g.a[c] = ar.take(Nv * H); g.soft0[c] = ar.take(Nv); g.soft[c] = ar.take(Ec); }
    g.pq = ar.take(S * 2 * H); g.term = ar.take(S * H); g.h0 = ar.take(Nv); g.ha = ar.take(Ec);
    g.g0 = ar.take(Nv * H); g.g1 = ar.take(Nv * H);
    g.gz_all = ar.take(g.n_blk * Nv * H); g.gterm_all = ar.take(g.n_blk * S * H); g.story_all = ar.take(g.n_blk * S * H);
    g.sb_all = ar.take(g.n_blk * S * H); g.pdm4_rep = ar.take(g.n_blk * Nv * 4);
    g.gx = ar.take(Nv * 2 * H); g.gsb = ar.take(S * 2 * H); g.g_story = ar.take(S * H);
    g.G_al_all = ar.take(g.n_att * Nv * (H + 4)); g.gg = ar.take(Nv * H); g.d_att = ar.take(Ec); g.dpre = ar.take(Ec * H);
    g.dth_all = ar.take(g.n_att * nb * H);
    g.gb_acc = ar.take(Np * H); g.gx_acc = ar.take(Np * H); g.tmpNp = ar.take(Np * H); g.zero = ar.take(Nv * H);
    int64_t sk = 0;
    auto need = [&](int64_t M, std::initializer_list<int64_t> ns, int64_t R) {
        int64_t Ns[4] = {0, 0, 0, 0}, N = 0; int n = 0;
        for (int64_t v : ns) { Ns[n++] = v; N += v; }
        const int64_t x = (int64_t)tn_splits_multi(M, Ns, n, R) * M * N;
        if (x > sk) sk = x;
    };
    need(H, {H, H, 4}, g.n_blk * Nv); need(H, {H, H}, g.n_blk * S); need(H + 4, {H}, g.n_att * Nv); need(H, {4}, Nv);
    need(H, {H}, Np);
    g.splitk = ar.take(sk);
    g.colsum = ar.take(256 * (H + 4));
    return BG_OK;
}

int gstack_shape_ok(const bgGeneratorWeights* w, const bgGeneratorPlan* p) {
    const int64_t H = w->H, L = w->voxel_layer;
    if (H <= 0 || p->Np <= 0 || p->Nv <= 0 || p->S <= 0 || p->Ec <= 0 || L < 1) return BG_ERR_ARG;
    if (H % 64 != 0 || H > 256 || p->Nv > 65536) return BG_ERR_SHAPE;
    if (!p->vox_all_has || p->cx_max_len < 1 || p->cx_max_len > 64) return BG_ERR_SHAPE;
    return BG_OK;
}

// C [M, sum Ns] (+)= A^T [B_0 | B_1 | ..] for A [R, M] (contiguous), B_s [R, Ns[s]] with pitch ldb[s]
int tn_multi(int64_t M, int64_t R, const float* A_, int n_seg, const float* const* Bs, const int64_t* ldb, const int64_t* Ns,
             float* C_, float* partial, int accumulate, bgStream s) {
    int64_t N = 0;
    for (int i = 0; i < n_seg; ++i) N += Ns[i];
    return bg_gemm_tn_splitk_f32(M, R, A_, M, n_seg, Bs, ldb, Ns, C_, N, partial, tn_splits_multi(M, Ns, n_seg, R), accumulate, s);
}

}  // namespace

extern "C" int64_t bg_gstack_workspace(int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H, int64_t L) {
    if (Np <= 0 || Nv <= 0 || S <= 0 || Ec <= 0 || H <= 0 || L < 1) return -1;
    Arena ar{nullptr, 0, 0};
    GStack g;
    if (carve_gstack(ar, Np, Nv, S, Ec, H, L, g) != BG_OK) return -1;
    return ar.used + 64;
}

extern "C" int bg_gstack_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const float* x,
                                     const float* voxel_in, const float* pdm4, const float* const* exp_mask,
                                     const float* const* exp_att, int64_t n_calls, float* workspace,
                                     int64_t workspace_floats, float* mask_hard, float* mask_soft, float* att_hard,
                                     float* att_soft, bgStream s) {
    if (!w || !p || !x || !voxel_in || !pdm4 || !exp_mask || !exp_att || !workspace || !mask_hard || !mask_soft || !att_hard ||
        !att_soft) return BG_ERR_ARG;
    int rc = gstack_shape_ok(w, p);
    if (rc != BG_OK) return rc;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    const int64_t H = w->H, Np = p->Np, Nv = p->Nv, S = p->S, Ec = p->Ec, L = w->voxel_layer;
    Arena ar{workspace, 0, workspace_floats};
    GStack g;
    rc = carve_gstack(ar, Np, Nv, S, Ec, H, L, g);
    if (rc != BG_OK) return rc;
    if (ar.used > workspace_floats || n_calls != g.n_att) return BG_ERR_ARG;

    BG_TRY(linear(Nv, H, 1, voxel_in, 4, nullptr, 0, nullptr, 0, 4, w->v_enc_w, 4, g.v[0], w->v_enc_b, nullptr, BG_ACT_NONE,
                  nullptr, w->pe, H, p->level, nullptr, s));
    BG_TRY(linear(Np, H, 1, x, H, nullptr, 0, nullptr, 0, H, w->nnp_w, H, g.nnp, w->nnp_b, nullptr, BG_ACT_NONE, nullptr,
                  nullptr, 0, nullptr, nullptr, s));
    BG_TRY(linear(Nv, H, 1, pdm4, 4, nullptr, 0, nullptr, 0, 4, w->Wc + 2 * H, w->ldwc, g.posb, w->bc, nullptr, BG_ACT_NONE,
                  nullptr, nullptr, 0, nullptr, nullptr, s));
    for (int64_t k = 0; k < g.n_ops; ++k) {
        const float* v = g.v[k];
        float* vn = g.v[k + 1];
        if (g.kind[k] == 0) {
            const int c = g.idx[k];
            const bool last = c == g.n_att - 1;
            BG_TRY(linear(Nv, H, 1, v, H, nullptr, 0, nullptr, 0, H, w->Wcat, H, g.a[c], w->bcat, nullptr, BG_ACT_NONE, nullptr,
                          nullptr, 0, nullptr, nullptr, s));
            BG_TRY(bg_attention_fwd_f32(g.a[c], H, nullptr, 0, g.nnp, H, x, H, v, H, w->theta, p->cx_rowptr, p->cx_col,
                                        exp_mask[c], exp_att[c], -1.0f, last ? mask_soft : g.soft0[c], last ? mask_hard : g.h0,
                                        last ? att_soft : g.soft[c], last ? att_hard : g.ha, vn, H, Nv, H,
                                        w->Wcat + H * H, H, w->bcat + H, s));
        } else {
            const int i = g.idx[k];
            BG_TRY(bg_gather_story_f32(v, H, p->vx_rowptr, p->vx_col, p->vx_count, BG_SCALE_DIV, g.gm[i], H, Nv,
                                       p->st_rowptr, p->st_col, p->st_count, w->Wsb, 2 * H, H, g.pq, 2 * H, S, H, s));
            const float* A[4] = {v, g.gm[i], nullptr, nullptr};
            const int64_t ld[4] = {H, H, 0, 0}, Ks[4] = {H, H, 0, 0};
            // one launch where the row-panel kernel takes the shape (the story term formed on load), else two
            const int rc_nbr = bg_gemm_nbr_f32(Nv, H, 2, A, ld, Ks, w->Wc, w->ldwc, vn, H, nullptr, BG_ACT_LEAKY, 0.01f, v, H,
                                               g.pq, 2 * H, g.pq + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count,
                                               p->story_of_voxel, g.posb, H, g.y[i], H, s);
            if (rc_nbr == BG_ERR_SHAPE || rc_nbr == BG_ERR_ALIGN) {
                BG_TRY(bg_spmm_add_f32(g.pq + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count, BG_SCALE_DIV, nullptr,
                                       BG_SCALE_NONE, g.pq, 2 * H, nullptr, 0, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE,
                                       g.term, H, S, H, s));
                BG_TRY(bg_gemm_f32(0, 1, Nv, H, 2, A, ld, Ks, w->Wc, w->ldwc, vn, H, nullptr, nullptr, BG_ACT_LEAKY, 0.01f, v, H,
                                   g.term, H, p->story_of_voxel, g.posb, H, g.y[i], H, s));
            } else {
                BG_TRY(rc_nbr);
            }
        }
    }
    return BG_OK;
}

extern "C" int bg_gstack_backward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const bgGeneratorBackPlan* bp,
                                      const float* x, const float* voxel_in, const float* pdm4, int64_t n_calls,
                                      float* workspace, int64_t workspace_floats, const float* mask_soft,
                                      const float* att_soft, const float* g_mask_hard, const float* g_mask_soft,
                                      const float* g_att_hard, const float* g_att_soft, float* g_x, float* g_enc_w,
                                      float* g_enc_b, float* g_nnp_w, float* g_nnp_b, float* g_Wcat, float* g_bcat,
                                      float* g_theta, float* g_Wc, float* g_Wsb, bgStream s) {
    if (!w || !p || !bp || !x || !voxel_in || !pdm4 || !workspace || !mask_soft || !att_soft || !g_x || !g_enc_w || !g_enc_b ||
        !g_nnp_w || !g_nnp_b || !g_Wcat || !g_bcat || !g_theta || !g_Wc || !g_Wsb) return BG_ERR_ARG;
    if (!bp->vt_rowptr || !bp->vt_col || !bp->cp_rowptr || !bp->cp_col || !bp->cp_perm) return BG_ERR_ARG;
    int rc = gstack_shape_ok(w, p);
    if (rc != BG_OK) return rc;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    const int64_t H = w->H, Np = p->Np, Nv = p->Nv, S = p->S, Ec = p->Ec, L = w->voxel_layer;
    Arena ar{workspace, 0, workspace_floats};
    GStack g;
    rc = carve_gstack(ar, Np, Nv, S, Ec, H, L, g);
    if (rc != BG_OK) return rc;
    if (ar.used > workspace_floats || n_calls != g.n_att) return BG_ERR_ARG;
    const int nb = cross_logit_blocks(Nv);
    bool have_att = false, have_blk = false, have_np = false;
    BG_TRY(bg_fill_f32(g.zero, Nv * H, 0.f, s));
    const float* gin = g.zero;                 // gradient at the current call's output (the last call's features are unused)
    float* bufs[2] = {g.g0, g.g1};
    int flip = 0;
    for (int64_t k = g.n_ops - 1; k >= 0; --k) {
        const float* v = g.v[k];
        float* gout = bufs[flip];
        flip ^= 1;
        if (g.kind[k] == 0) {
            const int c = g.idx[k];
            const bool last = c == g.n_att - 1;
            const float* soft0 = last ? mask_soft : g.soft0[c];
            const float* soft = last ? att_soft : g.soft[c];
            float* G_al = g.G_al_all + (int64_t)c * Nv * (H + 4);
            float* dth_p = g.dth_all + (int64_t)c * nb * H;
            BG_TRY(bg_attention_bwd_head_f32(gin, H, x, H, soft0, soft, p->cx_rowptr, p->cx_col, last ? g_att_soft : nullptr,
                                             last ? g_att_hard : nullptr, last ? g_mask_soft : nullptr,
                                             last ? g_mask_hard : nullptr, -1.0f, g.gg, H, g.d_att, G_al + H, H + 4, Nv, H, s));
            if (!last) {                       // (the last call's gg is zero: nothing reaches x through its gather)
                float* dst = have_np ? g.tmpNp : g.gx_acc;
                if (bp->cp_one_launch && have_np) {      // sum and accumulation in one launch (same bits as sum, then add)
                    BG_TRY(bg_segsum_acc_f32(g.gg, H, bp->cp_rowptr, bp->cp_col, soft, bp->cp_perm, g.gx_acc, H, Np, H, s));
                } else {
                    if (bp->cp_one_launch)
                        BG_TRY(bg_segreduce_sum_f32(g.gg, H, bp->cp_rowptr, bp->cp_col, soft, bp->cp_perm, nullptr, BG_SCALE_NONE,
                                                    nullptr, 0, 0, nullptr, dst, H, Np, H, s));
                    else
                        BG_TRY(bg_spmm_f32(g.gg, H, bp->cp_rowptr, bp->cp_col, soft, bp->cp_perm, nullptr, BG_SCALE_NONE, nullptr,
                                           BG_SCALE_NONE, dst, H, Np, H, s));
                    if (have_np) BG_TRY(bg_add_f32(g.gx_acc, g.tmpNp, g.gx_acc, Np * H, s));
                }
            }
            BG_TRY(bg_cross_logits_bwd_f32(g.a[c], H, g.nnp, H, w->theta, p->cx_rowptr, p->cx_col, g.d_att, g.dpre, G_al,
                                           H + 4, dth_p, nb, Nv, H, s));
            if (bp->cp_one_launch && have_att) {
                BG_TRY(bg_segsum_acc_f32(g.dpre, H, bp->cp_rowptr, bp->cp_perm, nullptr, nullptr, g.gb_acc, H, Np, H, s));
            } else {
                float* dst = have_att ? g.tmpNp : g.gb_acc;
                if (bp->cp_one_launch)
                    BG_TRY(bg_segreduce_sum_f32(g.dpre, H, bp->cp_rowptr, bp->cp_perm, nullptr, nullptr, nullptr, BG_SCALE_NONE,
                                                nullptr, 0, 0, nullptr, dst, H, Np, H, s));
                else
                    BG_TRY(bg_spmm_f32(g.dpre, H, bp->cp_rowptr, bp->cp_perm, nullptr, nullptr, nullptr, BG_SCALE_NONE, nullptr,
                                       BG_SCALE_NONE, dst, H, Np, H, s));
                if (have_att) BG_TRY(bg_add_f32(g.gb_acc, g.tmpNp, g.gb_acc, Np * H, s));
            }
            if (!last && !have_np) have_np = true;
            {   // (the weight gradients of all attention calls: one product / column sum each after the loop)
                const float* A[4] = {G_al, nullptr, nullptr, nullptr};
                const int64_t ld[4] = {H + 4, 0, 0, 0}, Ks[4] = {H + 4, 0, 0, 0};
                BG_TRY(bg_gemm_f32(0, 0, Nv, H, 1, A, ld, Ks, w->Wcat, H, gout, H, nullptr, nullptr, BG_ACT_NONE, 0.f, gin, H,
                                   nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
            }
            have_att = true;
        } else {
            const int i = g.idx[k];
            float* gz = g.gz_all + (int64_t)i * Nv * H;
            float* gterm = g.gterm_all + (int64_t)i * S * H;
            float* story = g.story_all + (int64_t)i * S * H;
            BG_TRY(bg_story_bwd_f32(gin, H, g.y[i], H, v, H, p->st_rowptr, p->st_col, p->st_count, 0.01f, gz, H, gterm, H,
                                    story, H, S, H, s));
            // (the weight gradients of all block calls: one product each after the loop, over the stacked gz / gterm;
            //  the story -> building mean of every call's story term, which only that product reads: one launch there)
            {
                const float* A[4] = {gz, nullptr, nullptr, nullptr};
                const int64_t ld[4] = {H, 0, 0, 0}, Ks[4] = {H, 0, 0, 0};
                BG_TRY(bg_gemm_f32(0, 0, Nv, 2 * H, 1, A, ld, Ks, w->Wc, w->ldwc, g.gx, 2 * H, nullptr, nullptr, BG_ACT_NONE, 0.f,
                                   nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
                A[0] = gterm;
                BG_TRY(bg_gemm_f32(0, 0, S, 2 * H, 1, A, ld, Ks, w->Wsb, 2 * H, g.gsb, 2 * H, nullptr, nullptr, BG_ACT_NONE, 0.f,
                                   nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
            }
            BG_TRY(bg_spmm_add_f32(g.gsb + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count, BG_SCALE_DIV, nullptr, BG_SCALE_NONE,
                                   g.gsb, 2 * H, nullptr, 0, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE, g.g_story, H, S, H, s));
            BG_TRY(bg_spmm_add_f32(g.gx + H, 2 * H, bp->vt_rowptr, bp->vt_col, nullptr, BG_SCALE_NONE, p->vx_count, BG_SCALE_DIV,
                                   gin, H, g.gx, 2 * H, g.g_story, H, p->story_of_voxel, p->st_count, BG_SCALE_DIV, gout, H, Nv,
                                   H, s));
            have_blk = true;
        }
        gin = gout;
    }
    // shared weights: the sum over the calls IS the reduction dimension of one product over the stacked rows
    if (have_blk) {
        BG_TRY(bg_spmm_batch_f32(g.story_all, H, S * H, p->sb_rowptr, p->sb_col, p->sb_count, BG_SCALE_DIV, nullptr,
                                 BG_SCALE_NONE, g.sb_all, H, S * H, S, H, g.n_blk, s));
        BG_TRY(bg_tile_f32(pdm4, g.pdm4_rep, Nv * 4, g.n_blk, s));
        const float* Bs[4] = {g.vb, g.gm_all, g.pdm4_rep, nullptr};
        const int64_t ldb[4] = {H, H, 4, 0}, Ns[4] = {H, H, 4, 0};
        BG_TRY(tn_multi(H, g.n_blk * Nv, g.gz_all, 3, Bs, ldb, Ns, g_Wc, g.splitk, 0, s));
        const float* Bt[4] = {g.story_all, g.sb_all, nullptr, nullptr};
        const int64_t ldt[4] = {H, H, 0, 0}, Nt[4] = {H, H, 0, 0};
        BG_TRY(tn_multi(H, g.n_blk * S, g.gterm_all, 2, Bt, ldt, Nt, g_Wsb, g.splitk, 0, s));
    }
    if (have_att) {
        const float* Bs[4] = {g.va, nullptr, nullptr, nullptr};
        const int64_t ldb[4] = {H, 0, 0, 0}, Ns[4] = {H, 0, 0, 0};
        BG_TRY(tn_multi(H + 4, g.n_att * Nv, g.G_al_all, 1, Bs, ldb, Ns, g_Wcat, g.splitk, 0, s));
        BG_TRY(bg_colsum_f32(g.G_al_all, H + 4, g.n_att * Nv, H + 4, g.colsum, colsum_blocks(g.n_att * Nv), g_bcat, 0, s));
        BG_TRY(bg_colsum_f32(g.dth_all, H, g.n_att * nb, H, g.colsum, colsum_blocks(g.n_att * nb), g_theta, 0, s));
    }
    // encoder and NN_p
    {
        const float* Bs[4] = {voxel_in, nullptr, nullptr, nullptr};
        const int64_t ldb[4] = {4, 0, 0, 0}, Ns[4] = {4, 0, 0, 0};
        BG_TRY(tn_multi(H, Nv, gin, 1, Bs, ldb, Ns, g_enc_w, g.splitk, 0, s));
        BG_TRY(bg_colsum_f32(gin, H, Nv, H, g.colsum, colsum_blocks(Nv), g_enc_b, 0, s));
        Bs[0] = x;
        const int64_t ldx[4] = {H, 0, 0, 0}, Nx[4] = {H, 0, 0, 0};
        BG_TRY(tn_multi(H, Np, g.gb_acc, 1, Bs, ldx, Nx, g_nnp_w, g.splitk, 0, s));
        BG_TRY(bg_colsum_f32(g.gb_acc, H, Np, H, g.colsum, colsum_blocks(Np), g_nnp_b, 0, s));
        const float* A[4] = {g.gb_acc, nullptr, nullptr, nullptr};
        const int64_t ld[4] = {H, 0, 0, 0}, Ks[4] = {H, 0, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, Np, H, 1, A, ld, Ks, w->nnp_w, H, g_x, H, nullptr, nullptr, BG_ACT_NONE, 0.f, nullptr, 0,
                           nullptr, 0, nullptr, have_np ? g.gx_acc : nullptr, have_np ? H : 0, nullptr, 0, s));
    }
    return BG_OK;
}

// ---------------------------------------------------------------- program GNN: forward that saves, backward
namespace {
struct PStack {
    float *x[33], *both[32], *aggr[32], *y[32];
    // backward: the layers' gz / gpre and the two halves of `both` as contiguous [L Np, H] stacks -- the shared weights'
    // gradients are ONE product (and one column sum) each over the stacked layers after the loop, as in the voxel stack
    float *gz_all, *gpre_all, *top_all, *bot_all;
    float *g0, *g1, *gcat, *gcat2, *t, *splitk, *colsum;
};
int carve_pstack(Arena& ar, int64_t Np, int64_t H, int64_t kin, int64_t L, PStack& b) {
    if (L < 0 || L > 32) return BG_ERR_SHAPE;
    for (int64_t l = 0; l <= L; ++l) b.x[l] = ar.take(Np * H);                 // x[0 .. L-1] and aggr[.]: contiguous stacks
    for (int64_t l = 0; l < L; ++l) b.aggr[l] = ar.take(Np * H);
    for (int64_t l = 0; l < L; ++l) { b.both[l] = ar.take(2 * Np * H); b.y[l] = ar.take(Np * H); }
    const int64_t Ls = L > 0 ? L : 1;
    b.gz_all = ar.take(Ls * Np * H); b.gpre_all = ar.take(Ls * Np * H);
    b.top_all = ar.take(Ls * Np * H); b.bot_all = ar.take(Ls * Np * H);
    b.g0 = ar.take(Np * H); b.g1 = ar.take(Np * H); b.gcat = ar.take(Np * 3 * H);
    b.gcat2 = ar.take(Np * 2 * H); b.t = ar.take(Np * H);
    int64_t sk = 0;
    auto need = [&](int64_t M, std::initializer_list<int64_t> ns, int64_t R) {
        int64_t Ns[4] = {0, 0, 0, 0}, N = 0; int n = 0;
        for (int64_t v : ns) { Ns[n++] = v; N += v; }
        const int64_t x = (int64_t)tn_splits_multi(M, Ns, n, R) * M * N;
        if (x > sk) sk = x;
    };
    need(H, {H, H, H}, Ls * Np); need(H, {H, H}, Ls * Np); need(H, {kin}, Np);
    b.splitk = ar.take(sk);
    b.colsum = ar.take(256 * H);
    return BG_OK;
}
}  // namespace

extern "C" int64_t bg_pstack_workspace(int64_t Np, int64_t H, int64_t kin, int64_t L) {
    if (Np <= 0 || H <= 0 || kin <= 0 || L < 0) return -1;
    Arena ar{nullptr, 0, 0};
    PStack b;
    if (carve_pstack(ar, Np, H, kin, L, b) != BG_OK) return -1;
    return ar.used + 64;
}

extern "C" int bg_pstack_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const float* program_in,
                                     float* workspace, int64_t workspace_floats, float* x_out, bgStream s) {
    if (!w || !p || !program_in || !workspace || !x_out) return BG_ERR_ARG;
    if (!w->p_enc_w || !w->p_enc_b || !w->p_msg_w || !w->p_msg_b || !w->p_upd_w || !w->p_upd_b || !p->pp_rowptr || !p->pp_col ||
        !p->pp_scale || !p->prog_has) return BG_ERR_ARG;
    const int64_t H = w->H, Np = p->Np, L = w->program_layer, kin = w->n_class + 1 + w->noise_dim;
    if (H <= 0 || Np <= 0 || L < 0 || kin <= 0) return BG_ERR_ARG;
    if (H % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    PStack b;
    int rc = carve_pstack(ar, Np, H, kin, L, b);
    if (rc != BG_OK) return rc;
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    BG_TRY(linear(Np, H, 1, program_in, kin, nullptr, 0, nullptr, 0, kin, w->p_enc_w, kin, L == 0 ? x_out : b.x[0], w->p_enc_b,
                  nullptr, BG_ACT_NONE, nullptr, nullptr, 0, nullptr, nullptr, s));
    for (int64_t l = 0; l < L; ++l) {
        const float* x = b.x[l];
        float* xn = l == L - 1 ? x_out : b.x[l + 1];
        BG_TRY(bg_spmm_f32(x, H, p->pp_rowptr, p->pp_col, nullptr, nullptr, p->pp_scale, BG_SCALE_MUL, nullptr, BG_SCALE_NONE,
                           b.both[l], H, 2 * Np, H, s));
        BG_TRY(linear(Np, H, 2, x, H, b.both[l], H, nullptr, 0, H, w->p_msg_w, 2 * H, b.aggr[l], w->p_msg_b, p->prog_has,
                      BG_ACT_NONE, nullptr, nullptr, 0, nullptr, nullptr, s));
        const float* A[4] = {x, b.aggr[l], b.both[l] + Np * H, nullptr};
        const int64_t ld[4] = {H, H, H, 0}, Ks[4] = {H, H, H, 0};
        BG_TRY(bg_gemm_f32(0, 1, Np, H, 3, A, ld, Ks, w->p_upd_w, 3 * H, xn, H, w->p_upd_b, nullptr, BG_ACT_LEAKY, 0.01f, x, H,
                           nullptr, 0, nullptr, nullptr, 0, b.y[l], H, s));
    }
    return BG_OK;
}

extern "C" int bg_pstack_backward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const int32_t* pr_rowptr,
                                      const int32_t* pr_col, const float* pf_count, const float* program_in, const float* g,
                                      float* workspace, int64_t workspace_floats, float* g_enc_w, float* g_enc_b,
                                      float* g_msg_w, float* g_msg_b, float* g_upd_w, float* g_upd_b, bgStream s) {
    if (!w || !p || !pr_rowptr || !pr_col || !pf_count || !program_in || !g || !workspace || !g_enc_w || !g_enc_b || !g_msg_w ||
        !g_msg_b || !g_upd_w || !g_upd_b) return BG_ERR_ARG;
    const int64_t H = w->H, Np = p->Np, L = w->program_layer, kin = w->n_class + 1 + w->noise_dim;
    if (H <= 0 || Np <= 0 || L < 0 || kin <= 0) return BG_ERR_ARG;
    if (H % 4 != 0) return BG_ERR_SHAPE;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    PStack b;
    int rc = carve_pstack(ar, Np, H, kin, L, b);
    if (rc != BG_OK) return rc;
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const float* gin = g;
    float* bufs[2] = {b.g0, b.g1};
    int flip = 0;
    if (L == 0) {      // no block: nothing accumulates into these
        BG_TRY(bg_fill_f32(g_msg_w, H * 2 * H, 0.f, s)); BG_TRY(bg_fill_f32(g_msg_b, H, 0.f, s));
        BG_TRY(bg_fill_f32(g_upd_w, H * 3 * H, 0.f, s)); BG_TRY(bg_fill_f32(g_upd_b, H, 0.f, s));
    }
    for (int64_t l = L - 1; l >= 0; --l) {
        float* gout = bufs[flip];
        flip ^= 1;
        float* gz = b.gz_all + l * Np * H;
        float* gpre = b.gpre_all + l * Np * H;
        BG_TRY(bg_leaky_bwd_f32(gin, b.y[l], 0.01f, gz, Np * H, s));
        {
            const float* A[4] = {gz, nullptr, nullptr, nullptr};
            const int64_t ld[4] = {H, 0, 0, 0}, Ks[4] = {H, 0, 0, 0};
            BG_TRY(bg_gemm_f32(0, 0, Np, 3 * H, 1, A, ld, Ks, w->p_upd_w, 3 * H, b.gcat, 3 * H, nullptr, nullptr, BG_ACT_NONE, 0.f,
                               nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
        }
        BG_TRY(bg_rowscale_f32(b.gcat + H, 3 * H, p->prog_has, gpre, H, Np, H, s));
        {
            const float* A[4] = {gpre, nullptr, nullptr, nullptr};
            const int64_t ld[4] = {H, 0, 0, 0}, Ks[4] = {H, 0, 0, 0};
            BG_TRY(bg_gemm_f32(0, 0, Np, 2 * H, 1, A, ld, Ks, w->p_msg_w, 2 * H, b.gcat2, 2 * H, nullptr, nullptr, BG_ACT_NONE, 0.f,
                               nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
        }
        // g + gx(update) + transpose of the in-edge mean applied to g_gm ...
        BG_TRY(bg_spmm_add_f32(b.gcat2 + H, 2 * H, pr_rowptr, pr_col, nullptr, BG_SCALE_NONE, pf_count, BG_SCALE_DIV, gin, H,
                               b.gcat, 3 * H, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE, b.t, H, Np, H, s));
        // ... + gx(message) + the class-slot operator (symmetric: rows [Np, 2 Np) of the pair operator) applied to g_on
        BG_TRY(bg_spmm_add_f32(b.gcat + 2 * H, 3 * H, p->pp_rowptr + Np, p->pp_col, p->pp_scale + Np, BG_SCALE_MUL, nullptr,
                               BG_SCALE_NONE, b.t, H, b.gcat2, 2 * H, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE, gout, H, Np,
                               H, s));
        gin = gout;
    }
    if (L > 0) {
        // the block's weights are shared by the layers: their gradients' sum over the layers IS the reduction dimension of one
        // product over the stacked rows.  `both` holds [neighbour mean; class-slot rows] per layer; its halves are copied
        // into two stacks by one launch per 4 layers (the other operands are carved as stacks)
        const int64_t bytes = (int64_t)sizeof(float) * Np * H;
        for (int64_t l0 = 0; l0 < L; l0 += 4) {
            void* dst[8]; const void* src[8]; int64_t wd[8], one[8];
            int n = 0;
            for (int64_t l = l0; l < L && l < l0 + 4; ++l) {
                dst[n] = b.top_all + l * Np * H; src[n] = b.both[l]; wd[n] = bytes; one[n++] = 1;
                dst[n] = b.bot_all + l * Np * H; src[n] = b.both[l] + Np * H; wd[n] = bytes; one[n++] = 1;
            }
            BG_TRY(bg_fill_copy_batch(n, dst, wd, src, wd, wd, one, s));
        }
        const int64_t Rl = L * Np;
        {
            const float* Bs[4] = {b.x[0], b.aggr[0], b.bot_all, nullptr};
            const int64_t ldb[4] = {H, H, H, 0}, Ns[4] = {H, H, H, 0};
            BG_TRY(tn_multi(H, Rl, b.gz_all, 3, Bs, ldb, Ns, g_upd_w, b.splitk, 0, s));
        }
        BG_TRY(bg_colsum_f32(b.gz_all, H, Rl, H, b.colsum, colsum_blocks(Rl), g_upd_b, 0, s));
        {
            const float* Bs[4] = {b.x[0], b.top_all, nullptr, nullptr};
            const int64_t ldb[4] = {H, H, 0, 0}, Ns[4] = {H, H, 0, 0};
            BG_TRY(tn_multi(H, Rl, b.gpre_all, 2, Bs, ldb, Ns, g_msg_w, b.splitk, 0, s));
        }
        BG_TRY(bg_colsum_f32(b.gpre_all, H, Rl, H, b.colsum, colsum_blocks(Rl), g_msg_b, 0, s));
    }
    {
        const float* Bs[4] = {program_in, nullptr, nullptr, nullptr};
        const int64_t ldb[4] = {kin, 0, 0, 0}, Ns[4] = {kin, 0, 0, 0};
        BG_TRY(tn_multi(H, Np, gin, 1, Bs, ldb, Ns, g_enc_w, b.splitk, 0, s));
        BG_TRY(bg_colsum_f32(gin, H, Np, H, b.colsum, colsum_blocks(Np), g_enc_b, 0, s));
    }
    return BG_OK;
}

// ---------------------------------------------------------------- generator objective through the frozen discriminator
namespace {
struct GLossBuffers {
    float *Zl, *v, *Xs, *Xb, *gp_s, *gp_b, *vals, *valb, *part, *loss2, *dvb, *dvs, *dXb, *T, *dXs, *dZl, *gl;
    int32_t *as, *ab;
};
GLossBuffers carve_gloss(Arena& ar, const bgCriticHead& h) {
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, dl = h.dl, H2 = h.H2;
    GLossBuffers b;
    b.Zl = ar.take(N * K1 * dl); b.v = ar.take(N * H2); b.Xs = ar.take(S * H2); b.Xb = ar.take(B * H2);
    b.as = reinterpret_cast<int32_t*>(ar.take(S * H2)); b.ab = reinterpret_cast<int32_t*>(ar.take(B * H2));
    b.gp_s = ar.take(H2); b.gp_b = ar.take(H2); b.vals = ar.take(S); b.valb = ar.take(B); b.part = ar.take(128);
    b.loss2 = ar.take(4);
    b.dvb = ar.take(B); b.dvs = ar.take(S); b.dXb = ar.take(B * H2); b.T = ar.take(S * H2);
    b.dXs = ar.take(S * H2); b.dZl = ar.take(N * K1 * dl); b.gl = ar.take(N * dl);
    return b;
}
}  // namespace

extern "C" int64_t bg_gloss_head_workspace(const bgCriticHead* h) {
    if (!h || critic_shape_ok(*h) != BG_OK) return -1;
    Arena ar{nullptr, 0, 0};
    carve_gloss(ar, *h);
    return ar.used + 64;
}

extern "C" int bg_gloss_head_forward_f32(const bgCriticHead* hp, const float* label, int64_t ldl, int64_t n_lab,
                                         float* workspace, int64_t workspace_floats, float* loss, bgStream s) {
    if (!hp || !label || !workspace || !loss) return BG_ERR_ARG;
    const bgCriticHead& h = *hp;
    int rc = critic_shape_ok(h);
    if (rc != BG_OK) return rc;
    if (n_lab <= 0 || n_lab > h.dl || ldl < n_lab) return BG_ERR_ARG;
    if (!h.Zs || !h.W || !h.b_w1 || !h.b_b1 || !h.b_w2 || !h.b_b2 || !h.s_w1 || !h.s_b1 || !h.s_w2 || !h.s_b2 || !h.wb || !h.ws ||
        !h.st_rowptr || !h.st_col || !h.sg_rowptr || !h.sg_col || !h.a_rowptr || !h.a_col || !h.a_count || !h.blk_ptr)
        return BG_ERR_ARG;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    GLossBuffers b = carve_gloss(ar, h);
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, ds = h.ds, dl = h.dl, H2 = h.H2, Hd = h.Hd;
    const int64_t ldz = K1 * dl;
    {   // label into column group 0 (zero-padded to dl), then the powers of A
        const float* src[4] = {label, nullptr, nullptr, nullptr};
        const int64_t lds[4] = {ldl, 0, 0, 0};
        BG_TRY(bg_pack_columns_f32(src, lds, 1, n_lab, dl, b.Zl, ldz, N, s));
    }
    BG_TRY(bg_spmm_chain_f32(b.Zl, ldz, h.a_rowptr, h.a_col, h.a_count, BG_SCALE_DIV, nullptr, BG_SCALE_NONE, h.blk_ptr, h.blk_n,
                             h.blk_max_rows, h.blk_max_nnz, h.blk_max_deg, nullptr, 0, 0, b.Zl + dl, ldz, dl, h.L, dl,
                             h.basis ? BG_CHAIN_CHEB : BG_CHAIN_POWER, s));
    {
        const float* A[4] = {h.Zs, b.Zl, nullptr, nullptr};
        const int64_t ld[4] = {K1 * ds, ldz, 0, 0}, Ks[4] = {K1 * ds, ldz, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, N, H2, 2, A, ld, Ks, h.W, H2, b.v, H2, nullptr, nullptr, BG_ACT_NONE, 0.f, nullptr, 0, nullptr, 0,
                           nullptr, nullptr, 0, nullptr, 0, s));
    }
    BG_TRY(bg_segmax_f32(b.v, H2, h.st_rowptr, h.st_col, nullptr, 0, 0, nullptr, nullptr, b.Xs, H2, b.as, S, H2, s));
    BG_TRY(bg_segmax_f32(b.Xs, H2, h.sg_rowptr, h.sg_col, nullptr, 0, 0, nullptr, nullptr, b.Xb, H2, b.ab, B, H2, s));
    BG_TRY(bg_decoder_fwd_pair_f32(H2, Hd, S, b.Xs, H2, h.s_w1, H2, h.s_b1, h.s_w2, h.s_b2, b.vals, b.gp_s,
                                   B, b.Xb, H2, h.b_w1, H2, h.b_b1, h.b_w2, h.b_b2, b.valb, b.gp_b, s));
    BG_TRY(bg_critic_loss_f32(b.valb, h.wb, B, b.vals, h.ws, S, nullptr, 0, 0, 0, 0.f, b.part, loss, s));
    return BG_OK;
}

extern "C" int bg_gloss_head_backward_f32(const bgCriticHead* hp, int64_t n_lab, const float* go, float* workspace,
                                          int64_t workspace_floats, float* g_label, bgStream s) {
    if (!hp || !go || !workspace || !g_label) return BG_ERR_ARG;
    const bgCriticHead& h = *hp;
    int rc = critic_shape_ok(h);
    if (rc != BG_OK) return rc;
    if (n_lab <= 0 || n_lab > h.dl) return BG_ERR_ARG;
    if (!h.story_of_voxel || !h.graph_of_story || !h.at_rowptr || !h.at_col) return BG_ERR_ARG;
    if (!bg_aligned16(workspace)) return BG_ERR_ALIGN;
    Arena ar{workspace, 0, workspace_floats};
    GLossBuffers b = carve_gloss(ar, h);
    if (ar.used > workspace_floats) return BG_ERR_ARG;
    const int64_t N = h.N, S = h.S, B = h.B, K1 = h.L + 1, ds = h.ds, dl = h.dl, H2 = h.H2, Hd = h.Hd;
    const int64_t ldz = K1 * dl;
    const float* Wl = h.W + K1 * ds * H2;
    BG_TRY(bg_critic_loss_bwd_f32(go, h.wb, B, h.ws, S, nullptr, 0, 0, 0, 0.f, b.dvb, b.dvs, nullptr, s));
    // the decoders are frozen here: only d X = dv (x) (w2 W1), one launch each
    BG_TRY(bg_decoder_bwd_f32(B, 0, H2, Hd, b.dvb, b.Xb, H2, nullptr, nullptr, 0, nullptr, 0, nullptr, b.gp_b, nullptr, 0,
                              nullptr, 0, nullptr, nullptr, nullptr, b.dXb, H2, s));
    BG_TRY(bg_scatter_arg_f32(b.dXb, H2, b.ab, h.graph_of_story, b.T, H2, B, S, H2, s));
    BG_TRY(bg_decoder_bwd_f32(S, 0, H2, Hd, b.dvs, b.Xs, H2, nullptr, nullptr, 0, nullptr, 0, nullptr, b.gp_s, b.T, H2,
                              nullptr, 0, nullptr, nullptr, nullptr, b.dXs, H2, s));
    BG_TRY(bg_scatter_arg_f32(b.dXs, H2, b.as, h.story_of_voxel, b.v, H2, S, N, H2, s));          // v is free: it takes d v
    BG_TRY(mm(N, ldz, H2, b.v, H2, Wl, H2, 1, b.dZl, nullptr, nullptr, s));                       // d Zl = d v Wl^T
    BG_TRY(bg_spmm_chain_f32(b.dZl + h.L * dl, ldz, h.at_rowptr, h.at_col, nullptr, BG_SCALE_NONE, h.a_count, BG_SCALE_DIV,
                             h.blk_ptr, h.blk_n, h.blk_max_rows, h.blk_max_nnz, h.blk_max_deg,
                             b.dZl + (h.L - 1) * dl, ldz, -dl, b.gl, dl, 0, h.L, dl, h.basis ? BG_CHAIN_CLENSHAW : BG_CHAIN_POWER, s));
    {
        const float* src[4] = {b.gl, nullptr, nullptr, nullptr};
        const int64_t lds[4] = {dl, 0, 0, 0};
        BG_TRY(bg_pack_columns_f32(src, lds, 1, n_lab, n_lab, g_label, n_lab, N, s));
    }
    return BG_OK;
}
