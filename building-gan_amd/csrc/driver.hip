// Whole passes as one host call: the launch sequences of the graph-free paths, issued from C++ instead of from the
// interpreter (a training step over a stream of differently-shaped batches cannot replay a captured hipGraph and
// is bound by host launch time: ~17 us of Python per kernel against ~2 us here).  Every step calls the same
// extern "C" launcher with the same arguments as the op-by-op host path, so results are bit-identical.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

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

// BG_DRIVER_SYNC=1 (diagnosis): every step is named on stderr before it is issued and the stream is drained after
// it, so a faulting launch is the last line printed.
static int g_step_sync = -1;
static int step_sync() {
    if (g_step_sync < 0) {
        const char* e = getenv("BG_DRIVER_SYNC");
        g_step_sync = (e && e[0] == '1') ? 1 : 0;
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
        BG_TRY(bg_spmm_add_f32(b.pq + H, 2 * H, p->sb_rowptr, p->sb_col, p->sb_count, BG_SCALE_DIV, nullptr, BG_SCALE_NONE,
                               b.pq, 2 * H, nullptr, 0, nullptr, 0, nullptr, nullptr, BG_SCALE_NONE, b.term, H, S, H, s));
        BG_TRY(linear(Nv, H, 2, v, H, b.gm, H, nullptr, 0, H, w->Wc, w->ldwc, vn, nullptr, nullptr, BG_ACT_LEAKY, v, b.term,
                      H, p->story_of_voxel, b.posb, s));
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

extern "C" int bg_d_coef_forward_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b,
                                     const float* pe, const float* Wc, int64_t ldw, const float* bc, int64_t H,
                                     int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L,
                                     float* xall, float* efull, bgStream s) {
    if (!fe_w || !fe_b || !le_w || !le_b || !pe || !Wc || !bc || !xall || !efull) return BG_ERR_ARG;
    if (H <= 0 || L < 0 || ds < nf + ML + 5 || dl < nl || ldw < 4 * H + 4) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, du = ds + dl, N = (L + 1) * du, S = N + du;
    hipStream_t st = (hipStream_t)s;
    BG_HIP(hipMemsetAsync(xall, 0, sizeof(float) * (size_t)((L + 1) * S * H2), st));
    BG_HIP(hipMemsetAsync(efull, 0, sizeof(float) * (size_t)(N * H2), st));
    BG_TRY(bg_d_coef_init_f32(fe_w, fe_b, le_w, le_b, pe, Wc, ldw, bc, H, nf, nl, ML, ds, dl, xall + du * H2, efull, s));
    for (int64_t m = 0; m < L; ++m) {
        const int64_t rows = N < (m + 2) * du ? N : (m + 2) * du;       // blocks above m+1 are still zero
        float* cur = xall + m * S * H2;
        const float* A[4] = {cur + du * H2, cur, nullptr, nullptr};
        const int64_t ld[4] = {H2, H2, 0, 0}, Ks[4] = {H2, H2, 0, 0};
        BG_TRY(bg_gemm_f32(0, 1, rows, H2, 2, A, ld, Ks, Wc, ldw, xall + (m + 1) * S * H2 + du * H2, H2, nullptr, nullptr,
                           BG_ACT_NONE, 0.01f, cur + du * H2, H2, efull, H2, nullptr, nullptr, 0, nullptr, 0, s));
    }
    return BG_OK;
}

extern "C" int64_t bg_d_coef_backward_workspace(int64_t H, int64_t ds, int64_t dl, int64_t L) {
    if (H <= 0 || ds < 0 || dl < 0 || L < 0) return -1;
    const int64_t H2 = 2 * H, du = ds + dl, S = (L + 2) * du;
    return coef_splits(H2, L * S) * H2 * 2 * H2;
}

extern "C" int bg_d_coef_backward_f32(const float* g, const float* xall, const float* Wc, int64_t ldw, int64_t H,
                                      int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L,
                                      float* gall, float* wab, float* partial, float* gW, float* g_bc, float* g_fe_w,
                                      float* g_fe_b, float* g_le_w, float* g_le_b, bgStream s) {
    if (!g || !xall || !Wc || !gall || !wab || !partial || !gW || !g_bc || !g_fe_w || !g_fe_b || !g_le_w || !g_le_b)
        return BG_ERR_ARG;
    if (H <= 0 || L < 1 || ldw < 4 * H + 4) return BG_ERR_ARG;
    const int64_t H2 = 2 * H, du = ds + dl, N = (L + 1) * du, S = N + du;
    hipStream_t st = (hipStream_t)s;
    // G^m at rows [0, N) of slab m, du zero rows below it (the shift-up view reads them)
    BG_HIP(hipMemsetAsync(gall, 0, sizeof(float) * (size_t)((L + 1) * S * H2), st));
    BG_HIP(hipMemcpyAsync(gall + L * S * H2, g, sizeof(float) * (size_t)(N * H2), hipMemcpyDeviceToDevice, st));
    // Wab = [Wc[:, :2H]; Wc[:, 2H:4H]]  ([4H, 2H]): G Wa + shiftup(G) Wb
    BG_HIP(hipMemcpy2DAsync(wab, sizeof(float) * H2, Wc, sizeof(float) * ldw, sizeof(float) * H2, H2, hipMemcpyDeviceToDevice, st));
    BG_HIP(hipMemcpy2DAsync(wab + H2 * H2, sizeof(float) * H2, Wc + H2, sizeof(float) * ldw, sizeof(float) * H2, H2,
                            hipMemcpyDeviceToDevice, st));
    for (int64_t m = L - 1; m >= 0; --m) {
        const float* nxt = gall + (m + 1) * S * H2;
        const float* A[4] = {nxt, nxt + du * H2, nullptr, nullptr};
        const int64_t ld[4] = {H2, H2, 0, 0}, Ks[4] = {H2, H2, 0, 0};
        BG_TRY(bg_gemm_f32(0, 0, N, H2, 2, A, ld, Ks, wab, H2, gall + m * S * H2, H2, nullptr, nullptr, BG_ACT_NONE, 0.01f,
                           nxt, H2, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, s));
    }
    // dWc[:, :4H] = sum_m G^{m+1 T} [X^m | shift(X^m)]: one long reduction over the stacked slabs
    const float* B[4] = {xall + du * H2, xall, nullptr, nullptr};
    const int64_t ldb[4] = {H2, H2, 0, 0}, Ns[4] = {H2, H2, 0, 0};
    BG_TRY(bg_gemm_tn_splitk_f32(H2, L * S, gall + S * H2, H2, 2, B, ldb, Ns, gW, 2 * H2 + 4, partial,
                                 (int32_t)coef_splits(H2, L * S), 0, s));
    return bg_d_coef_init_bwd_f32(gall, S * H2, L, H, nf, nl, ML, ds, dl, g_fe_w, g_fe_b, g_le_w, g_le_b, gW + 2 * H2,
                                  2 * H2 + 4, g_bc, s);
}
