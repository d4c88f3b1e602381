/* libbgan_hip.so -- C ABI of the MI355X (gfx950) kernels behind Building-GAN's message-passing
 * hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer owned by the
 * caller; the library never allocates, frees or synchronises; every function enqueues its
 * kernels on `stream` (a hipStream_t passed as void*) and returns BG_OK or a negative code.
 *
 * The reference (AutodeskAILab/Building-GAN) has no native code: each entry point below
 * replaces a call the reference makes into torch_scatter / torch_geometric / torch.nn.  The
 * "replaces:" lines cite that call site under /root/reference/.
 *
 * Conventions: matrices are row-major float32 with an explicit row pitch `ld*` (in floats);
 * CSR plans are int32 (rowptr[n_rows+1], col[nnz]); "first index wins" on arg-max ties.
 *
 * State and environment: the shipped library keeps no state between calls (the one exception is
 * declared beside bg_segmax_static_f32) and reads NO environment variable: which kernel serves a
 * shape is a function of the arguments alone.  The BG_GEMM_*, BG_SPMM_* and BG_DRIVER_SYNC variables that
 * tools/ mention exist only in a build made with -DBG_TUNING_KNOBS (`make -C building-gan_amd/csrc
 * tuning`, csrc/common.h): they select among kernel variants for A/B timing, never change a result,
 * and are not part of this interface.
 */
#ifndef BGAN_H
#define BGAN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_OK 0
#define BG_ERR_ARG (-1)      /* null pointer / negative size / inconsistent sizes        */
#define BG_ERR_SHAPE (-2)    /* shape not supported by the kernel (see each function)   */
#define BG_ERR_ALIGN (-3)    /* pointer or pitch not aligned as required                */
#define BG_ERR_LAUNCH (-4)   /* HIP reported a launch / enqueue error                   */
#define BG_ERR_SCRATCH (-5)  /* request exceeds the library's static scratch buffer     */

typedef void* bgStream;

int bg_version(void);
const char* bg_error_string(int code);

/* ------------------------------------------------------------------ integer plans ------- */

/* Stable counting sort of E edges by key[e] in [0,N): rowptr[N+1], perm[E] (original edge id
 * of each CSR slot, ascending inside a row) and col[E] = other[perm].  cursor: int32[N+1]
 * scratch.  flags: int32[4], zeroed by the caller: [0] += keys outside [0,N); [1] += slots with
 * perm[k] != k; [2] += order violations key[e] < key[e-1] (0 => identity permutation, no sorting
 * is done); [3] = number of long rows that were sorted by whole workgroups.
 * replaces: the COO->segment bookkeeping inside torch_scatter that every scatter call of
 * Model/models.py:65,158 (MessagePassing.propagate), :71,:166-167 (avg_pool_x) performs.   */
int bg_plan_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N,
                int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* cursor, int32_t* flags,
                bgStream stream);

/* Pointer array of a non-decreasing id vector: ptr[s] = first i with seg_id[i] >= s, ptr[nseg]=n.
 * flags[0] += 1 if the ids are not non-decreasing or fall outside [0,nseg).
 * replaces: consecutive_cluster / torch.unique inside avg_pool_x, Model/models.py:166-167.  */
int bg_plan_segptr(const int64_t* seg_id, int64_t n, int64_t nseg, int32_t* ptr, int32_t* flags,
                   bgStream stream);

/* level[i] = vfc[i] - min{vfc[j] : vbi[j] == vbi[i]}.  first: int32[B] scratch.
 * replaces: get_voxel_floor_level (max_pool_x on -vfc), Model/models.py:22-25.              */
int bg_plan_floor_level(const int64_t* vfc, const int64_t* vbi, int64_t n, int64_t B,
                        int32_t* first, int32_t* level, bgStream stream);

/* count[i] = (float)max(rowptr[i+1]-rowptr[i], 1): the divisor of scatter-mean.
 * replaces: the clamp(count,1) of torch_scatter.scatter_mean (aggr='mean', models.py:55,147). */
int bg_plan_counts(const int32_t* rowptr, int64_t n_rows, float* count, bgStream stream);

/* Chunk pointer of a CSR / segment pointer: chunk_ptr[r+1]-chunk_ptr[r] = ceil(len_r / rows_per_chunk).
 * Used by the two-phase reductions over long segments (stories, graphs, program nodes).     */
int bg_plan_chunks(const int32_t* rowptr, int64_t n_rows, int32_t rows_per_chunk,
                   int32_t* chunk_ptr, bgStream stream);

/* ---- the whole index plan of a batch in one host call ------------------------------------------------------
 * Inputs: the batch's int64 index vectors as the reference's layout holds them (row a0 of SURVEY.md section 8).  */
typedef struct {
    int64_t Np, Nv, Ep, Ev, Ec, B, S, n_slot;
    const int64_t *pe_src, *pe_dst;          /* program_edge[0], program_edge[1]                 [Ep] */
    const int64_t *ve_src, *ve_dst;          /* voxel_edge[0], voxel_edge[1]                     [Ev] */
    const int64_t *cep, *cev;                /* cross_edge_{program,voxel}_index_select          [Ec] */
    const int64_t *vfc, *vbi;                /* voxel_floor_cluster, voxel_feature_batch         [Nv] */
    const int64_t *pcc;                      /* program_class_cluster                            [Np] */
    const int64_t *slot_graph;               /* program_target_ratio_batch                       [n_slot] */
    const float* target_ratio;               /* program_target_ratio                             [n_slot] */
    int32_t chunk_rows[4];                   /* rows per chunk of: story, graph, story-by-graph, cross-by-program (0: none) */
} bgPlanIn;

/* Pieces of the plan inside the two arenas (element offsets, every piece 256-byte aligned).  int32 pieces: */
enum {
    BGP_PF_ROWPTR, BGP_PF_COL, BGP_PF_PERM, BGP_PR_ROWPTR, BGP_PR_COL, BGP_PR_PERM,          /* program edges by target / source */
    BGP_SLOT_ROWPTR, BGP_SLOT_COL, BGP_SLOT_PERM, BGP_SG_ROWPTR, BGP_SG_COL, BGP_SG_PERM,    /* nodes by class slot, slots by graph */
    BGP_VFW_ROWPTR, BGP_VFW_COL, BGP_VFW_PERM, BGP_VRV_ROWPTR, BGP_VRV_COL, BGP_VRV_PERM,    /* voxel edges by target / source */
    BGP_ST_ROWPTR, BGP_ST_COL, BGP_ST_PERM, BGP_ST_CHUNK,                                    /* voxels by story */
    BGP_GR_ROWPTR, BGP_GR_COL, BGP_GR_PERM, BGP_GR_CHUNK,                                    /* voxels by graph */
    BGP_SGR_ROWPTR, BGP_SGR_COL, BGP_SGR_PERM, BGP_SGR_CHUNK,                                /* stories by graph */
    BGP_CV_ROWPTR, BGP_CV_COL, BGP_CV_PERM, BGP_CP_ROWPTR, BGP_CP_COL, BGP_CP_PERM, BGP_CP_CHUNK,   /* cross edges by voxel / program */
    BGP_ARANGE, BGP_PCC32, BGP_SLOTGRAPH32, BGP_VFC32, BGP_VBI32, BGP_STORYGRAPH32, BGP_CEV32, BGP_CEP32,
    BGP_FIRST, BGP_LEVEL, BGP_SS_ROWPTR, BGP_SS_COL, BGP_PP_ROWPTR, BGP_PP_COL, BGP_FLAGS, BGP_STATS, BGP_CURSOR,
    BGP_STORYGRAPH64,                                                                        /* (int64: two int32 slots per entry) */
    BGP_N_I32
};
/* float pieces: */
enum { BGP_PF_COUNT, BGP_SLOT_COUNT, BGP_VFW_COUNT, BGP_ST_COUNT, BGP_SGR_COUNT, BGP_SS_COUNT, BGP_PP_SCALE, BGP_VOX_HAS,
       BGP_PROG_HAS, BGP_N_F32 };
/* BGP_FLAGS: 11 x 4 validity flags, one quadruple per CSR in the order pf, pr, slot, sg, vfw, vrv, st, gr, sgr, cv, cp
 * (meaning as bg_plan_csr).  BGP_STATS (int32): */
enum { BGS_MAX_LEVEL, BGS_MIN_VDEG, BGS_MIN_PDEG, BGS_MIN_CROSS, BGS_MIN_STORY, BGS_MAXLEN_ST, BGS_MAXLEN_GR, BGS_MAXLEN_SGR,
       BGS_MAXLEN_CP, BGS_MAXLEN_CV, BGS_NNZ_SS, BGS_N_LEAVING, BGS_BLK_NNZ, BGS_VOX_MAX_DEG, BGS_NNZ_SLOT, BGS_N };

/* Element offsets of every piece (offs_i32[BGP_N_I32], offs_f32[BGP_N_F32]) and the arena sizes (total[0] int32
 * elements, total[1] floats) for these sizes.  BG_ERR_SHAPE when S * S or Np * Np exceeds 2^22 (the two derived
 * operators are laid out for their upper bound) or a size is zero: build the plan piece by piece instead.          */
int bg_plan_layout(const bgPlanIn* in, int64_t* offs_i32, int64_t* offs_f32, int64_t* total);

/* Every CSR, segment table, chunk table, floor level, derived operator and statistic of a batch's plan (what
 * building-gan_amd/plan.py otherwise assembles from ~25 bg_plan_* calls and ~100 host-issued tensor operations) by
 * ONE host call: 13 launches (+ one memset) issued back to back, no synchronisation; the caller reads BGP_FLAGS / BGP_STATS back
 * once.  arena_i32 must be 16-byte aligned.
 * replaces: the COO bookkeeping of every torch_scatter / torch_geometric call of a pass (see bg_plan_csr).          */
int bg_plan_build(const bgPlanIn* in, int32_t* arena_i32, float* arena_f32, bgStream stream);

/* ------------------------------------------------------------------ segment kernels ----- */

/* Scale modes for bg_spmm_f32 / bg_segreduce_f32 */
#define BG_SCALE_NONE 0
#define BG_SCALE_MUL 1
#define BG_SCALE_DIV 2

/* out[i,:] = rs(i) * sum_{k=rowptr[i]}^{rowptr[i+1]-1} w(k) * cs(c_k) * x[c_k,:],  c_k = col ? col[k] : k,
 * w(k) = edge_w ? edge_w[eperm ? eperm[k] : k] : 1; rs / cs multiply or divide by row_scale[i] /
 * col_scale[c] according to row_mode / col_mode.  Terms are added in slot order (the order
 * torch_scatter's CPU kernels use), one wave per few rows, 16-byte gathers.  H % 4 == 0 uses the
 * vector path (x, out 16-byte aligned, pitches % 4 == 0); any other H a scalar path.
 * replaces: x.index_select + scatter(mean|sum) pairs: Model/models.py:65-68,158-161 (K1,K3),
 * :71-72 (K4), :122-123 (K11), :133-138 (K12), :168-169; util_graph.py:28-40 (K14).         */
int bg_spmm_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                const float* edge_w, const int32_t* eperm,
                const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream);

/* bg_spmm_f32 (no edge weights) on n_batch matrices under ONE operator in one launch: batch b reads x + b * x_step
 * and writes out + b * out_step (steps in elements).  Same kernel and arithmetic as the single call, bit for bit.  The
 * one-launch form takes H % 4 == 0, H <= 256 and 16-byte aligned operands; other shapes run bg_spmm_f32 per batch.
 * replaces: the story -> building mean of Model/models.py:166-168 under autograd, which the backward of the generator's
 * 11 VoxelBlock calls needs once per call for the shared weight's gradient -- here once for all calls.            */
int bg_spmm_batch_f32(const float* x, int64_t ldx, int64_t x_step, const int32_t* rowptr, const int32_t* col,
                      const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                      float* out, int64_t ldo, int64_t out_step, int64_t n_rows, int64_t H, int64_t n_batch, bgStream stream);

/* bg_spmm_f32 (without edge weights) plus up to two addend matrices and one GATHERED addend fused into the
 * store:  out[i,:] = rs(i) * sum_k cs(c_k) x[c_k,:] + add1[i,:] + add2[i,:] + gs(grow[i]) gadd[grow[i],:]
 * (add1 / add2 / gadd may be NULL; gs = the gmode scale by gscale[grow[i]], BG_SCALE_*; the caller keeps
 * grow[i] inside gadd's rows).  Used by the backward of a residual message-passing layer:
 * g + gx_self + A^T gx_nbr + (the voxel's story's share) in one pass.
 * replaces: the index_add / `+` chain autograd runs for Model/models.py:198,228 (v = v + block(v)). */
int bg_spmm_add_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                    const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                    const float* add1, int64_t ld1, const float* add2, int64_t ld2,
                    const float* gadd, int64_t ldga, const int32_t* grow, const float* gscale, int gmode,
                    float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream);

/* n_steps applications of one BLOCK-DIAGONAL square operator in a single launch:
 *   y_0 = x0 ([n, W], pitch ldx);  y_k = rs * P (cs * y_{k-1}) + add_k,  k = 1..n_steps
 * with P, rs, cs as in bg_spmm_f32 (no edge weights), add_k = add + (k-1)*add_step (pitch ld_add; add
 * may be NULL; add_step may be negative).  y_k is written to out + (k-1)*out_step (pitch ldo);
 * out_step == 0 keeps only y_{n_steps}, at out.  Rows [blk_ptr[b], blk_ptr[b+1]) form diagonal block b
 * (the voxels of one graph): every column index of those rows must lie in the same range; max_rows /
 * max_nnz bound the rows / slots of a block, max_deg the slots of a row (<= 8: the column ids live in
 * registers).  One workgroup per (block, 4 columns) keeps the block in LDS between steps; BG_ERR_SHAPE when a block does not fit (the caller then launches the steps one by
 * one with bg_spmm_add_f32).  W % 4 == 0, 16-byte aligned operands.  Each step is the same arithmetic
 * in the same order as the separate launch.
 * basis selects the recurrence: BG_CHAIN_POWER as above (y_k = P^k y_0: the monomial basis);
 *   BG_CHAIN_CHEB      y_1 = P y_0, y_k = 2 P y_{k-1} - y_{k-2}  (y_k = T_k(P) y_0, Chebyshev polynomials of the
 *                      operator: a mean-aggregation operator has its spectrum in [-1, 1], where the T_k stay bounded
 *                      by 1 and the collapsed discriminator's coefficient table stays well conditioned);
 *   BG_CHAIN_CLENSHAW  y_k = 2 P y_{k-1} - y_{k-2} + add_k with factor 1 at the last step: for x0 = g_n and
 *                      add_k = g_{n-k} the result is sum_j T_j(P) g_j (Clenshaw), the transpose of BG_CHAIN_CHEB.
 * replaces: the repeated propagate() of one VoxelBlock over the SAME operator, Model/models.py:227-228
 * (12 layers of the discriminator's linear voxel GNN, evaluated as operator powers on the input columns,
 * and the transposed chain of its input gradient, Model/losses.py:83-84).                          */
enum { BG_CHAIN_POWER = 0, BG_CHAIN_CHEB = 1, BG_CHAIN_CLENSHAW = 2 };
int bg_spmm_chain_f32(const float* x0, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                      const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                      const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                      const float* add, int64_t ld_add, int64_t add_step,
                      float* out, int64_t ldo, int64_t out_step, int64_t n_steps, int64_t W, int32_t basis,
                      bgStream stream);

/* bg_spmm_chain_f32 on matrices kept as column GROUPS: the W columns are W / grp_w groups of grp_w columns
 * (grp_w % 4 == 0), and for a matrix with a non-zero group step (x0_gstep / add_gstep / out_gstep, in elements) group g
 * is its own [rows, grp_w] column block starting g * step elements behind group 0 -- the same row pitch, step and
 * add_step as group 0 -- instead of lying g * grp_w columns further right.  A step of 0 leaves that matrix a plain
 * [rows, W] one; all steps 0 (grp_w 0) is bg_spmm_chain_f32.  Same kernels, same arithmetic per column: the result
 * equals regrouping the operands, running the plain chain and regrouping back (bg_regroup_f32), bit for bit.
 * replaces: the permute / reshape copies either side of the penalty's chains in the critic head (the chain's operand
 * is a product's [2 N, (L+1) dl] output, rows = building and story halves; Model/losses.py:83-88 through
 * Model/models.py:150-172 transposed).                                                                           */
int bg_spmm_chain_grouped_f32(const float* x0, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                              const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                              const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                              const float* add, int64_t ld_add, int64_t add_step,
                              float* out, int64_t ldo, int64_t out_step, int64_t n_steps, int64_t W, int32_t basis,
                              int64_t grp_w, int64_t x0_gstep, int64_t add_gstep, int64_t out_gstep, bgStream stream);

/* bg_spmm_chain_f32 (no addend) on TWO matrices under the same operator and step count in one launch: the
 * column chunks of both fill the chip side by side (a chain's time does not depend on its width).  Short-row
 * plans only (max_deg <= 8 and max_rows <= 2048, or max_deg <= 6 and max_rows <= 3072), else BG_ERR_SHAPE and the caller launches them one by one.
 * basis: BG_CHAIN_POWER or BG_CHAIN_CHEB.
 * replaces: the label-free and the label operator powers of the collapsed discriminator, Model/models.py:227-228. */
int bg_spmm_chain_pair_f32(const float* x0, int64_t ldx, float* out, int64_t ldo, int64_t out_step, int64_t W,
                           const float* x0b, int64_t ldxb, float* outb, int64_t ldob, int64_t out_stepb, int64_t Wb,
                           const int32_t* rowptr, const int32_t* col,
                           const float* row_scale, int row_mode, const float* col_scale, int col_mode,
                           const int32_t* blk_ptr, int64_t n_blk, int64_t max_rows, int64_t max_nnz, int64_t max_deg,
                           int64_t n_steps, int32_t basis, bgStream stream);

/* Same contraction for LONG rows, two phases: phase 1 reduces chunks of at most rows_per_chunk
 * slots (chunk_ptr from bg_plan_chunks, n_chunks_ub >= chunk_ptr[n_rows]) into
 * partial[n_chunks_ub, H]; phase 2 adds a row's partials in chunk order.  Deterministic.
 * chunk_ptr == NULL selects the one-launch form: a workgroup per row walks the whole row and writes
 * the scaled sum (for many rows of moderate length, e.g. stories; partial is not used; needs the
 * 16-byte path: H % 4 == 0, aligned x / out, pitches % 4 == 0).
 * replaces: avg_pool_x over stories / graphs, Model/models.py:166-167 (K5); the program-node
 * side of the cross-edge scatters in util_graph.py:32 and the backward of models.py:113,122. */
int bg_segreduce_sum_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const float* edge_w, const int32_t* eperm,
                         const float* row_scale, int row_mode,
                         const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                         float* partial, float* out, int64_t ldo, int64_t n_rows, int64_t H,
                         bgStream stream);

/* Forward-only fusion of the generator block's per-story term: for every segment (story) s,
 * m = mean of its rows of x (count[s] = max(len, 1)), out[s, 0:Hout] = m W[:, 0:H]^T and
 * out[s, Hout:2Hout] = m W[:, H:2H]^T for a row-major W [Hout, 2H] (pitch ldw).  One workgroup per
 * segment; H % 4 == 0, H <= 256.
 * replaces: the avg_pool_x over a story and the story / building columns of updateMLP's Linear,
 * Model/models.py:166,170, when nothing is differentiated.                                    */
int bg_story_project_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const float* count, const float* W, int64_t ldw, int64_t Hout,
                         float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream);

/* Backward companion of bg_story_project_f32 for a block with a leaky ReLU: per story segment s (voxels c),
 *   gz[c] = g[c] * (y[c] > 0 ? 1 : slope),  gterm[s] = sum_c gz[c],  story[s] = (sum_c v[c]) / count[s]
 * in ONE launch (one workgroup per story).  The segments must partition the voxel rows (each exactly once),
 * or rows of gz stay unwritten.  H % 4 == 0, H <= 256, 16-byte aligned operands and pitches.
 * replaces: the backward of LeakyReLU and of the story gather of updateMLP's input, and the avg_pool_x
 * recomputation, Model/models.py:166-170, in the generator's backward.                              */
int bg_story_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* v, int64_t ldv,
                     const int32_t* rowptr, const int32_t* col, const float* count, float slope,
                     float* gz, int64_t ldz, float* gterm, int64_t ldt, float* story, int64_t lds,
                     int64_t n_story, int64_t H, bgStream stream);

/* bg_spmm_f32 (no edge weights, no column scale: gm = rs * P x over n_rows rows) and bg_story_project_f32
 * (pq over n_story segments) of the SAME x in ONE launch -- a generator block needs both of its input and
 * they are independent.  H % 4 == 0, H <= 256, 16-byte aligned operands.
 * replaces: the propagate() gather + scatter-mean and the avg_pool_x over stories of one VoxelBlock call,
 * Model/models.py:158,166, when nothing is differentiated.                                      */
int bg_gather_story_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                        const float* row_scale, int row_mode, float* gm, int64_t ldg, int64_t n_rows,
                        const int32_t* s_rowptr, const int32_t* s_col, const float* s_count,
                        const float* W, int64_t ldw, int64_t Hout, float* pq, int64_t ldq, int64_t n_story,
                        int64_t H, bgStream stream);

/* out[s,:] += sum_{k in seg s} w_k x[c_k,:]: bg_segreduce_sum_f32's one-launch form (no row scale) ADDED to what out
 * holds -- each segment's sum is formed in that call's order and then added, so the result equals that call followed
 * by an elementwise addition, in one launch.  Needs the 16-byte path (BG_ERR_ALIGN otherwise).
 * replaces: the accumulation of the program features' gradient over the Attention calls of one g_loss.backward()
 * (autograd's add of the gradients arriving at a tensor used several times, Model/models.py:185-203).           */
int bg_segsum_acc_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col, const float* edge_w,
                      const int32_t* eperm, float* out, int64_t ldo, int64_t n_rows, int64_t H, bgStream stream);

/* Segment max with arg: out[s,h] = max_{k in seg s} x[c_k,h], arg[s,h] = c_k of the FIRST maximum
 * (slot order); an empty segment gives 0 and arg = -1.  Two phases as above;
 * partial_val / partial_arg: [n_chunks_ub, H]; chunk_ptr == NULL: one launch, a workgroup per row.
 * replaces: tg.nn.global_max_pool, Model/models.py:262-263 (K13) incl. its scatter_max arg. */
int bg_segmax_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                  const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                  float* partial_val, int32_t* partial_arg,
                  float* out, int64_t ldo, int32_t* arg, int64_t n_rows, int64_t H, bgStream stream);

/* n_batch maxima over the SAME segments in one launch (bg_segmax_f32's one-launch form, chunk_ptr == NULL): batch b
 * reduces the rows x + b * x_step into out + b * out_step and arg + b * arg_step.  Steps are in elements and multiples
 * of 4; H % 4 == 0, 16-byte aligned operands (BG_ERR_ALIGN otherwise); n_batch <= 65535.
 * replaces: the global_max_pool calls of D(real), D(fake) and D(sample) in one critic update, Model/models.py:262-263
 * reached three times from train.py:88-96.                                                                          */
int bg_segmax_batch_f32(const float* x, int64_t ldx, int64_t x_step, const int32_t* rowptr, const int32_t* col,
                        float* out, int64_t ldo, int64_t out_step, int32_t* arg, int64_t arg_step,
                        int64_t n_rows, int64_t H, int64_t n_batch, bgStream stream);

/* bg_segmax_f32's two-phase form for callers that cannot allocate (code under hipGraph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the chunk partials live in a buffer of BG_SEGMAX_STATIC_ELEMS values +
 * as many arg entries (8 MiB in all) that the library holds in every GPU's memory until the process ends, so the
 * call takes no workspace and allocates nothing.  NO COUNTERPART IN THE REFERENCE (its pooling allocates inside
 * torch_scatter on every call).  n_chunks_ub * H > BG_SEGMAX_STATIC_ELEMS returns BG_ERR_SCRATCH; chunk_ptr is
 * required (the one-launch form of bg_segmax_f32 needs no workspace to begin with).  Every call on a GPU shares
 * that GPU's buffer: calls that overlap on one GPU (two streams, two threads) race; calls on one stream do not.
 * ONE CALL IN FLIGHT PER GPU.  The buffer is the one of the stream's device, which must be the current device
 * (BG_ERR_ARG otherwise, before anything is launched).  The first call on a device looks the buffer's address up
 * (hipGetSymbolAddress): make it outside stream capture. */
#define BG_SEGMAX_STATIC_ELEMS (1 << 20)
int bg_segmax_static_f32(const float* x, int64_t ldx, const int32_t* rowptr, const int32_t* col,
                         const int32_t* chunk_ptr, int32_t rows_per_chunk, int64_t n_chunks_ub,
                         float* out, int64_t ldo, int32_t* arg, int64_t n_rows, int64_t H, bgStream stream);

/* out[s,h] = arg[s,h] >= 0 ? x[arg[s,h], h] : 0   (gather along the saved arg-max)
 * replaces: the backward of torch_scatter.scatter_max used by global_max_pool's double backward. */
int bg_gather_arg_f32(const float* x, int64_t ldx, const int32_t* arg, float* out, int64_t ldo,
                      int64_t n_seg, int64_t H, bgStream stream);

/* out (n_rows x H, zero-filled here) ; out[arg[s,h], h] = g[s,h] where arg >= 0.
 * seg_of (may be NULL): int32 [n_rows], the segment every row belongs to (-1: none; every arg[s,h] must then be
 * a row of segment s): the same result in ONE launch, each row looking up its own value -- no fill, no scatter.
 * replaces: torch_scatter.scatter_max backward (gradient to the arg element only).          */
int bg_scatter_arg_f32(const float* g, int64_t ldg, const int32_t* arg, const int32_t* seg_of, float* out, int64_t ldo,
                       int64_t n_seg, int64_t n_rows, int64_t H, bgStream stream);

/* n_batch scatters of bg_scatter_arg_f32's row-lookup form (seg_of required) that share the segments, in one launch:
 * batch b reads g + b * g_step and arg + b * arg_step and writes out + b * out_step (steps in elements).
 * sum_out (may be NULL; pitch ld_sum): the elementwise sum of the batches' results in batch order, (v_0 + v_1) + ..,
 * written by the same launch (16-byte path, n_batch <= 8) -- what an addition launch after the scatters would give.
 * replaces: the scatter_max backward of the pooled D(real) and D(fake) in d_loss.backward(), train.py:98, and
 * autograd's sum of the two gradients arriving at the shared label-free activations.                            */
int bg_scatter_arg_batch_f32(const float* g, int64_t ldg, int64_t g_step, const int32_t* arg, int64_t arg_step,
                             const int32_t* seg_of, float* out, int64_t ldo, int64_t out_step, int64_t n_seg,
                             int64_t n_rows, int64_t H, int64_t n_batch, float* sum_out, int64_t ld_sum, bgStream stream);

/* Two bg_gather_arg_f32 of the same width H and output pitch ldo in ONE launch (results bit for bit those of the two
 * calls): out_i[s, h] = arg_i[s, h] >= 0 ? x_i[arg_i[s, h], h] : 0 for s < n_seg_i.
 * replaces: the two scatter_max backward reads of the penalty's double backward (Model/losses.py:83-88 through
 * global_max_pool by building and by story, Model/models.py:262-263).                                            */
int bg_gather_arg_pair_f32(const float* x0, int64_t ldx0, const int32_t* arg0, float* out0, int64_t n_seg0,
                           const float* x1, int64_t ldx1, const int32_t* arg1, float* out1, int64_t n_seg1,
                           int64_t ldo, int64_t H, bgStream stream);

/* Two bg_scatter_arg_f32 of the row-lookup form over the same n_rows rows, each with its own segments (seg_of_i, arg_i,
 * g_i, n_seg_i), in ONE launch; same results bit for bit.  Shapes the one-launch form does not take (a NULL seg_of,
 * H % 4 != 0, rows not 16-byte aligned) are served by two calls of bg_scatter_arg_f32.
 * replaces: the scatter_max backward of both global_max_pool levels at the penalty's sample point
 * (Model/losses.py:83-84: d D(sample) / d voxel features, closed form).                                          */
int bg_scatter_arg_pair_f32(const float* g0, int64_t ldg0, const int32_t* arg0, const int32_t* seg_of0, float* out0, int64_t n_seg0,
                            const float* g1, int64_t ldg1, const int32_t* arg1, const int32_t* seg_of1, float* out1, int64_t n_seg1,
                            int64_t ldo, int64_t n_rows, int64_t H, bgStream stream);

/* get_program_ratio, values only (no gradient), in two launches:
 *   program_weight[p] = sum over the cross edges e of program node p (prog_rowptr / prog_edge: CSR by program node over
 *                       edge ids) of att[e] * (mask[v] * area[v * ld_area]), v = voxel_of_edge[e]
 *   class_weight[s]   = sum of program_weight over the program nodes of (graph, class) slot s (slot_rowptr / slot_col)
 *   far[b]            = sum of class_weight over the slots of graph b (graph_rowptr / graph_col)
 *   ratio[s]          = class_weight[s] / (far[graph_of_slot[s]] + 1e-16)
 * att [Ec], mask [Nv], area = a column of the voxel features (pitch ld_area), ratio [n_slot], program_weight [Np],
 * far [B].  The per-edge products are those of the op-by-op form; a program node's sum is taken by 256 threads in a
 * fixed tree (the op-by-op form adds in slot order), so the results agree to rounding, run to run identical.
 * BG_ERR_SHAPE when n_slot + B > 12000 (the caller then takes the op-by-op form).
 * replaces: util_graph.py:21-41 where no gradient is asked for (util_eval.py:77,97; Model/losses.py:65 with zero weights). */
int bg_program_ratio_f32(const float* att, const float* mask, const float* area, int64_t ld_area,
                         const int32_t* voxel_of_edge, const int32_t* prog_rowptr, const int32_t* prog_edge,
                         const int32_t* slot_rowptr, const int32_t* slot_col, const int32_t* graph_rowptr,
                         const int32_t* graph_col, const int32_t* graph_of_slot, float* ratio, float* program_weight,
                         float* far, int64_t Np, int64_t n_slot, int64_t B, bgStream stream);

/* out[k] = dot(a[r_k,:], b[c_k,:]) for every CSR slot k (r_k = the row owning slot k),
 * written to out[eperm ? eperm[k] : k].   The weight-gradient of bg_spmm_f32.                */
int bg_sddmm_f32(const float* a, int64_t lda, const float* b, int64_t ldb,
                 const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                 float* out, int64_t n_rows, int64_t H, bgStream stream);

/* ------------------------------------------------------------------ cross attention ----- */

/* att[k] = sum_h theta[h] * tanh(a[v,h] + b[c_k,h]) for the cross edges k of voxel v (CSR by
 * voxel).  H % 64 == 0, H <= 512.
 * replaces: Model/models.py:113-116 (K9) with NN_v / NN_p applied per node beforehand.      */
int bg_cross_logits_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* theta,
                        const int32_t* rowptr, const int32_t* col, float* att,
                        int64_t n_vox, int64_t H, bgStream stream);

/* Backward of bg_cross_logits_f32 given g[k]: dpre[k,h] = g[k]*theta[h]*(1-t^2) (written, [nnz,H]),
 * da[v,:] = sum over v's edges of dpre, dtheta_partial[block,h] = sum g[k]*t[k,h] over the
 * block's voxels (n_blocks rows; add them up with bg_segreduce_sum_f32 / a column sum).      */
int bg_cross_logits_bwd_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* theta,
                            const int32_t* rowptr, const int32_t* col, const float* g,
                            float* dpre, float* da, int64_t ldda, float* dtheta_partial, int32_t n_blocks,
                            int64_t n_vox, int64_t H, bgStream stream);

/* Segment Gumbel-softmax pointer over CSR segments (one segment per voxel, slots contiguous):
 * y = (att - log(expn)) / tau; soft = exp(y - max y) / (sum + 1e-16); arg[v] = slot of the first
 * maximum of soft; hard = (onehot - soft) + soft.
 * replaces: util.py:13-28 gumbel_softmax (K10), called at Model/models.py:118.               */
int bg_seg_gumbel_softmax_f32(const float* att, const float* expn, const int32_t* rowptr, float tau,
                              float* soft, float* hard, int32_t* arg, int64_t n_seg, bgStream stream);

/* d_att[k] = soft[k] * (g[k] - sum_seg soft*g) / tau    (backward of the soft output)        */
int bg_seg_softmax_bwd_f32(const float* soft, const float* g, const int32_t* rowptr, float tau,
                           float* d_att, int64_t n_seg, bgStream stream);

/* arg[v] = slot of the first maximum of logit over v's slots (-1 if none).
 * replaces: util_graph.py:44-48 find_max_out_program_index's scatter_max (K12).              */
int bg_seg_argmax_f32(const float* logit, const int32_t* rowptr, int32_t* arg, int64_t n_seg,
                      bgStream stream);

/* Two-way Gumbel-softmax of the voxel mask, column 0 of the soft and of the straight-through hard
 * output: z = logits - log(expn) (logits [n,2] with row pitch ldl, Exp(1) noise [n,2]), soft = softmax(z),
 * hard = (onehot(first argmax of soft) - soft) + soft.
 * replaces: Model/models.py:109-111 (torch.nn.functional.gumbel_softmax + util.py:6-10
 * softmax_to_hard on the mask decoder's logits).                                              */
int bg_mask_gumbel_f32(const float* logits, int64_t ldl, const float* expn, float* soft0, float* hard0, int64_t n,
                       bgStream stream);

/* d_logits of <g_soft + g_more, soft0> + <g_hard, hard0>; any gradient may be NULL.  d_logits is [n,2]
 * (ldd == 2) or a 16-byte aligned 4-column slot (d, -d, 0, 0) of a wider row (ldd % 4 == 0).        */
int bg_mask_gumbel_bwd_f32(const float* soft0, const float* g_soft, const float* g_hard, const float* g_more,
                           float* d_logits, int64_t ldd, int64_t n, bgStream stream);

/* One whole Attention call, forward only (no autograd graph: inference, critic-only iterations): the
 * two-way mask, the tanh cross logits, the segment Gumbel-softmax pointer, the soft-weighted gather of
 * program features and  out = v + soft_mask * gathered  in one launch (one wave per voxel; every voxel
 * must have between 1 and 64 cross edges).  a = NN_v(v), b = NN_p(x) as for bg_cross_logits_f32.  The mask
 * logits are either given (logits [n,2], wd == NULL) or computed here as v wd^T + bd from the composed
 * two-row decoder wd [2,H] (pitch ldwd), bd [2] (logits may then be NULL).
 * replaces: Model/models.py:109-126 (Attention.forward) when nothing is differentiated.      */
int bg_attention_fwd_f32(const float* a, int64_t lda, const float* logits, int64_t ldl,
                         const float* b, int64_t ldb, const float* x, int64_t ldx,
                         const float* v, int64_t ldv, const float* theta,
                         const int32_t* rowptr, const int32_t* col,
                         const float* exp_mask, const float* exp_att, float tau,
                         float* soft0, float* hard0, float* soft, float* hard,
                         float* out, int64_t ldo, int64_t n_vox, int64_t H,
                         const float* wd, int64_t ldwd, const float* bd, bgStream stream);

/* ------------------------------------------------------------------ critic update ------ */

/* The sample at which the gradient penalty is taken: mixed = label*u + out*(1-u) (u [n]: one weight per
 * voxel), nonzero = (max_j mixed != 0), sample = ((onehot(first argmax) - mixed) + mixed) * nonzero;
 * every product / sum rounded separately (no fma), C <= 8 classes.
 * replaces: Model/losses.py:77-79 (incl. util.py:6-10 softmax_to_hard): 13 elementwise / reduce /
 * scatter launches.                                                                             */
int bg_gp_sample_f32(const float* label, int64_t ldl, const float* out, int64_t ldo, const float* u,
                     float* sample, int64_t lds, int64_t n, int64_t C, bgStream stream);

/* dst[v, p*group + j] = j < C ? src[p][v, j] : 0 for p < n_src (<= 4), j < group: several label sets side
 * by side in zero-padded column groups (the step-0 operand of bg_spmm_chain_f32).
 * replaces: torch.zeros + one strided copy per label set in front of Model/models.py:221 (label_enc). */
int bg_pack_columns_f32(const float* const* src, const int64_t* ld, int n_src, int64_t C, int64_t group,
                        float* dst, int64_t ldd, int64_t n, bgStream stream);

/* Generator output stage, forward only, one thread per voxel over its cross edges (CSR by voxel, slots in
 * edge order): idx[v] = col of the first maximum of att_hard*mask_hard; out[v,:] = mask_hard * sum att_hard *
 * cls[col,:]; soft_out likewise with att_soft.  out / soft_out are [n_vox, C] contiguous, C <= 8.
 * replaces: Model/models.py:128-141 (Attention.construct_output) + util_graph.py:44-48 when nothing is
 * differentiated.                                                                               */
int bg_construct_output_f32(const float* mask_hard, const float* att_hard, const float* att_soft,
                            const int32_t* rowptr, const int32_t* col, const float* cls, int64_t ldc, int64_t C,
                            float* out, float* soft_out, int64_t* idx, int64_t n_vox, bgStream stream);

/* out[v, 0:3] = has(v) * pos[v] - mean over v's in-neighbours j of pos[j], out[v, 3] = pad (out [n, 4]
 * contiguous; pad 0, or 1 to carry a bias through the weight's pad column; pos[v] = 3 floats at pos + v*ldp; CSR by target): the per-voxel mean of pos_i - pos_j.
 * replaces: the pos_i - pos_j columns of the message and their scatter-mean, Model/models.py:158-161
 * (constant over the layers of a pass: evaluated once).                                          */
int bg_pos_terms_f32(const float* pos, int64_t ldp, const int32_t* rowptr, const int32_t* col, float* out, int64_t n,
                     float pad, bgStream stream);

/* out[v, 0:ds] = [ feat[v, 0:nf] | onehot(level[v]) (ML) | pdm4[v, 0:4] | 1 | 0 .. ]: the label-free input
 * columns of the discriminator's collapsed voxel GNN (ds >= nf + ML + 5).
 * replaces: the inputs of feature_enc and pe[level], Model/models.py:221-222, as columns the operator
 * powers act on.                                                                                 */
int bg_d_static_cols_f32(const float* feat, int64_t ldf, int64_t nf, const int32_t* level, int64_t ML, const float* pdm4,
                         float* out, int64_t ldo, int64_t ds, int64_t n, bgStream stream);

/* The parameter-dependent rows of the collapsed discriminator's coefficient recurrence (ops.PolyCoefD), H = the
 * width of each encoder (hidden state [feature half | label half] = 2H channels), du = ds + dl input columns:
 *   x0[r, :] (how input column r enters v_0): r < nf: [fe_w[:, r]^T | 0];  nf <= r < nf+ML: [pe[r-nf] | 0];
 *             r == rb = nf+ML+4: [fe_b | le_b];  ds <= r < ds+nl: [0 | le_w[:, r-ds]^T];  else 0
 *   e[r, :] (what every layer adds): nf+ML <= r < rb: Wc[:, 4H + (r-nf-ML)]^T;  r == rb: bc;  else 0
 * x0, e: [du, 2H] contiguous; fe_w [H, nf], le_w [H, nl], pe [ML, H] contiguous; Wc [2H, >= 4H+4] pitch ldw.
 * replaces: feature_enc / label_enc / pe[level] and the position + bias terms of a VoxelBlock as linear maps
 * of the input columns, Model/models.py:221-228 (a dozen cat / transpose / copy launches).              */
int bg_d_coef_init_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b, const float* pe,
                       const float* Wc, int64_t ldw, const float* bc, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                       int64_t ds, int64_t dl, float* x0, float* e, bgStream stream);

/* Multiplication by x on a table of Chebyshev coefficients: x [K, block_floats] (block j = the coefficient of T_j,
 * contiguous), out the same shape, blocks [0, K_out) written:
 *   transpose == 0:  out_i = d_i x_{i-1} + x_{i+1} / 2  with d_1 = 1, d_i = 1/2 otherwise (x_{-1} = x_K = 0)
 *   transpose != 0:  out_0 = x_1,  out_j = (x_{j-1} + x_{j+1}) / 2          (the adjoint map)
 * block_floats % 4 == 0, 16-byte aligned, out != x.
 * no counterpart in the reference (which evaluates its 12 layers one by one, Model/models.py:227-228): this is the
 * layer recurrence of the collapsed discriminator written in the basis T_j(A) instead of A^j.                */
int bg_cheb_shift_f32(const float* x, float* out, int64_t K, int64_t K_out, int64_t block_floats, int32_t transpose,
                      bgStream stream);

/* bg_cheb_shift_f32 on a RUN of tables laid out with a period: block b of the run (b < n_blocks) is coefficient block
 * (b % period) - lead of its table; blocks outside [0, K) are padding and are written as zeros.  x must extend one block
 * past the run unless the run's last block is padding or a table's last coefficient.
 * no counterpart in the reference: the shifts of all layers' states of the coefficient recurrence by one launch.   */
int bg_cheb_shift_slabs_f32(const float* x, float* out, int64_t n_blocks, int64_t period, int64_t lead, int64_t K,
                            int64_t block_floats, int32_t transpose, bgStream stream);

/* C [M, N] = [X | shift(X)] op(B) (+ pre_add) (+ residual): a two-segment product whose second operand is the
 * bg_cheb_shift_f32 image of the first's table, formed on load (identical arithmetic, so the result equals
 * bg_cheb_shift_f32 followed by bg_gemm_f32 bit for bit).  X [>= M rows, K] (pitch ldx) is a table of n_blocks row
 * blocks of block_rows rows, M <= n_blocks * block_rows; op(B) = B^T with B [N, 2K] (transB) or B [2K, N]; C != X.
 * Skinny shapes only (<= 256 output tiles of 32 x 32, 2K <= 4096, K % 4 == 0, 16-byte operands): else BG_ERR_SHAPE /
 * BG_ERR_ALIGN and the caller runs the two launches.
 * no counterpart in the reference: one layer of the collapsed discriminator's coefficient recurrence, Chebyshev basis. */
int bg_gemm_cheb_f32(int transB, int64_t M, int64_t N, const float* X, int64_t ldx, int64_t K, int64_t block_rows,
                     int64_t n_blocks, int transpose, const float* B, int64_t ldb, float* C, int64_t ldc,
                     const float* residual, int64_t ldr, const float* pre_add, int64_t ldp, bgStream stream);

/* bg_d_coef_init_f32 with bc read at a stride (bc[c * ld_bc]): the bias as a column of a wider matrix.            */
int bg_d_coef_init_strided_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b,
                               const float* pe, const float* Wc, int64_t ldw, const float* bc, int64_t ld_bc, int64_t H,
                               int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, float* x0, float* e,
                               bgStream stream);

/* The two elementwise ends of models.VoxelBlock.compose for the discriminator's block (H2 = 2H, P = 2 H2 + 8):
 * prep:   wm_ext [H2, P] = [Wm | 0 | bm | 0 0 0],  r [H2, P] = [Wu1 | 0 .. | bu at column 2 H2 + 4 | 0 0 0], so that
 *         Wu2 wm_ext + r = [Wc | bc | 0 0 0] is ONE product with a residual;
 * finish: from gp = d / d [Wc | bc | pad] and gwm_ext = Wu2^T gp: g_Wm [H2, 2 H2 + 3], g_bm, the first half of g_Wu's
 *         rows ([H2, 2 H2]), g_bu.
 * replaces: the cat / split / add launches of the composition (Model/models.py:149-150 hold the two Linears).     */
int bg_d_compose_prep_f32(const float* Wm, int64_t ldwm, const float* bm, const float* Wu, int64_t ldwu, const float* bu,
                          int64_t H2, float* wm_ext, float* r, bgStream stream);
int bg_d_compose_bwd_finish_f32(const float* gp, const float* gwm_ext, int64_t H2, float* g_Wm, float* g_bm, float* g_Wu,
                                float* g_bu, bgStream stream);

/* Backward of bg_d_coef_init_f32: G = L+1 slabs (pitch `slab` floats) of gradient states [.., 2H]; rows [0, du)
 * of slab 0 are d/d x0, the sum over slabs 1..L of rows [0, du) is d/d e.  Writes the gradients of fe_w, fe_b,
 * le_w, le_b (contiguous), of Wc's position block (gw_tail = &gWc[0, 4H], pitch ldw) and of bc.           */
int bg_d_coef_init_bwd_strided_f32(const float* G, int64_t slab, int64_t L, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                                   int64_t ds, int64_t dl, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b,
                                   float* gw_tail, int64_t ldw, float* g_bc, int64_t ld_gbc, int32_t zero_pad,
                                   bgStream stream);   /* g_bc at a stride; zero_pad: also clear gw_tail[c, 5..7] */
int bg_d_coef_init_bwd_f32(const float* G, int64_t slab, int64_t L, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                           int64_t ds, int64_t dl, float* g_fe_w, float* g_fe_b, float* g_le_w, float* g_le_b,
                           float* gw_tail, int64_t ldw, float* g_bc, bgStream stream);

/* Head of one Attention call's backward in one launch (one wave per voxel; slots of a voxel in edge order,
 * degree <= 64, H % 64 == 0, H <= 512).  With out = v + s0 * sum_k soft_k x[col_k] and the incoming gradients
 * g_out [n_vox, H], g_soft / g_hard [Ec] (NULL = none), g_soft0 / g_hard0 [n_vox] (NULL = none):
 *   gg[v, :]  = g_out[v, :] * soft0[v]
 *   g_k       = <gg[v], x[col_k]> + g_soft_k + g_hard_k
 *   d_att_k   = soft_k (g_k - sum_j soft_j g_j) / tau
 *   d_mask[v] = (d, -d, 0, 0),  d = (g_soft0 + g_hard0 + <g_out[v], sum_k soft_k x[col_k]>) s0 (1 - s0)
 * (d_mask: 16-byte aligned 4-float slots, pitch ldd).
 * replaces: the backward of Model/models.py:109-111,120-126 (weighted gather, its two sampled dot products, the
 * segment softmax and the mask's Gumbel-softmax) that autograd runs as seven or eight launches.            */
int bg_attention_bwd_head_f32(const float* g_out, int64_t ldg, const float* x, int64_t ldx,
                              const float* soft0, const float* soft, const int32_t* rowptr, const int32_t* col,
                              const float* g_soft, const float* g_hard, const float* g_soft0, const float* g_hard0,
                              float tau, float* gg, int64_t ldgg, float* d_att, float* d_mask, int64_t ldd,
                              int64_t n_vox, int64_t H, bgStream stream);

/* WGAN-GP critic objective in two launches:
 *   loss[0] = <val_b, wb> + <val_s, ws> + coef * sum_r (||grads[r, 0:W]|| - 1)^2 ,  r < n_rows
 * (wb / ws carry the -+1/count weights of the real / fake means; grads may be NULL with n_rows = 0);
 * loss[1] = the bare penalty sum (for logging).  partial: float[128] workspace.  Deterministic (fixed
 * partition, partials added in order).
 * replaces: Model/losses.py:16-19,85-87 and the d_loss sum of train.py:98.                      */
int bg_critic_loss_f32(const float* val_b, const float* wb, int64_t nb, const float* val_s, const float* ws, int64_t ns,
                       const float* grads, int64_t ldg, int64_t n_rows, int64_t W, float coef,
                       float* partial, float* loss, bgStream stream);

/* Forward of a two-layer decoder WITHOUT activation on a batch's few pooled rows: val [R] = (X W1^T + b1) w2^T + b2
 * evaluated as X gp^T + (w2 . b1 + b2) with gp = w2 W1 [H2], which is written too (the backward below and the
 * penalty's closed form read it).  X [R, H2], w1 [Hd, H2], b1 [Hd], w2 [Hd], b2 [1].
 * replaces: building_decoder / story_decoder (Model/models.py:251-264): two Linear launches + one product as ONE.  */
int bg_decoder_fwd_f32(int64_t R, int64_t H2, int64_t Hd, const float* X, int64_t ldx, const float* w1, int64_t ldw1,
                       const float* b1, const float* w2, const float* b2, float* val, float* gp, bgStream stream);

/* Two such decoders of the same widths (H2, Hd), each on its own rows and parameters, in ONE launch: the results of two
 * bg_decoder_fwd_f32 calls bit for bit (the first decoder's workgroups, then the second's).
 * replaces: story_decoder and building_decoder of one DiscriminatorVoxel.forward (Model/models.py:262-263), which do
 * not depend on each other.                                                                                         */
int bg_decoder_fwd_pair_f32(int64_t H2, int64_t Hd, int64_t R0, const float* X0, int64_t ldx0, const float* w1_0, int64_t ldw1_0,
                            const float* b1_0, const float* w2_0, const float* b2_0, float* val0, float* gp0,
                            int64_t R1, const float* X1, int64_t ldx1, const float* w1_1, int64_t ldw1_1,
                            const float* b1_1, const float* w2_1, const float* b2_1, float* val1, float* gp1, bgStream stream);

/* Its backward, with the gradient-penalty path's share -- every gradient is rank one:
 *   u = dv^T X, sdv = sum dv, c = column sums of dgp [Rp, H2] (Rp may be 0), gp = w2 W1 (from the forward)
 *   g_w1 = w2 (x) (c + u)   g_b1 = sdv w2   g_w2 = W1 (c + u) + sdv b1   g_b2 = sdv   dX [R, H2] = dv (x) gp (+ add)
 * All outputs are assigned.  The four parameter gradients may all be NULL (a frozen decoder: dX only; w1 / w2 / b1 are
 * then not read).
 * replaces: autograd's backward of those decoders on D(real), D(fake) and the penalty's sample point: two split-K
 * products with their reductions, four column sums and four small products.                                         */
int bg_decoder_bwd_f32(int64_t R, int64_t Rp, int64_t H2, int64_t Hd, const float* dv, const float* X, int64_t ldx,
                       const float* b1, const float* dgp, int64_t ldg, const float* w1, int64_t ldw1,
                       const float* w2, const float* gp, const float* add, int64_t lda, float* g_w1, int64_t ldgw1,
                       float* g_b1, float* g_w2, float* g_b2, float* dX, int64_t ldd, bgStream stream);


/* Its gradient for the upstream scalar go[0]: d_val_b = go*wb, d_val_s = go*ws,
 * d_grads[r,:] = go * coef * 2 (||g_r|| - 1) / ||g_r|| * g_r (0 where the norm is 0); d_grads is [n_rows, W]
 * contiguous; any output may be NULL.                                                           */
int bg_critic_loss_bwd_f32(const float* go, const float* wb, int64_t nb, const float* ws, int64_t ns,
                           const float* grads, int64_t ldg, int64_t n_rows, int64_t W, float coef,
                           float* d_val_b, float* d_val_s, float* d_grads, bgStream stream);

/* ------------------------------------------------------------------ dense (fp32 MFMA) --- */

#define BG_ACT_NONE 0
#define BG_ACT_LEAKY 1

/* C[M,N] = epilogue( sum_s opA(A_s) * opB(B)[koff_s : koff_s+K_s, :] ),  koff_s = K_0+..+K_{s-1}
 *   transA == 0: A_s is [M, K_s] (pitch lda[s]);  transA == 1: A_s is [K_s, M]
 *   transB == 0: B is [K, N] (pitch ldb);         transB == 1: B is [N, K]  (nn.Linear weight)
 *   epilogue(v)[m,n] = act(row_scale[m] * (v + bias[n]) + pre_add[pr(m),n]) + residual[m,n]; each part
 *   optional; pr(m) = pre_add_rows[m] when given (a gather of per-story rows onto voxels fused into
 *   the product: the index_select at models.py:168-169), else m.
 *   acc_init (may be NULL; pitch ld_acc): the sum v starts from acc_init[m,n] instead of 0 -- a term that does
 *   not depend on the layer (the position columns times their weight block plus the bias of a block applied
 *   12 times with the same weights, models.py:196-198) is then computed once and costs no K step.
 *   pre_residual_out (may be NULL; pitch ld_pro): the value before the residual is added (the activation's
 *   result, which a backward pass needs apart from the sum) is written here as well.
 * n_seg in 1..4 fuses torch.cat((A_0,..), -1) into the product.  Exact fp32 (v_mfma_f32_32x32x2_f32).
 * Vector path needs 16-byte aligned bases and pitches / K_s % 4 == 0; anything else runs a
 * scalar-load path of the same kernel.
 * replaces: nn.Linear / torch.cat at Model/models.py:68,73,89,161,170,172,191,221,224,257-258 (K2,K15). */
int bg_gemm_f32(int transA, int transB, int64_t M, int64_t N, int n_seg,
                const float* const* A, const int64_t* lda, const int64_t* Ks,
                const float* B, int64_t ldb, float* C, int64_t ldc,
                const float* bias, const float* row_scale, int act, float slope,
                const float* residual, int64_t ldr, const float* pre_add, int64_t ldp,
                const int32_t* pre_add_rows, const float* acc_init, int64_t ld_acc,
                float* pre_residual_out, int64_t ld_pro, bgStream stream);

/* bg_gemm_f32 (K-contiguous A segments, B = nn.Linear weight W [N, K]: transA = 0, transB = 1) whose gathered
 * pre-activation addend is FORMED ON LOAD from a small matrix and a neighbour list instead of being read:
 *   addend(i) = (sum_{k = nbr_rowptr[s]}^{nbr_rowptr[s+1]-1} nbr[nbr_col[k], :]) / nbr_count[s] + self_add[s, :],  s = rows[i]
 * -- the arithmetic of bg_spmm_add_f32 (slot order starting from +0, IEEE division, then the addend), so the result
 * equals that launch followed by bg_gemm_f32(pre_add = its output, pre_add_rows = rows) bit for bit.
 * Only the row-panel kernel does this (N == 128, 2048 <= M <= 16384, every K_s % 64 == 0, 16-byte operands): any other
 * shape returns BG_ERR_SHAPE before anything is launched and the caller issues the two launches.  acc_init and
 * pre_residual_out as in bg_gemm_f32.
 * replaces: one generator VoxelBlock's cross-story mean + update Linear, Model/models.py:163-170, as ONE launch (the
 * [S, H] story term no longer travels through memory).                                                          */
int bg_gemm_nbr_f32(int64_t M, int64_t N, int n_seg, const float* const* A, const int64_t* lda, const int64_t* Ks,
                    const float* W, int64_t ldw, float* C, int64_t ldc, const float* bias, int act, float slope,
                    const float* residual, int64_t ldr,
                    const float* self_add, int64_t ld_self, const float* nbr, int64_t ld_nbr,
                    const int32_t* nbr_rowptr, const int32_t* nbr_col, const float* nbr_count, const int32_t* rows,
                    const float* acc_init, int64_t ld_acc, float* pre_residual_out, int64_t ld_pro, bgStream stream);

/* C[M,N] = A^T[M,R] * [B_0 | B_1 | ..][R,N] with the long reduction R split over `splits`
 * workgroups per tile: A is [R, M] (pitch lda), B_s is [R, Ns[s]] (pitch ldb[s]), N = sum Ns
 * (n_seg in 1..4: the concatenation of the layer inputs is never built); partial: [splits, M, N].
 * Partials are added in split order (deterministic); accumulate != 0 adds the result to C instead of
 * overwriting it (layers that share a weight sum their gradients here, in backward order, instead of in
 * separate add launches).  The weight-gradient GEMM.
 * replaces: the grad-weight product of nn.Linear's autograd for the layers listed above.     */
int bg_gemm_tn_splitk_f32(int64_t M, int64_t R, const float* A, int64_t lda,
                          int n_seg, const float* const* B, const int64_t* ldb, const int64_t* Ns,
                          float* C, int64_t ldc, float* partial, int32_t splits, int accumulate, bgStream stream);

/* out[n] (+)= sum_m x[m,n]   (bias gradient; accumulate != 0 adds to out); partial: [n_blocks, N] workspace */
int bg_colsum_f32(const float* x, int64_t ldx, int64_t M, int64_t N, float* partial, int32_t n_blocks,
                  float* out, int accumulate, bgStream stream);

/* ------------------------------------------------------------------ whole passes -------- */

/* The generator's parameters as a graph-free forward reads them (plain device pointers; all matrices row-major
 * nn.Linear layout [out, in], contiguous unless a pitch is given).  Wc / bc / Wsb are the composed block weights
 * and Wcat / bcat the attention's stacked weight (functions of the parameters alone: models.VoxelGNN_G.derived_weights). */
typedef struct {
    int64_t H, noise_dim, program_layer, voxel_layer, n_class, max_len;
    const float *p_enc_w, *p_enc_b;          /* programGNN.enc         [H, n_class + 1 + noise_dim], [H] */
    const float *p_msg_w, *p_msg_b;          /* programGNN.block.messageMLP  [H, 2H], [H]                */
    const float *p_upd_w, *p_upd_b;          /* programGNN.block.updateMLP   [H, 3H], [H]                */
    const float *v_enc_w, *v_enc_b;          /* voxelGNN.enc           [H, 4], [H]                       */
    const float *pe;                         /* voxelGNN.pe            [max_len, H]                      */
    const float *Wc; int64_t ldwc;           /* composed block weight  [H, 2H + 4], pitch ldwc           */
    const float *bc, *Wsb;                   /* [H];  [H, 2H] acting on [story mean | building mean]     */
    const float *Wcat, *bcat;                /* [H + 4, H], [H + 4]: NN_v | composed 2-row mask decoder  */
    const float *nnp_w, *nnp_b, *theta;      /* attention.NN_p [H, H], [H];  theta [H]                   */
} bgGeneratorWeights;

/* The index plan of one batch as that forward reads it (device pointers into the plan built by bg_plan_*). */
typedef struct {
    int64_t Np, Nv, S, Ec;
    int64_t cx_max_len;                                         /* most cross edges of one voxel (1..64 required)   */
    int64_t vox_all_has;                                        /* != 0: every voxel has an in-edge (required)      */
    const int32_t *pp_rowptr, *pp_col; const float *pp_scale;   /* program pair operator: 2 Np rows (multipliers)   */
    const float *prog_has;                                      /* [Np] 1 where a program node has an in-edge       */
    const int32_t *vx_rowptr, *vx_col; const float *vx_count;   /* voxel edges by target; max(in-degree, 1)         */
    const int32_t *st_rowptr, *st_col; const float *st_count;   /* voxels by story; max(size, 1)                    */
    const int32_t *story_of_voxel;                              /* [Nv]                                             */
    const int32_t *sb_rowptr, *sb_col; const float *sb_count;   /* stories x stories of the same building           */
    const int32_t *level;                                       /* [Nv] floor level inside the building             */
    const int32_t *cx_rowptr, *cx_col;                          /* cross edges by voxel (slots in edge order)       */
} bgGeneratorPlan;

/* sizeof(bgGeneratorWeights) (which = 0) / sizeof(bgGeneratorPlan) (which = 1) as this library was compiled, so a host
 * binding can refuse a library built from another revision of this header (a layout mismatch hands the kernels
 * shifted pointers); -1 for an unknown `which`. */
int64_t bg_abi_sizeof(int which);

/* Floats of workspace bg_generator_forward_f32 needs for these sizes (every piece 256-byte aligned inside it). */
int64_t bg_generator_forward_workspace(int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H);

/* One whole graph-free Generator.forward (inference, the critic-only iterations of training) as ONE host call:
 * the ~70 launches of the program GNN, the voxel encoder, 6 Attention calls, 11 VoxelBlock calls (the 12th block's
 * result is never read: Model/models.py:196-203,241-250) and the output stage, issued back to back on `stream` --
 * the same kernels with the same arguments as the op-by-op host path (bit-identical results), without ~1.2 ms of
 * interpreter time per pass.  program_in = [class one-hot | story level | z] [Np, n_class + 1 + noise_dim];
 * voxel_in = voxel_feature[:, {0,1,2,6}] [Nv, 4]; pdm4 = bg_pos_terms_f32's result (pad 1); cls = program class
 * one-hot [Np, n_class]; exp_mask[c] [Nv, 2] / exp_att[c] [Ec] = the Exp(1) noise of Attention call c
 * (1 + (voxel_layer - 1) / 2 calls).  Outputs as Generator.forward returns them; idx is int64.
 * Needs: every voxel with an in-edge and 1..64 cross edges, H % 64 == 0, H <= 256, Nv <= 65536, n_class <= 8
 * (else BG_ERR_SHAPE: run the op-by-op path).
 * replaces: Generator.forward, Model/models.py:241-250 (with :83-92, :185-203, :104-141) under torch.no_grad(). */
int bg_generator_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p,
                             const float* program_in, const float* voxel_in, const float* pdm4, const float* cls,
                             const float* const* exp_mask, const float* const* exp_att, int64_t n_calls,
                             float* workspace, int64_t workspace_floats,
                             float* out, float* soft_out, float* mask_hard, float* mask_soft,
                             float* att_hard, float* att_soft, int64_t* idx, bgStream stream);

/* The collapsed discriminator's coefficient recurrence (ops.PolyCoefD; a function of the parameters alone, so its
 * shapes never depend on the batch) as ONE host call each way.  du = ds + dl, N = (L+1) du, slab S = N + du rows.
 * forward: xall [(L+1) S, 2H] and efull [N, 2H] are cleared here, bg_d_coef_init_f32 writes the parameter rows, then
 *   L two-segment products X^{m+1} = X^m + [X^m | shift(X^m)] Wc[:, :4H]^T + E.  The result X^L is xall's last slab
 *   from row du on.
 * backward: g [N, 2H] (contiguous) = d/d X^L.  gall [(L+1) S, 2H], wab [4H, 2H] and partial (floats: see
 *   bg_d_coef_backward_workspace) are workspaces; writes gW [2H, 4H+4] (pitch 4H+4) and the encoders' gradients.
 * basis == BG_CHAIN_CHEB: row block j of a state is the coefficient of T_j(A) instead of A^j; shift(X) becomes the
 *   multiplication-by-x map of bg_cheb_shift_f32, formed by bg_gemm_cheb_f32 while it loads the operand (no extra
 *   launch); the backward rebuilds the shifts of all layers' states with one bg_cheb_shift_slabs_f32 launch into
 *   shall [L S + du, 2H] (workspace; NULL for BG_CHAIN_POWER) for the weight gradient.
 * replaces: the same 15 / 18 launches issued one by one from the host.                                  */
int bg_d_coef_forward_f32(const float* fe_w, const float* fe_b, const float* le_w, const float* le_b, const float* pe,
                          const float* Wc, int64_t ldw, const float* bc, int64_t H, int64_t nf, int64_t nl, int64_t ML,
                          int64_t ds, int64_t dl, int64_t L, int32_t basis, float* xall, float* efull, bgStream stream);
int64_t bg_d_coef_backward_workspace(int64_t H, int64_t ds, int64_t dl, int64_t L);   /* floats of `partial` */
int bg_d_coef_backward_f32(const float* g, const float* xall, const float* Wc, int64_t ldw, int64_t H,
                           int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L, int32_t basis,
                           float* shall, float* gall, float* wab,
                           float* partial, float* gW, float* g_bc, float* g_fe_w, float* g_fe_b, float* g_le_w,
                           float* g_le_b, bgStream stream);

/* bg_d_coef_* with the block's composition (models.VoxelBlock.compose: Wc = [Wu1 + Wu2 Wm1 | Wu2 Wm2 | Wu2 Wm3],
 * bc = Wu2 bm + bu) inside: from messageMLP (Wm [2H, 4H+3], bm) and updateMLP (Wu [2H, 4H], bu) to the coefficient table and
 * back to their gradients.  P = 4H + 8; wm_ext, r, wcb (forward; wm_ext and wcb = [Wc | bc | pad] are read again by the
 * backward), gp, gwm_ext (backward) are [2H, P] buffers of the caller.  Forward: 2 launches + bg_d_coef_forward_f32's;
 * backward: bg_d_coef_backward_f32's + 3.  g_Wm [2H, 4H+3], g_bm [2H], g_Wu [2H, 4H], g_bu [2H] are written completely.
 * replaces: the eleven small stock launches (cat / add / split and their backward) around the recurrence.        */
int bg_d_table_forward_f32(const float* Wm, int64_t ldwm, const float* bm, const float* Wu, int64_t ldwu, const float* bu,
                           const float* fe_w, const float* fe_b, const float* le_w, const float* le_b, const float* pe,
                           int64_t H, int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L, int32_t basis,
                           float* wm_ext, float* r, float* wcb, float* xall, float* efull, bgStream stream);
int bg_d_table_backward_f32(const float* g, const float* xall, const float* wcb, const float* wm_ext, const float* Wu,
                            int64_t ldwu, int64_t H, int64_t nf, int64_t nl, int64_t ML, int64_t ds, int64_t dl, int64_t L,
                            int32_t basis, float* shall, float* gall, float* wab, float* partial, float* gp, float* gwm_ext,
                            float* g_Wm, float* g_bm, float* g_Wu, float* g_bu, float* g_fe_w, float* g_fe_b, float* g_le_w,
                            float* g_le_b, bgStream stream);

/* ---- the generator's voxel GNN with a backward: forward that keeps what the backward reads, and the backward ----
 * Plan pieces only the backward reads (besides bgGeneratorPlan).                                                  */
typedef struct {
    const int32_t *vt_rowptr, *vt_col;               /* voxel edges by source: A^T = P^T diag(1 / vx_count)          */
    const int32_t *cp_rowptr, *cp_col, *cp_perm;     /* cross edges by program node (slots -> voxel, slot -> edge id) */
    int64_t cp_one_launch;                           /* != 0: those rows are long: one workgroup per row
                                                        (bg_segreduce_sum_f32 without a chunk table), else bg_spmm_f32 */
} bgGeneratorBackPlan;

/* Floats of workspace shared by bg_gstack_forward_f32 and bg_gstack_backward_f32 of one pass (the forward leaves
 * every call's input, aggregate and activation in it).                                                          */
int64_t bg_gstack_workspace(int64_t Np, int64_t Nv, int64_t S, int64_t Ec, int64_t H, int64_t L);

/* VoxelGNN_G.forward (Model/models.py:185-203) from the program features x [Np, H] on -- encoder + pe[level],
 * NN_p(x), then Attention / VoxelBlock calls in the reference's order -- as ONE host call that keeps every call's
 * input, neighbour mean, activation, NN_v product, soft mask and soft pointer in the workspace.  The same launches
 * as ops.GBlock / ops.AttentionNode issue one by one (bit-identical).  Outputs: the last Attention call's mask and
 * pointer.  Uses w's v_enc_*, pe, Wc / bc / Wsb, Wcat / bcat, nnp_*, theta (the program GNN's fields are not read).
 * Limits as bg_generator_forward_f32.                                                                            */
int bg_gstack_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const float* x, const float* voxel_in,
                          const float* pdm4, const float* const* exp_mask, const float* const* exp_att, int64_t n_calls,
                          float* workspace, int64_t workspace_floats,
                          float* mask_hard, float* mask_soft, float* att_hard, float* att_soft, bgStream stream);

/* Its backward for the gradients arriving at the four outputs (any may be NULL): every weight gradient summed over
 * the calls inside the weight-gradient launches (accumulate), in backward order.  Writes g_x [Np, H], g_enc_w [H, 4],
 * g_enc_b [H], g_nnp_w [H, H], g_nnp_b [H], g_Wcat [H + 4, H], g_bcat [H + 4], g_theta [H], g_Wc [H, 2H + 4]
 * (contiguous; its last column is the gradient of bc: pdm4's pad column is 1), g_Wsb [H, 2H].
 * replaces: g_loss.backward() below the program GNN, train.py:112 (autograd over Model/models.py:185-203).        */
int bg_gstack_backward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const bgGeneratorBackPlan* bp,
                           const float* x, const float* voxel_in, const float* pdm4, int64_t n_calls,
                           float* workspace, int64_t workspace_floats,
                           const float* mask_soft, const float* att_soft,
                           const float* g_mask_hard, const float* g_mask_soft, const float* g_att_hard, const float* g_att_soft,
                           float* g_x, float* g_enc_w, float* g_enc_b, float* g_nnp_w, float* g_nnp_b, float* g_Wcat,
                           float* g_bcat, float* g_theta, float* g_Wc, float* g_Wsb, bgStream stream);

/* out = g * (y > 0 ? 1 : slope) over n floats (y = the LeakyReLU's result; n % 4 == 0, 16-byte aligned), and
 * out[r, :] = scale[r] * x[r, :] (H, pitches % 4 == 0): the two elementwise steps of a ProgramBlock's backward.
 * replaces: aten leaky_relu_backward and the row mask's product in the backward of Model/models.py:68,73.       */
int bg_leaky_bwd_f32(const float* g, const float* y, float slope, float* out, int64_t n, bgStream stream);
int bg_rowscale_f32(const float* x, int64_t ldx, const float* scale, float* out, int64_t ldo, int64_t n, int64_t H,
                    bgStream stream);

/* ProgramGNN.forward (Model/models.py:83-92 with :52-76) keeping every layer's input, aggregates and activation,
 * and its backward, as ONE host call each.  program_in = [class one-hot | story level | z] [Np, kin],
 * kin = n_class + 1 + noise_dim; uses w's p_* fields and p's pp_* / prog_has; vt_* of the back plan are not read
 * here: pr_rowptr / pr_col = program edges by source (the transpose of the in-edge mean, column scale pf_count).
 * forward: x_out [Np, H].  backward: g [Np, H] = d / d x_out; writes g_enc_w [H, kin], g_enc_b, g_msg_w [H, 2H],
 * g_msg_b, g_upd_w [H, 3H], g_upd_b (weight gradients summed over the layers inside the launches).
 * replaces: the program GNN's part of Generator.forward and of g_loss.backward(), train.py:84,112.               */
int64_t bg_pstack_workspace(int64_t Np, int64_t H, int64_t kin, int64_t L);
int bg_pstack_forward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const float* program_in,
                          float* workspace, int64_t workspace_floats, float* x_out, bgStream stream);
int bg_pstack_backward_f32(const bgGeneratorWeights* w, const bgGeneratorPlan* p, const int32_t* pr_rowptr,
                           const int32_t* pr_col, const float* pf_count, const float* program_in, const float* g,
                           float* workspace, int64_t workspace_floats, float* g_enc_w, float* g_enc_b, float* g_msg_w,
                           float* g_msg_b, float* g_upd_w, float* g_upd_b, bgStream stream);

/* src [n, K, P, W] -> dst [P, n, K, W] (to_outer != 0) or src [P, n, K, W] -> dst [n, K, P, W] (to_outer == 0):
 * column groups that several label sets / both critic outputs share in one operator chain, regrouped into one
 * contiguous matrix per set for the products on either side.  W % 4 == 0, 16-byte aligned, contiguous.
 * replaces: view + permute + contiguous around the collapsed discriminator's chains (no reference counterpart: the
 * reference runs one pass per label set, train.py:88-96).                                                  */
int bg_regroup_f32(const float* src, float* dst, int64_t n, int64_t K, int64_t P, int64_t W, int to_outer, bgStream stream);

/* p[0..n) = value (n % 4 == 0, 16-byte aligned): a fill as a kernel launch (the critic head and the generator
 * stack issue kernels only, no runtime memset / memcpy nodes). */
int bg_fill_f32(float* p, int64_t n, float value, bgStream stream);

/* out[i] = i mod `mod` (int32): the row index that lets ONE product over several row-stacked label sets read the
 * same [mod, N] addend (bg_gemm_f32's pre_add_rows).  No reference counterpart (see bg_regroup_f32).            */
int bg_iota_mod_i32(int32_t* out, int64_t n, int64_t mod, bgStream stream);

/* bg_regroup_f32 (same arguments) and bg_iota_mod_i32 (idx[t] = t mod idx_mod for t < idx_n) in ONE launch: two independent
 * writes that the critic head needs before its row-stacked product (the label powers per label set; the row of the
 * label-free part that row r of the 3 N stacked rows adds).  Same values as the two calls.
 * replaces: as bg_regroup_f32 and bg_iota_mod_i32.                                                               */
int bg_regroup_iota_f32(const float* src, float* dst, int64_t n, int64_t K, int64_t P, int64_t W, int to_outer,
                        int32_t* idx, int64_t idx_n, int64_t idx_mod, bgStream stream);

/* out = a + b over n floats (n % 4 == 0, 16-byte aligned): the one elementwise sum the critic's backward needs
 * (the gradients of D(real) and D(fake) with respect to the shared label-free part).
 * replaces: autograd's gradient accumulation for a tensor used twice.                                      */
int bg_add_f32(const float* a, const float* b, float* out, int64_t n, bgStream stream);
/* dst [reps, n] = src [n] repeated (n % 4 == 0, 16-byte aligned).
 * no counterpart in the reference: the position columns stacked once per generator-block call, so that the weight gradient
 * of all calls of the shared block (Model/models.py:198-200) is ONE product over the stacked rows.             */
int bg_tile_f32(const float* src, float* dst, int64_t n, int64_t reps, bgStream stream);

/* Everything one WGAN-GP critic update evaluates AFTER the collapsed GNN's coefficient table W and the operator powers
 * exist -- D(real), D(fake), the closed-form penalty gradients at the sample point, the objective -- and its
 * backward, as ONE host call each (models.DiscriminatorVoxel.critic_terms + ops.CriticLoss issued from C++).
 * Sizes: N voxels, S stories, B graphs, L layers, ds / dl static / label columns per power (multiples of 4),
 * H2 = 2 H the GNN width, Hd the decoders' hidden width.  Decoders are two Linears without activation.          */
typedef struct {
    int64_t N, S, B, L, ds, dl, H2, Hd;
    int64_t blk_n, blk_max_rows, blk_max_nnz, blk_max_deg;      /* diagonal blocks of the voxel operator (bg_spmm_chain_f32) */
    int64_t basis;                                              /* BG_CHAIN_POWER: Zs / Zk / W in powers A^k; BG_CHAIN_CHEB: in T_k(A) */
    const float *Zs;                                            /* [N, (L+1) ds] label-free operator powers                  */
    const float *Zk;                                            /* [N, L+1, 3, dl] label powers of (real, fake, sample)      */
    const float *W;                                             /* [(L+1) ds + (L+1) dl, H2] coefficient table (rows: static | label) */
    const float *b_w1, *b_b1, *b_w2, *b_b2;                     /* building_dec: [Hd, H2], [Hd], [1, Hd], [1]                */
    const float *s_w1, *s_b1, *s_w2, *s_b2;                     /* story_dec                                                 */
    const float *wb, *ws;                                       /* [2 B], [2 S]: -+1/count weights, real rows first          */
    const int32_t *st_rowptr, *st_col, *story_of_voxel;         /* voxels by story (every story non-empty, <= 4096 voxels)   */
    const int32_t *sg_rowptr, *sg_col, *graph_of_story;         /* stories by graph (<= 256 stories per graph)               */
    const int32_t *graph_of_voxel;                              /* [N]                                                       */
    const int32_t *a_rowptr, *a_col; const float *a_count;      /* voxel edges by target + max(in-degree, 1): the operator A */
    const int32_t *at_rowptr, *at_col;                          /* voxel edges by source: A^T = P^T diag(1 / count)          */
    const int32_t *blk_ptr;                                     /* [blk_n + 1]                                               */
} bgCriticHead;

/* Floats of workspace (shared by the forward and the backward call of one update; the forward leaves what the
 * backward reads in it) for these sizes; -1 for sizes the calls do not take.                                   */
int64_t bg_critic_head_workspace(const bgCriticHead* h);

/* loss[0] = <D(real), D(fake) values, wb / ws> + coef * sum over voxels and both outputs of (||grad|| - 1)^2,
 * loss[1] = the bare penalty sum.  ~35 launches; BG_ERR_SHAPE for plans outside the limits above (the caller
 * then runs the op-by-op path).
 * replaces: Model/losses.py:14-19,73-89 with train.py:88-98 (three discriminator passes + two autograd.grad calls). */
int bg_critic_head_forward_f32(const bgCriticHead* h, float coef, float* workspace, int64_t workspace_floats,
                               float* loss, bgStream stream);

/* Gradients of go[0] * loss[0] with respect to W and the decoders' parameters (each written, not accumulated):
 * gW [(L+1)(ds + dl), H2], g_b_w1 [Hd, H2], g_b_b1 [Hd], g_b_w2 [Hd], g_b_b2 [1], the same for story_dec.  ~50 launches.
 * replaces: d_loss.backward() of train.py:98 below the coefficient table.                                       */
int bg_critic_head_backward_f32(const bgCriticHead* h, float coef, const float* go, float* workspace,
                                int64_t workspace_floats, float* gW, float* g_b_w1, float* g_b_b1, float* g_b_w2,
                                float* g_b_b2, float* g_s_w1, float* g_s_b1, float* g_s_w2, float* g_s_b2, bgStream stream);

/* The generator's adversarial objective through the (frozen) collapsed discriminator as ONE host call each way:
 * label [N, n_lab] (pitch ldl; the generated labels, differentiable) -> its operator powers -> v = [Zs | Zl] W -> story /
 * graph max pooling -> decoders -> loss[0] = <D_building, wb[0..B)> + <D_story, ws[0..S)> (loss: 2 floats, the second is 0); the backward writes
 * g_label [N, n_lab] (contiguous) = go[0] * d loss / d label.  h as for bg_critic_head_* (Zk is not read; wb / ws: the
 * first B / S entries are used).  Workspace shared by both calls: bg_gloss_head_workspace(h) floats.
 * replaces: discriminator(batch, out) + VolumetricDesignLoss_G's adversarial term and their backward down to the
 * generated labels, train.py:108-112 with Model/losses.py:52-61.                                                  */
int64_t bg_gloss_head_workspace(const bgCriticHead* h);
int bg_gloss_head_forward_f32(const bgCriticHead* h, const float* label, int64_t ldl, int64_t n_lab, float* workspace,
                              int64_t workspace_floats, float* loss, bgStream stream);
int bg_gloss_head_backward_f32(const bgCriticHead* h, int64_t n_lab, const float* go, float* workspace,
                               int64_t workspace_floats, float* g_label, bgStream stream);

/* ------------------------------------------------------------------ optimiser ----------- */

/* check_connectivity: which program-graph edges a generated design realises.
 *   code [Np * Np] int32 (workspace, cleared here): bit 0 of code[src * Np + dst] = some voxel edge (a, b) with
 *   mask[a] + mask[b] == 2 has prog_of_voxel[a] = src, prog_of_voxel[b] = dst; bit 1 = (src, dst) is a program edge.
 *   inter / missing / gen: int64 [2, cap_*] (row 0 = src, row 1 = dst) -- the pairs with code 3 / 2 / 1 in ascending
 *   (src, dst) order, i.e. what torch.unique(dim=1) returns; counts [3] int64 = how many of each (entries beyond a
 *   capacity are counted, not written).  voxel_edge [2, Ev], program_edge [2, Ep], prog_of_voxel [Nv]: int64 as the
 *   batch holds them.  Three launches, no host synchronisation; Np <= 8192.
 * replaces: util_eval.py:102-136 (three torch.unique(dim=1) sorts and their set arithmetic).                       */
int bg_connectivity(const int64_t* voxel_edge, int64_t Ev, const float* mask, const int64_t* prog_of_voxel, int64_t Nv,
                    const int64_t* program_edge, int64_t Ep, int64_t Np, int32_t* code, int64_t* inter, int64_t cap_inter,
                    int64_t* missing, int64_t cap_missing, int64_t* gen, int64_t cap_gen, int64_t* counts, bgStream stream);

/* Zero fill and strided device-to-device copy as KERNELS (sizes and pitches in bytes, multiples of 4; 4-byte aligned
 * pointers; `width` bytes of each of `rows` rows).  NO COUNTERPART IN THE REFERENCE: they stand where the composite
 * entry points would otherwise call hipMemsetAsync / hipMemcpy2DAsync, which turn into memset / memcpy graph nodes
 * under stream capture -- on ROCm 7.2 / gfx950 a replayed graph was seen to run such a node out of order with its
 * neighbouring kernel nodes (csrc/common.h).  Callers that capture this library's calls should use them likewise. */
int bg_zero_bytes(void* p, int64_t bytes, bgStream stream);
int bg_copy2d_bytes(void* dst, int64_t dpitch, const void* src, int64_t spitch, int64_t width, int64_t rows,
                    bgStream stream);

/* Up to 8 of the two calls above as ONE launch: job i zero-fills (src[i] == NULL; dpitch / width / rows as a copy's
 * destination) or copies rows[i] rows of width[i] bytes.  Jobs must not overlap.
 * NO COUNTERPART IN THE REFERENCE: the preparation steps of bg_d_coef_* / bg_d_table_* (clear two workspaces; clear one,
 * copy the incoming gradient and two weight blocks) as one launch each instead of two and four.                      */
int bg_fill_copy_batch(int32_t n_jobs, void* const* dst, const int64_t* dpitch, const void* const* src,
                       const int64_t* spitch, const int64_t* width, const int64_t* rows, bgStream stream);

/* One Adam step on a flat fp32 buffer (torch.optim.Adam semantics, no weight decay/amsgrad):
 * m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * t = step, or *step_dev (a device int32) when step_dev is not NULL, so that a captured hipGraph
 * replays with the current step count.
 * replaces: optimizer_G.step() / optimizer_D.step(), train.py:101,114.                       */
int bg_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                float lr, float beta1, float beta2, float eps, int32_t step, const int32_t* step_dev,
                bgStream stream);

#ifdef __cplusplus
}
#endif
#endif /* BGAN_H */
