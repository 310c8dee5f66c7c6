/* libddmp_hip -- C ABI of the MI355X (gfx950) dual-GCN mesh-denoising hot path.
 *
 * Drop-in boundary: the reference (astaka-pe/Dual-DMP) has no FFI layer;
 * its hot path calls the Python operator API of torch_geometric / torch at
 *   util/networks.py:15-26,51-62,76-87,112-123   GCNConv(in,out)(x, edge_index)
 *   util/networks.py:31-44,64-67,125-129         BatchNorm1d / LeakyReLU / Linear heads
 *   util/loss.py:16,37,55,86,140,261             the five losses + mad
 *   main.py:106-110                              backward, clip_grad_norm_, Adam.step
 * This library is what a ctypes/cffi binding for that path binds instead.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - matrices are row-major float32 with an explicit leading dimension (elements);
 *   - the library never allocates or frees caller-visible memory: outputs and workspaces are
 *     caller-owned (PyTorch tensors in the Python host side); only ddmp_graph owns device memory;
 *   - work is enqueued on the caller's hipStream_t (void* here so the header needs no HIP);
 *     nothing synchronises the stream except the *_host helpers and ddmp_graph_create;
 *   - return value: 0 = OK, < 0 = invalid argument (DDMP_E*), > 0 = hipError_t;
 *     nothing throws across the ABI.
 */
#ifndef DDMP_HIP_H
#define DDMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define DDMP_OK 0
#define DDMP_EINVAL (-1)      /* bad argument (null pointer, negative size, unsupported width) */
#define DDMP_ERANGE (-2)      /* index out of range in an input table */
#define DDMP_ENOMEM (-3)      /* host allocation failed */
#define DDMP_EWORKSPACE (-4)  /* caller workspace too small */

#define DDMP_ABI_VERSION 3      /* 3 (round 5): per-call options (ddmp_opts, the *_o entry points) replace the armed state;
                                   2 (round 4): rules for the "armed for the next call" state + ddmp_next_pending / ddmp_next_cancel /
                                   ddmp_gemm_forget_planes; additions only otherwise (the *_bf16 and dtype-tagged entry points) */

typedef struct ddmp_graph ddmp_graph;
typedef void* ddmp_stream;    /* hipStream_t */

int ddmp_abi_version(void);
const char* ddmp_status_string(int status);

/* ------------------------------------------------------------------ graph (gcn_norm, once)
 * Replaces the per-call `gcn_norm` of PyG 2.2.0 GCNConv (cached=False: recomputed 24x per step
 * in the reference, util/networks.py:51-62).  edge_index is the reference's [2, nnz] int64
 * tensor (row 0 = source j, row 1 = target i; util/datamaker.py:90-92), no self loops needed:
 * explicit self loops are dropped and exactly one per node is added (add_remaining_self_loops),
 * multi-edges keep their multiplicity.  CSR row i lists the sources of i (ascending) plus i.
 * dinv[i] = (1 + in-degree(i))^-1/2.
 */
int ddmp_csr_build_host(int64_t n_nodes, int64_t nnz, const int64_t* edge_index_host,
                        int32_t* rowptr_host /*[n+1]*/, int32_t* col_host /*[nnz_out]*/,
                        float* dinv_host /*[n]*/, int64_t* nnz_out /*in: capacity, out: used*/);
/* breadth-first (Cuthill-McKee style) node order for gather locality: order[k] = old id */
int ddmp_csr_bfs_order_host(int64_t n_nodes, const int32_t* rowptr_host, const int32_t* col_host,
                            int32_t* order_host /*[n]*/);
/* recursive coordinate bisection of the node coordinates (xyz_host [n,3] float64) into leaves of exactly `leaf` nodes
 * (the last one may be shorter), every split along the longest axis of the subset's bounding box at a multiple of
 * `leaf`: order[k] = old id.  Consecutive ids form compact patches of the surface whose sizes are multiples of the
 * gather kernels' 64-row chunks (a chunk of a Morton order straddles several cells of the curve: 1.9 distinct
 * neighbour rows per output row on the vertex graph of a 1M-face mesh against 1.6 here). */
int ddmp_rcb_order_host(int64_t n_nodes, const double* xyz_host, int leaf, int32_t* order_host /*[n]*/);

int ddmp_graph_create(int64_t n_nodes, int64_t nnz, const int64_t* edge_index, int edge_index_on_device,
                      ddmp_graph** out);
/* from ready CSR tables on the host (partitioned graphs: n_rows owned rows, cols < n_cols) */
int ddmp_graph_create_csr_host(int64_t n_rows, int64_t n_cols, const int32_t* rowptr_host,
                               const int32_t* col_host, const float* dinv_host, ddmp_graph** out);
/* rows [row0, row1) of such tables as a graph of their own: output row i is node row0 + i, the columns keep the numbering of the
 * whole local graph (pass the same X, and Y + row0 * ldy).  The interior / boundary halves of a partitioned graph (SURVEY.md 8e:
 * the halo exchange of a layer travels while the rows that reference no halo row are aggregated -- dual-dmp_amd/dist.py) */
int ddmp_graph_create_csr_rows_host(int64_t n_rows_all, int64_t n_cols, const int32_t* rowptr_host, const int32_t* col_host,
                                    const float* dinv_host, int64_t row0, int64_t row1, ddmp_graph** out);
int ddmp_graph_destroy(ddmp_graph* g);
int ddmp_graph_info(const ddmp_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int* max_row_nnz);
/* How the LDS-patch gather takes this graph's 64-row chunks: patch_kd = patch rows per LDS buffer / 32 (0: no patch tables,
 * every aggregation runs the lean gather), n_heavy = chunks that go to the lean gather in a second launch, n_split = the extra passes of the chunks walked
 * as 2 halves / 4 quarters inside the same launch (= their extra record slots).  Diagnostic (tests, DESIGN figures); any pointer may be NULL. */
int ddmp_graph_patch_info(const ddmp_graph* g, int* patch_kd, int* n_heavy, int* n_split);

/* ------------------------------------------------------------------ aggregation  Y = A_hat . f(X) (+ bias)
 * Replaces GCNConv.propagate (index_select * norm -> scatter_add) and the `+ bias`.
 * A_hat = D^-1/2 (A + I) D^-1/2 is symmetric for the reference's graphs, so the same call is
 * the backward aggregation.  Optional fused prologue f(x) = LeakyReLU(pro_scale[c]*x + pro_shift[c])
 * (the BatchNorm1d + LeakyReLU of the previous layer, util/networks.py:51-62) applied to every
 * gathered element; pass NULL/NULL for f = identity.  C must be > 0; the vector path needs
 * C % 4 == 0 and ldx, ldy % 4 == 0, other widths take a scalar path.
 */
int ddmp_spmm_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                  const float* bias, const float* pro_scale, const float* pro_shift, float slope,
                  ddmp_stream stream);
/* 1: ddmp_spmm* runs its LDS-patch form for this graph and shape (graphs from 64k rows with <= 12 entries per row on average,
 * C >= 256; rows of any length -- the selection is per 64-row chunk since round 5, chunks the form cannot take go to the lean
 * gather; has_red: 0 | 1 the fused BatchNorm-backward reduction | 2 the fused statistics; dtype DDMP_BF16: plain, prologue and
 * statistics only): distinct rows of a chunk copied to LDS once, gathers from LDS -- csrc/spmm_patch.hip; same results either way */
int ddmp_spmm_patch_selected(const ddmp_graph* g, int C, int dtype, int has_pro, int has_red);

/* ------------------------------------------------------------------ dense steps (MFMA; float32 operands and results, the
 * arithmetic is ddmp_set_gemm_mode's: by default SPLIT-precision 16-bit MFMA products with f32 accumulation -- f32-class
 * accuracy, not bit-exact f32; mode 0 = f32-input MFMA, the strict one)
 * Replace GCNConv.lin (X.W^T, no bias) and its autograd (dX = dH.W, dW = dH^T.X).
 *   nt : Y[n,M] = f(A[n,K]) . W[M,K]^T (+ bias[M])       forward / dgrad with a transposed table
 *   nn : Y[n,K] =   A[n,M]  . W[M,K]                     dgrad
 *   tn : dW[M,K] = G[n,M]^T . f(Z[n,K])                  wgrad, split over rows; needs workspace
 * f as in ddmp_spmm_f32 (per column of A resp. Z).  K, lda, ldz must be multiples of 4.
 */
size_t ddmp_gemm_rows_workspace_bytes(int K, int M);   /* optional scratch of nt / nn: W pre-split into bf16 planes */
int ddmp_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy,
                     int64_t n_rows, int K, int M, const float* bias,
                     const float* pro_scale, const float* pro_shift, float slope,
                     void* workspace /*nullable*/, size_t workspace_bytes, ddmp_stream stream);
size_t ddmp_gemm_tn_workspace_bytes(int64_t n_rows, int M, int K);
/* GEMM arithmetic (process-wide; the environment variable DDMP_GEMM_MODE sets the initial value):
 *   6  = bf16x6 split MFMA: every f32 operand is split into three bf16 terms, six bf16 MFMA products accumulated in
 *        f32 -- f32-class accuracy at 2.67x the f32-MFMA rate;
 *   13 = (default) f16x3 in the row-panel kernels (the wide layers of big meshes), bf16x6 everywhere else: the
 *        operand, scaled by a power of two, is split into two _Float16 terms (22-23 bits) and three f16 MFMA products
 *        are accumulated in f32 (the "3xTF32" scheme; csrc/gemm_f16s.inc) -- f32-class accuracy, half the MFMA work;
 *   3  = bf16x3 (three products, ~2^-16 relative);   0 = f32-input MFMA (v_mfma_f32_32x32x2_f32).
 * Mode 13's power of two comes from the operand's absolute maximum, kept in a "scale slot" (device float[4]: {maximum
 * in use, maximum seen by the last kernels, overflow flag, healed events}).  By default the library measures it in a pre-pass over
 * the operand.  A training loop avoids that pass: DDMP_OPT_SCALES of the *_o forms names persistent slots for THAT call
 * (slot_a: the row operand A / dZ / G; slot_b: Z of the tn forms; prime != 0: measure now
 * anyway, e.g. first iteration), the GEMM kernels record the maximum they see, and ddmp_gemm_scales_roll, once per
 * iteration, makes it the next iteration's scale (6 bits of head-room).  An operand that still outgrows its scale raises
 * the slot's flag ([2]) and is HEALED inside the same ddmp_gemm_* call: the kernel is launched a second time, returns at
 * once while the flag is clear and otherwise redoes the product with the maximum the first launch has just recorded
 * ([1]), overwriting the clamped result before the caller's next kernel sees it.  ddmp_gemm_scales_roll counts healed
 * events per slot in [3] (-1: the operand was not finite) and clears the flag. */
int ddmp_set_gemm_mode(int mode);
int ddmp_get_gemm_mode(void);
int ddmp_gemm_scales_roll(float* slots, int n_slots, ddmp_stream stream);

/* ------------------------------------------------------------------ BatchNorm1d (train mode) + LeakyReLU
 * Replace nn.BatchNorm1d(C) in train mode + nn.LeakyReLU() (util/networks.py:31-44,51-62): batch
 * statistics over all n rows, biased variance, running stats with `momentum`.  Statistics are
 * float64 column sums (sum, sumsq) so that several devices can add theirs before `prepare`.
 * The normalise+activate is expressed as z = LeakyReLU(scale[c]*y + shift[c]) and is applied by the
 * CONSUMER kernels (pro_scale / pro_shift arguments); ddmp_bn_lrelu_apply_f32 materialises it.
 * Widths: C a power of two in [8, 1024].  Workspace: ddmp_colreduce_workspace_bytes(n_rows, C).
 */
size_t ddmp_colreduce_workspace_bytes(int64_t n_rows, int C);
int ddmp_bn_stats_f32(const float* Y, int64_t ldy, int64_t n_rows, int C, double* sums /*[2C]*/,
                      void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_prepare_f32(const double* sums /*[2C]*/, double n_total, int C, const float* gamma,
                        const float* beta, float eps, float momentum, float* scale, float* shift,
                        float* mean, float* rstd, float* running_mean /*nullable*/,
                        float* running_var /*nullable*/, ddmp_stream stream);
int ddmp_bn_lrelu_apply_f32(const float* Y, int64_t ldy, float* Z, int64_t ldz, int64_t n_rows, int C,
                            const float* scale, const float* shift, float slope, ddmp_stream stream);
/* backward reductions sums2 = (sum g, sum g*yhat), g = dZ * LeakyReLU'(scale*y+shift), yhat = (y-mean)*rstd: ddmp_bn_bwd_reduce
 * (dtype-tagged, below) */
/* sums2 -> dgamma, dbeta and the two folded constants of dY = scale*g + c1*y + c0 */
int ddmp_bn_bwd_prepare_f32(const double* sums2, double n_total, int C, const float* scale, const float* mean,
                            const float* rstd, float* dgamma, float* dbeta, float* c1, float* c0,
                            ddmp_stream stream);
/* Tail-fused coefficients (round 3) and the other per-call options are arguments of the *_o entry points at the end of this header
 * (ABI 3); the ABI-2 calls that armed them "for the next call of this host thread" left the ABI in round 6 (they are the hidden
 * implementation of the _o scopes: csrc/ddmp_internal.h).  Diagnostic: the bit mask of options still recorded for this host thread
 * -- BatchNorm coefficients (bit 0), GEMM scale slots (bit 1), prepared weight planes (bit 2); 0 between calls. */
int ddmp_next_pending(void);
/* dY (gradient w.r.t. the conv output) and its column sums (= gradient of the conv bias; dbias_sums NULL: dY only -- behind
 * a BatchNorm those sums are zero in exact arithmetic) */
int ddmp_colsum_f32(const float* X, int64_t ldx, int64_t n_rows, int C, double* sums /*[C]*/, void* workspace,
                    size_t workspace_bytes, ddmp_stream stream);
int ddmp_f64_to_f32(const double* in, float* out, int64_t n, ddmp_stream stream);

/* ------------------------------------------------------------------ output heads
 * Replace linear1 -> LeakyReLU -> linear2 (+ residual | tanh + row normalise), util/networks.py:64-67
 * (kind 0, PosNet: out = x_pos + u) and :125-129 (kind 1, NormalNet: out = tanh(u)/(|tanh(u)|+1e-12)).
 * Y is the conv12 output [n,32]; its BatchNorm+LeakyReLU is the scale/shift prologue.  W1 [16,32],
 * b1 [16], W2 [3,16], b2 [3]; out / dout [n,3] contiguous.  Backward writes dZ [n,32] (gradient w.r.t.
 * the activated conv12 features) and overwrites the four parameter gradients.
 */
size_t ddmp_head_bwd_workspace_bytes(int64_t n_rows);

/* ------------------------------------------------------------------ losses (util/loss.py)
 * Index tables are int32 device arrays: faces [F,3], f2f [F,3] (-1 padded, symmetric on valid entries),
 * vv_ptr/vv_idx = 1-ring CSR without self (util/mesh.py:189-197), vf_ptr/vf_corner = vertex -> incident
 * (3*face + corner).  Targets are float64 as in the reference (n_mesh.vs / n_mesh.fn are f64 numpy).
 * `partials` (ddmp_loss_partials_bytes()) collects per-block float64 sums of S1..S5 and sigma_c;
 * ddmp_loss_finalize turns them into
 *   lossbuf[0..4] = pos_rec (:16), laplacian (:37), norm_rec (:55), fn_bnf (:86, ungated), pos_norm (:140)
 *   lossbuf[5]    = k1*L1 + k2*L2 + k3*L3 + k4*gate4*L4 + k5*L5        (main.py:101-106)
 *   lossbuf[6..10]= gradient coefficients c1..c5 consumed by the *_bwd calls, lossbuf[11] = sigma_c.
 * The *_bwd calls take `coef` = device double[5] (c1..c5); a zero coefficient switches a term off.
 */
size_t ddmp_loss_partials_bytes(void);
int ddmp_loss_vertex_fwd(int64_t V, const float* pos, const double* real_pos, const int32_t* vv_ptr,
                         const int32_t* vv_idx, float* resid /*[V,3]*/, double* partials, ddmp_stream stream);
int ddmp_loss_face_fwd(int64_t F, const float* pos, const float* norm, const double* real_norm,
                       const int32_t* faces, float* fc /*[F,3]*/, float* fa /*[F]*/, float* pn_coef /*[F,3]*/,
                       float* pn_dn /*[F,3]*/, double* partials, ddmp_stream stream);
int ddmp_loss_bnf_fwd(int64_t F, const float* norm, const int32_t* f2f, const float* fc, const float* fa,
                      int loop, float* fcd /*[F,3]*/, float* bnf_n /*[loop+1,F,3]*/, float* bnf_A /*[loop,F,3]*/,
                      double* partials, ddmp_stream stream);
int ddmp_loss_finalize(const double* partials, int64_t V, int64_t F, const double* k_host /*[5]*/, double gate4,
                       double* lossbuf /*[12]*/, ddmp_stream stream);
int ddmp_loss_bnf_bwd(int64_t F, const int32_t* f2f, const float* fa, const float* fcd, int loop,
                      const float* bnf_n, const float* bnf_A, const double* partials, const double* coef,
                      float* G0 /*[F,3]*/, float* scratch /*[2,F,3]*/, ddmp_stream stream);
int ddmp_loss_face_bwd(int64_t F, const float* norm, const double* real_norm, const float* pn_dn,
                       const float* G0 /*nullable*/, const float* n_last /*with G0*/, const double* coef,
                       float* dnorm /*[F,3]*/, ddmp_stream stream);
int ddmp_loss_vertex_bwd(int64_t V, const float* pos, const double* real_pos, const float* resid,
                         const int32_t* vv_ptr, const int32_t* vv_idx, const int32_t* vf_ptr,
                         const int32_t* vf_corner, const float* pn_coef, const float* norm, const double* coef,
                         float* dpos /*[V,3]*/, ddmp_stream stream);

/* Sharded form of the same losses (one rank of a partitioned mesh).  The rank runs the kernels above on its
 * LOCAL sub-mesh = owned vertices / faces plus a ghost closure deep enough that every value an owned row's loss term or
 * gradient reads is computed locally (dist.LossShard: 2*loop+1 face rings, 2 vertex rings, the mesh's last face for the
 * "-1" slots): no exchange inside the losses.  What differs from the single-device calls:
 *   - `own` (uint8 per local row, nullable = all): partial sums count owned rows only; ghost rows are computed, not counted;
 *   - the caller all-reduces the partial sums across ranks -- the sigma_c slice between ddmp_loss_bnf_sigma and
 *     ddmp_loss_bnf_filter, the S1..S5 slices before ddmp_loss_finalize (which takes the GLOBAL V, F);
 *   - F_glob: sigma_c = sum / (3 F_glob) is a mean over all faces of the mesh; the local F still addresses the local
 *     "last face" (the caller places the mesh's last face at the end of its local numbering).
 * partials layout: 6 slices of ddmp_loss_partials_bytes() / 48 doubles: S1..S5, sigma_c. */
int ddmp_loss_vertex_fwd_part(int64_t V, const float* pos, const double* real_pos, const int32_t* vv_ptr,
                              const int32_t* vv_idx, float* resid, double* partials, const uint8_t* own, ddmp_stream stream);
int ddmp_loss_face_fwd_part(int64_t F, const float* pos, const float* norm, const double* real_norm, const int32_t* faces,
                            float* fc, float* fa, float* pn_coef, float* pn_dn, double* partials, const uint8_t* own,
                            ddmp_stream stream);
int ddmp_loss_bnf_sigma(int64_t F, const float* norm, const int32_t* f2f, const float* fc, float* fcd, float* bnf_n,
                        double* partials, const uint8_t* own, ddmp_stream stream);
int ddmp_loss_bnf_filter(int64_t F, int64_t F_glob, const int32_t* f2f, const float* fcd, const float* fa, int loop,
                         float* bnf_n, float* bnf_A, double* partials, const uint8_t* own, ddmp_stream stream);
int ddmp_loss_bnf_bwd_part(int64_t F, int64_t F_glob, const int32_t* f2f, const float* fa, const float* fcd, int loop,
                           const float* bnf_n, const float* bnf_A, const double* partials, const double* coef, float* G0,
                           float* scratch, ddmp_stream stream);

/* ------------------------------------------------------------------ clip_grad_norm_ + Adam (main.py:108-110)
 * One flat float32 arena per net for param / grad / exp_avg / exp_avg_sq.  clip: coefficient
 * min(1, max_norm / (sqrt(sumsq) + 1e-6)) (torch.nn.utils.clip_grad_norm_); Adam: torch defaults
 * (no weight decay, no amsgrad), `step` is 1-based.  Passing clip_sumsq to ddmp_adam_step_f32 applies
 * the clip on the fly (grad buffer left unscaled); ddmp_grad_clip_f32 scales it in place instead.
 */
size_t ddmp_sumsq_workspace_bytes(void);
int ddmp_grad_sumsq_f32(const float* g, int64_t n, double* out /*[1]*/, void* workspace, size_t workspace_bytes,
                        ddmp_stream stream);
int ddmp_grad_clip_f32(float* g, int64_t n, const double* sumsq, float max_norm, ddmp_stream stream);
int ddmp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                       float beta2, float eps, int step, const double* clip_sumsq /*nullable*/, float max_norm,
                       ddmp_stream stream);
/* ddmp_spmm_bnred (dtype-tagged, below; + ddmp_spmm_bnred_ws_bytes): the aggregation (no bias, no prologue) whose output Y is a
 * gradient dZ consumed next by a BatchNorm+LeakyReLU backward also returns that layer's column reductions sums2 =
 * ddmp_bn_bwd_reduce(Y, Yp, scale, shift, mean, rstd) from the kernel's epilogue (float32 partials per 64-row chunk, summed in
 * float64); other widths run the two calls one after the other. */
/* GCNConv.propagate of a transform-first layer (forward) that also returns the BatchNorm statistics of its output
 * (float64 [2C]: column sums of Y and of Y^2 = ddmp_bn_stats_f32(Y)) from the gather kernel's epilogue (round 3).  `ref`
 * [C]: a per-column reference near the column means (the caller's previous batch means; zeros are valid): the kernel
 * sums (y - ref) and (y - ref)^2 in float32 over 16 rows at a time and in float64 from there on, and the shift is undone
 * exactly -- around ref those partial sums are well conditioned whatever mean / std is (float32-class statistics, as
 * nn.BatchNorm1d computes them).  Workspace: ddmp_spmm_bnred_ws_bytes(n_rows, C, DDMP_F32).  Where the fused kernel does not
 * apply (C % 32, misaligned, ref NULL) it runs ddmp_spmm_f32 + ddmp_bn_stats_f32. */
int ddmp_spmm_stats_supported(int C);
int ddmp_spmm_stats_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                        const float* bias /*nullable*/, const float* pro_scale /*nullable*/,
                        const float* pro_shift /*nullable*/, float slope, const float* ref /*[C], nullable*/,
                        double* sums2 /*[2C]*/, void* workspace, size_t workspace_bytes, ddmp_stream stream);
/* dtype-tagged form; workspace >= max(ddmp_spmm_bnred_ws_bytes, ddmp_colreduce_workspace_bytes) */
int ddmp_spmm_stats(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
                    const float* bias, const float* pro_scale, const float* pro_shift, float slope, const float* ref,
                    double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream);
/* backward of GCNConv.propagate of a transform-first layer, with the backward of the BatchNorm1d+LeakyReLU behind it
 * (util/networks.py:51-62 under autograd) rebuilt on the gather: out = A_hat . dY,
 * dY = a * dZ * lrelu'(a * Yb + b) + c1 * Yb + c0 per column (what ddmp_bn_bwd_apply_f32 would have written and this
 * kernel read back).  C % 32 == 0 (ddmp_spmm_bnbwd_supported). */
int ddmp_spmm_bnbwd_supported(int C);
int ddmp_spmm_bnbwd_f32(const ddmp_graph* g, const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb,
                        float* out, int64_t ld_out, int C, const float* a, const float* b, const float* c1,
                        const float* c0, float slope, ddmp_stream stream);


/* ddmp_gemm_nt_f32 that also returns the BatchNorm statistics of its output (float64 [2M]: column sums of Y and of
 * Y^2 over the n_rows rows = what ddmp_bn_stats_f32(Y) returns; GCNConv -> BatchNorm1d, util/networks.py:52-53).  The
 * row-panel kernel produces them in its epilogue (float64 partials per 64 rows); other shapes run
 * the GEMM followed by ddmp_bn_stats_f32. */
size_t ddmp_gemm_nt_stats_workspace_bytes(int64_t n_rows, int M);
int ddmp_gemm_nt_stats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy,
                           int64_t n_rows, int K, int M, const float* bias /*nullable*/,
                           const float* pro_scale /*nullable*/, const float* pro_shift /*nullable*/, float slope,
                           double* sums2 /*[2M]*/, void* workspace, size_t workspace_bytes, void* stats_ws,
                           size_t stats_ws_bytes, ddmp_stream stream);

/* BatchNorm+LeakyReLU backward fused into the operand load of the GEMMs that consume dY (agg-first layers of the
 * engine; replaces ddmp_bn_bwd_apply_f32 + ddmp_gemm_nn_f32 + ddmp_gemm_tn_f32 of util/networks.py's autograd chain):
 *     dY = a * dZ * lrelu'(a * Yb + b) + c1 * Yb + c0        (a, b, c1, c0 per column, from ddmp_bn_bwd_prepare_f32)
 * ddmp_gemm_bnbwd_supported(cout, cin, n_rows) says whether BOTH fused GEMMs exist for a layer of that shape and row
 * count in the current GEMM mode (the panel kernels they live in are used from ~30k rows); the conv-bias gradient (column sums of dY) is exactly zero in exact arithmetic and is written as 0. */
/* dgrad of a transform-first layer with the NEXT BatchNorm-backward column reductions from its epilogue (round 3, row-register
 * kernel): out[n,K] = A[n,M] . W[M,K] and sums2[2K] = what ddmp_bn_bwd_reduce_f32(out, Yp, scale, shift, mean, rstd) returns,
 * Yp [n,K] = the previous layer's conv output; stats_ws >= ddmp_gemm_nt_stats_workspace_bytes(n_rows, K) */
/* Weights prepared once per iteration (round 3).  Every ddmp_gemm_nt* / ddmp_gemm_nn* call splits its weight matrix into
 * 16-bit planes in its workspace before the product (absolute maximum + split: 2-3 launches per call).
 * ddmp_gemm_prepare_weights does that for n <= 24 matrices in two launches, into caller-owned buffers planes[i] (>=
 * ddmp_gemm_rows_workspace_bytes(K[i], M[i]) bytes, 16-byte aligned), in the layout the product over n_rows rows will want:
 * W[i] is [M[i], K[i]] float32 with leading dimension ldw[i]; form[i] 0 = forward (ddmp_gemm_nt*: Y[n,M] = f(A[n,K]) . W^T,
 * has_pro[i] = with a prologue), 1 = dgrad (ddmp_gemm_nn*: Y[n,K] = A[n,M] . W).  scratch: 8 n floats.
 * A GEMM call then passes planes[i] as its `workspace` and is announced by ddmp_gemm_next_prepared() (this host thread, the
 * NEXT ddmp_gemm_* call): it skips its own split if the buffer was prepared for exactly this matrix, shape and route, and
 * splits as before otherwise.  The caller re-prepares whenever the weights change. */
int ddmp_gemm_prepare_weights(int n, const float* const* W, const int64_t* ldw, const int* M, const int* K, const int* form,
                              const int* has_pro /*nullable*/, void* const* planes, const size_t* planes_bytes,
                              int64_t n_rows, float* scratch, ddmp_stream stream);
/* the owner of a plane buffer is about to free it: drop what this thread recorded for it (NULL: everything) */
int ddmp_gemm_forget_planes(const void* planes);
int ddmp_gemm_tn_bnbwd_supported(int cout, int cin, int64_t n_rows);   /* the fused wgrad alone (a first layer has no dgrad) */
int ddmp_gemm_nn_bnred_supported(int M, int K, int64_t n_rows);
int ddmp_gemm_nn_bnred_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* out, int64_t ld_out,
                           int64_t n_rows, int M, int K, const float* Yp, int64_t ldyp, const float* scale,
                           const float* shift, const float* mean, const float* rstd, float slope, double* sums2,
                           void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes,
                           ddmp_stream stream);
int ddmp_gemm_bnbwd_supported(int cout, int cin, int64_t n_rows);
int ddmp_gemm_nn_bnbwd_f32(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* W, int64_t ldw,
                           float* out, int64_t ld_out, int64_t n_rows, int M, int K, const float* a, const float* b,
                           const float* c1, const float* c0, float slope, void* workspace, size_t workspace_bytes,
                           ddmp_stream stream);
int ddmp_gemm_tn_bnbwd_f32(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* Z, int64_t ldz,
                           float* dW, int64_t lddw, int64_t n_rows, int M, int K, const float* a, const float* b,
                           const float* c1, const float* c0, const float* pro_scale /*nullable*/,
                           const float* pro_shift /*nullable*/, float slope, void* workspace, size_t workspace_bytes,
                           ddmp_stream stream);

/* graph-replay form: the step count lives on the device.  ddmp_adam_prepare does ++(*step_counter) and writes
 * coef = { lr / (1 - beta1^t), sqrt(1 - beta2^t) }; ddmp_adam_step_dev_f32 reads coef instead of host values. */
int ddmp_adam_prepare(int32_t* step_counter, float lr, float beta1, float beta2, float* coef /*[2]*/,
                      ddmp_stream stream);
int ddmp_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                           float eps, const float* coef /*[2]*/, const double* clip_sumsq /*nullable*/,
                           float max_norm, ddmp_stream stream);

/* ------------------------------------------------------------------ evaluation block (main.py:117-123)
 * Face normals of predicted positions (util/mesh.py:87-92; float32 like the reference's float32 `new_pos`)
 * and MAD = mean over faces of deg(arccos(clip(n1.n2, -1, 1))) in float64 (util/loss.py:261-272).
 */
int ddmp_face_normals_f32(int64_t F, const float* pos, const int32_t* faces, float* fn /*[F,3]*/,
                          float* fa /*[F] nullable*/, ddmp_stream stream);
/* vertex update from predicted normals (util/models.py:31-44): `loop` sweeps of
 * p_v += mean_{f in F(v)} ((c_f - p_v).n_f) n_f, in place; fc_scratch [F,3] */
int ddmp_vertex_update_f32(int64_t V, int64_t F, float* pos, const float* norm, const int32_t* faces,
                           const int32_t* vf_ptr, const int32_t* vf_corner, float* fc_scratch, int loop,
                           ddmp_stream stream);
size_t ddmp_mad_workspace_bytes(void);
int ddmp_mad_f64(int64_t F, const float* n1 /*[F,3] f32*/, const double* n2 /*[F,3] f64*/, double* out /*[1]*/,
                 void* workspace, size_t workspace_bytes, ddmp_stream stream);

/* ------------------------------------------------------------------ bfloat16-feature mode (SURVEY.md §8b `dtype`)
 * BASELINE.json configs[1] ("bf16 features"): node features, saved activations and activation gradients [N, C] are
 * bfloat16 in HBM (raw uint16_t bits here; rows 16-byte aligned: C and every leading dimension a multiple of 8);
 * kernels unpack to float32, accumulate in float32 (BatchNorm statistics in float64) and round to nearest-even once on
 * the store.  Parameters, parameter gradients, optimizer state, coefficients (bias, scale, shift, c1, c0 ...) and the
 * [n,3] outputs / loss gradients stay float32.  The GEMMs run ONE v_mfma_f32_32x32x16_bf16 product per step (the float32
 * weights are converted to bf16 planes per call: workspace), no operand splitting and no scale slots.
 * Each *_bf16 entry point mirrors its *_f32 namesake argument for argument; the dtype-tagged forms
 * ddmp_spmm / ddmp_gemm_nt / ... below take `void*` features and dispatch on `dtype`.
 */
#define DDMP_F32 0
#define DDMP_BF16 1

/* bfloat16 features, plain GEMMs: through the dtype-tagged ddmp_gemm_nt / _nn / _tn below (nt / nn: K resp. M -- the contraction -- a
 * multiple of 32, nt also K = 8 | 16: the first layer; outputs <= 512 columns; workspace >= ddmp_gemm_rows_ws_bytes(K, M, DDMP_BF16)
 * holds the bf16 weight planes) */
/* Fused forms on the row-register kernel (csrc/gemm_rr_b16.inc; round 3): the streaming BatchNorm passes of the bf16 step
 * folded into the GEMMs, as the *_f32 namesakes do for float32 features.
 *   ddmp_gemm_fused_bf16_supported  bit 0: ddmp_gemm_nt_stats_bf16, bit 1: the two *_bnbwd_bf16 forms exist AND pay for a
 *                                   layer cin -> cout over n_rows rows (measured thresholds, csrc/gemm_b16.hip)
 *   ddmp_gemm_nt_stats_bf16         ddmp_gemm_nt_bf16 + sums2[2M] (float64) = column sums of Y and Y^2 as STORED (bf16),
 *                                   i.e. ddmp_bn_stats_bf16(Y), from the epilogue; stats_ws >= ..._stats_bf16_workspace_bytes
 *   ddmp_gemm_nn_bnbwd_bf16         out[n,K] = dY . W[M,K], dY = bf16(a dZ lrelu'(a Yb + b) + c1 Yb + c0) rebuilt on the operand
 *                                   load: what ddmp_bn_bwd_apply_bf16 would have written, never stored
 *   ddmp_gemm_tn_bnbwd_bf16         dW[M,K] = dY^T . f(Z), the same dY (reference op: GCNConv.lin backward behind
 *                                   BatchNorm1d + LeakyReLU, util/networks.py:31-44,51-62)
 *   (round 5's ddmp_gemm_nn_bnred_bf16 -- measured no gain inside the step -- left the library in round 6: experiments/r05/) */
int ddmp_gemm_fused_bf16_supported(int cout, int cin, int64_t n_rows);
size_t ddmp_gemm_nt_stats_bf16_workspace_bytes(int64_t n_rows, int M);
int ddmp_gemm_nt_stats_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                            int64_t n_rows, int K, int M, const float* bias, const float* pro_scale,
                            const float* pro_shift, float slope, double* sums2, void* workspace, size_t workspace_bytes,
                            void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream);
int ddmp_gemm_nn_bnbwd_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Yb, int64_t ldyb, const float* W,
                            int64_t ldw, uint16_t* out, int64_t ld_out, int64_t n_rows, int M, int K, const float* a,
                            const float* b, const float* c1, const float* c0, float slope, void* workspace,
                            size_t workspace_bytes, ddmp_stream stream);
int ddmp_gemm_tn_bnbwd_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Yb, int64_t ldyb, const uint16_t* Z,
                            int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M, int K, const float* a,
                            const float* b, const float* c1, const float* c0, const float* pro_scale,
                            const float* pro_shift, float slope, void* workspace, size_t workspace_bytes,
                            ddmp_stream stream);
/* BatchNorm passes: C a power of two in [16, 1024]; workspace = ddmp_colreduce_workspace_bytes(n_rows, C) */
int ddmp_bn_stats_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, int C, double* sums /*[2C]*/, void* workspace,
                       size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_lrelu_apply_bf16(const uint16_t* Y, int64_t ldy, uint16_t* Z, int64_t ldz, int64_t n_rows, int C,
                             const float* scale, const float* shift, float slope, ddmp_stream stream);
int ddmp_bn_bwd_reduce_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, int64_t n_rows, int C,
                            const float* scale, const float* shift, const float* mean, const float* rstd, float slope,
                            double* sums2 /*[2C]*/, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_bwd_apply_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, uint16_t* dY, int64_t lddy,
                           int64_t n_rows, int C, const float* scale, const float* shift, const float* c1,
                           const float* c0, float slope, double* dbias_sums /*[C]*/, void* workspace,
                           size_t workspace_bytes, ddmp_stream stream);
/* Multi-device halo packing (the reference is single-device, main.py:51): dst[r,:] = src[idx[r],:]
 * (scatter = 0: the boundary rows a rank sends, in the order of its halo plan) or dst[idx[r],:] = src[r,:] (scatter = 1);
 * idx int64 on the device; rows of C elements with C * element size a multiple of 16 bytes. */
int ddmp_rows_gather(const void* src, int64_t ld_src, const int64_t* idx, int64_t n, int C, int dtype, void* dst,
                     int64_t ld_dst, int scatter, ddmp_stream stream);
/* Communicator + halo plan + exchange on the caller's stream (csrc/comm.hip): one process per GPU, RCCL over xGMI.
 *   ddmp_comm_unique_id    rank 0 draws the 128-byte id, the host side distributes it (any out-of-band channel)
 *   ddmp_comm_create       ncclCommInitRank; status 1000 + ncclResult_t on RCCL errors
 *   ddmp_halo_plan_create  send_idx_host [sum send_counts]: LOCAL owned row of every boundary row, grouped by destination
 *                          rank in the order the receiver stores its halo; recv_counts per source rank (sum = n_cols - n_rows)
 *   ddmp_halo_exchange     rows [0, n_rows) of T are owned, rows [n_rows, n_cols) the halo (grouped by source rank): packs the
 *                          boundary rows (kernel), then ONE ncclGroup of send/recv pairs that lands the halo rows in place
 *                          (ld == C) and, when sums != NULL, the all-reduce of `n_sums` float64 BatchNorm column sums
 *   ddmp_comm_allreduce_sum / ddmp_comm_allgather   gradient arena, loss inputs */
typedef struct ddmp_comm ddmp_comm;
typedef struct ddmp_halo_plan ddmp_halo_plan;
int ddmp_comm_unique_id(char* id128_host);
int ddmp_comm_create(int rank, int world, const char* id128_host, ddmp_comm** out);
int ddmp_comm_destroy(ddmp_comm* comm);
int ddmp_halo_plan_create(int world, int rank, int64_t n_rows, int64_t n_cols, const int64_t* send_idx_host,
                          const int64_t* send_counts_host, const int64_t* recv_counts_host, ddmp_halo_plan** out);
int ddmp_halo_plan_destroy(ddmp_halo_plan* plan);
size_t ddmp_halo_pack_bytes(const ddmp_halo_plan* plan, int C, int dtype);
int ddmp_halo_exchange(ddmp_comm* comm, const ddmp_halo_plan* plan, void* T, int64_t ld, int C, int dtype, void* pack_ws,
                       size_t ws_bytes, double* sums /*nullable*/, int n_sums, ddmp_stream stream);
int ddmp_comm_allreduce_sum(ddmp_comm* comm, void* buf, int64_t n, int is_f64, ddmp_stream stream);
int ddmp_comm_allgather(ddmp_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, ddmp_stream stream);
/* a one-thread kernel named ddmp_trace_marker_kernel: brackets a region in a rocprofv3 kernel trace (measurement aid) */
int ddmp_trace_marker(ddmp_stream stream);
/* 0 for the product library; a bit mask of the timing-only ablation macros a diagnostic build was compiled with (such builds
 * compute WRONG results: scripts/build_ablation*.sh).  The Python host side refuses a non-zero library unless it was named
 * explicitly through DDMP_LIB. */
int ddmp_build_ablation_flags(void);
/* measurement aid (bench.py's copy yardstick): a streaming device copy of `bytes` (multiple of 16, both pointers 16-byte
 * aligned) written like the library's HBM-bound kernels; mode 0 plain, 1 nontemporal loads / stores */
int ddmp_copy_probe(const void* src, void* dst, int64_t bytes, int mode, ddmp_stream stream);
/* ... and the same copy in the gather's access pattern: 64-row chunks of a row-major [n_rows, row_bytes] matrix walked one
 * 128-byte slab at a time (row_bytes a multiple of 128): the ceiling of that pattern, whatever the graph */
int ddmp_copy_probe_rows(const void* src, void* dst, int64_t n_rows, int row_bytes, ddmp_stream stream);
/* float32 -> bfloat16 (round to nearest even) of n contiguous elements */
int ddmp_f32_to_bf16(const float* in, uint16_t* out, int64_t n, ddmp_stream stream);

/* dtype-tagged forms (dtype = DDMP_F32 | DDMP_BF16): what a binding of the reference's GCNConv / BatchNorm1d / Linear
 * call sites (util/networks.py:15-26,31-44,51-67) uses when the feature dtype is a run-time choice */
int ddmp_spmm(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype, const float* bias,
              const float* pro_scale, const float* pro_shift, float slope, ddmp_stream stream);
size_t ddmp_spmm_bnred_ws_bytes(int64_t n_rows, int C, int dtype);
int ddmp_spmm_bnred(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
                    const void* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean,
                    const float* rstd, float slope, double* sums2, void* workspace, size_t workspace_bytes,
                    ddmp_stream stream);
int ddmp_spmm_bnbwd(const ddmp_graph* g, const void* dZ, int64_t lddz, const void* Yb, int64_t ldyb, void* out,
                    int64_t ld_out, int C, int dtype, const float* a, const float* b, const float* c1, const float* c0,
                    float slope, ddmp_stream stream);
size_t ddmp_gemm_rows_ws_bytes(int K, int M, int dtype);
int ddmp_gemm_nt(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows, int K,
                 int M, int dtype, const float* bias, const float* pro_scale, const float* pro_shift, float slope,
                 void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_gemm_nn(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows, int M,
                 int K, int dtype, void* workspace, size_t workspace_bytes, ddmp_stream stream);
size_t ddmp_gemm_tn_ws_bytes(int64_t n_rows, int M, int K, int dtype);
int ddmp_gemm_tn(const void* G, int64_t ldg, const void* Z, int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M,
                 int K, int dtype, const float* pro_scale, const float* pro_shift, float slope, void* workspace,
                 size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_stats(const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype, double* sums, void* workspace,
                  size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_bwd_reduce(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype,
                       const float* scale, const float* shift, const float* mean, const float* rstd, float slope,
                       double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_bwd_apply(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, void* dY, int64_t lddy, int64_t n_rows,
                      int C, int dtype, const float* scale, const float* shift, const float* c1, const float* c0,
                      float slope, double* dbias_sums, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_head_fwd(const void* Y, int64_t ldy, int64_t n_rows, int dtype, const float* scale, const float* shift,
                  float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                  const float* x_pos, float* out, ddmp_stream stream);
int ddmp_head_bwd(const void* Y, int64_t ldy, int64_t n_rows, int dtype, const float* scale, const float* shift,
                  float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                  const float* dout, void* dZ, int64_t lddz, float* dW1, float* db1, float* dW2, float* db2,
                  void* workspace, size_t workspace_bytes, ddmp_stream stream);


/* ------------------------------------------------------------------ ABI 3: per-call options instead of "armed" state
 * Everything the ddmp_*_next_* calls used to attach to "the next call of this host thread" is an explicit, nullable last
 * argument of the call itself: `name_o(<the arguments of name>, const ddmp_opts* opts)`.  The options apply to THAT call and
 * to nothing else (whatever it returns); opts == NULL is the plain call.  The arming calls above stay for one more round as
 * deprecated wrappers (they fill the same per-thread record the _o forms set and clear around their call).
 *   DDMP_OPT_BN_FWD   the call's float64 [2C] reduction (sum y, sum y^2) also yields what ddmp_bn_prepare_f32 would write:
 *                     bn_in = {gamma, beta}, bn_out = {scale, shift, mean, rstd, running_mean | NULL, running_var | NULL}
 *   DDMP_OPT_BN_BWD   ... (sum g, sum g yhat) also yields what ddmp_bn_bwd_prepare_f32 would write:
 *                     bn_in = {scale, mean, rstd}, bn_out = {dgamma, dbeta, c1, c0}
 *                     (bn_C must be the reduction's width, bn_n_total the row count of the whole batch)
 *   DDMP_OPT_SCALES   f16x3 GEMM mode: persistent scale slots of the operands (slot_a: the row operand A / dZ / G, slot_b:
 *                     Z of the tn forms, either may be NULL); prime != 0: measure now (first iteration)
 *   DDMP_OPT_PREPARED `workspace` holds weight planes written by ddmp_gemm_prepare_weights for exactly this product */
#define DDMP_OPT_BN_FWD 1u
#define DDMP_OPT_BN_BWD 2u
#define DDMP_OPT_SCALES 4u
#define DDMP_OPT_PREPARED 8u
typedef struct ddmp_opts {
    uint32_t struct_size; /* sizeof(ddmp_opts) of the caller's header */
    uint32_t flags;       /* DDMP_OPT_* */
    double bn_n_total;
    int32_t bn_C;
    float bn_eps, bn_momentum;
    int32_t prime;
    const float* bn_in[3];
    float* bn_out[6];
    float* slot_a;
    float* slot_b;
} ddmp_opts;
int ddmp_bn_stats_o(const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype, double* sums, void* workspace,
    size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_bn_bwd_reduce_o(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype,
    const float* scale, const float* shift, const float* mean, const float* rstd, float slope, double* sums2,
    void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_spmm_stats_o(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
    const float* bias, const float* pro_scale, const float* pro_shift, float slope, const float* ref, double* sums2,
    void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_spmm_bnred_o(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
    const void* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean, const float* rstd,
    float slope, double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_nt_o(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows,
    int K, int M, int dtype, const float* bias, const float* pro_scale, const float* pro_shift, float slope,
    void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_nn_o(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows,
    int M, int K, int dtype, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_tn_o(const void* G, int64_t ldg, const void* Z, int64_t ldz, float* dW, int64_t lddw, int64_t n_rows,
    int M, int K, int dtype, const float* pro_scale, const float* pro_shift, float slope, void* workspace,
    size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_nt_stats_f32_o(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy,
    int64_t n_rows, int K, int M, const float* bias , const float* pro_scale , const float* pro_shift , float slope,
    double* sums2 , void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream,
    const ddmp_opts* opts);
int ddmp_gemm_nt_stats_bf16_o(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
    int64_t n_rows, int K, int M, const float* bias, const float* pro_scale, const float* pro_shift, float slope,
    double* sums2, void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream,
    const ddmp_opts* opts);
int ddmp_gemm_nn_bnred_f32_o(const float* A, int64_t lda, const float* W, int64_t ldw, float* out, int64_t ld_out,
    int64_t n_rows, int M, int K, const float* Yp, int64_t ldyp, const float* scale, const float* shift,
    const float* mean, const float* rstd, float slope, double* sums2, void* workspace, size_t workspace_bytes,
    void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_nn_bnbwd_f32_o(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* W, int64_t ldw,
    float* out, int64_t ld_out, int64_t n_rows, int M, int K, const float* a, const float* b, const float* c1,
    const float* c0, float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);
int ddmp_gemm_tn_bnbwd_f32_o(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* Z, int64_t ldz,
    float* dW, int64_t lddw, int64_t n_rows, int M, int K, const float* a, const float* b, const float* c1,
    const float* c0, const float* pro_scale , const float* pro_shift , float slope, void* workspace,
    size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DDMP_HIP_H */
