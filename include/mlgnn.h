/*
 * mlgnn.h -- C ABI of libmlgnn.so: the MI355X (gfx950) hot path of the multilevel-GNN
 * forward/backward pass.
 *
 * Boundary contract:
 *   - every entry point is extern "C", takes raw DEVICE pointers + sizes and the caller's
 *     HIP stream (a hipStream_t passed as void*), returns int:
 *         0  ok,   >0  a hipError_t from the launch,   <0  argument error (MLGNN_E_*);
 *   - no allocation, no global state, no host synchronisation, never throws or exits;
 *     scratch memory is a caller-provided workspace (size from the *_workspace_bytes call);
 *   - all tensors are contiguous row-major; float tensors are fp32; indices are int32.
 *
 * Graph layout ("CSR by destination"): rowptr[N+1], col[E] = source node of every edge,
 * edges of one destination kept in their original (COO) order; eid[E] = original COO
 * position of the edge.  The transposed layout ("CSR by source") carries col_t[E] =
 * destination node, pos_t[E] = position of that edge in the by-destination order.
 *
 * Each entry point names the reference interface it stands in for
 * (paths relative to the reference tree).
 */
#ifndef MLGNN_H
#define MLGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLGNN_ABI_VERSION 19

/* argument errors */
#define MLGNN_E_NULL      (-1)  /* a required pointer is NULL                  */
#define MLGNN_E_SHAPE     (-2)  /* negative / inconsistent size                 */
#define MLGNN_E_MODE      (-3)  /* unknown or unsupported mode / aggregator     */
#define MLGNN_E_DTYPE     (-4)  /* dtype not supported by this build            */
#define MLGNN_E_WORKSPACE (-5)  /* workspace too small                          */
#define MLGNN_E_ALIGN     (-6)  /* pointer not aligned for the vector width     */

/* message: what one edge (j -> i) contributes before the reduction */
#define MLGNN_MSG_IDENTITY 0  /* x_j                                                            */
#define MLGNN_MSG_WEIGHTED 1  /* x_j * w_e            SAGEConv.message, torch_vertex.py:279-281  */
#define MLGNN_MSG_GEN      2  /* relu(x_j + e_e)+eps  GENConv.message,  torch_vertex.py:94-101   */

/* edge term e_e of MLGNN_MSG_GEN */
#define MLGNN_EDGE_NONE  0    /* e_e = 0                                                         */
#define MLGNN_EDGE_RANK1 1    /* e_e[c] = sum_k a_e[k]*U[k][c] + v[c]: edge_rank (1, 2, 4 or 8)  *
                               * raw attributes per edge through the two stacked Linear encoders  *
                               * deepergcn.py:87-90,213 + torch_vertex.py:68,77, kept factored    */
#define MLGNN_EDGE_FULL  2    /* e_e = efull[eid_e, :]: materialised [E,d] embedding             */

/* aggregator: GenMessagePassing.aggregate, torch_message.py:44-85 */
#define MLGNN_AGGR_SUM     0  /* 'add'  (PyG SumAggregation)                                     */
#define MLGNN_AGGR_MEAN    1  /* 'mean' sum / clamp(count,1)                                     */
#define MLGNN_AGGR_MAX     2  /* 'max'  first maximal edge wins, empty row -> 0                  */
#define MLGNN_AGGR_SOFTMAX 3  /* sum_e m_e * softmax_e(t*m_e)   (:49-57)                         */
#define MLGNN_AGGR_POWER   4  /* clamp(mean(clamp(m,1e-7,10)^p),1e-7,10)^(1/p)   (:68-74)        */

#define MLGNN_DTYPE_F32  0
#define MLGNN_DTYPE_BF16 1   /* storage of [N,d] / [E,d] activations only; arithmetic, aux and edge vectors stay fp32 */

int mlgnn_version(void);

/*
 * Long rows ("hub" nodes: a transcription factor's cross-omics edges, dataloader/multiloader.py:664-671).  One
 * wavefront owns one row in the aggregation kernels; rows with more than `cap` edges are cut into chunks of `cap`
 * that run as rows of their own and are combined in chunk order (no atomics; csrc/hub.hip).  mlgnn_hub_rows builds
 * the chunk tables of ONE direction on the device from a row pointer (rowptr for the forward, rowptr_t for the
 * backward): vrows [capacity][3] = {real row, first edge, one past the last edge} of every chunk behind a row's first
 * `cap` edges, hubs [capacity][3] = {row, first chunk, number of chunks} per split row, counts[2] = their numbers.
 * capacity = mlgnn_hub_capacity(E, cap); tmp: mlgnn_hub_scratch_bytes(capacity, d) bytes of scratch for the chunks'
 * partial results, private to one call.  Pass the struct to mlgnn_csr_aggregate_fwd / _bwd (NULL or cap = 0: off).
 */
typedef struct mlgnn_hub {
  int32_t cap;
  int32_t capacity;
  const int32_t* vrows;
  const int32_t* hubs;
  const int32_t* counts;
  void* tmp;
  int64_t tmp_bytes;
} mlgnn_hub_t;
int64_t mlgnn_hub_capacity(int64_t E, int cap);
int64_t mlgnn_hub_scratch_bytes(int64_t capacity, int64_t d);
int mlgnn_hub_rows(const int32_t* rowptr, int64_t N, int cap, int64_t capacity, int32_t* vrows, int32_t* hubs,
                   int32_t* counts, void* stream);

/* Number of float elements of workspace mlgnn_csr_aggregate_bwd needs: per-workgroup partials of a
 * factored edge term of edge_rank attributes (0 when the edge mode is not EDGE_RANK1) plus, for
 * AGGR_SOFTMAX without learn_t, the rescaled cotangent [N,d] of the one-row gather path, or, for AGGR_MAX with
 * d % 4 == 0, the one-byte winner slots [N,d] the backward gathers instead of argmax.  May be 0 (then workspace may be NULL). */
int64_t mlgnn_csr_aggregate_bwd_workspace_floats(int64_t N, int64_t d, int dtype, int edge_rank,
                                                 int aggr, int learn_t);

/*
 * Fused message + aggregation over the incoming edges of every node.
 * Replaces: MessagePassing.propagate -> message -> aggregate as called from
 *   GENConv.forward  models/gcn_lib/sparse/torch_vertex.py:81-82 (+ :94-101, torch_message.py:44-85)
 *   SAGEConv.forward models/gcn_lib/sparse/torch_vertex.py:277-286 (aggregate-then-transform)
 *
 *   x       [N,d]   node features (rows gathered by col)
 *   rowptr  [N+1], col [E]                 CSR by destination
 *   ew      [E] / [E,edge_rank]  per-edge scalars in CSR order: weight (MSG_WEIGHTED) or the raw
 *                   attribute row a_e, zero padded to edge_rank columns (EDGE_RANK1); NULL otherwise
 *   eu [edge_rank,d], ev [d]     factored edge term (EDGE_RANK1); edge_rank in {1,2,4,8}, ignored
 *                   for the other edge modes
 *   efull   [E0,d], eid [E]               materialised edge embedding + COO position (EDGE_FULL)
 *   out     [N,d]
 *   aux     [N,d]   SOFTMAX: log2-sum-exp of t*m per (node,channel); POWER: mean(clamp(m)^p)
 *                   before the outer clamp; may be NULL when no backward is wanted
 *   aux2    [N,d]   optional second moment for the learnable temperature / exponent:
 *                   SOFTMAX: sum_e w_e m_e^2 ; POWER: mean(clamp(m)^p * ln clamp(m)); NULL to skip
 *   argmax  [N,d]   MAX: by-destination edge position of the winner; -1 where no gradient flows: an empty row, or a
 *                   winner on relu's flat side (z <= 0: its message is the constant eps)
 *   row_max [N]     optional (NULL to skip): max_c |out[i][c]| per row, for the per-row scaling of the Linear that
 *                   consumes `out` (mlgnn_tallgemm_nt); only for d = 4 * 2^k <= 256 (fp32), 8 * 2^k <= 512 (bf16)
 *   t, p            softmax temperature / power exponent; when t_dev / p_dev is non-NULL the value is
 *                   read from that device address instead (learnable parameters: no host sync)
 *   add_root        non-zero: out = x + aggregate (GENConv's h = x + m, torch_vertex.py:89, same pass);
 *                   the backward then adds grad_out to grad_x.  Not combinable with learn_t.
 *   hub             long-row tables of the by-destination CSR (mlgnn_hub_rows over rowptr), or NULL; with
 *                   SOFTMAX / POWER it needs aux (the chunks are combined through it)
 */
int mlgnn_csr_aggregate_fwd(const void* x, const int32_t* rowptr, const int32_t* col,
                            const float* ew, const float* eu, const float* ev,
                            const void* efull, const int32_t* eid,
                            void* out, float* aux, float* aux2, int32_t* argmax, float* row_max,
                            int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                            int aggr, float t, float p, const float* t_dev, const float* p_dev,
                            float eps, int add_root, const mlgnn_hub_t* hub, void* stream);

/*
 * Backward of the above with respect to x (and the edge term), atomic-free, on the
 * transposed graph: one wavefront per SOURCE node walks its outgoing edges.
 *
 *   grad_out [N,d]  cotangent of out.  POWER: the caller passes
 *                   q = grad_out * mu^(1/p-1) * [1e-7<=mu<=10] / clamp(deg,1)  instead.
 *   x, out, aux, argmax: as produced / consumed by the forward
 *   rowptr_t [N+1], col_t [E] (destination), pos_t [E] (by-destination position)
 *   rowptr   [N+1]  by-destination row pointer (in-degree for MEAN; nodes with incoming edges for SOFTMAX)
 *   ew_t [E] / [E,edge_rank], eid_t [E]: ew / eid permuted to by-source order
 *   grad_x   [N,d]
 *   grad_efull [E0,d]  EDGE_FULL: d loss / d efull, written at eid (every row written once)
 *   geid_t [E] or NULL  row of grad_efull each edge writes when it differs from the row of efull it reads: efull may be
 *                   a TABLE read through eid (several edges per row, e.g. one row per edge type, deepergcn.py:103-104)
 *                   while the gradient is still produced per edge ([E,d], geid_t = the edge's own id) and reduced to
 *                   the table afterwards (mlgnn_embedding_bwd); NULL: geid_t = eid_t
 *   accumulate_efull  non-zero: grad_efull += instead of = (the same [E0,d] embedding feeds several layers --
 *                   deepergcn.py:232-281 passes one edge_emb to every GENConv -- and their edge gradients are
 *                   summed in place instead of by separate [E0,d] additions)
 *                   3 (MAX over fp32 rows with a TABLE read through eid): nothing per edge is produced, grad_efull is
 *                   NULL -- the table's gradient comes from the destination side (mlgnn_max_table_grad)
 *                   2 (MAX over fp32 rows with a TABLE read through eid): grad_efull is the fixed-point accumulator of
 *                   the TABLE's gradient prepared by mlgnn_table_grad_begin and geid_t names every edge's table row
 *                   (= eid_t): with max only the winning edge of (i, c) has a gradient, 1 / degree of the [E,d]
 *                   per-edge gradient is non-zero, and it is added to the table directly with integer atomics (sums
 *                   independent of arrival order: bitwise reproducible) -- no [E,d] gradient is written or re-read
 *   grad_uv  [edge_rank+1,d]  EDGE_RANK1: d loss / d eu (edge_rank rows), then d loss / d ev
 *   learn_t  non-zero: SOFTMAX weights carry gradient (torch_message.py:51-52)
 *   workspace: mlgnn_csr_aggregate_bwd_workspace_floats(N,d,dtype,edge_rank,aggr,learn_t) floats
 *   [N,d] tensors of 4 GiB and more (N * d * 4 >= 2^32) run on instantiations with 64-bit row addresses (d % 4 == 0 for
 *            fp32, d % 8 == 0 for bf16 and 16-byte aligned operands; MLGNN_E_SHAPE otherwise); same results.
 *   grad_shifted [N,d], shift_flag [4 x int32]  optional (both or neither; SOFTMAX without learn_t): the rescaled
 *            cotangent grad_out * 2^(-aux) and its overflow flag, already written by the producer of grad_out
 *            (mlgnn_tallgemm_nt_shift); the streaming pre-pass that would compute them is skipped and the workspace
 *            needs the edge-term partials only (query with aggr = MLGNN_AGGR_SUM)
 */
int mlgnn_csr_aggregate_bwd(const void* grad_out, const void* x, const void* out, const float* aux,
                            const int32_t* argmax,
                            const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                            const int32_t* rowptr,
                            const float* ew_t, const float* eu, const float* ev,
                            const void* efull, const int32_t* eid_t, const int32_t* geid_t,
                            void* grad_x, void* grad_efull, float* grad_uv,
                            float* workspace, int64_t workspace_floats,
                            int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                            int aggr, int learn_t, float t, float p, const float* t_dev, const float* p_dev,
                            float eps, int add_root, int accumulate_efull, const mlgnn_hub_t* hub,
                            const void* grad_shifted, const int32_t* shift_flag, void* stream);

/*
 * The same backward when x was  y = relu?(LayerNorm(h))  -- the res+ block's norm + ReLU in front of the conv,
 * models/deepergcn.py:236-241 -- and this aggregation (with add_root) is what consumes y: the row epilogue, which
 * holds the finished row of d loss / d y, takes it through that LayerNorm's backward, so
 *     grad_x [N,d]          receives  d loss / d h  (+ grad_extra)  instead of d loss / d y
 *     grad_gamma_beta [2,d] receives  d loss / d gamma, d loss / d beta of the LayerNorm
 *     row_max [N] or NULL   max |grad_x| per row (what the GEMMs that consume grad_x scale their operand by)
 * and the separate LayerNorm-backward pass over [N,d] (mlgnn_layernorm_act_bwd) does not run.
 *   h [N,d], mean [N], rstd [N], gamma [d], beta [d]: the LayerNorm's input, statistics and affine parameters
 *   grad_extra [N,d] or NULL: gradient arriving on h along the block's identity branch (deepergcn.py:241), added here
 *   workspace: mlgnn_csr_aggregate_bwd_ln_workspace_floats(N, d) floats
 * fp32, d = 4 * 2^k <= 256 (one channel chunk per lane group), no long-row split (hub NULL or cap 0), no learn_t:
 * MLGNN_E_MODE otherwise, and the caller runs the two passes.
 */
typedef struct mlgnn_ln_fold {
  const float* h;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  const float* grad_extra;
  float* row_max;
  float* grad_gamma_beta;
  float* workspace;
  int64_t workspace_floats;
  int32_t relu;
} mlgnn_ln_fold_t;
int64_t mlgnn_csr_aggregate_bwd_ln_workspace_floats(int64_t N, int64_t d);
int mlgnn_csr_aggregate_bwd_ln(const void* grad_out, const void* x, const void* out, const float* aux,
                               const int32_t* argmax,
                               const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                               const int32_t* rowptr,
                               const float* ew_t, const float* eu, const float* ev,
                               const void* efull, const int32_t* eid_t, const int32_t* geid_t,
                               void* grad_x, void* grad_efull, float* grad_uv,
                               float* workspace, int64_t workspace_floats,
                               int64_t N, int64_t d, int dtype, int msg, int edge_mode, int edge_rank,
                               int aggr, int learn_t, float t, float p, const float* t_dev, const float* p_dev,
                               float eps, int add_root, int accumulate_efull, const mlgnn_hub_t* hub,
                               const void* grad_shifted, const int32_t* shift_flag, const mlgnn_ln_fold_t* ln,
                               void* stream);

/*
 * Backward prologue of the `power` aggregator (models/gcn_lib/sparse/torch_message.py:66-76): what
 * mlgnn_csr_aggregate_bwd expects as grad_out for MLGNN_AGGR_POWER,
 *     q = grad_out * mu_c^(1/p - 1) * [1e-7 <= mu <= 10] / max(deg, 1),   mu = aux of the forward, mu_c = clamp(mu, 1e-7, 10),
 * and, when grad_p is non-NULL (learnable p; needs out and aux2 of the forward and the workspace),
 *     grad_p[0] = sum grad_out * out * (-ln(mu_c) / p^2 + [in range] * aux2 / (p * mu_c)),
 * in one streaming pass (fixed-order partial sums).  p: immediate, or read from p_dev (device, no host sync).
 * grad_out, out, q: [N,d] in `dtype`; mu, aux2: [N,d] fp32; rowptr: by-destination row pointer (in-degree).
 */
int64_t mlgnn_power_bwd_prologue_workspace_floats(void);
int mlgnn_power_bwd_prologue(const void* grad_out, const float* mu, const int32_t* rowptr, const void* out,
                             const float* aux2, float p, const float* p_dev, void* q, float* grad_p,
                             float* workspace, int64_t workspace_floats, int64_t N, int64_t d, int dtype,
                             void* stream);

/*
 * Gene -> pathway learnable-projection pooling.
 * Replaces: models/multilevel_gnn.py:212-239 (advanced-index gather, repeat, mul, permute,
 * Tensor.scatter_reduce('sum')):
 *   out[b,c,s,k] = sum_{g: raw_indice[b,g]=s} x[b*NN + match[b,g], c] * [match>=0] * w[g,k]
 *
 * Members are addressed by their flat index f = b*G + g (M = B*G of them):
 *   seg_ptr [n_segments+1], seg_mem [M]   members grouped by (batch, segment), n_segments = B*S
 *   mem_row [M]   x row of the member (b*NN + match) or -1 when absent
 *   mem_seg [M]   (batch, segment) id of the member
 *   node_ptr [n_rows+1], node_mem [.]     present members grouped by x row
 *   x [n_rows, C], w [G, K] (K <= 4), out_t / gout_t [n_segments, K, C] (channel-contiguous)
 * n_groups = 0: that layout ([B, S, K, C]; segs_per_sample unused).  n_groups = NG > 0 with S = segs_per_sample a multiple
 * of NG and n_segments a multiple of S: the "pooled" layout [B, NG * K, S / NG, C] -- segment s = p * NG + o of a sample
 * goes to row ((b * NG + o) * K + k) * (S / NG) + p: the [B * NG * K, 146, C] batch of pathway graphs the DiffPool
 * levels consume (models/vae.py:238-243 `x.permute(0, 3, 2, 1).reshape(-1, 146, C)` on the [B, C, 146, NG * K] view),
 * written directly instead of through a transposing copy of the result and of its gradient.
 */
int mlgnn_segment_project_fwd(const void* x, const float* w, const int32_t* seg_ptr,
                              const int32_t* seg_mem, const int32_t* mem_row, void* out_t,
                              int64_t n_segments, int64_t C, int64_t G, int64_t K, int64_t segs_per_sample,
                              int64_t n_groups, int dtype, void* stream);

/*
 * Backward of the above.  grad_x [n_rows, C] (NULL to skip) and gw_partial [M, K] (NULL to skip):
 * gw_partial[f,k] = <x[mem_row[f],:], gout_t[mem_seg[f],k,:]>; the caller sums it over the batch
 * index to obtain d loss / d w [G,K].
 */
int mlgnn_segment_project_bwd(const void* gout_t, const void* x, const float* w,
                              const int32_t* seg_ptr, const int32_t* seg_mem,
                              const int32_t* mem_row, const int32_t* mem_seg,
                              const int32_t* node_ptr, const int32_t* node_mem,
                              void* grad_x, float* gw_partial,
                              int64_t n_segments, int64_t n_rows, int64_t C, int64_t G, int64_t K,
                              int64_t segs_per_sample, int64_t n_groups, int dtype, void* stream);

/*
 * Fused LayerNorm (+ ReLU) over [rows, d]: fp32 (d <= 256 with d % 4 == 0, or d <= 512 with d % 8 == 0) or bf16 storage with fp32 statistics and
 * arithmetic (MLGNN_DTYPE_BF16: x, out, grad_out, grad_extra, grad_x are bf16, d <= 512, d % 8 == 0; gamma, beta,
 * mean, rstd and the parameter gradients stay fp32).
 * Replaces: norm_layer('layer') followed by act_layer('relu') as chained by MLP
 * (models/gcn_lib/sparse/torch_nn.py:27-38,54-75) and by the res+ block (models/deepergcn.py:236-241).
 *   out = relu?( (x - mean) * rstd * gamma + beta ),  rstd = 1/sqrt(var_biased + eps)
 * mean / rstd [rows] are saved for the backward, which recomputes the ReLU mask from x.  Backward with
 * mean = NULL: x is already the normalised activation (x - mean) * rstd (mlgnn_tallgemm_nt ln_mode 1).
 * grad_gamma_beta [2,d]; workspace: mlgnn_layernorm_bwd_workspace_floats(rows, d, dtype) floats.
 * grad_extra [rows,d] or NULL: a gradient that reaches x on another branch (the identity branch of the
 * res+ block, deepergcn.py:241), added into grad_x in the same pass.
 * keep_mask [rows,d] bytes or NULL: the dropout that follows norm + activation in the res+ block (deepergcn.py:239-240,
 * 246-247) in the same pass -- out = act(norm(x)) * (keep ? keep_scale : 0); the backward scales grad_out alike.
 * row_max [rows] or NULL (both directions): max |.| per row of out / grad_x, handed to the Linear that
 * consumes it (mlgnn_tallgemm_nt) so that it does not have to read its operand twice.
 */
int64_t mlgnn_layernorm_bwd_workspace_floats(int64_t rows, int64_t d, int dtype);
int mlgnn_layernorm_act_fwd(const void* x, const float* gamma, const float* beta, void* out,
                            float* mean, float* rstd, float* row_max, const uint8_t* keep_mask, float keep_scale,
                            int64_t rows, int64_t d, float eps,
                            int relu, int dtype, void* stream);
int mlgnn_layernorm_act_bwd(const void* grad_out, const void* x, const float* gamma,
                            const float* beta, const float* mean, const float* rstd,
                            const void* grad_extra, void* grad_x, float* row_max, float* grad_gamma_beta,
                            float* workspace,
                            int64_t workspace_floats, const uint8_t* keep_mask, float keep_scale,
                            int64_t rows, int64_t d, int relu,
                            int dtype, void* stream);

/*
 * Weight + bias gradient of y = x W^T + b over a tall activation matrix (N >> M, K):
 *   grad_w_b[0 : M*K]     = grad_out^T x   ([M,K] row-major, the layout of nn.Linear.weight)
 *   grad_w_b[M*K : M*K+M] = column sums of grad_out
 * Replaces: the autograd of nn.Linear inside MLP (models/gcn_lib/sparse/torch_nn.py:54-75).
 * grad_out [N,M], x [N,K] fp32; ceil(M/32)*ceil(K/32) <= 32 tiles, otherwise MLGNN_E_SHAPE
 * (the caller then uses a library GEMM).  workspace: mlgnn_linear_wgrad_workspace_floats floats.
 * x_gamma, x_beta [K] or NULL: x is a layer-normalised activation (mlgnn_tallgemm_nt ln_mode 1) and the Linear's
 * real input was relu(x_gamma x + x_beta): applied to the operand as it is loaded.
 * grad_out_row_max, x_row_max [N] or NULL (fp32 only): max |row| of grad_out and of x (of the activated x when x_gamma
 * is set; upper bounds within a few binades will do) -- the row_max side outputs of the kernels that produced the
 * operands.  Both given: the operands are scaled by exact powers of two derived from their global maxima and split in
 * two fp16 terms (three MFMAs per product instead of six; 7e-7 per product at worst for elements within 2^-17 of the
 * maximum).
 * MLGNN_DTYPE_BF16 (BASELINE configs[4]): grad_out, x are bf16, grad_w_b stays fp32 (bf16 products accumulated in
 * fp32, per-slab partials summed in fixed order); M % 64 == 0, K % 128 == 0, x_gamma / x_beta must be NULL.
 */
int64_t mlgnn_linear_wgrad_workspace_floats(int64_t N, int64_t M, int64_t K, int dtype);
int mlgnn_linear_wgrad(const void* grad_out, const void* x, const float* x_gamma, const float* x_beta,
                       const float* grad_out_row_max, const float* x_row_max, float* grad_w_b, float* workspace,
                       int64_t workspace_floats, int64_t N, int64_t M, int64_t K, int dtype,
                       void* stream);

/*
 * DiffPool soft-assignment contraction, fused forward (one workgroup per batch element, fp32 MFMA).
 * Replaces: torch_geometric.nn.dense_diff_pool as called from models/diff_pooling.py:64:
 *   S = softmax(s_logits, -1);  x_out = S^T z;  adj_out = S^T adj S;
 *   partial[b] = { sum (adj - S S^T)^2 , sum_n sum_k -S log(S + 1e-15) }   (caller: sqrt / numel, mean)
 * z [B,N,C], s_logits [B,N,K], adj [N,N] (adj_batched = 0) or [B,N,N]; s_out [B,N,K] (saved for
 * the backward), x_out [B,K,C], adj_out [B,K,K], partial [B,2].
 * Supported while N <= 160, K <= 48, C <= 64 (mlgnn_diffpool_fwd_supported); MLGNN_E_SHAPE otherwise.
 */
int mlgnn_diffpool_fwd_supported(int64_t N, int64_t K, int64_t C);
int mlgnn_diffpool_fwd(const void* z, const void* adj, const void* s_logits, void* s_out,
                       void* x_out, void* adj_out, float* partial, int64_t B, int64_t N,
                       int64_t K, int64_t C, int adj_batched, int dtype, void* stream);

/*
 * Backward of mlgnn_diffpool_fwd, one fused MFMA launch.  s_softmax = the forward's s_out;
 * grad_x [B,K,C], grad_adj_out [B,K,K] = cotangents of x_out / adj_out;
 * coef (device, float[2]) = { grad_link / (numel(adj) * ||adj - S S^T||_F),  grad_ent / (B*N) };
 * outputs grad_z [B,N,C], grad_s [B,N,K] (w.r.t. the logits) and, when non-NULL, grad_adj [B,N,N]
 * per batch element (the caller sums over the batch for a shared adjacency).
 */
int mlgnn_diffpool_bwd(const void* z, const void* adj, const void* s_softmax, const void* grad_x,
                       const void* grad_adj_out, const float* coef, void* grad_z, void* grad_s,
                       void* grad_adj, int64_t B, int64_t N, int64_t K, int64_t C,
                       int adj_batched, int dtype, void* stream);

/*
 * DenseSAGEConv over a batch of small pooled graphs, one fused MFMA launch each way.
 * Replaces: torch_geometric DenseSAGEConv as used by SAGEConvolutions / DiffPoolLayer
 * (models/diff_pooling.py:24-32,45-53,58-65):
 *   y = normalize(lin_rel(A x / clamp(rowsum A, 1)) + lin_root(x), p=2, dim=-1)
 * x [B,n,C], adj [n,n] (adj_batched = 0) or [B,n,n], w_rel / w_root [O,C], bias [O] or NULL,
 * y [B,n,O], rinv [B,n] = 1 / max(||out||_2, 1e-12) (saved for the backward; 1 when normalize = 0).
 * Supported while n <= 160, C <= 128, O <= 64, and n <= 48 when the adjacency gradient is wanted
 * (mlgnn_dense_sage_supported); MLGNN_E_SHAPE otherwise.
 * Backward: grad_x [B,n,C]; grad_adj [B,n,n] per batch element or NULL; grad_w = [grad_w_rel (O*C) |
 * grad_w_root (O*C) | grad_bias (O)]; workspace: mlgnn_dense_sage_bwd_workspace_floats(B,C,O) floats.
 */
int mlgnn_dense_sage_supported(int64_t n, int64_t C, int64_t O, int need_grad_adj);
int64_t mlgnn_dense_sage_bwd_workspace_floats(int64_t B, int64_t C, int64_t O);
int mlgnn_dense_sage_fwd(const void* x, const void* adj, const void* w_rel, const void* w_root,
                         const float* bias, void* y, float* rinv, int64_t B, int64_t n, int64_t C,
                         int64_t O, int adj_batched, int normalize, int dtype, void* stream);
int mlgnn_dense_sage_bwd(const void* grad_y, const void* y, const float* rinv, const void* x,
                         const void* adj, const void* w_rel, const void* w_root, void* grad_x,
                         void* grad_adj, float* grad_w, float* workspace, int64_t workspace_floats,
                         int64_t B, int64_t n, int64_t C, int64_t O, int adj_batched, int normalize,
                         int dtype, void* stream);

/*
 * COO -> CSR (by destination) + transposed CSR (by source), on the device, stable in COO order.
 * Replaces: the per-layer gather/scatter index handling of MessagePassing.propagate
 * (models/gcn_lib/sparse/torch_vertex.py:82,277) by one topology sort per batch.
 * edge_index [2,E] int64 row-major (row 0 = source j, row 1 = destination i), node ids in [0,N).
 * Outputs (int32): rowptr [N+1], col [E], eid [E], rowptr_t [N+1], col_t [E], pos_t [E], eid_t [E].
 * bad_ids (nullable, int32[1]): number of node ids outside [0,N); such ids are clamped so that no
 * later kernel reads out of bounds -- the caller decides when to look at the counter.
 * workspace: mlgnn_coo_to_csr_workspace_bytes(N, E) bytes.
 */
int64_t mlgnn_coo_to_csr_workspace_bytes(int64_t N, int64_t E);
int mlgnn_coo_to_csr(const int64_t* edge_index, int64_t E, int64_t N,
                     int32_t* rowptr, int32_t* col, int32_t* eid,
                     int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t,
                     int32_t* bad_ids, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Raw edge attributes in COO order -> the two CSR orders the aggregation kernels read (`ew`, `ew_t`):
 * by_dst[e] = attr[eid[e]], by_src[e] = attr[eid_t[e]], each row zero padded from r to `width` columns
 * (width <= 8).  attr [E0, r] fp32 with a row stride of row_stride floats (a column view is fine).
 * Replaces: the edge_attr index_select of MessagePassing.propagate (torch_vertex.py:82) -- once per batch.
 */
int mlgnn_edge_table_to_csr(const float* attr, int64_t row_stride, int64_t r, int64_t width,
                            const int32_t* eid, const int32_t* eid_t, float* by_dst, float* by_src,
                            int64_t E, void* stream);

/*
 * MsgNorm fused with GENConv's root add:  h = x + normalize(m, p=2, dim=1) * ||x||_2 * scale[0]
 * Replaces: MsgNorm.forward (models/gcn_lib/sparse/torch_message.py:175-179) + h = x + m
 * (models/gcn_lib/sparse/torch_vertex.py:86-89).  x, m, h [rows, d]: MLGNN_DTYPE_F32 with d <= 256, d % 4 == 0, or
 * MLGNN_DTYPE_BF16 storage (fp32 arithmetic) with d <= 512, d % 8 == 0; scale: device pointer to the (learnable)
 * fp32 scalar.  Backward returns grad_x, grad_m and
 * grad_scale[1]; workspace: mlgnn_msgnorm_bwd_workspace_floats(rows, d) floats.
 */
int64_t mlgnn_msgnorm_bwd_workspace_floats(int64_t rows, int64_t d);
int mlgnn_msgnorm_add_fwd(const void* x, const void* m, const float* scale, void* h,
                          int64_t rows, int64_t d, int dtype, void* stream);
int mlgnn_msgnorm_add_bwd(const void* grad_h, const void* x, const void* m, const float* scale,
                          void* grad_x, void* grad_m, float* grad_scale, float* workspace,
                          int64_t workspace_floats, int64_t rows, int64_t d, int dtype, void* stream);

/*
 * Per-graph readout: out[b,:] = sum | mean | max over the rows [ptr[b], ptr[b+1]) of x.
 * Replaces: global_{add,mean,max}_pool (models/deepergcn.py:148-155,319).  kind: 0 sum, 1 mean, 2 max;
 * an empty graph gives 0; max also returns argmax [B,d] (row index, -1 when empty; first maximal row).
 * x [N,d] fp32 with d % 4 == 0; workspace: mlgnn_segment_pool_workspace_bytes(B, d) bytes.
 */
int64_t mlgnn_segment_pool_workspace_bytes(int64_t B, int64_t d);
int mlgnn_segment_pool_fwd(const void* x, const int32_t* ptr, void* out, int32_t* argmax,
                           void* workspace, int64_t workspace_bytes, int64_t B, int64_t d,
                           int kind, int dtype, void* stream);

/*
 * Tall-skinny fp32 GEMM on the fp16 matrix cores with power-of-two scaled hi/lo split precision
 * (3 MFMAs per product, fp32 accumulate):
 *   c[N,J] = a[N,R] * bt[J,R]^T (+ bias[J]) (+ residual[N,J]),  N >> R,J;  R in {16,32,64,128,256},
 *   J in {32,64,128,256}, R*J*4 <= 128 KiB;  residual (nullable) = the identity branch of the res+ block
 *   (h = conv(...) + h, deepergcn.py:241) folded into the epilogue of the conv's last Linear; J <= 128
 * Replaces: forward and input gradient of the nn.Linear layers of MLP
 * (models/gcn_lib/sparse/torch_nn.py:54-75).  Relative error per product <= 3*2^-22 (see csrc/tallgemm.hip).
 * row_max [N] or NULL: max_k |a[i][k]| per row when the producer of `a` already knows it (any upper bound
 * within a factor of 2 of the true maximum keeps full accuracy); NULL: the kernel streams `a` twice.
 * ln_mode (R, J in {64,128,256}): the LayerNorm + ReLU that sits between the two Linears of MLP
 * (torch_nn.py:54-75) without a pass of its own --
 *   1: c receives the layer-normalised result xhat = (v - mean(v)) * rstd (no affine), rstd_out [N] = 1/sigma,
 *      row_max_out [N] = max_j relu(gamma[j] xhat + beta[j])   (gamma, beta [J]; residual must be NULL);
 *   2: a is such an xhat: relu(gamma[k] a + beta[k]) is applied as it is loaded (gamma, beta [R]; pass the
 *      producer's row_max_out as row_max);
 *   0: neither (gamma, beta, rstd_out, row_max_out ignored).
 * bt_transposed non-zero (fp32 only): bt is stored [R,J] -- a Linear's own weight used for its input gradient -- and is
 * read with swapped indices while the weight image is built, instead of being copied first.
 * workspace: mlgnn_tallgemm_workspace_bytes(R, J, dtype) bytes (weight image in MFMA fragment order).
 * MLGNN_DTYPE_BF16 (BASELINE configs[4]): a, bt, residual, c are bf16 (bias stays fp32), one bf16 MFMA per product
 * with fp32 accumulation and a single rounding at the store; R % 16 == 0, J % 32 == 0 (R <= 1024, J <= 4096, the
 * weight is cut into column slices of <= 128 KiB that each stream `a` once); ln_mode must be 0, row_max is ignored.
 */
int mlgnn_tallgemm_supported(int64_t N, int64_t R, int64_t J, int dtype);
int64_t mlgnn_tallgemm_workspace_bytes(int64_t R, int64_t J, int dtype);
int mlgnn_tallgemm_nt(const void* a, const void* bt, int bt_transposed, const float* bias, const void* residual,
                      const float* row_max, int ln_mode, const float* gamma, const float* beta, float ln_eps,
                      float* rstd_out, float* row_max_out, void* c, void* workspace, int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, int dtype, void* stream);

/*
 * Input gradient of a Linear whose INPUT is the output of a softmax aggregation (fp32): the plain product
 *     c [N,J] = a [N,R] * bt (mlgnn_tallgemm_nt, ln_mode 0, no bias / residual)
 * plus, from the same epilogue, what that aggregation's backward gathers per edge (csrc/aggregate_bwd.hip):
 *     grad_shifted[i][j] = c[i][j] * 2^(-lse[i][j])   (0 for nodes without incoming edges: rowptr [N+1] by destination),
 *     shift_flag[0] = 1 when some |lse| > 60 (the consumer then takes its two-row path), else 0   (4 x int32, zeroed here)
 * Replaces: the streaming pre-pass of the softmax backward (read grad_out and lse, write grad_shifted) -- a load and a
 * store of rows this kernel already holds.  Hand both to mlgnn_csr_aggregate_bwd.  J in {64, 128}, R in {64, 128, 256}.
 */
int mlgnn_tallgemm_nt_shift_supported(int64_t N, int64_t R, int64_t J);
int mlgnn_tallgemm_nt_shift(const float* a, const float* bt, int bt_transposed, const float* row_max, const float* lse,
                            const int32_t* rowptr, float* c, float* grad_shifted, int32_t* shift_flag, void* workspace,
                            int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, void* stream);

/*
 * The same for bf16 storage (BASELINE configs[4]): c [N,J] = a [N,R] * bt [J,R]^T (bf16, fp32 accumulation) and
 * grad_shifted = c * 2^(-lse) (bf16, from the rounded c: bitwise what the streaming pre-pass would write), lse [N,J]
 * fp32 in log2 units (0 for nodes without incoming edges, as mlgnn_csr_aggregate_fwd writes it); shift_flag as above.
 * Shapes: those of mlgnn_tallgemm_nt with MLGNN_DTYPE_BF16; workspace: mlgnn_tallgemm_workspace_bytes(R, J, bf16).
 */
int mlgnn_tallgemm_bf16_shift_supported(int64_t N, int64_t R, int64_t J);
int mlgnn_tallgemm_bf16_shift(const void* a, const void* bt, const float* lse, void* c, void* grad_shifted,
                              int32_t* shift_flag, void* workspace, int64_t workspace_bytes, int64_t N, int64_t R,
                              int64_t J, void* stream);

/*
 * The MLP's second Linear with the NEXT block's pre-conv LayerNorm (+ ReLU) in its epilogue (fp32):
 *     c [N,J] = relu(gamma xhat + beta) [N,R] * bt[J,R]^T (+ bias) (+ residual)          (= mlgnn_tallgemm_nt, ln_mode 2)
 *     y [N,J] = relu?(post_gamma (c - mean_row(c)) rstd_row + post_beta),   post_mean / post_rstd [N] = mean, 1/sigma
 * Replaces: the last Linear of GENConv's MLP (torch_nn.py:54-75) + the residual add and the `norm -> relu` the res+
 * block applies before the next conv (models/deepergcn.py:236-241; the final norm of :247 with post_relu = 0): the
 * separate LayerNorm pass (read c, write y) becomes one more store of rows this kernel already holds.  J in {64, 128}
 * (a wave holds whole result rows), R in {64, 128, 256}, R * J * 4 <= 128 KiB.  Two-pass statistics, biased variance,
 * eps inside the square root, like nn.LayerNorm.  workspace: mlgnn_tallgemm_workspace_bytes(R, J, fp32) bytes.
 */
int mlgnn_tallgemm_lnin_postln_supported(int64_t N, int64_t R, int64_t J);
int mlgnn_tallgemm_lnin_postln(const float* xhat, const float* bt, const float* bias, const float* residual,
                               const float* row_max, const float* gamma, const float* beta, const float* post_gamma,
                               const float* post_beta, float post_eps, int post_relu, float* c, float* y,
                               float* post_mean, float* post_rstd, void* workspace, int64_t workspace_bytes,
                               int64_t N, int64_t R, int64_t J, void* stream);

/*
 * Backward of a Linear layer y = x W^T + b over a tall x in ONE pass over the cotangent (fp32):
 *     dx [N,K] = grad_out [N,M] * W [M,K],   grad_w_b = (grad_out^T x' [M,K], column sums of grad_out [M])
 * and the epilogue the MLP of a GENConv layer needs behind dx (models/gcn_lib/sparse/torch_nn.py:54-75,
 * torch_vertex.py:35): the autograd of nn.Linear would stream grad_out and x twice (input and weight gradient).
 *   epilogue 0 (LayerNorm backward; M = 128, K = 256): x = xhat, the hidden activation stored normalised with its rstd
 *     [N] (mlgnn_tallgemm_nt ln_mode 1); x' = relu(gamma xhat + beta); dx is taken through ReLU + LayerNorm backward,
 *         gy = dx [gamma xhat + beta > 0],  g = gamma gy,  dx <- rstd (g - mean_row(g) - xhat mean_row(g xhat)),
 *     and grad_w_b carries two more rows behind the bias gradient: sum_rows gy xhat [K], sum_rows gy [K]
 *     (M K + M + 2 K floats);
 *   epilogue 1 (plain; M = 256, K = 128): x' = x, dx as it is;
 *   epilogue 2 (shift; M = 256, K = 128): as 1, and grad_shifted [N,K] = dx * 2^(-lse) for the softmax aggregation
 *     that produced x (lse [N,K] in log2 units, 0 for nodes without incoming edges as mlgnn_csr_aggregate_fwd writes
 *     it); *shift_flag (int32[4], zeroed here) is raised when some |lse| > 60: what mlgnn_csr_aggregate_bwd takes as
 *     grad_shifted / shift_flag.
 * Arithmetic: scaled two-way fp16 split on the fp16 matrix cores (3 * 2^-22 per product for everything within 2^-16 of
 * an operand's largest magnitude, fp32 accumulation), scales from grad_out_max (row maxima [N], or 256 partial maxima
 * when grad_out_max_is_parts: the dx_max_parts of the call that produced grad_out) and x_row_max [N] (max |x'| per row).
 * dx_max_parts [256] or NULL: per-workgroup max |dx|.  N * max(M, K) * 4 < 4 GiB.  Deterministic.
 * workspace: mlgnn_linear_bwd_workspace_floats floats.
 */
int mlgnn_linear_bwd_supported(int64_t N, int64_t M, int64_t K, int epilogue);
int64_t mlgnn_linear_bwd_workspace_floats(int64_t N, int64_t M, int64_t K, int epilogue);
int mlgnn_linear_bwd(const float* grad_out, const float* w, const float* x, const float* grad_out_max,
                     int grad_out_max_is_parts, const float* x_row_max, int epilogue, const float* rstd,
                     const float* gamma, const float* beta, const float* lse, float* dx, float* grad_shifted,
                     int32_t* shift_flag, float* grad_w_b, float* dx_max_parts, float* workspace,
                     int64_t workspace_floats, int64_t N, int64_t M, int64_t K, void* stream);

/*
 * The MLP's second Linear run backwards with the ReLU + LayerNorm backward in the epilogue (fp32):
 *     dA = grad_out [N,R] * W [R,J] (w_transposed != 0: W is the Linear's own weight [R,J]; 0: its transpose [J,R]),
 *     gy = dA [gamma xhat + beta > 0],  g = gamma gy,
 *     grad_h [N,J] = rstd (g - mean_row(g) - xhat mean_row(g xhat)),
 *     grad_gamma_beta [2,J] = (sum_rows gy xhat, sum_rows gy),  row_max_out [N] = max_c |grad_h|
 * Replaces: the autograd of Linear <- ReLU <- LayerNorm inside MLP (models/gcn_lib/sparse/torch_nn.py:54-75) for a
 * hidden activation stored normalised (xhat, rstd: mlgnn_tallgemm_nt ln_mode 1) -- the product dA never reaches memory.
 * R, J in {64, 128, 256} with R * J * 4 <= 128 KiB; any N <= INT32_MAX (operands past 4 GiB are walked in row slabs
 * inside the call, as in mlgnn_linear_bwd / mlgnn_linear_wgrad); row_max [N] or NULL: max |grad_out[i]|.
 * workspace: mlgnn_tallgemm_lnbwd_workspace_bytes(R, J) bytes.  Deterministic (fixed-order partial sums).
 */
int mlgnn_tallgemm_lnbwd_supported(int64_t N, int64_t R, int64_t J);
int64_t mlgnn_tallgemm_lnbwd_workspace_bytes(int64_t R, int64_t J);
int mlgnn_tallgemm_lnbwd(const float* grad_out, const float* w, int w_transposed, const float* row_max,
                         const float* xhat, const float* rstd, const float* gamma, const float* beta,
                         float* grad_h, float* row_max_out, float* grad_gamma_beta, void* workspace,
                         int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, void* stream);

/*
 * Gradient of an embedding lookup e = table[idx] for a dense cotangent:
 *   grad_table[t,:] = sum_{e: idx[e] = t} grad_e[e,:]       (grad_e [E,d], grad_table [T,d], fp32, d % 4 == 0)
 * Replaces: the autograd of DeeperGCN's edge-type embedding, nn.Embedding(pathway_edge_num, hidden) applied to every
 * edge (models/deepergcn.py:103-104,189-190,213).  perm [E] int32: the edge ids sorted (stably) by idx;
 * rowptr [T+1] int32: row t owns perm[rowptr[t] .. rowptr[t+1]).  Deterministic (fixed summation order per row).
 */
int mlgnn_embedding_bwd(const float* grad_e, const int32_t* perm, const int32_t* rowptr, float* grad_table,
                        int64_t T, int64_t d, int dtype, void* stream);

/*
 * Table gradient through a fixed-point accumulator -- the max aggregator's way to the gradient of a table edge term
 * (deepergcn.py:103-104,189-190: nn.Embedding(pathway_edge_num, hidden) read by every GENConv layer; the reference's
 * DEFAULT flags gcn_aggr=max, global_edge=onehot):
 *   accumulator: mlgnn_table_grad_bytes(T, d) bytes, 16-byte aligned, ZEROED once by the caller;
 *   mlgnn_table_grad_begin(grad_out [rows,d] fp32 -- the cotangent mlgnn_csr_aggregate_bwd is about to receive)
 *     records max |grad_out| (and a non-finite flag) on the device: the power-of-two scale of the sums;
 *   mlgnn_csr_aggregate_bwd(..., grad_efull = accumulator, geid_t = table row per edge, accumulate_efull = 2);
 *   mlgnn_table_grad_finish: grad_table [T,d] = (accumulate ? grad_table : 0) + sums / scale, accumulator cleared for
 *     the next layer.  Resolution 2^-(61 - log2(rows)) of max |grad_out|; a non-finite cotangent gives an all-NaN table
 *     gradient (the per-edge path would confine it to the rows it touches).
 */
int64_t mlgnn_table_grad_bytes(int64_t T, int64_t d);

/*
 * The same gradient from the DESTINATION side, without atomics and without a per-edge buffer (the default for the
 * reference's default flags).  The forward's argmax [N,d] names the winning edge of (i, c) by its by-destination position,
 * or -1 where no gradient flows (no incoming edge, or the winner's relu is on its flat side), so
 *   grad_table[t][c] (+)= sum over { i : rows_by_dst[argmax[i][c]] = t } of grad_out[i][c]
 * is one streaming pass over grad_out and argmax (N d 8 bytes); rows_by_dst [E] int32: table row of every edge in
 * by-destination order.  The aggregation backward then runs with accumulate_efull = 3 and writes nothing per edge.
 * fp32, d % 4 == 0, d <= 1024, T <= 36 (mlgnn_max_table_grad_supported); workspace:
 * mlgnn_max_table_grad_workspace_floats(N, d, T) floats (per-workgroup partial tables, added in workgroup order:
 * bitwise reproducible).  accumulate: grad_table += (the layers of one backward that share the table).
 *
 * mlgnn_max_table_grad_by_type: the same sum for a table of ANY size (nn.Embedding(pathway_edge_num, hidden) has one row
 * per KEGG membership: tens of thousands) -- one wavefront per table row walks that row's edges and gathers the
 * cotangent and argmax rows of their destinations.  pos_sorted [E] int32: the by-destination edge positions sorted
 * (stably) by table row, dst_sorted [E]: the destination node of each, rowptr [T+1]: row t owns
 * [rowptr[t], rowptr[t+1]).  fp32, d % 4 == 0; no workspace; fixed order (bitwise reproducible).
 * slots (or NULL) + rel_sorted [E] (position of each edge INSIDE its destination row): the one-byte winner slots the
 * MAX backward of the same layer left in its workspace -- workspace + mlgnn_csr_aggregate_bwd_slots_offset_floats(N, d,
 * edge_rank) floats: {int32 flag (non-zero: some row is longer than 254 edges, slots unusable), 12 bytes, slots [N,d]};
 * call after that backward on the same stream.  The kernel then gathers 1 instead of 4 bytes of winner information per
 * channel (flag set: it reads argmax as without slots).  Same results.
 */
int mlgnn_max_table_grad_supported(int64_t N, int64_t d, int64_t T);
int64_t mlgnn_max_table_grad_workspace_floats(int64_t N, int64_t d, int64_t T);
int mlgnn_max_table_grad(const float* grad_out, const int32_t* argmax, const int32_t* rows_by_dst, float* grad_table,
                         float* workspace, int64_t workspace_floats, int64_t N, int64_t d, int64_t T, int accumulate,
                         void* stream);
int mlgnn_max_table_grad_by_type(const float* grad_out, const int32_t* argmax, const int32_t* dst_sorted,
                                 const int32_t* pos_sorted, const int32_t* rel_sorted, const int32_t* rowptr,
                                 const void* slots, float* grad_table, int64_t N, int64_t d, int64_t T, int accumulate,
                                 void* stream);
int64_t mlgnn_csr_aggregate_bwd_slots_offset_floats(int64_t N, int64_t d, int edge_rank);
int mlgnn_table_grad_begin(const float* grad_out, int64_t rows, int64_t d, void* accumulator, void* stream);

/*
 * Backward of the MAX aggregator (the reference's default, opt.py:144; torch_message.py:46-47) from compact winner lists
 * (csrc/max_sparse.hip) -- for graphs whose rows hold at most 256 edges in BOTH directions (the caller knows: longer
 * rows take mlgnn_csr_aggregate_bwd), fp32, 32 <= d <= 256, d % 4 == 0 (mlgnn_max_sparse_supported):
 *   mlgnn_max_winners       by destination row: records [mlgnn_max_sparse_records(N, d, E)] x 8 bytes = {cotangent value,
 *                           channel} of row i sorted by winning edge (row i starts at even(i (d + 2) + rowptr[i]), every
 *                           run on an even record), meta [E] x 8 bytes = {first record, count} of every edge's run
 *                           (by-destination position).  argmax as written by the forward (-1: no gradient).
 *   mlgnn_max_sparse_bwd    grad_x[j][ch] = sum of the runs of j's outgoing edges (+ root[j] when given: the identity
 *                           branch of h = x + m; pass grad_out) -- rowptr_t / pos_t: the by-source CSR.
 *   mlgnn_max_sparse_table_grad   grad_table[t] (+)= the same sum over the edges that read table row t: pos_sorted /
 *                           rowptr as for mlgnn_max_table_grad_by_type.
 * About d / in-degree (value, channel) pairs are read per edge instead of the destination's whole cotangent and winner
 * rows.  No atomics between workgroups; bitwise reproducible.  records and meta: 16-byte aligned.
 */
int mlgnn_max_sparse_supported(int64_t N, int64_t d);
int64_t mlgnn_max_sparse_records(int64_t N, int64_t d, int64_t E);
int mlgnn_max_winners(const float* grad_out, const int32_t* argmax, const int32_t* rowptr, void* records, void* meta,
                      int64_t N, int64_t d, void* stream);
int mlgnn_max_sparse_bwd(const void* records, const void* meta, const int32_t* rowptr_t, const int32_t* pos_t,
                         const float* root, float* grad_x, int64_t N, int64_t d, void* stream);
int mlgnn_max_sparse_table_grad(const void* records, const void* meta, const int32_t* pos_sorted, const int32_t* rowptr,
                                float* grad_table, int64_t N, int64_t d, int64_t T, int accumulate, void* stream);
int mlgnn_table_grad_finish(void* accumulator, float* grad_table, int64_t T, int64_t d, int accumulate, void* stream);

/*
 * Large bf16 GEMM with fp32 accumulation on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), both operands with the
 * contraction index contiguous:
 *     C[M,N] = sum_{s < nseg} A_s[M,K_s] * B_s[N,K_s]^T  (+ alpha * aux[M,N])
 * Replaces: the dense products of torch_geometric's dense_diff_pool (s^T x, s^T adj s, s s^T) as called from
 * DiffPoolLayer.forward (models/diff_pooling.py:59-65) at sizes past the fused small-graph kernel (BASELINE
 * configs[4]: 4096 pooled nodes, 1024 clusters), and their gradients.
 *   a[s], b[s]   device pointers to bf16 matrices, leading dimensions lda[s], ldb[s] (elements, multiples of 8,
 *                pointers 16-byte aligned); k[s] % 64 == 0; 1 <= nseg <= 4;  M % 128 == 0, N % 128 == 0
 *   splits, slab slab = NULL (splits must be 1): the epilogue below runs in the kernel.  slab != NULL: the contraction
 *                range is cut into `splits` >= 1 parts whose fp32 results go to slab [splits][M][N] (nothing else is
 *                written); sum them in a fixed order afterwards -- there are no atomics anywhere
 *   c            [M,N] result, leading dimension ldc, c_dtype MLGNN_DTYPE_BF16 or MLGNN_DTYPE_F32 (nullable)
 *   ct           [N,M] bf16 transposed copy of the result, leading dimension ldct (nullable)
 *   aux, alpha   optional [M,N] term added before the stores (aux_dtype as c_dtype)
 *   dot          optional [M,N] bf16: dot_partial[w] = sum over workgroup w's tile of dot[i][j] * (fp32 result, before
 *                alpha * aux); dot_partial has mlgnn_gemm_bf16_nt_workgroups(M, N, 1) entries
 * One workgroup per 128 x 128 tile (and split): mlgnn_gemm_bf16_nt_workgroups(M, N, splits) in all (0: bad shape).
 */
int mlgnn_gemm_bf16_nt_workgroups(int64_t M, int64_t N, int splits);
int mlgnn_gemm_bf16_nt(const void* const* a, const void* const* b, const int64_t* lda, const int64_t* ldb,
                       const int64_t* k, int nseg, int64_t M, int64_t N, int splits, float* slab,
                       void* c, int64_t ldc, int c_dtype, void* ct, int64_t ldct,
                       const void* aux, int64_t ldaux, int aux_dtype, float alpha,
                       const void* dot, int64_t lddot, float* dot_partial, void* stream);

/*
 * dense_diff_pool for LARGE pooled graphs on the bf16 matrix cores (BASELINE configs[4]: N = 4096 nodes, K = 1024
 * clusters, C = 256 channels); the shapes below are those of ONE pooled graph, a batch is described at the end.
 * Replaces: torch_geometric.nn.dense_diff_pool as called from DiffPoolLayer.forward (models/diff_pooling.py:59-65):
 *   S = softmax(s_logits, -1);  x_out = S^T z;  adj_out = S^T adj S;
 *   stats[0] = ||adj - S S^T||_F / numel(adj);  stats[1] = mean_n(sum_k -S log(S + 1e-15));  stats[2] = ||adj - S S^T||_F
 * z [N,C], adj [N,N] bf16; s_logits [N,K] fp32 or bf16 (logits_dtype); s_out [N,K] bf16 = the rounded softmax every
 * product uses (saved for the backward); x_out [K,C], adj_out [K,K], scal_out [2] = {stats[0], stats[1]} in out_dtype;
 * stats float[3] (device).
 * N, K, C multiples of 128 (mlgnn_diffpool_large_supported).  The link term is evaluated as
 * ||adj||^2 - 2 <S, adj S> + ||S^T S||^2 (exact identity, fp32 partial sums in a fixed order; csrc/diffpool_large.hip).
 * workspace: mlgnn_diffpool_large_workspace_bytes(N, K, C) bytes, 256-byte aligned; its first
 * mlgnn_diffpool_large_saved_bytes(N, K, C) bytes (T = adj S, its transpose, S^T, z^T, S^T S) must reach the backward
 * unchanged (`saved`).
 *
 * Backward: grad_x [K,C], grad_adj_out [K,K] (grad_dtype) = cotangents of x_out / adj_out; grad_link, grad_ent = the
 * scalar cotangents of stats[0] / stats[1] ON THE DEVICE (scalar_dtype; no host sync); stats = the forward's;
 * outputs grad_z [N,C], grad_logits [N,K] in logits_dtype; grad_adj [N,N] (logits_dtype) or NULL: the adjacency
 * gradient  S (dA' - cI) S^T + c adj,  c = grad_link / (numel ||adj - S S^T||)  -- two more products; the next pooling
 * level's adjacency is this level's adj_out (models/diff_pooling.py:116-127).
 * adj_symmetric non-zero promises adj = adj^T and saves the product adj^T S (one third of the backward).
 * workspace: mlgnn_diffpool_large_bwd_workspace_bytes(N, K, C, adj_symmetric) bytes.
 *
 * A batch: B >= 1 pooled graphs of one shape run as ONE grouped launch per step of the chain (grid.y = graph), not a
 * host loop: z [B,N,C], s_logits / s_out [B,N,K], x_out [B,K,C], adj_out [B,K,K]; adj [B,N,N] when adj_batched, else
 * ONE [N,N] adjacency shared by the batch (PyG broadcasts a 2-D adj).  stats / scal_out are the batch's scalars as the
 * reference computes them on a batched call: ONE Frobenius norm over all graphs divided by numel(adj) of the ARGUMENT
 * (N*N for a shared adjacency), the entropy averaged over all B*N nodes.  workspace (and `saved`): B consecutive blocks
 * of the per-graph size.  Backward: grad_x [B,K,C], grad_adj_out [B,K,K], grad_z [B,N,C], grad_logits [B,N,K];
 * grad_adj [B,N,N] -- one block per graph also for a shared adjacency, whose gradient is their sum (the caller's).
 */
int mlgnn_diffpool_large_supported(int64_t N, int64_t K, int64_t C);
int64_t mlgnn_diffpool_large_workspace_bytes(int64_t N, int64_t K, int64_t C);
int64_t mlgnn_diffpool_large_saved_bytes(int64_t N, int64_t K, int64_t C);
int mlgnn_diffpool_large_fwd(const void* z, const void* adj, const void* s_logits, int logits_dtype,
                             void* s_out, void* x_out, void* adj_out, void* scal_out, int out_dtype,
                             float* stats, void* workspace, int64_t workspace_bytes, int64_t N, int64_t K,
                             int64_t C, int64_t B, int adj_batched, void* stream);
int64_t mlgnn_diffpool_large_bwd_workspace_bytes(int64_t N, int64_t K, int64_t C, int adj_symmetric);
int mlgnn_diffpool_large_bwd(const void* z, const void* adj, const void* s_logits, int logits_dtype,
                             const void* s_soft, const void* saved, const void* grad_x,
                             const void* grad_adj_out, int grad_dtype, const void* grad_link,
                             const void* grad_ent, int scalar_dtype, const float* stats, void* grad_z,
                             void* grad_logits, void* grad_adj, int adj_symmetric, void* workspace,
                             int64_t workspace_bytes, int64_t N, int64_t K, int64_t C, int64_t B, int adj_batched,
                             void* stream);

/*
 * The same call (models/diff_pooling.py:59-65, dense_diff_pool) on fp32 tensors at the large sizes: the product chain of
 * mlgnn_diffpool_large_fwd / _bwd with every product as THREE bf16 terms on the matrix cores (x = x_hi + x_lo up to
 * 2^-17 |x|; hi*hi + hi*lo + lo*hi, fp32 accumulation): fp32-level accuracy, within 1e-4 of fp64 (tests).  All
 * tensors fp32: z [B,N,C], s_logits / s_out [B,N,K] (s_out = softmax), adj [B,N,N] (adj_batched) or ONE [N,N],
 * x_out [B,K,C], adj_out [B,K,K], scal_out [2] = {link, entropy}, stats float[3] as above.  A batch runs as grouped
 * launches with the reference's batch semantics for the scalars (see above).  Shapes: mlgnn_diffpool_large_supported.
 * workspace: B consecutive blocks of mlgnn_diffpool_large_f32_workspace_bytes(N, K, C) bytes (256-byte aligned); the first
 * mlgnn_diffpool_large_f32_saved_bytes(N, K, C) bytes of every block must reach the backward unchanged (`saved` = the
 * same pointer: the backward finds block b at saved + b * workspace_bytes(N, K, C)).
 * Backward: grad_x [B,K,C], grad_adj_out [B,K,K], grad_link / grad_ent (device scalars), outputs grad_z [B,N,C],
 * grad_logits [B,N,K], grad_adj [B,N,N] or NULL (one block per graph also for a shared adjacency: the caller sums).
 * workspace: B blocks of mlgnn_diffpool_large_f32_bwd_workspace_bytes(N, K, C, adj_symmetric) bytes.
 */
int64_t mlgnn_diffpool_large_f32_workspace_bytes(int64_t N, int64_t K, int64_t C);
int64_t mlgnn_diffpool_large_f32_saved_bytes(int64_t N, int64_t K, int64_t C);
int mlgnn_diffpool_large_f32_fwd(const float* z, const float* adj, const float* s_logits, float* s_out, float* x_out,
                                 float* adj_out, float* scal_out, float* stats, void* workspace,
                                 int64_t workspace_bytes, int64_t N, int64_t K, int64_t C, int64_t B, int adj_batched,
                                 void* stream);
int64_t mlgnn_diffpool_large_f32_bwd_workspace_bytes(int64_t N, int64_t K, int64_t C, int adj_symmetric);
int mlgnn_diffpool_large_f32_bwd(const float* adj, const float* s_logits, const void* saved, const float* grad_x,
                                 const float* grad_adj_out, const float* grad_link, const float* grad_ent,
                                 const float* stats, float* grad_z, float* grad_logits, float* grad_adj,
                                 int adj_symmetric, void* workspace, int64_t workspace_bytes, int64_t N, int64_t K,
                                 int64_t C, int64_t B, int adj_batched, void* stream);

/*
 * fp32 nn.Linear on tall inputs past the widths of mlgnn_tallgemm_nt / mlgnn_linear_wgrad (hidden width 512 at d = 256:
 * the fp32 weight image does not fit LDS) -- torch_nn.py:54-75 at BASELINE configs[4]'s sizes in fp32.  Every product
 * as three bf16 terms on the matrix cores (see mlgnn_diffpool_large_f32_fwd), fp32 accumulation; the library's fp32
 * GEMMs for these shapes run at ~40 TFLOP/s.  R, J multiples of 128 (mlgnn_linear_f32x3_supported).
 *   fwd:  y [Npad, J] = x [N,R] w[J,R]^T + bias [J] (or NULL); Npad = mlgnn_linear_f32x3_padded_rows(N): the first N
 *         rows are the result (the rest is scratch)
 *   bwd:  grad_x [Npad, R] (or NULL) = grad_out [N,J] w;  grad_w [J,R] = grad_out^T x  (one product over the row index,
 *         split along it, fixed-order reduce);  grad_bias [J] (or NULL) = column sums of grad_out (partial sums per 64
 *         rows from the launch that reads grad_out anyway, fixed-order reduce)
 */
int mlgnn_linear_f32x3_supported(int64_t N, int64_t R, int64_t J);
int64_t mlgnn_linear_f32x3_padded_rows(int64_t N);
int64_t mlgnn_linear_f32x3_fwd_workspace_bytes(int64_t N, int64_t R, int64_t J);
int mlgnn_linear_f32x3_fwd(const float* x, const float* w, const float* bias, float* y, void* workspace,
                           int64_t workspace_bytes, int64_t N, int64_t R, int64_t J, void* stream);
int64_t mlgnn_linear_f32x3_bwd_workspace_bytes(int64_t N, int64_t R, int64_t J);
int mlgnn_linear_f32x3_bwd(const float* grad_out, const float* x, const float* w, float* grad_x, float* grad_w,
                           float* grad_bias, void* workspace, int64_t workspace_bytes, int64_t N, int64_t R, int64_t J,
                           void* stream);

/*
 * Optimizer step on one flat fp32 buffer: global-norm gradient clipping + Adam with L2 weight decay, two launches.
 * Replaces: train.py:63-66,112-114 -- clip_grad_norm_(parameters, max_norm=20, norm_type=2) (when --clip_grad) and
 * torch.optim.Adam(lr, betas, weight_decay).step(); torch's single-tensor formula, element by element:
 *   g' = clip g (+ weight_decay p);  m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= step_size m / (sqrt(v) / bias2_sqrt + eps),     clip = min(1, max_norm / (||g||_2 + 1e-6))
 * params, grads, exp_avg, exp_avg_sq: [n] fp32 (the module's tensors laid out contiguously, mlgnn.optim.FlatAdam);
 * step_size = lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t) for step count t (host doubles, like torch);
 * max_norm <= 0: no clipping (grads untouched); otherwise the clipped gradients are written back like
 * clip_grad_norm_ does and workspace[256] receives ||g||_2.
 * max_norm's coefficient follows clip_grad_norm_ for a non-finite norm too (a NaN norm poisons every gradient).
 * param_offsets (device, int64 [n_params + 1], first element of every parameter, last entry = n) and live (device,
 * float [n_params], > 0 = this parameter received a gradient this step): elements of a parameter that is not live are
 * left untouched (torch skips parameters whose .grad is None).  The flags are device data so that a data-parallel
 * caller can append them to the gradient all-reduce (every rank then steps the union); live = NULL: every element is
 * live (param_offsets is not read).  Any number of parameters.  The norm is taken over all n gradients: the caller
 * keeps the gradients of unreached parameters at zero.  workspace: mlgnn_adam_workspace_floats() floats.
 */
int64_t mlgnn_adam_workspace_floats(void);
int mlgnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                    const int64_t* param_offsets, const float* live, int64_t n_params, float max_norm, float beta1,
                    float beta2, float eps, float weight_decay, float step_size, float bias2_sqrt,
                    float* workspace, void* stream);

/*
 * The edge list SAGEConv propagates over (models/gcn_lib/sparse/torch_vertex.py:272-273: remove_self_loops, then
 * add_self_loops with weight 1.0), one pass, no compaction:  out_edge_index [2, E + N] int64 = the E input edges with
 * every existing self loop (i, i) re-pointed to the SPARE node N, followed by (i, i) for i < N;  out_weight [E + N]
 * (NULL: skipped) = edge_attr[e * attr_stride] (NULL: 1) then N ones.  Build the CSR over N + 1 nodes
 * (mlgnn_coo_to_csr) and aggregate over the first N rows: row N is never a destination and never gathered from.
 */
int mlgnn_sage_rewrite(const int64_t* edge_index, const float* edge_attr, int64_t attr_stride, int64_t E, int64_t N,
                       int64_t* out_edge_index, float* out_weight, void* stream);

/*
 * CSR (both orderings) of `copies` block-diagonal copies of ONE graph with N nodes and E edges, from that graph's CSR:
 * node ids shifted by b N, edge positions by b E (copy b's edges are COO positions [b E, (b + 1) E) of the batch, as a
 * PyG collate concatenates them).  Bit for bit what mlgnn_coo_to_csr builds from the batched edge list, in one stream.
 * For the fold-constant topology every sample of a TCGA batch shares (dataloader/multiloader.py:687-691).
 */
int mlgnn_csr_replicate(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const int32_t* rowptr_t,
                        const int32_t* col_t, const int32_t* pos_t, const int32_t* eid_t, int32_t* out_rowptr,
                        int32_t* out_col, int32_t* out_eid, int32_t* out_rowptr_t, int32_t* out_col_t, int32_t* out_pos_t,
                        int32_t* out_eid_t, int64_t N, int64_t E, int64_t copies, void* stream);

/*
 * SAGE update of one graph layer in ONE product (fp32):
 *     c [N,J] = leaky_relu([a | a2] * Bt^T + bias, act_slope) * row_scale[row]
 * a [N,R1], a2 [N,R2]: two column blocks of the left operand in tensors of their own (the node features and their
 * weighted neighbourhood mean) -- the concatenation [x || aggr] of the reference never exists; bt [J, R1+R2] row major.
 * Replaces: SAGEConv.update `self.nn(torch.cat((x, aggr_out), dim=1))` with nn = MLP([in+out, out], act) =
 * Linear -> LeakyReLU(0.2) (models/gcn_lib/sparse/torch_vertex.py:288-291,297-304; torch_nn.py:9-24,54-75), with
 * `lin_r` (:282-286) folded into bt by the caller (mean and lin_r commute), and the value mask of
 * MultilevelGNN.forward (models/multilevel_gnn.py:205-207: `x * mask_x`) as row_scale [N] (NULL: none).
 * act_slope: 1 = identity, 0 = ReLU, 0.2 = the reference's LeakyReLU.  row_max_out [N] (NULL: skipped): max |c[i]|;
 * a_row_max_out [N] (NULL: skipped): max over both blocks of |A[i]| -- the operand scales of the backward's products.
 * R1, R2 multiples of 16 with R1 + R2 in {32, 64, 128, 256} (R2 = 0, a2 = NULL: one operand -- a Linear with the
 * activation epilogue, e.g. the 1x1 convolutions + ReLU of the pathway head, models/multilevel_gnn.py:98-104);
 * J in {32, 64, 128}; workspace:
 * mlgnn_tallgemm_workspace_bytes(R1 + R2, J, f32); 16-byte aligned a, a2, bt, workspace.
 */
int mlgnn_tallgemm_dual_supported(int64_t N, int64_t R1, int64_t R2, int64_t J);
int mlgnn_tallgemm_dual(const float* a, const float* a2, const float* bt, const float* bias, float act_slope,
                        const float* row_scale, float* c, float* row_max_out, float* a_row_max_out, void* workspace,
                        int64_t workspace_bytes, int64_t N, int64_t R1, int64_t R2, int64_t J, void* stream);

/*
 * The fold of `lin_r` into the update's weight and its chain rule (one small launch each instead of mm / sub / cat chains):
 *   fwd: W_c = W_a W_r with w_nn = [W_x | W_a] ([out, in + out]), w_r [out, in];  w_cat [out, 2 in] = [W_x - rel W_c | W_c],
 *        and the two halves w_x1, w_c [out, in] on their own (operands of the backward's input-gradient products)
 *   bwd: G_c = grad_w_c - rel grad_w_x1;  grad_w_nn [out, in + out] = [grad_w_x1 | G_c W_r^T];  grad_w_r [out, in] = W_a^T G_c
 * Replaces: nothing of the reference (its per-edge `lin_r`, torch_vertex.py:282-286, is what the fold removes).
 */
int mlgnn_sage_fold_fwd(const float* w_nn, const float* w_r, float* w_cat, float* w_x1, float* w_c, int64_t cin, int64_t cout,
                        int relative, void* stream);
int mlgnn_sage_fold_bwd(const float* grad_w_x1, const float* grad_w_c, const float* w_nn, const float* w_r, float* grad_w_nn,
                        float* grad_w_r, int64_t cin, int64_t cout, int relative, void* stream);

/*
 * dst [B, C, R] = src [B, R, C]^T per sample (fp32).  Replaces: the copy inside `torch.flatten(x, start_dim=1)` in front of
 * MultilevelGNN's first head Linear (models/multilevel_gnn.py:277) when x lives channel-last, and its autograd.
 */
int mlgnn_transpose_batched(const float* src, float* dst, int64_t B, int64_t R, int64_t C, void* stream);

/*
 * Backward of that epilogue (csrc/sage.hip):  grad_z = grad_out * row_scale[row] * (z > 0 ? 1 : slope), the sign of z
 * taken from the stored result y = leaky_relu(z) * row_scale (slope > 0);  grad_z_row_max [N] (NULL: skipped) =
 * max |grad_z[i]|.  Replaces: autograd of LeakyReLU and of the mask product (torch_nn.py:9-24, multilevel_gnn.py:205-207).
 * J / 4 a power of two <= 64; 16-byte aligned tensors.
 */
int mlgnn_leaky_relu_bwd(const float* grad_out, const float* y, const float* row_scale, float slope, float* grad_z,
                         float* grad_z_row_max, int64_t N, int64_t J, void* stream);

/*
 * Node embedding of MultilevelGNN (models/multilevel_gnn.py:151: `x.reshape(-1, nodes, 1) * self.node_embedding`):
 *     h [batch * nodes, C] = x[b, n] * embedding[n, :]        (+ h_row_max [batch * nodes] or NULL)
 *     grad_embedding [nodes, C] = sum_b x[b, n] * grad_h[b, n, :]     (samples in order: bitwise reproducible)
 * C / 4 a power of two <= 64.
 */
int mlgnn_node_embed_fwd(const float* x, const float* embedding, float* h, float* h_row_max, int64_t batch,
                         int64_t nodes, int64_t C, void* stream);
int mlgnn_node_embed_bwd(const float* x, const float* grad_h, float* grad_embedding, int64_t batch, int64_t nodes,
                         int64_t C, void* stream);

/*
 * Linear with 1..8 input columns over tall rows (fp32): out [N,J] = x [N,R] w[J,R]^T + bias, and its weight / bias
 * gradient grad_w_b = [grad_out^T x  (J x R) | column sums of grad_out (J)] (fixed-order partial sums: deterministic).
 * Replaces: DeeperGCN.node_features_encoder = Linear(3 [+ emb], hidden) forward and weight gradient
 * (models/deepergcn.py:199-210), which a GEMM library runs as K = 3 products; here both are single streams over the
 * [N, J] tensor.  J in {32, 64, 128, 256}; workspace: mlgnn_narrow_linear_bwd_workspace_floats(R, J) floats.
 */
int mlgnn_narrow_linear_supported(int64_t N, int64_t R, int64_t J);
int64_t mlgnn_narrow_linear_bwd_workspace_floats(int64_t R, int64_t J);
int mlgnn_narrow_linear_fwd(const float* x, const float* w, const float* bias, float* out, int64_t N, int64_t R, int64_t J,
                            void* stream);
int mlgnn_narrow_linear_bwd(const float* grad_out, const float* x, float* grad_w_b, float* workspace,
                            int64_t workspace_floats, int64_t N, int64_t R, int64_t J, void* stream);

/*
 * Linear over a handful of rows with a very long input (fp32): y [M,J] = x [M,K] w[J,K]^T + bias, M <= 64, K % 4 == 0 --
 * the first layer of MultilevelGNN's head (models/multilevel_gnn.py:121-127; config/kirc.yaml: Linear(84 096, 512) on
 * 64 samples).  Every product of the layer is one stream over the weight (172 MB there): forward with fixed-order
 * partial sums over K ranges (workspace: mlgnn_skinny_linear_fwd_workspace_floats), backward = grad_x [M,K] (NULL:
 * skipped), grad_w [J,K] (NULL: skipped) and grad_b [J] (NULL: skipped), each written once.  Plain fp32 FMA arithmetic.
 */
int mlgnn_skinny_linear_supported(int64_t M, int64_t J, int64_t K);
int64_t mlgnn_skinny_linear_fwd_workspace_floats(int64_t M, int64_t J, int64_t K);
int mlgnn_skinny_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* workspace,
                            int64_t workspace_floats, int64_t M, int64_t J, int64_t K, void* stream);
int mlgnn_skinny_linear_bwd(const float* grad_out, const float* x, const float* w, float* grad_x, float* grad_w,
                            float* grad_b, int64_t M, int64_t J, int64_t K, void* stream);

/*
 * Measurement aid (bench.py: the box's streaming ceiling next to the 8 TB/s spec peak): dst = src, 16 bytes per lane,
 * non_temporal != 0: non-temporal loads and stores.  bytes a multiple of 16, 16-byte aligned pointers.
 */
int mlgnn_stream_copy(const void* src, void* dst, int64_t bytes, int non_temporal, void* stream);

/*
 * Debug facility (csrc/canary.hip; never on the product path, the entry points above never allocate): a guard-band
 * device allocator in the shape torch.cuda.memory.CUDAPluggableAllocator binds -- every block sits between two 4 KiB
 * bands of a byte pattern, the rear one starting at the first byte past the requested size -- and a checker.
 * mlgnn/_lib.py installs the pair as torch's allocator when MLGNN_CANARY=1 and calls mlgnn_canary_check() after every
 * C-ABI call: device synchronise, compare all bands of all live and recently freed blocks; returns 0 (intact), 1 (a
 * band was overwritten: `message` names the block, side and distance; the band is repaired) or -1 (the synchronise
 * itself failed: a faulting kernel).  mlgnn_canary_stats: out4 = {live blocks, device allocations, reuses, checks}.
 */
void* mlgnn_canary_malloc(int64_t size, int device, void* stream);
void mlgnn_canary_free(void* ptr, int64_t size, int device, void* stream);
int64_t mlgnn_canary_check(char* message, int64_t message_bytes);
int64_t mlgnn_canary_stats(int64_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* MLGNN_H */
