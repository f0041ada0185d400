/*
 * deepfm_hip.h — C ABI of the MI355X (gfx950) CTR feature-interaction library.
 *
 * The reference (CodexploreRepo/deepfm) has no FFI layer: its operator interface is
 * the forward() of four torch.nn modules plus autograd.  Each entry point below is
 * the native replacement of one of those Python call sites; the host-side mirror in
 * deepfm_amd/models/layers/ binds them with ctypes.
 *
 * Conventions
 *   - every pointer named d_* / in the "device" column is a DEVICE pointer (HBM);
 *     pointer tables passed as `const T* const*` are HOST arrays of device pointers;
 *   - all floating point is IEEE fp32, ids are int64 (reference dataset.py:28-38);
 *   - `stream` is a hipStream_t (NULL = the null stream); calls only enqueue work,
 *     they never synchronise and are safe under hipGraph stream capture;
 *   - return value: 0 = DFM_OK, otherwise an error code; dfm_last_error() gives text;
 *   - id range errors (id < 0 or id >= vocabulary) never fault: the id is treated as
 *     the padding id 0 and bit 0 of *d_error_flag is set (the reference raises
 *     IndexError from ATen; the Python mirror raises IndexError when it reads the flag).
 */
#ifndef DEEPFM_HIP_H
#define DEEPFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFM_ABI_VERSION 8   /* bump whenever a struct layout or a signature in this header changes */
#define DFM_MAX_FIELDS 64      /* per-call pointer tables travel as kernel arguments */
#define DFM_MAX_RANKS 64       /* data-parallel ranks of one job (csrc/shard.hip) */
#define DFM_ROWPLAN_CHUNK 4096 /* ids per sorted list (one LDS-resident sort) */

enum dfm_status { DFM_OK = 0, DFM_ERR_INVALID = 1, DFM_ERR_HIP = 2, DFM_ERR_UNSUPPORTED = 3 };
enum dfm_field_kind { DFM_SPARSE = 0, DFM_DENSE = 1, DFM_SEQUENCE = 2 };
enum dfm_combiner { DFM_MEAN = 0, DFM_SUM = 1, DFM_MAX = 2 };

typedef void* dfm_stream_t;

int dfm_abi_version(void);
const char* dfm_last_error(void);
/* CU count, wavefront size and gcnArchName of the current device. */
int dfm_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);
/* Launches an empty kernel (timing calibration of event brackets in bench.py). */
int dfm_debug_empty_launch(dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * FeatureEmbedding  (reference deepfm/models/layers/embedding.py:20-126)
 * ------------------------------------------------------------------------------- */

/* One schema field with its parameter tensors (reference embedding.py:32-62):
 *   SPARSE / SEQUENCE: w2 = second_order_embeddings.<f>.weight (vocab, dim),
 *                      w1 = first_order_embeddings.<f>.weight  (vocab, 1)
 *   DENSE:             w2 = Linear(1,dim).weight (dim,1), b2 = .bias (dim),
 *                      w1 = Linear(1,1).weight (1,1),     b1 = .bias (1)
 *   proj = projections.<f>.weight (fm_dim, dim) or NULL when dim == fm_dim. */
typedef struct dfm_field {
  int32_t kind;        /* dfm_field_kind */
  int32_t dim;         /* embedding_dim of the field */
  int32_t vocab;       /* rows of w2 / w1 (0 for DENSE) */
  int32_t max_len;     /* SEQUENCE: ids per bag (L) */
  int32_t combiner;    /* SEQUENCE: dfm_combiner */
  int32_t flat_offset; /* column of this field inside flat_embeddings */
  int32_t stride2;     /* floats between consecutive rows of w2 (0 = dim: contiguous) */
  int32_t stride1;     /* floats between consecutive rows of w1 (0 = 1: contiguous) */
  const float* w2;
  const float* b2;
  const float* w1;
  const float* b1;
  const float* proj;
} dfm_field;

/* Gradient buffers matching dfm_field (dense, same shapes as the parameters). */
typedef struct dfm_field_grad {
  float* w2;
  float* b2;
  float* w1;
  float* b1;
  float* proj;
} dfm_field_grad;

typedef struct dfm_embedding_plan dfm_embedding_plan;

/* Uploads the field table once.  `fields` is a host array in schema order. */
int dfm_embedding_plan_create(const dfm_field* fields, int num_fields, int fm_dim,
                              dfm_embedding_plan** out_plan);
int dfm_embedding_plan_destroy(dfm_embedding_plan* plan);
/* 1 when every field is SPARSE or DENSE with dim == fm_dim (dim % 4 == 0) and no
 * projection: flat_embeddings is then field_embeddings reshaped (same bytes) and the
 * fused gather kernel is used. */
int dfm_embedding_plan_is_uniform(const dfm_embedding_plan* plan);
/* Bytes of scratch the forward/backward calls need for a batch of `batch` samples. */
size_t dfm_embedding_workspace_bytes(const dfm_embedding_plan* plan, int64_t batch);

/* FeatureEmbedding.forward (embedding.py:76-126).
 *   inputs[f]: SPARSE int64 (B,), SEQUENCE int64 (B, L) row-major, DENSE float (B,)
 *   d_first_order (B,1)  d_field_emb (B,F,fm_dim)  d_flat_emb (B, sum dim)
 * For a uniform plan d_flat_emb may be NULL or equal to d_field_emb (aliased).
 * d_fm_out (B,1), optional (uniform plans only): the FMInteraction value
 * 0.5*sum_d[(sum_f e)^2 - sum_f e^2] (fm.py:18-23) computed from the rows while they
 * are in registers; d_fm_sum (B, fm_dim), optional: S = sum_f e, which the FM backward
 * g*(S - e) needs (dfm_linear_backward's dfm_fm_bwd epilogue). */
int dfm_embedding_forward(const dfm_embedding_plan* plan, const void* const* inputs, int64_t batch,
                          float* d_first_order, float* d_field_emb, float* d_flat_emb,
                          float* d_fm_out, float* d_fm_sum, void* d_workspace, int32_t* d_error_flag,
                          dfm_stream_t stream);

/* dfm_embedding_forward for a uniform plan whose inputs sit in a batch record that changes every
 * step: the kernel also copies every field's raw input to stage_out[f] (host array of device
 * pointers, schema order: int64 (B,) / float (B,)) and, optionally, one float per sample from
 * d_extra_src to d_extra_dst (the labels) — the "load the next batch into the step's static
 * buffers" copy (reference trainer.py:214-217) without a launch of its own. */
int dfm_embedding_forward_staged(const dfm_embedding_plan* plan, const void* const* inputs,
                                 void* const* stage_out, const float* d_extra_src, float* d_extra_dst,
                                 int64_t batch, float* d_first_order, float* d_field_emb, float* d_fm_out,
                                 float* d_fm_sum, int32_t* d_error_flag, dfm_stream_t stream);

/* dfm_embedding_forward_staged captured inside a graph (e.g. by torch.cuda.graph): rewrites the kernel
 * node of the INSTANTIATED graph (`graph_exec`: hipGraphExec_t, `node`: hipGraphNode_t from
 * dfm_graph_last_node right after the captured call) so that its next launch reads another batch
 * record.  Same arguments as the captured call; host-side only, nothing is enqueued.  Do not update
 * an exec whose previous launch may still be pending. */
int dfm_embedding_forward_staged_update(const dfm_embedding_plan* plan, void* graph_exec, void* node,
                                        const void* const* inputs, void* const* stage_out,
                                        const float* d_extra_src, float* d_extra_dst, int64_t batch,
                                        float* d_first_order, float* d_field_emb, float* d_fm_out,
                                        float* d_fm_sum, int32_t* d_error_flag);

/* Graph plumbing: the node of the operation captured last on `stream` (call right after the launch). */
int dfm_graph_last_node(dfm_stream_t stream, void** node_out);

/* Kernel-accurate timing of the uniform gather (measurement aid, bench.py): after
 * dfm_gather_timing_begin(n) the next n dfm_embedding_forward launches of a uniform plan carry HIP
 * start/stop events recorded around the dispatch itself (hipExtLaunchKernelGGL), on the stream the
 * kernel is launched on; dfm_gather_timing_end synchronises them, writes up to `capacity` durations
 * in microseconds to the HOST array h_us, the number written to *h_count, and disarms. */
int dfm_gather_timing_begin(int launches);
int dfm_gather_timing_end(float* h_us, int capacity, int* h_count);

/* Launch shape of the uniform gather (tuning aid, tools/time_gather.py): 0 = automatic (default),
 * 1 = one wave owns its samples across all fields (26 sparse + 13 dense slots in flight, no LDS),
 * 2 = 2 waves x (13 + 7), 3 = 4 waves x (8 + 4), 4 = 8 waves x (4 + 2), 5 = the two-wave kernel with
 * compile-time slot indices (the automatic choice where it applies), 6 = shape 5 with streaming stores.
 * Every shape computes the
 * same values (bitwise for the gathered rows); only the work split inside a workgroup changes. */
int dfm_gather_set_shape(int shape);

/* Autograd of the forward w.r.t. every parameter as DENSE gradients — the reference
 * semantics (nn.Embedding(sparse=False), embedding.py:35-40).  Gradients are ADDED
 * into the buffers of `grads` (caller zero-fills them); row 0 receives none.
 * d_g_flat may be NULL for a uniform plan whose flat view aliases field_emb (the
 * caller has already summed both into d_g_field).  d_flat_saved: the forward's
 * flat_embeddings, needed only when the plan has projections (else NULL). */
int dfm_embedding_backward_dense(const dfm_embedding_plan* plan, const void* const* inputs,
                                 int64_t batch, const float* d_g_first, const float* d_g_field,
                                 const float* d_g_flat, const dfm_field_grad* grads,
                                 const void* d_flat_saved, dfm_stream_t stream);

/* Gradients of the DENSE fields' Linear parameters only (deterministic tree
 * reduction over the batch); used by the row-sparse mode together with
 * dfm_rowgrad_build for the SPARSE fields. */
int dfm_embedding_backward_dense_fields(const dfm_embedding_plan* plan, const void* const* inputs,
                                        int64_t batch, const float* d_g_first,
                                        const float* d_g_field, const float* d_g_flat,
                                        const dfm_field_grad* grads, dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Row plan + row-wise gradient: the backward scatter-add of the SPARSE fields without
 * atomics (reference: autograd of embedding.py:95-98, i.e. aten::embedding_dense_backward)
 * ------------------------------------------------------------------------------- */

/* For each of `num_sparse` id vectors (n ids each) and each chunk of DFM_ROWPLAN_CHUNK
 * ids: stable sort by id, drop id 0, find the distinct ids.  With C = ceil(n/chunk):
 *   d_sorted_pos (C, S, chunk) int32  sample position of every sorted entry
 *   d_uniq_rows  (C, S, chunk) int32  distinct ids, ascending
 *   d_seg_start  (C, S, chunk+1) int32  first sorted entry of every distinct id
 *   d_num_uniq   (C, S) int32
 * Entries of a list behind its num_uniq (+ 1 for d_seg_start) are working space of the library: the tail of
 * d_seg_start carries the list's runs of more than half a chunk and an arrival counter for
 * dfm_rowgrad_build, which sums such a run with several workgroups.
 * Depends on the ids only, so it can run ahead of the forward pass.
 * touch_tables (optional; dfm_table per SPARSE field, only w2 / stride2 are read; dim = row length for stride2 = 0):
 * extra workgroups of the same launch read the ids and TOUCH the first line of every row the batch will gather
 * (embedding.py:95-98 reads them next), pulling the ids and the 128-B row lines into the Infinity Cache while the
 * sort occupies 26 of the 256 CUs — launched in FRONT of dfm_embedding_forward_staged on the same batch record
 * (training/step.py), the gather then finds its operands on-die.  Results do not depend on it.
 * _update: the launch was captured into a graph; point its node (dfm_graph_last_node) of the instantiated graph
 * at other id columns — host-side only, rules of dfm_embedding_forward_staged_update. */
typedef struct dfm_table dfm_table;
int dfm_rowplan_build(const int64_t* const* ids, const int32_t* vocab, int num_sparse, int64_t n,
                      int32_t* d_sorted_pos, int32_t* d_uniq_rows, int32_t* d_seg_start,
                      int32_t* d_num_uniq, int32_t* d_error_flag, const dfm_table* touch_tables, int dim,
                      dfm_stream_t stream);
int dfm_rowplan_build_update(void* graph_exec, void* node, const int64_t* const* ids, const int32_t* vocab,
                             int num_sparse, int64_t n, int32_t* d_sorted_pos, int32_t* d_uniq_rows,
                             int32_t* d_seg_start, int32_t* d_num_uniq, int32_t* d_error_flag,
                             const dfm_table* touch_tables, int dim);

/* Row gradients of the distinct ids, contributions added in increasing sample order (runs of more than 64
 * contributions: by a fixed tree — reproducible run to run, equal to the sequential sum up to rounding):
 *   d_row_g2 (C, S, chunk, dim)   d_row_g1 (C, S, chunk)      rows behind num_uniq: working space
 * d_seg_start must come from dfm_rowplan_build (its tail is read, and its counter written).
 * `field_of_sparse[s]` is the schema position of sparse field s inside d_g_field
 * (B, F, dim); d_g_first is (B,1).  Uniform plans only (dim == fm_dim). */
int dfm_rowgrad_build(const int32_t* field_of_sparse, int num_sparse, int num_fields, int dim,
                      int64_t n, const float* d_g_first, const float* d_g_field,
                      const int32_t* d_sorted_pos, int32_t* d_seg_start,
                      const int32_t* d_num_uniq, float* d_row_g2, float* d_row_g1,
                      dfm_stream_t stream);

/* Row-wise Adam over `num_lists` = ranks x chunks lists (all-gathered under data
 * parallelism).  A row present in several lists is owned by its first list; the owner
 * adds the other lists' gradients in list order (bit-identical on every replica),
 * then:  g = grad_scale * sum + 2*l2*w ;  g *= clip ;  Adam(w, m, v, g)
 * (reference trainer.py:224-237 restricted to the rows the batch touched — DESIGN.md).
 * Pass A (dfm_rowadam_merge) writes the merged gradients in place and one partial sum of
 * |g|^2 per block into d_partials[0 .. dfm_rowadam_num_partials); pass B
 * (dfm_rowadam_apply) reads *d_clip_coef (NULL = 1). */
typedef struct dfm_table {
  float* w2; float* m2; float* v2;   /* (V, dim) weights and Adam moments, row stride `stride2` floats */
  float* w1; float* m1; float* v1;   /* (V, 1) first-order ones, row stride `stride1` floats */
  int32_t stride2;                   /* 0 = dim (contiguous tensors) */
  int32_t stride1;                   /* 0 = 1 */
} dfm_table;
/* Packed row records (FeatureEmbedding.pack_tables_): all six arrays are views of one (V, RS)
 * buffer — [w2 | w1 m1 v1 pad | m2 | v2 | pad], RS a multiple of 32 floats — so a row's forward
 * reads hit one 128-B line and its Adam update touches one contiguous record. */

int64_t dfm_rowadam_num_partials(int num_sparse, int dim, int num_lists);
int dfm_rowadam_merge(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                      const int32_t* d_uniq_rows, const int32_t* d_num_uniq, float* d_row_g2,
                      float* d_row_g1, int32_t* d_owner_flag, float grad_scale, float l2,
                      float* d_partials, dfm_stream_t stream);
int dfm_rowadam_apply(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                      const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                      const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef,
                      float lr, float beta1, float beta2, float eps, const int32_t* d_step,
                      dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Dense parameters on one flat fp32 buffer: L2 term, gradient norm, clip, Adam
 * (reference base.py:78-83, trainer.py:224-237)
 * ------------------------------------------------------------------------------- */
/* g[i] += 2*l2*p[i] for i < n_l2 (the embedding parameters come first in the buffer);
 * one partial sum of |g|^2 per block into d_partials[0 .. dfm_dense_num_partials(n)). */
int64_t dfm_dense_num_partials(int64_t n);
int dfm_dense_grad_prepare(float* d_g, const float* d_p, int64_t n, int64_t n_l2, float l2,
                           float* d_partials, dfm_stream_t stream);
/* *d_sq_norm = sum(partials) in a fixed order; *d_clip_coef = min(1, max_norm/(sqrt+1e-6))
 * (clip_grad_norm_, trainer.py:232-235; max_norm <= 0 disables clipping: coef 1).
 * Optional "tick": *d_step_tick += 1 (Adam's step count, read by the update kernels enqueued after
 * this call) and *d_seed_tick += 1 (dropout seed of the next step) — saves two launches. */
int dfm_grad_norm_finalize(const float* d_partials, int64_t num_partials, float max_norm,
                           float* d_sq_norm, float* d_clip_coef, int32_t* d_step_tick,
                           int64_t* d_seed_tick, dfm_stream_t stream);
/* torch.optim.Adam (trainer.py:67-70) on flat buffers with g scaled by *d_clip_coef;
 * zero_grad != 0 also clears d_g (optimizer.zero_grad() of the next step, trainer.py:219). */
int dfm_dense_adam(float* d_p, float* d_m, float* d_v, float* d_g, int64_t n,
                   const float* d_clip_coef, float lr, float beta1, float beta2, float eps,
                   const int32_t* d_step, int zero_grad, dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * FMInteraction  (reference deepfm/models/layers/fm.py:18-23)
 * ------------------------------------------------------------------------------- */
int dfm_fm_forward(const float* d_field_emb, int64_t batch, int num_fields, int dim, float* d_out,
                   dfm_stream_t stream);
/* d e[b,f,:] = g[b] * (S[b,:] - e[b,f,:]) */
int dfm_fm_backward(const float* d_field_emb, const float* d_g_out, int64_t batch, int num_fields,
                    int dim, float* d_g_field, dfm_stream_t stream);
/* d field_embeddings of a model with several consumers of the embeddings (attention_deepfm.py:48-66):
 * out[b, :] = g_flat[b, :width] (row stride ld_flat floats: a slice of the DNN's d input)
 *           + g_extra[b, :]                     (d from another layer, e.g. the attention stack; or NULL)
 *           + g_fm[b] * (S[b, d] - e[b, f, d])  (FMInteraction backward, fm.py:18-23; or NULL). */
int dfm_embedding_grad_combine(const float* d_g_flat, int64_t ld_flat, const float* d_g_extra, const float* d_g_fm,
                               const float* d_fm_sum, const float* d_field_emb, int64_t batch, int num_fields, int dim,
                               float* d_g_field, dfm_stream_t stream);
/* dst[r, :width] = src[r, :width] for r < rows, row strides ld_src / ld_dst floats (slices of the
 * concatenated DNN input of attention_deepfm.py:57-61 and of its gradient). */
int dfm_copy_2d(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t rows, int width,
                dfm_stream_t stream);

/* Arithmetic of the matrix-core CIN path: 0 = bf16 x 3 split products (default; meets the 1e-4 bar
 * against cin.py:66-105), 1 = plain bf16 (throughput mode, its own looser tolerance), 2 = exact-fp32
 * VALU kernels only.  Process-wide; bench.py records it and refuses to report headline numbers in a
 * non-default mode. */
int dfm_cin_set_mode(int mode);
int dfm_cin_get_mode(void);

/* ---------------------------------------------------------------------------------
 * CIN  (reference deepfm/models/layers/cin.py:26-105)
 *   per layer i:  Z[b,h*F+f,d] = hidden_i[b,h,d] * x0[b,f,d]          (cin.py:84-87, never stored)
 *                 Y_i = relu(W_i Z + bias_i)  (B, C_i, D)              (cin.py:90-91)
 *                 split [direct | next] along channels, direct first   (cin.py:93-96)
 *                 out[:, col_i : col_i+direct_i] = sum_d Y_i[:, :direct_i, :]  (cin.py:102-105)
 *   weights[i] = conv_layers.<i>.weight (C_i, H_i*F[, 1]),  biases[i] = conv_layers.<i>.bias (C_i)
 *   layer_sizes / split_half as in CIN.__init__ (cin.py:41-64).
 * d_saved receives the post-ReLU Y_i of every layer (dfm_cin_saved_bytes) for the backward
 * (may be NULL for inference on the matrix-core path).  d_workspace
 * (dfm_cin_forward_workspace_bytes) holds the bf16 hi/lo weight fragments of the MFMA path
 * (F <= 40, D in {8,16,32}, layer sizes <= 128); without it, or for other shapes, the general
 * fp32 kernels run.  Environment DFM_CIN_MODE = split (default, bf16 x 3: parity grade) |
 * bf16 (plain bf16 MFMA, throughput mode) | fp32 (general kernels only).
 * ------------------------------------------------------------------------------- */
int dfm_cin_output_dim(const int32_t* layer_sizes, int num_layers, int split_half);
size_t dfm_cin_saved_bytes(const int32_t* layer_sizes, int num_layers, int split_half, int64_t batch,
                           int num_fields, int dim);
size_t dfm_cin_forward_workspace_bytes(const int32_t* layer_sizes, int num_layers, int split_half,
                                       int num_fields, int dim);
size_t dfm_cin_backward_workspace_bytes(const int32_t* layer_sizes, int num_layers, int split_half,
                                        int64_t batch, int num_fields, int dim);
int dfm_cin_forward(const float* d_x0, int64_t batch, int num_fields, int dim,
                    const float* const* weights, const float* const* biases,
                    const int32_t* layer_sizes, int num_layers, int split_half, float* d_out,
                    float* d_saved, void* d_workspace, dfm_stream_t stream);
/* d_g_x0 (B,F,D) is overwritten; weight / bias gradients are ADDED into g_weights[i] / g_biases[i]. */
int dfm_cin_backward(const float* d_x0, int64_t batch, int num_fields, int dim,
                     const float* const* weights, const int32_t* layer_sizes, int num_layers,
                     int split_half, const float* d_saved, const float* d_g_out, float* d_g_x0,
                     float* const* g_weights, float* const* g_biases, void* d_workspace,
                     dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Field self-attention block  (reference deepfm/models/layers/attention.py:67-120)
 *   one _AttentionBlock per call; MultiHeadSelfAttention.forward (attention.py:52-64) calls it
 *   once per layer.  params[] (device pointers), PyTorch layouts:
 *     0 W_q.weight (A,D) 1 W_q.bias (A) 2 W_k.weight 3 W_k.bias 4 W_v.weight 5 W_v.bias
 *     6 W_out.weight (D,A) 7 W_out.bias (D) 8 layer_norm.weight (D) 9 layer_norm.bias (D)
 *   (8, 9 only when use_residual).  x, out, g_out, g_x are (B, F, D).
 * The backward recomputes the forward per sample; parameter gradients are ADDED into
 * g_params[] (same order/shapes as params[]); d_g_x is overwritten.
 * ------------------------------------------------------------------------------- */
int dfm_attention_forward(const float* d_x, int64_t batch, int num_fields, int embed_dim,
                          int attention_dim, int num_heads, int use_residual,
                          const float* const* params, float* d_out, dfm_stream_t stream);
size_t dfm_attention_backward_workspace_bytes(int64_t batch, int embed_dim, int attention_dim);
int dfm_attention_backward(const float* d_x, const float* d_g_out, int64_t batch, int num_fields,
                           int embed_dim, int attention_dim, int num_heads, int use_residual,
                           const float* const* params, float* d_g_x, float* const* g_params,
                           void* d_workspace, dfm_stream_t stream);

/* GEMM-structured attention path (attention.py:91-120): the projections run on dfm_gemm_f32 over
 * the B*F rows; these are the remaining per-sample pieces.
 *   core   : o[b,i,h*hd:(h+1)*hd] = softmax(q_h k_h^T / sqrt(hd)) v_h, qkv (B*F, 3A) rows [q|k|v]
 *            (F <= 64, head_dim in {4,8,16,32}); the backward recomputes the probabilities
 *   LN     : out = LayerNorm(y + res) * gamma + beta over the last dim; stats (rows, 2) = mean, rstd;
 *            backward gives d(y+res) and ADDS d gamma / d beta */
int dfm_attention_core_supported(int num_fields, int attention_dim, int num_heads);
int dfm_attention_core_forward(const float* d_qkv, int64_t batch, int num_fields, int attention_dim,
                               int num_heads, float* d_o, dfm_stream_t stream);
int dfm_attention_core_backward(const float* d_qkv, const float* d_g_o, int64_t batch, int num_fields,
                                int attention_dim, int num_heads, float* d_g_qkv, dfm_stream_t stream);
/* The same core with the Q | K | V projection (attention.py:95-97, three Linear(D, A)) inside the kernel:
 * d_x (batch * num_fields, embed_dim), d_w_qkv (3 * attention_dim, embed_dim) = [W_q; W_k; W_v] stacked,
 * d_b_qkv (3 * attention_dim).  The (batch * num_fields, 3 * attention_dim) projection is never written;
 * the backward recomputes it from d_x and returns its gradient d_g_qkv, which feeds the weight / input
 * gradient GEMMs.  _supported(): head_dim 16, <= 48 fields, embed_dim in {16, 32, 48, 64}; all buffers
 * 16-byte aligned. */
int dfm_attention_qkv_core_supported(int num_fields, int embed_dim, int attention_dim, int num_heads);
int dfm_attention_qkv_core_forward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv, int64_t batch,
                                   int num_fields, int embed_dim, int attention_dim, int num_heads, float* d_o,
                                   dfm_stream_t stream);
int dfm_attention_qkv_core_backward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv,
                                    const float* d_g_o, int64_t batch, int num_fields, int embed_dim,
                                    int attention_dim, int num_heads, float* d_g_qkv, dfm_stream_t stream);
/* The whole forward of one _AttentionBlock (attention.py:91-120) in ONE launch, num_heads == 4 (a workgroup's
 * four waves are the four heads of a sample): projection + softmax(QK^T / sqrt(hd)) V + W_out + b_out and, when
 * d_gamma / d_beta are given, LayerNorm(y + x).  Writes what the backward needs: d_o (batch * num_fields,
 * attention_dim) head outputs, d_y (batch * num_fields, embed_dim) = the block's output before the residual,
 * d_stats (rows, 2) mean / rstd; d_out as dfm_layernorm_forward (out_group_stride > 0: sample b's rows at
 * d_out + b * out_group_stride).  _supported(): dfm_attention_qkv_core_supported and num_heads == 4. */
int dfm_attention_block_supported(int num_fields, int embed_dim, int attention_dim, int num_heads);
int dfm_attention_block_forward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv, const float* d_w_out,
                                const float* d_b_out, const float* d_gamma, const float* d_beta, float eps,
                                int64_t batch, int num_fields, int embed_dim, int attention_dim, int num_heads,
                                float* d_o, float* d_y, float* d_out, float* d_stats, int64_t out_group_stride,
                                float* d_x_copy, int64_t x_copy_group_stride, dfm_stream_t stream);
/* (d_x_copy, optional: the block's input rows once more, sample b at d_x_copy + b * x_copy_group_stride — the
 * second half of AttentionDeepFM's cat([attention(e), e]), attention_deepfm.py:57-61, without a copy pass.)
 * The backward of the same block from d_g_y (gradient of the output projection's result y; residual != 0: also
 * the gradient reaching x through the residual) to d_g_x (batch * num_fields, embed_dim), in ONE launch: the
 * head's d O = d y W_out[:, head], the core backward, and d x = d Q W_q + d K W_k + d V W_v (+ d y) all stay in
 * the kernel.  d_g_qkv (batch * num_fields, 3 * attention_dim) is written for the weight-gradient GEMM
 * d W_qkv = d_g_qkv^T x; d W_out = d_g_y^T o is a GEMM on the forward's d_o.  d_g_x must not alias an input.
 * Optional further terms of d_g_x for the block that reads the field embeddings themselves (what
 * dfm_embedding_grad_combine adds in a pass of its own): d_g_flat (rows of num_fields * embed_dim floats at
 * stride ld_flat: the flat half of the DNN's d input) and the FM backward d_g_fm[b] * (d_fm_sum[b, :] - x). */
int dfm_attention_block_backward(const float* d_x, const float* d_w_qkv, const float* d_b_qkv, const float* d_w_out,
                                 const float* d_g_y, int residual, int64_t batch, int num_fields, int embed_dim,
                                 int attention_dim, int num_heads, float* d_g_qkv, float* d_g_x,
                                 const float* d_g_flat, int64_t ld_flat, const float* d_g_fm, const float* d_fm_sum,
                                 dfm_stream_t stream);
size_t dfm_layernorm_workspace_bytes(int64_t rows, int dim);
/* out_group_rows > 0: output row r is written at d_out + (r / out_group_rows) * out_group_stride +
 * (r % out_group_rows) * dim (the attention output as the first half of the DNN's concatenated input,
 * attention_deepfm.py:57-61, without a copy); 0: contiguous (rows, dim). */
int dfm_layernorm_forward(const float* d_y, const float* d_res, int64_t rows, int dim, const float* d_gamma,
                          const float* d_beta, float eps, float* d_out, float* d_stats, int64_t out_group_rows,
                          int64_t out_group_stride, dfm_stream_t stream);
/* g_group_rows / g_group_stride: the same addressing for the incoming gradient d_g_out. */
int dfm_layernorm_backward(const float* d_g_out, const float* d_y, const float* d_res, const float* d_stats,
                           int64_t rows, int dim, const float* d_gamma, float* d_g_sum, float* d_g_gamma,
                           float* d_g_beta, void* d_workspace, int64_t g_group_rows, int64_t g_group_stride,
                           dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * DNN tower glue (reference deepfm/models/layers/dnn.py:45-55): BatchNorm1d (training
 * statistics) -> ReLU -> Dropout between the Linear GEMMs, two launches each way.
 *   forward : out = dropout_p(relu(gamma * (z - mean) * rstd + beta)), z (batch, features);
 *             saves mean / rstd (2, features); updates running_mean / running_var (unbiased,
 *             momentum) and num_batches_tracked like nn.BatchNorm1d when they are non-NULL
 *   backward: d z from d out (recomputes y and the dropout mask); ADDS d gamma / d beta
 * Dropout keeps element i iff hash(*d_seed, salt, i) >= p * 2^32 (scale 1/(1-p)); the seed is
 * read on the device, so forward and backward agree and a replayed graph gets fresh masks.
 * ------------------------------------------------------------------------------- */
size_t dfm_bn_workspace_bytes(int64_t batch, int features);
int dfm_bn_relu_dropout_forward(const float* d_z, int64_t batch, int features, const float* d_gamma,
                                const float* d_beta, float* d_running_mean, float* d_running_var,
                                int64_t* d_num_batches, float momentum, float eps, float p_drop,
                                const int64_t* d_seed, int salt, float* d_out, float* d_mean_rstd,
                                void* d_workspace, dfm_stream_t stream);
int dfm_bn_relu_dropout_backward(const float* d_g_out, const float* d_z, const float* d_mean_rstd,
                                 const float* d_gamma, const float* d_beta, int64_t batch, int features,
                                 float p_drop, const int64_t* d_seed, int salt, float* d_g_z,
                                 float* d_g_gamma, float* d_g_beta, void* d_workspace,
                                 dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * BCEWithLogitsLoss, mean reduction (reference deepfm/training/trainer.py:59, 221):
 * *d_loss = mean(max(z,0) - z*y + log1p(exp(-|z|)));  d_g_logits = (sigmoid(z) - y) / n.
 * ------------------------------------------------------------------------------- */
size_t dfm_bce_workspace_bytes(int64_t n);
int dfm_bce_with_logits(const float* d_logits, const float* d_labels, int64_t n, float* d_loss,
                        float* d_g_logits, void* d_workspace, dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Fused DNN tower + head of the training step (reference dnn.py:45-55 [Linear -> BatchNorm1d ->
 * ReLU -> Dropout] * n; deepfm.py:30-42 first_order + fm + output_linear(dnn); trainer.py:59,221
 * BCEWithLogitsLoss).  Few launches, none with a serial tail (csrc/tower.hip):
 *   forward, per layer : dfm_linear_bn_forward (GEMM + per-tile column statistics),
 *                        dfm_bn_relu_dropout_apply (merges the statistics, then normalises)
 *   head               : dfm_head_bce (logits, d logits, the last BN's mask, partial sums), or
 *                        dfm_head_bn_bce = the last block's dfm_bn_relu_dropout_apply + dfm_head_bce
 *   backward, per layer: dfm_bn_backward_apply (merges the partial sums, then dz),
 *                        dfm_linear_backward (dW and dx in one launch; the dx epilogue is the lower
 *                        layer's BatchNorm mask or the FM backward)
 * All reductions have a fixed order (no floating-point atomics, no arrival counters).
 * ------------------------------------------------------------------------------- */

/* One BatchNorm -> ReLU -> Dropout block as the backward sees it.  Producer (dfm_head_bce or
 * dfm_linear_backward's bn_below): dy = g * [y > 0] * dropout mask/(1-p), y = gamma*(z-mean)*rstd
 * + beta, plus partial column sums of dy and dy*xhat in `workspace`.  Consumer
 * (dfm_bn_backward_apply): merges them, adds the affine gradients, writes dz. */
typedef struct dfm_bn_bwd {
  const float* z;          /* (batch, features) pre-activations saved by the forward */
  const float* mean_rstd;  /* (2, features) batch mean, 1/sqrt(var+eps) from dfm_bn_relu_dropout_apply */
  const float* gamma;      /* (features) BatchNorm weight */
  const float* beta;       /* (features) BatchNorm bias */
  float* dy;               /* (batch, features) masked upstream gradient */
  float* g_gamma;          /* (features), ADDED: sum dy * xhat */
  float* g_beta;           /* (features), ADDED: sum dy */
  const int64_t* seed;     /* device dropout seed (NULL when p_drop == 0) */
  void* workspace;         /* dfm_bn_bwd_workspace_bytes(batch, features) */
  float p_drop;
  int32_t salt;            /* layer index, as passed to dfm_bn_relu_dropout_apply */
} dfm_bn_bwd;

/* What else flows back into the embeddings, folded into the first Linear's d input:
 * d e = d flat + g_fm * (S - e)  [FMInteraction backward, fm.py:18-23; g_fm NULL: no FM term]
 *              + addend          [gradient from another consumer of field_embeddings: CIN
 *                                 (xdeepfm.py:36-48) or the attention stack; NULL: none]. */
typedef struct dfm_fm_bwd {
  const float* g_fm;       /* (batch) d loss / d fm value, or NULL */
  const float* fm_sum;     /* (batch, dim) S = sum_f e (dfm_embedding_forward's d_fm_sum) */
  const float* e;          /* (batch, fields * dim) field embeddings */
  const float* addend;     /* (batch, fields * dim), or NULL */
  int32_t dim;
} dfm_fm_bwd;

/* What dfm_head_bce leaves for the dfm_bn_backward_apply that follows it to finish. */
typedef struct dfm_head_tail {
  float* g_w;              /* (features) head weight gradient, ADDED */
  float* g_b;              /* (1) head bias gradient, ADDED (may be NULL) */
  float* loss;             /* (1) mean BCE-with-logits, written */
  float* g_b2;             /* (1) ADDED the same sum as g_b: the bias of another 1-wide Linear that is added to the
                            * logit (xDeepFM's cin_linear, xdeepfm.py:41-47), or NULL */
} dfm_head_tail;

size_t dfm_linear_bn_workspace_bytes(int64_t batch, int features);
/* z (batch, out) = x W^T + b; per-column (mean, M2) of every 32-row tile go to d_workspace for
 * dfm_bn_relu_dropout_apply. */
int dfm_linear_bn_forward(const float* d_x, int64_t ldx, const float* d_w, const float* d_bias,
                          int64_t batch, int out_features, int in_features, float* d_z,
                          void* d_workspace, dfm_stream_t stream);
/* Batch statistics from dfm_linear_bn_forward's workspace -> d_mean_rstd (2, features) = mean and
 * 1/sqrt(biased var + eps); running_mean / running_var / num_batches_tracked updated like
 * nn.BatchNorm1d when non-NULL; out = dropout_p(relu(gamma * (z - mean) * rstd + beta)).
 * features % 4 == 0. */
int dfm_bn_relu_dropout_apply(const float* d_z, int64_t batch, int features, const void* d_workspace,
                              const float* d_gamma, const float* d_beta, float* d_mean_rstd,
                              float* d_running_mean, float* d_running_var, int64_t* d_num_batches,
                              float momentum, float eps, float p_drop, const int64_t* d_seed, int salt,
                              float* d_out, dfm_stream_t stream);
size_t dfm_bn_bwd_workspace_bytes(int64_t batch, int features);
/* logits = (first_order + fm) + (a w^T + b)   (NULL first_order / fm / b count as 0);
 * d_g_logits = (sigmoid - y) / batch; the gradient w.r.t. a goes through `bn` (the BatchNorm block
 * that produced a): bn->dy and partial sums in bn->workspace.  The loss and the head's own
 * gradients are finished by dfm_bn_backward_apply(bn, ..., head).  features % 32 == 0, <= 256. */
int dfm_head_bce(const float* d_a, int64_t batch, int features, const float* d_w, const float* d_b,
                 const float* d_first_order, const float* d_fm, const float* d_labels, float* d_logits,
                 float* d_g_logits, const dfm_bn_bwd* bn, dfm_stream_t stream);
/* dfm_bn_relu_dropout_apply of the tower's LAST block and dfm_head_bce in one launch: every workgroup merges
 * the column statistics in d_fwd_workspace (dfm_linear_bn_forward's) exactly as dfm_bn_relu_dropout_apply does,
 * workgroup 0 writes d_mean_rstd (= bn->mean_rstd) and the running statistics, and
 * a = dropout(relu(gamma * (bn->z - mean) * rstd + beta)) is formed in registers and never stored (the backward
 * needs z, the statistics and the mask).  Results are bit-identical to the two calls it replaces. */
int dfm_head_bn_bce(const void* d_fwd_workspace, float* d_mean_rstd, float* d_running_mean, float* d_running_var,
                    int64_t* d_num_batches, float momentum, float eps, int64_t batch, int features,
                    const float* d_w, const float* d_b, const float* d_first_order, const float* d_fm,
                    const float* d_labels, float* d_logits, float* d_g_logits, const dfm_bn_bwd* bn,
                    dfm_stream_t stream);
/* dz = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)), d gamma / d beta ADDED; d_dz may be
 * bn->dy.  `head` non-NULL iff bn was filled by dfm_head_bce (it then also receives the loss and
 * the head gradients). */
int dfm_bn_backward_apply(const dfm_bn_bwd* bn, int64_t batch, int features, const dfm_head_tail* head,
                          float* d_dz, dfm_stream_t stream);
size_t dfm_linear_backward_workspace_bytes(int64_t batch, int out_features, int in_features);
/* Backward of z = x W^T + b given dz (batch, out): the d weight product dz^T x, split over the
 * batch into partial products left in d_workspace (dfm_linear_backward_finish adds them into the
 * gradient buffers of all layers in one launch), and the gradient
 * w.r.t. x (batch, in) either stored to d_g_x (plus the FM backward when `fm` is given), or pushed
 * through the BatchNorm block that produced x (`bn_below`; d_g_x unused).  The bias gradient is
 * not computed: in front of a training-mode BatchNorm it is identically zero.
 * parts: 1 = d weight only, 2 = d input only, 3 = both in one launch.  The two halves are
 * independent given dz, so a caller may put the d weight half on a side stream: only d input is on
 * the backward's critical path. */
int dfm_linear_backward(const float* d_dz, int64_t batch, int out_features, const float* d_x,
                        int in_features, const float* d_w, float* d_g_x,
                        const dfm_bn_bwd* bn_below, const dfm_fm_bwd* fm, int parts, void* d_workspace,
                        dfm_stream_t stream);
/* A Linear with ONE output that is added to the logit (xDeepFM's cin_linear, xdeepfm.py:41-47).
 * forward: d_out[b] = d_x[b, :] . d_w (+ d_b[0]).  backward: d_g_x[b, :] = d_g[b] * d_w, and the weight
 * gradient as dfm_linear1_backward_splits(batch) slabs of `features` floats in d_workspace — add them with a
 * dfm_slab_ref {workspace, g_w, batch 1, out 1, in features, splits}; the bias gradient is sum(d_g) =
 * dfm_head_tail.g_b2 when d_g is the head's d logits.  features % 4 == 0, <= 1024 (dfm_linear1_supported). */
int dfm_linear1_supported(int features);
int dfm_linear1_forward(const float* d_x, int64_t batch, int features, const float* d_w, const float* d_b,
                        float* d_out, dfm_stream_t stream);
int dfm_linear1_backward_splits(int64_t batch);
int dfm_linear1_backward(const float* d_g, const float* d_x, int64_t batch, int features, const float* d_w,
                         float* d_g_x, void* d_workspace, dfm_stream_t stream);
/* d_g_w (out, in) += sum of the batch-split partial products of one dfm_linear_backward call. */
typedef struct dfm_slab_ref {
  const void* workspace;   /* the d_workspace that call was given */
  float* g_w;              /* (out, in) gradient buffer, ADDED */
  int64_t batch;
  int32_t out_features;
  int32_t in_features;
  int32_t splits;          /* 0: the batch split dfm_linear_backward chose for (batch, out, in); > 0: that many
                            * slabs of out * in floats from any producer (dfm_step_embedding_backward's
                            * batch-split DENSE-field gradients) */
  int32_t reserved;
} dfm_slab_ref;
int dfm_linear_backward_finish(const dfm_slab_ref* refs, int count, dfm_stream_t stream);
/* Number of batch splits (slabs) dfm_linear_backward leaves in its workspace for this shape. */
/* Arithmetic of the DNN tower's GEMMs (dnn.py:45-59): 0 = exact fp32 matrix pipe; 1 = fp32 forward, bf16 x 3 split
 * (fp32 accumulate, 2^-16 per product) inside dfm_linear_backward — ReLU / dropout masks never change; 2 = bf16 x 6
 * (every fp32 operand split exactly into three bf16 values, the six largest partial products kept: 2^-23 per
 * product, i.e. fp32-faithful) forward and backward — the fused training steps then run the tower through the
 * *_x6 / *_planes entry points below; the library only records the choice.
 * Process-wide, explicit; applies to steps built / calls enqueued afterwards (a captured graph keeps what it
 * captured). */
int dfm_tower_set_mode(int mode);
int dfm_tower_get_mode(void);

/* ---- bf16 x 6 tower (csrc/gemm_x6.h, tower.hip): operands of the tower's GEMMs as "planes" ---------------------
 * A plane set holds an fp32 matrix split exactly into three bf16 matrices (x = h + m + l), laid out for one
 * contraction: bf16 [3][G][Rp][8], G = 8 ceil(contraction / 64), Rp = 64 ceil(rows / 64).  "Role F" of a
 * [rows][cols] matrix contracts over its columns (x in z = x W^T; d z in d x = d z W), "role S" over its rows
 * (d z and x in d W = d z^T x; W in d x = d z W).  Buffers are dfm_planes_bytes(rows, contraction) bytes and must be
 * ZERO-FILLED once by the caller (pads are never written and must read as zero). */
size_t dfm_planes_bytes(int64_t rows, int64_t contraction);
int dfm_tower_x6_supported(int64_t batch, int out_features, int in_features);   /* batch % 64, features % 8 */
typedef struct {
  const float* src;        /* [rows][cols] row-major, 16-byte aligned, cols % 4 == 0 */
  int64_t rows, cols;
  void* planes_f;          /* role F: dfm_planes_bytes(rows, cols); cols % 8 == 0; or NULL */
  void* planes_s;          /* role S: dfm_planes_bytes(cols, rows); rows % 8 == 0; or NULL */
} dfm_split_job;
/* Up to 8 matrices in one launch (the tower's weights, once per step after the optimizer). */
int dfm_split_planes(const dfm_split_job* jobs, int count, dfm_stream_t stream);
/* dfm_linear_bn_forward on planes: x either fp32 (d_x, split by the kernel: the first layer's input) or role-F planes
 * of (batch, in_features); W as role-F planes of (out_features, in_features).  Same workspace and statistics. */
int dfm_linear_bn_forward_x6(const float* d_x, int64_t ldx, const void* d_x_planes_f, const void* d_w_planes_f,
                             const float* d_bias, int64_t batch, int out_features, int in_features, float* d_z,
                             void* d_workspace, dfm_stream_t stream);
/* dfm_bn_relu_dropout_apply / dfm_bn_backward_apply that also (d_out / d_dz may be NULL: only) write their result as
 * planes: role F of (batch, features) and role S = planes of (features, batch).  batch % 32 == 0, features % 8 == 0. */
int dfm_bn_relu_dropout_apply_planes(const float* d_z, int64_t batch, int features, const void* d_workspace,
                                     const float* d_gamma, const float* d_beta, float* d_mean_rstd,
                                     float* d_running_mean, float* d_running_var, int64_t* d_num_batches,
                                     float momentum, float eps, float p_drop, const int64_t* d_seed, int salt,
                                     float* d_out, void* d_planes_f, void* d_planes_s, dfm_stream_t stream);
int dfm_bn_backward_apply_planes(const dfm_bn_bwd* bn, int64_t batch, int features, const dfm_head_tail* head,
                                 float* d_dz, void* d_planes_f, void* d_planes_s, dfm_stream_t stream);
/* dfm_linear_backward (both parts) on planes: d z in both roles, x as role-S planes of (in_features, batch) or fp32
 * (d_x, the first layer), W as role-S planes of (in_features, out_features).  Its own batch split:
 * dfm_linear_backward_x6_splits slabs in a workspace of dfm_linear_backward_x6_workspace_bytes (pass the split count
 * in dfm_slab_ref.splits). */
int dfm_linear_backward_x6_splits(int64_t batch, int out_features, int in_features);
size_t dfm_linear_backward_x6_workspace_bytes(int64_t batch, int out_features, int in_features);
int dfm_linear_backward_x6(const void* d_dz_planes_f, const void* d_dz_planes_s, int64_t batch, int out_features,
                           const float* d_x, const void* d_x_planes_s, int in_features, const void* d_w_planes_s,
                           float* d_g_x, const dfm_bn_bwd* bn_below, const dfm_fm_bwd* fm, void* d_workspace,
                           dfm_stream_t stream);

int dfm_linear_backward_splits(int64_t batch, int out_features, int in_features);

/* ---------------------------------------------------------------------------------
 * Grouped launches for the tail of the training step (csrc/step_tail.hip): kernels of the step that
 * do not depend on each other share one dispatch.  Same arithmetic and reduction order as the
 * stand-alone entry points they combine (bit-identical results).
 * ------------------------------------------------------------------------------- */
/* dfm_embedding_backward_dense_fields + dfm_rowgrad_build for a uniform plan in one launch.
 * d_dense_list: device array with the schema positions of the num_dense DENSE fields; dense_x and
 * dense_grads: host arrays indexed by schema position (as dfm_embedding_forward's inputs and
 * dfm_field_grad[]); the remaining arguments as in dfm_rowgrad_build. */
int dfm_step_embedding_backward(const int32_t* d_dense_list, int num_dense, const void* const* dense_x,
                                const dfm_field_grad* dense_grads, const int32_t* field_of_sparse,
                                int num_sparse, int num_fields, int dim, int64_t batch,
                                const float* d_g_first, const float* d_g_field,
                                const int32_t* d_sorted_pos, int32_t* d_seg_start,
                                const int32_t* d_num_uniq, float* d_row_g2, float* d_row_g1,
                                float* d_dense_partials, int dense_parts, const float* d_dense_grad_base,
                                int64_t dense_grad_elems, dfm_stream_t stream);
/* d_dense_partials (optional): instead of adding into the dense_grads buffers, the DENSE-field gradients are
 * computed over dense_parts batch slices (dense_parts x as many workgroups: the single-slice form is a
 * latency chain over the whole batch on 65 workgroups) and slice p's values are STORED at
 * d_dense_partials[p * dense_grad_elems + (address of the element - d_dense_grad_base)]; every dense_grads
 * buffer must lie inside [d_dense_grad_base, + dense_grad_elems).  The slices are added by whoever consumes
 * the dfm_slab_ref {d_dense_partials, d_dense_grad_base, 1, 1, dense_grad_elems, dense_parts}. */
/* dfm_rowadam_merge + dfm_dense_grad_prepare in one launch; `slabs` (optional) are dfm_linear_backward
 * workspaces whose batch-split products are added into their d_g views first (replaces
 * dfm_linear_backward_finish).  d_partials: dfm_step_prepare_num_partials floats, to be summed by
 * dfm_grad_norm_finalize.  d_match (optional, dfm_step_match_bytes): with three or more lists the
 * memberships of every row in the other lists are resolved first by one extra launch through LDS
 * (the merge then does byte reads instead of num_lists - 1 global binary searches per row).
 * d_dense_gathered (optional, data parallel): (world, n) — every rank's dense gradient buffer from
 * the step's exchange, rank r's at d_dense_gathered + r * gathered_stride floats (0 = n: contiguous);
 * d_g is then REPLACED by grad_scale * their sum in rank order (an all-reduce with a fixed summation
 * order and no launch of its own).  dense_partial_offset: where in d_partials the dense buffer's partials
 * start (0 = right behind the dfm_rowadam_num_partials row partials; field-sharded tables pad the row part
 * to the same length on every rank so that it can be all-gathered). */
int64_t dfm_step_prepare_num_partials(int num_sparse, int dim, int num_lists, int64_t n);
size_t dfm_step_match_bytes(int num_sparse, int num_lists);
int dfm_step_prepare(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                     const int32_t* d_uniq_rows, const int32_t* d_num_uniq, float* d_row_g2,
                     float* d_row_g1, int32_t* d_owner_flag, float grad_scale, float l2, float* d_g,
                     const float* d_p, int64_t n, int64_t n_l2, const dfm_slab_ref* slabs, int num_slabs,
                     const float* d_dense_gathered, int world, int64_t gathered_stride, float* d_partials,
                     int64_t dense_partial_offset, void* d_match, dfm_stream_t stream);
/* dfm_rowadam_apply + dfm_dense_adam in one launch. */
int dfm_step_apply(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                   const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                   const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef, float lr,
                   float beta1, float beta2, float eps, const int32_t* d_step, float* d_p, float* d_m,
                   float* d_v, float* d_g, int64_t n, int zero_grad, dfm_stream_t stream);
/* dfm_step_apply of step t + dfm_rowplan_build (with its row touch) of step t + 1 in ONE launch: the plan depends on
 * the next batch's ids only (trainer.py:212-217 hands batches over one by one; a graph of several steps knows them
 * all), and sorts on the first workgroups of the optimizer's last launch instead of costing a ~13 us launch of its
 * own at the head of the next step.  d_next_ids: (num_sparse, batch) int64, column s at d_next_ids + s * ids_stride
 * (a batch record); d_vocab (num_sparse) int32 on the device, max_vocab their maximum; d_next_*: the plan buffers of
 * the NEXT step (not the ones this step's lists live in).  _update: re-point the captured node at another record. */
int dfm_step_apply_plan(const dfm_table* tables, int num_sparse, int dim, int num_lists,
                        const int32_t* d_uniq_rows, const int32_t* d_num_uniq, const float* d_row_g2,
                        const float* d_row_g1, const int32_t* d_owner_flag, const float* d_clip_coef, float lr,
                        float beta1, float beta2, float eps, const int32_t* d_step, float* d_p, float* d_m,
                        float* d_v, float* d_g, int64_t n, int zero_grad, const int64_t* d_next_ids,
                        int64_t ids_stride, const int32_t* d_vocab, int max_vocab, int64_t batch,
                        int32_t* d_next_sorted_pos, int32_t* d_next_uniq_rows, int32_t* d_next_seg_start,
                        int32_t* d_next_num_uniq, int32_t* d_error_flag, dfm_stream_t stream);
int dfm_step_apply_plan_update(void* graph_exec, void* node, const dfm_table* tables, int num_sparse, int dim,
                               int num_lists, const int32_t* d_uniq_rows, const int32_t* d_num_uniq,
                               const float* d_row_g2, const float* d_row_g1, const int32_t* d_owner_flag,
                               const float* d_clip_coef, float lr, float beta1, float beta2, float eps,
                               const int32_t* d_step, float* d_p, float* d_m, float* d_v, float* d_g, int64_t n,
                               int zero_grad, const int64_t* d_next_ids, int64_t ids_stride, const int32_t* d_vocab,
                               int max_vocab, int64_t batch, int32_t* d_next_sorted_pos, int32_t* d_next_uniq_rows,
                               int32_t* d_next_seg_start, int32_t* d_next_num_uniq, int32_t* d_error_flag);

/* ---------------------------------------------------------------------------------
 * Exact-fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32) for the DNN tower's Linear
 * layers (reference dnn.py:45-47):  C[m,n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]).
 * `*_k_contiguous` = 1: element (r,k) at base[r*ld + k];  0: at base[k*ld + r].  Forward
 * (x W^T + b), d input (dz W) and d weight (dz^T x, accumulate = 1) of nn.Linear are this one
 * entry point without transposed copies.  d_workspace (dfm_gemm_workspace_bytes) enables a
 * split reduction with fixed-order slab summation for small outputs with a long k (d weight).
 * ------------------------------------------------------------------------------- */
size_t dfm_gemm_workspace_bytes(int m, int n, int k);
int dfm_gemm_f32(const float* d_a, int64_t lda, int a_k_contiguous, const float* d_b, int64_t ldb,
                 int b_k_contiguous, float* d_c, int64_t ldc, int m, int n, int k,
                 const float* d_bias, int accumulate, void* d_workspace, dfm_stream_t stream);

/* Weight and bias gradient of an nn.Linear over many rows (the attention projections W_q|W_k|W_v and
 * W_out, attention.py:95-97, :115; rows = B*F):  dW[n1,n2] (+)= sum_r g[r,n1] * x[r,n2],
 * db[n1] (+)= sum_r g[r,n1] (d_db may be NULL).  Both operands are streamed once, per-workgroup
 * partials are summed in a fixed order (bitwise reproducible).  Shapes: n1, n2 multiples of 32 with
 * (n1/32, n2/32) in {(6,1),(3,1),(2,1),(1,1),(1,2),(2,2),(1,3)} and rows >= 8192 — otherwise
 * dfm_weight_grad_workspace_bytes returns 0 and the call DFM_ERR_UNSUPPORTED (use dfm_gemm_f32).
 * dfm_gemm_f32 itself routes these shapes (and the many-rows x tiny-weight products
 * C = A W^T with n*k <= 6144) to the same kernels (csrc/gemm_skinny.hip). */
size_t dfm_weight_grad_workspace_bytes(int64_t rows, int n1, int n2);
int dfm_weight_grad_f32(const float* d_g, int64_t ldg, const float* d_x, int64_t ldx, int64_t rows, int n1,
                        int n2, float* d_dw, int64_t lddw, float* d_db, int accumulate, void* d_workspace,
                        dfm_stream_t stream);
/* Deferred finish: dfm_weight_grad_partials_f32 runs the streamed pass only and leaves
 * dfm_weight_grad_partial_blocks(rows) partial rows of n1 * n2 + n1 floats in the workspace; dfm_layernorm_backward
 * with d_g_gamma == d_g_beta == NULL leaves dfm_layernorm_partial_blocks(rows) partial planes [2][dim] in its
 * workspace.  dfm_partials_finish then adds up to 8 such sets in ONE launch (same fixed order as the stand-alone
 * finishes: bit-identical) — an attention block's backward (attention.py:91-120 under autograd) ends in one
 * reduction launch instead of three. */
int dfm_weight_grad_partial_blocks(int64_t rows);
int dfm_weight_grad_partials_f32(const float* d_g, int64_t ldg, const float* d_x, int64_t ldx, int64_t rows, int n1,
                                 int n2, void* d_workspace, dfm_stream_t stream);
/* Two such passes over the same rows in one launch (an attention block's dW_qkv and dW_out); DFM_ERR_UNSUPPORTED when
 * the pair of shapes has no joint kernel ((192, 32) + (32, 64) has one) — make the two calls then. */
int dfm_weight_grad_partials_pair_f32(const float* d_g_a, int64_t ldg_a, const float* d_x_a, int64_t ldx_a, int n1_a,
                                      int n2_a, void* d_workspace_a, const float* d_g_b, int64_t ldg_b,
                                      const float* d_x_b, int64_t ldx_b, int n1_b, int n2_b, void* d_workspace_b,
                                      int64_t rows, dfm_stream_t stream);
int dfm_layernorm_partial_blocks(int64_t rows);
typedef struct {
  int32_t kind;            /* 0: weight gradient partials -> out_w = dW (row stride ldw), out_b = db or NULL;
                            * 1: LayerNorm partials -> out_w = d gamma, out_b = d beta (always accumulated into) */
  int32_t blocks;          /* partial rows */
  int32_t n1, n2;          /* kind 0: dW is n1 x n2; kind 1: n1 = dim */
  int32_t accumulate;      /* kind 0: 0 store, 1 add */
  int32_t reserved;
  const float* partial;
  float* out_w;
  float* out_b;
  int64_t ldw;
} dfm_partial_job;
int dfm_partials_finish(const dfm_partial_job* jobs, int count, dfm_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Field-sharded embedding tables under data parallelism (csrc/shard.hip).  The reference trains on one
 * device (deepfm/training/trainer.py:47-56), so nothing of it is replaced here: these are the device
 * halves of this build's N-GPU step — rank r owns the tables of a contiguous block of
 * SPARSE fields, serves their rows to every rank's batch and receives the gradients back; three
 * all-to-alls per step (ids, rows, gradients) carry plain contiguous buffers:
 *   ids   segment for owner q : the batch's ids of q's fields               (nf_q, batch) int64
 *   rows  segment for rank p  : [e (batch, nf_r, dim) | first-order w (batch, nf_r)]
 *   grads segment for owner q : [d e (batch, nf_q, dim) | d first (batch) | dense gradients (n_dense)]
 * ------------------------------------------------------------------------------- */
/* Copy of a batch record into the step's static inputs as a kernel of its own (the first node of a
 * captured step); _update re-points that node of an instantiated graph at another record (host-side
 * only; see dfm_embedding_forward_staged_update for the rules). */
int dfm_stage_record(const void* d_src, void* d_dst, int64_t nbytes, dfm_stream_t stream);
int dfm_stage_record_update(void* graph_exec, void* node, const void* d_src, void* d_dst, int64_t nbytes);
/* Owner side, forward: d_ids (world, num_owned, batch) as received -> d_send (world segments of
 * batch * num_owned * (dim + 1) floats, layout above) and d_gids (num_owned, world * batch): the same
 * ids field-major, the input of dfm_rowplan_build over the global batch.  tables[j] / vocab[j]: the
 * owned tables in field order (Adam state pointers unused).  Out-of-range ids: row 0 + *d_error_flag |= 1. */
int dfm_shard_gather(const dfm_table* tables, const int32_t* vocab, int num_owned, int dim, int world,
                     int64_t batch, const int64_t* d_ids, float* d_send, int64_t* d_gids,
                     int32_t* d_error_flag, dfm_stream_t stream);
/* Batch side, backward: length in floats of one gradient segment, and the kernel that fills all of
 * them: first_field[q] / field_count[q] = rank q's block of SPARSE fields (indices into
 * field_of_sparse, which maps a SPARSE field to its schema position in d_g_field (batch, num_fields, dim)).
 * `slabs` (optional, as in dfm_step_prepare): dfm_linear_backward workspaces whose batch-split products are
 * added to their weights' part of the dense segment on the way (d_dense itself is not modified). */
int64_t dfm_shard_pack_segment(int64_t batch, int num_owned, int dim, int64_t n_dense);
int dfm_shard_pack(const int32_t* first_field, const int32_t* field_count, int world,
                   const int32_t* field_of_sparse, int num_sparse, int num_fields, int dim, int64_t batch,
                   const float* d_g_field, const float* d_g_first, const float* d_dense, int64_t n_dense,
                   const dfm_slab_ref* slabs, int num_slabs, float* d_send, dfm_stream_t stream);
/* Owner side, backward: dfm_rowgrad_build over the received gradient segments (d_recv: world segments
 * of `segment` floats each, source-rank major; sample p * batch + b of the row plan is sample b of
 * segment p). */
int dfm_shard_rowgrad(int num_owned, int dim, int world, int64_t batch, const float* d_recv, int64_t segment,
                      const int32_t* d_sorted_pos, int32_t* d_seg_start, const int32_t* d_num_uniq,
                      float* d_row_g2, float* d_row_g1, dfm_stream_t stream);
/* *d_out = x[0] + ... + x[n-1] in a fixed order (one workgroup): a rank's share of |g|^2. */
int dfm_sum_floats(const float* d_x, int64_t n, float* d_out, dfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPFM_HIP_H */
