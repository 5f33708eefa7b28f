/*
 * trx_nn.h -- C ABI of libtrxnn.so: the two memory-/latency-bound hot spots of the TextReact
 * predictor as hand-written gfx950 kernels.
 *
 * The reference builds its encoder-decoder from Hugging Face modules (textreact/model.py:21-31:
 * EncoderDecoderModel = BertModel encoder + RobertaForCausalLM decoder, decoder shape from
 * textreact/configs/bert_l6.json); the ops below replace, inside those modules,
 *   - BertSelfAttention / RobertaSelfAttention forward (self-, cross- and causal attention):
 *       softmax(q k^T * scale + mask) v, heads of 64, output already merged to [B, Lq, H*64]
 *   - BertSelfOutput / BertOutput / embeddings / lm_head:  LayerNorm(dense_out + residual)
 * Plain device pointers + sizes + a hipStream_t passed as void*; 0 or a negative TRX_NN_E* code;
 * never throws; trx_nn_last_error() gives the message.  No CPU path: without a device the calls
 * fail.  dtype selects the storage type of activations (math is always fp32).
 */
#ifndef TRX_NN_H
#define TRX_NN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TRX_NN_F32 = 0, TRX_NN_BF16 = 1 };
enum { TRX_NN_OK = 0, TRX_NN_EINVAL = -1, TRX_NN_EHIP = -3 };
enum { TRX_NN_MASK_NONE = 0, TRX_NN_MASK_KEY = 1 /* float [B, Lk] */, TRX_NN_MASK_FULL = 2 /* float [B, Lq, Lk] */ };

/* y = LayerNorm(x + res) * gamma + beta over the last dimension (biased variance, like
 * torch.nn.functional.layer_norm).  res may be NULL (plain LayerNorm).  x, res, y: [rows, cols] of
 * `dtype`; gamma, beta: float[cols]; mean, rstd: float[rows], may be NULL (inference). */
int trx_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, float eps,
                          int64_t rows, int cols, int dtype, void* y, float* mean, float* rstd, void* stream);

/* Backward of the above.  dz = d(loss)/d(x + res) (it is the gradient of both x and res);
 * dgamma, dbeta: float[cols] (overwritten).  ws: float[2 * nblk * cols] workspace with
 * nblk = trx_add_layernorm_bwd_blocks(rows) (deterministic two-stage column reduction). */
int trx_add_layernorm_bwd(const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                          const float* rstd, int64_t rows, int cols, int dtype, void* dz, float* dgamma,
                          float* dbeta, float* ws, void* stream);
int trx_add_layernorm_bwd_blocks(int64_t rows);

/* out[b, i, h*64 + :] = sum_j softmax_j(scale * q[b,i,h,:].k[b,j,h,:] + mask[b,(i),j]) v[b,j,h,:]
 * q: [B, Lq, H, 64], k, v: [B, Lk, H, 64] (the layout a Linear + view gives, no transposes),
 * out: [B, Lq, H*64], all `dtype`.  mask: additive float, see TRX_NN_MASK_*.  causal != 0: key j is
 * visible to query i iff j <= i + (Lk - Lq) (decoder self-attention, with or without a KV prefix). */
int trx_attention_fwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, void* out, void* stream);

/* Inference against a key/value cache (beam-search decoding, main.py:218-226): k and v are the first Lk
 * positions of cache tensors [B, Lmax, H, 64]; kv_batch_stride = Lmax * H * 64 elements between batch items,
 * so no step has to compact the cache first.  Forward only. */
int trx_attention_fwd_kvcache(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int64_t kv_batch_stride, float scale, int dtype, void* out,
                              void* stream);

/* One-token self-attention of beam-search decoding over a cache that is never re-ordered (generate with num_beams > 1,
 * main.py:218-226; Hugging Face re-orders the whole cache by the beams' parents at every step).  kv [n, T, 2, H, 64]
 * bf16: row i holds what beam SLOT i wrote (keys at [.., 0, ..], values at [.., 1, ..]); anc [n, T] int32:
 * anc[i][s] = the slot whose position-s entry belongs to beam i's history (the caller re-orders this table by the
 * parents instead of the cache and sets anc[i][t] = i for the position just written).  *t_dev (device int64) = t, the
 * last filled position: positions 0 .. t are attended.  It is read on the device so that a captured graph can replay
 * the launch at every step.  q [n, H, 64] bf16 with row stride ldq elements; out [n, H*64] bf16.  T <= 256. */
int trx_attention_decode_gather(const void* q, int ldq, const void* kv, const int32_t* anc, const int64_t* t_dev, void* out, int n, int H,
                                int T, float scale, void* stream);

/* Same, additionally writing lse[B, H, Lq] = log sum_j exp(score_ij) (float), which the backward
 * pass needs to recompute the probabilities instead of storing the Lq x Lk matrix. */
int trx_attention_fwd_lse(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                          int B, int H, int Lq, int Lk, float scale, int dtype, void* out, float* lse, void* stream);

/* Backward of trx_attention_fwd: dq [B, Lq, H, 64], dk, dv [B, Lk, H, 64] from dout [B, Lq, H*64],
 * the forward inputs, the forward output `out` and `lse`.  Probabilities are recomputed (flash
 * style); the mask gets no gradient.  Deterministic (no atomics): one pass with a lane per query
 * row for dq, one pass with a lane per key row for dk and dv. */
int trx_attention_bwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, const void* out, const void* dout,
                      const float* lse, void* dq, void* dk, void* dv, void* stream);

/* ---- training mode: dropout ------------------------------------------------------------------------
 * The reference trains with hidden_dropout_prob = attention_probs_dropout_prob = 0.1 (BERT defaults,
 * textreact/configs/bert_l6.json; [3P] modeling_bert.py: BertSelfAttention.dropout on the
 * probabilities, BertSelfOutput / BertOutput / BertEmbeddings.dropout before the residual LayerNorm).
 * Dropout decisions are a pure function of (seed, stream, row, col) -- nothing is stored; every kernel
 * that needs a decision recomputes it, and trx_dropout_keep_mask materialises the same decisions
 * (tests, reference backends).  p in [0, 1); kept elements are scaled by 1 / (1 - p).
 *   LayerNorm ops: y = LayerNorm(dropout(x) + res); index space stream = 0, row, col of x.
 *   attention    : dropout on the softmax probabilities; stream = b * H + h, row = query, col = key. */
int trx_add_layernorm_fwd_dropout(const void* x, const void* res, const float* gamma, const float* beta, float eps,
                                  int64_t rows, int cols, int dtype, float p, uint64_t seed, void* y, float* mean,
                                  float* rstd, void* stream);
/* dz = d(loss)/d(dropout(x) + res) (the gradient of res); dx = dz * keep / (1 - p) (the gradient of x);
 * dx is required when p > 0 and ignored otherwise. */
int trx_add_layernorm_bwd_dropout(const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                                  const float* rstd, int64_t rows, int cols, int dtype, float p, uint64_t seed, void* dz,
                                  void* dx, float* dgamma, float* dbeta, float* ws, void* stream);
int trx_attention_fwd_dropout(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, float scale, int dtype, float p, uint64_t seed, void* out,
                              float* lse, void* stream);
int trx_attention_bwd_dropout(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, float scale, int dtype, float p, uint64_t seed, const void* out,
                              const void* dout, const float* lse, void* dq, void* dk, void* dv, void* stream);
/* Mixed storage, the autocast case: x is a bf16 dense output, the residual stream (res, y) is fp32 --
 * torch runs layer_norm in fp32 under autocast and so does the reference's --precision 16-mixed.
 * cols % 4 == 0, cols <= 1024.  y_bf16 (may be NULL): a bf16 copy of y for the Linear layers that
 * consume it next -- the cast autocast would otherwise run per use.  Backward: dy_f32 (gradient that
 * reached the fp32 y) and dy_bf16 (gradient that reached the bf16 copy), either may be NULL, are summed;
 * dz fp32 (gradient of res), dx bf16 (gradient of x, through the dropout when p > 0), always written. */
int trx_add_layernorm_fwd_mixed(const void* x_bf16, const void* res_f32, const float* gamma, const float* beta, float eps,
                                int64_t rows, int cols, float p, uint64_t seed, void* y_f32, void* y_bf16, float* mean, float* rstd,
                                const float* x_bias, void* stream);
/* x_bias (may be NULL): float[cols] added to x before the dropout -- the bias of the Linear that produced x,
 * kept out of its GEMM so that its gradient (dx_bias = column sums of dx) falls out of this backward instead of
 * a separate reduction over rows.  ws: float[(x_bias ? 3 : 2) * nblk * cols]. */
int trx_add_layernorm_bwd_mixed(const void* dy_f32, const void* dy_bf16, const void* x_bf16, const void* res_f32, const float* gamma,
                                const float* mean, const float* rstd, int64_t rows, int cols, float p, uint64_t seed, void* dz_f32,
                                void* dx_bf16, float* dgamma, float* dbeta, const float* x_bias, float* dx_bias, float* ws,
                                void* stream);
/* dgamma == dbeta == NULL (dx_bias then too): the first stage only -- dz and dx are written, the per-workgroup partial column
 * sums stay in ws.  trx_add_layernorm_bwd_reduce_many finishes any number of such calls in ONE launch: a backward pass makes
 * ~40 of these calls, each followed by a second stage of a few microseconds that nothing waits for before the optimizer.
 * Same sums, bit for bit, as the per-call second stage.  items: HOST array (it travels by value in the kernel arguments: no
 * device copy, capturable); every item's ws must stay alive until the launch has run. */
#define TRX_LN_REDUCE_MAX 48
typedef struct trx_ln_reduce_item {
    const float* ws;   /* the ws of a partials-only trx_add_layernorm_bwd_mixed call: float[(dxbias ? 3 : 2) * nblk * cols] */
    float* dgamma;     /* float[cols] */
    float* dbeta;      /* float[cols] */
    float* dxbias;     /* float[cols] or NULL */
    int nblk;          /* trx_add_layernorm_bwd_blocks(rows) of that call */
    int reserved;
} trx_ln_reduce_item;
int trx_add_layernorm_bwd_reduce_many(const trx_ln_reduce_item* items, int n, int cols, void* stream);
/* Packed projections (bf16, matrix-core kernels): q, k, v (and dq, dk, dv) are slices of one projection
 * output, e.g. [B, L, 3 * H * 64] from a single QKV GEMM.  ldq / ldkv = elements between consecutive
 * tokens of q (dq) and of k, v (dk, dv); out, dout, lse are dense as above. */
int trx_attention_fwd_strided(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, float p, uint64_t seed, void* out,
                              float* lse, void* stream);
int trx_attention_bwd_strided(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, float p, uint64_t seed, const void* out,
                              const void* dout, const float* lse, void* dq, void* dk, void* dv, void* stream);
/* keep[streams][rows][cols] (1 = kept) exactly as the kernels above decide for (seed, p). */
int trx_dropout_keep_mask(uint64_t seed, float p, int64_t streams, int64_t rows, int64_t cols, unsigned char* keep, void* stream);

/* Device-side seed source (process-wide; NULL = off, the default).  While a pointer is set, every dropout launch above
 * passes it along and its kernels use mix(*seed_dev, seed) as their seed: `seed` then only numbers the dropout site, and
 * the uint64 behind the pointer -- which the caller bumps once per optimisation step, on the stream -- makes the
 * decisions differ from step to step while the launch arguments stay the same.  This is what lets one HIP graph of the
 * whole training step (forward, backward, optimizer) be replayed: textreact_amd/predictor/train.py: GraphedStep.
 * trx_dropout_keep_mask follows the same rule, so a reference can still materialise the decisions. */
int trx_nn_set_seed_device(const uint64_t* seed_dev);

/* trx_attention_bwd_dropout / _strided with the caller's scratch for the matrix-core path's per-query scalars
 * (trx_attention_bwd_ws_bytes(B, H, Lq) bytes, device memory; NULL = a stream-ordered allocation inside the call,
 * which a stream capture cannot record).  ldq = ldkv = 0: dense operands; dtype as above. */
int64_t trx_attention_bwd_ws_bytes(int B, int H, int Lq);
int trx_attention_bwd_ws(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                         int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, int dtype, float p, uint64_t seed,
                         const void* out, const void* dout, const float* lse, void* dq, void* dk, void* dv, void* ws, void* stream);

/* C[N, K] = A[M, N]^T . B[M, K]: bf16 operands and result, fp32 accumulation -- the weight gradient of a Linear
 * layer, dW = dY^T X, with the contraction over the M token rows split across workgroups (fp32 partial tiles in
 * ws, summed in a fixed order).  Any M >= 1 (the rows past M of the last 64-row step read as zeros), N % 256 == 0, K % 256 == 0, lda / ldb % 8 == 0, ldc % 4 == 0, A, B, ws
 * 16-byte aligned; anything else returns TRX_NN_EINVAL (-1 from the size query) and the caller keeps its library
 * GEMM.  ws: trx_gemm_tn_ws_bytes(M, N, K) bytes. */
int64_t trx_gemm_tn_ws_bytes(int M, int N, int K);
int trx_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, void* ws, void* C, int ldc, void* colsum,
                     int out_f32, int M, int N, int K, void* stream);
/* colsum (may be NULL): [N] column sums of A over its M rows -- the bias gradient db = sum_rows dY of the same
 * Linear, computed from the A tiles while they are in LDS for dW.  out_f32 != 0: C and colsum are written as fp32
 * (the gradients of fp32 parameters: no rounding and no cast afterwards; C then 16-byte aligned), else bf16. */

/* MANY weight gradients in one persistent launch (dW_i = dY_i^T X_i for every Linear layer of a backward pass: the callers
 * of torch.nn.functional.linear that the reference's model makes through Hugging Face, /root/reference textreact/model.py:21-31,
 * differentiated by autograd at main.py:164-175).  Nothing is split and nothing is reduced: with all of a step's problems in
 * one work list every CU gets whole 256 x 256 tiles, and a tile's fp32 sums go straight to C.  Per problem: A [M, N] (dY),
 * B [M, K] (X), bf16 row-major with leading dimensions lda / ldb; C [N, K] fp32 with ldc; colsum [N] fp32 or NULL (the bias
 * gradient, from the same pass).  The constraints of trx_gemm_tn_bf16 apply to every problem, except that N may be any
 * multiple of 8 (a vocabulary projection).
 * Three steps, so that no memory management hides behind the ABI:
 *   trx_gemm_tn_grouped_block_bytes  size of the plan for these problems (-1: a problem this path does not take)
 *   trx_gemm_tn_grouped_plan         writes the plan (problem table, per-XCD tile lists, zeroed counters) into the caller's
 *                                    HOST memory
 *   trx_gemm_tn_grouped_run          the caller has copied the block to the device (stream-ordered before this call);
 *                                    launches on `stream`.  host_block is read during the call only. */
typedef struct trx_tn_problem {
    const void* A; const void* B; void* C; void* colsum;
    int M, N, K, lda, ldb, ldc;
} trx_tn_problem;
int64_t trx_gemm_tn_grouped_block_bytes(const trx_tn_problem* problems, int n);
int trx_gemm_tn_grouped_plan(const trx_tn_problem* problems, int n, void* host_block, int64_t host_bytes);
int trx_gemm_tn_grouped_run(const void* dev_block, const void* host_block, void* stream);
/* The three steps in one call: the plan is staged in pinned memory the library owns (a ring of event-guarded slots; on a
 * capturing stream a block of its own that is never reused, because the captured copy reads it again at every replay) and
 * copied into dev_block (the caller's device memory, trx_gemm_tn_grouped_block_bytes() bytes, alive until the launch has
 * run) on `stream`. */
int trx_gemm_tn_grouped(const trx_tn_problem* problems, int n, void* dev_block, int64_t dev_bytes, void* stream);

const char* trx_nn_last_error(void);
const char* trx_nn_version(void);

#ifdef __cplusplus
}
#endif
#endif
