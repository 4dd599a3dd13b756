/*
 * oneprot_hip.h -- C ABI of liboneprot_hip.so: the MI355X (gfx950) kernels behind the OneProt contrastive
 * alignment training step.
 *
 * The reference (klemens-floege/oneprot) is pure Python: it has no FFI of its own.  Its "plugin API" for this path
 * is the encoder/loss protocol used by OneProtLitModule (ref src/models/oneprot_module.py:67-108); the arithmetic is
 * delegated to torch/ATen and HF transformers.  Each entry point below replaces one such library call site and cites
 * it ("ref" = /root/reference, "hf" = transformers/models/esm/modeling_esm.py or .../bert/modeling_bert.py).
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers (device memory owned by the caller), sizes, and an opaque stream (hipStream_t passed as void*);
 *   - every function that takes a `stream` only enqueues work on it: no allocation, no synchronisation, graph-capturable (per-launch bookkeeping that
 *     a replayed graph must see advance -- work-queue heads, launch tags -- lives in caller-provided device memory and is advanced by the kernels themselves,
 *     see "sched workspace" below).  The exceptions say so: *_workspace*() / *_eligible() size queries, oneprot_alloc_uncached / oneprot_free_uncached (host
 *     allocation calls, for set-up code), oneprot_gemm_resid_ln8_error (a host-synchronous read for tests and end-of-epoch checks), and the test / tuning hooks;
 *   - return 0 on success, -1 invalid argument / unsupported shape, -2 launch failure.  Nothing throws;
 *   - "bf16" pointers are raw 16-bit bfloat16 storage; statistics, residual stream, losses and all parameter
 *     gradients are fp32;
 *   - workspaces are caller-provided; *_workspace() returns the byte count.
 */
#ifndef ONEPROT_HIP_H
#define ONEPROT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int oneprot_abi_version(void);

/* ---------------- embeddings (hf modeling_esm.py:224-271; ref sequence_encoder.py:78) ---------------------------- */
/* x[b,l,:] = table[id] * 0.88/(1 - n_mask_b/n_valid_b); rows of <mask> and <pad> ids are zero. row_scale[b] (optional) keeps the factor. */
int oneprot_esm_embed_fwd(const int64_t* ids, const float* table, float* x, float* row_scale, int B, int L, int d, int vocab,
                          int pad_id, int mask_id, int token_dropout, void* stream);
size_t oneprot_esm_embed_bwd_workspace(int T, int d, int vocab);
int oneprot_esm_embed_bwd(const int64_t* ids, const float* dx, const float* row_scale, float* dtable, void* workspace, int B, int L, int d,
                          int vocab, int pad_id, int mask_id, int token_dropout, int accumulate, void* stream);
/* BERT: x = LN(word[id] + pos[l] + type[0])  (hf modeling_bert.py:53-108; ref text_encoder.py:59) */
int oneprot_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma, const float* beta,
                           float* x_f32, void* x_bf16, int B, int L, int d, int vocab, float eps, void* stream);

/* ---------------- LayerNorm (nn.LayerNorm: hf modeling_esm.py:429,518,552; ref base_encoder.py:153,158,162) ------ */
int oneprot_layernorm_fwd(const void* x, int x_is_bf16, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                          float* rstd, int64_t T, int d, float eps, void* stream);
size_t oneprot_layernorm_bwd_workspace(int d);
/* dy_mode 0: bf16 [T,d]; 1: fp32 [T,d]; 2: dy[t] = dpool[t/L] * wrow[t].  dx = (add_to ? add_to : 0) + LN'(dy). */
int oneprot_layernorm_bwd(const void* dy, int dy_mode, const float* wrow, int L, const void* x, int x_is_bf16, const float* gamma,
                          const float* mean, const float* rstd, const float* add_to, float* dx, void* dx_bf16 /* optional bf16 copy of dx */,
                          float* dgamma, float* dbeta, void* workspace, int64_t T, int d, int accumulate_param_grads, void* stream);
/* final LayerNorm fused with pooling (ref base_encoder.py:109-126): mode 0 masked mean (CLS/EOS included), 1 CLS. */
int oneprot_lnpool_fwd(const float* x, const int64_t* ids, int pad_id, const float* gamma, const float* beta, float* pooled, float* mean,
                       float* rstd, float* wrow, void* hidden_bf16, float* hidden_f32, int B, int L, int d, float eps, int mode, void* stream);

/* pooling of an already-normalised hidden state (BERT): mode 0 masked mean, 1 CLS (ref base_encoder.py:109-126). */
int oneprot_pool_fwd(const float* x, const int64_t* ids, int pad_id, float* pooled, int B, int L, int d, int mode, void* stream);
/* its backward: g[b,l,:] = dpooled[b,:]/n_b on non-pad tokens (mode 0) or dpooled[b,:] at l=0 (mode 1), zero elsewhere; optional bf16 copy. */
int oneprot_pool_bwd(const float* dpooled, const int64_t* ids, int pad_id, float* g, void* g_bf16, int B, int L, int d, int mode, void* stream);
/* Embedding-table gradient for a large vocabulary (BertEmbeddings.word_embeddings, hf modeling_bert.py:53-108): `perm` = stable argsort of
 * the token ids, `seg_start[s]` / `seg_row[s]` = first sorted position / token id of run s; rows are added in sorted order (deterministic);
 * run with id == skip_row (padding_idx) is skipped; untouched rows must have been zeroed by the caller. */
int oneprot_embed_scatter_sorted(const float* dx, const int64_t* perm, const int64_t* seg_start, const int64_t* seg_row, int64_t n_tokens, int n_seg,
                                 int d, int skip_row, float* dtable, void* stream);
/* out[j] = sum_r x[r,j], fp32 [R,n] (position / token-type embedding gradients: sums over batch, then over positions). */
int oneprot_rowsum_f32(const float* x, float* out, int R, int64_t n, void* stream);

/* Attention1dPooling (ref base_encoder.py:40-103): pooled = sum_l softmax_l(x_l.w + b | padding -> -inf) x_l on an fp32 hidden state [B,L,d];
   attn [B,L] (optional in fwd, required by bwd).  bwd: dw [d], db [1], dx [B,L,d] (optional). */
int oneprot_attnpool_fwd(const float* x, const int64_t* ids, int pad_id, const float* w, const float* bias, float* pooled, float* attn, int B, int L,
                         int d, void* stream);
size_t oneprot_attnpool_bwd_workspace(int B, int d);
int oneprot_attnpool_bwd(const float* x, const float* attn, const float* w, const float* dpooled, float* dw, float* db, float* dx, void* workspace,
                         int B, int L, int d, void* stream);

/* ---------------- dense contractions on MFMA (nn.Linear call sites: hf modeling_esm.py:362-368,399-409,442-463) -- */
enum {
  ONEPROT_EPI_BF16 = 0,        /* out0 bf16 [M,N] = acc (+bias)                                                   */
  ONEPROT_EPI_F32 = 1,         /* out0 fp32 [M,N] = acc (+bias)                                                   */
  ONEPROT_EPI_BIAS_GELU = 2,   /* z = acc+bias; out0 bf16 = gelu_erf(z); out1 u8 [M,N] = gelu_erf'(z) as the code rint(192 g' + 25) (optional, for bwd) */
  ONEPROT_EPI_BIAS_RESID = 3,  /* out0 fp32 = acc + bias + resid fp32 (out0 may alias resid); out1 bf16 copy opt. */
  ONEPROT_EPI_QKV_ROPE = 4,    /* N = 3*H*hd: q=(acc+b)*q_scale -> rope -> out0 [B,H,L,hd]; k -> rope -> out1; v -> out2 */
  ONEPROT_EPI_GELU_BWD = 5     /* out0 bf16 = acc * g', g' = (aux - 25) / 192, aux u8 [M,N] = the codes saved by ONEPROT_EPI_BIAS_GELU  */
};
/* C[M,N] = A[M,K] * B[N,K]^T, A and B bf16 row-major with leading dims lda/ldb (elements), fp32 accumulation. */
int oneprot_gemm_bf16_nt(const void* A, const void* Bw, int64_t M, int N, int K, int lda, int ldb, int epilogue, const float* bias,
                         void* out0, void* out1, void* out2, const void* aux, const float* rope_cos, const float* rope_sin, float q_scale,
                         int L, int H, int hd, void* stream);
/* Full-row form for N = 640 with the FOLLOWING LayerNorm fused into the epilogue (hf modeling_esm.py:399-409 + :429 / :455-463 + :518 of the next
   layer): x_out fp32 [M,640] = A * W^T + bias + resid (x_out may alias resid); h_out bf16 [M,640] = LayerNorm(x_out; gamma, beta, eps); mean / rstd [M]
   (optional) as oneprot_layernorm_fwd would return them.  Wp = the weight packed by oneprot_gemm_ln_pack_weight (same byte count as W).
   Needs N == 640, M % 128 == 0, K % 64 == 0; -1 otherwise (callers then use oneprot_gemm_bf16_nt + oneprot_layernorm_fwd). */
int oneprot_gemm_ln_pack_weight(const void* W_bf16 /* [N,K] row-major */, void* Wp, int N, int K, void* stream);
int oneprot_gemm_bf16_nt_resid_ln(const void* A, const void* Wp, int64_t M, int N, int K, int lda, const float* bias, const float* resid, float* x_out,
                                  const float* gamma, const float* beta, float eps, void* h_out, float* mean, float* rstd, void* stream);
/* ---- sched workspace: device memory through which the persistent one-work-group-per-CU kernels hand out work at run time and keep per-launch
 * bookkeeping on the device (csrc/sched_ws.h: eight per-XCD queue heads, an arrival counter, a launch epoch, a sticky error word, and room for the partial
 * row statistics of oneprot_gemm_bf16_nt_resid_ln8 for up to M_max rows).  The caller owns it: oneprot_sched_workspace_bytes(M_max) bytes, 128-byte aligned,
 * zeroed ONCE (oneprot_sched_workspace_init: a memset enqueued on `stream`), then only ever touched by the kernels -- the last work-group to leave a launch
 * checks that the launch computed exactly the tiles it was given, resets the queue heads and advances the epoch, so a captured graph replays correctly.  One workspace serves one stream (launches that overlap in time must
 * not share one).  The statistics exchange crosses the XCDs' L2s: allocate the workspace with oneprot_alloc_uncached (hipExtMallocWithFlags(
 * hipDeviceMallocUncached); a host call like any allocation -- set-up code, never a launch path) and release it with oneprot_free_uncached. */
size_t oneprot_sched_workspace_bytes(int64_t M_max);
int oneprot_alloc_uncached(void** out, size_t bytes);
int oneprot_free_uncached(void* ptr);
int oneprot_sched_workspace_init(void* sched_ws, size_t bytes, void* stream);
/* Tiles of the persistent GEMM kernels drawn from the work queues of `sched_ws` instead of static per-work-group lists (NULL: static lists, the default).
 * With static lists a work-group that cannot start with the others -- another kernel (an RCCL channel of an overlapped gradient all-reduce, another process)
 * holds its CU -- runs its whole list after the others have finished: 1.47 x per launch whatever the number of CUs held.  With queues the running
 * work-groups share the tiles and late ones find the queue empty.  Results are bit-identical (which CU computes a tile changes no summation order).
 * Process-wide setting; the kernels of every stream then use this one workspace, so the streams must not overlap such launches. */
void oneprot_dynamic_tiles(void* sched_ws, size_t bytes);
/* diagnostic, host-synchronous: ticket draws that had not returned where the kernel first looked for them (each costs one drained operand prefetch; 0 expected) */
int oneprot_sched_late_draws(const void* sched_ws);
/* diagnostic, host-synchronous: launches completed on this workspace = its device-side epoch (-1: read failed) */
int64_t oneprot_sched_epoch(const void* sched_ws);
/* The same product with the row statistics completed ACROSS work-groups, for the launches whose K loop the full-row kernel above runs too slowly
 * (FFN-2, K = 4 d): x_out = resid + A W^T + bias (fp32; may alias resid) and h = LayerNorm(x_out) (bf16), by the 8-phase GEMM on 256 x 320 tiles; the
 * column tiles of a row panel exchange (mean, M2) partials through the sched workspace (hf modeling_esm.py:442-463 followed by :429 of the next layer or by
 * emb_layer_norm_after, sequence_encoder.py:76-81).  W as for oneprot_gemm_bf16_nt (not packed).  stats: fp32 [2][M] = mean | rstd, or NULL.
 * sched_ws / sched_ws_bytes: a sched workspace sized for at least M rows (-1 otherwise); the launch's tag comes from its device-side epoch.
 * oneprot_gemm_resid_ln8_eligible: 0 when (M, N, K) is not made of whole tiles this form serves (M % 256 == 0, N in {320, 640, 1280}, K % 128 == 0) -- the
 * caller runs oneprot_gemm_bf16_nt (ONEPROT_EPI_BIAS_RESID) + oneprot_layernorm_fwd --, 2 when it is and has the >= 192 tiles from which a persistent
 * work-group per CU pays, 1 when it is served but smaller (right, not faster: what the small-batch parity tests run).
 * FAILURE SEMANTICS.  A work-group waits, bounded, for the partials of the other column tiles of its row panel.  A wait that runs out (the partner work-group
 * never got a CU: fewer CUs free than the form needs) sets the workspace's sticky error word and the wave writes NaN into its rows of h, mean and rstd --
 * the next loss is NaN, not plausibly wrong -- and every later wait of that launch gives up at its first miss instead of spinning again.
 * oneprot_clip_coef(…, sched_ws, …) folds the word into the step's gradient norm on the device.  oneprot_gemm_resid_ln8_error: host-synchronous read of the
 * word (0 = clean; 1 = some launch on this workspace wrote NaN rows; 2 = a launch that drew its tiles from the work queues did not compute exactly the tiles it was
 * given -- a queue head that was not zero when it started: another stream sharing the workspace, a launch that never finished); _error_clear: enqueues its reset.  oneprot_gemm_resid_ln8_poll_bound: test hook, polls per wait
 * (default 2^20, about a second; < 0 restores it). */
int oneprot_gemm_resid_ln8_eligible(int64_t M, int N, int K);
int oneprot_gemm_resid_ln8_error(const void* sched_ws);
int oneprot_gemm_resid_ln8_error_clear(void* sched_ws, void* stream);
void oneprot_gemm_resid_ln8_poll_bound(int polls);
int oneprot_gemm_bf16_nt_resid_ln8(const void* A, const void* W, int64_t M, int N, int K, int lda, int ldb, const float* bias, const float* resid,
                                   float* x_out, const float* gamma, const float* beta, float eps, void* h_bf16, float* stats, void* sched_ws,
                                   size_t sched_ws_bytes, void* stream);
/* test / tuning hook: kernel form of oneprot_gemm_bf16_nt_resid_ln -- 0 (default): eight-wave work-groups on 128-row tiles, one per CU; 1: four-wave
   work-groups on 64-row tiles, two per CU (measured 5-9 % slower: kept as a tested variant).  Bit-identical results.  oneprot_gemm_ln_form_get: the current form. */
void oneprot_gemm_ln_form(int form);
int oneprot_gemm_ln_form_get(void);
/* test / tuning hook: force the block shape of oneprot_gemm_bf16_nt (0..5, see csrc/gemm_nt.hip; -1 = heuristic). */
void oneprot_gemm_force_shape(int shape);
/* test / tuning hook: L2 super-tile of the per-tile kernels (sup_m row panels x sup_n column tiles per XCD at a time; <= 0 keeps a value). */
void oneprot_gemm_tune(int sup_m, int sup_n);
/* dW[N,K] (+)= dY[M,N]^T * X[M,K]  (contraction over the M tokens; split over workgroups, fp32 slabs in workspace);
   dbias[N] (+)= column sums of dY (optional, fused: an all-ones MFMA operand in the k-tile-0 workgroups). */
/* workspace bytes for an (N, K) weight gradient, for any M (the split count is capped by M inside the call, never raised). */
size_t oneprot_gemm_bf16_tn_workspace(int N, int K);
/* test / tuning hook: -1 = auto (default: the 8-phase 320 x 128 form where the problem is made of whole tiles, else 2), 0 = 64-token stages x2 (LDS-DMA ring),
   1 = 32-token stages x3, 2 = 64-token stages x2 with register-staged fill, 3 = 8-phase form where eligible (else 2). */
void oneprot_gemm_tn_variant(int v);
/* CUs left to co-resident kernels (the RCCL channels of a gradient all-reduce overlapped with the backward) when oneprot_gemm_bf16_tn cuts the token range into
   one-shot work items: at most (CUs - reserve) of them, so that none has to wait for a CU and run after the others (twice the launch time).  Changes the number of
   token splits, i.e. the (fixed) summation order.  Process-wide; default 0. */
void oneprot_cu_reserve(int cus);
/* workspace_bytes = size of `workspace`; -1 (invalid argument) when it is smaller than oneprot_gemm_bf16_tn_workspace(N, K). */
int oneprot_gemm_bf16_tn(const void* dY, const void* X, int64_t M, int N, int K, int ldy, int ldx, float* dW, float* dbias, void* workspace,
                         size_t workspace_bytes, int accumulate, void* stream);
/* fp32 GEMM for the small head / logits contractions (ref base_encoder.py:155,159,164; loss.py:91-99):
   C[M,N] = alpha * op(A) * op(B) (+ C if accumulate);  transA: A stored [K,M]; transB: B stored [K,N] else [N,K]. */
int oneprot_sgemm(const float* A, const float* B, float* C, int M, int N, int K, int transA, int b_is_kn, float alpha, int accumulate, void* stream);

/* ---------------- attention (hf modeling_esm.py:292-317,340-395) ------------------------------------------------- */
/* q = rotary(projection * hd^-1/2 * log2(e)) (what the QKV_ROPE epilogue writes when it is given q_scale = hd^-1/2 * log2 e: scores are in
   log2 units inside the kernels, so the softmax is exp2 without a multiply), k (rotated), v : bf16 [B,H,L,hd]; key_bias fp32 [B,L] (0 valid,
   -inf-like for padding); ctx bf16 [B*L, H*hd]; lse fp32 [B,H,L] (NATURAL log of the softmax denominator, incl. max). */
int oneprot_attn_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, int hd,
                     void* stream);
/* dctx bf16 [B*L, H*hd] -> dqkv bf16 [B*L, 3*H*hd] = gradient w.r.t. the un-rotated, un-scaled q/k/v projections; q_scale here is the
   natural hd^-1/2 (the log2 e inside the stored q cancels in dq and is divided out of dk). */
size_t oneprot_attn_bwd_workspace(int B, int H, int L);
int oneprot_attn_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse,
                     const float* rope_cos, const float* rope_sin, float q_scale, void* dqkv, void* workspace, int B, int H, int L, int hd,
                     void* stream);
/* The forward with attention-probability dropout (hf modeling_bert.py BertSelfAttention: softmax -> nn.Dropout(attention_probs_dropout_prob) -> @ V;
   the reference leaves it active whenever the text tower is in train mode, text_encoder.py:59).  keep(b, h, q, k) is a pure function of
   (seed, stream_id, b*H+h, q, k) -- one integer hash per element, 16-bit threshold, kept probabilities scaled by the inverse keep probability;
   ctx = (keep * softmax / keep_prob) @ V, lse = that of the undropped softmax.  Same arguments as oneprot_attn_fwd otherwise. */
int oneprot_attn_fwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, void* ctx, float* lse, int B, int H, int L, int hd,
                             float p, uint64_t seed, uint64_t stream_id, void* stream);
/* Its backward (the two split kernels, any hd / L): dV = (keep * P / keep_prob)^T dO, dS = P * (keep * dP / keep_prob - delta), the mask regenerated
   from the same (p, seed, stream_id).  Same arguments as oneprot_attn_bwd otherwise. */
int oneprot_attn_bwd_dropout(const void* q, const void* k, const void* v, const float* key_bias, const void* ctx, const void* dctx, const float* lse,
                             const float* rope_cos, const float* rope_sin, float q_scale, void* dqkv, void* workspace, int B, int H, int L, int hd,
                             float p, uint64_t seed, uint64_t stream_id, void* stream);
/* keep[B][H][L][L] (one byte each, 0 / 1): the mask the call above applies -- for tests and for a reference that is handed the mask. */
int oneprot_attn_dropout_keep(void* keep, int B, int H, int L, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* Test / A-B hook: -1 automatic (default: by sequence length -- the split kernels up to L = 288, the 16-wave fused kernel up to 416, the
   8-wave fused kernel up to 512), 0 the two split kernels (dQ, then dK/dV), 1 the fused short-sequence kernel with 16 waves of 32 keys where
   eligible (L <= 512, hd <= 32: S, P, dP, dS formed once per tile; dQ summed in LDS in a fixed ticket order: deterministic like the split
   kernels), 2 the fused kernel with 8 waves of 64 keys (two key blocks per wave: Q / dO fragments read once per two tiles, one ticketed
   read-add-write of dQ per two tiles, deferred under the next step's MFMA chains). */
void oneprot_attn_force_bwd_path(int path);
/* Test / A-B hook for oneprot_attn_fwd: -1 automatic (default), 0 the round-1 kernel (per-tile running maximum), 1 the kernels without a row
   maximum (persistent LDS-DMA kernel for hd <= 32 and L <= 512, the chunked kernel otherwise; rows whose sums leave [2^-60, 2^60] are repeated
   with the running maximum), 2 the chunked kernel everywhere.  All three produce the same context / LSE up to the bf16 rounding of P. */
void oneprot_attn_force_fwd_path(int path);

/* ---------------- small fp32 feature ops (ref base_encoder.py:6-38, loss.py:103-114, oneprot_module.py:99-101) --- */
int oneprot_gelu_f32(const float* x, float* y, int64_t n, void* stream);
int oneprot_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* y = scale * x / max(||x||, 1e-12) per row; inv_norm[r] saved for backward. */
int oneprot_l2norm_fwd(const float* x, float* y, float* inv_norm, int R, int D, float scale, void* stream);
/* dx = scale*inv_norm*(g - xhat*(xhat.g)), g = dy + l1_coef*sign(y)  (l1_coef = 0.01/(R*D) folds the L1 term's gradient). */
int oneprot_l2norm_bwd(const float* y, const float* dy, const float* inv_norm, float* dx, int R, int D, float scale, float l1_coef, void* stream);
/* Softmax cross-entropy over rows of logits [R,C] with label[r] = r + label_offset.  loss_sum += sum_r (lse_r - logit[r,label]) * row_weight;
   logits are overwritten with dlogits = (softmax - onehot) * row_weight. */
int oneprot_ce_fwd_bwd(float* logits, float* loss_sum, float* row_loss_ws /* R floats */, int R, int C, int label_offset, float row_weight, void* stream);
/* SigLIP block (ref loss.py:229-255) on logits [B,B] = scale*m@s^T: loss_sum += -sum logsigmoid(label*(logit+bias))/B with label +1 on the
   diagonal (-1 everywhere if negative_only); logits are overwritten with dloss/dlogit. */
int oneprot_siglip_fwd_bwd(float* logits, float* loss_sum, float* row_loss_ws /* B floats */, int B, float logit_bias, int negative_only, void* stream);
/* the same with the bias read from device memory (a learnable logit_bias tensor, ref loss.py:243-245; NULL = 0): no host synchronisation */
int oneprot_siglip_fwd_bwd_dev(float* logits, float* loss_sum, float* row_loss_ws /* B floats */, int B, const float* logit_bias_dev, int negative_only, void* stream);
/* retrieval ranks of the diagonal of logits [N,N] (ref retrieval_metric.py:83-102): rank_row[i] = #{j: logits[i][j] > logits[i][i]}, rank_col likewise on columns. */
int oneprot_diag_rank(const float* logits, int* rank_row, int* rank_col, int N, void* stream);
/* sum_abs += coef * sum |x|   (L1 feature regulariser, ref oneprot_module.py:101) */
int oneprot_abs_sum(const float* x, float* out_sum, void* workspace /* oneprot_sumsq_workspace() bytes */, int64_t n, float coef, void* stream);

/* out_sum += coef * sum x[i]*y[i]   (gradient of a tensor logit scale: sum(dlogits * logits) / scale, ref oneprot_module.py:142 feeds `log_logit_scale.exp()`) */
int oneprot_dot_f32(const float* x, const float* y, float* out_sum, void* workspace /* oneprot_sumsq_workspace() bytes */, int64_t n, float coef, void* stream);

/* dx (+)= coef * upstream[0] * sign(x); upstream is a device scalar (NULL = 1). */
int oneprot_l1_bwd(const float* x, float* dx, int64_t n, float coef, const float* upstream, int accumulate, void* stream);
int oneprot_scale_by_device_scalar(float* x, int64_t n, const float* s, void* stream);
/* bias[i] = ids[i]==pad ? -FLT_MAX : 0  (additive key-padding mask, hf masking_utils.create_bidirectional_mask) */
int oneprot_key_padding_bias(const int64_t* ids, float* bias, int64_t n, int pad_id, void* stream);

/* y = dropout(x) on a bf16 operand of n elements (n % 8 == 0), keep probability 1 - p, kept values scaled by the inverse keep probability; the mask
   is a pure function of (seed, stream_id, element index) -- Philox4x32-10 -- and is never stored.  Replaces the `lora_dropout` module peft 0.5.0
   applies to the adapter branch's input (ref src/models/components/sequence_encoder.py:61-74, text_encoder.py:39-52: LoraConfig(lora_dropout=...)).
   The generator is not torch's: same distribution, different stream. */
int oneprot_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* the same on fp32 (hidden-state dropout of the BERT tower: hf modeling_bert.py BertEmbeddings / BertSelfOutput / BertOutput); y may alias x. */
int oneprot_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* y = resid + dropout(x): the dense output's dropout and the residual add that follows it in BertSelfOutput / BertOutput (hf modeling_bert.py) in one pass;
   the same mask as oneprot_dropout_f32 for the same (seed, stream_id); y may alias x or resid. */
int oneprot_dropout_add_f32(const float* x, const float* resid, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* s = resid + dropout(x), then LayerNorm(s), one pass (hf modeling_bert.py:BertSelfOutput / BertOutput.forward: dense -> dropout -> LayerNorm(. + input);
 * the reference runs them whenever the text tower is in train mode, src/models/components/text_encoder.py:56-62).  Same mask as oneprot_dropout_add_f32 for
 * the same (seed, stream_id).  s_out (fp32 [T,d], the sum the LayerNorm backward needs) may be NULL; y_bf16 / y_f32 / mean / rstd as oneprot_layernorm_fwd;
 * d a multiple of 8, d <= 2048. */
int oneprot_dropout_add_layernorm_fwd(const float* x, const float* resid, float* s_out, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                      float* mean, float* rstd, int64_t T, int d, float eps, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* dx += mask(seed, stream_id) * dy / keep: the backward of the call above with the same (p, seed, stream_id), added into an existing bf16 gradient. */
int oneprot_dropout_bwd_add_bf16(const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* the same into an fp32 gradient (the post-LN BERT tower keeps the layer-input gradient in fp32; ref text_encoder.py:39-52). */
int oneprot_dropout_bwd_add_f32(const void* dy, float* dx, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);

/* ---------------- optimiser (torch.optim.Adam, ref configs/model/default.yaml:2-6; clip: oneprot_module.py:106) -- */
/* sumsq[0] += sum x^2 (two-stage deterministic reduction through workspace of oneprot_sumsq_workspace() bytes). */
size_t oneprot_sumsq_workspace(void);
int oneprot_sumsq(const float* x, int64_t n, float* sumsq, void* workspace, void* stream);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); norm_out[0] = sqrt(sumsq[0]) */
int oneprot_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, const void* sched_ws /* optional: its error word turns norm and coef into NaN */, void* stream);
/* Adam step on a flat arena; g is multiplied by grad_scale[0] (device scalar, may be NULL) before use. step >= 1. */
int oneprot_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int step, const float* grad_scale, void* stream);

/* ---------------- casts ------------------------------------------------------------------------------------------ */
int oneprot_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int oneprot_transpose_cast_f32_to_bf16(const float* src, void* dst, int R, int C, void* stream);
/* the same for `count` matrices src + z*src_stride -> dst + z*dst_stride (strides in elements): one launch for a weight of every layer */
int oneprot_transpose_cast_f32_to_bf16_batched(const float* src, void* dst, int R, int C, int64_t src_stride, int64_t dst_stride, int count, void* stream);
size_t oneprot_colsum_workspace(int N);
int oneprot_colsum_bf16(const void* a, float* out, void* workspace, int64_t M, int N, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif
