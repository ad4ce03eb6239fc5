/*
 * hcir.h — C ABI of libhcir.so, the MI355X (gfx950) retrieval hot path.
 *
 * The reference (atunnd/Hair-centric-Image-Retrieval) has no FFI: its hot path is
 * Python calling torch / torchvision / lightly / scikit-learn.  Each entry point
 * below replaces one of those library calls; the reference call site it stands in
 * for is cited as  <file>:<line>  relative to the reference root
 * (HP/ = HairPretraining/).  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add at each site.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch tensor
 *     .data_ptr()); the library allocates nothing and keeps no global state.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *     .cuda_stream); every call is asynchronous on that stream.
 *   - Return value: HCIR_OK (0) or a negative hcir_status.  Nothing throws
 *     across the ABI.  hcir_status_string() maps a code to text.
 *   - Row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 *   - Tie-break of every top-k: score descending, then index ascending.
 */
#ifndef HCIR_H
#define HCIR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HCIR_OK = 0,
  HCIR_ERR_INVALID = -1,      /* bad argument (null pointer, k > n, d % 8 != 0 ...) */
  HCIR_ERR_UNSUPPORTED = -2,  /* valid request this build has no kernel for       */
  HCIR_ERR_LAUNCH = -3,       /* HIP reported a launch error                      */
  HCIR_ERR_WORKSPACE = -4     /* workspace too small                              */
} hcir_status;

typedef enum {
  HCIR_F32 = 0, /* fp32 storage, exact-fp32 MFMA (k-ordered fmaf chain)    */
  HCIR_F16 = 1, /* fp16 storage, fp16 MFMA, fp32 accumulate               */
  HCIR_BF16 = 2 /* bf16 storage, bf16 MFMA, fp32 accumulate               */
} hcir_dtype;

int hcir_version(void);
const char* hcir_status_string(int status);
/* "<16 hex digits: sha256 of the kernel sources this binary was compiled from>[ -D build flags]".  bench.py reports a
 * recorded PMC summary only when it was taken on a binary with this id. */
const char* hcir_build_id(void);

/* ------------------------------------------------------------------ *
 * Row norms.  out[i] = 1 / max(||x_i||_2, eps)   (fp32)
 * Replaces the normalisation half of
 *   sklearn cosine_similarity / cosine_distances   (HP/src/classification_engine.py:80-82,
 *                                                    src/models/hair_encoder.py:193)
 *   embeddings / norm.clamp(min=1e-8)               (HP/src/neg_sampling.py:35)
 * ------------------------------------------------------------------ */
int hcir_row_invnorm(const void* x, int64_t n, int32_t d, int64_t ldx, int dtype,
                     float eps, float* out, void* stream);

/* In-place-capable L2 normalise of fp32 rows, optional second output in fp16:
 *   y = x / max(||x||, eps)            torch.nn.functional.normalize(x, dim=1)
 * (HP/src/classification_engine.py:50,62).  y_f32 and/or y_f16 may be NULL. */
int hcir_l2_normalize(const float* x, int64_t n, int32_t d, float eps, float* y_f32,
                      void* y_f16, void* stream);

/* ------------------------------------------------------------------ *
 * Brute-force cosine / inner-product top-k:  query x gallery.
 *   score[i][j] = <q_i, g_j> * (q_inv_norm ? q_inv_norm[i] : 1)
 *                            * (g_inv_norm ? g_inv_norm[j] : 1)
 *   out_val[i][0..k) = the k largest scores of row i, descending
 *   out_idx[i][0..k) = idx_base + j of those scores (ties: smaller j first)
 * Replaces
 *   KNeighborsClassifier(metric="cosine").kneighbors   HP/src/classification_engine.py:80-82
 *   cosine_similarity + np.argsort[::-1][:top_k]       src/models/hair_encoder.py:193-194
 *   torch.mm + torch.sort (rank-k pick)                HP/src/neg_sampling.py:37,45-51
 * The Q x N score matrix is never materialised.  dtype is the storage type of
 * BOTH q and g.  HCIR_F32 scores are bit-reproducible: each score is one fp32
 * fmaf chain over k in the order documented in DESIGN.md ("sim_topk k-order"),
 * which oracle/knn_oracle.c follows.
 * Requirements: d % 8 == 0, 1 <= k <= min(ng, HCIR_TOPK_MAX), ng < 2^31.
 * ------------------------------------------------------------------ */
#define HCIR_TOPK_MAX 1024
size_t hcir_sim_topk_workspace_bytes(int64_t nq, int64_t ng, int32_t d, int32_t k,
                                     int dtype);
int hcir_sim_topk(const void* q, int64_t nq, const void* g, int64_t ng, int32_t d,
                  int32_t k, int dtype, const float* q_inv_norm,
                  const float* g_inv_norm, int64_t idx_base, float* out_val,
                  int64_t* out_idx, void* workspace, size_t workspace_bytes,
                  void* stream);

/* Exact re-scoring + certification of a candidate list ("exact by verification" search,
 * DESIGN.md §3 "filtered search"; no reference equivalent).  cand_idx/cand_val [nq][kc] come from
 * hcir_sim_topk on a reduced-precision MIRROR of the gallery (kc <= 64, sorted).  Every candidate is
 * re-scored against the fp32 gallery `g` with the HCIR_F32 fmaf chain (bit-identical to
 * hcir_sim_topk(HCIR_F32)), ranked (score desc, index asc) into out_val/out_idx [nq][k], and
 * certified[i] = 1 iff  cand_val[i][kc-1] + err_bound[i] < exact k-th score, i.e. no row outside
 * the candidate set can belong to the exact top-k, given |exact - mirror score| <= err_bound[i].
 * err_bound is a device array [nq], or NULL: then E_i is computed in the kernel from the HOST array
 * mirror_consts = {max_j ||g_j - g~_j||, max_j ||g~_j||, max_j ||g_j||, gamma_d} of an fp16 mirror
 * (derivation: hcir/gallery.py).  The caller re-runs uncertified queries through
 * hcir_sim_topk(HCIR_F32). */
int hcir_topk_refine_f32(const float* q, int64_t nq, const float* g, int64_t ng, int32_t d,
                         const int64_t* cand_idx, const float* cand_val, int32_t kc, int32_t k,
                         int64_t idx_base, const float* q_inv_norm, const float* g_inv_norm,
                         const float* err_bound, const float* mirror_consts, float* out_val,
                         int64_t* out_idx, int32_t* certified, void* stream);

/* Merge `nlists` sorted top-k lists per query into one top-k_out list.
 * vals/idx layout: [nlists][nq][k_in].  Used for the per-shard merge after the
 * RCCL all-gather (no reference equivalent: the reference is single-GPU,
 * SURVEY.md §2.3) and internally by hcir_sim_topk.  Entries with idx < 0 are
 * empty slots. */
int hcir_topk_merge(const float* vals, const int64_t* idx, int32_t nlists, int64_t nq,
                    int32_t k_in, int32_t k_out, float* out_val, int64_t* out_idx,
                    void* stream);

/* ------------------------------------------------------------------ *
 * NT-Xent forward (lightly.loss.NTXentLoss semantics; call site
 * HP/src/pretrain_engine.py:93,725; in-tree restatement
 * experiments/DualViewHair/src/losses/ntxent_loss.py:30-57).
 *   z = [normalize(z0); normalize(z1)]  (2B x D),  logits = z z^T * inv_t,
 *   diagonal removed, positive of row i is (i + B) mod 2B,
 *   loss = mean_i ( logsumexp_{j != i} logits[i][j] - logits[i][pos(i)] ).
 * Outputs: loss (1 float), row_lse[2B] (saved for a backward pass, may be NULL).
 * The 2B x 2B logits are never materialised.  d % 8 == 0.
 * ------------------------------------------------------------------ */
size_t hcir_ntxent_workspace_bytes(int64_t b, int32_t d, int dtype);
int hcir_ntxent_fwd(const void* z0, const void* z1, int64_t b, int32_t d, int dtype,
                    float inv_t, float* loss, float* row_lse, void* workspace,
                    size_t workspace_bytes, void* stream);

/* NT-Xent backward: dL/dz0, dL/dz1 (same dtype as the inputs) for the loss of hcir_ntxent_fwd.
 *   G = (P - Y)/2B,  dU = (1/T)(G + G^T) U,  dz_i = grad_out * (dU_i - (dU_i.u_i) u_i) / ||z_i||
 * W = P + P^T - 2Y is written once as fp16 [2B][2B] into the workspace and dU = W.U runs on
 * hcir_gemm_f16 (fp16 MFMA: gradients carry fp16-level relative error for every input dtype).
 * row_lse: the [2B] vector saved by hcir_ntxent_fwd.  Requirements: d % 8 == 0, b % 4 == 0. */
size_t hcir_ntxent_bwd_workspace_bytes(int64_t b, int32_t d, int dtype);
int hcir_ntxent_bwd(const void* z0, const void* z1, int64_t b, int32_t d, int dtype, float inv_t,
                    const float* row_lse, float grad_out, void* dz0, void* dz1, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * ViT building blocks (fp16 MFMA, fp32 accumulate, fp32 residual stream).
 * ------------------------------------------------------------------ */

/* LayerNorm over the last dim of fp32 rows -> fp16 rows.
 * nn.LayerNorm(D, eps=1e-6)   torchvision EncoderBlock.ln_1/ln_2 (HP/src/main_backbone.py:554)
 *                             models_vit.LayerNorm (HP/src/models_vit.py:23-27) */
int hcir_layernorm_f16(const void* x, int x_dtype /* HCIR_F32 | HCIR_F16 residual stream */,
                       int64_t rows, int32_t d, int64_t ldx, const float* gamma, const float* beta,
                       float eps, void* y_f16, int64_t ldy, void* stream);

/* Epilogues of hcir_gemm_f16:  acc = A[M,K] . W[N,K]^T  (fp32 accumulate) */
typedef enum {
  HCIR_EPI_BIAS_F16 = 0,      /* out_f16 = acc + bias                      (qkv / in_proj)      */
  HCIR_EPI_BIAS_GELU_F16 = 1, /* out_f16 = gelu_erf(acc + bias)            (mlp fc1)            */
  HCIR_EPI_BIAS_RESID_F32 = 2,/* out_f32 += scale[n] * (acc + bias)        (attn proj, mlp fc2; */
                              /*   scale = LayerScale gamma or NULL)                            */
  HCIR_EPI_BIAS_F32 = 3,      /* out_f32 = acc + bias                                           */
  HCIR_EPI_AFFINE_RELU_F16 = 4,/* out_f16 = relu(acc * scale[n] + bias[n])  (proj-head Linear+BN+ReLU, eval) */
  HCIR_EPI_AFFINE_F32 = 5,    /* out_f32 = acc * scale[n] + bias[n]        (proj-head Linear+BN, eval)      */
  HCIR_EPI_BIAS_RESID_F16 = 6 /* out_f16 += scale[n] * (acc + bias)        (fp16 residual stream; the     */
                              /*   product term is formed in fp32 and rounded to fp16, the sum of the  */
                              /*   two fp16 numbers is rounded once)                                   */
} hcir_epilogue;

/* out[M,N] = epilogue(A[M,K] . W[N,K]^T).  A, W fp16 row-major (W is the
 * nn.Linear weight as stored).  Replaces nn.Linear / MultiheadAttention
 * in_proj / out_proj / MLPBlock / timm Mlp  (HP/src/models_vit.py:63,66,70,79;
 * HP/src/main_backbone.py:554 -> torchvision EncoderBlock).
 * Requirements: K % 8 == 0, N % 8 == 0; bias may be NULL.  lda / ldw are row pitches in
 * elements (>= K, multiples of 8); pitches other than K are carried by the persistent
 * 256 x 256 kernel only (M >= 1024, N % 256 == 0, K % 64 == 0), else HCIR_ERR_UNSUPPORTED. */
int hcir_gemm_f16(const void* a, int64_t lda, const void* w, int64_t ldw,
                  const float* bias, const float* scale, int64_t m, int32_t n,
                  int32_t k, int epilogue, void* out, int64_t ldo, void* stream);
/* The *_RESID_* epilogues out of place: out = resid + scale[n] * (acc + bias), resid [M, N] of out's type and row
 * pitch (the residual add of a Block, x = x + attn(...) / x + mlp(...), HP/src/models_vit.py:147-149, when the
 * block's input must survive - the training forward keeps it for the backward).  resid NULL or == out: in place,
 * i.e. hcir_gemm_f16.  Any other epilogue with resid != NULL: HCIR_ERR_INVALID. */
int hcir_gemm_f16_resid(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                        const float* scale, int64_t m, int32_t n, int32_t k, int epilogue, const void* resid,
                        void* out, int64_t ldo, void* stream);

/* fc1 of the TRAINING forward: out_pre = fp16(acc + bias) (the pre-activation the GELU backward needs) and
 * out_act = fp16(gelu(acc + bias)) (the next GEMM's operand) from one pass over the accumulators - the value is
 * formed once in fp32 and rounded twice - instead of a GEMM plus an hcir_gelu_fwd_f16 pass over M x N.
 * (MLPBlock / timm Mlp fc1 + act, HP/src/main_backbone.py:554; HP/src/models_vit.py:19.)  Both outputs fp16 with row
 * pitch ldo.  Persistent kernel only: HCIR_ERR_UNSUPPORTED unless hcir_gemm_fused_supported(m, n, k). */
int hcir_gemm_f16_gelu_dual(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m,
                            int32_t n, int32_t k, void* out_pre, void* out_act, int64_t ldo, void* stream);

/* LayerNorm fused into the GEMMs on either side of it (persistent 256 x 256 kernel only:
 * hcir_gemm_fused_supported(m, n, k) != 0, i.e. M >= 1024, N % 256 == 0, K % 64 == 0).
 * Replaces the  x -> norm1/ln_1 -> qkv   and   x -> norm2/ln_2 -> fc1   pairs of a Block
 * (HP/src/models_vit.py:147-149; torchvision EncoderBlock via HP/src/main_backbone.py:554)
 * without ever writing the normalised tokens:
 *  (1) producer: epilogue HCIR_EPI_BIAS_RESID_F16 with stats_part != NULL also writes, per
 *      stored fp16 row, one (mean, sum of squared deviations from it) pair per 64-feature slice:
 *          stats_part[slice][row][2],  slice < hcir_gemm_stats_slices(n) = n / 64;
 *  (2) hcir_ln_stats_finalize combines the slices (Chan's parallel-variance formula, index order:
 *      deterministic, no E[x^2] - mean^2 cancellation) and writes
 *          ln_stats[row] = (mean, 1 / sqrt(var + eps)),  var biased (/ n_features = 64 * slices);
 *  (3) consumer: epilogue HCIR_EPI_BIAS_F16 / HCIR_EPI_BIAS_GELU_F16 with ln_stats, ln_c1:
 *      A = the RAW residual rows x (fp16), W' = fp16(gamma o W),
 *          out = act( rstd[m] * (A . W'^T - mean[m] * ln_c1[n]) + bias[n] )
 *      where the caller prepares  ln_c1[n] = sum_k W'[n][k]  (of the ROUNDED fp16 values) and
 *      bias[n] = sum_k beta[k] * W[n][k] + b[n]:  algebraically LayerNorm(x) . W^T + b.
 * Exactly one of {ln_stats + ln_c1, stats_part} must be given.  scale as in hcir_gemm_f16. */
int hcir_gemm_fused_supported(int64_t m, int32_t n, int32_t k);
int32_t hcir_gemm_stats_slices(int32_t n);
int hcir_gemm_f16_fused(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                        const float* scale, int64_t m, int32_t n, int32_t k, int epilogue,
                        void* out, int64_t ldo, const float* ln_stats, const float* ln_c1,
                        float* stats_part, void* stream);
int hcir_ln_stats_finalize(const float* stats_part, int32_t slices, int64_t m, int32_t n_features,
                           float eps, float* ln_stats, void* stream);

/* Patch embedding = Conv2d(C, D, kernel=P, stride=P) as an im2col-free MFMA GEMM
 * over the fp32 NCHW image, fused with bias, class token and positional add:
 *   tok[b][0]     = cls + pos_mult * pos[0]
 *   tok[b][1 + p] = W . patch(b,p) + bias + pos_mult * pos[1 + p]
 * (HP/src/main_backbone.py:543-551 with pos_mult = 2, see DESIGN.md "double
 *  positional add"; HP/src/models_vit.py:42,48,229-233 with pos_mult = 1).
 * img fp32 [B][C][H][W]; w fp16 [D][ldw], row = the C*P*P weights of one output feature in
 * (c, ky, kx) order, zero-padded to ldw (a multiple of 64); tok [B][1 + (H/P)(W/P)][D] in
 * tok_dtype (HCIR_F32 or HCIR_F16: the residual-stream storage type).  Any patch size P <= 32
 * (P == 16 takes a vectorised path). */
int hcir_patch_embed(const float* img, int64_t b, int32_t c, int32_t h, int32_t w_px,
                     int32_t p, const void* w_f16, int64_t ldw, const float* bias,
                     const float* cls, const float* pos, float pos_mult, int32_t d, void* tok,
                     int tok_dtype, void* stream);

/* Fused multi-head self-attention forward over packed qkv:
 *   qkv fp16 [B][T][3][H][hd]  (rows of the in_proj / qkv GEMM),
 *   out fp16 [B][nq][H*hd] = softmax((q * scale) k^T) v   per (b, head), for the FIRST nq query
 *   rows of every image (nq = T: all tokens; nq = 1: the class token only, which is all the last
 *   block of an embedding forward needs).
 * K and V of one (b, head) are LDS-resident; QK^T and PV are MFMA 32x32x16
 * tiles; softmax stays in registers.  (HP/src/models_vit.py:69-78;
 * nn.MultiheadAttention inside torchvision EncoderBlock, HP/src/main_backbone.py:554.)
 * Requirements: T <= 288; hd == 64 runs the tuned kernel, hd in {32, 48, 80, 96, 128} (vit_huge_patch14: 80,
 * HP/src/models_vit.py:266-270) a generic one with unswizzled, register-staged LDS images. */
int hcir_attn_fwd(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd,
                  float scale, int32_t nq, void* out, void* stream);

/* Final step of extract_features for the CLS token:
 *   e = ln ? LayerNorm(tok[b][0]) : tok[b][0];  optionally L2-normalised.
 * (torchvision Encoder.ln + x[:,0]  HP/src/main_backbone.py:554,557;
 *  F.normalize  HP/src/classification_engine.py:50.)
 * emb_f32 [B][D]; emb_f16 optional. */
int hcir_cls_head(const void* tok, int tok_dtype, int64_t b, int32_t t, int32_t d,
                  const float* gamma, const float* beta, float eps, int l2_normalize,
                  float* emb_f32, void* emb_f16, void* stream);

/* Mean over patch tokens 1..T-1 after the optional final LayerNorm
 * (pooled_patches of ViTWrapper.forward, HP/src/main_backbone.py:558-561). */
int hcir_patch_mean(const void* tok, int tok_dtype, int64_t b, int32_t t, int32_t d,
                    const float* gamma, const float* beta, float eps, float* out_f32,
                    void* stream);

/* knn_transform on the device (HP/utils/transform.py:10-14): CenterCrop(size) -> ToTensor ->
 * Normalize(mean, std) of a batch of decoded RGB8 images of one size, img [B][H][W][3] uint8 (device)
 * -> out [B][3][size][size] fp32.  Bit-exact with torchvision's arithmetic (u8 / 255, (x - mean) / std,
 * IEEE fp32); images smaller than the window are zero-padded as CenterCrop does.  mean3 / std3 are
 * HOST pointers to 3 floats. */
int hcir_knn_transform_u8(const uint8_t* img, int64_t b, int32_t h, int32_t w, int32_t size,
                          const float* mean3, const float* std3, float* out, void* stream);

/* EMA update of the momentum encoder, every parameter tensor in ONE launch:
 *     ema = ema * m + p * one_minus_m        (fp32, two rounded multiplies + one rounded add: bit-exact with
 *                                             lightly's update_momentum, HP/src/pretrain_engine.py:618-619)
 * dst_ptrs / src_ptrs / counts are DEVICE arrays of n_chunks entries: chunk c covers counts[c] consecutive
 * floats at address dst_ptrs[c] (ema) and src_ptrs[c] (online parameters).  The caller builds the table once
 * per model pair (hcir.momentum) and splits big tensors into chunks of its choice. */
int hcir_ema_update(const uint64_t* dst_ptrs, const uint64_t* src_ptrs, const int64_t* counts,
                    int64_t n_chunks, float m, float one_minus_m, void* stream);

/* PositiveMaskingTransform on the device (HP/utils/transform.py:84-150; the masked positive view of the
 * pretrain step, HP/src/pretrain_engine.py:692-694).  images/out fp32 [B][C][H][W] (may not alias).  A patch
 * (patch x patch, non-overlapping grid) is a "hair" patch when its mean over (C, patch, patch) > threshold;
 * per image int(n_hair * u[b]) hair patches — those with the smallest keys[b][patch] (ties: smaller patch
 * index) — are zeroed.  u [B] and keys [B][(H/patch)*(W/patch)] are DEVICE arrays of the caller's random
 * numbers (u ~ U(mask_ratio_range), keys ~ U(0,1): the reference's uniform_ + randperm as data).
 * n_masked [B] (optional) receives the number of zeroed patches. */
int hcir_positive_masking(const float* images, int64_t b, int32_t c, int32_t h, int32_t w, int32_t patch,
                          float threshold, const float* u, const float* keys, float* out,
                          int32_t* n_masked, void* stream);

/* ------------------------------------------------------------------ *
 * What the reference does with the top-k list, on the device.
 * `ks` is a HOST array of nk <= 16 strictly ascending values, each in [1, kmax].
 * ------------------------------------------------------------------ */

/* KNeighborsClassifier(n_neighbors=k, metric="cosine").predict for EVERY k of the sweep from ONE
 * top-kmax neighbour list (HP/src/classification_engine.py:71,79-82 re-fits and re-scans per k):
 * uniform-weight mode of the first k neighbour labels, smallest label on ties (scipy.stats.mode as sklearn
 * uses it).  nbr_idx [nq][kmax] as written by hcir_sim_topk (idx_base is subtracted), labels [nlabels] in
 * [0, nclass), nclass <= 16384; pred [nk][nq].  *bad (device int, zeroed by the caller) is set when a
 * neighbour slot is empty or a label is out of range. */
int hcir_knn_vote(const int64_t* nbr_idx, int64_t nq, int32_t kmax, int64_t idx_base, const int64_t* labels,
                  int64_t nlabels, int32_t nclass, const int32_t* ks, int32_t nk, int64_t* pred, int32_t* bad,
                  void* stream);

/* sklearn confusion_matrix counts (HP/src/classification_engine.py:85): cm[t][p] += 1 over n samples;
 * cm [nclass][nclass] int32 zeroed by the caller; accuracy_score (:83) is its trace / n. */
int hcir_confusion_matrix(const int64_t* y_true, const int64_t* y_pred, int64_t n, int32_t nclass, int32_t* cm,
                          int32_t* bad, void* stream);

/* Recall@K and AP@K of a retrieved list against per-query ground-truth ids
 * (experiments/DualViewHair/scripts/quantitative_eval.py:194-209):
 *   hit[s][q] = any ground-truth id among the first ks[s] retrieved ids
 *   ap[s][q]  = ( sum over hit ranks i < ks[s] of hits_so_far / (i + 1) ) / min(|gt_q|, ks[s]),  0 if |gt_q| = 0
 * (fp64, summed in rank order like the reference's Python floats).  retrieved [nq][kmax] (ids < 0 = empty),
 * gt [nq][gmax] padded with -1.  recall_mean / map_mean [nk] (optional, both or neither): the means over the
 * queries in index order (:228-234). */
int hcir_retrieval_metrics(const int64_t* retrieved, int64_t nq, int32_t kmax, const int64_t* gt, int32_t gmax,
                           const int32_t* ks, int32_t nk, int32_t* hit, double* ap, double* recall_mean,
                           double* map_mean, void* stream);

/* ------------------------------------------------------------------ *
 * Training side of the HSimCLR step (HP/src/pretrain_engine.py:681-751):
 * the backward of the ViT blocks and the two small losses.  Gradients flow as fp32 along the residual stream
 * and as fp16 GEMM operands; any loss scale (torch GradScaler, :745) passes through linearly.
 * ------------------------------------------------------------------ */

/* nn.GELU() forward / backward on fp16 buffers (n % 8 == 0): h = gelu(u);  du = dh * gelu'(u)
 * (MLPBlock / timm Mlp activation, HP/src/main_backbone.py:554; HP/src/models_vit.py:19). */
int hcir_gelu_fwd_f16(const void* u, int64_t n, void* h, void* stream);
int hcir_gelu_bwd_f16(const void* u, const void* dh, int64_t n, void* du, void* stream);
/* The same backward over an [m, n] matrix (row pitch ld) that also leaves colsum[n] (+)= sum_m du[m][n] - the bias
 * gradient of fc1 - in the same pass; bit-identical to hcir_gelu_bwd_f16 followed by hcir_colsum_f16.
 * workspace: hcir_colsum_chunks(m) * n floats. */
int hcir_gelu_bwd_colsum_f16(const void* u, const void* dh, int64_t m, int32_t n, int64_t ld, void* du, float* colsum,
                             int accumulate, float* workspace, size_t workspace_bytes, void* stream);

/* y_f16 = fp16(a + b)  (b may be NULL): the fp32 residual gradient as the next GEMM's fp16 operand. n % 4 == 0 */
int hcir_add_f32_f16(const float* a, const float* b, int64_t n, void* y, void* stream);

/* nn.LayerNorm(d, eps) backward over `rows` rows (ln_1 / ln_2 / encoder.ln; models_vit norm1 / norm2):
 *   xhat = (x - mean) rstd,  g = dy o gamma,  dx = rstd (g - mean(g) - xhat mean(g o xhat))
 *   dres_out[row] = (dres_in ? dres_in[row] : 0) + dx            fp32, row pitch ldr (dres_out may alias dres_in)
 *   dgamma (+)= sum_rows dy o xhat,  dbeta (+)= sum_rows dy       (accumulate != 0: added to the existing values)
 * x is the layer's INPUT (residual stream, HCIR_F32 or HCIR_F16; statistics are recomputed), dy fp16.
 * workspace: 2 * hcir_layernorm_bwd_blocks(rows) * d floats (two-stage, fixed-order column sums). */
int32_t hcir_layernorm_bwd_blocks(int64_t rows);
int hcir_layernorm_bwd(const void* x, int x_dtype, int64_t rows, int32_t d, int64_t ldx, const void* dy_f16,
                       int64_t lddy, const float* gamma, float eps, const float* dres_in, float* dres_out,
                       int64_t ldr, float* dgamma, float* dbeta, int accumulate, float* workspace,
                       size_t workspace_bytes, void* stream);
/* hcir_layernorm_bwd that also writes dres_f16[row] = fp16(dres_out[row]) (row pitch ldh) - the operand of the next
 * dgrad / wgrad GEMM, otherwise one hcir_add_f32_f16 pass - and, if dres_colsum != NULL, dres_colsum[c] = sum_rows of
 * that fp16 copy - the bias gradient of the Linear layer whose output gradient it is, otherwise one hcir_colsum_f16
 * pass.  dres_f16 NULL: exactly hcir_layernorm_bwd.  workspace: 3 * hcir_layernorm_bwd_blocks(rows) * d floats. */
int hcir_layernorm_bwd_fused(const void* x, int x_dtype, int64_t rows, int32_t d, int64_t ldx, const void* dy_f16,
                             int64_t lddy, const float* gamma, float eps, const float* dres_in, float* dres_out,
                             int64_t ldr, float* dgamma, float* dbeta, int accumulate, void* dres_f16, int64_t ldh,
                             float* dres_colsum, float* workspace, size_t workspace_bytes, void* stream);

/* out[n] (+)= sum_m x[m][n] of an fp16 matrix: the bias gradient of a Linear layer.
 * workspace: hcir_colsum_chunks(m) * n floats. */
int32_t hcir_colsum_chunks(int64_t m);
int hcir_colsum_f16(const void* x, int64_t m, int32_t n, int64_t ldx, float* out, int accumulate, float* workspace,
                    size_t workspace_bytes, void* stream);

/* Weight gradient of a Linear layer:  dw[N][K] (+)= A[M][N]^T . B[M][K]   (A = dY, B = the layer's input), fp16
 * operands, fp32 accumulate, fp32 output.  Both operands are read as stored (row-major, M the slow index): tiles go
 * to LDS by LDS-DMA and the MFMA fragments are read TRANSPOSED (ds_read_b64_tr_b16).  M is split over workgroups;
 * the partial tiles meet in `workspace` (hcir_gemm_f16_tn_workspace_bytes) and are summed in split order
 * (deterministic, no atomics).  Requirements: N % 256 == 0 or N % 128 == 0, K % 128 == 0, lda/ldb % 8 == 0. */
size_t hcir_gemm_f16_tn_workspace_bytes(int64_t m, int32_t n, int32_t k);
int hcir_gemm_f16_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t m, int32_t n, int32_t k,
                     float* dw, int64_t lddw, int accumulate, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Training forward of the attention: hcir_attn_fwd for all T query rows that also writes, per (b, head, query),
 * lse[b][h][q] = log2 sum_k exp2((s_qk - max) scale log2e) + max scale log2e  (log2 domain), from which the
 * backward recomputes P = exp2(s scale log2e - lse). */
int hcir_attn_fwd_lse(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale, void* out,
                      float* lse, void* stream);

/* Backward of hcir_attn_fwd over packed qkv [B][T][3][H][64] (fp16): given the forward output `out` [B][T][H*64],
 * its gradient d_out (fp16) and lse, writes d_qkv [B][T][3][H][64] (fp16):
 *   P = softmax(scale q k^T);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - rowsum(dO o O));
 *   dQ = scale dS K;  dK = scale dS^T Q.
 * Requirements: hd == 64, T <= 256.  (HP/src/models_vit.py:69-78; nn.MultiheadAttention.) */
int hcir_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int64_t b, int32_t t,
                  int32_t h, int32_t hd, float scale, void* d_qkv, void* stream);

/* The same pair for the CLASS-TOKEN query only (token 0 of every image): the last block of a training pass whose loss
 * reads cls_token = x[:, 0] alone (HP/src/main_backbone.py:625-627).  out / d_out fp16 [B][H*64] (compact), lse fp32
 * [B][H]; d_qkv [B][T][3][H][64] receives dK, dV of every token, dQ of token 0 and ZERO for the other tokens' dQ.
 * hd == 64, T <= 256. */
int hcir_attn_cls_fwd_lse(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale, void* out,
                          float* lse, void* stream);
int hcir_attn_cls_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int64_t b, int32_t t,
                      int32_t h, int32_t hd, float scale, void* d_qkv, void* stream);

/* nn.TripletMarginLoss(margin, p=2, eps, reduction='mean') (HP/src/pretrain_engine.py:96-97,717-721), fp32 [B][D]:
 *   d(x, y) = || x - y + eps ||_2,  loss = mean_i max(d(a_i, p_i) - d(a_i, n_i) + margin, 0).
 * row_loss [B] and dist [2][B] are kept for the backward; grad_out is a DEVICE scalar. */
int hcir_triplet_margin_fwd(const float* anchor, const float* positive, const float* negative, int64_t b, int32_t d,
                            float margin, float eps, float* loss, float* row_loss, float* dist, void* stream);
int hcir_triplet_margin_bwd(const float* anchor, const float* positive, const float* negative, int64_t b, int32_t d,
                            float eps, const float* row_loss, const float* dist, const float* grad_out,
                            float* d_anchor, float* d_positive, float* d_negative, void* stream);

/* F.mse_loss(x, y, reduction='mean') (HP/src/pretrain_engine.py:730) and its backward; workspace: 256 floats. */
int hcir_mse_fwd(const float* x, const float* y, int64_t n, float* loss, float* workspace, void* stream);
int hcir_mse_bwd(const float* x, const float* y, int64_t n, const float* grad_out, float* dx, float* dy,
                 void* stream);

/* fp32 -> fp16 / bf16 conversion of a contiguous buffer (gallery upload). */
int hcir_convert_f32(const float* x, int64_t n, int dtype, void* y, void* stream);

/* positive_transform of the pretrain step (HP/utils/transform.py:21-24, applied to the device batch at
 * HP/src/pretrain_engine.py:686): T.RandomRotation((-15, 15)) (nearest neighbour, zero fill) followed by
 * T.GaussianBlur(kernel_size=3) (reflect padding), in one kernel.  torchvision draws ONE angle and ONE sigma per call
 * for a batch tensor; the caller draws them and passes, as HOST arrays, theta4 = {a, b, c, d} of torchvision's inverse
 * affine matrix [[a, b, 0], [c, d, 0]] (= {cos r, sin r, -sin r, cos r}, r = radians(-angle)... computed as
 * torchvision's _get_inverse_affine_matrix does) and taps2 = {k(+-1), k(0)} of the normalised 3-tap Gaussian.
 * images / out fp32 [B][C][H][W], must not alias. */
int hcir_positive_transform(const float* images, int64_t b, int32_t c, int32_t h, int32_t w, const float* theta4,
                            const float* taps2, float* out, void* stream);

/* BatchNorm1d in TRAINING mode over x fp32 [rows][f] (lightly SimCLRProjectionHead, HP/src/main_backbone.py:589):
 * batch mean / biased variance, y = (x - mean) rstd gamma + beta (+ ReLU), running statistics updated with `momentum`
 * and the unbiased variance as torch does (running_* may be NULL).  Outputs y_f32 and / or y_f16 (one may be NULL);
 * save_mean / save_rstd [f] are kept for the backward. */
int hcir_bn1d_fwd(const float* x, int64_t ldx, int64_t rows, int32_t f, const float* gamma, const float* beta,
                  float eps, float momentum, int relu, float* running_mean, float* running_var, float* save_mean,
                  float* save_rstd, float* y_f32, int64_t ldy32, void* y_f16, int64_t ldy16, void* stream);
/* Its backward: dx (fp16 [rows][f], the operand of the dgrad / wgrad GEMMs in front), dgamma, dbeta [f].
 * relu_out_f16 (optional): the forward's fp16 ReLU output; the incoming gradient is zero where it is <= 0.
 * dy_scale (optional DEVICE scalar): dy is multiplied by it first (the caller's power-of-two renormalisation). */
int hcir_bn1d_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int32_t f,
                  const float* gamma, const float* save_mean, const float* save_rstd, const void* relu_out_f16,
                  int64_t ldr, const float* dy_scale, void* dx_f16, int64_t lddx, float* dgamma, float* dbeta,
                  void* stream);

/* ------------------------------------------------------------------ *
 * Baseline-JPEG decode on the device.  Replaces
 *   read_file + torchvision.io.decode_image(img_bytes, mode=RGB)      HP/utils/dataloader.py:28-31
 *   PIL.Image.open(path).convert('RGB')                               src/models/hair_encoder.py:108,169
 * in front of knn_transform's CenterCrop (HP/utils/transform.py:11): only the window is reconstructed.
 * Output bytes equal libjpeg-turbo's default decompressor (islow IDCT, fancy upsampling, its YCbCr
 * tables) — what both reference decoders link.
 *
 * Scope: baseline / extended sequential Huffman, 8-bit, one interleaved scan, grey or YCbCr with luma
 * sampling 1x1 (4:4:4), 2x1 (4:2:2) or 2x2 (4:2:0) and 1x1 chroma, with or without restart intervals.
 * Everything else (progressive, arithmetic, 12-bit, CMYK, 4:4:0, 4:1:1, multi-scan) is
 * HCIR_ERR_UNSUPPORTED from hcir_jpeg_stage and the loader keeps its host decoder for that file.
 *
 * Two steps.  (1) HOST: hcir_jpeg_stage parses the markers of one file into a header (frame geometry,
 * quantisation tables in natural order, derived Huffman tables) and copies the entropy-coded bytes into the
 * caller's (pinned) staging blob with byte stuffing (FF 00) and RSTn markers removed: the copy a loader
 * makes into pinned memory anyway.  The blob for a batch is
 *     [ hcir_jpeg_header x b ][ image 0: stream words | segment table ][ image 1 ... ]
 * with header.stage_offset = byte offset of the image's stream inside the blob.  (2) DEVICE: after ONE
 * H2D copy of the blob, hcir_jpeg_decode_window_u8 runs, per image, a workgroup that Huffman-decodes the
 * stream with one thread per fixed-size subsequence (speculative decode, then the self-synchronisation
 * hand-over of Weissenberger & Schmidt until every subsequence's start state is verified against its
 * predecessor's chain), prefix-sums the DC differences, and two small kernels for IDCT and
 * upsampling + colour conversion of the MCUs the window touches.
 * ------------------------------------------------------------------ */
#define HCIR_JPEG_LOOK_BITS 11
#define HCIR_JPEG_LOOK2 256
/* One decoded symbol, packed: bits 0-4 = bits consumed (code + magnitude bits, 1..31); bits 5-9 = zigzag advance
 * (run + 1 for a coefficient, 16 for ZRL, 1 for a DC symbol; 0 = end of block); bits 10-14 = code length.  0 = no
 * code of at most the indexed length starts with these bits. */
typedef struct hcir_jpeg_lut { /* the part of a table the device keeps in LDS */
  uint16_t look[1 << HCIR_JPEG_LOOK_BITS]; /* indexed by the next LOOK_BITS stream bits                           */
  uint16_t look2[HCIR_JPEG_LOOK2]; /* longer codes, indexed by the low bits of the 16-bit prefix: canonical codes */
  uint32_t base2;                  /* put the long ones at the top of the prefix space, from base2 on             */
  uint32_t use2;                   /* base2 >= 65536 - HCIR_JPEG_LOOK2; else long codes take the range search     */
} hcir_jpeg_lut;
typedef struct hcir_jpeg_hufftab {
  hcir_jpeg_lut lut;
  uint32_t limit[18]; /* limit[l]: first 16-bit left-aligned prefix that is NOT a code of length <= l       */
  int32_t valoff[17]; /* symbol index = (prefix >> (16 - l)) + valoff[l]                                    */
  uint8_t vals[256];
  uint32_t is_ac;
} hcir_jpeg_hufftab;

typedef struct hcir_jpeg_header {
  int32_t width, height, ncomp;
  int32_t hmax, vmax;
  int32_t hs[3], vs[3];         /* sampling factors as decoded (a grey image is 1x1)   */
  int32_t mcus_x, mcus_y, blocks_per_mcu;
  int32_t restart_interval;     /* MCUs per restart segment, 0 = none                   */
  int32_t nsegments;            /* entropy-coded segments (restart intervals), >= 1     */
  uint32_t stream_bits;         /* bits of the staged stream (segments concatenated)    */
  uint32_t stream_words;        /* 32-bit words staged, incl. 3 words of 1-bit padding  */
  uint64_t stage_offset;        /* byte offset of the stream inside the staging blob    */
  uint8_t blk_comp[12];         /* component of block i of an MCU                       */
  uint8_t dc_tab[4], ac_tab[4]; /* per component: index into huff[] (DC 0-1, AC 2-3)    */
  uint16_t quant[3][64];        /* per component, natural (row-major) order             */
  hcir_jpeg_hufftab huff[4];
} hcir_jpeg_header;

/* HOST.  Bytes of staging blob needed for this file's stream + segment table (0 if not a JPEG). */
size_t hcir_jpeg_stage_bytes(const uint8_t* file, size_t nbytes);
/* HOST.  Parse `file`, fill *hdr, write the unstuffed stream (big-endian 32-bit words) and the segment
 * table (uint32 start bit of each segment, nsegments + 1 entries) at blob + blob_offset (multiple of 16);
 * *used = bytes written (multiple of 16).  HCIR_ERR_UNSUPPORTED / HCIR_ERR_INVALID (corrupt) /
 * HCIR_ERR_WORKSPACE (cap too small). */
int hcir_jpeg_stage(const uint8_t* file, size_t nbytes, hcir_jpeg_header* hdr, uint8_t* blob,
                    size_t blob_offset, size_t blob_cap, size_t* used);
/* HOST.  The same for b files into one blob (headers at blob[0 .. b * sizeof(hcir_jpeg_header)), streams behind),
 * `nthreads` worker threads.  status[i] = per-file result; a rejected file gets a zeroed header (width 0), which
 * hcir_jpeg_decode_window_u8 skips (its window stays zero; the loader decodes that file on the host).
 * blob == NULL: only *blob_used (bytes to allocate: an upper bound from the file sizes, the entropy-coded bytes are
 * not read) and status[] are produced.  A blob that is too small: HCIR_ERR_WORKSPACE with *blob_used set - a loader
 * that recycles its (pinned) blobs calls once per batch and re-allocates only then. */
int hcir_jpeg_stage_batch(const uint8_t* const* files, const size_t* nbytes, int64_t b, uint8_t* blob,
                          size_t blob_cap, size_t* blob_used, int32_t* status, int32_t nthreads);
/* HOST.  Workspace bytes for a batch (per-image strides are the maxima over the batch). */
size_t hcir_jpeg_workspace_bytes(const hcir_jpeg_header* hdrs_host, int64_t b, int32_t win_h, int32_t win_w);
/* blob_dev: DEVICE copy of the staging blob (headers first); hdrs_host: the same headers on the host (grid
 * sizing only).  out [b][win_h][win_w][3] uint8 = the CenterCrop((win_h, win_w)) window of every decoded
 * image, zero where the window leaves the image (torchvision pads before cropping).  status_dev [b]
 * int32 (may be NULL): 0, or a negative hcir_status when a stream did not decode to the frame's block count. */
int hcir_jpeg_decode_window_u8(const void* blob_dev, const hcir_jpeg_header* hdrs_host, int64_t b, int32_t win_h,
                               int32_t win_w, uint8_t* out, int32_t* status_dev, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * PNG decode on the device: the format of every hair-region crop the reference lists —
 *   HairPretraining/data/data_train.csv:2-3 ("133101_hair.png", ...: 103 945 of 103 945 rows),
 *   HairPretraining/data/data_{train,test}_combination3.csv, the four PNGs of assets/hair_region_only.
 * Replaces, as the JPEG entry points above do,
 *   read_file + torchvision.io.decode_image(img_bytes, mode=RGB)      HP/utils/dataloader.py:28-31
 *   PIL.Image.open(path).convert('RGB')                               src/models/hair_encoder.py:108,169
 * in front of knn_transform's CenterCrop (HP/utils/transform.py:11).  Output bytes equal libpng's / Pillow's:
 * inflate and the scanline filters are exact integer algorithms (RFC 1950/1951, PNG spec §9).
 *
 * Scope: 8-bit samples, non-interlaced, colour types 0 (grey), 2 (RGB), 3 (palette), 4 (grey + alpha),
 * 6 (RGBA); the RGB conversion is the one `convert('RGB')` / `mode=RGB` make: grey replicated, alpha dropped,
 * palette indices through PLTE.  Everything else (16-bit, 1/2/4-bit, Adam7) is HCIR_ERR_UNSUPPORTED from
 * hcir_png_stage and the loader keeps its host decoder for that file.
 *
 * Two steps, as for JPEG.  (1) HOST: hcir_png_stage walks the chunks (IHDR, PLTE, IDAT ..., optionally verifying
 * every chunk's CRC-32 as Pillow does), fills a header and concatenates the IDAT payloads — the zlib stream —
 * into the caller's (pinned) staging blob:
 *     [ hcir_png_header x b ][ image 0: zlib stream, zero padded to 16 B + 16 B ][ image 1 ... ]
 * (2) DEVICE: hcir_png_decode_window_u8 runs one wavefront per image that inflates the stream up to the last
 * scanline the window needs (stored, fixed and dynamic Huffman blocks; the 32 KB history is an LDS ring; match
 * copies are lane-parallel), then one wavefront per image that undoes the filters of rows 0..last along the
 * anti-diagonals of (row, pixel) — lane = row, so Paeth's left / up / upper-left are a register and two
 * neighbour-lane reads — and writes the window's RGB8 pixels.
 * ------------------------------------------------------------------ */
typedef struct hcir_png_header {
  int32_t width, height;
  int32_t color_type;    /* 0, 2, 3, 4, 6                                                */
  int32_t bpp;           /* bytes per pixel of the filtered scanlines: 1, 3, 1, 2, 4      */
  uint32_t stream_bytes; /* bytes of the zlib stream (IDAT payloads concatenated)         */
  uint32_t reserved;
  uint64_t stage_offset; /* byte offset of the stream inside the staging blob (16-aligned) */
  uint8_t palette[768];  /* PLTE, RGB triples (colour type 3)                             */
} hcir_png_header;

#define HCIR_PNG_VERIFY_CRC 1 /* flags: verify the CRC-32 of every chunk while staging (Pillow's default)   */

/* HOST.  Bytes of staging blob needed for this file's stream (0 if it is not a PNG of the subset above). */
size_t hcir_png_stage_bytes(const uint8_t* file, size_t nbytes);
/* HOST.  Parse `file`, fill *hdr, copy the zlib stream to blob + blob_offset (multiple of 16); *used = bytes
 * written (multiple of 16).  HCIR_ERR_UNSUPPORTED / HCIR_ERR_INVALID (not a PNG, corrupt chunk structure or CRC) /
 * HCIR_ERR_WORKSPACE. */
int hcir_png_stage(const uint8_t* file, size_t nbytes, int32_t flags, hcir_png_header* hdr, uint8_t* blob,
                   size_t blob_offset, size_t blob_cap, size_t* used);
/* HOST.  The same for b files into one blob (headers first), `nthreads` worker threads; status[i] per file, a
 * rejected file gets a zeroed header (width 0) that the device skips.  blob == NULL: only *blob_used and status[]. */
int hcir_png_stage_batch(const uint8_t* const* files, const size_t* nbytes, int64_t b, int32_t flags, uint8_t* blob,
                         size_t blob_cap, size_t* blob_used, int32_t* status, int32_t nthreads);
/* HOST.  Workspace bytes for a batch: the inflated scanlines 0..last needed row of every image. */
size_t hcir_png_workspace_bytes(const hcir_png_header* hdrs_host, int64_t b, int32_t win_h, int32_t win_w);
/* blob_dev: DEVICE copy of the staging blob; hdrs_host: the same headers on the host (sizing only).
 * out [b][win_h][win_w][3] uint8 = the CenterCrop((win_h, win_w)) window of every decoded image, zero where the
 * window leaves the image.  status_dev [b] int32 (may be NULL): 0, or HCIR_ERR_INVALID when the stream is corrupt
 * (bad block type / code set / distance, a filter type > 4, or it ends before the last needed scanline). */
int hcir_png_decode_window_u8(const void* blob_dev, const hcir_png_header* hdrs_host, int64_t b, int32_t win_h,
                              int32_t win_w, uint8_t* out, int32_t* status_dev, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * Resize(224, bicubic) + CenterCrop(224) on the device: the input transform of the hair_retrieval path,
 *   transforms.Resize(224, interpolation=3) -> CenterCrop(224)        src/models/hair_encoder.py:44-48
 * applied in extract_dataset_features (:116-117) and encode_single_image (:175).  torchvision hands a PIL image to
 * Image.resize(size, BICUBIC): Pillow's two-pass separable resampler (third-party, not under /root/reference;
 * Pillow 12.2 src/libImaging/Resample.c): per output pixel a window of the bicubic kernel (a = -0.5) widened by the
 * scale (antialias), coefficients normalised in double and rounded to 22-bit fixed point, horizontal pass first,
 * an 8-bit clipped intermediate, then the vertical pass.  The coefficient tables are computed on the HOST with the
 * same double arithmetic; the two passes run on the device in the same integer arithmetic: bytes equal Pillow's.
 * Only the pixels the crop window needs are computed.
 * ------------------------------------------------------------------ */
/* HOST.  Taps per output pixel (Pillow's ksize) of one axis resampled from in_size to out_size; 0 on bad sizes. */
int32_t hcir_resize_bicubic_ksize(int32_t in_size, int32_t out_size);
/* HOST.  bounds [2 * out_size] = (first source index, tap count) per output index; kk [out_size * ksize] = the
 * 22-bit fixed-point coefficients (precompute_coeffs + normalize_coeffs_8bpc). */
int hcir_resize_bicubic_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk);

typedef struct hcir_resize_job { /* one image of a batch */
  uint64_t src_offset;          /* bytes from `src`: RGB8 rows of the source image                              */
  int64_t src_pitch;            /* bytes between its rows                                                        */
  int32_t src_h, src_w;
  int32_t out_h, out_w;         /* size after the resize (before the crop)                                       */
  int32_t crop_top, crop_left;  /* origin of the window inside the resized image (negative: zero padding)       */
  int32_t coef_h, coef_v;       /* int32 offsets into `coef`: [bounds (2 * out) | kk (out * ksize)] of each axis; */
  int32_t ksize_h, ksize_v;     /* -1 offset: that axis is not resampled (out == in)                             */
} hcir_resize_job;

/* HOST.  Workspace bytes: the horizontal pass's 8-bit intermediate of every image. */
size_t hcir_resize_crop_workspace_bytes(const hcir_resize_job* jobs_host, int64_t b, int32_t win_h, int32_t win_w);
/* src, coef, jobs_dev: DEVICE; jobs_host: the same records on the host (sizing).  out [b][win_h][win_w][3] uint8. */
int hcir_resize_crop_bicubic_u8(const uint8_t* src, const int32_t* coef, const hcir_resize_job* jobs_dev,
                                const hcir_resize_job* jobs_host, int64_t b, int32_t win_h, int32_t win_w,
                                uint8_t* out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HCIR_H */
