/*
 * ltr_mi355x.h -- C ABI of libltr_mi355x.so: the MI355X (gfx950) listwise learning-to-rank hot path.
 *
 * The reference (Haiga/nn-with-pytorch-personalized-losses) has no FFI layer: its boundary for this path
 * is the Python import surface.  Every entry point below therefore replaces one
 * reference *Python function* (file:line given per entry); the Python host code under
 * nn-with-pytorch-personalized-losses_amd/{losses,architeture}/ keeps the reference's names/signatures and
 * binds these symbols through ctypes (see INTEGRATION.md for the binding stub).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless marked host.  Row-major, contiguous, fp32 unless noted.
 *   - `stream` is a hipStream_t passed as void*.  Launchers never allocate, never synchronise, never throw.
 *   - Return value: 0 = launched; < 0 = argument rejected before any launch (LTR_ERR_*); > 0 = hipError_t.
 *   - "slate" = one query's documents; B slates of S documents each; F features per document.
 *   - Per-slate partial results are written; the final mean/sum over slates is a deterministic
 *     fixed-order reduction (ltr_reduce_sum_f32), never a float atomic.
 */
#ifndef LTR_MI355X_H
#define LTR_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTR_ABI_VERSION 1

enum {
    LTR_OK = 0,
    LTR_ERR_NULL = -1,   /* required pointer is NULL                      */
    LTR_ERR_SHAPE = -2,  /* B/S/F/n outside what the kernels support       */
    LTR_ERR_PARAM = -3,  /* bad enum / scalar (scheme id, log base, ...)   */
    LTR_ERR_ALIGN = -4,  /* pointer not aligned as the entry point states  */
    LTR_ERR_IO = -5,     /* (host entry points) file cannot be opened/mapped */
    LTR_ERR_PARSE = -6   /* (host entry points) malformed input line        */
};

/* Largest slate length the loss kernels take (LDS-resident slate state). */
#define LTR_MAX_SLATE 2048

/* lambdaLoss weighing schemes, losses/lambdaL.py:96-127 (string-dispatched there, :46). */
enum {
    LTR_SCHEME_NONE = 0,                 /* weighing_scheme=None                      -> w = 1              */
    LTR_SCHEME_NDCG_LOSS1 = 1,           /* ndcgLoss1_scheme                  :96-97                        */
    LTR_SCHEME_NDCG_LOSS2 = 2,           /* ndcgLoss2_scheme                  :100-106                      */
    LTR_SCHEME_LAMBDA_RANK = 3,          /* lamdbaRank_scheme [sic]           :109-111                      */
    LTR_SCHEME_NDCG_LOSS2PP = 4,         /* ndcgLoss2PP_scheme                :114-115                      */
    LTR_SCHEME_RANKNET = 5,              /* rankNet_scheme                    :118-119                      */
    LTR_SCHEME_RANKNET_GT_DIFF = 6,      /* rankNetWeightedByGTDiff_scheme    :122-123                      */
    LTR_SCHEME_RANKNET_GT_DIFF_POW = 7   /* rankNetWeightedByGTDiffPowed_scheme :126-127                    */
};
enum { LTR_LOG_BINARY = 0, LTR_LOG_NATURAL = 1 }; /* reduction_log, lambdaL.py:52-57 */

int ltr_abi_version(void);
/* Static description of a return code (LTR_ERR_* or hipError_t). Host pointer, never NULL. */
const char *ltr_error_string(int code);

/* out[0] = scale * sum(in[0..n)), fixed summation order (bit-reproducible). */
int ltr_reduce_sum_f32(const float *in, int64_t n, float scale, float *out, void *stream);

/* ---- approxNDCGLoss(y_pred, y_true, eps, padded_value_indicator, alpha)   losses/approxNDCG.py:7-53
 * One pass, one workgroup-slice per slate: soft ranks from pairwise sigmoids in LDS, forward and
 * analytic backward together.
 *   slate_loss[b] = -sum_i G_i / log2(1 + pos_i)            (caller averages: loss = mean_b, :53)
 *   dscores[b,i]  = grad_scale * d slate_loss[b] / d scores[b,i]   (NULL: forward only)
 * labels == pad marks padding (:22-24).  grad_scale is normally 1/B_global. */
int ltr_approxndcg_fwd_bwd(const float *scores, const float *labels, int B, int S, float alpha, float eps,
                           float pad, float grad_scale, float *slate_loss, float *dscores, void *stream);

/* ---- listnetLoss(y_true, y_predicted, apply_sigmoid)                        losses/listnet.py:5-16
 *   slate_loss[b] = -sum_i p_i log q_i   (p = softmax(y_true), q = softmax(y_pred); caller SUMS, :16)
 *   dscores[b,i]  = grad_scale * (q_i sum(p) - p_i);  apply_sigmoid != 0: the :13-15 variant. */
int ltr_listnet_fwd_bwd(const float *y_true, const float *y_pred, int B, int S, int apply_sigmoid,
                        float grad_scale, float *slate_loss, float *dscores, void *stream);

/* ---- lambdaLoss(...)                                                   losses/lambdaL.py:67-93 (+7-64)
 *   slate_loss[b]  = -sum over kept pairs of log_b(clamp(clamp(sigmoid(sigma d), eps)^w, eps))
 *   slate_count[b] = number of kept pairs (for reduction="mean": loss = sum(slate_loss)/sum(count))
 *   dscores[b,i]   = grad_scale * d slate_loss[b] / d scores[b,i]  (NULL: forward only)
 * k <= 0 means k=None (no truncation).  Ranks by counting, ties by index. */
int ltr_lambda_fwd_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                       float mu, float eps, float pad, int log_base, float grad_scale, float *slate_loss,
                       float *slate_count, float *dscores, void *stream);

/* ---- lambdaMask(..., return_losses=True)                                losses/lambdaL.py:7-60
 * losses[b, ri, rj] for ALL pairs in predicted-rank order (ri, rj = 0-based ranks), optional keep mask
 * (the boolean mask of :62, 1 byte per pair) and rank[b, i] = predicted rank of document i. */
int ltr_lambda_pairs_fwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                         float mu, float eps, float pad, int log_base, float *losses, uint8_t *keep,
                         int32_t *rank, void *stream);
/* Backward of the above: dscores[b,i] = sum_{pairs} grad_losses[b,ri,rj] * d losses[b,ri,rj] / d scores[b,i]. */
int ltr_lambda_pairs_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                         float mu, float eps, float pad, int log_base, const float *grad_losses,
                         float *dscores, void *stream);

/* ---- torch.sum(lambdaMask(..., return_losses=True), dim=1)        losses/riskLosses/riskLosses.py:72-83,192-203,302-310
 * The only thing the risk losses take from the pair matrix is its column sums c[b, rj] = sum_ri losses[b, ri, rj]
 * (rj = 0-based predicted rank): computed without materialising [B,S,S].  The backward takes d L / d c[B,S]. */
int ltr_lambda_colsum_fwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, float *colsum, void *stream);
int ltr_lambda_colsum_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, const float *grad_colsum, float *dscores,
                          void *stream);
/* The same column sums for EVERY system of a Lambda-type risk loss in ONE launch (losses/riskLosses/riskLosses.py:63-83, :183-203,
 * :294-310): system 0 = the model (y_pred), 1..n_base = the baseline rankers (y_base [B][S][n_base], NULL when n_base == 0),
 * n_base + 1 = the ideal ranking (the labels as scores).  The label vector and every score vector are soft-maxed over the slate first
 * (:65-70, torch.softmax(dim=1)) inside the kernel.  colsum [n_base + 2][B][S]. */
int ltr_lambda_colsum_sys_fwd(const float *y_pred, const float *y_true, const float *y_base, int B, int S, int n_base, int scheme,
                              int k, float sigma, float mu, float eps, float pad, int log_base, float *colsum, void *stream);
/* Backward of system 0 w.r.t. the RAW y_pred (pair backward on the soft-maxed vectors, then the softmax's Jacobian):
 * grad_colsum [B][S] = d L / d colsum[0] -> dy_pred [B][S]. */
int ltr_lambda_colsum_sys_bwd(const float *y_pred, const float *y_true, int B, int S, int scheme, int k, float sigma, float mu,
                              float eps, float pad, int log_base, const float *grad_colsum, float *dy_pred, void *stream);

/* ---- zRisk / geoRisk(mat, alpha, requires_grad, i)          losses/riskLosses/riskFunctions.py:4-22 / :25-33
 * mat[Q][n_systems]: effectiveness of every system (column) on every query (row); col = the system under test
 * (negative counts from the end, like the reference's i=-1).
 *   LTR_RISK_Z  : value[0] = sum_q d_q (1 + alpha [d_q < 0]),  d_q = (mat[q,col] - e_q)/sqrt(e_q),
 *                 e_q = (sum_q mat[q,col]) (sum_j mat[q,j]) / sum(mat)
 *   LTR_RISK_GEO: value[0] = sqrt(mean_q mat[q,col] * Phi(zRisk / Q))
 *   dmat[Q][n_systems] (NULL: forward only) = d value / d mat, analytic (the [d_q < 0] indicator carries none).
 * One workgroup, fp64 accumulation, fixed summation order. */
enum { LTR_RISK_Z = 0, LTR_RISK_GEO = 1,
       LTR_RISK_ZERO_GUARD = 4 };  /* OR-ed in: e_q == 0 contributes 0 and the (1 + alpha) weight keys on the raw residual --
                                      the numpy metric's rules, utils/metrics.py:26-36 (getGeoRiskDefault) */
int ltr_risk_fwd_bwd(const float *mat, int Q, int n_systems, int col, float alpha, int kind, float *value,
                     float *dmat, void *stream);

/* ---- tRisk tail                                            losses/riskLosses/riskLosses.py:278-291, :332-345
 *   delta_q = (model[q] - baseline[q]) (1 + alpha [model[q] < baseline[q]]);  value[0] = mean(delta)/std(delta)
 *   (unbiased std, torch.std);  dmodel / dbaseline [Q] (may be NULL) = d value / d model, d baseline. */
int ltr_trisk_fwd_bwd(const float *model, const float *baseline, int Q, float alpha, float *value, float *dmodel,
                      float *dbaseline, void *stream);

/* The tail of a geoRisk (kind 1) / zRisk (kind 0) loss in one launch (riskLosses.py:47-60, :118-125, :170-180, :237-244): optional
 * flip mat' = -mat + max(mat), risk of column 0 and (strategies 2 / 3) of the last column, strategy 1: f R0, 2: f (R1 - R0)
 * [zquirk != 0: f R1 - R0, the reference's precedence in zRiskListnetLoss :176], 3: f (R1 - R0)^2; value [1] and dmat [Q][n] =
 * d value / d mat (NULL = not wanted; the max passes its gradient to the maximal entries, evenly among ties). */
int ltr_risk_tail_fwd_bwd(const float *mat, int Q, int n_systems, float alpha, int kind, int strategy, int flip, float factor,
                          int zquirk, float *value, float *dmat, void *stream);

/* The tail of a tRisk loss in one launch (riskLosses.py:269-291, :332-345): mat [Q][2] = (model, baseline) per query; optional flip
 * mat' = -mat + max(mat) (transformation 1); value [1] = factor * mean(delta) / std(delta) as ltr_trisk_fwd_bwd; dmat [Q][2] (NULL = not
 * wanted) = d value / d mat. */
int ltr_trisk_tail_fwd_bwd(const float *mat, int Q, float alpha, int flip, float factor, float *value, float *dmat, void *stream);

/* The [queries x systems] effectiveness matrix of the six risk-sensitive losses in ONE launch (losses/riskLosses/riskLosses.py:8-49,
 * :63-117, :128-169, :183-236, :247-276, :294-330), mat [B][1 + n_rest + (ideal != 0)] row-major: column 0 the model, then the
 * baseline rankers, optionally the ideal ranking (the reference vector itself); and jac [B][S] = d mat[b][0] / d x0[b][j] (NULL = not
 * wanted) -- the only gradient the losses need.
 *   mode 0 (Listnet type): ref = y_true, x0 = y_predicted, rest = y_baselines [B][S][n_rest]; each vector is soft-maxed over the slate
 *     (:10-12), then lt 1: sum (t p - t^2)^2, 2: cosine(t, p), 3: (sum t p - sum t^2)^2.
 *   mode 1 (Lambda type): ref, x0, rest [n_rest][B][S] are lambdaMask column sums (ltr_lambda_colsum_fwd), taken as they are;
 *     lt 1: sum (x - t)^2, 2: cosine(t, x), 3: (sum x - sum t)^2.
 *   mode 2 (tRiskListnetLoss, :247-276): as mode 0 except lt 2 = cosine(t^2, t p), the cosine of the products.
 * The caller applies the `-mat + max(mat)` flip of lt 1 / 3 (:47-49; a whole-matrix maximum).  S <= 2048. */
int ltr_risk_matrix_fwd(const float *ref, const float *x0, const float *rest, int B, int S, int n_rest, int mode, int lt, int ideal,
                        float *mat, float *jac, void *stream);

/* ---- mNdcg / ndcg / dcg / torchNdcg                                            utils/metrics.py:48-104
 * Per-query NDCG@k on the device (the reference loops over queries in Python after every epoch,
 * main_batch_execution.py:173-200).  y_true, y_score: [Q][S] fp32.  k is clamped to S (:54-55).
 *   gains: LTR_GAINS_LINEAR (y) | LTR_GAINS_EXPONENTIAL (2^y - 1)                                   (:57-62)
 *   no_relevant != 0: a query whose ideal DCG is 0 scores 1.0, else 0.0                              (:70-71)
 *   reverse_ties == 0: tied scores rank lower index first (Python's stable sorted(reverse=True), :51 -- the default
 *   path); != 0: higher index first (np.argsort(...)[::-1] under a stable argsort, :53).
 * ndcg[Q] and dcg[Q] (either may be NULL) are fp64, like the reference's numpy results; dcg = the un-normalised
 * DCG@k of :48-65.  torchNdcg(k) (:83-104) = exponential, no_relevant = 0. */
enum { LTR_GAINS_LINEAR = 0, LTR_GAINS_EXPONENTIAL = 1 };
int ltr_ndcg_at_k(const float *y_true, const float *y_score, int Q, int S, int k, int gains, int no_relevant,
                  int reverse_ties, double *ndcg, double *dcg, void *stream);

/* ---- ordinalLoss(y_pred[B,S,n], y_true[B,S], n, padded_value_indicator)  losses/ordinal.py:27-53
 * n_docs = B*S documents, n ordinal probabilities each.  Targets 1[y >= k] are built with the default
 * indicator -1 (ordinal.py:39), then entries whose target == pad are masked (:41-45).
 *   sums[0] = sum of masked BCE terms, sums[1] = number of documents with >= 1 unmasked target
 *   dpred[doc,k] = (p - t) / max((1-p) p, 1e-12), 0 where masked  (caller scales by 1/sums[1])
 * block_partials: workspace of 2*ltr_ordinal_num_blocks(n_docs) floats. */
int64_t ltr_ordinal_num_blocks(int64_t n_docs);
int ltr_ordinal_fwd_bwd(const float *y_pred, const float *y_true, int64_t n_docs, int n, float pad,
                        float *block_partials, float *sums, float *dpred, void *stream);

/* =====================================================================================================
 * FC scorers and the fused slate pipeline (scorer forward -> listwise loss -> scorer backward -> dW).
 *
 *   LTR_NET_DOUBLE  architeture/doubleLayer.py:54-73  fc3(drop(relu(fc2(drop(relu(fc1 x))))))  136-136-136-1
 *   LTR_NET_TRIPLE  architeture/tripleLayer.py:5-17   l3(sigmoid(l2(l1 x)))                     136-64-32-1
 *
 * Parameters are handed over exactly as the nn.Module holds them (nn.Linear: weight [out][in], bias [out]);
 * ltr_mlp_pack turns them into lane-ordered MFMA fragments once per optimizer step.  X is [n_docs][F]
 * fp32, 16-byte aligned, documents of a slate contiguous ([B][S][F] viewed as [B*S][F]).
 * Gradients come back as ONE flat fp32 buffer in nn.Module.parameters() order
 *   [W1 (H1*F) | b1 (H1) | W2 (H2*H1) | b2 (H2) | w3 (H2) | b3 (1)]
 * -- the buffer the data-parallel all-reduce runs on.  No gradient w.r.t. X is produced (the reference
 * computes one only because its drivers set requires_grad on the data, main_batch_execution.py:79).
 */
enum { LTR_NET_DOUBLE = 0, LTR_NET_TRIPLE = 1,         /* 136 input features (MSLR-WEB10K/30K) */
       LTR_NET_DOUBLE_64 = 2, LTR_NET_TRIPLE_64 = 3,    /* the same classes on 64 features (TD2003): 64-64-64-1, 64-64-32-1 */
       LTR_NET_TWO_LAYER_64H = 4,    /* 136 -> 64 -> 1, ReLU: the commented-out two-Linear DoubleLayerNet variant of
                                        architeture/doubleLayer.py:38-51 (BASELINE.json configs[0]); benchmark use.  No fc2:
                                        W2 / b2 are NULL in ltr_mlp_pack and the flat gradient is [W1 | b1 | w3 | b3].
                                        */
       LTR_NET_TRIPLE_FOLDED = 5,  /* TripleLayerNet with l2 . l1 folded into one 136 -> 32 sigmoid layer (tripleLayer.py:14-16 has
                                        no activation between them): a two-layer net [64 x 136 | 64 | 64 | 1] whose rows 32..63
                                        repeat rows 0..31 (each copy runs on half of a tile's documents).  Parameters from
                                        ltr_triple_fold, gradients back through ltr_triple_unfold_grads.  ltr_fused_step* only. */
       LTR_NET_TRIPLE_FOLDED_32 = 6,   /* the same folded network as a plain two-layer net [32 x 136 | 32 | 32 | 1] (copies = 1) for
                                        ltr_mlp_forward / _backward / _forward_save / _backward_saved. */
       LTR_NET_TRIPLE_FOLDED_32_64 = 7 }; /* TripleLayerNet on 64 features (TD2003) folded: [32 x 64 | 32 | 32 | 1], every entry
                                        point (its fused step runs on the generic pipeline). */
enum { LTR_LOSS_APPROXNDCG = 0, LTR_LOSS_LISTNET = 1 };

/* TripleLayerNet (F input features) -> its folded two-layer form, fp64 accumulation; R = 32 copies rows (copies = 2: LTR_NET_TRIPLE_FOLDED,
 * copies = 1: LTR_NET_TRIPLE_FOLDED_32 / _32_64):
 *   W1e [R][F]: rows c 32 + u = (W2 W1)[u];  b1e [R] = (W2 b1 + b2)[u];  w3e [R] = w3[u]  (u < 32, c < copies).
 * Once per step (weights change every step): 32 x 64 x (F + 1) multiply-adds. */
int ltr_triple_fold(const float *W1, const float *b1, const float *W2, const float *b2, const float *w3, int F, int copies, float *W1e,
                    float *b1e, float *w3e, void *stream);
/* The flat gradient of the folded net, g2 = [dW1e R x F | db1e R | dw3e R | db3], -> TripleLayerNet's flat gradient
 * [dW1 64 x F | db1 64 | dW2 32 x 64 | db2 32 | dw3 32 | db3]: with G[u] = sum_c dW1e[c 32 + u], gb[u] = sum_c db1e[c 32 + u],
 *   dW1 = W2^T G, db1 = W2^T gb, dW2 = G W1^T + gb b1^T, db2 = gb, dw3[u] = sum_c dw3e[c 32 + u].  fp64 accumulation. */
int ltr_triple_unfold_grads(const float *g2, int F, int copies, const float *W1, const float *b1, const float *W2, float *flat,
                            void *stream);

/* info[0..7] = F, H1, H2, n_params, packed_floats, partial_floats (per workgroup), docs_per_tile, lds_bytes */
int ltr_net_info(int net, int32_t *info);
/* Number of persistent workgroups the fused step of `net` wants on a device with n_cus compute units (one per CU; two for
 * kernels that run two 256-thread workgroups per CU).  `partials` must hold that many * partial_floats floats. */
int ltr_fused_grid(int net, int n_cus);

int ltr_mlp_pack(int net, const float *W1, const float *b1, const float *W2, const float *b2, const float *w3,
                 const float *b3, float *packed, void *stream);
/* Narrower networks on a compiled geometry (the reference's constructors take any input size: DoubleLayerNet(input_size) is
 * input_size -> input_size -> input_size -> 1, doubleLayer.py:55-60; TripleLayerNet(N_features) is N_features -> 64 -> 32 -> 1,
 * tripleLayer.py:6-10): f / h1 / h2 are the LOGICAL widths of the parameter tensors handed in (W1 [h1][f], W2 [h2][h1], w3 [h2]),
 * each <= the compiled width; the padded inputs / hidden units get zero weights.  The caller pads X rows with zeros to the
 * compiled feature count.  ltr_mlp_reduce_grads_sub writes the flat gradient in the LOGICAL layout
 * [W1 (h1*f) | b1 (h1) | W2 (h2*h1) | b2 (h2) | w3 (h2) | b3 (1)]. */
int ltr_mlp_pack_sub(int net, int f, int h1, int h2, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *w3, const float *b3, float *packed, void *stream);
int ltr_mlp_reduce_grads_sub(int net, int f, int h1, int h2, const float *partials, int grid, float *flat_grad, void *stream);

/* Scorer forward: scores[n_docs] = net(X).  dropout != 0 applies training-mode Dropout(0.5) after each ReLU
 * (doubleLayer.py:60-65) from the counter-based stream (seed, document, feature); keep1/keep2, when non-NULL,
 * are explicit keep masks [n_docs][H1] / [n_docs][H2] (bytes, != 0 keeps) used INSTEAD of the generator
 * (parity tests).  grid = number of persistent workgroups (normally the CU count). */
int ltr_mlp_forward(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                    const uint8_t *keep1, const uint8_t *keep2, float *scores, int grid, void *stream);

/* out[doc][n] (bytes, n < H) = keep bit of dropout layer `layer` (0 after fc1, 1 after fc2) that the pipeline
 * kernels derive from (seed, document index, feature index): the counter-based replacement for the torch
 * CPU generator stream behind nn.Dropout (doubleLayer.py:60), which cannot be reproduced on the device. */
int ltr_dropout_keep_mask(uint64_t seed, int layer, int64_t n_docs, int H, uint8_t *out, void *stream);
/* Dropout probabilities other than the reference's 0.5 (a caller that sets `net.dropout.p`): every `dropout` argument below is
 * 0 (off), 1 (on, p = 0.5) or `1 | (bits of the float p with its lowest bit cleared)` (on, that p; kept units scaled by
 * 1 / (1 - p)).  p = 0.5 draws one hash bit per hidden unit, any other p 16 bits (unit dropped when they are < round(p * 65536)).
 * Only the forward kernels carry the 16-bit stream: with such a p use ltr_mlp_forward / ltr_mlp_forward_save and
 * ltr_mlp_backward_saved (the saved activations make the backward independent of the stream); ltr_mlp_backward and the fused
 * steps return LTR_ERR_PARAM for it unless explicit keep masks are passed.  ltr_dropout_keep_mask_p exports the stream. */
int ltr_dropout_keep_mask_p(uint64_t seed, int layer, int64_t n_docs, int H, float p, uint8_t *out, void *stream);

/* Scorer backward given dL/dscores[n_docs]: recomputes the forward from X (same seed/masks) and leaves one
 * gradient partial per workgroup in `partials` (grid * partial_floats floats); ltr_mlp_reduce_grads then sums
 * them in a fixed order into the flat parameter gradient. */
int ltr_mlp_backward(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                     const uint8_t *keep1, const uint8_t *keep2, const float *dscores, float *partials, int grid,
                     void *stream);
int ltr_mlp_reduce_grads(int net, const float *partials, int grid, float *flat_grad, void *stream);

/* The same forward / backward pair with ONE forward (what autograd does in the reference, main_batch_execution.py:128-170):
 * ltr_mlp_forward_save also writes the post-activation hidden layers h1 / h2 (dropout applied) of every document to `acts`
 * (ltr_mlp_acts_floats(net, n_docs) floats, 16-byte aligned: one lane-ordered 1 KiB fragment per 16-document tile and 16
 * features), ltr_mlp_backward_saved reads them back instead of recomputing fc1 / fc2 (the backward kernel is bound by the
 * fp32 matrix pipe, the round trip costs HBM bandwidth it does not use).  Used for slate lengths the one-launch fused
 * step does not cover (BASELINE config 3: slate 512).  `dropout` only selects the ReLU-dropout slope (2) of the backward.
 */
int64_t ltr_mlp_acts_floats(int net, int64_t n_docs);      /* < 0: LTR_ERR_* */
int ltr_mlp_forward_save(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                         const uint8_t *keep1, const uint8_t *keep2, float *scores, float *acts, int grid, void *stream);
int ltr_mlp_backward_saved(int net, const float *X, int64_t n_docs, const float *packed, int dropout, const float *acts,
                           const float *dscores, float *partials, int grid, void *stream);

/* One fused training pass over B slates of S documents (S in {32, 64, 128}): scorer forward, per-slate loss
 * (LTR_LOSS_*; labels [B][S]), loss backward, scorer backward -- scores never leave the CU, X is read once.
 *   slate_loss[b]  : per-slate loss (caller reduces: mean for approxNDCG, sum for ListNet)
 *   partials       : per-workgroup partials of d (sum_b grad_scale * slate_loss[b]) / d params, to be summed by
 *                    ltr_mlp_reduce_grads (grad_scale = 1/B_global for a mean)
 * main_batch_execution.py:128-170 is the reference call chain this replaces. */
int ltr_fused_step(int net, int loss_kind, const float *X, const float *labels, int B, int S, const float *packed,
                   int dropout, uint64_t seed, const uint8_t *keep1, const uint8_t *keep2, float alpha, float eps,
                   float pad, int apply_sigmoid, float grad_scale, float *slate_loss, float *partials, int grid,
                   void *stream);

/* Diagnostics: in a build compiled with -DLTR_STAMPS the pipeline kernels write s_memtime stamps at their phase
 * boundaries for workgroup-local tile `tile` to buf[grid][8][16] (uint64); returns 1 in such a build, 0 otherwise
 * (the stamps are then compiled out).  buf = NULL disables. */
int ltr_debug_set_stamps(void *buf, int tile);

/* The same fused pass with lambdaLoss (losses/lambdaL.py:67-93; the loss main_batch_execution.py:135 trains
 * with: weighing_scheme="ndcgLoss2PP_scheme").  scheme/k/sigma/mu/eps/log_base as in ltr_lambda_fwd_bwd;
 * slate_loss[b] = -sum of the kept pair terms, slate_count[b] (may be NULL) = kept pairs.  reduction="sum":
 * grad_scale = 1; a "mean" is obtained by dividing loss and flat gradient by the (global) pair count. */
int ltr_fused_step_lambda(int net, const float *X, const float *labels, int B, int S, const float *packed, int dropout,
                          uint64_t seed, const uint8_t *keep1, const uint8_t *keep2, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, float grad_scale, float *slate_loss,
                          float *slate_count, float *partials, int grid, void *stream);

/* =====================================================================================================
 * Data path in front of the pipeline.
 *
 * ---- get_data(info_dataset, type_file)                                        utils/dataset.py:33-69
 * LETOR / svmlight text ("<label> qid:<q> <fid>:<val> ... # comment") -> packed rows, HOST pointers, N host threads
 * over an mmap (n_threads <= 0: all cores).  Two calls: scan (documents, feature-id range: sklearn's zero_based="auto"
 * rule is `min id == 0 ? zero-based : one-based`), then load into caller-allocated arrays in FILE order:
 *   X[n_docs][n_features] fp32 (absent features 0; values parsed as float64 then cast, like dataset.py:63),
 *   y[n_docs] fp64 (svmlight labels are float64), qid[n_docs] (-1 if the line has none).
 * Grouping documents into queries (a new query starts where qid changes, dataset.py:54-60) is the caller's. */
int ltr_svmlight_scan(const char *path, int64_t *n_docs, int32_t *min_feature_id, int32_t *max_feature_id, int n_threads);
int ltr_svmlight_load(const char *path, int64_t n_docs, int n_features, int feature_id_base, float *X, double *y,
                      int64_t *qid, int n_threads);

/* ---- X_train[idx] / y_train[idx] / y_baseline_train[idx]                  main_batch_execution.py:112-117
 * dst[r][:] = src[idx[r]][:] for r < n_rows (device pointers; rows of row_floats fp32; idx int64, entries outside
 * [0, src_rows) leave the destination row untouched).  16-byte-per-lane copies when row_floats % 4 == 0 and both
 * bases are 16-byte aligned.  HBM-bound: every byte is read once and written once. */
int ltr_gather_rows_f32(const float *src, int64_t src_rows, const int64_t *idx, int64_t n_rows, int64_t row_floats,
                        float *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LTR_MI355X_H */
