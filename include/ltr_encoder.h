/*
 * ltr_encoder.h -- C ABI of the set-transformer scorer (BASELINE config 5): the kernels behind
 * architeture/multiLayer.py (`make_model`, FCModel, LTRModel, OutputLayer, :13-149) and architeture/transformer.py
 * (Encoder, LayerNorm, SublayerConnection, attention, MultiHeadedAttention, PositionwiseFeedForward, :29-257).
 * Exported by libltr_mi355x.so next to ltr_mi355x.h; same conventions (device pointers, hipStream_t as void*,
 * 0 = launched, < 0 = LTR_ERR_*, > 0 = hipError_t, launchers never allocate or synchronise).
 *
 * Arithmetic: bf16 operands on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the residual stream, layer-norm
 * statistics, softmax and every parameter gradient are fp32.  bf16 buffers are raw uint16_t bit patterns.
 * Tokens: T = B * S rows (slate b, document i -> row b*S + i); feature counts must be multiples of 8.
 *
 * Dropout everywhere in this file is a counter-based stream: element `idx` of stream `stream_id` is dropped when
 * the 16-bit hash word (seed, stream_id, idx) < round(p * 65536); kept values are scaled by 1/(1-p).  The backward
 * recomputes it; ltr_enc_dropout_mask / ltr_enc_attn_dropout_mask export it for tests.
 */
#ifndef LTR_ENCODER_H
#define LTR_ENCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fp32 -> bf16 (round to nearest even), n elements.  Used once per step on the master weights. */
int ltr_enc_cast_bf16(const float *src, uint16_t *dst, int64_t n, void *stream);

/* Device-side dropout epoch.  Every kernel of this file draws its masks from (seed + EPOCH, stream_id, index), where EPOCH is a
 * 64-bit word in device memory (0 after load).  The reference draws a new mask per forward from torch's generator
 * (architeture/transformer.py:30,52,161 -- nn.Dropout); a host passes a new `seed` per step for that.  A launch sequence recorded
 * into a hipGraph has its `seed` arguments frozen, so it begins with ltr_enc_seed_advance (a kernel node): each replay then uses
 * the next epoch, forward and backward of the same replay the same one.  All three are stream-ordered like any other launch;
 * ltr_enc_seed_get synchronises the device (tests). */
int ltr_enc_seed_set(uint64_t value, void *stream);
int ltr_enc_seed_advance(uint64_t delta, void *stream);
int ltr_enc_seed_get(uint64_t *value);

/* out[i] = 1 if element i of `stream_id` is kept (see the header comment), i in [0, n). */
int ltr_enc_dropout_mask(uint64_t seed, int stream_id, int64_t n, float p, uint8_t *out, void *stream);
/* The keep mask of the attention-probability dropout (transformer.py:161-163), out [B][h][S][S]. */
int ltr_enc_attn_dropout_mask(uint64_t seed, int stream_id, int B, int S, int h, float p, uint8_t *out, void *stream);

/* out[i] = sum_z parts[z*n + i], z in [0, nsplit) in that fixed order (fp64 accumulator); accumulate != 0: out += . */
int ltr_enc_sum_partials(const float *parts, int nsplit, int64_t n, int accumulate, float *out, void *stream);
/* Epilogue of a split-K activation GEMM (ltr_enc_gemm_bf16 with splits > 1 writes raw partials [splits][M][N]):
 * out [M][N] = residual + dropout(sum_s parts[s] + bias), the arithmetic and dropout stream of the GEMM's own epilogue
 * (transformer.py:231-237 followed by the SublayerConnection residual, :107-114).  bias / residual may be NULL; N % 4 == 0. */
int ltr_enc_splitk_epilogue(const float *parts, int nsplit, int64_t M, int N, const float *bias, float drop_p, uint64_t seed, int stream_id,
                            const float *residual, float *out, void *stream);

/* The same for `njobs` independent reductions in as few launches as possible (16 jobs per launch); `jobs` is a HOST array. */
typedef struct ltr_reduce_job {
    const float *parts;
    float *out;
    int64_t n;          /* outputs */
    int64_t stride;     /* floats between consecutive partial rows; 0 = n (a job may reduce a column slice of wider rows) */
    int32_t nsplit, reserved;
} ltr_reduce_job;
int ltr_enc_sum_partials_batch(const ltr_reduce_job *jobs, int njobs, void *stream);

/* ---- LayerNorm, transformer.py:64-88:  y = a_2 * (x - mean) / (std + eps) + b_2, std UNBIASED (torch.std).
 * standard != 0: nn.LayerNorm (multiLayer.py:27: biased variance, eps inside the root).
 *   x [T][d] fp32; y_bf16 [T][d] (the next GEMM's operand) and/or y_f32 [T][d] (either may be NULL, not both). */
int ltr_enc_layernorm_fwd(const float *x, const float *a, const float *b, int64_t T, int d, float eps, int standard,
                          uint16_t *y_bf16, float *y_f32, void *stream);
/* dx[T][d] += d loss / d x through the norm, given dy [T][d] fp32 (gradient w.r.t. y); partials [nblk][2*d]:
 * per-workgroup sums of d a_2 (first d) and d b_2 (last d) -- reduce with ltr_enc_sum_partials.  nblk = grid size. */
int ltr_enc_layernorm_bwd(const float *x, const float *a, const float *dy, int64_t T, int d, float eps, int standard,
                          float *dx, float *partials, int nblk, void *stream);

/* ---- The one GEMM all Linear layers use (multiLayer.py:28,104; transformer.py:184,227-228), forward and backward:
 *        C[m][n] = sum_k A(m,k) * B(n,k)        m < M, n < N, k < K
 *   A(m,k) = a_kmajor ? A[k*lda + m] : A[m*lda + k]   (same for B): "k-major" operands are read through transposed
 *   LDS reads (ds_read_b64_tr_b16), so no operand is ever transposed in memory:
 *        forward   y = x W^T      A = x [T][K],  B = W [N][K]            (0, 0)
 *        d x       = dy W         A = dy [T][N], B = W [N][K] k-major    (0, 1)
 *        d W       = dy^T x       A = dy [T][N] k-major, B = x [T][K] k-major   (1, 1), split over T
 *   Epilogue, in this order (all optional): + bias[n]; relu; gate (v = gate[m][n] > 0 ? v * gate_scale : 0, the
 *   ReLU+dropout backward through the saved activation); dropout(drop_p, seed, drop_stream, idx = m*N + n);
 *   + residual[m][n] (fp32, leading dimension ldc); store fp32 Cf and/or bf16 Cb (leading dimension ldc).
 *   splits > 1: split-K; slice z writes the raw partial to Cf + z*M*ldc (no epilogue, Cb ignored) -- reduce with
 *   ltr_enc_sum_partials.  Requirements: N, K-contiguous extents and leading dimensions multiples of 8; pointers
 *   16-byte aligned. */
typedef struct ltr_gemm_desc {
    const uint16_t *A, *B;
    int64_t M, N, K, lda, ldb, ldc;
    int32_t a_kmajor, b_kmajor, splits, relu;
    float *Cf;
    uint16_t *Cb;
    const float *bias;
    const float *residual;
    const uint16_t *gate;
    float gate_scale, drop_p;
    uint64_t seed;
    int32_t drop_stream, reserved;
} ltr_gemm_desc;
int ltr_enc_gemm_bf16(const ltr_gemm_desc *desc /* host pointer */, void *stream);

/* Column sums (bias gradients): partials[blk][n] = sum over the workgroup's rows of y[t][n]; y bf16 [T][N]. */
int ltr_enc_colsum_bf16(const uint16_t *y, int64_t T, int N, float *partials, int nblk, void *stream);
/* Backward of `x + dropout(sublayer)` (transformer.py:113-114) into the sublayer's last Linear: out = bf16(dx * keep /
 * (1-p)) with idx = t*N + n, plus the column sums of the same values (that Linear's bias gradient). */
int ltr_enc_drop_cast_colsum(const float *dx, int64_t T, int N, float p, uint64_t seed, int stream_id, uint16_t *out,
                             float *partials, int nblk, void *stream);

/* ---- attention(query, key, value, mask, dropout), transformer.py:145-164, for all heads of all slates.
 *   qkv [T][3*d] bf16 (d = h*dk; Q | K | V, head hd at columns hd*dk ..), mask [B][S] uint8 (1 = padded document:
 *   masked_fill(mask == 1, -inf), :158-159; NULL = no padding), ctx [T][d] bf16 (heads concatenated, :207-209).
 *   Whole-row softmax in registers (no online rescaling), S <= 512, dk <= 32.  A slate whose documents are all
 *   masked yields zeros (the reference yields NaN). */
int ltr_enc_attention_fwd(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p,
                          uint64_t seed, int stream_id, uint16_t *ctx, void *stream);
/* dqkv [T][3*d] bf16 from dctx [T][d] bf16 and the forward's output ctx [T][d] (row sums dP . P = dctx . ctx per head);
 * probabilities are recomputed. */
int ltr_enc_attention_bwd(const uint16_t *qkv, const uint16_t *ctx, const uint16_t *dctx, const uint8_t *mask, int B, int S, int h,
                          int dk, float drop_p, uint64_t seed, int stream_id, uint16_t *dqkv, void *stream);

/* The same pair with the row statistic kept between them: lse [B*h][S] fp32 = log2 sum_k 2^(log2(e) s_qk / sqrt(dk)) over the
 * unmasked keys (+inf for a query without one), written by the forward (NULL = not wanted) and read by the backward, which
 * then evaluates every probability ONCE, key-major (dk <= 16 and S <= 256; otherwise, or with lse == NULL, the two-phase
 * kernel of ltr_enc_attention_bwd runs).  This is what the training step uses. */
int ltr_enc_attention_fwd_lse(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p,
                              uint64_t seed, int stream_id, uint16_t *ctx, float *lse, void *stream);
int ltr_enc_attention_bwd_lse(const uint16_t *qkv, const uint16_t *ctx, const uint16_t *dctx, const float *lse, const uint8_t *mask,
                              int B, int S, int h, int dk, float drop_p, uint64_t seed, int stream_id, uint16_t *dqkv, void *stream);

/* p_attn, the second value `attention()` returns (transformer.py:161-164; MultiHeadedAttention keeps it in `.attn`, :207):
 * probs [B][h][S][S] fp32 = dropout(softmax(q k^T / sqrt(dk) masked)), from the same bf16 q, k and the same dropout stream
 * as ltr_enc_attention_fwd.  Forward-only helper for callers that inspect the attention map; not on the training path. */
int ltr_enc_attention_probs(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p, uint64_t seed,
                            int stream_id, float *probs, void *stream);

/* ---- PositionwiseFeedForward + its residual tail (transformer.py:215-237, :113-114) without a [T][d_ff] tensor in HBM:
 * the hidden activation is produced 128 units at a time in registers and recomputed in the backward.
 *   forward    x2 = x1 + dropout(dropout(relu(n2 W1^T + b1)) W2^T + b2)      (streams: stream_hidden idx = t*dff + h,
 *                                                                              stream_out idx = t*d + c -- the same
 *                                                                              streams ltr_enc_gemm_bf16 would use)
 *   backward X dn2 [T][d] fp32 = ((dy W2) gated by the recomputed activation) W1,  dy [T][d] bf16 = d loss / d (FFN output
 *              before the output dropout's inverse, i.e. what ltr_enc_drop_cast_colsum produces)
 *   backward W per-range partials dw1 [nsplit][dff][d], dw2 [nsplit][d][dff], db1 [nsplit][dff] (reduce with
 *              ltr_enc_sum_partials); workgroup = (128-unit chunk, token range).
 * n2 [T][d], w1 [dff][d], w2 [d][dff] bf16; b1, b2, x1, x2 fp32.  d in {64, 128}, dff a multiple of 128
 * (ltr_enc_ffn_supported); other shapes use the GEMM entry point. */
int ltr_enc_ffn_supported(int d, int dff);
int ltr_enc_ffn_fwd(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const float *b2, const float *x1,
                    int64_t T, int d, int dff, float drop_p, uint64_t seed, int stream_hidden, int stream_out, float *x2, void *stream);
int ltr_enc_ffn_bwd_x(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const uint16_t *dy, int64_t T, int d,
                      int dff, float drop_p, uint64_t seed, int stream_hidden, float *dn2, void *stream);
int ltr_enc_ffn_bwd_w(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const uint16_t *dy, int64_t T, int d,
                      int dff, float drop_p, uint64_t seed, int stream_hidden, int nsplit, float *dw1_parts, float *dw2_parts,
                      float *db1_parts, void *stream);

/* ---- Encoder.norm + OutputLayer.w_1 with d_output = 1 (transformer.py:59, multiLayer.py:104-113):
 *   scores[t] = w . LN(x[t]) + bias     (norm: 0 = none, 1 = transformer.py LayerNorm, 2 = nn.LayerNorm)
 * `bias` is a device pointer to one float.  The listwise loss then runs on scores [B][S] (ltr_mi355x.h). */
int ltr_enc_score_fwd(const float *x, const float *a, const float *b, const float *w, const float *bias, int64_t T, int d,
                      float eps, int norm, float *scores, void *stream);
/* dx [T][d] (overwritten) and per-workgroup partials [nblk][3*d + 8]: d a_2 | d b_2 | d w | d bias (1 float, padded). */
int ltr_enc_score_bwd(const float *x, const float *a, const float *b, const float *w, const float *dscores, int64_t T, int d,
                      float eps, int norm, float *dx, float *partials, int nblk, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LTR_ENCODER_H */
