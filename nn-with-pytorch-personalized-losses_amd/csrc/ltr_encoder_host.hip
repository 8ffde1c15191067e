// ltr_encoder_host.hip -- native (C++) orchestration of the set-transformer scorer: one call runs the whole forward
// (ltr_enc_forward) or the whole backward (ltr_enc_backward) of a `make_model` network -- the same launch sequence as
// ltr_mi355x/encoder.py (_run_forward / _body_backward), issued from C++ so that a training step costs two FFI calls
// instead of ~450: the entry point a non-Python host binds.  (Measured: no faster than the Python-driven sequence -- at the
// reference's own batch size, 64 queries x 100 documents, the step is bound by the GPU-side dispatch of ~450 small
// kernels, 3.1 ms either way -- so the Python host keeps its own sequence by default; results are bit-identical.)
// Host code only: it composes the entry points of include/ltr_encoder.h; every activation and temporary lives in ONE
// caller-provided workspace whose layout is a pure function of (spec, B, S).
#include "../../include/ltr_encoder.h"
#include "../../include/ltr_mi355x.h"
#include <hip/hip_runtime.h>

#include <vector>

namespace {

constexpr float kLnEps = 1e-6f, kStdLnEps = 1e-5f;
constexpr int kNblk = 512;

inline int stream_attn(int l) { return 8 * l; }
inline int stream_attn_out(int l) { return 8 * l + 1; }
inline int stream_ffn_hidden(int l) { return 8 * l + 2; }
inline int stream_ffn_out(int l) { return 8 * l + 3; }
inline int stream_fc(int i) { return 100000 + i; }

struct Bump {
    char *base;
    size_t off = 0;
    template <class T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct LayerBuf {
    const float *x0;                 // residual stream into the block (the previous block's x2 / the FC output)
    uint16_t *n1, *qkv, *ctx, *n2, *hid;
    float *x1, *x2, *bqkv;
    float *lse;                      // [B*h][S] attention row statistic, forward -> backward
    uint16_t *wqkv, *wo, *w1, *w2;
    // backward temporaries
    uint16_t *dy2, *dz1, *dyo, *dctx, *dqkv;
    float *dn;                       // dn2 / dn1 (one at a time)
    float *p_dy2, *p_dyo, *p_ln2, *p_ln1, *p_b1, *p_bqkv;           // column-sum partials
    float *p_w1, *p_w2, *p_ffb1, *p_wo, *p_wqkv;                      // weight-gradient partials
    int s_w1, s_w2, s_wo, s_wqkv, ffn_split;
};

struct Layout {
    int B, S, F, T, d, dff, h, dk, n_fc, n_layers, n_fc_prm, fused, nblk_rows, nblk_ln, nblk_tail;
    int sizes[LTR_ENC_MAX_FC + 1];
    uint16_t *fc_in[LTR_ENC_MAX_FC], *fc_w16[LTR_ENC_MAX_FC];
    float *fc_out;                   // fp32 output of the last FC layer (the encoder's input stream)
    std::vector<LayerBuf> L;
    // backward
    float *dx, *p_tail, *fc_dx[LTR_ENC_MAX_FC], *p_fc_b[LTR_ENC_MAX_FC], *p_fc_w[LTR_ENC_MAX_FC], *p_in_ln, *in_ln_scratch;
    uint16_t *fc_dy[LTR_ENC_MAX_FC];
    int s_fc_w[LTR_ENC_MAX_FC];
    size_t bytes;
};

inline int dw_splits(long long M, long long N, long long K) {
    const long long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    long long s = 512 / tiles;
    s = s < 1 ? 1 : (s > 256 ? 256 : s);
    const long long byk = (K + 255) / 256;
    s = s < byk ? s : byk;
    return (int)(s < 1 ? 1 : s);
}

int check_spec(const ltr_enc_spec *sp, int B, int S) {
    if (!sp) return LTR_ERR_NULL;
    if (B < 1 || S < 1 || sp->n_features < 8 || sp->n_features % 8 || sp->n_fc < 0 || sp->n_fc > LTR_ENC_MAX_FC) return LTR_ERR_SHAPE;
    if ((long long)B * S > (1ll << 30)) return LTR_ERR_SHAPE;
    for (int i = 0; i < sp->n_fc; ++i)
        if (sp->fc_sizes[i] < 8 || sp->fc_sizes[i] % 8) return LTR_ERR_SHAPE;
    if (sp->input_norm && !sp->n_fc) return LTR_ERR_PARAM;
    const int d = sp->n_fc ? sp->fc_sizes[sp->n_fc - 1] : sp->n_features;
    if (d > 512) return LTR_ERR_SHAPE;
    if (sp->has_encoder) {
        if (sp->n_layers < 1 || sp->n_layers > 64 || sp->heads < 1 || d % sp->heads || d / sp->heads > 32 || sp->d_ff < 8 || sp->d_ff % 8 || S > 512)
            return LTR_ERR_SHAPE;
    }
    if (!(sp->fc_dropout >= 0.f) || sp->fc_dropout >= 1.f || !(sp->enc_dropout >= 0.f) || sp->enc_dropout >= 1.f) return LTR_ERR_PARAM;
    return 0;
}

// The one place that decides where everything lives.  base == nullptr: sizes only.
void make_layout(const ltr_enc_spec &sp, int B, int S, void *base, Layout &Y) {
    Bump ws{static_cast<char *>(base)};
    Y.B = B; Y.S = S; Y.F = sp.n_features; Y.T = B * S; Y.n_fc = sp.n_fc;
    Y.n_layers = sp.has_encoder ? sp.n_layers : 0;
    Y.sizes[0] = Y.F;
    for (int i = 0; i < sp.n_fc; ++i) Y.sizes[i + 1] = sp.fc_sizes[i];
    Y.d = Y.sizes[sp.n_fc];
    Y.dff = sp.d_ff; Y.h = sp.heads; Y.dk = sp.has_encoder ? Y.d / sp.heads : 0;
    Y.n_fc_prm = (sp.input_norm ? 2 : 0) + 2 * sp.n_fc;
    Y.fused = sp.has_encoder && ltr_enc_ffn_supported(Y.d, Y.dff);
    const long long T = Y.T;
    Y.nblk_rows = (int)(T < kNblk ? T : kNblk);
    Y.nblk_ln = (int)((T + 7) / 8 < 4 * kNblk ? (T + 7) / 8 : 4 * kNblk);
    Y.nblk_tail = (int)((T + 3) / 4 < kNblk ? (T + 3) / 4 : kNblk);
    // ---- forward
    Y.fc_out = nullptr;
    for (int i = 0; i < sp.n_fc; ++i) {
        Y.fc_in[i] = i == 0 ? ws.take<uint16_t>(T * Y.F) : Y.fc_in[i];     // layer 0 input: cast / normed x
        Y.fc_w16[i] = ws.take<uint16_t>((size_t)Y.sizes[i + 1] * Y.sizes[i]);
        if (i == sp.n_fc - 1) Y.fc_out = ws.take<float>(T * Y.sizes[i + 1]);
        else Y.fc_in[i + 1] = ws.take<uint16_t>(T * Y.sizes[i + 1]);
    }
    const size_t d = Y.d, dff = Y.dff;
    Y.L.assign(Y.n_layers, LayerBuf{});
    for (auto &l : Y.L) {
        l.wqkv = ws.take<uint16_t>(3 * d * d);
        l.wo = ws.take<uint16_t>(d * d);
        l.w1 = ws.take<uint16_t>(dff * d);
        l.w2 = ws.take<uint16_t>(d * dff);
        l.bqkv = ws.take<float>(3 * d);
        l.n1 = ws.take<uint16_t>(T * d);
        l.qkv = ws.take<uint16_t>(T * 3 * d);
        l.ctx = ws.take<uint16_t>(T * d);
        l.lse = ws.take<float>(T * (size_t)Y.h);
        l.x1 = ws.take<float>(T * d);
        l.n2 = ws.take<uint16_t>(T * d);
        l.hid = Y.fused ? nullptr : ws.take<uint16_t>(T * dff);
        l.x2 = ws.take<float>(T * d);
    }
    // ---- backward (temporaries are shared between blocks: one block is processed at a time; the PARTIAL buffers are
    //      per block because their reductions are deferred to the end)
    Y.dx = ws.take<float>(T * d);
    Y.p_tail = ws.take<float>((size_t)Y.nblk_tail * (3 * d + 8));
    uint16_t *dy2 = Y.n_layers ? ws.take<uint16_t>(T * d) : nullptr, *dz1 = (Y.n_layers && !Y.fused) ? ws.take<uint16_t>(T * dff) : nullptr;
    uint16_t *dctx = Y.n_layers ? ws.take<uint16_t>(T * d) : nullptr, *dqkv = Y.n_layers ? ws.take<uint16_t>(T * 3 * d) : nullptr;
    float *dn = Y.n_layers ? ws.take<float>(T * d) : nullptr;
    for (auto &l : Y.L) {
        l.dy2 = l.dyo = dy2; l.dz1 = dz1; l.dctx = dctx; l.dqkv = dqkv; l.dn = dn;
        l.p_dy2 = ws.take<float>((size_t)Y.nblk_rows * d);
        l.p_dyo = ws.take<float>((size_t)Y.nblk_rows * d);
        l.p_ln2 = ws.take<float>((size_t)Y.nblk_ln * 2 * d);
        l.p_ln1 = ws.take<float>((size_t)Y.nblk_ln * 2 * d);
        l.p_bqkv = ws.take<float>((size_t)Y.nblk_rows * 3 * d);
        l.s_wo = dw_splits(d, d, T);
        l.s_wqkv = dw_splits(3 * d, d, T);
        l.p_wo = ws.take<float>((size_t)l.s_wo * d * d);
        l.p_wqkv = ws.take<float>((size_t)l.s_wqkv * 3 * d * d);
        if (Y.fused) {
            long long ns = 256 / (long long)(dff / 128), tiles = (T + 127) / 128;
            ns = ns < 1 ? 1 : ns;
            l.ffn_split = (int)(ns < tiles ? ns : tiles);
            l.p_w1 = ws.take<float>((size_t)l.ffn_split * dff * d);
            l.p_w2 = ws.take<float>((size_t)l.ffn_split * d * dff);
            l.p_ffb1 = ws.take<float>((size_t)l.ffn_split * dff);
            l.p_b1 = nullptr;
            l.s_w1 = l.s_w2 = 0;
        } else {
            l.s_w1 = dw_splits(dff, d, T);
            l.s_w2 = dw_splits(d, dff, T);
            l.p_w1 = ws.take<float>((size_t)l.s_w1 * dff * d);
            l.p_w2 = ws.take<float>((size_t)l.s_w2 * d * dff);
            l.p_b1 = ws.take<float>((size_t)Y.nblk_rows * dff);
            l.p_ffb1 = nullptr;
            l.ffn_split = 0;
        }
    }
    for (int i = 0; i < sp.n_fc; ++i) {
        Y.fc_dy[i] = ws.take<uint16_t>(T * Y.sizes[i + 1]);
        Y.p_fc_b[i] = ws.take<float>((size_t)Y.nblk_rows * Y.sizes[i + 1]);
        Y.s_fc_w[i] = dw_splits(Y.sizes[i + 1], Y.sizes[i], T);
        Y.p_fc_w[i] = ws.take<float>((size_t)Y.s_fc_w[i] * Y.sizes[i + 1] * Y.sizes[i]);
        Y.fc_dx[i] = (i > 0 || sp.input_norm) ? ws.take<float>(T * Y.sizes[i]) : nullptr;
    }
    Y.p_in_ln = sp.input_norm ? ws.take<float>((size_t)Y.nblk_ln * 2 * Y.F) : nullptr;
    Y.in_ln_scratch = sp.input_norm ? ws.take<float>(T * Y.F) : nullptr;
    Y.bytes = (ws.off + 255) & ~(size_t)255;
}

int gemm(const uint16_t *A, const uint16_t *Bm, long long M, long long N, long long K, bool akm, bool bkm, float *Cf, uint16_t *Cb,
         const float *bias, const float *residual, const uint16_t *gate, float gate_scale, int relu, float drop_p, uint64_t seed,
         int drop_stream, int splits, void *stream) {
    ltr_gemm_desc g{};
    g.A = A; g.B = Bm; g.M = M; g.N = N; g.K = K;
    g.lda = akm ? M : K; g.ldb = bkm ? N : K; g.ldc = N;
    g.a_kmajor = akm; g.b_kmajor = bkm; g.splits = splits; g.relu = relu;
    g.Cf = Cf; g.Cb = Cb; g.bias = bias; g.residual = residual; g.gate = gate; g.gate_scale = gate_scale;
    g.drop_p = drop_p; g.seed = seed; g.drop_stream = drop_stream;
    return ltr_enc_gemm_bf16(&g, stream);
}

#define TRY(expr)              \
    do {                       \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

}  // namespace

extern "C" {

int64_t ltr_enc_workspace_bytes(const ltr_enc_spec *spec, int B, int S) {
    if (int rc = check_spec(spec, B, S)) return rc;
    Layout Y;
    make_layout(*spec, B, S, nullptr, Y);
    return (int64_t)Y.bytes;
}

int ltr_enc_forward(const ltr_enc_spec *spec, const float *x, const uint8_t *mask, int B, int S, const float *const *params,
                    int n_params, uint64_t seed, int training, void *workspace, float *scores, void *stream) {
    if (int rc = check_spec(spec, B, S)) return rc;
    if (!x || !params || !workspace || !scores || (spec->has_encoder && !mask)) return LTR_ERR_NULL;
    if ((uintptr_t)workspace & 255u) return LTR_ERR_ALIGN;
    const ltr_enc_spec &sp = *spec;
    Layout Y;
    make_layout(sp, B, S, workspace, Y);
    if (n_params != Y.n_fc_prm + 16 * Y.n_layers + (sp.has_encoder ? 2 : 0) + 2) return LTR_ERR_SHAPE;
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) return LTR_ERR_NULL;
    const float p_fc = training ? sp.fc_dropout : 0.f, p_enc = training ? sp.enc_dropout : 0.f;
    const long long T = Y.T;
    hipStream_t hs = (hipStream_t)stream;
    int it = 0;
    // ---- FCModel (multiLayer.py:42-51)
    if (sp.input_norm) {
        TRY(ltr_enc_layernorm_fwd(x, params[0], params[1], T, Y.F, kStdLnEps, 1, Y.fc_in[0], nullptr, stream));
        it = 2;
    } else if (sp.n_fc) TRY(ltr_enc_cast_bf16(x, Y.fc_in[0], T * Y.F, stream));
    const float *stream_x = x;
    for (int i = 0; i < sp.n_fc; ++i) {
        const float *W = params[it++], *bvec = params[it++];
        const int n_in = Y.sizes[i], n_out = Y.sizes[i + 1];
        const bool last = i == sp.n_fc - 1;
        TRY(ltr_enc_cast_bf16(W, Y.fc_w16[i], (int64_t)n_out * n_in, stream));
        TRY(gemm(Y.fc_in[i], Y.fc_w16[i], T, n_out, n_in, false, false, last ? Y.fc_out : nullptr, last ? nullptr : Y.fc_in[i + 1], bvec,
                 nullptr, nullptr, 1.f, 0, p_fc, seed, stream_fc(i), 1, stream));
        if (last) stream_x = Y.fc_out;
    }
    // ---- Encoder blocks (transformer.py:44-59, 132-142)
    const long long d = Y.d, dff = Y.dff;
    for (int l = 0; l < Y.n_layers; ++l) {
        LayerBuf &b = Y.L[l];
        const float *const *P = params + Y.n_fc_prm + 16 * l;      // a1 b1n Wq bq Wk bk Wv bv Wo bo a2 b2n W1 b1 W2 b2
        for (int j = 0; j < 3; ++j) {
            TRY(ltr_enc_cast_bf16(P[2 + 2 * j], b.wqkv + j * d * d, d * d, stream));
            hipError_t e = hipMemcpyAsync(b.bqkv + j * d, P[3 + 2 * j], d * sizeof(float), hipMemcpyDeviceToDevice, hs);
            if (e != hipSuccess) return (int)e;
        }
        TRY(ltr_enc_cast_bf16(P[8], b.wo, d * d, stream));
        TRY(ltr_enc_cast_bf16(P[12], b.w1, dff * d, stream));
        TRY(ltr_enc_cast_bf16(P[14], b.w2, d * dff, stream));
        b.x0 = stream_x;
        TRY(ltr_enc_layernorm_fwd(b.x0, P[0], P[1], T, (int)d, kLnEps, 0, b.n1, nullptr, stream));
        TRY(gemm(b.n1, b.wqkv, T, 3 * d, d, false, false, nullptr, b.qkv, b.bqkv, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, 1, stream));
        TRY(ltr_enc_attention_fwd_lse(b.qkv, mask, B, S, Y.h, Y.dk, p_enc, seed, stream_attn(l), b.ctx, b.lse, stream));
        TRY(gemm(b.ctx, b.wo, T, d, d, false, false, b.x1, nullptr, P[9], b.x0, nullptr, 1.f, 0, p_enc, seed, stream_attn_out(l), 1, stream));
        TRY(ltr_enc_layernorm_fwd(b.x1, P[10], P[11], T, (int)d, kLnEps, 0, b.n2, nullptr, stream));
        if (Y.fused)
            TRY(ltr_enc_ffn_fwd(b.n2, b.w1, P[13], b.w2, P[15], b.x1, T, (int)d, (int)dff, p_enc, seed, stream_ffn_hidden(l),
                                stream_ffn_out(l), b.x2, stream));
        else {
            TRY(gemm(b.n2, b.w1, T, dff, d, false, false, nullptr, b.hid, P[13], nullptr, nullptr, 1.f, 1, p_enc, seed, stream_ffn_hidden(l), 1, stream));
            TRY(gemm(b.hid, b.w2, T, d, dff, false, false, b.x2, nullptr, P[15], b.x1, nullptr, 1.f, 0, p_enc, seed, stream_ffn_out(l), 1, stream));
        }
        stream_x = b.x2;
    }
    const float *fa = sp.has_encoder ? params[n_params - 4] : nullptr, *fb = sp.has_encoder ? params[n_params - 3] : nullptr;
    return ltr_enc_score_fwd(stream_x, fa, fb, params[n_params - 2], params[n_params - 1], T, (int)d, kLnEps, sp.has_encoder ? 1 : 0, scores, stream);
}

int ltr_enc_backward(const ltr_enc_spec *spec, const float *x, const uint8_t *mask, int B, int S, const float *const *params,
                     int n_params, uint64_t seed, int training, const float *dscores, void *workspace, float *const *grads,
                     void *stream) {
    if (int rc = check_spec(spec, B, S)) return rc;
    if (!x || !params || !workspace || !dscores || !grads || (spec->has_encoder && !mask)) return LTR_ERR_NULL;
    if ((uintptr_t)workspace & 255u) return LTR_ERR_ALIGN;
    const ltr_enc_spec &sp = *spec;
    Layout Y;
    make_layout(sp, B, S, workspace, Y);
    if (n_params != Y.n_fc_prm + 16 * Y.n_layers + (sp.has_encoder ? 2 : 0) + 2) return LTR_ERR_SHAPE;
    for (int i = 0; i < n_params; ++i)
        if (!params[i] || !grads[i]) return LTR_ERR_NULL;
    const float p_fc = training ? sp.fc_dropout : 0.f, p_enc = training ? sp.enc_dropout : 0.f;
    const long long T = Y.T, d = Y.d, dff = Y.dff;
    std::vector<ltr_reduce_job> jobs;
    auto reduce = [&](const float *parts, int nsplit, long long n, float *out, long long stride = 0) {
        jobs.push_back(ltr_reduce_job{parts, out, n, stride, nsplit, 0});
    };
    // ---- scoring tail
    const float *final_x = Y.n_layers ? Y.L.back().x2 : (sp.n_fc ? Y.fc_out : x);
    const float *fa = sp.has_encoder ? params[n_params - 4] : nullptr, *fb = sp.has_encoder ? params[n_params - 3] : nullptr;
    TRY(ltr_enc_score_bwd(final_x, fa, fb, params[n_params - 2], dscores, T, (int)d, kLnEps, sp.has_encoder ? 1 : 0, Y.dx, Y.p_tail,
                          Y.nblk_tail, stream));
    const long long tw = 3 * d + 8;
    if (sp.has_encoder) {
        reduce(Y.p_tail, Y.nblk_tail, d, grads[n_params - 4], tw);
        reduce(Y.p_tail + d, Y.nblk_tail, d, grads[n_params - 3], tw);
    }
    reduce(Y.p_tail + 2 * d, Y.nblk_tail, d, grads[n_params - 2], tw);
    reduce(Y.p_tail + 3 * d, Y.nblk_tail, 1, grads[n_params - 1], tw);
    // ---- encoder blocks, last to first
    for (int l = Y.n_layers - 1; l >= 0; --l) {
        LayerBuf &b = Y.L[l];
        b.x0 = l ? Y.L[l - 1].x2 : (sp.n_fc ? Y.fc_out : x);
        const int base = Y.n_fc_prm + 16 * l;
        const float *const *P = params + base;
        float *const *G = grads + base;
        // FFN sublayer: x2 = x1 + drop(hid W2^T + b2)
        TRY(ltr_enc_drop_cast_colsum(Y.dx, T, (int)d, p_enc, seed, stream_ffn_out(l), b.dy2, b.p_dy2, Y.nblk_rows, stream));
        reduce(b.p_dy2, Y.nblk_rows, d, G[15]);
        if (Y.fused) {
            TRY(ltr_enc_ffn_bwd_x(b.n2, b.w1, P[13], b.w2, b.dy2, T, (int)d, (int)dff, p_enc, seed, stream_ffn_hidden(l), b.dn, stream));
            TRY(ltr_enc_ffn_bwd_w(b.n2, b.w1, P[13], b.w2, b.dy2, T, (int)d, (int)dff, p_enc, seed, stream_ffn_hidden(l), b.ffn_split, b.p_w1,
                                  b.p_w2, b.p_ffb1, stream));
            reduce(b.p_w1, b.ffn_split, dff * d, G[12]);
            reduce(b.p_w2, b.ffn_split, d * dff, G[14]);
            reduce(b.p_ffb1, b.ffn_split, dff, G[13]);
        } else {
            TRY(gemm(b.dy2, b.hid, d, dff, T, true, true, b.p_w2, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, b.s_w2, stream));
            reduce(b.p_w2, b.s_w2, d * dff, G[14]);
            TRY(gemm(b.dy2, b.w2, T, dff, d, false, true, nullptr, b.dz1, nullptr, nullptr, b.hid, 1.f / (1.f - p_enc), 0, 0.f, 0, 0, 1, stream));
            TRY(ltr_enc_colsum_bf16(b.dz1, T, (int)dff, b.p_b1, Y.nblk_rows, stream));
            reduce(b.p_b1, Y.nblk_rows, dff, G[13]);
            TRY(gemm(b.dz1, b.n2, dff, d, T, true, true, b.p_w1, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, b.s_w1, stream));
            reduce(b.p_w1, b.s_w1, dff * d, G[12]);
            TRY(gemm(b.dz1, b.w1, T, d, dff, false, true, b.dn, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, 1, stream));
        }
        TRY(ltr_enc_layernorm_bwd(b.x1, P[10], b.dn, T, (int)d, kLnEps, 0, Y.dx, b.p_ln2, Y.nblk_ln, stream));
        reduce(b.p_ln2, Y.nblk_ln, d, G[10], 2 * d);
        reduce(b.p_ln2 + d, Y.nblk_ln, d, G[11], 2 * d);
        // attention sublayer: x1 = x0 + drop(ctx Wo^T + bo)
        TRY(ltr_enc_drop_cast_colsum(Y.dx, T, (int)d, p_enc, seed, stream_attn_out(l), b.dyo, b.p_dyo, Y.nblk_rows, stream));
        reduce(b.p_dyo, Y.nblk_rows, d, G[9]);
        TRY(gemm(b.dyo, b.ctx, d, d, T, true, true, b.p_wo, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, b.s_wo, stream));
        reduce(b.p_wo, b.s_wo, d * d, G[8]);
        TRY(gemm(b.dyo, b.wo, T, d, d, false, true, nullptr, b.dctx, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, 1, stream));
        TRY(ltr_enc_attention_bwd_lse(b.qkv, b.ctx, b.dctx, b.lse, mask, B, S, Y.h, Y.dk, p_enc, seed, stream_attn(l), b.dqkv, stream));
        TRY(ltr_enc_colsum_bf16(b.dqkv, T, (int)(3 * d), b.p_bqkv, Y.nblk_rows, stream));
        TRY(gemm(b.dqkv, b.n1, 3 * d, d, T, true, true, b.p_wqkv, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, b.s_wqkv, stream));
        for (int j = 0; j < 3; ++j) {
            reduce(b.p_bqkv + j * d, Y.nblk_rows, d, G[3 + 2 * j], 3 * d);
            reduce(b.p_wqkv + j * d * d, b.s_wqkv, d * d, G[2 + 2 * j], 3 * d * d);
        }
        TRY(gemm(b.dqkv, b.wqkv, T, d, 3 * d, false, true, b.dn, nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, 1, stream));
        TRY(ltr_enc_layernorm_bwd(b.x0, P[0], b.dn, T, (int)d, kLnEps, 0, Y.dx, b.p_ln1, Y.nblk_ln, stream));
        reduce(b.p_ln1, Y.nblk_ln, d, G[0], 2 * d);
        reduce(b.p_ln1 + d, Y.nblk_ln, d, G[1], 2 * d);
    }
    // ---- FCModel backward
    const int n_fc0 = sp.input_norm ? 2 : 0;
    const float *dxi = Y.dx;
    for (int i = sp.n_fc - 1; i >= 0; --i) {
        const int n_in = Y.sizes[i], n_out = Y.sizes[i + 1];
        TRY(ltr_enc_drop_cast_colsum(dxi, T, n_out, p_fc, seed, stream_fc(i), Y.fc_dy[i], Y.p_fc_b[i], Y.nblk_rows, stream));
        reduce(Y.p_fc_b[i], Y.nblk_rows, n_out, grads[n_fc0 + 2 * i + 1]);
        TRY(gemm(Y.fc_dy[i], Y.fc_in[i], n_out, n_in, T, true, true, Y.p_fc_w[i], nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0,
                 Y.s_fc_w[i], stream));
        reduce(Y.p_fc_w[i], Y.s_fc_w[i], (long long)n_out * n_in, grads[n_fc0 + 2 * i]);
        if (i > 0 || sp.input_norm) {
            TRY(gemm(Y.fc_dy[i], Y.fc_w16[i], T, n_in, n_out, false, true, Y.fc_dx[i], nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0.f, 0, 0, 1,
                     stream));
            dxi = Y.fc_dx[i];
        }
    }
    if (sp.input_norm) {
        hipError_t e = hipMemsetAsync(Y.in_ln_scratch, 0, (size_t)T * Y.F * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        TRY(ltr_enc_layernorm_bwd(x, params[0], dxi, T, Y.F, kStdLnEps, 1, Y.in_ln_scratch, Y.p_in_ln, Y.nblk_ln, stream));
        reduce(Y.p_in_ln, Y.nblk_ln, Y.F, grads[0], 2 * Y.F);
        reduce(Y.p_in_ln + Y.F, Y.nblk_ln, Y.F, grads[1], 2 * Y.F);
    }
    return ltr_enc_sum_partials_batch(jobs.data(), (int)jobs.size(), stream);
}

}  // extern "C"
