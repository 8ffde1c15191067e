// ltr_losses.hip -- standalone listwise-loss kernels (gfx950) + their C-ABI launchers.
//
// One slate group (64..1024 threads) per slate, slate state in LDS, forward and analytic backward in one
// pass.  Small slates share a 256-thread workgroup (4 slates at S <= 32), so a launch has B/4..B
// workgroups >> 256 CUs for any realistic batch.  These kernels are VALU/transcendental-bound
// (S^2 sigmoids per slate against 12*S bytes of HBM traffic), see DESIGN.md.
#include "../../include/ltr_mi355x.h"
#include "ltr_slate_losses.h"

using namespace ltr;

namespace {

inline int check_slates(const void *a, const void *b, const void *c, int B, int S) {
    if (!a || !b || !c) return LTR_ERR_NULL;
    if (B < 0 || S < 1 || S > LTR_MAX_SLATE) return LTR_ERR_SHAPE;
    return LTR_OK;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

struct SlateLaunch {
    int group, block, gpb, grid;
    size_t lds;
};

// arrays_per_slate: number of [S] float arrays each slate group keeps in LDS.
inline SlateLaunch plan(int B, int S, int arrays_per_slate) {
    SlateLaunch L;
    L.group = pick_group(S);
    L.block = L.group < 256 ? 256 : L.group;
    L.gpb = L.block / L.group;
    L.grid = (B + L.gpb - 1) / L.gpb;
    const int s_al = (S + 3) & ~3;
    L.lds = (size_t)L.gpb * (arrays_per_slate * s_al + L.group + 32) * sizeof(float);
    return L;
}

template <class K>
inline int allow_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return LTR_OK;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e == hipSuccess ? LTR_OK : (int)e;
}

// ------------------------------------------------------------------------------------ approxNDCG
constexpr int kApproxArrays = 7;  // sc yl gn gg uu mk um
__global__ void __launch_bounds__(1024)
approxndcg_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S, int group,
                  float alpha, float eps, float pad, float gscale, float *__restrict__ slate_loss,
                  float *__restrict__ dscores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const int gid = threadIdx.x / group;
    const long long slate = ltr_block_id() * (blockDim.x / group) + gid;
    const bool active = slate < B;
    float *base = smem + (size_t)gid * (kApproxArrays * s_al + group + 32);
    float *sc = base, *yl = sc + s_al, *gn = yl + s_al, *gg = gn + s_al, *uu = gg + s_al, *mk = uu + s_al;
    ApproxScratch xs;
    xs.um = mk + s_al;
    const SlateGroup g = make_group(S, group, xs.um + s_al);
    const size_t off = (size_t)slate * S;
    approx_ndcg_init(g);
    for (int j = g.t; j < S; j += group) {
        const float y = active ? labels[off + j] : pad;
        sc[j] = active ? scores[off + j] : 0.f;
        stage_label(y, pad, yl[j], gn[j]);
    }
    __syncthreads();
    float *dst = dscores ? dscores + off : nullptr;
    const float loss = approx_ndcg_slate(g, sc, yl, gn, gg, uu, mk, alpha, eps, gscale, dscores != nullptr,
                                         [&](int i, float v) { if (active) dst[i] = v; }, NoStamp(), xs);
    if (active && g.t == 0) slate_loss[slate] = loss;
}

// --------------------------------------------------------------------------------------- ListNet
__global__ void __launch_bounds__(1024)
listnet_kernel(const float *__restrict__ y_true, const float *__restrict__ y_pred, int B, int S, int group,
               int apply_sigmoid, float gscale, float *__restrict__ slate_loss, float *__restrict__ dscores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const int gid = threadIdx.x / group;
    const long long slate = ltr_block_id() * (blockDim.x / group) + gid;
    const bool active = slate < B;
    float *base = smem + (size_t)gid * (2 * s_al + group + 32);
    float *yt = base, *yp = yt + s_al;
    const SlateGroup g = make_group(S, group, yp + s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        yt[j] = active ? y_true[off + j] : 0.f;
        yp[j] = active ? y_pred[off + j] : 0.f;
    }
    __syncthreads();
    float *dst = dscores ? dscores + off : nullptr;
    const float loss = listnet_slate(g, yt, yp, apply_sigmoid != 0, gscale, dscores != nullptr,
                                     [&](int i, float v) { if (active) dst[i] = v; });
    if (active && g.t == 0) slate_loss[slate] = loss;
}

// ------------------------------------------------------------------------------------ LambdaLoss
constexpr int kLambdaArrays = 7;  // sc yl gn w1 invd delta rk

__device__ __forceinline__ LambdaLds lambda_carve(float *base, int s_al) {
    LambdaLds L;
    L.sc = base;
    L.yl = L.sc + s_al;
    L.gn = L.yl + s_al;
    L.w1 = L.gn + s_al;
    L.invd = L.w1 + s_al;
    L.delta = L.invd + s_al;
    L.rk = reinterpret_cast<int *>(L.delta + s_al);
    return L;
}

template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S, int group,
              LambdaParams P, float pad, float gscale, float *__restrict__ slate_loss,
              float *__restrict__ slate_count, float *__restrict__ dscores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const int gid = threadIdx.x / group;
    const long long slate = ltr_block_id() * (blockDim.x / group) + gid;
    const bool active = slate < B;
    float *base = smem + (size_t)gid * (kLambdaArrays * s_al + group + 32);
    const LambdaLds L = lambda_carve(base, s_al);
    const SlateGroup g = make_group(S, group, base + kLambdaArrays * s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        const float y = active ? labels[off + j] : pad;
        L.sc[j] = active ? scores[off + j] : 0.f;
        stage_label(y, pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    float *dst = dscores ? dscores + off : nullptr;
    float count;
    const float loss = lambda_slate<SCH>(g, L, P, gscale, dscores != nullptr, &count,
                                         [&](int i, float v) { if (active) dst[i] = v; });
    if (active && g.t == 0) {
        slate_loss[slate] = loss;
        slate_count[slate] = count;
    }
}

// Long slates (256 <= S <= 1024, every scheme but ndcgLoss1): rank-space sweep, every unordered pair once
// (lambda_slate_blocked).  One slate per workgroup, one wave per 64 ranks.
template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_blocked_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S, LambdaParams P,
                      float pad, float gscale, float *__restrict__ slate_loss, float *__restrict__ slate_count,
                      float *__restrict__ dscores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3, s64 = (S + 63) & ~63;
    const long long slate = ltr_block_id();
    if (slate >= B) return;                      // (whole block: the y-padding of a two-dimensional grid)
    const LambdaLds L = lambda_carve(smem, s_al);
    float *rb = smem + kLambdaArrays * s_al;
    LambdaRankLds R;
    R.rs = rb;
    R.ry = rb + s64;
    R.rg = rb + 2 * s64;
    R.colacc = rb + 3 * s64;
    R.doc_of = reinterpret_cast<int *>(rb + 4 * s64);
    const SlateGroup g = make_group(S, blockDim.x, rb + 5 * s64);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += blockDim.x) {
        L.sc[j] = scores[off + j];
        stage_label(labels[off + j], pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    float *dst = dscores ? dscores + off : nullptr;
    float count;
    const float loss = lambda_slate_blocked<SCH>(g, L, R, P, gscale, dscores != nullptr, &count,
                                                 [&](int i, float v) { dst[i] = v; });
    if (g.t == 0) {
        slate_loss[slate] = loss;
        slate_count[slate] = count;
    }
}

// Full pair matrix in predicted-rank order (lambdaMask(return_losses=True), lambdaL.py:49-60): every
// (ri, rj) including padded documents, whose score/label are -inf in the reference (:13-15):
//   d = clamp(s_i - s_j, +-1e8), NaN -> 0;  G = 0 and clamped label = 0 for padded documents.
template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_pairs_fwd_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S,
                        int group, LambdaParams P, float pad, float *__restrict__ losses,
                        uint8_t *__restrict__ keep, int32_t *__restrict__ rank) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const long long slate = ltr_block_id();
    if (slate >= B) return;                      // (whole block: the y-padding of a two-dimensional grid)
    float *base = smem;
    const LambdaLds L = lambda_carve(base, s_al);
    int *dar = reinterpret_cast<int *>(base + kLambdaArrays * s_al);   // document at rank r
    const SlateGroup g = make_group(S, group, base + (kLambdaArrays + 1) * s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        L.sc[j] = scores[off + j];
        stage_label(labels[off + j], pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    lambda_prepare(g, L, P);
    for (int j = g.t; j < S; j += group) {
        dar[L.rk[j]] = j;
        if (rank) rank[off + j] = L.rk[j];
    }
    __syncthreads();
    const size_t moff = (size_t)slate * S * S;
    for (int e = g.t; e < S * S; e += group) {
        const int ri = e / S, rj = e - ri * S;
        const int i = dar[ri], j = dar[rj];
        const bool pi = L.gn[i] < 0.f, pj = L.gn[j] < 0.f;
        const float si = pi ? -INFINITY : L.sc[i], sj = pj ? -INFINITY : L.sc[j];
        float d = si - sj;
        d = (d != d) ? 0.f : fminf(fmaxf(d, -1e8f), 1e8f);
        float u, um;
        sigmoid_pair(P.sigma * d, u, um);
        const float Gi = fmaxf(L.gn[i], 0.f), Gj = fmaxf(L.gn[j], 0.f);
        const float yci = fmaxf(L.yl[i], 0.f), ycj = fmaxf(L.yl[j], 0.f);
        float ell, dl;
        lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, L.rk[i], L.rk[j], L.invd[i], L.invd[j], L.w1[i], Gi, Gj, yci, ycj),
                         u, um, ell, dl);
        losses[moff + e] = ell;
        if (keep) {
            const bool ok = !pi && !pj && (P.k <= 0 || (ri < P.k && rj < P.k)) && (SCH == 1 || L.yl[i] > L.yl[j]);
            keep[moff + e] = ok ? 1 : 0;
        }
    }
}

// Column sums of the pair matrix, c[b, rj] = sum_ri losses[b, ri, rj] (what the risk losses take from
// lambdaMask(return_losses=True): torch.sum(..., dim=1), riskLosses.py:72-83) WITHOUT materialising [B,S,S].
template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_colsum_fwd_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S,
                         int group, LambdaParams P, float pad, float *__restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const long long slate = ltr_block_id();
    if (slate >= B) return;                      // (whole block: the y-padding of a two-dimensional grid)
    float *base = smem;
    const LambdaLds L = lambda_carve(base, s_al);
    int *dar = reinterpret_cast<int *>(base + kLambdaArrays * s_al);   // document at rank r
    const SlateGroup g = make_group(S, group, base + (kLambdaArrays + 1) * s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        L.sc[j] = scores[off + j];
        stage_label(labels[off + j], pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    lambda_prepare(g, L, P);
    for (int j = g.t; j < S; j += group) dar[L.rk[j]] = j;
    __syncthreads();
    for (int rj = g.t; rj < S; rj += group) {
        const int j = dar[rj];
        const bool pj = L.gn[j] < 0.f;
        const float sj = pj ? -INFINITY : L.sc[j];
        const float Gj = fmaxf(L.gn[j], 0.f), ycj = fmaxf(L.yl[j], 0.f);
        float acc = 0.f;
        for (int ri = 0; ri < S; ++ri) {          // rank order: the order torch.sum(dim=1) walks the column in
            const int i = dar[ri];
            const bool pi = L.gn[i] < 0.f;
            const float si = pi ? -INFINITY : L.sc[i];
            float d = si - sj;
            d = (d != d) ? 0.f : fminf(fmaxf(d, -1e8f), 1e8f);
            float u, um;
            sigmoid_pair(P.sigma * d, u, um);
            const float Gi = fmaxf(L.gn[i], 0.f), yci = fmaxf(L.yl[i], 0.f);
            float ell, dl;
            lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, L.rk[i], L.rk[j], L.invd[i], L.invd[j], L.w1[i], Gi, Gj, yci, ycj),
                             u, um, ell, dl);
            acc += ell;
        }
        colsum[off + rj] = acc;
    }
}

template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_pairs_bwd_kernel(const float *__restrict__ scores, const float *__restrict__ labels, int B, int S,
                        int group, LambdaParams P, float pad, const float *__restrict__ gup, int colsum,
                        float *__restrict__ dscores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const long long slate = ltr_block_id();
    if (slate >= B) return;                      // (whole block: the y-padding of a two-dimensional grid)
    float *base = smem;
    const LambdaLds L = lambda_carve(base, s_al);
    const SlateGroup g = make_group(S, group, base + kLambdaArrays * s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        L.sc[j] = scores[off + j];
        stage_label(labels[off + j], pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    lambda_prepare(g, L, P);
    // colsum != 0: the upstream gradient is that of  c[b, rj] = sum_ri losses[b, ri, rj]  -> G[ri, rj] = gup[b, rj]
    const float *G = gup + (size_t)slate * S * (colsum ? 1 : S);
    for (int i0 = 0; i0 < S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < S;
        const bool vi = row && L.gn[i] >= 0.f;
        float gr = 0.f;
        if (vi) {
            const float si = L.sc[i], Gi = L.gn[i], yci = fmaxf(L.yl[i], 0.f);
            const int ri = L.rk[i];
            for (int j = g.cg; j < S; j += g.CG) {
                const float Gj = L.gn[j];
                if (j == i || Gj < 0.f) continue;   // padded pairs sit in a clamped / NaN->0 region: no grad
                const int rj = L.rk[j];
                const float draw = si - L.sc[j];
                if (!(fabsf(draw) <= 1e8f)) continue;
                float u, um;
                sigmoid_pair(P.sigma * draw, u, um);
                const float ycj = fmaxf(L.yl[j], 0.f);
                float ell, dl_ij, dl_ji;
                lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, ri, rj, L.invd[i], L.invd[j], L.w1[i], Gi, Gj, yci, ycj),
                                 u, um, ell, dl_ij);
                lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, rj, ri, L.invd[j], L.invd[i], L.w1[j], Gj, Gi, ycj, yci),
                                 um, u, ell, dl_ji);
                const float g_ij = colsum ? G[rj] : G[(size_t)ri * S + rj];
                const float g_ji = colsum ? G[ri] : G[(size_t)rj * S + ri];
                gr += g_ij * dl_ij - g_ji * dl_ji;
            }
        }
        const float tot = row_reduce(g, gr);
        if (row && g.cg == 0) dscores[off + i] = vi ? P.sigma * tot : 0.f;
    }
}

// ---- Column sums for ALL systems of a Lambda-type risk loss in one launch (riskLosses.py:63-83, :183-203, :294-310): the reference
// soft-maxes labels, scores and every baseline over the slate (:65-70) and then takes torch.sum(lambdaMask(p, p_true, ...,
// return_losses=True), dim=1) once per system -- three softmax launches and three column-sum launches here until round 3.  One
// workgroup per (system, query): system 0 = the model, 1..n = the baselines, n + 1 = the ideal ranking (the labels as scores).
// v[0..S) -> softmax(v) in place (torch: exp(x - max) / sum); all threads of the block call (barriers inside)
__device__ __forceinline__ void slate_softmax(const SlateGroup &g, float *v, int S) {
    float m = -INFINITY;
    for (int j = g.t; j < S; j += g.group) m = fmaxf(m, v[j]);
    m = group_max(g, m);
    float z = 0.f;
    for (int j = g.t; j < S; j += g.group) {
        const float e = expf(v[j] - m);
        v[j] = e;
        z += e;
    }
    z = group_sum(g, z);
    for (int j = g.t; j < S; j += g.group) v[j] = v[j] / z;
    __syncthreads();
}

template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_colsum_sys_fwd_kernel(const float *__restrict__ y_pred, const float *__restrict__ y_true, const float *__restrict__ y_base, int B,
                             int S, int nb, int group, LambdaParams P, float pad, float *__restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const long long blk = ltr_block_id();
    if (blk >= (long long)(nb + 2) * B) return;  // (whole block: the y-padding of a two-dimensional grid)
    const int sys = (int)(blk / B);
    const long long slate = blk - (long long)sys * B;
    float *base = smem;
    const LambdaLds L = lambda_carve(base, s_al);
    int *dar = reinterpret_cast<int *>(base + kLambdaArrays * s_al);   // document at rank r
    const SlateGroup g = make_group(S, group, base + (kLambdaArrays + 1) * s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        L.yl[j] = y_true[off + j];
        L.sc[j] = sys == 0 ? y_pred[off + j] : (sys <= nb ? y_base[(off + j) * nb + (sys - 1)] : 0.f);
    }
    __syncthreads();
    slate_softmax(g, L.yl, S);
    if (sys <= nb) slate_softmax(g, L.sc, S);    // (block-uniform)
    for (int j = g.t; j < S; j += group) {
        const float pt = L.yl[j];
        if (sys > nb) L.sc[j] = pt;              // the ideal ranking scores the documents with the (soft-maxed) labels themselves
        stage_label(pt, pad, L.yl[j], L.gn[j]);
    }
    __syncthreads();
    lambda_prepare(g, L, P);
    for (int j = g.t; j < S; j += group) dar[L.rk[j]] = j;
    __syncthreads();
    float *out = colsum + ((size_t)sys * B + slate) * S;
    for (int rj = g.t; rj < S; rj += group) {
        const int j = dar[rj];
        const bool pj = L.gn[j] < 0.f;
        const float sj = pj ? -INFINITY : L.sc[j];
        const float Gj = fmaxf(L.gn[j], 0.f), ycj = fmaxf(L.yl[j], 0.f);
        float acc = 0.f;
        for (int ri = 0; ri < S; ++ri) {          // rank order: the order torch.sum(dim=1) walks the column in
            const int i = dar[ri];
            const bool pi = L.gn[i] < 0.f;
            const float si = pi ? -INFINITY : L.sc[i];
            float d = si - sj;
            d = (d != d) ? 0.f : fminf(fmaxf(d, -1e8f), 1e8f);
            float u, um;
            sigmoid_pair(P.sigma * d, u, um);
            const float Gi = fmaxf(L.gn[i], 0.f), yci = fmaxf(L.yl[i], 0.f);
            float ell, dl;
            lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, L.rk[i], L.rk[j], L.invd[i], L.invd[j], L.w1[i], Gi, Gj, yci, ycj),
                             u, um, ell, dl);
            acc += ell;
        }
        out[rj] = acc;
    }
}

// d L / d y_pred from d L / d colsum[system 0]: the pair backward on the soft-maxed vectors, then the softmax's Jacobian
// (d p_i / d s_k = p_i (delta_ik - p_k)):  ds_k = p_k (dp_k - sum_i p_i dp_i)
template <int SCH>
__global__ void __launch_bounds__(1024)
lambda_colsum_sys_bwd_kernel(const float *__restrict__ y_pred, const float *__restrict__ y_true, int B, int S, int group, LambdaParams P,
                             float pad, const float *__restrict__ gup, float *__restrict__ dy_pred) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s_al = (S + 3) & ~3;
    const long long slate = ltr_block_id();
    if (slate >= B) return;
    float *base = smem;
    const LambdaLds L = lambda_carve(base, s_al);
    float *dp = base + kLambdaArrays * s_al;
    const SlateGroup g = make_group(S, group, dp + s_al);
    const size_t off = (size_t)slate * S;
    for (int j = g.t; j < S; j += group) {
        L.yl[j] = y_true[off + j];
        L.sc[j] = y_pred[off + j];
    }
    __syncthreads();
    slate_softmax(g, L.yl, S);
    slate_softmax(g, L.sc, S);
    for (int j = g.t; j < S; j += group) stage_label(L.yl[j], pad, L.yl[j], L.gn[j]);
    __syncthreads();
    lambda_prepare(g, L, P);
    const float *G = gup + off;
    for (int i0 = 0; i0 < S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < S;
        const bool vi = row && L.gn[i] >= 0.f;
        float gr = 0.f;
        if (vi) {
            const float si = L.sc[i], Gi = L.gn[i], yci = fmaxf(L.yl[i], 0.f);
            const int ri = L.rk[i];
            for (int j = g.cg; j < S; j += g.CG) {
                const float Gj = L.gn[j];
                if (j == i || Gj < 0.f) continue;
                const int rj = L.rk[j];
                const float draw = si - L.sc[j];
                if (!(fabsf(draw) <= 1e8f)) continue;
                float u, um;
                sigmoid_pair(P.sigma * draw, u, um);
                const float ycj = fmaxf(L.yl[j], 0.f);
                float ell, dl_ij, dl_ji;
                lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, ri, rj, L.invd[i], L.invd[j], L.w1[i], Gi, Gj, yci, ycj),
                                 u, um, ell, dl_ij);
                lambda_pair_term(P, lambda_weight<SCH>(P, L.delta, rj, ri, L.invd[j], L.invd[i], L.w1[j], Gj, Gi, ycj, yci),
                                 um, u, ell, dl_ji);
                gr += G[rj] * dl_ij - G[ri] * dl_ji;
            }
        }
        const float tot = row_reduce(g, gr);
        if (row && g.cg == 0) dp[i] = vi ? P.sigma * tot : 0.f;
    }
    __syncthreads();
    float dot = 0.f;
    for (int j = g.t; j < S; j += group) dot += L.sc[j] * dp[j];
    dot = group_sum(g, dot);
    for (int j = g.t; j < S; j += group) dy_pred[off + j] = L.sc[j] * (dp[j] - dot);
}

// --------------------------------------------------------------------------------------- ordinal
constexpr int kOrdBlock = 256;

__global__ void __launch_bounds__(kOrdBlock)
ordinal_kernel(const float *__restrict__ y_pred, const float *__restrict__ y_true, int64_t n_docs, int n,
               float pad, float *__restrict__ block_partials, float *__restrict__ dpred) {
    __shared__ float red[2 * (kOrdBlock / LTR_WAVE)];
    if (ltr_block_id() * kOrdBlock >= n_docs) return;             // (whole block: the y-padding of a two-dimensional grid)
    const int64_t doc = ltr_block_id() * kOrdBlock + threadIdx.x;
    float ls = 0.f, nv = 0.f;
    if (doc < n_docs) {
        const float y = y_true[doc];
        bool any = false;
        for (int k = 1; k <= n; ++k) {
            // with_ordinals uses the DEFAULT indicator -1 (ordinal.py:39), the mask uses `pad` (:41)
            const float t = (y == -1.f) ? -1.f : (y >= (float)k ? 1.f : 0.f);
            const bool masked = (t == pad);
            const float p = y_pred[doc * n + (k - 1)];
            float l = 0.f, gq = 0.f;
            if (!masked) {
                const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
                l = -(t * lp + (1.f - t) * l1p);
                gq = (p - t) / fmaxf((1.f - p) * p, 1e-12f);
                any = true;
            }
            ls += l;
            if (dpred) dpred[doc * n + (k - 1)] = gq;
        }
        nv = any ? 1.f : 0.f;
    }
    ls = wave_allsum(ls);
    nv = wave_allsum(nv);
    const int w = threadIdx.x / LTR_WAVE;
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) {
        red[2 * w] = ls;
        red[2 * w + 1] = nv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < kOrdBlock / LTR_WAVE; ++i) {
            a += red[2 * i];
            b += red[2 * i + 1];
        }
        block_partials[2 * ltr_block_id()] = a;
        block_partials[2 * ltr_block_id() + 1] = b;
    }
}

// Fixed-order two-column sum of the block partials: one workgroup, strided then tree.
__global__ void __launch_bounds__(1024)
reduce_pairs_kernel(const float *__restrict__ in, int64_t n, float *__restrict__ out) {
    __shared__ float red[2 * 16];
    float a = 0.f, b = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        a += in[2 * i];
        b += in[2 * i + 1];
    }
    a = wave_allsum(a);
    b = wave_allsum(b);
    const int w = threadIdx.x / LTR_WAVE;
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) {
        red[2 * w] = a;
        red[2 * w + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f, y = 0.f;
        for (int i = 0; i < 16; ++i) {
            x += red[2 * i];
            y += red[2 * i + 1];
        }
        out[0] = x;
        out[1] = y;
    }
}

__global__ void __launch_bounds__(1024)
reduce_sum_kernel(const float *__restrict__ in, int64_t n, float scale, float *__restrict__ out) {
    __shared__ float red[16];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += in[i];
    a = wave_allsum(a);
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) red[threadIdx.x / LTR_WAVE] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f;
        for (int i = 0; i < 16; ++i) x += red[i];
        out[0] = x * scale;
    }
}

inline int make_lambda_params(int scheme, int k, float sigma, float mu, float eps, int log_base, LambdaParams *P) {
    if (scheme < 0 || scheme > 7) return LTR_ERR_PARAM;
    if (log_base != LTR_LOG_BINARY && log_base != LTR_LOG_NATURAL) return LTR_ERR_PARAM;
    if (!(eps > 0.f)) return LTR_ERR_PARAM;
    P->scheme = scheme;
    P->k = k;
    P->sigma = sigma;
    P->mu = mu;
    P->eps = eps;
    P->log_scale = log_base == LTR_LOG_BINARY ? (float)(1.0 / 0.693147180559945309417) : 1.f;
    P->log_floor = log_base == LTR_LOG_BINARY ? log2f(eps) : logf(eps);
    return LTR_OK;
}

#define LTR_DISPATCH_SCHEME(scheme, CALL)          \
    switch (scheme) {                              \
        case 0: { CALL(0); break; }                \
        case 1: { CALL(1); break; }                \
        case 2: { CALL(2); break; }                \
        case 3: { CALL(3); break; }                \
        case 4: { CALL(4); break; }                \
        case 5: { CALL(5); break; }                \
        case 6: { CALL(6); break; }                \
        default: { CALL(7); break; }               \
    }

}  // namespace

extern "C" {

int ltr_abi_version(void) { return LTR_ABI_VERSION; }

const char *ltr_error_string(int code) {
    switch (code) {
        case LTR_OK: return "ok";
        case LTR_ERR_NULL: return "ltr: required pointer is NULL";
        case LTR_ERR_SHAPE: return "ltr: shape outside supported range";
        case LTR_ERR_PARAM: return "ltr: invalid parameter";
        case LTR_ERR_ALIGN: return "ltr: pointer not aligned as required";
        case LTR_ERR_IO: return "ltr: cannot open / map the input file";
        case LTR_ERR_PARSE: return "ltr: malformed input line";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ltr: unknown error";
    }
}

int ltr_reduce_sum_f32(const float *in, int64_t n, float scale, float *out, void *stream) {
    if (!in || !out) return LTR_ERR_NULL;
    if (n < 0) return LTR_ERR_SHAPE;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, n, scale, out);
    return launch_status();
}

int ltr_approxndcg_fwd_bwd(const float *scores, const float *labels, int B, int S, float alpha, float eps,
                           float pad, float grad_scale, float *slate_loss, float *dscores, void *stream) {
    if (int rc = check_slates(scores, labels, slate_loss, B, S)) return rc;
    if (B == 0) return LTR_OK;
    const SlateLaunch L = plan(B, S, kApproxArrays);
    if (int rc = allow_lds(approxndcg_kernel, L.lds)) return rc;
    hipLaunchKernelGGL(approxndcg_kernel, ltr_grid(L.grid), dim3(L.block), L.lds, (hipStream_t)stream, scores, labels,
                       B, S, L.group, alpha, eps, pad, grad_scale, slate_loss, dscores);
    return launch_status();
}

int ltr_listnet_fwd_bwd(const float *y_true, const float *y_pred, int B, int S, int apply_sigmoid,
                        float grad_scale, float *slate_loss, float *dscores, void *stream) {
    if (int rc = check_slates(y_true, y_pred, slate_loss, B, S)) return rc;
    if (B == 0) return LTR_OK;
    SlateLaunch L = plan(B, S, 2);
    // O(S) work per slate: one thread per document is plenty.
    L.group = next_pow2(S) < 64 ? 64 : (next_pow2(S) > 1024 ? 1024 : next_pow2(S));
    L.block = L.group < 256 ? 256 : L.group;
    L.gpb = L.block / L.group;
    L.grid = (B + L.gpb - 1) / L.gpb;
    L.lds = (size_t)L.gpb * (2 * ((S + 3) & ~3) + L.group + 32) * sizeof(float);
    hipLaunchKernelGGL(listnet_kernel, ltr_grid(L.grid), dim3(L.block), L.lds, (hipStream_t)stream, y_true, y_pred, B,
                       S, L.group, apply_sigmoid, grad_scale, slate_loss, dscores);
    return launch_status();
}

int ltr_lambda_fwd_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                       float mu, float eps, float pad, int log_base, float grad_scale, float *slate_loss,
                       float *slate_count, float *dscores, void *stream) {
    if (int rc = check_slates(scores, labels, slate_loss, B, S)) return rc;
    if (!slate_count) return LTR_ERR_NULL;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    if (S >= 256 && S <= 1024 && scheme != LTR_SCHEME_NDCG_LOSS1) {
        const int s_al = (S + 3) & ~3, s64 = (S + 63) & ~63;
        const size_t lds = (size_t)(kLambdaArrays * s_al + 5 * s64 + s64 + 32) * sizeof(float);
#define CALLB(SCH)                                                                                                \
    if (int rc = allow_lds(lambda_blocked_kernel<SCH>, lds)) return rc;                                           \
    hipLaunchKernelGGL(lambda_blocked_kernel<SCH>, ltr_grid(B), dim3(s64), lds, (hipStream_t)stream, scores, labels,  \
                       B, S, P, pad, grad_scale, slate_loss, slate_count, dscores)
        switch (scheme) {
            case 0: { CALLB(0); break; }
            case 2: { CALLB(2); break; }
            case 3: { CALLB(3); break; }
            case 4: { CALLB(4); break; }
            case 5: { CALLB(5); break; }
            case 6: { CALLB(6); break; }
            default: { CALLB(7); break; }
        }
#undef CALLB
        return launch_status();
    }
    const SlateLaunch L = plan(B, S, kLambdaArrays);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_kernel<SCH>, L.lds)) return rc;                                                 \
    hipLaunchKernelGGL(lambda_kernel<SCH>, ltr_grid(L.grid), dim3(L.block), L.lds, (hipStream_t)stream, scores,       \
                       labels, B, S, L.group, P, pad, grad_scale, slate_loss, slate_count, dscores)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_pairs_fwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                         float mu, float eps, float pad, int log_base, float *losses, uint8_t *keep,
                         int32_t *rank, void *stream) {
    if (int rc = check_slates(scores, labels, losses, B, S)) return rc;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = pick_group(S) < 256 ? 256 : pick_group(S);
    const size_t lds = (size_t)((kLambdaArrays + 1) * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_pairs_fwd_kernel<SCH>, lds)) return rc;                                         \
    hipLaunchKernelGGL(lambda_pairs_fwd_kernel<SCH>, ltr_grid(B), dim3(group), lds, (hipStream_t)stream, scores,      \
                       labels, B, S, group, P, pad, losses, keep, rank)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_pairs_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                         float mu, float eps, float pad, int log_base, const float *grad_losses,
                         float *dscores, void *stream) {
    if (int rc = check_slates(scores, labels, dscores, B, S)) return rc;
    if (!grad_losses) return LTR_ERR_NULL;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = pick_group(S) < 256 ? 256 : pick_group(S);
    const size_t lds = (size_t)(kLambdaArrays * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_pairs_bwd_kernel<SCH>, lds)) return rc;                                         \
    hipLaunchKernelGGL(lambda_pairs_bwd_kernel<SCH>, ltr_grid(B), dim3(group), lds, (hipStream_t)stream, scores,      \
                       labels, B, S, group, P, pad, grad_losses, 0, dscores)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_colsum_fwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, float *colsum, void *stream) {
    if (int rc = check_slates(scores, labels, colsum, B, S)) return rc;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = next_pow2(S) < 64 ? 64 : (next_pow2(S) > 1024 ? 1024 : next_pow2(S));
    const size_t lds = (size_t)((kLambdaArrays + 1) * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_colsum_fwd_kernel<SCH>, lds)) return rc;                                        \
    hipLaunchKernelGGL(lambda_colsum_fwd_kernel<SCH>, ltr_grid(B), dim3(group), lds, (hipStream_t)stream, scores,     \
                       labels, B, S, group, P, pad, colsum)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_colsum_bwd(const float *scores, const float *labels, int B, int S, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, const float *grad_colsum, float *dscores,
                          void *stream) {
    if (int rc = check_slates(scores, labels, dscores, B, S)) return rc;
    if (!grad_colsum) return LTR_ERR_NULL;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = pick_group(S) < 256 ? 256 : pick_group(S);
    const size_t lds = (size_t)(kLambdaArrays * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_pairs_bwd_kernel<SCH>, lds)) return rc;                                         \
    hipLaunchKernelGGL(lambda_pairs_bwd_kernel<SCH>, ltr_grid(B), dim3(group), lds, (hipStream_t)stream, scores,      \
                       labels, B, S, group, P, pad, grad_colsum, 1, dscores)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_colsum_sys_fwd(const float *y_pred, const float *y_true, const float *y_base, int B, int S, int n_base, int scheme,
                              int k, float sigma, float mu, float eps, float pad, int log_base, float *colsum, void *stream) {
    if (int rc = check_slates(y_pred, y_true, colsum, B, S)) return rc;
    if (n_base < 0 || n_base > 4096 || (n_base > 0 && !y_base)) return n_base > 0 && !y_base ? LTR_ERR_NULL : LTR_ERR_SHAPE;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = next_pow2(S) < 64 ? 64 : (next_pow2(S) > 1024 ? 1024 : next_pow2(S));
    const size_t lds = (size_t)((kLambdaArrays + 1) * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_colsum_sys_fwd_kernel<SCH>, lds)) return rc;                                    \
    hipLaunchKernelGGL(lambda_colsum_sys_fwd_kernel<SCH>, ltr_grid((long long)(n_base + 2) * B), dim3(group), lds, (hipStream_t)stream,   \
                       y_pred, y_true, y_base, B, S, n_base, group, P, pad, colsum)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int ltr_lambda_colsum_sys_bwd(const float *y_pred, const float *y_true, int B, int S, int scheme, int k, float sigma, float mu,
                              float eps, float pad, int log_base, const float *grad_colsum, float *dy_pred, void *stream) {
    if (int rc = check_slates(y_pred, y_true, dy_pred, B, S)) return rc;
    if (!grad_colsum) return LTR_ERR_NULL;
    LambdaParams P;
    if (int rc = make_lambda_params(scheme, k, sigma, mu, eps, log_base, &P)) return rc;
    if (B == 0) return LTR_OK;
    const int group = pick_group(S) < 256 ? 256 : pick_group(S);
    const size_t lds = (size_t)((kLambdaArrays + 1) * ((S + 3) & ~3) + group + 32) * sizeof(float);
#define CALL(SCH)                                                                                                 \
    if (int rc = allow_lds(lambda_colsum_sys_bwd_kernel<SCH>, lds)) return rc;                                    \
    hipLaunchKernelGGL(lambda_colsum_sys_bwd_kernel<SCH>, ltr_grid(B), dim3(group), lds, (hipStream_t)stream, y_pred, y_true, B, S, \
                       group, P, pad, grad_colsum, dy_pred)
    LTR_DISPATCH_SCHEME(scheme, CALL)
#undef CALL
    return launch_status();
}

int64_t ltr_ordinal_num_blocks(int64_t n_docs) { return n_docs <= 0 ? 0 : (n_docs + kOrdBlock - 1) / kOrdBlock; }

int ltr_ordinal_fwd_bwd(const float *y_pred, const float *y_true, int64_t n_docs, int n, float pad,
                        float *block_partials, float *sums, float *dpred, void *stream) {
    if (!y_pred || !y_true || !block_partials || !sums) return LTR_ERR_NULL;
    if (n_docs < 0 || n < 1 || n > 64 || ltr_ordinal_num_blocks(n_docs) > 0x7fffffff) return LTR_ERR_SHAPE;
    const int64_t nb = ltr_ordinal_num_blocks(n_docs);
    if (nb > 0) {
        hipLaunchKernelGGL(ordinal_kernel, ltr_grid(nb), dim3(kOrdBlock), 0, (hipStream_t)stream, y_pred,
                           y_true, n_docs, n, pad, block_partials, dpred);
        if (int rc = launch_status()) return rc;
    }
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_partials, nb, sums);
    return launch_status();
}

}  // extern "C"
