// ltr_slate_losses.h -- per-slate listwise losses on LDS-resident slate state (gfx950).
//
// Each function is called by ALL threads of a block (uniform barrier count); the threads of one
// SlateGroup cooperate on one slate whose scores/labels were already staged in LDS.  The same device
// code serves the standalone loss kernels (ltr_losses.hip) and the fused scorer+loss kernel
// (ltr_scorer.hip), where the scores never leave the CU.
//
// Maths and reference lines: SURVEY.md section 8(a); the closed forms are restated (and pinned against
// the reference) in oracle/ltr_oracle.py (*_closed_form).
#pragma once
#include "ltr_device.h"

namespace ltr {

// Staging convention shared by approxNDCG and lambdaLoss:
//   sc[j] : score                                   yl[j] : label, -inf if padded (y == pad)
//   gn[j] : 2^max(y,0) - 1, or -1 if padded         (turned into G_j = gain / maxDCG in place)
__device__ __forceinline__ void stage_label(float y, float pad, float &yl, float &gn) {
    bool p = (y == pad);
    yl = p ? -INFINITY : y;
    gn = p ? -1.f : (exp2f(fmaxf(y, 0.f)) - 1.f);
}

// Ideal DCG by rank counting on the labels (approxNDCG.py:28,43 / lambdaL.py:19,39): ties by index.
// kk > 0 truncates at rank kk.  Returns max(sum, eps) to every thread.
__device__ __forceinline__ float ideal_dcg(const SlateGroup &g, const float *yl, const float *gn, float eps,
                                           int kk) {
    float acc = 0.f;
    for (int i0 = 0; i0 < g.S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < g.S;
        float cnt = 0.f;
        if (row) {
            const float yi = yl[i];
            for (int j = g.cg; j < g.S; j += g.CG) {
                const float yj = yl[j];
                cnt += ((yj > yi) || (yj == yi && j < i)) ? 1.f : 0.f;
            }
        }
        const float r = row_reduce(g, cnt);
        if (row && g.cg == 0) {
            const float gi = gn[i];
            if (gi > 0.f && (kk <= 0 || r < (float)kk)) acc += gi / log2f(2.f + r);
        }
    }
    return fmaxf(group_sum(g, acc), eps);
}

// ------------------------------------------------------------------------------------------------
// approxNDCG (losses/approxNDCG.py:7-53).  Returns the slate loss -sum_i G_i / log2(1 + pos_i) to every
// thread; if want_grad, calls store(i, gscale * dloss/ds_i) once per document (0 for padded documents).
// All arrays are LDS, length S rounded up to a multiple of 4 (`s_al`); gg, uu, mk are scratch.
// gn is read-only here: normalisation by maxDCG is applied to the reduced sums, not per document.
//
// Pair sweep: row i belongs to a row lane, the columns j of a row are split into CG CONTIGUOUS blocks (one per
// column group) so a thread reads its block with 16-byte LDS broadcast loads and runs 4 independent pairs
// per step.  Fast path (score range alpha*(max-min) <= 160): one exponential PER DOCUMENT,
//     u_k = exp(alpha (s_k - mid)),   sigmoid(-alpha (s_i - s_j)) = u_j / (u_i + u_j),
// i.e. one v_rcp per pair and no v_exp.  Wider ranges (where u would leave fp32) take the per-pair exp path.
typedef float lds_f4 __attribute__((ext_vector_type(4)));

struct NoStamp {
    __device__ __forceinline__ void operator()(int) const {}
};

// Optional extra LDS scratch that enables the NO-CLAMP path of approx_ndcg_slate (s_al floats):
//   um[j] = u_j for real documents, 0 for padded ones (the numerators).
// Contract: the 32 integer label bins  (int *)(g.part + 32)  are ZERO on entry (approx_ndcg_init before a barrier
// ahead of the first call; the function re-zeroes them before it returns).
struct ApproxScratch {
    float *um = nullptr;
};

__device__ __forceinline__ void approx_ndcg_init(const SlateGroup &g) {
    if (g.t < 32) reinterpret_cast<int *>(g.part + 32)[g.t] = 0;
}

// No-clamp path ("ultra"), taken per slate when
//   (a) every |alpha (s_k - s_0)| <= 8   -> u in [3.4e-4, 2981]: products of up to eight (u_i + u_j) stay inside fp32, and
//       every pair sigmoid is >= sigmoid(-16) = 1.1e-7 >= eps, so no max(., eps) clamp of approxNDCG.py:49 is active;
//   (b) every clamped label max(y, 0) is an integer <= 15 (graded relevance; gains <= 32767 keep the batched products in range) -> the ideal DCG comes from a 32-bin
//       histogram (LDS integer adds: exact, order-free) instead of an S^2 rank count.
// Then FOUR pair terms share ONE v_rcp_f32:
//   sum_k n_k / d_k  (k = 0..3)  =  (N01 D23 + N23 D01) / (D01 D23),   N01 = n0 d1 + n1 d0,  D01 = d0 d1, ...
// 15 plain fp32 VALU instructions + one quarter-rate v_rcp_f32 per FOUR ordered pairs in sweep 1 (~4.5 issue slots per
// pair; was ~13: add, rcp = 4 slots, mul, max, fma and 5 of rank counting) and 23 + rcp in sweep 2.  The library is built
// with -fno-slp-vectorize: packed fp32 (v_pk_*_f32) has no rate advantage on CDNA4 and its operand-pair moves are pure cost.
// sweep 1:  pos_i - 1 = sum_j um_j / (u_i + u_j) - 1/2 [i real]
// sweep 2:  d loss / d s_k = alpha u_k sum_j um_j (g_j - g_k) / (u_k + u_j)^2        (t_kj = u_k u_j / (u_k + u_j)^2)
template <int JB = 0, int UNR = 2, class Store, class Stamp = NoStamp>
__device__ __forceinline__ float approx_ndcg_slate(const SlateGroup &g, float *sc, float *yl, const float *gn,
                                                   float *gg, float *uu, float *mk, float alpha, float eps,
                                                   float gscale, bool want_grad, Store store, Stamp stamp = Stamp(),
                                                   ApproxScratch xs = ApproxScratch()) {
    const int s_al = (g.S + 3) & ~3;
    // column QUADS interleaved between the CG lanes of a row (lane cg takes quads cg, cg + CG, ...): the CG broadcast addresses of one
    // ds_read_b128 then sit 16 bytes apart instead of a whole column block apart (= the same banks: a CG-way conflict, 131 M conflict
    // cycles per 100 k slates at S = 128, profiles/r04_rocprof_pmc_summaries.json).  (JB: kept for callers that name a block length.)
    const int nq = s_al >> 2, qs = 4 * g.CG;        // quads per row; column stride between a lane's quads
    const int jq0 = 4 * g.cg;                       // this lane's first column
    const bool ultra_ok = xs.um != nullptr && eps <= 1e-7f;
    int *bins = reinterpret_cast<int *>(g.part + 32);

    // Per-document exponentials relative to the first document's score.  The fast path is valid while every
    // |alpha (s_k - s_0)| <= 69 (u in [1e-30, 1e30]: sums and ratios stay in fp32 range); one flag word per wave,
    // combined behind the SAME barrier that publishes uu / mk / gg, decides ultra vs fast vs slow for the whole slate.
    const float sref = alpha * sc[0];
    bool bad = false, noultra = false;
    float ymax = 0.f;
    for (int j = g.t; j < s_al; j += g.group) {
        const bool real = j < g.S && gn[j] >= 0.f;
        const float x = real ? alpha * sc[j] - sref : 0.f;
        bad = bad || !(fabsf(x) <= 69.f);          // NaN scores take the slow path too
        const float u = real ? expf(x) : 1.f;
        mk[j] = real ? 1.f : 0.f;
        uu[j] = u;
        if (ultra_ok) {
            noultra = noultra || !(fabsf(x) <= 8.f);
            xs.um[j] = real ? u : 0.f;
            if (real) {
                const float yc = fmaxf(yl[j], 0.f);
                const bool isint = yc <= 15.f && yc == floorf(yc);
                noultra = noultra || !isint;
                ymax = fmaxf(ymax, isint ? yc : 0.f);
                if (isint) __hip_atomic_fetch_add(&bins[(int)yc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (j >= g.S) {
            sc[j] = 0.f;
            yl[j] = -INFINITY;
        }
        gg[j] = 0.f;
    }
    {
        const float fl = (__ballot(bad) ? 2.f : 0.f) + ((!ultra_ok || __ballot(noultra)) ? 1.f : 0.f);
        const float ym = ultra_ok ? wave_allmax(ymax) : 0.f;
        if ((threadIdx.x & (LTR_WAVE - 1)) == 0) {
            g.part[g.wig] = fl;
            g.part[16 + g.wig] = ym;
        }
    }
    __syncthreads();
    stamp(10);
    bool fast = true, ultra = ultra_ok;
    int ytop = 0;
    for (int w = 0; w < g.nw; ++w) {
        const float fl = g.part[w];
        fast = fast && fl < 2.f;
        ultra = ultra && fl == 0.f;
        ytop = max(ytop, (int)g.part[16 + w]);
    }
    stamp(11);

    // One sweep per row: label rank by counting (ideal DCG term, approxNDCG.py:28,43) and the soft rank
    // pos_i = 1 + sum_{j != i, both valid} max(sigmoid(-alpha (s_i - s_j)), eps)            (:47-49)
    float idcg_acc = 0.f, loss_acc = 0.f;
    for (int i0 = 0; i0 < g.S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < g.S;
        const bool vi = row && gn[i] >= 0.f;
        float p = 0.f, cnt = 0.f;
        if (ultra) {
            if (vi) {
                const float ui = uu[i];
                // plain fp32 issue (packed fp32 has no rate advantage on CDNA4 and costs operand-pair moves); two independent
                // accumulator chains, eight columns per trip
                auto quad = [&](int j, float acc) {
                    const lds_f4 u = *reinterpret_cast<const lds_f4 *>(uu + j);
                    const lds_f4 n = *reinterpret_cast<const lds_f4 *>(xs.um + j);
                    const float d0 = ui + u[0], d1 = ui + u[1], d2 = ui + u[2], d3 = ui + u[3];
                    const float D01 = d0 * d1, D23 = d2 * d3;
                    const float N01 = fmaf(n[0], d1, n[1] * d0), N23 = fmaf(n[2], d3, n[3] * d2);
                    return fmaf(fmaf(N01, D23, N23 * D01), ltr_rcp(D01 * D23), acc);
                };
                float acc0 = 0.f, acc1 = 0.f;
                int j = jq0;
#pragma unroll UNR
                for (; j + qs < s_al; j += 2 * qs) {
                    acc0 = quad(j, acc0);
                    acc1 = quad(j + qs, acc1);
                }
                if (j < s_al) acc0 = quad(j, acc0);
                p = acc0 + acc1;
                if (((i >> 2) & (g.CG - 1)) == g.cg) p -= 0.5f;                                 // the j == i term: u_i / (2 u_i)
            }
            // ideal-DCG term of RANK i from the label histogram: the document at sorted position i has the label v with
            // (#labels > v) <= i < (#labels >= v); labels 0 carry no gain
            if (row && g.cg == 0) {
                int cum = 0;
                float gr = 0.f;
                for (int v = ytop; v >= 1; --v) {
                    const int c = bins[v];
                    gr = (i >= cum && i < cum + c) ? exp2f((float)v) - 1.f : gr;
                    cum += c;
                }
                cnt = gr;                                                                        // gain at sorted rank i
            }
        } else {
        if (row) {
            const float yi = yl[i];
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll 2
            for (int j = jq0; j < s_al; j += qs) {
                const lds_f4 y = *reinterpret_cast<const lds_f4 *>(yl + j);
                c0 += ((y[0] > yi) || (y[0] == yi && j + 0 < i)) ? 1.f : 0.f;
                c1 += ((y[1] > yi) || (y[1] == yi && j + 1 < i)) ? 1.f : 0.f;
                c2 += ((y[2] > yi) || (y[2] == yi && j + 2 < i)) ? 1.f : 0.f;
                c3 += ((y[3] > yi) || (y[3] == yi && j + 3 < i)) ? 1.f : 0.f;
            }
            cnt = (c0 + c1) + (c2 + c3);
        }
        if (vi) {
            if (fast) {
                const float ui = uu[i];
                float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll 2
                for (int j = jq0; j < s_al; j += qs) {
                    const lds_f4 u = *reinterpret_cast<const lds_f4 *>(uu + j);
                    const lds_f4 m = *reinterpret_cast<const lds_f4 *>(mk + j);
                    p0 = fmaf(m[0], fmaxf(u[0] * ltr_rcp(ui + u[0]), eps), p0);
                    p1 = fmaf(m[1], fmaxf(u[1] * ltr_rcp(ui + u[1]), eps), p1);
                    p2 = fmaf(m[2], fmaxf(u[2] * ltr_rcp(ui + u[2]), eps), p2);
                    p3 = fmaf(m[3], fmaxf(u[3] * ltr_rcp(ui + u[3]), eps), p3);
                }
                p = (p0 + p1) + (p2 + p3);
                if (((i >> 2) & (g.CG - 1)) == g.cg) p -= fmaxf(ui * ltr_rcp(ui + ui), eps);   // the j == i term
            } else {
                const float si = sc[i];
                for (int j4 = jq0; j4 < s_al; j4 += qs)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int j = j4 + e4;
                        const float e = __expf(alpha * (si - sc[j]));
                        const float c = fmaxf(ltr_rcp(1.f + e), eps);
                        p += (j != i && mk[j] != 0.f) ? c : 0.f;
                    }
            }
        }
        }
        const float r = ultra ? 0.f : row_reduce(g, cnt);
        const float pos = 1.f + row_reduce(g, p);
        if (ultra) {
            // per-row epilogue on the hardware transcendentals (v_log_f32 / v_rcp_f32, 1 ulp each): the IEEE log2f / division
            // sequences of the general path are ~100 instructions per row, executed by every wave of the workgroup
            if (row && g.cg == 0 && cnt > 0.f) idcg_acc = fmaf(cnt, ltr_rcp(__log2f(2.f + (float)i)), idcg_acc);
            if (vi && g.cg == 0) {
                const float gain = gn[i];
                const float iL = ltr_rcp(__log2f(1.f + pos));
                const float gl = gain * iL;
                loss_acc += gl;
                gg[i] = gl * iL * ltr_rcp((1.f + pos) * LTR_LN2);    // d(-sum gain/L)/d pos_i, not yet / maxDCG
            }
        } else if (vi && g.cg == 0) {
            const float gain = gn[i];
            const float L = log2f(1.f + pos);
            if (gain > 0.f) idcg_acc += gain / log2f(2.f + r);
            loss_acc += gain / L;
            const float gi = gain / (L * L * (1.f + pos) * LTR_LN2);   // d(-sum gain/L)/d pos_i, not yet / maxDCG
            gg[i] = gi;
        }
    }
    stamp(12);
    // wave partials -> LDS, ONE barrier (it also publishes gg), every thread adds the waves in fixed order
    idcg_acc = wave_allsum(idcg_acc);
    loss_acc = wave_allsum(loss_acc);
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) {
        g.red[2 * g.wig] = idcg_acc;
        g.red[2 * g.wig + 1] = loss_acc;
    }
    __syncthreads();
    if (ultra_ok && g.t < 32) bins[g.t] = 0;                    // every reader of the histogram is past the barrier
    idcg_acc = 0.f;
    loss_acc = 0.f;
    for (int w = 0; w < g.nw; ++w) {
        idcg_acc += g.red[2 * w];
        loss_acc += g.red[2 * w + 1];
    }
    stamp(13);
    const float inv_idcg = 1.f / fmaxf(idcg_acc, eps);          // maxDCG clamp (:43)
    const float total = loss_acc * inv_idcg;
    if (!want_grad) return -total;

    // d loss / d s_k = alpha * sum_j t_kj (g_j [c_jk >= eps] - g_k [c_kj >= eps]),  t = c_kj c_jk
    const float kscale = alpha * gscale * inv_idcg;
    for (int i0 = 0; i0 < g.S; i0 += g.sp) {
        const int k = i0 + g.ri;
        const bool row = k < g.S;
        const bool vk = row && gn[k] >= 0.f;
        float a = 0.f;
        if (vk && ultra) {
            const float gk = gg[k], uk = uu[k];
            auto quad = [&](int j, float acc) {
                const lds_f4 u = *reinterpret_cast<const lds_f4 *>(uu + j);
                const lds_f4 n = *reinterpret_cast<const lds_f4 *>(xs.um + j);
                const lds_f4 gj = *reinterpret_cast<const lds_f4 *>(gg + j);
                float q0 = uk + u[0], q1 = uk + u[1], q2 = uk + u[2], q3 = uk + u[3];
                q0 *= q0, q1 *= q1, q2 *= q2, q3 *= q3;
                const float t0 = n[0] * (gj[0] - gk), t1 = n[1] * (gj[1] - gk);                 // exactly 0 at j == k
                const float t2 = n[2] * (gj[2] - gk), t3 = n[3] * (gj[3] - gk);
                const float Q01 = q0 * q1, Q23 = q2 * q3;
                const float N01 = fmaf(t0, q1, t1 * q0), N23 = fmaf(t2, q3, t3 * q2);
                return fmaf(fmaf(N01, Q23, N23 * Q01), ltr_rcp(Q01 * Q23), acc);
            };
            float acc0 = 0.f, acc1 = 0.f;
            int j = jq0;
#pragma unroll UNR
            for (; j + qs < s_al; j += 2 * qs) {
                acc0 = quad(j, acc0);
                acc1 = quad(j + qs, acc1);
            }
            if (j < s_al) acc0 = quad(j, acc0);
            a = uk * (acc0 + acc1);
        } else if (vk) {
            const float gk = gg[k];
            if (fast) {
                const float uk = uu[k];
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
                for (int j = jq0; j < s_al; j += qs) {
                    const lds_f4 u = *reinterpret_cast<const lds_f4 *>(uu + j);
                    const lds_f4 m = *reinterpret_cast<const lds_f4 *>(mk + j);
                    const lds_f4 gj = *reinterpret_cast<const lds_f4 *>(gg + j);
#define LTR_PAIR(e, acc)                                                            \
    {                                                                               \
        const float r = ltr_rcp(uk + u[e]);                                       \
        const float ckj = u[e] * r, cjk = uk * r;                                   \
        const float term = (cjk >= eps ? gj[e] : 0.f) - (ckj >= eps ? gk : 0.f);    \
        acc = fmaf(m[e] * (ckj * cjk), term, acc);                                  \
    }
                    LTR_PAIR(0, a0) LTR_PAIR(1, a1) LTR_PAIR(2, a2) LTR_PAIR(3, a3)
#undef LTR_PAIR
                }
                a = (a0 + a1) + (a2 + a3);      // j == k contributes exactly 0 (term = g_k - g_k)
            } else {
                const float sk = sc[k];
                for (int j4 = jq0; j4 < s_al; j4 += qs)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int j = j4 + e4;
                        const float e = __expf(alpha * (sk - sc[j]));
                        const float ckj = ltr_rcp(1.f + e);                 // sigmoid(-alpha (s_k - s_j))
                        const float cjk = (e < 1e30f) ? e * ckj : 1.f;        // sigmoid(-alpha (s_j - s_k))
                        const float term = (cjk >= eps ? gg[j] : 0.f) - (ckj >= eps ? gk : 0.f);
                        a += (j != k && mk[j] != 0.f) ? ckj * cjk * term : 0.f;
                    }
            }
        }
        const float tot = row_reduce(g, a);
        if (row && g.cg == 0) store(k, vk ? kscale * tot : 0.f);
    }
    stamp(14);
    return -total;
}

// invd[r] = 1 / log2(2 + r), r < n: the rank discounts of the ideal DCG (approxNDCG.py:41-43), same expression as the per-row epilogue
// invd[n + c - 1] = sum_{r < c} invd[r], c = 1 .. n: the discount mass of the first c positions (ideal_dcg_by_counts below).
__device__ __forceinline__ void ltr_fill_inv_discount(float *invd, int n, int tid, int nthreads) {
    for (int r = tid; r < n; r += nthreads) {
        invd[r] = ltr_rcp(__log2f(2.f + (float)r));
        float cum = 0.f;
        for (int q = 0; q <= r; ++q) cum += ltr_rcp(__log2f(2.f + (float)q));      // once per kernel
        invd[n + r] = cum;
    }
}

// Ideal DCG of a slate with integer grades 0 .. 15 (gain 2^y - 1; approxNDCG.py:38-43), WITHOUT sorting and without a per-grade
// placement pass: the gain of grade y is the sum of the increments 2^(w-1), w = 1 .. y, and increment w applies to every
// document with a grade >= w -- in the descending order those are the FIRST c_w = #{labels >= w} positions.  Hence
//     idcg = sum_w 2^(w-1) D[c_w],   D[c] = sum_{r < c} 1 / log2(2 + r)   (table dcum[c - 1], ltr_fill_inv_discount).
// Per grade present: one ballot + popcount per slot and one v_writelane; then ONE table read, one multiply and a 16-lane sum.
// (The histogram loop this replaces placed every grade's gain by position: ~70 dependent vector instructions, 2.3 k cycles alone and
// 7.8 k beside another workgroup's fp32 MFMA stream, which leaves a vector instruction an issue slot only every ~32 cycles:
// profiles/r04_fcw_stamps_fp32_prologue_detail.jsonl.)  yi: the lane's grades, 0 for padded / inactive documents.  Wave-uniform result.
template <int W, int SLOTS>
__device__ __forceinline__ void grade_counts(const int (&yi)[SLOTS], int &cl) {      // lane w - 1 of cl <- c_w, w = W .. 15, until c_w = 0
    if constexpr (W <= 15) {
        int c = 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) c += __popcll(__ballot(yi[sl] >= W));
        if (c == 0) return;                         // no label reaches W: none reaches W + 1 either
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cl) : "s"(c), "n"(W - 1));      // (one SGPR per VOP3 on gfx9: the lane is an immediate)
        grade_counts<W + 1, SLOTS>(yi, cl);
    }
}
template <int SLOTS>
__device__ __forceinline__ float ideal_dcg_by_counts(const int (&yi)[SLOTS], const float *dcum, int lane) {
    int cl = 0;
    grade_counts<1, SLOTS>(yi, cl);
    float v = cl > 0 ? dcum[cl - 1] * __builtin_ldexpf(1.f, lane) : 0.f;         // lanes >= 15 hold 0
    v += LTR_DPP(v, LTR_DPP_XOR1);
    v += LTR_DPP(v, LTR_DPP_XOR2);
    v += LTR_DPP(v, LTR_DPP_HALF_MIRROR);
    v += LTR_DPP(v, LTR_DPP_MIRROR);
    return lane_bcast(v, 0);
}

// ------------------------------------------------------------------------------------------------
// approxNDCG for the FUSED kernels (ltr_fcw.h, slate_pipeline_kernel): same maths and the same three paths as
// approx_ndcg_slate above (losses/approxNDCG.py:7-53), restructured so that a slate costs ONE workgroup barrier instead of four.
//
// Why: inside the fused kernels a slate's loss ran 3-6 x slower than its throughput cost (profiles/r03_fcw_stamps_fp32.jsonl:
// 16 k cycles per tile, 6.3 k of them pure synchronisation), and an fp32 MFMA stream of the co-resident workgroup leaves a VALU
// wave no issue slot at all (profiles/r04_simd_coissue_microbench.jsonl) -- so every instruction and every barrier of this phase
// is serial time on its SIMD.  What changed:
//   * geometry (slate length S, NW waves per slate, CG = 64 NW / S lanes per row) is compile-time: no runtime group arithmetic;
//   * the prologue is WAVE-PRIVATE: every wave of the slate evaluates ALL S per-document exponentials (1-2 per lane), the path
//     flags (ballots), the label histogram (ballot + popcount per grade, counts in SGPRs) and -- on the no-clamp path -- the whole
//     ideal DCG for itself, and writes the u / um / mask arrays in full: all waves write bit-identical values and every wave
//     reads back only what it wrote itself (LDS is in order per wave), so no barrier separates the prologue from sweep 1;
//   * the column quads of a row's CG lanes are INTERLEAVED (lane cg takes quads cg, cg + CG, ...): the CG broadcast addresses of
//     one ds_read_b128 are 16 bytes apart instead of a whole column block (= same banks: a 2- / 4-way conflict before);
//   * one barrier publishes the per-document gradient weights gg (and the per-wave loss / ideal-DCG partials) between the two
//     sweeps; the caller's barrier behind the call publishes the score gradients.
// `score(j)` returns the score of document j (the fused kernels sum their per-wave partials here: no separate exchange pass);
// every wave calls it for all S documents.  sc / uu / um / mk / gg: LDS [S]; red: LDS [2 NW], private to the slate; invd: LDS [>= S],
// invd[r] = 1 / log2(2 + r) (ltr_fill_inv_discount once per kernel: as registers those per-lane constants were hoisted and spilled).
// t = the thread's index inside the slate's 64 NW threads.  All threads of the BLOCK must call (one __syncthreads inside).
template <int S, int NW, bool WRITE_SC, class Score, class Store, class Stamp = NoStamp>
__device__ __forceinline__ float approx_ndcg_fused(int t, Score score, float *sc, const float *yl, const float *gn, float *gg,
                                                   float *uu, float *um, float *mk, float *red, const float *invd, float alpha,
                                                   float eps, float gscale, Store store, Stamp stamp = Stamp()) {
    static_assert(S == 32 || S == 64 || S == 128, "fused slates are 32, 64 or 128 documents");
    constexpr int CG = 64 * NW / S;                 // lanes per row
    static_assert(CG == 2 || CG == 4, "two or four lanes per document row");
    constexpr int RPW = 64 / CG;                    // rows per wave
    constexpr int QPL = S / 4 / CG;                 // column quads per lane
    static_assert(QPL % 2 == 0, "two accumulator chains");
    constexpr int SLOTS = (S + 63) / 64;            // documents per lane in the prologue
    const int lane = t & 63;
    const int wig = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cg = lane & (CG - 1);
    const int i = wig * RPW + lane / CG;            // this lane's row
    const bool ultra_ok = eps <= 1e-7f;

    // ---- prologue (wave-private)
    float sv[SLOTS], yc[SLOTS];
    bool real[SLOTS];
    bool bad = false, noultra = false;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int j = lane + 64 * sl;
        const bool act = (S % 64 == 0) || j < S;
        sv[sl] = act ? score(act ? j : 0) : 0.f;
        real[sl] = act && gn[act ? j : 0] >= 0.f;
        yc[sl] = act ? fmaxf(yl[act ? j : 0], 0.f) : 0.f;
    }
    stamp(15);
    const float sref = alpha * lane_bcast(sv[0], 0);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int j = lane + 64 * sl;
        const bool act = (S % 64 == 0) || j < S;
        const float x = real[sl] ? alpha * sv[sl] - sref : 0.f;
        bad = bad || !(fabsf(x) <= 69.f);            // NaN scores take the slow path too
        noultra = noultra || !(fabsf(x) <= 8.f);
        const float u = real[sl] ? expf(x) : 1.f;
        const bool isint = yc[sl] <= 15.f && yc[sl] == floorf(yc[sl]);
        noultra = noultra || (real[sl] && !isint);
        if (act) {
            uu[j] = u;
            um[j] = real[sl] ? u : 0.f;
            mk[j] = real[sl] ? 1.f : 0.f;
            if (WRITE_SC) sc[j] = sv[sl];
        }
    }
    stamp(11);
    const bool fast = __ballot(bad) == 0ull;
    const bool ultra = ultra_ok && fast && __ballot(noultra) == 0ull;
    const float *ub = uu + 4 * cg, *nb = um + 4 * cg, *mb = mk + 4 * cg, *gb = gg + 4 * cg;
    const bool mine = ((i >> 2) & (CG - 1)) == cg;          // the j == i column is one of this lane's
    const bool vi = gn[i] >= 0.f;
    const float ui = uu[i];
    float total = 0.f;
    // state handed from the first half of a path (up to the barrier) to its second half
    float rr[4 * QPL];                       // pair reciprocals of the no-clamp path, sweep 1 -> sweep 2
    float idcg_u = 0.f, gout = 0.f, loss_acc = 0.f, idcg_acc = 0.f;

    // ================= no-clamp path: integer grades, |alpha (s_k - s_0)| <= 8 =================
    auto ultra_first = [&]() {
        // ideal DCG from per-grade counts (ideal_dcg_by_counts): wave-uniform, one LDS read
        {
            int yi[SLOTS];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) yi[sl] = real[sl] ? (int)yc[sl] : 0;
            idcg_u = ideal_dcg_by_counts<SLOTS>(yi, invd + 128, lane);
        }
        stamp(10);
        // ---- sweep 1: pos_i - 1 = sum_j um_j / (u_i + u_j) - 1/2 [i real]
        // The pair reciprocals r_ij = 1 / (u_i + u_j) of this lane's 4 QPL columns STAY IN REGISTERS from sweep 1 to sweep 2
        // (64 VGPRs at two lanes per row, 32 at four): sweep 1 is add + v_rcp_f32 + fma per pair, sweep 2 one multiply and two fma
        // -- 7 issue slots per ordered pair against 11.25 for the two four-pairs-per-reciprocal sweeps (whose batching saves
        // nothing where a lone wave issues v_rcp_f32 in two plain slots, profiles/r04_simd_coissue_microbench.jsonl).
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int m = 0; m < QPL; ++m) {
            const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
            const lds_f4 n = *reinterpret_cast<const lds_f4 *>(nb + 4 * CG * m);
#pragma unroll
            for (int e = 0; e < 4; ++e) rr[4 * m + e] = ltr_rcp(ui + u[e]);
            acc0 = fmaf(n[0], rr[4 * m + 0], acc0);
            acc1 = fmaf(n[1], rr[4 * m + 1], acc1);
            acc0 = fmaf(n[2], rr[4 * m + 2], acc0);
            acc1 = fmaf(n[3], rr[4 * m + 3], acc1);
        }
        float p = acc0 + acc1;
        if (mine) p -= 0.5f;                                     // the j == i term: u_i / (2 u_i)
        p += LTR_DPP(p, LTR_DPP_XOR1);                           // the CG partials of a row sit in adjacent lanes
        if (CG == 4) p += LTR_DPP(p, LTR_DPP_XOR2);
        const float pos = 1.f + p;
        if (vi && cg == 0) {
            // per-row epilogue on the hardware transcendentals (v_log_f32 / v_rcp_f32, 1 ulp each)
            const float iL = ltr_rcp(__log2f(1.f + pos));
            const float gl = gn[i] * iL;
            loss_acc = gl;
            gout = gl * iL * ltr_rcp((1.f + pos) * LTR_LN2);     // d(-sum gain/L)/d pos_i, not yet / maxDCG
        }
        // column j of sweep 2 needs only um_j g_j (um_i = u_i for a real document)
        if (cg == 0) gg[i] = gout * ui;                          // 0 for padded documents
        stamp(12);
        loss_acc = wave_allsum(loss_acc);
    };
    auto ultra_second = [&](float loss_sum) {
        const float inv_idcg = 1.f / fmaxf(idcg_u, eps);         // maxDCG clamp (approxNDCG.py:43)
        total = loss_sum * inv_idcg;
        // ---- sweep 2: d loss / d s_k = alpha u_k sum_j um_j (g_j - g_k) / (u_k + u_j)^2       (k = i)
        const float gk = LTR_DPP(gout, CG == 2 ? 0xA0 : 0x00);   // g_k from the row's first lane (quad_perm broadcast)
        // a_k = u_k (sum_j um_j g_j r_kj^2 - g_k sum_j um_j r_kj^2); the j == k terms of the two sums cancel
        float A0 = 0.f, A1 = 0.f, T0 = 0.f, T1 = 0.f;
#pragma unroll
        for (int m = 0; m < QPL; ++m) {
            const lds_f4 mg = *reinterpret_cast<const lds_f4 *>(gb + 4 * CG * m);
            const lds_f4 n = *reinterpret_cast<const lds_f4 *>(nb + 4 * CG * m);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const float q0 = rr[4 * m + e] * rr[4 * m + e], q1 = rr[4 * m + e + 1] * rr[4 * m + e + 1];
                A0 = fmaf(mg[e], q0, A0);
                T0 = fmaf(n[e], q0, T0);
                A1 = fmaf(mg[e + 1], q1, A1);
                T1 = fmaf(n[e + 1], q1, T1);
            }
        }
        float a = ui * ((A0 + A1) - gk * (T0 + T1));
        a += LTR_DPP(a, LTR_DPP_XOR1);
        if (CG == 4) a += LTR_DPP(a, LTR_DPP_XOR2);
        if (cg == 0) store(i, vi ? alpha * gscale * inv_idcg * a : 0.f);
        stamp(14);
    };

    // ================= general paths: label ranks by counting; one exponential per document (fast) or per pair =================
    auto general_first = [&]() {
        stamp(10);
        stamp(11);
        float p = 0.f, cnt = 0.f;
        const float yi = yl[i];
        {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            const float *yb = yl + 4 * cg;
            for (int m = 0; m < QPL; ++m) {
                const int j = 4 * (cg + CG * m);
                const lds_f4 y = *reinterpret_cast<const lds_f4 *>(yb + 4 * CG * m);
                c0 += ((y[0] > yi) || (y[0] == yi && j + 0 < i)) ? 1.f : 0.f;
                c1 += ((y[1] > yi) || (y[1] == yi && j + 1 < i)) ? 1.f : 0.f;
                c2 += ((y[2] > yi) || (y[2] == yi && j + 2 < i)) ? 1.f : 0.f;
                c3 += ((y[3] > yi) || (y[3] == yi && j + 3 < i)) ? 1.f : 0.f;
            }
            cnt = (c0 + c1) + (c2 + c3);
        }
        if (fast) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            for (int m = 0; m < QPL; ++m) {
                const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
                const lds_f4 mm = *reinterpret_cast<const lds_f4 *>(mb + 4 * CG * m);
                p0 = fmaf(mm[0], fmaxf(u[0] * ltr_rcp(ui + u[0]), eps), p0);
                p1 = fmaf(mm[1], fmaxf(u[1] * ltr_rcp(ui + u[1]), eps), p1);
                p2 = fmaf(mm[2], fmaxf(u[2] * ltr_rcp(ui + u[2]), eps), p2);
                p3 = fmaf(mm[3], fmaxf(u[3] * ltr_rcp(ui + u[3]), eps), p3);
            }
            p = (p0 + p1) + (p2 + p3);
            if (mine) p -= fmaxf(ui * ltr_rcp(ui + ui), eps);    // the j == i term
        } else {
            const float si = sc[i];
            for (int m = 0; m < QPL; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * (cg + CG * m) + e;
                    const float ex = __expf(alpha * (si - sc[j]));
                    const float c = fmaxf(ltr_rcp(1.f + ex), eps);
                    p += (j != i && mk[j] != 0.f) ? c : 0.f;
                }
        }
        p += LTR_DPP(p, LTR_DPP_XOR1);
        cnt += LTR_DPP(cnt, LTR_DPP_XOR1);
        if (CG == 4) {
            p += LTR_DPP(p, LTR_DPP_XOR2);
            cnt += LTR_DPP(cnt, LTR_DPP_XOR2);
        }
        const float pos = 1.f + p;
        if (vi && cg == 0) {
            const float gain = gn[i];
            const float Lg = log2f(1.f + pos);
            if (gain > 0.f) idcg_acc = gain / log2f(2.f + cnt);
            loss_acc = gain / Lg;
            gout = gain / (Lg * Lg * (1.f + pos) * LTR_LN2);     // d(-sum gain/L)/d pos_i, not yet / maxDCG
        }
        if (cg == 0) gg[i] = gout;                               // 0 for padded documents
        stamp(12);
        loss_acc = wave_allsum(loss_acc);
        idcg_acc = wave_allsum(idcg_acc);
    };
    auto general_second = [&](float loss_sum, float idcg_sum) {
        const float inv_idcg = 1.f / fmaxf(idcg_sum, eps);       // maxDCG clamp (approxNDCG.py:43)
        total = loss_sum * inv_idcg;
        // ---- sweep 2: d loss / d s_k = alpha * sum_j t_kj (g_j [c_jk >= eps] - g_k [c_kj >= eps]),  t = c_kj c_jk   (k = i)
        const float gk = gg[i];
        float a = 0.f;
        if (fast) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int m = 0; m < QPL; ++m) {
                const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
                const lds_f4 mm = *reinterpret_cast<const lds_f4 *>(mb + 4 * CG * m);
                const lds_f4 gj = *reinterpret_cast<const lds_f4 *>(gb + 4 * CG * m);
#define LTR_PAIR(e, acc)                                                            \
    {                                                                               \
        const float r = ltr_rcp(ui + u[e]);                                         \
        const float ckj = u[e] * r, cjk = ui * r;                                   \
        const float term = (cjk >= eps ? gj[e] : 0.f) - (ckj >= eps ? gk : 0.f);    \
        acc = fmaf(mm[e] * (ckj * cjk), term, acc);                                 \
    }
                LTR_PAIR(0, a0) LTR_PAIR(1, a1) LTR_PAIR(2, a2) LTR_PAIR(3, a3)
#undef LTR_PAIR
            }
            a = (a0 + a1) + (a2 + a3);          // j == k contributes exactly 0 (term = g_k - g_k)
        } else {
            const float sk = sc[i];
            for (int m = 0; m < QPL; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * (cg + CG * m) + e;
                    const float ex = __expf(alpha * (sk - sc[j]));
                    const float ckj = ltr_rcp(1.f + ex);                  // sigmoid(-alpha (s_k - s_j))
                    const float cjk = (ex < 1e30f) ? ex * ckj : 1.f;      // sigmoid(-alpha (s_j - s_k))
                    const float term = (cjk >= eps ? gg[j] : 0.f) - (ckj >= eps ? gk : 0.f);
                    a += (j != i && mk[j] != 0.f) ? ckj * cjk * term : 0.f;
                }
        }
        a += LTR_DPP(a, LTR_DPP_XOR1);
        if (CG == 4) a += LTR_DPP(a, LTR_DPP_XOR2);
        if (cg == 0) store(i, vi ? alpha * gscale * inv_idcg * a : 0.f);
        stamp(14);
    };
    // publish the wave partials, ONE barrier (it also publishes gg), every thread adds the waves in fixed order
    auto exchange = [&](float &loss_sum, float &idcg_sum) {
        if (lane == 0) {
            red[wig] = loss_acc;
            red[NW + wig] = idcg_acc;
        }
        __syncthreads();
        loss_sum = 0.f;
        idcg_sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            loss_sum += red[w];
            idcg_sum += red[NW + w];
        }
        stamp(13);
    };
    // The path branch is WAVE-uniform (ballots); each path runs start to end inside its own branch, with its own __syncthreads():
    // waves of different slates of one workgroup may take different branches -- every wave still arrives at exactly one s_barrier
    // per call.  Keeping the paths apart is what lets the common path hold its pair reciprocals in registers across the barrier
    // without a three-way merge of everything else that is live (the shared-sweep form spilled 384 B/lane in the fcw kernel).
    float ls, is;
    if (ultra) {
        ultra_first();
        exchange(ls, is);
        ultra_second(ls);
    } else {
        general_first();
        exchange(ls, is);
        general_second(ls, is);
    }
    return -total;
}

// ------------------------------------------------------------------------------------------------
// The same loss with ONE barrier site and shared sweeps (the path decides inside each half): the form the 136-wide pipeline
// kernels allocate registers best with (profiles/r04_variant_ab.json: 0.614 of the fp32 MFMA peak against 0.594 for the
// same halves composed from lambdas, which is the same arithmetic).  Four pair terms share one v_rcp_f32.
template <int S, int NW, bool WRITE_SC, class Score, class Store, class Stamp = NoStamp>
__device__ __forceinline__ float approx_ndcg_fused_shared(int t, Score score, float *sc, const float *yl, const float *gn, float *gg,
                                                          float *uu, float *um, float *mk, float *red, const float *invd, float alpha,
                                                          float eps, float gscale, Store store, Stamp stamp = Stamp()) {
    static_assert(S == 32 || S == 64 || S == 128, "fused slates are 32, 64 or 128 documents");
    constexpr int CG = 64 * NW / S;                 // lanes per row
    static_assert(CG == 2 || CG == 4, "two or four lanes per document row");
    constexpr int RPW = 64 / CG;                    // rows per wave
    constexpr int QPL = S / 4 / CG;                 // column quads per lane
    static_assert(QPL % 2 == 0, "two accumulator chains");
    constexpr int SLOTS = (S + 63) / 64;            // documents per lane in the prologue
    const int lane = t & 63;
    const int wig = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cg = lane & (CG - 1);
    const int i = wig * RPW + lane / CG;            // this lane's row
    const bool ultra_ok = eps <= 1e-7f;

    // ---- prologue (wave-private)
    float sv[SLOTS], yc[SLOTS];
    bool real[SLOTS];
    bool bad = false, noultra = false;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int j = lane + 64 * sl;
        const bool act = (S % 64 == 0) || j < S;
        sv[sl] = act ? score(act ? j : 0) : 0.f;
        real[sl] = act && gn[act ? j : 0] >= 0.f;
        yc[sl] = act ? fmaxf(yl[act ? j : 0], 0.f) : 0.f;
    }
    const float sref = alpha * lane_bcast(sv[0], 0);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int j = lane + 64 * sl;
        const bool act = (S % 64 == 0) || j < S;
        const float x = real[sl] ? alpha * sv[sl] - sref : 0.f;
        bad = bad || !(fabsf(x) <= 69.f);            // NaN scores take the slow path too
        noultra = noultra || !(fabsf(x) <= 8.f);
        const float u = real[sl] ? expf(x) : 1.f;
        const bool isint = yc[sl] <= 15.f && yc[sl] == floorf(yc[sl]);
        noultra = noultra || (real[sl] && !isint);
        if (act) {
            uu[j] = u;
            um[j] = real[sl] ? u : 0.f;
            mk[j] = real[sl] ? 1.f : 0.f;
            if (WRITE_SC) sc[j] = sv[sl];
        }
    }
    const bool fast = __ballot(bad) == 0ull;
    const bool ultra = ultra_ok && fast && __ballot(noultra) == 0ull;
    float idcg_own = 0.f;
    if (ultra) {
        // ideal DCG from per-grade counts (ideal_dcg_by_counts): wave-uniform, one LDS read
        int yi[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) yi[sl] = real[sl] ? (int)yc[sl] : 0;
        idcg_own = ideal_dcg_by_counts<SLOTS>(yi, invd + 128, lane);
    }
    stamp(10);
    stamp(11);

    // ---- sweep 1: soft rank of row i (and, off the no-clamp path, its label rank by counting)
    const float *ub = uu + 4 * cg, *nb = um + 4 * cg, *mb = mk + 4 * cg, *gb = gg + 4 * cg;
    const bool mine = ((i >> 2) & (CG - 1)) == cg;          // the j == i column is one of this lane's
    const bool vi = gn[i] >= 0.f;
    const float ui = uu[i];
    float p = 0.f, cnt = 0.f;
    if (ultra) {
        auto quad = [&](int m, float acc) {
            const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
            const lds_f4 n = *reinterpret_cast<const lds_f4 *>(nb + 4 * CG * m);
            const float d0 = ui + u[0], d1 = ui + u[1], d2 = ui + u[2], d3 = ui + u[3];
            const float D01 = d0 * d1, D23 = d2 * d3;
            const float N01 = fmaf(n[0], d1, n[1] * d0), N23 = fmaf(n[2], d3, n[3] * d2);
            return fmaf(fmaf(N01, D23, N23 * D01), ltr_rcp(D01 * D23), acc);
        };
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int m = 0; m < QPL; m += 2) {
            acc0 = quad(m, acc0);
            acc1 = quad(m + 1, acc1);
        }
        p = acc0 + acc1;
        if (mine) p -= 0.5f;                                     // the j == i term: u_i / (2 u_i)
    } else {
        const float yi = yl[i];
        {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            const float *yb = yl + 4 * cg;
#pragma unroll 2
            for (int m = 0; m < QPL; ++m) {
                const int j = 4 * (cg + CG * m);
                const lds_f4 y = *reinterpret_cast<const lds_f4 *>(yb + 4 * CG * m);
                c0 += ((y[0] > yi) || (y[0] == yi && j + 0 < i)) ? 1.f : 0.f;
                c1 += ((y[1] > yi) || (y[1] == yi && j + 1 < i)) ? 1.f : 0.f;
                c2 += ((y[2] > yi) || (y[2] == yi && j + 2 < i)) ? 1.f : 0.f;
                c3 += ((y[3] > yi) || (y[3] == yi && j + 3 < i)) ? 1.f : 0.f;
            }
            cnt = (c0 + c1) + (c2 + c3);
        }
        if (fast) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll 2
            for (int m = 0; m < QPL; ++m) {
                const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
                const lds_f4 mm = *reinterpret_cast<const lds_f4 *>(mb + 4 * CG * m);
                p0 = fmaf(mm[0], fmaxf(u[0] * ltr_rcp(ui + u[0]), eps), p0);
                p1 = fmaf(mm[1], fmaxf(u[1] * ltr_rcp(ui + u[1]), eps), p1);
                p2 = fmaf(mm[2], fmaxf(u[2] * ltr_rcp(ui + u[2]), eps), p2);
                p3 = fmaf(mm[3], fmaxf(u[3] * ltr_rcp(ui + u[3]), eps), p3);
            }
            p = (p0 + p1) + (p2 + p3);
            if (mine) p -= fmaxf(ui * ltr_rcp(ui + ui), eps);    // the j == i term
        } else {
            const float si = sc[i];
            for (int m = 0; m < QPL; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * (cg + CG * m) + e;
                    const float ex = __expf(alpha * (si - sc[j]));
                    const float c = fmaxf(ltr_rcp(1.f + ex), eps);
                    p += (j != i && mk[j] != 0.f) ? c : 0.f;
                }
        }
    }
    // the CG partials of a row sit in adjacent lanes
    p += LTR_DPP(p, LTR_DPP_XOR1);
    if (CG == 4) p += LTR_DPP(p, LTR_DPP_XOR2);
    float idcg_acc = 0.f, loss_acc = 0.f, gout = 0.f;
    const float pos = 1.f + p;
    if (ultra) {
        if (vi && cg == 0) {
            const float iL = ltr_rcp(__log2f(1.f + pos));
            const float gl = gn[i] * iL;
            loss_acc = gl;
            gout = gl * iL * ltr_rcp((1.f + pos) * LTR_LN2);     // d(-sum gain/L)/d pos_i, not yet / maxDCG
        }
    } else {
        cnt += LTR_DPP(cnt, LTR_DPP_XOR1);
        if (CG == 4) cnt += LTR_DPP(cnt, LTR_DPP_XOR2);
        if (vi && cg == 0) {
            const float gain = gn[i];
            const float L = log2f(1.f + pos);
            if (gain > 0.f) idcg_acc = gain / log2f(2.f + cnt);
            loss_acc = gain / L;
            gout = gain / (L * L * (1.f + pos) * LTR_LN2);
        }
    }
    if (cg == 0) gg[i] = gout;                                   // 0 for padded documents
    stamp(12);
    loss_acc = wave_allsum(loss_acc);
    if (!ultra) idcg_acc = wave_allsum(idcg_acc);
    if (lane == 0) {
        red[wig] = loss_acc;
        red[NW + wig] = idcg_acc;
    }
    __syncthreads();                                             // gg and the wave partials are out
    float loss_sum = 0.f, idcg_sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        loss_sum += red[w];
        idcg_sum += red[NW + w];
    }
    stamp(13);
    const float inv_idcg = 1.f / fmaxf(ultra ? idcg_own : idcg_sum, eps);   // maxDCG clamp (:43)
    const float total = loss_sum * inv_idcg;

    // ---- sweep 2: d loss / d s_k = alpha * sum_j t_kj (g_j [c_jk >= eps] - g_k [c_kj >= eps]),  t = c_kj c_jk   (k = i)
    const float kscale = alpha * gscale * inv_idcg;
    const float gk = gg[i];
    float a = 0.f;
    if (ultra) {
        auto quad = [&](int m, float acc) {
            const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
            const lds_f4 n = *reinterpret_cast<const lds_f4 *>(nb + 4 * CG * m);
            const lds_f4 gj = *reinterpret_cast<const lds_f4 *>(gb + 4 * CG * m);
            float q0 = ui + u[0], q1 = ui + u[1], q2 = ui + u[2], q3 = ui + u[3];
            q0 *= q0, q1 *= q1, q2 *= q2, q3 *= q3;
            const float t0 = n[0] * (gj[0] - gk), t1 = n[1] * (gj[1] - gk);                     // exactly 0 at j == k
            const float t2 = n[2] * (gj[2] - gk), t3 = n[3] * (gj[3] - gk);
            const float Q01 = q0 * q1, Q23 = q2 * q3;
            const float N01 = fmaf(t0, q1, t1 * q0), N23 = fmaf(t2, q3, t3 * q2);
            return fmaf(fmaf(N01, Q23, N23 * Q01), ltr_rcp(Q01 * Q23), acc);
        };
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int m = 0; m < QPL; m += 2) {
            acc0 = quad(m, acc0);
            acc1 = quad(m + 1, acc1);
        }
        a = ui * (acc0 + acc1);
    } else if (fast) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
        for (int m = 0; m < QPL; ++m) {
            const lds_f4 u = *reinterpret_cast<const lds_f4 *>(ub + 4 * CG * m);
            const lds_f4 mm = *reinterpret_cast<const lds_f4 *>(mb + 4 * CG * m);
            const lds_f4 gj = *reinterpret_cast<const lds_f4 *>(gb + 4 * CG * m);
#define LTR_PAIR(e, acc)                                                            \
    {                                                                               \
        const float r = ltr_rcp(ui + u[e]);                                         \
        const float ckj = u[e] * r, cjk = ui * r;                                   \
        const float term = (cjk >= eps ? gj[e] : 0.f) - (ckj >= eps ? gk : 0.f);    \
        acc = fmaf(mm[e] * (ckj * cjk), term, acc);                                 \
    }
            LTR_PAIR(0, a0) LTR_PAIR(1, a1) LTR_PAIR(2, a2) LTR_PAIR(3, a3)
#undef LTR_PAIR
        }
        a = (a0 + a1) + (a2 + a3);              // j == k contributes exactly 0 (term = g_k - g_k)
    } else {
        const float sk = sc[i];
        for (int m = 0; m < QPL; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * (cg + CG * m) + e;
                const float ex = __expf(alpha * (sk - sc[j]));
                const float ckj = ltr_rcp(1.f + ex);                  // sigmoid(-alpha (s_k - s_j))
                const float cjk = (ex < 1e30f) ? ex * ckj : 1.f;      // sigmoid(-alpha (s_j - s_k))
                const float term = (cjk >= eps ? gg[j] : 0.f) - (ckj >= eps ? gk : 0.f);
                a += (j != i && mk[j] != 0.f) ? ckj * cjk * term : 0.f;
            }
    }
    a += LTR_DPP(a, LTR_DPP_XOR1);
    if (CG == 4) a += LTR_DPP(a, LTR_DPP_XOR2);
    if (cg == 0) store(i, vi ? kscale * a : 0.f);
    stamp(14);
    return -total;
}

// ------------------------------------------------------------------------------------------------
// ListNet (losses/listnet.py:5-16).  yt/yp: LDS [S] labels / scores (no padding concept in the reference).
// Returns -sum_i p_i log q_i (or the apply_sigmoid variant); store(i, gscale * dloss/dscore_i).
template <class Store>
__device__ __forceinline__ float listnet_slate(const SlateGroup &g, const float *yt, const float *yp,
                                               bool apply_sigmoid, float gscale, bool want_grad, Store store) {
    float my = -INFINITY, ms = -INFINITY;
    for (int j = g.t; j < g.S; j += g.group) {
        my = fmaxf(my, yt[j]);
        ms = fmaxf(ms, yp[j]);
    }
    my = group_max(g, my);
    ms = group_max(g, ms);
    float zy = 0.f, zs = 0.f;
    for (int j = g.t; j < g.S; j += g.group) {
        zy += expf(yt[j] - my);
        zs += expf(yp[j] - ms);
    }
    zy = group_sum(g, zy);
    zs = group_sum(g, zs);
    float loss = 0.f, wsum = 0.f;
    for (int j = g.t; j < g.S; j += g.group) {
        const float p = expf(yt[j] - my) / zy;
        const float q = expf(yp[j] - ms) / zs;
        const float plq = p * logf(q);                // log(softmax) taken literally (:16)
        if (apply_sigmoid) {
            const float r = 1.f / (1.f + expf(-plq));
            loss += r;
            wsum += r * (1.f - r) * p;
        } else {
            loss += plq;
            wsum += p;
        }
    }
    loss = group_sum(g, loss);
    if (!want_grad) return -loss;
    wsum = group_sum(g, wsum);
    for (int j = g.t; j < g.S; j += g.group) {
        const float p = expf(yt[j] - my) / zy;
        const float q = expf(yp[j] - ms) / zs;
        float w = p;
        if (apply_sigmoid) {
            const float r = 1.f / (1.f + expf(-p * logf(q)));
            w = r * (1.f - r) * p;
        }
        store(j, gscale * (q * wsum - w));
    }
    return -loss;
}

// ------------------------------------------------------------------------------------------------
// LambdaLoss (losses/lambdaL.py:7-127).
struct LambdaParams {
    int scheme;      // LTR_SCHEME_*
    int k;           // <= 0: no truncation
    float sigma, mu, eps;
    float log_scale;  // 1/ln2 for binary, 1 for natural: log_b(x) = ln(x) * log_scale
    float log_floor;  // log_b(eps)
};

struct LambdaLds {
    float *sc;     // [S] scores
    float *yl;     // [S] labels, -inf if padded
    float *gn;     // [S] gains -> G (padded -1)
    float *w1;     // [S] G_i / D_{r_i}                  (ndcgLoss1)
    float *invd;   // [S] 1 / D_{r_i},  D_r = log2(2 + r)
    float *delta;  // [S] delta_m = |1/D_{m-1} - 1/D_m|, delta_0 = 0      (ndcgLoss2)
    int *rk;       // [S] 0-based predicted rank of document i
};

// Rank by counting on labels (ideal DCG) and on scores (predicted rank; padded documents rank last), G, and the
// per-document weight inputs -- ONE sweep over contiguous column blocks with 16-byte LDS broadcast reads.
// All LambdaLds arrays have s_al = (S+3)&~3 entries; on entry sc/yl/gn hold [0,S).  L.delta doubles as scratch
// for the masked scores during the sweep and receives the delta table afterwards.
__device__ __forceinline__ void lambda_prepare(const SlateGroup &g, const LambdaLds &L, const LambdaParams &P) {
    const int s_al = (g.S + 3) & ~3;
    const int jb = (((g.S + g.CG - 1) / g.CG) + 3) & ~3;
    const int j0 = g.cg * jb;
    const int j1 = min(j0 + jb, s_al);
    float *sm = L.delta;
    for (int j = g.t; j < s_al; j += g.group) {
        const bool real = j < g.S && L.gn[j] >= 0.f;
        sm[j] = real ? L.sc[j] : -INFINITY;
        if (j >= g.S) {
            L.sc[j] = 0.f;
            L.yl[j] = -INFINITY;
            L.gn[j] = -1.f;
            L.invd[j] = 0.f;
            L.w1[j] = 0.f;
            L.rk[j] = g.S;
        }
    }
    __syncthreads();
    float idcg_acc = 0.f;
    for (int i0 = 0; i0 < g.S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < g.S;
        float cl = 0.f, cs = 0.f;
        if (row) {
            const float yi = L.yl[i], si = sm[i];
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll 2
            for (int j = j0; j < j1; j += 4) {
                const lds_f4 y = *reinterpret_cast<const lds_f4 *>(L.yl + j);
                const lds_f4 v = *reinterpret_cast<const lds_f4 *>(sm + j);
                a0 += ((y[0] > yi) || (y[0] == yi && j + 0 < i)) ? 1.f : 0.f;
                a1 += ((y[1] > yi) || (y[1] == yi && j + 1 < i)) ? 1.f : 0.f;
                a0 += ((y[2] > yi) || (y[2] == yi && j + 2 < i)) ? 1.f : 0.f;
                a1 += ((y[3] > yi) || (y[3] == yi && j + 3 < i)) ? 1.f : 0.f;
                b0 += ((v[0] > si) || (v[0] == si && j + 0 < i)) ? 1.f : 0.f;
                b1 += ((v[1] > si) || (v[1] == si && j + 1 < i)) ? 1.f : 0.f;
                b0 += ((v[2] > si) || (v[2] == si && j + 2 < i)) ? 1.f : 0.f;
                b1 += ((v[3] > si) || (v[3] == si && j + 3 < i)) ? 1.f : 0.f;
            }
            cl = a0 + a1;
            cs = b0 + b1;
        }
        const float rl = row_reduce(g, cl);
        const float rs = row_reduce(g, cs);
        if (row && g.cg == 0) {
            const float gi = L.gn[i];
            if (gi > 0.f && (P.k <= 0 || rl < (float)P.k)) idcg_acc += gi / log2f(2.f + rl);
            L.rk[i] = (int)rs;
            L.invd[i] = 1.f / log2f(2.f + rs);
        }
    }
    idcg_acc = wave_allsum(idcg_acc);
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) g.red[g.wig] = idcg_acc;
    __syncthreads();          // sweep done with sm (= L.delta); rk / invd / wave partials published
    float idcg = 0.f;
    for (int w = 0; w < g.nw; ++w) idcg += g.red[w];
    idcg = fmaxf(idcg, P.eps);
    for (int j = g.t; j < s_al; j += g.group) {
        if (j < g.S) {
            const float v = L.gn[j];
            const float G = v < 0.f ? -1.f : v / idcg;
            L.gn[j] = G;
            L.w1[j] = fmaxf(G, 0.f) / log2f(2.f + (float)L.rk[j]);      // G / D as the reference divides (:97)
        }
        // D[m] = log2(m + 2);  delta_m = |1/D[m-1] - 1/D[m]|  (lambdaL.py:100-104)
        L.delta[j] = j == 0 ? 0.f : fabsf(1.f / log2f((float)j + 1.f) - 1.f / log2f((float)j + 2.f));
    }
    __syncthreads();
}

// Pair weight w for the pair whose FIRST element has (rank r1, 1/D inv1, G/D w1f, gain G1, clamped label y1) and
// whose second has (r2, inv2, G2, y2), lambdaL.py:96-127.  G is 0 for padded documents.
// SCH >= 0: scheme fixed at compile time (standalone kernels); SCH < 0: taken from P.scheme at run time
// (slate-uniform branches; used inside the fused pipeline kernel, which is too large to instantiate 8 times).
template <int SCH>
__device__ __forceinline__ float lambda_weight(const LambdaParams &P, const float *delta, int r1, int r2, float inv1,
                                               float inv2, float w1f, float G1, float G2, float y1, float y2) {
    const int sch = SCH < 0 ? P.scheme : SCH;
    if (sch == 0 || sch == 5) return 1.f;
    if (sch == 1) return w1f;
    if (sch == 6) return fabsf(y1 - y2);
    if (sch == 7) return fabsf(y1 * y1 - y2 * y2);
    const float dG = fabsf(G1 - G2);
    float w = 0.f;
    if (sch == 2 || sch == 4) {
        int m = r1 - r2;
        m = m < 0 ? -m : m;
        w = delta[m] * dG;
        if (sch == 4) w *= P.mu;
    }
    if (sch == 3 || sch == 4) w += fabsf(inv1 - inv2) * dG;
    return w;
}

// log_b(clamp(clamp(u, eps)^w, eps)) and d/dx of it w.r.t. x = sigma * d (times sigma later).
// u = sigmoid(x), um = 1 - u = sigmoid(-x).  live: no clamp active (torch clamp passes grad iff x >= min).
__device__ __forceinline__ void lambda_pair_term(const LambdaParams &P, float w, float u, float um, float &ell,
                                                 float &dldx) {
    const float lg = __logf(fmaxf(u, P.eps)) * P.log_scale;
    const float wl = w * lg;
    ell = fmaxf(wl, P.log_floor);
    const bool live = (u >= P.eps) && (wl >= P.log_floor);
    dldx = live ? w * um * P.log_scale : 0.f;
}

// Returns (to every thread) the slate loss -sum_kept ell and writes the kept-pair count to *count_out
// (every thread gets it); store(i, gscale * dloss/ds_i).
//
// Row i sweeps its contiguous column block.  Except for ndcgLoss1 every scheme's weight is symmetric and a pair
// is kept in exactly ONE orientation (higher label first), so each (i, j) needs one sigmoid pair, one log and
// one weight: thread i takes the loss term when it owns the first element and the gradient either way.
template <int SCH, class Store>
__device__ __forceinline__ float lambda_slate(const SlateGroup &g, const LambdaLds &L, const LambdaParams &P,
                                              float gscale, bool want_grad, float *count_out, Store store) {
    lambda_prepare(g, L, P);
    const int s_al = (g.S + 3) & ~3;
    const int jb = (((g.S + g.CG - 1) / g.CG) + 3) & ~3;
    const int j0 = g.cg * jb;
    const int j1 = min(j0 + jb, s_al);
    const bool all_pairs = (SCH < 0 ? P.scheme : SCH) == 1;
    typedef int lds_i4 __attribute__((ext_vector_type(4)));
    float lossacc = 0.f, cntacc = 0.f;
    for (int i0 = 0; i0 < g.S; i0 += g.sp) {
        const int i = i0 + g.ri;
        const bool row = i < g.S;
        const bool vi = row && L.gn[i] >= 0.f;
        float ls = 0.f, cn = 0.f, gr = 0.f;
        if (vi) {
            const float si = L.sc[i], yi = L.yl[i], Gi = L.gn[i], yci = fmaxf(yi, 0.f);
            const float invi = L.invd[i], w1i = L.w1[i];
            const int ri = L.rk[i];
            const bool ki = P.k <= 0 || ri < P.k;
            for (int j4 = j0; j4 < j1; j4 += 4) {
                const lds_f4 vs = *reinterpret_cast<const lds_f4 *>(L.sc + j4);
                const lds_f4 vy = *reinterpret_cast<const lds_f4 *>(L.yl + j4);
                const lds_f4 vg = *reinterpret_cast<const lds_f4 *>(L.gn + j4);
                const lds_f4 vd = *reinterpret_cast<const lds_f4 *>(L.invd + j4);
                const lds_f4 vw = *reinterpret_cast<const lds_f4 *>(L.w1 + j4);
                const lds_i4 vr = *reinterpret_cast<const lds_i4 *>(L.rk + j4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = j4 + e;
                    const float Gj = vg[e], yj = vy[e];
                    const bool ok = ki && Gj >= 0.f && (P.k <= 0 || vr[e] < P.k);
                    const float draw = si - vs[e];
                    const float dcl = fminf(fmaxf(draw, -1e8f), 1e8f);
                    const bool dlive = fabsf(draw) <= 1e8f;
                    float u, um;
                    sigmoid_pair(P.sigma * dcl, u, um);
                    const float ycj = fmaxf(yj, 0.f), Gjp = fmaxf(Gj, 0.f);
                    if (!all_pairs) {
                        const bool hi = yi > yj;                       // i is the first element of the kept pair
                        const bool kept = ok && yi != yj;
                        const float w = lambda_weight<SCH>(P, L.delta, ri, vr[e], invi, vd[e], 0.f, Gi, Gjp, yci, ycj);
                        float ell, dl;
                        lambda_pair_term(P, w, hi ? u : um, hi ? um : u, ell, dl);
                        ls += (kept && hi) ? ell : 0.f;
                        cn += (kept && hi) ? 1.f : 0.f;
                        gr += (kept && dlive) ? (hi ? -dl : dl) : 0.f;
                    } else {
                        // ndcgLoss1 keeps every valid pair incl. the diagonal, in both orientations (:26-27)
                        float ell, dl;
                        if (ok) {
                            lambda_pair_term(P, w1i, u, um, ell, dl);                 // pair (i, j)
                            ls += ell;
                            cn += 1.f;
                            gr -= (dlive && j != i) ? dl : 0.f;
                        }
                        if (ok && j != i) {
                            lambda_pair_term(P, vw[e], um, u, ell, dl);               // pair (j, i): gradient only
                            gr += dlive ? dl : 0.f;
                        }
                    }
                }
            }
        }
        lossacc += row_reduce(g, ls);     // every cg replica adds the same row total ...
        cntacc += row_reduce(g, cn);
        if (want_grad) {
            const float tot = row_reduce(g, gr);
            if (row && g.cg == 0) store(i, vi ? gscale * P.sigma * tot : 0.f);
        }
    }
    // ... so count each row once: only column group 0 contributes to the group total.
    const float total = group_sum(g, g.cg == 0 ? lossacc : 0.f);
    *count_out = group_sum(g, g.cg == 0 ? cntacc : 0.f);
    return -total;
}

// ------------------------------------------------------------------------------------------------
// LambdaLoss for long slates (S >= 256), RANK SPACE + every unordered pair ONCE.
//
// The slate is first permuted into predicted-rank order (the order the reference sorts into, lambdaL.py:17-21): row r
// of the pair matrix is then "the document ranked r", so the rank-dependent inputs of the weighing schemes stop being
// data -- 1/D_r is a table indexed by the lane's own row, delta_{|ri - rj|} is delta[|r - c|] with CONSECUTIVE lanes
// reading CONSECUTIVE entries (conflict-free, against a random per-lane gather and 55 % LDS bank-conflict cycles in
// the document-order sweep), the top-k mask is r < k && c < k.  Every scheme but ndcgLoss1 keeps a pair in exactly one
// orientation with a symmetric weight, and the gradient of a pair is antisymmetric (+g on one document, -g on the
// other): the S x S matrix is cut into 64 x 64 blocks, wave v owns block-row v, and in round t it evaluates block
// (v, v + t mod nb) -- each unordered block pair once (t = 0: the upper triangle of the diagonal block; t = nb/2 for
// even nb: only v < nb/2).  Row sums stay in the lane's registers; the column side of block (v, u) is reduced across
// the wave per column (DPP butterfly, ~10 VALU per 64 pairs) and added to colacc[] of block u -- in one round every
// wave writes a different column block and rounds are separated by a barrier, so the result is bit-reproducible.
// 40 VALU per unordered pair against 2 x 41 in the document-order sweep.
//
// Group layout: one wave per 64 ranks (blockDim = 64 * nb, nb = ceil(S / 64) <= 16).  LDS (floats, s64 = 64 nb each):
//   L.* as for lambda_prepare, plus rs / ry / rg (score, clamped label -- -1 for padded --, G by rank), colacc, doc_of.
struct LambdaRankLds {
    float *rs, *ry, *rg, *colacc;
    int *doc_of;
};

template <int SCH, class Store>
__device__ __forceinline__ float lambda_slate_blocked(const SlateGroup &g, const LambdaLds &L, const LambdaRankLds &R,
                                                      const LambdaParams &P, float gscale, bool want_grad,
                                                      float *count_out, Store store) {
    static_assert(SCH != 1, "ndcgLoss1 keeps both orientations of every pair: served by lambda_slate");
    lambda_prepare(g, L, P);                       // ranks by counting, G, delta table (ends with a barrier)
    const int S = g.S;
    const int nb = (S + 63) >> 6, s64 = nb << 6;
    const int lane = threadIdx.x & 63, v = g.t >> 6;          // wave v of the group owns ranks 64 v .. 64 v + 63
    // ---- permute into rank order (padded documents rank last; ranks >= S are filler)
    float *invr = L.w1;                            // 1 / D_r by RANK (w1 itself is only used by ndcgLoss1)
    for (int r = g.t; r < s64; r += g.group) {
        R.rg[r] = -1.f;
        R.rs[r] = 0.f;
        R.ry[r] = 0.f;
        R.colacc[r] = 0.f;
        R.doc_of[r] = -1;
        if (r < ((S + 3) & ~3)) invr[r] = 1.f / log2f(2.f + (float)r);
    }
    __syncthreads();
    for (int i = g.t; i < S; i += g.group) {
        const int r = L.rk[i];
        R.rs[r] = L.sc[i];
        R.rg[r] = L.gn[i];                                     // G, -1 for padded documents
        R.ry[r] = L.yl[i];                                     // raw label (-inf if padded): pairs compare raw labels (:24-27)
        R.doc_of[r] = i;
    }
    __syncthreads();
    const int row = 64 * v + lane;                             // this lane's rank
    const float si = R.rs[row], Gi = R.rg[row], yi = R.ry[row];
    const bool vi = Gi >= 0.f && (P.k <= 0 || row < P.k);
    const float invi = row < S ? invr[row] : 0.f;
    float ls = 0.f, cn = 0.f, gr = 0.f;
    for (int t = 0; 2 * t <= nb; ++t) {
        const int u = v + t < nb ? v + t : v + t - nb;         // column block of this round
        const bool active = (2 * t < nb) || (v < nb / 2);      // t == nb/2 (even nb): each block pair only once
        float colreg = 0.f;                                    // lane c holds the column-side sum of column 64 u + c
        if (active) {
            for (int c0 = 0; c0 < 64; ++c0) {
                const int col = 64 * u + c0;
                const float sj = R.rs[col], Gj = R.rg[col], yj = R.ry[col];      // broadcast reads
                const bool vj = Gj >= 0.f && (P.k <= 0 || col < P.k);
                const bool pair = vi && vj && yi != yj && (t > 0 || c0 > lane);   // diagonal block: upper triangle only
                const float draw = si - sj;
                const float dcl = fminf(fmaxf(draw, -1e8f), 1e8f);
                const bool dlive = fabsf(draw) <= 1e8f;
                float uu, um;
                sigmoid_pair(P.sigma * dcl, uu, um);
                const bool hi = yi > yj;                                          // this lane's document is the first of the kept pair
                const float invj = col < S ? invr[col] : 0.f;                     // broadcast read
                const float w = lambda_weight<SCH>(P, L.delta, row, col, invi, invj, 0.f, fmaxf(Gi, 0.f), fmaxf(Gj, 0.f),
                                                   fmaxf(yi, 0.f), fmaxf(yj, 0.f));
                float ell, dl;
                lambda_pair_term(P, w, hi ? uu : um, hi ? um : uu, ell, dl);
                ls += pair ? ell : 0.f;
                cn += pair ? 1.f : 0.f;
                const float gi = (pair && dlive) ? (hi ? -dl : dl) : 0.f;         // d loss-sum / d s_row; the column document gets -gi
                gr += gi;
                if (want_grad) {
                    const float csum = wave_allsum(-gi);
                    colreg = lane == c0 ? csum : colreg;
                }
            }
            if (want_grad) R.colacc[64 * u + lane] += colreg;  // this round, this wave is the only writer of block u
        }
        __syncthreads();                                       // next round writes a different column block per wave
    }
    const float total = group_sum(g, ls);
    *count_out = group_sum(g, cn);
    if (want_grad && row < s64) {
        const int doc = R.doc_of[row];
        if (doc >= 0) store(doc, gscale * P.sigma * (gr + R.colacc[row]));
    }
    return -total;
}

}  // namespace ltr
