// ltr_risk.hip -- risk-sensitive evaluation functions on a [queries x systems] effectiveness matrix (gfx950).
//
// Replaces, behind the reference's Python surface:
//   losses/riskLosses/riskFunctions.py:4-22   zRisk(mat, alpha, requires_grad, i)
//   losses/riskLosses/riskFunctions.py:25-33  geoRisk(mat, alpha, requires_grad, i)
//   losses/riskLosses/riskLosses.py:247-291   the tRisk tail (mean / std of the alpha-weighted deltas)
// The matrix is tiny next to the slate tensors (Q x <= ~10 systems): these are latency-bound reductions, so
// each runs as ONE workgroup that sweeps the matrix three times (column/row sums -> standardised residuals ->
// gradient) with fp64 accumulators and fixed-order cross-wave sums (bit-reproducible).  Forward value and the
// analytic gradient w.r.t. every matrix entry come out of the same launch.
#include "../../include/ltr_mi355x.h"
#include "ltr_device.h"
#include <math.h>

using namespace ltr;

namespace {

constexpr int kRiskThreads = 1024;

__device__ __forceinline__ double wave_allsum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, LTR_WAVE);
    return v;   // butterfly: every lane holds the same bits
}

// sum over the block; every thread gets the total (fixed order: lanes by butterfly, waves sequentially)
__device__ __forceinline__ double block_sum_f64(double v, double *red) {
    v = wave_allsum_f64(v);
    __syncthreads();
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) red[threadIdx.x / LTR_WAVE] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < kRiskThreads / LTR_WAVE; ++w) s += red[w];
    return s;
}

// zRisk:  si = sum_q mat[q,i];  t_q = sum_j mat[q,j];  n = sum_q t_q;  e_q = si t_q / n;
//         d_q = (mat[q,i] - e_q) / sqrt(e_q);  Z = sum_q d_q (1 + alpha [d_q < 0])                 (:5-22)
// geoRisk: sqrt((si / Q) * Phi(Z / Q)), Phi = standard normal cdf                                   (:26-33)
// dmat (optional): d value / d mat[q,j] (closed form; the indicator [d_q < 0] carries no gradient, like the
// reference's boolean mask).
__global__ void __launch_bounds__(kRiskThreads)
risk_kernel(const float *__restrict__ mat, int Q, int n, int col, float alpha, int geo, int guard,
            float *__restrict__ value, float *__restrict__ dmat) {
    __shared__ double red[kRiskThreads / LTR_WAVE];
    const int tid = threadIdx.x;
    double s_i = 0.0, s_n = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float *row = mat + (size_t)q * n;
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += (double)row[j];
        s_n += t;
        s_i += (double)row[col];
    }
    const double si = block_sum_f64(s_i, red);
    const double nn = block_sum_f64(s_n, red);
    double z_acc = 0.0, t1_acc = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float *row = mat + (size_t)q * n;
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += (double)row[j];
        const double x = (double)row[col];
        const double e = si * (t / nn);
        const bool dead = guard && e == 0.0;                       // numpy metric: e == 0 contributes 0 (metrics.py:30-33)
        const double d = dead ? 0.0 : (x - e) / sqrt(e);
        const double c = (guard ? (x - e < 0.0) : (d < 0.0)) ? 1.0 + (double)alpha : 1.0;
        z_acc += c * d;
        const double A = dead ? 0.0 : c * (-0.5 * (x + e) / (e * sqrt(e)));     // c_q * d d_q / d e_q
        t1_acc += A * t;
    }
    const double Z = block_sum_f64(z_acc, red);
    const double T1 = block_sum_f64(t1_acc, red) / nn;              // sum_r A_r t_r / n
    double val = Z, dZ = 1.0, dSi = 0.0;
    if (geo) {
        const double v = Z / (double)Q;
        const double Phi = 0.5 * erfc(-v * 0.70710678118654752440);
        const double phi = 0.39894228040143267794 * exp(-0.5 * v * v);
        const double M = si / (double)Q;
        val = sqrt(M * Phi);
        dZ = 0.5 / val * M * phi / (double)Q;
        dSi = 0.5 / val * Phi / (double)Q;
    }
    if (tid == 0) value[0] = (float)val;
    if (!dmat) return;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float *row = mat + (size_t)q * n;
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += (double)row[j];
        const double x = (double)row[col];
        const double e = si * (t / nn);
        const bool dead = guard && e == 0.0;
        const double rs = dead ? 0.0 : 1.0 / sqrt(e);
        const double d = (x - e) * rs;
        const double c = (guard ? (x - e < 0.0) : (d < 0.0)) ? 1.0 + (double)alpha : 1.0;
        const double A = dead ? 0.0 : c * (-0.5 * (x + e) * rs / e);
        const double common = A * si / nn - si * T1 / nn;          // via t_q and via n
        const double own = c * rs + T1;                             // via mat[q,i] itself and via si
        for (int j = 0; j < n; ++j) {
            double gz = common + (j == col ? own : 0.0);
            dmat[(size_t)q * n + j] = (float)(dZ * gz + (j == col ? dSi : 0.0));
        }
    }
}

// tRisk tail: delta_q = (a_q - b_q)(1 + alpha [a_q < b_q]);  value = mean(delta) / std(delta)  (unbiased std)
// da / db (optional): d value / d a_q, d value / d b_q.                       riskLosses.py:278-291 / :332-345
__global__ void __launch_bounds__(kRiskThreads)
trisk_kernel(const float *__restrict__ a, const float *__restrict__ b, int Q, float alpha, float *__restrict__ value,
             float *__restrict__ da, float *__restrict__ db) {
    __shared__ double red[kRiskThreads / LTR_WAVE];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const double x = (double)a[q] - (double)b[q];
        s += x * (a[q] < b[q] ? 1.0 + (double)alpha : 1.0);
    }
    const double mu = block_sum_f64(s, red) / (double)Q;
    double v = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const double x = ((double)a[q] - (double)b[q]) * (a[q] < b[q] ? 1.0 + (double)alpha : 1.0);
        v += (x - mu) * (x - mu);
    }
    const double var = block_sum_f64(v, red) / (double)(Q - 1);
    const double se = sqrt(var);
    if (tid == 0) value[0] = (float)(mu / se);
    if (!da && !db) return;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const double c = a[q] < b[q] ? 1.0 + (double)alpha : 1.0;
        const double x = ((double)a[q] - (double)b[q]) * c;
        const double g = (1.0 / ((double)Q * se) - mu * (x - mu) / ((double)(Q - 1) * se * var)) * c;
        if (da) da[q] = (float)g;
        if (db) db[q] = (float)(-g);
    }
}

// The TAIL of a tRisk loss in one launch (riskLosses.py:269-291 / :332-345) on the [Q][2] matrix (model, baseline): the flip of
// transformation 1 (mat' = -mat + max(mat), in the reference's fp32 arithmetic), the alpha-weighted deltas, mean / std, the
// `negative` factor; value and d value / d mat together.  The flip's whole-matrix maximum gets no gradient: the gradients of the
// two columns of a query are g and -g, so their sum over the matrix is zero.
__global__ void __launch_bounds__(kRiskThreads)
trisk_tail_kernel(const float *__restrict__ mat, int Q, float alpha, int flip, float factor, float *__restrict__ value,
                  float *__restrict__ dmat) {
    __shared__ double red[kRiskThreads / LTR_WAVE];
    const int tid = threadIdx.x;
    float M = 0.f;
    if (flip) {
        float mx = -INFINITY;
        for (int e = tid; e < 2 * Q; e += kRiskThreads) mx = fmaxf(mx, mat[e]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, LTR_WAVE));
        __syncthreads();
        if ((tid & (LTR_WAVE - 1)) == 0) red[tid / LTR_WAVE] = (double)mx;
        __syncthreads();
        M = (float)red[0];
        for (int w = 1; w < kRiskThreads / LTR_WAVE; ++w) M = fmaxf(M, (float)red[w]);
        __syncthreads();
    }
    auto at = [&](int q, int j) -> float { return flip ? -mat[2 * (size_t)q + j] + M : mat[2 * (size_t)q + j]; };
    double s = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float a = at(q, 0), b = at(q, 1);
        s += ((double)a - (double)b) * (a < b ? 1.0 + (double)alpha : 1.0);
    }
    const double mu = block_sum_f64(s, red) / (double)Q;
    double v = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float a = at(q, 0), b = at(q, 1);
        const double x = ((double)a - (double)b) * (a < b ? 1.0 + (double)alpha : 1.0);
        v += (x - mu) * (x - mu);
    }
    const double var = block_sum_f64(v, red) / (double)(Q - 1);
    const double se = sqrt(var);
    if (tid == 0) value[0] = (float)(mu / se) * factor;
    if (!dmat) return;
    for (int q = tid; q < Q; q += kRiskThreads) {
        const float a = at(q, 0), b = at(q, 1);
        const double c = a < b ? 1.0 + (double)alpha : 1.0;
        const double x = ((double)a - (double)b) * c;
        const float g = (float)((1.0 / ((double)Q * se) - mu * (x - mu) / ((double)(Q - 1) * se * var)) * c) * factor;
        dmat[2 * (size_t)q] = flip ? -g : g;
        dmat[2 * (size_t)q + 1] = flip ? g : -g;
    }
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The TAIL of a geoRisk / zRisk loss in one launch (riskLosses.py:47-60, :118-125, :170-180, :237-244): the "larger = better" flip
// mat' = -mat + max(mat), the risk of column 0 and (return strategies 2 / 3) of the last column, the strategy's combination, the
// `negative` factor -- value and d value / d mat (w.r.t. the UNFLIPPED matrix; the whole-matrix max passes its gradient to the
// maximal entries, evenly among ties, like torch.max) come out together.
//   strategy 1: f R0      2: f (R1 - R0)   [zquirk: f R1 - R0, the reference's operator precedence in zRiskListnetLoss, :176]
//   strategy 3: f (R1 - R0)^2
struct RiskCol { double si, Z, T1, val, dZ, dSi; };

__global__ void __launch_bounds__(kRiskThreads)
risk_tail_kernel(const float *__restrict__ mat, int Q, int n, float alpha, int geo, int strategy, int flip, float factor, int zquirk,
                 float *__restrict__ value, float *__restrict__ dmat) {
    __shared__ double red[kRiskThreads / LTR_WAVE];
    const int tid = threadIdx.x;
    double M = 0.0;
    if (flip) {
        double mx = -INFINITY;
        for (int e = tid; e < Q * n; e += kRiskThreads) mx = fmax(mx, (double)mat[e]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, LTR_WAVE));
        __syncthreads();
        if ((tid & (LTR_WAVE - 1)) == 0) red[tid / LTR_WAVE] = mx;
        __syncthreads();
        M = red[0];
        for (int w = 1; w < kRiskThreads / LTR_WAVE; ++w) M = fmax(M, red[w]);
        M = (double)(float)M;
    }
    // the flipped entry in the reference's fp32 arithmetic (-mat + max), then promoted
    auto at = [&](int q, int j) -> double { return flip ? (double)(-mat[(size_t)q * n + j] + (float)M) : (double)mat[(size_t)q * n + j]; };
    const int ncol = strategy == 1 ? 1 : 2;
    const int cols[2] = {0, n - 1};
    RiskCol rc[2];
    double s_n = 0.0, s_i0 = 0.0, s_i1 = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += at(q, j);
        s_n += t;
        s_i0 += at(q, 0);
        s_i1 += at(q, n - 1);
    }
    const double nn = block_sum_f64(s_n, red);
    rc[0].si = block_sum_f64(s_i0, red);
    rc[1].si = block_sum_f64(s_i1, red);
    for (int i = 0; i < ncol; ++i) {
        const double si = rc[i].si;
        double z_acc = 0.0, t1_acc = 0.0;
        for (int q = tid; q < Q; q += kRiskThreads) {
            double t = 0.0;
            for (int j = 0; j < n; ++j) t += at(q, j);
            const double x = at(q, cols[i]);
            const double e = si * (t / nn);
            const double d = (x - e) / sqrt(e);
            const double c = d < 0.0 ? 1.0 + (double)alpha : 1.0;
            z_acc += c * d;
            t1_acc += c * (-0.5 * (x + e) / (e * sqrt(e))) * t;
        }
        rc[i].Z = block_sum_f64(z_acc, red);
        rc[i].T1 = block_sum_f64(t1_acc, red) / nn;
        rc[i].val = rc[i].Z;
        rc[i].dZ = 1.0;
        rc[i].dSi = 0.0;
        if (geo) {
            const double v = rc[i].Z / (double)Q;
            const double Phi = 0.5 * erfc(-v * 0.70710678118654752440);
            const double phi = 0.39894228040143267794 * exp(-0.5 * v * v);
            const double Mq = si / (double)Q;
            rc[i].val = sqrt(Mq * Phi);
            rc[i].dZ = 0.5 / rc[i].val * Mq * phi / (double)Q;
            rc[i].dSi = 0.5 / rc[i].val * Phi / (double)Q;
        }
    }
    // the reference evaluates each risk in fp32 tensors: round the two values where it does before combining them
    const double R0 = (double)(float)rc[0].val, R1 = ncol > 1 ? (double)(float)rc[1].val : 0.0, f = (double)factor;
    double out, c0, c1 = 0.0;
    if (strategy == 1) { out = f * R0; c0 = f; }
    else if (strategy == 2) {
        if (zquirk) { out = f * R1 - R0; c0 = -1.0; c1 = f; }
        else { out = f * (R1 - R0); c0 = -f; c1 = f; }
    } else { const double df = R1 - R0; out = f * df * df; c0 = -2.0 * f * df; c1 = 2.0 * f * df; }
    if (tid == 0) value[0] = (float)out;
    if (!dmat) return;
    const double coef[2] = {c0, c1};
    double g_acc = 0.0;
    for (int q = tid; q < Q; q += kRiskThreads) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += at(q, j);
        double common = 0.0, own[2] = {0.0, 0.0};
        for (int i = 0; i < ncol; ++i) {
            const double si = rc[i].si, x = at(q, cols[i]);
            const double e = si * (t / nn), rs = 1.0 / sqrt(e), d = (x - e) * rs;
            const double c = d < 0.0 ? 1.0 + (double)alpha : 1.0;
            const double A = c * (-0.5 * (x + e) * rs / e);
            common += coef[i] * rc[i].dZ * (A * si / nn - si * rc[i].T1 / nn);
            own[i] = coef[i] * (rc[i].dZ * (c * rs + rc[i].T1) + rc[i].dSi);
        }
        for (int j = 0; j < n; ++j) {
            double gq = common;
            if (j == 0) gq += own[0];
            if (ncol > 1 && j == n - 1) gq += own[1];
            g_acc += gq;
            dmat[(size_t)q * n + j] = (float)(flip ? -gq : gq);
        }
    }
    if (flip) {      // d max(mat) / d mat: the sum of all d value / d mat' entries lands on the maximal entries
        const double G = block_sum_f64(g_acc, red);
        double ties_a = 0.0;
        for (int e = tid; e < Q * n; e += kRiskThreads) ties_a += (double)mat[e] == M ? 1.0 : 0.0;
        const double ties = block_sum_f64(ties_a, red);
        for (int e = tid; e < Q * n; e += kRiskThreads)
            if ((double)mat[e] == M) dmat[e] += (float)(G / ties);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The [queries x systems] effectiveness matrix of the risk-sensitive losses, one workgroup per query, one launch for all systems
// (riskLosses.py:8-49 / :128-169 / :247-276 Listnet type, :63-117 / :183-236 / :294-330 Lambda type) -- and d mat[q][0] / d (the
// model's input), the only gradient the losses need (column 0 is the model; baselines and the ideal ranking are constants).
//   mode 0 (Listnet type): ref = labels, x0 = the model's scores, rest = baseline scores [B][S][nr]; every vector is soft-maxed over
//     the slate first (:10-12);  lt 1: sum_j (t p - t^2)^2   lt 2: cos(t, p)   lt 3: (sum t p - sum t^2)^2        (:16-46)
//   mode 1 (Lambda type): ref / x0 / rest [nr][B][S] are the lambdaMask column sums, used as they are;
//     lt 1: sum_j (x - t)^2   lt 2: cos(t, x)   lt 3: (sum x - sum t)^2                                          (:71-104)
//   mode 2 (the tRisk Listnet loss, :247-276): as mode 0, but transformation 2 is the cosine of the PRODUCTS, cos(t^2, t p).
//   ideal != 0 appends the column of the reference vector itself.
// fp32 elementwise arithmetic, fp64 sums, fixed-order reductions.  The cosine follows ATen: w12 / sqrt(max(w1 w2, eps^2)), eps 1e-8.
constexpr int kMatThreads = 256;
constexpr int kMatMaxS = 2048;

__device__ __forceinline__ double mat_block_sum(double v, double *red) {
    v = wave_allsum_f64(v);
    __syncthreads();
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) red[threadIdx.x / LTR_WAVE] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < kMatThreads / LTR_WAVE; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float mat_block_max(float v, double *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, LTR_WAVE));
    __syncthreads();
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) red[threadIdx.x / LTR_WAVE] = (double)v;
    __syncthreads();
    float m = -INFINITY;
    for (int w = 0; w < kMatThreads / LTR_WAVE; ++w) m = fmaxf(m, (float)red[w]);
    return m;
}
// in place: v[0..S) -> softmax(v) (torch: exp(x - max) / sum)
__device__ __forceinline__ void mat_softmax(float *v, int S, double *red) {
    float m = -INFINITY;
    for (int j = threadIdx.x; j < S; j += kMatThreads) m = fmaxf(m, v[j]);
    m = mat_block_max(m, red);
    double z = 0.0;
    for (int j = threadIdx.x; j < S; j += kMatThreads) {
        const float e = expf(v[j] - m);
        v[j] = e;
        z += (double)e;
    }
    const float inv = (float)(1.0 / mat_block_sum(z, red));
    for (int j = threadIdx.x; j < S; j += kMatThreads) v[j] *= inv;
    __syncthreads();
}

__global__ void __launch_bounds__(kMatThreads)
risk_matrix_kernel(const float *__restrict__ ref, const float *__restrict__ x0, const float *__restrict__ rest, int B, int S, int nr,
                   int mode, int lt, int ideal, float *__restrict__ mat, float *__restrict__ jac) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double red[kMatThreads / LTR_WAVE];
    float *t = smem, *x = smem + S;
    const int b = blockIdx.x, tid = threadIdx.x, nsys = 1 + nr + (ideal ? 1 : 0);
    for (int j = tid; j < S; j += kMatThreads) t[j] = ref[(size_t)b * S + j];
    __syncthreads();
    if (mode != 1) mat_softmax(t, S, red);
    double nt_a = 0.0, st_a = 0.0;
    for (int j = tid; j < S; j += kMatThreads) {
        nt_a += (double)t[j] * t[j];
        st_a += (double)t[j];
    }
    const double nt = mat_block_sum(nt_a, red), st = mat_block_sum(st_a, red);
    for (int sys = 0; sys < nsys; ++sys) {
        const bool is_ideal = ideal && sys == nsys - 1;
        __syncthreads();
        for (int j = tid; j < S; j += kMatThreads) {
            float v;
            if (is_ideal) v = t[j];
            else if (sys == 0) v = x0[(size_t)b * S + j];
            else v = mode != 1 ? rest[((size_t)b * S + j) * nr + (sys - 1)] : rest[((size_t)(sys - 1) * B + b) * S + j];
            x[j] = v;
        }
        __syncthreads();
        if (mode != 1 && !is_ideal) mat_softmax(x, S, red);
        double a_a = 0.0, nx_a = 0.0, c_a = 0.0, sx_a = 0.0, ac_a = 0.0, nu_a = 0.0, nv_a = 0.0;
        for (int j = tid; j < S; j += kMatThreads) {
            const float tj = t[j], xj = x[j];
            a_a += (double)tj * xj;
            nx_a += (double)xj * xj;
            sx_a += (double)xj;
            const float df = mode == 1 ? xj - tj : tj * xj - tj * tj;
            c_a += (double)df * df;
            if (mode == 2) {                       // the tRisk pair's cosine is taken between the PRODUCTS t^2 and t x (:256-258)
                const float u = tj * tj, v = tj * xj;
                ac_a += (double)u * v;
                nu_a += (double)u * u;
                nv_a += (double)v * v;
            }
        }
        const double a = mat_block_sum(a_a, red), nx = mat_block_sum(nx_a, red), c = mat_block_sum(c_a, red), sx = mat_block_sum(sx_a, red);
        // cosine operands: (u, v) = (t, x) [modes 0, 1] or (t^2, t x) [mode 2]
        const double ca = mode == 2 ? mat_block_sum(ac_a, red) : a, cnu = mode == 2 ? mat_block_sum(nu_a, red) : nt,
                     cnv = mode == 2 ? mat_block_sum(nv_a, red) : nx;
        // torch.nn.CosineSimilarity (riskLosses.py: nn.CosineSimilarity(dim=1), eps = 1e-8) clamps EACH norm: u . v / (max(|u|, eps) max(|v|, eps))
        const double nrm_u = sqrt(cnu), nrm_v = sqrt(cnv);
        const double den = (nrm_u > 1e-8 ? nrm_u : 1e-8) * (nrm_v > 1e-8 ? nrm_v : 1e-8);
        double m;
        if (lt == 1) m = c;
        else if (lt == 2) m = ca / den;
        else m = mode == 1 ? (sx - st) * (sx - st) : (a - nt) * (a - nt);
        if (tid == 0) mat[(size_t)b * nsys + sys] = (float)m;
        if (sys == 0 && jac) {
            // g_j = d m / d x_j; mode 0: x = softmax(s): d m / d s_j = x_j (g_j - sum_k x_k g_k)
            // The installed torch (2.10, ATen cosine_similarity) clamps the two norms IN PLACE under no-grad: the VALUE uses
            // max(|v|, eps), the gradient still flows through the norm as d|v| / dv = v / |v| (0 at v = 0) -- measured on a vector of
            // norm 3e-9 (tests/test_risk_gpu.py::test_fused_risk_matrix_cosine_of_a_zero_vector):
            //   d m / d v = u / den - m v / (max(|v|, eps) |v|)
            const double nv_c = nrm_v > 1e-8 ? nrm_v : 1e-8;
            const double inv_vv = nrm_v > 0.0 ? 1.0 / (nv_c * nrm_v) : 0.0;
            auto grad = [&](int j) -> double {
                const double tj = t[j], xj = x[j];
                if (lt == 1) return mode == 1 ? 2.0 * (xj - tj) : 2.0 * tj * (tj * xj - tj * tj);
                if (lt == 2) {
                    const double u = mode == 2 ? tj * tj : tj, v = mode == 2 ? tj * xj : xj, w = mode == 2 ? tj : 1.0;
                    return w * (u / den - m * v * inv_vv);
                }
                return mode == 1 ? 2.0 * (sx - st) : 2.0 * (a - nt) * tj;
            };
            if (mode != 1) {
                double dot_a = 0.0;
                for (int j = tid; j < S; j += kMatThreads) dot_a += (double)x[j] * grad(j);
                const double dot = mat_block_sum(dot_a, red);
                for (int j = tid; j < S; j += kMatThreads) jac[(size_t)b * S + j] = (float)((double)x[j] * (grad(j) - dot));
            } else {
                for (int j = tid; j < S; j += kMatThreads) jac[(size_t)b * S + j] = (float)grad(j);
            }
        }
    }
}

}  // namespace

extern "C" {

int ltr_risk_fwd_bwd(const float *mat, int Q, int n_systems, int col, float alpha, int kind, float *value,
                     float *dmat, void *stream) {
    if (!mat || !value) return LTR_ERR_NULL;
    if (Q < 1 || n_systems < 1 || n_systems > 4096) return LTR_ERR_SHAPE;
    if (col < 0) col += n_systems;                       /* i = -1: the last system, as Python indexes */
    const int guard = (kind & LTR_RISK_ZERO_GUARD) ? 1 : 0;
    kind &= ~LTR_RISK_ZERO_GUARD;
    if (col < 0 || col >= n_systems || (kind != LTR_RISK_Z && kind != LTR_RISK_GEO)) return LTR_ERR_PARAM;
    hipLaunchKernelGGL(risk_kernel, dim3(1), dim3(kRiskThreads), 0, (hipStream_t)stream, mat, Q, n_systems, col, alpha,
                       kind == LTR_RISK_GEO ? 1 : 0, guard, value, dmat);
    return status();
}

int ltr_trisk_fwd_bwd(const float *model, const float *baseline, int Q, float alpha, float *value, float *dmodel,
                      float *dbaseline, void *stream) {
    if (!model || !baseline || !value) return LTR_ERR_NULL;
    if (Q < 1) return LTR_ERR_SHAPE;
    hipLaunchKernelGGL(trisk_kernel, dim3(1), dim3(kRiskThreads), 0, (hipStream_t)stream, model, baseline, Q, alpha, value,
                       dmodel, dbaseline);
    return status();
}

int ltr_trisk_tail_fwd_bwd(const float *mat, int Q, float alpha, int flip, float factor, float *value, float *dmat, void *stream) {
    if (!mat || !value) return LTR_ERR_NULL;
    if (Q < 1 || Q > (1 << 29)) return LTR_ERR_SHAPE;
    hipLaunchKernelGGL(trisk_tail_kernel, dim3(1), dim3(kRiskThreads), 0, (hipStream_t)stream, mat, Q, alpha, flip ? 1 : 0, factor, value,
                       dmat);
    return status();
}

int ltr_risk_tail_fwd_bwd(const float *mat, int Q, int n_systems, float alpha, int kind, int strategy, int flip, float factor,
                          int zquirk, float *value, float *dmat, void *stream) {
    if (!mat || !value) return LTR_ERR_NULL;
    if (Q < 1 || n_systems < 1 || (long long)Q * n_systems > (1ll << 30)) return LTR_ERR_SHAPE;
    if ((kind != 0 && kind != 1) || strategy < 1 || strategy > 3) return LTR_ERR_PARAM;
    hipLaunchKernelGGL(risk_tail_kernel, dim3(1), dim3(kRiskThreads), 0, (hipStream_t)stream, mat, Q, n_systems, alpha, kind, strategy,
                       flip ? 1 : 0, factor, zquirk ? 1 : 0, value, dmat);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

int ltr_risk_matrix_fwd(const float *ref, const float *x0, const float *rest, int B, int S, int n_rest, int mode, int lt, int ideal,
                        float *mat, float *jac, void *stream) {
    if (!ref || !x0 || !mat || (n_rest > 0 && !rest)) return LTR_ERR_NULL;
    if (B < 0 || S < 1 || S > kMatMaxS || n_rest < 0 || n_rest > 64) return LTR_ERR_SHAPE;
    if (mode < 0 || mode > 2 || lt < 1 || lt > 3) return LTR_ERR_PARAM;
    if (B == 0) return LTR_OK;
    hipLaunchKernelGGL(risk_matrix_kernel, dim3(B), dim3(kMatThreads), (size_t)2 * S * sizeof(float), (hipStream_t)stream, ref, x0, rest, B,
                       S, n_rest, mode, lt, ideal ? 1 : 0, mat, jac);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

}  // extern "C"
