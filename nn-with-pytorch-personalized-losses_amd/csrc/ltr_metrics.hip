// ltr_metrics.hip -- ranking evaluation metrics on the device (gfx950): NDCG@k per query.
//
// Replaces the per-query Python loops the reference runs after every epoch (main_batch_execution.py:173-200):
//   utils/metrics.py:48-65   dcg(true, pred, k, gains, use_numpy)
//   utils/metrics.py:67-73   ndcg(..., no_relevant)       ideal DCG 0 -> 1.0 (no_relevant) or 0.0
//   utils/metrics.py:76-80   mNdcg                        one ndcg per query
//   utils/metrics.py:83-104  torchNdcg                    = exponential gains, ideal DCG 0 -> 0.0
// One workgroup per query, labels and scores in LDS, sort-free: a document's rank is the number of documents
// that beat it (ties: lower index first = Python's stable sorted(..., reverse=True), the reference's default path;
// `reverse_ties` = higher index first = np.argsort(...)[::-1] with a stable argsort).  O(S^2) compares against
// 8 S bytes of HBM traffic per query: VALU-bound like the loss kernels, microseconds per thousand queries.
// Sums and logarithms are fp64 (the reference's numpy path is fp64); results are written as fp64.
#include "../../include/ltr_mi355x.h"
#include "ltr_device.h"
#include <math.h>

using namespace ltr;

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, LTR_WAVE);
    return v;
}

__global__ void __launch_bounds__(1024)
ndcg_kernel(const float *__restrict__ y_true, const float *__restrict__ y_score, int Q, int S, int k, int exponential,
            double no_relevant_value, int reverse_ties, double *__restrict__ out, double *__restrict__ dcg_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double red[2 * 16];
    float *y = smem, *s = smem + S;
    const long long q = ltr_block_id();
    if (q >= Q) return;
    const size_t off = (size_t)q * S;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        y[j] = y_true[off + j];
        s[j] = y_score[off + j];
    }
    __syncthreads();
    const int kk = k < S ? k : S;                                  // metrics.py:54-55
    double dcg = 0.0, idcg = 0.0;
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const float yi = y[i], si = s[i];
        int rs = 0, rl = 0;
        for (int j = 0; j < S; ++j) {
            const float sj = s[j], yj = y[j];
            rs += ((sj > si) || (sj == si && (reverse_ties ? j > i : j < i))) ? 1 : 0;
            rl += ((yj > yi) || (yj == yi && j < i)) ? 1 : 0;
        }
        const double g = exponential ? exp2((double)yi) - 1.0 : (double)yi;
        if (rs < kk) dcg += g / log2((double)rs + 2.0);
        if (rl < kk) idcg += g / log2((double)rl + 2.0);
    }
    dcg = wave_sum_f64(dcg);
    idcg = wave_sum_f64(idcg);
    const int w = threadIdx.x / LTR_WAVE, nw = blockDim.x / LTR_WAVE;
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) {
        red[2 * w] = dcg;
        red[2 * w + 1] = idcg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < nw; ++i) {
            a += red[2 * i];
            b += red[2 * i + 1];
        }
        if (out) out[q] = (b == 0.0) ? no_relevant_value : a / b;
        if (dcg_out) dcg_out[q] = a;
    }
}

}  // namespace

extern "C" {

int ltr_ndcg_at_k(const float *y_true, const float *y_score, int Q, int S, int k, int gains, int no_relevant,
                  int reverse_ties, double *ndcg, double *dcg, void *stream) {
    if (!y_true || !y_score || (!ndcg && !dcg)) return LTR_ERR_NULL;
    if (Q < 0 || S < 1 || S > 16384) return LTR_ERR_SHAPE;
    if (k < 1 || (gains != LTR_GAINS_LINEAR && gains != LTR_GAINS_EXPONENTIAL)) return LTR_ERR_PARAM;
    if (Q == 0) return LTR_OK;
    int block = next_pow2(S);
    block = block < 64 ? 64 : (block > 1024 ? 1024 : block);
    const size_t lds = (size_t)2 * S * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)ndcg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(ndcg_kernel, ltr_grid(Q), dim3(block), lds, (hipStream_t)stream, y_true, y_score, Q, S, k,
                       gains == LTR_GAINS_EXPONENTIAL ? 1 : 0, no_relevant ? 1.0 : 0.0, reverse_ties ? 1 : 0, ndcg, dcg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

}  // extern "C"
