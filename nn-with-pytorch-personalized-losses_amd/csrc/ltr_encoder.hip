// ltr_encoder.hip -- the set-transformer scorer of SURVEY.md row f-3 (BASELINE config 5) on gfx950:
// architeture/multiLayer.py:13-149 (FCModel / LTRModel / OutputLayer) and architeture/transformer.py:29-257
// (pre-norm encoder blocks: LayerNorm -> multi-head attention over the slate -> residual, LayerNorm -> FFN ->
// residual).  C ABI in include/ltr_encoder.h; the host side (ltr_mi355x/encoder.py) strings these launches together.
//
//   * ONE bf16 GEMM kernel (v_mfma_f32_16x16x32_bf16, fp32 accumulate) serves every Linear forward, input gradient
//     and weight gradient: 128 x 128 x 64 tiles, 4 waves of 64 x 64, register-prefetched double-buffered LDS images.
//     Operands whose contraction index is NOT the contiguous one (W in dx = dy W, both operands of dW = dy^T x) are
//     staged as they lie in memory and read with ds_read_b64_tr_b16, so nothing is transposed in HBM.
//   * attention: one workgroup per (slate, head); the whole S x S score row block of a 16-query tile lives in a
//     wave's registers (S^T = K Q^T orientation: the softmax axis is register-local + two lane swaps), P^T feeds the
//     P V product as the B operand straight from those registers.  The backward recomputes P in both orientations
//     (query-major for dQ, key-major for dK / dV) instead of transposing through LDS.
//   * LayerNorm / output scoring: one wave per token, fp32 statistics; parameter gradients as fixed-order
//     per-workgroup partials (no float atomics anywhere).
#include "../../include/ltr_encoder.h"
#include "../../include/ltr_mi355x.h"
#include "ltr_slate_losses.h"
#include <hip/hip_runtime.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ bf16_t to_bf16(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }
__device__ __forceinline__ float from_bf16(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// ------------------------------------------------------------------------------------------- dropout stream
__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// 32 hash bits shared by the FOUR elements 4*quad .. 4*quad + 3 of stream `stream_id`: one murmur3 finaliser over the quad
// counter under two per-(seed, stream) keys (below).  Element r of the quad
// is kept when the 16-bit window of the word starting at byte r (low byte r, high byte r+1 mod 4: each byte is the HIGH byte
// of one element and the tie-break LOW byte of its neighbour) is >= the threshold p * 65536: every element's window is an
// exactly uniform 16-bit number (two distinct bytes of a uniform word), so the keep probability is exactly
// 1 - thr / 65536, and a neighbour's byte decides an element only in the 1 / 256 of cases where its own high byte ties
// the threshold's.  Half the multiplies per element of one word per pair (the hash is ~20 % of the fused-FFN and ~25 %
// of the attention kernels' time at p > 0); the statistics are tested in tests/test_encoder_gpu.py.
// TWO 32-bit keys per (seed, stream): k0 is xored into the counter, k1 is ADDED between the two multiplies of the finaliser, so the
// streams of two (seed, stream) pairs are not XOR-translations of one fixed sequence (they would be with a single key folded in
// in front: mask_A[i] == mask_B[i ^ delta]).  Both are wave-uniform and hoisted out of every loop; the cost is one add per word.
// DEVICE-SIDE EPOCH: every kernel adds the 64-bit word g_drop_epoch to the seed it was launched with.  The word lives in device
// memory and only ltr_enc_seed_set / ltr_enc_seed_advance (one-thread kernels on the caller's stream) write it, so a launch
// sequence recorded ONCE into a hipGraph (kernel arguments frozen, seed included) draws fresh masks on every replay when the
// graph begins with an advance node.  0 (the value after load) leaves the stream of a seed exactly what it was.  Read through the
// constant address space: invariant for the duration of a kernel, one scalar load, hoisted out of every loop like the keys.
__device__ unsigned long long g_drop_epoch = 0ull;
__device__ __forceinline__ unsigned long long drop_seed(unsigned long long seed) {
    return seed + *(const __attribute__((address_space(4))) unsigned long long *)(&g_drop_epoch);
}
__device__ __forceinline__ unsigned drop_key0(unsigned long long seed, int stream_id) {
    return mix32((unsigned)drop_seed(seed) ^ (0x9E3779B9u * (unsigned)(stream_id + 1)));
}
__device__ __forceinline__ unsigned drop_key1(unsigned long long seed, int stream_id) {
    return mix32((unsigned)(drop_seed(seed) >> 32) + 0x85EBCA6Bu * (unsigned)stream_id + 0x165667B1u);
}
__global__ void seed_epoch_kernel(unsigned long long v, int add) { g_drop_epoch = add ? g_drop_epoch + v : v; }
// (32-bit integer multiplies are quarter rate: the high word of the quad counter -- zero below 2^34 elements per stream, i.e. always
// in practice -- joins the additive key between the two multiplies as (hi << 16 | hi) instead of costing a third multiply.)
__device__ __forceinline__ unsigned drop_word(unsigned long long seed, int stream_id, unsigned long long quad) {
    const unsigned hi = (unsigned)(quad >> 32);
    unsigned h = (unsigned)quad ^ drop_key0(seed, stream_id);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h += drop_key1(seed, stream_id) + ((hi << 16) | hi); h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline unsigned drop_threshold(float p) {
    if (!(p > 0.f)) return 0u;
    unsigned t = (unsigned)(p * 65536.f + 0.5f);
    return t > 65535u ? 65535u : t;
}
// the 16-bit window of element r (0..3) of a quad's word
__device__ __forceinline__ unsigned drop_win(unsigned w, unsigned r) { return __builtin_amdgcn_alignbit(w, w, 8u * r) & 0xffffu; }
__device__ __forceinline__ bool drop_keep(unsigned long long seed, int stream_id, unsigned long long idx, unsigned thr) {
    return drop_win(drop_word(seed, stream_id, idx >> 2), (unsigned)(idx & 3)) >= thr;
}
// keep flags of 4 consecutive elements idx .. idx+3, idx a multiple of 4: bit r = element idx + r
__device__ __forceinline__ unsigned drop_keep4(unsigned long long seed, int stream_id, unsigned long long idx, unsigned thr) {
    const unsigned w = drop_word(seed, stream_id, idx >> 2);
    return (drop_win(w, 0) >= thr ? 1u : 0u) | (drop_win(w, 1) >= thr ? 2u : 0u) | (drop_win(w, 2) >= thr ? 4u : 0u) |
           (drop_win(w, 3) >= thr ? 8u : 0u);
}

// The same four decisions as BOOLEANS (compare -> select, no bit packing and unpacking between producer and consumer: the packed form
// costs ~3 more vector instructions per element in kernels that are bound by exactly those).
struct Keep4 { bool k[4]; };
__device__ __forceinline__ Keep4 keep_all() { return Keep4{{true, true, true, true}}; }
__device__ __forceinline__ Keep4 drop_keep4b(unsigned long long seed, int stream_id, unsigned long long idx, unsigned thr) {
    const unsigned w = drop_word(seed, stream_id, idx >> 2);
    return Keep4{{drop_win(w, 0) >= thr, drop_win(w, 1) >= thr, drop_win(w, 2) >= thr, drop_win(w, 3) >= thr}};
}

// keep flags of 4 elements in ONE column over 4 consecutive rows (idx0 + r * row_stride, r = 0..3) when the four lanes of an
// aligned lane quad hold four adjacent columns of one element quad ((idx0 & 3) == (lane & 3), row_stride a multiple of 4):
// the word of row r is the same for the whole lane quad, so lane i evaluates row i and takes the other three with one DPP
// quad-perm broadcast each -- ONE hash per lane for its four elements.  EXEC must be full.
__device__ __forceinline__ unsigned drop_keep_col4(unsigned long long seed, int stream_id, unsigned long long idx0,
                                                   unsigned long long row_stride, unsigned thr, int lane) {
    const unsigned i = (unsigned)lane & 3u;
    const unsigned mine = drop_word(seed, stream_id, (idx0 >> 2) + (unsigned long long)i * (row_stride >> 2));
    const unsigned w0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xF, 0xF, false);      // quad_perm [0,0,0,0]
    const unsigned w1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xF, 0xF, false);      // [1,1,1,1]
    const unsigned w2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xAA, 0xF, 0xF, false);      // [2,2,2,2]
    const unsigned w3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xFF, 0xF, 0xF, false);      // [3,3,3,3]
    return (drop_win(w0, i) >= thr ? 1u : 0u) | (drop_win(w1, i) >= thr ? 2u : 0u) | (drop_win(w2, i) >= thr ? 4u : 0u) |
           (drop_win(w3, i) >= thr ? 8u : 0u);
}

__device__ __forceinline__ Keep4 drop_keep_col4b(unsigned long long seed, int stream_id, unsigned long long idx0,
                                                 unsigned long long row_stride, unsigned thr, int lane) {
    const unsigned i = (unsigned)lane & 3u;
    const unsigned mine = drop_word(seed, stream_id, (idx0 >> 2) + (unsigned long long)i * (row_stride >> 2));
    const unsigned w0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xF, 0xF, false);
    const unsigned w1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xF, 0xF, false);
    const unsigned w2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xAA, 0xF, 0xF, false);
    const unsigned w3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xFF, 0xF, 0xF, false);
    return Keep4{{drop_win(w0, i) >= thr, drop_win(w1, i) >= thr, drop_win(w2, i) >= thr, drop_win(w3, i) >= thr}};
}

__global__ void dropout_mask_kernel(unsigned long long seed, int stream_id, long long n, unsigned thr, uint8_t *out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        out[e] = drop_keep(seed, stream_id, (unsigned long long)e, thr) ? 1 : 0;
}

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// attention-probability element (slate-head bh, query q, key k) -> index in its dropout stream (Sp = S padded to 32)
__device__ __forceinline__ unsigned long long attn_idx(int bh, int Sp, int q, int k) {
    return ((unsigned long long)bh * Sp + q) * Sp + k;
}
__global__ void attn_dropout_mask_kernel(unsigned long long seed, int stream_id, int BH, int S, unsigned thr, uint8_t *out) {
    const int Sp = round_up(S, 32);
    const long long n = (long long)BH * S * S;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e % S), q = (int)((e / S) % S), bh = (int)(e / ((long long)S * S));
        out[e] = drop_keep(seed, stream_id, attn_idx(bh, Sp, q, k), thr) ? 1 : 0;
    }
}

__global__ void cast_bf16_kernel(const float *__restrict__ src, bf16_t *__restrict__ dst, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        dst[e] = to_bf16(src[e]);
}

// out[e] = sum_z parts[z][e] in a fixed order: 256 threads = ZL z-lanes x EW outputs (EW = 256 / ZL consecutive e, so
// loads stay coalesced); z-lane l sums z = l, l + ZL, ... in fp64, the ZL lane sums are then added in lane order.
template <int ZL>
__device__ __forceinline__ void sum_partials_body(const float *__restrict__ parts, int nsplit, long long n, long long stride,
                                                  int accumulate, float *__restrict__ out, int block, int nblocks,
                                                  double *red /* LDS [256] */) {
    constexpr int EW = 256 / ZL;
    const int el = threadIdx.x % EW, zl = threadIdx.x / EW;
    for (long long e0 = (long long)block * EW; e0 < n; e0 += (long long)nblocks * EW) {
        const long long e = e0 + el;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (e < n) {
            int z = zl;
            for (; z + 3 * ZL < nsplit; z += 4 * ZL) {
                s0 += (double)parts[(long long)z * stride + e];
                s1 += (double)parts[(long long)(z + ZL) * stride + e];
                s2 += (double)parts[(long long)(z + 2 * ZL) * stride + e];
                s3 += (double)parts[(long long)(z + 3 * ZL) * stride + e];
            }
            for (; z < nsplit; z += ZL) s0 += (double)parts[(long long)z * stride + e];
        }
        double s = (s0 + s1) + (s2 + s3);
        if (ZL > 1) {
            __syncthreads();
            red[threadIdx.x] = s;
            __syncthreads();
            if (zl == 0) {
                s = 0.0;
#pragma unroll
                for (int l = 0; l < ZL; ++l) s += red[l * EW + el];
            }
        }
        if (zl == 0 && e < n) out[e] = (float)(accumulate ? (double)out[e] + s : s);
    }
}
// few outputs: spend the threads on the split axis; many outputs: one thread per output
__host__ __device__ inline int reduce_zl(int nsplit, long long n) { return (n >= 65536 || nsplit < 8) ? 1 : (n >= 4096 ? 4 : 16); }
__device__ __forceinline__ void sum_partials_any(const float *parts, int nsplit, long long n, long long stride, int accumulate,
                                                 float *out, int block, int nblocks, double *red) {
    switch (reduce_zl(nsplit, n)) {        // workgroup-uniform
        case 1: sum_partials_body<1>(parts, nsplit, n, stride, accumulate, out, block, nblocks, red); break;
        case 4: sum_partials_body<4>(parts, nsplit, n, stride, accumulate, out, block, nblocks, red); break;
        default: sum_partials_body<16>(parts, nsplit, n, stride, accumulate, out, block, nblocks, red); break;
    }
}
__global__ void __launch_bounds__(256) sum_partials_kernel(const float *__restrict__ parts, int nsplit, long long n, int accumulate,
                                                           float *__restrict__ out) {
    __shared__ double red[256];
    sum_partials_any(parts, nsplit, n, n, accumulate, out, blockIdx.x, gridDim.x, red);
}
// The epilogue of a split-K activation GEMM whose result goes through bias + dropout + residual (the GEMM-path FFN's second layer at
// small steps): out[m][n] = residual[m][n] + drop(sum_s parts[s][m][n] + bias[n]), the same arithmetic, dropout stream and
// element order as gemm_bf16_kernel's own epilogue.  N a multiple of 4; four elements (one hash word) per thread.
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const float *__restrict__ parts, int nsplit, long long M, int N,
                                                              const float *__restrict__ bias, float drop_p, unsigned long long seed,
                                                              int stream_id, const float *__restrict__ residual, float *__restrict__ out) {
    const long long quads = M * (N / 4);
    const unsigned thr = drop_threshold(drop_p);
    const float keep_scale = thr ? 1.f / (1.f - drop_p) : 1.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < quads; e += (long long)gridDim.x * blockDim.x) {
        const long long m = e / (N / 4);
        const int n = 4 * (int)(e % (N / 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4 *>(parts + ((long long)sp * M + m) * N + n);
        if (bias) v += *reinterpret_cast<const f32x4 *>(bias + n);
        if (thr) {
            const Keep4 keep = drop_keep4b(seed, stream_id, (unsigned long long)m * N + n, thr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = keep.k[r] ? v[r] * keep_scale : 0.f;
        }
        if (residual) v += *reinterpret_cast<const f32x4 *>(residual + m * N + n);
        *reinterpret_cast<f32x4 *>(out + m * N + n) = v;
    }
}

// up to kReduceJobs independent reductions in one launch (blockIdx.y = job): the backward of one training step ends in
// ~60 of them, each a few microseconds of work
constexpr int kReduceJobs = 16;
struct ReduceJobs {
    const float *parts[kReduceJobs];
    float *out[kReduceJobs];
    long long n[kReduceJobs];
    long long stride[kReduceJobs];
    int nsplit[kReduceJobs];
    int blocks[kReduceJobs];
};
__global__ void __launch_bounds__(256) sum_partials_batch_kernel(ReduceJobs j) {
    __shared__ double red[256];
    const int k = blockIdx.y;
    if ((int)blockIdx.x >= j.blocks[k]) return;
    sum_partials_any(j.parts[k], j.nsplit[k], j.n[k], j.stride[k], 0, j.out[k], blockIdx.x, j.blocks[k], red);
}

// ------------------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------- LayerNorm
// One wave per token; lane l holds features l, l + 64, ...  (d <= 64 * kLnMax).
constexpr int kLnMax = 8;
constexpr int kLnThreads = 256;

struct LnStats {
    float mean, r, sigma;   // r = 1 / (std + eps)  (or rsqrt(var + eps)); sigma = std (annotated flavour only)
};
__device__ __forceinline__ LnStats ln_stats(const float (&v)[kLnMax], int d, int lane, float eps, int standard) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) s += (lane + 64 * j < d) ? v[j] : 0.f;
    LnStats st;
    st.mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        const float c = (lane + 64 * j < d) ? v[j] - st.mean : 0.f;
        q += c * c;
    }
    q = wave_sum(q);
    if (standard) {
        st.sigma = sqrtf(q / (float)d + eps);
        st.r = 1.f / st.sigma;
    } else {
        st.sigma = sqrtf(q / (float)(d - 1));
        st.r = 1.f / (st.sigma + eps);
    }
    return st;
}

__global__ void __launch_bounds__(kLnThreads)
layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b, long long T, int d,
                     float eps, int standard, bf16_t *__restrict__ yb, float *__restrict__ yf) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float av[kLnMax], bv[kLnMax];
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        const int f = lane + 64 * j;
        av[j] = f < d ? a[f] : 0.f;
        bv[j] = f < d ? b[f] : 0.f;
    }
    for (long long t = (long long)blockIdx.x * 4 + w; t < T; t += (long long)gridDim.x * 4) {
        float v[kLnMax];
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) v[j] = (lane + 64 * j < d) ? x[t * d + lane + 64 * j] : 0.f;
        const LnStats st = ln_stats(v, d, lane, eps, standard);
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) {
            const int f = lane + 64 * j;
            if (f < d) {
                const float y = av[j] * (v[j] - st.mean) * st.r + bv[j];
                if (yb) yb[t * d + f] = to_bf16(y);
                if (yf) yf[t * d + f] = y;
            }
        }
    }
}

// d % 4 == 0, d <= 4 LPT: LPT lanes x 4 consecutive features per token (one 16-byte load), 64 / LPT tokens per wave.
// LPT = 32: d <= 128 (two tokens per wave); LPT = 64: d <= 256 (the reference's default d_model = 136).  See layernorm_bwd_v4_kernel.
template <int LPT>
__global__ void __launch_bounds__(kLnThreads)
layernorm_fwd_v4_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b, long long T, int d,
                        float eps, int standard, bf16_t *__restrict__ yb, float *__restrict__ yf) {
    constexpr int TPW = 64 / LPT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane / LPT, f0 = 4 * (lane % LPT);
    const bool in = f0 < d;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = av;
    if (in) {
        av = *reinterpret_cast<const f32x4 *>(a + f0);
        bv = *reinterpret_cast<const f32x4 *>(b + f0);
    }
    for (long long t = ((long long)blockIdx.x * 4 + w) * TPW + half; t < T; t += (long long)gridDim.x * 4 * TPW) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (in) v = *reinterpret_cast<const f32x4 *>(x + t * d + f0);
        float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int o = LPT / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)d;
        f32x4 c = v - mean;
        if (!in) c = f32x4{0.f, 0.f, 0.f, 0.f};
        float q = (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
#pragma unroll
        for (int o = LPT / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        const float r = standard ? 1.f / sqrtf(q / (float)d + eps) : 1.f / (sqrtf(q / (float)(d - 1)) + eps);
        if (in) {
            const f32x4 y = av * c * r + bv;
            if (yb) *reinterpret_cast<u32x2 *>(yb + t * d + f0) = u32x2{pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3])};
            if (yf) *reinterpret_cast<f32x4 *>(yf + t * d + f0) = y;
        }
    }
}

// dx of one token through the norm, given g = dy * a per feature (in v_g) and the centred inputs.
//   annotated:  dx_j = r (g_j - mean g) - r^2 (sum g c) c_j / ((d-1) sigma)
//   standard:   dx_j = r (g_j - mean g - xhat_j mean(g xhat))
__device__ __forceinline__ void ln_dx(const float (&c)[kLnMax], const float (&g)[kLnMax], const LnStats &st, int d, int lane,
                                      int standard, float (&dx)[kLnMax]) {
    float sg = 0.f, sgc = 0.f;
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        sg += g[j];
        sgc += g[j] * c[j];
    }
    sg = wave_sum(sg);
    sgc = wave_sum(sgc);
    const float mg = sg / (float)d;
    float k2;
    if (standard) k2 = st.r * st.r * st.r * sgc / (float)d;
    else k2 = st.sigma > 0.f ? st.r * st.r * sgc / ((float)(d - 1) * st.sigma) : 0.f;
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) dx[j] = (lane + 64 * j < d) ? st.r * (g[j] - mg) - k2 * c[j] : 0.f;
}

// cross-wave sum of per-thread column accumulators -> partials[blockIdx.x][col_off + f]
__device__ __forceinline__ void block_cols_out(const float (&acc)[kLnMax], int d, float *red /* LDS [4][64*kLnMax] */,
                                               float *__restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) red[w * 64 * kLnMax + 64 * j + lane] = acc[j];
    __syncthreads();
    for (int f = threadIdx.x; f < d; f += kLnThreads)
        out[f] = (red[f] + red[64 * kLnMax + f]) + (red[2 * 64 * kLnMax + f] + red[3 * 64 * kLnMax + f]);
}

__global__ void __launch_bounds__(kLnThreads)
layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ dy, long long T, int d,
                     float eps, int standard, float *__restrict__ dxo, float *__restrict__ partials) {
    __shared__ float red[4 * 64 * kLnMax];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float av[kLnMax], da[kLnMax], db[kLnMax];
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        av[j] = (lane + 64 * j < d) ? a[lane + 64 * j] : 0.f;
        da[j] = db[j] = 0.f;
    }
    for (long long t = (long long)blockIdx.x * 4 + w; t < T; t += (long long)gridDim.x * 4) {
        float v[kLnMax], g[kLnMax], c[kLnMax], dx[kLnMax];
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) {
            const bool in = lane + 64 * j < d;
            v[j] = in ? x[t * d + lane + 64 * j] : 0.f;
            g[j] = in ? dy[t * d + lane + 64 * j] : 0.f;
        }
        const LnStats st = ln_stats(v, d, lane, eps, standard);
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) {
            c[j] = (lane + 64 * j < d) ? v[j] - st.mean : 0.f;
            da[j] += g[j] * c[j] * st.r;
            db[j] += g[j];
            g[j] *= av[j];
        }
        ln_dx(c, g, st, d, lane, standard, dx);
#pragma unroll
        for (int j = 0; j < kLnMax; ++j)
            if (lane + 64 * j < d) dxo[t * d + lane + 64 * j] += dx[j];
    }
    block_cols_out(da, d, red, partials + (long long)blockIdx.x * 2 * d);
    block_cols_out(db, d, red, partials + (long long)blockIdx.x * 2 * d + d);
}

// LayerNorm backward for d % 4 == 0, d <= 4 LPT: a token is LPT lanes x 4 consecutive features (one 16-byte load per
// array), a wave works on 64 / LPT tokens at once (LPT = 32: d <= 128, two tokens -- twice the bytes in flight of the general kernel
// above; LPT = 64: d <= 256).
template <int LPT>
__global__ void __launch_bounds__(kLnThreads)
layernorm_bwd_v4_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ dy, long long T, int d,
                        float eps, int standard, float *__restrict__ dxo, float *__restrict__ partials) {
    constexpr int TPW = 64 / LPT, WD = 4 * LPT;      // token slots per wave, features per slot
    __shared__ float red[4 * TPW * WD * 2];   // [4 waves x TPW token slots][da | db][WD]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane / LPT, l = lane % LPT, f0 = 4 * l;
    const bool in = f0 < d;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, da = av, db = av;
    if (in) av = *reinterpret_cast<const f32x4 *>(a + f0);
    for (long long t = ((long long)blockIdx.x * 4 + w) * TPW + half; t < T; t += (long long)gridDim.x * 4 * TPW) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = v;
        if (in) {
            v = *reinterpret_cast<const f32x4 *>(x + t * d + f0);
            g = *reinterpret_cast<const f32x4 *>(dy + t * d + f0);
        }
        // statistics over the token's LPT lanes (the xor distances stay inside the token's lane group)
        float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int o = LPT / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)d;
        f32x4 c = v - mean;
        if (!in) c = f32x4{0.f, 0.f, 0.f, 0.f};
        float q = (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
#pragma unroll
        for (int o = LPT / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        float sigma, r;
        if (standard) { sigma = sqrtf(q / (float)d + eps); r = 1.f / sigma; }
        else { sigma = sqrtf(q / (float)(d - 1)); r = 1.f / (sigma + eps); }
        da += g * c * r;
        db += g;
        g *= av;
        float sg = (g[0] + g[1]) + (g[2] + g[3]), sgc = (g[0] * c[0] + g[1] * c[1]) + (g[2] * c[2] + g[3] * c[3]);
#pragma unroll
        for (int o = LPT / 2; o >= 1; o >>= 1) {
            sg += __shfl_xor(sg, o, 64);
            sgc += __shfl_xor(sgc, o, 64);
        }
        const float mg = sg / (float)d;
        const float k2 = standard ? r * r * r * sgc / (float)d : (sigma > 0.f ? r * r * sgc / ((float)(d - 1) * sigma) : 0.f);
        if (in) {
            f32x4 o4 = *reinterpret_cast<const f32x4 *>(dxo + t * d + f0);
            o4 += r * (g - mg) - k2 * c;
            *reinterpret_cast<f32x4 *>(dxo + t * d + f0) = o4;
        }
    }
    // column sums: 4 TPW (wave, token slot) partial rows -> fixed-order sum
    float *row = red + (TPW * w + half) * 2 * WD;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        row[f0 + k] = da[k];
        row[WD + f0 + k] = db[k];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < 2 * d; f += kLnThreads) {
        const int col = f < d ? f : WD + (f - d);
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * TPW; ++k) sacc += red[k * 2 * WD + col];
        partials[(long long)blockIdx.x * 2 * d + f] = sacc;
    }
}

// scores[t] = w . LN(x[t]) + bias
__global__ void __launch_bounds__(kLnThreads)
score_fwd_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ wv,
                 const float *__restrict__ bias, long long T, int d, float eps, int norm, float *__restrict__ scores) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float aw[kLnMax], bw = 0.f;
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        const int f = lane + 64 * j;
        const float wj = f < d ? wv[f] : 0.f;
        aw[j] = norm ? (f < d ? a[f] * wj : 0.f) : wj;
        bw += (norm && f < d) ? b[f] * wj : 0.f;
    }
    bw = wave_sum(bw) + bias[0];
    for (long long t = (long long)blockIdx.x * 4 + w; t < T; t += (long long)gridDim.x * 4) {
        float v[kLnMax];
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) v[j] = (lane + 64 * j < d) ? x[t * d + lane + 64 * j] : 0.f;
        float s = 0.f;
        if (norm) {
            const LnStats st = ln_stats(v, d, lane, eps, norm == 2);
#pragma unroll
            for (int j = 0; j < kLnMax; ++j) s += (lane + 64 * j < d) ? aw[j] * (v[j] - st.mean) * st.r : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < kLnMax; ++j) s += aw[j] * v[j];
        }
        s = wave_sum(s);
        if (lane == 0) scores[t] = s + bw;
    }
}

__global__ void __launch_bounds__(kLnThreads)
score_bwd_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ wv,
                 const float *__restrict__ ds, long long T, int d, float eps, int norm, float *__restrict__ dxo,
                 float *__restrict__ partials) {
    __shared__ float red[4 * 64 * kLnMax];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float av[kLnMax], bv[kLnMax], wj[kLnMax], da[kLnMax], db[kLnMax], dw[kLnMax], dbias[kLnMax];
#pragma unroll
    for (int j = 0; j < kLnMax; ++j) {
        const int f = lane + 64 * j;
        wj[j] = f < d ? wv[f] : 0.f;
        av[j] = (norm && f < d) ? a[f] : 0.f;
        bv[j] = (norm && f < d) ? b[f] : 0.f;
        da[j] = db[j] = dw[j] = dbias[j] = 0.f;
    }
    for (long long t = (long long)blockIdx.x * 4 + w; t < T; t += (long long)gridDim.x * 4) {
        const float g0 = ds[t];
        float v[kLnMax];
#pragma unroll
        for (int j = 0; j < kLnMax; ++j) v[j] = (lane + 64 * j < d) ? x[t * d + lane + 64 * j] : 0.f;
        if (lane == 0) dbias[0] += g0;
        if (norm) {
            const LnStats st = ln_stats(v, d, lane, eps, norm == 2);
            float c[kLnMax], g[kLnMax], dx[kLnMax];
#pragma unroll
            for (int j = 0; j < kLnMax; ++j) {
                c[j] = (lane + 64 * j < d) ? v[j] - st.mean : 0.f;
                const float xh = c[j] * st.r;
                const float dyj = g0 * wj[j];                  // d loss / d y_j
                dw[j] += g0 * (av[j] * xh + bv[j]);
                da[j] += dyj * xh;
                db[j] += dyj;
                g[j] = dyj * av[j];
            }
            ln_dx(c, g, st, d, lane, norm == 2, dx);
#pragma unroll
            for (int j = 0; j < kLnMax; ++j)
                if (lane + 64 * j < d) dxo[t * d + lane + 64 * j] = dx[j];
        } else {
#pragma unroll
            for (int j = 0; j < kLnMax; ++j) {
                dw[j] += g0 * v[j];
                if (lane + 64 * j < d) dxo[t * d + lane + 64 * j] = g0 * wj[j];
            }
        }
    }
    float *out = partials + (long long)blockIdx.x * (3 * d + 8);
    block_cols_out(da, d, red, out);
    block_cols_out(db, d, red, out + d);
    block_cols_out(dw, d, red, out + 2 * d);
    block_cols_out(dbias, 1, red, out + 3 * d);
}

// ------------------------------------------------------------------------------------------- column sums
// 256 threads as (row lanes) x (column chunks of W floats): every thread sums its rows' chunk, the row lanes are then
// combined through LDS in a fixed order.
template <int W>
struct ColGeom {
    int cpr, rl, c, rlanes;   // chunks per row, this thread's row lane / chunk (loops when cpr > 256), row lanes in flight
    __device__ ColGeom(int N) {
        cpr = N / W;
        rlanes = cpr >= 256 ? 1 : 256 / cpr;
        rl = cpr >= 256 ? 0 : (int)threadIdx.x / cpr;
        c = cpr >= 256 ? (int)threadIdx.x : (int)threadIdx.x % cpr;
    }
    __device__ bool active() const { return rl < rlanes; }
};
template <int W>
__device__ __forceinline__ void colsum_out(const ColGeom<W> &cg, const float (&acc)[W], int c, float *red /* [256][W] */,
                                           float *__restrict__ out) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) red[threadIdx.x * W + k] = acc[k];
    __syncthreads();
    if (cg.rl == 0 && c < cg.cpr) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float s = 0.f;
            for (int r = 0; r < cg.rlanes; ++r) s += red[(cg.cpr >= 256 ? (int)threadIdx.x : r * cg.cpr + c) * W + k];
            out[c * W + k] = s;
        }
    }
}

__global__ void __launch_bounds__(256)
colsum_bf16_kernel(const bf16_t *__restrict__ y, long long T, int N, float *__restrict__ partials) {
    __shared__ float red[256 * 8];
    const ColGeom<8> cg(N);
    const long long rows_per = (T + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per, r1 = r0 + rows_per < T ? r0 + rows_per : T;
    for (int c0 = 0; c0 < cg.cpr; c0 += 256) {      // one pass unless N > 2048
        const int c = c0 + cg.c;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (cg.active() && c < cg.cpr)
            for (long long t = r0 + cg.rl; t < r1; t += cg.rlanes) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(y + t * N + c * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[2 * k] += bf16_lo(v[k]);
                    acc[2 * k + 1] += bf16_hi(v[k]);
                }
            }
        colsum_out<8>(cg, acc, c, red, partials + (long long)blockIdx.x * N);
    }
}

__global__ void __launch_bounds__(256)
drop_cast_colsum_kernel(const float *__restrict__ dx, long long T, int N, unsigned thr, float scale, unsigned long long seed,
                        int stream_id, bf16_t *__restrict__ out, float *__restrict__ partials) {
    __shared__ float red[256 * 4];
    const ColGeom<4> cg(N);
    const long long rows_per = (T + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per, r1 = r0 + rows_per < T ? r0 + rows_per : T;
    for (int c0 = 0; c0 < cg.cpr; c0 += 256) {
        const int c = c0 + cg.c;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (cg.active() && c < cg.cpr)
            for (long long t = r0 + cg.rl; t < r1; t += cg.rlanes) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(dx + t * N + c * 4);
                const Keep4 keep = thr ? drop_keep4b(seed, stream_id, (unsigned long long)t * N + c * 4, thr) : keep_all();
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // the GEMMs consume the ROUNDED value: sum that, so that db = colsum(dy) holds exactly
                    o[k] = from_bf16(to_bf16(keep.k[k] ? v[k] * scale : 0.f));
                    acc[k] += o[k];
                }
                *reinterpret_cast<u32x2 *>(out + t * N + c * 4) = u32x2{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            }
        colsum_out<4>(cg, acc, c, red, partials + (long long)blockIdx.x * N);
    }
}

// ------------------------------------------------------------------------------------------- GEMM
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDK = BK + 8;      // k-contiguous image [rows][LDK]: 144-byte rows, conflict-free b128 / b64 fragment reads
constexpr int LDM = BM + 16;     // k-major image [k][LDM]: 72 dwords = 8 mod 64 -> conflict-free transposed reads
constexpr int kTileElems = BM * LDK;   // == BK * LDM == 9216 bf16
static_assert(BM * LDK == BK * LDM && BM == BN, "both image shapes share one buffer size");
constexpr size_t kGemmLds = (size_t)4 * kTileElems * sizeof(bf16_t);   // A, B double-buffered

// ds_read_b64_tr_b16: fragment of row tile t (16 rows of the GEMM's m / n axis) over
// the 32 k rows starting at r0 of a [k][LD] image: lane (i = lane & 15, g = lane >> 4) gets row 16 t + i at
// k = r0 + 4 g + {0..3} and r0 + 16 + 4 g + {0..3}.
template <int LD>
__device__ __forceinline__ u32x4 tr_frag(const bf16_t *img, int r0, int t, int lane) {
    typedef __attribute__((address_space(3))) s16x4 *lp4;
    const int i = lane & 15, g = lane >> 4;
    const bf16_t *p = img + (r0 + 4 * g + (i >> 2)) * LD + 16 * t + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p + 16 * LD));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// global -> registers: this thread's 4 16-byte pieces of the (row0, k0) tile of one operand
template <bool KM>
__device__ __forceinline__ void gemm_load(const bf16_t *__restrict__ P, long long ld, long long rows, long long row0, long long k0,
                                          long long kend, int tid, u32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (KM) {
            const int kr = p >> 4, ch = p & 15;
            const long long k = k0 + kr, m = row0 + 8 * ch;
            r[i] = (k < kend && m < rows) ? *reinterpret_cast<const u32x4 *>(P + k * ld + m) : z;
        } else {
            const int row = p >> 3, ch = p & 7;
            const long long m = row0 + row, k = k0 + 8 * ch;
            r[i] = (m < rows && k < kend) ? *reinterpret_cast<const u32x4 *>(P + m * ld + k) : z;
        }
    }
}
template <bool KM>
__device__ __forceinline__ void gemm_store(bf16_t *img, int tid, const u32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        if (KM) *reinterpret_cast<u32x4 *>(img + (p >> 4) * LDM + 8 * (p & 15)) = r[i];
        else *reinterpret_cast<u32x4 *>(img + (p >> 3) * LDK + 8 * (p & 7)) = r[i];
    }
}
// fragment of 16-row tile `t` of the image for the 32-wide k sub-step kk.  SPLIT: the k <-> slot assignment of the
// transposed reads (4g.., 16+4g..), otherwise 8 consecutive k per lane group.
template <bool KM, bool SPLIT>
__device__ __forceinline__ u32x4 gemm_frag(const bf16_t *img, int t, int kk, int lane) {
    if (KM) return tr_frag<LDM>(img, 32 * kk, t, lane);
    const int i = lane & 15, g = lane >> 4;
    const bf16_t *row = img + (16 * t + i) * LDK + 32 * kk;
    if (SPLIT) {
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 4 * g), hi = *reinterpret_cast<const u32x2 *>(row + 16 + 4 * g);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
    return *reinterpret_cast<const u32x4 *>(row + 8 * g);
}

template <bool AKM, bool BKM>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(ltr_gemm_desc g) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr bool SPLIT = AKM || BKM;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const long long m0 = (long long)blockIdx.y * BM, n0 = (long long)blockIdx.x * BN;
    // split-K slice of this workgroup (whole BK steps)
    const long long ksteps = (g.K + BK - 1) / BK, per = (ksteps + g.splits - 1) / g.splits;
    const long long kbeg = (long long)blockIdx.z * per * BK;
    const long long kend = kbeg + per * BK < g.K ? kbeg + per * BK : g.K;
    const int nk = kend > kbeg ? (int)((kend - kbeg + BK - 1) / BK) : 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    if (nk > 0) {
        gemm_load<AKM>(g.A, g.lda, g.M, m0, kbeg, kend, tid, ra);
        gemm_load<BKM>(g.B, g.ldb, g.N, n0, kbeg, kend, tid, rb);
        gemm_store<AKM>(smem, tid, ra);
        gemm_store<BKM>(smem + kTileElems, tid, rb);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const bf16_t *As = smem + (t & 1) * 2 * kTileElems, *Bs = As + kTileElems;
        if (t + 1 < nk) {
            gemm_load<AKM>(g.A, g.lda, g.M, m0, kbeg + (long long)(t + 1) * BK, kend, tid, ra);
            gemm_load<BKM>(g.B, g.ldb, g.N, n0, kbeg + (long long)(t + 1) * BK, kend, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            u32x4 af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = gemm_frag<AKM, SPLIT>(As, wm * 4 + mi, kk, lane);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = gemm_frag<BKM, SPLIT>(Bs, wn * 4 + ni, kk, lane);
            // operands swapped: the accumulator tile is C^T (lane & 15 = m, registers = 4 consecutive n)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_bf16(bf[ni], af[mi], acc[mi][ni]);
        }
        if (t + 1 < nk) {
            bf16_t *An = smem + ((t + 1) & 1) * 2 * kTileElems;
            gemm_store<AKM>(An, tid, ra);
            gemm_store<BKM>(An + kTileElems, tid, rb);
        }
        __syncthreads();
    }

    // epilogue
    const int j = lane & 15, q = lane >> 4;
    const bool raw = g.splits > 1;
    float *Cf = raw ? g.Cf + (long long)blockIdx.z * g.M * g.ldc : g.Cf;
    const unsigned thr = raw ? 0u : drop_threshold(g.drop_p);
    const float keep_scale = thr ? 1.f / (1.f - g.drop_p) : 1.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const long long m = m0 + wm * 64 + mi * 16 + j;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const long long n = n0 + wn * 64 + ni * 16 + 4 * q;
            if (n >= g.N) continue;
            f32x4 v = acc[mi][ni];
            if (!raw) {
                if (g.bias) v += *reinterpret_cast<const f32x4 *>(g.bias + n);
                if (g.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (g.gate) {
                    const u32x2 gt = *reinterpret_cast<const u32x2 *>(g.gate + m * g.ldc + n);
                    v[0] = bf16_lo(gt[0]) > 0.f ? v[0] * g.gate_scale : 0.f;
                    v[1] = bf16_hi(gt[0]) > 0.f ? v[1] * g.gate_scale : 0.f;
                    v[2] = bf16_lo(gt[1]) > 0.f ? v[2] * g.gate_scale : 0.f;
                    v[3] = bf16_hi(gt[1]) > 0.f ? v[3] * g.gate_scale : 0.f;
                }
                if (thr) {
                    const Keep4 keep = drop_keep4b(g.seed, g.drop_stream, (unsigned long long)m * g.N + n, thr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = keep.k[r] ? v[r] * keep_scale : 0.f;
                }
                if (g.residual) v += *reinterpret_cast<const f32x4 *>(g.residual + m * g.ldc + n);
            }
            if (Cf) *reinterpret_cast<f32x4 *>(Cf + m * g.ldc + n) = v;
            if (!raw && g.Cb) *reinterpret_cast<u32x2 *>(g.Cb + m * g.ldc + n) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        }
    }
}

template <bool AKM, bool BKM>
int launch_gemm(const ltr_gemm_desc &g, hipStream_t stream) {
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)gemm_bf16_kernel<AKM, BKM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kGemmLds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_done[dev] = true;
    }
    const dim3 grid((unsigned)((g.N + BN - 1) / BN), (unsigned)((g.M + BM - 1) / BM), (unsigned)g.splits);
    hipLaunchKernelGGL((gemm_bf16_kernel<AKM, BKM>), grid, dim3(256), kGemmLds, stream, g);
    return status();
}

// ------------------------------------------------------------------------------------------- fused FFN
// PositionwiseFeedForward (transformer.py:215-237) + its SublayerConnection tail, WITHOUT a [T][d_ff] tensor in HBM:
// with d_model = 128 every FFN GEMM has at most 64 FLOP per byte of that tensor, so the unfused path is bound by
// writing and re-reading it (section 4.5 of DESIGN.md).  Here the hidden activation lives in registers only: it is
// produced 128 hidden units at a time in the accumulator layout whose registers ARE the next product's MFMA operand
// (k-slot permutation absorbed by how the weights are read from LDS), and the backward recomputes it.
//   forward   : x2 = x1 + drop(relu-drop(n2 W1^T + b1) W2^T + b2)              wave = 32 tokens, loops over d_ff chunks
//   backward X: dn2 = ((dy W2) gated) W1                                        same partition
//   backward W: dW1, dW2, db1 partials                                          workgroup = (chunk, token range), wave = 16 hidden
// D = d_model (multiple of 32, <= 128), d_ff multiple of 128.
constexpr int kFfnThreads = 512;
constexpr int kFfnChunk = 128;
constexpr int kFfnTok = 256;        // tokens per workgroup (forward / backward X)

struct FfnArgs {
    const bf16_t *n2, *w1, *w2, *dy;
    const float *b1, *b2, *x1;
    long long T;
    int dff;
    float p;
    unsigned long long seed;
    int stream_hidden, stream_out;
    float *out;            // fwd: x2 [T][D];  bwd X: dn2 [T][D]
    float *dw1, *dw2, *db1;   // bwd W partials [nsplit][dff][D], [nsplit][D][dff], [nsplit][dff]
    int nsplit;
};

// cooperative copy of a [ROWS][COLS] bf16 block (global row stride ld) into an LDS image with row stride LD, in two
// halves so that the global latency hides under the MFMA work in between: ffn_load issues this thread's 16-byte pieces
// into registers (rows >= row_lim read as zero), ffn_store writes them to the image.
template <int ROWS, int COLS>
__device__ __forceinline__ void ffn_load(const bf16_t *__restrict__ src, long long ld, long long row_lim, u32x4 (&r)[ROWS * COLS / 8 / kFfnThreads]) {
    constexpr int CPR = COLS / 8;
    static_assert(ROWS * CPR % kFfnThreads == 0, "whole pieces per thread");
#pragma unroll
    for (int i = 0; i < ROWS * CPR / kFfnThreads; ++i) {
        const int p = threadIdx.x + kFfnThreads * i, row = p / CPR, ch = p - row * CPR;
        const u32x4 z = {0u, 0u, 0u, 0u};
        r[i] = row < row_lim ? *reinterpret_cast<const u32x4 *>(src + (long long)row * ld + 8 * ch) : z;
    }
}
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ void ffn_store(bf16_t *img, const u32x4 (&r)[ROWS * COLS / 8 / kFfnThreads]) {
    constexpr int CPR = COLS / 8;
#pragma unroll
    for (int i = 0; i < ROWS * CPR / kFfnThreads; ++i) {
        const int p = threadIdx.x + kFfnThreads * i, row = p / CPR, ch = p - row * CPR;
        *reinterpret_cast<u32x4 *>(img + row * LD + 8 * ch) = r[i];
    }
}

// the same copy in batches of NB pieces (NB x 4 registers; each batch exposes one global-load latency).  NB = 1 for
// the weight-gradient kernel: with NB = 4 it spills 356 B / lane and runs 610 us instead of 303 (measured)
template <int ROWS, int COLS, int LD, int NB = 1>
__device__ __forceinline__ void ffn_stage(bf16_t *img, const bf16_t *__restrict__ src, long long ld, long long row_lim) {
    constexpr int CPR = COLS / 8, PIECES = ROWS * CPR / kFfnThreads;
    static_assert(ROWS * CPR % kFfnThreads == 0 && PIECES % NB == 0, "whole batches per thread");
#pragma unroll 1
    for (int b = 0; b < PIECES / NB; ++b) {
        u32x4 r[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = threadIdx.x + kFfnThreads * (b * NB + i), row = p / CPR, ch = p - row * CPR;
            const u32x4 z = {0u, 0u, 0u, 0u};
            r[i] = row < row_lim ? *reinterpret_cast<const u32x4 *>(src + (long long)row * ld + 8 * ch) : z;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p = threadIdx.x + kFfnThreads * (b * NB + i), row = p / CPR, ch = p - row * CPR;
            *reinterpret_cast<u32x4 *>(img + row * LD + 8 * ch) = r[i];
        }
    }
}

// ---- 128 x 128 bf16 tiles as XOR-swizzled 256-byte-row LDS images filled by LDS-DMA (D = 128 path) -----------------
// off(row, ch) = 256 row + 16 (ch ^ sw(row)), sw(row) = ((row & 3) << 2) | ((row >> 2) & 3): 16-byte chunk ch of row
// `row`.  One image serves ds_read_b128 row fragments, 8-byte split fragments and ds_read_b64_tr_b16 transposed
// fragments without padding, so global_load_lds (which can only write 64 lanes x 16 B contiguously) can fill it: the
// swizzle is applied to the SOURCE address each lane fetches.  No registers, asynchronous: issued a whole chunk / tile
// ahead, waited for (vmcnt) right before the barrier that publishes the buffer.
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ int sw_of(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int sw_off(int row, int ch) { return 256 * row + 16 * (ch ^ sw_of(row)); }     // bytes
// rows [0, 128) of a [rows][128] bf16 block (global row stride ld elements); rows >= row_lim re-read row row_lim - 1
// (finite data; their products are masked by the callers).  All 8 waves: wave w fills rows 16 w .. 16 w + 15.  The
// per-lane byte offsets of the four pieces depend on ld only: computed once (DmaLane), 32 bits each.
struct DmaLane { unsigned off[4]; };
__device__ __forceinline__ DmaLane dma_lane(long long ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    DmaLane L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * w + 4 * i + (lane >> 4);
        L.off[i] = (unsigned)(row * (int)ld + 8 * ((lane & 15) ^ sw_of(row))) * 2u;
    }
    return L;
}
__device__ __forceinline__ void dma_tile(bf16_t *img, const bf16_t *__restrict__ src, const DmaLane &L, long long ld, long long row_lim) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char *base = reinterpret_cast<const char *>(src);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned off = L.off[i];
        if (row_lim < 128) {            // workgroup-uniform: only a block's last, partial tile
            const int row = 16 * w + 4 * i + (lane >> 4);
            if (row >= row_lim) off = (unsigned)(((int)row_lim - 1) * (int)ld + 8 * ((lane & 15) ^ sw_of(row))) * 2u;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(reinterpret_cast<char *>(img) + (16 * w + 4 * i) * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Plain row-major variant (landing buffer [128][128], no swizzle) + the LDS -> LDS copy into a padded [128][LD] image:
// for the weight-gradient kernel, which has neither the registers for staged loads nor for swizzled addressing.
__device__ __forceinline__ void dma_tile_linear(bf16_t *landing, const bf16_t *__restrict__ src, long long row_lim) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char *base = reinterpret_cast<const char *>(src);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * w + 4 * i + (lane >> 4);
        const unsigned off = (unsigned)((row < row_lim ? row : (int)row_lim - 1) * 128 + 8 * (lane & 15)) * 2u;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(reinterpret_cast<char *>(landing) + (16 * w + 4 * i) * 256), 16, 0, 0);
    }
}
template <int LD>
__device__ __forceinline__ void lds_copy_tile(bf16_t *img, const bf16_t *landing) {
#pragma unroll 2
    for (int p = threadIdx.x; p < 128 * 16; p += kFfnThreads)
        *reinterpret_cast<u32x4 *>(img + (p >> 4) * LD + 8 * (p & 15)) = *reinterpret_cast<const u32x4 *>(landing + p * 8);
}
// Fragment addressing: sw(row) depends on row & 15 only, and every fragment's row is 16 * tile + (a lane constant), so
// the swizzle splits into a LANE-CONSTANT byte offset plus (compile-time chunk bits) ^ (lane-constant mask): a handful
// of address registers per image, tiles and k rows reached through immediate offsets.
struct SwRow { int base, mhi; };             // ds_read_b128 row fragments: row 16 tile + (lane & 15), chunk 4 k + (lane >> 4)
struct SwSplit { int lo, hi, mhi; };         // 8-byte split fragments: elements 32 u + 4 q .. and 32 u + 16 + 4 q ..
struct SwTr { int base, mhi; };              // ds_read_b64_tr_b16 fragments (same contract as tr_frag)
__device__ __forceinline__ SwRow sw_row_lane(int lane) {
    const int j = lane & 15, q = lane >> 4, m = sw_of(j) << 4;
    return SwRow{256 * j + ((16 * q) ^ (m & 0x30)), m & 0xC0};
}
__device__ __forceinline__ u32x4 sw_row_frag(const bf16_t *img, const SwRow &s, int tile, int k) {
    const char *lane_ptr = reinterpret_cast<const char *>(img) + (s.base + ((64 * k) ^ s.mhi));     // 4 distinct values per image
    return *reinterpret_cast<const u32x4 *>(lane_ptr + 4096 * tile);                                   // immediate offset
}
__device__ __forceinline__ SwSplit sw_split_lane(int lane) {
    const int j = lane & 15, q = lane >> 4, m = sw_of(j) << 4, c = 16 * (q >> 1), h = 8 * (q & 1);
    return SwSplit{256 * j + (c ^ (m & 0x30)) + h, 256 * j + ((c + 32) ^ (m & 0x30)) + h, m & 0xC0};
}
__device__ __forceinline__ u32x4 sw_split_frag(const bf16_t *img, const SwSplit &s, int tile, int u) {
    const int x = (64 * u) ^ s.mhi;
    const char *pl = reinterpret_cast<const char *>(img) + (s.lo + x), *ph = reinterpret_cast<const char *>(img) + (s.hi + x);
    // volatile: keeps these two ds_read_b64 (64 banks, 256 B/clk: conflict-free on this swizzle) from being merged with the
    // neighbouring tile's into ds_read2st64_b64, which banks modulo 32 -- rows j and j ^ 2 then collide (2-way) and the instruction
    // moves 128 B/clk (MI355X_MICROARCH.md, LDS table; SQ_LDS_BANK_CONFLICT 12.6 M cycles per launch of the forward kernel before)
    typedef const volatile __attribute__((address_space(3))) u32x2 *lds_v2;
    const u32x2 lo = *(lds_v2)(pl + 4096 * tile), hi = *(lds_v2)(ph + 4096 * tile);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ SwTr sw_tr_lane(int lane) {
    const int i = lane & 15, g = lane >> 4, rl = 4 * g + (i >> 2), p = i & 3, m = sw_of(rl) << 4;
    return SwTr{256 * rl + ((16 * (p >> 1)) ^ (m & 0x10)) + 8 * (p & 1), m & 0xE0};
}
// rows r0 .. r0 + 31 (r0 a multiple of 16), columns 16 t .. 16 t + 15
__device__ __forceinline__ u32x4 sw_tr_frag(const bf16_t *img, const SwTr &s, int r0, int t) {
    typedef __attribute__((address_space(3))) s16x4 *lp4;
    const char *lane_ptr = reinterpret_cast<const char *>(img) + (s.base + ((32 * t) ^ s.mhi));    // 8 distinct values per image
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(lane_ptr + 256 * r0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(lane_ptr + 256 * r0 + 16 * 256));      // row + 16: same swizzle
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// B-operand fragments of a token tile straight from global: token = lane & 15, k-step k of 32 features.
//   SPLIT = false: features 32 k + 8 g .. + 7 (pairs with ds_read_b128 row fragments)
//   SPLIT = true : features 32 k + 4 g .. + 3 and 32 k + 16 + 4 g .. + 3 (pairs with transposed reads)
template <bool SPLIT>
__device__ __forceinline__ u32x4 tok_frag(const bf16_t *__restrict__ base, long long tok, long long T, int D, int k, int g) {
    u32x4 r = {0u, 0u, 0u, 0u};
    if (tok >= T) return r;
    const bf16_t *row = base + tok * D + 32 * k;
    if (!SPLIT) return *reinterpret_cast<const u32x4 *>(row + 8 * g);
    const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 4 * g), hi = *reinterpret_cast<const u32x2 *>(row + 16 + 4 * g);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// hidden pre-activations of the wave's two token tiles for hidden tile ht of the chunk in LDS:
// z[tt][r] = (W1c x^T)[hidden 16 ht + 4 q + r][token (lane & 15) of tile tt]
template <int D, int LD, int TT>
__device__ __forceinline__ void ffn_z(const bf16_t *w1img, int ht, const u32x4 (&xf)[TT][D / 32], f32x4 (&z)[TT], int lane) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) z[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D / 32; ++k) {
        const u32x4 a = D == 128 ? sw_row_frag(w1img, sw_row_lane(lane), ht, k)
                                 : *reinterpret_cast<const u32x4 *>(w1img + (16 * ht + (lane & 15)) * LD + 32 * k + 8 * (lane >> 4));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) z[tt] = mfma_bf16(a, xf[tt][k], z[tt]);
    }
}

// TT = token tiles (of 16) per wave: 2 -> 256 tokens per workgroup; 1 -> 128 (twice the workgroups when T is small)
template <int D, int TT>
__global__ void __launch_bounds__(kFfnThreads) ffn_fwd_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr int DK = D / 32, DT = D / 16, LD1 = D == 128 ? 128 : D + 8, LD2 = D == 128 ? 128 : kFfnChunk + 8;
    constexpr bool DMA = D == 128;        // swizzled 256-byte-row images filled by LDS-DMA
    constexpr int IMG = kFfnChunk * LD1 + D * LD2;           // W1c [128][LD1] then W2c [D][LD2]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const long long tok0 = (long long)blockIdx.x * (128 * TT) + 16 * TT * w;
    const unsigned thr = drop_threshold(a.p);
    const float ks = thr ? 1.f / (1.f - a.p) : 1.f;
    u32x4 xf[TT][DK];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int k = 0; k < DK; ++k) xf[tt][k] = tok_frag<false>(a.n2, tok0 + 16 * tt + j, a.T, D, k, q);
    f32x4 y[DT][TT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) y[dt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nc = a.dff / kFfnChunk;
    u32x4 r1[kFfnChunk * D / 8 / kFfnThreads], r2[D * kFfnChunk / 8 / kFfnThreads];
    const DmaLane L1 = dma_lane(D), L2 = dma_lane(a.dff);
    if (DMA) {
        dma_tile(smem, a.w1, L1, D, kFfnChunk);
        dma_tile(smem + kFfnChunk * LD1, a.w2, L2, a.dff, D);
        dma_wait();
    } else {
        ffn_load<kFfnChunk, D>(a.w1, D, kFfnChunk, r1);
        ffn_load<D, kFfnChunk>(a.w2, a.dff, D, r2);
        ffn_store<kFfnChunk, D, LD1>(smem, r1);
        ffn_store<D, kFfnChunk, LD2>(smem + kFfnChunk * LD1, r2);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        const bf16_t *w1img = smem + (c & 1) * IMG, *w2img = w1img + kFfnChunk * LD1;
        if (c + 1 < nc) {       // next chunk into the other buffer (its last readers passed the barrier below)
            bf16_t *nx = smem + ((c + 1) & 1) * IMG;
            if (DMA) {
                dma_tile(nx, a.w1 + (long long)(c + 1) * kFfnChunk * D, L1, D, kFfnChunk);
                dma_tile(nx + kFfnChunk * LD1, a.w2 + (long long)(c + 1) * kFfnChunk, L2, a.dff, D);
            } else {
                ffn_load<kFfnChunk, D>(a.w1 + (long long)(c + 1) * kFfnChunk * D, D, kFfnChunk, r1);
                ffn_load<D, kFfnChunk>(a.w2 + (long long)(c + 1) * kFfnChunk, a.dff, D, r2);
                ffn_store<kFfnChunk, D, LD1>(nx, r1);
                ffn_store<D, kFfnChunk, LD2>(nx + kFfnChunk * LD1, r2);
            }
        }
        u32x2 hb[8][TT];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            f32x4 z[TT];
            ffn_z<D, LD1, TT>(w1img, ht, xf, z, lane);
            const int h0 = c * kFfnChunk + 16 * ht + 4 * q;
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b1 + h0);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                f32x4 v = z[tt] + bias;
                const Keep4 keep = thr ? drop_keep4b(a.seed, a.stream_hidden, (unsigned long long)(tok0 + 16 * tt + j) * a.dff + h0, thr) : keep_all();
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = keep.k[r] ? fmaxf(v[r], 0.f) * ks : 0.f;
                hb[ht][tt] = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 bfr[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) bfr[tt] = u32x4{hb[2 * u][tt][0], hb[2 * u][tt][1], hb[2 * u + 1][tt][0], hb[2 * u + 1][tt][1]};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                u32x4 af;
                if (DMA) af = sw_split_frag(w2img, sw_split_lane(lane), dt, u);
                else {
                    const bf16_t *pr = w2img + (16 * dt + j) * LD2 + 32 * u + 4 * q;
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(pr), hi = *reinterpret_cast<const u32x2 *>(pr + 16);
                    af = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) y[dt][tt] = mfma_bf16(af, bfr[tt], y[dt][tt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA) dma_wait();
        __syncthreads();
    }
    // y^T tiles: lane = token, registers = 4 consecutive d
    const unsigned thro = drop_threshold(a.p);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const long long tok = tok0 + 16 * tt + j;
        if (tok >= a.T) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * q;
            f32x4 v = y[dt][tt] + *reinterpret_cast<const f32x4 *>(a.b2 + d0);
            if (thro) {
                const Keep4 keep = drop_keep4b(a.seed, a.stream_out, (unsigned long long)tok * D + d0, thro);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = keep.k[r] ? v[r] * ks : 0.f;
            }
            v += *reinterpret_cast<const f32x4 *>(a.x1 + tok * D + d0);
            *reinterpret_cast<f32x4 *>(a.out + tok * D + d0) = v;
        }
    }
}

template <int D, int TT>
__global__ void __launch_bounds__(kFfnThreads) ffn_bwd_x_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr bool DMA = D == 128;        // swizzled 256-byte-row images filled by LDS-DMA
    constexpr int DK = D / 32, DT = D / 16, LD1 = DMA ? 128 : D + 16, LD2 = DMA ? 128 : kFfnChunk + 16;   // both images are also read transposed
    constexpr int IMG = kFfnChunk * LD1 + D * LD2;
    static_assert(DMA || (LD1 / 2) % 64 == 8 || (LD1 / 2) % 64 == 24 || (LD1 / 2) % 64 == 40 || (LD1 / 2) % 64 == 56, "transposed-read stride");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const long long tok0 = (long long)blockIdx.x * (128 * TT) + 16 * TT * w;
    const unsigned thr = drop_threshold(a.p);
    const float ks = thr ? 1.f / (1.f - a.p) : 1.f;
    u32x4 xf[TT][DK], dyf[TT][DK];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            xf[tt][k] = tok_frag<false>(a.n2, tok0 + 16 * tt + j, a.T, D, k, q);
            dyf[tt][k] = tok_frag<true>(a.dy, tok0 + 16 * tt + j, a.T, D, k, q);
        }
    f32x4 dn[DT][TT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) dn[dt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nc = a.dff / kFfnChunk;
    const DmaLane L1 = dma_lane(D), L2 = dma_lane(a.dff);
    if (DMA) {
        dma_tile(smem, a.w1, L1, D, kFfnChunk);
        dma_tile(smem + kFfnChunk * LD1, a.w2, L2, a.dff, D);
        dma_wait();
    } else {
        ffn_stage<kFfnChunk, D, LD1>(smem, a.w1, D, kFfnChunk);
        ffn_stage<D, kFfnChunk, LD2>(smem + kFfnChunk * LD1, a.w2, a.dff, D);
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
        const bf16_t *w1img = smem + (c & 1) * IMG, *w2img = w1img + kFfnChunk * LD1;
        if (c + 1 < nc) {       // next chunk into the other buffer (its last readers passed the barrier below)
            bf16_t *nx = smem + ((c + 1) & 1) * IMG;
            if (DMA) {
                dma_tile(nx, a.w1 + (long long)(c + 1) * kFfnChunk * D, L1, D, kFfnChunk);
                dma_tile(nx + kFfnChunk * LD1, a.w2 + (long long)(c + 1) * kFfnChunk, L2, a.dff, D);
            } else {
                ffn_stage<kFfnChunk, D, LD1>(nx, a.w1 + (long long)(c + 1) * kFfnChunk * D, D, kFfnChunk);
                ffn_stage<D, kFfnChunk, LD2>(nx + kFfnChunk * LD1, a.w2 + (long long)(c + 1) * kFfnChunk, a.dff, D);
            }
        }
        // gate bits: the hidden unit is alive (relu) and kept (dropout)
        unsigned gate[TT] = {};               // bit 4 ht + r of word tt
        {
#pragma unroll
            for (int ht = 0; ht < 8; ++ht) {
                f32x4 z[TT];
                ffn_z<D, LD1, TT>(w1img, ht, xf, z, lane);
                const int h0 = c * kFfnChunk + 16 * ht + 4 * q;
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b1 + h0);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const unsigned keep = thr ? drop_keep4(a.seed, a.stream_hidden, (unsigned long long)(tok0 + 16 * tt + j) * a.dff + h0, thr) : 15u;
                    unsigned bits = 0u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // alive iff the FORWARD's stored bf16 activation is > 0 (same test as the unfused gate)
                        const float hv = from_bf16(to_bf16(fmaxf(z[tt][r] + bias[r], 0.f) * ks));
                        bits |= (hv > 0.f ? 1u : 0u) << r;
                    }
                    gate[tt] |= (bits & keep) << (4 * ht);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // dh^T = W2c^T dy^T (A: transposed read of the [d][hidden] image), gated -> dz^T as the next B operand
        u32x2 dzb[8][TT];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            f32x4 dh[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) dh[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                const u32x4 af = DMA ? sw_tr_frag(w2img, sw_tr_lane(lane), 32 * k, ht) : tr_frag<LD2>(w2img, 32 * k, ht, lane);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) dh[tt] = mfma_bf16(af, dyf[tt][k], dh[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[tt][r] = (gate[tt] >> (4 * ht + r)) & 1u ? dh[tt][r] * ks : 0.f;
                dzb[ht][tt] = u32x2{pack_bf16(dh[tt][0], dh[tt][1]), pack_bf16(dh[tt][2], dh[tt][3])};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // dn2^T += W1c^T dz^T (A: transposed read of the [hidden][d] image)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 bfr[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) bfr[tt] = u32x4{dzb[2 * u][tt][0], dzb[2 * u][tt][1], dzb[2 * u + 1][tt][0], dzb[2 * u + 1][tt][1]};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u32x4 af = DMA ? sw_tr_frag(w1img, sw_tr_lane(lane), 32 * u, dt) : tr_frag<LD1>(w1img, 32 * u, dt, lane);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) dn[dt][tt] = mfma_bf16(af, bfr[tt], dn[dt][tt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA) dma_wait();
        __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const long long tok = tok0 + 16 * tt + j;
        if (tok >= a.T) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4 *>(a.out + tok * D + 16 * dt + 4 * q) = dn[dt][tt];
    }
}

// workgroup = (chunk c, token range rg); wave w = hidden units 16 w .. 16 w + 15 of the chunk.  XCD-aware ids: all chunks
// of one token range share a residue class mod 8 (one XCD, one L2), so a range's n2 / dy tiles come from HBM once instead
// of once per XCD (measured: 1.82 -> GB of HBM traffic per step for this kernel)
template <int D>
__global__ void __launch_bounds__(kFfnThreads) ffn_bwd_w_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    // LDS-DMA + swizzled images measured SLOWER here (456 vs 303 us at T = 65 536: the kernel then needs ~60 registers of
    // scratch next to its 64 dW accumulators), and staging through registers exposes 8 load latencies per tile.  So
    // (D = 128): LDS-DMA the NEXT tile into a plain landing buffer while this one is consumed from the padded images,
    // then copy LDS -> LDS (LAND).  D = 64 stages through registers, piece by piece.
    constexpr bool DMA = false, LAND = D == 128;
    constexpr int DK = D / 32, DT = D / 16, LD = DMA ? 128 : D + 16, TOK = 128, IMG = 2 * TOK * LD;   // n2 tile then dy tile, [token][d]
    bf16_t *landing = smem + IMG;          // LAND: [2][128][128] behind the single image pair
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int nchunk = a.dff / kFfnChunk, xcd = blockIdx.x & 7, nn = blockIdx.x >> 3;
    const int c = nn % nchunk, rg = (nn / nchunk) * 8 + xcd;               // grid = ceil(nsplit / 8) * 8 * nchunk
    if (rg >= a.nsplit) return;
    const int hid = c * kFfnChunk + 16 * w + j;          // this lane's hidden unit (as a column)
    const unsigned thr = drop_threshold(a.p);
    const float ks = thr ? 1.f / (1.f - a.p) : 1.f;
    // token range of this workgroup, whole 128-token tiles
    const long long tiles = (a.T + TOK - 1) / TOK, per = (tiles + a.nsplit - 1) / a.nsplit;
    const long long t_beg = (long long)rg * per, t_end = t_beg + per < tiles ? t_beg + per : tiles;
    // B operands of the two recomputed products, constant for the wave: W1[hid][d 32k + 8g ..], W2[d 32k + 8g ..][hid]
    u32x4 w1f[DK], w2f[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) {
        w1f[k] = *reinterpret_cast<const u32x4 *>(a.w1 + (long long)hid * D + 32 * k + 8 * q);
        unsigned short e[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) e[x] = a.w2[(long long)(32 * k + 8 * q + x) * a.dff + hid];
        w2f[k] = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                       (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
    }
    const float b1v = a.b1[hid];
    f32x4 dw2[DT], dw1[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dw2[dt] = dw1[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db1 = 0.f;
    const DmaLane L1 = dma_lane(D);
    if (LAND && t_beg < t_end) {
        dma_tile_linear(landing, a.n2 + t_beg * TOK * D, a.T - t_beg * TOK);
        dma_tile_linear(landing + TOK * 128, a.dy + t_beg * TOK * D, a.T - t_beg * TOK);
        dma_wait();
        __syncthreads();
        lds_copy_tile<LD>(smem, landing);
        lds_copy_tile<LD>(smem + TOK * LD, landing + TOK * 128);
        __syncthreads();
        if (t_beg + 1 < t_end) {
            dma_tile_linear(landing, a.n2 + (t_beg + 1) * TOK * D, a.T - (t_beg + 1) * TOK);
            dma_tile_linear(landing + TOK * 128, a.dy + (t_beg + 1) * TOK * D, a.T - (t_beg + 1) * TOK);
        }
    } else if (t_beg < t_end) {
        if (DMA) {
            dma_tile(smem, a.n2 + t_beg * TOK * D, L1, D, a.T - t_beg * TOK);
            dma_tile(smem + TOK * LD, a.dy + t_beg * TOK * D, L1, D, a.T - t_beg * TOK);
            dma_wait();
        } else {
            ffn_stage<TOK, D, LD>(smem, a.n2 + t_beg * TOK * D, D, a.T - t_beg * TOK);
            ffn_stage<TOK, D, LD>(smem + TOK * LD, a.dy + t_beg * TOK * D, D, a.T - t_beg * TOK);
        }
    }
    __syncthreads();
    for (long long t = t_beg; t < t_end; ++t) {
        const bf16_t *ximg = smem + (LAND ? 0 : ((t - t_beg) & 1) * IMG), *dyimg = ximg + TOK * LD;
        if (!LAND && t + 1 < t_end) {
            bf16_t *nx = smem + ((t + 1 - t_beg) & 1) * IMG;
            if (DMA) {
                dma_tile(nx, a.n2 + (t + 1) * TOK * D, L1, D, a.T - (t + 1) * TOK);
                dma_tile(nx + TOK * LD, a.dy + (t + 1) * TOK * D, L1, D, a.T - (t + 1) * TOK);
            } else {
                ffn_stage<TOK, D, LD>(nx, a.n2 + (t + 1) * TOK * D, D, a.T - (t + 1) * TOK);
                ffn_stage<TOK, D, LD>(nx + TOK * LD, a.dy + (t + 1) * TOK * D, D, a.T - (t + 1) * TOK);
            }
        }
        u32x2 hb[8], dzb[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f}, dh = z;        // rows = tokens 16 tt + 4 q + r, column = this lane's hidden unit
#pragma unroll
            for (int k = 0; k < DK; ++k) {
                const u32x4 ax = DMA ? sw_row_frag(ximg, sw_row_lane(lane), tt, k) : *reinterpret_cast<const u32x4 *>(ximg + (16 * tt + j) * LD + 32 * k + 8 * q);
                const u32x4 ay = DMA ? sw_row_frag(dyimg, sw_row_lane(lane), tt, k) : *reinterpret_cast<const u32x4 *>(dyimg + (16 * tt + j) * LD + 32 * k + 8 * q);
                z = mfma_bf16(ax, w1f[k], z);
                dh = mfma_bf16(ay, w2f[k], dh);
            }
            f32x4 h, dz;
            const long long tok0 = t * TOK + 16 * tt + 4 * q;
            const Keep4 keep4 = thr ? drop_keep_col4b(a.seed, a.stream_hidden, (unsigned long long)tok0 * a.dff + hid, (unsigned long long)a.dff, thr, lane) : keep_all();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long tok = tok0 + r;
                const bool keep = keep4.k[r];
                h[r] = keep && tok < a.T ? from_bf16(to_bf16(fmaxf(z[r] + b1v, 0.f) * ks)) : 0.f;
                dz[r] = h[r] > 0.f ? from_bf16(to_bf16(dh[r] * ks)) : 0.f;
                db1 += dz[r];
            }
            hb[tt] = u32x2{pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3])};
            dzb[tt] = u32x2{pack_bf16(dz[0], dz[1]), pack_bf16(dz[2], dz[3])};
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {        // 32 tokens per k step
            const u32x4 hB = {hb[2 * u][0], hb[2 * u][1], hb[2 * u + 1][0], hb[2 * u + 1][1]};
            const u32x4 dzA = {dzb[2 * u][0], dzb[2 * u][1], dzb[2 * u + 1][0], dzb[2 * u + 1][1]};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dw2[dt] = mfma_bf16(DMA ? sw_tr_frag(dyimg, sw_tr_lane(lane), 32 * u, dt) : tr_frag<LD>(dyimg, 32 * u, dt, lane), hB, dw2[dt]);     // [d 16 dt + 4 q + r][hid]
                dw1[dt] = mfma_bf16(dzA, DMA ? sw_tr_frag(ximg, sw_tr_lane(lane), 32 * u, dt) : tr_frag<LD>(ximg, 32 * u, dt, lane), dw1[dt]);     // [hidden 16 w + 4 q + r][d 16 dt + j]
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA || LAND) dma_wait();
        __syncthreads();
        if (LAND && t + 1 < t_end) {        // every wave is done with the images; the landing buffer holds tile t + 1
            lds_copy_tile<LD>(smem, landing);
            lds_copy_tile<LD>(smem + TOK * LD, landing + TOK * 128);
            __syncthreads();
            if (t + 2 < t_end) {
                dma_tile_linear(landing, a.n2 + (t + 2) * TOK * D, a.T - (t + 2) * TOK);
                dma_tile_linear(landing + TOK * 128, a.dy + (t + 2) * TOK * D, a.T - (t + 2) * TOK);
            }
        }
    }
    float *p1 = a.dw1 + (long long)rg * a.dff * D, *p2 = a.dw2 + (long long)rg * D * a.dff;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p2[(long long)(16 * dt + 4 * q + r) * a.dff + hid] = dw2[dt][r];
            p1[(long long)(c * kFfnChunk + 16 * w + 4 * q + r) * D + 16 * dt + j] = dw1[dt][r];
        }
    db1 += __shfl_xor(db1, 16, 64);
    db1 += __shfl_xor(db1, 32, 64);
    if (q == 0) a.db1[(long long)rg * a.dff + hid] = db1;
}

// ------------------------------------------------------------------------------------------- attention
constexpr int kAttThreads = 256;
constexpr int kDkPad = 32;              // head dimension padded to one MFMA k step
constexpr int kRowLd = kDkPad + 8;      // [token][d] images: 80-byte rows, conflict-free b128 row fragments
// [d][token] images (V^T, K^T, Q^T, dO^T): row stride = 4 dwords mod 64 -> the two 8-byte reads per fragment
// (tokens 4g.., 16+4g..) of 16 rows x 2 lane groups cover all 64 banks
__host__ __device__ inline int tr_ld(int Sp) { return ((Sp / 2 + 59) / 64 * 64 + 4) * 2; }

struct AttArgs {
    const bf16_t *qkv;
    const bf16_t *dctx;
    const bf16_t *ctx;    // backward only: the forward's output
    const uint8_t *mask;
    bf16_t *out;          // fwd: ctx [T][d]; bwd: dqkv [T][3d]
    int B, S, h, dk;
    float drop_p;
    unsigned long long seed;
    int stream_id;
    float *lse;           // [B*h][S] fp32 log2-sum-exp of the scaled scores: written by the forward when non-null, read by the
                          // key-major backward (attention_bwd_km_kernel); nullptr = the two-phase backward recomputes it
};

// 8 consecutive head features d0 .. d0+7 of one token row (zero beyond dk / for missing rows)
__device__ __forceinline__ u32x4 load8(const bf16_t *row, int d0, int dk, bool vec, bool valid) {
    u32x4 r = {0u, 0u, 0u, 0u};
    if (!valid) return r;
    if (vec) {
        if (d0 + 8 <= dk) r = *reinterpret_cast<const u32x4 *>(row + d0);
        return r;
    }
    // head slices that are not multiples of 16 bytes (dk = 17: 34-byte slices): the (at most two) ALIGNED 16-byte chunks that hold
    // the valid elements, funnel-shifted into place.  Token rows are multiples of 16 bytes (d % 8 == 0) and d0 is a multiple of 8, so
    // the misalignment is the same for every lane of the workgroup (scalar); a chunk is only touched if it holds a valid element, so
    // nothing outside the row is read.  (Was: eight guarded 2-byte loads per lane.)
    const int nvalid = dk - d0 < 8 ? dk - d0 : 8;
    if (nvalid <= 0) return r;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(row + d0);
    const unsigned sh = (unsigned)__builtin_amdgcn_readfirstlane((int)(addr & 15u));
    const u32x4 *q = reinterpret_cast<const u32x4 *>(addr - (addr & 15u));
    const u32x4 c0 = q[0];
    u32x4 c1 = {0u, 0u, 0u, 0u};
    if ((int)sh + 2 * nvalid > 16) c1 = q[1];
    const unsigned bs = sh & 3u;                  // 0 or 2 bytes (bf16 elements)
    switch (sh >> 2) {                            // scalar branch
        case 0: r = u32x4{__builtin_amdgcn_alignbyte(c0[1], c0[0], bs), __builtin_amdgcn_alignbyte(c0[2], c0[1], bs),
                          __builtin_amdgcn_alignbyte(c0[3], c0[2], bs), __builtin_amdgcn_alignbyte(c1[0], c0[3], bs)}; break;
        case 1: r = u32x4{__builtin_amdgcn_alignbyte(c0[2], c0[1], bs), __builtin_amdgcn_alignbyte(c0[3], c0[2], bs),
                          __builtin_amdgcn_alignbyte(c1[0], c0[3], bs), __builtin_amdgcn_alignbyte(c1[1], c1[0], bs)}; break;
        case 2: r = u32x4{__builtin_amdgcn_alignbyte(c0[3], c0[2], bs), __builtin_amdgcn_alignbyte(c1[0], c0[3], bs),
                          __builtin_amdgcn_alignbyte(c1[1], c1[0], bs), __builtin_amdgcn_alignbyte(c1[2], c1[1], bs)}; break;
        default: r = u32x4{__builtin_amdgcn_alignbyte(c1[0], c0[3], bs), __builtin_amdgcn_alignbyte(c1[1], c1[0], bs),
                           __builtin_amdgcn_alignbyte(c1[2], c1[1], bs), __builtin_amdgcn_alignbyte(c1[3], c1[2], bs)}; break;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) r[jj] &= nvalid >= 2 * jj + 2 ? 0xffffffffu : (nvalid == 2 * jj + 1 ? 0x0000ffffu : 0u);
    return r;
}
// 4 consecutive head features d0..d0+3 of an accumulator -> bf16 row (guarded)
__device__ __forceinline__ void store4(bf16_t *row, int d0, int dk, bool vec, const f32x4 &v) {
    if (vec && d0 + 4 <= dk) {
        *reinterpret_cast<u32x2 *>(row + d0) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (d0 + r < dk) row[d0 + r] = to_bf16(v[r]);
}

// Workgroup id -> (slate, head), XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each
// with its own L2), and the h heads of one slate read interleaved 2*dk-byte slices of the SAME qkv rows -- so all heads of
// a slate are given ids of one residue class mod 8 (one XCD): its L2 then fetches every qkv line once instead of once per
// head.  The grid is ceil(B / 8) * 8 * h; ids whose slate is >= B exit.  Returns false for those.
__device__ __forceinline__ bool att_slate_head(int B, int h, int &b, int &hd) {
    const int x = blockIdx.x & 7, n = blockIdx.x >> 3;
    b = (n / h) * 8 + x;
    hd = n % h;
    return b < B;
}
inline unsigned att_grid(int B, int h) { return (unsigned)((B + 7) / 8 * 8 * h); }

// stage `which` (0 Q, 1 K, 2 V of qkv; 3 = dctx) of slate b / head hd: row image [Sp][kRowLd] and/or transposed image
// [32][ldt].  All threads of the workgroup.
__device__ __forceinline__ void stage_head(const AttArgs &a, int b, int hd, int which, int Sp, bf16_t *rows, bf16_t *tr, int ldt,
                                           int trch = 4) {
    const int d = a.h * a.dk;
    const bool vec = a.dk % 8 == 0;
    const bf16_t *base = which < 3 ? a.qkv + (long long)b * a.S * 3 * d + which * d + hd * a.dk
                                   : a.dctx + (long long)b * a.S * d + hd * a.dk;
    const long long ld = which < 3 ? 3 * d : d;
    for (int e = threadIdx.x; e < Sp * 4; e += kAttThreads) {
        const int tok = e >> 2, ch = e & 3;
        const u32x4 v = load8(base + tok * ld, 8 * ch, a.dk, vec, tok < a.S);
        if (rows) *reinterpret_cast<u32x4 *>(rows + tok * kRowLd + 8 * ch) = v;
        if (tr && ch < trch) {       // trch = 2: a 16-row image (dk <= 16)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                tr[(8 * ch + 2 * x) * ldt + tok] = (bf16_t)(v[x] & 0xffffu);
                tr[(8 * ch + 2 * x + 1) * ldt + tok] = (bf16_t)(v[x] >> 16);
            }
        }
    }
}

// row fragment (A or B operand with the token on lane & 15, 8 consecutive d per lane group)
__device__ __forceinline__ u32x4 row_frag(const bf16_t *rows, int tile, int lane) {
    return *reinterpret_cast<const u32x4 *>(rows + (16 * tile + (lane & 15)) * kRowLd + 8 * (lane >> 4));
}
// fragment of a [d][token] image: d = 16 dt + (lane & 15), tokens 32 u + 4 g + {0..3} and 32 u + 16 + 4 g + {0..3}
__device__ __forceinline__ u32x4 col_frag(const bf16_t *tr, int ldt, int dt, int u, int lane) {
    const bf16_t *p = tr + (16 * dt + (lane & 15)) * ldt + 32 * u + 4 * (lane >> 4);
    const u32x2 lo = *reinterpret_cast<const u32x2 *>(p), hi = *reinterpret_cast<const u32x2 *>(p + 16);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
// reduce over the 4 lane groups holding one query (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}

// Softmax probabilities of one 16-query tile in the S^T orientation: st[kt][r] = raw score (q . k) of key
// 16 kt + 4 g + r for query (lane & 15), in place -> p.  Base-2 arithmetic: c2 = log2(e) / sqrt(dk) is folded into the
// exponent; lse2 = log2 sum_k 2^(c2 s_k) (+inf for a query without any unmasked key) lets the key-major backward
// phase recompute p = 2^(c2 s - lse2) with one fma and one v_exp_f32.
template <int KTMAX>
__device__ __forceinline__ void softmax_tile(f32x4 (&st)[KTMAX], int KT, const float *biasS, float c2, int g, float &lse2) {
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt)
        if (kt < KT) {
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(biasS + 16 * kt + 4 * g);     // 0, or -inf for masked keys
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = fmaf(st[kt][r], c2, bias[r]);
                m = fmaxf(m, st[kt][r]);
            }
        }
    m = quad_max(m);
    if (m == -INFINITY) m = 0.f;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt)
        if (kt < KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] - m);
                l += st[kt][r];
            }
        }
    l = quad_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    lse2 = l > 0.f ? m + __builtin_amdgcn_logf(l) : INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTMAX; ++kt)
        if (kt < KT) st[kt] *= inv;
}

// FULL: the padded slate fills all KTMAX key tiles (the `kt < KT` guards fold away)
template <int KTMAX, bool FULL>
__global__ void __launch_bounds__(kAttThreads, (KTMAX <= 16 ? 3 : 1)) attention_fwd_kernel(AttArgs a) {   // S <= 256: 3 workgroups per CU (170 VGPRs, 38 KB of LDS each)
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int Sp = round_up(a.S, 32), KT = FULL ? KTMAX : Sp / 16, ldt = tr_ld(Sp);
    bf16_t *Kimg = smem, *VT = Kimg + Sp * kRowLd;
    float *biasS = reinterpret_cast<float *>(VT + kDkPad * ldt);
    int b, hd;
    if (!att_slate_head(a.B, a.h, b, hd)) return;
    const int d = a.h * a.dk, bh = b * a.h + hd;          // bh: index of this (slate, head) in the dropout stream
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: scalar tile loops
    const bool vec = a.dk % 8 == 0;
    stage_head(a, b, hd, 1, Sp, Kimg, nullptr, 0);
    stage_head(a, b, hd, 2, Sp, nullptr, VT, ldt);
    for (int k = threadIdx.x; k < Sp; k += kAttThreads)
        biasS[k] = k >= a.S || (a.mask && a.mask[(long long)b * a.S + k] == 1) ? -INFINITY : 0.f;
    __syncthreads();
    const float c2 = 1.44269504088896341f / sqrtf((float)a.dk);
    const unsigned thr = drop_threshold(a.drop_p);
    const float ks = thr ? 1.f / (1.f - a.drop_p) : 1.f;
    for (int qt = w; qt * 16 < a.S; qt += kAttThreads / 64) {
        const int query = 16 * qt + j;
        const u32x4 qf = load8(a.qkv + ((long long)b * a.S + query) * 3 * d + hd * a.dk, 8 * g, a.dk, vec, query < a.S);
        f32x4 st[KTMAX];
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt)
            if (kt < KT) {
                st[kt] = mfma_bf16(row_frag(Kimg, kt, lane), qf, f32x4{0.f, 0.f, 0.f, 0.f});
                __builtin_amdgcn_sched_barrier(0);     // keep the fragment reads next to their MFMA (register pressure)
            }
        float lse2;
        softmax_tile<KTMAX>(st, KT, biasS, c2, g, lse2);
        if (a.lse && g == 0 && query < a.S) a.lse[(long long)bh * a.S + query] = lse2;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < KTMAX / 2; ++u)
            if (2 * u < KT) {
                unsigned pk[4];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4 p = st[2 * u + half];
                    if (thr) {
                        const Keep4 keep = drop_keep4b(a.seed, a.stream_id, attn_idx(bh, Sp, query, 32 * u + 16 * half + 4 * g), thr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[r] = keep.k[r] ? p[r] * ks : 0.f;
                    }
                    pk[2 * half] = pack_bf16(p[0], p[1]);
                    pk[2 * half + 1] = pack_bf16(p[2], p[3]);
                }
                const u32x4 pf = {pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma_bf16(col_frag(VT, ldt, dt, u, lane), pf, o[dt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (query < a.S) {
            bf16_t *row = a.out + ((long long)b * a.S + query) * d + hd * a.dk;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store4(row, 16 * dt + 4 * g, a.dk, a.dk % 4 == 0, o[dt]);
        }
    }
}

inline size_t att_fwd_lds(int S) {
    const int Sp = round_up(S, 32);
    return (size_t)(Sp * kRowLd + kDkPad * tr_ld(Sp)) * sizeof(bf16_t) + (size_t)Sp * 4;
}

// dot of two 8-element bf16 fragments
__device__ __forceinline__ float dot8(const u32x4 &x, const u32x4 &y) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += bf16_lo(x[k]) * bf16_lo(y[k]) + bf16_hi(x[k]) * bf16_hi(y[k]);
    return s;
}

// Backward.  With Pd = dropout(P), O = Pd V:  D_q = sum_k dP_qk P_qk = dO_q . O_q (O = the saved forward output), so no
// pass over the keys is needed for it.
// Phase A (query-major, like the forward): P^T recomputed, dS^T = P (dP - D) / sqrt(dk) -> dQ; lse2 and D to LDS.
// Phase B (key-major): P and dS recomputed as [query][key] tiles -> dV^T = dO^T Pd, dK^T = Q^T dS.
template <int KTMAX, bool FULL>
__global__ void __launch_bounds__(kAttThreads) attention_bwd_kernel(AttArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int Sp = round_up(a.S, 32), KT = FULL ? KTMAX : Sp / 16, ldt = tr_ld(Sp);
    // phase A images: K rows, V rows, K^T;  phase B images: Q rows, dO rows, Q^T, dO^T (same memory)
    bf16_t *img0 = smem, *img1 = img0 + Sp * kRowLd, *tr0 = img1 + Sp * kRowLd, *tr1 = tr0 + kDkPad * ldt;
    float *lseS = reinterpret_cast<float *>(tr1 + kDkPad * ldt), *DS = lseS + Sp, *biasS = DS + Sp;       // [Sp] each
    int b, hd;
    if (!att_slate_head(a.B, a.h, b, hd)) return;
    const int d = a.h * a.dk, bh = b * a.h + hd;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: scalar tile loops
    const bool vec = a.dk % 8 == 0;
    const float scale = 1.f / sqrtf((float)a.dk), c2 = 1.44269504088896341f * scale;
    const unsigned thr = drop_threshold(a.drop_p);
    const float ks = thr ? 1.f / (1.f - a.drop_p) : 1.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    stage_head(a, b, hd, 1, Sp, img0, tr0, ldt);
    stage_head(a, b, hd, 2, Sp, img1, nullptr, 0);
    for (int k = threadIdx.x; k < Sp; k += kAttThreads) {
        biasS[k] = k >= a.S || (a.mask && a.mask[(long long)b * a.S + k] == 1) ? -INFINITY : 0.f;
        lseS[k] = INFINITY;      // padded queries: p = 0 in phase B
        DS[k] = 0.f;
    }
    __syncthreads();
#ifndef LTR_ATT_SKIP_A      /* timing experiments only */
    for (int qt = w; qt * 16 < a.S; qt += kAttThreads / 64) {
        const int query = 16 * qt + j;
        const long long tok = (long long)b * a.S + query;
        const u32x4 qf = load8(a.qkv + tok * 3 * d + hd * a.dk, 8 * g, a.dk, vec, query < a.S);
        const u32x4 dof = load8(a.dctx + tok * d + hd * a.dk, 8 * g, a.dk, vec, query < a.S);
        const u32x4 of = load8(a.ctx + tok * d + hd * a.dk, 8 * g, a.dk, vec, query < a.S);
        const float D = quad_sum(dot8(dof, of));
        f32x4 st[KTMAX];
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt)
            if (kt < KT) {
                st[kt] = mfma_bf16(row_frag(img0, kt, lane), qf, zero);
                __builtin_amdgcn_sched_barrier(0);
            }
        float lse2;
        softmax_tile<KTMAX>(st, KT, biasS, c2, g, lse2);
        if (g == 0 && query < a.S) {
            lseS[query] = lse2;
            DS[query] = D;
        }
        f32x4 o[2] = {zero, zero};
#pragma unroll
        for (int u = 0; u < KTMAX / 2; ++u)
            if (2 * u < KT) {
                unsigned pk[4];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int kt = 2 * u + half;
                    const f32x4 dp = mfma_bf16(row_frag(img1, kt, lane), dof, zero);     // dPd^T = V dO^T
                    const Keep4 keep = thr ? drop_keep4b(a.seed, a.stream_id, attn_idx(bh, Sp, query, 16 * kt + 4 * g), thr) : keep_all();
                    f32x4 ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = st[kt][r] * ((keep.k[r] ? dp[r] * ks : 0.f) - D) * scale;
                    pk[2 * half] = pack_bf16(ds[0], ds[1]);
                    pk[2 * half + 1] = pack_bf16(ds[2], ds[3]);
                }
                const u32x4 dsf = {pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma_bf16(col_frag(tr0, ldt, dt, u, lane), dsf, o[dt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (query < a.S) {
            bf16_t *row = a.out + tok * 3 * d + hd * a.dk;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store4(row, 16 * dt + 4 * g, a.dk, a.dk % 4 == 0, o[dt]);
        }
    }
#endif
    __syncthreads();
    stage_head(a, b, hd, 0, Sp, img0, tr0, ldt);
    stage_head(a, b, hd, 3, Sp, img1, tr1, ldt);
    __syncthreads();
#ifndef LTR_ATT_SKIP_B
    for (int kt = w; kt * 16 < a.S; kt += kAttThreads / 64) {
        const int key = 16 * kt + j;
        const long long tok = (long long)b * a.S + key;
        const u32x4 kf = load8(a.qkv + tok * 3 * d + d + hd * a.dk, 8 * g, a.dk, vec, key < a.S);
        const u32x4 vf = load8(a.qkv + tok * 3 * d + 2 * d + hd * a.dk, 8 * g, a.dk, vec, key < a.S);
        const bool masked = biasS[key] != 0.f;
        f32x4 dv[2] = {zero, zero}, dkk[2] = {zero, zero};
        for (int u = 0; u < KT / 2; ++u) {
            unsigned pk[4], dk4[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * u + half;
                const f32x4 s = mfma_bf16(row_frag(img0, qt, lane), kf, zero);      // S[query 16 qt + 4 g + r][key]
                const f32x4 dpd = mfma_bf16(row_frag(img1, qt, lane), vf, zero);    // dPd[query][key]
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lseS + 16 * qt + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(DS + 16 * qt + 4 * g);
                f32x4 pd, ds;
                const Keep4 keep4 = thr ? drop_keep_col4b(a.seed, a.stream_id, attn_idx(bh, Sp, 16 * qt + 4 * g, key), (unsigned long long)Sp, thr, lane) : keep_all();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(s[r] * c2 - l4[r]);
                    const bool keep = keep4.k[r];
                    pd[r] = keep ? p * ks : 0.f;
                    ds[r] = p * ((keep ? dpd[r] * ks : 0.f) - d4[r]) * scale;
                }
                pk[2 * half] = pack_bf16(pd[0], pd[1]);
                pk[2 * half + 1] = pack_bf16(pd[2], pd[3]);
                dk4[2 * half] = pack_bf16(ds[0], ds[1]);
                dk4[2 * half + 1] = pack_bf16(ds[2], ds[3]);
            }
            const u32x4 pf = {pk[0], pk[1], pk[2], pk[3]}, dsf = {dk4[0], dk4[1], dk4[2], dk4[3]};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = mfma_bf16(col_frag(tr1, ldt, dt, u, lane), pf, dv[dt]);
                dkk[dt] = mfma_bf16(col_frag(tr0, ldt, dt, u, lane), dsf, dkk[dt]);
            }
        }
        if (key < a.S) {
            bf16_t *row = a.out + tok * 3 * d + hd * a.dk;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                store4(row + d, 16 * dt + 4 * g, a.dk, a.dk % 4 == 0, dkk[dt]);
                store4(row + 2 * d, 16 * dt + 4 * g, a.dk, a.dk % 4 == 0, dv[dt]);
            }
        }
    }
#endif
}

inline size_t att_bwd_lds(int S) {
    const int Sp = round_up(S, 32);
    return (size_t)(2 * Sp * kRowLd + 2 * kDkPad * tr_ld(Sp)) * sizeof(bf16_t) + (size_t)Sp * 12;
}

// ---- key-major backward in ONE pass (dk <= 16, S <= 256, the forward's lse2 saved) -------------------------------------
// The two-phase kernel above evaluates every probability twice (S^T tiles for dQ, S tiles for dK / dV): ~57 VALU
// instructions per (query, key) pair, which is what bounds it (SQ_INSTS_VALU 119 M per launch at config-5 shapes against
// 3.7 M MFMAs).  Here every pair is evaluated ONCE, key-major: wave w owns KPW = KTMAX / 4 CONTIGUOUS key tiles and walks
// all query tiles; per (query tile, key tile) it has S and dPd as [query 4g+r][key j] accumulator tiles, from which
//   p = 2^(c2 s - lse2),  Pd = keep ? p ks : 0,  dS = p (keep ? dPd ks : 0  -  D) / sqrt(dk)
// feed dV^T += dO^T Pd and dK^T += Q^T dS directly (contraction over queries: the accumulator layout IS the B operand),
// while dQ^T += K^T dS^T contracts over KEYS and needs dS with the query on lane & 15: each wave writes its bf16 dS tile
// [key][16 queries] (one ds_write_b64 per lane, 32-byte rows) and takes it back with ds_read_b64_tr_b16 as the B operand
// (same-wave LDS ordering, no barrier).  dQ is accumulated per wave over its own keys for all query tiles (registers) and
// the four partials are summed through LDS in wave order at the end: deterministic.
__device__ __forceinline__ u32x4 tr_frag_bf16(const bf16_t *lane_base, int r0) {
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4_t *lp4;
    const bf16_t *p = lane_base + r0 * 16;
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p)));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p + 16 * 16)));
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <int KTMAX, bool FULL>
__global__ void __launch_bounds__(kAttThreads, 2) attention_bwd_km_kernel(AttArgs a) {
    constexpr int KPW = KTMAX / 4;                       // key tiles per wave
    static_assert(KPW % 2 == 0, "32-key chunks");
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int Sp = round_up(a.S, 32), KT = FULL ? KTMAX : Sp / 16, ldt = tr_ld(Sp);
    bf16_t *Qr = smem, *dOr = Qr + Sp * kRowLd, *QT = dOr + Sp * kRowLd, *dOT = QT + 16 * ldt, *KTi = dOT + 16 * ldt;
    bf16_t *Tq = KTi + 16 * ldt;                                                   // [4 waves][16 KPW keys][16 queries]
    float *lseS = reinterpret_cast<float *>(Tq + 4 * 16 * KPW * 16), *DS = lseS + Sp, *biasS = DS + Sp;
    int b, hd;
    if (!att_slate_head(a.B, a.h, b, hd)) return;
    const int d = a.h * a.dk, bh = b * a.h + hd;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 3;        // wave-uniform (scalar branches below)
    const bool vec = a.dk % 8 == 0;
    const float scale = 1.f / sqrtf((float)a.dk), c2 = 1.44269504088896341f * scale;
    const unsigned thr = drop_threshold(a.drop_p);
    const float ks = thr ? 1.f / (1.f - a.drop_p) : 1.f, kss = ks * scale;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    stage_head(a, b, hd, 0, Sp, Qr, QT, ldt, 2);
    stage_head(a, b, hd, 3, Sp, dOr, dOT, ldt, 2);
    stage_head(a, b, hd, 1, Sp, nullptr, KTi, ldt, 2);
    for (int q = threadIdx.x; q < Sp; q += kAttThreads) {
        biasS[q] = q >= a.S || (a.mask && a.mask[(long long)b * a.S + q] == 1) ? -INFINITY : 0.f;
        float D = 0.f;
        if (q < a.S) {
            const long long tok = (long long)b * a.S + q;
            const bf16_t *dr = a.dctx + tok * d + hd * a.dk, *orow = a.ctx + tok * d + hd * a.dk;
            if (vec) {
                for (int x = 0; x < a.dk; x += 8)
                    D += dot8(*reinterpret_cast<const u32x4 *>(dr + x), *reinterpret_cast<const u32x4 *>(orow + x));
            } else {
                for (int x = 0; x < a.dk; ++x) D += from_bf16(dr[x]) * from_bf16(orow[x]);
            }
        }
        DS[q] = D * scale;
        lseS[q] = q < a.S ? a.lse[(long long)bh * a.S + q] : INFINITY;     // padded queries: p = 0
    }
    // this wave's keys: B operands of S = Q K^T and dPd = dO V^T straight from global memory
    u32x4 kf[KPW], vf[KPW];
    bool msk[KPW];
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        const int key = 16 * (w * KPW + c) + j;
        const long long tok = (long long)b * a.S + key;
        kf[c] = load8(a.qkv + tok * 3 * d + d + hd * a.dk, 8 * g, a.dk, vec, key < a.S);
        vf[c] = load8(a.qkv + tok * 3 * d + 2 * d + hd * a.dk, 8 * g, a.dk, vec, key < a.S);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        const int key = 16 * (w * KPW + c) + j;
        msk[c] = key < Sp ? biasS[key] != 0.f : true;
    }
    bf16_t *Tw = Tq + w * 16 * KPW * 16;
    const bf16_t *Tlane = Tw + (4 * g + (j >> 2)) * 16 + 4 * (j & 3);
    f32x4 dv[KPW], dkk[KPW], dq[KTMAX];
#pragma unroll
    for (int c = 0; c < KPW; ++c) dv[c] = dkk[c] = zero;
#pragma unroll
    for (int qt = 0; qt < KTMAX; ++qt) dq[qt] = zero;

#pragma unroll
    for (int u2 = 0; u2 < KTMAX / 2; ++u2)
        if (2 * u2 < KT) {
            unsigned pk[KPW][4], dsk[KPW][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * u2 + half;
                const u32x4 qa = row_frag(Qr, qt, lane), da = row_frag(dOr, qt, lane);
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lseS + 16 * qt + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(DS + 16 * qt + 4 * g);
#pragma unroll
                for (int c = 0; c < KPW; ++c) {       // key tiles past the slate run too (masked: p = 0) -- no per-tile branches
                        const f32x4 s = mfma_bf16(qa, kf[c], zero);          // S[query 16 qt + 4 g + r][key]
                        const f32x4 dpd = mfma_bf16(da, vf[c], zero);        // dPd[query][key]
                        const Keep4 keep4 = thr ? drop_keep_col4b(a.seed, a.stream_id, attn_idx(bh, Sp, 16 * qt + 4 * g, 16 * (w * KPW + c) + j),
                                                                  (unsigned long long)Sp, thr, lane) : keep_all();
                        f32x4 pd, ds;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = msk[c] ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c2, -l4[r]));
                            const bool keep = keep4.k[r];
                            pd[r] = keep ? p * ks : 0.f;
                            ds[r] = p * ((keep ? dpd[r] * kss : 0.f) - d4[r]);
                        }
                        pk[c][2 * half] = pack_bf16(pd[0], pd[1]);
                        pk[c][2 * half + 1] = pack_bf16(pd[2], pd[3]);
                        dsk[c][2 * half] = pack_bf16(ds[0], ds[1]);
                        dsk[c][2 * half + 1] = pack_bf16(ds[2], ds[3]);
                        *reinterpret_cast<u32x2 *>(Tw + (16 * c + j) * 16 + 4 * g) = u32x2{dsk[c][2 * half], dsk[c][2 * half + 1]};
                        __builtin_amdgcn_sched_barrier(0);     // one tile at a time (register pressure: the loops are fully unrolled)
                    }
                asm volatile("" ::: "memory");      // the tile stores stay in front of the transposing reads (same wave: in order)
                // dQ^T[d][query of tile qt] += K^T[d][this wave's keys] dS^T
#pragma unroll
                for (int uu = 0; uu < KPW / 2; ++uu)
                    if (FULL || w * KPW + 2 * uu < KT)
                        dq[qt] = mfma_bf16(col_frag(KTi, ldt, 0, (w * KPW) / 2 + uu, lane), tr_frag_bf16(Tlane, 32 * uu), dq[qt]);
                asm volatile("" ::: "memory");      // ... and in front of the next tile's stores
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < KPW; ++c) {
                    const u32x4 pf = {pk[c][0], pk[c][1], pk[c][2], pk[c][3]}, dsf = {dsk[c][0], dsk[c][1], dsk[c][2], dsk[c][3]};
                    dv[c] = mfma_bf16(col_frag(dOT, ldt, 0, u2, lane), pf, dv[c]);
                    dkk[c] = mfma_bf16(col_frag(QT, ldt, 0, u2, lane), dsf, dkk[c]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        const int key = 16 * (w * KPW + c) + j;
        if ((FULL || w * KPW + c < KT) && key < a.S) {
            bf16_t *row = a.out + ((long long)b * a.S + key) * 3 * d + hd * a.dk;
            store4(row + d, 4 * g, a.dk, a.dk % 4 == 0, dkk[c]);
            store4(row + 2 * d, 4 * g, a.dk, a.dk % 4 == 0, dv[c]);
        }
    }
    // dQ: the four waves' partials (disjoint key ranges) summed in wave order through LDS (the image memory, now free)
    __syncthreads();
    float *R = reinterpret_cast<float *>(smem);                                    // [4][Sp][16]
#pragma unroll
    for (int qt = 0; qt < KTMAX; ++qt)
        if (qt < KT) *reinterpret_cast<f32x4 *>(R + ((size_t)w * Sp + 16 * qt + j) * 16 + 4 * g) = dq[qt];
    __syncthreads();
    for (int e = threadIdx.x; e < Sp * 4; e += kAttThreads) {
        const int q = e >> 2, ch = e & 3;
        if (q >= a.S) continue;
        f32x4 v = *reinterpret_cast<const f32x4 *>(R + (size_t)q * 16 + 4 * ch);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) v += *reinterpret_cast<const f32x4 *>(R + ((size_t)ww * Sp + q) * 16 + 4 * ch);
        store4(a.out + ((long long)b * a.S + q) * 3 * d + hd * a.dk, 4 * ch, a.dk, a.dk % 4 == 0, v);
    }
}

inline size_t att_bwd_km_lds(int S) {
    const int Sp = round_up(S, 32), KPW = (Sp <= 128 ? 8 : 16) / 4;
    return (size_t)(2 * Sp * kRowLd + 3 * 16 * tr_ld(Sp) + 4 * 16 * KPW * 16) * sizeof(bf16_t) + (size_t)Sp * 12;
}

// p_attn of transformer.py:161-163 as a tensor (what `attention()` RETURNS next to its output; nothing on the training path
// reads it): probs [B][h][S][S] fp32 = dropout(softmax(q k^T / sqrt(dk) + mask)), the same numbers attention_fwd_kernel
// feeds into P.V before its bf16 rounding.  One wave per (slate-head, query); lanes over keys.  Not a hot kernel.
__global__ void __launch_bounds__(256) attn_probs_kernel(AttArgs a, float *__restrict__ probs) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)a.B * a.h * a.S) return;
    const int query = (int)(row % a.S), bh = (int)(row / a.S), b = bh / a.h, hd = bh % a.h;
    const int d = a.h * a.dk, Sp = round_up(a.S, 32);
    const bf16_t *base = a.qkv + (long long)b * a.S * 3 * d + hd * a.dk;
    const bf16_t *qrow = base + (long long)query * 3 * d;
    const float c2 = 1.44269504088896341f / sqrtf((float)a.dk);
    float sc[8];                                             // S <= 512: 8 keys per lane
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = lane + 64 * t;
        float v = -INFINITY;
        if (k < a.S && !(a.mask && a.mask[(long long)b * a.S + k] == 1)) {
            const bf16_t *krow = base + (long long)k * 3 * d + d;
            float acc = 0.f;
            for (int e = 0; e < a.dk; ++e) acc = fmaf(from_bf16(qrow[e]), from_bf16(krow[e]), acc);
            v = acc * c2;
        }
        sc[t] = v;
        m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (m == -INFINITY) m = 0.f;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        sc[t] = __builtin_amdgcn_exp2f(sc[t] - m);
        l += sc[t];
    }
    l = wave_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const unsigned thr = drop_threshold(a.drop_p);
    const float ks = thr ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = lane + 64 * t;
        if (k < a.S) {
            float pv = sc[t] * inv;
            if (thr) pv = drop_keep(a.seed, a.stream_id, attn_idx(bh, Sp, query, k), thr) ? pv * ks : 0.f;
            probs[row * a.S + k] = pv;
        }
    }
}

// one instantiation (and one set of per-device attribute flags) per kernel variant
template <int KTMAX, bool FULL, bool BWD>
int launch_att_tagged(const AttArgs &a, size_t lds, hipStream_t stream) {
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    auto kernel = BWD ? attention_bwd_kernel<KTMAX, FULL> : attention_fwd_kernel<KTMAX, FULL>;
    if (dev < 0 || !done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) done[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(att_grid(a.B, a.h)), dim3(kAttThreads), lds, stream, a);
    return status();
}
template <int KTMAX, bool FULL>
int launch_att_km(const AttArgs &a, hipStream_t stream) {
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    auto kernel = attention_bwd_km_kernel<KTMAX, FULL>;
    if (dev < 0 || !done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) done[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(att_grid(a.B, a.h)), dim3(kAttThreads), att_bwd_km_lds(a.S), stream, a);
    return status();
}
// the one-pass key-major backward: head dimension within one 16-row image, slate within 16 key tiles, lse2 saved
inline bool att_km_ok(const AttArgs &a) { return a.lse != nullptr && a.dk <= 16 && a.S <= 256; }
int dispatch_att_km(const AttArgs &a, hipStream_t stream) {
    const int Sp = round_up(a.S, 32);
    if (Sp <= 128) return Sp == 128 ? launch_att_km<8, true>(a, stream) : launch_att_km<8, false>(a, stream);
    return Sp == 256 ? launch_att_km<16, true>(a, stream) : launch_att_km<16, false>(a, stream);
}
template <bool BWD>
int dispatch_att(const AttArgs &a, size_t lds, hipStream_t stream) {
    const int Sp = round_up(a.S, 32);
    if (Sp <= 128) return Sp == 128 ? launch_att_tagged<8, true, BWD>(a, lds, stream) : launch_att_tagged<8, false, BWD>(a, lds, stream);
    if (Sp <= 256) return Sp == 256 ? launch_att_tagged<16, true, BWD>(a, lds, stream) : launch_att_tagged<16, false, BWD>(a, lds, stream);
    return Sp == 512 ? launch_att_tagged<32, true, BWD>(a, lds, stream) : launch_att_tagged<32, false, BWD>(a, lds, stream);
}

int check_att(const uint16_t *qkv, const void *other, int B, int S, int h, int dk, float p) {
    if (!qkv || !other) return LTR_ERR_NULL;
    if (B < 0 || S < 1 || S > 512 || h < 1 || dk < 1 || dk > kDkPad) return LTR_ERR_SHAPE;
    if ((h * dk) % 8 != 0) return LTR_ERR_SHAPE;
    if (!(p >= 0.f) || p >= 1.f) return LTR_ERR_PARAM;
    if (((uintptr_t)qkv & 15u) || ((uintptr_t)other & 15u)) return LTR_ERR_ALIGN;
    return 0;
}

inline int elt_grid(int64_t n, int threads, int cap = 256 * 8) {
    int64_t b = (n + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int ffn_check(const void *n2, const void *w1, const void *b1, const void *w2, int64_t T, int d, int dff, float p) {
    if (!n2 || !w1 || !b1 || !w2) return LTR_ERR_NULL;
    if (T < 0 || (d != 64 && d != 128) || dff < kFfnChunk || dff % kFfnChunk) return LTR_ERR_SHAPE;
    if (!(p >= 0.f) || p >= 1.f) return LTR_ERR_PARAM;
    if (((uintptr_t)n2 & 15u) || ((uintptr_t)w1 & 15u) || ((uintptr_t)w2 & 15u) || ((uintptr_t)b1 & 15u)) return LTR_ERR_ALIGN;
    return 0;
}
template <class K>
int ffn_launch(K kernel, bool (&done)[64], dim3 grid, size_t lds, const FfnArgs &a, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) done[dev] = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(kFfnThreads), lds, stream, a);
    return status();
}

}  // namespace

extern "C" {

int ltr_enc_cast_bf16(const float *src, uint16_t *dst, int64_t n, void *stream) {
    if (!src || !dst) return LTR_ERR_NULL;
    if (n < 0) return LTR_ERR_SHAPE;
    if (n == 0) return LTR_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(elt_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n);
    return status();
}

int ltr_enc_seed_set(uint64_t value, void *stream) {
    hipLaunchKernelGGL(seed_epoch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long)value, 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ltr_enc_seed_advance(uint64_t delta, void *stream) {
    hipLaunchKernelGGL(seed_epoch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long)delta, 1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ltr_enc_seed_get(uint64_t *value) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_drop_epoch), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    *value = (uint64_t)v;
    return 0;
}
int ltr_enc_dropout_mask(uint64_t seed, int stream_id, int64_t n, float p, uint8_t *out, void *stream) {
    if (!out) return LTR_ERR_NULL;
    if (n < 0) return LTR_ERR_SHAPE;
    if (!(p >= 0.f) || p >= 1.f) return LTR_ERR_PARAM;
    if (n == 0) return LTR_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(elt_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed,
                       stream_id, (long long)n, drop_threshold(p), out);
    return status();
}

int ltr_enc_attn_dropout_mask(uint64_t seed, int stream_id, int B, int S, int h, float p, uint8_t *out, void *stream) {
    if (!out) return LTR_ERR_NULL;
    if (B < 0 || S < 1 || h < 1) return LTR_ERR_SHAPE;
    if (!(p >= 0.f) || p >= 1.f) return LTR_ERR_PARAM;
    if (B == 0) return LTR_OK;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(elt_grid((int64_t)B * h * S * S, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)seed, stream_id, B * h, S, drop_threshold(p), out);
    return status();
}

int ltr_enc_sum_partials(const float *parts, int nsplit, int64_t n, int accumulate, float *out, void *stream) {
    if (!parts || !out) return LTR_ERR_NULL;
    if (nsplit < 1 || n < 0) return LTR_ERR_SHAPE;
    if (n == 0) return LTR_OK;
    const int ew = 256 / reduce_zl(nsplit, n);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(elt_grid(n, ew)), dim3(256), 0, (hipStream_t)stream, parts, nsplit, (long long)n,
                       accumulate, out);
    return status();
}

int ltr_enc_splitk_epilogue(const float *parts, int nsplit, int64_t M, int N, const float *bias, float drop_p, uint64_t seed, int stream_id,
                            const float *residual, float *out, void *stream) {
    if (!parts || !out) return LTR_ERR_NULL;
    if (nsplit < 1 || M < 0 || N < 4 || N % 4) return LTR_ERR_SHAPE;
    if (!(drop_p >= 0.f) || !(drop_p < 1.f)) return LTR_ERR_PARAM;
    if (M == 0) return LTR_OK;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(elt_grid(M * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream, parts, nsplit,
                       (long long)M, N, bias, drop_p, (unsigned long long)seed, stream_id, residual, out);
    return status();
}

int ltr_enc_sum_partials_batch(const ltr_reduce_job *jobs, int njobs, void *stream) {
    if (!jobs) return LTR_ERR_NULL;
    if (njobs < 0) return LTR_ERR_SHAPE;
    for (int j0 = 0; j0 < njobs; j0 += kReduceJobs) {
        ReduceJobs r{};
        int cnt = 0, maxb = 1;
        for (int k = j0; k < njobs && cnt < kReduceJobs; ++k) {
            const ltr_reduce_job &q = jobs[k];
            if (!q.parts || !q.out) return LTR_ERR_NULL;
            if (q.nsplit < 1 || q.n < 0 || (q.stride != 0 && q.stride < q.n)) return LTR_ERR_SHAPE;
            if (q.n == 0) continue;
            r.parts[cnt] = q.parts;
            r.out[cnt] = q.out;
            r.n[cnt] = q.n;
            r.stride[cnt] = q.stride > 0 ? q.stride : q.n;
            r.nsplit[cnt] = q.nsplit;
            r.blocks[cnt] = elt_grid(q.n, 256 / reduce_zl(q.nsplit, q.n));
            maxb = r.blocks[cnt] > maxb ? r.blocks[cnt] : maxb;
            ++cnt;
        }
        if (!cnt) continue;
        hipLaunchKernelGGL(sum_partials_batch_kernel, dim3(maxb, cnt), dim3(256), 0, (hipStream_t)stream, r);
        if (int rc = status()) return rc;
    }
    return LTR_OK;
}

int ltr_enc_layernorm_fwd(const float *x, const float *a, const float *b, int64_t T, int d, float eps, int standard,
                          uint16_t *y_bf16, float *y_f32, void *stream) {
    if (!x || !a || !b || (!y_bf16 && !y_f32)) return LTR_ERR_NULL;
    if (T < 0 || d < 2 || d > 64 * kLnMax) return LTR_ERR_SHAPE;
    if (T == 0) return LTR_OK;
    const bool v4 = d % 4 == 0 && !(((uintptr_t)x | (uintptr_t)a | (uintptr_t)b | (uintptr_t)y_f32) & 15u) && !((uintptr_t)y_bf16 & 7u);
    if (v4 && d <= 128)
        hipLaunchKernelGGL(layernorm_fwd_v4_kernel<32>, dim3(elt_grid(T, 8, 256 * 16)), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, b,
                           (long long)T, d, eps, standard, y_bf16, y_f32);
    else if (v4 && d <= 256)
        hipLaunchKernelGGL(layernorm_fwd_v4_kernel<64>, dim3(elt_grid(T, 4, 256 * 16)), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, b,
                           (long long)T, d, eps, standard, y_bf16, y_f32);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(elt_grid(T, 4)), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, b, (long long)T, d,
                           eps, standard, y_bf16, y_f32);
    return status();
}

int ltr_enc_layernorm_bwd(const float *x, const float *a, const float *dy, int64_t T, int d, float eps, int standard,
                          float *dx, float *partials, int nblk, void *stream) {
    if (!x || !a || !dy || !dx || !partials) return LTR_ERR_NULL;
    if (T < 0 || d < 2 || d > 64 * kLnMax || nblk < 1 || nblk > 65535) return LTR_ERR_SHAPE;
    const bool v4 = d % 4 == 0 && !(((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)a) & 15u);
    if (v4 && d <= 128)
        hipLaunchKernelGGL(layernorm_bwd_v4_kernel<32>, dim3(nblk), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, dy, (long long)T, d, eps,
                           standard, dx, partials);
    else if (v4 && d <= 256)
        hipLaunchKernelGGL(layernorm_bwd_v4_kernel<64>, dim3(nblk), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, dy, (long long)T, d, eps,
                           standard, dx, partials);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, dy, (long long)T, d, eps,
                           standard, dx, partials);
    return status();
}

int ltr_enc_gemm_bf16(const ltr_gemm_desc *desc, void *stream) {
    if (!desc || !desc->A || !desc->B || (!desc->Cf && !desc->Cb)) return LTR_ERR_NULL;
    const ltr_gemm_desc &g = *desc;
    if (g.M < 0 || g.N < 1 || g.K < 1 || g.splits < 1 || g.splits > 1024) return LTR_ERR_SHAPE;
    if (g.M == 0) return LTR_OK;
    if (g.N % 8 || g.lda % 8 || g.ldb % 8 || g.ldc % 4 || g.ldc < g.N) return LTR_ERR_SHAPE;
    if (!g.a_kmajor && g.K % 8) return LTR_ERR_SHAPE;
    if (!g.b_kmajor && g.K % 8) return LTR_ERR_SHAPE;
    if (g.a_kmajor && g.M % 8) return LTR_ERR_SHAPE;
    if (g.splits > 1 && !g.Cf) return LTR_ERR_NULL;
    if (!(g.drop_p >= 0.f) || g.drop_p >= 1.f) return LTR_ERR_PARAM;
    if (((uintptr_t)g.A & 15u) || ((uintptr_t)g.B & 15u) || ((uintptr_t)g.Cf & 15u) || ((uintptr_t)g.Cb & 7u) ||
        ((uintptr_t)g.bias & 15u) || ((uintptr_t)g.residual & 15u) || ((uintptr_t)g.gate & 7u))
        return LTR_ERR_ALIGN;
    if ((g.M + BM - 1) / BM > 65535) return LTR_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (g.a_kmajor) return g.b_kmajor ? launch_gemm<true, true>(g, s) : launch_gemm<true, false>(g, s);
    return g.b_kmajor ? launch_gemm<false, true>(g, s) : launch_gemm<false, false>(g, s);
}

int ltr_enc_colsum_bf16(const uint16_t *y, int64_t T, int N, float *partials, int nblk, void *stream) {
    if (!y || !partials) return LTR_ERR_NULL;
    if (T < 0 || N < 8 || N % 8 || nblk < 1) return LTR_ERR_SHAPE;
    if ((uintptr_t)y & 15u) return LTR_ERR_ALIGN;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, (long long)T, N, partials);
    return status();
}

int ltr_enc_drop_cast_colsum(const float *dx, int64_t T, int N, float p, uint64_t seed, int stream_id, uint16_t *out,
                             float *partials, int nblk, void *stream) {
    if (!dx || !out || !partials) return LTR_ERR_NULL;
    if (T < 0 || N < 8 || N % 8 || nblk < 1) return LTR_ERR_SHAPE;
    if (!(p >= 0.f) || p >= 1.f) return LTR_ERR_PARAM;
    if (((uintptr_t)dx & 15u) || ((uintptr_t)out & 7u)) return LTR_ERR_ALIGN;
    const unsigned thr = drop_threshold(p);
    hipLaunchKernelGGL(drop_cast_colsum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dx, (long long)T, N, thr,
                       thr ? 1.f / (1.f - p) : 1.f, (unsigned long long)seed, stream_id, out, partials);
    return status();
}

int ltr_enc_attention_fwd_lse(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p,
                              uint64_t seed, int stream_id, uint16_t *ctx, float *lse, void *stream) {
    if (int rc = check_att(qkv, ctx, B, S, h, dk, drop_p)) return rc;
    if (B == 0) return LTR_OK;
    AttArgs a{qkv, nullptr, nullptr, mask, ctx, B, S, h, dk, drop_p, (unsigned long long)seed, stream_id, lse};
    return dispatch_att<false>(a, att_fwd_lds(S), (hipStream_t)stream);
}
int ltr_enc_attention_fwd(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p,
                          uint64_t seed, int stream_id, uint16_t *ctx, void *stream) {
    return ltr_enc_attention_fwd_lse(qkv, mask, B, S, h, dk, drop_p, seed, stream_id, ctx, nullptr, stream);
}

int ltr_enc_attention_bwd_lse(const uint16_t *qkv, const uint16_t *ctx, const uint16_t *dctx, const float *lse, const uint8_t *mask,
                              int B, int S, int h, int dk, float drop_p, uint64_t seed, int stream_id, uint16_t *dqkv, void *stream) {
    if (!dctx || !ctx) return LTR_ERR_NULL;
    if (int rc = check_att(qkv, dqkv, B, S, h, dk, drop_p)) return rc;
    if (((uintptr_t)dctx & 15u) || ((uintptr_t)ctx & 15u)) return LTR_ERR_ALIGN;
    if (B == 0) return LTR_OK;
    AttArgs a{qkv, dctx, ctx, mask, dqkv, B, S, h, dk, drop_p, (unsigned long long)seed, stream_id, const_cast<float *>(lse)};
    if (att_km_ok(a)) return dispatch_att_km(a, (hipStream_t)stream);
    return dispatch_att<true>(a, att_bwd_lds(S), (hipStream_t)stream);
}
int ltr_enc_attention_bwd(const uint16_t *qkv, const uint16_t *ctx, const uint16_t *dctx, const uint8_t *mask, int B, int S, int h,
                          int dk, float drop_p, uint64_t seed, int stream_id, uint16_t *dqkv, void *stream) {
    return ltr_enc_attention_bwd_lse(qkv, ctx, dctx, nullptr, mask, B, S, h, dk, drop_p, seed, stream_id, dqkv, stream);
}

int ltr_enc_attention_probs(const uint16_t *qkv, const uint8_t *mask, int B, int S, int h, int dk, float drop_p, uint64_t seed,
                            int stream_id, float *probs, void *stream) {
    if (int rc = check_att(qkv, probs, B, S, h, dk, drop_p)) return rc;
    if (B == 0) return LTR_OK;
    AttArgs a{qkv, nullptr, nullptr, mask, nullptr, B, S, h, dk, drop_p, (unsigned long long)seed, stream_id, nullptr};
    const long long rows = (long long)B * h * S;
    hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, probs);
    return status();
}

int ltr_enc_ffn_supported(int d, int dff) { return (d == 64 || d == 128) && dff >= kFfnChunk && dff % kFfnChunk == 0; }

int ltr_enc_ffn_fwd(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const float *b2, const float *x1,
                    int64_t T, int d, int dff, float drop_p, uint64_t seed, int stream_hidden, int stream_out, float *x2, void *stream) {
    if (int rc = ffn_check(n2, w1, b1, w2, T, d, dff, drop_p)) return rc;
    if (!b2 || !x1 || !x2) return LTR_ERR_NULL;
    if (T == 0) return LTR_OK;
    FfnArgs a{n2, w1, w2, nullptr, b1, b2, x1, (long long)T, dff, drop_p, (unsigned long long)seed, stream_hidden, stream_out, x2,
              nullptr, nullptr, nullptr, 0};
    // 256 tokens per workgroup once that fills the chip; 128 below (twice the workgroups, each streams the weights once)
    const bool small = T < (int64_t)256 * kFfnTok;
    const dim3 grid((unsigned)(small ? (T + 127) / 128 : (T + kFfnTok - 1) / kFfnTok));
    static bool d64[64] = {}, d128[64] = {}, s64[64] = {}, s128[64] = {};
    const size_t l64 = 2 * (kFfnChunk * (64 + 8) + 64 * (kFfnChunk + 8)) * sizeof(bf16_t), l128 = 2 * 2 * 128 * 128 * sizeof(bf16_t);
    if (d == 64) return small ? ffn_launch(ffn_fwd_kernel<64, 1>, s64, grid, l64, a, (hipStream_t)stream)
                              : ffn_launch(ffn_fwd_kernel<64, 2>, d64, grid, l64, a, (hipStream_t)stream);
    return small ? ffn_launch(ffn_fwd_kernel<128, 1>, s128, grid, l128, a, (hipStream_t)stream)
                 : ffn_launch(ffn_fwd_kernel<128, 2>, d128, grid, l128, a, (hipStream_t)stream);
}

int ltr_enc_ffn_bwd_x(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const uint16_t *dy, int64_t T, int d,
                      int dff, float drop_p, uint64_t seed, int stream_hidden, float *dn2, void *stream) {
    if (int rc = ffn_check(n2, w1, b1, w2, T, d, dff, drop_p)) return rc;
    if (!dy || !dn2) return LTR_ERR_NULL;
    if ((uintptr_t)dy & 15u) return LTR_ERR_ALIGN;
    if (T == 0) return LTR_OK;
    FfnArgs a{n2, w1, w2, dy, b1, nullptr, nullptr, (long long)T, dff, drop_p, (unsigned long long)seed, stream_hidden, 0, dn2,
              nullptr, nullptr, nullptr, 0};
    const bool small = T < (int64_t)256 * kFfnTok;
    const dim3 grid((unsigned)(small ? (T + 127) / 128 : (T + kFfnTok - 1) / kFfnTok));
    static bool d64[64] = {}, d128[64] = {}, s64[64] = {}, s128[64] = {};
    const size_t l64 = 2 * (kFfnChunk * (64 + 16) + 64 * (kFfnChunk + 16)) * sizeof(bf16_t), l128 = 2 * 2 * 128 * 128 * sizeof(bf16_t);
    if (d == 64) return small ? ffn_launch(ffn_bwd_x_kernel<64, 1>, s64, grid, l64, a, (hipStream_t)stream)
                              : ffn_launch(ffn_bwd_x_kernel<64, 2>, d64, grid, l64, a, (hipStream_t)stream);
    return small ? ffn_launch(ffn_bwd_x_kernel<128, 1>, s128, grid, l128, a, (hipStream_t)stream)
                 : ffn_launch(ffn_bwd_x_kernel<128, 2>, d128, grid, l128, a, (hipStream_t)stream);
}

int ltr_enc_ffn_bwd_w(const uint16_t *n2, const uint16_t *w1, const float *b1, const uint16_t *w2, const uint16_t *dy, int64_t T, int d,
                      int dff, float drop_p, uint64_t seed, int stream_hidden, int nsplit, float *dw1_parts, float *dw2_parts,
                      float *db1_parts, void *stream) {
    if (int rc = ffn_check(n2, w1, b1, w2, T, d, dff, drop_p)) return rc;
    if (!dy || !dw1_parts || !dw2_parts || !db1_parts) return LTR_ERR_NULL;
    if ((uintptr_t)dy & 15u) return LTR_ERR_ALIGN;
    if (nsplit < 1 || nsplit > 1024) return LTR_ERR_SHAPE;
    FfnArgs a{n2, w1, w2, dy, b1, nullptr, nullptr, (long long)T, dff, drop_p, (unsigned long long)seed, stream_hidden, 0, nullptr,
              dw1_parts, dw2_parts, db1_parts, nsplit};
    const dim3 grid((unsigned)((nsplit + 7) / 8 * 8 * (dff / kFfnChunk)));
    static bool d64[64] = {}, d128[64] = {};
    if (d == 64) return ffn_launch(ffn_bwd_w_kernel<64>, d64, grid, 2 * 2 * 128 * (64 + 16) * sizeof(bf16_t), a, (hipStream_t)stream);
    return ffn_launch(ffn_bwd_w_kernel<128>, d128, grid, (2 * 128 * (128 + 16) + 2 * 128 * 128) * sizeof(bf16_t), a, (hipStream_t)stream);
}

int ltr_enc_score_fwd(const float *x, const float *a, const float *b, const float *w, const float *bias, int64_t T, int d,
                      float eps, int norm, float *scores, void *stream) {
    if (!x || !w || !bias || !scores || (norm && (!a || !b))) return LTR_ERR_NULL;
    if (T < 0 || d < 2 || d > 64 * kLnMax) return LTR_ERR_SHAPE;
    if (norm < 0 || norm > 2) return LTR_ERR_PARAM;
    if (T == 0) return LTR_OK;
    hipLaunchKernelGGL(score_fwd_kernel, dim3(elt_grid(T, 4)), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, b, w, bias, (long long)T,
                       d, eps, norm, scores);
    return status();
}

int ltr_enc_score_bwd(const float *x, const float *a, const float *b, const float *w, const float *dscores, int64_t T, int d,
                      float eps, int norm, float *dx, float *partials, int nblk, void *stream) {
    if (!x || !w || !dscores || !dx || !partials || (norm && (!a || !b))) return LTR_ERR_NULL;
    if (T < 0 || d < 2 || d > 64 * kLnMax || nblk < 1 || nblk > 65535) return LTR_ERR_SHAPE;
    if (norm < 0 || norm > 2) return LTR_ERR_PARAM;
    hipLaunchKernelGGL(score_bwd_kernel, dim3(nblk), dim3(kLnThreads), 0, (hipStream_t)stream, x, a, b, w, dscores, (long long)T, d, eps,
                       norm, dx, partials);
    return status();
}

}  // extern "C"
