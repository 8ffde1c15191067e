// ltr_device.h -- device-side building blocks shared by the gfx950 listwise-LTR kernels.
//
// A *slate group* is the set of threads (64..1024, power of two) that cooperates on one slate whose
// state (scores, labels, gains, ...) is staged in LDS.  Rows i of the S x S pair matrix are spread over
// `sp` row lanes, and when the group has more threads than rows the columns j are split over
// CG = group / sp column groups (adjacent lanes of one wave); partial row sums are combined by symmetric DPP
// butterflies and cross-wave sums in a FIXED order through LDS, so every result is bit-reproducible (no float
// atomics on shared accumulators anywhere in this library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LTR_WAVE 64

// Launch grids.  A dispatch's grid is counted in WORK-ITEMS per dimension in 32 bits: blocks x threads-per-block >= 2^32 in x is
// truncated silently (17 M slates x 256 threads already pass it -- found by tests/test_large_shard_gpu.py).  Kernels whose block count
// scales with the batch take their linear block id from ltr_block_id() and are launched with ltr_grid(blocks): x <= 2^20 blocks
// (x 1024 threads = 2^30 work-items), the rest in y; blocks past the end see an id >= the count and must idle through.
#define LTR_GRID_X_MAX (1ll << 20)
__device__ __forceinline__ long long ltr_block_id() { return (long long)blockIdx.y * gridDim.x + blockIdx.x; }
inline dim3 ltr_grid(long long blocks) {
    if (blocks <= LTR_GRID_X_MAX) return dim3((unsigned)(blocks > 0 ? blocks : 1));
    return dim3((unsigned)LTR_GRID_X_MAX, (unsigned)((blocks + LTR_GRID_X_MAX - 1) / LTR_GRID_X_MAX));
}
#define LTR_LN2 0.69314718055994530942f

namespace ltr {

// 1 / x on the hardware reciprocal (v_rcp_f32, 1 ulp).  NOT __frcp_rn: without fast-math hipcc expands that to the correctly
// rounded IEEE division sequence (v_div_scale, v_rcp, four fma, v_div_fmas, v_div_fixup: an 11-instruction dependent chain) --
// which is what sat in the inner loop of every pair sweep until round 3 (profiles/r03_variant_ab.json).
__device__ __forceinline__ float ltr_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct SlateGroup {
    int S;      // documents in the slate
    int group;  // threads cooperating on the slate
    int sp;     // row lanes  = min(nextpow2(S), group)
    int CG;     // column groups = group / sp
    int t;      // thread index inside the group
    int ri;     // row lane   = t / CG
    int cg;     // column grp = t % CG
    int wig;    // wave index inside the group
    int nw;     // waves per group
    float *part;  // LDS [group]  row-partial scratch
    float *red;   // LDS [32]     cross-wave scratch
};

#ifndef LTR_DPP_REDUCE
#define LTR_DPP_REDUCE 1
#endif

// Cross-lane moves inside a 16-lane DPP row (VALU, no LDS crossbar): quad xor-1, quad xor-2, half-row mirror
// (i <-> 7-i), row mirror (i <-> 15-i).  Each step pairs lanes SYMMETRICALLY, so after the four steps all 16
// lanes of a row hold bit-identical results for any commutative op.
#define LTR_DPP(x, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, false))
#define LTR_DPP_XOR1 0xB1
#define LTR_DPP_XOR2 0x4E
#define LTR_DPP_HALF_MIRROR 0x141
#define LTR_DPP_MIRROR 0x140

__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// All-lanes wave reductions; every lane ends with the same bits (replicated control flow depends on it).
__device__ __forceinline__ float wave_allsum(float v) {
#if LTR_DPP_REDUCE
    v += LTR_DPP(v, LTR_DPP_XOR1);
    v += LTR_DPP(v, LTR_DPP_XOR2);
    v += LTR_DPP(v, LTR_DPP_HALF_MIRROR);
    v += LTR_DPP(v, LTR_DPP_MIRROR);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
#else
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, LTR_WAVE);
    return v;  // butterfly: every lane holds the same bits
#endif
}

__device__ __forceinline__ float wave_allmax(float v) {
#if LTR_DPP_REDUCE
    v = fmaxf(v, LTR_DPP(v, LTR_DPP_XOR1));
    v = fmaxf(v, LTR_DPP(v, LTR_DPP_XOR2));
    v = fmaxf(v, LTR_DPP(v, LTR_DPP_HALF_MIRROR));
    v = fmaxf(v, LTR_DPP(v, LTR_DPP_MIRROR));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
#else
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, LTR_WAVE));
    return v;
#endif
}

__device__ __forceinline__ float wave_allmin(float v) { return -wave_allmax(-v); }

// Sum over all threads of the slate group; every thread gets the total.  All threads of the BLOCK must
// call it (uniform __syncthreads count).
__device__ __forceinline__ float group_sum(const SlateGroup &g, float v) {
    v = wave_allsum(v);
    if (g.nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) g.red[g.wig] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < g.nw; ++w) s += g.red[w];
    return s;
}

__device__ __forceinline__ float group_max(const SlateGroup &g, float v) {
    v = wave_allmax(v);
    if (g.nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & (LTR_WAVE - 1)) == 0) g.red[g.wig] = v;
    __syncthreads();
    float s = g.red[0];
    for (int w = 1; w < g.nw; ++w) s = fmaxf(s, g.red[w]);
    return s;
}

// Combine the CG column-group partials of each row: returns sum_c v(row, c) to every replica of the row.
// The CG replicas of a row are ADJACENT lanes of one wave (t = ri*CG + cg, CG a power of two <= 64), so this
// is a log2(CG)-step butterfly: no LDS, no barrier, same bits in every replica.
__device__ __forceinline__ float row_reduce(const SlateGroup &g, float v) {
#if LTR_DPP_REDUCE
    if (g.CG <= 16) {   // group-uniform
        if (g.CG >= 2) v += LTR_DPP(v, LTR_DPP_XOR1);
        if (g.CG >= 4) v += LTR_DPP(v, LTR_DPP_XOR2);
        if (g.CG >= 8) v += LTR_DPP(v, LTR_DPP_HALF_MIRROR);
        if (g.CG >= 16) v += LTR_DPP(v, LTR_DPP_MIRROR);
        return v;
    }
#endif
    for (int o = g.CG >> 1; o >= 1; o >>= 1) v += __shfl_xor(v, o, LTR_WAVE);
    return v;
}

__host__ __device__ inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Threads per slate: two threads per row when possible (halves the serial j sweep), one wave minimum.
inline int pick_group(int S) {
    int g = next_pow2(2 * S);
    if (g < 64) g = 64;
    if (g > 1024) g = 1024;
    return g;
}

// `tid`: the thread's index in the block.  Persistent kernels pass a value laundered through an empty asm inside their
// tile loop so that the group geometry is recomputed per tile (a dozen integer ops) instead of being hoisted out of the
// loop and kept alive -- in a kernel at its register limit those hoisted values end up in scratch.
__device__ __forceinline__ SlateGroup make_group(int S, int group, float *scratch /* [group + 32] */, int tid) {
    SlateGroup g;
    g.S = S;
    g.group = group;
    int np2 = next_pow2(S);
    g.sp = np2 < group ? np2 : group;
    g.CG = group / g.sp;
    int gid = tid / group;
    g.t = tid - gid * group;
    g.ri = g.t / g.CG;          // row lane; its CG column groups are adjacent lanes (see row_reduce)
    g.cg = g.t & (g.CG - 1);
    g.wig = g.t / LTR_WAVE;
    g.nw = group / LTR_WAVE;
    g.part = scratch;
    g.red = scratch + group;
    return g;
}

__device__ __forceinline__ SlateGroup make_group(int S, int group, float *scratch) {
    return make_group(S, group, scratch, (int)threadIdx.x);
}

// sigmoid pair of x: big = sigma(|x|), small = sigma(-|x|), without overflow and with full relative
// precision on the small side (small = e * big, e = exp(-|x|)).
__device__ __forceinline__ void sigmoid_pair(float x, float &s_pos, float &s_neg) {
    float e = __expf(-fabsf(x));
    float r = ltr_rcp(1.f + e);
    float big = r, small = e * r;
    bool nonneg = x >= 0.f;
    s_pos = nonneg ? big : small;   // sigmoid(x)
    s_neg = nonneg ? small : big;   // sigmoid(-x)
}

}  // namespace ltr
