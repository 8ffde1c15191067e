// ltr_scorer.hip -- the slate pipeline: FC scorer forward -> listwise loss -> scorer backward -> weight
// gradients, one persistent workgroup per CU, X read from HBM exactly once (gfx950 / MI355X).
//
// Replaces, behind the reference's nn.Module surface:
//   architeture/doubleLayer.py:54-73  DoubleLayerNet  fc3(drop(relu(fc2(drop(relu(fc1 x))))))
//   architeture/tripleLayer.py:5-17   TripleLayerNet  l3(sigmoid(l2(l1 x)))
// and, in fused mode, the scorer->loss->backward chain of main_batch_execution.py:128-170.
//
// Layout of the computation (see DESIGN.md "slate pipeline"):
//   * workgroup = 8 waves (two per SIMD, 256 VGPRs each); a SUPER-TILE is 128 consecutive documents
//     (= 1 slate of 128, 2 of 64, 4 of 32); wave w owns documents 16w..16w+15 as one 16-document tile.
//   * every FC layer is computed TRANSPOSED with v_mfma_f32_16x16x4_f32 (exact fp32):
//         z^T[n][doc] = sum_f W[n][f] x^T[f][doc]      A = W fragment, B = activations, lane&15 = document.
//     The accumulator layout (row = 4*(lane>>4)+reg = output feature, col = lane&15 = document) is exactly
//     the B-operand layout of the next layer when k-step s of input tile T covers features 16T+4q+s, so
//     activations NEVER leave the wave's registers between layers.  Biases ride along as a ones feature.
//   * weight fragments are pre-packed in lane order (one coalesced 1 KiB load per 16x16 tile of W, L2-hot).
//   * only the weight gradients dW = dz^T h contract over DOCUMENTS: dz and h tiles of a 64-document chunk
//     are staged in LDS ([doc][feature], stride 144 -> conflict-free fragment reads) and each wave owns a
//     quarter of the dW tiles, accumulated in registers across the whole persistent loop; X is staged in
//     LDS once per super-tile and serves both fc1 (B operand) and dW1 (B operand).
//   * per-workgroup dW partials go to a workspace and are summed in a FIXED order by a second kernel
//     (no float atomics: bit-reproducible gradients).
#include "../../include/ltr_mi355x.h"
#include "ltr_slate_losses.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>


// LTR_F16X2 = 1 builds the f16 x 2 split variant (libltr_mi355x_f16x2.so, same C ABI): fc1 / fc2 / dh1 -- the GEMMs whose
// B operand lives in the wave's registers -- run on v_mfma_f32_16x16x32_f16 with every fp32 operand split into two f16
// pieces (hi + lo, power-of-two pre-scaling per document / per weight matrix / running maximum) and the piece products
// (hi hi, hi lo, lo hi and -- LTR_LOLO -- lo lo) accumulated in fp32; the weight-gradient GEMMs run the same way on f16 image
// pairs staged in LDS.  4 B per element like fp32, 16/4 = 4 x the fp32 matrix rate.  See DESIGN.md section 4.3.
#ifndef LTR_F16X2
#define LTR_F16X2 0
#endif
#ifndef LTR_LOLO
#define LTR_LOLO 1               // 1: all FOUR piece products (the lo lo term too): the product of the split operands is then exact and
#endif                           //    the error is the operands' 2^-22 representation alone.  The matrix pipe has the slack (the FC GEMMs
                                 //    are bound by weight-fragment bandwidth, the dW GEMMs by LDS latency); with three products the
                                 //    cancelling bias gradients of a 48-document golden missed the 1e-5 bar by 6 %.

using namespace ltr;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTileDocs = 128;   // documents per super-tile
constexpr int kWaves = 8;        // 2 waves per SIMD: the co-resident wave hides LDS / L2 latency
constexpr int kThreads = kWaves * 64;
constexpr int kChunkDocs = 64;   // documents per dW staging chunk (the 16-doc tiles of 4 waves)
#ifndef LTR_RING
#define LTR_RING 6
#endif
constexpr int kRing = LTR_RING;  // weight fragments kept in flight per wave (1 KiB each, L2 -> registers)
// Next-X-tile prefetch policy, per kernel instantiation (A/B on MI355X, profiles/r01_variant_ab.json):
//   registers (36 VGPRs live across the loop) where they are free: forward-only kernels -8..-19 %, the 136-64-32
//   net's fused kernel -4 %;  L2-only prefetch (one dword per 128-B line) for the 136-136-136 backward/fused
//   kernels, where 36 more live registers spill and cost +11 %.
template <class N, int MODE>
__host__ __device__ constexpr bool x_reg_prefetch() { return MODE == 0 /*MODE_FWD*/ || N::H1 <= 64; }

enum { ACT_ID = 0, ACT_RELU_DROP = 1, ACT_SIGMOID = 2 };
enum { MODE_FWD = 0, MODE_BWD = 1, MODE_FUSED = 2, MODE_BWD_SAVED = 3 };   // BWD_SAVED: h1 / h2 read back, not recomputed

// TWO_: a two-Linear-layer scorer  w3 . act1(W1 x + b1) + b3  (no fc2: the commented-out 136-64-1 DoubleLayerNet variant of
// doubleLayer.py:38-51, BASELINE.json configs[0]).  Declared with H2 = H1 and A2 = A1: "h2" is then h1 itself, the
// backward through fc3 yields dz1 directly, and the fc2 / dh1 / dW2 GEMMs are compiled out.
// DS_ (two-layer nets, ltr_fcw.h): 2 = the 64 hidden units are TWO COPIES of 32 real ones, copy c working on documents
// 64 c .. 64 c + 63 of a tile only (waves 2 c, 2 c + 1): half the matrix work per tile; the host duplicates the weight rows and adds
// the two gradient halves (ltr_triple_fold / ltr_triple_unfold_grads).
template <int F_, int H1_, int H2_, int A1_, int A2_, int BH1_, int BH2_, bool TWO_ = false, int DS_ = 1>
struct NetT {
    static constexpr int F = F_, H1 = H1_, H2 = H2_, A1 = A1_, A2 = A2_;
    static constexpr bool TWO = TWO_;
    static constexpr int DS = DS_;
    static_assert(DS_ == 1 || (TWO_ && DS_ == 2), "document-split copies exist for the two-layer kernel only");
    static_assert(!TWO_ || (H1_ == H2_ && A1_ == A2_), "two-layer nets are declared with H2 = H1, A2 = A1");
    static constexpr int XT = (F + 1 + 15) / 16;    // x tiles incl. the ones feature at index F
    static constexpr int NT1 = (H1 + 15) / 16;      // fc1 output tiles
    static constexpr int H1T = (H1 + 1 + 15) / 16;  // h1 tiles incl. the ones feature at index H1
    static constexpr int NT2 = (H2 + 15) / 16;      // fc2 output tiles
    static constexpr int LD = (XT > H1T ? XT : H1T) * 16;   // LDS row stride in floats (144: LD % 32 == 16)
    static constexpr int NW1 = NT1 * XT;            // dW1 tiles  [H1 rows][F+1 cols]
    static constexpr int NW2 = NT2 * H1T;           // dW2 tiles  [H2 rows][H1+1 cols]
    static constexpr int TW1 = (NW1 + kWaves - 1) / kWaves;
    static constexpr int TW2 = (NW2 + kWaves - 1) / kWaves;
    static constexpr int BH1 = BH1_, BH2 = BH2_;    // row-band height of the per-wave dW tile sets
    // packed weights (floats)
    static constexpr int W1F_OFF = 0;                          // [NT1][XT][64][4]
    static constexpr int W2F_OFF = W1F_OFF + NT1 * XT * 256;   // [NT2][H1T][64][4]
    static constexpr int W2T_OFF = W2F_OFF + NT2 * H1T * 256;  // [NT1][NT2][64][4]
    static constexpr int W3_OFF = W2T_OFF + NT1 * NT2 * 256;   // [NT2*16] then b3
#if LTR_F16X2
    // f16 hi / lo A-fragments in CONSUMPTION order [k-step P][output tile To][hi, lo][64 lanes][8 halfs]: 2 KiB per (P, To)
    static constexpr int KP1 = (XT + 1) / 2, KP2 = (H1T + 1) / 2, KPT = (NT2 + 1) / 2;   // 32-wide k-steps = tile pairs
    static constexpr int W1H_OFF = W3_OFF + NT2 * 16 + 16;
    static constexpr int W2H_OFF = W1H_OFF + KP1 * NT1 * 512;
    static constexpr int W2TH_OFF = W2H_OFF + (TWO_ ? 0 : KP2 * NT2 * 512);
    // two-layer nets: the same W1 once more as B-fragments of the feature-partitioned kernel (ltr_fcw.h):
    // [hidden tile][k-step][hi, lo][64 lanes][8 halfs], lane (i, g) = W1aug[16 tile + i][32 P + 8 g + j]
    static constexpr int W1B_OFF = W2TH_OFF + (TWO_ ? 0 : KPT * NT1 * 512);
    static constexpr int PACKED = W1B_OFF + (TWO_ ? KP1 * NT1 * 512 : 0);
    // 1 / scale of W1aug, W2aug, W2 ride in the spare floats of the w3 section: packed[W3_OFF + NT2*16 + 4 .. + 6]
#else
    static constexpr int PACKED = W3_OFF + NT2 * 16 + 16;
#endif
    // per-workgroup gradient partial (floats)
    static constexpr int P_W1 = 0;                             // [NT1*16][XT*16]
    static constexpr int P_W2 = P_W1 + NT1 * 16 * XT * 16;     // [NT2*16][H1T*16]
    static constexpr int P_W3 = P_W2 + NT2 * 16 * H1T * 16;    // [NT2*16]
    static constexpr int P_B3 = P_W3 + NT2 * 16;
    static constexpr int PART = P_B3 + 16;
    static constexpr int NPARAM = TWO ? H1 * F + H1 + H2 + 1 : H1 * F + H1 + H2 * H1 + H2 + H2 + 1;
    static_assert(LD % 32 == 16, "LDS stride must be 16 mod 32 for conflict-free fragment reads");
    static_assert(F % 4 == 0 && H1 % 4 == 0 && H2 % 4 == 0, "feature counts must be multiples of 4");
    static_assert(NT1 % BH1 == 0 && NT2 % BH2 == 0, "band height must divide the dW row-tile count");
};
using DoubleNet = NetT<136, 136, 136, ACT_RELU_DROP, ACT_RELU_DROP, 3, 3>;   // doubleLayer.py:54-66
using TripleNet = NetT<136, 64, 32, ACT_ID, ACT_SIGMOID, 2, 2>;              // tripleLayer.py:5-17
// the same classes on the reference's 64-feature collection (TD2003, utils/dataset.py:23-30)
using DoubleNet64 = NetT<64, 64, 64, ACT_RELU_DROP, ACT_RELU_DROP, 2, 2>;
using TripleNet64 = NetT<64, 64, 32, ACT_ID, ACT_SIGMOID, 2, 2>;
using TwoLayerNet64h = NetT<136, 64, 64, ACT_RELU_DROP, ACT_RELU_DROP, 2, 2, true>;   // 136 -> 64 -> 1 (bench-only)
// TripleLayerNet FOLDED: tripleLayer.py:14-16 puts NO activation between l1 and l2, so l3(sigmoid(l2(l1 x))) is the two-layer net
// w3 . sigmoid(Weff x + beff) + b3 with Weff = W2 W1 [32 x 136], beff = W2 b1 + b2 -- 2.46 x fewer multiply-adds per document, and
// the gradients of W1 / b1 / W2 / b2 follow from d Weff / d beff by two tiny products per STEP (ltr_triple_unfold_grads).  Run as
// the two-layer kernel with its 64 hidden rows = two document-split copies of the 32 units (DS = 2).
using TripleFolded = NetT<136, 64, 64, ACT_SIGMOID, ACT_SIGMOID, 2, 2, true, 2>;
// the same folded network as a plain 136 -> 32 -> 1 two-layer net for the generic pipeline's forward / backward launches (module path
// `net(x)`, slates other than 32 / 64 / 128): MODE_FWD / MODE_BWD / MODE_BWD_SAVED only
using TripleFolded32 = NetT<136, 32, 32, ACT_SIGMOID, ACT_SIGMOID, 2, 2, true>;
// ... and on the 64-feature collection (TD2003): every path of that network, the one-launch step included (generic pipeline)
using TripleFolded32F64 = NetT<64, 32, 32, ACT_SIGMOID, ACT_SIGMOID, 2, 2, true>;
#define LTR_FOR_NET_EXTRA(...)                                                              \
    case LTR_NET_TWO_LAYER_64H: { using NET = TwoLayerNet64h; __VA_ARGS__; } break;         \
    case LTR_NET_TRIPLE_FOLDED: { using NET = TripleFolded; __VA_ARGS__; } break;           \
    case LTR_NET_TRIPLE_FOLDED_32: { using NET = TripleFolded32; __VA_ARGS__; } break;     \
    case LTR_NET_TRIPLE_FOLDED_32_64: { using NET = TripleFolded32F64; __VA_ARGS__; } break;

// run the statement(s) given after `net` with NET bound to the network type of id `net` (LTR_NET_*); unknown ids ->
// LTR_ERR_PARAM.  Variadic: the statement may contain top-level commas (kernel launches).
#define LTR_FOR_NET(net, ...)                                                     \
    switch (net) {                                                                \
        case LTR_NET_DOUBLE: { using NET = DoubleNet; __VA_ARGS__; } break;       \
        case LTR_NET_TRIPLE: { using NET = TripleNet; __VA_ARGS__; } break;       \
        case LTR_NET_DOUBLE_64: { using NET = DoubleNet64; __VA_ARGS__; } break;  \
        case LTR_NET_TRIPLE_64: { using NET = TripleNet64; __VA_ARGS__; } break;  \
        LTR_FOR_NET_EXTRA(__VA_ARGS__)                                            \
        default: return LTR_ERR_PARAM;                                            \
    }

struct PipeArgs {
    const float *X;          // [n_docs][F]
    long long n_docs;
    const float *labels;     // fused: [B*S]
    int B, S;
    const float *packed;     // NetT::PACKED floats
    float *scores_out;       // MODE_FWD : [n_docs]
    const float *dscores_in; // MODE_BWD : [n_docs]
    float *slate_loss;       // MODE_FUSED: [B]
    float *partials;         // [grid][PART]
    float *acts_out;         // MODE_FWD, optional: post-activation h1 / h2 of every 16-document wave tile, lane-ordered
    const float *acts_in;    // MODE_BWD, optional: the same buffer -> fc1 / fc2 are NOT recomputed (ltr_mlp_backward_saved)
    const uint8_t *keep1;    // optional explicit dropout keep masks [n_docs][H1] / [n_docs][H2]
    const uint8_t *keep2;
    unsigned long long seed;
    int dropout;             // 1: training-mode dropout on ACT_RELU_DROP layers
    unsigned drop_thr16;     // 0: p = 0.5, one hash BIT per unit (doubleLayer.py:60); else drop a unit when its 16 hash bits < thr
    float drop_scale;        // 1 / (1 - p): what a kept unit is multiplied by
    int loss_kind;           // 0 approxNDCG, 1 ListNet, 2 LambdaLoss
    LambdaParams lp;         // loss_kind 2
    float *slate_count;      // loss_kind 2: kept pairs per slate (may be NULL)
    float alpha, eps, pad, gscale;
    int apply_sigmoid;
    int n_super;
    unsigned long long *stamps;   // diagnostic build (-DLTR_STAMPS) only: [grid][8 waves][16] s_memtime values
    int stamp_tile;               // ... of this workgroup-local tile iteration
    int debug_skip;          // diagnostic builds (-DLTR_DIAG) only: 1 loss, 2 dW GEMMs, 4 dh1, 8 fc2, 16 fc1 skipped
};

// Phase-ablation switch for timing experiments (results are WRONG when a phase is skipped): compiled in only
// under -DLTR_DIAG, like the phase stamps under -DLTR_STAMPS.  The shipped library has no such switch.
#ifdef LTR_DIAG
#define LTR_SKIP(a, bit) ((a).debug_skip & (bit))
#else
#define LTR_SKIP(a, bit) 0
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// 32 keep bits for features [32*word, 32*word+32) of document `doc` in dropout layer `layer`.
__device__ __forceinline__ unsigned keep_word(unsigned long long seed, int layer, long long doc, int word) {
    unsigned h = (unsigned)seed ^ (0x9E3779B9u * (unsigned)(layer + 1));
    h = mix32(h ^ (unsigned)doc);
    h = mix32(h ^ (unsigned)((unsigned long long)doc >> 32) ^ (unsigned)(seed >> 32));
    return mix32(h + 0x27D4EB2Fu * (unsigned)(word + 1));
}

// Sum over the 16 lanes of a DPP row (= the 16 documents sharing one q); the total lands in lane 15 of the row.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
#define LTR_DPP_SHR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + (n), 0xF, 0xF, true))
    v += LTR_DPP_SHR(v, 1);
    v += LTR_DPP_SHR(v, 2);
    v += LTR_DPP_SHR(v, 4);
    v += LTR_DPP_SHR(v, 8);
#undef LTR_DPP_SHR
    return v;
}
// The same for FOUR values at once, every step as ONE v_add_f32_dpp: hipcc fuses only the row_shr:2 / :4 steps of the form above and
// expands the other two into copy + v_mov_b32_dpp + v_add_f32 (two extra instructions per step and value, in kernels whose non-matrix
// time is instruction count).  The four chains are interleaved, so a DPP read is always >= 3 instructions behind the write of its
// source (the hardware wants 2 wait states); the leading s_nop covers the caller's last write.
#ifndef LTR_ROWSUM_ASM
#define LTR_ROWSUM_ASM 1
#endif
__device__ __forceinline__ f32x4 row_sum4_to_lane15(f32x4 v) {
#if LTR_ROWSUM_ASM
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define LTR_ROWSUM_STEP(n)                                                                  \
    "v_add_f32_dpp %0, %0, %0 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
    "v_add_f32_dpp %1, %1, %1 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
    "v_add_f32_dpp %2, %2, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
    "v_add_f32_dpp %3, %3, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm("s_nop 1\n\t" LTR_ROWSUM_STEP(1) LTR_ROWSUM_STEP(2) LTR_ROWSUM_STEP(4) LTR_ROWSUM_STEP(8) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef LTR_ROWSUM_STEP
    return f32x4{a, b, c, d};
#else
    return f32x4{row_sum_to_lane15(v[0]), row_sum_to_lane15(v[1]), row_sum_to_lane15(v[2]), row_sum_to_lane15(v[3])};
#endif
}

// z^T tiles = W fragments x B fragments for the wave's 16-document tile.
// wf: [NT][KT][64] float4, lane-ordered; lane (q = lane>>4, i = lane&15) holds W[16*To + i][16*T + 4q + s], s = 0..3.
// Output tiles are processed in pairs (two independent accumulator chains); the fragments stream through a
// ring of kRing registers-quads loaded kRing steps ahead of their use, so L2 latency hides behind MFMAs.
// Fragments are fetched with buffer loads: ONE descriptor (SGPRs) + ONE lane-offset VGPR + a scalar byte offset
// per fragment.  (With plain pointers hipcc hoists all NT*KT 64-bit fragment addresses out of the persistent
// loop and spills them.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 load_frag(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

template <int NT, int KT, int KMAX, int NMAX>
__device__ __forceinline__ void gemm_wx(__amdgpu_buffer_rsrc_t rsrc, int base_bytes, int lane_off,
                                        const f32x4 (&bin)[KMAX], f32x4 (&out)[NMAX]) {
    constexpr int NSTEP = NT * KT;
    // step n -> (To, T): pairs of output tiles interleaved so consecutive steps alternate accumulators
    auto tile_of = [](int n, int &To, int &T) {
        constexpr int NPAIR = NT / 2;
        if (n < NPAIR * 2 * KT) {
            const int pr = n / (2 * KT), r = n - pr * 2 * KT;
            To = 2 * pr + (r & 1);
            T = r >> 1;
        } else {
            To = NT - 1;
            T = n - NPAIR * 2 * KT;
        }
    };
    f32x4 ring[kRing];
#pragma unroll
    for (int n = 0; n < kRing; ++n) {
        if (n < NSTEP) {
            int To, T;
            tile_of(n, To, T);
            ring[n] = load_frag(rsrc, lane_off, base_bytes + (To * KT + T) * 1024);
        }
    }
#pragma unroll
    for (int To = 0; To < NT; ++To) out[To] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NSTEP; ++n) {
        int To, T;
        tile_of(n, To, T);
        const f32x4 a = ring[n % kRing];
        if (n + kRing < NSTEP) {
            int To2, T2;
            tile_of(n + kRing, To2, T2);
            ring[n % kRing] = load_frag(rsrc, lane_off, base_bytes + (To2 * KT + T2) * 1024);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) out[To] = mfma4(a[s], bin[T][s], out[To]);
        // pin [prefetch n+kRing][4 MFMAs of n] as one scheduling region: left alone, hipcc hoists every
        // fragment load of the unrolled GEMM to its top (SSA renaming defeats the ring) and spills
        __builtin_amdgcn_sched_barrier(0);
    }
}

#if LTR_F16X2
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#ifndef LTR_RING_H
#define LTR_RING_H 8             // 16-byte fragment pieces in flight per wave (hi and lo of 4 (k-step, tile) pairs)
#endif

// The f16 x 2 form of gemm_wx: out^T tiles = W x B with B = the wave's activation tiles `bin` (accumulator layout).
// k-step P covers input tiles 2P and 2P+1: element j of lane (q, d) is feature 16 (2P + (j >> 2)) + 4q + (j & 3) -- the
// weight packing (pack_kernel) uses the same order.  Each document (lane & 15) is scaled by its own power of two so that its
// largest |activation| lands in [2^13, 2^14); the weights were scaled the same way per matrix (inv_w = 1 / that scale); the
// accumulators are un-scaled at the end.  Loop order: k-steps outermost (the hi / lo split of a k-step's 8 values per lane is
// made once and used by every output tile; successive MFMAs of one accumulator are NT steps apart).
template <int NT, int KT, int KMAX, int NMAX>
__device__ __forceinline__ void gemm_wx_h(__amdgpu_buffer_rsrc_t rsrc, int base_bytes, int lane_off, const f32x4 (&bin)[KMAX],
                                          f32x4 (&out)[NMAX], float inv_w) {
    constexpr int KP = (KT + 1) / 2;
    constexpr int NSTEP = KP * NT;
    constexpr int NLOAD = 2 * NSTEP;
    constexpr int R = LTR_RING_H;
    static_assert(R % 2 == 0, "hi and lo pieces travel together");
    f32x4 ring[R];
#pragma unroll
    for (int n = 0; n < R; ++n)
        if (n < NLOAD) ring[n] = load_frag(rsrc, lane_off, base_bytes + n * 1024);
    // per-document scale: max |b| over the document's K features (this lane's registers x the 4 q lanes of the document)
    float m = 0.f;
#pragma unroll
    for (int T = 0; T < KT; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(bin[T][r]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    int e = __builtin_amdgcn_frexp_expf(m);               // m = f 2^e, f in [0.5, 1); 0 for m == 0
    e = e < -100 ? -100 : e;
    const float sc = ldexpf(1.f, 14 - e);
    const float un = ldexpf(inv_w, e - 14);
#pragma unroll
    for (int To = 0; To < NT; ++To) out[To] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 bhi, blo;
#pragma unroll
    for (int n = 0; n < NSTEP; ++n) {
        const int P = n / NT, To = n % NT;
        if (To == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int T = 2 * P + (j >> 2);
                const float v = (T < KT ? bin[T < KT ? T : 0][j & 3] : 0.f) * sc;
                const _Float16 h = (_Float16)v;
                bhi[j] = h;
                blo[j] = (_Float16)(v - (float)h);
            }
        }
        const h16x8 ahi = __builtin_bit_cast(h16x8, ring[(2 * n) % R]);
        const h16x8 alo = __builtin_bit_cast(h16x8, ring[(2 * n + 1) % R]);
        if (2 * n + R < NLOAD) ring[(2 * n) % R] = load_frag(rsrc, lane_off, base_bytes + (2 * n + R) * 1024);
        if (2 * n + 1 + R < NLOAD) ring[(2 * n + 1) % R] = load_frag(rsrc, lane_off, base_bytes + (2 * n + 1 + R) * 1024);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, out[To], 0, 0, 0);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, out[To], 0, 0, 0);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, out[To], 0, 0, 0);
        if (LTR_LOLO) out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, blo, out[To], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int To = 0; To < NT; ++To) out[To] *= un;
}

// (Sharing these fragments through LDS instead -- LDS-DMA chunks of two k-steps ping-ponging in the dW staging region, three chunk
// barriers per GEMM -- was built and measured no faster than this L2 stream: profiles/r03_variant_ab.json, step 2.  Removed.)

#endif

// Activation (+ dropout) on accumulator tiles; features >= H forced to 0 (only the partial last tile needs it).
// ReLU + Dropout(0.5): 2*max(v,0) AND-ed with a 0 / ~0 mask made from the keep bit (4 VALU ops per value).
// GENP: the kernel also carries the 16-bit-per-unit stream of dropout probabilities other than 0.5 (forward-only kernels: they
// have the registers; the fused / recomputing kernels are compiled for p = 0.5 -- 2.5 % on the f16x2 headline otherwise -- and
// other probabilities run through ltr_mlp_forward_save / ltr_mlp_backward_saved, where the backward needs no stream at all).
template <int ACT, int H, int NT, bool GENP, int NMAX>
__device__ __forceinline__ void activate(f32x4 (&h)[NMAX], int q, const PipeArgs &a, int layer, const uint8_t *keep,
                                         long long doc) {
    const bool in_range = doc < a.n_docs;
    const bool drop = (ACT == ACT_RELU_DROP) && a.dropout;
#pragma unroll
    for (int To = 0; To < NT; ++To) {
        unsigned kb = 0xFu;
        if (drop) {
            if (keep) {
                const int n0 = 16 * To + 4 * q;
                unsigned bytes = 0;
                if (in_range && n0 < H) bytes = *reinterpret_cast<const unsigned *>(keep + doc * H + n0);
                kb = ((bytes & 0xFFu) ? 1u : 0u) | ((bytes & 0xFF00u) ? 2u : 0u) | ((bytes & 0xFF0000u) ? 4u : 0u) |
                     ((bytes & 0xFF000000u) ? 8u : 0u);
            } else if (GENP && a.drop_thr16) {      // any p: 16 hash bits per unit (stream layer + 2, word = unit pair)
                const unsigned w0 = keep_word(a.seed, layer + 2, doc, 8 * To + 2 * q), w1 = keep_word(a.seed, layer + 2, doc, 8 * To + 2 * q + 1);
                kb = ((w0 & 0xffffu) >= a.drop_thr16 ? 1u : 0u) | ((w0 >> 16) >= a.drop_thr16 ? 2u : 0u) |
                     ((w1 & 0xffffu) >= a.drop_thr16 ? 4u : 0u) | ((w1 >> 16) >= a.drop_thr16 ? 8u : 0u);
            } else {
                const unsigned wbits = keep_word(a.seed, layer, doc, To >> 1);
                kb = (wbits >> (16 * (To & 1) + 4 * q)) & 0xFu;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = h[To][r];
            if (ACT == ACT_RELU_DROP) {
                v = fmaxf(v, 0.f);
                if (drop) {
                    const int m = ((int)(kb << (31 - r))) >> 31;                  // 0 or ~0
                    v = __builtin_bit_cast(float, __builtin_bit_cast(int, v * a.drop_scale) & m);   // p = 0.5: v * 2 == v + v
                }
            } else if (ACT == ACT_SIGMOID) {
                v = __frcp_rn(1.f + __expf(-v))      /* correctly rounded: h (1 - h) of the backward cancels when h is near 1 */;
            }
            if (16 * To + 16 > H) v = (16 * To + 4 * q + r < H) ? v : 0.f;      // compile-time: last tile only
            h[To][r] = v;
        }
    }
}

// g * act'(h) with the ReLU/dropout slope (1 or 2, wave-uniform) pre-multiplied into g by the caller.
template <int ACT>
__device__ __forceinline__ float apply_act_grad(float g, float h) {
    if (ACT == ACT_RELU_DROP) return h > 0.f ? g : 0.f;
    if (ACT == ACT_SIGMOID) return g * (h * (1.f - h));
    return g;
}

// Per-wave dW tile sets, fixed at compile time.  The NR x NC tile grid is walked in row bands of height BH,
// column-major inside a band; wave W owns the contiguous run [W*TW, W*TW+TW) of that walk: a compact ~BH x TW/BH
// block whose A (row) and B (column) fragments are shared between its tiles.
template <int NR, int NC, int BH>
__host__ __device__ constexpr int dw_row(int g) { return (g / (NC * BH)) * BH + (g % (NC * BH)) % BH; }
template <int NR, int NC, int BH>
__host__ __device__ constexpr int dw_col(int g) { return (g % (NC * BH)) / BH; }

// dW tiles of wave W += A^T B over one 64-document chunk.
//   a_base / b_base: this lane's LDS addresses of A[row q][16*0 + i] / B[row q][16*0 + i] for k-step 0;
//   k-step s adds 4*LD floats; tile (To, Ti) adds 16*To / 16*Ti floats: all immediates.
template <int W, int TW, int NR, int NC, int BH>
struct DwSet {   // which A (row) / B (column) fragments wave W's tile set touches
    static constexpr bool uses_row(int To) {
        for (int j = 0; j < TW; ++j)
            if (W * TW + j < NR * NC && dw_row<NR, NC, BH>(W * TW + j) == To) return true;
        return false;
    }
    static constexpr bool uses_col(int Ti) {
        for (int j = 0; j < TW; ++j)
            if (W * TW + j < NR * NC && dw_col<NR, NC, BH>(W * TW + j) == Ti) return true;
        return false;
    }
    static constexpr int n_frags() {
        int n = 0;
        for (int t = 0; t < NR; ++t) n += uses_row(t) ? 1 : 0;
        for (int t = 0; t < NC; ++t) n += uses_col(t) ? 1 : 0;
        return n;
    }
    static constexpr int n_tiles() {
        int n = 0;
        for (int j = 0; j < TW; ++j) n += (W * TW + j < NR * NC) ? 1 : 0;
        return n;
    }
};

template <int W, int TW, int NR, int NC, int BH, int LD>
__device__ __forceinline__ void dw_load(float (&af)[NR], float (&bf)[NC], const float *ar, const float *br) {
#pragma unroll
    for (int To = 0; To < NR; ++To)
        if (DwSet<W, TW, NR, NC, BH>::uses_row(To)) af[To] = ar[16 * To];
#pragma unroll
    for (int Ti = 0; Ti < NC; ++Ti)
        if (DwSet<W, TW, NR, NC, BH>::uses_col(Ti)) bf[Ti] = br[16 * Ti];
}

template <int W, int TW, int NR, int NC, int BH>
__device__ __forceinline__ void dw_mfma(f32x4 (&acc)[TW], const float (&af)[NR], const float (&bf)[NC]) {
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int g = W * TW + j;
        if (g < NR * NC) acc[j] = mfma4(af[dw_row<NR, NC, BH>(g)], bf[dw_col<NR, NC, BH>(g)], acc[j]);
    }
}

#ifndef LTR_DW_PIPE
#define LTR_DW_PIPE 0
#endif
#ifndef LTR_PRIO
#define LTR_PRIO 0               // 1: s_setprio 1 for waves 4-7 (A/B experiment)
#endif
#ifndef LTR_KROT
#define LTR_KROT 0               // 1: waves 4-7 walk the dW k steps half a chunk out of phase (A/B experiment)
#endif
#ifndef LTR_H1_BITS
#define LTR_H1_BITS 0
#endif
#ifndef LTR_STG_SWZ
#define LTR_STG_SWZ 0            // 1: XOR-swizzled dW staging images (A/B experiment): row r keeps its 4-float pieces at
#endif                           //    position (4 q) ^ 4 ((r >> 1) & 3) inside each 16-feature tile -> 2-way instead of 8-way
                                 //    conflicted ds_write_b128; fragment reads use two lane bases (even / odd k steps)

template <int W, int TW, int NR, int NC, int BH, int LD>
__device__ __forceinline__ void dw_chunk_w(f32x4 (&acc)[TW], const float *a_base, const float *b_base, const float *a_odd,
                                           const float *b_odd) {
    constexpr int KS = kChunkDocs / 4;
#if LTR_DW_PIPE
    static_assert(!LTR_STG_SWZ, "the pipelined variant reads unswizzled images");
    // Fragments double-buffered and the interleave PINNED (one LDS read per MFMA): the reads of k-step s+1 are in
    // flight under the MFMAs of k-step s.  (Left to itself hipcc sinks the reads next to their use and waits
    // lgkmcnt(0) every 5-6 MFMAs.)
    constexpr int NF = DwSet<W, TW, NR, NC, BH>::n_frags();
    constexpr int NM = DwSet<W, TW, NR, NC, BH>::n_tiles();
    float af0[NR], bf0[NC], af1[NR], bf1[NC];
    dw_load<W, TW, NR, NC, BH, LD>(af0, bf0, a_base, b_base);
#pragma unroll
    for (int s = 0; s < KS; s += 2) {
        dw_load<W, TW, NR, NC, BH, LD>(af1, bf1, a_base + (s + 1) * 4 * LD, b_base + (s + 1) * 4 * LD);
        dw_mfma<W, TW, NR, NC, BH>(acc, af0, bf0);
#pragma unroll
        for (int i = 0; i < (NM > NF ? NM : NF); ++i) {
            if (i < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            if (i < NM) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        }
        if (s + 2 < KS)
            dw_load<W, TW, NR, NC, BH, LD>(af0, bf0, a_base + (s + 2) * 4 * LD, b_base + (s + 2) * 4 * LD);
        dw_mfma<W, TW, NR, NC, BH>(acc, af1, bf1);
#pragma unroll
        for (int i = 0; i < (NM > NF ? NM : NF); ++i) {
            if (i < NF && s + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i < NM) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
#else
    for (int s0 = 0; s0 < KS; ++s0) {
        // LTR_KROT: the second half of the workgroup walks the k steps half a chunk out of phase with its SIMD partner
        const int s = LTR_KROT ? ((s0 + (W >= kWaves / 2 ? KS / 2 : 0)) % KS) : s0;
        float af[NR], bf[NC];
        dw_load<W, TW, NR, NC, BH, LD>(af, bf, ((s & 1) ? a_odd : a_base) + s * 4 * LD, ((s & 1) ? b_odd : b_base) + s * 4 * LD);
        dw_mfma<W, TW, NR, NC, BH>(acc, af, bf);
    }
#endif
}

template <int TW, int NR, int NC, int BH, int LD>
__device__ __forceinline__ void dw_chunk(int w, f32x4 (&acc)[TW], const float *a_base, const float *b_base, const float *a_odd = nullptr,
                                         const float *b_odd = nullptr) {
    if (!a_odd) a_odd = a_base;
    if (!b_odd) b_odd = b_base;
    switch (w) {   // wave-uniform: one specialised, branch-free body per wave
        case 0: dw_chunk_w<0, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 1: dw_chunk_w<1, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 2: dw_chunk_w<2, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 3: dw_chunk_w<3, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 4: dw_chunk_w<4, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 5: dw_chunk_w<5, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        case 6: dw_chunk_w<6, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
        default: dw_chunk_w<7, TW, NR, NC, BH, LD>(acc, a_base, b_base, a_odd, b_odd); break;
    }
}

#if LTR_F16X2
// ---- step 3: the weight-gradient GEMMs dW = dz^T [h | 1] on the f16 matrix cores as well.
// Both operands are staged in LDS as f16 hi / lo IMAGES [document][feature] (2 x 2 B per element = the bytes of the fp32
// staging they replace) and read k-major (k = document) with the transposing ds_read_b64_tr_b16.  The contraction runs
// over documents, so an operand's power-of-two scale must be the same for every document of a launch-long accumulation:
// each operand carries a RUNNING-MAX exponent (it only grows; when it grows, the wave multiplies its dW accumulators by
// the matching power of two -- exact -- before adding the tile).  X itself lives in LDS as such an image pair from the
// moment it lands, and serves fc1 (B operand, read row-wise) and dW1 (read k-major) alike.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// fragment of feature tile t (16 features: lane & 15) over the 32 documents from r0 of a [document][LDH] image; lane
// (i, g) gets documents r0 + 4g + {0..3} and r0 + 16 + 4g + {0..3} -- the same k-slot order for both operands.
// lane_base = img + (4 g + (i >> 2)) * LDH + 4 (i & 3).  EXEC must be full.
template <int LDH>
__device__ __forceinline__ h16x8 tr_frag_h(const uint16_t *lane_base, int r0, int t) {
    typedef __attribute__((address_space(3))) s16x4 *lp4;
    const uint16_t *p = lane_base + r0 * LDH + 16 * t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p + 16 * LDH));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return __builtin_bit_cast(h16x8, u32x4{l2[0], l2[1], h2[0], h2[1]});
}

// 4 fp32 values -> scaled hi / lo f16 quads (8 bytes each)
__device__ __forceinline__ void split4(const f32x4 &v, float sc, u32x2 &hi, u32x2 &lo) {
    h16x4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = v[r] * sc;
        h[r] = (_Float16)x;
        l[r] = (_Float16)(x - (float)h[r]);
    }
    hi = __builtin_bit_cast(u32x2, h);
    lo = __builtin_bit_cast(u32x2, l);
}

// dW tiles of wave W += A^T B over KS k-steps of 32 documents; A / B: lane bases of the hi and lo images.  The wave's row
// fragments stay resident for a k-step, its columns are walked one at a time.  Scheduling is left to hipcc: pinning every
// column's [reads][MFMAs] with scheduling barriers and software-pipelining the column reads by hand were both measured
// slower (5.25 / 5.48 vs 4.77 ms per step, profiles/r03_variant_ab.json).
template <int W, int TW, int NR, int NC, int BH, int LDH, int KS>
__device__ __forceinline__ void dw_chunk_h_w(f32x4 (&acc)[TW], const uint16_t *ahi, const uint16_t *alo, const uint16_t *bhi,
                                             const uint16_t *blo) {
    using S = DwSet<W, TW, NR, NC, BH>;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        h16x8 ah[NR], al[NR];
#pragma unroll
        for (int To = 0; To < NR; ++To)
            if (S::uses_row(To)) {
                ah[To] = tr_frag_h<LDH>(ahi, 32 * s, To);
                al[To] = tr_frag_h<LDH>(alo, 32 * s, To);
            }
#pragma unroll
        for (int Ti = 0; Ti < NC; ++Ti)
            if (S::uses_col(Ti)) {
                const h16x8 bh = tr_frag_h<LDH>(bhi, 32 * s, Ti), bl = tr_frag_h<LDH>(blo, 32 * s, Ti);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int g = W * TW + j;
                    if (g < NR * NC && dw_col<NR, NC, BH>(g) == Ti) {
                        const int r = dw_row<NR, NC, BH>(g);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh, acc[j], 0, 0, 0);
                        if (LTR_LOLO) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bl, acc[j], 0, 0, 0);
                    }
                }
            }
    }
}

template <int TW, int NR, int NC, int BH, int LDH, int KS>
__device__ __forceinline__ void dw_chunk_h(int w, f32x4 (&acc)[TW], const uint16_t *ahi, const uint16_t *alo, const uint16_t *bhi,
                                           const uint16_t *blo) {
    switch (w) {   // wave-uniform: one specialised, branch-free body per wave (full EXEC for the transposing reads)
        case 0: dw_chunk_h_w<0, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 1: dw_chunk_h_w<1, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 2: dw_chunk_h_w<2, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 3: dw_chunk_h_w<3, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 4: dw_chunk_h_w<4, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 5: dw_chunk_h_w<5, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        case 6: dw_chunk_h_w<6, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
        default: dw_chunk_h_w<7, TW, NR, NC, BH, LDH, KS>(acc, ahi, alo, bhi, blo); break;
    }
}

// fc1 with B read from the X image pair (uniform scale): out^T tiles = W x X^T for this lane's document row.
template <int NT, int KT, int LDH, int NMAX>
__device__ __forceinline__ void gemm_wx_hx(__amdgpu_buffer_rsrc_t rsrc, int base_bytes, int lane_off, const uint16_t *xhi_row,
                                           const uint16_t *xlo_row, f32x4 (&out)[NMAX], float un) {
    constexpr int KP = (KT + 1) / 2;
    constexpr int NSTEP = KP * NT, NLOAD = 2 * NSTEP, R = LTR_RING_H;
    f32x4 ring[R];
#pragma unroll
    for (int n = 0; n < R; ++n)
        if (n < NLOAD) ring[n] = load_frag(rsrc, lane_off, base_bytes + n * 1024);
#pragma unroll
    for (int To = 0; To < NT; ++To) out[To] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 bhi, blo;
#pragma unroll
    for (int n = 0; n < NSTEP; ++n) {
        const int P = n / NT, To = n % NT;
        if (To == 0) {
            const u32x2 z = {0u, 0u};
            const u32x2 h0 = *reinterpret_cast<const u32x2 *>(xhi_row + 32 * P), l0 = *reinterpret_cast<const u32x2 *>(xlo_row + 32 * P);
            const u32x2 h1 = 2 * P + 1 < KT ? *reinterpret_cast<const u32x2 *>(xhi_row + 32 * P + 16) : z;
            const u32x2 l1 = 2 * P + 1 < KT ? *reinterpret_cast<const u32x2 *>(xlo_row + 32 * P + 16) : z;
            bhi = __builtin_bit_cast(h16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
            blo = __builtin_bit_cast(h16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
        }
        const h16x8 ahi = __builtin_bit_cast(h16x8, ring[(2 * n) % R]);
        const h16x8 alo = __builtin_bit_cast(h16x8, ring[(2 * n + 1) % R]);
        if (2 * n + R < NLOAD) ring[(2 * n) % R] = load_frag(rsrc, lane_off, base_bytes + (2 * n + R) * 1024);
        if (2 * n + 1 + R < NLOAD) ring[(2 * n + 1) % R] = load_frag(rsrc, lane_off, base_bytes + (2 * n + 1 + R) * 1024);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, out[To], 0, 0, 0);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, out[To], 0, 0, 0);
        out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, out[To], 0, 0, 0);
        if (LTR_LOLO) out[To] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, blo, out[To], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int To = 0; To < NT; ++To) out[To] *= un;
}

// running-max exponent of a staged operand: ex such that max|v| * 2^(14 - ex) stays below 2^14
__device__ __forceinline__ int grow_exp(int ex, float m) {
    int e = __builtin_amdgcn_frexp_expf(m);
    e = m > 0.f ? e : -100;
    e = e < -100 ? -100 : e;
    return e > ex ? e : ex;
}
#endif

// accumulator tiles of wave W -> workgroup partial (row = 4q + r, col = lane & 15)
template <int W, int TW, int NR, int NC, int BH>
__device__ __forceinline__ void dw_store_w(const f32x4 (&acc)[TW], float *dst, int q, int d, float scale) {
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int g = W * TW + j;
        if (g < NR * NC) {
            const int To = dw_row<NR, NC, BH>(g), Ti = dw_col<NR, NC, BH>(g);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * To + 4 * q + r) * (NC * 16) + 16 * Ti + d] = acc[j][r] * scale;
        }
    }
}

template <int TW, int NR, int NC, int BH>
__device__ __forceinline__ void dw_store(int w, const f32x4 (&acc)[TW], float *dst, int q, int d, float scale = 1.f) {
    switch (w) {
        case 0: dw_store_w<0, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 1: dw_store_w<1, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 2: dw_store_w<2, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 3: dw_store_w<3, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 4: dw_store_w<4, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 5: dw_store_w<5, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        case 6: dw_store_w<6, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
        default: dw_store_w<7, TW, NR, NC, BH>(acc, dst, q, d, scale); break;
    }
}

#ifdef LTR_STAMPS
#define LTR_STAMP(k)                                                                                          \
    if (a.stamps && lane == 0 && (st - (int)blockIdx.x) / (int)gridDim.x == a.stamp_tile)                      \
        a.stamps[((size_t)blockIdx.x * kWaves + w) * 16 + (k)] = __builtin_readcyclecounter();
#else
#define LTR_STAMP(k)
#endif

// LOSS (MODE_FUSED only): 0 approxNDCG, 1 ListNet, 2 LambdaLoss -- a template parameter so that each fused
// kernel carries only its own loss code (sharing one kernel cost the approxNDCG path 2 % in registers/code).
// This wave's 16 documents of a super-tile: kXV4 float4 per lane, zeros past the end of the batch.
template <class N>
__host__ __device__ constexpr int kXV4() { return (16 * (N::F / 4) + 63) / 64; }

template <class N>
__device__ __forceinline__ void load_x_tile(f32x4 (&xr)[kXV4<N>()], const PipeArgs &a, long long row0, int lane) {
    constexpr int V4_PER_ROW = N::F / 4;
    constexpr int V4 = 16 * V4_PER_ROW;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.X + row0 * N::F);
#pragma unroll
    for (int m = 0; m < kXV4<N>(); ++m) {
        const int e = lane + 64 * m;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e < V4 && row0 + e / V4_PER_ROW < a.n_docs) v = src[e];
        xr[m] = v;
    }
}

#ifndef LTR_LOSS_UNR
#define LTR_LOSS_UNR 1           // trips of 8 columns unrolled in the approxNDCG sweeps of the 136-wide nets: 2 costs 72 B/lane more scratch and 2 % (profiles/r03_variant_ab.json)
#endif
#ifndef LTR_XDMA
#define LTR_XDMA 1
#endif
#ifndef LTR_PIPE_ST128_MAXH
#define LTR_PIPE_ST128_MAXH 64      // widest first hidden layer that gets the slate-128 instantiation (A/B: profiles/r04_variant_ab.json)
#endif
#ifndef LTR_PIPE_ST128
#define LTR_PIPE_ST128 1          // a slate-128 instantiation of the fused approxNDCG pipeline kernels (one loss copy with fixed geometry)
#endif
#ifndef LTR_X_AUX
#define LTR_X_AUX 2              // cache policy of the X stream: 2 = nt (read once by one CU: keeps the L2-resident weight
                                 // fragments from being evicted; FETCH_SIZE -16 % at equal time), 0 = default
#endif
// X: HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs, no address math): one wave-instruction
// per document row, lanes 0..F/4-1 active, destination = row base + lane*16 (rows keep their padded LD stride;
// the pad columns -- ones feature + zeros -- are constant and written once per kernel).  Rows past the end of
// the batch are zero-filled.  Issued and waited for at the top of a tile, with no other vector-memory load in
// flight: while an LDS-DMA is pending hipcc turns every vmcnt wait into vmcnt(0), so overlapping it with the
// weight-fragment ring or the dW GEMMs costs more than it hides (measured, profiles/r01_variant_ab.json).
template <class N>
__device__ __forceinline__ void dma_x_rows(const PipeArgs &a, float *Xs, long long row0, int w, int lane) {
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float *dst = Xs + (16 * w + r) * N::LD;                 // wave-uniform
        if (row0 + r < a.n_docs) {                               // wave-uniform
            if (lane < N::F / 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(a.X + (row0 + r) * N::F + 4 * lane), (lptr_t)dst, 16, 0, LTR_X_AUX);
        } else if (lane < N::F / 4) {
            *reinterpret_cast<f32x4 *>(dst + 4 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

#include "ltr_fcw.h"
// A weights-stationary kernel for the live DoubleLayerNet was built and measured in round 4: parity-green but
// slower than this pipeline (0.567 of the fp32 MFMA peak against 0.612), so it is not in the tree; its source is csrc/ltr_wst.h at
// commit 139af5e, its measurements profiles/r04_wst_weights_stationary_experiment.json.

// ST (MODE_FUSED with approxNDCG only): 128 = the slate length is the compile-time constant 128 (the kernel then carries ONE copy
// of the loss: three copies cost the 136-wide kernel 180 B/lane of scratch); 0 = a.S at run time.
template <class N, int MODE, int LOSS, int ST = 0>
__global__ void __launch_bounds__(kThreads, 2) slate_pipeline_kernel(const PipeArgs a) {
    constexpr int LD = N::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                         // [128][LD]
    float *Ds = Xs + kTileDocs * LD;          // [64][LD]   dz chunk (A operand of dW)
    float *Hs = Ds + kChunkDocs * LD;         // [64][LD]   h1 chunk (B operand of dW2)
    float *sc = Hs + kChunkDocs * LD;         // [128] scores
    float *yl = sc + kTileDocs;               // [128] labels (-inf padded)
    float *gn = yl + kTileDocs;               // [128] gains
    float *gg = gn + kTileDocs;               // [128] loss scratch
    float *dsc = gg + kTileDocs;              // [128] d loss / d score
    float *uu = dsc + kTileDocs;              // [128] loss scratch (per-document exponentials)
    float *mk = uu + kTileDocs;               // [128] loss scratch (valid-document mask)
    float *xt = mk + kTileDocs;               // [128] loss scratch (LambdaLoss ranks; approxNDCG: masked exponentials)
    float *w3s = xt + kTileDocs;              // [NT2*16 + 16] w3 (zero padded), b3
    float *dw3 = w3s + N::NT2 * 16 + 16;      // [kWaves][NT2*16] per-wave dw3 accumulators
    float *scratch = dw3 + kWaves * N::NT2 * 16;   // [512 + 4*32] slate-group scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, d = lane & 15;

    // packed weights behind one buffer descriptor (wave-uniform: kernel argument)
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, N::PACKED * 4, 0x00020000);
    const int lane_off = lane * 16;

#if LTR_PRIO
    // static priority for the second-dispatched half of the workgroup (the arbitration loser on every segment when the
    // two waves of a SIMD run the same program): MI355X_MICROARCH.md, "Two waves per SIMD", item 4
    if (w >= kWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (int j = tid; j < N::NT2 * 16 + 16; j += kThreads) w3s[j] = a.packed[N::W3_OFF + j];
    for (int j = tid; j < kWaves * N::NT2 * 16; j += kThreads) dw3[j] = 0.f;
    for (int j = tid; j < kThreads + 4 * 32; j += kThreads) scratch[j] = 0.f;   // slate-group scratch (ListNet / LambdaLoss groups; approxNDCG: [0, 128) wave partials)
    if (MODE == MODE_FUSED && LOSS == 0) {
        __syncthreads();
        ltr_fill_inv_discount(scratch + 256, kTileDocs, tid, kThreads);          // [256, 384): 1 / log2(2 + rank)
    }
    float db3 = 0.f;
    f32x4 accW1[N::TW1], accW2[N::TW2];
    if (MODE != MODE_FWD) {
#pragma unroll
        for (int n = 0; n < N::TW1; ++n) accW1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < N::TW2; ++n) accW2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int my_row = 16 * w + d;                         // this lane's document inside the super-tile
    const int chunk = w >> 2;                              // dW chunk this wave's tile belongs to
    const int crow = 16 * (w & 3) + d;                     // ... and its row inside the chunk

    constexpr bool XPREF = x_reg_prefetch<N, MODE>();
    constexpr bool XDMA = LTR_XDMA && !XPREF;
#if LTR_F16X2
#ifndef LTR_DWH
#define LTR_DWH 1                // 1: the weight-gradient GEMMs on the f16 matrix cores too (step 3)
#endif
    constexpr bool DWH = LTR_DWH && MODE != MODE_FWD && (XDMA || XPREF);
    constexpr int LDH = LD;                                            // halfs per row of an f16 image (LD / 2 dwords: 8 mod 64 at LD = 144)
    uint16_t *Xhi = reinterpret_cast<uint16_t *>(Xs), *Xlo = Xhi + kTileDocs * LDH;   // X as hi / lo images: the bytes of the fp32 tile
    uint16_t *Sg = reinterpret_cast<uint16_t *>(Ds);                   // staging images (and the landing zone of the X DMA)
    float *exch = scratch + kThreads + 4 * 32;                         // [64] per-wave maxima, exchanged at barriers that exist anyway
    int exx = 1, exh = 1, exd2 = -100, exd1 = -100, E1 = -400, E2 = -400;   // running-max exponents (>= 1 where a ones feature rides along)
    float w3max = 0.f;
#else
    constexpr bool DWH = false;
#endif
    f32x4 xn[kXV4<N>()];
#if LTR_F16X2
    if (DWH) {    // zero pad columns of both X images (the ones feature at column F is rewritten per tile: it carries the scale)
        for (int e = tid; e < kTileDocs * (LD - N::F); e += kThreads) {
            const int r = e / (LD - N::F), c = N::F + e % (LD - N::F);
            Xhi[r * LDH + c] = 0;
            Xlo[r * LDH + c] = 0;
        }
        __syncthreads();                                   // w3s is complete
        float m = 0.f;
        for (int j = lane; j < N::NT2 * 16; j += 64) m = fmaxf(m, fabsf(w3s[j]));
        w3max = wave_allmax(m);
    } else
#endif
    if (XDMA) {   // pad columns of the X tile (ones feature at column F, zeros up to LD): constant, written once
        constexpr int PADV4 = (LD - N::F) / 4;
        for (int e = tid; e < kTileDocs * PADV4; e += kThreads) {
            const int r = e / PADV4, c4 = e - r * PADV4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c4 == 0) v[0] = 1.f;
            *reinterpret_cast<f32x4 *>(Xs + r * LD + N::F + 4 * c4) = v;
        }
    }
    const int lane_outer = lane;
    for (int st = blockIdx.x; st < a.n_super; st += gridDim.x) {
        const long long doc_base = (long long)st * kTileDocs;
        // Per-lane geometry is re-derived per tile from a laundered lane id: left loop-invariant, hipcc hoists the dozens
        // of LDS addresses built from it out of the persistent loop and, at the register limit, parks them in scratch.
        // (Only where registers are the limit -- the 136-wide backward / fused kernels, 6.78 -> 6.48 ms; the 136-64-32
        // net and the forward-only kernels have registers to spare and lose 3-4 % to the recomputation.)
        constexpr bool DEHOIST = N::H1 > 64 && MODE != MODE_FWD;
        int lane = lane_outer;
        if (DEHOIST) asm volatile("" : "+v"(lane));
        const int q = lane >> 4, d = lane & 15;
        const int my_row = 16 * w + d;
        const int crow = 16 * (w & 3) + d;
        const int lane_off = lane * 16;
        const int tid = 64 * w + lane;
#if LTR_F16X2
        // this wave's 16 rows (held in xn) -> hi / lo images, once the tile's scale is known (all 8 wave maxima published)
        auto convert_x = [&]() {
            float xm = 1.f;                                // the ones feature
#pragma unroll
            for (int i = 0; i < kWaves; ++i) xm = fmaxf(xm, exch[i]);
            exx = grow_exp(exx, xm);
            const float sx = ldexpf(1.f, 14 - exx);
            constexpr int V4_PER_ROW = N::F / 4, V4 = 16 * V4_PER_ROW;
#pragma unroll
            for (int mm = 0; mm < kXV4<N>(); ++mm) {
                const int e = lane + 64 * mm;
                if (e < V4) {
                    u32x2 hi, lo;
                    split4(xn[mm], sx, hi, lo);
                    const int off = (16 * w + e / V4_PER_ROW) * LDH + 4 * (e % V4_PER_ROW);
                    *reinterpret_cast<u32x2 *>(Xhi + off) = hi;
                    *reinterpret_cast<u32x2 *>(Xlo + off) = lo;
                }
            }
            if (lane < 16) {
                Xhi[(16 * w + lane) * LDH + N::F] = __builtin_bit_cast(uint16_t, (_Float16)sx);
                Xlo[(16 * w + lane) * LDH + N::F] = 0;
            }
        };
        auto publish_xmax = [&]() {
            float m = 0.f;
#pragma unroll
            for (int mm = 0; mm < kXV4<N>(); ++mm)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(xn[mm][r]));
            m = wave_allmax(m);
            if (lane == 0) exch[w] = m;
        };
        if (DWH && XPREF) {   // register-prefetch kernels: the slice is in xn already (first tile: fetched here), its max goes out
            if (st == (int)blockIdx.x) load_x_tile<N>(xn, a, doc_base + 16 * w, lane);      // before the barrier below
            publish_xmax();
        }
#endif
        LTR_STAMP(0)
        __syncthreads();   // previous super-tile done with Xs / sc / dsc
        // ---- X: HBM -> LDS, coalesced 16 B per lane; the wave's 16 documents are contiguous in memory.
#if LTR_F16X2
        if (DWH && XPREF) {
            convert_x();
        } else if (DWH) {
            // X lands as fp32 in the (idle) staging region; every wave takes ITS 16 rows into registers and publishes their max
            dma_x_rows<N>(a, Ds, doc_base + 16 * w, w, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            constexpr int V4_PER_ROW = N::F / 4, V4 = 16 * V4_PER_ROW;
#pragma unroll
            for (int mm = 0; mm < kXV4<N>(); ++mm) {
                const int e = lane + 64 * mm;
                xn[mm] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < V4) xn[mm] = *reinterpret_cast<const f32x4 *>(Ds + (16 * w + e / V4_PER_ROW) * LD + 4 * (e % V4_PER_ROW));
            }
            publish_xmax();
        } else
#endif
        if (XDMA) {
            dma_x_rows<N>(a, Xs, doc_base + 16 * w, w, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // register-prefetch kernels: only the first tile is loaded here, later ones arrive ahead of time
            if (!XPREF || st == (int)blockIdx.x) load_x_tile<N>(xn, a, doc_base + 16 * w, lane);
            constexpr int V4_PER_ROW = N::F / 4;
            constexpr int V4 = 16 * V4_PER_ROW;             // float4s per wave
#pragma unroll
            for (int m = 0; m < kXV4<N>(); ++m) {
                const int e = lane + 64 * m;
                if (e < V4) {
                    const int r = e / V4_PER_ROW, c4 = e - r * V4_PER_ROW;
                    *reinterpret_cast<f32x4 *>(Xs + (16 * w + r) * LD + 4 * c4) = xn[m];
                }
            }
            // ones feature at column F, zeros up to LD
            constexpr int PADV4 = (LD - N::F) / 4;
            for (int e = lane; e < 16 * PADV4; e += 64) {
                const int r = e / PADV4, c4 = e - r * PADV4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c4 == 0) v[0] = 1.f;
                *reinterpret_cast<f32x4 *>(Xs + (16 * w + r) * LD + N::F + 4 * c4) = v;
            }
        }
        if (MODE == MODE_FUSED && tid < kTileDocs) {
            const long long doc = doc_base + tid;
            const float y = doc < (long long)a.B * a.S ? a.labels[doc] : a.pad;
            if (LOSS != 1) stage_label(y, a.pad, yl[tid], gn[tid]);
            else yl[tid] = doc < (long long)a.B * a.S ? y : 0.f;
        }
        if ((MODE == MODE_BWD || MODE == MODE_BWD_SAVED) && tid < kTileDocs) {
            const long long doc = doc_base + tid;
            dsc[tid] = doc < a.n_docs ? a.dscores_in[doc] : 0.f;
        }
        __syncthreads();
#if LTR_F16X2
        if (DWH && !XPREF) convert_x();   // the tile's scale is known now (read back by this wave in fc1, by all waves in dW1)
#endif

        LTR_STAMP(1)
        const long long gdoc = doc_base + my_row;
        // forward-only kernels never reach the backward's prefetch point: fetch the next X slice here, it lands
        // during fc1/fc2 (xn was copied to LDS above)
        if (MODE == MODE_FWD && XPREF && st + (int)gridDim.x < a.n_super)
            load_x_tile<N>(xn, a, (long long)(st + gridDim.x) * kTileDocs + 16 * w, lane);
        // ---- saved activations (ltr_mlp_forward_save -> ltr_mlp_backward_saved): one 1 KiB lane-ordered fragment per
        //      (16-document wave tile, feature tile), h1 tiles then h2 tiles: coalesced 16 B per lane both ways
        constexpr int ACT_FRAGS = N::NT1 + (N::TWO ? 0 : N::NT2);
        constexpr bool saved = MODE == MODE_BWD_SAVED;
        const size_t act_base = (((size_t)st * kWaves + w) * ACT_FRAGS) * 256 + (size_t)lane * 4;
        // ---- fc1
        f32x4 h1[N::H1T];
        f32x4 h2[N::NT2];
        if (saved) {
            if (N::H1T > N::NT1) h1[N::H1T - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < N::NT1; ++T) h1[T] = *reinterpret_cast<const f32x4 *>(a.acts_in + act_base + (size_t)T * 256);
#pragma unroll
            for (int T = 0; T < N::NT2; ++T)
                h2[T] = N::TWO ? h1[T] : *reinterpret_cast<const f32x4 *>(a.acts_in + act_base + (size_t)(N::NT1 + T) * 256);
        }
#if LTR_F16X2
        else if (DWH) {
            if (N::H1T > N::NT1) h1[N::H1T - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_wx_hx<N::NT1, N::XT, LDH>(wrsrc, N::W1H_OFF * 4, lane_off, Xhi + my_row * LDH + 4 * q, Xlo + my_row * LDH + 4 * q, h1,
                                           ldexpf(w3s[N::NT2 * 16 + 4], exx - 14));
        }
#endif
        else {
            f32x4 xb[N::XT];
#pragma unroll
            for (int T = 0; T < N::XT; ++T)
                xb[T] = *reinterpret_cast<const f32x4 *>(Xs + my_row * LD + 16 * T + 4 * q);
            if (N::H1T > N::NT1) h1[N::H1T - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#if LTR_F16X2
            if (!LTR_SKIP(a, 16)) gemm_wx_h<N::NT1, N::XT>(wrsrc, N::W1H_OFF * 4, lane_off, xb, h1, w3s[N::NT2 * 16 + 4]);
#else
            if (!LTR_SKIP(a, 16)) gemm_wx<N::NT1, N::XT>(wrsrc, N::W1F_OFF * 4, lane_off, xb, h1);
#endif
            else
#pragma unroll
                for (int To = 0; To < N::NT1; ++To) h1[To] = xb[To];
        }
        if (!saved) activate<N::A1, N::H1, N::NT1, MODE == MODE_FWD>(h1, q, a, 0, a.keep1, gdoc);
        if (MODE == MODE_FWD && a.acts_out) {
#pragma unroll
            for (int T = 0; T < N::NT1; ++T) *reinterpret_cast<f32x4 *>(a.acts_out + act_base + (size_t)T * 256) = h1[T];
        }
        {   // ones feature at index H1 (carries b2 through fc2 and db2 through dW2)
            constexpr int Tn = N::H1 / 16, p = N::H1 % 16;
            h1[Tn][p % 4] = (q == p / 4) ? 1.f : h1[Tn][p % 4];
        }
#if LTR_F16X2
        if (DWH) {
            float m = 0.f;
#pragma unroll
            for (int T = 0; T < N::NT1; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(h1[T][r]));
            m = wave_allmax(m);
            if (lane == 0) exch[8 + w] = m;
        }
#endif
        LTR_STAMP(2)
        // ---- fc2
        if (saved) {
        } else if (N::TWO) {       // no fc2: the last hidden layer IS h1
#pragma unroll
            for (int To = 0; To < N::NT2; ++To) h2[To] = h1[To];
        } else {
#if LTR_F16X2
            if (!LTR_SKIP(a, 8)) gemm_wx_h<N::NT2, N::H1T>(wrsrc, N::W2H_OFF * 4, lane_off, h1, h2, w3s[N::NT2 * 16 + 5]);
#else
            if (!LTR_SKIP(a, 8)) gemm_wx<N::NT2, N::H1T>(wrsrc, N::W2F_OFF * 4, lane_off, h1, h2);
#endif
            else
#pragma unroll
                for (int To = 0; To < N::NT2; ++To) h2[To] = h1[To];
            activate<N::A2, N::H2, N::NT2, MODE == MODE_FWD>(h2, q, a, 1, a.keep2, gdoc);
            if (MODE == MODE_FWD && a.acts_out) {
#pragma unroll
                for (int T = 0; T < N::NT2; ++T)
                    *reinterpret_cast<f32x4 *>(a.acts_out + act_base + (size_t)(N::NT1 + T) * 256) = h2[T];
            }
        }
        LTR_STAMP(3)
        // ---- fc3: s = w3 . h2 + b3, reduced over the 4 q-lanes of each document
        {
            float p0 = 0.f;
#pragma unroll
            for (int To = 0; To < N::NT2; ++To) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w3s + 16 * To + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) p0 += wv[r] * h2[To][r];
            }
            p0 += __shfl_xor(p0, 16, 64);
            p0 += __shfl_xor(p0, 32, 64);
            if (q == 0) sc[my_row] = p0 + w3s[N::NT2 * 16];
        }
        if (MODE == MODE_FWD) {
            __syncthreads();
            if (tid < kTileDocs && doc_base + tid < a.n_docs) a.scores_out[doc_base + tid] = sc[tid];
            continue;
        }
        LTR_STAMP(4)
        // ---- listwise loss on the LDS-resident scores (fused) -> dsc
        if (MODE == MODE_FUSED && LTR_SKIP(a, 1)) {
            __syncthreads();
            if (tid < kTileDocs) dsc[tid] = 1e-3f * sc[tid];
        } else if (MODE == MODE_FUSED) {
            __syncthreads();
            int tl = tid;                              // laundered: keeps the group geometry below from being hoisted out
            if (DEHOIST) asm volatile("" : "+v"(tl));  // of the tile loop and spilled (see make_group)
            const int S_ = ST ? ST : a.S;
            const int group = 4 * S_;                  // S in {32, 64, 128}: 4 threads per document row
            const int gid = tl / group;
            const SlateGroup g = make_group(S_, group, scratch + gid * (group + 32), tl);
            const int so = gid * S_;
            const long long slate = (long long)st * (kTileDocs / S_) + gid;
            float loss;
            if (LOSS == 0) {
                // one slate per 4 S threads (four lanes per document row); wave-private prologue, one barrier inside
                auto stamp_fn = [&](int k) { LTR_STAMP(k) };
                auto run = [&](auto s_tag, auto nw_tag, auto stamper) {
                    constexpr int SS = decltype(s_tag)::value, NWS = decltype(nw_tag)::value;
                    const int gi = tl / (64 * NWS), so2 = gi * SS;
                    const float *scp = sc + so2;
                    return approx_ndcg_fused_shared<SS, NWS, false>(tl - gi * 64 * NWS, [&](int j) { return scp[j]; }, sc + so2, yl + so2, gn + so2,
                                                                    gg + so2, uu + so2, xt + so2, mk + so2, scratch + gi * 32, scratch + 256, a.alpha, a.eps,
                                                                    a.gscale, [&](int i, float v) { dsc[so2 + i] = v; }, stamper);
                };
                if constexpr (ST == 128) loss = run(std::integral_constant<int, 128>(), std::integral_constant<int, 8>(), stamp_fn);
                else if (a.S == 128) loss = run(std::integral_constant<int, 128>(), std::integral_constant<int, 8>(), stamp_fn);
                else if (a.S == 64) loss = run(std::integral_constant<int, 64>(), std::integral_constant<int, 4>(), NoStamp());
                else loss = run(std::integral_constant<int, 32>(), std::integral_constant<int, 2>(), NoStamp());
            }
            else if (LOSS == 1)
                loss = listnet_slate(g, yl + so, sc + so, a.apply_sigmoid != 0, a.gscale, true,
                                     [&](int i, float v) { dsc[so + i] = v; });
            else {
                LambdaLds L;
                L.sc = sc + so; L.yl = yl + so; L.gn = gn + so; L.w1 = gg + so; L.invd = uu + so; L.delta = mk + so;
                L.rk = reinterpret_cast<int *>(xt + so);
                float count;
                loss = lambda_slate<-1>(g, L, a.lp, a.gscale, true, &count, [&](int i, float v) { dsc[so + i] = v; });
                if (g.t == 0 && slate < a.B && a.slate_count) a.slate_count[slate] = count;
            }
            if (g.t == 0 && slate < a.B) a.slate_loss[slate] = loss;
        }
        __syncthreads();

        LTR_STAMP(5)
        // ---- pull the NEXT super-tile of X into L2 while this one is in its backward (one dword per 128-B line
        //      per lane; the value is only kept alive until the end of the iteration so the load is waited for)
        float pf = 0.f;
        if (!XPREF) {
            const long long nb = (long long)(st + gridDim.x) * kTileDocs;
            const long long fl = nb * N::F + (long long)tid * 32;          // float index of this lane's line
            if (st + (int)gridDim.x < a.n_super && fl < a.n_docs * N::F) pf = a.X[fl];
            constexpr int LINES = (kTileDocs * N::F + 31) / 32;
            if (tid + kThreads < LINES) {
                const long long fl2 = fl + (long long)kThreads * 32;
                if (st + (int)gridDim.x < a.n_super && fl2 < a.n_docs * N::F) pf += a.X[fl2];
            }
        }
        // ---- backward through fc3: dw3 += ds * h2 (sum over documents = lanes d), dz2 = ds * w3 * act2'(h2)
        // documents past the end of the batch must not contribute to any gradient
        const float ds0 = gdoc < a.n_docs ? dsc[my_row] : 0.f;
        db3 += wave_allsum((q == 0) ? ds0 : 0.f);
        const float slope = a.dropout ? a.drop_scale : 1.f;              // d relu-dropout / dz where it is live
        const float ds2 = (N::A2 == ACT_RELU_DROP) ? ds0 * slope : ds0;
#pragma unroll
        for (int To = 0; To < N::NT2; ++To) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w3s + 16 * To + 4 * q);
            const f32x4 sv = row_sum4_to_lane15(h2[To] * ds0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // fire-and-forget ds_add_f32 into this wave's private slot: single adder per address, program
                // order across tiles -> deterministic, and no read-modify-write round trip on the critical path
                if (d == 15)
                    __hip_atomic_fetch_add(&dw3[w * N::NT2 * 16 + 16 * To + 4 * q + r], sv[r], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                h2[To][r] = apply_act_grad<N::A2>(ds2 * wv[r], h2[To][r]);
            }
        }
        // h2 now holds dz2.
#if LTR_H1_BITS
        // From here on h1 is needed as a VALUE only by the dW2 staging; the dz1 mask needs just sign(h1) for
        // ReLU layers: keep NT1*4 sign bits (2 VGPRs) so the 36 h1 registers are dead during the dh1 GEMM,
        // the register-pressure peak of the kernel.
        unsigned hbits[(N::NT1 * 4 + 31) / 32] = {};
        if (N::A1 == ACT_RELU_DROP) {
#pragma unroll
            for (int To = 0; To < N::NT1; ++To)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    hbits[(To * 4 + r) >> 5] |= (h1[To][r] > 0.f ? 1u : 0u) << ((To * 4 + r) & 31);
        }
#endif
        LTR_STAMP(6)
#if LTR_F16X2
        if (DWH && !N::TWO) {
            // operand scales of dW2 (running maxima): h1 from the per-wave maxima published before the loss barrier; dz2 from
            // the bound |dz2| <= |d score| slope max|w3| (the scores' gradients are LDS-resident: every wave takes their max)
            float hm = 1.f;
#pragma unroll
            for (int i = 0; i < kWaves; ++i) hm = fmaxf(hm, exch[8 + i]);
            exh = grow_exp(exh, hm);
            const float dm = wave_allmax(fmaxf(fabsf(dsc[lane]), fabsf(dsc[lane + 64])));
            exd2 = grow_exp(exd2, dm * slope * w3max);
            if (exd2 + exh != E2) {                        // the scale grew: bring the accumulators along (exact: a power of two)
                // (a branch-free multiply by 1 and SGPR-resident exponents cut the scratch 276 -> 248 B but moved the reloads
                //  into the dW1 loop: 112.8 k -> 129.6 k cycles per tile, profiles/r03_variant_ab.json)
                const float f = ldexpf(1.f, E2 - (exd2 + exh));
#pragma unroll
                for (int n = 0; n < N::TW2; ++n) accW2[n] *= f;
                E2 = exd2 + exh;
            }
            const float sd = ldexpf(1.f, 14 - exd2), sh = ldexpf(1.f, 14 - exh);
            const int lb = (4 * q + (d >> 2)) * LDH + 4 * (d & 3);          // lane base of the transposing reads
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c > 0) __syncthreads();   // chunk 0 fully consumed
                if (chunk == c) {             // images of this 64-document chunk: dz2 hi | dz2 lo | h1 hi | h1 lo, [64][LDH] each
#pragma unroll
                    for (int To = 0; To < N::NT2; ++To) {
                        u32x2 hi, lo;
                        split4(h2[To], sd, hi, lo);
                        *reinterpret_cast<u32x2 *>(Sg + crow * LDH + 16 * To + 4 * q) = hi;
                        *reinterpret_cast<u32x2 *>(Sg + (64 + crow) * LDH + 16 * To + 4 * q) = lo;
                    }
#pragma unroll
                    for (int T = 0; T < N::H1T; ++T) {
                        u32x2 hi, lo;
                        split4(h1[T], sh, hi, lo);
                        *reinterpret_cast<u32x2 *>(Sg + (128 + crow) * LDH + 16 * T + 4 * q) = hi;
                        *reinterpret_cast<u32x2 *>(Sg + (192 + crow) * LDH + 16 * T + 4 * q) = lo;
                    }
                }
                __syncthreads();
                dw_chunk_h<N::TW2, N::NT2, N::H1T, N::BH2, LDH, 2>(w, accW2, Sg + lb, Sg + 64 * LDH + lb, Sg + 128 * LDH + lb,
                                                                   Sg + 192 * LDH + lb);
            }
        } else
#endif
        // ---- dW2 += dz2^T [h1 | 1] over the two 64-document chunks (tiles of waves 0-3, then waves 4-7)
#pragma unroll
        for (int c = 0; c < (N::TWO ? 0 : 2); ++c) {
            if (c > 0) __syncthreads();   // chunk 0 fully consumed
            if (chunk == c) {
#pragma unroll
                for (int To = 0; To < N::NT2; ++To)
                    *reinterpret_cast<f32x4 *>(Ds + crow * LD + 16 * To + ((4 * q) ^ (LTR_STG_SWZ ? 4 * ((crow >> 1) & 3) : 0))) = h2[To];
#pragma unroll
                for (int T = 0; T < N::H1T; ++T)
                    *reinterpret_cast<f32x4 *>(Hs + crow * LD + 16 * T + ((4 * q) ^ (LTR_STG_SWZ ? 4 * ((crow >> 1) & 3) : 0))) = h1[T];
            }
#ifdef LTR_STAMPS_DW
            if (c == 0) { LTR_STAMP(10) } else { LTR_STAMP(13) }
#endif
            __syncthreads();
#ifdef LTR_STAMPS_DW
            if (c == 0) { LTR_STAMP(11) } else { LTR_STAMP(14) }
#endif
            if (!LTR_SKIP(a, 2))
                dw_chunk<N::TW2, N::NT2, N::H1T, N::BH2, LD>(w, accW2, Ds + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (q >> 1) : 0)),
                                                             Hs + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (q >> 1) : 0)),
                                                             Ds + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (2 + (q >> 1)) : 0)),
                                                             Hs + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (2 + (q >> 1)) : 0)));
#ifdef LTR_STAMPS_DW
            if (c == 0) { LTR_STAMP(12) }
#endif
        }
        LTR_STAMP(7)
        // ---- dh1^T = W2^T dz2^T  (A = packed W2^T fragments, B = dz2 registers), then
        //      dz1 = dh1 * act1'(h1), features >= H1 (incl. the ones feature) zeroed
        f32x4 dz1[N::NT1];
        if (N::TWO) {       // the backward through fc3 above already produced dz1 (A2 = A1, "h2" = h1)
#pragma unroll
            for (int To = 0; To < N::NT1; ++To) dz1[To] = h2[To];
        }
#if LTR_F16X2
        else if (!LTR_SKIP(a, 4)) gemm_wx_h<N::NT1, N::NT2>(wrsrc, N::W2TH_OFF * 4, lane_off, h2, dz1, w3s[N::NT2 * 16 + 6]);
#else
        else if (!LTR_SKIP(a, 4)) gemm_wx<N::NT1, N::NT2>(wrsrc, N::W2T_OFF * 4, lane_off, h2, dz1);
#endif
        else
#pragma unroll
            for (int To = 0; To < N::NT1; ++To) dz1[To] = h2[To < N::NT2 ? To : 0];
#pragma unroll
        for (int To = 0; To < (N::TWO ? 0 : N::NT1); ++To)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gsc = (N::A1 == ACT_RELU_DROP) ? dz1[To][r] * slope : dz1[To][r];
#if LTR_H1_BITS
                float v = (N::A1 == ACT_RELU_DROP)
                              ? (((hbits[(To * 4 + r) >> 5] >> ((To * 4 + r) & 31)) & 1u) ? gsc : 0.f)
                              : apply_act_grad<N::A1>(gsc, h1[To][r]);
#else
                float v = apply_act_grad<N::A1>(gsc, h1[To][r]);
#endif
                if (16 * To + 16 > N::H1) v = (16 * To + 4 * q + r < N::H1) ? v : 0.f;   // last tile only
                dz1[To][r] = v;
            }
        LTR_STAMP(8)
#if LTR_F16X2
        if (DWH) {
            float m = 0.f;
#pragma unroll
            for (int To = 0; To < N::NT1; ++To)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(dz1[To][r]));
            m = wave_allmax(m);
            if (lane == 0) exch[16 + w] = m;
            __syncthreads();              // every wave is done reading the dW2 images; the dz1 maxima are published
            float dm = 0.f;
#pragma unroll
            for (int i = 0; i < kWaves; ++i) dm = fmaxf(dm, exch[16 + i]);
            exd1 = grow_exp(exd1, dm);
            if (exd1 + exx != E1) {
                const float f = ldexpf(1.f, E1 - (exd1 + exx));
#pragma unroll
                for (int n = 0; n < N::TW1; ++n) accW1[n] *= f;
                E1 = exd1 + exx;
            }
            const float sd = ldexpf(1.f, 14 - exd1);
#pragma unroll
            for (int To = 0; To < N::NT1; ++To) {          // dz1 of all 128 documents: hi image | lo image, [128][LDH] each
                u32x2 hi, lo;
                split4(dz1[To], sd, hi, lo);
                *reinterpret_cast<u32x2 *>(Sg + my_row * LDH + 16 * To + 4 * q) = hi;
                *reinterpret_cast<u32x2 *>(Sg + (128 + my_row) * LDH + 16 * To + 4 * q) = lo;
            }
            __syncthreads();
            // next super-tile's X slice -> registers now (register-prefetch kernels); it lands during the dW1 MFMAs below
            if (XPREF && st + (int)gridDim.x < a.n_super)
                load_x_tile<N>(xn, a, (long long)(st + gridDim.x) * kTileDocs + 16 * w, lane);
            const int lb = (4 * q + (d >> 2)) * LDH + 4 * (d & 3);
            dw_chunk_h<N::TW1, N::NT1, N::XT, N::BH1, LDH, 4>(w, accW1, Sg + lb, Sg + 128 * LDH + lb, Xhi + lb, Xlo + lb);
        } else {
#endif
        // next super-tile's X slice -> registers now (h1/h2 are dead); it lands during the dW1 MFMAs below
        if (MODE != MODE_FWD && XPREF && st + (int)gridDim.x < a.n_super)
            load_x_tile<N>(xn, a, (long long)(st + gridDim.x) * kTileDocs + 16 * w, lane);
        // ---- dW1 += dz1^T [x | 1]; B operand straight from the X tile in LDS.
#ifndef LTR_DW1_SINGLE
#define LTR_DW1_SINGLE 1    // all nets: dz1 of the whole tile staged at once (one barrier pair instead of two; -1 % on the backward)
#endif
        if (N::H1 <= 64 || LTR_DW1_SINGLE) {
            // dW2 is done with the two staging buffers, which are contiguous (Ds..Hs = one [128][LD] region): every
            // wave stages its dz1 tile at once, one barrier, then both 64-document halves back to back.  (-3.5 % on
            // the 136-64-32 net; on the register-bound 136-136-136 kernels the chunked form below is faster.)
            __syncthreads();              // every wave is done reading Ds / Hs (dW2 chunk 1)
#pragma unroll
            for (int To = 0; To < N::NT1; ++To)
                *reinterpret_cast<f32x4 *>(Ds + my_row * LD + 16 * To + ((4 * q) ^ (LTR_STG_SWZ ? 4 * ((my_row >> 1) & 3) : 0))) = dz1[To];
            __syncthreads();
            if (!LTR_SKIP(a, 2)) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    dw_chunk<N::TW1, N::NT1, N::XT, N::BH1, LD>(w, accW1, Ds + (64 * c + q) * LD + (d ^ (LTR_STG_SWZ ? 4 * (q >> 1) : 0)),
                                                                Xs + (64 * c + q) * LD + d,
                                                                Ds + (64 * c + q) * LD + (d ^ (LTR_STG_SWZ ? 4 * (2 + (q >> 1)) : 0)),
                                                                Xs + (64 * c + q) * LD + d);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                __syncthreads();              // previous chunk's readers are done with Ds
                if (chunk == c) {
#pragma unroll
                    for (int To = 0; To < N::NT1; ++To)
                        *reinterpret_cast<f32x4 *>(Ds + crow * LD + 16 * To + ((4 * q) ^ (LTR_STG_SWZ ? 4 * ((crow >> 1) & 3) : 0))) = dz1[To];
                }
                __syncthreads();
                if (!LTR_SKIP(a, 2))
                    dw_chunk<N::TW1, N::NT1, N::XT, N::BH1, LD>(w, accW1, Ds + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (q >> 1) : 0)),
                                                                Xs + (64 * c + q) * LD + d,
                                                                Ds + q * LD + (d ^ (LTR_STG_SWZ ? 4 * (2 + (q >> 1)) : 0)),
                                                                Xs + (64 * c + q) * LD + d);
            }
        }
#if LTR_F16X2
        }
#endif
        LTR_STAMP(9)
        asm volatile("" ::"v"(pf));   // keep the prefetch load alive (and waited for) until here
    }

    if (MODE == MODE_FWD) return;
    // ---- per-workgroup partial gradients -> workspace
    float *part = a.partials + (size_t)blockIdx.x * N::PART;
#if LTR_F16X2
    const float us1 = DWH ? ldexpf(1.f, E1 - 28) : 1.f, us2 = DWH ? ldexpf(1.f, E2 - 28) : 1.f;   // accumulators -> true scale
#else
    const float us1 = 1.f, us2 = 1.f;
#endif
    dw_store<N::TW1, N::NT1, N::XT, N::BH1>(w, accW1, part + N::P_W1, q, d, us1);
    if (!N::TWO) dw_store<N::TW2, N::NT2, N::H1T, N::BH2>(w, accW2, part + N::P_W2, q, d, us2);
    __syncthreads();
    for (int j = tid; j < N::NT2 * 16; j += kThreads) {
        float s = 0.f;
        for (int ww = 0; ww < kWaves; ++ww) s += dw3[ww * N::NT2 * 16 + j];
        part[N::P_W3 + j] = s;
    }
    // db3: every lane of a wave holds the wave's sum; combine the waves through LDS in fixed order
    if (lane == 0) scratch[w] = db3;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int ww = 0; ww < kWaves; ++ww) s += scratch[ww];
        part[N::P_B3] = s;
    }
}


// The keep mask the pipeline's counter-based dropout stream produces (tests / reproducibility tooling).
__global__ void dropout_mask_kernel(unsigned long long seed, int layer, long long n_docs, int H, unsigned thr16, uint8_t *out) {
    const long long e = ltr_block_id() * blockDim.x + threadIdx.x;
    if (e >= n_docs * H) return;
    const long long doc = e / H;
    const int n = (int)(e - doc * H);
    if (thr16) out[e] = ((keep_word(seed, layer + 2, doc, n >> 1) >> (16 * (n & 1))) & 0xffffu) >= thr16 ? 1 : 0;
    else out[e] = (keep_word(seed, layer, doc, n >> 5) >> (n & 31)) & 1u;
}

// `dropout` argument of the launchers: 0 = off; bit 0 set = on with p = the float whose bits are (dropout & ~1) -- 0 there (the
// plain value 1) means the reference's p = 0.5.  p = 0.5 keeps the one-bit-per-unit stream, any other p draws 16 bits per unit.
static inline void decode_dropout(int dropout, int &on, unsigned &thr16, float &scale) {
    on = dropout & 1;
    unsigned bits = (unsigned)dropout & ~1u;
    float p = 0.5f;
    if (bits) memcpy(&p, &bits, sizeof p);
    if (!(p > 0.f)) on = 0;
    if (!(p < 1.f)) p = 0.5f;           // rejected by fill_common
    thr16 = p == 0.5f ? 0u : (unsigned)(p * 65536.f + 0.5f);
    if (thr16 > 65535u) thr16 = 65535u;
    scale = 1.f / (1.f - p);
}

// Pack nn.Linear parameters into lane-ordered MFMA A-fragments (once per optimizer step; 37k params).
template <class N>
__global__ void pack_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                            const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
                            float *__restrict__ packed, int Fr, int H1r, int H2r) {
    // Fr / H1r / H2r: the LOGICAL layer widths (<= the compiled N::F / N::H1 / N::H2): a narrower network runs zero-padded
    // on the compiled geometry -- padded inputs and hidden units carry zero weights, so they contribute nothing, and the
    // bias columns stay at the compiled positions N::F / N::H1.
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    // augmented weights: Waug[n][f] = W[n][f] (f < in), b[n] (f == in), 0 otherwise; rows n >= out are 0
    auto w1aug = [&](int n, int f) { return n < H1r ? (f < Fr ? W1[n * Fr + f] : (f == N::F ? b1[n] : 0.f)) : 0.f; };
    auto w2aug = [&](int n, int f) { return n < H2r ? (f < H1r ? W2[n * H1r + f] : (f == N::H1 ? b2[n] : 0.f)) : 0.f; };
    for (int e = gt; e < N::NT1 * N::XT * 256; e += stride) {
        const int s = e & 3, lane = (e >> 2) & 63, tile = e >> 8, To = tile / N::XT, T = tile - To * N::XT;
        packed[N::W1F_OFF + e] = w1aug(16 * To + (lane & 15), 16 * T + 4 * (lane >> 4) + s);
    }
    for (int e = gt; e < (N::TWO ? 0 : N::NT2 * N::H1T * 256); e += stride) {
        const int s = e & 3, lane = (e >> 2) & 63, tile = e >> 8, To = tile / N::H1T, T = tile - To * N::H1T;
        packed[N::W2F_OFF + e] = w2aug(16 * To + (lane & 15), 16 * T + 4 * (lane >> 4) + s);
    }
    // W2^T fragments for dh1: output tile Ti over fc2 INPUT features, k over fc2 OUTPUT features
    for (int e = gt; e < (N::TWO ? 0 : N::NT1 * N::NT2 * 256); e += stride) {
        const int s = e & 3, lane = (e >> 2) & 63, tile = e >> 8, Ti = tile / N::NT2, T = tile - Ti * N::NT2;
        const int in = 16 * Ti + (lane & 15), o = 16 * T + 4 * (lane >> 4) + s;
        packed[N::W2T_OFF + e] = (o < H2r && in < H1r) ? W2[o * H1r + in] : 0.f;
    }
#if LTR_F16X2
    // ---- f16 hi / lo fragments of the same three matrices, each scaled by ONE power of two so that its largest |entry| lands
    //      in [2^13, 2^14) (every workgroup recomputes the three maxima: 56 k reads, nothing to synchronise)
    __shared__ float red[3][256];
    float mx[3] = {0.f, 0.f, 0.f};
    for (int e = threadIdx.x; e < N::H1 * (N::F + 1); e += blockDim.x) mx[0] = fmaxf(mx[0], fabsf(w1aug(e / (N::F + 1), e % (N::F + 1))));
    for (int e = threadIdx.x; e < (N::TWO ? 0 : N::H2 * (N::H1 + 1)); e += blockDim.x) {
        const float v = fabsf(w2aug(e / (N::H1 + 1), e % (N::H1 + 1)));
        mx[1] = fmaxf(mx[1], v);
        if (e % (N::H1 + 1) < N::H1) mx[2] = fmaxf(mx[2], v);
    }
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = mx[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = fmaxf(red[k][threadIdx.x], red[k][threadIdx.x + o]);
        __syncthreads();
    }
    float scl[3], inv[3];
    for (int k = 0; k < 3; ++k) {
        int ex = __builtin_amdgcn_frexp_expf(red[k][0]);
        ex = ex < -100 ? -100 : ex;
        scl[k] = ldexpf(1.f, 14 - ex);
        inv[k] = ldexpf(1.f, ex - 14);
    }
    uint16_t *ph = reinterpret_cast<uint16_t *>(packed);
    auto put = [&](size_t off_halfs, int hh, float v) {          // half index hh inside a section -> hi or lo piece of v
        const _Float16 hi = (_Float16)v;
        const _Float16 out = ((hh >> 9) & 1) ? (_Float16)(v - (float)hi) : hi;
        ph[off_halfs + hh] = __builtin_bit_cast(uint16_t, out);
    };
    auto geom = [](int hh, int NT, int &To, int &i, int &f) {   // (row tile, row in tile, feature) of half hh
        const int j = hh & 7, lane = (hh >> 3) & 63, step = hh >> 10, P = step / NT;
        To = step % NT;
        i = lane & 15;
        f = 32 * P + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    };
    for (int hh = gt; hh < N::KP1 * N::NT1 * 1024; hh += stride) {
        int To, i, f;
        geom(hh, N::NT1, To, i, f);
        put((size_t)N::W1H_OFF * 2, hh, w1aug(16 * To + i, f) * scl[0]);
    }
    for (int hh = gt; hh < (N::TWO ? 0 : N::KP2 * N::NT2 * 1024); hh += stride) {
        int To, i, f;
        geom(hh, N::NT2, To, i, f);
        put((size_t)N::W2H_OFF * 2, hh, w2aug(16 * To + i, f) * scl[1]);
    }
    for (int hh = gt; hh < (N::TWO ? 0 : N::KPT * N::NT1 * 1024); hh += stride) {
        int Ti, i, o;
        geom(hh, N::NT1, Ti, i, o);
        const int in = 16 * Ti + i;
        put((size_t)N::W2TH_OFF * 2, hh, (o < H2r && in < H1r) ? W2[o * H1r + in] * scl[2] : 0.f);
    }
    for (int hh = gt; hh < (N::TWO ? N::KP1 * N::NT1 * 1024 : 0); hh += stride) {
        const int j = hh & 7, lane = (hh >> 3) & 63, step = hh >> 10, tile = step / N::KP1, P = step % N::KP1;
        put((size_t)N::W1B_OFF * 2, hh, w1aug(16 * tile + (lane & 15), 32 * P + 8 * (lane >> 4) + j) * scl[0]);
    }
#endif
    for (int e = gt; e < N::NT2 * 16 + 16; e += stride) {
        float v = e < H2r ? w3[e] : (e == N::NT2 * 16 ? b3[0] : 0.f);
#if LTR_F16X2
        if (e >= N::NT2 * 16 + 4 && e < N::NT2 * 16 + 7) v = inv[e - N::NT2 * 16 - 4];
#endif
        packed[N::W3_OFF + e] = v;
    }
}


// Sum the per-workgroup partials in a fixed order and scatter into the flat gradient
// [W1 (H1 x F) | b1 (H1) | W2 (H2 x H1) | b2 (H2) | w3 (H2) | b3 (1)]  (= nn.Module parameter order).
// ---- TripleLayerNet <-> its folded two-layer form (see TripleFolded) ------------------------------------------------------
constexpr int kTrH1 = 64, kTrH2 = 32;       // TripleLayerNet's hidden widths (tripleLayer.py:8-10); the input width F is a run-time argument
__global__ void triple_fold_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                                   const float *__restrict__ b2, const float *__restrict__ w3, int kTrF, int copies,
                                   float *__restrict__ W1e, float *__restrict__ b1e, float *__restrict__ w3e) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;           // one thread per element of [32][136 + 1]
    if (e >= kTrH2 * (kTrF + 1)) return;
    const int u = e / (kTrF + 1), f = e % (kTrF + 1);
    double acc = 0.0;
    if (f < kTrF) {
        for (int k = 0; k < kTrH1; ++k) acc += (double)W2[u * kTrH1 + k] * (double)W1[k * kTrF + f];
        for (int c = 0; c < copies; ++c) W1e[(c * kTrH2 + u) * kTrF + f] = (float)acc;
    } else {
        for (int k = 0; k < kTrH1; ++k) acc += (double)W2[u * kTrH1 + k] * (double)b1[k];
        for (int c = 0; c < copies; ++c) {
            b1e[c * kTrH2 + u] = (float)(acc + (double)b2[u]);
            w3e[c * kTrH2 + u] = w3[u];
        }
    }
}
__global__ void triple_unfold_kernel(const float *__restrict__ g2, int kTrF, int copies, const float *__restrict__ W1, const float *__restrict__ b1,
                                     const float *__restrict__ W2, float *__restrict__ flat) {
    // g2 = [dW1e R x 136 | db1e R | dw3e R | db3], R = 32 copies;  flat = [dW1 64 x 136 | db1 64 | dW2 32 x 64 | db2 32 | dw3 32 | db3]
    const int R = kTrH2 * copies;
    const float *gW = g2, *gb = g2 + R * kTrF, *gw3 = gb + R, *gb3 = gw3 + R;
    auto G = [&](int u, int f) { double v = 0.0; for (int c = 0; c < copies; ++c) v += (double)gW[(c * kTrH2 + u) * kTrF + f]; return v; };
    auto Gb = [&](int u) { double v = 0.0; for (int c = 0; c < copies; ++c) v += (double)gb[c * kTrH2 + u]; return v; };
    const int nW1 = kTrH1 * kTrF, nb1 = kTrH1, nW2 = kTrH2 * kTrH1, nb2 = kTrH2, nw3 = kTrH2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    int k = e;
    if (k < nW1) {                                   // dW1[h][f] = sum_u W2[u][h] G[u][f]
        const int h = k / kTrF, f = k % kTrF;
        double acc = 0.0;
        for (int u = 0; u < kTrH2; ++u) acc += (double)W2[u * kTrH1 + h] * G(u, f);
        flat[e] = (float)acc;
    } else if ((k -= nW1) < nb1) {                   // db1[h] = sum_u W2[u][h] gb[u]
        double acc = 0.0;
        for (int u = 0; u < kTrH2; ++u) acc += (double)W2[u * kTrH1 + k] * Gb(u);
        flat[e] = (float)acc;
    } else if ((k -= nb1) < nW2) {                   // dW2[u][h] = sum_f G[u][f] W1[h][f] + gb[u] b1[h]
        const int u = k / kTrH1, h = k % kTrH1;
        double acc = Gb(u) * (double)b1[h];
        for (int f = 0; f < kTrF; ++f) acc += G(u, f) * (double)W1[h * kTrF + f];
        flat[e] = (float)acc;
    } else if ((k -= nW2) < nb2) {
        flat[e] = (float)Gb(k);
    } else if ((k -= nb2) < nw3) {
        double v = 0.0;
        for (int c = 0; c < copies; ++c) v += (double)gw3[c * kTrH2 + k];
        flat[e] = (float)v;
    } else if (k - nw3 == 0) {
        flat[e] = gb3[0];
    }
}

template <class N>
__global__ void reduce_grads_kernel(const float *__restrict__ partials, int nparts, float *__restrict__ flat, int Fr, int H1r, int H2r) {
    // 8 lanes per parameter: lane s sums partials s, s+8, s+16, ... (independent loads in flight), then the 8
    // strided sums are combined by a fixed butterfly -> same bits on every run, ~8x less serial latency.
    // The cross-workgroup sum runs in fp64 (37 k parameters x 256 partials: free) so the only fp32 rounding left
    // in a parameter gradient is the per-workgroup MFMA accumulation chain.
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = gt >> 3, sub = gt & 7;
    double s = 0.0;
    const int nparam = N::TWO ? H1r * Fr + H1r + H2r + 1 : H1r * Fr + H1r + H2r * H1r + H2r + H2r + 1;   // logical widths
    if (e < nparam) {
        int off;
        int k = e;
        if (k < H1r * Fr) {
            off = N::P_W1 + (k / Fr) * (N::XT * 16) + (k % Fr);
        } else if ((k -= H1r * Fr) < H1r) {
            off = N::P_W1 + k * (N::XT * 16) + N::F;
        } else if (!N::TWO && (k -= H1r) < H2r * H1r) {
            off = N::P_W2 + (k / H1r) * (N::H1T * 16) + (k % H1r);
        } else if (!N::TWO && (k -= H2r * H1r) < H2r) {
            off = N::P_W2 + k * (N::H1T * 16) + N::H1;
        } else if ((k -= (N::TWO ? H1r : H2r)) < H2r) {
            off = N::P_W3 + k;
        } else {
            off = N::P_B3;
        }
        for (int p = sub; p < nparts; p += 8) s += (double)partials[(size_t)p * N::PART + off];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (e < nparam && sub == 0) flat[e] = (float)s;
}

template <class N>
constexpr size_t pipeline_lds() {
    return sizeof(float) * (size_t)(kTileDocs * N::LD + 2 * kChunkDocs * N::LD + 8 * kTileDocs + N::NT2 * 16 + 16 +
                                    kWaves * N::NT2 * 16 + kThreads + 4 * 32 + 64);
}


inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

template <class N, int MODE, int LOSS, int ST = 0>
int launch_pipeline(const PipeArgs &a, int grid, hipStream_t stream) {
    constexpr size_t lds = pipeline_lds<N>();
    // the dynamic-LDS limit is a per-DEVICE function attribute: remember it per device, not per process
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)slate_pipeline_kernel<N, MODE, LOSS, ST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_done[dev] = true;
    }
    hipLaunchKernelGGL((slate_pipeline_kernel<N, MODE, LOSS, ST>), dim3(grid), dim3(kThreads), lds, stream, a);
    return status();
}

template <class N>
int pipeline_dispatch(int mode, const PipeArgs &a, int grid, hipStream_t stream) {
    if constexpr (N::DS != 1) {          // document-split copies exist in the fused two-layer kernel only (the generic pipeline would
        if (mode != MODE_FUSED) return LTR_ERR_PARAM;      // run every copy on every document)
    }
    switch (mode) {
        case MODE_FWD: if constexpr (N::DS == 1) return launch_pipeline<N, MODE_FWD, 0>(a, grid, stream); else return LTR_ERR_PARAM;
        case MODE_BWD: if constexpr (N::DS == 1) return launch_pipeline<N, MODE_BWD, 0>(a, grid, stream); else return LTR_ERR_PARAM;
        case MODE_BWD_SAVED: if constexpr (N::DS == 1) return launch_pipeline<N, MODE_BWD_SAVED, 0>(a, grid, stream); else return LTR_ERR_PARAM;
        default:
            if constexpr (N::TWO && N::H1 != 64 && N::F == 136) {
                return LTR_ERR_PARAM;    // (136 features: the fused step runs the document-split form on ltr_fcw.h, TripleFolded)
            } else if constexpr (N::TWO && N::H1 == 64) {      // two-layer nets: the feature-partitioned kernel, two workgroups per CU (ltr_fcw.h)
                switch (a.loss_kind) {
                    case 0: return launch_fcw<N, 0>(a, grid, stream);
                    case 1: return launch_fcw<N, 1>(a, grid, stream);
                    default: return launch_fcw<N, 2>(a, grid, stream);
                }
            } else {
                switch (a.loss_kind) {
                    case 0:
                        // a slate-128 instantiation (ONE copy of the loss, fixed geometry) for the narrow nets: 136-64-32 +10.6 %; the
                        // 136-wide kernels, at their register limit, allocate better with the run-time form (0.594 vs 0.579 of the fp32
                        // MFMA peak; profiles/r04_variant_ab.json)
                        if constexpr (LTR_PIPE_ST128 && N::H1 <= LTR_PIPE_ST128_MAXH) {
                            if (a.S == 128) return launch_pipeline<N, MODE_FUSED, 0, 128>(a, grid, stream);
                        }
                        return launch_pipeline<N, MODE_FUSED, 0, 0>(a, grid, stream);
                    case 1: return launch_pipeline<N, MODE_FUSED, 1>(a, grid, stream);
                    default: return launch_pipeline<N, MODE_FUSED, 2>(a, grid, stream);
                }
            }
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int ltr_net_info(int net, int32_t *info) {
    if (!info) return LTR_ERR_NULL;
    LTR_FOR_NET(net, (info[0] = NET::F, info[1] = NET::H1, info[2] = NET::H2, info[3] = NET::NPARAM,
                      info[4] = NET::PACKED, info[5] = NET::PART, info[7] = (int32_t)pipeline_lds<NET>()))
    info[6] = kTileDocs;
    return LTR_OK;
}

int ltr_fused_grid(int net, int n_cus) {
    if (n_cus < 1) return LTR_ERR_PARAM;
    if (net < LTR_NET_DOUBLE || net > LTR_NET_TRIPLE_FOLDED_32_64) return LTR_ERR_PARAM;
    if (net == LTR_NET_TWO_LAYER_64H || net == LTR_NET_TRIPLE_FOLDED) return 2 * n_cus;       /* 256-thread workgroups, two per CU (ltr_fcw.h) */
    return n_cus;
}

int ltr_dropout_keep_mask_p(uint64_t seed, int layer, int64_t n_docs, int H, float p, uint8_t *out, void *stream) {
    if (!out) return LTR_ERR_NULL;
    if (n_docs < 0 || H < 1 || H > 4096 || layer < 0 || layer > 1) return LTR_ERR_SHAPE;
    if (!(p > 0.f && p < 1.f)) return LTR_ERR_PARAM;
    const long long n = n_docs * H;
    if (n == 0) return LTR_OK;
    unsigned thr16 = p == 0.5f ? 0u : (unsigned)(p * 65536.f + 0.5f);
    if (thr16 > 65535u) thr16 = 65535u;
    hipLaunchKernelGGL(dropout_mask_kernel, ltr_grid((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                       layer, (long long)n_docs, H, thr16, out);
    return status();
}

int ltr_dropout_keep_mask(uint64_t seed, int layer, int64_t n_docs, int H, uint8_t *out, void *stream) {
    if (!out) return LTR_ERR_NULL;
    if (n_docs < 0 || H < 1 || H > 4096 || layer < 0 || layer > 1) return LTR_ERR_SHAPE;
    const long long n = n_docs * H;
    if (n == 0) return LTR_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, ltr_grid((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                       layer, (long long)n_docs, H, 0u, out);
    return status();
}

int ltr_mlp_pack_sub(int net, int f, int h1, int h2, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *w3, const float *b3, float *packed, void *stream) {
    if (!W1 || !b1 || !w3 || !b3 || !packed) return LTR_ERR_NULL;
    if ((!W2 || !b2) && net != LTR_NET_TWO_LAYER_64H && net != LTR_NET_TRIPLE_FOLDED && net != LTR_NET_TRIPLE_FOLDED_32 && net != LTR_NET_TRIPLE_FOLDED_32_64) return LTR_ERR_NULL;     /* two-layer nets have no fc2 */
    if (!aligned16(packed)) return LTR_ERR_ALIGN;
    LTR_FOR_NET(net, {
        if (f < 1 || f > NET::F || h1 < 1 || h1 > NET::H1 || h2 < 1 || h2 > NET::H2) return LTR_ERR_SHAPE;
        hipLaunchKernelGGL(pack_kernel<NET>, dim3(64), dim3(256), 0, (hipStream_t)stream, W1, b1, W2, b2, w3, b3, packed, f, h1, h2);
    })
    return status();
}

int ltr_mlp_pack(int net, const float *W1, const float *b1, const float *W2, const float *b2, const float *w3,
                 const float *b3, float *packed, void *stream) {
    int32_t info[8];
    if (int rc = ltr_net_info(net, info)) return rc;
    return ltr_mlp_pack_sub(net, info[0], info[1], info[2], W1, b1, W2, b2, w3, b3, packed, stream);
}

// diagnostic builds (-DLTR_STAMPS): where the phase stamps of the next launches go (NULL: none)
static unsigned long long *g_stamps = nullptr;
static int g_stamp_tile = 0;
int ltr_debug_set_stamps(void *buf, int tile) {
    g_stamps = static_cast<unsigned long long *>(buf);
    g_stamp_tile = tile;
#ifdef LTR_STAMPS
    return 1;
#else
    return 0;
#endif
}

static int fill_common(PipeArgs &a, int net, const float *X, int64_t n_docs, const float *packed, int dropout,
                       uint64_t seed, const uint8_t *keep1, const uint8_t *keep2) {
    if (!X || !packed) return LTR_ERR_NULL;
    if (net < LTR_NET_DOUBLE || net > LTR_NET_TRIPLE_FOLDED_32_64) return LTR_ERR_PARAM;
    if (n_docs < 0 || n_docs > ((int64_t)1 << 37)) return LTR_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(packed)) return LTR_ERR_ALIGN;
    a = PipeArgs{};
    a.X = X;
    a.n_docs = n_docs;
    a.packed = packed;
    {
        unsigned bits = (unsigned)dropout & ~1u;
        float p = 0.5f;
        if (bits) memcpy(&p, &bits, sizeof p);
        if ((dropout & 1) && !(p >= 0.f && p < 1.f)) return LTR_ERR_PARAM;
    }
    decode_dropout(dropout, a.dropout, a.drop_thr16, a.drop_scale);
    a.seed = seed;
    a.keep1 = keep1;
    a.keep2 = keep2;
    a.n_super = (int)((n_docs + kTileDocs - 1) / kTileDocs);
#ifdef LTR_DIAG
    static const int dbg = getenv("LTR_DEBUG_SKIP") ? atoi(getenv("LTR_DEBUG_SKIP")) : 0;
    a.debug_skip = dbg;
#endif
    a.stamps = g_stamps;
    a.stamp_tile = g_stamp_tile;
    return LTR_OK;
}

int ltr_mlp_forward(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                    const uint8_t *keep1, const uint8_t *keep2, float *scores, int grid, void *stream) {
    PipeArgs a;
    if (int rc = fill_common(a, net, X, n_docs, packed, dropout, seed, keep1, keep2)) return rc;
    if (!scores) return LTR_ERR_NULL;
    if (grid < 1) return LTR_ERR_PARAM;
    if (a.n_super == 0) return LTR_OK;
    a.scores_out = scores;
    if (grid > a.n_super) grid = a.n_super;
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_FWD, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

int ltr_mlp_backward(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                     const uint8_t *keep1, const uint8_t *keep2, const float *dscores, float *partials, int grid,
                     void *stream) {
    PipeArgs a;
    if (int rc = fill_common(a, net, X, n_docs, packed, dropout, seed, keep1, keep2)) return rc;
    if (!dscores || !partials) return LTR_ERR_NULL;
    if (grid < 1) return LTR_ERR_PARAM;
    a.dscores_in = dscores;
    a.partials = partials;
    if (a.dropout && a.drop_thr16 && !keep1) return LTR_ERR_PARAM;     /* p != 0.5: ltr_mlp_forward_save / ltr_mlp_backward_saved */
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_BWD, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

int64_t ltr_mlp_acts_floats(int net, int64_t n_docs) {
    if (n_docs < 0) return LTR_ERR_SHAPE;
    const int64_t tiles = (n_docs + kTileDocs - 1) / kTileDocs * kWaves;        // 16-document wave tiles, whole super-tiles
    LTR_FOR_NET(net, return tiles * (NET::NT1 + (NET::TWO ? 0 : NET::NT2)) * 256)
    return LTR_ERR_PARAM;
}

int ltr_mlp_forward_save(int net, const float *X, int64_t n_docs, const float *packed, int dropout, uint64_t seed,
                         const uint8_t *keep1, const uint8_t *keep2, float *scores, float *acts, int grid, void *stream) {
    PipeArgs a;
    if (int rc = fill_common(a, net, X, n_docs, packed, dropout, seed, keep1, keep2)) return rc;
    if (!scores || !acts) return LTR_ERR_NULL;
    if (!aligned16(acts)) return LTR_ERR_ALIGN;
    if (grid < 1) return LTR_ERR_PARAM;
    if (a.n_super == 0) return LTR_OK;
    a.scores_out = scores;
    a.acts_out = acts;
    if (grid > a.n_super) grid = a.n_super;
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_FWD, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

int ltr_mlp_backward_saved(int net, const float *X, int64_t n_docs, const float *packed, int dropout, const float *acts,
                           const float *dscores, float *partials, int grid, void *stream) {
    PipeArgs a;
    if (int rc = fill_common(a, net, X, n_docs, packed, dropout, 0, nullptr, nullptr)) return rc;
    if (!dscores || !partials || !acts) return LTR_ERR_NULL;
    if (!aligned16(acts)) return LTR_ERR_ALIGN;
    if (grid < 1) return LTR_ERR_PARAM;
    a.dscores_in = dscores;
    a.partials = partials;
    a.acts_in = acts;
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_BWD_SAVED, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

int ltr_mlp_reduce_grads_sub(int net, int f, int h1, int h2, const float *partials, int grid, float *flat_grad, void *stream) {
    if (!partials || !flat_grad) return LTR_ERR_NULL;
    if (grid < 1) return LTR_ERR_PARAM;
    LTR_FOR_NET(net, {
        if (f < 1 || f > NET::F || h1 < 1 || h1 > NET::H1 || h2 < 1 || h2 > NET::H2) return LTR_ERR_SHAPE;
        hipLaunchKernelGGL(reduce_grads_kernel<NET>, dim3((NET::NPARAM * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, partials,
                           grid, flat_grad, f, h1, h2);
    })
    return status();
}

int ltr_mlp_reduce_grads(int net, const float *partials, int grid, float *flat_grad, void *stream) {
    int32_t info[8];
    if (int rc = ltr_net_info(net, info)) return rc;
    return ltr_mlp_reduce_grads_sub(net, info[0], info[1], info[2], partials, grid, flat_grad, stream);
}

int ltr_triple_fold(const float *W1, const float *b1, const float *W2, const float *b2, const float *w3, int F, int copies, float *W1e,
                    float *b1e, float *w3e, void *stream) {
    if (!W1 || !b1 || !W2 || !b2 || !w3 || !W1e || !b1e || !w3e) return LTR_ERR_NULL;
    if (F < 1 || F > 4096) return LTR_ERR_SHAPE;
    if (copies != 1 && copies != 2) return LTR_ERR_PARAM;
    const int n = kTrH2 * (F + 1);
    hipLaunchKernelGGL(triple_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1, b1, W2, b2, w3, F, copies, W1e, b1e,
                       w3e);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

int ltr_triple_unfold_grads(const float *g2, int F, int copies, const float *W1, const float *b1, const float *W2, float *flat,
                            void *stream) {
    if (!g2 || !W1 || !b1 || !W2 || !flat) return LTR_ERR_NULL;
    if (F < 1 || F > 4096) return LTR_ERR_SHAPE;
    if (copies != 1 && copies != 2) return LTR_ERR_PARAM;
    const int n = kTrH1 * F + kTrH1 + kTrH2 * kTrH1 + kTrH2 + kTrH2 + 1;
    hipLaunchKernelGGL(triple_unfold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g2, F, copies, W1, b1, W2, flat);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

int ltr_fused_step(int net, int loss_kind, const float *X, const float *labels, int B, int S, const float *packed,
                   int dropout, uint64_t seed, const uint8_t *keep1, const uint8_t *keep2, float alpha, float eps,
                   float pad, int apply_sigmoid, float grad_scale, float *slate_loss, float *partials, int grid,
                   void *stream) {
    PipeArgs a;
    if (B < 0 || (S != 32 && S != 64 && S != 128)) return LTR_ERR_SHAPE;
    if (int rc = fill_common(a, net, X, (int64_t)B * S, packed, dropout, seed, keep1, keep2)) return rc;
    if (!labels || !slate_loss || !partials) return LTR_ERR_NULL;
    if (loss_kind != LTR_LOSS_APPROXNDCG && loss_kind != LTR_LOSS_LISTNET) return LTR_ERR_PARAM;
    if (grid < 1) return LTR_ERR_PARAM;
    a.labels = labels;
    a.B = B;
    a.S = S;
    a.slate_loss = slate_loss;
    a.partials = partials;
    a.loss_kind = loss_kind;
    a.alpha = alpha;
    a.eps = eps;
    a.pad = pad;
    a.gscale = grad_scale;
    a.apply_sigmoid = apply_sigmoid;
    if (a.dropout && a.drop_thr16 && !keep1) return LTR_ERR_PARAM;     /* p != 0.5: the three-launch path */
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_FUSED, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

int ltr_fused_step_lambda(int net, const float *X, const float *labels, int B, int S, const float *packed, int dropout,
                          uint64_t seed, const uint8_t *keep1, const uint8_t *keep2, int scheme, int k, float sigma,
                          float mu, float eps, float pad, int log_base, float grad_scale, float *slate_loss,
                          float *slate_count, float *partials, int grid, void *stream) {
    PipeArgs a;
    if (B < 0 || (S != 32 && S != 64 && S != 128)) return LTR_ERR_SHAPE;
    if (int rc = fill_common(a, net, X, (int64_t)B * S, packed, dropout, seed, keep1, keep2)) return rc;
    if (!labels || !slate_loss || !partials) return LTR_ERR_NULL;
    if (scheme < 0 || scheme > 7 || (log_base != LTR_LOG_BINARY && log_base != LTR_LOG_NATURAL) || !(eps > 0.f) || grid < 1)
        return LTR_ERR_PARAM;
    a.labels = labels;
    a.B = B;
    a.S = S;
    a.slate_loss = slate_loss;
    a.slate_count = slate_count;
    a.partials = partials;
    a.loss_kind = 2;
    a.pad = pad;
    a.gscale = grad_scale;
    a.lp.scheme = scheme;
    a.lp.k = k;
    a.lp.sigma = sigma;
    a.lp.mu = mu;
    a.lp.eps = eps;
    a.lp.log_scale = log_base == LTR_LOG_BINARY ? (float)(1.0 / 0.693147180559945309417) : 1.f;
    a.lp.log_floor = log_base == LTR_LOG_BINARY ? log2f(eps) : logf(eps);
    if (a.dropout && a.drop_thr16 && !keep1) return LTR_ERR_PARAM;     /* p != 0.5: the three-launch path */
    LTR_FOR_NET(net, return pipeline_dispatch<NET>(MODE_FUSED, a, grid, (hipStream_t)stream))
    return LTR_ERR_PARAM;
}

}  // extern "C"
