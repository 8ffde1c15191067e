// ltr_bf16_split.h -- building blocks of the split-precision (bf16 x 3) GEMMs on v_mfma_f32_16x16x32_bf16 (gfx950),
// used by the split-precision pipeline (ltr_pipeline_bf16x3.h)
// weight-gradient GEMMs (LTR_DW_BF16 in ltr_scorer.hip):
//     a = a1 + a2 + a3 (bf16 pieces, round-to-nearest residual split);  a b ~ the six piece products down to 2^-24.
// Included inside ltr_scorer.hip's anonymous namespace (uses f32x4, kWaves, DwSet / dw_row / dw_col from there).
#pragma once

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// (lo, hi) -> one dword of two bf16, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// Two fp32 values -> NP packed bf16 piece pairs (piece 0 the leading 8 bits, each next piece the rounded residual).
template <int NP>
__device__ __forceinline__ void split2(float v0, float v1, unsigned (&out)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        out[p] = pack_bf16(v0, v1);
        if (p + 1 < NP) {
            v0 -= bf16_lo(out[p]);
            v1 -= bf16_hi(out[p]);
        }
    }
}

// Accumulator tiles (feature 16 To + 4 q + r of document lane&15) -> B-operand k tiles of the next GEMM.
// k tile T takes tiles 2T and 2T+1: slot j of lane group q is feature 32 T + 16 (j >> 2) + 4 q + (j & 3) -- the weight
// packing uses the same permutation, so no data moves between lanes.  Tiles >= NTILE are zero.
template <int NTILE, int KT, int NP, int KA>
__device__ __forceinline__ void tiles_to_operand(const f32x4 *t, u32x4 (&out)[KA][NP]) {
#pragma unroll
    for (int T = 0; T < KT; ++T) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int To = 2 * T + half;
            unsigned a[NP], b[NP];
            if (To < NTILE) {
                split2<NP>(t[To][0], t[To][1], a);
                split2<NP>(t[To][2], t[To][3], b);
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) a[p] = b[p] = 0u;
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                out[T][p][2 * half] = a[p];
                out[T][p][2 * half + 1] = b[p];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- dW GEMMs (K = documents)
// A [doc][feature] bf16 image: `LD` features per row.  Fragment of feature tile t for a 32-document k step starting
// at row r0: lane (i = lane & 15, g = lane >> 4) gets feature 16 t + i of documents r0 + 4 g + {0..3} and
// r0 + 16 + 4 g + {0..3} (the document <-> k-slot assignment is free as long as both operands use the same one).
// ds_read_b64_tr_b16: lane 4a+p of a 16-lane group supplies the address of row a, columns 4p..4p+3.
template <int LD>
__device__ __forceinline__ u32x4 tr_frag(const unsigned short *img, int r0, int t, int lane) {
    typedef __attribute__((address_space(3))) s16x4 *lp4;
    const int i = lane & 15, g = lane >> 4;
    const unsigned short *p = img + (r0 + 4 * g + (i >> 2)) * LD + 16 * t + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(p + 16 * LD));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// dW tiles of wave W += A^T B over ONE k step of 32 documents (three pieces per operand, six products).
//   a_img / b_img: piece p of the A / B operand is the [rows][LD] image at a_img + p * a_plane (b_img + p * b_plane)
// The wave's tile set is walked in its compile-time order (row bands, column-major inside a band): the A fragments
// of the current band (BH rows) and the B fragments of the current column stay in registers.
template <int W, int TW, int NR, int NC, int BH, int LD>
__device__ __forceinline__ void dw_kstep_bf16_w(f32x4 (&acc)[TW], const unsigned short *a_img, int a_plane, int a_r0,
                                                const unsigned short *b_img, int b_plane, int b_r0, int lane) {
    u32x4 af[BH][3], bf[3];
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int g = W * TW + j;
        if (g < NR * NC) {
            const int To = dw_row<NR, NC, BH>(g), Ti = dw_col<NR, NC, BH>(g);
            const int band = To / BH;
            const bool new_band = j == 0 || dw_row<NR, NC, BH>(g - 1) / BH != band;
            const bool new_col = j == 0 || new_band || dw_col<NR, NC, BH>(g - 1) != Ti;
            if (new_band) {
#pragma unroll
                for (int r = 0; r < BH; ++r)
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[r][p] = tr_frag<LD>(a_img + p * a_plane, a_r0, band * BH + r, lane);
            }
            if (new_col) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[p] = tr_frag<LD>(b_img + p * b_plane, b_r0, Ti, lane);
            }
            const int r = To - band * BH;
            acc[j] = mfma_bf16(af[r][2], bf[0], acc[j]);
            acc[j] = mfma_bf16(af[r][0], bf[2], acc[j]);
            acc[j] = mfma_bf16(af[r][1], bf[1], acc[j]);
            acc[j] = mfma_bf16(af[r][1], bf[0], acc[j]);
            acc[j] = mfma_bf16(af[r][0], bf[1], acc[j]);
            acc[j] = mfma_bf16(af[r][0], bf[0], acc[j]);
            // one scheduling region per tile: left alone, hipcc hoists every fragment read of the k step to its top and
            // spills the dW accumulators to make room
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int TW, int NR, int NC, int BH, int LD>
__device__ __forceinline__ void dw_kstep_bf16(int w, f32x4 (&acc)[TW], const unsigned short *a_img, int a_plane, int a_r0,
                                              const unsigned short *b_img, int b_plane, int b_r0, int lane) {
#if defined(LTR_EXP) && (LTR_EXP & 4)     /* timing experiment only (results wrong): one body for every wave */
    dw_kstep_bf16_w<0, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane);
    return;
#endif
    switch (w) {   // wave-uniform: one specialised, branch-free body per wave
        case 0: dw_kstep_bf16_w<0, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 1: dw_kstep_bf16_w<1, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 2: dw_kstep_bf16_w<2, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 3: dw_kstep_bf16_w<3, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 4: dw_kstep_bf16_w<4, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 5: dw_kstep_bf16_w<5, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        case 6: dw_kstep_bf16_w<6, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
        default: dw_kstep_bf16_w<7, TW, NR, NC, BH, LD>(acc, a_img, a_plane, a_r0, b_img, b_plane, b_r0, lane); break;
    }
}

// k-tile operand (3 pieces, 8 features per lane) -> row `row` of the three piece images (img + p * plane): the low
// 8 bytes of a k tile are tile 2T (features 16(2T) + 4q .. +3), the high 8 bytes tile 2T+1.  NT16 = tiles to write.
template <int KT, int NT16, int LD, int KA>
__device__ __forceinline__ void operand_to_images(const u32x4 (&op)[KA][3], unsigned short *img, int plane, int row, int q) {
#pragma unroll
    for (int T = 0; T < KT; ++T)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int To = 2 * T + half;
            if (To < NT16) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<u32x2 *>(img + p * plane + row * LD + 16 * To + 4 * q) =
                        u32x2{op[T][p][2 * half], op[T][p][2 * half + 1]};
            }
        }
}

