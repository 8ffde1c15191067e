// ltr_fcw.h -- the fused slate pipeline of the TWO-LAYER scorer (136 -> 64 -> 1: the DoubleLayerNet variant of
// architeture/doubleLayer.py:38-51 that BASELINE.json configs[0] / [1] name), feature-partitioned.  Two arithmetic versions of
// the same layout, chosen at compile time: exact fp32 (v_mfma_f32_16x16x4_f32; the default library) and f16 x 2 (the variant
// library, -DLTR_F16X2=1).  Included by ltr_scorer.hip inside its anonymous namespace.
//
// The generic pipeline (slate_pipeline_kernel) gives every wave 16 documents and ALL hidden units: every wave then reads every
// weight, and dz has to be staged through LDS for the document contraction of dW.  Here the HIDDEN UNITS are partitioned instead
// ("weights-stationary"): a workgroup is 4 waves, wave w owns hidden units 16w .. 16w+15 for all 128 documents of the tile.
//   * its W1 fragments (hi / lo f16, 5 k-steps: 40 VGPRs) are loaded ONCE per kernel and stay in registers;
//   * fc1 runs NON-transposed, z1[doc][n] = sum_f x[doc][f] W1[n][f]: A = rows of the X image pair in LDS (ds_read_b128, conflict
//     free), B = the resident fragments; the accumulator tile (row = document, col = hidden unit) of two document tiles IS the A
//     operand of the weight gradient dW1[n][f] = sum_doc dz1[doc][n] x[doc][f] (k = documents, same k-slot order as the
//     transposing LDS read that supplies x) -- no staging, no barrier between the backward through fc3 and dW1;
//   * the scores need one exchange (each wave holds a 16-unit partial of w3 . h1): 2 KB through LDS.
// 256 threads and 80 KB of LDS per workgroup -> TWO workgroups per CU: the listwise loss of one slate (barrier- and latency-bound,
// matrix pipe idle) runs under the other workgroup's load / convert / MFMA phases.
//
// Exact fp32 version: X lives in LDS as fp32 rows of F + 4 floats (the ones feature and three zeros behind the F inputs; 140
// floats: rows 12 banks apart, so both access patterns below are conflict-free).  For v_mfma_f32_16x16x4_f32 a lane holds ONE
// value per operand, k-slot = lane >> 4, and the assignment of k-slots to features is free as long as both operands agree:
//   * fc1 (z1[doc][n], A = x, B = W1): MFMA i of feature group S takes feature 16 S + 4 (lane >> 4) + i, so ONE ds_read_b128 of
//     a document's row feeds four MFMAs and the resident B fragments are exactly the [64 lanes][4] tiles ltr_mlp_pack writes for
//     the generic kernel (N::W1F_OFF): 36 VGPRs per wave, loaded once per kernel;
//   * dW1 (A = dz1, B = x): the accumulator tile of fc1 -- lane (n, q) holds documents 4 q + r -- IS the A operand of MFMA r with
//     k-slot q = document 4 q + r; B is one ds_read_b32 per MFMA, 16 consecutive floats of four rows: conflict-free.
// 576 MFMAs per wave and tile (18.4 k cycles on its SIMD); no staging of dz through LDS, no weight stream.
#pragma once

// the sigmoid's reciprocal: v_rcp_f32 (1 ulp) by default.  The backward's h (1 - h) loses RELATIVE accuracy when h is within 1e-4 of
// 1, but what the parity metric sees is the absolute error of the gradient contribution, g x 6e-8 -- the size of any other fp32
// rounding of the step; the correctly rounded division (__frcp_rn: ~10 instructions) bought nothing measurable and cost ~7 % of the
// folded TripleLayerNet tile (profiles/r04_variant_ab.json)
#ifndef FCW_SIGMOID_RCP
#define FCW_SIGMOID_RCP ltr_rcp
#endif
#ifdef LTR_STAMPS
#define FCW_STAMP(k)                                                                                          \
    if (a.stamps && lane == 0 && (st - (int)blockIdx.x) / (int)gridDim.x == a.stamp_tile)                      \
        a.stamps[((size_t)blockIdx.x * 8 + w) * 16 + (k)] = __builtin_readcyclecounter();
#else
#define FCW_STAMP(k)
#endif

#ifndef FCW_X_PREFETCH
#define FCW_X_PREFETCH 0           // before the dW1 GEMM of a tile: 1 = the next tile's X rows -> registers (68 VGPRs: spills 356 B/lane,
                                   // 19.0 -> 12.1 M slates/s), 2 = -> L2 only (17.7 M); 0 = none: both measured SLOWER (profiles/r04_variant_ab.json)
#endif
#ifndef FCW_W1_RELOAD
#define FCW_W1_RELOAD 1            // exact fp32: the W1 fragments are re-read from L2 per tile (9 KB per wave) instead of pinning 36 VGPRs through
#endif                             // the loss and dW1 phases
constexpr int kFcwWaves = 4;
constexpr int kFcwThreads = kFcwWaves * 64;

template <class N>
constexpr size_t fcw_lds() {
#if LTR_F16X2
    // X image pair [128][LD] halfs x 2 (+ 32 B so that the last k-step's over-read of the last row stays inside finite f16s),
    // 8 loss arrays [128], score partials [4][128], slate-group scratch [256 + 4*32], 16 exchange floats
    return (size_t)2 * kTileDocs * N::LD * 2 + 32 + sizeof(float) * (8 * kTileDocs + kFcwWaves * kTileDocs + kFcwThreads + 4 * 32 + 16);
#else
    // X [128][F + 4] fp32, 8 loss arrays [128], score partials [4][128], slate-group scratch [256 + 4*32]
    return sizeof(float) * ((size_t)kTileDocs * (N::F + 4) + 8 * kTileDocs + kFcwWaves * kTileDocs + kFcwThreads + 4 * 32);
#endif
}

// ST: the slate length when it is a compile-time constant (approxNDCG: one kernel per slate length, so that each carries ONE
// copy of the loss with fixed geometry -- three copies in one kernel cost it 90 B/lane of scratch); 0 = a.S at run time.
template <class N, int LOSS, int ST>
__global__ void __launch_bounds__(kFcwThreads, 2) fcw_fused_kernel(const PipeArgs a) {
    static_assert(N::TWO && N::H1 == 16 * kFcwWaves, "one hidden tile per wave");
    static_assert(ST == 0 || ST == 32 || ST == 64 || ST == 128, "slate length");
    constexpr int XT = N::XT;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
#if LTR_F16X2
    constexpr int LDH = N::LD;                       // halfs per image row (72 dwords at LD = 144: conflict-free both ways)
    constexpr int KP = N::KP1;                       // 32-wide k-steps over the features (+ the ones feature)
    uint16_t *Xhi = reinterpret_cast<uint16_t *>(smem_f), *Xlo = Xhi + kTileDocs * LDH;
    float *sc = reinterpret_cast<float *>(Xlo + kTileDocs * LDH + 16);
#else
    constexpr int LDX = N::F + 4;                    // floats per X row: F inputs, the ones feature, three zeros
    constexpr int LASTQ = (LDX - 16 * (XT - 1)) / 4; // 4-float groups of the last 16-feature tile that lie inside a row
    static_assert(LDX % 8 == 4, "rows an odd number of 16-byte bank groups apart: conflict-free ds_read_b128 row fragments");
    static_assert(LDX > 16 * (XT - 1) && LDX <= 16 * XT && LDX % 4 == 0, "the last feature tile starts inside the row");
    float *Xs = smem_f;
    float *sc = Xs + kTileDocs * LDX;
#endif
    float *yl = sc + kTileDocs, *gn = yl + kTileDocs, *gg = gn + kTileDocs, *dsc = gg + kTileDocs, *uu = dsc + kTileDocs,
          *mk = uu + kTileDocs, *xt = mk + kTileDocs;
    float *part = xt + kTileDocs;                    // [4][128] per-wave score partials
    float *scratch = part + kFcwWaves * kTileDocs;   // [256 + 4*32]
#if LTR_F16X2
    float *exch = scratch + kFcwThreads + 4 * 32;    // [16]
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, d = lane & 15;
    const int n_mine = 16 * w + d;                   // this lane's hidden unit (accumulator column)
    // document split (N::DS == 2: the hidden rows are two copies of 32 units, NetT): this wave's units see the document tiles
    // T0 .. T0 + NTW - 1 of a super-tile only; DS == 1: all eight
    constexpr int NTW = 8 / N::DS, WPC = kFcwWaves / N::DS;         // document tiles per wave; waves per copy
    const int T0 = N::DS == 1 ? 0 : (w / WPC) * NTW;

    // ---- once per kernel: resident W1 fragments, w3, pads of the X rows, loss scratch
    const float w3n = a.packed[N::W3_OFF + n_mine];
    const float b3 = a.packed[N::W3_OFF + N::NT2 * 16];
#if LTR_F16X2
    h16x8 wh[KP], wl[KP];
#if FCW_W1_RELOAD
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, N::PACKED * 4, 0x00020000);
#else
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.packed + N::W1B_OFF) + (size_t)w * KP * 2 * 64 + lane;
#pragma unroll
        for (int P = 0; P < KP; ++P) {
            wh[P] = __builtin_bit_cast(h16x8, src[(2 * P) * 64]);
            wl[P] = __builtin_bit_cast(h16x8, src[(2 * P + 1) * 64]);
        }
    }
#endif
    const float inv_w1 = a.packed[N::W3_OFF + N::NT2 * 16 + 4];
    float w3max = 0.f;
    for (int j = lane; j < N::H2; j += 64) w3max = fmaxf(w3max, fabsf(a.packed[N::W3_OFF + j]));
    w3max = wave_allmax(w3max);
    for (int e = tid; e < kTileDocs * (LDH - N::F); e += kFcwThreads) {
        const int r = e / (LDH - N::F), c = N::F + e % (LDH - N::F);
        Xhi[r * LDH + c] = 0;
        Xlo[r * LDH + c] = 0;
    }
    if (tid < 16) Xlo[kTileDocs * LDH + tid] = 0;
    int exx = 1, exd = -100, E = -400;
#else
    f32x4 wf[XT];                                    // lane (n, q): W1aug[16 w + n][16 S + 4 q .. + 3]
#if FCW_W1_RELOAD
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, N::PACKED * 4, 0x00020000);
#else
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.packed + N::W1F_OFF) + (size_t)w * XT * 64 + lane;
#pragma unroll
        for (int S = 0; S < XT; ++S) wf[S] = src[S * 64];
    }
#endif
    for (int e = tid; e < kTileDocs * 4; e += kFcwThreads) Xs[(e >> 2) * LDX + N::F + (e & 3)] = (e & 3) ? 0.f : 1.f;
#endif
    for (int j = tid; j < kFcwThreads + 4 * 32; j += kFcwThreads) scratch[j] = 0.f;
    if (LOSS == 0) {
        __syncthreads();
        ltr_fill_inv_discount(scratch + 128, kTileDocs, tid, kFcwThreads);      // [128, 256): 1 / log2(2 + rank); [0, 64): wave partials
    }
    f32x4 accW[XT];                                  // dW1 tiles (rows = my 16 hidden units, cols = 16 features each)
#pragma unroll
    for (int t = 0; t < XT; ++t) accW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dw3 = 0.f, db3 = 0.f;

    constexpr int V4_PER_ROW = N::F / 4, ROWS = kTileDocs / kFcwWaves, V4 = ROWS * V4_PER_ROW;   // this wave converts 32 rows
    constexpr int NV = (V4 + 63) / 64;
    // Which 16-byte piece of the wave's 32 rows a lane takes in load m.  XPAIR = false: piece lane + 64 m of the contiguous block (1 KiB
    // per load; row = e / 34, column = e % 34).  XPAIR = true: two rows x 32 columns per load for m < NV - 1 (row 2 m + lane / 32, column
    // lane % 32) and the two left-over columns of all 32 rows in the last load (row lane / 2, column 32 + lane % 2): every global and
    // LDS address is one lane constant plus an immediate, and the LDS stores of a load land in two rows instead of straddling three.
    // Measured (profiles/r04_variant_ab.json, r4ae): 136-64-1 exact fp32 +4.5 %, its f16 x 2 variant +2.3 %, folded TripleLayerNet
    // exact +0.7 %, its f16 x 2 variant -4 % (the one kernel fast enough to feel the 32-byte runs of the tail load) -- hence:
    constexpr bool XPAIR = !(LTR_F16X2 && N::DS == 2);
    constexpr int XMAIN = 32, XTAIL = V4_PER_ROW - XMAIN;
    static_assert(V4 % 64 == 0, "the wave's row block is a whole number of 1 KiB wave loads");
    static_assert(!XPAIR || (XTAIL > 0 && ROWS * XTAIL == 64 && NV == ROWS / 2 + 1), "16 two-row loads and one tail load per wave and tile");
    const int xrow_main = lane >> 5, xcol_main = lane & 31, xrow_tail = lane / XTAIL, xcol_tail = XMAIN + lane % XTAIL;
    // (row, 4-float column) of this lane's piece of load m
    auto x_row = [&](int m) { return XPAIR ? (m < NV - 1 ? 2 * m + xrow_main : xrow_tail) : (lane + 64 * m) / V4_PER_ROW; };
    auto x_col = [&](int m) { return XPAIR ? (m < NV - 1 ? xcol_main : xcol_tail) : (lane + 64 * m) % V4_PER_ROW; };
    f32x4 xn[NV];
    auto load_x = [&](int tile) {
        // one buffer descriptor per wave and tile, sized to the rows that exist: the hardware bounds check returns zeros for
        // rows past the end of the batch (no per-element compare, no 64-bit address registers)
        const long long row0 = (long long)tile * kTileDocs + ROWS * w;
        long long rows_here = a.n_docs - row0;
        rows_here = rows_here < 0 ? 0 : (rows_here > ROWS ? ROWS : rows_here);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.X + (rows_here ? row0 : 0) * N::F), 0,
                                                                             (int)rows_here * N::F * 4, 0x00020000);
        if (XPAIR) {
            const int lo = (xrow_main * N::F + 4 * xcol_main) * 4;
#pragma unroll
            for (int m = 0; m < NV - 1; ++m)
                xn[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, lo + m * (2 * N::F * 4), 0, 2 /* nt */));
            xn[NV - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (xrow_tail * N::F + 4 * xcol_tail) * 4, 0, 2));
        } else {
#pragma unroll
            for (int m = 0; m < NV; ++m)
                xn[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16 + m * 1024, 0, 2 /* nt */));
        }
    };
#if FCW_X_PREFETCH == 1
    if ((int)blockIdx.x < a.n_super) load_x(blockIdx.x);
#endif
    for (int st = blockIdx.x; st < a.n_super; st += gridDim.x) {
        const long long doc_base = (long long)st * kTileDocs;
        FCW_STAMP(0)
        // ---- X: this wave's 32 rows are in flight (or landed) in xn: fetched ahead of time, before the previous tile's dW1 GEMM
        //      (FCW_X_PREFETCH), so that the HBM latency hides under its MFMAs instead of opening every tile
#if FCW_X_PREFETCH != 1
        load_x(st);
#endif
#if LTR_F16X2
        float xm = 0.f;
#pragma unroll
        for (int m = 0; m < NV; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) xm = fmaxf(xm, fabsf(xn[m][r]));
        xm = wave_allmax(xm);
#endif
#if FCW_W1_RELOAD
        {   // this wave's W1 fragments, L2 -> registers, behind the X loads (consumed in fc1, two barriers from here); the byte offset
            // is laundered per tile so that the loads are not hoisted out of the persistent loop and pinned for the whole kernel
#if LTR_F16X2
            int wo = N::W1B_OFF * 4 + w * KP * 2 * 1024;
            asm volatile("" : "+s"(wo));
#pragma unroll
            for (int P = 0; P < KP; ++P) {
                wh[P] = __builtin_bit_cast(h16x8, load_frag(wrs, lane * 16, wo + (2 * P) * 1024));
                wl[P] = __builtin_bit_cast(h16x8, load_frag(wrs, lane * 16, wo + (2 * P + 1) * 1024));
            }
#else
            int wo = (N::W1F_OFF + w * XT * 256) * 4;
            asm volatile("" : "+s"(wo));
#pragma unroll
            for (int S = 0; S < XT; ++S) wf[S] = load_frag(wrs, lane * 16, wo + S * 1024);
#endif
        }
#endif
        FCW_STAMP(1)
        __syncthreads();                              // A: every wave is done with the previous tile's images, scores, gradients
#if LTR_F16X2
        if (lane == 0) exch[w] = xm;
#endif
        if (tid < kTileDocs) {
            const long long doc = doc_base + tid;
            const float y = doc < a.n_docs ? a.labels[doc] : a.pad;
            if (LOSS != 1) stage_label(y, a.pad, yl[tid], gn[tid]);
            else yl[tid] = doc < a.n_docs ? y : 0.f;
        }
#if LTR_F16X2
        __syncthreads();                              // B: the four row-block maxima are out
        FCW_STAMP(2)
        {
            float m4 = 1.f;                           // the ones feature
#pragma unroll
            for (int i = 0; i < kFcwWaves; ++i) m4 = fmaxf(m4, exch[i]);
            exx = grow_exp(exx, m4);
            const float sx = ldexpf(1.f, 14 - exx);
#pragma unroll
            for (int m = 0; m < NV; ++m) {
                u32x2 hi, lo;
                split4(xn[m], sx, hi, lo);
                const int off = (ROWS * w + x_row(m)) * LDH + 4 * x_col(m);
                *reinterpret_cast<u32x2 *>(Xhi + off) = hi;
                *reinterpret_cast<u32x2 *>(Xlo + off) = lo;
            }
            if (lane < ROWS) {
                Xhi[(ROWS * w + lane) * LDH + N::F] = __builtin_bit_cast(uint16_t, (_Float16)sx);
                Xlo[(ROWS * w + lane) * LDH + N::F] = 0;
            }
        }
#else
        FCW_STAMP(2)
#pragma unroll
        for (int m = 0; m < NV; ++m)                  // rows of F floats -> rows of LDX floats (the pads were written once)
            *reinterpret_cast<f32x4 *>(Xs + (ROWS * w + x_row(m)) * LDX + 4 * x_col(m)) = xn[m];
#endif
        __syncthreads();                              // C: the tile is complete in LDS
        FCW_STAMP(3)
        // ---- fc1: z1[doc][n] for my 16 hidden units, all 8 document tiles; A = image rows, B = resident fragments
        f32x4 h1[NTW];
#if LTR_F16X2
        {
            const uint16_t *rh = Xhi + (16 * T0 + d) * LDH + 8 * q, *rl = Xlo + (16 * T0 + d) * LDH + 8 * q;
            u32x4 fh[2][KP], fl[2][KP];              // A fragments of two document tiles: the reads of T+1 fly under the MFMAs of T
#pragma unroll
            for (int P = 0; P < KP; ++P) {
                fh[0][P] = *reinterpret_cast<const u32x4 *>(rh + 32 * P);
                fl[0][P] = *reinterpret_cast<const u32x4 *>(rl + 32 * P);
            }
#pragma unroll
            for (int T = 0; T < NTW; ++T) {
                if (T + 1 < NTW) {
#pragma unroll
                    for (int P = 0; P < KP; ++P) {
                        fh[(T + 1) & 1][P] = *reinterpret_cast<const u32x4 *>(rh + 16 * (T + 1) * LDH + 32 * P);
                        fl[(T + 1) & 1][P] = *reinterpret_cast<const u32x4 *>(rl + 16 * (T + 1) * LDH + 32 * P);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int P = 0; P < KP; ++P) {
                    const h16x8 ahi = __builtin_bit_cast(h16x8, fh[T & 1][P]), alo = __builtin_bit_cast(h16x8, fl[T & 1][P]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, wh[P], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, wl[P], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, wh[P], acc, 0, 0, 0);
                    if (LTR_LOLO) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, wl[P], acc, 0, 0, 0);
                }
                h1[T] = acc;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        {
            // lane (doc, q) reads floats 16 S + 4 q .. + 3 of its document's row; in the last feature group the quads past the row
            // (their W1 fragments are zero) are taken from the row's last quad instead (finite, and the lane's own row)
            const float *ra = Xs + (16 * T0 + d) * LDX + 4 * q;
            const int last = 16 * (XT - 1) + (q < LASTQ ? 0 : 4 * (LASTQ - 1 - q));
            f32x4 fa[2][XT];                          // A fragments of two document tiles: the reads of T+1 fly under the MFMAs of T
#pragma unroll
            for (int S = 0; S < XT; ++S) fa[0][S] = *reinterpret_cast<const f32x4 *>(ra + (S + 1 < XT ? 16 * S : last));
#pragma unroll
            for (int T = 0; T < NTW; ++T) {
                if (T + 1 < NTW) {
#pragma unroll
                    for (int S = 0; S < XT; ++S)
                        fa[(T + 1) & 1][S] = *reinterpret_cast<const f32x4 *>(ra + 16 * (T + 1) * LDX + (S + 1 < XT ? 16 * S : last));
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains: 32-cycle issue vs 40-cycle dependent latency
#pragma unroll
                for (int S = 0; S < XT; ++S) {
                    acc0 = mfma4(fa[T & 1][S][0], wf[S][0], acc0);
                    acc1 = mfma4(fa[T & 1][S][1], wf[S][1], acc1);
                    acc0 = mfma4(fa[T & 1][S][2], wf[S][2], acc0);
                    acc1 = mfma4(fa[T & 1][S][3], wf[S][3], acc1);
                }
                h1[T] = acc0 + acc1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        FCW_STAMP(4)
        // ---- activation (+ dropout), score partial w3 . h1 over my 16 units
#if LTR_F16X2
        const float un = ldexpf(inv_w1, exx - 14);
#else
        const float un = 1.f;
#endif
        const bool drop = (N::A1 == ACT_RELU_DROP) && a.dropout;
        long long left64 = a.n_docs - doc_base;
        const int docs_left = (int)(left64 > kTileDocs ? kTileDocs : left64);        // documents of this tile that exist (uniform)
#pragma unroll
        for (int T = 0; T < NTW; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = h1[T][r] * un;
                if (N::A1 == ACT_RELU_DROP) v = fmaxf(v, 0.f);
                else if (N::A1 == ACT_SIGMOID) v = FCW_SIGMOID_RCP(1.f + __expf(-v));
                h1[T][r] = v;
            }
        if (drop) {      // (uniform) training-mode dropout: its own loop, so that the hash arithmetic does not sit in the common path
#pragma unroll
            for (int T = 0; T < NTW; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dl = 16 * (T0 + T) + 4 * q + r;
                    const long long doc = doc_base + dl;
                    bool keep;
                    if (a.keep1) keep = dl < docs_left && a.keep1[doc * N::H1 + n_mine] != 0;
                    else keep = (keep_word(a.seed, 0, doc, n_mine >> 5) >> (n_mine & 31)) & 1u;
                    h1[T][r] = keep ? h1[T][r] * a.drop_scale : 0.f;
                }
        }
        // score partial w3 . h1 over my 16 hidden units: sum over the 16 lanes of a DPP row, for each of my 32 documents
#pragma unroll
        for (int T = 0; T < NTW; ++T) {
            const f32x4 sv = row_sum4_to_lane15(h1[T] * w3n);
            if (d == 15) *reinterpret_cast<f32x4 *>(part + w * kTileDocs + 16 * (T0 + T) + 4 * q) = sv;
        }
        FCW_STAMP(5)
        // score of document j of the tile: the partials of the waves whose units saw it (all four; DS == 2: the two of its half)
        auto doc_score = [&](int j) {
            const float *pp = part + j;
            if (N::DS == 1) return ((pp[0] + pp[kTileDocs]) + (pp[2 * kTileDocs] + pp[3 * kTileDocs])) + b3;
            const float *ph = pp + (j / (kTileDocs / N::DS)) * WPC * kTileDocs;
            return (ph[0] + ph[kTileDocs]) + b3;
        };
        __syncthreads();                              // D: the four partials of every document are out
        if (LOSS != 0) {                              // (approxNDCG sums the partials inside its wave-private prologue: no second barrier)
            if (tid < kTileDocs) sc[tid] = doc_score(tid);
            __syncthreads();
        }
        FCW_STAMP(6)
        // ---- listwise loss on the LDS-resident scores -> dsc (2 threads per document row)
        {
            const int S_ = ST ? ST : a.S;
            const int group = 2 * S_;
            const int gid = tid / group;
            const SlateGroup g = make_group(S_, group, scratch + gid * (group + 32), tid);
            const int so = gid * S_;
            const long long slate = (long long)st * (kTileDocs / S_) + gid;
            float loss;
            if constexpr (LOSS == 0) {
                // one slate per 2 S threads; every wave of a slate sums the score partials of ALL its documents itself
                auto stamp_fn = [&](int k) { FCW_STAMP(k) };
                auto run = [&](auto s_tag, auto nw_tag, auto stamper) {
                    constexpr int SS = decltype(s_tag)::value, NWS = decltype(nw_tag)::value;
                    const int gi = tid / (64 * NWS), so2 = gi * SS;
                    auto score = [&](int j) { return doc_score(so2 + j); };
                    return approx_ndcg_fused<SS, NWS, true>(tid - gi * 64 * NWS, score, sc + so2, yl + so2, gn + so2, gg + so2, uu + so2,
                                                            xt + so2, mk + so2, scratch + gi * 16, scratch + 128, a.alpha, a.eps, a.gscale,
                                                            [&](int i, float v) { dsc[so2 + i] = v; }, stamper);      // (documents past the batch carry the pad label: gradient 0)
                };
                loss = run(std::integral_constant<int, ST ? ST : 128>(), std::integral_constant<int, (ST ? ST : 128) / 32>(), stamp_fn);
            } else if constexpr (LOSS == 1) {
                loss = listnet_slate(g, yl + so, sc + so, a.apply_sigmoid != 0, a.gscale, true, [&](int i, float v) { dsc[so + i] = slate < a.B ? v : 0.f; });
            } else {
                LambdaLds L;
                L.sc = sc + so; L.yl = yl + so; L.gn = gn + so; L.w1 = gg + so; L.invd = uu + so; L.delta = mk + so;
                L.rk = reinterpret_cast<int *>(xt + so);
                float count;
                loss = lambda_slate<-1>(g, L, a.lp, a.gscale, true, &count, [&](int i, float v) { dsc[so + i] = slate < a.B ? v : 0.f; });
                if (g.t == 0 && slate < a.B && a.slate_count) a.slate_count[slate] = count;
            }
            if (g.t == 0 && slate < a.B) a.slate_loss[slate] = loss;
        }
        __syncthreads();                              // E: d loss / d score of the whole tile
        FCW_STAMP(7)
        // ---- backward through fc3: dw3[n] += sum_doc ds h1, dz1 = ds w3 act'(h1); documents past the batch carry no gradient
        const float slope = drop ? a.drop_scale : 1.f;
#if LTR_F16X2
        float dmx = fmaxf(fabsf(dsc[lane]), fabsf(dsc[lane + 64]));
        dmx = wave_allmax(dmx);
#endif
        // (every loss stores a ZERO score gradient for documents past the end of the batch: no per-document test below)
        if (w == 0) db3 += wave_allsum(dsc[lane] + dsc[lane + 64]);
#if LTR_F16X2
        exd = grow_exp(exd, dmx * slope * w3max);
        if (exd + exx != E) {
            const float f = ldexpf(1.f, E - (exd + exx));
#pragma unroll
            for (int t = 0; t < XT; ++t) accW[t] *= f;
            E = exd + exx;
        }
        const float sd = ldexpf(1.f, 14 - exd);
#endif
#pragma unroll
        for (int T = 0; T < NTW; ++T) {
            const f32x4 ds4 = *reinterpret_cast<const f32x4 *>(dsc + 16 * (T0 + T) + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ds = ds4[r];
                dw3 = fmaf(ds, h1[T][r], dw3);
                h1[T][r] = apply_act_grad<N::A1>(ds * slope * w3n, h1[T][r]);        // now dz1
            }
        }
        FCW_STAMP(8)
#if FCW_X_PREFETCH == 1
        if (st + (int)gridDim.x < a.n_super) load_x(st + gridDim.x);      // the next tile's rows: they land under the dW1 MFMAs below
#elif FCW_X_PREFETCH == 2
        // pull the NEXT tile of X into L2 while this one is in its dW1 GEMM: one dword per 128-B line per lane (544 lines); the values
        // are only kept alive until the end of the iteration so that the loads are issued here and waited for there
        float pf = 0.f;
        {
            const long long nb = (long long)(st + gridDim.x) * kTileDocs * N::F;         // first float of the next tile
            const long long lim = a.n_docs * N::F;
            if (st + (int)gridDim.x < a.n_super) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const long long fl = nb + ((long long)tid + 256 * k) * 32;
                    if (tid + 256 * k < (kTileDocs * N::F + 31) / 32 && fl < lim) pf += a.X[fl];
                }
            }
        }
#endif
#if LTR_F16X2
        // ---- dW1[n][f] += sum_doc dz1[doc][n] x[doc][f]: A = two accumulator tiles of dz1 (split in registers), B = x k-major
        {
            const int lb = (16 * T0 + 4 * q + (d >> 2)) * LDH + 4 * (d & 3);
#pragma unroll
            for (int Kp = 0; Kp < NTW / 2; ++Kp) {
                h16x8 ahi, alo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = h1[2 * Kp + (j >> 2)][j & 3] * sd;
                    const _Float16 h = (_Float16)v;
                    ahi[j] = h;
                    alo[j] = (_Float16)(v - (float)h);
                }
#pragma unroll
                for (int Ti = 0; Ti < XT; ++Ti) {
                    const h16x8 bh = tr_frag_h<LDH>(Xhi + lb, 32 * Kp, Ti), bl = tr_frag_h<LDH>(Xlo + lb, 32 * Kp, Ti);
                    accW[Ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh, accW[Ti], 0, 0, 0);
                    accW[Ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl, accW[Ti], 0, 0, 0);
                    accW[Ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh, accW[Ti], 0, 0, 0);
                    if (LTR_LOLO) accW[Ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bl, accW[Ti], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        // ---- dW1[n][f] += sum_doc dz1[doc][n] x[doc][f]: A = register r of the dz1 accumulator tile (k-slot q = document 4 q + r),
        //      B = x[document 16 T + 4 q + r][feature of (tile Ti, column d)].  Which feature a column stands for is free (the store at
        //      the end of the kernel undoes it): column d of tile Ti < 8 is feature 64 (Ti >> 2) + 4 d + (Ti & 3), so ONE ds_read_b128
        //      of 4 consecutive floats feeds FOUR tiles (3 LDS instructions per step instead of 9); tile 8 keeps 128 + d, its columns
        //      past the row read the row's last float (a zero pad).
        {
            static_assert(XT == 9, "two groups of four feature tiles and the tail tile");
            const float *rb = Xs + (16 * T0 + 4 * q) * LDX + 4 * d;
            const float *rt = Xs + (16 * T0 + 4 * q) * LDX + (128 + d < LDX ? 128 + d : LDX - 1);
            // software pipeline over the 32 (document tile, register) steps: the B values of step s + 1 are in flight under the nine
            // MFMAs (288 cycles) of step s -- left to itself the compiler reads each value right before its MFMA
            f32x4 bq[2][2];
            float bt[2];
            auto load_b = [&](int step, f32x4 (&dst)[2], float &dt) {
                const int ro = (16 * (step >> 2) + (step & 3)) * LDX;
                dst[0] = *reinterpret_cast<const f32x4 *>(rb + ro);
                dst[1] = *reinterpret_cast<const f32x4 *>(rb + ro + 64);
                dt = rt[ro];
            };
            load_b(0, bq[0], bt[0]);
#pragma unroll
            for (int step = 0; step < 4 * NTW; ++step) {
                if (step + 1 < 4 * NTW) load_b(step + 1, bq[(step + 1) & 1], bt[(step + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const float av = h1[step >> 2][step & 3];
#pragma unroll
                for (int Ti = 0; Ti < 8; ++Ti) accW[Ti] = mfma4(av, bq[step & 1][Ti >> 2][Ti & 3], accW[Ti]);
                accW[8] = mfma4(av, bt[step & 1], accW[8]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        FCW_STAMP(9)
#if FCW_X_PREFETCH == 2
        asm volatile("" ::"v"(pf));   // keep the prefetch loads alive (and waited for) until here
#endif
    }
    // ---- per-workgroup partial gradients -> workspace (the layout reduce_grads_kernel<N> sums)
    float *out = a.partials + (size_t)blockIdx.x * N::PART;
#if LTR_F16X2
    const float us = ldexpf(1.f, E - 28);
#else
    const float us = 1.f;
#endif
#pragma unroll
    for (int Ti = 0; Ti < XT; ++Ti) {
#if LTR_F16X2
        const int col = 16 * Ti + d;
#else
        const int col = Ti < 8 ? 64 * (Ti >> 2) + 4 * d + (Ti & 3) : 128 + d;       // the feature column d of tile Ti stands for (dW1 above)
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) out[N::P_W1 + (16 * w + 4 * q + r) * (XT * 16) + col] = accW[Ti][r] * us;
    }
    dw3 += __shfl_xor(dw3, 16, 64);
    dw3 += __shfl_xor(dw3, 32, 64);
    if (lane < 16) out[N::P_W3 + 16 * w + lane] = dw3;
    if (tid == 0) out[N::P_B3] = db3;
}

template <class N, int LOSS, int ST>
int launch_fcw_s(const PipeArgs &a, int grid, hipStream_t stream) {
    constexpr size_t lds = fcw_lds<N>();
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)fcw_fused_kernel<N, LOSS, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_done[dev] = true;
    }
    hipLaunchKernelGGL((fcw_fused_kernel<N, LOSS, ST>), dim3(grid), dim3(kFcwThreads), lds, stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

template <class N, int LOSS>
int launch_fcw(const PipeArgs &a, int grid, hipStream_t stream) {
    if constexpr (LOSS != 0) return launch_fcw_s<N, LOSS, 0>(a, grid, stream);
    else switch (a.S) {
        case 128: return launch_fcw_s<N, 0, 128>(a, grid, stream);
        case 64: return launch_fcw_s<N, 0, 64>(a, grid, stream);
        default: return launch_fcw_s<N, 0, 32>(a, grid, stream);
    }
}
