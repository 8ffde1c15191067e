// ltr_wst.h -- the fused slate pipeline of the live DoubleLayerNet (architeture/doubleLayer.py:54-66: 136 -> 136 -> 136 -> 1, ReLU +
// Dropout(0.5) twice), WEIGHTS-STATIONARY: the layout DESIGN.md section 7.1 proposed in round 3 and ltr_fcw.h proved on the two-layer
// net.  Exact fp32 (v_mfma_f32_16x16x4_f32).  Included by ltr_scorer.hip inside its anonymous namespace -- ONLY under -DLTR_WST=1:
// EXPERIMENT, round 4.  Parity-green on the whole fused test matrix (193 tests), measured SLOWER than the generic pipeline (0.567 of the
// fp32 MFMA peak with 8 waves, 0.529 with 4 waves of 512 registers, against 0.612): profiles/r04_wst_*.json* name the phases that lose.
//
// The generic pipeline (slate_pipeline_kernel) gives every wave 16 documents and ALL units of every layer: every wave streams every
// weight (8 x 330 KB per tile through L2 -> registers), and the weight gradients -- which contract over DOCUMENTS, the lane axis of
// that layout -- need dz and h staged through LDS per wave tile set (dW2 / dW1 ran at 68-73 % of their MFMA bound, round 3).
// Here a workgroup is 4 waves (one per SIMD, 512 registers each) and wave w owns hidden units 32 w .. 32 w + 31 of BOTH hidden layers
// for all 128 documents of a tile:
//   * every layer reads its input as ROWS of an LDS image [document][feature] (one ds_read_b128 feeds four MFMAs), its weights are the
//     wave's own 32 x 144 fragment set (18 KB from L2 per GEMM and tile instead of 330 KB), and its output lands in the accumulator
//     layout  lane (unit, q) register r  <->  z[document 16 T + 4 q + r][unit]  -- which IS the A operand of the weight gradient
//     (k-slot q <-> that document): dW2 / dW1 run with no staging and no barrier inside, B = rows of the image of the layer's input;
//   * activations travel between layers through ONE image region: h1 -> image -> fc2; after the loss, dW2 (A = dz2 registers,
//     B = the h1 image), then dz2 -> the same region -> dh1 = dz2 W2 -> dz1 registers -> dW1 (A = dz1 registers, B = the X image,
//     still in place).  X is read from HBM once; the next tile's X lands by LDS-DMA in the free region under dW1 and the two regions
//     swap roles per tile.
// 136 units are 8.5 MFMA tiles: the ninth half tile (units 128..135) is split by DOCUMENT tile -- wave w computes it for documents
// 32 w .. 32 w + 31 in the forward / dh1 GEMMs (1/4 of a tile's work each: balanced), and its weight-gradient rows are computed with
// the column tile as the owned axis (A = the 8 dz columns from a 4 KB side image, B = image rows; two accumulator tiles per wave).
// LDS: 2 x 71 680 (images) + loss arrays + partials + scratch = 158 720 B: one workgroup per CU.
#pragma once

#ifdef LTR_STAMPS
#define WST_STAMP(k)                                                                                          \
    if (a.stamps && lane == 0 && (st - (int)blockIdx.x) / (int)gridDim.x == a.stamp_tile)                      \
        a.stamps[((size_t)blockIdx.x * 8 + w) * 16 + (k)] = __builtin_readcyclecounter();
#else
#define WST_STAMP(k)
#endif

constexpr int kWstWaves = WST_WAVES;              // 4: one wave per SIMD, 512 registers each; 8: two per SIMD, 256 each
constexpr int kWstThreads = kWstWaves * 64;
constexpr int kWstUT = 8 / kWstWaves;             // unit tiles per wave
#ifndef WST_DMA_EARLY
#define WST_DMA_EARLY 0
#endif
#ifndef WST_X_DMA
#define WST_X_DMA 0                  // 1: X by LDS-DMA issued behind dW1; 0: through registers at the top of the tile
#endif

template <class N>
constexpr size_t wst_lds() {
    // two images [128][F + 4], 8 loss arrays [128], score partials [5][128], slate-group scratch [256 + 4*32 + 256], dropout keep words
    // [4 waves][128], the side image of the ninth half tile's dz columns [128][8]
    return sizeof(float) * ((size_t)2 * kTileDocs * (N::F + 4) + 8 * kTileDocs + (kWstWaves + 1) * kTileDocs + 640 + kWstWaves * kTileDocs +
                            kTileDocs * 8);
}

template <class N, int LOSS, int ST>
__global__ void __launch_bounds__(kWstThreads, kWstWaves / 4) wst_fused_kernel(const PipeArgs a) {
    static_assert(!N::TWO && N::F == 136 && N::H1 == 136 && N::H2 == 136, "built for the 136-136-136-1 net");
    static_assert(N::XT == 9 && N::H1T == 9 && N::NT1 == 9 && N::NT2 == 9, "eight whole unit tiles and one half tile per layer");
    static_assert(N::A1 == ACT_RELU_DROP && N::A2 == ACT_RELU_DROP, "ReLU + dropout layers");
    constexpr int UT = kWstUT;
    static_assert(UT * kWstWaves == 8, "the eight whole unit tiles are dealt to the waves");
    constexpr int LDX = N::F + 4;                    // floats per image row: 136 values, the ones feature, three zeros
    constexpr int LASTQ = (LDX - 128) / 4;           // 4-float groups of the last 16-feature k-tile that lie inside a row (3)
    constexpr int DT8 = 8 / kWstWaves;               // document tiles of the ninth half tile per wave (2)
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    float *img0 = smem_w, *img1 = img0 + kTileDocs * LDX;
    float *sc = img1 + kTileDocs * LDX;
    float *yl = sc + kTileDocs, *gn = yl + kTileDocs, *gg = gn + kTileDocs, *dsc = gg + kTileDocs, *uu = dsc + kTileDocs,
          *mk = uu + kTileDocs, *xt = mk + kTileDocs;
    float *part = xt + kTileDocs;                    // [5][128] score partials: wave w's 32 units; row 4 = units 128..135
    float *scratch = part + (kWstWaves + 1) * kTileDocs;   // [640]: slate-group scratch [0, 384), 1 / log2(2 + rank) [384, 512), spare
    unsigned *keepw = reinterpret_cast<unsigned *>(scratch + 640);                    // [4][128] dropout keep words (per wave)
    float *dzt8 = reinterpret_cast<float *>(keepw + kWstWaves * kTileDocs);            // [128][8] dz columns of units 128..135

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int q = lane >> 4, d = lane & 15;               // re-derived per tile from a laundered lane id (see the tile loop)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.packed), 0, N::PACKED * 4, 0x00020000);

    // ---- once per kernel
    float w3n[UT];                                   // w3 of my units 16 (UT w + u) + d; of unit 128 + d (zero padded past 136)
#pragma unroll
    for (int u = 0; u < UT; ++u) w3n[u] = a.packed[N::W3_OFF + 16 * (UT * w + u) + d];
    const float w3t = a.packed[N::W3_OFF + 128 + d];
    const float b3 = a.packed[N::W3_OFF + N::NT2 * 16];
    for (int e = tid; e < 2 * kTileDocs * 4; e += kWstThreads)     // pad columns of both images: ones feature, three zeros
        img0[(e >> 2) * LDX + N::F + (e & 3)] = (e & 3) ? 0.f : 1.f;
    for (int j = tid; j < 640; j += kWstThreads) scratch[j] = 0.f;
    if (LOSS == 0) {
        __syncthreads();
        ltr_fill_inv_discount(scratch + 384, kTileDocs, tid, kWstThreads);        // [384, 512): 1 / log2(2 + rank)
    }
    f32x4 accW1[UT][9], accW2[UT][9];                // my 32 rows of dW1 / dW2, nine column tiles (columns permuted, see dw_rows)
    f32x4 accW1t[UT], accW2t[UT], accW1u, accW2u;    // rows 128..143 x column tiles UT w + u (every wave) / x column tile 8 (wave 0)
#pragma unroll
    for (int u = 0; u < UT; ++u) {
#pragma unroll
        for (int t = 0; t < 9; ++t) accW1[u][t] = accW2[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        accW1t[u] = accW2t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    accW1u = accW2u = f32x4{0.f, 0.f, 0.f, 0.f};
    float dw3[UT] = {}, dw3t = 0.f, db3 = 0.f;

    // X tile -> image by LDS-DMA: wave w moves rows 32 w .. 32 w + 31 (one wave-instruction per row, lanes 0..33 active); rows past
    // the end of the batch are zero-filled
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    constexpr int RPW = kTileDocs / kWstWaves;
    auto dma_x = [&](int tile, float *img) {
        const long long row0 = (long long)tile * kTileDocs + RPW * w;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float *dst = img + (RPW * w + r) * LDX;                 // wave-uniform
            if (row0 + r < a.n_docs) {
                if (lane < N::F / 4)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a.X + (row0 + r) * N::F + 4 * lane), (lptr_t)dst, 16, 0, LTR_X_AUX);
            } else if (lane < N::F / 4) {
                *reinterpret_cast<f32x4 *>(dst + 4 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // z[doc][unit] for NU consecutive unit tiles (weights at wbytes, 9 KB per tile) over document tiles T0 .. T0 + NTT - 1:
    // A = image rows (lane (doc, q) reads floats 16 S + 4 q .. + 3 of its row: ONE ds_read_b128 per 4 NU MFMAs), B = the tiles' weight
    // fragments (lane (unit, q): W[unit][16 S + 4 q .. + 3]).  In the last k-tile the quads past the row (their weights are zero) are
    // taken from the row's last quad: finite, the lane's own row.  out[u * NTT + T].
    auto gemm_rows = [&](const float *img, int wbytes, int T0, auto nu_tag, auto ntt_tag, f32x4 *out) {
        constexpr int NU = decltype(nu_tag)::value, NTT = decltype(ntt_tag)::value;
        const float *ra = img + (16 * T0 + d) * LDX + 4 * q;
        const int last = 128 + (q < LASTQ ? 0 : 4 * (LASTQ - 1 - q));
        // k-groups of three k-tiles OUTERMOST: a group's weight fragments (3 per unit tile: 12 NU registers, the next group's in
        // flight) serve every document tile; the accumulators are the output tiles themselves.  Inside a group the document tiles
        // are software-pipelined: the three A fragments of tile T + 1 fly under the 12 NU MFMAs of tile T.
        f32x4 wf[2][NU][3];
        auto load_w = [&](int g, f32x4 (&dst)[NU][3]) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) dst[u][k] = load_frag(wrs, lane * 16, wbytes + (u * 9 + 3 * g + k) * 1024);
        };
        f32x4 fa[2][3];
        auto load_a = [&](int g, int T, f32x4 (&dst)[3]) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[k] = *reinterpret_cast<const f32x4 *>(ra + 16 * T * LDX + (3 * g + k < 8 ? 16 * (3 * g + k) : last));
        };
        load_w(0, wf[0]);
        load_a(0, 0, fa[0]);
        // (no zero-initialised accumulators: an output tile's FIRST MFMA takes the constant 0 as its C operand -- zero vectors kept in
        //  registers across the scheduling barriers below were hoisted out of the tile loop and re-read from scratch per GEMM)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (g + 1 < 3) load_w(g + 1, wf[(g + 1) & 1]);
#pragma unroll
            for (int T = 0; T < NTT; ++T) {
                const int n = g * NTT + T;                          // running unit index: A-fragment buffer parity
                if (T + 1 < NTT) load_a(g, T + 1, fa[(n + 1) & 1]);
                else if (g + 1 < 3) load_a(g + 1, 0, fa[(n + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int u = 0; u < NU; ++u)
                            out[u * NTT + T] = mfma4(fa[n & 1][k][i], wf[g & 1][u][k][i],
                                                     (g == 0 && k == 0 && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : out[u * NTT + T]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // my 16 UT rows of dW += dz^T [in | 1]: A = register r of the dz accumulator tile T (k-slot q = document 16 T + 4 q + r), B = the
    // input image's rows.  Which input feature a column stands for is free (the store at the end undoes it): column j of tile
    // Ti < 8 is feature 64 (Ti >> 2) + 4 j + (Ti & 3), so one ds_read_b128 feeds FOUR tiles of BOTH row tiles; tile 8 is feature
    // 128 + j (columns past the row read the row's last float, a zero pad).
    auto dw_rows = [&](const float *img, const f32x4 (&dz)[UT][8], f32x4 (&acc)[UT][9]) {
        const float *rb = img + 4 * q * LDX + 4 * d;
        const float *rt = img + 4 * q * LDX + (128 + d < LDX ? 128 + d : LDX - 1);
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
        for (int step = 0; step < 32; ++step) {
            if (step + 1 < 32) load_b(step + 1, bq[(step + 1) & 1], bt[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int Ti = 0; Ti < 8; ++Ti)
#pragma unroll
                for (int u = 0; u < UT; ++u) acc[u][Ti] = mfma4(dz[u][step >> 2][step & 3], bq[step & 1][Ti >> 2][Ti & 3], acc[u][Ti]);
#pragma unroll
            for (int u = 0; u < UT; ++u) acc[u][8] = mfma4(dz[u][step >> 2][step & 3], bt[step & 1], acc[u][8]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // rows 128..143 of dW (units 128..135 are real) x NC column tiles from column c0: A = the dz side image [doc][8] (k-slot q =
    // document 4 s + q, row i = unit 128 + i; rows i >= 8 are zero), B = image[doc][c0 + 16 c + j]
    auto dw_t8 = [&](const float *img, int c0, auto nc_tag, f32x4 *acc) {
        constexpr int NC = decltype(nc_tag)::value;
        const float *pa = dzt8 + q * 8 + (d & 7);
        const float *pb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pb[c] = img + q * LDX + (c0 + 16 * c + d < LDX ? c0 + 16 * c + d : LDX - 1);
        // operands of FOUR k-steps in flight ahead of their MFMAs; even / odd k-steps on separate accumulator chains (a lone
        // wave's dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles instead of 32)
        f32x4 alt[NC];
        float av[2][4], bv[2][4][NC];
        auto load = [&](int s0, float (&a4)[4], float (&b4)[4][NC]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a4[k] = pa[4 * (s0 + k) * 8];
#pragma unroll
                for (int c = 0; c < NC; ++c) b4[k][c] = pb[c][4 * (s0 + k) * LDX];
            }
        };
        load(0, av[0], bv[0]);
#pragma unroll
        for (int s0 = 0; s0 < 32; s0 += 4) {
            if (s0 + 4 < 32) load(s0 + 4, av[((s0 >> 2) + 1) & 1], bv[((s0 >> 2) + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a1 = d < 8 ? av[(s0 >> 2) & 1][k] : 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (k & 1) alt[c] = mfma4(a1, bv[(s0 >> 2) & 1][k][c], (s0 == 0 && k == 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : alt[c]);
                    else acc[c] = mfma4(a1, bv[(s0 >> 2) & 1][k][c], acc[c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] += alt[c];
    };
    // dropout keep words of layer L for my unit tiles (both lie in word w of a document's keep bits): hashed ONCE per document (two per
    // lane) and shared through this wave's private LDS row -- hashed per accumulator value it would be 16 x redundant
    auto keep_words = [&](int L, long long doc_base, unsigned (&kw)[8][4], unsigned (&kt)[DT8][4]) {
        unsigned *row = keepw + w * kTileDocs;
        row[lane] = keep_word(a.seed, L, doc_base + lane, (UT * w) >> 1);
        row[lane + 64] = keep_word(a.seed, L, doc_base + lane + 64, (UT * w) >> 1);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + 16 * T + 4 * q);      // same-wave LDS: in order
            kw[T][0] = v.x, kw[T][1] = v.y, kw[T][2] = v.z, kw[T][3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < DT8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) kt[t][r] = keep_word(a.seed, L, doc_base + 16 * (DT8 * w + t) + 4 * q + r, 4);   // units 128..135: word 4
    };
    // ReLU (+ dropout) on my accumulator tiles: unit 16 (UT w + u) + d is bit 16 u + d of word w; unit 128 + d is bit d of word 4
    auto activate_wst = [&](int L, const uint8_t *keep, long long doc_base, f32x4 (&h)[UT][8], f32x4 (&ht)[DT8]) {
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int T = 0; T < 8; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[u][T][r] = fmaxf(h[u][T][r], 0.f);
#pragma unroll
        for (int t = 0; t < DT8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ht[t][r] = d < 8 ? fmaxf(ht[t][r], 0.f) : 0.f;
        if (!a.dropout) return;                                              // (uniform)
        if (keep) {                                                          // explicit masks [n_docs][136] (tests)
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int T = 0; T < 8; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long long doc = doc_base + 16 * T + 4 * q + r;
                        const bool k = doc < a.n_docs && keep[doc * N::H1 + 16 * (UT * w + u) + d] != 0;
                        h[u][T][r] = k ? h[u][T][r] * a.drop_scale : 0.f;
                    }
#pragma unroll
            for (int t = 0; t < DT8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long doc = doc_base + 16 * (DT8 * w + t) + 4 * q + r;
                    const bool k = d < 8 && doc < a.n_docs && keep[doc * N::H1 + 128 + (d & 7)] != 0;
                    ht[t][r] = k ? ht[t][r] * a.drop_scale : 0.f;
                }
            return;
        }
        unsigned kw[8][4], kt[DT8][4];
        keep_words(L, doc_base, kw, kt);
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int T = 0; T < 8; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[u][T][r] = ((kw[T][r] >> (16 * ((UT * w + u) & 1) + d)) & 1u) ? h[u][T][r] * a.drop_scale : 0.f;
#pragma unroll
        for (int t = 0; t < DT8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ht[t][r] = ((kt[t][r] >> d) & 1u) ? ht[t][r] * a.drop_scale : 0.f;       // (d >= 8: already 0)
    };
    // accumulator tiles -> image columns of my units: image[doc 16 T + 4 q + r][16 (UT w + u) + d]; units 128..135 of my document tiles
    auto to_image = [&](float *img, const f32x4 (&h)[UT][8], const f32x4 (&ht)[DT8]) {
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int T = 0; T < 8; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) img[(16 * T + 4 * q + r) * LDX + 16 * (UT * w + u) + d] = h[u][T][r];
        if (d < 8) {
#pragma unroll
            for (int t = 0; t < DT8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) img[(16 * (DT8 * w + t) + 4 * q + r) * LDX + 128 + d] = ht[t][r];
        }
    };
    auto t8_columns = [&](const f32x4 (&ht)[DT8]) {   // dz of units 128..135, my document tiles -> side image [doc][8]
        if (d < 8) {
#pragma unroll
            for (int t = 0; t < DT8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dzt8[(16 * (DT8 * w + t) + 4 * q + r) * 8 + d] = ht[t][r];
        }
    };

    // the same through registers: RPW rows of 34 float4 per wave = RPW * 34 / 64 loads per lane (buffer descriptor sized to the rows that
    // exist: rows past the end of the batch read zeros), then ds_write_b128 into the padded rows
    constexpr int XV4 = RPW * (N::F / 4), XNV = (XV4 + 63) / 64;
    auto load_x_regs = [&](int tile, f32x4 (&xn)[XNV]) {
        const long long row0 = (long long)tile * kTileDocs + RPW * w;
        long long rows_here = a.n_docs - row0;
        rows_here = rows_here < 0 ? 0 : (rows_here > RPW ? RPW : rows_here);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.X + (rows_here ? row0 : 0) * N::F), 0,
                                                                             (int)rows_here * N::F * 4, 0x00020000);
#pragma unroll
        for (int m = 0; m < XNV; ++m)
            xn[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (lane + 64 * m) * 16, 0, 2 /* nt */));
    };
    auto store_x_regs = [&](float *img, const f32x4 (&xn)[XNV]) {
#pragma unroll
        for (int m = 0; m < XNV; ++m) {
            const int e = lane + 64 * m;
            if (e < XV4) *reinterpret_cast<f32x4 *>(img + (RPW * w + e / (N::F / 4)) * LDX + 4 * (e % (N::F / 4))) = xn[m];
        }
    };
    float *Xi = img0, *Hi = img1;                     // the image that holds X / the image activations travel through (swap per tile)
#if WST_X_DMA
    if ((int)blockIdx.x < a.n_super) dma_x(blockIdx.x, Xi);
#endif
    using one_t = std::integral_constant<int, 1>;
    using ut_t = std::integral_constant<int, UT>;
    using dt8_t = std::integral_constant<int, DT8>;
    using eight_t = std::integral_constant<int, 8>;
    const int wb1 = (N::W1F_OFF + UT * w * 9 * 256) * 4, wb1t = (N::W1F_OFF + 8 * 9 * 256) * 4;
    const int wb2 = (N::W2F_OFF + UT * w * 9 * 256) * 4, wb2t = (N::W2F_OFF + 8 * 9 * 256) * 4;
    const int wbT = (N::W2T_OFF + UT * w * 9 * 256) * 4, wbTt = (N::W2T_OFF + 8 * 9 * 256) * 4;

    for (int st = blockIdx.x; st < a.n_super; st += gridDim.x) {
        const long long doc_base = (long long)st * kTileDocs;
        {   // Per-lane geometry is re-derived per tile from a laundered lane id: left loop-invariant, hipcc hoists the ~170 LDS
            // addresses and constants built from it out of the persistent loop and parks them in scratch (172 spill stores in front of
            // the loop, 160 reloads per tile inside the dW GEMMs -- each a vector-memory round trip at one wave per SIMD)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            q = ln >> 4;
            d = ln & 15;
        }
        WST_STAMP(0)
#if WST_X_DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my rows of this tile's X have landed
#else
        {
            f32x4 xn[XNV];
            load_x_regs(st, xn);
            store_x_regs(Xi, xn);                             // (Xi was the activation image of the previous tile: free since its barrier G)
        }
#endif
        if (tid < kTileDocs) {
            const long long doc = doc_base + tid;
            const float y = doc < a.n_docs ? a.labels[doc] : a.pad;
            if (LOSS != 1) stage_label(y, a.pad, yl[tid], gn[tid]);
            else yl[tid] = doc < a.n_docs ? y : 0.f;
        }
        __syncthreads();                                      // A: X complete; every wave is done with the previous tile
        WST_STAMP(1)
        // ---- fc1 -> h1 (registers) -> image
        f32x4 h1[UT][8], h1t[DT8], h2[UT][8], h2t[DT8];
        gemm_rows(Xi, wb1, 0, ut_t(), eight_t(), &h1[0][0]);
        gemm_rows(Xi, wb1t, DT8 * w, one_t(), dt8_t(), &h1t[0]);
        activate_wst(0, a.keep1, doc_base, h1, h1t);
        to_image(Hi, h1, h1t);
        // From here on h1 is needed as a VALUE only through its image (fc2's A rows, dW2's B rows); the backward needs just
        // "h1 > 0" per unit for dz1 = dh1 act'(h1): 72 registers become 3 words of bits until then
        unsigned hb[UT], hbt = 0u;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            hb[u] = 0u;
#pragma unroll
            for (int T = 0; T < 8; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[u] |= (h1[u][T][r] > 0.f ? 1u : 0u) << (4 * T + r);
        }
#pragma unroll
        for (int t = 0; t < DT8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) hbt |= (h1t[t][r] > 0.f ? 1u : 0u) << (4 * t + r);
        WST_STAMP(2)
        __syncthreads();                                      // B: the h1 image is complete
        // ---- fc2 -> h2 (registers); score partials w3 . h2 over my 32 units (and units 128..135 of my document tiles)
        gemm_rows(Hi, wb2, 0, ut_t(), eight_t(), &h2[0][0]);
        gemm_rows(Hi, wb2t, DT8 * w, one_t(), dt8_t(), &h2t[0]);
        activate_wst(1, a.keep2, doc_base, h2, h2t);
        WST_STAMP(3)
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            f32x4 sv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = h2[0][T][r] * w3n[0];
#pragma unroll
                for (int u = 1; u < UT; ++u) v = fmaf(h2[u][T][r], w3n[u], v);
                sv[r] = row_sum_to_lane15(v);
            }
            if (d == 15) *reinterpret_cast<f32x4 *>(part + w * kTileDocs + 16 * T + 4 * q) = sv;
        }
#pragma unroll
        for (int t = 0; t < DT8; ++t) {
            f32x4 sv;
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = row_sum_to_lane15(h2t[t][r] * w3t);
            if (d == 15) *reinterpret_cast<f32x4 *>(part + kWstWaves * kTileDocs + 16 * (DT8 * w + t) + 4 * q) = sv;
        }
        __syncthreads();                                      // D: the five partials of every document are out
        WST_STAMP(4)
        // ---- listwise loss on the LDS-resident scores -> dsc (2 threads per document row)
        {
            auto score_of = [&](int j) {
                const float *pp = part + j;
                float s4 = (pp[0] + pp[kTileDocs]) + (pp[2 * kTileDocs] + pp[3 * kTileDocs]);
                if (kWstWaves == 8) s4 += (pp[4 * kTileDocs] + pp[5 * kTileDocs]) + (pp[6 * kTileDocs] + pp[7 * kTileDocs]);
                return s4 + (pp[kWstWaves * kTileDocs] + b3);
            };
            const int S_ = ST ? ST : a.S;
            const int group = (kWstThreads / kTileDocs) * S_;
            const int gid = tid / group;
            const int so = gid * S_;
            const long long slate = (long long)st * (kTileDocs / S_) + gid;
            float loss;
            if constexpr (LOSS == 0) {
                auto stamp_fn = [&](int k) { WST_STAMP(k) };
                constexpr int SS = ST ? ST : 128, NWS = SS * kWstWaves / kTileDocs;
                const int gi = tid / (64 * NWS), so2 = gi * SS;
                loss = approx_ndcg_fused<SS, NWS, true>(tid - gi * 64 * NWS, [&](int j) { return score_of(so2 + j); }, sc + so2, yl + so2, gn + so2,
                                                        gg + so2, uu + so2, xt + so2, mk + so2, scratch + gi * 16, scratch + 384, a.alpha, a.eps,
                                                        a.gscale, [&](int i, float v) { dsc[so2 + i] = v; }, stamp_fn);
            } else {
                if (tid < kTileDocs) sc[tid] = score_of(tid);
                __syncthreads();
                const SlateGroup g = make_group(S_, group, scratch + gid * (group + 32), tid);
                if constexpr (LOSS == 1) {
                    loss = listnet_slate(g, yl + so, sc + so, a.apply_sigmoid != 0, a.gscale, true,
                                         [&](int i, float v) { dsc[so + i] = slate < a.B ? v : 0.f; });
                } else {
                    LambdaLds L;
                    L.sc = sc + so; L.yl = yl + so; L.gn = gn + so; L.w1 = gg + so; L.invd = uu + so; L.delta = mk + so;
                    L.rk = reinterpret_cast<int *>(xt + so);
                    float count;
                    loss = lambda_slate<-1>(g, L, a.lp, a.gscale, true, &count, [&](int i, float v) { dsc[so + i] = slate < a.B ? v : 0.f; });
                    if (tid - gid * group == 0 && slate < a.B && a.slate_count) a.slate_count[slate] = count;
                }
            }
            if (tid - gid * group == 0 && slate < a.B) a.slate_loss[slate] = loss;
        }
        __syncthreads();                                      // E: d loss / d score of the whole tile (0 for documents past the batch)
        WST_STAMP(5)
        // ---- backward through fc3: dw3 += ds h2, dz2 = ds w3 act'(h2) in place
        const float slope = a.dropout ? a.drop_scale : 1.f;
        if (w == 0) db3 += wave_allsum(dsc[lane] + dsc[lane + 64]);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const f32x4 ds4 = *reinterpret_cast<const f32x4 *>(dsc + 16 * T + 4 * q);
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dw3[u] = fmaf(ds4[r], h2[u][T][r], dw3[u]);
                    h2[u][T][r] = apply_act_grad<N::A2>(ds4[r] * slope * w3n[u], h2[u][T][r]);
                }
        }
#pragma unroll
        for (int t = 0; t < DT8; ++t) {
            const f32x4 ds4 = *reinterpret_cast<const f32x4 *>(dsc + 16 * (DT8 * w + t) + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dw3t = fmaf(ds4[r], h2t[t][r], dw3t);
                h2t[t][r] = apply_act_grad<N::A2>(ds4[r] * slope * w3t, h2t[t][r]);
            }
        }
        t8_columns(h2t);
        WST_STAMP(6)
        // ---- dW2: my rows straight from the dz2 registers against the h1 image; then rows 128..143 from the side image
        dw_rows(Hi, h2, accW2);
        __syncthreads();                                      // F1: the dz2 side image is complete
        dw_t8(Hi, 16 * UT * w, ut_t(), accW2t);
        if (w == 0) dw_t8(Hi, 128, one_t(), &accW2u);
        WST_STAMP(7)
        __syncthreads();                                      // F2: every wave is done with the h1 image and the side image
        to_image(Hi, h2, h2t);                                // dz2 -> the image region
        __syncthreads();                                      // F3
        // ---- dh1 = dz2 W2 -> dz1 = dh1 act'(h1) in place of h1
        {
            gemm_rows(Hi, wbT, 0, ut_t(), eight_t(), &h1[0][0]);
            gemm_rows(Hi, wbTt, DT8 * w, one_t(), dt8_t(), &h1t[0]);
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int T = 0; T < 8; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[u][T][r] = ((hb[u] >> (4 * T + r)) & 1u) ? h1[u][T][r] * slope : 0.f;
#pragma unroll
            for (int t8 = 0; t8 < DT8; ++t8)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1t[t8][r] = (d < 8 && ((hbt >> (4 * t8 + r)) & 1u)) ? h1t[t8][r] * slope : 0.f;
        }
        t8_columns(h1t);
        WST_STAMP(8)
        __syncthreads();                                      // G: the dz1 side image is complete; the activation image is free
#if WST_X_DMA && WST_DMA_EARLY
        if (st + (int)gridDim.x < a.n_super) dma_x(st + gridDim.x, Hi);     // the next tile's X lands under dW1
#endif
        // ---- dW1
        dw_rows(Xi, h1, accW1);
        WST_STAMP(15)
        dw_t8(Xi, 16 * UT * w, ut_t(), accW1t);
        if (w == 0) dw_t8(Xi, 128, one_t(), &accW1u);
#if WST_X_DMA && !WST_DMA_EARLY
        // the next tile's X -> the (free) activation image.  Behind dW1, not under it: while an LDS-DMA is pending hipcc turns every
        // vmcnt wait into vmcnt(0), and any vector-memory access of the GEMM (a spill reload is enough) then waits out the whole HBM
        // round trip -- dW1 took 97.8 k cycles instead of 20.5 k with the DMA in front of it (profiles/r04_variant_ab.json)
        if (st + (int)gridDim.x < a.n_super) dma_x(st + gridDim.x, Hi);
#endif
        WST_STAMP(9)
        float *tmp = Xi;
        Xi = Hi;
        Hi = tmp;
    }
    // ---- per-workgroup partial gradients -> workspace (the layout reduce_grads_kernel<N> sums)
    float *out = a.partials + (size_t)blockIdx.x * N::PART;
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int Ti = 0; Ti < 9; ++Ti) {
            const int col = Ti < 8 ? 64 * (Ti >> 2) + 4 * d + (Ti & 3) : 128 + d;     // the input feature column d of tile Ti stands for
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[N::P_W1 + (16 * (UT * w + u) + 4 * q + r) * (N::XT * 16) + col] = accW1[u][Ti][r];
                out[N::P_W2 + (16 * (UT * w + u) + 4 * q + r) * (N::H1T * 16) + col] = accW2[u][Ti][r];
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            out[N::P_W1 + (128 + 4 * q + r) * (N::XT * 16) + 16 * (UT * w + u) + d] = accW1t[u][r];
            out[N::P_W2 + (128 + 4 * q + r) * (N::H1T * 16) + 16 * (UT * w + u) + d] = accW2t[u][r];
        }
        if (w == 0) {
            out[N::P_W1 + (128 + 4 * q + r) * (N::XT * 16) + 128 + d] = accW1u[r];
            out[N::P_W2 + (128 + 4 * q + r) * (N::H1T * 16) + 128 + d] = accW2u[r];
        }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        float v = dw3[u];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) out[N::P_W3 + 16 * (UT * w + u) + lane] = v;
    }
    dw3t += __shfl_xor(dw3t, 16, 64);
    dw3t += __shfl_xor(dw3t, 32, 64);
    __syncthreads();
    if (lane < 16) scratch[16 * w + lane] = dw3t;             // units 128..143: one partial per wave (its document tiles)
    __syncthreads();
    if (tid < 16) {
        float s = 0.f;
        for (int ww = 0; ww < kWstWaves; ++ww) s += scratch[16 * ww + tid];
        out[N::P_W3 + 128 + tid] = s;
    }
    if (tid == 0) out[N::P_B3] = db3;
}

template <class N, int LOSS, int ST>
int launch_wst_s(const PipeArgs &a, int grid, hipStream_t stream) {
    constexpr size_t lds = wst_lds<N>();
    static_assert(lds <= 163840, "one workgroup must fit the CU's 160 KiB of LDS");
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)wst_fused_kernel<N, LOSS, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_done[dev] = true;
    }
    hipLaunchKernelGGL((wst_fused_kernel<N, LOSS, ST>), dim3(grid), dim3(kWstThreads), lds, stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

template <class N, int LOSS>
int launch_wst(const PipeArgs &a, int grid, hipStream_t stream) {
    if constexpr (LOSS != 0) return launch_wst_s<N, LOSS, 0>(a, grid, stream);
    else switch (a.S) {
        case 128: return launch_wst_s<N, 0, 128>(a, grid, stream);
        case 64: return launch_wst_s<N, 0, 64>(a, grid, stream);
        default: return launch_wst_s<N, 0, 32>(a, grid, stream);
    }
}
