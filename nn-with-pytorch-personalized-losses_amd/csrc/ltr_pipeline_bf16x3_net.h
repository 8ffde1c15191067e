// ltr_pipeline_bf16x3_net.h -- network geometry of the split-precision (bf16 x 3) slate pipeline.
// Included by ltr_scorer.hip (inside its anonymous namespace) when LTR_SPLIT_BF16 is set; same template
// signature and member names as the fp32 NetT so the shared host code (ltr_net_info, reduce kernel, launchers)
// compiles against either.
//
// GEMM tiles are 16 x 16 outputs with K = 32 per v_mfma_f32_16x16x32_bf16.  Feature axes are padded to whole
// tiles: 16 on the output side (NT*), 32 on the contraction side (*K); the bias of each layer rides along as a
// constant-one input feature at index F / H1 (as in the fp32 pipeline).
template <int F_, int H1_, int H2_, int A1_, int A2_, int BH1_, int BH2_>
struct NetT {
    static constexpr int F = F_, H1 = H1_, H2 = H2_, A1 = A1_, A2 = A2_;
    static constexpr bool TWO = false;             // two-layer nets exist in the fp32 library only
    static constexpr int XT = (F + 1 + 15) / 16;    // x feature tiles of 16 incl. the ones feature (dW1 columns)
    static constexpr int XK = (F + 1 + 31) / 32;    // fc1 k-tiles of 32
    static constexpr int NT1 = (H1 + 15) / 16;      // fc1 output tiles
    static constexpr int H1T = (H1 + 1 + 15) / 16;  // h1 tiles incl. the ones feature at index H1 (dW2 columns)
    static constexpr int H1K = (H1 + 1 + 31) / 32;  // fc2 k-tiles
    static constexpr int NT2 = (H2 + 15) / 16;      // fc2 output tiles
    static constexpr int H2K = (H2 + 31) / 32;      // dh1 k-tiles (contraction over fc2 outputs)
    static constexpr int LD = (XT > H1T ? XT : H1T) * 16;   // row length (features) of the bf16 LDS images
    static constexpr int NW1 = NT1 * XT;            // dW1 tiles  [H1 rows][F+1 cols]
    static constexpr int NW2 = NT2 * H1T;           // dW2 tiles  [H2 rows][H1+1 cols]
    static constexpr int TW1 = (NW1 + kWaves - 1) / kWaves;
    static constexpr int TW2 = (NW2 + kWaves - 1) / kWaves;
    static constexpr int BH1 = BH1_, BH2 = BH2_;    // row-band height of the per-wave dW tile sets
    static constexpr int KMAX = XK > H1K ? (XK > H2K ? XK : H2K) : (H1K > H2K ? H1K : H2K);
    // packed weights: bf16 A-fragments [out tile][k tile][piece 0..2][lane 0..63][8 bf16] = 1 KiB per fragment;
    // offsets in BYTES, the whole buffer is counted in floats for the host (PACKED)
    static constexpr int W1F_OFF = 0;                              // [NT1][XK][3] fragments
    static constexpr int W2F_OFF = W1F_OFF + NT1 * XK * 3 * 1024;  // [NT2][H1K][3]
    static constexpr int W2T_OFF = W2F_OFF + NT2 * H1K * 3 * 1024; // [NT1][H2K][3]   (W2^T, for dh1)
    static constexpr int W3_OFF = W2T_OFF + NT1 * H2K * 3 * 1024;  // fp32: [NT2*16] w3 (zero padded) then b3 (+15 pad)
    static constexpr int PACKED = (W3_OFF + (NT2 * 16 + 16) * 4) / 4;
    // per-workgroup gradient partial (floats): same layout as the fp32 pipeline
    static constexpr int P_W1 = 0;                             // [NT1*16][XT*16]
    static constexpr int P_W2 = P_W1 + NT1 * 16 * XT * 16;     // [NT2*16][H1T*16]
    static constexpr int P_W3 = P_W2 + NT2 * 16 * H1T * 16;    // [NT2*16]
    static constexpr int P_B3 = P_W3 + NT2 * 16;
    static constexpr int PART = P_B3 + 16;
    static constexpr int NPARAM = H1 * F + H1 + H2 * H1 + H2 + H2 + 1;
    static_assert(F % 8 == 0, "input features are loaded in 8-float (32-byte) pieces");
    static_assert(H1 % 4 == 0 && H2 % 4 == 0, "feature counts must be multiples of 4");
    static_assert(NT1 % BH1 == 0 && NT2 % BH2 == 0, "band height must divide the dW row-tile count");
    static_assert(A1 == 0 /*ACT_ID*/ || A1 == 1 /*ACT_RELU_DROP*/, "layer-1 backward keeps only the sign of h1");
    // transposed LDS reads of the [doc][feature] bf16 images are conflict-free when the row stride (in dwords) is
    // 8 or 40 modulo 64 (four consecutive rows x four 8-byte pieces x two lane groups cover all 64 banks)
    static_assert((LD / 2) % 64 == 8 || (LD / 2) % 64 == 40, "bf16 image row stride must be 8 or 40 dwords mod 64");
};
