// ltr_pipeline_bf16x3.h -- split-precision slate pipeline (gfx950): the SAME computation as the fp32 pipeline in
// ltr_scorer.hip (scorer forward -> listwise loss -> scorer backward -> weight gradients, one persistent workgroup
// per CU, documents on the MFMA lane axis, activations resident in registers between layers), with every GEMM
// running on the bf16 matrix cores at fp32-level accuracy:
//
//     a = a1 + a2 + a3   (three bf16 pieces, round-to-nearest residual split; 3 x 8 significand bits)
//     a * b  ~  a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (the six products down to 2^-16 ... 2^-24)
//
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 MFMAs of 16 cycles per 16x16x32 block against 8 MFMAs of 32
// cycles for the same block on v_mfma_f32_16x16x4_f32 -- 2.7x the matrix rate; measured error against the fp64
// oracle is at or below plain fp32's (profiles/r02_split_precision_study.json).  The weight-gradient GEMMs use the
// same three pieces and six products: with two pieces a strongly cancelling bias gradient came out 1.7e-4 off.
//
// What changes structurally against the fp32 kernel:
//   * Weight fragments are SHARED: 3 KiB per (out tile, k tile) instead of 1 KiB would put ~25 TB/s on L2 if every
//     wave streamed its own copy, so one copy per workgroup goes L2 -> LDS by LDS-DMA into a two-slot ring (one slot =
//     one output-tile row of a layer, 15 KiB), one barrier per slot, and all 8 waves read it (ds_read_b128).
//   * X is loaded fragment-shaped straight into registers (8 floats per lane per k tile) and split there; no fp32 X
//     tile lives in LDS.  It is loaded a second time (L2 / Infinity Cache hit) for the dW1 GEMM at the end of the tile.
//   * The dW GEMMs contract over documents: operands live in LDS as [doc][feature] bf16 images, one per piece, and are
//     read TRANSPOSED with ds_read_b64_tr_b16 (row stride 8 or 40 dwords mod 64 -> conflict-free).  The B side (h1 for
//     dW2, x for dW1) is PARKED for all 128 documents (h1's pieces are written right after fc1, which also takes the
//     60 registers they occupy out of the loss phase); the A side (dz) is staged 32 documents at a time in the region
//     the weight ring uses during the forward and dh1 GEMMs.
// Included by ltr_scorer.hip inside its anonymous namespace when LTR_SPLIT_BF16 is set.

#include "ltr_bf16_split.h"

// ---------------------------------------------------------------------------------------------------- weight ring
// C slots of slot_bytes in LDS (the parked-operand region + the staging region: both are free while a GEMM phase runs).
// A slot holds the A-fragments of ONE output-tile row of a layer: [k tile][piece][64 lanes][16 B].  Fragment i of a
// slot is fetched by wave i % 8 (one 1-KiB LDS-DMA each).  L2 -> LDS takes ~1.1 us from issue to landed (2.6 k cycles)
// against ~1 k cycles of MFMA work per slot, so the ring runs C - 1 slots ahead: at the barrier that opens epoch g
// every wave has finished epoch g - 1, whose slot then receives epoch g + C - 1.
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__host__ __device__ constexpr int frags_of_wave(int nfrag, int w) { return nfrag > w ? (nfrag - w + kWaves - 1) / kWaves : 0; }

// s_waitcnt vmcnt(n), n a run-time wave-uniform value (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define LTR_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        LTR_VMCNT_CASE(0) LTR_VMCNT_CASE(1) LTR_VMCNT_CASE(2) LTR_VMCNT_CASE(3) LTR_VMCNT_CASE(4) LTR_VMCNT_CASE(5)
        LTR_VMCNT_CASE(6) LTR_VMCNT_CASE(7) LTR_VMCNT_CASE(8) LTR_VMCNT_CASE(9) LTR_VMCNT_CASE(10) LTR_VMCNT_CASE(11)
        LTR_VMCNT_CASE(12) LTR_VMCNT_CASE(13) LTR_VMCNT_CASE(14) LTR_VMCNT_CASE(15) LTR_VMCNT_CASE(16) LTR_VMCNT_CASE(17)
        LTR_VMCNT_CASE(18) LTR_VMCNT_CASE(19) LTR_VMCNT_CASE(20) LTR_VMCNT_CASE(21) LTR_VMCNT_CASE(22) LTR_VMCNT_CASE(23)
        LTR_VMCNT_CASE(24)
#undef LTR_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// One ring PHASE = the epochs of up to two consecutive GEMMs (layer A: NA epochs of KA k tiles at byte offset OA of the
// packed buffer; layer B likewise, NB may be 0).  Epoch g sits in slot g % C.
template <int NA_, int KA_, int OA_, int NB_, int KB_, int OB_, int C_, int SLOT_>
struct RingPhase {
    static constexpr int NA = NA_, KA = KA_, OA = OA_, NB = NB_, KB = KB_, OB = OB_, C = C_, SLOT = SLOT_, TOT = NA_ + NB_;
    static_assert(C_ >= 2, "the weight ring needs at least two slots");
    __host__ __device__ static constexpr int nfrag(int g) { return (g < NA ? KA : KB) * 3; }
    __host__ __device__ static constexpr int src(int g) { return g < NA ? OA + g * KA * 3 * 1024 : OB + (g - NA) * KB * 3 * 1024; }
    // this wave's LDS-DMA instructions for epochs (g, last issued before epoch g's wait]
    __host__ __device__ static constexpr int later(int g, int w) {
        int n = 0;
        for (int j = g + 1; j < TOT && j <= g + C - 2; ++j) n += frags_of_wave(nfrag(j), w);
        return n;
    }
};

template <class P>
__device__ __forceinline__ void ring_issue(char *ring, const char *packed, int g, int w, int lane) {
    char *dst = ring + (g % P::C) * P::SLOT;
    const int nf = P::nfrag(g), sb = P::src(g);
    for (int i = w; i < nf; i += kWaves)     // wave-uniform trip count
        __builtin_amdgcn_global_load_lds((gptr_t)(packed + sb + i * 1024 + lane * 16), (lptr_t)(dst + i * 1024), 16, 0, 0);
}

// start a phase: the first C - 1 epochs go out at once (the caller guarantees that the whole ring region is free)
template <class P>
__device__ __forceinline__ void ring_start(char *ring, const char *packed, int w, int lane) {
#pragma unroll
    for (int g = 0; g < P::C - 1 && g < P::TOT; ++g) ring_issue<P>(ring, packed, g, w, lane);
}

// One GEMM of a phase: epochs [G0, G0 + NT), out[To] = sum over KT k tiles of the 6 piece products.
template <class P, int G0, int NT, int KT, int KMAX, int NMAX, class Stamp = NoStamp>
__device__ __forceinline__ void gemm_ring(char *ring, const char *packed, const u32x4 (&bin)[KMAX][3], f32x4 (&out)[NMAX],
                                          int w, int lane, int diag = 0, Stamp stamp = Stamp()) {
#pragma unroll
    for (int To = 0; To < NT; ++To) {
        const int g = G0 + To;
        // this epoch's fragments have landed: each wave waits for its own DMAs (all but the ones it issued for later
        // epochs), the barrier publishes them and tells everyone that epoch g - 1 has been read by all waves
        int later = 0;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) later = (w == ww) ? P::later(g, ww) : later;
#ifdef LTR_DIAG
        if (!(diag & 32))
#endif
        wait_vmcnt(later);
        if (To == 1) stamp(10);
#ifdef LTR_DIAG
        if (!(diag & 128))
#endif
        // RAW barrier: __syncthreads() carries a fence that hipcc lowers to s_waitcnt vmcnt(0) while LDS-DMAs are
        // pending -- it would drain the C - 1 epochs in flight at every epoch (measured: 2.2 k cycles per epoch, the
        // full L2 -> LDS latency).  This wave's earlier LDS reads were consumed by MFMAs already; its LDS-DMAs for
        // this epoch were waited for just above.
        __builtin_amdgcn_s_barrier();
        if (To == 1) stamp(11);
        // An LDS-DMA costs its issuing wave 100-200 cycles of issue time.  The two waves of a SIMD (w and w + 4) leave
        // the barrier together: the first half refills before its MFMAs, the second half after them, so one wave's
        // DMA issue runs under the other's matrix work instead of both stalling the pipe at the top of the epoch.
        if (w < kWaves / 2 && g + P::C - 1 < P::TOT) ring_issue<P>(ring, packed, g + P::C - 1, w, lane);
        if (To == 1) stamp(12);
        const char *slot = ring + (g % P::C) * P::SLOT + lane * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        u32x4 a0 = *reinterpret_cast<const u32x4 *>(slot + 0 * 1024);
        u32x4 a1 = *reinterpret_cast<const u32x4 *>(slot + 1 * 1024);
        u32x4 a2 = *reinterpret_cast<const u32x4 *>(slot + 2 * 1024);
#pragma unroll
        for (int T = 0; T < KT; ++T) {
            u32x4 n0, n1, n2;
            if (T + 1 < KT) {       // next k tile's fragments are in flight under this one's MFMAs
                n0 = *reinterpret_cast<const u32x4 *>(slot + ((T + 1) * 3 + 0) * 1024);
                n1 = *reinterpret_cast<const u32x4 *>(slot + ((T + 1) * 3 + 1) * 1024);
                n2 = *reinterpret_cast<const u32x4 *>(slot + ((T + 1) * 3 + 2) * 1024);
            }
#ifdef LTR_DIAG
            if (diag & 64) { acc[0] += __builtin_bit_cast(float, a0[0] ^ a1[1] ^ a2[2]); } else
#endif
            {
            // small products first, the leading one last
            acc = mfma_bf16(a2, bin[T][0], acc);
            acc = mfma_bf16(a0, bin[T][2], acc);
            acc = mfma_bf16(a1, bin[T][1], acc);
            acc = mfma_bf16(a1, bin[T][0], acc);
            acc = mfma_bf16(a0, bin[T][1], acc);
            acc = mfma_bf16(a0, bin[T][0], acc);
            }
            if (T + 1 < KT) {
                a0 = n0;
                a1 = n1;
                a2 = n2;
            }
        }
        out[To] = acc;
        if (w >= kWaves / 2 && g + P::C - 1 < P::TOT) ring_issue<P>(ring, packed, g + P::C - 1, w, lane);
        if (To == 1) stamp(13);
        if (To == 2) stamp(14);
    }
}

// This lane's 8 features of every k tile of one document row of X (fragment-shaped, natural feature order), split into
// three pieces; the ones feature (bias) sits at index F, everything past it is zero.
template <class N, bool NT_HINT>
__device__ __forceinline__ void load_x_operand(u32x4 (&xb)[N::KMAX][3], const float *X, long long gdoc, long long n_docs, int q) {
    const bool in_range = gdoc < n_docs;
    const float *xrow = X + gdoc * N::F;
#pragma unroll
    for (int T = 0; T < N::XK; ++T) {
        const int f0 = 32 * T + 8 * q;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (in_range && f0 + 8 <= N::F) {
            if (NT_HINT) {
                v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xrow + f0));
                v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xrow + f0 + 4));
            } else {
                v0 = *reinterpret_cast<const f32x4 *>(xrow + f0);
                v1 = *reinterpret_cast<const f32x4 *>(xrow + f0 + 4);
            }
        }
        if (f0 == N::F) v0[0] = 1.f;                      // the ones feature (carries b1 / db1)
        unsigned p0[3], p1[3], p2[3], p3[3];
        split2<3>(v0[0], v0[1], p0);
        split2<3>(v0[2], v0[3], p1);
        split2<3>(v1[0], v1[1], p2);
        split2<3>(v1[2], v1[3], p3);
#pragma unroll
        for (int p = 0; p < 3; ++p) xb[T][p] = u32x4{p0[p], p1[p], p2[p], p3[p]};
    }
}

template <class N>
constexpr int park_plane() { return kTileDocs * N::LD; }              // elements of one [128][LD] piece image
template <class N>
constexpr int stage_plane() { return 32 * N::LD; }                    // elements of one [32][LD] piece image
template <class N>
constexpr int split_slot_bytes() { return N::KMAX * 3 * 1024; }
template <class N>
constexpr size_t stage_region_bytes() {  // the three [32][LD] A-operand images of a dW k step, rounded up to whole ring slots
    return ((size_t)3 * stage_plane<N>() * 2 + split_slot_bytes<N>() - 1) / split_slot_bytes<N>() * split_slot_bytes<N>();
}
template <class N>
constexpr size_t image_region_bytes() {  // parked B-operand images (3 pieces x [128][LD] bf16) + A staging = the ring, in GEMM phases
    return (size_t)3 * park_plane<N>() * 2 + stage_region_bytes<N>();
}
template <class N>
constexpr int ring_slots() { return (int)(image_region_bytes<N>() / split_slot_bytes<N>()); }
template <class N>
constexpr size_t pipeline_lds() {
    return image_region_bytes<N>() +
           sizeof(float) * (size_t)(8 * kTileDocs + N::NT2 * 16 + 16 + kWaves * N::NT2 * 16 + kThreads + 4 * 32 + 64);
}

template <class N, int MODE, int LOSS>
__global__ void __launch_bounds__(kThreads, 2) slate_pipeline_kernel(const PipeArgs a) {
    constexpr int LD = N::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = park_plane<N>();           // elements per parked piece image [128][LD]
    constexpr int SP = stage_plane<N>();          // elements per staged piece image [32][LD]
    char *lds = reinterpret_cast<char *>(smem);
    unsigned short *park = reinterpret_cast<unsigned short *>(lds);              // 3 x [128][LD]: h1 pieces, later x pieces
    unsigned short *stage = park + 3 * PP;                                       // 3 x [32][LD] A-operand images of a dW k step
    // during the GEMM phases (fc1 + fc2, dh1) the whole image region is the weight ring
    typedef RingPhase<N::NT1, N::XK, N::W1F_OFF, N::NT2, N::H1K, N::W2F_OFF, ring_slots<N>(), split_slot_bytes<N>()> PhaseFwd;
    typedef RingPhase<N::NT1, N::H2K, N::W2T_OFF, 0, 1, 0, ring_slots<N>(), split_slot_bytes<N>()> PhaseBwd;
    float *sc = reinterpret_cast<float *>(lds + image_region_bytes<N>());   // [128] scores
    float *yl = sc + kTileDocs;               // [128] labels (-inf padded)
    float *gn = yl + kTileDocs;               // [128] gains
    float *gg = gn + kTileDocs;               // [128] loss scratch
    float *dsc = gg + kTileDocs;              // [128] d loss / d score
    float *uu = dsc + kTileDocs;              // [128] loss scratch
    float *mk = uu + kTileDocs;               // [128] loss scratch
    float *xt = mk + kTileDocs;               // [128] loss scratch (LambdaLoss ranks)
    float *w3s = xt + kTileDocs;              // [NT2*16 + 16] w3 (zero padded), b3
    float *dw3 = w3s + N::NT2 * 16 + 16;      // [kWaves][NT2*16] per-wave dw3 accumulators
    float *scratch = dw3 + kWaves * N::NT2 * 16;   // [512 + 4*32] slate-group scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, d = lane & 15;
    const char *packed = reinterpret_cast<const char *>(a.packed);
    const float *w3g = reinterpret_cast<const float *>(packed + N::W3_OFF);

    for (int j = tid; j < N::NT2 * 16 + 16; j += kThreads) w3s[j] = w3g[j];
    for (int j = tid; j < kWaves * N::NT2 * 16; j += kThreads) dw3[j] = 0.f;
    float db3 = 0.f;
    f32x4 accW1[N::TW1], accW2[N::TW2];
    if (MODE != MODE_FWD) {
#pragma unroll
        for (int n = 0; n < N::TW1; ++n) accW1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < N::TW2; ++n) accW2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int my_row = 16 * w + d;                         // this lane's document inside the super-tile

    const int lane_outer = lane;
    for (int st = blockIdx.x; st < a.n_super; st += gridDim.x) {
        const long long doc_base = (long long)st * kTileDocs;
        // per-lane geometry re-derived per tile from a laundered lane id (see the fp32 kernel: keeps hipcc from hoisting
        // dozens of LDS addresses out of the persistent loop and spilling them)
        int lane = lane_outer;
        asm volatile("" : "+v"(lane));
        const int q = lane >> 4, d = lane & 15;
        const int my_row = 16 * w + d;
        const int tid = 64 * w + lane;
        const long long gdoc = doc_base + my_row;
        LTR_STAMP(0)
        __syncthreads();   // previous super-tile done with the images, the staging region and sc / dsc
        ring_start<PhaseFwd>(lds, packed, w, lane);      // fc1's weight rows are on their way (fc2's follow as slots free up)
        // ---- X: 8 floats per lane per k tile straight from HBM/L2 (the previous tile's backward prefetched the lines
        //      into L2), split into three bf16 pieces = fc1's B operand; the two leading pieces also go to the X images
        u32x4 xb[N::KMAX][3];
        load_x_operand<N, false>(xb, a.X, gdoc, a.n_docs, q);
        if (MODE == MODE_FUSED && tid < kTileDocs) {
            const long long doc = doc_base + tid;
            const float y = doc < (long long)a.B * a.S ? a.labels[doc] : a.pad;
            if (LOSS != 1) stage_label(y, a.pad, yl[tid], gn[tid]);
            else yl[tid] = doc < (long long)a.B * a.S ? y : 0.f;
        }
        if (MODE == MODE_BWD && tid < kTileDocs) {
            const long long doc = doc_base + tid;
            dsc[tid] = doc < a.n_docs ? a.dscores_in[doc] : 0.f;
        }
        LTR_STAMP(1)
        // ---- fc1 (its last epoch prefetches fc2's first slot)
        f32x4 h1[N::H1T];
        if (N::H1T > N::NT1) h1[N::H1T - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_ring<PhaseFwd, 0, N::NT1, N::XK>(lds, packed, xb, h1, w, lane, a.debug_skip, [&](int k) { LTR_STAMP(k) });
        LTR_STAMP(15)
        activate<N::A1, N::H1, N::NT1>(h1, q, a, 0, a.keep1, gdoc);
        {   // ones feature at index H1 (carries b2 through fc2 and db2 through dW2)
            constexpr int Tn = N::H1 / 16, p = N::H1 % 16;
            h1[Tn][p % 4] = (q == p / 4) ? 1.f : h1[Tn][p % 4];
        }
        // h1 > 0 (ReLU / dropout survivor) is all the layer-1 backward needs besides the pieces below
        unsigned hbits[(N::NT1 * 4 + 31) / 32] = {};
        if (MODE != MODE_FWD && N::A1 == ACT_RELU_DROP) {
#pragma unroll
            for (int To = 0; To < N::NT1; ++To)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    hbits[(To * 4 + r) >> 5] |= (h1[To][r] > 0.f ? 1u : 0u) << ((To * 4 + r) & 31);
        }
        u32x4 hb[N::KMAX][3];                       // fc2's B operand = dW2's B operand
        tiles_to_operand<N::H1T, N::H1K, 3>(h1, hb);
        LTR_STAMP(2)
        // ---- fc2
        f32x4 h2[N::NT2];
        gemm_ring<PhaseFwd, N::NT1, N::NT2, N::H1K>(lds, packed, hb, h2, w, lane, a.debug_skip);
        if (MODE != MODE_FWD) {
            // park [h1 | 1] for dW2: the ring is idle until dh1, and the 60 registers are free for the loss phase
            __syncthreads();          // every wave is past its last ring read
            operand_to_images<N::H1K, N::H1T, LD>(hb, park, PP, my_row, q);
        }
        activate<N::A2, N::H2, N::NT2>(h2, q, a, 1, a.keep2, gdoc);
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
        if (MODE == MODE_FUSED) {
            __syncthreads();
            const int group = 4 * a.S;                 // S in {32, 64, 128}: 4 threads per document row
            const int gid = tid / group;
            const SlateGroup g = make_group(a.S, group, scratch + gid * (group + 32), tid);
            const int so = gid * a.S;
            const long long slate = (long long)st * (kTileDocs / a.S) + gid;
            float loss;
            if (LOSS == 0) {
                auto st_ds = [&](int i, float v) { dsc[so + i] = v; };
                if (a.S == 128)
                    loss = approx_ndcg_slate<32>(g, sc + so, yl + so, gn + so, gg + so, uu + so, mk + so, a.alpha,
                                                 a.eps, a.gscale, true, st_ds);
                else if (a.S == 64)
                    loss = approx_ndcg_slate<16>(g, sc + so, yl + so, gn + so, gg + so, uu + so, mk + so, a.alpha,
                                                 a.eps, a.gscale, true, st_ds);
                else
                    loss = approx_ndcg_slate<8>(g, sc + so, yl + so, gn + so, gg + so, uu + so, mk + so, a.alpha,
                                                a.eps, a.gscale, true, st_ds);
            } else if (LOSS == 1)
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
        __syncthreads();      // dsc published; every wave is past its last ring read (the staging region is free)

        LTR_STAMP(5)
        // ---- pull the NEXT super-tile of X into L2 while this one is in its backward (one dword per 128-B line)
        float pf = 0.f;
        {
            const long long nb = (long long)(st + gridDim.x) * kTileDocs;
            const long long fl = nb * N::F + (long long)tid * 32;
            if (st + (int)gridDim.x < a.n_super && fl < a.n_docs * N::F) pf = a.X[fl];
            constexpr int LINES = (kTileDocs * N::F + 31) / 32;
            if (tid + kThreads < LINES) {
                const long long fl2 = fl + (long long)kThreads * 32;
                if (st + (int)gridDim.x < a.n_super && fl2 < a.n_docs * N::F) pf += a.X[fl2];
            }
        }
        // ---- backward through fc3: dw3 += ds * h2, dz2 = ds * w3 * act2'(h2); padding documents contribute nothing
        const float ds0 = gdoc < a.n_docs ? dsc[my_row] : 0.f;
        db3 += wave_allsum((q == 0) ? ds0 : 0.f);
        const float slope = a.dropout ? 2.f : 1.f;
        const float ds2 = (N::A2 == ACT_RELU_DROP) ? ds0 * slope : ds0;
#pragma unroll
        for (int To = 0; To < N::NT2; ++To) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w3s + 16 * To + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = row_sum_to_lane15(ds0 * h2[To][r]);
                if (d == 15)
                    __hip_atomic_fetch_add(&dw3[w * N::NT2 * 16 + 16 * To + 4 * q + r], v, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                h2[To][r] = apply_act_grad<N::A2>(ds2 * wv[r], h2[To][r]);
            }
        }
        // h2 now holds dz2 -> dh1's B operand (3 pieces) = dW2's A operand
        u32x4 zb[N::KMAX][3];
        tiles_to_operand<N::NT2, N::H2K, 3>(h2, zb);
        LTR_STAMP(6)
        // ---- dW2 += dz2^T [h1 | 1]: four k steps of 32 documents; the A operand (dz2 of waves 2c, 2c+1) is staged in
        //      the ring region, the B operand is the parked h1 images
#ifndef LTR_EXP
#define LTR_EXP 0
#endif
#pragma unroll
        for (int c = 0; c < ((LTR_EXP & 2) ? 0 : kTileDocs / 32); ++c) {
            if (c > 0) __syncthreads();   // the previous k step's A images are consumed
            if ((w >> 1) == c) operand_to_images<N::H2K, N::NT2, LD>(zb, stage, SP, 16 * (w & 1) + d, q);
            __syncthreads();
            dw_kstep_bf16<N::TW2, N::NT2, N::H1T, N::BH2, LD>(w, accW2, stage, SP, 0, park, PP, 32 * c, lane);
        }
        LTR_STAMP(7)
        // ---- dh1^T = W2^T dz2^T through the ring (the staging region becomes the ring again), then dz1 = dh1 * act1'(h1)
        __syncthreads();              // every wave is done with the A images and with parked h1: the region is the ring again
        ring_start<PhaseBwd>(lds, packed, w, lane);
        f32x4 dz1[N::NT1];
        gemm_ring<PhaseBwd, 0, N::NT1, N::H2K>(lds, packed, zb, dz1, w, lane, a.debug_skip);
#pragma unroll
        for (int To = 0; To < N::NT1; ++To)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dz1[To][r];
                if (N::A1 == ACT_RELU_DROP) v = ((hbits[(To * 4 + r) >> 5] >> ((To * 4 + r) & 31)) & 1u) ? v * slope : 0.f;
                if (16 * To + 16 > N::H1) v = (16 * To + 4 * q + r < N::H1) ? v : 0.f;   // last tile only
                dz1[To][r] = v;
            }
        LTR_STAMP(8)
        // ---- dW1 += dz1^T [x | 1]: x is loaded a second time (the lines are in L2 / the Infinity Cache since the top
        //      of the tile), split and parked where h1 was (every wave is past dW2: dh1's barriers lie in between);
        //      dz1 is staged 32 documents at a time like dz2
        {
            __syncthreads();              // every wave is past its last ring read: the park images may be rewritten
            {
                const bool in_range = gdoc < a.n_docs;
                const float *xrow = a.X + gdoc * N::F;
                f32x4 xv[N::XK][2];
#pragma unroll
                for (int T = 0; T < N::XK; ++T) {
                    const int f0 = 32 * T + 8 * q;
                    xv[T][0] = xv[T][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (in_range && f0 + 8 <= N::F) {
                        xv[T][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xrow + f0));
                        xv[T][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xrow + f0 + 4));
                    }
                    if (f0 == N::F) xv[T][0][0] = 1.f;
                }
#pragma unroll
                for (int T = 0; T < N::XK; ++T) {
                    const int f0 = 32 * T + 8 * q;
                    if (f0 < LD) {
                        unsigned p0[3], p1[3], p2[3], p3[3];
                        split2<3>(xv[T][0][0], xv[T][0][1], p0);
                        split2<3>(xv[T][0][2], xv[T][0][3], p1);
                        split2<3>(xv[T][1][0], xv[T][1][1], p2);
                        split2<3>(xv[T][1][2], xv[T][1][3], p3);
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            *reinterpret_cast<u32x4 *>(park + p * PP + my_row * LD + f0) = u32x4{p0[p], p1[p], p2[p], p3[p]};
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < ((LTR_EXP & 1) ? 0 : kTileDocs / 32); ++c) {
                if (c > 0) __syncthreads();   // previous A images consumed
                if ((w >> 1) == c) {
                    u32x4 z1[(N::NT1 + 1) / 2][3];
                    tiles_to_operand<N::NT1, (N::NT1 + 1) / 2, 3>(dz1, z1);
                    operand_to_images<(N::NT1 + 1) / 2, N::NT1, LD>(z1, stage, SP, 16 * (w & 1) + d, q);
                }
                __syncthreads();
                dw_kstep_bf16<N::TW1, N::NT1, N::XT, N::BH1, LD>(w, accW1, stage, SP, 0, park, PP, 32 * c, lane);
            }
        }
        LTR_STAMP(9)
        asm volatile("" ::"v"(pf));   // keep the prefetch load alive (and waited for) until here
    }

    if (MODE == MODE_FWD) return;
    // ---- per-workgroup partial gradients -> workspace
    float *part = a.partials + (size_t)blockIdx.x * N::PART;
    dw_store<N::TW1, N::NT1, N::XT, N::BH1>(w, accW1, part + N::P_W1, q, d);
    dw_store<N::TW2, N::NT2, N::H1T, N::BH2>(w, accW2, part + N::P_W2, q, d);
    __syncthreads();
    for (int j = tid; j < N::NT2 * 16; j += kThreads) {
        float s = 0.f;
        for (int ww = 0; ww < kWaves; ++ww) s += dw3[ww * N::NT2 * 16 + j];
        part[N::P_W3 + j] = s;
    }
    if (lane == 0) scratch[w] = db3;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int ww = 0; ww < kWaves; ++ww) s += scratch[ww];
        part[N::P_B3] = s;
    }
}

// Pack nn.Linear parameters into bf16 x 3 A-fragments (once per optimizer step): fragment [out tile To][k tile T][piece p],
// lane (n = lane & 15, q = lane >> 4), slot j = the k-th contraction feature of that lane group.
//   fc1 : natural order,  feature 32 T + 8 q + j                          (x is loaded in that order)
//   fc2 / dh1 : feature 32 T + 16 (j >> 2) + 4 q + (j & 3)                (= accumulator tiles 2T, 2T+1 as they stand)
template <class N>
__global__ void pack_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                            const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
                            float *__restrict__ packed_f) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    unsigned short *pk = reinterpret_cast<unsigned short *>(packed_f);
    auto w1aug = [&](int n, int f) { return n < N::H1 ? (f < N::F ? W1[n * N::F + f] : (f == N::F ? b1[n] : 0.f)) : 0.f; };
    auto w2aug = [&](int n, int f) { return n < N::H2 ? (f < N::H1 ? W2[n * N::H1 + f] : (f == N::H1 ? b2[n] : 0.f)) : 0.f; };
    auto put3 = [&](int byte_off, int tile, int lane, int j, float v) {   // the three pieces of v -> fragments tile*3 + p
        float r = v;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const __bf16 b = (__bf16)r;
            pk[byte_off / 2 + ((tile * 3 + p) * 64 + lane) * 8 + j] = __builtin_bit_cast(unsigned short, b);
            r -= (float)b;
        }
    };
    for (int e = gt; e < N::NT1 * N::XK * 512; e += stride) {
        const int j = e & 7, lane = (e >> 3) & 63, tile = e >> 9, To = tile / N::XK, T = tile - To * N::XK;
        put3(N::W1F_OFF, tile, lane, j, w1aug(16 * To + (lane & 15), 32 * T + 8 * (lane >> 4) + j));
    }
    for (int e = gt; e < N::NT2 * N::H1K * 512; e += stride) {
        const int j = e & 7, lane = (e >> 3) & 63, tile = e >> 9, To = tile / N::H1K, T = tile - To * N::H1K;
        put3(N::W2F_OFF, tile, lane, j, w2aug(16 * To + (lane & 15), 32 * T + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)));
    }
    for (int e = gt; e < N::NT1 * N::H2K * 512; e += stride) {
        const int j = e & 7, lane = (e >> 3) & 63, tile = e >> 9, Ti = tile / N::H2K, T = tile - Ti * N::H2K;
        const int in = 16 * Ti + (lane & 15), o = 32 * T + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        put3(N::W2T_OFF, tile, lane, j, (o < N::H2 && in < N::H1) ? W2[o * N::H1 + in] : 0.f);
    }
    float *w3p = reinterpret_cast<float *>(reinterpret_cast<char *>(packed_f) + N::W3_OFF);
    for (int e = gt; e < N::NT2 * 16 + 16; e += stride) w3p[e] = e < N::H2 ? w3[e] : (e == N::NT2 * 16 ? b3[0] : 0.f);
}
