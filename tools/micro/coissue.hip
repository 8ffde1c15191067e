// Micro-benchmark (gfx950): what the two waves of one SIMD cost each other, by instruction class.
// One 512-thread workgroup per CU: waves 0-3 land on SIMDs 0..3, waves 4-7 are their co-residents.  Each wave runs ONE
// role for a fixed number of instructions and reports its own elapsed s_memtime cycles; roles alone vs paired show how
// fp32 MFMA (v_mfma_f32_16x16x4_f32), f16 MFMA (v_mfma_f32_16x16x32_f16), plain fp32 VALU and v_rcp_f32 share a SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/coissue.hip -o variants/coissue && variants/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define R8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
enum { IDLE = 0, MFMA32 = 1, MFMA16 = 2, VALU = 3, RCP = 4, MIX = 5, LDSB = 6 };

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ void __launch_bounds__(512, 2) k(int role_a, int role_b, int n, unsigned long long *cyc, float *sink) {
    __shared__ float lds[1024];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = w < 4 ? role_a : role_b;
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = 1.f + i * 1e-3f;
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float a = 1.f + lane * 1e-3f, b = 0.999f, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    h16x8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(0.01f * (lane + j)); hb[j] = (_Float16)(0.02f * (lane - j)); }
    const unsigned long long t0 = now();
    // every role: n / 8 trips of a body repeated 8 times by the preprocessor (loop overhead amortised)
    if (role == MFMA32) {
        for (int i = 0; i < n; i += 8) { R8({
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }) }
    } else if (role == MFMA16) {
        for (int i = 0; i < n; i += 8) { R8({
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, c1, 0, 0, 0);
        }) }
    } else if (role == VALU) {          // 4 independent fma chains
        for (int i = 0; i < n; i += 8) { R8({
            x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }) }
    } else if (role == RCP) {
        for (int i = 0; i < n; i += 8) { R8({
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }) }
    } else if (role == MIX) {           // the pair-sweep mix: 15 plain + 1 rcp per "quad"
        for (int i = 0; i < n; i += 8) { R8({
            const float d0 = a + x0, d1 = a + x1, d2 = a + x2, d3 = a + x3;
            const float D01 = d0 * d1, D23 = d2 * d3;
            const float N01 = fmaf(x0, d1, x1 * d0), N23 = fmaf(x2, d3, x3 * d2);
            x0 = fmaf(fmaf(N01, D23, N23 * D01), __builtin_amdgcn_rcpf(D01 * D23), x0);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }) }
    } else if (role == LDSB) {          // broadcast ds_read_b128 + 4 fma
        for (int i = 0; i < n; i += 8) { R8({
            const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + ((i * 4) & 1020));
            x0 = fmaf(x0, v[0], a); x1 = fmaf(x1, v[1], a); x2 = fmaf(x2, v[2], a); x3 = fmaf(x3, v[3], a);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }) }
    }
    const unsigned long long t1 = now();
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + x0 + x1 + x2 + x3;
}

int main() {
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    const char *nm[] = {"idle", "mfma_f32_16x16x4", "mfma_f16_16x16x32", "valu_fma", "rcp", "sweep_mix(15+rcp)", "lds_b128+4fma"};
    const int per[] = {0, 4, 4, 4, 4, 16, 5};   // instructions per loop trip
    const int n = 4096;
    auto run = [&](int ra, int rb) {
        std::vector<unsigned long long> h(256 * 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, ra, rb, n, cyc, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> A, B;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? A : B).push_back((double)h[b * 8 + w]);
        std::sort(A.begin(), A.end()); std::sort(B.begin(), B.end());
        const double ca = A[A.size() / 2] / n, cb = B[B.size() / 2] / n;
        printf("{\"a\": \"%s\", \"b\": \"%s\", \"cyc_per_instr_a\": %.2f, \"cyc_per_instr_b\": %.2f}\n", nm[ra], nm[rb],
               per[ra] ? ca / per[ra] : 0.0, per[rb] ? cb / per[rb] : 0.0);
    };
    const int roles[] = {MFMA32, MFMA16, VALU, RCP, MIX, LDSB};
    for (int r : roles) run(r, IDLE);
    for (int r : roles) run(r, r);
    for (int r : {VALU, RCP, MIX, LDSB}) { run(MFMA32, r); run(MFMA16, r); run(r, MFMA32); run(r, MFMA16); }
    return 0;
}
