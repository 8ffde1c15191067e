// Micro-benchmark: issue cost of v_mul_lo_u32 vs simple VALU ops on gfx950 (one wave per SIMD, 4 independent chains).
// Measured (round 2): v_mul_lo_u32 14.2 vs 8.1 "cycles" per instruction for a shift+xor pair member at the nominal 2.4 GHz,
// i.e. a 32-bit multiply costs ~1.8 simple ops -- not enough to justify a 24-bit-multiply dropout hash.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(unsigned *out, int n, unsigned c) {
    unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 7, d = threadIdx.x ^ 0x55, e = threadIdx.x + 99;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { a *= c; b *= c; d *= c; e *= c; }
        if (MODE == 1) { a = __umul24(a, c) ; b = __umul24(b, c); d = __umul24(d, c); e = __umul24(e, c); }
        if (MODE == 2) { a ^= a >> 7; b ^= b >> 7; d ^= d >> 7; e ^= e >> 7; }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + d + e;
}
template <int MODE> float run(unsigned *out, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, 16, 0x9E3779B1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, n, 0x9E3779B1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    unsigned *out; hipMalloc(&out, 256 * 256 * 4);
    const int n = 1 << 20;
    const char *names[3] = {"v_mul_lo_u32", "v_mul_u32_u24", "xor-shift (2 ops)"};
    float t[3] = {run<0>(out, n), run<1>(out, n), run<2>(out, n)};
    // one wave per SIMD (256 threads = 4 waves on a CU): cycles per instruction = time * clock / (n * 4 ops)
    for (int m = 0; m < 3; ++m) printf("%-20s %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz (4 independent chains)\n", names[m], t[m], t[m] * 1e-3 * 2.4e9 / (n * 4.0 * (m == 2 ? 2 : 1)));
    return 0;
}
