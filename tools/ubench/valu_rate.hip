// VALU issue-rate probe: time N independent ops per wave for a few opcodes (MI355X: v_mul_lo_u32,
// v_mul_hi_u32, v_min_f64 and v_lshrrev_b64 issue at the full rate; a u64 multiply is 4 ops, a u64 min 3).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8]; uint64_t b[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 3u); d[i] = (double)a[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + seed ^ 0x55u;                       // add + xor: 2 full-rate ops (v_add + v_xor) or 1 fused
            if (OP == 1) a[i] = a[i] * 0x85EBCA6Bu;                        // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(a[i], seed | 1u);                 // v_mul_hi_u32
            if (OP == 3) b[i] = b[i] * 0x9E3779B97F4A7C15ull;              // 64-bit multiply
            if (OP == 4) a[i] = __umul24(a[i], 0x5BD1E9u) ; // v_mul_u32_u24
            if (OP == 5) { double r; asm volatile("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(d[i]), "v"(d[(i + 1) & 7])); d[i] = r; }
            if (OP == 6) b[i] = b[i] >> (seed & 31);                       // v_lshrrev_b64
            if (OP == 7) b[i] = (b[i] < b[(i + 1) & 7]) ? b[i] : b[(i + 1) & 7]; // u64 min: cmp + 2 cndmask
            if (OP == 8) a[i] = __builtin_bitreverse32(a[i]) ^ seed;      // v_bfrev + xor
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char *name, uint32_t *out)
{
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 12345u, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // ops per SIMD: blocks*4 waves / 1024 SIMDs, each 8*iters source ops
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * 8.0 * iters);
    printf("%-28s %8.3f ms  %6.2f cycles per wave-op (2.4 GHz)\n", name, ms, cyc);
}
int main()
{
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("add+xor (2 ops)", out); run<1>("mul_lo_u32", out); run<2>("mul_hi_u32", out);
    run<3>("mul u64", out); run<4>("mul_u32_u24", out); run<5>("min_f64", out);
    run<6>("lshr_b64", out); run<7>("min_u64 (cmp+2sel)", out); run<8>("bfrev+xor (2 ops)", out);
    return 0;
}
