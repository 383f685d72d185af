// Random 128-byte line gather, nothing else: what the memory system delivers to the access pattern of the query kernels
// (a line index from a hash, LPL lanes fetch one aligned 128-byte line with 16-byte loads, ROUNDS wave-instructions in
// flight per wave before any is used, nontemporal loads, a persistent grid of OCC workgroups of 4 waves per CU).
// The ceiling the kernels' HBM figures are read against.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gather128 tools/ubench/gather128.hip
// Run:   tools/ubench/gather128 [table GB = 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// LPL lanes per line: 8 (16 bytes each, one load: the minimizer kernel) or 4 (two loads of 16 bytes, 64 bytes apart: the super-k-mer kernel)
template <int LPL, int ROUNDS>
__global__ __launch_bounds__(256) void gather(const uint8_t *table, uint32_t n_lines, uint32_t iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    constexpr uint32_t per_round = 64 / LPL;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        u32x4 v[ROUNDS], w[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const uint32_t id = ((wave * iters + it) * ROUNDS + rd) * per_round + lane / LPL;
            const uint32_t ln = (uint32_t)(((uint64_t)mix(id) * n_lines) >> 32);
            const uint8_t *p = table + ((uint64_t)ln << 7) + (lane % LPL) * 16u;
            v[rd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
            if (LPL == 4) w[rd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + 64));
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            acc ^= v[rd][0] ^ v[rd][3];
            if (LPL == 4) acc ^= w[rd][1];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// The same with lines of LINE bytes (32 / 64 / 128, aligned to their size, LINE / 16 lanes a line, one 16-byte load each): does the
// card deliver random lines by the byte or by the request?  (What a 64-byte line would buy an index: DESIGN.md 4.)
template <int LINE, int ROUNDS>
__global__ __launch_bounds__(256) void gather_line(const uint8_t *table, uint64_t n_lines, uint32_t iters, uint32_t *out)
{
    constexpr uint32_t LPL = LINE / 16, per_round = 64 / LPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        u32x4 v[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) {
            const uint32_t id = ((wave * iters + it) * ROUNDS + rd) * per_round + lane / LPL;
            const uint64_t ln = ((uint64_t)mix(id) * n_lines) >> 32;                 // (a 32-bit hash times at most 2^33 lines)
            const uint8_t *p = table + ln * LINE + (lane % LPL) * 16u;
            v[rd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; rd++) acc ^= v[rd][0] ^ v[rd][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LINE, int ROUNDS>
static void run_line(const uint8_t *table, uint64_t bytes, int occ, uint32_t *out)
{
    const uint64_t n_lines = bytes / LINE;
    const int blocks = 256 * occ;
    const uint64_t want = 400ull * 1000 * 1000;
    const uint64_t per_iter = (uint64_t)blocks * 4 * ROUNDS * (64 / (LINE / 16));
    const uint32_t iters = (uint32_t)(want / per_iter);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gather_line<LINE, ROUNDS><<<blocks, 256>>>(table, n_lines, iters / 8 + 1, out);
    hipEventRecord(e0);
    gather_line<LINE, ROUNDS><<<blocks, 256>>>(table, n_lines, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lines = (double)per_iter * iters;
    printf("table %6.1f GB  %3d-byte lines  %d rounds (%3d lines) in flight/wave  %d WG/CU: %6.2f G lines/s = %5.2f TB/s of lines\n", bytes / 1e9, LINE, ROUNDS,
           ROUNDS * 64 / (LINE / 16), occ, lines / ms / 1e6, lines * LINE / ms / 1e9);
    fflush(stdout);
}

template <int LPL, int ROUNDS>
static void run(const uint8_t *table, uint64_t bytes, int occ, uint32_t *out)
{
    const uint32_t n_lines = (uint32_t)(bytes >> 7);
    const int blocks = 256 * occ;
    const uint64_t want = 400ull * 1000 * 1000;                         // lines per launch
    const uint64_t per_iter = (uint64_t)blocks * 4 * ROUNDS * (64 / LPL);
    const uint32_t iters = (uint32_t)(want / per_iter);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gather<LPL, ROUNDS><<<blocks, 256>>>(table, n_lines, iters / 8 + 1, out);
    hipEventRecord(e0);
    gather<LPL, ROUNDS><<<blocks, 256>>>(table, n_lines, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lines = (double)per_iter * iters;
    printf("table %6.1f GB  %d lanes/line  %d rounds (%3d lines) in flight/wave  %d WG/CU: %6.2f G lines/s = %5.2f TB/s\n", bytes / 1e9, LPL, ROUNDS,
           ROUNDS * 64 / LPL, occ, lines / ms / 1e6, lines * 128.0 / ms / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 200.0;
    const uint64_t bytes = (uint64_t)(gb * 1e9) & ~127ull;
    uint8_t *table; uint32_t *out;
    if (hipMalloc(&table, bytes) != hipSuccess) { fprintf(stderr, "no room for %.0f GB\n", gb); return 1; }
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemset(table, 1, bytes);
    hipDeviceSynchronize();
    const uint64_t sizes[] = {2ull << 20, 128ull << 20, 8ull << 30, bytes};
    for (uint64_t sz : sizes) {
        if (sz > bytes) continue;
        run<8, 4>(table, sz, 7, out);
        run<4, 2>(table, sz, 7, out);
    }
    for (int occ : {2, 4, 5, 6, 8}) run<8, 4>(table, bytes, occ, out);
    run<8, 2>(table, bytes, 7, out);
    run<8, 8>(table, bytes, 7, out);
    run<4, 4>(table, bytes, 7, out);
    // line size
    run_line<128, 4>(table, bytes, 7, out);
    run_line<64, 4>(table, bytes, 7, out);
    run_line<64, 2>(table, bytes, 7, out);
    run_line<64, 8>(table, bytes, 7, out);
    run_line<32, 4>(table, bytes, 7, out);
    run_line<32, 2>(table, bytes, 7, out);
    run_line<64, 4>(table, 8ull << 30, 7, out);
    run_line<32, 4>(table, 8ull << 30, 7, out);
    return 0;
}
