// mc_synth.hip -- libmcsynth.so: deterministic synthetic cuCLARK databases generated
// directly in HBM (bench.py / tests tooling; NOT part of the drop-in C ABI).
//
// A RefSeq-bacteria-scale table (HTSIZE = 1610612741 buckets, several 10^9 k-mers,
// SURVEY.md section 8d config 3) cannot be shipped or read from disk inside a bench
// run, so it is generated on the device in the exact array form of the on-disk format
// (.sz = sizes u8, .ky = quotients u32 ascending per bucket, .lb = labels u16;
// reference src/hashTable_hh.hh:473-546) and then handed to mc_load_db_device(), i.e.
// through the same re-layout the file loader uses.
//
// Every bucket is a pure function of (seed, bucket index): the host twin
// mcs_bucket_host() regenerates any bucket on the CPU, which is how tests check the
// generator and how a sample of the full-size table can be cross-checked without
// copying it.  Genome-derived k-mers (so that reads sampled from the genomes hit
// runs of overlapping k-mers, as real reads do) are appended with mcs_append_*.
#include "mc_device.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return -1; }
#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct GenParams {
    uint64_t seed;
    uint64_t htsize;
    uint64_t qmax;        // quotients are drawn from [0, qmax)
    uint32_t n_targets;
    uint32_t n_cdf;
    uint32_t cdf[64];     // Poisson CDF scaled to 2^24
};

__host__ __device__ inline uint64_t sm64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t h3(uint64_t seed, uint64_t a, uint64_t b)
{
    return sm64(sm64(seed ^ (a * 0xD6E8FEB86659FD93ull)) + b);
}

__host__ __device__ inline uint32_t bg_count(const GenParams &g, uint64_t b)
{
    const uint32_t u = (uint32_t)(h3(g.seed, b, 0xC0DEull) >> 40);     // 24 bits
    uint32_t c = 0;
    while (c < g.n_cdf && u >= g.cdf[c]) c++;
    return c;
}
// j-th (ascending, distinct) quotient of bucket b holding c background k-mers
__host__ __device__ inline uint32_t bg_key(const GenParams &g, uint64_t b, uint32_t j, uint32_t c)
{
    const uint64_t slot = g.qmax / c;
    const uint64_t span = slot - slot / 10;
    return (uint32_t)((uint64_t)j * slot + h3(g.seed, b, 1 + j) % span);
}
__host__ __device__ inline uint16_t bg_label(const GenParams &g, uint64_t b, uint32_t j)
{
    return (uint16_t)(h3(g.seed, b, 0x10000ull + j) % g.n_targets);
}

int make_params(GenParams &g, uint64_t seed, uint32_t k, uint64_t htsize, uint32_t n_targets, double lambda)
{
    if (k < 2 || k > 32 || htsize < 2 || n_targets < 1 || n_targets > 65535 || lambda < 0 || lambda > 24)
        return fail("mcs: bad parameters");
    const unsigned __int128 maxkmer = k == 32 ? (unsigned __int128)~0ull : (((unsigned __int128)1 << (2 * k)) - 1);
    const unsigned __int128 qm = maxkmer / htsize;
    if (qm >= 0xFFFFFFFFull || qm < 4096) return fail("mcs: k/htsize outside the 4-byte-key regime");
    g.seed = seed; g.htsize = htsize; g.qmax = (uint64_t)qm; g.n_targets = n_targets;
    double p = std::exp(-lambda), cum = 0;
    g.n_cdf = 0;
    for (int i = 0; i < 64; i++) {
        cum += p;
        const double s = cum * 16777216.0;
        g.cdf[i] = s >= 16777216.0 ? 16777216u : (uint32_t)s;
        g.n_cdf = i + 1;
        if (g.cdf[i] >= 16777216u - 1) { g.cdf[i] = 16777216u; break; }
        p *= lambda / (i + 1);
    }
    g.cdf[g.n_cdf - 1] = 16777216u;   // u < 2^24 always stops here: count <= n_cdf-1 <= 63
    return 0;
}

// ---- kernels ---------------------------------------------------------------------
__global__ void counts_kernel(const GenParams g, uint64_t b0, uint64_t nb, uint8_t *sz)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride)
        sz[i] = (uint8_t)bg_count(g, b0 + i);
}

// extra[b] += 1 for every appended k-mer (bytes, via 32-bit atomics on the containing word)
__global__ void add_counts_kernel(const int64_t *r, uint64_t n, uint64_t b0, uint8_t *sz)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = (uint64_t)r[i] - b0;
        atomicAdd(reinterpret_cast<unsigned int *>(sz + (b & ~3ull)), 1u << (8 * (b & 3)));
    }
}

// exclusive offsets of every bucket (u64), two-level: per-workgroup base + in-group scan
__global__ __launch_bounds__(mc::RL_THREADS)
void offsets_kernel(const uint8_t *sz, uint64_t nb, const uint64_t *blk_off, uint64_t *off)
{
    __shared__ uint32_t s_a[mc::RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * mc::RL_BUCKETS + (uint64_t)threadIdx.x * mc::RL_PER_THREAD;
    uint32_t c[mc::RL_PER_THREAD], sum = 0;
    for (int i = 0; i < mc::RL_PER_THREAD; i++) { c[i] = (b0 + i < nb) ? sz[b0 + i] : 0u; sum += c[i]; }
    uint32_t tot;
    uint64_t o = blk_off[blockIdx.x] + mc::block_exclusive_scan(sum, s_a, tot);
    for (int i = 0; i < mc::RL_PER_THREAD; i++) { if (b0 + i < nb) off[b0 + i] = o; o += c[i]; }
}

// background k-mers of every bucket, written at the head of the bucket's run
__global__ void fill_bg_kernel(const GenParams g, uint64_t b0, uint64_t nb, const uint64_t *off,
                               uint32_t *keys, uint16_t *labels)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) {
        const uint32_t c = bg_count(g, b0 + i);
        const uint64_t o = off[i];
        for (uint32_t j = 0; j < c; j++) {
            keys[o + j] = bg_key(g, b0 + i, j, c);
            labels[o + j] = bg_label(g, b0 + i, j);
        }
    }
}

// appended k-mers go behind the background ones; cursor[] counts them per bucket
__global__ void append_kernel(const GenParams g, const int64_t *r, const int64_t *q, const int16_t *lab,
                              uint64_t n, uint64_t b0, const uint64_t *off, uint8_t *cursor,
                              uint32_t *keys, uint16_t *labels)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = (uint64_t)r[i] - b0;
        const uint32_t sh = 8 * (b & 3);
        const uint32_t old = atomicAdd(reinterpret_cast<unsigned int *>(cursor + (b & ~3ull)), 1u << sh);
        const uint32_t slot = (old >> sh) & 0xFFu;
        const uint64_t p = off[b] + bg_count(g, b0 + b) + slot;
        keys[p] = (uint32_t)q[i];
        labels[p] = (uint16_t)lab[i];
    }
}

// restore ascending order in the buckets that received appended k-mers
__global__ void sort_buckets_kernel(const uint8_t *sz, const uint8_t *cursor, uint64_t nb,
                                    const uint64_t *off, uint32_t *keys, uint16_t *labels)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) {
        if (cursor[i] == 0) continue;
        const uint32_t c = sz[i];
        uint32_t *k = keys + off[i];
        uint16_t *l = labels + off[i];
        for (uint32_t a = 1; a < c; a++) {
            const uint32_t kv = k[a]; const uint16_t lv = l[a];
            uint32_t p = a;
            while (p > 0 && k[p - 1] > kv) { k[p] = k[p - 1]; l[p] = l[p - 1]; p--; }
            k[p] = kv; l[p] = lv;
        }
    }
}

__global__ void max_byte_kernel(const uint8_t *sz, uint64_t nb, unsigned int *mx)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned int m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) m = sz[i] > m ? sz[i] : m;
    atomicMax(mx, m);
}

int grid_for(uint64_t n)
{
    const uint64_t want = (n + 255) / 256;
    return (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

} // namespace

extern "C" {

const char *mcs_last_error(void) { return g_err.c_str(); }

// Phase 1: bucket sizes of buckets [b0, b0+nb) -> d_sz; appended k-mers (bucket ids
// r[], may be NULL) are counted in as well.  *n_keys = total number of k-mers.
int mcs_counts_device(uint64_t seed, uint32_t k, uint64_t htsize, uint32_t n_targets, double lambda,
                      uint64_t b0, uint64_t nb, const int64_t *d_app_r, uint64_t n_app,
                      uint8_t *d_sz, uint64_t *n_keys, void *stream)
{
    GenParams g;
    if (make_params(g, seed, k, htsize, n_targets, lambda)) return -1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(counts_kernel, dim3(grid_for(nb)), dim3(256), 0, st, g, b0, nb, d_sz);
    HIPCHK(hipGetLastError());
    if (n_app) {
        hipLaunchKernelGGL(add_counts_kernel, dim3(grid_for(n_app)), dim3(256), 0, st, d_app_r, n_app, b0, d_sz);
        HIPCHK(hipGetLastError());
    }
    // total via the per-workgroup sums the loader also uses
    const uint32_t nblk = (uint32_t)((nb + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    uint32_t *d_bk = nullptr, *d_bo = nullptr;
    HIPCHK(hipMalloc(&d_bk, (size_t)nblk * 4));
    HIPCHK(hipMalloc(&d_bo, (size_t)nblk * 4));
    hipLaunchKernelGGL(mc::block_sums_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz, nb, 255u, d_bk, d_bo);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> bk(nblk);
    HIPCHK(hipMemcpyAsync(bk.data(), d_bk, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d_bk); (void)hipFree(d_bo);
    uint64_t tot = 0;
    for (uint32_t v : bk) tot += v;
    *n_keys = tot;
    return 0;
}

// Phase 2: fill d_keys / d_labels (n_keys elements as counted by phase 1).
// d_off (nb u64) and d_cursor (nb bytes, rounded up to 4) are caller-provided scratch.
int mcs_fill_device(uint64_t seed, uint32_t k, uint64_t htsize, uint32_t n_targets, double lambda,
                    uint64_t b0, uint64_t nb, const uint8_t *d_sz,
                    const int64_t *d_app_r, const int64_t *d_app_q, const int16_t *d_app_lab, uint64_t n_app,
                    uint64_t *d_off, uint8_t *d_cursor, uint32_t *d_keys, uint16_t *d_labels, void *stream)
{
    GenParams g;
    if (make_params(g, seed, k, htsize, n_targets, lambda)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nblk = (uint32_t)((nb + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    uint32_t *d_bk = nullptr, *d_bo = nullptr;
    uint64_t *d_boff = nullptr;
    HIPCHK(hipMalloc(&d_bk, (size_t)nblk * 4));
    HIPCHK(hipMalloc(&d_bo, (size_t)nblk * 4));
    HIPCHK(hipMalloc(&d_boff, (size_t)nblk * 8));
    hipLaunchKernelGGL(mc::block_sums_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz, nb, 255u, d_bk, d_bo);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> bk(nblk);
    HIPCHK(hipMemcpyAsync(bk.data(), d_bk, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint64_t> boff(nblk);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < nblk; i++) { boff[i] = acc; acc += bk[i]; }
    HIPCHK(hipMemcpyAsync(d_boff, boff.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(offsets_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz, nb, d_boff, d_off);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(fill_bg_kernel, dim3(grid_for(nb)), dim3(256), 0, st, g, b0, nb, d_off, d_keys, d_labels);
    HIPCHK(hipGetLastError());
    if (n_app) {
        unsigned int *d_mx = nullptr, mx = 0;
        HIPCHK(hipMalloc(&d_mx, 4));
        HIPCHK(hipMemsetAsync(d_mx, 0, 4, st));
        hipLaunchKernelGGL(max_byte_kernel, dim3(grid_for(nb)), dim3(256), 0, st, d_sz, nb, d_mx);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&mx, d_mx, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(d_mx);
        if (mx >= 250) return fail("mcs: a bucket grew to >= 250 k-mers; lower lambda or the appended set");
        HIPCHK(hipMemsetAsync(d_cursor, 0, (nb + 3) & ~3ull, st));
        hipLaunchKernelGGL(append_kernel, dim3(grid_for(n_app)), dim3(256), 0, st, g, d_app_r, d_app_q, d_app_lab,
                           n_app, b0, d_off, d_cursor, d_keys, d_labels);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(sort_buckets_kernel, dim3(grid_for(nb)), dim3(256), 0, st, d_sz, d_cursor, nb, d_off, d_keys, d_labels);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d_bk); (void)hipFree(d_bo); (void)hipFree(d_boff);
    return 0;
}

// CPU twin: background content of one bucket (no appended k-mers).
int mcs_bucket_host(uint64_t seed, uint32_t k, uint64_t htsize, uint32_t n_targets, double lambda,
                    uint64_t bucket, uint32_t *count, uint32_t *keys, uint16_t *labels)
{
    GenParams g;
    if (make_params(g, seed, k, htsize, n_targets, lambda)) return -1;
    const uint32_t c = bg_count(g, bucket);
    *count = c;
    for (uint32_t j = 0; j < c; j++) { keys[j] = bg_key(g, bucket, j, c); labels[j] = bg_label(g, bucket, j); }
    return 0;
}

} // extern "C"
