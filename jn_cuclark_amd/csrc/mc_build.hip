// mc_build.hip -- GPU build of the discriminative-k-mer table (include/mc_build.h).
//
// A counting sort by bucket on the device: count -> scan -> scatter -> per-bucket sort
// -> keep k-mers seen in exactly one target (reference RemoveCommon,
// src/HashTableStorage_hh.hh:229-280) -> compact -> .sz/.ky/.lb.  All integer work,
// HBM-bound streaming plus one random 4-byte atomic and one random 10-byte scatter per
// occurrence; nothing here is on the classification hot path.
#include "../../include/mc_build.h"
#include "mc_device.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" void mc_set_last_error_(const char *msg);     // mc_api.hip

namespace {

int bfail(const std::string &m) { mc_set_last_error_(m.c_str()); return -1; }
#define BCHK(expr)                                                                     \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return bfail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int SC_THREADS = 256, SC_PER = 4, SC_BUCKETS = SC_THREADS * SC_PER;

__device__ __forceinline__ void split_kmer(uint64_t x, uint32_t k, const mc::DivU64 &dv, uint64_t &q, uint64_t &r)
{
    const uint64_t rc = mc::revcomp(x, k);
    const uint64_t c = x < rc ? x : rc;                 // canonical (ref HashTableStorage_hh.hh:425-435)
    q = mc::div_u64(c, dv);
    r = c - q * dv.d;
}

__global__ void bld_count_kernel(const uint64_t *km, uint64_t n, uint32_t k, mc::DivU64 dv, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t q, r;
        split_kmer(km[i], k, dv, q, r);
        atomicAdd(&count[r], 1u);
    }
}

__global__ __launch_bounds__(SC_THREADS)
void bld_blocksum_kernel(const uint32_t *v, uint64_t nb, unsigned long long *blk)
{
    __shared__ unsigned long long s[SC_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * SC_BUCKETS + (uint64_t)threadIdx.x * SC_PER;
    unsigned long long sum = 0;
    for (int i = 0; i < SC_PER; i++) if (b0 + i < nb) sum += v[b0 + i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < SC_THREADS / 64; w++) t += s[w]; blk[blockIdx.x] = t; }
}

__global__ __launch_bounds__(SC_THREADS)
void bld_offsets_kernel(const uint32_t *v, uint64_t nb, const uint64_t *blk_off, uint64_t *off)
{
    __shared__ uint32_t s_a[SC_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * SC_BUCKETS + (uint64_t)threadIdx.x * SC_PER;
    uint32_t c[SC_PER], sum = 0;
    for (int i = 0; i < SC_PER; i++) { c[i] = (b0 + i < nb) ? v[b0 + i] : 0u; sum += c[i]; }
    uint32_t tot;
    uint64_t o = blk_off[blockIdx.x] + mc::block_exclusive_scan(sum, s_a, tot);
    for (int i = 0; i < SC_PER; i++) { if (b0 + i < nb) off[b0 + i] = o; o += c[i]; }
}

// count[] is used as a countdown: it ends at zero, the bucket is filled back to front
__global__ void bld_fill_kernel(const uint64_t *km, const uint16_t *tg, uint64_t n, uint32_t k, mc::DivU64 dv,
                                uint32_t *count, const uint64_t *off, uint64_t *qs, uint16_t *ts)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t q, r;
        split_kmer(km[i], k, dv, q, r);
        const uint32_t slot = atomicSub(&count[r], 1u) - 1u;
        const uint64_t p = off[r] + slot;
        qs[p] = q;
        ts[p] = tg[i];
    }
}

__device__ __forceinline__ bool pair_less(uint64_t qa, uint16_t ta, uint64_t qb, uint16_t tb)
{
    return qa < qb || (qa == qb && ta < tb);
}

// sort (q, t) of one bucket in place: insertion sort for the usual handful, heap sort
// for repeat-rich buckets
__device__ void sort_bucket(uint64_t *q, uint16_t *t, uint64_t n)
{
    if (n <= 24) {
        for (uint64_t a = 1; a < n; a++) {
            const uint64_t qv = q[a]; const uint16_t tv = t[a];
            uint64_t p = a;
            while (p > 0 && pair_less(qv, tv, q[p - 1], t[p - 1])) { q[p] = q[p - 1]; t[p] = t[p - 1]; p--; }
            q[p] = qv; t[p] = tv;
        }
        return;
    }
    auto sift = [&](uint64_t root, uint64_t end) {
        const uint64_t qv = q[root]; const uint16_t tv = t[root];
        for (;;) {
            uint64_t ch = 2 * root + 1;
            if (ch >= end) break;
            if (ch + 1 < end && pair_less(q[ch], t[ch], q[ch + 1], t[ch + 1])) ch++;
            if (!pair_less(qv, tv, q[ch], t[ch])) break;
            q[root] = q[ch]; t[root] = t[ch];
            root = ch;
        }
        q[root] = qv; t[root] = tv;
    };
    for (uint64_t s = n / 2; s-- > 0;) sift(s, n);
    for (uint64_t e = n - 1; e > 0; e--) {
        const uint64_t qv = q[0]; const uint16_t tv = t[0];
        q[0] = q[e]; t[0] = t[e]; q[e] = qv; t[e] = tv;
        sift(0, e);
    }
}

// per bucket: sort, then keep the runs of one k-mer that belong to ONE target and are
// longer than min_count (multiplicity 1, count > minCount); kept entries move to the
// front of the bucket.
__global__ void bld_sort_filter_kernel(uint64_t nb, const uint64_t *off, uint64_t *qs, uint16_t *ts, uint32_t min_count,
                                       uint32_t *kept, unsigned long long *n_distinct, unsigned int *max_kept)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long distinct = 0;
    unsigned int mx = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
        const uint64_t o = off[b], n = off[b + 1] - o;
        uint32_t w = 0;
        if (n) {
            uint64_t *q = qs + o; uint16_t *t = ts + o;
            sort_bucket(q, t, n);
            for (uint64_t i = 0; i < n;) {
                uint64_t j = i;
                bool multi = false;
                while (j < n && q[j] == q[i]) { multi |= t[j] != t[i]; j++; }
                distinct++;
                if (!multi && (j - i) > min_count) { q[w] = q[i]; t[w] = t[i]; w++; }
                i = j;
            }
        }
        kept[b] = w;
        mx = w > mx ? w : mx;
    }
    atomicAdd(n_distinct, distinct);
    atomicMax(max_kept, mx);
}

__global__ void bld_compact_kernel(uint64_t nb, const uint64_t *off, const uint64_t *koff, const uint32_t *kept,
                                   const uint64_t *qs, const uint16_t *ts, uint64_t *q_out, uint16_t *t_out, uint8_t *sz)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
        const uint32_t n = kept[b];
        sz[b] = (uint8_t)n;
        const uint64_t s = off[b], d = koff[b];
        for (uint32_t i = 0; i < n; i++) { q_out[d + i] = qs[s + i]; t_out[d + i] = ts[s + i]; }
    }
}

int grid_for(uint64_t n) { const uint64_t w = (n + 255) / 256; return (int)(w < 1 ? 1 : (w > 16384 ? 16384 : w)); }

} // namespace

struct mc_builder {
    int device = 0;
    uint32_t k = 0;
    uint64_t htsize = 0;
    mc::DivU64 div{};
    hipStream_t st = nullptr;
    uint32_t *d_count = nullptr;       // htsize (also: kept sizes after finish)
    uint64_t *d_off = nullptr;         // htsize + 1
    uint64_t *d_q = nullptr; uint16_t *d_t = nullptr;   // n_occ
    uint64_t *d_qf = nullptr; uint16_t *d_tf = nullptr; uint8_t *d_sz = nullptr;   // final
    uint64_t *d_chunk_k = nullptr; uint16_t *d_chunk_t = nullptr;
    uint64_t chunk_cap = 0;
    uint64_t n_occ = 0, n_filled = 0, n_final = 0;
    int phase = 0;                     // 0 count, 1 fill, 2 finished
};


namespace {

int ensure_chunk(mc_builder *b, uint64_t n)
{
    if (n <= b->chunk_cap) return 0;
    if (b->d_chunk_k) (void)hipFree(b->d_chunk_k);
    if (b->d_chunk_t) (void)hipFree(b->d_chunk_t);
    b->chunk_cap = std::max<uint64_t>(n, 1u << 22);
    BCHK(hipMalloc(&b->d_chunk_k, b->chunk_cap * 8));
    BCHK(hipMalloc(&b->d_chunk_t, b->chunk_cap * 2));
    return 0;
}

// exclusive scan of a u32 array of nb elements into off[0..nb] (u64)
int scan_u32(mc_builder *b, const uint32_t *d_v, uint64_t nb, uint64_t *d_off, uint64_t *total)
{
    const uint32_t nblk = (uint32_t)((nb + SC_BUCKETS - 1) / SC_BUCKETS);
    unsigned long long *d_blk = nullptr;
    uint64_t *d_boff = nullptr;
    BCHK(hipMalloc(&d_blk, (size_t)nblk * 8));
    BCHK(hipMalloc(&d_boff, (size_t)nblk * 8));
    hipLaunchKernelGGL(bld_blocksum_kernel, dim3(nblk), dim3(SC_THREADS), 0, b->st, d_v, nb, d_blk);
    BCHK(hipGetLastError());
    std::vector<unsigned long long> blk(nblk);
    BCHK(hipMemcpyAsync(blk.data(), d_blk, (size_t)nblk * 8, hipMemcpyDeviceToHost, b->st));
    BCHK(hipStreamSynchronize(b->st));
    std::vector<uint64_t> boff(nblk);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < nblk; i++) {
        if (blk[i] >= 0xFFFFFFFFull) return bfail("mc_builder: more than 2^32 occurrences in 1024 consecutive buckets");
        boff[i] = acc; acc += blk[i];
    }
    BCHK(hipMemcpyAsync(d_boff, boff.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, b->st));
    hipLaunchKernelGGL(bld_offsets_kernel, dim3(nblk), dim3(SC_THREADS), 0, b->st, d_v, nb, d_boff, d_off);
    BCHK(hipGetLastError());
    BCHK(hipMemcpyAsync(d_off + nb, &acc, 8, hipMemcpyHostToDevice, b->st));
    BCHK(hipStreamSynchronize(b->st));
    (void)hipFree(d_blk); (void)hipFree(d_boff);
    *total = acc;
    return 0;
}

} // namespace

extern "C" {

int mc_builder_open(mc_builder **out, int device, uint32_t k, uint64_t htsize)
{
    if (!out) return bfail("out is NULL");
    *out = nullptr;
    if (k < 2 || k > 32 || htsize < 2 || htsize >= 0xFFFFFFFFull) return bfail("mc_builder_open: bad k / htsize");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return bfail("no HIP device visible");
    if (device < 0) device = 0;
    if (device >= n) return bfail("device index out of range");
    mc_builder *b = new mc_builder();
    b->device = device; b->k = k; b->htsize = htsize; b->div = mc::make_div(htsize);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&b->d_count, htsize * 4);
    if (e == hipSuccess) e = hipMemsetAsync(b->d_count, 0, htsize * 4, b->st);
    if (e != hipSuccess) { delete b; return bfail(std::string("mc_builder_open: ") + hipGetErrorString(e)); }
    *out = b;
    return 0;
}

int mc_builder_count(mc_builder *b, const uint64_t *km, uint64_t n)
{
    if (!b || b->phase != 0) return bfail("mc_builder_count: wrong phase");
    BCHK(hipSetDevice(b->device));
    if (n == 0) return 0;
    if (ensure_chunk(b, n)) return -1;
    BCHK(hipMemcpyAsync(b->d_chunk_k, km, n * 8, hipMemcpyHostToDevice, b->st));
    hipLaunchKernelGGL(bld_count_kernel, dim3(grid_for(n)), dim3(256), 0, b->st, b->d_chunk_k, n, b->k, b->div, b->d_count);
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(b->st));
    b->n_occ += n;
    return 0;
}

int mc_builder_begin_fill(mc_builder *b)
{
    if (!b || b->phase != 0) return bfail("mc_builder_begin_fill: wrong phase");
    BCHK(hipSetDevice(b->device));
    BCHK(hipMalloc(&b->d_off, (b->htsize + 1) * 8));
    uint64_t total = 0;
    if (scan_u32(b, b->d_count, b->htsize, b->d_off, &total)) return -1;
    if (total != b->n_occ) return bfail("mc_builder: internal count mismatch");
    BCHK(hipMalloc(&b->d_q, (total ? total : 1) * 8));
    BCHK(hipMalloc(&b->d_t, (total ? total : 1) * 2));
    b->phase = 1;
    return 0;
}

int mc_builder_fill(mc_builder *b, const uint64_t *km, const uint16_t *tg, uint64_t n)
{
    if (!b || b->phase != 1) return bfail("mc_builder_fill: wrong phase");
    BCHK(hipSetDevice(b->device));
    if (n == 0) return 0;
    if (b->n_filled + n > b->n_occ) return bfail("mc_builder_fill: more occurrences than were counted");
    if (ensure_chunk(b, n)) return -1;
    BCHK(hipMemcpyAsync(b->d_chunk_k, km, n * 8, hipMemcpyHostToDevice, b->st));
    BCHK(hipMemcpyAsync(b->d_chunk_t, tg, n * 2, hipMemcpyHostToDevice, b->st));
    hipLaunchKernelGGL(bld_fill_kernel, dim3(grid_for(n)), dim3(256), 0, b->st, b->d_chunk_k, b->d_chunk_t, n, b->k, b->div,
                       b->d_count, b->d_off, b->d_q, b->d_t);
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(b->st));
    b->n_filled += n;
    return 0;
}

int mc_builder_finish(mc_builder *b, uint32_t min_count, uint64_t *n_distinct, uint64_t *n_stored)
{
    if (!b || b->phase != 1) return bfail("mc_builder_finish: wrong phase");
    if (b->n_filled != b->n_occ) return bfail("mc_builder_finish: the second pass saw fewer occurrences than the first");
    BCHK(hipSetDevice(b->device));
    unsigned long long *d_dist = nullptr; unsigned int *d_max = nullptr;
    BCHK(hipMalloc(&d_dist, 8)); BCHK(hipMalloc(&d_max, 4));
    BCHK(hipMemsetAsync(d_dist, 0, 8, b->st)); BCHK(hipMemsetAsync(d_max, 0, 4, b->st));
    hipLaunchKernelGGL(bld_sort_filter_kernel, dim3(grid_for(b->htsize)), dim3(256), 0, b->st, b->htsize, b->d_off, b->d_q,
                       b->d_t, min_count, b->d_count, d_dist, d_max);
    BCHK(hipGetLastError());
    unsigned long long dist = 0; unsigned int mx = 0;
    BCHK(hipMemcpyAsync(&dist, d_dist, 8, hipMemcpyDeviceToHost, b->st));
    BCHK(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, b->st));
    BCHK(hipStreamSynchronize(b->st));
    (void)hipFree(d_dist); (void)hipFree(d_max);
    if (mx > 255) return bfail("This table can not be stored on disk: Some bucket list size exceeds 255.");
    uint64_t *d_koff = nullptr;
    BCHK(hipMalloc(&d_koff, (b->htsize + 1) * 8));
    uint64_t total = 0;
    if (scan_u32(b, b->d_count, b->htsize, d_koff, &total)) return -1;
    BCHK(hipMalloc(&b->d_qf, (total ? total : 1) * 8));
    BCHK(hipMalloc(&b->d_tf, (total ? total : 1) * 2));
    BCHK(hipMalloc(&b->d_sz, b->htsize));
    hipLaunchKernelGGL(bld_compact_kernel, dim3(grid_for(b->htsize)), dim3(256), 0, b->st, b->htsize, b->d_off, d_koff, b->d_count,
                       b->d_q, b->d_t, b->d_qf, b->d_tf, b->d_sz);
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(b->st));
    (void)hipFree(d_koff);
    (void)hipFree(b->d_q); (void)hipFree(b->d_t); (void)hipFree(b->d_off); (void)hipFree(b->d_count);
    b->d_q = nullptr; b->d_t = nullptr; b->d_off = nullptr; b->d_count = nullptr;
    b->n_final = total;
    b->phase = 2;
    if (n_distinct) *n_distinct = dist;
    if (n_stored) *n_stored = total;
    return 0;
}

int mc_builder_write(mc_builder *b, const char *base, int key_bytes)
{
    if (!b || b->phase != 2 || !base) return bfail("mc_builder_write: wrong phase");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return bfail("key_bytes must be 2, 4 or 8");
    BCHK(hipSetDevice(b->device));
    const std::string s(base);
    FILE *fs = fopen((s + ".sz").c_str(), "wb"), *fk = fopen((s + ".ky").c_str(), "wb"), *fl = fopen((s + ".lb").c_str(), "wb");
    auto closeall = [&]() { if (fs) fclose(fs); if (fk) fclose(fk); if (fl) fclose(fl); };
    if (!fs || !fk || !fl) { closeall(); return bfail("Failed to create " + s + ".sz/.ky/.lb"); }
    const uint64_t CH = 1ull << 24;
    std::vector<uint8_t> hb(CH * 8);
    std::vector<uint8_t> nb(CH * 8);
    for (uint64_t o = 0; o < b->htsize; o += CH) {
        const uint64_t n = std::min(CH, b->htsize - o);
        BCHK(hipMemcpy(hb.data(), b->d_sz + o, n, hipMemcpyDeviceToHost));
        if (fwrite(hb.data(), 1, n, fs) != n) { closeall(); return bfail("short write on " + s + ".sz"); }
    }
    for (uint64_t o = 0; o < b->n_final; o += CH) {
        const uint64_t n = std::min(CH, b->n_final - o);
        BCHK(hipMemcpy(hb.data(), b->d_qf + o, n * 8, hipMemcpyDeviceToHost));
        const uint64_t *q = reinterpret_cast<const uint64_t *>(hb.data());
        if (key_bytes == 8) { if (fwrite(q, 8, n, fk) != n) { closeall(); return bfail("short write on .ky"); } }
        else {
            for (uint64_t i = 0; i < n; i++) {
                if (key_bytes == 4) { if (q[i] > 0xFFFFFFFFull) { closeall(); return bfail("quotient does not fit 4-byte keys"); } reinterpret_cast<uint32_t *>(nb.data())[i] = (uint32_t)q[i]; }
                else                { if (q[i] > 0xFFFFull)     { closeall(); return bfail("quotient does not fit 2-byte keys"); } reinterpret_cast<uint16_t *>(nb.data())[i] = (uint16_t)q[i]; }
            }
            if (fwrite(nb.data(), (size_t)key_bytes, n, fk) != n) { closeall(); return bfail("short write on .ky"); }
        }
        BCHK(hipMemcpy(hb.data(), b->d_tf + o, n * 2, hipMemcpyDeviceToHost));
        if (fwrite(hb.data(), 2, n, fl) != n) { closeall(); return bfail("short write on .lb"); }
    }
    closeall();
    return 0;
}

int mc_builder_close(mc_builder *b)
{
    if (!b) return 0;
    (void)hipSetDevice(b->device);
    void *ptrs[] = {b->d_count, b->d_off, b->d_q, b->d_t, b->d_qf, b->d_tf, b->d_sz, b->d_chunk_k, b->d_chunk_t};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
    return 0;
}

} // extern "C"
