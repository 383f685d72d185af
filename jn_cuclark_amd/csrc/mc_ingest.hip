// mc_ingest.hip -- FASTQ text in, final rows out: record boundaries, 2-bit packing and classification on the device.
//
// The host driver parses 21 GB/s of FASTQ text on 16 threads to feed a kernel that classifies 1.6 G reads/s: file -> CSV
// ran at 58-83 M reads/s (round 3), 1/20 of the kernel.  Here the host only copies the byte range of a batch into a
// pinned buffer; the card finds the newlines, cuts the records (four lines each), packs the sequence lines into the batch
// format of the reference (src/CuCLARK_hh.hh:1615-1715: parts = maximal runs of ACGTU / acgtu of at least k bases, a part =
// [length][containers of 8 bases, first base in the high bits, A=3 C=2 G=1 T=U=0], :1629-1707, :277-302) and classifies them
// from there.  Back come the final rows and, per read, where its header line starts and how long its sequence line is:
// the names stay in the host's text.
//
// Plain 4-line FASTQ only, and only records this code is sure about: a batch with anything else (a line count that is
// not a multiple of four, a header that does not start with '@' or whose name starts with a separator, more reads or
// containers than the buffers take) comes back with a non-zero status and the host driver does that file on its own
// (host/reads.hpp) -- same CSV either way (tests/test_host_cli.py).
#include "mc_internal.hpp"
#include "mc_skm.hpp"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using mcint::fail;
using mcint::set_dev;

namespace mcint {

// One text batch in flight: everything it needs on the card and, pinned, what comes back.  The text itself is uploaded
// straight from the caller's (pageable) memory on the buffer's own stream -- measured (tools/ubench/h2d_pageable.hip):
// hipMemcpyAsync from pageable memory does 20 GB/s from one thread and 50 GB/s from two or more, as much as copies from
// pinned buffers, while PINNING a ring large enough for whole batches (8 x 400 MB) cost a second per file.
struct TextBuf {
    // device
    uint8_t *text = nullptr;
    uint32_t *nl = nullptr;           // positions of the newlines
    uint32_t *blk = nullptr, *off32 = nullptr, *tot = nullptr; uint64_t *base = nullptr;          // scan: newlines per workgroup
    uint32_t *ncon = nullptr, *coff = nullptr, *ctot = nullptr; uint64_t *cbase = nullptr;        // scan: containers per read
    uint32_t *d_hdr = nullptr, *d_len = nullptr, *ptr = nullptr, *d_counts = nullptr;
    uint16_t *con = nullptr, *d_fin = nullptr;
    // pinned
    uint32_t *hdr = nullptr, *len = nullptr;      // per read: offset of the header line ('@'), bytes of the sequence line
    uint16_t *fin = nullptr;
    uint32_t *counts = nullptr;       // [0] reads, [1] containers, [2] status, [3] lines
    hipStream_t up = nullptr;         // the upload of this buffer's text
    hipEvent_t ev_up = nullptr, ev_k = nullptr, ev_counts = nullptr, ev_done = nullptr;
    uint64_t n_bytes = 0;
    bool submitted = false;
};
struct TextState {
    std::vector<TextBuf> bufs;
    uint64_t max_text = 0, max_reads = 0, max_con = 0;
    std::mutex mu;                    // the compute and copy-out queues of the context are fed under it
};

} // namespace mcint

namespace {

constexpr int ING_THREADS = 256, ING_PER = 16, ING_TILE = ING_THREADS * ING_PER;      // bytes of text per workgroup

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *t, uint64_t at, uint64_t n)
{
    uint32_t m = 0;
    if (at + ING_PER <= n && ((uintptr_t)(t + at) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(t + at);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int b = 0; b < 4; b++) m |= (((w[j] >> (8 * b)) & 0xFFu) == 10u ? 1u : 0u) << (4 * j + b);
    } else {
        for (int i = 0; i < ING_PER; i++) if (at + i < n && t[at + i] == 10) m |= 1u << i;
    }
    return m;
}

__global__ __launch_bounds__(ING_THREADS)
void ing_count_kernel(const uint8_t *t, uint64_t n, uint32_t *blk)
{
    __shared__ uint32_t s_a[ING_THREADS / 64];
    const uint64_t at = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_PER;
    const uint32_t c = at < n ? (uint32_t)__popc(nl_mask16(t, at, n)) : 0u;
    uint32_t tot;
    mc::block_exclusive_scan(c, s_a, tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

__global__ __launch_bounds__(ING_THREADS)
void ing_scatter_kernel(const uint8_t *t, uint64_t n, const uint32_t *off32, const uint64_t *base, uint32_t *nl, uint64_t nl_cap)
{
    __shared__ uint32_t s_a[ING_THREADS / 64];
    const uint64_t at = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_PER;
    uint32_t m = at < n ? nl_mask16(t, at, n) : 0u;
    uint32_t tot;
    uint64_t o = base[blockIdx.x >> 10] + off32[blockIdx.x] + mc::block_exclusive_scan((uint32_t)__popc(m), s_a, tot);
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        m &= m - 1u;
        if (o < nl_cap) nl[o] = (uint32_t)(at + b);
        o++;
    }
}

__device__ __forceinline__ int base_code(uint8_t c)
{
    switch (c | 0x20u) {
    case 'a': return 3;
    case 'c': return 2;
    case 'g': return 1;
    case 't': case 'u': return 0;
    default: return -1;
    }
}

// lines -> reads; status bits: 1 a record this code does not vouch for, 2 lines not a multiple of four, 4 more reads than room
__global__ void ing_begin_kernel(const uint64_t *nl_total, uint64_t nl_cap, uint64_t max_reads, uint32_t *counts)
{
    const uint64_t lines = *nl_total;
    uint32_t st = 0;
    if (lines % 4u) st |= 2u;
    if (lines > nl_cap || lines / 4u > max_reads) st |= 4u;
    counts[3] = (uint32_t)lines;
    counts[0] = st & 4u ? 0u : (uint32_t)(lines / 4u);
    counts[1] = 1u;
    counts[2] = st;
}

// one thread per read: where its lines are, how many containers it packs to
__global__ __launch_bounds__(256)
void ing_records_kernel(const uint8_t *t, const uint32_t *nl, uint32_t k, uint64_t max_reads, uint32_t *counts,
                        uint32_t *hdr, uint32_t *len, uint32_t *ncon)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t n_reads = counts[0];
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= max_reads; r += stride) {
        if (r >= n_reads) { ncon[r] = 0; continue; }
        const uint32_t hs = r ? nl[4 * r - 1] + 1u : 0u, he = nl[4 * r], ss = he + 1u, se = nl[4 * r + 1];
        // the host's indexer (host/reads.hpp) takes the byte behind '@' as the first byte of the name whatever it is, and lets
        // a header whose name starts with a separator run on: not reproduced here, handed back instead
        bool odd = t[hs] != '@' || he < hs + 2u;
        if (!odd) { const uint8_t c = t[hs + 1]; odd = c == ' ' || c == '\t'; }
        if (odd) atomicOr(&counts[2], 1u);
        hdr[r] = hs;
        len[r] = se - ss;
        uint32_t c = 0;
        if (se - ss >= k) {
            uint32_t run = 0;
            for (uint32_t i = ss; i <= se; i++) {
                const bool in = i < se && base_code(t[i]) >= 0;
                if (in) run++;
                else { if (run >= k) c += 1u + (run + 7u) / 8u; run = 0; }
            }
        }
        ncon[r] = c;
    }
}

__global__ __launch_bounds__(256)
void ing_pack_kernel(const uint8_t *t, const uint32_t *nl, uint32_t k, uint64_t max_con, uint32_t *counts,
                     const uint32_t *coff, const uint64_t *cbase, uint32_t *ptr, uint16_t *con)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t n_reads = counts[0];
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += stride) {
        const uint64_t at0 = cbase[r >> 10] + coff[r];
        if (r == n_reads) {
            ptr[r] = (uint32_t)(at0 <= max_con ? at0 : 0u);
            counts[1] = (uint32_t)(at0 <= max_con ? (at0 ? at0 : 1u) : 1u);
            if (at0 > max_con) { counts[0] = 0u; atomicOr(&counts[2], 4u); }
            continue;
        }
        if (at0 > max_con) { ptr[r] = 0; continue; }            // (the batch is handed back: status 4)
        ptr[r] = (uint32_t)at0;
        const uint32_t ss = nl[4 * r] + 1u, se = nl[4 * r + 1];
        if (se - ss < k) continue;
        uint64_t at = at0;
        uint32_t i = ss;
        while (i < se) {
            while (i < se && base_code(t[i]) < 0) i++;
            uint32_t e = i;
            while (e < se && base_code(t[e]) >= 0) e++;
            const uint32_t plen = e - i;
            if (plen >= k && at + 1u + (plen + 7u) / 8u <= max_con) {
                con[at++] = (uint16_t)plen;                       // (a FASTQ sequence line of more than 65535 bases: the host's rule, the low 16 bits)
                for (uint32_t p = i; p < e; p += 8u) {
                    uint32_t w = 0;
                    for (uint32_t j = 0; j < 8u; j++) w = (w << 2) | (p + j < e ? (uint32_t)base_code(t[p + j]) : 0u);
                    con[at++] = (uint16_t)w;
                }
            }
            i = e;
        }
    }
}

void text_free(mc_ctx *c)
{
    mcint::TextState *T = c->text;
    if (!T) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto &b : T->bufs) {
        void *d[] = {b.text, b.nl, b.blk, b.off32, b.tot, b.base, b.ncon, b.coff, b.ctot, b.cbase, b.d_hdr, b.d_len, b.ptr, b.d_counts, b.con, b.d_fin};
        for (void *q : d) if (q) (void)hipFree(q);
        void *h[] = {b.hdr, b.len, b.fin, b.counts};
        for (void *q : h) if (q) (void)hipHostFree(q);
        hipEvent_t e[] = {b.ev_up, b.ev_k, b.ev_counts, b.ev_done};
        for (hipEvent_t q : e) if (q) (void)hipEventDestroy(q);
        if (b.up) (void)hipStreamDestroy(b.up);
    }
    delete T;
    c->text = nullptr;
}

} // namespace

namespace mcint {
void text_release(mc_ctx *c) { text_free(c); }
}

extern "C" {

int mc_text_alloc(mc_ctx *c, uint32_t n_bufs, uint64_t max_text, uint64_t max_reads, uint64_t max_con)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    if (n_bufs < 1 || max_text < 16 || max_reads < 1) return fail(MC_EINVAL, "n_bufs, max_text and max_reads must be positive");
    if (max_text >= 0xFFFFFFF0ull || max_con > 0xFFFFFFFFull || max_reads > 0x3FFFFFFFull) return fail(MC_EINVAL, "a text batch holds less than 4 GB, 2^32 containers, 2^30 reads");
    int rc = set_dev(c); if (rc) return rc;
    text_free(c);
    if (max_con < 8) max_con = 8;
    mcint::TextState *T = new mcint::TextState();
    c->text = T;
    T->max_text = max_text; T->max_reads = max_reads; T->max_con = max_con;
    T->bufs.resize(n_bufs);
    const uint64_t nl_cap = 4 * max_reads + 4;
    const size_t nblk = (size_t)((max_text + ING_TILE) / ING_TILE) + 1;
    bool ok = true;
    for (auto &b : T->bufs) {
        ok = ok && hipMalloc(&b.text, max_text + 16) == hipSuccess && hipMalloc(&b.nl, nl_cap * 4) == hipSuccess;
        ok = ok && hipMalloc(&b.blk, nblk * 4) == hipSuccess && hipMalloc(&b.off32, nblk * 4) == hipSuccess && hipMalloc(&b.base, (nblk / 1024 + 4) * 8) == hipSuccess;
        ok = ok && hipMalloc(&b.tot, (nblk / 1024 + 4) * 4) == hipSuccess && hipMalloc(&b.ctot, ((max_reads + 1) / 1024 + 4) * 4) == hipSuccess;
        ok = ok && hipMalloc(&b.ncon, (max_reads + 1) * 4) == hipSuccess && hipMalloc(&b.coff, (max_reads + 1) * 4) == hipSuccess;
        ok = ok && hipMalloc(&b.cbase, ((max_reads + 1) / 1024 + 4) * 8) == hipSuccess;
        ok = ok && hipMalloc(&b.d_hdr, max_reads * 4) == hipSuccess && hipMalloc(&b.d_len, max_reads * 4) == hipSuccess;
        ok = ok && hipMalloc(&b.ptr, (max_reads + 1) * 4) == hipSuccess && hipMalloc(&b.d_counts, 16) == hipSuccess;
        ok = ok && hipMalloc(&b.con, (max_con + 8) * 2) == hipSuccess && hipMalloc(&b.d_fin, max_reads * MC_FINAL_ROW * 2) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b.hdr, max_reads * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b.len, max_reads * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b.fin, max_reads * MC_FINAL_ROW * 2, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&b.counts, 16, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&b.up, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.ev_up, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b.ev_k, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.ev_counts, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); text_free(c); return fail(MC_ENOMEM, "text batch buffers: not enough pinned or device memory -- use more, smaller batches (-b)"); }
    return MC_OK;
}

int mc_text_buffers(mc_ctx *c, uint32_t buf, uint32_t **hdr_off, uint32_t **seq_len, uint16_t **final_rows)
{
    if (!c || !c->text || buf >= c->text->bufs.size()) return fail(MC_EINVAL, "bad text buffer index");
    mcint::TextBuf &b = c->text->bufs[buf];
    if (hdr_off) *hdr_off = b.hdr;
    if (seq_len) *seq_len = b.len;
    if (final_rows) *final_rows = b.fin;
    return MC_OK;
}

int mc_text_submit(mc_ctx *c, uint32_t buf, const uint8_t *text, uint64_t n_bytes)
{
    if (!c || !c->text || buf >= c->text->bufs.size()) return fail(MC_EINVAL, "bad text buffer index");
    if (!c->db_loaded) return fail(MC_ESTATE, "mc_text_submit before a database was loaded");
    if (n_bytes && !text) return fail(MC_EINVAL, "text is NULL");
    mcint::TextState *T = c->text;
    if (n_bytes > T->max_text) return fail(MC_EINVAL, "text batch larger than allocated");
    int rc = set_dev(c); if (rc) return rc;
    mcint::TextBuf &b = T->bufs[buf];
    if (b.submitted) return fail(MC_ESTATE, "text buffer resubmitted before mc_text_wait");
    // the upload: on this buffer's own stream, from the caller's memory, outside the lock -- several threads upload at once
    if (n_bytes) HIPCHK(hipMemcpyAsync(b.text, text, n_bytes, hipMemcpyHostToDevice, b.up));
    if (n_bytes && text[n_bytes - 1] != '\n') {                       // a file that ends without a newline
        HIPCHK(hipMemsetAsync(b.text + n_bytes, '\n', 1, b.up));
        n_bytes++;
    }
    HIPCHK(hipEventRecord(b.ev_up, b.up));
    b.n_bytes = n_bytes;
    std::lock_guard<std::mutex> lk(T->mu);
    b.submitted = true;
    hipStream_t st = c->streams[0];
    HIPCHK(hipStreamWaitEvent(st, b.ev_up, 0));
    const uint64_t nl_cap = 4 * T->max_reads + 4;
    // newlines: per workgroup -> offsets -> positions
    const uint32_t nblk = (uint32_t)((n_bytes + ING_TILE - 1) / ING_TILE) + 1u;
    const uint32_t nb2 = (nblk + 1023u) / 1024u;
    hipLaunchKernelGGL(ing_count_kernel, dim3(nblk), dim3(ING_THREADS), 0, st, b.text, n_bytes, b.blk);
    hipLaunchKernelGGL(mc::sk::sk_scan_kernel, dim3(nb2), dim3(mc::RL_THREADS), 0, st, b.blk, (uint64_t)nblk, b.off32, b.tot);
    hipLaunchKernelGGL(mc::sk::sk_scan_blocks_kernel, dim3(1), dim3(256), 0, st, b.tot, nb2, b.base);
    hipLaunchKernelGGL(ing_scatter_kernel, dim3(nblk), dim3(ING_THREADS), 0, st, b.text, n_bytes, b.off32, b.base, b.nl, nl_cap);
    hipLaunchKernelGGL(ing_begin_kernel, dim3(1), dim3(1), 0, st, b.base + nb2, nl_cap, T->max_reads, b.d_counts);
    // records -> containers per read -> offsets -> packed reads
    const uint32_t g = (uint32_t)std::min<uint64_t>((T->max_reads + 256) / 256, (uint64_t)c->n_cu * 32);
    hipLaunchKernelGGL(ing_records_kernel, dim3(g), dim3(256), 0, st, b.text, b.nl, c->k, T->max_reads, b.d_counts, b.d_hdr, b.d_len, b.ncon);
    const uint32_t nr1 = (uint32_t)(T->max_reads + 1), nb3 = (nr1 + 1023u) / 1024u;
    hipLaunchKernelGGL(mc::sk::sk_scan_kernel, dim3(nb3), dim3(mc::RL_THREADS), 0, st, b.ncon, (uint64_t)nr1, b.coff, b.ctot);
    hipLaunchKernelGGL(mc::sk::sk_scan_blocks_kernel, dim3(1), dim3(256), 0, st, b.ctot, nb3, b.cbase);
    hipLaunchKernelGGL(ing_pack_kernel, dim3(g), dim3(256), 0, st, b.text, b.nl, c->k, T->max_con, b.d_counts, b.coff, b.cbase, b.ptr, b.con);
    HIPCHK(hipGetLastError());
    // classification, reads and containers counted on the device
    rc = mcint::launch_query(c, b.ptr, b.con, T->max_reads, T->max_con, MC_F_FINAL, b.d_fin, nullptr, st, b.d_counts);
    if (rc) return rc;
    HIPCHK(hipEventRecord(b.ev_k, st));
    HIPCHK(hipStreamWaitEvent(c->s_out, b.ev_k, 0));
    HIPCHK(hipMemcpyAsync(b.counts, b.d_counts, 16, hipMemcpyDeviceToHost, c->s_out));
    HIPCHK(hipEventRecord(b.ev_counts, c->s_out));
    return MC_OK;
}

int mc_text_wait(mc_ctx *c, uint32_t buf, uint64_t *n_reads, uint32_t *status)
{
    if (!c || !c->text || buf >= c->text->bufs.size()) return fail(MC_EINVAL, "bad text buffer index");
    mcint::TextState *T = c->text;
    mcint::TextBuf &b = T->bufs[buf];
    if (!b.submitted) return fail(MC_ESTATE, "text batch was never submitted");
    int rc = set_dev(c); if (rc) return rc;
    HIPCHK(hipEventSynchronize(b.ev_counts));
    const uint64_t n = b.counts[2] ? 0 : b.counts[0];
    {
        std::lock_guard<std::mutex> lk(T->mu);          // exactly the rows and record offsets of this batch
        if (n) {
            HIPCHK(hipMemcpyAsync(b.fin, b.d_fin, n * MC_FINAL_ROW * 2, hipMemcpyDeviceToHost, c->s_out));
            HIPCHK(hipMemcpyAsync(b.hdr, b.d_hdr, n * 4, hipMemcpyDeviceToHost, c->s_out));
            HIPCHK(hipMemcpyAsync(b.len, b.d_len, n * 4, hipMemcpyDeviceToHost, c->s_out));
        }
        HIPCHK(hipEventRecord(b.ev_done, c->s_out));
    }
    HIPCHK(hipEventSynchronize(b.ev_done));
    if (n_reads) *n_reads = n;
    if (status) *status = b.counts[2];
    b.submitted = false;
    return MC_OK;
}

int mc_text_free(mc_ctx *c)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    text_free(c);
    return MC_OK;
}

} // extern "C"
