// mc_api.hip -- host side of libmcclark.so: context, database load + re-layout,
// batch buffers, streams, launches.  The C ABI is declared in include/mc_api.h; each
// function there names the CuClarkDB member it replaces (reference src/CuClarkDB.cu).
//
// There is NO CPU fallback in this library: without a gfx950 device every entry point
// that needs one fails with MC_ENODEVICE / MC_EHIP.
#include "mc_internal.hpp"
#include "mc_minimizer.hpp"
#include "mc_skm.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
}

namespace mcint {

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

int set_dev(mc_ctx *c)
{
    HIPCHK(hipSetDevice(c->device));
    return MC_OK;
}


} // namespace mcint

using mcint::fail;
using mcint::set_dev;
using mcint::Scope;
using mcint::Batch;
using mcint::Slot;

namespace {

void free_db(mc_ctx *c)
{
    if (c->d_lines) (void)hipFree(c->d_lines);
    if (c->d_ovf_keys) (void)hipFree(c->d_ovf_keys);
    if (c->d_ovf_labels) (void)hipFree(c->d_ovf_labels);
    if (c->d_mz_lines) (void)hipFree(c->d_mz_lines);
    if (c->d_mz_extra && c->mz_extra_own_alloc) (void)hipFree(c->d_mz_extra);
    if (c->d_sk_lines) (void)hipFree(c->d_sk_lines);
    if (c->d_sk_extra) (void)hipFree(c->d_sk_extra);
    c->d_sk_lines = nullptr; c->d_sk_extra = nullptr; c->sk_n_lines = 0;
    c->mz_extra_own_alloc = false; c->mz_extra_reserved = 0;
    c->d_lines = nullptr; c->d_ovf_keys = nullptr; c->d_ovf_labels = nullptr;
    c->d_mz_lines = nullptr; c->d_mz_extra = nullptr;
    c->db_loaded = false;
}

template <int LINE, bool WIDE>
int launch_fill(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, const uint16_t *d_labels,
                uint64_t nb, const uint64_t *d_koff, const uint64_t *d_ooff, uint32_t nblk)
{
    typedef typename mc::KeyOf<WIDE>::type key_t;
    hipLaunchKernelGGL((mc::fill_lines_kernel<LINE, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, c->streams[0],
                       d_sz, static_cast<const key_t *>(d_keys), d_labels, nb, d_koff, d_ooff, c->d_lines,
                       static_cast<key_t *>(c->d_ovf_keys), c->d_ovf_labels);
    HIPCHK(hipGetLastError());
    return MC_OK;
}

template <int LINE, bool WIDE>
int query_occupancy(int &occ)
{
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mc::query_kernel<LINE, WIDE>, mc::BLOCK_THREADS, 0));
    return MC_OK;
}

// ---------------------------------------------------------------------------
// minimizer index, built from bucket-order chunks in two passes (count, place)
// ---------------------------------------------------------------------------
void index_abort(mc_ctx *c)
{
    if (c->build.d_count) (void)hipFree(c->build.d_count);
    if (c->build.d_failed) (void)hipFree(c->build.d_failed);
    if (c->build.d_blk) (void)hipFree(c->build.d_blk);
    if (c->build.d_koff) (void)hipFree(c->build.d_koff);
    if (c->build.d_st_sz) (void)hipFree(c->build.d_st_sz);
    if (c->build.d_st_keys) (void)hipFree(c->build.d_st_keys);
    if (c->build.d_st_labels) (void)hipFree(c->build.d_st_labels);
    if (c->build.d_cursor) (void)hipFree(c->build.d_cursor);
    if (c->build.d_off32) (void)hipFree(c->build.d_off32);
    if (c->build.d_blk_base) (void)hipFree(c->build.d_blk_base);
    if (c->build.d_entries) { if (c->build.entries_on_host) (void)hipHostFree(c->build.d_entries); else (void)hipFree(c->build.d_entries); }
    if (c->build.d_count_mz) (void)hipFree(c->build.d_count_mz);
    c->build = mcint::IndexBuild();
}

// the minimizer index needs a minimizer space (canonical m-mers) well above the number of lines
bool mz_eligible(const mc_ctx *c, uint64_t n_keys_total)
{
    const uint32_t mz_m = mc::mz::mmer_len(c->k);
    return c->k >= 16 && mz_m >= 6 && (mz_m >= 20 || (1ull << (2 * mz_m - 1)) >= 4 * (n_keys_total / 6 + 1024));
}

int sk_begin(mc_ctx *c, uint64_t n_keys_total, uint32_t part, uint32_t n_parts);
int sk_next_pass(mc_ctx *c);
int sk_end(mc_ctx *c);
bool use_sk(const mc_ctx *c) { return c->index_mode == 2 && mc::sk::sk_supported(c->k); }

bool use_both(const mc_ctx *c) { return c->index_mode == 3 && mc::sk::sk_supported(c->k); }

// The lines of the minimizer index and, right behind them in the SAME allocation, exactly the extra lines the first build
// pass counted.  Round 3 allocated the lines at mc_index_begin with a guessed share for the extra lines behind them
// (12-90 %: 28 GB behind the headline table, which used 1.7); round 4 allocates between the passes, when the number is known.
// Measured on the way (one box, 10 M reads per launch): extra lines in an allocation of their own -- a late hipMalloc or a
// hipMemCreate / hipMemMap range -- cost the genome-shaped table 9.5 ms instead of 7.1 (it is being ANOTHER address range
// that costs, not small pages); lines AND extra lines mapped into one reserved range (hipMemAddressReserve + two hipMemMap)
// run 7.1 there but the headline table 6.18 instead of 6.02 ms.  One hipMalloc of the exact size has neither cost.
int mz_alloc_lines(mc_ctx *c, uint64_t n_extra)
{
    const size_t lbytes = (size_t)(c->mz_n_local ? c->mz_n_local : 1) * mc::mz::MZ_LINE;
    const size_t ebytes = (size_t)(n_extra ? n_extra : 1) * mc::mz::MZ_LINE;
    // MC_MZ_ALLOC_LIMIT (bytes): a cap on the lines of one context -- a card shared with other tenants, and how the tests
    // reach the "does not fit" paths (the fallback to the bucket-line table, a group that cuts the table into more parts)
    if (const char *e = getenv("MC_MZ_ALLOC_LIMIT")) {
        const uint64_t lim = strtoull(e, nullptr, 10);
        if (lim && (uint64_t)lbytes > lim)
            return fail(MC_ENOMEM, "minimizer lines of " + std::to_string(lbytes) + " bytes exceed MC_MZ_ALLOC_LIMIT");
    }
    if (hipMalloc(&c->d_mz_lines, lbytes + ebytes) != hipSuccess) {
        (void)hipGetLastError();
        c->d_mz_lines = nullptr;
        return fail(MC_ENOMEM, "not enough HBM for " + std::to_string(lbytes + ebytes) + " bytes of minimizer lines (" +
                               std::to_string(n_extra) + " extra lines among them); a denser MC_MZ_FILL needs less");
    }
    c->d_mz_extra = c->d_mz_lines + lbytes;
    c->mz_extra_reserved = n_extra; c->mz_extra_own_alloc = false;
    HIPCHK(hipMemsetAsync(c->d_mz_lines, 0xFF, lbytes + ebytes, c->streams[0]));
    return MC_OK;
}

int index_begin(mc_ctx *c, uint64_t n_keys_total, uint32_t part, uint32_t n_parts)
{
    if (n_parts < 1 || part >= n_parts) return fail(MC_EINVAL, "bad part / n_parts");
    if (use_sk(c)) return sk_begin(c, n_keys_total, part, n_parts);
    if (!mz_eligible(c, n_keys_total))
        return fail(MC_EINVAL, "the minimizer index needs k >= 16 and a minimizer space above the line count");
    free_db(c);
    index_abort(c);
    // k-mers per 12-slot line.  Fewer per line = fewer overflowing lines = a faster kernel (genome-shaped table,
    // 5.5e9 k-mers: 4 / 5 / 6 / 8 per line -> 993 / 913 / 828 / 705 Mreads/s at 193 / 161 / 140 / 126 GB), so a table
    // that is alone on the card takes the room it finds: the sparsest fill in [3, 12] that leaves 16 GB free.
    // Parts of one table must agree on the fill: a caller that builds the parts one by one (mc_load_db_part,
    // mc_index_begin with n_parts > 1) gets the fill every part can afford IF each has a card like this one to
    // itself -- a pure function of (n_keys_total, n_parts, HBM of the card) -- unless a group loader chose
    // (fill_hint) or MC_MZ_FILL says otherwise.
    double per_line = c->fill_hint;
    if (const char *e = getenv("MC_MZ_FILL")) { const double v = atof(e); if (v >= 1.0 && v <= 64.0) per_line = v; }
    if (per_line <= 0.0) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        per_line = mcint::choose_fill(n_keys_total, n_parts, n_parts == 1 ? (uint64_t)fr : (uint64_t)tot);
    }
    const uint64_t want = mcint::lines_per_part(n_keys_total, n_parts, per_line);
    if (want == 0)
        return fail(MC_EINVAL, "minimizer index: " + std::to_string(n_keys_total) + " k-mers at " + std::to_string(per_line) +
                               " per line over " + std::to_string(n_parts) + " part(s) need more than 2^32 lines per part");
    if (!c->group_loading) { c->auto_decision = 0; c->sk_d_hint = 0; }
    const bool both = use_both(c) && (n_parts == 1 || c->group_loading);
    if (both) {          // the super-k-mer build's counters first (it resets the context), the minimizer index's next to them
        const int rc = sk_begin(c, n_keys_total, part, n_parts);
        if (rc != MC_OK) return rc;
        c->build.both = true;
    }
    c->mz_n_local = (uint32_t)want;
    c->mz_part = part; c->mz_n_parts = n_parts;
    c->mz_m = mc::mz::mmer_len(c->k);
    c->info = mc_db_info{};
    c->info.part = part; c->info.n_parts = n_parts;
    c->build.mz_per_line = per_line;
    hipStream_t st = c->streams[0];
    uint32_t **cnt = both ? &c->build.d_count_mz : &c->build.d_count;
    // (the lines are allocated between the passes: then it is known how many extra lines go behind them -- and, with
    // MC_INDEX=auto, which index they are for)
    int rc = MC_OK;
    if (const char *e = getenv("MC_MZ_ALLOC_LIMIT")) {          // (refused here already: the callers' fallbacks key on mc_index_begin)
        const uint64_t lim = strtoull(e, nullptr, 10);
        if (lim && (uint64_t)c->mz_n_local * mc::mz::MZ_LINE > lim && !both)
            rc = fail(MC_ENOMEM, "minimizer lines of " + std::to_string((uint64_t)c->mz_n_local * mc::mz::MZ_LINE) + " bytes exceed MC_MZ_ALLOC_LIMIT");
    }
    if (rc == MC_OK && (hipMalloc(cnt, (size_t)(c->mz_n_local ? c->mz_n_local : 1) * 4) != hipSuccess ||
                        (!c->build.d_failed && hipMalloc(&c->build.d_failed, 4) != hipSuccess))) {
        (void)hipGetLastError();
        rc = fail(MC_ENOMEM, "not enough HBM for the line counters of the minimizer index");
    }
    if (rc != MC_OK) { const std::string keep = g_err; free_db(c); index_abort(c); g_err = keep; return rc; }
    HIPCHK(hipMemsetAsync(*cnt, 0, (size_t)(c->mz_n_local ? c->mz_n_local : 1) * 4, st));
    HIPCHK(hipMemsetAsync(c->build.d_failed, 0, 4, st));
    c->build.open = true; c->build.pass = 0; c->build.n_keys_total = n_keys_total;
    return MC_OK;
}

// One chunk, device arrays in the context's key width.  Nothing is allocated per call (the per-workgroup sums
// and their offsets live in the build's grow-only scratch), nothing is copied to the host and nothing is waited
// for: the launches are queued on the build stream and the caller may reuse its arrays once that stream passed.
template <bool WIDE>
int index_add_typed(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, const uint16_t *d_labels, uint64_t n_keys,
                    uint64_t b0, uint64_t b1)
{
    typedef typename mc::KeyOf<WIDE>::type key_t;
    const uint64_t nb = b1 - b0;
    hipStream_t st = c->streams[0];
    const uint32_t nblk = (uint32_t)((nb + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    mcint::IndexBuild &B = c->build;
    if (nblk > B.blk_cap) {
        HIPCHK(hipStreamSynchronize(st));                   // an earlier chunk may still read the old scratch
        if (B.d_blk) (void)hipFree(B.d_blk);
        if (B.d_koff) (void)hipFree(B.d_koff);
        B.d_blk = nullptr; B.d_koff = nullptr; B.blk_cap = 0;
        HIPCHK(hipMalloc(&B.d_blk, (size_t)nblk * 2 * 4));   // k-mers per workgroup | (unused) overflow sums
        HIPCHK(hipMalloc(&B.d_koff, (size_t)nblk * 8));
        B.blk_cap = nblk;
    }
    hipLaunchKernelGGL(mc::block_sums_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz, nb, 255u, B.d_blk, B.d_blk + nblk);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(mc::mz::mz_scan_blocks_kernel, dim3(1), dim3(256), 0, st, B.d_blk, nblk, B.d_koff, n_keys, B.d_failed);
    HIPCHK(hipGetLastError());
    if (B.both && c->build.pass == 0) {
        hipLaunchKernelGGL((mc::mz::mz_build_kernel<0, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz,
                           static_cast<const key_t *>(d_keys), d_labels, nb, n_keys, b0, c->htsize, B.d_koff, c->k, c->mz_m,
                           c->mz_part, c->mz_n_parts, c->mz_n_local, B.d_count_mz,
                           (uint8_t *)nullptr, (uint8_t *)nullptr, B.d_failed);
        HIPCHK(hipGetLastError());
    }
    if (B.sk) {
        if (c->build.pass == 0)
            hipLaunchKernelGGL((mc::sk::sk_build_kernel<0, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz,
                               static_cast<const key_t *>(d_keys), d_labels, nb, n_keys, b0, c->htsize, B.d_koff, c->k, c->mz_m,
                               c->mz_part, c->mz_n_parts, B.sk_n_fine, B.d_count, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                               (const uint64_t *)nullptr, (mc::sk::SkSlot *)nullptr, B.d_failed);
        else
            hipLaunchKernelGGL((mc::sk::sk_build_kernel<1, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz,
                               static_cast<const key_t *>(d_keys), d_labels, nb, n_keys, b0, c->htsize, B.d_koff, c->k, c->mz_m,
                               c->mz_part, c->mz_n_parts, B.sk_n_fine, B.d_cursor, B.d_count, B.d_off32, B.d_blk_base,
                               static_cast<mc::sk::SkSlot *>(B.d_entries), B.d_failed);
    } else if (c->build.pass == 0)
        hipLaunchKernelGGL((mc::mz::mz_build_kernel<0, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz,
                           static_cast<const key_t *>(d_keys), d_labels, nb, n_keys, b0, c->htsize, B.d_koff, c->k, c->mz_m,
                           c->mz_part, c->mz_n_parts, c->mz_n_local, c->build.d_count,
                           (uint8_t *)nullptr, (uint8_t *)nullptr, B.d_failed);
    else
        hipLaunchKernelGGL((mc::mz::mz_build_kernel<1, WIDE>), dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz,
                           static_cast<const key_t *>(d_keys), d_labels, nb, n_keys, b0, c->htsize, B.d_koff, c->k, c->mz_m,
                           c->mz_part, c->mz_n_parts, c->mz_n_local, c->build.d_count,
                           c->d_mz_lines, c->d_mz_extra, B.d_failed);
    HIPCHK(hipGetLastError());
    c->build.fed[c->build.pass] += n_keys;
    c->build.bucket_lo = std::min(c->build.bucket_lo, b0);
    c->build.bucket_hi = std::max(c->build.bucket_hi, b1);
    return MC_OK;
}

int convert_keys(mc_ctx *c, void *d_raw, int key_bytes, uint64_t n, bool raw_owned, void **out, bool *out_owned);

// sync: return only when the build stream has passed (the caller may then reuse or free its arrays)
int index_add_device(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, int key_bytes, const uint16_t *d_labels,
                     uint64_t n_keys, uint64_t b0, uint64_t b1, bool sync = false)
{
    if (!c->build.open) return fail(MC_ESTATE, "mc_index_add before mc_index_begin");
    if (b0 >= b1 || b1 > c->htsize) return fail(MC_EINVAL, "bad bucket range of the chunk");
    Scope tmp;
    void *keys = nullptr; bool owned = false;
    int rc = convert_keys(c, const_cast<void *>(d_keys), key_bytes, n_keys, false, &keys, &owned);
    if (rc) return rc;
    if (owned) tmp.add(keys);
    rc = c->wide ? index_add_typed<true>(c, d_sz, keys, d_labels, n_keys, b0, b1)
                 : index_add_typed<false>(c, d_sz, keys, d_labels, n_keys, b0, b1);
    if ((owned || sync) && rc == MC_OK) HIPCHK(hipStreamSynchronize(c->streams[0]));       // the widened copy goes out of scope
    return rc;
}

// staging arrays of the build for a chunk of nb buckets / n_keys k-mers of key_bytes each (grow-only)
int index_staging(mc_ctx *c, uint64_t nb, uint64_t n_keys, int key_bytes)
{
    mcint::IndexBuild &B = c->build;
    (void)key_bytes;                                         // the key staging holds 8 bytes per k-mer: any width
    if (nb > B.st_sz_cap || n_keys > B.st_key_cap) {
        HIPCHK(hipStreamSynchronize(c->streams[0]));
        if (B.d_st_sz) (void)hipFree(B.d_st_sz);
        if (B.d_st_keys) (void)hipFree(B.d_st_keys);
        if (B.d_st_labels) (void)hipFree(B.d_st_labels);
        B.d_st_sz = nullptr; B.d_st_keys = nullptr; B.d_st_labels = nullptr;
        const size_t cb = (size_t)std::max<uint64_t>(nb, B.st_sz_cap), ck = (size_t)std::max<uint64_t>(n_keys ? n_keys : 1, B.st_key_cap);
        B.st_sz_cap = 0; B.st_key_cap = 0;
        HIPCHK(hipMalloc(&B.d_st_sz, cb));
        HIPCHK(hipMalloc(&B.d_st_keys, ck * 8));
        HIPCHK(hipMalloc(&B.d_st_labels, ck * 2));
        B.st_sz_cap = cb; B.st_key_cap = ck;
    }
    return MC_OK;
}

int index_add_host(mc_ctx *c, const uint8_t *sz, const void *keys, int key_bytes, const uint16_t *labels,
                   uint64_t n_keys, uint64_t b0, uint64_t b1)
{
    if (!c->build.open) return fail(MC_ESTATE, "mc_index_add before mc_index_begin");
    if (b0 >= b1 || b1 > c->htsize) return fail(MC_EINVAL, "bad bucket range of the chunk");
    const uint64_t nb = b1 - b0;
    int rc = index_staging(c, nb, n_keys, key_bytes);
    if (rc) return rc;
    mcint::IndexBuild &B = c->build;
    hipStream_t st = c->streams[0];
    // queued behind the kernels of the chunk before (same stream): the staging arrays are free by then
    HIPCHK(hipMemcpyAsync(B.d_st_sz, sz, nb, hipMemcpyHostToDevice, st));
    if (n_keys) {
        HIPCHK(hipMemcpyAsync(B.d_st_keys, keys, n_keys * (size_t)key_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(B.d_st_labels, labels, n_keys * 2, hipMemcpyHostToDevice, st));
    }
    rc = index_add_device(c, B.d_st_sz, B.d_st_keys, key_bytes, B.d_st_labels, n_keys, b0, b1);
    // the caller's host arrays are its own again when this returns (pageable memory is staged by the runtime before
    // the copy call returns; pinned memory is read by the DMA engine later)
    if (rc == MC_OK) HIPCHK(hipStreamSynchronize(st));
    return rc;
}

// between the passes: sizes known -> headers, extra lines
int index_next_pass(mc_ctx *c)
{
    if (!c->build.open || c->build.pass != 0) return fail(MC_ESTATE, "mc_index_next_pass out of order");
    if (c->build.both) {
        // How do the k-mers clump around their minimizers?  Isolated k-mers spread over the fine lines like a Poisson
        // process (variance / mean of the counts = 1); the k-mers of genomes come as runs that share a minimizer, related
        // genomes put several runs behind one (the genome-shaped table: ~20).  Clumped tables get records, the others lines.
        mcint::IndexBuild &B = c->build;
        Scope tmp;
        unsigned long long *d_mom = nullptr, mom[2] = {0, 0};
        TMP_MALLOC(tmp, d_mom, 16);
        HIPCHK(hipMemsetAsync(d_mom, 0, 16, c->streams[0]));
        const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)B.sk_n_fine + 255) / 256, (uint64_t)c->n_cu * 8));
        hipLaunchKernelGGL(mc::sk::sk_moments_kernel, dim3(g), dim3(256), 0, c->streams[0], B.d_count, (uint64_t)B.sk_n_fine, d_mom);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(mom, d_mom, 16, hipMemcpyDeviceToHost, c->streams[0]));
        HIPCHK(hipStreamSynchronize(c->streams[0]));
        const double nf = (double)B.sk_n_fine, mean = (double)mom[0] / nf;
        const double fano = mean > 0.0 ? ((double)mom[1] / nf - mean * mean) / mean : 0.0;
        double need = 2.5;
        if (const char *e = getenv("MC_AUTO_CLUMP")) { const double v = atof(e); if (v > 0.0) need = v; }
        const bool records = c->auto_decision ? c->auto_decision == 1 : fano >= need;
        c->auto_decision = records ? 1 : 2;
        if (getenv("MC_SKM_VERBOSE"))
            fprintf(stderr, "libmcclark: MC_INDEX=auto: variance / mean of the k-mers per fine line %.2f -> %s\n", fano, records ? "super-k-mer records" : "minimizer lines");
        B.both = false;
        if (records) {
            (void)hipFree(B.d_count_mz); B.d_count_mz = nullptr;
            c->mz_n_local = 0;
        } else {
            (void)hipFree(B.d_count); (void)hipFree(B.d_cursor); (void)hipFree(B.d_off32); (void)hipFree(B.d_blk_base);
            B.d_count = B.d_count_mz; B.d_count_mz = nullptr; B.d_cursor = nullptr; B.d_off32 = nullptr; B.d_blk_base = nullptr;
            B.sk = false;
        }
    }
    if (c->build.sk) return sk_next_pass(c);
    hipStream_t st = c->streams[0];
    const uint64_t n = c->mz_n_local;
    const uint32_t nblk = (uint32_t)std::max<uint64_t>(1, (n + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    Scope tmp;
    uint32_t *d_blk = nullptr; unsigned long long *d_tot = nullptr; uint64_t *d_boff = nullptr;
    TMP_MALLOC(tmp, d_blk, (size_t)nblk * 4);
    TMP_MALLOC(tmp, d_tot, 4 * 8);
    TMP_MALLOC(tmp, d_boff, (size_t)nblk * 8);
    HIPCHK(hipMemsetAsync(d_tot, 0, 4 * 8, st));
    hipLaunchKernelGGL(mc::mz::mz_extras_blocksum_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, c->build.d_count, n, d_blk, d_tot);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> blk(nblk);
    unsigned long long tot[4];
    unsigned int failed0 = 0;
    HIPCHK(hipMemcpyAsync(blk.data(), d_blk, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&failed0, c->build.d_failed, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (failed0 & 4u) { free_db(c); index_abort(c); return fail(MC_EINVAL, "bucket sizes do not sum to n_keys"); }
    std::vector<uint64_t> boff(nblk);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < nblk; i++) { boff[i] = acc; acc += blk[i]; }
    if (acc >= 0xFFFFFFFFull) { free_db(c); index_abort(c); return fail(MC_EINVAL, "minimizer index: more than 2^32 extra lines"); }
    c->build.n_extra = acc; c->build.n_spilled = tot[0]; c->build.n_over = tot[1]; c->build.longest = (uint32_t)tot[2];
    c->build.n_crowded = tot[3];
    {
        const int rc = mz_alloc_lines(c, acc);
        if (rc != MC_OK) { const std::string keep = g_err; free_db(c); index_abort(c); g_err = keep; return rc; }
    }
    HIPCHK(hipMemcpyAsync(d_boff, boff.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mc::mz::mz_header_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, c->build.d_count, n, d_boff, c->d_mz_lines, c->k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->build.d_count, 0, (size_t)(n ? n : 1) * 4, st));      // the counters become the cursors
    HIPCHK(hipStreamSynchronize(st));
    c->build.pass = 1;
    return MC_OK;
}

int index_end(mc_ctx *c)
{
    if (!c->build.open || c->build.pass != 1) return fail(MC_ESTATE, "mc_index_end out of order");
    if (c->build.sk) return sk_end(c);
    if (c->build.fed[0] != c->build.fed[1]) { free_db(c); index_abort(c); return fail(MC_EINVAL, "the two passes were fed different k-mer counts"); }
    hipStream_t st = c->streams[0];
    if (c->build.n_over && !getenv("MC_MZ_NO_REGROUP")) {      // whole minimizer groups first in overflowing lines
        const int gr = (int)std::min<uint64_t>(((uint64_t)c->mz_n_local + 255) / 256, (uint64_t)c->n_cu * 16);
        hipLaunchKernelGGL(mc::mz::mz_regroup_kernel, dim3(gr), dim3(256), 0, st, c->build.d_count, c->mz_n_local, c->k,
                           c->mz_m, c->d_mz_lines, c->d_mz_extra);
        HIPCHK(hipGetLastError());
    }
    if (c->mz_n_local) {      // first lines in ascending key order: what lets a lookup scan half a line (mz_match_line)
        const int gs = (int)std::min<uint64_t>(((uint64_t)c->mz_n_local + 255) / 256, (uint64_t)c->n_cu * 32);
        hipLaunchKernelGGL(mc::mz::mz_sort_lines_kernel, dim3(gs), dim3(256), 0, st, c->d_mz_lines, c->mz_n_local);
        HIPCHK(hipGetLastError());
    }
    if (c->build.n_extra && c->build.n_extra < 0xFFFFFFFFull) {      // extra lines too, each on its own
        const int gs = (int)std::min<uint64_t>((c->build.n_extra + 255) / 256, (uint64_t)c->n_cu * 32);
        hipLaunchKernelGGL(mc::mz::mz_sort_lines_kernel, dim3(gs), dim3(256), 0, st, c->d_mz_extra, (uint32_t)c->build.n_extra);
        HIPCHK(hipGetLastError());
    }
    unsigned int failed = 0;
    HIPCHK(hipMemcpyAsync(&failed, c->build.d_failed, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (failed) {
        free_db(c); index_abort(c);
        return fail(MC_EINVAL, (failed & 4u) ? "bucket sizes do not sum to n_keys" :
                               (failed & 2u) ? "minimizer index: the second pass held k-mers the first did not (a line outgrew the chain sized for it)"
                                             : "minimizer index: a chain of a crowded line overflowed twice");
    }
    mc_db_info &I = c->info;
    I.htsize = c->htsize;
    I.shard_begin = c->build.bucket_lo == ~0ull ? 0 : c->build.bucket_lo;
    I.shard_end = c->build.bucket_hi ? c->build.bucket_hi : c->htsize;
    I.n_keys = c->build.fed[0];
    I.n_overflow_buckets = c->build.n_extra;            // extra lines
    I.n_overflow_keys = c->build.n_spilled;
    I.line_bytes = mc::mz::MZ_LINE;
    I.line_capacity = mc::mz::MZ_CAP;
    I.device_bytes = ((uint64_t)c->mz_n_local + c->mz_extra_reserved) * mc::mz::MZ_LINE;
    I.index_kind = MC_INDEX_MINIMIZER;
    I.n_lines = (uint64_t)c->mz_n_local * c->mz_n_parts;                 // part p = lines [p * n, (p + 1) * n) of the table's
    I.line_begin = (uint64_t)c->mz_n_local * c->mz_part; I.line_end = I.line_begin + c->mz_n_local;
    I.n_extra_lines = c->build.n_extra; I.n_lines_crowded = c->build.n_crowded;
    I.n_lines_overflowing = c->build.n_over; I.n_spilled_keys = c->build.n_spilled; I.largest_line = c->build.longest;
    {   // k-mers this part owns = the sum of its line counters (fed[] counts what streamed past)
        Scope tmp;
        unsigned long long *d_sum = nullptr, sum = 0;
        TMP_MALLOC(tmp, d_sum, 8);
        HIPCHK(hipMemsetAsync(d_sum, 0, 8, st));
        const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)c->mz_n_local + 255) / 256, (uint64_t)c->n_cu * 8));
        hipLaunchKernelGGL(mc::mz::mz_sum_u32_kernel, dim3(g), dim3(256), 0, st, c->build.d_count, (uint64_t)c->mz_n_local, d_sum);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&sum, d_sum, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        I.n_keys_owned = sum;
    }
    index_abort(c);         // releases the counters
    int occ = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (mc::mz::mz_query_kernel<mc::mz::MZ_ALL, 0>), mc::BLOCK_THREADS, 0));
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;
    if (const char *e = getenv("MC_GRID_OCC")) { const int v = atoi(e); if (v >= 1 && v < occ) occ = v; }
    c->grid_blocks = occ * c->n_cu;
    c->db_loaded = true;
    return MC_OK;
}


// ---------------------------------------------------------------------------
// super-k-mer index (mc_skm.hpp): the same two passes over the chunks; the final line count is chosen at the end
// ---------------------------------------------------------------------------
static const uint32_t SK_D_LCM = 27720;          // every merge factor d in 1 .. 12 divides the fine line count

int sk_begin(mc_ctx *c, uint64_t n_keys_total, uint32_t part, uint32_t n_parts)
{
    free_db(c);
    index_abort(c);
    if (!c->group_loading) c->sk_d_hint = 0;
    // fine lines: about one per two k-mers of this part (the final lines are 1 .. 12 of them each)
    double per_fine = 2.0;
    if (const char *e = getenv("MC_SKM_FINE")) { const double v = atof(e); if (v >= 0.25 && v <= 64.0) per_fine = v; }
    uint64_t want = (uint64_t)((double)n_keys_total / (double)n_parts / per_fine) + 1;
    want = (want + SK_D_LCM - 1) / SK_D_LCM * SK_D_LCM;
    if (want >= mcint::MZ_MAX_LINES) want = mcint::MZ_MAX_LINES / SK_D_LCM * SK_D_LCM;
    mcint::IndexBuild &B = c->build;
    B.sk = true; B.sk_n_fine = (uint32_t)want;
    c->mz_part = part; c->mz_n_parts = n_parts; c->mz_m = mc::mz::mmer_len(c->k);
    c->info = mc_db_info{};
    c->info.part = part; c->info.n_parts = n_parts;
    const size_t nb = (size_t)want * 4;
    if (hipMalloc(&B.d_count, nb) != hipSuccess || hipMalloc(&B.d_cursor, nb) != hipSuccess || hipMalloc(&B.d_off32, nb) != hipSuccess ||
        hipMalloc(&B.d_blk_base, ((size_t)(want / mc::RL_BUCKETS) + 2) * 8) != hipSuccess || hipMalloc(&B.d_failed, 4) != hipSuccess) {
        (void)hipGetLastError();
        index_abort(c);
        return fail(MC_ENOMEM, "not enough HBM for the counters of " + std::to_string(want) + " fine lines");
    }
    hipStream_t st = c->streams[0];
    HIPCHK(hipMemsetAsync(B.d_count, 0, nb, st));
    HIPCHK(hipMemsetAsync(B.d_failed, 0, 4, st));
    B.open = true; B.pass = 0; B.n_keys_total = n_keys_total;
    return MC_OK;
}

// exclusive scan of n u32 counters: off32 (offset inside the counter's workgroup of 1024) + blk_base (u64 per workgroup, the
// total behind the last); d_blk_tot is scratch of (n / 1024 + 1) u32
static int sk_scan(mc_ctx *c, const uint32_t *d_cnt, uint64_t n, uint32_t *d_off32, uint64_t *d_blk_base, uint64_t *total)
{
    hipStream_t st = c->streams[0];
    const uint32_t nblk = (uint32_t)std::max<uint64_t>(1, (n + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    Scope tmp;
    uint32_t *d_tot = nullptr;
    TMP_MALLOC(tmp, d_tot, (size_t)nblk * 4);
    hipLaunchKernelGGL(mc::sk::sk_scan_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_cnt, n, d_off32, d_tot);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(mc::sk::sk_scan_blocks_kernel, dim3(1), dim3(256), 0, st, d_tot, nblk, d_blk_base);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(total, d_blk_base + nblk, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MC_OK;
}

int sk_next_pass(mc_ctx *c)
{
    mcint::IndexBuild &B = c->build;
    hipStream_t st = c->streams[0];
    unsigned int failed0 = 0;
    HIPCHK(hipMemcpyAsync(&failed0, B.d_failed, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (failed0 & 4u) { free_db(c); index_abort(c); return fail(MC_EINVAL, "bucket sizes do not sum to n_keys"); }
    uint64_t total = 0;
    int rc = sk_scan(c, B.d_count, B.sk_n_fine, B.d_off32, B.d_blk_base, &total);
    if (rc) return rc;
    B.sk_n_entries = total;
    // The entries (16 bytes per stored k-mer) live until the lines are written, NEXT TO the lines: on a card they would crowd
    // (more than 45 % of what is free: the lines of a genome-shaped table take 9-16 bytes per k-mer themselves) they go to pinned
    // host memory instead, written once and read once over PCIe -- slower to build, and a table of 12e9 instead of 8e9 k-mers
    // per card.  MC_SKM_ENTRIES=host|device forces either.
    const size_t ebytes = (size_t)(total ? total : 1) * sizeof(mc::sk::SkSlot);
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    const char *where = getenv("MC_SKM_ENTRIES");
    bool on_host = where ? !strcmp(where, "host") : (double)ebytes > 0.45 * (double)fr;
    if (!on_host && hipMalloc(&B.d_entries, ebytes) != hipSuccess) { (void)hipGetLastError(); B.d_entries = nullptr; on_host = !(where && !strcmp(where, "device")); }
    if (on_host && !B.d_entries) {
        if (hipHostMalloc(&B.d_entries, ebytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); B.d_entries = nullptr; }
        else B.entries_on_host = true;
    }
    if (!B.d_entries) {
        free_db(c); index_abort(c);
        return fail(MC_ENOMEM, "super-k-mer index: neither HBM nor pinned host memory for " + std::to_string(total) + " entries of 16 bytes while it is built");
    }
    if (B.entries_on_host && getenv("MC_SKM_VERBOSE")) fprintf(stderr, "libmcclark: super-k-mer index: %.1f GB of entries in pinned host memory\n", (double)ebytes / 1e9);
    HIPCHK(hipMemsetAsync(B.d_cursor, 0, (size_t)B.sk_n_fine * 4, st));
    HIPCHK(hipStreamSynchronize(st));
    B.pass = 1;
    return MC_OK;
}

int sk_end(mc_ctx *c)
{
    using namespace mc::sk;
    mcint::IndexBuild &B = c->build;
    hipStream_t st = c->streams[0];
    if (B.fed[0] != B.fed[1]) { free_db(c); index_abort(c); return fail(MC_EINVAL, "the two passes were fed different k-mer counts"); }
    const uint32_t n_fine = B.sk_n_fine;
    const SkSlot *ent = static_cast<const SkSlot *>(B.d_entries);
    uint32_t *d_nrec = B.d_cursor;                   // the cursors have done their work
    {
        const int gr = (int)std::min<uint64_t>(((uint64_t)n_fine + 255) / 256, (uint64_t)c->n_cu * 16);
        hipLaunchKernelGGL(sk_records_kernel, dim3(gr), dim3(256), 0, st, B.d_count, B.d_off32, B.d_blk_base, ent, n_fine, d_nrec);
        HIPCHK(hipGetLastError());
    }
    // the merge factor: the largest d whose lines overflow rarely enough and fit the card
    Scope tmp;
    unsigned long long *d_out = nullptr;
    TMP_MALLOC(tmp, d_out, 12 * 6 * 8);
    HIPCHK(hipMemsetAsync(d_out, 0, 12 * 6 * 8, st));
    for (uint32_t d = 1; d <= 12; d++) {
        const uint64_t nl = n_fine / d;
        const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>((nl + 255) / 256, (uint64_t)c->n_cu * 16));
        hipLaunchKernelGGL(sk_eval_kernel, dim3(g), dim3(256), 0, st, B.d_count, d_nrec, nl, d, d_out + (d - 1) * 6);
        HIPCHK(hipGetLastError());
    }
    unsigned long long ev[12][6];
    unsigned int failed = 0;
    HIPCHK(hipMemcpyAsync(ev, d_out, sizeof ev, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&failed, B.d_failed, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (failed) {
        free_db(c); index_abort(c);
        return fail(MC_EINVAL, (failed & 4u) ? "bucket sizes do not sum to n_keys"
                                             : "super-k-mer index: the second pass held k-mers the first did not");
    }
    // (genome-shaped table, 5.5e9 k-mers: d = 7 / 4 / 3 -> 0.59 / 0.03 / 0.006 % of the lines overflow, 51 / 88 / 117 GB,
    // 8.1 / 7.4 / 7.4 ms per 10 M reads: past 0.05 % the memory buys nothing)
    double max_over = 0.0005;
    if (const char *e = getenv("MC_SKM_OVERFLOW")) { const double v = atof(e); if (v > 0.0 && v <= 1.0) max_over = v; }
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    uint32_t d = 0;
    for (uint32_t t = 12; t >= 1; t--) {
        const uint64_t nl = n_fine / t;
        const uint64_t bytes = (nl + ev[t - 1][1]) * (uint64_t)SK_LINE + nl * 8 + (1ull << 30);
        if ((double)ev[t - 1][0] <= max_over * (double)nl && bytes + (2ull << 30) <= (uint64_t)fr) { d = t; break; }
    }
    if (d == 0) {         // no d meets the overflow bound in the room there is: the sparsest that fits
        for (uint32_t t = 1; t <= 12 && d == 0; t++) {
            const uint64_t nl = n_fine / t;
            if ((nl + ev[t - 1][1]) * (uint64_t)SK_LINE + nl * 8 + (3ull << 30) <= (uint64_t)fr) d = t;
        }
    }
    if (c->sk_d_hint >= 1 && c->sk_d_hint <= 12) d = c->sk_d_hint;       // a group loader: one layout for all members
    if (const char *e = getenv("MC_SKM_D")) { const int v = atoi(e); if (v >= 1 && v <= 12) d = (uint32_t)v; }
    if (d == 0 || d > 12) { free_db(c); index_abort(c); return fail(MC_ENOMEM, "super-k-mer index: the lines do not fit the free HBM at any merge factor"); }
    const uint32_t n_lines = n_fine / d;
    const unsigned long long *E = ev[d - 1];
    if (E[1] >= 0xFFFFFFF0ull) { free_db(c); index_abort(c); return fail(MC_EINVAL, "super-k-mer index: more than 2^32 extra lines"); }
    const size_t lbytes = (size_t)n_lines * SK_LINE, xbytes = (size_t)(E[1] ? E[1] : 1) * SK_LINE;
    uint32_t *d_xcnt = nullptr, *d_xoff = nullptr; uint64_t *d_xblk = nullptr;
    if (hipMalloc(&c->d_sk_lines, lbytes) != hipSuccess || hipMalloc(&c->d_sk_extra, xbytes) != hipSuccess) {
        (void)hipGetLastError();
        free_db(c); index_abort(c);
        return fail(MC_ENOMEM, "super-k-mer index: not enough HBM for " + std::to_string(lbytes + xbytes) + " bytes of lines");
    }
    TMP_MALLOC(tmp, d_xcnt, (size_t)n_lines * 4);
    TMP_MALLOC(tmp, d_xoff, (size_t)n_lines * 4);
    TMP_MALLOC(tmp, d_xblk, ((size_t)(n_lines / mc::RL_BUCKETS) + 2) * 8);
    HIPCHK(hipMemsetAsync(c->d_sk_lines, 0, lbytes, st));
    HIPCHK(hipMemsetAsync(c->d_sk_extra, 0, xbytes, st));
    {
        const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n_lines + 255) / 256, (uint64_t)c->n_cu * 16));
        hipLaunchKernelGGL(sk_extras_kernel, dim3(g), dim3(256), 0, st, B.d_count, d_nrec, (uint64_t)n_lines, d, d_xcnt);
        HIPCHK(hipGetLastError());
    }
    uint64_t x_total = 0;
    int rc = sk_scan(c, d_xcnt, n_lines, d_xoff, d_xblk, &x_total);
    if (rc) { free_db(c); index_abort(c); return rc; }
    if (x_total != E[1]) { free_db(c); index_abort(c); return fail(MC_EINVAL, "internal: extra lines counted twice differ"); }
    {
        const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n_lines + 255) / 256, (uint64_t)c->n_cu * 16));
        hipLaunchKernelGGL(sk_encode_kernel, dim3(g), dim3(256), 0, st, B.d_count, B.d_off32, B.d_blk_base, ent, d_nrec, n_lines, d,
                           d_xoff, d_xblk, c->d_sk_lines, c->d_sk_extra, B.d_failed);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(&failed, B.d_failed, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (failed) {
        free_db(c); index_abort(c);
        return fail(MC_EINVAL, (failed & 1u) ? "super-k-mer index: a hashed chain overflowed" : "internal: the records of a line came out differently the second time");
    }
    c->sk_n_lines = n_lines; c->sk_d = d;
    mc_db_info &I = c->info;
    I.htsize = c->htsize;
    I.shard_begin = B.bucket_lo == ~0ull ? 0 : B.bucket_lo;
    I.shard_end = B.bucket_hi ? B.bucket_hi : c->htsize;
    I.n_keys = B.fed[0];
    I.n_keys_owned = B.sk_n_entries;                    // entries: k-mers of this part, ties stored once per window
    I.n_overflow_buckets = E[1]; I.n_overflow_keys = 0;
    I.line_bytes = SK_LINE; I.line_capacity = SK_SLOTS * SK_W;
    I.device_bytes = (uint64_t)lbytes + xbytes;
    I.index_kind = MC_INDEX_SUPERKMER;
    I.n_lines = (uint64_t)n_lines * c->mz_n_parts;
    I.line_begin = (uint64_t)n_lines * c->mz_part; I.line_end = I.line_begin + n_lines;
    I.n_extra_lines = E[1]; I.n_lines_crowded = E[2]; I.n_lines_overflowing = E[0]; I.n_spilled_keys = E[4]; I.largest_line = (uint32_t)E[3];
    if (getenv("MC_SKM_VERBOSE")) {
        fprintf(stderr, "libmcclark: super-k-mer index: %llu entries in %llu records (%.2f per record); merge factor d:", (unsigned long long)B.sk_n_entries,
                E[4], (double)B.sk_n_entries / (double)(E[4] ? E[4] : 1));
        for (uint32_t t = 1; t <= 12; t++)
            fprintf(stderr, " %u:%.3f%%/%.1fGB", t, 100.0 * (double)ev[t - 1][0] / (double)(n_fine / t), (double)((n_fine / t + ev[t - 1][1]) * 128ull) / 1e9);
        fprintf(stderr, " -> d = %u, %u lines\n", d, n_lines);
    }
    index_abort(c);
    int occ = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (mc::sk::sk_query_kernel<mc::mz::MZ_ALL, 0>), mc::BLOCK_THREADS, 0));
    if (occ < 1) occ = 1;
    // a persistent grid of 7 workgroups per CU although 8 fit (measured, genome-shaped table, one box, twice each:
    // 8 / 7 / 6 / 5 / 4 per CU -> 7.47 / 6.68 / 6.73 / 6.77 / 7.14 ms per 10 M reads)
    const int fits = occ;
    if (occ > 7) occ = 7;
    if (const char *e = getenv("MC_GRID_OCC")) { const int v = atoi(e); if (v >= 1 && v <= fits) occ = v; }
    c->grid_blocks = occ * c->n_cu;
    c->db_loaded = true;
    return MC_OK;
}

// whole (shard of a) table resident on the device as raw arrays: one chunk per pass
int relayout_mz(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, const uint16_t *d_labels,
                uint64_t n_keys, uint64_t shard_begin, uint64_t shard_end)
{
    int rc = index_begin(c, n_keys, 0, 1);
    if (rc) return rc;
    const int kb = c->wide ? 8 : 4;
    rc = index_add_device(c, d_sz, d_keys, kb, d_labels, n_keys, shard_begin, shard_end);
    if (rc == MC_OK) rc = index_next_pass(c);
    if (rc == MC_OK) rc = index_add_device(c, d_sz, d_keys, kb, d_labels, n_keys, shard_begin, shard_end);
    if (rc == MC_OK) rc = index_end(c);
    if (rc != MC_OK) { const std::string keep = g_err; free_db(c); index_abort(c); g_err = keep; }
    return rc;
}

// Build the bucket lines from raw arrays resident on the device; d_keys holds u32
// quotients, or u64 when the context is in wide-key mode.
int relayout(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, const uint16_t *d_labels,
             uint64_t n_keys, uint64_t shard_begin, uint64_t shard_end, bool fallback = false)
{
    if (!fallback && c->index_mode >= 1 && mz_eligible(c, n_keys)) {
        const int rcm = relayout_mz(c, d_sz, d_keys, d_labels, n_keys, shard_begin, shard_end);
        if (rcm != MC_ENOMEM) return rcm;
        // not enough HBM for the minimizer lines: the direct table (about 3x slower to query) -- said aloud
        fprintf(stderr, "libmcclark: %s; falling back to the bucket-line table\n", g_err.c_str());
        fallback = true;
    }
    const uint64_t nb = shard_end - shard_begin;
    hipStream_t st = c->streams[0];

    // 1. histogram of bucket sizes -> line size
    Scope tmp;
    unsigned long long *d_hist = nullptr;
    TMP_MALLOC(tmp, d_hist, 256 * sizeof(unsigned long long));
    HIPCHK(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), st));
    {
        const uint64_t want = (nb + 256 * 64 - 1) / (256 * 64);
        const uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)c->n_cu * 8));
        hipLaunchKernelGGL(mc::size_hist_kernel, dim3(g), dim3(256), 0, st, d_sz, nb, d_hist);
        HIPCHK(hipGetLastError());
    }
    unsigned long long hist[256];
    HIPCHK(hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));

    uint64_t total = 0, nonempty = 0;
    for (int i = 0; i < 256; i++) { total += hist[i] * (uint64_t)i; if (i) nonempty += hist[i]; }
    if (total != n_keys)
        return fail(MC_EINVAL, "bucket sizes sum to " + std::to_string(total) + " but n_keys is " + std::to_string(n_keys));
    auto over = [&](int cap) { uint64_t b = 0; for (int i = cap + 1; i < 256; i++) b += hist[i]; return b; };
    // 64-byte lines unless more than 3 % of the non-empty buckets would overflow them
    const char *force = getenv("MC_LINE_BYTES");
    const int cap64 = c->wide ? mc::LineCfg<64, true>::CAP : mc::LineCfg<64, false>::CAP;
    const int cap128 = c->wide ? mc::LineCfg<128, true>::CAP : mc::LineCfg<128, false>::CAP;
    uint32_t line = (nonempty == 0 || over(cap64) * 100 <= nonempty * 3) ? 64u : 128u;
    if (force && (atoi(force) == 64 || atoi(force) == 128)) line = (uint32_t)atoi(force);
    const int cap = line == 64 ? cap64 : cap128;
    const size_t kb = c->wide ? 8 : 4;
    uint64_t n_ovf_b = 0, n_ovf_k = 0;
    for (int i = cap + 1; i < 256; i++) { n_ovf_b += hist[i]; n_ovf_k += hist[i] * (uint64_t)i; }

    // 2. per-workgroup sums -> exclusive offsets (host scan of ~nb/1024 numbers)
    const uint32_t nblk = (uint32_t)((nb + mc::RL_BUCKETS - 1) / mc::RL_BUCKETS);
    uint32_t *d_bk = nullptr, *d_bo = nullptr;
    uint64_t *d_koff = nullptr, *d_ooff = nullptr;
    TMP_MALLOC(tmp, d_bk, (size_t)nblk * 4);
    TMP_MALLOC(tmp, d_bo, (size_t)nblk * 4);
    hipLaunchKernelGGL(mc::block_sums_kernel, dim3(nblk), dim3(mc::RL_THREADS), 0, st, d_sz, nb, (uint32_t)cap, d_bk, d_bo);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> bk(nblk), bo(nblk);
    HIPCHK(hipMemcpyAsync(bk.data(), d_bk, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(bo.data(), d_bo, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint64_t> koff(nblk), ooff(nblk);
    uint64_t ak = 0, ao = 0;
    for (uint32_t i = 0; i < nblk; i++) { koff[i] = ak; ooff[i] = ao; ak += bk[i]; ao += bo[i]; }
    if (ak != n_keys || ao != n_ovf_k) return fail(MC_EINVAL, "internal: block sums disagree with histogram");
    TMP_MALLOC(tmp, d_koff, (size_t)nblk * 8);
    TMP_MALLOC(tmp, d_ooff, (size_t)nblk * 8);
    HIPCHK(hipMemcpyAsync(d_koff, koff.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ooff, ooff.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, st));

    // 3. allocate and fill the lines
    free_db(c);
    const size_t line_bytes = (size_t)nb * line;
    if (hipMalloc(&c->d_lines, line_bytes ? line_bytes : 16) != hipSuccess) {
        (void)hipGetLastError();
        return fail(MC_ENOMEM, "hipMalloc of " + std::to_string(line_bytes) + " bytes of bucket lines failed");
    }
    HIPCHK(hipMalloc(&c->d_ovf_keys, (size_t)(n_ovf_k ? n_ovf_k : 4) * kb));
    HIPCHK(hipMalloc(&c->d_ovf_labels, (size_t)(n_ovf_k ? n_ovf_k : 4) * 2));
    int rc;
    if (!c->wide) rc = line == 64 ? launch_fill<64, false>(c, d_sz, d_keys, d_labels, nb, d_koff, d_ooff, nblk)
                                  : launch_fill<128, false>(c, d_sz, d_keys, d_labels, nb, d_koff, d_ooff, nblk);
    else          rc = line == 64 ? launch_fill<64, true>(c, d_sz, d_keys, d_labels, nb, d_koff, d_ooff, nblk)
                                  : launch_fill<128, true>(c, d_sz, d_keys, d_labels, nb, d_koff, d_ooff, nblk);
    if (rc != MC_OK) return rc;
    HIPCHK(hipStreamSynchronize(st));

    c->info = mc_db_info{};
    c->info.index_kind = MC_INDEX_BUCKET_LINES;
    c->info.index_fallback = fallback ? 1u : 0u;
    c->info.n_parts = 1;
    c->info.n_keys_owned = n_keys;
    c->info.htsize = c->htsize;
    c->info.shard_begin = shard_begin;
    c->info.shard_end = shard_end;
    c->info.n_keys = n_keys;
    c->info.n_overflow_buckets = n_ovf_b;
    c->info.n_overflow_keys = n_ovf_k;
    c->info.line_bytes = line;
    c->info.line_capacity = (uint32_t)cap;
    c->info.device_bytes = line_bytes + n_ovf_k * (kb + 2);

    // persistent grid: as many workgroups as stay resident
    int occ = 0;
    if (!c->wide) rc = line == 64 ? query_occupancy<64, false>(occ) : query_occupancy<128, false>(occ);
    else          rc = line == 64 ? query_occupancy<64, true>(occ) : query_occupancy<128, true>(occ);
    if (rc != MC_OK) return rc;
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;
    if (const char *e = getenv("MC_GRID_OCC")) {          // tuning knob: resident workgroups per CU
        const int v = atoi(e);
        if (v >= 1 && v < occ) occ = v;
    }
    c->grid_blocks = occ * c->n_cu;
    c->db_loaded = true;
    return MC_OK;
}

template <typename IN, typename OUT>
int widen(mc_ctx *c, const void *in, uint64_t n, void *out)
{
    hipLaunchKernelGGL((mc::widen_keys_kernel<IN, OUT>), dim3(c->n_cu * 8), dim3(256), 0, c->streams[0],
                       static_cast<const IN *>(in), n, static_cast<OUT *>(out));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->streams[0]));
    return MC_OK;
}

// d_raw: n quotients of key_bytes each (the width of the .ky file).  *out: the same
// quotients in the context's key width.  When the widths agree *out = d_raw; otherwise a
// new array (and d_raw is freed if raw_owned).  *out_owned: caller frees *out.
int convert_keys(mc_ctx *c, void *d_raw, int key_bytes, uint64_t n, bool raw_owned, void **out, bool *out_owned)
{
    const int want = c->wide ? 8 : 4;
    if (key_bytes == want) { *out = d_raw; *out_owned = raw_owned; return MC_OK; }
    Scope raw;                                   // the raw array, if ours, is released on every path below
    if (raw_owned) raw.add(d_raw);
    if (key_bytes > want)
        return fail(MC_EINVAL, "8-byte key file for a k/htsize whose quotients fit 4 bytes: rebuild the database "
                               "(the reference writes 4-byte keys here, src/main.cc:267-275)");
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, (n ? n : 1) * (size_t)want));
    int rc;
    if (key_bytes == 2) rc = c->wide ? widen<uint16_t, uint64_t>(c, d_raw, n, d) : widen<uint16_t, uint32_t>(c, d_raw, n, d);
    else                rc = widen<uint32_t, uint64_t>(c, d_raw, n, d);
    if (rc != MC_OK) { (void)hipFree(d); return rc; }
    *out = d; *out_owned = true;
    return MC_OK;
}

int norm_shard(mc_ctx *c, uint64_t &sb, uint64_t &se)
{
    if (sb == 0 && se == 0) se = c->htsize;
    if (sb >= se || se > c->htsize) return fail(MC_EINVAL, "bad shard range");
    return MC_OK;
}

} // namespace

namespace mcint {

int launch_query(mc_ctx *c, const uint32_t *d_ptr, const uint16_t *d_con, uint64_t n_reads,
                 uint64_t n_con, uint32_t flags, uint16_t *d_final, uint16_t *d_rows, hipStream_t st, const uint32_t *n_dev)
{
    if (n_reads == 0) return MC_OK;
    // reads_ptr holds u32 container offsets: one batch is below 2^32 reads and containers (the kernels count in 32 bits)
    if (n_reads > 0xFFFFFFE0ull || n_con > 0xFFFFFFFFull) return fail(MC_EINVAL, "a batch holds at most 2^32 - 32 reads and 2^32 - 1 containers");
    mc::QueryArgs a{};
    a.reads_ptr = d_ptr; a.containers = d_con; a.n_reads = n_reads; a.n_containers = n_con ? n_con : 1; a.n_dev = n_dev;
    a.lines = c->d_lines; a.ovf_keys = c->d_ovf_keys; a.ovf_labels = c->d_ovf_labels;
    a.shard_begin = c->info.shard_begin; a.shard_end = c->info.shard_end;
    a.div = c->div; a.k = c->k; a.maxhits = c->maxhits; a.flags = flags;
    a.stage_ok = (((uintptr_t)d_con) & 15u) == 0 ? 1u : 0u;
    a.final_rows = d_final; a.sparse_rows = d_rows; a.over_maxhits = c->d_over;
    const uint64_t n_groups = (n_reads + mc::GROUP_READS - 1) / mc::GROUP_READS;
    const uint64_t want = (n_groups + mc::WAVES_PER_BLOCK - 1) / mc::WAVES_PER_BLOCK;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)c->grid_blocks);
    const dim3 g(grid), b(mc::BLOCK_THREADS);
    if (c->d_sk_lines) {
        mc::sk::SkArgs m{};
        m.q = a; m.lines = c->d_sk_lines; m.extra = c->d_sk_extra;
        m.n_lines = c->sk_n_lines; m.part = c->mz_part; m.n_parts = c->mz_n_parts; m.m = c->mz_m;
#define MC_SK_LAUNCH(SH)                                                                                                   \
        do {                                                                                                               \
            if (c->k == 31u) hipLaunchKernelGGL((mc::sk::sk_query_kernel<SH, 31>), g, b, 0, st, m);                       \
            else if (c->k == 27u) hipLaunchKernelGGL((mc::sk::sk_query_kernel<SH, 27>), g, b, 0, st, m);                  \
            else hipLaunchKernelGGL((mc::sk::sk_query_kernel<SH, 0>), g, b, 0, st, m);                                    \
        } while (0)
        if (c->info.n_parts > 1) MC_SK_LAUNCH(mc::mz::MZ_LINES); else MC_SK_LAUNCH(mc::mz::MZ_ALL);
#undef MC_SK_LAUNCH
    } else if (c->d_mz_lines) {
        mc::mz::MzArgs m{};
        m.q = a; m.lines = c->d_mz_lines; m.extra = c->d_mz_extra;
        m.n_lines = c->mz_n_local; m.part = c->mz_part; m.n_parts = c->mz_n_parts;
        m.m = c->mz_m;
        // canonical k-mers are below 4^k: the floating-point remainder needs k-mer / HTSIZE < 2^32
        const bool fp_ok = c->htsize > 1024 && c->htsize < (1ull << 32) &&
                           (c->k < 32 ? ((unsigned __int128)1 << (2 * c->k)) <= ((unsigned __int128)c->htsize << 32) : false);
        m.inv_htsize = fp_ok ? 1.0 / (double)c->htsize : 0.0;
        const int shard = c->info.n_parts > 1 ? mc::mz::MZ_LINES
                        : (c->info.shard_begin != 0 || c->info.shard_end != c->htsize) ? mc::mz::MZ_BUCKETS : mc::mz::MZ_ALL;
        // k = 31 (cuCLARK's default) and k = 27 (cuCLARK-l) have kernels with the k-mer length compiled in
#define MC_MZ_LAUNCH(SH)                                                                                                   \
        do {                                                                                                               \
            if (c->k == 31u && c->mz_m == mc::mz::mmer_len(31u)) hipLaunchKernelGGL((mc::mz::mz_query_kernel<SH, 31>), g, b, 0, st, m);      \
            else if (c->k == 27u && c->mz_m == mc::mz::mmer_len(27u)) hipLaunchKernelGGL((mc::mz::mz_query_kernel<SH, 27>), g, b, 0, st, m); \
            else hipLaunchKernelGGL((mc::mz::mz_query_kernel<SH, 0>), g, b, 0, st, m);                                     \
        } while (0)
        switch (shard) {
        case mc::mz::MZ_LINES:   MC_MZ_LAUNCH(mc::mz::MZ_LINES); break;
        case mc::mz::MZ_BUCKETS: MC_MZ_LAUNCH(mc::mz::MZ_BUCKETS); break;
        default:                 MC_MZ_LAUNCH(mc::mz::MZ_ALL); break;
        }
#undef MC_MZ_LAUNCH
    } else if (!c->wide) {
        if (c->info.line_bytes == 64) hipLaunchKernelGGL((mc::query_kernel<64, false>), g, b, 0, st, a);
        else                          hipLaunchKernelGGL((mc::query_kernel<128, false>), g, b, 0, st, a);
    } else {
        if (c->info.line_bytes == 64) hipLaunchKernelGGL((mc::query_kernel<64, true>), g, b, 0, st, a);
        else                          hipLaunchKernelGGL((mc::query_kernel<128, true>), g, b, 0, st, a);
    }
    HIPCHK(hipGetLastError());
    c->stats.reads += n_reads;
    c->stats.kernel_launches++;
    return MC_OK;
}

int launch_merge_result(mc_ctx *c, const uint16_t *const *d_srcs, uint32_t n_srcs, uint64_t n_reads,
                        uint16_t *d_out_rows, uint16_t *d_final, hipStream_t st)
{
    if (n_srcs < 1 || n_srcs > (uint32_t)mc::MERGE_MAX_SRCS) return fail(MC_EINVAL, "1 to 16 row sources");
    if (n_reads == 0) return MC_OK;
    mc::RowSrcs S{};
    for (uint32_t i = 0; i < n_srcs; i++) { if (!d_srcs[i]) return fail(MC_EINVAL, "NULL row source"); S.p[i] = d_srcs[i]; }
    S.n = n_srcs;
    const uint32_t row_len = 2 * c->maxhits + 2;
    const uint32_t g = (uint32_t)((n_reads + 255) / 256);
    hipLaunchKernelGGL(mc::merge_result_kernel, dim3(g), dim3(256), 0, st, S, row_len, n_reads, d_out_rows, d_final);
    HIPCHK(hipGetLastError());
    return MC_OK;
}

// One batch through the three queues of a context: H2D on s_in, kernel on streams[0], D2H on s_out.  The two device
// slots alternate by submission; a slot's next H2D waits for its previous kernel, its next kernel for its previous
// D2H.  `done` is recorded behind the D2H.
int submit_batch(mc_ctx *c, const uint32_t *h_ptr, const uint16_t *h_con, uint16_t *h_final, uint16_t *h_rows,
                 uint64_t n_reads, uint64_t n_con, uint32_t flags, hipEvent_t done)
{
    const int si = (int)(c->n_submitted++ & 1u);
    Slot &s = c->slots[si];
    const size_t row_len = 2 * (size_t)c->maxhits + 2;
    if (n_reads) {
        HIPCHK(hipStreamWaitEvent(c->s_in, c->ev_k[si], 0));
        HIPCHK(hipMemcpyAsync(s.d_ptr, h_ptr, (n_reads + 1) * 4, hipMemcpyHostToDevice, c->s_in));
        if (n_con) HIPCHK(hipMemcpyAsync(s.d_con, h_con, n_con * 2, hipMemcpyHostToDevice, c->s_in));
        HIPCHK(hipEventRecord(c->ev_in[si], c->s_in));
        HIPCHK(hipStreamWaitEvent(c->streams[0], c->ev_in[si], 0));
        HIPCHK(hipStreamWaitEvent(c->streams[0], c->ev_out[si], 0));
        const int rc = launch_query(c, s.d_ptr, s.d_con, n_reads, n_con, flags, s.d_final, s.d_rows, c->streams[0]);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_k[si], c->streams[0]));
        HIPCHK(hipStreamWaitEvent(c->s_out, c->ev_k[si], 0));
        if (flags & MC_F_FINAL)
            HIPCHK(hipMemcpyAsync(h_final, s.d_final, n_reads * MC_FINAL_ROW * 2, hipMemcpyDeviceToHost, c->s_out));
        if (flags & MC_F_ROWS)
            HIPCHK(hipMemcpyAsync(h_rows, s.d_rows, n_reads * row_len * 2, hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(hipEventRecord(c->ev_out[si], c->s_out));
    }
    HIPCHK(hipEventRecord(done, c->s_out));
    return MC_OK;
}

// ---- database files, streamed in bucket order ------------------------------------
namespace {
bool pread_all(int fd, void *dst, size_t n, uint64_t off)
{
    char *p = (char *)dst;
    while (n) {
        ssize_t g = pread(fd, p, n, (off_t)off);
        if (g <= 0) return false;
        p += g; off += (uint64_t)g; n -= (size_t)g;
    }
    return true;
}
}

DbFileStream::~DbFileStream()
{
    if (fs >= 0) close(fs);
    if (fk >= 0) close(fk);
    if (fl >= 0) close(fl);
}

int DbFileStream::open(const char *base_path, int kb, uint32_t smp, uint64_t ht, uint64_t b, uint64_t e)
{
    base = base_path; key_bytes = kb; sampling = smp; htsize = ht; sb = b; se = e;
    fs = ::open((base + ".sz").c_str(), O_RDONLY);
    fk = ::open((base + ".ky").c_str(), O_RDONLY);
    fl = ::open((base + ".lb").c_str(), O_RDONLY);
    if (fs < 0 || fk < 0 || fl < 0) return fail(MC_EIO, "Failed to open " + base + ".sz/.ky/.lb");
    // bucket sizes of the whole table (the sampling counter runs over all non-empty
    // buckets, reference CuClarkDB.cu:503-513)
    sz.resize(htsize);
    // the 1.6 GB of bucket sizes of a full table: read and summed in slices by several threads (one thread: 0.5 s for the
    // read, 1.5 s for the sums -- a quarter of what loading the table takes once its files are read only once)
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>(8, htsize >> 24));
    std::vector<uint64_t> below(nt, 0), inside(nt, 0);
    std::atomic<bool> bad{false};
    const bool all = sampling <= 1;
    auto slice = [&](int t) {
        const uint64_t i0 = htsize * (uint64_t)t / nt, i1 = htsize * (uint64_t)(t + 1) / nt;
        if (!pread_all(fs, sz.data() + i0, i1 - i0, i0)) { bad = true; return; }
        if (!all) return;
        uint64_t lo = 0, in = 0;
        for (uint64_t i = i0; i < i1; i++) { const uint64_t v = sz[i]; if (i < sb) lo += v; else if (i < se) in += v; }
        below[t] = lo; inside[t] = in;
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(slice, t);
        slice(0);
        for (auto &t : th) t.join();
    }
    if (bad) return fail(MC_EIO, base + ".sz is shorter than htsize");
    file_k0 = 0; n_keys_kept = 0;
    if (all) {
        for (int t = 0; t < nt; t++) { file_k0 += below[t]; n_keys_kept += inside[t]; }
        return MC_OK;
    }
    // -s sampling: the counter runs over all non-empty buckets in order (reference CuClarkDB.cu:503-513)
    fsz = sz;
    uint64_t nonzero = 0;
    for (uint64_t i = 0; i < htsize; i++) {
        if (sz[i] == 0) continue;
        nonzero++;
        const bool kp = (nonzero % sampling) == 0;
        if (i < sb) file_k0 += sz[i];
        if (!kp) sz[i] = 0;
        else if (i >= sb && i < se) n_keys_kept += sz[i];
    }
    return MC_OK;
}

int DbFileStream::pass(const ChunkFn &f)
{
    const uint64_t CH = 1ull << 24;   // buckets per step
    const std::vector<uint8_t> &file_sz = fsz.empty() ? sz : fsz;
    std::vector<uint8_t> kbuf, lbuf;
    uint64_t fpos = file_k0;
    for (uint64_t b0 = sb; b0 < se; b0 += CH) {
        const uint64_t b1 = std::min(se, b0 + CH);
        uint64_t nfile = 0;
        for (uint64_t i = b0; i < b1; i++) nfile += file_sz[i];
        kbuf.resize(nfile * (size_t)key_bytes); lbuf.resize(nfile * 2);
        if (nfile && (!pread_all(fk, kbuf.data(), kbuf.size(), fpos * (uint64_t)key_bytes) ||
                      !pread_all(fl, lbuf.data(), lbuf.size(), fpos * 2)))
            return fail(MC_EIO, base + ".ky/.lb shorter than the bucket sizes say");
        uint64_t nkeep = nfile;
        if (!fsz.empty()) {                      // drop the unsampled buckets
            uint64_t r = 0, w = 0;
            for (uint64_t i = b0; i < b1; i++) {
                const uint64_t n = file_sz[i];
                if (n && sz[i]) {
                    memmove(kbuf.data() + w * key_bytes, kbuf.data() + r * key_bytes, n * key_bytes);
                    memmove(lbuf.data() + w * 2, lbuf.data() + r * 2, n * 2);
                    w += n;
                }
                r += n;
            }
            nkeep = w;
        }
        const int rc = f(sz.data() + b0, (const void *)kbuf.data(), (const uint16_t *)lbuf.data(), nkeep, b0, b1);
        if (rc != MC_OK) return rc;
        fpos += nfile;
    }
    return MC_OK;
}

void DbFileStream::plan()
{
    const uint64_t CH = 1ull << 24;   // buckets per chunk
    const std::vector<uint8_t> &file_sz = fsz.empty() ? sz : fsz;
    chunks.clear(); max_nfile = 0; max_nb = 0;
    for (uint64_t b0 = sb; b0 < se; b0 += CH) chunks.push_back(Chunk{b0, std::min(se, b0 + CH), 0, 0});
    // the k-mers of every chunk: summed by several threads, chunk by chunk
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t j; (j = next.fetch_add(1)) < chunks.size();) {
            uint64_t nfile = 0;
            for (uint64_t i = chunks[j].b0; i < chunks[j].b1; i++) nfile += file_sz[i];
            chunks[j].nfile = nfile;
        }
    };
    {
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>(8, std::max<size_t>(1, chunks.size()));
        for (int t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    uint64_t fpos = file_k0;
    for (auto &C : chunks) {
        C.fpos = fpos; fpos += C.nfile;
        max_nfile = std::max(max_nfile, C.nfile); max_nb = std::max(max_nb, C.b1 - C.b0);
    }
}

int DbFileStream::read(size_t i, uint8_t *sz_out, void *keys_out, uint16_t *labels_out, uint64_t *n_kept, int threads)
{
    const Chunk &C = chunks[i];
    memcpy(sz_out, sz.data() + C.b0, C.b1 - C.b0);
    // byte ranges of the two files, cut into slices of <= 32 MB, dealt to the threads
    struct Slice { int fd; char *dst; size_t n; uint64_t off; };
    std::vector<Slice> sl;
    const size_t SL = 32u << 20;
    auto cut = [&](int fd, void *dst, size_t bytes, uint64_t off) {
        for (size_t at = 0; at < bytes; at += SL) sl.push_back(Slice{fd, (char *)dst + at, std::min(SL, bytes - at), off + at});
    };
    cut(fk, keys_out, C.nfile * (size_t)key_bytes, C.fpos * (uint64_t)key_bytes);
    cut(fl, labels_out, C.nfile * 2, C.fpos * 2);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&]() {
        for (size_t j; (j = next.fetch_add(1)) < sl.size();)
            if (!pread_all(sl[j].fd, sl[j].dst, sl[j].n, sl[j].off)) bad = true;
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, sl.size()));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) return fail(MC_EIO, base + ".ky/.lb shorter than the bucket sizes say");
    uint64_t nkeep = C.nfile;
    if (!fsz.empty()) {                      // drop the unsampled buckets
        uint64_t r = 0, w = 0;
        char *kb = (char *)keys_out; char *lb = (char *)labels_out;
        for (uint64_t b = C.b0; b < C.b1; b++) {
            const uint64_t n = fsz[b];
            if (n && sz[b]) {
                memmove(kb + w * key_bytes, kb + r * key_bytes, n * key_bytes);
                memmove(lb + w * 2, lb + r * 2, n * 2);
                w += n;
            }
            r += n;
        }
        nkeep = w;
    }
    *n_kept = nkeep;
    return MC_OK;
}

// primary lines of one part (every part has the same number); 0 = more than the 32-bit line index of a context holds
uint64_t lines_per_part(uint64_t n_keys_total, uint32_t n_parts, double fill)
{
    if (n_parts < 1 || fill <= 0.0) return 0;
    const uint64_t total = (uint64_t)((double)n_keys_total / fill) + 1024;
    const uint64_t per = (total + n_parts - 1) / n_parts;
    return per >= MZ_MAX_LINES ? 0 : per;
}

// Extra lines per primary line the budget assumes at `fill` k-mers per line.  Round 3 took the shares of the genome-shaped
// table (12 % at 3-4 per line ... 90 % at 12) and ALLOCATED them: 28 GB behind the headline table, which used 1.7.  Round 4:
// the extra lines are backed exactly (mc_ctx::Vmm) and tables whose k-mers clump around their minimizers go to the
// super-k-mer index (MC_INDEX=auto), so the estimate is that of k-mers that spread like a Poisson process over 12-slot lines
// (P(more than 12 | fill) = 0.001 / 0.09 / 0.9 / 6.4 / 21 / 42 % at 3.5 / 5 / 7 / 8 / 10 / 12 per line), with a margin; a table
// that needs more gets it while the card has it (MC_ENOMEM at mc_index_next_pass otherwise: MC_MZ_FILL).
double extra_share(double fill)
{
    return fill <= 4.0 ? 0.02 : fill <= 5.0 ? 0.03 : fill <= 6.0 ? 0.05 : fill <= 7.0 ? 0.08 : fill <= 8.0 ? 0.14 : fill <= 10.0 ? 0.40 : 0.8;
}

// HBM one context needs for its share of a minimizer index at `fill` k-mers per line: lines, extra lines (the
// share measured on a genome-shaped table, which overflows more than a random one), build counters
uint64_t index_bytes(uint64_t n_keys_total, uint32_t n_parts, double fill)
{
    const double extra = extra_share(fill);
    const double lines = ((double)n_keys_total / fill + 1024.0) / (double)n_parts;
    return (uint64_t)(lines * 128.0 * (1.0 + extra) + lines * 4.0) + (2ull << 30);
}

double choose_fill(uint64_t n_keys_total, uint32_t n_parts, uint64_t free_bytes)
{
    auto fits = [&](double f) {
        return lines_per_part(n_keys_total, n_parts, f) != 0 && index_bytes(n_keys_total, n_parts, f) + MZ_RESERVE_BYTES <= free_bytes;
    };
    // (round 3: from 3 per line, not 4 -- where the card has the room, the genome-shaped table gains 4 % at 3 and the
    // headline table 1 % at 3.5: fewer lines on which related genomes and background k-mers meet)
    for (double f = 3.0; f < 8.0; f += 0.5)
        if (fits(f)) return f;
    // past 8 the lines are mostly full and the chains long: slower, but the table stays on the card
    for (double f = 8.0; f < 12.0; f += 1.0)
        if (fits(f)) return f;
    return 12.0;
}

uint32_t min_parts(uint64_t n_keys_total, uint32_t max_parts, uint64_t free_bytes, double max_fill)
{
    for (uint32_t s = 1; s <= max_parts; s++)
        if (lines_per_part(n_keys_total, s, max_fill) != 0 && index_bytes(n_keys_total, s, max_fill) + MZ_RESERVE_BYTES <= free_bytes) return s;
    return 0;
}

bool minimizer_index_possible(const mc_ctx *c, uint64_t n_keys_total) { return c->index_mode >= 1 && mz_eligible(c, n_keys_total); }

// The files -> the index of every member, as a pipeline.  A reader thread fills one of two pinned chunk buffers
// (parallel preads) while the other is in flight; each chunk is uploaded ONCE per device on that device's copy stream
// into one of two staging sets -- members that share a device share the upload -- and every member queues its build
// kernels behind the upload on its own stream: no allocation, no host round trip and no synchronisation per chunk
// and member (round 2 did all three, member after member: an N-member group loaded in 2 N sequential passes).
int load_streamed(mc_ctx *const *ctxs, uint32_t n, DbFileStream &F, uint32_t n_parts, uint32_t part0, double fill)
{
    int rc = MC_OK;
    const bool loud = getenv("MC_LOAD_VERBOSE") != nullptr;
    auto now_s = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double t_start = now_s();
    auto lap = [&](const char *what) { if (loud) fprintf(stderr, "libmcclark: %6.2f s  %s\n", now_s() - t_start, what); };
    // ONE pass over the files when the card has the room (round 4): the chunks of the first build pass stay in HBM -- one
    // copy per device, 6 bytes per k-mer + 1 per bucket: 40 GB for the headline table -- and the second pass is fed from
    // there instead of reading and uploading the 41 GB of files again (8.3-9.2 s for the two passes in round 3).  The
    // room comes out of the index's budget: taken only if the fill stays within one k-mer per line of what it would be.
    const uint64_t resident_bytes = (F.se - F.sb) + F.n_keys_kept * (uint64_t)(F.key_bytes + 2) + (4ull << 20);
    bool one_pass = !getenv("MC_LOAD_TWO_PASSES");
    uint64_t free_min = ~0ull;
    uint32_t sharing_max = 1;
    for (uint32_t i = 0; i < n; i++) {
        size_t fr = 0, tot = 0;
        if (set_dev(ctxs[i]) != MC_OK || hipMemGetInfo(&fr, &tot) != hipSuccess) continue;
        uint32_t sharing = 0;
        for (uint32_t o = 0; o < n; o++) sharing += ctxs[o]->device == ctxs[i]->device ? 1u : 0u;
        free_min = std::min<uint64_t>(free_min, fr / sharing);
        sharing_max = std::max(sharing_max, sharing);
    }
    if (const char *e = getenv("MC_GROUP_HBM_BYTES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v) free_min = std::min(free_min, v); }
    {
        const uint64_t take = resident_bytes / sharing_max + (1ull << 30);
        const double f0 = fill > 0.0 ? fill : choose_fill(F.n_keys_kept, n_parts, free_min);
        const double f1 = fill > 0.0 ? fill : (free_min > take ? choose_fill(F.n_keys_kept, n_parts, free_min - take) : 99.0);
        one_pass = one_pass && free_min > take && f1 <= std::max(4.0, f0 + 1.0) &&
                   index_bytes(F.n_keys_kept, n_parts, f1) + MZ_RESERVE_BYTES <= free_min - take;
        if (one_pass) free_min -= take;
    }
    if (fill > 0.0) {
        for (uint32_t i = 0; i < n; i++) ctxs[i]->fill_hint = fill;
    } else if (!getenv("MC_MZ_FILL")) {
        // one fill for all members: what the member with the least free HBM can afford
        const double f = choose_fill(F.n_keys_kept, n_parts, free_min);
        for (uint32_t i = 0; i < n; i++) ctxs[i]->fill_hint = f;
    }
    auto abort_all = [&]() { const std::string keep = g_err; for (uint32_t i = 0; i < n; i++) { (void)hipSetDevice(ctxs[i]->device); free_db(ctxs[i]); index_abort(ctxs[i]); } g_err = keep; };
    for (uint32_t i = 0; i < n && rc == MC_OK; i++) {
        rc = set_dev(ctxs[i]);
        ctxs[i]->group_loading = true; ctxs[i]->auto_decision = 0;
        if (rc == MC_OK) rc = index_begin(ctxs[i], F.n_keys_kept, (part0 + i) % n_parts, n_parts);
    }
    if (rc != MC_OK) { abort_all(); return rc; }

    F.plan();
    const size_t kb = (size_t)F.key_bytes;
    const size_t cap_k = (size_t)std::max<uint64_t>(F.max_nfile, 1), cap_b = (size_t)std::max<uint64_t>(F.max_nb, 1);
    // per distinct device: copy stream, two staging sets, "copied" events; per member: "built" events
    struct Dev { int device; hipStream_t cs = nullptr; uint8_t *d_sz[2] = {nullptr, nullptr}; char *d_keys[2] = {nullptr, nullptr};
                 uint16_t *d_labels[2] = {nullptr, nullptr}; hipEvent_t copied[2] = {nullptr, nullptr};
                 uint8_t *r_sz = nullptr; char *r_keys = nullptr; uint16_t *r_labels = nullptr; };      // the whole table, kept for the second pass
    std::vector<Dev> devs;
    std::vector<int> dev_of(n);
    std::vector<hipEvent_t> built((size_t)n * 2, nullptr);
    uint8_t *h_buf[2] = {nullptr, nullptr};
    const size_t h_bytes = ((cap_b + 255) & ~(size_t)255) + cap_k * kb + cap_k * 2 + 256;
    auto h_sz = [&](int s) { return h_buf[s]; };
    auto h_keys = [&](int s) { return (void *)(h_buf[s] + ((cap_b + 255) & ~(size_t)255)); };
    auto h_labels = [&](int s) { return (uint16_t *)(h_buf[s] + ((cap_b + 255) & ~(size_t)255) + ((cap_k * kb + 1) & ~(size_t)1)); };
    auto cleanup = [&]() {
        for (auto &D : devs) {
            (void)hipSetDevice(D.device);
            (void)hipDeviceSynchronize();
            for (int s2 = 0; s2 < 2; s2++) {
                if (D.d_sz[s2]) (void)hipFree(D.d_sz[s2]);
                if (D.d_keys[s2]) (void)hipFree(D.d_keys[s2]);
                if (D.d_labels[s2]) (void)hipFree(D.d_labels[s2]);
                if (D.copied[s2]) (void)hipEventDestroy(D.copied[s2]);
            }
            if (D.cs) (void)hipStreamDestroy(D.cs);
            if (D.r_sz) (void)hipFree(D.r_sz);
            if (D.r_keys) (void)hipFree(D.r_keys);
            if (D.r_labels) (void)hipFree(D.r_labels);
            D.r_sz = nullptr; D.r_keys = nullptr; D.r_labels = nullptr;
        }
        for (uint32_t i = 0; i < n; i++)
            for (int s2 = 0; s2 < 2; s2++)
                if (built[(size_t)i * 2 + s2]) { (void)hipSetDevice(ctxs[i]->device); (void)hipEventDestroy(built[(size_t)i * 2 + s2]); }
        for (int s2 = 0; s2 < 2; s2++) if (h_buf[s2]) (void)hipHostFree(h_buf[s2]);
    };
    auto setup = [&]() -> int {
        for (uint32_t i = 0; i < n; i++) {
            int at = -1;
            for (size_t d = 0; d < devs.size(); d++) if (devs[d].device == ctxs[i]->device) at = (int)d;
            if (at < 0) { Dev D; D.device = ctxs[i]->device; devs.push_back(D); at = (int)devs.size() - 1; }
            dev_of[i] = at;
        }
        for (auto &D : devs) {
            HIPCHK(hipSetDevice(D.device));
            HIPCHK(hipStreamCreateWithFlags(&D.cs, hipStreamNonBlocking));
            for (int s2 = 0; s2 < 2; s2++) {
                if (hipMalloc(&D.d_sz[s2], cap_b) != hipSuccess || hipMalloc(&D.d_keys[s2], cap_k * kb) != hipSuccess ||
                    hipMalloc(&D.d_labels[s2], cap_k * 2) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(MC_ENOMEM, "not enough HBM for the loader's staging chunks next to the minimizer lines");
                }
                HIPCHK(hipEventCreateWithFlags(&D.copied[s2], hipEventDisableTiming));
            }
            if (one_pass && (hipMalloc(&D.r_sz, (size_t)(F.se - F.sb) + 16) != hipSuccess ||
                             hipMalloc(&D.r_keys, (size_t)std::max<uint64_t>(F.n_keys_kept, 1) * kb) != hipSuccess ||
                             hipMalloc(&D.r_labels, (size_t)std::max<uint64_t>(F.n_keys_kept, 1) * 2) != hipSuccess)) {
                (void)hipGetLastError();
                one_pass = false;                       // (no room after all: two passes; what was allocated is released by cleanup)
            }
        }
        for (uint32_t i = 0; i < n; i++) {
            HIPCHK(hipSetDevice(ctxs[i]->device));
            for (int s2 = 0; s2 < 2; s2++) HIPCHK(hipEventCreateWithFlags(&built[(size_t)i * 2 + s2], hipEventDisableTiming));
        }
        for (int s2 = 0; s2 < 2; s2++) HIPCHK(hipHostMalloc((void **)&h_buf[s2], h_bytes, hipHostMallocPortable));
        return MC_OK;
    };
    rc = setup();
    int n_threads = 12;
    if (const char *e = getenv("MC_LOAD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_threads = v; }

    std::vector<uint64_t> kept_off(F.chunks.size() + 1, 0);      // k-mers kept before chunk i (one pass: where it sits in the resident arrays)
    lap("counters, chunk plan, pinned and staging buffers allocated");
    if (loud) fprintf(stderr, "libmcclark: loading %.2fe9 k-mers for %u member(s): %s over the files (%d reader threads)\n", (double)F.n_keys_kept / 1e9, n,
                      one_pass ? "ONE pass" : "two passes", n_threads);
    for (int pass = 0; pass < 2 && rc == MC_OK; pass++) {
        if (pass == 1) lap("first pass over the files done, lines allocated");
        if (pass == 1 && one_pass) {
            // the second pass from the chunks that stayed in HBM
            for (size_t i = 0; i < F.chunks.size() && rc == MC_OK; i++) {
                const DbFileStream::Chunk &C = F.chunks[i];
                for (uint32_t m = 0; m < n && rc == MC_OK; m++) {
                    Dev &D = devs[dev_of[m]];
                    rc = set_dev(ctxs[m]);
                    if (rc == MC_OK)
                        rc = index_add_device(ctxs[m], D.r_sz + (C.b0 - F.sb), D.r_keys + kept_off[i] * kb, F.key_bytes, D.r_labels + kept_off[i],
                                              kept_off[i + 1] - kept_off[i], C.b0, C.b1);
                }
            }
            lap("second pass queued (from HBM)");
            // the raw arrays have done their work: released before the indexes take their final shape
            for (auto &D : devs) {
                if (hipSetDevice(D.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); continue; }
                (void)hipFree(D.r_sz); (void)hipFree(D.r_keys); (void)hipFree(D.r_labels);
                D.r_sz = nullptr; D.r_keys = nullptr; D.r_labels = nullptr;
            }
            for (uint32_t i = 0; i < n && rc == MC_OK; i++) {
                rc = set_dev(ctxs[i]);
                ctxs[i]->sk_d_hint = i ? ctxs[0]->sk_d : 0;
                if (i == 0) lap("raw arrays released");
                if (rc == MC_OK) rc = index_end(ctxs[i]);
            }
            break;
        }
        // reader: chunk i into host buffer i % 2
        std::mutex mu;
        std::condition_variable cv;
        bool is_free[2] = {true, true}, is_ready[2] = {false, false};
        uint64_t nk[2] = {0, 0};
        int reader_rc = MC_OK;
        std::string reader_err;
        bool stop = false;
        std::thread reader([&]() {
            for (size_t i = 0; i < F.chunks.size(); i++) {
                const int s2 = (int)(i & 1);
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return is_free[s2] || stop; }); if (stop) return; is_free[s2] = false; }
                uint64_t kept = 0;
                const int r = F.read(i, h_sz(s2), h_keys(s2), h_labels(s2), &kept, n_threads);
                { std::lock_guard<std::mutex> lk(mu); nk[s2] = kept; is_ready[s2] = true; if (r != MC_OK) { reader_rc = r; reader_err = g_err; } }
                cv.notify_all();
                if (r != MC_OK) return;
            }
        });
        for (size_t i = 0; i < F.chunks.size() && rc == MC_OK; i++) {
            const int s2 = (int)(i & 1);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return is_ready[s2]; }); is_ready[s2] = false; if (reader_rc != MC_OK) { rc = reader_rc; g_err = reader_err; } }
            if (rc != MC_OK) break;
            const DbFileStream::Chunk &C = F.chunks[i];
            const uint64_t k_n = nk[s2];
            if (pass == 0) kept_off[i + 1] = kept_off[i] + k_n;
            auto enqueue = [&]() -> int {
                for (size_t d = 0; d < devs.size(); d++) {
                    Dev &D = devs[d];
                    HIPCHK(hipSetDevice(D.device));
                    for (uint32_t m = 0; m < n; m++)          // the staging set was read by the kernels of chunk i - 2
                        if (dev_of[m] == (int)d && i >= 2) HIPCHK(hipStreamWaitEvent(D.cs, built[(size_t)m * 2 + s2], 0));
                    HIPCHK(hipMemcpyAsync(D.d_sz[s2], h_sz(s2), C.b1 - C.b0, hipMemcpyHostToDevice, D.cs));
                    if (k_n) {
                        HIPCHK(hipMemcpyAsync(D.d_keys[s2], h_keys(s2), k_n * kb, hipMemcpyHostToDevice, D.cs));
                        HIPCHK(hipMemcpyAsync(D.d_labels[s2], h_labels(s2), k_n * 2, hipMemcpyHostToDevice, D.cs));
                    }
                    if (one_pass && pass == 0) {          // and a copy that stays, device to device
                        HIPCHK(hipMemcpyAsync(D.r_sz + (C.b0 - F.sb), D.d_sz[s2], C.b1 - C.b0, hipMemcpyDeviceToDevice, D.cs));
                        if (k_n) {
                            HIPCHK(hipMemcpyAsync(D.r_keys + kept_off[i] * kb, D.d_keys[s2], k_n * kb, hipMemcpyDeviceToDevice, D.cs));
                            HIPCHK(hipMemcpyAsync(D.r_labels + kept_off[i], D.d_labels[s2], k_n * 2, hipMemcpyDeviceToDevice, D.cs));
                        }
                    }
                    HIPCHK(hipEventRecord(D.copied[s2], D.cs));
                }
                for (uint32_t m = 0; m < n; m++) {
                    Dev &D = devs[dev_of[m]];
                    int r = set_dev(ctxs[m]);
                    if (r != MC_OK) return r;
                    HIPCHK(hipStreamWaitEvent(ctxs[m]->streams[0], D.copied[s2], 0));
                    r = index_add_device(ctxs[m], D.d_sz[s2], D.d_keys[s2], F.key_bytes, D.d_labels[s2], k_n, C.b0, C.b1);
                    if (r != MC_OK) return r;
                    HIPCHK(hipEventRecord(built[(size_t)m * 2 + s2], ctxs[m]->streams[0]));
                }
                for (auto &D : devs) { HIPCHK(hipSetDevice(D.device)); HIPCHK(hipEventSynchronize(D.copied[s2])); }    // the host buffer is free again
                return MC_OK;
            };
            rc = enqueue();
            { std::lock_guard<std::mutex> lk(mu); is_free[s2] = true; }
            cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        reader.join();
        for (uint32_t i = 0; i < n && rc == MC_OK; i++) {
            rc = set_dev(ctxs[i]);
            // (super-k-mer records: the first member's merge factor is everybody's -- the parts of one table must agree
            // on the line count, and replicas may as well)
            if (pass == 1) ctxs[i]->sk_d_hint = i ? ctxs[0]->sk_d : 0;
            if (pass == 0 && i) ctxs[i]->auto_decision = ctxs[0]->auto_decision;     // MC_INDEX=auto: one choice for the group
            if (rc == MC_OK) rc = pass == 0 ? index_next_pass(ctxs[i]) : index_end(ctxs[i]);
        }
    }
    lap("index built");
    const std::string keep = g_err;
    cleanup();
    g_err = keep;
    lap("loader's buffers released");
    for (uint32_t i = 0; i < n; i++) { ctxs[i]->group_loading = false; ctxs[i]->auto_decision = 0; ctxs[i]->sk_d_hint = 0; }
    if (rc != MC_OK) abort_all();
    return rc;
}


// The bucket-line table from the files: the raw arrays of the range go to the device whole.  The fallback of every
// loader when the minimizer lines do not fit (`fallback` says so in mc_db_info).
int load_lines_from_files(mc_ctx *c, DbFileStream &F, uint64_t sb, uint64_t se, bool fallback)
{
    const int key_bytes = F.key_bytes;
    int rc = MC_OK;
    const uint64_t nb = se - sb, kept = F.n_keys_kept;
    Scope tmp;
    uint8_t *d_sz = nullptr; char *d_raw = nullptr; uint16_t *d_labels = nullptr;
    TMP_MALLOC(tmp, d_sz, nb ? nb : 1);
    TMP_MALLOC(tmp, d_raw, (kept ? kept : 1) * (size_t)key_bytes);
    TMP_MALLOC(tmp, d_labels, (kept ? kept : 1) * 2);
    uint64_t dpos = 0;
    rc = F.pass([&](const uint8_t *, const void *keys, const uint16_t *labels, uint64_t nk, uint64_t, uint64_t) {
        if (nk) {
            HIPCHK(hipMemcpy(d_raw + dpos * (size_t)key_bytes, keys, nk * (size_t)key_bytes, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d_labels + dpos, labels, nk * 2, hipMemcpyHostToDevice));
        }
        dpos += nk;
        return (int)MC_OK;
    });
    if (rc) return rc;
    if (dpos != kept) return fail(MC_EINVAL, "internal: kept-key count mismatch");
    HIPCHK(hipMemcpy(d_sz, F.sz.data() + sb, nb, hipMemcpyHostToDevice));
    void *d_keys = nullptr; bool owned = false;
    tmp.forget(d_raw);                       // convert_keys takes it over (frees or returns it)
    rc = convert_keys(c, d_raw, key_bytes, kept, true, &d_keys, &owned);
    if (rc != MC_OK) return rc;
    tmp.add(d_keys);
    return relayout(c, d_sz, d_keys, d_labels, kept, sb, se, fallback);
}
} // namespace mcint

using mcint::launch_query;

extern "C" {

void mc_set_last_error_(const char *msg) { g_err = msg ? msg : ""; }     // for mc_build.hip

const char *mc_last_error(void) { return g_err.c_str(); }
int mc_api_version(void) { return MC_API_VERSION; }

int mc_device_count(int *count)
{
    if (!count) return fail(MC_EINVAL, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return fail(MC_ENODEVICE, "hipGetDeviceCount failed"); }
    *count = n;
    return MC_OK;
}

int mc_open(mc_ctx **out, int device, uint32_t k, uint64_t htsize, uint32_t num_targets, uint32_t maxhits)
{
    if (!out) return fail(MC_EINVAL, "out is NULL");
    *out = nullptr;
    if (k < 2 || k > 32) return fail(MC_EINVAL, "k must be in [2,32]");
    if (htsize < 2) return fail(MC_EINVAL, "htsize must be >= 2");
    if (maxhits < 1 || maxhits > 63) return fail(MC_EINVAL, "maxhits must be in [1,63]");
    if (htsize >= 0xFFFFFFFFull) return fail(MC_EINVAL, "htsize must be < 2^32-1 (bucket indices are 32-bit, reference ITYPE)");
    // narrow lines need every quotient < 0xFFFFFFFF (the sentinel); otherwise 64-bit keys
    const unsigned __int128 maxkmer = k == 32 ? (unsigned __int128)~0ull : (((unsigned __int128)1 << (2 * k)) - 1);
    const bool wide = maxkmer / htsize >= 0xFFFFFFFFull;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(MC_ENODEVICE, "no HIP device visible");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) return fail(MC_ENODEVICE, "device index out of range");
    mc_ctx *c = new mc_ctx();
    c->device = device; c->k = k; c->htsize = htsize; c->num_targets = num_targets; c->maxhits = maxhits;
    c->div = mc::make_div(htsize);
    c->wide = wide;
    if (const char *e = getenv("MC_INDEX")) c->index_mode = strcmp(e, "lines") == 0 ? 0 : strcmp(e, "skm") == 0 ? 2 : strcmp(e, "auto") == 0 ? 3 : 1;      // default: auto
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) { c->n_cu = prop.multiProcessorCount; }
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
    // (The runtime multiplexes the streams of a process over GPU_MAX_HW_QUEUES = 4 hardware queues; when two of the
    // three queues below share one, a barrier packet of one holds the other back and the streamed rate is ~680
    // instead of ~1000 Mreads/s.  Stream priorities did not separate them (measured); bin/cuCLARK and bench.py set
    // GPU_MAX_HW_QUEUES=8 before the runtime starts, which does.)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc(&c->d_over, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->d_over, 0, sizeof(unsigned long long));
    if (e != hipSuccess) { delete c; return fail(MC_EHIP, std::string("mc_open: ") + hipGetErrorString(e)); }
    *out = c;
    return MC_OK;
}

int mc_close(mc_ctx *c)
{
    if (!c) return MC_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    mc_free_batches(c);
    mcint::text_release(c);
    free_db(c);
    index_abort(c);
    if (c->d_over) (void)hipFree(c->d_over);
    for (int i = 0; i < 2; i++) if (c->streams[i]) (void)hipStreamDestroy(c->streams[i]);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    for (int i = 0; i < 2; i++) {
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_k[i]) (void)hipEventDestroy(c->ev_k[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    delete c;
    return MC_OK;
}

int mc_load_db_device(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, int key_bytes, const uint16_t *d_labels,
                      uint64_t n_keys, uint64_t sb, uint64_t se)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    int rc = set_dev(c); if (rc) return rc;
    rc = norm_shard(c, sb, se); if (rc) return rc;
    Scope tmp;
    void *keys = nullptr; bool owned = false;
    rc = convert_keys(c, const_cast<void *>(d_keys), key_bytes, n_keys, false, &keys, &owned);
    if (rc) return rc;
    if (owned) tmp.add(keys);
    return relayout(c, d_sz, keys, d_labels, n_keys, sb, se);
}

int mc_load_db_host(mc_ctx *c, const uint8_t *sz, const void *keys, int key_bytes, const uint16_t *labels,
                    uint64_t n_keys, uint64_t sb, uint64_t se)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    int rc = set_dev(c); if (rc) return rc;
    rc = norm_shard(c, sb, se); if (rc) return rc;
    // keys of the shard are one contiguous run of the arrays
    uint64_t k0 = 0, kn = 0;
    for (uint64_t b = 0; b < sb; b++) k0 += sz[b];
    for (uint64_t b = sb; b < se; b++) kn += sz[b];
    if (k0 + kn > n_keys) return fail(MC_EINVAL, "bucket sizes exceed n_keys");
    const uint64_t nb = se - sb;
    Scope tmp;
    uint8_t *d_sz = nullptr; void *d_raw = nullptr; uint16_t *d_labels = nullptr;
    TMP_MALLOC(tmp, d_sz, nb ? nb : 1);
    TMP_MALLOC(tmp, d_raw, (kn ? kn : 1) * (size_t)key_bytes);
    TMP_MALLOC(tmp, d_labels, (kn ? kn : 1) * 2);
    HIPCHK(hipMemcpy(d_sz, sz + sb, nb, hipMemcpyHostToDevice));
    if (kn) {
        HIPCHK(hipMemcpy(d_raw, (const char *)keys + k0 * (size_t)key_bytes, kn * (size_t)key_bytes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_labels, labels + k0, kn * 2, hipMemcpyHostToDevice));
    }
    void *d_keys = nullptr; bool owned = false;
    tmp.forget(d_raw);                       // convert_keys takes it over (frees or returns it)
    rc = convert_keys(c, d_raw, key_bytes, kn, true, &d_keys, &owned);
    if (rc != MC_OK) return rc;
    tmp.add(d_keys);
    return relayout(c, d_sz, d_keys, d_labels, kn, sb, se);
}

int mc_load_db(mc_ctx *c, const char *base, int key_bytes, uint32_t sampling, uint64_t sb, uint64_t se)
{
    if (!c || !base) return fail(MC_EINVAL, "ctx/base is NULL");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    int rc = set_dev(c); if (rc) return rc;
    rc = norm_shard(c, sb, se); if (rc) return rc;
    mcint::DbFileStream F;
    rc = F.open(base, key_bytes, sampling, c->htsize, sb, se); if (rc) return rc;
    bool fallback = false;
    if (mcint::minimizer_index_possible(c, F.n_keys_kept)) {
        // two passes over the files in chunks: neither the raw arrays nor a second copy ever sits in HBM
        rc = mcint::load_streamed(&c, 1, F, 1, 0, 0.0);
        if (rc != MC_ENOMEM) return rc;
        fprintf(stderr, "libmcclark: %s; falling back to the bucket-line table\n", g_err.c_str());
        fallback = true;
    }
    return mcint::load_lines_from_files(c, F, sb, se, fallback);
}

int mc_load_db_part(mc_ctx *c, const char *base, int key_bytes, uint32_t sampling, uint32_t part, uint32_t n_parts)
{
    if (!c || !base) return fail(MC_EINVAL, "ctx/base is NULL");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    if (n_parts < 1 || part >= n_parts) return fail(MC_EINVAL, "bad part / n_parts");
    int rc = set_dev(c); if (rc) return rc;
    mcint::DbFileStream F;
    rc = F.open(base, key_bytes, sampling, c->htsize, 0, c->htsize); if (rc) return rc;
    if (!mz_eligible(c, F.n_keys_kept)) return fail(MC_EINVAL, "line-range parts need the minimizer index (k >= 16)");
    // the parts of a table are loaded by different processes and must agree on the fill: a function of the table, the
    // part count and the card's TOTAL memory (index_begin), not of what happens to be free here
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    return mcint::load_streamed(&c, 1, F, n_parts, part, mcint::choose_fill(F.n_keys_kept, n_parts, (uint64_t)tot));
}

/* ---- the streamed index build, for callers that produce the table in chunks ---- */
int mc_index_begin(mc_ctx *c, uint64_t n_keys_total, uint32_t part, uint32_t n_parts)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    int rc = set_dev(c); if (rc) return rc;
    return index_begin(c, n_keys_total, part, n_parts);
}

int mc_index_add_device(mc_ctx *c, const uint8_t *d_sz, const void *d_keys, int key_bytes, const uint16_t *d_labels,
                        uint64_t n_keys, uint64_t bucket_begin, uint64_t bucket_end)
{
    if (!c || !d_sz) return fail(MC_EINVAL, "NULL argument");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    int rc = set_dev(c); if (rc) return rc;
    return index_add_device(c, d_sz, d_keys, key_bytes, d_labels, n_keys, bucket_begin, bucket_end, true);
}

int mc_index_add_host(mc_ctx *c, const uint8_t *sz, const void *keys, int key_bytes, const uint16_t *labels,
                      uint64_t n_keys, uint64_t bucket_begin, uint64_t bucket_end)
{
    if (!c || !sz) return fail(MC_EINVAL, "NULL argument");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    int rc = set_dev(c); if (rc) return rc;
    return index_add_host(c, sz, keys, key_bytes, labels, n_keys, bucket_begin, bucket_end);
}

int mc_index_next_pass(mc_ctx *c)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    int rc = set_dev(c); if (rc) return rc;
    return index_next_pass(c);
}

int mc_index_end(mc_ctx *c)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    int rc = set_dev(c); if (rc) return rc;
    return index_end(c);
}

int mc_index_plan(uint64_t n_keys_total, uint32_t n_parts, uint64_t hbm_bytes, mc_index_plan_t *out)
{
    if (!out || n_parts < 1) return fail(MC_EINVAL, "mc_index_plan: out is NULL or n_parts is 0");
    const double f = mcint::choose_fill(n_keys_total, n_parts, hbm_bytes);
    out->fill = f;
    out->lines_per_part = mcint::lines_per_part(n_keys_total, n_parts, f);
    out->bytes_per_part = mcint::index_bytes(n_keys_total, n_parts, f);
    out->fits = out->lines_per_part != 0 && out->bytes_per_part + mcint::MZ_RESERVE_BYTES <= hbm_bytes ? 1u : 0u;
    out->min_parts = mcint::min_parts(n_keys_total, 4096, hbm_bytes, MC_GROUP_MAX_FILL);
    return MC_OK;
}

int mc_get_db_info(mc_ctx *c, mc_db_info *out)
{
    if (!c || !out) return fail(MC_EINVAL, "NULL argument");
    if (!c->db_loaded) return fail(MC_ESTATE, "no database loaded");
    *out = c->info;
    return MC_OK;
}

#ifdef MC_MZ_STATS
// measurement builds only: the counters of mc_minimizer.hpp (MZ_STAT), read and reset
int mc_debug_stats(unsigned long long *out16)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(mc::mz::g_mz_stats), 16 * sizeof(unsigned long long)));
    unsigned long long z[16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mc::mz::g_mz_stats), z, sizeof z));
    return MC_OK;
}
#endif

int mc_get_stats(mc_ctx *c, mc_stats *out)
{
    if (!c || !out) return fail(MC_EINVAL, "NULL argument");
    int rc = set_dev(c); if (rc) return rc;
    unsigned long long v = 0;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(&v, c->d_over, sizeof v, hipMemcpyDeviceToHost));
    c->stats.reads_over_maxhits = v;
    *out = c->stats;
    return MC_OK;
}

int mc_alloc_batches(mc_ctx *c, uint32_t n_batches, uint64_t max_reads, uint64_t max_con, int want_rows)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    if (n_batches < 1 || max_reads < 1) return fail(MC_EINVAL, "n_batches and max_reads must be >= 1");
    if (max_con > 0xFFFFFFFFull) return fail(MC_EINVAL, "max_containers exceeds the 32-bit offsets of the batch format");
    int rc = set_dev(c); if (rc) return rc;
    mc_free_batches(c);
    if (max_con < 8) max_con = 8;
    max_con = (max_con + 7) & ~7ull;
    c->max_reads = max_reads; c->max_con = max_con; c->want_rows = want_rows != 0;
    const size_t row_len = 2 * (size_t)c->maxhits + 2;
    c->batches.resize(n_batches);
    const unsigned hflags = hipHostMallocDefault;
    for (auto &b : c->batches) {
        hipError_t e = hipHostMalloc((void **)&b.h_ptr, (max_reads + 1) * 4, hflags);
        if (e == hipSuccess) e = hipHostMalloc((void **)&b.h_con, max_con * 2, hflags);
        if (e == hipSuccess) e = hipHostMalloc((void **)&b.h_final, max_reads * MC_FINAL_ROW * 2, hflags);
        if (e == hipSuccess && c->want_rows) e = hipHostMalloc((void **)&b.h_rows, max_reads * row_len * 2, hflags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            mc_free_batches(c);
            return fail(MC_ENOMEM, std::string("pinned batch buffers: ") + hipGetErrorString(e) +
                                       " -- use more, smaller batches (-b)");
        }
    }
    for (auto &s : c->slots) {
        hipError_t e = hipMalloc(&s.d_ptr, (max_reads + 1) * 4);
        if (e == hipSuccess) e = hipMalloc(&s.d_con, max_con * 2);
        if (e == hipSuccess) e = hipMalloc(&s.d_final, max_reads * MC_FINAL_ROW * 2);
        if (e == hipSuccess && c->want_rows) e = hipMalloc(&s.d_rows, max_reads * row_len * 2);
        if (e != hipSuccess) {
            mc_free_batches(c);
            return fail(MC_ENOMEM, std::string("device batch buffers: ") + hipGetErrorString(e) +
                                       " -- use more, smaller batches (-b)");
        }
    }
    return MC_OK;
}

int mc_batch_buffers(mc_ctx *c, uint32_t batch, uint32_t **reads_ptr, uint16_t **containers,
                     uint16_t **final_rows, uint16_t **sparse_rows)
{
    if (!c || batch >= c->batches.size()) return fail(MC_EINVAL, "bad batch index");
    Batch &b = c->batches[batch];
    if (reads_ptr) *reads_ptr = b.h_ptr;
    if (containers) *containers = b.h_con;
    if (final_rows) *final_rows = b.h_final;
    if (sparse_rows) *sparse_rows = b.h_rows;
    return MC_OK;
}

int mc_submit(mc_ctx *c, uint32_t batch, uint64_t n_reads, uint64_t n_con, uint32_t flags)
{
    if (!c || batch >= c->batches.size()) return fail(MC_EINVAL, "bad batch index");
    if (!c->db_loaded) return fail(MC_ESTATE, "mc_submit before a database was loaded");
    if (n_reads > c->max_reads || n_con > c->max_con) return fail(MC_EINVAL, "batch larger than allocated");
    if (!(flags & (MC_F_FINAL | MC_F_ROWS))) return fail(MC_EINVAL, "flags select no output");
    if ((flags & MC_F_ROWS) && !c->want_rows) return fail(MC_ESTATE, "sparse rows were not allocated");
    int rc = set_dev(c); if (rc) return rc;
    Batch &b = c->batches[batch];
    if (n_reads && b.h_ptr[n_reads] != n_con) return fail(MC_EINVAL, "reads_ptr[n_reads] != n_containers");
    rc = mcint::submit_batch(c, b.h_ptr, b.h_con, b.h_final, b.h_rows, n_reads, n_con, flags, b.ev);
    if (rc) return rc;
    b.submitted = true;
    return MC_OK;
}

int mc_wait(mc_ctx *c, uint32_t batch)
{
    if (!c || batch >= c->batches.size()) return fail(MC_EINVAL, "bad batch index");
    Batch &b = c->batches[batch];
    if (!b.submitted) return fail(MC_ESTATE, "batch was never submitted");
    int rc = set_dev(c); if (rc) return rc;
    HIPCHK(hipEventSynchronize(b.ev));
    return MC_OK;
}

int mc_sync(mc_ctx *c)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    int rc = set_dev(c); if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    return MC_OK;
}

int mc_free_batches(mc_ctx *c)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto &b : c->batches) {
        if (b.h_ptr) (void)hipHostFree(b.h_ptr);
        if (b.h_con) (void)hipHostFree(b.h_con);
        if (b.h_final) (void)hipHostFree(b.h_final);
        if (b.h_rows) (void)hipHostFree(b.h_rows);
        if (b.ev) (void)hipEventDestroy(b.ev);
    }
    c->batches.clear();
    for (auto &s : c->slots) {
        if (s.d_ptr) (void)hipFree(s.d_ptr);
        if (s.d_con) (void)hipFree(s.d_con);
        if (s.d_final) (void)hipFree(s.d_final);
        if (s.d_rows) (void)hipFree(s.d_rows);
        s = Slot();
    }
    return MC_OK;
}

int mc_query_device(mc_ctx *c, const uint32_t *d_ptr, const uint16_t *d_con, uint64_t n_reads, uint64_t n_con,
                    uint32_t flags, uint16_t *d_final, uint16_t *d_rows, void *stream)
{
    if (!c) return fail(MC_EINVAL, "ctx is NULL");
    if (!c->db_loaded) return fail(MC_ESTATE, "mc_query_device before a database was loaded");
    if (!(flags & (MC_F_FINAL | MC_F_ROWS))) return fail(MC_EINVAL, "flags select no output");
    if ((flags & MC_F_FINAL) && !d_final) return fail(MC_EINVAL, "MC_F_FINAL without d_final_rows");
    if ((flags & MC_F_ROWS) && !d_rows) return fail(MC_EINVAL, "MC_F_ROWS without d_sparse_rows");
    int rc = set_dev(c); if (rc) return rc;
    return launch_query(c, d_ptr, d_con, n_reads, n_con, flags, d_final, d_rows, (hipStream_t)stream);
}

int mc_merge_rows_device(mc_ctx *c, const uint16_t *d_a, const uint16_t *d_b, uint64_t n_reads,
                         uint16_t *d_out, void *stream)
{
    if (!c || !d_a || !d_b || !d_out) return fail(MC_EINVAL, "NULL argument");
    int rc = set_dev(c); if (rc) return rc;
    if (n_reads == 0) return MC_OK;
    const uint32_t row_len = 2 * c->maxhits + 2;
    const uint32_t g = (uint32_t)((n_reads + 255) / 256);
    hipLaunchKernelGGL(mc::merge_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream,
                       d_a, d_b, row_len, n_reads, d_out);
    HIPCHK(hipGetLastError());
    return MC_OK;
}

int mc_result_rows_device(mc_ctx *c, const uint16_t *d_rows, uint64_t n_reads, uint16_t *d_final, void *stream)
{
    if (!c || !d_rows || !d_final) return fail(MC_EINVAL, "NULL argument");
    int rc = set_dev(c); if (rc) return rc;
    if (n_reads == 0) return MC_OK;
    const uint32_t row_len = 2 * c->maxhits + 2;
    const uint32_t g = (uint32_t)((n_reads + 255) / 256);
    hipLaunchKernelGGL(mc::result_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream,
                       d_rows, row_len, n_reads, d_final);
    HIPCHK(hipGetLastError());
    return MC_OK;
}

int mc_merge_result_device(mc_ctx *c, const uint16_t *const *d_srcs, uint32_t n_srcs, uint64_t n_reads,
                           uint16_t *d_out_rows, uint16_t *d_final, void *stream)
{
    if (!c || !d_srcs) return fail(MC_EINVAL, "NULL argument");
    if (!d_out_rows && !d_final) return fail(MC_EINVAL, "no output selected");
    int rc = set_dev(c); if (rc) return rc;
    return mcint::launch_merge_result(c, d_srcs, n_srcs, n_reads, d_out_rows, d_final, (hipStream_t)stream);
}

} // extern "C"
