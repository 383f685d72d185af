// mc_internal.hpp -- what the translation units of libmcclark.so share: the context, the
// error channel, scoped device temporaries.  Nothing here is part of the C ABI.
#pragma once

#include "../../include/mc_api.h"
#include "mc_device.hpp"

#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

namespace mcint {

// sets the thread-local message mc_last_error() returns; returns `code`
int fail(int code, const std::string &msg);

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return mcint::fail(MC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Device temporaries of one API call: released when the call returns, on every path.
struct Scope {
    std::vector<void *> ptrs;
    ~Scope() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    void add(void *p) { ptrs.push_back(p); }
    void drop(void *p)      // release early (large temporaries), or after ownership moved on
    {
        for (auto &q : ptrs) if (q == p && p) { (void)hipFree(p); q = nullptr; }
    }
    void forget(void *p) { for (auto &q : ptrs) if (q == p) q = nullptr; }
};
#define TMP_MALLOC(scope, ptr, bytes)                                                   \
    do {                                                                                \
        HIPCHK(hipMalloc(&(ptr), (bytes)));                                             \
        (scope).add(ptr);                                                               \
    } while (0)

struct Batch {
    uint32_t *h_ptr = nullptr;
    uint16_t *h_con = nullptr;
    uint16_t *h_final = nullptr;
    uint16_t *h_rows = nullptr;
    hipEvent_t ev = nullptr;
    bool submitted = false;
};

struct Slot {
    uint32_t *d_ptr = nullptr;
    uint16_t *d_con = nullptr;
    uint16_t *d_final = nullptr;
    uint16_t *d_rows = nullptr;
};

// a minimizer index under construction (mc_index_begin .. mc_index_end)
struct IndexBuild {
    bool open = false;
    int pass = 0;
    uint64_t n_keys_total = 0;         // k-mers of the whole table (all parts): sizes the line space
    uint64_t fed[2] = {0, 0};          // k-mers fed in each pass
    uint64_t bucket_lo = ~0ull, bucket_hi = 0;
    uint32_t *d_count = nullptr;       // per owned line: k-mers (pass 0), cursor (pass 1)
    unsigned int *d_failed = nullptr;  // bit 0: a k-mer found no slot; bit 1: the second pass outgrew a chain; bit 2: a chunk's
                                       // bucket sizes do not add up to its k-mers
    // scratch that lives as long as the build (grow-only): feeding a chunk allocates nothing and waits for nothing
    uint32_t *d_blk = nullptr; uint64_t *d_koff = nullptr; size_t blk_cap = 0;
    uint8_t *d_st_sz = nullptr; void *d_st_keys = nullptr; uint16_t *d_st_labels = nullptr;   // staging of mc_index_add_host
    size_t st_sz_cap = 0, st_key_cap = 0;                                                     // buckets / k-mers
    uint64_t n_extra = 0, n_spilled = 0, n_over = 0, n_crowded = 0;
    uint32_t longest = 0;
    // the super-k-mer index (mc_skm.hpp): entries per FINE line (d_count), cursors of the second pass, offsets of the fine
    // lines into the entries, the entries themselves
    bool sk = false;
    // MC_INDEX=auto: the first pass counts for BOTH indexes; mc_index_next_pass looks at how the k-mers clump around their
    // minimizers (the counts of the fine lines) and goes on with one of them
    bool both = false;
    uint32_t *d_count_mz = nullptr;
    double mz_per_line = 0.0;
    uint32_t sk_n_fine = 0;
    uint32_t *d_cursor = nullptr, *d_off32 = nullptr;
    uint64_t *d_blk_base = nullptr;
    void *d_entries = nullptr;
    bool entries_on_host = false;      // the entries did not fit next to the lines they turn into: pinned host memory, written and read over PCIe
    uint64_t sk_n_entries = 0;
};

} // namespace mcint

namespace mcint { struct TextState; void text_release(mc_ctx *c); }

struct mc_ctx {
    int device = 0;
    mcint::TextState *text = nullptr;  // FASTQ text batches (mc_ingest.hip)
    uint32_t k = 0, num_targets = 0, maxhits = 0;
    uint64_t htsize = 0;
    mc::DivU64 div{};
    bool wide = false;          // quotients need 64 bits (reference T64 regime: k = 32)
    int n_cu = 0;

    hipStream_t streams[2] = {nullptr, nullptr};      // [0]: compute (query kernels of the batch interface, index build); [1]: group shards, slot 1
    // The batch interface runs three queues (copy in, compute, copy out) chained by events, not one stream per
    // slot: two streams that each do copy -> kernel -> copy fall into lockstep (both copy, then both compute) and
    // overlap nothing (measured: 688 Mreads/s at 30 GB/s although the link does 57 GB/s and the kernel 1260).
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};   // per slot
    uint64_t n_submitted = 0;

    // database
    bool db_loaded = false;
    uint8_t *d_lines = nullptr;
    void *d_ovf_keys = nullptr;
    uint16_t *d_ovf_labels = nullptr;
    mc_db_info info{};
    int grid_blocks = 0;

    // locality-aware index (mc_minimizer.hpp): the default for k >= 16; MC_INDEX=lines
    // selects the direct bucket-line table instead (also the fallback when the minimizer
    // lines do not fit in HBM)
    int index_mode = 3;                // 0 = bucket lines (MC_INDEX=lines), 1 = minimizer lines (=minimizer), 2 = super-k-mer records
                                       // (=skm; where k allows, else 1), 3 = minimizer lines or super-k-mer records, whichever suits
                                       // the table (=auto, the default; where k allows and the table is built whole or by a group
                                       // loader -- parts that are built one by one cannot agree on a choice and get minimizer lines)
    int auto_decision = 0;             // a group loader's first member decided: 1 = records, 2 = lines (0 = decide here)
    bool group_loading = false;        // the build is driven by load_streamed (all members of a group in step)
    uint32_t sk_d = 0;                 // merge factor the super-k-mer index was built with ...
    uint32_t sk_d_hint = 0;            // ... and the one a group loader prescribes (all members: one layout; 0 = choose here)
    uint8_t *d_sk_lines = nullptr, *d_sk_extra = nullptr;      // super-k-mer index: lines of 8 slots, extra lines (chains)
    uint32_t sk_n_lines = 0;
    uint8_t *d_mz_lines = nullptr, *d_mz_extra = nullptr;
    // The extra lines live BEHIND the primary lines in the same allocation whenever the room reserved for them at
    // mc_index_begin (the loader's estimate of their share) suffices: an allocation of its own, made after the first
    // build pass from a heap that the build's temporaries have cut up, ends up on small pages, and a fetch from it costs
    // several times a fetch from the primary lines (62-200 ns against 24 ns per line)
    uint64_t mz_extra_reserved = 0;    // extra lines behind the primary lines, in the same allocation (round 4: exactly the
                                       // number the first build pass counted; the lines are allocated between the passes)
    bool mz_extra_own_alloc = false;   // (no longer set: the extra lines never get an allocation of their own)
    uint32_t mz_n_local = 0;           // primary lines held here (every part of a table has the same number)
    uint32_t mz_part = 0, mz_n_parts = 1, mz_m = 0;
    double fill_hint = 0.0;            // k-mers per line chosen by a group loader for all its members (0 = choose here)
    mcint::IndexBuild build;

    unsigned long long *d_over = nullptr;

    // batches
    std::vector<mcint::Batch> batches;
    mcint::Slot slots[2];
    uint64_t max_reads = 0, max_con = 0;
    bool want_rows = false;

    mc_stats stats{};
};

namespace mcint {

int set_dev(mc_ctx *c);

// one query launch on `st` (no synchronisation)
// n_dev != nullptr: the batch's reads ([0]) and containers ([1]) are counted on the device; n_reads / n_con are upper bounds
int launch_query(mc_ctx *c, const uint32_t *d_ptr, const uint16_t *d_con, uint64_t n_reads, uint64_t n_con,
                 uint32_t flags, uint16_t *d_final, uint16_t *d_rows, hipStream_t st, const uint32_t *n_dev = nullptr);
// one batch of the batch interface: H2D, kernel, D2H on the context's three queues; `done` behind the D2H.  The
// device buffers are c->slots[], dealt by submission order.
int submit_batch(mc_ctx *c, const uint32_t *h_ptr, const uint16_t *h_con, uint16_t *h_final, uint16_t *h_rows,
                 uint64_t n_reads, uint64_t n_con, uint32_t flags, hipEvent_t done);
// k-way merge of sparse rows (+ top-2) on `st`
int launch_merge_result(mc_ctx *c, const uint16_t *const *d_srcs, uint32_t n_srcs, uint64_t n_reads,
                        uint16_t *d_out_rows, uint16_t *d_final, hipStream_t st);

// Streams <base>.sz/.ky/.lb in bucket order: calls chunk(pass, sz, keys, labels, n_keys, b0, b1) for
// every chunk of buckets [b0, b1) inside [sb, se), twice (pass 0, then between(…), then pass 1), with the
// -s sampling rule applied (dropped buckets arrive with size 0).  n_keys_kept = k-mers of [sb, se).
struct DbFileStream {
    std::string base;
    int key_bytes = 4;
    uint32_t sampling = 1;
    uint64_t htsize = 0, sb = 0, se = 0;
    uint64_t n_keys_kept = 0;
    std::vector<uint8_t> sz;           // effective sizes of all buckets (dropped = 0)
    std::vector<uint8_t> fsz;          // sizes in the files (only when sampling drops buckets)
    uint64_t file_k0 = 0;              // file position (in k-mers) of bucket sb
    int fs = -1, fk = -1, fl = -1;
    ~DbFileStream();
    int open(const char *base_path, int key_bytes, uint32_t sampling, uint64_t htsize, uint64_t sb, uint64_t se);
    // one pass over the chunks; f(sz of [b0,b1), keys, labels, n_keys, b0, b1) returns MC_OK to continue
    typedef std::function<int(const uint8_t *, const void *, const uint16_t *, uint64_t, uint64_t, uint64_t)> ChunkFn;
    int pass(const ChunkFn &f);
    // The same chunks, addressable: plan() lists them once (bucket range, file position, k-mers in the files);
    // read(i, ...) fills caller-provided (pinned) arrays with chunk i -- the .ky/.lb byte ranges are cut into slices
    // read by several threads at once (one thread copies from the page cache at 2-3 GB/s; the files of a full
    // table are 40 GB, twice) -- and returns the k-mers kept after sampling.
    struct Chunk { uint64_t b0, b1, fpos, nfile; };
    std::vector<Chunk> chunks;
    uint64_t max_nfile = 0, max_nb = 0;
    void plan();
    int read(size_t i, uint8_t *sz_out, void *keys_out, uint16_t *labels_out, uint64_t *n_kept, int threads);
};

// Build the minimizer index of every context from one stream of the files (each chunk is read once per
// pass and fed to all contexts).  Context i builds part i % n_parts of n_parts (n_parts = 1: every context
// builds the whole [sb, se) range -- replicas; n > n_parts: several groups that each hold the whole table).
// MC_ENOMEM when a context cannot hold its lines.
// Context i builds part (part0 + i) % n_parts.  fill > 0 fixes the k-mers per line (otherwise: what the member with
// the least free HBM affords).
int load_streamed(mc_ctx *const *ctxs, uint32_t n, DbFileStream &F, uint32_t n_parts, uint32_t part0 = 0, double fill = 0.0);
bool minimizer_index_possible(const mc_ctx *c, uint64_t n_keys_total);
// The arithmetic of the index plan (no device needed: mc_index_plan exposes it to tests).
// lines_per_part: primary lines of ONE part at `fill` k-mers per line; 0 when that exceeds the 32-bit line index.
uint64_t lines_per_part(uint64_t n_keys_total, uint32_t n_parts, double fill);
uint64_t index_bytes(uint64_t n_keys_total, uint32_t n_parts, double fill);
double extra_share(double fill);
// the sparsest fill in [4, 12] whose share fits `free_bytes` (16 GB kept in reserve) AND the line index
double choose_fill(uint64_t n_keys_total, uint32_t n_parts, uint64_t free_bytes);
// smallest part count in [1, max_parts] whose share fits at a fill of at most `max_fill`; 0 = none does
uint32_t min_parts(uint64_t n_keys_total, uint32_t max_parts, uint64_t free_bytes, double max_fill);
static constexpr uint64_t MZ_MAX_LINES = 0xFFFFFFF0ull;      // per context
static constexpr uint64_t MZ_RESERVE_BYTES = 16ull << 30;    // left free next to the index (batch buffers, runtime)

} // namespace mcint
