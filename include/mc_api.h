/*
 * mc_api.h -- C ABI of the MI355X metagenomic classifier core (libmcclark.so).
 *
 * This is the drop-in boundary for ONE path of ardaicoz/jn_cuclark: the device
 * manager + kernels behind `class CuClarkDB<HKMERr>` (reference
 * src/CuClarkDB.cuh:39-153, src/CuClarkDB.cu).  Every entry point below names the
 * reference member it replaces.  Plain C types only: pointers, sizes, integers.
 * INTEGRATION.md shows the C++ shim a maintainer drops in place of CuClarkDB.cu.
 *
 * Conventions
 *   - every function returns MC_OK (0) or a negative MC_E* code; the message is
 *     available from mc_last_error() (thread-local).  Nothing calls exit(): the
 *     reference's CUERR macro (CuClarkDB.cu:45-53) printed and exited instead.
 *   - one mc_ctx = one GPU (one HIP device, its streams, one database or one part of it).
 *     Multi-GPU = mc_group.h (one process drives all GPUs, as the reference's object does),
 *     or one process per GPU, each with its own ctx, combined with mc_merge_result_device
 *     after the exchange (jn_cuclark_amd/dist.py).
 *   - HTSIZE and MAXHITS are run-time parameters (the reference compiles two
 *     binaries, src/parameters.hh:37-48 vs src/parameters_light_hh:38-49).
 *   - wire types are the reference's: containers/labels/results are uint16_t,
 *     read offsets uint32_t (src/dataType.hh:37-43).
 *
 * Defined behaviour where the reference has none (DESIGN.md "Deviations"):
 *   a read that hits more than `maxhits` distinct targets keeps the `maxhits`
 *   SMALLEST target ids (row and final result alike) and is counted in
 *   mc_stats.reads_over_maxhits; the reference overruns shared memory there
 *   (CuClarkDB.cu:1140-1151).
 */
#ifndef MC_API_H
#define MC_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_API_VERSION 4   /* 3 (round 4): mc_text_*, mc_group_text_*, MC_INDEX_SUPERKMER; 4: database cycles (mc_group_set_cycle,
                              MC_F_FOLLOWUP, two fields at the end of mc_group_info); nothing removed or changed */

enum {
    MC_OK          =  0,
    MC_EINVAL      = -1,   /* bad argument / unsupported parameter combination */
    MC_EIO         = -2,   /* database files missing or short                  */
    MC_ENOMEM      = -3,   /* host or device allocation failed                 */
    MC_EHIP        = -4,   /* HIP runtime error (message has the HIP string)   */
    MC_ESTATE      = -5,   /* call out of order (e.g. submit before load_db)   */
    MC_ENODEVICE   = -6    /* no usable gfx950 device                          */
};

typedef struct mc_ctx mc_ctx;

/* Size of one final result row: [sumN, idxBest+1, best, idxSecond+1, second]
 * (CuClarkDB.cu:1401-1405, CuCLARK_hh.hh:1592). */
#define MC_FINAL_ROW 5

/* flags of mc_submit / mc_query_device */
#define MC_F_FINAL    1u   /* produce the 5-u16 final rows (the fused query+top-2 path) */
#define MC_F_ROWS     2u   /* produce sparse rows [n, t0,h0, ...] (--extended, shards)  */
#define MC_F_FOLLOWUP 4u   /* mc_group_submit only: the batch's sparse-row buffer holds the rows of earlier database cycles;
                              they are merged with this cycle's (queryBatch(.., followup), CuClarkDB.cu:932-948) */

typedef struct mc_db_info {
    uint64_t htsize;          /* total buckets of the table                         */
    uint64_t shard_begin;     /* first bucket held by this ctx                      */
    uint64_t shard_end;       /* one past the last bucket held by this ctx          */
    uint64_t n_keys;          /* k-mers resident on this device                     */
    uint64_t n_overflow_buckets; /* bucket-line table: buckets larger than a line (kept in a side CSR); minimizer index: extra lines */
    uint64_t n_overflow_keys;    /* bucket-line table: k-mers of those buckets; minimizer index: k-mers in the chains of crowded lines */
    uint32_t line_bytes;      /* bucket line size chosen at load (64 or 128)         */
    uint32_t line_capacity;   /* k-mers per line                                     */
    uint64_t device_bytes;    /* HBM held by the database                            */
    /* --- version 2 --- */
    uint32_t index_kind;      /* MC_INDEX_BUCKET_LINES, MC_INDEX_MINIMIZER or MC_INDEX_SUPERKMER: which in-HBM index was built */
    uint32_t index_fallback;  /* 1: the minimizer index was wanted but did not fit; bucket lines were built
                                 (also reported on stderr)                                                 */
    uint32_t part, n_parts;   /* line-range part of a table spread over n_parts contexts (0 of 1 = whole)  */
    uint64_t n_keys_owned;    /* k-mers this context answers for (= n_keys unless n_parts > 1)            */
    /* minimizer index only (for it n_overflow_buckets = extra lines, n_overflow_keys = spilled k-mers): */
    uint64_t n_lines;             /* primary lines of the whole table (all parts; every part holds n_lines / n_parts) */
    uint64_t line_begin, line_end;/* primary lines held here: part * n_lines / n_parts and one part further            */
    uint64_t n_extra_lines;       /* lines chained behind overflowing primary lines    */
    uint64_t n_lines_crowded;     /* primary lines with more k-mers than one chain holds (48): their k-mers
                                     are spread over 2^s chains picked by a hash of the k-mer              */
    uint64_t n_lines_overflowing; /* primary lines with more k-mers than slots         */
    uint64_t n_spilled_keys;      /* k-mers in the chains of crowded lines             */
    uint32_t largest_line;        /* most k-mers that share one primary line           */
    uint32_t reserved_;
} mc_db_info;

#define MC_INDEX_BUCKET_LINES 0u
#define MC_INDEX_MINIMIZER    1u
#define MC_INDEX_SUPERKMER    2u   /* records of up to 9 k-mers stored relative to their minimizer (csrc/mc_skm.hpp): MC_INDEX=skm */

typedef struct mc_stats {
    uint64_t reads;               /* reads classified since mc_open                  */
    uint64_t reads_over_maxhits;  /* reads that hit more than maxhits targets        */
    uint64_t kernel_launches;
} mc_stats;

const char *mc_last_error(void);
int mc_api_version(void);

/* Number of gfx950 devices visible.  replaces: cudaGetDeviceCount, CuClarkDB.cu:120 */
int mc_device_count(int *count);

/* Create a context on `device` (-1 = current HIP device).
 * replaces: CuClarkDB::CuClarkDB (CuClarkDB.cu:94-241).  k in [2,32]; when some
 * quotient (4^k-1)/htsize does not fit 32 bits (the reference's T64 regime, k = 32,
 * main.cc:277-286) the table uses 64-bit keys.  num_targets = targetsName.size()-1;
 * maxhits = MAXHITS (15 full / 23 light), at most 63. */
int mc_open(mc_ctx **out, int device, uint32_t k, uint64_t htsize,
            uint32_t num_targets, uint32_t maxhits);
int mc_close(mc_ctx *ctx);

/* Load <base>.sz/.ky/.lb, keep the buckets [shard_begin, shard_end) (0,0 = all),
 * and lay them out in HBM as the minimizer index (default; streamed from the files in two passes) or the
 * bucket-line table (mc_db_info.index_kind says which).
 * replaces: CuClarkDB::read + swapDbParts (CuClarkDB.cu:463-815).  key_bytes =
 * sizeof(HKMERr) of the files (2, 4 or 8).  sampling = the -s factor (<=1: none,
 * CuClarkDB.cu:490-513).  MC_EIO when a file is missing (the reference returns
 * false and the caller rebuilds, CuCLARK_hh.hh:622-684). */
int mc_load_db(mc_ctx *ctx, const char *base, int key_bytes, uint32_t sampling,
               uint64_t shard_begin, uint64_t shard_end);

/* Same, from the three arrays already in host memory (contents of .sz/.ky/.lb). */
int mc_load_db_host(mc_ctx *ctx, const uint8_t *sz, const void *keys, int key_bytes,
                    const uint16_t *labels, uint64_t n_keys,
                    uint64_t shard_begin, uint64_t shard_end);

/* Same, from arrays already in HBM on ctx's device: d_sz covers buckets
 * [shard_begin, shard_end) only, d_keys (key_bytes each) / d_labels their n_keys elements. */
int mc_load_db_device(mc_ctx *ctx, const uint8_t *d_sz, const void *d_keys, int key_bytes,
                      const uint16_t *d_labels, uint64_t n_keys,
                      uint64_t shard_begin, uint64_t shard_end);

/* Part of a table that is spread over n_parts contexts (GPUs): this context keeps the k-mers whose MINIMIZER
 * hashes to `part` (mc_minimizer.hpp part_of; line range `part` of the table's line space), streams the WHOLE
 * files through the build kernels and drops the rest.  All parts must be loaded with the same n_parts on cards of
 * the same size (the fill is a function of the table, n_parts and the card's HBM; MC_MZ_FILL fixes it).  Replaces the same members as mc_load_db for the
 * multi-device case (the reference gives device d the bucket range m_partPointer[d..d+1],
 * CuClarkDB.cu:552-559, and every device every read batch, :842-851).  Where the reference's bucket ranges
 * scatter the consecutive k-mers of a read over all devices (every device fetches nearly every line), a
 * line range keeps a read's run on ONE device: fetch, match and scoring divide by n_parts.  The per-read
 * rows of the parts add up exactly like those of bucket ranges (mc_merge_result_device).  Needs k >= 16. */
int mc_load_db_part(mc_ctx *ctx, const char *base, int key_bytes, uint32_t sampling,
                    uint32_t part, uint32_t n_parts);

/* The same index built from a table the caller produces in bucket-order CHUNKS (any size), fed twice:
 *   mc_index_begin -> mc_index_add_* for every chunk -> mc_index_next_pass -> the same chunks again
 *   -> mc_index_end.
 * Neither the raw arrays of the whole table nor a second copy are ever resident next to the lines
 * (CuClarkDB::read budgets pinned host memory for exactly that, CuClarkDB.cu:516-540).  n_keys_total =
 * k-mers of the whole table (all parts): it sizes the line space and must agree between the parts.
 * A chunk = buckets [bucket_begin, bucket_end): sz has one byte per bucket, keys/labels n_keys entries. */
int mc_index_begin(mc_ctx *ctx, uint64_t n_keys_total, uint32_t part, uint32_t n_parts);
int mc_index_add_device(mc_ctx *ctx, const uint8_t *d_sz, const void *d_keys, int key_bytes,
                        const uint16_t *d_labels, uint64_t n_keys, uint64_t bucket_begin, uint64_t bucket_end);
int mc_index_add_host(mc_ctx *ctx, const uint8_t *sz, const void *keys, int key_bytes,
                      const uint16_t *labels, uint64_t n_keys, uint64_t bucket_begin, uint64_t bucket_end);
int mc_index_next_pass(mc_ctx *ctx);
int mc_index_end(mc_ctx *ctx);

/* The loader's arithmetic, without a device: how a table of n_keys_total k-mers spread over n_parts contexts with
 * hbm_bytes of free HBM each would be laid out -- k-mers per line (the sparsest of 3 .. 12 that fits, 16 GB kept in
 * reserve), primary lines and bytes per part, whether it fits at all, and the smallest part count that holds the
 * table at no more than MC_GROUP_MAX_FILL k-mers per line (what mc_group_load_db cuts it into; 0 = more than 4096).
 * replaces: the memory budget of CuClarkDB::read (CuClarkDB.cu:516-559: minParts, m_partPointer).
 * A context addresses at most 2^32 - 16 primary lines (550 GB); a table spread over several contexts has that many
 * PER PART (the part is drawn from a second hash of the minimizer key, csrc/mc_minimizer.hpp part_of). */
#define MC_GROUP_MAX_FILL 10.0
typedef struct mc_index_plan_t {
    double   fill;            /* k-mers per 12-slot line                                  */
    uint64_t lines_per_part;  /* primary lines of one part (0: beyond the line index)     */
    uint64_t bytes_per_part;  /* estimate: lines + extra lines + build counters           */
    uint32_t fits;            /* 1: bytes_per_part + reserve <= hbm_bytes                  */
    uint32_t min_parts;       /* smallest n_parts that fits at fill <= MC_GROUP_MAX_FILL   */
} mc_index_plan_t;
int mc_index_plan(uint64_t n_keys_total, uint32_t n_parts, uint64_t hbm_bytes, mc_index_plan_t *out);

int mc_get_db_info(mc_ctx *ctx, mc_db_info *out);
int mc_get_stats(mc_ctx *ctx, mc_stats *out);

/* Allocate per-batch pinned input buffers, the result tables and the device
 * staging buffers.  The library owns all of them.
 * replaces: CuClarkDB::malloc (CuClarkDB.cu:321-421).  max_reads / max_containers
 * bound one batch.  want_rows != 0 also allocates sparse-row outputs
 * (row = 2*maxhits+2 u16, CuCLARK_hh.hh:1586-1589). */
int mc_alloc_batches(mc_ctx *ctx, uint32_t n_batches, uint64_t max_reads,
                     uint64_t max_containers, int want_rows);

/* Pinned host pointers of one batch: the caller fills reads_ptr[n_reads+1] and
 * containers[], and reads final_rows / sparse_rows after mc_wait.  Any out pointer
 * may be NULL.  (In the reference these come back from malloc, .cuh:112-123.) */
int mc_batch_buffers(mc_ctx *ctx, uint32_t batch, uint32_t **reads_ptr,
                     uint16_t **containers, uint16_t **final_rows, uint16_t **sparse_rows);

/* Enqueue one filled batch: H2D, query kernel, D2H, completion event; returns
 * without waiting.  Three queues chained by events (copy in, compute, copy out) and two
 * device slots: the copy-in of the next batch and the copy-out of the previous one run
 * under the kernel of the current one.  Batches may be submitted in any order.
 * replaces: readyBatch + queryBatch (CuClarkDB.cu:820-987). */
int mc_submit(mc_ctx *ctx, uint32_t batch, uint64_t n_reads, uint64_t n_containers,
              uint32_t flags);

/* Block until the batch's results are in its host buffers.
 * replaces: waitForBatch (CuClarkDB.cu:441-446). */
int mc_wait(mc_ctx *ctx, uint32_t batch);

/* replaces: sync (CuClarkDB.cu:426-436) */
int mc_sync(mc_ctx *ctx);

/* replaces: freeBatchMemory (CuClarkDB.cu:284-316) */
int mc_free_batches(mc_ctx *ctx);

/* ---- device-resident entry points (inputs and outputs already in HBM) -------
 * `stream` is a hipStream_t (NULL = the HIP default stream).  Used by the
 * multi-GPU path (rows travel over RCCL between the calls) and by bench.py. */

/* queryKernel (+ fused resultKernel when MC_F_FINAL): CuClarkDB.cu:999-1254.  One batch holds at most
 * 2^32 - 32 reads and 2^32 - 1 containers (reads_ptr is u32, as the reference's, :1034-1035): MC_EINVAL beyond. */
int mc_query_device(mc_ctx *ctx, const uint32_t *d_reads_ptr, const uint16_t *d_containers,
                    uint64_t n_reads, uint64_t n_containers, uint32_t flags,
                    uint16_t *d_final_rows, uint16_t *d_sparse_rows, void *stream);

/* mergeKernel: out = union(a, b) with counts added; out may alias a.
 * CuClarkDB.cu:1261-1355. */
int mc_merge_rows_device(mc_ctx *ctx, const uint16_t *d_a, const uint16_t *d_b,
                         uint64_t n_reads, uint16_t *d_out, void *stream);

/* resultKernel: sparse rows -> final rows.  CuClarkDB.cu:1361-1411. */
int mc_result_rows_device(mc_ctx *ctx, const uint16_t *d_rows, uint64_t n_reads,
                          uint16_t *d_final_rows, void *stream);

/* mergeKernel over all shards at once + resultKernel: d_srcs[0..n_srcs) (a HOST array of device pointers,
 * 1 <= n_srcs <= 16) each hold n_reads sparse rows; writes the merged rows (d_out_rows, may be NULL) and the
 * final rows (d_final_rows, may be NULL).  Equal to folding the sources pairwise with mc_merge_rows_device
 * and then mc_result_rows_device.  CuClarkDB.cu:909-928 (merge tree) + :963-968. */
int mc_merge_result_device(mc_ctx *ctx, const uint16_t *const *d_srcs, uint32_t n_srcs, uint64_t n_reads,
                           uint16_t *d_out_rows, uint16_t *d_final_rows, void *stream);

/* ---- FASTQ text batches: record boundaries, 2-bit packing and classification on the device (csrc/mc_ingest.hip) ----
 * An ADDITION to the reference's interface (like --gpu-build): the reference parses and packs on the host
 * (getObjectsDataComputeFullGPU, src/CuCLARK_hh.hh:1405-1534 FASTQ records, :1629-1707 packing) and hands packed reads to
 * readyBatch / queryBatch.  Here the caller copies the BYTES of a batch -- whole 4-line FASTQ records, starting at an '@' --
 * into a pinned buffer; back come the final rows (as mc_wait leaves them) and, per read, the offset of its header line in
 * the text and the length of its sequence line (the names stay with the caller's text).
 * status != 0 (mc_text_wait): the batch was NOT classified -- bit 0 a record this code does not vouch for (a header that
 * does not start with '@', a name that starts with a blank), bit 1 a line count that is not a multiple of four, bit 2 more
 * reads / containers than allocated -- and the caller packs it on the host (mc_submit).  MC_F_FINAL only. */
int mc_text_alloc(mc_ctx *ctx, uint32_t n_buffers, uint64_t max_text_bytes, uint64_t max_reads, uint64_t max_containers);
/* where the results of a buffer's batch arrive (pinned host memory, valid after mc_text_wait) */
int mc_text_buffers(mc_ctx *ctx, uint32_t buffer, uint32_t **header_offsets, uint32_t **sequence_lengths, uint16_t **final_rows);
/* `text` is the caller's own memory (a mapped file will do: it is uploaded from where it lies, on the buffer's own stream --
 * several threads may submit different buffers at once and their uploads overlap); everything behind the upload is queued.
 * A buffer takes its next batch after mc_text_wait. */
int mc_text_submit(mc_ctx *ctx, uint32_t buffer, const uint8_t *text, uint64_t n_bytes);
int mc_text_wait(mc_ctx *ctx, uint32_t buffer, uint64_t *n_reads, uint32_t *status);
int mc_text_free(mc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MC_API_H */
