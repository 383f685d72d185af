/*
 * mc_build.h -- GPU build of a cuCLARK database (part of libmcclark.so).
 *
 * Replaces the CPU hash-table build of the reference
 * (CuCLARK<T>::makeSpecificTargetSets, src/CuCLARK_hh.hh:690-1329, with
 * EHashtable::addElement / SortAllHashTable / RemoveCommon / Write,
 * src/HashTableStorage_hh.hh:421-461, :229-280, src/hashTable_hh.hh:203-216, :473-546)
 * for the same result: the three files <base>.sz/.ky/.lb holding, bucket by bucket and
 * ascending inside a bucket, the canonical k-mers that occur in exactly ONE target more
 * than min_count times.  The reference needs 16 bytes x HTSIZE (25.8 GB) of empty
 * chained buckets before the first k-mer; this build is a counting sort by bucket:
 *
 *     pass 1  mc_builder_count   canonical, q = c / HTSIZE, r = c % HTSIZE, count[r]++
 *             mc_builder_begin_fill   exclusive scan of the counts -> bucket offsets
 *     pass 2  mc_builder_fill    scatter (q, target) into bucket r
 *             mc_builder_finish  sort every bucket, keep single-target runs, compact
 *             mc_builder_write   copy back, write the files (keys narrowed to key_bytes)
 *
 * The caller streams the SAME occurrences (forward k-mer value + 0-based target id) twice,
 * in chunks of any size; extracting them from FASTA/FASTQ stays on the host
 * (jn_cuclark_amd/host/dbbuild.hpp).  Plain C ABI like mc_api.h; errors via mc_last_error().
 */
#ifndef MC_BUILD_H
#define MC_BUILD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_builder mc_builder;

int mc_builder_open(mc_builder **out, int device, uint32_t k, uint64_t htsize);
int mc_builder_count(mc_builder *b, const uint64_t *kmers_fwd, uint64_t n);
int mc_builder_begin_fill(mc_builder *b);
int mc_builder_fill(mc_builder *b, const uint64_t *kmers_fwd, const uint16_t *targets, uint64_t n);
/* n_distinct: distinct canonical k-mers seen ("Mother Hashtable"), n_stored: kept. */
int mc_builder_finish(mc_builder *b, uint32_t min_count, uint64_t *n_distinct, uint64_t *n_stored);
/* key_bytes = sizeof(HKMERr) of the files: 2, 4 or 8 (src/main.cc:251-286). */
int mc_builder_write(mc_builder *b, const char *base, int key_bytes);
int mc_builder_close(mc_builder *b);

#ifdef __cplusplus
}
#endif
#endif
