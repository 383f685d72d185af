/*
 * clark_oracle.h -- CPU restatement of the cuCLARK classification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker.  The product path (jn_cuclark_amd/csrc) never links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - k-mer encode / reverse-complement / canonical / (quotient, remainder) / bucket
 *     lookup / .sz .ky .lb format: PINNED against the reference's own CPU hash table
 *     (src/hashTable_hh.hh, src/HashTableStorage_hh.hh) compiled from /root/reference
 *     into oracle/_ref (oracle/Makefile), fixtures in tests/golden/.
 *   - FASTA/FASTQ indexing, read packing, mate pairing, database build from targets and
 *     CSV formatting: PINNED against the reference's own host driver (src/main.cc +
 *     src/CuCLARK_hh.hh compiled into oracle/_ref/ref_host_mc_* over our backend,
 *     tests/test_ref_host.py: byte-identical database files and CSV).
 *   - per-read k-mer enumeration and scoring, sparse rows, merge, top-2 (the three
 *     kernels of src/CuClarkDB.cu): restated from source; the reference ships no tests,
 *     fixtures or golden vectors for them and CuClarkDB.cu needs nvcc + an NVIDIA GPU,
 *     so for these stages parity is pinned by source reading only ("parity unpinned" by
 *     a reference run).
 *
 * All "ref:" citations are relative to /root/reference/src/.
 */
#ifndef CLARK_ORACLE_H
#define CLARK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k-mer arithmetic -------------------------------------------------- */

/* 2-bit code of a nucleotide byte as the read packer uses it: A=3 C=2 G=1 T/U=0,
 * either case; -1 for anything else.  ref: CuCLARK_hh.hh:294-297 (m_rTable) */
int orc_nt_code(uint8_t c);

/* value of the first k bases of s, first base most significant.
 * ref: kmersConversion.cc:49-68 (getKmers).  returns -1 on a non-ACGT byte. */
int orc_kmer_from_string(const char *s, int k, uint64_t *out);

/* reverse complement of a k-mer value.  ref: CuClarkDB.cu:1196-1203,
 * kmersConversion.cc:39-47 (getReverse) */
uint64_t orc_revcomp(uint64_t x, int k);

/* min(x, revcomp(x)).  ref: CuClarkDB.cu:1206, HashTableStorage_hh.hh:435 */
uint64_t orc_canonical(uint64_t x, int k);

/* ---- database (the arrays CuClarkDB::read produces) --------------------- */

typedef struct orc_db {
    uint64_t  htsize;     /* number of buckets (HTSIZE, parameters.hh:37)          */
    uint64_t  n;          /* number of stored k-mers                               */
    int       key_bytes;  /* sizeof(HKMERr): 2, 4 or 8 (main.cc:251-286)           */
    uint64_t *off;        /* htsize+1 exclusive prefix sums (CuClarkDB.cu:589-617) */
    void     *keys;       /* n quotients, ascending inside a bucket                */
    uint16_t *labels;     /* n 0-based target ids                                  */
} orc_db;

/* Build from the three raw arrays (contents of .sz / .ky / .lb).  Arrays are copied. */
orc_db *orc_db_from_arrays(uint64_t htsize, const uint8_t *sz, const void *keys,
                           int key_bytes, const uint16_t *labels, uint64_t n);

/* Read base.sz/.ky/.lb.  sampling <= 1 keeps every bucket; otherwise every
 * sampling-th NON-EMPTY bucket is kept (ref: CuClarkDB.cu:490-513, :677-739). */
orc_db *orc_db_load(const char *base, uint64_t htsize, int key_bytes, uint32_t sampling);

void orc_db_free(orc_db *db);

/* Write base.sz/.ky/.lb from a list of (canonical k-mer, label) sorted by
 * (kmer % htsize, kmer / htsize).  ref: hashTable_hh.hh:473-546 (write).
 * returns 0, or -1 on I/O error, -2 if a bucket exceeds 255, -3 if unsorted. */
int orc_db_write(const char *base, uint64_t htsize, int key_bytes,
                 const uint64_t *canon_kmers, const uint16_t *labels, uint64_t n);

/* Look up ONE forward k-mer value in the bucket range [part_start, part_end).
 * ref: CuClarkDB.cu:1189-1254 (queryElement), hashTable_hh.hh:358-396 (find). */
int orc_db_lookup(const orc_db *db, int k, uint64_t kmer_fwd,
                  uint64_t part_start, uint64_t part_end, uint16_t *label);

/* ---- discriminative k-mer selection (DB build semantics) ---------------- */

/* In: m (forward k-mer, target id) occurrences in any order.  Out: the canonical
 * k-mers that occur in exactly ONE target (any number of times, count > min_count),
 * sorted by (kmer % htsize, kmer / htsize), with that target id.
 * ref: HashTableStorage_hh.hh:421-461 (addElement), :229-280 (RemoveCommon),
 *      hashTable_hh.hh:398-410 (updateElement).  Outputs sized m.  Returns count. */
uint64_t orc_build_discriminative(const uint64_t *kmers_fwd, const uint16_t *targets,
                                  uint64_t m, int k, uint64_t htsize, uint32_t min_count,
                                  uint64_t *out_canon, uint16_t *out_label);

/* ---- reads ------------------------------------------------------------- */

/* Pack reads the way the host driver does.  text = whole file image;
 * read i occupies text[spos[i] .. epos[i]) and has nominal length len[i]
 * (FASTA: sequence bytes without newlines; FASTQ: the sequence line).
 * Output: reads_ptr[n_reads+1] (u32 container offsets), containers[] (u16).
 * Returns the number of containers written, or (size_t)-1 if cap is too small.
 * ref: CuCLARK_hh.hh:1615-1715. */
size_t orc_pack_reads(const uint8_t *text, const uint64_t *spos, const uint64_t *epos,
                      const uint64_t *len, size_t n_reads, int k,
                      uint32_t *reads_ptr, uint16_t *containers, size_t cap);

/* Index a FASTA ('>') or FASTQ ('@') file image as ONE batch.
 * Outputs (each sized max_reads): name start/end, sequence start/end, length.
 * Returns number of reads, or -1 (unknown format) / -2 (max_reads too small).
 * ref: CuCLARK_hh.hh:1340-1404 (fasta), :1476-1533 (fastq), with m_numBatches = 1. */
long orc_index_reads(const uint8_t *text, size_t nb, size_t max_reads,
                     uint64_t *name_s, uint64_t *name_e,
                     uint64_t *spos, uint64_t *epos, uint64_t *len);

/* ---- per-read scoring --------------------------------------------------- */

/* queryKernel: per read, count k-mer hits per target over all parts, then emit the
 * sparse row [n, t0,h0, t1,h1, ...] in ascending target order.  row_len = 2*maxhits+2
 * u16 per read (ref: CuCLARK_hh.hh:1586-1589).  Reads that hit more than maxhits
 * distinct targets are undefined behaviour in the reference (CuClarkDB.cu:1140-1151);
 * here the row keeps the maxhits SMALLEST target ids and *n_overflow counts such reads.
 * ref: CuClarkDB.cu:999-1183. */
void orc_query_batch(const orc_db *db, int k, uint32_t num_targets,
                     const uint32_t *reads_ptr, const uint16_t *containers, size_t n_reads,
                     uint64_t part_start, uint64_t part_end,
                     uint16_t *rows, size_t row_len, uint64_t *n_overflow);

/* mergeKernel: sorted-row union with count addition.  ref: CuClarkDB.cu:1261-1355.
 * Rows whose union exceeds (row_len-2)/2 targets keep the smallest ids (see above). */
void orc_merge_rows(const uint16_t *a, const uint16_t *b, size_t row_len, size_t n_reads,
                    uint16_t *out);

/* resultKernel: [sumN, idxBest+1, best, idxSecond+1, second] per read.
 * ref: CuClarkDB.cu:1361-1411. */
void orc_result_rows(const uint16_t *rows, size_t row_len, size_t n_reads, uint16_t *out5);

/* Whole path for one unsharded DB: query + result.  OpenMP over reads when built
 * with -fopenmp (this is the bench.py cpu_baseline "port"). */
void orc_classify_batch(const orc_db *db, int k, uint32_t num_targets, uint32_t maxhits,
                        const uint32_t *reads_ptr, const uint16_t *containers, size_t n_reads,
                        uint16_t *out5, uint64_t *n_overflow);

int orc_num_threads(void);
void orc_set_num_threads(int n);

/* ---- CSV line ------------------------------------------------------------ */

/* One non-extended result line "name,gamma,assignment,best,confidence\n".
 * name_len is clipped to 39 bytes; norm_len = read length (paired: length - 1).
 * ref: CuCLARK_hh.hh:2096-2118, parameters.hh:46-47.  Returns bytes written. */
int orc_csv_line(char *dst, size_t cap, const char *name, size_t name_len,
                 const uint16_t res5[5], uint64_t norm_len, int k, const char *assignment);

#ifdef __cplusplus
}
#endif
#endif
