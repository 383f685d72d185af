/*
 * clark_oracle.c -- CPU restatement of the cuCLARK classification path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY -- see clark_oracle.h for the rules and the parity status.
 * Every function cites the reference lines it restates ("ref:", relative to
 * /root/reference/src/).  Written for clarity first; the only tuned piece is the
 * OpenMP loop of orc_classify_batch, which bench.py times as the CPU baseline.
 */
#include "clark_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* k-mer arithmetic                                                          */
/* ------------------------------------------------------------------------ */

/* ref: CuCLARK_hh.hh:294-297 -- m_rTable: A/a=3 C/c=2 G/g=1 T/t/U/u=0 */
int orc_nt_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 3;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 1;
    case 'T': case 't': case 'U': case 'u': return 0;
    default: return -1;
    }
}

/* ref: kmersConversion.cc:49-68 (only ACGT in either case are accepted there) */
int orc_kmer_from_string(const char *s, int k, uint64_t *out)
{
    uint64_t v = 0;
    for (int i = 0; i < k; i++) {
        int c = orc_nt_code((uint8_t)s[i]);
        if (c < 0 || s[i] == 'U' || s[i] == 'u') return -1;
        v = (v << 2) ^ (uint64_t)c;
    }
    *out = v;
    return 0;
}

/* ref: CuClarkDB.cu:1196-1203 -- reverse the order of the 2-bit groups, complement
 * every group (x -> 3-x), drop the 64-2k unused low bits. */
uint64_t orc_revcomp(uint64_t x, int k)
{
    uint64_t r = x;
    r = ((r >> 2)  & 0x3333333333333333ULL) | ((r & 0x3333333333333333ULL) << 2);
    r = ((r >> 4)  & 0x0F0F0F0F0F0F0F0FULL) | ((r & 0x0F0F0F0F0F0F0F0FULL) << 4);
    r = ((r >> 8)  & 0x00FF00FF00FF00FFULL) | ((r & 0x00FF00FF00FF00FFULL) << 8);
    r = ((r >> 16) & 0x0000FFFF0000FFFFULL) | ((r & 0x0000FFFF0000FFFFULL) << 16);
    r = (r >> 32) | (r << 32);
    r = (~(uint64_t)0 - r) >> (64 - (k << 1));
    return r;
}

/* ref: CuClarkDB.cu:1206 */
uint64_t orc_canonical(uint64_t x, int k)
{
    uint64_t r = orc_revcomp(x, k);
    return x < r ? x : r;
}

/* ------------------------------------------------------------------------ */
/* database                                                                  */
/* ------------------------------------------------------------------------ */

static uint64_t key_at(const orc_db *db, uint64_t i)
{
    switch (db->key_bytes) {
    case 2:  return ((const uint16_t *)db->keys)[i];
    case 4:  return ((const uint32_t *)db->keys)[i];
    default: return ((const uint64_t *)db->keys)[i];
    }
}

orc_db *orc_db_from_arrays(uint64_t htsize, const uint8_t *sz, const void *keys,
                           int key_bytes, const uint16_t *labels, uint64_t n)
{
    orc_db *db = (orc_db *)calloc(1, sizeof(orc_db));
    if (!db) return NULL;
    db->htsize = htsize;
    db->key_bytes = key_bytes;
    db->off = (uint64_t *)malloc((htsize + 1) * sizeof(uint64_t));
    if (!db->off) { free(db); return NULL; }
    /* ref: CuClarkDB.cu:589-617 -- exclusive prefix sums of the bucket sizes */
    uint64_t acc = 0;
    for (uint64_t b = 0; b < htsize; b++) { db->off[b] = acc; acc += sz[b]; }
    db->off[htsize] = acc;
    if (acc != n) { free(db->off); free(db); return NULL; }
    db->n = n;
    db->keys = malloc(n ? n * (size_t)key_bytes : 1);
    db->labels = (uint16_t *)malloc(n ? n * sizeof(uint16_t) : 1);
    if (!db->keys || !db->labels) { orc_db_free(db); return NULL; }
    memcpy(db->keys, keys, n * (size_t)key_bytes);
    memcpy(db->labels, labels, n * sizeof(uint16_t));
    return db;
}

/* ref: CuClarkDB.cu:463-770 (read).  The part split of the reference (device memory
 * budget, :526-559) does not change any answer, so the oracle keeps one part. */
orc_db *orc_db_load(const char *base, uint64_t htsize, int key_bytes, uint32_t sampling)
{
    size_t bl = strlen(base);
    char *p = (char *)malloc(bl + 4);
    if (!p) return NULL;
    FILE *fs, *fk, *fl;
    sprintf(p, "%s.sz", base); fs = fopen(p, "rb");
    sprintf(p, "%s.ky", base); fk = fopen(p, "rb");
    sprintf(p, "%s.lb", base); fl = fopen(p, "rb");
    free(p);
    if (!fs || !fk || !fl) {
        if (fs) fclose(fs);
        if (fk) fclose(fk);
        if (fl) fclose(fl);
        return NULL;
    }
    uint8_t *sz = (uint8_t *)malloc(htsize);
    orc_db *db = NULL;
    if (!sz || fread(sz, 1, htsize, fs) != htsize) goto done;

    /* ref: CuClarkDB.cu:490-513 -- choice: keep every bucket, or every sampling-th
     * NON-EMPTY one (counter counts non-empty buckets, kept when counter % s == 0) */
    const int all = sampling <= 1;
    uint64_t n_file = 0, n_keep = 0, nonzero = 0;
    uint8_t *keep = (uint8_t *)malloc(htsize);
    if (!keep) goto done;
    for (uint64_t b = 0; b < htsize; b++) {
        keep[b] = 0;
        if (sz[b] > 0) {
            nonzero++;
            keep[b] = (all || (nonzero % sampling) == 0) ? 1 : 0;
            n_file += sz[b];
            if (keep[b]) n_keep += sz[b];
        }
    }
    if (all) {
        /* every bucket is kept: the two arrays are the files as they are (one read each, no second copy -- a
         * full-size table is 39 GB of them: tests/test_gpu_filesize.py) */
        db = (orc_db *)calloc(1, sizeof(orc_db));
        if (db) {
            db->htsize = htsize; db->key_bytes = key_bytes; db->n = n_file;
            db->off = (uint64_t *)malloc((htsize + 1) * sizeof(uint64_t));
            db->keys = malloc(n_file ? n_file * (size_t)key_bytes : 1);
            db->labels = (uint16_t *)malloc(n_file ? n_file * sizeof(uint16_t) : 1);
            int ok = db->off && db->keys && db->labels;
            if (ok) {
                uint64_t acc = 0;       /* ref: CuClarkDB.cu:589-617 -- exclusive prefix sums of the bucket sizes */
                for (uint64_t b = 0; b < htsize; b++) { db->off[b] = acc; acc += sz[b]; }
                db->off[htsize] = acc;
                ok = fread(db->keys, (size_t)key_bytes, n_file, fk) == n_file && fread(db->labels, 2, n_file, fl) == n_file;
            }
            if (!ok) { orc_db_free(db); db = NULL; }
        }
    } else {
        void *keys = malloc(n_keep ? n_keep * (size_t)key_bytes : 1);
        uint16_t *labels = (uint16_t *)malloc(n_keep ? n_keep * 2 : 1);
        uint8_t *sz_kept = (uint8_t *)malloc(htsize);
        int ok = keys && labels && sz_kept;
        uint64_t w = 0;
        /* ref: CuClarkDB.cu:677-739 -- stream keys/labels, skipping unchosen buckets */
        for (uint64_t b = 0; ok && b < htsize; b++) {
            sz_kept[b] = keep[b] ? sz[b] : 0;
            if (sz[b] == 0) continue;
            if (keep[b]) {
                ok = fread((char *)keys + w * (size_t)key_bytes, (size_t)key_bytes, sz[b], fk) == sz[b]
                  && fread(labels + w, 2, sz[b], fl) == sz[b];
                w += sz[b];
            } else {
                ok = fseek(fk, (long)sz[b] * key_bytes, SEEK_CUR) == 0
                  && fseek(fl, (long)sz[b] * 2, SEEK_CUR) == 0;
            }
        }
        if (ok) db = orc_db_from_arrays(htsize, sz_kept, keys, key_bytes, labels, n_keep);
        free(keys); free(labels); free(sz_kept);
    }
    free(keep);
    (void)n_file;
done:
    free(sz);
    fclose(fs); fclose(fk); fclose(fl);
    return db;
}

void orc_db_free(orc_db *db)
{
    if (!db) return;
    free(db->off); free(db->keys); free(db->labels); free(db);
}

/* ref: hashTable_hh.hh:473-546 -- one size byte per bucket, then per stored element
 * sizeof(HKMERr) bytes of quotient (.ky) and 2 bytes of label (.lb), bucket by bucket,
 * elements ascending inside a bucket (sortall, :203-216). */
int orc_db_write(const char *base, uint64_t htsize, int key_bytes,
                 const uint64_t *canon, const uint16_t *labels, uint64_t n)
{
    size_t bl = strlen(base);
    char *p = (char *)malloc(bl + 4);
    if (!p) return -1;
    FILE *fs, *fk, *fl;
    sprintf(p, "%s.sz", base); fs = fopen(p, "wb");
    sprintf(p, "%s.ky", base); fk = fopen(p, "wb");
    sprintf(p, "%s.lb", base); fl = fopen(p, "wb");
    free(p);
    int rc = 0;
    if (!fs || !fk || !fl) { rc = -1; goto out; }
    enum { CH = 1 << 20 };
    uint8_t *buf = (uint8_t *)calloc(CH, 1);
    if (!buf) { rc = -1; goto out; }
    uint64_t i = 0;
    for (uint64_t b0 = 0; b0 < htsize && rc == 0; b0 += CH) {
        uint64_t b1 = b0 + CH < htsize ? b0 + CH : htsize;
        memset(buf, 0, CH);
        while (i < n && canon[i] % htsize < b1) {
            uint64_t r = canon[i] % htsize, q = canon[i] / htsize;
            if (r < b0) { rc = -3; break; }
            if (i > 0 && canon[i - 1] % htsize == r && canon[i - 1] / htsize >= q) { rc = -3; break; }
            if (buf[r - b0] == 255) { rc = -2; break; }
            buf[r - b0]++;
            switch (key_bytes) {
            case 2: { uint16_t v = (uint16_t)q; fwrite(&v, 2, 1, fk); break; }
            case 4: { uint32_t v = (uint32_t)q; fwrite(&v, 4, 1, fk); break; }
            default: fwrite(&q, 8, 1, fk);
            }
            fwrite(&labels[i], 2, 1, fl);
            i++;
        }
        if (rc == 0 && fwrite(buf, 1, b1 - b0, fs) != b1 - b0) rc = -1;
    }
    if (rc == 0 && i != n) rc = -3;
    free(buf);
out:
    if (fs) fclose(fs);
    if (fk) fclose(fk);
    if (fl) fclose(fl);
    return rc;
}

/* ref: CuClarkDB.cu:1189-1254.  Same order of tests as the device function:
 * canonical, quotient/remainder, part-range filter, empty bucket, first/last-key
 * range test, ascending scan until key > quotient. */
int orc_db_lookup(const orc_db *db, int k, uint64_t kmer_fwd,
                  uint64_t part_start, uint64_t part_end, uint16_t *label)
{
    uint64_t c = orc_canonical(kmer_fwd, k);
    uint64_t quotient = c / db->htsize;
    uint64_t remainder = c - quotient * db->htsize;
    if (remainder < part_start || remainder >= part_end) return 0;
    uint64_t b = db->off[remainder], e = db->off[remainder + 1];
    if (e == b) return 0;
    uint64_t key = key_at(db, b);
    if (key > quotient || key_at(db, e - 1) < quotient) return 0;
    uint64_t i = b;
    while (key <= quotient) {
        if (key == quotient) { *label = db->labels[i]; return 1; }
        key = key_at(db, ++i);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* discriminative k-mers                                                      */
/* ------------------------------------------------------------------------ */

typedef struct { uint64_t r, q; uint16_t t; } rqt;

static int cmp_rqt(const void *a, const void *b)
{
    const rqt *x = (const rqt *)a, *y = (const rqt *)b;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    if (x->q != y->q) return x->q < y->q ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return 0;
}

/* ref: HashTableStorage_hh.hh:421-461 -- every occurrence is canonicalised and either
 * inserted (multiplicity 1) or, if present, gets its count raised and -- when the
 * label differs -- its multiplicity raised (hashTable_hh.hh:398-410).  RemoveCommon
 * (:229-280) keeps elements with multiplicity == 1 and count > min_count.
 * Sorting by (r, q, target) makes both tests a run scan. */
uint64_t orc_build_discriminative(const uint64_t *kmers_fwd, const uint16_t *targets,
                                  uint64_t m, int k, uint64_t htsize, uint32_t min_count,
                                  uint64_t *out_canon, uint16_t *out_label)
{
    rqt *v = (rqt *)malloc((m ? m : 1) * sizeof(rqt));
    if (!v) return 0;
    for (uint64_t i = 0; i < m; i++) {
        uint64_t c = orc_canonical(kmers_fwd[i], k);
        v[i].r = c % htsize; v[i].q = c / htsize; v[i].t = targets[i];
    }
    qsort(v, m, sizeof(rqt), cmp_rqt);
    uint64_t w = 0;
    for (uint64_t i = 0; i < m;) {
        uint64_t j = i;
        int multi = 0;
        while (j < m && v[j].r == v[i].r && v[j].q == v[i].q) {
            if (v[j].t != v[i].t) multi = 1;
            j++;
        }
        if (!multi && (j - i) > min_count) {
            out_canon[w] = v[i].q * htsize + v[i].r;
            out_label[w] = v[i].t;
            w++;
        }
        i = j;
    }
    free(v);
    return w;
}

/* ------------------------------------------------------------------------ */
/* reads                                                                     */
/* ------------------------------------------------------------------------ */

/* ref: CuCLARK_hh.hh:1615-1715.  Kept structurally close to the original loop on
 * purpose (same variables: partBegin, containerCount, curNucs, newSeq), because the
 * handling of too-short parts is a side effect of how the length slot is reused:
 *  - a read shorter than k is skipped entirely (:1632);
 *  - a part (maximal run of ACGTU, newlines ignored, :1673-1677) starts with a length
 *    slot; when the NEXT part starts and the previous one was shorter than k, the
 *    previous part's slot and containers are overwritten (:1646-1650);
 *  - at the end of the read a last part shorter than k is dropped (:1700-1703);
 *  - 8 bases per u16, first base in the high bits, a trailing partial container is
 *    left-aligned (:1660-1661, :1682, :1694).                                      */
size_t orc_pack_reads(const uint8_t *text, const uint64_t *spos, const uint64_t *epos,
                      const uint64_t *len, size_t n_reads, int k,
                      uint32_t *reads_ptr, uint16_t *con, size_t cap)
{
    const size_t nucs_per = 8;
    size_t count = 0;
    for (size_t ir = 0; ir < n_reads; ir++) {
        uint16_t kc = 0;
        size_t cur = 0;
        int new_seq = 1;
        size_t i_c = len[ir] < (uint64_t)k ? epos[ir] : spos[ir];
        size_t part_begin = count;
        reads_ptr[ir] = (uint32_t)part_begin;
        /* worst case for this read: one slot per part + one container per 8 bases */
        if (count + (epos[ir] - i_c) + 2 > cap) return (size_t)-1;
        while (i_c < epos[ir]) {
            int code = orc_nt_code(text[i_c]);
            if (code >= 0) {
                if (new_seq) {
                    con[count++] = 0;
                    if (con[part_begin] < k) {
                        count = part_begin + 1;
                        con[part_begin] = 0;
                    } else {
                        part_begin = count - 1;
                    }
                    new_seq = 0;
                }
                kc = (uint16_t)((kc << 2) ^ code);
                cur++;
                i_c++;
                if (cur == nucs_per) {
                    con[count++] = kc;
                    con[part_begin] = (uint16_t)(con[part_begin] + cur);
                    cur = 0;
                }
                continue;
            }
            if (text[i_c] == '\n') { i_c++; continue; }
            if (cur > 0) {
                kc = (uint16_t)(kc << (2 * (nucs_per - cur)));
                con[count++] = kc;
                con[part_begin] = (uint16_t)(con[part_begin] + cur);
            }
            kc = 0; cur = 0; new_seq = 1;
            i_c++;
        }
        if (cur > 0) {
            kc = (uint16_t)(kc << (2 * (nucs_per - cur)));
            con[count++] = kc;
            con[part_begin] = (uint16_t)(con[part_begin] + cur);
        }
        /* ref :1700-1703.  When the read produced nothing (count == part_begin) the
         * reference reads a stale slot here; either outcome leaves count unchanged. */
        if (count > part_begin && con[part_begin] < k) count = part_begin;
    }
    reads_ptr[n_reads] = (uint32_t)count;
    return count;
}

static int is_sep(uint8_t c) { return c == ' ' || c == '\t' || c == '\n'; }

/* ref: CuCLARK_hh.hh:1340-1404 (FASTA) and :1476-1533 (FASTQ), one batch.
 * FASTA: name = bytes after '>' up to the first separator; sequence runs to the next
 * '>' (or EOF); length = bytes minus newlines (:1385-1389).
 * FASTQ: 4-line records; length = length of line 2 (:1513). */
long orc_index_reads(const uint8_t *t, size_t nb, size_t max_reads,
                     uint64_t *name_s, uint64_t *name_e,
                     uint64_t *spos, uint64_t *epos, uint64_t *len)
{
    size_t n = 0;
    if (nb == 0) return -1;
    if (t[0] == '>') {
        size_t i = 0;
        while (t[i++] != '>') {}
        for (;;) {
            if (n >= max_reads) return -2;
            size_t lines = 0;
            name_s[n] = i;
            while (i < nb && !is_sep(t[++i])) {}
            name_e[n] = i;
            while (i < nb && t[i++] != '\n') {}
            spos[n] = i; epos[n] = i;
            while (i < nb && t[i] != '>') {
                while (i < nb && t[i] != '\n') i++;
                lines++;
                epos[n] = i++;
            }
            /* :1388-1389 length = -(#lines) + epos - spos + 1 */
            len[n] = (uint64_t)((long)epos[n] - (long)spos[n] + 1 - (long)lines);
            n++;
            if (i >= nb) break;
            i++;
        }
        return (long)n;
    }
    if (t[0] == '@') {
        size_t i = 1;
        for (;;) {
            if (n >= max_reads) return -2;
            name_s[n] = i;
            while (i < nb && !is_sep(t[++i])) {}
            name_e[n] = i;
            while (i < nb && t[i++] != '\n') {}
            spos[n] = i; epos[n] = i;
            while (i < nb && t[i] != '\n') i++;
            epos[n] = i++;
            len[n] = epos[n] - spos[n];
            while (i < nb && t[i++] != '\n') {}
            while (i < nb && t[i++] != '\n') {}
            n++;
            if (++i >= nb) break;
        }
        return (long)n;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* scoring                                                                   */
/* ------------------------------------------------------------------------ */

/* k-mer starting at base position p of a part whose containers start at c[0].
 * ref: CuClarkDB.cu:1061-1083 -- shift whole containers in, then the remaining
 * bases of the next one, then mask to 2k bits. */
static uint64_t kmer_at(const uint16_t *c, uint32_t p, int k)
{
    uint64_t v = 0;
    for (int j = 0; j < k; j++) {
        uint32_t b = p + (uint32_t)j;
        uint32_t code = (c[b >> 3] >> (2 * (7 - (b & 7)))) & 3u;
        v = (v << 2) | code;
    }
    return v;
}

/* Count hits per target for one read into hits[] (dense, caller-zeroed, size
 * num_targets); touched[] collects the distinct targets.  ref: CuClarkDB.cu:1042-1117 */
static uint32_t score_read(const orc_db *db, int k,
                           const uint16_t *con, uint32_t beg, uint32_t end,
                           uint64_t ps, uint64_t pe, uint32_t *hits, uint16_t *touched)
{
    uint32_t nt = 0;
    uint32_t pp = beg;
    while (pp < end) {
        uint32_t plen = con[pp];
        const uint16_t *c = con + pp + 1;
        pp += 1 + (plen ? (plen - 1) / 8 + 1 : 0);
        if (plen < (uint32_t)k) continue;
        uint32_t nk = plen - (uint32_t)k + 1;
        uint64_t x = kmer_at(c, 0, k);
        const uint64_t mask = k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
        for (uint32_t p = 0; p < nk; p++) {
            if (p) {
                uint32_t b = p + (uint32_t)k - 1;
                uint32_t code = (c[b >> 3] >> (2 * (7 - (b & 7)))) & 3u;
                x = ((x << 2) | code) & mask;
            }
            uint16_t lab;
            if (orc_db_lookup(db, k, x, ps, pe, &lab)) {
                if (hits[lab]++ == 0) touched[nt++] = lab;
            }
        }
    }
    return nt;
}

static int cmp_u16(const void *a, const void *b)
{
    return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

/* ref: CuClarkDB.cu:999-1183 */
void orc_query_batch(const orc_db *db, int k, uint32_t num_targets,
                     const uint32_t *reads_ptr, const uint16_t *containers, size_t n_reads,
                     uint64_t part_start, uint64_t part_end,
                     uint16_t *rows, size_t row_len, uint64_t *n_overflow)
{
    const uint32_t maxhits = (uint32_t)((row_len - 2) / 2);
    uint64_t ovf = 0;
    (void)num_targets; /* labels are u16, so the dense counter array is sized 65536 */
    uint32_t *hits = (uint32_t *)calloc(65536, sizeof(uint32_t));
    uint16_t *touched = (uint16_t *)malloc(65536 * sizeof(uint16_t));
    for (size_t r = 0; r < n_reads; r++) {
        uint16_t *row = rows + r * row_len;
        memset(row, 0, row_len * sizeof(uint16_t));
        uint32_t nt = score_read(db, k, containers, reads_ptr[r], reads_ptr[r + 1],
                                 part_start, part_end, hits, touched);
        /* ascending target order, as the ballot compaction produces it (:1130-1182) */
        qsort(touched, nt, sizeof(uint16_t), cmp_u16);
        uint32_t keep = nt;
        if (keep > maxhits) { keep = maxhits; ovf++; }
        row[0] = (uint16_t)keep;
        for (uint32_t i = 0; i < keep; i++) {
            row[1 + 2 * i] = touched[i];
            /* counts saturate at 65535 in every output (ours, DESIGN.md 7: the reference's packed
             * u16 atomics carry into the neighbouring target there, :1104-1108) */
            row[2 + 2 * i] = (uint16_t)(hits[touched[i]] > 0xFFFFu ? 0xFFFFu : hits[touched[i]]);
        }
        for (uint32_t i = 0; i < nt; i++) hits[touched[i]] = 0;
    }
    free(hits); free(touched);
    if (n_overflow) *n_overflow = ovf;
}

/* ref: CuClarkDB.cu:1261-1355 -- two-pointer merge, equal targets add their counts */
void orc_merge_rows(const uint16_t *a, const uint16_t *b, size_t row_len, size_t n_reads,
                    uint16_t *out)
{
    const uint32_t maxhits = (uint32_t)((row_len - 2) / 2);
    for (size_t r = 0; r < n_reads; r++) {
        const uint16_t *ra = a + r * row_len, *rb = b + r * row_len;
        uint16_t *ro = out + r * row_len;
        uint32_t na = ra[0], nb = rb[0], ia = 0, ib = 0, n = 0;
        uint16_t tmp[2 * 64 + 2];
        uint16_t *w = (out == a || out == b) ? tmp : ro;
        memset(tmp, 0, sizeof tmp);
        while ((ia < na || ib < nb) && n < maxhits) {
            uint16_t t, h;
            if (ib >= nb || (ia < na && ra[1 + 2 * ia] < rb[1 + 2 * ib])) {
                t = ra[1 + 2 * ia]; h = ra[2 + 2 * ia]; ia++;
            } else if (ia >= na || rb[1 + 2 * ib] < ra[1 + 2 * ia]) {
                t = rb[1 + 2 * ib]; h = rb[2 + 2 * ib]; ib++;
            } else {
                uint32_t s = (uint32_t)ra[2 + 2 * ia] + rb[2 + 2 * ib];      /* saturating, as above */
                t = ra[1 + 2 * ia]; h = (uint16_t)(s > 0xFFFFu ? 0xFFFFu : s); ia++; ib++;
            }
            w[1 + 2 * n] = t; w[2 + 2 * n] = h; n++;
        }
        w[0] = (uint16_t)n;
        if (w == tmp) memcpy(ro, tmp, row_len * sizeof(uint16_t));
        else for (size_t i = 1 + 2 * (size_t)n; i < row_len; i++) ro[i] = 0;
    }
}

/* ref: CuClarkDB.cu:1361-1411 -- ascending scan, strict '>' for best and second best,
 * u16 arithmetic throughout (sumN wraps). */
void orc_result_rows(const uint16_t *rows, size_t row_len, size_t n_reads, uint16_t *out5)
{
    for (size_t r = 0; r < n_reads; r++) {
        const uint16_t *row = rows + r * row_len;
        uint16_t best = 0, s_best = 0, ibest = 0, isbest = 0, sum = 0;
        uint16_t count = row[0];
        for (uint32_t i = 0; i < count; i++) {
            uint16_t sc = row[2 * i + 2];
            if (sc > best) {
                s_best = best; isbest = ibest;
                best = sc; ibest = (uint16_t)(row[2 * i + 1] + 1);
            } else if (sc > s_best) {
                s_best = sc; isbest = (uint16_t)(row[2 * i + 1] + 1);
            }
            sum = (uint16_t)(sum + sc);
        }
        uint16_t *o = out5 + r * 5;
        o[0] = sum; o[1] = ibest; o[2] = best; o[3] = isbest; o[4] = s_best;
    }
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* query + result for an unsharded DB; OpenMP schedule(dynamic) over reads mirrors the
 * reference's only host-side parallel loop (CuCLARK_hh.hh:1608-1615). */
void orc_classify_batch(const orc_db *db, int k, uint32_t num_targets, uint32_t maxhits,
                        const uint32_t *reads_ptr, const uint16_t *containers, size_t n_reads,
                        uint16_t *out5, uint64_t *n_overflow)
{
    const size_t row_len = 2 * (size_t)maxhits + 2;
    uint64_t ovf_total = 0;
#ifdef _OPENMP
#pragma omp parallel reduction(+ : ovf_total)
#endif
    {
        enum { CH = 256 };
        uint16_t *rows = (uint16_t *)malloc(CH * row_len * sizeof(uint16_t));
        long nchunks = (long)((n_reads + CH - 1) / CH);
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (long c = 0; c < nchunks; c++) {
            size_t r0 = (size_t)c * CH;
            size_t nr = n_reads - r0 < CH ? n_reads - r0 : CH;
            uint64_t ovf = 0;
            orc_query_batch(db, k, num_targets, reads_ptr + r0, containers, nr,
                            0, db->htsize, rows, row_len, &ovf);
            orc_result_rows(rows, row_len, nr, out5 + r0 * 5);
            ovf_total += ovf;
        }
        free(rows);
    }
    if (n_overflow) *n_overflow = ovf_total;
}

/* ------------------------------------------------------------------------ */
/* CSV                                                                       */
/* ------------------------------------------------------------------------ */

/* ref: CuCLARK_hh.hh:2096-2118.  gamma = total / (norm_len - k + 1);
 * confidence = best / (best + second), 0 when that sum is below 0.001;
 * "%s,%g,%s,%u,%g\n".  Object names are clipped to OBJECTNAMEMAX-1 = 39 bytes. */
int orc_csv_line(char *dst, size_t cap, const char *name, size_t name_len,
                 const uint16_t res5[5], uint64_t norm_len, int k, const char *assignment)
{
    char nm[40];
    if (name_len >= 40) name_len = 39;
    memcpy(nm, name, name_len);
    nm[name_len] = '\0';
    uint32_t total = res5[0], best = res5[2], s_best = res5[4];
    double gamma = (double)total / (((double)(uint32_t)norm_len - (double)k) + 1.0);
    double delta = (double)(best + s_best);
    delta = (delta < 0.001) ? 0 : ((double)best) / delta;
    return snprintf(dst, cap, "%s,%g,%s,%u,%g\n", nm, gamma, assignment, best, delta);
}
