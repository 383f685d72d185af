// reads.hpp -- FASTA/FASTQ indexing, the 2-bit read packer and mate pairing.
//
// Output format is the batch format the GPU kernels consume, unchanged from the
// reference (src/CuCLARK_hh.hh:1615-1715):
//   reads_ptr[i]  = u32 offset of read i in containers[]     (reads_ptr[n] = total)
//   a read        = its parts, one after the other; a part is a maximal run of
//                   ACGTU (either case, '\n' ignored) at least k long:
//                   [length][ceil(length/8) containers], 8 bases per u16, first base
//                   in the high bits, last container left-aligned.
//   reads shorter than k, and parts shorter than k, contribute nothing.
#pragma once

#include "common.hpp"
#include "simd.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

namespace host {

// a growable u64 array that never zero-fills: resizing the five columns of a 2 M-read index through
// std::vector cost a serial 80 MB memset plus its page faults -- a third of the whole run
class Column {
public:
    Column() = default;
    Column(const Column &) = delete;
    Column &operator=(const Column &) = delete;
    Column(Column &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    Column &operator=(Column &&o) noexcept
    {
        if (this != &o) { std::free(p_); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        return *this;
    }
    ~Column() { std::free(p_); }
    size_t size() const { return n_; }
    uint64_t &operator[](size_t i) { return p_[i]; }
    const uint64_t &operator[](size_t i) const { return p_[i]; }
    void reserve(size_t c)
    {
        if (c <= cap_) return;
        void *q = std::realloc(p_, c * sizeof(uint64_t));
        if (!q) throw std::bad_alloc();
        p_ = static_cast<uint64_t *>(q); cap_ = c;
    }
    void push_back(uint64_t v) { if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 1024); p_[n_++] = v; }
    void resize(size_t n) { reserve(n); n_ = n; }          // new elements are NOT initialised
private:
    uint64_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

struct ReadIndex {
    Column name_s, name_e;   // object name = bytes [name_s, name_e)
    Column spos, epos;       // sequence bytes [spos, epos) (may span lines)
    Column len;              // bases, newlines not counted
    size_t size() const { return len.size(); }
};

inline bool is_sep(uint8_t c) { return c == ' ' || c == '\t' || c == '\n'; }

// One pass over the file image.  Record rules as in the reference
// (src/CuCLARK_hh.hh:1340-1404 FASTA, :1476-1533 FASTQ): the name ends at the first
// space/tab/newline; a FASTA sequence runs to the next '>' ; FASTQ records are 4 lines.
template <typename Scan>
inline bool index_reads_with(const uint8_t *t, size_t nb, ReadIndex &R, std::string &err)
{
    if (nb == 0) { err = "empty file"; return false; }
    Scan nl(t, nb);
    {   // most records are a few hundred bytes: fewer reallocations of the five arrays
        const size_t guess = nb / 200 + 16;
        R.name_s.reserve(guess); R.name_e.reserve(guess); R.spos.reserve(guess); R.epos.reserve(guess); R.len.reserve(guess);
    }
    if (t[0] == '>') {
        size_t i = 1;
        while (true) {
            R.name_s.push_back(i);
            while (i < nb && !is_sep(t[++i])) {}      // as the reference: the first byte is never a separator
            R.name_e.push_back(i);
            i = nl.next(i);
            if (i < nb) i++;
            const size_t s = i;
            size_t e = i, lines = 0;
            while (i < nb && t[i] != '>') {
                const size_t j = nl.next(i);
                lines++;
                e = j;
                i = j < nb ? j + 1 : nb;
            }
            R.spos.push_back(s);
            R.epos.push_back(e);
            // bytes minus the newlines inside (the last line's newline is outside [s, e))
            R.len.push_back(e > s ? e - s - (lines ? lines - 1 : 0) : 0);
            if (i >= nb) break;
            i++;   // past '>'
        }
        return true;
    }
    if (t[0] == '@') {
        size_t i = 1;
        while (true) {
            R.name_s.push_back(i);
            while (i < nb && !is_sep(t[++i])) {}      // as the reference: the first byte is never a separator
            R.name_e.push_back(i);
            i = nl.next(i);
            if (i < nb) i++;
            const size_t s = i;
            const size_t e = nl.next(i);
            R.spos.push_back(s);
            R.epos.push_back(e);
            R.len.push_back(e - s);
            i = e < nb ? e + 1 : nb;
            for (int l = 0; l < 2; l++) {          // '+' line and quality line
                const size_t q = nl.next(i);
                i = q < nb ? q + 1 : nb;
            }
            if (i + 1 >= nb) break;
            i++;   // past '@'
        }
        return true;
    }
    err = "Failed to recognize the format of the file.";
    return false;
}

#ifdef MC_HOST_X86
// FASTQ in ONE sweep over the newline bitmaps of 64-byte blocks: the same records as index_reads_with
// (a record is four lines; the byte where its '@' should be is skipped unseen; the name ends at the first
// space/tab/newline BEHIND its first byte, so the line of an empty name does not end the header) without a
// call per line -- half the time per record.  The last partial block and a file that ends inside a record
// go through the same state machine with the end of the text standing in for the missing newlines.
// first separator behind `from`, searched up to `limit` (a newline position, or nb: then nb when there is none)
__attribute__((target("avx2")))
inline size_t name_end_avx2(const uint8_t *t, size_t nb, size_t from, size_t limit)
{
    const __m256i nl = _mm256_set1_epi8(10), sp = _mm256_set1_epi8(' '), tab = _mm256_set1_epi8('\t');
    size_t i = from + 1;
    while (i + 32 <= nb && i <= limit) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + i));
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, nl), _mm256_cmpeq_epi8(v, sp)), _mm256_cmpeq_epi8(v, tab)));
        if (m) return i + (size_t)__builtin_ctz(m);
        i += 32;
    }
    for (; i < nb; i++) if (is_sep(t[i])) return i;
    return nb;
}

__attribute__((target("avx2")))
inline void index_fastq_avx2(const uint8_t *t, size_t nb, ReadIndex &R)
{
    {
        const size_t guess = nb / 200 + 16;
        R.name_s.reserve(guess); R.name_e.reserve(guess); R.spos.reserve(guess); R.epos.reserve(guess); R.len.reserve(guess);
    }
    const __m256i nl = _mm256_set1_epi8(10);
    auto name_end = [&](size_t from, size_t limit) -> size_t { return name_end_avx2(t, nb, from, limit); };
    int line = 0;                 // 0 header, 1 sequence, 2 '+', 3 quality
    size_t ls = 0;                // start of the current line
    size_t name_s = 1;
    bool open = true;             // a record has been started and its header is not complete yet
    auto at_newline = [&](size_t pos) {
        switch (line) {
        case 0:
            if (pos <= name_s) return;                       // the name's first byte is never a separator: the header goes on
            R.name_s.push_back(name_s);
            R.name_e.push_back(name_end(name_s, pos));
            open = false;
            break;
        case 1:
            R.spos.push_back(ls); R.epos.push_back(pos); R.len.push_back(pos - ls);
            break;
        default: break;
        }
        ls = pos + 1;
        line = (line + 1) & 3;
    };
    bool stop = false;
    size_t b = 0;
    for (; b + 64 <= nb && !stop; b += 64) {
        const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + b)), nl));
        const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + b + 32)), nl));
        uint64_t m = (uint64_t)m0 | ((uint64_t)m1 << 32);
        while (m) {
            const size_t pos = b + (size_t)__builtin_ctzll(m);
            m &= m - 1;
            const int was = line;
            at_newline(pos);
            if (was == 3) {                                   // a record is complete: the next one starts at ls, if there is room for one
                if (ls + 1 >= nb) { stop = true; break; }
                name_s = ls + 1;
                open = true;
            }
        }
    }
    if (!stop) {
        for (size_t i = b; i < nb && !stop; i++) {
            if (t[i] != 10) continue;
            const int was = line;
            at_newline(i);
            if (was == 3) {
                if (ls + 1 >= nb) { stop = true; break; }
                name_s = ls + 1;
                open = true;
            }
        }
    }
    if (!stop) {
        // the text ends inside a record: what index_reads_with sees when its newline search returns nb
        if (line == 0 && open) {
            R.name_s.push_back(name_s);
            R.name_e.push_back(name_end(name_s, nb));
            R.spos.push_back(nb); R.epos.push_back(nb); R.len.push_back(0);
        } else if (line == 1) {
            const size_t s0 = ls < nb ? ls : nb;
            R.spos.push_back(s0); R.epos.push_back(nb); R.len.push_back(nb - s0);
        }
    }
}
#endif

#ifdef MC_HOST_X86
// FASTA in one sweep over the same newline bitmaps (records as index_reads_with: a header line -- the name ends at the
// first space/tab/newline BEHIND its first byte, so an empty name swallows the next line -- then sequence lines up to
// the next line that starts with '>'; the length is the bytes of those lines without the newlines between them).
__attribute__((target("avx2")))
inline void index_fasta_avx2(const uint8_t *t, size_t nb, ReadIndex &R)
{
    {
        const size_t guess = nb / 160 + 16;
        R.name_s.reserve(guess); R.name_e.reserve(guess); R.spos.reserve(guess); R.epos.reserve(guess); R.len.reserve(guess);
    }
    const __m256i nl = _mm256_set1_epi8(10);
    bool header = true;           // the record's header line is not complete yet
    size_t name_s = 1, s = 0, e = 0, lines = 0;
    bool done = false;            // the last record was closed by the end of the text
    auto close_record = [&]() {
        R.spos.push_back(s); R.epos.push_back(e);
        R.len.push_back(e > s ? e - s - (lines ? lines - 1 : 0) : 0);
    };
    // returns false when the text is used up
    auto at_newline = [&](size_t pos) -> bool {
        const size_t next = pos + 1;                      // start of the next line
        if (header) {
            if (pos <= name_s) return true;               // the name's first byte is never a separator: the header goes on
            R.name_s.push_back(name_s);
            R.name_e.push_back(name_end_avx2(t, nb, name_s, pos));
            header = false;
            s = next < nb ? next : nb; e = s; lines = 0;
        } else {
            lines++;
            e = pos;
        }
        if (next >= nb) { close_record(); done = true; return false; }
        if (t[next] == '>') {                             // the sequence (possibly empty) ends here
            close_record();
            header = true;
            name_s = next + 1;
        }
        return true;
    };
    size_t b = 0;
    bool more = true;
    for (; b + 64 <= nb && more; b += 64) {
        const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + b)), nl));
        const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + b + 32)), nl));
        uint64_t m = (uint64_t)m0 | ((uint64_t)m1 << 32);
        while (m && more) {
            const size_t pos = b + (size_t)__builtin_ctzll(m);
            m &= m - 1;
            more = at_newline(pos);
        }
    }
    if (more) for (size_t i = b; i < nb && more; i++) if (t[i] == 10) more = at_newline(i);
    if (!done) {
        // the text ends inside a line
        if (header) {
            R.name_s.push_back(name_s);
            R.name_e.push_back(name_end_avx2(t, nb, name_s, nb));
            s = nb; e = nb; lines = 0;
        } else {
            lines++;
            e = nb;
        }
        close_record();
    }
}
#endif

inline bool index_reads(const uint8_t *t, size_t nb, ReadIndex &R, std::string &err)
{
#ifdef MC_HOST_X86
    if (cpu_has_avx2() && nb >= 64 && t[0] == '@' && !getenv("MC_HOST_GENERIC_INDEX")) { index_fastq_avx2(t, nb, R); return true; }
    if (cpu_has_avx2() && nb >= 64 && t[0] == '>' && !getenv("MC_HOST_GENERIC_INDEX")) { index_fasta_avx2(t, nb, R); return true; }
    if (cpu_has_avx2()) return index_reads_with<NewlineScanAvx2>(t, nb, R, err);
#endif
    return index_reads_with<NewlineScanLibc>(t, nb, R, err);
}

// Parallel front end of index_reads: the file is cut into `nthreads` byte ranges, every
// range is moved forward to the next record start and indexed on its own thread, and
// the pieces are concatenated.  Record starts are found without context:
//   FASTA  a '>' that begins a line;
//   FASTQ  a line starting with '@' whose second-next line starts with '+' (a quality
//          line may start with '@', but then the line two below it is a sequence line).
// The reference does the same cut per batch (src/CuCLARK_hh.hh:1345-1365, :1409-1471);
// per-read results do not depend on where the cuts fall.
inline size_t next_line(const uint8_t *t, size_t nb, size_t i)
{
    const void *p = i < nb ? std::memchr(t + i, '\n', nb - i) : nullptr;
    return p ? (size_t)((const uint8_t *)p - t) + 1 : nb;
}

inline size_t record_start_at_or_after(const uint8_t *t, size_t nb, size_t pos, bool fastq)
{
    if (pos == 0) return 0;
    size_t i = next_line(t, nb, pos - 1);          // first line start >= pos
    while (i < nb) {
        if (!fastq) { if (t[i] == '>') return i; }
        else if (t[i] == '@') {
            const size_t l2 = next_line(t, nb, next_line(t, nb, i));
            if (l2 < nb && t[l2] == '+') return i;
        }
        i = next_line(t, nb, i);
    }
    return nb;
}

inline bool index_reads_parallel(const uint8_t *t, size_t nb, int nthreads, ReadIndex &R, std::string &err)
{
    if (nb == 0) { err = "empty file"; return false; }
    if (t[0] != '>' && t[0] != '@') { err = "Failed to recognize the format of the file."; return false; }
    const bool fastq = t[0] == '@';
    if (nthreads < 1) nthreads = 1;
    if (nb < (size_t)(1 << 20) || nthreads == 1) return index_reads(t, nb, R, err);
    std::vector<size_t> cut(nthreads + 1);
    for (int p = 0; p <= nthreads; p++) cut[p] = p == nthreads ? nb : record_start_at_or_after(t, nb, nb / nthreads * p, fastq);
    std::vector<ReadIndex> part(nthreads);
    std::vector<std::string> errs(nthreads);
    std::vector<char> ok(nthreads, 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
#endif
    for (int p = 0; p < nthreads; p++) {
        if (cut[p] >= cut[p + 1]) continue;
        ok[p] = index_reads(t + cut[p], cut[p + 1] - cut[p], part[p], errs[p]);
    }
    size_t total = 0;
    for (int p = 0; p < nthreads; p++) { if (!ok[p]) { err = errs[p]; return false; } total += part[p].size(); }
    R.name_s.resize(total); R.name_e.resize(total); R.spos.resize(total); R.epos.resize(total); R.len.resize(total);
    std::vector<size_t> first(nthreads + 1, 0);
    for (int p = 0; p < nthreads; p++) first[p + 1] = first[p] + part[p].size();
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
#endif
    for (int p = 0; p < nthreads; p++) {
        const size_t n = part[p].size(), off = cut[p], at = first[p];
        for (size_t i = 0; i < n; i++) {
            R.name_s[at + i] = part[p].name_s[i] + off; R.name_e[at + i] = part[p].name_e[i] + off;
            R.spos[at + i] = part[p].spos[i] + off;     R.epos[at + i] = part[p].epos[i] + off;
            R.len[at + i] = part[p].len[i];
        }
    }
    return true;
}

// Worst-case number of containers for reads [r0, r1): one length slot per part (a part
// needs >= k bases) plus one container per 8 bases, rounded up per part.
inline size_t container_bound(const ReadIndex &R, size_t r0, size_t r1, unsigned k)
{
    size_t c = 0;
    for (size_t i = r0; i < r1; i++) {
        const size_t L = R.len[i];
        if (L < k) continue;
        const size_t parts = L / k;
        c += L / 8 + 2 * parts + 2;
    }
    return c + 8;
}

// The parts of the bytes [i, e) of t -- maximal runs of bases at least k long -- appended to con[] at `count`; returns
// the new count.  `avail`: bytes that may be read from t (0 = unknown: no loads past e).
inline size_t pack_segment(const uint8_t *t, size_t i, const size_t e, unsigned k, uint16_t *con, size_t count, size_t avail)
{
    const auto &ct = codes();
    while (i < e) {
        // skip to the start of a run
        while (i < e && ct.r[t[i]] < 0) i++;
        if (i >= e) break;
        const size_t slot = count++;      // length slot of this part
        uint32_t plen = 0, cur = 0;
        uint16_t w = 0;
        {   // whole 32-base blocks of the run, vectorised; the scalar loop takes over at the first other byte
            const size_t done = pack_blocks(t + i, e - i, con + count);
            i += done; count += done / 8; plen += (uint32_t)done;
            // what is left of a run that ends with the sequence (fewer than 32 bases): one masked block
            const size_t rem = e - i;
            if (rem && rem < 32 && i + 32 <= avail && pack_tail(t + i, rem, con + count)) {
                i = e; count += (rem + 7) / 8; plen += (uint32_t)rem;
            }
        }
        while (i < e) {
            const int code = ct.r[t[i]];
            if (code < 0) {
                if (t[i] == '\n') { i++; continue; }
                break;
            }
            w = (uint16_t)((w << 2) | code);
            i++;
            if (++cur == 8) { con[count++] = w; plen += 8; cur = 0; w = 0; }
        }
        if (cur) { con[count++] = (uint16_t)(w << (2 * (8 - cur))); plen += cur; }
        if (plen < k) count = slot;           // too short: drop the part
        else con[slot] = (uint16_t)plen;      // a part is at most 65535 bases (u16 slot)
    }
    return count;
}

// Pack reads [r0, r1) of the file image.  Returns the number of containers written.
// `avail`: bytes that may be read from t (the rest of the file image; 0 = unknown: no loads past a read's end).
inline size_t pack_reads(const uint8_t *t, const ReadIndex &R, size_t r0, size_t r1, unsigned k,
                         uint32_t *ptr, uint16_t *con, size_t avail = 0)
{
    size_t count = 0;
    for (size_t ir = r0; ir < r1; ir++) {
        ptr[ir - r0] = (uint32_t)count;
        if (R.len[ir] < k) continue;
        count = pack_segment(t, R.spos[ir], R.epos[ir], k, con, count, avail);
    }
    ptr[r1 - r0] = (uint32_t)count;
    return count;
}

// Paired mates straight from their two files: read i = the sequence of record i of file 1, an 'N', the sequence of
// record i of file 2 (what the joined record ">id\nR1NR2" of src/file.cc:205-268 packs to: the N ends a part).
// R1.len holds the joined length (len1 + 1 + len2).
inline size_t pack_mates(const uint8_t *t1, const ReadIndex &R1, const uint8_t *t2, const ReadIndex &R2, size_t n, unsigned k,
                         uint32_t *ptr, uint16_t *con, size_t avail1, size_t avail2)
{
    size_t count = 0;
    for (size_t ir = 0; ir < n; ir++) {
        ptr[ir] = (uint32_t)count;
        if (R1.len[ir] < k) continue;
        count = pack_segment(t1, R1.spos[ir], R1.epos[ir], k, con, count, avail1);
        count = pack_segment(t2, R2.spos[ir], R2.epos[ir], k, con, count, avail2);
    }
    ptr[n] = (uint32_t)count;
    return count;
}

} // namespace host
