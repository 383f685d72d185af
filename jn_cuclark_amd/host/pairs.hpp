// pairs.hpp -- the mate join of input.hpp on several threads (needs the record-start search of reads.hpp).
#pragma once

#include "input.hpp"
#include "reads.hpp"

#include <atomic>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

namespace host {

namespace pairs_detail {

// body(i) for i in [0, n) on `threads` plain threads (work handed out one index at a time)
template <class F>
inline void for_each_index(int n, int threads, F &&body)
{
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    auto work = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= n) break; body(i); } };
    for (int w = 1; w < threads; w++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

struct Range { size_t begin = 0, end = 0, records = 0; bool regular = true; };

// records of the byte range [begin, end) of a FASTQ text that starts at a record start: counted four lines at a
// time; `regular` is false when a record does not look like one or the range does not end after a fourth line
inline void count_records(const uint8_t *t, size_t nb, Range &R)
{
    size_t i = R.begin, n = 0;
    bool ok = true;
    while (i < R.end) {
        const size_t l1 = next_line(t, nb, i);        // sequence line
        const size_t l2 = next_line(t, nb, l1);       // '+' line
        const size_t l3 = next_line(t, nb, l2);       // quality line
        if (t[i] != '@' || l1 >= nb || l2 >= nb || l3 >= nb || t[l2] != '+') { ok = false; break; }
        i = next_line(t, nb, l3);
        n++;
    }
    R.records = n;
    R.regular = ok && i == R.end;
}

// byte offset of the record `skip` records behind `from` (a record start)
inline size_t skip_records(const uint8_t *t, size_t nb, size_t from, size_t skip)
{
    size_t i = from;
    for (size_t r = 0; r < skip; r++)
        for (int l = 0; l < 4; l++) i = next_line(t, nb, i);
    return i;
}

} // namespace pairs_detail

// the id of a FASTQ header line as mergePairedFiles cuts it (src/file.cc:230-247: leading ' ', '/', tab, '@' skipped,
// then up to the next of them); [line, end) = the line without its newline
inline void mate_id(const uint8_t *t, size_t line, size_t end, size_t &s, size_t &e)
{
    auto sep = [](uint8_t c) { return c == ' ' || c == '/' || c == '\t' || c == '@'; };
    size_t i = line;
    while (i < end && sep(t[i])) i++;
    size_t j = i;
    while (j < end && !sep(t[j])) j++;
    s = i; e = j;
}
inline size_t line_end(const uint8_t *t, size_t nb, size_t from)
{
    const void *p = from < nb ? std::memchr(t + from, '\n', nb - from) : nullptr;
    return p ? (size_t)((const uint8_t *)p - t) : nb;
}
// the record of file 2 whose id equals the id of file 1's record at `at1`, looked for around the same relative
// position of file 2 (mates come in the same order; equal read lengths put them at the same fraction exactly)
inline bool find_mate(const uint8_t *a, size_t na, size_t at1, const uint8_t *b, size_t nb, size_t &at2)
{
    size_t s1, e1;
    mate_id(a, at1, line_end(a, na, at1), s1, e1);
    const size_t guess = (size_t)((unsigned __int128)nb * at1 / (na ? na : 1));
    size_t window = 256u << 10;
    if (const char *e = getenv("MC_MATE_WINDOW")) window = (size_t)std::strtoull(e, nullptr, 10);      // (tests)
    size_t i = record_start_at_or_after(b, nb, guess > window ? guess - window : 0, true);
    const size_t stop = std::min(nb, guess + window);
    while (i < stop) {
        size_t s2, e2;
        mate_id(b, i, line_end(b, nb, i), s2, e2);
        if (e2 - s2 == e1 - s1 && std::memcmp(a + s1, b + s2, e1 - s1) == 0) {
            if (getenv("MC_DEBUG_MATES")) std::cerr << "find_mate: at1 " << at1 << " guess " << guess << " window " << window << " found at " << i << "\n";
            at2 = i; return true;
        }
        for (int l = 0; l < 4; l++) i = next_line(b, nb, i);
    }
    return false;
}

// Where the records that start at cutA[i] in file 1 have their mates in file 2, by counting: the records of every range
// of file 1 and of as many byte ranges of file 2 are counted on `n_threads` threads, and the mate of the first record of
// range i is the record with the same number.  false: the files are not regular (a record that is not four lines, or
// different record counts).  cutA: ascending record starts, cutA.front() = 0, cutA.back() = na.
inline bool align_mates(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, const std::vector<size_t> &cutA,
                        int n_threads, std::vector<size_t> &cutB)
{
    using namespace pairs_detail;
    const int P = (int)cutA.size() - 1;
    if (P < 1) return false;
    std::vector<Range> RA(P), RB(P);
    for (int i = 0; i < P; i++) {
        RA[i].begin = cutA[i]; RA[i].end = cutA[i + 1];
        RB[i].begin = record_start_at_or_after(b, nb, nb / P * i, true);
    }
    for (int i = 0; i < P; i++) RB[i].end = i + 1 < P ? RB[i + 1].begin : nb;
    for_each_index(2 * P, n_threads, [&](int i) {
        if (i < P) count_records(a, na, RA[i]); else count_records(b, nb, RB[i - P]);
    });
    std::vector<size_t> firstA(P + 1, 0), firstB(P + 1, 0);
    bool regular = true;
    for (int i = 0; i < P; i++) {
        regular = regular && RA[i].regular && RB[i].regular;
        firstA[i + 1] = firstA[i] + RA[i].records;
        firstB[i + 1] = firstB[i] + RB[i].records;
    }
    if (!regular || firstA[P] != firstB[P]) return false;
    cutB.assign(P + 1, nb);
    for_each_index(P, n_threads, [&](int i) {
        int j = 0;
        while (j + 1 < P && firstB[j + 1] <= firstA[i]) j++;
        cutB[i] = skip_records(b, nb, RB[j].begin, firstA[i] - firstB[j]);
    });
    cutB[P] = nb;
    return true;
}

inline bool merge_paired_parallel(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int n_threads,
                                  uint8_t **out, size_t *out_len, std::string &err)
{
    using namespace pairs_detail;
    *out = nullptr; *out_len = 0;
    auto sequential = [&]() -> bool {
        std::vector<uint8_t> buf;
        if (!merge_paired(a, na, b, nb, buf, err)) return false;
        uint8_t *p = static_cast<uint8_t *>(std::malloc(buf.size() ? buf.size() : 1));
        if (!p) { err = "out of memory"; return false; }
        std::memcpy(p, buf.data(), buf.size());
        *out = p; *out_len = buf.size();
        return true;
    };
    if (n_threads < 4 || na < (4u << 20) || nb < (4u << 20) || a[0] != '@' || b[0] != '@') return sequential();
    const int P = n_threads * 4;
    std::vector<Range> RA(P), RB(P);
    for (int i = 0; i < P; i++) {
        RA[i].begin = record_start_at_or_after(a, na, na / P * i, true);
        RB[i].begin = record_start_at_or_after(b, nb, nb / P * i, true);
    }
    for (int i = 0; i < P; i++) { RA[i].end = i + 1 < P ? RA[i + 1].begin : na; RB[i].end = i + 1 < P ? RB[i + 1].begin : nb; }
    for_each_index(2 * P, n_threads, [&](int i) {
        if (i < P) count_records(a, na, RA[i]); else count_records(b, nb, RB[i - P]);
    });
    std::vector<size_t> firstA(P + 1, 0), firstB(P + 1, 0);
    bool regular = true;
    for (int i = 0; i < P; i++) {
        regular = regular && RA[i].regular && RB[i].regular;
        firstA[i + 1] = firstA[i] + RA[i].records;
        firstB[i + 1] = firstB[i] + RB[i].records;
    }
    if (!regular || firstA[P] != firstB[P] || firstA[P] == 0) return sequential();
    // range i of file 1 = records [firstA[i], firstA[i + 1]); where those start in file 2
    std::vector<size_t> offB(P, 0);
    std::vector<std::vector<uint8_t>> part(P);
    std::vector<std::string> errs(P);
    std::vector<char> okv(P, 1);
    for_each_index(P, n_threads, [&](int i) {
        if (RA[i].records == 0) return;
        int j = 0;
        while (j + 1 < P && firstB[j + 1] <= firstA[i]) j++;
        offB[i] = skip_records(b, nb, RB[j].begin, firstA[i] - firstB[j]);
        // (file 2 from its matching record to its end: the join stops with the last line of file 1's range)
        okv[i] = merge_paired(a + RA[i].begin, RA[i].end - RA[i].begin, b + offB[i], nb - offB[i], part[i], errs[i],
                              (RA[i].end - RA[i].begin) + 1024) ? 1 : 0;
    });
    for (int i = 0; i < P; i++) if (!okv[i]) { err = errs[i]; return false; }
    std::vector<size_t> at(P + 1, 0);
    for (int i = 0; i < P; i++) at[i + 1] = at[i] + part[i].size();
    uint8_t *p = static_cast<uint8_t *>(std::malloc(at[P] ? at[P] : 1));
    if (!p) { err = "out of memory"; return false; }
    for_each_index(P, n_threads, [&](int i) {
        if (!part[i].empty()) std::memcpy(p + at[i], part[i].data(), part[i].size());
        std::vector<uint8_t>().swap(part[i]);
    });
    *out = p; *out_len = at[P];
    return true;
}

} // namespace host
