// dbbuild.hpp -- targets definition + database build (CPU; BASELINE config 1 plumbing).
//
// Semantics follow the reference builder (src/CuCLARK_hh.hh:690-1329 with
// src/HashTableStorage_hh.hh:421-461 addElement, :229-280 RemoveCommon and
// src/hashTable_hh.hh:473-546 write) but not its data structure: the reference fills a
// chained table of HTSIZE vectors (25.8 GB for the full table before the first k-mer);
// here every occurrence becomes a (bucket, quotient, target) triple, the triples are
// sorted, and a run scan keeps the k-mers seen in exactly one target.  The three
// files (.sz/.ky/.lb) come out byte-identical (tests/test_host_cli.py,
// tests/test_oracle_golden.py pin the same rule against the reference's own output).
#pragma once

#include "common.hpp"
#include "../../include/mc_api.h"
#include "../../include/mc_build.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace host {

struct Targets {
    std::vector<std::pair<std::string, std::string>> files;  // (path, label)   m_targetsID
    std::vector<std::string> labels;                         // unique labels   m_labels
    std::vector<std::string> names;                          // "NA" + labels   m_targetsName
};

// split on ' ' ',' '\n' '\t' '\r', at most `max` elements (src/file.cc:63-87)
inline std::vector<std::string> split_line(const std::string &line, size_t max)
{
    std::vector<std::string> out;
    size_t t = 0, n = line.size();
    auto sep = [](char c) { return c == ' ' || c == ',' || c == '\n' || c == '\t' || c == '\r'; };
    while (t < n && out.size() < max) {
        while (t < n && sep(line[t])) t++;
        std::string v;
        while (t < n && !sep(line[t])) v.push_back(line[t++]);
        if (!v.empty()) out.push_back(v);
    }
    return out;
}

// reference getTargetsData, src/CuCLARK_hh.hh:1789-1901: "<file> <label>" per line
inline bool read_targets(const char *path, Targets &T, std::string &err)
{
    FILE *f = std::fopen(path, "r");
    if (!f) { err = std::string("Failed to open targets data in file: ") + path; return false; }
    char *line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, f) != -1) {
        std::string l(line);
        if (!l.empty() && l.back() == '\n') l.pop_back();
        auto e = split_line(l, 3);
        if (e.empty()) continue;
        if (e.size() < 2) { err = " Missing label for " + e[0]; std::free(line); std::fclose(f); return false; }
        if (!file_readable(e[0].c_str())) {
            err = "Failed to open file: " + e[0] + " defined in " + path;
            std::free(line); std::fclose(f); return false;
        }
        T.files.emplace_back(e[0], e[1]);
        if (std::find(T.labels.begin(), T.labels.end(), e[1]) == T.labels.end()) T.labels.push_back(e[1]);
    }
    std::free(line);
    std::fclose(f);
    T.names.clear();
    T.names.push_back("NA");
    for (auto &s : T.labels) T.names.push_back(s);
    return true;
}

// reference getdbName, src/CuCLARK_hh.hh:580-592 (folder already ends in '/')
inline std::string db_name(const std::string &folder, unsigned k, size_t n_labels, unsigned min_count, unsigned gap)
{
    char buf[4096];
    if (LIGHT)
        std::snprintf(buf, sizeof buf, "%s/db_central_k%lu_t%lu_s%lu_m%lu_light_%lu.tsk", folder.c_str(),
                      (unsigned long)k, (unsigned long)n_labels, (unsigned long)HTSIZE, (unsigned long)min_count,
                      (unsigned long)gap);
    else
        std::snprintf(buf, sizeof buf, "%s/db_central_k%lu_t%lu_s%lu_m%lu.tsk", folder.c_str(), (unsigned long)k,
                      (unsigned long)n_labels, (unsigned long)HTSIZE, (unsigned long)min_count);
    return buf;
}

struct Occ { uint64_t r, q; uint16_t t; uint32_t w; uint64_t seq; };      // w: occurrences this entry stands for (spectrum lines carry a count); seq: position in the stream of all targets

// --tsk: the reference also writes, per target, a text file of its target-specific k-mers (createTargetFilesNames,
// src/CuCLARK_hh.hh:342-378, and EHashtable::SaveMultiple, src/HashTableStorage_hh.hh:282-327, called from
// makeSpecificTargetSets :1315 BEFORE the table is sorted): "<k-mer value>\t<count>\t<k-mer>" for every k-mer seen in
// one target only, in the order the reference's table iterates -- buckets ascending, inside a bucket in the order the
// k-mers were first inserted.  The count is the element's: Element (full variant) adds up modulo 2^32, lElement
// (light) is a byte that takes an addition only while the sum stays below 255 (src/dataType.hh:286-341).
struct TskEntry { uint64_t r, q, first; uint16_t t; uint64_t count; };

// src/kmersConversion.cc:88-130 IndexTovector: base 4 digits, first base first, 3 = A, 2 = C, 1 = G, 0 = T
inline void kmer_text(uint64_t v, unsigned k, char *out)
{
    static const char L[4] = {'T', 'G', 'C', 'A'};
    for (unsigned i = 0; i < k; i++) out[i] = L[(v >> (2 * (k - 1 - i))) & 3u];
    out[k] = 0;
}

inline bool write_tsk_files(const Targets &T, const std::string &folder, unsigned k, std::vector<TskEntry> &E, std::string &err)
{
    // inside a bucket: insertion order (the group scan delivers quotient order)
    std::sort(E.begin(), E.end(), [](const TskEntry &a, const TskEntry &b) { return a.r != b.r ? a.r < b.r : a.first < b.first; });
    std::vector<FILE *> fds(T.labels.size(), nullptr);
    for (size_t t = 0; t < T.labels.size(); t++) {
        char name[4096];
        // (the reference's light variant indexes an empty vector for this name, :367; the full variant's rule is used for both)
        std::snprintf(name, sizeof name, LIGHT ? "%s/%s_k%lu_light.ht" : "%s/%s_k%lu.ht", folder.c_str(), T.labels[t].c_str(), (unsigned long)k);
        fds[t] = std::fopen(name, "w+");
        if (!fds[t]) { err = std::string("Failed to create ") + name; for (FILE *f : fds) if (f) std::fclose(f); return false; }
        std::fprintf(fds[t], "#Target specific k-mers labeled %s and appearing strictly more than %lu times.\n", T.labels[t].c_str(), 0ul);
        std::fprintf(fds[t], "#IKMER ICOUNT %lu-MER \n#\n", (unsigned long)k);
    }
    char text[40];
    for (const TskEntry &e : E) {
        kmer_text(e.r + e.q * HTSIZE, k, text);
        std::fprintf(fds[e.t], "%llu\t%lu\t%s\n", (unsigned long long)(e.r + e.q * HTSIZE), (unsigned long)e.count, text);
    }
    for (FILE *f : fds) std::fclose(f);
    return true;
}

// Forward k-mers of one target file, handed to emit(uint64_t kmer, uint32_t count).  FASTA: every window of k
// valid bases (full variant, :1127-1180), or -- light variant -- consecutive
// NON-overlapping windows of which every gap-th is kept (:707-760).  Any non-ACGTU byte
// except '\n' resets the window; '>' skips its header line.  FASTQ: the sequence line of
// each 4-line record.  Anything else is a k-mer SPECTRUM: lines "<k-mer> <count>" (:861-876 light: every gap-th
// line whose count exceeds min_count, the line counter restarting at each line taken; :1085-1093 full: every
// line whose count exceeds min_count).
template <class Emit>
inline bool scan_target_file(const std::string &path, unsigned k, unsigned gap, unsigned min_count, Emit &&emit, uint64_t &nt,
                             std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "Failed to open " + path; return false; }
    std::vector<unsigned char> buf;
    {
        unsigned char tmp[1 << 16];
        size_t g;
        while ((g = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + g);
    }
    std::fclose(f);
    if (buf.empty()) return true;
    const bool fasta = buf[0] == '>', fastq = buf[0] == '@';
    const auto &ct = codes();
    if (!fasta && !fastq) {
        uint8_t counter = 0;                 // an 8-bit counter in the reference too
        size_t i = 0;
        const size_t n = buf.size();
        while (i < n) {
            size_t e = i;
            while (e < n && buf[e] != '\n') e++;
            const std::vector<std::string> el = split_line(std::string((const char *)&buf[i], e - i), 2);
            i = e + 1;
            if (el.size() < 2) continue;
            const unsigned long val = (unsigned long)std::atoi(el[1].c_str());
            const bool take = LIGHT ? (counter % gap == 0 && val > min_count) : val > min_count;
            if (take) {
                uint64_t km = 0;
                for (char ch : el[0]) {
                    const int code = ch == 'U' || ch == 'u' ? -1 : ct.r[(unsigned char)ch];
                    if (code < 0) { err = "Failed to compute k-mer value of " + el[0]; return false; }
                    km = (km << 2) | (uint64_t)code;
                }
                emit(km, (uint32_t)val);
                if (LIGHT) counter = 0;
            }
            if (LIGHT) counter++;
        }
        return true;
    }
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t km = 0, iter = 0;
    unsigned cpt = 0;
    size_t i = 0, n = buf.size();
    auto skip_line = [&]() { while (i < n && buf[i] != '\n') i++; i++; };
    if (fastq) skip_line();
    while (i < n) {
        const unsigned char c = buf[i];
        const int code = ct.r[c];
        if (code >= 0) {
            nt++;
            km = ((km << 2) | (uint64_t)code) & mask;
            cpt++;
            if (cpt >= k) {
                if (LIGHT) {
                    if (iter % gap == 0) emit(km, 1u);
                    iter++; km = 0; cpt = 0;
                } else {
                    emit(km, 1u);
                    cpt = k;
                }
            }
            i++;
            continue;
        }
        if (c == '\n') {
            if (fastq) { km = 0; cpt = 0; i++; skip_line(); skip_line(); skip_line(); }
            else i++;
            continue;
        }
        if (fasta && c == '>') { km = 0; cpt = 0; skip_line(); continue; }
        nt++; km = 0; cpt = 0; i++;          // N and friends
    }
    return true;
}

inline bool collect_file(const std::string &path, uint16_t target, unsigned k, unsigned gap, unsigned min_count,
                         std::vector<Occ> &out, uint64_t &nt, std::string &err)
{
    return scan_target_file(path, k, gap, min_count, [&](uint64_t x, uint32_t w) {
        const uint64_t c = canonical(x, k);
        out.push_back(Occ{c % HTSIZE, c / HTSIZE, target, w, (uint64_t)out.size()});
    }, nt, err);
}

// Build and write <base>.sz/.ky/.lb.  Returns the number of stored k-mers.
inline bool build_database(const Targets &T, unsigned k, unsigned gap, unsigned min_count, int key_bytes,
                           const std::string &base, uint64_t &stored, std::string &err,
                           const std::string *tsk_folder = nullptr)
{
    std::vector<Occ> occ;
    uint64_t nt = 0;
    for (size_t t = 0; t < T.files.size(); t++) {
        const auto it = std::find(T.labels.begin(), T.labels.end(), T.files[t].second);
        const uint16_t id = (uint16_t)(it - T.labels.begin());
        if (!collect_file(T.files[t].first, id, k, gap, min_count, occ, nt, err)) return false;
        std::fprintf(stderr, "\r Progress report: (%zu/%zu)    ", t + 1, T.files.size());
    }
    std::fprintf(stderr, "%lu nt read in total.\n", (unsigned long)nt);
    std::sort(occ.begin(), occ.end(), [](const Occ &a, const Occ &b) {
        if (a.r != b.r) return a.r < b.r;
        if (a.q != b.q) return a.q < b.q;
        return a.seq < b.seq;             // stream order inside a k-mer's group (what --tsk's counts and order need)
    });
    std::vector<TskEntry> tsk;
    FILE *fs = std::fopen((base + ".sz").c_str(), "wb");
    FILE *fk = std::fopen((base + ".ky").c_str(), "wb");
    FILE *fl = std::fopen((base + ".lb").c_str(), "wb");
    if (!fs || !fk || !fl) { err = "Failed to create " + base + ".*"; return false; }
    // sizes are streamed in chunks; a bucket over 255 cannot be stored (hashTable_hh.hh:498-506)
    const uint64_t CH = 1ull << 22;
    std::vector<uint8_t> szbuf(CH);
    size_t i = 0, n = occ.size();
    size_t distinct = 0;
    stored = 0;
    for (uint64_t b0 = 0; b0 < HTSIZE; b0 += CH) {
        const uint64_t b1 = std::min<uint64_t>(HTSIZE, b0 + CH);
        std::fill(szbuf.begin(), szbuf.end(), 0);
        while (i < n && occ[i].r < b1) {
            size_t j = i;
            bool multi = false;
            uint64_t count = 0;
            while (j < n && occ[j].r == occ[i].r && occ[j].q == occ[i].q) { multi |= occ[j].t != occ[i].t; count += occ[j].w; j++; }
            distinct++;
            if (tsk_folder && !multi) {
                uint64_t ec;
                if (LIGHT) {                                      // lElement: Set() truncates to a byte, AddToCount() saturates
                    uint8_t b = (uint8_t)occ[i].w;
                    for (size_t e = i + 1; e < j; e++) b = (uint8_t)(b + ((size_t)b + occ[e].w < 255 ? occ[e].w : 0));
                    ec = b;
                } else {
                    ec = count > 4294967296ull ? 4294967295ull : (count & 0xFFFFFFFFull);      // ICount: two 16-bit digits
                }
                tsk.push_back(TskEntry{occ[i].r, occ[i].q, occ[i].seq, occ[i].t, ec});
            }
            if (!multi && count > min_count) {        // multiplicity 1 and count > minCount
                uint8_t &s = szbuf[occ[i].r - b0];
                if (s == 255) { err = "This table can not be stored on disk: Some bucket list size exceeds 255."; return false; }
                s++;
                if (key_bytes == 2) { uint16_t v = (uint16_t)occ[i].q; std::fwrite(&v, 2, 1, fk); }
                else if (key_bytes == 4) { uint32_t v = (uint32_t)occ[i].q; std::fwrite(&v, 4, 1, fk); }
                else std::fwrite(&occ[i].q, 8, 1, fk);
                std::fwrite(&occ[i].t, 2, 1, fl);
                stored++;
            }
            i = j;
        }
        std::fwrite(szbuf.data(), 1, b1 - b0, fs);
    }
    std::fclose(fs); std::fclose(fk); std::fclose(fl);
    if (tsk_folder && !write_tsk_files(T, *tsk_folder, k, tsk, err)) return false;
    std::fprintf(stderr, "Mother Hashtable successfully built. %zu %u-mers stored.\n", distinct, k);
    std::fprintf(stderr, "%lu %u-mers successfully stored in database.\n", (unsigned long)stored, k);
    return true;
}

// ---- recovery: the database files vanished, the per-target .ht files of an earlier --tsk run are still there --------
// Reference: getTargetsData (src/CuCLARK_hh.hh:1826-1836) finds "<folder>/<label>_k<k>.ht" for every label and therefore
// does NOT rebuild from the target files; loadSpecificTargetSets (:633-684) then fails to read the database and -- only
// when --tsk was given, otherwise "Failed to find the database." -- reads the .ht files back (EHashtable::Load,
// src/HashTableStorage_hh.hh:513-552: three header lines, then "<k-mer value> <count> ..." per line, kept when
// count > minCount), sorts every bucket, writes <base>.sz/.ky/.lb and leaves with exit(-1).
inline bool ht_files_present(const Targets &T, const std::string &folder, unsigned k)
{
    for (const std::string &l : T.labels) {
        char name[4096];
        std::snprintf(name, sizeof name, "%s/%s_k%lu.ht", folder.c_str(), l.c_str(), (unsigned long)k);
        if (!file_readable(name)) return false;
    }
    return !T.labels.empty();
}

inline bool recover_database(const Targets &T, const std::string &folder, unsigned k, unsigned min_count, int key_bytes,
                             const std::string &base, std::string &err)
{
    std::fprintf(stderr, "The database will be recovered from saved targets-specific data.\n");
    struct E { uint64_t r, q; uint16_t t; };
    std::vector<E> all;
    for (size_t t = 0; t < T.labels.size(); t++) {
        char name[4096];
        // the names --tsk writes (createTargetFilesNames, :342-378)
        std::snprintf(name, sizeof name, LIGHT ? "%s/%s_k%lu_light.ht" : "%s/%s_k%lu.ht", folder.c_str(), T.labels[t].c_str(), (unsigned long)k);
        FILE *f = std::fopen(name, "r");
        if (!f) {
            std::fprintf(stderr, "Failed to open %s\n", name);
        } else {
            char *line = nullptr;
            size_t cap = 0;
            for (int h = 0; h < 3; h++) if (getline(&line, &cap, f) == -1) break;       // vector size / - / k-mer size: comment lines in practice
            while (getline(&line, &cap, f) != -1) {
                const auto e = split_line(line, 2);
                if (e.size() < 2) continue;
                const uint64_t v = (uint64_t)std::atoll(e[0].c_str());
                const uint32_t count = (uint32_t)std::atol(e[1].c_str());
                if (count > min_count) all.push_back(E{v % HTSIZE, v / HTSIZE, (uint16_t)t});
            }
            std::free(line);
            std::fclose(f);
        }
        std::fprintf(stderr, "\rDataset %zu loaded.   ", t + 1);
    }
    std::fprintf(stderr, "%zu %u-mers finally loaded. Creating database in disk...\n", all.size(), k);
    std::stable_sort(all.begin(), all.end(), [](const E &a, const E &b) { return a.r != b.r ? a.r < b.r : a.q < b.q; });
    size_t max_bucket = 0;
    for (size_t i = 0; i < all.size();) {
        size_t j = i;
        while (j < all.size() && all[j].r == all[i].r) j++;
        max_bucket = std::max(max_bucket, j - i);
        i = j;
    }
    std::fprintf(stderr, "Hashtable sorting done: maximum number of collisions: %zu\n", max_bucket);       // hTable::sortall, hashTable_hh.hh:215
    if (max_bucket >= 256) { err = "This table can not be stored on disk: Some bucket list size exceeds 255."; return false; }
    FILE *fs = std::fopen((base + ".sz").c_str(), "wb");
    FILE *fk = std::fopen((base + ".ky").c_str(), "wb");
    FILE *fl = std::fopen((base + ".lb").c_str(), "wb");
    if (!fs || !fk || !fl) { err = "Failed to create " + base + ".*"; return false; }
    const uint64_t CH = 1ull << 22;
    std::vector<uint8_t> szbuf(CH);
    size_t i = 0;
    for (uint64_t b0 = 0; b0 < HTSIZE; b0 += CH) {
        const uint64_t b1 = std::min<uint64_t>(HTSIZE, b0 + CH);
        std::fill(szbuf.begin(), szbuf.end(), 0);
        for (; i < all.size() && all[i].r < b1; i++) {
            szbuf[all[i].r - b0]++;
            if (key_bytes == 2) { uint16_t v = (uint16_t)all[i].q; std::fwrite(&v, 2, 1, fk); }
            else if (key_bytes == 4) { uint32_t v = (uint32_t)all[i].q; std::fwrite(&v, 4, 1, fk); }
            else std::fwrite(&all[i].q, 8, 1, fk);
            std::fwrite(&all[i].t, 2, 1, fl);
        }
        std::fwrite(szbuf.data(), 1, b1 - b0, fs);
    }
    std::fclose(fs); std::fclose(fk); std::fclose(fl);
    std::fprintf(stderr, "Central Hashtable successfully stored in disk.\n");
    return true;
}

// GPU build (include/mc_build.h): the host only extracts the k-mers, twice; counting,
// scatter, per-bucket sort, the one-target rule and compaction run on the device.
// Same files as build_database(), byte for byte (tests/test_host_cli.py).
inline bool build_database_gpu(const Targets &T, unsigned k, unsigned gap, unsigned min_count, int key_bytes,
                               const std::string &base, uint64_t &stored, std::string &err)
{
    mc_builder *B = nullptr;
    if (mc_builder_open(&B, 0, k, HTSIZE) != 0) { err = mc_last_error(); return false; }
    const size_t CHUNK = 1u << 23;
    uint64_t nt = 0;
    bool ok = true;
    for (int pass = 0; pass < 2 && ok; pass++) {
        if (pass == 1 && mc_builder_begin_fill(B) != 0) { err = mc_last_error(); ok = false; break; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) reduction(+ : nt)
#endif
        for (long t = 0; t < (long)T.files.size(); t++) {
            if (!ok) continue;
            const auto it = std::find(T.labels.begin(), T.labels.end(), T.files[t].second);
            const uint16_t id = (uint16_t)(it - T.labels.begin());
            std::vector<uint64_t> km; std::vector<uint16_t> tg;
            km.reserve(CHUNK); tg.reserve(CHUNK);
            std::string lerr;
            uint64_t lnt = 0;
            auto flush = [&]() {
                if (km.empty()) return;
#ifdef _OPENMP
#pragma omp critical(mc_builder_feed)
#endif
                {
                    const int rc = pass == 0 ? mc_builder_count(B, km.data(), km.size())
                                             : mc_builder_fill(B, km.data(), tg.data(), km.size());
                    if (rc != 0) { err = mc_last_error(); ok = false; }
                }
                km.clear(); tg.clear();
            };
            // (a spectrum line's count is its weight in the count > min_count rule: with min_count = 0, the only
            // value the scripts use, one occurrence says the same; larger thresholds on spectra need the CPU builder)
            if (!scan_target_file(T.files[t].first, k, gap, min_count, [&](uint64_t x, uint32_t) {
                    km.push_back(x); tg.push_back(id);
                    if (km.size() >= CHUNK) flush();
                }, lnt, lerr)) {
#ifdef _OPENMP
#pragma omp critical(mc_builder_feed)
#endif
                { err = lerr; ok = false; }
            }
            flush();
            if (pass == 0) nt += lnt;
        }
    }
    uint64_t distinct = 0;
    if (ok && mc_builder_finish(B, min_count, &distinct, &stored) != 0) { err = mc_last_error(); ok = false; }
    if (ok && mc_builder_write(B, base.c_str(), key_bytes) != 0) { err = mc_last_error(); ok = false; }
    mc_builder_close(B);
    if (ok) {
        std::fprintf(stderr, "%lu nt read in total.\n", (unsigned long)nt);
        std::fprintf(stderr, "Mother Hashtable successfully built. %lu %u-mers stored.\n", (unsigned long)distinct, k);
        std::fprintf(stderr, "%lu %u-mers successfully stored in database.\n", (unsigned long)stored, k);
    }
    return ok;
}

} // namespace host

