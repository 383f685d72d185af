// Test harness for jn_cuclark_amd/host/input.hpp (no GPU needed):
//   host_input load <file> [threads] -> the input image (gzip inflated; BGZF blocks on several threads) on stdout
//   host_input gzseg <file.gz> <segment bytes> [threads] -> the segments of whole records a gzip file is classified in (gzstream.hpp),
//                                    concatenated on stdout; on stderr "segments <n> largest <bytes> bad_starts <n> bgzf_blocks <n, -1: not BGZF or one thread>"
//   host_input pair <f1> <f2>     -> the joined mates on stdout
//   host_input pairp <f1> <f2> <threads> -> the same join on several threads (pairs.hpp)
//   host_input packm <f1> <f2> <k>   -> the mates packed straight from the two files (pack_mates), same output format as pack
//   host_input pack <file> <k> <threads>  -> index + 2-bit pack of the whole file as ONE batch, on stdout:
//                                    u64 n_reads, u64 n_containers, u32 reads_ptr[n+1], u16 containers[]
// exit code 2 + message on stderr on failure.
#include "../../jn_cuclark_amd/host/input.hpp"
#include "../../jn_cuclark_amd/host/reads.hpp"
#include "../../jn_cuclark_amd/host/pairs.hpp"
#include "../../jn_cuclark_amd/host/gzstream.hpp"

#include <cstdio>
#include <iostream>
#include <random>

int main(int argc, char **argv)
{
    std::string err;
    if ((argc == 3 || argc == 4) && std::string(argv[1]) == "load") {          // load <file> [threads]
        host::InputImage img;
        if (!img.load(argv[2], err, argc == 4 ? atoi(argv[3]) : 1)) { std::cerr << err << std::endl; return 2; }
        std::fwrite(img.data(), 1, img.size(), stdout);
        return 0;
    }
    if ((argc == 4 || argc == 5) && std::string(argv[1]) == "gzseg") {
        if (!host::GzSegments::is_gzip(argv[2])) { std::cerr << "not a gzip file" << std::endl; return 2; }
        host::GzSegments G;
        if (!G.open(argv[2], (size_t)std::strtoull(argv[3], nullptr, 10), err, argc == 5 ? atoi(argv[4]) : 1)) { std::cerr << err << std::endl; return 2; }
        host::GzSegments::Segment s;
        size_t n = 0, bad = 0, largest = 0;
        uint8_t first = 0;
        bool seen_last = false;
        while (G.next(s, err)) {
            if (seen_last) bad++;                              // nothing comes after the segment marked last
            if (n == 0 && s.size) first = s.data[0];
            if (s.size && (first == '>' || first == '@') && s.data[0] != first) bad++;       // every segment opens with a record
            if (s.size && (first == '>' || first == '@') && !s.last && s.data[s.size - 1] != '\n') bad++;
            std::fwrite(s.data, 1, s.size, stdout);
            largest = std::max(largest, s.size);
            seen_last = s.last;
            n++;
        }
        if (!err.empty()) { std::cerr << err << std::endl; return 2; }
        if (!seen_last) bad++;
        std::cerr << "segments " << n << " largest " << largest << " bad_starts " << bad << " bgzf_blocks " << (G.bgzf() ? (long)G.bgzf_blocks() : -1L) << std::endl;
        return 0;
    }
    if (argc == 5 && std::string(argv[1]) == "packm") {       // mates packed straight from their two files (reads.hpp pack_mates)
        host::InputImage a, b;
        if (!a.load(argv[2], err) || !b.load(argv[3], err)) { std::cerr << err << std::endl; return 2; }
        const unsigned k = (unsigned)atoi(argv[4]);
        host::ReadIndex R1, R2;
        if (!host::index_reads(a.data(), a.size(), R1, err) || !host::index_reads(b.data(), b.size(), R2, err)) { std::cerr << err << std::endl; return 2; }
        if (R1.size() != R2.size()) { std::cerr << "record counts differ" << std::endl; return 2; }
        const uint64_t n = R1.size();
        for (uint64_t i = 0; i < n; i++) R1.len[i] = R1.len[i] + 1 + R2.len[i];          // joined length: R1 'N' R2
        std::vector<uint32_t> ptr(n + 1);
        std::vector<uint16_t> con(host::container_bound(R1, 0, n, k));
        const uint64_t c = host::pack_mates(a.data(), R1, b.data(), R2, n, k, ptr.data(), con.data(), a.size(), b.size());
        std::fwrite(&n, 8, 1, stdout); std::fwrite(&c, 8, 1, stdout);
        std::fwrite(ptr.data(), 4, n + 1, stdout); std::fwrite(con.data(), 2, c, stdout);
        return 0;
    }
    if (argc == 6 && std::string(argv[1]) == "mates") {       // mates <f1> <f2> <ranges> <window>: where the ranges of file 1 pair up in file 2
        host::InputImage a, b;
        if (!a.load(argv[2], err) || !b.load(argv[3], err)) { std::cerr << err << std::endl; return 2; }
        const int P = atoi(argv[4]);
        setenv("MC_MATE_WINDOW", argv[5], 1);
        std::vector<size_t> cut(P + 1), by_id(P + 1, 0), by_count;
        for (int i = 0; i <= P; i++) cut[i] = i == P ? a.size() : host::record_start_at_or_after(a.data(), a.size(), a.size() / P * i, true);
        bool found = true;
        for (int i = 1; i < P && found; i++) found = cut[i] >= a.size() ? (by_id[i] = b.size(), true) : host::find_mate(a.data(), a.size(), cut[i], b.data(), b.size(), by_id[i]);
        by_id[P] = b.size();
        const bool counted = host::align_mates(a.data(), a.size(), b.data(), b.size(), cut, 3, by_count);
        std::cout << "by_id " << (found ? 1 : 0);
        if (found) for (size_t v : by_id) std::cout << " " << v;
        std::cout << "\nby_count " << (counted ? 1 : 0);
        if (counted) for (size_t v : by_count) std::cout << " " << v;
        std::cout << "\ncuts";
        for (size_t v : cut) std::cout << " " << v;
        std::cout << std::endl;
        return 0;
    }
    if (argc == 5 && std::string(argv[1]) == "pairp") {       // the join on argv[4] threads (pairs.hpp)
        host::InputImage a, b;
        if (!a.load(argv[2], err) || !b.load(argv[3], err)) { std::cerr << err << std::endl; return 2; }
        uint8_t *out = nullptr;
        size_t n = 0;
        if (!host::merge_paired_parallel(a.data(), a.size(), b.data(), b.size(), atoi(argv[4]), &out, &n, err)) { std::cerr << err << std::endl; return 2; }
        std::fwrite(out, 1, n, stdout);
        std::free(out);
        return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "pair") {
        host::InputImage a, b;
        if (!a.load(argv[2], err) || !b.load(argv[3], err)) { std::cerr << err << std::endl; return 2; }
        std::vector<uint8_t> out;
        if (!host::merge_paired(a.data(), a.size(), b.data(), b.size(), out, err)) { std::cerr << err << std::endl; return 2; }
        std::fwrite(out.data(), 1, out.size(), stdout);
        return 0;
    }
    if (argc == 5 && std::string(argv[1]) == "pack") {
        host::InputImage img;
        if (!img.load(argv[2], err)) { std::cerr << err << std::endl; return 2; }
        const unsigned k = (unsigned)atoi(argv[3]);
        host::ReadIndex R;
        if (!host::index_reads_parallel(img.data(), img.size(), atoi(argv[4]), R, err)) { std::cerr << err << std::endl; return 2; }
        const uint64_t n = R.size();
        std::vector<uint32_t> ptr(n + 1);
        std::vector<uint16_t> con(host::container_bound(R, 0, n, k));
        const uint64_t c = host::pack_reads(img.data(), R, 0, n, k, ptr.data(), con.data(), img.size());
        std::fwrite(&n, 8, 1, stdout); std::fwrite(&c, 8, 1, stdout);
        std::fwrite(ptr.data(), 4, n + 1, stdout); std::fwrite(con.data(), 2, c, stdout);
        std::cerr << (host::cpu_has_avx2() ? "avx2" : "scalar") << std::endl;
        return 0;
    }
#ifdef MC_HOST_X86
    if (argc == 3 && std::string(argv[1]) == "indexfuzz") {
        // the one-sweep AVX2 FASTQ indexer against the line-by-line indexer (libc memchr) on texts made to hurt:
        // soups of '@', newlines, blanks and bases; well-formed records cut off anywhere; records with a few bytes
        // overwritten by separators.  Prints the number of cases; exit code 3 on the first difference.
        if (!host::cpu_has_avx2()) { std::cout << "no avx2" << std::endl; return 0; }
        std::mt19937_64 rng(7);
        const long cases = atol(argv[2]);
        for (long it = 0; it < cases; it++) {
            std::vector<uint8_t> t;
            const int mode = (int)(it % 3);
            const size_t target = 64 + rng() % 700;
            t.push_back('@');
            if (mode == 0) {
                const char al[] = "@\n\n\n \tACGT+I>";
                while (t.size() < target) t.push_back((uint8_t)al[rng() % (sizeof al - 1)]);
            } else {
                while (t.size() < target + 200) {
                    if (t.size() > 1) t.push_back('@');
                    const size_t nl = rng() % 40;
                    for (size_t j = 0; j < nl; j++) t.push_back(rng() % 9 == 0 ? ' ' : (uint8_t)('a' + rng() % 26));
                    t.push_back('\n');
                    const size_t L = rng() % 120;
                    for (size_t j = 0; j < L; j++) t.push_back((uint8_t)"ACGTN"[rng() % 5]);
                    t.push_back('\n'); t.push_back('+'); t.push_back('\n');
                    for (size_t j = 0; j < L; j++) t.push_back(rng() % 30 == 0 ? '@' : 'I');
                    t.push_back('\n');
                }
                t.resize(target + rng() % 200);
                if (mode == 2) for (int j = 0; j < 3; j++) t[1 + rng() % (t.size() - 1)] = (uint8_t)"\n @\t"[rng() % 5];
            }
            t.push_back(0); t.pop_back();          // (the line-by-line indexer peeks one byte past the end)
            host::ReadIndex A, B;
            host::index_reads_with<host::NewlineScanLibc>(t.data(), t.size(), A, err);
            host::index_fastq_avx2(t.data(), t.size(), B);
            bool same = A.size() == B.size() && A.name_s.size() == B.name_s.size() && A.name_e.size() == B.name_e.size() &&
                        A.spos.size() == B.spos.size() && A.epos.size() == B.epos.size() && A.name_s.size() == A.size() &&
                        A.name_e.size() == A.size() && A.spos.size() == A.size() && A.epos.size() == A.size();
            for (size_t i = 0; same && i < A.size(); i++)
                same = A.name_s[i] == B.name_s[i] && A.name_e[i] == B.name_e[i] && A.spos[i] == B.spos[i] &&
                       A.epos[i] == B.epos[i] && A.len[i] == B.len[i];
            if (!same) {
                std::cerr << "indexers differ on case " << it << " (" << t.size() << " bytes)" << std::endl;
                std::fwrite(t.data(), 1, t.size(), stderr);
                return 3;
            }
        }
        std::cout << cases << std::endl;
        return 0;
    }
#endif
#ifdef MC_HOST_X86
    if (argc == 3 && std::string(argv[1]) == "indexfuzz_fasta") {
        // the same for the one-sweep FASTA indexer: multi-line records of random line widths, '>' inside sequences,
        // blank lines, empty names, records cut off anywhere, bytes overwritten by separators
        if (!host::cpu_has_avx2()) { std::cout << "no avx2" << std::endl; return 0; }
        std::mt19937_64 rng(9);
        const long cases = atol(argv[2]);
        for (long it = 0; it < cases; it++) {
            std::vector<uint8_t> t;
            const int mode = (int)(it % 3);
            const size_t target = 64 + rng() % 700;
            t.push_back('>');
            if (mode == 0) {
                const char al[] = ">\n\n\n \tACGT>N";
                while (t.size() < target) t.push_back((uint8_t)al[rng() % (sizeof al - 1)]);
            } else {
                while (t.size() < target + 200) {
                    if (t.size() > 1) t.push_back('>');
                    const size_t nl = rng() % 30;
                    for (size_t j = 0; j < nl; j++) t.push_back(rng() % 9 == 0 ? ' ' : (uint8_t)('a' + rng() % 26));
                    t.push_back('\n');
                    const size_t L = rng() % 200, w = 1 + rng() % 70;
                    for (size_t j = 0; j < L; j++) {
                        t.push_back((uint8_t)"ACGTN>"[rng() % (rng() % 40 == 0 ? 6 : 5)]);
                        if ((j + 1) % w == 0) t.push_back('\n');
                    }
                    if (rng() % 4) t.push_back('\n');
                    if (rng() % 10 == 0) t.push_back('\n');
                }
                t.resize(target + rng() % 200);
                if (mode == 2) for (int j = 0; j < 3; j++) t[1 + rng() % (t.size() - 1)] = (uint8_t)"\n >\t"[rng() % 5];
            }
            t.push_back(0); t.pop_back();
            host::ReadIndex A, B;
            host::index_reads_with<host::NewlineScanLibc>(t.data(), t.size(), A, err);
            host::index_fasta_avx2(t.data(), t.size(), B);
            bool same = A.size() == B.size() && A.name_s.size() == B.name_s.size() && A.name_e.size() == B.name_e.size() &&
                        A.spos.size() == B.spos.size() && A.epos.size() == B.epos.size() && A.name_s.size() == A.size() &&
                        A.name_e.size() == A.size() && A.spos.size() == A.size() && A.epos.size() == A.size();
            for (size_t i = 0; same && i < A.size(); i++)
                same = A.name_s[i] == B.name_s[i] && A.name_e[i] == B.name_e[i] && A.spos[i] == B.spos[i] &&
                       A.epos[i] == B.epos[i] && A.len[i] == B.len[i];
            if (!same) {
                std::cerr << "indexers differ on case " << it << " (" << t.size() << " bytes)" << std::endl;
                std::fwrite(t.data(), 1, t.size(), stderr);
                return 3;
            }
        }
        std::cout << cases << std::endl;
        return 0;
    }
#endif
    std::cerr << "usage: host_input load <file> | gzseg <file.gz> <bytes> | pair <f1> <f2> | pack <file> <k> <threads> | indexfuzz[_fasta] <cases>" << std::endl;
    return 1;
}
