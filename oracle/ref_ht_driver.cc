// ref_ht_driver.cc -- harness around the REFERENCE's own CPU hash table.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it calls
// (EHashtable / hTable / vectorToIndex / getReverseComplement) is compiled straight from
// /root/reference/src by oracle/Makefile into oracle/_ref/ (git-ignored).  It exists
// to pin oracle/clark_oracle.c and to generate tests/golden/ vectors:
//
//   ref_ht build <k> <minCount> <in.tsv> <out_base>
//        in.tsv: "<forward k-mer as decimal u64>\t<label string>" per occurrence.
//        Runs addElement -> SortAllHashTable(2) -> RemoveCommon -> Write(out_base, 2),
//        the sequence of CuCLARK_hh.hh:1097-1105, and prints the label table.
//   ref_ht query <k> <base> <in.txt>
//        in.txt: one forward k-mer (decimal u64) per line.  Loads base.sz/.ky/.lb with
//        EHashtable::Read (mmap path) and answers with EHashtable::queryElement(uint64)
//        = hTable::find(uint64, ILBL&) (hashTable_hh.hh:358-396), the CPU function the
//        GPU lookup mirrors.  Prints "<kmer>\t<found 0/1>\t<label>".
//   ref_ht time <k> <base> <n> <hit_every>
//        Single-thread timing of hTable::find: n pseudo-random forward k-mers (splitmix64 of 1..n,
//        masked to 2k bits); every <hit_every>-th one is replaced by a k-mer read back from the table
//        (a sure hit).  Prints "lookups <n> found <f> ns_per_lookup <t>".  oracle/time_lookup.c does
//        the same with the restatement: the cpu_baseline "port" is not slower than the original.
//   ref_ht kmer <k> <in.txt>
//        in.txt: one k-base string per line.  Prints "<string>\t<vectorToIndex>\t<getReverseComplement>".
//
// Built twice: with -include parameters_light_hh (HTSIZE 57777779, "ref_ht_light") and
// plain (HTSIZE 1610612741, "ref_ht_full"; its hash table alone needs ~26 GB of RAM).
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cinttypes>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <iostream>
#include <stdint.h>
#include <time.h>

#include "HashTableStorage_hh.hh"

#ifndef REF_KEY_T
#define REF_KEY_T uint32_t
#endif
#ifndef REF_BUILD_ELEMENT
#define REF_BUILD_ELEMENT lElement
#endif

static int cmd_build(int k, size_t minCount, const char *in, const char *out)
{
    FILE *f = fopen(in, "r");
    if (!f) { perror(in); return 2; }
    std::vector<std::pair<uint64_t, std::string> > occ;
    std::vector<std::string> labels;
    char lab[256];
    uint64_t km;
    while (fscanf(f, "%" SCNu64 " %255s", &km, lab) == 2) {
        occ.push_back(std::make_pair(km, std::string(lab)));
        if (std::find(labels.begin(), labels.end(), lab) == labels.end()) labels.push_back(lab);
    }
    fclose(f);
    std::vector<std::string> labels_c;
    EHashtable<REF_KEY_T, REF_BUILD_ELEMENT> ht(k, labels, labels_c);
    for (size_t i = 0; i < occ.size(); i++) ht.addElement(occ[i].first, occ[i].second, 1);
    fprintf(stderr, "mother table: %zu k-mers\n", ht.Size());
    ht.SortAllHashTable(2);
    ht.RemoveCommon(labels_c, minCount);
    uint64_t n = ht.Write(out, 2);
    printf("stored\t%" PRIu64 "\n", n);
    for (size_t i = 0; i < labels.size(); i++) printf("label\t%zu\t%s\n", i, labels[i].c_str());
    return 0;
}

static int cmd_query(int k, const char *base, const char *in)
{
    EHashtable<REF_KEY_T, rElement> ht(k);
    size_t fileSize = 0;
    if (!ht.Read(base, fileSize, 1, 1, true)) { fprintf(stderr, "Read failed\n"); return 2; }
    FILE *f = fopen(in, "r");
    if (!f) { perror(in); return 2; }
    uint64_t km;
    while (fscanf(f, "%" SCNu64, &km) == 1) {
        ILBL lab = 0;
        bool found = ht.queryElement(km, lab);
        printf("%" PRIu64 "\t%d\t%u\n", km, found ? 1 : 0, found ? (unsigned)lab : 0u);
    }
    fclose(f);
    return 0;
}

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// the same k-mer stream as oracle/time_lookup.c: stored k-mers come from base.hits (u64 values, binary)
static int cmd_time(int k, const char *base, size_t n, size_t hit_every)
{
    EHashtable<REF_KEY_T, rElement> ht(k);
    size_t fileSize = 0;
    if (!ht.Read(base, fileSize, 1, 1, true)) { fprintf(stderr, "Read failed\n"); return 2; }
    std::vector<uint64_t> hits;
    {
        std::string hf = std::string(base) + ".hits";
        FILE *f = fopen(hf.c_str(), "rb");
        if (f) { uint64_t v; while (fread(&v, 8, 1, f) == 1) hits.push_back(v); fclose(f); }
    }
    const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    std::vector<uint64_t> q(n);
    for (size_t i = 0; i < n; i++) {
        q[i] = splitmix64(i + 1) & mask;
        if (hit_every && !hits.empty() && i % hit_every == 0) q[i] = hits[(i / hit_every) % hits.size()];
    }
    struct timespec t0, t1;
    size_t found = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < n; i++) { ILBL lab = 0; found += ht.queryElement(q[i], lab) ? 1 : 0; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double ns = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / (double)n;
    printf("lookups %zu found %zu ns_per_lookup %.1f\n", n, found, ns);
    return 0;
}

static int cmd_kmer(int k, const char *in)
{
    FILE *f = fopen(in, "r");
    if (!f) { perror(in); return 2; }
    char s[128];
    while (fscanf(f, "%127s", s) == 1) {
        uint64_t fw = 0;
        std::string str(s);
        str.resize((size_t)k);
        vectorToIndex(str, fw);                       // = getKmers (kmersConversion.cc:49)
        uint64_t rv = 0;
        getReverseComplement(fw, (size_t)k, rv);      // = getReverse (kmersConversion.cc:39)
        printf("%s\t%" PRIu64 "\t%" PRIu64 "\n", s, fw, rv);
    }
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 6 && !strcmp(argv[1], "build"))
        return cmd_build(atoi(argv[2]), (size_t)atol(argv[3]), argv[4], argv[5]);
    if (argc >= 5 && !strcmp(argv[1], "query"))
        return cmd_query(atoi(argv[2]), argv[3], argv[4]);
    if (argc >= 6 && !strcmp(argv[1], "time"))
        return cmd_time(atoi(argv[2]), argv[3], (size_t)atol(argv[4]), (size_t)atol(argv[5]));
    if (argc >= 4 && !strcmp(argv[1], "kmer"))
        return cmd_kmer(atoi(argv[2]), argv[3]);
    if (argc >= 2 && !strcmp(argv[1], "htsize")) { printf("%zu\n", (size_t)HTSIZE); return 0; }
    fprintf(stderr, "usage: %s build|query|time|kmer|htsize ...\n", argv[0]);
    return 1;
}
