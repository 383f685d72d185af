// CPU model of the super-k-mer index (csrc/mc_skm.hpp): the host-callable core -- hash, entries of a stored k-mer,
// record formation, run descriptor, record match -- against a plain set lookup of canonical k-mers, on genome-shaped
// data (related genomes, low-complexity stretches, both strands, substitutions, every k the index serves).
// Test infrastructure; built and run by tests/test_skm_model.py.  Exit code 0 = every count agrees.
#include "../../jn_cuclark_amd/csrc/mc_skm.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <unordered_map>
#include <vector>

using namespace mc::sk;
using mc::mz::mmer_len;

static uint64_t canon(uint64_t x, uint32_t k) { const uint64_t r = sk_revcomp(x, k); return x < r ? x : r; }

int run_case(uint32_t k, uint32_t n_lines, uint64_t seed, bool lowc)
{
    const uint32_t m = mmer_len(k);
    std::mt19937_64 rng(seed);
    // genomes: 6 targets in 2 genera (members differ by 4 % substitutions), a shared block, a tandem repeat, poly-A
    const int T = 6, G = 3000;
    std::vector<std::vector<uint8_t>> g(T, std::vector<uint8_t>(G));
    for (int t = 0; t < T; t++) {
        if (t % 3 == 0) for (int i = 0; i < G; i++) g[t][i] = rng() & 3;
        else for (int i = 0; i < G; i++) g[t][i] = (rng() % 100 < 4) ? (uint8_t)(rng() & 3) : g[t - t % 3][i];
        if (lowc) {
            for (int i = 500; i < 560; i++) g[t][i] = 3;                                  // poly-A
            for (int i = 900; i < 1000; i++) g[t][i] = (uint8_t)("\0\1\2"[(i - 900) % 3] + (t & 1));   // period-3 repeat
            for (int i = 1500; i < 1540; i++) g[t][i] = (uint8_t)((i & 1) ? 3 : 0);      // (AT)n: palindromic m-mers for even m
        }
    }
    // stored k-mers: canonical, seen in exactly one target
    std::unordered_map<uint64_t, int> owner;
    const uint64_t kmask = k >= 32 ? ~0ull : (1ull << (2 * k)) - 1;
    for (int t = 0; t < T; t++) {
        uint64_t x = 0;
        for (int i = 0; i < G; i++) {
            x = ((x << 2) | g[t][i]) & kmask;
            if (i + 1 < (int)k) continue;
            const uint64_t c = canon(x, k);
            auto it = owner.find(c);
            if (it == owner.end()) owner[c] = t; else if (it->second != t) it->second = -1;
        }
    }
    // build: entries -> lines -> records (first fit)
    std::vector<std::vector<SkSlot>> ent(n_lines), rec(n_lines);
    size_t n_stored = 0, n_entries = 0;
    for (auto &kv : owner) {
        if (kv.second < 0) continue;
        n_stored++;
        SkSlot e[2 * SK_W]; uint64_t K;
        const int n = sk_entries(kv.first, k, m, (uint32_t)kv.second, e, &K);
        if (n < 1) { printf("no entry\n"); return 1; }
        for (int i = 0; i < n; i++) { ent[sk_line_of(K, n_lines)].push_back(e[i]); n_entries++; }
    }
    size_t n_rec = 0;
    for (uint32_t l = 0; l < n_lines; l++) {
        for (auto &e : ent[l]) {
            bool done = false;
            for (auto &r : rec[l]) if (sk_consistent(r, e)) { sk_merge(r, e); done = true; break; }
            if (!done) rec[l].push_back(e);
        }
        n_rec += rec[l].size();
    }
    // queries: windows of the genomes (either strand, 2 % substitutions) and random reads, one part each
    size_t bad = 0, kmers = 0, hits = 0, runs = 0;
    for (int q = 0; q < 3000; q++) {
        const int len = 60 + (int)(rng() % 200);
        std::vector<uint8_t> rd(len);
        if (q % 5 == 4) for (auto &b : rd) b = rng() & 3;
        else {
            const int t = (int)(rng() % T), p = (int)(rng() % (G - len));
            for (int i = 0; i < len; i++) rd[i] = (rng() % 100 < 2) ? (uint8_t)(rng() & 3) : g[t][p + i];
            if (rng() & 1) { std::vector<uint8_t> r2(len); for (int i = 0; i < len; i++) r2[i] = 3 - rd[len - 1 - i]; rd = r2; }
        }
        const int nk = len - (int)k + 1, nm = len - (int)m + 1;
        if (nk < 1) continue;
        // keys of every m-mer position (as the kernel's front half makes them), k-mers, ends
        std::vector<uint64_t> key(nm), xs(nk);
        const uint64_t mmask = (1ull << (2 * m)) - 1;
        for (int p = 0; p < nm; p++) {
            uint64_t w = 0;
            for (uint32_t i = 0; i < m; i++) w = (w << 2) | rd[p + i];
            const uint64_t rw = sk_revcomp(w, m);
            key[p] = SK_ONE | sk_key_bits(w < rw ? w : rw, m) | ((uint64_t)(p & 15) << 1) | (w <= rw ? 1u : 0u);
            (void)mmask;
        }
        for (int p = 0; p < nk; p++) { uint64_t x = 0; for (uint32_t i = 0; i < k; i++) x = (x << 2) | rd[p + i]; xs[p] = x; }
        auto ends = [&](int p) { return (uint32_t)(((xs[p] >> (2 * k - 16)) & 0xFFFF) << 16) | (uint32_t)(xs[p] & 0xFFFF); };
        std::map<uint32_t, uint32_t> want, got;
        for (int p = 0; p < nk; p++) {
            auto it = owner.find(canon(xs[p], k));
            if (it != owner.end() && it->second >= 0) { want[(uint32_t)it->second]++; hits++; }
            kmers++;
        }
        std::vector<uint64_t> K(nk);
        for (int p = 0; p < nk; p++) { uint64_t b = key[p]; for (int j = 1; j < SK_W; j++) b = key[p + j] < b ? key[p + j] : b; K[p] = b; }
        for (int a = 0; a < nk;) {
            int b = a;
            while (b + 1 < nk && (uint32_t)K[b + 1] == (uint32_t)K[a] && (b + 1) % 128 != 0) b++;     // a run ends at a step boundary
            if (K[b] != K[a]) { printf("low words equal, keys differ\n"); return 1; }
            const SkRun r = sk_run(K[a], (uint32_t)a, (uint32_t)b, ends(a), ends(b));
            runs++;
            const uint32_t line = sk_line_of(K[a], n_lines);
            if ((q & 1) == 0) {
                for (auto &s : rec[line]) { const uint32_t c = sk_match(s, r.kd0, r.kd1, r.o_lo, r.o_hi, r.lr); if (c) got[s.d3] += c; }
            } else {            // as a hashed chain answers: k-mer by k-mer against single entries, found through the entry hash
                for (uint32_t o = r.o_lo; o <= r.o_hi; o++) {
                    const uint32_t h = sk_entry_hash(r.kd0, o, sk_lr_of(r.lr, o));
                    for (auto &e : ent[line]) {
                        const uint32_t c = sk_match(e, r.kd0, r.kd1, o, o, r.lr);
                        if (c && sk_entry_hash(e.d0, sk_ctz(e.d1 >> 20), e.d2) != h) { printf("entry hash differs between build and query\n"); return 1; }
                        if (c) got[e.d3] += c;
                    }
                }
            }
            a = b + 1;
        }
        if (want != got) {
            if (bad < 5) {
                printf("k=%u read %d len %d: want", k, q, len);
                for (auto &w : want) printf(" %u:%u", w.first, w.second);
                printf(" got");
                for (auto &w : got) printf(" %u:%u", w.first, w.second);
                printf("\n");
            }
            bad++;
        }
    }
    printf("k=%u m=%u lines=%u lowc=%d: %zu stored k-mers, %zu entries, %zu records (%.2f k-mers per record), %zu k-mers looked up in %zu runs, %zu hits, %zu reads differ\n",
           k, m, n_lines, (int)lowc, n_stored, n_entries, n_rec, (double)n_stored / (double)n_rec, kmers, runs, hits, bad);
    return bad ? 1 : 0;
}

int main()
{
    int rc = 0;
    // the hash is a bijection of the 2m-bit m-mers: no two of a dense sample collide, for every m served
    for (uint32_t k = 25; k <= 31; k++) {
        const uint32_t m = mmer_len(k);
        std::unordered_map<uint64_t, uint64_t> seen;
        std::mt19937_64 rng(k);
        for (int i = 0; i < 400000; i++) {
            const uint64_t w = (i < 100000 ? (uint64_t)i : rng()) & ((1ull << (2 * m)) - 1);
            const uint64_t h = sk_key_bits(w, m);
            if (h >> 52 || (h & ((1ull << (52 - 2 * m)) - 1))) { printf("hash bits outside the field\n"); return 1; }
            auto it = seen.find(h);
            if (it != seen.end() && it->second != w) { printf("hash collision\n"); return 1; }
            seen[h] = w;
        }
    }
    for (uint32_t k = 25; k <= 31; k++)
        for (int lowc = 0; lowc < 2; lowc++)
            rc |= run_case(k, lowc ? 64u : 3000u, 1000 + k, lowc != 0);
    return rc;
}
