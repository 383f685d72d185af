/*
 * time_lookup.c -- single-thread timing of orc_db_lookup on the k-mer stream of
 * "ref_ht time" (oracle/ref_ht_driver.cc).  TEST INFRASTRUCTURE ONLY.
 *
 *   time_lookup <k> <htsize> <base> <n> <hit_every>
 *
 * prints "lookups <n> found <f> ns_per_lookup <t>".
 */
#include "clark_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s <k> <htsize> <base> <n> <hit_every>\n", argv[0]); return 1; }
    const int k = atoi(argv[1]);
    const uint64_t htsize = strtoull(argv[2], NULL, 10);
    const size_t n = (size_t)atol(argv[4]), hit_every = (size_t)atol(argv[5]);
    orc_db *db = orc_db_load(argv[3], htsize, 4, 1);
    if (!db) { fprintf(stderr, "load failed\n"); return 2; }
    uint64_t *hits = NULL;
    size_t n_hits = 0;
    {
        char hf[4096];
        snprintf(hf, sizeof hf, "%s.hits", argv[3]);
        FILE *f = fopen(hf, "rb");
        if (f) {
            fseek(f, 0, SEEK_END);
            n_hits = (size_t)ftell(f) / 8;
            fseek(f, 0, SEEK_SET);
            hits = (uint64_t *)malloc(n_hits * 8 + 8);
            if (fread(hits, 8, n_hits, f) != n_hits) n_hits = 0;
            fclose(f);
        }
    }
    const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t *q = (uint64_t *)malloc(n * 8 + 8);
    for (size_t i = 0; i < n; i++) {
        q[i] = splitmix64(i + 1) & mask;
        if (hit_every && n_hits && i % hit_every == 0) q[i] = hits[(i / hit_every) % n_hits];
    }
    struct timespec t0, t1;
    size_t found = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < n; i++) { uint16_t lab = 0; found += orc_db_lookup(db, k, q[i], 0, htsize, &lab) ? 1 : 0; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double ns = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / (double)n;
    printf("lookups %zu found %zu ns_per_lookup %.1f\n", n, found, ns);
    free(q); free(hits);
    orc_db_free(db);
    return 0;
}
