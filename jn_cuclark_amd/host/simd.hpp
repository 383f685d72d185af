// simd.hpp -- AVX2 inner loops of the host ingest (chosen at run time; the scalar loops in reads.hpp
// remain for CPUs without AVX2 and for everything that is not a plain run of bases).
//
// Replaces the byte-at-a-time packer of the reference (src/CuCLARK_hh.hh:1637-1689: one table lookup,
// shift and branch per base) for the common case -- a run of ACGTU of either case: 32 bases per step
// become four u16 containers (first base in the high bits, A=3 C=2 G=1 T/U=0, :294-297).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#define MC_HOST_X86 1
#endif

namespace host {

inline bool cpu_has_avx2()
{
#ifdef MC_HOST_X86
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
#else
    return false;
#endif
}

#ifdef MC_HOST_X86
// Packs whole 32-base blocks of p[0, n) into out (4 containers per block) as long as every byte of the
// block is one of ACGTUacgtu; returns the number of bytes consumed (a multiple of 32).
__attribute__((target("avx2")))
inline size_t pack_blocks_avx2(const uint8_t *p, size_t n, uint16_t *out)
{
    // by low nibble: 'A'/'a' = 1, 'C'/'c' = 3, 'T'/'t' = 4, 'U'/'u' = 5, 'G'/'g' = 7
    const __m256i code_tbl = _mm256_setr_epi8(0, 3, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0,
                                              0, 3, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0);
    const char X = (char)0xFF;      // no byte & 0xDF equals 0xFF: nibbles without a base never match
    const __m256i char_tbl = _mm256_setr_epi8(X, 'A', X, 'C', 'T', 'U', X, 'G', X, X, X, X, X, X, X, X,
                                              X, 'A', X, 'C', 'T', 'U', X, 'G', X, X, X, X, X, X, X, X);
    const __m256i nib = _mm256_set1_epi8(0x0F), upper = _mm256_set1_epi8((char)0xDF);
    const __m256i w1 = _mm256_set1_epi16(0x0104);        // bytes (4, 1): b0*4 + b1
    const __m256i w2 = _mm256_set1_epi32(0x00010010);    // words (16, 1): (..)*16 + (..)
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i));
        const __m256i lo = _mm256_and_si256(v, nib);
        const __m256i ok = _mm256_cmpeq_epi8(_mm256_and_si256(v, upper), _mm256_shuffle_epi8(char_tbl, lo));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) break;
        const __m256i codes = _mm256_shuffle_epi8(code_tbl, lo);
        const __m256i t1 = _mm256_maddubs_epi16(codes, w1);              // 16 x (2 bases, 4 bits)
        const __m256i t2 = _mm256_madd_epi16(t1, w2);                    // 8 x (4 bases, 8 bits)
        const __m256i t3 = _mm256_or_si256(_mm256_slli_epi64(t2, 8), _mm256_srli_epi64(t2, 32));   // low 16 bits of each u64: 8 bases
        uint16_t *o = out + i / 8;
        o[0] = (uint16_t)_mm256_extract_epi16(t3, 0);
        o[1] = (uint16_t)_mm256_extract_epi16(t3, 4);
        o[2] = (uint16_t)_mm256_extract_epi16(t3, 8);
        o[3] = (uint16_t)_mm256_extract_epi16(t3, 12);
    }
    return i;
}
#endif

// The last n < 32 bases of a run that ENDS after them (a FASTQ sequence line: 150 bp = four whole blocks and 22 bases,
// and the byte-at-a-time loop spent more time on those 22 than the blocks on 128): the same block arithmetic on a
// 32-byte load whose bytes behind the n-th are masked to code 0, i.e. to the zero padding of a left-aligned last
// container.  The caller guarantees 32 readable bytes.  Returns false -- nothing written -- unless all n bytes are bases.
__attribute__((target("avx2")))
inline bool pack_tail_avx2(const uint8_t *p, size_t n, uint16_t *out)
{
    const __m256i code_tbl = _mm256_setr_epi8(0, 3, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0,
                                              0, 3, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0);
    const char X = (char)0xFF;
    const __m256i char_tbl = _mm256_setr_epi8(X, 'A', X, 'C', 'T', 'U', X, 'G', X, X, X, X, X, X, X, X,
                                              X, 'A', X, 'C', 'T', 'U', X, 'G', X, X, X, X, X, X, X, X);
    const __m256i nib = _mm256_set1_epi8(0x0F), upper = _mm256_set1_epi8((char)0xDF);
    const __m256i w1 = _mm256_set1_epi16(0x0104), w2 = _mm256_set1_epi32(0x00010010);
    const __m256i idx = _mm256_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                         16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
    const __m256i lo = _mm256_and_si256(v, nib);
    const __m256i ok = _mm256_cmpeq_epi8(_mm256_and_si256(v, upper), _mm256_shuffle_epi8(char_tbl, lo));
    const uint32_t want = (uint32_t)((1ull << n) - 1ull);
    if (((uint32_t)_mm256_movemask_epi8(ok) & want) != want) return false;
    const __m256i keep = _mm256_cmpgt_epi8(_mm256_set1_epi8((char)n), idx);          // byte i < n
    const __m256i codes = _mm256_and_si256(_mm256_shuffle_epi8(code_tbl, lo), keep);
    const __m256i t1 = _mm256_maddubs_epi16(codes, w1);
    const __m256i t2 = _mm256_madd_epi16(t1, w2);
    const __m256i t3 = _mm256_or_si256(_mm256_slli_epi64(t2, 8), _mm256_srli_epi64(t2, 32));
    const uint16_t c[4] = {(uint16_t)_mm256_extract_epi16(t3, 0), (uint16_t)_mm256_extract_epi16(t3, 4),
                           (uint16_t)_mm256_extract_epi16(t3, 8), (uint16_t)_mm256_extract_epi16(t3, 12)};
    for (size_t j = 0; j < (n + 7) / 8; j++) out[j] = c[j];
    return true;
}

// Newline finder for the indexer: keeps the newline bitmap of the current 32-byte block, so that the
// 4 lines of a FASTQ record cost ~10 block loads instead of four memchr calls.
struct NewlineScanAvx2 {
    const uint8_t *t;
    size_t nb;
    size_t blk = (size_t)-1;      // start of the cached block
    uint32_t mask = 0;
    NewlineScanAvx2(const uint8_t *t_, size_t nb_) : t(t_), nb(nb_) {}
#ifdef MC_HOST_X86
    __attribute__((target("avx2")))
    size_t next(size_t from)      // position of the first newline at or after `from`, or nb
    {
        const __m256i nl = _mm256_set1_epi8(10);
        size_t b = from & ~(size_t)31;
        uint32_t m;
        if (b == blk) m = mask & (~0u << (from & 31));
        else m = 0;
        for (;;) {
            if (b != blk) {
                if (b + 32 > nb) {                      // tail: byte by byte
                    for (size_t i = from > b ? from : b; i < nb; i++) if (t[i] == 10) return i;
                    return nb;
                }
                blk = b;
                mask = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(t + b)), nl));
                m = b < from ? mask & (~0u << (from & 31)) : mask;
            }
            if (m) return b + (size_t)__builtin_ctz(m);
            b += 32;
        }
    }
#endif
};

struct NewlineScanLibc {
    const uint8_t *t;
    size_t nb;
    NewlineScanLibc(const uint8_t *t_, size_t nb_) : t(t_), nb(nb_) {}
    size_t next(size_t from)
    {
        const void *p = from < nb ? std::memchr(t + from, 10, nb - from) : nullptr;
        return p ? (size_t)((const uint8_t *)p - t) : nb;
    }
};

inline bool pack_tail(const uint8_t *p, size_t n, uint16_t *out)
{
#ifdef MC_HOST_X86
    if (cpu_has_avx2()) return pack_tail_avx2(p, n, out);
#endif
    (void)p; (void)n; (void)out;
    return false;
}

inline size_t pack_blocks(const uint8_t *p, size_t n, uint16_t *out)
{
#ifdef MC_HOST_X86
    if (cpu_has_avx2()) return pack_blocks_avx2(p, n, out);
#endif
    (void)p; (void)n; (void)out;
    return 0;
}

} // namespace host
