// common.hpp -- parameters and k-mer arithmetic of the host driver.
//
// The host side mirrors the reference's host driver (src/main.cc, src/CuCLARK_hh.hh)
// for the classification path only; the GPU work goes through the C ABI
// (include/mc_api.h).  One source tree builds both variants, selected with -DMC_LIGHT
// exactly as the reference swaps src/parameters.hh for src/parameters_light_hh
// (src/Makefile:27-34).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace host {

// reference src/parameters.hh:35-53 and src/parameters_light_hh:35-54
#ifdef MC_LIGHT
static constexpr uint64_t HTSIZE = 57777779ull;
static constexpr uint32_t MAXHITS = 23;
static constexpr bool     LIGHT = true;
#else
static constexpr uint64_t HTSIZE = 1610612741ull;
static constexpr uint32_t MAXHITS = 15;
static constexpr bool     LIGHT = false;
#endif
static constexpr uint64_t LHTSIZE = 57777779ull;
static constexpr uint32_t NBN = 1;                 // 'N' joining paired mates
static constexpr uint32_t SFACTORMAX = 30;
static constexpr uint32_t OBJECTNAMEMAX = 40;
static constexpr uint32_t MAXK = 32;               // reference src/main.cc:40
#define MC_HOST_VERSION "1.1"

// 2-bit code of the READ packer: A=3 C=2 G=1 T/U=0 (src/CuCLARK_hh.hh:294-297)
struct CodeTable {
    int8_t r[256];      // -1 = not a base
    CodeTable()
    {
        for (int i = 0; i < 256; i++) r[i] = -1;
        r['A'] = r['a'] = 3; r['C'] = r['c'] = 2; r['G'] = r['g'] = 1;
        r['T'] = r['t'] = 0; r['U'] = r['u'] = 0;
    }
};
inline const CodeTable &codes()
{
    static const CodeTable t;
    return t;
}

// reference src/kmersConversion.cc:39-47 / src/CuClarkDB.cu:1196-1203
inline uint64_t revcomp(uint64_t x, unsigned k)
{
    uint64_t r = x;
    r = ((r >> 2)  & 0x3333333333333333ULL) | ((r & 0x3333333333333333ULL) << 2);
    r = ((r >> 4)  & 0x0F0F0F0F0F0F0F0FULL) | ((r & 0x0F0F0F0F0F0F0F0FULL) << 4);
    r = ((r >> 8)  & 0x00FF00FF00FF00FFULL) | ((r & 0x00FF00FF00FF00FFULL) << 8);
    r = ((r >> 16) & 0x0000FFFF0000FFFFULL) | ((r & 0x0000FFFF0000FFFFULL) << 16);
    r = (r >> 32) | (r << 32);
    return (~r) >> (64 - 2 * k);
}
inline uint64_t canonical(uint64_t x, unsigned k)
{
    const uint64_t r = revcomp(x, k);
    return x < r ? x : r;
}

// key width chosen like src/main.cc:245-286
inline int key_bytes_for(unsigned k)
{
    const size_t t_b = (size_t)(std::log((double)HTSIZE) / std::log(4.0));
    if (k <= t_b + 8) return 2;
    if (k <= t_b + 16) return 4;
    return 8;
}

inline bool file_readable(const char *p)
{
    FILE *f = std::fopen(p, "r");
    if (!f) return false;
    std::fclose(f);
    return true;
}

} // namespace host
