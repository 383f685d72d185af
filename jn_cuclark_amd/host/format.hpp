// printf("%g") of a ratio of two small integers, without printf.
//
// The CSV writer prints two doubles per read with "%g" (reference CuCLARK_hh.hh:2110-2117): gamma =
// hits / (length - k + 1) and confidence = best / (best + second).  Both are a/b with a <= b small, and
// snprintf is what the formatting threads spend their time in.  For 0 <= a <= b <= 2^20 the six
// significant digits "%g" shows can be had exactly by integer division: the double nearest to a/b lies
// within 2^-53 (relative) of a/b, while a/b is at least 1/(2 a 10^(5+p)) > 4e-13 (relative) away from any
// six-digit rounding boundary it does not sit on exactly -- and when it does sit on one (2r == b) this
// function declines and the caller uses snprintf.  Checked against snprintf for every a < b <= 3000 and
// random larger pairs (tests/test_host_cli.py::test_fast_g_format_matches_printf).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace host {

// Writes a/b as "%g" would and returns the number of bytes, or 0 when the case is not covered
// (b == 0, a > b, b > 2^20, or an exact rounding tie): use snprintf then.
inline int fmt_ratio_g(char *dst, uint64_t a, uint64_t b)
{
    if (b == 0 || a > b || b > (1u << 20)) return 0;
    if (a == 0) { dst[0] = '0'; return 1; }
    if (a == b) { dst[0] = '1'; return 1; }
    // decimal exponent X of a/b: the smallest p >= 1 with a * 10^p >= b gives X = -p
    // (64 bits are enough: a * 10^p < 10 b <= 10 * 2^20, times 10^5 stays below 2^44)
    int p = 1;
    uint64_t n = a * 10u;
    while (n < b) { n *= 10u; p++; }
    n *= 100000u;                                   // a * 10^(p+5): six significant digits before the point
    uint64_t q = n / b;
    const uint64_t r = n - q * b;
    if (2 * r == b) return 0;                       // exact tie: printf rounds the double, not the ratio
    if (2 * r > b) q++;
    int X = -p;
    if (q == 1000000u) { q = 100000u; X++; }        // rounded up into the next decade
    if (X == 0) { dst[0] = '1'; return 1; }         // 0.9999995.. -> 1
    char dig[6];
    for (int i = 5; i >= 0; i--) { dig[i] = (char)('0' + q % 10u); q /= 10u; }
    int nd = 6;
    while (nd > 1 && dig[nd - 1] == '0') nd--;      // %g strips trailing zeros
    char *o = dst;
    if (X >= -4) {                                  // fixed notation: 0.000ddd
        *o++ = '0'; *o++ = '.';
        for (int z = 0; z < -X - 1; z++) *o++ = '0';
        std::memcpy(o, dig, (size_t)nd); o += nd;
    } else {                                        // d.ddddde-XX
        *o++ = dig[0];
        if (nd > 1) { *o++ = '.'; std::memcpy(o, dig + 1, (size_t)(nd - 1)); o += nd - 1; }
        *o++ = 'e'; *o++ = '-';
        const int e = -X;
        if (e >= 100) { *o++ = (char)('0' + e / 100); }
        *o++ = (char)('0' + (e / 10) % 10);
        *o++ = (char)('0' + e % 10);
    }
    return (int)(o - dst);
}

// unsigned decimal, returns the number of bytes
inline int fmt_u32(char *dst, uint32_t v)
{
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    for (int i = 0; i < n; i++) dst[i] = tmp[n - 1 - i];
    return n;
}

} // namespace host
