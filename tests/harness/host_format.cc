// Test harness for jn_cuclark_amd/host/format.hpp: fmt_ratio_g against snprintf("%g").
//   host_format <max_b> <n_random>   -> "checked N declined D", exit 1 on the first mismatch
#include "../../jn_cuclark_amd/host/format.hpp"

#include <cstdlib>
#include <iostream>

static bool check(uint64_t a, uint64_t b, uint64_t &declined)
{
    char got[64], want[64];
    const int m = host::fmt_ratio_g(got, a, b);
    if (!m) { declined++; return true; }
    got[m] = 0;
    std::snprintf(want, sizeof want, "%g", (double)a / (double)b);
    if (std::strcmp(got, want) != 0) {
        std::cerr << a << "/" << b << ": got " << got << " want " << want << std::endl;
        return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    const uint64_t max_b = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 3000;
    const uint64_t n_rand = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1000000;
    uint64_t n = 0, declined = 0;
    for (uint64_t b = 1; b <= max_b; b++)
        for (uint64_t a = 0; a <= b; a++, n++)
            if (!check(a, b, declined)) return 1;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i < n_rand; i++, n++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t b = 1 + (x >> 20) % (1u << 20);
        const uint64_t a = (x & 0xFFFFu) % (b + 1);
        if (!check(a, b, declined)) return 1;
    }
    // cases the function must decline
    char tmp[64];
    if (host::fmt_ratio_g(tmp, 1, 0) || host::fmt_ratio_g(tmp, 3, 2) || host::fmt_ratio_g(tmp, 1, (1u << 20) + 1)) return 2;
    std::cout << "checked " << n << " declined " << declined << std::endl;
    return 0;
}
