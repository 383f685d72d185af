// Test harness for jn_cuclark_amd/host/input.hpp (no GPU needed):
//   host_input load <file>        -> the input image (gzip inflated) on stdout
//   host_input pair <f1> <f2>     -> the joined mates on stdout
//   host_input pack <file> <k> <threads>  -> index + 2-bit pack of the whole file as ONE batch, on stdout:
//                                    u64 n_reads, u64 n_containers, u32 reads_ptr[n+1], u16 containers[]
// exit code 2 + message on stderr on failure.
#include "../../jn_cuclark_amd/host/input.hpp"
#include "../../jn_cuclark_amd/host/reads.hpp"

#include <cstdio>
#include <iostream>

int main(int argc, char **argv)
{
    std::string err;
    if (argc == 3 && std::string(argv[1]) == "load") {
        host::InputImage img;
        if (!img.load(argv[2], err)) { std::cerr << err << std::endl; return 2; }
        std::fwrite(img.data(), 1, img.size(), stdout);
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
        const uint64_t c = host::pack_reads(img.data(), R, 0, n, k, ptr.data(), con.data());
        std::fwrite(&n, 8, 1, stdout); std::fwrite(&c, 8, 1, stdout);
        std::fwrite(ptr.data(), 4, n + 1, stdout); std::fwrite(con.data(), 2, c, stdout);
        std::cerr << (host::cpu_has_avx2() ? "avx2" : "scalar") << std::endl;
        return 0;
    }
    std::cerr << "usage: host_input load <file> | pair <f1> <f2> | pack <file> <k> <threads>" << std::endl;
    return 1;
}
