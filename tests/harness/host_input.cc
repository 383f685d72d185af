// Test harness for jn_cuclark_amd/host/input.hpp (no GPU needed):
//   host_input load <file>        -> the input image (gzip inflated) on stdout
//   host_input pair <f1> <f2>     -> the joined mates on stdout
// exit code 2 + message on stderr on failure.
#include "../../jn_cuclark_amd/host/input.hpp"

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
    std::cerr << "usage: host_input load <file> | pair <f1> <f2>" << std::endl;
    return 1;
}
