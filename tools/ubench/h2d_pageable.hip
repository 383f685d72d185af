// h2d_pageable: how fast does text get from PAGEABLE host memory (an mmap'ed file) to the card?  N threads, each with its own
// stream, copy disjoint chunks with hipMemcpyAsync (the runtime stages pageable memory itself); then the same through pinned
// ring buffers filled by memcpy (what csrc/mc_ingest.hip did first), and the cost of hipHostRegister on the chunks.
//   hipcc -O2 --offload-arch=gfx950 -o h2d_pageable h2d_pageable.hip -lpthread && ./h2d_pageable [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t total = (size_t)((argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30)), chunk = 64ull << 20;
    char *src = (char *)malloc(total);
    memset(src, 'A', total);
    char *dst; hipMalloc(&dst, total);
    const size_t n_chunks = total / chunk;
    for (int nt : {1, 2, 4, 8, 16}) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (size_t c = t; c < n_chunks; c += nt) hipMemcpyAsync(dst + c * chunk, src + c * chunk, chunk, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s); hipStreamDestroy(s);
        });
        for (auto &x : th) x.join();
        printf("pageable hipMemcpyAsync, %2d threads: %.1f GB/s\n", nt, total / (now() - t0) / 1e9);
    }
    for (int nt : {4, 8, 16}) {
        std::vector<char *> pin(nt);
        const double ta = now();
        for (int t = 0; t < nt; t++) hipHostMalloc((void **)&pin[t], chunk, hipHostMallocDefault);
        const double tb = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (size_t c = t; c < n_chunks; c += nt) { memcpy(pin[t], src + c * chunk, chunk); hipMemcpyAsync(dst + c * chunk, pin[t], chunk, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
            hipStreamDestroy(s);
        });
        for (auto &x : th) x.join();
        printf("memcpy into one pinned 64 MB buffer per thread + H2D, %2d threads: %.1f GB/s (pinning the buffers: %.3f s)\n", nt, total / (now() - tb) / 1e9, tb - ta);
        for (int t = 0; t < nt; t++) hipHostFree(pin[t]);
    }
    for (int nt : {1, 4, 16}) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (size_t c = t; c < n_chunks; c += nt) {
                hipHostRegister(src + c * chunk, chunk, hipHostRegisterDefault);
                hipMemcpyAsync(dst + c * chunk, src + c * chunk, chunk, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
                hipHostUnregister(src + c * chunk);
            }
            hipStreamDestroy(s);
        });
        for (auto &x : th) x.join();
        printf("hipHostRegister chunk + H2D + unregister, %2d threads: %.1f GB/s\n", nt, total / (now() - t0) / 1e9);
    }
    return 0;
}
