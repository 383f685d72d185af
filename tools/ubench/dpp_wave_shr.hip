// What does DPP wave_shr:1 deliver on this part?  Prints, per lane, the value received from "lane - 1".
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dpp_wave_shr tools/ubench/dpp_wave_shr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xf, 0xf, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xf, 0xf, false);   // row_shr:1 for comparison
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("wave_shr:1 :"); for (int i = 0; i < 64; i++) printf(" %d", h[i]); printf("\n");
    printf("row_shr:1  :"); for (int i = 0; i < 64; i++) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}
