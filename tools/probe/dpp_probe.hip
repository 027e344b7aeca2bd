// Probe: DPP wave_shl:1 / wave_shr:1 on gfx950 (which lane reads which, what the edge lane gets).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out)
{
    const int l = threadIdx.x;
    const int v = 100 + l;
    out[l] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);        // wave_shl:1
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);   // wave_shr:1
}
int main()
{
    int* d; int h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("wave_shl:1 lane0..3 = %d %d %d %d  lane15,16 = %d %d lane31,32 = %d %d lane 62,63 = %d %d\n", h[0], h[1], h[2], h[3], h[15], h[16], h[31], h[32], h[62], h[63]);
    printf("wave_shr:1 lane0..3 = %d %d %d %d  lane15,16 = %d %d lane31,32 = %d %d lane 62,63 = %d %d\n", h[64], h[65], h[66], h[67], h[79], h[80], h[95], h[96], h[126], h[127]);
    return 0;
}
