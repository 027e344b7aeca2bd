// r06 probe: for a frame's parameter set the headline kernel's division dl / z has only 65536 possible operands z (the 16-bit depth
// codes).  How often do the shorter Markstein sequences (no v_div_scale / v_div_fixup, no operand guards) give the bits of the IEEE
// division over exactly that set?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void k(float mult, float scale, float dl, unsigned* bad)
{
    const unsigned code = blockIdx.x * blockDim.x + threadIdx.x;      // 0 .. 65535
    const float z = ((float)(code << 16) * mult) * scale;
    if (!(z > 0.0001f)) return;
    const float want = dl / z;
    const float y0 = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, y0, 1.0f);
    const float y = __builtin_fmaf(e, y0, y0);
    const float q0 = dl * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-z, q0, dl), y, q0);
    const float q2 = __builtin_fmaf(__builtin_fmaf(-z, q1, dl), y, q1);
    const float p0 = dl * y0;                                            // from the raw 1-ulp reciprocal
    const float p1 = __builtin_fmaf(__builtin_fmaf(-z, p0, dl), y0, p0);
    const float p2 = __builtin_fmaf(__builtin_fmaf(-z, p1, dl), y0, p1);
    if (__float_as_uint(q1) != __float_as_uint(want)) atomicAdd(&bad[0], 1u);
    if (__float_as_uint(q2) != __float_as_uint(want)) atomicAdd(&bad[1], 1u);
    if (__float_as_uint(p1) != __float_as_uint(want)) atomicAdd(&bad[2], 1u);
    if (__float_as_uint(p2) != __float_as_uint(want)) atomicAdd(&bad[3], 1u);
    if (__float_as_uint(1.0f / z) != __float_as_uint(y)) atomicAdd(&bad[4], 1u);
}
int main()
{
    unsigned* d; hipMalloc(&d, 32);
    const double fovs[] = {45.0, 30.0, 60.0, 75.0, 90.0, 50.0, 37.5};
    const double ipds[] = {0.065, 0.063, 0.07, 0.03};
    const int Ws[] = {1920, 3840, 640, 1280};
    printf("# W xfov ipd scale : mismatches vs IEEE over the 65535 codes: q1(6 ops) q2(8 ops) p1(4 ops, raw rcp) p2(6 ops, raw rcp) y!=1/z\n");
    for (int W : Ws) for (double fov : fovs) for (double ipd : ipds) for (double sc : {1.0, 0.7320508, 1.37}) {
        const double fx = W / (2.0 * tan(fov * M_PI / 360.0));
        const float dl = (float)(fx * ipd / 2.0), mult = (float)(100.0 / 4228250625.0), scale = (float)sc;
        hipMemset(d, 0, 32);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, mult, scale, dl, d);
        unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("%d %.1f %.3f %.3f : %u %u %u %u %u\n", W, fov, ipd, sc, h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
