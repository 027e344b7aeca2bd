// Probe: semantics of v_cvt_pk_u8_f32 and a few other single instructions on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* in, unsigned* out, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    unsigned r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
    out[i] = r;
}
int main()
{
    float h[] = {0.0f, 0.49999997f, 0.5f, 0.50000006f, 1.5f, 2.5f, 3.5f, -0.5f, -0.49f, -1.0f, 254.5f, 254.49998f, 255.4f, 255.5f, 256.0f, 300.0f, 1e10f, NAN, INFINITY, -INFINITY, 0.99999994f, 1.0f, 127.5f, 128.5f};
    int n = sizeof h / sizeof h[0];
    float* d; unsigned* o; unsigned ho[64];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%14.8g -> %u   (rint-clamp %g)\n", h[i], ho[i], fminf(fmaxf(rintf(h[i]), 0.f), 255.f));
    return 0;
}
