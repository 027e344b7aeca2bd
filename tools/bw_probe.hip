// bw_probe.hip -- what can the MI355X memory system deliver for THIS path's I/O pattern?
// (6 B/px in as two interleaved-RGB streams, 8 B/px out as two RGB + two mask streams.)
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// one workgroup per row, like k_points_rows, no compute
template <int TPB>
__global__ void __launch_bounds__(TPB) k_pattern_rows(const uint8_t* d, const uint8_t* c, uint8_t* sbs, uint8_t* mask, int W, int H)
{
    const int fr = blockIdx.x / H, i = blockIdx.x - fr * H;
    const uint32_t* dr = (const uint32_t*)(d + ((size_t)fr * H + i) * 3 * W);
    const uint32_t* cr = (const uint32_t*)(c + ((size_t)fr * H + i) * 3 * W);
    uint32_t* l = (uint32_t*)(sbs + ((size_t)fr * H + i) * 6 * W);
    uint32_t* r = l + 3 * W / 4;
    uint32_t* ml = (uint32_t*)(mask + ((size_t)fr * H + i) * 2 * W);
    uint32_t* mr = ml + W / 4;
    for (int g = threadIdx.x; g < W / 4; g += TPB) {
        uint32_t a0 = dr[3 * g], a1 = dr[3 * g + 1], a2 = dr[3 * g + 2];
        uint32_t b0 = cr[3 * g], b1 = cr[3 * g + 1], b2 = cr[3 * g + 2];
        l[3 * g] = a0 ^ b0; l[3 * g + 1] = a1 ^ b1; l[3 * g + 2] = a2 ^ b2;
        r[3 * g] = a0 + b0; r[3 * g + 1] = a1 + b1; r[3 * g + 2] = a2 + b2;
        ml[g] = a0 & b1; mr[g] = a2 | b0;
    }
}

// the same with non-temporal loads and stores (NT bit 0 loads, bit 1 stores)
template <int TPB, int NT>
__global__ void __launch_bounds__(TPB) k_pattern_rows_nt(const uint8_t* d, const uint8_t* c, uint8_t* sbs, uint8_t* mask, int W, int H)
{
    const int fr = blockIdx.x / H, i = blockIdx.x - fr * H;
    const uint32_t* dr = (const uint32_t*)(d + ((size_t)fr * H + i) * 3 * W);
    const uint32_t* cr = (const uint32_t*)(c + ((size_t)fr * H + i) * 3 * W);
    uint32_t* l = (uint32_t*)(sbs + ((size_t)fr * H + i) * 6 * W);
    uint32_t* r = l + 3 * W / 4;
    uint32_t* ml = (uint32_t*)(mask + ((size_t)fr * H + i) * 2 * W);
    uint32_t* mr = ml + W / 4;
#define LD(p) ((NT & 1) ? __builtin_nontemporal_load(p) : *(p))
#define ST(v, p) do { if (NT & 2) __builtin_nontemporal_store((uint32_t)(v), p); else *(p) = (v); } while (0)
    for (int g = threadIdx.x; g < W / 4; g += TPB) {
        uint32_t a0 = LD(dr + 3 * g), a1 = LD(dr + 3 * g + 1), a2 = LD(dr + 3 * g + 2);
        uint32_t b0 = LD(cr + 3 * g), b1 = LD(cr + 3 * g + 1), b2 = LD(cr + 3 * g + 2);
        ST(a0 ^ b0, l + 3 * g); ST(a1 ^ b1, l + 3 * g + 1); ST(a2 ^ b2, l + 3 * g + 2);
        ST(a0 + b0, r + 3 * g); ST(a1 + b1, r + 3 * g + 1); ST(a2 + b2, r + 3 * g + 2);
        ST(a0 & b1, ml + g); ST(a2 | b0, mr + g);
    }
#undef LD
#undef ST
}

// R consecutive rows per workgroup (longer contiguous bursts per stream), non-temporal
template <int TPB, int R>
__global__ void __launch_bounds__(TPB) k_pattern_multirow_nt(const uint8_t* d, const uint8_t* c, uint8_t* sbs, uint8_t* mask, int W, int H)
{
    const size_t row0 = (size_t)blockIdx.x * R;
    for (int rr = 0; rr < R; ++rr) {
        const size_t row = row0 + rr;
        const uint32_t* dr = (const uint32_t*)(d + row * 3 * W);
        const uint32_t* cr = (const uint32_t*)(c + row * 3 * W);
        uint32_t* l = (uint32_t*)(sbs + row * 6 * W);
        uint32_t* r = l + 3 * W / 4;
        uint32_t* ml = (uint32_t*)(mask + row * 2 * W);
        uint32_t* mr = ml + W / 4;
        for (int g = threadIdx.x; g < W / 4; g += TPB) {
            uint32_t a0 = __builtin_nontemporal_load(dr + 3 * g), a1 = __builtin_nontemporal_load(dr + 3 * g + 1), a2 = __builtin_nontemporal_load(dr + 3 * g + 2);
            uint32_t b0 = __builtin_nontemporal_load(cr + 3 * g), b1 = __builtin_nontemporal_load(cr + 3 * g + 1), b2 = __builtin_nontemporal_load(cr + 3 * g + 2);
            __builtin_nontemporal_store(a0 ^ b0, l + 3 * g); __builtin_nontemporal_store(a1 ^ b1, l + 3 * g + 1); __builtin_nontemporal_store(a2 ^ b2, l + 3 * g + 2);
            __builtin_nontemporal_store(a0 + b0, r + 3 * g); __builtin_nontemporal_store(a1 + b1, r + 3 * g + 1); __builtin_nontemporal_store(a2 + b2, r + 3 * g + 2);
            __builtin_nontemporal_store(a0 & b1, ml + g); __builtin_nontemporal_store(a2 | b0, mr + g);
        }
    }
}

// flat grid-stride version of the same traffic
__global__ void k_pattern_flat(const uint32_t* d, const uint32_t* c, uint32_t* l, uint32_t* r, uint32_t* ml, uint32_t* mr, size_t ngroups)
{
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
        uint32_t a0 = d[3 * g], a1 = d[3 * g + 1], a2 = d[3 * g + 2];
        uint32_t b0 = c[3 * g], b1 = c[3 * g + 1], b2 = c[3 * g + 2];
        l[3 * g] = a0 ^ b0; l[3 * g + 1] = a1 ^ b1; l[3 * g + 2] = a2 ^ b2;
        r[3 * g] = a0 + b0; r[3 * g + 1] = a1 + b1; r[3 * g + 2] = a2 + b2;
        ml[g] = a0 & b1; mr[g] = a2 | b0;
    }
}

// same traffic with 16-byte accesses only (48 B per thread = 16 px)
__global__ void k_pattern_flat16(const uint4* d, const uint4* c, uint4* l, uint4* r, uint4* ml, uint4* mr, size_t ngroups16)
{
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ngroups16; g += (size_t)gridDim.x * blockDim.x) {
        // lanes cover 3 consecutive 16-B vectors each would be strided; instead read vector g of each third
        uint4 a = d[g], b = c[g];
        l[g] = make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
        r[g] = make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if ((g % 3) == 0) { ml[g / 3] = a; mr[g / 3] = b; }
    }
}

template <typename F>
float time_ms(F f, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main()
{
    const int W = 1920, H = 1080, N = 32;
    const size_t npx = (size_t)W * H * N;
    uint8_t *d, *c, *sbs, *mask;
    CK(hipMalloc(&d, npx * 3)); CK(hipMalloc(&c, npx * 3)); CK(hipMalloc(&sbs, npx * 6)); CK(hipMalloc(&mask, npx * 2));
    CK(hipMemset(d, 1, npx * 3)); CK(hipMemset(c, 2, npx * 3));
    const double bytes = (double)npx * 14;
    {
        size_t n16 = npx * 7 / 16;   // same total traffic as read 7 + write 7
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, 0, (const uint4*)sbs, (uint4*)d, n16 > npx * 3 / 16 ? npx * 3 / 16 : n16); }, 20);
        printf("copy16 (3 B/px read + 3 B/px write)      : %.1f us  %.2f TB/s\n", ms * 1e3, (double)npx * 6 / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_pattern_rows<256>, dim3(N * H), dim3(256), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=256                      : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pattern_rows<512>, dim3(N * H), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=512                      : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pattern_rows<128>, dim3(N * H), dim3(128), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=128                      : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_rows_nt<512, 1>), dim3(N * H), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=512, nt loads            : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_rows_nt<512, 2>), dim3(N * H), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=512, nt stores           : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_rows_nt<512, 3>), dim3(N * H), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=512, nt loads + stores   : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_rows_nt<256, 3>), dim3(N * H), dim3(256), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("pattern rows TPB=256, nt loads + stores   : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_multirow_nt<512, 2>), dim3(N * H / 2), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("2 rows per WG, TPB=512, nt                : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_multirow_nt<512, 4>), dim3(N * H / 4), dim3(512), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("4 rows per WG, TPB=512, nt                : %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_pattern_multirow_nt<1024, 2>), dim3(N * H / 2), dim3(1024), 0, 0, d, c, sbs, mask, W, H); }, 20);
        printf("2 rows per WG, TPB=1024 (2 rows in flight): %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_pattern_flat, dim3(blocks), dim3(256), 0, 0, (const uint32_t*)d, (const uint32_t*)c,
                                                    (uint32_t*)sbs, (uint32_t*)(sbs + npx * 3), (uint32_t*)mask, (uint32_t*)(mask + npx), npx / 4); }, 20);
        printf("pattern flat dwordx3, %5d blocks         : %.1f us  %.2f TB/s\n", blocks, ms * 1e3, bytes / ms / 1e9);
    }
    for (int blocks : {2048, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_pattern_flat16, dim3(blocks), dim3(256), 0, 0, (const uint4*)d, (const uint4*)c,
                                                    (uint4*)sbs, (uint4*)(sbs + npx * 3), (uint4*)mask, (uint4*)(mask + npx), npx * 3 / 16); }, 20);
        printf("pattern flat 16B, %5d blocks             : %.1f us  %.2f TB/s\n", blocks, ms * 1e3, bytes / ms / 1e9);
    }
    return 0;
}
