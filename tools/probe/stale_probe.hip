// stale_probe.hip -- can a kernel read a STALE copy of a small device buffer that was just rewritten in stream order?
//
// The r03 parity soak found a frame rendered with the previous context's per-frame parameter block (1 in ~20 000 context
// create / render / destroy cycles when 14 processes share the GPU): the block's device address is recycled by
// hipFree / hipMalloc, the new content arrives by a stream-ordered copy, and the render kernels read it with scalar loads.
// This probe reproduces the pattern without the library to find out which ingredient matters:
//   mode 0: hipMalloc + hipMemcpyAsync(pinned -> device) + kernel + hipFree per iteration      (what the library did)
//   mode 1: ONE allocation for the whole run, rewritten by hipMemcpyAsync every iteration
//   mode 2: as 0, but the kernel reads the block with vector loads (no scalar cache)
//   mode 3: as 0, with a hipStreamSynchronize between the copy and the kernel
//   mode 4: as 0, parameters written by a kernel from mapped pinned memory instead of hipMemcpyAsync
//   mode 5: as 0, with hipDeviceSynchronize before hipFree (what mdvt_destroy does)
//   mode 6: as 5, and the PINNED HOST staging buffer is hipHostMalloc'ed / hipHostFree'd per iteration too (what a context did)
//   mode 7: as 6, with a pageable (malloc) staging buffer instead of a pinned one
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probe/stale_probe.hip -o /tmp/stale_probe && /tmp/stale_probe <mode> <seconds>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Block { uint32_t v[96]; };      // 384 bytes, like FrameDev

__global__ void __launch_bounds__(256) k_check_scalar(const Block* __restrict__ p, uint32_t expect, uint32_t* bad, uint32_t* seen)
{
    const uint32_t a = p->v[0], b = p->v[95];          // uniform address: scalar loads
    if (threadIdx.x == 0 && (a != expect || b != expect)) { atomicAdd(bad, 1u); *seen = a; }
}
__global__ void __launch_bounds__(256) k_check_vector(const Block* p, uint32_t expect, uint32_t* bad, uint32_t* seen)
{
    const volatile uint32_t* q = p->v;
    const uint32_t a = q[threadIdx.x % 96];
    if (a != expect) { atomicAdd(bad, 1u); *seen = a; }
}
__global__ void k_write(Block* dst, const Block* src)
{
    if (threadIdx.x < 96) dst->v[threadIdx.x] = src->v[threadIdx.x];
}
__global__ void k_other(uint32_t* scratch, int n)      // unrelated work between iterations (the render kernels' role)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scratch[i] = scratch[i] * 1664525u + 1013904223u;
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 20.0;
    Block* h = nullptr;
    CK(hipHostMalloc((void**)&h, sizeof(Block), hipHostMallocDefault));
    uint32_t *bad = nullptr, *seen = nullptr, *scratch = nullptr;
    CK(hipMalloc((void**)&bad, 8)); seen = bad + 1;
    CK(hipMemset(bad, 0, 8));
    CK(hipMalloc((void**)&scratch, 1 << 20));
    Block* persistent = nullptr;
    if (mode == 1) CK(hipMalloc((void**)&persistent, sizeof(Block)));
    hipStream_t s = nullptr;           // the legacy default stream, as PyTorch's current stream is
    uint64_t it = 0;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t total_bad = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int rep = 0; rep < 64; ++rep, ++it) {
            const uint32_t val = (uint32_t)(it * 2654435761u) | 1u;
            Block* d = persistent;
            void* extra = nullptr;
            if (mode != 1) {
                CK(hipMalloc((void**)&d, sizeof(Block)));
                CK(hipMalloc(&extra, 4096 + (it % 7) * 512));          // the library allocates more than one thing per context
            }
            Block* hh = h;
            if (mode == 6) CK(hipHostMalloc((void**)&hh, sizeof(Block), hipHostMallocDefault));
            if (mode == 7) hh = (Block*)malloc(sizeof(Block));
            for (int k = 0; k < 96; ++k) hh->v[k] = val;
            if (mode == 4) {
                void* hd = nullptr;
                CK(hipHostGetDevicePointer(&hd, hh, 0));
                hipLaunchKernelGGL(k_write, dim3(1), dim3(128), 0, s, d, (const Block*)hd);
            } else {
                CK(hipMemcpyAsync(d, hh, sizeof(Block), hipMemcpyHostToDevice, s));
            }
            if (mode == 3) CK(hipStreamSynchronize(s));
            if (mode == 2) hipLaunchKernelGGL(k_check_vector, dim3(1080), dim3(256), 0, s, d, val, bad, seen);
            else hipLaunchKernelGGL(k_check_scalar, dim3(1080), dim3(256), 0, s, d, val, bad, seen);
            hipLaunchKernelGGL(k_other, dim3(64), dim3(256), 0, s, scratch, 16384);
            CK(hipStreamSynchronize(s));           // (the host buffer is rewritten next iteration)
            if (mode >= 5) CK(hipDeviceSynchronize());
            if (mode == 6) CK(hipHostFree(hh));
            if (mode == 7) free(hh);
            if (mode != 1) { CK(hipFree(extra)); CK(hipFree(d)); }
        }
        uint32_t hb[2];
        CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
        if (hb[0] != total_bad) {
            printf("pid %d mode %d: iteration <= %llu: %u workgroups saw a stale block (one saw 0x%08x)\n", (int)getpid(), mode, (unsigned long long)it, hb[0] - total_bad, hb[1]);
            total_bad = hb[0];
        }
    }
    printf("pid %d mode %d: %llu iterations, %u stale workgroup reads\n", (int)getpid(), mode, (unsigned long long)it, total_bad);
    return 0;
}
