// fresh_alloc_probe.hip -- is the content of a FRESH device allocation safe across kernels on its first use?
//
// The r04 soak found one first render of a fresh context in ~3 000 (12 processes sharing the GPU, each creating and destroying a
// context per render) with a queued triangle lost -- only once the triangle queue's block had grown past 2 MB, i.e. once that block
// no longer came out of the runtime's cache of small fragments but was allocated from / returned to the driver by every context
// (profiles/r04_soak_summary.md).  This probe repeats the pattern without the library:
//
//   per iteration:  hipMalloc(bytes)  [+ a few small companions]  ->  small pinned H2D copy (the parameter block)
//                   -> k_fill: workgroup b writes slice b with f(iteration, index)                      (the producing kernel)
//                   -> k_append: every wave appends records through a per-slice counter the fill zeroed  (the rasteriser's queue)
//                   -> k_check: workgroup b verifies slice b + 1 and the appended records of slice b + 3  (another XCD reads them)
//                   -> hipStreamSynchronize, D2H of the error word -> hipDeviceSynchronize -> hipFree
//
// A workgroup b runs on XCD b % 8, so the checker of a slice sits on another XCD (another L2, another TLB) than its writer.
//   mode 0: as above                               mode 1: ONE allocation for the whole run (control)
//   mode 2: as 0 + hipDeviceSynchronize() right after the hipMalloc
//   mode 3: as 0 + hipMemsetAsync over the whole block + hipStreamSynchronize before first use
//   mode 4: as 0, but without the H2D copy (no SDMA work from this process between the kernels)
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probe/fresh_alloc_probe.hip -o /tmp/fresh_probe
//               /tmp/fresh_probe <mode> <bytes> <seconds>          (run 12 at once; HSA_ENABLE_SDMA=0 for the A/B)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kSliceDw = 4096;                 // dwords per slice: [0] = append counter, [1 .. kAppend] = appended records, rest = pattern
constexpr int kAppend = 1024;

__device__ __forceinline__ uint32_t pat(uint32_t it, uint32_t i) { return (it * 2654435761u) ^ (i * 40503u) ^ 0x5bd1e995u; }

struct Params { uint32_t it, nslices, pad[94]; };         // 384 bytes, like the library's per-frame block

__global__ void __launch_bounds__(256) k_fill(uint32_t* blk, const Params* p)
{
    const uint32_t it = p->it;
    uint32_t* s = blk + (size_t)blockIdx.x * kSliceDw;
    for (int k = threadIdx.x; k < kSliceDw; k += 256) s[k] = k == 0 ? 0u : (k <= kAppend ? 0xDEADBEEFu : pat(it, (uint32_t)blockIdx.x * kSliceDw + k));
}
__global__ void __launch_bounds__(256) k_append(uint32_t* blk, const Params* p)
{
    // workgroup b appends kAppend records to slice (b + 5) % n: a wave reserves 64 places with one returning atomic
    const uint32_t n = p->nslices, it = p->it;
    uint32_t* s = blk + (size_t)((blockIdx.x + 5u) % n) * kSliceDw;
    for (int k = threadIdx.x; k < kAppend; k += 256) {
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&s[0], 64u);
        base = __shfl(base, 0);
        s[1 + base + (threadIdx.x & 63)] = it ^ ((uint32_t)k << 8) ^ 0xA0000000u;
    }
}
__global__ void __launch_bounds__(256) k_check(const uint32_t* blk, const Params* p, uint32_t* err)
{
    const uint32_t n = p->nslices, it = p->it;
    const uint32_t b1 = (blockIdx.x + 1u) % n, b3 = (blockIdx.x + 3u) % n;
    const uint32_t* s1 = blk + (size_t)b1 * kSliceDw;
    for (int k = kAppend + 1 + threadIdx.x; k < kSliceDw; k += 256) {
        const uint32_t v = s1[k], w = pat(it, b1 * kSliceDw + k);
        if (v != w) { if (atomicAdd(&err[0], 1u) == 0u) { err[2] = v; err[3] = w; err[4] = b1 * kSliceDw + k; err[5] = 1u; } }
    }
    const uint32_t* s3 = blk + (size_t)b3 * kSliceDw;
    if (threadIdx.x == 0 && s3[0] != (uint32_t)kAppend) { if (atomicAdd(&err[1], 1u) == 0u) { err[2] = s3[0]; err[3] = kAppend; err[4] = b3 * kSliceDw; err[5] = 2u; } }
    // every record k (k = 0 .. kAppend - 1) must be present exactly once: sum and xor of the k fields
    __shared__ uint32_t acc[2];
    if (threadIdx.x == 0) { acc[0] = 0u; acc[1] = 0u; }
    __syncthreads();
    uint32_t sum = 0, x = 0, bad = 0;
    for (int k = threadIdx.x; k < kAppend; k += 256) {
        const uint32_t v = s3[1 + k] ^ it ^ 0xA0000000u;
        if (v & 0xFFFC00FFu) { bad = s3[1 + k] | 1u; continue; }
        sum += v >> 8; x ^= v >> 8;
    }
    atomicAdd(&acc[0], sum); atomicXor(&acc[1], x);
    if (bad) { if (atomicAdd(&err[1], 1u) == 0u) { err[2] = bad; err[3] = it; err[4] = b3 * kSliceDw; err[5] = 3u; } }
    __syncthreads();
    if (threadIdx.x == 0 && (acc[0] != (uint32_t)kAppend * (kAppend - 1) / 2 || acc[1] != 0u))
        if (atomicAdd(&err[1], 1u) == 0u) { err[2] = acc[0]; err[3] = acc[1]; err[4] = b3 * kSliceDw; err[5] = 4u; }
}

int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)2200000;
    const double secs = argc > 3 ? atof(argv[3]) : 20.0;
    const uint32_t nslices = (uint32_t)(bytes / (kSliceDw * 4));
    if (nslices < 8) { printf("block too small\n"); return 1; }
    Params* h = nullptr;
    CK(hipHostMalloc((void**)&h, sizeof(Params), hipHostMallocDefault));
    Params* dp = nullptr;
    CK(hipMalloc((void**)&dp, sizeof(Params)));
    uint32_t* err = nullptr;
    CK(hipMalloc((void**)&err, 32));
    CK(hipMemset(err, 0, 32));
    uint32_t* persistent = nullptr;
    if (mode == 1) CK(hipMalloc((void**)&persistent, bytes));
    hipStream_t s = nullptr;
    uint64_t it = 0, bad_iters = 0;
    uint32_t seen[2] = {0, 0};
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        ++it;
        uint32_t* blk = persistent;
        void* extra[3] = {nullptr, nullptr, nullptr};
        if (mode != 1) {
            CK(hipMalloc((void**)&blk, bytes));
            for (int k = 0; k < 3; ++k) CK(hipMalloc(&extra[k], 24800 + 4096 * k));     // a context owns small blocks too
            if (mode == 2) CK(hipDeviceSynchronize());
            if (mode == 3) { CK(hipMemsetAsync(blk, 0, bytes, s)); CK(hipStreamSynchronize(s)); }
        }
        h->it = (uint32_t)it; h->nslices = nslices;
        if (mode == 4) { CK(hipMemcpy(dp, h, sizeof(Params), hipMemcpyHostToDevice)); }
        else CK(hipMemcpyAsync(dp, h, sizeof(Params), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fill, dim3(nslices), dim3(256), 0, s, blk, dp);
        hipLaunchKernelGGL(k_append, dim3(nslices), dim3(256), 0, s, blk, dp);
        hipLaunchKernelGGL(k_check, dim3(nslices), dim3(256), 0, s, blk, dp, err);
        CK(hipStreamSynchronize(s));
        uint32_t he[8];
        CK(hipMemcpy(he, err, 32, hipMemcpyDeviceToHost));
        if (he[0] != seen[0] || he[1] != seen[1]) {
            ++bad_iters;
            if (bad_iters <= 5)
                printf("pid %d mode %d: iteration %llu: %u pattern words, %u append errors; first: kind %u at dword %u saw 0x%08x want 0x%08x "
                       "(slice %u: filled by XCD %u, appended to by XCD %u, checked by XCD %u)\n",
                       (int)getpid(), mode, (unsigned long long)it, he[0] - seen[0], he[1] - seen[1], he[5], he[4], he[2], he[3],
                       he[4] / kSliceDw, (he[4] / kSliceDw) % 8u, (he[4] / kSliceDw + nslices - 5u) % nslices % 8u,
                       (he[4] / kSliceDw + nslices - (he[5] == 1u ? 1u : 3u)) % nslices % 8u);
            seen[0] = he[0]; seen[1] = he[1];
            CK(hipMemset(err + 2, 0, 24));
            // reset the "first" latch: counts restart from what was seen
            uint32_t z[2] = {0, 0};
            CK(hipMemcpy(err, z, 8, hipMemcpyHostToDevice));
            seen[0] = seen[1] = 0;
        }
        CK(hipDeviceSynchronize());
        if (mode != 1) { for (int k = 0; k < 3; ++k) CK(hipFree(extra[k])); CK(hipFree(blk)); }
    }
    printf("pid %d mode %d bytes %zu: %llu iterations, %llu bad\n", (int)getpid(), mode, bytes, (unsigned long long)it, (unsigned long long)bad_iters);
    return 0;
}
