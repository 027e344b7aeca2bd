// grid_barrier_probe.hip -- what would a persistent level loop save the infill-mask completion?  (verdict r04 item 5b)
//
// The completion's passes B and C are 2 R - 1 DEPENDENT launches (R = 131 levels on converged views), many of them at the floor
// of a launch.  One persistent launch with a grid barrier between the levels would replace that floor by the barrier's cost.
// Between two levels the work of one XCD must become visible to the other seven (plain stores to T / the work image, read
// through the other XCDs' L2s at the next level): an agent-scope release + acquire per level and workgroup -- the L2 write-back
// and invalidate a kernel boundary performs too.  This probe measures both floors with the same tiny "level":
//   every workgroup writes 64 words of its slice, the next level's workgroup b reads the slice of workgroup b + 1 (another XCD)
//   and checks it.
//   mode L: N dependent launches of G workgroups                     -> us per level
//   mode P: ONE launch of G workgroups, N levels, grid barrier       -> us per level   (G must be resident at once)
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier_probe.hip -o /tmp/grid_probe && /tmp/grid_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kSlice = 64;

// Double-buffered by level parity so that a workgroup running ahead cannot overwrite what a slower neighbour still has to read.
__global__ void __launch_bounds__(256) k_level(uint32_t* buf0, uint32_t* buf1, uint32_t* err, uint32_t level)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    uint32_t* wr = (level & 1) ? buf1 : buf0;
    const uint32_t* rd = (level & 1) ? buf0 : buf1;
    if (t < (uint32_t)kSlice) {
        if (level > 0) {
            const uint32_t v = rd[(size_t)((b + 1) % nb) * kSlice + t];
            if (v != ((level - 1) * 2654435761u ^ (((b + 1) % nb) * kSlice + t))) atomicAdd(err, 1u);
        }
        wr[(size_t)b * kSlice + t] = level * 2654435761u ^ (b * kSlice + t);
    }
}

__global__ void __launch_bounds__(256) k_persistent(uint32_t* buf0, uint32_t* buf1, uint32_t* err, uint32_t* bar, uint32_t nlevels, int fences)
{
    const uint32_t nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    for (uint32_t level = 0; level < nlevels; ++level) {
        uint32_t* wr = (level & 1) ? buf1 : buf0;
        const uint32_t* rd = (level & 1) ? buf0 : buf1;
        if (t < (uint32_t)kSlice) {
            if (level > 0) {
                const uint32_t v = rd[(size_t)((b + 1) % nb) * kSlice + t];
                if (v != ((level - 1) * 2654435761u ^ (((b + 1) % nb) * kSlice + t))) atomicAdd(err, 1u);
            }
            wr[(size_t)b * kSlice + t] = level * 2654435761u ^ (b * kSlice + t);
        }
        // grid barrier: monotonic counter, one arrival per workgroup
        __syncthreads();
        if (t == 0) {
            if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = nb * (level + 1);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main(int argc, char** argv)
{
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 260;
    uint32_t *buf0, *buf1, *err, *bar;
    const int grids[] = {64, 128, 256, 512, 1024};      // (<= half of what is resident at once: 256 CUs x 8 workgroups of 256)
    CK(hipMalloc((void**)&buf0, 2048 * kSlice * 4)); CK(hipMalloc((void**)&buf1, 2048 * kSlice * 4));
    CK(hipMalloc((void**)&err, 4)); CK(hipMalloc((void**)&bar, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : grids) {
        for (int mode = 0; mode < 3; ++mode) {           // 0: launches, 1: persistent with fences, 2: persistent without (how much is the fence?)
            float best = 1e30f;
            uint32_t herr = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(err, 0, 4)); CK(hipMemset(bar, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, nullptr));
                if (mode == 0) for (uint32_t l = 0; l < N; ++l) hipLaunchKernelGGL(k_level, dim3(G), dim3(256), 0, nullptr, buf0, buf1, err, l);
                else hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, nullptr, buf0, buf1, err, bar, N, mode == 1 ? 1 : 0);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            }
            printf("G = %4d  %-34s %7.2f us per level   (%u stale reads in the last run of %u levels)\n", G,
                   mode == 0 ? "dependent launches" : mode == 1 ? "persistent, release/acquire fences" : "persistent, NO fences (may be stale)",
                   best * 1e3f / (float)N, herr, N);
        }
    }
    return 0;
}
