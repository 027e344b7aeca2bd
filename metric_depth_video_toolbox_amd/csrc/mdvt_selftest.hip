// mdvt_selftest.hip -- device self-test of the correctly rounded reciprocal / division helpers (mdvt_device.h)
// against the compiler's IEEE expansion (-fhip-fp32-correctly-rounded-divide-sqrt), exhaustively over the operand
// range.  Called through mdvt_selftest() by tests/test_gpu_arith.py.
#include "mdvt_device.h"

namespace mdvt {

constexpr uint32_t kSelfLo = 0x2F800000u;      // 2^-32
constexpr uint32_t kSelfCount = 0x20000000u;   // 64 binades

__global__ void __launch_bounds__(256) k_selftest_rcp(unsigned long long* mism)
{
    unsigned long long bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < kSelfCount; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(kSelfLo + i);
        const float want = 1.0f / x;
        const float got = rcp_exact(x);
        bad += __float_as_uint(want) != __float_as_uint(got);
    }
    if (bad) atomicAdd(mism, bad);
}

__global__ void __launch_bounds__(256) k_selftest_div(unsigned long long* mism, unsigned long long seed)
{
    unsigned long long bad = 0;
    // 16 numerators: a few typical disparity numerators (fx * ipd/2) and pseudo-random ones in [2^-8, 2^16)
    float num[16];
    unsigned long long s = seed * 6364136223846793005ull + 1442695040888963407ull;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t mant = (uint32_t)(s >> 20) & 0x7FFFFFu, ex = 119u + (uint32_t)((s >> 50) % 24u);
        num[k] = __uint_as_float((ex << 23) | mant);
    }
    num[0] = 75.323463f; num[1] = 150.64693f; num[2] = 25.107821f; num[3] = 1.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < kSelfCount; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(kSelfLo + i);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float inv, q;
            rcp_div_exact(num[k], fast_operand(num[k]), x, inv, q);
            const float wi = 1.0f / x, wq = num[k] / x;
            bad += (__float_as_uint(wi) != __float_as_uint(inv)) || (__float_as_uint(wq) != __float_as_uint(q));
        }
    }
    if (bad) atomicAdd(mism, bad);
}

// v_cvt_pk_u8_f32 (what shade_px uses) against the decree's rint / clamp / NaN -> 0 for every f32 bit pattern
__global__ void __launch_bounds__(256) k_selftest_u8(unsigned long long* mism)
{
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < 0x100000000ull;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float v = __uint_as_float((uint32_t)i);
        bad += __builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u) != shade_channel_reference(v);
        bad += __builtin_amdgcn_cvt_pk_u8_f32(v, 2u, 0x11223344u) != ((shade_channel_reference(v) << 16) | 0x11003344u);
    }
    if (bad) atomicAdd(mism, bad);
}

// ---- r05 diagnosis (mdvt_debug_read, what = 1; tuning build): is a device block coherent across the chip's eight XCDs from one kernel to
// the next?  k_coh_fill: workgroup b writes slice b (64 dwords: tag ^ index) with plain stores and posts one device-scope atomic add
// into the slice's last dword; k_coh_check (the next kernel in the stream): workgroup b reads slice b + 1 -- written by the workgroup on
// the next XCD (workgroups are dealt round-robin: b % 8) -- and counts wrong words per (writer XCD, reader XCD).
constexpr int kCohSlice = 64;
__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu; }      // HW_REG_XCC_ID[3:0]

__global__ void __launch_bounds__(64) k_coh_fill(uint32_t* blk, uint32_t nslices, uint32_t tag, uint32_t* writer_xcc)
{
    const uint32_t b = blockIdx.x;
    if (b >= nslices) return;
    uint32_t* sl = blk + (size_t)b * kCohSlice;
    sl[threadIdx.x] = threadIdx.x == kCohSlice - 1 ? 0u : tag ^ (b * kCohSlice + threadIdx.x);
    if (threadIdx.x == 0) writer_xcc[b] = xcc_id();
}
__global__ void __launch_bounds__(64) k_coh_add(uint32_t* blk, uint32_t nslices)
{
    const uint32_t b = blockIdx.x;
    if (b >= nslices) return;
    atomicAdd(&blk[(size_t)((b + 3u) % nslices) * kCohSlice + kCohSlice - 1], 1u);      // one add per lane: 64 per slice, from another XCD than the zero's
}
__global__ void __launch_bounds__(64) k_coh_check(const uint32_t* blk, uint32_t nslices, uint32_t tag, const uint32_t* writer_xcc, uint32_t* out)
{
    const uint32_t b = (blockIdx.x + 1u) % nslices;
    if (blockIdx.x >= nslices) return;
    const uint32_t* sl = blk + (size_t)b * kCohSlice;
    const uint32_t v = sl[threadIdx.x];
    const uint32_t want = threadIdx.x == kCohSlice - 1 ? 64u : tag ^ (b * kCohSlice + threadIdx.x);
    if (v != want) {
        const uint32_t w = writer_xcc[b] & 7u, r = xcc_id() & 7u;
        atomicAdd(&out[threadIdx.x == kCohSlice - 1 ? 64 + r : w * 8 + r], 1u);
        if (atomicAdd(&out[72], 1u) == 0u) { out[73] = v; out[74] = want; out[75] = b * kCohSlice + threadIdx.x; }
    }
}
// d_out: 80 dwords (zeroed here): [w * 8 + r] wrong pattern words by writer / reader XCD, [64 + r] wrong atomic sums by reader XCD,
// [72] total, [73..75] first wrong word seen / wanted / dword index.  d_xcc: nslices dwords of scratch.
hipError_t launch_coherence_test(uint32_t* blk, size_t dwords, uint32_t tag, uint32_t* d_xcc, uint32_t* d_out, hipStream_t s)
{
    const uint32_t nslices = (uint32_t)(dwords / kCohSlice);
    hipError_t e = hipMemsetAsync(d_out, 0, 80 * sizeof(uint32_t), s);
    if (e != hipSuccess || nslices == 0) return e;
    hipLaunchKernelGGL(k_coh_fill, dim3(nslices), dim3(64), 0, s, blk, nslices, tag, d_xcc);
    hipLaunchKernelGGL(k_coh_add, dim3(nslices), dim3(64), 0, s, blk, nslices);
    hipLaunchKernelGGL(k_coh_check, dim3(nslices), dim3(64), 0, s, blk, nslices, tag, d_xcc, d_out);
    return hipGetLastError();
}

hipError_t launch_selftest(int which, unsigned long long seed, unsigned long long* d_mism, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_mism, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (which == 0) hipLaunchKernelGGL(k_selftest_rcp, dim3(8192), dim3(256), 0, s, d_mism);
    else if (which == 1) hipLaunchKernelGGL(k_selftest_div, dim3(8192), dim3(256), 0, s, d_mism, seed);
    else hipLaunchKernelGGL(k_selftest_u8, dim3(16384), dim3(256), 0, s, d_mism);
    return hipGetLastError();
}

}  // namespace mdvt
