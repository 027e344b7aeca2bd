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
