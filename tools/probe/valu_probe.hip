// valu_probe.hip -- how fast does one MI355X SIMD issue the VALU instructions the mesh rasterisers are made of?
//
// DESIGN.md's "instruction bound" argument for k_mesh_band / k_mesh_raster_small needs the issue rate of a wave64 VALU
// instruction: /opt/skills/guides/MI355X_MICROARCH.md says 2 cycles (SIMD-32), the 157.3 TFLOP/s vector peak is also what a
// 4-cycle SIMD-16 with double-rate packed f32 gives (64 FLOP/clk/SIMD either way), and SQ_ACTIVE_INST_VALU counts in
// quad-cycles, so the PMC numbers cannot tell the two apart.  This probe can: W waves per SIMD on every SIMD of the chip run
// a long stream of INDEPENDENT instructions of one kind (8 accumulator chains per lane, so latency is covered from one
// wave up); wave-instructions per second per SIMD / shader clock = issue cycles per instruction.
//
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kChains = 8;        // independent dependency chains per lane
constexpr int kUnroll = 4;        // x kChains instructions per loop iteration

// One instruction kind per OP; every asm statement is one VALU instruction on chain c.
template <int OP>
__device__ __forceinline__ void op1(float& a, float& b, float x, float y)
{
    // a, b: the chain's registers (b only for 64-bit / packed kinds); x, y: loop-invariant operands
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 5) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a));
    if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x));
    if (OP == 11) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 12) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x));
    if (OP == 13) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a) : "v"(x));
    if (OP == 14) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
    if (OP == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");
    if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(x) : "s20", "s21");
    if (OP == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(x), "v"(y) : "vcc");    // 2 instructions
    if (OP == 18) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a) : "v"(x), "v"(y) : "s20", "s21");   // 2 instructions
    if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 20) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(x));
    if (OP == 23) asm volatile("v_rndne_f32 %0, %0" : "+v"(a));
    if (OP == 24) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
    if (OP == 25) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a) : "v"(x));
    if (OP == 26) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 27) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 28) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 29) asm volatile("v_ashrrev_i32 %0, 8, %0" : "+v"(a));
    // pairs on two chains: do a 4-cycle-class and a fast-class instruction share one issue pipe (times add) or not (max)?
    if (OP == 30) asm volatile("v_mul_u32_u24 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(x));
    if (OP == 31) asm volatile("v_cvt_f32_i32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(x), "v"(y));
    if (OP == 32) asm volatile("v_mul_u32_u24 %0, %0, %2\n\tv_perm_b32 %1, %1, %2, %2" : "+v"(a), "+v"(b) : "v"(x));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__device__ __forceinline__ void op2(f32x2& a, f32x2 x, f32x2 y)
{
    if (OP == 100) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 101) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 102) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    if (OP == 103) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x.x), "v"(y.x) : "vcc");
    if (OP == 104) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    if (OP == 105) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a));
}

template <int OP>
__global__ void __launch_bounds__(512) k_stream(float* out, int iters, float x, float y)
{
    float acc[kChains], accb[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) { acc[c] = (float)(threadIdx.x + c) * 1e-3f + 1.0f; accb[c] = 0.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int c = 0; c < kChains; ++c) op1<OP>(acc[c], accb[c], x, y);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += acc[c] + accb[c];
    if (s == 123456.789f) out[threadIdx.x] = s;     // never true: keeps the chains alive
}

template <int OP>
__global__ void __launch_bounds__(512) k_stream2(float* out, int iters, float x, float y)
{
    f32x2 acc[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc[c] = f32x2{(float)(threadIdx.x + c) * 1e-3f + 1.0f, 1.0f};
    const f32x2 xx = {x, x}, yy = {y, y};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int c = 0; c < kChains; ++c) op2<OP>(acc[c], xx, yy);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += acc[c].x + acc[c].y;
    if (s == 123456.789f) out[threadIdx.x] = s;
}

// a dependent chain: one accumulator, issue-to-issue latency of a wave alone on its SIMD
template <int OP>
__global__ void __launch_bounds__(64) k_chain(float* out, int iters, float x, float y)
{
    float a = (float)threadIdx.x * 1e-3f + 1.0f, b = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) op1<OP>(a, b, x, y);
    }
    if (a + b == 123456.789f) out[threadIdx.x] = a;
}

struct Row { const char* name; void (*fn)(float*, int, float, float); int lanes_per_instr_x; };

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock attribute %.0f MHz\n", prop.gcnArchName, cus, clock_khz / 1e3);
    float* out = nullptr;
    CK(hipMalloc((void**)&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    const double instr_per_wave = (double)iters * kUnroll * kChains;

#define ROW(NAME, K, OP) {NAME, (void (*)(float*, int, float, float))K<OP>, 0}
    const Row rows[] = {
        ROW("v_fma_f32", k_stream, 0), ROW("v_mul_f32", k_stream, 1), ROW("v_add_f32", k_stream, 2),
        ROW("v_pk_fma_f32", k_stream2, 100), ROW("v_pk_mul_f32", k_stream2, 101), ROW("v_pk_add_f32", k_stream2, 102),
        ROW("v_mul_u32_u24", k_stream, 3), ROW("v_mad_u32_u24", k_stream, 4), ROW("v_mul_lo_u32", k_stream, 7),
        ROW("v_mad_u64_u32", k_stream2, 103), ROW("v_add_u32", k_stream, 8), ROW("v_lshlrev_b32", k_stream, 14),
        ROW("v_lshlrev_b64", k_stream2, 105), ROW("v_cvt_f32_i32", k_stream, 5), ROW("v_rcp_f32", k_stream, 6),
        ROW("v_perm_b32", k_stream, 9), ROW("v_cndmask_b32", k_stream, 10), ROW("v_min3_i32", k_stream, 11),
        ROW("v_cmp_lt_f32", k_stream, 15), ROW("v_mov_b32_dpp wave_shl", k_stream, 12), ROW("v_cvt_pk_u8_f32", k_stream, 13),
        ROW("v_fma_f64", k_stream2, 104),
        ROW("v_cndmask_b32_e64 (sgpr mask)", k_stream, 16), ROW("v_cmp + v_cndmask vcc (PAIRS/ns)", k_stream, 17),
        ROW("v_cmp + v_cndmask sgpr (PAIRS/ns)", k_stream, 18), ROW("v_max_f32", k_stream, 19), ROW("v_sub_f32", k_stream, 28),
        ROW("v_sub_u32", k_stream, 20), ROW("v_and_b32", k_stream, 21), ROW("v_mov_b32", k_stream, 22),
        ROW("v_rndne_f32", k_stream, 23), ROW("v_cvt_i32_f32", k_stream, 24), ROW("v_alignbit_b32", k_stream, 25),
        ROW("v_add3_u32", k_stream, 26), ROW("v_mad_i32_i24", k_stream, 27), ROW("v_ashrrev_i32", k_stream, 29),
        ROW("v_mul_u32_u24 + v_add_f32 (PAIRS)", k_stream, 30), ROW("v_cvt_f32_i32 + v_fma_f32 (PAIRS)", k_stream, 31),
        ROW("v_mul_u32_u24 + v_perm_b32 (PAIRS)", k_stream, 32),
    };
    printf("%-26s", "wave-instr/ns/SIMD at waves/SIMD =");
    const int wps[] = {1, 2, 4, 8};
    for (int w : wps) printf("%9d", w);
    printf("   cycles/instr @%.1f GHz (8 waves)\n", clock_khz / 1e6);
    for (const Row& r : rows) {
        printf("%-34s", r.name);
        double last = 0.0;
        for (int w : wps) {
            // w waves per SIMD = 4 w waves per CU: blocks of 64 * 4 w threads... one block per CU of 256 w threads (<= 512: two blocks)
            const int threads = 256 * w > 512 ? 512 : 256 * w;
            const int blocks = cus * (256 * w / threads);
            void (*fn)(float*, int, float, float) = r.fn;
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, 64, 1.0000001f, 1e-9f);     // warm
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001f, 1e-9f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double per_simd = instr_per_wave * w / (ms * 1e6);        // wave-instructions per ns per SIMD
            printf("%9.3f", per_simd);
            last = per_simd;
        }
        printf("   %6.2f\n", (clock_khz / 1e6) / last);
    }
    // dependent-chain latency, one wave on one SIMD of every CU
    printf("\ndependent chain, one wave per CU: ns per instruction (cycles @ clock attribute)\n");
    const Row chains[] = {ROW("v_fma_f32", k_chain, 0), ROW("v_mul_u32_u24", k_chain, 3), ROW("v_cvt_f32_i32", k_chain, 5),
                          ROW("v_rcp_f32", k_chain, 6), ROW("v_mul_lo_u32", k_chain, 7), ROW("v_mov_b32_dpp wave_shl", k_chain, 12)};
    for (const Row& r : chains) {
        void (*fn)(float*, int, float, float) = r.fn;
        hipLaunchKernelGGL(fn, dim3(cus), dim3(64), 0, 0, out, 16, 1.0000001f, 1e-9f);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(cus), dim3(64), 0, 0, out, 8192, 1.0000001f, 1e-9f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms * 1e6 / (8192.0 * 32.0);
        printf("%-34s %7.3f ns  (%5.2f cycles)\n", r.name, ns, ns * clock_khz / 1e6);
    }
    (void)argc; (void)argv;
    return 0;
}
