// mdvt_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the stereo-rerender hot path.
//
// Replaces the NumPy + Open3D/OpenGL stages of the reference's frame loop (stereo_rerender.py:512-907):
//   decode (dfh:63-75, 13-24) -> master scale (sr:541) -> unproject (dmt:1112-1133) -> eye/pose
//   transform (sr:615-619, 724-725, 832-836) -> z-buffered render (dmt:1422-1572) -> colour-key hole
//   mask (sr:740, 793) -> edge-point splat (sr:745-814).
//
// The path is an HBM-bound gather/scatter: no MFMA.  What matters is (i) every HBM byte is read and
// written once, coalesced (12 B/lane dwordx3 = 768 contiguous bytes per wave), (ii) the z-buffer
// lives in LDS for the row-local (pure stereo shift) case so the atomics never leave the CU,
// (iii) one launch covers a whole batch of frames (n_frames * H workgroups >> 256 CUs).
//
// Compiled with -ffp-contract=off: see the arithmetic decree in mdvt_device.h / DESIGN.md.
#include "mdvt_device.h"
#include <type_traits>

#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace mdvt {

// Sections that do not depend on the sub-pixel grid are compiled once, with the default grid (-DMDVT_SUBPIX_BITS=8), into
// namespace mdvt; the rasterising sections once per grid into mdvt::MDVT_GRID (mdvt_internal.h).
#if MDVT_SUBPIX_BITS == 8
// =================================================================================================
// depth codec (dfh)
// =================================================================================================

__global__ void k_decode_depth(const uint8_t* __restrict__ rgb, size_t rgb_pitch, float* __restrict__ out,
                               size_t out_pitch, int W, int H, float mult, float scale)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= W || i >= H) return;
    const uint32_t px = load_px_bytes(rgb + (size_t)i * rgb_pitch, j);
    float* orow = (float*)((uint8_t*)out + (size_t)i * out_pitch);
    orow[j] = decode_z(code16_of(px), mult, scale);
}

// 4 pixels per thread: 12 B coalesced load, 16 B coalesced store.
__global__ void k_decode_depth4(const uint8_t* __restrict__ rgb, size_t rgb_pitch, float* __restrict__ out,
                                size_t out_pitch, int W4, int H, float mult, float scale)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (g >= W4 || i >= H) return;
    const uint32_t* src = (const uint32_t*)(rgb + (size_t)i * rgb_pitch) + 3 * (size_t)g;
    uint32_t px[4];
    unpack4(src[0], src[1], src[2], px);
    float4 z;
    z.x = decode_z(code16_of(px[0]), mult, scale);
    z.y = decode_z(code16_of(px[1]), mult, scale);
    z.z = decode_z(code16_of(px[2]), mult, scale);
    z.w = decode_z(code16_of(px[3]), mult, scale);
    ((float4*)((uint8_t*)out + (size_t)i * out_pitch))[g] = z;
}

hipError_t launch_decode_depth(const uint8_t* rgb, size_t rgb_pitch, float* out, size_t out_pitch, int W, int H,
                               float mult, float scale, hipStream_t s)
{
    const bool vec = (W % 4 == 0) && (rgb_pitch % 4 == 0) && (out_pitch % 16 == 0) &&
                     ((uintptr_t)rgb % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec) {
        const int W4 = W / 4;
        dim3 grid((W4 + 255) / 256, H);
        hipLaunchKernelGGL(k_decode_depth4, grid, dim3(256), 0, s, rgb, rgb_pitch, out, out_pitch, W4, H, mult, scale);
    } else {
        dim3 grid((W + 255) / 256, H);
        hipLaunchKernelGGL(k_decode_depth, grid, dim3(256), 0, s, rgb, rgb_pitch, out, out_pitch, W, H, mult, scale);
    }
    return hipGetLastError();
}

// dfh:5-11: clip to [0,max] in f32, f64 multiply by 255^4/max, truncate to u32; dfh:53-55: R = G = byte 3,
// B = byte 2.
__global__ void k_encode_depth(const float* __restrict__ depth, size_t depth_pitch, uint8_t* __restrict__ rgb,
                               size_t rgb_pitch, int W, int H, double multi, float fmax_depth, int bgr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= W || i >= H) return;
    float d = ((const float*)((const uint8_t*)depth + (size_t)i * depth_pitch))[j];
    if (d > fmax_depth) d = fmax_depth;
    if (d < 0.0f) d = 0.0f;
    const double e = multi * (double)d;
    const uint32_t code = (e >= 0.0 && e < 4294967296.0) ? (uint32_t)e : 0u;     // NaN -> 0
    const uint32_t hi = code >> 24, lo = (code >> 16) & 0xFFu;
    const uint32_t px = bgr ? (lo | (hi << 8) | (hi << 16)) : (hi | (hi << 8) | (lo << 16));
    store_px_bytes(rgb + (size_t)i * rgb_pitch, j, px);
}

hipError_t launch_encode_depth(const float* depth, size_t depth_pitch, uint8_t* rgb, size_t rgb_pitch, int W, int H,
                               double max_depth, int bgr, hipStream_t s)
{
    dim3 grid((W + 255) / 256, H);
    hipLaunchKernelGGL(k_encode_depth, grid, dim3(256), 0, s, depth, depth_pitch, rgb, rgb_pitch, W, H,
                       4228250625.0 / max_depth, (float)max_depth, bgr);
    return hipGetLastError();
}

// =================================================================================================
// 89-degree oblique-triangle filter (dmt:1283-1294, 1339-1344), f64 exactly as NumPy >= 2 evaluates it
// =================================================================================================

__device__ __forceinline__ bool tri_oblique(const double (&a)[3], const double (&b)[3], const double (&c)[3])
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const double e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const double nx = e1y * e2z - e1z * e2y;
    const double ny = e1z * e2x - e1x * e2z;
    const double nz = e1x * e2y - e1y * e2x;
    const double vx = -((a[0] + b[0]) + c[0]) / 3.0;
    const double vy = -((a[1] + b[1]) + c[1]) / 3.0;
    const double vz = -((a[2] + b[2]) + c[2]) / 3.0;
    const double dot = (nx * vx + ny * vy) + nz * vz;
    const double len_n = sqrt((nx * nx + ny * ny) + nz * nz);
    const double len_v = sqrt((vx * vx + vy * vy) + vz * vz);
    const double cosine = dot / (len_n * len_v + 1e-15);
    return cosine < 0x1.1df0b2b89dd37p-6;          // np.cos(np.radians(89.0))
}

// Division-free screening of the same test.  cos < c0  <=>  dot < c0 * (|n||v| + 1e-15); with the
// unnormalised view vector vs = a+b+c (v = -vs/3) this is  -n.vs < c0 * |n| |vs|  up to the 1e-15 term.
// Squaring removes the square roots.  The screening value carries ~1e-15 relative rounding error and
// ignores the 1e-15 term, which moves the threshold by the relative amount 1e-15 / (|n||v|): a triangle with
// |n||v| < 1e-8 (content nearer than ~0.3 m at 1080p: tiny triangles, tiny view vectors) is therefore NOT
// decided here, nor is one whose margin is below kScreenMargin (5e-7 in linear terms, against <= 1e-7 from the
// dropped term); both go through the exact formula.  Everything else provably gets the exact formula's decision.
constexpr double kScreenMargin = 1e-6;
constexpr double kScreenMinNNSS = 9e-16;          // (3 |n||v|)^2 for |n||v| = 1e-8

// returns 0 = valid, 1 = oblique (removed), 2 = undecided
__device__ __forceinline__ int tri_oblique_screen(const double (&a)[3], const double (&b)[3], const double (&c)[3])
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const double e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    // (explicit fused multiply-adds: the screen only has to be accurate, not bit-identical to anything, and the
    //  filter is bound by the f64 VALU rate -- the exact path below keeps the decree's one-rounding-per-node form)
    const double nx = fma(e1y, e2z, -(e1z * e2y));
    const double ny = fma(e1z, e2x, -(e1x * e2z));
    const double nz = fma(e1x, e2y, -(e1y * e2x));
    const double sx = (a[0] + b[0]) + c[0], sy = (a[1] + b[1]) + c[1], sz = (a[2] + b[2]) + c[2];
    const double d = -fma(nz, sz, fma(ny, sy, nx * sx));               // 3 * dot
    const double nn = fma(nz, nz, fma(ny, ny, nx * nx));
    const double ss = fma(sz, sz, fma(sy, sy, sx * sx));               // 9 * |v|^2
    const double c0 = 0x1.1df0b2b89dd37p-6;
    const double rhs = (c0 * c0) * (nn * ss);                          // (c0 |n| |vs|)^2
    if (!(nn * ss > kScreenMinNNSS)) return 2;                         // degenerate / zero depth / tiny: exact path
    const double lhs = d * fabs(d);                                    // signed square
    const double tol = kScreenMargin * rhs;
    if (lhs < rhs - tol) return 1;
    if (lhs > rhs + tol) return 0;
    return 2;
}

// f32 pre-screen of the same test (r04), from the closed form of a triangle whose vertices are P_k = z_k (u_k, v_k, 1) on the
// grid's rays u in {a, a + px}, v in {b, b + py}:  n = (P1 - P0) x (P2 - P0) and n . P_k = z0 z1 z2 det(r0, r1, r2) = -px py z0 z1 z2
// for both triangles of a cell, so with s = P0 + P1 + P2 (the view vector is -s / 3)
//     3 dot = -n . s = 3 px py z0 z1 z2 > 0,    tri1 (A, B, C): nx = py zA (zC - zB), ny = px zC (zB - zA),
//                                               tri2 (A, C, D): nx = py zC (zD - zA), ny = px zA (zC - zD),
//     nz = -(a nx + b ny) - px py z1 z2   (from n . P0),
// and the test cos < cos 89 is  (3 dot)^2 < c0^2 |n|^2 |s|^2.  No difference of nearly equal PRODUCTS is left (the generic
// cross product of the edge vectors cancels ~4 digits): every factor is an input or one f32 subtraction of inputs, |n|^2 and
// |s|^2 are sums of squares, and nz's one cancellation is bounded by eps (|a nx| + |b ny| + ...) <= eps (|a| + |b| + 1) |n|.
// With |a|, |b| <= 64 both sides are good to ~1e-5 relative in f32, the f64 formula of the reference's order differs from
// the exact value by ~1e-15, its 1e-15 term moves the threshold by < 1e-7 when (3 |n||v|)^2 > 9e-16: a margin of 1e-3
// decides the same way for sure.  The rest -- a fraction of a per cent of the triangles -- takes the f64 screen and, inside
// its own margin, the exact formula.  returns 0 = valid, 1 = oblique (removed), 2 = undecided
__device__ __forceinline__ int tri_oblique_screen_f32(float a, float b, float pp, float z0, float t12 /* pp z1 z2 */,
                                                      float nx, float ny, float sx, float sy, float sz)
{
    // (explicit fused multiply-adds: the screen only has to be accurate, not bit-identical to anything)
    const float nz = -__builtin_fmaf(a, nx, __builtin_fmaf(b, ny, t12));
    const float nn = __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx));
    const float ss = __builtin_fmaf(sz, sz, __builtin_fmaf(sy, sy, sx * sx));
    const float d3 = (3.0f * z0) * t12;
    const float lhs = d3 * d3;
    const float nnss = nn * ss;
    const float rhs = 0x1.3f61d0p-12f * nnss;                           // cos^2 89 degrees (f32: 6e-8 relative, inside the margin)
    // (no branches: 0 = valid, 1 = oblique (removed), 2 = undecided -- degenerate / zero depth / tiny / NaN / inside the margin)
    const bool ok = nnss > 1.0e-15f;
    const bool rem = ok && lhs < rhs * 0.999f, val = ok && lhs > rhs * 1.001f;
    return rem ? 1 : (val ? 0 : 2);
}

struct CellRaysF32 { float a, a1, b, b1, px, py; bool ok; };     // the f32 pre-screen's view of the cell's rays (ok: all within +-64)
__device__ __forceinline__ CellRaysF32 cell_rays_f32(double x0r, double x1r, double y0r, double y1r)
{
    CellRaysF32 r;
    r.a = (float)x0r; r.b = (float)y0r; r.a1 = (float)x1r; r.b1 = (float)y1r;
    r.px = (float)(x1r - x0r); r.py = (float)(y1r - y0r);
    r.ok = fabsf(r.a) <= 64.0f && fabsf(r.b) <= 64.0f && fabsf(r.a1) <= 64.0f && fabsf(r.b1) <= 64.0f;
    return r;
}

// f32 pre-screen of both triangles of a cell: s1 | s2 << 2, each 0 = valid, 1 = oblique (removed), 2 = undecided
__device__ __forceinline__ uint32_t edge_filter_prescreen(const CellRaysF32& r, float zA, float zB, float zC, float zD)
{
    if (!r.ok) return 2u | (2u << 2);
    // tri1 = (A, B, C) = rays (a, b), (a, b'), (a', b');  tri2 = (A, C, D) = (a, b), (a', b'), (a', b)
    const float pp = r.px * r.py;
    const int s1 = tri_oblique_screen_f32(r.a, r.b, pp, zA, (pp * zB) * zC, (r.py * zA) * (zC - zB), (r.px * zC) * (zB - zA),
                                          __builtin_fmaf(r.a1, zC, r.a * (zA + zB)), __builtin_fmaf(r.b1, zB + zC, r.b * zA), (zA + zB) + zC);
    const int s2 = tri_oblique_screen_f32(r.a, r.b, pp, zA, (pp * zC) * zD, (r.py * zC) * (zD - zA), (r.px * zA) * (zC - zD),
                                          __builtin_fmaf(r.a1, zC + zD, r.a * zA), __builtin_fmaf(r.b1, zC, r.b * (zA + zD)), (zA + zC) + zD);
    return (uint32_t)s1 | ((uint32_t)s2 << 2);
}

// Both triangles of cell (i, j): bit 0 = tri1 (A, B, C) removed, bit 1 = tri2 (A, C, D) removed.  `pre` = edge_filter_prescreen's
// verdict; x0r.. are the cell's ray coordinates (g - c) * (1 / f) in f64 (the f64 screen; the exact path recomputes the vertices
// in the reference's own order).
__device__ __forceinline__ uint32_t edge_filter_cell(const FrameDev& f, int i, int j, int of_by_one, double x0r, double x1r, double y0r,
                                                     double y1r, uint32_t pre, float zA, float zB, float zC, float zD)
{
    int s1 = (int)(pre & 3u), s2 = (int)(pre >> 2);
    if (s1 == 2 || s2 == 2) {
        asm volatile("; f64 screen" ::: "memory");
        const double dA = (double)zA, dB = (double)zB, dC = (double)zC, dD = (double)zD;
        const double A[3] = {x0r * dA, y0r * dA, dA};
        const double B[3] = {x0r * dB, y1r * dB, dB};
        const double Cc[3] = {x1r * dC, y1r * dC, dC};
        const double D[3] = {x1r * dD, y0r * dD, dD};
        if (s1 == 2) s1 = tri_oblique_screen(A, B, Cc);      // tri1 = (v[i,j], v[i+1,j], v[i+1,j+1])
        if (s2 == 2) s2 = tri_oblique_screen(A, Cc, D);      // tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])
    }
    if (s1 == 2 || s2 == 2) {
        // exact path: the reference's own evaluation order (dmt:1127-1128, 1283-1294)
        double Ae[3], Be[3], Ce[3], De[3];
        vertex_f64(f, i, j, of_by_one, zA, Ae);
        vertex_f64(f, i + 1, j, of_by_one, zB, Be);
        vertex_f64(f, i + 1, j + 1, of_by_one, zC, Ce);
        vertex_f64(f, i, j + 1, of_by_one, zD, De);
        if (s1 == 2) s1 = tri_oblique(Ae, Be, Ce) ? 1 : 0;
        if (s2 == 2) s2 = tri_oblique(Ae, Ce, De) ? 1 : 0;
    }
    return (s1 == 1 ? 1u : 0u) | (s2 == 1 ? 2u : 0u);
}

__device__ __forceinline__ void edge_filter_mark_unused(uint8_t* u, int W, int i, int j, uint32_t inv)
{
    const size_t a = (size_t)i * W + j;
    u[a] = 1;                          // A
    u[a + W + 1] = 1;                  // C
    if (inv & 1u) u[a + W] = 1;        // B
    if (inv & 2u) u[a + 1] = 1;        // D
}

// One thread per grid cell: both triangles of the cell.  `unused` must be zeroed beforehand.  (Any width / alignment.)
// NOTE f.sx / f.sy hold the mesh grid scale only when the frame was prepared for mesh mode; the host
// passes scale factors explicitly so the filter can be run standalone for either grid.
__global__ void __launch_bounds__(128) k_edge_filter(const uint8_t* __restrict__ depth_rgb, size_t pitch, size_t stride,
                              const FrameDev* __restrict__ fp, int frame0, int W, int H, int of_by_one,
                              float sx, float sy,
                              uint8_t* __restrict__ tri_invalid, size_t tri_stride,
                              uint8_t* __restrict__ unused, size_t unused_stride)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int fr = blockIdx.z;
    if (j >= W - 1 || i >= H - 1) return;
    const uint8_t* r0 = depth_rgb + (size_t)(frame0 + fr) * stride + (size_t)i * pitch;
    const uint8_t* r1 = r0 + pitch;
    const uint32_t pA = load_px_bytes(r0, j), pD = load_px_bytes(r0, j + 1);
    const uint32_t pB = load_px_bytes(r1, j), pC = load_px_bytes(r1, j + 1);
    FrameDev f = fp[frame0 + fr];
    f.sx = sx; f.sy = sy;
    const float zA = decode_z(code16_of(pA), f.mult, f.scale);
    const float zD = decode_z(code16_of(pD), f.mult, f.scale);
    const float zB = decode_z(code16_of(pB), f.mult, f.scale);
    const float zC = decode_z(code16_of(pC), f.mult, f.scale);
    // screening vertices: the same unprojection with a reciprocal instead of the two divisions
    const double rfx = f.rKd[0], rfy = f.rKd[1];
    const double x0 = (of_by_one ? (double)((float)j * f.sx) : (double)j) - f.Kd[2];
    const double x1 = (of_by_one ? (double)((float)(j + 1) * f.sx) : (double)(j + 1)) - f.Kd[2];
    const double y0 = (of_by_one ? (double)((float)i * f.sy) : (double)i) - f.Kd[3];
    const double y1 = (of_by_one ? (double)((float)(i + 1) * f.sy) : (double)(i + 1)) - f.Kd[3];
    const double x0r = x0 * rfx, x1r = x1 * rfx, y0r = y0 * rfy, y1r = y1 * rfy;
    const uint32_t inv = edge_filter_cell(f, i, j, of_by_one, x0r, x1r, y0r, y1r, edge_filter_prescreen(cell_rays_f32(x0r, x1r, y0r, y1r), zA, zB, zC, zD), zA, zB, zC, zD);
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    const size_t cell = (size_t)i * (W - 1) + j;
    if (tri_invalid) {
        uint8_t* t = tri_invalid + (size_t)fr * tri_stride;
        t[cell] = inv & 1u;
        t[ncell + cell] = (inv >> 1) & 1u;
    }
    if (unused && inv) edge_filter_mark_unused(unused + (size_t)fr * unused_stride, W, i, j, inv);
}

// The same for dword-addressable rows (W % 4 == 0, 4-byte aligned base / pitch / stride): a thread takes FOUR cells of a
// row -- five pixels of two rows as 2 x 4 dwords instead of 2 x 10 byte loads, the eight validity bytes as two dword stores.
// (r04: with the f32 pre-screen no f64 instruction runs on ordinary content, and the one-cell kernel turned out to be bound
// by its byte accesses, not by the f64 rate its design assumed: 10.4 -> see DESIGN.md us per 1080p frame.)
typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));
__global__ void __launch_bounds__(128) k_edge_filter4(const uint8_t* __restrict__ depth_rgb, size_t pitch, size_t stride,
                              const FrameDev* __restrict__ fp, int frame0, int W, int H, int of_by_one,
                              float sx, float sy,
                              uint8_t* __restrict__ tri_invalid, size_t tri_stride,
                              uint8_t* __restrict__ unused, size_t unused_stride)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int fr = blockIdx.z;
    const int j0 = 4 * g;
    if (j0 >= W - 1 || i >= H - 1) return;
    const uint32_t* r0 = (const uint32_t*)(depth_rgb + (size_t)(frame0 + fr) * stride + (size_t)i * pitch) + 3 * g;
    const uint32_t* r1 = (const uint32_t*)((const uint8_t*)r0 + pitch);
    const bool five = j0 + 4 < W;                          // (the last group of a row has no fifth pixel -- and no fourth cell)
    uint32_t p0[5], p1[5];
    unpack4(r0[0], r0[1], r0[2], *(uint32_t(*)[4])p0);
    unpack4(r1[0], r1[1], r1[2], *(uint32_t(*)[4])p1);
    p0[4] = five ? r0[3] & 0xFFFFFFu : 0u;
    p1[4] = five ? r1[3] & 0xFFFFFFu : 0u;
    FrameDev f = fp[frame0 + fr];
    f.sx = sx; f.sy = sy;
    float z0[5], z1[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        z0[q] = decode_z(code16_of(p0[q]), f.mult, f.scale);
        z1[q] = decode_z(code16_of(p1[q]), f.mult, f.scale);
    }
    const double rfx = f.rKd[0], rfy = f.rKd[1];
    const double y0r = ((of_by_one ? (double)((float)i * f.sy) : (double)i) - f.Kd[3]) * rfy;
    const double y1r = ((of_by_one ? (double)((float)(i + 1) * f.sy) : (double)(i + 1)) - f.Kd[3]) * rfy;
    double xr[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) xr[q] = ((of_by_one ? (double)((float)(j0 + q) * f.sx) : (double)(j0 + q)) - f.Kd[2]) * rfx;
    const int ncells = five ? 4 : 3;
    float fx[5];
    const float fb = (float)y0r, fb1 = (float)y1r, fpy = (float)(y1r - y0r);
    bool rays_ok = fabsf(fb) <= 64.0f && fabsf(fb1) <= 64.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) { fx[q] = (float)xr[q]; rays_ok = rays_ok && fabsf(fx[q]) <= 64.0f; }
    uint32_t w1 = 0, w2 = 0, any = 0;
    uint32_t inv[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
    // all eight triangles through the f32 pre-screen first (straight-line code); the f64 paths behind ONE branch
    uint32_t und = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < ncells) {
            CellRaysF32 r;
            r.a = fx[q]; r.a1 = fx[q + 1]; r.b = fb; r.b1 = fb1; r.px = (float)(xr[q + 1] - xr[q]); r.py = fpy;
            r.ok = rays_ok;
            pre[q] = edge_filter_prescreen(r, z0[q], z1[q], z1[q + 1], z0[q + 1]);
        }
        und |= pre[q] & 0xAu;                                // (a 2 in either field)
        inv[q] = (pre[q] & 1u) | ((pre[q] >> 1) & 2u);       // decided: 1 -> removed
    }
    if (und) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (pre[q] & 0xAu) inv[q] = edge_filter_cell(f, i, j0 + q, of_by_one, xr[q], xr[q + 1], y0r, y1r, pre[q], z0[q], z1[q], z1[q + 1], z0[q + 1]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w1 |= (inv[q] & 1u) << (8 * q);
        w2 |= ((inv[q] >> 1) & 1u) << (8 * q);
        any |= inv[q];
    }
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    const size_t cell = (size_t)i * (W - 1) + j0;
    if (tri_invalid) {
        uint8_t* t = tri_invalid + (size_t)fr * tri_stride;
        if (five) {
            *(u32_unaligned_t*)(t + cell) = w1;
            *(u32_unaligned_t*)(t + ncell + cell) = w2;
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) { t[cell + q] = (w1 >> (8 * q)) & 1u; t[ncell + cell + q] = (w2 >> (8 * q)) & 1u; }
        }
    }
    if (unused && any) {
        uint8_t* u = unused + (size_t)fr * unused_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (inv[q]) edge_filter_mark_unused(u, W, i, j0 + q, inv[q]);
    }
}

// Zero `bytes` bytes at `p` (any alignment): 16-byte stores over the aligned body, byte stores at its two ends.  Replaces the per-set
// hipMemsetAsync of the 89-degree filter's flag plane (two runtime fill launches, 13 us per 16 MB): one launch; measured, a single
// product-default frame per call 148.9 -> 146.7 us, 32 frames per call unchanged (the fills hid behind the other bank's walk).
__global__ void __launch_bounds__(256) k_zero_bytes(uint8_t* p, size_t bytes)
{
    const size_t head = min(bytes, (size_t)((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u));
    const size_t body = (bytes - head) >> 4, tail0 = head + (body << 4);
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x, nt = (size_t)gridDim.x * 256u;
    for (size_t k = t; k < body; k += nt) q[k] = make_uint4(0u, 0u, 0u, 0u);
    if (t < head) p[t] = 0;
    if (t < bytes - tail0) p[tail0 + t] = 0;
}
hipError_t launch_zero_bytes(void* p, size_t bytes, hipStream_t s)
{
    if (!bytes) return hipSuccess;
    const size_t blocks = (bytes / 16 + 255) / 256;
    hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks))), dim3(256), 0, s, (uint8_t*)p, bytes);
    return hipGetLastError();
}

hipError_t launch_edge_filter(const uint8_t* depth_rgb, size_t pitch, size_t stride, const FrameDev* fp, int frame0,
                              int n, int W, int H, int of_by_one, uint8_t* tri_invalid, size_t tri_stride,
                              uint8_t* unused, size_t unused_stride, hipStream_t s)
{
    const float sx = of_by_one ? (float)(((double)W + 1.0) / (double)W) : 1.0f;
    const float sy = of_by_one ? (float)(((double)H + 1.0) / (double)H) : 1.0f;
    const bool dwords = W % 4 == 0 && W >= 8 && pitch % 4 == 0 && stride % 4 == 0 && ((uintptr_t)depth_rgb % 4) == 0;
    if (dwords) {
        dim3 grid((W / 4 + 127) / 128, H - 1, n);
        hipLaunchKernelGGL(k_edge_filter4, grid, dim3(128), 0, s, depth_rgb, pitch, stride, fp, frame0, W, H, of_by_one,
                           sx, sy, tri_invalid, tri_stride, unused, unused_stride);
    } else {
        dim3 grid((W - 1 + 127) / 128, H - 1, n);
        hipLaunchKernelGGL(k_edge_filter, grid, dim3(128), 0, s, depth_rgb, pitch, stride, fp, frame0, W, H, of_by_one,
                           sx, sy, tri_invalid, tri_stride, unused, unused_stride);
    }
    return hipGetLastError();
}

#endif  // grid-independent sections

namespace MDVT_GRID {
// =================================================================================================
// POINT MODE, pure stereo shift: one workgroup per (frame,row), z-buffer in LDS
// =================================================================================================
//
// Row locality: with K == Krender, no pose and no toe-in, v = i exactly and u = j +- dl/Z, so an
// output row depends on one input row.  Z is strictly monotone in the 16-bit depth code, so the
// whole fragment fits one 64-bit LDS word
//        key = code16 << 40 | j << 24 | R | G<<8 | B<<16
// whose unsigned minimum is "nearest Z, ties to the lower source column" -- and it carries the
// colour, so the resolve phase is a plain LDS read (no gather, no second pass over the inputs).
// HBM traffic = algorithmic bytes: 6 B/px in, 8 B/px out (+8 B/px with the optional depth planes).

template <int PX>   // pixels per thread-iteration: 4 (dwordx3 path) or 1 (byte path, any W / alignment)
struct RowIO;

template <>
struct RowIO<4> {
    static __device__ __forceinline__ void load(const uint8_t* row, int g, uint32_t (&px)[4])
    {
        const uint32_t* p = (const uint32_t*)row + 3 * (size_t)g;
        unpack4(p[0], p[1], p[2], px);
    }
    // for data that is read exactly once (non-temporal: does not displace what the caches hold)
    static __device__ __forceinline__ void load_nt(const uint8_t* row, int g, uint32_t (&px)[4])
    {
        const uint32_t* p = (const uint32_t*)row + 3 * (size_t)g;
        unpack4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), px);
    }
    // outputs are written once and not read back by the library: non-temporal stores
    static __device__ __forceinline__ void store_rgb(uint8_t* row, int g, const uint32_t (&px)[4])
    {
        uint32_t w0, w1, w2;
        pack4(px, w0, w1, w2);
        uint32_t* p = (uint32_t*)row + 3 * (size_t)g;
        __builtin_nontemporal_store(w0, p); __builtin_nontemporal_store(w1, p + 1); __builtin_nontemporal_store(w2, p + 2);
    }
    static __device__ __forceinline__ void store_mask(uint8_t* row, int g, const uint32_t (&m)[4])
    {
        __builtin_nontemporal_store(m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24), (uint32_t*)row + g);
    }
    static __device__ __forceinline__ void store_z(float* row, int g, const float (&z)[4])
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = {z[0], z[1], z[2], z[3]};
        __builtin_nontemporal_store(v, (f32x4*)row + g);
    }
    static __device__ __forceinline__ void load_u8(const uint8_t* row, int g, uint32_t (&v)[4])
    {
        const uint32_t w = ((const uint32_t*)row)[g];
        v[0] = w & 0xFF; v[1] = (w >> 8) & 0xFF; v[2] = (w >> 16) & 0xFF; v[3] = w >> 24;
    }
};

template <>
struct RowIO<1> {
    static __device__ __forceinline__ void load(const uint8_t* row, int g, uint32_t (&px)[1]) { px[0] = load_px_bytes(row, g); }
    static __device__ __forceinline__ void load_nt(const uint8_t* row, int g, uint32_t (&px)[1]) { px[0] = load_px_bytes(row, g); }
    static __device__ __forceinline__ void store_rgb(uint8_t* row, int g, const uint32_t (&px)[1]) { store_px_bytes(row, g, px[0]); }
    static __device__ __forceinline__ void store_mask(uint8_t* row, int g, const uint32_t (&m)[1]) { row[g] = (uint8_t)m[0]; }
    static __device__ __forceinline__ void store_z(float* row, int g, const float (&z)[1]) { row[g] = z[0]; }
    static __device__ __forceinline__ void load_u8(const uint8_t* row, int g, uint32_t (&v)[1]) { v[0] = row[g]; }
};

// FLAGS bit 0: optional depth planes, bit 1: `unused` vertices are not drawn (remove_edges),
// bit 2: edge points splatted into holes.
// `need`: bit 2 q + eye set for the edge points whose column the f32 estimate cannot decide (inside the guard band): the caller
// runs the reference's chain for those in ONE loop behind its unrolled ones (points_edge_chain) -- inlined at each of the 32
// (group, pixel, eye) sites the chain made the edge variants 9 600 instructions, 77 KB, more than the instruction cache.
template <int PX, int FLAGS>
__device__ __forceinline__ uint32_t points_splat_group(int g, const uint32_t (&dpx)[PX], const uint32_t (&cpx)[PX],
                                                       const uint32_t (&un)[PX], u64* zb, uint32_t* eb, int W,
                                                       float mult, float scale, float dl, const FrameDev& fp, bool edge_on, float guard)
{
    uint32_t need = 0;
    constexpr bool UNUSED = FLAGS & 2, EDGE = FLAGS & 4;
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        const int j = g * PX + q;
        const uint32_t code = code16_of(dpx[q]);
        const float z = decode_z(code, mult, scale);
        if (!(z > kNear)) continue;
        const float d = dl / z;
        const float fj = (float)j;
        if (!(UNUSED && un[q])) {
            const u64 key = ((u64)code << 40) | ((u64)(uint32_t)j << 24) | (u64)cpx[q];
            const int xL = point_col_row_kernel(fj + d), xR = point_col_row_kernel(fj - d);        // (the row: point_row(f32(i)) == i)
            if ((uint32_t)xL < (uint32_t)W) atomicMin(&zb[xL], key);
            if ((uint32_t)xR < (uint32_t)W) atomicMin(&zb[W + xR], key);
        } else if (EDGE && edge_on) {
            // sr:599-600, 746: the column the reference's f64 chain rounds to (mdvt_device.h "edge points")
            const uint32_t ekey = (code << 16) | (uint32_t)j;
            const int xL = edge_col_estimate(fp, 0, fj, d, W, guard), xR = edge_col_estimate(fp, 1, fj, d, W, guard);
            if (xL >= 0) atomicMin(&eb[xL], ekey);
            if (xR >= 0) atomicMin(&eb[W + xR], ekey);
            need |= (xL == -2 ? 1u : 0u) << (2 * q) | (xR == -2 ? 2u : 0u) << (2 * q);
        }
    }
    return need;
}

// the edge points points_splat_group left undecided: bit (it * PX + q) * 2 + eye of `need`, group of iteration `it` = g0 + it * gstep
template <int PX, int NIT>
__device__ __forceinline__ void points_edge_chain(uint32_t need, int g0, int gstep, const uint32_t (&dpx)[NIT][PX], uint32_t* eb, int W,
                                                  float mult, float scale, const FrameDev& fp)
{
#pragma unroll 1
    while (need) {
        const int b = __ffs((int)need) - 1, eye = b & 1, q = (b >> 1) % PX, it = (b >> 1) / PX;
        need &= need - 1u;
        uint32_t px = dpx[0][0];
#pragma unroll
        for (int k = 0; k < NIT; ++k)
#pragma unroll
            for (int r = 0; r < PX; ++r)
                if (k == it && r == q) px = dpx[k][r];
        const int j = (g0 + it * gstep) * PX + q;
        const uint32_t code = code16_of(px);
        const int x = edge_col_chain(fp, eye, (float)j, decode_z(code, mult, scale), W);
        if (x >= 0 && x < W) atomicMin(&eb[eye * W + x], (code << 16) | (uint32_t)j);
    }
}

// TPB threads; ITERS > 0: every thread owns exactly ITERS groups (ngroups <= TPB*ITERS) whose HBM
// loads are all issued before the LDS clear, so 2*ITERS 768-byte wave loads are in flight per wave
// while the z-buffer is initialised.  ITERS == 0: plain strided loop (any W).
template <int PX, int FLAGS, int TPB, int ITERS>
__global__ void __launch_bounds__(TPB, ((FLAGS & 5) == 4) ? 6 : 1) k_points_rows(RenderArgs a)
{
    constexpr bool ZOUT = FLAGS & 1, UNUSED = FLAGS & 2, EDGE = FLAGS & 4, SEED = FLAGS & 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W;
    u64* zb = (u64*)smem;                        // [2][W] main z keys, left then right eye
    uint32_t* eb = (uint32_t*)(zb + 2 * (size_t)W);   // [2][W] edge-point keys (code16<<16 | j), EDGE only

    const int fr = blockIdx.x / a.H;
    const int i = blockIdx.x - fr * a.H;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const float mult = fp.mult, scale = fp.scale, dl = fp.dl;
    const int tid = threadIdx.x;
    const int ngroups = W / PX;

    const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch;
    const uint8_t* crow = a.color + (size_t)f * a.color_stride + (size_t)i * a.color_pitch;
    const uint8_t* urow = UNUSED ? a.unused + (size_t)fr * a.ws_stride_px + (size_t)i * W : nullptr;
    // (the edge points of scanlines erow_lo .. erow_hi are k_edge_rows_exact's: their row is not the source row)
    const bool edge_on = EDGE && !edge_row_deferred(fp, i);
    const float guard = edge_col_guard(W);

    constexpr int NIT = ITERS > 0 ? ITERS : 1;
    uint32_t dpx[NIT][PX], cpx[NIT][PX], un[NIT][PX];
    if (ITERS > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = tid + it * TPB;
            if (g < ngroups) {
                RowIO<PX>::load_nt(drow, g, dpx[it]);
                RowIO<PX>::load_nt(crow, g, cpx[it]);
                if (UNUSED) RowIO<PX>::load_u8(urow, g, un[it]);
            }
        }
    }

    // clear the z-buffers (16 B per store where the layout allows)
    {
        uint4* z4 = (uint4*)zb;
        const int n4 = W;                        // 2*W u64 = W uint4
        for (int x = tid; x < n4; x += TPB) z4[x] = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (EDGE) for (int x = tid; x < 2 * W; x += TPB) eb[x] = kEmpty32;
    }
    __syncthreads();

    if (ITERS > 0) {
        uint32_t need = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = tid + it * TPB;
            if (g < ngroups)
                need |= points_splat_group<PX, FLAGS>(g, dpx[it], cpx[it], un[it], zb, eb, W, mult, scale, dl, fp, edge_on, guard) << (2 * PX * it);
        }
        if (EDGE) points_edge_chain<PX, NIT>(need, tid, TPB, dpx, eb, W, mult, scale, fp);
    } else {
        for (int g = tid; g < ngroups; g += TPB) {
            RowIO<PX>::load_nt(drow, g, dpx[0]);
            RowIO<PX>::load_nt(crow, g, cpx[0]);
            if (UNUSED) RowIO<PX>::load_u8(urow, g, un[0]);
            const uint32_t need = points_splat_group<PX, FLAGS>(g, dpx[0], cpx[0], un[0], zb, eb, W, mult, scale, dl, fp, edge_on, guard);
            if (EDGE) points_edge_chain<PX, NIT>(need, g, 0, dpx, eb, W, mult, scale, fp);
        }
    }
    __syncthreads();

#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
        uint8_t* orow = a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)i * a.rgb_pitch;
        uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)i * a.mask_pitch;
        float* zrow = ZOUT && a.zout[eye]
                          ? (float*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)i * a.zout_pitch)
                          : nullptr;
        const u64* zrow_lds = zb + (size_t)eye * W;
        for (int g = tid; g < ngroups; g += TPB) {
            uint32_t opx[PX], om[PX], spx[PX], eseed = 0;
            float oz[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const int x = g * PX + q;
                const u64 key = zrow_lds[x];
                const uint32_t rgb = (uint32_t)key & 0xFFFFFFu;
                const bool covered = key != kEmpty64;
                const bool hole = !covered || rgb == a.key_rgb;       // sr:740 colour-key compare
                uint32_t out = hole ? 0u : rgb;                       // sr:793
                if (EDGE && hole && edge_on) {
                    const uint32_t ek = eb[(size_t)eye * W + x];
                    if (ek != kEmpty32 && a.edge_paint) {
                        // colour of source column (ek & 0xFFFF) of this row (sr:813-814)
                        out = load_px_bytes(crow, (int)(ek & 0xFFFFu));
                    }
                }
                opx[q] = out;
                om[q] = hole ? 255u : 0u;
                if (ZOUT) oz[q] = covered ? decode_z((uint32_t)(key >> 40), mult, scale) : 0.0f;
                if (SEED && a.seed[eye]) {
                    spx[q] = seed_pixel(a, fp, f, eye, x, i, hole, ~0u, 0);
                    if (EDGE && hole && edge_on) { const uint32_t ek = eb[(size_t)eye * W + x]; if (ek != kEmpty32) eseed |= 1u << q; }
                }
            }
            if (SEED && EDGE) {
                // the pixels whose seed colour is an edge point's normal (f64, a few hundred instructions): one copy of that code
#pragma unroll 1
                while (eseed) {
                    const int q = __ffs((int)eseed) - 1;
                    eseed &= eseed - 1u;
                    const uint32_t c = edge_normal_colour(a, fp, f, eye, i, (int)(eb[(size_t)eye * W + g * PX + q] & 0xFFFFu), 0);
#pragma unroll
                    for (int k = 0; k < PX; ++k) if (k == q) spx[k] = c;
                }
            }
            RowIO<PX>::store_rgb(orow, g, opx);
            RowIO<PX>::store_mask(mrow, g, om);
            if (ZOUT && zrow) RowIO<PX>::store_z(zrow, g, oz);
            if (SEED && a.seed[eye]) RowIO<PX>::store_rgb(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)i * a.seed_pitch, g, spx);
        }
    }
}

// Wavefront hole-mask compaction.  Each lane holds the hole flags of 4 consecutive pixels (nib, LSB = lowest
// x; lane & 7 == g & 7).  Eight lanes' nibbles are OR-combined into one dword of the packed 1-bit/px mask with three DPP
// operands on the VALU (two quad permutes and a row shift: no LDS round trip -- __shfl_xor compiles to ds_bpermute_b32 here,
// three of them per eye behind the row's 8-byte LDS atomics were most of what the fused variant cost in r05); the hole count
// of the wave is the popcount of four 64-lane ballots (scalar unit).  The dword is valid in the lanes with g & 7 == 0.
template <int CTRL>
__device__ __forceinline__ uint32_t or_dpp(uint32_t v)
{
    return v | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t mask_dword_of_8_lanes(uint32_t nib, int g)
{
    uint32_t v = nib << (4 * (g & 7));
    v = or_dpp<0xB1>(v);          // quad_perm [1,0,3,2]: lane ^ 1
    v = or_dpp<0x4E>(v);          // quad_perm [2,3,0,1]: lane ^ 2
    return or_dpp<0x104>(v);      // row_shl:4: lane i takes lane i + 4 (of its row of 16)
}

__device__ __forceinline__ int wave_hole_count(uint32_t nib)
{
    return __popcll(__ballot(nib & 1)) + __popcll(__ballot(nib & 2)) + __popcll(__ballot(nib & 4)) + __popcll(__ballot(nib & 8));
}

__device__ __forceinline__ int compact_hole_nibble(uint32_t nib, int g, bool act, uint8_t* bits_row, bool want_count)
{
    if (bits_row) {
        const uint32_t v = mask_dword_of_8_lanes(nib, g);
        if (act && (g & 7) == 0) ((uint32_t*)bits_row)[g >> 3] = v;
    }
    return want_count ? wave_hole_count(nib) : 0;
}

// The fused points kernel's hole counts: every WAVE stores its two counts as one dword (left | right << 16: at most 256 each) to
// wave_counts[(frame * H + row) * waves + wave] -- a plain fire-and-forget store, no LDS total, no barrier, no atomic -- and
// k_reduce_wave_counts adds them up per frame.  r06 measured the alternatives on 128 frames of 1080p (595 us without counts):
// LDS totals + barrier + row store + reduce launch 664; LDS totals without a barrier + one RETURNING atomic per workgroup into
// per-frame accumulators that the last arrival finishes (no second launch) 664 -> 639 without the atomics; fire-and-forget atomics per
// wave with the frame's last workgroup waiting for the arrivals: dropped -- its wait has to read with read-modify-write atomics (a
// load may be served from the XCD's own L2, which is not coherent with the other XCDs' atomics within a kernel: the first version hung
// at 1080p on stale values) and with 128 frames in flight the waiting workgroups' polling stalled the launch.
// wave_counts[(frame * H + row) * waves + wave], waves = threads per workgroup / 64: a frame's counts are H * waves consecutive dwords
__global__ void __launch_bounds__(256) k_reduce_wave_counts(const uint32_t* __restrict__ wave_counts, uint32_t* __restrict__ hole_counts,
                                                            int H, int waves, int frame0)
{
    __shared__ uint32_t part[2][4];
    const int fr = blockIdx.x;
    const int n = H * waves;                                       // (a multiple of 4: waves is 4, 8 or 16)
    const uint4* src = reinterpret_cast<const uint4*>(wave_counts + (size_t)fr * n);
    uint32_t l = 0, r = 0;
    for (int k = threadIdx.x; k < n / 4; k += 256) {
        const uint4 v = src[k];
        l += (v.x & 0xFFFFu) + (v.y & 0xFFFFu) + (v.z & 0xFFFFu) + (v.w & 0xFFFFu);
        r += (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
    }
    for (int off = 32; off > 0; off >>= 1) { l += __shfl_down((int)l, off); r += __shfl_down((int)r, off); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = l; part[1][threadIdx.x >> 6] = r; }
    __syncthreads();
    if (threadIdx.x < 2) hole_counts[2 * (size_t)(frame0 + fr) + threadIdx.x] = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
}

// row_counts[(2*frame + eye)*H + row] -> hole_counts[2*frame + eye]
__global__ void __launch_bounds__(256) k_reduce_counts(const uint32_t* __restrict__ row_counts, uint32_t* __restrict__ hole_counts,
                                                       int H, int frame0)
{
    __shared__ uint32_t part[4];
    const int fe = blockIdx.x;                       // 2*fr + eye within this launch
    uint32_t c = 0;
    for (int i = threadIdx.x; i < H; i += 256) c += row_counts[(size_t)fe * H + i];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down((int)c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) hole_counts[2 * (size_t)frame0 + fe] = part[0] + part[1] + part[2] + part[3];
}

// Post-pass for the kernels that do not compact in place: byte mask -> packed bits + per-row counts.
// One workgroup per (row, frame, eye).
__global__ void __launch_bounds__(256) k_pack_mask(RenderArgs a)
{
    __shared__ uint32_t wcount[4];
    const int W = a.W;
    const int ngroups = (W + 3) / 4;
    const int i = blockIdx.x;
    const int fr = blockIdx.y >> 1, eye = blockIdx.y & 1;
    const int f = a.frame0 + fr;
    const uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)i * a.mask_pitch;
    uint8_t* brow = a.maskbits[eye] ? a.maskbits[eye] + (size_t)f * a.maskbits_stride + (size_t)i * a.maskbits_pitch : nullptr;
    int cnt = 0;
    for (int g0 = 0; g0 < ngroups; g0 += 256) {
        const int g = g0 + threadIdx.x;
        const bool act = g < ngroups;
        uint32_t nib = 0;
        if (act)
            for (int q = 0; q < 4; ++q) if (4 * g + q < W && mrow[4 * g + q]) nib |= 1u << q;
        cnt += compact_hole_nibble(nib, g, act, brow, a.hole_counts != nullptr);
    }
    if (a.hole_counts) {
        if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = (uint32_t)cnt;
        __syncthreads();
        if (threadIdx.x == 0) a.row_counts[(2 * (size_t)fr + eye) * a.H + i] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    }
}

hipError_t launch_pack_mask(const RenderArgs& a, int n, hipStream_t s)
{
    dim3 grid(a.H, n * 2);
    hipLaunchKernelGGL(k_pack_mask, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_reduce_counts(const RenderArgs& a, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_reduce_counts, dim3(2 * n), dim3(256), 0, s, a.row_counts, a.hole_counts, a.H, a.frame0);
    return hipGetLastError();
}

// The headline kernel's division, proven per parameter set.  d = dl / z is IEEE-correctly rounded by decree, and hipcc's expansion of it
// is 10 VALU instructions of the ~220 a lane spends on its four pixels -- with the kernel at ~80 % of the VALU issue rate next to
// ~78 % of HBM peak (r06: every instruction added to it shows up in its time).  But a frame's z has only 65535 values:
// z = (f32(code << 16) * mult) * scale.  points_div_short is v_rcp_f32, one multiply and two fma -- Markstein's correction step from the
// raw reciprocal, faithful in general and correctly rounded unless the quotient sits within ~2^-23 ulp of a rounding boundary --
// and k_divcheck compares it with the IEEE division for every code of one (mult, scale, dl): a counter of 0 is a proof for that frame's
// operands, anything else leaves the frame on the expansion.  (tools/probe/div_probe.hip: 0 mismatches in 336 parameter sets.)
__device__ __forceinline__ float points_div_short(float dl, float z)
{
    const float y = __builtin_amdgcn_rcpf(z);
    const float q = dl * y;
    return __builtin_fmaf(__builtin_fmaf(-z, q, dl), y, q);
}

__global__ void __launch_bounds__(256) k_divcheck(float mult, float scale, float dl, uint32_t* __restrict__ bad)
{
    const uint32_t code = blockIdx.x * 256u + threadIdx.x;           // 0 .. 65535
    const float z = ((float)(code << 16) * mult) * scale;
    bool differs = false;
    if (z > kNear) differs = __float_as_uint(points_div_short(dl, z)) != __float_as_uint(dl / z);
    if (__ballot(differs) && (threadIdx.x & 63) == 0) atomicAdd(bad, (uint32_t)__popcll(__ballot(differs)));
}

hipError_t launch_divcheck(float mult, float scale, float dl, uint32_t* bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_divcheck, dim3(256), dim3(256), 0, s, mult, scale, dl, bad);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Hand-scheduled variant of the row kernel for the headline case (4 px/lane, one group per thread, no
// edge filter): byte shuffles are single v_perm_b32 ops, out-of-range fragments are steered to a trash
// LDS word instead of branching (no exec-mask traffic).  Same arithmetic, same results as k_points_rows.
// -------------------------------------------------------------------------------------------------
// BITS: 0 = byte mask only (the headline); 1 = packed mask and / or hole counts, whichever of the optional buffers are there (run-time
// tests; the wave counts its holes by ballots); 8 | mask | 2 pack | 4 count = the same with the set of outputs fixed at compile time
// (r06: the generic form ran the CU's scalar unit at ~70 % -- null tests, branches and exec masks around the optional stores: 267 SALU
// instructions per wave against the headline's 161; counting from the packed mask in a kernel afterwards instead of by ballots here
// was measured too: 37 us per 128 frames against 6).
template <int TPB, bool ZOUT, int BITS, int NT = 3>
__global__ void __launch_bounds__(TPB) k_points_rows_fast(RenderArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W, W4 = W >> 2;
    u64* zb = (u64*)smem;                        // [2W] keys (left, right) + [1] trash word
    const int fr = blockIdx.x / a.H;
    const int i = blockIdx.x - fr * a.H;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const float mult = fp.mult, scale = fp.scale, dl = fp.dl;
    const int g = threadIdx.x;
    const bool act = g < W4;

    uint32_t d0 = 0, d1 = 0, d2 = 0, c0 = 0, c1 = 0, c2 = 0;
    if (act) {
        const uint32_t* dp = (const uint32_t*)(a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch) + 3 * g;
        const uint32_t* cp = (const uint32_t*)(a.color + (size_t)f * a.color_stride + (size_t)i * a.color_pitch) + 3 * g;
        if (NT & 1) {                              // every input byte is read exactly once: non-temporal (161 -> 148 us per 32 frames with both)
            d0 = __builtin_nontemporal_load(dp); d1 = __builtin_nontemporal_load(dp + 1); d2 = __builtin_nontemporal_load(dp + 2);
            c0 = __builtin_nontemporal_load(cp); c1 = __builtin_nontemporal_load(cp + 1); c2 = __builtin_nontemporal_load(cp + 2);
        } else {
            d0 = dp[0]; d1 = dp[1]; d2 = dp[2];
            c0 = cp[0]; c1 = cp[1]; c2 = cp[2];
        }
    }
    {
        uint4* z4 = (uint4*)zb;
        for (int x = g; x < W; x += TPB) z4[x] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    __syncthreads();

    if (act) {
        // R<<24 | B<<16 of each pixel (dfh:67-69): one byte permute each
        uint32_t c32[4], cpx[4];
        c32[0] = __builtin_amdgcn_perm(d0, d0, 0x00020c0cu);
        c32[1] = __builtin_amdgcn_perm(d1, d0, 0x03050c0cu);
        c32[2] = __builtin_amdgcn_perm(d2, d1, 0x02040c0cu);
        c32[3] = __builtin_amdgcn_perm(d2, d2, 0x01030c0cu);
        cpx[0] = c0 & 0xFFFFFFu;
        cpx[1] = __builtin_amdgcn_perm(c1, c0, 0x0c050403u);
        cpx[2] = __builtin_amdgcn_perm(c2, c1, 0x0c040302u);
        cpx[3] = c2 >> 8;
        const float fj0 = (float)(g << 2);
        const uint32_t jhi = (uint32_t)g >> 6, jlo = ((uint32_t)g << 2) & 0xFFu;
        const int trash = 2 * W;
        auto splat = [&](auto short_div) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float z = ((float)c32[q] * mult) * scale;
                const bool ok = z > kNear;
                const float d = decltype(short_div)::value ? points_div_short(dl, z) : dl / z;
                const float fj = fj0 + (float)q;
                const int xL = point_col_row_kernel(fj + d), xR = point_col_row_kernel(fj - d);
                const bool okL = ok && (uint32_t)xL < (uint32_t)W;
                const bool okR = ok && (uint32_t)xR < (uint32_t)W;
                const int sL = okL ? xL : trash;
                const int sR = okR ? W + xR : trash;
                const uint32_t hi = __builtin_amdgcn_perm(c32[q], jhi, 0x0c070600u);      // code16 << 8 | j >> 8
                const uint32_t lo = __builtin_amdgcn_perm(jlo + (uint32_t)q, cpx[q], 0x04020100u);   // (j & 255) << 24 | rgb
                const u64 key = ((u64)hi << 32) | lo;
                atomicMin(&zb[sL], key);
                atomicMin(&zb[sR], key);
            }
        };
        // (wave-uniform: the frame's parameter set has been proven, or it has not)
        if (fp.div_slot >= 0 && a.divcheck[fp.div_slot] == 0u && !(MDVT_DEBUG_SKIP(a) & 128)) splat(std::true_type{});      // (128: tuning build's A/B)
        else splat(std::false_type{});
    }
    __syncthreads();

    constexpr bool kSpec = (BITS & 8) != 0;
    const bool counting = kSpec ? (BITS & 4) != 0 : (BITS && a.hole_counts && !(MDVT_DEBUG_SKIP(a) & 32));
    uint32_t cnt[2] = {0u, 0u};
    if (act || BITS) {
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            uint4 k01 = make_uint4(0, 0, 0, 0), k23 = make_uint4(0, 0, 0, 0);
            if (act) {
                const uint4* zq = (const uint4*)(zb + (size_t)eye * W) + 2 * g;
                k01 = zq[0]; k23 = zq[1];
            }
            const uint32_t hi[4] = {k01.y, k01.w, k23.y, k23.w};
            const uint32_t lo[4] = {k01.x, k01.z, k23.x, k23.z};
            uint32_t o[4], mw = 0, nib = 0;
            float oz[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool covered = hi[q] != ~0u;                 // code16<<8 | j>>8 < 2^24 when covered
                const uint32_t rgb = lo[q] & 0xFFFFFFu;
                const bool hole = !covered || rgb == a.key_rgb;    // sr:740
                o[q] = hole ? 0u : rgb;                            // sr:793
                if (BITS) {
                    nib |= hole ? (1u << q) : 0u;
                    // the compare's lane mask IS the ballot: the wave's count costs scalar instructions only
                    if (counting && !(MDVT_DEBUG_SKIP(a) & 64)) cnt[eye] += (uint32_t)__popcll(__ballot(hole && act));
                } else {
                    mw |= hole ? (0xFFu << (8 * q)) : 0u;
                }
                if (ZOUT) oz[q] = covered ? decode_z(hi[q] >> 8, mult, scale) : 0.0f;
            }
            uint8_t* mbase = a.mask[eye];                          // (NULL with BITS: the caller takes the packed mask only)
            const bool has_mask = kSpec ? (BITS & 1) != 0 : (!BITS || mbase != nullptr);
            const bool has_pack = kSpec ? (BITS & 2) != 0 : (BITS && a.maskbits[eye] != nullptr && !(MDVT_DEBUG_SKIP(a) & 64));
            if (BITS && has_mask) {                                // nibble -> four bytes of 0 / 255
                const uint32_t b = __umul24(nib, 0x204081u) & 0x01010101u;
                mw = (b << 8) - b;
            }
            if (act) {
                uint32_t* op = (uint32_t*)(a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)i * a.rgb_pitch) + 3 * g;
                uint32_t* mp = (uint32_t*)(mbase + (size_t)f * a.mask_stride + (size_t)i * a.mask_pitch) + g;
                if (NT & 2) {                          // ... and every output byte written once
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(o[1], o[0], 0x04020100u), op);
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(o[2], o[1], 0x05040201u), op + 1);
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(o[3], o[2], 0x06050402u), op + 2);
                    if (has_mask) __builtin_nontemporal_store(mw, mp);
                } else {
                op[0] = __builtin_amdgcn_perm(o[1], o[0], 0x04020100u);
                op[1] = __builtin_amdgcn_perm(o[2], o[1], 0x05040201u);
                op[2] = __builtin_amdgcn_perm(o[3], o[2], 0x06050402u);
                if (has_mask) *mp = mw;
                }
                if (ZOUT && a.zout[eye]) {
                    float4* zp = (float4*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)i * a.zout_pitch) + g;
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 v = {oz[0], oz[1], oz[2], oz[3]};
                    if (NT & 2) __builtin_nontemporal_store(v, (f32x4*)zp);
                    else *zp = make_float4(oz[0], oz[1], oz[2], oz[3]);
                }
            }
            if (BITS && has_pack) {
                if (!act) nib = 0u;
                const uint32_t v = mask_dword_of_8_lanes(nib, g);
                if (act && (g & 7) == 0 && !(MDVT_DEBUG_SKIP(a) & 8))
                    ((uint32_t*)(a.maskbits[eye] + (size_t)f * a.maskbits_stride + (size_t)i * a.maskbits_pitch))[g >> 3] = v;
            }
        }
    }
    if (counting && (g & 63) == 0)
        a.wave_counts[((size_t)fr * a.H + i) * (TPB / 64) + (g >> 6)] = cnt[0] | (cnt[1] << 16);
}

// =================================================================================================
// POINT MODE, general (pose / convergence / K != Krender): global 64-bit z keys
// =================================================================================================
//   key = f32 bits of Z' << 32 | i << 16 | j      (Z' > 0, so the bit pattern orders like the value)

// An edge point's key into the general paths' edge-key plane.  Whoever finds the word EMPTY also lists it (segment of the
// point's source row: at most one word per vertex and eye, so 2 W entries hold them all): the resolve pass then reads the
// edge keys only where the render left a hole, and k_edge_keys_reset empties exactly the written words -- instead of the
// resolve rewriting the whole plane (8 B/px per eye) for the ~3 % of the words that were touched.
// The lanes of a wave that reach this together (`on`: this lane has a key to post) list their first writes with ONE atomic per
// source row among them (r05): neighbouring lanes work on neighbouring vertices of one source row, and a returning atomic per lane on
// that row's counter -- ~60 of them per row from all over the chip -- was most of the splat's time (96 us per 8 frames for 428
// instructions a vertex).
__device__ __forceinline__ void post_edge_key(const RenderArgs& a, int eye, int fr, int src_row, bool on, size_t pix, u64 key)
{
    u64 old = 0ull;
    if (on) old = atomicMin(&a.ekeys[eye][(size_t)fr * a.ws_stride_px + pix], key);
    const bool first = on && old == kEmpty64;
    const int lane = (int)(threadIdx.x & 63u);
    u64 m = __ballot(first);
    while (m) {                                                     // (uniform over the lanes that are here)
        const int l = __ffsll((long long)m) - 1;
        const int row = __builtin_amdgcn_readlane(src_row, l);
        const bool mine = first && src_row == row;
        const u64 same = __ballot(mine);
        const size_t seg = (size_t)fr * a.H + row;
        uint32_t base = 0;
        if (lane == l) base = atomicAdd(&a.elist_count[seg], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
        if (mine) a.elist[seg * (size_t)(2 * a.W) + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = ((uint32_t)eye << 31) | (uint32_t)pix;
        m &= ~same;
    }
}

__global__ void __launch_bounds__(64) k_edge_keys_reset(RenderArgs a)
{
    const int fr = blockIdx.y;
    const size_t seg = (size_t)fr * a.H + blockIdx.x;
    const uint32_t n = a.elist_count[seg];
    if (n == 0u) return;
    const uint32_t* list = a.elist + seg * (size_t)(2 * a.W);
    for (uint32_t k = threadIdx.x; k < n; k += 64) {
        const uint32_t e = list[k];
        a.ekeys[e >> 31][(size_t)fr * a.ws_stride_px + (e & 0x7FFFFFFFu)] = kEmpty64;
    }
    if (threadIdx.x == 0) a.elist_count[seg] = 0u;           // (every lane of the one wave has read n)
}

template <int FLAGS>
__global__ void __launch_bounds__(256) k_points_splat_general(RenderArgs a)
{
    constexpr bool UNUSED = FLAGS & 2, EDGE = FLAGS & 4, EDGE_ONLY = FLAGS & 8;
    const int W = a.W, H = a.H;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int fr = blockIdx.z;
    if (j >= W) return;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const bool un = UNUSED && a.unused[(size_t)fr * a.ws_stride_px + (size_t)i * W + j];
    if (EDGE_ONLY && !un) return;         // mesh mode: only the ~3 % removed vertices are splatted
    const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch;
    const uint32_t code = code16_of(load_px_bytes(drow, j));
    const float z = decode_z(code, fp.mult, fp.scale);
    if (!(z > kNear)) return;
    const float gx = (float)j * fp.sx, gy = (float)i * fp.sy;
    float xc, yc;
    camera_point(fp, gx, gy, z, xc, yc);
    const uint32_t src = ((uint32_t)i << 16) | (uint32_t)j;
    if (!un) {
        if (EDGE_ONLY) return;
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            const Vert v = vertex_for_eye(fp, eye, gx, gy, z, xc, yc);
            if (!v.ok) continue;
            const int px = point_col(v.u), py = point_row(v.v);
            if (!((uint32_t)px < (uint32_t)W && (uint32_t)py < (uint32_t)H)) continue;
            zkey_post<false>(&a.keys[eye][(size_t)fr * a.ws_stride_px + (size_t)py * W + px], (a.key_parity >> fr) & 1u, __float_as_uint(v.z), src);   // v.z > 0
        }
    } else if (EDGE) {
        EdgePx ep;
        edge_point_pixels(fp, W, H, i, j, fp.sx != 1.0f, z, ep);          // the reference's f64 chain (mdvt_device.h)
#pragma unroll
        for (int eye = 0; eye < 2; ++eye)
            post_edge_key(a, eye, fr, i, ep.ok[eye], (size_t)ep.y[eye] * W + ep.x[eye], ((u64)ep.zkey[eye] << 32) | src);
    }
}

// Mesh mode only splats the removed vertices (~3 % of the grid): the reference's f64 chain, ~10^3 instructions a vertex.  Two launches
// (r05): k_edge_vertices_list turns the 1 B/px plane of `unused` flags into a compact per-frame list of vertices (source row << 16 |
// column; one atomic per workgroup of 16 K pixels), k_edge_points_splat_list runs the chain over the list with every lane busy.
// Until r05 one kernel did both, a workgroup per 8 source rows compacting into LDS: 135 workgroups per 1080p frame, each a chain of
// 15 load / append / barrier steps and two rounds of the chain -- 11.6 us per frame for 3.8 MB, latency all of it.  W % 4 == 0.
// (`valid`: this lane has a vertex; every lane of the wave comes along for post_edge_key's sake)
__device__ __forceinline__ void edge_point_splat_one(const RenderArgs& a, const FrameDev& fp, const uint8_t* dbase, int fr, uint32_t e, bool valid)
{
    const int i = (int)(e >> 16), j = (int)(e & 0xFFFFu);
    EdgePx ep;
    ep.ok[0] = ep.ok[1] = false;
    ep.x[0] = ep.x[1] = ep.y[0] = ep.y[1] = 0;
    ep.zkey[0] = ep.zkey[1] = 0u;
    if (valid) {
        const float z = decode_z(code16_of(load_px_bytes(dbase + (size_t)i * a.depth_pitch, j)), fp.mult, fp.scale);
        if (z > kNear) edge_point_pixels(fp, a.W, a.H, i, j, 1, z, ep);  // the reference's f64 chain (mdvt_device.h)
    }
#pragma unroll
    for (int eye = 0; eye < 2; ++eye)
        post_edge_key(a, eye, fr, i, ep.ok[eye], (size_t)ep.y[eye] * a.W + ep.x[eye], ((u64)ep.zkey[eye] << 32) | e);
}
constexpr int kListTPB = 1024, kListDwords = 4;          // a workgroup of the list pass: 1024 threads x 4 dwords x 4 flags
__global__ void __launch_bounds__(kListTPB) k_edge_vertices_list(RenderArgs a)
{
    __shared__ uint32_t wsum[kListTPB / 64];
    __shared__ uint32_t wbase;
    const int W = a.W, fr = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ndw = (uint32_t)W * (uint32_t)a.H / 4u;              // (W % 4 == 0: the plane is whole dwords, rows contiguous)
    const uint32_t* flags = (const uint32_t*)(a.unused + (size_t)fr * a.ws_stride_px);
    uint32_t fl[kListDwords], cnt = 0;
#pragma unroll
    for (int k = 0; k < kListDwords; ++k) {
        const uint32_t d = (blockIdx.x * kListDwords + k) * kListTPB + tid;
        fl[k] = d < ndw ? flags[d] : 0u;
        cnt += ((fl[k] & 0xFFu) != 0u) + ((fl[k] & 0xFF00u) != 0u) + ((fl[k] & 0xFF0000u) != 0u) + ((fl[k] >> 24) != 0u);
    }
    uint32_t inc = cnt;                                                   // inclusive sums along the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < kListTPB / 64; ++k) { const uint32_t v = wsum[k]; wsum[k] = run; run += v; }
        wbase = run ? atomicAdd(&a.vlist_count[fr], run) : 0u;
    }
    __syncthreads();
    if (!cnt) return;
    uint32_t* out = a.vlist + (size_t)fr * a.ws_stride_px + wbase + wsum[wave] + (inc - cnt);
#pragma unroll
    for (int k = 0; k < kListDwords; ++k) {
        if (!fl[k]) continue;
        const uint32_t o = ((blockIdx.x * kListDwords + k) * kListTPB + tid) * 4u;
        const uint32_t i = o / (uint32_t)W, j = o - i * (uint32_t)W;      // (the dword's four pixels share a row)
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((fl[k] >> (8 * q)) & 0xFFu) *out++ = (i << 16) | (j + (uint32_t)q);
    }
}
__global__ void __launch_bounds__(256) k_edge_points_splat_list(RenderArgs a)
{
    const int fr = blockIdx.y, f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
    const uint32_t total = a.vlist_count[fr];
    const uint32_t* list = a.vlist + (size_t)fr * a.ws_stride_px;
    for (uint32_t k0 = blockIdx.x * 256u; k0 < total; k0 += gridDim.x * 256u) {          // (workgroup uniform)
        const uint32_t k = k0 + threadIdx.x;
        edge_point_splat_one(a, fp, dbase, fr, k < total ? list[k] : 0u, k < total);
    }
}

// The one-kernel form (until r05 the only one; kept for launches under a POSE, where the list form measures 4.6 % slower -- 1080p x 32
// with a pose and edge removal 13.0 -> 12.4 k frames/s, C4 mesh with edge removal 3.06 -> 2.92 k -- while convergence-only launches
// gain from it: product default 12.5 -> 12.9 k, a single frame 163 -> 143 us).  A workgroup scans kSplatRows source rows of the flag
// plane, compacts the flagged vertices of all of them into one list in LDS and runs the chain over it 256 vertices at a time.
constexpr int kSplatRows = 8;
__global__ void __launch_bounds__(256) k_edge_points_splat4(RenderArgs a)
{
    const int W = a.W, H = a.H;
    const int fr = blockIdx.y, tid = threadIdx.x;
    const int i0 = blockIdx.x * kSplatRows, rows = min(kSplatRows, H - i0);
    __shared__ uint32_t list[1024 + 256];             // source row << 16 | column: a step adds at most 1024, fewer than 256 stay behind
    __shared__ uint32_t cnt;
    if (tid == 0) cnt = 0u;
    __syncthreads();
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
    const uint8_t* ubase = a.unused + (size_t)fr * a.ws_stride_px + (size_t)i0 * W;
    const int per_row = W / 4, total = rows * per_row;
    for (int g0 = 0; g0 < total; g0 += 256) {         // (workgroup uniform)
        const int g = g0 + tid;
        uint32_t flags = 0;
        if (g < total) flags = *(const uint32_t*)(ubase + (size_t)g * 4);          // (rows are contiguous: W % 4 == 0)
        if (flags) {
            const uint32_t b0 = (flags & 0xFFu) != 0u, b1 = (flags & 0xFF00u) != 0u, b2 = (flags & 0xFF0000u) != 0u, b3 = (flags >> 24) != 0u;
            uint32_t pos = atomicAdd(&cnt, b0 + b1 + b2 + b3);
            const uint32_t r = (uint32_t)(g / per_row), col = (uint32_t)(g - (int)r * per_row) * 4u;
            const uint32_t base = (((uint32_t)i0 + r) << 16) | col;
            if (b0) list[pos++] = base;
            if (b1) list[pos++] = base + 1u;
            if (b2) list[pos++] = base + 2u;
            if (b3) list[pos++] = base + 3u;
        }
        __syncthreads();
        uint32_t n = cnt;
        while (n >= 256u) {
            n -= 256u;
            edge_point_splat_one(a, fp, dbase, fr, list[n + tid], true);
        }
        __syncthreads();
        if (tid == 0) cnt = n;
        __syncthreads();
    }
    if (cnt) edge_point_splat_one(a, fp, dbase, fr, (uint32_t)tid < cnt ? list[tid] : 0u, (uint32_t)tid < cnt);      // (uniform: cnt is the workgroup's)
}

// mdvt_edge_point_pixels: the pixel the edge point of EVERY vertex of one frame lands on, both eyes (INT32_MIN twice: outside
// the frame, or depth code 0).  how = 0: the chain, as the global-key kernels and k_edge_rows_exact take it; how = 1: as the
// LDS row kernels of a pure-shift frame take it -- column from the f32 estimate unless it lies within the guard of a tie
// (edge_col_pure), row = the source row, the scanlines of edge_row_deferred by the chain.
__global__ void __launch_bounds__(256) k_edge_point_pixels(const uint8_t* depth, size_t pitch, const FrameDev* fpp, int W, int H,
                                                           int of_by_one, int how, int32_t* out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= W) return;
    const FrameDev& fp = fpp[0];
    const float z = decode_z(code16_of(load_px_bytes(depth + (size_t)i * pitch, j)), fp.mult, fp.scale);
    int32_t* o = out + 4 * ((size_t)i * W + j);
    EdgePx ep;
    ep.ok[0] = ep.ok[1] = false;
    if (how == 0 || edge_row_deferred(fp, i)) edge_point_pixels(fp, W, H, i, j, of_by_one, z, ep);
    else if (z > kNear) {
        const float guard = edge_col_guard(W), gx = (float)j * fp.sx, d = fp.dl / z;
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            ep.x[eye] = edge_col_pure(fp, eye, gx, z, d, W, guard);
            ep.y[eye] = i;
            ep.ok[eye] = ep.x[eye] >= 0;
        }
    }
#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
        o[2 * eye] = ep.ok[eye] ? ep.x[eye] : INT32_MIN;
        o[2 * eye + 1] = ep.ok[eye] ? ep.y[eye] : INT32_MIN;
    }
}

hipError_t launch_edge_point_pixels(const uint8_t* depth, size_t pitch, const FrameDev* fp, int W, int H, int of_by_one, int how,
                                    int32_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_edge_point_pixels, dim3((W + 255) / 256, H), dim3(256), 0, s, depth, pitch, fp, W, H, of_by_one, how, out);
    return hipGetLastError();
}

hipError_t launch_edge_keys_reset(const RenderArgs& a, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_edge_keys_reset, dim3(a.H, n), dim3(64), 0, s, a);
    return hipGetLastError();
}

// Resolve of both general paths: PX pixels per thread, coalesced key reads; the z keys need no clearing pass (parity
// scheme, mdvt_device.h: only the uncovered words are rewritten; the edge keys: k_edge_keys_reset), colour from the key word (mesh: colour
// keys) or gathered from the source frame by the winning source index (points).
template <int PX, int FLAGS, bool MESH>
__global__ void __launch_bounds__(256) k_resolve_general(RenderArgs a)
{
    constexpr bool ZOUT = FLAGS & 1, EDGE = FLAGS & 4, SEED = FLAGS & 8;
    const int W = a.W;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int fr = blockIdx.z >> 1, eye = blockIdx.z & 1;
    if (g >= W / PX) return;
    if (MDVT_DEBUG_SKIP(a) & 512) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (r05 diagnosis, tuning build)
    const int f = a.frame0 + fr;
    const uint8_t* cbase = a.color + (size_t)f * a.color_stride;
    u64* krow = a.keys[eye] + (size_t)fr * a.ws_stride_px + (size_t)y * W + (size_t)g * PX;
    u64* erow = EDGE ? a.ekeys[eye] + (size_t)fr * a.ws_stride_px + (size_t)y * W + (size_t)g * PX : nullptr;
    const uint32_t parity = (a.key_parity >> fr) & 1u;
    u64 key[PX];
#pragma unroll
    for (int q = 0; q < PX; ++q) key[q] = krow[q];
#pragma unroll
    for (int q = 0; q < PX; ++q)                               // only what this use left uncovered needs the next use's empty value
        if (!zkey_covered(key[q], parity)) krow[q] = zkey_empty(parity ^ 1u);
    uint32_t opx[PX], om[PX], spx[PX];
    float oz[PX];
    const u64* crow = MESH ? a.cbuf[eye] + (size_t)fr * a.ws_stride_px + (size_t)y * W + (size_t)g * PX : nullptr;
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        const bool covered = zkey_covered(key[q], parity);
        uint32_t rgb = 0;
        float zval = 0.0f;
        uint32_t kdepth = 0, ktie = 0;
        if (covered) {
            if (MESH) zkey_decode<true>(key[q], parity, kdepth, ktie); else zkey_decode<false>(key[q], parity, kdepth, ktie);
            if (MESH) {
                // the word carries the nearest fragment's colour -- unless the pixel was marked as an exact depth tie between colours:
                // then the rasteriser's second pass left draw id << 32 | colour of the first-drawn fragment at that depth in the side word
                if (ZOUT) zval = 1.0f / __uint_as_float(kdepth);
                rgb = ktie & 0xFFFFFFu;
                if (!(ktie & kNoTie)) rgb = (uint32_t)crow[q] & 0xFFFFFFu;
            } else {
                const uint32_t src = ktie;
                rgb = load_px_bytes(cbase + (size_t)(src >> 16) * a.color_pitch, (int)(src & 0xFFFFu));
                if (ZOUT) zval = __uint_as_float(kdepth);
            }
        }
        const bool hole = !covered || rgb == a.key_rgb;
        uint32_t out = hole ? 0u : rgb;
        uint32_t esrc = ~0u;
        if (EDGE && hole) {          // an edge point only matters where the render left a hole (sr:776): ~3 % of the pixels
            const u64 ek = erow[q];
            if (ek != kEmpty64) {
                esrc = (uint32_t)ek;
                if (a.edge_paint) out = load_px_bytes(cbase + (size_t)(esrc >> 16) * a.color_pitch, (int)(esrc & 0xFFFFu));
            }
        }
        opx[q] = out;
        om[q] = hole ? 255u : 0u;
        oz[q] = zval;
        if (SEED && a.seed[eye]) spx[q] = seed_pixel(a, a.fp[f], f, eye, g * PX + q, y, hole, esrc, MESH ? 1 : 0);
    }
    if (SEED && a.seed[eye]) RowIO<PX>::store_rgb(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)y * a.seed_pitch, g, spx);
    RowIO<PX>::store_rgb(a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)y * a.rgb_pitch, g, opx);
    RowIO<PX>::store_mask(a.mask[eye] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch, g, om);
    if (ZOUT && a.zout[eye])
        RowIO<PX>::store_z((float*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)y * a.zout_pitch), g, oz);
}

template <bool MESH>
static hipError_t launch_resolve_general(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const bool zout = a.zout[0] || a.zout[1];
    const bool edge = plan.remove_edges && plan.edge_points;
    const int flags = (zout ? 1 : 0) | (edge ? 4 : 0) | (a.seed[0] ? 8 : 0);
    const int px = plan.vec4 ? 4 : 1;
    const dim3 grid((a.W / px + 255) / 256, a.H, plan.n * 2), block(256);
#define MDVT_CASE(F)                                                                                  \
    case F:                                                                                           \
        if (px == 4) hipLaunchKernelGGL((k_resolve_general<4, F, MESH>), grid, block, 0, s, a);       \
        else hipLaunchKernelGGL((k_resolve_general<1, F, MESH>), grid, block, 0, s, a);               \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(4) MDVT_CASE(5) MDVT_CASE(8) MDVT_CASE(9) MDVT_CASE(12) MDVT_CASE(13)
    }
#undef MDVT_CASE
    if (edge) return launch_edge_keys_reset(a, plan.n, s);
    return hipGetLastError();
}

// =================================================================================================
// MESH MODE, pure stereo shift: one workgroup per (frame, output row), z-buffer in LDS
// =================================================================================================
//
// With v = grid_y (depth independent) every vertex row is a horizontal line on screen, so an output
// scanline k is covered by exactly ONE row of grid cells: the row c whose snapped span (Yt, Yb]
// contains the scanline (a scanline lying exactly on a vertex row belongs to the cells ABOVE it: the
// row below touches it only with top edges / a top vertex, which the fill rule -- left and bottom edges
// own their pixel centres, edge_in -- excludes for either orientation).  The 2-D rasterisation
// collapses to interval coverage along x.
//
//   stage    the two vertex rows c, c+1 once per workgroup: decode, d = dl/Z, 1/Z, snapped x for BOTH
//            eyes, packed colour -> 16 B per vertex in LDS (coalesced 12 B/lane HBM reads)
//   raster   per eye: one thread per cell walks the pixel centres inside its two triangles, shades each
//            covered fragment perspective-correctly and posts
//               key = ~bits(1/Z interpolated) << 32 | R | G<<8 | B<<16
//            with ds_min_u64 (min == nearest; an exact 1/Z tie between overlapping triangles of different
//            colour is detected and settled by draw order in two more passes over the row -- RowTies in
//            mdvt_device.h).  Spans longer than kShortSpan (rubber-sheet triangles
//            across depth edges) are handed to the whole wave: one lane's triangle is broadcast with
//            v_readlane and 64 lanes test 64 pixel centres at a time.
//   resolve  per eye: plain LDS reads -> colour-key hole test -> coalesced dwordx3 / dword stores.
// The z-buffer holds one eye at a time (the resolve of the left eye resets it), so LDS is
// 2*16*W + 8*W (+4*W with edge points) bytes: 77 KB at 1080p -> two workgroups per CU.

// One covered fragment of a row kernel: interpolated 1/Z and shaded colour into the row's z-buffer.
__device__ __forceinline__ void mesh_row_fragment(u64* zb, int px, float q0, float q1, float q2, uint32_t c0, uint32_t c1, uint32_t c2,
                                                  uint32_t draw, const RowTies& ties)
{
    const float iz = (q0 + q1) + q2;
    const float riz = rcp_exact(iz);
    post_row_fragment(zb, px, iz, shade_px(q0, q1, q2, riz, c0, c1, c2), draw, ties);
}

struct MeshVert { int XL, XR; float iz; uint32_t rgb; };
static_assert(sizeof(MeshVert) == 16, "MeshVert is one ds_read_b128");
constexpr int kShortSpan = 4;

__device__ __forceinline__ MeshVert mesh_vertex(uint32_t dpx, uint32_t cpx, int j, const FrameDev& fp)
{
    MeshVert v;
    const float z = decode_z(code16_of(dpx), fp.mult, fp.scale);
    const bool ok = z > kNear;
    float iz, d;
    rcp_div_exact(fp.dl, fast_operand(fp.dl), z, iz, d);
    const float gx = (float)j * fp.sx;
    v.XL = snap(gx + d);
    v.XR = snap(gx - d);
    v.iz = ok ? iz : 0.0f;                  // iz == 0 flags a vertex behind the near plane
    v.rgb = cpx;
    return v;
}

// Triangle `pass` (0: tri1 = A,B,C; 1: tri2 = A,C,D) of cell column j for one eye from the LDS vertices.
__device__ __forceinline__ bool mesh_tri_lds(TriSetup& t, const MeshVert& A, const MeshVert& B, const MeshVert& Cv,
                                             const MeshVert& D, int pass, int eye, int Yt, int Yb, int cull)
{
    const int XA = eye == 0 ? A.XL : A.XR, XB = eye == 0 ? B.XL : B.XR;
    const int XC = eye == 0 ? Cv.XL : Cv.XR, XD = eye == 0 ? D.XL : D.XR;
    // vertex order of the reference: tri1 = (v[i,j], v[i+1,j], v[i+1,j+1]); tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])
    if (pass == 0) return tri_setup_snapped(t, XA, Yt, A.iz, XB, Yb, B.iz, XC, Yb, Cv.iz, cull);
    return tri_setup_snapped(t, XA, Yt, A.iz, XC, Yb, Cv.iz, XD, Yt, D.iz, cull);
}

// LDS vertex array with 16-byte (colour inside) or 12-byte (colour re-read from the frame in the
// resolve phase; for wide frames whose 16-byte rows would not fit the 160 KB LDS) records.
template <bool VRGB>
struct VertStore {
    static constexpr int kDwords = VRGB ? 4 : 3;
    int* base;
    __device__ __forceinline__ void put(int idx, const MeshVert& v) const
    {
        int* p = base + (size_t)idx * kDwords;
        if (VRGB) *(int4*)p = make_int4(v.XL, v.XR, __float_as_int(v.iz), (int)v.rgb);
        else { p[0] = v.XL; p[1] = v.XR; p[2] = __float_as_int(v.iz); }
    }
    __device__ __forceinline__ MeshVert get(int idx) const
    {
        const int* p = base + (size_t)idx * kDwords;
        MeshVert v;
        if (VRGB) { const int4 q = *(const int4*)p; v.XL = q.x; v.XR = q.y; v.iz = __int_as_float(q.z); v.rgb = (uint32_t)q.w; }
        else { v.XL = p[0]; v.XR = p[1]; v.iz = __int_as_float(p[2]); v.rgb = 0; }
        return v;
    }
};

// Regular cell (both triangles in the grid's own orientation): on the scanline the cell is the interval
// between its two column edges, split by the diagonal A-C.  With
//   E_col_j(X) = h (X - XA) - (XB - XA) t,   E_diag(X) = h (X - XA) - (XC - XA) t
// tri1 = {E_col_j >= 0, E_diag < 0}, tri2 = {E_diag >= 0, E_col_j+1 < 0} (left edges own their centres), and the integer
// barycentric weights are these same edge values -- identical to the generic edge functions, three 64-bit
// subtractions per pixel instead of two triangle set-ups.
struct RegularCell {
    int XA, XB, XC, XD;
    float izA, izB, izC, izD;
    uint32_t cA, cB, cC, cD;
    int flags;              // bit0: tri1 removed, bit1: tri2 removed (dmt:1372)
    int j;                  // cell column (draw order)
};

__device__ __forceinline__ void regular_cell_pixel(const RegularCell& r, int px, int tt, int bb, int hh,
                                                   i64 kcol0, i64 kcol1, i64 kdiag, u64* zb, const RowTies& ties)
{
    const i64 hX = mul64(hh, px * kSubpix + kSubpix / 2);
    const i64 e0 = hX - kcol0, e1 = hX - kcol1, ed = hX - kdiag;
    const bool in1 = e0 >= 0 && ed < 0 && !(r.flags & 1);
    const bool in2 = ed >= 0 && e1 < 0 && !(r.flags & 2);
    if (!(in1 || in2)) return;
    const int wd = in1 ? r.XC - r.XB : r.XD - r.XA;          // the triangle's horizontal edge (operands selected, not products)
    const i64 area2 = mul64(hh, wd);
    const i64 wh = mul64(wd, in1 ? bb : tt);
    const i64 w0 = in1 ? wh : -e1;
    const i64 w1 = in1 ? -ed : wh;
    const i64 w2 = in1 ? e0 : ed;
    float q0, q1, q2;
    tri_weights(area2, r.izA, in1 ? r.izB : r.izC, in1 ? r.izC : r.izD, w0, w1, w2, q0, q1, q2);
    mesh_row_fragment(zb, px, q0, q1, q2, r.cA, in1 ? r.cB : r.cC, in1 ? r.cC : r.cD, ((in1 ? 0u : 1u) << 16) | (uint32_t)r.j, ties);
}

template <int PX, int FLAGS, int TPB, bool VRGB>
__global__ void __launch_bounds__(TPB) k_mesh_rows(RenderArgs a)
{
    constexpr bool ZOUT = FLAGS & 1, EDGES = FLAGS & 2, EDGEPTS = FLAGS & 4, SEED = FLAGS & 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W, H = a.H;
    u64* zb = (u64*)smem;                              // [W] z keys of the eye being rendered
    VertStore<VRGB> verts{(int*)(zb + W)};             // [2][W]: vertex rows c and c+1
    uint32_t* eb = (uint32_t*)(verts.base + 2 * (size_t)W * VertStore<VRGB>::kDwords);   // [W] edge-point keys (EDGEPTS)
    uint8_t* cfl = (uint8_t*)(eb + (EDGEPTS ? W : 0));  // [W] per-column flags (EDGES): bit0 tri1 removed, bit1 tri2 removed, bit2 vertex (k,j) unused
    uint16_t* kcode = (uint16_t*)(cfl + (EDGES ? (((size_t)W + 15) & ~(size_t)15) : 0));   // [W] 16-bit depth codes of source row k (EDGEPTS)
    RowTies ties;                                       // [W/32 + 1] tie bits + flag (exact depth ties, mdvt_device.h)
    ties.bits = (uint32_t*)(((uintptr_t)(kcode + (EDGEPTS ? W : 0)) + 15) & ~(uintptr_t)15);
    ties.nwords = (W + 31) / 32;
    ties.mode = 0;
    ties.force = (MDVT_DEBUG_SKIP(a) & 32) != 0;

    const int fr = blockIdx.x / H;
    const int k = blockIdx.x - fr * H;                 // output row
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    // ---- the cell row covering scanline k (uniform) ----
    const int Yc = k * kSubpix + kSubpix / 2;
    int ilo = (int)(((float)k + 0.5f) / fp.sy);
    ilo = ilo < 0 ? 0 : (ilo > H - 1 ? H - 1 : ilo);
    while (ilo > 0 && snap((float)ilo * fp.sy) >= Yc) --ilo;
    while (ilo + 1 <= H - 1 && snap((float)(ilo + 1) * fp.sy) < Yc) ++ilo;
    const int c = (ilo <= H - 2) ? ilo : -1;           // largest i with Ys(i) < Yc (fill rule: bottom edges own their centres); -1: below the last vertex row
    const int Yt = c >= 0 ? snap((float)c * fp.sy) : 0;
    const int Yb = c >= 0 ? snap((float)(c + 1) * fp.sy) : 0;
    const int tt = Yc - Yt, bb = Yb - Yc, hh = Yb - Yt;     // scanline position inside the cell row (sub-pixels)
    const float tf = c >= 0 ? (float)tt / (float)hh : 0.0f; // scanline height inside the cell row, for the candidate estimate only

    // ---- stage the two vertex rows, clear the z-buffer ----
    if (c >= 0 && !(MDVT_DEBUG_SKIP(a) & 4)) {
        const int ngroups = W / PX;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)(c + r) * a.depth_pitch;
            const uint8_t* crow = a.color + (size_t)f * a.color_stride + (size_t)(c + r) * a.color_pitch;
            for (int g = tid; g < ngroups; g += TPB) {
                uint32_t dpx[PX], cpx[PX];
                RowIO<PX>::load(drow, g, dpx);
                RowIO<PX>::load(crow, g, cpx);
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    verts.put(r * W + g * PX + q, mesh_vertex(dpx[q], cpx[q], g * PX + q, fp));
                    if (EDGEPTS && c + r == k) kcode[g * PX + q] = (uint16_t)code16_of(dpx[q]);
                }
            }
        }
    }
    for (int x = tid; x < W; x += TPB) zb[x] = kEmpty64;
    for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
    if (EDGEPTS) for (int x = tid; x < W; x += TPB) eb[x] = kEmpty32;
    if (EDGES) {
        // removed-triangle flags of this cell row and unused-vertex flags of source row k: one coalesced pass
        const size_t ncell_ = (size_t)(W - 1) * (H - 1);
        const uint8_t* ti = c >= 0 ? a.tri_invalid + (size_t)fr * a.ws_stride_tri + (size_t)c * (W - 1) : nullptr;
        const uint8_t* ur = a.unused + (size_t)fr * a.ws_stride_px + (size_t)k * W;
        for (int x = tid; x < W; x += TPB) {
            uint32_t fl = 0;
            if (ti && x < W - 1) fl = (ti[x] ? 1u : 0u) | (ti[ncell_ + x] ? 2u : 0u);
            if (EDGEPTS && ur[x]) fl |= 4u;
            cfl[x] = (uint8_t)fl;
        }
    }
    __syncthreads();

    const uint8_t* crow_k = a.color + (size_t)f * a.color_stride + (size_t)k * a.color_pitch;

#pragma unroll 1
    for (int eye = 0; eye < 2; ++eye) {
        // ---- rasterise this eye (passes 1 and 2 only for a row with exact depth ties, see RowTies) ----
#pragma unroll 1
        for (ties.mode = 0; ties.mode < 3; ++ties.mode) {
        if (c >= 0 && !(MDVT_DEBUG_SKIP(a) & 1)) {
            for (int j0 = 0; j0 < W - 1; j0 += TPB) {
                const int j = j0 + tid;
                TriSetup t[2];
                int px0[2], px1[2];
                uint32_t col[2][3];
                bool lng[2] = {false, false};
                RegularCell rc;
                int rp0 = 0, rp1 = -1;
                bool lngcell = false;
                if (j < W - 1) {
                    const MeshVert A = verts.get(j), D = verts.get(j + 1), B = verts.get(W + j), Cv = verts.get(W + j + 1);
                    const int XA = eye == 0 ? A.XL : A.XR, XB = eye == 0 ? B.XL : B.XR;
                    const int XC = eye == 0 ? Cv.XL : Cv.XR, XD = eye == 0 ? D.XL : D.XR;
                    const uint32_t fl = EDGES ? cfl[j] : 0u;
                    const bool inv1 = fl & 1u, inv2 = fl & 2u;                                       // dmt:1372
                    uint32_t cA = A.rgb, cB = B.rgb, cC = Cv.rgb, cD = D.rgb;
                    if (!VRGB) {
                        const uint8_t* cr0 = a.color + (size_t)f * a.color_stride + (size_t)c * a.color_pitch;
                        const uint8_t* cr1 = cr0 + a.color_pitch;
                        cA = load_px_bytes(cr0, j); cD = load_px_bytes(cr0, j + 1);
                        cB = load_px_bytes(cr1, j); cC = load_px_bytes(cr1, j + 1);
                    }
                    col[0][0] = cA; col[0][1] = cB; col[0][2] = cC;
                    col[1][0] = cA; col[1][1] = cC; col[1][2] = cD;
                    const bool allok = A.iz > 0.0f && B.iz > 0.0f && Cv.iz > 0.0f && D.iz > 0.0f;
                    int p0 = floordiv_subpix(min(XA, XB) - kSubpix / 2 + kSubpix - 1);
                    int p1 = floordiv_subpix(max(XC, XD) - kSubpix / 2);
                    if (p0 < 0) p0 = 0;
                    if (p1 > W - 1) p1 = W - 1;
                    rc.XA = XA; rc.XB = XB; rc.XC = XC; rc.XD = XD;
                    rc.izA = A.iz; rc.izB = B.iz; rc.izC = Cv.iz; rc.izD = D.iz;
                    rc.cA = cA; rc.cB = cB; rc.cC = cC; rc.cD = cD;
                    rc.flags = (inv1 ? 1 : 0) | (inv2 ? 2 : 0);
                    rc.j = j;
                    if (allok && XC > XB && XD > XA) {
                        if (a.cull == 2) rc.flags = 3;                 // a regular cell is two front faces
                        if (rc.flags != 3) {
                            const i64 kcol0 = mul64(XB - XA, tt) + mul64(hh, XA);
                            const i64 kcol1 = mul64(XC - XD, tt) + mul64(hh, XD);
                            // The cell meets the scanline in [kcol0/h, kcol1/h) -- for a cell sheared across a
                            // horizontal depth edge that is ~1 px although its bounding box spans the whole
                            // disparity jump.  Candidate pixels from a float estimate of the two crossings,
                            // widened by one pixel (the estimate is good to ~0.1 px inside the snap range); the
                            // exact integer tests in regular_cell_pixel decide.
                            // crossings XA + (XB-XA) t/h and XD + (XC-XD) t/h from 32-bit conversions (the 64-bit kcol values
                            // would cost ~10 instructions each to convert); only the estimate, the integer tests decide
                            constexpr float kHalf = (float)(kSubpix / 2), kInv = 1.0f / (float)kSubpix;
                            int q0 = (int)floorf((((float)XA + (float)(XB - XA) * tf) - kHalf) * kInv);
                            int q1 = (int)floorf((((float)XD + (float)(XC - XD) * tf) - kHalf) * kInv) + 1;
                            if (q0 < p0) q0 = p0;
                            if (q1 > p1) q1 = p1;
                            rp0 = q0; rp1 = q1;
                            if (q1 - q0 >= kShortSpan) {
                                lngcell = true;           // rubber-sheet cell across a vertical depth edge: whole-wave path below
                            } else if (!(MDVT_DEBUG_SKIP(a) & 16)) {
                                const i64 kdiag = mul64(XC - XA, tt) + mul64(hh, XA);
                                for (int px = q0; px <= q1; ++px) regular_cell_pixel(rc, px, tt, bb, hh, kcol0, kcol1, kdiag, zb, ties);
                            }
                        }
                    } else {
                        // folded / degenerate / near-plane / long-span cells: the generic triangle path
#pragma unroll
                        for (int pass = 0; pass < 2; ++pass) {
                            if (pass == 0 ? inv1 : inv2) continue;
                            if (!mesh_tri_lds(t[pass], A, B, Cv, D, pass, eye, Yt, Yb, a.cull)) continue;
                            const TriSetup& tr = t[pass];
                            int q0p = floordiv_subpix(tr.minX - kSubpix / 2 + kSubpix - 1), q1p = floordiv_subpix(tr.maxX - kSubpix / 2);
                            if (q0p < 0) q0p = 0;
                            if (q1p > W - 1) q1p = W - 1;
                            px0[pass] = q0p; px1[pass] = q1p;
                            if (q1p - q0p >= kShortSpan) { lng[pass] = true; continue; }
                            for (int px = q0p; px <= q1p; ++px) {
                                float q0, q1, q2;
                                if (!tri_sample(tr, px, k, q0, q1, q2)) continue;
                                mesh_row_fragment(zb, px, q0, q1, q2, col[pass][0], col[pass][1], col[pass][2], ((uint32_t)pass << 16) | (uint32_t)j, ties);
                            }
                        }
                    }
                }
                // long regular cells: the whole wave works on one lane's cell at a time (64 pixel centres per step)
                {
                    u64 m = (MDVT_DEBUG_SKIP(a) & 8) ? 0ull : __ballot(lngcell);
                    while (m) {
                        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                        m &= m - 1;
                        RegularCell b;
#define MDVT_BI(fld) b.fld = __builtin_amdgcn_readlane(rc.fld, l)
#define MDVT_BF(fld) b.fld = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rc.fld), l))
#define MDVT_BU(fld) b.fld = (uint32_t)__builtin_amdgcn_readlane((int)rc.fld, l)
                        MDVT_BI(XA); MDVT_BI(XB); MDVT_BI(XC); MDVT_BI(XD); MDVT_BI(flags); MDVT_BI(j);
                        MDVT_BF(izA); MDVT_BF(izB); MDVT_BF(izC); MDVT_BF(izD);
                        MDVT_BU(cA); MDVT_BU(cB); MDVT_BU(cC); MDVT_BU(cD);
#undef MDVT_BI
#undef MDVT_BF
#undef MDVT_BU
                        const int bp0 = __builtin_amdgcn_readlane(rp0, l), bp1 = __builtin_amdgcn_readlane(rp1, l);
                        const i64 kcol0 = mul64(b.XB - b.XA, tt) + mul64(hh, b.XA);
                        const i64 kcol1 = mul64(b.XC - b.XD, tt) + mul64(hh, b.XD);
                        const i64 kdiag = mul64(b.XC - b.XA, tt) + mul64(hh, b.XA);
                        for (int px = bp0 + lane; px <= bp1; px += 64) regular_cell_pixel(b, px, tt, bb, hh, kcol0, kcol1, kdiag, zb, ties);
                    }
                }
                // long spans of irregular cells: the whole wave works on one lane's triangle at a time
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    u64 m = (MDVT_DEBUG_SKIP(a) & 8) ? 0ull : __ballot(lng[pass]);
                    while (m) {
                        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                        m &= m - 1;
                        TriSetup b;
#define MDVT_BCAST(fld) b.fld = __builtin_amdgcn_readlane(t[pass].fld, l)
                        MDVT_BCAST(dx0); MDVT_BCAST(dy0); MDVT_BCAST(dx1); MDVT_BCAST(dy1); MDVT_BCAST(dx2); MDVT_BCAST(dy2);
                        MDVT_BCAST(bx0); MDVT_BCAST(by0); MDVT_BCAST(bx1); MDVT_BCAST(by1); MDVT_BCAST(bx2); MDVT_BCAST(by2);
#undef MDVT_BCAST
                        const uint32_t alo = __builtin_amdgcn_readlane((int)(uint32_t)t[pass].area2, l);
                        const uint32_t ahi = __builtin_amdgcn_readlane((int)(uint32_t)((u64)t[pass].area2 >> 32), l);
                        b.area2 = (i64)(((u64)ahi << 32) | alo);
                        b.iz0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t[pass].iz0), l));
                        b.iz1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t[pass].iz1), l));
                        b.iz2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t[pass].iz2), l));
                        const int bp0 = __builtin_amdgcn_readlane(px0[pass], l), bp1 = __builtin_amdgcn_readlane(px1[pass], l);
                        const uint32_t bc0 = __builtin_amdgcn_readlane((int)col[pass][0], l);
                        const uint32_t bc1 = __builtin_amdgcn_readlane((int)col[pass][1], l);
                        const uint32_t bc2 = __builtin_amdgcn_readlane((int)col[pass][2], l);
                        const int bj = __builtin_amdgcn_readlane(j, l);
                        for (int px = bp0 + lane; px <= bp1; px += 64) {
                            float q0, q1, q2;
                            if (!tri_sample(b, px, k, q0, q1, q2)) continue;
                            mesh_row_fragment(zb, px, q0, q1, q2, bc0, bc1, bc2, ((uint32_t)pass << 16) | (uint32_t)bj, ties);
                        }
                    }
                }
            }
        }
        if (ties.mode == 0) {
        // ---- edge points of source row k (sr:589-606, 745-781): vertices of removed triangles ----
        if (EDGEPTS && !edge_row_deferred(fp, k)) {        // (scanlines erow_lo .. erow_hi: k_edge_rows_exact)
            const uint8_t* drow_k = a.depth + (size_t)f * a.depth_stride + (size_t)k * a.depth_pitch;
            const bool k_staged = c >= 0 && (k == c || k == c + 1) && !(MDVT_DEBUG_SKIP(a) & 4);
            const float guard = edge_col_guard(W);
            for (int j = tid; j < W; j += TPB) {
                if (!(cfl[j] & 4u)) continue;
                const uint32_t code = k_staged ? (uint32_t)kcode[j] : code16_of(load_px_bytes(drow_k, j));
                const float z = decode_z(code, fp.mult, fp.scale);
                if (!(z > kNear)) continue;
                const int x = edge_col_pure(fp, eye, (float)j * fp.sx, z, fp.dl / z, W, guard);     // sr:599-600, 746
                if (x >= 0) atomicMin(&eb[x], (code << 16) | (uint32_t)j);
            }
        }
        }
        __syncthreads();
        if (ties.mode == 0) {
            if (ties.bits[ties.nwords] == 0u) break;
            row_ties_prepare(zb, W, ties, tid, TPB);
            __syncthreads();
        }
        }
        const bool had_ties = ties.mode == 3;            // the loop ran to its end (a row without ties leaves it at mode 0)
        ties.mode = 0;

        // ---- resolve this eye ----
        uint8_t* orow = a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)k * a.rgb_pitch;
        uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)k * a.mask_pitch;
        float* zrow = ZOUT && a.zout[eye]
                          ? (float*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)k * a.zout_pitch)
                          : nullptr;
        for (int g = tid; g < W / PX && !(MDVT_DEBUG_SKIP(a) & 2); g += TPB) {
            uint32_t opx[PX], om[PX], spx[PX];
            float oz[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const int x = g * PX + q;
                const u64 key = zb[x];
                const bool covered = key != kEmpty64;
                const uint32_t rgb = covered ? (uint32_t)key & 0xFFFFFFu : 0u;
                float zval = 0.0f;
                if (ZOUT && covered) zval = 1.0f / row_word_iz((uint32_t)(key >> 32));
                const bool hole = !covered || rgb == a.key_rgb;
                uint32_t out = hole ? 0u : rgb;
                uint32_t esrc = ~0u;
                if (EDGEPTS) {
                    const uint32_t ek = eb[x];
                    if (hole && ek != kEmpty32) { if (a.edge_paint) out = load_px_bytes(crow_k, (int)(ek & 0xFFFFu)); esrc = ((uint32_t)k << 16) | (ek & 0xFFFFu); }
                    if (eye == 0) eb[x] = kEmpty32;
                }
                if (SEED && a.seed[eye]) spx[q] = seed_pixel(a, fp, f, eye, x, k, hole, esrc, 1);
                if (eye == 0) zb[x] = kEmpty64;               // ready for the right eye
                opx[q] = out;
                om[q] = hole ? 255u : 0u;
                if (ZOUT) oz[q] = zval;
            }
            RowIO<PX>::store_rgb(orow, g, opx);
            RowIO<PX>::store_mask(mrow, g, om);
            if (ZOUT && zrow) RowIO<PX>::store_z(zrow, g, oz);
            if (SEED && a.seed[eye]) RowIO<PX>::store_rgb(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)k * a.seed_pitch, g, spx);
        }
        if (had_ties) for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
        if (eye == 0) __syncthreads();
    }
}

}  // namespace MDVT_GRID

#if MDVT_SUBPIX_BITS == 8
using MDVT_GRID::RowIO;      // (the row loads / stores of the points section: plain memory access, nothing of the grid in them)
// =================================================================================================
// MESH MODE, general (pose / convergence): triangles rasterised into global 64-bit z keys
// =================================================================================================
//   key = ~bits(1/Z') << 32 | draw id (pass << 31 | i << 16 | j): the 64-bit minimum is "nearest, then first drawn" = GL_LESS
//   with the reference's draw order (dmt:1243-1254).  The colour travels beside it: EVERY fragment also leaves
//   draw id << 32 | rgb in a side buffer with a plain 64-bit store.  Fragments of one pixel overwrite each other in no
//   particular order, so the resolve pass takes the side buffer's colour only if its id is the winner's -- always the
//   case for the ~97 % of pixels that received one fragment -- and otherwise shades the winner from its id (three vertex
//   records read back).  Measured and rejected: (1) shading every pixel in the resolve (deferred shading proper): the
//   three 16-byte gathers per pixel make the resolve pass HBM-bound at 58 us per 1080p frame against 26; (2) storing the
//   colour only when the fragment takes the pixel: the returning atomic that needs stalls the rasteriser (product
//   default 2508 us per 16 frames against 2392 for (1)).

// (Stage 1, the vertex records: computed by the rasterisers themselves since r05 -- mdvt_mesh_general.hip, vertex_records.)

// (infill_using_normals and mark_lower_side: mdvt_normal_infill.hip)

// Touchly inverse-depth plane (sr:549-551, 689-691, 825-829).
__global__ void __launch_bounds__(256) k_touchly_depth(const float* __restrict__ depth, size_t depth_pitch,
                                                       uint8_t* __restrict__ rgb, size_t rgb_pitch, int W, int H,
                                                       float tmax, float tmin, float k, int zero_is_far)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const float d = ((const float*)((const uint8_t*)depth + (size_t)y * depth_pitch))[x];
    const float v = rintf(fmaxf(0.0f, fminf(d, tmax) - tmin) * k);
    uint32_t q = (uint32_t)v & 0xFFu;                       // .astype(np.uint8)
    if (zero_is_far && q == 0) q = 255;                     // sr:690 / 827
    q = 255u - q;                                           // Touchly uses reverse depth
    store_px_bytes(rgb + (size_t)y * rgb_pitch, x, q | (q << 8) | (q << 16));
}

hipError_t launch_touchly_depth(const float* depth, size_t depth_pitch, uint8_t* rgb, size_t rgb_pitch, int W, int H,
                                float tmax, float tmin, float k, int zero_is_far, hipStream_t s)
{
    dim3 grid((W + 255) / 256, H);
    hipLaunchKernelGGL(k_touchly_depth, grid, dim3(256), 0, s, depth, depth_pitch, rgb, rgb_pitch, W, H, tmax, tmin, k, zero_is_far);
    return hipGetLastError();
}

// =================================================================================================
// VR180: convert_to_equirectangular (sr:25-86) = cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) through
// separable lookup tables
// =================================================================================================
// One thread = PX output pixels of one row of one image.  Coordinates are rounded to 1/32 px (half to even),
// the four taps get the integer weights (32-fx)(32-fy)*32 ... (sum 2^15), taps outside the image are 0, the
// result is (sum + 2^14) >> 15.  A table entry of -1 marks an angle outside the input fov: the pixel is black.
__device__ __forceinline__ uint32_t remap_tap4(const uint8_t* __restrict__ src, size_t pitch, int W, int H,
                                               int ix, int iy, int fx, int fy)
{
    const int w00 = (32 - fx) * (32 - fy) * 32, w10 = fx * (32 - fy) * 32, w01 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
    const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const uint32_t p00 = (x0 && y0) ? load_px_bytes(src + (size_t)iy * pitch, ix) : 0u;
    const uint32_t p10 = (x1 && y0 && w10) ? load_px_bytes(src + (size_t)iy * pitch, ix + 1) : 0u;
    const uint32_t p01 = (x0 && y1 && w01) ? load_px_bytes(src + (size_t)(iy + 1) * pitch, ix) : 0u;
    const uint32_t p11 = (x1 && y1 && w11) ? load_px_bytes(src + (size_t)(iy + 1) * pitch, ix + 1) : 0u;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sh = 8 * c;
        const int acc = w00 * (int)((p00 >> sh) & 0xFF) + w10 * (int)((p10 >> sh) & 0xFF) +
                        w01 * (int)((p01 >> sh) & 0xFF) + w11 * (int)((p11 >> sh) & 0xFF);
        out |= (uint32_t)((acc + (1 << 14)) >> 15) << sh;
    }
    return out;
}

template <int PX>
__global__ void __launch_bounds__(256) k_equirect_remap(const uint8_t* __restrict__ src, size_t src_pitch, size_t src_stride,
                                                        uint8_t* __restrict__ dst, size_t dst_pitch, size_t dst_stride,
                                                        int W, int H, const float* __restrict__ mx, const float* __restrict__ my)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (g >= W / PX) return;
    const uint8_t* simg = src + (size_t)blockIdx.z * src_stride;
    uint8_t* drow = dst + (size_t)blockIdx.z * dst_stride + (size_t)y * dst_pitch;
    const float fyv = my[y];
    uint32_t out[PX];
    if (fyv == -1.0f) {
#pragma unroll
        for (int q = 0; q < PX; ++q) out[q] = 0u;
    } else {
        const int sy = (int)rintf(fyv * 32.0f);
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const float fxv = mx[g * PX + q];
            if (fxv == -1.0f) { out[q] = 0u; continue; }
            const int sx = (int)rintf(fxv * 32.0f);
            out[q] = remap_tap4(simg, src_pitch, W, H, sx >> 5, sy >> 5, sx & 31, sy & 31);
        }
    }
    RowIO<PX>::store_rgb(drow, g, out);
}

hipError_t launch_equirect_remap(const uint8_t* src, size_t src_pitch, size_t src_stride, uint8_t* dst, size_t dst_pitch,
                                 size_t dst_stride, int n, int W, int H, const float* mx, const float* my, hipStream_t s)
{
    const bool vec4 = W % 4 == 0 && ((uintptr_t)dst % 4 == 0) && dst_pitch % 4 == 0 && dst_stride % 4 == 0;
    if (vec4) {
        dim3 grid((W / 4 + 255) / 256, H, n);
        hipLaunchKernelGGL((k_equirect_remap<4>), grid, dim3(256), 0, s, src, src_pitch, src_stride, dst, dst_pitch, dst_stride, W, H, mx, my);
    } else {
        dim3 grid((W + 255) / 256, H, n);
        hipLaunchKernelGGL((k_equirect_remap<1>), grid, dim3(256), 0, s, src, src_pitch, src_stride, dst, dst_pitch, dst_stride, W, H, mx, my);
    }
    return hipGetLastError();
}

// =================================================================================================
// infill-mask completion (sr:803-808, 114-153): level-synchronous Telea inpaint + masked Gaussian
// =================================================================================================
// State per image: stamp u16 (0 known from the start, 0xFFFF unknown, r = filled in round r), T f32 (written by the fill pass only: known pixels read as 0), the work
// image (seed copy, filled in place).  Round r reads only pixels with stamp < r, so the in-place writes of the
// same round (stamp = r) are never observed: one launch = one Jacobi step, no double buffering.
constexpr uint16_t kTeleaUnknown = 0xFFFFu;
constexpr uint32_t kTeleaNeedBit = 0x8000u;        // from the list scatter on: top bit of a level word = "this pixel's estimate is needed"
constexpr uint32_t kTeleaLevelMask = 0x7FFFu;      // (levels stay below 32767: max_rounds <= 32766)
#ifndef MDVT_NC_STRIDE
#define MDVT_NC_STRIDE 32
#endif
// The per-level counters of needed pixels take the appends of every workgroup of a launch: atomics on one address serialise at
// ~4 ns each, and neighbouring levels' counters in one cache line queue behind each other -- one counter per 128-byte line.
constexpr uint32_t kNcStride = MDVT_NC_STRIDE;

// The level of a pixel -- the round in which the level-synchronous front reaches it -- is its 4-connected distance to the
// nearest known pixel: an L1 distance transform, two separable passes (A) instead of one dependent launch per level.
// Then, level by level, so that the expensive estimate only runs where the result can reach a hole:
//   A  k_telea_dt_rows / k_telea_dt_cols   stamp = L1 distance to the nearest known pixel (0 = known), capped at max_rounds;
//                                   per image last_round = the level of its deepest key-coloured pixel (later levels
//                                   are never needed) and remaining = key-coloured pixels beyond max_rounds;
//      (level sizes: last sweep of the transform) / k_telea_scan / k_telea_sort   offsets, and the key-coloured pixels
//                                   of level r appended to nlist[offs[r] ..) -- the first needed pixels of each level.
//   B  k_telea_need    r = R .. 2   which estimates are needed: key-coloured pixels, and every pixel of a lower
//                                   level that a needed pixel reads (its radius-3 disc and their 4-neighbours) -- which
//                                   also closes the set under "T of a pixel needs T of its lower 4-neighbours".  The launch
//                                   for level r walks the needed pixels of that level only (complete by then: levels are
//                                   1-Lipschitz, so they were all marked by levels r+1 .. r+5) and appends what it marks to
//                                   the lists of levels r-5 .. r-1; a pixel is appended by whoever sets its need flag first.
//   C  k_telea_fill    r = 1 .. R   T (FastMarching_solve over the four quadrants) and Telea's estimate for the needed
//                                   pixels of level r, reading levels < r.
// A black (non-hole) pixel that no key-coloured pixel depends on is never estimated -- it returns to black at
// sr:807 anyway -- which removes ~90 % of the estimates (and T solves) of a front that also grows outwards from the
// holes.  R (the deepest level any image needs) is read back by the host after pass A: passes B and C are launched
// for exactly the levels that exist (round 1 launched all max_rounds levels of all three passes, 768 launches of which
// ~620 found nothing to do).
struct TeleaArgs {
    uint16_t* stamp; float* T; uint8_t* img;      // [n][H*W] / [n][H*W*3]
    uint8_t* need;                                // [n][H*W] 1 = this pixel's estimate is needed (and it is in nlist)
    uint32_t* nlist;                              // the needed pixels of each level; level r owns [offs[r], offs[r] + counts[r])
    uint32_t* counts;                             // [max_rounds + 2] level sizes (all pixels of the level: the capacity of its nlist part)
    uint32_t* offs;                               // [max_rounds + 2] level offsets into nlist
    uint32_t* ncounts;                            // [max_rounds + 2] needed pixels per level so far
    uint32_t* remaining;                          // [n] key-coloured pixels not reached yet
    uint32_t* last_round;                         // [n]
    int W, H, n;
    uint32_t key_rgb;
};

constexpr int kDtInf = 1 << 20;          // "no known pixel in this direction" (any real distance is < 2^17)

// T is zeroed with a memset beforehand; this pass writes 0 (known) / 0xFFFF (to fill) stamps, the need flags (key-coloured
// pixels) and the work image.
template <int PX>
__global__ void __launch_bounds__(128) k_telea_init(ImageSet seed, TeleaArgs a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    const int W = a.W, H = a.H;
    if (g * PX >= W) return;
    const size_t o = (size_t)im * W * H + (size_t)y * W + (size_t)g * PX;
    uint32_t px[PX];
    RowIO<PX>::load(seed.image(im) + (size_t)y * seed.pitch, g, px);
    uint32_t st = 0, nd = 0;
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        const bool green = px[q] == a.key_rgb;
        if (green || px[q] == 0u) {                                        // sr:803-805: key-coloured or black = to inpaint
            if (PX == 4) { if (q < 2) st |= (uint32_t)kTeleaUnknown << (16 * q); }
            else a.stamp[o + q] = kTeleaUnknown;
        } else if (PX != 4) a.stamp[o + q] = 0;
        if (PX == 4) nd |= (green ? 1u : 0u) << (8 * q); else a.need[o + q] = green ? 1 : 0;
    }
    if (PX == 4) {
        uint32_t st1 = 0;
#pragma unroll
        for (int q = 2; q < 4; ++q) if (px[q] == a.key_rgb || px[q] == 0u) st1 |= (uint32_t)kTeleaUnknown << (16 * (q - 2));
        *reinterpret_cast<uint2*>(a.stamp + o) = make_uint2(st, st1);
        *reinterpret_cast<uint32_t*>(a.need + o) = nd;
    }
    RowIO<PX>::store_rgb(a.img + 3 * ((size_t)im * W * H + (size_t)y * W), g, px);
}

// Pass A, rows: stamp[x] = distance to the nearest known pixel of the same row (0xFFFF: none), in place.  One workgroup per
// (row, image); a thread owns a contiguous segment, the nearest known pixels outside it come from a block-wide scan.
__global__ void __launch_bounds__(256) k_telea_dt_rows(uint16_t* __restrict__ stamp, int W, int H)
{
    __shared__ int sl[256], sf[256];
    uint16_t* d = stamp + ((size_t)blockIdx.y * H + blockIdx.x) * W;
    const int t = threadIdx.x;
    const int seg = (W + 255) / 256, x0 = min(t * seg, W), x1 = min(x0 + seg, W);
    int last = -kDtInf, first = kDtInf;
    for (int x = x0; x < x1; ++x)
        if (d[x] == 0) { last = x; if (first == kDtInf) first = x; }
    sl[t] = last; sf[t] = first;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // inclusive prefix max of `last`, inclusive suffix min of `first`
        const int vl = t >= off ? sl[t - off] : -kDtInf, vf = t + off < 256 ? sf[t + off] : kDtInf;
        __syncthreads();
        sl[t] = max(sl[t], vl); sf[t] = min(sf[t], vf);
        __syncthreads();
    }
    int run = t > 0 ? sl[t - 1] : -kDtInf;                // nearest known pixel left of the segment
    for (int x = x0; x < x1; ++x) {
        if (d[x] == 0) run = x;
        const int v = x - run;
        d[x] = (uint16_t)(v < 0xFFFF ? v : 0xFFFF);
    }
    run = t < 255 ? sf[t + 1] : kDtInf;                   // ... and right of it
    for (int x = x1 - 1; x >= x0; --x) {
        if (d[x] == 0) run = x;
        const int v = run - x;
        if (v < (int)d[x]) d[x] = (uint16_t)v;
    }
}

// The same with 16-byte row accesses: a thread owns 8 * VEC consecutive pixels (W % 8 == 0, W <= 2048 * VEC).
template <int VEC>
__global__ void __launch_bounds__(256) k_telea_dt_rows_vec(uint16_t* __restrict__ stamp, int W, int H)
{
    __shared__ int sl[256], sf[256];
    uint16_t* d = stamp + ((size_t)blockIdx.y * H + blockIdx.x) * W;
    const int t = threadIdx.x;
    constexpr int N = 8 * VEC;
    const int x0 = t * N;
    uint32_t w[4 * VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        uint4 q = make_uint4(~0u, ~0u, ~0u, ~0u);                       // past the row end: "unknown", never a zero
        if (x0 + 8 * v < W) q = *reinterpret_cast<const uint4*>(d + x0 + 8 * v);
        w[4 * v] = q.x; w[4 * v + 1] = q.y; w[4 * v + 2] = q.z; w[4 * v + 3] = q.w;
    }
    auto val = [&](int k) { return (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu; };
    int last = -kDtInf, first = kDtInf;
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (val(k) == 0u) { last = x0 + k; if (first == kDtInf) first = x0 + k; }
    sl[t] = last; sf[t] = first;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // inclusive prefix max of `last`, inclusive suffix min of `first`
        const int vl = t >= off ? sl[t - off] : -kDtInf, vf = t + off < 256 ? sf[t + off] : kDtInf;
        __syncthreads();
        sl[t] = max(sl[t], vl); sf[t] = min(sf[t], vf);
        __syncthreads();
    }
    int out[N];
    int run = t > 0 ? sl[t - 1] : -kDtInf;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (val(k) == 0u) run = x0 + k;
        out[k] = min(x0 + k - run, 0xFFFF);
    }
    run = t < 255 ? sf[t + 1] : kDtInf;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        if (val(k) == 0u) run = x0 + k;
        out[k] = min(out[k], run - (x0 + k));
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        if (x0 + 8 * v >= W) continue;
        uint4 q;
        q.x = (uint32_t)out[8 * v] | ((uint32_t)out[8 * v + 1] << 16); q.y = (uint32_t)out[8 * v + 2] | ((uint32_t)out[8 * v + 3] << 16);
        q.z = (uint32_t)out[8 * v + 4] | ((uint32_t)out[8 * v + 5] << 16); q.w = (uint32_t)out[8 * v + 6] | ((uint32_t)out[8 * v + 7] << 16);
        *reinterpret_cast<uint4*>(d + x0 + 8 * v) = q;
    }
}

constexpr int kLevelBins = 4096;      // levels counted / slotted in LDS; deeper ones go straight to the global counters

// Pass A, columns: the two sweeps of the L1 transform (down: D[y] = min(D[y-1] + 1, d[y]); up the same from below), in
// place.  One workgroup = 64 columns x 16 row segments; the value entering a segment comes from a scan over the segments'
// exit values.  The last sweep also caps the level at max_rounds and collects, per image, the deepest key-coloured level
// (last_round) and the number of key-coloured pixels beyond the cap (remaining).
__global__ void __launch_bounds__(1024) k_telea_dt_cols(TeleaArgs a, uint32_t max_rounds)
{
    __shared__ int ex[16][64], carry[16][64];
    __shared__ uint32_t s_rem, s_max;
    __shared__ uint32_t hist[kLevelBins];          // the level sizes (every reached pixel), added to a.counts at the end
    const int W = a.W, H = a.H;
    const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + cx, im = blockIdx.y;
    const bool act = x < W;
    const int seglen = (H + 15) / 16, y0 = min(sg * seglen, H), y1 = min(y0 + seglen, H);
    const size_t base = (size_t)im * W * H + (act ? x : 0);
    uint16_t* d = a.stamp + base;
    if (threadIdx.x == 0) { s_rem = 0u; s_max = 0u; }
    for (int b = threadIdx.x; b < kLevelBins; b += 1024) hist[b] = 0u;
    auto val = [&](int y) { const int v = d[(size_t)y * W]; return v == 0xFFFF ? kDtInf : v; };
    auto put = [&](int y, int v) { d[(size_t)y * W] = (uint16_t)(v < 0xFFFF ? v : 0xFFFF); };
    // ---- down ----
    int run = kDtInf;
    if (act) for (int y = y0; y < y1; ++y) run = min(run + 1, val(y));
    ex[sg][cx] = run;
    __syncthreads();
    if (sg == 0) {
        int c = kDtInf;
        for (int q = 0; q < 16; ++q) {
            carry[q][cx] = c;
            const int len = min((q + 1) * seglen, H) - min(q * seglen, H);
            c = min(ex[q][cx], c + len);
        }
    }
    __syncthreads();
    run = carry[sg][cx];
    if (act) for (int y = y0; y < y1; ++y) { run = min(run + 1, val(y)); put(y, run); }
    __syncthreads();            // (a column's segments are all in this workgroup: its writes above are visible below)
    // ---- up ----
    run = kDtInf;
    if (act) for (int y = y1 - 1; y >= y0; --y) run = min(run + 1, val(y));
    ex[sg][cx] = run;
    __syncthreads();
    if (sg == 0) {
        int c = kDtInf;
        for (int q = 15; q >= 0; --q) {
            carry[q][cx] = c;
            const int len = min((q + 1) * seglen, H) - min(q * seglen, H);
            c = min(ex[q][cx], c + len);
        }
    }
    __syncthreads();
    run = carry[sg][cx];
    uint32_t rem = 0, lmax = 0;
    if (act) {
        const uint8_t* key = a.need + base;
        for (int y = y1 - 1; y >= y0; --y) {
            run = min(run + 1, val(y));
            const bool reached = run <= (int)max_rounds;
            d[(size_t)y * W] = reached ? (uint16_t)run : kTeleaUnknown;
            if (reached && run >= 1) { if (run < kLevelBins) atomicAdd(&hist[run], 1u); else atomicAdd(&a.counts[run], 1u); }
            if (key[(size_t)y * W]) { if (reached) lmax = max(lmax, (uint32_t)run); else ++rem; }
        }
    }
    if (rem) atomicAdd(&s_rem, rem);
    if (lmax) atomicMax(&s_max, lmax);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_rem) atomicAdd(&a.remaining[im], s_rem);
        if (s_max) atomicMax(&a.last_round[im], s_max);
    }
    for (int b = threadIdx.x; b < kLevelBins; b += 1024)
        if (hist[b]) atomicAdd(&a.counts[b], hist[b]);
}

// The same with a column segment held in registers between the sweeps (H <= 16 * SEG): the stamps are read once and
// written once instead of four times and twice.
template <int SEG>
__global__ void __launch_bounds__(1024) k_telea_dt_cols_reg(TeleaArgs a, uint32_t max_rounds)
{
    __shared__ int ex[16][64], carry[16][64];
    __shared__ uint32_t s_rem, s_max;
    __shared__ uint32_t hist[kLevelBins];          // the level sizes (every reached pixel), added to a.counts at the end
    const int W = a.W, H = a.H;
    const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + cx, im = blockIdx.y;
    const bool act = x < W;
    const int seglen = (H + 15) / 16, y0 = min(sg * seglen, H), y1 = min(y0 + seglen, H);
    const int len = act ? y1 - y0 : 0;
    const size_t base = (size_t)im * W * H + (act ? x : 0);
    uint16_t* d = a.stamp + base;
    if (threadIdx.x == 0) { s_rem = 0u; s_max = 0u; }
    for (int b = threadIdx.x; b < kLevelBins; b += 1024) hist[b] = 0u;
    int v[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
        v[k] = kDtInf;
        if (k < len) { const int q = d[(size_t)(y0 + k) * W]; v[k] = q == 0xFFFF ? kDtInf : q; }
    }
    // ---- down ----
    int run = kDtInf;
#pragma unroll
    for (int k = 0; k < SEG; ++k) if (k < len) run = min(run + 1, v[k]);
    ex[sg][cx] = run;
    __syncthreads();
    if (sg == 0) {
        int c = kDtInf;
        for (int q = 0; q < 16; ++q) {
            carry[q][cx] = c;
            const int l = min((q + 1) * seglen, H) - min(q * seglen, H);
            c = min(ex[q][cx], c + l);
        }
    }
    __syncthreads();
    run = carry[sg][cx];
#pragma unroll
    for (int k = 0; k < SEG; ++k) if (k < len) { run = min(run + 1, v[k]); v[k] = run; }
    __syncthreads();
    // ---- up ----
    run = kDtInf;
#pragma unroll
    for (int k = SEG - 1; k >= 0; --k) if (k < len) run = min(run + 1, v[k]);
    ex[sg][cx] = run;
    __syncthreads();
    if (sg == 0) {
        int c = kDtInf;
        for (int q = 15; q >= 0; --q) {
            carry[q][cx] = c;
            const int l = min((q + 1) * seglen, H) - min(q * seglen, H);
            c = min(ex[q][cx], c + l);
        }
    }
    __syncthreads();
    run = carry[sg][cx];
    uint32_t rem = 0, lmax = 0;
    const uint8_t* key = a.need + base;
#pragma unroll
    for (int k = SEG - 1; k >= 0; --k) {
        if (k >= len) continue;
        run = min(run + 1, v[k]);
        const bool reached = run <= (int)max_rounds;
        d[(size_t)(y0 + k) * W] = reached ? (uint16_t)run : kTeleaUnknown;
        if (reached && run >= 1) { if (run < kLevelBins) atomicAdd(&hist[run], 1u); else atomicAdd(&a.counts[run], 1u); }
        if (key[(size_t)(y0 + k) * W]) { if (reached) lmax = max(lmax, (uint32_t)run); else ++rem; }
    }
    if (rem) atomicAdd(&s_rem, rem);
    if (lmax) atomicMax(&s_max, lmax);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_rem) atomicAdd(&a.remaining[im], s_rem);
        if (s_max) atomicMax(&a.last_round[im], s_max);
    }
    for (int b = threadIdx.x; b < kLevelBins; b += 1024)
        if (hist[b]) atomicAdd(&a.counts[b], hist[b]);
}

// counts[0] = the deepest level any image needs (the host reads it back: passes B and C get exactly that many launches)
__global__ void k_telea_rmax(TeleaArgs a)
{
    uint32_t m = 0;
    for (int im = 0; im < a.n; ++im) m = max(m, a.last_round[im]);
    a.counts[0] = m;
}

// The first entries of the level lists: the key-coloured pixels.  (The level sizes -- every reached pixel of a level, the room
// its list may need -- are counted by the last sweep of the distance transform.)  A workgroup takes a 64 x 64 tile of one
// image -- so that the pixels of a level stay together tile by tile in the list, and the half-waves that later work through
// consecutive list entries read overlapping 9 x 9 neighbourhoods --; levels below kLevelBins are slotted in LDS first (one
// global atomic per occupied level and workgroup), deeper ones directly.
constexpr int kSortTile = 64;

__global__ void __launch_bounds__(256) k_telea_sort(TeleaArgs a)
{
    __shared__ uint32_t hist[kLevelBins];
    __shared__ uint32_t slot[kLevelBins];
    const int im = blockIdx.y;
    const uint32_t lr = a.last_round[im];
    if (lr == 0u) return;                                   // nothing key-coloured (or nothing reachable): nothing to do
    const uint32_t npx = (uint32_t)a.W * (uint32_t)a.H;
    const int tiles_x = (a.W + kSortTile - 1) / kSortTile;
    const int tx0 = (int)(blockIdx.x % tiles_x) * kSortTile, ty0 = (int)(blockIdx.x / tiles_x) * kSortTile;
    // a thread takes four consecutive pixels of a row (one dword of need flags where the row allows it): 16 threads per tile
    // row, 16 rows per step, 4 steps
    const int lx = (threadIdx.x & 15) * 4, ly0 = threadIdx.x >> 4;
    const uint16_t* st = a.stamp + (size_t)im * npx;
    const uint8_t* nd = a.need + (size_t)im * npx;
    const bool dwords = (a.W & 3) == 0;                     // (then every row starts on a dword of the flag plane)
    for (int b = threadIdx.x; b < kLevelBins; b += 256) hist[b] = 0u;
    __syncthreads();
    constexpr int kSteps = kSortTile / 16;
    uint32_t lv[kSteps][4];
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        const int px = tx0 + lx, py = ty0 + ly0 + 16 * k;
        const uint32_t o = (uint32_t)py * (uint32_t)a.W + (uint32_t)px;
        uint32_t flags = 0;
        if (py < a.H) {
            if (dwords && px + 3 < a.W) flags = *reinterpret_cast<const uint32_t*>(nd + o);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (px + q < a.W && nd[o + q]) flags |= 1u << (8 * q);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lv[k][q] = 0u;
            if ((flags >> (8 * q)) & 0xFFu) {               // key-coloured (~4 % of the pixels): only those look their level up
                const uint32_t sv = st[o + q];
                if (sv >= 1u && sv <= lr) {
                    lv[k][q] = sv;
                    a.stamp[(size_t)im * npx + o + q] = (uint16_t)(sv | kTeleaNeedBit);      // needed from the start
                    if (sv < (uint32_t)kLevelBins) atomicAdd(&hist[sv], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kLevelBins; b += 256) {
        const uint32_t c = hist[b];
        if (!c) continue;
        slot[b] = atomicAdd(&a.ncounts[(uint32_t)b * kNcStride], c); hist[b] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSteps; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t l = lv[k][q];
            if (!l) continue;
            const uint32_t e = (uint32_t)im * npx + (uint32_t)(ty0 + ly0 + 16 * k) * (uint32_t)a.W + (uint32_t)(tx0 + lx + q);
            const uint32_t pos = l < (uint32_t)kLevelBins ? slot[l] + atomicAdd(&hist[l], 1u) : atomicAdd(&a.ncounts[l * kNcStride], 1u);
            a.nlist[a.offs[l] + pos] = e;
        }
}

// offs[r] = counts[1] + ... + counts[r-1] for r = 1 .. n_levels + 1 (level 1 starts at 0).  One workgroup.
__global__ void __launch_bounds__(1024) k_telea_scan(TeleaArgs a, int n_levels)
{
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x, n = n_levels + 1;             // entries 1 .. n
    const int per = (n + 1023) / 1024, lo = min(1 + t * per, n + 1), hi = min(lo + per, n + 1);
    uint32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += a.counts[k];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int k = lo; k < hi; ++k) { a.offs[k] = run; run += a.counts[k]; }
}

// OpenCV's FastMarching_solve for one quadrant: k = the neighbour is known (in the image, filled before this level), t = its T.
__device__ __forceinline__ float telea_solve(bool k1, float t1, bool k2, float t2)
{
    const double a11 = k1 ? (double)t1 : 1.0e6, a22 = k2 ? (double)t2 : 1.0e6;
    const double m12 = a11 < a22 ? a11 : a22;
    double sol;
    if (k1) {
        if (k2) sol = fabs(a11 - a22) >= 1.0 ? 1.0 + m12 : (a11 + a22 + sqrt(2.0 - (a11 - a22) * (a11 - a22))) * 0.5;
        else sol = 1.0 + a11;
    } else if (k2) sol = 1.0 + a22;
    else sol = 1.0 + m12;
    return (float)sol;
}

// What the estimate of a pixel reads: the radius-3 disc and the 4-neighbours of its pixels (57 offsets, all within an L1
// distance of 5: the level of any of them differs from the pixel's own by at most 5).
struct NeedOffsets { int8_t dx[64], dy[64]; int n; };
constexpr NeedOffsets make_need_offsets()
{
    NeedOffsets t{};
    int n = 0;
    for (int dy = -4; dy <= 4; ++dy)
        for (int dx = -4; dx <= 4; ++dx) {
            const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
            if (ax + ay > 5 || (ax == 4 && ay > 1) || (ay == 4 && ax > 1) || (ax == 0 && ay == 0)) continue;
            t.dx[n] = (int8_t)dx; t.dy[n] = (int8_t)dy; ++n;
        }
    t.n = n;
    return t;
}
__device__ __constant__ const NeedOffsets kNeedOffsets = make_need_offsets();

constexpr int kNeedLanes = 8;            // lanes sharing the 56 offsets of one needed pixel
constexpr int kNeedStage = 512;          // newly marked pixels a workgroup collects per target level before it appends them

__global__ void __launch_bounds__(256) k_telea_need(TeleaArgs a, uint32_t r)
{
    __shared__ uint32_t stage[5][kNeedStage];
    __shared__ uint32_t cnt[5], base[5];
    const uint32_t count = a.ncounts[r * kNcStride], off = a.offs[r];
    constexpr uint32_t per_block = 256 / kNeedLanes;
    // XCD-aware dealing (workgroup b runs on XCD b % 8, each XCD has its own L2): every XCD walks one contiguous eighth of the
    // level's list -- neighbouring entries are neighbouring pixels, whose 9 x 9 windows share their cache lines
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const uint32_t per_xcd = (count + 7u) >> 3, lo_x = xcd * per_xcd, hi_x = min(lo_x + per_xcd, count);
    if (lo_x + slot * per_block >= hi_x) return;                       // (workgroup-uniform)
    const int W = a.W, H = a.H;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    if (threadIdx.x < 5) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int sub = threadIdx.x & (kNeedLanes - 1);
    // (from the list scatter on, "needed" is the top bit of a pixel's level word: one load tells level and flag)
    uint32_t* stamp_words = reinterpret_cast<uint32_t*>(a.stamp);
    uint32_t idx = lo_x + slot * per_block + threadIdx.x / kNeedLanes;
    uint32_t e_next = idx < hi_x ? a.nlist[off + idx] : 0u;
    for (; idx < hi_x; idx += nslot * per_block) {
        const uint32_t e = e_next, im = e / npx, o = e - im * npx;
        if (idx + nslot * per_block < hi_x) e_next = a.nlist[off + idx + nslot * per_block];     // (in flight during this entry)
        const int y = (int)(o / (uint32_t)W), x = (int)(o - (uint32_t)y * (uint32_t)W);
        const size_t ib = (size_t)im * npx;
        // three rounds with everything of a round in flight together (a loop over the lane's offsets with the flag test, the
        // atomic and the append inside is seven dependent round trips to L2 per entry: the floor of a level's launch)
        constexpr int kPer = (56 + kNeedLanes - 1) / kNeedLanes;
        static_assert(kPer * kNeedLanes >= 56, "every offset has a lane");
        uint32_t uu[kPer], su[kPer], old[kPer];
        bool want[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int q = sub + k * kNeedLanes;
            const int xx = x + kNeedOffsets.dx[q < kNeedOffsets.n ? q : 0], yy = y + kNeedOffsets.dy[q < kNeedOffsets.n ? q : 0];
            const bool in = q < kNeedOffsets.n && xx >= 0 && xx < W && yy >= 0 && yy < H;
            uu[k] = in ? (uint32_t)(ib + (size_t)yy * W + xx) : e;           // (the entry itself: level r, never marked)
            const uint32_t sw = a.stamp[uu[k]];
            su[k] = sw & kTeleaLevelMask;
            want[k] = su[k] != 0u && su[k] < r && !(sw & kTeleaNeedBit);
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t bit = kTeleaNeedBit << (16u * (uu[k] & 1u));
            old[k] = bit;                                                      // "already set"
            if (want[k]) old[k] = atomicOr(stamp_words + (uu[k] >> 1), bit);
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (old[k] & (kTeleaNeedBit << (16u * (uu[k] & 1u)))) continue;    // flag was set: somebody else appends (or has appended) it
            const uint32_t d = r - 1u - su[k];
            const uint32_t pos = d < 5u ? atomicAdd(&cnt[d], 1u) : (uint32_t)kNeedStage;
            if (pos < (uint32_t)kNeedStage) stage[d][pos] = uu[k];
            else a.nlist[a.offs[su[k]] + atomicAdd(&a.ncounts[su[k] * kNcStride], 1u)] = uu[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const uint32_t n = min(cnt[threadIdx.x], (uint32_t)kNeedStage);
        cnt[threadIdx.x] = n;
        base[threadIdx.x] = n ? a.offs[r - 1u - threadIdx.x] + atomicAdd(&a.ncounts[(r - 1u - threadIdx.x) * kNcStride], n) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 5; ++d)
        for (uint32_t i = threadIdx.x; i < cnt[d]; i += 256) a.nlist[base[d] + i] = stage[d][i];
}

// The radius-3 disc without its centre, in the oracle's row-major order (28 pixels): offset (dx, dy) and the distance factor
// (float)(1.0 / (vl * sqrt(vl))), vl = dx^2 + dy^2 -- six distinct values, written out (f64 arithmetic, rounded once to f32).
struct DiscPixel { float dx, dy, dst; int cell; };             // cell = index of the pixel in the 9 x 9 neighbourhood
#define MDVT_DISC(dx, dy, dst) {(float)(dx), (float)(dy), dst, ((dy) + 4) * 9 + (dx) + 4}
#define MDVT_D1 0x1.000000p+0f
#define MDVT_D2 0x1.6a09e6p-2f
#define MDVT_D4 0x1.000000p-3f
#define MDVT_D5 0x1.6e5b7ep-4f
#define MDVT_D8 0x1.6a09e6p-5f
#define MDVT_D9 0x1.2f684cp-5f
__device__ __constant__ const DiscPixel kDisc[32] = {
    MDVT_DISC(0, -3, MDVT_D9),
    MDVT_DISC(-2, -2, MDVT_D8), MDVT_DISC(-1, -2, MDVT_D5), MDVT_DISC(0, -2, MDVT_D4), MDVT_DISC(1, -2, MDVT_D5), MDVT_DISC(2, -2, MDVT_D8),
    MDVT_DISC(-2, -1, MDVT_D5), MDVT_DISC(-1, -1, MDVT_D2), MDVT_DISC(0, -1, MDVT_D1), MDVT_DISC(1, -1, MDVT_D2), MDVT_DISC(2, -1, MDVT_D5),
    MDVT_DISC(-3, 0, MDVT_D9), MDVT_DISC(-2, 0, MDVT_D4), MDVT_DISC(-1, 0, MDVT_D1), MDVT_DISC(1, 0, MDVT_D1), MDVT_DISC(2, 0, MDVT_D4), MDVT_DISC(3, 0, MDVT_D9),
    MDVT_DISC(-2, 1, MDVT_D5), MDVT_DISC(-1, 1, MDVT_D2), MDVT_DISC(0, 1, MDVT_D1), MDVT_DISC(1, 1, MDVT_D2), MDVT_DISC(2, 1, MDVT_D5),
    MDVT_DISC(-2, 2, MDVT_D8), MDVT_DISC(-1, 2, MDVT_D5), MDVT_DISC(0, 2, MDVT_D4), MDVT_DISC(1, 2, MDVT_D5), MDVT_DISC(2, 2, MDVT_D8),
    MDVT_DISC(0, 3, MDVT_D9),
    MDVT_DISC(0, 0, 0.0f), MDVT_DISC(0, 0, 0.0f), MDVT_DISC(0, 0, 0.0f), MDVT_DISC(0, 0, 0.0f)};
#undef MDVT_DISC

constexpr int kRedStride = 36;           // floats between the running sums of one pixel in LDS: 16-byte aligned rows, b128 reads without bank conflicts

// Pass C, lane-parallel: one half-wave (32 lanes) per needed pixel, lane j < 28 = disc pixel j.  The 9 x 9 neighbourhood is
// fetched once into LDS, coalesced along its rows; T comes from four lanes solving one quadrant each; every lane weighs
// its own disc pixel; the 10 running sums (Ia, Jx, Jy per channel and the weight) are then added up in the oracle's order
// j = 0..27 by 10 lanes reading the terms back from LDS -- the same left-to-right f32 chain as the oracle's loop, so the
// result is bit-identical to it.  (The level's latency is what bounds the deep levels, instruction issue the first ones.)
__global__ void __launch_bounds__(256) k_telea_fill(TeleaArgs a, uint32_t r)
{
    const uint32_t off = a.offs[r];
    const int W = a.W, H = a.H;
    const uint32_t npx = (uint32_t)W * (uint32_t)H;
    __shared__ __attribute__((aligned(16))) float red[8][10][kRedStride];
    __shared__ uint32_t wcol[8][81];
    __shared__ float wt[8][81];
    __shared__ uint8_t wkn[8][84];
    const int lane32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const uint32_t nneed = a.ncounts[r * kNcStride];
    const DiscPixel dp = kDisc[lane32];
    const int qv = 4 + ((lane32 & 1) ? 9 : -9), qh = 4 * 9 + 4 + ((lane32 & 2) ? 1 : -1);     // this lane's quadrant: cells (0, +-1) and (+-1, 0)
    // every entry of nlist is a pixel to estimate: they are dealt round-robin to all half-waves of the grid
    // (XCD-aware dealing as in the need pass: one contiguous eighth of the list per XCD)
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const uint32_t per_xcd = (nneed + 7u) >> 3, lo_x = xcd * per_xcd, hi_x = min(lo_x + per_xcd, nneed);
    // Software pipeline over a half-wave's pixels: the neighbourhood of pixel i + 1 (loads into registers) and the list index of
    // pixel i + 2 are in flight while pixel i is worked out from LDS -- pixels of one level never read each other's results.
    struct Cells { uint32_t c[3]; float t[3]; uint32_t sv[3]; bool inb[3]; };
    auto fetch = [&](uint32_t e, Cells& p) {
        const uint32_t im = e / npx, o = e - im * npx;
        const int y = (int)(o / (uint32_t)W), x = (int)(o - (uint32_t)y * (uint32_t)W);
        const size_t ib = (size_t)im * npx;
        const uint16_t* stamp = a.stamp + ib;
        const float* Tm = a.T + ib;
        const uint8_t* img = a.img + 3 * ib;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = min(lane32 + 32 * it, 80);                  // (lanes past cell 80 repeat it: their values are not committed)
            const int wy = q / 9, wx = q - 9 * wy;
            const int xx = x - 4 + wx, yy = y - 4 + wy;
            p.inb[it] = xx >= 0 && xx < W && yy >= 0 && yy < H;
            const size_t oo = (size_t)(p.inb[it] ? yy : y) * W + (p.inb[it] ? xx : x);
            __builtin_memcpy(&p.c[it], img + 3 * oo, 4);              // unaligned dword: the work image is padded by 4 bytes
            p.sv[it] = stamp[oo];
            p.t[it] = Tm[oo];
        }
    };
    auto commit = [&](const Cells& p) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = lane32 + 32 * it;
            if (q >= 81) continue;
            const uint32_t sv = p.sv[it] & kTeleaLevelMask;           // (an unreached pixel, 0xFFFF, stays beyond every level)
            wcol[hw][q] = p.c[it] & 0xFFFFFFu;
            wt[hw][q] = sv == 0u ? 0.0f : p.t[it];                    // T = 0 at every originally known pixel (nobody writes it there)
            wkn[hw][q] = (p.inb[it] && sv < r) ? 1 : 0;
        }
    };
    const uint32_t stride = nslot * 8;
    uint32_t k = lo_x + slot * 8 + hw;
    uint32_t e_cur = k < hi_x ? a.nlist[off + k] : 0u;
    uint32_t e_next = k + stride < hi_x ? a.nlist[off + k + stride] : 0u;
    Cells cells;
    if (k < hi_x) fetch(e_cur, cells);
    for (; k < hi_x; k += stride) {                                                         // half-wave uniform
        const uint32_t e = e_cur, im = e / npx, o = e - im * npx;
        const size_t ib = (size_t)im * npx;
        commit(cells);
        e_cur = e_next;
        if (k + stride < hi_x) fetch(e_cur, cells);
        if (k + 2 * stride < hi_x) e_next = a.nlist[off + k + 2 * stride];
        __builtin_amdgcn_wave_barrier();                   // LDS is in order within a wave: the reads below see these writes
        const uint8_t* kn = wkn[hw];
        const float* tt = wt[hw];
        const uint32_t* cc = wcol[hw];
        constexpr int C0 = 4 * 9 + 4;                      // the pixel itself
        // T of the pixel: FastMarching_solve over the four quadrants, one per lane of a quad; lane 0 keeps it for the levels above
        float t = telea_solve(kn[C0 - 4 + qv] != 0, tt[C0 - 4 + qv], kn[qh] != 0, tt[qh]);
        t = fminf(t, __shfl_xor(t, 1));
        t = fminf(t, __shfl_xor(t, 2));
        if (lane32 == 0) a.T[e] = t;
        const bool kxp = kn[C0 + 1] != 0, kxm = kn[C0 - 1] != 0, kyp = kn[C0 + 9] != 0, kym = kn[C0 - 9] != 0;
        const float txp = tt[C0 + 1], txm = tt[C0 - 1], typ = tt[C0 + 9], tym = tt[C0 - 9];
        float gtx, gty;
        if (kxp) gtx = kxm ? (txp - txm) * 0.5f : txp - t;
        else gtx = kxm ? t - txm : 0.0f;
        if (kyp) gty = kym ? (typ - tym) * 0.5f : typ - t;
        else gty = kym ? t - tym : 0.0f;

        // terms of this lane's disc pixel: +Ia (3), -Jx (3), -Jy (3), +s; all 0 where the disc pixel is not known
        float term[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) term[c] = 0.0f;
        const int cell = dp.cell;
        if (lane32 < 28 && kn[cell]) {
            const float rx = -dp.dx, ry = -dp.dy;
            const float lev = (float)(1.0 / (1.0 + fabs((double)(tt[cell] - t))));
            float dir = rx * gtx + ry * gty;
            if (fabsf(dir) <= 0.01f) dir = 0.000001f;
            const float w = fabsf((dp.dst * lev) * dir);
            const bool xp = kn[cell + 1] != 0, xm = kn[cell - 1] != 0, yp = kn[cell + 9] != 0, ym = kn[cell - 9] != 0;
            const uint32_t c0 = cc[cell];
            const uint32_t cxp = xp ? cc[cell + 1] : c0, cxm = xm ? cc[cell - 1] : c0;        // an unknown neighbour stands in as the pixel itself:
            const uint32_t cyp = yp ? cc[cell + 9] : c0, cym = ym ? cc[cell - 9] : c0;        // one-sided and missing differences fall out of a - b
            const float sx = (xp && xm) ? 2.0f : 1.0f, sy = (yp && ym) ? 2.0f : 1.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int sh = 8 * ch;
                const int v0 = (c0 >> sh) & 0xFF;
                const int ddx = (int)((cxp >> sh) & 0xFF) - (int)((cxm >> sh) & 0xFF), ddy = (int)((cyp >> sh) & 0xFF) - (int)((cym >> sh) & 0xFF);
                const float gix = (float)ddx * sx, giy = (float)ddy * sy;
                term[ch] = w * (float)v0;              // Ia += .
                term[3 + ch] = -(w * (gix * rx));      // Jx -= .
                term[6 + ch] = -(w * (giy * ry));      // Jy -= .
            }
            term[9] = w;                               // s  += .
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) red[hw][c][lane32] = term[c];
        __builtin_amdgcn_wave_barrier();                   // the half-wave's LDS writes precede its reads (same wave: program order + lgkmcnt)
        float acc = 0.0f;
        if (lane32 < 10) {
            acc = lane32 == 9 ? 1.0e-20f : 0.0f;
            const float4* row = reinterpret_cast<const float4*>(red[hw][lane32]);
#pragma unroll
            for (int j4 = 0; j4 < 7; ++j4) {
                const float4 v = row[j4];
                acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w;      // a term of 0 (pixel not known) leaves acc unchanged, exactly
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int hbase = (threadIdx.x & 63) & 32;         // first lane of this half-wave inside the wave
        const int ch = lane32 < 3 ? lane32 : 0;
        const float Ia = __shfl(acc, hbase + ch), Jx = __shfl(acc, hbase + 3 + ch), Jy = __shfl(acc, hbase + 6 + ch), sw = __shfl(acc, hbase + 9);
        const float jj = Jx * Jx + Jy * Jy;
        const float sat = (float)(((double)(Ia / sw) + (double)(Jx + Jy) / (sqrt((double)jj) + (double)1.0e-20f)) + (double)0.5f);
        float v = rintf(sat);
        if (!(v >= 0.0f)) v = 0.0f;
        if (v > 255.0f) v = 255.0f;
        const uint32_t byte = (uint32_t)v;
        const uint32_t out = __shfl(byte, hbase) | (__shfl(byte, hbase + 1) << 8) | (__shfl(byte, hbase + 2) << 16);
        if (lane32 == 0) store_px_bytes(a.img + 3 * ib, (int)o, out);
    }
}

// sr:807: only the key-coloured pixels take the inpainted value, black ones go back to black; then masked_blur.
// Both in one pass: the 36 taps read the work image and zero it on the fly where the seed was black.
struct BlurSrc { const uint8_t* ibase; const uint8_t* sbase; size_t img_pitch, seed_pitch; bool masked; };

__device__ __forceinline__ uint32_t blur_px(const BlurSrc& b, int x, int y)
{
    uint32_t c = load_px_bytes(b.ibase + (size_t)y * b.img_pitch, x);
    if (b.masked && load_px_bytes(b.sbase + (size_t)y * b.seed_pitch, x) == 0u) c = 0u;
    return c;
}

// The 6 x 6 correlation around a non-black pixel (a black pixel stays black whatever surrounds it, sr:151).
__device__ __forceinline__ uint32_t masked_blur_pixel(const BlurSrc& b, int x, int y, int W, int H, const BlurKernel& K)
{
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
    uint32_t centre = 0;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        const int sy = y + ky - 3;
        if (sy < 0 || sy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const int sx = x + kx - 3;
            if (sx < 0 || sx >= W) continue;
            const uint32_t px = blur_px(b, sx, sy);
            const float k = K.k[6 * ky + kx];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)((px >> (8 * c)) & 0xFF);
            if (px) wsum = wsum + k;
            if (ky == 3 && kx == 3) centre = px;
        }
    }
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (wsum == 0.0f || centre == 0u) ? 0.0f : acc[c] / wsum;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        o |= (uint32_t)v << (8 * c);
    }
    return o;
}

__device__ __forceinline__ BlurSrc blur_src(const ImageSet& imgs, const ImageSet& seeds, int im, uint32_t key_rgb)
{
    return BlurSrc{imgs.image(im), seeds.base ? seeds.image(im) : nullptr, imgs.pitch, seeds.pitch, seeds.base != nullptr && key_rgb != 0u};
}

__global__ void __launch_bounds__(256) k_masked_blur(ImageSet imgs, ImageSet seeds, ImageSet outs, int W, int H, BlurKernel K,
                                                     uint32_t key_rgb)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= W) return;
    const BlurSrc b = blur_src(imgs, seeds, im, key_rgb);
    uint8_t* orow = outs.image(im) + (size_t)y * outs.pitch;
    if (blur_px(b, x, y) == 0u) { store_px_bytes(orow, x, 0u); return; }
    store_px_bytes(orow, x, masked_blur_pixel(b, x, y, W, H, K));
}

// The same in two passes, for images that are mostly black (an infill mask is: ~4 % of its pixels are not, in strips a few
// pixels wide -- a wave of 64 consecutive pixels that meets one runs all 36 taps for a handful of lanes): the first pass
// writes the black pixels and lists the columns of the others row by row (a counter per image row: one counter for the whole
// pass serialised 3 * 10^5 atomics on one address, 1.2 ms), the second gives every lane of a row's wave a listed pixel.
template <int PX>      // 4: rows addressable as dwords (12 bytes per lane), 1: any width / alignment
__global__ void __launch_bounds__(128) k_masked_blur_scan(ImageSet imgs, ImageSet seeds, ImageSet outs, int W, int H, uint32_t key_rgb,
                                                          uint32_t* __restrict__ list, uint32_t* __restrict__ row_count)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    uint32_t c[PX];
#pragma unroll
    for (int q = 0; q < PX; ++q) c[q] = 0u;
    const bool in = g * PX < W;
    uint8_t* orow = outs.image(im) + (size_t)y * outs.pitch;
    if (in) {
        uint8_t* irow = imgs.image(im) + (size_t)y * imgs.pitch;
        RowIO<PX>::load(irow, g, c);
        if (seeds.base && key_rgb != 0u) {
            uint32_t sd[PX];
            RowIO<PX>::load(seeds.image(im) + (size_t)y * seeds.pitch, g, sd);
#pragma unroll
            for (int q = 0; q < PX; ++q)
                if (sd[q] == 0u && c[q] != 0u) {           // an estimate nobody keeps (sr:807): black in the work image too, so that the
                    c[q] = 0u;                              // second pass reads one image per tap instead of two
                    store_px_bytes(irow, g * PX + q, 0u);
                }
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < PX; ++q) any |= c[q] != 0u;
        if (!any) { const uint32_t z[PX] = {}; RowIO<PX>::store_rgb(orow, g, z); }
        else {
#pragma unroll
            for (int q = 0; q < PX; ++q) if (c[q] == 0u) store_px_bytes(orow, g * PX + q, 0u);
        }
    }
    u64 m[PX];
    uint32_t total = 0;
#pragma unroll
    for (int q = 0; q < PX; ++q) { m[q] = __ballot(c[q] != 0u); total += (uint32_t)__popcll(m[q]); }
    if (total) {
        const size_t row = (size_t)im * H + y;
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&row_count[row], total);
        base = __shfl(base, 0);
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            if (c[q] != 0u) list[row * (size_t)W + base + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull))] = (uint32_t)(g * PX + q);
            base += (uint32_t)__popcll(m[q]);
        }
    }
}

__global__ void __launch_bounds__(64) k_masked_blur_list(ImageSet imgs, ImageSet seeds, ImageSet outs, int W, int H, BlurKernel K,
                                                         uint32_t key_rgb, const uint32_t* __restrict__ list, const uint32_t* __restrict__ row_count)
{
    const int y = blockIdx.x, im = blockIdx.y;
    const size_t row = (size_t)im * H + y;
    const uint32_t n = row_count[row];
    if (n == 0u) return;
    const BlurSrc b{imgs.image(im), nullptr, imgs.pitch, 0, false};          // (the scan pass has merged the seed's black pixels into the work image)
    uint8_t* orow = outs.image(im) + (size_t)y * outs.pitch;
    for (uint32_t k = threadIdx.x; k < n; k += 64) {
        const int x = (int)list[row * (size_t)W + k];
        store_px_bytes(orow, x, masked_blur_pixel(b, x, y, W, H, K));
    }
}

size_t telea_counter_words(int max_rounds) { return (2 + (size_t)kNcStride) * ((size_t)max_rounds + 2); }

static TeleaArgs telea_args(const TeleaWorkspace& ws, int n, int W, int H, uint32_t key_rgb)
{
    return TeleaArgs{ws.stamp, ws.T, ws.img, ws.need, ws.nlist, ws.counts, ws.offs, ws.ncounts, ws.remaining, ws.last_round, W, H, n, key_rgb};
}

// Per-call part: reset the counters, copy the seeds into the work image, pass A (levels by distance transform, level
// lists by counting sort).  h_levels (pinned host word) receives the deepest level any image needs -- the call waits for
// it, once per pass, so that passes B and C can be launched for exactly the levels that exist.
hipError_t launch_telea_init(const ImageSet& seed, const TeleaWorkspace& ws, int n, int W, int H, int max_rounds, uint32_t key_rgb,
                             uint32_t* h_levels, hipStream_t s)
{
    const TeleaArgs a = telea_args(ws, n, W, H, key_rgb);
    hipError_t e = hipMemsetAsync(ws.remaining, 0, (size_t)kTeleaMaxImages * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(ws.last_round, 0, (size_t)kTeleaMaxImages * sizeof(uint32_t), s)) != hipSuccess) return e;
    e = hipMemsetAsync(ws.counts, 0, (2 + (size_t)kNcStride) * ((size_t)max_rounds + 2) * sizeof(uint32_t), s);      // counts, offs and ncounts (adjacent)
    if (e != hipSuccess) return e;
    if (W % 4 == 0 && (((uintptr_t)seed.base | seed.pitch | seed.stride | (size_t)seed.eye_offset) & 3) == 0)
        hipLaunchKernelGGL(k_telea_init<4>, dim3((W / 4 + 127) / 128, H, n), dim3(128), 0, s, seed, a);
    else
        hipLaunchKernelGGL(k_telea_init<1>, dim3((W + 127) / 128, H, n), dim3(128), 0, s, seed, a);
    if (W % 8 == 0 && W <= 2048) hipLaunchKernelGGL(k_telea_dt_rows_vec<1>, dim3(H, n), dim3(256), 0, s, ws.stamp, W, H);
    else if (W % 8 == 0 && W <= 4096) hipLaunchKernelGGL(k_telea_dt_rows_vec<2>, dim3(H, n), dim3(256), 0, s, ws.stamp, W, H);
    else hipLaunchKernelGGL(k_telea_dt_rows, dim3(H, n), dim3(256), 0, s, ws.stamp, W, H);
    if (H <= 16 * 68) hipLaunchKernelGGL(k_telea_dt_cols_reg<68>, dim3((W + 63) / 64, n), dim3(1024), 0, s, a, (uint32_t)max_rounds);
    else hipLaunchKernelGGL(k_telea_dt_cols, dim3((W + 63) / 64, n), dim3(1024), 0, s, a, (uint32_t)max_rounds);
    hipLaunchKernelGGL(k_telea_rmax, dim3(1), dim3(1), 0, s, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    int R = max_rounds;
    if (h_levels) {       // (NULL: the asynchronous form -- every level up to max_rounds gets its launches; the ones that do not exist have empty lists)
        if ((e = hipMemcpyAsync(h_levels, ws.counts, sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        R = (int)*h_levels;
        if (R == 0) return hipSuccess;
    }
    if ((e = hipMemsetAsync(ws.counts, 0, sizeof(uint32_t), s)) != hipSuccess) return e;         // counts[0] carried R; level 0 is empty
    const dim3 grid_s((unsigned)(((W + kSortTile - 1) / kSortTile) * ((H + kSortTile - 1) / kSortTile)), n);
    hipLaunchKernelGGL(k_telea_scan, dim3(1), dim3(1024), 0, s, a, R);
    hipLaunchKernelGGL(k_telea_sort, grid_s, dim3(256), 0, s, a);
    return hipGetLastError();
}

// Passes B and C: one launch per existing level each (a captured HIP graph replays them no faster: the ~4 us between
// dependent kernels is the device's, not the host's; one cooperative launch with grid-wide barriers is 3 x slower).
hipError_t launch_telea_rounds(const TeleaWorkspace& ws, int W, int H, int levels, uint32_t key_rgb, hipStream_t s)
{
    const TeleaArgs a = telea_args(ws, kTeleaMaxImages, W, H, key_rgb);
    // both passes wait on memory, not on arithmetic (PMC: `need` spends 90 % of its wave cycles waiting): a grid large enough
    // for one entry per thread takes 7.1 -> 5.8 ms off a 32-image pass compared with 512 workgroups looping
    int nb = 2048;
    if (const char* e = tuning_env(TUNE_TELEA_BLOCKS)) { const int v = atoi(e); if (v > 0) nb = (v + 7) & ~7; }      // tuning hook (a multiple of 8: XCDs)
    const dim3 grid(nb), block(256);
    for (int r = levels; r >= 2; --r) hipLaunchKernelGGL(k_telea_need, grid, block, 0, s, a, (uint32_t)r);
    for (int r = 1; r <= levels; ++r) hipLaunchKernelGGL(k_telea_fill, dim3(4 * nb), block, 0, s, a, (uint32_t)r);
    if (tuning_env(TUNE_TELEA_DUMP)) {         // tuning hook: level sizes / needed pixels of this pass on stderr
        std::vector<uint32_t> c(levels + 2), nc((size_t)(levels + 2) * kNcStride);
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipMemcpy(c.data(), ws.counts, c.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(nc.data(), ws.ncounts, nc.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        for (int r = 1; r <= levels; ++r) fprintf(stderr, "level %d count %u need %u\n", r, c[r], nc[(size_t)r * kNcStride]);
    }
    return hipGetLastError();
}

hipError_t launch_masked_blur(const ImageSet& img, const ImageSet* seed, const ImageSet& out, int n, int W, int H,
                              const BlurKernel& K, uint32_t key_rgb, hipStream_t s, uint32_t* list, uint32_t* count)
{
    const dim3 grid((W + 255) / 256, H, n), block(256);
    const ImageSet none{nullptr, 0, 0, 0, 1};
    if (list && count && tuning_env(TUNE_BLUR_ONE_PASS) == nullptr) {          // list: n * W * H entries, count: n * H row counters
        hipError_t e = hipMemsetAsync(count, 0, (size_t)n * H * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        auto dwords = [](const ImageSet& i) { return !i.base || (((uintptr_t)i.base | i.pitch | i.stride | (size_t)i.eye_offset) & 3) == 0; };
        if (W % 4 == 0 && dwords(img) && dwords(out) && (!seed || dwords(*seed)))
            hipLaunchKernelGGL(k_masked_blur_scan<4>, dim3((W / 4 + 127) / 128, H, n), dim3(128), 0, s, img, seed ? *seed : none, out, W, H, key_rgb, list, count);
        else
            hipLaunchKernelGGL(k_masked_blur_scan<1>, dim3((W + 127) / 128, H, n), dim3(128), 0, s, img, seed ? *seed : none, out, W, H, key_rgb, list, count);
        hipLaunchKernelGGL(k_masked_blur_list, dim3(H, n), dim3(64), 0, s, img, seed ? *seed : none, out, W, H, K, key_rgb, list, count);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_masked_blur, grid, block, 0, s, img, seed ? *seed : none, out, W, H, K, key_rgb);
    return hipGetLastError();
}

// cv2.cvtColor(BGR2RGB / RGB2BGR) of interleaved u8 frames (sr:493, 505, 928, 941): bytes 0 and 2 of every pixel swap.
template <int PX>
__global__ void __launch_bounds__(256) k_swap_rb(ImageSet src, ImageSet dst, int W, int H)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (g >= W / PX) return;
    uint32_t px[PX];
    RowIO<PX>::load_nt(src.image(im) + (size_t)y * src.pitch, g, px);
#pragma unroll
    for (int q = 0; q < PX; ++q) px[q] = (px[q] & 0x00FF00u) | ((px[q] >> 16) & 0xFFu) | ((px[q] & 0xFFu) << 16);
    RowIO<PX>::store_rgb(dst.image(im) + (size_t)y * dst.pitch, g, px);
}

hipError_t launch_swap_rb(const ImageSet& src, const ImageSet& dst, int n, int W, int H, hipStream_t s)
{
    const bool vec4 = W % 4 == 0 && ((uintptr_t)src.base % 4 == 0) && ((uintptr_t)dst.base % 4 == 0) && src.pitch % 4 == 0 &&
                      dst.pitch % 4 == 0 && src.stride % 4 == 0 && dst.stride % 4 == 0;
    if (vec4) hipLaunchKernelGGL((k_swap_rb<4>), dim3((W / 4 + 255) / 256, H, n), dim3(256), 0, s, src, dst, W, H);
    else hipLaunchKernelGGL((k_swap_rb<1>), dim3((W + 255) / 256, H, n), dim3(256), 0, s, src, dst, W, H);
    return hipGetLastError();
}

#endif  // grid-independent sections

namespace MDVT_GRID {
// =================================================================================================
// launch plumbing
// =================================================================================================

constexpr size_t kMaxLds = 160 * 1024;

bool render_fits_lds(const RenderPlan& plan, int W)
{
    RenderPlan p = plan;
    p.general = 0;
    return render_lds_bytes(p, W) <= kMaxLds;
}

static size_t mesh_rows_tie_bytes(int W) { return (((size_t)W + 31) / 32 + 1) * sizeof(uint32_t) + 32; }     // RowTies + alignment slack

size_t render_lds_bytes(const RenderPlan& plan, int W)
{
    if (plan.general) return 0;
    if (plan.mode == MDVT_MODE_POINTS) {
        size_t b = (2 * (size_t)W + 2) * sizeof(u64) + 64;       // + the trash word and the per-wave counters of the fast kernel
        if (plan.edge_points) b += 2 * (size_t)W * sizeof(uint32_t);
        return b;
    }
    const size_t extra = (size_t)W * sizeof(u64) + (plan.edge_points ? (size_t)W * sizeof(uint32_t) + 2 * (size_t)W + 16 : 0) +
                         (plan.remove_edges ? (((size_t)W + 15) & ~(size_t)15) : 0) + mesh_rows_tie_bytes(W);
    size_t b = 2 * (size_t)W * 16 + extra;              // 16-byte vertices (colour in LDS)
    if (b > kMaxLds) b = 2 * (size_t)W * 12 + extra;    // 12-byte vertices (colour re-read in the resolve)
    return b;
}

template <int PX, int TPB, int ITERS>
static hipError_t launch_points_rows_cfg(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const size_t lds = render_lds_bytes(plan, a.W);
    const dim3 grid((unsigned)(plan.n * a.H)), block(TPB);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0) |
                      (plan.remove_edges && a.seed[0] ? 8 : 0);
#define MDVT_CASE(F)                                                                                   \
    case F:                                                                                            \
        if (lds > 48 * 1024)                                                                           \
            (void)hipFuncSetAttribute((const void*)k_points_rows<PX, F, TPB, ITERS>,                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        hipLaunchKernelGGL((k_points_rows<PX, F, TPB, ITERS>), grid, block, lds, s, a);                \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        MDVT_CASE(10) MDVT_CASE(11) MDVT_CASE(14) MDVT_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

// Tuning override for experiments (tools/kbench.py): MDVT_POINTS_CFG = "<TPB>x<ITERS>" (e.g. 512x1) selects the
// templated kernel with that geometry instead of the defaults.  Re-read on every launch.
static int points_cfg_override()
{
    const char* e = tuning_env(TUNE_POINTS_CFG);
    int t = 0, it = 0;
    if (e && sscanf(e, "%dx%d", &t, &it) == 2) return t * 16 + it;
    return 0;
}

template <int TPB, bool ZOUT, int BITS>
static hipError_t launch_points_rows_fast_cfg(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);      // (tuning build: the BITS tail's ablations, 8 / 16 / 32 / 64)
    const size_t lds = (2 * (size_t)a.W + 2) * sizeof(u64);
    const dim3 grid((unsigned)(plan.n * a.H)), block(TPB);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k_points_rows_fast<TPB, ZOUT, BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (!ZOUT && !BITS && lds <= 48 * 1024) {       // tuning hook (tools/kbench.py --env MDVT_POINTS_NT): 0 = temporal loads and stores
        const char* e = tuning_env(TUNE_POINTS_NT);
        if (e && *e) {
            const int nt = atoi(e);
            if (nt == 0) { hipLaunchKernelGGL((k_points_rows_fast<TPB, false, 0, 0>), grid, block, lds, s, a); return hipGetLastError(); }
            if (nt == 1) { hipLaunchKernelGGL((k_points_rows_fast<TPB, false, 0, 1>), grid, block, lds, s, a); return hipGetLastError(); }
            if (nt == 2) { hipLaunchKernelGGL((k_points_rows_fast<TPB, false, 0, 2>), grid, block, lds, s, a); return hipGetLastError(); }
        }
    }
    hipLaunchKernelGGL((k_points_rows_fast<TPB, ZOUT, BITS>), grid, block, lds, s, a);
    if (BITS && a.hole_counts && !(MDVT_DEBUG_SKIP(a) & 16)) {
        hipLaunchKernelGGL(k_reduce_wave_counts, dim3(plan.n), dim3(256), 0, s, a.wave_counts, a.hole_counts, a.H, TPB / 64, a.frame0);
    }
    return hipGetLastError();
}

template <int TPB>
static hipError_t launch_points_rows_fast(RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const bool zout = a.zout[0] || a.zout[1];
    const bool pack = a.maskbits[0] != nullptr, count = a.hole_counts != nullptr, mask = a.mask[0] != nullptr;
    const bool bits = pack || count;
    plan.fused_bits = bits;
    if (zout) return bits ? launch_points_rows_fast_cfg<TPB, true, 1>(plan, a, s) : launch_points_rows_fast_cfg<TPB, true, 0>(plan, a, s);
    if (!bits) return launch_points_rows_fast_cfg<TPB, false, 0>(plan, a, s);
    if (tuning_env(TUNE_DEBUG_SKIP)) return launch_points_rows_fast_cfg<TPB, false, 1>(plan, a, s);      // (the ablations live in the generic form)
    switch ((mask ? 1 : 0) | (pack ? 2 : 0) | (count ? 4 : 0)) {           // the set of outputs fixed at compile time
        case 2: return launch_points_rows_fast_cfg<TPB, false, 8 | 2>(plan, a, s);
        case 3: return launch_points_rows_fast_cfg<TPB, false, 8 | 3>(plan, a, s);
        case 6: return launch_points_rows_fast_cfg<TPB, false, 8 | 6>(plan, a, s);
        case 7: return launch_points_rows_fast_cfg<TPB, false, 8 | 7>(plan, a, s);
        case 5: return launch_points_rows_fast_cfg<TPB, false, 8 | 5>(plan, a, s);
        default: return launch_points_rows_fast_cfg<TPB, false, 1>(plan, a, s);
    }
}

// Will this launch be rendered by k_points_rows_fast with the mask compaction fused in (the one kernel that can leave the byte
// mask out)?  The same conditions as the dispatch below.
bool points_fused_bits_applies(const RenderPlan& plan, const RenderArgs& a)
{
    return plan.mode == MDVT_MODE_POINTS && !plan.general && plan.vec4 && !plan.remove_edges && points_cfg_override() == 0 &&
           a.W / 4 <= 1024 && (a.maskbits[0] || a.maskbits[1]);
}

static hipError_t launch_points_rows_vec4(RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const int ngroups = a.W / 4;
    int cfg = points_cfg_override();
    if (cfg == 0 && !plan.remove_edges) {          // the headline kernel
        if (ngroups <= 256) return launch_points_rows_fast<256>(plan, a, s);
        if (ngroups <= 512) return launch_points_rows_fast<512>(plan, a, s);
        if (ngroups <= 1024) return launch_points_rows_fast<1024>(plan, a, s);
    }
    if (cfg == 0 || (cfg % 16 != 0 && (cfg / 16) * (cfg % 16) < ngroups)) {
        if (ngroups <= 256) cfg = 256 * 16 + 1;
        else if (ngroups <= 512) cfg = 512 * 16 + 1;        // 1080p: measured best (tools/kbench.py)
        else if (ngroups <= 1024) cfg = 512 * 16 + 2;
        else if (ngroups <= 2048) cfg = 1024 * 16 + 2;
        else cfg = 256 * 16 + 0;
    }
#define MDVT_CFG(T, I) case T * 16 + I: return launch_points_rows_cfg<4, T, I>(plan, a, s);
    switch (cfg) {
        MDVT_CFG(128, 4) MDVT_CFG(256, 1) MDVT_CFG(256, 2) MDVT_CFG(512, 1) MDVT_CFG(512, 2) MDVT_CFG(1024, 2)
        default: return launch_points_rows_cfg<4, 256, 0>(plan, a, s);
    }
#undef MDVT_CFG
}

// The global key buffers are EMPTY on entry: the host clears them when they are (re)allocated and every
// resolve pass leaves them EMPTY again.
static hipError_t launch_points_general(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const bool edge = plan.remove_edges && plan.edge_points;
    hipError_t e;
    const dim3 block(256);
    const dim3 grid_s((a.W + 255) / 256, a.H, plan.n);
    const int sflags = (plan.remove_edges ? 2 : 0) | (edge ? 4 : 0);
    switch (sflags) {
        case 0: hipLaunchKernelGGL((k_points_splat_general<0>), grid_s, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_points_splat_general<2>), grid_s, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_points_splat_general<6>), grid_s, block, 0, s, a); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (plan.after_vertices && (e = hipEventRecord(plan.after_vertices, s)) != hipSuccess) return e;     // (banks: the next set may start)
    return launch_resolve_general<false>(plan, a, s);
}

template <int PX, int TPB>
static hipError_t launch_mesh_rows_tpb(const RenderPlan& plan, const RenderArgs& a, size_t lds, bool vrgb, hipStream_t s)
{
    const dim3 grid((unsigned)(plan.n * a.H)), block(TPB);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0) |
                      (plan.remove_edges && a.seed[0] ? 8 : 0);
#define MDVT_CASE(F)                                                                                   \
    case F:                                                                                            \
        if (vrgb) {                                                                                    \
            if (lds > 48 * 1024)                                                                       \
                (void)hipFuncSetAttribute((const void*)k_mesh_rows<PX, F, TPB, true>,                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            hipLaunchKernelGGL((k_mesh_rows<PX, F, TPB, true>), grid, block, lds, s, a);               \
        } else {                                                                                       \
            (void)hipFuncSetAttribute((const void*)k_mesh_rows<PX, F, TPB, false>,                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            hipLaunchKernelGGL((k_mesh_rows<PX, F, TPB, false>), grid, block, lds, s, a);              \
        }                                                                                              \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        MDVT_CASE(10) MDVT_CASE(11) MDVT_CASE(14) MDVT_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

template <int PX>
static hipError_t launch_mesh_rows(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);
    const size_t lds = render_lds_bytes(plan, a.W);
    if (lds > kMaxLds) return hipErrorNotSupported;         // W > ~4300 with edge points (5120 without)
    const bool vrgb = lds == 2 * (size_t)a.W * 16 + (size_t)a.W * 8 + (plan.edge_points ? (size_t)a.W * 4 + 2 * (size_t)a.W + 16 : 0) +
                             (plan.remove_edges ? (((size_t)a.W + 15) & ~(size_t)15) : 0) + mesh_rows_tie_bytes(a.W);
    // two 512-thread workgroups per CU when two fit in the 160 KB LDS, otherwise one 1024-thread workgroup:
    // either way 16 waves per CU
    int tpb = (2 * lds <= kMaxLds) ? 512 : 1024;
    if (const char* e = tuning_env(TUNE_MESH_TPB)) tpb = atoi(e);
    if (tpb == 1024) return launch_mesh_rows_tpb<PX, 1024>(plan, a, lds, vrgb, s);
    return launch_mesh_rows_tpb<PX, 512>(plan, a, lds, vrgb, s);
}

// mesh mode: the vertices of removed triangles into the global edge keys (sr:589-606)
hipError_t launch_edge_points_splat(const RenderArgs& a, int n, bool as_list, bool counters_zeroed, hipStream_t s)
{
    if (a.W % 4 == 0 && a.vlist && as_list) {
        if (!counters_zeroed) {                              // (the general path's k_mesh_queue_reset has done it on its way)
            hipError_t e = hipMemsetAsync(a.vlist_count, 0, (size_t)n * sizeof(uint32_t), s);
            if (e != hipSuccess) return e;
        }
        const uint32_t ndw = (uint32_t)a.W * (uint32_t)a.H / 4u, per = (uint32_t)(kListTPB * kListDwords);
        hipLaunchKernelGGL(k_edge_vertices_list, dim3((ndw + per - 1) / per, n), dim3(kListTPB), 0, s, a);
        hipLaunchKernelGGL(k_edge_points_splat_list, dim3(256, n), dim3(256), 0, s, a);
    } else if (a.W % 4 == 0) {
        const dim3 grid_s((a.H + kSplatRows - 1) / kSplatRows, n);
        hipLaunchKernelGGL(k_edge_points_splat4, grid_s, dim3(256), 0, s, a);
    } else {
        const dim3 grid_s((a.W + 255) / 256, a.H, n);
        hipLaunchKernelGGL((k_points_splat_general<14>), grid_s, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

static hipError_t launch_mesh_general(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);
    const bool edge = plan.remove_edges && plan.edge_points;
    hipError_t e;
    if ((e = launch_mesh_raster_general(plan, a, s)) != hipSuccess) return e;
    if (edge && (e = launch_edge_points_splat(a, plan.n, plan.conv_raster != 0, true, s)) != hipSuccess) return e;
    return launch_resolve_general<true>(plan, a, s);
}

// Edge points of the scanlines the LDS row kernels leave out (pure-shift frames, scanlines erow_lo .. erow_hi of the frame's
// FrameDev): there the chain puts a point of source row i on row i or i + 1, so scanline y collects the vertices of removed
// triangles of source rows y - 1 and y whose chain row is y -- nearest first (depth code; then the lower source index) --
// and paints them where the render left a hole (sr:776, 813-814), after the row kernel wrote the scanline.  One workgroup
// per (frame, scanline of the range, eye); W x 8 B of LDS.
template <bool MESH>
__global__ void __launch_bounds__(1024) k_edge_rows_exact(RenderArgs a, int rows_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* ek = (u64*)smem;
    const int W = a.W, H = a.H, tid = threadIdx.x;
    const int fr = blockIdx.x / rows_max, r = blockIdx.x - fr * rows_max;
    const int eye = blockIdx.y;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    if (fp.erow_lo >= fp.erow_hi) return;
    const int y = fp.erow_lo + r;
    if (y > fp.erow_hi || y >= H) return;
    for (int x = tid; x < W; x += 1024) ek[x] = kEmpty64;
    __syncthreads();
    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
    for (int si = y - 1; si <= y; ++si) {
        if (si < 0) continue;
        const uint8_t* urow = a.unused + (size_t)fr * a.ws_stride_px + (size_t)si * W;
        const uint8_t* drow = dbase + (size_t)si * a.depth_pitch;
        for (int j = tid; j < W; j += 1024) {
            if (!urow[j]) continue;
            const uint32_t code = code16_of(load_px_bytes(drow, j));
            const float z = decode_z(code, fp.mult, fp.scale);
            EdgePx ep;
            edge_point_pixels(fp, W, H, si, j, MESH ? 1 : 0, z, ep);
            if (ep.ok[eye] && ep.y[eye] == y)
                atomicMin(&ek[ep.x[eye]], ((u64)code << 17) | ((u64)(uint32_t)(si - (y - 1)) << 16) | (u64)(uint32_t)j);
        }
    }
    __syncthreads();
    const uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch;
    uint8_t* orow = a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)y * a.rgb_pitch;
    for (int x = tid; x < W; x += 1024) {
        const u64 k = ek[x];
        if (k == kEmpty64 || !mrow[x]) continue;
        const int sj = (int)(k & 0xFFFFu), si = y - 1 + (int)((k >> 16) & 1u);
        if (a.edge_paint) store_px_bytes(orow, x, load_px_bytes(a.color + (size_t)f * a.color_stride + (size_t)si * a.color_pitch, sj));
        if (a.seed[eye])
            store_px_bytes(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)y * a.seed_pitch, x, edge_normal_colour(a, fp, f, eye, si, sj, MESH ? 1 : 0));
    }
}

static hipError_t launch_edge_rows_exact(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    if (!(plan.remove_edges && plan.edge_points) || plan.edge_rows_max <= 0) return hipSuccess;
    const dim3 grid((unsigned)(plan.n * plan.edge_rows_max), 2u), block(1024);
    const size_t lds = (size_t)a.W * sizeof(u64);
    if (plan.mode == MDVT_MODE_MESH) hipLaunchKernelGGL(k_edge_rows_exact<true>, grid, block, lds, s, a, plan.edge_rows_max);
    else hipLaunchKernelGGL(k_edge_rows_exact<false>, grid, block, lds, s, a, plan.edge_rows_max);
    return hipGetLastError();
}

// Edge points of every OTHER scanline of a pure-shift frame (sr:589-606, 745-781, 813-814): there a point of source row y lands
// on scanline y, in the column edge_col_pure gives (f32 estimate, the reference's f64 chain inside the guard band), nearest first
// (depth code, then the lower column), and is painted where the render left a hole -- after the row kernel, which then renders
// WITHOUT edge points: inside k_mesh_band they cost 11 us per 1080p frame (keys through the z-buffer row's LDS, two more barriers
// per scanline and eye, 17 VGPRs at the 128 limit).  One workgroup per (kEdgeRowsPer scanlines, frame), both eyes; 2 x W x 4 B of LDS.
constexpr int kEdgeRowsPer = 8;           // scanlines per workgroup of k_edge_rows_pure (<= 16: hit list entries)
constexpr int kEdgeHitCap = 1024;         // entries of its hit list (a workgroup with more paints the rest where it finds them)

// what a thread of k_edge_rows_pure<., true> reads from memory for one scanline: flags, depth codes, both eyes' hole masks of its 4-column groups
template <int NIT> struct EdgeRowRegs { uint32_t u4[NIT], dw[NIT][3], m0[NIT], m1[NIT]; };
template <int NIT>
__device__ __forceinline__ void edge_row_fetch(const RenderArgs& a, int fr, int f, int y, int tid, EdgeRowRegs<NIT>& r)
{
    const int W = a.W;
    const uint8_t* urow = a.unused + (size_t)fr * a.ws_stride_px + (size_t)y * W;
    const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)y * a.depth_pitch;
    const uint8_t* mrow0 = a.mask[0] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch;
    const uint8_t* mrow1 = a.mask[1] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j0 = tid * 4 + it * 1024;
        r.u4[it] = 0; r.m0[it] = r.m1[it] = 0; r.dw[it][0] = r.dw[it][1] = r.dw[it][2] = 0;
        if (j0 < W && y < a.H) {
            r.u4[it] = *(const uint32_t*)(urow + j0);
            const uint32_t* dp = (const uint32_t*)(drow + 3 * (size_t)j0);
            r.dw[it][0] = dp[0]; r.dw[it][1] = dp[1]; r.dw[it][2] = dp[2];
            r.m0[it] = *(const uint32_t*)(mrow0 + j0); r.m1[it] = *(const uint32_t*)(mrow1 + j0);
        }
    }
}

// NIT: 4-column groups per thread (W <= 1024 NIT), 0: rows that are not dword-addressable, column by column
template <bool MESH, int NIT>
__global__ void __launch_bounds__(256, 5) k_edge_rows_pure(RenderArgs a, int rows_per)      // (5 waves per SIMD: 94 VGPRs without a spill; unhinted the compiler takes 128)
{
    constexpr bool VEC = NIT > 0;
    constexpr int NR = NIT > 0 ? NIT : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* ek = (uint32_t*)smem;                       // [2][W]: depth code << 16 | source column; EMPTY between scanlines
    const int W = a.W, H = a.H, fr = blockIdx.y, tid = threadIdx.x;
    // the workgroup's hits (a key in a hole), painted together at the end: x | source column << 16, scanline - y0 | eye << 4
    uint2* hitq = (uint2*)(ek + 2 * W);
    __shared__ uint32_t nhit;
    if (tid == 0) nhit = 0u;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const int y0 = blockIdx.x * rows_per, y1 = min(y0 + rows_per, H);          // (rows_per <= kEdgeRowsPer)
    // VEC (dword-addressable rows, W <= 4096): everything a scanline needs from memory -- flags, depth codes, both hole masks -- is
    // requested together, and a scanline ahead: taken one after the other (flags, then the flagged columns' depth, then the mask
    // under a key) by a workgroup per scanline, the round trips made this a 14 us workgroup for a microsecond of work
    EdgeRowRegs<NR> cur, nxt;
    if (VEC) edge_row_fetch(a, fr, f, y0, tid, cur);
    for (int x = tid; x < 2 * W; x += 256) ek[x] = kEmpty32;
    __syncthreads();
    const float guard = edge_col_guard(W);
    auto paint = [&](int eye, int x, int y, int sj) {                          // sr:776, 813-814: the caller has seen the hole
        if (a.edge_paint)
            store_px_bytes(a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)y * a.rgb_pitch, x,
                           load_px_bytes(a.color + (size_t)f * a.color_stride + (size_t)y * a.color_pitch, sj));
        if (a.seed[eye])
            store_px_bytes(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)y * a.seed_pitch, x, edge_normal_colour(a, fp, f, eye, y, sj, MESH ? 1 : 0));
    };
#pragma unroll 1
    for (int y = y0; y < y1; ++y) {
        if (VEC) edge_row_fetch(a, fr, f, y + 1 < y1 ? y + 1 : H, tid, nxt);      // (row H: nothing is read)
        bool any = false;
        if (!edge_row_deferred(fp, y)) {                  // (those are k_edge_rows_exact's)
            const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)y * a.depth_pitch;
            auto splat = [&](int j, uint32_t px) {
                const uint32_t code = code16_of(px);
                const float z = decode_z(code, fp.mult, fp.scale);
                if (!(z > kNear)) return;
                const float gx = (float)j * fp.sx, d = fp.dl / z;
#pragma unroll
                for (int eye = 0; eye < 2; ++eye) {
                    const int x = edge_col_pure(fp, eye, gx, z, d, W, guard);      // sr:599-600, 746
                    if (x >= 0) { atomicMin(&ek[eye * W + x], (code << 16) | (uint32_t)j); any = true; }
                }
            };
            if (VEC) {
                // the flagged columns of the thread, one bit each; ONE copy of the splat code walks them (unrolled over the 16 columns
                // the kernel was 14 000 instructions, 110 KB: every wave stalled on instruction fetch)
                uint32_t todo = 0;
#pragma unroll
                for (int it = 0; it < NR; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((cur.u4[it] >> (8 * q)) & 0xFFu) todo |= 1u << (it * 4 + q);
#pragma unroll 1
                while (todo) {
                    const int b = __ffs((int)todo) - 1, it = b >> 2, q = b & 3;
                    todo &= todo - 1u;
                    uint32_t d0 = cur.dw[0][0], d1 = cur.dw[0][1], d2 = cur.dw[0][2];
#pragma unroll
                    for (int k = 1; k < NR; ++k)
                        if (it == k) { d0 = cur.dw[k][0]; d1 = cur.dw[k][1]; d2 = cur.dw[k][2]; }
                    const u64 lo = (u64)d0 | ((u64)d1 << 32), hi = (u64)d1 | ((u64)d2 << 32);
                    const uint32_t px = (q < 2 ? (uint32_t)(lo >> (24 * q)) : (uint32_t)(hi >> (24 * q - 32))) & 0xFFFFFFu;
                    splat(tid * 4 + it * 1024 + q, px);
                }
            } else {
                const uint8_t* urow = a.unused + (size_t)fr * a.ws_stride_px + (size_t)y * W;
                for (int j = tid; j < W; j += 256)
                    if (urow[j]) splat(j, load_px_bytes(drow, j));
            }
        }
        if (__syncthreads_or((int)any)) {                 // (workgroup uniform)
            if (VEC) {
                uint32_t hits = 0;                       // bit (it * 2 + eye) * 4 + q: a key in a hole
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    const int x0 = tid * 4 + it * 1024;
                    if (x0 >= W) continue;
#pragma unroll
                    for (int eye = 0; eye < 2; ++eye) {
                        const uint32_t m = eye ? cur.m1[it] : cur.m0[it];
                        if (!m) continue;                                           // no hole among the four
                        const uint4 k4 = *(const uint4*)(ek + eye * W + x0);
                        const uint32_t kq[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (kq[q] != kEmpty32 && ((m >> (8 * q)) & 0xFFu)) hits |= 1u << ((it * 2 + eye) * 4 + q);
                    }
                }
                // the hits go to the workgroup's list and are painted when all its scanlines are through: painting is a chain of
                // dependent loads (source colour; for the seed image the 3 x 3 neighbourhood of the vertex) that a lane walking its own
                // hits one after the other waited 5 us per scanline for
#pragma unroll 1
                while (hits) {
                    const int b = __ffs((int)hits) - 1, it = b >> 3, eye = (b >> 2) & 1, q = b & 3;
                    hits &= hits - 1u;
                    const int x = tid * 4 + it * 1024 + q;
                    const int sj = (int)(ek[eye * W + x] & 0xFFFFu);
                    const uint32_t slot = atomicAdd(&nhit, 1u);
                    if (slot < (uint32_t)kEdgeHitCap) hitq[slot] = make_uint2((uint32_t)x | ((uint32_t)sj << 16), (uint32_t)(y - y0) | ((uint32_t)eye << 4));
                    else paint(eye, x, y, sj);           // (list full: at once)
                }
#pragma unroll
                for (int it = 0; it < NR; ++it) {         // (every thread leaves its own words EMPTY for the next scanline)
                    const int x0 = tid * 4 + it * 1024;
                    if (x0 >= W) continue;
                    *(uint4*)(ek + x0) = make_uint4(kEmpty32, kEmpty32, kEmpty32, kEmpty32);
                    *(uint4*)(ek + W + x0) = make_uint4(kEmpty32, kEmpty32, kEmpty32, kEmpty32);
                }
            } else {
                for (int eye = 0; eye < 2; ++eye) {
                    const uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch;
                    for (int x = tid; x < W; x += 256) {
                        const uint32_t key = ek[eye * W + x];
                        ek[eye * W + x] = kEmpty32;
                        if (key != kEmpty32 && mrow[x]) paint(eye, x, y, (int)(key & 0xFFFFu));
                    }
                }
            }
            __syncthreads();
        }
        if (VEC) cur = nxt;
    }
    __syncthreads();
    const uint32_t nh = min(nhit, (uint32_t)kEdgeHitCap);
    for (uint32_t e = tid; e < nh; e += 256) {
        const uint2 h = hitq[e];
        paint((int)(h.y >> 4), (int)(h.x & 0xFFFFu), y0 + (int)(h.y & 15u), (int)(h.x >> 16));
    }
}

static hipError_t launch_edge_rows_pure(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    // (scanlines per workgroup: 8 for a launch of 32 frames, fewer for small launches -- as launch_mesh_band's bands)
    int rows_per = (2 * plan.n * a.H + 2880) / (2 * 2880);
    rows_per = rows_per < 1 ? 1 : (rows_per > kEdgeRowsPer ? kEdgeRowsPer : rows_per);
    const dim3 grid((unsigned)((a.H + rows_per - 1) / rows_per), (unsigned)plan.n), block(256);
    const size_t lds = 2 * (size_t)a.W * sizeof(uint32_t) + (size_t)kEdgeHitCap * sizeof(uint2);
    const int nit = !(plan.vec4 && a.W <= 4096) ? 0 : a.W <= 1024 ? 1 : a.W <= 2048 ? 2 : 4;
#define MDVT_CASE(M, N) hipLaunchKernelGGL((k_edge_rows_pure<M, N>), grid, block, lds, s, a, rows_per)
    if (plan.mode != MDVT_MODE_MESH) return hipErrorInvalidValue;          // (points: see launch_render)
    if (nit == 0) MDVT_CASE(true, 0); else if (nit == 1) MDVT_CASE(true, 1); else if (nit == 2) MDVT_CASE(true, 2); else MDVT_CASE(true, 4);
#undef MDVT_CASE
    return hipGetLastError();
}

hipError_t launch_render(RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    plan.fused_bits = 0;
    if (plan.mode == MDVT_MODE_POINTS) {
        if (plan.general) return launch_points_general(plan, a, s);
        // (the edge points stay inside k_points_rows: placed afterwards, as for the mesh below, 58.5 k against 61 k frames/s at 1080p)
        const hipError_t e = plan.vec4 ? launch_points_rows_vec4(plan, a, s) : launch_points_rows_cfg<1, 256, 0>(plan, a, s);
        return e != hipSuccess ? e : launch_edge_rows_exact(plan, a, s);
    }
    if (plan.conv) return launch_mesh_conv(plan, a, s);
    if (plan.general) return launch_mesh_general(plan, a, s);
    hipError_t e;
    // pure-shift mesh frames: the row kernel renders without edge points, k_edge_rows_pure / k_edge_rows_exact place them afterwards
    // (tuning build: MDVT_EDGE_INBAND=1 keeps them inside the row kernels, as until r04, for the A/B)
    const bool edge_after = plan.remove_edges && plan.edge_points && a.W <= 65535 && tuning_env(TUNE_EDGE_INBAND) == nullptr;
    RenderPlan rp = plan;
    if (edge_after) rp.edge_points = 0;
    // (tuning build: MDVT_MESH_BAND3=0 / 1 picks k_mesh_band / k_mesh_band3 for the A/B)
    const char* b3 = tuning_env(TUNE_MESH_BAND3);
    const bool band3 = b3 ? b3[0] == '1' : false;
    if (band3 && mesh_band3_supported(rp, a) && tuning_env(TUNE_MESH_OLD) == nullptr) e = launch_mesh_band3(rp, a, s);
    else if (mesh_band_supported(rp, a) && tuning_env(TUNE_MESH_OLD) == nullptr) e = launch_mesh_band(rp, a, s);
    else e = rp.vec4 ? launch_mesh_rows<4>(rp, a, s) : launch_mesh_rows<1>(rp, a, s);
    plan.fused_bits = rp.fused_bits;
    if (e == hipSuccess && edge_after) e = launch_edge_rows_pure(plan, a, s);
    return e != hipSuccess ? e : launch_edge_rows_exact(plan, a, s);
}

}  // namespace MDVT_GRID
}  // namespace mdvt
