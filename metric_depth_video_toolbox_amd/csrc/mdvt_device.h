// mdvt_device.h -- device-side helpers shared by the kernels in mdvt_kernels.hip.
//
// Arithmetic decree: every f32 expression below is one IEEE operation per node, the
// translation unit is compiled with -ffp-contract=off, divisions are the correctly rounded ones
// (hipcc default), and nothing here may be rewritten into an FMA or a reciprocal-multiply.
#pragma once

#include "mdvt_internal.h"

namespace mdvt {

typedef unsigned long long u64;
typedef long long i64;

constexpr u64 kEmpty64 = ~0ull;
constexpr uint32_t kEmpty32 = ~0u;

// ---- interleaved u8 RGB <-> packed pixels -------------------------------------------------
// Four pixels = 12 bytes = 3 dwords (little endian):  w0 = R0 G0 B0 R1, w1 = G1 B1 R2 G2,
// w2 = B2 R3 G3 B3.  A packed pixel is R | G<<8 | B<<16.
__device__ __forceinline__ void unpack4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t (&px)[4])
{
    px[0] = w0 & 0xFFFFFFu;
    px[1] = __builtin_amdgcn_alignbit(w1, w0, 24) & 0xFFFFFFu;
    px[2] = __builtin_amdgcn_alignbit(w2, w1, 16) & 0xFFFFFFu;
    px[3] = w2 >> 8;
}

__device__ __forceinline__ void pack4(const uint32_t (&px)[4], uint32_t& w0, uint32_t& w1, uint32_t& w2)
{
    w0 = px[0] | (px[1] << 24);
    w1 = (px[1] >> 8) | (px[2] << 16);
    w2 = (px[2] >> 16) | (px[3] << 8);
}

// dfh:67-69 (bit16): high byte = R, low byte = B, G ignored.
__device__ __forceinline__ uint32_t code16_of(uint32_t px) { return ((px & 0xFFu) << 8) | (px >> 16); }

// dfh:21-23 + sr:541: f32(code << 16) is exact; one rounding for *mult, one for *scale.
__device__ __forceinline__ float decode_z(uint32_t code16, float mult, float scale)
{
    const float d = (float)(code16 << 16) * mult;
    return d * scale;
}

// A pixel of an interleaved RGB row at an arbitrary (unaligned) column, via byte loads.
__device__ __forceinline__ uint32_t load_px_bytes(const uint8_t* row, int j)
{
    const uint8_t* p = row + 3 * (size_t)j;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

__device__ __forceinline__ void store_px_bytes(uint8_t* row, int j, uint32_t px)
{
    uint8_t* p = row + 3 * (size_t)j;
    p[0] = (uint8_t)px; p[1] = (uint8_t)(px >> 8); p[2] = (uint8_t)(px >> 16);
}

// ---- vertex programme (mirrors the decree, not the oracle's source) --------------------------
struct Vert { float u, v, z; bool ok; };

__device__ __forceinline__ void camera_point(const FrameDev& f, float gx, float gy, float z, float& xc, float& yc)
{
    xc = ((gx - f.cx) * z) / f.fx;      // dmt:1127
    yc = ((gy - f.cy) * z) / f.fy;      // dmt:1128
}

__device__ __forceinline__ Vert vertex_general(const FrameDev& f, const float* M, float xc, float yc, float z)
{
    Vert o;
    const float X = ((M[0] * xc + M[1] * yc) + M[2] * z) + M[3];
    const float Y = ((M[4] * xc + M[5] * yc) + M[6] * z) + M[7];
    const float Z = ((M[8] * xc + M[9] * yc) + M[10] * z) + M[11];
    o.ok = (z > kNear) && (Z > kNear);
    o.u = (f.fxr * X) / Z + f.cxr;
    o.v = (f.fyr * Y) / Z + f.cyr;
    o.z = Z;
    return o;
}

// The frame's own arithmetic for kernels that serve both kinds of frame: the general 3x4 map, or
// for a pure-shift frame u = gx +- dl/Z, v = gy exactly as the LDS row kernels evaluate it (frames too wide for
// their LDS z-buffers are rendered by the global-key kernels and must not change a bit because of it).
__device__ __forceinline__ Vert vertex_for_eye(const FrameDev& f, int eye, float gx, float gy, float z, float xc, float yc)
{
    if (f.general) return vertex_general(f, f.M[eye], xc, yc, z);
    Vert o;
    const float d = f.dl / z;
    o.u = eye == 0 ? gx + d : gx - d;
    o.v = gy;
    o.z = z;
    o.ok = z > kNear;
    return o;
}

// Edge point (sr:599-600, 746) of vertex (i, j): screen position before rounding.
__device__ __forceinline__ Vert edge_point_for_eye(const FrameDev& f, int eye, int i, float gx, float z, float xc, float yc)
{
    if (f.general) return vertex_general(f, f.M[eye], xc * f.sW, yc * f.sH, z);
    Vert o;
    const float ex = ((gx - f.cx) * f.sW) + f.cx;
    const float d = f.dl / z;
    o.u = eye == 0 ? ex + d : ex - d;
    o.v = (float)i;                      // the exact-arithmetic row i*(1-1/H^2)+1/2 rounds to i (decree)
    o.z = z;
    o.ok = z > kNear;
    return o;
}

// ---- rasteriser pieces --------------------------------------------------------------------
// Snapped coordinates are int32 (|x| <= 2^21 px * 256 = 2^29), so every coordinate difference fits
// int32 and every product below is one 32x32->64 multiply (v_mad_i64_i32).
__device__ __forceinline__ int snap(float x)
{
    x = fminf(fmaxf(x, -kSnapLimit), kSnapLimit);
    return (int)rintf(x * (float)kSubpix);
}

// floor division by the sub-pixel grid (power of two): arithmetic shift.
__device__ __forceinline__ int floordiv_subpix(int a) { return a >> 8; }
static_assert(kSubpix == 256, "floordiv_subpix assumes a 1/256 grid");

__device__ __forceinline__ bool edge_in(i64 w, int dx, int dy)
{
    if (w > 0) return true;
    if (w < 0) return false;
    return (dy < 0) || (dy == 0 && dx > 0);     // top-left rule, clockwise (y down)
}

// One triangle prepared for point-in-triangle queries on the sub-pixel grid.  The three edge
// functions are kept in the form  w_k(X, Y) = dx_k * (Y - Yk) - dy_k * (X - Xk)  with orientation-
// normalised deltas (clockwise, y down), so a query is three 32x32->64 multiply-adds.
struct TriSetup {
    int dx0, dy0, dx1, dy1, dx2, dy2;   // v1->v2 (weight of v0), v2->v0 (v1), v0->v1 (v2)
    int bx0, by0, bx1, by1, bx2, by2;   // base point of each edge: v1, v2, v0
    int minX, maxX, minY, maxY;         // snapped bounding box
    i64 area2;                          // |area2|; 0 = degenerate / dropped
    float iz0, iz1, iz2;
};

__device__ __forceinline__ i64 mul64(int a, int b) { return (i64)a * (i64)b; }
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

// iz_k = 1/Z_k of the three vertices; a vertex behind the near plane is flagged by iz == 0.
__device__ __forceinline__ bool tri_setup_snapped(TriSetup& t, int X0, int Y0, float iz0, int X1, int Y1, float iz1,
                                                  int X2, int Y2, float iz2)
{
    t.area2 = 0;
    if (!(iz0 > 0.0f && iz1 > 0.0f && iz2 > 0.0f)) return false;    // near plane: whole triangle dropped
    const i64 a2 = mul64(X1 - X0, Y2 - Y0) - mul64(Y1 - Y0, X2 - X0);
    if (a2 == 0) return false;
    const bool neg = a2 < 0;
    t.area2 = neg ? -a2 : a2;
    t.dx0 = neg ? X1 - X2 : X2 - X1; t.dy0 = neg ? Y1 - Y2 : Y2 - Y1;
    t.dx1 = neg ? X2 - X0 : X0 - X2; t.dy1 = neg ? Y2 - Y0 : Y0 - Y2;
    t.dx2 = neg ? X0 - X1 : X1 - X0; t.dy2 = neg ? Y0 - Y1 : Y1 - Y0;
    t.bx0 = X1; t.by0 = Y1; t.bx1 = X2; t.by1 = Y2; t.bx2 = X0; t.by2 = Y0;
    t.minX = min3i(X0, X1, X2); t.maxX = max3i(X0, X1, X2);
    t.minY = min3i(Y0, Y1, Y2); t.maxY = max3i(Y0, Y1, Y2);
    t.iz0 = iz0; t.iz1 = iz1; t.iz2 = iz2;
    return true;
}

__device__ __forceinline__ bool tri_setup(TriSetup& t, const Vert& a, const Vert& b, const Vert& c)
{
    return tri_setup_snapped(t, snap(a.u), snap(a.v), a.ok ? 1.0f / a.z : 0.0f, snap(b.u), snap(b.v),
                             b.ok ? 1.0f / b.z : 0.0f, snap(c.u), snap(c.v), c.ok ? 1.0f / c.z : 0.0f);
}

// i64 -> f32, round to nearest even.  When the value fits int32 the single-instruction conversion
// gives the same correctly rounded result.
__device__ __forceinline__ float i64_to_f32(i64 w, bool fits32) { return fits32 ? (float)(int)w : (float)w; }

// q_k = lambda_k * (1/Z_k) with lambda_k = f32(w_k) * (1/f32(area2)): one division per covered pixel.
__device__ __forceinline__ void tri_weights(i64 area2, float iz0, float iz1, float iz2, i64 w0, i64 w1, i64 w2,
                                            float& q0, float& q1, float& q2)
{
    const bool small = area2 < 0x7FFFFFFFll;                              // 0 <= w_k <= area2 inside
    float fa, f0, f1, f2;
    if (__ballot(!small) == 0ull) {
        // every active lane's triangle is small (the rule, except rubber-sheet triangles): one v_cvt_f32_i32 each.  A
        // per-lane select would make the compiler evaluate the ~10-instruction 64-bit conversion as well, four times.
        fa = (float)(int)area2; f0 = (float)(int)w0; f1 = (float)(int)w1; f2 = (float)(int)w2;
    } else {
        fa = i64_to_f32(area2, small); f0 = i64_to_f32(w0, small); f1 = i64_to_f32(w1, small); f2 = i64_to_f32(w2, small);
    }
    const float ra = 1.0f / fa;
    const float l0 = f0 * ra, l1 = f1 * ra, l2 = f2 * ra;
    q0 = l0 * iz0; q1 = l1 * iz1; q2 = l2 * iz2;
}

// Pixel (px,py) centre against the triangle: returns true and the three q = lambda*invz weights.
__device__ __forceinline__ bool tri_sample(const TriSetup& t, int px, int py, float& q0, float& q1, float& q2)
{
    const int Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
    const i64 w0 = mul64(t.dx0, Yc - t.by0) - mul64(t.dy0, Xc - t.bx0);
    const i64 w1 = mul64(t.dx1, Yc - t.by1) - mul64(t.dy1, Xc - t.bx1);
    const i64 w2 = mul64(t.dx2, Yc - t.by2) - mul64(t.dy2, Xc - t.bx2);
    if (!(edge_in(w0, t.dx0, t.dy0) && edge_in(w1, t.dx1, t.dy1) && edge_in(w2, t.dx2, t.dy2))) return false;
    tri_weights(t.area2, t.iz0, t.iz1, t.iz2, w0, w1, w2, q0, q1, q2);
    return true;
}

// The same for a walk over a small pixel box: the three edge values at one pixel centre, advanced by exact integer
// steps (one pixel right: w_k -= 256 dy_k; one pixel down: w_k += 256 dx_k) instead of six products per pixel.
struct TriWalk { i64 w0, w1, w2; };
__device__ __forceinline__ TriWalk tri_walk_start(const TriSetup& t, int px, int py)
{
    const int Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
    return TriWalk{mul64(t.dx0, Yc - t.by0) - mul64(t.dy0, Xc - t.bx0), mul64(t.dx1, Yc - t.by1) - mul64(t.dy1, Xc - t.bx1),
                   mul64(t.dx2, Yc - t.by2) - mul64(t.dy2, Xc - t.bx2)};
}
__device__ __forceinline__ void tri_walk_right(const TriSetup& t, TriWalk& w)
{
    w.w0 -= (i64)t.dy0 * kSubpix; w.w1 -= (i64)t.dy1 * kSubpix; w.w2 -= (i64)t.dy2 * kSubpix;
}
__device__ __forceinline__ void tri_walk_down(const TriSetup& t, TriWalk& w)
{
    w.w0 += (i64)t.dx0 * kSubpix; w.w1 += (i64)t.dx1 * kSubpix; w.w2 += (i64)t.dx2 * kSubpix;
}
__device__ __forceinline__ bool tri_walk_sample(const TriSetup& t, const TriWalk& w, float& q0, float& q1, float& q2)
{
    if (!(edge_in(w.w0, t.dx0, t.dy0) && edge_in(w.w1, t.dx1, t.dy1) && edge_in(w.w2, t.dx2, t.dy2))) return false;
    tri_weights(t.area2, t.iz0, t.iz1, t.iz2, w.w0, w.w1, w.w2, q0, q1, q2);
    return true;
}

// Perspective-correct colour, rounded half-even to u8 (decree): rint(((q0 c0 + q1 c1) + q2 c2) * (1/iz)).
__device__ __forceinline__ uint32_t shade_channel(float q0, float q1, float q2, float riz,
                                                  uint32_t c0, uint32_t c1, uint32_t c2)
{
    const float num = (q0 * (float)c0 + q1 * (float)c1) + q2 * (float)c2;
    float val = rintf(num * riz);
    if (!(val >= 0.0f)) val = 0.0f;
    if (val > 255.0f) val = 255.0f;
    return (uint32_t)val;
}

__device__ __forceinline__ uint32_t shade_px(float q0, float q1, float q2, float riz,
                                             uint32_t p0, uint32_t p1, uint32_t p2)
{
    const uint32_t r = shade_channel(q0, q1, q2, riz, p0 & 0xFF, p1 & 0xFF, p2 & 0xFF);
    const uint32_t g = shade_channel(q0, q1, q2, riz, (p0 >> 8) & 0xFF, (p1 >> 8) & 0xFF, (p2 >> 8) & 0xFF);
    const uint32_t b = shade_channel(q0, q1, q2, riz, p0 >> 16, p1 >> 16, p2 >> 16);
    return r | (g << 8) | (b << 16);
}

}  // namespace mdvt
