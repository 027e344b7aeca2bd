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

// Pure +-ipd/2 shift with K == Krender:  u = grid_x +- (fxr*ipd/2)/Z,  v = grid_y.
__device__ __forceinline__ Vert vertex_pure(float gx, float gy, float z, float d, int eye)
{
    Vert o;
    o.u = eye == 0 ? gx + d : gx - d;
    o.v = gy;
    o.z = z;
    o.ok = z > kNear;
    return o;
}

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

// ---- rasteriser pieces --------------------------------------------------------------------
__device__ __forceinline__ i64 snap(float x)
{
    x = fminf(fmaxf(x, -kSnapLimit), kSnapLimit);
    return (i64)rintf(x * (float)kSubpix);
}

__device__ __forceinline__ i64 floordiv_pos(i64 a, i64 b)   // b > 0
{
    i64 q = a / b;
    if ((a % b) < 0) --q;
    return q;
}

// floor division by the sub-pixel grid (power of two): arithmetic shift.
__device__ __forceinline__ i64 floordiv_subpix(i64 a) { return a >> 8; }
static_assert(kSubpix == 256, "floordiv_subpix assumes a 1/256 grid");

__device__ __forceinline__ bool edge_in(i64 w, i64 dx, i64 dy)
{
    if (w > 0) return true;
    if (w < 0) return false;
    return (dy < 0) || (dy == 0 && dx > 0);     // top-left rule, clockwise (y down)
}

// One triangle prepared for point-in-triangle queries on the sub-pixel grid.
struct TriSetup {
    i64 X0, Y0, X1, Y1, X2, Y2;
    i64 s;           // orientation sign; 0 = degenerate
    float fa;        // (float)|area2|
    float iz0, iz1, iz2;
};

__device__ __forceinline__ bool tri_setup(TriSetup& t, const Vert& a, const Vert& b, const Vert& c)
{
    t.s = 0;
    if (!(a.ok && b.ok && c.ok)) return false;          // near plane: whole triangle dropped
    t.X0 = snap(a.u); t.Y0 = snap(a.v);
    t.X1 = snap(b.u); t.Y1 = snap(b.v);
    t.X2 = snap(c.u); t.Y2 = snap(c.v);
    i64 area2 = (t.X1 - t.X0) * (t.Y2 - t.Y0) - (t.Y1 - t.Y0) * (t.X2 - t.X0);
    if (area2 == 0) return false;
    t.s = area2 > 0 ? 1 : -1;
    area2 *= t.s;
    t.fa = (float)area2;
    t.iz0 = 1.0f / a.z; t.iz1 = 1.0f / b.z; t.iz2 = 1.0f / c.z;
    return true;
}

// Pixel (px,py) centre against the triangle: returns true and the three q = lambda*invz weights.
__device__ __forceinline__ bool tri_sample(const TriSetup& t, i64 px, i64 py, float& q0, float& q1, float& q2)
{
    const i64 Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
    const i64 w0 = t.s * ((t.X2 - t.X1) * (Yc - t.Y1) - (t.Y2 - t.Y1) * (Xc - t.X1));
    const i64 w1 = t.s * ((t.X0 - t.X2) * (Yc - t.Y2) - (t.Y0 - t.Y2) * (Xc - t.X2));
    const i64 w2 = t.s * ((t.X1 - t.X0) * (Yc - t.Y0) - (t.Y1 - t.Y0) * (Xc - t.X0));
    if (!(edge_in(w0, t.s * (t.X2 - t.X1), t.s * (t.Y2 - t.Y1)) &&
          edge_in(w1, t.s * (t.X0 - t.X2), t.s * (t.Y0 - t.Y2)) &&
          edge_in(w2, t.s * (t.X1 - t.X0), t.s * (t.Y1 - t.Y0))))
        return false;
    const float l0 = (float)w0 / t.fa, l1 = (float)w1 / t.fa, l2 = (float)w2 / t.fa;
    q0 = l0 * t.iz0; q1 = l1 * t.iz1; q2 = l2 * t.iz2;
    return true;
}

// Perspective-correct colour of one channel, rounded half-even to u8 (decree).
__device__ __forceinline__ uint32_t shade_channel(float q0, float q1, float q2, float iz,
                                                  uint32_t c0, uint32_t c1, uint32_t c2)
{
    const float num = (q0 * (float)c0 + q1 * (float)c1) + q2 * (float)c2;
    float val = rintf(num / iz);
    if (!(val >= 0.0f)) val = 0.0f;
    if (val > 255.0f) val = 255.0f;
    return (uint32_t)val;
}

__device__ __forceinline__ uint32_t shade_px(float q0, float q1, float q2, float iz,
                                             uint32_t p0, uint32_t p1, uint32_t p2)
{
    const uint32_t r = shade_channel(q0, q1, q2, iz, p0 & 0xFF, p1 & 0xFF, p2 & 0xFF);
    const uint32_t g = shade_channel(q0, q1, q2, iz, (p0 >> 8) & 0xFF, (p1 >> 8) & 0xFF, (p2 >> 8) & 0xFF);
    const uint32_t b = shade_channel(q0, q1, q2, iz, p0 >> 16, p1 >> 16, p2 >> 16);
    return r | (g << 8) | (b << 16);
}

}  // namespace mdvt
