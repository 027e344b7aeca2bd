// mdvt_band_common.h -- the per-cell arithmetic of the pure-shift mesh band kernels (mdvt_mesh_band.hip: vertex rows in
// an LDS ring, one eye at a time, two workgroups per CU; mdvt_mesh_band3.hip: no vertex rows in LDS, both eyes per pass,
// three).  See the header of mdvt_mesh_band.hip for the scanline-interval formulation these helpers implement.
#pragma once

#include "mdvt_device.h"

namespace mdvt {
namespace {

constexpr int kQueueWave = 128;        // (cell, pixel) items per wave: pushes of <= 64 followed by a drain keep it < 128
constexpr int kCoordBound = 1 << 20;   // |X| of the fast path (sub-pixels)

struct BandVert { int XL, XR; float iz; uint32_t rgb; };

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// One vertex of the grid: decode, 1/Z and dl/Z (correctly rounded), snapped x of both eyes.
// (cpx may carry per-vertex flags in its top byte: they ride along in the record's colour word)
__device__ __forceinline__ int4 band_vertex(uint32_t dpx, uint32_t cpx, int j, const FrameDev& fp, bool dl_ok)
{
    const float z = decode_z(code16_of(dpx), fp.mult, fp.scale);
    const bool ok = z > kNear;
    float iz, d;
    rcp_div_exact(fp.dl, dl_ok, z, iz, d);
    const float gx = (float)j * fp.sx;
    return make_int4(snap(gx + d), snap(gx - d), __float_as_int(ok ? iz : 0.0f), (int)cpx);
}

// P(k) = ceil((k - 128 hh) / D), D = 256 hh, clamped to [0, W].  c128 = 128 hh - 1, rD ~ 1/D (any f32 close to it: the
// estimate is corrected by one exact step; |estimate - quotient| < 1 because quotient < 2^15 and the three roundings
// are 2^-24 relative each).
__device__ __forceinline__ int first_pixel(int k, int c128, int D, float rD, int W)
{
    int n = k + c128;
    n = n < 0 ? 0 : n;
    int q = (int)((float)n * rD);
    const int rem = n - mul24(q, D);
    q += rem < 0 ? -1 : (rem >= D ? 1 : 0);
    return q > W ? W : q;
}

struct RowGeom {                 // uniform per output row
    int c;                       // cell row covering the scanline, -1: none
    int Yt, Yb, tt, bb, hh;
    int D, c128;
    float rD;
};

__device__ __forceinline__ RowGeom row_geometry(int k, const RowCell* __restrict__ table)
{
    RowGeom g;
    const RowCell r = table[k];                // uniform: scalar loads
    const int Yc = k * kSubpix + kSubpix / 2;
    g.c = r.c; g.Yt = r.Yt; g.Yb = r.Yb;
    g.tt = Yc - g.Yt; g.bb = g.Yb - Yc; g.hh = g.Yb - g.Yt;
    g.D = g.hh * kSubpix; g.c128 = g.hh * (kSubpix / 2) - 1;
    g.rD = __builtin_amdgcn_rcpf((float)g.D);  // an estimate is all first_pixel needs
    return g;
}

// One pixel of a fast-path cell (both triangles in one orientation, all coordinates bounded, no near-plane vertex).
// kcol0 / kcol1: the scanline crossings of the cell's two column edges (see the header); skip: bit 0 tri1 / bit 1 tri2 draws nothing.
__device__ __forceinline__ void cell_pixel(int XA, int XB, int XC, int XD, float izA, float izB, float izC, float izD,
                                           uint32_t cA, uint32_t cB, uint32_t cC, uint32_t cD, int kcol0, int kcol1, int px,
                                           int j, uint32_t skip, const RowGeom& g, u64* zb, const RowTies& ties)
{
    const int kdiag = mad24(XC - XA, g.tt, mul24(g.hh, XA));
    const int hX = mul24(g.hh, px * kSubpix + kSubpix / 2);
    const bool regular = XD > XA;
    const bool in1 = (hX < kdiag) == regular;
    if (skip & (in1 ? 1u : 2u)) return;                      // that triangle was removed by the 89-degree filter (dmt:1372)
    int wd = in1 ? XC - XB : XD - XA;
    wd = wd < 0 ? -wd : wd;
    const int area2 = mul24(g.hh, wd);
    const int wconst = mul24(wd, in1 ? g.bb : g.tt);          // the vertex alone on its row: weight constant along the scanline
    int ea = (in1 ? kdiag : kcol1) - hX;
    int eb = hX - (in1 ? kcol0 : kdiag);
    ea = ea < 0 ? -ea : ea;
    eb = eb < 0 ? -eb : eb;
    // tri1 = (A, B, C): w0 = wconst, w1 = |kdiag - hX|, w2 = |hX - kcol0|;  tri2 = (A, C, D): w0 = |kcol1 - hX|, w1 = wconst, w2 = |hX - kdiag|
    const float f0 = (float)(in1 ? wconst : ea), f1 = (float)(in1 ? ea : wconst), f2 = (float)eb;
    const float ra = rcp_exact((float)area2);
    const float l0 = f0 * ra, l1 = f1 * ra, l2 = f2 * ra;
    const float iz1 = in1 ? izB : izC, iz2 = in1 ? izC : izD;
    const float q0 = l0 * izA, q1 = l1 * iz1, q2 = l2 * iz2;
    const float iz = (q0 + q1) + q2;
    const float riz = rcp_exact(iz);
    const uint32_t rgb = shade_px(q0, q1, q2, riz, cA, in1 ? cB : cC, in1 ? cC : cD);
    post_row_fragment(zb, px, iz, rgb, ((in1 ? 0u : 1u) << 16) | (uint32_t)j, ties);
}

// lane i <- lane i + 1 of the wave (lane 63 gets 0)
__device__ __forceinline__ int from_next_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ float from_next_lane(float v) { return __int_as_float(from_next_lane(__float_as_int(v))); }
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) { return (uint32_t)from_next_lane((int)v); }

// Generic 64-bit path for one exotic cell, rasterised by the whole wave (uniform arguments).
__device__ __forceinline__ void exotic_cell_wave(int XA, int XB, int XC, int XD, float izA, float izB, float izC, float izD,
                                              uint32_t cA, uint32_t cB, uint32_t cC, uint32_t cD, uint32_t skip, int Yt, int Yb,
                                              int k, int W, int lane, int j, u64* zb, const RowTies& ties)
{
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (skip & (1u << pass)) continue;
        TriSetup t;
        // vertex order of the reference: tri1 = (v[i,j], v[i+1,j], v[i+1,j+1]); tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])
        const bool ok = pass == 0 ? tri_setup_snapped(t, XA, Yt, izA, XB, Yb, izB, XC, Yb, izC)
                                  : tri_setup_snapped(t, XA, Yt, izA, XC, Yb, izC, XD, Yt, izD);
        if (!ok) continue;
        int q0p = floordiv_subpix(t.minX - kSubpix / 2 + kSubpix - 1), q1p = floordiv_subpix(t.maxX - kSubpix / 2);
        if (q0p < 0) q0p = 0;
        if (q1p > W - 1) q1p = W - 1;
        const uint32_t c1 = pass == 0 ? cB : cC, c2 = pass == 0 ? cC : cD;
        for (int px = q0p + lane; px <= q1p; px += 64) {
            float q0, q1, q2;
            if (!tri_sample(t, px, k, q0, q1, q2)) continue;
            const float iz = (q0 + q1) + q2;
            const float riz = rcp_exact(iz);
            post_row_fragment(zb, px, iz, shade_px(q0, q1, q2, riz, cA, c1, c2), ((uint32_t)pass << 16) | (uint32_t)j, ties);
        }
    }
}

}  // namespace
}  // namespace mdvt
