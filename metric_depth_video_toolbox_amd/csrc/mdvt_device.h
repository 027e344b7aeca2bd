// mdvt_device.h -- device-side helpers shared by the kernels in mdvt_kernels.hip.
//
// Arithmetic decree: every f32 expression below is one IEEE operation per node, the
// translation unit is compiled with -ffp-contract=off and divisions are the correctly rounded ones --
// either hipcc's IEEE expansion or rcp_exact / rcp_div_exact below, which produce the same bits (checked
// exhaustively by mdvt_selftest).  Nothing may be rewritten into a contraction that changes a result.
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

// ---- correctly rounded 1/x and a/b without the generic IEEE expansion ---------------------------
// The decree asks for correctly rounded divisions.  hipcc's expansion (v_div_scale x2, v_rcp, 5 fma,
// v_div_fmas, v_div_fixup) also covers denormals, infinities and over/underflowing quotients; the
// rasteriser's operands are ordinary normal numbers, where Markstein's theorems give the same bits
// in fewer operations (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, ch. 5):
//   * y0 = v_rcp_f32(b) is within 1 ulp of 1/b; e = fma(-b, y0, 1); y = fma(e, y0, y0) is then RN(1/b)
//     unless b's significand is all ones (the one exceptional case of the theorem);
//   * with y = RN(1/b): q0 = RN(a y); q1 = fma(fma(-b, q0, a), y, q0) is a faithful rounding of a/b and
//     q2 = fma(fma(-b, q1, a), y, q1) is RN(a/b).
// Operands outside [2^-30, 2^30] or with an all-ones significand take the compiler's IEEE division, so
// the result is the correctly rounded one for every input.  (A lone a/b gains nothing: the IEEE expansion is
// the same iteration plus scaling, 10 instructions against 9; the gain is in 1/b -- 6 against 10 -- and in
// sharing the reciprocal between 1/Z and dl/Z.)  Checked exhaustively against the IEEE
// expansion on the device by mdvt_selftest (tests/test_gpu_arith.py).
constexpr uint32_t kFastLoBits = 0x30800000u;       // 2^-30
constexpr uint32_t kFastSpanBits = 0x1E000000u;     // 60 binades: [2^-30, 2^30)

__device__ __forceinline__ bool fast_operand(float x)
{
    const uint32_t b = __float_as_uint(x);
    return (b - kFastLoBits) < kFastSpanBits && (b & 0x7FFFFFu) != 0x7FFFFFu;
}

// (the empty asm keeps the compiler from speculating the slow path into a select: it must stay a branch that a wave
//  skips unless one of its lanes needs it)
__device__ __forceinline__ float rcp_exact(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    if (__builtin_expect(!fast_operand(b), 0)) { asm volatile("; ieee 1/x" ::: "memory"); y = 1.0f / b; }
    return y;
}

// RN(1/b) and RN(a/b) together (a vertex needs both 1/Z and dl/Z).  `a_ok`: a is in [2^-30, 2^30).
__device__ __forceinline__ void rcp_div_exact(float a, bool a_ok, float b, float& inv, float& quot)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    const float q0 = a * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    float q = __builtin_fmaf(__builtin_fmaf(-b, q1, a), y, q1);
    if (__builtin_expect(!(a_ok && fast_operand(b)), 0)) { asm volatile("; ieee a/x" ::: "memory"); y = 1.0f / b; q = a / b; }
    inv = y; quot = q;
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

// vertex_general for a frame with nothing but a toe-in (FrameDev.conv_band: fill_frame_dev has checked M[1] = M[4] = M[6] = M[7] = M[9] =
// M[11] = 0 and M[5] = 1 EXACTLY): the products with those entries are +-0 and the sums with them change nothing, so X, Y = yc
// and Z' come out bit for bit as vertex_general's with eleven operations fewer per eye -- except for the SIGN of a zero (x + 0 is +0
// where x is -0), which no result depends on: a zero X or Y gives u = cxr, v = cyr either way, and Z' = +-0 is a vertex behind the
// near plane whose record carries iz = 0 and is never looked at.
__device__ __forceinline__ Vert vertex_conv_only(const FrameDev& f, const float* M, float xc, float yc, float z)
{
    Vert o;
    const float X = (M[0] * xc + M[2] * z) + M[3];
    const float Z = M[8] * xc + M[10] * z;
    o.ok = (z > kNear) && (Z > kNear);
#ifdef MDVT_EXP_RCP_VERTEX      // timing experiment only (r06, verdict r05 item 4): ONE reciprocal per vertex and eye, as GL hardware divides -- other bits
    const float r = rcp_exact(Z);
    o.u = (f.fxr * X) * r + f.cxr;
    o.v = (f.fyr * yc) * r + f.cyr;
#else
    o.u = (f.fxr * X) / Z + f.cxr;
    o.v = (f.fyr * yc) / Z + f.cyr;
#endif
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

// ---- rasteriser pieces --------------------------------------------------------------------
// Snapped coordinates are int32 (|x| <= 2^21 px * 256 = 2^29), so every coordinate difference fits
// int32 and every product below is one 32x32->64 multiply (v_mad_i64_i32).
__device__ __forceinline__ int snap(float x)
{
    x = fminf(fmaxf(x, -kSnapLimit), kSnapLimit);
    return (int)rintf(x * (float)kSubpix);
}

// floor division by the sub-pixel grid (power of two): arithmetic shift.
__device__ __forceinline__ int floordiv_subpix(int a) { return a >> kSubpixBits; }

// GL_POINTS of size 1 (dmt:1510): the unit square around the SNAPPED vertex, rasterised with the fill rule below -- the
// pixel whose centre lies inside it, a centre on the square's left or bottom edge belongs to it:
//   column = ceil(X / S) - 1,  row = floor(Y / S)        (X, Y snapped, S = kSubpix)
// -1 / out of range = not drawn.  Pinned by tests/golden/render_gl_points_*.npz.
__device__ __forceinline__ int point_col(float u) { return floordiv_subpix(snap(u) - 1); }
__device__ __forceinline__ int point_row(float v) { return floordiv_subpix(snap(v)); }
// point_col in three instructions, for the LDS row kernels: the bits of fma(u, S, 1.5 * 2^23) are 0x4B400000 + rint(u S) while
// |u S| < 2^22 (a single rounding of an exact product: round-half-even, like snap's rintf), and outside that range they still
// lie on the same side of every column of a row that fits the LDS (W <= 10240 << 2^22 / S): "not drawn" either way.
__device__ __forceinline__ int point_col_row_kernel(float u)
{
    const int X1 = __float_as_int(__builtin_fmaf(u, (float)kSubpix, 12582912.0f)) - 0x4B400001;     // rint(u S) - 1
    return X1 >> kSubpixBits;
}

// Fill rule: a pixel centre exactly on an edge belongs to the triangle if the edge is a LEFT edge or a horizontal BOTTOM
// edge (image space, y down; deltas orientation-normalised to clockwise).  This is OpenGL's convention -- the rasteriser's
// "top-left" rule acts towards window y = 0, the bottom of the picture -- as observed on the pinned GL
// (tests/golden/render_gl_*.npz); until r06 the decree had Direct3D's (top edges).
__device__ __forceinline__ bool edge_owns_centre(int dx, int dy) { return (dy < 0) || (dy == 0 && dx < 0); }
__device__ __forceinline__ bool edge_in(i64 w, int dx, int dy)
{
    if (w > 0) return true;
    if (w < 0) return false;
    return edge_owns_centre(dx, dy);
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
// cull: 0 none, 1 back faces (area2 > 0 with y down: clockwise on screen), 2 front faces (mdvt_config.cull).
__device__ __forceinline__ bool tri_setup_snapped(TriSetup& t, int X0, int Y0, float iz0, int X1, int Y1, float iz1,
                                                  int X2, int Y2, float iz2, int cull = 0)
{
    t.area2 = 0;
    if (!(iz0 > 0.0f && iz1 > 0.0f && iz2 > 0.0f)) return false;    // near plane: whole triangle dropped
    const i64 a2 = mul64(X1 - X0, Y2 - Y0) - mul64(Y1 - Y0, X2 - X0);
    if (a2 == 0) return false;
    const bool neg = a2 < 0;
    if (cull && (cull == 1) != neg) return false;
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

__device__ __forceinline__ bool tri_setup(TriSetup& t, const Vert& a, const Vert& b, const Vert& c, int cull = 0)
{
    return tri_setup_snapped(t, snap(a.u), snap(a.v), a.ok ? rcp_exact(a.z) : 0.0f, snap(b.u), snap(b.v),
                             b.ok ? rcp_exact(b.z) : 0.0f, snap(c.u), snap(c.v), c.ok ? rcp_exact(c.z) : 0.0f, cull);
}

// ---- z-buffer words and exact depth ties ------------------------------------------------------------------------
// A z-buffer word is 64 bits, high half = ~bits(1/Z) of the fragment (1/Z > 0, so the unsigned minimum is the nearest
// fragment; all ones = empty).  OpenGL resolves an EXACT depth tie between overlapping triangles by draw order: with
// GL_LESS the triangle drawn first keeps the pixel, and the reference draws all tri1 row-major, then all tri2
// (dmt:1243-1254).  Such ties are not rare (a few per 10^5 fragments on the benchmark scenes), so both kinds of
// kernel implement that rule exactly:
//   * global-key kernels (general path, points): low half = source index, so that the 64-bit minimum IS "nearest, then lower
//     source index";
//   * global-key kernels (general path, mesh): low half = "no tie yet" bit | the shaded colour (the resolve stays a plain read of
//     ONE word per pixel).  A fragment that meets its own depth with another colour in the word clears the bit (ColourKeys
//     below), and the frames where that happened are rasterised once more (only the cells near such pixels do any work) to
//     post draw id << 32 | colour of the fragments at the winning depth of each tied pixel into a side word: its minimum is
//     "first drawn";
//   * LDS row kernels (pure shift): low half = the shaded colour (the resolve stays a plain read); a fragment that meets
//     its own depth with another colour in the word raises the pixel's tie bit, and a row with tie bits is rasterised
//     twice more -- once to find the lowest draw id among the fragments at the winning depth of each tied pixel, once
//     to let exactly that fragment write its colour (RowTies below).
__device__ __forceinline__ uint32_t draw_id_global(int pass, int i, int j) { return ((uint32_t)pass << 31) | ((uint32_t)i << 16) | (uint32_t)j; }
__device__ __forceinline__ uint32_t depth_bits(float iz) { return ~__float_as_uint(iz); }

// ---- z keys of the general paths (global memory), parity scheme ------------------------------------------------------
// A slot of the key planes is used with alternating parity.  Parity 0: keys have the top bit clear and are posted with
// atomicMin; parity 1: keys have it set and are posted with atomicMax.  "Covered by this use" is then simply "top bit ==
// parity": whatever the previous use (other parity) left in a word loses to every fragment of this one and does not count as
// covered, so the resolve pass only has to rewrite the words this use left UNcovered (the holes, a few per cent) with the
// next parity's empty value -- not reset the whole plane (8 B/px per eye) after every use.
//   payload (63 bits) = 31-bit order key of the depth << 32 | 32-bit tie breaker, arranged so that the winner is the
//   nearest fragment, ties to the smaller tie breaker (first drawn triangle / lower source index).
__device__ __forceinline__ u64 zkey_empty(uint32_t parity) { return parity ? 0ull : ~0ull; }
__device__ __forceinline__ bool zkey_covered(u64 key, uint32_t parity) { return (uint32_t)(key >> 63) == parity; }
// `nearer_is_larger`: the 31-bit value grows towards the camera (1/Z bits: mesh) or away from it (Z bits: points)
template <bool NEARER_IS_LARGER>
__device__ __forceinline__ void zkey_post(u64* word, uint32_t parity, uint32_t depth31, uint32_t tie)
{
    // parity 0 / atomicMin wants "nearer = smaller, first = smaller"; parity 1 / atomicMax the opposite
    const uint32_t d_min = NEARER_IS_LARGER ? 0x7FFFFFFFu - depth31 : depth31;
    if (parity == 0u) atomicMin(word, ((u64)d_min << 32) | tie);
    else atomicMax(word, (1ull << 63) | ((u64)(0x7FFFFFFFu - d_min) << 32) | (uint32_t)~tie);
}
// The word zkey_post would post.
template <bool NEARER_IS_LARGER>
__device__ __forceinline__ u64 zkey_word(uint32_t parity, uint32_t depth31, uint32_t tie)
{
    const uint32_t d_min = NEARER_IS_LARGER ? 0x7FFFFFFFu - depth31 : depth31;
    return parity == 0u ? ((u64)d_min << 32) | tie : (1ull << 63) | ((u64)(0x7FFFFFFFu - d_min) << 32) | (uint32_t)~tie;
}
__device__ __forceinline__ u64 zkey_post_word(u64* word, uint32_t parity, u64 w)          // returns what the word held
{
    return parity == 0u ? atomicMin(word, w) : atomicMax(word, w);
}

// ---- colour keys of the general mesh path ----------------------------------------------------------------------------
// tie breaker field = kNoTie | rgb.  Clearing kNoTie in a word (in the posted representation: parity 1 stores the field
// complemented) makes it beat every unmarked word of the same depth, whatever its colour: once a pixel is marked at the winning
// depth it stays marked; a nearer fragment replaces the word, mark and all.  A mark that lands on a word other than the one it
// was meant for (a nearer fragment got in between) is harmless: a marked pixel is settled by draw id among the fragments AT ITS
// FINAL DEPTH, which is right whether or not there was a tie.
constexpr uint32_t kNoTie = 1u << 24;
__device__ __forceinline__ void zkey_mark_tied(u64* word, uint32_t parity)
{
    if (parity == 0u) atomicAnd(word, ~(u64)kNoTie); else atomicOr(word, (u64)kNoTie);
}
// same depth (and parity), other colour, not yet marked?  `old` is what a post of `mine` returned.
__device__ __forceinline__ bool zkey_colour_conflict(u64 old, u64 mine, uint32_t parity)
{
    const u64 x = old ^ mine;
    const bool unmarked = (((uint32_t)old >> 24) & 1u) != parity;
    return (x >> 32) == 0ull && ((uint32_t)x & 0xFFFFFFu) != 0u && unmarked;
}
__device__ __forceinline__ bool zkey_is_tied(u64 key, uint32_t parity) { return (((uint32_t)key >> 24) & 1u) == parity; }

template <bool NEARER_IS_LARGER>
__device__ __forceinline__ void zkey_decode(u64 key, uint32_t parity, uint32_t& depth31, uint32_t& tie)
{
    const uint32_t hi = (uint32_t)(key >> 32) & 0x7FFFFFFFu;
    const uint32_t d_min = parity ? 0x7FFFFFFFu - hi : hi;
    depth31 = NEARER_IS_LARGER ? 0x7FFFFFFFu - d_min : d_min;
    tie = parity ? ~(uint32_t)key : (uint32_t)key;
}

struct RowTies {
    uint32_t* bits;      // LDS: one bit per pixel of the row, then one flag word (index nwords)
    int nwords;
    int mode;            // 0: normal pass; 1: lowest draw id at the winning depth of tied pixels; 2: that fragment's colour
    bool force;          // test hook (MDVT_DEBUG_SKIP bit 5): every pixel that receives a second fragment counts as tied
    __device__ __forceinline__ bool tied(int px) const { return (bits[px >> 5] >> (px & 31)) & 1u; }
};

// One shaded fragment of a row kernel.  draw = pass << 16 | cell column (one row of cells per scanline).
__device__ __forceinline__ void post_row_fragment(u64* zb, int px, float iz, uint32_t rgb, uint32_t draw, const RowTies& t)
{
    const uint32_t hi = depth_bits(iz);
    if (t.mode == 0) {
        const u64 key = ((u64)hi << 32) | rgb;
        const u64 old = atomicMin(&zb[px], key);
        if (((uint32_t)(old >> 32) == hi && (uint32_t)old != rgb) || (t.force && old != kEmpty64)) {   // same depth, another colour: whoever came second sees it
            atomicOr(&t.bits[px >> 5], 1u << (px & 31));
            t.bits[t.nwords] = 1u;
        }
    } else if (t.tied(px)) {
        if (t.mode == 1) {
            if ((uint32_t)(zb[px] >> 32) == hi) atomicMin(&zb[px], ((u64)hi << 32) | draw);
        } else if (zb[px] == (((u64)hi << 32) | draw)) {
            zb[px] = ((u64)(hi & 0x7FFFFFFFu) << 32) | rgb;             // top bit cleared = final (no depth looks like this)
        }
    }
}
// between mode 0 and mode 1: a tied pixel keeps its depth and forgets its colour
__device__ __forceinline__ void row_ties_prepare(u64* zb, int W, const RowTies& t, int tid, int nthreads)
{
    for (int x = tid; x < W; x += nthreads)
        if (t.tied(x)) zb[x] |= 0xFFFFFFFFull;
}
// 1/Z bits of a resolved row word (the top bit of ~bits is always set; a finalised tie has it cleared)
__device__ __forceinline__ float row_word_iz(uint32_t hi) { return __uint_as_float(~(hi | 0x80000000u)); }

// ---- 32-bit twin of the triangle set-up for small triangles ---------------------------------------------------------
// A triangle whose snapped extent is below 2^13 sub-pixels (32 px) has every edge value and its doubled area below
// 2^27 and every factor below 2^23: the same integers as the 64-bit path from 24-bit multiplies (full rate; the
// 32x32 -> 64 multiplies of the generic path are quarter rate) and 32-bit adds and compares.
constexpr int kSmallTriExtent = 8192;
struct TriSmall {
    int dx0, dy0, dx1, dy1, dx2, dy2;   // orientation-normalised, as in TriSetup
    int bx0, by0, bx1, by1, bx2, by2;
    int area2;
    float iz0, iz1, iz2;
};

__device__ __forceinline__ bool tri_small_setup(TriSmall& t, int X0, int Y0, float iz0, int X1, int Y1, float iz1,
                                                int X2, int Y2, float iz2, int cull)
{
    if (!(iz0 > 0.0f && iz1 > 0.0f && iz2 > 0.0f)) return false;    // near plane: whole triangle dropped
    const int a2 = __mul24(X1 - X0, Y2 - Y0) - __mul24(Y1 - Y0, X2 - X0);
    if (a2 == 0) return false;
    const bool neg = a2 < 0;
    if (cull && (cull == 1) != neg) return false;
    t.area2 = neg ? -a2 : a2;
    t.dx0 = neg ? X1 - X2 : X2 - X1; t.dy0 = neg ? Y1 - Y2 : Y2 - Y1;
    t.dx1 = neg ? X2 - X0 : X0 - X2; t.dy1 = neg ? Y2 - Y0 : Y0 - Y2;
    t.dx2 = neg ? X0 - X1 : X1 - X0; t.dy2 = neg ? Y0 - Y1 : Y1 - Y0;
    t.bx0 = X1; t.by0 = Y1; t.bx1 = X2; t.by1 = Y2; t.bx2 = X0; t.by2 = Y0;
    t.iz0 = iz0; t.iz1 = iz1; t.iz2 = iz2;
    return true;
}

// The walk carries BIASED edge values: w_k + 1 where the fill rule admits w_k == 0 (edge_in), w_k otherwise, so
// "inside" is min(w0, w1, w2) > 0 -- two instructions per pixel centre instead of a dozen compares.
struct TriWalk32 { int w0, w1, w2; };
__device__ __forceinline__ int edge_bias(int dx, int dy) { return edge_owns_centre(dx, dy) ? 1 : 0; }
__device__ __forceinline__ TriWalk32 tri_small_start(const TriSmall& t, int px, int py)
{
    const int Xc = px * kSubpix + kSubpix / 2, Yc = py * kSubpix + kSubpix / 2;
    return TriWalk32{__mul24(t.dx0, Yc - t.by0) - __mul24(t.dy0, Xc - t.bx0) + edge_bias(t.dx0, t.dy0),
                     __mul24(t.dx1, Yc - t.by1) - __mul24(t.dy1, Xc - t.bx1) + edge_bias(t.dx1, t.dy1),
                     __mul24(t.dx2, Yc - t.by2) - __mul24(t.dy2, Xc - t.bx2) + edge_bias(t.dx2, t.dy2)};
}
__device__ __forceinline__ void tri_small_right(const TriSmall& t, TriWalk32& w) { w.w0 -= t.dy0 * kSubpix; w.w1 -= t.dy1 * kSubpix; w.w2 -= t.dy2 * kSubpix; }
__device__ __forceinline__ void tri_small_down(const TriSmall& t, TriWalk32& w) { w.w0 += t.dx0 * kSubpix; w.w1 += t.dx1 * kSubpix; w.w2 += t.dx2 * kSubpix; }
__device__ __forceinline__ bool tri_small_inside(const TriSmall& t, const TriWalk32& w) { return min3i(w.w0, w.w1, w.w2) > 0; }
// q_k = (f32(w_k) * (1 / f32(area2))) * (1/Z_k): what tri_weights computes for a small triangle
// (half the small triangles cover no pixel centre at all, so 1/area2 is worked out here, per fragment -- the empty asm
//  keeps the compiler from hoisting it, with both branches of rcp_exact, into every triangle's set-up)
__device__ __forceinline__ void tri_small_weights(const TriSmall& t, const TriWalk32& w, float& q0, float& q1, float& q2)
{
    float fa = (float)t.area2;
    asm volatile("" : "+v"(fa));
    const float ra = rcp_exact(fa);
    q0 = ((float)(w.w0 - edge_bias(t.dx0, t.dy0)) * ra) * t.iz0;
    q1 = ((float)(w.w1 - edge_bias(t.dx1, t.dy1)) * ra) * t.iz1;
    q2 = ((float)(w.w2 - edge_bias(t.dx2, t.dy2)) * ra) * t.iz2;
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
    const float ra = rcp_exact(fa);
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

// Conservative pixel-column range [lo, hi] of a triangle on pixel row py: each orientation-normalised edge
// w_k = A_k - dy_k * X >= 0 bounds X from above (dy_k > 0) or below (dy_k < 0); the crossings are estimated
// in float and widened by one pixel (they are good to ~0.1 px inside the snap range), so the range is a
// superset of the covered pixels and the exact integer test still decides.  Returns false for an empty row.
__device__ __forceinline__ bool tri_row_range(const TriSetup& t, int py, int px0, int px1, int& lo, int& hi)
{
    const int Yc = py * kSubpix + kSubpix / 2;
    float flo = -3.0e9f, fhi = 3.0e9f;
    const int dxs[3] = {t.dx0, t.dx1, t.dx2}, dys[3] = {t.dy0, t.dy1, t.dy2};
    const int bxs[3] = {t.bx0, t.bx1, t.bx2}, bys[3] = {t.by0, t.by1, t.by2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const i64 A = mul64(dxs[k], Yc - bys[k]) + mul64(dys[k], bxs[k]);
        if (dys[k] == 0) { if (A < 0) return false; continue; }
        const float x = (float)A / (float)dys[k];
        if (dys[k] > 0) fhi = fminf(fhi, x); else flo = fmaxf(flo, x);
    }
    constexpr float kHalf = (float)(kSubpix / 2), kInv = 1.0f / (float)kSubpix;
    const float l = floorf((flo - kHalf) * kInv) - 1.0f, h = floorf((fhi - kHalf) * kInv) + 1.0f;
    lo = l < (float)px0 ? px0 : (l > (float)px1 ? px1 + 1 : (int)l);
    hi = h > (float)px1 ? px1 : (h < (float)px0 ? px0 - 1 : (int)h);
    return lo <= hi;
}

// Perspective-correct colour, rounded half-even to u8 (decree): rint(((q0 c0 + q1 c1) + q2 c2) * (1/iz)) clamped to
// [0, 255], NaN -> 0.  v_cvt_pk_u8_f32 IS that conversion (round to nearest even, saturating, NaN -> 0 -- checked for
// every f32 by mdvt_selftest) and packs the byte in place; R and G travel as one v_pk_mul/add_f32 pair (two IEEE f32
// operations per instruction).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t shade_px(float q0, float q1, float q2, float riz,
                                             uint32_t p0, uint32_t p1, uint32_t p2)
{
    const f32x2 a = {(float)(p0 & 0xFFu), (float)((p0 >> 8) & 0xFFu)};
    const f32x2 b = {(float)(p1 & 0xFFu), (float)((p1 >> 8) & 0xFFu)};
    const f32x2 c = {(float)(p2 & 0xFFu), (float)((p2 >> 8) & 0xFFu)};
    const f32x2 num = (q0 * a + q1 * b) + q2 * c;
    const f32x2 val = num * riz;
    const float nb = (q0 * (float)((p0 >> 16) & 0xFFu) + q1 * (float)((p1 >> 16) & 0xFFu)) + q2 * (float)((p2 >> 16) & 0xFFu);
    const float vb = nb * riz;
    uint32_t out = __builtin_amdgcn_cvt_pk_u8_f32(val.x, 0u, 0u);
    out = __builtin_amdgcn_cvt_pk_u8_f32(val.y, 1u, out);
    return __builtin_amdgcn_cvt_pk_u8_f32(vb, 2u, out);
}

// The decree's own form of one channel (what shade_px must reproduce): used by the self-test only.
__device__ __forceinline__ uint32_t shade_channel_reference(float v)
{
    float val = rintf(v);
    if (!(val >= 0.0f)) val = 0.0f;
    if (val > 255.0f) val = 255.0f;
    return (uint32_t)val;
}

// ---- f64 vertex of the 89-degree filter (dmt:1117-1128 as NumPy >= 2 evaluates it) ----
__device__ __forceinline__ void vertex_f64(const FrameDev& f, int i, int j, int of_by_one, float z, double (&p)[3])
{
    const double x = of_by_one ? (double)((float)j * f.sx) : (double)j;   // dmt:1117-1122 (f32 grid)
    const double y = of_by_one ? (double)((float)i * f.sy) : (double)i;
    p[0] = (x - f.Kd[2]) * (double)z / f.Kd[0];
    p[1] = (y - f.Kd[3]) * (double)z / f.Kd[1];
    p[2] = (double)z;
}


// =================================================================================================
// edge points: the reference's own f64 chain (sr:589-606, 615-619, 727-735, 745-752, 838-858)
// =================================================================================================
// A vertex of a removed triangle is splatted where THIS sequence of f64 operations puts it: NumPy's unprojection
// (vertex_f64 above), the "undo" of the off-by-one scale (sr:599-600), Open3D's in-place transform / rotate / translate of
// the point cloud (4x4 times (x,y,z,1) divided by w; R (p - 0) + 0; p += t), the right eye's operations applied on top of
// the left eye's (sr:838-847), cv2.projectPoints with the camera matrix cast to f32 (dmt:1058: z = z ? 1/z : 1, x *= z,
// u = x fx + cx in double) and np.round.  One IEEE operation per node, sums left to right, no contraction (the oracle's
// orc_echain_* and the golden tests/golden/edge_points.npz hold the same sequence).  Exact simplifications: a product
// with an exact 0 or 1 entry of Ry and the additions of 0.0 in y and z of a translate change no finite value.
__device__ __forceinline__ void echain_roty(double c, double s, double (&q)[3])      // Ry = [[c,0,s],[0,1,0],[-s,0,c]]
{
    const double x = c * q[0] + s * q[2];
    const double z = (-s) * q[0] + c * q[2];
    q[0] = x; q[2] = z;
}

// a point of the cloud -> where it stands when the left / the right eye is rendered
static __device__ void echain_eyes(const FrameDev& f, const double (&q0)[3], double (&L)[3], double (&R)[3])
{
    double q[3] = {q0[0], q0[1], q0[2]};
    if (f.has_T) {                                                   // sr:615-619
        const double* T = f.Td;
        double hh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hh[r] = ((T[4 * r] * q[0] + T[4 * r + 1] * q[1]) + T[4 * r + 2] * q[2]) + T[4 * r + 3] * 1.0;
        if (hh[3] != 1.0) { hh[0] = hh[0] / hh[3]; hh[1] = hh[1] / hh[3]; hh[2] = hh[2] / hh[3]; }     // (x / 1.0 == x)
        q[0] = hh[0]; q[1] = hh[1]; q[2] = hh[2];
    }
    if (f.has_conv) echain_roty(f.cs[0], -f.cs[1], q);               // sr:729
    q[0] += f.hd;                                                    // sr:731
    L[0] = q[0]; L[1] = q[1]; L[2] = q[2];
    q[0] += -f.hd;                                                   // sr:839
    if (f.has_conv) { echain_roty(f.cs[0], f.cs[1], q); echain_roty(f.cs[0], f.cs[1], q); }       // sr:842-843
    q[0] += -f.hd;                                                   // sr:846
    R[0] = q[0]; R[1] = q[1]; R[2] = q[2];
}

// cv2.projectPoints + np.round; false if the rounded pixel lies outside the frame (sr:747-750)
__device__ __forceinline__ bool echain_pixel(const FrameDev& f, int W, int H, const double (&q)[3], int& px, int& py)
{
    const double iz = q[2] != 0.0 ? 1.0 / q[2] : 1.0;
    const double u = (q[0] * iz) * (double)f.fxr + (double)f.cxr;
    const double v = (q[1] * iz) * (double)f.fyr + (double)f.cyr;
    const double ru = rint(u), rv = rint(v);
    if (!(ru >= 0.0 && ru < (double)W && rv >= 0.0 && rv < (double)H)) return false;      // (NaN fails)
    px = (int)ru; py = (int)rv;
    return true;
}

// The painter's order of the edge points (sr:752: far to near, i.e. the nearest one keeps the pixel) as an unsigned key:
// the chain's depth rounded to f32, mapped so that a smaller key is a smaller (nearer, or more negative) depth.
__device__ __forceinline__ uint32_t echain_order_key(double z)
{
    const uint32_t b = __float_as_uint((float)z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct EdgePx { int x[2], y[2]; uint32_t zkey[2]; bool ok[2]; };

// Edge point of vertex (i, j), both eyes.  Depth code 0 (Z = 0) is not splatted (decree).
__device__ inline void edge_point_pixels(const FrameDev& f, int W, int H, int i, int j, int of_by_one, float z, EdgePx& o)
{
    o.ok[0] = o.ok[1] = false;
    if (!(z > kNear)) return;
    double p[3], L[3], R[3];
    vertex_f64(f, i, j, of_by_one, z, p);
    p[0] *= f.sWd; p[1] *= f.sHd;                                    // sr:599-600
    echain_eyes(f, p, L, R);
    o.ok[0] = echain_pixel(f, W, H, L, o.x[0], o.y[0]);
    o.ok[1] = echain_pixel(f, W, H, R, o.x[1], o.y[1]);
    o.zkey[0] = echain_order_key(L[2]); o.zkey[1] = echain_order_key(R[2]);
}

// ---- pure-shift frames in the LDS row kernels: the column from an f32 estimate, the chain only near a rounding tie ----
// Without pose and convergence the chain's column is  round( ((gx - cx) z / fx sW + h [- h - h]) (1/z) fxr + cxr ), which
// the f32 expression ex +- dl/z, ex = (gx - cx) sW + cx, misses by at most ~5 W 2^-23 px (three roundings of values <= W,
// the f32 sW, dl and fx against their f64 counterparts, |dl/z| <= W + 1 for a point near the frame): wherever that estimate
// is further than `guard` = W 2^-19 from a tie its rounding IS the chain's, and only the other points (~1 %) run the chain.
static __device__ int edge_col_chain(const FrameDev& f, int eye, float gx, float z, int W)
{
    double X = (((double)gx - f.Kd[2]) * (double)z) / f.Kd[0];
    X *= f.sWd;
    X += f.hd;
    if (eye) { X += -f.hd; X += -f.hd; }
    const double iz = 1.0 / (double)z;                               // (z > kNear)
    const double u = (X * iz) * (double)f.fxr + (double)f.cxr;
    const double r = rint(u);
    return (r >= 0.0 && r < (double)W) ? (int)r : -1;
}

__device__ __forceinline__ float edge_col_guard(int W) { return (float)(W < 1024 ? 1024 : W) * 1.9073486328125e-06f; }    // W 2^-19

// column of the edge point, or -1 if it falls outside the frame.  d = dl / z.
__device__ __forceinline__ int edge_col_pure(const FrameDev& f, int eye, float gx, float z, float d, int W, float guard)
{
    const float ex = ((gx - f.cx) * f.sW) + f.cx;
    const float u = eye == 0 ? ex + d : ex - d;
    if (!(u > -1.0f && u < (float)W + 1.0f)) return -1;
    const float r = rintf(u);
    int x = (int)r;
    if (__builtin_expect(fabsf(u - r) > 0.5f - guard, 0)) { asm volatile("; edge column by the chain" ::: "memory"); x = edge_col_chain(f, eye, gx, z, W); }
    return (x >= 0 && x < W) ? x : -1;
}

// The same without the chain: -2 = inside the guard band (the caller runs edge_col_chain, once, outside its unrolled loops).
__device__ __forceinline__ int edge_col_estimate(const FrameDev& f, int eye, float gx, float d, int W, float guard)
{
    const float ex = ((gx - f.cx) * f.sW) + f.cx;
    const float u = eye == 0 ? ex + d : ex - d;
    if (!(u > -1.0f && u < (float)W + 1.0f)) return -1;
    const float r = rintf(u);
    if (fabsf(u - r) > 0.5f - guard) return -2;
    const int x = (int)r;
    return (x >= 0 && x < W) ? x : -1;
}

// scanlines whose edge points the LDS row kernels leave to k_edge_rows_exact
__device__ __forceinline__ bool edge_row_deferred(const FrameDev& f, int k) { return k >= f.erow_lo && k <= f.erow_hi && f.erow_lo < f.erow_hi; }


// =================================================================================================
// infill-mask seed image (sr:787-803): colour of one hole pixel
// =================================================================================================

// Unit normal (f64) of the LAST triangle in the reference's draw order that contains vertex (i,j) -- what
// the last-writer-wins scatter of dmt:1358-1364 leaves in normals_of_vertexes -- and the vertex itself.
static __device__ void removed_vertex_normal(const RenderArgs& a, const FrameDev& fp, int f, int i, int j, int of_by_one,
                                      double (&n)[3], double (&p)[3])
{
    const int W = a.W, H = a.H;
    int ci, cj, pass;
    if (i <= H - 2 && j <= W - 2) { pass = 1; ci = i; cj = j; }               // tri2(i,j): vertex is A
    else if (i <= H - 2 && j >= 1) { pass = 1; ci = i; cj = j - 1; }          // tri2(i,j-1): D
    else if (i >= 1 && j >= 1) { pass = 1; ci = i - 1; cj = j - 1; }          // tri2(i-1,j-1): C
    else { pass = 0; ci = i - 1; cj = j; }                                    // only (H-1, 0): tri1(H-2,0): B
    const int vi[3] = {ci, ci + 1, pass == 0 ? ci + 1 : ci};
    const int vj[3] = {cj, pass == 0 ? cj : cj + 1, cj + 1};
    double v[3][3];
    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float z = decode_z(code16_of(load_px_bytes(dbase + (size_t)vi[k] * a.depth_pitch, vj[k])), fp.mult, fp.scale);
        vertex_f64(fp, vi[k], vj[k], of_by_one, z, v[k]);
    }
    const double e1x = v[1][0] - v[0][0], e1y = v[1][1] - v[0][1], e1z = v[1][2] - v[0][2];
    const double e2x = v[2][0] - v[0][0], e2y = v[2][1] - v[0][1], e2z = v[2][2] - v[0][2];
    const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double len = sqrt((nx * nx + ny * ny) + nz * nz);
    if (len > 0.0) { n[0] = nx / len; n[1] = ny / len; n[2] = nz / len; }
    else { n[0] = n[1] = n[2] = 1.0; }                                        // dmt:1348-1353
    const float zp = decode_z(code16_of(load_px_bytes(dbase + (size_t)i * a.depth_pitch, j)), fp.mult, fp.scale);
    vertex_f64(fp, i, j, of_by_one, zp, p);
}

// (n'+1)/2*255 truncated; n' = the point n + p (p before the undo scale, sr:596) and the undo-scaled edge point, both taken
// through the chain, their difference normalised (sr:596-600, 727-733, 777-802).
static __device__ uint32_t edge_normal_colour(const RenderArgs& a, const FrameDev& fp, int f, int eye, int i, int j, int of_by_one)
{
    double n[3], p[3];
    removed_vertex_normal(a, fp, f, i, j, of_by_one, n, p);
    const double pa[3] = {n[0] + p[0], n[1] + p[1], n[2] + p[2]};
    const double q[3] = {p[0] * fp.sWd, p[1] * fp.sHd, p[2]};
    double aL[3], aR[3], qL[3], qR[3], d[3];
    echain_eyes(fp, pa, aL, aR);
    echain_eyes(fp, q, qL, qR);
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = eye == 0 ? aL[r] - qL[r] : aR[r] - qR[r];
    const double len = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    uint32_t rgb = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double c = ((d[r] / len) + 1.0) / 2.0 * 255.0;
        rgb |= ((c >= 0.0 && c < 256.0) ? (uint32_t)c : 0u) << (8 * r);
    }
    return rgb;
}

// Seed colour of output pixel (x,y): black outside holes; in holes the edge point's normal colour if one
// was splatted there (esrc = i<<16 | j of the winning edge vertex, or ~0u), else border normals / key colour.
__device__ __forceinline__ uint32_t seed_pixel(const RenderArgs& a, const FrameDev& fp, int f, int eye, int x, int y,
                                               bool hole, uint32_t esrc, int of_by_one)
{
    if (!hole) return 0u;
    if (esrc != ~0u) return edge_normal_colour(a, fp, f, eye, (int)(esrc >> 16), (int)(esrc & 0xFFFFu), of_by_one);
    if (x == 0) return 255u | (127u << 8) | (127u << 16);             // sr:796 normal pointing right
    if (x == a.W - 1) return 0u | (127u << 8) | (127u << 16);         // sr:797 pointing left
    if (y == 0) return 127u | (127u << 8) | (0u << 16);               // sr:798 pointing down
    if (y == a.H - 1) return 127u | (127u << 8) | (255u << 16);       // sr:799 pointing up
    return a.key_rgb;
}


}  // namespace mdvt
