// mdvt_mesh_conv.hip -- MESH MODE with per-frame CONVERGENCE and nothing else: the toe-in of sr:707-726 / 831-836 (each
// eye rotates about the camera's y axis by -+a, then shifts by +-ipd/2), no pose file, K == Krender.  This is what
// movie_2_3D.py:433-445 runs (mesh + --infill_mask + --convergence_file), the product default.
//
// Same results as the general path (k_mesh_vertices_general -> k_mesh_raster_small/queue -> k_resolve_general) and the
// oracle -- the vertex programme IS vertex_general, triangles, tie rule and shading are the decree's -- but with the
// z-buffer of a scanline in LDS and no intermediate in HBM at all (the general path round-trips 16-byte vertex
// records, 64-bit z keys and (until r04) a 64-bit colour side buffer per pixel and eye: 438 MB per 1080p frame, 15 x the
// algorithmic bytes).  What makes that possible:
//
//   * a rotation about the y axis leaves Y' = yc untouched and gives Z' = z (m10 + m8 (gx - cx) / fx): the projected ROW
//     v = (gy - cy) / (m10 + m8 (gx - cx) / fx) + cy of a vertex does not depend on its depth, only on its grid
//     position.  A vertex row is therefore still (almost) a horizontal line on screen -- tilted by a few rows across
//     the frame, in opposite directions for the two eyes -- and an output scanline is covered, column by column, by ONE
//     row of cells: the row i(j) with Y[i][j] < Yc <= Y[i+1][j] ("the bracket of column j": bottom edges own their centres).  Depth only enters the
//     snapped Y through f32 rounding, so the brackets are found from the staged vertices themselves, never assumed.
//   * one workgroup renders (frame, band of scanlines, EYE) -- an eye's vertices are 16 bytes {X, Y, 1/Z', rgb} and two
//     rows of them plus the scanline's z keys fit 2 workgroups per CU, exactly like k_mesh_band.  The "two rows" are
//     per COLUMN: ring slot (i & 1) of column j holds row i of that column, and every column advances its own
//     bracket as the scanline moves down (the bracket rows form a staircase across the frame).
//   * a cell whose two columns have the same bracket row is crossed by the scanline through its two column edges and
//     its diagonal, never through its top or bottom edge: the covered pixels are again [P(left crossing), P(right
//     crossing)) split at P(diagonal crossing), with P(k, h) = ceil((k - 128 h) / (256 h)) now carrying the edge's OWN
//     height h (the three edges of a tilted cell have slightly different heights); the barycentric weights are the
//     three edge functions evaluated directly in 32 bits (24-bit multiplies: |X| < 2^19, cell height < 512).
//   * where the bracket row changes between two neighbouring columns (a handful of places per scanline: the staircase's
//     steps) the scanline leaves a cell through its top or bottom edge, and the ring does not hold all the vertices of the
//     cells involved.  The waves only LIST those column pairs; the last wave to finish the scanline takes the whole list,
//     one step per lane: vertices from the source frame, triangles through the 32-bit small-triangle set-up (a step at a
//     depth edge -- a long triangle -- goes to the generic 64-bit path on the whole wave, like every other irregular cell:
//     near plane, twisted, out of range).
//   * edge points (sr:589-606) land up to a few rows away from their source row: they keep the general path's global
//     64-bit edge keys (the edge-point splat before this kernel, read back here only where the render left a hole,
//     k_edge_keys_reset after it).
#include "mdvt_device.h"

#include <stdlib.h>

namespace mdvt {
namespace MDVT_GRID {      // one copy per sub-pixel grid (mdvt_internal.h)

namespace {

constexpr int kQueueWave = 128;        // (cell, pixel) items per wave: pushes of <= 64 followed by a drain keep it < 128
constexpr int kConvCoord = 1 << 19;    // |X| of the fast path (sub-pixels): every edge value then fits 32 bits
constexpr int kConvMaxH = 512;         // height of a fast-path edge (sub-pixels)
constexpr int kStepCap = 62;           // listed staircase steps per scanline pass (+ the two counters = 64 words of LDS); more: whole-wave path

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// lane i <- lane i + 1 of the wave (lane 63 gets 0)
__device__ __forceinline__ int from_next_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ float from_next_lane(float v) { return __int_as_float(from_next_lane(__float_as_int(v))); }
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) { return (uint32_t)from_next_lane((int)v); }

// A vertex record in the ring: x = snapped X, y = snapped Y, z = bits of 1/Z' (0: behind the near plane), w = rgb |
// flags << 24 (bit 0 / 1: tri1 / tri2 of the cell whose top-left corner this vertex is were removed, dmt:1372) | (row & 31) << 27.
// Z'/z of grid column j (exact-arithmetic value m10 + m8 (gx - cx)/fx; used for ESTIMATES only)
__device__ __forceinline__ float rz_of(const FrameDev& fp, const float* M, int j)
{
    return M[10] + M[8] * (((float)j * fp.sx - fp.cx) / fp.fx);
}
// the grid row whose projection is about the scanline centre Yc (sub-pixels) in a column with Z'/z = rz
__device__ __forceinline__ int est_row(const FrameDev& fp, int Yc, float rz)
{
    const float gy = (((float)Yc * (1.0f / (float)kSubpix) - fp.cyr) * rz) * (fp.fy / fp.fyr) + fp.cy;
    return (int)floorf(gy / fp.sy);
}

// One vertex of the grid for one eye: exactly what k_mesh_vertices_general stores (decode, unproject, vertex_general, snap,
// correctly rounded 1/Z').  A vertex behind the near plane has iz = 0 and gets its analytic row as Y, so that the column's
// brackets stay meaningful (its triangles draw nothing either way).
__device__ __forceinline__ int4 conv_vertex(uint32_t dpx, uint32_t cw, int i, int j, const FrameDev& fp, const float* M)
{
    const float z = decode_z(code16_of(dpx), fp.mult, fp.scale);
    const float gx = (float)j * fp.sx, gy = (float)i * fp.sy;
    float xc, yc;
    camera_point(fp, gx, gy, z, xc, yc);
    const Vert v = vertex_general(fp, M, xc, yc, z);
    float iz = 0.0f, vv = v.v;
    if (v.ok) iz = rcp_exact(v.z);
    else vv = ((gy - fp.cy) * (fp.fyr / fp.fy)) / rz_of(fp, M, j) + fp.cyr;
    return make_int4(snap(v.u), snap(vv), __float_as_int(iz), (int)cw);
}

// P(k, h) = ceil((k - 128 h) / (256 h)) clamped to [0, W]: the first pixel whose centre is at or right of the crossing
// k / h of an edge of height h with the scanline (h in (0, 512), |k| < 2^30).
__device__ __forceinline__ int first_pixel_h(int k, int h, int W)
{
    const int D = h * kSubpix;
    int n = k + h * (kSubpix / 2) - 1;
    n = n < 0 ? 0 : n;
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)D));
    const int rem = n - mul24(q, D);
    q += rem < 0 ? -1 : (rem >= D ? 1 : 0);
    return q > W ? W : q;
}

// One pixel of a fast-path cell.  Triangle = tri1 (A, B, C) or tri2 (A, C, D) in the reference's vertex order
// (dmt:1243-1254); the three raw edge functions of tri_setup_snapped in 32 bits -- inside the triangle they all carry the
// sign of the doubled area, so their absolute values are the orientation-normalised weights of the generic path.
__device__ __forceinline__ void conv_cell_pixel(int XA, int YA, int XB, int YB, int XC, int YC, int XD, int YD, float izA, float izB,
                                                float izC, float izD, uint32_t cA, uint32_t cB, uint32_t cC, uint32_t cD, int px, bool in1,
                                                uint32_t skip, uint32_t cull, int Yc, uint32_t rowcol, u64* zb, const RowTies& ties)
{
    if (skip & (in1 ? 1u : 2u)) return;                      // that triangle was removed by the 89-degree filter (dmt:1372)
    const int X1 = in1 ? XB : XC, Y1 = in1 ? YB : YC, X2 = in1 ? XC : XD, Y2 = in1 ? YC : YD;
    const int Xc = px * kSubpix + kSubpix / 2;
    const int w0 = mul24(X2 - X1, Yc - Y1) - mul24(Y2 - Y1, Xc - X1);
    const int w1 = mul24(XA - X2, Yc - Y2) - mul24(YA - Y2, Xc - X2);
    const int w2 = mul24(X1 - XA, Yc - YA) - mul24(Y1 - YA, Xc - XA);
    const int a2 = (w0 + w1) + w2;                           // = (X1-XA)(Y2-YA) - (Y1-YA)(X2-XA) for any point
    if (a2 == 0) return;
    if (cull && (cull == 1u) != (a2 < 0)) return;            // mdvt_config.cull, as tri_setup_snapped
    const float f0 = (float)(w0 < 0 ? -w0 : w0), f1 = (float)(w1 < 0 ? -w1 : w1), f2 = (float)(w2 < 0 ? -w2 : w2);
    const float ra = rcp_exact((float)(a2 < 0 ? -a2 : a2));
    const float l0 = f0 * ra, l1 = f1 * ra, l2 = f2 * ra;
    const float iz1 = in1 ? izB : izC, iz2 = in1 ? izC : izD;
    const float q0 = l0 * izA, q1 = l1 * iz1, q2 = l2 * iz2;
    const float iz = (q0 + q1) + q2;
    const float riz = rcp_exact(iz);
    const uint32_t rgb = shade_px(q0, q1, q2, riz, cA, in1 ? cB : cC, in1 ? cC : cD);
    post_row_fragment(zb, px, iz, rgb, ((in1 ? 0u : 1u) << 31) | rowcol, ties);
}

// Generic 64-bit path for one cell, this scanline only, rasterised by the whole wave (uniform arguments).
__device__ __forceinline__ void conv_exotic_cell(int XA, int YA, float izA, uint32_t cA, int XB, int YB, float izB, uint32_t cB,
                                                 int XC, int YC, float izC, uint32_t cC, int XD, int YD, float izD, uint32_t cD,
                                                 uint32_t skip, int cull, int k, int W, int lane, uint32_t rowcol, u64* zb, const RowTies& ties)
{
    const int Yc = k * kSubpix + kSubpix / 2;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (skip & (1u << pass)) continue;
        // the scanline must pass through the triangle's rows at all (cheap, before the 64-bit set-up)
        const int Ym = pass == 0 ? min3i(YA, YB, YC) : min3i(YA, YC, YD), YM = pass == 0 ? max3i(YA, YB, YC) : max3i(YA, YC, YD);
        if (Yc < Ym || Yc > YM) continue;
        TriSetup t;
        // vertex order of the reference: tri1 = (v[i,j], v[i+1,j], v[i+1,j+1]); tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])
        const bool ok = pass == 0 ? tri_setup_snapped(t, XA, YA, izA, XB, YB, izB, XC, YC, izC, cull)
                                  : tri_setup_snapped(t, XA, YA, izA, XC, YC, izC, XD, YD, izD, cull);
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
            post_row_fragment(zb, px, iz, shade_px(q0, q1, q2, riz, cA, c1, c2), ((uint32_t)pass << 31) | rowcol, ties);
        }
    }
}

}  // namespace

// FLAGS bit 0: depth planes; bit 1: triangles removed by the 89-degree filter draw nothing; bit 2: edge points (global edge
// keys, posted by the edge-point splat before this kernel); bit 3: the infill-mask seed image (sr:787-803).
template <int FLAGS, int TPB>
__global__ void __launch_bounds__(TPB, 4) k_mesh_conv(RenderArgs a, int rows_per_band, int nbands)
{
    constexpr bool ZOUT = FLAGS & 1, EDGES = FLAGS & 2, EDGEPTS = FLAGS & 4, SEED = FLAGS & 8;
    const uint32_t cull = (uint32_t)a.cull;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W, H = a.H, W4 = W >> 2;
    u64* zb = (u64*)smem;                                   // [W] z keys of the scanline
    int4* ring = (int4*)(zb + W);                           // [2][W]: row i of column j lives in ring[(i & 1) * W + j]
    uint32_t* queue = (uint32_t*)(ring + 2 * (size_t)W);    // [TPB/64][kQueueWave]
    RowTies ties;                                           // [W/32 + 1] exact-depth-tie bits + flag (mdvt_device.h)
    ties.bits = queue + (TPB / 64) * kQueueWave;
    ties.nwords = (W + 31) / 32;
    ties.mode = 0;
    ties.force = (MDVT_DEBUG_SKIP(a) & 32) != 0;
    uint32_t* steps = ties.bits + ties.nwords + 1;          // [kStepCap] listed staircase steps, then [0] their count, [1] waves done
    uint32_t* steps_n = steps + kStepCap;

    const int eye = blockIdx.x & 1;                         // the two eyes of a band run side by side: they read the same source rows
    const int bf = blockIdx.x >> 1;
    const int fr = bf / nbands;
    const int band = bf - fr * nbands;
    const int k0 = band * rows_per_band;
    const int k1 = min(k0 + rows_per_band, H);
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const float* M = fp.M[eye];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wq = queue + wave * kQueueWave;
    const bool act4 = tid < W4;

    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
    const uint8_t* cbase = a.color + (size_t)f * a.color_stride;
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    const uint8_t* tinv = EDGES ? a.tri_invalid + (size_t)fr * a.ws_stride_tri : nullptr;

    // the vertex (row, col) from the source frame
    auto vertex_at = [&](int row, int col) -> int4 {
        const uint32_t dpx = load_px_bytes(dbase + (size_t)row * a.depth_pitch, col);
        uint32_t cw = load_px_bytes(cbase + (size_t)row * a.color_pitch, col);
        if (EDGES && row <= H - 2 && col <= W - 2) {
            const uint8_t* ti = tinv + (size_t)row * (W - 1) + col;
            cw |= (ti[0] ? 1u << 24 : 0u) | (ti[ncell] ? 2u << 24 : 0u);
        }
        cw |= ((uint32_t)row & 31u) << 27;
        return conv_vertex(dpx, cw, row, col, fp, M);
    };
    // rows of a scanline's brackets lie in [ibase, ibase + 31] (the host admits a frame only if the staircase is that low)
    const float rzL = rz_of(fp, M, 0), rzR = rz_of(fp, M, W - 1);
    auto ibase_of = [&](int Yc) {
        const int e = min(est_row(fp, Yc, rzL), est_row(fp, Yc, rzR)) - 4;
        return e < 0 ? 0 : e;
    };
    // Every column's ring is advanced until it brackets the scanline centre Ycn: Y[a] < Ycn <= Y[a + 1] (or the grid ends).
    auto stage_to = [&](int Ycn, int ib, bool init) {
        for (int col = tid; col < W; col += TPB) {
            int arow, Ybot = 0, pending = 0;
            if (init) {
                int e = est_row(fp, Ycn, rz_of(fp, M, col)) - 1;         // one row early: the loop below then walks into the bracket
                e = e < 0 ? 0 : (e > H - 2 ? H - 2 : e);
                arow = e - 2; pending = 2;
            } else {
                const int4 r0 = ring[col], r1 = ring[W + col];
                const bool t0 = r0.y < r1.y;
                const uint32_t tw = (uint32_t)(t0 ? r0.w : r1.w);
                arow = ib + (int)(((tw >> 27) - (uint32_t)ib) & 31u);
                Ybot = t0 ? r1.y : r0.y;
            }
            while (pending > 0 || (Ybot < Ycn && arow + 2 <= H - 1)) {
                const int row = arow + 2;
                const int4 rec = vertex_at(row, col);
                ring[(size_t)(row & 1) * W + col] = rec;
                ++arow; Ybot = rec.y; --pending;
            }
        }
    };

    // The common case of stage_to -- a column moves on by one row per scanline -- with the loads in flight while the scanline is
    // rasterised: a thread owns the columns tid, tid + TPB, ...; for each one whose bracket has to move for the next scanline
    // the next row's pixel is fetched BEFORE the raster passes (prefetch_next) and turned into a vertex record AFTER them
    // (stage_prefetched).  Columns are independent, so the steps of the staircase (columns that stay, or move two rows) cost a
    // masked lane, not a serial pass; a column that has to move a second row is stage_rest's work (as is the band's first scanline:
    // stage_to).
    // prefetch registers of column tid + q * TPB: depth pixel | 1 << 31 (valid), colour word incl. flags and the row's tag
    uint32_t pdx[4] = {0, 0, 0, 0}, pcw[4] = {0, 0, 0, 0};
    auto column_state = [&](int col, int ib, int& arow, int& Ybot) {
        const int4 r0 = ring[col], r1 = ring[W + col];
        const bool t0 = r0.y < r1.y;
        const uint32_t tw = (uint32_t)(t0 ? r0.w : r1.w);
        arow = ib + (int)(((tw >> 27) - (uint32_t)ib) & 31u);
        Ybot = t0 ? r1.y : r0.y;
    };
    auto prefetch_next = [&](int Ycn, int ib) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {            // W <= 4 * TPB (launcher)
            const int col = tid + q * TPB;
            pdx[q] = 0u;
            if (col >= W) continue;
            int arow, Ybot;
            column_state(col, ib, arow, Ybot);
            if (!(Ybot < Ycn && arow + 2 <= H - 1)) continue;
            const int row = arow + 2;
            pdx[q] = load_px_bytes(dbase + (size_t)row * a.depth_pitch, col) | 0x80000000u;
            uint32_t cw = load_px_bytes(cbase + (size_t)row * a.color_pitch, col) | (((uint32_t)row & 31u) << 27);
            if (EDGES && row <= H - 2 && col <= W - 2) {
                const uint8_t* ti = tinv + (size_t)row * (W - 1) + col;
                cw |= (ti[0] ? 1u << 24 : 0u) | (ti[ncell] ? 2u << 24 : 0u);
            }
            pcw[q] = cw;
        }
    };
    // returns true if some column of this thread still does not bracket Ycn
    auto stage_prefetched = [&](int Ycn, int ib) -> bool {
        bool more = false;
        // (one copy of the vertex programme, four trips: unrolled, the four independent programmes cost 60 more VGPRs than the
        //  kernel has at four waves per SIMD)
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const uint32_t dpv = q == 0 ? pdx[0] : (q == 1 ? pdx[1] : (q == 2 ? pdx[2] : pdx[3]));
            if (!(dpv >> 31)) continue;
            const uint32_t dpx = dpv & 0xFFFFFFu;
            const uint32_t cw = q == 0 ? pcw[0] : (q == 1 ? pcw[1] : (q == 2 ? pcw[2] : pcw[3]));
            const int row = ib + (int)(((cw >> 27) - (uint32_t)ib) & 31u);
            const int col = tid + q * TPB;
            const int4 rec = conv_vertex(dpx, cw, row, col, fp, M);
            ring[(size_t)(row & 1) * W + col] = rec;
            more |= rec.y < Ycn && row + 1 <= H - 1;
        }
        return more;
    };
    auto stage_rest = [&](int Ycn, int ib) {      // stage_to for this thread's own columns
        for (int col = tid; col < W; col += TPB) {
            int arow, Ybot;
            column_state(col, ib, arow, Ybot);
            while (Ybot < Ycn && arow + 2 <= H - 1) {
                const int row = arow + 2;
                const int4 rec = vertex_at(row, col);
                ring[(size_t)(row & 1) * W + col] = rec;
                ++arow; Ybot = rec.y;
            }
        }
    };

    // One listed staircase step: the cells (r, bj), r = rlo .. rhi, with their vertices from the source frame, on the whole wave.
    auto step_whole_wave = [&](int bj, int rlo, int rhi, int k) {
        if (rhi > rlo + 29) rhi = rlo + 29;                   // (64 lanes hold the vertices of 30 cells; the host bounds the staircase far below)
        if (rhi > H - 2) rhi = H - 2;
        int4 v = make_int4(0, 0, 0, 0);
        if (lane < 2 * (rhi - rlo + 2)) v = vertex_at(rlo + (lane >> 1), bj + (lane & 1));
        for (int r = rlo; r <= rhi; ++r) {
            const int q = 2 * (r - rlo);
#define MDVT_V(field, idx) __builtin_amdgcn_readlane(v.field, (idx))
            const uint32_t wAq = (uint32_t)MDVT_V(w, q);
            conv_exotic_cell(MDVT_V(x, q), MDVT_V(y, q), __int_as_float(MDVT_V(z, q)), wAq & 0xFFFFFFu,
                             MDVT_V(x, q + 2), MDVT_V(y, q + 2), __int_as_float(MDVT_V(z, q + 2)), (uint32_t)MDVT_V(w, q + 2) & 0xFFFFFFu,
                             MDVT_V(x, q + 3), MDVT_V(y, q + 3), __int_as_float(MDVT_V(z, q + 3)), (uint32_t)MDVT_V(w, q + 3) & 0xFFFFFFu,
                             MDVT_V(x, q + 1), MDVT_V(y, q + 1), __int_as_float(MDVT_V(z, q + 1)), (uint32_t)MDVT_V(w, q + 1) & 0xFFFFFFu,
                             EDGES ? (wAq >> 24) & 3u : 0u, (int)cull, k, W, lane, ((uint32_t)r << 16) | (uint32_t)bj, zb, ties);
#undef MDVT_V
        }
    };
    // One cell of a listed step on ONE lane, its four vertices given: the 32-bit small-triangle set-up, this scanline's pixels
    // walked by the lane.  Returns false -- nothing drawn -- for a cell that is not small (a depth edge: long triangles) or has a
    // vertex out of the 24-bit range or behind the near plane: those go to step_whole_wave.
    auto cell_on_lane = [&](const int4& vA, const int4& vB, const int4& vC, const int4& vD, int r, int bj, int k) -> bool {
        const int Yc = k * kSubpix + kSubpix / 2;
        const int mnX = min(min(vA.x, vB.x), min(vC.x, vD.x)), mxX = max(max(vA.x, vB.x), max(vC.x, vD.x));
        const int mnY = min(vA.y, vD.y), mxY = max(vB.y, vC.y);
        const bool small = mxX - mnX < kSmallTriExtent / 2 && mxY - mnY < kSmallTriExtent / 2 && mnX > -kConvCoord && mxX < kConvCoord &&
                           vA.z != 0 && vB.z != 0 && vC.z != 0 && vD.z != 0;
        if (!small) return false;
        const uint32_t skip = EDGES ? ((uint32_t)vA.w >> 24) & 3u : 0u;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (skip & (1u << pass)) continue;
            const int4 v1 = pass == 0 ? vB : vC, v2 = pass == 0 ? vC : vD;
            if (Yc < min3i(vA.y, v1.y, v2.y) || Yc > max3i(vA.y, v1.y, v2.y)) continue;
            TriSmall ts;
            if (!tri_small_setup(ts, vA.x, vA.y, __int_as_float(vA.z), v1.x, v1.y, __int_as_float(v1.z), v2.x, v2.y, __int_as_float(v2.z), (int)cull)) continue;
            int bx0 = floordiv_subpix(min3i(vA.x, v1.x, v2.x) - kSubpix / 2 + kSubpix - 1), bx1 = floordiv_subpix(max3i(vA.x, v1.x, v2.x) - kSubpix / 2);
            bx0 = max(bx0, 0); bx1 = min(bx1, W - 1);
            if (bx1 < bx0) continue;
            TriWalk32 w = tri_small_start(ts, bx0, k);
            for (int px = bx0; px <= bx1; ++px) {
                if (tri_small_inside(ts, w)) {
                    float q0, q1, q2;
                    tri_small_weights(ts, w, q0, q1, q2);
                    const float iz = (q0 + q1) + q2;
                    post_row_fragment(zb, px, iz, shade_px(q0, q1, q2, rcp_exact(iz), (uint32_t)vA.w & 0xFFFFFFu, (uint32_t)v1.w & 0xFFFFFFu, (uint32_t)v2.w & 0xFFFFFFu),
                                      ((uint32_t)pass << 31) | ((uint32_t)r << 16) | (uint32_t)bj, ties);
                }
                tri_small_right(ts, w);
            }
        }
        return true;
    };

    for (int x = tid; x < W; x += TPB) zb[x] = kEmpty64;
    for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
    if (tid < 2) steps_n[tid] = 0u;
    stage_to(k0 * kSubpix + kSubpix / 2, 0, true);
    __syncthreads();

#pragma unroll 1
    for (int k = k0; k < k1; ++k) {
        const int Yc = k * kSubpix + kSubpix / 2;
        const int ib = ibase_of(Yc);
        const int4* r0p = ring;
        const int4* r1p = ring + W;
        if (k + 1 < k1 && !(MDVT_DEBUG_SKIP(a) & 128)) prefetch_next(Yc + kSubpix, ib);          // in flight while the scanline is rasterised
        // (passes 1 and 2 only for a row with exact depth ties between different colours: RowTies in mdvt_device.h)
#pragma unroll 1
        for (ties.mode = 0; ties.mode < 3; ++ties.mode) {
            if (!(MDVT_DEBUG_SKIP(a) & 1)) {
                int qn = 0;                                          // items on this wave's stack (uniform)
                constexpr int kCellsPerPass = 63 * (TPB / 64);
#pragma unroll 1
                for (int c0 = 0; ; c0 += kCellsPerPass) {
                    const bool final_pass = c0 >= W - 1;
                    const int j = c0 + wave * 63 + lane;
                    const int jc = j < W ? j : W - 1;
                    const int4 s0 = r0p[jc], s1 = r1p[jc];
                    const bool t0 = s0.y < s1.y;
                    const int XA = t0 ? s0.x : s1.x, YA = t0 ? s0.y : s1.y, XB = t0 ? s1.x : s0.x, YB = t0 ? s1.y : s0.y;
                    const float izA = __int_as_float(t0 ? s0.z : s1.z), izB = __int_as_float(t0 ? s1.z : s0.z);
                    const uint32_t wA = (uint32_t)(t0 ? s0.w : s1.w), wB = (uint32_t)(t0 ? s1.w : s0.w);
                    const uint32_t cA = wA & 0xFFFFFFu, cB = wB & 0xFFFFFFu;
                    const int rowA = (int)(wA >> 27);
                    const int hAB = YB - YA;
                    const int brk = (YA < Yc && Yc <= YB) ? 1 : 0;                 // this column's bracket holds
                    const int colok = (izA > 0.0f && izB > 0.0f && hAB < kConvMaxH &&
                                       (((uint32_t)(XA + kConvCoord) | (uint32_t)(XB + kConvCoord)) >> 20) == 0u) ? 1 : 0;
                    const int tA = Yc - YA;
                    const int kAB = mad24(XB - XA, tA, mul24(hAB, XA));
                    const int pAB = first_pixel_h(kAB, hAB, W);
                    const int XD = from_next_lane(XA), YD = from_next_lane(YA), XC = from_next_lane(XB), YC = from_next_lane(YB);
                    const float izD = from_next_lane(izA), izC = from_next_lane(izB);
                    const uint32_t cD = from_next_lane(cA), cC = from_next_lane(cB);
                    const int pDC = from_next_lane(pAB), okD = from_next_lane(colok), brkD = from_next_lane(brk), rowD = from_next_lane(rowA);
                    const bool cell = lane < 63 && j < W - 1 && !final_pass;
                    const bool same_row = rowA == rowD;
                    const int hAC = YC - YA;
                    const int kAC = mad24(XC - XA, tA, mul24(hAC, XA));
                    const bool hok = hAC > 0 && hAC < kConvMaxH;
                    const int pAC = first_pixel_h(kAC, hok ? hAC : 1, W);
                    const bool regular = pAB <= pDC;
                    const bool mono = regular ? (pAB <= pAC && pAC <= pDC) : (pAC <= pAB && pDC <= pAC);
                    const bool fast = cell && same_row && brk && brkD && colok && okD && hok && mono;
                    // a cell both of whose columns put the scanline at / above its top or below its bottom holds no pixel centre of it
                    const bool outside = (Yc <= YA && Yc <= YD) || (YB < Yc && YC < Yc);
                    const bool exotic = cell && same_row && !fast && !outside;
                    const bool step = cell && !same_row;                    // the staircase steps between these two columns
                    const uint32_t skip = EDGES ? (wA >> 24) & 3u : 0u;                                   // dmt:1372
                    const uint32_t rowcol = ((uint32_t)(ib + (int)(((uint32_t)rowA - (uint32_t)ib) & 31u)) << 16) | (uint32_t)j;
                    const int plo = regular ? pAB : pDC;
                    int n = (fast && skip != 3u) ? (regular ? pDC - pAB : pAB - pDC) : 0;
                    if (n > 0 && !(MDVT_DEBUG_SKIP(a) & 16))
                        conv_cell_pixel(XA, YA, XB, YB, XC, YC, XD, YD, izA, izB, izC, izD, cA, cB, cC, cD, plo,
                                        (mul24(hAC, plo * kSubpix + kSubpix / 2) < kAC) == regular, skip, cull, Yc, rowcol, zb, ties);
                    // Further pixels of the cell become (cell, pixel) items on the wave's stack, shaded 64 at a time (as k_mesh_band)
                    if (MDVT_DEBUG_SKIP(a) & 8) n = 0;
                    u64 lm = __ballot(n > 4);
                    if (__ballot(n > 1) != 0ull || final_pass) {
                        int round = 1, lcell = 0, lpix = 0, lrem = 0;
                        for (;;) {
                            if (qn >= 64 || (final_pass && qn > 0)) {
                                const int cnt = qn >= 64 ? 64 : qn;
                                qn -= cnt;
                                if (lane < cnt) {
                                    const uint32_t it = wq[qn + lane];
                                    const int ij = (int)(it >> 16), px = (int)(it & 0xFFFu);
                                    const bool ireg = (it & 0x8000u) != 0u;
                                    const int4 a0 = r0p[ij], a1 = r1p[ij], d0 = r0p[ij + 1], d1 = r1p[ij + 1];
                                    const bool ta = a0.y < a1.y, td = d0.y < d1.y;
                                    const int4 vA = ta ? a0 : a1, vB = ta ? a1 : a0, vD = td ? d0 : d1, vC = td ? d1 : d0;
                                    const int ihAC = vC.y - vA.y;
                                    const int ikAC = mad24(vC.x - vA.x, Yc - vA.y, mul24(ihAC, vA.x));
                                    const uint32_t iw = (uint32_t)vA.w;
                                    conv_cell_pixel(vA.x, vA.y, vB.x, vB.y, vC.x, vC.y, vD.x, vD.y, __int_as_float(vA.z), __int_as_float(vB.z),
                                                    __int_as_float(vC.z), __int_as_float(vD.z), iw & 0xFFFFFFu, (uint32_t)vB.w & 0xFFFFFFu,
                                                    (uint32_t)vC.w & 0xFFFFFFu, (uint32_t)vD.w & 0xFFFFFFu, px,
                                                    (mul24(ihAC, px * kSubpix + kSubpix / 2) < ikAC) == ireg, EDGES ? (iw >> 24) & 3u : 0u, cull, Yc,
                                                    ((uint32_t)(ib + (int)(((iw >> 27) - (uint32_t)ib) & 31u)) << 16) | (uint32_t)ij, zb, ties);
                                }
                                continue;
                            }
                            if (round < 4) {
                                const bool want = n > round;
                                const u64 m = __ballot(want);
                                if (want) wq[qn + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)j << 16) | (regular ? 0x8000u : 0u) | (uint32_t)(plo + round);
                                qn += (int)__popcll(m);
                                round = m ? round + 1 : 4;
                                continue;
                            }
                            if (lrem == 0 && lm) {
                                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                                lm &= lm - 1;
                                lcell = (__builtin_amdgcn_readlane(j, l) << 16) | (__builtin_amdgcn_readlane(regular ? 1 : 0, l) ? 0x8000 : 0);
                                lpix = __builtin_amdgcn_readlane(plo, l) + 4;
                                lrem = __builtin_amdgcn_readlane(n, l) - 4;
                            }
                            if (lrem > 0) {
                                const int cnt = lrem < 64 ? lrem : 64;
                                if (lane < cnt) wq[qn + lane] = (uint32_t)lcell | (uint32_t)(lpix + lane);
                                qn += cnt; lpix += cnt; lrem -= cnt;
                                continue;
                            }
                            break;
                        }
                    }
                    if (final_pass) break;
#define MDVT_BI(v) __builtin_amdgcn_readlane(v, l)
#define MDVT_BF(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l))
#define MDVT_BU(v) (uint32_t)__builtin_amdgcn_readlane((int)(v), l)
                    // irregular cells whose four vertices are in the ring (near plane, out of range, twisted, grid border)
                    u64 em = (MDVT_DEBUG_SKIP(a) & 8) ? 0ull : __ballot(exotic);
                    while (em) {
                        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)em) - 1);
                        em &= em - 1;
                        conv_exotic_cell(MDVT_BI(XA), MDVT_BI(YA), MDVT_BF(izA), MDVT_BU(cA), MDVT_BI(XB), MDVT_BI(YB), MDVT_BF(izB), MDVT_BU(cB),
                                         MDVT_BI(XC), MDVT_BI(YC), MDVT_BF(izC), MDVT_BU(cC), MDVT_BI(XD), MDVT_BI(YD), MDVT_BF(izD), MDVT_BU(cD),
                                         MDVT_BU(skip), (int)cull, k, W, lane, MDVT_BU(rowcol), zb, ties);
                    }
                    // steps of the staircase: listed for the last wave out (below); a full list: here and now, on this wave
                    if (__ballot(step) != 0ull && !(MDVT_DEBUG_SKIP(a) & 8)) {
                        uint32_t slot = kStepCap;
                        if (step) {
                            const int ra = ib + (int)(((uint32_t)rowA - (uint32_t)ib) & 31u), rd = ib + (int)(((uint32_t)rowD - (uint32_t)ib) & 31u);
                            const int rlo = min(ra, rd), rhi = min(max(ra, rd), rlo + 29);
                            slot = atomicAdd(&steps_n[0], 1u);
                            if (slot < (uint32_t)kStepCap) steps[slot] = (uint32_t)j | ((uint32_t)rlo << 12) | ((uint32_t)(rhi - rlo) << 27);
                        }
                        u64 sm = __ballot(step && slot >= (uint32_t)kStepCap);
                        while (sm) {
                            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)sm) - 1);
                            sm &= sm - 1;
                            const int ra = ib + (int)(((uint32_t)MDVT_BI(rowA) - (uint32_t)ib) & 31u), rd = ib + (int)(((uint32_t)MDVT_BI(rowD) - (uint32_t)ib) & 31u);
                            step_whole_wave(MDVT_BI(j), min(ra, rd), max(ra, rd), k);
                        }
                    }
#undef MDVT_BI
#undef MDVT_BF
#undef MDVT_BU
                }
                // ---- the listed steps: the last wave to get here takes them all, one step per lane ----
                uint32_t arrived = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // this wave's list entries before its arrival
                if (lane == 0) arrived = atomicAdd(&steps_n[1], 1u);
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)arrived) == (uint32_t)(TPB / 64 - 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const uint32_t ns = min(steps_n[0], (uint32_t)kStepCap);
                    // eight steps per round, eight lanes per step: lane `sub` works out vertex (rlo + sub / 2, bj + (sub & 1)) -- one copy
                    // of the vertex programme --, the lane holding a cell's top-left vertex collects the other three and draws
                    // the cell (steps of more than three cell rows do not exist within the host's admission bound: whole wave)
                    for (uint32_t base = 0; base < ns; base += 8) {
                        const uint32_t si = base + (uint32_t)(lane >> 3);
                        const int sub = lane & 7;
                        const uint32_t e = si < ns ? steps[si] : 0u;
                        const int bj = (int)(e & 0xFFFu), rlo = (int)((e >> 12) & 0x7FFFu);
                        int nr = si < ns ? (int)(e >> 27) + 1 : 0;              // cell rows of the step
                        if (rlo + nr - 1 > H - 2) nr = H - 1 - rlo;
                        const bool wide_step = nr > 3;
                        int4 v = make_int4(0, 0, 0, 0);
                        if (!wide_step && sub < 2 * (nr + 1)) v = vertex_at(rlo + (sub >> 1), bj + (sub & 1));
                        int4 vD, vB, vC;
                        vD.x = __shfl_down(v.x, 1); vD.y = __shfl_down(v.y, 1); vD.z = __shfl_down(v.z, 1); vD.w = __shfl_down(v.w, 1);
                        vB.x = __shfl_down(v.x, 2); vB.y = __shfl_down(v.y, 2); vB.z = __shfl_down(v.z, 2); vB.w = __shfl_down(v.w, 2);
                        vC.x = __shfl_down(v.x, 3); vC.y = __shfl_down(v.y, 3); vC.z = __shfl_down(v.z, 3); vC.w = __shfl_down(v.w, 3);
                        bool left = false;
                        const int r = rlo + (sub >> 1);
                        if (!wide_step && !(sub & 1) && (sub >> 1) < nr) left = !cell_on_lane(v, vB, vC, vD, r, bj, k);
                        u64 lm = __ballot(left);                            // cells at a depth edge / near plane / out of range: generic path
                        while (lm) {
                            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                            lm &= lm - 1;
                            const int rr = __builtin_amdgcn_readlane(r, l);
                            step_whole_wave(__builtin_amdgcn_readlane(bj, l), rr, rr, k);
                        }
                        u64 wm = __ballot(wide_step && sub == 0);
                        while (wm) {
                            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)wm) - 1);
                            wm &= wm - 1;
                            const int rl = __builtin_amdgcn_readlane(rlo, l);
                            step_whole_wave(__builtin_amdgcn_readlane(bj, l), rl, rl + __builtin_amdgcn_readlane(nr, l) - 1, k);
                        }
                    }
                    if (lane < 2) steps_n[lane] = 0u;
                }
            }
            __syncthreads();
            if (ties.mode == 0) {
                if (ties.bits[ties.nwords] == 0u) break;
                row_ties_prepare(zb, W, ties, tid, TPB);
                __syncthreads();
            }
        }
        const bool had_ties = ties.mode == 3;        // the loop ran to its end (a row without ties leaves it at mode 0)
        ties.mode = 0;

        // the brackets of the next scanline (the ring's last readers were the raster passes above)
        if (k + 1 < k1 && stage_prefetched(Yc + kSubpix, ib) && !(MDVT_DEBUG_SKIP(a) & 4)) stage_rest(Yc + kSubpix, ib);

        // ---- resolve: LDS keys -> colour-key hole test -> coalesced stores; the keys are reset on the way ----
        if (act4 && !(MDVT_DEBUG_SKIP(a) & 2)) {
            uint4* zq = (uint4*)zb + 2 * tid;
            const uint4 k01 = zq[0], k23 = zq[1];
            zq[0] = make_uint4(~0u, ~0u, ~0u, ~0u); zq[1] = make_uint4(~0u, ~0u, ~0u, ~0u);
            const uint32_t hi[4] = {k01.y, k01.w, k23.y, k23.w};
            const uint32_t lo[4] = {k01.x, k01.z, k23.x, k23.z};
            uint32_t o[4], mw = 0, spx[4];
            float oz[4];
            const u64* erow = EDGEPTS ? a.ekeys[eye] + (size_t)fr * a.ws_stride_px + (size_t)k * W + 4 * tid : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool covered = !(hi[q] == ~0u && lo[q] == ~0u);          // (a settled tie has the top bit of hi cleared)
                const uint32_t rgb = lo[q] & 0xFFFFFFu;
                const bool hole = !covered || rgb == a.key_rgb;          // sr:740
                o[q] = hole ? 0u : rgb;                                  // sr:793
                mw |= hole ? (0xFFu << (8 * q)) : 0u;
                if (ZOUT) oz[q] = covered ? 1.0f / row_word_iz(hi[q]) : 0.0f;
                uint32_t esrc = ~0u;
                if (EDGEPTS && hole) {                                   // sr:776, 813-814: only where the render left a hole
                    const u64 ek = erow[q];
                    if (ek != kEmpty64) {
                        esrc = (uint32_t)ek;
                        if (a.edge_paint) o[q] = load_px_bytes(cbase + (size_t)(esrc >> 16) * a.color_pitch, (int)(esrc & 0xFFFFu));
                    }
                }
                if (SEED && a.seed[eye]) spx[q] = seed_pixel(a, fp, f, eye, 4 * tid + q, k, hole, esrc, 1);
            }
            if (SEED && a.seed[eye]) {
                uint32_t* sp = (uint32_t*)(a.seed[eye] + (size_t)f * a.seed_stride + (size_t)k * a.seed_pitch) + 3 * tid;
                sp[0] = __builtin_amdgcn_perm(spx[1], spx[0], 0x04020100u);
                sp[1] = __builtin_amdgcn_perm(spx[2], spx[1], 0x05040201u);
                sp[2] = __builtin_amdgcn_perm(spx[3], spx[2], 0x06050402u);
            }
            uint32_t* op = (uint32_t*)(a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)k * a.rgb_pitch) + 3 * tid;
            __builtin_nontemporal_store(__builtin_amdgcn_perm(o[1], o[0], 0x04020100u), op);
            __builtin_nontemporal_store(__builtin_amdgcn_perm(o[2], o[1], 0x05040201u), op + 1);
            __builtin_nontemporal_store(__builtin_amdgcn_perm(o[3], o[2], 0x06050402u), op + 2);
            __builtin_nontemporal_store(mw, (uint32_t*)(a.mask[eye] + (size_t)f * a.mask_stride + (size_t)k * a.mask_pitch) + tid);
            if (ZOUT && a.zout[eye]) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 v = {oz[0], oz[1], oz[2], oz[3]};
                __builtin_nontemporal_store(v, (f32x4*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)k * a.zout_pitch) + tid);
            }
        }
        if (had_ties) for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
        __syncthreads();
    }
}

size_t mesh_conv_lds_bytes(int W, int tpb)
{
    return (size_t)W * sizeof(u64) + 2 * (size_t)W * 16 + (size_t)(tpb / 64) * kQueueWave * sizeof(uint32_t) +
           (((size_t)W + 31) / 32 + 1) * sizeof(uint32_t) + (kStepCap + 2) * sizeof(uint32_t);
}

static int mesh_conv_tpb(int W)
{
    return (W / 4 <= 512 && 2 * mesh_conv_lds_bytes(W, 512) <= 160 * 1024) ? 512 : 1024;
}

// Can the convergence band kernel render this launch?  (4-byte addressable rows, LDS for one scanline + two vertex rows; the
// frames themselves are admitted by fill_frame_dev: FrameDev.conv_band.)
bool mesh_conv_supported(const RenderPlan& plan, const RenderArgs& a)
{
    if (!plan.vec4 || plan.mode != MDVT_MODE_MESH) return false;
    // Opt-in (MDVT_MESH_CONV=1): measured at 1080p, 32 frames per launch (profiles/r03_conv_band.md), the kernel moves 44 MB per
    // frame where the general path moves 438, but it is VALU bound like the general rasteriser and only draws level with it
    // without edge removal (89.7 against 91.7 us per frame) and loses with it (the product default: 123 against 91 us per
    // frame): the general vertex programme runs once per eye AND per band here.  The general path stays the default.
    if (!plan.allow_conv) return false;
    if (a.W < 8 || a.W > 4096 || a.H < 2) return false;
    return mesh_conv_lds_bytes(a.W, mesh_conv_tpb(a.W)) <= 160 * 1024;
}

template <int TPB>
static hipError_t launch_mesh_conv_tpb(const RenderPlan& plan, const RenderArgs& a, int rows, hipStream_t s)
{
    const size_t lds = mesh_conv_lds_bytes(a.W, TPB);
    const int nbands = (a.H + rows - 1) / rows;
    const dim3 grid((unsigned)(plan.n * nbands * 2)), block(TPB);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0) |
                      (plan.remove_edges && a.seed[0] ? 8 : 0);
#define MDVT_CASE(F)                                                                                                        \
    case F:                                                                                                                 \
        (void)hipFuncSetAttribute((const void*)k_mesh_conv<F, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        hipLaunchKernelGGL((k_mesh_conv<F, TPB>), grid, block, lds, s, a, rows, nbands);                                   \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        MDVT_CASE(10) MDVT_CASE(11) MDVT_CASE(14) MDVT_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

hipError_t launch_mesh_conv(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);
    int rows = 16;
    if (const char* e = tuning_env(TUNE_MESH_BAND)) { const int v = atoi(e); if (v > 0) rows = v; }     // tuning hook
    if (rows > a.H) rows = a.H;
    hipError_t e;
    const bool edge = plan.remove_edges && plan.edge_points;
    if (edge && (e = launch_edge_points_splat(a, plan.n, true, false, s)) != hipSuccess) return e;     // the edge keys this kernel's resolve reads
    e = mesh_conv_tpb(a.W) == 512 ? launch_mesh_conv_tpb<512>(plan, a, rows, s) : launch_mesh_conv_tpb<1024>(plan, a, rows, s);
    if (e != hipSuccess) return e;
    if (edge) return launch_edge_keys_reset(a, plan.n, s);
    return hipSuccess;
}

}  // namespace MDVT_GRID
}  // namespace mdvt
