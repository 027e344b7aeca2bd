// mdvt_mesh_band.hip -- MESH MODE, pure stereo shift: the reference's default draw mode (dmt:1243-1254 rendered by
// dmt:1422-1572 after sr:724-725 / 832-836), one workgroup per (frame, band of output rows), z-buffer in LDS.
//
// Same results as k_mesh_rows (mdvt_kernels.hip) and the oracle -- the arithmetic decree of DESIGN.md section 3 is
// untouched -- but organised around what the pure shift makes cheap:
//
//   * every vertex row is a horizontal line on screen (v = grid y), so output scanline k is covered by exactly one row
//     c(k) of grid cells, and a cell meets the scanline in the interval between its two column edges, split by the
//     diagonal A-C.  With tt = Yc - Yt, hh = Yb - Yt (uniform for the row) and
//         kcol0 = (XB-XA) tt + hh XA,   kdiag = (XC-XA) tt + hh XA,   kcol1 = (XC-XD) tt + hh XD
//     pixel centre Xc lies in tri1 = (A,B,C) iff kcol0 <= hh Xc < kdiag and in tri2 = (A,C,D) iff kdiag <= hh Xc < kcol1
//     (a cell mirrored by a fold: the same with the bounds swapped) -- the fill rule of the generic edge functions
//     written out for edges that start and end on the two vertex rows.  The covered pixel columns of a cell are
//     therefore [P(klo), P(khi)) with P(k) = ceil((k - 128 hh) / (256 hh)): an exact integer division by a
//     row-uniform constant instead of candidate pixels tested one by one, and the integer barycentric weights are
//     |differences| of the same three values.
//   * all of this fits 32-bit integers with 24-bit multiplies (full rate; 32x32 and 64-bit multiplies are quarter
//     rate) when the snapped coordinates are within +-2^20 sub-pixels (4096 px) -- cells outside that range, cells
//     with a vertex behind the near plane and twisted / zero-width cells take the generic 64-bit triangle path, one
//     cell at a time on the whole wave (rare).
//   * a thread owns a cell and shades its FIRST covered pixel itself; further pixels of a cell (stretched cells, the
//     rubber sheet across depth edges) go to a small per-wave stack in LDS as (cell, pixel) items that the wave
//     shades 64 at a time -- no lane waits while a neighbour walks a long span.
//   * a workgroup renders a band of output rows and keeps the two vertex rows it needs in an LDS ring, so each
//     vertex row is decoded once per band instead of once per output row (HBM reads 6 B/px * (R+1)/R), with the
//     next row's loads in flight while the current one is rasterised.
//   * divisions: correctly rounded through rcp + fma corrections (mdvt_device.h) instead of the generic expansion.
#include "mdvt_band_common.h"

#include <stdlib.h>
#include <type_traits>

namespace mdvt {
namespace MDVT_GRID {      // one copy per sub-pixel grid (mdvt_internal.h)

// FLAGS bit 0: depth planes; bit 1: triangles removed by the 89-degree filter draw nothing (remove_edges; tri_invalid /
// unused come from k_edge_filter); bit 2: the vertices of removed triangles are splatted into the holes (sr:589-606,
// 745-781); bit 3: the infill-mask seed image (sr:787-803).
// DBG: the ablation / test hooks of RenderArgs.debug_skip and face culling (mdvt_config.cull) are compiled in: the launcher picks
// it when a hook is set (tuning build only) or cull != 0; the common case carries none of their scalar state
// (`a` must stay the kernel's FIRST parameter and a plain-old-data struct: the body re-reads it through the kernarg segment at offset 0)
static_assert(std::is_trivially_copyable<RenderArgs>::value && std::is_standard_layout<RenderArgs>::value,
              "k_mesh_band reads RenderArgs through the kernarg segment: it must be passed as raw bytes");
template <int FLAGS, int TPB, bool DBG>
__global__ void __launch_bounds__(TPB, TPB == 1024 ? 4 : 4) k_mesh_band(RenderArgs a, int rows_per_band, int nbands)
{
    // The launch arguments are read where they are used, through the kernarg segment (s_load: scalar memory, no VALU slot) -- taken
    // by value they are all loaded at entry and live in ~60 scalar registers for the whole kernel, which has 102 and wants ~250: what
    // did not fit was spilled to VGPR lanes and came back through v_readlane, one VALU instruction each, in every row's prologue.
    // `a` is the kernel's first parameter: offset 0 of the segment.  KARGS(p) gives a pointer the compiler cannot see through, so the
    // loads behind it stay where they are written.
    typedef const __attribute__((address_space(4))) RenderArgs* KArgs;
    const KArgs ka0 = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
#define KARGS(p) KArgs p = ka0; asm volatile("" : "+s"(p))
    const int dbg = DBG ? MDVT_DEBUG_SKIP(a) : 0;
    const uint32_t cull = DBG ? (uint32_t)a.cull : 0u;
    constexpr bool ZOUT = FLAGS & 1, EDGES = FLAGS & 2, EDGEPTS = FLAGS & 4, SEED = FLAGS & 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W, H = a.H, W4 = W >> 2;
    u64* zb = (u64*)smem;                                   // [W] z keys of the eye being rendered
    int4* verts = (int4*)(zb + W);                          // [2][W]: vertex row i lives in slot i & 1
    uint32_t* queue = (uint32_t*)(verts + 2 * (size_t)W);   // [TPB/64][kQueueWave]
    RowTies ties;                                           // [W/32 + 1] exact-depth-tie bits + flag (mdvt_device.h)
    ties.bits = queue + (TPB / 64) * kQueueWave;
    ties.nwords = (W + 31) / 32;
    ties.mode = 0;
    ties.force = (dbg & 32) != 0;
    // With edge removal a vertex record's colour word carries three flags in its top byte: bit 24 / 25 = tri1 / tri2 of the
    // cell whose top-left corner the vertex is were removed (dmt:1372), bit 26 = the vertex belongs to a removed triangle
    // (its edge point is splatted, sr:589-606).  The edge-point keys of a (scanline, eye) (code16 << 16 | column, nearest wins)
    // live in a row of a global buffer, not in LDS (read back only where the render left a hole, reset by the lanes that wrote them): with them in LDS only one workgroup fits a CU, which costs this
    // barrier-heavy row loop a factor of two; the ~100 atomics and 8 B/px of extra traffic per row do not show.

    const int fr = blockIdx.x / nbands;
    const int band = blockIdx.x - fr * nbands;
    const int k0 = band * rows_per_band;
    const int k1 = min(k0 + rows_per_band, H);
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool dl_ok = fast_operand(fp.dl);
    uint32_t* wq = queue + wave * kQueueWave;
    const bool act4 = tid < W4;


    int have0 = -1, have1 = -1;                 // vertex row held by slot 0 / 1 (uniform)
    int pf_row = -1;                            // vertex row sitting in the prefetch registers (uniform)
    uint32_t pd0 = 0, pd1 = 0, pd2 = 0, pc0 = 0, pc1 = 0, pc2 = 0;
    uint32_t pfl = 0;                            // EDGES: the four columns' flags of that row, one byte each
    // EDGEPTS: this thread's edge points of the current (row, eye), one word each: depth code << 16 | (target column + 1), 0 = none
    // (their keys code16 << 16 | source column follow from it: the source column is tid + q * TPB)
    uint32_t ept[4] = {0, 0, 0, 0};

    auto fetch_row = [&](int r) {
        if (act4) {
            KARGS(kp);
            const uint32_t* dp = (const uint32_t*)(kp->depth + (size_t)f * kp->depth_stride + (size_t)r * kp->depth_pitch) + 3 * tid;
            const uint32_t* cp = (const uint32_t*)(kp->color + (size_t)f * kp->color_stride + (size_t)r * kp->color_pitch) + 3 * tid;
            pd0 = dp[0]; pd1 = dp[1]; pd2 = dp[2];
            pc0 = cp[0]; pc1 = cp[1]; pc2 = cp[2];
            if (EDGES) {
                const size_t ncell_ = (size_t)(W - 1) * (H - 1);
                const uint8_t* ti = kp->tri_invalid + (size_t)fr * kp->ws_stride_tri + (size_t)r * (W - 1) + 4 * tid;
                const uint8_t* ur = kp->unused + (size_t)fr * kp->ws_stride_px + (size_t)r * W + 4 * tid;
                pfl = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t fl = 0;
                    if (r <= H - 2 && 4 * tid + q < W - 1) fl = (ti[q] ? 1u : 0u) | (ti[ncell_ + q] ? 2u : 0u);
                    if (EDGEPTS && ur[q]) fl |= 4u;
                    pfl |= fl << (8 * q);
                }
            }
        }
        pf_row = r;
    };
    auto stage_row = [&](int r) {               // prefetch registers -> LDS vertex records
        if (pf_row != r) fetch_row(r);
        if (act4) {
            uint32_t dpx[4], cpx[4];
            unpack4(pd0, pd1, pd2, dpx);
            unpack4(pc0, pc1, pc2, cpx);
            if (EDGES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cpx[q] |= ((pfl >> (8 * q)) & 0xFFu) << 24;
            }
            int4* dst = verts + (size_t)(r & 1) * W + 4 * tid;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = band_vertex(dpx[q], cpx[q], 4 * tid + q, fp, dl_ok);
        }
        if (r & 1) have1 = r; else have0 = r;
        pf_row = -1;
    };

    for (int x = tid; x < W; x += TPB) zb[x] = kEmpty64;
    for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
    // The loop starts one row early: that prologue pass only stages the two vertex rows of scanline k0 (so that the
    // staging code exists once).
#pragma unroll 1
    for (int k = k0 - 1; k < k1; ++k) {
        // this scanline's row of cells, from the table every time (16 bytes of scalar memory): carried from the iteration before --
        // and the next row's beside it -- the geometry was eighteen scalar registers of loop state
        RowGeom g;
        int gnc = -1;                           // the next scanline's row of cells
        {
            KARGS(kt);
            g = row_geometry(k < k0 ? k0 : k, kt->rowcell);
            if (k < k0) g.c = -1;
            if (k + 1 < k1) gnc = kt->rowcell[k + 1].c;
        }
        // the vertex row the next scanline may need, in flight while this one is rasterised
        if (gnc >= 0) {
            const int need = ((gnc & 1) ? have1 : have0) != gnc ? gnc : gnc + 1;
            const int held = (need & 1) ? have1 : have0;
            if (held != need && pf_row != need) fetch_row(need);
        }
        // The row constants every lane computes with, in VECTOR registers (r04): the kernel needs ~250 scalar registers for its
        // arguments, loop state and lane masks and has 102; what did not fit was spilled to VGPR lanes and read back with
        // v_readlane -- a VALU instruction each, 22 of them per pass of the cell loop.  A VGPR operand costs nothing.
        RowGeom gv = g;
        int vt_i = (g.c & 1) * W, vb_i = ((g.c + 1) & 1) * W, Wv = W;       // (likewise: the two vertex rows' slots, the width)
        if (!EDGEPTS) {                      // (the edge-point variants sit at the 128-VGPR limit: there the scalars stay where they are)
            asm volatile("" : "+v"(gv.tt), "+v"(gv.bb), "+v"(gv.hh), "+v"(gv.D), "+v"(gv.c128), "+v"(gv.rD));
            asm volatile("" : "+v"(vt_i), "+v"(vb_i), "+v"(Wv));
        }

#pragma unroll 1
        for (int eye = 0; eye < 2; ++eye) {
            // (passes 1 and 2 only for a row with exact depth ties between different colours: RowTies in mdvt_device.h)
#pragma unroll 1
            for (ties.mode = 0; ties.mode < 3; ++ties.mode) {
            if (g.c >= 0 && !(dbg & 1)) {
                int qn = 0;                                          // items on this wave's stack (uniform)
                // A wave takes 63 consecutive cells per pass: lane l works out column c0 + l (its crossing of the scanline and
                // the first pixel at or after it) and owns the cell between its column and the next lane's, whose values
                // arrive through one DPP move each; lane 63 only provides that neighbour.  One more pass than there are
                // cells: the last one only empties the stack.
                constexpr int kCellsPerPass = 63 * (TPB / 64);
#pragma unroll 1
                for (int c0 = 0; ; c0 += kCellsPerPass) {
                    const bool final_pass = c0 >= W - 1;
                    const int j = c0 + wave * 63 + lane;
                    // (columns past the row end read the last column: never used, `cell` below masks them)
                    const int jc = j < Wv ? j : Wv - 1;
                    const int4 A = verts[vt_i + jc], B = verts[vb_i + jc];
                    const int XA = eye == 0 ? A.x : A.y, XB = eye == 0 ? B.x : B.y;
                    const float izA = __int_as_float(A.z), izB = __int_as_float(B.z);
                    const uint32_t cA = (uint32_t)A.w & 0xFFFFFFu, cB = (uint32_t)B.w & 0xFFFFFFu;
                    const int colok = (izA > 0.0f && izB > 0.0f && (((uint32_t)(XA + kCoordBound) | (uint32_t)(XB + kCoordBound)) >> 21) == 0u) ? 1 : 0;
                    const int kcol0 = mad24(XB - XA, gv.tt, mul24(gv.hh, XA));
                    const int pA = first_pixel(kcol0, gv.c128, gv.D, gv.rD, Wv);
                    const int XD = from_next_lane(XA), XC = from_next_lane(XB), kcol1 = from_next_lane(kcol0), pD = from_next_lane(pA);
                    const float izD = from_next_lane(izA), izC = from_next_lane(izB);
                    const uint32_t cD = from_next_lane(cA), cC = from_next_lane(cB);
                    const int okD = from_next_lane(colok);
                    const bool cell = lane < 63 && j < Wv - 1 && !final_pass;
                    const int s1 = XC - XB, s2 = XD - XA;
                    const bool regular = s2 > 0;
                    const bool fast = cell && colok && okD && ((s1 > 0 && s2 > 0) || (s1 < 0 && s2 < 0));
                    const bool exotic = cell && !fast;
                    // culling (mdvt_config.cull): the grid's own orientation is the front face
                    const uint32_t skip = EDGES ? ((uint32_t)A.w >> 24) & 3u : 0u;                        // dmt:1372
                    const bool drawn = fast && skip != 3u && !(cull && (cull == 1u) != regular);
                    const int plo = regular ? pA : pD;
                    int n = drawn ? (regular ? pD - pA : pA - pD) : 0;
                    if (n > 0 && !(dbg & 16))
                        cell_pixel(XA, XB, XC, XD, izA, izB, izC, izD, cA, cB, cC, cD, kcol0, kcol1, plo, j, skip, gv, zb, ties);
                    // Further pixels of the cell become (cell, pixel) items on the wave's stack, shaded 64 at a time: first
                    // one item per lane and round (spans of up to 4 px), then the long spans (rubber sheet across a depth
                    // edge), one cell at a time written by the whole wave.  One loop, so that the shading code exists once.
                    if (dbg & 8) n = 0;
                    u64 lm = __ballot(n > 4);
                    if (__ballot(n > 1) != 0ull || final_pass) {
                        int round = 1, lcell = 0, lpix = 0, lrem = 0;
                        for (;;) {
                            if (qn >= 64 || (final_pass && qn > 0)) {
                                const int cnt = qn >= 64 ? 64 : qn;
                                qn -= cnt;
                                if (lane < cnt) {
                                    const uint32_t it = wq[qn + lane];
                                    const int ij = (int)(it >> 16), px = (int)(it & 0xFFFFu);
                                    const int4 A = verts[vt_i + ij], D = verts[vt_i + ij + 1], B = verts[vb_i + ij], Cv = verts[vb_i + ij + 1];
                                    const int iXA = eye == 0 ? A.x : A.y, iXB = eye == 0 ? B.x : B.y, iXC = eye == 0 ? Cv.x : Cv.y, iXD = eye == 0 ? D.x : D.y;
                                    cell_pixel(iXA, iXB, iXC, iXD, __int_as_float(A.z), __int_as_float(B.z), __int_as_float(Cv.z), __int_as_float(D.z),
                                               (uint32_t)A.w & 0xFFFFFFu, (uint32_t)B.w & 0xFFFFFFu, (uint32_t)Cv.w & 0xFFFFFFu, (uint32_t)D.w & 0xFFFFFFu,
                                               mad24(iXB - iXA, gv.tt, mul24(gv.hh, iXA)), mad24(iXC - iXD, gv.tt, mul24(gv.hh, iXD)), px, ij,
                                               EDGES ? ((uint32_t)A.w >> 24) & 3u : 0u, gv, zb, ties);
                                }
                                continue;
                            }
                            if (round < 4) {
                                const bool want = n > round;
                                const u64 m = __ballot(want);
                                if (want) wq[qn + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)j << 16) | (uint32_t)(plo + round);
                                qn += (int)__popcll(m);
                                round = m ? round + 1 : 4;
                                continue;
                            }
                            if (lrem == 0 && lm) {
                                const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                                lm &= lm - 1;
                                lcell = __builtin_amdgcn_readlane(j, l);
                                lpix = __builtin_amdgcn_readlane(plo, l) + 4;
                                lrem = __builtin_amdgcn_readlane(n, l) - 4;
                            }
                            if (lrem > 0) {
                                const int cnt = lrem < 64 ? lrem : 64;
                                if (lane < cnt) wq[qn + lane] = ((uint32_t)lcell << 16) | (uint32_t)(lpix + lane);
                                qn += cnt; lpix += cnt; lrem -= cnt;
                                continue;
                            }
                            break;
                        }
                    }
                    if (final_pass) break;
                    // exotic cells (near plane, out of the 24-bit range, twisted, zero width): generic path, whole wave
                    u64 em = (dbg & 8) ? 0ull : __ballot(exotic);
                    while (em) {
                        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)em) - 1);
                        em &= em - 1;
#define MDVT_BI(v) __builtin_amdgcn_readlane(v, l)
#define MDVT_BF(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l))
#define MDVT_BU(v) (uint32_t)__builtin_amdgcn_readlane((int)(v), l)
                        uint32_t sk = MDVT_BU(skip);
                        const int bXA = MDVT_BI(XA), bXB = MDVT_BI(XB), bXC = MDVT_BI(XC), bXD = MDVT_BI(XD);
                        if (cull) {        // per triangle: area2 of tri1 = hh (XB - XC), of tri2 = -hh (XD - XA); front = negative
                            const bool back1 = bXB > bXC, back2 = bXA > bXD;
                            if ((cull == 1u) == back1 && bXB != bXC) sk |= 1u;
                            if ((cull == 1u) == back2 && bXA != bXD) sk |= 2u;
                        }
                        exotic_cell_wave(bXA, bXB, bXC, bXD, MDVT_BF(izA), MDVT_BF(izB), MDVT_BF(izC), MDVT_BF(izD),
                                         MDVT_BU(cA), MDVT_BU(cB), MDVT_BU(cC), MDVT_BU(cD), sk, g.Yt, g.Yb, k, W, lane, MDVT_BI(j), zb, ties);      // (cA.. are already masked)
#undef MDVT_BI
#undef MDVT_BF
#undef MDVT_BU
                    }
                }
            }
            // ---- edge points of source row k (sr:589-606, 745-781): the vertices of removed triangles, nearest wins.  They land
            //      on scanline k itself, the scanline this workgroup is rendering: their keys go into the z-buffer row's own
            //      memory once its words have been read out (below); here only position and key are worked out ----
            // (the edge points of scanlines erow_lo .. erow_hi are k_edge_rows_exact's: their row is not the source row)
            if (EDGEPTS && ties.mode == 0 && k >= k0 && !edge_row_deferred(fp, k)) {
                KARGS(ke);
                const uint8_t* drow_k = ke->depth + (size_t)f * ke->depth_stride + (size_t)k * ke->depth_pitch;
                const float guard = edge_col_guard(W);
                // (source row k is one of the two staged vertex rows: c(k) is k or k - 1)
                const int4* vk = ((k & 1) ? have1 : have0) == k ? verts + (size_t)(k & 1) * W : nullptr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                          // W <= 4 * TPB (launcher)
                    const int jj = tid + q * TPB;
                    ept[q] = 0u;
                    if (jj >= W) continue;
                    const bool un = vk ? (((uint32_t)vk[jj].w >> 26) & 1u) != 0u
                                       : ke->unused[(size_t)fr * ke->ws_stride_px + (size_t)k * W + jj] != 0;
                    if (!un) continue;
                    const uint32_t code = code16_of(load_px_bytes(drow_k, jj));
                    const float z = decode_z(code, fp.mult, fp.scale);
                    if (!(z > kNear)) continue;
                    const int x = edge_col_pure(fp, eye, (float)jj * fp.sx, z, fp.dl / z, W, guard);      // sr:599-600, 746
                    if (x >= 0) ept[q] = (code << 16) | (uint32_t)(x + 1);
                }
            } else if (EDGEPTS && ties.mode == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ept[q] = 0u;
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

            // the next scanline's vertex row replaces the one no longer needed (its last reader was the raster above)
            if (eye == 1 && gnc >= 0) {
#pragma unroll 1
                for (int r = gnc; r <= gnc + 1; ++r)
                    if (((r & 1) ? have1 : have0) != r) stage_row(r);
            }

            // ---- resolve this eye: LDS keys -> colour-key hole test -> coalesced stores; the keys are reset on the way ----
            const bool resolving = act4 && k >= k0 && !(dbg & 2);
            KARGS(kr);
            uint4 k01 = make_uint4(~0u, ~0u, ~0u, ~0u), k23 = k01, ek4 = k01;
            if (resolving) {
                uint4* zq = (uint4*)zb + 2 * tid;
                k01 = zq[0]; k23 = zq[1];
                zq[0] = make_uint4(~0u, ~0u, ~0u, ~0u); zq[1] = make_uint4(~0u, ~0u, ~0u, ~0u);
            }
            if (EDGEPTS && k >= k0) {            // (workgroup-uniform)
                // the row's words are in registers and EMPTY in LDS: its first W dwords now hold the edge keys, ds_min_u32 each,
                // are read back four per thread and left EMPTY again by their readers
                uint32_t* eb = reinterpret_cast<uint32_t*>(zb);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ept[q] & 0xFFFFu) atomicMin(&eb[(ept[q] & 0xFFFFu) - 1u], (ept[q] & 0xFFFF0000u) | (uint32_t)(tid + q * TPB));
                __syncthreads();
                if (resolving) {
                    uint4* eq = reinterpret_cast<uint4*>(eb) + tid;
                    ek4 = *eq;
                    *eq = make_uint4(~0u, ~0u, ~0u, ~0u);
                }
            }
            if (resolving) {
                const uint32_t hi[4] = {k01.y, k01.w, k23.y, k23.w};
                const uint32_t lo[4] = {k01.x, k01.z, k23.x, k23.z};
                uint32_t o[4], mw = 0, spx[4];
                float oz[4];
                const uint32_t ek[4] = {ek4.x, ek4.y, ek4.z, ek4.w};
                const uint8_t* crow_k = kr->color + (size_t)f * kr->color_stride + (size_t)k * kr->color_pitch;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool covered = !(hi[q] == ~0u && lo[q] == ~0u);          // (a settled tie has the top bit of hi cleared)
                    const uint32_t rgb = lo[q] & 0xFFFFFFu;
                    const bool hole = !covered || rgb == kr->key_rgb;          // sr:740
                    o[q] = hole ? 0u : rgb;                                  // sr:793
                    mw |= hole ? (0xFFu << (8 * q)) : 0u;
                    if (ZOUT) oz[q] = covered ? 1.0f / row_word_iz(hi[q]) : 0.0f;
                    uint32_t esrc = ~0u;
                    if (EDGEPTS && hole && ek[q] != kEmpty32) {              // sr:776, 813-814: only where the render left a hole
                        if (kr->edge_paint) o[q] = load_px_bytes(crow_k, (int)(ek[q] & 0xFFFFu));
                        esrc = ((uint32_t)k << 16) | (ek[q] & 0xFFFFu);
                    }
                    if (SEED && kr->seed[eye]) spx[q] = seed_pixel(a, fp, f, eye, 4 * tid + q, k, hole, esrc, 1);
                }
                if (SEED && kr->seed[eye]) {
                    uint32_t* sp = (uint32_t*)(kr->seed[eye] + (size_t)f * kr->seed_stride + (size_t)k * kr->seed_pitch) + 3 * tid;
                    sp[0] = __builtin_amdgcn_perm(spx[1], spx[0], 0x04020100u);
                    sp[1] = __builtin_amdgcn_perm(spx[2], spx[1], 0x05040201u);
                    sp[2] = __builtin_amdgcn_perm(spx[3], spx[2], 0x06050402u);
                }
                uint32_t* op = (uint32_t*)(kr->rgb[eye] + (size_t)f * kr->rgb_stride + (size_t)k * kr->rgb_pitch) + 3 * tid;
                __builtin_nontemporal_store(__builtin_amdgcn_perm(o[1], o[0], 0x04020100u), op);
                __builtin_nontemporal_store(__builtin_amdgcn_perm(o[2], o[1], 0x05040201u), op + 1);
                __builtin_nontemporal_store(__builtin_amdgcn_perm(o[3], o[2], 0x06050402u), op + 2);
                __builtin_nontemporal_store(mw, (uint32_t*)(kr->mask[eye] + (size_t)f * kr->mask_stride + (size_t)k * kr->mask_pitch) + tid);
                if (ZOUT && kr->zout[eye]) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 v = {oz[0], oz[1], oz[2], oz[3]};
                    __builtin_nontemporal_store(v, (f32x4*)((uint8_t*)kr->zout[eye] + (size_t)f * kr->zout_stride + (size_t)k * kr->zout_pitch) + tid);
                }
            }
            if (had_ties) for (int x = tid; x <= ties.nwords; x += TPB) ties.bits[x] = 0u;
            __syncthreads();
        }
    }
#undef KARGS
}

size_t mesh_band_lds_bytes(int W, int tpb, bool, bool)
{
    return (size_t)W * sizeof(u64) + 2 * (size_t)W * 16 + (size_t)(tpb / 64) * kQueueWave * sizeof(uint32_t) +
           (((size_t)W + 31) / 32 + 1) * sizeof(uint32_t);
}

static int mesh_band_tpb(const RenderPlan& plan, int W)
{
    // two 512-thread workgroups per CU when two fit in the 160 KB LDS, otherwise one of 1024 threads: 16 waves per CU either way
    return (W / 4 <= 512 && 2 * mesh_band_lds_bytes(W, 512, plan.remove_edges, plan.edge_points) <= 160 * 1024) ? 512 : 1024;
}

// Can the band kernel render this launch?  (4-byte aligned rows, LDS for one row.)
bool mesh_band_supported(const RenderPlan& plan, const RenderArgs& a)
{
    if (!plan.vec4 || plan.general) return false;
    if (a.W < 8 || a.W > 4096 || a.W > 4 * 1024) return false;       // 1024 threads x 4 px; the 24-bit fast path assumes W*256 <= 2^20
    return mesh_band_lds_bytes(a.W, mesh_band_tpb(plan, a.W), plan.remove_edges, plan.edge_points) <= 160 * 1024;
}

template <int TPB>
static hipError_t launch_mesh_band_tpb(const RenderPlan& plan, const RenderArgs& a, int rows, hipStream_t s)
{
    size_t lds = mesh_band_lds_bytes(a.W, TPB, plan.remove_edges, plan.edge_points);
    if (const char* e = tuning_env(TUNE_LDS_PAD)) lds += (size_t)atoi(e);      // occupancy probe (tools/kbench.py)
    const int nbands = (a.H + rows - 1) / rows;
    const dim3 grid((unsigned)(plan.n * nbands)), block(TPB);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0) |
                      (plan.remove_edges && a.seed[0] ? 8 : 0);
#define MDVT_CASE(F)                                                                                                        \
    case F:                                                                                                                 \
        if (MDVT_DEBUG_SKIP(a) || a.cull) {                                                                                       \
            (void)hipFuncSetAttribute((const void*)k_mesh_band<F, TPB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL((k_mesh_band<F, TPB, true>), grid, block, lds, s, a, rows, nbands);                               \
        } else {                                                                                                            \
            (void)hipFuncSetAttribute((const void*)k_mesh_band<F, TPB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL((k_mesh_band<F, TPB, false>), grid, block, lds, s, a, rows, nbands);                              \
        }                                                                                                                   \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        MDVT_CASE(10) MDVT_CASE(11) MDVT_CASE(14) MDVT_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

hipError_t launch_mesh_band(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);
    // scanlines per band: 8 for a launch of 32 frames (a band stages its own vertex rows: 9 for 8 scanlines); fewer when the launch is
    // small, so that a single 1080p frame is still ~1000 workgroups -- at 8 it was 135 for 512 slots: 127 us per frame, now 59;
    // 2 / 4 / 8 / 16 frames: 148 -> 100, 219 -> 170, 350 -> 317, 614 -> 598 us per launch (tools/kbench.py --mesh --frames N)
    int rows = (2 * plan.n * a.H + 2880) / (2 * 2880);
    rows = rows < 1 ? 1 : (rows > 8 ? 8 : rows);
    if (const char* e = tuning_env(TUNE_MESH_BAND)) { const int v = atoi(e); if (v > 0) rows = v; }     // tuning hook
    if (rows > a.H) rows = a.H;
    if (mesh_band_tpb(plan, a.W) == 512) return launch_mesh_band_tpb<512>(plan, a, rows, s);
    return launch_mesh_band_tpb<1024>(plan, a, rows, s);
}

}  // namespace MDVT_GRID
}  // namespace mdvt
