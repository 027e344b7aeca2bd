// mdvt_mesh_band3.hip -- MESH MODE, pure stereo shift (the reference's default draw mode: dmt:1243-1254 rendered by
// dmt:1422-1572 after sr:724-725 / 832-836): k_mesh_band's arithmetic with THREE workgroups per CU.
//
// k_mesh_band (mdvt_mesh_band.hip) keeps the two vertex rows a scanline needs as 16-byte records in LDS: 79 KB per 1080p
// workgroup, two workgroups per CU, four waves per SIMD -- and 46 % of its wave cycles wait (barriers between a scanline's
// phases, returning LDS atomics).  A third workgroup was measured to be worth 20 %, so this kernel
// gives the vertex rows up:
//   * no vertex records in LDS: a lane fetches the two source pixels of its column (rows c and c + 1 of depth and colour,
//     four unaligned dword loads, one pass ahead of their use) and decodes them itself -- 1/Z, dl/Z, both eyes' snapped x --,
//     ONCE for both eyes: the eye loop sits inside the pass, both eyes' z-buffer rows live in LDS (2 x 8 B x W = 31 KB at
//     1080p; with the wave stacks 35 KB);
//   * half the barriers: raster both eyes | resolve both eyes | next scanline;
//   * the (cell, pixel) items of stretched cells carry their eye and re-fetch their cell's four vertices when they are
//     shaded (a few per cent of the fragments).
// Everything per cell -- coverage intervals, weights, shading, exact depth ties, exotic cells -- is mdvt_band_common.h's,
// shared with k_mesh_band; the results are the same bits (tests/test_gpu_render.py runs both).
#include "mdvt_band_common.h"

#include <stdlib.h>

namespace mdvt {
namespace MDVT_GRID {      // one copy per sub-pixel grid (mdvt_internal.h)

namespace {

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// Pixel j of an interleaved u8 RGB row as R | G << 8 | B << 16: one dword load that stays inside the row's bytes (for
// j > 0 the dword starts one byte early; column 0 reads one byte of column 1 -- W >= 2).
__device__ __forceinline__ uint32_t load_px_dword(const uint8_t* row, int j)
{
    const uint32_t w = *(const u32_unaligned*)(row + (3 * j - (j > 0 ? 1 : 0)));
    return j > 0 ? w >> 8 : w & 0xFFFFFFu;
}

}  // namespace

// FLAGS as k_mesh_band: bit 0 depth planes; bit 1 triangles removed by the 89-degree filter draw nothing; bit 2 the vertices
// of removed triangles are splatted into the holes; bit 3 the infill-mask seed image.
template <int FLAGS, int TPB>
__global__ void __launch_bounds__(TPB, TPB == 1024 ? 4 : 6) k_mesh_band3(RenderArgs a, int rows_per_band, int nbands)
{
    constexpr bool ZOUT = FLAGS & 1, EDGES = FLAGS & 2, EDGEPTS = FLAGS & 4, SEED = FLAGS & 8;
    const uint32_t cull = (uint32_t)a.cull;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W, H = a.H, W4 = W >> 2;
    u64* zb2 = (u64*)smem;                                  // [2][W] z keys, left eye then right eye
    uint32_t* queue = (uint32_t*)(zb2 + 2 * (size_t)W);     // [TPB/64][kQueueWave]
    const int nwords = (W + 31) / 32;
    uint32_t* tbits = queue + (TPB / 64) * kQueueWave;      // per eye [W/32 + 1] exact-depth-tie bits + flag (RowTies, mdvt_device.h)
    const bool tforce = (MDVT_DEBUG_SKIP(a) & 32) != 0;
    // (built where it is used: an array of RowTies indexed by the eye would live in scratch memory)
    auto ties_of = [&](int eye, int mode) { RowTies t; t.bits = tbits + eye * (nwords + 1); t.nwords = nwords; t.mode = mode; t.force = tforce; return t; };

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

    const uint8_t* dbase = a.depth + (size_t)f * a.depth_stride;
    const uint8_t* cbase = a.color + (size_t)f * a.color_stride;
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    const uint8_t* tibase = EDGES ? a.tri_invalid + (size_t)fr * a.ws_stride_tri : nullptr;

    for (int x = tid; x < 2 * W; x += TPB) zb2[x] = kEmpty64;
    for (int x = tid; x < 2 * (nwords + 1); x += TPB) tbits[x] = 0u;
    __syncthreads();

    int etx[4] = {-1, -1, -1, -1};               // EDGEPTS: where this thread's edge points of the current (row, eye) land,
    uint32_t ekey[4] = {0, 0, 0, 0};             //          and their keys code16 << 16 | source column

#pragma unroll 1
    for (int k = k0; k < k1; ++k) {
        const RowGeom g = row_geometry(k, a.rowcell);
        const uint8_t* dA = dbase + (size_t)(g.c < 0 ? 0 : g.c) * a.depth_pitch;       // vertex row c and, one pitch on, c + 1
        const uint8_t* cA = cbase + (size_t)(g.c < 0 ? 0 : g.c) * a.color_pitch;
        const uint8_t* tiA = EDGES ? tibase + (size_t)(g.c < 0 ? 0 : g.c) * (W - 1) : nullptr;
        // the two triangles of cell (c, j) removed by the 89-degree filter (dmt:1372): bit 0 tri1, bit 1 tri2
        auto cell_skip = [&](int j) -> uint32_t {
            if (!EDGES || j >= W - 1) return 0u;
            return (tiA[j] ? 1u : 0u) | (tiA[ncell + j] ? 2u : 0u);
        };
        constexpr int kCellsPerPass = 63 * (TPB / 64);
        bool serve0 = true, serve1 = true;           // which eyes the current raster pass serves (mode 0: both)
        bool had_ties = false;

#pragma unroll 1
        for (int mode = 0; mode < 3; ++mode) {
            if (g.c >= 0 && !(MDVT_DEBUG_SKIP(a) & 1)) {
                int qn = 0;                                          // items on this wave's stack (uniform)
                // the first pass's source pixels
                int jn = wave * 63 + lane;
                jn = jn < W ? jn : W - 1;
                uint32_t pdA = load_px_dword(dA, jn), pdB = load_px_dword(dA + a.depth_pitch, jn);
                uint32_t pcA = load_px_dword(cA, jn), pcB = load_px_dword(cA + a.color_pitch, jn);
                uint32_t psk = cell_skip(jn);
#pragma unroll 1
                for (int c0 = 0; ; c0 += kCellsPerPass) {
                    const bool final_pass = c0 >= W - 1;
                    const int j = c0 + wave * 63 + lane;
                    // (columns past the row end read the last column: never used, `cell` below masks them)
                    const int jc = j < W ? j : W - 1;
                    const int4 A = band_vertex(pdA, pcA, jc, fp, dl_ok), B = band_vertex(pdB, pcB, jc, fp, dl_ok);
                    const uint32_t skip = psk;
                    if (c0 + kCellsPerPass < W - 1) {        // the next pass's pixels, in flight during this one
                        jn = j + kCellsPerPass;
                        jn = jn < W ? jn : W - 1;
                        pdA = load_px_dword(dA, jn); pdB = load_px_dword(dA + a.depth_pitch, jn);
                        pcA = load_px_dword(cA, jn); pcB = load_px_dword(cA + a.color_pitch, jn);
                        psk = cell_skip(jn);
                    }
                    const float izA = __int_as_float(A.z), izB = __int_as_float(B.z);
                    const uint32_t cAc = (uint32_t)A.w, cBc = (uint32_t)B.w;
                    const bool cellok = lane < 63 && j < W - 1 && !final_pass;
#pragma unroll 1
                    for (int eye = 0; eye < 2; ++eye) {
                        if (!(eye ? serve1 : serve0)) continue;                          // (uniform)
                        u64* zb = zb2 + (size_t)eye * W;
                        const RowTies tie = ties_of(eye, mode);
                        const float izD = from_next_lane(izA), izC = from_next_lane(izB);
                        const uint32_t cD = from_next_lane(cAc), cC = from_next_lane(cBc);
                        const int XA = eye == 0 ? A.x : A.y, XB = eye == 0 ? B.x : B.y;
                        const int colok = (izA > 0.0f && izB > 0.0f && (((uint32_t)(XA + kCoordBound) | (uint32_t)(XB + kCoordBound)) >> 21) == 0u) ? 1 : 0;
                        const int kcol0 = mad24(XB - XA, g.tt, mul24(g.hh, XA));
                        const int pA = first_pixel(kcol0, g.c128, g.D, g.rD, W);
                        const int XD = from_next_lane(XA), XC = from_next_lane(XB), kcol1 = from_next_lane(kcol0), pD = from_next_lane(pA);
                        const int okD = from_next_lane(colok);
                        const int s1 = XC - XB, s2 = XD - XA;
                        const bool regular = s2 > 0;
                        const bool fast = cellok && colok && okD && ((s1 > 0 && s2 > 0) || (s1 < 0 && s2 < 0));
                        const bool exotic = cellok && !fast;
                        // culling (mdvt_config.cull): the grid's own orientation is the front face
                        const bool drawn = fast && skip != 3u && !(cull && (cull == 1u) != regular);
                        const int plo = regular ? pA : pD;
                        int n = drawn ? (regular ? pD - pA : pA - pD) : 0;
                        if (n > 0 && !(MDVT_DEBUG_SKIP(a) & 16))
                            cell_pixel(XA, XB, XC, XD, izA, izB, izC, izD, cAc, cBc, cC, cD, kcol0, kcol1, plo, j, skip, g, zb, tie);
                        // Further pixels of the cell become (eye, cell, pixel) items on the wave's stack, shaded 64 at a time: first
                        // one item per lane and round (spans of up to 4 px), then the long spans (rubber sheet across a depth
                        // edge), one cell at a time written by the whole wave.  One loop, so that the shading code exists once.
                        if (MDVT_DEBUG_SKIP(a) & 8) n = 0;
                        u64 lm = __ballot(n > 4);
                        const bool last = final_pass && eye == (serve1 ? 1 : 0);        // the last eye this pass serves: the stack is emptied
                        if (__ballot(n > 1) != 0ull || last) {
                            int round = 1, lcell = 0, lpix = 0, lrem = 0;
                            for (;;) {
                                if (qn >= 64 || (last && qn > 0)) {
                                    const int cnt = qn >= 64 ? 64 : qn;
                                    qn -= cnt;
                                    if (lane < cnt) {
                                        const uint32_t it = wq[qn + lane];
                                        const int ie = (int)(it >> 31), ij = (int)((it >> 16) & 0x7FFFu), px = (int)(it & 0xFFFFu);
                                        const int4 vA = band_vertex(load_px_dword(dA, ij), load_px_dword(cA, ij), ij, fp, dl_ok);
                                        const int4 vD = band_vertex(load_px_dword(dA, ij + 1), load_px_dword(cA, ij + 1), ij + 1, fp, dl_ok);
                                        const int4 vB = band_vertex(load_px_dword(dA + a.depth_pitch, ij), load_px_dword(cA + a.color_pitch, ij), ij, fp, dl_ok);
                                        const int4 vC = band_vertex(load_px_dword(dA + a.depth_pitch, ij + 1), load_px_dword(cA + a.color_pitch, ij + 1), ij + 1, fp, dl_ok);
                                        const int iXA = ie == 0 ? vA.x : vA.y, iXB = ie == 0 ? vB.x : vB.y, iXC = ie == 0 ? vC.x : vC.y, iXD = ie == 0 ? vD.x : vD.y;
                                        cell_pixel(iXA, iXB, iXC, iXD, __int_as_float(vA.z), __int_as_float(vB.z), __int_as_float(vC.z), __int_as_float(vD.z),
                                                   (uint32_t)vA.w, (uint32_t)vB.w, (uint32_t)vC.w, (uint32_t)vD.w,
                                                   mad24(iXB - iXA, g.tt, mul24(g.hh, iXA)), mad24(iXC - iXD, g.tt, mul24(g.hh, iXD)), px, ij,
                                                   cell_skip(ij), g, zb2 + (size_t)ie * W, ties_of(ie, mode));
                                    }
                                    continue;
                                }
                                if (round < 4) {
                                    const bool want = n > round;
                                    const u64 m = __ballot(want);
                                    if (want) wq[qn + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)eye << 31) | ((uint32_t)j << 16) | (uint32_t)(plo + round);
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
                                    if (lane < cnt) wq[qn + lane] = ((uint32_t)eye << 31) | ((uint32_t)lcell << 16) | (uint32_t)(lpix + lane);
                                    qn += cnt; lpix += cnt; lrem -= cnt;
                                    continue;
                                }
                                break;
                            }
                        }
                        if (final_pass) continue;
                        // exotic cells (near plane, out of the 24-bit range, twisted, zero width): generic path, whole wave
                        u64 em = (MDVT_DEBUG_SKIP(a) & 8) ? 0ull : __ballot(exotic);
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
                                             MDVT_BU(cAc), MDVT_BU(cBc), MDVT_BU(cC), MDVT_BU(cD), sk, g.Yt, g.Yb, k, W, lane, MDVT_BI(j), zb, tie);
#undef MDVT_BI
#undef MDVT_BF
#undef MDVT_BU
                        }
                    }
                    if (final_pass) break;
                }
            }
            __syncthreads();
            if (mode == 0) {
                // (passes 1 and 2 only for the eyes of a row with exact depth ties between different colours: RowTies in mdvt_device.h)
                serve0 = tbits[nwords] != 0u;
                serve1 = tbits[2 * nwords + 1] != 0u;
                if (!serve0 && !serve1) break;
                had_ties = true;
                if (serve0) row_ties_prepare(zb2, W, ties_of(0, 0), tid, TPB);
                if (serve1) row_ties_prepare(zb2 + W, W, ties_of(1, 0), tid, TPB);
                __syncthreads();
            }
        }

        // ---- resolve both eyes: LDS keys -> colour-key hole test -> coalesced stores; the keys are reset on the way ----
        const bool resolving = act4 && !(MDVT_DEBUG_SKIP(a) & 2);
        const bool edge_row = EDGEPTS && !edge_row_deferred(fp, k);     // (scanlines erow_lo .. erow_hi: k_edge_rows_exact)
#pragma unroll 1
        for (int eye = 0; eye < 2; ++eye) {
            u64* zb = zb2 + (size_t)eye * W;
            // ---- edge points of source row k (sr:589-606, 745-781): the vertices of removed triangles, nearest wins.  They land
            //      on scanline k itself; their keys go into the z-buffer row's own memory once its words have been read out ----
            if (EDGEPTS) {
                const uint8_t* drow_k = dbase + (size_t)k * a.depth_pitch;
                const uint8_t* urow_k = a.unused + (size_t)fr * a.ws_stride_px + (size_t)k * W;
                const float guard = edge_col_guard(W);
#pragma unroll
                for (int q = 0; q < 4; ++q) {                          // W <= 4 * TPB (launcher)
                    const int jj = tid + q * TPB;
                    etx[q] = -1;
                    if (!edge_row || jj >= W || !urow_k[jj]) continue;
                    const uint32_t code = code16_of(load_px_bytes(drow_k, jj));
                    const float z = decode_z(code, fp.mult, fp.scale);
                    if (!(z > kNear)) continue;
                    const int x = edge_col_pure(fp, eye, (float)jj * fp.sx, z, fp.dl / z, W, guard);      // sr:599-600, 746
                    if (x >= 0) { etx[q] = x; ekey[q] = (code << 16) | (uint32_t)jj; }
                }
            }
            uint4 k01 = make_uint4(~0u, ~0u, ~0u, ~0u), k23 = k01, ek4 = k01;
            if (resolving) {
                uint4* zq = (uint4*)zb + 2 * tid;
                k01 = zq[0]; k23 = zq[1];
                zq[0] = make_uint4(~0u, ~0u, ~0u, ~0u); zq[1] = make_uint4(~0u, ~0u, ~0u, ~0u);
            }
            if (EDGEPTS) {            // (workgroup-uniform)
                // the row's words are in registers and EMPTY in LDS: its first W dwords now hold the edge keys, ds_min_u32 each,
                // are read back four per thread and left EMPTY again by their readers
                uint32_t* eb = reinterpret_cast<uint32_t*>(zb);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) if (etx[q] >= 0) atomicMin(&eb[etx[q]], ekey[q]);
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
                const uint8_t* crow_k = cbase + (size_t)k * a.color_pitch;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool covered = !(hi[q] == ~0u && lo[q] == ~0u);          // (a settled tie has the top bit of hi cleared)
                    const uint32_t rgb = lo[q] & 0xFFFFFFu;
                    const bool hole = !covered || rgb == a.key_rgb;          // sr:740
                    o[q] = hole ? 0u : rgb;                                  // sr:793
                    mw |= hole ? (0xFFu << (8 * q)) : 0u;
                    if (ZOUT) oz[q] = covered ? 1.0f / row_word_iz(hi[q]) : 0.0f;
                    uint32_t esrc = ~0u;
                    if (EDGEPTS && hole && ek[q] != kEmpty32) {              // sr:776, 813-814: only where the render left a hole
                        if (a.edge_paint) o[q] = load_px_bytes(crow_k, (int)(ek[q] & 0xFFFFu));
                        esrc = ((uint32_t)k << 16) | (ek[q] & 0xFFFFu);
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
        }
        if (had_ties) for (int x = tid; x < 2 * (nwords + 1); x += TPB) tbits[x] = 0u;
        __syncthreads();
    }
}

size_t mesh_band3_lds_bytes(int W, int tpb)
{
    return 2 * (size_t)W * sizeof(u64) + (size_t)(tpb / 64) * kQueueWave * sizeof(uint32_t) + 2 * (((size_t)W + 31) / 32 + 1) * sizeof(uint32_t);
}

static int mesh_band3_tpb(int W) { return W / 4 <= 512 ? 512 : 1024; }

// Can the three-workgroup band kernel render this launch?  (The same frames as k_mesh_band; the item words hold a 15-bit cell column.)
bool mesh_band3_supported(const RenderPlan& plan, const RenderArgs& a)
{
    if (!plan.vec4 || plan.general) return false;
    if (a.W < 8 || a.W > 4096) return false;       // 1024 threads x 4 px; the 24-bit fast path assumes W*256 <= 2^20
    return mesh_band3_lds_bytes(a.W, mesh_band3_tpb(a.W)) <= 160 * 1024;
}

template <int TPB>
static hipError_t launch_mesh_band3_tpb(const RenderPlan& plan, const RenderArgs& a, int rows, hipStream_t s)
{
    size_t lds = mesh_band3_lds_bytes(a.W, TPB);
    if (const char* e = tuning_env(TUNE_LDS_PAD)) lds += (size_t)atoi(e);      // occupancy probe (tools/kbench.py)
    const int nbands = (a.H + rows - 1) / rows;
    const dim3 grid((unsigned)(plan.n * nbands)), block(TPB);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0) |
                      (plan.remove_edges && a.seed[0] ? 8 : 0);
#define MDVT_CASE(F)                                                                                                         \
    case F:                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)k_mesh_band3<F, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        hipLaunchKernelGGL((k_mesh_band3<F, TPB>), grid, block, lds, s, a, rows, nbands);                                   \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        MDVT_CASE(10) MDVT_CASE(11) MDVT_CASE(14) MDVT_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

hipError_t launch_mesh_band3(const RenderPlan& plan, const RenderArgs& a_in, hipStream_t s)
{
    RenderArgs a = a_in;
    if (const char* e = tuning_env(TUNE_DEBUG_SKIP)) a.debug_skip = atoi(e);
    int rows = 8;
    if (const char* e = tuning_env(TUNE_MESH_BAND)) { const int v = atoi(e); if (v > 0) rows = v; }     // tuning hook
    if (rows > a.H) rows = a.H;
    if (mesh_band3_tpb(a.W) == 512) return launch_mesh_band3_tpb<512>(plan, a, rows, s);
    return launch_mesh_band3_tpb<1024>(plan, a, rows, s);
}

}  // namespace MDVT_GRID
}  // namespace mdvt
